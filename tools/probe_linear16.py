#!/usr/bin/env python3
"""16-bit token GEMMs (1x1): persistent linear kernel vs the one-tile-per-workgroup kernel (DIFFSAL_NO_PERSIST=1).  GPU only."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diff_sal_amd import _lib, ops  # noqa: E402
from tools.tune_igemm16 import timed  # noqa: E402

dt = {"bf16": torch.bfloat16, "fp16": torch.float16}[sys.argv[1] if len(sys.argv) > 1 else "bf16"]
for M, K, N in [(193536, 96, 96), (193536, 96, 192), (193536, 192, 96), (48384, 192, 192), (48384, 192, 384), (48384, 384, 192),
                (12096, 384, 384), (12096, 384, 768), (12096, 768, 384), (3024, 768, 768), (3024, 768, 1536), (3024, 1536, 768),
                (648, 768, 768), (21504, 96, 192), (1000, 160, 72), (130, 96, 100)]:
    x = torch.randn(M, K, device="cuda").to(dt)
    w = (torch.randn(N, K, device="cuda") * 0.05).to(dt)
    b = torch.randn(N, device="cuda")
    r = torch.randn(M, N, device="cuda").to(dt)
    _lib.set_tuning("DIFFSAL_NO_PERSIST", 1)
    y0 = ops.linear(x, w, b, residual=r, act=ops.ACT_NONE)
    t0 = timed(lambda: ops.linear(x, w, b, residual=r))
    t0p = timed(lambda: ops.linear(x, w, b))
    _lib.set_tuning("DIFFSAL_NO_PERSIST", 0)
    y1 = ops.linear(x, w, b, residual=r)
    t1 = timed(lambda: ops.linear(x, w, b, residual=r))
    t1p = timed(lambda: ops.linear(x, w, b))
    print(f"M={M:6d} K={K:4d} N={N:4d}: one-tile {t0:6.1f} us (plain {t0p:6.1f}) | persistent {t1:6.1f} us (plain {t1p:6.1f})  "
          f"max|diff| {(y0.float() - y1.float()).abs().max().item():.2e}", flush=True)
