#!/usr/bin/env python3
"""F(4x4, 3x3) convolutions of the step with the batched position products on 96 x 96 tiles (two workgroups per CU) against 96 x 128
tiles (one per CU), alternating on one box: DIFFSAL_BATCH_TILE = 0 / 1, and the planner's own choice (unset).  us per convolution."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diff_sal_amd import _lib, ops  # noqa: E402
from tools.tune_igemm16 import timed  # noqa: E402

SHAPES = [("res1.conv2", 4, 28, 48, 384, 384, 1), ("res2.conv1", 4, 14, 24, 384, 768, 1), ("res2.conv2", 4, 14, 24, 768, 768, 1),
          ("s1.pe2", 36, 14, 24, 384, 384, 2), ("s1.pe1c", 36, 14, 24, 384, 384, 1), ("res1.conv1", 4, 28, 48, 192, 384, 1)]

for name, N, H, W, Ci, Co, d in SHAPES:
    x = torch.randn(N, H, W, Ci, device="cuda")
    w = torch.randn(Co, Ci, 3, 3, device="cuda") * 0.05
    wp, ww = ops.pack_conv_weight(w), ops.WinoWeights(w)
    kw = dict(kh=3, kw=3, pad=(d, d), dil=(d, d))
    _lib.set_tuning("DIFFSAL_FORCE_WINOGRAD", 1)
    res = {}
    for rep in range(2):
        for mode in (0, 1, None):
            _lib.set_tuning("DIFFSAL_BATCH_TILE", mode)
            y = ops.conv_igemm(x, wp, wino=ww, **kw)
            res.setdefault(mode, []).append(timed(lambda: ops.conv_igemm(x, wp, wino=ww, **kw)))
            if mode == 0:
                y0 = y
            elif mode == 1:
                assert (y - y0).abs().max().item() <= 1e-5 * y0.abs().max().item()
    _lib.set_tuning("DIFFSAL_BATCH_TILE", None)
    print(f"{name:11s} N={N:2d} {H}x{W} {Ci}->{Co} d{d}: 96x96 {min(res[0]):6.1f}  96x128 {min(res[1]):6.1f}  planner {min(res[None]):6.1f} us", flush=True)
