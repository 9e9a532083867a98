#!/usr/bin/env python3
"""fp32 implicit-GEMM tile sweep on the 3x3 / 5x1 convolutions of the denoising step (B=4): the planner's choice (with its
split-K) against each forced tile shape (DIFFSAL_IGEMM_CFG: no split-K).  GPU only."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diff_sal_amd import _lib, ops  # noqa: E402
from tools.tune_igemm16 import timed  # noqa: E402

SPLITS = [int(v) for v in os.environ.get("TUNE_SPLITS", "0").split(",")]     # log2 of the forced K splits (value / 8 of DIFFSAL_IGEMM_CFG)
CFG = ["128x192", "128x128", "128x96", "64x128", "128x64", "64x64"]
# (images, H, W, Cin, Cout, kh, kw, dil, stride)
SHAPES = [(4, 56, 96, 96, 192, 3, 3, 1, 1), (4, 56, 96, 192, 192, 3, 3, 1, 1), (4, 56, 96, 192, 192, 3, 3, 1, 2),
          (4, 28, 48, 192, 384, 3, 3, 1, 1), (4, 28, 48, 384, 384, 3, 3, 1, 1), (4, 28, 48, 384, 384, 3, 3, 1, 2),
          (4, 14, 24, 384, 768, 3, 3, 1, 1), (4, 14, 24, 768, 768, 3, 3, 1, 1), (4, 14, 24, 768, 768, 3, 3, 1, 2),
          (36, 14, 24, 384, 384, 3, 3, 2, 1), (36, 28, 48, 192, 192, 3, 3, 2, 1), (36, 56, 96, 96, 96, 3, 3, 2, 1)]
if os.environ.get("TUNE_ONLY_STRIDED"):
    SHAPES = [s_ for s_ in SHAPES if s_[8] == 2]


def main():
    for n, H, W, ci, co, kh, kw, dil, st in SHAPES:
        x = torch.randn(n, H, W, ci, device="cuda")
        w = ops.pack_conv_weight(torch.randn(co, ci, kh, kw, device="cuda") * 0.05)
        pad = (dil, dil) if st == 1 else (0, 0)
        oh = (H, W) if st == 1 else ((H - 2) // 2 + 1, (W - 2) // 2 + 1)
        f = lambda: ops.conv_igemm(x, w, kh=kh, kw=kw, stride=(st, st), pad=pad, dil=(dil, dil), out_hw=oh)  # noqa: E731
        _lib.set_tuning("DIFFSAL_IGEMM_CFG", None)
        f()
        us = timed(f)
        M, K = n * oh[0] * oh[1], kh * kw * ci
        line = f"M={M:6d} K={K:5d} N={co:4d} s{st} d{dil} planner {us:7.1f} us {2.0 * M * K * co / us / 1e6:6.1f} TF/s |"
        for c, cn in enumerate(CFG):
            if (c == 0 and co < 161) or (c in (1, 3) and co < 97):
                continue
            for sp in SPLITS:
                _lib.set_tuning("DIFFSAL_IGEMM_CFG", c + 8 * sp)
                line += f" {cn}{'/' + str(1 << sp) if sp else ''} {timed(f):6.1f}"
        _lib.set_tuning("DIFFSAL_IGEMM_CFG", None)
        print(line, flush=True)


if __name__ == "__main__":
    main()
