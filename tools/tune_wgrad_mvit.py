#!/usr/bin/env python3
"""Weight-gradient tile sweep on the token GEMMs of MViTv2-S at 4 clips (dW[Cout, K] = dY^T X, M tokens): the planner's choice
against each forced tile shape (DIFFSAL_WGRAD_CFG; the M split stays the planner's for that shape).  GPU only."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diff_sal_amd import _lib, ops  # noqa: E402
from tools.tune_igemm16 import timed  # noqa: E402

CFG = ["128x192", "96x256", "64x384", "32x384", "96x128"]
SHAPES = [(10756, 384, 1152), (10756, 384, 384), (10756, 384, 1536), (10756, 1536, 384),
          (43012, 192, 576), (43012, 192, 192), (43012, 192, 768), (43012, 768, 192),
          (172036, 96, 288), (172036, 96, 96), (172036, 96, 384), (172036, 384, 96),
          (2692, 768, 2304), (2692, 768, 768), (2692, 768, 3072), (2692, 3072, 768)]


def main():
    for M, K, N in SHAPES:
        x = torch.randn(1, 1, M, K, device="cuda")
        dy = torch.randn(1, 1, M, N, device="cuda")
        fl = 2.0 * M * K * N
        _lib.set_tuning("DIFFSAL_WGRAD_CFG", None)
        us = timed(lambda: ops.conv_wgrad(x, dy, kh=1, kw=1))
        line = f"M={M:6d} K={K:4d} Cout={N:4d} auto {us:7.1f} us {fl / us / 1e6:6.1f} TF/s |"
        for c, cn in enumerate(CFG):
            _lib.set_tuning("DIFFSAL_WGRAD_CFG", c)
            u = timed(lambda: ops.conv_wgrad(x, dy, kh=1, kw=1))
            line += f" {cn} {u:6.1f}"
        _lib.set_tuning("DIFFSAL_WGRAD_CFG", None)
        print(line, flush=True)


if __name__ == "__main__":
    main()
