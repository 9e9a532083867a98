#!/usr/bin/env python3
"""fp32 implicit-GEMM tile sweep on the thin-K token GEMMs of the denoiser (K10 / K13 / K4-1x1 shapes, B=4): the planner's
choice and each forced tile shape (DIFFSAL_IGEMM_CFG, no split-K).  GPU only."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diff_sal_amd import _lib, ops  # noqa: E402
from tools.tune_igemm16 import timed  # noqa: E402

CFG = ["128x192", "128x128", "128x96", "64x128", "128x64", "64x64"]
TAP_SHAPES = [(48384, 192, 864), (12096, 384, 1728), (3024, 768, 3456), (28560, 768, 864)]   # tap GEMMs (K12 conv1, K14)
SHAPES = [(193536, 96, 96), (193536, 96, 192), (193536, 192, 96), (48384, 192, 192), (48384, 192, 384), (48384, 384, 192),
          (12096, 384, 384), (12096, 384, 768), (12096, 768, 384), (3024, 768, 768), (3024, 768, 1536), (3024, 1536, 768),
          (648, 768, 768), (648, 384, 384), (48384, 3456, 192), (12096, 6912, 384), (5376, 1728, 384), (1344, 3456, 768), (21504, 96, 192), (5376, 192, 384), (1344, 384, 768)]
# MViTv2-S token GEMMs at 4 clips of 16x224x384 (forward and data-gradient orientations): --mvit
MVIT_SHAPES = [(10756, 384, 1152), (10756, 384, 384), (10756, 384, 1536), (10756, 1536, 384), (10756, 1152, 384),
               (43012, 192, 576), (43012, 192, 192), (43012, 192, 768), (43012, 768, 192), (43012, 576, 192), (43012, 192, 1152),
               (172036, 96, 288), (172036, 96, 96), (172036, 96, 384), (172036, 384, 96), (172036, 96, 576), (172036, 288, 96),
               (2692, 768, 2304), (2692, 768, 768), (2692, 768, 3072), (2692, 3072, 768), (2692, 2304, 768)]


def main():
    lowp = "--bf16" in sys.argv          # 16-bit storage: the igemm16 kernels and their own tile table
    var = "DIFFSAL_IGEMM16_CFG" if lowp else "DIFFSAL_IGEMM_CFG"
    names = ["128x192", "128x128", "128x96", "64x128", "128x64", "64x64", "256x96", "256x128"] if lowp else CFG
    for M, K, N in (TAP_SHAPES if "--tap" in sys.argv else MVIT_SHAPES if "--mvit" in sys.argv else SHAPES):
        x = torch.randn(M, K, device="cuda")
        w = torch.randn(N, K, device="cuda") * 0.05
        if lowp:
            x, w = x.bfloat16(), w.bfloat16()
        b = torch.randn(N, device="cuda")
        _lib.set_tuning(var, None)
        _lib.set_tuning("DIFFSAL_NO_PERSIST", 1)
        y0 = ops.linear(x, w, b)
        us0 = timed(lambda: ops.linear(x, w, b))
        _lib.set_tuning("DIFFSAL_NO_PERSIST", 0)
        y1 = ops.linear(x, w, b)
        us = timed(lambda: ops.linear(x, w, b))
        fl = 2.0 * M * K * N
        line = (f"M={M:6d} K={K:4d} N={N:4d} one-tile {us0:7.1f} us | persistent {us:7.1f} us {fl / us / 1e6:6.1f} TF/s "
                f"diff {(y0.float() - y1.float()).abs().max().item():.1e} |")
        for c, cn in enumerate(names):
            if (c == 0 and N < 161) or (c in (1, 3, 7) and N < 97):
                continue
            _lib.set_tuning(var, c)
            u = timed(lambda: ops.linear(x, w, b))
            line += f" {cn} {u:6.1f}"
        _lib.set_tuning(var, None)
        print(line, flush=True)


if __name__ == "__main__":
    main()
