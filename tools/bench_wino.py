#!/usr/bin/env python3
"""Winograd F(2x2,3x3) against the direct implicit-GEMM kernel on the fp32 3x3 stride-1 convolutions of the denoiser at B=4
(ResnetBlock convs, UpEmbed second convs): microseconds per call, both paths.  GPU only."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diff_sal_amd import _lib, ops  # noqa: E402
from tools.tune_igemm16 import timed  # noqa: E402

SHAPES = [("res0.conv1", 4, 56, 96, 96, 192, 1), ("res0.conv2", 4, 56, 96, 192, 192, 1), ("res1.conv1", 4, 28, 48, 192, 384, 1),
          ("res1.conv2", 4, 28, 48, 384, 384, 1), ("res2.conv1", 4, 14, 24, 384, 768, 1), ("res2.conv2", 4, 14, 24, 768, 768, 1),
          ("s1.pe2", 36, 14, 24, 384, 384, 2), ("s2.pe2", 36, 28, 48, 192, 192, 2), ("s3.pe2", 36, 56, 96, 96, 96, 2)]


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    for name, N, H, W, Ci, Co, d in SHAPES:
        N = N * B // 4
        x = torch.randn(N, H, W, Ci, device="cuda")
        w = torch.randn(Co, Ci, 3, 3, device="cuda") * 0.05
        b = torch.randn(Co, device="cuda")
        wp, wu = ops.pack_conv_weight(w), ops.pack_wino_weight(w)
        ww = ops.WinoWeights(w)
        kw = dict(kh=3, kw=3, pad=(d, d), dil=(d, d), bias=b)
        _lib.set_tuning("DIFFSAL_FORCE_WINOGRAD", 1)
        y1 = ops.conv_igemm(x, wp, wino=wu, **kw)
        t1 = timed(lambda: ops.conv_igemm(x, wp, wino=wu, **kw))
        t4, e4 = float("nan"), float("nan")
        if ww.f4 is not None:
            y4 = ops.conv_igemm(x, wp, wino=ww, **kw)
            t4 = timed(lambda: ops.conv_igemm(x, wp, wino=ww, **kw))
        _lib.set_tuning("DIFFSAL_FORCE_WINOGRAD", None)
        y0 = ops.conv_igemm(x, wp, **kw)
        t0 = timed(lambda: ops.conv_igemm(x, wp, **kw))
        fl = 2.0 * N * H * W * Co * 9 * Ci
        print(f"{name:11s} N={N:2d} {H:3d}x{W:3d} {Ci:3d}->{Co:3d} d{d}: direct {t0:7.1f} us ({fl / t0 / 1e6:6.1f} TF/s)  winograd {t1:7.1f} us "
              f"({fl / t1 / 1e6:6.1f} TF/s-equivalent)  x{t0 / t1:4.2f}  diff {(y1 - y0).abs().max().item() / y0.abs().max().item():.1e}"
              + (f"  F(4x4) {t4:7.1f} us x{t0 / t4:4.2f} diff {(y4 - y0).abs().max().item() / y0.abs().max().item():.1e}" if ww.f4 is not None else ""), flush=True)


if __name__ == "__main__":
    main()
