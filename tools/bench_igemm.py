#!/usr/bin/env python3
"""Micro-benchmark of the implicit-GEMM kernel on the SalUNet shapes (B=4).  GPU only.
usage: tools/bench_igemm.py [filter]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diff_sal_amd import ops  # noqa: E402

# name, N, H, W, Cin, Cout, k, stride, pad, dil
SHAPES = [
    ("s3.pe1  192->96 d2", 36, 56, 96, 192, 96, 3, 1, 2, 2),
    ("s3.pe2   96->96 d2", 36, 56, 96, 96, 96, 3, 1, 2, 2),
    ("s2.pe1 384->192 d2", 36, 28, 48, 384, 192, 3, 1, 2, 2),
    ("s1.pe1 768->384 d2", 36, 14, 24, 768, 384, 3, 1, 2, 2),
    ("mt_proj 768->96   ", 4, 112, 192, 768, 96, 3, 1, 1, 1),
    ("s3.fc1 96->192 lin", 1, 1, 193536, 96, 192, 1, 1, 0, 1),
    ("s3.proj 96->96 lin", 1, 1, 193536, 96, 96, 1, 1, 0, 1),
    ("s2.fc1 192->384   ", 1, 1, 48384, 192, 384, 1, 1, 0, 1),
    ("s0.q 768->768 lin ", 1, 1, 3024, 768, 768, 1, 1, 0, 1),
    ("enc conv2 768->768", 4, 14, 24, 768, 768, 3, 1, 1, 1),
    ("enc down 768 s2   ", 4, 14, 24, 768, 768, 3, 2, 0, 1),
    ("s3.redu 5x96->768 ", 1, 9, 21504, 96, 768, 0, 5, 0, 1),
]


def main():
    flt = sys.argv[1] if len(sys.argv) > 1 else ""
    dev = "cuda"
    ops.set_gemm_precision(os.environ.get("DIFFSAL_PRECISION", "fp32"))
    print("precision:", ops.get_gemm_precision())
    for name, N, H, W, Cin, Cout, k, st, pad, dil in SHAPES:
        if flt and flt not in name:
            continue
        x = torch.relu(torch.randn(N, H, W, Cin, device=dev))   # post-ReLU-like operands (half zeros), as in the network
        if k == 0:  # ReduceTemp view: kh=5, kw=1, stride (5,1)
            w = torch.randn(Cout, 5 * Cin, device=dev) * 0.05
            kw = dict(kh=5, kw=1, stride=(5, 1))
            flops = 2.0 * N * 1 * W * Cout * 5 * Cin
        else:
            w = torch.randn(Cout, k * k * Cin, device=dev) * 0.05
            Ho = (H + 2 * pad - dil * (k - 1) - 1) // st + 1 if not (st == 2 and pad == 0) else (H - 2) // 2 + 1
            Wo = (W + 2 * pad - dil * (k - 1) - 1) // st + 1 if not (st == 2 and pad == 0) else (W - 2) // 2 + 1
            kw = dict(kh=k, kw=k, stride=(st, st), pad=(pad, pad), dil=(dil, dil), out_hw=(Ho, Wo))
            flops = 2.0 * N * Ho * Wo * Cout * k * k * Cin
        for _ in range(3):
            ops.conv_igemm(x, w, **kw)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 10
        e0.record()
        for _ in range(reps):
            ops.conv_igemm(x, w, **kw)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / reps
        print(f"{name}  {us:9.1f} us  {flops / us / 1e6:7.1f} TF/s  ({flops / 1e9:6.1f} GF)", flush=True)


if __name__ == "__main__":
    main()
