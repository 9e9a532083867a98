#!/usr/bin/env python3
"""Timing of the bilinear kernels on the SalUNet shapes (B=4): the 4-scale sum in front of mt_proj (K13-up) and the x2
up-sampling in front of the UpEmbed convolutions (K12-up).  GPU only.  usage: tools/probe_resize.py [fp32|bf16|fp16]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diff_sal_amd import ops  # noqa: E402
from tools.tune_igemm16 import timed  # noqa: E402


def main():
    dt = {"fp32": torch.float32, "bf16": torch.bfloat16, "fp16": torch.float16}[sys.argv[1] if len(sys.argv) > 1 else "fp32"]
    es = torch.empty(0, dtype=dt).element_size()
    C = 768
    xs = [torch.randn(4, 112 >> i, 192 >> i, C, device="cuda").to(dt) for i in (1, 2, 3, 4)]
    us = timed(lambda: ops.resize_sum(xs, 112, 192))
    nb = (sum(x.numel() for x in xs) + 4 * 112 * 192 * C) * es
    print(f"resize_sum 4 scales -> 4x112x192x{C}: {us:7.1f} us  {nb / us / 1e3:6.0f} GB/s")
    for n, h, w, c in ((36, 28, 48, 192), (36, 14, 24, 384), (36, 7, 12, 768)):
        x = torch.randn(n, h, w, c, device="cuda").to(dt)
        us = timed(lambda: ops.resize_bilinear(x, 2 * h, 2 * w))
        nb = x.numel() * 5 * es
        print(f"resize x2 {n}x{h}x{w}x{c}: {us:7.1f} us  {nb / us / 1e3:6.0f} GB/s")


if __name__ == "__main__":
    main()
