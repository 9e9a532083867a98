#!/usr/bin/env python3
"""Tile-shape sweep of the 16-bit implicit-GEMM kernel (csrc/igemm16.hip) on the SalUNet shapes (B=4).  GPU only.
For every shape: the planner's own choice and each forced tile configuration (DIFFSAL_IGEMM16_CFG, no split-K).
usage: tools/tune_igemm16.py [bf16|fp16] [filter]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diff_sal_amd import _lib, ops  # noqa: E402
from tools.bench_igemm import SHAPES  # noqa: E402

SPLITS = [int(v) for v in os.environ.get("TUNE_SPLITS", "0").split(",")]     # log2 of the forced K splits (value / 8 of DIFFSAL_IGEMM16_CFG)
CFG_NAMES = ["128x192", "128x128", "128x96", "64x128", "128x64", "64x64", "256x96", "256x128"]


def timed(fn, reps=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


def main():
    dt = {"bf16": torch.bfloat16, "fp16": torch.float16}[sys.argv[1] if len(sys.argv) > 1 else "bf16"]
    flt = sys.argv[2] if len(sys.argv) > 2 else ""
    dev = "cuda"
    extra = [("s2.pe2 192->192 d2", 36, 28, 48, 192, 192, 3, 1, 2, 2), ("s1.pe2 384->384 d2", 36, 14, 24, 384, 384, 3, 1, 2, 2),
             ("s3.fc2 192->96 lin", 1, 1, 193536, 192, 96, 1, 1, 0, 1), ("res0.conv2 192    ", 4, 56, 96, 192, 192, 3, 1, 1, 1),
             ("res1.conv2 384    ", 4, 28, 48, 384, 384, 3, 1, 1, 1), ("s1.fc1 384->768   ", 1, 1, 12096, 384, 768, 1, 1, 0, 1),
             ("res0.conv1 96->192", 4, 56, 96, 96, 192, 3, 1, 1, 1), ("down0 192 s2      ", 4, 56, 96, 192, 192, 3, 2, 0, 1),
             ("res1.conv1 192->384", 4, 28, 48, 192, 384, 3, 1, 1, 1), ("down1 384 s2      ", 4, 28, 48, 384, 384, 3, 2, 0, 1),
             ("res2.conv1 384->768", 4, 14, 24, 384, 768, 3, 1, 1, 1),
             ("s0.redu 5x768->768", 1, 9, 336, 768, 768, 0, 5, 0, 1), ("s1.redu 5x384->768", 1, 9, 1344, 384, 768, 0, 5, 0, 1),
             ("s2.redu 5x192->768", 1, 9, 5376, 192, 768, 0, 5, 0, 1),
             ("s1.pe1c ext 768->384", 36, 9, 14, 768, 384, 3, 1, 1, 1), ("s2.pe1c ext 384->192", 36, 16, 26, 384, 192, 3, 1, 1, 1),
             ("s3.pe1c ext 192->96", 36, 30, 50, 192, 96, 3, 1, 1, 1)]
    for name, N, H, W, Cin, Cout, k, st, pad, dil in SHAPES + extra:
        if flt and flt not in name:
            continue
        x = torch.relu(torch.randn(N, H, W, Cin, device=dev)).to(dt)
        if k == 0:
            w = (torch.randn(Cout, 5 * Cin, device=dev) * 0.05).to(dt)
            kw = dict(kh=5, kw=1, stride=(5, 1))
            flops = 2.0 * N * 1 * W * Cout * 5 * Cin
            Ho, Wo = 1, W
        else:
            w = (torch.randn(Cout, k * k * Cin, device=dev) * 0.05).to(dt)
            Ho = (H + 2 * pad - dil * (k - 1) - 1) // st + 1 if not (st == 2 and pad == 0) else (H - 2) // 2 + 1
            Wo = (W + 2 * pad - dil * (k - 1) - 1) // st + 1 if not (st == 2 and pad == 0) else (W - 2) // 2 + 1
            kw = dict(kh=k, kw=k, stride=(st, st), pad=(pad, pad), dil=(dil, dil), out_hw=(Ho, Wo))
            flops = 2.0 * N * Ho * Wo * Cout * k * k * Cin
        nbytes = (x.numel() + w.numel() + N * Ho * Wo * Cout) * 2
        _lib.set_tuning("DIFFSAL_IGEMM16_CFG", None)
        us = timed(lambda: ops.conv_igemm(x, w, **kw))
        line = f"{name} M={N * Ho * Wo:7d} planner {us:8.1f} us {flops / us / 1e6:7.1f} TF/s {nbytes / us / 1e3:6.0f} GB/s |"
        for c, cn in enumerate(CFG_NAMES):
            if (c in (0,) and Cout < 192 - 31) or (c in (1, 3, 7) and Cout < 97):
                continue
            for sp in SPLITS:
                _lib.set_tuning("DIFFSAL_IGEMM16_CFG", c + 8 * sp)
                u = timed(lambda: ops.conv_igemm(x, w, **kw))
                line += f" {cn}{'/' + str(1 << sp) if sp else ''} {u:6.1f}"
        _lib.set_tuning("DIFFSAL_IGEMM16_CFG", None)
        print(line, flush=True)


if __name__ == "__main__":
    main()
