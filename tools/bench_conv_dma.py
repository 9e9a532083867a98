#!/usr/bin/env python3
"""A/B of the LDS-DMA kernel in convolution mode (csrc/gemm_dma.hip, DIFFSAL_CONV_DMA) against the tiled implicit-GEMM kernel on
the convolutions of one fp32 step (B = 4) that do not run as Winograd / tap products: correctness against torch conv2d in fp64
on a sub-batch, then interleaved timing rounds in ONE process.  GPU only.
usage: tools/bench_conv_dma.py [filter] [--cfgs 1,2] [--rounds 5]"""
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diff_sal_amd import _lib, ops  # noqa: E402

# name, N, H, W, Cin, Cout, kh, kw, stride, pad(top,left), dil, out_hw (None = "same" formula)
SHAPES = [
    ("K13 s0 5x1    ", 4, 9, 84, 768, 768, 5, 1, (5, 1), (0, 0), (1, 1), (1, 84)),
    ("K13 s1 5x1    ", 4, 9, 336, 384, 768, 5, 1, (5, 1), (0, 0), (1, 1), (1, 336)),
    ("K13 s2 5x1    ", 4, 9, 1344, 192, 768, 5, 1, (5, 1), (0, 0), (1, 1), (1, 1344)),
    ("K13 s3 5x1    ", 4, 9, 5376, 96, 768, 5, 1, (5, 1), (0, 0), (1, 1), (1, 5376)),
    ("K5 192 s2     ", 4, 56, 96, 192, 192, 3, 3, (2, 2), (0, 0), (1, 1), (28, 48)),
    ("K5 384 s2     ", 4, 28, 48, 384, 384, 3, 3, (2, 2), (0, 0), (1, 1), (14, 24)),
    ("K5 768 s2     ", 4, 14, 24, 768, 768, 3, 3, (2, 2), (0, 0), (1, 1), (7, 12)),
    ("K12 s3 pe2 d2 ", 36, 56, 96, 96, 96, 3, 3, (1, 1), (2, 2), (2, 2), None),
    ("K12 s2 pe2 d2 ", 36, 28, 48, 192, 192, 3, 3, (1, 1), (2, 2), (2, 2), None),
    ("K4 s0 conv2   ", 4, 56, 96, 192, 192, 3, 3, (1, 1), (1, 1), (1, 1), None),
    ("K4 s2 conv2   ", 4, 14, 24, 768, 768, 3, 3, (1, 1), (1, 1), (1, 1), None),
    ("mt_proj direct", 4, 112, 192, 768, 96, 3, 3, (1, 1), (1, 1), (1, 1), None),
    ("ragged 3x3    ", 3, 13, 19, 96, 100, 3, 3, (1, 1), (1, 1), (1, 1), None),
]


def main():
    args = list(sys.argv[1:])
    cfgs, rounds, flt = [1], 5, ""
    i = 0
    while i < len(args):
        if args[i] == "--cfgs":
            cfgs = [int(c) for c in args[i + 1].split(",")]
            i += 2
        elif args[i] == "--rounds":
            rounds = int(args[i + 1])
            i += 2
        else:
            flt = args[i]
            i += 1
    dev = "cuda"
    g = torch.Generator(device=dev).manual_seed(11)
    print(f"{'shape':15s} {'M':>7s} {'K':>5s} {'N':>4s} | tiled us (TF/s) err  | " + " | ".join(f"dma cfg {c} us (TF/s) err vs-tiled" for c in cfgs))
    for name, N, H, W, Cin, Cout, kh, kw, st, pad, dil, ohw in SHAPES:
        if flt and flt not in name:
            continue
        x = torch.relu(torch.randn(N, H, W, Cin, device=dev, generator=g))
        wt = torch.randn(Cout, Cin, kh, kw, device=dev, generator=g) * (Cin * kh * kw) ** -0.5
        b = torch.randn(Cout, device=dev, generator=g) * 0.1
        wp = ops.pack_conv_weight(wt)
        if ohw is None:
            Ho = (H + 2 * pad[0] - dil[0] * (kh - 1) - 1) // st[0] + 1
            Wo = (W + 2 * pad[1] - dil[1] * (kw - 1) - 1) // st[1] + 1
        else:
            Ho, Wo = ohw
        kwargs = dict(kh=kh, kw=kw, stride=st, pad=pad, dil=dil, out_hw=(Ho, Wo), bias=b, act=ops.ACT_RELU)
        # fp64 reference on the first image (zero fill to the right / bottom as out_hw implies)
        pb = max(0, (Ho - 1) * st[0] + dil[0] * (kh - 1) + 1 - H - pad[0])
        pr = max(0, (Wo - 1) * st[1] + dil[1] * (kw - 1) + 1 - W - pad[1])
        x1 = F.pad(x[:1].permute(0, 3, 1, 2).double(), (pad[1], pr, pad[0], pb))
        ref = torch.relu(F.conv2d(x1, wt.double(), b.double(), stride=st, dilation=dil))[:, :, :Ho, :Wo].permute(0, 2, 3, 1)
        scale = ref.abs().max().item()
        variants = [0] + cfgs
        outs, errs, times = {}, {}, {v: [] for v in variants}
        for v in variants:
            _lib.set_tuning("DIFFSAL_CONV_DMA", v)
            y = ops.conv_igemm(x, wp, **kwargs)
            torch.cuda.synchronize()
            outs[v] = y
            errs[v] = (y[:1].double() - ref).abs().max().item() / scale
        reps = 10
        for _ in range(rounds):
            for v in variants:
                _lib.set_tuning("DIFFSAL_CONV_DMA", v)
                ops.conv_igemm(x, wp, **kwargs)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(reps):
                    ops.conv_igemm(x, wp, **kwargs)
                e1.record()
                torch.cuda.synchronize()
                times[v].append(e0.elapsed_time(e1) * 1e3 / reps)
        _lib.set_tuning("DIFFSAL_CONV_DMA", None)
        M, K = N * Ho * Wo, kh * kw * Cin
        fl = 2.0 * M * K * Cout
        cells = []
        for v in variants:
            t = sorted(times[v])[len(times[v]) // 2]
            extra = "" if v == 0 else f" {(outs[v] - outs[0]).abs().max().item() / scale:.1e}"
            cells.append(f"{t:7.1f} ({fl / t / 1e6:5.1f}) {errs[v]:.1e}{extra}")
        print(f"{name:15s} {M:7d} {K:5d} {Cout:4d} | " + " | ".join(cells), flush=True)


if __name__ == "__main__":
    main()
