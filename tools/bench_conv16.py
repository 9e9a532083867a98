#!/usr/bin/env python3
"""16-bit 3x3 convolutions of a 64-clip pass: the LDS-DMA halo kernel (csrc/conv16_dma.hip) against what the planner took before it
(DIFFSAL_NO_STREAM16=1: conv16_halo / igemm16).  usage: tools/bench_conv16.py [--batch 64] [--dtype fp16]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diff_sal_amd import _lib, ops  # noqa: E402

DEV = "cuda"
# name, images per clip, H, W, Cin, Cout, pad, dil, residual
SHAPES = [
    ("res0.conv1", 1, 56, 96, 96, 192, 1, 1, False),
    ("res0.conv2", 1, 56, 96, 192, 192, 1, 1, True),
    ("res1.conv1", 1, 28, 48, 192, 384, 1, 1, False),
    ("res1.conv2", 1, 28, 48, 384, 384, 1, 1, True),
    ("res2.conv1", 1, 14, 24, 384, 768, 1, 1, False),
    ("res2.conv2", 1, 14, 24, 768, 768, 1, 1, True),
    ("s1.pe1 ext", 9, 7, 12, 768, 384, 2, 1, False),
    ("s1.pe2    ", 9, 14, 24, 384, 384, 2, 2, True),
    ("s2.pe1 ext", 9, 14, 24, 384, 192, 2, 1, False),
    ("s2.pe2    ", 9, 28, 48, 192, 192, 2, 2, True),
    ("s3.pe1 ext", 9, 28, 48, 192, 96, 2, 1, False),
    ("s3.pe2    ", 9, 56, 96, 96, 96, 2, 2, False),
]


def main():
    args = sys.argv[1:]
    B, dt = 64, torch.float16
    i = 0
    while i < len(args):
        if args[i] == "--batch":
            B = int(args[i + 1]); i += 2
        elif args[i] == "--dtype":
            dt = {"fp16": torch.float16, "bf16": torch.bfloat16}[args[i + 1]]; i += 2
        else:
            i += 1
    g = torch.Generator(device=DEV).manual_seed(7)
    lib = _lib.load()
    for name, ipc, H, W, Cin, Cout, pad, dil, has_res in SHAPES:
        N = B * ipc
        Ho, Wo = H + 2 * pad - 2 * dil, W + 2 * pad - 2 * dil
        x = torch.randn(N, H, W, Cin, device=DEV, generator=g).to(dt)
        w = (torch.randn(Cout, Cin, 3, 3, device=DEV, generator=g) / (3 * Cin ** 0.5))
        wp = ops.cast(ops.pack_conv_weight(w), dt)
        sc, sh = torch.rand(Cout, device=DEV, generator=g) + 0.5, torch.randn(Cout, device=DEV, generator=g)
        res = torch.randn(N, Ho, Wo, Cout, device=DEV, generator=g).to(dt) if has_res else None
        fl = 2.0 * N * Ho * Wo * 9 * Cin * Cout
        cells, outs = [], []
        for old, tile, waves in ((1, None, None), (None, 0, 0), (None, 0, 1), (None, None, None)):
            _lib.set_tuning("DIFFSAL_NO_STREAM16", old)
            _lib.set_tuning("DIFFSAL_CONV16_TILE", tile)
            _lib.set_tuning("DIFFSAL_CONV16_HALF", waves)
            _lib.set_tuning("DIFFSAL_FORCE_HALO", 2 if (old is None and os.environ.get("CD_FORCE")) else None)
            run = lambda: ops.conv_igemm(x, wp, kh=3, kw=3, pad=(pad, pad), dil=(dil, dil), out_hw=(Ho, Wo), scale=sc, shift=sh, residual=res, act=1)
            outs.append(run())
            kern = lib.diffsal_last_gemm_kernel().decode()
            kern = (kern.split("[")[0][:22] + kern.split("[")[1][:24]) if "[" in kern else kern[:46]
            torch.cuda.synchronize()
            ts = []
            for _ in range(3):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(10):
                    run()
                e1.record()
                torch.cuda.synchronize()
                ts.append(e0.elapsed_time(e1) * 100)
            t = sorted(ts)[1]
            cells.append(f"{t:7.1f} us {fl / t / 1e6:6.1f} TF/s {kern:46s}")
        _lib.set_tuning("DIFFSAL_NO_STREAM16", None)
        _lib.set_tuning("DIFFSAL_FORCE_HALO", None)
        _lib.set_tuning("DIFFSAL_CONV16_TILE", None)
        _lib.set_tuning("DIFFSAL_CONV16_HALF", None)
        print(f"{name} M={N * Ho * Wo:8d} K={9 * Cin:5d} N={Cout:4d} | " + " | ".join(cells) + f" | same bits: {all(torch.equal(outs[0], o) for o in outs[1:])}", flush=True)


if __name__ == "__main__":
    main()
