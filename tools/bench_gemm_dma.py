#!/usr/bin/env python3
"""A/B of the LDS-DMA GEMM (csrc/gemm_dma.hip) against the planner's kernel on the plain products of one fp32 step (B = 4):
correctness against an fp64 product, then interleaved timing rounds in ONE process.  GPU only.
usage: tools/bench_gemm_dma.py [filter] [--cfgs 1,2,3,4] [--rounds 5]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diff_sal_amd import _lib, ops  # noqa: E402

# name, M, K, N, bias, act, residual
SHAPES = [
    ("s0.q      ", 3024, 768, 768, True, 0, False),
    ("s0.proj   ", 3024, 768, 768, True, 0, True),
    ("s0.fc1    ", 3024, 768, 1536, True, 2, False),
    ("s0.fc2    ", 3024, 1536, 768, True, 0, True),
    ("s1.tap    ", 3024, 768, 3456, False, 0, False),
    ("s1.q      ", 12096, 384, 384, True, 0, False),
    ("s1.proj   ", 12096, 384, 384, True, 0, True),
    ("s1.fc1    ", 12096, 384, 768, True, 2, False),
    ("s1.fc2    ", 12096, 768, 384, True, 0, True),
    ("s2.tap    ", 12096, 384, 1728, False, 0, False),
    ("s2.q      ", 48384, 192, 192, True, 0, False),
    ("s2.proj   ", 48384, 192, 192, True, 0, True),
    ("s2.fc1    ", 48384, 192, 384, True, 2, False),
    ("s2.fc2    ", 48384, 384, 192, True, 0, True),
    ("s3.tap    ", 48384, 192, 864, False, 0, False),
    ("mt.tap    ", 28560, 768, 864, False, 0, False),
    ("kv s0     ", 648, 768, 768, True, 0, False),
    ("nin0      ", 21504, 96, 192, True, 0, False),
    ("nin1      ", 5376, 192, 384, True, 0, False),
    ("nin2      ", 1344, 384, 768, True, 0, False),
    ("ragged    ", 3000, 288, 200, True, 1, True),
    # one clip per step (B = 1: 9 frames)
    ("b1 s0.q   ", 756, 768, 768, True, 0, False),
    ("b1 s0.fc1 ", 756, 768, 1536, True, 2, False),
    ("b1 s0.fc2 ", 756, 1536, 768, True, 0, True),
    ("b1 s1.tap ", 756, 768, 3456, False, 0, False),
    ("b1 s1.q   ", 3024, 384, 384, True, 0, False),
    ("b1 s1.fc1 ", 3024, 384, 768, True, 2, False),
    ("b1 s1.fc2 ", 3024, 768, 384, True, 0, True),
    ("b1 s2.tap ", 3024, 384, 1728, False, 0, False),
    ("b1 s2.q   ", 12096, 192, 192, True, 0, False),
    ("b1 s2.fc1 ", 12096, 192, 384, True, 2, False),
    ("b1 s2.fc2 ", 12096, 384, 192, True, 0, True),
    ("b1 s3.tap ", 12096, 192, 864, False, 0, False),
    ("b1 mt.tap ", 7140, 768, 864, False, 0, False),
    # MViTv2-S token GEMMs of a training step (4 clips): stage 3 (11 blocks), stages 2 / 1 / 4
    ("mv3.qkv   ", 10756, 384, 1152, True, 0, False),
    ("mv3.proj  ", 10756, 384, 384, True, 0, True),
    ("mv3.fc1   ", 10756, 384, 1536, True, 0, False),
    ("mv3.fc2   ", 10756, 1536, 384, True, 0, True),
    ("mv2.qkv   ", 43012, 192, 576, True, 0, False),
    ("mv2.fc1   ", 43012, 192, 768, True, 0, False),
    ("mv2.fc2   ", 43012, 768, 192, True, 0, True),
    ("mv1.fc1   ", 172036, 96, 384, True, 0, False),
    ("mv1.fc2   ", 172036, 384, 96, True, 0, True),
    ("mv4.fc1   ", 2692, 768, 3072, True, 0, False),
    ("mv4.fc2   ", 2692, 3072, 768, True, 0, True),
]


def run(x, w, b, act, res):
    return ops.linear(x, w, b, act=act, residual=res)


def main():
    args = [a for a in sys.argv[1:]]
    # 16-bit (DMA_DTYPE): 1 / 2 = gemm_dma.hip 96 x 96 / 192 x 192, 3 / 4 = gemm16_dma.hip 256 x 96 / 192 x 192 (two workgroups per CU);
    # -1 = the library's own choice (key unset)
    cfgs = [1, 2, 3, 4]
    rounds = 5
    flt = ""
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
    dt = {"bf16": torch.bfloat16, "fp16": torch.float16}.get(os.environ.get("DMA_DTYPE", ""), torch.float32)
    key = os.environ.get("DMA_KEY") or ("DIFFSAL_GEMM_DMA" if dt == torch.float32 else "DIFFSAL_GEMM_DMA16")
    g = torch.Generator(device=dev).manual_seed(5)
    print(f"{'shape':10s} {'M':>6s} {'K':>5s} {'N':>5s} | {'DMA off us (TF/s)':>20s} | " + " | ".join(f"dma cfg {c} us (TF/s) err" for c in cfgs))
    mscale = int(os.environ.get("DMA_MSCALE", "1"))      # 16: the shapes of a 64-clip pass (BASELINE configs[4])
    for name, M, K, N, has_b, act, has_r in SHAPES:
        if flt and flt not in name:
            continue
        M *= mscale
        if M * max(K, N) * 8 > 6e9:
            continue
        x = torch.randn(M, K, device=dev, generator=g).to(dt)
        w = (torch.randn(N, K, device=dev, generator=g) * (K ** -0.5)).to(dt)
        b = torch.randn(N, device=dev, generator=g) if has_b else None
        res = torch.randn(M, N, device=dev, generator=g).to(dt) if has_r else None
        ref = x.double() @ w.double().t()
        if b is not None:
            ref = ref + b.double()
        if act == 1:
            ref = torch.relu(ref)
        elif act == 2:
            ref = torch.nn.functional.gelu(ref)
        if res is not None:
            ref = ref + res.double()
        scale = ref.abs().max().item()
        variants = [0] + cfgs
        errs, times = {}, {v: [] for v in variants}
        for v in variants:
            _lib.set_tuning(key, None if v < 0 else v)
            y = run(x, w, b, act, res)
            torch.cuda.synchronize()
            errs[v] = ((y.double() - ref).abs().max().item()) / scale
        reps = 20
        for _ in range(rounds):
            for v in variants:
                _lib.set_tuning(key, None if v < 0 else v)
                run(x, w, b, act, res)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(reps):
                    run(x, w, b, act, res)
                e1.record()
                torch.cuda.synchronize()
                times[v].append(e0.elapsed_time(e1) * 1e3 / reps)
        _lib.set_tuning(key, None)
        fl = 2.0 * M * K * N
        cells = []
        for v in variants:
            t = sorted(times[v])[len(times[v]) // 2]
            cells.append(f"{t:7.1f} ({fl / t / 1e6:5.1f}) {errs[v]:.1e}")
        print(f"{name:10s} {M:7d} {K:5d} {N:5d} | " + " | ".join(cells), flush=True)


if __name__ == "__main__":
    main()
