#!/usr/bin/env python3
"""Grouped ReduceTemp launch (diffsal_conv_igemm_group) against the four single launches, B = 4 shapes; DIFFSAL_GROUP_GRID sweep."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diff_sal_amd import _lib, ops  # noqa: E402


def timeit(fn, reps=20, rounds=5):
    ts = []
    for _ in range(rounds):
        fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3 / reps)
    return sorted(ts)[len(ts) // 2]


def main():
    dev = "cuda"
    g = torch.Generator(device=dev).manual_seed(3)
    B, T = 4, 9
    probs = []
    for HW, C in ((84, 768), (336, 384), (1344, 192), (5376, 96)):
        x = torch.randn(B, T, HW, C, device=dev, generator=g)
        w = ops.pack_conv_weight(torch.randn(768, C, 5, 1, device=dev, generator=g) * (5 * C) ** -0.5)
        probs.append(dict(x=x, w=w, kh=5, kw=1, stride=(5, 1), out_hw=(1, HW), act=ops.ACT_RELU,
                          out=torch.empty(B, 1, HW, 768, device=dev)))
    fl = sum(2.0 * B * p["x"].shape[2] * 768 * 5 * p["x"].shape[3] for p in probs)

    def singles():
        for p in probs:
            ops.conv_igemm(p["x"], p["w"], kh=5, kw=1, stride=(5, 1), out_hw=p["out_hw"], act=ops.ACT_RELU, out=p["out"])

    t = timeit(singles)
    print(f"four single launches      {t:7.1f} us  {fl / t / 1e6:6.1f} TF/s")
    for grid in (None, 256, 512, 768, 1024):
        _lib.set_tuning("DIFFSAL_GROUP_GRID", grid)
        t = timeit(lambda: ops.conv_igemm_group(probs))
        print(f"grouped, grid {str(grid):>5s}       {t:7.1f} us  {fl / t / 1e6:6.1f} TF/s")
    _lib.set_tuning("DIFFSAL_GROUP_GRID", None)


if __name__ == "__main__":
    main()
