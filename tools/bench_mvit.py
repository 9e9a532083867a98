#!/usr/bin/env python3
"""MViTv2-S forward (eval) on 4 synthetic clips: wall time per batch; run under `rocprofv3 --kernel-trace --stats` for the per-kernel split."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diff_sal_amd.mvit import MViT  # noqa: E402

torch.manual_seed(7)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
DT = {"fp32": torch.float32, "bf16": torch.bfloat16, "fp16": torch.float16}[sys.argv[2] if len(sys.argv) > 2 else "fp32"]
enc = MViT(arch="small", out_scales=[0, 1, 2, 3], compute_dtype=DT).cuda().eval().requires_grad_(False)
clip = torch.randn((B, 3, 16, 224, 384), device="cuda")
with torch.no_grad():
    for _ in range(3):
        enc(clip)
    torch.cuda.synchronize()
    ts = []
    for _ in range(10):
        t0 = time.perf_counter()
        enc(clip)
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
ts.sort()
print(f"MViT-S forward ({DT}), {B} clips: median {ts[len(ts) // 2] * 1e3:.2f} ms  ({255.6 * B / ts[len(ts) // 2] / 1e3:.1f} TF/s on 255.6 GFLOP per clip)")
