#!/usr/bin/env python3
"""GroupNorm+swish (K3) on the six shapes of one denoiser step at B = 4, timed as HIP-graph replays (kernels back to back, as in
the sampling graph), over the tuning knobs DIFFSAL_GN_CHUNKS x DIFFSAL_GN_APPLY_WGS.  usage: python tools/bench_gn.py [fp32|bf16]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diff_sal_amd import _lib, ops  # noqa: E402

DT = {"fp32": torch.float32, "bf16": torch.bfloat16}[sys.argv[1] if len(sys.argv) > 1 else "fp32"]
shapes = [(4, 56, 96, 96), (4, 56, 96, 192), (4, 28, 48, 192), (4, 28, 48, 384), (4, 14, 24, 384), (4, 14, 24, 768)]
REP = 20
for chunks in (32, 16, 64, 128):
    for cap in (512, 128, 256, 1024):
        _lib.set_tuning("DIFFSAL_GN_CHUNKS", chunks)
        _lib.set_tuning("DIFFSAL_GN_APPLY_WGS", cap)
        per = []
        for B, H, W, C in shapes:
            x = torch.randn((B, H, W, C), device="cuda").to(DT)
            g, b = torch.ones(C, device="cuda"), torch.zeros(C, device="cuda")
            ops.groupnorm_swish(x, g, b)
            torch.cuda.synchronize()
            gr = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gr):
                for _ in range(REP):
                    y = ops.groupnorm_swish(x, g, b)
            gr.replay()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                gr.replay()
            e1.record()
            torch.cuda.synchronize()
            per.append(e0.elapsed_time(e1) * 1e3 / (5 * REP))
        print(f"{DT} chunks={chunks:3d} apply_wgs={cap:4d}: " + " ".join(f"{u:5.1f}" for u in per) + f"  sum {sum(per):6.1f} us")
