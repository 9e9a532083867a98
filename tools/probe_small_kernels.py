#!/usr/bin/env python3
"""Timing probes for the small HBM/latency-bound kernels of a SalUNet step (GPU only): how the time scales with the
number of images / queries tells launch-bound from throughput-bound."""
import sys, os
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diff_sal_amd import ops


def timed(fn, reps=20):
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


dev = "cuda"
print("attention core (N images, Lq, C): us")
for N, Lq, C in [(36, 84, 768), (9, 84, 768), (36, 21, 768), (36, 336, 384), (36, 1344, 192), (36, 5376, 96), (9, 5376, 96), (144, 84, 768)]:
    q = torch.randn(N, Lq, C, device=dev)
    k, v = torch.randn(N, 18, C, device=dev), torch.randn(N, 18, C, device=dev)
    print(f"  N={N:4d} Lq={Lq:5d} C={C:4d}: {timed(lambda: ops.attention(q, k, v, 2, C ** -0.5)):8.1f}")
print("groupnorm_swish (B, H, W, C): us")
for B, H, W, C in [(4, 56, 96, 96), (4, 56, 96, 192), (4, 28, 48, 384), (4, 14, 24, 768), (1, 56, 96, 96), (16, 56, 96, 96)]:
    x = torch.randn(B, H, W, C, device=dev)
    g, b = torch.ones(C, device=dev), torch.zeros(C, device=dev)
    print(f"  B={B:3d} {H}x{W}x{C}: {timed(lambda: ops.groupnorm_swish(x, g, b, 32, 1e-6)):8.1f}")
print("layernorm (M, C): us")
for M, C in [(3024, 768), (12096, 384), (48384, 192), (193536, 96)]:
    x = torch.randn(M, C, device=dev)
    g, b = torch.ones(C, device=dev), torch.zeros(C, device=dev)
    print(f"  M={M:7d} C={C:4d}: {timed(lambda: ops.layernorm(x, g, b)):8.1f}")
print("empty-ish launch (axpbypcz on 4 floats): us", timed(lambda: ops.axpbypcz(torch.zeros(4, device=dev), 1.0)))
x = torch.zeros(4, device=dev)
print("launch only (preallocated):", timed(lambda: ops.axpbypcz(x, 1.0, out=x)))
