#!/usr/bin/env python3
"""The three pooling convolutions of an MViTv2-S block (training forms: convolution only) at B clips of 16x224x384, every stage: the
token-per-lane-group kernels against the run forms (csrc/mvit_pool.hip).  usage: tools/bench_pool.py [B]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diff_sal_amd import _lib, ops  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
# heads, (T, H, W), stride_q, stride_kv: blocks of stages 1-4 and the first block of stages 2-4 (q pooled by 2)
CASES = [(1, (8, 56, 96), (1, 1, 1), (1, 8, 8)), (1, (8, 56, 96), (1, 2, 2), (1, 8, 8)), (2, (8, 28, 48), (1, 1, 1), (1, 4, 4)),
         (2, (8, 28, 48), (1, 2, 2), (1, 4, 4)), (4, (8, 14, 24), (1, 1, 1), (1, 2, 2)), (4, (8, 14, 24), (1, 2, 2), (1, 2, 2)),
         (8, (8, 7, 12), (1, 1, 1), (1, 1, 1))]


def timed(fn, n=10):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for heads, size, sq, skv in CASES:
    N = 1 + size[0] * size[1] * size[2]
    qkv = torch.randn(B, N, 3, heads, 96, device="cuda")
    ws = [torch.randn(27, 96, device="cuda") * 0.2 for _ in range(3)]
    outs = ops.qkv_pool(qkv, ws, size, sq, skv)[:3]
    gs = [torch.randn_like(o) for o in outs]
    line = f"heads {heads} {size} q{sq[1]} kv{skv[1]}:"
    for form in ("tokens", "runs"):
        _lib.set_tuning("DIFFSAL_NO_POOL_RUNS", 1 if form == "tokens" else None)
        f = timed(lambda: ops.qkv_pool(qkv, ws, size, sq, skv))
        d = timed(lambda: ops.qkv_pool_bwd_data(gs, ws, qkv.shape, size, sq, skv))
        w = timed(lambda: ops.qkv_pool_bwd_weight(qkv, gs, size, sq, skv))
        line += f"   {form}: fwd {f:6.1f}  data {d:6.1f}  filter {w:6.1f} us"
    print(line, flush=True)
