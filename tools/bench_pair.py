#!/usr/bin/env python3
"""proj_k / proj_v on the pooled tokens (diffsal_linear_pair: two products of one shape in one launch), every stage, fp32 and 16-bit."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diff_sal_amd import ops  # noqa: E402


def timed(fn, n=20):
    for _ in range(5):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for dt in (torch.float32, torch.bfloat16):
    for Cc in (768, 384, 192, 96):
        x0, x1 = (torch.randn(36, 18, Cc, device="cuda").to(dt) for _ in range(2))
        w0, w1 = ((torch.randn(Cc, Cc, device="cuda") * 0.05).to(dt) for _ in range(2))
        b0, b1 = (torch.randn(Cc, device="cuda") for _ in range(2))
        t = timed(lambda: ops.linear_pair(x0, x1, w0, w1, b0, b1))
        t1 = timed(lambda: ops.linear(x0, w0, b0))
        print(f"{dt} M=648 K=N={Cc}: pair {t:6.1f} us   one product {t1:6.1f} us")
