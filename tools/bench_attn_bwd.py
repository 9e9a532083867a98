#!/usr/bin/env python3
"""Backward of MViTv2-S's pooling attention at the four stages' shapes (B clips of 16x224x384): the recomputing dq kernel against the dS
form (include/diffsal.h diffsal_attention_general_bwd, ds_ws), HIP-event time per call.   usage: tools/bench_attn_bwd.py [B]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diff_sal_amd import ops  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
torch.manual_seed(3)
SHAPES = [(1, 43009, 673, 48), (2, 10753, 673, 48), (4, 2689, 673, 32), (8, 673, 673, 32)]     # heads, Lq, Lk, E


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


for H, Lq, Lk, E in SHAPES:
    D = 96
    q, k, v = (torch.randn(B, H, L, D, device="cuda") for L in (Lq, Lk, Lk))
    qe = torch.randn(B, H, Lq, E, device="cuda") * 0.3
    ke = ops.relpos_onehot((8, 7, 12), E, "cuda")          # carries its slot form (DIFFSAL_NO_ATTN_SLOTS=1: contraction form)
    G = torch.randn(B, Lq, H * D, device="cuda")
    kw = dict(scale=D ** -0.5, q_extra=qe, k_extra=ke, residual=q, skip_first=True)
    out, lse = ops.attention_general(q, k, v, want_lse=True, **kw)
    t_fwd = timed(lambda: ops.attention_general(q, k, v, want_lse=True, **kw))
    t_a = timed(lambda: ops.attention_general_bwd(q, k, v, out, lse, G, ds_form=False, **kw))
    t_b = timed(lambda: ops.attention_general_bwd(q, k, v, out, lse, G, ds_form=True, **kw))
    gf = 2.0 * B * H * Lq * Lk * (D + E + D) / 1e9
    print(f"H={H} Lq={Lq} Lk={Lk} E={E}: fwd {t_fwd:8.1f} us ({gf / t_fwd * 1e3:5.1f} TF/s)   bwd recompute {t_a:8.1f} us   dS form {t_b:8.1f} us"
          f"   ({2.5 * gf / t_b * 1e3:5.1f} TF/s on 5 products)   dS {4e-6 * B * H * Lq * ((Lk + 31) // 32 * 32):.0f} MB")
