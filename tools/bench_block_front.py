#!/usr/bin/env python3
"""Time diffsal_block_front (csrc/tblock.hip) at the finest decoder stage's shape, beside the unfused launches it replaces.  GPU only."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diff_sal_amd import ops  # noqa: E402


def timed(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def main():
    Lk, heads = 18, 2
    scale = int(os.environ.get("FRONT_SCALE", "1"))       # 16: the 64-clip pass
    for dt, (N, H, W, C) in ((torch.float32, (36, 56, 96, 96)), (torch.bfloat16, (36, 56, 96, 96)), (torch.float16, (36, 56, 96, 96)),
                             (torch.bfloat16, (36, 28, 48, 192)), (torch.float16, (36, 28, 48, 192))):
        N *= scale
        g = torch.Generator(device="cuda").manual_seed(0)
        r = lambda *s, sc=1.0: torch.randn(*s, device="cuda", generator=g) * sc
        x = r(N, H, W, C).to(dt)
        k, v = r(N, Lk, C).to(dt), r(N, Lk, C).to(dt)
        g1, b1, gq, bq = r(C, sc=0.1) + 1, r(C, sc=0.1), r(C, sc=0.1) + 1, r(C, sc=0.1)
        w9 = r(9, C, sc=0.4)
        wq, wp = r(C, C, sc=C ** -0.5).to(dt), r(C, C, sc=C ** -0.5).to(dt)
        biq, bip = r(C, sc=0.1), r(C, sc=0.1)
        f32 = dt == torch.float32
        fused = lambda: ops.block_front(x, k, v, (g1, b1, 1e-5), w9, (gq, bq, 1e-5), (wq, biq), (wp, bip) if f32 else None, heads, C ** -0.5)

        def unfused():
            xn = ops.layernorm(x, g1, b1, 1e-5)
            q = ops.dwconv3_ln(xn, w9, gq, bq, 1e-5)
            q = ops.linear(q, wq, biq)
            o = ops.attention(q, k, v, heads, C ** -0.5)
            if f32:
                return ops.linear(o, wp, bip, residual=x.view(N, H * W, C))
            return o

        M = N * H * W
        tf, tu = timed(fused), timed(unfused)
        nb = 2 * M * C * x.element_size()
        print(f"{str(dt):15s} C={C} M={M}: fused {tf:7.1f} us ({nb / tf / 1e3:6.0f} GB/s once-through)   unfused {tu:7.1f} us")
        if C != 96:
            continue
        # the block's second half on the same tokens
        w1, b1_, w2, b2_ = r(2 * C, C, sc=0.1).to(dt), r(2 * C, sc=0.1), r(C, 2 * C, sc=0.08).to(dt), r(C, sc=0.1)
        xt = x.view(N, H * W, C)
        geom = (H * W, 9, 5)
        if f32:
            tb = timed(lambda: ops.mlp_block(xt, (g1, b1, 1e-5), (w1, b1_), (w2, b2_), (gq, bq, 1e-5), geom))
            print(f"{'':15s} mlp_block (LN + fc1 + GELU + fc2 + res + norm_mts): {tb:7.1f} us  {8 * M * C * C / tb / 1e6:6.1f} TF/s")
        else:
            tb = timed(lambda: ops.block16(xt, xt, (wp, bip), (g1, b1, 1e-5), (w1, b1_), (w2, b2_), (gq, bq, 1e-5), geom))
            print(f"{'':15s} block16 (proj + res + LN + MLP + res + norm_mts): {tb:7.1f} us  {10 * M * C * C / tb / 1e6:6.1f} TF/s")


if __name__ == "__main__":
    main()
