#!/usr/bin/env python3
"""A/B of the LDS-DMA weight-gradient kernel (wgrad_dma_kernel, DIFFSAL_WGRAD_DMA) against wgrad_kernel on the plain-product
weight gradients of a training step (MViT-S and decoder token GEMMs, B = 4): error against fp64, interleaved timing.  GPU only."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diff_sal_amd import _lib, ops  # noqa: E402

# M, K (= Cin), N (= Cout)
SHAPES = [(28560, 768, 864), (172036, 96, 576), (3024, 768, 3456), (43012, 192, 1152), (10756, 384, 2304), (172036, 96, 384),
          (172032, 448, 96), (12096, 384, 1728), (48384, 192, 864), (2692, 3072, 768), (2692, 768, 3072), (43012, 768, 192),
          (172036, 384, 96), (10756, 384, 1536), (10756, 1536, 384), (12096, 384, 384), (48384, 192, 192), (3024, 768, 768),
          (1000, 96, 100), (777, 160, 72)]


def main():
    dev = "cuda"
    g = torch.Generator(device=dev).manual_seed(2)
    print(f"{'M':>7s} {'K':>5s} {'N':>5s} | wgrad_kernel us (TF/s) err | wgrad_dma us (TF/s) err")
    tot = [0.0, 0.0, 0.0]
    for M, K, N in SHAPES:
        x = torch.randn(1, 1, M, K, device=dev, generator=g)
        dy = torch.randn(1, 1, M, N, device=dev, generator=g)
        ref = (dy.reshape(M, N).double().t() @ x.reshape(M, K).double())
        scale = ref.abs().max().item()
        res = []
        for v in (0, 1, 2):
            _lib.set_tuning("DIFFSAL_WGRAD_DMA", v)
            out = ops.conv_wgrad(x, dy)
            dw = out[0] if isinstance(out, (tuple, list)) else out
            torch.cuda.synchronize()
            err = (dw.reshape(N, K).double() - ref).abs().max().item() / scale
            ts = []
            for _ in range(5):
                ops.conv_wgrad(x, dy)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(10):
                    ops.conv_wgrad(x, dy)
                e1.record()
                torch.cuda.synchronize()
                ts.append(e0.elapsed_time(e1) * 1e3 / 10)
            res.append((sorted(ts)[2], err))
        _lib.set_tuning("DIFFSAL_WGRAD_DMA", None)
        fl = 2.0 * M * K * N
        tot[0] += res[0][0]
        tot[1] += res[1][0]
        tot[2] += res[2][0]
        print(f"{M:7d} {K:5d} {N:5d} | " + " | ".join(f"{t:8.1f} ({fl / t / 1e6:5.1f}) {e:.1e}" for t, e in res), flush=True)
    print("sum", tot)


if __name__ == "__main__":
    main()
