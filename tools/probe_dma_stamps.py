#!/usr/bin/env python3
"""Development aid: per-workgroup phase stamps of gemm_dma_kernel (wall_clock64, 100 MHz) for one F(4x4) Winograd convolution (its
batched position products) or one plain product.  Needs a development build:
DIFFSAL_EXTRA_HIPCC_FLAGS=-DDIFFSAL_DEV_STAMPS python -m diff_sal_amd.build --force
usage: probe_dma_stamps.py wino N H W Cin Cout d | probe_dma_stamps.py gemm M K N"""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diff_sal_amd import _lib, ops  # noqa: E402

mode = sys.argv[1] if len(sys.argv) > 1 else "wino"
if mode == "wino":
    N, H, W, Ci, Co, d = map(int, sys.argv[2:8]) if len(sys.argv) > 7 else (36, 56, 96, 96, 96, 2)
    x = torch.randn(N, H, W, Ci, device="cuda")
    w = torch.randn(Co, Ci, 3, 3, device="cuda") * 0.05
    wp, ww = ops.pack_conv_weight(w), ops.WinoWeights(w)
    _lib.set_tuning("DIFFSAL_FORCE_WINOGRAD", 1)
    run = lambda: ops.conv_igemm(x, wp, wino=ww, kh=3, kw=3, pad=(d, d), dil=(d, d))
else:
    M, K, N = map(int, sys.argv[2:5])
    x = torch.randn(M, K, device="cuda")
    w = torch.randn(N, K, device="cuda") * 0.05
    _lib.set_tuning("DIFFSAL_GEMM_DMA", 1)
    run = lambda: ops.linear(x, w)
for _ in range(3):
    run()
buf = torch.zeros(512 * 32, dtype=torch.int64, device="cuda")
lib = _lib.load()
if not hasattr(lib, "diffsal_set_dma_stamps"):
    sys.exit("this libdiffsal_hip.so was built without -DDIFFSAL_DEV_STAMPS")
lib.diffsal_set_dma_stamps.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
lib.diffsal_set_dma_stamps(buf.data_ptr(), buf.numel() * 8)
run()
torch.cuda.synchronize()
lib.diffsal_set_dma_stamps(None, 0)
s = buf.view(-1, 32).cpu()
s = s[s[:, 0] > 0].double()
t0 = s[:, 0].min()
print("workgroups", s.shape[0], " start spread us", ((s[:, 0] - t0) / 100).max().item())
nu = int(((s[:, 1::3] > 0).sum(1)).max().item())
print("units per workgroup (stamped):", nu, " kernel span us", ((s.max() - t0) / 100).item())
print(f"entry -> first unit: {((s[:, 1] - s[:, 0]) / 100).mean().item():.2f} us")
for t in range(min(nu, 10)):
    ok = s[:, 3 + 3 * t] > 0
    if ok.sum() == 0:
        break
    walk = ((s[ok, 2 + 3 * t] - s[ok, 1 + 3 * t]) / 100)
    epi = ((s[ok, 3 + 3 * t] - s[ok, 2 + 3 * t]) / 100)
    print(f"unit {t}: {int(ok.sum())} workgroups; K walk mean {walk.mean().item():.2f} us (min {walk.min().item():.2f}, max {walk.max().item():.2f}); "
          f"epilogue mean {epi.mean().item():.2f} (max {epi.max().item():.2f}); ends at {((s[ok, 3 + 3 * t] - t0) / 100).mean().item():.2f}")
