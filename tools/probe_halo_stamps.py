#!/usr/bin/env python3
"""Development aid: per-workgroup phase stamps of the halo kernel (wall_clock64, 100 MHz) for one shape.  Needs a development
build of the library: DIFFSAL_EXTRA_HIPCC_FLAGS=-DDIFFSAL_DEV_STAMPS python -m diff_sal_amd.build --force (the shipped build has
neither the stamp stores nor the diffsal_set_halo_stamps entry)."""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diff_sal_amd import _lib, ops  # noqa: E402

N, H, W, Cin, Cout, dil = 36, 56, 96, 192, 96, 1
if len(sys.argv) > 1:
    N, H, W, Cin, Cout, dil = map(int, sys.argv[1:7])
x = torch.relu(torch.randn(N, H, W, Cin, device="cuda")).bfloat16()
w = (torch.randn(Cout, 9 * Cin, device="cuda") * 0.05).bfloat16()
kw = dict(kh=3, kw=3, stride=(1, 1), pad=(dil, dil), dil=(dil, dil), out_hw=(H, W))
_lib.set_tuning("DIFFSAL_FORCE_HALO", 1)
for _ in range(3):
    ops.conv_igemm(x, w, **kw)
buf = torch.zeros(8192 * 8, dtype=torch.int64, device="cuda")
lib = _lib.load()
if not hasattr(lib, "diffsal_set_halo_stamps"):
    sys.exit("this libdiffsal_hip.so was built without -DDIFFSAL_DEV_STAMPS")
lib.diffsal_set_halo_stamps.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
lib.diffsal_set_halo_stamps(buf.data_ptr(), buf.numel() * 8)
ops.conv_igemm(x, w, **kw)
torch.cuda.synchronize()
lib.diffsal_set_halo_stamps(None, 0)
s = buf.view(-1, 8).cpu()
s = s[s[:, 0] > 0].double()
t0 = s[:, 0].min()
print("workgroups", s.shape[0], "span us", (s[:, 5].max() - t0).item() / 100)
ph = ["index math", "first fetch+park", "main loop", "epilogue pass 0", "epilogue pass 1"]
for k, name in enumerate(ph):
    d = (s[:, k + 1] - s[:, k]) / 100
    print(f"{name:18s} mean {d.mean().item():7.2f} us  min {d.min().item():7.2f}  max {d.max().item():7.2f}")
st = (s[:, 0] - t0) / 100
print("start times: first round (<1us):", (st < 1).sum().item(), " later:", (st >= 1).sum().item(), " median late start", st[st >= 1].median().item() if (st >= 1).any() else 0)
tot = (s[:, 5] - s[:, 0]) / 100
print("per-WG total mean", tot.mean().item())
