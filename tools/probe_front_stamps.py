#!/usr/bin/env python3
"""Development aid: per-workgroup phase stamps of block_front (wall_clock64, 100 MHz).  Needs
DIFFSAL_EXTRA_HIPCC_FLAGS=-DDIFFSAL_DEV_STAMPS python -m diff_sal_amd.build --force   (not in the shipped build)."""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diff_sal_amd import _lib, ops  # noqa: E402

lib = _lib.load()
if not hasattr(lib, "diffsal_set_front_stamps"):
    sys.exit("this libdiffsal_hip.so was built without -DDIFFSAL_DEV_STAMPS")
lib.diffsal_set_front_stamps.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
Lk, heads = 18, 2
NB = int(os.environ.get("FRONT_N", "36"))      # images: 36 = 4 clips, 576 = 64 clips
for dt, (N, H, W, C) in ((torch.float32, (36, 56, 96, 96)), (torch.bfloat16, (NB, 56, 96, 96)), (torch.bfloat16, (NB, 28, 48, 192))):
    r = lambda *s, sc=1.0: torch.randn(*s, device="cuda") * sc
    x, k, v = r(N, H, W, C).to(dt), r(N, Lk, C).to(dt), r(N, Lk, C).to(dt)
    g1, b1, gq, bq, w9 = r(C, sc=0.1) + 1, r(C, sc=0.1), r(C, sc=0.1) + 1, r(C, sc=0.1), r(9, C, sc=0.4)
    wq, wp, biq, bip = r(C, C, sc=0.1).to(dt), r(C, C, sc=0.1).to(dt), r(C, sc=0.1), r(C, sc=0.1)
    f32 = dt == torch.float32
    run = lambda: ops.block_front(x, k, v, (g1, b1, 1e-5), w9, (gq, bq, 1e-5), (wq, biq), (wp, bip) if f32 else None, heads, C ** -0.5)
    for _ in range(5):
        run()
    buf = torch.zeros(512 * 64, dtype=torch.int64, device="cuda")
    lib.diffsal_set_front_stamps(buf.data_ptr(), buf.numel() * 8)
    run()
    torch.cuda.synchronize()
    lib.diffsal_set_front_stamps(None, 0)
    s = buf.view(-1, 8, 8).cpu().double()
    s = s[s[:, 0, 0] > 0]
    t0 = s[:, 0, 0].min()
    print(dt, f"C={C} {H}x{W}", "workgroups", s.shape[0], "span us", (s[:, :, 5].max() - t0).item() / 100)
    names = ["A: halo + LN1 -> LDS", "B: dwconv + LNq", "C: K/V -> LDS", "D: q-proj, attention", "E: proj + store"]
    for it in range(min(6, s.shape[1])):
        live = s[:, it, 5] > 0
        if not live.any():
            break
        row = []
        for kph in range(5):
            d = (s[live, it, kph + 1] - s[live, it, kph]) / 100
            row.append(f"{d.mean().item():6.2f}")
        print(f"  tile {it} ({int(live.sum())} WGs): " + "  ".join(f"{n.split(':')[0]} {v}" for n, v in zip(names, row)),
              f" total {((s[live, it, 5] - s[live, it, 0]) / 100).mean().item():6.2f} us")
