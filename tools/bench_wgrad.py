#!/usr/bin/env python3
"""Time every weight-gradient (and data-gradient) GEMM of one full-size training step, per unique shape.
usage: tools/bench_wgrad.py [av|vis] [batch]"""
import os, sys, collections, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from diff_sal_amd import ops

av = (sys.argv[1] if len(sys.argv) > 1 else "av") == "av"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 4
dev = torch.device("cuda", 0)
cfg = bench.Config()
net, sd = bench.build_net(cfg, dev)
H, W = cfg.img_size
g = torch.Generator().manual_seed(0)
x = torch.randn((B, 1, H, W), generator=g).to(dev)
feats = [torch.randn((B, c, 8, H // s, W // s), generator=g).to(dev) for c, s in zip(cfg.up_channel, (32, 16, 8, 4))]
audio = torch.randn((B, 512, 9, H // 32, W // 32), generator=g).to(dev) if av else None
t = torch.full((B,), 500, device=dev)
rec = collections.OrderedDict()
o_w, o_c = ops.conv_wgrad, ops.conv_igemm
state = {"bwd": False}
def spy_w(xi, dy, **kw):
    key = ("wgrad", tuple(xi.shape), tuple(dy.shape), tuple(sorted(kw.items())))
    rec.setdefault(key, [0, (xi, dy, kw)])[0] += 1
    return o_w(xi, dy, **kw)
def spy_c(xi, w, **kw):
    y = o_c(xi, w, **kw)
    if state["bwd"]:
        kk = {k: v for k, v in kw.items() if k in ("kh", "kw", "stride", "pad", "dil", "out_hw")}
        key = ("dgrad", tuple(xi.shape), tuple(y.shape), tuple(sorted(kk.items())))
        rec.setdefault(key, [0, (xi, w, kk)])[0] += 1
    return y
ops.conv_wgrad, ops.conv_igemm = spy_w, spy_c
net.train()
out = net(x, t, feats, audio)
state["bwd"] = True
out.sum().backward()
state["bwd"] = False
ops.conv_wgrad, ops.conv_igemm = o_w, o_c
tot = {"wgrad": [0.0, 0.0], "dgrad": [0.0, 0.0]}
rows = []
for key, (cnt, (a, b, kw)) in rec.items():
    fn = o_w if key[0] == "wgrad" else o_c
    for _ in range(2): fn(a, b, **kw)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): fn(a, b, **kw)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 5
    if key[0] == "wgrad":
        N, Hh, Ww, Cin = a.shape; _, Ho, Wo, Cout = b.shape
        fl = 2.0 * N * Ho * Wo * Cout * kw.get("kh", 1) * kw.get("kw", 1) * Cin
        desc = f"x{tuple(a.shape)} dy{tuple(b.shape)} k{kw.get('kh',1)}x{kw.get('kw',1)}"
    else:
        N, Hh, Ww, Cin = a.shape; Cout = key[2][-1]; Ho, Wo = key[2][1:3]
        fl = 2.0 * N * Ho * Wo * Cout * kw.get("kh", 1) * kw.get("kw", 1) * Cin
        desc = f"g{tuple(a.shape)} -> {key[2]} k{kw.get('kh',1)}x{kw.get('kw',1)}"
    tot[key[0]][0] += us * cnt; tot[key[0]][1] += fl * cnt
    rows.append((us * cnt, key[0], cnt, us, fl / us / 1e6, desc))
rows.sort(reverse=True)
for r in rows[:40]:
    print("%8.1f us total  %s x%d  %8.1f us  %6.1f TF  %s" % r)
for k, (us, fl) in tot.items():
    print(f"{k}: {us/1e3:.2f} ms, {fl/1e9:.1f} GFLOP, {fl/us/1e6:.1f} TF/s")
