"""Which Python lines launch the stock (non-diffsal) kernels of a denoising step?  torch.profiler over one 50-NFE trajectory of
bench.py's sampler, kernels grouped by name with the Python stack of the operator that launched them.
    python3 tools/find_aten.py [--precision bf16]"""
import argparse
import os
import sys
from collections import Counter, defaultdict

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from diff_sal_amd.sampling import DiffusionSampler  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--precision", default="fp32")
ap.add_argument("--batch", type=int, default=4)
a = ap.parse_args()
dev = torch.device("cuda:0")
cfg = bench.Config()
net, _ = bench.build_net(cfg, dev)
net.compute_dtype = {"bf16": torch.bfloat16, "fp16": torch.float16}.get(a.precision, torch.float32)
g = torch.Generator(device="cpu").manual_seed(1234)
H, W = cfg.img_size
B = a.batch
x_T = torch.randn((B, 1, H, W), generator=g).to(dev)
feats = [torch.randn((B, c, 8, H // s, W // s), generator=g).to(dev) for c, s in zip(cfg.up_channel, (32, 16, 8, 4))]
s = DiffusionSampler(bench.Top(net), timesteps=50, sample_type="dpmsolver", skip_type="logSNR", denoise=True, training_target="x0",
                     hip_graph=False)
s.sample_dpm_solver(x_T, feats, None)
torch.cuda.synchronize()
from torch.profiler import ProfilerActivity, profile  # noqa: E402

with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    s.sample_dpm_solver(x_T, feats, None)
    torch.cuda.synchronize()
ev = prof.events()
byk = defaultdict(Counter)
cnt = Counter()
cpu_ops = [e for e in ev if e.device_type == torch.autograd.DeviceType.CPU]
for e in ev:
    if e.device_type != torch.autograd.DeviceType.CUDA:
        continue
    n = e.name
    if "diffsal" in n or n.startswith("void gn_slab") or n.startswith("gn_"):
        continue
    cnt[n[:100]] += 1
# map: launching CPU op -> stack
for e in cpu_ops:
    if e.name.startswith("aten::") and e.stack:
        st = [f for f in e.stack if "diff_sal_amd" in f or "bench.py" in f][:3]
        if st and e.name in ("aten::index_select", "aten::copy_", "aten::fill_", "aten::clone", "aten::contiguous", "aten::full", "aten::cat",
                             "aten::to", "aten::_to_copy", "aten::zeros", "aten::zero_", "aten::empty_like", "aten::mul", "aten::add"):
            byk[e.name][" <- ".join(st)] += 1
print("stock kernels over one 50-step trajectory:")
for n, c in cnt.most_common():
    print(f"  {c:5d}  {n}")
print("\nATen operators with a diff_sal_amd / bench frame on the stack:")
for op, c in byk.items():
    for st, k in c.most_common(12):
        print(f"  {k:5d}  {op:22s} {st}")
