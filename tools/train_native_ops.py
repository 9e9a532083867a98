#!/usr/bin/env python3
"""Where the torch-native kernels of one full-model training step come from: aten ops that launch a GPU kernel (copy_, add, fill_,
zero_, index ...), grouped by the innermost frame inside diff_sal_amd/.  GPU only."""
import collections
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

dev = torch.device("cuda:0")
cfg = bench.Config()
net, _ = bench.build_net(cfg, dev)
ts, sal, cond = bench.build_train_step(cfg, net, None, None, dev, 0, batch=4, av=True, full=True)
for _ in range(3):
    ts.step(sal, cond)
torch.cuda.synchronize()
with torch.profiler.profile(activities=[torch.profiler.ProfilerActivity.CPU, torch.profiler.ProfilerActivity.CUDA], with_stack=True,
                            experimental_config=torch._C._profiler._ExperimentalConfig(verbose=True)) as prof:
    ts.step(sal, cond)
    torch.cuda.synchronize()
agg = collections.Counter()
tim = collections.Counter()
for ev in prof.events():
    if not ev.name.startswith("aten::") or ev.device_time_total <= 0 or not any(k for k in ev.kernels):
        continue
    frame = "?"
    for fr in ev.stack or []:
        if "diff_sal_amd/" in fr:
            frame = fr.split("diff_sal_amd/")[-1]
            break
    if frame == "?" and ev.stack:
        frame = ev.stack[0][-60:]
    agg[(ev.name, frame)] += len(ev.kernels)
    tim[(ev.name, frame)] += sum(k.duration for k in ev.kernels)
for (name, frame), n in agg.most_common(45):
    print(f"{n:4d} launches {tim[(name, frame)]:8.1f} us  {name:22s} {frame}")
print("total native launches", sum(agg.values()), "us", sum(tim.values()))
