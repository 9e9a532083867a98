#!/usr/bin/env python3
"""Where the HOST time of a training step goes (the step is close to issue-bound): cProfile over a few steps of the bench's
training workload, top functions by own and cumulative time.  usage: python tools/host_profile_train.py [steps]"""
import cProfile
import io
import os
import pstats
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    torch.manual_seed(0)
    cfg = bench.Config()
    net, _ = bench.build_net(cfg, dev)
    ts, sal, cond = bench.build_train_step(cfg, net, None, None, dev, 0, batch=4, av=True, full=True)
    for _ in range(3):
        ts.step(sal, cond)
    torch.cuda.synchronize()
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(steps):
        ts.step(sal, cond)
    pr.disable()
    torch.cuda.synchronize()
    for key in ("tottime", "cumtime"):
        st = io.StringIO()
        pstats.Stats(pr, stream=st).sort_stats(key).print_stats(28)
        print(st.getvalue()[:6000])


if __name__ == "__main__":
    main()
