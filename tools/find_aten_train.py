"""Which Python lines launch the stock (non-diffsal) operators of a TRAINING step (BASELINE configs[3], full AV model)?  A TorchDispatchMode
logs every ATen operator that launches a kernel with the innermost diff_sal_amd frame (autograd's backward runs in the calling thread for
this: set_multithreading_enabled(False)).     python3 tools/find_aten_train.py"""
import os
import sys
import traceback
from collections import Counter

import torch
from torch.utils._python_dispatch import TorchDispatchMode

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

WATCH = ("copy_", "fill_", "clone", "cat", "zeros", "zero_", "add", "mul", "sub", "div", "sin", "cos", "exp", "sum", "index_select",
         "_to_copy", "zeros_like", "ones_like", "full", "neg", "sqrt", "clamp", "where", "stack", "split_with_sizes", "slice_backward",
         "select_backward", "as_strided_", "new_zeros", "normal_", "uniform_", "rand", "randn")


class Log(TorchDispatchMode):
    def __init__(self):
        super().__init__()
        self.c = Counter()

    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = func.__name__.split(".")[0]
        if name in WATCH:
            big = any(isinstance(a, torch.Tensor) and a.is_cuda for a in args)
            if big:
                fr = [f for f in traceback.extract_stack() if "/diff_sal_amd/" in f.filename or f.filename.endswith("bench.py")]
                where = " <- ".join(f"{os.path.basename(f.filename)}:{f.lineno}" for f in fr[-2:][::-1]) if fr else "(autograd engine)"
                numel = max((a.numel() for a in args if isinstance(a, torch.Tensor)), default=0)
                self.c[(name, where, numel)] += 1
        return func(*args, **(kwargs or {}))


dev = torch.device("cuda:0")
cfg = bench.Config()
net, _ = bench.build_net(cfg, dev)
ts, sal, cond = bench.build_train_step(cfg, net, None, None, dev, 0, batch=4, av=True, full=True)
for _ in range(2):
    ts.step(sal, cond)
torch.cuda.synchronize()
torch.autograd.set_multithreading_enabled(False)
with Log() as log:
    ts.step(sal, cond)
torch.cuda.synchronize()
tot = Counter()
for (name, where, numel), k in log.c.items():
    tot[(name, where)] += k
print("ATen operators on GPU tensors in one training step (count, operator, innermost frames, largest operand):")
big = {}
for (name, where, numel), k in log.c.items():
    big[(name, where)] = max(big.get((name, where), 0), numel)
for (name, where), k in tot.most_common(60):
    print(f"  {k:4d}  {name:14s} {big[(name, where)]:>12d}  {where}")
