#!/usr/bin/env python3
"""Print the fp32 planner's tile shape / split-K for every GEMM-family launch of one denoising step (DIFFSAL_PLAN_DEBUG=1)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
env = dict(os.environ, DIFFSAL_PLAN_DEBUG="1")
out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1", "--warmup", "1", "--repeats", "1", "--no-cpu-baseline",
                      "--no-encoders", "--no-alt-precision", "--no-reference-graph"], env=env, capture_output=True, text=True)
seen = []
for line in out.stderr.splitlines():
    if line.startswith("[diffsal plan]") and line not in seen:
        seen.append(line)
print("\n".join(seen))
