#!/usr/bin/env python3
"""Idle time between consecutive kernels of a rocprofv3 --kernel-trace CSV: busy vs gap share over the steady-state tail of
the trace, and the gap that follows each kernel name.  usage: tools/trace_gaps.py <kernel_trace.csv> [n_last_kernels]"""
import csv
import sys
from collections import defaultdict

rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
n = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
rows = rows[-n:]
busy = sum(e - s for s, e, _ in rows)
span = rows[-1][1] - rows[0][0]
gaps = defaultdict(lambda: [0, 0.0, 0.0])
tot_gap = 0
for (s0, e0, k0), (s1, e1, k1) in zip(rows, rows[1:]):
    g = max(0, s1 - e0)
    tot_gap += g
    a = gaps[k0[:70]]
    a[0] += 1
    a[1] += g
    a[2] += e0 - s0
print(f"kernels {len(rows)} span {span / 1e3:.1f} us busy {busy / 1e3:.1f} us ({busy / span:.3f}) gaps {tot_gap / 1e3:.1f} us ({tot_gap / span:.3f}) "
      f"mean gap {tot_gap / len(rows) / 1e3:.2f} us")
for k, (c, g, d) in sorted(gaps.items(), key=lambda kv: -kv[1][1])[:25]:
    print(f"{c:6d} x  gap-after mean {g / c / 1e3:6.2f} us  dur mean {d / c / 1e3:8.2f} us  {k}")
