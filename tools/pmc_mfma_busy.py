#!/usr/bin/env python3
"""Matrix-pipe busy fraction and effective clock per kernel from a rocprofv3 --pmc pass (GRBM_GUI_ACTIVE
SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES) joined with a --kernel-trace pass of the same command (durations).
MI355X_MICROARCH.md: effective clock = GRBM_GUI_ACTIVE / 8 / kernel time (reads high on dispatches under ~0.3 ms);
SQ_VALU_MFMA_BUSY_CYCLES sums over the 1024 SIMDs, so busy fraction = MFMA_BUSY / (GUI_ACTIVE / 8 * 1024).
usage: tools/pmc_mfma_busy.py <pmc_dir> <trace_dir> [min_us]"""
import collections
import csv
import glob
import os
import sys


def first(d, pat):
    return sorted(glob.glob(os.path.join(d, "**", pat), recursive=True))[0]


def main():
    pmc_dir, trace_dir = sys.argv[1:3]
    min_us = float(sys.argv[3]) if len(sys.argv) > 3 else 20.0
    cnt = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(first(pmc_dir, "*counter_collection.csv"))):
        cnt[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    dur = collections.defaultdict(list)
    for r in csv.DictReader(open(first(trace_dir, "*kernel_trace.csv"))):
        dur[r["Kernel_Name"]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    rows = []
    for k, cs in cnt.items():
        if "GRBM_GUI_ACTIVE" not in cs or k not in dur:
            continue
        n = len(cs["GRBM_GUI_ACTIVE"])
        gui = sum(cs["GRBM_GUI_ACTIVE"]) / n
        mf = sum(cs.get("SQ_VALU_MFMA_BUSY_CYCLES", [0.0])) / max(len(cs.get("SQ_VALU_MFMA_BUSY_CYCLES", [0.0])), 1)
        us = sum(dur[k]) / len(dur[k])
        if us < min_us:
            continue
        rows.append((us * len(dur[k]), k[:90], n, us, gui / 8.0 / us / 1e3, mf / (gui / 8.0 * 1024.0) if gui > 0 else 0.0))
    rows.sort(reverse=True)
    print("| kernel | launches (pmc pass) | mean us (trace pass) | effective clock GHz | MFMA busy of all SIMD cycles |")
    print("|---|---|---|---|---|")
    for _, k, n, us, ghz, busy in rows[:24]:
        print(f"| `{k}` | {n} | {us:.1f} | {ghz:.2f} | {busy:.3f} |")


if __name__ == "__main__":
    main()
