#!/usr/bin/env python3
"""Print the kernel timeline of one denoise step from a rocprofv3 --kernel-trace CSV.

usage: tools/step_timeline.py <dir-or-csv> [step_index] [delimiter-kernel-substring]
Steps are delimited by the temb kernel (first launch of SalUNet.forward) unless another kernel name is given
(training: "adam_kernel", the last launch of a step)."""
import csv
import glob
import os
import sys


def main():
    src = sys.argv[1]
    which = int(sys.argv[2]) if len(sys.argv) > 2 else 30
    if os.path.isdir(src):
        src = sorted(glob.glob(os.path.join(src, "**", "*kernel_trace.csv"), recursive=True))[0]
    rows = list(csv.DictReader(open(src)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    delim = sys.argv[3] if len(sys.argv) > 3 else None
    if delim:
        idx = [i + 1 for i, r in enumerate(rows) if delim in r["Kernel_Name"]]
    else:
        idx = [i for i, r in enumerate(rows) if "temb_dense0" in r["Kernel_Name"] or "temb_kernel" in r["Kernel_Name"]]
    a, b = idx[which], idx[which + 1]
    t0 = int(rows[a]["Start_Timestamp"])
    tot, groups = 0.0, {}
    for r in rows[a:b]:
        d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        tot += d
        name = r["Kernel_Name"].split("(")[0].replace("void diffsal::", "").replace("diffsal::", "")
        groups[name.split("<")[0]] = groups.get(name.split("<")[0], 0.0) + d
        gx = int(r["Grid_Size_X"]) // int(r["Workgroup_Size_X"])
        print(f"{(int(r['Start_Timestamp']) - t0) / 1e3:9.1f} {d:8.1f}  {name[:44]:44s} grid={gx:6d}x{r['Grid_Size_Y']}")
    print(f"sum of kernels {tot:.1f} us; span {(int(rows[b]['Start_Timestamp']) - t0) / 1e3:.1f} us")
    for k, v in sorted(groups.items(), key=lambda kv: -kv[1]):
        print(f"  {k:32s} {v:9.1f} us  {100 * v / tot:5.1f} %")


if __name__ == "__main__":
    main()
