#!/usr/bin/env python3
"""Turn two rocprofv3 --pmc passes (FETCH_SIZE / WRITE_SIZE, collected separately as the MI355X guide prescribes) of `bench.py`
into profiles/rNN_hbm_traffic_<precision>.json: HBM bytes per step of the GEMM family AND per kernel name (so that traffic above
the once-through bytes can be pinned on a kernel instead of a family total).

usage: tools/pmc_traffic.py <fetch_dir> <write_dir> <out.json> [fp32|bf16|fp16] [build-id] [launches.json]
gfx950 correction: FETCH_SIZE reports 1/2 of the bytes of wide coalesced reads -> doubled; WRITE_SIZE is exact.  Both in KiB.
With a launches.json (bench.py --dump-launches) the once-through bytes of the same step are put beside the measured ones."""
import csv
import glob
import json
import os
import re
import sys


def load(d, counter):
    f = max(glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True), key=os.path.getmtime)   # newest run
    per = {}
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != counter:
            continue
        name = re.sub(r"\(.*", "", r["Kernel_Name"])             # drop the argument list
        name = re.sub(r"^void ", "", name).replace("diffsal::", "")
        e = per.setdefault(name, [0.0, 0])
        e[0] += float(r["Counter_Value"])
        e[1] += 1
    return per


def main():
    fetch_dir, write_dir, out = sys.argv[1:4]
    prec = sys.argv[4] if len(sys.argv) > 4 else "fp32"
    build = sys.argv[5] if len(sys.argv) > 5 else "build n/a"
    launches = sys.argv[6] if len(sys.argv) > 6 else None
    fam = (["igemm_kernel", "igemm_linear_kernel", "splitk_reduce_kernel", "lin_stream_kernel", "mlp_block_kernel", "tapsum_kernel",
            "block_front_kernel", "wino_gemm_kernel", "wino_input_kernel", "wino_reduce_kernel", "gemm_dma_kernel",
            "gemm_dma_reduce_kernel"] if prec == "fp32" else
           ["igemm16_kernel", "igemm16_linear_kernel", "splitk16_reduce_kernel", "conv16_halo_kernel", "conv16_dma_kernel", "gemm16_dma2_kernel", "block16_kernel",
            "block_front_kernel", "gemm_dma_kernel", "gemm_dma_reduce_kernel"])
    fe, wr = load(fetch_dir, "FETCH_SIZE"), load(write_dir, "WRITE_SIZE")
    steps = sum(n for k, (_, n) in fe.items() if "conv_in" in k)    # one conv_in (or fused conv_in_s4) launch per denoising step
    assert steps > 0
    table, fam_r, fam_w, fam_n = [], 0.0, 0.0, 0
    for k in sorted(set(fe) | set(wr)):
        r_b = 2.0 * fe.get(k, [0.0, 0])[0] * 1024.0
        w_b = wr.get(k, [0.0, 0])[0] * 1024.0
        n = max(fe.get(k, [0, 0])[1], wr.get(k, [0, 0])[1])
        in_fam = any(m in k for m in fam)
        if in_fam:
            fam_r, fam_w, fam_n = fam_r + r_b, fam_w + w_b, fam_n + n
        table.append({"kernel": k[:160], "gemm_family": in_fam, "launches_per_step": round(n / steps, 2),
                      "read_mb_per_step": round(r_b / steps / 1e6, 2), "write_mb_per_step": round(w_b / steps / 1e6, 2)})
    table.sort(key=lambda t: -(t["read_mb_per_step"] + t["write_mb_per_step"]))
    res = {
        "kernel": " + ".join(fam), "precision": prec, "build": build, "launches_profiled": fam_n,
        "read_bytes_per_launch": fam_r / max(fam_n, 1), "write_bytes_per_launch": fam_w / max(fam_n, 1),
        "bytes_per_launch": (fam_r + fam_w) / max(fam_n, 1), "steps_profiled": steps, "bytes_per_step": (fam_r + fam_w) / steps,
        "all_kernels_bytes_per_step": sum(t["read_mb_per_step"] + t["write_mb_per_step"] for t in table) * 1e6,
        "per_kernel": table,
        "method": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes over bench.py (B=4); "
                  "FETCH_SIZE doubled (gfx950 reports half of wide coalesced reads), counters in KiB",
    }
    if launches and os.path.exists(launches):
        L = json.load(open(launches))["launches"]
        res["once_through_mb_per_step"] = round(sum(l["mbytes"] for l in L), 1)
        by = {}
        for l in L:
            by[l["class"]] = by.get(l["class"], 0.0) + l["mbytes"]
        res["once_through_mb_by_class"] = {k: round(v, 1) for k, v in sorted(by.items(), key=lambda kv: -kv[1])}
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps({k: v for k, v in res.items() if k != "per_kernel"}, indent=1))
    for t in table[:25]:
        print(t)


if __name__ == "__main__":
    main()
