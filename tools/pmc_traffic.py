#!/usr/bin/env python3
"""Turn two rocprofv3 --pmc passes (FETCH_SIZE / WRITE_SIZE, collected separately as the MI355X guide prescribes)
of `bench.py` into profiles/igemm_hbm_traffic.json: HBM bytes per launch of the implicit-GEMM kernel.

usage: tools/pmc_traffic.py <fetch_dir> <write_dir> <out.json> [fp32|bf16|fp16] [build-id]
gfx950 correction: FETCH_SIZE reports 1/2 of the bytes of wide coalesced reads -> doubled; WRITE_SIZE is exact.
Both counters are in KiB."""
import csv
import glob
import json
import os
import sys


def per_kernel(d, counter, match):
    f = max(glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True), key=os.path.getmtime)   # newest run
    tot, n = 0.0, 0
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == counter and any(m in r["Kernel_Name"] for m in match):
            tot += float(r["Counter_Value"])
            n += 1
    return tot, n


def main():
    fetch_dir, write_dir, out = sys.argv[1:4]
    prec = sys.argv[4] if len(sys.argv) > 4 else "fp32"
    build = sys.argv[5] if len(sys.argv) > 5 else "build n/a"
    # the GEMM family of the precision: tiled implicit GEMM (+ split-K reduce), streaming linears, fused MLP / block kernels
    match = (["igemm_kernel", "igemm_linear_kernel", "splitk_reduce_kernel", "lin_stream_kernel", "mlp_block_kernel", "tapsum_kernel"] if prec == "fp32" else
             ["igemm16_kernel", "igemm16_linear_kernel", "splitk16_reduce_kernel", "conv16_halo_kernel", "block16_kernel"])
    fk, nf = per_kernel(fetch_dir, "FETCH_SIZE", match)
    wk, nw = per_kernel(write_dir, "WRITE_SIZE", match)
    assert nf == nw and nf > 0, (nf, nw)
    _, steps = per_kernel(fetch_dir, "FETCH_SIZE", ["conv_in"])      # one conv_in (or fused conv_in_s4) launch per denoising step
    assert steps > 0
    read_b, write_b = 2.0 * fk * 1024.0, wk * 1024.0
    res = {
        "kernel": " + ".join(match), "precision": prec, "build": build, "launches_profiled": nf,
        "read_bytes_per_launch": read_b / nf, "write_bytes_per_launch": write_b / nw,
        "bytes_per_launch": (read_b + write_b) / nf, "steps_profiled": steps, "bytes_per_step": (read_b + write_b) / steps,
        "method": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes over bench.py (vis, B=4); "
                  "FETCH_SIZE doubled (gfx950 reports half of wide coalesced reads), counters in KiB",
    }
    json.dump(res, open(out, "w"), indent=1)
    print(res)


if __name__ == "__main__":
    main()
