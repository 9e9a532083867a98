#!/usr/bin/env python3
"""List the kernels of libdiffsal_hip.so that use scratch memory (register spills or private arrays), from the code object's
metadata notes.  usage: tools/check_scratch.py [lib.so]   (needs /opt/rocm/lib/llvm/bin; no GPU)"""
import os, re, struct, subprocess, sys, tempfile

LLVM = "/opt/rocm/lib/llvm/bin"
lib = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(__file__), "..", "diff_sal_amd", "libdiffsal_hip.so")
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"
blob = open(lib, "rb").read()
notes = ""
with tempfile.TemporaryDirectory() as d:
    pos, idx = blob.find(MAGIC), 0
    while pos >= 0:                                   # one bundle per translation unit
        n = struct.unpack_from("<Q", blob, pos + 24)[0]
        q = pos + 32
        for _ in range(n):
            off, size, tl = struct.unpack_from("<QQQ", blob, q)
            triple = blob[q + 24:q + 24 + tl].decode()
            q += 24 + tl
            if "gfx950" in triple and size:
                co = os.path.join(d, f"co{idx}.o"); idx += 1
                open(co, "wb").write(blob[pos + off:pos + off + size])
                notes += subprocess.check_output([f"{LLVM}/llvm-readelf", "--notes", co], text=True)
        pos = blob.find(MAGIC, pos + 24)
names = re.findall(r"\.name:\s+(\S+)", notes)
priv = [int(x) for x in re.findall(r"\.private_segment_fixed_size:\s+(\d+)", notes)]
vg = [int(x) for x in re.findall(r"\.vgpr_count:\s+(\d+)", notes)]
kern = [n for n in names if n.startswith("_Z") or n.startswith("diffsal")]
spill = [int(x) for x in re.findall(r"\.vgpr_spill_count:\s+(\d+)", notes)]
print(f"{len(priv)} kernels")
bad = 0
for i, b in enumerate(priv):
    if b:
        bad += 1
        nm = subprocess.run(["c++filt", kern[i]], capture_output=True, text=True).stdout.strip() if i < len(kern) else "?"
        print(f"  scratch {b:6d} B  spilled vgprs {spill[i] if i < len(spill) else '?':>4}  {nm[:150]}")
print(f"{bad} kernels with scratch")
