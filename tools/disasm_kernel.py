#!/usr/bin/env python3
"""Disassemble the gfx950 code object of one translation unit and print per-kernel instruction statistics (or the ISA of one kernel).
usage: tools/disasm_kernel.py csrc/tapsum.o [substring-of-kernel-name [--dump]]   (needs /opt/rocm/lib/llvm/bin; no GPU)"""
import os, re, struct, subprocess, sys, tempfile
from collections import Counter

LLVM = "/opt/rocm/lib/llvm/bin"
obj = sys.argv[1]
pat = sys.argv[2] if len(sys.argv) > 2 else ""
dump = "--dump" in sys.argv
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"
blob = open(obj, "rb").read()
with tempfile.TemporaryDirectory() as d:
    pos = blob.find(MAGIC)
    while pos >= 0:
        n = struct.unpack_from("<Q", blob, pos + 24)[0]
        q = pos + 32
        for _ in range(n):
            off, size, tl = struct.unpack_from("<QQQ", blob, q)
            triple = blob[q + 24:q + 24 + tl].decode()
            q += 24 + tl
            if "gfx950" in triple and size:
                co = os.path.join(d, "co.o")
                open(co, "wb").write(blob[pos + off:pos + off + size])
                txt = subprocess.check_output([f"{LLVM}/llvm-objdump", "-d", co], text=True)
                notes = subprocess.check_output([f"{LLVM}/llvm-readelf", "--notes", co], text=True)
                kn = re.findall(r"\.name:\s+(_Z\S+|diffsal\S+)", notes)
                vg = dict(zip(kn, re.findall(r"\.vgpr_count:\s+(\d+)", notes)))
                ag = dict(zip(kn, re.findall(r"\.agpr_count:\s+(\d+)", notes)))
                for m in re.finditer(r"^[0-9a-f]+ <([^>]+)>:\n((?:.+\n)+)", txt, re.M):
                    name, body = m.group(1), m.group(2)
                    regs = f"vgpr {vg.get(name, '?')} (agpr {ag.get(name, '?')})"
                    name = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip() or name
                    if pat not in name or name.endswith(".kd"):
                        continue
                    ins = [l.split("\t")[1].split()[0] for l in body.splitlines() if "\t" in l and len(l.split("\t")) > 1 and l.split("\t")[1].strip()]
                    c = Counter(ins)
                    fam = Counter()
                    for k, v in c.items():
                        fam["mfma" if "mfma" in k else "valu" if k.startswith("v_") else "salu" if k.startswith("s_") and "waitcnt" not in k else
                            "waitcnt" if "waitcnt" in k else "vmem" if k.startswith(("global_", "buffer_", "flat_", "scratch_")) else "lds" if k.startswith("ds_") else "other"] += v
                    print(f"{name[:110]}\n   {len(ins)} instructions: {dict(fam)}  {regs}  pk_fma {c.get('v_pk_fma_f32', 0)} fma {c.get('v_fma_f32', 0) + c.get('v_fmac_f32_e32', 0)}")
                    if dump:
                        print(body)
        pos = blob.find(MAGIC, pos + 24)
