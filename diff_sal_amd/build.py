"""Builds libdiffsal_hip.so (gfx950) in-tree with hipcc.  `python -m diff_sal_amd.build`."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libdiffsal_hip.so")
SOURCES = ["igemm.hip", "igemm16.hip", "lin_stream.hip", "wgrad.hip", "backward.hip", "optim.hip", "pack.hip", "norm.hip", "misc.hip", "metrics.hip", "attention.hip", "mvit.hip", "block16.hip", "conv16_halo.hip", "tapsum.hip"]


def _hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    return "hipcc"


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, s) for s in SOURCES] + [
        os.path.join(CSRC, "common.h"),
        os.path.join(os.path.dirname(HERE), "include", "diffsal.h"),
    ]
    # an installed copy may ship the binary without its sources: nothing to compare against, nothing to rebuild
    return any(os.path.exists(d) and os.path.getmtime(d) > t for d in deps)


def build_library(force=False, verbose=False):
    """Compile every HIP source for gfx950 into one shared object next to the package."""
    if not force and not needs_build():
        return LIB
    objs, procs = [], []
    for s in SOURCES:  # one hipcc per translation unit, all at once (8 files, a few hundred MB each)
        o = os.path.join(CSRC, s.replace(".hip", ".o"))
        cmd = [_hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-c", os.path.join(CSRC, s), "-o", o]
        if verbose:
            print(" ".join(cmd))
        procs.append((cmd, subprocess.Popen(cmd)))
        objs.append(o)
    failed = [cmd for cmd, pr in procs if pr.wait() != 0]
    if failed:
        raise subprocess.CalledProcessError(1, failed[0])
    cmd = [_hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build_library(force="--force" in sys.argv, verbose=True))
