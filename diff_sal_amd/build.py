"""Builds libdiffsal_hip.so (gfx950) in-tree with hipcc.  `python -m diff_sal_amd.build`."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libdiffsal_hip.so")
SOURCES = ["igemm.hip", "igemm16.hip", "lin_stream.hip", "wgrad.hip", "backward.hip", "optim.hip", "pack.hip", "norm.hip", "misc.hip", "metrics.hip", "attention.hip", "mvit.hip", "mvit_pool.hip", "block16.hip", "conv16_halo.hip", "conv16_dma.hip", "gemm16_dma.hip", "attn16_mfma.hip", "tapsum.hip", "tblock.hip", "wino.hip", "gemm_dma.hip", "wino4.hip", "upconv.hip"]


def _hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    return "hipcc"


FLAGS_STAMP = os.path.join(CSRC, ".build_flags")


def _flags_changed(extra):
    """The objects / library on disk were built with another DIFFSAL_EXTRA_HIPCC_FLAGS set (e.g. -DDIFFSAL_DEV_STAMPS): a plain
    build must not take an instrumented library for up to date (and vice versa)."""
    try:
        return open(FLAGS_STAMP).read() != " ".join(extra)
    except OSError:
        return bool(extra)


def needs_build():
    if not os.path.exists(LIB):
        return True
    if _flags_changed(os.environ.get("DIFFSAL_EXTRA_HIPCC_FLAGS", "").split()):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, s) for s in SOURCES] + [
        os.path.join(CSRC, "common.h"),
        os.path.join(os.path.dirname(HERE), "include", "diffsal.h"),
    ]
    # an installed copy may ship the binary without its sources: nothing to compare against, nothing to rebuild
    return any(os.path.exists(d) and os.path.getmtime(d) > t for d in deps)


def _jobs():
    """Concurrent hipcc processes: DIFFSAL_BUILD_JOBS, else half the host's cores (each compile needs a few hundred MB)."""
    env = os.environ.get("DIFFSAL_BUILD_JOBS")
    if env and env.isdigit() and int(env) > 0:
        return int(env)
    return max(1, min(len(SOURCES), (os.cpu_count() or 2) // 2))


def _stale(src, obj, common_deps):
    if not os.path.exists(obj):
        return True
    t = os.path.getmtime(obj)
    return any(os.path.exists(d) and os.path.getmtime(d) > t for d in [src] + common_deps)


def build_library(force=False, verbose=False):
    """Compile the HIP sources for gfx950 into one shared object next to the package.  Only translation units whose object
    is older than the source or a shared header are recompiled; at most _jobs() hipcc processes run at a time.  Extra compiler
    flags (e.g. -DDIFFSAL_DEV_STAMPS for tools/probe_halo_stamps.py) come from DIFFSAL_EXTRA_HIPCC_FLAGS."""
    if not force and not needs_build():
        return LIB
    common = [os.path.join(CSRC, "common.h"), os.path.join(os.path.dirname(HERE), "include", "diffsal.h")]
    extra = os.environ.get("DIFFSAL_EXTRA_HIPCC_FLAGS", "").split()
    force = force or _flags_changed(extra)
    objs, todo = [], []
    for s in SOURCES:
        src, o = os.path.join(CSRC, s), os.path.join(CSRC, s.replace(".hip", ".o"))
        objs.append(o)
        if force or _stale(src, o, common):
            todo.append([_hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC"] + extra + ["-c", src, "-o", o])
    running, failed, jobs = [], [], _jobs()
    while todo or running:
        while todo and len(running) < jobs:
            cmd = todo.pop(0)
            if verbose:
                print(" ".join(cmd), flush=True)
            running.append((cmd, subprocess.Popen(cmd)))
        cmd, pr = running.pop(0)
        if pr.wait() != 0:
            failed.append(cmd)
    if failed:
        raise subprocess.CalledProcessError(1, failed[0])
    cmd = [_hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    with open(FLAGS_STAMP, "w") as f:
        f.write(" ".join(extra))
    return LIB


if __name__ == "__main__":
    print(build_library(force="--force" in sys.argv, verbose=True))
