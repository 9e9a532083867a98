#!/usr/bin/env python3
"""Identity of the code a measurement was taken on: a hash of every source that determines the kernels and the host path
(csrc/*.hip, csrc/common.h, include/diffsal.h, diff_sal_amd/*.py, bench.py).  Independent of git (the GPU box has no .git and
the artefacts are committed after they are produced): bench.py recomputes it and refuses a profile whose id differs."""
import glob
import hashlib
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def source_id() -> str:
    files = sorted(glob.glob(os.path.join(ROOT, "diff_sal_amd", "csrc", "*.hip")) + glob.glob(os.path.join(ROOT, "diff_sal_amd", "csrc", "*.h"))
                   + glob.glob(os.path.join(ROOT, "include", "*.h")) + glob.glob(os.path.join(ROOT, "diff_sal_amd", "*.py"))
                   + [os.path.join(ROOT, "bench.py")])
    h = hashlib.sha1()
    for f in files:
        h.update(os.path.relpath(f, ROOT).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()[:12]


if __name__ == "__main__":
    print(source_id())
