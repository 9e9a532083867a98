#!/usr/bin/env python3
"""mt_proj's head gather (csrc/tapsum.hip, row-streamed kernel) at 4 .. 64 clips: 4 x 4 against 8 x 4 patches per wavefront,
interleaved timing in one process.  GPU only.  usage: tools/bench_tapsum_head.py [clips ...]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diff_sal_amd import _lib, ops  # noqa: E402


FORMS = (1, 3, 4, None)


def main():
    clips = [int(a) for a in sys.argv[1:]] or [4, 8, 16, 64]
    dev = "cuda"
    g = torch.Generator(device=dev).manual_seed(3)
    H, W, C = 112, 192, 96
    for B in clips:
        ys = [torch.randn(B, H // f, W // f, 9 * C, device=dev, generator=g) for f in (16, 8, 4, 2)]
        b, sc, sh = (torch.randn(C, device=dev, generator=g) * 0.1 for _ in range(3))
        hw, hb = torch.randn(C, device=dev, generator=g) * 0.3, torch.zeros(1, device=dev)

        def run():
            return ops.tapsum(ys, H, W, C, dil=1, bias=b, scale=1 + sc, shift=sh, act=ops.ACT_RELU, head=(hw, hb))
        outs, times = {}, {}
        for form in FORMS:
            _lib.set_tuning("DIFFSAL_TAPSUM_ROWS_FORM", form)
            outs[form] = run()
            times[form] = []
        for _ in range(5):
            for form in FORMS:
                _lib.set_tuning("DIFFSAL_TAPSUM_ROWS_FORM", form)
                run()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(10):
                    run()
                e1.record()
                torch.cuda.synchronize()
                times[form].append(e0.elapsed_time(e1) * 100)
        _lib.set_tuning("DIFFSAL_TAPSUM_ROWS_FORM", None)
        nbytes = sum(y.numel() for y in ys) * 4
        cells = []
        for form in FORMS:
            t = sorted(times[form])[2]
            cells.append(f"form {form}: {t:8.1f} us ({nbytes / t / 1e6:5.2f} TB/s)")
        print(f"clips {B:3d}: " + "   ".join(cells) + f"   equal {all(torch.equal(outs[1], outs[f]) for f in FORMS)}", flush=True)


if __name__ == "__main__":
    main()
