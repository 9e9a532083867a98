#!/usr/bin/env python3
"""The fused second half of the C = 96 TransformerBlock on 16-bit storage (csrc/block16.hip) at 4 .. 64 clips: four against eight
wavefronts per workgroup, interleaved timing in one process.  GPU only.  usage: tools/bench_block16.py [clips ...]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diff_sal_amd import _lib, ops  # noqa: E402


def main():
    clips = [int(a) for a in sys.argv[1:]] or [4, 16, 64]
    dev, C, HID = "cuda", 96, 192
    g = torch.Generator(device=dev).manual_seed(3)
    for dt in (torch.float16, torch.bfloat16):
        r = lambda *s, sc=1.0: torch.randn(*s, device=dev, generator=g) * sc
        wp, w1, w2 = r(C, C, sc=0.1).to(dt), r(HID, C, sc=0.12).to(dt), r(C, HID, sc=0.08).to(dt)
        bp, b1, b2 = r(C, sc=0.1), r(HID, sc=0.1), r(C, sc=0.1)
        g2, be2, gz, bz = r(C, sc=0.1) + 1, r(C, sc=0.1), r(C, sc=0.1) + 1, r(C, sc=0.1)
        for B in clips:
            M = B * 9 * 56 * 96
            o, x = r(M, C).to(dt), r(M, C).to(dt)

            def run():
                return ops.block16(o, x, (wp, bp), (g2, be2, 1e-5), (w1, b1), (w2, b2), (gz, bz, 1e-5), (56 * 96, 9, 5))
            outs, times = {}, {}
            for nw in (4, 8, None):
                _lib.set_tuning("DIFFSAL_BLOCK16_WAVES", nw)
                outs[nw] = run()
                times[nw] = []
            for _ in range(5):
                for nw in (4, 8, None):
                    _lib.set_tuning("DIFFSAL_BLOCK16_WAVES", nw)
                    run()
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    for _ in range(10):
                        run()
                    e1.record()
                    torch.cuda.synchronize()
                    times[nw].append(e0.elapsed_time(e1) * 100)
            _lib.set_tuning("DIFFSAL_BLOCK16_WAVES", None)
            nbytes = M * C * 2 * (3 + 5 / 9)
            cells = [f"{nw} waves: {sorted(times[nw])[2]:7.1f} us ({nbytes / sorted(times[nw])[2] / 1e6:4.2f} TB/s)" for nw in (4, 8, None)]
            kept = ((torch.arange(M, device=dev) // (56 * 96)) % 9) < 5          # z is written for the kept frames only
            same = all(torch.equal(outs[4][0], outs[nw][0]) and torch.equal(outs[4][1][kept], outs[nw][1][kept]) for nw in (8, None))
            print(f"{str(dt)[6:]:9s} clips {B:3d}: " + "   ".join(cells) + f"   same bits {same}", flush=True)


if __name__ == "__main__":
    main()
