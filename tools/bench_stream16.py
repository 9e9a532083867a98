#!/usr/bin/env python3
"""Old (8-byte) against new (16-byte) forms of the HBM-bound kernels on 16-bit storage at the shapes of a 64-clip pass.
usage: tools/bench_stream16.py [filter] [--batch 64] [--dtype fp16]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diff_sal_amd import _lib, ops  # noqa: E402

DEV = "cuda"
STAGES = [(7, 12, 768), (14, 24, 384), (28, 48, 192), (56, 96, 96)]


def timeit(fn, reps=10):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


def ab(name, fn, nbytes):
    cells = []
    for old in (1, None):
        _lib.set_tuning("DIFFSAL_NO_STREAM16", old)
        t = timeit(fn)
        cells.append(f"{t:8.1f} us {nbytes / t / 1e6:6.2f} TB/s")
    _lib.set_tuning("DIFFSAL_NO_STREAM16", None)
    print(f"{name:34s} | old {cells[0]} | new {cells[1]}", flush=True)


def main():
    args = sys.argv[1:]
    B, dt, flt = 64, torch.float16, ""
    i = 0
    while i < len(args):
        if args[i] == "--batch":
            B = int(args[i + 1]); i += 2
        elif args[i] == "--dtype":
            dt = {"fp16": torch.float16, "bf16": torch.bfloat16}[args[i + 1]]; i += 2
        else:
            flt = args[i]; i += 1
    T = 9
    g = torch.Generator(device=DEV).manual_seed(3)
    es = 2
    for si, (H, W, C) in enumerate(STAGES):
        x = torch.randn(B, T, H, W, C, device=DEV, generator=g).to(dt)
        if not flt or "fuse" in flt:
            a_all = torch.randn(B * T, 84, 1440, device=DEV, generator=g).to(dt)
            off = [0, 768, 1152, 1344][si]
            a_small = a_all[:, :, off:off + C]
            ab(f"K7 audio_fuse s{si} {H}x{W}x{C}", lambda: ops.audio_fuse(a_small, x, 7, 12), 2 * x.numel() * es + a_small.numel() * es)
        if not flt or "prep" in flt:
            k = [2, 4, 8, 16][si]
            N = B * T
            xn = x.view(N, H, W, C)
            xk = torch.randn(N, H, W, C, device=DEV, generator=g).to(dt)
            f = lambda *s_: torch.randn(*s_, device=DEV, generator=g)
            w9, wk, wv = f(9, C), f(k * k, C), f(k * k, C)
            gs = [f(C) for _ in range(8)]
            if si < 2:
                ab(f"K9 qkv_prep s{si}", lambda: ops.qkv_prep(xn, w9, gs[0], gs[1], xk, xn, wk, wv, gs[2], gs[3], gs[4], gs[5], k, 1e-5),
                   4 * xn.numel() * es)
            else:
                ab(f"K9 kv_prep(preln) s{si}", lambda: ops.kv_prep(xk, xn, wk, wv, gs[2], gs[3], gs[4], gs[5], k, 1e-5, pre_ln=(gs[6], gs[7], 1e-6, False)),
                   2 * xn.numel() * es)
            del xk
        if not flt or "ln" in flt:
            gam, bet = torch.randn(C, device=DEV, generator=g), torch.randn(C, device=DEV, generator=g)
            ab(f"K8 layernorm s{si} C={C}", lambda: ops.layernorm(x, gam, bet, 1e-6), 2 * x.numel() * es)
        if (not flt or "attn" in flt) and si < 2:
            N = B * T
            qq = x.view(N, H * W, C)
            kk = torch.randn(N, 18, C, device=DEV, generator=g).to(dt)
            vv = torch.randn(N, 18, C, device=DEV, generator=g).to(dt)
            ab(f"K11 attention s{si} Lq={H * W} C={C}", lambda: ops.attention(qq, kk, vv, 2, C ** -0.5), 2 * qq.numel() * es + 4 * kk.numel() * es)
        if (not flt or "commute" in flt) and si < 3:
            # UpEmbed conv1 of the NEXT stage at this stage's resolution: source h x w, Cout = C / 2
            Co = C // 2
            N = B * T
            lib = _lib.load()
            c_ext = torch.randn(N, H + 2, W + 2, Co, device=DEV, generator=g).to(dt)
            tb = torch.randn(N, 2 * W + 2 * H - 4, 9 * Co, device=DEV, generator=g).to(dt)
            sc, sh = torch.randn(Co, device=DEV, generator=g), torch.randn(Co, device=DEV, generator=g)
            out = torch.empty(N, 2 * H, 2 * W, Co, device=DEV, dtype=dt)
            code = ops.DTYPE_CODES[dt]
            st = torch.cuda.current_stream().cuda_stream
            ab(f"K12-tap commute {H}x{W} C={Co}", lambda: _lib.check(lib.diffsal_up2_conv_commute(
                c_ext.data_ptr(), tb.data_ptr(), sc.data_ptr(), sh.data_ptr(), out.data_ptr(), N, H, W, Co, 1, code, st), "c"),
               (c_ext.numel() + out.numel()) * es)
            ab(f"K12-tap   ring only {H}x{W} C={Co}", lambda: _lib.check(lib.diffsal_up2_conv_commute_ring(
                c_ext.data_ptr(), tb.data_ptr(), sc.data_ptr(), sh.data_ptr(), out.data_ptr(), N, H, W, Co, 1, code, st), "c"),
               (c_ext.numel() + out.numel()) * es)
            del c_ext, tb, out
        del x


EXTRA = []

if __name__ == "__main__":
    main()
