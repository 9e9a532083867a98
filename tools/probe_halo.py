#!/usr/bin/env python3
"""LDS-halo 3x3 kernel (csrc/conv16_halo.hip) vs the generic 16-bit implicit-GEMM kernel on the SalUNet 3x3 shapes:
max abs difference of the two outputs (same arithmetic, different summation order) and both timings.  GPU only.
usage: tools/probe_halo.py [bf16|fp16]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diff_sal_amd import _lib, ops  # noqa: E402
from tools.tune_igemm16 import timed  # noqa: E402

SHAPES = [  # name, N, H, W, Cin, Cout, dil
    ("s3.pe1 192->96", 36, 56, 96, 192, 96, 1), ("s3.pe2 96->96 d2", 36, 56, 96, 96, 96, 2),
    ("s2.pe1 384->192", 36, 28, 48, 384, 192, 1), ("s2.pe2 192->192 d2", 36, 28, 48, 192, 192, 2),
    ("s1.pe1 768->384", 36, 14, 24, 768, 384, 1), ("s1.pe2 384->384 d2", 36, 14, 24, 384, 384, 2),
    ("mt_proj 96->96", 4, 112, 192, 96, 96, 1), ("res0.c1 96->192", 4, 56, 96, 96, 192, 1),
    ("res0.c2 192->192", 4, 56, 96, 192, 192, 1), ("res1.c1 192->384", 4, 28, 48, 192, 384, 1),
    ("res1.c2 384->384", 4, 28, 48, 384, 384, 1), ("res2.c1 384->768", 4, 14, 24, 384, 768, 1),
    ("res2.c2 768->768", 4, 14, 24, 768, 768, 1), ("odd 13x19 64->80", 3, 13, 19, 64, 80, 2),
]


def main():
    dt = {"bf16": torch.bfloat16, "fp16": torch.float16}[sys.argv[1] if len(sys.argv) > 1 else "bf16"]
    for name, N, H, W, Cin, Cout, dil in SHAPES:
        x = torch.relu(torch.randn(N, H, W, Cin, device="cuda")).to(dt)
        w = (torch.randn(Cout, 9 * Cin, device="cuda") * 0.05).to(dt)
        bias = torch.randn(Cout, device="cuda")
        res = torch.randn(N, H, W, Cout, device="cuda").to(dt)
        kw = dict(kh=3, kw=3, stride=(1, 1), pad=(dil, dil), dil=(dil, dil), out_hw=(H, W), bias=bias, residual=res)
        _lib.set_tuning("DIFFSAL_NO_HALO", 1)
        ref = ops.conv_igemm(x, w, **kw)
        t_old = timed(lambda: ops.conv_igemm(x, w, **kw))
        _lib.set_tuning("DIFFSAL_NO_HALO", 0)
        _lib.set_tuning("DIFFSAL_FORCE_HALO", 1)
        out = ops.conv_igemm(x, w, **kw)
        t_new = timed(lambda: ops.conv_igemm(x, w, **kw))
        kp = {k: v for k, v in kw.items() if k not in ("bias", "residual")}
        t_new_plain = timed(lambda: ops.conv_igemm(x, w, **kp))
        _lib.set_tuning("DIFFSAL_NO_HALO", 1)
        t_old_plain = timed(lambda: ops.conv_igemm(x, w, **kp))
        err = (out.float() - ref.float()).abs().max().item()
        flops = 2.0 * N * H * W * Cout * 9 * Cin
        print(f"{name:20s} M={N * H * W:7d} old {t_old:7.1f} us {flops / t_old / 1e6:6.0f} TF/s | halo {t_new:7.1f} us "
              f"{flops / t_new / 1e6:6.0f} TF/s  x{t_old / t_new:4.2f} | plain old {t_old_plain:7.1f} halo {t_new_plain:7.1f} | max|diff| {err:.3e} (ref max {ref.float().abs().max().item():.2f})",
              flush=True)


if __name__ == "__main__":
    main()
