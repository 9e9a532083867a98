"""Winograd F(2x2, 3x3) form of the fp32 3x3 stride-1 convolutions (csrc/wino.hip) against a plain fp32 PyTorch convolution and
against the direct implicit-GEMM kernel it replaces: dilation 1 and 2 (polyphase sub-grids), odd sub-grid sizes, partial tile
and channel blocks, every epilogue term, the split over input channels.  Tolerance 2e-5 of the output maximum (the transforms
add ~1e-6 relative rounding to the exact-fp32 products)."""
import pytest
import torch
import torch.nn.functional as F

from oracle import salunet_oracle as orc

pytestmark = pytest.mark.gpu

DEV = "cuda"


def rnd(name, *shape, scale=1.0):
    return orc.synth_tensor(name, shape, scale)


def rel_err(got, ref):
    ref = ref.float().cpu()
    return (got.float().cpu() - ref).abs().max().item() / (ref.abs().max().item() + 1e-12)


CASES = [
    # N, H, W, Cin, Cout, d, epilogue
    (2, 14, 24, 192, 192, 1, "bias"),
    (2, 14, 24, 192, 192, 2, "bn_relu_res"),          # sub-grids of 7 x 12: a half-empty tile row
    (1, 7, 9, 64, 68, 2, "bias_rowvec"),              # odd everything, Cout past one block of 64
    (3, 5, 6, 32, 100, 1, "none"),
    (1, 6, 8, 768, 192, 1, "bias_rowvec"),            # few tiles, deep: the input channels are split over workgroups
    (4, 28, 48, 192, 384, 1, "bias_res"),             # ResnetBlock of stage 1 at its real size
    (36, 14, 24, 384, 384, 2, "bn_relu_res"),         # UpEmbed-2 of stage 1: 324 blocks = 252 whole + 72 cut into 3 pieces (tail mode)
]


@pytest.mark.parametrize("case", CASES, ids=[f"{c[1]}x{c[2]}_{c[3]}to{c[4]}_d{c[5]}_{c[6]}" for c in CASES])
def test_winograd_conv_matches_direct_convolution(case, tuning):
    from diff_sal_amd import ops

    N, H, W, Cin, Cout, d, epi = case
    tuning.set("DIFFSAL_FORCE_WINOGRAD", 1)           # also the shapes the planner would leave to the direct kernel
    x = rnd("wx", N, H, W, Cin)
    w = rnd("ww", Cout, Cin, 3, 3, scale=0.05)
    bias = rnd("wb", Cout, scale=0.2) if "bias" in epi or "bn" in epi else None
    scale = (rnd("ws", Cout, scale=0.1) + 1.0) if "bn" in epi else None
    shift = rnd("wh", Cout, scale=0.1) if "bn" in epi else None
    rowvec = rnd("wr", N, Cout + 4, scale=0.3)[:, :Cout] if "rowvec" in epi else None
    res = rnd("wq", N, H, W, Cout) if "res" in epi else None
    act = ops.ACT_RELU if "relu" in epi else ops.ACT_NONE
    ref = F.conv2d(x.permute(0, 3, 1, 2), w, bias, padding=d, dilation=d).permute(0, 2, 3, 1)
    if scale is not None:
        ref = ref * scale + shift
    if rowvec is not None:
        ref = ref + rowvec[:, None, None, :]
    if act == ops.ACT_RELU:
        ref = ref.relu()
    if res is not None:
        ref = ref + res
    dv = lambda t: None if t is None else t.to(DEV)
    wd = w.to(DEV)
    kw = dict(kh=3, kw=3, pad=(d, d), dil=(d, d), bias=dv(bias), scale=dv(scale), shift=dv(shift), rowvec=dv(rowvec),
              residual=dv(res), act=act)
    direct = ops.conv_igemm(x.to(DEV), ops.pack_conv_weight(wd), **kw)
    wino = ops.conv_igemm(x.to(DEV), ops.pack_conv_weight(wd), wino=ops.pack_wino_weight(wd), **kw)
    assert rel_err(direct, ref) < 2e-5
    assert rel_err(wino, ref) < 2e-5
    assert not torch.equal(wino, direct), "the Winograd path did not run (results are bit-equal to the direct kernel)"
    tuning.set("DIFFSAL_NO_WINOGRAD", 1)              # the switch wins over everything: the direct kernel, bit for bit
    off = ops.conv_igemm(x.to(DEV), ops.pack_conv_weight(wd), wino=ops.pack_wino_weight(wd), **kw)
    assert torch.equal(off, direct)
