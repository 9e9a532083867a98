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


CASES4 = [
    # N, H, W, Cin, Cout, d, epilogue            F(4x4, 3x3): Cin % 96 == 0
    (2, 14, 24, 192, 192, 1, "bias"),                 # 14 rows: the last tile row is half empty
    (2, 14, 24, 192, 192, 2, "bn_relu_res"),          # sub-grids of 7 x 12
    (1, 7, 9, 96, 68, 2, "bias_rowvec"),              # odd everything, Cout not a multiple of 16
    (3, 5, 6, 96, 100, 1, "none"),
    (4, 28, 48, 192, 384, 1, "bias_res"),             # ResnetBlock of stage 1 at its real size
    (4, 14, 24, 768, 768, 1, "bias_rowvec"),          # ResnetBlock of stage 2: deepest contraction
    (36, 14, 24, 384, 384, 2, "bn_relu_res"),         # UpEmbed-2 of stage 1
]


@pytest.mark.parametrize("case", CASES4, ids=[f"{c[1]}x{c[2]}_{c[3]}to{c[4]}_d{c[5]}_{c[6]}" for c in CASES4])
def test_winograd_f4_conv_matches_direct_convolution(case, tuning):
    """F(4x4, 3x3) (csrc/wino4.hip: input transform, 36 batched position products on gemm_dma_kernel, output transform + epilogue)
    against torch's fp32 convolution and the direct kernel.  Bar 1e-4 of the output maximum (measured ~1e-5: the transforms
    multiply by up to 8 and cancel); DIFFSAL_NO_WINOGRAD4 falls back to F(2x2), DIFFSAL_NO_WINOGRAD to the direct kernel."""
    from diff_sal_amd import ops

    N, H, W, Cin, Cout, d, epi = case
    tuning.set("DIFFSAL_FORCE_WINOGRAD", 1)
    x = rnd("wx", N, H, W, Cin)
    w = rnd("ww", Cout, Cin, 3, 3, scale=0.05)
    bias = rnd("wb", Cout, scale=0.2) if "bias" in epi or "bn" in epi else None
    scale = (rnd("ws", Cout, scale=0.1) + 1.0) if "bn" in epi else None
    shift = rnd("wh", Cout, scale=0.1) if "bn" in epi else None
    rowvec = rnd("wr", N, Cout + 4, scale=0.3)[:, :Cout] if "rowvec" in epi else None
    res = rnd("wq", N, H, W, Cout) if "res" in epi else None
    act = ops.ACT_RELU if "relu" in epi else ops.ACT_NONE
    ref = F.conv2d(x.double().permute(0, 3, 1, 2), w.double(), None if bias is None else bias.double(), padding=d, dilation=d).permute(0, 2, 3, 1)
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
    ww = ops.WinoWeights(wd)
    direct = ops.conv_igemm(x.to(DEV), ops.pack_conv_weight(wd), **kw)
    f4 = ops.conv_igemm(x.to(DEV), ops.pack_conv_weight(wd), wino=ww, **kw)
    tuning.set("DIFFSAL_NO_WINOGRAD4", 1)
    f2 = ops.conv_igemm(x.to(DEV), ops.pack_conv_weight(wd), wino=ww, **kw)
    e4, e2, e0 = rel_err(f4, ref), rel_err(f2, ref), rel_err(direct, ref)
    print(f"F(4x4) {e4:.2e}  F(2x2) {e2:.2e}  direct {e0:.2e}")
    assert e0 < 2e-5 and e2 < 2e-5
    assert e4 < 1e-4
    assert not torch.equal(f4, f2) and not torch.equal(f4, direct), "the F(4x4) path did not run"
    again = None
    tuning.set("DIFFSAL_NO_WINOGRAD4", None)
    again = ops.conv_igemm(x.to(DEV), ops.pack_conv_weight(wd), wino=ww, **kw)
    assert torch.equal(again, f4)                     # deterministic
    tuning.set("DIFFSAL_NO_WINOGRAD", 1)
    off = ops.conv_igemm(x.to(DEV), ops.pack_conv_weight(wd), wino=ww, **kw)
    assert torch.equal(off, direct)


def test_winograd_f4_staged_calls_equal_the_single_call(tuning):
    """diffsal_conv_wino4_stages (input transform / position products / output transform as separate calls: what ops does
    under bench.py's per-launch profiler) produces the bits of diffsal_conv_wino4."""
    from diff_sal_amd import ops

    tuning.set("DIFFSAL_FORCE_WINOGRAD", 1)
    x = rnd("wx", 4, 28, 48, 192).to(DEV)
    w = rnd("ww", 384, 192, 3, 3, scale=0.05).to(DEV)
    b = rnd("wb", 384, scale=0.2).to(DEV)
    res = rnd("wq", 4, 28, 48, 384).to(DEV)
    ww, wp = ops.WinoWeights(w), ops.pack_conv_weight(w)
    kw = dict(kh=3, kw=3, pad=(1, 1), dil=(1, 1), bias=b, residual=res, act=ops.ACT_RELU, wino=ww)
    one = ops.conv_igemm(x, wp, **kw)
    ops.PROFILE = []
    try:
        staged = ops.conv_igemm(x, wp, tag="K4", **kw)
        classes = [e[3] for e in ops.PROFILE]
        kernels = [e[6] if len(e) > 6 else None for e in ops.PROFILE]
    finally:
        ops.PROFILE = None
    assert torch.equal(one, staged)
    assert classes == ["K4-xf", "K4", "K4-xf"], classes
    assert kernels[0] == "wino4_input_kernel" and kernels[2] == "wino4_output_kernel" and "gemm_dma_kernel" in kernels[1] and "batch 36" in kernels[1]
