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


@pytest.mark.parametrize("d", [1, 2])
def test_winograd_f4_on_a_checkpoint_like_dynamic_range(d, tuning):
    """The 1e-4 bar of F(4x4) evidenced where it matters: activations whose magnitude spans 1e-3 .. 1e2 across channels and
    pixels (a trained network's feature maps are not unit Gaussians: the transforms multiply by up to 8 and cancel, so small
    outputs next to large inputs are the hard case) and weights with per-output-channel scales over two decades.  The error is
    measured against the fp64 convolution on the scale of the output maximum, like every per-layer bar, and -- the stricter view --
    per output channel on that channel's own maximum."""
    from diff_sal_amd import ops

    tuning.set("DIFFSAL_FORCE_WINOGRAD", 1)
    N, H, W, Cin, Cout = 4, 28, 48, 192, 192
    g = torch.Generator().manual_seed(17)
    chan = 10.0 ** (torch.rand(Cin, generator=g) * 5 - 3)                       # per-channel magnitudes 1e-3 .. 1e2
    spot = 10.0 ** (torch.rand(N, H, W, 1, generator=g) * 2 - 1.5)              # per-pixel 0.03 .. 3
    x = torch.randn(N, H, W, Cin, generator=g) * chan * spot
    wsc = 10.0 ** (torch.rand(Cout, 1, 1, 1, generator=g) * 2 - 2)              # per-output-channel 1e-2 .. 1
    w = torch.randn(Cout, Cin, 3, 3, generator=g) * wsc / (3 * Cin ** 0.5)
    ref = F.conv2d(x.double().permute(0, 3, 1, 2), w.double(), None, padding=d, dilation=d).permute(0, 2, 3, 1)
    wd = w.to(DEV)
    kw = dict(kh=3, kw=3, pad=(d, d), dil=(d, d))
    f4 = ops.conv_igemm(x.to(DEV), ops.pack_conv_weight(wd), wino=ops.WinoWeights(wd), **kw)
    direct = ops.conv_igemm(x.to(DEV), ops.pack_conv_weight(wd), **kw)
    assert not torch.equal(f4, direct), "the F(4x4) path did not run"
    err = (f4.double().cpu() - ref).abs()
    e_all = err.max().item() / ref.abs().max().item()
    per_ch = (err.amax(dim=(0, 1, 2)) / ref.abs().amax(dim=(0, 1, 2))).max().item()
    e_dir = (direct.double().cpu() - ref).abs().max().item() / ref.abs().max().item()
    print(f"F(4x4) on |x| in 1e-3..1e2: {e_all:.2e} of the output maximum, worst channel {per_ch:.2e} of its own maximum; direct {e_dir:.2e}")
    assert e_all < 1e-4 and per_ch < 1e-4 and e_dir < 2e-5


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


# ---- ResnetBlock extras of the F(4x4) path (diffsal_conv_wino4_ex): GroupNorm + swish on load, the 1x1 shortcut inside the
# launch of the position products, statistics of the result for the next GroupNorm ----
def _gn_ab_ref(x_nhwc, gamma, beta, groups, eps):
    """GroupNorm in affine form from fp64 statistics: ab [N,2,C]."""
    N, H, W, Cc = x_nhwc.shape
    xg = x_nhwc.double().reshape(N, H * W, groups, Cc // groups)
    mean = xg.mean(dim=(1, 3))
    var = xg.var(dim=(1, 3), unbiased=False)
    rstd = (var + eps).rsqrt()
    a = rstd.repeat_interleave(Cc // groups, dim=1) * gamma.double()
    b = beta.double() - mean.repeat_interleave(Cc // groups, dim=1) * a
    return torch.stack([a, b], dim=1).float()


@pytest.mark.parametrize("shape", [(4, 56, 96, 96), (4, 14, 24, 768), (2, 9, 7, 192), (3, 28, 48, 384)])
def test_gn_affine_matches_group_norm(shape):
    from diff_sal_amd import ops

    N, H, W, Cc = shape
    x = rnd("gax", *shape) * 1.7 + 0.4
    gamma, beta = rnd("gag", Cc, scale=0.3) + 1.0, rnd("gab", Cc, scale=0.2)
    ab = ops.gn_affine(x.to(DEV), gamma.to(DEV), beta.to(DEV), 32, 1e-6).cpu()
    ref = _gn_ab_ref(x, gamma, beta, 32, 1e-6)
    assert tuple(ab.shape) == (N, 2, Cc)
    assert (ab - ref).abs().max().item() < 2e-6 * ref.abs().max().item()
    y = x * ab[:, 0][:, None, None, :] + ab[:, 1][:, None, None, :]
    yr = F.group_norm(x.permute(0, 3, 1, 2), 32, gamma, beta, 1e-6).permute(0, 2, 3, 1)
    assert rel_err(y, yr) < 2e-6


RES_CASES = [
    # N, H, W, Cin, Cout          the three ResnetBlocks of the noise encoder at B = 4 and a ragged one
    (4, 56, 96, 96, 192),
    (4, 28, 48, 192, 384),
    (4, 14, 24, 384, 768),        # 14 rows: half-empty last tile row; 1344 shortcut rows = 14 x 96 tiles
    (2, 12, 20, 96, 96),          # no shortcut (Cin == Cout); statistics over few workgroups
]


@pytest.mark.parametrize("case", RES_CASES, ids=[f"{c[1]}x{c[2]}_{c[3]}to{c[4]}" for c in RES_CASES])
def test_resblock_extras_of_the_f4_path(case, tuning):
    """h = conv1(swish(GN(x))) + temb (+ norm2 statistics, + the shortcut product in the same launch); out = conv2(swish(GN(h))) + sc."""
    from diff_sal_amd import ops

    N, H, W, Cin, Cout = case
    tuning.set("DIFFSAL_FORCE_WINOGRAD", 1)
    x = rnd("rx", N, H, W, Cin) * 1.3 + 0.2
    w1, w2 = rnd("rw1", Cout, Cin, 3, 3, scale=0.05), rnd("rw2", Cout, Cout, 3, 3, scale=0.04)
    b1, b2 = rnd("rb1", Cout, scale=0.2), rnd("rb2", Cout, scale=0.2)
    g1, be1 = rnd("rg1", Cin, scale=0.3) + 1.0, rnd("re1", Cin, scale=0.2)
    g2, be2 = rnd("rg2", Cout, scale=0.3) + 1.0, rnd("re2", Cout, scale=0.2)
    temb = rnd("rt", N, Cout + 8, scale=0.3)[:, 4:4 + Cout]
    wn = rnd("rwn", Cout, Cin, scale=0.1) if Cin != Cout else None
    # fp64 reference of the block (R/models/saliency_decoder/sal_unet.py:123-142, eval mode)
    xd = x.double().permute(0, 3, 1, 2)
    h_ref = F.conv2d(F.silu(F.group_norm(xd, 32, g1.double(), be1.double(), 1e-6)), w1.double(), b1.double(), padding=1) + \
        temb.double()[:, :, None, None]
    sc_ref = xd if wn is None else F.conv2d(xd, wn.double()[:, :, None, None])
    out_ref = F.conv2d(F.silu(F.group_norm(h_ref, 32, g2.double(), be2.double(), 1e-6)), w2.double(), b2.double(), padding=1) + sc_ref
    h_ref, sc_ref, out_ref = (t.permute(0, 2, 3, 1).float() for t in (h_ref, sc_ref, out_ref))

    xg = x.to(DEV)
    plan = ops.resblock_wino4_plan(xg, Cout, 32)
    assert plan is not None and plan["stats"]
    assert plan["side"] == ((N * H * W) % (N * ((H + 3) // 4) * ((W + 3) // 4)) == 0)
    u1, u2 = ops.pack_wino4_weight(w1.to(DEV)), ops.pack_wino4_weight(w2.to(DEV))
    ab1 = ops.gn_affine(xg, g1.to(DEV), be1.to(DEV), 32, 1e-6)
    side = (xg, wn.to(DEV)) if wn is not None and plan["side"] else None
    h, sc, st = ops.conv3x3_wino4_ex(xg, u1, bias=b1.to(DEV), rowvec=temb.to(DEV), gn_ab=ab1, side=side, stats_groups=32)
    assert rel_err(h, h_ref) < 5e-5
    if side is not None:
        assert rel_err(sc, sc_ref) < 2e-6
        # the shortcut product is bit-equal to the stand-alone plain product on the same kernel
        tuning.set("DIFFSAL_GEMM_DMA", 1)
        alone = ops.linear(xg.view(N * H * W, Cin), wn.to(DEV), None)
        assert torch.equal(sc.view(N * H * W, Cout), alone)
    ab2 = ops.gn_affine_from_stats(st, g2.to(DEV), be2.to(DEV), 1e-6)
    ab2_pass = ops.gn_affine(h, g2.to(DEV), be2.to(DEV), 32, 1e-6)               # the statistics pass over the same tensor
    assert (ab2 - ab2_pass).abs().max().item() < 2e-6 * ab2_pass.abs().max().item()
    assert (ab2.cpu() - _gn_ab_ref(h_ref, g2, be2, 32, 1e-6)).abs().max().item() < 1e-4 * ab2_pass.abs().max().item()
    resid = sc if sc is not None else (xg if wn is None else ops.linear(xg.view(N * H * W, Cin), wn.to(DEV), None).view(N, H, W, Cout))
    out, _, _ = ops.conv3x3_wino4_ex(h, u2, bias=b2.to(DEV), residual=resid, gn_ab=ab2)
    assert rel_err(out, out_ref) < 1e-4
    # against the per-operator launches (GroupNorm kernel, plain F(4x4) convolutions): same arithmetic up to rounding order
    hn = ops.groupnorm_swish(xg, g1.to(DEV), be1.to(DEV), 32, 1e-6)
    h_un = ops.conv_igemm(hn, ops.pack_conv_weight(w1.to(DEV)), kh=3, kw=3, pad=(1, 1), bias=b1.to(DEV), rowvec=temb.to(DEV),
                          wino=ops.WinoWeights(w1.to(DEV)))
    assert rel_err(h, h_un.cpu()) < 2e-5


@pytest.mark.parametrize("case", [(4, 28, 48, 192, 384), (36, 14, 24, 384, 384), (4, 14, 24, 384, 768)], ids=str)
def test_batched_position_products_xcd_order_is_bit_identical(case, tuning):
    """The XCD-aware unit order of the batched launch (an XCD owns whole (position, M tile) rows) changes which workgroup computes a
    unit, not the unit's arithmetic: results equal the plain order bit for bit, with and without the appended shortcut product."""
    from diff_sal_amd import ops

    N, H, W, Cin, Cout = case
    tuning.set("DIFFSAL_FORCE_WINOGRAD", 1)
    x = rnd("xx", N, H, W, Cin).to(DEV)
    u = ops.pack_wino4_weight(rnd("xw", Cout, Cin, 3, 3, scale=0.05).to(DEV))
    wn = rnd("xn", Cout, Cin, scale=0.1).to(DEV)
    plan = ops.resblock_wino4_plan(x, Cout, 32)
    side = (x, wn) if plan is not None and plan["side"] else None
    tuning.set("DIFFSAL_BATCH_XCD", 1)              # off by default (measured slower): the order is still kept correct
    a, sa, _ = ops.conv3x3_wino4_ex(x, u, side=side)
    tuning.set("DIFFSAL_BATCH_XCD", None)
    b, sb, _ = ops.conv3x3_wino4_ex(x, u, side=side)
    assert torch.equal(a, b)
    if side is not None:
        assert torch.equal(sa, sb)
