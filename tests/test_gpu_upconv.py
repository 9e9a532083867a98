"""UpEmbed's first convolution -- bilinear x2 (align_corners=False) -> 3x3, dilation 2, padding 2 -> BatchNorm -> ReLU
(R/models/saliency_decoder/common_block.py:196-206) -- computed at the SOURCE resolution: conv3x3 on the grid extended by one pixel
(F(4x4) Winograd, extended-grid form), unclamped interpolation, corrections on the 3-pixel border ring from the tap products of
the border lines (csrc/upconv.hip).  Against torch in fp64 and against the tap path it replaces."""
import pytest
import torch
import torch.nn.functional as F

from oracle import salunet_oracle as orc

pytestmark = pytest.mark.gpu

DEV = "cuda"


def rnd(name, *shape, scale=1.0):
    return orc.synth_tensor(name, shape, scale)


CASES = [
    # N, h, w, Cin, Cout, force Winograd
    (1, 2, 2, 96, 8, True),            # every output is border ring, both rings overlap
    (2, 3, 5, 96, 12, True),
    (2, 7, 9, 96, 68, True),           # odd sizes, Cout not a multiple of 16
    (3, 13, 12, 192, 96, False),       # planner's choice for the extended-grid convolution
    (36, 14, 24, 384, 192, False),     # stage 2 at B = 4
    (36, 28, 48, 192, 96, False),      # stage 3 at B = 4
]


@pytest.mark.parametrize("case", CASES, ids=[f"{c[1]}x{c[2]}_{c[3]}to{c[4]}" for c in CASES])
def test_up2_conv_commute_matches_interpolate_then_conv(case, tuning):
    from diff_sal_amd import ops

    N, h, w, Cin, Cout, force = case
    if force:
        tuning.set("DIFFSAL_FORCE_WINOGRAD", 1)
    z = rnd("uz", N, h, w, Cin)
    wt = rnd("uw", Cout, Cin, 3, 3, scale=0.05)
    scale = rnd("us", Cout, scale=0.1) + 1.0
    shift = rnd("uh", Cout, scale=0.1)
    up = F.interpolate(z.double().permute(0, 3, 1, 2), scale_factor=2, mode="bilinear", align_corners=False)
    ref = F.conv2d(up, wt.double(), None, padding=2, dilation=2).permute(0, 2, 3, 1) * scale + shift
    ref = ref.relu()
    wd = wt.to(DEV)
    tapw = wd.permute(2, 3, 0, 1).reshape(9 * Cout, Cin).contiguous()
    got = ops.up2_conv3x3_d2(z.to(DEV), ops.pack_conv_weight(wd), ops.WinoWeights(wd), tapw, scale=scale.to(DEV), shift=shift.to(DEV),
                             act=ops.ACT_RELU)
    err = (got.double().cpu() - ref).abs().max().item() / ref.abs().max().item()
    # the tap path it replaces
    y9 = ops.linear(z.to(DEV), tapw, None)
    tap = ops.tapsum([y9], 2 * h, 2 * w, Cout, dil=2, scale=scale.to(DEV), shift=shift.to(DEV), act=ops.ACT_RELU)
    err_tap = (tap.double().cpu() - ref).abs().max().item() / ref.abs().max().item()
    print(f"commuted {err:.2e}  tap path {err_tap:.2e}")
    assert err_tap < 2e-5
    assert err < 1e-4                  # F(4x4) transform rounding on the source-resolution convolution
    # without Winograd (direct kernel on the extended grid): same construction, exact-fp32 products
    tuning.set("DIFFSAL_FORCE_WINOGRAD", None)
    tuning.set("DIFFSAL_NO_WINOGRAD", 1)
    got_d = ops.up2_conv3x3_d2(z.to(DEV), ops.pack_conv_weight(wd), ops.WinoWeights(wd), tapw, scale=scale.to(DEV), shift=shift.to(DEV),
                               act=ops.ACT_RELU)
    err_d = (got_d.double().cpu() - ref).abs().max().item() / ref.abs().max().item()
    print(f"commuted, direct kernel {err_d:.2e}")
    assert err_d < 2e-5


@pytest.mark.parametrize("dname", ["bf16", "fp16"])
@pytest.mark.parametrize("shape", [(2, 7, 12, 768, 384), (4, 14, 24, 384, 192), (4, 28, 48, 192, 96)], ids=["s1", "s2", "s3"])
def test_up2_conv_commute_16bit_storage(dname, shape, tuning):
    """The same construction on bf16 / fp16 storage (direct 16-bit convolution on the extended grid, tap products and the result in
    the storage type, fp32 arithmetic in the interpolation / ring kernels) against fp64 on the ROUNDED inputs, next to the path it
    replaces there (up-sample, then the dilated convolution on the up-sampled map).  Bar: 3x the one-rounding operator bar of
    tests/test_gpu_lowp.py (c, the tap products and the result are each rounded once)."""
    from diff_sal_amd import ops

    dt = {"bf16": torch.bfloat16, "fp16": torch.float16}[dname]
    bar = 3 * {"bf16": 6e-3, "fp16": 8e-4}[dname]
    N, h, w, Cin, Cout = shape
    z = rnd("uz", N, h, w, Cin).to(dt)
    wt = rnd("uw", Cout, Cin, 3, 3, scale=0.05).to(dt)
    scale = rnd("us", Cout, scale=0.1) + 1.0
    shift = rnd("uh", Cout, scale=0.1)
    up = F.interpolate(z.double().permute(0, 3, 1, 2), scale_factor=2, mode="bilinear", align_corners=False)
    ref = (F.conv2d(up, wt.double(), None, padding=2, dilation=2).permute(0, 2, 3, 1) * scale + shift).relu()
    wd = wt.to(DEV)
    wp = ops.cast(ops.pack_conv_weight(wd.float()), dt)
    tapw = wd.permute(2, 3, 0, 1).reshape(9 * Cout, Cin).contiguous()
    got = ops.up2_conv3x3_d2(z.to(DEV), wp, None, tapw, scale=scale.to(DEV), shift=shift.to(DEV), act=ops.ACT_RELU)
    assert got.dtype == dt
    err = (got.double().cpu() - ref).abs().max().item() / ref.abs().max().item()
    u = ops.resize_bilinear(z.to(DEV), 2 * h, 2 * w)
    old = ops.conv_igemm(u, wp, kh=3, kw=3, pad=(2, 2), dil=(2, 2), scale=scale.to(DEV), shift=shift.to(DEV), act=ops.ACT_RELU)
    err_old = (old.double().cpu() - ref).abs().max().item() / ref.abs().max().item()
    print(f"{dname}: commuted {err:.2e}  up-sample + convolution {err_old:.2e}  (bar {bar:.1e})")
    assert err < bar


UP_PAIR = [
    # N, h, w, Cin, C      UpEmbed = conv1 (Cin -> C, at the source resolution) then conv2 (C -> C, dilation 2) + skip
    (2, 2, 3, 96, 96),          # every pixel of conv1's output is border ring
    (3, 5, 7, 192, 96),         # odd source sizes: both parities of patch rows / columns start on the ring
    (36, 14, 24, 384, 192),     # stage 2 at B = 4
    (36, 28, 48, 192, 96),      # stage 3 at B = 4
]


@pytest.mark.parametrize("case", UP_PAIR, ids=[f"{c[1]}x{c[2]}_{c[3]}to{c[4]}" for c in UP_PAIR])
def test_second_upembed_convolution_reads_the_source_resolution_result(case, tuning):
    """conv2's input transform forms act(BN(interpolation of c_ext)) itself and reads only the border ring of conv1's output
    (ops.conv3x3_wino4_ex(up2=...) after ops.up2_conv3x3_d2(ring_only=True)): the same arithmetic as the two-step form up to FMA
    contraction, and within F(4x4) rounding of fp64 torch."""
    from diff_sal_amd import ops

    N, h, w, Cin, Cc = case
    tuning.set("DIFFSAL_FORCE_WINOGRAD", 1)
    z = rnd("pz", N, h, w, Cin)
    w1, w2 = rnd("pw1", Cc, Cin, 3, 3, scale=0.05), rnd("pw2", Cc, Cc, 3, 3, scale=0.05)
    s1, h1 = rnd("ps1", Cc, scale=0.1) + 1.0, rnd("ph1", Cc, scale=0.1)
    s2, h2 = rnd("ps2", Cc, scale=0.1) + 1.0, rnd("ph2", Cc, scale=0.1)
    skip = rnd("pk", N, 2 * h, 2 * w, Cc)
    up = F.interpolate(z.double().permute(0, 3, 1, 2), scale_factor=2, mode="bilinear", align_corners=False)
    u1 = (F.conv2d(up, w1.double(), None, padding=2, dilation=2) * s1.double()[None, :, None, None] + h1.double()[None, :, None, None]).relu()
    ref = (F.conv2d(u1, w2.double(), None, padding=2, dilation=2) * s2.double()[None, :, None, None] + h2.double()[None, :, None, None]).relu()
    ref = ref.permute(0, 2, 3, 1) + skip.double()
    w1d, w2d = w1.to(DEV), w2.to(DEV)
    tapw = w1d.permute(2, 3, 0, 1).reshape(9 * Cc, Cin).contiguous()
    kw1 = dict(scale=s1.to(DEV), shift=h1.to(DEV), act=ops.ACT_RELU)
    ww1, u2 = ops.WinoWeights(w1d), ops.pack_wino4_weight(w2d)
    # two-step form
    a1 = ops.up2_conv3x3_d2(z.to(DEV), ops.pack_conv_weight(w1d), ww1, tapw, **kw1)
    a2, _, _ = ops.conv3x3_wino4_ex(a1, u2, scale=s2.to(DEV), shift=h2.to(DEV), act=ops.ACT_RELU, residual=skip.to(DEV), dil=2, tag="K12")
    # fused form: poison the interior of the ring buffer to prove that it is never read
    ring, c_ext = ops.up2_conv3x3_d2(z.to(DEV), ops.pack_conv_weight(w1d), ww1, tapw, ring_only=True, **kw1)
    if h >= 4 and w >= 4:
        ring[:, 3:2 * h - 3, 3:2 * w - 3, :] = float("nan")
    b2, _, _ = ops.conv3x3_wino4_ex(ring, u2, scale=s2.to(DEV), shift=h2.to(DEV), act=ops.ACT_RELU, residual=skip.to(DEV), dil=2,
                                    up2=(c_ext, s1.to(DEV), h1.to(DEV), ops.ACT_RELU), tag="K12")
    assert torch.isfinite(b2).all()
    err = (b2.double().cpu() - ref).abs().max().item() / ref.abs().max().item()
    d = (a2 - b2).abs().max().item() / a2.abs().max().item()
    print(f"fused vs fp64 {err:.2e}; fused vs two-step {d:.2e}")
    assert err < 2e-4
    # same expressions, but hipcc contracts them into FMAs differently in the two kernels: the interpolated values differ in the last
    # bit, which the F(4x4) transforms amplify to ~1e-6 of the result
    assert d < 1e-5
