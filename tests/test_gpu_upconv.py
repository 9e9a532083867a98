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
