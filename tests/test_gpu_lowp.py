"""Reduced-precision STORAGE datapaths (BASELINE configs[1] "bf16", configs[4] "fp16"; SURVEY 8d Config 2 / 5).

bf16 / fp16 activations and packed weights in HBM, native 16-bit MFMA, fp32 accumulation, fp32 statistics.  The reference
itself is fp32-only (no autocast near R/diffusion_trainer.py:212-218), so these modes are opt-in
(``SalUNet(..., compute_dtype=torch.bfloat16)``) and carry their OWN tolerances, written here and in DESIGN.md:

  * per operator: against an fp32 PyTorch evaluation of the SAME 16-bit-rounded inputs; the only differences are the
    accumulation order and ONE rounding of the result to the storage type (bf16: 2^-9 relative, fp16: 2^-11);
  * end to end: against the reference's fp32 golden vectors (tests/golden/*.npz), all six forward cases, the 10-step
    DDIM and the 50-NFE DPM-Solver trajectories: LOWP_ATOL below (absolute, outputs live in (0, 1)).
"""
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import salunet_oracle as orc
from tests._cases import CASES, load_case

pytestmark = pytest.mark.gpu
DEV = "cuda"

DTYPES = {"bf16": torch.bfloat16, "fp16": torch.float16}
# one rounding of an O(1) result: half an ulp is 2^-9 (bf16) / 2^-12 (fp16) relative; 3x margin for the max over a tensor
# measured against |ref|_max, plus accumulation-order noise
OP_RTOL = {"bf16": 6e-3, "fp16": 8e-4}
# end-to-end absolute tolerance on the (0,1) saliency map vs the fp32 reference fixtures (measured values are printed
# by the tests and tabulated in DESIGN.md section 2b)
LOWP_ATOL = {"bf16": 3e-2, "fp16": 4e-3}


def rnd(name, *shape, scale=1.0):
    return orc.synth_tensor(name, shape, scale)


def nhwc(x):
    return x.permute(0, 2, 3, 1).contiguous()


def q(x, dt):
    """round to the storage type and back: the exact values the kernel sees"""
    return x.to(dt).float()


def rel_err(got, ref):
    ref = ref.float().cpu()
    return (got.float().cpu() - ref).abs().max().item() / (ref.abs().max().item() + 1e-12)


@pytest.fixture(scope="module")
def ops():
    from diff_sal_amd import ops as o

    assert torch.cuda.is_available()
    return o


CONV_CASES = [
    # N, H, W, Cin, Cout, k, stride, pad, dil, asym
    (2, 14, 24, 96, 192, 3, 1, 1, 1, False),    # K/32 = 27 sub-slices: odd count -> half-empty last stage
    (2, 14, 24, 192, 192, 3, 2, 0, 1, True),    # Downsample: pad (0,1,0,1), stride 2
    (1, 30, 46, 96, 96, 3, 4, 0, 1, True),      # Downsample4x4
    (3, 14, 24, 384, 192, 3, 1, 2, 2, False),   # UpEmbed dilated conv
    (2, 9, 13, 64, 32, 3, 1, 1, 1, False),      # tiny channel counts / row and column remainders
    (2, 20, 36, 768, 96, 3, 1, 1, 1, False),    # mt_proj family
    (1, 40, 64, 128, 256, 1, 1, 0, 1, False),   # 1x1
    (1, 64, 96, 96, 768, 1, 1, 0, 1, False),    # K = 96: three sub-slices, wide N
    (1, 7, 12, 768, 768, 3, 1, 1, 1, False),    # M = 84: the split-K path of the encoder's coarse convs
    (4, 56, 96, 192, 96, 3, 1, 2, 2, False),    # stage-3 UpEmbed shape: 256-row tiles
]


@pytest.mark.parametrize("dname", list(DTYPES))
@pytest.mark.parametrize("case", CONV_CASES)
def test_conv_igemm_16bit_storage(ops, case, dname):
    dt = DTYPES[dname]
    N, H, W, Cin, Cout, k, s, p, d, asym = case
    x = q(rnd("lx%d%d" % (Cin, Cout), N, Cin, H, W), dt)
    w = q(rnd("lw%d%d" % (Cin, Cout), Cout, Cin, k, k, scale=1.0 / math.sqrt(Cin * k * k)), dt)
    b = rnd("lb", Cout, scale=0.1)
    if asym:
        ref = F.conv2d(F.pad(x, (0, 1, 0, 1)), w, b, stride=s)
        kw = dict(stride=(s, s), pad=(0, 0), out_hw=ref.shape[-2:])
    else:
        ref = F.conv2d(x, w, b, stride=s, padding=p, dilation=d)
        kw = dict(stride=(s, s), pad=(p, p), dil=(d, d))
    wp = ops.cast(ops.pack_conv_weight(w.to(DEV)), dt)
    got = ops.conv_igemm(nhwc(x).to(DEV).to(dt), wp, kh=k, kw=k, bias=b.to(DEV), **kw)
    assert got.dtype == dt and got.shape == nhwc(ref).shape
    assert rel_err(got, nhwc(ref)) < OP_RTOL[dname]


@pytest.mark.parametrize("dname", list(DTYPES))
@pytest.mark.parametrize("cfg", [0, 1, 2, 3, 4, 5, 6, 7])
def test_conv_igemm_16bit_every_tile_shape(ops, dname, cfg, tuning):
    """Each of the eight tile configurations of the 16-bit kernel on a shape with row / column / K remainders."""
    tuning.set("DIFFSAL_IGEMM16_CFG", cfg)
    dt = DTYPES[dname]
    N, H, W, Cin, Cout = 2, 19, 27, 96, 224
    x = q(rnd("tx", N, Cin, H, W), dt)
    w = q(rnd("tw", Cout, Cin, 3, 3, scale=0.04), dt)
    ref = F.conv2d(x, w, None, padding=1)
    got = ops.conv_igemm(nhwc(x).to(DEV).to(dt), ops.cast(ops.pack_conv_weight(w.to(DEV)), dt), kh=3, kw=3, pad=(1, 1))
    assert rel_err(got, nhwc(ref)) < OP_RTOL[dname]


@pytest.mark.parametrize("dname", list(DTYPES))
def test_conv_igemm_16bit_epilogue(ops, dname):
    """bias, BN affine, per-image vector, ReLU and residual are applied to the fp32 accumulator; one rounding at the end."""
    dt = DTYPES[dname]
    N, H, W, Cin, Cout = 3, 12, 20, 96, 192
    x = q(rnd("ex", N, Cin, H, W), dt)
    w = q(rnd("ew", Cout, Cin, 3, 3, scale=0.03), dt)
    bias, scale, shift = rnd("eb", Cout, scale=0.1), rnd("es", Cout, scale=0.2) + 1.0, rnd("eh", Cout, scale=0.1)
    rowvec = rnd("er", N, Cout, scale=0.5)
    res = q(rnd("eres", N, Cout, H, W), dt)
    ref = F.relu((F.conv2d(x, w, bias, padding=1)) * scale[None, :, None, None] + shift[None, :, None, None]
                 + rowvec[:, :, None, None]) + res
    got = ops.conv_igemm(nhwc(x).to(DEV).to(dt), ops.cast(ops.pack_conv_weight(w.to(DEV)), dt), kh=3, kw=3, pad=(1, 1),
                         bias=bias.to(DEV), scale=scale.to(DEV), shift=shift.to(DEV), rowvec=rowvec.to(DEV),
                         residual=nhwc(res).to(DEV).to(dt), act=1)
    assert rel_err(got, nhwc(ref)) < OP_RTOL[dname]
    # GELU epilogue of the MLP (token GEMM)
    xt = q(rnd("gx", 500, 96), dt)
    wl = q(rnd("gw", 192, 96, scale=0.1), dt)
    ref2 = F.gelu(xt @ wl.t() + rnd("gb", 192, scale=0.1))
    got2 = ops.linear(xt.to(DEV).to(dt), wl.to(DEV).to(dt), rnd("gb", 192, scale=0.1).to(DEV), act=2)
    assert rel_err(got2, ref2) < OP_RTOL[dname]


@pytest.mark.parametrize("dname", list(DTYPES))
def test_norm_family_16bit(ops, dname):
    dt = DTYPES[dname]
    tol = OP_RTOL[dname]
    # LayerNorm
    for C in (96, 192, 384, 768):
        x = q(rnd("lnx%d" % C, 300, C) * 2 + 0.3, dt)
        g, b = rnd("lng", C, scale=0.1) + 1, rnd("lnb", C, scale=0.1)
        got = ops.layernorm(x.to(DEV).to(dt), g.to(DEV), b.to(DEV), 1e-5)
        assert got.dtype == dt and rel_err(got, F.layer_norm(x, (C,), g, b, 1e-5)) < tol
    # GroupNorm + swish
    x = q(rnd("gnx", 2, 192, 14, 24) * 1.5 + 0.2, dt)
    g, b = rnd("gng", 192, scale=0.1) + 1, rnd("gnb", 192, scale=0.1)
    ref = F.silu(F.group_norm(x, 32, g, b, 1e-6))
    got = ops.groupnorm_swish(nhwc(x).to(DEV).to(dt), g.to(DEV), b.to(DEV), 32, 1e-6)
    assert rel_err(got, nhwc(ref)) < tol
    # depthwise 3x3 + LN (q projection)
    C, H, W = 96, 10, 14
    x = q(rnd("dqx", 3, C, H, W), dt)
    w9 = rnd("dqw", C, 1, 3, 3, scale=0.3)
    ref = F.layer_norm(F.conv2d(x, w9, None, padding=1, groups=C).permute(0, 2, 3, 1).reshape(3, H * W, C), (C,), g[:C], b[:C], 1e-5)
    got = ops.dwconv3_ln(nhwc(x).to(DEV).to(dt), w9.reshape(C, 9).t().contiguous().to(DEV), g[:C].contiguous().to(DEV),
                         b[:C].contiguous().to(DEV), 1e-5)
    assert rel_err(got, ref) < tol
    # pooled k / v projections + LN
    k = 4
    xk, xv = q(rnd("pkx", 2, C, 12, 24), dt), q(rnd("pvx", 2, C, 12, 24), dt)
    wk, wv = rnd("pkw", C, 1, k, k, scale=0.2), rnd("pvw", C, 1, k, k, scale=0.2)
    rk = F.layer_norm(F.conv2d(xk, wk, None, stride=k, groups=C).flatten(2).transpose(1, 2), (C,), g[:C], b[:C], 1e-5)
    rv = F.layer_norm(F.conv2d(xv, wv, None, stride=k, groups=C).flatten(2).transpose(1, 2), (C,), g[:C], b[:C], 1e-5)
    gk, gv = ops.dwpool_ln_kv(nhwc(xk).to(DEV).to(dt), nhwc(xv).to(DEV).to(dt), wk.reshape(C, k * k).t().contiguous().to(DEV),
                              wv.reshape(C, k * k).t().contiguous().to(DEV), g[:C].contiguous().to(DEV),
                              b[:C].contiguous().to(DEV), g[:C].contiguous().to(DEV), b[:C].contiguous().to(DEV), k, 1e-5)
    assert rel_err(gk, rk) < tol and rel_err(gv, rv) < tol


@pytest.mark.parametrize("dname", list(DTYPES))
def test_attention_resize_pack_head_16bit(ops, dname):
    dt = DTYPES[dname]
    tol = OP_RTOL[dname]
    # attention core, 2 heads, Lk = 18
    N, Lq, Lk, C = 3, 84, 18, 192
    qq, kk, vv = (q(rnd(n, N, L, C), dt) for n, L in (("aq", Lq), ("ak", Lk), ("av", Lk)))
    d = C // 2
    qh, kh, vh = (t.view(N, -1, 2, d).transpose(1, 2) for t in (qq, kk, vv))
    ref = (torch.softmax(qh @ kh.transpose(-1, -2) * C ** -0.5, -1) @ vh).transpose(1, 2).reshape(N, Lq, C)
    got = ops.attention(qq.to(DEV).to(dt), kk.to(DEV).to(dt), vv.to(DEV).to(dt), 2, C ** -0.5)
    assert got.dtype == dt and rel_err(got, ref) < tol
    # bilinear x2 and the 4-scale sum
    x = q(rnd("rx", 2, 96, 7, 12), dt)
    ref = F.interpolate(x, scale_factor=2, mode="bilinear", align_corners=False)
    assert rel_err(ops.resize_bilinear(nhwc(x).to(DEV).to(dt), 14, 24), nhwc(ref)) < tol
    xs = [q(rnd("rs%d" % i, 2, 64, 4 * 2 ** i, 8 * 2 ** i), dt) for i in range(4)]
    ref = sum(F.interpolate(t, size=(64, 128), mode="bilinear", align_corners=False) for t in xs)
    got = ops.resize_sum([nhwc(t).to(DEV).to(dt) for t in xs], 64, 128)
    assert rel_err(got, nhwc(ref)) < 2 * tol  # four rounded inputs, one rounded sum
    # frame packing: fp32 NCTHW features + 16-bit noise map -> 16-bit frames
    vis, nz = rnd("pf", 2, 96, 8, 7, 12), q(rnd("pn", 2, 7, 12, 96), dt)
    got = ops.pack_frames(vis.to(DEV), nz.to(DEV).to(dt))
    ref = torch.cat([vis.permute(0, 2, 3, 4, 1), nz[:, None]], 1)
    assert got.dtype == dt and rel_err(got, q(ref, dt)) < 1e-7
    # sigmoid head: 16-bit in, fp32 out
    y = q(rnd("hy", 2, 10, 12, 96), dt)
    hw_, hb = rnd("hw", 96, scale=0.2), rnd("hb", 1, scale=0.1)
    got = ops.head_sigmoid(y.to(DEV).to(dt), hw_.to(DEV), hb.to(DEV))
    assert got.dtype == torch.float32 and rel_err(got, torch.sigmoid(y @ hw_ + hb)[..., None]) < 1e-5
    # conv_in: fp32 in, 16-bit NHWC out
    xi = rnd("ci", 2, 1, 16, 20)
    wi, bi = rnd("ciw", 96, 1, 3, 3, scale=0.3), rnd("cib", 96, scale=0.1)
    got = ops.conv_in(xi.to(DEV), wi.reshape(96, 9).contiguous().to(DEV), bi.to(DEV), 0, out_dtype=dt)
    assert got.dtype == dt and rel_err(got, nhwc(F.conv2d(xi, wi, bi, padding=1))) < tol
    # audio fusion
    B, T, H, W, Cc, h, w = 1, 9, 14, 24, 64, 7, 12
    a_small, xf = q(rnd("afa", B * T, h * w, Cc), dt), q(rnd("afx", B, T, H, W, Cc), dt)
    a_up = a_small.view(B, T, h, w, Cc).repeat_interleave(2, 2).repeat_interleave(2, 3)
    m = torch.softmax((a_up * xf).mean(1), dim=2)                      # [B,H,W,C], softmax over W
    ref = (a_up * m[:, None]).permute(0, 4, 1, 2, 3)                   # [B,C,T,H,W]
    got = ops.audio_fuse(a_small.to(DEV).to(dt), xf.to(DEV).to(dt), h, w)
    assert rel_err(got, ref) < tol


def build(cfg, sd, dt):
    from diff_sal_amd.sal_unet import SalUNet

    net = SalUNet(
        image_based=cfg.image_based, img_size=cfg.img_size, frames_len=1, mid_num_stages=cfg.num_stages,
        temporal_size=9, temporal_list=list(cfg.temporal_list), futr_num_stages=0, ori_embed_dim=cfg.ori_embed_dim,
        down_embed_dim=cfg.down_embed_dim, idx_to_planes={0: cfg.down_embed_dim, 1: 192, 2: 384, 3: cfg.ori_embed_dim},
        patch_size=[0, 3, 3, 3], patch_stride=[0, 1, 1, 1], patch_padding=list(cfg.dilation),
        up_channel=list(cfg.up_channel), num_heads=list(cfg.num_heads), mlp_ratio=[2.0] * 4,
        drop_path_rate=[0.15] * 4, qkv_bias=[True] * 4, kv_proj_method=["avg"] * 4, kernel_kv=list(cfg.kernel_kv),
        padding_kv=[0] * 4, stride_kv=list(cfg.kernel_kv), q_proj_method=["dw_bn"] * 4, kernel_q=[3] * 4,
        padding_q=[1] * 4, stride_q=[1] * 4, compute_dtype=dt)
    net.load_state_dict(sd, strict=True)
    return net.to(DEV).eval()


@pytest.mark.parametrize("dname", list(DTYPES))
@pytest.mark.parametrize("name", list(CASES))
def test_lowp_forward_vs_reference_golden(golden_dir, name, dname):
    """All six forward fixtures of the fp32 reference through the 16-bit storage datapath (tolerance table row)."""
    cfg, sd, x, t, feats, audio, g = load_case(golden_dir, name)
    net = build(cfg, sd, DTYPES[dname])
    assert all(p.dtype == torch.float32 for p in net.parameters())      # parameters stay fp32; packed copies are 16-bit
    with torch.no_grad():
        out = net(x.to(DEV), t.to(DEV), [f.to(DEV) for f in feats], None if audio is None else audio.to(DEV))
    ref = torch.from_numpy(g["output"])
    assert out.dtype == torch.float32 and out.shape == ref.shape
    err = (out.cpu() - ref).abs()
    print(f"LOWP {dname} {name}: max abs {err.max().item():.3e} mean abs {err.mean().item():.3e} "
          f"(output range {ref.min().item():.3f}..{ref.max().item():.3f})")
    assert err.max().item() < LOWP_ATOL[dname]
    assert err.max().item() > 0.0


@pytest.mark.parametrize("dname", list(DTYPES))
@pytest.mark.parametrize("name", ["small_vis", "full_av_b1"])
def test_lowp_forward_with_tap_convolutions(golden_dir, name, dname):
    """Opt-in on 16-bit storage (SalUNet.tap_conv16): UpEmbed's first convolution / mt_proj as tap GEMM + tapsum gather; same bar."""
    cfg, sd, x, t, feats, audio, g = load_case(golden_dir, name)
    net = build(cfg, sd, DTYPES[dname])
    net.tap_conv16 = ("s1", "s2", "s3", "mt")
    with torch.no_grad():
        out = net(x.to(DEV), t.to(DEV), [f.to(DEV) for f in feats], None if audio is None else audio.to(DEV))
        net.tap_conv16 = ()
        base = net(x.to(DEV), t.to(DEV), [f.to(DEV) for f in feats], None if audio is None else audio.to(DEV))
    ref = torch.from_numpy(g["output"])
    err = (out.cpu() - ref).abs().max().item()
    print(f"LOWP {dname} {name} tap form: max abs {err:.3e} (direct form {(base.cpu() - ref).abs().max().item():.3e})")
    assert err < LOWP_ATOL[dname]
    assert not torch.equal(out, base)        # the switch did change the executed path


@pytest.mark.parametrize("dname", list(DTYPES))
def test_lowp_sampling_trajectories(golden_dir, dname):
    """10-step DDIM of the reference trainer and the 50-NFE DPM-Solver of the reference sampler with a 16-bit denoiser:
    the sampler state x stays fp32, only the network evaluation is reduced precision."""
    from diff_sal_amd.sampling import DiffusionSampler

    cfg = CASES["tiny_av"][0]
    sd = orc.synth_state_dict(orc.state_dict_template(cfg))
    net = build(cfg, sd, DTYPES[dname])

    class Top(torch.nn.Module):
        def __init__(self, n):
            super().__init__()
            self.decoder_net, self.audio_net, self.visual_net = n, None, None

    top = Top(net)
    for fix, tag, mk in (("ddim_tiny_av", "ddim", lambda: DiffusionSampler(top, timesteps=10, sample_type="ddim")),
                         ("dpm50_tiny_av", "dpm50", lambda: DiffusionSampler(top, timesteps=50, sample_type="dpmsolver"))):
        g = np.load(f"{golden_dir}/{fix}.npz")
        x, feats, audio = orc.synth_inputs(cfg, 1, True, tag="ddim")     # both fixtures start from the "ddim" inputs
        s = mk()
        fd, ad = [f.to(DEV) for f in feats], audio.to(DEV)
        out = s.sample_ddim(x.to(DEV), fd, ad) if tag == "ddim" else s.sample_dpm_solver(x.to(DEV), fd, ad)
        ref = torch.from_numpy(g["output"])
        err = (out.cpu() - ref).abs()
        print(f"LOWP {dname} {fix}: max abs {err.max().item():.3e} mean abs {err.mean().item():.3e}")
        assert err.max().item() < LOWP_ATOL[dname]


def test_lowp_training_is_refused():
    cfg = CASES["tiny_vis"][0]
    net = build(cfg, orc.synth_state_dict(orc.state_dict_template(cfg)), torch.bfloat16).train()
    x, feats, _ = orc.synth_inputs(cfg, 1, False)
    with pytest.raises(RuntimeError, match="inference option"):
        net(x.to(DEV), torch.tensor([1], device=DEV), [f.to(DEV) for f in feats])


@pytest.mark.parametrize("dname", list(DTYPES))
@pytest.mark.parametrize("M,with_z", [(24001, True), (77, False)])
def test_block16_fused_kernel(ops, dname, M, with_z):
    """proj + residual + norm2 + fc1 + GELU + fc2 + residual (+ norm_mts) of the C = 96 stage in one launch on 16-bit storage,
    vs the same chain in fp32 torch on the same rounded inputs / weights (the hidden activations are rounded once to the
    storage type inside the kernel, as the unfused path does when it stores them)."""
    dt = DTYPES[dname]
    C, HID = 96, 192
    o, x = q(rnd("b16o", M, C), dt), q(rnd("b16x", M, C) * 1.2, dt)
    wp, bp = q(rnd("b16wp", C, C, scale=0.1), dt), rnd("b16bp", C, scale=0.1)
    g2, be2 = rnd("b16g", C, scale=0.1) + 1, rnd("b16b", C, scale=0.1)
    w1, b1 = q(rnd("b16w1", HID, C, scale=0.12), dt), rnd("b16b1", HID, scale=0.1)
    w2, b2 = q(rnd("b16w2", C, HID, scale=0.08), dt), rnd("b16b2", C, scale=0.1)
    gz, bz = rnd("b16gz", C, scale=0.1) + 1, rnd("b16bz", C, scale=0.1)
    x1 = x + F.linear(o, wp, bp)
    ref_x2 = x1 + F.linear(F.gelu(F.linear(F.layer_norm(x1, (C,), g2, be2, 1e-5), w1, b1)), w2, b2)
    ref_z = F.layer_norm(ref_x2, (C,), gz, bz, 1e-5)
    d = lambda t: t.to(DEV)
    dq_ = lambda t: t.to(DEV).to(dt)
    hw, T, keep = 7, 9, 5
    x2, z = ops.block16(dq_(o), dq_(x), (dq_(wp), d(bp)), (d(g2), d(be2), 1e-5), (dq_(w1), d(b1)), (dq_(w2), d(b2)),
                        (d(gz), d(bz), 1e-5) if with_z else None, (hw, T, keep))
    assert x2.dtype == dt
    # four or eight wavefronts per workgroup (DIFFSAL_BLOCK16_WAVES; eight from 4096 tiles on): the same arithmetic per token
    from diff_sal_amd import _lib
    kept_rows = ((torch.arange(M, device=DEV) // hw) % T) < keep
    for nw in (4, 8):
        _lib.set_tuning("DIFFSAL_BLOCK16_WAVES", nw)
        try:
            x2w, zw = ops.block16(dq_(o), dq_(x), (dq_(wp), d(bp)), (d(g2), d(be2), 1e-5), (dq_(w1), d(b1)), (dq_(w2), d(b2)),
                                  (d(gz), d(bz), 1e-5) if with_z else None, (hw, T, keep))
        finally:
            _lib.set_tuning("DIFFSAL_BLOCK16_WAVES", None)
        assert torch.equal(x2w, x2) and (not with_z or torch.equal(zw[kept_rows], z[kept_rows]))
    # three chained GEMMs with 16-bit operands: the normalised activations and the hidden layer are rounded to the storage
    # type before they are multiplied (2-3 roundings along the path instead of 1)
    assert rel_err(x2, ref_x2) < 3 * OP_RTOL[dname]
    if with_z:
        kept = ((torch.arange(M) // hw) % T) < keep
        assert rel_err(z.float().cpu()[kept], ref_z[kept]) < 3 * OP_RTOL[dname]


HALO_CASES = [
    # N, H, W, Cin, Cout, dil   (forced: the planner only picks the halo kernel from 80 000 pixels up)
    (3, 13, 19, 64, 80, 2),      # narrow tile, ragged rows / columns / channels (80 = 2.5 MFMA column tiles)
    (2, 16, 32, 96, 96, 1),      # exactly one wide tile per image
    (1, 40, 70, 32, 128, 1),     # wide tile with a column remainder, one channel chunk, 128-wide N tile
    (2, 33, 48, 160, 192, 2),    # two N tiles, five chunks, row remainder of one
]


@pytest.mark.parametrize("dname", list(DTYPES))
@pytest.mark.parametrize("case", HALO_CASES)
def test_halo_conv_kernel_matches_generic_kernel_and_torch(ops, case, dname, tuning):
    """conv16_halo_kernel (LDS patch, weight-row ring) on shapes with every kind of remainder: bit-identical to igemm16_kernel
    (same k order, same fp32 accumulation) and within the datapath tolerance of torch, with the full epilogue."""
    dt = DTYPES[dname]
    N, H, W, Cin, Cout, d = case
    x = q(rnd("hx%d" % Cin, N, Cin, H, W), dt)
    w = q(rnd("hw%d" % Cout, Cout, Cin, 3, 3, scale=1.0 / math.sqrt(9 * Cin)), dt)
    b = rnd("hb", Cout, scale=0.1)
    res = q(rnd("hr", N, Cout, H, W), dt)
    ref = F.relu(F.conv2d(x, w, b, padding=d, dilation=d)) + res
    wp = ops.cast(ops.pack_conv_weight(w.to(DEV)), dt)
    args = dict(kh=3, kw=3, pad=(d, d), dil=(d, d), bias=b.to(DEV), act=ops.ACT_RELU, residual=nhwc(res).to(DEV).to(dt))
    xin = nhwc(x).to(DEV).to(dt)
    tuning.set("DIFFSAL_NO_HALO", 1)
    generic = ops.conv_igemm(xin, wp, **args)
    tuning.set("DIFFSAL_NO_HALO", 0)
    tuning.set("DIFFSAL_FORCE_HALO", 1)
    halo = ops.conv_igemm(xin, wp, **args)
    assert torch.equal(halo, generic)
    assert rel_err(halo, nhwc(ref)) < OP_RTOL[dname]


@pytest.mark.parametrize("dname", list(DTYPES))
@pytest.mark.parametrize("cfg", [None, 0, 1, 2, 3, 4, 5, 6, 7])
def test_persistent_linear_kernel_16bit(ops, dname, cfg, tuning):
    """igemm16_linear_kernel (tiles walked by persistent workgroups, prefetch across tile boundaries) == the one-tile kernel,
    bit for bit, for every tile shape, on row / column remainders, K = 32 (one stage), 96 (half-empty stage) and 160."""
    dt = DTYPES[dname]
    if cfg is not None:
        tuning.set("DIFFSAL_IGEMM16_CFG", cfg)
    # the last two shapes are large launches: there the tiles are walked XCD by XCD (DIFFSAL_NO_XCD_ORDER=1: plain order)
    for M, K, N in ((1000, 160, 72), (130, 96, 100), (4100, 32, 224), (777, 384, 96), (48421, 96, 864), (70001, 64, 136)):
        x = q(rnd("px%d" % K, M, K), dt).to(DEV).to(dt)
        w = q(rnd("pw%d" % N, N, K, scale=1.0 / math.sqrt(K)), dt).to(DEV).to(dt)
        b = rnd("pb", N, scale=0.1).to(DEV)
        r = q(rnd("pr%d" % M, M, N), dt).to(DEV).to(dt)
        tuning.set("DIFFSAL_NO_PERSIST", 1)
        one = ops.linear(x, w, b, residual=r, act=ops.ACT_GELU)
        tuning.set("DIFFSAL_NO_PERSIST", 0)
        per = ops.linear(x, w, b, residual=r, act=ops.ACT_GELU)
        assert torch.equal(one, per), (M, K, N)
        tuning.set("DIFFSAL_NO_XCD_ORDER", 1)
        plain = ops.linear(x, w, b, residual=r, act=ops.ACT_GELU)
        tuning.set("DIFFSAL_NO_XCD_ORDER", 0)
        assert torch.equal(plain, per), (M, K, N)
        ref = F.gelu(x.float() @ w.float().t() + b) + r.float()
        assert rel_err(per, ref) < OP_RTOL[dname]


@pytest.mark.parametrize("dname", list(DTYPES))
@pytest.mark.parametrize("M,K,N", [(648, 768, 768), (648, 96, 96), (650, 384, 384), (70, 64, 36)])
def test_linear_pair_16bit_equals_two_launches(ops, dname, M, K, N, tuning):
    """diffsal_linear_pair on 16-bit storage == the two single launches bit for bit."""
    dt = DTYPES[dname]
    tuning.set("DIFFSAL_NO_PERSIST", 1)
    x0 = q(rnd("q0x%d" % K, M, K), dt).to(DEV).to(dt)
    x1 = q(rnd("q1x%d" % K, M, K), dt).to(DEV).to(dt)
    w0 = q(rnd("q0w%d" % N, N, K, scale=1.0 / math.sqrt(K)), dt).to(DEV).to(dt)
    w1 = q(rnd("q1w%d" % N, N, K, scale=1.0 / math.sqrt(K)), dt).to(DEV).to(dt)
    b0, b1 = rnd("q0b", N, scale=0.1).to(DEV), rnd("q1b", N, scale=0.1).to(DEV)
    y0, y1 = ops.linear_pair(x0, x1, w0, w1, b0, b1)
    assert torch.equal(y0, ops.linear(x0, w0, b0)) and torch.equal(y1, ops.linear(x1, w1, b1))
    assert rel_err(y1, x1.float() @ w1.float().t() + b1) < OP_RTOL[dname]


@pytest.mark.parametrize("dname", list(DTYPES))
@pytest.mark.parametrize("C,H,W,k", [(96, 16, 32, 16), (768, 7, 12, 2), (192, 28, 48, 8)])
def test_qkv_prep_folded_layernorm_16bit(ops, dname, C, H, W, k):
    """diffsal_qkv_prep with the block's LayerNorm folded into its loads == layernorm launch + plain qkv_prep, bit for bit, on
    16-bit storage (the folded form rounds the normalised value through the storage type like the store + reload does)."""
    dt = DTYPES[dname]
    n = 3
    raw = q(rnd("fx%d" % C, n, H, W, C), dt).to(DEV).to(dt)
    xa = q(rnd("fa%d" % C, n, H, W, C), dt).to(DEV).to(dt)
    w9 = rnd("fw9", 9, C, scale=0.3).to(DEV)
    wk, wv = rnd("fwk", k * k, C, scale=1.0 / k).to(DEV), rnd("fwv", k * k, C, scale=1.0 / k).to(DEV)
    g = [(rnd("fg%d" % i, C, scale=0.2) + 1.0).to(DEV) for i in range(4)]
    b = [rnd("fb%d" % i, C, scale=0.2).to(DEV) for i in range(4)]
    from diff_sal_amd import _lib

    # the folded QUERY branch exists in the 4-channel-per-lane form only (SalUNet.fold_norm1, off by default); the 16-byte form of
    # the unfolded launch (csrc/norm.hip::qkv_prep16_kernel) adds in another order and has its own tests (tests/test_gpu_stream16.py)
    _lib.set_tuning("DIFFSAL_NO_STREAM16", 1)
    try:
        normed = ops.layernorm(raw, g[3], b[3], 1e-5)
        for ln_k in (True, False):
            ref = ops.qkv_prep(normed, w9, g[0], b[0], normed if ln_k else xa, normed, wk, wv, g[1], b[1], g[2], b[2], k)
            got = ops.qkv_prep(raw, w9, g[0], b[0], raw if ln_k else xa, raw, wk, wv, g[1], b[1], g[2], b[2], k,
                               pre_ln=(g[3], b[3], 1e-5, ln_k))
            assert all(torch.equal(r_, g_) for r_, g_ in zip(ref, got)), (C, ln_k)
    finally:
        _lib.set_tuning("DIFFSAL_NO_STREAM16", None)
