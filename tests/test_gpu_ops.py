"""Per-kernel parity: every C-ABI entry point against the CPU oracle / a plain fp32 PyTorch reference.

Tolerance: north_star's bar is 1e-3 relative (fp32); kernels are exact-fp32 so we test far tighter
(2e-5 of the tensor's max magnitude unless noted).
"""
import math

import pytest
import torch
import torch.nn.functional as F

from oracle import salunet_oracle as orc

pytestmark = pytest.mark.gpu

DEV = "cuda"


def rel_err(got, ref):
    ref = ref.float().cpu()
    return (got.float().cpu() - ref).abs().max().item() / (ref.abs().max().item() + 1e-12)


def rnd(name, *shape, scale=1.0):
    return orc.synth_tensor(name, shape, scale)


def nhwc(x):
    return x.permute(0, 2, 3, 1).contiguous()


def pack_conv(w):
    from diff_sal_amd.ops import pack_conv_weight

    return pack_conv_weight(w.to(DEV)).cpu() if not w.is_cuda else pack_conv_weight(w)


@pytest.fixture(scope="module")
def ops():
    from diff_sal_amd import ops as o

    assert torch.cuda.is_available()
    return o


CONV_CASES = [
    # N, H, W, Cin, Cout, k, stride, pad, dil, asym
    (2, 14, 24, 96, 192, 3, 1, 1, 1, False),    # ResnetBlock conv1 shape family
    (2, 14, 24, 192, 192, 3, 2, 0, 1, True),    # Downsample: pad (0,1,0,1), stride 2
    (1, 30, 46, 96, 96, 3, 4, 0, 1, True),      # Downsample4x4 (odd sizes exercise the bounds)
    (3, 14, 24, 384, 192, 3, 1, 2, 2, False),   # UpEmbed dilated conv
    (2, 9, 13, 64, 32, 3, 1, 1, 1, False),      # tiny channel counts / N remainder
    (2, 20, 36, 768, 96, 3, 1, 1, 1, False),    # mt_proj family (BN=96 tile)
    (1, 40, 64, 128, 256, 1, 1, 0, 1, False),   # 1x1 (nin_shortcut / linear), 128-wide tiles
    (1, 64, 96, 96, 768, 1, 1, 0, 1, False),    # wide-N: 128x192 tile path needs many blocks
]


@pytest.mark.parametrize("case", CONV_CASES)
def test_conv_igemm_matches_conv2d(ops, case):
    N, H, W, Cin, Cout, k, s, p, d, asym = case
    x = rnd("cx%d%d" % (Cin, Cout), N, Cin, H, W)
    w = rnd("cw%d%d" % (Cin, Cout), Cout, Cin, k, k, scale=1.0 / math.sqrt(Cin * k * k))
    b = rnd("cb", Cout, scale=0.1)
    if asym:
        ref = F.conv2d(F.pad(x, (0, 1, 0, 1)), w, b, stride=s)
        kw = dict(stride=(s, s), pad=(0, 0), out_hw=ref.shape[-2:])
    else:
        ref = F.conv2d(x, w, b, stride=s, padding=p, dilation=d)
        kw = dict(stride=(s, s), pad=(p, p), dil=(d, d))
    got = ops.conv_igemm(nhwc(x).to(DEV), pack_conv(w).to(DEV), kh=k, kw=k, bias=b.to(DEV), **kw)
    assert got.shape == nhwc(ref).shape
    assert rel_err(got, nhwc(ref)) < 2e-5


def test_conv_igemm_epilogue_bn_relu_rowvec_residual(ops):
    N, H, W, Cin, Cout = 3, 12, 20, 96, 192
    x = rnd("ex", N, Cin, H, W)
    w = rnd("ew", Cout, Cin, 3, 3, scale=0.03)
    bias, scale, shift = rnd("eb", Cout, scale=0.1), rnd("es", Cout, scale=0.2) + 1.0, rnd("eh", Cout, scale=0.1)
    rowvec = rnd("er", N, Cout + 64)  # a wider table: the kernel must honour the leading dimension
    res = rnd("ee", N, Cout, H, W)
    y = F.conv2d(x, w, bias, padding=1)
    y = y * scale[None, :, None, None] + shift[None, :, None, None] + rowvec[:, 32:32 + Cout, None, None]
    ref = F.relu(y) + res
    got = ops.conv_igemm(nhwc(x).to(DEV), pack_conv(w).to(DEV), kh=3, kw=3, pad=(1, 1), bias=bias.to(DEV),
                         scale=scale.to(DEV), shift=shift.to(DEV), rowvec=rowvec.to(DEV)[:, 32:32 + Cout],
                         residual=nhwc(res).to(DEV), act=ops.ACT_RELU)
    assert rel_err(got, nhwc(ref)) < 2e-5


@pytest.mark.parametrize("M,K,N,act", [(3024, 768, 768, 0), (648, 96, 96, 0), (1000, 192, 384, 2), (12096, 384, 192, 0)])
def test_linear_matches_torch(ops, M, K, N, act):
    x, w, b, r = rnd("lx", 4, M // 4, K), rnd("lw", N, K, scale=K ** -0.5), rnd("lb", N, scale=0.1), rnd("lr", 4, M // 4, N)
    ref = F.linear(x, w, b)
    ref = (F.gelu(ref) if act == 2 else ref) + r
    got = ops.linear(x.to(DEV), w.to(DEV), b.to(DEV), act=act, residual=r.to(DEV))
    assert rel_err(got, ref) < 2e-5


@pytest.mark.parametrize("M,K,N", [(648, 768, 768), (648, 96, 96), (650, 384, 384), (3024, 192, 200), (70, 64, 36)])
def test_linear_pair_equals_two_launches(ops, M, K, N, tuning):
    """diffsal_linear_pair (key and value projections in one grid, z = 2) == the two single launches bit for bit (same tile
    plan, same split-K order), with and without bias, ragged M / N."""
    tuning.set("DIFFSAL_NO_PERSIST", 1)     # the single launches on the same one-tile kernel the pair uses
    tuning.set("DIFFSAL_GEMM_DMA", 0)       # ... and not on the LDS-DMA kernel (same sums, another order)
    x0, x1 = rnd("p0x%d" % K, 2, M // 2, K).to(DEV), rnd("p1x%d" % K, 2, M // 2, K).to(DEV)
    w0, w1 = rnd("p0w%d" % N, N, K, scale=K ** -0.5).to(DEV), rnd("p1w%d" % N, N, K, scale=K ** -0.5).to(DEV)
    b0, b1 = rnd("p0b", N, scale=0.1).to(DEV), rnd("p1b", N, scale=0.1).to(DEV)
    for bias in ((b0, b1), (None, None)):
        y0, y1 = ops.linear_pair(x0, x1, w0, w1, bias[0], bias[1])
        assert torch.equal(y0, ops.linear(x0, w0, bias[0])) and torch.equal(y1, ops.linear(x1, w1, bias[1]))
        assert rel_err(y1, F.linear(x1, w1, bias[1])) < 2e-5


@pytest.mark.parametrize("cfg", [None, 0, 1, 2, 3, 4, 5])
def test_persistent_linear_kernel_fp32(ops, cfg, tuning):
    """igemm_linear_kernel (persistent workgroups, prefetch across tile boundaries, register-direct stores) == the one-tile
    kernel bit for bit, for every tile shape, with row / column remainders, K = 32 (a single slice), 96 and 160."""
    if cfg is not None:
        tuning.set("DIFFSAL_IGEMM_CFG", cfg)
    for M, K, N in ((1000, 160, 72), (130, 96, 100), (4100, 32, 224), (777, 384, 96)):
        x, w, b, r = rnd("px%d" % K, M, K).to(DEV), rnd("pw%d" % N, N, K, scale=K ** -0.5).to(DEV), rnd("pb", N, scale=0.1).to(DEV), rnd("pr", M, N).to(DEV)
        tuning.set("DIFFSAL_NO_PERSIST", 1)
        one = ops.linear(x, w, b, residual=r, act=ops.ACT_GELU)
        tuning.set("DIFFSAL_NO_PERSIST", 0)
        per = ops.linear(x, w, b, residual=r, act=ops.ACT_GELU)
        assert torch.equal(one, per), (M, K, N)
        assert rel_err(per, F.gelu(F.linear(x, w, b)) + r) < 2e-5


@pytest.mark.parametrize("cfg", [None, 0, 1, 2, 3, 4, 5])
def test_persistent_linear_kernel_xcd_tile_order(ops, cfg, tuning):
    """Large launches of igemm_linear_kernel walk the tiles XCD by XCD (one XCD owns whole M-tile rows): every tile still
    computed exactly once -- bit-equal with the plain order and with the one-tile kernel; M-tile counts that are not multiples
    of 8, ragged rows and columns."""
    if cfg is not None:
        tuning.set("DIFFSAL_IGEMM_CFG", cfg)
    for M, K, N in ((48421, 96, 864), (33000, 32, 200), (70001, 64, 136)):
        x, w, b, r = rnd("qx%d" % K, M, K).to(DEV), rnd("qw%d" % N, N, K, scale=K ** -0.5).to(DEV), rnd("qb", N, scale=0.1).to(DEV), rnd("qr%d" % N, M, N).to(DEV)
        tuning.set("DIFFSAL_NO_PERSIST", 1)
        one = ops.linear(x, w, b, residual=r, act=ops.ACT_RELU)
        tuning.set("DIFFSAL_NO_PERSIST", 0)
        tuning.set("DIFFSAL_NO_XCD_ORDER", 1)
        plain = ops.linear(x, w, b, residual=r, act=ops.ACT_RELU)
        tuning.set("DIFFSAL_NO_XCD_ORDER", 0)
        xcd = ops.linear(x, w, b, residual=r, act=ops.ACT_RELU)
        assert torch.equal(one, plain) and torch.equal(one, xcd), (M, K, N)
        assert rel_err(xcd, F.relu(F.linear(x, w, b)) + r) < 2e-5


@pytest.mark.parametrize("K,N,act,res", [(96, 96, 0, True), (96, 96, 0, False), (96, 192, 2, False), (192, 96, 0, True),
                                         (96, 192, 1, True)])
def test_streaming_short_k_linear(ops, K, N, act, res):
    """M >= 65536 rows with K, N in {96, 192} take the barrier-free streaming kernel (lin_stream.hip);
    M is deliberately not a multiple of the 32-row wave tile."""
    M = 70003
    x, w, b = rnd("sx%d" % K, M, K), rnd("sw%d%d" % (K, N), N, K, scale=K ** -0.5), rnd("sb", N, scale=0.1)
    r = rnd("sr", M, N) if res else None
    ref = F.linear(x, w, b)
    ref = F.gelu(ref) if act == 2 else (F.relu(ref) if act == 1 else ref)
    if res:
        ref = ref + r
    got = ops.linear(x.to(DEV), w.to(DEV), b.to(DEV), act=act, residual=None if r is None else r.to(DEV))
    assert rel_err(got, ref) < 2e-5


def test_conv_igemm_reduce_temp_view(ops):
    """Conv3d (5,1,1)/5 over frames == a (5x1)-tap conv on the [B, T, H*W, C] view (quirk Q10)."""
    B, T, H, W, C, Co = 2, 9, 6, 10, 64, 96
    x5 = rnd("rt", B, C, T, H, W)
    w = rnd("rtw", Co, C, 5, 1, 1, scale=0.05)
    ref = F.relu(F.conv3d(x5, w, stride=(5, 1, 1))).squeeze(2)  # [B,Co,H,W]
    xf = x5.permute(0, 2, 3, 4, 1).reshape(B, T, H * W, C).contiguous()
    wp = pack_conv(w)
    got = ops.conv_igemm(xf.to(DEV), wp.to(DEV), kh=5, kw=1, stride=(5, 1), act=ops.ACT_RELU)
    assert got.shape == (B, 1, H * W, Co)
    assert rel_err(got.view(B, H, W, Co), nhwc(ref)) < 2e-5


def test_conv_igemm_rejects_bad_channels(ops):
    with pytest.raises(RuntimeError, match="multiple of 32"):
        ops.conv_igemm(torch.zeros(1, 4, 4, 24, device=DEV), torch.zeros(8, 24, device=DEV))


@pytest.mark.parametrize("tdtype", [torch.int64, torch.float32])
def test_temb_mlp(ops, tdtype):
    sd = orc.synth_state_dict(orc.state_dict_template(orc.SalUNetConfig()))
    t = torch.tensor([0, 3, 999, 500], dtype=tdtype) if tdtype == torch.int64 else torch.tensor([998.996, 0.46, 17.25, 0.0])
    ref = orc.temb_mlp(sd, t, 96)
    freq = torch.exp(torch.arange(48, dtype=torch.float32) * -(math.log(10000) / 47))
    got = ops.temb_mlp(t.to(DEV), freq.to(DEV), *[sd[k].to(DEV) for k in (
        "temb.dense.0.weight", "temb.dense.0.bias", "temb.dense.1.weight", "temb.dense.1.bias")])
    assert rel_err(got, ref) < 2e-5
    w, b = rnd("tpw", 1344, 384, scale=0.05), rnd("tpb", 1344, scale=0.1)
    assert rel_err(ops.dense_small(got, w.to(DEV), b.to(DEV), True), F.linear(orc.swish(ref), w, b)) < 2e-5


@pytest.mark.parametrize("B", [17, 64, 100])
def test_temb_mlp_many_clips_lane_per_clip_form(ops, B):
    """K1 at 17+ clips per call (BASELINE configs[4]: 64 clips per GPU): a lane per clip, one pass over K (csrc/misc.hip::dense_lanes_kernel);
    fractional and integer timesteps, every clip its own."""
    sd = orc.synth_state_dict(orc.state_dict_template(orc.SalUNetConfig()))
    t = (torch.arange(B, dtype=torch.float32) * 9.73 + 0.46) % 1000.0
    ref = orc.temb_mlp(sd, t, 96)
    freq = torch.exp(torch.arange(48, dtype=torch.float32) * -(math.log(10000) / 47))
    got = ops.temb_mlp(t.to(DEV), freq.to(DEV), *[sd[k].to(DEV) for k in (
        "temb.dense.0.weight", "temb.dense.0.bias", "temb.dense.1.weight", "temb.dense.1.bias")])
    assert got.shape == (B, 384) and rel_err(got, ref) < 2e-5
    w, b = rnd("tpw", 1344, 384, scale=0.05), rnd("tpb", 1344, scale=0.1)
    assert rel_err(ops.dense_small(got, w.to(DEV), b.to(DEV), True), F.linear(orc.swish(ref), w, b)) < 2e-5


def test_conv_in_and_skip(ops):
    x, w, b = rnd("cix", 2, 1, 24, 40), rnd("ciw", 96, 1, 3, 3, scale=0.3), rnd("cib", 96, scale=0.1)
    ref = nhwc(F.conv2d(x, w, b, padding=1))
    got = ops.conv_in(x.to(DEV), w.reshape(96, 9).to(DEV), b.to(DEV), 0)
    assert rel_err(got, ref) < 1e-5
    got4 = ops.conv_in(x.to(DEV), w.reshape(96, 9).to(DEV), b.to(DEV), 4).cpu()
    keep = torch.ones(24, 40, dtype=torch.bool)
    keep[3::4, :] = False
    keep[:, 3::4] = False
    assert torch.allclose(got4[:, keep], ref[:, keep], atol=1e-5)


@pytest.mark.parametrize("C,HW", [(96, (14, 24)), (192, (7, 9)), (768, (5, 6)), (384, (28, 48))])
def test_groupnorm_swish(ops, C, HW):
    x = rnd("gn%d" % C, 3, C, *HW) * 2.0 + 0.7
    g, b = rnd("gng", C, scale=0.2) + 1.0, rnd("gnb", C, scale=0.2)
    ref = orc.group_norm_swish(x, g, b)
    got = ops.groupnorm_swish(nhwc(x).to(DEV), g.to(DEV), b.to(DEV), 32, 1e-6)
    assert rel_err(got, nhwc(ref)) < 2e-5


@pytest.mark.parametrize("C", [32, 64, 96, 192, 256, 384, 768])
def test_layernorm(ops, C):
    x = rnd("ln%d" % C, 5, 77, C) * 1.5 + 0.3
    g, b = rnd("lng", C, scale=0.2) + 1.0, rnd("lnb", C, scale=0.2)
    got = ops.layernorm(x.to(DEV), g.to(DEV), b.to(DEV), 1e-5)
    assert rel_err(got, F.layer_norm(x, (C,), g, b, 1e-5)) < 2e-5


@pytest.mark.parametrize("C,H,W,k", [(96, 16, 32, 16), (192, 8, 12, 4), (768, 7, 12, 2), (32, 16, 32, 16), (128, 10, 27, 2), (192, 28, 48, 8)])
def test_depthwise_projections(ops, C, H, W, k):
    n = 3
    xn, xa = rnd("dq%d" % C, n, C, H, W), rnd("da%d" % C, n, C, H, W)
    w3 = rnd("dw3", C, 1, 3, 3, 3, scale=0.3)
    wk, wv = rnd("dwk", C, 1, 1, k, k, scale=1.0 / k), rnd("dwv", C, 1, 1, k, k, scale=1.0 / k)
    g = [rnd("dg%d" % i, C, scale=0.2) + 1.0 for i in range(3)]
    b = [rnd("db%d" % i, C, scale=0.2) for i in range(3)]

    def tok(m):
        return m.flatten(2).transpose(1, 2)

    q_ref = F.layer_norm(tok(F.conv3d(xn.unsqueeze(2), w3, padding=1, groups=C).squeeze(2)), (C,), g[0], b[0], 1e-5)
    k_ref = F.layer_norm(tok(F.conv2d(xa, wk[:, :, 0], stride=k, groups=C)), (C,), g[1], b[1], 1e-5)
    v_ref = F.layer_norm(tok(F.conv2d(xn, wv[:, :, 0], stride=k, groups=C)), (C,), g[2], b[2], 1e-5)
    dv = lambda t: t.to(DEV)  # noqa: E731
    q = ops.dwconv3_ln(dv(nhwc(xn)), dv(w3[:, 0, 1].reshape(C, 9).t().contiguous()), dv(g[0]), dv(b[0]))
    kk, vv = ops.dwpool_ln_kv(dv(nhwc(xa)), dv(nhwc(xn)), dv(wk.reshape(C, k * k).t().contiguous()),
                              dv(wv.reshape(C, k * k).t().contiguous()), dv(g[1]), dv(b[1]), dv(g[2]), dv(b[2]), k)
    assert rel_err(q, q_ref) < 2e-5 and rel_err(kk, k_ref) < 2e-5 and rel_err(vv, v_ref) < 2e-5
    # the merged launch (query branch + pooled key / value branch in one grid) == the two entries, bit for bit
    q2, k2, v2 = ops.qkv_prep(dv(nhwc(xn)), dv(w3[:, 0, 1].reshape(C, 9).t().contiguous()), dv(g[0]), dv(b[0]), dv(nhwc(xa)),
                              dv(nhwc(xn)), dv(wk.reshape(C, k * k).t().contiguous()), dv(wv.reshape(C, k * k).t().contiguous()),
                              dv(g[1]), dv(b[1]), dv(g[2]), dv(b[2]), k)
    assert torch.equal(q2, q) and torch.equal(k2, kk) and torch.equal(v2, vv)
    # the block's first LayerNorm folded into the loads == layernorm launch + plain form (key input normalised or not)
    pg, pb = dv(rnd("dpg", C, scale=0.2) + 1.0), dv(rnd("dpb", C, scale=0.2))
    raw = dv(nhwc(xn))
    normed = ops.layernorm(raw, pg, pb, 1e-5)
    args = (dv(w3[:, 0, 1].reshape(C, 9).t().contiguous()), dv(g[0]), dv(b[0]))
    kvw = (dv(wk.reshape(C, k * k).t().contiguous()), dv(wv.reshape(C, k * k).t().contiguous()), dv(g[1]), dv(b[1]), dv(g[2]), dv(b[2]), k)
    for ln_k in (True, False):
        xk_plain = normed if ln_k else dv(nhwc(xa))
        xk_fold = raw if ln_k else dv(nhwc(xa))
        ref3 = ops.qkv_prep(normed, *args, xk_plain, normed, *kvw)
        got3 = ops.qkv_prep(raw, *args, xk_fold, raw, *kvw, pre_ln=(pg, pb, 1e-5, ln_k))
        assert all(torch.equal(r_, g_) for r_, g_ in zip(ref3, got3)), (C, H, W, k, ln_k)


@pytest.mark.parametrize("C,Lq,Lk,heads", [(96, 200, 18, 2), (768, 84, 18, 2), (32, 64, 2, 2), (192, 333, 18, 2)])
def test_attention_core(ops, C, Lq, Lk, heads):
    n = 3
    q, k, v = rnd("aq", n, Lq, C), rnd("ak", n, Lk, C), rnd("av", n, Lk, C)
    d = C // heads
    qh, kh, vh = (t.reshape(n, -1, heads, d).transpose(1, 2) for t in (q, k, v))
    ref = (F.softmax(qh @ kh.transpose(-1, -2) * C ** -0.5, -1) @ vh).transpose(1, 2).reshape(n, Lq, C)
    got = ops.attention(q.to(DEV), k.to(DEV), v.to(DEV), heads, C ** -0.5)
    assert rel_err(got, ref) < 2e-5


@pytest.mark.parametrize("stage", [0, 1, 3])
def test_audio_fuse_matches_reference_layout_quirk(ops, stage):
    """Q3/Q4/Q5: nearest upsample iff both dims differ, softmax over W, NCTHW buffer reinterpreted as tokens."""
    B, T, C, ha, wa = 2, 9, 64, 2, 4
    H, W = ha * 2 ** stage, wa * 2 ** stage
    x5 = rnd("afx", B, C, T, H, W)
    audio = rnd("afa", B, 512, T, ha, wa)
    sd = {"p.align_conv.weight": rnd("afw", C, 512, 1, 1, scale=0.05), "p.align_conv.bias": rnd("afb", C, scale=0.1)}
    ref = orc.audio_fusion(sd, "p.", x5, audio)  # [B*T, HW, C] via the raw view
    a_tok = audio.permute(0, 2, 3, 4, 1).reshape(B * T, ha * wa, 512).contiguous()
    a_small = ops.linear(a_tok.to(DEV), sd["p.align_conv.weight"].reshape(C, 512).to(DEV), sd["p.align_conv.bias"].to(DEV))
    xf = x5.permute(0, 2, 3, 4, 1).contiguous().to(DEV)
    got = ops.audio_fuse(a_small, xf, ha, wa)
    assert got.shape == (B, C, T, H, W)
    assert rel_err(got.view(B * T, H * W, C), ref) < 2e-5


def test_pack_frames(ops):
    B, C, Tv, h, w = 2, 96, 8, 7, 12
    vis, nz = rnd("pfv", B, C, Tv, h, w), rnd("pfn", B, h, w, C)
    got = ops.pack_frames(vis.to(DEV), nz.to(DEV)).cpu()
    ref = torch.cat([vis, nz.permute(0, 3, 1, 2).unsqueeze(2)], dim=2).permute(0, 2, 3, 4, 1)
    assert torch.equal(got, ref.contiguous())
    got2 = ops.pack_frames(vis.to(DEV), None).cpu()
    assert torch.equal(got2, vis.permute(0, 2, 3, 4, 1).contiguous())


def test_pack_frames_multi_equals_per_stage_launches(ops):
    """The frame tensors of several stages in one launch == one pack_frames per stage (pure data movement: exact)."""
    B, Tv = 2, 8
    shapes = ((768, 7, 12), (384, 14, 24), (100, 5, 9))          # the last: channels not a multiple of 64, ragged token tiles
    vis = [rnd("pmv%d" % c, B, c, Tv, h, w).to(DEV) for c, h, w in shapes]
    nzs = [rnd("pmn%d" % c, B, h, w, c).to(DEV) for c, h, w in shapes]
    for noise in (nzs, [nzs[0], None, nzs[2]], [None, None, None]):
        outs = ops.pack_frames_multi(vis, noise, torch.float32)
        for v, nz, o in zip(vis, noise, outs):
            assert torch.equal(o, ops.pack_frames(v, nz))


@pytest.mark.parametrize("hw,HW,C", [((7, 12), (14, 24), 96), ((7, 12), (112, 192), 32), ((112, 192), (224, 384), 1), ((5, 9), (10, 18), 6)])
def test_resize_bilinear(ops, hw, HW, C):
    x = rnd("rs", 2, C, *hw)
    ref = F.interpolate(x, size=HW, mode="bilinear", align_corners=False)
    got = ops.resize_bilinear(nhwc(x).to(DEV), *HW)
    assert rel_err(got, nhwc(ref)) < 1e-5


def test_resize_sum_accumulates_in_stage_order(ops):
    xs = [rnd("rsum%d" % i, 2, 64, 3 * 2 ** i, 5 * 2 ** i) for i in range(4)]
    ref = 0
    for x in xs:
        ref = ref + F.interpolate(x, size=(48, 80), mode="bilinear", align_corners=False)
    got = ops.resize_sum([nhwc(x).to(DEV) for x in xs], 48, 80)
    assert rel_err(got, nhwc(ref)) < 1e-5


def test_head_sigmoid_and_axpy(ops):
    x, w, b = rnd("hx", 2, 96, 10, 12), rnd("hw", 1, 96, 1, 1, scale=0.2), rnd("hb", 1)
    ref = torch.sigmoid(F.conv2d(x, w, b))
    got = ops.head_sigmoid(nhwc(x).to(DEV), w.reshape(-1).to(DEV), b.to(DEV))
    assert rel_err(got, nhwc(ref)) < 1e-5
    a, c, e = rnd("a1", 1000), rnd("a2", 1000), rnd("a3", 1000)
    got = ops.axpbypcz(a.to(DEV), 0.5, c.to(DEV), -1.25, e.to(DEV), 2.0)
    assert rel_err(got, 0.5 * a - 1.25 * c + 2.0 * e) < 1e-6


@pytest.mark.parametrize("shape", [(96, 64, 3, 3), (40, 32, 3, 3), (256, 96, 5, 1, 1), (192, 96), (32, 512, 1, 1)])
def test_weight_pack_dgrad_pack_and_unpack_match_the_permute_definitions(shape):
    """csrc/pack.hip modes 0/1/2 against the layout definitions written as torch permutes (include/diffsal.h)."""
    from diff_sal_amd import ops

    torch.manual_seed(3)
    w = torch.randn(shape)
    w4 = w[:, :, :, 0, 0].unsqueeze(-1) if w.dim() == 5 else (w[:, :, None, None] if w.dim() == 2 else w)
    co, ci, kh, kw = w4.shape

    def pack_ref(v):  # [Co, Ci, kh, kw] -> [Co, (ci/32, tap, ci%32)]
        o, i = v.shape[:2]
        return v.permute(0, 2, 3, 1).reshape(o, kh * kw, i // 32, 32).permute(0, 2, 1, 3).reshape(o, kh * kw * i)

    wd = w.to("cuda")
    assert torch.equal(ops.pack_conv_weight(wd).cpu(), pack_ref(w4))
    if co % 32 == 0:
        assert torch.equal(ops.pack_dgrad_weight(wd).cpu(), pack_ref(w4.permute(1, 0, 2, 3).flip(2, 3)))
    # unpack is the autograd backward of the differentiable pack
    p = wd.clone().requires_grad_(True)
    g = torch.randn(co, kh * kw * ci)
    ops.pack_conv_weight_diff(p).backward(g.to("cuda"))
    ref = w4.clone().requires_grad_(True)
    pack_ref(ref).backward(g)
    assert torch.equal(p.grad.cpu().reshape(ref.grad.shape), ref.grad)


@pytest.mark.parametrize("shape", [(2, 14, 24, 96, 192, 3, 1, 1), (1, 28, 48, 384, 96, 3, 2, 2), (1, 1, 3000, 768, 768, 1, 0, 1)])
def test_conv_igemm_bf16x3_mode_accuracy(shape):
    """Opt-in split-precision mode of the GEMM kernel vs an fp64 convolution: error <= 2e-5 of the output maximum (each
    fp32 operand = bf16 hi + bf16 lo, product = hi*hi + hi*lo + lo*hi with fp32 accumulation; lo*lo ~ 2^-16 is dropped),
    and the result differs from the exact-fp32 mode (so the switch really selects another arithmetic)."""
    from diff_sal_amd import ops

    N, H, W, Cin, Cout, k, pad, dil = shape
    torch.manual_seed(1)
    x = torch.relu(torch.randn(N, Cin, H, W))
    w = torch.randn(Cout, Cin, k, k) / (Cin * k * k) ** 0.5
    ref = F.conv2d(x.double(), w.double(), padding=pad, dilation=dil).permute(0, 2, 3, 1)
    xd, wp = nhwc(x).to(DEV), ops.pack_conv_weight(w.to(DEV))
    kw = dict(kh=k, kw=k, pad=(pad, pad), dil=(dil, dil))
    y32 = ops.conv_igemm(xd, wp, **kw)
    with ops.gemm_precision("bf16x3"):                            # per-call descriptor field, scoped on the Python side
        y3 = ops.conv_igemm(xd, wp, **kw)
        y3s = ops.conv_igemm(xd, ops.split_weight(wp), **kw)     # weights pre-split on the host side of the launch
    assert ops.get_gemm_precision() == "fp32"
    assert torch.equal(y3s, y3)                                   # same arithmetic, the split just happens earlier
    # a pre-split weight carries its arithmetic: outside the context it still runs (and only runs) as bf16x3 ...
    assert torch.equal(ops.conv_igemm(xd, ops.split_weight(wp), **kw), y3)
    # ... and the library refuses the inconsistent descriptor (w_format = 1 with precision = fp32) outright
    import ctypes

    from diff_sal_amd import _lib
    d = _lib.ConvDesc(N, H, W, Cin, y32.shape[1], y32.shape[2], Cout, k, k, 1, 1, pad, pad, dil, dil, 0, 0, 1, 0, 0)
    rc = _lib.load().diffsal_conv_igemm(ctypes.byref(d), xd.data_ptr(), wp.data_ptr(), None, None, None, None, None,
                                        y32.data_ptr(), None, 0, torch.cuda.current_stream().cuda_stream)
    assert rc == -4 and b"w_format" in _lib.load().diffsal_last_error()
    m = ref.abs().max().item()
    e32 = (y32.cpu().double() - ref).abs().max().item() / m
    e3 = (y3.cpu().double() - ref).abs().max().item() / m
    print("fp32 err %.2e  bf16x3 err %.2e" % (e32, e3))
    assert e32 < 2e-6 and e3 < 2e-5
    assert not torch.equal(y3, y32)


@pytest.mark.parametrize("M,with_z", [(193536 // 8, True), (4001, False), (64, True)])
def test_mlp_block_fused_kernel(ops, M, with_z):
    """norm2 -> fc1 -> GELU -> fc2 -> +x1 (-> norm_mts on the kept frames) of the C = 96 TransformerBlock in one launch
    (transformer.py:153-157, common_block.py:125-147, sal_unet.py:447) vs the same chain in fp32 torch."""
    C, HID = 96, 192
    x1 = rnd("mbx", M, C) * 1.3 + 0.1
    g2, b2n = rnd("mbg", C, scale=0.1) + 1, rnd("mbb", C, scale=0.1)
    w1, bb1 = rnd("mbw1", HID, C, scale=0.12), rnd("mbb1", HID, scale=0.1)
    w2, bb2 = rnd("mbw2", C, HID, scale=0.08), rnd("mbb2", C, scale=0.1)
    gz, bz = rnd("mbgz", C, scale=0.1) + 1, rnd("mbbz", C, scale=0.1)
    ref_x2 = x1 + F.linear(F.gelu(F.linear(F.layer_norm(x1, (C,), g2, b2n, 1e-5), w1, bb1)), w2, bb2)
    ref_z = F.layer_norm(ref_x2, (C,), gz, bz, 1e-5)
    hw, T, keep = 7, 9, 5
    d = lambda t: t.to(DEV)
    x2, z = ops.mlp_block(d(x1), (d(g2), d(b2n), 1e-5), (d(w1), d(bb1)), (d(w2), d(bb2)),
                          (d(gz), d(bz), 1e-5) if with_z else None, (hw, T, keep))
    assert rel_err(x2, ref_x2) < 2e-5
    if with_z:
        rows = torch.arange(M)
        kept = ((rows // hw) % T) < keep
        assert rel_err(z.cpu()[kept], ref_z[kept]) < 2e-5
    else:
        assert z is None


@pytest.mark.parametrize("B,H,W", [(2, 32, 64), (1, 224, 384), (3, 8, 12)])
def test_conv_in_s4_equals_conv_in_then_stride4_downsample(ops, B, H, W):
    """The composed 5x5 stride-4 convolution == Conv2d(1, 96, 3, pad 1) followed by pad (0,1,0,1) + Conv2d(96, 96, 3, stride 4)
    (sal_unet.py:240,292 + :67-84) for H, W multiples of 4, and == the two-kernel HIP path."""
    C = 96
    x = rnd("s4x", B, 1, H, W)
    w1, b1 = rnd("s4w1", C, 1, 3, 3, scale=0.3), rnd("s4b1", C, scale=0.1)
    w2, b2 = rnd("s4w2", C, C, 3, 3, scale=(9 * C) ** -0.5), rnd("s4b2", C, scale=0.1)
    ref = F.conv2d(F.pad(F.conv2d(x, w1, b1, padding=1), (0, 1, 0, 1)), w2, b2, stride=4)
    w25, beff = ops.compose_conv_in_s4(w1.to(DEV), b1.to(DEV), w2.to(DEV), b2.to(DEV))
    got = ops.conv_in_s4(x.to(DEV), w25, beff)
    assert got.shape == nhwc(ref).shape
    assert rel_err(got, nhwc(ref)) < 2e-6
    f = ops.conv_in(x.to(DEV), w1.reshape(C, 9).contiguous().to(DEV), b1.to(DEV), skip_mod=4)
    two = ops.conv_igemm(f, ops.pack_conv_weight(w2.to(DEV)), kh=3, kw=3, stride=(4, 4), out_hw=tuple(ref.shape[-2:]), bias=b2.to(DEV))
    assert rel_err(got, two) < 5e-6


TAPSUM_CASES = [
    # N, Cin, Cout, (H, W), factors, dil
    (3, 64, 96, (16, 24), (2,), 2),          # UpEmbed conv1: bilinear x2 then 3x3 dilation 2; odd batch (half-empty wave)
    (2, 96, 160, (8, 12), (2,), 2),          # two 128-channel slabs
    (2, 64, 96, (32, 48), (2, 4, 8, 16), 1),  # mt_proj: 3x3 on the 4-scale sum
    (1, 32, 32, (16, 16), (1, 4), 1),        # a source already at the target resolution
    (3, 64, 96, (14, 24), (2,), 2),          # stage-1 UpEmbed: 7x12 -> 14x24, H not a multiple of 4 (ragged patch row)
    (2, 32, 64, (6, 10), (2,), 1),           # both extents ragged
]


@pytest.mark.parametrize("case", TAPSUM_CASES, ids=[str(c) for c in TAPSUM_CASES])
def test_tapsum_equals_conv_of_upsampled_sum(ops, case):
    """conv3x3(sum_i bilinear(z_i)) == tapsum of the per-source GEMMs with the nine 1x1 tap mixings (csrc/tapsum.hip)."""
    N, Cin, Cout, (H, W), factors, dil = case
    zs = [rnd("tz%d" % f, N, Cin, H // f, W // f) for f in factors]
    w = rnd("tw", Cout, Cin, 3, 3, scale=(9 * Cin) ** -0.5)
    b, sc, sh = rnd("tb", Cout, scale=0.1), 1.0 + rnd("tsc", Cout, scale=0.1), rnd("tsh", Cout, scale=0.1)
    up = sum(F.interpolate(z, size=(H, W), mode="bilinear", align_corners=False) if z.shape[-2:] != (H, W) else z for z in zs)
    ref = F.relu((F.conv2d(up, w, b, padding=dil, dilation=dil)) * sc[None, :, None, None] + sh[None, :, None, None])
    wcat = ops.tap_weight(w.to(DEV))
    ys = [ops.linear(nhwc(z).to(DEV), wcat, None) for z in zs]
    got = ops.tapsum(ys, H, W, Cout, dil=dil, bias=b.to(DEV), scale=sc.to(DEV), shift=sh.to(DEV), act=ops.ACT_RELU)
    assert got.shape == nhwc(ref).shape
    assert rel_err(got, nhwc(ref)) < 1e-5


@pytest.mark.parametrize("N,Cout,H,W,factors", [(4, 96, 16, 32, (2, 4)), (3, 32, 20, 24, (2, 4)), (1, 128, 8, 16, (2, 4)),
                                                (2, 96, 32, 64, (16, 8, 4, 2)),      # mt_proj's four scales
                                                (3, 96, 64, 64, (32, 16, 8)),         # 2-line sources, odd batch
                                                (3, 96, 24, 40, (2, 4)),              # H, W multiples of 8: 3 x 5 patch pairs
                                                (1, 60, 16, 16, (2, 8))])             # 4 channels per lane, 15 live lanes
def test_tapsum_with_folded_head(ops, tuning, N, Cout, H, W, factors):
    """MLPHead (1x1 to one channel + sigmoid, common_block.py:111-122) in the gather's epilogue == tapsum then head_sigmoid,
    for even / odd batches (the second image of a lane pair may be missing) and channel counts below the 128-lane slab."""
    Cin = 64
    zs = [rnd("hz%d" % f, N, Cin, H // f, W // f) for f in factors]
    w = rnd("hw", Cout, Cin, 3, 3, scale=(9 * Cin) ** -0.5)
    b, sc, sh = rnd("hb", Cout, scale=0.1), 1.0 + rnd("hsc", Cout, scale=0.1), rnd("hsh", Cout, scale=0.1)
    hw_, hb_ = rnd("hhw", Cout, scale=0.3).to(DEV), rnd("hhb", 1, scale=0.1).to(DEV)
    wcat = ops.tap_weight(w.to(DEV))
    ys = [ops.linear(nhwc(z).to(DEV), wcat, None) for z in zs]
    kw = dict(dil=1, bias=b.to(DEV), scale=sc.to(DEV), shift=sh.to(DEV), act=ops.ACT_RELU)
    two = ops.head_sigmoid(ops.tapsum(ys, H, W, Cout, **kw), hw_, hb_)
    one = ops.tapsum(ys, H, W, Cout, head=(hw_, hb_), **kw)
    assert one.shape == two.shape == (N, H, W, 1) and one.dtype == torch.float32
    assert (one - two).abs().max().item() < 2e-6
    up = sum(F.interpolate(z, size=(H, W), mode="bilinear", align_corners=False) for z in zs)
    ref = F.relu(F.conv2d(up, w, b, padding=1) * sc[None, :, None, None] + sh[None, :, None, None])
    ref = torch.sigmoid((ref * hw_.cpu()[None, :, None, None]).sum(1) + hb_.cpu())
    assert (one.cpu()[..., 0] - ref).abs().max().item() < 1e-5
    # the lane mappings of the row-streamed kernel (8 x 4 patches, two of one image or one of two images per wavefront; 4 x 4 patches)
    # sum the same terms in the same order
    for form in (1, 3):
        tuning.set("DIFFSAL_TAPSUM_ROWS_FORM", form)
        assert torch.equal(one, ops.tapsum(ys, H, W, Cout, head=(hw_, hb_), **kw))
    tuning.set("DIFFSAL_TAPSUM_ROWS_FORM", None)
    # H, W multiples of 4: the row-streamed head kernel ran; the general gather gives the same map up to summation order
    tuning.set("DIFFSAL_NO_TAPSUM_ROWS", 1)
    gen = ops.tapsum(ys, H, W, Cout, head=(hw_, hb_), **kw)
    assert (one - gen).abs().max().item() < 2e-6
    if H % 4 == 0 and W % 4 == 0:
        assert not torch.equal(one, gen), "the row-streamed head kernel did not run"


@pytest.mark.parametrize("case", TAPSUM_CASES, ids=[str(c) for c in TAPSUM_CASES])
def test_tapsum_autograd_matches_conv_of_upsampled_sum(ops, case):
    """Training path: d/dz_i, d/dW, d/dbias of conv3x3(sum_i bilinear(z_i)) through GEMM + tapsum and their adjoints
    (autograd_ops.TapSumFn, diffsal_tapsum_bwd) == torch autograd of the direct expression."""
    from diff_sal_amd import autograd_ops as ag
    N, Cin, Cout, (H, W), factors, dil = case
    zs = [rnd("gz%d" % f, N, Cin, H // f, W // f).requires_grad_() for f in factors]
    w = rnd("gw", Cout, Cin, 3, 3, scale=(9 * Cin) ** -0.5).requires_grad_()
    b = rnd("gb", Cout, scale=0.1).requires_grad_()
    g = rnd("gg", N, Cout, H, W)
    up = sum(F.interpolate(z, size=(H, W), mode="bilinear", align_corners=False) if z.shape[-2:] != (H, W) else z for z in zs)
    F.conv2d(up, w, b, padding=dil, dilation=dil).backward(g)
    zd = [nhwc(z.detach()).to(DEV).requires_grad_() for z in zs]
    wd, bd = w.detach().to(DEV).requires_grad_(), b.detach().to(DEV).requires_grad_()
    z_all = torch.cat([z.reshape(-1, Cin) for z in zd], 0)
    y9 = ag.linear(z_all, ag.tap_weight(wd))
    out = ag.tapsum(y9, [z.shape[1:3] for z in zd], N, H, W, Cout, dil=dil, bias=bd)
    out.backward(nhwc(g).to(DEV))
    for z, zr in zip(zd, zs):
        assert rel_err(z.grad, nhwc(zr.grad)) < 1e-5
    assert rel_err(wd.grad, w.grad) < 1e-5
    assert rel_err(bd.grad, b.grad) < 1e-5


@pytest.mark.parametrize("B,C,HW", [(4, 192, (56, 96)), (4, 96, (56, 96)), (2, 384, (28, 48)), (4, 768, (14, 24)), (1, 192, (9, 7))])
def test_groupnorm_single_launch_slab_path_equals_the_two_launch_path(ops, tuning, B, C, HW):
    """fp32 GroupNorm + swish: the LDS-resident one-launch kernel (a workgroup per image and group; up to 129 KB of slab at
    192 channels x 56 x 96) against the statistics + normalisation launches and against torch; B * groups < 128 (the last two
    cases) stays on the two-launch path by itself."""
    x = (rnd("gs%d_%d" % (C, HW[0]), B, *HW, C) * 1.7 + 0.4).to(DEV)
    g, b = (rnd("gsg", C, scale=0.2) + 1.0).to(DEV), rnd("gsb", C, scale=0.2).to(DEV)
    one = ops.groupnorm_swish(x, g, b, 32, 1e-6)
    tuning.set("DIFFSAL_NO_GN_SLAB", 1)
    two = ops.groupnorm_swish(x, g, b, 32, 1e-6)
    ref = F.silu(F.group_norm(x.permute(0, 3, 1, 2).double(), 32, g.double(), b.double(), 1e-6)).permute(0, 2, 3, 1)
    assert rel_err(one, ref) < 2e-5 and rel_err(two, ref) < 2e-5
    assert rel_err(one, two) < 3e-6
