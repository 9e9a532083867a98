"""Backward kernels (training step): every autograd Function against PyTorch autograd of the
same fp32 op on the CPU."""
import math

import pytest
import torch
import torch.nn.functional as F

from oracle import salunet_oracle as orc

pytestmark = pytest.mark.gpu
DEV = "cuda"


def rnd(name, *shape, scale=1.0):
    return orc.synth_tensor(name, shape, scale)


def nhwc(x):
    return x.permute(0, 2, 3, 1).contiguous()


def rel(got, ref):
    ref = ref.detach().float().cpu()
    return (got.detach().float().cpu() - ref).abs().max().item() / (ref.abs().max().item() + 1e-12)


@pytest.fixture(scope="module")
def ag():
    from diff_sal_amd import autograd_ops, ops

    return autograd_ops, ops


CASES = [
    # N, H, W, Cin, Cout, k, stride, pad, dil, asym, act
    (2, 14, 24, 96, 192, 3, 1, 1, 1, False, 0),
    (3, 12, 20, 64, 96, 3, 1, 2, 2, False, 1),     # dilated + ReLU
    (2, 15, 23, 96, 96, 3, 2, 0, 1, True, 0),      # Downsample (asym pad, stride 2)
    (1, 30, 46, 96, 96, 3, 4, 0, 1, True, 0),      # Downsample4x4
    # benchmark-size layers (one clip): mt_proj, stage-3 UpEmbed conv (9 frames, dilation 2), the first Downsample
    (1, 112, 192, 768, 96, 3, 1, 1, 1, False, 0),
    (9, 56, 96, 192, 96, 3, 1, 2, 2, False, 0),
    (1, 224, 384, 96, 96, 3, 4, 0, 1, True, 0),
    (2, 10, 12, 128, 64, 1, 1, 0, 1, False, 0),    # 1x1
]


@pytest.mark.parametrize("case", CASES)
def test_conv_backward_matches_autograd(ag, case):
    agops, ops = ag
    N, H, W, Cin, Cout, k, s, p, d, asym, act = case
    x = rnd("tx%d" % Cin, N, Cin, H, W).requires_grad_(True)
    w = rnd("tw%d%d" % (Cin, Cout), Cout, Cin, k, k, scale=1.0 / math.sqrt(Cin * k * k)).requires_grad_(True)
    b = rnd("tb", Cout, scale=0.1).requires_grad_(True)
    rv = rnd("trv", N, Cout, scale=0.3).requires_grad_(True)
    xin = F.pad(x, (0, 1, 0, 1)) if asym else x
    y = F.conv2d(xin, w, b, stride=s, padding=0 if asym else p, dilation=d) + rv[:, :, None, None]
    y = F.relu(y) if act else y
    gy = rnd("tgy", *y.shape)
    y.backward(gy)

    xd = nhwc(x.detach()).to(DEV).requires_grad_(True)
    wd = w.detach().to(DEV).requires_grad_(True)
    bd, rvd = b.detach().to(DEV).requires_grad_(True), rv.detach().to(DEV).requires_grad_(True)
    kw = dict(stride=(s, s), pad=(0, 0) if asym else (p, p), dil=(d, d), out_hw=tuple(y.shape[-2:]))
    yd = agops.conv(xd, ops.pack_conv_weight_diff(wd), kh=k, kw=k, bias=bd, rowvec=rvd, act=act,
                    w_dgrad=agops.dgrad_weight(wd, (s, s)), **kw)
    assert rel(yd, nhwc(y)) < 2e-5
    yd.backward(nhwc(gy).to(DEV))
    assert rel(xd.grad, nhwc(x.grad)) < 2e-5
    assert rel(wd.grad, w.grad) < 2e-5
    assert rel(bd.grad, b.grad) < 2e-5
    assert rel(rvd.grad, rv.grad) < 2e-5


@pytest.mark.parametrize("rows,K,N", [(50, 96, 192), (23000, 96, 96), (23000, 192, 96), (252, 768, 1536)])
def test_linear_backward_with_residual(ag, rows, K, N):
    """Token GEMMs incl. the shapes that take the streaming kernel in the data gradient (M >= 65536, K, N in {96,192})
    and the 1536-wide stage-0 MLP."""
    agops, ops = ag
    x = rnd("lx", 3, rows, K).requires_grad_(True)
    w = rnd("lw", N, K, scale=0.1).requires_grad_(True)
    b = rnd("lb", N, scale=0.1).requires_grad_(True)
    r = rnd("lr", 3, rows, N).requires_grad_(True)
    y = F.linear(x, w, b) + r
    gy = rnd("lgy", 3, rows, N)
    y.backward(gy)
    xd, wd, bd, rd = (t.detach().to(DEV).requires_grad_(True) for t in (x, w, b, r))
    yd = agops.linear(xd, wd, bd, residual=rd)
    yd.backward(gy.to(DEV))
    for got, ref in ((xd.grad, x.grad), (wd.grad, w.grad), (bd.grad, b.grad), (rd.grad, r.grad)):
        assert rel(got, ref) < 2e-5


@pytest.mark.parametrize("rows,K,N", [(672, 384, 96), (3361, 1536, 384), (21507, 192, 96), (29, 64, 32)])
def test_linear_with_the_gelu_in_front_folded_in(ag, rows, K, N):
    """Mlp's fc2(gelu(h)) as one node: the backward multiplies by gelu'(h) in the data-gradient product's epilogue
    (DIFFSAL_ACT_GELU_GRAD: persistent, tiled, split-K and scalar epilogues by shape); bit-equal to the two-node form."""
    agops, ops = ag
    h = rnd("gh", 3, rows, K).requires_grad_(True)
    w = rnd("gw", N, K, scale=0.1).requires_grad_(True)
    b = rnd("gb", N, scale=0.1).requires_grad_(True)
    r = rnd("gr", 3, rows, N).requires_grad_(True)
    gy = rnd("ggy", 3, rows, N)
    (F.linear(F.gelu(h), w, b) + r).backward(gy)
    hd, wd, bd, rd = (t.detach().to(DEV).requires_grad_(True) for t in (h, w, b, r))
    agops.linear(hd, wd, bd, residual=rd, in_gelu=True).backward(gy.to(DEV))
    for got, ref in ((hd.grad, h.grad), (wd.grad, w.grad), (bd.grad, b.grad), (rd.grad, r.grad)):
        assert rel(got, ref) < 2e-5
    h2, w2, b2, r2 = (t.detach().to(DEV).requires_grad_(True) for t in (h, w, b, r))
    agops.linear(agops.gelu(h2), w2, b2, residual=r2).backward(gy.to(DEV))
    assert torch.equal(h2.grad, hd.grad) and torch.equal(w2.grad, wd.grad)


@pytest.mark.parametrize("C", [96, 384, 32])
def test_layernorm_fork_adds_the_residual_gradient_in_its_backward_kernel(ag, C):
    """Pre-norm residual block: x -> (x, LN(x)) as one node; dx = LN'(d_branch) + d_residual from one launch."""
    agops, _ = ag
    x = (rnd("lfx%d" % C, 3, 61, C) * 1.7 + 0.2).requires_grad_(True)
    g = (rnd("lfg", C, scale=0.2) + 1.0).requires_grad_(True)
    b = rnd("lfb", C, scale=0.2).requires_grad_(True)
    w = rnd("lfw", C, C, scale=0.1)
    gy = rnd("lfgy", 3, 61, C)
    (x * 1.5 + F.linear(F.layer_norm(x, (C,), g, b, 1e-5), w)).backward(gy)
    xd, gd, bd = (t.detach().to(DEV).requires_grad_(True) for t in (x, g, b))
    xr, y = agops.layernorm_fork(xd, gd, bd, 1e-5)
    (xr * 1.5 + F.linear(y, w.to(DEV))).backward(gy.to(DEV))
    assert rel(xd.grad, x.grad) < 2e-5 and rel(gd.grad, g.grad) < 2e-5 and rel(bd.grad, b.grad) < 2e-5
    x2 = x.detach().to(DEV).requires_grad_(True)           # only the branch is used: no residual gradient
    agops.layernorm_fork(x2, gd.detach(), bd.detach(), 1e-5)[1].backward(gy.to(DEV))
    x3 = x.detach().clone().requires_grad_(True)
    F.layer_norm(x3, (C,), g.detach(), b.detach(), 1e-5).backward(gy)
    assert rel(x2.grad, x3.grad) < 2e-5


@pytest.mark.parametrize("C", [96, 192, 768, 32])
def test_layernorm_backward(ag, C):
    agops, _ = ag
    x = (rnd("lnx%d" % C, 4, 57, C) * 1.7 + 0.2).requires_grad_(True)
    g = (rnd("lng", C, scale=0.2) + 1.0).requires_grad_(True)
    b = rnd("lnb", C, scale=0.2).requires_grad_(True)
    gy = rnd("lngy", 4, 57, C)
    F.layer_norm(x, (C,), g, b, 1e-5).backward(gy)
    xd, gd, bd = (t.detach().to(DEV).requires_grad_(True) for t in (x, g, b))
    y = agops.layernorm(xd, gd, bd, 1e-5)
    y.backward(gy.to(DEV))
    assert rel(xd.grad, x.grad) < 2e-5 and rel(gd.grad, g.grad) < 2e-5 and rel(bd.grad, b.grad) < 2e-5


@pytest.mark.parametrize("C,HW", [(96, (14, 24)), (192, (7, 9)), (768, (5, 6))])
def test_groupnorm_swish_backward(ag, C, HW):
    agops, _ = ag
    x = (rnd("gnx%d" % C, 3, C, *HW) * 2.0 + 0.7).requires_grad_(True)
    g = (rnd("gng", C, scale=0.2) + 1.0).requires_grad_(True)
    b = rnd("gnb", C, scale=0.2).requires_grad_(True)
    y = orc.group_norm_swish(x, g, b)
    gy = rnd("gngy", *y.shape)
    y.backward(gy)
    xd = nhwc(x.detach()).to(DEV).requires_grad_(True)
    gd, bd = g.detach().to(DEV).requires_grad_(True), b.detach().to(DEV).requires_grad_(True)
    yd = agops.groupnorm_swish(xd, gd, bd, 32, 1e-6)
    assert rel(yd, nhwc(y)) < 2e-5
    yd.backward(nhwc(gy).to(DEV))
    assert rel(xd.grad, nhwc(x.grad)) < 5e-5 and rel(gd.grad, g.grad) < 5e-5 and rel(bd.grad, b.grad) < 5e-5


@pytest.mark.parametrize("relu", [True, False])
def test_batchnorm_train_forward_backward_and_running_stats(ag, relu):
    agops, _ = ag
    N, C, H, W = 5, 96, 9, 11
    x = (rnd("bnx", N, C, H, W) * 1.3 + 0.4).requires_grad_(True)
    bn = torch.nn.BatchNorm2d(C)
    with torch.no_grad():
        bn.weight.copy_(rnd("bnw", C, scale=0.2) + 1.0)
        bn.bias.copy_(rnd("bnb", C, scale=0.2))
    bn.train()
    y = bn(x)
    y = F.relu(y) if relu else y
    gy = rnd("bngy", *y.shape)
    y.backward(gy)
    bnd = torch.nn.BatchNorm2d(C).to(DEV)
    with torch.no_grad():
        bnd.weight.copy_(bn.weight.detach())
        bnd.bias.copy_(bn.bias.detach())
    xd = nhwc(x.detach()).to(DEV).requires_grad_(True)
    yd = agops.batchnorm_relu_train(xd, bnd, relu=relu)
    assert rel(yd, nhwc(y)) < 2e-5
    yd.backward(nhwc(gy).to(DEV))
    assert rel(xd.grad, nhwc(x.grad)) < 5e-5
    assert rel(bnd.weight.grad, bn.weight.grad) < 5e-5 and rel(bnd.bias.grad, bn.bias.grad) < 5e-5
    assert rel(bnd.running_mean, bn.running_mean) < 1e-5 and rel(bnd.running_var, bn.running_var) < 1e-5
    assert int(bnd.num_batches_tracked) == 1


def test_dropout_is_its_own_backward_and_unbiased(ag):
    agops, _ = ag
    x = torch.ones(1 << 20, device=DEV, requires_grad=True)
    y = agops.dropout(x, 0.1, seed=1234)
    keep = (y != 0).float().mean().item()
    assert abs(keep - 0.9) < 2e-3 and abs(y.mean().item() - 1.0) < 3e-3
    y.backward(torch.ones_like(y))
    assert torch.equal(x.grad, y.detach())
    y2 = agops.dropout(x, 0.1, seed=1235)
    assert not torch.equal(y2, y)


def test_gelu_and_add_backward(ag):
    agops, _ = ag
    x = rnd("gex", 7, 33, 96).requires_grad_(True)
    z = rnd("gez", 7, 33, 96).requires_grad_(True)
    gy = rnd("gegy", 7, 33, 96)
    (F.gelu(x) + z).backward(gy)
    xd, zd = x.detach().to(DEV).requires_grad_(True), z.detach().to(DEV).requires_grad_(True)
    agops.add(agops.gelu(xd), zd).backward(gy.to(DEV))
    assert rel(xd.grad, x.grad) < 2e-5 and rel(zd.grad, z.grad) < 1e-6


@pytest.mark.parametrize("C,H,W,k,stride,pad", [(96, 16, 32, 16, 16, 0), (192, 9, 13, 4, 4, 0), (64, 11, 7, 3, 1, 1), (768, 7, 12, 2, 2, 0)])
def test_depthwise_conv_backward(ag, C, H, W, k, stride, pad):
    agops, _ = ag
    x = rnd("dwx%d" % C, 2, C, H, W).requires_grad_(True)
    w = rnd("dww%d" % C, C, 1, k, k, scale=1.0 / k).requires_grad_(True)
    y = F.conv2d(x, w, stride=stride, padding=pad, groups=C)
    gy = rnd("dwgy", *y.shape)
    y.backward(gy)
    xd = nhwc(x.detach()).to(DEV).requires_grad_(True)
    wd = w.detach().reshape(C, k * k).t().contiguous().to(DEV).requires_grad_(True)
    yd = agops.dwconv(xd, wd, k, stride, pad)
    assert rel(yd, nhwc(y)) < 2e-5
    yd.backward(nhwc(gy).to(DEV))
    assert rel(xd.grad, nhwc(x.grad)) < 2e-5
    assert rel(wd.grad, w.grad.reshape(C, k * k).t()) < 2e-5


@pytest.mark.parametrize("C,Lq,Lk,heads", [(96, 200, 18, 2), (768, 84, 18, 2), (32, 64, 2, 2), (96, 5376, 18, 2)])
def test_attention_backward(ag, C, Lq, Lk, heads):
    agops, _ = ag
    n = 3   # the last case is one stage-3 frame of the benchmark (Lq = 56*96)
    q, k, v = (rnd(nm, n, L, C).requires_grad_(True) for nm, L in (("abq", Lq), ("abk", Lk), ("abv", Lk)))
    d = C // heads
    qh, kh, vh = (t.reshape(n, -1, heads, d).transpose(1, 2) for t in (q, k, v))
    o = (F.softmax(qh @ kh.transpose(-1, -2) * C ** -0.5, -1) @ vh).transpose(1, 2).reshape(n, Lq, C)
    go = rnd("abgo", n, Lq, C)
    o.backward(go)
    qd, kd, vd = (t.detach().to(DEV).requires_grad_(True) for t in (q, k, v))
    od = agops.attention(qd, kd, vd, heads, C ** -0.5)
    od.backward(go.to(DEV))
    assert rel(qd.grad, q.grad) < 5e-5 and rel(kd.grad, k.grad) < 5e-5 and rel(vd.grad, v.grad) < 5e-5


@pytest.mark.parametrize("hw,HW,C", [((7, 12), (14, 24), 96), ((3, 5), (48, 80), 32), ((56, 96), (112, 192), 1)])
def test_resize_backward_is_the_adjoint(ag, hw, HW, C):
    agops, _ = ag
    x = rnd("rbx", 2, C, *hw).requires_grad_(True)
    y = F.interpolate(x, size=HW, mode="bilinear", align_corners=False)
    gy = rnd("rbgy", *y.shape)
    y.backward(gy)
    xd = nhwc(x.detach()).to(DEV).requires_grad_(True)
    agops.resize_bilinear(xd, *HW).backward(nhwc(gy).to(DEV))
    assert rel(xd.grad, nhwc(x.grad)) < 2e-5


def test_resize_sum_backward(ag):
    agops, _ = ag
    xs = [rnd("rsb%d" % i, 2, 64, 3 * 2 ** i, 5 * 2 ** i).requires_grad_(True) for i in range(4)]
    y = sum(F.interpolate(x, size=(48, 80), mode="bilinear", align_corners=False) for x in xs)
    gy = rnd("rsbgy", *y.shape)
    y.backward(gy)
    xd = [nhwc(x.detach()).to(DEV).requires_grad_(True) for x in xs]
    agops.resize_sum(xd, 48, 80).backward(nhwc(gy).to(DEV))
    for a, b in zip(xd, xs):
        assert rel(a.grad, nhwc(b.grad)) < 2e-5


def test_pack_frames_backward(ag):
    agops, _ = ag
    vis = rnd("pbv", 2, 96, 8, 7, 12).requires_grad_(True)
    nz = rnd("pbn", 2, 7, 12, 96).requires_grad_(True)
    out = torch.cat([vis, nz.permute(0, 3, 1, 2).unsqueeze(2)], dim=2).permute(0, 2, 3, 4, 1)
    g = rnd("pbg", *out.shape)
    out.backward(g)
    vd, nd = vis.detach().to(DEV).requires_grad_(True), nz.detach().to(DEV).requires_grad_(True)
    agops.pack_frames(vd, nd).backward(g.to(DEV))
    assert torch.equal(vd.grad.cpu(), vis.grad) and torch.equal(nd.grad.cpu(), nz.grad)


def test_head_conv_in_and_dense_small_backward(ag):
    agops, _ = ag
    y, w, b = rnd("hby", 2, 96, 10, 12).requires_grad_(True), rnd("hbw", 1, 96, 1, 1, scale=0.2).requires_grad_(True), rnd("hbb", 1).requires_grad_(True)
    s = torch.sigmoid(F.conv2d(y, w, b))
    gs = rnd("hbg", *s.shape)
    s.backward(gs)
    yd = nhwc(y.detach()).to(DEV).requires_grad_(True)
    wd, bd = w.detach().reshape(-1).to(DEV).requires_grad_(True), b.detach().to(DEV).requires_grad_(True)
    agops.head_sigmoid(yd, wd, bd).backward(nhwc(gs).to(DEV))
    assert rel(yd.grad, nhwc(y.grad)) < 2e-5 and rel(wd.grad, w.grad.reshape(-1)) < 2e-5 and rel(bd.grad, b.grad) < 2e-5

    x = rnd("cibx", 2, 1, 24, 40)
    w2, b2 = rnd("cibw", 96, 1, 3, 3, scale=0.3).requires_grad_(True), rnd("cibb", 96, scale=0.1).requires_grad_(True)
    o = F.conv2d(x, w2, b2, padding=1)
    go = rnd("cibg", *o.shape)
    o.backward(go)
    w2d, b2d = w2.detach().reshape(96, 9).to(DEV).requires_grad_(True), b2.detach().to(DEV).requires_grad_(True)
    agops.conv_in(x.to(DEV), w2d, b2d, 0).backward(nhwc(go).to(DEV))
    assert rel(w2d.grad, w2.grad.reshape(96, 9)) < 2e-5 and rel(b2d.grad, b2.grad) < 2e-5

    xi, w3, b3 = rnd("dsx", 4, 384).requires_grad_(True), rnd("dsw", 1344, 384, scale=0.05).requires_grad_(True), rnd("dsb", 1344, scale=0.1).requires_grad_(True)
    z = F.linear(orc.swish(xi), w3, b3)
    gz = rnd("dsg", *z.shape)
    z.backward(gz)
    xid, w3d, b3d = (t.detach().to(DEV).requires_grad_(True) for t in (xi, w3, b3))
    agops.dense_small(xid, w3d, b3d, True).backward(gz.to(DEV))
    assert rel(xid.grad, xi.grad) < 2e-5 and rel(w3d.grad, w3.grad) < 2e-5 and rel(b3d.grad, b3.grad) < 2e-5


@pytest.mark.parametrize("stage", [0, 1, 3])
def test_audio_fuse_backward(ag, stage):
    agops, ops = ag
    B, T, C, ha, wa = 2, 9, 64, 2, 4
    H, W = ha * 2 ** stage, wa * 2 ** stage
    x5 = rnd("abx", B, C, T, H, W).requires_grad_(True)
    a_small = rnd("aba", B * T, ha * wa, C).requires_grad_(True)
    # reference: same math as oracle.audio_fusion after the 1x1 conv
    a = a_small.reshape(B, T, ha, wa, C).permute(0, 4, 1, 2, 3)
    if ha != H and wa != W:
        a = F.interpolate(a.reshape(B, C * T, ha, wa), scale_factor=H // ha, mode="nearest").reshape(B, C, T, H, W)
    m = F.softmax((a * x5).mean(dim=2, keepdim=True), dim=-1)
    out = a * m
    g = rnd("abg", *out.shape)
    out.backward(g)
    xd = x5.detach().permute(0, 2, 3, 4, 1).contiguous().to(DEV).requires_grad_(True)
    ad = a_small.detach().to(DEV).requires_grad_(True)
    od = agops.audio_fuse(ad, xd, ha, wa)
    assert rel(od, out) < 2e-5
    od.backward(g.to(DEV))
    assert rel(xd.grad, x5.grad.permute(0, 2, 3, 4, 1)) < 5e-5
    assert rel(ad.grad, a_small.grad) < 5e-5


def test_reduce_temp_backward_uses_disjoint_tap_path(ag):
    """Conv3d k=(5,1,1) stride 5 over T=9 frames -> one frame (common_block.py:125-142, quirk Q10): frames 0-4 get one
    tap each, frames 5-8 get zero gradient."""
    agops, ops = ag
    B, T, HW, C, Co = 2, 9, 40, 64, 96
    x = rnd("rtx", B, C, T, HW, 1).requires_grad_(True)
    w = rnd("rtw", Co, C, 5, 1, 1, scale=0.05).requires_grad_(True)
    y = F.relu(F.conv3d(x, w, None, stride=(5, 1, 1)))
    gy = rnd("rtg", *y.shape)
    y.backward(gy)
    xd = x.detach()[..., 0].permute(0, 2, 3, 1).contiguous().to(DEV).requires_grad_(True)     # [B, T, HW, C]
    wd = w.detach().to(DEV).requires_grad_(True)
    yd = agops.conv(xd, ops.pack_conv_weight_diff(wd), kh=5, kw=1, stride=(5, 1), act=1, w_dgrad=agops.dgrad_weight(wd, (5, 1)))
    assert tuple(agops.dgrad_weight(wd, (5, 1)).shape) == (5 * C, Co)
    assert rel(yd, y[..., 0].permute(0, 2, 3, 1)) < 2e-5
    yd.backward(gy[..., 0].permute(0, 2, 3, 1).contiguous().to(DEV))
    assert rel(xd.grad, x.grad[..., 0].permute(0, 2, 3, 1)) < 2e-5
    assert float(xd.grad[:, 5:].abs().max()) == 0.0
    assert rel(wd.grad, w.grad) < 2e-5


@pytest.mark.parametrize("M,C,seg", [(756, 1536, 756), (4096, 96, 4096), (3000, 2048, 750), (64, 4, 8)])
def test_colsum_wide_and_segmented(ag, M, C, seg):
    """Bias / per-image column sums, including rows wider than 1024 channels (stage-0 MLP hidden = 1536 at full size:
    a 256-thread row mapping once silently dropped channels >= 1024; only the full-size gradient test caught it)."""
    _, ops = ag
    torch.manual_seed(C)
    dy = torch.randn(M, C)
    got = ops.colsum(dy.to(DEV), seg).cpu()
    ref = dy.double().reshape(M // seg, seg, C).sum(1)
    assert tuple(got.shape) == (M // seg, C)
    assert (got.double() - ref).abs().max().item() < 1e-5 * ref.abs().max().item()


# ---- loss / clip / optimizer kernels (csrc/optim.hip) ----
def test_mse_loss_value_and_gradient(ag):
    A, _ = ag
    torch.manual_seed(11)
    pred = torch.rand(3, 1, 28, 48)
    tgt = torch.rand(3, 1, 28, 48)
    p = pred.clone().requires_grad_(True)
    ref = 0.7 * (p - tgt).square().sum(dim=(1, 2, 3)).mean(dim=0)   # R/models/sal_losses.py:189-192
    (3.0 * ref).backward()
    q = pred.to(DEV).requires_grad_(True)
    loss = A.mse_loss(q, tgt.to(DEV), 0.7 / 3)
    (3.0 * loss).backward()
    assert loss.shape == ()
    assert abs(loss.item() - ref.item()) < 1e-6 * abs(ref.item())
    assert (q.grad.cpu() - p.grad).abs().max().item() < 1e-6 * p.grad.abs().max().item()


@pytest.mark.parametrize("n", [4, 1000, 1 << 20, (1 << 22) + 3])
def test_grad_norm_matches_fp64(ag, n):
    _, ops = ag
    torch.manual_seed(n)
    g = torch.randn(n)
    got = ops.grad_norm(g.to(DEV), 0.5).item()
    ref = 0.5 * g.double().norm().item()
    assert abs(got - ref) < 1e-6 * ref


@pytest.mark.parametrize("max_norm,world", [(1.0, 1), (1.0, 4), (0.0, 1)])
def test_adam_step_matches_torch_adam_with_clipping(ag, max_norm, world):
    """3 steps of clip_grad_norm_ + torch.optim.Adam (R/diffusion_trainer.py:228-235, R/util/utils.py:116-123) on CPU vs
    grad_norm + adam_step on flat buffers; gradients arrive as rank SUMS, the mean is folded into gscale."""
    _, ops = ag
    torch.manual_seed(5)
    n = 70003
    p0 = torch.randn(n)
    ref_p = torch.nn.Parameter(p0.clone())
    opt = torch.optim.Adam([ref_p], lr=1e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, amsgrad=False)
    p, m, v = p0.to(DEV), torch.zeros(n, device=DEV), torch.zeros(n, device=DEV)
    for step in range(1, 4):
        gsum = torch.randn(n) * (10.0 if step == 2 else 0.001) * world   # step 2 is clipped, the others are not
        ref_p.grad = gsum / world
        if max_norm > 0:
            torch.nn.utils.clip_grad_norm_([ref_p], max_norm)
        opt.step()
        g = gsum.to(DEV)
        norm = ops.grad_norm(g, 1.0 / world) if max_norm > 0 else None
        ops.adam_step(p, g, m, v, step=step, lr=1e-4, betas=(0.9, 0.999), eps=1e-8, gscale=1.0 / world, norm=norm,
                      max_norm=max_norm, store_clipped_grad=True)
        assert (g.cpu() - ref_p.grad).abs().max().item() <= 2e-6 * ref_p.grad.abs().max().item()   # clipped grad written back
        st = opt.state[ref_p]
        assert (m.cpu() - st["exp_avg"]).abs().max().item() <= 1e-6 * st["exp_avg"].abs().max().item()
        assert (v.cpu() - st["exp_avg_sq"]).abs().max().item() <= 2e-6 * st["exp_avg_sq"].abs().max().item()
        # the update is ~lr * m / sqrt(v): compare the step taken, not the parameter (|p| ~ 1 hides it)
        assert (p.cpu() - ref_p.detach()).abs().max().item() <= 1e-4 * 1e-4 * step + 1.2e-7 * 4


@pytest.mark.parametrize("case", [(4, 28, 48, 192, 384, 1, True), (36, 14, 24, 384, 384, 2, False), (4, 56, 96, 96, 192, 1, True)])
def test_training_convolution_on_the_winograd_path(case):
    """ConvFn with the parameter at hand (w_raw) runs its forward and its data gradient on the F(4x4, 3x3) path with the weight transform
    computed on the device: against the direct convolution of the same node (forward, dx, dw, dbias, drowvec) and the device transform
    against the fp64 host transform."""
    from diff_sal_amd import autograd_ops as ag
    from diff_sal_amd import ops

    N, H, W, Cin, Cout, dil, extras = case
    w = orc.synth_tensor("tw%d" % Cout, (Cout, Cin, 3, 3), 0.05).to(DEV)
    U_host = ops.pack_wino4_weight(w)
    assert (ops.wino4_weight(w) - U_host).abs().max().item() <= 2e-6 * U_host.abs().max().item()
    wt = w.flip(2, 3).transpose(0, 1).contiguous()                  # the data gradient's convolution weight [Cin, Cout, 3, 3]
    Ud_host = ops.pack_wino4_weight(wt)
    assert (ops.wino4_weight(w, dgrad=True) - Ud_host).abs().max().item() <= 2e-6 * Ud_host.abs().max().item()
    x0 = orc.synth_tensor("tx%d" % Cin, (N, H, W, Cin)).to(DEV)
    assert ops.wino4_supported(x0, Cout, dil)
    bias = orc.synth_tensor("tb", (Cout,), 0.1).to(DEV) if extras else None
    rv0 = orc.synth_tensor("trv", (N, Cout), 0.1).to(DEV) if extras else None
    G = orc.synth_tensor("tg%d" % Cout, (N, H, W, Cout)).to(DEV)
    res = {}
    for form in ("wino", "direct"):
        ag.WINO4_TRAIN = form == "wino"
        try:
            x = x0.clone().requires_grad_(True)
            wl = w.clone().requires_grad_(True)
            b = bias.clone().requires_grad_(True) if extras else None
            rv = rv0.clone().requires_grad_(True) if extras else None
            y = ag.conv(x, ops.pack_conv_weight_diff(wl), kh=3, kw=3, pad=(dil, dil), dil=(dil, dil), bias=b, rowvec=rv, w_raw=wl)
            (y * G).sum().backward()
            res[form] = (y.detach(), x.grad, wl.grad) + ((b.grad, rv.grad) if extras else ())
        finally:
            ag.WINO4_TRAIN = True
    for a, b_, n in zip(res["wino"], res["direct"], ("y", "dx", "dw", "dbias", "drowvec")):
        e = (a - b_).abs().max().item() / b_.abs().max().item()
        print(f"{n}: winograd vs direct {e:.2e}")
        assert e < (1e-4 if n in ("y", "dx") else 1e-6), n          # dw / dbias / drowvec come from the same kernels on the same operands


@pytest.mark.parametrize("M,K,N", [(5000, 160, 100), (70001, 96, 576), (3024, 768, 864), (333, 32, 4)])
def test_wgrad_dma_kernel_equals_the_register_staged_kernel(M, K, N, tuning):
    """Plain-product weight gradients on the 128 x 192 tile run on wgrad_dma_kernel (both operands by LDS-DMA, three stages, one
    barrier per 32-row step): same rows in the same order through the same MFMA chains as wgrad_kernel -- bit-equal, including
    the bias column sums; ragged M / K / N (rows past the split, columns past Cout / K are cut by the DMA's range check)."""
    from diff_sal_amd import ops

    tuning.set("DIFFSAL_WGRAD_CFG", 0)           # the tile shape that has the DMA variant
    tuning.set("DIFFSAL_WGRAD_SPLITS", 5)        # one split of M for both (the two kernels are planned in rounds of 256 / 512 workgroups)
    x = orc.synth_tensor("wdx%d" % M, (1, 1, M, K)).to(DEV)
    dy = orc.synth_tensor("wdy%d" % M, (1, 1, M, N)).to(DEV)
    tuning.set("DIFFSAL_WGRAD_DMA", 0)
    dw0, db0 = ops.conv_wgrad(x, dy, want_bias=True)
    tuning.set("DIFFSAL_WGRAD_DMA", 1)           # the three-stage form, one workgroup per CU
    dw1, db1 = ops.conv_wgrad(x, dy, want_bias=True)
    assert torch.equal(dw0, dw1) and torch.equal(db0, db1)
    tuning.set("DIFFSAL_WGRAD_DMA", 2)           # the two-stage form, two workgroups per CU (round 6; what an unset switch takes)
    dw2, db2 = ops.conv_wgrad(x, dy, want_bias=True)
    assert torch.equal(dw0, dw2) and torch.equal(db0, db2)
    ref = dy.reshape(M, N).double().t() @ x.reshape(M, K).double()
    assert (dw1.double() - ref).abs().max().item() < 2e-5 * ref.abs().max().item()
    assert (db1.double() - dy.reshape(M, N).double().sum(0)).abs().max().item() < 1e-4 * (dy.abs().sum(dim=(0, 1, 2)).max().item())
