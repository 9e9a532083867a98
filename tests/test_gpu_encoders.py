"""Once-per-clip encoders: the general attention kernel, the MViTv2 pieces and the whole video encoder
against the CPU restatement / the reference's golden vectors; the on-device evaluation metrics."""
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import mvit_oracle as mo
from oracle import salunet_oracle as orc
from tests._cases import check_taps

pytestmark = pytest.mark.gpu
DEV = "cuda"
RTOL = 1e-3


def rel_err(got, ref):
    ref = ref.float().cpu()
    return (got.float().cpu() - ref).abs().max().item() / (ref.abs().max().item() + 1e-12)


def rnd(name, *shape, scale=1.0):
    return orc.synth_tensor(name, shape, scale)


@pytest.fixture(scope="module")
def ops():
    from diff_sal_amd import ops as o

    return o


def test_attention_general_tail_mode(ops):
    """352 workgroups (MViT stage 3 at batch 4): the 96 blocks past the first round run as two key-range pieces each and a
    finishing launch merges them by their log-sum-exps; output (with the residual) and the row log-sum-exp must not change."""
    B, H, Lq, Lk, D = 1, 16, 2689, 673, 96
    from diff_sal_amd import _lib
    assert _lib.load().diffsal_attention_general_tail_floats(B, H, Lq, Lk, D) == 2 * B * H * Lq * (D + 1)
    q, k, v = rnd("tq", B, H, Lq, D), rnd("tk", B, H, Lk, D), rnd("tv", B, H, Lk, D)
    s = (q * D ** -0.5) @ k.transpose(-1, -2)
    o = torch.softmax(s, -1) @ v
    o[:, :, 1:] += q[:, :, 1:]
    qd = q.to(DEV)
    got, lse = ops.attention_general(qd, k.to(DEV), v.to(DEV), scale=D ** -0.5, residual=qd, skip_first=True, want_lse=True)
    assert rel_err(got, o.transpose(1, 2).reshape(B, Lq, H * D)) < 2e-5
    assert (lse.cpu() - torch.logsumexp(s, -1)).abs().max().item() < 2e-5


@pytest.mark.parametrize("shape", [(2, 2, 300, 75, 64), (1, 1, 129, 673, 96), (2, 4, 33, 32, 96), (1, 2, 1, 5, 32)])
def test_attention_general_plain(ops, shape):
    """softmax(scale q k^T) v for several (B, H, Lq, Lk, D): ragged query / key tile remainders included."""
    B, H, Lq, Lk, D = shape
    q, k, v = rnd("gq", B, H, Lq, D), rnd("gk", B, H, Lk, D), rnd("gv", B, H, Lk, D)
    scale = D ** -0.5
    ref = (torch.softmax((q * scale) @ k.transpose(-1, -2), -1) @ v).transpose(1, 2).reshape(B, Lq, H * D)
    got = ops.attention_general(q.to(DEV), k.to(DEV), v.to(DEV), scale=scale)
    assert rel_err(got, ref) < 2e-5


def test_attention_general_reads_a_fused_qkv_in_place(ops):
    """q / k / v as strided views of one [B, N, 3, H, D] GEMM output (the AudioAttnNet call shape, heads 2 x 64)."""
    B, N, H, D = 2, 756, 2, 64
    qkv = rnd("fq", B, N, 3, H, D)
    q, k, v = (qkv[:, :, i].permute(0, 2, 1, 3) for i in range(3))
    ref = (torch.softmax((q * D ** -0.5) @ k.transpose(-1, -2), -1) @ v).transpose(1, 2).reshape(B, N, H * D)
    qd = qkv.to(DEV)
    got = ops.attention_general(*(qd[:, :, i].permute(0, 2, 1, 3) for i in range(3)), scale=D ** -0.5)
    assert rel_err(got, ref) < 2e-5


@pytest.mark.parametrize("E", [48, 32])
def test_attention_general_with_relative_position_bias_and_residual(ops, E):
    """The MViT call: decomposed rel-pos bias through the extra contraction columns (48- and 32-column layouts), class token
    without bias, residual pooling."""
    B, H, D = 2, 2, 96
    q_size, k_size = (2, 6, 10), (2, 3, 5)
    Lq, Lk = 1 + 2 * 6 * 10, 1 + 2 * 3 * 5
    q, k, v = rnd("rq", B, H, Lq, D), rnd("rk", B, H, Lk, D), rnd("rv", B, H, Lk, D)
    rel_t, rel_h, rel_w = rnd("rt", 2 * 2 - 1, D, scale=0.2), rnd("rh", 2 * 6 - 1, D, scale=0.2), rnd("rw", 2 * 10 - 1, D, scale=0.2)
    Rt, Rh, Rw = (mo.resize_decomposed_rel_pos(r, a, b) for r, a, b in ((rel_t, 2, 2), (rel_h, 6, 3), (rel_w, 10, 5)))
    attn = (q * D ** -0.5) @ k.transpose(-1, -2)
    rq = q[:, :, 1:].reshape(B, H, *q_size, D)
    rel = (torch.einsum("bythwc,tkc->bythwk", rq, Rt)[..., :, None, None] + torch.einsum("bythwc,hkc->bythwk", rq, Rh)[..., None, :, None]
           + torch.einsum("bythwc,wkc->bythwk", rq, Rw)[..., None, None, :])
    attn[:, :, 1:, 1:] += rel.reshape(B, H, Lq - 1, Lk - 1)
    o = attn.softmax(-1) @ v
    o[:, :, 1:] += q[:, :, 1:]
    ref = o.transpose(1, 2).reshape(B, Lq, H * D)
    qd = q.to(DEV)
    extra = ops.relpos_project(qd, Rt.contiguous().to(DEV), Rh.contiguous().to(DEV), Rw.contiguous().to(DEV), q_size, k_size, E)
    kt, kh, kw = k_size
    assert ops.relpos_columns(k_size) == 32 and ops.relpos_columns((8, 9, 12)) == 48
    oh = torch.zeros(Lk, E)
    l = torch.arange(Lk - 1)
    oh[1 + l, l // (kh * kw)] = 1
    oh[1 + l, 8 + (l // kw) % kh] = 1
    oh[1 + l, (24 if E == 48 else 16) + l % kw] = 1
    assert torch.equal(oh, ops.relpos_onehot(k_size, E, "cpu"))
    got = ops.attention_general(qd, k.to(DEV), v.to(DEV), scale=D ** -0.5, q_extra=extra, k_extra=oh.to(DEV), residual=qd,
                                skip_first=True)
    assert rel_err(got, ref) < 2e-5


@pytest.mark.parametrize("stride", [(1, 1, 1), (1, 2, 2), (1, 8, 8)])
def test_pool3d_ln_matches_attention_pool(ops, stride):
    B, heads, D, size = 2, 2, 96, (4, 9, 17)
    N = 1 + size[0] * size[1] * size[2]
    qkv = rnd("pq", B, N, 3, heads, D)
    w = rnd("pw", D, 1, 3, 3, 3, scale=0.2)
    g, b = rnd("pg", D, scale=0.1) + 1, rnd("pb", D, scale=0.1)
    ref, ref_size = mo.attention_pool(qkv[:, :, 1].permute(0, 2, 1, 3), w, stride, size, g, b)
    got, out_size = ops.pool3d_ln(qkv.to(DEV)[:, :, 1], w.reshape(D, 27).t().contiguous().to(DEV), g.to(DEV), b.to(DEV), size, stride)
    assert tuple(out_size) == tuple(ref_size) and rel_err(got, ref) < 2e-5


@pytest.mark.parametrize("strides", [((1, 1, 1), (1, 2, 2)), ((1, 2, 2), (1, 4, 4)), ((1, 1, 1), (1, 8, 8)), ((1, 1, 1), (1, 1, 1))])
def test_qkv_pool_one_launch_equals_three_pool3d_calls(ops, strides):
    """The fused q / k / v pooling (8 lanes x 12 channels per token) against the per-tensor kernel: the convolution is the same
    fmaf chain (bit-equal), the LayerNorm sums in another lane order (tolerance); the data gradient is bit-equal as well."""
    B, heads, D, size = 2, 4, 96, (3, 10, 13)
    N = 1 + size[0] * size[1] * size[2]
    qkv = rnd("fq", B, N, 3, heads, D).to(DEV)
    ws = [rnd(f"fw{i}", 27, D, scale=0.2).to(DEV) for i in range(3)]
    norms = [(rnd(f"fg{i}", D, scale=0.1).to(DEV) + 1, rnd(f"fb{i}", D, scale=0.1).to(DEV), 1e-5) for i in range(3)]
    sts = (strides[0], strides[1], strides[1])
    conv = ops.qkv_pool(qkv, ws, size, strides[0], strides[1])
    full = ops.qkv_pool(qkv, ws, size, strides[0], strides[1], norms=norms)
    for i in range(3):
        ref_c, sz = ops.pool3d(qkv[:, :, i], ws[i], size, sts[i])
        ref_f, _ = ops.pool3d_ln(qkv[:, :, i], ws[i], norms[i][0], norms[i][1], size, sts[i], norms[i][2])
        assert tuple(sz) == tuple(conv[3] if i == 0 else conv[4])
        assert torch.equal(conv[i], ref_c), f"tensor {i}: convolution differs"
        assert rel_err(full[i].cpu(), ref_f.cpu()) < 2e-6
    dys = [rnd(f"fd{i}", *conv[i].shape).to(DEV) for i in range(3)]
    got = ops.qkv_pool_bwd_data(dys, ws, qkv.shape, size, strides[0], strides[1])
    ref = torch.empty_like(qkv)
    for i in range(3):
        ops.pool3d_bwd(qkv[:, :, i], ws[i], dys[i], ref[:, :, i], size, sts[i])
    assert torch.equal(got, ref)
    dws = ops.qkv_pool_bwd_weight(qkv, dys, size, strides[0], strides[1])      # other chunking than the per-tensor kernel
    for i in range(3):
        assert rel_err(dws[i].cpu(), ops.pool3d_bwd_weight(qkv[:, :, i], dys[i], size, sts[i]).cpu()) < 2e-6


def test_maxpool_tokens_im2col_and_transpose(ops):
    B, C, size = 2, 192, (3, 8, 11)
    x = rnd("mx", B, 1 + size[0] * size[1] * size[2], C)
    t = x[:, 1:].reshape(B, *size, C).permute(0, 4, 1, 2, 3)
    ref = torch.cat([x[:, :1], F.max_pool3d(t, (1, 3, 3), (1, 2, 2), (0, 1, 1)).reshape(B, C, -1).transpose(1, 2)], 1)
    assert torch.equal(ops.maxpool_tokens(x.to(DEV), size, (1, 3, 3), (1, 2, 2)).cpu(), ref)
    # patch embedding = im2col + GEMM
    clip = rnd("clip", 1, 3, 6, 20, 28)
    w, bias = rnd("pew", 96, 3, 3, 7, 7, scale=0.05), rnd("peb", 96, scale=0.1)
    ref = F.conv3d(clip, w, bias, stride=(2, 4, 4), padding=(1, 3, 3))
    cols, size_o = ops.im2col3d(clip.to(DEV), (3, 7, 7), (2, 4, 4), (1, 3, 3), 448)
    assert tuple(size_o) == tuple(ref.shape[2:])
    wp = F.pad(w.reshape(96, -1), (0, 7)).contiguous().to(DEV)
    got = ops.linear(cols, wp, bias.to(DEV))
    assert rel_err(got, ref.flatten(2).transpose(1, 2).reshape(-1, 96)) < 2e-5
    # tokens -> channels first, skipping the class-token row
    tk = rnd("tk", 2, 1 + 70, 96)
    assert torch.equal(ops.tokens_to_channels_first(tk.to(DEV), 1).cpu(), tk[:, 1:].transpose(1, 2).contiguous())


def build_mvit(arch):
    from diff_sal_amd.mvit import MViT

    cfg = mo.MViTConfig(arch=arch)
    sd = mo.synth_state_dict(mo.state_dict_template(cfg))
    net = MViT(arch=arch if isinstance(arch, str) else dict(arch), out_scales=[0, 1, 2, 3])
    assert set(net.state_dict()) == set(sd) and all(tuple(v.shape) == tuple(sd[k].shape) for k, v in net.state_dict().items())
    net.load_state_dict(sd, strict=True)
    return net.to(DEV).eval().requires_grad_(False), cfg, sd


MVIT_CASES = {"tiny": (dict(embed_dims=96, num_layers=5, num_heads=1, downscale_indices=[1, 2, 4]), (2, 3, 16, 64, 96)),
              "small_full": ("small", (1, 3, 16, 224, 384))}


@pytest.mark.parametrize("name", list(MVIT_CASES))
def test_mvit_forward_matches_reference_golden(golden_dir, name):
    """The whole video encoder against the reference's own outputs (oracle/gen_golden.py::gen_mvit): four scales,
    coarsest first, plus per-block token taps.  1e-3 relative (north_star)."""
    arch, shape = MVIT_CASES[name]
    net, cfg, sd = build_mvit(arch)
    g = np.load(f"{golden_dir}/mvit_{name}.npz")
    x = orc.synth_tensor(f"mvit.{name}.x", shape)
    taps = {}
    with torch.no_grad():
        outs = net(x.to(DEV), taps=taps)
    torch.cuda.synchronize()
    assert len(outs) == 4 and outs[0].shape[1] == 768 and outs[3].shape[1] == 96
    named = {f"out{i}": o for i, o in enumerate(outs)}
    named.update({k: v for k, v in taps.items() if k.startswith("block")})
    worst = check_taps(named, g, RTOL)
    print(name, "worst rel err", max(worst.values()), "over", len(worst), "tensors")
    assert {"out0", "out3", "block0"} <= set(worst)


# 16-bit storage of the encoder's token stream and GEMM weights (MViT(compute_dtype=...)): relative error of the four feature maps
# against the reference's fp32 outputs, in units of each map's maximum.  bf16 keeps 8 mantissa bits through 16 residual blocks.
MVIT_LOWP_TOL = {torch.bfloat16: 2.5e-2, torch.float16: 4e-3}      # measured 1.0e-2 / 1.5e-3 on the full-size fixture


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16], ids=["bf16", "fp16"])
def test_mvit_16bit_storage_path_against_reference_golden(golden_dir, dtype):
    from diff_sal_amd.mvit import MViT

    arch, shape = MVIT_CASES["small_full"]
    net, cfg, sd = build_mvit(arch)
    lp = MViT(arch=arch, out_scales=[0, 1, 2, 3], compute_dtype=dtype)
    lp.load_state_dict(sd, strict=True)
    lp = lp.to(DEV).eval().requires_grad_(False)
    g = np.load(f"{golden_dir}/mvit_small_full.npz")
    x = orc.synth_tensor("mvit.small_full.x", shape).to(DEV)
    with torch.no_grad():
        outs = lp(x)
    assert all(o.dtype == torch.float32 for o in outs)
    worst = check_taps({f"out{i}": o for i, o in enumerate(outs)}, g, MVIT_LOWP_TOL[dtype])   # strided samples of the reference's maps
    print(dtype, "feature-map errors", worst)
    assert set(worst) == {"out0", "out1", "out2", "out3"}
    with torch.no_grad():                                                                   # and every element against the fp32 HIP path
        ref = net(x)
    assert max(rel_err(o, r) for o, r in zip(outs, ref)) < MVIT_LOWP_TOL[dtype]


def test_mvit_feeds_the_denoiser_through_video_saliency_model():
    """VideoSaliencyModel(visual_net=MViT, decoder_net=SalUNet): clip in, saliency map out, all on the HIP path."""
    from diff_sal_amd.diff_model import VideoSaliencyModel
    from tests.test_gpu_salunet import build

    cfg = orc.SalUNetConfig(img_size=(64, 96))
    dec = build(cfg, orc.synth_state_dict(orc.state_dict_template(cfg)))
    enc, mcfg, msd = build_mvit("small")
    model = VideoSaliencyModel(channel_list=None, visual_net=enc, decoder_net=dec).eval()
    clip = orc.synth_tensor("e2e.clip", (1, 3, 16, 64, 96))
    xt = orc.synth_tensor("e2e.x", (1, 1, 64, 96))
    with torch.no_grad():
        out = model({"img": clip.to(DEV), "input": xt.to(DEV)}, torch.tensor([500], device=DEV))
        feats = mo.mvit_forward(msd, mcfg, clip)
        ref = orc.salunet_forward(dec.state_dict() if False else orc.synth_state_dict(orc.state_dict_template(cfg)), cfg, xt,
                                  torch.tensor([500]), feats, None)
    assert out.shape == (1, 1, 64, 96) and (out.cpu() - ref).abs().max().item() < RTOL * ref.abs().max().item()


def test_saliency_metrics_match_the_reference_formulas(golden_dir):
    """CC / SIM / NSS / KL of R/models/sal_losses.py on the device vs the reference's own values (fixture) and vs a direct
    torch evaluation at the benchmark map size."""
    from diff_sal_amd import sal_losses as sl

    g = np.load(f"{golden_dir}/sal_metrics.npz")
    pred, gt = torch.from_numpy(g["pred"]), torch.from_numpy(g["gt"])
    m = sl.saliency_metrics(pred.to(DEV), gt.to(DEV))
    for k in ("cc", "sim", "nss", "kl"):
        ref = float(g[k])
        assert abs(float(m[k]) - ref) < 1e-4 * max(1.0, abs(ref)), (k, float(m[k]), ref)
    assert abs(float(sl.cc_s2(pred.to(DEV), gt.to(DEV))) - float(g["cc"])) < 1e-4
    # full-size maps, B = 4: formulas restated in fp64
    torch.manual_seed(0)
    p, t = torch.rand(4, 1, 224, 384), (torch.rand(4, 1, 224, 384) > 0.97).float() + 0.01 * torch.rand(4, 1, 224, 384)
    m = sl.saliency_metrics(p.to(DEV), t.to(DEV))
    pd, td = p.double().flatten(1), t.double().flatten(1)
    cc = (((pd - pd.mean(1, keepdim=True)) * (td - td.mean(1, keepdim=True))).sum(1)
          / ((pd - pd.mean(1, keepdim=True)).square().sum(1) * (td - td.mean(1, keepdim=True)).square().sum(1)).sqrt()).mean()
    nss = ((((pd - pd.mean(1, keepdim=True)) / (pd.std(1, keepdim=True) + 2.2204e-16)) * td).sum(1) / td.sum(1)).mean()
    assert abs(float(m["cc"]) - float(cc)) < 1e-5 and abs(float(m["nss"]) - float(nss)) < 1e-5 * max(1.0, abs(float(nss)))
    assert m["per_image"].shape == (4, 4)


def build_audio():
    from diff_sal_amd.audio_attention import AudioAttnNet
    from diff_sal_amd.vggish import VGGish
    from oracle import audio_oracle as ao

    vsd, asd = ao.synth_state_dict(ao.vgg_template(), "vgg."), ao.synth_state_dict(ao.attn_template(), "aan.")
    vgg = VGGish(pretrained=False)
    net = AudioAttnNet(depth=1, heads=2, dim=512, mlp_dim=256, patch_dim=512, num_patches=16, height=7, width=12, pool="cls",
                       dim_head=64, dropout=0.0, emb_dropout=0.0)
    assert set(vgg.state_dict()) == set(vsd) and set(net.state_dict()) == set(asd)
    vgg.load_state_dict(vsd)
    net.load_state_dict(asd)
    return vgg.to(DEV).eval(), net.to(DEV).eval(), vsd, asd


@pytest.mark.parametrize("name,shape", [("tiny", (2, 1, 9, 32, 64)), ("full", (1, 1, 9, 112, 192))])
def test_audio_branch_matches_reference_golden(golden_dir, name, shape):
    """VGGish.forward_feat and AudioAttnNet (module contracts, NCHW / NCTHW) and the fused channels-last path of
    VideoSaliencyModel.forward_vggish, against the reference's own outputs (oracle/gen_golden.py::gen_audio)."""
    from diff_sal_amd.diff_model import VideoSaliencyModel

    vgg, net, _, _ = build_audio()
    g = np.load(f"{golden_dir}/audio_{name}.npz")
    audio = orc.synth_tensor(f"audio.{name}", shape).to(DEV)
    bs, T = shape[0], shape[2]
    with torch.no_grad():
        f = vgg.forward_feat(audio.view(-1, 1, shape[3], shape[4]))
        out = net(f.reshape(bs, T, *f.shape[1:]).permute(0, 2, 1, 3, 4).contiguous())
        model = VideoSaliencyModel(channel_list=None, audio_net=vgg, spatiotemp_net=net)
        fused, fused2 = model.forward_vggish(audio)
    worst = check_taps({"features": f, "out": out}, g, RTOL)
    print(name, worst)
    assert fused is fused2 and fused.shape == out.shape and (fused - out).abs().max().item() < 1e-5 * out.abs().max().item()
    with pytest.raises(RuntimeError, match="non-singleton dimension 2"):
        net(torch.zeros(1, 512, 5, 2, 4, device=DEV))                 # the reference fails the same way for T != 9


def test_audio_visual_model_end_to_end_on_the_hip_path():
    """configs[2] plumbing: VideoSaliencyModel(MViT + VGGish + AudioAttnNet + SalUNet).forward(data, t) == the chain of
    the three CPU restatements on the same clip / audio / noisy map."""
    from diff_sal_amd.diff_model import VideoSaliencyModel
    from oracle import audio_oracle as ao
    from tests.test_gpu_salunet import build

    cfg = orc.SalUNetConfig()                                          # full widths; 224 x 384 so that audio (7 x 12) lines up
    dsd = orc.synth_state_dict(orc.state_dict_template(cfg))
    dec = build(cfg, dsd)
    enc, mcfg, msd = build_mvit("small")
    vgg, net, vsd, asd = build_audio()
    model = VideoSaliencyModel(channel_list=None, visual_net=enc, audio_net=vgg, spatiotemp_net=net, decoder_net=dec).eval()
    clip = orc.synth_tensor("av.clip", (1, 3, 16, 224, 384))
    audio = orc.synth_tensor("av.audio", (1, 1, 9, 112, 192))
    xt = orc.synth_tensor("av.x", (1, 1, 224, 384))
    t = torch.tensor([321])
    with torch.no_grad():
        out = model({"img": clip.to(DEV), "input": xt.to(DEV), "audio": audio.to(DEV)}, t.to(DEV))
        ref = orc.salunet_forward(dsd, cfg, xt, t, mo.mvit_forward(msd, mcfg, clip), ao.audio_branch(vsd, asd, audio))
    err = (out.cpu() - ref).abs().max().item()
    print("AV end-to-end max abs err", err)
    assert err < RTOL * ref.abs().max().item()


@pytest.mark.parametrize("name", ["all", "mse"])
def test_training_loss_terms_and_gradients_match_reference(golden_dir, name):
    """get_lossv2 with the saliency terms switched ON (R/models/sal_losses.py:179-259): loss values and d total / d pred on the
    device against the reference's own autograd (fixture of oracle/gen_golden.py::gen_loss_grads).  'all' = KL + CC + SIM + NSS,
    'mse' = weighted MSE + CC + NSS."""
    import types

    from diff_sal_amd import sal_losses as sl

    g = np.load(f"{golden_dir}/sal_loss_grads.npz")
    m = np.load(f"{golden_dir}/sal_metrics.npz")
    kl_w, cc_w, sim_w, nss_w, mse_w = (float(v) for v in g[f"{name}.cfg"])
    lc = dict(loss_kl=name == "all", loss_ce=False, loss_mse=name == "mse", loss_cc=True, loss_sim=name == "all", loss_nss=True,
              kl_weight=kl_w, cc_weight=cc_w, sim_weight=sim_w, nss_weight=nss_w, mse_weight=mse_w, ce_weight=1.0)
    cfg = types.SimpleNamespace(loss=types.SimpleNamespace(**lc))
    pred = torch.from_numpy(m["pred"]).to(DEV).requires_grad_(True)
    gt = torch.from_numpy(m["gt"]).to(DEV)
    out = sl.get_lossv2(cfg, pred, gt)
    out["total"].backward()
    for k in ("total", "main", "cc", "sim", "nss"):
        ref = float(g[f"{name}.{k}"])
        assert abs(float(out[k]) - ref) <= 2e-5 * max(1.0, abs(ref)), (k, float(out[k]), ref)
    gref = torch.from_numpy(g[f"{name}.grad"])
    err = (pred.grad.cpu() - gref).abs().max().item() / gref.abs().max().item()
    print(f"loss gradient [{name}]: rel err {err:.2e}")
    assert err < 1e-4


@pytest.mark.gpu
def test_training_loss_ce_branch_matches_reference(golden_dir):
    """get_lossv2 with loss_ce as its main term (cross_entropy_loss, R/models/sal_losses.py:48-63,187-188) + CC: values and
    d total / d pred against the reference's autograd (tests/golden/sal_loss_ce.npz, oracle/gen_golden.py::gen_loss_grads)."""
    import types

    from diff_sal_amd import sal_losses as sl

    g = np.load(f"{golden_dir}/sal_loss_ce.npz")
    ce_w, cc_w = (float(v) for v in g["cfg"])
    lc = dict(loss_kl=False, loss_ce=True, loss_mse=False, loss_cc=True, loss_sim=False, loss_nss=False,
              kl_weight=1.0, cc_weight=cc_w, sim_weight=1.0, nss_weight=1.0, mse_weight=1.0, ce_weight=ce_w)
    cfg = types.SimpleNamespace(loss=types.SimpleNamespace(**lc))
    pred = torch.from_numpy(g["pred"]).to(DEV).requires_grad_(True)
    out = sl.get_lossv2(cfg, pred, torch.from_numpy(g["gt"]).to(DEV))
    out["total"].backward()
    for k in ("total", "main", "cc"):
        ref = float(g[k])
        assert abs(float(out[k]) - ref) <= 2e-5 * max(1.0, abs(ref)), (k, float(out[k]), ref)
    gref = torch.from_numpy(g["grad"])
    err = (pred.grad.cpu() - gref).abs().max().item() / gref.abs().max().item()
    print(f"loss gradient [ce]: rel err {err:.2e}")
    assert err < 1e-4
