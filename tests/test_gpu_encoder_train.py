"""Training of the encoders on the HIP path: every new backward kernel against PyTorch autograd of the same fp32
operator, the whole MViT / AudioAttnNet gradients against autograd through the CPU restatements, and one
DiffusionTrainStep through VideoSaliencyModel(MViT + VGGish + AudioAttnNet + SalUNet)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import audio_oracle as ao
from oracle import mvit_oracle as mo
from oracle import salunet_oracle as orc

pytestmark = pytest.mark.gpu
DEV = "cuda"


def rnd(name, *shape, scale=1.0):
    return orc.synth_tensor(name, shape, scale)


def close(got, ref, tol, what=""):
    ref = ref.detach().float().cpu()
    e = (got.detach().float().cpu() - ref).abs().max().item() / (ref.abs().max().item() + 1e-12)
    assert e < tol, (what, e)
    return e


@pytest.mark.parametrize("shape", [(2, 2, 150, 70, 64, 0), (1, 2, 1 + 2 * 6 * 10, 1 + 2 * 3 * 5, 96, 48), (1, 2, 1 + 2 * 6 * 10, 1 + 2 * 3 * 5, 96, 32),
                                   (1, 1, 40, 129, 96, 0),
                                   (1, 16, 2689, 673, 96, 32)])      # MViT stage 3 at 4 clips: 352 dq blocks = 256 whole + 96 cut in two (tail mode)
def test_attention_general_backward(shape):
    """dq (incl. the residual path), dq_extra, dk, dv of the flash-attention backward vs autograd of the dense formula."""
    from diff_sal_amd import encoder_autograd as eg

    B, H, Lq, Lk, D, E = shape
    q, k, v = (rnd(n, B, H, L, D).requires_grad_(True) for n, L in (("bq", Lq), ("bk", Lk), ("bv", Lk)))
    qe = (rnd("bqe", B, H, Lq, E, scale=0.3).requires_grad_(True) if E else None)
    ke = None
    if E:
        ke = torch.zeros(Lk, E)
        ke[torch.arange(1, Lk), torch.randint(0, E, (Lk - 1,), generator=torch.Generator().manual_seed(1))] = 1.0
    G = rnd("bg", B, Lq, H * D)
    s = (q * D ** -0.5) @ k.transpose(-1, -2)
    if E:
        s = s + qe @ ke.t()
    o = s.softmax(-1) @ v
    if E:
        o = torch.cat([o[:, :, :1], o[:, :, 1:] + q[:, :, 1:]], 2)
    (o.transpose(1, 2).reshape(B, Lq, H * D) * G).sum().backward()
    qd, kd, vd = (t.detach().to(DEV).requires_grad_(True) for t in (q, k, v))
    qed = qe.detach().to(DEV).requires_grad_(True) if E else None
    out = eg.attention_general(qd, kd, vd, scale=D ** -0.5, q_extra=qed, k_extra=None if ke is None else ke.to(DEV),
                               residual_q=bool(E), skip_first=bool(E))
    (out * G.to(DEV)).sum().backward()
    close(out, o.transpose(1, 2).reshape(B, Lq, H * D), 2e-5, "out")
    for n, a, b in (("dq", qd, q), ("dk", kd, k), ("dv", vd, v)) + ((("dqe", qed, qe),) if E else ()):
        close(a.grad, b.grad, 5e-5, n)


@pytest.mark.parametrize("shape", [(2, 2, 150, 70, 64, 0), (1, 2, 121, 31, 96, 48), (1, 16, 2689, 673, 96, 32), (2, 1, 1000, 97, 96, 32)])
def test_attention_backward_ds_form_is_bit_identical(shape):
    """The dS form (dk / dv kernel leaves dS, the dq kernel is the one product dS K') against the recomputing dq kernel: the same
    sums in the same order -> identical bits in every output, with and without the dq kernel's tail mode."""
    from diff_sal_amd import ops

    B, H, Lq, Lk, D, E = shape
    q, k, v = (rnd(n, B, H, L, D).to(DEV) for n, L in (("bq", Lq), ("bk", Lk), ("bv", Lk)))
    qe = rnd("bqe", B, H, Lq, E, scale=0.3).to(DEV) if E else None
    ke = None
    if E:
        ke = torch.zeros(Lk, E)
        ke[torch.arange(1, Lk), torch.randint(0, E, (Lk - 1,), generator=torch.Generator().manual_seed(1))] = 1.0
        ke = ke.to(DEV)
    G = rnd("bg", B, Lq, H * D).to(DEV)
    kw = dict(scale=D ** -0.5, q_extra=qe, k_extra=ke, residual=q if E else None, skip_first=bool(E))
    out, lse = ops.attention_general(q, k, v, want_lse=True, **kw)
    a = ops.attention_general_bwd(q, k, v, out, lse, G, ds_form=False, **kw)
    b = ops.attention_general_bwd(q, k, v, out, lse, G, ds_form=True, **kw)
    for x, y, n in zip(a, b, ("dq", "dq_extra", "dk", "dv")):
        assert (x is None) == (y is None)
        if x is not None:
            assert torch.equal(x, y), n


RUN_CASES = [
    # size (T, H, W), stride_q, stride_kv        row lengths a multiple of 4: the run forms of csrc/mvit_pool.hip take the tensors they can
    ((3, 6, 8), (1, 1, 1), (1, 2, 2)),           # q: runs of 8, stride 1; k, v: stride 2 (forward / filter gradient runs of 4, data gradient runs of 8)
    ((2, 4, 12), (1, 2, 2), (1, 4, 4)),          # q: stride 2, runs of 4 (6 outputs per row: not a multiple of 4 -> forward stays on the token form); k, v: token form
    ((2, 8, 16), (1, 2, 2), (1, 1, 1)),          # q: stride 2 with 8 outputs per row; k, v: stride 1, runs of 8
    ((3, 5, 12), (1, 1, 1), (1, 1, 1)),          # runs of 4 at stride 1, odd height
]


@pytest.mark.parametrize("case", RUN_CASES, ids=[f"{c[0][1]}x{c[0][2]}_q{c[1][1]}_kv{c[2][1]}" for c in RUN_CASES])
def test_pool_run_forms(case, tuning):
    """Run forms of the three pooling kernels (a thread owns a run of consecutive tokens of one row): against autograd of the same
    depthwise convolutions, and against the token-per-lane-group forms -- forward and data gradient bit for bit (same taps, same fmaf
    chains), the filter gradient to summation order."""
    from diff_sal_amd import ops

    size, sq, skv = case
    B, heads, D = 2, 2, 96
    N = 1 + size[0] * size[1] * size[2]
    qkv = rnd("rq", B, N, 3, heads, D).requires_grad_(True)
    ws = [rnd(f"rw{i}", D, 1, 3, 3, 3, scale=0.2).requires_grad_(True) for i in range(3)]
    strides = (sq, skv, skv)
    refs = []
    for i in range(3):
        x = qkv[:, :, i].permute(0, 2, 1, 3)
        t = x[:, :, 1:].reshape(B * heads, *size, D).permute(0, 4, 1, 2, 3)
        t = F.conv3d(t, ws[i], None, stride=strides[i], padding=1, groups=D)
        refs.append(torch.cat([x[:, :, :1], t.reshape(B, heads, D, -1).transpose(2, 3)], 2))
    Gs = [rnd(f"rg{i}", *r.shape) for i, r in enumerate(refs)]
    sum((r * g).sum() for r, g in zip(refs, Gs)).backward()
    qd = qkv.detach().to(DEV)
    wd = [w.detach().reshape(D, 27).t().contiguous().to(DEV) for w in ws]
    gd = [g.to(DEV) for g in Gs]
    res = {}
    for form in ("runs", "tokens"):
        tuning.set("DIFFSAL_NO_POOL_RUNS", 1 if form == "tokens" else None)
        outs = ops.qkv_pool(qd, wd, size, sq, skv)[:3]
        dqkv = ops.qkv_pool_bwd_data(gd, wd, qd.shape, size, sq, skv)
        dws = ops.qkv_pool_bwd_weight(qd, gd, size, sq, skv)
        res[form] = (outs, dqkv, dws)
    for o, r in zip(res["runs"][0], refs):
        close(o, r, 2e-5, "pool fwd")
    close(res["runs"][1], qkv.grad, 5e-5, "dqkv")
    for i in range(3):
        close(res["runs"][2][i], ws[i].grad.reshape(D, 27).t(), 5e-5, f"dw{i}")
    for a, b in zip(res["runs"][0], res["tokens"][0]):
        assert torch.equal(a, b)
    assert torch.equal(res["runs"][1], res["tokens"][1])
    close(res["runs"][2], res["tokens"][2], 2e-6, "dw runs vs tokens")


@pytest.mark.parametrize("case", [(2, 2, (2, 6, 10), (2, 3, 5)), (1, 4, (8, 14, 24), (8, 7, 12)), (1, 1, (4, 28, 48), (4, 7, 12)), (2, 3, (2, 5, 7), (2, 5, 20))])
def test_attention_slot_form(case, tuning):
    """MViT's relative-position bias in slot form (k_slots: the three one-hot columns of a key; the forward adds three gathered q_extra
    values per (query, key) on the vector unit) against the contraction form (E extra columns of QK^T on the matrix pipe): output and
    log-sum-exp to summation order, and the gradients computed from either forward's log-sum-exp."""
    from diff_sal_amd import ops

    B, H, q_size, k_size = case
    D = 96
    E = ops.relpos_columns(k_size)
    Lq, Lk = 1 + q_size[0] * q_size[1] * q_size[2], 1 + k_size[0] * k_size[1] * k_size[2]
    q, k, v = (rnd(n, B, H, L, D).to(DEV) for n, L in (("sq", Lq), ("sk", Lk), ("sv", Lk)))
    qe = rnd("sqe", B, H, Lq, E, scale=0.3).to(DEV)
    oh = ops.relpos_onehot(k_size, E, DEV)
    assert oh.slots.shape == (Lk, 4) and int(oh.slots[0, 0]) == E
    # the slot table describes the one-hot table
    dense = torch.zeros(Lk, E + 1, device=DEV)
    dense.scatter_(1, oh.slots[:, :3].long(), 1.0)
    assert torch.equal(dense[:, :E], oh)
    G = rnd("sg", B, Lq, H * D).to(DEV)
    kw = dict(scale=D ** -0.5, q_extra=qe, k_extra=oh, residual=q, skip_first=True)
    res = {}
    for form in ("slots", "contraction"):
        tuning.set("DIFFSAL_NO_ATTN_SLOTS", 1 if form == "contraction" else None)
        out, lse = ops.attention_general(q, k, v, want_lse=True, **kw)
        grads = [ops.attention_general_bwd(q, k, v, out, lse, G, ds_form=f, **kw) for f in (True, False)]
        res[form] = (out, lse, grads)
    close(res["slots"][0], res["contraction"][0], 2e-6, "out")
    close(res["slots"][1], res["contraction"][1], 2e-6, "lse")
    for f in (0, 1):
        for a, b, n in zip(res["slots"][2][f], res["contraction"][2][f], ("dq", "dq_extra", "dk", "dv")):
            close(a, b, 5e-6, f"{n} (ds_form={not f})")


def test_layernorm_of_three_tensors_is_the_three_layernorms():
    """ops.layernorm_multi / layernorm_bwd_multi (one launch for norm_q / norm_k / norm_v of an MViT block) against the single-tensor
    kernels: outputs and dx bit for bit (same rows through the same code), dgamma / dbeta to the partial-sum partition."""
    from diff_sal_amd import autograd_ops as ag
    from diff_sal_amd import ops

    C = 96
    xs = [rnd(f"lx{i}", 2, 2, L, C).to(DEV) for i, L in enumerate((1 + 2 * 9 * 11, 34, 34))]
    gs = [(rnd(f"lg{i}", C, scale=0.2) + 1.0).to(DEV) for i in range(3)]
    bs = [rnd(f"lb{i}", C, scale=0.2).to(DEV) for i in range(3)]
    dys = [rnd(f"ld{i}", *x.shape).to(DEV) for i, x in enumerate(xs)]
    eps = (1e-6, 1e-6, 1e-5)
    outs = ops.layernorm_multi(xs, gs, bs, eps)
    dxs, dgs, dbs = ops.layernorm_bwd_multi(xs, dys, gs, eps)
    for i in range(3):
        assert torch.equal(outs[i], ops.layernorm(xs[i], gs[i], bs[i], eps[i]))
        dx, dg, db = ops.layernorm_bwd(xs[i], dys[i], gs[i], eps[i])
        assert torch.equal(dxs[i], dx)
        close(dgs[i], dg, 2e-6, "dgamma")
        close(dbs[i], db, 2e-6, "dbeta")
    # the autograd node against torch
    norms = [torch.nn.LayerNorm(C, eps=e).to(DEV) for e in eps]
    for n, g, b in zip(norms, gs, bs):
        n.weight.data.copy_(g)
        n.bias.data.copy_(b)
    xr = [x.clone().requires_grad_(True) for x in xs]
    sum((n(x) * d).sum() for n, x, d in zip(norms, xr, dys)).backward()
    ref = [(x.grad.clone(), n.weight.grad.clone(), n.bias.grad.clone()) for x, n in zip(xr, norms)]
    for n in norms:
        n.weight.grad = n.bias.grad = None
    xd = [x.clone().requires_grad_(True) for x in xs]
    ys = ag.layernorm3(xd, norms)
    sum((y * d).sum() for y, d in zip(ys, dys)).backward()
    for x, n, (rx, rg, rb) in zip(xd, norms, ref):
        close(x.grad, rx, 2e-5, "dx vs torch")
        close(n.weight.grad, rg, 2e-5, "dgamma vs torch")
        close(n.bias.grad, rb, 2e-5, "dbeta vs torch")


def test_pool_maxpool_relpos_backward():
    from diff_sal_amd import encoder_autograd as eg

    # depthwise pooling of q / k / v on a fused qkv tensor
    B, heads, D, size = 2, 2, 96, (3, 7, 9)
    N = 1 + size[0] * size[1] * size[2]
    qkv = rnd("tq", B, N, 3, heads, D).requires_grad_(True)
    ws = [rnd(f"tw{i}", D, 1, 3, 3, 3, scale=0.2).requires_grad_(True) for i in range(3)]
    # stride classes of the data-gradient kernel: 1 (three taps reach an input pixel per axis), 2 (two), >= 3 (one), unequal (all 27 tried)
    for strides in (((1, 2, 2), (1, 4, 4), (1, 4, 4)), ((1, 1, 1), (1, 8, 8), (1, 8, 8)), ((1, 1, 1), (1, 2, 4), (1, 2, 4))):
        qkv.grad = None
        for w in ws:
            w.grad = None
        refs = []
        for i in range(3):
            x = qkv[:, :, i].permute(0, 2, 1, 3)
            t = x[:, :, 1:].reshape(B * heads, *size, D).permute(0, 4, 1, 2, 3)
            t = F.conv3d(t, ws[i], None, stride=strides[i], padding=1, groups=D)
            refs.append(torch.cat([x[:, :, :1], t.reshape(B, heads, D, -1).transpose(2, 3)], 2))
        Gs = [rnd(f"tg{i}", *r.shape) for i, r in enumerate(refs)]
        sum((r * g).sum() for r, g in zip(refs, Gs)).backward()
        for param_layout in (False, True):     # filters tap-major [27, D] (packed) or in the parameter's own [D, 27] layout
            qd = qkv.detach().to(DEV).requires_grad_(True)
            wd = [(w.detach().reshape(D, 27) if param_layout else w.detach().reshape(D, 27).t().contiguous()).to(DEV).requires_grad_(True)
                  for w in ws]
            outs = eg.qkv_pool(qd, wd[0], wd[1], wd[2], size, strides[0], strides[1])
            sum((o * g.to(DEV)).sum() for o, g in zip(outs, Gs)).backward()
            for o, r in zip(outs, refs):
                close(o, r, 2e-5, "pool fwd")
            close(qd.grad, qkv.grad, 5e-5, "dqkv")
            for i in range(3):
                ref_dw = ws[i].grad.reshape(D, 27)
                close(wd[i].grad, ref_dw if param_layout else ref_dw.t(), 5e-5, f"dw{i}")
    # max-pool of the skip path
    C, size = 64, (2, 6, 9)
    x = rnd("mp", 2, 1 + size[0] * size[1] * size[2], C).requires_grad_(True)
    t = x[:, 1:].reshape(2, *size, C).permute(0, 4, 1, 2, 3)
    ref = torch.cat([x[:, :1], F.max_pool3d(t, (1, 3, 3), (1, 2, 2), (0, 1, 1)).reshape(2, C, -1).transpose(1, 2)], 1)
    g = rnd("mpg", *ref.shape)
    (ref * g).sum().backward()
    xd = x.detach().to(DEV).requires_grad_(True)
    got = eg.maxpool_tokens(xd, size, (1, 3, 3), (1, 2, 2))
    (got * g.to(DEV)).sum().backward()
    assert torch.equal(got.detach().cpu(), ref.detach())
    close(xd.grad, x.grad, 1e-6, "maxpool dx")       # an input that wins several windows sums their gradients in another order
    # relative-position projections
    B, heads, D, q_size, k_size = 2, 2, 96, (2, 5, 7), (2, 3, 4)
    L = q_size[0] * q_size[1] * q_size[2]
    q = rnd("rq", B, heads, 1 + L, D).requires_grad_(True)
    Rs = [rnd(f"rr{i}", a, b, D, scale=0.2).requires_grad_(True) for i, (a, b) in enumerate(zip(q_size, k_size))]
    rq = q[:, :, 1:].reshape(B, heads, *q_size, D)
    for E, w0 in ((48, 24), (32, 16)):          # both column layouts of the bias
        q.grad = None
        for r in Rs:
            r.grad = None
        ex = torch.zeros(B, heads, 1 + L, E)
        ex[:, :, 1:, 0:k_size[0]] = torch.einsum("bythwc,tkc->bythwk", rq, Rs[0]).reshape(B, heads, L, -1)
        ex[:, :, 1:, 8:8 + k_size[1]] = torch.einsum("bythwc,hkc->bythwk", rq, Rs[1]).reshape(B, heads, L, -1)
        ex[:, :, 1:, w0:w0 + k_size[2]] = torch.einsum("bythwc,wkc->bythwk", rq, Rs[2]).reshape(B, heads, L, -1)
        g = rnd(f"rg{E}", *ex.shape)
        (ex * g).sum().backward()
        qd = q.detach().to(DEV).requires_grad_(True)
        Rd = [r.detach().to(DEV).requires_grad_(True) for r in Rs]
        got = eg.relpos_project(qd, Rd[0], Rd[1], Rd[2], q_size, k_size, E)
        (got * g.to(DEV)).sum().backward()
        close(got, ex, 2e-5, "relpos fwd")
        close(qd.grad, q.grad, 5e-5, "relpos dq")
        for i in range(3):
            close(Rd[i].grad, Rs[i].grad, 5e-5, f"dR{i}")


def test_mvit_all_parameter_gradients_match_oracle_autograd():
    """Whole video encoder: d(sum_i <out_i, G_i>) / d(every parameter) on the HIP path vs autograd through the CPU
    restatement of R/models/mvit.py (smooth graph: GELU / softmax / LayerNorm; the only kinks are max-pool ties)."""
    from tests.test_gpu_encoders import build_mvit

    arch = dict(embed_dims=96, num_layers=5, num_heads=1, downscale_indices=[1, 2, 4])
    net, cfg, sd = build_mvit(arch)
    net.requires_grad_(True)
    clip = rnd("mvit.train.x", 2, 3, 16, 64, 96)
    leaf = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    outs_ref = mo.mvit_forward(leaf, cfg, clip)
    Gs = [rnd(f"mvit.train.g{i}", *o.shape) for i, o in enumerate(outs_ref)]
    sum((o * g).sum() for o, g in zip(outs_ref, Gs)).backward()
    outs = net(clip.to(DEV))
    assert all(o.requires_grad for o in outs)
    sum((o * g.to(DEV)).sum() for o, g in zip(outs, Gs)).backward()
    for o, r in zip(outs, outs_ref):
        close(o, r, 1e-4, "mvit out")
    scale = torch.stack([v.grad.abs().max() for v in leaf.values()]).median().item()
    worst = {}
    for name, p in net.named_parameters():
        ref = leaf[name].grad
        assert p.grad is not None, name
        e = (p.grad.cpu() - ref).abs().max().item() / max(ref.abs().max().item(), 1e-3 * scale)
        worst[name] = e
    bad = {k: v for k, v in worst.items() if v > 2e-3}
    print("mvit grads: worst", max(worst.values()), "median", float(np.median(list(worst.values()))))
    assert not bad, bad


def test_audio_attn_gradients_match_oracle_autograd():
    from tests.test_gpu_encoders import build_audio

    _, net, _, asd = build_audio()
    f = rnd("aan.train.f", 2, 512, 9, 2, 4)
    leaf = {k: v.clone().requires_grad_(True) for k, v in asd.items()}
    fr = f.clone().requires_grad_(True)
    ref = ao.audio_attn_forward(leaf, fr)
    g = rnd("aan.train.g", *ref.shape)
    (ref * g).sum().backward()
    fd = f.to(DEV).requires_grad_(True)
    out = net(fd)
    (out * g.to(DEV)).sum().backward()
    close(out, ref, 1e-4, "aan out")
    close(fd.grad, fr.grad, 1e-3, "aan dinput")
    for name, p in net.named_parameters():
        if leaf[name].grad is None:          # to_patch_embedding / pos_embedding: dead in the reference too (Q13)
            assert p.grad is None or float(p.grad.abs().max()) == 0.0, name
            continue
        close(p.grad, leaf[name].grad, 1e-3, name)


def test_training_step_through_the_full_audio_visual_model():
    """configs[3] as the reference runs it (R/diffusion_trainer.py:212-235): MViT + (frozen) VGGish + AudioAttnNet + SalUNet
    in one DiffusionTrainStep.  The first step's loss and gradient norm against autograd through the chained CPU
    restatements; then the loss must fall on a repeated batch."""
    from diff_sal_amd.diff_model import VideoSaliencyModel
    from diff_sal_amd.train_step import DiffusionTrainStep
    from tests.test_gpu_encoders import build_audio, build_mvit
    from tests.test_gpu_salunet import build

    cfg = orc.SalUNetConfig(img_size=(64, 128))
    dsd = orc.synth_state_dict(orc.state_dict_template(cfg))
    dec = build(cfg, dsd)
    dec.dropout_p = 0.0
    enc, mcfg, msd = build_mvit("small")
    enc.requires_grad_(True)
    vgg, aan, vsd, asd = build_audio()
    model = VideoSaliencyModel(channel_list=None, visual_net=enc, audio_net=vgg, spatiotemp_net=aan, decoder_net=dec).to(DEV)
    B = 2
    clip, audio = rnd("avt.clip", B, 3, 16, 64, 128), rnd("avt.audio", B, 1, 9, 32, 64)
    sal = torch.sigmoid(rnd("avt.sal", B, 1, 64, 128))
    noise = rnd("avt.noise", B, 1, 64, 128)
    t0 = 400
    ts = DiffusionTrainStep(model, lr=1e-4, grad_clip=1.0, store_clipped_grad=False)
    n_train = sum(p.numel() for p in model.parameters() if p.requires_grad)
    assert ts.flat.live_numel == n_train and not any(p.requires_grad for p in vgg.parameters())
    # reference: chained restatements, train-mode BatchNorm in the denoiser
    leaf_d = {k: v.clone().requires_grad_(v.dtype.is_floating_point and "running" not in k) for k, v in dsd.items()}
    leaf_m = {k: v.clone().requires_grad_(True) for k, v in msd.items()}
    leaf_a = {k: v.clone().requires_grad_(True) for k, v in asd.items()}
    a_hat = (1.0 - ts_betas()).cumprod(0)
    x_t = a_hat[t0].sqrt() * sal + (1 - a_hat[t0]).sqrt() * noise
    orc.BN_TRAIN = True
    try:
        with torch.no_grad():
            vf = ao.vgg_features(vsd, audio.reshape(-1, 1, 32, 64))
        vf = vf.reshape(B, 9, *vf.shape[1:]).permute(0, 2, 1, 3, 4)
        pred = orc.salunet_forward(leaf_d, cfg, x_t, torch.full((B,), t0), mo.mvit_forward(leaf_m, mcfg, clip),
                                   ao.audio_attn_forward(leaf_a, vf))
    finally:
        orc.BN_TRAIN = False
    loss_ref = ((pred - sal) ** 2).sum(dim=(1, 2, 3)).mean()
    loss_ref.backward()
    gn_ref = torch.sqrt(sum(v.grad.double().pow(2).sum() for d in (leaf_d, leaf_m, leaf_a) for v in d.values() if v.grad is not None))
    data = {"img": clip.to(DEV), "audio": audio.to(DEV)}
    kw = dict(t0=t0, noise=noise.to(DEV), dequant_noise=torch.zeros_like(sal).to(DEV))
    l0 = ts.step(sal.to(DEV), data, **kw)
    print("loss", float(l0), "ref", float(loss_ref), "grad norm", float(ts.last_norm), "ref", float(gn_ref))
    assert abs(float(l0) - float(loss_ref)) < 1e-4 * abs(float(loss_ref))
    assert abs(float(ts.last_norm) - float(gn_ref)) < 2e-3 * float(gn_ref)
    losses = [float(l0)] + [float(ts.step(sal.to(DEV), data, **kw)) for _ in range(4)]
    print("losses", losses)
    assert losses[-1] < losses[0]


def ts_betas():
    from diff_sal_amd.diffusion_utils import get_beta_schedule, to_torch

    return to_torch(get_beta_schedule("cosine", beta_start=1e-4, beta_end=0.02, num_diffusion_timesteps=1000))


def test_eval_after_training_uses_the_trained_encoder_weights():
    """The fused Adam kernel writes parameters through the flat buffer (no autograd version bump): the eval-mode caches of
    EVERY module (MViT's packed patch / pooling weights and gathered relative-position tables, not only the denoiser's GEMM
    weights) must be rebuilt.  After training steps the no_grad forward is compared with the chained CPU restatements
    evaluated on the UPDATED state_dict; EMAHelper.ema() on the top-level module must reach the nested modules too."""
    from diff_sal_amd.diff_model import VideoSaliencyModel
    from diff_sal_amd.ema import EMAHelper
    from diff_sal_amd.train_step import DiffusionTrainStep
    from tests.test_gpu_encoders import build_mvit
    from tests.test_gpu_salunet import build

    cfg = orc.SalUNetConfig(img_size=(64, 128))
    dec = build(cfg, orc.synth_state_dict(orc.state_dict_template(cfg)))
    dec.dropout_p = 0.0
    enc, mcfg, _ = build_mvit("small")
    enc.requires_grad_(True)
    model = VideoSaliencyModel(channel_list=None, visual_net=enc, decoder_net=dec).to(DEV)
    B = 2
    clip = rnd("evt.clip", B, 3, 16, 64, 128)
    sal = torch.sigmoid(rnd("evt.sal", B, 1, 64, 128))
    x_in, t = rnd("evt.x", B, 1, 64, 128), torch.tensor([300, 700])

    def hip_eval():
        model.eval()
        with torch.no_grad():
            feats = model.visual_net(clip.to(DEV))
            return [f.float().cpu() for f in feats], model.decoder_net(x_in.to(DEV), t.to(DEV), feats).float().cpu()

    def oracle_eval():
        msd = {k: v.detach().float().cpu() for k, v in model.visual_net.state_dict().items()}
        dsd = {k: v.detach().cpu() for k, v in model.decoder_net.state_dict().items()}
        with torch.no_grad():
            feats = mo.mvit_forward(msd, mcfg, clip)
            return feats, orc.salunet_forward(dsd, cfg, x_in, t, feats, None)

    hip_eval()                                   # populate every eval cache with the INITIAL weights
    ts = DiffusionTrainStep(model, lr=2e-3, grad_clip=1.0)      # large steps: stale weights would be far outside the tolerance
    model.train()
    for _ in range(3):
        ts.step(sal.to(DEV), {"img": clip.to(DEV)}, t0=400, noise=rnd("evt.n", B, 1, 64, 128).to(DEV),
                dequant_noise=torch.zeros_like(sal).to(DEV))
    f_hip, o_hip = hip_eval()
    f_ref, o_ref = oracle_eval()
    for i, (a, b) in enumerate(zip(f_hip, f_ref)):
        close(a, b, 1e-4, f"feature {i} after training")
    assert (o_hip - o_ref).abs().max().item() < 1e-4

    # EMA on the top-level module: shadow = the weights before three more steps; ema() must make the eval path use them again
    ema = EMAHelper(mu=0.0)
    ema.register(model)
    ema.update(model)                             # mu = 0: shadow = current parameters
    f_keep, o_keep = hip_eval()
    model.train()
    for _ in range(2):
        ts.step(sal.to(DEV), {"img": clip.to(DEV)}, t0=200, noise=rnd("evt.n2", B, 1, 64, 128).to(DEV),
                dequant_noise=torch.zeros_like(sal).to(DEV))
    f_moved, _ = hip_eval()
    assert max((a - b).abs().max().item() for a, b in zip(f_moved, f_keep)) > 1e-4      # the two steps did move the encoder
    ema.ema(model)
    f_back, o_back = hip_eval()
    for a, b in zip(f_back, f_keep):
        assert torch.equal(a, b)
    # the denoiser's BatchNorm running statistics are buffers (EMA covers parameters only), so its output is not o_keep; what
    # must hold is that no module still evaluates with a stale packed copy: an explicit invalidation changes nothing
    for m in model.modules():
        if hasattr(m, "parameters_updated"):
            m.parameters_updated()
    f_again, o_again = hip_eval()
    assert torch.equal(o_back, o_again) and all(torch.equal(a, b) for a, b in zip(f_back, f_again))
