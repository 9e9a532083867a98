"""The fused first half of a TransformerBlock (csrc/tblock.hip) against plain fp32 torch of the same chain
(R/models/saliency_decoder/transformer.py:150-152, attention.py:36-47,87-110) and against the unfused HIP kernels."""
import pytest
import torch
import torch.nn.functional as F

from oracle import salunet_oracle as orc

pytestmark = pytest.mark.gpu
DEV = "cuda"
DT = {"fp32": torch.float32, "bf16": torch.bfloat16, "fp16": torch.float16}
# relative to the output maximum; 16-bit: the storage rounding of xn, q_in, q, P and o (tests/test_gpu_lowp.py: OP_RTOL x 3,
# the bar of the other fused 16-bit kernel)
TOL = {"fp32": 2e-5, "bf16": 1.8e-2, "fp16": 2.4e-3}


@pytest.fixture(scope="module")
def ops():
    from diff_sal_amd import ops as o

    return o


def rnd(name, *shape, scale=1.0):
    return orc.synth_tensor(name, shape, scale)


def reference(x, k, v, p, heads, with_proj, rt=lambda t: t):
    """x [N,H,W,C]; k, v [N,Lk,C]; rt rounds to the storage type where the unfused 16-bit path stores a tensor."""
    N, H, W, C = x.shape
    xn = rt(F.layer_norm(x, (C,), p["g1"], p["b1"], 1e-5))
    qd = F.conv2d(xn.permute(0, 3, 1, 2), p["w9"].t().reshape(C, 1, 3, 3), None, padding=1, groups=C).permute(0, 2, 3, 1)
    qin = rt(F.layer_norm(qd, (C,), p["gq"], p["bq"], 1e-5)).reshape(N, H * W, C)
    q = rt(F.linear(qin, p["wq"], p["bias_q"]))
    d = C // heads
    qh = q.view(N, H * W, heads, d).transpose(1, 2)
    kh = k.view(N, -1, heads, d).transpose(1, 2)
    vh = v.view(N, -1, heads, d).transpose(1, 2)
    s = (qh @ kh.transpose(-1, -2)) * float(C) ** -0.5          # the scale uses the full C (quirk Q6)
    o = (s.softmax(-1) @ vh).transpose(1, 2).reshape(N, H, W, C)
    if not with_proj:
        return o
    return x + F.linear(rt(o), p["wp"], p["bias_p"])


def params(C, tag):
    return dict(g1=rnd(tag + "g1", C, scale=0.1) + 1, b1=rnd(tag + "b1", C, scale=0.1), w9=rnd(tag + "w9", 9, C, scale=0.4),
                gq=rnd(tag + "gq", C, scale=0.1) + 1, bq=rnd(tag + "bq", C, scale=0.1),
                wq=rnd(tag + "wq", C, C, scale=C ** -0.5), bias_q=rnd(tag + "biq", C, scale=0.1),
                wp=rnd(tag + "wp", C, C, scale=C ** -0.5), bias_p=rnd(tag + "bip", C, scale=0.1))


@pytest.mark.parametrize("dname", list(DT))
@pytest.mark.parametrize("shape", [(3, 16, 32, 18), (2, 8, 16, 2), (2, 13, 21, 18), (36, 56, 96, 18), (1, 5, 7, 32),
                                   (3, 16, 24, 18, 192), (36, 28, 48, 18, 192), (1, 9, 11, 7, 192)])
def test_block_front_matches_torch(ops, shape, dname):
    N, H, W, Lk = shape[:4]
    C, heads, dt = (shape[4] if len(shape) > 4 else 96), 2, DT[dname]
    if C == 192 and dt == torch.float32:
        pytest.skip("C = 192 is built for 16-bit storage only (two fp32 192 x 192 weights do not fit the LDS)")
    p = params(C, f"bf{H}")
    x = rnd(f"bfx{H}", N, H, W, C) * 1.5 + 0.2
    k, v = rnd(f"bfk{H}", N, Lk, C, scale=1.2), rnd(f"bfv{H}", N, Lk, C)
    rt = (lambda t: t) if dt == torch.float32 else (lambda t: t.to(dt).float())
    xs, ks, vs = rt(x), rt(k), rt(v)
    pr = dict(p, wq=rt(p["wq"]), wp=rt(p["wp"]))
    ref = reference(xs, ks, vs, pr, heads, with_proj=dt == torch.float32, rt=rt)
    d = lambda t: t.to(DEV)
    got = ops.block_front(d(x).to(dt), d(k).to(dt), d(v).to(dt), (d(p["g1"]), d(p["b1"]), 1e-5), d(p["w9"]),
                          (d(p["gq"]), d(p["bq"]), 1e-5), (d(p["wq"]).to(dt), d(p["bias_q"])),
                          (d(p["wp"]).to(dt), d(p["bias_p"])) if dt == torch.float32 else None, heads, float(C) ** -0.5)
    torch.cuda.synchronize()
    assert got.shape == ref.shape and got.dtype == dt
    err = (got.float().cpu() - ref).abs().max().item() / ref.abs().max().item()
    print(f"block_front {dname} {shape}: rel err {err:.3e}")
    assert err < TOL[dname]


def test_block_front_equals_unfused_hip_kernels(ops):
    """fp32: the fused launch against layernorm -> qkv_prep (query branch) -> linear -> attention -> linear(+residual)."""
    N, H, W, Lk, C, heads = 4, 24, 40, 18, 96, 2
    p = {k_: v_.to(DEV) for k_, v_ in params(C, "bu").items()}
    x = (rnd("bux", N, H, W, C) * 1.5 + 0.2).to(DEV)
    k, v = rnd("buk", N, Lk, C, scale=1.2).to(DEV), rnd("buv", N, Lk, C).to(DEV)
    xn = ops.layernorm(x, p["g1"], p["b1"], 1e-5)
    q = ops.dwconv3_ln(xn, p["w9"], p["gq"], p["bq"], 1e-5)
    q = ops.linear(q, p["wq"], p["bias_q"])
    o = ops.attention(q, k, v, heads, float(C) ** -0.5)
    x1 = ops.linear(o, p["wp"], p["bias_p"], residual=x.view(N, H * W, C)).view(N, H, W, C)
    got = ops.block_front(x, k, v, (p["g1"], p["b1"], 1e-5), p["w9"], (p["gq"], p["bq"], 1e-5), (p["wq"], p["bias_q"]),
                          (p["wp"], p["bias_p"]), heads, float(C) ** -0.5)
    err = (got - x1).abs().max().item() / x1.abs().max().item()
    print("block_front vs unfused HIP: rel err", err)
    assert err < 1e-5


@pytest.mark.parametrize("dname", list(DT))
@pytest.mark.parametrize("C,kk", [(96, 4), (192, 8), (64, 4)])
def test_kv_prep_equals_the_pooled_branch_of_qkv_prep(ops, dname, C, kk):
    """out_q = NULL runs only the pooled key / value workgroups (with the folded first LayerNorm).  Same arithmetic as the full
    launch; for C = 96 / 192 / 384 / 768 a wider lane mapping (three pieces per lane), so equal up to summation order."""
    N, H, W = 3, 16, 32
    dt = DT[dname]
    x = (rnd("kvx", N, H, W, C) * 1.5 + 0.2).to(DEV).to(dt)
    d = lambda name, *s, **kw: rnd(name, *s, **kw).to(DEV)
    w9, wk, wv = d("kvw9", 9, C, scale=0.4), d("kvwk", kk * kk, C, scale=0.3), d("kvwv", kk * kk, C, scale=0.3)
    g = [d(f"kvg{i}", C, scale=0.1) + (1 if i % 2 == 0 else 0) for i in range(8)]
    pre = (g[6], g[7], 1e-5, True)
    q, k1, v1 = ops.qkv_prep(x, w9, g[0], g[1], x, x, wk, wv, g[2], g[3], g[4], g[5], kk, 1e-5, pre_ln=pre)
    k2, v2 = ops.kv_prep(x, x, wk, wv, g[2], g[3], g[4], g[5], kk, 1e-5, pre_ln=pre)
    tol = 2e-6 if dt == torch.float32 else (8e-3 if dt == torch.bfloat16 else 1e-3)     # 16-bit: one rounding step of the output
    for a, b in ((k1, k2), (v1, v2)):
        assert (a.float() - b.float()).abs().max().item() <= tol * b.float().abs().max().item()


@pytest.mark.parametrize("dname", list(DT))
@pytest.mark.parametrize("C,kk,av", [(96, 4, False), (192, 8, True)])
def test_kv_prep_proj_equals_kv_prep_then_linear_pair(ops, dname, C, kk, av):
    """proj_k / proj_v folded into the pooled launch == kv_prep followed by linear_pair (same rounding points, summation order aside);
    visual-only (key from the same normalised tensor) and audio-visual (key from another, un-normalised tensor)."""
    N, H, W = 3, 16, 32
    dt = DT[dname]
    d = lambda name, *s, **kw: rnd(name, *s, **kw).to(DEV)
    x = (d("kpx", N, H, W, C) * 1.5 + 0.2).to(dt)
    xk = (d("kpa", N, H, W, C) * 0.7).to(dt) if av else x
    wk, wv = d("kpwk", kk * kk, C, scale=0.3), d("kpwv", kk * kk, C, scale=0.3)
    g = [d(f"kpg{i}", C, scale=0.1) + (1 if i % 2 == 0 else 0) for i in range(6)]
    pre = (g[4], g[5], 1e-5, not av)
    lk, lv = (d("kplk", C, C, scale=C ** -0.5).to(dt), d("kpbk", C, scale=0.1)), (d("kplv", C, C, scale=C ** -0.5).to(dt), d("kpbv", C, scale=0.1))
    k0, v0 = ops.kv_prep(xk, x, wk, wv, g[0], g[1], g[2], g[3], kk, 1e-5, pre_ln=pre)
    k1, v1 = ops.linear_pair(k0, v0, lk[0], lv[0], lk[1], lv[1])
    k2, v2 = ops.kv_prep_proj(xk, x, wk, wv, g[0], g[1], g[2], g[3], kk, 1e-5, pre, lk, lv)
    tol = 2e-6 if dt == torch.float32 else (8e-3 if dt == torch.bfloat16 else 1e-3)
    for a, b in ((k1, k2), (v1, v2)):
        assert a.shape == b.shape
        assert (a.float() - b.float()).abs().max().item() <= tol * a.float().abs().max().item()
