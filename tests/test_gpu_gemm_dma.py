"""The LDS-DMA plain-product kernel (csrc/gemm_dma.hip) against plain fp32 / fp64 PyTorch of the same operator, through the
C ABI (diffsal_conv_igemm): every tile configuration, every epilogue term, ragged M / N (the buffer range check cuts the DMA),
multi-tile persistent walks, the XCD tile order, and the planner's automatic choice.

Tolerance 2e-5 of the output maximum (the kernel is an exact-fp32 fmaf chain; its k order inside a 32-wide slice differs from
igemm_kernel's, so it is compared with the reference, not bit for bit with that kernel).
"""
import pytest
import torch
import torch.nn.functional as F

from oracle import salunet_oracle as orc

pytestmark = pytest.mark.gpu

DEV = "cuda"
TOL = 2e-5


def rel_err(got, ref):
    ref = ref.double().cpu()
    return (got.double().cpu() - ref).abs().max().item() / (ref.abs().max().item() + 1e-12)


def rnd(name, *shape, scale=1.0):
    return orc.synth_tensor(name, shape, scale)


@pytest.fixture(scope="module")
def ops():
    from diff_sal_amd import ops as o

    assert torch.cuda.is_available()
    return o


def reference(x, w, b, act, r, scale=None, shift=None, rowvec=None, rows_per_img=1):
    y = x.double() @ w.double().t()
    if b is not None:
        y = y + b.double()
    if scale is not None:
        y = y * scale.double() + shift.double()
    if rowvec is not None:
        y = y + rowvec.double().repeat_interleave(rows_per_img, dim=0)[: y.shape[0]]
    if act == 1:
        y = torch.relu(y)
    elif act == 2:
        y = F.gelu(y)
    elif act == 3:
        y = torch.sigmoid(y)
    if r is not None:
        y = y + r.double()
    return y


# cfg (DIFFSAL_GEMM_DMA value): 1 = 96x96 / 3 stages, 2 = 96x96 / 6 stages;
# K / 32 must be a multiple of the stage count (otherwise the library falls back to the tiled kernel -- also a valid result)
@pytest.mark.parametrize("cfg", [1, 2])
@pytest.mark.parametrize("M,K,N", [(3024, 768, 768), (1000, 384, 200), (96, 96, 96), (97, 192, 100), (5000, 1152, 388),
                                   (50011, 192, 192)])
def test_gemm_dma_every_tile_configuration(ops, tuning, cfg, M, K, N):
    tuning.set("DIFFSAL_GEMM_DMA", cfg)
    x, w = rnd("dx%d_%d" % (M, K), M, K).to(DEV), rnd("dw%d_%d" % (N, K), N, K, scale=K ** -0.5).to(DEV)
    b, r = rnd("db%d" % N, N, scale=0.3).to(DEV), rnd("dr%d_%d" % (M, N), M, N).to(DEV)
    for act, bias, res in ((0, None, None), (0, b, r), (2, b, None), (1, b, r), (3, None, r)):
        y = ops.linear(x, w, bias, act=act, residual=res)
        assert rel_err(y, reference(x, w, bias, act, res)) < TOL, (cfg, M, K, N, act)


@pytest.mark.parametrize("cfg", [1, 2])
def test_gemm_dma_affine_rowvec_epilogue(ops, tuning, cfg):
    """BN affine (scale / shift) and the per-image vector (rowvec, rows_per_img = Ho * Wo) as the 1x1 convolutions use them."""
    tuning.set("DIFFSAL_GEMM_DMA", cfg)
    Nimg, H, W, Cin, Cout = 5, 9, 22, 288, 200
    x = rnd("ax", Nimg, H, W, Cin).to(DEV)
    w = rnd("aw", Cout, Cin, scale=Cin ** -0.5).to(DEV)
    b, sc, sh = rnd("ab", Cout, scale=0.2).to(DEV), (1.0 + 0.3 * rnd("as", Cout)).to(DEV), rnd("ah", Cout, scale=0.2).to(DEV)
    rv = rnd("av", Nimg, Cout, scale=0.5).to(DEV)
    res = rnd("ar", Nimg, H, W, Cout).to(DEV)
    y = ops.conv_igemm(x, w, bias=b, scale=sc, shift=sh, rowvec=rv, residual=res, act=ops.ACT_RELU)
    ref = reference(x.reshape(-1, Cin), w, b, 1, res.reshape(-1, Cout), sc, sh, rv, H * W).reshape(Nimg, H, W, Cout)
    assert rel_err(y, ref) < TOL


def test_gemm_dma_zero_fill_past_the_last_row(ops, tuning):
    """Rows past M / N never reach memory (range-checked DMA): NaN-poisoned neighbours of the operands must not leak, and
    memory behind the output stays untouched."""
    tuning.set("DIFFSAL_GEMM_DMA", 1)
    M, K, N = 200, 96, 100
    big_x = torch.full((M + 96, K), float("nan"), device=DEV)
    big_w = torch.full((N + 96, K), float("nan"), device=DEV)
    big_x[:M] = rnd("zx", M, K).to(DEV)
    big_w[:N] = rnd("zw", N, K, scale=K ** -0.5).to(DEV)
    out = torch.full((M + 8, N), 7.0, device=DEV)
    from diff_sal_amd.ops import conv_igemm
    y = conv_igemm(big_x[:M].reshape(1, 1, M, K), big_w[:N], out=out[:M].reshape(1, 1, M, N))
    assert torch.isfinite(y).all()
    assert rel_err(y.reshape(M, N), reference(big_x[:M], big_w[:N], None, 0, None)) < TOL
    assert (out[M:] == 7.0).all()


def test_gemm_dma_xcd_order_and_plain_order_agree(ops, tuning):
    """Large launches walk the tiles XCD by XCD: every tile exactly once -- bit-equal with the plain order."""
    tuning.set("DIFFSAL_GEMM_DMA", 1)
    M, K, N = 70001, 96, 488
    x, w, b = rnd("xx", M, K).to(DEV), rnd("xw", N, K, scale=K ** -0.5).to(DEV), rnd("xb", N, scale=0.1).to(DEV)
    tuning.set("DIFFSAL_NO_XCD_ORDER", 1)
    plain = ops.linear(x, w, b, act=ops.ACT_GELU)
    tuning.set("DIFFSAL_NO_XCD_ORDER", 0)
    xcd = ops.linear(x, w, b, act=ops.ACT_GELU)
    assert torch.equal(plain, xcd)
    assert rel_err(xcd, reference(x, w, b, 2, None)) < TOL


def test_planner_picks_dma_kernel_for_token_gemms_and_results_agree(ops, tuning):
    """Default planner (DMA kernel from ~192 tiles on, K % 96 == 0) against the tiled kernels (DIFFSAL_GEMM_DMA=0) on the
    decoder's token-GEMM shapes: both within 2e-5 of fp64, and within 1e-6 of each other."""
    for M, K, N in ((3024, 768, 768), (12096, 384, 768), (3024, 768, 3456)):
        x, w, b = rnd("tx%d" % K, M, K).to(DEV), rnd("tw%d_%d" % (N, K), N, K, scale=K ** -0.5).to(DEV), rnd("tb%d" % N, N, scale=0.1).to(DEV)
        r = rnd("tr%d_%d" % (M, N), M, N).to(DEV)
        tuning.set("DIFFSAL_GEMM_DMA", 0)
        tiled = ops.linear(x, w, b, residual=r)
        tuning.set("DIFFSAL_GEMM_DMA", None)
        auto = ops.linear(x, w, b, residual=r)
        ref = reference(x, w, b, 0, r)
        assert rel_err(tiled, ref) < TOL and rel_err(auto, ref) < TOL
        assert rel_err(auto, tiled) < 2e-6


def test_gemm_dma_is_deterministic(ops, tuning):
    tuning.set("DIFFSAL_GEMM_DMA", 1)
    x, w = rnd("ex", 12096, 384).to(DEV), rnd("ew", 384, 384, scale=0.05).to(DEV)
    a = ops.linear(x, w)
    for _ in range(5):
        assert torch.equal(a, ops.linear(x, w))


# ---- convolution mode (DIFFSAL_CONV_DMA): tap displacement per K slice, padding = out-of-range DMA lanes (zeros), split-K ----
CONV_CASES = [
    # N, H, W, Cin, Cout, kh, kw, stride, pad, dil, asym
    (2, 14, 24, 96, 192, 3, 3, 1, 1, 1, False),     # ResnetBlock conv family, zero padding on all four sides
    (2, 14, 24, 192, 192, 3, 3, 2, 0, 1, True),     # Downsample: pad (0,1,0,1), stride 2
    (1, 30, 46, 96, 96, 3, 3, 4, 0, 1, True),       # Downsample4x4 (odd sizes exercise the bounds)
    (3, 14, 24, 384, 192, 3, 3, 1, 2, 2, False),    # UpEmbed dilated conv
    (5, 13, 19, 96, 100, 3, 3, 1, 1, 1, False),     # ragged rows / channels
    (4, 7, 12, 768, 768, 3, 3, 1, 1, 1, False),     # few rows, long K: split-K units + the slab sum
    (1, 1, 40, 96, 768, 1, 1, 1, 0, 1, False),      # a 1x1 routed through the convolution form when forced
]


@pytest.mark.parametrize("cfg", [1, 2])
@pytest.mark.parametrize("case", CONV_CASES)
def test_conv_dma_matches_conv2d(ops, tuning, cfg, case):
    N, H, W, Cin, Cout, kh, kw, s, p, d, asym = case
    tuning.set("DIFFSAL_CONV_DMA", cfg)
    tuning.set("DIFFSAL_GEMM_DMA", cfg)
    x = rnd("vx%d%d" % (Cin, Cout), N, Cin, H, W)
    w = rnd("vw%d%d" % (Cin, Cout), Cout, Cin, kh, kw, scale=(Cin * kh * kw) ** -0.5)
    b = rnd("vb", Cout, scale=0.1)
    if asym:
        ref = F.conv2d(F.pad(x, (0, 1, 0, 1)).double(), w.double(), b.double(), stride=s)
        kwargs = dict(stride=(s, s), pad=(0, 0), out_hw=ref.shape[-2:])
    else:
        ref = F.conv2d(x.double(), w.double(), b.double(), stride=s, padding=p, dilation=d)
        kwargs = dict(stride=(s, s), pad=(p, p), dil=(d, d))
    res = rnd("vr", *ref.shape)
    ref = torch.relu(ref) + res.double()
    from diff_sal_amd.ops import pack_conv_weight
    got = ops.conv_igemm(x.permute(0, 2, 3, 1).contiguous().to(DEV), pack_conv_weight(w.to(DEV)), kh=kh, kw=kw, bias=b.to(DEV),
                         residual=res.permute(0, 2, 3, 1).contiguous().to(DEV), act=ops.ACT_RELU, **kwargs)
    assert got.shape == ref.permute(0, 2, 3, 1).shape
    assert rel_err(got, ref.permute(0, 2, 3, 1)) < TOL, case


def test_conv_dma_reduce_temp_form(ops, tuning):
    """ReduceTemp (R/models/saliency_decoder/common_block.py:150-173): Conv3d (5,1,1) stride 5 over 9 frames = a 5x1 stride-(5,1)
    convolution of the [B, 9, HW, C] view; only frames 0-4 reach the output."""
    tuning.set("DIFFSAL_CONV_DMA", 1)
    B, T, HW, C, Co = 3, 9, 200, 96, 192
    x = rnd("rx", B, T, HW, C).to(DEV)
    w = rnd("rw", Co, C, 5, 1, scale=(5 * C) ** -0.5).to(DEV)
    from diff_sal_amd.ops import pack_conv_weight
    y = ops.conv_igemm(x, pack_conv_weight(w), kh=5, kw=1, stride=(5, 1), out_hw=(1, HW), act=ops.ACT_RELU)
    ref = torch.relu(torch.einsum("bthc,octz->bho", x[:, :5].double(), w.double())).reshape(B, 1, HW, Co)
    assert rel_err(y, ref) < TOL


def test_conv_dma_planner_rule_agrees_with_tiled_kernel(ops, tuning):
    """Shapes the planner routes to the DMA form by itself (many tiles x short K; few rows x long K) against the tiled kernel."""
    from diff_sal_amd.ops import pack_conv_weight
    for N, H, W, Cin, Cout, st in ((36, 56, 96, 96, 96, 1), (4, 14, 24, 768, 768, 2)):
        x = torch.relu(rnd("px%d" % Cin, N, H, W, Cin)).to(DEV)
        w = rnd("pw%d" % Cin, Cout, Cin, 3, 3, scale=(9 * Cin) ** -0.5).to(DEV)
        kwargs = dict(kh=3, kw=3, stride=(st, st), pad=(2, 2) if st == 1 else (0, 0), dil=(2, 2) if st == 1 else (1, 1),
                      out_hw=(H, W) if st == 1 else (H // 2, W // 2))
        tuning.set("DIFFSAL_CONV_DMA", 0)
        tiled = ops.conv_igemm(x, pack_conv_weight(w), **kwargs)
        tuning.set("DIFFSAL_CONV_DMA", None)
        auto = ops.conv_igemm(x, pack_conv_weight(w), **kwargs)
        assert rel_err(auto, tiled) < 3e-6


def test_conv_igemm_group_equals_single_launches(ops, tuning):
    """diffsal_conv_igemm_group: four ReduceTemp-shaped products (and a mix of plain products) in one launch == the same
    problems one by one (within fp32 summation order), and a group with a problem the
    grouped kernel cannot take (K not a multiple of 96) falls back to single launches."""
    from diff_sal_amd.ops import pack_conv_weight
    B, T = 2, 9
    probs = []
    for i, (HW, C) in enumerate(((40, 768), (170, 384), (650, 192), (2600, 96))):
        x = rnd("gx%d" % i, B, T, HW, C).to(DEV)
        w = pack_conv_weight(rnd("gw%d" % i, 768, C, 5, 1, scale=(5 * C) ** -0.5).to(DEV))
        probs.append(dict(x=x, w=w, kh=5, kw=1, stride=(5, 1), out_hw=(1, HW), act=ops.ACT_RELU))
    tuning.set("DIFFSAL_CONV_DMA", 1)
    singles = [ops.conv_igemm(p["x"], p["w"], kh=5, kw=1, stride=(5, 1), out_hw=p["out_hw"], act=ops.ACT_RELU) for p in probs]
    grouped = ops.conv_igemm_group(probs)
    for s_, g_ in zip(singles, grouped):       # the single launches may split K (few rows, long K): same sums, other order
        assert rel_err(g_, s_) < 3e-6
    # plain products with biases, different shapes, written into caller-provided outputs
    lin = []
    for i, (M, K, N) in enumerate(((648, 768, 768), (648, 384, 384), (1000, 96, 200))):
        x = rnd("hx%d" % i, 1, 1, M, K).to(DEV)
        w = rnd("hw%d" % i, N, K, scale=K ** -0.5).to(DEV)
        b = rnd("hb%d" % i, N, scale=0.2).to(DEV)
        lin.append(dict(x=x, w=w, bias=b, out=torch.full((1, 1, M, N), 3.0, device=DEV)))
    outs = ops.conv_igemm_group(lin)
    for p, o in zip(lin, outs):
        assert o.data_ptr() == p["out"].data_ptr()
        assert rel_err(o.reshape(o.shape[2], -1), reference(p["x"].reshape(p["x"].shape[2], -1), p["w"], p["bias"], 0, None)) < TOL
    # fallback: K = 160 is not a multiple of 96
    odd = [dict(x=rnd("ox", 1, 1, 500, 160).to(DEV), w=rnd("ow", 64, 160, scale=0.1).to(DEV)), lin[1]]
    outs = ops.conv_igemm_group(odd)
    assert rel_err(outs[0].reshape(500, 64), reference(odd[0]["x"].reshape(500, 160), odd[0]["w"], None, 0, None)) < TOL


@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("M,K,N", [(3024, 768, 768), (1000, 384, 200), (97, 192, 100), (50011, 192, 192), (648, 1536, 768)])
def test_gemm_dma_16bit_storage(ops, tuning, dt, M, K, N):
    """The same kernel on bf16 / fp16 storage (64-element K slices, v_mfma_f32_16x16x32, fp32 accumulation, one rounding of the
    output): against fp64 of the SAME rounded operands within the 16-bit operator tolerance of tests/test_gpu_lowp.py, and
    against the tiled 16-bit kernel (same products, other summation order)."""
    tol = 6e-3 if dt == torch.bfloat16 else 8e-4
    x = rnd("sx%d_%d" % (M, K), M, K).to(DEV).to(dt)
    w = rnd("sw%d_%d" % (N, K), N, K, scale=K ** -0.5).to(DEV).to(dt)
    b = rnd("sb%d" % N, N, scale=0.3).to(DEV)
    r = rnd("sr%d_%d" % (M, N), M, N).to(DEV).to(dt)
    for act, bias, res in ((0, None, None), (2, b, None), (0, b, r)):
        tuning.set("DIFFSAL_GEMM_DMA16", 1)
        y = ops.linear(x, w, bias, act=act, residual=res)
        tuning.set("DIFFSAL_GEMM_DMA16", 0)
        y0 = ops.linear(x, w, bias, act=act, residual=res)
        ref = reference(x.float(), w.float(), bias, act, None if res is None else res.float())
        assert y.dtype == dt
        assert rel_err(y.float(), ref) < tol, (dt, M, K, N, act)
        assert rel_err(y.float(), y0.float()) < tol
