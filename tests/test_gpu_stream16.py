"""16-byte forms of the HBM-bound kernels on 16-bit storage (BASELINE configs[1] "bf16", configs[4] "fp16").

Each kernel here replaces a 4-channel (8-byte) form at the decoder's shapes.  Two checks per kernel: against the 8-byte form it
replaces (``DIFFSAL_NO_STREAM16=1`` routes a call back to it) -- bit for bit where the arithmetic order is unchanged -- and
against an fp32 PyTorch evaluation of the same 16-bit-rounded inputs within the per-operator bar of tests/test_gpu_lowp.py.
"""
import pytest
import torch

from oracle import salunet_oracle as orc

pytestmark = pytest.mark.gpu
DEV = "cuda"
DTYPES = {"bf16": torch.bfloat16, "fp16": torch.float16}
OP_RTOL = {"bf16": 6e-3, "fp16": 8e-4}


def rnd(name, *shape, scale=1.0):
    return orc.synth_tensor(name, shape, scale)


def rel_err(got, ref):
    ref = ref.float().cpu()
    return (got.float().cpu() - ref).abs().max().item() / (ref.abs().max().item() + 1e-12)


@pytest.fixture(scope="module")
def ops():
    from diff_sal_amd import ops as o

    assert torch.cuda.is_available()
    return o


class old_forms:
    """with old_forms(): the calls inside take the 8-byte forms"""

    def __enter__(self):
        from diff_sal_amd import _lib

        _lib.set_tuning("DIFFSAL_NO_STREAM16", 1)

    def __exit__(self, *a):
        from diff_sal_amd import _lib

        _lib.set_tuning("DIFFSAL_NO_STREAM16", None)


# the decoder's four stages at 224 x 384 (H, W, C, up-sampling factor of the 7 x 12 audio map), and two odd ones
FUSE_CASES = [(7, 12, 768, 7, 12), (14, 24, 384, 7, 12), (28, 48, 192, 7, 12), (56, 96, 96, 7, 12), (8, 16, 64, 2, 4), (6, 8, 128, 3, 4)]


@pytest.mark.parametrize("dname", list(DTYPES))
@pytest.mark.parametrize("H,W,C,h,w", FUSE_CASES)
def test_audio_fuse_16_byte_form(ops, dname, H, W, C, h, w):
    """K7 (R/models/saliency_decoder/transformer.py:133-146): same arithmetic order as the 8-byte form -> identical bits; the
    audio rows sit inside a wider row (the stages' align products side by side) as in the shipped step."""
    dt = DTYPES[dname]
    B, T = 2, 9
    ld = C + 96
    a_wide = rnd("s16a", B * T, h * w, ld).to(DEV).to(dt)
    a_small = a_wide[:, :, 64:64 + C]            # a 16-byte aligned channel slice of the wide rows
    x = rnd("s16x", B, T, H, W, C).to(DEV).to(dt)
    got = ops.audio_fuse(a_small, x, h, w)
    with old_forms():
        old = ops.audio_fuse(a_small, x, h, w)
    assert torch.equal(got, old)
    up = H // h
    a_up = a_small.float().view(B, T, h, w, C).repeat_interleave(up, 2).repeat_interleave(up, 3)
    m = torch.softmax((a_up * x.float()).mean(1), dim=2)
    ref = (a_up * m[:, None]).permute(0, 4, 1, 2, 3)
    assert rel_err(got, ref) < OP_RTOL[dname]


def _ln(x, g, b, eps):
    return torch.nn.functional.layer_norm(x, (x.shape[-1],), g, b, eps)


# (H, W, C, k): the four decoder stages (k = kernel_kv: 18 pooled keys per frame) and a ragged one
PREP_CASES = [(7, 12, 768, 2), (14, 24, 384, 4), (28, 48, 192, 8), (56, 96, 96, 16), (9, 10, 96, 3)]


@pytest.mark.parametrize("dname", list(DTYPES))
@pytest.mark.parametrize("H,W,C,k", PREP_CASES)
@pytest.mark.parametrize("mode", ["qkv", "kv_preln_vis", "kv_preln_av"])
def test_qkv_prep_16_byte_form(ops, dname, H, W, C, k, mode):
    """K9 (R/models/saliency_decoder/attention.py:36-76,88-95; the block's norm, transformer.py:150): depthwise 3x3 + LayerNorm on
    every token, depthwise k x k stride-k pooling + LayerNorm of key / value; the pooled branch alone with the block's first
    LayerNorm applied as the tokens are loaded (the form behind block_front)."""
    dt = DTYPES[dname]
    N = 3
    tol = OP_RTOL[dname]
    x = rnd("p16x", N, H, W, C).to(dt)
    xk = rnd("p16k", N, H, W, C).to(dt)
    w9 = rnd("p16w9", 9, C, scale=0.3)
    wk, wv = rnd("p16wk", k * k, C, scale=1.0 / k), rnd("p16wv", k * k, C, scale=1.0 / k)
    gq, bq = 1 + rnd("p16gq", C, scale=0.1), rnd("p16bq", C, scale=0.1)
    gk, bk = 1 + rnd("p16gk", C, scale=0.1), rnd("p16bk", C, scale=0.1)
    gv, bv = 1 + rnd("p16gv", C, scale=0.1), rnd("p16bv", C, scale=0.1)
    pg, pb = 1 + rnd("p16pg", C, scale=0.1), rnd("p16pb", C, scale=0.1)
    d = lambda t: t.to(DEV)
    gh, gw = (H - k) // k + 1, (W - k) // k + 1

    def pooled(src, w, g, b):          # [N,H,W,C] fp32 -> [N, gh*gw, C]
        win = src[:, :gh * k, :gw * k].reshape(N, gh, k, gw, k, C).permute(0, 1, 3, 2, 4, 5).reshape(N, gh * gw, k * k, C)
        return _ln((win * w).sum(2), g, b, 1e-5)

    if mode == "qkv":
        def run():
            return ops.qkv_prep(d(x), d(w9), d(gq), d(bq), d(xk), d(x), d(wk), d(wv), d(gk), d(bk), d(gv), d(bv), k, 1e-5)
        got = run()
        with old_forms():
            old = run()
        xf = x.float().permute(0, 3, 1, 2)
        qc = torch.nn.functional.conv2d(xf, w9.t().reshape(C, 1, 3, 3), padding=1, groups=C).permute(0, 2, 3, 1)
        ref = (_ln(qc, gq, bq, 1e-5).reshape(N, H * W, C), pooled(xk.float(), wk, gk, bk), pooled(x.float(), wv, gv, bv))
    else:
        av = mode == "kv_preln_av"
        src_k = xk if av else x

        def run():
            return ops.kv_prep(d(src_k), d(x), d(wk), d(wv), d(gk), d(bk), d(gv), d(bv), k, 1e-5, pre_ln=(d(pg), d(pb), 1e-6, not av))
        got = run()
        with old_forms():
            old = run()
        xn = _ln(x.float(), pg, pb, 1e-6).to(dt).float()
        ref = (pooled(src_k.float() if av else xn, wk, gk, bk), pooled(xn, wv, gv, bv))
    for g_, o_, r_ in zip(got, old, ref):
        assert g_.dtype == dt and g_.shape == o_.shape == r_.shape
        assert rel_err(g_, o_.float()) < tol           # the 8-byte form: same formulas, another summation order, one rounding
        assert rel_err(g_, r_) < 2 * tol               # pre-LN rounds the normalised tokens to the storage type first


@pytest.mark.parametrize("dname", list(DTYPES))
@pytest.mark.parametrize("h,w,C", [(7, 12, 384), (14, 24, 192), (28, 48, 96), (5, 9, 40)])
def test_up2_commute_interior_16_byte_form(ops, dname, h, w, C):
    """K12 (R/models/saliency_decoder/common_block.py:196-206), the interpolation kernel of the source-resolution form: 8 channels per
    item instead of 4, the same arithmetic per element -> identical bits."""
    from diff_sal_amd import _lib

    dt = DTYPES[dname]
    N = 3
    lib = _lib.load()
    c_ext = rnd("c16c", N, h + 2, w + 2, C).to(DEV).to(dt)
    tb = rnd("c16t", N, 2 * w + 2 * h - 4, 9 * C).to(DEV).to(dt)
    scale, shift = (1 + rnd("c16s", C, scale=0.1)).to(DEV), rnd("c16h", C, scale=0.1).to(DEV)
    code = ops.DTYPE_CODES[dt]

    def run():
        out = torch.zeros((N, 2 * h, 2 * w, C), device=DEV, dtype=dt)
        _lib.check(lib.diffsal_up2_conv_commute(c_ext.data_ptr(), tb.data_ptr(), scale.data_ptr(), shift.data_ptr(), out.data_ptr(),
                                                N, h, w, C, 1, code, torch.cuda.current_stream().cuda_stream), "up2_conv_commute")
        return out
    got = run()
    with old_forms():
        old = run()
    assert torch.equal(got, old)


# N, H, W, Cin, Cout, pad, dil, epilogue
CONV_DMA_CASES = [
    (2, 56, 96, 96, 96, 2, 2, "bn_relu_res"),     # stage-3 UpEmbed conv2: 8 x 32 tiles, one N tile
    (3, 28, 48, 192, 192, 2, 2, "bn_relu_res"),   # stage 2: 16 x 16 tiles, two N tiles
    (2, 14, 24, 384, 192, 2, 1, "bn_relu"),       # UpEmbed conv1 at the source resolution on the extended grid (out 16 x 26)
    (2, 56, 96, 96, 192, 1, 1, "bias_rowvec"),    # ResnetBlock conv1
    (2, 9, 17, 32, 72, 1, 1, "bias"),             # ragged: partial tiles in both directions, Cout < 96 and not a multiple of 32
    (1, 20, 36, 768, 96, 1, 1, "none"),           # 24 chunks
    (5, 7, 12, 64, 96, 2, 1, "bn_relu_res"),      # 9 x 14 extended grid: two whole images per tile, odd image count
    (4, 6, 10, 32, 200, 1, 1, "bias_rowvec"),     # 60-pixel maps, a per-image vector, a partial N tile
]


@pytest.mark.parametrize("tile,waves", [(0, 0), (0, 1), (1, 0)], ids=["256x96", "128x96", "192x192"])
@pytest.mark.parametrize("dname", list(DTYPES))
@pytest.mark.parametrize("case", CONV_DMA_CASES)
def test_conv16_dma_halo_kernel(ops, dname, case, tile, waves):
    """csrc/conv16_dma.hip (R/models/saliency_decoder/common_block.py:196-216, sal_unet.py:104-142 on 16-bit storage): forced on
    shapes of a few tiles, against the generic 16-bit implicit-GEMM kernel -- same accumulation order, identical bits -- and
    against F.conv2d on the rounded operands."""
    import torch.nn.functional as F

    from diff_sal_amd import _lib

    dt = DTYPES[dname]
    N, H, W, Cin, Cout, pad, dil, epi = case
    x = rnd("cdx%d" % Cin, N, Cin, H, W).to(dt)
    w = rnd("cdw%d%d" % (Cin, Cout), Cout, Cin, 3, 3, scale=1.0 / (3 * Cin ** 0.5)).to(dt)
    Ho, Wo = H + 2 * pad - 2 * dil, W + 2 * pad - 2 * dil
    bias = rnd("cdb", Cout, scale=0.1) if "bias" in epi else None
    scale = (1 + rnd("cds", Cout, scale=0.2)) if "bn" in epi else None
    shift = rnd("cdh", Cout, scale=0.1) if "bn" in epi else None
    rowvec = rnd("cdr", N, Cout, scale=0.5) if "rowvec" in epi else None
    res = rnd("cdres", N, Cout, Ho, Wo).to(dt) if "res" in epi else None
    act = 1 if "relu" in epi else 0
    dv = lambda t: None if t is None else t.to(DEV)
    xn = x.permute(0, 2, 3, 1).contiguous().to(DEV)
    wp = ops.cast(ops.pack_conv_weight(w.float().to(DEV)), dt)
    rn = None if res is None else res.permute(0, 2, 3, 1).contiguous().to(DEV)

    def run():
        return ops.conv_igemm(xn, wp, kh=3, kw=3, pad=(pad, pad), dil=(dil, dil), out_hw=(Ho, Wo), bias=dv(bias), scale=dv(scale),
                              shift=dv(shift), rowvec=dv(rowvec), residual=rn, act=act)
    _lib.set_tuning("DIFFSAL_FORCE_HALO", 2)
    _lib.set_tuning("DIFFSAL_CONV16_TILE", tile)
    _lib.set_tuning("DIFFSAL_CONV16_HALF", waves)
    try:
        got = run()
        name = _lib.load().diffsal_last_gemm_kernel().decode()
    finally:
        _lib.set_tuning("DIFFSAL_FORCE_HALO", None)
        _lib.set_tuning("DIFFSAL_CONV16_TILE", None)
        _lib.set_tuning("DIFFSAL_CONV16_HALF", None)
    assert "conv16_dma_kernel" in name and ("x 192 channels" in name) == (tile == 1)
    assert ("2 images" in name) == (tile == 0 and Ho * Wo <= 128)
    if "2 images" not in name and tile == 0:
        assert (("4x32 pixels" in name) or ("8x16 pixels" in name)) == (waves == 1)
    # the generic kernel on one fixed tile shape WITHOUT a K split (its split sums the K ranges in another order)
    _lib.set_tuning("DIFFSAL_NO_HALO", 1)
    _lib.set_tuning("DIFFSAL_IGEMM16_CFG", 0)
    try:
        old = run()
        assert "igemm16_kernel" in _lib.load().diffsal_last_gemm_kernel().decode()
    finally:
        _lib.set_tuning("DIFFSAL_NO_HALO", None)
        _lib.set_tuning("DIFFSAL_IGEMM16_CFG", None)
    assert torch.equal(got, old)
    ref = F.conv2d(x.float(), w.float(), bias, padding=pad, dilation=dil)
    if scale is not None:
        ref = ref * scale[None, :, None, None] + shift[None, :, None, None]
    if rowvec is not None:
        ref = ref + rowvec[:, :, None, None]
    if act:
        ref = F.relu(ref)
    if res is not None:
        ref = ref + res.float()
    assert rel_err(got, ref.permute(0, 2, 3, 1)) < OP_RTOL[dname]


# M, K, N, epilogue
GEMM2_CASES = [
    (3024, 768, 768, "bias"),
    (3024, 768, 1536, "bias_gelu"),
    (1000, 1536, 768, "bias_res"),        # partial last row tile
    (5000, 192, 200, "bias_relu_res"),    # N neither a multiple of 96 nor of 32
    (700, 96, 864, "none"),               # three K chunks
    (2000, 768, 864, "f32out"),           # mt_proj's tap products: fp32 out
]


@pytest.mark.parametrize("tile", [3, 4], ids=["256x96", "192x192"])
@pytest.mark.parametrize("dname", list(DTYPES))
@pytest.mark.parametrize("case", GEMM2_CASES)
def test_gemm16_dma2_plain_products(ops, dname, case, tile):
    """csrc/gemm16_dma.hip (R/models/saliency_decoder/attention.py:97-111, common_block.py:125-147 on 16-bit storage), forced on
    small shapes: identical bits to the generic 16-bit kernel without a K split (same MFMA, same K order); fp64 reference."""
    from diff_sal_amd import _lib

    dt = DTYPES[dname]
    M, K, N, epi = case
    x = rnd("g2x%d" % K, M, K).to(DEV).to(dt)
    w = rnd("g2w%d%d" % (K, N), N, K, scale=K ** -0.5).to(DEV).to(dt)
    b = rnd("g2b", N, scale=0.1).to(DEV) if "bias" in epi else None
    res = rnd("g2r", M, N).to(DEV).to(dt) if "res" in epi else None
    act = 2 if "gelu" in epi else (1 if "relu" in epi else 0)
    f32 = epi == "f32out"

    def run():
        return ops.linear(x, w, b, act=act, residual=res, out_f32=f32) if f32 else ops.linear(x, w, b, act=act, residual=res)
    _lib.set_tuning("DIFFSAL_GEMM_DMA16", tile)
    try:
        got = run()
        name = _lib.load().diffsal_last_gemm_kernel().decode()
    finally:
        _lib.set_tuning("DIFFSAL_GEMM_DMA16", None)
    assert "gemm16_dma2_kernel" in name and ("192x192" in name) == (tile == 4)
    ref = x.double() @ w.double().t()
    if b is not None:
        ref = ref + b.double()
    if act == 1:
        ref = torch.relu(ref)
    elif act == 2:
        ref = torch.nn.functional.gelu(ref)
    if res is not None:
        ref = ref + res.double()
    assert got.dtype == (torch.float32 if f32 else dt)
    assert rel_err(got, ref) < (1e-5 if f32 else OP_RTOL[dname])
    if not f32:
        _lib.set_tuning("DIFFSAL_GEMM_DMA16", 0)
        _lib.set_tuning("DIFFSAL_IGEMM16_CFG", 0)
        try:
            old = run()
        finally:
            _lib.set_tuning("DIFFSAL_GEMM_DMA16", None)
            _lib.set_tuning("DIFFSAL_IGEMM16_CFG", None)
        assert torch.equal(got, old)


@pytest.mark.parametrize("dname", list(DTYPES))
@pytest.mark.parametrize("Bn,T,HW,C,kt,Co", [(3, 9, 84, 768, 5, 768), (2, 9, 336, 96, 5, 768), (2, 5, 100, 64, 5, 200)])
@pytest.mark.parametrize("tile", [3, 4], ids=["256x96", "192x192"])
def test_gemm16_dma2_reduce_temp_row_form(ops, dname, Bn, T, HW, C, kt, Co, tile):
    """ReduceTemp (R/models/saliency_decoder/sal_unet.py:300-318) on 16-bit storage through csrc/gemm16_dma.hip's row form: a (kt, 1)
    kernel over the frame axis that leaves one frame, K ordered (chunk, tap, channel) as the packed weight is."""
    from diff_sal_amd import _lib

    dt = DTYPES[dname]
    x = rnd("rt%d" % C, Bn, T, HW, C).to(DEV).to(dt)
    w = rnd("rtw%d" % C, Co, C, kt, 1, scale=(kt * C) ** -0.5).to(DEV)
    wp = ops.cast(ops.pack_conv_weight(w), dt)

    def run():
        return ops.conv_igemm(x, wp, kh=kt, kw=1, stride=(kt, 1), act=1)
    _lib.set_tuning("DIFFSAL_GEMM_DMA16", tile)
    try:
        got = run()
        assert "gemm16_dma2_kernel" in _lib.load().diffsal_last_gemm_kernel().decode()
    finally:
        _lib.set_tuning("DIFFSAL_GEMM_DMA16", None)
    _lib.set_tuning("DIFFSAL_GEMM_DMA16", 0)
    _lib.set_tuning("DIFFSAL_IGEMM16_CFG", 0)
    try:
        old = run()
    finally:
        _lib.set_tuning("DIFFSAL_GEMM_DMA16", None)
        _lib.set_tuning("DIFFSAL_IGEMM16_CFG", None)
    assert got.shape == old.shape == (Bn, 1, HW, Co)
    assert torch.equal(got, old)
    ref = torch.relu(torch.einsum("bthc,octu->bho", x[:, :kt].double(), wp.new_tensor(w.to(dt).double().cpu().numpy(), dtype=torch.float64)))
    assert rel_err(got.view(Bn, HW, Co), ref) < OP_RTOL[dname]


@pytest.mark.parametrize("dname", list(DTYPES))
@pytest.mark.parametrize("n,Lq,Lk,C,heads", [(3, 84, 18, 768, 2), (2, 336, 18, 384, 2), (2, 100, 18, 192, 2), (2, 77, 5, 96, 2), (1, 33, 18, 384, 4),
                                                 (2, 130, 5, 384, 2), (1, 40, 32, 768, 2)])      # matrix-core form: Lk <= 16, Lk = 32
def test_attention_16_byte_form(ops, dname, n, Lq, Lk, C, heads):
    """K11 (R/models/saliency_decoder/attention.py:97-108; scale C^-1/2 of the FULL width, quirk Q6): one workgroup per (image, head),
    K / V in LDS once, two queries per lane group; head dims 192 / 384 on the matrix cores (csrc/attn16_mfma.hip: the probabilities
    enter the second product as a 16-bit hi + lo pair)."""
    import torch.nn.functional as F

    dt = DTYPES[dname]
    q, k, v = (rnd(nm, n, L, C).to(DEV).to(dt) for nm, L in (("a16q", Lq), ("a16k", Lk), ("a16v", Lk)))
    scale = C ** -0.5
    got = ops.attention(q, k, v, heads, scale)
    with old_forms():
        old = ops.attention(q, k, v, heads, scale)
    d = C // heads
    qh, kh, vh = (t.float().reshape(n, -1, heads, d).transpose(1, 2) for t in (q, k, v))
    ref = (F.softmax(qh @ kh.transpose(-1, -2) * scale, -1) @ vh).transpose(1, 2).reshape(n, Lq, C)
    assert got.dtype == dt and got.shape == ref.shape
    assert rel_err(got, ref) < OP_RTOL[dname]
    assert rel_err(got, old.float()) < OP_RTOL[dname]


@pytest.mark.parametrize("dname", list(DTYPES))
@pytest.mark.parametrize("M,C", [(3024, 768), (1000, 384), (777, 192), (5000, 96)])
def test_layernorm_16_byte_form(ops, dname, M, C):
    """K8 (R/models/saliency_decoder/transformer.py:110,121): three octets per lane."""
    dt = DTYPES[dname]
    x = (rnd("ln16x%d" % C, M, C) * 2 + 0.3).to(DEV).to(dt)
    g, b = (1 + rnd("ln16g", C, scale=0.1)).to(DEV), rnd("ln16b", C, scale=0.1).to(DEV)
    got = ops.layernorm(x, g, b, 1e-6)
    with old_forms():
        old = ops.layernorm(x, g, b, 1e-6)
    ref = torch.nn.functional.layer_norm(x.float(), (C,), g, b, 1e-6)
    assert rel_err(got, ref) < OP_RTOL[dname] and rel_err(got, old.float()) < OP_RTOL[dname]
