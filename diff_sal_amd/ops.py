"""Tensor-level wrappers over the C ABI of libdiffsal_hip.so.

PyTorch is used for device memory and streams only: each function hands raw device pointers and
the current HIP stream to a hand-written gfx950 kernel.  Activations are channels-last (images
[N,H,W,C], tokens [M,C]); fp32 by default, bf16 / fp16 storage for the forward operators when the
tensors passed in are 16-bit (BASELINE configs[1] / configs[4]; arithmetic and statistics stay fp32).
There is no CPU / PyTorch fallback: a missing library or a non-GPU tensor raises.
"""
from __future__ import annotations

import contextlib
import ctypes as C
import threading
from typing import List, Optional, Sequence, Tuple

import torch

from . import _lib
from ._lib import ACT_GELU, ACT_GELU_GRAD, ACT_NONE, ACT_RELU, ACT_SIGMOID, ConvDesc  # noqa: F401

Tensor = torch.Tensor

# When set to a list (bench.py, one step inside its timed region), every operator launch is bracketed by HIP events on
# the launching stream and (start, end, algorithmic_flops, class, algorithmic_bytes) is appended.  ``class`` is the
# SURVEY 8(a) row the launch belongs to ("K12", ...); bytes = once-through traffic (inputs + outputs + weights in
# their storage type), the denominators of SURVEY 8(d).
PROFILE = None


_PROF_DEPTH = [0]      # brackets nest (an operator's finishing reduction inside its own bracket): only the outermost one records


class _prof:
    __slots__ = ("cls", "flops", "nbytes", "e0", "note", "counted", "kernel")

    def __init__(self, cls, flops=0.0, nbytes=0.0, note="", kernel=""):
        self.cls, self.flops, self.nbytes, self.e0, self.note, self.counted = cls, float(flops), float(nbytes), None, note, False
        self.kernel = kernel      # which kernel ran (GEMM entry points: diffsal_last_gemm_kernel(), set by the call site)

    def __enter__(self):
        if PROFILE is not None:
            self.counted = True
            if _PROF_DEPTH[0] == 0:
                self.e0 = torch.cuda.Event(enable_timing=True)
                self.e0.record()
            _PROF_DEPTH[0] += 1
        return self

    def __exit__(self, et, ev, tb):
        if self.counted:
            _PROF_DEPTH[0] -= 1
        if self.e0 is not None and et is None and PROFILE is not None:
            e1 = torch.cuda.Event(enable_timing=True)
            e1.record()
            PROFILE.append((self.e0, e1, self.flops, self.cls, self.nbytes, self.note, self.kernel))
        return False


def _classed(cls: str):
    """Bracket an operator without FLOP / byte accounting under a class name of bench.py's table (training-side kernels: the
    class table then covers the whole step instead of its forward operators and GEMMs only)."""
    def deco(fn):
        import functools

        @functools.wraps(fn)
        def wrapper(*a, **k):
            if PROFILE is None:
                return fn(*a, **k)
            with _prof(cls):
                return fn(*a, **k)
        return wrapper
    return deco


def _nb(*ts) -> float:
    """bytes of the given tensors (None skipped)"""
    return float(sum(t.numel() * t.element_size() for t in ts if t is not None))


GEMM_PRECISIONS = {"fp32": _lib.PREC_FP32, "bf16x3": _lib.PREC_BF16X3}

# The arithmetic of the implicit-GEMM kernel is a PER-CALL field of diffsal_conv_desc (the library holds no mode).  On
# the Python side the value put into each descriptor comes from a thread-local context: a process default
# (set_gemm_precision, used by the tools and the bench) that `gemm_precision(...)` overrides for a block -- SalUNet
# wraps its forward in it when its own ``gemm_precision`` attribute is set.
_DEFAULT_PRECISION = ["fp32"]
_tls = threading.local()


def set_gemm_precision(mode: str) -> None:
    """Default arithmetic of conv_igemm on fp32 tensors: "fp32" (exact) or "bf16x3" (split-precision bf16 MFMA with fp32
    accumulation, ~4e-6 relative; opt-in, see include/diffsal.h)."""
    if mode not in GEMM_PRECISIONS:
        raise ValueError(f"unknown GEMM precision {mode!r}; choose from {sorted(GEMM_PRECISIONS)}")
    _DEFAULT_PRECISION[0] = mode


def get_gemm_precision() -> str:
    return getattr(_tls, "precision", None) or _DEFAULT_PRECISION[0]


@contextlib.contextmanager
def gemm_precision(mode: Optional[str]):
    """``with ops.gemm_precision("bf16x3"):`` -- descriptors built inside carry that precision (None = no change)."""
    if mode is not None and mode not in GEMM_PRECISIONS:
        raise ValueError(f"unknown GEMM precision {mode!r}; choose from {sorted(GEMM_PRECISIONS)}")
    prev = getattr(_tls, "precision", None)
    if mode is not None:
        _tls.precision = mode
    try:
        yield
    finally:
        _tls.precision = prev


DTYPE_CODES = {torch.float32: _lib.F32, torch.bfloat16: _lib.BF16, torch.float16: _lib.F16}


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def _stream() -> int:
    """hipStream_t of torch's current stream on the current device (the raw getter: ~10x cheaper than building a
    torch.cuda.Stream object per launch, and a training step makes ~1 000 launches from Python)."""
    if _raw_stream is not None:
        return _raw_stream(torch.cuda.current_device())
    return torch.cuda.current_stream().cuda_stream


def _p(t: Optional[Tensor]):
    """Device pointer of a contiguous fp32 GPU tensor (parameters, statistics, every training-side tensor)."""
    if t is None:
        return None
    if not t.is_cuda or t.dtype != torch.float32 or not t.is_contiguous():
        raise ValueError(f"diff_sal_amd ops need contiguous fp32 GPU tensors, got {t.dtype} {t.device} "
                         f"contiguous={t.is_contiguous()}")
    return t.data_ptr()


def _dt(t: Tensor) -> int:
    """DIFFSAL_F32 / BF16 / F16 code of an activation tensor."""
    try:
        return DTYPE_CODES[t.dtype]
    except KeyError:
        raise ValueError(f"diff_sal_amd ops: unsupported activation dtype {t.dtype}") from None


def _pa(t: Optional[Tensor], code: int):
    """Device pointer of an activation tensor that must be contiguous, on the GPU and of the call's storage type."""
    if t is None:
        return None
    if not t.is_cuda or not t.is_contiguous() or DTYPE_CODES.get(t.dtype) != code:
        raise ValueError(f"diff_sal_amd ops: expected a contiguous GPU tensor of storage code {code}, got {t.dtype} "
                         f"{t.device} contiguous={t.is_contiguous()}")
    return t.data_ptr()


@_classed("pack")
def cast(x: Tensor, dtype: torch.dtype) -> Tensor:
    """Storage-type conversion on the device (round to nearest even)."""
    if x.dtype == dtype:
        return x
    out = torch.empty(x.shape, device=x.device, dtype=dtype)
    xc = x.contiguous()
    _lib.check(_lib.load().diffsal_cast(_pa(xc, _dt(xc)), _dt(xc), out.data_ptr(), _dt(out), xc.numel(), _stream()), "cast")
    return out


def temb_mlp(t: Tensor, freq: Tensor, w0: Tensor, b0: Tensor, w1: Tensor, b1: Tensor) -> Tensor:
    """K1. t: [B] int64 or float32 -> temb [B, 4*ch]."""
    lib = _lib.load()
    B, ch = t.shape[0], w0.shape[1]
    if t.dtype == torch.int64:
        is_f32 = 0
    elif t.dtype == torch.float32:
        is_f32 = 1
    else:
        t, is_f32 = t.to(torch.float32), 1
    if not t.is_cuda or not t.is_contiguous():
        raise ValueError("t must be a contiguous GPU tensor")
    out = torch.empty((2, B, 4 * ch), device=t.device, dtype=torch.float32)  # [0] = hidden scratch, [1] = temb
    # the kernels keep a [rows, 4 ch] operand in 64 KiB of LDS: larger batches in slices (17 clips and more: a lane per clip, 64 a call)
    rows = 64 if B > 16 else max(1, (64 * 1024) // (16 * ch))
    with _prof("K1", 2.0 * B * (ch * 4 * ch + 16 * ch * ch), _nb(w0, w1)):
        for s in range(0, B, rows):
            n = min(rows, B - s)
            _lib.check(lib.diffsal_temb_mlp(t[s:s + n].data_ptr(), is_f32, n, ch, _p(freq), _p(w0), _p(b0), _p(w1), _p(b1),
                                            _p(out[0, s:s + n]), _p(out[1, s:s + n]), _stream()), "temb_mlp")
    out = out[1]
    return out


def dense_small(x: Tensor, w: Tensor, bias: Optional[Tensor], swish_in: bool) -> Tensor:
    lib = _lib.load()
    B, K = x.shape
    N = w.shape[0]
    out = torch.empty((B, N), device=x.device, dtype=torch.float32)
    rows = 64 if B > 16 else max(1, (64 * 1024) // (4 * K))         # [rows, K] operand in 64 KiB of LDS: larger batches in slices
    with _prof("K1", 2.0 * B * K * N, _nb(w, x, out)):
        for s in range(0, B, rows):
            n = min(rows, B - s)
            _lib.check(lib.diffsal_dense_small(_p(x[s:s + n]), n, K, int(swish_in), _p(w), _p(bias), N, _p(out[s:s + n]), _stream()),
                       "dense_small")
    return out


def conv_in(x: Tensor, w9: Tensor, bias: Tensor, skip_mod: int = 0, out_dtype: torch.dtype = torch.float32,
            act: int = ACT_NONE) -> Tensor:
    """K2. x [B,1,H,W] (fp32) -> NHWC [B,H,W,C] of ``out_dtype``; with skip_mod=4 the pixels a stride-4 3x3 consumer
    never reads are left unwritten."""
    lib = _lib.load()
    B, _, H, W = x.shape
    Cc = w9.shape[0]
    out = torch.empty((B, H, W, Cc), device=x.device, dtype=out_dtype)
    frac = ((skip_mod - 1) / skip_mod) ** 2 if skip_mod > 0 else 1.0     # share of the pixels actually produced
    with _prof("K2-conv_in", 18.0 * B * H * W * Cc * frac, _nb(x) + _nb(out) * frac):
        _lib.check(lib.diffsal_conv_in(_p(x), _p(w9), _p(bias), out.data_ptr(), B, H, W, Cc, skip_mod, act, _dt(out), _stream()),
                   "conv_in")
    return out


def conv_in_s4(x: Tensor, w25: Tensor, bias: Tensor, out_dtype: torch.dtype = torch.float32) -> Tensor:
    """K2 fused: conv_in + the stride-4 Downsample as one 5x5 stride-4 convolution.  x [B,1,H,W] fp32 (H, W % 4 == 0),
    w25 [25, C] / bias [C] from ``compose_conv_in_s4`` -> NHWC [B,H/4,W/4,C] of ``out_dtype``."""
    lib = _lib.load()
    B, _, H, W = x.shape
    Cc = w25.shape[1]
    out = torch.empty((B, H // 4, W // 4, Cc), device=x.device, dtype=out_dtype)
    with _prof("K2", 50.0 * B * (H // 4) * (W // 4) * Cc, _nb(x, out)):
        _lib.check(lib.diffsal_conv_in_s4(_p(x), _p(w25), _p(bias), out.data_ptr(), B, H, W, Cc, _dt(out), _stream()), "conv_in_s4")
    return out


def compose_conv_in_s4(w1: Tensor, b1: Tensor, w2: Tensor, b2: Tensor):
    """Host-side composition (fp64) of conv_in [C1,1,3,3] + bias and the following 3x3 stride-4 conv [C,C1,3,3] + bias into the
    5x5 weights [25, C] (tap-major) and bias [C] of diffsal_conv_in_s4.  A parameter-layout transform, like weight packing."""
    w1d, w2d = w1.detach().double().cpu()[:, 0], w2.detach().double().cpu()
    C = w2d.shape[0]
    weff = torch.zeros((C, 5, 5), dtype=torch.float64)
    for ky2 in range(3):
        for kx2 in range(3):
            # contribution of tap (ky2, kx2) of the second conv: [C, C1] x [C1, 3, 3] placed at offset (ky2, kx2)
            weff[:, ky2:ky2 + 3, kx2:kx2 + 3] += torch.einsum("oc,cyx->oyx", w2d[:, :, ky2, kx2], w1d)
    beff = b2.detach().double().cpu() + torch.einsum("ocyx,c->o", w2d, b1.detach().double().cpu())
    dev = w1.device
    return weff.reshape(C, 25).t().contiguous().float().to(dev), beff.float().to(dev)


def groupnorm_swish(x: Tensor, gamma: Tensor, beta: Tensor, groups: int = 32, eps: float = 1e-6) -> Tensor:
    """K3 on NHWC [B,H,W,C]."""
    lib = _lib.load()
    B, H, W, Cc = x.shape
    out = torch.empty_like(x)
    nbytes = lib.diffsal_groupnorm_ws_bytes(B, groups)
    ws = torch.empty((nbytes // 8,), device=x.device, dtype=torch.float64)
    dt = _dt(x)
    with _prof("K3", 0.0, _nb(x, out)):
        _lib.check(lib.diffsal_groupnorm_swish(_pa(x, dt), _p(gamma), _p(beta), _pa(out, dt), B, H * W, Cc, groups, eps,
                                               ws.data_ptr(), nbytes, dt, _stream()), "groupnorm_swish")
    return out


def gn_affine(x: Tensor, gamma: Tensor, beta: Tensor, groups: int = 32, eps: float = 1e-6) -> Tensor:
    """GroupNorm(groups, eps) of NHWC x [B,H,W,C] in AFFINE form: ab [B,2,C] fp32 with norm(x)[b,..,c] = ab[b,0,c] x + ab[b,1,c]
    (statistics only: one read of x, nothing written back; the consumer applies it as it loads x -- ``conv3x3_wino4_ex(gn_ab=...)``,
    R/models/saliency_decoder/sal_unet.py:41-44)."""
    lib = _lib.load()
    B, H, W, Cc = x.shape
    ab = torch.empty((B, 2, Cc), device=x.device, dtype=torch.float32)
    nbytes = lib.diffsal_groupnorm_ws_bytes(B, groups)
    ws = torch.empty((nbytes // 8,), device=x.device, dtype=torch.float64)
    dt = _dt(x)
    with _prof("K3", 0.0, _nb(x)):
        _lib.check(lib.diffsal_gn_affine(_pa(x, dt), _p(gamma), _p(beta), _p(ab), B, H * W, Cc, groups, eps, ws.data_ptr(), nbytes, dt,
                                         _stream()), "gn_affine")
    return ab


def _wino4_desc(x: Tensor, Cout: int, act: int, rowvec: Optional[Tensor], dil: int = 1):
    N, H, W, Cin = x.shape
    d = ConvDesc(N, H, W, Cin, H, W, Cout, 3, 3, 1, 1, dil, dil, dil, dil, act, rowvec.shape[-1] if rowvec is not None else 0, 0,
                 _lib.PREC_FP32, _lib.F32)
    if rowvec is not None:
        assert rowvec.stride(-1) == 1 and rowvec.dtype == torch.float32 and rowvec.is_cuda
        d.rowvec_ld = rowvec.stride(0)
    return d


def resblock_wino4_plan(x: Tensor, Cout: int, groups: int = 32) -> Optional[dict]:
    """None unless BOTH 3x3 convolutions of a ResnetBlock on NHWC x [N,H,W,Cin] (Cin -> Cout, Cout -> Cout, padding 1) take the
    F(4x4, 3x3) path (``diffsal_conv_wino4_supported``: fp32, Cin % 96 == 0, enough tiles ...); else what the extended entry point
    offers for this shape: ``side`` -- the 1x1 shortcut can share the launch of conv1's position products --, ``stats`` -- conv1's
    output transform can emit norm2's statistics."""
    if x.dtype != torch.float32 or get_gemm_precision() != "fp32":
        return None
    key = ("res", tuple(x.shape), Cout, groups, _tuning_epoch())
    if key in _PLAN_CACHE:
        return _PLAN_CACHE[key]
    plan = _resblock_wino4_plan(x, Cout, groups)
    _PLAN_CACHE[key] = plan
    return plan


_PLAN_CACHE: dict = {}      # shape -> planner answers (they depend on the shape and the library's tuning switches only)


def _tuning_epoch() -> int:
    return _lib.TUNING_EPOCH[0]


def _resblock_wino4_plan(x: Tensor, Cout: int, groups: int):
    lib = _lib.load()
    N, H, W, Cin = x.shape
    d1 = _wino4_desc(x, Cout, ACT_NONE, None)
    d2 = ConvDesc(N, H, W, Cout, H, W, Cout, 3, 3, 1, 1, 1, 1, 1, 1, ACT_NONE, 0, 0, _lib.PREC_FP32, _lib.F32)
    if Cout % 96 != 0 or not lib.diffsal_conv_wino4_supported(C.byref(d1)) or not lib.diffsal_conv_wino4_supported(C.byref(d2)):
        return None
    return dict(side=bool(lib.diffsal_conv_wino4_side_supported(C.byref(d1), N * H * W)),
                stats=lib.diffsal_conv_wino4_stats_bytes(C.byref(d1), groups) > 0)


def wino4_supported(x: Tensor, Cout: int, dil: int = 1) -> bool:
    """The planner takes the F(4x4, 3x3) path for a 3x3 / padding = dilation convolution of NHWC fp32 x to Cout channels."""
    if x.dtype != torch.float32 or get_gemm_precision() != "fp32":
        return False
    key = ("w4", tuple(x.shape), Cout, dil, _tuning_epoch())
    if key not in _PLAN_CACHE:
        d = _wino4_desc(x, Cout, ACT_NONE, None, dil)
        _PLAN_CACHE[key] = bool(_lib.load().diffsal_conv_wino4_supported(C.byref(d)))
    return _PLAN_CACHE[key]


def conv3x3_wino4_ex(x: Tensor, wino4: Tensor, *, bias: Optional[Tensor] = None, scale: Optional[Tensor] = None,
                     shift: Optional[Tensor] = None, rowvec: Optional[Tensor] = None,
                     residual: Optional[Tensor] = None, act: int = ACT_NONE, gn_ab: Optional[Tensor] = None, gn_swish: bool = True,
                     side=None, stats_groups: int = 0, dil: int = 1, up2=None, tag: str = "K4"):
    """3x3 / padding 1 convolution of NHWC fp32 x on the F(4x4, 3x3) path with the ResnetBlock extras of ``diffsal_conv_wino4_ex``:
    ``gn_ab`` [N,2,Cin] (``gn_affine``): the input is normalised (+ swish) as the input transform loads it; ``side = (a, w)``: the
    plain product a [N,H,W,K] x w [Cout, K]^T (the 1x1 shortcut, no bias) computed by the launch of the position products;
    ``stats_groups`` > 0: per-(image, group) sums of the result for the next GroupNorm.  Returns (out, side_out or None, stats or
    None) -- stats is the opaque buffer ``gn_affine_from_stats`` takes.

    ``dil`` 1 or 2 (padding = dilation); ``scale`` / ``shift``: BatchNorm affine of the epilogue.  ``up2 = (c_ext, scale1, shift1,
    act1)``: x is the [N,H,W,Cin] buffer of which ``up2_conv3x3_d2(..., ring_only=True)`` wrote only the border ring; the input
    transform forms the interior act1(BN1(interpolation of c_ext)) itself (dilation 2 only)."""
    lib = _lib.load()
    N, H, W, Cin = x.shape
    Cout = wino4.shape[1]
    d = _wino4_desc(x, Cout, act, rowvec, dil)
    out = torch.empty((N, H, W, Cout), device=x.device, dtype=torch.float32)
    ws_bytes = lib.diffsal_conv_wino4_ws_bytes(C.byref(d))
    ws = torch.empty((ws_bytes // 4,), device=x.device, dtype=torch.float32)
    ext = _lib.Wino4Ext()
    side_out = stats = None
    if gn_ab is not None:
        assert gn_ab.shape == (N, 2, Cin) and gn_ab.dtype == torch.float32 and gn_ab.is_contiguous()
        ext.in_ab, ext.in_swish = gn_ab.data_ptr(), 1 if gn_swish else 0
    if side is not None:
        sa, sw = side
        assert sa.dtype == torch.float32 and sa.is_contiguous() and sa.shape[-1] == Cin and sw.shape == (Cout, Cin)
        rows = sa.numel() // Cin
        side_out = torch.empty(tuple(sa.shape[:-1]) + (Cout,), device=x.device, dtype=torch.float32)
        ext.side_a, ext.side_w, ext.side_out, ext.side_rows = sa.data_ptr(), sw.data_ptr(), side_out.data_ptr(), rows
    if up2 is not None:
        c_ext, s1, h1, a1 = up2
        assert dil == 2 and gn_ab is None and c_ext.dtype == torch.float32 and c_ext.is_contiguous()
        assert tuple(c_ext.shape) == (N, H // 2 + 2, W // 2 + 2, Cin), (c_ext.shape, x.shape)
        ext.up2_c, ext.up2_scale, ext.up2_shift, ext.up2_act = c_ext.data_ptr(), _p(s1), _p(h1), a1
    if stats_groups > 0:
        sb = lib.diffsal_conv_wino4_stats_bytes(C.byref(d), stats_groups)
        if sb == 0:
            raise RuntimeError("conv3x3_wino4_ex: output statistics are not available for this shape (resblock_wino4_plan)")
        stats = torch.empty((sb // 8,), device=x.device, dtype=torch.float64)
        ext.out_stats, ext.out_groups = stats.data_ptr(), stats_groups
    rv = rowvec.data_ptr() if rowvec is not None else None
    args = (C.byref(d), _p(x), _p(wino4), _p(bias), _p(scale), _p(shift), rv, _p(residual), _p(out), _p(ws), ws_bytes, C.byref(ext))
    n_tiles = N * dil * dil * (((H + dil - 1) // dil + 3) // 4) * (((W + dil - 1) // dil + 3) // 4)
    if PROFILE is None:
        _lib.check(lib.diffsal_conv_wino4_ex(*args, 7, _stream()), "conv_wino4_ex")
    else:
        note = f"M={N * H * W} K={9 * Cin} N={Cout} 3x3 winograd F(4x4,3x3)"
        vb, mb = 36 * n_tiles * Cin * 4, 36 * n_tiles * Cout * 4
        nside = 0 if side is None else side[0].numel() // Cin
        xin = _nb(x) if up2 is None else _nb(up2[0])
        with _prof(tag + "-xf", 0.0, xin + vb, note + (": input transform (GroupNorm + swish on load)" if gn_ab is not None else
                                                        ": input transform (interpolation of the source-resolution convolution on load)" if up2 is not None
                                                        else ": input transform")) as pr:
            _lib.check(lib.diffsal_conv_wino4_ex(*args, 1, _stream()), "conv_wino4_ex")
            pr.kernel = "wino4_input_kernel"
        with _prof(tag, 2.0 * (n_tiles * 36 + nside) * Cin * Cout, vb + mb + _nb(wino4) + (0 if side is None else _nb(side[0], side_out)),
                   note + f": 36 x (M={n_tiles} K={Cin} N={Cout})" + (f" + 1x1 shortcut M={nside}" if nside else "")) as pr:
            _lib.check(lib.diffsal_conv_wino4_ex(*args, 2, _stream()), "conv_wino4_ex")
            pr.kernel = lib.diffsal_last_gemm_kernel().decode()
        with _prof(tag + "-xf", 0.0, mb + _nb(residual, out), note + ": output transform" + (" (+ GroupNorm sums)" if stats is not None else "")) as pr:
            _lib.check(lib.diffsal_conv_wino4_ex(*args, 4, _stream()), "conv_wino4_ex")
            pr.kernel = "wino4_output_kernel"
    if stats is not None:
        stats = (stats, d, stats_groups)
    return out, side_out, stats


def gn_affine_from_stats(stats, gamma: Tensor, beta: Tensor, eps: float = 1e-6) -> Tensor:
    """ab [N,2,C] of GroupNorm over the output of the convolution whose ``conv3x3_wino4_ex(stats_groups=...)`` call returned ``stats``."""
    lib = _lib.load()
    buf, d, groups = stats
    ab = torch.empty((d.N, 2, d.Cout), device=buf.device, dtype=torch.float32)
    with _prof("K3", 0.0, _nb(buf)):
        _lib.check(lib.diffsal_gn_affine_wino4(C.byref(d), buf.data_ptr(), _p(gamma), _p(beta), groups, eps, _p(ab), _stream()),
                   "gn_affine_wino4")
    return ab


def _as_conv4(w: Tensor) -> Tensor:
    """Conv3d weights [Cout, Cin, KT, 1, 1] and Linear weights [N, K] as [Cout, Cin, taps, 1] views."""
    if w.dim() == 5:
        return w[:, :, :, 0, 0].unsqueeze(-1)
    if w.dim() == 2:
        return w[:, :, None, None]
    return w


@_classed("pack")
def _pack(w4: Tensor, mode: int) -> Tensor:
    """csrc/pack.hip on a contiguous [Cout, Cin, kh, kw] (modes 0, 1) or packed [Cout, K] (mode 2, shape from w4)."""
    lib = _lib.load()
    co, ci, kh, kw = w4.shape
    taps = kh * kw
    src = w4.contiguous()
    if mode == 1:
        out = torch.empty((ci, taps * co), device=w4.device, dtype=torch.float32)
    elif mode == 3:
        out = torch.empty((taps * ci, co), device=w4.device, dtype=torch.float32)
    else:
        out = torch.empty((co, taps * ci), device=w4.device, dtype=torch.float32)
    _lib.check(lib.diffsal_pack_weight(_p(src), _p(out), co, ci, taps, mode, _stream()), "pack_weight")
    return out


class _PackBatch:
    """Weight repacks of a training step in one launch.  Inside ``with batched_packs():`` (DiffusionTrainStep wraps forward +
    backward in it) a repack of an ``nn.Parameter`` is looked up by (storage address, shape, layout mode): a known entry returns
    its persistent output buffer, which the ONE ``diffsal_pack_weight_many`` launch at the top of the context has just rebuilt
    from the parameter's current values; an unknown one is packed directly and joins the table for the next step.  Only
    Parameters are cached (their storage lives as long as the module; temporaries could reuse an address), entries whose
    Parameter is gone or moved are dropped, and outside the context every call packs directly -- nothing stale can be read."""

    def __init__(self):
        self.entries = {}       # key -> [weakref(param), out, co, ci, taps, mode]
        self.table = None       # device job table (uint8) of the current entry set
        self.dirty = True
        self.active = False
        self.fresh = set()
        self.tiles = 0
        self.max_taps = 1

    def refresh(self):
        import weakref  # noqa: F401

        dead = [k for k, e in self.entries.items() if e[0]() is None or e[0]().data_ptr() != k[0]]
        for k in dead:
            del self.entries[k]
            self.dirty = True
        self.fresh = set()
        if not self.entries:
            return
        if self.dirty or self.table is None:
            import numpy as np

            dt = np.dtype([("src", "u8"), ("dst", "u8"), ("Cout", "i4"), ("Cin", "i4"), ("taps", "i4"), ("mode", "i4"),
                           ("tile0", "i4"), ("reserved", "i4")])
            jobs = np.zeros(len(self.entries), dtype=dt)
            t0, mt, dev = 0, 1, None
            for i, (k, e) in enumerate(self.entries.items()):
                _, out, co, ci, taps, mode = e
                jobs[i] = (k[0], out.data_ptr(), co, ci, taps, mode, t0, 0)
                t0 += (ci // 32) * ((co + 31) // 32)
                mt = max(mt, taps)
                dev = out.device
            self.table = torch.from_numpy(jobs.view(np.uint8).copy()).to(dev)
            self.tiles, self.max_taps, self.dirty = t0, mt, False
        with _prof("pack"):
            _lib.check(_lib.load().diffsal_pack_weight_many(self.table.data_ptr(), len(self.entries), self.tiles, self.max_taps,
                                                            _stream()), "pack_weight_many")
        self.fresh = set(self.entries)

    def get(self, w, w4: Tensor, mode: int):
        """Packed form of Parameter ``w`` (``w4`` = its contiguous [Cout, Cin, kh, kw] view) in layout ``mode``, or None when the
        cache does not apply (the caller packs directly)."""
        import weakref

        if not self.active or not isinstance(w, torch.nn.Parameter) or not w4.is_contiguous() or w4.data_ptr() != w.data_ptr():
            return None
        co, ci, kh, kw = w4.shape
        if ci % 32 or ((mode == 1 or mode == 3) and co % 32) or kh * kw > 25:
            return None
        key = (w.data_ptr(), (co, ci, kh, kw), mode)
        e = self.entries.get(key)
        if e is not None and key in self.fresh:
            return e[1]
        out = _pack(w4, mode)                              # first sight (or registered after this step's refresh): direct
        if e is None:
            self.entries[key] = [weakref.ref(w), out, co, ci, kh * kw, mode]
            self.dirty = True
        else:
            e[1] = out
            self.dirty = True
        return out


_PACKS = _PackBatch()


@contextlib.contextmanager
def batched_packs():
    """One launch for the weight repacks of the parameters seen in earlier uses of this context (see ``_PackBatch``)."""
    pb = _PACKS
    if pb.active:
        yield
        return
    pb.active = True
    try:
        pb.refresh()
        yield
    finally:
        pb.active = False
        pb.fresh = set()


def pack_conv_weight(w: Tensor) -> Tensor:
    """[Cout, Cin, KH, KW] (or Conv3d [Cout, Cin, KT, 1, 1]) -> [Cout, K] in the kernel's k order
    (ci // 32, tap, ci % 32); see include/diffsal.h."""
    return _pack(_as_conv4(w.detach()), 0)


def pack_wino_weight(w: Tensor) -> Tensor:
    """[Cout, Cin, 3, 3] -> the Winograd F(2x2, 3x3) weight U = G w G^T in the blocked layout diffsal_conv_wino reads:
    [Cin/8][ceil(Cout/64)][16][64][8] fp32 (output channels past Cout zero); include/diffsal.h."""
    w = w.detach().float()
    Cout, Cin = w.shape[:2]
    assert tuple(w.shape[2:]) == (3, 3) and Cin % 8 == 0
    G = torch.tensor([[1.0, 0.0, 0.0], [0.5, 0.5, 0.5], [0.5, -0.5, 0.5], [0.0, 0.0, 1.0]], device=w.device)
    U = torch.einsum("ia,ocab,jb->ijoc", G, w, G).reshape(16, Cout, Cin)
    cb = (Cout + 63) // 64
    if cb * 64 != Cout:
        U = torch.cat([U, U.new_zeros(16, cb * 64 - Cout, Cin)], 1)
    return U.view(16, cb, 64, Cin // 8, 8).permute(3, 1, 0, 2, 4).contiguous()


def pack_wino4_weight(w: Tensor) -> Tensor:
    """[Cout, Cin, 3, 3] -> the Winograd F(4x4, 3x3) weight U = G w G^T as [36][Cout][Cin] fp32 (position-major: the B operand of
    the 36 position products of diffsal_conv_wino4); include/diffsal.h.  Computed in fp64, rounded once."""
    w = w.detach().double()
    Cout, Cin = w.shape[:2]
    assert tuple(w.shape[2:]) == (3, 3)
    G = torch.tensor([[1 / 4, 0, 0], [-1 / 6, -1 / 6, -1 / 6], [-1 / 6, 1 / 6, -1 / 6], [1 / 24, 1 / 12, 1 / 6],
                      [1 / 24, -1 / 12, 1 / 6], [0, 0, 1]], device=w.device, dtype=torch.float64)
    return torch.einsum("ia,ocab,jb->ijoc", G, w, G).reshape(36, Cout, Cin).float().contiguous()


@_classed("pack")
def wino4_weight(w: Tensor, dgrad: bool = False) -> Tensor:
    """The F(4x4, 3x3) weight U = G w G^T computed on the device in fp32 (training; ``pack_wino4_weight`` is the fp64 host form of
    inference): [36, Cout, Cin], or with ``dgrad`` [36, Cin, Cout] of the flipped kernel (the data-gradient convolution)."""
    lib = _lib.load()
    w = w.detach()
    Cout, Cin = w.shape[:2]
    assert tuple(w.shape[2:]) == (3, 3) and w.dtype == torch.float32
    U = torch.empty((36, Cin, Cout) if dgrad else (36, Cout, Cin), device=w.device, dtype=torch.float32)
    _lib.check(lib.diffsal_wino4_weight(_p(w.contiguous()), _p(U), Cout, Cin, int(dgrad), _stream()), "wino4_weight")
    return U


class WinoWeights:
    """The Winograd forms of one 3x3 weight: ``f2`` (pack_wino_weight) and ``f4`` (pack_wino4_weight); conv_igemm(wino=...) asks
    the library per shape which of them -- if any -- to use."""

    def __init__(self, w: Tensor, f2: bool = True, f4: bool = True):
        self.f2 = pack_wino_weight(w) if f2 and w.shape[1] % 8 == 0 else None
        self.f4 = pack_wino4_weight(w) if f4 and w.shape[1] % 96 == 0 else None


@_classed("pack")
def split_weight(w_packed: Tensor) -> Tensor:
    """bf16x3 mode only: pre-split a packed fp32 weight [Cout, K] into bf16 hi/lo halves per 32-k slice (same shape and
    size; include/diffsal.h, w_format = 1).  The result is tagged: conv_igemm then announces w_format = 1 together with
    precision = bf16x3 in that call's descriptor."""
    lib = _lib.load()
    out = torch.empty_like(w_packed)
    _lib.check(lib.diffsal_split_weight(_p(w_packed.contiguous()), _p(out), w_packed.numel(), _stream()), "split_weight")
    out._diffsal_split = True
    return out


def pack_dgrad_weight(w: Tensor) -> Tensor:
    """Packed weight of the data-gradient convolution: [Cin, KH*KW*Cout], taps flipped (include/diffsal.h, mode 1)."""
    w4 = _as_conv4(w.detach())
    hit = _PACKS.get(w, w4, 1)
    return hit if hit is not None else _pack(w4, 1)


def pack_cols_weight(w: Tensor) -> Tensor:
    """[KH*KW*Cin, Cout]: weight of the GEMM dXcols = dY W for non-overlapping convolutions (mode 3)."""
    w4 = _as_conv4(w.detach())
    hit = _PACKS.get(w, w4, 3)
    return hit if hit is not None else _pack(w4, 3)


@_classed("col2im")
def col2im_disjoint(cols: Tensor, in_shape, out_hw, kh: int, kw: int, stride, pad) -> Tensor:
    """cols [N*Ho*Wo, kh*kw*C] -> dx [N,H,W,C] for a convolution with stride >= kernel (each input pixel has one source)."""
    lib = _lib.load()
    N, H, W, Cc = in_shape
    dx = torch.empty((N, H, W, Cc), device=cols.device, dtype=torch.float32)
    _lib.check(lib.diffsal_col2im_disjoint(_p(cols), _p(dx), N, H, W, Cc, out_hw[0], out_hw[1], kh, kw, stride[0], stride[1],
                                           pad[0], pad[1], _stream()), "col2im_disjoint")
    return dx


@_classed("col2im")
def col2im_gather(cols: Tensor, in_shape, out_hw, kh: int, kw: int, stride, pad) -> Tensor:
    """cols [N*Ho*Wo, kh*kw*C] -> dx [N,H,W,C] for a strided convolution whose taps overlap (1 < stride < kernel): every
    input pixel sums the column entries that map onto it (fixed order, no atomics)."""
    lib = _lib.load()
    N, H, W, Cc = in_shape
    dx = torch.empty((N, H, W, Cc), device=cols.device, dtype=torch.float32)
    _lib.check(lib.diffsal_col2im_gather(_p(cols), _p(dx), N, H, W, Cc, out_hw[0], out_hw[1], kh, kw, stride[0], stride[1],
                                         pad[0], pad[1], _stream()), "col2im_gather")
    return dx


class _PackWeightFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, w, ready):
        ctx.shape = tuple(w.shape)
        return ready.detach() if ready is not None else _pack(_as_conv4(w), 0)     # (a fresh tensor object over the batch's buffer)

    @staticmethod
    def backward(ctx, dwp):
        lib = _lib.load()
        shape = ctx.shape
        co, ci = shape[0], shape[1]
        taps = 1
        for s_ in shape[2:]:
            taps *= s_
        dw = torch.empty(shape, device=dwp.device, dtype=torch.float32)
        _lib.check(lib.diffsal_pack_weight(_p(dwp.contiguous()), _p(dw), co, ci, taps, 2, _stream()), "pack_weight")
        return dw, None


def pack_conv_weight_diff(w: Tensor) -> Tensor:
    """pack_conv_weight that stays on the autograd tape: dW (packed, from conv_wgrad) flows back to ``w`` in the
    parameter layout."""
    return _PackWeightFn.apply(w, _PACKS.get(w, _as_conv4(w.detach()), 0))


def conv_igemm(x: Tensor, w_packed: Tensor, *, kh: int = 1, kw: int = 1, stride=(1, 1), pad=(0, 0), dil=(1, 1),
               out_hw: Optional[Sequence[int]] = None, bias: Optional[Tensor] = None, scale: Optional[Tensor] = None,
               shift: Optional[Tensor] = None, rowvec: Optional[Tensor] = None, residual: Optional[Tensor] = None,
               act: int = ACT_NONE, out: Optional[Tensor] = None, tag: str = "gemm", wino=None) -> Tensor:
    """Implicit-GEMM conv on NHWC x [N,H,W,Cin] with packed weight [Cout, kh*kw*Cin] -> [N,Ho,Wo,Cout].

    ``pad`` is (top, left); bottom/right padding is implied by ``out_hw`` (zero fill outside).  ``wino``: the same weight in
    Winograd form (a ``pack_wino_weight`` tensor, or ``WinoWeights`` = F(2x2) and F(4x4) forms: F(4x4) where
    ``diffsal_conv_wino4_supported`` says so, ~1e-5 relative rounding); used instead of the direct kernel when the library's planner expects a gain
    (``diffsal_conv_wino_supported``: fp32 3x3 stride-1, padding = dilation in {1, 2}, Cin % 32 == 0, Cout >= 128, enough
    workgroups and a transformed input of at most 160 MB -- both depend on the BATCH, so the same clip can take the Winograd
    kernel in a large pass and the direct kernel alone: results then differ by the transforms' rounding, ~1e-6 relative for
    F(2x2) and ~1e-5 for F(4x4); ``SalUNet.winograd = False`` / DIFFSAL_NO_WINOGRAD=1 pin the direct kernel for callers that need
    chunked and whole-batch evaluation to agree bit for bit)."""
    lib = _lib.load()
    N, H, W, Cin = x.shape
    Cout = w_packed.shape[0]
    assert w_packed.shape[1] == kh * kw * Cin, (w_packed.shape, kh, kw, Cin)
    if out_hw is None:
        Ho = (H + 2 * pad[0] - dil[0] * (kh - 1) - 1) // stride[0] + 1
        Wo = (W + 2 * pad[1] - dil[1] * (kw - 1) - 1) // stride[1] + 1
    else:
        Ho, Wo = out_hw
    dt = _dt(x)
    if out is None:
        out = torch.empty((N, Ho, Wo, Cout), device=x.device, dtype=x.dtype)
    split = bool(getattr(w_packed, "_diffsal_split", False))
    prec = GEMM_PRECISIONS["bf16x3" if split else get_gemm_precision()] if dt == _lib.F32 else _lib.PREC_FP32
    d = ConvDesc(N, H, W, Cin, Ho, Wo, Cout, kh, kw, stride[0], stride[1], pad[0], pad[1], dil[0], dil[1], act,
                 rowvec.shape[-1] if rowvec is not None else 0, 1 if split else 0, prec, dt)
    if rowvec is not None:
        assert rowvec.stride(-1) == 1 and rowvec.dtype == torch.float32 and rowvec.is_cuda
        d.rowvec_ld = rowvec.stride(0)
        rv = rowvec.data_ptr()
    else:
        rv = None
    wino4 = None
    if isinstance(wino, WinoWeights):
        wino, wino4 = wino.f2, wino.f4
    if wino4 is not None and lib.diffsal_conv_wino4_supported(C.byref(d)):
        ws_bytes = lib.diffsal_conv_wino4_ws_bytes(C.byref(d))
        ws = torch.empty((ws_bytes // 4,), device=x.device, dtype=torch.float32)
        dd = dil[0]
        n_tiles = N * dd * dd * (((Ho + dd - 1) // dd + 3) // 4) * (((Wo + dd - 1) // dd + 3) // 4)
        args = (C.byref(d), _p(x), _p(wino4), _p(bias), _p(scale), _p(shift), rv, _p(residual), _p(out), _p(ws), ws_bytes)
        if PROFILE is None:
            _lib.check(lib.diffsal_conv_wino4(*args, _stream()), "conv_wino4")
            return out
        # profiling: the three launches bracketed one by one -- the transforms are streaming kernels (class <tag>-xf, HBM-bound),
        # the products a launch of the GEMM kernel.  FLOPs actually issued: 36 products per 4x4 tile, input and output channel
        # (the direct form has 144)
        note = f"M={N * Ho * Wo} K={kh * kw * Cin} N={Cout} {kh}x{kw} winograd F(4x4,3x3)"
        vb, mb = 36 * n_tiles * Cin * 4, 36 * n_tiles * Cout * 4
        with _prof(tag + "-xf", 0.0, _nb(x) + vb, note + ": input transform") as pr:
            _lib.check(lib.diffsal_conv_wino4_stages(*args, 1, _stream()), "conv_wino4")
            pr.kernel = "wino4_input_kernel"
        with _prof(tag, 2.0 * n_tiles * 36 * Cin * Cout, vb + mb + _nb(wino4), note + f": 36 x (M={n_tiles} K={Cin} N={Cout})") as pr:
            _lib.check(lib.diffsal_conv_wino4_stages(*args, 2, _stream()), "conv_wino4")
            pr.kernel = lib.diffsal_last_gemm_kernel().decode()
        with _prof(tag + "-xf", 0.0, mb + _nb(residual, out), note + ": output transform") as pr:
            _lib.check(lib.diffsal_conv_wino4_stages(*args, 4, _stream()), "conv_wino4")
            pr.kernel = "wino4_output_kernel"
        return out
    if wino is not None and lib.diffsal_conv_wino_supported(C.byref(d)):
        ws_bytes = lib.diffsal_conv_wino_ws_bytes(C.byref(d))
        ws = torch.empty((ws_bytes // 4,), device=x.device, dtype=torch.float32)
        dd = dil[0]
        n_tiles = N * dd * dd * (((H + dd - 1) // dd + 1) // 2) * (((W + dd - 1) // dd + 1) // 2)
        # FLOPs actually issued: 16 products per 2x2 tile, input and output channel (the direct form has 36)
        with _prof(tag, 2.0 * n_tiles * 16 * Cin * Cout, _nb(x, wino, residual, out) + 2 * 16 * n_tiles * Cin * 4,
                   f"M={N * Ho * Wo} K={kh * kw * Cin} N={Cout} {kh}x{kw} winograd F(2x2,3x3)" if PROFILE is not None else "") as pr:
            _lib.check(lib.diffsal_conv_wino(C.byref(d), _p(x), _p(wino), _p(bias), _p(scale), _p(shift), rv, _p(residual), _p(out),
                                             _p(ws), ws_bytes, _stream()), "conv_wino")
            if PROFILE is not None:
                pr.kernel = lib.diffsal_last_gemm_kernel().decode()
        return out
    ws_bytes = lib.diffsal_conv_igemm_ws_bytes(C.byref(d))
    ws = torch.empty((ws_bytes // 4,), device=x.device, dtype=torch.float32) if ws_bytes else None
    with _prof(tag, 2.0 * N * Ho * Wo * Cout * kh * kw * Cin, _nb(x, w_packed, residual, out),
               f"M={N * Ho * Wo} K={kh * kw * Cin} N={Cout} {kh}x{kw}" if PROFILE is not None else "") as pr:
        _lib.check(lib.diffsal_conv_igemm(C.byref(d), _pa(x, dt), _pa(w_packed, dt), _p(bias), _p(scale), _p(shift), rv,
                                          _pa(residual, dt), _pa(out, dt), _p(ws), ws_bytes, _stream()), "conv_igemm")
        if PROFILE is not None:
            pr.kernel = lib.diffsal_last_gemm_kernel().decode()
    return out


def up2_conv3x3_d2(z: Tensor, w_packed: Tensor, wino, tapw: Tensor, *, scale: Optional[Tensor] = None, shift: Optional[Tensor] = None,
                   act: int = ACT_NONE, tag: str = "K12", ring_only: bool = False):
    """act(BN(conv3x3(dilation 2, padding 2)(bilinear_up2(z)))) for z [N,h,w,Cin] (fp32, or bf16 / fp16 storage) -> [N,2h,2w,Cout]:
    the convolution runs at the SOURCE resolution on the grid extended by one pixel (F(4x4) Winograd where ``wino`` qualifies -- fp32
    only --, else the direct kernel of the storage type), the
    nine tap products only on the border pixels, and one kernel interpolates and corrects the 3-pixel border ring
    (``diffsal_up2_conv_commute``, csrc/upconv.hip).  ``w_packed``: pack_conv_weight(w); ``tapw``: the [9*Cout, Cin] tap matrix (row =
    tap * Cout + co).  Exact up to summation order.

    ``ring_only``: returns ``(out, c_ext)`` with only the 3-pixel border ring of ``out`` written (``diffsal_up2_conv_commute_ring``): the
    consumer forms the interior from ``c_ext`` itself -- ``conv3x3_wino4_ex(up2=(c_ext, scale, shift, act))``, UpEmbed's second
    convolution reading the first one's source-resolution result."""
    lib = _lib.load()
    N, h, w, Cin = z.shape
    Cout = w_packed.shape[0]
    c_ext = conv_igemm(z, w_packed, kh=3, kw=3, pad=(2, 2), dil=(1, 1), out_hw=(h + 2, w + 2), tag=tag, wino=wino)
    z = z.contiguous()
    zb = torch.empty((N, 2 * w + 2 * h - 4, Cin), device=z.device, dtype=z.dtype)
    dt = _dt(z)
    _lib.check(lib.diffsal_border_gather(_pa(z, dt), _pa(zb, dt), N, h, w, Cin, dt, _stream()), "border_gather")
    tb = linear(zb, tapw, None, tag=tag)                                   # [N, 2w + 2h - 4, 9 * Cout]
    out = torch.empty((N, 2 * h, 2 * w, Cout), device=z.device, dtype=z.dtype)
    with _prof(tag + "-tap", 0.0, _nb(c_ext, tb) + (0 if ring_only else _nb(out)),
               f"up2 commute {h}x{w} C={Cout}" + (" (border ring only)" if ring_only else "") if PROFILE is not None else "") as pr:
        dt = _dt(z)
        fn = lib.diffsal_up2_conv_commute_ring if ring_only else lib.diffsal_up2_conv_commute
        _lib.check(fn(_pa(c_ext, dt), _pa(tb, dt), _p(scale), _p(shift), _pa(out, dt), N, h, w, Cout, act, dt, _stream()), "up2_conv_commute")
        pr.kernel = "up2_conv_commute_ring_kernel" if ring_only else "up2_conv_commute_kernel"
    return (out, c_ext) if ring_only else out


def conv_igemm_group(problems: Sequence[dict], tag: str = "gemm") -> List[Tensor]:
    """Up to four independent convolutions / plain products in ONE launch (``diffsal_conv_igemm_group``).  Each problem is a dict
    of ``conv_igemm`` keywords: ``x`` (NHWC), ``w`` (packed [Cout, kh*kw*Cin]) and optionally ``kh, kw, stride, pad, dil, out_hw,
    bias, act, out``; epilogue = bias + activation only.  Returns the outputs in order."""
    lib = _lib.load()
    n = len(problems)
    if not 1 <= n <= 4:
        raise RuntimeError("conv_igemm_group: 1..4 problems")
    if get_gemm_precision() != "fp32" or any(getattr(pr["w"], "_diffsal_split", False) for pr in problems):
        # split-precision mode (pre-split weights, per-call precision): the grouped entry point is exact-arithmetic only
        return [conv_igemm(pr["x"], pr["w"], kh=pr.get("kh", 1), kw=pr.get("kw", 1), stride=pr.get("stride", (1, 1)),
                           pad=pr.get("pad", (0, 0)), dil=pr.get("dil", (1, 1)), out_hw=pr.get("out_hw"), bias=pr.get("bias"),
                           act=pr.get("act", ACT_NONE), out=pr.get("out"), tag=tag) for pr in problems]
    descs, outs, keep = [], [], []
    flops = nbytes = 0.0
    ws_bytes = 0
    notes = []
    for pr in problems:
        x, w = pr["x"], pr["w"]
        kh, kw = pr.get("kh", 1), pr.get("kw", 1)
        stride, pad, dil = pr.get("stride", (1, 1)), pr.get("pad", (0, 0)), pr.get("dil", (1, 1))
        N, H, W, Cin = x.shape
        Cout = w.shape[0]
        assert w.shape[1] == kh * kw * Cin, (w.shape, kh, kw, Cin)
        if pr.get("out_hw") is None:
            Ho = (H + 2 * pad[0] - dil[0] * (kh - 1) - 1) // stride[0] + 1
            Wo = (W + 2 * pad[1] - dil[1] * (kw - 1) - 1) // stride[1] + 1
        else:
            Ho, Wo = pr["out_hw"]
        dt = _dt(x)
        out = pr.get("out")
        if out is None:
            out = torch.empty((N, Ho, Wo, Cout), device=x.device, dtype=x.dtype)
        d = ConvDesc(N, H, W, Cin, Ho, Wo, Cout, kh, kw, stride[0], stride[1], pad[0], pad[1], dil[0], dil[1], pr.get("act", ACT_NONE),
                     0, 0, _lib.PREC_FP32, dt)
        descs.append(d)
        outs.append(out)
        ws_bytes = max(ws_bytes, lib.diffsal_conv_igemm_ws_bytes(C.byref(d)))
        flops += 2.0 * N * Ho * Wo * Cout * kh * kw * Cin
        nbytes += _nb(x, w, out)
        notes.append(f"M={N * Ho * Wo} K={kh * kw * Cin} N={Cout} {kh}x{kw}")
    ws = torch.empty((ws_bytes // 4,), device=outs[0].device, dtype=torch.float32) if ws_bytes else None
    PA = C.c_void_p * n
    dp = (C.POINTER(ConvDesc) * n)(*[C.pointer(d) for d in descs])
    xs = PA(*[_pa(pr["x"], _dt(pr["x"])) for pr in problems])
    wsl = PA(*[_pa(pr["w"], _dt(pr["x"])) for pr in problems])
    bs = PA(*[_p(pr.get("bias")) for pr in problems])
    os_ = PA(*[_pa(o, _dt(o)) for o in outs])
    keep.extend([dp, xs, wsl, bs, os_])
    with _prof(tag, flops, nbytes, "group: " + " | ".join(notes) if PROFILE is not None else "") as pr_:
        _lib.check(lib.diffsal_conv_igemm_group(n, C.cast(dp, C.c_void_p), C.cast(xs, C.c_void_p), C.cast(wsl, C.c_void_p),
                                                C.cast(bs, C.c_void_p), C.cast(os_, C.c_void_p), _p(ws), ws_bytes, _stream()),
                   "conv_igemm_group")
        if PROFILE is not None:
            pr_.kernel = lib.diffsal_last_gemm_kernel().decode()
    return outs


def linear(x: Tensor, w: Tensor, bias: Optional[Tensor] = None, *, act: int = ACT_NONE,
           residual: Optional[Tensor] = None, tag: str = "K10", out_f32: bool = False) -> Tensor:
    """Token GEMM: x [..., K] @ w[N, K]^T (+bias, act, +residual) through the same MFMA kernel.  ``out_f32`` (16-bit storage
    only, no residual, K % 192 == 0): the result is fp32 -- the sums are not rounded to the storage type (``diffsal_linear_f32out``)."""
    lead = x.shape[:-1]
    M = 1
    for s in lead:
        M *= s
    if out_f32:
        if x.dtype == torch.float32 or residual is not None:
            raise RuntimeError("linear(out_f32=True): 16-bit storage in, no residual")
        lib = _lib.load()
        K, N = x.shape[-1], w.shape[0]
        dt = _dt(x)
        d = ConvDesc(1, 1, M, K, 1, M, N, 1, 1, 1, 1, 0, 0, 1, 1, act, 0, 0, _lib.PREC_FP32, dt)
        out = torch.empty((M, N), device=x.device, dtype=torch.float32)
        xc = x.reshape(M, K)
        with _prof(tag, 2.0 * M * N * K, _nb(xc, w, out), f"M={M} K={K} N={N} 1x1 (fp32 out)" if PROFILE is not None else "") as pr:
            _lib.check(lib.diffsal_linear_f32out(C.byref(d), _pa(xc, dt), _pa(w, dt), _p(bias), _p(out), _stream()), "linear_f32out")
            if PROFILE is not None:
                pr.kernel = lib.diffsal_last_gemm_kernel().decode()
        return out.reshape(*lead, N)
    y = conv_igemm(x.reshape(1, 1, M, x.shape[-1]), w, bias=bias, act=act, tag=tag,
                   residual=None if residual is None else residual.reshape(1, 1, M, w.shape[0]))
    return y.reshape(*lead, w.shape[0])


def linear_group(xs: Sequence[Tensor], ws: Sequence[Tensor], biases: Sequence[Optional[Tensor]], tag: str = "K10") -> List[Tensor]:
    """Up to four token GEMMs of ANY shapes in one launch (``conv_igemm_group``): x_i [..., K_i] @ w_i[N_i, K_i]^T + b_i.  fp32
    storage, exact arithmetic (the caller checks)."""
    probs, leads = [], []
    for x, w, b in zip(xs, ws, biases):
        lead = x.shape[:-1]
        M = 1
        for s_ in lead:
            M *= s_
        probs.append(dict(x=x.reshape(1, 1, M, x.shape[-1]), w=w, bias=b))
        leads.append(lead)
    outs = conv_igemm_group(probs, tag=tag)
    return [o.reshape(*lead, w.shape[0]) for o, lead, w in zip(outs, leads, ws)]


def linear_pair(x0: Tensor, x1: Tensor, w0: Tensor, w1: Tensor, b0: Optional[Tensor], b1: Optional[Tensor],
                tag: str = "K10") -> Tuple[Tensor, Tensor]:
    """Two token GEMMs of one shape in one launch: (x0 @ w0^T + b0, x1 @ w1^T + b1); x_i [..., K], w_i [N, K] (exact
    arithmetic or 16-bit storage; not the pre-split bf16x3 weight format)."""
    lib = _lib.load()
    if x0.shape != x1.shape or w0.shape != w1.shape or x0.dtype != x1.dtype or (b0 is None) != (b1 is None):
        raise RuntimeError("linear_pair: the two products must have one shape and storage type")
    lead, K = x0.shape[:-1], x0.shape[-1]
    M = 1
    for s_ in lead:
        M *= s_
    N = w0.shape[0]
    dt = _dt(x0)
    d = ConvDesc(1, 1, M, K, 1, M, N, 1, 1, 1, 1, 0, 0, 1, 1, ACT_NONE, 0, 0, _lib.PREC_FP32, dt)
    y0 = torch.empty((*lead, N), device=x0.device, dtype=x0.dtype)
    y1 = torch.empty_like(y0)
    ws_bytes = 2 * lib.diffsal_conv_igemm_ws_bytes(C.byref(d))
    ws = torch.empty((ws_bytes // 4,), device=x0.device, dtype=torch.float32) if ws_bytes else None
    x0c, x1c = x0.contiguous(), x1.contiguous()
    with _prof(tag, 4.0 * M * K * N, _nb(x0c, x1c, w0, w1, y0, y1), f"2 x (M={M} K={K} N={N}) 1x1 pair") as pr:
        _lib.check(lib.diffsal_linear_pair(C.byref(d), _pa(x0c, dt), _pa(x1c, dt), _pa(w0, dt), _pa(w1, dt), _p(b0), _p(b1),
                                           y0.data_ptr(), y1.data_ptr(), _p(ws), ws_bytes, _stream()), "linear_pair")
        if PROFILE is not None:
            pr.kernel = lib.diffsal_last_gemm_kernel().decode()
    return y0, y1


def pack_frames(vis: Tensor, noise: Optional[Tensor], t_out: Optional[int] = None,
                out_dtype: Optional[torch.dtype] = None) -> Tensor:
    """K6. vis [B,C,Tv,h,w] (NCTHW, fp32) + noise [B,h,w,C] -> [B,Tv+1,h,w,C] in the storage type of ``noise`` (or
    ``out_dtype`` when there is no noise map)."""
    lib = _lib.load()
    B, Cc, Tv, h, w = vis.shape
    Tout = t_out if t_out is not None else Tv + (1 if noise is not None else 0)
    odt = noise.dtype if noise is not None else (out_dtype or torch.float32)
    out = torch.empty((B, Tout, h, w, Cc), device=vis.device, dtype=odt)
    dt = _dt(out)
    with _prof("K6", 0.0, _nb(vis, noise, out)):
        _lib.check(lib.diffsal_pack_frames(_p(vis), _pa(noise, dt), out.data_ptr(), B, Cc, Tv, Tout, h * w, dt, _stream()),
                   "pack_frames")
    return out


def pack_frames_multi(vis: Sequence[Tensor], noise: Sequence[Optional[Tensor]], out_dtype: torch.dtype) -> list:
    """K6 for several stages in one launch: vis[j] [B,C_j,Tv,h_j,w_j] (fp32) + noise[j] [B,h_j,w_j,C_j] or None ->
    [B,Tv+1 (or Tv),h_j,w_j,C_j] each, all in ``out_dtype`` (the noise maps must already have that type)."""
    lib = _lib.load()
    n = len(vis)
    B = vis[0].shape[0]
    outs, Cs, Tvs, Touts, hws = [], [], [], [], []
    for v, nz in zip(vis, noise):
        b, Cc, Tv, h, w = v.shape
        if b != B or v.dtype != torch.float32 or not v.is_contiguous() or (nz is not None and (nz.dtype != out_dtype or not nz.is_contiguous())):
            raise RuntimeError("pack_frames_multi: one batch size, contiguous fp32 features, noise maps in the output type")
        Tout = Tv + (1 if nz is not None else 0)
        outs.append(torch.empty((B, Tout, h, w, Cc), device=v.device, dtype=out_dtype))
        Cs.append(Cc); Tvs.append(Tv); Touts.append(Tout); hws.append(h * w)
    dt = _dt(outs[0])
    arr = lambda xs: (C.c_void_p * n)(*[None if x is None else x.data_ptr() for x in xs])
    ints = lambda xs: (C.c_int * n)(*xs)
    with _prof("K6", 0.0, sum(_nb(v, nz, o) for v, nz, o in zip(vis, noise, outs))):
        _lib.check(lib.diffsal_pack_frames_multi(arr(vis), arr(noise), arr(outs), n, B, ints(Cs), ints(Tvs), ints(Touts), ints(hws),
                                                 dt, _stream()), "pack_frames_multi")
    return outs


def resize_bilinear(x: Tensor, H: int, W: int, tag: str = "K12-up") -> Tensor:
    lib = _lib.load()
    N, h, w, Cc = x.shape
    out = torch.empty((N, H, W, Cc), device=x.device, dtype=x.dtype)
    dt = _dt(x)
    with _prof(tag, 0.0, _nb(x, out)):
        _lib.check(lib.diffsal_resize_bilinear(_pa(x, dt), out.data_ptr(), N, h, w, H, W, Cc, dt, _stream()), "resize_bilinear")
    return out


def resize_sum(xs: Sequence[Tensor], H: int, W: int) -> Tensor:
    """sum_i bilinear(xs[i] -> (H, W)), NHWC, accumulated in list order."""
    lib = _lib.load()
    n = len(xs)
    N, _, _, Cc = xs[0].shape
    dt = _dt(xs[0])
    ptrs = (C.c_void_p * n)(*[_pa(x, dt) for x in xs])
    hs = (C.c_int * n)(*[x.shape[1] for x in xs])
    ws = (C.c_int * n)(*[x.shape[2] for x in xs])
    out = torch.empty((N, H, W, Cc), device=xs[0].device, dtype=xs[0].dtype)
    with _prof("K13-up", 0.0, _nb(*xs) + _nb(out)):
        _lib.check(lib.diffsal_resize_sum(ptrs, hs, ws, n, out.data_ptr(), N, H, W, Cc, dt, _stream()), "resize_sum")
    return out


def tapsum(ys: Sequence[Tensor], H: int, W: int, Cout: int, dil: int = 1, bias: Optional[Tensor] = None,
           scale: Optional[Tensor] = None, shift: Optional[Tensor] = None, act: int = ACT_NONE, tag: str = "K12-tap",
           head: Optional[Tuple[Tensor, Tensor]] = None) -> Tensor:
    """conv3x3(dil)(sum_i bilinear(z_i -> (H, W))) from the tap products ys[i] = z_i x Wcat^T  [N, h_i, w_i, 9*Cout]
    (``tap_weight``): out NHWC [N,H,W,Cout] = act(scale * (bias + gathered sum) + shift).  See csrc/tapsum.hip.
    ``head = (w [Cout], b [1])``: MLPHead folded into the epilogue -> fp32 [N,H,W,1] = sigmoid(b + out . w); the Cout-channel
    map is never stored."""
    lib = _lib.load()
    n = len(ys)
    N = ys[0].shape[0]
    dt = _dt(ys[0])
    for y in ys:
        if y.shape[-1] != 9 * Cout or not y.is_contiguous() or y.dtype != ys[0].dtype:
            raise RuntimeError(f"tapsum: every source must be a contiguous [N,h,w,{9 * Cout}] tensor of one dtype, got {tuple(y.shape)}")
    ptrs = (C.c_void_p * n)(*[_pa(y, dt) for y in ys])
    hs = (C.c_int * n)(*[y.shape[1] for y in ys])
    ws = (C.c_int * n)(*[y.shape[2] for y in ys])
    if head is not None:
        out = torch.empty((N, H, W, 1), device=ys[0].device, dtype=torch.float32)
        with _prof(tag, 2.0 * 36.0 * n * N * H * W * Cout + 2.0 * N * H * W * Cout, _nb(*ys) + _nb(out)):
            _lib.check(lib.diffsal_tapsum_head(ptrs, hs, ws, n, N, H, W, Cout, dil, _p(bias), _p(scale), _p(shift), act,
                                               _p(head[0]), _p(head[1]), out.data_ptr(), dt, _stream()), "tapsum_head")
        return out
    out = torch.empty((N, H, W, Cout), device=ys[0].device, dtype=ys[0].dtype)
    with _prof(tag, 2.0 * 36.0 * n * out.numel(), _nb(*ys) + _nb(out)):
        _lib.check(lib.diffsal_tapsum(ptrs, hs, ws, n, out.data_ptr(), N, H, W, Cout, dil, _p(bias), _p(scale), _p(shift), act, dt,
                                      _stream()), "tapsum")
    return out


def tapsum_bwd(du: Tensor, h: int, w: int, dil: int, out: Optional[Tensor] = None) -> Tensor:
    """Adjoint of ``tapsum`` (no affine / activation) for one source: du [N,H,W,C] fp32 -> dy [N,h,w,9*C].  ``out`` lets the
    caller place the result inside a larger buffer (the sources of one GEMM share a row-concatenated matrix)."""
    lib = _lib.load()
    N, H, W, Cc = du.shape
    if du.dtype != torch.float32 or not du.is_contiguous():
        raise RuntimeError("tapsum_bwd: fp32 contiguous NHWC gradient expected")
    if out is None:
        out = torch.empty((N, h, w, 9 * Cc), device=du.device, dtype=torch.float32)
    elif tuple(out.shape) != (N, h, w, 9 * Cc) or not out.is_contiguous() or out.dtype != torch.float32:
        raise RuntimeError(f"tapsum_bwd: out must be a contiguous fp32 [{N},{h},{w},{9 * Cc}] tensor")
    ws = torch.empty((int(lib.diffsal_tapsum_bwd_ws_bytes(N, W, Cc, h)) // 4,), device=du.device, dtype=torch.float32)
    with _prof("tapsum_bwd", 0.0, _nb(du) + 2 * _nb(ws) + _nb(out)):
        _lib.check(lib.diffsal_tapsum_bwd(du.data_ptr(), out.data_ptr(), ws.data_ptr(), N, H, W, Cc, h, w, dil, _stream()),
                   "tapsum_bwd")
    return out


def tap_weight(w: Tensor) -> Tensor:
    """Conv2d weight [Cout, Cin, 3, 3] -> the [9*Cout, Cin] matrix of the nine 1x1 channel mixings (row = tap * Cout + co),
    already in the GEMM's packed k order (a 1x1 weight is its own packed form).  Parameter-layout transform."""
    co, ci, kh, kw = w.shape
    if (kh, kw) != (3, 3):
        raise RuntimeError("tap_weight: 3x3 kernels only")
    return pack_conv_weight(w.detach().permute(2, 3, 0, 1).reshape(9 * co, ci, 1, 1).contiguous())


def audio_fuse(a_small: Tensor, x: Tensor, h: int, w: int) -> Tensor:
    """K7. a_small [B*T, h*w, C] (contiguous, or a channel slice of a wider row: the stages' align products side by side),
    x [B,T,H,W,C] -> fused audio in the reference's [B,C,T,H,W] order."""
    lib = _lib.load()
    B, T, H, W, Cc = x.shape
    out = torch.empty((B, Cc, T, H, W), device=x.device, dtype=x.dtype)
    dt = _dt(x)
    ld = a_small.stride(-2)
    if (not a_small.is_cuda or DTYPE_CODES.get(a_small.dtype) != dt or a_small.shape != (B * T, h * w, Cc) or a_small.stride(-1) != 1
            or a_small.stride(0) != h * w * ld or a_small.data_ptr() % 16):
        raise ValueError(f"audio_fuse: a_small must be [B*T, h*w, C] rows of one pitch in x's storage type, got {tuple(a_small.shape)} "
                         f"strides {a_small.stride()} {a_small.dtype}")
    with _prof("K7", 0.0, _nb(x, out) + a_small.numel() * a_small.element_size()):
        _lib.check(lib.diffsal_audio_fuse(a_small.data_ptr(), ld, _pa(x, dt), out.data_ptr(), B, T, H, W, Cc, h, w, dt, _stream()),
                   "audio_fuse")
    return out


def layernorm(x: Tensor, gamma: Tensor, beta: Tensor, eps: float = 1e-5) -> Tensor:
    lib = _lib.load()
    Cc = x.shape[-1]
    M = x.numel() // Cc
    out = torch.empty_like(x)
    dt = _dt(x)
    with _prof("K8", 0.0, _nb(x, out)):
        _lib.check(lib.diffsal_layernorm(_pa(x, dt), _p(gamma), _p(beta), out.data_ptr(), M, Cc, eps, dt, _stream()), "layernorm")
    return out


def dwconv3_ln(x: Tensor, w9: Tensor, gamma: Tensor, beta: Tensor, eps: float = 1e-5) -> Tensor:
    """x NHWC [N,H,W,C] -> tokens [N, H*W, C]."""
    lib = _lib.load()
    N, H, W, Cc = x.shape
    out = torch.empty((N, H * W, Cc), device=x.device, dtype=x.dtype)
    dt = _dt(x)
    with _prof("K9", 18.0 * N * H * W * Cc, _nb(x, out)):
        _lib.check(lib.diffsal_dwconv3_ln(_pa(x, dt), _p(w9), _p(gamma), _p(beta), out.data_ptr(), N, H, W, Cc, eps, dt,
                                          _stream()), "dwconv3_ln")
    return out


def qkv_prep(xq: Tensor, w9: Tensor, gq: Tensor, bq: Tensor, xk: Tensor, xv: Tensor, wk: Tensor, wv: Tensor, gk: Tensor,
             bk: Tensor, gv: Tensor, bv: Tensor, k: int, eps: float = 1e-5, pre_ln=None):
    """``dwconv3_ln(xq, ...)`` and ``dwpool_ln_kv(xk, xv, ...)`` of one transformer block in one launch -> (q, k, v) tokens.
    ``pre_ln = (gamma, beta, eps, ln_k)``: xq / xv (and xk when ``ln_k``) are un-normalised and pass through that LayerNorm
    as they are loaded (the block's ``norm``), bit-equal with a separate ``layernorm`` launch."""
    lib = _lib.load()
    N, H, W, Cc = xq.shape
    if xk.shape != xq.shape or xv.shape != xq.shape or xk.dtype != xq.dtype or xv.dtype != xq.dtype:
        raise RuntimeError("qkv_prep: the three inputs must share shape and storage type")
    gh, gw = (H - k) // k + 1, (W - k) // k + 1
    oq = torch.empty((N, H * W, Cc), device=xq.device, dtype=xq.dtype)
    ok = torch.empty((N, gh * gw, Cc), device=xq.device, dtype=xq.dtype)
    ov = torch.empty_like(ok)
    dt = _dt(xq)
    same = xk.data_ptr() == xv.data_ptr() == xq.data_ptr()
    with _prof("K9", 22.0 * N * H * W * Cc, _nb(xq, oq, ok, ov) + (0.0 if same else _nb(xk))):
        pg, pb, pe, plk = (pre_ln[0], pre_ln[1], float(pre_ln[2]), int(bool(pre_ln[3]))) if pre_ln is not None else (None, None, 0.0, 0)
        _lib.check(lib.diffsal_qkv_prep(_pa(xq, dt), _p(w9), _p(gq), _p(bq), oq.data_ptr(), _pa(xk, dt), _pa(xv, dt), _p(wk),
                                        _p(wv), _p(gk), _p(bk), _p(gv), _p(bv), ok.data_ptr(), ov.data_ptr(), N, H, W, Cc, k,
                                        eps, _p(pg), _p(pb), pe, plk, dt, _stream()), "qkv_prep")
    return oq, ok, ov


def kv_prep(xk: Tensor, xv: Tensor, wk: Tensor, wv: Tensor, gk: Tensor, bk: Tensor, gv: Tensor, bv: Tensor, k: int,
            eps: float = 1e-5, pre_ln=None):
    """The pooled key / value branch of ``qkv_prep`` alone (the query branch lives in ``block_front``): depthwise k x k
    stride-k pooling + LayerNorm of xk / xv [N,H,W,C] -> (k, v) [N, gh*gw, C].  ``pre_ln = (gamma, beta, eps, ln_k)`` applies
    the block's first LayerNorm to xv (and to xk when ``ln_k``) as the tokens are loaded."""
    lib = _lib.load()
    N, H, W, Cc = xv.shape
    if xk.shape != xv.shape or xk.dtype != xv.dtype:
        raise RuntimeError("kv_prep: the two inputs must share shape and storage type")
    gh, gw = (H - k) // k + 1, (W - k) // k + 1
    ok = torch.empty((N, gh * gw, Cc), device=xv.device, dtype=xv.dtype)
    ov = torch.empty_like(ok)
    dt = _dt(xv)
    with _prof("K9", 4.0 * N * H * W * Cc, _nb(xv, ok, ov) + (0.0 if xk.data_ptr() == xv.data_ptr() else _nb(xk))):
        pg, pb, pe, plk = (pre_ln[0], pre_ln[1], float(pre_ln[2]), int(bool(pre_ln[3]))) if pre_ln is not None else (None, None, 0.0, 0)
        _lib.check(lib.diffsal_qkv_prep(None, None, None, None, None, _pa(xk, dt), _pa(xv, dt), _p(wk), _p(wv), _p(gk), _p(bk),
                                        _p(gv), _p(bv), ok.data_ptr(), ov.data_ptr(), N, H, W, Cc, k, eps, _p(pg), _p(pb), pe,
                                        plk, dt, _stream()), "qkv_prep(kv only)")
    return ok, ov


def kv_prep_proj(xk: Tensor, xv: Tensor, wk: Tensor, wv: Tensor, gk: Tensor, bk: Tensor, gv: Tensor, bv: Tensor, k: int, eps: float,
                 pre_ln, lin_k, lin_v):
    """``kv_prep`` with proj_k / proj_v folded in (C = 96 / 192): the whole pooled key / value branch of a block in one launch.
    lin_k / lin_v = (weight [C, C] in the storage type, bias)."""
    lib = _lib.load()
    N, H, W, Cc = xv.shape
    gh, gw = (H - k) // k + 1, (W - k) // k + 1
    ok = torch.empty((N, gh * gw, Cc), device=xv.device, dtype=xv.dtype)
    ov = torch.empty_like(ok)
    dt = _dt(xv)
    with _prof("K9", 4.0 * N * H * W * Cc + 4.0 * N * gh * gw * Cc * Cc,
               _nb(xv, ok, ov, lin_k[0], lin_v[0]) + (0.0 if xk.data_ptr() == xv.data_ptr() else _nb(xk))):
        _lib.check(lib.diffsal_kv_prep_proj(_pa(xk, dt), _pa(xv, dt), _p(wk), _p(wv), _p(gk), _p(bk), _p(gv), _p(bv),
                                            _pa(lin_k[0], dt), _p(lin_k[1]), _pa(lin_v[0], dt), _p(lin_v[1]), ok.data_ptr(),
                                            ov.data_ptr(), N, H, W, Cc, k, eps, _p(pre_ln[0]), _p(pre_ln[1]), float(pre_ln[2]),
                                            int(bool(pre_ln[3])), dt, _stream()), "kv_prep_proj")
    return ok, ov


def block_front_supported(C: int, heads: int, Lk: int, dtype: torch.dtype = torch.float32) -> bool:
    """csrc/tblock.hip: C = 96 on every storage type, C = 192 on 16-bit storage (there Wq alone fits the LDS next to the tile)."""
    return heads == 2 and 0 < Lk <= 32 and (C == 96 or (C == 192 and dtype != torch.float32))


def block_front(x: Tensor, k: Tensor, v: Tensor, norm1, w9: Tensor, norm_q, lin_q, lin_p, heads: int, scale: float) -> Tensor:
    """Fused first half of a TransformerBlock (csrc/tblock.hip): LayerNorm -> depthwise 3x3 -> LayerNorm -> proj_q -> attention
    over the projected keys / values k, v [N, Lk, C] (-> + proj + residual on fp32 storage).  x [N,H,W,C]; norm1 / norm_q =
    (gamma, beta, eps); lin_q = (weight [C,C] in the storage type, bias); lin_p likewise (fp32 storage) or None.
    Returns x1 = x + proj(o) on fp32 storage, the attention output o on 16-bit storage."""
    lib = _lib.load()
    N, H, W, Cc = x.shape
    Lk = k.shape[1]
    dt = _dt(x)
    out = torch.empty_like(x)
    with_p = x.dtype == torch.float32
    if with_p and lin_p is None:
        raise RuntimeError("block_front: fp32 storage needs the output projection")
    M = N * H * W
    fl = 2.0 * M * Cc * Cc * (2 if with_p else 1) + 4.0 * M * Lk * Cc + 22.0 * M * Cc
    with _prof("K10f", fl, _nb(x, out, k, v, lin_q[0]) + (_nb(lin_p[0]) if with_p else 0.0), f"block_front M={M} C={Cc}",
               kernel=f"block_front_kernel<{'float' if with_p else '16-bit'}, {Cc}>"):
        _lib.check(lib.diffsal_block_front(_pa(x, dt), _pa(k, dt), _pa(v, dt), _p(norm1[0]), _p(norm1[1]), float(norm1[2]),
                                           _p(w9), _p(norm_q[0]), _p(norm_q[1]), float(norm_q[2]), _pa(lin_q[0], dt),
                                           _p(lin_q[1]), _pa(lin_p[0], dt) if with_p else None,
                                           _p(lin_p[1]) if with_p else None, out.data_ptr(), N, H, W, Cc, Lk, heads,
                                           float(scale), dt, _stream()), "block_front")
    return out


def dwpool_ln_kv(xk: Tensor, xv: Tensor, wk: Tensor, wv: Tensor, gk: Tensor, bk: Tensor, gv: Tensor, bv: Tensor,
                 k: int, eps: float = 1e-5):
    lib = _lib.load()
    N, H, W, Cc = xv.shape
    gh, gw = (H - k) // k + 1, (W - k) // k + 1
    ok = torch.empty((N, gh * gw, Cc), device=xv.device, dtype=xv.dtype)
    ov = torch.empty_like(ok)
    dt = _dt(xv)
    with _prof("K9", 4.0 * N * H * W * Cc, _nb(xk, ok, ov) + (0.0 if xk.data_ptr() == xv.data_ptr() else _nb(xv))):
        _lib.check(lib.diffsal_dwpool_ln_kv(_pa(xk, dt), _pa(xv, dt), _p(wk), _p(wv), _p(gk), _p(bk), _p(gv), _p(bv),
                                            ok.data_ptr(), ov.data_ptr(), N, H, W, Cc, k, eps, dt, _stream()), "dwpool_ln_kv")
    return ok, ov


def attention(q: Tensor, k: Tensor, v: Tensor, heads: int, scale: float) -> Tensor:
    lib = _lib.load()
    N, Lq, Cc = q.shape
    Lk = k.shape[1]
    o = torch.empty_like(q)
    dt = _dt(q)
    with _prof("K11", 4.0 * N * Lq * Lk * Cc, _nb(q, k, v, o)):
        _lib.check(lib.diffsal_attention(_pa(q, dt), _pa(k, dt), _pa(v, dt), o.data_ptr(), N, Lq, Lk, Cc, heads, scale, dt,
                                         _stream()), "attention")
    return o


def head_sigmoid(x: Tensor, w: Tensor, bias: Tensor) -> Tensor:
    """x NHWC [N,H,W,C] (any storage type) -> fp32 [N,H,W,1]."""
    lib = _lib.load()
    N, H, W, Cc = x.shape
    out = torch.empty((N, H, W, 1), device=x.device, dtype=torch.float32)
    dt = _dt(x)
    with _prof("K14-head", 2.0 * N * H * W * Cc, _nb(x, out)):
        _lib.check(lib.diffsal_head_sigmoid(_pa(x, dt), _p(w), _p(bias), _p(out), N * H * W, Cc, dt, _stream()), "head_sigmoid")
    return out


def axpbypcz(x: Tensor, a: float, y: Optional[Tensor] = None, b: float = 0.0, z: Optional[Tensor] = None,
             c: float = 0.0, out: Optional[Tensor] = None) -> Tensor:
    lib = _lib.load()
    if out is None:
        out = torch.empty_like(x)
    with _prof("K15", 0.0, _nb(x, y, z, out)):
        _lib.check(lib.diffsal_axpbypcz(_p(x), _p(y), _p(z), float(a), float(b), float(c), _p(out), x.numel(), _stream()),
                   "axpbypcz")
    return out


# ------------------------------------------------------------------------------------------------
# training-side kernels
# ------------------------------------------------------------------------------------------------
def conv_wgrad(x: Tensor, dy: Tensor, *, kh: int = 1, kw: int = 1, stride=(1, 1), pad=(0, 0), dil=(1, 1),
               want_bias: bool = False):
    """dW in the packed layout [Cout, kh*kw*Cin] for the conv  y = conv_igemm(x, w_packed, ...).  x NHWC, dy NHWC.
    want_bias: also return db [Cout] (column sums of dy, accumulated by the same kernel) -> (dW, db)."""
    lib = _lib.load()
    N, H, W, Cin = x.shape
    _, Ho, Wo, Cout = dy.shape
    d = ConvDesc(N, H, W, Cin, Ho, Wo, Cout, kh, kw, stride[0], stride[1], pad[0], pad[1], dil[0], dil[1], 0, 0, 0, 0, 0)
    nws = lib.diffsal_conv_wgrad_ws_bytes(C.byref(d))
    ws = torch.empty((nws // 4,), device=x.device, dtype=torch.float32)
    dw = torch.empty((Cout, kh * kw * Cin), device=x.device, dtype=torch.float32)
    bpart = db = None
    if want_bias:                # per-split fp64 column sums of dy; the library finishes them into db (no separate reduction launch)
        splits = lib.diffsal_conv_wgrad_splits(C.byref(d))
        bpart = torch.empty((splits, Cout), device=x.device, dtype=torch.float64)
        db = torch.empty((Cout,), device=x.device, dtype=torch.float32)
    with _prof("wgrad", 2.0 * N * Ho * Wo * Cout * kh * kw * Cin, _nb(x, dy, dw),
               f"M={N * Ho * Wo} K={kh * kw * Cin} N={Cout} {kh}x{kw}" if PROFILE is not None else ""):
        _lib.check(lib.diffsal_conv_wgrad(C.byref(d), _p(x), _p(dy), _p(dw), bpart.data_ptr() if want_bias else None,
                                          _p(db), _p(ws), nws, _stream()), "conv_wgrad")
    if want_bias:
        return dw, db
    return dw


@_classed("reduce")
def colsum(dy: Tensor, seg_rows: Optional[int] = None) -> Tensor:
    """Column sums of dy [M, C] per segment of seg_rows rows -> [M // seg_rows, C] (one segment: [1, C])."""
    lib = _lib.load()
    Cc = dy.shape[-1]
    M = dy.numel() // Cc
    seg = M if seg_rows is None else seg_rows
    out = torch.empty((M // seg, Cc), device=dy.device, dtype=torch.float32)
    nws = (M // seg) * 512 * Cc * 8
    ws = torch.empty((nws // 8,), device=dy.device, dtype=torch.float64)
    _lib.check(lib.diffsal_colsum(_p(dy), _p(out), M, Cc, seg, ws.data_ptr(), nws, _stream()), "colsum")
    return out


@_classed("act-bwd")
def act_bwd(dy: Tensor, ref: Tensor, mode: int) -> Tensor:
    """dy * act'(.): mode 1 ReLU (ref = y), 2 GELU (ref = pre-activation), 3 sigmoid (ref = y)."""
    lib = _lib.load()
    dx = torch.empty_like(dy)
    _lib.check(lib.diffsal_act_bwd(_p(dy), _p(ref), _p(dx), dy.numel(), mode, _stream()), "act_bwd")
    return dx


def relu_bwd(dy: Tensor, y: Tensor) -> Tensor:
    return act_bwd(dy, y, 1)


@_classed("norm-train")
def rowstats(x: Tensor, seg_rows: int, mode: int = 0, dy=None, y=None, mu=None, rs=None, gamma=None, beta=None,
             stat_per_seg: bool = False) -> Tensor:
    """Per-channel dual sums over the rows of each segment -> [segs, 2, C] (float64, chunk partials already summed)."""
    lib = _lib.load()
    Cc = x.shape[-1]
    M = x.numel() // Cc
    chunks = lib.diffsal_rowstats_chunks(M, seg_rows)
    part = torch.empty((M // seg_rows, chunks, 2, Cc), device=x.device, dtype=torch.float64)
    _lib.check(lib.diffsal_rowstats(_p(x), _p(dy), _p(y), _p(mu), _p(rs), _p(gamma), _p(beta), part.data_ptr(), M, Cc, seg_rows,
                                    mode, int(stat_per_seg), _stream()), "rowstats")
    return reduce_partials(part, M // seg_rows, chunks, 2 * Cc, f64=True).view(M // seg_rows, 2, Cc)


@_classed("reduce")
def reduce_partials(part: Tensor, segs: int, chunks: int, width: int, f64: bool = False) -> Tensor:
    """Sum fp64 partials [segs, chunks, width] over the chunks (fixed order) -> [segs, width] (fp64 or fp32)."""
    lib = _lib.load()
    out = torch.empty((segs, width), device=part.device, dtype=torch.float64 if f64 else torch.float32)
    _lib.check(lib.diffsal_reduce_partials(part.data_ptr(), out.data_ptr(), segs, chunks, width, int(f64), _stream()),
               "reduce_partials")
    return out


@_classed("norm-train")
def norm_finalize_fwd(sums: Tensor, gamma: Tensor, beta: Tensor, groups: int, n: float, eps: float, bn=None):
    """sums [segs, 2, C] fp64 -> (mu, rs, scale, shift) [segs, C] (+ (mean, var) [C] when ``bn`` = (running_mean,
    running_var, momentum, unbias) is given: BatchNorm, running statistics updated in place)."""
    lib = _lib.load()
    segs, _, Cc = sums.shape
    mu, rs, scale, shift = (torch.empty((segs, Cc), device=sums.device, dtype=torch.float32) for _ in range(4))
    if bn is not None:
        mean, var = torch.empty((Cc,), device=sums.device), torch.empty((Cc,), device=sums.device)
        rm, rv, mom, unb = bn
        args = (_p(mean), _p(var), _p(rm), _p(rv), float(mom), float(unb))
    else:
        mean = var = None
        args = (None, None, None, None, 0.0, 1.0)
    _lib.check(lib.diffsal_norm_finalize_fwd(sums.data_ptr(), _p(gamma), _p(beta), _p(mu), _p(rs), _p(scale), _p(shift), segs,
                                             Cc, groups, float(n), float(eps), *args, _stream()), "norm_finalize_fwd")
    return mu, rs, scale, shift, mean, var


@_classed("norm-train")
def norm_finalize_bwd(t: Tensor, gamma: Tensor, rs: Tensor, groups: int, n: float):
    """t [segs, 2, C] fp64 -> (dgamma [C], dbeta [C], k1, k2, k3 [segs, C])."""
    lib = _lib.load()
    segs, _, Cc = t.shape
    dg, db = torch.empty((Cc,), device=t.device), torch.empty((Cc,), device=t.device)
    k1, k2, k3 = (torch.empty((segs, Cc), device=t.device, dtype=torch.float32) for _ in range(3))
    _lib.check(lib.diffsal_norm_finalize_bwd(t.data_ptr(), _p(gamma), _p(rs), _p(dg), _p(db), _p(k1), _p(k2), _p(k3), segs, Cc,
                                             groups, float(n), _stream()), "norm_finalize_bwd")
    return dg, db, k1, k2, k3


@_classed("norm-train")
def affine_act(x: Tensor, scale: Tensor, shift: Tensor, seg_rows: int, act: int) -> Tensor:
    lib = _lib.load()
    Cc = x.shape[-1]
    out = torch.empty_like(x)
    _lib.check(lib.diffsal_affine_act(_p(x), _p(scale), _p(shift), _p(out), x.numel() // Cc, Cc, seg_rows, act,
                                      _stream()), "affine_act")
    return out


@_classed("norm-train")
def norm_bwd_apply(x, dy, y, mu, rs, gamma, beta, k1, k2, k3, seg_rows: int, mode: int) -> Tensor:
    lib = _lib.load()
    Cc = x.shape[-1]
    dx = torch.empty_like(x)
    _lib.check(lib.diffsal_norm_bwd_apply(_p(x), _p(dy), _p(y), _p(mu), _p(rs), _p(gamma), _p(beta), _p(k1), _p(k2),
                                          _p(k3), _p(dx), x.numel() // Cc, Cc, seg_rows, mode, _stream()),
               "norm_bwd_apply")
    return dx


@_classed("K8")
def layernorm_multi(xs, gammas, betas, epss):
    """LayerNorm of up to three fp32 tensors of one width in one launch (include/diffsal.h) -> list of outputs."""
    lib = _lib.load()
    n, Cc = len(xs), xs[0].shape[-1]
    xs = [_cont(x) for x in xs]
    outs = [torch.empty_like(x) for x in xs]
    M = (C.c_int * n)(*[x.numel() // Cc for x in xs])
    eps = (C.c_float * n)(*[float(e) for e in epss])
    ptrs = lambda ts: (C.c_void_p * n)(*[t.data_ptr() for t in ts])
    _lib.check(lib.diffsal_layernorm_multi(ptrs(xs), ptrs([_cont(g) for g in gammas]), ptrs([_cont(b) for b in betas]), ptrs(outs), M, n, Cc,
                                           eps, _stream()), "layernorm_multi")
    return outs


@_classed("K8-bwd")
def layernorm_bwd_multi(xs, dys, gammas, epss):
    """-> (dxs, dgammas, dbetas) of ``layernorm_multi``: one launch + one reduction."""
    lib = _lib.load()
    n, Cc = len(xs), xs[0].shape[-1]
    xs, dys = [_cont(x) for x in xs], [_cont(d) for d in dys]
    rows = [x.numel() // Cc for x in xs]
    blocks = lib.diffsal_layernorm_bwd_blocks(max(rows), Cc)
    part = torch.empty((n, blocks, 2, Cc), device=xs[0].device, dtype=torch.float64)
    dxs = [torch.empty_like(x) for x in xs]
    M = (C.c_int * n)(*rows)
    eps = (C.c_float * n)(*[float(e) for e in epss])
    ptrs = lambda ts: (C.c_void_p * n)(*[t.data_ptr() for t in ts])
    _lib.check(lib.diffsal_layernorm_bwd_multi(ptrs(xs), ptrs(dys), ptrs([_cont(g) for g in gammas]), ptrs(dxs), part.data_ptr(), M, n, Cc,
                                               eps, _stream()), "layernorm_bwd_multi")
    s_ = reduce_partials(part, n, blocks, 2 * Cc).view(n, 2, Cc)
    return dxs, [s_[t, 0] for t in range(n)], [s_[t, 1] for t in range(n)]


@_classed("K8-bwd")
def layernorm_bwd(x: Tensor, dy: Tensor, gamma: Tensor, eps: float = 1e-5, add: Optional[Tensor] = None):
    """-> (dx (+ add), dgamma, dbeta)."""
    lib = _lib.load()
    Cc = x.shape[-1]
    M = x.numel() // Cc
    blocks = lib.diffsal_layernorm_bwd_blocks(M, Cc)
    part = torch.empty((blocks, 2, Cc), device=x.device, dtype=torch.float64)
    dx = torch.empty_like(x)
    _lib.check(lib.diffsal_layernorm_bwd(_p(x), _p(dy), _p(gamma), _p(add), _p(dx), part.data_ptr(), M, Cc, eps, _stream()),
               "layernorm_bwd")
    s = reduce_partials(part, 1, blocks, 2 * Cc).view(2, Cc)
    return dx, s[0], s[1]


@_classed("dropout")
def dropout(x: Tensor, p: float, seed: int) -> Tensor:
    lib = _lib.load()
    out = torch.empty_like(x)
    _lib.check(lib.diffsal_dropout(_p(x), _p(out), x.numel(), float(p), int(seed) & (2 ** 64 - 1), _stream()), "dropout")
    return out


_GELU_CONST: dict = {}


def gelu(x: Tensor) -> Tensor:
    """Stand-alone erf-GELU (training path keeps the pre-activation for the backward)."""
    Cc = x.shape[-1]
    key = (Cc, x.device)
    cst = _GELU_CONST.get(key)
    if cst is None:                                   # identity affine of the shared kernel: built once per width and device
        cst = _GELU_CONST[key] = (torch.ones((1, Cc), device=x.device, dtype=torch.float32),
                                  torch.zeros((1, Cc), device=x.device, dtype=torch.float32))
    return affine_act(x, cst[0], cst[1], x.numel() // Cc, ACT_GELU)


@_classed("K9-train")
def dwconv(x: Tensor, w: Tensor, k: int, stride: int, pad: int) -> Tensor:
    """Depthwise k x k conv on NHWC x with w [k*k, C]."""
    lib = _lib.load()
    N, H, W, Cc = x.shape
    Ho, Wo = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
    out = torch.empty((N, Ho, Wo, Cc), device=x.device, dtype=torch.float32)
    _lib.check(lib.diffsal_dwconv(_p(x), _p(w), _p(out), N, H, W, Cc, k, stride, pad, _stream()), "dwconv")
    return out


@_classed("K9-train")
def dwconv_bwd(x: Tensor, w: Tensor, du: Tensor, k: int, stride: int, pad: int, need_dx=True, need_dw=True):
    lib = _lib.load()
    N, H, W, Cc = x.shape
    dx = dw = None
    if need_dx:
        dx = torch.empty_like(x)
        _lib.check(lib.diffsal_dwconv_bwd_data(_p(du), _p(w), _p(dx), N, H, W, Cc, k, stride, pad, _stream()),
                   "dwconv_bwd_data")
    if need_dw:
        chunks = lib.diffsal_dwconv_bwd_weight_chunks(N, H, W, k, stride, pad)
        part = torch.empty((k * k, chunks, Cc), device=x.device, dtype=torch.float64)
        _lib.check(lib.diffsal_dwconv_bwd_weight(_p(x), _p(du), part.data_ptr(), N, H, W, Cc, k, stride, pad, _stream()),
                   "dwconv_bwd_weight")
        dw = reduce_partials(part, k * k, chunks, Cc)
    return dx, dw


@_classed("wgrad")
def wgrad_segmented(x: Tensor, dy: Tensor, segments: int) -> Tensor:
    """out[s] = dy_s^T @ x_s for the ``segments`` equal row blocks of x [M, K] and dy [M, Cout] -> [segments, Cout, K]."""
    lib = _lib.load()
    M, K = x.shape
    Cout = dy.shape[1]
    if M % segments or dy.shape[0] != M:
        raise RuntimeError(f"wgrad_segmented: {M} rows do not split into {segments} segments")
    nws = lib.diffsal_wgrad_segmented_ws_bytes(segments, M // segments, K, Cout)
    ws = torch.empty((max(nws // 4, 4),), device=x.device, dtype=torch.float32)
    out = torch.empty((segments, Cout, K), device=x.device, dtype=torch.float32)
    _lib.check(lib.diffsal_wgrad_segmented(_p(x), _p(dy), _p(out), segments, M // segments, K, Cout, _p(ws), nws, _stream()),
               "wgrad_segmented")
    return out


@_classed("K11-bwd")
def attention_bwd(q: Tensor, k: Tensor, v: Tensor, dout: Tensor, heads: int, scale: float):
    """-> (dq, dk, dv).  Pass 1: per-query kernel (dq, P, dS); pass 2: dk = dS^T q, dv = P^T dout per image on the
    matrix cores (segmented weight-gradient GEMM), keeping each head's own column block."""
    lib = _lib.load()
    N, Lq, Cc = q.shape
    Lk = k.shape[1]
    d = Cc // heads
    ld = (heads * Lk + 3) // 4 * 4
    alloc = torch.empty if ld == heads * Lk else torch.zeros
    P = alloc((N * Lq, ld), device=q.device, dtype=torch.float32)
    dS = alloc((N * Lq, ld), device=q.device, dtype=torch.float32)
    dq = torch.empty_like(q)
    _lib.check(lib.diffsal_attention_bwd(_p(q), _p(k), _p(v), _p(dout), _p(dq), _p(P), _p(dS), N, Lq, Lk, Cc, heads, ld,
                                         scale, _stream()), "attention_bwd")
    if Cc % 32:
        raise RuntimeError(f"attention_bwd: C={Cc} must be a multiple of 32")
    dk_full = wgrad_segmented(q.reshape(N * Lq, Cc), dS, N)       # [N, ld, C]: rows (head, t) x all channels
    dv_full = wgrad_segmented(dout.reshape(N * Lq, Cc), P, N)
    pick = lambda full: torch.cat([full[:, h * Lk:(h + 1) * Lk, h * d:(h + 1) * d] for h in range(heads)], dim=2)
    return dq, pick(dk_full).contiguous(), pick(dv_full).contiguous()


@_classed("resize-bwd")
def resize_bilinear_bwd(dy: Tensor, h: int, w: int) -> Tensor:
    lib = _lib.load()
    N, H, W, Cc = dy.shape
    dx = torch.empty((N, h, w, Cc), device=dy.device, dtype=torch.float32)
    ws = torch.empty((N * h * W * Cc,), device=dy.device, dtype=torch.float32) if Cc % 4 == 0 else None
    _lib.check(lib.diffsal_resize_bilinear_bwd(_p(dy), _p(dx), N, h, w, H, W, Cc, _p(ws), ws.numel() * 4 if ws is not None else 0,
                                               _stream()), "resize_bilinear_bwd")
    return dx


@_classed("K6-bwd")
def unpack_frames(frames_grad: Tensor, Tv: int) -> Tensor:
    """[B,Tin,h,w,C] -> gradient of the NCTHW visual features [B,C,Tv,h,w]."""
    lib = _lib.load()
    B, Tin, h, w, Cc = frames_grad.shape
    out = torch.empty((B, Cc, Tv, h, w), device=frames_grad.device, dtype=torch.float32)
    _lib.check(lib.diffsal_unpack_frames(_p(frames_grad), _p(out), B, Cc, Tv, Tin, h * w, _stream()), "unpack_frames")
    return out


@_classed("head-bwd")
def head_bwd(y: Tensor, w: Tensor, s_out: Tensor, ds: Tensor):
    """-> (dy, dw [C], db [1])."""
    lib = _lib.load()
    Cc = y.shape[-1]
    M = y.numel() // Cc
    blocks = min(1024, max(1, M // 64))
    part = torch.empty((blocks, Cc + 1), device=y.device, dtype=torch.float64)
    dy = torch.empty_like(y)
    _lib.check(lib.diffsal_head_bwd(_p(y), _p(w), _p(s_out), _p(ds), _p(dy), part.data_ptr(), blocks, M, Cc, _stream()),
               "head_bwd")
    s = reduce_partials(part, 1, blocks, Cc + 1).view(-1)
    return dy, s[:Cc].contiguous(), s[Cc:].contiguous()


@_classed("K2-bwd")
def conv_in_bwd(x: Tensor, dy: Tensor):
    """-> (dw [C, 9], db [C]) of conv_in."""
    lib = _lib.load()
    B, _, H, W = x.shape
    Cc = dy.shape[-1]
    chunks = 128
    part = torch.empty((10, chunks, Cc), device=x.device, dtype=torch.float64)
    _lib.check(lib.diffsal_conv_in_bwd(_p(x), _p(dy), part.data_ptr(), B, H, W, Cc, chunks, _stream()), "conv_in_bwd")
    s = reduce_partials(part, 10, chunks, Cc)
    return s[:9].t().contiguous(), s[9].contiguous()


@_classed("K1-bwd")
def dense_small_bwd(x: Tensor, w: Tensor, dout: Tensor, swish_in: bool):
    """-> (dx, dw, db)."""
    lib = _lib.load()
    B, K = x.shape
    N = w.shape[0]
    dw, db, dx = torch.empty_like(w), torch.empty((N,), device=x.device), torch.empty_like(x)
    _lib.check(lib.diffsal_dense_small_bwd(_p(x), _p(w), _p(dout), _p(dw), _p(db), _p(dx), B, K, N, int(swish_in),
                                           _stream()), "dense_small_bwd")
    return dx, dw, db


@_classed("K7-bwd")
def audio_fuse_bwd(a_small: Tensor, x: Tensor, dout: Tensor, h: int, w: int):
    """-> (dx [B,T,H,W,C], da_small [B*T, h*w, C])."""
    lib = _lib.load()
    B, T, H, W, Cc = x.shape
    dx = torch.empty_like(x)
    up = H // h if (h != H and w != W) else 1
    # part[(b,t,ys,xs)][dyr][c]: the `up` output rows over one audio cell, summed below in a fixed order
    part = torch.empty((B * T * h * w * up, Cc), device=x.device, dtype=torch.float32)
    _lib.check(lib.diffsal_audio_fuse_bwd(_p(a_small), _p(x), _p(dout), _p(dx), _p(part), B, T, H, W, Cc, h, w, _stream()),
               "audio_fuse_bwd")
    da = part if up == 1 else colsum(part, up)
    return dx, da.reshape(a_small.shape)


# ---- K16 tail: loss, clip, optimizer on flat buffers (csrc/optim.hip) --------------------------------
def _reduce_scratch(dev) -> Tensor:
    return torch.empty((_lib.load().diffsal_reduce_blocks(),), device=dev, dtype=torch.float64)


@_classed("optim")
def mse_loss(pred: Tensor, target: Tensor, loss_scale: float, want_grad: bool = True):
    """-> (loss [1], dpred or None): loss = loss_scale * sum (pred-target)^2 (R/models/sal_losses.py:189-192)."""
    lib = _lib.load()
    if pred.shape != target.shape:
        raise RuntimeError(f"mse_loss: shapes differ: {tuple(pred.shape)} vs {tuple(target.shape)}")
    loss = torch.empty((1,), device=pred.device)
    dpred = torch.empty_like(pred) if want_grad else None
    part = _reduce_scratch(pred.device)
    _lib.check(lib.diffsal_mse_loss(_p(pred), _p(target), _p(dpred) if want_grad else None, _p(loss), part.data_ptr(),
                                    pred.numel(), float(loss_scale), _stream()), "mse_loss")
    return loss, dpred


@_classed("optim")
def grad_norm(flat_grad: Tensor, gscale: float = 1.0) -> Tensor:
    """-> [1] tensor gscale * ||flat_grad||_2 (stays on the device)."""
    lib = _lib.load()
    norm = torch.empty((1,), device=flat_grad.device)
    part = _reduce_scratch(flat_grad.device)
    _lib.check(lib.diffsal_grad_norm(_p(flat_grad), flat_grad.numel(), float(gscale), _p(norm), part.data_ptr(), _stream()),
               "grad_norm")
    return norm


@_classed("optim")
def adam_step(p: Tensor, g: Tensor, m: Tensor, v: Tensor, *, step: int, lr: float, betas=(0.9, 0.999), eps: float = 1e-8,
              weight_decay: float = 0.0, gscale: float = 1.0, norm: Optional[Tensor] = None, max_norm: float = 0.0,
              store_clipped_grad: bool = False) -> None:
    """In-place torch.optim.Adam update of the flat buffers (p, m, v), gradient pre-scaled by
    gscale * min(1, max_norm / (norm + 1e-6))."""
    lib = _lib.load()
    n = p.numel()
    if not (g.numel() == n and m.numel() == n and v.numel() == n):
        raise RuntimeError("adam_step: p, g, m, v must have the same number of elements")
    _lib.check(lib.diffsal_adam_step(_p(p), _p(g), _p(m), _p(v), n, float(lr), float(betas[0]), float(betas[1]), float(eps),
                                     float(weight_decay), int(step), float(gscale), _p(norm) if norm is not None else None,
                                     float(max_norm), int(store_clipped_grad), _stream()), "adam_step")


@_classed("optim")
def scale_by(x: Tensor, s: Tensor) -> Tensor:
    """x * s[0] with s a one-element device tensor."""
    lib = _lib.load()
    out = torch.empty_like(x)
    _lib.check(lib.diffsal_scale_by(_p(x), _p(s.reshape(1)), _p(out), x.numel(), _stream()), "scale_by")
    return out


@_classed("optim")
def multi_copy(srcs: Sequence[Tensor], dst_offsets: Sequence[int], dst: Tensor) -> None:
    """dst[off_i : off_i + srcs[i].numel()] = srcs[i] for all i, in a handful of launches (csrc/optim.hip)."""
    lib = _lib.load()
    n = len(srcs)
    if n == 0:
        return
    keep = [t if t.is_contiguous() else t.contiguous() for t in srcs]
    ptrs = (C.c_void_p * n)(*[_p(t) for t in keep])
    offs = (C.c_long * n)(*[int(o) for o in dst_offsets])
    sizes = (C.c_long * n)(*[t.numel() for t in keep])
    _lib.check(lib.diffsal_multi_copy(ptrs, offs, sizes, n, _p(dst), _stream()), "multi_copy")


# ------------------------------------------------------------------------------------------------
# once-per-clip encoders: general attention + MViTv2 pieces (csrc/attention.hip, csrc/mvit.hip)
# ------------------------------------------------------------------------------------------------
def _bhl_strides(t: Tensor):
    """(batch, head, row) element strides of a [B,H,L,D] view whose last dimension is dense."""
    if t.dim() != 4 or t.stride(3) != 1 or t.dtype != torch.float32 or not t.is_cuda:
        raise ValueError("attention operands must be fp32 GPU views [B,H,L,D] with a dense last dimension")
    return (C.c_long * 3)(t.stride(0), t.stride(1), t.stride(2))


def _k_slots(k_extra, k_slots):
    """The slot form of a one-hot k_extra (include/diffsal.h): given explicitly, or carried by the table (``relpos_onehot``)."""
    if k_slots is None and k_extra is not None:
        k_slots = getattr(k_extra, "slots", None)
    if k_slots is not None and (k_slots.dtype != torch.int32 or k_slots.shape != (k_extra.shape[0], 4) or not k_slots.is_contiguous()):
        raise ValueError("k_slots must be a contiguous int32 [Lk, 4] tensor")
    return k_slots


def attention_general(q: Tensor, k: Tensor, v: Tensor, *, scale: float, q_extra: Optional[Tensor] = None,
                      k_extra: Optional[Tensor] = None, residual: Optional[Tensor] = None, skip_first: bool = False,
                      want_lse: bool = False, k_slots: Optional[Tensor] = None):
    """softmax(scale q k^T + q_extra k_extra^T) v (+ residual) for [B,H,L,D] views -> [B, Lq, H*DV] (include/diffsal.h);
    with want_lse also the row log-sum-exp [B,H,Lq] the backward needs."""
    lib = _lib.load()
    B, H, Lq, D = q.shape
    Lk, DV = k.shape[2], v.shape[3]
    E = 0 if q_extra is None else q_extra.shape[-1]
    out = torch.empty((B, Lq, H * DV), device=q.device, dtype=torch.float32)
    lse = torch.empty((B, H, Lq), device=q.device, dtype=torch.float32) if want_lse else None
    nt = lib.diffsal_attention_general_tail_floats(B, H, Lq, Lk, DV)      # pieces of the last, partly filled round of workgroups
    tail = torch.empty((nt,), device=q.device, dtype=torch.float32) if nt else None
    flops = 2.0 * B * H * Lq * Lk * (D + E + DV)
    k_slots = _k_slots(k_extra, k_slots)
    with _prof("attn", flops, _nb(q, k, v, out)):
        _lib.check(lib.diffsal_attention_general(
            q.data_ptr(), _p(q_extra), k.data_ptr(), _p(k_extra), (None if k_slots is None else k_slots.data_ptr()), v.data_ptr(), None if residual is None else residual.data_ptr(),
            _p(out), _p(lse), B, H, Lq, Lk, D, E, DV, _bhl_strides(q), _bhl_strides(k), _bhl_strides(v),
            None if residual is None else _bhl_strides(residual), float(scale), int(skip_first), _p(tail), nt, _stream()),
            "attention_general")
    return (out, lse) if want_lse else out


ATTN_BWD_DS = True      # dS form of the backward (include/diffsal.h): the dq kernel reads dS instead of recomputing S and dP


def attention_general_bwd(q, k, v, out, lse, dout, *, scale: float, q_extra=None, k_extra=None, residual=None,
                          skip_first: bool = False, ds_form: Optional[bool] = None):
    """-> (dq [B,H,Lq,D], dq_extra [B,H,Lq,E] or None, dk [B,H,Lk,D], dv [B,H,Lk,DV]); dq includes the residual path."""
    lib = _lib.load()
    B, H, Lq, D = q.shape
    Lk, DV = k.shape[2], v.shape[3]
    E = 0 if q_extra is None else q_extra.shape[-1]
    dev = q.device
    dq = torch.empty((B, H, Lq, D), device=dev)
    dqe = torch.empty((B, H, Lq, E), device=dev) if E else None
    dk, dv = torch.empty((B, H, Lk, D), device=dev), torch.empty((B, H, Lk, DV), device=dev)
    delta = torch.empty((B, H, Lq), device=dev)
    splits = lib.diffsal_attention_general_bwd_splits(B, H, Lq, Lk)
    kv_part = torch.empty((splits, B * H * Lk * (D + DV)), device=dev) if splits > 1 else None
    nq = lib.diffsal_attention_general_bwd_qtail_floats(B, H, Lq, Lk, D, E)      # dq kernel: pieces of its last partial round
    q_tail = torch.empty((nq,), device=dev) if nq else None
    nds = lib.diffsal_attention_general_bwd_ds_floats(B, H, Lq, Lk) if (ATTN_BWD_DS if ds_form is None else ds_form) else 0
    ds = torch.empty((nds,), device=dev) if nds else None
    flops = 2.0 * B * H * Lq * Lk * (2 * (D + E) + 2 * DV + D + E + DV)
    with _prof("attn-bwd", flops, _nb(q, k, v, out, dout, dq, dk, dv)):
        _lib.check(lib.diffsal_attention_general_bwd(
            q.data_ptr(), _p(q_extra), k.data_ptr(), _p(k_extra), v.data_ptr(), None if residual is None else residual.data_ptr(),
            _p(out), _p(lse), _p(dout.contiguous()), _p(delta), _p(kv_part), _p(q_tail), nq, _p(ds), nds, _p(dq), _p(dqe), _p(dk), _p(dv), B, H, Lq, Lk,
            D, E, DV,
            _bhl_strides(q), _bhl_strides(k), _bhl_strides(v), None if residual is None else _bhl_strides(residual),
            float(scale), int(skip_first), _stream()), "attention_general_bwd")
    return dq, dqe, dk, dv


def im2col3d(x: Tensor, kernel, stride, pad, Kp: int) -> Tensor:
    """x [B,C,T,H,W] -> cols [B*To*Ho*Wo, Kp] (k = (c, kt, ky, kx), zero-padded)."""
    lib = _lib.load()
    B, Cc, T, H, W = x.shape
    To, Ho, Wo = ((s + 2 * p - k) // st + 1 for s, p, k, st in zip((T, H, W), pad, kernel, stride))
    cols = torch.empty((B * To * Ho * Wo, Kp), device=x.device, dtype=torch.float32)
    with _prof("patch", 0.0, _nb(x, cols)):
        _lib.check(lib.diffsal_im2col3d(_p(x), _p(cols), B, Cc, T, H, W, *kernel, *stride, *pad, Kp, _stream()), "im2col3d")
    return cols, (To, Ho, Wo)


def pool3d_ln(x: Tensor, w27: Tensor, gamma: Tensor, beta: Tensor, size, stride, eps: float = 1e-5) -> Tensor:
    """x: [B, N, heads, D] view of the fused qkv output (N = 1 + T*H*W) -> [B, heads, 1 + To*Ho*Wo, D]."""
    lib = _lib.load()
    B, N, heads, D = x.shape
    T, H, W = size
    assert N == 1 + T * H * W and x.stride(3) == 1 and x.stride(2) == D
    To, Ho, Wo = ((s - 1) // st + 1 for s, st in zip(size, stride))
    out = torch.empty((B, heads, 1 + To * Ho * Wo, D), device=x.device, dtype=torch.float32)
    with _prof("pool", 54.0 * out.numel(), _nb(out) * 2):
        _lib.check(lib.diffsal_pool3d_ln(x.data_ptr(), _p(w27), _p(gamma), _p(beta), _p(out), B, heads, D, T, H, W, *stride,
                                         x.stride(0), x.stride(1), eps, _stream()), "pool3d_ln")
    return out, (To, Ho, Wo)


def _cont(t: Tensor) -> Tensor:
    return t if t.is_contiguous() else t.contiguous()


def _ptr3(ts):
    return (C.c_void_p * 3)(*[_p(t) for t in ts])


def _filter_layout(ws) -> int:
    """0: tap-major [27, 96] filters (the packed eval weights), 1: the parameter's own [96, 27] (= Conv3d weight.reshape(96, 27):
    the training path hands the parameters over without a transposed copy and gets their gradients back in that layout)."""
    shapes = {tuple(w.shape) for w in ws}
    if shapes == {(27, 96)}:
        return 0
    if shapes == {(96, 27)}:
        return 1
    raise ValueError(f"pooling filters must all be [27, 96] or all [96, 27], got {sorted(shapes)}")


def qkv_pool(qkv: Tensor, w27, size, stride_q, stride_kv, norms=None):
    """The three attention_pool convolutions of a block in one launch (head dimension 96).  qkv: [B, N, 3, heads, 96]
    contiguous, fp32 or 16-bit storage (outputs are fp32 either way); w27 = (wq, wk, wv) each [27, 96]; norms = ((gamma, beta, eps) x 3) adds the LayerNorms, None = convolutions
    only.  -> (q, k, v) each [B, heads, 1 + Lo, 96], q_size, k_size."""
    lib = _lib.load()
    B, N, three, heads, D = qkv.shape
    T, H, W = size
    assert three == 3 and N == 1 + T * H * W and qkv.is_contiguous()
    q_size = tuple((s - 1) // st + 1 for s, st in zip(size, stride_q))
    k_size = tuple((s - 1) // st + 1 for s, st in zip(size, stride_kv))
    Lq, Lk = q_size[0] * q_size[1] * q_size[2], k_size[0] * k_size[1] * k_size[2]
    outs = [torch.empty((B, heads, 1 + L, D), device=qkv.device, dtype=torch.float32) for L in (Lq, Lk, Lk)]
    w27 = [_cont(w) for w in w27]
    cm = _filter_layout(w27)
    if norms is not None:
        gam, bet = [_cont(n[0]) for n in norms], [_cont(n[1]) for n in norms]
        eps = (C.c_float * 3)(*[float(n[2]) for n in norms])
        g3, b3 = _ptr3(gam), _ptr3(bet)
    else:
        g3 = b3 = eps = None
    sq, skv = (C.c_int * 3)(*stride_q), (C.c_int * 3)(*stride_kv)
    nbytes = sum(o.numel() for o in outs) * 8
    with _prof("pool", 54.0 * sum(o.numel() for o in outs), nbytes):
        _lib.check(lib.diffsal_qkv_pool(qkv.data_ptr(), _ptr3(w27), g3, b3, eps, _ptr3(outs), B, heads, D, T, H, W, sq, skv,
                                        _dt(qkv), cm, _stream()), "qkv_pool")
    return outs[0], outs[1], outs[2], q_size, k_size


def qkv_pool_bwd_data(dys, w27, qkv_shape, size, stride_q, stride_kv) -> Tensor:
    """dqkv [B, N, 3, heads, 96] of ``qkv_pool`` (convolution form) from the three output gradients."""
    lib = _lib.load()
    B, N, _, heads, D = qkv_shape
    T, H, W = size
    dys = [_cont(d) for d in dys]
    w27 = [_cont(w) for w in w27]
    cm = _filter_layout(w27)
    dqkv = torch.empty(tuple(qkv_shape), device=dys[0].device, dtype=torch.float32)
    sq, skv = (C.c_int * 3)(*stride_q), (C.c_int * 3)(*stride_kv)
    with _prof("pool-bwd", 54.0 * sum(d.numel() for d in dys), sum(d.numel() for d in dys) * 4 + dqkv.numel() * 4):
        _lib.check(lib.diffsal_qkv_pool_bwd_data(_ptr3(dys), _ptr3(w27), _p(dqkv), B, heads, D, T, H, W, sq, skv, cm, _stream()),
                   "qkv_pool_bwd_data")
    return dqkv


def qkv_pool_bwd_weight(qkv: Tensor, dys, size, stride_q, stride_kv, channel_major: bool = False) -> Tensor:
    """The three filter gradients of ``qkv_pool`` in one launch + one reduction: [3, 27, 96], or [3, 96, 27] (the parameter's
    layout) with ``channel_major``."""
    lib = _lib.load()
    B, N, _, heads, D = qkv.shape
    T, H, W = size
    dys = [_cont(d) for d in dys]
    sq, skv = (C.c_int * 3)(*stride_q), (C.c_int * 3)(*stride_kv)
    chunks = lib.diffsal_qkv_pool_bwd_weight_chunks(B, heads, T, H, W, sq)
    part = torch.empty((3, chunks, 27 * D), device=qkv.device, dtype=torch.float64)
    with _prof("pool-bwd", 54.0 * sum(d.numel() for d in dys), sum(d.numel() for d in dys) * 8):
        _lib.check(lib.diffsal_qkv_pool_bwd_weight(_p(qkv), _ptr3(dys), part.data_ptr(), B, heads, D, T, H, W, sq, skv,
                                                   int(channel_major), _stream()), "qkv_pool_bwd_weight")
    out = reduce_partials(part, 3, chunks, 27 * D)
    return out.view(3, D, 27) if channel_major else out.view(3, 27, D)


def _iptr3(ts):
    return (C.c_void_p * 3)(*[t.data_ptr() for t in ts])


@_classed("relpos")
def rel_tables(rels, plans):
    """(Rt, Rh, Rw) gathered relative-position tables [q, k, D] of one block from the learnt tables and their host-built sparse
    row maps (``MViT._rel_plan``): one launch."""
    lib = _lib.load()
    D = rels[0].shape[1]
    rels = [_cont(r) for r in rels]
    outs = [torch.empty((pl["q"], pl["k"], D), device=rels[0].device, dtype=torch.float32) for pl in plans]
    M = (C.c_int * 3)(*[pl["q"] * pl["k"] for pl in plans])
    _lib.check(lib.diffsal_rel_tables(_ptr3(rels), _iptr3([pl["idx2"] for pl in plans]), _ptr3([pl["w2"] for pl in plans]),
                                      _ptr3(outs), M, D, _stream()), "rel_tables")
    return outs


@_classed("relpos-bwd")
def rel_tables_bwd(douts, plans):
    """Gradients of the three learnt tables [len, D] from the gradients of the gathered ones: one launch."""
    lib = _lib.load()
    D = douts[0].shape[-1]
    douts = [_cont(d) for d in douts]
    outs = [torch.empty((pl["len"], D), device=douts[0].device, dtype=torch.float32) for pl in plans]
    R = (C.c_int * 3)(*[pl["len"] for pl in plans])
    _lib.check(lib.diffsal_rel_tables_bwd(_ptr3(douts), _iptr3([pl["csr_ptr"] for pl in plans]),
                                          _iptr3([pl["csr_col"] for pl in plans]), _ptr3([pl["csr_w"] for pl in plans]),
                                          _ptr3(outs), R, D, _stream()), "rel_tables_bwd")
    return outs


def maxpool_tokens(x: Tensor, size, kernel, stride) -> Tensor:
    lib = _lib.load()
    B, N, Cc = x.shape
    T, H, W = size
    To, Ho, Wo = ((s + 2 * (k // 2) - k) // st + 1 for s, k, st in zip(size, kernel, stride))
    out = torch.empty((B, 1 + To * Ho * Wo, Cc), device=x.device, dtype=torch.float32)
    with _prof("pool", 0.0, _nb(x, out)):
        _lib.check(lib.diffsal_maxpool_tokens(_p(x), _p(out), B, Cc, T, H, W, *kernel, *stride, _stream()), "maxpool_tokens")
    return out


def relpos_columns(k_size) -> int:
    """Column layout of the relative-position bias for a key grid: 32 (t [0,8), h [8,16), w [16,32)) when it fits, else 48
    (t [0,8), h [8,24), w [24,48)); include/diffsal.h."""
    kt, kh, kw = k_size
    return 32 if (kt <= 8 and kh <= 8 and kw <= 16) else 48


def relpos_onehot(k_size, E: int, device) -> Tensor:
    """[1 + kt*kh*kw, E] one-hot key rows matching ``relpos_project``'s columns; class-token row 0 stays zero."""
    kt, kh, kw = k_size
    w0 = 16 if E == 32 else 24
    l = torch.arange(kt * kh * kw, device=device)
    oh = torch.zeros((1 + kt * kh * kw, E), device=device)
    oh[1 + l, l // (kh * kw)] = 1.0
    oh[1 + l, 8 + (l // kw) % kh] = 1.0
    oh[1 + l, w0 + l % kw] = 1.0
    # the same table in slot form (include/diffsal.h, k_slots): the three columns of a key, E = none (the class token)
    slots = torch.full((1 + kt * kh * kw, 4), E, device=device, dtype=torch.int32)
    slots[1:, 0] = (l // (kh * kw)).int()
    slots[1:, 1] = (8 + (l // kw) % kh).int()
    slots[1:, 2] = (w0 + l % kw).int()
    oh.slots = slots
    return oh


def relpos_project(q: Tensor, Rt: Tensor, Rh: Tensor, Rw: Tensor, q_size, k_size, E: int = 48) -> Tensor:
    """q [B,heads,1+Lq,D] -> the E (48 or 32, see ``relpos_columns``) bias columns per query [B,heads,1+Lq,E]."""
    lib = _lib.load()
    B, heads, N, D = q.shape
    extra = torch.empty((B, heads, N, E), device=q.device, dtype=torch.float32)
    with _prof("relpos", 2.0 * B * heads * N * D * sum(k_size), _nb(q, extra)):
        _lib.check(lib.diffsal_relpos_project(_p(q), _p(Rt), _p(Rh), _p(Rw), _p(extra), B * heads, D, *q_size, *k_size, E,
                                              _stream()), "relpos_project")
    return extra


def tokens_to_channels_first(x: Tensor, off: int = 0) -> Tensor:
    """x [B, off+L, C] -> [B, C, L]."""
    lib = _lib.load()
    B, N, Cc = x.shape
    out = torch.empty((B, Cc, N - off), device=x.device, dtype=torch.float32)
    with _prof("K6", 0.0, _nb(x, out)):
        _lib.check(lib.diffsal_tokens_to_channels_first(_p(x), _p(out), B, Cc, N - off, off, _stream()), "tokens_to_channels_first")
    return out


def saliency_metrics_bwd(pred: Tensor, gt: Tensor, ws: Tensor, weights4: Tensor) -> Tensor:
    """d(sum_k weights4[k] * term_k) / d pred for the four batch-mean terms (cc, sim, nss, kl) from the forward's workspace."""
    lib = _lib.load()
    B = pred.shape[0]
    n = pred.numel() // B
    dp = torch.empty_like(pred)
    with _prof("metrics", 0.0, 3 * _nb(pred)):
        _lib.check(lib.diffsal_saliency_metrics_bwd(_p(pred), _p(gt), B, n, ws.data_ptr(), ws.numel() * 8, _p(weights4), _p(dp),
                                                    _stream()), "saliency_metrics_bwd")
    return dp


def saliency_metrics(pred: Tensor, gt: Tensor, keep_ws: bool = False):
    """-> (means [4], per_image [B,4]) in the order (cc, sim, nss, kl); R/models/sal_losses.py:14-176.  keep_ws: also return
    the contiguous fp32 inputs and the workspace (what saliency_metrics_bwd needs)."""
    lib = _lib.load()
    if pred.shape != gt.shape:
        raise RuntimeError(f"saliency_metrics: shapes differ: {tuple(pred.shape)} vs {tuple(gt.shape)}")
    B = pred.shape[0]
    n = pred.numel() // B
    p, g = pred.contiguous().float(), gt.contiguous().float()
    nws = lib.diffsal_saliency_metrics_ws_bytes(B)
    ws = torch.empty((nws // 8,), device=pred.device, dtype=torch.float64)
    per = torch.empty((B, 4), device=pred.device, dtype=torch.float32)
    mean = torch.empty((4,), device=pred.device, dtype=torch.float32)
    with _prof("metrics", 0.0, 2 * _nb(p, g)):
        _lib.check(lib.diffsal_saliency_metrics(_p(p), _p(g), B, n, ws.data_ptr(), nws, _p(per), _p(mean), _stream()),
                   "saliency_metrics")
    if keep_ws:
        return mean, per, (p, g, ws)
    return mean, per


def maxpool2d(x: Tensor, k: int = 2, stride: int = 2) -> Tensor:
    """MaxPool2d(k, stride) on NHWC (no padding)."""
    lib = _lib.load()
    N, H, W, Cc = x.shape
    out = torch.empty((N, (H - k) // stride + 1, (W - k) // stride + 1, Cc), device=x.device, dtype=x.dtype)
    dt = _dt(x)
    with _prof("pool", 0.0, _nb(x, out)):
        _lib.check(lib.diffsal_maxpool2d(_pa(x, dt), out.data_ptr(), N, H, W, Cc, k, stride, dt, _stream()), "maxpool2d")
    return out


def resize_update(s_low: Tensor, x: Tensor, m_prev: Optional[Tensor], ex: float, e0: float, A: float, c0: float, c1: float,
                  want_x0: bool = False, want_next: bool = True):
    """Fused end of a denoising step (include/diffsal.h): s_low [N,h,w,1] sigmoid map, x [N,1,H,W] sampler state ->
    (m, x_next, x0) with x0 = resize(s_low), m = ex x + e0 x0, x_next = A x + c0 m + c1 m_prev."""
    lib = _lib.load()
    N, h, w = s_low.shape[0], s_low.shape[1], s_low.shape[2]
    H, W = x.shape[-2], x.shape[-1]
    m = torch.empty_like(x)
    xn = torch.empty_like(x) if want_next else None
    x0 = torch.empty_like(x) if want_x0 else None
    with _prof("K15", 0.0, _nb(s_low, x, m_prev, m, xn, x0)):
        _lib.check(lib.diffsal_resize_update(_p(s_low), _p(x), _p(m_prev), _p(x0), _p(m), _p(xn), N, h, w, H, W, float(ex), float(e0),
                                             float(A), float(c0), float(c1), _stream()), "resize_update")
    return m, xn, x0


# ---- training of the encoders: forward variants that keep what the backward needs, and the backward kernels ----
def pool3d(x: Tensor, w27: Tensor, size, stride) -> Tensor:
    """attention_pool's depthwise Conv3d alone (no LayerNorm): x [B,N,heads,D] view -> [B,heads,1+Lo,D]."""
    lib = _lib.load()
    B, N, heads, D = x.shape
    T, H, W = size
    assert N == 1 + T * H * W and x.stride(3) == 1 and x.stride(2) == D
    To, Ho, Wo = ((s - 1) // st + 1 for s, st in zip(size, stride))
    out = torch.empty((B, heads, 1 + To * Ho * Wo, D), device=x.device, dtype=torch.float32)
    with _prof("pool", 54.0 * out.numel(), _nb(out) * 2):
        _lib.check(lib.diffsal_pool3d_ln(x.data_ptr(), _p(w27), None, None, _p(out), B, heads, D, T, H, W, *stride, x.stride(0),
                                         x.stride(1), 0.0, _stream()), "pool3d")
    return out, (To, Ho, Wo)


def pool3d_bwd_weight(x: Tensor, dy: Tensor, size, stride) -> Tensor:
    """dw27 [27][D] of ``pool3d`` (per-workgroup double partial sums, reduced in a fixed order)."""
    lib = _lib.load()
    B, N, heads, D = x.shape
    T, H, W = size
    dy = dy.contiguous()
    chunks = lib.diffsal_pool3d_bwd_weight_chunks()
    part = torch.empty((chunks, 27 * D), device=x.device, dtype=torch.float64)
    with _prof("pool-bwd", 54.0 * dy.numel(), _nb(dy) * 2):
        _lib.check(lib.diffsal_pool3d_bwd_weight(x.data_ptr(), _p(dy), part.data_ptr(), B, heads, D, T, H, W, *stride, x.stride(0),
                                                 x.stride(1), _stream()), "pool3d_bwd_weight")
    return reduce_partials(part.view(1, chunks, 27 * D), 1, chunks, 27 * D).view(27, D)


def pool3d_bwd(x: Tensor, w27: Tensor, dy: Tensor, dx_view: Tensor, size, stride):
    """dx (written into ``dx_view``, a view with x's strides) and dw27 [27][D] of ``pool3d``."""
    lib = _lib.load()
    B, N, heads, D = x.shape
    T, H, W = size
    assert dx_view.stride() == x.stride() and dx_view.shape == x.shape
    dy = dy.contiguous()
    with _prof("pool-bwd", 54.0 * dy.numel(), _nb(dy) + dx_view.numel() * 4):
        _lib.check(lib.diffsal_pool3d_bwd_data(_p(dy), _p(w27), dx_view.data_ptr(), B, heads, D, T, H, W, *stride, x.stride(0),
                                               x.stride(1), _stream()), "pool3d_bwd_data")
    return pool3d_bwd_weight(x, dy, size, stride)


def maxpool_tokens_idx(x: Tensor, size, kernel, stride):
    lib = _lib.load()
    B, N, Cc = x.shape
    T, H, W = size
    To, Ho, Wo = ((s + 2 * (k // 2) - k) // st + 1 for s, k, st in zip(size, kernel, stride))
    out = torch.empty((B, 1 + To * Ho * Wo, Cc), device=x.device, dtype=torch.float32)
    idx = torch.empty((B, 1 + To * Ho * Wo, Cc), device=x.device, dtype=torch.int32)
    with _prof("pool", 0.0, _nb(x, out, idx)):
        _lib.check(lib.diffsal_maxpool_tokens_idx(_p(x), _p(out), idx.data_ptr(), B, Cc, T, H, W, *kernel, *stride, _stream()),
                   "maxpool_tokens_idx")
    return out, idx


def maxpool_tokens_bwd(dy: Tensor, idx: Tensor, size, kernel, stride) -> Tensor:
    lib = _lib.load()
    B, _, Cc = dy.shape
    T, H, W = size
    din = torch.empty((B, 1 + T * H * W, Cc), device=dy.device, dtype=torch.float32)
    with _prof("pool-bwd", 0.0, _nb(dy, din, idx)):
        _lib.check(lib.diffsal_maxpool_tokens_bwd(_p(dy.contiguous()), idx.data_ptr(), _p(din), B, Cc, T, H, W, *kernel, *stride,
                                                  _stream()), "maxpool_tokens_bwd")
    return din


def relpos_project_bwd(dextra: Tensor, q: Tensor, Rt: Tensor, Rh: Tensor, Rw: Tensor, q_size, k_size, dq_accum: Optional[Tensor] = None):
    """-> (dq [B,heads,N,D] (added into dq_accum when given), dRt, dRh, dRw)."""
    lib = _lib.load()
    B, heads, N, D = q.shape
    dq = dq_accum if dq_accum is not None else torch.empty_like(q)
    chunks = lib.diffsal_relpos_project_bwd_chunks()
    width = Rt.numel() + Rh.numel() + Rw.numel()
    part = torch.empty((chunks, width), device=q.device, dtype=torch.float64)
    with _prof("relpos-bwd", 4.0 * B * heads * N * D * sum(k_size), _nb(q, dextra, dq)):
        _lib.check(lib.diffsal_relpos_project_bwd(_p(dextra.contiguous()), _p(q), _p(Rt), _p(Rh), _p(Rw), _p(dq),
                                                  int(dq_accum is not None), part.data_ptr(), B * heads, D, *q_size, *k_size,
                                                  dextra.shape[-1], _stream()), "relpos_project_bwd")
        flat = reduce_partials(part.view(1, chunks, width), 1, chunks, width).view(-1)
    a, b = Rt.numel(), Rt.numel() + Rh.numel()
    return dq, flat[:a].view(Rt.shape), flat[a:b].view(Rh.shape), flat[b:].view(Rw.shape)


def mlp_block(x1: Tensor, norm2, fc1, fc2, norm_z=None, frames=None):
    """Fused norm2 -> fc1 -> GELU -> fc2 -> +x1 (-> norm_z) of the C = 96 TransformerBlock (include/diffsal.h).
    x1 [..., 96] fp32; norm2 / norm_z: (gamma, beta, eps); fc1 / fc2: (weight, bias); frames = (hw, T, t_keep) limits the
    rows of z that are written.  -> (x2, z or None)."""
    lib = _lib.load()
    Cc = x1.shape[-1]
    M = x1.numel() // Cc
    x2 = torch.empty_like(x1)
    z = torch.empty_like(x1) if norm_z is not None else None
    hw, T, tk = frames if frames is not None else (M, 1, 1)
    gz, bz, ez = norm_z if norm_z is not None else (None, None, 0.0)
    hid = fc1[0].shape[0]
    with _prof("K10", 4.0 * M * Cc * hid, _nb(x1, x2, z), f"mlp_block M={M} C={Cc}", kernel="mlp_block_kernel"):
        _lib.check(lib.diffsal_mlp_block(_p(x1), _p(norm2[0]), _p(norm2[1]), float(norm2[2]), _p(fc1[0]), _p(fc1[1]), _p(fc2[0]),
                                         _p(fc2[1]), _p(x2), _p(z), _p(gz), _p(bz), float(ez), M, Cc, hid, hw, T, tk, _stream()),
                   "mlp_block")
    return x2, z


def block16(o: Tensor, x: Tensor, proj, norm2, fc1, fc2, norm_z=None, frames=None):
    """16-bit storage: x1 = x + proj(o); x2 = x1 + fc2(gelu(fc1(norm2(x1)))); z = norm_z(x2) on the kept frames -- one
    launch (include/diffsal.h).  proj / fc1 / fc2: (weight in the storage type, fp32 bias); norms: (gamma, beta, eps)."""
    lib = _lib.load()
    Cc = x.shape[-1]
    M = x.numel() // Cc
    dt = _dt(x)
    x2 = torch.empty_like(x)
    z = torch.empty_like(x) if norm_z is not None else None
    hw, T, tk = frames if frames is not None else (M, 1, 1)
    gz, bz, ez = norm_z if norm_z is not None else (None, None, 0.0)
    hid = fc1[0].shape[0]
    with _prof("K10", 2.0 * M * Cc * (Cc + 2 * hid), _nb(o, x, x2, z), f"block16 M={M} C={Cc}", kernel="block16_kernel"):
        _lib.check(lib.diffsal_block16(_pa(o, dt), _pa(x, dt), _pa(proj[0], dt), _p(proj[1]), _p(norm2[0]), _p(norm2[1]), float(norm2[2]),
                                       _pa(fc1[0], dt), _p(fc1[1]), _pa(fc2[0], dt), _p(fc2[1]), x2.data_ptr(),
                                       None if z is None else z.data_ptr(), _p(gz), _p(bz), float(ez), M, Cc, hid, hw, T, tk, dt,
                                       _stream()), "block16")
    return x2, z


# ---- pieces of the legacy DDPM-style UNet (diffusion_unet.py; R/models/diffusion_decoder/diffusion.py) ----------------
def groupnorm(x: Tensor, gamma: Tensor, beta: Tensor, groups: int = 32, eps: float = 1e-6, swish: bool = False) -> Tensor:
    """GroupNorm on NHWC [B,H,W,C] with the swish optional (AttnBlock.norm has none)."""
    lib = _lib.load()
    B, H, W, Cc = x.shape
    out = torch.empty_like(x)
    nbytes = lib.diffsal_groupnorm_ws_bytes(B, groups)
    ws = torch.empty((nbytes // 8,), device=x.device, dtype=torch.float64)
    dt = _dt(x)
    with _prof("K3", 0.0, _nb(x, out)):
        _lib.check(lib.diffsal_groupnorm(_pa(x, dt), _p(gamma), _p(beta), _pa(out, dt), B, H * W, Cc, groups, eps, int(swish),
                                         ws.data_ptr(), nbytes, dt, _stream()), "groupnorm")
    return out


def softmax_rows(x: Tensor, scale: float = 1.0) -> Tensor:
    """softmax(scale * x) over the last axis of a contiguous fp32 tensor."""
    lib = _lib.load()
    cols = x.shape[-1]
    out = torch.empty_like(x)
    with _prof("softmax", 0.0, _nb(x, out)):
        _lib.check(lib.diffsal_softmax_rows(_p(x), _p(out), x.numel() // cols, cols, float(scale), _stream()), "softmax_rows")
    return out


def upsample_nearest2(x: Tensor) -> Tensor:
    lib = _lib.load()
    N, H, W, Cc = x.shape
    out = torch.empty((N, 2 * H, 2 * W, Cc), device=x.device, dtype=torch.float32)
    with _prof("K12-up", 0.0, _nb(x, out)):
        _lib.check(lib.diffsal_upsample_nearest2(_p(x), _p(out), N, H, W, Cc, _stream()), "upsample_nearest2")
    return out


def avgpool2(x: Tensor) -> Tensor:
    lib = _lib.load()
    N, H, W, Cc = x.shape
    out = torch.empty((N, H // 2, W // 2, Cc), device=x.device, dtype=torch.float32)
    with _prof("pool", 0.0, _nb(x, out)):
        _lib.check(lib.diffsal_avgpool2(_p(x), _p(out), N, H, W, Cc, _stream()), "avgpool2")
    return out


def sigmoid_gate(y: Tensor, x: Tensor) -> Tensor:
    """sigmoid(y) * x, same shape, fp32 contiguous."""
    lib = _lib.load()
    out = torch.empty_like(x)
    with _prof("K15", 0.0, _nb(y, x, out)):
        _lib.check(lib.diffsal_sigmoid_gate(_p(y), _p(x), _p(out), x.numel(), _stream()), "sigmoid_gate")
    return out
