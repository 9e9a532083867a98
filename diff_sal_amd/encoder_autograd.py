"""torch.autograd.Function wrappers of the encoder kernels (csrc/attention.hip, csrc/mvit.hip): HIP forward AND HIP
backward, so that VideoSaliencyModel(MViT + AudioAttnNet + SalUNet) trains end to end on the native path
(R/diffusion_trainer.py:212-235 runs MViT and the audio transformer inside the training step; VGGish is frozen there,
R/models/vggish.py:33-49, R/models/diff_model.py:73-74).  PyTorch supplies the tape only."""
from __future__ import annotations

import torch

from . import ops

Tensor = torch.Tensor


class AttentionGeneralFn(torch.autograd.Function):
    """softmax(scale q k^T + q_extra k_extra^T) v (+ residual q) -> [B, Lq, H*DV]; q, k, v: contiguous [B,H,L,D]."""

    @staticmethod
    def forward(ctx, q, k, v, q_extra, k_extra, scale, residual_q, skip_first):
        res = q if residual_q else None
        out, lse = ops.attention_general(q, k, v, scale=scale, q_extra=q_extra, k_extra=k_extra, residual=res,
                                         skip_first=skip_first, want_lse=True)
        ctx.scale, ctx.residual_q, ctx.skip_first = scale, residual_q, skip_first
        ctx.has_extra = q_extra is not None
        ctx.save_for_backward(q, k, v, q_extra if q_extra is not None else q.new_empty(0),
                              k_extra if k_extra is not None else q.new_empty(0), out, lse)
        return out

    @staticmethod
    def backward(ctx, dout):
        q, k, v, qe, ke, out, lse = ctx.saved_tensors
        qe, ke = (qe, ke) if ctx.has_extra else (None, None)
        dq, dqe, dk, dv = ops.attention_general_bwd(q, k, v, out, lse, dout.contiguous(), scale=ctx.scale, q_extra=qe, k_extra=ke,
                                                    residual=q if ctx.residual_q else None, skip_first=ctx.skip_first)
        return dq, dk, dv, dqe, None, None, None, None


def attention_general(q, k, v, *, scale, q_extra=None, k_extra=None, residual_q=False, skip_first=False):
    return AttentionGeneralFn.apply(q, k, v, q_extra, k_extra, float(scale), bool(residual_q), bool(skip_first))


class QKVPoolFn(torch.autograd.Function):
    """The three attention_pool convolutions of one block on the fused qkv tensor [B,N,3,heads,D]: one Function so that the
    three input gradients land in ONE qkv-gradient buffer (each writes its own slice; no add, no concat)."""

    @staticmethod
    def forward(ctx, qkv, wq, wk, wv, size, stride_q, stride_kv):
        ctx.size, ctx.strides = size, (stride_q, stride_kv, stride_kv)
        ctx.save_for_backward(qkv, wq, wk, wv)
        if qkv.shape[-1] == 96 and qkv.is_contiguous():          # one launch for the three tensors (filters in either layout)
            return ops.qkv_pool(qkv, (wq, wk, wv), size, stride_q, stride_kv)[:3]
        cm = tuple(wq.shape) == (qkv.shape[-1], 27) and qkv.shape[-1] != 27
        return tuple(ops.pool3d(qkv[:, :, i], w.t().contiguous() if cm else w, size, st)[0]
                     for i, (w, st) in enumerate(zip((wq, wk, wv), ctx.strides)))

    @staticmethod
    def backward(ctx, dq, dk, dv):
        qkv, wq, wk, wv = ctx.saved_tensors
        sq, skv = ctx.strides[0], ctx.strides[1]
        fused = (qkv.shape[-1] == 96 and qkv.is_contiguous() and sq[0] == 1 and skv[0] == 1 and sq[1] == sq[2]
                 and skv[1] == skv[2])
        if fused:
            dqkv = ops.qkv_pool_bwd_data((dq, dk, dv), (wq, wk, wv), qkv.shape, ctx.size, sq, skv)
            dws = ops.qkv_pool_bwd_weight(qkv, (dq, dk, dv), ctx.size, sq, skv, channel_major=tuple(wq.shape) == (96, 27))
            return dqkv, dws[0], dws[1], dws[2], None, None, None
        dqkv = torch.empty_like(qkv)
        dws = []
        cm = tuple(wq.shape) == (qkv.shape[-1], 27) and qkv.shape[-1] != 27       # parameter layout: the per-tensor kernels are tap-major
        for i, (w, g, st) in enumerate(zip((wq, wk, wv), (dq, dk, dv), ctx.strides)):
            dw = ops.pool3d_bwd(qkv[:, :, i], w.t().contiguous() if cm else w, g, dqkv[:, :, i], ctx.size, st)
            dws.append(dw.t().contiguous() if cm else dw)
        return dqkv, dws[0], dws[1], dws[2], None, None, None


def qkv_pool(qkv, wq, wk, wv, size, stride_q, stride_kv):
    return QKVPoolFn.apply(qkv, wq, wk, wv, tuple(size), tuple(stride_q), tuple(stride_kv))


class RelposProjectFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, q, Rt, Rh, Rw, q_size, k_size, E):
        ctx.sizes = (q_size, k_size)
        ctx.save_for_backward(q, Rt, Rh, Rw)
        return ops.relpos_project(q, Rt, Rh, Rw, q_size, k_size, E)

    @staticmethod
    def backward(ctx, dextra):
        q, Rt, Rh, Rw = ctx.saved_tensors
        dq, dRt, dRh, dRw = ops.relpos_project_bwd(dextra, q, Rt, Rh, Rw, *ctx.sizes)
        return dq, dRt, dRh, dRw, None, None, None


def relpos_project(q, Rt, Rh, Rw, q_size, k_size, E=48):
    return RelposProjectFn.apply(q, Rt, Rh, Rw, tuple(q_size), tuple(k_size), int(E))


class RelposAttentionFn(torch.autograd.Function):
    """MViT's pooling attention with its decomposed relative-position bias as one node: relpos_project + attention_general
    forward; backward = attention backward, then the projection's query gradient ACCUMULATES into the attention's dq
    (`accumulate` of diffsal_relpos_project_bwd) -- q feeds both, and as two nodes the tape adds the two [B,H,Lq,96]
    gradients in a pass of its own."""

    @staticmethod
    def forward(ctx, q, k, v, Rt, Rh, Rw, onehot, scale, q_size, k_size, E):
        extra = ops.relpos_project(q, Rt, Rh, Rw, q_size, k_size, E)
        out, lse = ops.attention_general(q, k, v, scale=scale, q_extra=extra, k_extra=onehot, residual=q, skip_first=True,
                                         want_lse=True)
        ctx.scale, ctx.sizes = scale, (q_size, k_size)
        ctx.save_for_backward(q, k, v, extra, onehot, out, lse, Rt, Rh, Rw)
        return out

    @staticmethod
    def backward(ctx, dout):
        q, k, v, extra, onehot, out, lse, Rt, Rh, Rw = ctx.saved_tensors
        dq, dqe, dk, dv = ops.attention_general_bwd(q, k, v, out, lse, dout.contiguous(), scale=ctx.scale, q_extra=extra,
                                                    k_extra=onehot, residual=q, skip_first=True)
        dq, dRt, dRh, dRw = ops.relpos_project_bwd(dqe, q, Rt, Rh, Rw, *ctx.sizes, dq_accum=dq)
        return dq, dk, dv, dRt, dRh, dRw, None, None, None, None, None


def relpos_attention(q, k, v, Rt, Rh, Rw, onehot, *, scale, q_size, k_size, E):
    """softmax(scale q k^T + rel-pos bias) v + q (class-token row without bias / residual): mvit.py:363-410, 587-605."""
    from .autograd_ops import FUSE_RELPOS
    if not FUSE_RELPOS:
        extra = relpos_project(q, Rt, Rh, Rw, q_size, k_size, E)
        return attention_general(q, k, v, scale=scale, q_extra=extra, k_extra=onehot, residual_q=True, skip_first=True)
    return RelposAttentionFn.apply(q, k, v, Rt, Rh, Rw, onehot, float(scale), tuple(q_size), tuple(k_size), int(E))


class RelTablesFn(torch.autograd.Function):
    """resize_decomposed_rel_pos of the three axes of one block (linear resample + index gather) as one sparse row map per
    table: one launch forward, one backward (the torch form is ~14 small kernels per table and step, with a sort-based
    index_put backward)."""

    @staticmethod
    def forward(ctx, rel_t, rel_h, rel_w, plans):
        ctx.plans = plans
        return tuple(ops.rel_tables((rel_t, rel_h, rel_w), plans))

    @staticmethod
    def backward(ctx, dt, dh, dw):
        zeros = [None if g is not None else torch.zeros((pl["q"], pl["k"], 96), device=pl["idx2"].device)
                 for g, pl in zip((dt, dh, dw), ctx.plans)]
        gs = [g if g is not None else z for g, z in zip((dt, dh, dw), zeros)]
        d = ops.rel_tables_bwd(gs, ctx.plans)
        return d[0], d[1], d[2], None


def rel_tables(rel_t, rel_h, rel_w, plans):
    return RelTablesFn.apply(rel_t, rel_h, rel_w, plans)


class MaxPoolTokensFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, size, kernel, stride):
        out, idx = ops.maxpool_tokens_idx(x, size, kernel, stride)
        ctx.meta = (size, kernel, stride)
        ctx.save_for_backward(idx)
        return out

    @staticmethod
    def backward(ctx, dy):
        (idx,) = ctx.saved_tensors
        return ops.maxpool_tokens_bwd(dy, idx, *ctx.meta), None, None, None


def maxpool_tokens(x, size, kernel, stride):
    return MaxPoolTokensFn.apply(x, tuple(size), tuple(kernel), tuple(stride))


class TokensToChannelsFirstFn(torch.autograd.Function):
    """[B, off+L, C] -> [B, C, L]; backward = the inverse transpose into a zero-padded token tensor."""

    @staticmethod
    def forward(ctx, x, off):
        ctx.off, ctx.n = off, x.shape[1]
        return ops.tokens_to_channels_first(x, off)

    @staticmethod
    def backward(ctx, dy):
        B, C, L = dy.shape
        g = ops.pack_frames(dy.contiguous().view(B, C, 1, 1, L), None).view(B, L, C)      # NCTHW -> frames == [B, L, C]
        if ctx.off == 0:
            return g, None
        full = g.new_zeros((B, ctx.n, C))
        full[:, ctx.off:] = g
        return full, None


def tokens_to_channels_first(x, off=0):
    return TokensToChannelsFirstFn.apply(x, off)


class PackTokensFn(torch.autograd.Function):
    """NCTHW [B,C,T,h,w] -> tokens [B, T*h*w, C] (ops.pack_frames); backward = tokens_to_channels_first."""

    @staticmethod
    def forward(ctx, x):
        b, c, t, h, w = x.shape
        ctx.shape = (b, c, t, h, w)
        return ops.pack_frames(x.contiguous(), None).view(b, t * h * w, c)

    @staticmethod
    def backward(ctx, dy):
        return ops.tokens_to_channels_first(dy.contiguous(), 0).view(ctx.shape)


def pack_tokens(x):
    return PackTokensFn.apply(x)
