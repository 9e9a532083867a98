"""torch.autograd.Function wrappers: forward AND backward are hand-written gfx950 kernels (ops.py / C ABI).

Only the training step uses these; inference calls ops.* directly.  PyTorch's autograd supplies the
tape and the (pure layout) un-packing of weight gradients; every arithmetic kernel is ours.
"""
from __future__ import annotations

from typing import Optional

import torch

from . import ops
from .ops import ACT_NONE, ACT_RELU

Tensor = torch.Tensor


def dgrad_weight(w: Tensor) -> Tensor:
    """Packed weight of the data-gradient convolution: dX = conv(dY, W^T flipped).  w: [Cout, Cin, KH, KW]."""
    return ops.pack_conv_weight(w.detach().permute(1, 0, 2, 3).flip(2, 3).contiguous())


class ConvFn(torch.autograd.Function):
    """y = act(conv(x, w) + bias + rowvec[img]) (+ residual after the activation), channels-last.

    w_packed: [Cout, K] (differentiable: receives dW in the same layout); w_dgrad: packed weight of the
    data-gradient conv (constant), or None when x needs no gradient."""

    @staticmethod
    def forward(ctx, x, w_packed, bias, rowvec, residual, w_dgrad, meta):
        kh, kw, stride, pad, dil, out_hw, act = meta
        if act not in (ACT_NONE, ACT_RELU):
            raise NotImplementedError("training path: only none / ReLU are fused into the conv epilogue")
        if act == ACT_RELU and residual is not None:
            raise NotImplementedError("training path: ReLU + residual in one epilogue is not differentiable from y")
        y = ops.conv_igemm(x, w_packed, kh=kh, kw=kw, stride=stride, pad=pad, dil=dil, out_hw=out_hw, bias=bias,
                           rowvec=rowvec, residual=residual, act=act)
        ctx.meta = meta
        ctx.has = (bias is not None, rowvec is not None, residual is not None)
        ctx.save_for_backward(x, w_dgrad if w_dgrad is not None else x.new_empty(0), y if act == ACT_RELU else x.new_empty(0))
        return y

    @staticmethod
    def backward(ctx, dy):
        kh, kw, stride, pad, dil, out_hw, act = ctx.meta
        x, w_dgrad, y = ctx.saved_tensors
        has_b, has_rv, has_res = ctx.has
        dy = dy.contiguous()
        g = ops.relu_bwd(dy, y) if act == ACT_RELU else dy
        N, Ho, Wo, Cout = g.shape
        dw = ops.conv_wgrad(x, g, kh=kh, kw=kw, stride=stride, pad=pad, dil=dil) if ctx.needs_input_grad[1] else None
        db = ops.colsum(g).reshape(Cout) if (has_b and ctx.needs_input_grad[2]) else None
        drv = ops.colsum(g, Ho * Wo) if (has_rv and ctx.needs_input_grad[3]) else None
        dres = dy if (has_res and ctx.needs_input_grad[4]) else None
        dx = None
        if ctx.needs_input_grad[0]:
            if w_dgrad.numel() == 0:
                raise RuntimeError("ConvFn: input needs a gradient but no dgrad weight was supplied")
            H, W = x.shape[1:3]
            if stride == (1, 1):
                # flipped-kernel convolution over dY with padding dil*(k-1) - pad
                pt, pl = dil[0] * (kh - 1) - pad[0], dil[1] * (kw - 1) - pad[1]
                dx = ops.conv_igemm(g, w_dgrad, kh=kh, kw=kw, pad=(pt, pl), dil=dil, out_hw=(H, W))
            else:
                # strided conv: scatter dY onto a zero grid at the stride (layout plumbing), then a stride-1 conv
                Hd, Wd = (Ho - 1) * stride[0] + 1, (Wo - 1) * stride[1] + 1
                gd = g.new_zeros((N, Hd, Wd, Cout))
                gd[:, ::stride[0], ::stride[1]] = g
                pt, pl = dil[0] * (kh - 1) - pad[0], dil[1] * (kw - 1) - pad[1]
                dx = ops.conv_igemm(gd, w_dgrad, kh=kh, kw=kw, pad=(pt, pl), dil=dil, out_hw=(H, W))
        return dx, dw, db, drv, dres, None, None


def conv(x, w_packed, *, kh=1, kw=1, stride=(1, 1), pad=(0, 0), dil=(1, 1), out_hw=None, bias=None, rowvec=None,
         residual=None, act=ACT_NONE, w_dgrad=None):
    return ConvFn.apply(x, w_packed, bias, rowvec, residual, w_dgrad,
                        (kh, kw, tuple(stride), tuple(pad), tuple(dil), out_hw, act))


def linear(x, w, bias=None, *, residual=None, act=ACT_NONE):
    """Token GEMM with autograd; w: [N, K] (its own packed form); dX uses w^T."""
    lead = x.shape[:-1]
    M = 1
    for s in lead:
        M *= s
    wd = w.detach().t().contiguous() if x.requires_grad else None
    y = conv(x.reshape(1, 1, M, x.shape[-1]), w, bias=bias, act=act, w_dgrad=wd,
             residual=None if residual is None else residual.reshape(1, 1, M, w.shape[0]))
    return y.reshape(*lead, w.shape[0])
