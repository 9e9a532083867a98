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

# Test / measurement switch (read once at import): DIFFSAL_NO_TAPE_FUSION=1 keeps every operator its own tape node -- LayerNorm
# and the residual's gradient accumulation, GELU and fc2, rel-pos projection and attention -- i.e. the graph before the
# fused nodes below; same results up to summation order.
import os as _os

_NO_FUSION = int(_os.environ.get("DIFFSAL_NO_TAPE_FUSION", "0") or 0)     # bit mask: 1 LayerNorm fork, 2 GELU into fc2, 4 rel-pos attention
FUSE_LN_FORK, FUSE_GELU, FUSE_RELPOS = not (_NO_FUSION & 1), not (_NO_FUSION & 2), not (_NO_FUSION & 4)


def dgrad_weight(w: Tensor, stride=(1, 1)) -> Tensor:
    """Weight of the data-gradient computation of a conv with parameter w [Cout, Cin, KH, KW] (or Conv3d [.., KT,1,1]).

    Stride 1: packed weight of dX = conv(dY, W^T flipped).  Strided convolutions (taps disjoint or overlapping): the
    [KH*KW*Cin, Cout] matrix of the GEMM dXcols = dY W, whose columns col2im maps back onto the input (ConvFn.backward)."""
    kh, kw = (w.shape[2], 1) if w.dim() == 5 else tuple(w.shape[2:])
    if tuple(stride) != (1, 1) and (kh, kw) != (1, 1):
        return ops.pack_cols_weight(w)
    return ops.pack_dgrad_weight(w)


class ConvFn(torch.autograd.Function):
    """y = act(conv(x, w) + bias + rowvec[img]) (+ residual after the activation), channels-last.

    w_packed: [Cout, K] (differentiable: receives dW in the same layout); w_dgrad: packed weight of the
    data-gradient conv (constant), or None when x needs no gradient."""

    @staticmethod
    def forward(ctx, x, w_packed, bias, rowvec, residual, w_dgrad, meta):
        kh, kw, stride, pad, dil, out_hw, act = meta[:7]
        in_gelu = len(meta) > 7 and meta[7]      # x is the pre-activation of an erf-GELU in front of this layer (Mlp.fc2)
        if act not in (ACT_NONE, ACT_RELU):
            raise NotImplementedError("training path: only none / ReLU are fused into the conv epilogue")
        if act == ACT_RELU and residual is not None:
            raise NotImplementedError("training path: ReLU + residual in one epilogue is not differentiable from y")
        if in_gelu and ((kh, kw) != (1, 1) or stride != (1, 1)):
            raise NotImplementedError("training path: the GELU in front of a layer is folded into 1x1 stride-1 products only")
        xin = ops.gelu(x) if in_gelu else x
        # 3x3 stride-1 padding = dilation convolutions with the parameter at hand (w_raw): F(4x4, 3x3) where the planner takes it
        # (a quarter of the products; the weight transform runs on the device, the weights change every step)
        w_raw = meta[8] if len(meta) > 8 else None
        ctx.wino = bool(WINO4_TRAIN and w_raw is not None and (kh, kw) == (3, 3) and stride == (1, 1) and pad == dil and dil[0] == dil[1]
                        and dil[0] in (1, 2) and not in_gelu and ops.wino4_supported(xin, w_raw.shape[0], dil[0]))
        if ctx.wino:
            y = ops.conv3x3_wino4_ex(xin, ops.wino4_weight(w_raw), bias=bias, rowvec=rowvec, residual=residual, act=act, dil=dil[0],
                                     tag="gemm")[0]
        else:
            y = ops.conv_igemm(xin, w_packed, kh=kh, kw=kw, stride=stride, pad=pad, dil=dil, out_hw=out_hw, bias=bias,
                               rowvec=rowvec, residual=residual, act=act)
        ctx.meta = meta
        ctx.has = (bias is not None, rowvec is not None, residual is not None)
        ctx.save_for_backward(xin, w_dgrad if w_dgrad is not None else x.new_empty(0), y if act == ACT_RELU else x.new_empty(0),
                              x if in_gelu else x.new_empty(0))
        return y

    @staticmethod
    def backward(ctx, dy):
        kh, kw, stride, pad, dil, out_hw, act = ctx.meta[:7]
        in_gelu = len(ctx.meta) > 7 and ctx.meta[7]
        x, w_dgrad, y, pre = ctx.saved_tensors
        has_b, has_rv, has_res = ctx.has
        dy = dy.contiguous()
        g = ops.relu_bwd(dy, y) if act == ACT_RELU else dy
        N, Ho, Wo, Cout = g.shape
        want_b = has_b and ctx.needs_input_grad[2]
        dw = db = None
        if ctx.needs_input_grad[1] and want_b:      # the bias gradient rides along in the weight-gradient kernel
            dw, db = ops.conv_wgrad(x, g, kh=kh, kw=kw, stride=stride, pad=pad, dil=dil, want_bias=True)
        elif ctx.needs_input_grad[1]:
            dw = ops.conv_wgrad(x, g, kh=kh, kw=kw, stride=stride, pad=pad, dil=dil)
        elif want_b:
            db = ops.colsum(g).reshape(Cout)
        drv = ops.colsum(g, Ho * Wo) if (has_rv and ctx.needs_input_grad[3]) else None
        dres = dy if (has_res and ctx.needs_input_grad[4]) else None
        dx = None
        w_raw = ctx.meta[8] if len(ctx.meta) > 8 else None
        if ctx.needs_input_grad[0] and ctx.wino and ops.wino4_supported(g, x.shape[-1], dil[0]):
            # dX = conv(dY, flipped w^T), same padding = dilation: the F(4x4) path with the flipped kernel's transform
            dx = ops.conv3x3_wino4_ex(g, ops.wino4_weight(w_raw, dgrad=True), dil=dil[0], tag="gemm")[0]
        elif ctx.needs_input_grad[0]:
            if w_dgrad.numel() == 0:
                if w_raw is None:
                    raise RuntimeError("ConvFn: input needs a gradient but no dgrad weight was supplied")
                w_dgrad = dgrad_weight(w_raw, stride)
            H, W = x.shape[1:3]
            if stride == (1, 1):
                # flipped-kernel convolution over dY with padding dil*(k-1) - pad
                pt, pl = dil[0] * (kh - 1) - pad[0], dil[1] * (kw - 1) - pad[1]
                if in_gelu:       # d(pre) = (dY W) * gelu'(pre): the multiplication rides in the product's epilogue
                    dx = ops.conv_igemm(g, w_dgrad, kh=kh, kw=kw, pad=(pt, pl), dil=dil, out_hw=(H, W), residual=pre,
                                        act=ops.ACT_GELU_GRAD)
                else:
                    dx = ops.conv_igemm(g, w_dgrad, kh=kh, kw=kw, pad=(pt, pl), dil=dil, out_hw=(H, W))
            elif stride[0] >= kh and stride[1] >= kw and dil == (1, 1):
                # taps never overlap: one GEMM dXcols = dY W, then every input pixel copies its single source
                if tuple(w_dgrad.shape) != (kh * kw * x.shape[-1], Cout):
                    raise RuntimeError("ConvFn: disjoint-tap backward needs dgrad_weight(w, stride) (columns layout)")
                cols = ops.conv_igemm(g.reshape(1, 1, N * Ho * Wo, Cout), w_dgrad)
                dx = ops.col2im_disjoint(cols.reshape(N * Ho * Wo, -1), x.shape, (Ho, Wo), kh, kw, stride, pad)
            elif dil == (1, 1):
                # strided conv with overlapping taps (3x3 stride-2 Downsample): the same GEMM, then every input pixel sums
                # the column entries that map onto it -- exactly the forward's MACs, no zero-inserted dY
                if tuple(w_dgrad.shape) != (kh * kw * x.shape[-1], Cout):
                    raise RuntimeError("ConvFn: strided backward needs dgrad_weight(w, stride) (columns layout)")
                cols = ops.conv_igemm(g.reshape(1, 1, N * Ho * Wo, Cout), w_dgrad)
                dx = ops.col2im_gather(cols.reshape(N * Ho * Wo, -1), x.shape, (Ho, Wo), kh, kw, stride, pad)
            else:
                raise NotImplementedError("ConvFn: data gradient of a strided AND dilated convolution")
        return dx, dw, db, drv, dres, None, None


# F(4x4, 3x3) for the training-mode 3x3 convolutions (forward and data gradient) that pass their parameter (DIFFSAL_NO_WINO4_TRAIN=1: off)
WINO4_TRAIN = _os.environ.get("DIFFSAL_NO_WINO4_TRAIN", "0") != "1"


def conv(x, w_packed, *, kh=1, kw=1, stride=(1, 1), pad=(0, 0), dil=(1, 1), out_hw=None, bias=None, rowvec=None,
         residual=None, act=ACT_NONE, w_dgrad=None, in_gelu=False, w_raw=None):
    """w_raw: the parameter itself [Cout, Cin, KH, KW] (constant here; its gradient flows through w_packed): lets the node pick the
    F(4x4) path for 3x3 stride-1 convolutions and build the data-gradient weight it needs in the backward (w_dgrad may then be None)."""
    return ConvFn.apply(x, w_packed, bias, rowvec, residual, w_dgrad,
                        (kh, kw, tuple(stride), tuple(pad), tuple(dil), out_hw, act, bool(in_gelu), None if w_raw is None else w_raw.detach()))


def linear(x, w, bias=None, *, residual=None, act=ACT_NONE, in_gelu=False):
    """Token GEMM with autograd; w: [N, K] (its own packed form); dX uses w^T.  in_gelu: y = gelu(x) w^T + ... with x the
    pre-activation (Mlp: fc2(gelu(fc1 x))); its backward multiplies by gelu'(x) in the data-gradient product's epilogue
    instead of a separate pass over the hidden tensor."""
    if in_gelu and not FUSE_GELU:
        x, in_gelu = gelu(x), False
    lead = x.shape[:-1]
    M = 1
    for s in lead:
        M *= s
    wd = ops.pack_dgrad_weight(w) if (x.requires_grad and w.shape[0] % 32 == 0) else (
        w.detach().t().contiguous() if x.requires_grad else None)
    y = conv(x.reshape(1, 1, M, x.shape[-1]), w, bias=bias, act=act, w_dgrad=wd, in_gelu=in_gelu,
             residual=None if residual is None else residual.reshape(1, 1, M, w.shape[0]))
    return y.reshape(*lead, w.shape[0])


class LayerNormFn(torch.autograd.Function):
    """LayerNorm over the last dim; backward recomputes the row statistics in registers."""

    @staticmethod
    def forward(ctx, x, gamma, beta, eps):
        ctx.eps = eps
        ctx.save_for_backward(x, gamma)
        return ops.layernorm(x, gamma, beta, eps)

    @staticmethod
    def backward(ctx, dy):
        x, gamma = ctx.saved_tensors
        dx, dg, db = ops.layernorm_bwd(x, dy.contiguous(), gamma, ctx.eps)
        return dx, dg, db, None


def layernorm(x, gamma, beta, eps=1e-5):
    return LayerNormFn.apply(x, gamma, beta, eps)


class LayerNorm3Fn(torch.autograd.Function):
    """Three LayerNorms of one width as one node: one forward launch, one backward launch + one reduction (MViT's norm_q / norm_k /
    norm_v on the pooled tensors: the key / value tensors are a few hundred tokens per head)."""

    @staticmethod
    def forward(ctx, x0, x1, x2, g0, b0, g1, b1, g2, b2, eps):
        ctx.eps = eps
        ctx.save_for_backward(x0, x1, x2, g0, g1, g2)
        return tuple(ops.layernorm_multi((x0, x1, x2), (g0, g1, g2), (b0, b1, b2), eps))

    @staticmethod
    def backward(ctx, d0, d1, d2):
        x0, x1, x2, g0, g1, g2 = ctx.saved_tensors
        xs, ds = (x0, x1, x2), [d0, d1, d2]
        ds = [torch.zeros_like(x) if d is None else d for x, d in zip(xs, ds)]
        dxs, dgs, dbs = ops.layernorm_bwd_multi(xs, ds, (g0, g1, g2), ctx.eps)
        return dxs[0], dxs[1], dxs[2], dgs[0], dbs[0], dgs[1], dbs[1], dgs[2], dbs[2], None


def layernorm3(xs, norms):
    """LayerNorm of three tensors by three nn.LayerNorm modules of one width -> three outputs."""
    (n0, n1, n2) = norms
    return LayerNorm3Fn.apply(xs[0], xs[1], xs[2], n0.weight, n0.bias, n1.weight, n1.bias, n2.weight, n2.bias,
                              (float(n0.eps), float(n1.eps), float(n2.eps)))


class LayerNormForkFn(torch.autograd.Function):
    """(x, LayerNorm(x)) for a pre-norm residual block: x goes on over the residual connection, the normalised copy into the
    branch.  One node for both uses of x, so its backward is ONE kernel, dx = LN'(d_branch) + d_residual, where the tape
    would run LayerNorm's backward and then an accumulation pass over the token tensor."""

    @staticmethod
    def forward(ctx, x, gamma, beta, eps):
        ctx.eps = eps
        ctx.save_for_backward(x, gamma)
        return x.view_as(x), ops.layernorm(x, gamma, beta, eps)

    @staticmethod
    def backward(ctx, dres, dy):
        x, gamma = ctx.saved_tensors
        if dy is None:
            return dres, None, None, None
        add = None if dres is None else dres.contiguous()
        dx, dg, db = ops.layernorm_bwd(x, dy.contiguous(), gamma, ctx.eps, add=add)
        return dx, dg, db, None


def layernorm_fork(x, gamma, beta, eps=1e-5):
    """-> (x for the residual path, LayerNorm(x)); use the returned x, not the argument, downstream."""
    if not FUSE_LN_FORK:
        return x, LayerNormFn.apply(x, gamma, beta, eps)
    return LayerNormForkFn.apply(x, gamma, beta, eps)


class GroupNormSwishFn(torch.autograd.Function):
    """swish(GroupNorm(x)) on NHWC [B,H,W,C] (sal_unet.py:36-44), statistics kept for the backward."""

    @staticmethod
    def forward(ctx, x, gamma, beta, groups, eps):
        B, H, W, C = x.shape
        hw, cpg = H * W, C // groups
        st = ops.rowstats(x, hw, 0)                                  # [B, 2, C] float64
        mu_c, rs_c, scale, shift, _, _ = ops.norm_finalize_fwd(st, gamma, beta, groups, float(hw * cpg), eps)
        ctx.groups = groups
        ctx.save_for_backward(x, gamma, beta, mu_c, rs_c)
        return ops.affine_act(x, scale, shift, hw, 4)

    @staticmethod
    def backward(ctx, dy):
        x, gamma, beta, mu_c, rs_c = ctx.saved_tensors
        B, H, W, C = x.shape
        hw, G = H * W, ctx.groups
        dy = dy.contiguous()
        t = ops.rowstats(x, hw, 2, dy=dy, mu=mu_c, rs=rs_c, gamma=gamma, beta=beta, stat_per_seg=True)  # [B,2,C]
        dgamma, dbeta, k1, k2, k3 = ops.norm_finalize_bwd(t, gamma, rs_c, G, float(hw * (C // G)))
        dx = ops.norm_bwd_apply(x, dy, None, mu_c, rs_c, gamma, beta, k1, k2, k3, hw, 2)
        return dx, dgamma, dbeta, None, None


def groupnorm_swish(x, gamma, beta, groups=32, eps=1e-6):
    return GroupNormSwishFn.apply(x, gamma, beta, groups, eps)


class BatchNormReLUFn(torch.autograd.Function):
    """relu(BatchNorm2d(x)) in TRAIN mode on channels-last rows [M, C]: batch statistics over all rows (per rank,
    unsynchronised -- exactly what the reference's DDP does).  ``running`` = (running_mean, running_var,
    momentum) is updated in place by the statistics kernel, like nn.BatchNorm2d in train mode."""

    @staticmethod
    def forward(ctx, x, gamma, beta, eps, relu, running):
        C = x.shape[-1]
        M = x.numel() // C
        st = ops.rowstats(x, M, 0)                     # [1, 2, C] float64
        bn = None if running is None else (running[0], running[1], running[2], M / max(M - 1, 1))
        mu, rs, scale, shift, _, _ = ops.norm_finalize_fwd(st, gamma, beta, C, float(M), eps, bn=bn)
        y = ops.affine_act(x, scale, shift, M, ACT_RELU if relu else ACT_NONE)
        ctx.relu = relu
        ctx.save_for_backward(x, y if relu else x.new_empty(0), gamma, mu, rs)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, y, gamma, mu, rs = ctx.saved_tensors
        C = x.shape[-1]
        M = x.numel() // C
        dy = dy.contiguous()
        mode = 1 if ctx.relu else 3
        t = ops.rowstats(x, M, mode, dy=dy, y=y if ctx.relu else None, mu=mu, rs=rs)   # [1, 2, C]
        dgamma, dbeta, k1, k2, k3 = ops.norm_finalize_bwd(t, gamma, rs, C, float(M))
        dx = ops.norm_bwd_apply(x, dy, y if ctx.relu else None, mu, rs, None, None, k1, k2, k3, M, mode)
        return dx, dgamma, dbeta, None, None, None


def batchnorm_relu_train(x, bn: "torch.nn.BatchNorm2d", relu=True):
    """Functional train-mode BatchNorm(+ReLU) that also updates the module's running statistics like nn.BatchNorm2d."""
    mom = bn.momentum if bn.momentum is not None else 0.1
    y = BatchNormReLUFn.apply(x, bn.weight, bn.bias, bn.eps, relu, (bn.running_mean, bn.running_var, mom))
    bn.num_batches_tracked += 1
    return y


class DropoutFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, p, seed):
        ctx.p, ctx.seed = p, seed
        return ops.dropout(x, p, seed)

    @staticmethod
    def backward(ctx, dy):
        return ops.dropout(dy.contiguous(), ctx.p, ctx.seed), None, None


def dropout(x, p, seed):
    return DropoutFn.apply(x, p, seed) if p > 0 else x


class GeluFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        ctx.save_for_backward(x)
        return ops.gelu(x)

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        return ops.act_bwd(dy.contiguous(), x, 2)


def gelu(x):
    return GeluFn.apply(x)


class AddFn(torch.autograd.Function):
    """a + b through the axpy kernel (skip connections of the training graph)."""

    @staticmethod
    def forward(ctx, a, b):
        return ops.axpbypcz(a, 1.0, b, 1.0)

    @staticmethod
    def backward(ctx, dy):
        return dy, dy


def add(a, b):
    return AddFn.apply(a, b)


class DwConvFn(torch.autograd.Function):
    """Depthwise k x k convolution on NHWC; w: [k*k, C]."""

    @staticmethod
    def forward(ctx, x, w, k, stride, pad):
        ctx.cfg = (k, stride, pad)
        ctx.save_for_backward(x, w)
        return ops.dwconv(x, w, k, stride, pad)

    @staticmethod
    def backward(ctx, du):
        x, w = ctx.saved_tensors
        k, stride, pad = ctx.cfg
        dx, dw = ops.dwconv_bwd(x, w, du.contiguous(), k, stride, pad, ctx.needs_input_grad[0], ctx.needs_input_grad[1])
        return dx, dw, None, None, None


def dwconv(x, w, k, stride, pad):
    return DwConvFn.apply(x, w, k, stride, pad)


class AttentionFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, q, k, v, heads, scale):
        ctx.cfg = (heads, scale)
        ctx.save_for_backward(q, k, v)
        return ops.attention(q, k, v, heads, scale)

    @staticmethod
    def backward(ctx, do):
        q, k, v = ctx.saved_tensors
        heads, scale = ctx.cfg
        dq, dk, dv = ops.attention_bwd(q, k, v, do.contiguous(), heads, scale)
        return dq, dk, dv, None, None


def attention(q, k, v, heads, scale):
    return AttentionFn.apply(q, k, v, heads, scale)


class ResizeFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, H, W):
        ctx.hw = x.shape[1:3]
        return ops.resize_bilinear(x, H, W)

    @staticmethod
    def backward(ctx, dy):
        return ops.resize_bilinear_bwd(dy.contiguous(), ctx.hw[0], ctx.hw[1]), None, None


def resize_bilinear(x, H, W):
    return ResizeFn.apply(x, H, W)


class ResizeSumFn(torch.autograd.Function):
    """sum_i bilinear(x_i -> (H, W)); the adjoint is one gather-form resize backward per input."""

    @staticmethod
    def forward(ctx, H, W, *xs):
        ctx.shapes = [x.shape[1:3] for x in xs]
        return ops.resize_sum(list(xs), H, W)

    @staticmethod
    def backward(ctx, dy):
        dy = dy.contiguous()
        return (None, None) + tuple(ops.resize_bilinear_bwd(dy, h, w) for h, w in ctx.shapes)


def resize_sum(xs, H, W):
    return ResizeSumFn.apply(H, W, *xs)


class TapSumFn(torch.autograd.Function):
    """conv3x3(dil)(sum_i bilinear(z_i)) from the tap products y_all = cat_i(z_i) x Wcat^T (rows: source after source, each
    [N*h_i*w_i, 9*Cout]) -- ops.tapsum; the adjoint writes every source's block of d(y_all) with ops.tapsum_bwd."""

    @staticmethod
    def forward(ctx, y_all, bias, meta):
        shapes, N, H, W, Cout, dil = meta
        views, off = [], 0
        for h, w in shapes:
            views.append(y_all[off:off + N * h * w].view(N, h, w, 9 * Cout))
            off += N * h * w
        if off != y_all.shape[0]:
            raise RuntimeError("TapSumFn: the rows of y_all do not add up to the sources' pixels")
        ctx.meta = meta
        ctx.rows = y_all.shape[0]
        return ops.tapsum(views, H, W, Cout, dil=dil, bias=bias, tag="tapsum")

    @staticmethod
    def backward(ctx, du):
        shapes, N, H, W, Cout, dil = ctx.meta
        du = du.contiguous()
        d_all = torch.empty((ctx.rows, 9 * Cout), device=du.device, dtype=torch.float32)
        off = 0
        for h, w in shapes:
            ops.tapsum_bwd(du, h, w, dil, out=d_all[off:off + N * h * w].view(N, h, w, 9 * Cout))
            off += N * h * w
        db = ops.colsum(du.view(-1, Cout)).reshape(Cout) if ctx.needs_input_grad[1] else None
        return d_all, db, None


def tapsum(y_all, shapes, N, H, W, Cout, dil=1, bias=None):
    return TapSumFn.apply(y_all, bias, (tuple(shapes), N, H, W, Cout, dil))


def tap_weight(w):
    """Conv2d weight [Cout, Cin, 3, 3] -> [9*Cout, Cin] (row = tap * Cout + co), differentiable (torch permutation)."""
    co, ci = w.shape[:2]
    return w.permute(2, 3, 0, 1).reshape(9 * co, ci)


class PackFramesFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, vis, noise):
        ctx.tv = vis.shape[2]
        ctx.has_noise = noise is not None
        return ops.pack_frames(vis, noise)

    @staticmethod
    def backward(ctx, dout):
        dout = dout.contiguous()
        dvis = ops.unpack_frames(dout, ctx.tv) if ctx.needs_input_grad[0] else None
        dnoise = dout[:, ctx.tv].contiguous() if (ctx.has_noise and ctx.needs_input_grad[1]) else None
        return dvis, dnoise


def pack_frames(vis, noise):
    return PackFramesFn.apply(vis, noise)


class HeadSigmoidFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, y, w, b):
        s = ops.head_sigmoid(y, w, b)
        ctx.save_for_backward(y, w, s)
        return s

    @staticmethod
    def backward(ctx, ds):
        y, w, s = ctx.saved_tensors
        dy, dw, db = ops.head_bwd(y, w, s, ds.contiguous())
        return dy, dw, db


def head_sigmoid(y, w, b):
    return HeadSigmoidFn.apply(y, w, b)


class ConvInFn(torch.autograd.Function):
    """conv_in (1 -> C); the noisy map x carries no gradient."""

    @staticmethod
    def forward(ctx, x, w9, bias, skip_mod):
        ctx.save_for_backward(x)
        return ops.conv_in(x, w9, bias, skip_mod)

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        dw, db = ops.conv_in_bwd(x, dy.contiguous())
        return None, dw, db, None


def conv_in(x, w9, bias, skip_mod=0):
    return ConvInFn.apply(x, w9, bias, skip_mod)


class DenseSmallFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, b, swish_in):
        ctx.swish_in = swish_in
        ctx.save_for_backward(x, w)
        return ops.dense_small(x, w, b, swish_in)

    @staticmethod
    def backward(ctx, dout):
        x, w = ctx.saved_tensors
        dx, dw, db = ops.dense_small_bwd(x, w, dout.contiguous(), ctx.swish_in)
        return dx, dw, db, None


def dense_small(x, w, b, swish_in):
    return DenseSmallFn.apply(x, w, b, swish_in)


class AudioFuseFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a_small, x, h, w):
        ctx.hw = (h, w)
        ctx.save_for_backward(a_small, x)
        return ops.audio_fuse(a_small, x, h, w)

    @staticmethod
    def backward(ctx, dout):
        a_small, x = ctx.saved_tensors
        dx, da = ops.audio_fuse_bwd(a_small, x, dout.contiguous(), ctx.hw[0], ctx.hw[1])
        return da, dx, None, None


def audio_fuse(a_small, x, h, w):
    return AudioFuseFn.apply(a_small, x, h, w)


class MseLossFn(torch.autograd.Function):
    """loss = loss_scale * sum (pred - target)^2 (R/models/sal_losses.py:189-192); one kernel produces the loss
    and d(loss)/d(pred), the backward only applies the incoming scalar."""

    @staticmethod
    def forward(ctx, pred, target, loss_scale):
        loss, dpred = ops.mse_loss(pred.contiguous(), target.contiguous(), loss_scale, want_grad=True)
        ctx.save_for_backward(dpred)
        return loss.reshape(())

    @staticmethod
    def backward(ctx, dloss):
        (dpred,) = ctx.saved_tensors
        return ops.scale_by(dpred, dloss.contiguous()), None, None


def mse_loss(pred, target, loss_scale):
    return MseLossFn.apply(pred, target, loss_scale)
