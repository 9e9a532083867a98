"""The reference's legacy DDPM-style denoiser (``models/diffusion_decoder/diffusion.py``: ``DiffusionModel`` :197-357) on the
HIP path: same constructor argument (the ``config`` namespace), same parameter / buffer names and shapes, same
``forward(x, t, vis_feat)``.  Inference only (no configuration of the reference instantiates this class, let alone trains
it); fp32, channels-last inside.

Every arithmetic operator is one of this package's kernels:
  * timestep embedding MLP, ``temb_proj``                    -> ``dense_small`` (K1)
  * GroupNorm(32, eps 1e-6) + swish / without swish          -> ``groupnorm_swish`` (K3) / ``groupnorm`` (AttnBlock.norm)
  * every Conv2d 3x3 / 1x1 incl. the stride-2 Downsample     -> ``conv_igemm`` (bias, per-image temb vector and residual fused)
  * AttnBlock's full HW x HW self-attention                  -> two ``conv_igemm`` products per image (k and v^T of the image
    as the weight operand) around ``softmax_rows``; q comes from the normalised tensor, k and v from the un-normalised
    one and the scale is C^-1/2, as the reference has it (:157-167)
  * nearest x2 Upsample / avg-pool Downsample / sigmoid gate -> ``upsample_nearest2`` / ``avgpool2`` / ``sigmoid_gate``
  * ``feat_trans``                                            -> ``audio_attention.Transformer`` (the class the reference imports)
``torch.cat`` of the skip connections and the zero-padding of conv_in's input channels to 32 are layout plumbing."""
import math
from typing import List, Optional, Sequence

import torch
import torch.nn as nn
from torch import Tensor

from . import ops
from .audio_attention import Transformer


def _res_params(cin: int, cout: int, temb_ch: int) -> nn.Module:
    m = nn.Module()
    m.norm1 = nn.GroupNorm(32, cin, eps=1e-6, affine=True)
    m.conv1 = nn.Conv2d(cin, cout, 3, 1, 1)
    m.temb_proj = nn.Linear(temb_ch, cout)
    m.norm2 = nn.GroupNorm(32, cout, eps=1e-6, affine=True)
    m.conv2 = nn.Conv2d(cout, cout, 3, 1, 1)
    if cin != cout:
        m.nin_shortcut = nn.Conv2d(cin, cout, 1, 1, 0)
    return m


def _attn_params(c: int) -> nn.Module:
    m = nn.Module()
    m.norm = nn.GroupNorm(32, c, eps=1e-6, affine=True)
    for n in ("q", "k", "v", "proj_out"):
        setattr(m, n, nn.Conv2d(c, c, 1, 1, 0))
    return m


def _resample_params(c: int, with_conv: bool, stride: int) -> nn.Module:
    m = nn.Module()
    m.with_conv = with_conv
    if with_conv:
        m.conv = nn.Conv2d(c, c, 3, stride, 1 if stride == 1 else 0)
    return m


class _FeatInteraction(nn.Module):
    """Parameter holder of the reference's FeatInteraction (diffusion.py:359-369)."""

    def __init__(self, dim: int):
        super().__init__()
        self.feat_trans = Transformer(dim=dim, depth=1, heads=4, dim_head=64, mlp_dim=256)


class DiffusionModel(nn.Module):
    def __init__(self, config, _multiscale: bool = False):
        super().__init__()
        self.config = config
        self.multiscale = _multiscale
        mc = config.model
        ch, out_ch, ch_mult = mc.ch, mc.out_ch, tuple(mc.ch_mult)
        if float(mc.dropout) != 0.0:
            # eval-only module: dropout is the identity at inference whatever its rate, so the rate is only recorded
            pass
        for c in {ch * m for m in ch_mult} | {ch}:
            if c % 32 != 0:
                raise ValueError(f"DiffusionModel: channel count {c} must be a multiple of 32 (GroupNorm(32) and the GEMM k-slices)")
        if mc.type == "bayesian":
            self.logvar = nn.Parameter(torch.zeros(config.diffusion.num_diffusion_timesteps))
        self.ch, self.temb_ch = ch, ch * 4
        self.num_resolutions, self.num_res_blocks = len(ch_mult), mc.num_res_blocks
        self.resolution, self.width, self.in_channels = config.data.image_size, config.data.width, mc.in_channels
        attn_res, with_conv = list(mc.attn_resolutions), bool(mc.resamp_with_conv)

        self.temb = nn.Module()
        self.temb.dense = nn.ModuleList([nn.Linear(ch, self.temb_ch), nn.Linear(self.temb_ch, self.temb_ch)])
        self.conv_in = nn.Conv2d(self.in_channels, ch, 3, 1, 1)
        curr, in_mult, block_in = self.resolution, (1,) + ch_mult, None
        self.down = nn.ModuleList()
        for i in range(self.num_resolutions):
            block, attn = nn.ModuleList(), nn.ModuleList()
            block_in, block_out = ch * in_mult[i], ch * ch_mult[i]
            for _ in range(self.num_res_blocks):
                block.append(_res_params(block_in, block_out, self.temb_ch))
                block_in = block_out
                if curr in attn_res and not _multiscale:
                    attn.append(_attn_params(block_in))
            d = nn.Module()
            d.block, d.attn = block, attn
            if i != self.num_resolutions - 1:
                d.downsample = _resample_params(block_in, with_conv, 2)
                curr //= 2
            self.down.append(d)
        self.mid = nn.Module()
        self.mid.block_1 = _res_params(block_in, block_in, self.temb_ch)
        self.mid.attn_1 = _attn_params(block_in)
        self.mid.block_2 = _res_params(block_in, block_in, self.temb_ch)
        self.up = nn.ModuleList()
        for i in reversed(range(self.num_resolutions)):
            block, attn = nn.ModuleList(), nn.ModuleList()
            block_out, skip_in = ch * ch_mult[i], ch * ch_mult[i]
            for j in range(self.num_res_blocks + 1):
                if j == self.num_res_blocks:
                    skip_in = ch * in_mult[i]
                block.append(_res_params(block_in + skip_in, block_out, self.temb_ch))
                block_in = block_out
                if curr in attn_res and not _multiscale:
                    attn.append(_attn_params(block_in))
                if curr in attn_res and _multiscale and j == 0:
                    attn.append(_FeatInteraction(block_in))
            u = nn.Module()
            u.block, u.attn = block, attn
            if i != 0:
                u.upsample = _resample_params(block_in, with_conv, 1)
                curr *= 2
            self.up.insert(0, u)
        self.norm_out = nn.GroupNorm(32, block_in, eps=1e-6, affine=True)
        self.conv_out = nn.Conv2d(block_in, out_ch, 3, 1, 1)
        if not _multiscale:
            self.feat_trans = Transformer(dim=512, depth=1, heads=4, dim_head=64, mlp_dim=256)
        self._packed = None

    # ---- packed GEMM weights (rebuilt when a parameter changes) -------------------------------------------------------------
    def parameters_updated(self) -> None:
        self._packed = None

    def _pack(self):
        key = sum(p._version for p in self.parameters())
        if self._packed is not None and self._packed[0] == key:
            return self._packed[1]
        pk = {}
        with torch.no_grad():
            for name, m in self.named_modules():
                if isinstance(m, nn.Conv2d):
                    w = m.weight
                    if w.shape[1] % 32 != 0:          # conv_in: pad the input channels to one 32-wide k slice
                        wp = w.new_zeros((w.shape[0], 32 * ((w.shape[1] + 31) // 32), *w.shape[2:]))
                        wp[:, :w.shape[1]] = w
                        w = wp
                    pk[name] = ops.pack_conv_weight(w)
        self._packed = (key, pk)
        return pk

    def _load_from_state_dict(self, *args, **kwargs):
        super()._load_from_state_dict(*args, **kwargs)
        self._packed = None

    # ---- blocks ---------------------------------------------------------------------------------------------------------
    def _res(self, name: str, m: nn.Module, x: Tensor, temb_sw: Tensor, pk) -> Tensor:
        """ResnetBlock.forward (diffusion.py:115-133) on NHWC; temb_sw = the embedding (its swish is applied inside dense_small)."""
        h = ops.groupnorm_swish(x, m.norm1.weight, m.norm1.bias, 32, m.norm1.eps)
        tp = ops.dense_small(temb_sw, m.temb_proj.weight, m.temb_proj.bias, True)
        h = ops.conv_igemm(h, pk[name + ".conv1"], kh=3, kw=3, pad=(1, 1), bias=m.conv1.bias, rowvec=tp, tag="K4")
        h = ops.groupnorm_swish(h, m.norm2.weight, m.norm2.bias, 32, m.norm2.eps)
        sc = x
        if hasattr(m, "nin_shortcut"):
            sc = ops.conv_igemm(x, pk[name + ".nin_shortcut"], bias=m.nin_shortcut.bias, tag="K4")
        return ops.conv_igemm(h, pk[name + ".conv2"], kh=3, kw=3, pad=(1, 1), bias=m.conv2.bias, residual=sc, tag="K4")

    def _attn(self, name: str, m: nn.Module, x: Tensor, pk) -> Tensor:
        """AttnBlock.forward (diffusion.py:171-184): single-head attention over all H*W positions of an image."""
        B, H, W, C = x.shape
        L = H * W
        hn = ops.groupnorm(x, m.norm.weight, m.norm.bias, 32, m.norm.eps, swish=False)
        q = ops.linear(hn.view(B, L, C), pk[name + ".q"], m.q.bias, tag="attn-gemm")
        k = ops.linear(x.view(B, L, C), pk[name + ".k"], m.k.bias, tag="attn-gemm")      # keys / values: the UN-normalised x
        v = ops.linear(x.view(B, L, C), pk[name + ".v"], m.v.bias, tag="attn-gemm")
        Lp = 32 * ((L + 31) // 32)                    # contraction length of P V as whole 32-wide k slices
        vt = ops.tokens_to_channels_first(v)          # [B, C, L]
        if Lp != L:
            vt = torch.nn.functional.pad(vt, (0, Lp - L))
        o = torch.empty((B, L, C), device=x.device, dtype=torch.float32)
        for b in range(B):
            s = ops.linear(q[b], k[b], None, tag="attn-gemm")                            # [L, L] = q k^T
            p = ops.softmax_rows(s, float(C) ** -0.5)
            if Lp != L:
                p = torch.nn.functional.pad(p, (0, Lp - L))
            o[b] = ops.linear(p, vt[b].contiguous(), None, tag="attn-gemm")              # [L, C] = P v
        return ops.linear(o, pk[name + ".proj_out"], m.proj_out.bias, residual=x.view(B, L, C), tag="attn-gemm").view(B, H, W, C)

    def _downsample(self, name: str, m: nn.Module, x: Tensor, pk) -> Tensor:
        if not m.with_conv:
            return ops.avgpool2(x)
        H, W = x.shape[1:3]
        return ops.conv_igemm(x, pk[name + ".conv"], kh=3, kw=3, stride=(2, 2), out_hw=((H - 2) // 2 + 1, (W - 2) // 2 + 1),
                              bias=m.conv.bias, tag="K5")          # pad (0,1,0,1): rows / columns past the edge read zeros

    def _upsample(self, name: str, m: nn.Module, x: Tensor, pk) -> Tensor:
        x = ops.upsample_nearest2(x)
        if m.with_conv:
            x = ops.conv_igemm(x, pk[name + ".conv"], kh=3, kw=3, pad=(1, 1), bias=m.conv.bias, tag="K4")
        return x

    def feat_interact(self, x: Tensor, y: Tensor, trans: Optional[nn.Module] = None) -> Tensor:
        """x NHWC [B,h,w,C], y NCHW [B,C,h,w] (diffusion.py:311-319, FeatInteraction :364-369) -> NHWC."""
        b, c, h, w = y.shape
        if tuple(x.shape) != (b, h, w, c):
            raise RuntimeError(f"feat_interact: feature {tuple(y.shape)} does not match the UNet tensor {tuple(x.shape)} (NHWC)")
        yt = (trans or self.feat_trans)(y.permute(0, 2, 3, 1).reshape(b, h * w, c).contiguous())
        return ops.sigmoid_gate(yt.reshape(b, h, w, c), x)

    # ---- forward --------------------------------------------------------------------------------------------------------
    def forward(self, x: Tensor, t: Tensor, vis_feat: Optional[Sequence[Tensor]] = None) -> Tensor:
        """x [B,in_channels,H,W], t [B], vis_feat[0] [B,512,H/2^(levels-1),W/2^(levels-1)] -> [B,out_ch,H,W]."""
        if not x.is_cuda:
            raise RuntimeError("diff_sal_amd.DiffusionModel runs on the GPU only (no CPU fallback); got a CPU tensor")
        if self.training:
            raise NotImplementedError("DiffusionModel: inference only (the reference never trains this class)")
        if vis_feat is None:
            raise TypeError("DiffusionModel.forward: vis_feat is required (the reference indexes vis_feat[0])")
        pk = self._pack()
        with torch.no_grad():
            half = self.ch // 2
            freq = torch.exp(torch.arange(half, dtype=torch.float32) * -(math.log(10000) / (half - 1))).to(x.device)
            arg = t.to(torch.float32)[:, None] * freq[None, :]
            emb = torch.cat([arg.sin(), arg.cos()], dim=1)
            if self.ch % 2 == 1:
                emb = torch.nn.functional.pad(emb, (0, 1, 0, 0))
            d0, d1 = self.temb.dense[0], self.temb.dense[1]
            temb = ops.dense_small(ops.dense_small(emb.contiguous(), d0.weight, d0.bias, False), d1.weight, d1.bias, True)

            B, Cin, H, W = x.shape
            if self.multiscale and (H != self.resolution or W != self.width):     # the reference asserts this (:494-495)
                raise AssertionError(f"DiffusionModel_w_MultiScale: input {H}x{W} != configured {self.resolution}x{self.width}")
            xin = x.new_zeros((B, H, W, 32 * ((Cin + 31) // 32)))
            xin[..., :Cin] = x.permute(0, 2, 3, 1)
            hs: List[Tensor] = [ops.conv_igemm(xin, pk["conv_in"], kh=3, kw=3, pad=(1, 1), bias=self.conv_in.bias, tag="K2")]
            for i, d in enumerate(self.down):
                for j, blk in enumerate(d.block):
                    h = self._res(f"down.{i}.block.{j}", blk, hs[-1], temb, pk)
                    if len(d.attn) > 0:
                        h = self._attn(f"down.{i}.attn.{j}", d.attn[j], h, pk)
                    hs.append(h)
                if i != self.num_resolutions - 1:
                    hs.append(self._downsample(f"down.{i}.downsample", d.downsample, hs[-1], pk))
            h = self._res("mid.block_1", self.mid.block_1, hs[-1], temb, pk)
            h = self._attn("mid.attn_1", self.mid.attn_1, h, pk)
            h = self._res("mid.block_2", self.mid.block_2, h, temb, pk)
            next_feat = 0          # the reference pops vis_feat (mutating the caller's list, its own TODO); an index does the same
            if not self.multiscale:
                h = self.feat_interact(h, vis_feat[0])
            for i in reversed(range(self.num_resolutions)):
                u = self.up[i]
                for j, blk in enumerate(u.block):
                    h = self._res(f"up.{i}.block.{j}", blk, torch.cat([h, hs.pop()], dim=-1), temb, pk)
                    if self.multiscale:
                        if j == 0 and len(u.attn) > 0:
                            h = self.feat_interact(h, vis_feat[next_feat], u.attn[0].feat_trans)
                            next_feat += 1
                    elif len(u.attn) > 0:
                        h = self._attn(f"up.{i}.attn.{j}", u.attn[j], h, pk)
                if i != 0:
                    h = self._upsample(f"up.{i}.upsample", u.upsample, h, pk)
            h = ops.groupnorm_swish(h, self.norm_out.weight, self.norm_out.bias, 32, self.norm_out.eps)
            out = ops.conv_igemm(h, pk["conv_out"], kh=3, kw=3, pad=(1, 1), bias=self.conv_out.bias, tag="K14")
            return out.permute(0, 3, 1, 2).contiguous()


class DiffusionModel_w_MultiScale(DiffusionModel):
    """R/models/diffusion_decoder/diffusion.py:380-548: no attention in the encoder; in the decoder the first block of every
    level whose resolution is in ``attn_resolutions`` is followed by a FeatInteraction with the next entry of ``vis_feat``
    (coarsest first).  The caller's list is not mutated."""

    def __init__(self, config):
        super().__init__(config, _multiscale=True)
