"""CPU restatement of the reference's MViTv2 video encoder -- TEST INFRASTRUCTURE ONLY.

Plain fp32 PyTorch, functional over a ``state_dict`` with the reference's parameter names; every function cites the
lines of R/models/mvit.py it follows.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module; the product path (diff_sal_amd/) never does.  Pinned against the real reference by the fixtures of
oracle/gen_golden.py::gen_mvit (tests/golden/mvit_*.npz, checked in tests/test_oracle_golden.py).
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, List, Sequence, Tuple

import torch
import torch.nn.functional as F

Tensor = torch.Tensor

ARCH_ZOO = {  # mvit.py:888-917
    "tiny": dict(embed_dims=96, num_layers=10, num_heads=1, downscale_indices=[1, 3, 8]),
    "small": dict(embed_dims=96, num_layers=16, num_heads=1, downscale_indices=[1, 3, 14]),
    "base": dict(embed_dims=96, num_layers=24, num_heads=1, downscale_indices=[2, 5, 21]),
}


@dataclass
class LayerCfg:
    in_dims: int
    out_dims: int
    heads: int
    stride_q: Tuple[int, int, int]
    stride_kv: Tuple[int, int, int]
    rel_sizes: Tuple[int, int]          # (rows of rel_pos_h / rel_pos_w, rows of rel_pos_t)
    out_stage: int = -1                 # index of the output scale produced after this layer, or -1


@dataclass
class MViTConfig:
    """MViT(arch, out_scales=[0,1,2,3]) with the defaults of mvit.py:919-944 (class token on, rel-pos on, residual pooling,
    dim_mul_in_attention, mlp_ratio 4).  The reference hard-codes embed 96, patch kernel (3,7,7) / stride (2,4,4) /
    padding (1,3,3) and a nominal (16, 224, 224) input for the rel-pos table sizes (mvit.py:983-990)."""
    arch: object = "small"
    layers: List[LayerCfg] = field(default_factory=list)
    embed_dims: int = 96

    def __post_init__(self):
        a = ARCH_ZOO[self.arch] if isinstance(self.arch, str) else dict(self.arch)
        nl, heads, down = a["num_layers"], a["num_heads"], list(a["downscale_indices"])
        dim_mul_idx = list(a.get("dim_mul_indices", down))
        stage_of = {idx - 1: i for i, idx in enumerate(down)}
        stage_of[nl - 1] = len(down)
        dims = self.embed_dims
        stride_kv = (1, 8, 8)
        size = (8, 56, 56)                       # patch_resolution of the nominal input (mvit.py:990)
        for i in range(nl):                      # mvit.py:1011-1053
            if i in down or i in dim_mul_idx:
                heads *= 2
            if i in down:
                stride_q = (1, 2, 2)
                stride_kv = tuple(max(s // 2, 1) for s in stride_kv)
            else:
                stride_q = (1, 1, 1)
            out_dims = dims * 2 if i in dim_mul_idx else dims
            rel_hw = 2 * max(size[1] // stride_q[1], size[1] // stride_kv[1]) - 1     # mvit.py:569-575
            self.layers.append(LayerCfg(dims, out_dims, heads, stride_q, stride_kv, (rel_hw, 2 * size[0] - 1),
                                        stage_of.get(i, -1)))
            size = tuple(s // q for s, q in zip(size, stride_q))                      # mvit.py:771-775
            dims = out_dims


def state_dict_template(cfg: MViTConfig) -> Dict[str, Tensor]:
    """Names / shapes of the reference MViT.state_dict() (asserted against the real module in gen_golden.py)."""
    sd: Dict[str, Tensor] = {}

    def E(name, *shape):
        sd[name] = torch.empty(*shape)

    E("cls_token", 1, 1, cfg.embed_dims)
    E("patch_embed.projection.weight", cfg.embed_dims, 3, 3, 7, 7)
    E("patch_embed.projection.bias", cfg.embed_dims)
    for i, L in enumerate(cfg.layers):
        p = f"blocks.{i}."
        hd = L.out_dims // L.heads
        for n, c in (("norm1", L.in_dims), ("norm2", L.out_dims)):
            E(p + n + ".weight", c); E(p + n + ".bias", c)
        E(p + "attn.rel_pos_h", L.rel_sizes[0], hd)
        E(p + "attn.rel_pos_w", L.rel_sizes[0], hd)
        E(p + "attn.rel_pos_t", L.rel_sizes[1], hd)
        E(p + "attn.qkv.weight", 3 * L.out_dims, L.in_dims); E(p + "attn.qkv.bias", 3 * L.out_dims)
        E(p + "attn.proj.weight", L.out_dims, L.out_dims); E(p + "attn.proj.bias", L.out_dims)
        for w in "qkv":
            E(p + f"attn.pool_{w}.weight", hd, 1, 3, 3, 3)
            E(p + f"attn.norm_{w}.weight", hd); E(p + f"attn.norm_{w}.bias", hd)
        E(p + "mlp.fc1.weight", 4 * L.out_dims, L.out_dims); E(p + "mlp.fc1.bias", 4 * L.out_dims)
        E(p + "mlp.fc2.weight", L.out_dims, 4 * L.out_dims); E(p + "mlp.fc2.bias", L.out_dims)
        if L.in_dims != L.out_dims:
            E(p + "proj.weight", L.out_dims, L.in_dims); E(p + "proj.bias", L.out_dims)
        if L.out_stage >= 0:
            E(f"norm{L.out_stage}.weight", L.out_dims); E(f"norm{L.out_stage}.bias", L.out_dims)
    return sd


def synth_state_dict(template: Dict[str, Tensor], tag: str = "mvit") -> Dict[str, Tensor]:
    """Closed-form deterministic fill (same generator as the SalUNet fixtures): fan-in scaled matrices, near-identity
    norms, O(0.3) relative-position tables so the bias really shapes the attention."""
    from oracle.salunet_oracle import synth_tensor

    out = {}
    for k, v in template.items():
        shp = tuple(v.shape)
        if k == "cls_token":
            out[k] = synth_tensor(tag + k, shp, 0.5)
        elif "rel_pos" in k:
            out[k] = synth_tensor(tag + k, shp, 0.3 / math.sqrt(shp[1]))
        elif "pool_" in k:
            out[k] = synth_tensor(tag + k, shp, 1.0 / math.sqrt(27.0)) + (1.0 / 27.0)
        elif len(shp) >= 2:
            fan_in = 1
            for s in shp[1:]:
                fan_in *= s
            out[k] = synth_tensor(tag + k, shp, 1.0 / math.sqrt(fan_in))
        elif k.endswith("weight"):
            out[k] = 1.0 + synth_tensor(tag + k, shp, 0.1)
        else:
            out[k] = synth_tensor(tag + k, shp, 0.1)
    return out


def resize_decomposed_rel_pos(rel_pos: Tensor, q_size: int, k_size: int) -> Tensor:
    """[q_size, k_size, C] table of relative-position embeddings (mvit.py:330-361)."""
    max_rel = int(2 * max(q_size, k_size) - 1)
    if rel_pos.shape[0] != max_rel:
        rel_pos = F.interpolate(rel_pos.t().unsqueeze(0), size=max_rel, mode="linear").squeeze(0).t()
    q_ratio, k_ratio = max(k_size / q_size, 1.0), max(q_size / k_size, 1.0)
    idx = (torch.arange(q_size)[:, None] * q_ratio - torch.arange(k_size)[None, :] * k_ratio) + (k_size - 1) * k_ratio
    return rel_pos[idx.long()]


def attention_pool(x: Tensor, w: Tensor, stride, size, gamma: Tensor, beta: Tensor, eps: float = 1e-5):
    """x [B, heads, 1+T*H*W, D] -> pooled + normalised [B, heads, 1+T'H'W', D] (mvit.py:446-494; depthwise Conv3d 3x3x3,
    padding 1, class token untouched by the conv, LayerNorm on every token)."""
    B, nh, L, D = x.shape
    T, H, W = size
    cls, tok = x[:, :, :1], x[:, :, 1:]
    t = tok.reshape(B * nh, T, H, W, D).permute(0, 4, 1, 2, 3)
    t = F.conv3d(t, w, None, stride=tuple(stride), padding=1, groups=D)
    out_size = tuple(t.shape[2:])
    t = t.reshape(B, nh, D, -1).transpose(2, 3)
    y = torch.cat([cls, t], dim=2)
    return F.layer_norm(y, (D,), gamma, beta, eps), out_size


def multiscale_attention(sd, p: str, L: LayerCfg, x: Tensor, size) -> Tuple[Tensor, Tuple[int, int, int]]:
    """MultiScaleAttention.forward (mvit.py:548-605)."""
    B, N, _ = x.shape
    hd = L.out_dims // L.heads
    qkv = F.linear(x, sd[p + "qkv.weight"], sd[p + "qkv.bias"]).reshape(B, N, 3, L.heads, hd)
    q, k, v = qkv.permute(2, 0, 3, 1, 4).unbind(0)
    q, q_size = attention_pool(q, sd[p + "pool_q.weight"], L.stride_q, size, sd[p + "norm_q.weight"], sd[p + "norm_q.bias"])
    k, k_size = attention_pool(k, sd[p + "pool_k.weight"], L.stride_kv, size, sd[p + "norm_k.weight"], sd[p + "norm_k.bias"])
    v, _ = attention_pool(v, sd[p + "pool_v.weight"], L.stride_kv, size, sd[p + "norm_v.weight"], sd[p + "norm_v.bias"])
    attn = (q * hd ** -0.5) @ k.transpose(-2, -1)
    # add_decomposed_rel_pos (mvit.py:363-410): bias from the UNSCALED q, video tokens only
    Rt = resize_decomposed_rel_pos(sd[p + "rel_pos_t"], q_size[0], k_size[0])
    Rh = resize_decomposed_rel_pos(sd[p + "rel_pos_h"], q_size[1], k_size[1])
    Rw = resize_decomposed_rel_pos(sd[p + "rel_pos_w"], q_size[2], k_size[2])
    r_q = q[:, :, 1:].reshape(B, L.heads, *q_size, hd)
    rel = (torch.einsum("bythwc,tkc->bythwk", r_q, Rt)[..., :, None, None]
           + torch.einsum("bythwc,hkc->bythwk", r_q, Rh)[..., None, :, None]
           + torch.einsum("bythwc,wkc->bythwk", r_q, Rw)[..., None, None, :])
    bias = torch.zeros_like(attn)
    bias[:, :, 1:, 1:] = rel.reshape(B, L.heads, q_size[0] * q_size[1] * q_size[2], -1)
    attn = (attn + bias).softmax(dim=-1)
    o = attn @ v
    o = torch.cat([o[:, :, :1], o[:, :, 1:] + q[:, :, 1:]], dim=2)       # residual pooling, not on the class token
    o = o.transpose(1, 2).reshape(B, -1, L.out_dims)
    return F.linear(o, sd[p + "proj.weight"], sd[p + "proj.bias"]), q_size


def multiscale_block(sd, i: int, L: LayerCfg, x: Tensor, size):
    """MultiScaleBlock.forward (mvit.py:779-802) with dim_mul_in_attention=True."""
    p = f"blocks.{i}."
    xn = F.layer_norm(x, (L.in_dims,), sd[p + "norm1.weight"], sd[p + "norm1.bias"], 1e-5)
    xa, out_size = multiscale_attention(sd, p + "attn.", L, xn, size)
    skip = F.linear(xn, sd[p + "proj.weight"], sd[p + "proj.bias"]) if L.in_dims != L.out_dims else x
    if max(L.stride_q) > 1:                                                   # pool_skip: MaxPool3d (mvit.py:765-777)
        ks = [s + 1 if s > 1 else s for s in L.stride_q]
        B, _, C = skip.shape
        cls, tok = skip[:, :1], skip[:, 1:]
        t = tok.reshape(B, *size, C).permute(0, 4, 1, 2, 3)
        t = F.max_pool3d(t, ks, tuple(L.stride_q), [k // 2 for k in ks])
        skip = torch.cat([cls, t.reshape(B, C, -1).transpose(1, 2)], dim=1)
    x = skip + xa
    xn2 = F.layer_norm(x, (L.out_dims,), sd[p + "norm2.weight"], sd[p + "norm2.bias"], 1e-5)
    h = F.gelu(F.linear(xn2, sd[p + "mlp.fc1.weight"], sd[p + "mlp.fc1.bias"]))
    return x + F.linear(h, sd[p + "mlp.fc2.weight"], sd[p + "mlp.fc2.bias"]), out_size


def mvit_forward(sd: Dict[str, Tensor], cfg: MViTConfig, x: Tensor, taps: dict = None) -> List[Tensor]:
    """x [B,3,16,H,W] (or [B*16,3,H,W], mvit.py:1091-1092) -> 4 feature maps [B,C,8,h,w], COARSEST FIRST (mvit.py:1143)."""
    if x.dim() == 4:
        x = x.view(-1, x.shape[-3], 16, x.shape[-2], x.shape[-1])
    B = x.shape[0]
    t = F.conv3d(x, sd["patch_embed.projection.weight"], sd["patch_embed.projection.bias"], stride=(2, 4, 4), padding=(1, 3, 3))
    size = tuple(t.shape[2:])
    t = t.flatten(2).transpose(1, 2)
    t = torch.cat([sd["cls_token"].expand(B, -1, -1), t], dim=1)
    if taps is not None:
        taps["tokens0"] = t
    outs = []
    for i, L in enumerate(cfg.layers):
        t, size = multiscale_block(sd, i, L, t, size)
        if taps is not None:
            taps[f"block{i}"] = t
        if L.out_stage >= 0:
            # the normalised tensor REPLACES x and feeds the next block (mvit.py:1123-1126)
            t = F.layer_norm(t, (L.out_dims,), sd[f"norm{L.out_stage}.weight"], sd[f"norm{L.out_stage}.bias"], 1e-5)
            outs.append(t[:, 1:].transpose(1, 2).reshape(B, L.out_dims, *size))
    return outs[::-1]
