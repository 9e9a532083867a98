"""CPU oracle for the DiffSal per-step denoiser (SalUNet) -- TEST INFRASTRUCTURE ONLY.

This is a plain-PyTorch fp32 *restatement* of the reference algorithm, written
functionally over a flat ``state_dict`` (reference key names).
It exists to check the HIP path; nothing under ``diff_sal_amd/`` may import it.
Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg use it.

Parity pin: the reference ships no tests or golden vectors for this path
("parity unpinned" by the reference itself).  This restatement is pinned by
running the *real* reference (imported read-only from /root/reference in the
build container, see ``oracle/gen_golden.py``) on closed-form weights/inputs and
committing its outputs + intermediate taps under ``tests/golden/``;
``tests/test_oracle_golden.py`` re-checks the restatement against those files.

Every function cites the reference lines it follows (R/ = /root/reference/).
Differences from the reference, on purpose:
  * inputs are never mutated (reference defect D4, sal_unet.py:312-317);
  * BatchNorm is evaluated in eval mode (running statistics) only;
  * Dropout / DropPath are identity (eval).
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence

import torch
import torch.nn.functional as F

Tensor = torch.Tensor


@dataclass
class SalUNetConfig:
    """The subset of SalUNet's constructor arguments that changes the graph.

    Defaults = R/cfgs/audio_visual.py:50-82 (identical in R/cfgs/visual.py).
    """

    img_size: Sequence[int] = (224, 384)
    up_channel: Sequence[int] = (768, 384, 192, 96)
    ori_embed_dim: int = 768
    down_embed_dim: int = 96
    num_heads: Sequence[int] = (2, 2, 2, 2)
    kernel_kv: Sequence[int] = (2, 4, 8, 16)
    temporal_list: Sequence[int] = (5, 5, 5, 5)
    dilation: Sequence[int] = (0, 2, 2, 2)  # patch_padding (== dilation), 0 => no UpEmbed
    ch: int = 96  # SalUNet.ch, hard-coded at sal_unet.py:228
    image_based: bool = True

    @property
    def num_stages(self) -> int:
        return len(self.up_channel)


# --------------------------------------------------------------------------
# K1: timestep embedding + MLP
# --------------------------------------------------------------------------
def timestep_embedding(t: Tensor, dim: int) -> Tensor:
    """R/models/saliency_decoder/sal_unet.py:15-33. ``t`` may be int64 or float."""
    half = dim // 2
    freq = torch.exp(torch.arange(half, dtype=torch.float32) * -(math.log(10000.0) / (half - 1)))
    arg = t.to(torch.float32)[:, None] * freq.to(t.device)[None, :]
    emb = torch.cat([arg.sin(), arg.cos()], dim=1)
    if dim % 2:
        emb = F.pad(emb, (0, 1))
    return emb


def swish(x: Tensor) -> Tensor:
    """R/models/saliency_decoder/sal_unet.py:36-38."""
    return x * torch.sigmoid(x)


def temb_mlp(sd: Dict[str, Tensor], t: Tensor, ch: int) -> Tensor:
    """R/models/saliency_decoder/sal_unet.py:304-307."""
    e = timestep_embedding(t, ch)
    e = F.linear(e, sd["temb.dense.0.weight"], sd["temb.dense.0.bias"])
    e = swish(e)
    return F.linear(e, sd["temb.dense.1.weight"], sd["temb.dense.1.bias"])


# --------------------------------------------------------------------------
# K2-K5: noise encoder
# --------------------------------------------------------------------------
def group_norm_swish(x: Tensor, w: Tensor, b: Tensor) -> Tensor:
    """GroupNorm(32, eps=1e-6) then swish. sal_unet.py:41-44, :125-126, :131-132."""
    return swish(F.group_norm(x, 32, w, b, eps=1e-6))


def pad_conv(x: Tensor, w: Tensor, b: Tensor, stride: int) -> Tensor:
    """Zero-pad right/bottom by 1, then 3x3 conv, padding 0. sal_unet.py:57-61, :77-81."""
    return F.conv2d(F.pad(x, (0, 1, 0, 1)), w, b, stride=stride)


def resnet_block(sd: Dict[str, Tensor], p: str, x: Tensor, temb: Tensor) -> Tensor:
    """R/models/saliency_decoder/sal_unet.py:123-142 (eval: dropout is identity)."""
    h = group_norm_swish(x, sd[p + "norm1.weight"], sd[p + "norm1.bias"])
    h = F.conv2d(h, sd[p + "conv1.weight"], sd[p + "conv1.bias"], padding=1)
    h = h + F.linear(swish(temb), sd[p + "temb_proj.weight"], sd[p + "temb_proj.bias"])[:, :, None, None]
    h = group_norm_swish(h, sd[p + "norm2.weight"], sd[p + "norm2.bias"])
    h = F.conv2d(h, sd[p + "conv2.weight"], sd[p + "conv2.bias"], padding=1)
    if (p + "nin_shortcut.weight") in sd:
        x = F.conv2d(x, sd[p + "nin_shortcut.weight"], sd[p + "nin_shortcut.bias"])
    return x + h


def noise_downsample(sd: Dict[str, Tensor], x: Tensor, temb: Tensor, taps=None) -> List[Tensor]:
    """R/models/saliency_decoder/sal_unet.py:279-300. Returns coarsest-first 5-D maps."""
    f = F.conv2d(x, sd["conv_in.weight"], sd["conv_in.bias"], padding=1)
    f = pad_conv(f, sd["down1.conv.weight"], sd["down1.conv.bias"], stride=4)
    if taps is not None:
        taps["down1"] = f
    out = []
    i = 0
    while f"res_encoder.{i}.0.conv1.weight" in sd:
        f = resnet_block(sd, f"res_encoder.{i}.0.", f, temb)
        if taps is not None:
            taps[f"res{i}"] = f
        f = pad_conv(f, sd[f"res_encoder.{i}.1.conv.weight"], sd[f"res_encoder.{i}.1.conv.bias"], stride=2)
        out.append(f.unsqueeze(2))
        i += 1
    return out[::-1]


# --------------------------------------------------------------------------
# K12: UpEmbed
# --------------------------------------------------------------------------
BN_TRAIN = False  # tests of the training step flip this: batch statistics, as nn.BatchNorm2d in train mode


def _bn_eval(x: Tensor, sd: Dict[str, Tensor], p: str) -> Tensor:
    if BN_TRAIN:
        return F.batch_norm(x, None, None, sd[p + "weight"], sd[p + "bias"], True, 0.0, 1e-5)
    return F.batch_norm(
        x, sd[p + "running_mean"], sd[p + "running_var"], sd[p + "weight"], sd[p + "bias"], False, 0.0, 1e-5
    )


def up_embed(sd: Dict[str, Tensor], p: str, x: Tensor, dil: int) -> Tensor:
    """R/models/saliency_decoder/common_block.py:196-223; x is [B,C,T,h,w]."""
    B, C, T, h, w = x.shape
    y = x.permute(0, 2, 1, 3, 4).reshape(B * T, C, h, w)
    y = F.interpolate(y, scale_factor=2, mode="bilinear", align_corners=False)
    y = F.relu(_bn_eval(F.conv2d(y, sd[p + "proj.1.weight"], None, padding=dil, dilation=dil), sd, p + "proj.2."))
    y = F.relu(_bn_eval(F.conv2d(y, sd[p + "proj.4.weight"], None, padding=dil, dilation=dil), sd, p + "proj.5."))
    Co, H, W = y.shape[1:]
    return y.reshape(B, T, Co, H, W).permute(0, 2, 1, 3, 4).contiguous()


# --------------------------------------------------------------------------
# K7: audio fusion
# --------------------------------------------------------------------------
def audio_fusion(sd: Dict[str, Tensor], p: str, x5: Tensor, audio: Tensor) -> Tensor:
    """R/models/saliency_decoder/transformer.py:128-146.

    x5 [B,C,T,H,W], audio [B,512,T,h,w]  ->  audio tokens [B*T, H*W, C] obtained
    by *reinterpreting* the contiguous [B,C,T,H,W] result (quirk Q5, :146).
    """
    B, C, T, H, W = x5.shape
    Ta = audio.shape[2]
    a = audio.permute(0, 2, 1, 3, 4).reshape(B * Ta, audio.shape[1], *audio.shape[3:])
    a = F.conv2d(a, sd[p + "align_conv.weight"], sd[p + "align_conv.bias"])
    h, w = a.shape[-2:]
    if h != H and w != W:  # quirk Q3: `and`, nearest, factor H // h   (:133-136)
        a = F.interpolate(a, scale_factor=H // h, mode="nearest")
    a = a.reshape(B, Ta, C, *a.shape[-2:]).permute(0, 2, 1, 3, 4)
    m = (a * x5).mean(dim=2, keepdim=True)  # adaptive_avg_pool3d -> (1,H,W)   (:141-143)
    m = F.softmax(m, dim=-1)  # quirk Q4: over W only (:144)
    a = (a * m).contiguous()
    return a.view(B * T, -1, C)


# --------------------------------------------------------------------------
# K8-K11: attention + MLP
# --------------------------------------------------------------------------
def _tokens_to_map(tok: Tensor, h: int, w: int) -> Tensor:
    n, _, c = tok.shape
    return tok.reshape(n, h, w, c).permute(0, 3, 1, 2)


def _map_to_tokens(m: Tensor) -> Tensor:
    return m.flatten(2).transpose(1, 2)


def attention(
    sd: Dict[str, Tensor], p: str, xn: Tensor, h: int, w: int, heads: int, kkv: int, audio_tok: Optional[Tensor]
) -> Tensor:
    """R/models/saliency_decoder/attention.py:86-113 with fea_no == 1 (quirk Q1).

    The q projection is a depthwise Conv3d 3x3x3 on a T == 1 volume with temporal
    padding 1, i.e. a 2-D depthwise 3x3 with the centre temporal slice (quirk Q8).
    """
    C = xn.shape[-1]
    xm = _tokens_to_map(xn, h, w)
    km = xm if audio_tok is None else _tokens_to_map(audio_tok, h, w)

    wq = sd[p + "conv_proj_q.conv.weight"][:, :, 1]  # [C,1,3,3]
    wk = sd[p + "conv_proj_k.conv.weight"][:, :, 0]  # [C,1,k,k]
    wv = sd[p + "conv_proj_v.conv.weight"][:, :, 0]
    q = _map_to_tokens(F.conv2d(xm, wq, None, padding=1, groups=C))
    k = _map_to_tokens(F.conv2d(km, wk, None, stride=kkv, groups=C))
    v = _map_to_tokens(F.conv2d(xm, wv, None, stride=kkv, groups=C))
    q = F.layer_norm(q, (C,), sd[p + "conv_proj_q.bn.weight"], sd[p + "conv_proj_q.bn.bias"], 1e-5)
    k = F.layer_norm(k, (C,), sd[p + "conv_proj_k.bn.weight"], sd[p + "conv_proj_k.bn.bias"], 1e-5)
    v = F.layer_norm(v, (C,), sd[p + "conv_proj_v.bn.weight"], sd[p + "conv_proj_v.bn.bias"], 1e-5)

    q = F.linear(q, sd[p + "proj_q.weight"], sd.get(p + "proj_q.bias"))
    k = F.linear(k, sd[p + "proj_k.weight"], sd.get(p + "proj_k.bias"))
    v = F.linear(v, sd[p + "proj_v.weight"], sd.get(p + "proj_v.bias"))

    n, lq, _ = q.shape
    d = C // heads
    qh = q.reshape(n, lq, heads, d).transpose(1, 2)
    kh = k.reshape(n, -1, heads, d).transpose(1, 2)
    vh = v.reshape(n, -1, heads, d).transpose(1, 2)
    s = (qh @ kh.transpose(-1, -2)) * (C ** -0.5)  # quirk Q6: scale uses full C (:33,:101)
    o = F.softmax(s, dim=-1) @ vh
    o = o.transpose(1, 2).reshape(n, lq, C)
    return F.linear(o, sd[p + "proj.weight"], sd[p + "proj.bias"])


def transformer_block(
    sd: Dict[str, Tensor], p: str, x5: Tensor, heads: int, kkv: int, audio: Optional[Tensor]
) -> Tensor:
    """R/models/saliency_decoder/transformer.py:124-159 + fold/unfold :273-287."""
    B, C, T, H, W = x5.shape
    audio_tok = audio_fusion(sd, p, x5, audio) if audio is not None else None
    x = x5.permute(0, 2, 3, 4, 1).reshape(B * T, H * W, C)
    xn = F.layer_norm(x, (C,), sd[p + "norm.weight"], sd[p + "norm.bias"], 1e-5)
    x = attention(sd, p + "attn.", xn, H, W, heads, kkv, audio_tok) + x
    y = F.layer_norm(x, (C,), sd[p + "norm2.weight"], sd[p + "norm2.bias"], 1e-5)
    y = F.linear(y, sd[p + "mlp.fc1.weight"], sd[p + "mlp.fc1.bias"])
    y = F.gelu(y)  # exact erf GELU (nn.GELU default), common_block.py:137
    y = F.linear(y, sd[p + "mlp.fc2.weight"], sd[p + "mlp.fc2.bias"])
    x = x + y
    return x.reshape(B, T, H, W, C).permute(0, 4, 1, 2, 3).contiguous()


# --------------------------------------------------------------------------
# K13/K14 + decoder driver
# --------------------------------------------------------------------------
def decoder(
    sd: Dict[str, Tensor], cfg: SalUNetConfig, back_fea: List[Tensor], audio: Optional[Tensor], taps=None
) -> Tensor:
    """R/models/saliency_decoder/sal_unet.py:457-491 and transformer.py:259-289."""
    pre = "invpt_decoder."
    x = back_fea[0]
    h, w = x.shape[3:]
    ns = cfg.num_stages
    th, tw = h * 2 ** (ns - 1) * 2, w * 2 ** (ns - 1) * 2
    acc = 0
    for i in range(ns):
        sp = f"{pre}mid_stages.{i}."
        if cfg.dilation[i] != 0:
            x = up_embed(sd, sp + "patch_embed.0.", x, cfg.dilation[i])
            if i in (1, 2):  # quirk Q2: skip only at stages 1 and 2 (transformer.py:265-270)
                x = x + back_fea[i]
        x = transformer_block(sd, sp + "blocks.0.", x, cfg.num_heads[i], cfg.kernel_kv[i], audio)
        if taps is not None:
            taps[f"stage{i}"] = x
        B, C, T, H, W = x.shape
        z = x.permute(0, 2, 3, 4, 1)  # [B,T,H,W,C]
        z = F.layer_norm(z, (C,), sd[f"{pre}norm_mts.{i}.weight"], sd[f"{pre}norm_mts.{i}.bias"], 1e-5)
        z = z.permute(0, 4, 1, 2, 3)
        kt = cfg.temporal_list[i]
        z = F.relu(F.conv3d(z, sd[f"{pre}redu_chan_up.{i}.proj.0.weight"], None, stride=(kt, 1, 1)))
        z = z.squeeze(2)  # quirk Q10: T=9, k=s=5 -> one output frame from frames 0..4
        z = F.interpolate(z, size=(th, tw), mode="bilinear", align_corners=False)
        acc = acc + z
    if taps is not None:
        taps["multi_scale"] = acc
    y = F.conv2d(acc, sd[pre + "mt_proj.0.weight"], sd[pre + "mt_proj.0.bias"], padding=1)
    y = F.relu(_bn_eval(y, sd, pre + "mt_proj.1."))
    return y


def salunet_forward(
    sd: Dict[str, Tensor],
    cfg: SalUNetConfig,
    x: Tensor,
    t: Tensor,
    feat_list: Sequence[Tensor],
    audio: Optional[Tensor] = None,
    taps: Optional[dict] = None,
) -> Tensor:
    """R/models/saliency_decoder/sal_unet.py:302-328, non-mutating."""
    temb = temb_mlp(sd, t, cfg.ch)
    if taps is not None:
        taps["temb"] = temb
    noise = noise_downsample(sd, x, temb, taps)
    feats = list(feat_list)
    if cfg.image_based:
        for i in range(min(len(feats), len(noise))):
            if feats[i].shape[-2:] == noise[i].shape[-2:]:
                feats[i] = torch.cat([feats[i], noise[i]], dim=2)  # noise = last frame (Q2)
    if taps is not None:
        for i, n in enumerate(noise):
            taps[f"noise{i}"] = n
    y = decoder(sd, cfg, feats, audio, taps)
    y = torch.sigmoid(F.conv2d(y, sd["logits.linear_pred.weight"], sd["logits.linear_pred.bias"]))
    return F.interpolate(y, size=tuple(cfg.img_size), mode="bilinear", align_corners=False)


# --------------------------------------------------------------------------
# Deterministic parameters / inputs shared by the fixture generator, the
# tests, smoke() and bench.py (closed-form in (name, shape); no files needed).
# --------------------------------------------------------------------------
def _rng(name: str):
    import zlib

    import numpy as np

    return np.random.default_rng(zlib.crc32(name.encode()))


def synth_tensor(name: str, shape, scale: float = 1.0, shift: float = 0.0) -> Tensor:
    import numpy as np

    a = _rng(name).standard_normal(size=tuple(shape), dtype=np.float64) * scale + shift
    return torch.from_numpy(a.astype(np.float32))


def synth_state_dict(template: Dict[str, Tensor]) -> Dict[str, Tensor]:
    """He-scaled deterministic fill for every tensor of a SalUNet ``state_dict``.

    The reference init (std 0.01, sal_unet.py:263-277) gives an almost constant
    output; He scaling exercises the whole dynamic range.
    """
    out = {}
    for k, v in template.items():
        shp = tuple(v.shape)
        if k.endswith("num_batches_tracked"):
            out[k] = torch.zeros((), dtype=torch.int64)
        elif k.endswith("running_var"):
            out[k] = synth_tensor(k, shp).abs() * 0.3 + 0.7
        elif k.endswith("running_mean"):
            out[k] = synth_tensor(k, shp, 0.1)
        elif len(shp) >= 2:
            fan_in = 1
            for s in shp[1:]:
                fan_in *= s
            out[k] = synth_tensor(k, shp, math.sqrt(2.0 / fan_in))
        elif k.endswith("bias"):
            out[k] = synth_tensor(k, shp, 0.05)
        else:  # 1-D norm weights
            out[k] = synth_tensor(k, shp, 0.1, 1.0)
    return out


def synth_inputs(cfg: SalUNetConfig, batch: int, audio: bool, tag: str = "in", frames: int = 8):
    """x_t, visual feature list (coarsest first, diff_model.py:105-111 shapes), audio map."""
    H, W = cfg.img_size
    x = synth_tensor(f"{tag}.x", (batch, 1, H, W))
    ns = cfg.num_stages
    feats = []
    for i, c in enumerate(cfg.up_channel):
        s = 2 ** (ns + 1 - i)  # 32,16,8,4
        feats.append(synth_tensor(f"{tag}.feat{i}", (batch, c, frames, H // s, W // s)))
    a = synth_tensor(f"{tag}.audio", (batch, 512, frames + 1, H // 32, W // 32)) if audio else None
    return x, feats, a


def state_dict_template(cfg: SalUNetConfig) -> Dict[str, Tensor]:
    """Names and shapes of SalUNet.state_dict(), as empty tensors."""
    sd: Dict[str, Tensor] = {}

    def E(name, *shape):
        sd[name] = torch.empty(*shape)

    def bn(p, c):
        E(p + "weight", c), E(p + "bias", c), E(p + "running_mean", c), E(p + "running_var", c)
        sd[p + "num_batches_tracked"] = torch.zeros((), dtype=torch.int64)

    tc = cfg.ch * 4
    E("temb.dense.0.weight", tc, cfg.ch), E("temb.dense.0.bias", tc)
    E("temb.dense.1.weight", tc, tc), E("temb.dense.1.bias", tc)
    E("conv_in.weight", cfg.ch, 1, 3, 3), E("conv_in.bias", cfg.ch)
    E("down1.conv.weight", cfg.ch, cfg.ch, 3, 3), E("down1.conv.bias", cfg.ch)
    cin = cfg.ch
    for i, co in enumerate(list(cfg.up_channel[:-1])[::-1]):
        p = f"res_encoder.{i}.0."
        E(p + "norm1.weight", cin), E(p + "norm1.bias", cin)
        E(p + "conv1.weight", co, cin, 3, 3), E(p + "conv1.bias", co)
        E(p + "temb_proj.weight", co, tc), E(p + "temb_proj.bias", co)
        E(p + "norm2.weight", co), E(p + "norm2.bias", co)
        E(p + "conv2.weight", co, co, 3, 3), E(p + "conv2.bias", co)
        if cin != co:
            E(p + "nin_shortcut.weight", co, cin, 1, 1), E(p + "nin_shortcut.bias", co)
        E(f"res_encoder.{i}.1.conv.weight", co, co, 3, 3), E(f"res_encoder.{i}.1.conv.bias", co)
        cin = co
    prev = cfg.ori_embed_dim
    for s, c in enumerate(cfg.up_channel):
        sp = f"invpt_decoder.mid_stages.{s}."
        if cfg.dilation[s] != 0:
            E(sp + "patch_embed.0.proj.1.weight", c, prev, 3, 3)
            bn(sp + "patch_embed.0.proj.2.", c)
            E(sp + "patch_embed.0.proj.4.weight", c, c, 3, 3)
            bn(sp + "patch_embed.0.proj.5.", c)
        b = sp + "blocks.0."
        hid = int(c * 2.0)
        E(b + "mlp.fc1.weight", hid, c), E(b + "mlp.fc1.bias", hid)
        E(b + "mlp.fc2.weight", c, hid), E(b + "mlp.fc2.bias", c)
        E(b + "norm.weight", c), E(b + "norm.bias", c)
        k = cfg.kernel_kv[s]
        E(b + "attn.conv_proj_q.conv.weight", c, 1, 3, 3, 3)
        E(b + "attn.conv_proj_q.bn.weight", c), E(b + "attn.conv_proj_q.bn.bias", c)
        for n in ("k", "v"):
            E(b + f"attn.conv_proj_{n}.conv.weight", c, 1, 1, k, k)
            E(b + f"attn.conv_proj_{n}.bn.weight", c), E(b + f"attn.conv_proj_{n}.bn.bias", c)
        for n in ("proj_q", "proj_k", "proj_v", "proj"):
            E(b + f"attn.{n}.weight", c, c), E(b + f"attn.{n}.bias", c)
        E(b + "norm2.weight", c), E(b + "norm2.bias", c)
        E(b + "align_conv.weight", c, 512, 1, 1), E(b + "align_conv.bias", c)
        prev = c
    for s, c in enumerate(cfg.up_channel):
        E(f"invpt_decoder.norm_mts.{s}.weight", c), E(f"invpt_decoder.norm_mts.{s}.bias", c)
    for s, c in enumerate(cfg.up_channel):
        E(f"invpt_decoder.redu_chan_up.{s}.proj.0.weight", cfg.ori_embed_dim, c, cfg.temporal_list[s], 1, 1)
    E("invpt_decoder.mt_proj.0.weight", cfg.down_embed_dim, cfg.ori_embed_dim, 3, 3)
    E("invpt_decoder.mt_proj.0.bias", cfg.down_embed_dim)
    bn("invpt_decoder.mt_proj.1.", cfg.down_embed_dim)
    E("logits.linear_pred.weight", 1, cfg.down_embed_dim, 1, 1), E("logits.linear_pred.bias", 1)
    return sd
