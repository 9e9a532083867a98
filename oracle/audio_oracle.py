"""CPU restatement of the reference's audio branch (VGGish feature stack + AudioAttnNet) -- TEST INFRASTRUCTURE ONLY.

Plain fp32 PyTorch, functional over state_dicts with the reference's names; cites R/models/vggish.py and
R/models/audio_attention.py.  Only tests/, smoke() and bench.py's cpu_baseline leg may import it.  Pinned against the
real reference by oracle/gen_golden.py::gen_audio (tests/golden/audio_*.npz).
"""
from __future__ import annotations

import math
from typing import Dict

import torch
import torch.nn.functional as F

Tensor = torch.Tensor
VGG_CFG = [64, "M", 128, "M", 256, 256, "M", 512, 512, "M"]            # vggish.py:99-109


def vgg_template() -> Dict[str, Tensor]:
    sd, cin, idx = {}, 1, 0
    for v in VGG_CFG:
        if v == "M":
            idx += 1
        else:
            sd[f"features.{idx}.weight"] = torch.empty(v, cin, 3, 3)
            sd[f"features.{idx}.bias"] = torch.empty(v)
            cin, idx = v, idx + 2
    for i, (o, k) in zip((0, 2, 4), ((4096, 512 * 4 * 6), (4096, 4096), (128, 4096))):
        sd[f"embeddings.{i}.weight"] = torch.empty(o, k)
        sd[f"embeddings.{i}.bias"] = torch.empty(o)
    return sd


def attn_template(dim=512, heads=2, dim_head=64, mlp_dim=256, patch_dim=512, depth=1) -> Dict[str, Tensor]:
    sd = {"pos_embedding": torch.empty(1, 1, 9, 1, 1)}
    for n, c in (("to_patch_embedding.0", patch_dim), ("to_patch_embedding.2", dim), ("transformer.norm", dim)):
        sd[n + ".weight"], sd[n + ".bias"] = torch.empty(c), torch.empty(c)
    sd["to_patch_embedding.1.weight"], sd["to_patch_embedding.1.bias"] = torch.empty(dim, patch_dim), torch.empty(dim)
    inner = heads * dim_head
    for l in range(depth):
        p = f"transformer.layers.{l}."
        sd[p + "0.norm.weight"], sd[p + "0.norm.bias"] = torch.empty(dim), torch.empty(dim)
        sd[p + "0.to_qkv.weight"] = torch.empty(3 * inner, dim)
        sd[p + "0.to_out.0.weight"], sd[p + "0.to_out.0.bias"] = torch.empty(dim, inner), torch.empty(dim)
        sd[p + "1.net.0.weight"], sd[p + "1.net.0.bias"] = torch.empty(dim), torch.empty(dim)
        sd[p + "1.net.1.weight"], sd[p + "1.net.1.bias"] = torch.empty(mlp_dim, dim), torch.empty(mlp_dim)
        sd[p + "1.net.4.weight"], sd[p + "1.net.4.bias"] = torch.empty(dim, mlp_dim), torch.empty(dim)
    return sd


def synth_state_dict(template: Dict[str, Tensor], tag: str) -> Dict[str, Tensor]:
    from oracle.salunet_oracle import synth_tensor

    out = {}
    for k, v in template.items():
        shp = tuple(v.shape)
        if k.startswith("embeddings"):                       # 84 M values nobody reads in forward_feat: cheap constant fill
            out[k] = torch.full(shp, 0.001)
        elif len(shp) >= 2 and k != "pos_embedding":
            fan_in = 1
            for s in shp[1:]:
                fan_in *= s
            out[k] = synth_tensor(tag + k, shp, math.sqrt(2.0 / fan_in))
        elif k.endswith("weight"):
            out[k] = 1.0 + synth_tensor(tag + k, shp, 0.1)
        else:
            out[k] = synth_tensor(tag + k, shp, 0.1)
    return out


def vgg_features(sd: Dict[str, Tensor], x: Tensor) -> Tensor:
    """VGG.forward_feat (vggish.py:93-95): x [N,1,H,W] -> [N,512,H/16,W/16]."""
    idx = 0
    for v in VGG_CFG:
        if v == "M":
            x = F.max_pool2d(x, 2, 2)
            idx += 1
        else:
            x = F.relu(F.conv2d(x, sd[f"features.{idx}.weight"], sd[f"features.{idx}.bias"], padding=1))
            idx += 2
    return x


def audio_attn_forward(sd: Dict[str, Tensor], audio: Tensor, heads: int = 2, dim_head: int = 64) -> Tensor:
    """AudioAttnNet.forward (audio_attention.py:130-143): only the transformer acts on the output; the patch embedding and
    the position embedding are computed into a tensor that is then overwritten (quirk Q13)."""
    b, c, t, h, w = audio.shape
    x = audio.permute(0, 2, 3, 4, 1).reshape(b, t * h * w, c)
    l = 0
    while f"transformer.layers.{l}.0.norm.weight" in sd:
        p = f"transformer.layers.{l}."
        xn = F.layer_norm(x, (c,), sd[p + "0.norm.weight"], sd[p + "0.norm.bias"])
        q, k, v = (z.reshape(b, -1, heads, dim_head).transpose(1, 2) for z in F.linear(xn, sd[p + "0.to_qkv.weight"]).chunk(3, dim=-1))
        o = (torch.softmax(q @ k.transpose(-1, -2) * dim_head ** -0.5, dim=-1) @ v).transpose(1, 2).reshape(b, -1, heads * dim_head)
        x = F.linear(o, sd[p + "0.to_out.0.weight"], sd[p + "0.to_out.0.bias"]) + x
        y = F.layer_norm(x, (c,), sd[p + "1.net.0.weight"], sd[p + "1.net.0.bias"])
        x = F.linear(F.gelu(F.linear(y, sd[p + "1.net.1.weight"], sd[p + "1.net.1.bias"])), sd[p + "1.net.4.weight"], sd[p + "1.net.4.bias"]) + x
        l += 1
    x = F.layer_norm(x, (c,), sd["transformer.norm.weight"], sd["transformer.norm.bias"])
    return x.reshape(b, t, h, w, c).permute(0, 4, 1, 2, 3).contiguous()


def audio_branch(vgg_sd, attn_sd, audio: Tensor) -> Tensor:
    """VideoSaliencyModel.forward_vggish (R/models/diff_model.py:70-81): audio [B,1,T,H,W] -> [B,512,T,H/16,W/16]."""
    bs, T = audio.shape[0], audio.shape[2]
    f = vgg_features(vgg_sd, audio.reshape(-1, audio.shape[1], audio.shape[3], audio.shape[4]))
    f = f.reshape(bs, T, *f.shape[1:]).permute(0, 2, 1, 3, 4)
    return audio_attn_forward(attn_sd, f)
