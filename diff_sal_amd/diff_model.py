"""``VideoSaliencyModel`` -- the drop-in boundary module (R/models/diff_model.py:8-114).

Same constructor keywords and ``forward(data, t)`` / ``forward_vggish(audio)`` behaviour; the sub-networks may be
given as ready ``nn.Module`` instances or as mmcv-style dicts ``{"type": <class or registered name>, **kwargs}``.
The denoiser (``decoder_net``) is the MI355X-native ``SalUNet``; the once-per-clip encoders (``MViT``, ``VGGish``,
``AudioAttnNet``) have HIP forwards too and are registered under the reference's type names, and any other
``nn.Module`` is accepted in their place.
"""
from __future__ import annotations

import torch
from torch import nn

from . import ops
from .audio_attention import AudioAttnNet
from .mvit import MViT
from .sal_unet import SalUNet
from .vggish import VGGish

OBJECT_REGISTRY = {"SalUNet": SalUNet, "MViT": MViT, "VGGish": VGGish, "AudioAttnNet": AudioAttnNet}


def register_module(cls):
    OBJECT_REGISTRY[cls.__name__] = cls
    return cls


def _build(spec):
    if spec is None or isinstance(spec, nn.Module):
        return spec
    args = dict(spec)
    kind = args.pop("type")
    if isinstance(kind, str):
        kind = OBJECT_REGISTRY[kind]
    return kind(**args)


@register_module
class VideoSaliencyModel(nn.Module):
    def __init__(self, channel_list, visual_net=None, spatiotemp_net=None, audio_net=None, decoder_net=None):
        super().__init__()
        self.visual_net = _build(visual_net)
        self.spatiotemp_net = _build(spatiotemp_net)
        self.audio_net = _build(audio_net)
        if self.audio_net is not None:  # kept for state_dict compatibility; never called (diff_model.py:42-46)
            self.fc = nn.Sequential(nn.Linear(128, 512), nn.ReLU(inplace=True), nn.Linear(512, 768))
        self.decoder_net = _build(decoder_net)
        if channel_list is not None:
            self.channel_list = channel_list

    def forward_vggish(self, audio):
        """audio [B,1,T,H,W] -> (feat, feat) with feat [B,512,T,h,w]   (diff_model.py:70-81)."""
        bs, T = audio.shape[0], audio.shape[2]
        a = audio.reshape(-1, audio.shape[1], audio.shape[3], audio.shape[4])
        if isinstance(self.audio_net, VGGish) and isinstance(self.spatiotemp_net, AudioAttnNet) and a.is_cuda:
            # both stages are ours: stay channels-last tokens [B, T*h*w, C] from the last VGG layer to the end of the
            # transformer (the reference's NCHW -> NCTHW -> tokens -> NCTHW rearranges become one final transpose)
            with torch.no_grad():
                fm = self.audio_net.features_nhwc(a)                      # [(b t), h, w, C]
            _, h, w, c = fm.shape
            self.spatiotemp_net._check_frames(T)
            tok = self.spatiotemp_net.forward_tokens(fm.view(bs, T * h * w, c))
            if tok.requires_grad:                                         # training: keep the transpose on the tape
                from . import encoder_autograd as eg

                f = eg.tokens_to_channels_first(tok, 0).view(bs, c, T, h, w)
            else:
                f = ops.tokens_to_channels_first(tok, 0).view(bs, c, T, h, w)
            return f, f
        with torch.no_grad():
            f = self.audio_net.forward_feat(a)
        f = f.reshape(bs, T, *f.shape[1:]).permute(0, 2, 1, 3, 4).contiguous()
        if self.spatiotemp_net is not None:
            f = self.spatiotemp_net(f)
        return f, f

    def forward(self, data, t):
        """data: {"img": clip, "input": x_t [B,1,H,W], "audio"?: [B,1,T,h,w]}  ->  [B,1,H,W]."""
        imgs = data.get("img", None)
        x = data["input"]
        audio_embed = None
        if self.audio_net:
            _, audio_embed = self.forward_vggish(data.get("audio", None))
        if self.visual_net and imgs is not None:
            vis_list = self.visual_net(imgs)
        else:  # the reference's synthetic-feature fallback (diff_model.py:105-111), sized from x instead of audio
            b, dev = x.shape[0], x.device
            vis_list = [torch.randn((b, c, 8, h, w), device=dev)
                        for c, h, w in ((768, 7, 12), (384, 14, 24), (192, 28, 48), (96, 56, 96))]
        return self.decoder_net(x, t, vis_list, audio_embed)
