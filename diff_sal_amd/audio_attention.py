"""AudioAttnNet -- the 1-layer self-attention transformer over the audio feature map (once per clip), MI355X-native.

Drop-in for ``R/models/audio_attention.py::AudioAttnNet`` (:93-143): same constructor keywords, same ``state_dict``
(including ``to_patch_embedding`` and ``pos_embedding``, which the reference builds but whose result it discards --
quirk Q13, :133-138: ``x`` is overwritten by the rearranged input before the transformer), same
``forward(audio [B,C,T,h,w]) -> [B,C,T,h,w]``.  Tokens are [B, T*h*w, C]; LayerNorms, the qkv / out / MLP GEMMs (fused bias,
GELU, residual) and the attention core (``diffsal_attention_general`` reading q, k, v in place from the fused qkv output)
all run on the HIP path.
"""
from __future__ import annotations

import torch
from torch import nn

from . import ops
from .ops import ACT_GELU

Tensor = torch.Tensor


class _Attention(nn.Module):
    def __init__(self, dim, heads, dim_head, dropout):
        super().__init__()
        inner = dim_head * heads
        self.heads, self.dim_head, self.scale = heads, dim_head, dim_head ** -0.5
        self.norm = nn.LayerNorm(dim)
        self.to_qkv = nn.Linear(dim, inner * 3, bias=False)
        self.project_out = not (heads == 1 and dim_head == dim)
        self.to_out = nn.Sequential(nn.Linear(inner, dim), nn.Dropout(dropout)) if self.project_out else nn.Identity()


class _FeedForward(nn.Module):
    def __init__(self, dim, hidden, dropout):
        super().__init__()
        self.net = nn.Sequential(nn.LayerNorm(dim), nn.Linear(dim, hidden), nn.GELU(), nn.Dropout(dropout), nn.Linear(hidden, dim),
                                 nn.Dropout(dropout))


class Transformer(nn.Module):
    def __init__(self, dim, depth, heads, dim_head, mlp_dim, dropout=0.0):
        super().__init__()
        if dropout != 0.0:
            raise NotImplementedError("AudioAttnNet: dropout > 0 is not built (the shipped config uses 0.0)")
        self.norm = nn.LayerNorm(dim)
        self.layers = nn.ModuleList([nn.ModuleList([_Attention(dim, heads, dim_head, dropout), _FeedForward(dim, mlp_dim, dropout)])
                                     for _ in range(depth)])

    def forward_train(self, x: Tensor) -> Tensor:
        """Same graph on the autograd tape, HIP forward and backward per operator (the reference trains this module:
        spatiotemp_net has requires_grad parameters, R/models/diff_model.py:76-78)."""
        from . import autograd_ops as ag
        from . import encoder_autograd as eg

        B, N, dim = x.shape
        for attn, ff in self.layers:
            x, xn = ag.layernorm_fork(x, attn.norm.weight, attn.norm.bias, attn.norm.eps)
            qkv = ag.linear(xn, attn.to_qkv.weight, None).view(B, N, 3, attn.heads, attn.dim_head)
            q, k, v = (qkv[:, :, i].permute(0, 2, 1, 3).contiguous() for i in range(3))   # head-major copies (layout plumbing)
            o = eg.attention_general(q, k, v, scale=attn.scale)
            x = ag.linear(o, attn.to_out[0].weight, attn.to_out[0].bias, residual=x) if attn.project_out else ag.add(o, x)
            x, y = ag.layernorm_fork(x, ff.net[0].weight, ff.net[0].bias, ff.net[0].eps)
            h = ag.linear(y, ff.net[1].weight, ff.net[1].bias)            # pre-activation: the next node applies the GELU
            x = ag.linear(h, ff.net[4].weight, ff.net[4].bias, residual=x, in_gelu=True)
        return ag.layernorm(x, self.norm.weight, self.norm.bias, self.norm.eps)

    def forward(self, x: Tensor) -> Tensor:
        """x [B, N, dim] -> [B, N, dim]  (R/models/audio_attention.py:63-90)."""
        if torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for p in self.parameters())):
            return self.forward_train(x)
        B, N, dim = x.shape
        for attn, ff in self.layers:
            xn = ops.layernorm(x, attn.norm.weight, attn.norm.bias, attn.norm.eps)
            qkv = ops.linear(xn, attn.to_qkv.weight, None, tag="audio-gemm").view(B, N, 3, attn.heads, attn.dim_head)
            o = ops.attention_general(*(qkv[:, :, i].permute(0, 2, 1, 3) for i in range(3)), scale=attn.scale)
            if attn.project_out:
                x = ops.linear(o, attn.to_out[0].weight, attn.to_out[0].bias, residual=x, tag="audio-gemm")
            else:
                x = ops.axpbypcz(o, 1.0, x, 1.0)
            y = ops.layernorm(x, ff.net[0].weight, ff.net[0].bias, ff.net[0].eps)
            h = ops.linear(y, ff.net[1].weight, ff.net[1].bias, act=ACT_GELU, tag="audio-gemm")
            x = ops.linear(h, ff.net[4].weight, ff.net[4].bias, residual=x, tag="audio-gemm")
        return ops.layernorm(x, self.norm.weight, self.norm.bias, self.norm.eps)


class AudioAttnNet(nn.Module):
    def __init__(self, depth, heads, mlp_dim, dim=512, patch_dim=768, num_patches=16, height=7, width=7, pool="cls",
                 dim_head=64, dropout=0.0, emb_dropout=0.0):
        super().__init__()
        assert pool in {"cls", "mean"}, "pool type must be either cls (cls token) or mean (mean pooling)"
        self.to_patch_embedding = nn.Sequential(nn.LayerNorm(patch_dim), nn.Linear(patch_dim, dim), nn.LayerNorm(dim))
        self.num_patches = num_patches
        self.pos_embedding = nn.Parameter(torch.randn(1, 1, 9, 1, 1))
        self.transformer = Transformer(dim, depth, heads, dim_head, mlp_dim, dropout)

    def _check_frames(self, t: int) -> None:
        """The reference adds ``pos_embedding`` [1,1,9,1,1] to a (discarded) tensor: any other frame count fails there
        with a broadcast error before the transformer runs (audio_attention.py:136); same behaviour here."""
        pt = self.pos_embedding.shape[2]
        if t != pt and pt != 1 and t != 1:
            raise RuntimeError(f"The size of tensor a ({t}) must match the size of tensor b ({pt}) at non-singleton dimension 2")

    def forward_tokens(self, tok: Tensor) -> Tensor:
        """[B, T*h*w, C] -> [B, T*h*w, C]; the part of ``forward`` that computes something (quirk Q13)."""
        return self.transformer(tok.contiguous())

    def forward(self, audio: Tensor) -> Tensor:
        if not audio.is_cuda:
            raise RuntimeError("diff_sal_amd.AudioAttnNet runs on the GPU only (no CPU fallback); got a CPU tensor")
        b, c, t, h, w = audio.shape
        self._check_frames(t)
        if torch.is_grad_enabled() and (audio.requires_grad or any(p.requires_grad for p in self.parameters())):
            from . import encoder_autograd as eg

            return eg.tokens_to_channels_first(self.forward_tokens(eg.pack_tokens(audio.float())), 0).view(b, c, t, h, w)
        tok = ops.pack_frames(audio.contiguous().float(), None).view(b, t * h * w, c)      # b c t h w -> b (t h w) c
        out = self.forward_tokens(tok)
        return ops.tokens_to_channels_first(out, 0).view(b, c, t, h, w)
