"""MViTv2 video encoder (once per clip), MI355X-native forward.

Drop-in for the reference ``models/mvit.py::MViT`` as DiffSal configures it (R/cfgs/audio_visual.py:28-33:
``arch="small", out_scales=[0,1,2,3]``): same constructor keywords for that configuration, same ``state_dict`` names and
shapes (so the Kinetics checkpoint the reference loads keeps loading), same ``forward(x)`` contract -- a clip
[B,3,16,H,W] (or [B*16,3,H,W]) in, four NCTHW feature maps [B,C,8,h,w] out, COARSEST FIRST (R/models/mvit.py:1090-1143).

The nn layers below are parameter storage only.  The forward is a sequence of HIP kernels reached through the C ABI:
  * every Linear (qkv, proj, MLP, dim-change skip, the patch projection after an im2col) is the implicit-GEMM kernel with
    fused bias / GELU / residual epilogues;
  * pooling attention = ``pool3d_ln`` (depthwise Conv3d + LayerNorm, reading q / k / v in place from the fused qkv
    output) -> ``relpos_project`` -> ``attention_general`` (flash-style fp32-MFMA attention; the decomposed
    relative-position bias rides in 48 extra contraction columns; residual pooling is its epilogue);
  * tokens stay [B, 1 + T*H*W, C] throughout; the only layout change is the final transpose to NCTHW per output scale.

Built: the configuration the reference uses -- class token on, relative positions on, residual pooling on,
``dim_mul_in_attention=True``, no absolute position embedding.

Training (the reference trains the visual encoder inside the diffusion step, R/diffusion_trainer.py:212-235): when
gradients are enabled and a parameter requires them, ``forward`` runs the same graph on the autograd tape with every
operator an ``autograd.Function`` whose forward AND backward are HIP kernels (``autograd_ops.py``,
``encoder_autograd.py``: flash-attention backward with the rel-pos columns, depthwise-pool / max-pool / rel-pos-projection
backward; GEMM data and weight gradients through the implicit-GEMM and wgrad kernels).  DropPath is 0 in the shipped
configuration (``drop_path_rate`` default) and is not built.
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence

import torch
import torch.nn.functional as F
from torch import nn

from . import ops
from .ops import ACT_GELU

Tensor = torch.Tensor

ARCH_ZOO = {  # R/models/mvit.py:888-917
    "tiny": dict(embed_dims=96, num_layers=10, num_heads=1, downscale_indices=[1, 3, 8]),
    "small": dict(embed_dims=96, num_layers=16, num_heads=1, downscale_indices=[1, 3, 14]),
    "base": dict(embed_dims=96, num_layers=24, num_heads=1, downscale_indices=[2, 5, 21]),
    "large": dict(embed_dims=144, num_layers=48, num_heads=2, downscale_indices=[2, 8, 44]),
}


class _Attn(nn.Module):
    def __init__(self, in_dims, out_dims, heads, rel_hw, rel_t):
        super().__init__()
        hd = out_dims // heads
        self.rel_pos_h = nn.Parameter(torch.zeros(rel_hw, hd))
        self.rel_pos_w = nn.Parameter(torch.zeros(rel_hw, hd))
        self.rel_pos_t = nn.Parameter(torch.zeros(rel_t, hd))
        self.qkv = nn.Linear(in_dims, 3 * out_dims)
        self.proj = nn.Linear(out_dims, out_dims)
        for w in "qkv":
            setattr(self, f"pool_{w}", nn.Conv3d(hd, hd, 3, padding=1, groups=hd, bias=False))
            setattr(self, f"norm_{w}", nn.LayerNorm(hd))


class _Block(nn.Module):
    def __init__(self, in_dims, out_dims, heads, stride_q, stride_kv, rel_hw, rel_t):
        super().__init__()
        self.in_dims, self.out_dims, self.heads = in_dims, out_dims, heads
        self.stride_q, self.stride_kv = tuple(stride_q), tuple(stride_kv)
        self.norm1 = nn.LayerNorm(in_dims)
        self.attn = _Attn(in_dims, out_dims, heads, rel_hw, rel_t)
        self.norm2 = nn.LayerNorm(out_dims)
        mlp = nn.Module()
        mlp.fc1, mlp.fc2 = nn.Linear(out_dims, 4 * out_dims), nn.Linear(4 * out_dims, out_dims)
        self.mlp = mlp
        if in_dims != out_dims:
            self.proj = nn.Linear(in_dims, out_dims)


class MViT(nn.Module):
    """Keyword names follow R/models/mvit.py:919-944; unsupported non-default values raise."""

    def __init__(self, arch="base", spatial_size=224, temporal_size=16, in_channels=3, pretrained: Optional[str] = None,
                 out_scales=-1, drop_path_rate=0.0, use_abs_pos_embed=False, interpolate_mode="trilinear",
                 pool_kernel=(3, 3, 3), dim_mul=2, head_mul=2, adaptive_kv_stride=(1, 8, 8), rel_pos_embed=True,
                 residual_pooling=True, dim_mul_in_attention=True, with_cls_token=True, output_cls_token=False,
                 rel_pos_zero_init=False, mlp_ratio=4.0, qkv_bias=True, norm_layer=nn.LayerNorm,
                 compute_dtype: torch.dtype = torch.float32):
        """``compute_dtype`` (inference only; not a reference argument): torch.bfloat16 / torch.float16 keep the token stream and
        the weights of the token GEMMs in that storage type (native 16-bit MFMA, fp32 accumulation); the patch embedding, the
        pooling convolutions, every LayerNorm statistic, the relative-position projections and the attention core stay fp32."""
        super().__init__()
        if compute_dtype not in (torch.float32, torch.bfloat16, torch.float16):
            raise ValueError(f"MViT: compute_dtype {compute_dtype}")
        self.compute_dtype = compute_dtype
        if (use_abs_pos_embed or tuple(pool_kernel) != (3, 3, 3) or dim_mul != 2 or head_mul != 2 or not rel_pos_embed
                or not residual_pooling or not dim_mul_in_attention or not with_cls_token or output_cls_token
                or mlp_ratio != 4.0 or not qkv_bias or norm_layer is not nn.LayerNorm or in_channels != 3):
            raise NotImplementedError("MViT: only the configuration DiffSal uses is built (class token, relative positions, "
                                      "residual pooling, dim_mul_in_attention, 3x3x3 pooling, mlp_ratio 4)")
        a = dict(ARCH_ZOO[arch.lower()]) if isinstance(arch, str) else dict(arch)
        self.embed_dims, self.num_layers = a["embed_dims"], a["num_layers"]
        if self.embed_dims != 96:
            raise NotImplementedError("MViT: the reference hard-codes a 96-channel patch embedding (mvit.py:983-989)")
        down = list(a["downscale_indices"])
        dim_mul_idx = list(a.get("dim_mul_indices", down))
        self.num_scales = len(down) + 1
        stage_of = {idx - 1: i for i, idx in enumerate(down)}
        stage_of[self.num_layers - 1] = self.num_scales - 1
        scales = [out_scales] if isinstance(out_scales, int) else list(out_scales)
        self.out_scales = sorted(s + self.num_scales if s < 0 else s for s in scales)
        pe = nn.Module()
        pe.projection = nn.Conv3d(3, 96, (3, 7, 7), stride=(2, 4, 4), padding=(1, 3, 3))
        self.patch_embed = pe
        self.cls_token = nn.Parameter(torch.zeros(1, 1, self.embed_dims))
        self.blocks = nn.ModuleList()
        self.stage_of_layer: Dict[int, int] = {}
        heads, dims, stride_kv, size = a["num_heads"], self.embed_dims, tuple(adaptive_kv_stride), (8, 56, 56)
        for i in range(self.num_layers):                       # R/models/mvit.py:1011-1053
            if i in down or i in dim_mul_idx:
                heads *= head_mul
            if i in down:
                stride_q = (1, 2, 2)
                stride_kv = tuple(max(s // 2, 1) for s in stride_kv)
            else:
                stride_q = (1, 1, 1)
            out_dims = dims * dim_mul if i in dim_mul_idx else dims
            if out_dims // heads != 96:
                raise NotImplementedError(f"MViT: head dimension {out_dims // heads} (the attention kernel is built for 96)")
            rel_hw = 2 * max(size[1] // stride_q[1], size[1] // stride_kv[1]) - 1
            self.blocks.append(_Block(dims, out_dims, heads, stride_q, stride_kv, rel_hw, 2 * size[0] - 1))
            size = tuple(s // q for s, q in zip(size, stride_q))
            dims = out_dims
            if i in stage_of and stage_of[i] in self.out_scales:
                self.stage_of_layer[i] = stage_of[i]
                self.add_module(f"norm{stage_of[i]}", nn.LayerNorm(out_dims))
        self._tables: Dict = {}
        self._const_tables: Dict = {}      # parameter-independent: one-hot key columns, relative-position gather indices
        self._pack: Optional[Dict[str, Tensor]] = None
        self._pack_key = None
        self._pack_epoch = 0
        self._init_weights(rel_pos_zero_init)
        if pretrained:
            self.init_weights(pretrained)

    def _init_weights(self, rel_zero: bool):
        for m in self.modules():
            if isinstance(m, (nn.Linear, nn.Conv3d)):
                nn.init.trunc_normal_(m.weight, std=0.02)
                if getattr(m, "bias", None) is not None:
                    nn.init.zeros_(m.bias)
        if not rel_zero:
            for blk in self.blocks:
                for p in (blk.attn.rel_pos_h, blk.attn.rel_pos_w, blk.attn.rel_pos_t):
                    nn.init.trunc_normal_(p, std=0.02)

    def init_weights(self, pretrained: str) -> None:
        """Load a ``backbone.``-prefixed checkpoint; relative-position tables of another length are linearly resampled
        (R/models/mvit.py:1057-1088)."""
        ck = torch.load(pretrained, map_location="cpu")
        ck = ck.get("state_dict", ck)
        sd = {k[len("backbone."):]: v for k, v in ck.items() if k.startswith("backbone.")}
        if not sd:
            raise ValueError(f"backbone. is not in the pretrained model {pretrained}")
        mine = self.state_dict()
        for k in [k for k in sd if "attn.rel_pos" in k and k in mine]:
            (l1, d1), (l2, d2) = sd[k].shape, mine[k].shape
            if d1 == d2 and l1 != l2:
                sd[k] = F.interpolate(sd[k].t().unsqueeze(0), size=l2, mode="linear").view(d2, l2).permute(1, 0)
        self.load_state_dict(sd, strict=False)

    # ------------------------------------------------------------------ weight / table packing
    def _key(self):
        return (self._pack_epoch, self.compute_dtype) + tuple((p.data_ptr(), p._version) for p in self.parameters())

    def parameters_updated(self) -> None:
        """Parameters were rewritten behind autograd's version counters (the fused Adam kernel writes the flat buffer through
        raw pointers; ``param.data.copy_`` does not bump ``_version`` either): the packed patch / pooling weights and the
        gathered relative-position tables are rebuilt on next use."""
        self._pack_epoch += 1
        self._pack, self._pack_key, self._tables = None, None, {}

    def packed(self) -> Dict[str, Tensor]:
        key = self._key()
        if self._pack is not None and key == self._pack_key:
            return self._pack
        pk: Dict[str, Tensor] = {}
        w = self.patch_embed.projection.weight.detach().reshape(96, -1)            # [96, 441], k = (c, kt, ky, kx)
        pk["patch.w"] = F.pad(w, (0, 448 - w.shape[1])).contiguous()
        for i, blk in enumerate(self.blocks):
            a = blk.attn
            for n in "qkv":
                pk[f"b{i}.pool_{n}"] = getattr(a, f"pool_{n}").weight.detach().reshape(96, 27).t().contiguous()   # [27][D]
            if self.compute_dtype != torch.float32:            # 16-bit storage of the token GEMM weights
                for n, lin in (("qkv", a.qkv), ("aproj", a.proj), ("fc1", blk.mlp.fc1), ("fc2", blk.mlp.fc2)):
                    pk[f"b{i}.{n}.w"] = ops.cast(lin.weight.detach(), self.compute_dtype)
                if hasattr(blk, "proj"):
                    pk[f"b{i}.skip.w"] = ops.cast(blk.proj.weight.detach(), self.compute_dtype)
        self._pack, self._pack_key, self._tables = pk, key, {}
        return pk

    @staticmethod
    def _rel_table(rel: Tensor, q_size: int, k_size: int) -> Tensor:
        """[q_size, k_size, D] gathered relative-position table (R/models/mvit.py:330-361); parameter preprocessing,
        cached per (layer, grid) until a parameter changes."""
        max_rel = int(2 * max(q_size, k_size) - 1)
        r = rel.detach()
        if r.shape[0] != max_rel:
            r = F.interpolate(r.t().unsqueeze(0), size=max_rel, mode="linear").squeeze(0).t()
        q_ratio, k_ratio = max(k_size / q_size, 1.0), max(q_size / k_size, 1.0)
        idx = (torch.arange(q_size)[:, None] * q_ratio - torch.arange(k_size)[None, :] * k_ratio) + (k_size - 1) * k_ratio
        return r[idx.long().to(r.device)].contiguous()

    def _onehot_for(self, k_size, dev) -> Tensor:
        """One-hot key columns matching ops.relpos_project's layout for this key grid (ops.relpos_columns: 32 columns when the
        grid fits, else 48); class-token row 0.  Depends on the key grid only (not on any parameter): built once per grid and
        device, kept across optimizer steps."""
        ck = ("onehot", tuple(k_size), str(dev))
        oh = self._const_tables.get(ck)
        if oh is None:
            oh = ops.relpos_onehot(k_size, ops.relpos_columns(k_size), dev)
            self._const_tables[ck] = oh
        return oh

    def _tables_for(self, i: int, q_size, k_size):
        tk = (i, tuple(q_size), tuple(k_size))
        t = self._tables.get(tk)
        if t is None:
            a = self.blocks[i].attn
            Rt = self._rel_table(a.rel_pos_t, q_size[0], k_size[0])
            Rh = self._rel_table(a.rel_pos_h, q_size[1], k_size[1])
            Rw = self._rel_table(a.rel_pos_w, q_size[2], k_size[2])
            t = (Rt, Rh, Rw, self._onehot_for(k_size, Rt.device))
            self._tables[tk] = t
        return t

    # ------------------------------------------------------------------ forward
    def _block(self, i: int, x: Tensor, size, pk) -> Tensor:
        """MultiScaleBlock.forward (R/models/mvit.py:779-802) on tokens [B, 1+T*H*W, C] (fp32, or the 16-bit storage type of
        ``compute_dtype``: then the GEMMs read 16-bit weights and tokens, the pooled q / k / v, the relative-position columns
        and the attention core are fp32, and the attention output is rounded once on its way into the projection)."""
        blk = self.blocks[i]
        a = blk.attn
        B, N, _ = x.shape
        lp = x.dtype != torch.float32
        w = (lambda n, lin: pk[f"b{i}.{n}.w"]) if lp else (lambda n, lin: lin.weight)
        xn = ops.layernorm(x, blk.norm1.weight, blk.norm1.bias, blk.norm1.eps)
        qkv = ops.linear(xn, w("qkv", a.qkv), a.qkv.bias, tag="mvit-gemm").view(B, N, 3, blk.heads, 96)
        q, k, v, q_size, k_size = ops.qkv_pool(
            qkv, (pk[f"b{i}.pool_q"], pk[f"b{i}.pool_k"], pk[f"b{i}.pool_v"]), size, blk.stride_q, blk.stride_kv,
            norms=tuple((n.weight, n.bias, n.eps) for n in (a.norm_q, a.norm_k, a.norm_v)))
        Rt, Rh, Rw, onehot = self._tables_for(i, q_size, k_size)
        extra = ops.relpos_project(q, Rt, Rh, Rw, q_size, k_size, ops.relpos_columns(k_size))
        o = ops.attention_general(q, k, v, scale=96 ** -0.5, q_extra=extra, k_extra=onehot, residual=q, skip_first=True)
        skip = ops.linear(xn, w("skip", blk.proj), blk.proj.bias, tag="mvit-gemm") if hasattr(blk, "proj") else x
        if max(blk.stride_q) > 1:
            ks = tuple(s + 1 if s > 1 else s for s in blk.stride_q)
            skip = ops.cast(ops.maxpool_tokens(ops.cast(skip, torch.float32), size, ks, blk.stride_q), x.dtype)
        x = ops.linear(ops.cast(o, x.dtype), w("aproj", a.proj), a.proj.bias, residual=skip, tag="mvit-gemm")
        y = ops.layernorm(x, blk.norm2.weight, blk.norm2.bias, blk.norm2.eps)
        h = ops.linear(y, w("fc1", blk.mlp.fc1), blk.mlp.fc1.bias, act=ACT_GELU, tag="mvit-gemm")
        return ops.linear(h, w("fc2", blk.mlp.fc2), blk.mlp.fc2.bias, residual=x, tag="mvit-gemm"), q_size

    # ------------------------------------------------------------------ training forward (autograd tape, HIP fwd + bwd)
    def _block_train(self, i: int, x: Tensor, size) -> Tensor:
        from . import autograd_ops as ag
        from . import encoder_autograd as eg

        blk = self.blocks[i]
        a = blk.attn
        B, N, _ = x.shape
        if hasattr(blk, "proj"):      # the skip path starts from the normalised tokens: x has one consumer
            xn = ag.layernorm(x, blk.norm1.weight, blk.norm1.bias, blk.norm1.eps)
        else:                         # x feeds norm1 and the skip path: one node, one backward kernel for both (layernorm_fork)
            x, xn = ag.layernorm_fork(x, blk.norm1.weight, blk.norm1.bias, blk.norm1.eps)
        qkv = ag.linear(xn, a.qkv.weight, a.qkv.bias).view(B, N, 3, blk.heads, 96)
        w27 = [getattr(a, f"pool_{n}").weight.reshape(96, 27) for n in "qkv"]      # the parameter's own layout: views, no copies
        pq, pk_, pv = eg.qkv_pool(qkv, w27[0], w27[1], w27[2], size, blk.stride_q, blk.stride_kv)
        q_size = tuple((s - 1) // st + 1 for s, st in zip(size, blk.stride_q))
        k_size = tuple((s - 1) // st + 1 for s, st in zip(size, blk.stride_kv))
        q, k, v = ag.layernorm3((pq, pk_, pv), (a.norm_q, a.norm_k, a.norm_v))
        # relative-position tables: parameter preprocessing (linear resample + gather) stays on the tape so that the
        # table gradients of relpos_project flow back to rel_pos_t / rel_pos_h / rel_pos_w
        plans = tuple(self._rel_plan(r.shape[0], qs, ks, x.device)
                      for r, qs, ks in zip((a.rel_pos_t, a.rel_pos_h, a.rel_pos_w), q_size, k_size))
        Rt, Rh, Rw = eg.rel_tables(a.rel_pos_t, a.rel_pos_h, a.rel_pos_w, plans)
        onehot = self._onehot_for(k_size, x.device)
        o = eg.relpos_attention(q, k, v, Rt, Rh, Rw, onehot, scale=96 ** -0.5, q_size=q_size, k_size=k_size,
                                E=ops.relpos_columns(k_size))
        skip = ag.linear(xn, blk.proj.weight, blk.proj.bias) if hasattr(blk, "proj") else x
        if max(blk.stride_q) > 1:
            skip = eg.maxpool_tokens(skip, size, tuple(s + 1 if s > 1 else s for s in blk.stride_q), blk.stride_q)
        x = ag.linear(o, a.proj.weight, a.proj.bias, residual=skip)
        x, y = ag.layernorm_fork(x, blk.norm2.weight, blk.norm2.bias, blk.norm2.eps)
        h = ag.linear(y, blk.mlp.fc1.weight, blk.mlp.fc1.bias)       # pre-activation; the GELU belongs to fc2's node (its backward
        return ag.linear(h, blk.mlp.fc2.weight, blk.mlp.fc2.bias, residual=x, in_gelu=True), q_size    # rides in the dgrad epilogue)

    def _rel_plan(self, rel_len: int, q_size: int, k_size: int, dev) -> dict:
        """resize_decomposed_rel_pos (R/models/mvit.py:330-361) of one axis as a sparse row map, built once per (table length,
        grid) on the host: gathered[m] = w2[m,0] * rel[idx2[m,0]] + w2[m,1] * rel[idx2[m,1]] (the two taps of F.interpolate's
        linear resample at the gathered row; weights (1, 0) when the table already has 2*max(q,k)-1 rows), and its transpose in
        CSR form for the backward.  Geometry only -- no parameter enters."""
        ck = ("relplan", rel_len, q_size, k_size, str(dev))
        pl = self._const_tables.get(ck)
        if pl is not None:
            return pl
        max_rel = int(2 * max(q_size, k_size) - 1)
        q_ratio, k_ratio = max(k_size / q_size, 1.0), max(q_size / k_size, 1.0)
        o = ((torch.arange(q_size)[:, None] * q_ratio - torch.arange(k_size)[None, :] * k_ratio) + (k_size - 1) * k_ratio).long().reshape(-1)
        if rel_len == max_rel:
            i0, i1 = o, o
            w0, w1 = torch.ones(o.numel()), torch.zeros(o.numel())
        else:       # upsample_linear1d, align_corners=False: src = len/out * (dst + 0.5) - 0.5, clamped at 0 (float32 as in ATen)
            src = (torch.tensor(rel_len, dtype=torch.float32) / max_rel) * (o.float() + 0.5) - 0.5
            src = src.clamp_min(0.0)
            i0 = src.long()
            i1 = torch.where(i0 < rel_len - 1, i0 + 1, i0)
            w1 = src - i0.float()
            w0 = 1.0 - w1
        M = o.numel()
        idx2 = torch.stack([i0, i1], 1).int()
        w2 = torch.stack([w0, w1], 1).float()
        # transpose: entries (row = table row, col = m, weight), rows ascending, columns ascending inside a row
        rows = torch.cat([i0, i1])
        cols = torch.cat([torch.arange(M), torch.arange(M)])
        wts = torch.cat([w0, w1]).float()
        keep = wts != 0
        rows, cols, wts = rows[keep], cols[keep], wts[keep]
        order = torch.argsort(rows * (2 * M) + cols, stable=True)
        rows, cols, wts = rows[order], cols[order], wts[order]
        ptr = torch.zeros(rel_len + 1, dtype=torch.long)
        ptr[1:] = torch.cumsum(torch.bincount(rows, minlength=rel_len), 0)
        pl = dict(q=q_size, k=k_size, len=rel_len, idx2=idx2.contiguous().to(dev), w2=w2.contiguous().to(dev),
                  csr_ptr=ptr.int().to(dev), csr_col=cols.int().contiguous().to(dev), csr_w=wts.contiguous().to(dev))
        self._const_tables[ck] = pl
        return pl

    def forward_train(self, x: Tensor) -> List[Tensor]:
        from . import autograd_ops as ag
        from . import encoder_autograd as eg

        B = x.shape[0]
        cols, size = ops.im2col3d(x, (3, 7, 7), (2, 4, 4), (1, 3, 3), 448)           # the clip needs no gradient
        L = size[0] * size[1] * size[2]
        w = F.pad(self.patch_embed.projection.weight.reshape(96, -1), (0, 7))
        patches = ag.linear(cols.view(B, L, 448), w, self.patch_embed.projection.bias)
        tok = torch.cat([self.cls_token.expand(B, -1, -1), patches], dim=1)
        outs = []
        for i in range(self.num_layers):
            tok, size = self._block_train(i, tok, size)
            if i in self.stage_of_layer:
                nm = getattr(self, f"norm{self.stage_of_layer[i]}")
                tok = ag.layernorm(tok, nm.weight, nm.bias, nm.eps)
                outs.append(eg.tokens_to_channels_first(tok, 1).view(B, tok.shape[2], *size))
        return outs[::-1]

    def forward(self, x: Tensor, taps: Optional[dict] = None) -> List[Tensor]:
        if not x.is_cuda:
            raise RuntimeError("diff_sal_amd.MViT runs on the GPU only (no CPU fallback); got a CPU tensor")
        if x.dim() == 4:                                         # [B*16, 3, H, W]  (R/models/mvit.py:1091-1092)
            x = x.view(-1, x.shape[-3], 16, x.shape[-2], x.shape[-1])
        if torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()):
            return self.forward_train(x.contiguous().float())
        if x.dim() == 4:                                         # [B*16, 3, H, W]  (R/models/mvit.py:1091-1092)
            x = x.view(-1, x.shape[-3], 16, x.shape[-2], x.shape[-1])
        x = x.contiguous().float()
        B = x.shape[0]
        pk = self.packed()
        cols, size = ops.im2col3d(x, (3, 7, 7), (2, 4, 4), (1, 3, 3), 448)
        L = size[0] * size[1] * size[2]
        tok = torch.empty((B, 1 + L, 96), device=x.device, dtype=torch.float32)
        tok[:, 0] = self.cls_token.detach().view(1, 96)
        for b in range(B):                                       # the patch projection writes rows 1.. of each clip
            ops.conv_igemm(cols[b * L:(b + 1) * L].view(1, 1, L, 448), pk["patch.w"], bias=self.patch_embed.projection.bias,
                           out=tok[b, 1:].view(1, 1, L, 96), tag="mvit-gemm")
        if taps is not None:
            taps["tokens0"] = tok
        tok = ops.cast(tok, self.compute_dtype)                  # the patch embedding itself is an fp32 GEMM (K = 441)
        outs = []
        for i in range(self.num_layers):
            tok, size = self._block(i, tok, size, pk)
            if taps is not None:
                taps[f"block{i}"] = tok
            if i in self.stage_of_layer:
                nm = getattr(self, f"norm{self.stage_of_layer[i]}")
                tok = ops.layernorm(tok, nm.weight, nm.bias, nm.eps)       # replaces x for the next block (mvit.py:1123-1126)
                outs.append(ops.tokens_to_channels_first(ops.cast(tok, torch.float32), 1).view(B, tok.shape[2], *size))
        return outs[::-1]
