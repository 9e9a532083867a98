"""SalUNet -- DiffSal's per-step denoiser, MI355X-native.

Drop-in for the reference module ``models/saliency_decoder/sal_unet.py::SalUNet``
(R/models/saliency_decoder/sal_unet.py:146-328):

  * same constructor keywords (R/cfgs/audio_visual.py:50-82),
  * same ``forward(x, t, feat_list, audio_feat_list=None)`` contract,
  * same ``state_dict`` names and shapes, so reference checkpoints load,

but the forward pass is a sequence of hand-written gfx950 kernels reached through the C ABI of
``libdiffsal_hip.so`` (see ``ops.py`` / ``include/diffsal.h``).  Internally everything is
channels-last: frames [B,T,H,W,C] double as token matrices [B*T*H*W, C], so none of the
reference's ``rearrange(...).contiguous()`` copies exist; the only layout changes are one
transpose of each visual feature map per call (NCTHW is the module's input contract).

Unlike the reference, ``forward`` never mutates ``feat_list`` (reference defect D4).
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Sequence

import torch
from torch import nn

from . import ops
from .ops import ACT_GELU, ACT_NONE, ACT_RELU

Tensor = torch.Tensor


def _holder(**children) -> nn.Module:
    m = nn.Module()
    for k, v in children.items():
        setattr(m, k, v)
    return m


def _seq_named(pairs) -> nn.Sequential:
    from collections import OrderedDict

    return nn.Sequential(OrderedDict(pairs))


class _ParamTree:
    """Builders for the parameter containers.  The nn layers below are *storage only* (names,
    shapes, init, state_dict, .to(), optimizers); their own forward() is never called."""

    @staticmethod
    def res_block(cin: int, cout: int, temb_ch: int) -> nn.Module:
        m = nn.Module()
        m.norm1 = nn.GroupNorm(32, cin, eps=1e-6)
        m.conv1 = nn.Conv2d(cin, cout, 3, padding=1)
        m.temb_proj = nn.Linear(temb_ch, cout)
        m.norm2 = nn.GroupNorm(32, cout, eps=1e-6)
        m.conv2 = nn.Conv2d(cout, cout, 3, padding=1)
        if cin != cout:
            m.nin_shortcut = nn.Conv2d(cin, cout, 1)
        return m

    @staticmethod
    def up_embed(cin: int, cout: int, dil: int) -> nn.Module:
        # indices 1,2,4,5 carry parameters (0 = upsample, 3/6 = ReLU): common_block.py:196-216
        return _holder(proj=nn.Sequential(
            nn.Identity(),
            nn.Conv2d(cin, cout, 3, padding=dil, dilation=dil, bias=False), nn.BatchNorm2d(cout), nn.Identity(),
            nn.Conv2d(cout, cout, 3, padding=dil, dilation=dil, bias=False), nn.BatchNorm2d(cout), nn.Identity()))

    @staticmethod
    def dw_proj(c: int, k3) -> nn.Sequential:
        return _seq_named([("conv", nn.Conv3d(c, c, k3, groups=c, bias=False)), ("bn", nn.LayerNorm(c))])

    @classmethod
    def block(cls, c: int, mlp_ratio: float, kq: int, kkv: int, qkv_bias: bool) -> nn.Module:
        hid = int(c * mlp_ratio)
        attn = _holder(
            conv_proj_q=cls.dw_proj(c, (kq, kq, kq)),
            conv_proj_k=cls.dw_proj(c, (1, kkv, kkv)),
            conv_proj_v=cls.dw_proj(c, (1, kkv, kkv)),
            proj_q=nn.Linear(c, c, bias=qkv_bias), proj_k=nn.Linear(c, c, bias=qkv_bias),
            proj_v=nn.Linear(c, c, bias=qkv_bias), proj=nn.Linear(c, c))
        return _holder(mlp=_holder(fc1=nn.Linear(c, hid), fc2=nn.Linear(hid, c)), norm=nn.LayerNorm(c), attn=attn,
                       norm2=nn.LayerNorm(c), align_conv=nn.Conv2d(512, c, 1))


class _ShapeOnly:
    """What ops.wino4_supported reads of a tensor (shape, fp32): the planner is asked about a map that does not exist yet."""
    dtype = torch.float32

    def __init__(self, shape):
        self.shape = tuple(shape)


class SalUNet(nn.Module):
    """See module docstring.  Keyword names follow R/models/saliency_decoder/sal_unet.py:147-179."""

    def __init__(
        self,
        image_based=False,
        img_size=(224, 384),
        frames_len=2,
        tasks=("futr",),
        in_index=(0, 1, 2, 3),
        idx_to_planes=None,
        temporal_size=5,
        mid_num_stages=3,
        futr_num_stages=1,
        ori_embed_dim=768,
        down_embed_dim=96,
        keep_max_len=5,
        exclude_layers=(),
        temporal_list=(1, 9, 9),
        patch_size=(0, 3, 3),
        patch_stride=(0, 1, 1),
        patch_padding=(0, 2, 2),
        up_channel=(768, 384, 192),
        num_heads=(2, 2, 2),
        mlp_ratio=(4.0, 4.0, 4.0),
        drop_path_rate=(0.15, 0.15, 0.15),
        qkv_bias=(True, True, True),
        kv_proj_method=("avg", "avg", "avg"),
        kernel_kv=(2, 4, 8),
        padding_kv=(0, 0, 0),
        stride_kv=(2, 4, 8),
        q_proj_method=("dw_bn", "dw_bn", "dw_bn"),
        kernel_q=(3, 3, 3),
        padding_q=(1, 1, 1),
        stride_q=(1, 1, 1),
        compute_dtype=torch.float32,
        gemm_precision=None,
    ):
        """The keywords up to ``stride_q`` are the reference's (R/models/saliency_decoder/sal_unet.py:147-179).  Two
        extras select the reduced-precision inference datapaths (never the default; parameters stay fp32):
        ``compute_dtype`` = torch.bfloat16 / torch.float16 stores activations and packed weights in that type in HBM
        (native 16-bit MFMA, fp32 accumulation and statistics: BASELINE configs[1] / configs[4]);
        ``gemm_precision`` = "bf16x3" keeps fp32 storage and runs the GEMMs in split-bf16 arithmetic."""
        super().__init__()
        idx_to_planes = dict(idx_to_planes or {0: 96, 1: 192, 2: 384, 3: 768})
        ns = int(mid_num_stages)
        if int(frames_len) != 1:
            # the reference derives len(tasks) from str(range(frames_len))[0], i.e. always one task (quirk Q1)
            pass
        for name, seq in (("patch_size", patch_size), ("patch_padding", patch_padding), ("up_channel", up_channel),
                          ("num_heads", num_heads), ("mlp_ratio", mlp_ratio), ("kernel_kv", kernel_kv),
                          ("stride_kv", stride_kv), ("temporal_list", temporal_list), ("qkv_bias", qkv_bias),
                          ("kernel_q", kernel_q), ("padding_q", padding_q), ("stride_q", stride_q),
                          ("padding_kv", padding_kv)):
            if len(seq) < ns:
                raise ValueError(f"{name} needs {ns} entries, got {len(seq)}")
        if any(k != 3 for k in kernel_q[:ns]) or any(p != 1 for p in padding_q[:ns]) or any(s != 1 for s in stride_q[:ns]):
            raise NotImplementedError("q projection: only depthwise 3x3x3 / pad 1 / stride 1 is built")
        if list(kernel_kv[:ns]) != list(stride_kv[:ns]) or any(p != 0 for p in padding_kv[:ns]):
            raise NotImplementedError("k/v projection: only kernel == stride, padding 0 is built")
        if any(ps not in (0, 3) for ps in patch_size[:ns]) or any(s not in (0, 1) for s in patch_stride[:ns]):
            raise NotImplementedError("UpEmbed: only 3x3 stride-1 dilated convolutions are built")

        self.img_size = (int(img_size[0]), int(img_size[1]))
        self.image_based = bool(image_based)
        self.frame_len = frames_len
        self.num_stages = ns
        self.up_channels = [int(c) for c in up_channel[:ns]]
        self.heads = [int(h) for h in num_heads[:ns]]
        self.kernel_kv = [int(k) for k in kernel_kv[:ns]]
        self.temporal_list = [int(k) for k in temporal_list[:ns]]
        self.dilation = [int(p) if int(ps) != 0 else 0 for p, ps in zip(patch_padding[:ns], patch_size[:ns])]
        self.ori_embed_dim = int(ori_embed_dim)
        self.down_channel = int(idx_to_planes[0])
        self.ch = 96  # sal_unet.py:228
        self.temb_ch = self.ch * 4
        if compute_dtype not in (torch.float32, torch.bfloat16, torch.float16):
            raise ValueError(f"compute_dtype {compute_dtype}: float32, bfloat16 or float16")
        self.compute_dtype = compute_dtype
        self.gemm_precision = gemm_precision

        # ---- decoder parameters (names: SURVEY Appendix B) ----
        dec = nn.Module()
        dec.norm_mts = nn.ModuleList()
        dec.redu_chan_up = nn.ModuleList()
        dec.mid_stages = nn.ModuleList()
        prev = self.ori_embed_dim
        for i, c in enumerate(self.up_channels):
            st = nn.Module()
            st.patch_embed = (nn.ModuleList([_ParamTree.up_embed(prev, c, self.dilation[i])])
                              if self.dilation[i] != 0 else None)
            st.blocks = nn.ModuleList([_ParamTree.block(c, float(mlp_ratio[i]), int(kernel_q[i]),
                                                        self.kernel_kv[i], bool(qkv_bias[i]))])
            dec.mid_stages.append(st)
            dec.norm_mts.append(nn.LayerNorm(c))
            kt = self.temporal_list[i]
            dec.redu_chan_up.append(_holder(proj=nn.Sequential(
                nn.Conv3d(c, self.ori_embed_dim, (kt, 1, 1), stride=(kt, 1, 1), bias=False), nn.Identity())))
            prev = c
        dec.mt_proj = nn.Sequential(nn.Conv2d(self.ori_embed_dim, int(down_embed_dim), 3, padding=1),
                                    nn.BatchNorm2d(int(down_embed_dim)), nn.Identity())
        self.invpt_decoder = dec
        self.logits = _holder(linear_pred=nn.Conv2d(self.down_channel, 1, 1))

        # ---- noise encoder parameters ----
        self.temb = _holder(dense=nn.ModuleList([nn.Linear(self.ch, self.temb_ch),
                                                 nn.Linear(self.temb_ch, self.temb_ch)]))
        self.conv_in = nn.Conv2d(1, self.ch, 3, padding=1)
        self.down1 = _holder(conv=nn.Conv2d(self.ch, self.ch, 3, stride=4))
        self.res_encoder = nn.ModuleList()
        cin = self.ch
        for cout in self.up_channels[:-1][::-1]:
            self.res_encoder.append(nn.Sequential(_ParamTree.res_block(cin, cout, self.temb_ch),
                                                  _holder(conv=nn.Conv2d(cout, cout, 3, stride=2))))
            cin = cout

        self._pack_cache: Optional[Dict[str, Tensor]] = None
        self._pack_key = None
        self._pack_epoch = 0
        self.init_weights()

    # ------------------------------------------------------------------ init (quirk Q15)
    def init_weights(self):
        for m in self.modules():
            if isinstance(m, (nn.Conv2d, nn.Conv3d, nn.Linear)):
                nn.init.normal_(m.weight, 0.0, 0.01)
                if m.bias is not None:
                    nn.init.zeros_(m.bias)
            elif isinstance(m, (nn.LayerNorm, nn.BatchNorm2d, nn.GroupNorm)):
                nn.init.ones_(m.weight)
                nn.init.zeros_(m.bias)
        for st in self.invpt_decoder.mid_stages:  # transformer.py:236-248 re-initialises the stage's Linears
            for m in st.modules():
                if isinstance(m, nn.Linear):
                    nn.init.trunc_normal_(m.weight, std=0.02)
        nn.init.trunc_normal_(self.invpt_decoder.mt_proj[0].weight, std=0.02)  # sal_unet.py:408

    # ------------------------------------------------------------------ weight packing
    def _precision(self) -> str:
        return self.gemm_precision or ops.get_gemm_precision()

    def _cache_key(self):
        return (self._pack_epoch, self._precision(), self.compute_dtype, self.winograd) + tuple(
            (p.data_ptr(), p._version) for p in list(self.parameters()) + list(self.buffers()))

    def pack_epoch(self):
        """Changes whenever the kernel-layout weights would be rebuilt (HIP-graph caches key on it)."""
        return self._cache_key()

    def parameters_updated(self) -> None:
        """Tell the module its parameters were rewritten behind autograd's version counters (the fused Adam kernel
        writes through raw pointers): the eval-mode packed-weight cache is rebuilt on next use."""
        self._pack_epoch += 1

    def _pack_conv(self, w: Tensor) -> Tensor:
        """[Cout,Cin,KH,KW] -> [Cout, K] in the implicit-GEMM k order (channel chunk, tap, channel); cast to the 16-bit
        storage type of a reduced-precision module, or (fp32 storage, opt-in bf16x3 GEMM mode) pre-split into bf16
        hi/lo halves so the kernel does not convert its B operand."""
        wp = ops.pack_conv_weight(w)
        if self.compute_dtype != torch.float32:
            return ops.cast(wp, self.compute_dtype)
        return ops.split_weight(wp) if self._precision() == "bf16x3" else wp

    def _pack_wino(self, w: Tensor, f2: bool = True):
        """Winograd forms of a 3x3 weight for the exact-fp32 path (F(2x2): ops.pack_wino_weight, F(4x4): ops.pack_wino4_weight), else
        None: the library's planner then decides per shape which of them -- if any -- runs (ops.conv_igemm(wino=...)).  ``f2=False``:
        only the F(4x4) form (a layer that can only use that one does not keep 16/9 of its weight bytes for nothing)."""
        if self.compute_dtype != torch.float32 or self._precision() != "fp32" or not self.winograd:
            return None
        return ops.WinoWeights(w, f2=f2)

    def _tap_weight(self, w: Tensor) -> Tensor:
        """Conv2d 3x3 weight -> the [9*Cout, Cin] matrix of its nine 1x1 tap mixings (row = tap * Cout + co), in the storage /
        split format of this module's GEMM weights (see ops.tapsum)."""
        co, ci = w.shape[:2]
        return self._pack_conv(w.detach().permute(2, 3, 0, 1).reshape(9 * co, ci, 1, 1).contiguous())

    def _use_tap_conv(self, taps, which: str) -> bool:
        # fp32 storage: the 3x3 convolutions are bound by the matrix rate, so 3-4x fewer FLOPs is 3x less time -- every use.
        # 16-bit storage: only the uses listed in tap_conv16 (where GEMM + gather beat up-sampling + convolution).
        if not self.tap_conv or (taps is not None and self.taps_reference_forms):
            return False
        if (which == "mt" and self.compute_dtype != torch.float32 and self.mt_tap16_f32 and self.fold_head
                and self.ori_embed_dim % 192 == 0):       # diffsal_linear_f32out: K a multiple of 192
            return True
        return self.compute_dtype == torch.float32 or which in self.tap_conv16

    def _gemm_w(self, w: Tensor) -> Tensor:
        """Linear / 1x1 weight [N, K] as the GEMM reads it (K % 32 == 0: the packed k order is the identity)."""
        return w.detach() if self.compute_dtype == torch.float32 else ops.cast(w.detach().contiguous(), self.compute_dtype)

    @staticmethod
    def _bn_affine(bn: nn.BatchNorm2d):
        scale = (bn.weight.detach() / torch.sqrt(bn.running_var + bn.eps)).contiguous()
        shift = (bn.bias.detach() - bn.running_mean * scale).contiguous()
        return scale, shift

    def packed(self) -> Dict[str, Tensor]:
        """Kernel-layout copies of the parameters, rebuilt only when a parameter changes."""
        key = self._cache_key()
        if self._pack_cache is not None and key == self._pack_key:
            return self._pack_cache
        dev = self.conv_in.weight.device
        pk: Dict[str, Tensor] = {}
        half = self.ch // 2
        # same fp32 arithmetic as the reference table, sal_unet.py:25-27
        pk["freq"] = torch.exp(torch.arange(half, dtype=torch.float32) * -(math.log(10000) / (half - 1))).to(dev)
        pk["conv_in.w"] = self.conv_in.weight.detach().reshape(self.ch, 9).contiguous()
        pk["down1.w"] = self._pack_conv(self.down1.conv.weight)
        pk["in_s4.w"], pk["in_s4.b"] = ops.compose_conv_in_s4(self.conv_in.weight, self.conv_in.bias, self.down1.conv.weight,
                                                               self.down1.conv.bias)
        tw, tb = [], []
        for i, blk in enumerate(self.res_encoder):
            rb, dn = blk[0], blk[1]
            pk[f"res{i}.conv1.w"] = self._pack_conv(rb.conv1.weight)
            pk[f"res{i}.conv2.w"] = self._pack_conv(rb.conv2.weight)
            pk[f"res{i}.conv1.wino"] = self._pack_wino(rb.conv1.weight)
            pk[f"res{i}.conv2.wino"] = self._pack_wino(rb.conv2.weight)
            if hasattr(rb, "nin_shortcut"):
                pk[f"res{i}.nin.w"] = self._pack_conv(rb.nin_shortcut.weight)
                # fused block: the shortcut product rides in conv1's launch without its bias, which conv2's epilogue adds
                pk[f"res{i}.bias2"] = (rb.conv2.bias.detach() + rb.nin_shortcut.bias.detach()).contiguous()
            pk[f"res{i}.down.w"] = self._pack_conv(dn.conv.weight)
            tw.append(rb.temb_proj.weight.detach())
            tb.append(rb.temb_proj.bias.detach())
        pk["tproj.w"] = torch.cat(tw, 0).contiguous()
        pk["tproj.b"] = torch.cat(tb, 0).contiguous()
        dec = self.invpt_decoder
        for i, st in enumerate(dec.mid_stages):
            if st.patch_embed is not None:
                pe = st.patch_embed[0].proj
                pk[f"s{i}.pe1.w"] = self._pack_conv(pe[1].weight)
                pk[f"s{i}.pe1.tapw"] = self._tap_weight(pe[1].weight)
                # UpEmbed's first convolution uses a Winograd form only at the source resolution (extended grid: F(4x4) only) and only
                # where the source map is at least 12 x 12 (_forward_eval): stage i reads a map of img / 32 * 2^(i-1)
                hs, ws_ = (self.img_size[0] // 32) << (i - 1), (self.img_size[1] // 32) << (i - 1)
                pk[f"s{i}.pe1.wino"] = self._pack_wino(pe[1].weight, f2=False) if min(hs, ws_) >= 12 else None
                pk[f"s{i}.pe1.scale"], pk[f"s{i}.pe1.shift"] = self._bn_affine(pe[2])
                pk[f"s{i}.pe2.w"] = self._pack_conv(pe[4].weight)
                pk[f"s{i}.pe2.wino"] = self._pack_wino(pe[4].weight)
                pk[f"s{i}.pe2.scale"], pk[f"s{i}.pe2.shift"] = self._bn_affine(pe[5])
            a = st.blocks[0].attn
            c = self.up_channels[i]
            k = self.kernel_kv[i]
            pk[f"s{i}.wq9"] = a.conv_proj_q.conv.weight.detach()[:, 0, 1].reshape(c, 9).t().contiguous()  # Q8
            pk[f"s{i}.wk"] = a.conv_proj_k.conv.weight.detach().reshape(c, k * k).t().contiguous()
            pk[f"s{i}.wv"] = a.conv_proj_v.conv.weight.detach().reshape(c, k * k).t().contiguous()
            pk[f"s{i}.align.w"] = self._gemm_w(st.blocks[0].align_conv.weight.detach().reshape(c, 512).contiguous())
            blk = st.blocks[0]
            for nm, lin in (("q", a.proj_q), ("k", a.proj_k), ("v", a.proj_v), ("proj", a.proj), ("fc1", blk.mlp.fc1),
                            ("fc2", blk.mlp.fc2)):
                pk[f"s{i}.{nm}.w"] = self._gemm_w(lin.weight)
            pk[f"s{i}.redu.w"] = self._pack_conv(dec.redu_chan_up[i].proj[0].weight)  # [Co, C, kt, 1, 1]
        if all(f"s{i}.align.w" in pk for i in range(self.num_stages)) and self.num_stages > 1:
            pk["align_all.w"] = torch.cat([pk[f"s{i}.align.w"] for i in range(self.num_stages)], 0).contiguous()
            pk["align_all.b"] = torch.cat([dec.mid_stages[i].blocks[0].align_conv.bias.detach().float()
                                           for i in range(self.num_stages)], 0).contiguous()
            offs, o = [], 0
            for i in range(self.num_stages):
                offs.append(o)
                o += self.up_channels[i]
            pk["align_all.offs"] = offs
        pk["mt.w"] = self._pack_conv(dec.mt_proj[0].weight)
        pk["mt.tapw"] = self._tap_weight(dec.mt_proj[0].weight)
        pk["mt.scale"], pk["mt.shift"] = self._bn_affine(dec.mt_proj[1])
        pk["head.w"] = self.logits.linear_pred.weight.detach().reshape(-1).contiguous()
        self._pack_cache, self._pack_key = pk, key
        return pk

    # ------------------------------------------------------------------ forward pieces
    def _noise_encoder(self, x: Tensor, t: Tensor, pk, taps=None) -> List[Tensor]:
        """K1-K5 (sal_unet.py:279-307): returns NHWC noise maps, coarsest first."""
        d0, d1 = self.temb.dense[0], self.temb.dense[1]
        temb = ops.temb_mlp(t, pk["freq"], d0.weight, d0.bias, d1.weight, d1.bias)
        tproj = ops.dense_small(temb, pk["tproj.w"], pk["tproj.b"], swish_in=True)
        B, _, H, W = x.shape
        if H % 4 == 0 and W % 4 == 0:
            # conv_in and the stride-4 Downsample have nothing between them: one composed 5x5 stride-4 convolution (exact for
            # these sizes; see diffsal_conv_in_s4) instead of a [B,H,W,96] intermediate and a 3.6 GFLOP GEMM
            f = ops.conv_in_s4(x, pk["in_s4.w"], pk["in_s4.b"], out_dtype=self.compute_dtype)
        else:
            f = ops.conv_in(x, pk["conv_in.w"], self.conv_in.bias, skip_mod=4, out_dtype=self.compute_dtype)
            f = ops.conv_igemm(f, pk["down1.w"], kh=3, kw=3, stride=(4, 4), out_hw=((H - 2) // 4 + 1, (W - 2) // 4 + 1),
                               bias=self.down1.conv.bias, tag="K2")
        if taps is not None:
            taps["temb"], taps["down1"] = temb, f
        outs, off = [], 0
        for i, blk in enumerate(self.res_encoder):
            rb, dn = blk[0], blk[1]
            co = rb.conv1.out_channels
            w1, w2 = pk[f"res{i}.conv1.wino"], pk[f"res{i}.conv2.wino"]
            plan = None
            if self.fuse_resblock and w1 is not None and w2 is not None and w1.f4 is not None and w2.f4 is not None:
                plan = ops.resblock_wino4_plan(f, co, 32)
            if plan is not None:
                # both convolutions on the F(4x4) path: GroupNorm + swish are applied as the input transforms load their tensors
                # (statistics only: the normalised maps never exist), the 1x1 shortcut is computed by the launch of conv1's
                # position products, and conv1's output transform leaves norm2's statistics
                ab1 = ops.gn_affine(f, rb.norm1.weight, rb.norm1.bias, 32, rb.norm1.eps)
                has_nin = hasattr(rb, "nin_shortcut")
                side = (f, pk[f"res{i}.nin.w"]) if has_nin and plan["side"] else None
                h, sc, st = ops.conv3x3_wino4_ex(f, w1.f4, bias=rb.conv1.bias, rowvec=tproj[:, off:off + co], gn_ab=ab1, side=side,
                                                 stats_groups=32 if plan["stats"] else 0)
                off += co
                if st is not None:
                    ab2 = ops.gn_affine_from_stats(st, rb.norm2.weight, rb.norm2.bias, rb.norm2.eps)
                else:
                    ab2 = ops.gn_affine(h, rb.norm2.weight, rb.norm2.bias, 32, rb.norm2.eps)
                bias2 = rb.conv2.bias
                if not has_nin:
                    sc = f
                elif sc is None:
                    sc = ops.conv_igemm(f, pk[f"res{i}.nin.w"], bias=rb.nin_shortcut.bias, tag="K4")
                else:
                    bias2 = pk[f"res{i}.bias2"]
                f, _, _ = ops.conv3x3_wino4_ex(h, w2.f4, bias=bias2, residual=sc, gn_ab=ab2)
                if taps is not None:
                    taps[f"res{i}"] = f
                hh, ww = f.shape[1:3]
                f = ops.conv_igemm(f, pk[f"res{i}.down.w"], kh=3, kw=3, stride=(2, 2),
                                   out_hw=((hh - 2) // 2 + 1, (ww - 2) // 2 + 1), bias=dn.conv.bias, tag="K5")
                outs.append(f)
                continue
            h = ops.groupnorm_swish(f, rb.norm1.weight, rb.norm1.bias, 32, rb.norm1.eps)
            h = ops.conv_igemm(h, pk[f"res{i}.conv1.w"], kh=3, kw=3, pad=(1, 1), bias=rb.conv1.bias,
                               rowvec=tproj[:, off:off + co], tag="K4", wino=pk[f"res{i}.conv1.wino"])
            off += co
            h = ops.groupnorm_swish(h, rb.norm2.weight, rb.norm2.bias, 32, rb.norm2.eps)
            sc = f
            if hasattr(rb, "nin_shortcut"):
                sc = ops.conv_igemm(f, pk[f"res{i}.nin.w"], bias=rb.nin_shortcut.bias, tag="K4")
            f = ops.conv_igemm(h, pk[f"res{i}.conv2.w"], kh=3, kw=3, pad=(1, 1), bias=rb.conv2.bias, residual=sc, tag="K4",
                               wino=pk[f"res{i}.conv2.wino"])
            if taps is not None:
                taps[f"res{i}"] = f
            hh, ww = f.shape[1:3]
            f = ops.conv_igemm(f, pk[f"res{i}.down.w"], kh=3, kw=3, stride=(2, 2),
                               out_hw=((hh - 2) // 2 + 1, (ww - 2) // 2 + 1), bias=dn.conv.bias, tag="K5")
            outs.append(f)
        return outs[::-1]

    def _block(self, i: int, x: Tensor, pk, audio_tok: Optional[Tensor], audio_hw, norm_z=None):
        """TransformerBlock on frames x [B,T,H,W,C] (transformer.py:124-159, attention.py:86-113) -> (x_out, z) where
        z = norm_z(x_out) on the frames ReduceTemp reads if the fused C = 96 MLP kernel produced it, else None."""
        B, T, H, W, C = x.shape
        blk = self.invpt_decoder.mid_stages[i].blocks[0]
        a = blk.attn
        n9 = B * T
        k_src = None                       # None: the key branch reads the same normalised frames as q and v
        if audio_tok is not None:
            if isinstance(audio_tok, tuple):     # (all stages' align products side by side, channel offsets): this stage's slice
                a_all, offs = audio_tok
                a_small = a_all[:, :, offs[i]:offs[i] + C]
            else:
                a_small = ops.linear(audio_tok, pk[f"s{i}.align.w"], blk.align_conv.bias, tag="K7-align")  # [B*T, ha*wa, C]
            k_src = ops.audio_fuse(a_small, x, audio_hw[0], audio_hw[1])  # [B,C,T,H,W], read back as tokens (Q5)
        xt = x.view(n9, H * W, C)
        gh, gw = (H - self.kernel_kv[i]) // self.kernel_kv[i] + 1, (W - self.kernel_kv[i]) // self.kernel_kv[i] + 1
        if (self.fused_front and ops.block_front_supported(C, self.heads[i], gh * gw, x.dtype)
                and (x.dtype != torch.float32 or self._precision() == "fp32")
                and not getattr(pk[f"s{i}.k.w"], "_diffsal_split", False)):
            # [pooled k / v with the folded first LayerNorm] [their projections, one paired launch] [LayerNorm -> depthwise q -> LayerNorm -> proj_q ->
            # attention (-> proj + residual on fp32)]: x_n, q_in, q and (fp32) o never reach HBM.  C = 96 continues with the
            # fused second half (mlp_block / block16: 4 launches for the whole block, 3 with fold_kv_proj), C = 192 with the per-operator launches
            xv_ = x.view(n9, H, W, C)
            xk_ = xv_ if k_src is None else k_src.view(n9, H, W, C)
            kvn = (a.conv_proj_k.bn.weight, a.conv_proj_k.bn.bias, a.conv_proj_v.bn.weight, a.conv_proj_v.bn.bias)
            pre = (blk.norm.weight, blk.norm.bias, blk.norm.eps, k_src is None)
            if self.fold_kv_proj and C == 96:
                kk, vv = ops.kv_prep_proj(xk_, xv_, pk[f"s{i}.wk"], pk[f"s{i}.wv"], *kvn, self.kernel_kv[i], a.conv_proj_k.bn.eps, pre,
                                          (pk[f"s{i}.k.w"], a.proj_k.bias), (pk[f"s{i}.v.w"], a.proj_v.bias))
            else:
                kk, vv = ops.kv_prep(xk_, xv_, pk[f"s{i}.wk"], pk[f"s{i}.wv"], *kvn, self.kernel_kv[i], a.conv_proj_k.bn.eps, pre_ln=pre)
                kk, vv = ops.linear_pair(kk, vv, pk[f"s{i}.k.w"], pk[f"s{i}.v.w"], a.proj_k.bias, a.proj_v.bias)
            f32 = x.dtype == torch.float32
            y = ops.block_front(xv_, kk, vv, (blk.norm.weight, blk.norm.bias, blk.norm.eps), pk[f"s{i}.wq9"],
                                (a.conv_proj_q.bn.weight, a.conv_proj_q.bn.bias, a.conv_proj_q.bn.eps),
                                (pk[f"s{i}.q.w"], a.proj_q.bias), (pk[f"s{i}.proj.w"], a.proj.bias) if f32 else None,
                                self.heads[i], float(C) ** -0.5).view(n9, H * W, C)
            nz = None if norm_z is None else (norm_z.weight, norm_z.bias, norm_z.eps)
            if C == 96 and blk.mlp.fc1.out_features == 192:
                if f32:
                    x2, z = ops.mlp_block(y, (blk.norm2.weight, blk.norm2.bias, blk.norm2.eps), (pk[f"s{i}.fc1.w"], blk.mlp.fc1.bias),
                                          (pk[f"s{i}.fc2.w"], blk.mlp.fc2.bias), nz, (H * W, T, self.temporal_list[i]))
                else:
                    x2, z = ops.block16(y, xt, (pk[f"s{i}.proj.w"], a.proj.bias), (blk.norm2.weight, blk.norm2.bias, blk.norm2.eps),
                                        (pk[f"s{i}.fc1.w"], blk.mlp.fc1.bias), (pk[f"s{i}.fc2.w"], blk.mlp.fc2.bias), nz,
                                        (H * W, T, self.temporal_list[i]))
                return x2.view(B, T, H, W, C), (None if z is None else z.view(B, T, H, W, C))
            x1 = y if f32 else ops.linear(y, pk[f"s{i}.proj.w"], a.proj.bias, residual=xt)
            y2 = ops.layernorm(x1, blk.norm2.weight, blk.norm2.bias, blk.norm2.eps)
            y2 = ops.linear(y2, pk[f"s{i}.fc1.w"], blk.mlp.fc1.bias, act=ACT_GELU)
            x2 = ops.linear(y2, pk[f"s{i}.fc2.w"], blk.mlp.fc2.bias, residual=x1)
            return x2.view(B, T, H, W, C), None
        if self.merge_qkv_prep and self.fold_norm1 and a.conv_proj_q.bn.eps == a.conv_proj_k.bn.eps:
            # the block's `norm` is applied to the tokens as qkv_prep loads them: x_n = norm(x) is never written
            xv_ = x.view(n9, H, W, C)
            q, kk, vv = ops.qkv_prep(xv_, pk[f"s{i}.wq9"], a.conv_proj_q.bn.weight, a.conv_proj_q.bn.bias,
                                     xv_ if k_src is None else k_src.view(n9, H, W, C), xv_, pk[f"s{i}.wk"], pk[f"s{i}.wv"],
                                     a.conv_proj_k.bn.weight, a.conv_proj_k.bn.bias, a.conv_proj_v.bn.weight,
                                     a.conv_proj_v.bn.bias, self.kernel_kv[i], a.conv_proj_k.bn.eps,
                                     pre_ln=(blk.norm.weight, blk.norm.bias, blk.norm.eps, k_src is None))
        else:
            xn = ops.layernorm(x, blk.norm.weight, blk.norm.bias, blk.norm.eps)
            if k_src is None:
                k_src = xn
            if self.merge_qkv_prep and a.conv_proj_q.bn.eps == a.conv_proj_k.bn.eps and k_src.dtype == xn.dtype:
                q, kk, vv = ops.qkv_prep(xn.view(n9, H, W, C), pk[f"s{i}.wq9"], a.conv_proj_q.bn.weight, a.conv_proj_q.bn.bias,
                                         k_src.view(n9, H, W, C), xn.view(n9, H, W, C), pk[f"s{i}.wk"], pk[f"s{i}.wv"],
                                         a.conv_proj_k.bn.weight, a.conv_proj_k.bn.bias, a.conv_proj_v.bn.weight,
                                         a.conv_proj_v.bn.bias, self.kernel_kv[i], a.conv_proj_k.bn.eps)
            else:
                q = ops.dwconv3_ln(xn.view(n9, H, W, C), pk[f"s{i}.wq9"], a.conv_proj_q.bn.weight, a.conv_proj_q.bn.bias,
                                   a.conv_proj_q.bn.eps)
                kk, vv = ops.dwpool_ln_kv(k_src.view(n9, H, W, C), xn.view(n9, H, W, C), pk[f"s{i}.wk"], pk[f"s{i}.wv"],
                                          a.conv_proj_k.bn.weight, a.conv_proj_k.bn.bias, a.conv_proj_v.bn.weight,
                                          a.conv_proj_v.bn.bias, self.kernel_kv[i], a.conv_proj_k.bn.eps)
        if (self.group_qkv and self.compute_dtype == torch.float32 and ops.get_gemm_precision() == "fp32"
                and not getattr(pk[f"s{i}.k.w"], "_diffsal_split", False) and C % 96 == 0
                and (self.group_qkv_all or -(-(q.numel() // C) // 96) * (C // 96) <= 256)):
            # the three projections of the block in ONE launch when the query product alone leaves workgroup slots free (<= 256
            # tiles of 96 x 96: stage 0 at B = 4; the key / value products have 648 rows and would fill a quarter of the chip
            # for the length of a whole K walk).  Measured at B = 4: stage 0 64 us against 40 + 36; stages 1 / 2, whose query
            # products fill the chip, 60 / 65 us against 38 + 19 / 41 + 15 -- not grouped.
            q, kk, vv = ops.linear_group((q, kk, vv), (pk[f"s{i}.q.w"], pk[f"s{i}.k.w"], pk[f"s{i}.v.w"]),
                                         (a.proj_q.bias, a.proj_k.bias, a.proj_v.bias))
        elif self.pair_kv and not getattr(pk[f"s{i}.k.w"], "_diffsal_split", False) and (
                self.compute_dtype != torch.float32 or ops.get_gemm_precision() == "fp32"):
            q = ops.linear(q, pk[f"s{i}.q.w"], a.proj_q.bias)
            kk, vv = ops.linear_pair(kk, vv, pk[f"s{i}.k.w"], pk[f"s{i}.v.w"], a.proj_k.bias, a.proj_v.bias)   # one launch
        else:
            q = ops.linear(q, pk[f"s{i}.q.w"], a.proj_q.bias)
            kk = ops.linear(kk, pk[f"s{i}.k.w"], a.proj_k.bias)
            vv = ops.linear(vv, pk[f"s{i}.v.w"], a.proj_v.bias)
        o = ops.attention(q, kk, vv, self.heads[i], float(C) ** -0.5)  # scale uses full C (Q6)
        if C == 96 and blk.mlp.fc1.out_features == 192 and x.dtype != torch.float32:
            # finest stage on 16-bit storage: proj + residual + norm2 + MLP + residual + norm_mts in ONE launch
            x2, z = ops.block16(o, xt, (pk[f"s{i}.proj.w"], a.proj.bias), (blk.norm2.weight, blk.norm2.bias, blk.norm2.eps),
                                (pk[f"s{i}.fc1.w"], blk.mlp.fc1.bias), (pk[f"s{i}.fc2.w"], blk.mlp.fc2.bias),
                                None if norm_z is None else (norm_z.weight, norm_z.bias, norm_z.eps),
                                (H * W, T, self.temporal_list[i]))
            return x2.view(B, T, H, W, C), (None if z is None else z.view(B, T, H, W, C))
        x1 = ops.linear(o, pk[f"s{i}.proj.w"], a.proj.bias, residual=xt)
        if C == 96 and blk.mlp.fc1.out_features == 192 and x1.dtype == torch.float32 and self._precision() == "fp32":
            # finest stage: norm2 -> fc1 -> GELU -> fc2 -> +x1 -> norm_mts in ONE launch (both MLP weights live in LDS)
            x2, z = ops.mlp_block(x1, (blk.norm2.weight, blk.norm2.bias, blk.norm2.eps), (pk[f"s{i}.fc1.w"], blk.mlp.fc1.bias),
                                  (pk[f"s{i}.fc2.w"], blk.mlp.fc2.bias),
                                  None if norm_z is None else (norm_z.weight, norm_z.bias, norm_z.eps),
                                  (H * W, T, self.temporal_list[i]))
            return x2.view(B, T, H, W, C), (None if z is None else z.view(B, T, H, W, C))
        y = ops.layernorm(x1, blk.norm2.weight, blk.norm2.bias, blk.norm2.eps)
        y = ops.linear(y, pk[f"s{i}.fc1.w"], blk.mlp.fc1.bias, act=ACT_GELU)
        x2 = ops.linear(y, pk[f"s{i}.fc2.w"], blk.mlp.fc2.bias, residual=x1)
        return x2.view(B, T, H, W, C), None

    # Largest batch evaluated in one pass.  The implicit-GEMM loader addresses each operand with 32-bit BYTE offsets (< 4 GiB) and
    # the row kernels count elements in 32-bit ints, so a pass is sized such that the largest tensor it touches stays inside
    # both (``clips_per_pass``); larger batches are evaluated in passes (clips are independent in eval mode).  At the reference
    # configuration that is 64+ clips per pass on every datapath that uses the tap form (BASELINE configs[4]: 64 clips per GPU in
    # one pass), 32 on the direct fp32 form whose 4-scale sum [B,112,192,768] is 66 MB per clip.  ``max_clips_per_pass`` caps it.
    max_clips_per_pass = 64

    def clips_per_pass(self, tap_form: bool) -> int:
        es = 4 if self.compute_dtype == torch.float32 else 2
        H, W = self.img_size
        ns, T = self.num_stages, 9
        per_clip = [H * W * self.ch]                                   # conv_in output of the two-kernel K2 path
        h, w = H // 32, W // 32
        tok = 0
        for i in range(ns):
            if self.dilation[i] != 0:
                if tap_form:
                    per_clip.append(T * h * w * 9 * self.up_channels[i])   # tap products of UpEmbed conv1 at the source resolution
                h, w = 2 * h, 2 * w
            per_clip.append(T * h * w * self.up_channels[i] * 2)           # tokens and the MLP hidden layer
            tok += h * w
        # mt_proj taps / 4-scale sum; on 16-bit storage the tap products are fp32 (mt_tap16_f32): counted as two elements each
        per_clip.append(tok * 9 * self.down_channel * (2 if es == 2 else 1) if tap_form else 4 * h * w * self.ori_embed_dim)
        worst = max(per_clip)
        by_bytes = ((1 << 32) - (1 << 24)) // (worst * es)
        by_count = ((1 << 31) - (1 << 24)) // worst
        return max(1, min(self.max_clips_per_pass, by_bytes, by_count))

    # eval path: a 3x3 convolution that directly follows a bilinear up-sampling (UpEmbed conv1, mt_proj on the 4-scale sum) runs
    # as nine 1x1 tap mixings at the SOURCE resolution (one GEMM, 4x / 3x fewer FLOPs) + a gather of the interpolated taps
    # (ops.tapsum, csrc/tapsum.hip).  Exact up to summation order; off when intermediate taps are requested.
    tap_conv = True
    # the block's first LayerNorm applied inside qkv_prep (no normalised tensor in HBM).  Bit-equal, but measured slower: the
    # per-token reductions sit on the load path of latency-bound kernels (K9 0.163 -> 0.36 ms for 0.07 ms less K8): off.
    fold_norm1 = False
    # finest stage (C = 96): the block's first half is ONE launch (csrc/tblock.hip).  Off: the per-operator launches below
    fused_front = True
    # proj_k / proj_v inside the pooled launch (the whole C = 96 block is then 3 launches).  Measured: 45 us against 31 + 10 us for
    # the pooled launch + the paired projection GEMM at stage 3 (every one of the 648 workgroups re-reads both weight matrices
    # from L2), 43 against 30 us at C = 192: off.
    fold_kv_proj = False
    # fp32 3x3 stride-1 convolutions (ResnetBlock conv1 / conv2, UpEmbed's second convolution) as Winograd F(2x2, 3x3) where the
    # library's planner expects a gain (csrc/wino.hip; ~1e-6 relative transform rounding).  Off: always the direct kernel
    winograd = True
    # ResnetBlocks whose two convolutions take the F(4x4) path: GroupNorm + swish folded into the input transforms, the 1x1
    # shortcut inside conv1's batched launch, norm2's statistics from conv1's output transform (ops.conv3x3_wino4_ex).  Off: the
    # per-operator launches
    fuse_resblock = True
    # forward(..., taps=dict): intermediate tensors are recorded while the SHIPPED forms run (tap / source-resolution forms of
    # UpEmbed's first convolution and mt_proj, fused ResnetBlocks); only the 4-scale sum `multi_scale` does not exist there.  True:
    # the reference's operator order (bilinear up-sampling, then the 3x3 convolutions) with every tap, `multi_scale` included
    taps_reference_forms = False
    _freq_tables: dict = {}       # (device, half) -> timestep-embedding frequencies on the device (constant)
    fold_head = True        # tap path of mt_proj: MLPHead's 96 -> 1 dot product + sigmoid in the gather's epilogue
    merge_qkv_prep = True   # query (dw 3x3 + LN) and pooled key / value (dw k x k + LN) branches of a block in one launch
    pair_kv = True    # key and value projections of a block in one launch (ops.linear_pair)
    up_commute = True  # UpEmbed's first convolution at the source resolution (ops.up2_conv3x3_d2): fp32 where the map is >= 12 x 12,
    up_commute16 = True  # 16-bit storage on every stage
    # fp32, where both UpEmbed convolutions take the F(4x4) path: the second one's input transform interpolates the first one's
    # source-resolution result itself (ops.conv3x3_wino4_ex(up2=...)); the first one's output exists only on its 3-pixel border ring
    fuse_up_pe2 = True
    group_qkv = True  # fp32: query, key and value projections of a block in one grouped launch (ops.linear_group)
    group_qkv_all = False  # ... at every stage, not only where the query product alone leaves workgroup slots free (A/B aid)
    merge_align = True   # the stages' audio align convolutions as one product (eval)
    group_reduce_temp = True   # fp32 tap path: the stages' ReduceTemp products in one grouped launch after the last stage
    # uses of the tap form on 16-bit storage, from {"s1", "s2", "s3", "mt"}.  Off by default: ("s1", "s2") is +3.4 % on the bf16
    # step (1720 -> 1779 steps/s; "s3" and "mt" lose), but the nine tap products are rounded to 16 bits before they are summed
    # and the worst bf16 fixture error moves from 2.2e-2 to 2.9e-2 against a 3e-2 bar.
    tap_conv16 = ()
    # mt_proj on 16-bit storage as the tap form with fp32 tap products (no rounding before the nine interpolated products are summed,
    # so the accuracy reason against tap_conv16 = ("mt",) does not apply) + the fp32 head gather: instead of the 4-scale sum, the
    # 768 -> 96 convolution on it and the head kernel
    mt_tap16_f32 = True

    def forward(self, x: Tensor, t: Tensor, feat_list: Sequence[Tensor], audio_feat_list: Optional[Tensor] = None,
                taps: Optional[dict] = None) -> Tensor:
        """x [B,1,H,W], t [B] (int64 or float), feat_list: 4 x [B,C_i,Tv,h_i,w_i] coarsest first,
        audio_feat_list: [B,512,Tv+1,h_0,w_0] or None  ->  [B,1,img_H,img_W] in (0,1)."""
        B = x.shape[0]
        if self.training:
            # never chunked: BatchNorm batch statistics and the running-stat update are over the WHOLE per-rank batch,
            # as in the reference (R/models/saliency_decoder/common_block.py:196-216 under DDP, cfg batch_size 48)
            return self.forward_train(x, t, feat_list, audio_feat_list)
        if B == 0:
            return x.new_empty((0, 1, self.img_size[0], self.img_size[1]))
        cpp = self.clips_per_pass(self._use_tap_conv(taps, "mt"))
        if B > cpp and taps is None:
            outs = []
            for s in range(0, B, cpp):
                e = min(B, s + cpp)
                outs.append(self._forward_pass(x[s:e], t[s:e], [f[s:e] for f in feat_list],
                                               None if audio_feat_list is None else audio_feat_list[s:e], None))
            return torch.cat(outs, dim=0)
        return self._forward_pass(x, t, feat_list, audio_feat_list, taps)

    def _forward_pass(self, x: Tensor, t: Tensor, feat_list: Sequence[Tensor], audio_feat_list: Optional[Tensor],
                      taps: Optional[dict]) -> Tensor:
        if not x.is_cuda:
            raise RuntimeError("diff_sal_amd.SalUNet runs on the GPU only (no CPU fallback); got a CPU tensor")
        with ops.gemm_precision(self.gemm_precision):
            return self._forward_eval(x, t, feat_list, audio_feat_list, taps)

    def forward_fused_update(self, x: Tensor, t: Tensor, feat_list: Sequence[Tensor], audio_feat_list: Optional[Tensor], *,
                             ex: float, e0: float, A: float, c0: float, c1: float = 0.0, m_prev: Optional[Tensor] = None):
        """One evaluation whose last kernel also does the solver's work: returns (m, x_next) with
        x0 = the network output, m = ex x + e0 x0 (the wrapper's x_start -> noise conversion), x_next = A x + c0 m + c1 m_prev
        (the multistep update).  Eval mode only; bit-equal to forward() followed by the stand-alone sampler kernels."""
        if self.training:
            raise RuntimeError("forward_fused_update is an inference path")
        if not x.is_cuda:
            raise RuntimeError("diff_sal_amd.SalUNet runs on the GPU only (no CPU fallback); got a CPU tensor")
        x = x.contiguous().float()
        B = x.shape[0]
        ms, xs = [], []
        cpp = self.clips_per_pass(self._use_tap_conv(None, "mt"))
        with ops.gemm_precision(self.gemm_precision):
            for s in range(0, B, cpp):
                e = min(B, s + cpp)
                low = self._forward_eval(x[s:e], t[s:e], [f[s:e] for f in feat_list],
                                         None if audio_feat_list is None else audio_feat_list[s:e], None, lowres=True)
                m, xn, _ = ops.resize_update(low, x[s:e], None if m_prev is None else m_prev[s:e].contiguous(), ex, e0, A, c0, c1)
                ms.append(m)
                xs.append(xn)
        return (ms[0], xs[0]) if len(ms) == 1 else (torch.cat(ms), torch.cat(xs))

    def _forward_eval(self, x: Tensor, t: Tensor, feat_list: Sequence[Tensor], audio_feat_list: Optional[Tensor],
                      taps: Optional[dict], lowres: bool = False) -> Tensor:
        pk = self.packed()
        cdt = self.compute_dtype
        x = x.contiguous().float()
        t = t.contiguous()
        B = x.shape[0]
        ns = self.num_stages
        dec = self.invpt_decoder

        noise = self._noise_encoder(x, t, pk, taps)
        frames: List[Optional[Tensor]] = [None] * ns
        todo = []
        for i in range(min(ns, 3)):       # stage-3 features are never read by the decoder (quirk Q2): skip their transpose
            f = feat_list[i].contiguous().float() if i < len(feat_list) else None
            if f is None:
                continue
            nz = None
            if self.image_based and i < len(noise) and tuple(f.shape[-2:]) == tuple(noise[i].shape[1:3]):
                nz = noise[i]
            todo.append((i, f, nz))
        if len(todo) > 1 and all(nz is None or nz.dtype == cdt for _, _, nz in todo):
            outs = ops.pack_frames_multi([f for _, f, _ in todo], [nz for _, _, nz in todo], cdt)    # all stages, one launch
            for (i, _, _), o in zip(todo, outs):
                frames[i] = o
        else:
            for i, f, nz in todo:
                frames[i] = ops.pack_frames(f, nz, out_dtype=cdt)
        if taps is not None:
            for i, nzt in enumerate(noise):
                taps[f"noise{i}"] = nzt

        audio_tok, audio_hw = None, None
        if audio_feat_list is not None:
            a = audio_feat_list.contiguous().float()
            ap = ops.pack_frames(a, None, out_dtype=cdt)  # [B,Ta,ha,wa,512]
            if ap.shape[1] != frames[0].shape[1]:
                raise RuntimeError(f"audio has {ap.shape[1]} frames but the decoder input has {frames[0].shape[1]}")
            audio_hw = (ap.shape[2], ap.shape[3])
            audio_tok = ap.view(B * ap.shape[1], audio_hw[0] * audio_hw[1], ap.shape[4])
            if self.merge_align and "align_all.w" in pk:
                # the four stages' align 1x1 convolutions (transformer.py:133-135) read the same 512-channel audio tokens:
                # ONE product with their output channels side by side (N = 1440) instead of four with N = 768 .. 96
                a_all = ops.linear(audio_tok, pk["align_all.w"], pk["align_all.b"], tag="K7-align")
                audio_tok = (a_all, pk["align_all.offs"])

        xcur = frames[0]
        h0, w0 = xcur.shape[2:4]
        th, tw = h0 * 2 ** (ns - 1) * 2, w0 * 2 ** (ns - 1) * 2
        zs, redu = [], []
        # tap path of mt_proj: the ReduceTemp outputs of all stages land in ONE row-concatenated matrix, so that the nine tap
        # mixings of the four scales are a single GEMM
        z_all, z_off, sizes = None, 0, []
        hh, ww = h0, w0
        for i in range(ns):
            if self.dilation[i] != 0:
                hh, ww = 2 * hh, 2 * ww
            sizes.append((hh, ww))
        if self._use_tap_conv(taps, "mt") and all(th % a_ == 0 and (th // a_) & (th // a_ - 1) == 0 and a_ >= 2 and b_ >= 2 and
                                            tw == b_ * (th // a_) for a_, b_ in sizes):
            z_all = torch.empty((frames[0].shape[0] * sum(a_ * b_ for a_, b_ in sizes), self.ori_embed_dim),
                                device=frames[0].device, dtype=cdt)
        for i in range(ns):
            C = self.up_channels[i]
            if self.dilation[i] != 0:
                Bn, T, h, w, Cp = xcur.shape
                d = self.dilation[i]
                f32c = (self._use_tap_conv(taps, f"s{i}") and self.compute_dtype == torch.float32 and pk[f"s{i}.pe1.wino"] is not None
                        and h >= 12 and w >= 12)
                lowc = (self.compute_dtype != torch.float32 and self.up_commute16 and not (taps is not None and self.taps_reference_forms)
                        and h >= 2 and w >= 2)
                fuse_up = None
                if self.up_commute and d == 2 and (f32c or lowc):
                    # the convolution at the source resolution + interpolation + border-ring corrections.  fp32: F(4x4) there, and only
                    # where the interior is most of the map (stages 2 and 3 at 224 x 384; the alternative is the tap path).  16-bit
                    # storage: every stage (the alternative is the convolution on the up-sampled map: 4x the products)
                    w2 = pk[f"s{i}.pe2.wino"]
                    if (f32c and self.fuse_up_pe2 and w2 is not None and w2.f4 is not None
                            and ops.wino4_supported(_ShapeOnly((Bn * T, 2 * h, 2 * w, C)), C, 2)):
                        # ... and UpEmbed's second convolution forms the interpolated interior itself as its input transform gathers
                        # it: only the border ring of the first convolution's output is ever written
                        u, c_ext = ops.up2_conv3x3_d2(xcur.view(Bn * T, h, w, Cp), pk[f"s{i}.pe1.w"], pk[f"s{i}.pe1.wino"], pk[f"s{i}.pe1.tapw"],
                                                      scale=pk[f"s{i}.pe1.scale"], shift=pk[f"s{i}.pe1.shift"], act=ACT_RELU, tag="K12",
                                                      ring_only=True)
                        fuse_up = (c_ext, pk[f"s{i}.pe1.scale"], pk[f"s{i}.pe1.shift"], ACT_RELU)
                    else:
                        u = ops.up2_conv3x3_d2(xcur.view(Bn * T, h, w, Cp), pk[f"s{i}.pe1.w"], pk[f"s{i}.pe1.wino"], pk[f"s{i}.pe1.tapw"],
                                               scale=pk[f"s{i}.pe1.scale"], shift=pk[f"s{i}.pe1.shift"], act=ACT_RELU, tag="K12")
                elif self._use_tap_conv(taps, f"s{i}") and d in (1, 2) and h >= 2 and w >= 2:
                    y9 = ops.linear(xcur.view(Bn * T, h, w, Cp), pk[f"s{i}.pe1.tapw"], None, tag="K12")
                    u = ops.tapsum([y9], 2 * h, 2 * w, C, dil=d, scale=pk[f"s{i}.pe1.scale"], shift=pk[f"s{i}.pe1.shift"],
                                   act=ACT_RELU, tag="K12-tap")
                else:
                    u = ops.resize_bilinear(xcur.view(Bn * T, h, w, Cp), 2 * h, 2 * w)
                    u = ops.conv_igemm(u, pk[f"s{i}.pe1.w"], kh=3, kw=3, pad=(d, d), dil=(d, d),
                                       scale=pk[f"s{i}.pe1.scale"], shift=pk[f"s{i}.pe1.shift"], act=ACT_RELU, tag="K12")
                skip = frames[i] if i in (1, 2) else None  # transformer.py:265-270
                if fuse_up is not None:
                    u, _, _ = ops.conv3x3_wino4_ex(u, pk[f"s{i}.pe2.wino"].f4, scale=pk[f"s{i}.pe2.scale"], shift=pk[f"s{i}.pe2.shift"],
                                                   act=ACT_RELU, residual=None if skip is None else skip.view(Bn * T, 2 * h, 2 * w, C),
                                                   dil=2, up2=fuse_up, tag="K12")
                else:
                    u = ops.conv_igemm(u, pk[f"s{i}.pe2.w"], kh=3, kw=3, pad=(d, d), dil=(d, d),
                                       scale=pk[f"s{i}.pe2.scale"], shift=pk[f"s{i}.pe2.shift"], act=ACT_RELU, tag="K12",
                                       residual=None if skip is None else skip.view(Bn * T, 2 * h, 2 * w, C),
                                       wino=pk[f"s{i}.pe2.wino"])
                xcur = u.view(Bn, T, 2 * h, 2 * w, C)
            kt = self.temporal_list[i]
            if (xcur.shape[1] - kt) // kt + 1 != 1:
                raise RuntimeError(f"ReduceTemp: T={xcur.shape[1]}, kernel/stride {kt} must give exactly one frame")
            nm = dec.norm_mts[i]
            xcur, z = self._block(i, xcur, pk, audio_tok, audio_hw, norm_z=nm)
            if taps is not None:
                taps[f"stage{i}"] = xcur
            Bn, T, H, W, _ = xcur.shape
            if z is None:
                z = ops.layernorm(xcur, nm.weight, nm.bias, nm.eps)
            z_out = None
            if z_all is not None:
                if (H, W) != sizes[i]:
                    raise RuntimeError(f"stage {i}: {H}x{W} tokens, expected {sizes[i]}")
                z_out = z_all[z_off:z_off + Bn * H * W].view(Bn, 1, H * W, self.ori_embed_dim)
                z_off += Bn * H * W
            if z_out is not None and self.group_reduce_temp and cdt == torch.float32:
                # ReduceTemp of every stage feeds nothing but mt_proj: the four products (M = 336 .. 21504 rows, K = 3840 .. 480)
                # wait for the last stage and share ONE launch, longest K first -- each alone fills a fraction of the chip
                redu.append(dict(x=z.view(Bn, T, H * W, C), w=pk[f"s{i}.redu.w"], kh=kt, kw=1, stride=(kt, 1), act=ACT_RELU, out=z_out))
                zs.append(z_out.view(Bn, H, W, self.ori_embed_dim))
                continue
            z = ops.conv_igemm(z.view(Bn, T, H * W, C), pk[f"s{i}.redu.w"], kh=kt, kw=1, stride=(kt, 1), act=ACT_RELU,
                               out=z_out, tag="K13")
            zs.append(z.view(Bn, H, W, self.ori_embed_dim))
        if redu:
            # a grouped launch has no K split: its makespan is at least the longest unit's K walk (stage 0: 120 slices).  At one clip that
            # is 2.4x the launch's work per CU -- such a product runs on its own (the planner splits its K), the rest stays grouped
            def _units(pr):
                n, t, hw, c = pr["x"].shape
                return -(-n * hw // 96) * -(-pr["w"].shape[0] // 96), pr["kh"] * c // 32
            while len(redu) > 1:
                work = sum(u * k for u, k in map(_units, redu)) / 256.0
                j = max(range(len(redu)), key=lambda q: _units(redu[q])[1])
                if _units(redu[j])[1] <= 1.5 * work:
                    break
                pr = redu.pop(j)
                ops.conv_igemm(pr["x"], pr["w"], kh=pr["kh"], kw=pr["kw"], stride=pr["stride"], act=pr["act"], out=pr["out"], tag="K13")
            ops.conv_igemm_group(redu, tag="K13")
        mt = dec.mt_proj
        if z_all is not None:
            # 16-bit storage: the tap products leave the matrix cores in fp32 and the gather adds them in fp32 (ops.linear(out_f32)):
            # nothing of mt_proj is rounded to 16 bits -- the 4-scale sum [B,112,192,768] (2.1 GB at 64 clips) does not exist either
            f32_taps = cdt != torch.float32 and self.mt_tap16_f32 and "mt" not in self.tap_conv16 and self.ori_embed_dim % 192 == 0
            y9 = ops.linear(z_all, pk["mt.tapw"], None, tag="K14", out_f32=f32_taps)
            ys, off = [], 0
            for z_ in zs:
                m_ = z_.shape[0] * z_.shape[1] * z_.shape[2]
                ys.append(y9[off:off + m_].view(z_.shape[0], z_.shape[1], z_.shape[2], y9.shape[-1]))
                off += m_
            if self.fold_head and mt[0].out_channels <= 128 and mt[0].out_channels % 4 == 0:
                # MLPHead (1x1 to one channel + sigmoid) in the gather's epilogue: the [B,112,192,96] map never reaches HBM
                s = ops.tapsum(ys, th, tw, mt[0].out_channels, dil=1, bias=mt[0].bias, scale=pk["mt.scale"], shift=pk["mt.shift"],
                               act=ACT_RELU, tag="K14-tap", head=(pk["head.w"], self.logits.linear_pred.bias))
                if lowres:
                    return s
                out = ops.resize_bilinear(s, self.img_size[0], self.img_size[1], tag="K14-head")
                return out.view(B, 1, self.img_size[0], self.img_size[1])
            y = ops.tapsum(ys, th, tw, mt[0].out_channels, dil=1, bias=mt[0].bias, scale=pk["mt.scale"], shift=pk["mt.shift"],
                           act=ACT_RELU, tag="K14-tap")
        else:
            acc = ops.resize_sum(zs, th, tw)
            if taps is not None:
                taps["multi_scale"] = acc
            y = ops.conv_igemm(acc, pk["mt.w"], kh=3, kw=3, pad=(1, 1), bias=mt[0].bias, scale=pk["mt.scale"],
                               shift=pk["mt.shift"], act=ACT_RELU, tag="K14")
        s = ops.head_sigmoid(y, pk["head.w"], self.logits.linear_pred.bias)
        if lowres:                       # the caller fuses the final resize with the solver update
            return s
        out = ops.resize_bilinear(s, self.img_size[0], self.img_size[1], tag="K14-head")
        return out.view(B, 1, self.img_size[0], self.img_size[1])

    # ------------------------------------------------------------------ training forward
    dropout_p = 0.1          # ResnetBlock dropout (sal_unet.py:229); set to 0 for gradient-parity tests
    _dropout_calls = 0

    def forward_train(self, x: Tensor, t: Tensor, feat_list: Sequence[Tensor], audio_feat_list: Optional[Tensor] = None,
                      dropout_seed: Optional[int] = None) -> Tensor:
        """Train-mode forward on the autograd tape: BatchNorm uses (per-rank) batch statistics and updates its
        running buffers, ResnetBlock dropout is active, every op is a torch.autograd.Function whose forward AND
        backward are HIP kernels (autograd_ops.py).  Same graph as ``forward``; less epilogue fusion because the
        backward needs the pre-activation tensors."""
        from . import autograd_ops as ag

        if not x.is_cuda:
            raise RuntimeError("diff_sal_amd.SalUNet runs on the GPU only (no CPU fallback); got a CPU tensor")
        if self.compute_dtype != torch.float32:
            raise RuntimeError("SalUNet.forward_train: training stores activations in fp32 (the reference trains in fp32, "
                               "R/diffusion_trainer.py:212-235); compute_dtype is an inference option")
        x = x.contiguous().float()
        B, _, H, W = x.shape
        if B == 0:
            raise RuntimeError("SalUNet.forward_train: empty batch (BatchNorm batch statistics are undefined); "
                               "nn.BatchNorm2d raises for it too")
        ns, dec = self.num_stages, self.invpt_decoder
        if dropout_seed is None:
            SalUNet._dropout_calls += 1
            dropout_seed = (torch.initial_seed() * 1000003 + SalUNet._dropout_calls) & (2 ** 63 - 1)
        pw = ops.pack_conv_weight_diff
        dgw = ag.dgrad_weight

        # K1: embedding table lookup (no parameters) + three small dense layers
        half = self.ch // 2
        # the frequency table is a constant: built on the host once per device (a host-to-device copy of pageable memory in
        # the step blocks the host until the GPU has drained everything queued before it -- the video encoder's forward --
        # and the rest of the step is then issued into an empty queue)
        fk = (str(x.device), half)
        freq = SalUNet._freq_tables.get(fk)
        if freq is None:
            freq = SalUNet._freq_tables[fk] = torch.exp(
                torch.arange(half, dtype=torch.float32) * -(math.log(10000) / (half - 1))).to(x.device)
        arg = t.to(torch.float32)[:, None] * freq[None, :]
        emb = torch.cat([arg.sin(), arg.cos()], dim=1).contiguous()
        d0, d1 = self.temb.dense[0], self.temb.dense[1]
        temb = ag.dense_small(ag.dense_small(emb, d0.weight, d0.bias, False), d1.weight, d1.bias, True)
        tw = torch.cat([blk[0].temb_proj.weight for blk in self.res_encoder], 0)
        tb = torch.cat([blk[0].temb_proj.bias for blk in self.res_encoder], 0)
        tproj = ag.dense_small(temb, tw, tb, True)

        f = ag.conv_in(x, self.conv_in.weight.reshape(self.ch, 9), self.conv_in.bias, 0)
        f = ag.conv(f, pw(self.down1.conv.weight), kh=3, kw=3, stride=(4, 4), out_hw=((H - 2) // 4 + 1, (W - 2) // 4 + 1),
                    bias=self.down1.conv.bias, w_dgrad=dgw(self.down1.conv.weight, (4, 4)))
        noise, off = [], 0
        for i, blk in enumerate(self.res_encoder):
            rb, dn = blk[0], blk[1]
            co = rb.conv1.out_channels
            h = ag.groupnorm_swish(f, rb.norm1.weight, rb.norm1.bias, 32, rb.norm1.eps)
            h = ag.conv(h, pw(rb.conv1.weight), kh=3, kw=3, pad=(1, 1), bias=rb.conv1.bias,
                        rowvec=tproj[:, off:off + co], w_raw=rb.conv1.weight)
            off += co
            h = ag.groupnorm_swish(h, rb.norm2.weight, rb.norm2.bias, 32, rb.norm2.eps)
            h = ag.dropout(h, self.dropout_p, dropout_seed + 7919 * i)
            sc = f
            if hasattr(rb, "nin_shortcut"):
                sc = ag.conv(f, pw(rb.nin_shortcut.weight), bias=rb.nin_shortcut.bias, w_dgrad=dgw(rb.nin_shortcut.weight))
            f = ag.conv(h, pw(rb.conv2.weight), kh=3, kw=3, pad=(1, 1), bias=rb.conv2.bias, residual=sc, w_raw=rb.conv2.weight)
            hh, ww = f.shape[1:3]
            f = ag.conv(f, pw(dn.conv.weight), kh=3, kw=3, stride=(2, 2), out_hw=((hh - 2) // 2 + 1, (ww - 2) // 2 + 1),
                        bias=dn.conv.bias, w_dgrad=dgw(dn.conv.weight, (2, 2)))
            noise.append(f)
        noise = noise[::-1]

        frames: List[Optional[Tensor]] = []
        for i in range(ns):
            fi = feat_list[i].contiguous().float() if (i < len(feat_list) and i < 3) else None
            if fi is None:
                frames.append(None)
                continue
            nz = noise[i] if (self.image_based and i < len(noise) and tuple(fi.shape[-2:]) == tuple(noise[i].shape[1:3])) else None
            frames.append(ag.pack_frames(fi, nz))

        audio_tok, audio_hw = None, None
        if audio_feat_list is not None:
            ap = ag.pack_frames(audio_feat_list.contiguous().float(), None)
            audio_hw = (ap.shape[2], ap.shape[3])
            audio_tok = ap.view(B * ap.shape[1], audio_hw[0] * audio_hw[1], ap.shape[4])

        xcur = frames[0]
        h0, w0 = xcur.shape[2:4]
        th, tw_ = h0 * 2 ** (ns - 1) * 2, w0 * 2 ** (ns - 1) * 2
        zs = []
        for i in range(ns):
            C = self.up_channels[i]
            st = dec.mid_stages[i]
            if self.dilation[i] != 0:
                Bn, T, h, w, Cp = xcur.shape
                d = self.dilation[i]
                pe = st.patch_embed[0].proj
                if self.tap_conv and d in (1, 2) and h >= 2 and w >= 2:   # nine 1x1 mixings at the low resolution + gather (csrc/tapsum.hip), adjoint likewise
                    y9 = ag.linear(xcur.view(Bn * T * h * w, Cp), ag.tap_weight(pe[1].weight))
                    u = ag.tapsum(y9, [(h, w)], Bn * T, 2 * h, 2 * w, C, dil=d)
                else:
                    u = ag.resize_bilinear(xcur.view(Bn * T, h, w, Cp), 2 * h, 2 * w)
                    u = ag.conv(u, pw(pe[1].weight), kh=3, kw=3, pad=(d, d), dil=(d, d), w_raw=pe[1].weight)
                u = ag.batchnorm_relu_train(u, pe[2])
                u = ag.conv(u, pw(pe[4].weight), kh=3, kw=3, pad=(d, d), dil=(d, d), w_raw=pe[4].weight)
                u = ag.batchnorm_relu_train(u, pe[5])
                if i in (1, 2):
                    u = ag.add(u, frames[i].view(Bn * T, 2 * h, 2 * w, C))
                xcur = u.view(Bn, T, 2 * h, 2 * w, C)
            # ---- transformer block
            Bn, T, Hs, Ws, _ = xcur.shape
            n9 = Bn * T
            blk = st.blocks[0]
            a = blk.attn
            xcur, xn = ag.layernorm_fork(xcur, blk.norm.weight, blk.norm.bias, blk.norm.eps)
            k_src = xn.view(n9, Hs, Ws, C)
            if audio_tok is not None:
                a_small = ag.linear(audio_tok, blk.align_conv.weight.reshape(C, 512), blk.align_conv.bias)
                k_src = ag.audio_fuse(a_small, xcur, audio_hw[0], audio_hw[1]).view(n9, Hs, Ws, C)
            kk_ = self.kernel_kv[i]
            wq9 = a.conv_proj_q.conv.weight[:, 0, 1].reshape(C, 9).t().contiguous()
            wk = a.conv_proj_k.conv.weight.reshape(C, kk_ * kk_).t().contiguous()
            wv = a.conv_proj_v.conv.weight.reshape(C, kk_ * kk_).t().contiguous()
            q = ag.layernorm(ag.dwconv(xn.view(n9, Hs, Ws, C), wq9, 3, 1, 1).view(n9, Hs * Ws, C),
                             a.conv_proj_q.bn.weight, a.conv_proj_q.bn.bias, a.conv_proj_q.bn.eps)
            kt_ = ag.dwconv(k_src, wk, kk_, kk_, 0)
            vt_ = ag.dwconv(xn.view(n9, Hs, Ws, C), wv, kk_, kk_, 0)
            kk = ag.layernorm(kt_.view(n9, -1, C), a.conv_proj_k.bn.weight, a.conv_proj_k.bn.bias, a.conv_proj_k.bn.eps)
            vv = ag.layernorm(vt_.view(n9, -1, C), a.conv_proj_v.bn.weight, a.conv_proj_v.bn.bias, a.conv_proj_v.bn.eps)
            q = ag.linear(q, a.proj_q.weight, a.proj_q.bias)
            kk = ag.linear(kk, a.proj_k.weight, a.proj_k.bias)
            vv = ag.linear(vv, a.proj_v.weight, a.proj_v.bias)
            o = ag.attention(q, kk, vv, self.heads[i], float(C) ** -0.5)
            x1 = ag.linear(o, a.proj.weight, a.proj.bias, residual=xcur.view(n9, Hs * Ws, C))
            x1, y = ag.layernorm_fork(x1, blk.norm2.weight, blk.norm2.bias, blk.norm2.eps)
            y = ag.linear(y, blk.mlp.fc1.weight, blk.mlp.fc1.bias)        # pre-activation: fc2's node applies the GELU
            x2 = ag.linear(y, blk.mlp.fc2.weight, blk.mlp.fc2.bias, residual=x1, in_gelu=True)
            xcur = x2.view(Bn, T, Hs, Ws, C)
            # ---- norm + ReduceTemp
            nm = dec.norm_mts[i]
            z = ag.layernorm(xcur, nm.weight, nm.bias, nm.eps)
            kt = self.temporal_list[i]
            if (T - kt) // kt + 1 != 1:
                raise RuntimeError(f"ReduceTemp: T={T}, kernel/stride {kt} must give exactly one frame")
            w3 = dec.redu_chan_up[i].proj[0].weight
            z = ag.conv(z.view(Bn, T, Hs * Ws, C), pw(w3), kh=kt, kw=1, stride=(kt, 1), act=ACT_RELU,
                        w_dgrad=dgw(w3, (kt, 1)))
            zs.append(z.view(Bn, Hs, Ws, self.ori_embed_dim))
        mt = dec.mt_proj
        if self.tap_conv and all(th % z_.shape[1] == 0 and (th // z_.shape[1]) & (th // z_.shape[1] - 1) == 0 and
                                 z_.shape[1] >= 2 and z_.shape[2] >= 2 and tw_ == z_.shape[2] * (th // z_.shape[1]) for z_ in zs):
            z_all = torch.cat([z.reshape(-1, self.ori_embed_dim) for z in zs], 0)
            y9 = ag.linear(z_all, ag.tap_weight(mt[0].weight))
            y = ag.tapsum(y9, [z.shape[1:3] for z in zs], B, th, tw_, mt[0].weight.shape[0], dil=1, bias=mt[0].bias)
        else:
            acc = ag.resize_sum(zs, th, tw_)
            y = ag.conv(acc, pw(mt[0].weight), kh=3, kw=3, pad=(1, 1), bias=mt[0].bias, w_raw=mt[0].weight)
        y = ag.batchnorm_relu_train(y, mt[1])
        s_ = ag.head_sigmoid(y, self.logits.linear_pred.weight.reshape(-1), self.logits.linear_pred.bias)
        out = ag.resize_bilinear(s_, self.img_size[0], self.img_size[1])
        return out.view(B, 1, self.img_size[0], self.img_size[1])

    # taps come back channels-last; helpers for tests that compare with NCHW / NCTHW fixtures
    @staticmethod
    def tap_to_reference_layout(name: str, v: Tensor) -> Tensor:
        if name.startswith("stage"):
            return v.permute(0, 4, 1, 2, 3)
        if name.startswith("noise"):
            return v.permute(0, 3, 1, 2).unsqueeze(2)
        if name in ("down1", "multi_scale") or name.startswith("res"):
            return v.permute(0, 3, 1, 2)
        return v
