"""VGGish audio feature stack (once per clip), MI355X-native.

Drop-in for ``R/models/vggish.py::VGGish`` (``VGG`` with ``make_layers()``, :70-130): same ``state_dict`` names
(``features.{0,3,6,8,11,13}``, ``embeddings.{0,2,4}``), same ``forward_feat(x)`` / ``forward(x)`` contracts.  The six
3x3 convolutions run on the HIP path -- the single-channel first layer through ``diffsal_conv_in`` (ReLU fused), the rest
through the implicit-GEMM kernel with fused bias + ReLU -- with ``diffsal_maxpool2d`` between them, channels-last
throughout; ``forward_feat`` transposes once at the end because the reference contract is NCHW.
"""
from __future__ import annotations

import os

import torch
from torch import nn

from . import ops
from .ops import ACT_RELU

Tensor = torch.Tensor
CFG = [64, "M", 128, "M", 256, 256, "M", 512, 512, "M"]            # R/models/vggish.py:99-109


def make_layers() -> nn.Sequential:
    layers, cin = [], 1
    for v in CFG:
        if v == "M":
            layers.append(nn.MaxPool2d(kernel_size=2, stride=2))
        else:
            layers += [nn.Conv2d(cin, v, kernel_size=3, padding=1), nn.ReLU(inplace=True)]
            cin = v
    return nn.Sequential(*layers)


class VGGish(nn.Module):
    PRETRAINED = "data/pretrained_models/vggish.pth"               # R/models/vggish.py:115

    def __init__(self, pretrained: bool = True):
        super().__init__()
        self.features = make_layers()                               # parameter storage; forward() of these is never called
        self.embeddings = nn.Sequential(nn.Linear(512 * 4 * 6, 4096), nn.ReLU(True), nn.Linear(4096, 4096), nn.ReLU(True),
                                        nn.Linear(4096, 128), nn.ReLU(True))
        self._pack = None
        self._pack_key = None
        self._pack_epoch = 0
        # The reference runs this module under torch.no_grad() (R/models/diff_model.py:73-74): its 72 M parameters never
        # receive a gradient, yet DDP buckets and all-reduces them.  Freeze them so that optimizers / gradient exchanges
        # built on `requires_grad` skip them (state_dict is unaffected).
        self.requires_grad_(False)
        if pretrained:
            if not os.path.exists(self.PRETRAINED):
                raise FileNotFoundError(f"VGGish(pretrained=True) loads {self.PRETRAINED} as the reference does; not found")
            self.load_state_dict(torch.load(self.PRETRAINED, map_location="cpu"))

    def parameters_updated(self) -> None:
        """Parameters were rewritten without a version bump (``param.data.copy_``, raw-pointer writes): repack on next use."""
        self._pack_epoch += 1

    def _packed(self):
        key = (self._pack_epoch,) + tuple((p.data_ptr(), p._version) for p in self.features.parameters())
        if self._pack is None or key != self._pack_key:
            pk = {}
            for i, m in enumerate(self.features):
                if isinstance(m, nn.Conv2d):
                    pk[i] = (m.weight.detach().reshape(m.out_channels, 9).contiguous() if m.in_channels == 1
                             else ops.pack_conv_weight(m.weight))
            self._pack, self._pack_key = pk, key
        return self._pack

    def features_nhwc(self, x: Tensor) -> Tensor:
        """x [N,1,H,W] -> channels-last features [N, H/16, W/16, 512]."""
        if not x.is_cuda:
            raise RuntimeError("diff_sal_amd.VGGish runs on the GPU only (no CPU fallback); got a CPU tensor")
        pk = self._packed()
        h = x.contiguous().float()
        for i, m in enumerate(self.features):
            if isinstance(m, nn.Conv2d):
                if m.in_channels == 1:
                    h = ops.conv_in(h, pk[i], m.bias, 0, act=ACT_RELU)
                else:
                    h = ops.conv_igemm(h, pk[i], kh=3, kw=3, pad=(1, 1), bias=m.bias, act=ACT_RELU, tag="vgg")
            elif isinstance(m, nn.MaxPool2d):
                h = ops.maxpool2d(h, 2, 2)
        return h

    def forward_feat(self, x: Tensor) -> Tensor:
        """[N,1,H,W] -> [N,512,H/16,W/16] (R/models/vggish.py:93-95)."""
        h = self.features_nhwc(x)
        N, Ho, Wo, Cc = h.shape
        return ops.tokens_to_channels_first(h.view(N, Ho * Wo, Cc), 0).view(N, Cc, Ho, Wo)

    def forward(self, x: Tensor):
        """(features NCHW, 128-d embedding) (R/models/vggish.py:84-91, 126-128); the embedding MLP expects 96x64 inputs."""
        h = self.features_nhwc(x)
        N, Ho, Wo, Cc = h.shape
        e = h.view(N, Ho * Wo * Cc)                                  # == transpose(1,3).transpose(1,2).view(N,-1) of NCHW
        for m in self.embeddings:
            if isinstance(m, nn.Linear):
                e = ops.linear(e, m.weight, m.bias, act=ACT_RELU, tag="vgg")
        return ops.tokens_to_channels_first(h.view(N, Ho * Wo, Cc), 0).view(N, Cc, Ho, Wo), e
