"""Shim for `timm.models.layers` (test tooling only; not product code).

The reference imports `trunc_normal_`, `DropPath`, `to_2tuple`
(sal_unet.py:6, common_block.py:4, transformer.py:5, mvit.py:7).
"""
import collections.abc
import itertools

import torch
from torch import nn


def trunc_normal_(tensor, mean=0.0, std=1.0, a=-2.0, b=2.0):
    return nn.init.trunc_normal_(tensor, mean=mean, std=std, a=a, b=b)


def to_2tuple(v):
    if isinstance(v, collections.abc.Iterable) and not isinstance(v, str):
        return tuple(v)
    return tuple(itertools.repeat(v, 2))


class DropPath(nn.Module):
    """Per-sample stochastic depth. Identity in eval / p == 0."""

    def __init__(self, drop_prob=0.0):
        super().__init__()
        self.drop_prob = float(drop_prob)

    def forward(self, x):
        if not self.training or self.drop_prob == 0.0:
            return x
        keep = 1.0 - self.drop_prob
        mask = x.new_empty((x.shape[0],) + (1,) * (x.dim() - 1)).bernoulli_(keep)
        return x * mask / keep
