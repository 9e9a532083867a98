"""Exponential moving average of a module's trainable parameters: the reference's ``EMAHelper``
(``models/diffusion_decoder/ema.py:4-48``; same methods, same ``shadow`` dict keyed by parameter name, which is also its
``state_dict``).  On device tensors the update ``shadow = (1 - mu) * p + mu * shadow`` is one ``axpbypcz`` launch per
parameter (in place); on host tensors (a module that has not been moved yet) it is plain host arithmetic."""
import torch.nn as nn

from . import ops


def _unwrap(module):
    if isinstance(module, (nn.DataParallel, nn.parallel.DistributedDataParallel)):
        return module.module
    return module


class EMAHelper(object):
    def __init__(self, mu=0.999):
        self.mu = mu
        self.shadow = {}

    def register(self, module):
        for name, param in _unwrap(module).named_parameters():
            if param.requires_grad:
                self.shadow[name] = param.data.clone()

    def update(self, module):
        for name, param in _unwrap(module).named_parameters():
            if not param.requires_grad:
                continue
            s = self.shadow[name]
            if param.is_cuda and param.numel() > 0:
                if not (param.data.is_contiguous() and s.is_contiguous() and param.dtype == s.dtype):
                    raise RuntimeError(f"EMAHelper.update: parameter {name!r} must be contiguous and match its shadow's dtype")
                ops.axpbypcz(param.data, 1.0 - self.mu, s, self.mu, out=s)
            else:
                self.shadow[name] = (1.0 - self.mu) * param.data + self.mu * s

    def ema(self, module):
        for name, param in _unwrap(module).named_parameters():
            if param.requires_grad:
                param.data.copy_(self.shadow[name].data)
        # `param.data.copy_` does not move the version counters the packed-weight caches key on: tell EVERY module of the tree
        # (a VideoSaliencyModel nests the denoiser and the encoders) that its parameters changed
        for m in _unwrap(module).modules():
            if hasattr(m, "parameters_updated"):
                m.parameters_updated()

    def ema_copy(self, module):
        inner = _unwrap(module)
        dev = getattr(inner.config, "device", None)
        module_copy = type(inner)(inner.config)
        module_copy = module_copy.to(dev) if dev is not None else module_copy
        module_copy.load_state_dict(inner.state_dict())
        if inner is not module:
            module_copy = nn.DataParallel(module_copy)
        self.ema(module_copy)
        return module_copy

    def state_dict(self):
        return self.shadow

    def load_state_dict(self, state_dict):
        self.shadow = state_dict
