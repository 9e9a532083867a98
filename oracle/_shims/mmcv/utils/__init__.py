"""Shim for `mmcv.utils` (test tooling only): Registry / build_from_cfg / get_logger."""
import logging


def build_from_cfg(cfg, registry=None, default_args=None):
    args = dict(cfg)
    if default_args:
        for k, v in default_args.items():
            args.setdefault(k, v)
    kind = args.pop("type")
    if isinstance(kind, str):
        kind = registry.get(kind)
    return kind(**args)


class Registry:
    def __init__(self, name):
        self.name = name
        self._modules = {}

    def get(self, key):
        return self._modules[key]

    def register_module(self, name=None):
        def deco(cls):
            self._modules[name or cls.__name__] = cls
            return cls

        return deco

    def build(self, cfg):
        return build_from_cfg(cfg, self)


def get_logger(name, log_file=None, log_level=logging.INFO):
    return logging.getLogger(name)
