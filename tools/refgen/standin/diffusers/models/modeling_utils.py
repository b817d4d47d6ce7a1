import torch
from torch import nn


class ModelMixin(nn.Module):
    _supports_gradient_checkpointing = False

    def __getattr__(self, name):
        # diffusers 0.24.0 ModelMixin: config entries are reachable as attributes (transformer_3d.py:160 relies on it)
        d = self.__dict__.get("_internal_dict")
        if d is not None and name in d and name not in self.__dict__:
            return d[name]
        return super().__getattr__(name)

    @property
    def dtype(self):
        return next(self.parameters()).dtype

    @property
    def device(self):
        return next(self.parameters()).device

    def enable_gradient_checkpointing(self):
        self.apply(lambda m: self._set_gradient_checkpointing(m, value=True))

    def disable_gradient_checkpointing(self):
        self.apply(lambda m: self._set_gradient_checkpointing(m, value=False))

    def _set_gradient_checkpointing(self, module, value=False):
        if hasattr(module, "gradient_checkpointing"):
            module.gradient_checkpointing = value
