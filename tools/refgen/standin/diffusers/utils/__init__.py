import logging as _pylogging
from collections import OrderedDict
from dataclasses import fields

SAFETENSORS_WEIGHTS_NAME = "diffusion_pytorch_model.safetensors"
WEIGHTS_NAME = "diffusion_pytorch_model.bin"


class BaseOutput(OrderedDict):
    """dataclass-backed output: attribute access plus integer indexing over the fields."""

    def __post_init__(self):
        for f in fields(self):
            OrderedDict.__setitem__(self, f.name, getattr(self, f.name))

    def __getitem__(self, k):
        if isinstance(k, str):
            return OrderedDict.__getitem__(self, k)
        return tuple(self.values())[k]


class _Logging:
    @staticmethod
    def get_logger(name):
        return _pylogging.getLogger(name)


logging = _Logging()


USE_PEFT_BACKEND = False


def deprecate(*a, **k):
    pass


def is_torch_version(op, ver):
    return True


def scale_lora_layers(*a, **k):
    pass


def unscale_lora_layers(*a, **k):
    pass
