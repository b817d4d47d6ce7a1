from torch import nn


def get_activation(name):
    return {"silu": nn.SiLU(), "swish": nn.SiLU(), "mish": nn.Mish(), "gelu": nn.GELU(), "relu": nn.ReLU()}[name]
