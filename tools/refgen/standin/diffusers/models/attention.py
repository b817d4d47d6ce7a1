"""FeedForward/GEGLU, restating diffusers 0.24.0."""
import torch.nn.functional as F
from torch import nn

from .attention_processor import Attention  # noqa: F401


class AdaLayerNorm(nn.Module):  # never instantiated on this path (num_embeds_ada_norm is None)
    def __init__(self, *a, **k):
        raise NotImplementedError


class GEGLU(nn.Module):
    def __init__(self, dim_in, dim_out):
        super().__init__()
        self.proj = nn.Linear(dim_in, dim_out * 2)

    def forward(self, x):
        h, gate = self.proj(x).chunk(2, dim=-1)
        return h * F.gelu(gate)


class FeedForward(nn.Module):
    def __init__(self, dim, dim_out=None, mult=4, dropout=0.0, activation_fn="geglu", final_dropout=False):
        super().__init__()
        assert activation_fn == "geglu"
        inner = int(dim * mult)
        dim_out = dim_out if dim_out is not None else dim
        self.net = nn.ModuleList([GEGLU(dim, inner), nn.Dropout(dropout), nn.Linear(inner, dim_out)])

    def forward(self, x, scale=1.0):
        for m in self.net:
            x = m(x)
        return x
