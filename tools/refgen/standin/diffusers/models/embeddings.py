"""Timesteps / TimestepEmbedding, restating diffusers 0.24.0."""
import math

import torch
from torch import nn


class SinusoidalPositionalEmbedding(nn.Module):  # imported by the reference, unused on this path
    def __init__(self, *a, **k):
        raise NotImplementedError


def get_timestep_embedding(timesteps, embedding_dim, flip_sin_to_cos=False, downscale_freq_shift=1.0, scale=1.0,
                           max_period=10000):
    half = embedding_dim // 2
    exponent = -math.log(max_period) * torch.arange(0, half, dtype=torch.float32, device=timesteps.device)
    exponent = exponent / (half - downscale_freq_shift)
    emb = torch.exp(exponent)
    emb = timesteps[:, None].float() * emb[None, :]
    emb = scale * emb
    emb = torch.cat([torch.sin(emb), torch.cos(emb)], dim=-1)
    if flip_sin_to_cos:
        emb = torch.cat([emb[:, half:], emb[:, :half]], dim=-1)
    return emb


class Timesteps(nn.Module):
    def __init__(self, num_channels, flip_sin_to_cos, downscale_freq_shift):
        super().__init__()
        self.num_channels = num_channels
        self.flip_sin_to_cos = flip_sin_to_cos
        self.downscale_freq_shift = downscale_freq_shift

    def forward(self, timesteps):
        return get_timestep_embedding(timesteps, self.num_channels, self.flip_sin_to_cos, self.downscale_freq_shift)


class TimestepEmbedding(nn.Module):
    def __init__(self, in_channels, time_embed_dim, act_fn="silu", out_dim=None, post_act_fn=None, cond_proj_dim=None):
        super().__init__()
        assert act_fn == "silu" and post_act_fn is None and cond_proj_dim is None and out_dim is None
        self.linear_1 = nn.Linear(in_channels, time_embed_dim)
        self.act = nn.SiLU()
        self.linear_2 = nn.Linear(time_embed_dim, time_embed_dim)

    def forward(self, sample, condition=None):
        return self.linear_2(self.act(self.linear_1(sample)))


def _stub(name):
    class _S(nn.Module):
        def __init__(self, *a, **k):
            raise NotImplementedError(name)
    _S.__name__ = name
    return _S


GaussianFourierProjection = _stub("GaussianFourierProjection")
ImageHintTimeEmbedding = _stub("ImageHintTimeEmbedding")
ImageProjection = _stub("ImageProjection")
ImageTimeEmbedding = _stub("ImageTimeEmbedding")
PositionNet = _stub("PositionNet")
TextImageProjection = _stub("TextImageProjection")
TextImageTimeEmbedding = _stub("TextImageTimeEmbedding")
TextTimeEmbedding = _stub("TextTimeEmbedding")
CaptionProjection = _stub("CaptionProjection")
