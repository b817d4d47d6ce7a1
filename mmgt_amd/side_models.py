"""PoseGuider and AudioProjModel on the HIP kernels (once-per-clip conditioning producers of the Stage-2 path).

Mirrors src/models/pose_guider.py:12-57 and src/models/audio_proj.py:40-124: same constructors, forward signatures and
state-dict keys.  PoseGuider's narrow layers (3/16/32/96 channels) are zero-padded to multiples of 64 channels so they
run on the same implicit-GEMM conv kernel (padded outputs stay exactly zero: zero weights, zero bias, SiLU(0) = 0).
"""
from collections import OrderedDict
from typing import Tuple

import torch

from . import hip
from .packing import pack_conv3x3, pad_rows, round_up


def pose_guider_layers(conditioning_embedding_channels: int, block_out_channels, conditioning_channels: int = 3):
    """(name, cin, cout, stride) of PoseGuider's convs (src/models/pose_guider.py:14-52)."""
    boc = tuple(block_out_channels)
    layers = [("conv_in", conditioning_channels, boc[0], 1)]
    k = 0
    for i in range(len(boc) - 1):
        layers.append((f"blocks.{k}", boc[i], boc[i], 1))
        layers.append((f"blocks.{k + 1}", boc[i], boc[i + 1], 2))
        k += 2
    layers.append(("conv_out", boc[-1], conditioning_embedding_channels, 1))
    return layers


def pose_guider_spec(conditioning_embedding_channels: int, block_out_channels=(16, 32, 64, 128), conditioning_channels: int = 3):
    """State-dict keys -> shapes of a PoseGuider (no device needed)."""
    spec = OrderedDict()
    for name, cin, cout, _ in pose_guider_layers(conditioning_embedding_channels, block_out_channels, conditioning_channels):
        spec[name + ".weight"] = (cout, cin, 3, 3)
        spec[name + ".bias"] = (cout,)
    return spec


class PoseGuider:
    def __init__(self, conditioning_embedding_channels: int, conditioning_channels: int = 3,
                 block_out_channels: Tuple[int] = (16, 32, 64, 128), device="cuda", dtype=torch.bfloat16):
        self._device, self._dtype = torch.device(device), dtype
        hip.dtype_code(dtype)
        self.out_channels = conditioning_embedding_channels
        self.layers = pose_guider_layers(conditioning_embedding_channels, block_out_channels, conditioning_channels)
        self.spec = pose_guider_spec(conditioning_embedding_channels, block_out_channels, conditioning_channels)
        self.w = {}
        self._loaded = False

    @property
    def dtype(self):
        return self._dtype

    @property
    def device(self):
        return self._device

    def to(self, *a, **k):
        return self

    def load_state_dict(self, sd, strict=True):
        missing = [k for k in self.spec if k not in sd]
        if missing:
            raise RuntimeError(f"PoseGuider.load_state_dict: missing {missing[:3]}")
        for name, cin, cout, _ in self.layers:
            wt = sd[name + ".weight"]
            if tuple(wt.shape) != (cout, cin, 3, 3):
                raise RuntimeError(f"PoseGuider: shape mismatch for {name}.weight")
            cop = cout if name == "conv_out" else round_up(cout, 64)
            self.w[name + ".w"] = pack_conv3x3(wt, round_up(cin, 64), round_up(cop, 64)).to(self._device, self._dtype)
            self.w[name + ".bias"] = pad_rows(sd[name + ".bias"], round_up(cop, 64)).to(self._device, torch.float32).contiguous()
        self._loaded = True
        return [], [k for k in sd if k not in self.spec]

    def forward_nhwc(self, conditioning):
        """(b, 3, f, H, W) in [0, 1] -> channels-last ((b f), H/8, W/8, 320) in the model dtype."""
        if not self._loaded:
            raise RuntimeError("PoseGuider.forward before load_state_dict")
        x = hip.ncfhw_to_nhwc(conditioning.to(self._device, torch.float32).contiguous(), 64, self._dtype)
        for name, _, _, stride in self.layers:
            act = hip.ACT_NONE if name == "conv_out" else hip.ACT_SILU
            x = hip.conv3x3(x, self.w[name + ".w"], self.w[name + ".bias"], stride=stride, act=act)
        return x

    def forward(self, conditioning):
        b = conditioning.shape[0]
        y = self.forward_nhwc(conditioning)
        return hip.nhwc_to_ncfhw(y, b, self.out_channels).to(conditioning.dtype if conditioning.is_floating_point()
                                                                  else torch.float32)

    __call__ = forward


class AudioProjModel:
    def __init__(self, seq_len=5, blocks=12, channels=768, intermediate_dim=512, output_dim=768, context_tokens=32,
                 device="cuda", dtype=torch.bfloat16):
        self._device, self._dtype = torch.device(device), dtype
        hip.dtype_code(dtype)
        self.input_dim = seq_len * blocks * channels
        self.intermediate_dim, self.context_tokens, self.output_dim = intermediate_dim, context_tokens, output_dim
        self.spec = OrderedDict([
            ("proj1.weight", (intermediate_dim, self.input_dim)), ("proj1.bias", (intermediate_dim,)),
            ("proj2.weight", (intermediate_dim, intermediate_dim)), ("proj2.bias", (intermediate_dim,)),
            ("proj3.weight", (context_tokens * output_dim, intermediate_dim)), ("proj3.bias", (context_tokens * output_dim,)),
            ("norm.weight", (output_dim,)), ("norm.bias", (output_dim,))])
        if self.input_dim % 64 or intermediate_dim % 64:
            raise ValueError("AudioProjModel: input / intermediate dims must be multiples of 64 for the GEMM kernel")
        self.w = {}
        self._loaded = False

    @property
    def dtype(self):
        return self._dtype

    @property
    def device(self):
        return self._device

    def to(self, *a, **k):
        return self

    def load_state_dict(self, sd, strict=True):
        for k, shape in self.spec.items():
            if k not in sd or tuple(sd[k].shape) != tuple(shape):
                raise RuntimeError(f"AudioProjModel.load_state_dict: missing or mis-shaped {k}")
        for p in ("proj1", "proj2", "proj3"):
            self.w[p + ".w"] = sd[p + ".weight"].to(self._device, self._dtype).contiguous()
            self.w[p + ".bias"] = sd[p + ".bias"].to(self._device, torch.float32).contiguous()
        self.w["norm.g"] = sd["norm.weight"].to(self._device, torch.float32).contiguous()
        self.w["norm.b"] = sd["norm.bias"].to(self._device, torch.float32).contiguous()
        self._loaded = True
        return [], [k for k in sd if k not in self.spec]

    def forward(self, audio_embeds):
        """(bz, f, window, blocks, channels) -> (bz, f, context_tokens, output_dim)   (audio_proj.py:96-124)."""
        if not self._loaded:
            raise RuntimeError("AudioProjModel.forward before load_state_dict")
        bz, f = audio_embeds.shape[:2]
        x = audio_embeds.to(self._device, self._dtype).reshape(bz * f, -1).contiguous()
        x = hip.gemm(x, self.w["proj1.w"], self.w["proj1.bias"], act=hip.ACT_RELU)
        x = hip.gemm(x, self.w["proj2.w"], self.w["proj2.bias"], act=hip.ACT_RELU)
        x = hip.gemm(x, self.w["proj3.w"], self.w["proj3.bias"])
        x = hip.layernorm(x.view(bz * f * self.context_tokens, self.output_dim), self.w["norm.g"], self.w["norm.b"], 1e-5)
        return x.view(bz, f, self.context_tokens, self.output_dim).to(audio_embeds.dtype)

    __call__ = forward
