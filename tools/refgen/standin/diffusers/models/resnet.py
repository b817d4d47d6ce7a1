"""ResnetBlock2D / Downsample2D / Upsample2D restating diffusers 0.24.0 for the SD-1.5 settings used by MMGT's ReferenceNet
(time_embedding_norm "default", no up/down inside the resnet, output_scale_factor 1)."""
import torch
import torch.nn.functional as F
from torch import nn


class Upsample2D(nn.Module):
    def __init__(self, channels, use_conv=False, use_conv_transpose=False, out_channels=None, name="conv"):
        super().__init__()
        assert use_conv and not use_conv_transpose
        self.channels = channels
        self.out_channels = out_channels or channels
        self.conv = nn.Conv2d(channels, self.out_channels, 3, padding=1)

    def forward(self, hidden_states, output_size=None, scale=1.0):
        if output_size is None:
            hidden_states = F.interpolate(hidden_states, scale_factor=2.0, mode="nearest")
        else:
            hidden_states = F.interpolate(hidden_states, size=output_size, mode="nearest")
        return self.conv(hidden_states)


class Downsample2D(nn.Module):
    def __init__(self, channels, use_conv=False, out_channels=None, padding=1, name="conv"):
        super().__init__()
        assert use_conv and padding == 1
        self.conv = nn.Conv2d(channels, out_channels or channels, 3, stride=2, padding=padding)

    def forward(self, hidden_states, scale=1.0):
        return self.conv(hidden_states)


class ResnetBlock2D(nn.Module):
    def __init__(self, *, in_channels, out_channels=None, temb_channels=512, groups=32, eps=1e-6, dropout=0.0,
                 time_embedding_norm="default", non_linearity="swish", output_scale_factor=1.0, pre_norm=True,
                 skip_time_act=False, **unused):
        super().__init__()
        assert time_embedding_norm == "default"
        out_channels = out_channels or in_channels
        self.norm1 = nn.GroupNorm(groups, in_channels, eps=eps)
        self.conv1 = nn.Conv2d(in_channels, out_channels, 3, padding=1)
        self.time_emb_proj = nn.Linear(temb_channels, out_channels)
        self.norm2 = nn.GroupNorm(groups, out_channels, eps=eps)
        self.dropout = nn.Dropout(dropout)
        self.conv2 = nn.Conv2d(out_channels, out_channels, 3, padding=1)
        self.conv_shortcut = nn.Conv2d(in_channels, out_channels, 1) if in_channels != out_channels else None
        self.output_scale_factor = output_scale_factor

    def forward(self, input_tensor, temb, scale=1.0):
        h = self.conv1(F.silu(self.norm1(input_tensor)))
        h = h + self.time_emb_proj(F.silu(temb))[:, :, None, None]
        h = self.conv2(self.dropout(F.silu(self.norm2(h))))
        if self.conv_shortcut is not None:
            input_tensor = self.conv_shortcut(input_tensor)
        return (input_tensor + h) / self.output_scale_factor
