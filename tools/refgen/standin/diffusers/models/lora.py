"""LoRA-compatible layers without any LoRA attached == the plain layer (diffusers 0.24.0, lora_layer is None)."""
from torch import nn


class LoRACompatibleConv(nn.Conv2d):
    def forward(self, x, scale=1.0):
        return super().forward(x)


class LoRACompatibleLinear(nn.Linear):
    def forward(self, x, scale=1.0):
        return super().forward(x)
