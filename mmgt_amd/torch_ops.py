"""PyTorch dispatcher registration of the kernels: `torch.ops.mmgt_hip.*` (north_star "PyTorch-ROCm custom ops").

The C ABI of libmmgt_hip.so (include/mmgt_hip.h) stays the drop-in boundary; this module puts the ops a PyTorch caller would
reach for on top of it as `torch.library` custom ops (device type "cuda" = HIP on ROCm, with shape-only fake kernels so they
trace), so the reference's modules can call `torch.ops.mmgt_hip.gemm(x, w, b, None, 0)` where they call `F.linear` today.
There is no CPU kernel behind any of them: on CPU tensors the dispatcher raises NotImplementedError.

    import mmgt_amd.torch_ops            # registers the namespace
    y = torch.ops.mmgt_hip.gemm(x, w, bias, None, 0)
"""
from typing import Optional

import torch

from . import hip

_lib = torch.library


@_lib.custom_op("mmgt_hip::gemm", mutates_args=(), device_types="cuda")
def gemm(a: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor], residual: Optional[torch.Tensor], act: int) -> torch.Tensor:
    """epilogue(a[M, K] @ w[N, K]^T + bias) (+ residual); act: 0 none, 1 GEGLU (packed weights), 2 SiLU, 3 ReLU, 4 quick-GELU.
    Replaces nn.Linear / 1x1 conv / FeedForward call sites (include/mmgt_hip.h: mmgt_gemm)."""
    return hip.gemm(a, w, bias, residual=residual, act=act)


@gemm.register_fake
def _(a, w, bias, residual, act):
    n = w.shape[0] // 2 if act == hip.ACT_GEGLU else w.shape[0]
    return a.new_empty((a.shape[0], n))


@_lib.custom_op("mmgt_hip::conv3x3_nhwc", mutates_args=(), device_types="cuda")
def conv3x3_nhwc(x: torch.Tensor, w_packed: torch.Tensor, bias: Optional[torch.Tensor], residual: Optional[torch.Tensor],
                 stride: int, upsample: bool) -> torch.Tensor:
    """3x3 / pad 1 conv on channels-last (N, H, W, C), weights packed [Cout][3][3][Cin] (mmgt_amd.packing.pack_conv3x3);
    upsample = fused nearest-2x of the input.  Replaces InflatedConv3d / Upsample3D / Downsample3D (resnet.py:9-120)."""
    return hip.conv3x3(x, w_packed, bias, stride=stride, upsample=upsample, residual=residual)


@conv3x3_nhwc.register_fake
def _(x, w_packed, bias, residual, stride, upsample):
    nb, h, w_, _ = x.shape
    vh, vw = (2 * h, 2 * w_) if upsample else (h, w_)
    return x.new_empty((nb, (vh - 1) // stride + 1, (vw - 1) // stride + 1, w_packed.shape[0]))


@_lib.custom_op("mmgt_hip::attention", mutates_args=(), device_types="cuda")
def attention(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, heads: int, scale: float) -> torch.Tensor:
    """softmax(q k^T scale) v for token-major q (B, Nq, H*d), k / v (B, Nk, H*d), d in {40, 64, 80, 160}: what diffusers'
    AttnProcessor2_0 computes between to_q/to_k/to_v and to_out (include/mmgt_hip.h: mmgt_attention)."""
    b, nq, inner = q.shape
    nk = k.shape[1]
    q, k, v = q.contiguous(), k.contiguous(), v.contiguous()
    out = torch.empty_like(q)
    hip.attention(q, k, v, out, batch=b, heads=heads, hd=inner // heads, nq=nq, nk=nk, scale=scale,
                  q_str=(nq * inner, 0, inner), k_str=(nk * inner, 0, inner), v_str=(nk * inner, 0, inner),
                  o_str=(nq * inner, 0, inner))
    return out


@attention.register_fake
def _(q, k, v, heads, scale):
    return torch.empty_like(q)


@_lib.custom_op("mmgt_hip::groupnorm_silu", mutates_args=(), device_types="cuda")
def groupnorm_silu(x: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, groups: int, eps: float, silu: bool) -> torch.Tensor:
    """Per-image GroupNorm (+ SiLU) on channels-last (N, HW, C).  Replaces InflatedGroupNorm + SiLU (resnet.py:20-28)."""
    return hip.groupnorm(x, gamma, beta, groups, eps, silu=silu)


@groupnorm_silu.register_fake
def _(x, gamma, beta, groups, eps, silu):
    return torch.empty_like(x)


@_lib.custom_op("mmgt_hip::layernorm", mutates_args=(), device_types="cuda")
def layernorm(x: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, eps: float) -> torch.Tensor:
    """LayerNorm over the last dim of (rows, C).  Replaces nn.LayerNorm (attention.py:331-362, motion_module.py:228-234)."""
    return hip.layernorm(x, gamma, beta, eps)


@layernorm.register_fake
def _(x, gamma, beta, eps):
    return torch.empty_like(x)


@_lib.custom_op("mmgt_hip::cfg_ddim_step", mutates_args=(), device_types="cuda")
def cfg_ddim_step(pred_sum: torch.Tensor, counter: torch.Tensor, latents: torch.Tensor, guidance: float, sa_t: float,
                  sb_t: float, sa_p: float, sb_p: float) -> torch.Tensor:
    """Overlap average + CFG combine + DDIM v-prediction update (pipeline_pose2vid_long.py:627-635)."""
    return hip.cfg_ddim_step(pred_sum, counter, latents, guidance, sa_t, sb_t, sa_p, sb_p)


@cfg_ddim_step.register_fake
def _(pred_sum, counter, latents, guidance, sa_t, sb_t, sa_p, sb_p):
    return torch.empty_like(latents)


OPS = ("gemm", "conv3x3_nhwc", "attention", "groupnorm_silu", "layernorm", "cfg_ddim_step")
