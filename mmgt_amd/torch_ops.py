"""PyTorch dispatcher registration of the kernels: `torch.ops.mmgt_hip.*` (north_star "PyTorch-ROCm custom ops").

The C ABI of libmmgt_hip.so (include/mmgt_hip.h) stays the drop-in boundary; this module puts the ops a PyTorch caller would
reach for on top of it as `torch.library` custom ops (device type "cuda" = HIP on ROCm, with shape-only fake kernels so they
trace), so the reference's modules can call `torch.ops.mmgt_hip.gemm(x, w, b, None, 0)` where they call `F.linear` today.
There is no CPU kernel behind any of them: on CPU tensors the dispatcher raises NotImplementedError.

    import mmgt_amd.torch_ops            # registers the namespace
    y = torch.ops.mmgt_hip.gemm(x, w, bias, None, 0)
"""
from typing import List, Optional

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


@_lib.custom_op("mmgt_hip::attn_bank_fwd", mutates_args=(), device_types="cuda")
def attn_bank_fwd(qk: torch.Tensor, v_t: torch.Tensor, k_bank: torch.Tensor, v_bank_t: torch.Tensor, heads: int, scale: float, frames: int,
                  bank_first_batch: int, twin: bool) -> List[torch.Tensor]:
    """Reference attention in read mode (mutual_self_attention.py:149-188): qk (B, N, 2 inner) = [q | k] of B = rows x frames images,
    v_t (B, inner, N) = V^T; images b >= bank_first_batch also attend to the bank k_bank (nb, Nb, inner) / v_bank_t (nb, inner, Nb)
    of their CFG row b // frames.  Returns [out (B, N, inner)]; twin (bank_first_batch == 0): [out over own + bank keys, out over the own
    keys alone] from ONE pass (mmgt_attention_twin: the unconditional row of a CFG pair with shared input)."""
    B, n, two = qk.shape
    inner = two // 2
    hd = inner // heads
    out = qk.new_empty((B, n, inner))
    tw = torch.empty_like(out) if twin else None
    hip.attention(qk, qk[..., inner:], v_t, out, batch=B, heads=heads, hd=hd, nq=n, nk=n, scale=scale, q_str=(n * two, 0, two),
                  k_str=(n * two, 0, two), v_str=(v_t.stride(0), 0, v_t.stride(1)), o_str=(n * inner, 0, inner), v_transposed=True,
                  k2=k_bank, v2=v_bank_t, k2_str=(k_bank.stride(0), k_bank.stride(1)), v2_str=(v_bank_t.stride(0), v_bank_t.stride(1)),
                  k2_bdiv=frames, nk2=k_bank.shape[1], seg2_first_batch=bank_first_batch, twin_out=tw)
    return [out, tw] if twin else [out]


@attn_bank_fwd.register_fake
def _(qk, v_t, k_bank, v_bank_t, heads, scale, frames, bank_first_batch, twin):
    o = qk.new_empty((qk.shape[0], qk.shape[1], qk.shape[2] // 2))
    return [o, torch.empty_like(o)] if twin else [o]


@_lib.custom_op("mmgt_hip::temporal_attn", mutates_args=(), device_types="cuda")
def temporal_attn(qkv: torch.Tensor, frames: int, heads: int) -> torch.Tensor:
    """VersatileAttention over the frame axis (motion_module.py:351-388) IN PLACE of the reference's `(b f) d c -> (b d) f c` rearrange:
    qkv ((b frames), hw, 3C) = [q | k | v] of the token tensor; sequences are the `frames` entries of one (b, pixel).  -> ((b f), hw, C)."""
    bf, hw, c3 = qkv.shape
    c = c3 // 3
    b = bf // frames
    q2 = qkv.reshape(bf * hw, c3)
    o = qkv.new_empty((bf * hw, c))
    st = (frames * hw * c3, c3, hw * c3)
    hip.attention(q2, q2[:, c:], q2[:, 2 * c:], o, batch=b * hw, heads=heads, hd=c // heads, nq=frames, nk=frames, scale=(c // heads) ** -0.5,
                  q_str=st, k_str=st, v_str=st, o_str=(frames * hw * c, c, hw * c), bdiv=hw)
    return o.view(bf, hw, c)


@temporal_attn.register_fake
def _(qkv, frames, heads):
    return qkv.new_empty((qkv.shape[0], qkv.shape[1], qkv.shape[2] // 3))


@_lib.custom_op("mmgt_hip::mmhaa_cross", mutates_args=(), device_types="cuda")
def mmhaa_cross(q3: torch.Tensor, kv3: torch.Tensor, mask_scale: torch.Tensor, heads: int) -> torch.Tensor:
    """The three masked audio cross-attentions of MM-HAA (attention.py:700-760) as one launch: q3 (B, N, 3 inner) = the three branches'
    queries, kv3 (B, La, 6 inner) = [k0 k1 k2 | v0 v1 v2] of the audio tokens, mask_scale (3, B N) fp32 = mask_i * motion_scale_i per
    token.  -> (B, N, 3 inner): branch i's attention output times its mask (the operand of the merged out-projection)."""
    B, n, k3 = q3.shape
    la = kv3.shape[1]
    inner = k3 // 3
    out = torch.empty_like(q3)
    hip.attention(q3, kv3, kv3[..., k3:], out, batch=B, heads=3 * heads, hd=inner // heads, nq=n, nk=la, scale=(inner // heads) ** -0.5,
                  q_str=(n * k3, 0, k3), k_str=(la * 2 * k3, 0, 2 * k3), v_str=(la * 2 * k3, 0, 2 * k3), o_str=(n * k3, 0, k3),
                  out_scale=mask_scale, out_scale_heads=heads)
    return out


@mmhaa_cross.register_fake
def _(q3, kv3, mask_scale, heads):
    return torch.empty_like(q3)


@_lib.custom_op("mmgt_hip::ff_fused", mutates_args=(), device_types="cuda")
def ff_fused(x: torch.Tensor, ln_gamma: torch.Tensor, ln_beta: torch.Tensor, wimg: torch.Tensor, bias2: torch.Tensor, inner: int,
             wpo: Optional[torch.Tensor], bias_po: Optional[torch.Tensor], residual2: Optional[torch.Tensor]) -> torch.Tensor:
    """x + FeedForward(LayerNorm(x)) with the GEGLU FeedForward of diffusers (attention.py:361,642) in ONE launch (x (M, 320) bf16, wimg =
    packing.pack_ff_fused); with wpo (packing.pack_ff_proj_out) the transformer block's proj_out + residual2 ride on the same launch
    (transformer_3d.py:262-268)."""
    if wpo is None:
        return hip.ff_fused(x, ln_gamma, ln_beta, wimg, bias2, x, inner)
    return hip.ff_fused_po(x, ln_gamma, ln_beta, wimg, bias2, x, inner, wpo, bias_po, residual2)


@ff_fused.register_fake
def _(x, ln_gamma, ln_beta, wimg, bias2, inner, wpo, bias_po, residual2):
    return torch.empty_like(x)


@_lib.custom_op("mmgt_hip::rowgemm320", mutates_args=(), device_types="cuda")
def rowgemm320(x: torch.Tensor, wimg: torch.Tensor, n: int, bias: Optional[torch.Tensor], ln_gamma: Optional[torch.Tensor],
               ln_beta: Optional[torch.Tensor], residual: Optional[torch.Tensor]) -> torch.Tensor:
    """[LayerNorm ->] Linear of the 320-channel level as one row-stationary launch (x (M, 320) bf16, wimg = packing.pack_rowgemm(W (n, 320))):
    LayerNorm + to_q / to_k / to_v of attention.py:323-349, motion_module.py:351-366."""
    return hip.rowgemm320(x, wimg, n, bias, ln_gamma=ln_gamma, ln_beta=ln_beta, residual=residual)[0]


@rowgemm320.register_fake
def _(x, wimg, n, bias, ln_gamma, ln_beta, residual):
    return x.new_empty((x.shape[0], n))


@_lib.custom_op("mmgt_hip::temporal_leg", mutates_args=(), device_types="cuda")
def temporal_leg(x: torch.Tensor, ln_gamma: torch.Tensor, beta_pe: torch.Tensor, wimg: torch.Tensor, bias_o: torch.Tensor, batch: int,
                 frames: int, heads: int) -> torch.Tensor:
    """x + to_out(attention over the frames of each pixel(LayerNorm(x) + pe)) of a level-0 motion-module attention block in ONE launch
    (motion_module.py:236-259,351-388): x (batch * frames * n_pix, 320) bf16 rows (batch, frame, pixel), beta_pe (>= frames, 320) fp32 =
    LayerNorm bias + positional-encoding rows, wimg = packing.pack_tleg(Wq, Wk, Wv, Wo).  frames = 24 or 12."""
    n_pix = x.shape[0] // (batch * frames)
    if not hip.temporal_leg320_supported(x.dtype, x.shape[1], heads, frames, n_pix, batch):
        raise RuntimeError(f"mmgt_hip::temporal_leg: built for 8 heads of 40 channels, 12 or 24 frames, bf16 (got {heads} heads, {x.shape[1]} channels, "
                           f"{frames} frames, {x.dtype})")
    return hip.temporal_leg320(x, ln_gamma, beta_pe, wimg, bias_o, batch, frames, n_pix, (x.shape[1] // heads) ** -0.5)


@temporal_leg.register_fake
def _(x, ln_gamma, beta_pe, wimg, bias_o, batch, frames, heads):
    return torch.empty_like(x)


@_lib.custom_op("mmgt_hip::gn_silu_conv3x3", mutates_args=(), device_types="cuda")
def gn_silu_conv3x3(x: torch.Tensor, skip: Optional[torch.Tensor], gamma: torch.Tensor, beta: torch.Tensor, groups: int, eps: float, wimg: torch.Tensor,
                    cout: int, bias: Optional[torch.Tensor], temb: Optional[torch.Tensor], residual: Optional[torch.Tensor]) -> torch.Tensor:
    """conv3x3(silu(GroupNorm(cat(x, skip)))) + bias + temb[n] (+ residual) of a ResnetBlock3D leg (resnet.py:217-247: norm -> nonlinearity -> conv, + temb
    behind conv1, + shortcut behind conv2) on channels-last (N, H, W, C) bf16 frames: the statistics pass + ONE fused launch (csrc/rconv.hip); wimg =
    packing.pack_rconv(conv.weight); temb (rows, cout) fp32 with N % rows == 0 (one row per batch entry)."""
    nb, h, w, c0 = x.shape
    c1 = 0 if skip is None else skip.shape[3]
    if not hip.gn_silu_conv3x3_unet_supported(x.dtype, c0, c1, cout, h, w):
        raise RuntimeError(f"mmgt_hip::gn_silu_conv3x3: built for bf16, H and W multiples of 16, channel counts multiples of 64, Cout a multiple of 160 "
                           f"(got {x.dtype}, {h} x {w}, {c0} + {c1} -> {cout})")
    sc, sh = hip.groupnorm_affine(x.view(nb, h * w, c0), gamma, beta, groups, eps, x1=None if skip is None else skip.view(nb, h * w, c1))
    return hip.gn_silu_conv3x3_unet(x, sc, sh, wimg, cout, bias, temb, 0 if temb is None else nb // temb.shape[0], residual, x1=skip)


@gn_silu_conv3x3.register_fake
def _(x, skip, gamma, beta, groups, eps, wimg, cout, bias, temb, residual):
    return x.new_empty(tuple(x.shape[:3]) + (cout,))


_VAES = {}


@_lib.custom_op("mmgt_hip::vae_decode", mutates_args=(), device_types="cuda")
def vae_decode(z: torch.Tensor, weights: List[torch.Tensor]) -> torch.Tensor:
    """AutoencoderKL.decode of pipeline_pose2vid_long.py:112-125: z (n, 4, h, w) latents (already / 0.18215), weights = the decoder's
    state-dict tensors in the order of mmgt_amd.vae.vae_decoder_spec() -> (n, 3, 8h, 8w).  The packed weights are cached per weight list
    (keyed on the tensors' storage and version, held alive by the entry)."""
    from .vae import AutoencoderKL, vae_decoder_spec
    key = tuple((t.data_ptr(), t._version) for t in weights) + (z.dtype,)
    ent = _VAES.get(key)
    if ent is None:
        vae = AutoencoderKL(device=z.device, dtype=z.dtype)
        vae.load_state_dict(dict(zip(vae_decoder_spec(), weights)))
        _VAES.clear()
        ent = _VAES[key] = (vae, list(weights))
    return ent[0].decode(z).sample


@vae_decode.register_fake
def _(z, weights):
    return z.new_empty((z.shape[0], 3, 8 * z.shape[2], 8 * z.shape[3]))


OPS = ("gemm", "conv3x3_nhwc", "attention", "groupnorm_silu", "layernorm", "cfg_ddim_step", "attn_bank_fwd", "temporal_attn", "mmhaa_cross",
       "ff_fused", "rowgemm320", "temporal_leg", "gn_silu_conv3x3", "vae_decode")
