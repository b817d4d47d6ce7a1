"""Bit-reproducible synthetic weights and inputs for the Stage-2 denoising path.

No checkpoints ship with the reference (`pretrained_weights/put_weights_on_here`), so every parity test,
golden fixture and bench run fills the reference's state-dict keys from a counter-based integer hash keyed by
the parameter name.  The arithmetic is exact (integer mix -> 24-bit mantissa -> one IEEE multiply), so the
build container, the GPU box's host and the GPU itself all produce identical bits without sharing any file.
"""
import math
import zlib

import torch

_M32 = 0xFFFFFFFF


def _seed(name: str) -> int:
    return zlib.crc32(name.encode("utf-8")) & _M32


def hash_u24(name: str, numel: int, device="cpu") -> torch.Tensor:
    """uint24 stream for `name` (murmur3 finaliser over a Weyl sequence), as int64."""
    x = torch.arange(numel, dtype=torch.int64, device=device)
    x.mul_(0x9E3779B1).add_(_seed(name) * 0x85EBCA77 + 0x165667B1).bitwise_and_(_M32)
    x.bitwise_xor_(x >> 16)
    x.mul_(0x85EBCA6B).bitwise_and_(_M32)
    x.bitwise_xor_(x >> 13)
    x.mul_(0xC2B2AE35).bitwise_and_(_M32)
    x.bitwise_xor_(x >> 16)
    return x.bitwise_right_shift_(8)


def hash_uniform(name: str, shape, scale: float = 1.0, device="cpu", dtype=torch.float32) -> torch.Tensor:
    """U(-scale, scale) tensor, a pure function of (name, shape, scale)."""
    n = 1
    for s in shape:
        n *= int(s)
    u = hash_u24(name, n, device).to(torch.float32) * (1.0 / (1 << 24))  # exact: [0,1) on a 2^-24 grid
    v = (u - 0.5) * (2.0 * float(scale))
    return v.reshape(tuple(shape)).to(dtype)


def sinusoid_pe(max_len: int, d_model: int) -> torch.Tensor:
    """The `pos_encoder.pe` buffer of the motion modules (reference: src/models/motion_module.py:262-273)."""
    position = torch.arange(max_len).unsqueeze(1)
    div_term = torch.exp(torch.arange(0, d_model, 2) * (-math.log(10000.0) / d_model))
    pe = torch.zeros(1, max_len, d_model)
    pe[0, :, 0::2] = torch.sin(position * div_term)
    pe[0, :, 1::2] = torch.cos(position * div_term)
    return pe


_ZERO_INIT = ("zero_conv_full", "zero_conv_face", "zero_conv_lip", "temporal_transformer.proj_out")


def synth_tensor(name: str, shape, device="cpu", dtype=torch.float32, zero_init_gain: float = 0.5):
    """One state-dict entry.  Weights U(+-1/sqrt(fan_in)), biases U(+-0.05), norm gamma 1+-0.1, norm beta +-0.05.
    Layers the reference zero-initialises (attention.py:556-566, motion_module.py:72-75, pose_guider.py:38-45) are
    randomised too (at `zero_init_gain`) so that every branch contributes to the output."""
    shape = tuple(int(s) for s in shape)
    leaf = name.rsplit(".", 1)[-1]
    if leaf == "pe":
        return sinusoid_pe(shape[1], shape[2]).to(device=device, dtype=dtype)
    is_norm = any(t in name for t in (".norm", "norm1", "norm2", "norm3", "ff_norm", "norms.", "conv_norm_out", "layrnorm",
                                      "layernorm")) and len(shape) == 1
    if is_norm:
        base = 1.0 if leaf == "weight" else 0.0
        amp = 0.1 if leaf == "weight" else 0.05
        return (hash_uniform(name, shape, amp, device) + base).to(dtype)
    if leaf == "bias" or len(shape) == 1:
        return hash_uniform(name, shape, 0.05, device, dtype)
    fan_in = 1
    for s in shape[1:]:
        fan_in *= s
    gain = zero_init_gain if (any(z in name for z in _ZERO_INIT) or name.startswith("conv_out_pose")) else 1.0
    return hash_uniform(name, shape, gain / math.sqrt(fan_in), device, dtype)


def synth_state_dict(spec, prefix: str = "", device="cpu", dtype=torch.float32):
    """spec: mapping name -> shape (e.g. a reference `state_dict()` or `unet3d_spec`)."""
    out = {}
    for name, shape in spec.items():
        shape = tuple(shape.shape) if hasattr(shape, "shape") else tuple(shape)
        out[name] = synth_tensor(prefix + name, shape, device, dtype)
    return out


def synth_masks(tag: str, frames: int, latent_hw: int, device="cpu", dtype=torch.float32):
    """Motion-mask pyramid in the operator's layout: list[4] of (frames, (hw/2^k)^2) in [0,1]
    (reference producer: src/dataset/image_processor.py:311-333).  One Gaussian blob per frame at a hashed
    location, min-max normalised at level 0, then 2x average-pooled per level."""
    cx = hash_uniform(tag + ".cx", (frames,), 0.5, device) + 0.5
    cy = hash_uniform(tag + ".cy", (frames,), 0.5, device) + 0.5
    ys = (torch.arange(latent_hw, device=device, dtype=torch.float32) + 0.5) / latent_hw
    d2 = (ys[None, :, None] - cy[:, None, None]) ** 2 + (ys[None, None, :] - cx[:, None, None]) ** 2
    g = torch.exp(-d2 / (2 * 0.125 ** 2))
    g = (g - g.amin(dim=(1, 2), keepdim=True)) / (g.amax(dim=(1, 2), keepdim=True) - g.amin(dim=(1, 2), keepdim=True))
    levels = []
    cur = g[:, None]
    for k in range(4):
        levels.append(cur.reshape(frames, -1).to(dtype))
        if cur.shape[-1] >= 2:
            cur = torch.nn.functional.avg_pool2d(cur, 2)
    return levels


def bank_spatial(block_out_channels=(320, 640, 1280, 1280), latent=64):
    """{reference-attention reader prefix: (N, C)} in the reference's module order down -> up -> mid."""
    boc, h = block_out_channels, latent
    out = {}
    for i in range(3):
        for j in range(2):
            out[f"down_blocks.{i}.attentions.{j}"] = ((h >> i) ** 2, boc[i])
    for i in range(1, 4):
        for j in range(3):
            out[f"up_blocks.{i}.attentions.{j}"] = ((h >> (3 - i)) ** 2, boc[3 - i])
    out["mid_block.attentions.0"] = ((h >> 3) ** 2, boc[3])
    return out


def build_synthetic_pipeline(dev, dtype, with_prologue=True):
    """Pose2VideoPipeline over random-init weights of the reference architecture (no checkpoints ship with the reference):
    UNet3D denoiser (+ ReferenceNet, PoseGuider, VAE, CLIP ViT-L/14 when with_prologue), scripts/pose2vid.py:144-197."""
    from .pipeline import Pose2VideoPipeline
    from .scheduler import DDIMScheduler
    from .unet3d import UNet3DConditionModel
    from .unet3d_spec import unet2d_reference_spec, unet3d_spec
    unet = UNet3DConditionModel(device=dev, dtype=dtype)
    unet.load_state_dict(synth_state_dict(unet3d_spec(), device=dev))
    unet.enable_gradient_checkpointing()                      # scripts/pose2vid.py:183-184
    ref = pg = vae = clip = None
    if with_prologue:
        from .clip_vision import CLIPVisionModelWithProjection, clip_vision_spec
        from .reference_unet import UNet2DConditionModel
        from .side_models import PoseGuider
        from .vae import AutoencoderKL, vae_decoder_spec, vae_encoder_spec
        ref = UNet2DConditionModel(device=dev, dtype=dtype)
        ref.load_state_dict(synth_state_dict(unet2d_reference_spec(), prefix="refnet.", device=dev))
        pg = PoseGuider(320, block_out_channels=(16, 32, 96, 256), device=dev, dtype=dtype)   # :158
        pg.load_state_dict(synth_state_dict(pg.spec, prefix="pose_guider.", device=dev))
        vae = AutoencoderKL(device=dev, dtype=dtype)
        vae_spec = vae_decoder_spec()
        vae_spec.update(vae_encoder_spec())                   # the encoder turns the reference image into ref_image_latents
        vae.load_state_dict(synth_state_dict(vae_spec, prefix="vae.", device=dev))
        clip = CLIPVisionModelWithProjection(device=dev, dtype=dtype)     # ViT-L/14, the reference's image_encoder (:158-162)
        clip.load_state_dict(synth_state_dict(clip_vision_spec(), prefix="clip.", device=dev))
    return Pose2VideoPipeline(vae=vae, image_encoder=clip, reference_unet=ref, denoising_unet=unet, pose_guider=pg,
                              scheduler=DDIMScheduler())
