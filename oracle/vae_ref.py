"""ORACLE (test infrastructure): decoder and encoder of diffusers==0.24.0 `AutoencoderKL` (sd-vae-ft-mse layout) in plain torch fp32,
with the diffusers state-dict key names.  diffusers is an un-vendored dependency of the reference
(requirements.txt:36; call sites src/pipelines/pipeline_pose2vid_long.py:112-125,433) and the reference holds no test
for it: PARITY UNPINNED — the structure is restated from the published architecture, anchored by analytic checks
(tests/test_vae.py: identity-weights known answer, per-frame independence)."""
import torch
import torch.nn.functional as F


def _gn(sd, p, x, eps=1e-6):
    return F.group_norm(x, 32, sd[p + ".weight"], sd[p + ".bias"], eps)


def _conv(sd, p, x, padding=1):
    return F.conv2d(x, sd[p + ".weight"], sd[p + ".bias"], padding=padding)


def _resnet(sd, p, x):
    h = _conv(sd, p + ".conv1", F.silu(_gn(sd, p + ".norm1", x)))
    h = _conv(sd, p + ".conv2", F.silu(_gn(sd, p + ".norm2", h)))
    if (p + ".conv_shortcut.weight") in sd:
        x = _conv(sd, p + ".conv_shortcut", x, padding=0)
    return x + h


def _attn(sd, p, x):
    """Attention(512, heads=1, dim_head=512, bias=True, residual_connection=True, GroupNorm 32 eps 1e-6)."""
    b, c, h, w = x.shape
    r = x
    t = _gn(sd, p + ".group_norm", x).view(b, c, h * w).transpose(1, 2)
    q = F.linear(t, sd[p + ".to_q.weight"], sd[p + ".to_q.bias"])
    k = F.linear(t, sd[p + ".to_k.weight"], sd[p + ".to_k.bias"])
    v = F.linear(t, sd[p + ".to_v.weight"], sd[p + ".to_v.bias"])
    o = F.scaled_dot_product_attention(q[:, None], k[:, None], v[:, None])[:, 0]
    o = F.linear(o, sd[p + ".to_out.0.weight"], sd[p + ".to_out.0.bias"])
    return o.transpose(1, 2).reshape(b, c, h, w) + r


def vae_decode(sd, z):
    """AutoencoderKL.decode(z).sample for z (n, 4, h, w) -> (n, 3, 8h, 8w)."""
    x = _conv(sd, "post_quant_conv", z, padding=0)
    x = _conv(sd, "decoder.conv_in", x)
    x = _resnet(sd, "decoder.mid_block.resnets.0", x)
    x = _attn(sd, "decoder.mid_block.attentions.0", x)
    x = _resnet(sd, "decoder.mid_block.resnets.1", x)
    for i in range(4):
        for j in range(3):
            x = _resnet(sd, f"decoder.up_blocks.{i}.resnets.{j}", x)
        if i != 3:
            x = F.interpolate(x, scale_factor=2.0, mode="nearest")
            x = _conv(sd, f"decoder.up_blocks.{i}.upsamplers.0.conv", x)
    x = F.silu(_gn(sd, "decoder.conv_norm_out", x))
    return _conv(sd, "decoder.conv_out", x)


def vae_encode_mean(sd, x):
    """AutoencoderKL.encode(x).latent_dist.mean for x (n, 3, H, W) in [-1, 1] -> (n, 4, H/8, W/8): diffusers 0.24.0
    `Encoder` (DownEncoderBlock2D x4 with Downsample2D(padding=0) = F.pad (0,1,0,1) + stride-2 conv, UNetMidBlock2D with
    one attention head, GroupNorm(32, eps 1e-6) + SiLU + conv_out to 8 moment channels), quant_conv, first half of the
    moments.  Call site in the reference: pipeline_pose2vid_long.py:427-434."""
    h = _conv(sd, "encoder.conv_in", x)
    for i in range(4):
        for j in range(2):
            h = _resnet(sd, f"encoder.down_blocks.{i}.resnets.{j}", h)
        if i != 3:
            p = f"encoder.down_blocks.{i}.downsamplers.0.conv"
            h = F.conv2d(F.pad(h, (0, 1, 0, 1)), sd[p + ".weight"], sd[p + ".bias"], stride=2, padding=0)
    h = _resnet(sd, "encoder.mid_block.resnets.0", h)
    h = _attn(sd, "encoder.mid_block.attentions.0", h)
    h = _resnet(sd, "encoder.mid_block.resnets.1", h)
    h = _conv(sd, "encoder.conv_out", F.silu(_gn(sd, "encoder.conv_norm_out", h)))
    moments = _conv(sd, "quant_conv", h, padding=0)
    return moments[:, :4]


def decode_latents(sd, latents):
    """Pose2VideoPipeline.decode_latents (pipeline_pose2vid_long.py:112-125): frame by frame, (x/2+0.5).clamp(0,1)."""
    b, c, f, h, w = latents.shape
    z = (1 / 0.18215 * latents).permute(0, 2, 1, 3, 4).reshape(b * f, c, h, w)
    frames = torch.cat([vae_decode(sd, z[i:i + 1]) for i in range(z.shape[0])])
    video = frames.reshape(b, f, 3, 8 * h, 8 * w).permute(0, 2, 1, 3, 4)
    return (video / 2 + 0.5).clamp(0, 1)
