"""ORACLE (test infrastructure, not product): CPU fp32 restatement of MMGT's Stage-2 denoise-step operator.

Plain PyTorch, functional, driven by a state dict that uses the reference's own key names.
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this file; the product
(`mmgt_amd`) never does.

Pinning: every function below is checked against the reference's own modules (imported in the build container
through tools/refgen, goldens committed under tests/golden/, see tests/test_oracle_golden.py).  The arithmetic the
reference delegates to diffusers==0.24.0 (Attention/AttnProcessor2_0, FeedForward/GEGLU, Timesteps,
TimestepEmbedding; un-vendored, requirements.txt:36) is restated from its published algorithm and is NOT pinned by
any reference test (the reference has none): "parity unpinned" for those pieces, see DESIGN.md.

Reference citations are relative to /root/reference.
"""
import math
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence

import torch
import torch.nn.functional as F


@dataclass
class UNet3DConfig:
    """SD-1.5 unet/config.json merged with config/prompts/animation.yaml:47-75."""
    in_channels: int = 4
    out_channels: int = 4
    block_out_channels: Sequence[int] = (320, 640, 1280, 1280)
    layers_per_block: int = 2
    heads: int = 8                     # `attention_head_dim: 8` is used as the head COUNT (unet_3d.py:150,184,242)
    cross_attention_dim: int = 768
    audio_attention_dim: int = 768
    norm_num_groups: int = 32
    norm_eps: float = 1e-5             # resnets + conv_norm_out; transformer pre-norms use 1e-6
    pe_max_len: int = 32
    down_has_attn: Sequence[bool] = (True, True, True, False)
    up_has_attn: Sequence[bool] = (False, True, True, True)


# --------------------------------------------------------------------------------------------- primitives

def _lin(sd, p, x):
    return F.linear(x, sd[p + ".weight"], sd.get(p + ".bias"))


def _gn(sd, p, x, groups, eps):
    return F.group_norm(x, groups, sd[p + ".weight"], sd[p + ".bias"], eps)


def _ln(sd, p, x):
    return F.layer_norm(x, (x.shape[-1],), sd[p + ".weight"], sd[p + ".bias"], 1e-5)


def _conv(sd, p, x, stride=1, padding=1):
    return F.conv2d(x, sd[p + ".weight"], sd.get(p + ".bias"), stride=stride, padding=padding)


def attention(sd, p, x, ctx, heads):
    """diffusers Attention + AttnProcessor2_0 (App. B-1): bias-free q/k/v, biased out, scale = hd^-0.5."""
    b, n, _ = x.shape
    q = F.linear(x, sd[p + ".to_q.weight"])
    k = F.linear(ctx, sd[p + ".to_k.weight"])
    v = F.linear(ctx, sd[p + ".to_v.weight"])
    hd = q.shape[-1] // heads
    q = q.view(b, -1, heads, hd).transpose(1, 2)
    k = k.view(b, -1, heads, hd).transpose(1, 2)
    v = v.view(b, -1, heads, hd).transpose(1, 2)
    o = F.scaled_dot_product_attention(q, k, v)
    o = o.transpose(1, 2).reshape(b, n, heads * hd)
    return F.linear(o, sd[p + ".to_out.0.weight"], sd[p + ".to_out.0.bias"])


def feed_forward(sd, p, x):
    """diffusers FeedForward(geglu) (App. B-2): h, gate = proj(x).chunk(2); h * gelu(gate); Linear."""
    h, gate = _lin(sd, p + ".net.0.proj", x).chunk(2, dim=-1)
    return _lin(sd, p + ".net.2", h * F.gelu(gate))


def timestep_embedding(sd, timesteps, dim):
    """Timesteps(dim, flip_sin_to_cos=True, shift=0) -> TimestepEmbedding (unet_3d.py:102-105,496-502; App. B-3)."""
    half = dim // 2
    freqs = torch.exp(-math.log(10000.0) * torch.arange(half, dtype=torch.float32) / half)
    emb = timesteps[:, None].float() * freqs[None]
    emb = torch.cat([torch.cos(emb), torch.sin(emb)], dim=-1)
    emb = emb.to(sd["time_embedding.linear_1.weight"].dtype)      # t_emb.to(dtype=self.dtype), unet_3d.py:500
    emb = _lin(sd, "time_embedding.linear_1", emb)
    return _lin(sd, "time_embedding.linear_2", F.silu(emb))


# --------------------------------------------------------------------------------------------- blocks
# All block functions work on the per-frame 4-D view x: (b*f, C, H, W); `f` is the window length.

def resnet_block(sd, p, x, temb, cfg: UNet3DConfig, f):
    """ResnetBlock3D.forward (resnet.py:217-247), InflatedConv3d/InflatedGroupNorm = per-frame 2-D ops."""
    h = F.silu(_gn(sd, p + ".norm1", x, cfg.norm_num_groups, cfg.norm_eps))
    h = _conv(sd, p + ".conv1", h)
    t = _lin(sd, p + ".time_emb_proj", F.silu(temb))            # (b, Cout)
    h = h + t.repeat_interleave(f, dim=0)[:, :, None, None]
    h = F.silu(_gn(sd, p + ".norm2", h, cfg.norm_num_groups, cfg.norm_eps))
    h = _conv(sd, p + ".conv2", h)
    if (p + ".conv_shortcut.weight") in sd:
        x = _conv(sd, p + ".conv_shortcut", x, padding=0)
    return x + h                                                # output_scale_factor = 1


def spatial_transformer(sd, p, x, ehs, bank, cfg: UNet3DConfig, f, do_cfg=True, bank_fp16=True):
    """Transformer3DModel (transformer_3d.py:139-268) around TemporalBasicTransformerBlock as patched by
    ReferenceAttentionControl in read mode (mutual_self_attention.py:149-230).

    bank: (2, N, C) LayerNorm'd ReferenceNet features for this block or None (no reference attention).
    """
    bf, c, hh, ww = x.shape
    res = x
    h = _gn(sd, p + ".norm", x, cfg.norm_num_groups, 1e-6)
    h = _conv(sd, p + ".proj_in", h, padding=0)
    inner = h.shape[1]
    h = h.permute(0, 2, 3, 1).reshape(bf, hh * ww, inner)
    t = p + ".transformer_blocks.0"
    n1 = _ln(sd, t + ".norm1", h)
    if bank is not None:
        bk = bank.to(torch.float16).to(n1.dtype) if bank_fp16 else bank   # update(writer, dtype=fp16): :304,340
        bk = bk.repeat_interleave(f, dim=0)                               # (b t) l c: :150-156
        ctx = torch.cat([n1, bk], dim=1)
        out = attention(sd, t + ".attn1", n1, ctx, cfg.heads) + h
        if do_cfg:                                                        # uncond half recomputed without the bank: :168-188
            half = bf // 2
            out_uc = attention(sd, t + ".attn1", n1[:half], n1[:half], cfg.heads) + h[:half]
            out = torch.cat([out_uc, out[half:]], dim=0)
        h = out
    else:
        h = attention(sd, t + ".attn1", n1, n1, cfg.heads) + h
    n2 = _ln(sd, t + ".norm2", h)
    e = ehs.repeat_interleave(f, dim=0) if ehs.shape[0] != bf else ehs    # transformer_3d.py:166-169
    h = attention(sd, t + ".attn2", n2, e, cfg.heads) + h
    h = feed_forward(sd, t + ".ff", _ln(sd, t + ".norm3", h)) + h
    h = h.reshape(bf, hh, ww, inner).permute(0, 3, 1, 2)
    h = _conv(sd, p + ".proj_out", h, padding=0)
    return h + res


def audio_transformer(sd, p, x, audio, masks, depth, motion_scale, cfg: UNet3DConfig):
    """Transformer3DModel[audio] around AudioTemporalBasicTransformerBlock = MM-HAA (attention.py:649-771).

    audio: (b*f, 32, audio_dim); masks = (full, face, body) pyramids, each list[4] of (b*f, N_level).
    motion_scale None => unweighted sum (eval path), else list of 3 weights (script path).
    """
    bf, c, hh, ww = x.shape
    res = x
    h = _gn(sd, p + ".norm", x, cfg.norm_num_groups, 1e-6)
    h = _conv(sd, p + ".proj_in", h, padding=0)
    inner = h.shape[1]
    h = h.permute(0, 2, 3, 1).reshape(bf, hh * ww, inner)
    t = p + ".transformer_blocks.0"
    n1 = _ln(sd, t + ".norm1", h)
    h = attention(sd, t + ".attn1", n1, n1, cfg.heads) + h
    n2 = _ln(sd, t + ".norm2", h)
    branches = []
    for i, zc in enumerate(("zero_conv_full", "zero_conv_face", "zero_conv_lip")):
        a = attention(sd, f"{t}.attn2_{i}", n2, audio, cfg.heads) * masks[i][depth][:, :, None]
        sz = int(a.shape[1] ** 0.5)                                       # attention.py:726-729 (square latents)
        a = a.reshape(bf, sz, sz, inner).permute(0, 3, 1, 2)
        a = F.conv2d(a, sd[f"{t}.{zc}.weight"], sd[f"{t}.{zc}.bias"])
        branches.append(a.permute(0, 2, 3, 1).reshape(bf, -1, inner))
    if motion_scale is not None:
        h = motion_scale[0] * branches[0] + motion_scale[1] * branches[1] + motion_scale[2] * branches[2] + h
    else:
        h = branches[0] + branches[1] + branches[2] + h
    h = feed_forward(sd, t + ".ff", _ln(sd, t + ".norm3", h)) + h
    h = h.reshape(bf, hh, ww, inner).permute(0, 3, 1, 2)
    h = _conv(sd, p + ".proj_out", h, padding=0)
    return h + res


def motion_module(sd, p, x, cfg: UNet3DConfig, f):
    """VanillaTemporalModule -> TemporalTransformer3DModel (motion_module.py:146-182,236-259,351-388)."""
    bf, c, hh, ww = x.shape
    b = bf // f
    q = p + ".temporal_transformer"
    res = x
    h = _gn(sd, q + ".norm", x, cfg.norm_num_groups, 1e-6)
    h = h.permute(0, 2, 3, 1).reshape(bf, hh * ww, c)
    h = _lin(sd, q + ".proj_in", h)
    t = q + ".transformer_blocks.0"
    d = hh * ww
    for i in range(2):
        n = _ln(sd, f"{t}.norms.{i}", h)
        # (b f) d c -> (b d) f c ; PE is added to the attention input, so it reaches q, k and v (App. A-3c)
        s = n.reshape(b, f, d, c).permute(0, 2, 1, 3).reshape(b * d, f, c)
        s = s + sd[f"{t}.attention_blocks.{i}.pos_encoder.pe"][:, :f]
        s = attention(sd, f"{t}.attention_blocks.{i}", s, s, cfg.heads)
        s = s.reshape(b, d, f, c).permute(0, 2, 1, 3).reshape(bf, d, c)
        h = s + h
    h = feed_forward(sd, t + ".ff", _ln(sd, t + ".ff_norm", h)) + h
    h = _lin(sd, q + ".proj_out", h)
    h = h.reshape(bf, hh, ww, c).permute(0, 3, 1, 2)
    return h + res


# --------------------------------------------------------------------------------------------- whole UNet

def bank_keys(cfg: UNet3DConfig) -> List[str]:
    """Prefixes of the 16 reference-attention readers (one Transformer3DModel each) in the reference's module
    registration order down -> up -> mid (unet_3d.py:117-119,176: mid_block is first set to None, so it registers last)."""
    keys = []
    for i, has in enumerate(cfg.down_has_attn):
        if has:
            keys += [f"down_blocks.{i}.attentions.{j}" for j in range(cfg.layers_per_block)]
    for i, has in enumerate(cfg.up_has_attn):
        if has:
            keys += [f"up_blocks.{i}.attentions.{j}" for j in range(cfg.layers_per_block + 1)]
    keys.append("mid_block.attentions.0")
    return keys


def unet3d_forward(sd: Dict[str, torch.Tensor], cfg: UNet3DConfig, sample, timestep, encoder_hidden_states,
                   audio_embedding=None, pose_cond_fea=None, full_mask=None, face_mask=None, body_mask=None,
                   motion_scale=None, banks: Optional[Dict[str, torch.Tensor]] = None, weighted: bool = True,
                   do_cfg: bool = True, bank_fp16: bool = True, taps: Optional[dict] = None):
    """UNet3DConditionModel.forward (unet_3d.py:425-625).

    sample (b,4,f,h,w); timestep scalar; encoder_hidden_states (b,1,768); audio_embedding (b,f,32,768);
    pose_cond_fea (b,320,f,h,w); *_mask list[4] of (b*f, N_k); banks {prefix: (2,N,C)}.
    weighted=True is the scripts' behaviour (train mode + gradient checkpointing => motion_scale applied);
    weighted=False the eval path (motion_scale ignored).
    """
    b, _, f, hh, ww = sample.shape
    boc = list(cfg.block_out_channels)
    ms = motion_scale if weighted else None
    banks = banks or {}
    masks = (full_mask, face_mask, body_mask)
    t = torch.as_tensor(timestep)
    t = t.reshape(1).expand(b) if t.dim() == 0 else t.expand(b)
    emb = timestep_embedding(sd, t, boc[0])

    def tap(name, v):
        if taps is not None:
            taps[name] = v.reshape(b, f, *v.shape[1:]).permute(0, 2, 1, 3, 4).clone()

    x = sample.permute(0, 2, 1, 3, 4).reshape(b * f, -1, hh, ww)
    x = _conv(sd, "conv_in", x)
    if pose_cond_fea is not None:
        x = x + pose_cond_fea.permute(0, 2, 1, 3, 4).reshape(b * f, -1, hh, ww)
    audio = audio_embedding.reshape(b * f, *audio_embedding.shape[2:]) if audio_embedding is not None else None

    skips = [x]
    for i in range(len(boc)):
        p = f"down_blocks.{i}"
        for j in range(cfg.layers_per_block):
            x = resnet_block(sd, f"{p}.resnets.{j}", x, emb, cfg, f)
            if cfg.down_has_attn[i]:
                k = f"{p}.attentions.{j}"
                x = spatial_transformer(sd, k, x, encoder_hidden_states, banks.get(k), cfg, f, do_cfg, bank_fp16)
                if (f"{p}.audio_modules.{j}.norm.weight") in sd:
                    x = audio_transformer(sd, f"{p}.audio_modules.{j}", x, audio, masks, i, ms, cfg)
            if (f"{p}.motion_modules.{j}.temporal_transformer.norm.weight") in sd:
                x = motion_module(sd, f"{p}.motion_modules.{j}", x, cfg, f)
            skips.append(x)
        tap(f"down{i}", x)
        if i != len(boc) - 1:
            x = _conv(sd, f"{p}.downsamplers.0.conv", x, stride=2, padding=1)
            skips.append(x)

    x = resnet_block(sd, "mid_block.resnets.0", x, emb, cfg, f)
    k = "mid_block.attentions.0"
    x = spatial_transformer(sd, k, x, encoder_hidden_states, banks.get(k), cfg, f, do_cfg, bank_fp16)
    if "mid_block.motion_modules.0.temporal_transformer.norm.weight" in sd:
        x = motion_module(sd, "mid_block.motion_modules.0", x, cfg, f)
    x = resnet_block(sd, "mid_block.resnets.1", x, emb, cfg, f)
    tap("mid", x)

    for i in range(len(boc)):
        p = f"up_blocks.{i}"
        for j in range(cfg.layers_per_block + 1):
            x = torch.cat([x, skips.pop()], dim=1)
            x = resnet_block(sd, f"{p}.resnets.{j}", x, emb, cfg, f)
            if cfg.up_has_attn[i]:
                k = f"{p}.attentions.{j}"
                x = spatial_transformer(sd, k, x, encoder_hidden_states, banks.get(k), cfg, f, do_cfg, bank_fp16)
            if (f"{p}.motion_modules.{j}.temporal_transformer.norm.weight") in sd:
                x = motion_module(sd, f"{p}.motion_modules.{j}", x, cfg, f)
        tap(f"up{i}", x)
        if i != len(boc) - 1:
            x = F.interpolate(x, scale_factor=2.0, mode="nearest")       # Upsample3D [1,2,2] nearest: resnet.py:70-88
            x = _conv(sd, f"{p}.upsamplers.0.conv", x)

    x = F.silu(_gn(sd, "conv_norm_out", x, cfg.norm_num_groups, cfg.norm_eps))
    x = _conv(sd, "conv_out", x)
    return x.reshape(b, f, -1, hh, ww).permute(0, 2, 1, 3, 4).contiguous()


# --------------------------------------------------------------------------------------------- ReferenceNet

def reference_net_banks(sd, cfg: UNet3DConfig, latents, timestep, encoder_hidden_states):
    """ReferenceNet = the reference's UNet2DConditionModel (src/models/unet_2d_condition.py:872-1308, conv_out disabled
    :1296-1299) run once per clip in "write" mode: every BasicTransformerBlock stores its LayerNorm'd input
    (mutual_self_attention.py:139-148).  Returns {reader prefix: (b, N, C)} in module order down -> up -> mid.

    latents (b, 4, h, w); encoder_hidden_states (b, 1, 768).
    """
    b, _, hh, ww = latents.shape
    boc = list(cfg.block_out_channels)
    t = torch.as_tensor(timestep)
    t = t.reshape(1).expand(b) if t.dim() == 0 else t.expand(b)
    emb = timestep_embedding(sd, t, boc[0])
    banks = {}

    def transformer(p, x):
        bf, c, h, w = x.shape
        res = x
        hdn = _gn(sd, p + ".norm", x, cfg.norm_num_groups, 1e-6)
        hdn = _conv(sd, p + ".proj_in", hdn, padding=0)
        inner = hdn.shape[1]
        hdn = hdn.permute(0, 2, 3, 1).reshape(bf, h * w, inner)
        tb = p + ".transformer_blocks.0"
        n1 = _ln(sd, tb + ".norm1", hdn)
        banks[p] = n1.clone()
        hdn = attention(sd, tb + ".attn1", n1, n1, cfg.heads) + hdn
        hdn = attention(sd, tb + ".attn2", _ln(sd, tb + ".norm2", hdn), encoder_hidden_states, cfg.heads) + hdn
        hdn = feed_forward(sd, tb + ".ff", _ln(sd, tb + ".norm3", hdn)) + hdn
        hdn = hdn.reshape(bf, h, w, inner).permute(0, 3, 1, 2)
        return _conv(sd, p + ".proj_out", hdn, padding=0) + res

    x = _conv(sd, "conv_in", latents)
    skips = [x]
    for i in range(len(boc)):
        p = f"down_blocks.{i}"
        for j in range(cfg.layers_per_block):
            x = resnet_block(sd, f"{p}.resnets.{j}", x, emb, cfg, 1)
            if cfg.down_has_attn[i]:
                x = transformer(f"{p}.attentions.{j}", x)
            skips.append(x)
        if i != len(boc) - 1:
            x = _conv(sd, f"{p}.downsamplers.0.conv", x, stride=2, padding=1)
            skips.append(x)
    x = resnet_block(sd, "mid_block.resnets.0", x, emb, cfg, 1)
    mid_bank_holder = {}
    x = transformer("mid_block.attentions.0", x)
    mid = banks.pop("mid_block.attentions.0")
    x = resnet_block(sd, "mid_block.resnets.1", x, emb, cfg, 1)
    for i in range(len(boc)):
        p = f"up_blocks.{i}"
        for j in range(cfg.layers_per_block + 1):
            x = torch.cat([x, skips.pop()], dim=1)
            x = resnet_block(sd, f"{p}.resnets.{j}", x, emb, cfg, 1)
            if cfg.up_has_attn[i]:
                x = transformer(f"{p}.attentions.{j}", x)
        if i != len(boc) - 1:
            x = F.interpolate(x, scale_factor=2.0, mode="nearest")
            x = _conv(sd, f"{p}.upsamplers.0.conv", x)
    banks["mid_block.attentions.0"] = mid
    return banks, x


# --------------------------------------------------------------------------------------------- side models

def pose_guider_forward(sd, cond):
    """PoseGuider.forward (pose_guider.py:47-57): conv_in, 6 blocks (odd ones stride 2), SiLU between, conv_out."""
    b, c, f, hh, ww = cond.shape
    x = cond.permute(0, 2, 1, 3, 4).reshape(b * f, c, hh, ww)
    x = F.silu(_conv(sd, "conv_in", x))
    i = 0
    while f"blocks.{i}.weight" in sd:
        x = F.silu(_conv(sd, f"blocks.{i}", x, stride=2 if i % 2 == 1 else 1))
        i += 1
    x = _conv(sd, "conv_out", x)
    return x.reshape(b, f, *x.shape[1:]).permute(0, 2, 1, 3, 4).contiguous()


def audio_proj_forward(sd, audio_embeds, context_tokens=32, output_dim=768):
    """AudioProjModel.forward (audio_proj.py:96-124)."""
    bz, f = audio_embeds.shape[:2]
    x = audio_embeds.reshape(bz * f, -1)
    x = torch.relu(_lin(sd, "proj1", x))
    x = torch.relu(_lin(sd, "proj2", x))
    x = _lin(sd, "proj3", x).reshape(bz * f, context_tokens, output_dim)
    x = F.layer_norm(x, (output_dim,), sd["norm.weight"], sd["norm.bias"], 1e-5)
    return x.reshape(bz, f, context_tokens, output_dim)
