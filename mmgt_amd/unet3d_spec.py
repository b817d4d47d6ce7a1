"""State-dict key -> shape table of the reference's UNet3DConditionModel (1526 keys at SD-1.5 width).

Mirrors the constructor wiring of src/models/unet_3d.py:117-276 and src/models/unet_3d_blocks.py (down blocks carry
audio modules, mid/up blocks do not: SURVEY App. C-1; audio inner dim follows the block INPUT channels: App. C-3) so
that real checkpoints (`denoising_unet-*.pth`, motion-module ckpt) load by name.
"""
from collections import OrderedDict


def _norm(s, p, c):
    s[p + ".weight"] = (c,)
    s[p + ".bias"] = (c,)


def _attn(s, p, dim, inner, ctx):
    s[p + ".to_q.weight"] = (inner, dim)
    s[p + ".to_k.weight"] = (inner, ctx)
    s[p + ".to_v.weight"] = (inner, ctx)
    s[p + ".to_out.0.weight"] = (dim, inner)
    s[p + ".to_out.0.bias"] = (dim,)


def _ff(s, p, dim):
    s[p + ".net.0.proj.weight"] = (8 * dim, dim)
    s[p + ".net.0.proj.bias"] = (8 * dim,)
    s[p + ".net.2.weight"] = (dim, 4 * dim)
    s[p + ".net.2.bias"] = (dim,)


def _resnet(s, p, cin, cout, temb):
    _norm(s, p + ".norm1", cin)
    s[p + ".conv1.weight"] = (cout, cin, 3, 3)
    s[p + ".conv1.bias"] = (cout,)
    s[p + ".time_emb_proj.weight"] = (cout, temb)
    s[p + ".time_emb_proj.bias"] = (cout,)
    _norm(s, p + ".norm2", cout)
    s[p + ".conv2.weight"] = (cout, cout, 3, 3)
    s[p + ".conv2.bias"] = (cout,)
    if cin != cout:
        s[p + ".conv_shortcut.weight"] = (cout, cin, 1, 1)
        s[p + ".conv_shortcut.bias"] = (cout,)


def _spatial(s, p, c, inner, ctx):
    _norm(s, p + ".norm", c)
    s[p + ".proj_in.weight"] = (inner, c, 1, 1)
    s[p + ".proj_in.bias"] = (inner,)
    t = p + ".transformer_blocks.0"
    _attn(s, t + ".attn1", inner, inner, inner)
    _norm(s, t + ".norm1", inner)
    _attn(s, t + ".attn2", inner, inner, ctx)
    _norm(s, t + ".norm2", inner)
    _ff(s, t + ".ff", inner)
    _norm(s, t + ".norm3", inner)
    s[p + ".proj_out.weight"] = (c, inner, 1, 1)
    s[p + ".proj_out.bias"] = (c,)


def _audio(s, p, c, inner, ctx):
    _norm(s, p + ".norm", c)
    s[p + ".proj_in.weight"] = (inner, c, 1, 1)
    s[p + ".proj_in.bias"] = (inner,)
    t = p + ".transformer_blocks.0"
    for z in ("zero_conv_full", "zero_conv_face", "zero_conv_lip"):
        s[f"{t}.{z}.weight"] = (inner, inner, 1, 1)
        s[f"{t}.{z}.bias"] = (inner,)
    _attn(s, t + ".attn1", inner, inner, inner)
    _norm(s, t + ".norm1", inner)
    for i in range(3):
        _attn(s, f"{t}.attn2_{i}", inner, inner, ctx)
    _norm(s, t + ".norm2", inner)
    _ff(s, t + ".ff", inner)
    _norm(s, t + ".norm3", inner)
    s[p + ".proj_out.weight"] = (c, inner, 1, 1)
    s[p + ".proj_out.bias"] = (c,)


def _motion(s, p, c, pe_len):
    q = p + ".temporal_transformer"
    _norm(s, q + ".norm", c)
    s[q + ".proj_in.weight"] = (c, c)
    s[q + ".proj_in.bias"] = (c,)
    t = q + ".transformer_blocks.0"
    for i in range(2):
        _attn(s, f"{t}.attention_blocks.{i}", c, c, c)
        s[f"{t}.attention_blocks.{i}.pos_encoder.pe"] = (1, pe_len, c)
    for i in range(2):
        _norm(s, f"{t}.norms.{i}", c)
    _ff(s, t + ".ff", c)
    _norm(s, t + ".ff_norm", c)
    s[q + ".proj_out.weight"] = (c, c)
    s[q + ".proj_out.bias"] = (c,)


def unet3d_spec(block_out_channels=(320, 640, 1280, 1280), cross_attention_dim=768, audio_attention_dim=768,
                in_channels=4, out_channels=4, layers_per_block=2, pe_len=32):
    boc = list(block_out_channels)
    temb = boc[0] * 4
    s = OrderedDict()
    s["conv_in.weight"] = (boc[0], in_channels, 3, 3)
    s["conv_in.bias"] = (boc[0],)
    s["time_embedding.linear_1.weight"] = (temb, boc[0])
    s["time_embedding.linear_1.bias"] = (temb,)
    s["time_embedding.linear_2.weight"] = (temb, temb)
    s["time_embedding.linear_2.bias"] = (temb,)
    out_c = boc[0]
    for i in range(4):
        in_c, out_c = out_c, boc[i]
        p = f"down_blocks.{i}"
        has_attn = i < 3
        for j in range(layers_per_block):
            cin = in_c if j == 0 else out_c
            if has_attn:
                _spatial(s, f"{p}.attentions.{j}", out_c, out_c, cross_attention_dim)
        for j in range(layers_per_block):
            cin = in_c if j == 0 else out_c
            _resnet(s, f"{p}.resnets.{j}", cin, out_c, temb)
        if has_attn:
            for j in range(layers_per_block):
                cin = in_c if j == 0 else out_c
                # heads * (cin // heads) inner channels with out_c-channel I/O (unet_3d_blocks.py:466-471)
                _audio(s, f"{p}.audio_modules.{j}", out_c, (cin // 8) * 8, audio_attention_dim)
        for j in range(layers_per_block):
            _motion(s, f"{p}.motion_modules.{j}", out_c, pe_len)
        if i != 3:
            s[f"{p}.downsamplers.0.conv.weight"] = (out_c, out_c, 3, 3)
            s[f"{p}.downsamplers.0.conv.bias"] = (out_c,)
    rev = boc[::-1]
    out_c = rev[0]
    up = OrderedDict()
    for i in range(4):
        prev, out_c = out_c, rev[i]
        in_c = rev[min(i + 1, 3)]
        p = f"up_blocks.{i}"
        has_attn = i > 0
        if has_attn:
            for j in range(layers_per_block + 1):
                _spatial(up, f"{p}.attentions.{j}", out_c, out_c, cross_attention_dim)
        for j in range(layers_per_block + 1):
            skip = in_c if j == layers_per_block else out_c
            rin = prev if j == 0 else out_c
            _resnet(up, f"{p}.resnets.{j}", rin + skip, out_c, temb)
        for j in range(layers_per_block + 1):
            _motion(up, f"{p}.motion_modules.{j}", out_c, pe_len)
        if i != 3:
            up[f"{p}.upsamplers.0.conv.weight"] = (out_c, out_c, 3, 3)
            up[f"{p}.upsamplers.0.conv.bias"] = (out_c,)
    s.update(up)
    c = boc[-1]
    _spatial(s, "mid_block.attentions.0", c, c, cross_attention_dim)
    _resnet(s, "mid_block.resnets.0", c, c, temb)
    _resnet(s, "mid_block.resnets.1", c, c, temb)
    _motion(s, "mid_block.motion_modules.0", c, pe_len)
    _norm(s, "conv_norm_out", boc[0])
    s["conv_out.weight"] = (out_channels, boc[0], 3, 3)
    s["conv_out.bias"] = (out_channels,)
    return s


def unet2d_reference_spec(block_out_channels=(320, 640, 1280, 1280), cross_attention_dim=768, in_channels=4,
                          layers_per_block=2):
    """Keys of the reference's ReferenceNet (src/models/unet_2d_condition.py: SD-1.5 UNet2D without conv_norm_out /
    conv_out, :645-653): the 3-D table minus motion / audio modules and the output head (682 keys)."""
    full = unet3d_spec(block_out_channels, cross_attention_dim, cross_attention_dim, in_channels, 4, layers_per_block)
    return OrderedDict((k, v) for k, v in full.items()
                       if "motion_modules" not in k and "audio_modules" not in k
                       and not k.startswith("conv_norm_out") and not k.startswith("conv_out"))
