"""ORACLE (test infrastructure, never imported by the product): CPU fp32 restatement of the wav2vec2 feature extractor the reference
uses as `audio_encoder` -- src/models/wav2vec.py:42-127,196-209 on top of transformers' Wav2Vec2Model (un-vendored dependency,
requirements.txt:207; modeling_wav2vec2.py: Wav2Vec2FeatureEncoder with feat_extract_norm="group", Wav2Vec2FeatureProjection,
Wav2Vec2PositionalConvEmbedding + Wav2Vec2SamePadLayer, Wav2Vec2Encoder with do_stable_layer_norm=False) -- as plain functions over a
state dict with the transformers key names.  Pinned by tests/golden/wav2vec.npz, the outputs of the reference's own Wav2VecModel class
(tools/refgen/gen_wav2vec_golden.py)."""
import torch
import torch.nn.functional as F

KERNELS, STRIDES = (10, 3, 3, 3, 3, 2, 2), (5, 2, 2, 2, 2, 2, 2)


def _pos_weight(sd):
    pre = "encoder.pos_conv_embed.conv."
    g = sd.get(pre + "weight_g", sd.get(pre + "parametrizations.weight.original0"))
    v = sd.get(pre + "weight_v", sd.get(pre + "parametrizations.weight.original1"))
    return g * v / v.norm(dim=(0, 1), keepdim=True)          # nn.utils.weight_norm(conv, name="weight", dim=2)


def feature_extract(sd, wave, seq_len):
    """(1, T) -> (1, seq_len, 512): conv stack (wav2vec.py:73-75) + linear_interpolation (:196-209)."""
    h = wave[:, None]
    for i, (k, s) in enumerate(zip(KERNELS, STRIDES)):
        h = F.conv1d(h, sd[f"feature_extractor.conv_layers.{i}.conv.weight"], stride=s)
        if i == 0:
            h = F.group_norm(h, h.shape[1], sd["feature_extractor.conv_layers.0.layer_norm.weight"],
                             sd["feature_extractor.conv_layers.0.layer_norm.bias"], 1e-5)
        h = F.gelu(h)
    return F.interpolate(h, size=seq_len, align_corners=True, mode="linear").transpose(1, 2)


def encode(sd, feats, heads=12, eps=1e-5):
    """(1, S, 512) -> tuple of 13 hidden states (1, S, 768): wav2vec.py:164-194."""
    ln = lambda x, p: F.layer_norm(x, (x.shape[-1],), sd[p + ".weight"], sd[p + ".bias"], eps)
    lin = lambda x, p: F.linear(x, sd[p + ".weight"], sd[p + ".bias"])
    x = lin(ln(feats, "feature_projection.layer_norm"), "feature_projection.projection")
    pw = _pos_weight(sd)
    pos = F.conv1d(x.transpose(1, 2), pw, sd["encoder.pos_conv_embed.conv.bias"], padding=pw.shape[2] // 2, groups=x.shape[2] // pw.shape[1])
    pos = F.gelu(pos[:, :, :-1] if pw.shape[2] % 2 == 0 else pos).transpose(1, 2)          # Wav2Vec2SamePadLayer
    x = ln(x + pos, "encoder.layer_norm")
    states = [x]
    b, s, h = x.shape
    hd = h // heads
    for i in range(sum(1 for k in sd if k.endswith("attention.q_proj.weight"))):
        p = f"encoder.layers.{i}."
        q = lin(x, p + "attention.q_proj") * hd ** -0.5
        k, v = lin(x, p + "attention.k_proj"), lin(x, p + "attention.v_proj")
        sp = lambda t: t.view(b, s, heads, hd).transpose(1, 2)
        a = torch.softmax(sp(q) @ sp(k).transpose(-1, -2), dim=-1) @ sp(v)
        x = ln(x + lin(a.transpose(1, 2).reshape(b, s, h), p + "attention.out_proj"), p + "layer_norm")
        f = lin(F.gelu(lin(x, p + "feed_forward.intermediate_dense")), p + "feed_forward.output_dense")
        x = ln(x + f, p + "final_layer_norm")
        states.append(x)
    return tuple(states)


def audio_emb(sd, wave, seq_len):
    """audio_processor.py:117-126: (seq_len, 12, 768)."""
    st = encode(sd, feature_extract(sd, wave, seq_len))
    return torch.stack(st[1:], dim=1).squeeze(0).permute(1, 0, 2).contiguous()
