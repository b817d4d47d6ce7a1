"""ORACLE (test infrastructure, not product): CPU fp32 restatement of MMGT's Stage-1 SMGA audio->pose sampler.

Restates `GestureDecoder` (src/audio2pose_model/model.py:324-490, its layers :44-322), the rotary embedding it applies to the
attention inputs (src/audio2pose_model/rotary_embedding_torch.py:38-132), `SinusoidalPosEmb` / `prob_mask_like` / the cosine
`make_beta_schedule` (src/audio2pose_model/utils.py:38-99) and `GestureDiffusion.ddim_sample` with `model_predictions` /
`predict_noise_from_start` (src/audio2pose_model/diffusion.py:143-156, 241-274), in the configuration the `SMGA`/`LMDM` wrapper
builds (src/audio2pose_model/SMGA.py:62-108): nfeats 402, horizon 80 frames, latent 512, ff 1024, 8 layers, 8 heads,
cond_feature_dim 1059 (WavLM 1024 + 35 baseline features), GELU, rotary; cosine schedule, 1000 train steps, x0-prediction,
guidance weight 2, 50 DDIM steps with eta = 1 and x0 clipped to [-1, 1].

Plain functional PyTorch over a state dict with the reference's own key names.  Pinned against the reference's modules run
in the build container (tools/refgen/gen_smga_golden.py -> tests/golden/smga.npz, tests/test_smga.py).  Only tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg may import this file; the product (`mmgt_amd`) never does.
Reference citations are relative to /root/reference.
"""
import math
from dataclasses import dataclass

import torch
import torch.nn.functional as F


@dataclass
class SMGAConfig:
    nfeats: int = 402
    seq_len: int = 80
    latent_dim: int = 512
    ff_size: int = 1024
    num_layers: int = 8
    num_heads: int = 8
    cond_feature_dim: int = 1059
    n_timestep: int = 1000
    guidance_weight: float = 2.0
    sampling_timesteps: int = 50
    eta: float = 1.0


FACE_LO, FACE_HI = 24, 92       # key points of the face block, 3 values each (model.py:21-32)


def batch_mask(x):
    """model.py:13-41: (face part, body part) of (B, T, 402) key-point vectors: the face keeps points 24..91, the body the rest."""
    b, t, _ = x.shape
    k = x.reshape(b, t, 134, 3)
    face = torch.zeros_like(k)
    face[:, :, FACE_LO:FACE_HI] = k[:, :, FACE_LO:FACE_HI]
    body = k - face
    return face.reshape(b, t, -1), body.reshape(b, t, -1)


def rotary(x, dim):
    """RotaryEmbedding(dim).rotate_queries_or_keys (rotary_embedding_torch.py:106-132,38-61): positions = token index,
    freqs_for='lang', interleaved pairs."""
    n = x.shape[-2]
    freqs = 1.0 / (10000 ** (torch.arange(0, dim, 2)[: dim // 2].float() / dim))
    ang = torch.arange(n).float()[:, None] * freqs[None]
    ang = ang.repeat_interleave(2, dim=-1)                         # "... n -> ... (n r)", r = 2
    x2 = x.reshape(*x.shape[:-1], -1, 2)
    rot = torch.stack((-x2[..., 1], x2[..., 0]), dim=-1).reshape(x.shape)
    return (x * ang.cos() + rot * ang.sin()).to(x.dtype)


def mha(sd, p, q_in, k_in, v_in, heads):
    """nn.MultiheadAttention(batch_first=True), no masks, eval mode."""
    w, b = sd[p + ".in_proj_weight"], sd[p + ".in_proj_bias"]
    d = w.shape[1]
    q = F.linear(q_in, w[:d], b[:d])
    k = F.linear(k_in, w[d:2 * d], b[d:2 * d])
    v = F.linear(v_in, w[2 * d:], b[2 * d:])
    sp = lambda t: t.reshape(t.shape[0], t.shape[1], heads, d // heads).transpose(1, 2)
    o = F.scaled_dot_product_attention(sp(q), sp(k), sp(v)).transpose(1, 2).reshape(q.shape)
    return F.linear(o, sd[p + ".out_proj.weight"], sd[p + ".out_proj.bias"])


def _ln(sd, p, x):
    return F.layer_norm(x, (x.shape[-1],), sd[p + ".weight"], sd[p + ".bias"], 1e-5)


def _lin(sd, p, x):
    return F.linear(x, sd[p + ".weight"], sd[p + ".bias"])


def film(sd, p, t):
    """DenseFiLM (model.py:44-59): Mish -> Linear(d, 2d) -> (scale, shift), broadcast over the sequence."""
    s = _lin(sd, p + ".block.1", F.mish(t))[:, None, :]
    return s.chunk(2, dim=-1)


def affine(x, ss):
    return (ss[0] + 1) * x + ss[1]                                 # featurewise_affine, model.py:62-64


def encoder_layer(sd, p, x, cfg):
    """TransformerEncoderLayer, norm_first (model.py:68-135): rotary on the attention's q/k input only."""
    n1 = _ln(sd, p + ".norm1", x)
    qk = rotary(n1, cfg.latent_dim)
    x = x + mha(sd, p + ".self_attn", qk, qk, n1, cfg.num_heads)
    return x + _lin(sd, p + ".linear2", F.gelu(_lin(sd, p + ".linear1", _ln(sd, p + ".norm2", x))))


def decoder_layer(sd, p, x_face, x_body, cond_tokens, t, cfg):
    """FiLMTransformerDecoderLayer_split.forward (model.py:206-251): a face and a body stream (self-attention, cross-attention
    to the condition tokens, each FiLM-modulated by t), summed, then one feed-forward.  The *_3 norms / FiLMs are unused."""
    def stream(x, part):
        n1 = _ln(sd, f"{p}.norm_{part}_1", x)
        qk = rotary(n1, cfg.latent_dim)
        x = x + affine(mha(sd, f"{p}.{part}_self_attn", qk, qk, n1, cfg.num_heads), film(sd, f"{p}.film_{part}_1", t))
        n2 = _ln(sd, f"{p}.norm_{part}_2", x)
        x2 = mha(sd, f"{p}.{part}_cross_attn", rotary(n2, cfg.latent_dim), rotary(cond_tokens, cfg.latent_dim), cond_tokens,
                 cfg.num_heads)
        return x + affine(x2, film(sd, f"{p}.film_{part}_2", t))
    merged = stream(x_face, "face") + stream(x_body, "body")
    ff = _lin(sd, p + ".linear2", F.gelu(_lin(sd, p + ".linear1", _ln(sd, p + ".norm_final", merged))))
    return merged + affine(ff, film(sd, p + ".film_final", t))


def sinusoidal_pos_emb(times, dim):
    half = dim // 2
    e = math.log(10000) / (half - 1)
    e = torch.exp(torch.arange(half) * -e)
    e = times[:, None].float() * e[None, :]
    return torch.cat((e.sin(), e.cos()), dim=-1)                   # utils.py:38-50


def forward(sd, cfg, x, cond_frame, cond_embed, times, keep_cond: bool):
    """GestureDecoder.forward (model.py:433-489) with cond_drop_prob 0 (keep_cond) or 1 (the null condition)."""
    face_x, body_x = batch_mask(x)
    face_c, body_c = batch_mask(cond_frame[:, None, :])
    t_len = x.shape[1]
    x_face = _lin(sd, "input_projection", torch.cat([face_x, face_c.repeat(1, t_len, 1)], dim=-1))
    x_body = _lin(sd, "input_projection", torch.cat([body_x, body_c.repeat(1, t_len, 1)], dim=-1))
    cond_tokens = _lin(sd, "cond_projection", cond_embed)
    for i in range(2):
        cond_tokens = encoder_layer(sd, f"cond_encoder.{i}", cond_tokens, cfg)
    if not keep_cond:
        cond_tokens = sd["null_cond_embed"].expand_as(cond_tokens)
    pooled = cond_tokens.mean(dim=-2)
    h = _ln(sd, "non_attn_cond_projection.0", pooled)
    cond_hidden = _lin(sd, "non_attn_cond_projection.3", F.silu(_lin(sd, "non_attn_cond_projection.1", h)))
    t_hidden = F.mish(_lin(sd, "time_mlp.1", sinusoidal_pos_emb(times, cfg.latent_dim).to(sd["time_mlp.1.weight"].dtype)))
    t = _lin(sd, "to_time_cond.0", t_hidden)
    t_tokens = _lin(sd, "to_time_tokens.0", t_hidden).reshape(x.shape[0], 2, cfg.latent_dim)
    t = t + (cond_hidden if keep_cond else sd["null_cond_hidden"].expand_as(cond_hidden))
    c = _ln(sd, "norm_cond", torch.cat((cond_tokens, t_tokens), dim=-2))
    out = x_face
    for i in range(cfg.num_layers):                                # DecoderLayerStack: x = layer(x, y, cond, t); y never changes
        out = decoder_layer(sd, f"seqTransDecoder.stack.{i}", out, x_body, c, t, cfg)
    return _lin(sd, "final_layer", out)


def guided_forward(sd, cfg, x, cond_frame, cond_embed, times, weight):
    unc = forward(sd, cfg, x, cond_frame, cond_embed, times, keep_cond=False)
    cond = forward(sd, cfg, x, cond_frame, cond_embed, times, keep_cond=True)
    return unc + (cond - unc) * weight                             # model.py:419-423


def cosine_alphas_cumprod(n=1000, s=8e-3):
    """make_beta_schedule('cosine') -> alphas_cumprod as GestureDiffusion.__init__ builds them (utils.py:76-84,
    diffusion.py:60-64): float64 betas clipped to 0.999, cast to float32, then cumprod."""
    ts = torch.arange(n + 1, dtype=torch.float64) / n + s
    a = torch.cos(ts / (1 + s) * math.pi / 2).pow(2)
    a = a / a[0]
    betas = (1 - a[1:] / a[:-1]).clamp(0, 0.999)
    return torch.cumprod(1.0 - betas.float(), dim=0)


def ddim_time_pairs(cfg):
    times = torch.linspace(-1, cfg.n_timestep - 1, steps=cfg.sampling_timesteps + 1)
    times = list(reversed(times.int().tolist()))
    return list(zip(times[:-1], times[1:]))                        # diffusion.py:245-247


def ddim_sample(sd, cfg, cond_frame, cond_embed, noises, trajectory=None):
    """GestureDiffusion.ddim_sample (diffusion.py:241-274).  `noises`: the normal draws in the order the reference makes them --
    noises[0] = the initial x, noises[1 + i] = the step noise of DDIM step i (only steps with time_next >= 0 draw one)."""
    ac = cosine_alphas_cumprod(cfg.n_timestep)
    x = noises[0]
    it = iter(noises[1:])
    for time, time_next in ddim_time_pairs(cfg):
        tc = torch.full((x.shape[0],), time, dtype=torch.long)
        x_start = guided_forward(sd, cfg, x, cond_frame, cond_embed, tc, cfg.guidance_weight).clamp(-1.0, 1.0)
        pred_noise = ((1.0 / ac[time]).sqrt() * x - x_start) / (1.0 / ac[time] - 1).sqrt()   # predict_noise_from_start :143-147
        if time_next < 0:
            x = x_start
        else:
            alpha, alpha_next = ac[time], ac[time_next]
            sigma = cfg.eta * ((1 - alpha / alpha_next) * (1 - alpha_next) / (1 - alpha)).sqrt()
            c = (1 - alpha_next - sigma ** 2).sqrt()
            x = x_start * alpha_next.sqrt() + c * pred_noise + sigma * next(it)
        if trajectory is not None:
            trajectory.append(x.clone())
    return x
