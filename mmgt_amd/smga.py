"""SMGA (Stage 1 of MMGT): the audio -> pose diffusion sampler on MI355X (BASELINE config 3).

Host-side mirror of the reference's `GestureDecoder` (src/audio2pose_model/model.py:324-490), `GestureDiffusion.ddim_sample`
(src/audio2pose_model/diffusion.py:241-274) and the `SMGA` wrapper's `render_sample` (src/audio2pose_model/SMGA.py:48-108,
301-322; scripts/audio2vid.py:198-200,324-348), with the reference's state-dict key names.  All arithmetic runs in
libmmgt_hip.so: the Linear layers on the GEMM kernels (GELU / Mish epilogues), attention (8 heads x 64) on the flash kernel,
LayerNorm, and the element-wise glue of csrc/smga.hip (rotary rotation, FiLM residual, token mean, guided DDIM update).

What the reference recomputes in each of its 100 forwards per 3.2-second slice and this implementation computes once per
slice (results identical): the condition encoder (two transformer layers over the 80 audio tokens), its pooled FiLM vector,
the LayerNorm'd condition tokens, the projections of the masked condition frame.  The two guidance passes (null condition /
audio condition, model.py:419-423) run as ONE batch of 2 B sequences.
"""
import math
from typing import Dict, List, Optional

import torch

from . import hip
from .packing import pad_cols, round_up

FACE_LO, FACE_HI = 24 * 3, 92 * 3            # channels of the face key points 24..91 (model.py:21-32)


def cosine_alphas_cumprod(n_timestep=1000, s=8e-3):
    """make_beta_schedule('cosine') + GestureDiffusion.__init__ (utils.py:76-84, diffusion.py:60-64)."""
    ts = torch.arange(n_timestep + 1, dtype=torch.float64) / n_timestep + s
    a = torch.cos(ts / (1 + s) * math.pi / 2).pow(2)
    a = a / a[0]
    betas = (1 - a[1:] / a[:-1]).clamp(0, 0.999)
    return torch.cumprod(1.0 - betas.float(), dim=0)


def smga_spec(nfeats=402, seq_len=80, latent_dim=512, ff_size=1024, num_layers=8, cond_feature_dim=1059) -> Dict[str, tuple]:
    """{state-dict key: shape} of the reference's GestureDecoder (src/audio2pose_model/model.py:324-431; 473 keys at the SMGA
    configuration) in registration order -- what `load_state_dict` expects and what scripts/audio2vid.py --synthetic fills."""
    d, ff = latent_dim, ff_size
    spec: Dict[str, tuple] = {}

    def lin(p, n, k):
        spec[p + ".weight"], spec[p + ".bias"] = (n, k), (n,)

    def ln(p):
        spec[p + ".weight"], spec[p + ".bias"] = (d,), (d,)

    def mha(p):
        spec[p + ".in_proj_weight"], spec[p + ".in_proj_bias"] = (3 * d, d), (3 * d,)
        lin(p + ".out_proj", d, d)

    spec["null_cond_embed"], spec["null_cond_hidden"] = (1, seq_len, d), (1, d)
    spec["rotary.freqs"] = (d // 2,)
    lin("time_mlp.1", 4 * d, d)
    lin("to_time_cond.0", d, 4 * d)
    lin("to_time_tokens.0", 2 * d, 4 * d)
    ln("norm_cond")
    lin("input_projection", d, 2 * nfeats)
    for i in range(2):
        p = f"cond_encoder.{i}"
        mha(p + ".self_attn")
        lin(p + ".linear1", ff, d)
        lin(p + ".linear2", d, ff)
        ln(p + ".norm1")
        ln(p + ".norm2")
        spec[p + ".rotary.freqs"] = (d // 2,)
    lin("cond_projection", d, cond_feature_dim)
    ln("non_attn_cond_projection.0")
    lin("non_attn_cond_projection.1", d, d)
    lin("non_attn_cond_projection.3", d, d)
    for i in range(num_layers):
        p = f"seqTransDecoder.stack.{i}"
        for a in ("face_self_attn", "face_cross_attn", "body_self_attn", "body_cross_attn", "self_attn"):
            mha(f"{p}.{a}")
        lin(p + ".linear1", ff, d)
        lin(p + ".linear2", d, ff)
        for n in ("face_1", "face_2", "face_3", "body_1", "body_2", "body_3", "final"):
            ln(f"{p}.norm_{n}")
        for n in ("face_1", "face_2", "face_3", "body_1", "body_2", "body_3", "final"):
            lin(f"{p}.film_{n}.block.1", 2 * d, d)
        spec[p + ".rotary.freqs"] = (d // 2,)
    lin("final_layer", nfeats, d)
    return spec


class GestureDecoder:
    def __init__(self, nfeats=402, seq_len=80, latent_dim=512, ff_size=1024, num_layers=8, num_heads=8, dropout=0.1,
                 cond_feature_dim=1059, activation=None, use_rotary=True, device="cuda", dtype=torch.bfloat16, **kwargs):
        if not use_rotary or latent_dim % (64 * num_heads // 8) or latent_dim // num_heads != 64 or nfeats != 402:
            raise ValueError("GestureDecoder: the HIP path covers the SMGA configuration (402 features, rotary, head_dim 64)")
        self.nfeats, self.seq_len, self.d, self.ff, self.layers, self.heads = nfeats, seq_len, latent_dim, ff_size, num_layers, num_heads
        self.cond_dim = cond_feature_dim
        self._device, self._dtype = torch.device(device), dtype
        hip.dtype_code(dtype)
        self.w: Dict[str, torch.Tensor] = {}
        self._loaded = False
        self._prep = None
        self.weights_generation = 0     # bumped by load_state_dict: captured HIP graphs hold raw pointers into self.w

    @property
    def device(self):
        return self._device

    @property
    def dtype(self):
        return self._dtype

    def eval(self):
        return self

    def to(self, *a, **k):
        return self

    # ------------------------------------------------------------------------------------------ weights
    def _t(self, x):
        return x.to(device=self._device, dtype=self._dtype).contiguous()

    def _f(self, x):
        return x.to(device=self._device, dtype=torch.float32).contiguous()

    def load_state_dict(self, sd, strict=True):
        w, d = self.w, self.d
        need = ["input_projection.weight", "cond_projection.weight", "final_layer.weight", "null_cond_embed", "null_cond_hidden",
                f"seqTransDecoder.stack.{self.layers - 1}.film_final.block.1.weight"]
        missing = [k for k in need if k not in sd]
        if missing:
            raise RuntimeError(f"load_state_dict: missing {missing}")

        def lin(p, key=None, kpad=None):
            wt = sd[p + ".weight"]
            w[(key or p) + ".w"] = self._t(pad_cols(wt, kpad) if kpad else wt)
            w[(key or p) + ".bias"] = self._f(sd[p + ".bias"])

        def norm(p):
            w[p + ".g"], w[p + ".b"] = self._f(sd[p + ".weight"]), self._f(sd[p + ".bias"])

        def mha(p):
            iw, ib = sd[p + ".in_proj_weight"], sd[p + ".in_proj_bias"]
            w[p + ".qk.w"], w[p + ".qk.bias"] = self._t(iw[:2 * d]), self._f(ib[:2 * d])
            w[p + ".q.w"], w[p + ".q.bias"] = self._t(iw[:d]), self._f(ib[:d])
            w[p + ".k.w"], w[p + ".k.bias"] = self._t(iw[d:2 * d]), self._f(ib[d:2 * d])
            w[p + ".v.w"], w[p + ".v.bias"] = self._t(iw[2 * d:]), self._f(ib[2 * d:])
            lin(p + ".out_proj", p + ".o")

        # input projection of [masked x | masked condition frame] (model.py:439-448): masking = zeroed weight columns
        wi, bi = sd["input_projection.weight"].float(), sd["input_projection.bias"].float()
        face = torch.zeros(self.nfeats)
        face[FACE_LO:FACE_HI] = 1.0
        kp = round_up(self.nfeats, 64)
        for part, m in (("face", face), ("body", 1.0 - face)):
            w[f"in_{part}.w"] = self._t(pad_cols(wi[:, :self.nfeats] * m, kp))
            w[f"inc_{part}.w"] = self._t(pad_cols(wi[:, self.nfeats:] * m, kp))
        w["in.bias"] = self._f(bi)
        self._kp = kp
        lin("cond_projection", kpad=round_up(self.cond_dim, 64))
        for i in range(2):
            p = f"cond_encoder.{i}"
            mha(p + ".self_attn")
            lin(p + ".linear1")
            lin(p + ".linear2")
            norm(p + ".norm1")
            norm(p + ".norm2")
        norm("non_attn_cond_projection.0")
        lin("non_attn_cond_projection.1")
        lin("non_attn_cond_projection.3")
        lin("time_mlp.1")
        lin("to_time_cond.0")
        lin("to_time_tokens.0")
        norm("norm_cond")
        w["null_cond_embed"] = self._t(sd["null_cond_embed"].reshape(-1, d))
        w["null_cond_hidden"] = self._t(sd["null_cond_hidden"].reshape(1, d))
        films_w, films_b = [], []
        self._film_slot = {}
        for i in range(self.layers):
            p = f"seqTransDecoder.stack.{i}"
            for part in ("face", "body"):
                mha(f"{p}.{part}_self_attn")
                mha(f"{p}.{part}_cross_attn")
                norm(f"{p}.norm_{part}_1")
                norm(f"{p}.norm_{part}_2")
            norm(p + ".norm_final")
            lin(p + ".linear1")
            lin(p + ".linear2")
            for f in ("film_face_1", "film_face_2", "film_body_1", "film_body_2", "film_final"):   # the *_3 FiLMs are never called
                self._film_slot[f"{p}.{f}"] = len(films_w) * 2 * d
                films_w.append(sd[f"{p}.{f}.block.1.weight"])
                films_b.append(sd[f"{p}.{f}.block.1.bias"])
        w["film_all.w"] = self._t(torch.cat(films_w, 0))          # every DenseFiLM Linear of the decoder in one GEMM per step
        w["film_all.bias"] = self._f(torch.cat(films_b, 0))
        lin("final_layer")
        # tables: rotary angles (positions 0 .. seq_len + 1: 80 condition tokens + 2 time tokens), timestep sinusoids
        freqs = 1.0 / (10000 ** (torch.arange(0, d, 2)[: d // 2].float() / d))
        ang = torch.arange(self.seq_len + 2).float()[:, None] * freqs[None]
        w["rot"] = self._f(torch.stack((ang.cos(), ang.sin()), dim=-1))
        half = d // 2
        e = torch.exp(torch.arange(half) * -(math.log(10000) / (half - 1)))
        tt = torch.arange(1000).float()[:, None] * e[None]
        w["time_table"] = self._t(torch.cat((tt.sin(), tt.cos()), dim=-1))
        self._loaded = True
        self._prep = None
        self.weights_generation += 1
        return [], []

    # ------------------------------------------------------------------------------------------ blocks
    def _lin(self, p, x, **kw):
        return hip.gemm(x, self.w[p + ".w"], self.w[p + ".bias"], **kw)

    def _ln(self, p, x):
        return hip.layernorm(x, self.w[p + ".g"], self.w[p + ".b"], 1e-5)

    def _attn(self, q, k, v, nb, nq, nk, q_ld, k_ld, v_ld):
        o = torch.empty((nb * nq, self.d), device=self._device, dtype=self._dtype)
        hip.attention(q, k, v, o, batch=nb, heads=self.heads, hd=64, nq=nq, nk=nk, scale=64 ** -0.5, q_str=(nq * q_ld, 0, q_ld),
                      k_str=(nk * k_ld, 0, k_ld), v_str=(nk * v_ld, 0, v_ld), o_str=(nq * self.d, 0, self.d))
        return o

    def _self_attn(self, p, n1, nb, t):
        d = self.d
        qk = self._lin(p + ".qk", hip.rotary(n1, self.w["rot"], t))              # rotary on the q / k input only (model.py:121-131)
        v = self._lin(p + ".v", n1)
        return self._lin(p + ".o", self._attn(qk, qk[:, d:], v, nb, t, t, 2 * d, 2 * d, d))

    def _encoder_layer(self, p, x, nb, t):
        """TransformerEncoderLayer, norm_first (model.py:102-135)."""
        d = self.d
        n1 = self._ln(p + ".norm1", x)
        qk = self._lin(p + ".self_attn.qk", hip.rotary(n1, self.w["rot"], t))
        v = self._lin(p + ".self_attn.v", n1)
        x = self._lin(p + ".self_attn.o", self._attn(qk, qk[:, d:], v, nb, t, t, 2 * d, 2 * d, d), residual=x)
        h = self._lin(p + ".linear1", self._ln(p + ".norm2", x), act=hip.ACT_GELU)
        return self._lin(p + ".linear2", h, residual=x)

    # ------------------------------------------------------------------------------------------ step-invariant part
    def prepare(self, cond_frame, cond_embed):
        """Everything of forward() that depends on (cond_frame, cond_embed) only -- the reference recomputes it in every forward
        (model.py:439-461): masked condition-frame projections, condition encoder, pooled FiLM vector, LayerNorm'd memory tokens
        for the null and the audio condition, batched as [null rows | audio rows]."""
        dev, dt, d, t = self._device, self._dtype, self.d, self.seq_len
        b = cond_frame.shape[0]
        if cond_embed.shape[1] != t:
            raise RuntimeError(f"cond_embed has {cond_embed.shape[1]} tokens, the model is built for {t}")
        cf = pad_cols(cond_frame.to(dev).float(), self._kp).to(dt).contiguous()
        inc = {part: hip.gemm(cf, self.w[f"inc_{part}.w"], self.w["in.bias"]).float().repeat(2, 1).contiguous() for part in ("face", "body")}
        ce = pad_cols(cond_embed.to(dev).float().reshape(b * t, -1), self.w["cond_projection.w"].shape[1]).to(dt).contiguous()
        tok = self._lin("cond_projection", ce)
        for i in range(2):
            tok = self._encoder_layer(f"cond_encoder.{i}", tok, b, t)
        null_tok = self.w["null_cond_embed"].repeat(b, 1)                       # torch.where(keep_mask, tokens, null_cond_embed), :455-457
        tokens = torch.cat([null_tok, tok], 0).contiguous()                     # (2b * t, d): [null | cond]
        pooled = hip.mean_tokens(tok.view(b, t, d)).to(dt)
        h = self._lin("non_attn_cond_projection.1", self._ln("non_attn_cond_projection.0", pooled), act=hip.ACT_SILU)
        cond_hidden = self._lin("non_attn_cond_projection.3", h)
        hidden = torch.cat([self.w["null_cond_hidden"].repeat(b, 1), cond_hidden], 0).contiguous()   # :471-473
        mem = torch.empty((2 * b, t + 2, d), device=dev, dtype=dt)
        mem[:, :t] = self._ln("norm_cond", tokens).view(2 * b, t, d)            # norm_cond is per token: the 80 condition tokens once
        # `held` keeps the keyed tensors alive, so the allocator cannot hand their addresses to a different condition
        self._prep = dict(b=b, inc=inc, hidden=hidden, mem=mem, key=self._prep_key(cond_frame, cond_embed),
                          held=(cond_frame, cond_embed))
        return self._prep

    # ------------------------------------------------------------------------------------------ forward
    def _forward2(self, x, times, prep):
        """Both guidance passes as one batch: returns (2b * t, nfeats) rows [null-condition pass | audio-condition pass]."""
        dev, dt, d, t = self._device, self._dtype, self.d, self.seq_len
        b = prep["b"]
        nb = 2 * b
        xr = pad_cols(x.to(dev).float().reshape(b * t, -1), self._kp).to(dt)
        x2 = torch.cat([xr, xr], 0).contiguous()
        x_face = hip.gemm(x2, self.w["in_face.w"], None, bias2=prep["inc"]["face"], bias2_rows=t)
        x_body = hip.gemm(x2, self.w["in_body.w"], None, bias2=prep["inc"]["body"], bias2_rows=t)
        ti = torch.as_tensor(times, device=dev).reshape(-1).long().repeat(2)
        t_hidden = self._lin("time_mlp.1", self.w["time_table"][ti].contiguous(), act=hip.ACT_MISH)
        tvec = self._lin("to_time_cond.0", t_hidden, residual=prep["hidden"])     # t += cond_hidden / null_cond_hidden
        mem = prep["mem"]
        mem[:, t:] = self._ln("norm_cond", self._lin("to_time_tokens.0", t_hidden).view(nb * 2, d)).view(nb, 2, d)
        mem2 = mem.view(nb * (t + 2), d)
        mem_rot = hip.rotary(mem2, self.w["rot"], t + 2)
        ss = hip.gemm(hip.activation(tvec, hip.ACT_MISH), self.w["film_all.w"], self.w["film_all.bias"]).float()   # (nb, layers*5*2d)

        def film(name):
            o = self._film_slot[name]
            return ss[:, o:o + 2 * d]

        def stream(p, part, xs):
            n1 = self._ln(f"{p}.norm_{part}_1", xs)
            xs = hip.film_residual(self._self_attn(f"{p}.{part}_self_attn", n1, nb, t), film(f"{p}.film_{part}_1"), t, res=xs)
            n2 = self._ln(f"{p}.norm_{part}_2", xs)
            a = f"{p}.{part}_cross_attn"
            q = self._lin(a + ".q", hip.rotary(n2, self.w["rot"], t))
            k = self._lin(a + ".k", mem_rot)
            v = self._lin(a + ".v", mem2)
            o = self._lin(a + ".o", self._attn(q, k, v, nb, t, t + 2, d, d, d))
            return hip.film_residual(o, film(f"{p}.film_{part}_2"), t, res=xs)

        out = x_face
        for i in range(self.layers):                        # DecoderLayerStack: x = layer(x, y, cond, t); the body input never changes
            p = f"seqTransDecoder.stack.{i}"
            sf, sb = stream(p, "face", out), stream(p, "body", x_body)
            merged = hip.film_residual(sb, self._zero_ss(nb), t, res=sf)               # face + body (model.py:229): (0 + 1) * sb + 0 + sf
            ff = self._lin(p + ".linear2", self._lin(p + ".linear1", self._ln(p + ".norm_final", merged), act=hip.ACT_GELU))
            out = hip.film_residual(ff, film(p + ".film_final"), t, res=merged)        # :232-234
        return self._lin("final_layer", out)

    def _zero_ss(self, nb):
        z = self.w.get("zero_ss")
        if z is None or z.shape[0] < nb:
            z = self.w["zero_ss"] = torch.zeros((nb, 2 * self.d), device=self._device, dtype=torch.float32)
        return z[:nb]

    @staticmethod
    def _prep_key(cond_frame, cond_embed):
        # address alone can alias (a freed tensor's block is recycled; an in-place update keeps it): identity of the storage
        # AND its version counter, shape and device -- the cached state holds a reference to both tensors (prepare())
        return tuple((t.data_ptr(), t._version, tuple(t.shape), str(t.device), t.dtype) for t in (cond_frame, cond_embed))

    def _prep_for(self, cond_frame, cond_embed):
        key = self._prep_key(cond_frame, cond_embed)
        if self._prep is None or self._prep["key"] != key:
            self.prepare(cond_frame, cond_embed)
        return self._prep

    def forward(self, x, cond_frame, cond_embed, times, cond_drop_prob: float = 0.0):
        """GestureDecoder.forward (model.py:433-489) for cond_drop_prob in {0, 1} (the two values inference uses)."""
        if cond_drop_prob not in (0, 0.0, 1, 1.0):
            raise NotImplementedError("stochastic condition dropout is a training feature (cond_drop_prob must be 0 or 1)")
        if not self._loaded:
            raise RuntimeError("GestureDecoder.forward before load_state_dict")
        if not torch.as_tensor(x).is_cuda:
            raise RuntimeError("mmgt_amd.GestureDecoder runs on the GPU only (no CPU path exists)")
        b, t = x.shape[0], x.shape[1]
        out = self._forward2(x, times, self._prep_for(cond_frame, cond_embed)).view(2, b, t, self.nfeats)
        return out[0 if cond_drop_prob else 1].float()

    __call__ = forward

    def guided_forward(self, x, cond_frame, cond_embed, times, guidance_weight):
        b, t = x.shape[0], x.shape[1]
        out = self._forward2(x, times, self._prep_for(cond_frame, cond_embed)).view(2, b, t, self.nfeats).float()
        return out[0] + (out[1] - out[0]) * guidance_weight                       # model.py:419-423


class GestureDiffusion:
    """Sampling half of src/audio2pose_model/diffusion.py (the trainer half is out of scope): x0-prediction, cosine schedule,
    50-step DDIM with eta = 1 and x0 clipped to [-1, 1]."""

    def __init__(self, model, horizon, repr_dim, n_timestep=1000, schedule="cosine", loss_type="l2", clip_denoised=True,
                 predict_epsilon=False, guidance_weight=2, use_p2=False, cond_drop_prob=0.25):
        if schedule != "cosine" or predict_epsilon or not clip_denoised:
            raise NotImplementedError("the SMGA wrapper builds a cosine-schedule, x0-predicting, clipped sampler (SMGA.py:95-106)")
        self.model, self.horizon, self.transition_dim = model, horizon, repr_dim
        self.n_timestep, self.guidance_weight = int(n_timestep), float(guidance_weight)
        self.alphas_cumprod = cosine_alphas_cumprod(self.n_timestep)
        self.sampling_timesteps, self.eta = 50, 1.0

    def eval(self):
        return self

    def time_pairs(self):
        times = torch.linspace(-1, self.n_timestep - 1, steps=self.sampling_timesteps + 1)
        times = list(reversed(times.int().tolist()))
        return list(zip(times[:-1], times[1:]))

    def _step_coeffs(self, time, time_next):
        ac = self.alphas_cumprod
        a = float(ac[time])
        if time_next < 0:
            return math.sqrt(1.0 / a), math.sqrt(1.0 / a - 1), 0.0, 0.0, 0.0, True
        an = float(ac[time_next])
        sigma = self.eta * math.sqrt((1 - a / an) * (1 - an) / (1 - a))
        return math.sqrt(1.0 / a), math.sqrt(1.0 / a - 1), math.sqrt(an), math.sqrt(1 - an - sigma ** 2), sigma, False

    def _loop(self, x, prep, times_dev, noise_of):
        """The 50 DDIM steps (diffusion.py:252-274): decoder (both guidance passes batched) + guided x0-prediction update."""
        m = self.model
        n = x.shape[0] * x.shape[1]
        k = 0
        for i, (time, time_next) in enumerate(self.time_pairs()):
            pred = m._forward2(x, times_dev[i], prep)
            c0, c1, a_next_sqrt, c, sigma, last = self._step_coeffs(time, time_next)
            noise = None
            if not last:
                noise = noise_of(k)
                k += 1
            x = hip.smga_ddim_step(pred[:n], pred[n:], x, noise, self.guidance_weight, c0, c1, a_next_sqrt, c, sigma, last)
        return x

    @torch.no_grad()
    def ddim_sample(self, shape, cond_frame, cond, last_half=None, noises: Optional[List[torch.Tensor]] = None, generator=None,
                    **kwargs):
        """diffusion.py:241-274.  `noises` (parity tests): the normal draws in the reference's order -- the initial x, then one per
        DDIM step that draws; otherwise they come from torch.randn on the device (`generator` optional).

        The loop is ~5000 small launches (50 steps x ~100 kernels over 80 x 512 activations): launch-bound.  Without injected noises it
        is captured ONCE per shape into a HIP graph -- condition preparation, 50 decoder passes and updates -- over static input / noise
        buffers and replayed (same kernels, same draw order: bitwise the eager result; hip.tune("smga_graph", 0) keeps the eager loop)."""
        m = self.model
        dev = m.device
        b = shape[0]
        pairs = self.time_pairs()
        ndraw = sum(1 for _, tn in pairs if tn >= 0)
        if noises is not None or torch.device(dev).type != "cuda" or not hip.tune_get("smga_graph"):
            prep = m.prepare(cond_frame, cond)
            it = iter(noises) if noises is not None else None
            draw = (lambda: next(it).to(dev).float().contiguous()) if noises is not None else \
                (lambda: torch.randn(shape, device=dev, generator=generator))
            x = draw()
            times_dev = [torch.full((b,), t, device=dev, dtype=torch.long) for t, _ in pairs]
            return self._loop(x, prep, times_dev, lambda k: draw())
        # A captured graph bakes in the raw pointers of the model's weight tensors and the host constants of the loop (guidance
        # weight, eta, the time table): all of them are part of the key, a weight reload drops every older entry (its tensors may
        # be freed), and the entry keeps the weight dict it was captured over alive.
        key = (tuple(shape), tuple(cond_frame.shape), tuple(cond.shape), m.weights_generation, str(m.dtype), self.guidance_weight,
               self.eta, self.sampling_timesteps, self.n_timestep)
        if not hasattr(self, "_graphs"):
            self._graphs = {}
        for k_old in [k for k in self._graphs if k[3] != m.weights_generation]:
            del self._graphs[k_old]
        g = self._graphs.get(key)
        if g is None:
            st = dict(cf=torch.zeros(tuple(cond_frame.shape), device=dev, dtype=torch.float32),
                      cond=torch.zeros(tuple(cond.shape), device=dev, dtype=torch.float32),
                      x0=torch.zeros(shape, device=dev, dtype=torch.float32),
                      noise=torch.zeros((max(ndraw, 1),) + tuple(shape), device=dev, dtype=torch.float32),
                      times=[torch.full((b,), t, device=dev, dtype=torch.long) for t, _ in pairs])

            def body():
                prep = m.prepare(st["cf"], st["cond"])
                return self._loop(st["x0"], prep, st["times"], lambda k: st["noise"][k])
            side = torch.cuda.Stream(device=dev)
            side.wait_stream(torch.cuda.current_stream(dev))
            with torch.cuda.stream(side):            # warm-up outside the capture: library attributes, allocator pools
                body()
            torch.cuda.current_stream(dev).wait_stream(side)
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                st["out"] = body()
            st["weights"] = dict(m.w)                    # the captured pointers stay valid while this entry lives
            g = self._graphs[key] = (graph, st)
        graph, st = g
        st["cf"].copy_(cond_frame.to(dev).float())
        st["cond"].copy_(cond.to(dev).float())
        st["x0"].copy_(torch.randn(shape, device=dev, generator=generator))          # the reference's draw order: x first, then one per step
        for k in range(ndraw):
            st["noise"][k].copy_(torch.randn(shape, device=dev, generator=generator))
        graph.replay()
        return st["out"].clone()

    def render_sample(self, shape, cond_frame, cond, epoch=None, render_out=None, last_half=None, mode="normal", **kwargs):
        if mode != "normal":
            raise NotImplementedError("only the 'normal' (DDIM) mode is used by scripts/audio2vid.py")
        return self.ddim_sample(tuple(shape), cond_frame, cond, last_half=last_half, **kwargs).detach().cpu()


class SMGA:
    """The wrapper scripts/audio2vid.py instantiates (src/audio2pose_model/SMGA.py:48-108): builds the 402-feature / 80-frame
    GestureDecoder + GestureDiffusion pair and exposes render_sample(cond_frame, cond, last_half, mode)."""

    def __init__(self, feature_type="wavlm", checkpoint_path="", EMA=True, device="cuda", dtype=torch.bfloat16, state_dict=None):
        self.repr_dim, self.horizon = 402, int(3.2 * 25)
        feature_dim = 1024 + 35 if feature_type == "wavlm" else 35
        self.model = GestureDecoder(nfeats=self.repr_dim, seq_len=self.horizon, latent_dim=512, ff_size=1024, num_layers=8,
                                    num_heads=8, dropout=0.1, cond_feature_dim=feature_dim, device=device, dtype=dtype)
        self.diffusion = GestureDiffusion(self.model, self.horizon, self.repr_dim, schedule="cosine", n_timestep=1000,
                                          predict_epsilon=False, loss_type="l2", use_p2=False, cond_drop_prob=0.25,
                                          guidance_weight=2)
        if checkpoint_path:
            ck = torch.load(checkpoint_path, map_location="cpu", weights_only=True)
            state_dict = ck["ema_state_dict" if EMA else "model_state_dict"]
        if state_dict is not None:
            self.model.load_state_dict(state_dict)

    def eval(self):
        return self

    def render_sample(self, cond_frame, cond, last_half=None, mode="normal", **kwargs):
        cond_frame = torch.as_tensor(cond_frame).float().reshape(-1, self.repr_dim)
        cond = torch.as_tensor(cond).float()
        if cond.dim() == 2:
            cond = cond[None]
        shape = (cond_frame.shape[0], self.horizon, self.repr_dim)
        return self.diffusion.render_sample(shape, cond_frame, cond, epoch=None, render_out=None, last_half=last_half, mode=mode,
                                            **kwargs).to(self.model.device)
