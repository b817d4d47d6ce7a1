"""wav2vec2 audio feature extractor on MI355X: the `audio_encoder` of the reference's AudioProcessor.

The reference builds `Wav2VecModel.from_pretrained(wav2vec_model_path)` (src/dataset/audio_processor.py:49-52; src/models/wav2vec.py:
a transformers Wav2Vec2Model whose conv features are linearly interpolated to `seq_len` video frames before the transformer) and
calls `audio_encoder(audio_feature, seq_len=seq_len, output_hidden_states=True)` once per clip; the 12 layer outputs stacked as
(frames, 12, 768) are the `audio_emb` that `process_audio_emb` / AudioProjModel consume (audio_processor.py:117-126).  This class
keeps that call and the transformers state-dict key names (wav2vec2-base-960h loads by name, with either spelling of the positional
conv's weight norm: `weight_g` / `weight_v` of the checkpoint or torch's `parametrizations.weight.original0/1`) and runs on the HIP
kernels of libmmgt_hip.so:

  conv feature extractor   7 Conv1d layers (kernels 10,3,3,3,3,2,2; strides 5,2,2,2,2,2,2; no bias) as GEMMs over strided views of the
                           channels-last (T, 512) signal -- a 1-D convolution's patches ARE overlapping rows of that tensor --
                           with the exact GELU in the epilogue; layer 0's GroupNorm(512, 512) over time + GELU: csrc/wav2vec.hip
  linear_interpolation     csrc/wav2vec.hip (align_corners=True)
  feature projection       LayerNorm(512) -> Linear(512, 768)
  positional conv          Conv1d(768, 768, kernel 128, padding 64, groups 16) + SamePad + GELU: 16 GEMMs (K = 128 x 48) over strided
                           views of the group-major padded signal; the weight norm w = g v / |v| is folded once at load time
  12 encoder layers        post-LayerNorm (do_stable_layer_norm = False): q|k|v GEMM, flash attention (12 heads x 64), out-proj +
                           residual, LayerNorm, GELU feed-forward, + residual, LayerNorm
transformers is an un-vendored dependency of the reference (requirements.txt:207); the oracle (oracle/wav2vec_ref.py) is pinned by
goldens produced by the reference's own Wav2VecModel class on the transformers build of the build container
(tools/refgen/gen_wav2vec_golden.py).  Vocal separation, resampling (librosa) and file decoding stay outside.
"""
from collections import OrderedDict

import torch

from . import hip

CONV_KERNEL = (10, 3, 3, 3, 3, 2, 2)
CONV_STRIDE = (5, 2, 2, 2, 2, 2, 2)


def wav2vec_spec(hidden=768, layers=12, intermediate=3072, conv_dim=512, pos_kernel=128, pos_groups=16, weight_norm_keys="checkpoint"):
    """{key: shape} of Wav2Vec2Model.state_dict() at the wav2vec2-base geometry.  weight_norm_keys: "checkpoint" (weight_g / weight_v,
    what facebook/wav2vec2-base-960h and the reference's transformers 4.30 hold) or "parametrized" (torch >= 2.1 modules)."""
    s = OrderedDict()
    s["masked_spec_embed"] = (hidden,)
    for i, k in enumerate(CONV_KERNEL):
        s[f"feature_extractor.conv_layers.{i}.conv.weight"] = (conv_dim, 1 if i == 0 else conv_dim, k)
        if i == 0:
            s["feature_extractor.conv_layers.0.layer_norm.weight"] = (conv_dim,)
            s["feature_extractor.conv_layers.0.layer_norm.bias"] = (conv_dim,)
    s["feature_projection.layer_norm.weight"], s["feature_projection.layer_norm.bias"] = (conv_dim,), (conv_dim,)
    s["feature_projection.projection.weight"], s["feature_projection.projection.bias"] = (hidden, conv_dim), (hidden,)
    s["encoder.pos_conv_embed.conv.bias"] = (hidden,)
    g, v = ("weight_g", "weight_v") if weight_norm_keys == "checkpoint" else ("parametrizations.weight.original0", "parametrizations.weight.original1")
    s[f"encoder.pos_conv_embed.conv.{g}"] = (1, 1, pos_kernel)
    s[f"encoder.pos_conv_embed.conv.{v}"] = (hidden, hidden // pos_groups, pos_kernel)
    s["encoder.layer_norm.weight"], s["encoder.layer_norm.bias"] = (hidden,), (hidden,)
    for i in range(layers):
        p = f"encoder.layers.{i}."
        for n in ("k_proj", "v_proj", "q_proj", "out_proj"):
            s[p + f"attention.{n}.weight"], s[p + f"attention.{n}.bias"] = (hidden, hidden), (hidden,)
        s[p + "layer_norm.weight"], s[p + "layer_norm.bias"] = (hidden,), (hidden,)
        s[p + "feed_forward.intermediate_dense.weight"], s[p + "feed_forward.intermediate_dense.bias"] = (intermediate, hidden), (intermediate,)
        s[p + "feed_forward.output_dense.weight"], s[p + "feed_forward.output_dense.bias"] = (hidden, intermediate), (hidden,)
        s[p + "final_layer_norm.weight"], s[p + "final_layer_norm.bias"] = (hidden,), (hidden,)
    return s


class Wav2VecOutput:
    def __init__(self, last_hidden_state, hidden_states):
        self.last_hidden_state, self.hidden_states = last_hidden_state, hidden_states

    def __len__(self):
        return 2


class Wav2VecModel:
    def __init__(self, device="cuda", dtype=torch.bfloat16, hidden_size=768, num_hidden_layers=12, num_attention_heads=12,
                 intermediate_size=3072, conv_dim=512, num_conv_pos_embeddings=128, num_conv_pos_embedding_groups=16, layer_norm_eps=1e-5):
        if hidden_size // num_attention_heads != 64 or conv_dim % 64 or hidden_size % (8 * num_conv_pos_embedding_groups):
            raise ValueError("Wav2VecModel: the HIP path covers the wav2vec2-base geometry (head_dim 64, 512 conv channels, 16 groups)")
        self._device, self._dtype = torch.device(device), dtype
        hip.dtype_code(dtype)
        self.hidden, self.layers, self.heads, self.inter, self.cd = hidden_size, num_hidden_layers, num_attention_heads, intermediate_size, conv_dim
        self.pk, self.pg, self.eps = num_conv_pos_embeddings, num_conv_pos_embedding_groups, layer_norm_eps
        self.w = {}
        self._loaded = False

    dtype = property(lambda self: self._dtype)
    device = property(lambda self: self._device)

    def to(self, *a, **k):
        return self

    def eval(self):
        return self

    def _t(self, x):
        return x.to(device=self._device, dtype=self._dtype).contiguous()

    def _f(self, x):
        return x.to(device=self._device, dtype=torch.float32).contiguous()

    def load_state_dict(self, sd, strict=True):
        pre = "encoder.pos_conv_embed.conv."
        style = "checkpoint" if (pre + "weight_g") in sd else "parametrized"
        spec = wav2vec_spec(self.hidden, self.layers, self.inter, self.cd, self.pk, self.pg, style)
        missing = [k for k in spec if k not in sd and k != "masked_spec_embed"]
        if missing:
            raise RuntimeError(f"Wav2VecModel.load_state_dict: missing {len(missing)} keys, e.g. {missing[:3]}")
        for k, shape in spec.items():
            if k in sd and tuple(sd[k].shape) != tuple(shape):
                raise RuntimeError(f"shape mismatch for {k}: {tuple(sd[k].shape)} vs {shape}")
        w = self.w
        for i, k in enumerate(CONV_KERNEL):
            cw = sd[f"feature_extractor.conv_layers.{i}.conv.weight"].float()                 # (512, cin, k)
            if i == 0:
                w0 = torch.zeros((self.cd, 64))                                                # K = 10 taps, padded to the GEMM's 64
                w0[:, :k] = cw[:, 0]
                w["c0.w"] = self._t(w0)
                w["c0.g"], w["c0.b"] = self._f(sd["feature_extractor.conv_layers.0.layer_norm.weight"]), self._f(sd["feature_extractor.conv_layers.0.layer_norm.bias"])
            else:
                w[f"c{i}.w"] = self._t(cw.permute(0, 2, 1).reshape(self.cd, -1))               # [cout][tap][cin]: a patch = k consecutive rows
        w["fp.g"], w["fp.b"] = self._f(sd["feature_projection.layer_norm.weight"]), self._f(sd["feature_projection.layer_norm.bias"])
        w["fp.w"], w["fp.bias"] = self._t(sd["feature_projection.projection.weight"]), self._f(sd["feature_projection.projection.bias"])
        g_key, v_key = (pre + "weight_g", pre + "weight_v") if style == "checkpoint" else (pre + "parametrizations.weight.original0", pre + "parametrizations.weight.original1")
        v = sd[v_key].double()                                                                 # weight_norm(dim=2): w = g v / |v|, norm over (out, in) per tap
        pw = (sd[g_key].double() * v / v.norm(dim=(0, 1), keepdim=True)).float()               # (768, 48, 128)
        cg = self.hidden // self.pg
        w["pos.w"] = self._t(pw.view(self.pg, cg, cg, self.pk).permute(0, 1, 3, 2).reshape(self.pg, cg, self.pk * cg))   # [g][cout][tap][cin]
        w["pos.b"] = self._f(sd[pre + "bias"])
        w["enc.g"], w["enc.b"] = self._f(sd["encoder.layer_norm.weight"]), self._f(sd["encoder.layer_norm.bias"])
        for i in range(self.layers):
            p, q = f"encoder.layers.{i}.", f"l{i}."
            w[q + "qkv.w"] = self._t(torch.cat([sd[p + f"attention.{n}_proj.weight"] for n in ("q", "k", "v")], 0))
            w[q + "qkv.b"] = self._f(torch.cat([sd[p + f"attention.{n}_proj.bias"] for n in ("q", "k", "v")], 0))
            w[q + "o.w"], w[q + "o.b"] = self._t(sd[p + "attention.out_proj.weight"]), self._f(sd[p + "attention.out_proj.bias"])
            w[q + "ln1.g"], w[q + "ln1.b"] = self._f(sd[p + "layer_norm.weight"]), self._f(sd[p + "layer_norm.bias"])
            w[q + "fc1.w"], w[q + "fc1.b"] = self._t(sd[p + "feed_forward.intermediate_dense.weight"]), self._f(sd[p + "feed_forward.intermediate_dense.bias"])
            w[q + "fc2.w"], w[q + "fc2.b"] = self._t(sd[p + "feed_forward.output_dense.weight"]), self._f(sd[p + "feed_forward.output_dense.bias"])
            w[q + "ln2.g"], w[q + "ln2.b"] = self._f(sd[p + "final_layer_norm.weight"]), self._f(sd[p + "final_layer_norm.bias"])
        self._loaded = True
        return [], [k for k in sd if k not in spec]

    # ------------------------------------------------------------------------------------------ pieces
    def feature_extract(self, input_values, seq_len):
        """Wav2VecModel.feature_extract (src/models/wav2vec.py:112-127): (1, T) waveform -> (1, seq_len, 512)."""
        if not self._loaded:
            raise RuntimeError("Wav2VecModel.forward before load_state_dict")
        if not input_values.is_cuda:
            raise RuntimeError("mmgt_amd.Wav2VecModel runs on the GPU only (no CPU path exists)")
        if input_values.dim() != 2 or input_values.shape[0] != 1:
            raise RuntimeError("input_values must be (1, samples): the reference encodes one clip at a time (audio_processor.py:117)")
        x = input_values[0].to(self._device, torch.float32)
        k, st = CONV_KERNEL[0], CONV_STRIDE[0]
        t = (x.shape[0] - k) // st + 1
        a = torch.zeros((t, 64), device=self._device, dtype=self._dtype)      # layer 0 patches: 10 samples every 5 (host-side unfold, 1 channel)
        a[:, :k] = x.unfold(0, k, st).to(self._dtype)
        h = hip.gemm(a, self.w["c0.w"])
        h = hip.channel_norm_gelu(h, self.w["c0.g"], self.w["c0.b"], 1e-5)    # GroupNorm(512, 512) over time + GELU
        for i in range(1, len(CONV_KERNEL)):
            k, st = CONV_KERNEL[i], CONV_STRIDE[i]
            t = (h.shape[0] - k) // st + 1
            patches = h.as_strided((t, k * self.cd), (st * self.cd, 1))       # row r = rows st r .. st r + k - 1 of the signal, contiguous
            h = hip.gemm(patches, self.w[f"c{i}.w"], act=hip.ACT_GELU)
        return hip.lerp_rows(h, int(seq_len))[None]                           # linear_interpolation (:196-209)

    def encode(self, extract_features, output_hidden_states=True, **kw):
        """Wav2VecModel.encode (:129-194): feature projection, positional conv, 12 post-LN layers; .hidden_states[0] = the encoder input
        after its LayerNorm, [1..12] = the layer outputs, each (1, seq_len, 768) fp32."""
        H, heads, S = self.hidden, self.heads, extract_features.shape[1]
        f = extract_features[0].to(self._device, self._dtype).contiguous()
        x = hip.gemm(hip.layernorm(f, self.w["fp.g"], self.w["fp.b"], self.eps), self.w["fp.w"], self.w["fp.bias"])
        # positional conv embedding: Conv1d(768, 768, 128, padding=64, groups=16), last frame dropped (SamePad, even kernel), GELU.
        # group-major padded copy (16, S + 127, 48): a group's patch for frame s is rows s .. s + 127 of its (S + 127, 48) slab
        cg = H // self.pg
        xp = torch.zeros((self.pg, S + self.pk - 1, cg), device=self._device, dtype=self._dtype)
        xp[:, self.pk // 2: self.pk // 2 + S] = x.view(S, self.pg, cg).permute(1, 0, 2)
        xs = torch.empty((S, H), device=self._device, dtype=self._dtype)      # hidden_states + gelu(pos_conv(hidden_states)): residual epilogue
        for g in range(self.pg):
            patches = xp[g].as_strided((S, self.pk * cg), (cg, 1))
            sl = slice(g * cg, (g + 1) * cg)
            hip.gemm(patches, self.w["pos.w"][g], self.w["pos.b"][sl].contiguous(), act=hip.ACT_GELU, residual=x[:, sl], out=xs[:, sl])
        x = hip.layernorm(xs, self.w["enc.g"], self.w["enc.b"], self.eps)
        states = [x]
        hd = H // heads
        st = (S * 3 * H, 0, 3 * H)
        for i in range(self.layers):
            q = f"l{i}."
            qkv = hip.gemm(x, self.w[q + "qkv.w"], self.w[q + "qkv.b"])
            o = torch.empty((S, H), device=self._device, dtype=self._dtype)
            hip.attention(qkv, qkv[:, H:], qkv[:, 2 * H:], o, batch=1, heads=heads, hd=hd, nq=S, nk=S, scale=hd ** -0.5,
                          q_str=st, k_str=st, v_str=st, o_str=(S * H, 0, H))
            x = hip.layernorm(hip.gemm(o, self.w[q + "o.w"], self.w[q + "o.b"], residual=x), self.w[q + "ln1.g"], self.w[q + "ln1.b"], self.eps)
            f1 = hip.gemm(x, self.w[q + "fc1.w"], self.w[q + "fc1.b"], act=hip.ACT_GELU)
            x = hip.layernorm(hip.gemm(f1, self.w[q + "fc2.w"], self.w[q + "fc2.b"], residual=x), self.w[q + "ln2.g"], self.w[q + "ln2.b"], self.eps)
            states.append(x)
        return Wav2VecOutput(x.float()[None], tuple(s.float()[None] for s in states) if output_hidden_states else None)

    def forward(self, input_values, seq_len, output_hidden_states=True, **kw):
        """Wav2VecModel.forward (:42-110)."""
        return self.encode(self.feature_extract(input_values, seq_len), output_hidden_states=output_hidden_states)

    __call__ = forward

    def audio_emb(self, input_values, seq_len):
        """AudioProcessor.preprocess's tail (audio_processor.py:117-126): the 12 layer outputs as (seq_len, 12, 768) fp32."""
        out = self.forward(input_values, seq_len, output_hidden_states=True)
        return torch.stack(out.hidden_states[1:], dim=1).squeeze(0).permute(1, 0, 2).contiguous()
