"""CLIP vision tower with projection (the reference's `image_encoder`) on MI355X: the clip-embedding leg of the prologue.

The reference builds `transformers.CLIPVisionModelWithProjection.from_pretrained(image_encoder_path)` (scripts/pose2vid.py:
158-162) and calls `image_encoder(clip_image).image_embeds` once per clip (src/pipelines/pipeline_pose2vid_long.py:382-387).
This class keeps that call and the transformers state-dict key names (so `sd-image-variations-diffusers/image_encoder`
loads by name) and runs on the same HIP kernels as the UNet: one GEMM for the 14x14 patch embedding (the patches are a
strided view: host-side unfold, once per clip), LayerNorm, fused q|k|v GEMM, flash attention at head_dim 64, quick-GELU in
the fc1 epilogue, residuals in the out_proj / fc2 epilogues.  transformers is an un-vendored dependency of the reference
(requirements.txt:207); the oracle (oracle/clip_ref.py) is pinned by goldens generated from the transformers build
installed in the container (tools/refgen/gen_clip_golden.py).
"""
from collections import OrderedDict

import torch

from . import hip
from .packing import pad_cols, round_up


def clip_vision_spec(hidden=1024, intermediate=4096, layers=24, image_size=224, patch=14, projection=768):
    s = OrderedDict()
    v = "vision_model."
    ntok = (image_size // patch) ** 2 + 1
    s[v + "embeddings.class_embedding"] = (hidden,)
    s[v + "embeddings.patch_embedding.weight"] = (hidden, 3, patch, patch)
    s[v + "embeddings.position_embedding.weight"] = (ntok, hidden)
    for n in ("pre_layrnorm",):
        s[v + n + ".weight"] = (hidden,)
        s[v + n + ".bias"] = (hidden,)
    for i in range(layers):
        p = f"{v}encoder.layers.{i}."
        for n in ("k_proj", "v_proj", "q_proj", "out_proj"):
            s[p + f"self_attn.{n}.weight"] = (hidden, hidden)
            s[p + f"self_attn.{n}.bias"] = (hidden,)
        s[p + "layer_norm1.weight"] = (hidden,)
        s[p + "layer_norm1.bias"] = (hidden,)
        s[p + "mlp.fc1.weight"] = (intermediate, hidden)
        s[p + "mlp.fc1.bias"] = (intermediate,)
        s[p + "mlp.fc2.weight"] = (hidden, intermediate)
        s[p + "mlp.fc2.bias"] = (hidden,)
        s[p + "layer_norm2.weight"] = (hidden,)
        s[p + "layer_norm2.bias"] = (hidden,)
    s[v + "post_layernorm.weight"] = (hidden,)
    s[v + "post_layernorm.bias"] = (hidden,)
    s["visual_projection.weight"] = (projection, hidden)
    return s


class CLIPVisionOutput:
    def __init__(self, image_embeds, last_hidden_state):
        self.image_embeds = image_embeds
        self.last_hidden_state = last_hidden_state


class CLIPVisionModelWithProjection:
    def __init__(self, device="cuda", dtype=torch.bfloat16, hidden_size=1024, intermediate_size=4096, num_hidden_layers=24,
                 num_attention_heads=16, image_size=224, patch_size=14, projection_dim=768, layer_norm_eps=1e-5):
        if hidden_size % num_attention_heads or hidden_size // num_attention_heads not in (40, 64, 80, 160):
            raise ValueError("CLIPVisionModelWithProjection: head_dim must be one of 40, 64, 80, 160")
        if hidden_size % 64 or intermediate_size % 64 or projection_dim % 8:
            raise ValueError("CLIPVisionModelWithProjection: widths must be multiples of 64")
        self._device, self._dtype = torch.device(device), dtype
        hip.dtype_code(dtype)
        self.hidden, self.inter, self.layers, self.heads = hidden_size, intermediate_size, num_hidden_layers, num_attention_heads
        self.image_size, self.patch, self.proj, self.eps = image_size, patch_size, projection_dim, layer_norm_eps
        self.spec = clip_vision_spec(hidden_size, intermediate_size, num_hidden_layers, image_size, patch_size, projection_dim)
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
        missing = [k for k in self.spec if k not in sd]
        if missing:
            raise RuntimeError(f"CLIPVisionModelWithProjection.load_state_dict: missing {len(missing)} keys, e.g. {missing[:3]}")
        for k, shape in self.spec.items():
            if tuple(sd[k].shape) != tuple(shape):
                raise RuntimeError(f"shape mismatch for {k}: {tuple(sd[k].shape)} vs {shape}")
        w, v = self.w, "vision_model."
        pw = sd[v + "embeddings.patch_embedding.weight"].reshape(self.hidden, -1)            # (H, 3 * p * p), no bias
        w["patch.w"] = self._t(pad_cols(pw, round_up(pw.shape[1], 64)))
        w["cls"] = self._f(sd[v + "embeddings.class_embedding"])
        w["pos"] = self._f(sd[v + "embeddings.position_embedding.weight"])
        for n, key in (("pre", "pre_layrnorm"), ("post", "post_layernorm")):
            w[n + ".g"], w[n + ".b"] = self._f(sd[v + key + ".weight"]), self._f(sd[v + key + ".bias"])
        for i in range(self.layers):
            p, q = f"{v}encoder.layers.{i}.", f"l{i}."
            w[q + "qkv.w"] = self._t(torch.cat([sd[p + f"self_attn.{n}_proj.weight"] for n in ("q", "k", "v")], 0))
            w[q + "qkv.b"] = self._f(torch.cat([sd[p + f"self_attn.{n}_proj.bias"] for n in ("q", "k", "v")], 0))
            w[q + "o.w"], w[q + "o.b"] = self._t(sd[p + "self_attn.out_proj.weight"]), self._f(sd[p + "self_attn.out_proj.bias"])
            w[q + "fc1.w"], w[q + "fc1.b"] = self._t(sd[p + "mlp.fc1.weight"]), self._f(sd[p + "mlp.fc1.bias"])
            w[q + "fc2.w"], w[q + "fc2.b"] = self._t(sd[p + "mlp.fc2.weight"]), self._f(sd[p + "mlp.fc2.bias"])
            for n in ("layer_norm1", "layer_norm2"):
                w[q + n + ".g"], w[q + n + ".b"] = self._f(sd[p + n + ".weight"]), self._f(sd[p + n + ".bias"])
        w["proj.w"] = self._t(sd["visual_projection.weight"])
        self._loaded = True
        return [], [k for k in sd if k not in self.spec]

    def forward(self, pixel_values):
        """pixel_values (n, 3, S, S) (CLIPImageProcessor output) -> .image_embeds (n, projection_dim) fp32."""
        if not self._loaded:
            raise RuntimeError("CLIPVisionModelWithProjection.forward before load_state_dict")
        if not pixel_values.is_cuda:
            raise RuntimeError("mmgt_amd.CLIPVisionModelWithProjection runs on the GPU only (no CPU path exists)")
        n, c, hh, ww = pixel_values.shape
        p, H, heads = self.patch, self.hidden, self.heads
        if c != 3 or hh != self.image_size or ww != self.image_size:
            raise RuntimeError(f"pixel_values must be (n, 3, {self.image_size}, {self.image_size})")
        g = hh // p
        # patch embedding (Conv2d(3, H, p, stride p, bias=False), modeling_clip.py CLIPVisionEmbeddings): the patches are a
        # strided view of the image; unfolded on the host side of the boundary, multiplied on the device
        px = pixel_values.to(self._device, torch.float32).reshape(n, 3, g, p, g, p).permute(0, 2, 4, 1, 3, 5)
        px = px.reshape(n * g * g, 3 * p * p)
        a = torch.zeros((n * g * g, self.w["patch.w"].shape[1]), device=self._device, dtype=self._dtype)
        a[:, :3 * p * p] = px.to(self._dtype)
        emb = hip.gemm(a, self.w["patch.w"]).view(n, g * g, H).float()
        ntok = g * g + 1
        x = torch.cat([self.w["cls"].view(1, 1, H).expand(n, 1, H), emb], 1) + self.w["pos"][None, :ntok]
        x = x.to(self._dtype).reshape(n * ntok, H).contiguous()
        x = hip.layernorm(x, self.w["pre.g"], self.w["pre.b"], self.eps)
        hd = H // heads
        for i in range(self.layers):
            q = f"l{i}."
            h1 = hip.layernorm(x, self.w[q + "layer_norm1.g"], self.w[q + "layer_norm1.b"], self.eps)
            qkv = hip.gemm(h1, self.w[q + "qkv.w"], self.w[q + "qkv.b"])
            o = torch.empty((n * ntok, H), device=self._device, dtype=self._dtype)
            st = (ntok * 3 * H, 0, 3 * H)
            hip.attention(qkv, qkv[:, H:], qkv[:, 2 * H:], o, batch=n, heads=heads, hd=hd, nq=ntok, nk=ntok, scale=hd ** -0.5,
                          q_str=st, k_str=st, v_str=st, o_str=(ntok * H, 0, H))
            x = hip.gemm(o, self.w[q + "o.w"], self.w[q + "o.b"], residual=x)
            h2 = hip.layernorm(x, self.w[q + "layer_norm2.g"], self.w[q + "layer_norm2.b"], self.eps)
            f1 = hip.gemm(h2, self.w[q + "fc1.w"], self.w[q + "fc1.b"], act=hip.ACT_QUICK_GELU)
            x = hip.gemm(f1, self.w[q + "fc2.w"], self.w[q + "fc2.b"], residual=x)
        last = x.view(n, ntok, H)
        pooled = hip.layernorm(last[:, 0].contiguous(), self.w["post.g"], self.w["post.b"], self.eps)
        embeds = hip.gemm(pooled, self.w["proj.w"])
        return CLIPVisionOutput(embeds.float(), last.float())

    __call__ = forward
