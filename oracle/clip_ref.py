"""ORACLE (test infrastructure): CLIP vision tower + projection in plain torch fp32, with the transformers state-dict key
names.  transformers is an un-vendored dependency of the reference (requirements.txt:207 pins 4.30.2; call sites
scripts/pose2vid.py:158-162, src/pipelines/pipeline_pose2vid_long.py:382-387).  Pinned by tests/golden/clip_vision.npz,
generated from the transformers build installed in the build container (tools/refgen/gen_clip_golden.py): the CLIP vision
forward (modeling_clip.py: CLIPVisionEmbeddings, CLIPEncoderLayer, CLIPMLP with quick_gelu, pooled = post_layernorm of the
class token, visual_projection without bias) is unchanged between those versions."""
import torch
import torch.nn.functional as F


def clip_vision_forward(sd, pixel_values, heads, eps=1e-5):
    """pixel_values (n, 3, S, S) -> (image_embeds (n, P), last_hidden_state (n, T, H))."""
    v = "vision_model."
    wp = sd[v + "embeddings.patch_embedding.weight"]
    H, p = wp.shape[0], wp.shape[2]
    n = pixel_values.shape[0]
    emb = F.conv2d(pixel_values, wp, None, stride=p).flatten(2).transpose(1, 2)                 # (n, g*g, H)
    x = torch.cat([sd[v + "embeddings.class_embedding"].view(1, 1, H).expand(n, 1, H), emb], 1)
    x = x + sd[v + "embeddings.position_embedding.weight"][None, :x.shape[1]]
    x = F.layer_norm(x, (H,), sd[v + "pre_layrnorm.weight"], sd[v + "pre_layrnorm.bias"], eps)
    hd = H // heads
    i = 0
    while f"{v}encoder.layers.{i}.layer_norm1.weight" in sd:
        q_ = f"{v}encoder.layers.{i}."
        h = F.layer_norm(x, (H,), sd[q_ + "layer_norm1.weight"], sd[q_ + "layer_norm1.bias"], eps)
        proj = lambda name: F.linear(h, sd[q_ + f"self_attn.{name}_proj.weight"], sd[q_ + f"self_attn.{name}_proj.bias"]) \
            .view(n, -1, heads, hd).transpose(1, 2)
        o = F.scaled_dot_product_attention(proj("q"), proj("k"), proj("v"))                     # scale hd^-0.5
        o = o.transpose(1, 2).reshape(n, -1, H)
        x = x + F.linear(o, sd[q_ + "self_attn.out_proj.weight"], sd[q_ + "self_attn.out_proj.bias"])
        h = F.layer_norm(x, (H,), sd[q_ + "layer_norm2.weight"], sd[q_ + "layer_norm2.bias"], eps)
        h = F.linear(h, sd[q_ + "mlp.fc1.weight"], sd[q_ + "mlp.fc1.bias"])
        h = h * torch.sigmoid(1.702 * h)                                                        # quick_gelu
        x = x + F.linear(h, sd[q_ + "mlp.fc2.weight"], sd[q_ + "mlp.fc2.bias"])
        i += 1
    pooled = F.layer_norm(x[:, 0], (H,), sd[v + "post_layernorm.weight"], sd[v + "post_layernorm.bias"], eps)
    return F.linear(pooled, sd["visual_projection.weight"]), x
