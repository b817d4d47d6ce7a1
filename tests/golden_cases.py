"""Case definitions shared by tools/refgen/gen_golden.py (which runs the reference) and the tests (which run the
oracle and the HIP path): every input below is a pure function of its name via mmgt_amd/synthetic.py."""
import torch

from mmgt_amd.synthetic import hash_uniform, synth_masks

CONTEXT_CASES = [(8, 12, 4), (24, 12, 4), (24, 24, 4), (80, 12, 4), (96, 24, 8), (30, 12, 4), (13, 12, 4)]

UNET_CASES = {
    # G2: tiny width, pins all wiring incl. the odd audio geometry
    "tiny": dict(block_out_channels=(32, 64, 128, 128), cross_attention_dim=64, audio_attention_dim=48, frames=4,
                 latent=8, timestep=959),
    # BASELINE config 1 geometry at full width: 64x64 px -> 8x8 latent, 8 frames
    "full_cfg1": dict(block_out_channels=(320, 640, 1280, 1280), cross_attention_dim=768, audio_attention_dim=768,
                      frames=8, latent=8, timestep=499),
    # G4: BASELINE config 2 = the benchmarked shape, 512x512 px -> 64x64 latent, 24 frames.  The fixture holds a
    # strided sub-sample + moments of the reference's (2,4,24,64,64) output (tools/refgen/gen_golden.py --only full_cfg2).
    "full_cfg2": dict(block_out_channels=(320, 640, 1280, 1280), cross_attention_dim=768, audio_attention_dim=768,
                      frames=24, latent=64, timestep=499),
}

G4_STRIDE = 97          # flat-index stride of the committed sub-sample (prime: walks every (b, c, f, y, x) residue)


def g4_summary(out):
    """The committed summary of a (2,4,F,H,W) prediction: strided sub-sample + per-(b,c) and per-(b,frame) moments."""
    flat = out.reshape(-1)
    return dict(sub=flat[::G4_STRIDE].clone(), mean_bc=out.mean(dim=(2, 3, 4)), std_bc=out.std(dim=(2, 3, 4)),
                absmax_bc=out.abs().amax(dim=(2, 3, 4)), mean_bf=out.mean(dim=(1, 3, 4)), std_bf=out.std(dim=(1, 3, 4)),
                mean_abs=out.abs().mean())


def bank_spatial(case):
    """{reader prefix: (N, C)} in the reference's module order down -> up -> mid."""
    from mmgt_amd.synthetic import bank_spatial as _bs
    return _bs(case["block_out_channels"], case["latent"])


def unet_inputs(case, tag="u"):
    boc, f, h = case["block_out_channels"], case["frames"], case["latent"]
    cad, aad = case["cross_attention_dim"], case["audio_attention_dim"]
    sample = hash_uniform(tag + ".sample", (1, 4, f, h, h), 1.7).repeat(2, 1, 1, 1, 1)
    ehs = torch.cat([torch.zeros(1, 1, cad), hash_uniform(tag + ".ehs", (1, 1, cad), 1.0)])
    audio = torch.cat([torch.zeros(1, f, 32, aad), hash_uniform(tag + ".audio", (1, f, 32, aad), 1.7)])
    pose = hash_uniform(tag + ".pose", (1, boc[0], f, h, h), 0.5).repeat(2, 1, 1, 1, 1)
    lips = synth_masks(tag + ".lips", f, h)
    face = synth_masks(tag + ".face", f, h)
    full = [1 + l for l in lips]                                   # scripts/audio2vid.py:470-476
    cat2 = lambda L: [torch.cat([x] * 2) for x in L]               # pipeline_pose2vid_long.py:451-465
    banks = {k: hash_uniform(tag + ".bank." + k, (2, n, c), 1.0) for k, (n, c) in bank_spatial(case).items()}
    return dict(sample=sample, timestep=torch.tensor(case["timestep"]), ehs=ehs, audio=audio, pose=pose,
                full=cat2(full), face=cat2(face), lips=cat2(lips), motion_scale=[1.0, 1.0, 2.0], banks=banks)


# G3: full-width single modules at reduced H, W, F.  b = 2 (CFG), f frames, hw x hw latent.
BLOCK_CASES = {
    "resnet_320_320": dict(kind="resnet", cin=320, cout=320, f=3, hw=8),
    "resnet_960_320": dict(kind="resnet", cin=960, cout=320, f=2, hw=8),
    "resnet_320_640": dict(kind="resnet", cin=320, cout=640, f=2, hw=6),
    "resnet_2560_1280": dict(kind="resnet", cin=2560, cout=1280, f=2, hw=4),
    "down_320": dict(kind="down", c=320, f=2, hw=8),
    "up_640": dict(kind="up", c=640, f=2, hw=4),
    "spatial_320": dict(kind="spatial", c=320, f=3, hw=8),
    "spatial_640": dict(kind="spatial", c=640, f=2, hw=6),
    "spatial_1280": dict(kind="spatial", c=1280, f=2, hw=4),
    "audio_320_d0": dict(kind="audio", c=320, cin=320, depth=0, f=3, hw=8),
    "audio_640_in320_d1": dict(kind="audio", c=640, cin=320, depth=1, f=2, hw=4),     # odd geometry, App. C-3
    "audio_1280_d2": dict(kind="audio", c=1280, cin=1280, depth=2, f=2, hw=2),
    "motion_320": dict(kind="motion", c=320, f=8, hw=4),
    "motion_1280": dict(kind="motion", c=1280, f=24, hw=2),
}


def block_inputs(name):
    c = BLOCK_CASES[name]
    f, hw, kind = c["f"], c["hw"], c["kind"]
    cin = c.get("cin", c.get("c")) if kind == "resnet" else c["c"]
    x = hash_uniform(name + ".x", (2, cin, f, hw, hw), 1.7)
    out = dict(x=x)
    if kind == "resnet":
        out["temb"] = hash_uniform(name + ".temb", (2, 1280), 1.0)
    if kind == "spatial":
        out["ehs"] = torch.cat([torch.zeros(1, 1, 768), hash_uniform(name + ".ehs", (1, 1, 768), 1.0)])
        out["bank"] = hash_uniform(name + ".bank", (2, hw * hw, c["c"]), 1.0)
    if kind == "audio":
        out["audio"] = torch.cat([torch.zeros(1, f, 32, 768), hash_uniform(name + ".audio", (1, f, 32, 768), 1.7)])
        # pyramid level `depth` must match this block's N = hw*hw; build the pyramid so that level depth has hw
        base = hw << c["depth"]
        lips = synth_masks(name + ".lips", f, base)
        face = synth_masks(name + ".face", f, base)
        full = [1 + l for l in lips]
        cat2 = lambda L: [torch.cat([m] * 2) for m in L]
        out.update(full=cat2(full), face=cat2(face), lips=cat2(lips), motion_scale=[1.0, 1.0, 2.0])
    return out


def side_inputs():
    return dict(pose_rgb=hash_uniform("side.pose_rgb", (1, 3, 2, 64, 64), 0.5) + 0.5,
                audio_feats=hash_uniform("side.audio_feats", (1, 3, 5, 12, 768), 1.0))


def interp_inputs():
    return dict(v0=hash_uniform("interp.v0", (1, 4, 8, 8), 1.7), v1=hash_uniform("interp.v1", (1, 4, 8, 8), 1.7))


REFNET_CASES = {
    "tiny": dict(block_out_channels=(32, 64, 128, 128), cross_attention_dim=64, latent=8),
    "full": dict(block_out_channels=(320, 640, 1280, 1280), cross_attention_dim=768, latent=8),
}


def refnet_inputs(case, tag="rn"):
    cad, h = case["cross_attention_dim"], case["latent"]
    lat = hash_uniform(tag + ".latents", (1, 4, h, h), 1.0).repeat(2, 1, 1, 1)      # same latents in both CFG rows
    ehs = torch.cat([torch.zeros(1, 1, cad), hash_uniform(tag + ".ehs", (1, 1, cad), 1.0)])
    return dict(latents=lat, ehs=ehs, timestep=torch.tensor(0))


# ---- CLIP vision tower (the reference's image_encoder; transformers.CLIPVisionModelWithProjection) ----
CLIP_CASES = {
    # tiny: two layers, head_dim 64, 17 tokens (ragged against every tile size)
    "tiny": dict(hidden_size=128, intermediate_size=256, num_hidden_layers=2, num_attention_heads=2, image_size=56,
                 patch_size=14, projection_dim=64, batch=2),
    # ViT-L/14 width and token count (sd-image-variations image_encoder), two of its 24 identical layers
    "vitl_2layers": dict(hidden_size=1024, intermediate_size=4096, num_hidden_layers=2, num_attention_heads=16,
                         image_size=224, patch_size=14, projection_dim=768, batch=1),
}


def clip_state_dict(case, device="cpu"):
    """Hash-seeded weights for a CLIP vision case: generic rule of mmgt_amd.synthetic, LayerNorm gains around 1."""
    import torch
    from mmgt_amd.clip_vision import clip_vision_spec
    from mmgt_amd.synthetic import hash_uniform, synth_state_dict
    spec = clip_vision_spec(case["hidden_size"], case["intermediate_size"], case["num_hidden_layers"], case["image_size"],
                            case["patch_size"], case["projection_dim"])
    sd = synth_state_dict(spec, prefix="clip.", device=device)
    for k in spec:
        if "norm" in k and k.endswith(".weight"):
            sd[k] = (1.0 + hash_uniform("clip." + k, spec[k], 0.1, device)).to(torch.float32)
        if k.endswith("class_embedding") or k.endswith("position_embedding.weight"):
            sd[k] = hash_uniform("clip." + k, spec[k], 0.5, device)
    return sd


def clip_pixels(case, device="cpu"):
    from mmgt_amd.synthetic import hash_uniform
    s = case["image_size"]
    return hash_uniform("clip.pixels", (case["batch"], 3, s, s), 1.5, device)
