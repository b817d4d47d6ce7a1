#!/usr/bin/env python3
"""pose2vid on MI355X — counterpart of the reference's scripts/pose2vid.py (same flags, :306-321) for the HIP path.

    python scripts/pose2vid.py --synthetic -W 512 -H 512 -L 24 --steps 25         # BASELINE config 2 geometry
    python scripts/pose2vid.py --synthetic -W 64 -H 64 -L 8 --steps 4             # config 1 geometry

    python scripts/pose2vid.py --image_path ref.png --pose_path pose_frames/ --face_mask_path face.npy --lips_mask_path lips.npy \
        [--hands_mask_path hands/] -c configs/prompts/animation.yaml          # the reference's single-sample mode (:196-300)

--synthetic: random-init weights of the reference architecture (no checkpoints ship with the reference) and synthetic
pose / mask / audio inputs, everything a pure function of names (mmgt_amd/synthetic.py); prints the timing breakdown.
Without --synthetic the inputs are FILES, as in the reference (scripts/pose2vid.py:196-271): the reference image through PIL, the
pose / mask clips as directories of images, .npy stacks or animated images (mmgt_amd/inputs.py: video containers need PyAV / cv2,
which this build does not include, and say so), the masks blurred / resampled on the device (mmgt_amd/conditioning.py).  Weights
come from the checkpoints the config yaml names (:137-190) -- or, with --random-weights, from the hash-seeded initialisation (no
checkpoint exists in this image).  The clip is written as .gif / .npy (mp4 muxing is out of scope: mmgt_amd/video_out.py).
"""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def parse_args():
    p = argparse.ArgumentParser()
    p.add_argument("-c", "--config", type=str, default="./configs/prompts/animation.yaml")
    p.add_argument("--image_path", type=str)
    p.add_argument("--pose_path", type=str)
    p.add_argument("--face_mask_path", type=str)
    p.add_argument("--lips_mask_path", type=str)
    p.add_argument("--hands_mask_path", type=str,
                   help="(L, h, w) uint8 .npy of hands masks, or 'synthetic': full = clamp(1 - face + lips + hands, 0, 1) per level "
                        "(reference :239-271, SURVEY App. C-8); without it full = 1 + lips (the audio2vid convention)")
    p.add_argument("--out_dir", type=str, default="./output")
    p.add_argument("-W", type=int, default=512)
    p.add_argument("-H", type=int, default=512)
    p.add_argument("-L", type=int, default=80)
    p.add_argument("--seed", type=int, default=42)
    p.add_argument("--fps", type=int, default=25)
    p.add_argument("--num_c", type=int, default=12, help="context frames per window")
    p.add_argument("--steps", type=int, default=30)          # animation.yaml:28
    p.add_argument("--cfg", type=float, default=3.5)         # animation.yaml:29
    p.add_argument("--synthetic", action="store_true")
    p.add_argument("--random-weights", action="store_true",
                   help="file inputs with the hash-seeded random initialisation instead of the checkpoints of --config")
    p.add_argument("--format", default="gif", choices=["gif", "npy"], help="output container of the file-input mode")
    p.add_argument("--dtype", default="bf16", choices=["bf16", "fp32"])
    p.add_argument("--no-decode", action="store_true")
    p.add_argument("--clip-parallel", action="store_true",
                   help="BASELINE config 4: one clip per rank of a torch.distributed.run launch.  Rank 0 builds the weights and "
                        "broadcasts them (RCCL, a few large buckets), rank r samples clip r (seed + r), the uint8 frames are "
                        "gathered on rank 0; no collective inside the sampling loop (SURVEY 8e)")
    p.add_argument("--window-parallel", action="store_true",
                   help="one long video over all ranks of a torch.distributed.run launch: the windows of every DDIM step are "
                        "dealt to the ranks, one RCCL all-gather per round (SURVEY 8e, config 5)")
    return p.parse_args()


def build_synthetic(dev, dtype):
    from mmgt_amd.synthetic import build_synthetic_pipeline
    return build_synthetic_pipeline(dev, dtype)


def build_synthetic_broadcast(dev, dtype, rank):
    """--clip-parallel: rank 0 holds the checkpoint (here: the hash-seeded synthetic one), every other rank receives it through
    mmgt_amd.parallel.broadcast_state_dict -- the north-star's "broadcast weights" -- and packs it locally."""
    from mmgt_amd import parallel
    from mmgt_amd.clip_vision import CLIPVisionModelWithProjection, clip_vision_spec
    from mmgt_amd.pipeline import Pose2VideoPipeline
    from mmgt_amd.reference_unet import UNet2DConditionModel
    from mmgt_amd.scheduler import DDIMScheduler
    from mmgt_amd.side_models import PoseGuider
    from mmgt_amd.synthetic import synth_state_dict
    from mmgt_amd.unet3d import UNet3DConditionModel
    from mmgt_amd.unet3d_spec import unet2d_reference_spec, unet3d_spec
    from mmgt_amd.vae import AutoencoderKL, vae_decoder_spec, vae_encoder_spec
    pg = PoseGuider(320, block_out_channels=(16, 32, 96, 256), device=dev, dtype=dtype)
    vspec = vae_decoder_spec()
    vspec.update(vae_encoder_spec())
    parts = [("unet", unet3d_spec(), "", UNet3DConditionModel(device=dev, dtype=dtype)),
             ("refnet", unet2d_reference_spec(), "refnet.", UNet2DConditionModel(device=dev, dtype=dtype)),
             ("pose", dict(pg.spec), "pose_guider.", pg),
             ("vae", vspec, "vae.", AutoencoderKL(device=dev, dtype=dtype)),
             ("clip", clip_vision_spec(), "clip.", CLIPVisionModelWithProjection(device=dev, dtype=dtype))]
    mods = {}
    for name, spec, prefix, mod in parts:
        sd = synth_state_dict(spec, prefix=prefix, device=dev) if rank == 0 else None
        sd = parallel.broadcast_state_dict(sd, spec, src=0, device=dev, dtype=torch.float32)
        mod.load_state_dict(sd)
        mods[name] = mod
        del sd
    mods["unet"].enable_gradient_checkpointing()
    return Pose2VideoPipeline(vae=mods["vae"], image_encoder=mods["clip"], reference_unet=mods["refnet"], denoising_unet=mods["unet"],
                              pose_guider=mods["pose"], scheduler=DDIMScheduler())


def build_from_checkpoints(cfg_path, num_c, dev, dtype):
    """scripts/pose2vid.py:137-197 of the reference: the modules from the checkpoints named by the config yaml."""
    import yaml
    from mmgt_amd.clip_vision import CLIPVisionModelWithProjection
    from mmgt_amd.inputs import load_checkpoint, split_net_checkpoint
    from mmgt_amd.pipeline import Pose2VideoPipeline
    from mmgt_amd.reference_unet import UNet2DConditionModel
    from mmgt_amd.scheduler import DDIMScheduler
    from mmgt_amd.side_models import PoseGuider
    from mmgt_amd.unet3d import UNet3DConditionModel
    from mmgt_amd.vae import AutoencoderKL
    if not os.path.isfile(cfg_path):
        raise SystemExit(f"pose2vid: config {cfg_path} not found (it names the checkpoints: pretrained_vae_path, pretrained_base_model_path, "
                         "image_encoder_path, audio_ckpt_dir, inference_config); pass --random-weights to run without checkpoints")
    cfg = yaml.safe_load(open(cfg_path))
    infer = yaml.safe_load(open(cfg["inference_config"]))
    vae = AutoencoderKL(device=dev, dtype=dtype)
    vae.load_state_dict(load_checkpoint(cfg["pretrained_vae_path"]))
    refnet = UNet2DConditionModel(device=dev, dtype=dtype)
    unet = UNet3DConditionModel.from_pretrained_2d(cfg["pretrained_base_model_path"], os.path.join(cfg["audio_ckpt_dir"], f"net-{num_c}.pth"),
                                                   subfolder="unet", unet_additional_kwargs=infer["unet_additional_kwargs"], device=dev, dtype=dtype)
    pg = PoseGuider(320, block_out_channels=(16, 32, 96, 256), device=dev, dtype=dtype)
    clip = CLIPVisionModelWithProjection(device=dev, dtype=dtype)
    clip.load_state_dict(load_checkpoint(cfg["image_encoder_path"]))
    net = split_net_checkpoint(load_checkpoint(os.path.join(cfg["audio_ckpt_dir"], "modules", f"net-{num_c}.pth")))   # (:186-190)
    refnet.load_state_dict(net["reference_unet"])
    unet.load_state_dict(net["denoising_unet"], strict=False)
    pg.load_state_dict(net["pose_guider"])
    unet.enable_gradient_checkpointing()
    return Pose2VideoPipeline(vae=vae, image_encoder=clip, reference_unet=refnet, denoising_unet=unet, pose_guider=pg,
                              scheduler=DDIMScheduler(**infer.get("noise_scheduler_kwargs", {})))


def run_files(a, dev, dtype):
    """The reference's single-sample mode on files (:196-300)."""
    from PIL import Image
    from mmgt_amd import inputs
    from mmgt_amd.video_out import save_videos_grid
    for name in ("image_path", "pose_path", "face_mask_path", "lips_mask_path"):
        if not getattr(a, name):
            raise SystemExit(f"pose2vid: --{name} is required without --synthetic")
    t0 = time.time()
    pipe = build_synthetic(dev, dtype) if a.random_weights else build_from_checkpoints(a.config, a.num_c, dev, dtype)
    t_build = time.time() - t0
    ref_img = Image.open(a.image_path).convert("RGB")
    pose_frames = inputs.read_frames(a.pose_path, a.L)
    face_frames = inputs.read_frames(a.face_mask_path, a.L)
    lips_frames = inputs.read_frames(a.lips_mask_path, a.L)
    hands_frames = inputs.read_frames(a.hands_mask_path, a.L) if a.hands_mask_path and os.path.exists(a.hands_mask_path) else None
    L = min(len(pose_frames), len(face_frames), len(lips_frames), len(hands_frames) if hands_frames else 10 ** 9, a.L)
    if L < 1:
        raise SystemExit("pose2vid: no usable frames (check the pose / mask inputs)")
    if L < a.L:
        print(f"note: {L} usable frames < L = {a.L}: sampling {L} frames")
    pose = inputs.pose_tensor(pose_frames[:L], a.W, a.H)
    full, face, lips = inputs.motion_masks(face_frames, lips_frames, hands_frames, L, dev, a.H)
    audio = torch.zeros(1, L, 32, 768)                        # pose2vid runs with null audio (:279)
    gen = torch.manual_seed(a.seed)
    torch.cuda.synchronize()
    t0 = time.time()
    out = pipe(ref_img, pose, audio, full, face, lips, a.W, a.H, L, a.steps, a.cfg, generator=gen, motion_scale=[1.0, 1.0, 2.0],
               context_frames=a.num_c, output_type="uint8")
    torch.cuda.synchronize()
    dt = time.time() - t0
    save_dir = os.path.join(a.out_dir, f"multi_person_{a.num_c}")
    path = os.path.join(save_dir, f"{os.path.splitext(os.path.basename(a.image_path))[0]}.{a.format}")
    save_videos_grid(out.videos, path, n_rows=1, fps=a.fps)
    v = torch.as_tensor(out.videos)
    print(json.dumps({"video": list(v.shape), "saved": path, "frames": L, "build_s": round(t_build, 2), "sample_s": round(dt, 3),
                      "steps": a.steps, "weights": "random" if a.random_weights else a.config, "dtype": a.dtype}))


def main():
    a = parse_args()
    if not torch.cuda.is_available():
        raise SystemExit("pose2vid needs an MI355X (the product has no CPU path)")
    local = int(os.environ.get("LOCAL_RANK", "0"))
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)
    rank = 0
    if a.window_parallel or a.clip_parallel:
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=dev)        # RANK / WORLD_SIZE / MASTER_* from torch.distributed.run
        rank = dist.get_rank()
    dtype = torch.bfloat16 if a.dtype == "bf16" else torch.float32
    if not a.synthetic:
        if a.clip_parallel or a.window_parallel:
            raise SystemExit("pose2vid: --clip-parallel / --window-parallel run with --synthetic inputs")
        return run_files(a, dev, dtype)
    from mmgt_amd.synthetic import hash_uniform, synth_masks
    t0 = time.time()
    # window-parallel: the same hash-seeded weights are built on every rank; clip-parallel: rank 0 broadcasts them
    pipe = build_synthetic_broadcast(dev, dtype, rank) if a.clip_parallel else build_synthetic(dev, dtype)
    t_build = time.time() - t0
    lat = a.H // 8
    clip_id = rank if a.clip_parallel else 0                  # clip-parallel: rank r samples its own clip
    gen = torch.manual_seed(a.seed + clip_id)                 # :171
    lips, face = synth_masks("p2v.lips", a.L, lat), synth_masks("p2v.face", a.L, lat)
    if a.hands_mask_path:                                     # blur (21, 21) -> 4-level pyramid, like the lips masks (:249-263)
        from mmgt_amd import conditioning as C
        import numpy as np
        if a.hands_mask_path == "synthetic":
            hands_u8 = (synth_masks("p2v.hands", a.L, 64)[0].view(a.L, 64, 64) * 255).to(torch.uint8)
        else:
            hands_u8 = torch.from_numpy(np.load(a.hands_mask_path))[:a.L].to(torch.uint8)
        hands = [h.cpu() for h in C.mask_pyramid_device(C.blur_mask_device(hands_u8.to(dev).contiguous(), 21), a.H)]
        full = C.full_mask_with_hands(face, lips, hands)
    else:
        full = [1 + l for l in lips]                          # audio2vid convention (scripts/audio2vid.py:470-476)
    pose = hash_uniform(f"p2v.pose{clip_id or ''}", (1, 3, a.L, a.H, a.W), 0.5) + 0.5
    audio = torch.zeros(1, a.L, 32, 768)                      # pose2vid runs with null audio (:279)
    from PIL import Image
    ref_img = Image.fromarray(((hash_uniform("p2v.ref", (a.H, a.W, 3), 0.5) + 0.5) * 255).clamp(0, 255).to(torch.uint8).numpy())
    torch.cuda.synchronize()
    t0 = time.time()
    out = pipe(ref_img, pose, audio, full, face, lips, a.W, a.H, a.L, a.steps, a.cfg, generator=gen,
               motion_scale=[1.0, 1.0, 2.0], context_frames=a.num_c, output_type="uint8" if a.clip_parallel else "tensor",
               decode=not a.no_decode,                         # CLIP embedding and ref_image_latents: HIP encoders, from ref_img
               window_group=True if a.window_parallel else None)
    torch.cuda.synchronize()
    dt = time.time() - t0
    if a.clip_parallel:
        from mmgt_amd import parallel
        mine = torch.as_tensor(out.videos).to(dev)            # gather the decoded clips on rank 0 ("gather frames")
        clips = parallel.gather_frames(mine, dst=0)
        world = dist.get_world_size()
        dist.destroy_process_group()
        if rank != 0:
            return
        out.videos = torch.cat([c.cpu() for c in clips], dim=0)
        print(json.dumps({"clip_parallel_ranks": world, "clips": int(out.videos.shape[0])}))
    if a.window_parallel:
        dist.destroy_process_group()
        if rank != 0:                                         # every rank holds the same video; rank 0 writes it
            return
    v = out.videos
    os.makedirs(a.out_dir, exist_ok=True)
    path = os.path.join(a.out_dir, f"pose2vid_synth_{a.W}x{a.H}x{a.L}.pt")
    torch.save(v if torch.is_tensor(v) else torch.from_numpy(v), path)
    print(json.dumps({"video": list(v.shape), "saved": path, "build_s": round(t_build, 2), "sample_s": round(dt, 3),
                      "steps": a.steps, "windows_per_step": len(list(__import__("mmgt_amd.context", fromlist=["uniform"]).uniform(
                          0, a.steps, a.L, a.num_c, 1, 4))), "dtype": a.dtype, "finite": bool(torch.as_tensor(v).isfinite().all())}))


if __name__ == "__main__":
    main()
