"""ORACLE (test infrastructure): Pose2VideoPipeline.__call__ restated with the oracle pieces, CPU fp32
(src/pipelines/pipeline_pose2vid_long.py:337-660).  The CLIP image encoder and the VAE encoder (once-per-clip prologue,
third-party models) are outside the restatement: their outputs (`clip_image_embeds`, `ref_image_latents`) are inputs."""
import torch

from . import unet3d_ref as R
from .context_ref import uniform
from .ddim_ref import DDIMRef
from .vae_ref import decode_latents


def pose2vid(sd_unet, sd_refnet, sd_pose, sd_vae, *, clip_image_embeds, ref_image_latents, pose_images, audio_tensor,
             full_mask, face_mask, lip_mask, latents, num_inference_steps, guidance_scale, motion_scale,
             context_frames=12, context_stride=1, context_overlap=4, cfg=None, decode=True, trajectory=None,
             unet_dtype=torch.float32):
    """unet_dtype=torch.bfloat16 runs the denoiser (weights and activations) under PyTorch CPU bf16 while the sampler state
    stays fp32, as the HIP product mode does: the measured bf16 noise floor the bf16 gates of tests/ are set against."""
    cfg = cfg or R.UNet3DConfig()
    low = unet_dtype != torch.float32
    cst = (lambda t: t.to(unet_dtype)) if low else (lambda t: t)
    if low:
        sd_unet = {k: cst(v) for k, v in sd_unet.items()}
    sched = DDIMRef()
    sched.set_timesteps(num_inference_steps)
    ehs = clip_image_embeds.reshape(1, 1, -1)
    ehs = torch.cat([torch.zeros_like(ehs), ehs], dim=0)                                   # :388-394
    video_length = latents.shape[2]
    pose_fea = R.pose_guider_forward(sd_pose, pose_images)                                  # :437-448
    dup = lambda ms: [torch.cat([m] * 2) for m in ms]                                       # :451-465
    full_mask, face_mask, lip_mask = dup(full_mask), dup(face_mask), dup(lip_mask)
    audio = torch.cat([torch.zeros_like(audio_tensor), audio_tensor], dim=0)                # :484-486
    banks, _ = R.reference_net_banks(sd_refnet, cfg, ref_image_latents.repeat(2, 1, 1, 1), 0, ehs)   # :510-520
    latents = latents * 1.0                                                                 # init_noise_sigma = 1
    for t in sched.timesteps:
        noise_pred = torch.zeros((2,) + tuple(latents.shape[1:]))
        counter = torch.zeros((1, 1, video_length, 1, 1))
        for c in uniform(0, num_inference_steps, video_length, context_frames, context_stride, context_overlap):
            lat_in = latents[:, :, c].repeat(2, 1, 1, 1, 1)
            pose_in = pose_fea[:, :, c].repeat(2, 1, 1, 1, 1)
            sel = lambda ms: [m.view(2, video_length, -1)[:, c, :].reshape(-1, m.shape[-1]) for m in ms]   # :573-586
            pred = R.unet3d_forward(sd_unet, cfg, cst(lat_in), t, cst(ehs), cst(audio[:, c]), cst(pose_in),
                                    [cst(m) for m in sel(full_mask)], [cst(m) for m in sel(face_mask)],
                                    [cst(m) for m in sel(lip_mask)], motion_scale,
                                    {k: cst(v) for k, v in banks.items()} if low else banks, weighted=True).float()
            noise_pred[:, :, c] = noise_pred[:, :, c] + pred                               # :622-624
            counter[:, :, c] = counter[:, :, c] + 1
        u, ctext = (noise_pred / counter).chunk(2)                                          # :627-631
        latents = sched.step(u + guidance_scale * (ctext - u), t, latents)                  # :633-635
        if trajectory is not None:
            trajectory.append(latents.clone())
    if not decode:
        return latents
    return decode_latents(sd_vae, latents)                                                  # :651,112-125
