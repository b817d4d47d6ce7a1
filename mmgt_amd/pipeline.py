"""Pose2VideoPipeline on MI355X: windowed DDIM sampling around the HIP denoise-step operator.

Host-side mirror of src/pipelines/pipeline_pose2vid_long.py:35-660 (used by scripts/pose2vid.py and scripts/audio2vid.py)
and of the single-window src/pipelines/pipeline_pose2vid.py:284-506 (= context_frames >= video_length).  Same constructor
and __call__ signature; the per-step tensor arithmetic (window accumulate, overlap average, CFG combine, DDIM update) runs
in two HIP kernels on fp32 latents that never leave the GPU.

Extensions through **kwargs (allowed by the reference signature, :365): `latents=` (inject initial noise, parity tests),
`clip_image_embeds=` / `ref_image_latents=` / `reference_banks=` / `pose_features=` (hand over prologue results when the module
is None), `decode=False`, `window_group=` / `cfg_split=` (window-parallel sampling of one long video over several GPUs).
`output_type="uint8"` returns the frames as uint8 (b, f, H, W, 3), converted on the device (what save_videos_grid writes).
"""
import math
from dataclasses import dataclass
from typing import Callable, List, Optional, Union

import numpy as np

import torch

from . import hip, parallel
from .context import get_context_scheduler


@dataclass
class Pose2VideoPipelineOutput:
    videos: Union[torch.Tensor, np.ndarray]


def _pil_to_tensor(img, width, height, normalize):
    """VaeImageProcessor.preprocess for one PIL image: RGB, lanczos resize, /255, CHW, optional 2x-1 (App. B-6)."""
    from PIL import Image
    img = img.convert("RGB").resize((width, height), resample=Image.LANCZOS)
    arr = torch.from_numpy(np.asarray(img).astype(np.float32) / 255.0).permute(2, 0, 1)
    return arr * 2.0 - 1.0 if normalize else arr


class Pose2VideoPipeline:
    def __init__(self, vae, image_encoder, reference_unet, denoising_unet, pose_guider, scheduler,
                 image_proj_model=None, tokenizer=None, text_encoder=None):
        self.vae = vae
        self.image_encoder = image_encoder
        self.reference_unet = reference_unet
        self.denoising_unet = denoising_unet
        self.pose_guider = pose_guider
        self.scheduler = scheduler
        self.image_proj_model = image_proj_model
        self.tokenizer = tokenizer
        self.text_encoder = text_encoder
        self.vae_scale_factor = 8          # 2 ** (len(vae.config.block_out_channels) - 1) for sd-vae-ft-mse

    def to(self, device=None, dtype=None):
        return self

    @property
    def device(self):
        return self.denoising_unet.device

    # --------------------------------------------------------------------------------------------- helpers
    def prepare_latents(self, batch_size, num_channels_latents, width, height, video_length, dtype, device, generator,
                        latents=None):
        """pipeline_pose2vid_long.py:148-182; noise is drawn on the generator's device then moved (App. B-7)."""
        shape = (batch_size, num_channels_latents, video_length, height // self.vae_scale_factor,
                 width // self.vae_scale_factor)
        if isinstance(generator, list) and len(generator) != batch_size:
            raise ValueError(
                f"You have passed a list of generators of length {len(generator)}, but requested an effective batch"
                f" size of {batch_size}. Make sure the batch size matches the length of the generators.")
        if latents is None:
            gdev = generator.device if isinstance(generator, torch.Generator) else torch.device("cpu")
            latents = torch.randn(shape, generator=generator if isinstance(generator, torch.Generator) else None,
                                  device=gdev, dtype=torch.float32)
        latents = latents.to(device=device, dtype=torch.float32)
        return (latents * self.scheduler.init_noise_sigma).contiguous()

    def decode_latents(self, latents, window_group=None, uint8=False):
        """pipeline_pose2vid_long.py:112-125: frame-by-frame VAE decode of z / 0.18215, (x/2+0.5).clamp(0,1), fp32 CPU.
        With a window_group the frames (independent units) are dealt in contiguous runs to the ranks and all-gathered, so
        every rank returns the whole video ("VAE decode sharded by frame")."""
        if self.vae is None:
            raise RuntimeError("decode_latents needs a VAE (pass decode=False to get latents)")
        if window_group is None and uint8:
            return self.vae.decode_video_uint8(latents).cpu().numpy()      # (b, f, H, W, 3) uint8: save_videos_grid's frames
        if window_group is None:
            video = self.vae.decode_video(latents)           # (b, 3, f, H, W) fp32 in [0, 1] on the GPU
            return video.cpu().float().numpy()
        group = None if window_group is True else window_group
        world, rank = parallel.dist.get_world_size(group), parallel.dist.get_rank(group)
        f = latents.shape[2]
        per = (f + world - 1) // world                       # equal runs (the last ranks repeat the final frame as padding)
        idx = torch.arange(rank * per, (rank + 1) * per, device=latents.device).clamp_(max=f - 1)
        if uint8:                                            # (b, per, H, W, 3) uint8 per rank: a quarter of the fp32 bytes on the wire
            part = self.vae.decode_video_uint8(latents[:, :, idx])
            parts = parallel.allgather_window_predictions(part, group)
            return torch.cat(parts, dim=1)[:, :f].cpu().numpy()
        part = self.vae.decode_video(latents[:, :, idx])
        parts = parallel.allgather_window_predictions(part, group)
        return torch.cat(parts, dim=2)[:, :, :f].cpu().float().numpy()

    def interpolate_latents(self, latents, interpolation_factor, device=None):
        """pipeline_pose2vid_long.py:292-335 (no-op below factor 2): all pairs and rates in one batched expression."""
        from .interp import interpolate_frames
        return interpolate_frames(latents, interpolation_factor)

    def _pose_window(self, pose_fea, c):
        """Pose features of one window for both CFG rows (pipeline_pose2vid_long.py:576-580), converted ONCE to the
        operator's channels-last layout and dtype: they are step-invariant, so the per-step forward takes them as is."""
        if pose_fea is None:
            return None
        unet = self.denoising_unet
        p5 = pose_fea[:, :, c].repeat(2, 1, 1, 1, 1).to(torch.float32).contiguous()
        if not p5.is_cuda or not hasattr(unet, "boc"):
            return p5
        return hip.ncfhw_to_nhwc(p5, p5.shape[1], unet.dtype)

    # --------------------------------------------------------------------------------------------- the hot loop
    def denoise(self, latents, timesteps, encoder_hidden_states, pose_fea, audio_tensor_pre, full_masks, face_masks,
                lip_masks, guidance_scale, motion_scale, context_frames, context_stride, context_overlap,
                context_schedule="uniform", num_inference_steps=None, callback=None, callback_steps=1,
                window_group=None, cfg_split="auto"):
        """pipeline_pose2vid_long.py:494-643.  latents (1, C, L, h, w) fp32 on the GPU; returns the final latents.

        window_group: a torch.distributed process group (or True for the default group) turns on window-parallel sampling of
        ONE long video.  The units of a DDIM step -- the windows, or with `cfg_split` the (window, CFG
        row) pairs -- are dealt round-robin to the ranks; each round ends in ONE all-gather of the units' predictions, sliced
        to the C valid channels (fp32 (rows * Fw, h, w, C): 1.57 MB per CFG row at 24 x 64 x 64 x 4), and every rank then
        accumulates ALL units in the reference's window order and applies the identical overlap-average + CFG + DDIM update --
        so the latents stay bit-identical on every rank, with no other exchange.  cfg_split: "auto" splits the CFG rows when
        that shortens the critical path (6 windows on 4 ranks: 3 rounds of half units instead of 2 rounds of whole ones)."""
        video_length = latents.shape[2]
        dev = latents.device
        sched = get_context_scheduler(context_schedule)
        nsteps = num_inference_steps or len(timesteps)
        # the reference passes step = 0 at every iteration, so the windows never move (:534-543)
        windows = list(sched(0, nsteps, video_length, context_frames, context_stride, context_overlap))
        win_idx = [torch.tensor(c, device=dev, dtype=torch.int32) for c in windows]
        win_long = [torch.tensor(c, device=dev, dtype=torch.long) for c in windows]
        # per-window conditioning is step-invariant: gather it once
        cond = []
        for c in win_long:
            cond.append(dict(
                pose=self._pose_window(pose_fea, c),
                audio=audio_tensor_pre[:, c].contiguous(),
                full=[t.view(2, video_length, -1)[:, c].reshape(-1, t.shape[-1]).contiguous() for t in full_masks],
                face=[t.view(2, video_length, -1)[:, c].reshape(-1, t.shape[-1]).contiguous() for t in face_masks],
                lips=[t.view(2, video_length, -1)[:, c].reshape(-1, t.shape[-1]).contiguous() for t in lip_masks]))
        # the reference's unconditional audio row is zeros_like(audio) (:484-485): checked once here (one device sync per clip), and the
        # operator then skips that row's audio cross-attention, whose result is exactly zero
        hip_op = hasattr(self.denoising_unet, "boc")                # the HIP operator (CPU doubles of the tests take no extras)
        uncond_audio_zero = hip_op and hip.tune_get("zero_audio_skip") and audio_tensor_pre.shape[0] == 2 and \
            not bool(audio_tensor_pre[0].ne(0).any())
        keep_window_state = hip_op and bool(hip.tune_get("window_state"))
        C = latents.shape[1]
        group = world = rank = None
        units = [(w, None) for w in range(len(windows))]
        if window_group is not None:
            group = None if window_group is True else window_group
            world, rank = parallel.dist.get_world_size(group), parallel.dist.get_rank(group)
            nw = len(windows)
            if cfg_split == "auto":
                cfg_split = -(-2 * nw // world) < 2 * -(-nw // world)      # rounds of half units vs rounds of whole units
            if cfg_split:
                units = [(w, row) for w in range(nw) for row in (0, 1)]
        half_cond = {}

        def unit_cond(w, row):
            """Conditioning of one CFG row of window w (step-invariant, cut once)."""
            if (w, row) not in half_cond:
                cd, fw = cond[w], len(windows[w])
                cut = lambda t: t.view(2, fw, -1)[row].contiguous()
                pose = cd["pose"]
                if pose is not None:
                    pose = pose.view(2, fw, *pose.shape[1:])[row].contiguous() if pose.dim() == 4 else pose[row:row + 1]
                half_cond[(w, row)] = dict(pose=pose, audio=cd["audio"][row:row + 1].contiguous(),
                                           full=[cut(t) for t in cd["full"]], face=[cut(t) for t in cd["face"]],
                                           lips=[cut(t) for t in cd["lips"]])
            return half_cond[(w, row)]

        for i, t in enumerate(timesteps):
            pred_sum = torch.zeros((2,) + tuple(latents.shape[1:]), device=dev, dtype=torch.float32)
            counter = torch.zeros((video_length,), device=dev, dtype=torch.float32)

            def run_unit(w, row):
                lat_w = self.scheduler.scale_model_input(latents[:, :, win_long[w]], t)
                if row is None:
                    cd, kw = cond[w], {}
                    lat_w = lat_w.repeat(2, 1, 1, 1, 1)
                else:
                    cd, kw = unit_cond(w, row), dict(cfg_row=row)
                if keep_window_state:                            # the HIP operator memoises what it derives from the step-invariant inputs
                    kw["window_state"] = cd.setdefault("state", {})
                if uncond_audio_zero:
                    kw["audio_zero_rows"] = 1 if row in (None, 0) else 0
                if hip_op and row is None:
                    kw["cfg_rows_share_input"] = True            # lat_w.repeat(2 ...) above, pose.repeat(2 ...) in _pose_window
                return self.denoising_unet.denoise_window(
                    lat_w, t, encoder_hidden_states=encoder_hidden_states, audio_embedding=cd["audio"],
                    pose_cond_fea=cd["pose"], full_mask=cd["full"], face_mask=cd["face"], body_mask=cd["lips"],
                    motion_scale=motion_scale, **kw)

            if window_group is None:
                for w in range(len(windows)):
                    hip.accumulate_window(run_unit(w, None), pred_sum, counter, win_idx[w], C)
            else:
                rows = 1 if units[0][1] is not None else 2
                fw = len(windows[0])
                wire_shape = (rows * fw,) + tuple(latents.shape[3:]) + (C,)
                for u0 in range(0, len(units), world):
                    mine = u0 + rank
                    if mine < len(units):
                        # the operator returns ((rows * Fw), h, w, 64) channels-last with C valid channels: only those travel
                        pred = run_unit(*units[mine])[..., :C].float().contiguous()
                        assert pred.shape == wire_shape, (pred.shape, wire_shape)
                    else:                                   # idle rank in the last round: same shape and dtype on the wire
                        pred = torch.zeros(wire_shape, device=dev, dtype=torch.float32)
                    preds = parallel.allgather_window_predictions(pred, group)
                    for r in range(world):
                        if u0 + r < len(units):
                            w, row = units[u0 + r]
                            hip.accumulate_window(preds[r], pred_sum, counter, win_idx[w], C, rows=rows, row0=row or 0,
                                                  bump_counter=row in (None, 0))
            sa_t, sb_t, sa_p, sb_p = self.scheduler.step_coefficients(t)
            latents = hip.cfg_ddim_step(pred_sum, counter, latents, float(guidance_scale), sa_t, sb_t, sa_p, sb_p)
            if callback is not None and i % callback_steps == 0:
                callback(i, t, latents)
        return latents

    # --------------------------------------------------------------------------------------------- __call__
    @torch.no_grad()
    def __call__(self, ref_image, pose_images, audio_tensor, pixel_values_full_mask, pixel_values_face_mask,
                 pixel_values_lip_mask, width, height, video_length, num_inference_steps, guidance_scale,
                 num_images_per_prompt=1, eta: float = 0.0, motion_scale: Optional[List[float]] = None,
                 generator=None, output_type: Optional[str] = "tensor", return_dict: bool = True,
                 callback: Optional[Callable] = None, callback_steps: Optional[int] = 1, context_schedule="uniform",
                 context_frames=12, context_stride=1, context_overlap=4, context_batch_size=1, interpolation_factor=1,
                 **kwargs):
        if eta != 0.0:
            raise NotImplementedError("eta != 0 is not used by the reference scripts")
        if context_batch_size != 1:
            raise NotImplementedError("the reference's window loop is only consistent for context_batch_size=1 (App. C-7)")
        if not guidance_scale > 1.0:
            raise NotImplementedError("classifier-free guidance is mandatory in the reference loop (App. C-7)")
        unet = self.denoising_unet
        dev = unet.device
        self.scheduler.set_timesteps(num_inference_steps, device=None)
        timesteps = self.scheduler.timesteps

        # ---- prologue (once per clip) -------------------------------------------------------------------
        clip_embeds = kwargs.get("clip_image_embeds")
        if clip_embeds is None:
            if self.image_encoder is None:
                raise RuntimeError("no image_encoder: pass clip_image_embeds=(1, 768)")
            from transformers import CLIPImageProcessor
            clip_image = CLIPImageProcessor().preprocess(ref_image.resize((224, 224)), return_tensors="pt").pixel_values
            clip_embeds = self.image_encoder(clip_image.to(dev, dtype=self.image_encoder.dtype)).image_embeds
        ehs = clip_embeds.to(dev).float().reshape(1, 1, -1)
        encoder_hidden_states = torch.cat([torch.zeros_like(ehs), ehs], dim=0)          # :388-394

        banks = kwargs.get("reference_banks")
        if banks is None:
            if self.reference_unet is None:
                raise RuntimeError("no reference_unet: pass reference_banks={prefix: (2, N, C)}")
            ref_latents = kwargs.get("ref_image_latents")
            if ref_latents is None:
                if not hasattr(self.vae, "encode_mean"):
                    raise RuntimeError("the VAE encoder is not part of this build: pass ref_image_latents=(1,4,h,w) "
                                       "(= vae.encode(ref).latent_dist.mean * 0.18215)")
                ref_t = _pil_to_tensor(ref_image, width, height, True)[None].to(dev)
                ref_latents = self.vae.encode_mean(ref_t) * 0.18215                      # :427-434
            banks = self.reference_unet.write_banks(ref_latents.to(dev).float().repeat(2, 1, 1, 1), 0,
                                                    encoder_hidden_states)               # :510-520
            if unet.bank_fp16_roundtrip:                                                  # update(writer, dtype=fp16): :304,340
                banks = {k: v.to(torch.float16) for k, v in banks.items()}
        unet.set_banks(banks)

        latents = self.prepare_latents(num_images_per_prompt, unet.in_channels, width, height, video_length,
                                       torch.float32, dev, generator, latents=kwargs.get("latents"))

        pose_fea = kwargs.get("pose_features")
        if pose_fea is None:
            if torch.is_tensor(pose_images):
                pose_t = pose_images.to(dev).float()
            else:
                pose_t = torch.stack([_pil_to_tensor(p, width, height, False) for p in pose_images], dim=1)[None].to(dev)
            pose_fea = self.pose_guider(pose_t)                                           # :437-448
        pose_fea = pose_fea.to(dev).float()

        dup = lambda ms: [torch.cat([m.to(dev).float()] * 2) for m in ms]                 # :451-482
        full_masks, face_masks, lip_masks = dup(pixel_values_full_mask), dup(pixel_values_face_mask), dup(pixel_values_lip_mask)
        audio = audio_tensor.to(dev).float()
        audio_pre = torch.cat([torch.zeros_like(audio), audio], dim=0)                    # :484-486

        # ---- denoising loop ---------------------------------------------------------------------------------
        latents = self.denoise(latents, timesteps, encoder_hidden_states, pose_fea, audio_pre, full_masks, face_masks,
                               lip_masks, guidance_scale, motion_scale, context_frames, context_stride, context_overlap,
                               context_schedule, num_inference_steps, callback, callback_steps,
                               window_group=kwargs.get("window_group"), cfg_split=kwargs.get("cfg_split", "auto"))
        unet.clear_banks()                                                                # :645-646

        if interpolation_factor > 0:
            latents = self.interpolate_latents(latents, interpolation_factor, dev)
        if not kwargs.get("decode", True):
            return Pose2VideoPipelineOutput(videos=latents) if return_dict else latents
        images = self.decode_latents(latents, kwargs.get("window_group"), uint8=output_type == "uint8")
        if output_type in ("tensor", "uint8"):
            images = torch.from_numpy(images)
        if not return_dict:
            return images
        return Pose2VideoPipelineOutput(videos=images)
