#!/usr/bin/env python3
"""audio2vid on MI355X — counterpart of the reference's scripts/audio2vid.py (:185-530) for the HIP path: BASELINE config 3,
"SMGA audio2pose + Stage-2 denoise".

    python scripts/audio2vid.py --synthetic -W 512 -H 512 -L 24 --steps 25

The chain is the reference's (line numbers of its scripts/audio2vid.py):
  1. Stage-1 SMGA (:198-200,324-348): one guided 50-step DDIM sample of 80 key-point frames per 3.2-second audio slice, each slice
     conditioned on frame 59 of the previous one (or on the reference image's pose), optional 5-candidate motion selection
     (`find_best_slice`, :79-108)                                  -> mmgt_amd.smga.SMGA.render_sample (HIP)
  2. seam smoothing by cubic splines around every 60th frame (:351-374)   -> host numpy / scipy, as in the reference
  3. key points -> pose / face / lips / hands frames (:386 `pose_vid_generator`: cv2 drawing of src/dwpose into four mp4 files, read
     back at :426-430) -> drawn on the device (mmgt_dwpose_draw, csrc/dwpose.hip: bit-exact on uint8 with oracle/dwpose_ref.py, the restatement of the
     reference drawers and of the OpenCV primitives under them; cv2 itself is not in this image, so that restatement is parity-unpinned), no files
  4. wav2vec2 features of the waveform (:420-426, src/dataset/audio_processor.py:76-131) -> mmgt_amd.wav2vec.Wav2VecModel (HIP);
     mask blur + 4-level pyramid (:453-476), audio window stack + AudioProjModel (:426,439-441) -> device kernels
  5. Pose2VideoPipeline (:484-498), frames converted to uint8 on the device, written as .npy / .gif.

--synthetic: random-init weights of the reference architectures (no checkpoints ship with the reference), a synthetic 16-kHz
waveform for the wav2vec2 leg, and synthetic WavLM + baseline features for SMGA (WavLM / librosa extraction and vocal separation are
host-side and out of scope).  Without --synthetic the script stops with a clear message: it would need those extractors, cv2 and PyAV.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def parse_args():
    p = argparse.ArgumentParser()
    p.add_argument("-c", "--config", default="./configs/prompts/animation.yaml")
    p.add_argument("--image_path", type=str)
    p.add_argument("--audio_path", type=str)
    p.add_argument("--out_dir", type=str, default="scripts/output_videos")
    p.add_argument("--tem_dir", type=str, default="scripts/output_videos/temp")
    p.add_argument("-W", type=int, default=512)
    p.add_argument("-H", type=int, default=512)
    p.add_argument("-L", type=int, default=80)
    p.add_argument("--name", default="baseline_pose")
    p.add_argument("--seed", type=int, default=42)
    p.add_argument("--cfg", type=float, default=3.5)
    p.add_argument("--steps", type=int, default=30)
    p.add_argument("--fps", type=int)
    p.add_argument("--num_c", type=int, default=12, help="context frames per window")
    p.add_argument("--use_motion_selection", default=False, action="store_true")
    p.add_argument("--num_epoch", type=int, default=3400)
    p.add_argument("--feature_type", type=str, default="wavlm")
    p.add_argument("--motion_diffusion_ckpt", type=str, default="./pretrained_weights/MMGT_pretrained/stage_1/audio2pose_best_model.pt")
    p.add_argument("--synthetic", action="store_true")
    p.add_argument("--dtype", default="bf16", choices=["bf16", "fp32"])
    p.add_argument("--format", default="npy", choices=["npy", "gif"])
    return p.parse_args()


def find_best_slice(slice_candidates, last_half):
    """scripts/audio2vid.py:79-108: the candidate whose first poses and mean velocity direction continue `last_half` best."""
    last_pos = last_half[-5:]
    last_v = np.mean((last_half[1:] - last_half[:-1])[-5:], axis=0).reshape(-1, 2)

    def v_angle_score(a, b):
        cos = np.sum(a * b, axis=1) / (np.linalg.norm(a, axis=1) * np.linalg.norm(b, axis=1))
        return np.mean(np.arccos(np.clip(cos, -1.0, 1.0)))
    best, best_score = None, 1e9
    for cand in slice_candidates:
        cand_v = np.mean((cand[1:] - cand[:-1])[-5:], axis=0).reshape(-1, 2)
        score = np.sum(np.abs(cand[:5] - last_pos)) + v_angle_score(cand_v * 1000, last_v * 1000)
        if score < best_score:
            best, best_score = cand, score
    return best


def smooth_seams(tps_origin):
    """scripts/audio2vid.py:361-374: cubic-spline re-interpolation of 14 frames around every 60th frame."""
    from scipy.interpolate import CubicSpline
    out = tps_origin.copy()
    T = tps_origin.shape[0]
    for point in np.arange(60, T, 60):
        s, e = max(0, point - 5), min(T, point + 5)
        x = list(np.arange(s - 3, s)) + list(np.arange(e, e + 3))
        if min(x) < 0 or max(x) >= T:
            continue
        cs = CubicSpline(x, out[x], axis=0)
        out[s - 2:e + 2] = cs(np.arange(s - 2, e + 2))
    return out


def read_wav_16k(path):
    """Mono float waveform of a 16-kHz PCM .wav (stdlib `wave`; the reference resamples with librosa, which this build does not
    include: other rates are refused)."""
    import wave
    with wave.open(path, "rb") as w:
        if w.getframerate() != 16000:
            raise SystemExit(f"audio2vid: {path} is sampled at {w.getframerate()} Hz; resample to 16 kHz first (librosa is not part of this build)")
        if w.getsampwidth() != 2:
            raise SystemExit(f"audio2vid: {path} must be 16-bit PCM")
        pcm = np.frombuffer(w.readframes(w.getnframes()), dtype="<i2").reshape(-1, w.getnchannels())
    return torch.from_numpy(pcm.astype(np.float32).mean(1) / 32768.0)


def main():
    a = parse_args()
    if not a.synthetic:
        raise SystemExit("non-synthetic runs need WavLM / librosa feature extraction, audio decoding, DWPose (onnxruntime, cv2) and PyAV, "
                         "which this build does not include; see INTEGRATION.md for wiring mmgt_amd into the reference's own script")
    if not torch.cuda.is_available():
        raise SystemExit("audio2vid needs an MI355X (the product has no CPU path)")
    from mmgt_amd import conditioning as C
    from mmgt_amd import hip
    from mmgt_amd.side_models import AudioProjModel
    from mmgt_amd.smga import SMGA
    from mmgt_amd.synthetic import build_synthetic_pipeline, hash_uniform, synth_state_dict, synth_tensor
    from mmgt_amd.video_out import save_videos_grid
    dev = torch.device("cuda:0")
    dtype = torch.bfloat16 if a.dtype == "bf16" else torch.float32
    timing = {}
    t0 = time.time()
    from mmgt_amd.smga import smga_spec
    keys = smga_spec(cond_feature_dim=1059 if a.feature_type == "wavlm" else 35)
    smga_sd = {k: (synth_tensor("smga." + k, s) if not k.endswith("rotary.freqs") else torch.zeros(s)) for k, s in keys.items()}
    audio2pose = SMGA(feature_type=a.feature_type, device=dev, dtype=dtype, state_dict=smga_sd)             # :198-200
    pipe = build_synthetic_pipeline(dev, dtype)
    audioproj = AudioProjModel(seq_len=5, blocks=12, channels=768, intermediate_dim=512, output_dim=768, context_tokens=32,
                               device=dev, dtype=dtype)                                                     # :264-273
    audioproj.load_state_dict(synth_state_dict(audioproj.spec, prefix="audioproj.", device=dev))
    timing["build_s"] = round(time.time() - t0, 2)

    # ---- 1. SMGA: audio -> key points, one 80-frame slice per 3.2 s (:300-348)
    n_slices = max(1, -(-a.L // 80))
    cond_list = hash_uniform("a2v.wavlm+baseline", (n_slices, 80, 1059), 1.0)          # WavLM (1024) + baseline (35) features per frame
    init_feature = hash_uniform("a2v.init_pose", (1, 402), 0.8)                         # process_reference_image + mask_leg (:319-321)
    gen = torch.Generator(device=dev).manual_seed(a.seed)
    torch.cuda.synchronize()
    t0 = time.time()
    tps = []
    for i in range(n_slices):
        last_frame = init_feature if i == 0 else torch.from_numpy(tps[-1][59][None])                      # :327,333
        draw = lambda: audio2pose.render_sample(cond_frame=last_frame.float(), cond=cond_list[i], last_half=None, mode="normal",
                                                generator=gen).squeeze(0).cpu().numpy()
        if i > 0 and a.use_motion_selection:
            tps.append(find_best_slice([draw() for _ in range(5)], tps[-1]))                                # :334-342
        else:
            tps.append(draw())
    torch.cuda.synchronize()
    timing["smga_s"] = round(time.time() - t0, 3)
    tps_origin = np.concatenate([init_feature.numpy().astype(np.float32), np.concatenate(tps, 0)[:-1]], 0)  # :351-359
    kps = smooth_seams(tps_origin)[:a.L]                                                                     # (L, 402)

    # ---- 3. key points -> pose / face / lips frames: the reference's DWPose drawing (data/extract_movment_mask_all.py:319-321,
    # src/dwpose) on the device -- no mp4 files written and read back (:386,426-441)
    t0 = time.time()
    pose, face_u8, lips_u8, _hands_u8 = C.pose_frames_device(torch.from_numpy(kps).to(dev), a.H, a.W)
    # ---- 4. conditioning on the device (:426,439-441,453-476)
    face = C.mask_pyramid_device(C.blur_mask_device(face_u8, 31), a.H)
    lips = C.mask_pyramid_device(C.blur_mask_device(lips_u8, 21), a.H)
    full = C.full_mask_from_lips(lips)
    # wav2vec2 features of the (synthetic) 16-kHz waveform: AudioProcessor.preprocess (src/dataset/audio_processor.py:103-126) on the device
    from mmgt_amd.wav2vec import Wav2VecModel, wav2vec_spec
    w2v = Wav2VecModel(device=dev, dtype=dtype)
    w2v.load_state_dict(synth_state_dict(wav2vec_spec(), prefix="w2v.", device=dev))
    if a.audio_path:      # a 16-kHz PCM .wav drives the Stage-2 audio conditioning (audio_processor.py:103-110 loads with librosa at 16 kHz)
        wave = read_wav_16k(a.audio_path)[None, :a.L * 16000 // (a.fps or 25)]
    else:
        wave = hash_uniform("a2v.wave", (1, a.L * 16000 // (a.fps or 25)), 1.0)
    wave = ((wave - wave.mean()) / (wave.var(unbiased=False) + 1e-7).sqrt()).to(dev)                        # Wav2Vec2FeatureExtractor's normalisation
    feats = w2v.audio_emb(wave, a.L)                                                                        # (frames, 12 layers, 768)
    audio_tensor = audioproj(C.process_audio_emb_device(feats)[None])                                       # (1, L, 32, 768)
    torch.cuda.synchronize()
    timing["conditioning_s"] = round(time.time() - t0, 3)

    # ---- 5. Stage 2 (:484-498) + output path
    from PIL import Image
    if a.image_path:
        ref_img = Image.open(a.image_path).convert("RGB")
    else:
        ref_img = Image.fromarray(((hash_uniform("a2v.ref", (a.H, a.W, 3), 0.5) + 0.5) * 255).clamp(0, 255).to(torch.uint8).numpy())
    t0 = time.time()
    out = pipe(ref_img, pose, audio_tensor.float(), full, face, lips, a.W, a.H, a.L, a.steps, a.cfg,
               generator=torch.manual_seed(a.seed), motion_scale=[1.0, 1.0, 2.0], context_frames=a.num_c, output_type="uint8")
    torch.cuda.synchronize()
    timing["stage2_s"] = round(time.time() - t0, 3)
    v = torch.as_tensor(out.videos)                                                                          # (1, L, H, W, 3) uint8
    path = os.path.join(a.out_dir, f"audio2vid_synth_{a.W}x{a.H}x{a.L}.{a.format}")
    save_videos_grid(v, path, n_rows=1, fps=a.fps or 25)
    print(json.dumps({"video": list(v.shape), "video_dtype": str(v.dtype), "saved": path, "slices": n_slices, "steps": a.steps,
                      "dtype": a.dtype, "keypoints_finite": bool(np.isfinite(kps).all()),
                      "mask_levels": [list(m.shape) for m in face], **timing}))


if __name__ == "__main__":
    main()
