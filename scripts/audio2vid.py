#!/usr/bin/env python3
"""audio2vid Stage-2 half on MI355X — counterpart of the reference's scripts/audio2vid.py (:185-498) for the HIP path.

The reference chains Stage-1 SMGA (audio -> pose + motion masks; SURVEY.md section 8f rank 1, not part of this build)
before the Stage-2 sampler.  This driver covers the Stage-2 half with `--synthetic`: synthetic wav2vec-like features go
through `process_audio_emb` (+-2-frame window) and the HIP AudioProjModel (scripts/audio2vid.py:439-441), synthetic
pose / mask videos stand in for SMGA's output, then the same Pose2VideoPipeline call as :484-498.

    python scripts/audio2vid.py --synthetic -W 512 -H 512 -L 24 --steps 25
"""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from scripts.pose2vid import build_synthetic, parse_args  # noqa: E402


def main():
    a = parse_args()
    if not a.synthetic:
        raise SystemExit("only --synthetic is available: SMGA (audio -> pose) and wav2vec feature extraction are outside this build")
    if not torch.cuda.is_available():
        raise SystemExit("audio2vid needs an MI355X (the product has no CPU path)")
    from mmgt_amd.conditioning import full_mask_from_lips, mask_pyramid, process_audio_emb
    from mmgt_amd.side_models import AudioProjModel
    from mmgt_amd.synthetic import hash_uniform, synth_state_dict
    dev = torch.device("cuda:0")
    dtype = torch.bfloat16 if a.dtype == "bf16" else torch.float32
    pipe = build_synthetic(dev, dtype)
    audioproj = AudioProjModel(seq_len=5, blocks=12, channels=768, intermediate_dim=512, output_dim=768,
                               context_tokens=32, device=dev, dtype=dtype)                      # audio2vid.py:264-273
    audioproj.load_state_dict(synth_state_dict(audioproj.spec, prefix="audioproj.", device=dev))
    feats = hash_uniform("a2v.wav2vec", (a.L, 12, 768), 1.0)                                   # (frames, 12 layers, 768)
    audio_tensor = audioproj(process_audio_emb(feats)[None].to(dev))                            # (1, L, 32, 768)
    lat = a.H // 8
    blob = lambda tag: (hash_uniform(tag, (a.L, 64, 64), 0.5) + 0.5) * 255
    face, lips = mask_pyramid(blob("a2v.face"), a.H), mask_pyramid(blob("a2v.lips"), a.H)
    full = full_mask_from_lips(lips)                                                            # :470-476
    pose = hash_uniform("a2v.pose", (1, 3, a.L, a.H, a.W), 0.5) + 0.5
    torch.cuda.synchronize()
    t0 = time.time()
    out = pipe(None, pose, audio_tensor.float(), full, face, lips, a.W, a.H, a.L, a.steps, a.cfg,
               generator=torch.manual_seed(a.seed), motion_scale=[1.0, 1.0, 2.0], context_frames=a.num_c,
               clip_image_embeds=hash_uniform("a2v.clip", (1, 768), 1.0),
               ref_image_latents=hash_uniform("a2v.reflat", (1, 4, lat, a.W // 8), 1.0), decode=not a.no_decode)
    torch.cuda.synchronize()
    v = torch.as_tensor(out.videos)
    print(json.dumps({"video": list(v.shape), "sample_s": round(time.time() - t0, 3), "steps": a.steps,
                      "finite": bool(v.isfinite().all()), "dtype": a.dtype}))


if __name__ == "__main__":
    main()
