#!/usr/bin/env python3
"""Per call-site profile of the VAE decode of one 512x512 frame batch (HIP events around the mmgt_amd.hip entry points), as
tools/profile_step.py does for the denoise step.   python tools/profile_vae.py [frames]"""
import collections
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mmgt_amd import hip  # noqa: E402
from tools import profile_step as P  # noqa: E402


def main():
    frames = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    from mmgt_amd.synthetic import hash_uniform, synth_state_dict
    from mmgt_amd.vae import AutoencoderKL
    dev = torch.device("cuda:0")
    vae = AutoencoderKL(device=dev, dtype=torch.bfloat16)
    vae.load_state_dict(synth_state_dict(vae.spec, prefix="vae.", device=dev))
    lat = hash_uniform("pv.lat", (1, 4, frames, 64, 64), 1.0).to(dev)
    vae.decode_video(lat, frames_per_batch=frames)                      # warm-up
    for n in ["gemm", "gemm_batched", "gemm_batched_wx", "conv3x3", "groupnorm", "attention", "softmax_rows", "nhwc_to_ncfhw", "ncfhw_to_nhwc",
              "gemm_bf16_f32", "qk_split3", "softmax_rows_f32_bf16", "gn_silu_conv3x3_tables", "groupnorm_affine"]:
        if hasattr(hip, n):
            P.wrap(n)
    torch.cuda.synchronize()
    vae.decode_video(lat, frames_per_batch=frames)
    torch.cuda.synchronize()
    agg = collections.OrderedDict()
    for (k, fl), e0, e1 in P.records:
        a = agg.setdefault(k, [0, 0.0, 0])
        a[0] += 1
        a[1] += e0.elapsed_time(e1) * 1e3
        a[2] += fl
    tot = sum(a[1] for a in agg.values())
    print(f"VAE decode of {frames} frames: {tot / 1e3:.2f} ms inside calls = {tot / frames / 1e3:.2f} ms per frame")
    for k, (n, t, fl) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:30]:
        tf = f"{fl / t / 1e6:7.0f} TF/s" if fl else " " * 12
        print(f"{t / 1e3:8.3f} ms {100 * t / tot:5.1f}%  x{n:3d}  {t / n:8.1f} us  {tf}  {k}")


if __name__ == "__main__":
    main()
