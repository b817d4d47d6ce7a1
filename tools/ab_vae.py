#!/usr/bin/env python3
"""Same-process A/B of the VAE decode (8 frames of 512 x 512) over the `gnconv` host switch: 0 = GroupNorm and conv as two launches, 1 = fused
launch behind a statistics pass, 2 = statistics from the producing launch (the default).   python tools/ab_vae.py"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mmgt_amd import hip  # noqa: E402
from mmgt_amd.synthetic import hash_uniform, synth_state_dict  # noqa: E402
from mmgt_amd.vae import AutoencoderKL  # noqa: E402

dev = torch.device("cuda:0")
vae = AutoencoderKL(device=dev, dtype=torch.bfloat16)
vae.load_state_dict(synth_state_dict(vae.spec, prefix="vae.", device=dev))
z = hash_uniform("bench.z", (1, 4, 8, 64, 64), 1.0).to(dev)


def t(fn, reps=3):
    fn()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) * 1e3)
    return best


ref = None
for rnd in range(3):
    for v in (0, 1, 2):
        hip.tune("gnconv", v)
        ms = t(lambda: vae.decode_video(z))
        out = vae.decode_video(z)
        ref = out if ref is None else ref
        print(f"round {rnd}: gnconv={v}: {ms:7.2f} ms per 8 frames = {ms / 8:5.3f} ms per frame   max |d| against gnconv=0: {(out - ref).abs().max().item():.4f}", flush=True)
hip.tune("gnconv", 2)
