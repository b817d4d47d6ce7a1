#!/usr/bin/env python3
"""Times GroupNorm / LayerNorm at the step's shapes (HIP events, one process): python tools/bench_norm.py
SHAPES=vae: the VAE decoder's GroupNorm shapes (8 frames); GN_ROWS=a,b,..: rows per workgroup to try."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mmgt_amd import hip  # noqa: E402

dev = torch.device("cuda:0")
if os.environ.get("GN_NARROW"):
    hip.tune("gn_narrow", int(os.environ["GN_NARROW"]))
if os.environ.get("GN_LPR0"):
    hip.tune("gn_lpr0", int(os.environ["GN_LPR0"]))


def t_us(fn, reps=20):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


UNET = [(48, 4096, 320), (48, 4096, 640), (48, 1024, 640), (48, 256, 1280), (48, 64, 1280), (48, 4096, 960)]
VAE = [(8, 262144, 128), (8, 262144, 256), (8, 65536, 256), (8, 65536, 512), (8, 16384, 512), (8, 4096, 512)]   # decode of 8 frames
for nb, hw, c in (VAE if os.environ.get("SHAPES") == "vae" else UNET):
    x = torch.randn(nb, hw, c, device=dev).bfloat16()
    g, b = torch.ones(c, device=dev), torch.zeros(c, device=dev)
    o = torch.empty_like(x)
    for il in [int(v) for v in os.environ.get("GN_IL", "-1").split(",")]:
        hip.tune("gn_interleave", il)
        for rows in [int(v) for v in os.environ.get("GN_ROWS", "0").split(",")]:
            hip.tune("gn_rows", rows)
            us = t_us(lambda: hip.groupnorm(x, g, b, 32, 1e-5, silu=True, out=o))
            print(f"groupnorm nb={nb} hw={hw} c={c} interleave={il} rows/wg={rows}: {us:7.1f} us  {3 * x.numel() * 2 / us / 1e6:6.2f} TB/s (3 passes)")
            if c % 8 == 0 and hw >= 1024 and c <= 1280:
                ua = t_us(lambda: hip.groupnorm_affine(x, g, b, 32, 1e-5))
                print(f"   statistics only (groupnorm_affine): {ua:7.1f} us  {x.numel() * 2 / ua / 1e6:6.2f} TB/s")
hip.tune("gn_interleave", -1)
hip.tune("gn_rows", 0)
for rows, c in [(196608, 320), (49152, 640), (12288, 1280)]:
    x = torch.randn(rows, c, device=dev).bfloat16()
    g, b = torch.ones(c, device=dev), torch.zeros(c, device=dev)
    o = torch.empty_like(x)
    us = t_us(lambda: hip.layernorm(x, g, b, out=o))
    print(f"layernorm rows={rows} c={c}: {us:7.1f} us  {2 * x.numel() * 2 / us / 1e6:6.2f} TB/s (2 passes)")
