#!/usr/bin/env python3
"""The convs behind a nearest 2x upsampling at the step's shapes: the 3 x 3 kernel on the upsampled view (upsample=True) against four 2 x 2 convs on the
stored image (upsample=2, packing.pack_conv3x3_up2: the same function with 16 / 36 of the multiply-adds).   python tools/bench_up2.py"""
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mmgt_amd import hip  # noqa: E402
from mmgt_amd.packing import pack_conv3x3, pack_conv3x3_up2  # noqa: E402

dev = torch.device("cuda:0")


def t_us(fn, reps=10):
    for _ in range(3):
        fn()
    best = 1e9
    for _ in range(4):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / reps * 1e3)
    return best


SHAPES = [(48, 8, 1280), (48, 16, 1280), (48, 32, 640), (24, 16, 1280), (24, 32, 640), (8, 64, 512), (8, 128, 512), (8, 256, 256)]
print(f"{'nb x h^2 x c':20s} {'3x3 on 2h':>10s} {'4 x 2x2':>10s}   algorithmic TF/s (36 MACs per stored pixel)   max|d|")
for nb, h, c in SHAPES:
    x = (torch.rand((nb, h, h, c), device=dev) * 2 - 1).bfloat16()
    w = ((torch.rand((c, c, 3, 3), device=dev) * 2 - 1) / math.sqrt(9 * c)).bfloat16()
    b = torch.rand(c, device=dev)
    w9, w4 = pack_conv3x3(w), pack_conv3x3_up2(w)
    o = torch.empty((nb, 2 * h, 2 * h, c), device=dev, dtype=torch.bfloat16)
    t9 = t_us(lambda: hip.conv3x3(x, w9, b, upsample=True, out=o))
    o9 = o.clone()
    t4 = t_us(lambda: hip.conv3x3(x, w4, b, upsample=2, out=o))
    fl = 2.0 * nb * 4 * h * h * c * 9 * c
    print(f"{nb} x {h}^2 x {c}".ljust(20) + f" {t9:10.1f} {t4:10.1f}   {fl / t9 / 1e6:6.0f} -> {fl / t4 / 1e6:6.0f}   {(o.float() - o9.float()).abs().max().item():.2e}", flush=True)
