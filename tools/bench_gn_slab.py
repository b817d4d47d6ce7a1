#!/usr/bin/env python3
"""GroupNorm below the 64 x 64 level: the register-resident single-read kernel (csrc/norm.hip gn_slab_kernel, tune key gn_slab) against the forms it
replaces (gn_small_kernel's three passes over L2 for HW <= 256, the two-kernel form above), full op and statistics only.   python tools/bench_gn_slab.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mmgt_amd import hip  # noqa: E402

dev = torch.device("cuda:0")


def t_us(fn, reps=20):
    for _ in range(3):
        fn()
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / reps * 1e3)
    return best


SHAPES = [(48, 1024, 1280, 640), (48, 1024, 640, 320), (48, 1024, 640, 0), (48, 1024, 320, 0), (48, 1024, 1280, 0), (48, 1024, 640, 640), (48, 256, 1280, 0), (48, 256, 640, 0), (48, 256, 1280, 1280),
          (48, 64, 1280, 0), (48, 64, 1280, 1280), (24, 1024, 640, 0), (24, 256, 1280, 0)]
print(f"{'nb x hw x (c0 + c1)':28s} {'passes':>8s} {'slab':>8s}   {'stats: passes':>14s} {'slab':>8s}   TB/s of the slab form (read + write)   max|d| between the forms")
for nb, hw, c0, c1 in SHAPES:
    c = c0 + c1
    x0 = (torch.randn(nb, hw, c0, device=dev) * 1.5 + 0.7).bfloat16()
    x1 = torch.randn(nb, hw, c1, device=dev).bfloat16() if c1 else None
    g, b = torch.rand(c, device=dev) + 0.5, torch.rand(c, device=dev) - 0.5
    o = torch.empty((nb, hw, c), device=dev, dtype=torch.bfloat16)
    res, tab = {}, {}
    row = []
    for slab in (0, 1):
        hip.tune("gn_slab", slab)
        row.append(t_us(lambda: hip.groupnorm(x0, g, b, 32, 1e-5, silu=True, x1=x1, out=o)))
        res[slab] = o.clone()
        row.append(t_us(lambda: hip.groupnorm_affine(x0, g, b, 32, 1e-5, x1=x1)))
        tab[slab] = torch.stack(hip.groupnorm_affine(x0, g, b, 32, 1e-5, x1=x1)).clone()
    hip.tune("gn_slab", 1)
    d = (res[0].float() - res[1].float()).abs().max().item()
    dt = (tab[0] - tab[1]).abs().max().item()
    print(f"{nb} x {hw} x ({c0} + {c1})".ljust(28) + f" {row[0]:8.1f} {row[2]:8.1f}   {row[1]:14.1f} {row[3]:8.1f}   {2 * nb * hw * c * 2 / row[2] / 1e6:5.2f}   {d:.2e} / tables {dt:.2e}",
          flush=True)
