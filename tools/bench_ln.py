#!/usr/bin/env python3
"""LayerNorm at the step's shapes (and, with a library that has the experiment key ln_groups, against the row groups per wave).   python tools/bench_ln.py"""
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
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / reps * 1e3)
    return best


have = True
try:
    hip.tune("ln_groups", 0)
except Exception:
    have = False                                    # (an older library: one column)
for rows, c in [(196608, 320), (49152, 640), (12288, 1280), (3072, 1280), (98304, 320), (24576, 640)]:
    x = torch.randn(rows, c, device=dev).bfloat16()
    g, b = torch.rand(c, device=dev) + 0.5, torch.rand(c, device=dev) - 0.5
    o = torch.empty_like(x)
    row = []
    for r in ((1, 2, 4) if have else (0,)):
        if have:
            hip.tune("ln_groups", r)
        row.append(t_us(lambda: hip.layernorm(x, g, b, out=o)))
    if have:
        hip.tune("ln_groups", 0)
    print(f"layernorm rows={rows} c={c}: " + "  ".join(f"{t:6.1f} us" for t in row) + f"   ({'groups 1 / 2 / 4' if have else 'library without the key'};"
          f" {2 * x.numel() * 2 / min(row) / 1e6:5.2f} TB/s at the best)", flush=True)
