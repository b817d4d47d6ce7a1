#!/usr/bin/env python3
"""Calibration only: achievable HBM rates on this device for pure writes (fill), pure reads (sum) and copies."""
import torch

dev = torch.device("cuda:0")


def t(fn, reps=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e-3


for mb in (126, 377, 1024):
    n = mb * 1000 * 1000 // 2
    x = torch.empty(n, device=dev, dtype=torch.bfloat16)
    y = torch.empty(n, device=dev, dtype=torch.bfloat16)
    x.normal_()
    tw = t(lambda: y.zero_())
    tc = t(lambda: y.copy_(x))
    tr = t(lambda: x.view(torch.int16).max())
    ta = t(lambda: torch.add(x, x, out=y))
    print(f"{mb:5d} MB: fill {mb/tw/1e6:5.2f} TB/s | copy {2*mb/tc/1e6:5.2f} TB/s (r+w) | read-reduce {mb/tr/1e6:5.2f} TB/s | add {2*mb/ta/1e6:5.2f} TB/s")
