#!/usr/bin/env python3
"""Times the motion modules' temporal attention at the step's shapes (HIP events): python tools/bench_tattn.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mmgt_amd import hip  # noqa: E402

dev = torch.device("cuda:0")
for n, c in [(4096, 320), (1024, 640), (256, 1280), (64, 1280)]:
    b, frames, heads = 2, 24, 8
    hd = c // heads
    m = b * frames * n
    qkv = torch.randn(m, 3 * c, device=dev).bfloat16()
    o = torch.empty((m, c), device=dev, dtype=torch.bfloat16)
    st = (frames * n * 3 * c, 3 * c, n * 3 * c)

    def run():
        hip.attention(qkv, qkv[:, c:], qkv[:, 2 * c:], o, batch=b * n, heads=heads, hd=hd, nq=frames, nk=frames, scale=hd ** -0.5,
                      q_str=st, k_str=st, v_str=st, o_str=(frames * n * c, c, n * c), bdiv=n)
    for _ in range(3):
        run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        run()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    by = (qkv.numel() + o.numel()) * 2
    print(f"temporal attention n={n} c={c}: {us:7.1f} us  {by / us / 1e6:5.2f} TB/s")
