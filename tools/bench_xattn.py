#!/usr/bin/env python3
"""Time of MM-HAA's 24-head audio cross-attention (32 keys) at the step's shapes, heads-inner workgroup order on / off
(mmgt_tune attn_heads_inner): python tools/bench_xattn.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mmgt_amd import hip  # noqa: E402


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


dev = torch.device("cuda:0")
for nb, n, inner, heads in [(48, 4096, 320, 8), (48, 1024, 640, 8), (48, 256, 1280, 8)]:
    hd, k3 = inner // heads, 3 * inner
    q3 = torch.randn(nb * n, k3, device=dev).bfloat16()
    kv = torch.randn(nb * 32, 2 * k3, device=dev).bfloat16()
    rs = torch.rand(3, nb * n, device=dev)
    out = torch.empty(nb * n, k3 + 64, device=dev, dtype=torch.bfloat16)
    res = {}
    for hi in (0, 1, 0, 1):
        hip.tune("attn_heads_inner", hi)
        us = t_us(lambda: hip.attention(q3, kv, kv[:, k3:], out, batch=nb, heads=3 * heads, hd=hd, nq=n, nk=32, scale=hd ** -0.5,
                                        q_str=(n * k3, 0, k3), k_str=(32 * 2 * k3, 0, 2 * k3), v_str=(32 * 2 * k3, 0, 2 * k3),
                                        o_str=(n * (k3 + 64), 0, k3 + 64), out_scale=rs, out_scale_heads=heads))
        res.setdefault(hi, []).append(us)
        if hi == 1:
            ref = out.clone()
        else:
            base = out.clone()
    by = 2.0 * nb * n * k3 * 2
    print(f"audio cross-attention nb={nb} n={n} inner={inner}: row-major pairs {min(res[0]):7.1f} us ({by / min(res[0]) / 1e6:.2f} TB/s) | "
          f"heads inner {min(res[1]):7.1f} us ({by / min(res[1]) / 1e6:.2f} TB/s)   equal: {torch.equal(ref[:, :k3], base[:, :k3])}")
hip.tune("attn_heads_inner", 1)
