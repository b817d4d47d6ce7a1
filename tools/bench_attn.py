#!/usr/bin/env python3
"""Times the spatial self-attention kernel at the step's shapes (HIP events, min / median of 5 rounds).   python tools/bench_attn.py"""
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mmgt_amd import hip  # noqa: E402

dev = torch.device("cuda:0")


def case(n, c, nk2, frames=24):
    heads = 8
    hd = c // heads
    B = 2 * frames
    qk = torch.randn(B * n, 2 * c, device=dev).bfloat16()
    vt = torch.randn(B, c, n, device=dev).bfloat16()
    kb = torch.randn(2, max(nk2, 8), c, device=dev).bfloat16()
    vbt = torch.randn(2, c, max(nk2, 8), device=dev).bfloat16()
    o = torch.empty((B * n, c), device=dev, dtype=torch.bfloat16)
    kw = dict(k2=kb, v2=vbt, k2_str=(kb.stride(0), kb.stride(1)), v2_str=(vbt.stride(0), vbt.stride(1)), k2_bdiv=frames, nk2=nk2,
              seg2_first_batch=frames) if nk2 else {}

    def run():
        hip.attention(qk, qk[:, c:], vt, o, batch=B, heads=heads, hd=hd, nq=n, nk=n, scale=hd ** -0.5, q_str=(n * 2 * c, 0, 2 * c),
                      k_str=(n * 2 * c, 0, 2 * c), v_str=(c * n, 0, n), o_str=(n * c, 0, c), v_transposed=True, **kw)
    fl = 4.0 * heads * hd * n * (B * n + frames * nk2)
    return f"attn n={n} hd={hd} nk2={nk2}", run, fl, o


def t_us(fn, reps=3):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


if __name__ == "__main__":
    for name, run, fl, o in [case(4096, 320, 4096), case(4096, 320, 0), case(1024, 640, 1024), case(1024, 640, 0), case(256, 1280, 256)]:
        outs = {}
        for a64 in (0, 1):                              # 0: 32 queries per wave (attention.hip); 1: 64 per wave, LDS-DMA staged (attn64d_kernel)
            hip.tune("attn64", a64)
            t_us(run)
            ts = [t_us(run) for _ in range(5)]
            outs[a64] = o.float().clone()
            print(f"{name:32s} attn64={a64} | {min(ts):8.1f}/{statistics.median(ts):8.1f} us {fl / min(ts) / 1e6:5.0f} TF", flush=True)
        print(f"    max|attn64 - base| = {(outs[1] - outs[0]).abs().max().item():.3e}", flush=True)
    hip.tune("attn64", 1)
    # the head_dim-40 kernel with (attn_nomax = 0) and without (1, the default) the running maximum, alternating
    for name, run, fl, o in [case(4096, 320, 4096), case(4096, 320, 0)]:
        outs = {}
        for rnd_ in range(3):
            for nm in (0, 1):
                hip.tune("attn_nomax", nm)
                t_us(run)
                ts = [t_us(run) for _ in range(3)]
                outs[nm] = o.float().clone()
                print(f"{name:32s} attn_nomax={nm} | {min(ts):8.1f}/{statistics.median(ts):8.1f} us {fl / min(ts) / 1e6:5.0f} TF", flush=True)
        print(f"    max|nomax - running max| = {(outs[1] - outs[0]).abs().max().item():.3e}", flush=True)
    hip.tune("attn_nomax", 1)
    # head_dim 80 (the 32 x 32 level): attention.hip's register-staged kernel (attn80 = 0) against csrc/attn80.hip (1, the default), alternating
    for name, run, fl, o in [case(1024, 640, 1024), case(1024, 640, 0)]:
        outs = {}
        for rnd_ in range(3):
            for a80 in (0, 1):
                hip.tune("attn80", a80)
                t_us(run)
                ts = [t_us(run) for _ in range(3)]
                outs[a80] = o.float().clone()
                print(f"{name:32s} attn80={a80} | {min(ts):8.1f}/{statistics.median(ts):8.1f} us {fl / min(ts) / 1e6:5.0f} TF", flush=True)
        print(f"    max|attn80 - base| = {(outs[1] - outs[0]).abs().max().item():.3e}", flush=True)
    hip.tune("attn80", 1)
