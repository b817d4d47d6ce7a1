#!/usr/bin/env python3
"""Start-up stagger sweep for the spatial attention kernel (one process, one device).  python tools/attn_stagger.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mmgt_amd import hip  # noqa: E402

dev = torch.device("cuda:0")


def rnd(*shape):
    return (torch.rand(shape, device=dev) * 2 - 1).bfloat16()


def timeit(fn, reps=4):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


for hd, n in [(40, 4096), (80, 1024)]:
    inner, nb, f = 8 * hd, 48, 24
    qk, vt = rnd(nb * n, 2 * inner), rnd(nb, inner, n)
    kb, vbt = rnd(2, n, inner), rnd(2, inner, n)
    o = torch.empty((nb * n, inner), device=dev, dtype=torch.bfloat16)
    kw = dict(k2=kb, v2=vbt, k2_str=(kb.stride(0), kb.stride(1)), v2_str=(vbt.stride(0), vbt.stride(1)), k2_bdiv=f, nk2=n,
              seg2_first_batch=nb // 2)
    fn = lambda: hip.attention(qk, qk[:, inner:], vt, o, batch=nb, heads=8, hd=hd, nq=n, nk=n, scale=hd ** -0.5,
                               q_str=(n * 2 * inner, 0, 2 * inner), k_str=(n * 2 * inner, 0, 2 * inner),
                               v_str=(inner * n, 0, n), o_str=(n * inner, 0, inner), v_transposed=True, **kw)
    hip.tune("attn_stag_sleep", 0)
    ref = None
    base = min(timeit(fn) for _ in range(3))
    ref = o.clone()
    print(f"hd={hd} N={n}: no stagger {base:8.1f} us")
    for shift in (0, 3, 5, 8, 10):
        cells = []
        for sleep in (2, 4, 6, 8, 12, 16, 24):
            hip.tune("attn_stag_shift", shift)
            hip.tune("attn_stag_sleep", sleep)
            t = min(timeit(fn) for _ in range(3))
            assert torch.equal(o, ref)
            cells.append(f"s{sleep}:{t:7.1f}")
        print(f"   shift {shift:2d}: " + " ".join(cells))
hip.tune("attn_stag_sleep", 0)
