#!/usr/bin/env python3
"""The fused FeedForward with one persistent workgroup per CU walking the row blocks, the next block's rows and first weights requested in front
of the epilogue (tune key ffn_persist of the experiment build tools/micro/ffn_persistent_r6.patch) against one workgroup per block."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mmgt_amd import hip
from mmgt_amd.packing import pack_ff_fused, pack_ff_proj_out
from mmgt_amd.synthetic import hash_uniform
dev = "cuda:0"
def t_us(fn, reps=20):
    for _ in range(3): fn()
    best = 1e9
    for _ in range(4):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps): fn()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / reps * 1e3)
    return best
for M in (48 * 4096, 24 * 4096):
    C, INNER = 320, 1280
    bf = lambda t: t.to(torch.bfloat16)
    x = bf(hash_uniform("ffn.step.x", (M, C), 1.5, dev))
    g, b = (1 + 0.2 * hash_uniform("ffn.g", (C,), 1.0, dev)), 0.1 * hash_uniform("ffn.b", (C,), 1.0, dev)
    w1 = bf(hash_uniform("ffn.w1", (2 * INNER, C), 1.0, dev) * C ** -0.5)
    b1 = 0.1 * hash_uniform("ffn.b1", (2 * INNER,), 1.0, dev)
    w2 = bf(hash_uniform("ffn.w2", (C, INNER), 1.0, dev) * INNER ** -0.5)
    b2 = 0.1 * hash_uniform("ffn.b2", (C,), 1.0, dev)
    img = pack_ff_fused(w1, b1, w2)
    wpo = bf(hash_uniform("ffn.wpo", (C, C), 1.0, dev) * C ** -0.5)
    bpo = 0.1 * hash_uniform("ffn.bpo", (C,), 1.0, dev)
    res2 = bf(hash_uniform("ffn.res2", (M, C), 1.0, dev))
    imgpo = pack_ff_proj_out(wpo)
    out = torch.empty_like(x)
    outs = {}
    for p in (0, 1, 0, 1):
        try:
            hip.tune("ffn_persist", p)
        except Exception:
            pass                                   # (the product library: no such key, one workgroup per block)
        t1 = t_us(lambda: hip.ff_fused(x, g, b, img, b2, x, INNER, out=out)); o1 = out.clone()
        t2 = t_us(lambda: hip.ff_fused_po(x, g, b, img, b2, x, INNER, imgpo, bpo, res2, out=out)); o2 = out.clone()
        outs[p] = (o1, o2)
        print(f"M={M} ffn_persist={p}: ff_fused {t1:7.1f} us   ff_fused_po {t2:7.1f} us", flush=True)
    print("   bitwise equal:", torch.equal(outs[0][0], outs[1][0]), torch.equal(outs[0][1], outs[1][1]))
