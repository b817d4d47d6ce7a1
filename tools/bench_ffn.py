#!/usr/bin/env python3
"""Time of the fused LayerNorm -> FeedForward(GEGLU) -> + residual launch (csrc/ffn.hip) against the three launches it replaces,
at the level-0 step shape (M = 196 608, C = 320), interleaved in one process."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools import abl_lib  # noqa: E402
abl_lib.use()             # the timing ablations live in libmmgt_hip_abl.so only (make abl); the product library refuses their keys
from mmgt_amd import hip  # noqa: E402
from mmgt_amd.packing import pack_ff_fused, pack_ff_proj_out, pack_geglu  # noqa: E402
from mmgt_amd.synthetic import hash_uniform  # noqa: E402


def t_ms(fn, reps=20):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def main():
    dev = "cuda:0"
    M, C, INNER = int(sys.argv[1]) if len(sys.argv) > 1 else 48 * 4096, 320, 1280
    bf = lambda t: t.to(torch.bfloat16)
    x = bf(hash_uniform("ffn.step.x", (M, C), 1.5, dev))
    g, b = (1 + 0.2 * hash_uniform("ffn.g", (C,), 1.0, dev)), 0.1 * hash_uniform("ffn.b", (C,), 1.0, dev)
    w1 = bf(hash_uniform("ffn.w1", (2 * INNER, C), 1.0, dev) * C ** -0.5)
    b1 = 0.1 * hash_uniform("ffn.b1", (2 * INNER,), 1.0, dev)
    w2 = bf(hash_uniform("ffn.w2", (C, INNER), 1.0, dev) * INNER ** -0.5)
    b2 = 0.1 * hash_uniform("ffn.b2", (C,), 1.0, dev)
    img = pack_ff_fused(w1, b1, w2)
    wp, bp = pack_geglu(w1, b1)
    wp, bp = wp.contiguous(), bp.contiguous()
    out = torch.empty_like(x)
    fused = lambda: hip.ff_fused(x, g, b, img, b2, x, INNER, out=out)
    three = lambda: hip.gemm(hip.gemm(hip.layernorm(x, g, b, 1e-5), wp, bp, act=hip.ACT_GEGLU), w2, b2, residual=x)
    fl = 2.0 * M * (2 * INNER * C + C * INNER)
    wpo = bf(hash_uniform("ffn.wpo", (C, C), 1.0, dev) * C ** -0.5)
    bpo = 0.1 * hash_uniform("ffn.bpo", (C,), 1.0, dev)
    res2 = bf(hash_uniform("ffn.res2", (M, C), 1.0, dev))
    imgpo = pack_ff_proj_out(wpo)
    out2 = torch.empty_like(x)
    two = lambda: hip.gemm(hip.ff_fused(x, g, b, img, b2, x, INNER, out=out), wpo, bpo, residual=res2, out=out2)
    po = lambda: hip.ff_fused_po(x, g, b, img, b2, x, INNER, imgpo, bpo, res2, out=out2)
    for rnd in range(3):
        t2, tp = t_ms(two), t_ms(po)
        print(f"round {rnd}: ff_fused + proj_out GEMM (+res) {t2 * 1e3:7.1f} us | ff_fused_po {tp * 1e3:7.1f} us", flush=True)
    for rnd in range(3):
        t4v, t3 = t_ms(fused), t_ms(three)
        print(f"round {rnd}: fused {t4v * 1e3:7.1f} us = {fl / t4v / 1e9:6.0f} TFLOP/s | LN + ff1 + ff2 {t3 * 1e3:7.1f} us = {fl / t3 / 1e9:6.0f} TFLOP/s", flush=True)
    for dbg, what in ((1, "compute stream alone (no weight fetch)"), (2, "weight stream alone (no MFMA / GELU)")):
        hip.lib().mmgt_tune(b"ffn_dbg", dbg)
        print(f"ffn_dbg {dbg}: {t_ms(fused) * 1e3:7.1f} us   {what}", flush=True)
    hip.lib().mmgt_tune(b"ffn_dbg", 0)
