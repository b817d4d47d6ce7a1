#!/usr/bin/env python3
"""Time of mmgt_rowgemm320 (csrc/rowgemm.hip) against the launches it replaces at the level-0 step shape (M = 196 608, K = 320)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mmgt_amd import hip  # noqa: E402
from mmgt_amd.packing import pack_rowgemm  # noqa: E402
from mmgt_amd.synthetic import hash_uniform  # noqa: E402


def t_us(fn, reps=20):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return 1e3 * e0.elapsed_time(e1) / reps


def main():
    dev = "cuda:0"
    nb, n, C = 48, 4096, 320
    M = nb * n
    bf = lambda t: t.to(torch.bfloat16)
    x = bf(hash_uniform("rg.step.x", (M, C), 1.5, dev))
    g, b = (1 + 0.2 * hash_uniform("rg.g", (C,), 1.0, dev)), 0.1 * hash_uniform("rg.b", (C,), 1.0, dev)
    w3 = bf(hash_uniform("rg.w3", (3 * C, C), 1.0, dev) * C ** -0.5)
    wqk, wv = w3[:2 * C].contiguous(), w3[2 * C:].contiguous()
    wo = bf(hash_uniform("rg.wo", (C, C), 1.0, dev) * C ** -0.5)
    bo = 0.1 * hash_uniform("rg.bo", (C,), 1.0, dev)
    img3, imgo, imgq = pack_rowgemm(w3), pack_rowgemm(wo), pack_rowgemm(wo)
    vt = torch.empty((nb, C, n), device=dev, dtype=torch.bfloat16)
    qk = torch.empty((M, 2 * C), device=dev, dtype=torch.bfloat16)
    qkv = torch.empty((M, 3 * C), device=dev, dtype=torch.bfloat16)
    o = torch.empty((M, C), device=dev, dtype=torch.bfloat16)
    res = bf(hash_uniform("rg.step.res", (M, C), 1.0, dev))

    def old_attn1():
        n1 = hip.layernorm(x, g, b, 1e-5)
        hip.gemm(n1, wqk, out=qk)
        hip.gemm_batched_wx(wv, n1.view(nb, n, C), out=vt)
    cases = [
        ("LN -> q|k (640) + V^T (320)", old_attn1, lambda: hip.rowgemm320(x, img3, 3 * C, ln_gamma=g, ln_beta=b, n1=2 * C, n_tok=n, out=qk, out_t=vt), 2 * M * C * 3 * C),
        ("LN -> qkv (960)", lambda: hip.gemm(hip.layernorm(x, g, b, 1e-5), w3, out=qkv), lambda: hip.rowgemm320(x, img3, 3 * C, ln_gamma=g, ln_beta=b, out=qkv), 2 * M * C * 3 * C),
        ("LN -> q (320)", lambda: hip.gemm(hip.layernorm(x, g, b, 1e-5), wo, out=o), lambda: hip.rowgemm320(x, imgq, C, ln_gamma=g, ln_beta=b, out=o), 2 * M * C * C),
        ("out-proj (320) + residual", lambda: hip.gemm(x, wo, bo, residual=res, out=o), lambda: hip.rowgemm320(x, imgo, C, bo, residual=res, out=o), 2 * M * C * C),
        ("q (320), no LN", lambda: hip.gemm(x, wo, out=o), lambda: hip.rowgemm320(x, imgq, C, out=o), 2 * M * C * C),
    ]
    for rnd in range(1):
        for name, old, new, fl in cases:
            to, tn = t_us(old), t_us(new)
            print(f"round {rnd}: {name:32s} old {to:7.1f} us | rowgemm {tn:7.1f} us = {fl / tn / 1e6:6.0f} TFLOP/s", flush=True)
    for dbg, what in ((1, "no weight fetch"), (2, "no MFMA"), (3, "no stores (non-residual kernels)"), (4, "no DMA wait / barrier (non-residual kernels)")):
        hip.lib().mmgt_tune(b"rowgemm_dbg", dbg)
        print(f"rowgemm_dbg {dbg} ({what}): " + "  ".join(f"{t_us(c[2]):7.1f}" for c in cases), flush=True)
    hip.lib().mmgt_tune(b"rowgemm_dbg", 0)


if __name__ == "__main__":
    main()
