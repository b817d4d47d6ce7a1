"""Time the fused temporal-attention leg (csrc/tleg.hip) against the three launches it replaces at the in-step shapes.
    python tools/bench_tleg.py [frames]      (24: BASELINE config 2; 12: the reference's shipped window)"""
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools import abl_lib  # noqa: E402
abl_lib.use()             # the timing ablations live in libmmgt_hip_abl.so only (make abl); the product library refuses their keys
from mmgt_amd import hip  # noqa: E402
from mmgt_amd.packing import pack_rowgemm, pack_tleg  # noqa: E402
from mmgt_amd.synthetic import hash_uniform  # noqa: E402

DEV, C, H, HD = "cuda:0", 320, 8, 40


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
    return e0.elapsed_time(e1) / reps * 1e3


if __name__ == "__main__":
    F = int(sys.argv[1]) if len(sys.argv) > 1 else 24
    B, n = 2, 4096
    M = B * F * n
    bf = lambda t: t.to(torch.bfloat16)
    x = bf(hash_uniform("bt.x", (M, C), 1.5, DEV))
    w = {k: bf(hash_uniform("bt." + k, (C, C), 1.0, DEV) * C ** -0.5) for k in "qkvo"}
    g, bpe, bo = 1 + 0.2 * hash_uniform("bt.g", (C,), 1.0, DEV), 0.3 * hash_uniform("bt.bpe", (32, C), 1.0, DEV), 0.1 * hash_uniform("bt.bo", (C,), 1.0, DEV)
    img3 = pack_rowgemm(torch.cat([w["q"], w["k"], w["v"]]))
    img = pack_tleg(w["q"], w["k"], w["v"], w["o"])
    o = torch.empty((M, C), device=DEV, dtype=torch.bfloat16)
    out = torch.empty_like(x)
    st = (F * n * 3 * C, 3 * C, n * 3 * C)
    scale = HD ** -0.5

    def three():
        qkv, _ = hip.rowgemm320(x, img3, 3 * C, ln_gamma=g, ln_beta=bpe, pe_div=n, pe_mod=F)
        hip.attention(qkv, qkv[:, C:], qkv[:, 2 * C:], o, batch=B * n, heads=H, hd=HD, nq=F, nk=F, scale=scale, q_str=st, k_str=st, v_str=st,
                      o_str=(F * n * C, C, n * C), bdiv=n)
        return hip.gemm(o, w["o"], bo, residual=x, out=out)

    fused = lambda: hip.temporal_leg320(x, g, bpe, img, bo, B, F, n, scale, out=out)
    fl = 2.0 * M * C * 4 * C + 4.0 * M * H * HD * F
    for rnd in range(3):
        t3 = [t_us(three) for _ in range(3)]
        tf = [t_us(fused) for _ in range(3)]
        print(f"F={F} M={M}: three launches {min(t3):7.1f}/{statistics.median(t3):7.1f} us | fused leg {min(tf):7.1f}/{statistics.median(tf):7.1f} us "
              f"= {fl / min(tf) / 1e6:5.0f} TFLOP/s, {3 * M * C * 2 / min(tf) / 1e6:5.2f} TB/s of its 3 x {M * C * 2 / 1e6:.0f} MB", flush=True)
    if F == 24:
        print("timing ablations (mmgt_tune tleg_abl; results are garbage):")
        base = min(t_us(fused) for _ in range(3))
        for bit, what in ((1, "no LayerNorm arithmetic"), (2, "no attention arithmetic"), (4, "no weight DMA"), (8, "no epilogue loads / stores"),
                          (16, "no projection MFMAs"), (32, "no hand-over wait / barrier"), (128, "hand-over waits but no barrier"), (36, "no weight DMA, no wait / barrier"), (64, "no row loads after the first task")):
            hip.tune("tleg_abl", bit)
            t = min(t_us(fused) for _ in range(3))
            print(f"  abl {bit:3d} {what:36s} {t:7.1f} us  ({t - base:+6.1f})", flush=True)
        hip.tune("tleg_abl", 0)
