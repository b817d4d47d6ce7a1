#!/usr/bin/env python3
"""Timing ablations of attn64p_kernel at the in-step bank shape (results of the ablated builds are garbage; only the time is read):
which part of a tile's 3400 cycles is the barrier, the staging, the exponentials, the maxima, the LDS fragment reads; and one workgroup per CU.
    python tools/abl_attn64.py"""
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mmgt_amd import hip  # noqa: E402
from tools.bench_attn import case, t_us  # noqa: E402

if __name__ == "__main__":
    name, run, fl, o = case(4096, 320, 4096)
    rows = [("attn64_kernel (phased)", dict(attn64_ver=1)), ("attn64p (pipelined)", dict(attn64_ver=2)), ("attn64d (pipelined, LDS-DMA)", dict(attn64_ver=3))]
    rows += [(f"attn64p abl={a:2d} ({d})", dict(attn64_ver=2, attn64_abl=a)) for a, d in
             ((1, "no barrier"), (2, "no commit / prefetch"), (32, "no global loads"), (64, "no LDS writes"), (3, "neither"), (4, "v_mul for v_exp"), (8, "no maxima"), (16, "no LDS reads"),
              (19, "no barrier, staging, LDS reads"), (31, "MFMA + cvt only"))]
    rows += [("attn64p, 8 waves per workgroup", dict(attn64_nw=8)), ("attn64p, 8 waves, no commit / prefetch", dict(attn64_nw=8, attn64_abl=2))]
    rows += [("attn64p, one workgroup per CU", dict(attn64_ver=2, attn64_pad=64 * 1024)), ("attn64_kernel, one workgroup per CU", dict(attn64_ver=1, attn64_pad=96 * 1024))]
    for label, knobs in rows:
        for k, v in {**dict(attn64_ver=2, attn64_abl=0, attn64_pad=0, attn64_nw=4), **knobs}.items():
            hip.tune(k, v)
        t_us(run)
        ts = [t_us(run) for _ in range(5)]
        print(f"{label:45s} {min(ts):8.1f} / {statistics.median(ts):8.1f} us", flush=True)
    for k, v in dict(attn64_ver=2, attn64_abl=0, attn64_pad=0, attn64_nw=4).items():
        hip.tune(k, v)
