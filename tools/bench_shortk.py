#!/usr/bin/env python3
"""The short-K GEMMs of the step (K = 320 .. 1280 with a residual epilogue: out-projections, proj_out) against the start stagger of gemm16
(tune key g16_stagger: every second CU's workgroup starts late, so that the chip is not in the store / residual phase all at once).
    python3 tools/bench_shortk.py [stagger values ...]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mmgt_amd import hip  # noqa: E402
from tools.ab_cfg import gemm_case, time_call  # noqa: E402

stag = [int(v) for v in sys.argv[1:]] or [0, 4, 8, 12, 16, 24, 32]
cases = [gemm_case(49152, 640, 640, res=True), gemm_case(49152, 640, 640), gemm_case(12288, 1280, 1280, res=True), gemm_case(196608, 320, 320, res=True),
         gemm_case(49152, 640, 2560, res=True), gemm_case(12288, 1280, 5120, res=True), gemm_case(49152, 1920, 640), gemm_case(3072, 1280, 1280, res=True)]
print(f"{'':40s}" + "".join(f"{'stg ' + str(s):>9s}" for s in stag) + "   (us, best of 5 x 20 launches; bytes moved / best time)")
for name, fn, flops, out in cases:
    row = []
    for s in stag:
        hip.tune("g16_stagger", s)
        fn()
        row.append(min(time_call(fn, reps=20) for _ in range(5)))
    hip.tune("g16_stagger", 0)
    print(f"{name:40s}" + "".join(f"{t:9.1f}" for t in row) + f"   {flops / min(row) / 1e6:6.0f} TF/s")
