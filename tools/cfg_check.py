#!/usr/bin/env python3
"""Forced tile configurations must agree with the heuristic one (same fp32 accumulation order per output element up to
the K-chunk order, so bit-equal in practice).   CFGS=9,10 python tools/cfg_check.py"""
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mmgt_amd import hip  # noqa: E402
from mmgt_amd.packing import pack_geglu  # noqa: E402

dev = torch.device("cuda:0")
cfgs = [int(c) for c in os.environ.get("CFGS", "9,10").split(",")]
torch.manual_seed(0)
for M, N, K, epi in [(1000, 320, 320, "res"), (4096, 2560, 320, "geglu"), (777, 1280, 1280, "res"), (512, 512, 128, ""),
                     (300, 200, 64, "res")]:
    a = (torch.rand(M, K, device=dev) - 0.5).bfloat16()
    w = ((torch.rand(N, K, device=dev) - 0.5) / math.sqrt(K)).bfloat16()
    b = torch.rand(N, device=dev) - 0.5
    nout = N // 2 if epi == "geglu" else N
    res = (torch.rand(M, nout, device=dev) - 0.5).bfloat16() if epi == "res" else None
    if epi == "geglu":
        w, b = pack_geglu(w, b)
    act = hip.ACT_GEGLU if epi == "geglu" else 0
    hip.tune("gemm_cfg", 0)
    ref = hip.gemm(a, w, b, residual=res, act=act)
    for c in cfgs:
        hip.tune("gemm_cfg", c)
        out = hip.gemm(a, w, b, residual=res, act=act)
        d = (out.float() - ref.float()).abs().max().item()
        print(f"M={M} N={N} K={K} {epi:5s} cfg{c}: max|d| vs heuristic = {d:.3e}")
        assert d < 2e-2, "mismatch"
hip.tune("gemm_cfg", 0)
print("ok")
