#!/usr/bin/env python3
"""Forced tile configurations must agree with the heuristic one (same fp32 accumulation order per output element up to
the K-chunk order, so bit-equal in practice).   CFGS=9,12 python tools/cfg_check.py"""
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mmgt_amd import hip  # noqa: E402
from mmgt_amd.packing import pack_geglu  # noqa: E402

dev = torch.device("cuda:0")
cfgs = [int(c) for c in os.environ.get("CFGS", "9,12").split(",")]
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
# conv gather and the fp32-I/O instantiation on the same forced configurations
for dt in (torch.bfloat16, torch.float32):
    for nb, h, cin, cout, st, up in [(2, 16, 128, 256, 1, 0), (3, 8, 192, 320, 2, 0), (1, 8, 64, 512, 1, 1)]:
        x = (torch.rand(nb, h, h, cin, device=dev) - 0.5).to(dt)
        w = ((torch.rand(cout, 3, 3, cin, device=dev) - 0.5) / math.sqrt(9 * cin)).to(dt)
        b = torch.rand(cout, device=dev) - 0.5
        hip.tune("gemm_cfg", 0)
        ref = hip.conv3x3(x, w, b, stride=st, upsample=bool(up))
        for c in cfgs:
            hip.tune("gemm_cfg", c)
            out = hip.conv3x3(x, w, b, stride=st, upsample=bool(up))
            d = (out.float() - ref.float()).abs().max().item()
            print(f"conv {dt} nb={nb} h={h} cin={cin} cout={cout} s={st} up={up} cfg{c}: max|d| = {d:.3e}")
            assert d < (2e-2 if dt == torch.bfloat16 else 1e-4), "mismatch"
    a = (torch.rand(700, 192, device=dev) - 0.5).to(dt)
    w = ((torch.rand(520, 192, device=dev) - 0.5) / math.sqrt(192)).to(dt)
    hip.tune("gemm_cfg", 0)
    ref = hip.gemm(a, w, None)
    for c in cfgs:
        hip.tune("gemm_cfg", c)
        d = (hip.gemm(a, w, None).float() - ref.float()).abs().max().item()
        print(f"gemm {dt} 700x520x192 cfg{c}: max|d| = {d:.3e}")
        assert d < (2e-2 if dt == torch.bfloat16 else 1e-4), "mismatch"
hip.tune("gemm_cfg", 0)
print("ok")
