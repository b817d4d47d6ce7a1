#!/usr/bin/env python3
"""Experiment: start-time stagger of gemm16's persistent workgroups (tune key gemm_stagger = 10-ns ticks per phase step)."""
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mmgt_amd import hip  # noqa: E402
from tools.ab_cfg import conv_case, gemm_case, time_call  # noqa: E402

cases = [(16, gemm_case(196608, 2560, 320, act=1)), (16, gemm_case(196608, 1280, 320)), (17, gemm_case(196608, 960, 320, bias=False)),
         (17, gemm_case(196608, 320, 320, res=True)), (17, gemm_case(196608, 320, 1280, res=True)), (16, gemm_case(49152, 5120, 640, act=1)),
         (17, gemm_case(49152, 640, 2560, res=True)), (16, gemm_case(12288, 10240, 1280, act=1)), (17, conv_case(48, 64, 320, 320)),
         (16, conv_case(48, 32, 640, 640))]
stags = [int(x) for x in os.environ.get("STAGS", "0,50,100,150,200,300,400").split(",")]
for cfg, (name, fn, flops, out) in cases:
    hip.tune("gemm_cfg", cfg)
    times = {s: [] for s in stags}
    for r in range(6):
        for s in stags:
            hip.tune("gemm_stagger", s)
            t = time_call(fn)
            if r:
                times[s].append(t)
    print(f"{name:44s} c{cfg} | " + " | ".join(f"s{s}: {min(times[s]):7.1f}" for s in stags), flush=True)
hip.tune("gemm_stagger", 0)
hip.tune("gemm_cfg", 0)
