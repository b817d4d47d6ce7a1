#!/usr/bin/env python3
"""A/B of GEMM / conv tile configurations of the CURRENT library in one process on one device (guide rule 24): every case is
run under each forced `gemm_cfg` (0 = the dispatcher's own choice) in interleaved rounds; prints min and median times.

    CFGS=0,9,16 SET=step python tools/ab_cfg.py          # the denoise step's heavy shapes (profiles/r1/opshapes)
    CFGS=9,16 SET=big python tools/ab_cfg.py             # 8192^3 and friends
    AB=g16_pb:1,4,8 SET=step python tools/ab_cfg.py      # another mmgt_tune key instead of gemm_cfg
"""
import math
import os
import statistics
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mmgt_amd import hip  # noqa: E402

dev = torch.device("cuda:0")


def rnd(*shape, s=1.0):
    return (torch.rand(shape, device=dev) * 2 - 1).mul_(s).bfloat16()


def time_call(fn, reps=3):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


def gemm_case(M, N, K, act=0, res=False, bias=True):
    a, w = rnd(M, K), rnd(N, K, s=1 / math.sqrt(K))
    no = N // 2 if act == 1 else N
    b = (torch.rand(N, device=dev) - 0.5) if bias else None
    r = rnd(M, no) if res else None
    o = torch.empty((M, no), device=dev, dtype=torch.bfloat16)
    name = f"gemm M={M} N={N} K={K}" + (" geglu" if act == 1 else "") + (" +res" if res else "")
    return name, (lambda: hip.gemm(a, w, b, out=o, residual=r, act=act)), 2.0 * M * N * K, o


def conv_case(nb, h, cin, cout, up=False):
    from mmgt_amd.packing import pack_conv3x3
    x = rnd(nb, h, h, cin)
    w = pack_conv3x3(torch.randn(cout, cin, 3, 3) / math.sqrt(9 * cin)).to(dev).bfloat16()
    b = torch.rand(cout, device=dev) - 0.5
    ho = 2 * h if up else h
    o = torch.empty((nb, ho, ho, cout), device=dev, dtype=torch.bfloat16)
    return (f"conv nb={nb} h={h} cin={cin} cout={cout}" + (" up" if up else ""),
            (lambda: hip.conv3x3(x, w, b, out=o, upsample=up)), 2.0 * nb * ho * ho * cout * 9 * cin, o)


def wx_case(B, R, ntok, K):
    w, x = rnd(R, K, s=1 / math.sqrt(K)), rnd(B, ntok, K)
    o = torch.empty((B, R, ntok), device=dev, dtype=torch.bfloat16)
    return f"gemm_wx B={B} R={R} ntok={ntok} K={K}", (lambda: hip.gemm_batched_wx(w, x, out=o)), 2.0 * B * R * ntok * K, o


SETS = {
    "l3": lambda: [gemm_case(3072, 1280, 1280, res=True), gemm_case(3072, 1280, 1280), gemm_case(3072, 3840, 1280, bias=False),
                   gemm_case(3072, 1280, 5120, res=True), gemm_case(3072, 2560, 1280), gemm_case(3072, 10240, 1280, act=1),
                   gemm_case(49152, 640, 640), gemm_case(12288, 1280, 1280)],
    "l3small": lambda: [gemm_case(3072, 1280, 1280, res=True), gemm_case(3072, 3840, 1280, bias=False), gemm_case(3072, 1280, 5120, res=True),
                        gemm_case(3072, 10240, 1280, act=1), gemm_case(1536, 1280, 1280, res=True), gemm_case(1536, 3840, 1280, bias=False),
                        gemm_case(1536, 1280, 5120, res=True), gemm_case(1536, 10240, 1280, act=1), gemm_case(6144, 1280, 1280, res=True),
                        gemm_case(12288, 1280, 1280, res=True), gemm_case(24576, 640, 640, res=True)],
    "wx": lambda: [wx_case(48, 320, 4096, 320), wx_case(48, 640, 1024, 640), wx_case(48, 1280, 256, 1280)],
    "step2": lambda: [
        gemm_case(196608, 2560, 320, act=1), gemm_case(49152, 640, 2560, res=True), gemm_case(49152, 640, 640, res=True),
        gemm_case(49152, 640, 640), gemm_case(49152, 1280, 640), gemm_case(49152, 1920, 640, bias=False),
        gemm_case(12288, 1280, 1280), gemm_case(12288, 2560, 1280), gemm_case(12288, 1280, 1280, res=True),
        gemm_case(3072, 10240, 1280, act=1), gemm_case(3072, 1280, 1280, res=True), gemm_case(3072, 3840, 1280, bias=False),
        gemm_case(3072, 1280, 5120, res=True), gemm_case(196608, 320, 320), gemm_case(196608, 320, 640), gemm_case(196608, 640, 320, bias=False),
        conv_case(48, 8, 1280, 1280), conv_case(48, 16, 640, 1280), conv_case(48, 32, 320, 640)],
    "geglu": lambda: [gemm_case(196608, 2560, 320, act=1), gemm_case(196608, 1280, 320), gemm_case(196608, 2560, 320),
                      gemm_case(196608, 2560, 640, act=1), gemm_case(196608, 2560, 1280, act=1), gemm_case(196608, 1280, 640)],
    # the reference's shipped window: context_frames = 12 (pipeline_pose2vid_long.py:360-362) -> 24 images per CFG forward
    "ctx12": lambda: [
        gemm_case(98304, 320, 320, res=True), gemm_case(98304, 320, 320), gemm_case(98304, 960, 320, bias=False), gemm_case(98304, 640, 320, bias=False),
        gemm_case(24576, 5120, 640, act=1), gemm_case(24576, 640, 2560, res=True), gemm_case(24576, 640, 640, res=True), gemm_case(24576, 1920, 640, bias=False),
        gemm_case(6144, 10240, 1280, act=1), gemm_case(6144, 1280, 5120, res=True), gemm_case(6144, 1280, 1280, res=True), gemm_case(6144, 3840, 1280, bias=False),
        gemm_case(1536, 1280, 1280, res=True), gemm_case(1536, 1280, 5120, res=True), gemm_case(1536, 10240, 1280, act=1),
        conv_case(24, 64, 320, 320), conv_case(24, 64, 640, 320), conv_case(24, 32, 640, 640), conv_case(24, 32, 1920, 640),
        conv_case(24, 16, 1280, 1280), conv_case(24, 16, 2560, 1280), conv_case(24, 8, 1280, 1280), conv_case(24, 8, 2560, 1280)],
    # the VAE decoder at 8 frames of 512 x 512 (tools/profile_vae.py)
    "vae": lambda: [conv_case(8, 512, 128, 128), conv_case(8, 512, 256, 128), conv_case(8, 512, 128, 64), conv_case(8, 256, 256, 256),
                    conv_case(8, 256, 512, 256), conv_case(8, 128, 512, 512), conv_case(8, 64, 512, 512), conv_case(8, 256, 256, 256, up=True),
                    gemm_case(2097152, 128, 256), gemm_case(524288, 256, 512)],
    "g16s": lambda: [gemm_case(49152, 5120, 640, act=1), gemm_case(196608, 960, 320, bias=False), gemm_case(12288, 3840, 1280, bias=False),
                     conv_case(48, 64, 320, 320)],
    # shapes whose 256 x 320 tiling leaves a quarter of the CUs idle (192 tiles) or runs 1.5 rounds (384 tiles)
    "quant": lambda: [gemm_case(12288, 1280, 1280, res=True), gemm_case(12288, 1280, 5120, res=True), gemm_case(12288, 1280, 1280),
                      gemm_case(49152, 640, 640, res=True), gemm_case(49152, 640, 2560, res=True), gemm_case(49152, 640, 640),
                      gemm_case(12288, 2560, 1280), gemm_case(49152, 1280, 640)],
    # conv against the dense GEMM of the same M x N x K (what the gather itself costs)
    "convdense": lambda: [conv_case(48, 64, 320, 320), gemm_case(196608, 320, 2880), conv_case(48, 64, 640, 320), gemm_case(196608, 320, 5760),
                          conv_case(48, 32, 640, 640), gemm_case(49152, 640, 5760), conv_case(48, 16, 1280, 1280), gemm_case(12288, 1280, 11520)],
    # round 5: candidates of the 192 x 320 tile (cfg 19) at 24 and 12 frames
    "bm192": lambda: [gemm_case(49152, 640, 640, res=True), gemm_case(49152, 640, 2560, res=True), gemm_case(49152, 640, 640), gemm_case(49152, 640, 1280),
                      gemm_case(49152, 640, 320, res=True), gemm_case(12288, 1280, 1280, res=True), gemm_case(12288, 1280, 5120, res=True),
                      gemm_case(49152, 1280, 640), gemm_case(196608, 320, 320, res=True), gemm_case(196608, 960, 320, bias=False),
                      conv_case(48, 32, 640, 640), conv_case(48, 32, 1280, 640), conv_case(48, 32, 1920, 640), conv_case(48, 32, 320, 640),
                      conv_case(48, 64, 320, 320), conv_case(48, 64, 640, 320),
                      gemm_case(98304, 320, 320, res=True), gemm_case(24576, 640, 640, res=True), gemm_case(24576, 640, 2560, res=True),
                      conv_case(24, 32, 640, 640), conv_case(24, 64, 320, 320)],
    # the 256-column family (gemm16v candidates): GEGLU, q|k|v, long-K + residual at N = 1280, convs to 1280 / 640, 8192^3
    "v256": lambda: [gemm_case(49152, 5120, 640, act=1), gemm_case(12288, 10240, 1280, act=1), gemm_case(196608, 2560, 320, act=1),
                     gemm_case(12288, 3840, 1280, bias=False), gemm_case(49152, 1920, 640, bias=False), gemm_case(12288, 1280, 5120, res=True),
                     gemm_case(12288, 1280, 1280, res=True), gemm_case(49152, 1280, 640), gemm_case(8192, 8192, 8192, bias=False),
                     conv_case(48, 16, 1280, 1280), conv_case(48, 16, 2560, 1280), conv_case(48, 32, 640, 640), conv_case(48, 16, 1280, 1280, up=True)],
    "v3": lambda: [gemm_case(49152, 5120, 640, act=1), gemm_case(12288, 3840, 1280, bias=False), gemm_case(8192, 8192, 8192, bias=False), conv_case(48, 16, 1280, 1280)],
    "big": lambda: [gemm_case(8192, 8192, 8192, bias=False), gemm_case(4096, 4096, 4096, bias=False)],
    "step": lambda: [
        gemm_case(196608, 2560, 320, act=1), gemm_case(49152, 5120, 640, act=1), gemm_case(12288, 10240, 1280, act=1),
        gemm_case(196608, 320, 1280, res=True), gemm_case(49152, 640, 2560, res=True), gemm_case(12288, 1280, 5120, res=True),
        gemm_case(196608, 960, 320, bias=False), gemm_case(49152, 1920, 640, bias=False), gemm_case(12288, 3840, 1280, bias=False),
        gemm_case(196608, 320, 320, res=True), gemm_case(49152, 640, 640, res=True), gemm_case(12288, 1280, 1280, res=True),
        gemm_case(196608, 640, 320, bias=False),
        conv_case(48, 64, 320, 320), conv_case(48, 64, 640, 320), conv_case(48, 32, 640, 640), conv_case(48, 32, 1920, 640),
        conv_case(48, 16, 1280, 1280), conv_case(48, 16, 2560, 1280), conv_case(48, 8, 1280, 1280), conv_case(48, 8, 2560, 1280),
        conv_case(48, 16, 1280, 1280, up=True), conv_case(48, 32, 640, 640, up=True)],
}

if __name__ == "__main__":
    cfgs = [int(c) for c in os.environ.get("CFGS", "0,16").split(",")]
    key = "gemm_cfg"
    if os.environ.get("AB"):                       # AB=g16_pb:1,4,8 -> A/B over another mmgt_tune key instead of the tile configuration
        key, vals = os.environ["AB"].split(":")
        cfgs = [int(c) for c in vals.split(",")]
    rounds = int(os.environ.get("ROUNDS", "5"))
    for name, fn, flops, out in SETS[os.environ.get("SET", "step")]():
        times = {c: [] for c in cfgs}
        outs = {}
        for r in range(rounds + 1):
            for c in cfgs:
                hip.tune(key, c)
                t = time_call(fn)
                if r:
                    times[c].append(t)
                else:
                    outs[c] = out.float().clone()
        ref = outs[cfgs[0]]
        cells = []
        for c in cfgs:
            d = (outs[c] - ref).abs().max().item()
            mn, md = min(times[c]), statistics.median(times[c])
            cells.append(f"c{c}: {mn:7.1f}/{md:7.1f}us {flops / mn / 1e6:5.0f}TF d={d:.1e}")
        print(f"{name:44s} | " + " | ".join(cells), flush=True)
    hip.tune(key, {"gemm_cfg": 0}.get(key, -1))
