#!/usr/bin/env python3
"""GroupNorm + SiLU + conv3x3 of the UNet's resnets: the fused launch (csrc/rconv.hip, behind the statistics pass) against the pair it replaces
(hip.groupnorm(silu) -> hip.conv3x3 on gemm16) at the in-step shapes.   python tools/bench_rconv.py [reps]"""
import math
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
ABL = os.environ.get("ABL")                # "1,2,4,8,16,...": timing ablations of the fused launch (libmmgt_hip_abl.so; results are garbage)
if ABL:
    from tools import abl_lib  # noqa: E402
    abl_lib.use()
from mmgt_amd import hip  # noqa: E402
from mmgt_amd.packing import pack_conv3x3, pack_rconv  # noqa: E402

dev = torch.device("cuda:0")
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10


def t_us(fn):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


SHAPES = [(48, 64, 320, 0, 320), (48, 64, 320, 320, 320), (48, 64, 640, 320, 320), (24, 64, 320, 0, 320),
          (48, 32, 640, 0, 640), (48, 32, 640, 320, 640), (48, 32, 640, 640, 640), (48, 32, 1280, 640, 640), (48, 32, 320, 0, 640),
          (48, 16, 1280, 0, 1280), (48, 16, 1280, 1280, 1280), (48, 16, 1280, 640, 1280), (48, 16, 640, 0, 1280)]
if os.environ.get("SHAPES"):
    SHAPES = [tuple(int(v) for v in s.split("x")) for s in os.environ["SHAPES"].split(",")]
g = torch.Generator(device="cpu").manual_seed(1)
print(f"{'nb x H^2: C0 + C1 -> Cout':32s} {'GN+SiLU':>8s} {'conv':>8s} {'pair':>8s} | {'stats':>7s} {'fused':>8s} {'sum':>8s}  TF/s pair -> fused (conv alone -> fused launch)")
for nb, H, c0, c1, cout in SHAPES:
    cin = c0 + c1
    x0 = (torch.randn((nb, H, H, c0), generator=g) * 1.5).to(dev).bfloat16()
    x1 = torch.randn((nb, H, H, c1), generator=g).to(dev).bfloat16() if c1 else None
    w = torch.randn((cout, cin, 3, 3), generator=g) / math.sqrt(9 * cin)
    gamma, beta = torch.rand(cin).to(dev) + 0.5, torch.rand(cin).to(dev) - 0.5
    b = torch.rand(cout).to(dev)
    temb = torch.rand((2, cout)).to(dev)
    r = torch.randn((nb, H, H, cout), generator=g).to(dev).bfloat16()
    wp = pack_conv3x3(w).to(dev).bfloat16()
    wimg = pack_rconv(w.to(dev))
    v0 = x0.view(nb, H * H, c0)
    v1 = None if x1 is None else x1.view(nb, H * H, c1)
    hdn = hip.groupnorm(v0, gamma, beta, 32, 1e-5, silu=True, x1=v1)
    out = torch.empty((nb, H, H, cout), device=dev, dtype=torch.bfloat16)
    t_gn = t_us(lambda: hip.groupnorm(v0, gamma, beta, 32, 1e-5, silu=True, x1=v1, out=hdn))
    t_cv = t_us(lambda: hip.conv3x3(hdn.view(nb, H, H, cin), wp, b, bias2=temb, bias2_rows=nb // 2 * H * H, residual=r, out=out))
    sc, sh = hip.groupnorm_affine(v0, gamma, beta, 32, 1e-5, x1=v1)
    t_st = t_us(lambda: hip.groupnorm_affine(v0, gamma, beta, 32, 1e-5, x1=v1))
    t_f = t_us(lambda: hip.gn_silu_conv3x3_unet(x0, sc, sh, wimg, cout, b, temb, nb // 2, r, x1=x1, out=out))
    fl = 2.0 * nb * H * H * cout * 9 * cin
    if os.environ.get("ST"):
        g2, b2 = torch.rand(cout).to(dev) + 0.5, torch.rand(cout).to(dev) - 0.5
        t_a = t_us(lambda: hip.gn_silu_conv3x3_unet(x0, sc, sh, wimg, cout, b, temb, nb // 2, r, x1=x1, out=out))
        t_p = t_us(lambda: hip.groupnorm_affine(out.view(nb, H * H, cout), g2, b2, 32, 1e-5))
        t_s = t_us(lambda: hip.gn_silu_conv3x3_unet(x0, sc, sh, wimg, cout, b, temb, nb // 2, r, x1=x1, out=out, next_norm=(g2, b2, 32, 1e-5)))
        print(f"    launch {t_a:7.1f} + statistics pass over its output {t_p:6.1f} = {t_a + t_p:7.1f}   |   launch with epilogue statistics + fold {t_s:7.1f}")
    if os.environ.get("CB"):
        row = []
        for v in [int(x) for x in os.environ["CB"].split(",")]:
            if v and cout % v:
                continue
            hip.tune("rconv_cb", v)
            row.append(f"cb {v}: {t_us(lambda: hip.gn_silu_conv3x3_unet(x0, sc, sh, wimg, cout, b, temb, nb // 2, r, x1=x1, out=out)):7.1f}")
        hip.tune("rconv_cb", 0)
        print("    " + "  ".join(row))
    if os.environ.get("STAGGER"):
        row = []
        for v in [int(x) for x in os.environ["STAGGER"].split(",")]:
            hip.tune("rconv_stagger", v)
            row.append(f"stagger {v}: {t_us(lambda: hip.gn_silu_conv3x3_unet(x0, sc, sh, wimg, cout, b, temb, nb // 2, r, x1=x1, out=out)):7.1f}")
        hip.tune("rconv_stagger", 0)
        print("    " + "  ".join(row))
    if ABL:
        row = []
        for v in [int(x) for x in ABL.split(",")]:
            hip.tune("rconv_abl", v)
            row.append(f"abl {v}: {t_us(lambda: hip.gn_silu_conv3x3_unet(x0, sc, sh, wimg, cout, b, temb, nb // 2, r, x1=x1, out=out)):7.1f}")
        hip.tune("rconv_abl", 0)
        print("    " + "  ".join(row))
    print(f"{nb} x {H}^2: {c0} + {c1} -> {cout}".ljust(32) + f" {t_gn:8.1f} {t_cv:8.1f} {t_gn + t_cv:8.1f} | {t_st:7.1f} {t_f:8.1f} {t_st + t_f:8.1f}  "
          f"{fl / (t_gn + t_cv) * 1e-6:6.0f} -> {fl / (t_st + t_f) * 1e-6:6.0f}  ({fl / t_cv * 1e-6:6.0f} -> {fl / t_f * 1e-6:6.0f})", flush=True)
