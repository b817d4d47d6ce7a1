#!/usr/bin/env python3
"""GroupNorm + SiLU + conv3x3 128 -> 128 at the VAE's 512 x 512 level: the fused launch (csrc/gnconv.hip) against hip.groupnorm -> hip.conv3x3.
    python tools/bench_gnconv.py [frames]"""
import math
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tools import abl_lib  # noqa: E402
abl_lib.use()             # the timing ablations live in libmmgt_hip_abl.so only (make abl); the product library refuses their keys
from mmgt_amd import hip  # noqa: E402
from mmgt_amd.packing import pack_conv3x3, pack_gnconv  # noqa: E402

dev = torch.device("cuda:0")


def timeit(fn, reps=5):
    fn()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / reps * 1e3)
    return best


def main():
    nb = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    for H in (512, 256):
        x = (torch.randn((nb, H, H, 128), device=dev) * 1.5 + 0.2).bfloat16()
        w = torch.randn((128, 128, 3, 3), device=dev) / math.sqrt(9 * 128)
        gamma, beta, b = torch.rand(128, device=dev) + 0.5, torch.rand(128, device=dev) - 0.5, torch.rand(128, device=dev) - 0.5
        r = torch.randn((nb, H, H, 128), device=dev).bfloat16()
        wimg, wp = pack_gnconv(w), pack_conv3x3(w.cpu()).to(dev).bfloat16()
        out = torch.empty_like(x)
        y = torch.empty_like(x)
        scale, shift = hip.groupnorm_affine(x.view(nb, H * H, 128), gamma, beta, 32, 1e-6)
        fl = 2.0 * nb * H * H * 128 * 9 * 128
        t_gn = timeit(lambda: hip.groupnorm(x.view(nb, H * H, 128), gamma, beta, 32, 1e-6, silu=True, out=y.view(nb, H * H, 128)))
        t_cv = timeit(lambda: hip.conv3x3(y, wp, b, residual=r, out=out))
        t_st = timeit(lambda: hip.groupnorm_affine(x.view(nb, H * H, 128), gamma, beta, 32, 1e-6))
        t_fu = timeit(lambda: hip.gn_silu_conv3x3_tables(x, scale, shift, wimg, 128, b, r, out=out))
        t_fn = timeit(lambda: hip.gn_silu_conv3x3_tables(x, scale, shift, wimg, 128, b, None, out=out))
        print(f"{nb} x {H} x {H} x 128: groupnorm+silu {t_gn:7.1f} us + conv3x3(+res) {t_cv:7.1f} us ({fl / t_cv / 1e6:5.0f} TF/s) = {t_gn + t_cv:7.1f} us | "
              f"statistics {t_st:6.1f} us + fused(+res) {t_fu:7.1f} us ({fl / t_fu / 1e6:5.0f} TF/s) = {t_st + t_fu:7.1f} us | fused without residual {t_fn:7.1f} us", flush=True)


if __name__ == "__main__":
    main()


def ablations():
    nb, H = 8, 512
    x = (torch.randn((nb, H, H, 128), device=dev) * 1.5 + 0.2).bfloat16()
    w = torch.randn((128, 128, 3, 3), device=dev) / math.sqrt(9 * 128)
    scale, shift = torch.rand((nb, 128), device=dev) + 0.5, torch.rand((nb, 128), device=dev) - 0.5
    wimg = pack_gnconv(w)
    out = torch.empty_like(x)
    base = None
    print("timing ablations (mmgt_tune gnconv_abl; results are garbage):")
    for abl, what in [(0, "the kernel"), (1, "no MFMAs"), (2, "no weight DMA / wait"), (4, "no halo loads / normalisation / LDS writes"), (8, "no hand-over wait"),
                      (16, "no epilogue stores"), (32, "no hand-over barrier"), (6, "no weight DMA, no halo"), (22, "no weight DMA, no halo, no epilogue"),
                      (64, "halo loads, but no normalisation / LDS writes"), (128, "normalisation / LDS writes, but no halo loads")]:
        hip.tune("gnconv_abl", abl)
        t = timeit(lambda: hip.gn_silu_conv3x3_tables(x, scale, shift, wimg, 128, None, None, out=out))
        base = base or t
        print(f"  abl {abl:3d} {what:45s} {t:7.1f} us  ({t - base:+7.1f})", flush=True)
    hip.tune("gnconv_abl", 0)


if __name__ == "__main__" and os.environ.get("ABL"):
    ablations()
