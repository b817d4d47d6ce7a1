#!/usr/bin/env python3
"""A/B of GEMM / conv main-loop variants in ONE process on ONE device (guide rule 24): loads libmmgt_hip.so and the
libmmgt_hip_v*.so builds (make -C mmgt_amd/csrc ab) side by side and interleaves rounds."""
import ctypes
import glob
import math
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mmgt_amd import hip  # noqa: E402

dev = torch.device("cuda:0")
libs = {}
paths = [os.path.join(ROOT, "mmgt_amd", "libmmgt_hip.so")] + sorted(glob.glob(os.path.join(ROOT, "mmgt_amd", "csrc", "build", "libmmgt_hip_v*.so")))
if len(paths) < 2:
    sys.exit("ab_gemm: no variant library under mmgt_amd/csrc/build/ -- run `make -C mmgt_amd/csrc ab` first (nothing to compare)")
for path in paths:
    L = ctypes.CDLL(path)
    for name in ("mmgt_gemm", "mmgt_conv3x3_nhwc", "mmgt_tune"):
        fn = getattr(L, name)
        fn.restype, fn.argtypes = hip._SIGS[name]
    libs[os.path.basename(path).replace("libmmgt_hip", "").replace(".so", "") or "cur"] = L


def rnd(*shape, s=1.0):
    return (torch.rand(shape, device=dev) * 2 - 1).mul_(s).bfloat16()


def time_call(fn, reps=4):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


def run(cases, cfgs, rounds=5):
    st = torch.cuda.current_stream().cuda_stream
    for desc, make in cases:
        calls, flops = make()
        best = {}
        for _ in range(rounds):
            for lname, L in libs.items():
                for cfg in cfgs:
                    L.mmgt_tune(b"gemm_cfg", cfg)
                    t = time_call(lambda: calls(L, st))
                    k = (lname, cfg)
                    best[k] = min(best.get(k, 1e30), t)
        cells = [f"{ln}/c{c}:{t:7.1f}us {flops / t / 1e6:6.0f}TF" for (ln, c), t in sorted(best.items())]
        print(f"{desc:38s} | " + " | ".join(cells))


def gemm_case(M, N, K):
    def make():
        a, w = rnd(M, K), rnd(N, K, s=1 / math.sqrt(K))
        o = torch.zeros((M, N), device=dev, dtype=torch.bfloat16)
        o._is_out = True
        call = lambda L, st: L.mmgt_gemm(a.data_ptr(), K, w.data_ptr(), None, None, 0, None, 1.0, None, 0, o.data_ptr(), N,
                                         M, N, K, 0, 1, 0, 0, 0, 0, 1, st)
        return call, 2 * M * N * K
    return f"gemm M={M} N={N} K={K}", make


def gemm_epi_case(M, N, K, act=0):
    """bias + residual (attention to_out / ff2) or bias + GEGLU (ff1): the epilogues the UNet actually runs."""
    def make():
        a, w = rnd(M, K), rnd(N, K, s=1 / math.sqrt(K))
        no = N // 2 if act == 1 else N
        b = torch.rand(N, device=dev) - 0.5
        r = None if act == 1 else rnd(M, no)
        o = torch.zeros((M, no), device=dev, dtype=torch.bfloat16)
        o._is_out = True
        call = lambda L, st: L.mmgt_gemm(a.data_ptr(), K, w.data_ptr(), b.data_ptr(), None, 0, None, 1.0,
                                         r.data_ptr() if r is not None else None, no, o.data_ptr(), no, M, N, K, act, 1, 0,
                                         0, 0, 0, 1, st)
        return call, 2 * M * N * K
    return f"gemm+epi M={M} N={N} K={K} act={act}", make


def conv_case(nb, h, cin, cout):
    def make():
        x, w = rnd(nb, h, h, cin), rnd(cout, 3, 3, cin, s=1 / math.sqrt(9 * cin))
        o = torch.zeros((nb, h, h, cout), device=dev, dtype=torch.bfloat16)
        o._is_out = True
        call = lambda L, st: L.mmgt_conv3x3_nhwc(x.data_ptr(), cin, None, 0, nb, h, h, 1, 0, w.data_ptr(), None, None, 0, None,
                                                 o.data_ptr(), cout, 0, 1, st)
        return call, 2 * nb * h * h * cout * 9 * cin
    return f"conv nb={nb} h={h} cin={cin} cout={cout}", make


if __name__ == "__main__":
    cfgs = [int(c) for c in os.environ.get("CFGS", "1,6").split(",")]
    print("libs:", list(libs))
    # every variant must reproduce the current library (NODMA timing diagnostics, variants with bit 4, are skipped)
    st0 = torch.cuda.current_stream().cuda_stream
    for desc, make in [gemm_epi_case(1000, 320, 320), gemm_epi_case(2304, 640, 1280), gemm_epi_case(4096, 2560, 320, 1),
                       conv_case(3, 16, 192, 320), gemm_case(777, 1280, 2560)]:
        for cfg in cfgs:
            outs = {}
            for lname, L in libs.items():
                if lname.startswith("_v") and int(lname[2:]) & 16:
                    continue
                torch.manual_seed(1)
                calls, _ = make()
                L.mmgt_tune(b"gemm_cfg", cfg)
                calls(L, st0)
                torch.cuda.synchronize()
                outs[lname] = calls.__closure__
            ref = None
            for lname, cl in outs.items():
                o = [c.cell_contents for c in cl if torch.is_tensor(c.cell_contents)]
                o = [t for t in o if getattr(t, "_is_out", False)][0].float()
                if ref is None:
                    ref = o
                else:
                    d = (o - ref).abs().max().item()
                    print(f"check {desc} cfg{cfg} {lname}: max|d| vs cur = {d:.3e}")
                    assert d == 0.0, "variant differs from the current library"
    if os.environ.get("EPI"):
        run([gemm_epi_case(196608, 320, 320), gemm_epi_case(196608, 2560, 320, 1), gemm_epi_case(196608, 320, 1280),
             gemm_epi_case(49152, 640, 640), gemm_epi_case(49152, 5120, 640, 1), gemm_case(196608, 960, 320),
             gemm_epi_case(12288, 1280, 1280)], cfgs)
        sys.exit(0)
    if os.environ.get("BIG"):
        run([gemm_case(8192, 8192, 8192), gemm_case(12288, 1280, 5120), gemm_case(49152, 5120, 640)], cfgs)
        sys.exit(0)
    run([gemm_case(8192, 8192, 8192), gemm_case(12288, 1280, 5120), gemm_case(49152, 640, 2560),
         gemm_case(196608, 320, 1280), gemm_case(196608, 640, 320), gemm_case(49152, 1280, 640),
         conv_case(48, 64, 320, 320), conv_case(48, 32, 640, 640), conv_case(48, 32, 1920, 640),
         conv_case(48, 16, 1280, 1280), conv_case(48, 8, 1280, 1280)], cfgs)
