#!/usr/bin/env python3
"""Per-kernel micro-benchmark at the shapes of the 512x512x24 denoise step (one MI355X).  Prints TFLOP/s and GB/s per
shape so a kernel change can be judged in one short gpurun call.   python tools/bench_kernels.py [gemm conv attn norm]"""
import os
import sys
import math

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mmgt_amd import hip  # noqa: E402
from mmgt_amd.packing import pack_geglu  # noqa: E402

dev = torch.device("cuda:0")
DT = torch.bfloat16
CFGS = [int(c) for c in os.environ.get("CFGS", "0").split(",")]


def timeit(fn, reps=5, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e-3


def rnd(*shape, s=1.0):
    return (torch.rand(shape, device=dev) * 2 - 1).mul_(s).to(DT)


def bench_gemm():
    print("== gemm (M, N, K, epilogue) ==")
    shapes = [(196608, 320, 320, "res"), (196608, 640, 320, ""), (196608, 960, 320, ""), (196608, 2560, 320, "geglu"),
              (196608, 320, 1280, "res"), (49152, 640, 640, "res"), (49152, 1280, 640, ""), (49152, 5120, 640, "geglu"),
              (49152, 640, 2560, "res"), (12288, 1280, 1280, "res"), (12288, 2560, 1280, ""),
              (12288, 10240, 1280, "geglu"), (12288, 1280, 5120, "res"), (3072, 1280, 1280, "res"),
              (3072, 10240, 1280, "geglu"), (8192, 8192, 8192, "")]
    for M, N, K, epi in shapes:
        a, w = rnd(M, K), rnd(N, K, s=1 / math.sqrt(K))
        bias = torch.zeros(N, device=dev)
        nout = N // 2 if epi == "geglu" else N
        res = rnd(M, nout) if epi == "res" else None
        out = torch.empty((M, nout), device=dev, dtype=DT)
        if epi == "geglu":
            w, bias = pack_geglu(w, bias)
        fn = lambda: hip.gemm(a, w, bias, out=out, residual=res, act=hip.ACT_GEGLU if epi == "geglu" else 0)
        byt = 2 * (M * K + N * K + M * nout * (2 if res is not None else 1))
        cells = []
        for cfg in CFGS:
            hip.tune("gemm_cfg", cfg)
            t = timeit(fn)
            cells.append(f"cfg{cfg}: {t*1e6:7.1f}us {2*M*N*K/t/1e12:6.1f}TF {byt/t/1e9:5.0f}GB/s")
        hip.tune("gemm_cfg", 0)
        print(f"M={M:7d} N={N:6d} K={K:5d} {epi:6s} | " + " | ".join(cells))


def bench_conv():
    print("== conv3x3 (NB, H, Cin, Cout) ==")
    for nb, h, cin, cout, st, up in [(48, 64, 320, 320, 1, 0), (48, 64, 960, 320, 1, 0), (48, 32, 640, 640, 1, 0),
                                     (48, 32, 1920, 640, 1, 0), (48, 16, 1280, 1280, 1, 0), (48, 16, 2560, 1280, 1, 0),
                                     (48, 8, 1280, 1280, 1, 0), (48, 8, 2560, 1280, 1, 0), (48, 64, 320, 320, 2, 0),
                                     (48, 32, 640, 640, 1, 1)]:
        x = rnd(nb, h, h, cin)
        w = rnd(cout, 3, 3, cin, s=1 / math.sqrt(9 * cin))
        b = torch.zeros(cout, device=dev)
        fn = lambda: hip.conv3x3(x, w, b, stride=st, upsample=bool(up))
        oh = h * (2 if up else 1) // st
        fl = 2 * nb * oh * oh * cout * 9 * cin
        cells = []
        for cfg in CFGS:
            hip.tune("gemm_cfg", cfg)
            t = timeit(fn)
            cells.append(f"cfg{cfg}: {t*1e6:7.1f}us {fl/t/1e12:6.1f}TF")
        hip.tune("gemm_cfg", 0)
        print(f"nb={nb} h={h:3d} cin={cin:5d} cout={cout:5d} s={st} up={up} | " + " | ".join(cells))


def bench_attn():
    print("== spatial attention (hd, N, bank) ==")
    for hd, n, bank in [(40, 4096, True), (40, 4096, False), (80, 1024, True), (160, 256, True), (160, 64, True)]:
        inner, nb, f = 8 * hd, 48, 24
        qk = rnd(nb * n, 2 * inner)
        vt = rnd(nb, inner, n)
        kb, vbt = rnd(2, n, inner), rnd(2, inner, n)
        o = torch.empty((nb * n, inner), device=dev, dtype=DT)
        kw = dict(k2=kb, v2=vbt, k2_str=(kb.stride(0), kb.stride(1)), v2_str=(vbt.stride(0), vbt.stride(1)), k2_bdiv=f,
                  nk2=n, seg2_first_batch=nb // 2) if bank else {}
        fn = lambda: hip.attention(qk, qk[:, inner:], vt, o, batch=nb, heads=8, hd=hd, nq=n, nk=n, scale=hd ** -0.5,
                                   q_str=(n * 2 * inner, 0, 2 * inner), k_str=(n * 2 * inner, 0, 2 * inner),
                                   v_str=(inner * n, 0, n), o_str=(n * inner, 0, inner), v_transposed=True, **kw)
        t = timeit(fn)
        fl = 4 * 8 * hd * n * n * (nb + (nb // 2 if bank else 0))
        print(f"hd={hd:3d} N={n:5d} bank={bank!s:5s} {t*1e6:9.1f} us  {fl/t/1e12:7.1f} TF/s")
    print("== temporal attention (hd, HW) ==")
    for hd, hw in [(40, 4096), (80, 1024), (160, 256), (160, 64)]:
        c, f, b = 8 * hd, 24, 2
        qkv = rnd(b * f * hw, 3 * c)
        o = torch.empty((b * f * hw, c), device=dev, dtype=DT)
        st = (f * hw * 3 * c, 3 * c, hw * 3 * c)
        fn = lambda: hip.attention(qkv, qkv[:, c:], qkv[:, 2 * c:], o, batch=b * hw, heads=8, hd=hd, nq=f, nk=f,
                                   scale=hd ** -0.5, q_str=st, k_str=st, v_str=st, o_str=(f * hw * c, c, hw * c), bdiv=hw)
        t = timeit(fn)
        print(f"hd={hd:3d} hw={hw:5d} {t*1e6:9.1f} us  {2*(qkv.numel()+o.numel())/t/1e9:7.0f} GB/s")


def bench_norm():
    print("== layernorm (rows, C) / groupnorm (NB, HW, C) ==")
    for rows, c in [(196608, 320), (49152, 640), (12288, 1280), (3072, 1280)]:
        x = rnd(rows, c)
        g, b = torch.ones(c, device=dev), torch.zeros(c, device=dev)
        t = timeit(lambda: hip.layernorm(x, g, b))
        print(f"LN rows={rows:7d} C={c:5d} {t*1e6:9.1f} us  {4*rows*c/t/1e9:7.0f} GB/s")
    for nb, hw, c0, c1 in [(48, 4096, 320, 0), (48, 4096, 640, 320), (48, 1024, 640, 0), (48, 256, 1280, 1280), (48, 64, 1280, 0)]:
        x0 = rnd(nb, hw, c0)
        x1 = rnd(nb, hw, c1) if c1 else None
        g, b = torch.ones(c0 + c1, device=dev), torch.zeros(c0 + c1, device=dev)
        t = timeit(lambda: hip.groupnorm(x0, g, b, 32, 1e-5, silu=True, x1=x1))
        print(f"GN nb={nb} hw={hw:5d} C={c0+c1:5d} {t*1e6:9.1f} us  {2*3*nb*hw*(c0+c1)/t/1e9:7.0f} GB/s (2 reads + 1 write)")


if __name__ == "__main__":
    which = sys.argv[1:] or ["gemm", "conv", "attn", "norm"]
    for w in which:
        globals()["bench_" + w]()
