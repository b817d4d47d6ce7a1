import sys, os, math, torch
sys.path.insert(0, "/root/repo" if os.path.isdir("/root/repo/mmgt_amd") else os.getcwd())
from mmgt_amd import hip
from tools.bench_kernels import timeit, rnd, dev, DT
for cfg in (1,3):
    hip.tune("gemm_cfg", cfg)
    for N in (320, 640):
        for K in (64, 128, 320, 640, 1280, 2560):
            M=196608
            a, w = rnd(M, K), rnd(N, K, s=1/math.sqrt(K))
            out = torch.empty((M, N), device=dev, dtype=DT)
            t = timeit(lambda: hip.gemm(a, w, None, out=out))
            print(f"cfg{cfg} N={N} K={K:5d} {t*1e6:8.1f} us  {2*M*N*K/t/1e12:6.1f} TF  rd {2*M*K/1e6:6.0f}MB wr {2*M*N/1e6:6.0f}MB  {(2*M*K+2*M*N)/t/1e9:6.0f} GB/s")
