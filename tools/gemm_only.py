import os, sys, math, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mmgt_amd import hip
from tools.bench_kernels import timeit, rnd, dev, DT
for cfg in (1, 6):
    hip.tune("gemm_cfg", cfg)
    M = N = K = 8192
    a, w = rnd(M, K), rnd(N, K, s=1 / math.sqrt(K))
    out = torch.empty((M, N), device=dev, dtype=DT)
    t = timeit(lambda: hip.gemm(a, w, None, out=out))
    print(cfg, t * 1e6, 2 * M * N * K / t / 1e12)
