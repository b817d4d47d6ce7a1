import os, sys, math, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mmgt_amd import hip
from tools.bench_kernels import timeit, rnd, dev, DT
M, N, K = 196608, 320, 320
a, w = rnd(M, K), rnd(N, K, s=1 / math.sqrt(K))
r = rnd(M, N)
out = torch.empty((M, N), device=dev, dtype=DT)
for cfg in (3, 1):
    hip.tune("gemm_cfg", cfg)
    t = timeit(lambda: hip.gemm(a, w, None, out=out, residual=r))
    print(cfg, t * 1e6)
