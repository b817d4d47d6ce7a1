import os, sys, math, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mmgt_amd import hip
from tools.bench_kernels import timeit, rnd, dev, DT
for (M,N,K,res) in [(196608,320,320,True),(196608,320,320,False),(196608,640,320,False),(196608,320,64,False)]:
    a, w = rnd(M, K), rnd(N, K, s=1 / math.sqrt(K))
    r = rnd(M, N) if res else None
    out = torch.empty((M, N), device=dev, dtype=DT)
    for cfg in (3, 5, 1):
        hip.tune("gemm_cfg", cfg)
        ts=[]
        for dbg in (0, 1):
            hip.tune("gemm_dbg", dbg)
            ts.append(timeit(lambda: hip.gemm(a, w, None, out=out, residual=r))*1e6)
        print(f"M={M} N={N} K={K} res={res} cfg{cfg}: normal {ts[0]:7.1f} us   no-store {ts[1]:7.1f} us")
hip.tune("gemm_dbg", 0)
