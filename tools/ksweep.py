import math, os, sys, statistics, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from mmgt_amd import hip
dev = torch.device("cuda:0")
def rnd(*s, sc=1.0): return ((torch.rand(s, device=dev) * 2 - 1) * sc).bfloat16()
def t(fn, reps=5):
    fn(); torch.cuda.synchronize(); best = 1e9
    for _ in range(4):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps): fn()
        e1.record(); torch.cuda.synchronize(); best = min(best, e0.elapsed_time(e1) / reps * 1e3)
    return best
for (M, N) in [(49152, 640), (12288, 1280), (196608, 320)]:
    for res in (False, True):
        row = []
        for K in (64, 128, 320, 640, 1280, 2560):
            a, w = rnd(M, K), rnd(N, K, sc=1 / math.sqrt(K))
            b = torch.rand(N, device=dev) - 0.5
            r = rnd(M, N) if res else None
            o = torch.empty((M, N), device=dev, dtype=torch.bfloat16)
            us = t(lambda: hip.gemm(a, w, b, out=o, residual=r))
            row.append(f"K={K}: {us:6.1f}us")
        print(f"M={M} N={N} res={int(res)} | " + " | ".join(row), flush=True)
