#!/usr/bin/env python3
"""Where the two wave groups of gemm16s_kernel spend a phase: shader-clock stamps of waves 0 and 4 of workgroup 0 from the G16S_TRACE
diagnostic build (`make -C mmgt_amd/csrc trace`), from the workgroup's second tile on.

    MMGT_LIB=mmgt_amd/libmmgt_hip_trace.so python tools/trace_gemm16s.py [M N K [geglu|res]]

Five stamps per phase: phase start | DMA issue (+ epilogue rows) done | s_waitcnt vmcnt passed | first barrier passed | MFMAs issued; the
columns are the differences, the last one the wait at the phase's second barrier.  The stamps cost ~50 cycles each: read the SHARES."""
import ctypes
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mmgt_amd import hip  # noqa: E402


def main():
    a = sys.argv[1:]
    M, N, K = (int(a[0]), int(a[1]), int(a[2])) if len(a) >= 3 else (49152, 1920, 640)
    mode = a[3] if len(a) >= 4 else ""
    dev = torch.device("cuda:0")
    x = (torch.rand(M, K, device=dev) * 2 - 1).bfloat16()
    w = ((torch.rand(N, K, device=dev) * 2 - 1) / math.sqrt(K)).bfloat16()
    b = torch.rand(N, device=dev) - 0.5
    no = N // 2 if mode == "geglu" else N
    r = (torch.rand(M, no, device=dev) * 2 - 1).bfloat16() if mode == "res" else None
    o = torch.empty((M, no), device=dev, dtype=torch.bfloat16)
    L = hip.lib()
    try:
        set_trace = L.mmgt_gemm16s_set_trace
    except AttributeError:
        raise SystemExit("no stamps in this library: `make -C mmgt_amd/csrc trace`, MMGT_LIB=mmgt_amd/libmmgt_hip_trace.so")
    set_trace.restype, set_trace.argtypes = None, [ctypes.c_void_p]
    hip.tune("g16_ver", 2)

    def run():
        hip.gemm(x, w, b, out=o, residual=r, act=1 if mode == "geglu" else 0)
    for _ in range(10):
        run()
    buf = torch.zeros((2, 640), device=dev, dtype=torch.int64)
    set_trace(buf.data_ptr())
    run()
    torch.cuda.synchronize()
    set_trace(None)
    t = buf.cpu().numpy()
    nph = 5 if N % 256 and N % 320 == 0 else 4
    nch = K // 64
    print(f"# gemm16s M={M} N={N} K={K} {mode}: NPH={nph}, {nch} main slots + 1 epilogue slot per tile; cycles (s_memtime, 100 MHz x ... shader clock)")
    for g in (0, 1):
        st = t[g][t[g] > 0]
        n = len(st) // 5
        print(f"## group {g}: {n} phases; first stamp {st[0] - t[0][0]:+d} against group 0's")
        print("  slot ph |  issue   vmcnt  barrier1   mfma  barrier2 | phase")
        tot = [0] * 5
        for p in range(n - 1):
            s = st[5 * p:5 * p + 6]
            d = [int(s[i + 1] - s[i]) for i in range(5)]
            for i in range(5):
                tot[i] += d[i]
            if p < 6 * nph * 2 + nph:
                print(f"  {p // nph:4d} {p % nph:2d} | {d[0]:6d} {d[1]:7d} {d[2]:9d} {d[3]:6d} {d[4]:9d} | {sum(d):6d}")
        sm = sum(tot)
        print("  share   | " + " ".join(f"{100 * v / sm:6.1f}%" for v in tot) + f" | mean phase {sm / (n - 1):.0f}")


if __name__ == "__main__":
    main()
