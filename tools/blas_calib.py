#!/usr/bin/env python3
"""Calibration only (never the product path): what the vendor GEMM (torch.matmul -> hipBLASLt/rocBLAS) reaches on the
step's GEMM shapes on this device, to judge how far the hand-written kernel is from the practical ceiling."""
import torch

dev = torch.device("cuda:0")
shapes = [(196608, 320, 320), (196608, 960, 320), (196608, 2560, 320), (196608, 320, 1280), (49152, 640, 640),
          (49152, 5120, 640), (49152, 640, 2560), (12288, 1280, 1280), (12288, 10240, 1280), (12288, 1280, 5120),
          (3072, 10240, 1280), (8192, 8192, 8192)]
for M, N, K in shapes:
    a = torch.randn(M, K, device=dev, dtype=torch.bfloat16)
    w = torch.randn(N, K, device=dev, dtype=torch.bfloat16)
    o = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    for _ in range(3):
        torch.matmul(a, w.t(), out=o)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        torch.matmul(a, w.t(), out=o)
    e1.record()
    torch.cuda.synchronize()
    t = e0.elapsed_time(e1) / 5 * 1e-3
    print(f"M={M:7d} N={N:6d} K={K:5d}  {t*1e6:8.1f} us  {2*M*N*K/t/1e12:7.1f} TF/s")
