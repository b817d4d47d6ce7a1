#!/usr/bin/env python3
"""One GEMM shape launched a few times (for rocprofv3 --pmc passes).  python3 tools/gemm_one.py M N K [cfg] [res]"""
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mmgt_amd import hip  # noqa: E402

if os.environ.get("MMGT_ALT_LIB"):   # experiment builds
    hip.LIB_PATH = os.environ["MMGT_ALT_LIB"]

M, N, K = (int(v) for v in sys.argv[1:4])
cfg = int(sys.argv[4]) if len(sys.argv) > 4 else 0
with_res = len(sys.argv) > 5 and sys.argv[5] == "res"
dev = torch.device("cuda:0")
a = (torch.rand(M, K, device=dev) - 0.5).bfloat16()
w = ((torch.rand(N, K, device=dev) - 0.5) / math.sqrt(K)).bfloat16()
b = torch.rand(N, device=dev) - 0.5
res = (torch.rand(M, N, device=dev) - 0.5).bfloat16() if with_res else None
out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
hip.tune("gemm_cfg", cfg)
for _ in range(6):
    hip.gemm(a, w, b, out=out, residual=res)
torch.cuda.synchronize()
