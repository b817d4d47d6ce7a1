#!/usr/bin/env python3
"""The fused FeedForward kernel launched a few times at the level-0 step shape (for rocprofv3 --pmc passes).  python3 tools/ffn_one.py [dbg]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools import abl_lib  # noqa: E402
abl_lib.use()             # the timing ablations live in libmmgt_hip_abl.so only (make abl); the product library refuses their keys
from mmgt_amd import hip  # noqa: E402
from mmgt_amd.packing import pack_ff_fused  # noqa: E402
from mmgt_amd.synthetic import hash_uniform  # noqa: E402

dev = "cuda:0"
M, C, INNER = 48 * 4096, 320, 1280
bf = lambda t: t.to(torch.bfloat16)
x = bf(hash_uniform("ffn.step.x", (M, C), 1.5, dev))
g, b = (1 + 0.2 * hash_uniform("ffn.g", (C,), 1.0, dev)), 0.1 * hash_uniform("ffn.b", (C,), 1.0, dev)
w1 = bf(hash_uniform("ffn.w1", (2 * INNER, C), 1.0, dev) * C ** -0.5)
b1 = 0.1 * hash_uniform("ffn.b1", (2 * INNER,), 1.0, dev)
w2 = bf(hash_uniform("ffn.w2", (C, INNER), 1.0, dev) * INNER ** -0.5)
b2 = 0.1 * hash_uniform("ffn.b2", (C,), 1.0, dev)
img = pack_ff_fused(w1, b1, w2)
out = torch.empty_like(x)
if len(sys.argv) > 1:
    hip.lib().mmgt_tune(b"ffn_dbg", int(sys.argv[1]))
for _ in range(6):
    hip.ff_fused(x, g, b, img, b2, x, INNER, out=out)
torch.cuda.synchronize()
