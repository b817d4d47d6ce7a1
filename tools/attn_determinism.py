#!/usr/bin/env python3
"""Run-to-run determinism of the spatial attention kernels (a data race shows up as a result that changes between identical
launches): every kernel variant is launched REPS times on the same inputs and each output compared bitwise with the first."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mmgt_amd import hip  # noqa: E402
from tools.bench_attn import case  # noqa: E402

REPS = int(os.environ.get("REPS", "40"))
for n, c, nk2 in [(4096, 320, 4096), (4096, 320, 0), (1024, 640, 1024), (256, 1280, 256)]:
    name, run, fl, o = case(n, c, nk2)
    for a64 in (0, 1):
        hip.tune("attn64", a64)
        run()
        torch.cuda.synchronize()
        first = o.clone()
        bad = 0
        for _ in range(REPS):
            o.zero_()
            run()
            torch.cuda.synchronize()
            bad += int(not torch.equal(o, first))
        print(f"{name:32s} attn64={a64}: {bad} of {REPS} launches differ from the first", flush=True)
hip.tune("attn64", 1)
