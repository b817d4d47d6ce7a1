#!/usr/bin/env python3
"""Debug: where a workgroup of mmgt_rowgemm320 spends its cycles (shader-clock stamps of wave 0: start | rows loaded + LayerNorm |
first tile landed | after every tile pair)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mmgt_amd import hip  # noqa: E402
from mmgt_amd.packing import pack_rowgemm  # noqa: E402
from mmgt_amd.synthetic import hash_uniform  # noqa: E402


def main():
    dev = "cuda:0"
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 960
    ln = (sys.argv[2] != "0") if len(sys.argv) > 2 else True
    M, C = 48 * 4096, 320
    bf = lambda t: t.to(torch.bfloat16)
    x = bf(hash_uniform("rg.step.x", (M, C), 1.5, dev))
    g, b = (1 + 0.2 * hash_uniform("rg.g", (C,), 1.0, dev)), 0.1 * hash_uniform("rg.b", (C,), 1.0, dev)
    img = pack_rowgemm(bf(hash_uniform("rg.w", (N, C), 1.0, dev) * C ** -0.5))
    out = torch.empty((M, N), device=dev, dtype=torch.bfloat16)
    kw = dict(ln_gamma=g, ln_beta=b) if ln else {}
    hip.lib().mmgt_tune(b"rowgemm_dbg", 5)
    for _ in range(3):
        hip.rowgemm320(x, img, N, out=out, **kw)
    nwg = M // 128
    buf = torch.zeros((nwg, 32), device=dev, dtype=torch.int64)
    hip.lib().mmgt_rowgemm_set_trace(buf.data_ptr())
    hip.rowgemm320(x, img, N, out=out, **kw)
    torch.cuda.synchronize()
    hip.lib().mmgt_rowgemm_set_trace(None)
    t = buf.cpu()
    n = int((t[0] != 0).sum())
    d = (t[:, 1:n] - t[:, :n - 1]).float()
    tot = (t[:, n - 1] - t[:, 0]).float()
    print(f"N={N} ln={ln}: {n} stamps; whole wave: median {tot.median().item():.0f} ticks (min {tot.min().item():.0f}, max {tot.max().item():.0f})")
    names = ["start -> all requests issued", "requests landed (vmcnt 0)", "tables to LDS + barrier", "LayerNorm", "tile 0 wait + barrier"] + [f"tiles {2 * i}, {2 * i + 1}" for i in range(n)]
    for i in range(n - 1):
        col = d[:, i]
        print(f"  {names[i]:34s} median {col.median().item():8.0f}  p10 {col.quantile(0.1).item():8.0f}  p90 {col.quantile(0.9).item():8.0f}")
    start = t[:, 0].float()
    print(f"  workgroup start times: span {(start.max() - start.min()).item():.0f} ticks; wave-0 end - global start: {(t[:, n - 1].max() - t[:, 0].min()).item():.0f}")


if __name__ == "__main__":
    main()
