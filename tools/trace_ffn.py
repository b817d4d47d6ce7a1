#!/usr/bin/env python3
"""Debug: where a workgroup of mmgt_ff_fused spends its cycles (shader-clock stamps of wave 0 at the phase boundaries)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools import abl_lib  # noqa: E402
abl_lib.use()             # the timing ablations live in libmmgt_hip_abl.so only (make abl); the product library refuses their keys
from mmgt_amd import hip  # noqa: E402
from mmgt_amd.packing import pack_ff_fused  # noqa: E402
from mmgt_amd.synthetic import hash_uniform  # noqa: E402


def main():
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
    ver = 4
    for _ in range(3):
        hip.ff_fused(x, g, b, img, b2, x, INNER, out=out)
    nwg = M // 128
    buf = torch.zeros((nwg, 2, 32), device=dev, dtype=torch.int64)
    hip.lib().mmgt_ffn_set_trace(buf.data_ptr())
    hip.ff_fused(x, g, b, img, b2, x, INNER, out=out)
    torch.cuda.synchronize()
    hip.lib().mmgt_ffn_set_trace(None)
    t = buf.cpu()
    namesA = ["x load + tables + LayerNorm", "S(0)", "M(0) + ff1(0) + S(1)"]
    for i in range(1, 6):
        namesA += [f"it{i} GEGLU({i - 1})", f"it{i} M + ff1({i})", f"it{i} -> S({i + 1})"]
    namesA[-1] = "iterations 6 .. nsb - 1"
    namesA += ["last GEGLU + barriers"]
    namesB = ["tables + first DMA issue", "S(0) .. iteration 2"]
    for i in range(2, 8):
        namesB += [f"it{i} ff2({i - 2})", f"it{i} wait W1 + M", f"it{i} DMA issue + wait W2", f"it{i} S({i + 1})"]
    namesB[-1] = "iterations 8 .. nsb, last ff2"
    namesB += ["epilogue"]
    if ver == 4:     # single-role kernel: one stamp stream
        namesA = ["x load + tables + LayerNorm", "wait W1(0), ff1(0), GEGLU(0) 1st half, wait W1(1)"]
        for j in range(6):
            namesA += [f"it{j} A: ff1({j + 1}) || GEGLU({j}) 2nd part (+ hand-over)", f"it{j} B: ff2({j}) || GEGLU({j + 1}) 1st part (+ hand-over)"]
        namesA += ["-"]
        namesA[-1] = "iterations 6 .. nsb - 1"
        namesA += ["epilogue"]
    for role, names in ((0, namesA), (1, namesB))[:1 if ver == 4 else 2]:
        tr = t[:, role]
        n = int((tr[0] != 0).sum())
        d = (tr[:, 1:n] - tr[:, :n - 1]).float()
        tot = (tr[:, n - 1] - tr[:, 0]).float()
        print(f"role {'AB'[role]}: {n} stamps; whole wave: median {tot.median().item():.0f} ticks (min {tot.min().item():.0f}, max {tot.max().item():.0f})")
        for i in range(n - 1):
            col = d[:, i]
            print(f"  {names[i] if i < len(names) else str(i):45s} median {col.median().item():8.0f}  p10 {col.quantile(0.1).item():8.0f}  p90 {col.quantile(0.9).item():8.0f}")


if __name__ == "__main__":
    main()
