#!/usr/bin/env python3
"""Debug: where a workgroup of mmgt_ff_fused spends its cycles (shader-clock stamps of wave 0 at the phase boundaries)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
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
    for _ in range(3):
        hip.ff_fused(x, g, b, img, b2, x, INNER, out=out)
    nwg = M // 128
    buf = torch.zeros((nwg, 64), device=dev, dtype=torch.int64)
    hip.lib().mmgt_ffn_set_trace(buf.data_ptr())
    hip.ff_fused(x, g, b, img, b2, x, INNER, out=out)
    torch.cuda.synchronize()
    hip.lib().mmgt_ffn_set_trace(None)
    t = buf.cpu()
    n = int((t[0] != 0).sum())
    d = (t[:, 1:n] - t[:, :n - 1]).float()
    names = ["prologue (x load, LN)", "wait W1(0) + barrier .. iteration 0 start"]
    for j in range(8):
        names += [f"it{j} phase A (ff1 || GEGLU)", f"it{j} wait+barrier", f"it{j} phase B (ff2)", f"it{j} -> it{j + 1} wait+barrier"]
    names[-1] = "iterations 8 .. 39"
    names += ["epilogue"]
    tot = (t[:, n - 1] - t[:, 0]).float()
    print(f"{nwg} workgroups, {n} stamps; whole workgroup: median {tot.median().item():.0f} cycles (min {tot.min().item():.0f}, max {tot.max().item():.0f})")
    for i in range(n - 1):
        col = d[:, i]
        print(f"  {names[i] if i < len(names) else i:45s} median {col.median().item():8.0f}  p10 {col.quantile(0.1).item():8.0f}  p90 {col.quantile(0.9).item():8.0f}")
    first = t[:, 0].float()
    print(f"start spread: {(first.max() - first.min()).item():.0f} cycles (s_memtime is per-XCD-consistent only)")


if __name__ == "__main__":
    main()
