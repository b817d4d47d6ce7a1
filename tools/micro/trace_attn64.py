#!/usr/bin/env python3
"""Where a wave of attn64_kernel (head_dim 40, 64 queries per wave, two waves per SIMD) spends a 64-key tile: per-segment shader-clock sums
from the ATTN_TRACE diagnostic build (`make -C mmgt_amd/csrc trace`), at the in-step shape, beside the static issue-cost model of the
same loop (tools/isa_gaps.py).

    MMGT_LIB=mmgt_amd/libmmgt_hip_trace.so python tools/trace_attn64.py [nk2]      (nk2 = 4096: the bank launch; 0: own keys only)

The stamps fence the scheduler at the segment borders, so read the SHARES; the un-stamped kernel's time is printed beside them
(same process, product library loaded second is not possible: run tools/bench_attn.py for that)."""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mmgt_amd import hip  # noqa: E402

SEGS = ["[0] barrier: wait for the workgroup's previous tile", "[1] vmcnt + LDS writes of the tile + barrier", "[2] issue next tile's global loads",
        "[3] 6 K reads + issue of 12 QK^T MFMAs", "[4] MFMA drain + 34 v_max3 + 2 permlane + decision", "[5] 64 v_exp + 32 cvt_pk + 8 V^T reads + 16 PV MFMAs"]


def main():
    nk2 = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
    dev = torch.device("cuda:0")
    n, c, heads, frames = 4096, 320, 8, 24
    hd = c // heads
    B = 2 * frames
    g = torch.Generator(device=dev).manual_seed(0)
    qk = torch.randn(B * n, 2 * c, device=dev, generator=g).bfloat16()
    vt = torch.randn(B, c, n, device=dev, generator=g).bfloat16()
    kb = torch.randn(2, max(nk2, 8), c, device=dev, generator=g).bfloat16()
    vbt = torch.randn(2, c, max(nk2, 8), device=dev, generator=g).bfloat16()
    o = torch.empty((B * n, c), device=dev, dtype=torch.bfloat16)
    kw = dict(k2=kb, v2=vbt, k2_str=(kb.stride(0), kb.stride(1)), v2_str=(vbt.stride(0), vbt.stride(1)), k2_bdiv=frames, nk2=nk2,
              seg2_first_batch=frames) if nk2 else {}

    def run():
        hip.attention(qk, qk[:, c:], vt, o, batch=B, heads=heads, hd=hd, nq=n, nk=n, scale=hd ** -0.5, q_str=(n * 2 * c, 0, 2 * c),
                      k_str=(n * 2 * c, 0, 2 * c), v_str=(c * n, 0, n), o_str=(n * c, 0, c), v_transposed=True, **kw)
    L = hip.lib()
    try:
        set_trace = L.mmgt_attn64_set_trace
    except AttributeError:
        raise SystemExit("this library has no attention stamps: build `make -C mmgt_amd/csrc trace` and set MMGT_LIB=mmgt_amd/libmmgt_hip_trace.so")
    set_trace.restype, set_trace.argtypes = None, [ctypes.c_void_p]
    for _ in range(20):                                    # warm: the chip settles at the clock it holds under this load
        run()
    nwg = (n // 256) * B * heads
    buf = torch.zeros((nwg * 4, 16), device=dev, dtype=torch.int64)
    set_trace(buf.data_ptr())
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    run()
    e1.record()
    torch.cuda.synchronize()
    set_trace(None)
    t = buf.cpu().double()
    ntl = t[:, 8]
    for label, sel in (("waves with own keys only (64 tiles)", ntl == n // 64), (f"waves with own + bank keys ({(n + nk2) // 64} tiles)", ntl == (n + nk2) // 64)):
        if nk2 == 0 and "bank" in label or sel.sum() == 0:
            continue
        tt = t[sel]
        per = tt[:, :6] / tt[:, 8:9]                       # cycles per tile and segment
        tot = tt[:, 6] / tt[:, 8]
        clk = tt[:, 6] / tt[:, 7] * 100.0                  # MHz: shader cycles per 100-MHz tick
        print(f"\n{label}: {int(sel.sum())} waves; whole loop {tot.median().item():.0f} cycles per 64-query x 64-key wave tile "
              f"(p10 {tot.quantile(0.1).item():.0f}, p90 {tot.quantile(0.9).item():.0f}); in-kernel clock {clk.median().item():.0f} MHz")
        for k, name in enumerate(SEGS):
            col = per[:, k]
            print(f"  {name:58s} median {col.median().item():7.0f}  p10 {col.quantile(0.1).item():7.0f}  p90 {col.quantile(0.9).item():7.0f}"
                  f"  {100 * col.median().item() / tot.median().item():5.1f} %")
        rest = tot - per.sum(1)
        print(f"  {'(stamps themselves, loop control)':58s} median {rest.median().item():7.0f}")
    print(f"\nstamped launch: {e0.elapsed_time(e1) * 1e3:.0f} us (the product kernel is faster: no fences)")
    print("matrix-pipe floor of a wave tile: 28 MFMA x 32 = 896 cycles per wave, two waves per SIMD -> 1792 per SIMD per tile pair")


if __name__ == "__main__":
    main()
