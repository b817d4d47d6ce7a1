#!/usr/bin/env python3
"""Debug: 100-MHz stamps of wave 0 of every persistent workgroup of csrc/rconv.hip (mmgt_rconv_set_trace of the -DMMGT_ABLATE library): k-step starts and the epilogue."""
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tools import abl_lib  # noqa: E402
path = abl_lib.use()             # the stamps are compiled into the ablation library only
from mmgt_amd import hip  # noqa: E402
from mmgt_amd.packing import pack_rconv  # noqa: E402

lib = ctypes.CDLL(path)
lib.mmgt_rconv_set_trace.argtypes = [ctypes.c_void_p]
dev = torch.device("cuda:0")
nb, H, c0, c1, cout = [int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "48x64x320x0x320").split("x")]
cin = c0 + c1
x0 = torch.randn((nb, H, H, c0), device=dev).bfloat16()
x1 = torch.randn((nb, H, H, c1), device=dev).bfloat16() if c1 else None
w = torch.randn((cout, cin, 3, 3), device=dev) / (9 * cin) ** 0.5
tab = torch.rand((2, nb, cin), device=dev)
r = torch.randn((nb, H, H, cout), device=dev).bfloat16()
wimg = pack_rconv(w)
out = torch.empty((nb, H, H, cout), device=dev, dtype=torch.bfloat16)
if os.environ.get("CB"):
    hip.tune("rconv_cb", int(os.environ["CB"]))
run = lambda: hip.gn_silu_conv3x3_unet(x0, tab[0], tab[1], wimg, cout, None, None, 0, r, x1=x1, out=out)
for _ in range(3):
    run()
buf = torch.zeros((256, 512), dtype=torch.int64, device=dev)
lib.mmgt_rconv_set_trace(ctypes.c_void_p(buf.data_ptr()))
run()
torch.cuda.synchronize()
lib.mmgt_rconv_set_trace(None)
t = buf.cpu().double() * 0.01                       # microseconds
nph = cin // 64
per_unit = nph * 18 + 2
t0 = t[:, 0].min()
print(f"{nb} x {H}^2: {c0} + {c1} -> {cout}; stamps per unit {per_unit}; first stamp spread over workgroups {t[:, 0].max() - t0:.1f} us")
for u in range(4):
    b = u * per_unit
    if (t[:, b + per_unit - 1] <= 0).all():
        break
    v = t[:, b + per_unit - 1] > 0
    ks = t[v, b + 1:b + nph * 18] - t[v, b:b + nph * 18 - 1]            # k-step durations
    epi = t[v, b + per_unit - 1] - t[v, b + per_unit - 2]
    last = t[v, b + per_unit - 2] - t[v, b + nph * 18 - 1]
    print(f" unit {u}: {int(v.sum())} workgroups; start {float((t[v, b] - t0).mean()):7.1f} +- {float((t[v, b]).std()):4.1f} us; k-step mean {float(ks.mean()):.3f} us; "
          f"last k-step {float(last.mean()):.2f}; epilogue {float(epi.mean()):.2f} (max {float(epi.max()):.2f})")
    m = ks.mean(0)
    print("   k-step durations by index (first phase):", [round(float(x), 2) for x in m[:18]])
    print("   k-step durations by index (second phase):", [round(float(x), 2) for x in m[18:36]])
print(" end spread:", float(t.max() - t0), "us")
if os.environ.get("FINE"):
    # four stamps per k-step: start | in front of the counted wait | in front of the barrier | behind the barrier
    buf.zero_()
    lib.mmgt_rconv_set_trace(ctypes.c_void_p(buf.data_ptr() | 1))
    run()
    torch.cuda.synchronize()
    lib.mmgt_rconv_set_trace(None)
    t = buf.cpu().double() * 0.01
    nk = min(126, nph * 18)
    q = t[:, :4 * nk].view(256, nk, 4)
    nxt = t[:, 4:4 * nk + 4:4]
    head, wait, barr, tail = q[:, :, 1] - q[:, :, 0], q[:, :, 2] - q[:, :, 1], q[:, :, 3] - q[:, :, 2], nxt - q[:, :, 3]
    print(" fine stamps of the first", nk, "k-steps of unit 0, mean over workgroups, by k-step index inside the phase (phases 1.. averaged):")
    for name, v in (("MFMA groups in front of the hand-over", head), ("counted wait", wait), ("barrier", barr), ("hand-over issue + rest of the k-step", tail)):
        m = v.mean(0)[18:18 * (nk // 18)].view(-1, 18).mean(0)
        print(f"   {name:40s}", [round(float(x), 2) for x in m], " sum", round(float(m.sum()), 2))

