#!/usr/bin/env python3
"""Debug: per-tile time stamps of gemm16's persistent workgroups (mmgt_gemm16_set_trace): where a tile's time goes."""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mmgt_amd import hip  # noqa: E402
from tools.ab_cfg import gemm_case, time_call  # noqa: E402

lib = ctypes.CDLL(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "mmgt_amd", "libmmgt_hip.so"))
lib.mmgt_gemm16_set_trace.argtypes = [ctypes.c_void_p]
cases = [(16, gemm_case(196608, 1280, 320)), (16, gemm_case(196608, 2560, 320, act=1)), (17, gemm_case(196608, 960, 320, bias=False)),
         (17, gemm_case(196608, 320, 320, res=True)), (16, gemm_case(196608, 1280, 640)), (16, gemm_case(8192, 8192, 1024, bias=False)), (16, gemm_case(196608, 1280, 1024))]
if os.environ.get("SET") == "l1":          # the level-1 / level-2 shapes of the step (K = 640 .. 5120)
    cases = [(16, gemm_case(49152, 5120, 640, act=1)), (16, gemm_case(12288, 10240, 1280, act=1)), (16, gemm_case(12288, 1280, 5120, res=True)),
             (17, gemm_case(49152, 640, 2560, res=True)), (16, gemm_case(49152, 1920, 640, bias=False))]
if os.environ.get("SET") == "short":       # the short reductions with a residual epilogue (tile end = residual in + tile out); STG = start stagger
    cases = [(0, gemm_case(49152, 640, 640, res=True)), (0, gemm_case(49152, 640, 640)), (0, gemm_case(12288, 1280, 1280, res=True)),
             (0, gemm_case(196608, 320, 320, res=True)), (0, gemm_case(49152, 640, 2560, res=True))]
    hip.tune("g16_stagger", int(os.environ.get("STG", "-1")))
for cfg, (name, fn, flops, out) in cases:
    hip.tune("gemm_cfg", cfg)
    t_us = min(time_call(fn) for _ in range(3))
    buf = torch.zeros((65536 + 256 * 8 * 2 * 16,), dtype=torch.int64, device="cuda")
    lib.mmgt_gemm16_set_trace(ctypes.c_void_p(buf.data_ptr()))
    fn()
    torch.cuda.synchronize()
    lib.mmgt_gemm16_set_trace(None)
    ck = buf[65536:].view(256, 8, 2, 16).cpu().double() * 0.01
    b = buf[:65536].view(256, 32, 2, 4).cpu().double() * 0.01      # microseconds
    valid = (b[..., 3] > 0)
    print(f"{name}  cfg {cfg}: {t_us:.1f} us")
    for g in range(2):
        v = valid[:, :, g]
        main = (b[:, :, g, 1] - b[:, :, g, 0])[v]
        bias = (b[:, :, g, 2] - b[:, :, g, 1])[v]
        epi = (b[:, :, g, 3] - b[:, :, g, 2])[v]
        nxt = (b[:, 1:, g, 0] - b[:, :-1, g, 3])[valid[:, 1:, g]]
        ntile = v.sum(1).float().mean().item()
        span = (b[:, :, g, 3].max() - b[:, :, g, 0][v].min()).item()
        print(f"   group {g}: tiles/WG {ntile:.1f}  main {main.mean():6.2f}  bias {bias.mean():5.2f}  epilogue {epi.mean():5.2f}  to-next {nxt.mean():5.2f} us"
              f"   first-tile main {(b[:, 0, g, 1] - b[:, 0, g, 0]).mean():.2f}   span {span:.1f}")
    nch = int((ck[0, 0, 0] > 0).sum())
    if nch > 1:
        d = ck[:, 1:6, 0, 1:nch] - ck[:, 1:6, 0, :nch - 1]          # chunk durations, tiles 1..5, group 0
        print("   chunk durations (us, mean over WGs and tiles 1-5):", [round(x, 2) for x in d.mean((0, 1)).tolist()],
              " last chunk:", round((b[:, 1:6, 0, 1] - ck[:, 1:6, 0, nch - 1]).mean().item(), 2))
    wg0 = b[0, :6, 0] - b[0, 0, 0, 0]
    print("   WG 0 group 0 stamps:", [[round(x, 2) for x in r] for r in wg0.tolist()])
hip.tune("gemm_cfg", 0)
