#!/usr/bin/env python3
"""One in-step operator shape launched a few times (the target of the rocprofv3 PMC passes of tools/pmc_ops.sh).
    python3 tools/op_one.py gemm M N K [res]      | geglu M N K | conv NB H CIN COUT | ffn | ffn_po | rowgemm N [vt] | attn"""
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mmgt_amd import hip  # noqa: E402
from mmgt_amd.packing import pack_conv3x3, pack_ff_fused, pack_ff_proj_out, pack_geglu, pack_rowgemm  # noqa: E402

dev = torch.device("cuda:0")
rnd = lambda *s: (torch.rand(s, device=dev) * 2 - 1)
kind = sys.argv[1]
REPS = 6
if kind in ("gemm", "geglu"):
    M, N, K = (int(v) for v in sys.argv[2:5])
    a = rnd(M, K).bfloat16()
    w = (rnd(N, K) / math.sqrt(K)).bfloat16()
    b = rnd(N)
    if kind == "geglu":
        w, b = pack_geglu(w, b)
        w, b = w.contiguous(), b.contiguous()
        fn = lambda: hip.gemm(a, w, b, act=hip.ACT_GEGLU)
    else:
        res = rnd(M, N).bfloat16() if len(sys.argv) > 5 and sys.argv[5] == "res" else None
        out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
        fn = lambda: hip.gemm(a, w, b, out=out, residual=res)
elif kind == "conv":
    nb, h, cin, cout = (int(v) for v in sys.argv[2:6])
    x = rnd(nb, h, h, cin).bfloat16()
    w = pack_conv3x3((rnd(cout, cin, 3, 3) / math.sqrt(9 * cin)).bfloat16())
    b = rnd(cout)
    fn = lambda: hip.conv3x3(x, w, b)
elif kind == "rconv":
    from mmgt_amd.packing import pack_rconv
    nb, h, cin, cout = (int(v) for v in sys.argv[2:6])
    res = len(sys.argv) > 6 and sys.argv[6] == "res"
    x = rnd(nb, h, h, cin).bfloat16()
    wimg = pack_rconv(rnd(cout, cin, 3, 3) / math.sqrt(9 * cin))
    b, temb = rnd(cout), rnd(2, cout)
    tab = torch.rand((2, nb, cin), device=dev)
    r = rnd(nb, h, h, cout).bfloat16() if res else None
    out = torch.empty((nb, h, h, cout), device=dev, dtype=torch.bfloat16)
    fn = lambda: hip.gn_silu_conv3x3_unet(x, tab[0], tab[1], wimg, cout, b, temb, nb // 2, r, out=out)
elif kind == "ffn":
    M, C, INNER = 48 * 4096, 320, 1280
    x = rnd(M, C).bfloat16()
    g, b = 1 + 0.2 * rnd(C), 0.1 * rnd(C)
    img = pack_ff_fused((rnd(2 * INNER, C) * C ** -0.5).bfloat16(), 0.1 * rnd(2 * INNER), (rnd(C, INNER) * INNER ** -0.5).bfloat16())
    b2 = 0.1 * rnd(C)
    out = torch.empty_like(x)
    fn = lambda: hip.ff_fused(x, g, b, img, b2, x, INNER, out=out)
elif kind == "ffn_po":
    M, C, INNER = 48 * 4096, 320, 1280
    x = rnd(M, C).bfloat16()
    g, b = 1 + 0.2 * rnd(C), 0.1 * rnd(C)
    img = pack_ff_fused((rnd(2 * INNER, C) * C ** -0.5).bfloat16(), 0.1 * rnd(2 * INNER), (rnd(C, INNER) * INNER ** -0.5).bfloat16())
    b2, bpo = 0.1 * rnd(C), 0.1 * rnd(C)
    po = pack_ff_proj_out((rnd(C, C) * C ** -0.5).bfloat16())
    res2 = rnd(M, C).bfloat16()
    out = torch.empty_like(x)
    fn = lambda: hip.ff_fused_po(x, g, b, img, b2, x, INNER, po, bpo, res2, out=out)
elif kind == "rowgemm":
    M, C, N = 48 * 4096, 320, int(sys.argv[2])
    vt = len(sys.argv) > 3 and sys.argv[3] == "vt"
    x = rnd(M, C).bfloat16()
    g, b = 1 + 0.2 * rnd(C), 0.1 * rnd(C)
    img = pack_rowgemm((rnd(N, C) * C ** -0.5).bfloat16())
    n1 = N - C if vt else N
    out = torch.empty((M, n1), device=dev, dtype=torch.bfloat16)
    out_t = torch.empty((48, C, 4096), device=dev, dtype=torch.bfloat16) if vt else None
    fn = lambda: hip.rowgemm320(x, img, N, ln_gamma=g, ln_beta=b, n1=n1, n_tok=4096 if vt else 0, out=out, out_t=out_t)
elif kind == "attn":
    hd, n, nb, f = 40, 4096, 48, 24
    inner = 8 * hd
    qk, vt, kb, vbt = rnd(nb * n, 2 * inner).bfloat16(), rnd(nb, inner, n).bfloat16(), rnd(2, n, inner).bfloat16(), rnd(2, inner, n).bfloat16()
    o = torch.empty((nb * n, inner), device=dev, dtype=torch.bfloat16)
    fn = lambda: hip.attention(qk, qk[:, inner:], vt, o, batch=nb, heads=8, hd=hd, nq=n, nk=n, scale=hd ** -0.5,
                               q_str=(n * 2 * inner, 0, 2 * inner), k_str=(n * 2 * inner, 0, 2 * inner), v_str=(inner * n, 0, n),
                               o_str=(n * inner, 0, inner), v_transposed=True, k2=kb, v2=vbt, k2_str=(kb.stride(0), kb.stride(1)),
                               v2_str=(vbt.stride(0), vbt.stride(1)), k2_bdiv=f, nk2=n, seg2_first_batch=nb // 2)
else:
    raise SystemExit(__doc__)
for _ in range(REPS):
    fn()
torch.cuda.synchronize()
