"""One shape of the spatial attention kernel, a few launches (PMC target).  python tools/attn_one.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mmgt_amd import hip
dev = torch.device("cuda:0")
hd, n, nb, f = 40, 4096, 48, 24
inner = 8 * hd
rnd = lambda *s: (torch.rand(s, device=dev) * 2 - 1).bfloat16()
qk, vt, kb, vbt = rnd(nb * n, 2 * inner), rnd(nb, inner, n), rnd(2, n, inner), rnd(2, inner, n)
o = torch.empty((nb * n, inner), device=dev, dtype=torch.bfloat16)
for _ in range(3):
    hip.attention(qk, qk[:, inner:], vt, o, batch=nb, heads=8, hd=hd, nq=n, nk=n, scale=hd ** -0.5,
                  q_str=(n * 2 * inner, 0, 2 * inner), k_str=(n * 2 * inner, 0, 2 * inner), v_str=(inner * n, 0, n),
                  o_str=(n * inner, 0, inner), v_transposed=True, k2=kb, v2=vbt, k2_str=(kb.stride(0), kb.stride(1)),
                  v2_str=(vbt.stride(0), vbt.stride(1)), k2_bdiv=f, nk2=n, seg2_first_batch=nb // 2)
torch.cuda.synchronize()
