"""mmgt_temporal_leg320 (csrc/tleg.hip): one launch per temporal-attention leg of a level-0 motion module, through the C ABI.

Reference semantics: src/models/motion_module.py:236-259 (TemporalTransformerBlock: hidden = attention_block(norm(hidden)) + hidden), :351-388
(VersatileAttention: (b f) d c -> (b d) f c, pos_encoder, attention over the f frames of one pixel, to_out) and :262-273 (PositionalEncoding).

  * random operands against fp64 of the same bf16-rounded operands with the kernel's rounding points reproduced (LayerNorm + pe output, q | k | v,
    probabilities, attention output): gate = one output bf16 ulp + accumulation slack, for every window length the kernel is built for
    (24: BASELINE config 2; 12: the reference's shipped context_frames), several batch entries and tasks per workgroup;
  * a structured case that any row / pixel / frame / head / channel mix-up breaks: one-hot frame codes through Wq / Wk make the attention of head
    h copy frame perm_h(f) exactly, so the output has a closed form;
  * against the three launches it replaces (rowgemm320 -> mmgt_attention -> GEMM + residual) at the in-step shape 2 x 24 x 4096, in place
    (out aliasing x), bitwise run-to-run reproducible."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu

C, H, HD = 320, 8, 40
DEV = "cuda:0"


def _bf(x):
    return x.to(torch.bfloat16)


def _weights(tag, scale=1.0):
    from mmgt_amd.synthetic import hash_uniform
    w = {k: _bf(hash_uniform(f"tleg.{tag}.{k}", (C, C), 1.0, DEV) * (scale * C ** -0.5)) for k in ("q", "k", "v", "o")}
    w["bo"] = 0.1 * hash_uniform(f"tleg.{tag}.bo", (C,), 1.0, DEV)
    w["g"] = 1 + 0.2 * hash_uniform(f"tleg.{tag}.g", (C,), 1.0, DEV)
    w["bpe"] = 0.3 * hash_uniform(f"tleg.{tag}.bpe", (32, C), 1.0, DEV)
    return w


def _ref(x, w, B, F, n, scale, eps=1e-5):
    """fp64 with the kernel's rounding points."""
    M = B * F * n
    xd = x.double()
    mu = xd.mean(1, keepdim=True)
    var = ((xd - mu) ** 2).mean(1, keepdim=True)
    frame = (torch.arange(M, device=x.device) // n) % F
    xn = _bf(((xd - mu) / torch.sqrt(var + eps) * w["g"].double() + w["bpe"].double()[frame]).float()).double()
    q, k, v = (_bf((xn @ w[t].double().t()).float()).double() for t in ("q", "k", "v"))
    sp = lambda t: t.view(B, F, n, H, HD).permute(0, 2, 3, 1, 4)             # (B, n, H, F, HD)
    s = (sp(q) @ sp(k).transpose(-1, -2)) * scale
    e = torch.exp(s - s.max(-1, keepdim=True).values)
    pb = _bf(e.float()).double()
    o = (pb @ sp(v)) / e.sum(-1, keepdim=True)                                 # (the denominator sums the unrounded exponentials, as tattn_kernel does)
    o = _bf(o.permute(0, 3, 1, 2, 4).reshape(M, C).float()).double()
    return o @ w["o"].double().t() + w["bo"].double() + xd


def _run(x, w, B, F, n, scale, out=None):
    from mmgt_amd import hip
    from mmgt_amd.packing import pack_tleg
    img = pack_tleg(w["q"], w["k"], w["v"], w["o"])
    return hip.temporal_leg320(x, w["g"], w["bpe"], img, w["bo"], B, F, n, scale, out=out)


@pytest.mark.parametrize("B,F,n", [(2, 24, 64), (1, 24, 8), (3, 12, 48), (2, 12, 16), (1, 24, 1024)])
def test_temporal_leg_random_against_fp64(B, F, n):
    from mmgt_amd.synthetic import hash_uniform
    M = B * F * n
    x = _bf(hash_uniform(f"tleg.x{B}.{F}.{n}", (M, C), 1.5, DEV) + 0.3)
    w = _weights("rnd", scale=2.0)                   # scores of a few units: a softmax that is neither flat nor one-hot
    scale = HD ** -0.5
    ref = _ref(x, w, B, F, n, scale)
    got = _run(x, w, B, F, n, scale).double()
    torch.cuda.synchronize()
    tol = 2.0 ** -8 * ref.abs() + 8e-3               # one output ulp + flips of the bf16 intermediates (xn, q | k | v, p, o) through K = 320 sums (tail of 8 M outputs: 6.1e-3)
    d = (got - ref).abs()
    print(f"B={B} F={F} n={n}: max|d| {d.max().item():.3e} mean {d.mean().item():.3e} on mean|ref| {ref.abs().mean().item():.3f}; "
          f"worst d/tol {(d / tol).max().item():.2f}")
    assert torch.isfinite(got).all()
    assert (d <= tol).all()
    assert d.mean() <= 2.0 ** -9 * ref.abs().mean()


@pytest.mark.parametrize("F", [24, 12])
def test_temporal_leg_routes_every_pixel_frame_head(F):
    """A case with a closed form that any row / pixel / frame / head / channel mix-up breaks.  gamma = 0 makes the normalised row the table
    row alone, beta_pe[f] = 4 e_f (a one-hot frame code).  Wk copies the code into every head (k_h[f'] = 4 e_f'), Wq copies a per-head
    PERMUTED code (q_h[f] = 4 e_perm_h(f)): the score is 16 [perm_h(f) = f'], times scale 8 -> the softmax is exactly one-hot (exp2(-184)
    underflows to 0), so head h of frame f receives v_h[perm_h(f)], v[f'] = bf16(Wv beta_pe[f']).  The pixel and batch identity rides on the
    residual x (random): out = x + Wo . o + bo, to one output ulp."""
    from mmgt_amd.synthetic import hash_uniform
    assert F <= HD
    B, n = 2, 4 * 48 // F * 3
    M = B * F * n
    x = _bf(hash_uniform(f"tleg.route.x{F}", (M, C), 1.0, DEV))
    w = _weights(f"route{F}")
    w["g"] = torch.zeros(C, device=DEV)
    code = torch.zeros((32, C), device=DEV)
    wq = torch.zeros((C, C), device=DEV)
    wk = torch.zeros((C, C), device=DEV)
    gen = torch.Generator().manual_seed(F)
    perms = [torch.randperm(F, generator=gen) for _ in range(H)]
    for f in range(F):
        code[f, f] = 4.0
        for h in range(H):
            wk[HD * h + f, f] = 1.0
            wq[HD * h + int(perms[h][f]), f] = 1.0
    w["bpe"], w["q"], w["k"] = code, _bf(wq), _bf(wk)
    got = _run(x, w, B, F, n, 8.0).double()
    torch.cuda.synchronize()
    v = _bf((code[:F].double() @ w["v"].double().t()).float()).double()                                              # (F, C), the same for every pixel
    o = torch.stack([torch.cat([v[int(perms[h][f]), HD * h:HD * (h + 1)] for h in range(H)]) for f in range(F)])     # (F, C)
    att = o @ w["o"].double().t() + w["bo"].double()
    ref = x.double().view(B, F, n, C) + att[None, :, None, :]
    d = (got.view(B, F, n, C) - ref).abs()
    bad = d > 2.0 ** -8 * ref.abs() + 1e-3
    assert not bad.any(), (int(bad.sum()), bad.nonzero()[:5].tolist(), d.max().item())


def test_temporal_leg_equals_the_three_launches_at_step_shape_in_place_and_reproducible():
    from mmgt_amd import hip
    from mmgt_amd.packing import pack_rowgemm, pack_tleg
    from mmgt_amd.synthetic import hash_uniform
    B, F, n = 2, 24, 4096
    M = B * F * n
    x = _bf(hash_uniform("tleg.step.x", (M, C), 1.5, DEV))
    w = _weights("step", scale=2.0)
    scale = HD ** -0.5
    # the launches it replaces (mmgt_amd/unet3d.py::_motion_module, round 4)
    qkv, _ = hip.rowgemm320(x, pack_rowgemm(torch.cat([w["q"], w["k"], w["v"]])), 3 * C, ln_gamma=w["g"], ln_beta=w["bpe"].contiguous(), pe_div=n, pe_mod=F)
    o = torch.empty((M, C), device=DEV, dtype=torch.bfloat16)
    st = (F * n * 3 * C, 3 * C, n * 3 * C)
    hip.attention(qkv, qkv[:, C:], qkv[:, 2 * C:], o, batch=B * n, heads=H, hd=HD, nq=F, nk=F, scale=scale, q_str=st, k_str=st, v_str=st,
                  o_str=(F * n * C, C, n * C), bdiv=n)
    ref = hip.gemm(o, w["o"], w["bo"], residual=x)
    img = pack_tleg(w["q"], w["k"], w["v"], w["o"])
    got = hip.temporal_leg320(x, w["g"], w["bpe"], img, w["bo"], B, F, n, scale)
    torch.cuda.synchronize()
    # The two paths do not round at the same points (tattn_kernel rounds the SCALED q to bf16 once more and sums the unrounded probabilities),
    # so they are held against each other loosely and against fp64 of the fused kernel's rounding points tightly: the fused leg must be at
    # least as close to it as the three launches are.
    d = (got.float() - ref.float()).abs()
    print(f"fused leg vs three launches: max|d| {d.max().item():.3e} mean {d.mean().item():.3e}")
    assert torch.isfinite(got).all() and d.max() <= 8 * 2.0 ** -8 * ref.float().abs().max() and d.mean() <= 2.0 ** -8 * ref.float().abs().mean()
    r64 = _ref(x, w, B, F, n, scale)
    e_f, e_3 = (got.double() - r64).abs(), (ref.double() - r64).abs()
    print(f"against fp64: fused mean {e_f.mean().item():.3e} max {e_f.max().item():.3e} | three launches mean {e_3.mean().item():.3e} max {e_3.max().item():.3e}")
    assert (e_f <= 2.0 ** -8 * r64.abs() + 8e-3).all() and e_f.mean() <= 1.05 * e_3.mean()
    del r64, e_f, e_3
    again = hip.temporal_leg320(x, w["g"], w["bpe"], img, w["bo"], B, F, n, scale)
    assert torch.equal(got, again), "not bitwise reproducible"
    xin = x.clone()
    hip.temporal_leg320(xin, w["g"], w["bpe"], img, w["bo"], B, F, n, scale, out=xin)
    torch.cuda.synchronize()
    assert torch.equal(xin, got), "in-place result differs"


def test_temporal_leg_rejects_what_it_is_not_built_for():
    from mmgt_amd import hip
    from mmgt_amd.packing import pack_tleg
    w = _weights("rej")
    img = pack_tleg(w["q"], w["k"], w["v"], w["o"])
    x = torch.zeros((2 * 10 * 8, C), device=DEV, dtype=torch.bfloat16)
    with pytest.raises(RuntimeError, match="temporal_leg320"):
        hip.temporal_leg320(x, w["g"], w["bpe"], img, w["bo"], 2, 10, 8, 1.0)            # neither 24 nor 12 frames
    x = torch.zeros((24 * 12, C), device=DEV, dtype=torch.bfloat16)
    with pytest.raises(RuntimeError, match="temporal_leg320"):
        hip.temporal_leg320(x, w["g"], w["bpe"], img, w["bo"], 1, 24, 12, 1.0)           # 12 pixels: not a multiple of 8
    assert not hip.temporal_leg320_supported(torch.float32, 320, 8, 24, 4096)
    assert hip.temporal_leg320_supported(torch.bfloat16, 320, 8, 12, 4096)
