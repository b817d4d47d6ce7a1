"""mmgt_ff_fused (csrc/ffn.hip): LayerNorm -> GEGLU FeedForward -> + residual in one launch, through the C ABI.

  * exact-integer case: operands chosen so that every product and sum is an exactly representable integer -> the result must equal
    the int64 reference rounded once to bf16, BIT FOR BIT (any fragment / k-order / bias / epilogue indexing slip shows here);
  * random case against fp64 of the same bf16-rounded operands with the kernel's two rounding points (LN output, GEGLU output)
    reproduced, gate: one output bf16 ulp + the fp32 accumulation bound;
  * against the three-launch path (LayerNorm, GEMM + GEGLU, GEMM + residual) it replaces, at the in-step shape."""
import pytest
import torch

pytestmark = pytest.mark.gpu

C, INNER = 320, 1280


def _bf(x):
    return x.to(torch.bfloat16)


def _ref(x, g, b, w1, b1, w2, b2, res, eps=1e-5):
    """fp64 restatement with the kernel's rounding points (diffusers FeedForward geglu)."""
    xd = x.double()
    if g is not None:
        mu = xd.mean(1, keepdim=True)
        var = ((xd - mu) ** 2).mean(1, keepdim=True)
        xd = _bf(((xd - mu) / torch.sqrt(var + eps) * g.double() + b.double()).float()).double()
    hg = xd @ w1.double().t() + b1.double()
    h, gate = hg[:, :INNER], hg[:, INNER:]
    act = _bf((h * torch.nn.functional.gelu(gate)).float()).double()
    return act @ w2.double().t() + b2.double() + res.double()


def _run(x, g, b, w1, b1, w2, b2, res):
    from mmgt_amd import hip
    from mmgt_amd.packing import pack_ff_fused
    dev = "cuda:0"
    img = pack_ff_fused(w1.to(dev), b1.to(dev).float(), w2.to(dev))
    mv = lambda t: None if t is None else t.to(dev).contiguous()
    out = hip.ff_fused(mv(x), mv(g), mv(b), img, mv(b2.float()), mv(res), INNER)
    torch.cuda.synchronize()
    return out.cpu()


@pytest.mark.parametrize("M", [128, 5000])
def test_ff_fused_exact_integers(M):
    gen = torch.Generator().manual_seed(M)
    ri = lambda shape, lo, hi: torch.randint(lo, hi + 1, shape, generator=gen).float()
    x = _bf(ri((M, C), -2, 2))
    # h = sparse +-1 rows of W1 . x (|h| <= 8 * 2 = 16); gate rows are zero with bias 16 -> gelu(16) == 16 exactly in fp32
    w1 = torch.zeros(2 * INNER, C)
    cols = torch.stack([torch.randperm(C, generator=gen)[:8] for _ in range(INNER)])
    w1[:INNER].scatter_(1, cols, ri((INNER, 8), -1, 1))
    b1 = torch.cat([ri((INNER,), -3, 3), torch.full((INNER,), 16.0)])
    w2 = torch.zeros(C, INNER)
    cols2 = torch.stack([torch.randperm(INNER, generator=gen)[:6] for _ in range(C)])
    w2.scatter_(1, cols2, ri((C, 6), -1, 1))
    b2 = ri((C,), -4, 4)
    res = _bf(ri((M, C), -8, 8))
    ref = _ref(x, None, None, _bf(w1), b1, _bf(w2), b2, res)
    assert ref.abs().max() < 2 ** 22 and torch.equal(ref, ref.round())
    out = _run(x, None, None, _bf(w1), b1, _bf(w2), b2, res)
    assert torch.equal(out, _bf(ref.float())), f"max|d| {(out.double() - ref).abs().max().item()}"


@pytest.mark.parametrize("M,ln", [(4096, True), (777, True), (4096, False)])
def test_ff_fused_random_against_fp64(M, ln):
    from mmgt_amd.synthetic import hash_uniform
    x = _bf(hash_uniform(f"ffn.x{M}", (M, C), 1.5) + 0.3)
    g = (1 + 0.2 * hash_uniform("ffn.g", (C,), 1.0)) if ln else None
    b = 0.1 * hash_uniform("ffn.b", (C,), 1.0) if ln else None
    w1 = _bf(hash_uniform("ffn.w1", (2 * INNER, C), 1.0) * C ** -0.5)
    b1 = 0.1 * hash_uniform("ffn.b1", (2 * INNER,), 1.0)
    w2 = _bf(hash_uniform("ffn.w2", (C, INNER), 1.0) * INNER ** -0.5)
    b2 = 0.1 * hash_uniform("ffn.b2", (C,), 1.0)
    res = x if ln else _bf(hash_uniform("ffn.res", (M, C), 1.0))
    ref = _ref(x, g, b, w1, b1, w2, b2, res)
    out = _run(x, g, b, w1, b1, w2, b2, res).double()
    # one bf16 ulp of the output (2^-8 relative, half of it from the final rounding) + fp32 accumulation over K = 320 / 1280 and the
    # rare bf16 rounding flips of the two intermediate tensors (each moves one product by 2^-8 of a term of size ~ |w| |act|)
    tol = 2.0 ** -8 * ref.abs() + 4e-3
    d = (out - ref).abs()
    print(f"M={M} ln={ln}: max|d| {d.max().item():.3e} mean {d.mean().item():.3e} on mean|ref| {ref.abs().mean().item():.3f}; worst d/tol {(d / tol).max().item():.2f}")
    assert (d <= tol).all()
    assert d.mean() <= 2.0 ** -9 * ref.abs().mean()


def test_ff_fused_equals_three_launch_path_at_step_shape():
    """M = 196 608 (48 frames x 4096 tokens): same operands through LayerNorm -> gemm(GEGLU) -> gemm(+residual)."""
    from mmgt_amd import hip
    from mmgt_amd.packing import pack_ff_fused, pack_geglu
    from mmgt_amd.synthetic import hash_uniform
    dev = "cuda:0"
    M = 48 * 4096
    x = _bf(hash_uniform("ffn.step.x", (M, C), 1.5, dev))
    g, b = (1 + 0.2 * hash_uniform("ffn.g", (C,), 1.0, dev)), 0.1 * hash_uniform("ffn.b", (C,), 1.0, dev)
    w1 = _bf(hash_uniform("ffn.w1", (2 * INNER, C), 1.0, dev) * C ** -0.5)
    b1 = 0.1 * hash_uniform("ffn.b1", (2 * INNER,), 1.0, dev)
    w2 = _bf(hash_uniform("ffn.w2", (C, INNER), 1.0, dev) * INNER ** -0.5)
    b2 = 0.1 * hash_uniform("ffn.b2", (C,), 1.0, dev)
    fused = hip.ff_fused(x, g, b, pack_ff_fused(w1, b1, w2), b2, x, INNER)
    wp, bp = pack_geglu(w1, b1)
    n3 = hip.layernorm(x, g, b, 1e-5)
    three = hip.gemm(hip.gemm(n3, wp.contiguous(), bp.contiguous(), act=hip.ACT_GEGLU), w2, b2, residual=x)
    torch.cuda.synchronize()
    d = (fused.float() - three.float()).abs()
    ulp = 2.0 ** -8 * three.float().abs() + 1e-3
    frac = (d > ulp).float().mean().item()
    print(f"fused vs three launches: max|d| {d.max().item():.3e}, {100 * frac:.4f}% of outputs differ by more than one bf16 ulp")
    assert torch.isfinite(fused).all() and d.max() <= 4 * 2.0 ** -8 * three.float().abs().max() and frac < 1e-3
    again = hip.ff_fused(x, g, b, pack_ff_fused(w1, b1, w2), b2, x, INNER)
    assert torch.equal(fused, again), "not bitwise reproducible"


# ---- proj_out on the end of the same launch (mmgt_ff_fused_po)

def _run_po(x, g, b, w1, b1, w2, b2, res, wpo, bpo, res2):
    from mmgt_amd import hip
    from mmgt_amd.packing import pack_ff_fused, pack_ff_proj_out
    dev = "cuda:0"
    img = pack_ff_fused(w1.to(dev), b1.to(dev).float(), w2.to(dev))
    mv = lambda t: None if t is None else t.to(dev).contiguous()
    out = hip.ff_fused_po(mv(x), mv(g), mv(b), img, mv(b2.float()), mv(res), INNER, pack_ff_proj_out(wpo.to(dev)), mv(bpo.float()), mv(res2))
    torch.cuda.synchronize()
    return out.cpu()


@pytest.mark.parametrize("M", [128, 3000])
def test_ff_fused_proj_out_exact_integers(M):
    """FeedForward on exact integers (gate = 8: gelu(8) == 8 in fp32) -> hidden (integers <= 128: exact in bf16) -> proj_out with sparse
    +-1 rows: bit for bit."""
    gen = torch.Generator().manual_seed(7 * M)
    ri = lambda shape, lo, hi: torch.randint(lo, hi + 1, shape, generator=gen).float()
    x = _bf(ri((M, C), -1, 1))
    w1 = torch.zeros(2 * INNER, C)
    cols = torch.stack([torch.randperm(C, generator=gen)[:4] for _ in range(INNER)])
    w1[:INNER].scatter_(1, cols, ri((INNER, 4), -1, 1))
    b1 = torch.cat([ri((INNER,), -3, 3), torch.full((INNER,), 8.0)])
    w2 = torch.zeros(C, INNER)
    cols2 = torch.stack([torch.randperm(INNER, generator=gen)[:2] for _ in range(C)])
    w2.scatter_(1, cols2, ri((C, 2), -1, 1))
    b2 = ri((C,), -4, 4)
    res = _bf(ri((M, C), -8, 8))
    hidden = _ref(x, None, None, _bf(w1), b1, _bf(w2), b2, res)
    assert hidden.abs().max() <= 128 and torch.equal(hidden, hidden.round())          # exact in bf16
    wpo = torch.zeros(C, C)
    cols3 = torch.stack([torch.randperm(C, generator=gen)[:6] for _ in range(C)])
    wpo.scatter_(1, cols3, ri((C, 6), -1, 1))
    bpo = ri((C,), -4, 4)
    res2 = _bf(ri((M, C), -8, 8))
    ref = hidden @ wpo.double().t() + bpo.double() + res2.double()
    assert ref.abs().max() < 2 ** 15 and torch.equal(ref, ref.round())
    out = _run_po(x, None, None, _bf(w1), b1, _bf(w2), b2, res, _bf(wpo), bpo, res2)
    assert torch.equal(out, _bf(ref.float())), f"max|d| {(out.double() - ref).abs().max().item()}"


def test_ff_fused_proj_out_random_against_fp64_and_the_two_launches():
    from mmgt_amd import hip
    from mmgt_amd.packing import pack_ff_fused
    from mmgt_amd.synthetic import hash_uniform
    M = 4096
    x = _bf(hash_uniform("ffn.x4096", (M, C), 1.5) + 0.3)
    g, b = 1 + 0.2 * hash_uniform("ffn.g", (C,), 1.0), 0.1 * hash_uniform("ffn.b", (C,), 1.0)
    w1 = _bf(hash_uniform("ffn.w1", (2 * INNER, C), 1.0) * C ** -0.5)
    b1 = 0.1 * hash_uniform("ffn.b1", (2 * INNER,), 1.0)
    w2 = _bf(hash_uniform("ffn.w2", (C, INNER), 1.0) * INNER ** -0.5)
    b2 = 0.1 * hash_uniform("ffn.b2", (C,), 1.0)
    wpo = _bf(hash_uniform("ffn.wpo", (C, C), 1.0) * C ** -0.5)
    bpo = 0.1 * hash_uniform("ffn.bpo", (C,), 1.0)
    res2 = _bf(hash_uniform("ffn.res2", (M, C), 1.0))
    dev = "cuda:0"
    # the hidden tensor exactly as the kernel rounds it: the stand-alone launch's output
    hidden = hip.ff_fused(x.to(dev), g.to(dev), b.to(dev), pack_ff_fused(w1.to(dev), b1.to(dev), w2.to(dev)), b2.to(dev), x.to(dev), INNER)
    two = hip.gemm(hidden, wpo.to(dev), bpo.to(dev), residual=res2.to(dev)).cpu()
    ref = hidden.cpu().double() @ wpo.double().t() + bpo.double() + res2.double()
    out = _run_po(x, g, b, w1, b1, w2, b2, x, wpo, bpo, res2)
    d = (out.double() - ref).abs()
    tol = 2.0 ** -8 * ref.abs() + 2e-3
    print(f"ff_fused_po: max|d| {d.max().item():.3e}, worst d/tol {(d / tol).max().item():.2f}; vs ff_fused + gemm: {(out.float() - two.float()).abs().max().item():.3e}")
    assert (d <= tol).all() and d.mean() <= 2.0 ** -9 * ref.abs().mean()
    frac = ((out.float() - two.float()).abs() > 2.0 ** -8 * two.float().abs() + 1e-3).float().mean().item()
    assert frac < 1e-3
