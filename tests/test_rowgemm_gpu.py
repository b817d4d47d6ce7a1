"""mmgt_rowgemm320 (csrc/rowgemm.hip): [LayerNorm ->] Linear(s) of the 320-channel level with the rows stationary, through the C ABI.

  * exact-integer cases: every product and sum is an exactly representable integer -> the result must equal the int64 reference
    rounded once to bf16, BIT FOR BIT, for normal tiles, transposed (V^T) tiles, bias / bias2 row groups / residual, ragged M;
  * random cases against fp64 of the same bf16-rounded operands with the kernel's rounding point (the LayerNorm output) reproduced,
    gate: one output bf16 ulp + the fp32 accumulation bound; the motion module's per-frame beta rows;
  * at the in-step shape (M = 196 608) against the launches it replaces: LayerNorm -> GEMM (q | k) and the batched W . X^T GEMM (V^T),
    and out-projection + residual; bitwise reproducible."""
import pytest
import torch

pytestmark = pytest.mark.gpu

C = 320


def _bf(x):
    return x.to(torch.bfloat16)


def _ref(x, w, bias, g=None, b=None, pe_div=0, eps=1e-5, res=None, bias2=None, bias2_rows=0):
    """fp64 restatement with the kernel's rounding point (diffusers Attention to_q / to_k / to_v / to_out)."""
    xd = x.double()
    M = x.shape[0]
    if g is not None:
        mu = xd.mean(1, keepdim=True)
        var = ((xd - mu) ** 2).mean(1, keepdim=True)
        bb = b.double().reshape(-1, C)
        rows = (torch.arange(M) // max(pe_div, 1)) % bb.shape[0]
        xd = _bf(((xd - mu) / torch.sqrt(var + eps) * g.double() + bb[rows]).float()).double()
    y = xd @ w.double().t()
    if bias is not None:
        y = y + bias.double()
    if bias2 is not None:
        y = y + bias2.double()[torch.arange(M) // bias2_rows]
    if res is not None:
        y = y + res.double()
    return y


def _run(x, w, bias, n1=None, n_tok=0, **kw):
    from mmgt_amd import hip
    from mmgt_amd.packing import pack_rowgemm
    dev = "cuda:0"
    mv = lambda t: None if t is None else t.to(dev).contiguous()
    N = w.shape[0]
    out, out_t = hip.rowgemm320(mv(x), pack_rowgemm(w.to(dev)), N, mv(None if bias is None else bias.float()), n1=n1, n_tok=n_tok,
                                **{k: (mv(v) if torch.is_tensor(v) else v) for k, v in kw.items()})
    torch.cuda.synchronize()
    return (None if out is None else out.cpu()), (None if out_t is None else out_t.cpu())


def _untranspose(out_t, n_tok):
    """(M / n_tok, R, npad) -> (M, R)"""
    return out_t[:, :, :n_tok].permute(0, 2, 1).reshape(-1, out_t.shape[1])


@pytest.mark.parametrize("M,N,n1,res", [(128, 320, 320, True), (5000, 320, 320, True), (777, 960, 960, False), (1024, 960, 640, False),
                                          (256, 64, 0, False)])
def test_rowgemm_exact_integers(M, N, n1, res):
    gen = torch.Generator().manual_seed(M + N)
    ri = lambda shape, lo, hi: torch.randint(lo, hi + 1, shape, generator=gen).float()
    x = _bf(ri((M, C), -3, 3))
    w = torch.zeros(N, C)
    cols = torch.stack([torch.randperm(C, generator=gen)[:12] for _ in range(N)])
    w.scatter_(1, cols, ri((N, 12), -2, 2))
    bias = ri((N,), -5, 5)
    r = _bf(ri((M, N), -8, 8)) if res else None
    b2rows = 128 if M % 128 == 0 else 0
    bias2 = ri(((M + 127) // 128, N), -3, 3) if b2rows else None
    ref = _ref(x, _bf(w), bias, res=r, bias2=bias2, bias2_rows=b2rows)
    assert ref.abs().max() < 256 and torch.equal(ref, ref.round())
    n_tok = 128 if n1 < N else 0
    out, out_t = _run(x, _bf(w), bias, n1=n1, n_tok=n_tok, residual=r, bias2=bias2, bias2_rows=b2rows)
    got = torch.cat([t for t in (out, None if out_t is None else _untranspose(out_t, n_tok)) if t is not None], 1)
    assert torch.equal(got, _bf(ref.float())), f"max|d| {(got.double() - ref).abs().max().item()}"


@pytest.mark.parametrize("M,N,n1,ln,pe", [(4096, 960, 640, True, 0), (1000, 320, 320, True, 0), (2048, 960, 960, True, 4), (4096, 320, 320, False, 0)])
def test_rowgemm_random_against_fp64(M, N, n1, ln, pe):
    from mmgt_amd.synthetic import hash_uniform
    x = _bf(hash_uniform(f"rg.x{M}", (M, C), 1.5) + 0.3)
    g = (1 + 0.2 * hash_uniform("rg.g", (C,), 1.0)) if ln else None
    b = 0.1 * hash_uniform("rg.b", (max(pe, 1), C), 1.0) if ln else None
    w = _bf(hash_uniform(f"rg.w{N}", (N, C), 1.0) * C ** -0.5)
    bias = 0.1 * hash_uniform("rg.bias", (N,), 1.0)
    res = None if ln else _bf(hash_uniform("rg.res", (M, N), 1.0))
    pe_div = 256 if pe else 0
    ref = _ref(x, w, bias, g, b, pe_div=pe_div, res=res)
    n_tok = 256 if n1 < N else 0
    out, out_t = _run(x, w, bias, n1=n1, n_tok=n_tok, ln_gamma=g, ln_beta=b, pe_div=pe_div, pe_mod=pe, residual=res)
    got = torch.cat([t for t in (out, None if out_t is None else _untranspose(out_t, n_tok)) if t is not None], 1).double()
    tol = 2.0 ** -8 * ref.abs() + 2e-3        # one output bf16 ulp + fp32 accumulation over K = 320 and rare flips of the bf16 LayerNorm output
    d = (got - ref).abs()
    print(f"M={M} N={N} ln={ln}: max|d| {d.max().item():.3e} mean {d.mean().item():.3e} on mean|ref| {ref.abs().mean().item():.3f}; worst d/tol {(d / tol).max().item():.2f}")
    assert (d <= tol).all()
    assert d.mean() <= 2.0 ** -9 * ref.abs().mean()


def test_rowgemm_equals_the_launches_it_replaces_at_step_shape():
    """M = 196 608 (48 images x 4096 tokens): LayerNorm -> gemm(q | k) + gemm_batched_wx(V^T), and out-projection + residual."""
    from mmgt_amd import hip
    from mmgt_amd.packing import pack_rowgemm
    from mmgt_amd.synthetic import hash_uniform
    dev = "cuda:0"
    nb, n = 48, 4096
    M = nb * n
    x = _bf(hash_uniform("rg.step.x", (M, C), 1.5, dev))
    g, b = (1 + 0.2 * hash_uniform("rg.g", (C,), 1.0, dev)), 0.1 * hash_uniform("rg.b", (C,), 1.0, dev)
    wqk = _bf(hash_uniform("rg.wqk", (2 * C, C), 1.0, dev) * C ** -0.5)
    wv = _bf(hash_uniform("rg.wv", (C, C), 1.0, dev) * C ** -0.5)
    n1 = hip.layernorm(x, g, b, 1e-5)
    qk_ref = hip.gemm(n1, wqk)
    vt_ref = torch.empty((nb, C, n), device=dev, dtype=torch.bfloat16)
    hip.gemm_batched_wx(wv, n1.view(nb, n, C), out=vt_ref)
    img = pack_rowgemm(torch.cat([wqk, wv]))
    qk, vt = hip.rowgemm320(x, img, 3 * C, ln_gamma=g, ln_beta=b, n1=2 * C, n_tok=n)
    torch.cuda.synchronize()
    for name, got, ref in (("q|k", qk, qk_ref), ("V^T", vt, vt_ref)):
        d = (got.float() - ref.float()).abs()
        frac = (d > 2.0 ** -8 * ref.float().abs() + 1e-3).float().mean().item()
        print(f"{name}: max|d| {d.max().item():.3e}, {100 * frac:.4f}% of outputs differ by more than one bf16 ulp")
        assert torch.isfinite(got).all() and d.max() <= 4 * 2.0 ** -8 * ref.float().abs().max() and frac < 1e-3
    qk2, vt2 = hip.rowgemm320(x, img, 3 * C, ln_gamma=g, ln_beta=b, n1=2 * C, n_tok=n)
    assert torch.equal(qk, qk2) and torch.equal(vt, vt2), "not bitwise reproducible"
    wo = _bf(hash_uniform("rg.wo", (C, C), 1.0, dev) * C ** -0.5)
    bo = 0.1 * hash_uniform("rg.bo", (C,), 1.0, dev)
    o_ref = hip.gemm(qk[:, :C].contiguous(), wo, bo, residual=x)
    o, _ = hip.rowgemm320(qk[:, :C].contiguous(), pack_rowgemm(wo), C, bo, residual=x)
    torch.cuda.synchronize()
    d = (o.float() - o_ref.float()).abs()
    frac = (d > 2.0 ** -8 * o_ref.float().abs() + 1e-3).float().mean().item()
    print(f"out-proj + residual: max|d| {d.max().item():.3e}, {100 * frac:.4f}% beyond one ulp")
    assert d.max() <= 4 * 2.0 ** -8 * o_ref.float().abs().max() and frac < 1e-3


def test_rowgemm_layernorm_statistics_with_a_large_row_mean():
    """The fused LayerNorm takes its variance from one pass (E[x^2] - mean^2 in fp32): rows with |mean| = 8 std against fp64."""
    from mmgt_amd.synthetic import hash_uniform
    M, N = 1024, 320
    x = _bf(hash_uniform("rg.big.x", (M, C), 1.5) + 7.0)
    g = 1 + 0.2 * hash_uniform("rg.g", (C,), 1.0)
    b = 0.1 * hash_uniform("rg.b", (1, C), 1.0)
    w = _bf(hash_uniform("rg.w320", (N, C), 1.0) * C ** -0.5)
    ref = _ref(x, w, None, g, b)
    out, _ = _run(x, w, None, ln_gamma=g, ln_beta=b)
    d = (out.double() - ref).abs()
    tol = 2.0 ** -8 * ref.abs() + 2e-3
    print(f"|mean| = 8 std: max|d| {d.max().item():.3e}, worst d/tol {(d / tol).max().item():.2f}")
    assert (d <= tol).all() and d.mean() <= 2.0 ** -9 * ref.abs().mean()


def test_rowgemm_groupnorm_prologue_is_bitwise_the_two_launch_path():
    """GroupNorm -> proj_in (transformer_3d.py:174-188): statistics as scale / shift tables + one launch that applies them while it
    loads the rows, against groupnorm() followed by the same rowgemm without a prologue -- the same arithmetic, bit for bit --
    and against gemm() within one ulp."""
    from mmgt_amd import hip
    from mmgt_amd.packing import pack_rowgemm
    from mmgt_amd.synthetic import hash_uniform
    dev = "cuda:0"
    nb, n = 6, 4096
    x = _bf(hash_uniform("rg.gn.x", (nb, n, C), 1.5, dev) + 0.5 * hash_uniform("rg.gn.off", (nb, 1, C), 1.0, dev))
    g, b = 1 + 0.2 * hash_uniform("rg.gn.g", (C,), 1.0, dev), 0.1 * hash_uniform("rg.gn.b", (C,), 1.0, dev)
    w = _bf(hash_uniform("rg.gn.w", (C, C), 1.0, dev) * C ** -0.5)
    bias = 0.1 * hash_uniform("rg.gn.bias", (C,), 1.0, dev)
    img = pack_rowgemm(w)
    xn = hip.groupnorm(x, g, b, 32, 1e-6)
    two, _ = hip.rowgemm320(xn.view(nb * n, C), img, C, bias)
    sc, sh = hip.groupnorm_affine(x, g, b, 32, 1e-6)
    one, _ = hip.rowgemm320(x.view(nb * n, C), img, C, bias, pre_scale=sc, pre_shift=sh, pre_rows=n)
    ref = hip.gemm(xn.view(nb * n, C), w, bias)
    torch.cuda.synchronize()
    assert torch.equal(one, two), f"max|d| {(one.float() - two.float()).abs().max().item()}"
    d = (one.float() - ref.float()).abs()
    frac = (d > 2.0 ** -8 * ref.float().abs() + 1e-3).float().mean().item()
    assert d.max() <= 4 * 2.0 ** -8 * ref.float().abs().max() and frac < 1e-3
