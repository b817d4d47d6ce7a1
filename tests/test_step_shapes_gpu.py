"""The bf16-only production kernels (gemm16: csrc/gemm16.hip, attn64: csrc/attn64.hip) at the shapes they run at inside the
512x512x24 denoise step, against fp64 math of the same bf16-rounded operands (torch fp64 on the device is the checker here, as in
test_hip_kernels.py).  Gate per element: ONE output bf16 ulp (2^-8 relative: half an ulp of final rounding + half an ulp of
slack) plus the accumulation bound of the kernel's fp32 sums -- not the 2e-2 / 2e-2 of the small-shape tests."""
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from mmgt_amd.synthetic import hash_uniform  # noqa: E402

DEV = "cuda:0"
ULP = 2.0 ** -8


def rnd(name, shape, scale=1.0):
    return hash_uniform(name, shape, scale, DEV).to(torch.bfloat16)


def check(out, ref, acc_bound, what):
    d = (out.double() - ref).abs()
    tolr = ULP * ref.abs() + acc_bound
    worst = (d / tolr).max().item()
    print(f"{what}: max|d| {d.max().item():.3e} mean|d| {d.mean().item():.3e} on mean|ref| {ref.abs().mean().item():.3f}; worst d / gate {worst:.2f}")
    assert torch.isfinite(out).all() and worst <= 1.0, what
    assert d.mean() <= 2.0 ** -9 * ref.abs().mean() + acc_bound, what        # rounding errors average well below the per-element bound


def test_gemm16_geglu_at_level0_shape():
    """ff1 of the level-0 transformer blocks as the three-launch path runs it: M = 196 608, N = 2560 (packed GEGLU), K = 320."""
    from mmgt_amd import hip
    from mmgt_amd.packing import pack_geglu
    M, N, K = 48 * 4096, 2560, 320
    a, w, b = rnd("s.a", (M, K)), rnd("s.w1", (N, K), K ** -0.5), hash_uniform("s.b1", (N,), 0.2, DEV)
    wp, bp = pack_geglu(w, b)
    out = hip.gemm(a, wp.contiguous(), bp.contiguous(), act=hip.ACT_GEGLU)
    for r0 in range(0, M, 32768):                                    # fp64 reference in row chunks (4 GB as a whole)
        hg = a[r0:r0 + 32768].double() @ w.double().t() + b.double()
        ref = hg[:, :N // 2] * F.gelu(hg[:, N // 2:])
        check(out[r0:r0 + 32768], ref, 2e-5, f"GEGLU rows {r0}..")


def test_gemm16_residual_projection_at_level0_shape():
    """to_out / proj_out of level 0: M = 196 608, N = K = 320, + bias + residual (the HBM-bound 256 x 320 tile)."""
    from mmgt_amd import hip
    M, N, K = 48 * 4096, 320, 320
    a, w, b, res = rnd("s.a2", (M, K)), rnd("s.w2", (N, K), K ** -0.5), hash_uniform("s.b2", (N,), 0.2, DEV), rnd("s.r2", (M, N))
    out = hip.gemm(a, w, b, residual=res)
    ref = a.double() @ w.double().t() + b.double() + res.double()
    check(out, ref, 2e-5, "N = K = 320 + residual")


def test_gemm16_conv3x3_at_level0_shape():
    """conv 320 -> 320 at 64 x 64 over 48 frames (implicit GEMM: M = 196 608, K = 2880), + bias + per-CFG-row time embedding + residual."""
    from mmgt_amd import hip
    from mmgt_amd.packing import pack_conv3x3
    nb, h, c = 48, 64, 320
    x = rnd("s.x", (nb, h, h, c))
    w = rnd("s.wc", (c, c, 3, 3), (9 * c) ** -0.5)
    b, res = hash_uniform("s.bc", (c,), 0.2, DEV), rnd("s.rc", (nb, h, h, c))
    temb = hash_uniform("s.temb", (2, c), 0.3, DEV)
    out = hip.conv3x3(x, pack_conv3x3(w), b, bias2=temb, bias2_rows=(nb // 2) * h * h, residual=res)
    for n0 in range(0, nb, 12):
        ref = F.conv2d(x[n0:n0 + 12].permute(0, 3, 1, 2).double(), w.double(), b.double(), padding=1).permute(0, 2, 3, 1)
        ref = ref + temb[n0 // 24].double() + res[n0:n0 + 12].double()
        check(out[n0:n0 + 12], ref, 4e-5, f"conv frames {n0}..")


@pytest.mark.parametrize("pairs", [[(0, 0), (7, 3), (23, 7), (24, 0), (36, 5), (47, 7)]])
def test_attn64_with_bank_at_level0_shape(pairs):
    """Spatial attention of level 0 as it runs in the step: 48 frames x 8 heads x 4096 queries, head_dim 40, 4096 own keys and, for the
    conditional half (frames 24..47), 4096 reference-bank keys (mutual_self_attention.py:149-188).  fp64 softmax attention on six
    (frame, head) pairs of both halves.  P is a bf16 MFMA operand, so on top of the output ulp every output carries
    sum_j p_j v_j 2^-9 r_j noise (|v| <= 1: below 2^-9 even for a one-hot softmax, ~1e-4 for these spread-out rows)."""
    from mmgt_amd import hip
    hd, n, nb, f = 40, 4096, 48, 24
    inner = 8 * hd
    qk, vt = rnd("s.qk", (nb * n, 2 * inner), 2.0), rnd("s.vt", (nb, inner, n))
    kb, vbt = rnd("s.kb", (2, n, inner), 2.0), rnd("s.vbt", (2, inner, n))
    o = torch.empty((nb * n, inner), device=DEV, dtype=torch.bfloat16)
    hip.attention(qk, qk[:, inner:], vt, o, batch=nb, heads=8, hd=hd, nq=n, nk=n, scale=hd ** -0.5,
                  q_str=(n * 2 * inner, 0, 2 * inner), k_str=(n * 2 * inner, 0, 2 * inner), v_str=(inner * n, 0, n),
                  o_str=(n * inner, 0, inner), v_transposed=True, k2=kb, v2=vbt, k2_str=(kb.stride(0), kb.stride(1)),
                  v2_str=(vbt.stride(0), vbt.stride(1)), k2_bdiv=f, nk2=n, seg2_first_batch=nb // 2)
    qk3, o3 = qk.view(nb, n, 2 * inner), o.view(nb, n, inner)
    for b, hh in pairs:
        sl = slice(hh * hd, (hh + 1) * hd)
        q, k, v = qk3[b, :, sl].double(), qk3[b, :, inner:][:, sl].double(), vt[b, sl].double().t()
        if b >= nb // 2:                       # the conditional half also attends to the bank (row 1 of the (2, N, C) bank)
            k, v = torch.cat([k, kb[1, :, sl].double()]), torch.cat([v, vbt[1, sl].double().t()])
        ref = torch.softmax(q @ k.t() * hd ** -0.5, dim=-1) @ v
        check(o3[b, :, sl], ref, 2.0 ** -9, f"attention frame {b} head {hh} ({k.shape[0]} keys)")


@pytest.mark.parametrize("kind,shape", [("conv", (48, 8, 1280, 0, 1280)), ("conv", (48, 8, 1280, 1280, 1280)), ("conv", (6, 8, 640, 0, 512)),
                                        ("gemm", (3072, 1280, 5120)), ("gemm", (1000, 640, 2560))])
def test_gemm16_splitk_equals_unsplit_and_fp64(kind, shape):
    """Split-K (the 8x8-level convs / ff2: 60 tiles for 256 CUs): on small-integer operands every partial sum is exact, so the sliced
    reduction + fixed-order slab sum must equal the unsplit kernel BIT FOR BIT (slice start inside a tap, inside the second source of
    a concatenated input, ragged M); on random operands both are held to one output ulp against fp64."""
    from mmgt_amd import hip
    from mmgt_amd.packing import pack_conv3x3
    gen = torch.Generator(device=DEV).manual_seed(7)
    ri = lambda shape, lo, hi: torch.randint(lo, hi + 1, shape, generator=gen, device=DEV).float()

    def run(exact):
        if kind == "conv":
            nb, h, c0, c1, cout = shape
            mk = (lambda s, sc: ri(s, -2, 2).bfloat16()) if exact else (lambda s, sc: rnd(f"sk.{s}", s, sc))
            x0 = mk((nb, h, h, c0), 1.0)
            x1 = mk((nb, h, h, c1), 1.0) if c1 else None
            w = (ri((cout, c0 + c1, 3, 3), -1, 1) * (torch.rand((cout, c0 + c1, 3, 3), generator=gen, device=DEV) < 0.02)).bfloat16() if exact \
                else rnd("sk.w", (cout, c0 + c1, 3, 3), (9 * (c0 + c1)) ** -0.5)
            b = ri((cout,), -3, 3) if exact else hash_uniform("sk.b", (cout,), 0.2, DEV)
            res = mk((nb, h, h, cout), 1.0)
            wp = pack_conv3x3(w)
            f = lambda: hip.conv3x3(x0, wp, b, residual=res, x1=x1)
            xin = x0 if x1 is None else torch.cat([x0, x1], 3)
            ref = F.conv2d(xin.permute(0, 3, 1, 2).double(), w.double(), b.double(), padding=1).permute(0, 2, 3, 1) + res.double()
        else:
            M, N, K = shape
            a = ri((M, K), -2, 2).bfloat16() if exact else rnd("sk.a", (M, K))
            w = (ri((N, K), -1, 1) * (torch.rand((N, K), generator=gen, device=DEV) < 0.02)).bfloat16() if exact else rnd("sk.wd", (N, K), K ** -0.5)
            b = ri((N,), -3, 3) if exact else hash_uniform("sk.bd", (N,), 0.2, DEV)
            res = ri((M, N), -4, 4).bfloat16() if exact else rnd("sk.rd", (M, N))
            f = lambda: hip.gemm(a, w, b, residual=res)
            ref = a.double() @ w.double().t() + b.double() + res.double()
        hip.tune("splitk", 1)
        split = f()
        hip.tune("splitk", 0)
        plain = f()
        hip.tune("splitk", 1)
        return split, plain, ref

    split, plain, ref = run(exact=True)
    assert ref.abs().max() < 256 and torch.equal(ref, ref.round())
    assert torch.equal(split.double(), ref) and torch.equal(plain.double(), ref), "exact-integer case"
    split, plain, ref = run(exact=False)
    check(split, ref, 4e-5, f"split-K {kind} {shape}")
    check(plain, ref, 4e-5, f"unsplit {kind} {shape}")
    again, _, _ = run(exact=False)
    assert torch.equal(split, again), "split-K is not bitwise reproducible"


@pytest.mark.parametrize("kind,shape", [("conv", (48, 32, 640, 0, 640)), ("conv", (48, 32, 1280, 640, 640))])
def test_gemm16_tail_split_equals_one_launch_and_fp64(kind, shape):
    """Tail split (the 32x32 level: 384 tiles for 256 CUs): the rows of the last, half-empty round run as a second launch with the
    reduction split in two.  Small-integer operands: every partial sum is exact, so both schedules must equal the int64 reference BIT
    FOR BIT -- across the row split (image 32 of 48), with a per-image-group bias2 (the time embedding: its row group changes INSIDE
    the tail), a concatenated second source and a residual; random operands: one output ulp against fp64; bitwise reproducible."""
    from mmgt_amd import hip
    from mmgt_amd.packing import pack_conv3x3
    gen = torch.Generator(device=DEV).manual_seed(11)
    ri = lambda shape, lo, hi: torch.randint(lo, hi + 1, shape, generator=gen, device=DEV).float()

    def run(exact):
        if kind == "conv":
            nb, h, c0, c1, cout = shape
            mk = (lambda s, sc: ri(s, -2, 2).bfloat16()) if exact else (lambda s, sc: rnd(f"ts.{s}", s, sc))
            x0 = mk((nb, h, h, c0), 1.0)
            x1 = mk((nb, h, h, c1), 1.0) if c1 else None
            w = (ri((cout, c0 + c1, 3, 3), -1, 1) * (torch.rand((cout, c0 + c1, 3, 3), generator=gen, device=DEV) < 0.02)).bfloat16() if exact \
                else rnd("ts.w", (cout, c0 + c1, 3, 3), (9 * (c0 + c1)) ** -0.5)
            b = ri((cout,), -3, 3) if exact else hash_uniform("ts.b", (cout,), 0.2, DEV)
            b2 = ri((2, cout), -3, 3) if exact else hash_uniform("ts.b2", (2, cout), 0.2, DEV)      # one row per CFG batch entry: 24 images each
            res = mk((nb, h, h, cout), 1.0)
            wp = pack_conv3x3(w)
            f = lambda: hip.conv3x3(x0, wp, b, residual=res, x1=x1, bias2=b2, bias2_rows=(nb // 2) * h * h)
            xin = x0 if x1 is None else torch.cat([x0, x1], 3)
            ref = F.conv2d(xin.permute(0, 3, 1, 2).double(), w.double(), b.double(), padding=1).permute(0, 2, 3, 1) + res.double() + \
                b2.double().repeat_interleave(nb // 2, 0)[:, None, None, :]
        else:
            M, N, K = shape
            a = ri((M, K), -2, 2).bfloat16() if exact else rnd("ts.a", (M, K))
            w = (ri((N, K), -1, 1) * (torch.rand((N, K), generator=gen, device=DEV) < 0.02)).bfloat16() if exact else rnd("ts.wd", (N, K), K ** -0.5)
            b = ri((N,), -3, 3) if exact else hash_uniform("ts.bd", (N,), 0.2, DEV)
            res = ri((M, N), -4, 4).bfloat16() if exact else rnd("ts.rd", (M, N))
            f = lambda: hip.gemm(a, w, b, residual=res)
            ref = a.double() @ w.double().t() + b.double() + res.double()
        hip.tune("tailsplit", 1)
        split = f()
        hip.tune("tailsplit", 0)
        plain = f()
        hip.tune("tailsplit", 1)
        return split, plain, ref

    split, plain, ref = run(exact=True)
    assert ref.abs().max() < 256 and torch.equal(ref, ref.round())
    assert torch.equal(plain.double(), ref), "one launch, exact-integer case"
    assert torch.equal(split.double(), ref), "tail split, exact-integer case"
    split, plain, ref = run(exact=False)
    check(split, ref, 4e-5, f"tail split {kind} {shape}")
    check(plain, ref, 4e-5, f"one launch {kind} {shape}")
    again, _, _ = run(exact=False)
    assert torch.equal(split, again), "the tail split is not bitwise reproducible"
