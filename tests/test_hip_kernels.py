"""Per-kernel parity of libmmgt_hip.so (through the C ABI, via mmgt_amd.hip) against plain fp32 torch math on the same
inputs.  fp32-I/O mode is held to rtol 1e-3 / atol 1e-4 (the north-star tolerance); bf16 mode is compared with the fp32
result of the bf16-rounded inputs at a bf16-output tolerance."""
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from mmgt_amd.synthetic import hash_uniform  # noqa: E402

DT = [torch.float32, torch.bfloat16]


def tol(dt):
    return dict(rtol=1e-3, atol=1e-4) if dt == torch.float32 else dict(rtol=2e-2, atol=2e-2)


def dev():
    return torch.device("cuda:0")


def rnd(name, shape, scale=1.0, dt=torch.float32):
    return hash_uniform(name, shape, scale).to(dev()).to(dt)


def ref_gemm(a, w):
    return a.double() @ w.double().t()


@pytest.mark.parametrize("dt", DT)
@pytest.mark.parametrize("M,N,K", [(16, 320, 320), (200, 320, 640), (128, 128, 64), (333, 960, 320), (2, 1280, 320),
                                   (1536, 640, 768), (4096, 1280, 5120)])
def test_gemm_plain_and_epilogue(dt, M, N, K):
    from mmgt_amd import hip
    a = rnd("a", (M, K), 1.0, dt)
    w = rnd("w", (N, K), 1.0 / math.sqrt(K), dt)
    bias = rnd("b", (N,), 0.5)
    res = rnd("r", (M, N), 1.0, dt)
    rs = rnd("rs", (M,), 1.0) + 1.0
    rows = 7 if M > 7 else 1
    b2 = rnd("b2", ((M + rows - 1) // rows, N), 0.5)
    out = hip.gemm(a, w)
    torch.testing.assert_close(out.double(), ref_gemm(a, w), **tol(dt))
    out = hip.gemm(a, w, bias, residual=res, row_scale=rs, alpha=0.75, bias2=b2, bias2_rows=rows)
    ref = (ref_gemm(a, w) + bias.double() + b2.double().repeat_interleave(rows, 0)[:M]) * rs.double()[:, None] * 0.75 \
        + res.double()
    torch.testing.assert_close(out.double(), ref, **tol(dt))
    out = hip.gemm(a, w, bias, act=hip.ACT_SILU)
    torch.testing.assert_close(out.double(), F.silu(ref_gemm(a, w) + bias.double()), **tol(dt))


@pytest.mark.parametrize("dt", DT)
@pytest.mark.parametrize("cfg", [0, 16, 17])
@pytest.mark.parametrize("M,N,K", [(300, 320, 320), (4100, 640, 640), (77, 1280, 1280)])
def test_gemm_post_scale_bias(dt, M, N, K, cfg):
    """mmgt_gemm_post: (a W^T + bias) * row_scale * alpha + bias_post + residual, on a strided A view (MM-HAA branch); the
    dispatcher's choice and the two gemm16 tiles (bf16) forced."""
    from mmgt_amd import hip
    big = rnd("a3", (M, 3 * K), 1.0, dt)
    a = big[:, K:2 * K]
    w = rnd("w", (N, K), 1.0 / math.sqrt(K), dt)
    b, bp = rnd("b", (N,), 0.5), rnd("bp", (N,), 0.5)
    rs = rnd("rs", (M,), 1.0) + 1.0
    res = rnd("r", (M, N), 1.0, dt)
    try:
        hip.tune("gemm_cfg", cfg)
        out = hip.gemm_post(a, w, b, rs, 0.75, bp, res)
    finally:
        hip.tune("gemm_cfg", 0)
    ref = (ref_gemm(a, w) + b.double()) * rs.double()[:, None] * 0.75 + bp.double() + res.double()
    torch.testing.assert_close(out.double(), ref, **tol(dt))


@pytest.mark.parametrize("dt", DT)
def test_gemm_strided_views(dt):
    from mmgt_amd import hip
    big = rnd("big", (100, 960), 1.0, dt)
    a = big[:, 320:640]
    w = rnd("w", (320, 320), 0.05, dt)
    outbig = torch.zeros((100, 640), device=dev(), dtype=dt)
    hip.gemm(a, w, out=outbig[:, 320:])
    torch.testing.assert_close(outbig[:, 320:].double(), ref_gemm(a, w), **tol(dt))
    assert outbig[:, :320].abs().max() == 0


@pytest.mark.parametrize("dt", DT)
def test_gemm_geglu(dt):
    from mmgt_amd import hip
    from mmgt_amd.packing import pack_geglu
    M, C = 150, 320
    a = rnd("a", (M, C), 1.0, dt)
    w = rnd("w", (8 * C, C), 1.0 / math.sqrt(C), dt)
    b = rnd("b", (8 * C,), 0.5)
    wp, bp = pack_geglu(w, b)
    res = rnd("res", (M, 4 * C), 1.0, dt)
    out = hip.gemm(a, wp, bp, act=hip.ACT_GEGLU)
    h, g = (ref_gemm(a, w) + b.double()).chunk(2, dim=-1)
    torch.testing.assert_close(out.double(), h * F.gelu(g), **tol(dt))


@pytest.mark.parametrize("dt", DT)
def test_gemm_batched_vt(dt):
    from mmgt_amd import hip
    B, ntok, C, inner = 3, 100, 320, 320
    x = rnd("x", (B, ntok, C), 1.0, dt)
    w = rnd("w", (inner, C), 0.05, dt)
    ld = 104
    out = torch.zeros((B, inner, ld), device=dev(), dtype=dt)
    hip.gemm_batched_wx(w, x, out=out)
    ref = torch.einsum("rc,bnc->brn", w.double(), x.double())
    torch.testing.assert_close(out[:, :, :ntok].double(), ref, **tol(dt))
    assert out[:, :, ntok:].abs().max() == 0


@pytest.mark.parametrize("dt", DT)
@pytest.mark.parametrize("cfg", [1, 3, 6, 9, 12, 16, 17])
def test_gemm_and_conv_forced_tile_configs(dt, cfg):
    """Every tile configuration the heuristic can pick (128x128, 128x64, 256x128, 256x256 with 64-byte LDS rows, and the
    320-column tiles with the barrier inside the chunk) against fp64 torch math, incl. ragged M/N, two bias2 rows per
    tile, residual, a two-source strided conv and persistent workgroups that walk several tiles."""
    from mmgt_amd import hip
    from mmgt_amd.packing import pack_conv3x3
    try:
        hip.tune("gemm_cfg", cfg)
        for M, N, K in [(700, 640, 192), (1300, 320, 1280), (515, 968, 64)]:
            a = rnd("a", (M, K), 1.0, dt)
            w = rnd("w", (N, K), 1.0 / math.sqrt(K), dt)
            bias = rnd("b", (N,), 0.5)
            res = rnd("r", (M, N), 1.0, dt)
            b2 = rnd("b2", ((M + 299) // 300, N), 0.5)
            out = hip.gemm(a, w, bias, residual=res, bias2=b2, bias2_rows=300)
            ref = ref_gemm(a, w) + bias.double() + b2.double().repeat_interleave(300, 0)[:M] + res.double()
            torch.testing.assert_close(out.double(), ref, **tol(dt))
        x0 = rnd("x0", (5, 64, 12, 12), 1.0, dt)
        x1 = rnd("x1", (5, 128, 12, 12), 1.0, dt)
        w = rnd("wc", (320, 192, 3, 3), 1.0 / math.sqrt(9 * 192), dt)
        b = rnd("bc", (320,), 0.5)
        ref = F.conv2d(torch.cat([x0, x1], 1).double(), w.double(), b.double(), stride=2, padding=1)
        out = hip.conv3x3(_nhwc(x0), pack_conv3x3(w), b, x1=_nhwc(x1), stride=2)
        torch.testing.assert_close(out.double(), _nhwc(ref), **tol(dt))
        big = rnd("xb", (16, 64, 32, 32), 1.0, dt)                 # 16384 output pixels: several tiles per workgroup
        wb = rnd("wb", (640, 64, 3, 3), 1.0 / math.sqrt(9 * 64), dt)
        refb = F.conv2d(big.double(), wb.double(), None, padding=1)
        outb = hip.conv3x3(_nhwc(big), pack_conv3x3(wb), None)
        torch.testing.assert_close(outb.double(), _nhwc(refb), **tol(dt))
    finally:
        hip.tune("gemm_cfg", 0)


def _nhwc(x):
    return x.permute(0, 2, 3, 1).contiguous()


@pytest.mark.parametrize("cfg", [16, 17, 19, 20])
def test_gemm16_persistent_workgroups_exact(cfg):
    """Several tiles per persistent workgroup (more tiles than CUs): the tile-boundary machinery of gemm16 -- bias vectors
    fetched one tile ahead into LDS, the next W chunk issued in front of the epilogue stores with a counted wait that leaves
    those stores in flight, group 1's deferred barrier, residual prefetch -- on exact small-integer problems (sparse -1/0/1
    operands, integer bias / per-batch bias / residual): any stale stage, late bias copy or mixed-up tile shows as a wrong
    integer.  K = 64 is the one-chunk tile (no second chunk to cover the relaxed wait), M is ragged."""
    from mmgt_amd import hip
    from mmgt_amd.packing import pack_conv3x3
    dt = torch.bfloat16
    g = torch.Generator(device="cpu").manual_seed(1234 + cfg)

    def sparse(*shape):
        v = torch.randint(-1, 2, shape, generator=g) * (torch.rand(shape, generator=g) < 0.5)
        return v.to(dev()).to(dt)

    def ints(*shape, dtype=torch.float32):
        return torch.randint(-3, 4, shape, generator=g).to(dev()).to(dtype)
    try:
        hip.tune("gemm_cfg", cfg)
        for M, N, K, use_b2 in [(66000, 1280, 320, False), (140100, 320, 64, False), (70000, 640, 128, True), (66000, 960, 320, False)]:
            a, w = sparse(M, K), sparse(N, K)
            bias = ints(N) if N != 960 else None
            res = ints(M, N, dtype=dt) if N <= 640 else None
            b2 = ints((M + 4095) // 4096, N) if use_b2 else None
            ref = a.float() @ w.float().t()
            if bias is not None:
                ref += bias
            if b2 is not None:
                ref += b2.repeat_interleave(4096, 0)[:M]
            if res is not None:
                ref += res.float()
            assert ref.abs().max() < 256
            for rep in range(2):
                out = hip.gemm(a, w, bias, residual=res, bias2=b2, bias2_rows=4096 if use_b2 else 0)
                bad = (out.float() != ref)
                assert not bad.any(), (cfg, M, N, K, rep, int(bad.sum()), bad.nonzero()[:4].tolist())
        for M, N, K in [(70000, 1280 if cfg == 16 else 640, 320), (140100, 320, 128)]:     # row scale + post-scale bias (MM-HAA form)
            a, w = sparse(M, K), sparse(N, K)
            bias, bp, res = ints(N), ints(N), ints(M, N, dtype=dt)
            rs = (2 * torch.randint(1, 3, (M,), generator=g)).float().to(dev())          # x alpha 0.5 = 1 or 2
            ref = (a.float() @ w.float().t() + bias) * rs[:, None] * 0.5 + bp + res.float()
            assert ref.abs().max() < 256 and torch.equal(ref, ref.round())
            out = hip.gemm_post(a, w, bias, rs, 0.5, bp, res)
            bad = out.float() != ref
            assert not bad.any(), (cfg, "post", M, N, K, int(bad.sum()), bad.nonzero()[:4].tolist())
        nb, cin, cout = 20, 64, 640 if cfg == 16 else 320                      # 81920 output pixels: 320 row tiles
        x = sparse(nb, 64, 64, cin)
        wc = (torch.randint(-1, 2, (cout, cin, 3, 3), generator=g) * (torch.rand((cout, cin, 3, 3), generator=g) < 0.5)).float()
        bias, res = ints(cout), ints(nb, 64, 64, cout, dtype=dt)
        ref = F.conv2d(x.float().permute(0, 3, 1, 2), wc.to(dev()), bias, padding=1).permute(0, 2, 3, 1) + res.float()
        assert ref.abs().max() < 256
        out = hip.conv3x3(x, pack_conv3x3(wc).to(dev()).to(dt), bias, residual=res)
        assert torch.equal(out.float(), ref), (cfg, "conv", int((out.float() != ref).sum()))
    finally:
        hip.tune("gemm_cfg", 0)


@pytest.mark.parametrize("cfg", [16, 17, 19, 20])
def test_gemm16_core_exact_integers_and_geglu(cfg):
    """The 16x16x32 ping-pong core (cfg 16: 256 x 256 tile, cfg 17: 256 x 320, cfg 19: 192 x 320, cfg 20: 256 x 128; bf16 only; GEGLU runs on cfg 16 only): (a) exact small-integer operands with an ASYMMETRIC weight matrix --
    any row/column or k-order mix-up in the fragment maps, the permlane16 epilogue or the swizzle shows as a wrong integer;
    (b) GEGLU with packed weights, (c) ragged M / N edges and several tiles per persistent workgroup, (d) bias2 + residual."""
    from mmgt_amd import hip
    from mmgt_amd.packing import pack_geglu
    dt = torch.bfloat16
    try:
        hip.tune("gemm_cfg", cfg)
        for M, N, K in [(256, 256, 64), (300, 264, 128), (2000, 1288, 320), (1000, 320, 192), (520, 640, 64)]:
            a = torch.randint(-1, 2, (M, K), device=dev()).to(dt)
            w = torch.randint(-1, 2, (N, K), device=dev()).to(dt)
            w[:, 0] = (torch.arange(N, device=dev()) % 5 - 2).to(dt)               # column- and row-dependent pattern
            a[:, 1] = (torch.arange(M, device=dev()) % 7 - 3).to(dt)
            out = hip.gemm(a, w)
            ref = a.double() @ w.double().t()
            assert ref.abs().max() < 256                                            # exactly representable in bf16
            assert torch.equal(out.double(), ref), (M, N, K, (out.double() - ref).abs().max().item())
        M, C = 777, 320
        x = rnd("g16.x", (M, C), 1.0, dt)
        w1 = rnd("g16.w1", (8 * C, C), 1.0 / math.sqrt(C), dt)
        b1 = rnd("g16.b1", (8 * C,), 0.3)
        wp, bp = pack_geglu(w1.float().cpu(), b1.cpu())
        out = hip.gemm(x, wp.to(dev()).to(dt), bp.to(dev()), act=hip.ACT_GEGLU)
        y = x.double() @ w1.double().t() + b1.double()
        h, g = y.chunk(2, dim=-1)
        torch.testing.assert_close(out.double(), h * F.gelu(g), **tol(dt))
        M, N, K = 5000, 1280, 640
        a, w = rnd("g16.a", (M, K), 1.0, dt), rnd("g16.w", (N, K), 1.0 / math.sqrt(K), dt)
        bias, res = rnd("g16.b", (N,), 0.5), rnd("g16.r", (M, N), 1.0, dt)
        b2 = rnd("g16.b2", ((M + 2047) // 2048, N), 0.5)
        out = hip.gemm(a, w, bias, residual=res, bias2=b2, bias2_rows=2048)
        ref = ref_gemm(a, w) + bias.double() + b2.double().repeat_interleave(2048, 0)[:M] + res.double()
        torch.testing.assert_close(out.double(), ref, **tol(dt))
    finally:
        hip.tune("gemm_cfg", 0)


@pytest.mark.parametrize("dt", DT)
@pytest.mark.parametrize("cin,cout,hw,stride,up", [(64, 320, 8, 1, False), (320, 320, 8, 2, False),
                                                   (640, 640, 4, 1, True), (320, 64, 5, 1, False),
                                                   (1280, 1280, 2, 1, False), (64, 128, 1, 1, False)])
def test_conv3x3(dt, cin, cout, hw, stride, up):
    from mmgt_amd import hip
    from mmgt_amd.packing import pack_conv3x3
    nb = 6
    x = rnd("x", (nb, cin, hw, hw), 1.0, dt)
    w = rnd("w", (cout, cin, 3, 3), 1.0 / math.sqrt(9 * cin), dt)
    b = rnd("b", (cout,), 0.5)
    xin = F.interpolate(x.double(), scale_factor=2.0, mode="nearest") if up else x.double()
    ref = F.conv2d(xin, w.double(), b.double(), stride=stride, padding=1)
    out = hip.conv3x3(_nhwc(x), pack_conv3x3(w), b, stride=stride, upsample=up)
    torch.testing.assert_close(out.double(), _nhwc(ref), **tol(dt))
    # time-embedding add (one row per 3 images) + residual
    b2 = rnd("b2", (2, cout), 0.5)
    res = rnd("res", tuple(ref.shape), 1.0, dt)
    out = hip.conv3x3(_nhwc(x), pack_conv3x3(w), b, stride=stride, upsample=up, bias2=b2,
                      bias2_rows=3 * ref.shape[2] * ref.shape[3], residual=_nhwc(res))
    ref2 = ref + b2.double().repeat_interleave(3, 0)[:, :, None, None] + res.double()
    torch.testing.assert_close(out.double(), _nhwc(ref2), **tol(dt))


@pytest.mark.parametrize("nb,cin,cout,h,w_", [(3, 1280, 1280, 8, 8), (2, 640, 640, 16, 12), (5, 320, 640, 7, 9), (48, 1280, 1280, 8, 8), (2, 512, 512, 32, 32),
                                              (1, 64, 256, 1, 1)])
def test_conv3x3_upsample_as_four_phase_convs(nb, cin, cout, h, w_):
    """The conv behind a nearest 2x upsampling (resnet.py:31-77) as four 2 x 2 convs on the stored image (`upsample=2`, packing.pack_conv3x3_up2:
    the taps that fall on one stored pixel summed per output phase): the SAME function as upsample -> conv3x3 -- against torch in fp64 on the
    bf16-rounded operands, and against the 3 x 3 kernel on the upsampled view (`upsample=True`), whose error it must not exceed by more than the
    rounding of the summed weights; odd image sizes (every border pixel's padding), both gemm16 tile widths, the in-step 8 x 8 shape."""
    from mmgt_amd import hip
    from mmgt_amd.packing import pack_conv3x3, pack_conv3x3_up2
    dt = torch.bfloat16
    x = rnd("up2.x", (nb, cin, h, w_), 1.0, dt)
    w = rnd("up2.w", (cout, cin, 3, 3), 1.0 / math.sqrt(9 * cin), dt)
    b = rnd("up2.b", (cout,), 0.5)
    ref = F.conv2d(F.interpolate(x.double(), scale_factor=2.0, mode="nearest"), w.double(), b.double(), padding=1)
    nine = hip.conv3x3(_nhwc(x), pack_conv3x3(w), b, upsample=True)
    four = hip.conv3x3(_nhwc(x), pack_conv3x3_up2(w), b, upsample=2)
    assert four.shape == nine.shape == (nb, 2 * h, 2 * w_, cout)
    torch.testing.assert_close(four.double(), _nhwc(ref), **tol(dt))
    e4, e9 = (four.double() - _nhwc(ref)).abs().mean().item(), (nine.double() - _nhwc(ref)).abs().mean().item()
    assert e4 <= 1.6 * e9 + 1e-6, (e4, e9)       # (measured 1.4 x: a summed weight is rounded to bf16 once more, at up to four times a single weight's magnitude)
    assert torch.equal(four, hip.conv3x3(_nhwc(x), pack_conv3x3_up2(w), b, upsample=2))
    # exact-integer operands: every product and sum is exact in bf16 x bf16 -> fp32, so the two forms must agree bit for bit
    xi = torch.randint(-2, 3, (nb, h, w_, cin), device=dev()).to(dt)
    wi = torch.randint(-1, 2, (cout, cin, 3, 3), device=dev()).to(dt)
    assert torch.equal(hip.conv3x3(xi, pack_conv3x3_up2(wi), None, upsample=2), hip.conv3x3(xi, pack_conv3x3(wi), None, upsample=True))


@pytest.mark.parametrize("nb,h,w_", [(3, 16, 8), (2, 64, 64), (5, 8, 16), (1, 16, 24)])
def test_conv_out_as_gemm_over_the_taps_and_gather(nb, h, w_):
    """conv_norm_out -> SiLU -> conv_out (320 -> 4 channels, unet_3d.py:608-620) as ONE 36-column GEMM over every pixel's channels (all nine taps;
    GroupNorm tables + SiLU applied in `rowgemm320`'s prologue) + `conv_taps_gather` (sum of the neighbours' products, zero padding, bias) against
    torch in fp64 and against GroupNorm pass + implicit-GEMM conv (padded to 64 columns) on the same operands; channels 4 .. 7 are zero."""
    from mmgt_amd import hip
    from mmgt_amd.packing import pack_conv3x3, pack_conv_taps, pad_rows
    dt = torch.bfloat16
    x = rnd("co.x", (nb, h, w_, 320), 1.5, dt) + rnd("co.m", (320,), 1.0).to(dt)
    g, b = rnd("co.g", (320,), 0.2) + 1.0, rnd("co.b", (320,), 0.2)
    w = rnd("co.w", (4, 320, 3, 3), 1.0 / math.sqrt(9 * 320), dt)
    bias = rnd("co.bias", (4,), 0.5)
    xn = F.silu(F.group_norm(x.double().permute(0, 3, 1, 2), 32, g.double(), b.double(), 1e-5))
    ref = F.conv2d(xn, w.double(), bias.double(), padding=1).permute(0, 2, 3, 1)
    sc, sh = hip.groupnorm_affine(x.view(nb, h * w_, 320), g, b, 32, 1e-5)
    y, _ = hip.rowgemm320(x.view(nb * h * w_, 320), pack_conv_taps(w), 64, pre_scale=sc, pre_shift=sh, pre_rows=h * w_, pre_silu=True)
    out = hip.conv_taps_gather(y, bias, nb, h, w_)
    assert out.shape == (nb, h, w_, 8) and (out[..., 4:] == 0).all()
    torch.testing.assert_close(out[..., :4].double(), ref, **tol(dt))
    old = hip.conv3x3(hip.groupnorm(x.view(nb, h * w_, 320), g, b, 32, 1e-5, silu=True).view(nb, h, w_, 320), pack_conv3x3(w, None, 64),
                      pad_rows(bias, 64))[..., :4]
    e_new, e_old = (out[..., :4].double() - ref).abs().mean().item(), (old.double() - ref).abs().mean().item()
    assert e_new <= 2.0 * e_old + 1e-6, (e_new, e_old)       # (nine products rounded to bf16 before their sum; the implicit GEMM rounds once)
    y2, _ = hip.rowgemm320(x.view(nb * h * w_, 320), pack_conv_taps(w), 64, pre_scale=sc, pre_shift=sh, pre_rows=h * w_, pre_silu=True)
    assert torch.equal(out, hip.conv_taps_gather(y2, bias, nb, h, w_))


@pytest.mark.parametrize("rows,c0,c1,cout,res", [(4096 * 3 + 37, 640, 320, 320, False), (5000, 320, 320, 320, True), (3072, 1280, 1280, 1280, False),
                                                 (12288, 1280, 640, 1280, True), (777, 64, 128, 256, True), (49152, 640, 640, 640, False)])
def test_conv1x1_over_two_sources(rows, c0, c1, cout, res):
    """A resnet's conv_shortcut over the skip concatenation (resnet.py:243-245) as ONE launch -- gemm16's conv gather over two sources with a single
    tap (`conv1x1_cat`) -- against fp64 and against the two dense GEMMs chained through a residual that it replaces; ragged row counts, split-K-sized
    problems (3072 rows, K = 2560), with and without a residual; exact-integer operands must agree bit for bit."""
    from mmgt_amd import hip
    dt = torch.bfloat16
    x0, x1 = rnd("sc.x0", (rows, c0), 1.0, dt), rnd("sc.x1", (rows, c1), 1.0, dt)
    w = rnd("sc.w", (cout, c0 + c1), (c0 + c1) ** -0.5, dt)
    b = rnd("sc.b", (cout,), 0.5)
    r = rnd("sc.r", (rows, cout), 1.0, dt) if res else None
    ref = torch.cat([x0, x1], 1).double() @ w.double().t() + b.double() + (r.double() if res else 0)
    out = hip.conv1x1_cat(x0, x1, w, b, residual=r)
    torch.testing.assert_close(out.double(), ref, **tol(dt))
    two = hip.gemm(x1, w[:, c0:].contiguous(), None, residual=hip.gemm(x0, w[:, :c0].contiguous(), b))
    if res:
        two = (two.float() + r.float()).to(dt)
    e1, e2 = (out.double() - ref).abs().mean().item(), (two.double() - ref).abs().mean().item()
    assert e1 <= 1.05 * e2 + 1e-6, (e1, e2)          # (one rounding instead of two)
    assert torch.equal(out, hip.conv1x1_cat(x0, x1, w, b, residual=r))
    xi0, xi1 = torch.randint(-2, 3, (rows, c0), device=dev()).to(dt), torch.randint(-2, 3, (rows, c1), device=dev()).to(dt)
    wi = torch.randint(-1, 2, (cout, c0 + c1), device=dev()).to(dt)
    exact = (torch.cat([xi0, xi1], 1).float() @ wi.float().t()).to(dt)
    assert torch.equal(hip.conv1x1_cat(xi0, xi1, wi), exact)


@pytest.mark.parametrize("rows,C,inner", [(4096 + 64, 640, 2560), (3072, 1280, 5120)])
def test_ff2_and_proj_out_as_one_two_source_gemm(rows, C, inner):
    """A transformer block's tail, proj_out(hid + b2 + W2 g) + x_res (transformer_3d.py:262-268), as ONE GEMM over the two sources g and hid with the
    host-multiplied weight [W_po W2 | W_po] and bias b_po + W_po b2 (unet3d.py `_ffpo_weights`) against fp64 and against the two chained GEMMs
    (whose intermediate is rounded to bf16 once more)."""
    from mmgt_amd import hip
    dt = torch.bfloat16
    g = rnd("fp.g", (rows, inner), 1.0, dt)
    hid = rnd("fp.h", (rows, C), 1.0, dt)
    xres = rnd("fp.x", (rows, C), 1.0, dt)
    w2, b2 = rnd("fp.w2", (C, inner), inner ** -0.5, dt), rnd("fp.b2", (C,), 0.3)
    wpo, bpo = rnd("fp.wpo", (C, C), C ** -0.5, dt), rnd("fp.bpo", (C,), 0.3)
    ref = (hid.double() + b2.double() + g.double() @ w2.double().t()) @ wpo.double().t() + bpo.double() + xres.double()
    wpf = wpo.float()
    wcat = torch.cat([wpf @ w2.float(), wpf], 1).to(dt).contiguous()
    bcat = (bpo + wpf @ b2).contiguous()
    one = hip.conv1x1_cat(g, hid, wcat, bcat, residual=xres)
    two = hip.gemm(hip.gemm(g, w2, b2, residual=hid), wpo, bpo, residual=xres)
    torch.testing.assert_close(one.double(), ref, **tol(dt))
    e1, e2 = (one.double() - ref).abs().mean().item(), (two.double() - ref).abs().mean().item()
    assert e1 <= 1.25 * e2 + 1e-6, (e1, e2)


@pytest.mark.parametrize("dt", DT)
def test_conv3x3_two_sources(dt):
    from mmgt_amd import hip
    from mmgt_amd.packing import pack_conv3x3
    x0 = rnd("x0", (4, 320, 6, 6), 1.0, dt)
    x1 = rnd("x1", (4, 640, 6, 6), 1.0, dt)
    w = rnd("w", (320, 960, 3, 3), 0.01, dt)
    ref = F.conv2d(torch.cat([x0, x1], 1).double(), w.double(), None, padding=1)
    out = hip.conv3x3(_nhwc(x0), pack_conv3x3(w), None, x1=_nhwc(x1))
    torch.testing.assert_close(out.double(), _nhwc(ref), **tol(dt))


@pytest.mark.parametrize("dt", DT)
@pytest.mark.parametrize("c0,c1,hw,silu", [(320, 0, 64, True), (640, 320, 16, True), (1280, 1280, 4, False),
                                           (1280, 0, 1, True), (128, 0, 4096, True), (320, 0, 300, False)])
def test_groupnorm(dt, c0, c1, hw, silu):
    from mmgt_amd import hip
    nb = 3
    x0 = rnd("x0", (nb, hw, c0), 1.5, dt) + 0.3
    x1 = rnd("x1", (nb, hw, c1), 0.7, dt) if c1 else None
    g = rnd("g", (c0 + c1,), 0.2) + 1.0
    b = rnd("b", (c0 + c1,), 0.2)
    xx = torch.cat([x0, x1], 2) if c1 else x0
    ref = F.group_norm(xx.double().permute(0, 2, 1), 32, g.double(), b.double(), 1e-5).permute(0, 2, 1)
    if silu:
        ref = F.silu(ref)
    out = hip.groupnorm(x0, g, b, 32, 1e-5, silu=silu, x1=x1)
    torch.testing.assert_close(out.double(), ref, **tol(dt))


@pytest.mark.parametrize("M,N", [(256 * 11 + 37, 256 * 9), (256 * 8, 256 * 8 + 64), (256 * 21, 256 * 40)])
def test_gemm16_blocked_tile_order_is_a_permutation(M, N):
    """gemm16's tile sequence is row-major in groups of 8 row panels walked column-major (csrc/gemm16.hip `decode`): every tile is still
    computed exactly once by the same arithmetic, so the result is bitwise the row-major one -- ragged last group, ragged last tiles."""
    from mmgt_amd import hip
    K = 320
    a = rnd("pb.a", (M, K), 1.0, torch.bfloat16)
    w = rnd("pb.w", (N, K), K ** -0.5, torch.bfloat16)
    b = rnd("pb.b", (N,), 0.5)
    hip.tune("gemm_cfg", 16)
    try:
        hip.tune("g16_pb", 1)
        row_major = hip.gemm(a, w, b)
        hip.tune("g16_pb", 8)
        blocked = hip.gemm(a, w, b)
        hip.tune("g16_pb", 5)
        odd = hip.gemm(a, w, b)
    finally:
        hip.tune("g16_pb", -1)
        hip.tune("gemm_cfg", 0)
    assert torch.equal(row_major, blocked) and torch.equal(row_major, odd)
    torch.testing.assert_close(row_major.double(), ref_gemm(a, w) + b.double(), **tol(torch.bfloat16))


@pytest.mark.parametrize("dt", DT)
@pytest.mark.parametrize("M,N,K,act", [(1, 1280, 320, 2), (1, 1280, 1280, 0), (1, 20160, 1280, 0), (1, 1282, 320, 0), (1, 7, 1280, 2), (2, 1282, 320, 0), (4, 640, 64, 2)])
def test_gemm_of_a_few_rows(dt, M, N, K, act):
    """One row (the time-embedding MLP, every resnet's time_emb_proj of one timestep) runs on a wave-per-column kernel (csrc/gemm.hip
    gemv_kernel) instead of 128 x 128 tiles; 2 .. 4 rows stay on the tiles (see the dispatcher): against fp64, bias and SiLU, column counts off
    the workgroup size, both element types."""
    from mmgt_amd import hip
    a = rnd("gv.a", (M, K), 1.0, dt)
    w = rnd("gv.w", (N, K), K ** -0.5, dt)
    b = rnd("gv.b", (N,), 0.5)
    ref = ref_gemm(a, w) + b.double()
    if act == 2:
        ref = F.silu(ref)
    out = hip.gemm(a, w, b, act=hip.ACT_SILU if act == 2 else hip.ACT_NONE)
    assert out.shape == (M, N)
    torch.testing.assert_close(out.double(), ref, **tol(dt))
    assert torch.equal(out, hip.gemm(a, w, b, act=hip.ACT_SILU if act == 2 else hip.ACT_NONE))
    nob = hip.gemm(a, w)
    torch.testing.assert_close(nob.double(), ref_gemm(a, w), **tol(dt))


def test_gemm16_start_stagger_changes_no_result():
    """Every second CU's workgroup of a short reduction with a residual epilogue starts late (csrc/gemm16.hip `stg`: the chip is then not in
    the tile-end phase all at once): timing only -- the result is bitwise that of the unstaggered launch, for the shape rule's default too."""
    from mmgt_amd import hip
    M, N, K = 192 * 17 + 5, 640, 640
    a = rnd("stg.a", (M, K), 1.0, torch.bfloat16)
    w = rnd("stg.w", (N, K), K ** -0.5, torch.bfloat16)
    b = rnd("stg.b", (N,), 0.5)
    r = rnd("stg.r", (M, N), 1.0, torch.bfloat16)
    outs = []
    try:
        for v in (0, 10, 40, -1):
            hip.tune("g16_stagger", v)
            outs.append(hip.gemm(a, w, b, residual=r))
    finally:
        hip.tune("g16_stagger", -1)
    assert all(torch.equal(outs[0], o) for o in outs[1:])
    torch.testing.assert_close(outs[0].double(), ref_gemm(a, w) + b.double() + r.double(), **tol(torch.bfloat16))


@pytest.mark.parametrize("dt", DT)
@pytest.mark.parametrize("c,hw,silu", [(128, 16384, True), (256, 4096, True), (512, 4096, False), (256, 1024, True), (128, 4100, True),
                                       (320, 4096, True), (320, 1024, False), (448, 2048, True)])
def test_groupnorm_narrow_rows(dt, c, hw, silu):
    """The VAE decoder's shapes (C / VEC in {16, 32, 64}) and rows of 33..64 vectors that leave one row per wave (C = 320 / 448 in bf16:
    csrc/norm.hip gn_*_narrow_kernel; hw = 4100 and the wide fp32 cases fall back to the general kernels) against torch in fp64, and
    against the general two-pass kernels on the same input."""
    from mmgt_amd import hip
    nb = 2
    x = rnd("nx", (nb, hw, c), 1.5, dt) + rnd("nx.mean", (c,), 2.0).to(dt)
    g = rnd("g", (c,), 0.2) + 1.0
    b = rnd("b", (c,), 0.2)
    ref = F.group_norm(x.double().permute(0, 2, 1), 32, g.double(), b.double(), 1e-6).permute(0, 2, 1)
    if silu:
        ref = F.silu(ref)
    out = hip.groupnorm(x, g, b, 32, 1e-6, silu=silu)
    torch.testing.assert_close(out.double(), ref, **tol(dt))
    assert torch.equal(out, hip.groupnorm(x, g, b, 32, 1e-6, silu=silu))          # fixed-order reductions: bitwise reproducible
    hip.tune("gn_narrow", 0)
    try:
        general = hip.groupnorm(x, g, b, 32, 1e-6, silu=silu)
    finally:
        hip.tune("gn_narrow", 2)
    torch.testing.assert_close(out.double(), general.double(), **tol(dt))


@pytest.mark.parametrize("dt", DT)
@pytest.mark.parametrize("nb,c0,c1,hw,silu", [(8, 640, 0, 1024, False), (3, 640, 640, 1024, True), (8, 320, 0, 1024, True), (8, 1280, 0, 256, False),
                                              (3, 1280, 1280, 256, True), (8, 1280, 0, 64, True), (3, 1280, 1280, 64, True), (5, 640, 0, 900, True),
                                              (8, 640, 0, 64, False), (2, 2560, 0, 16, True), (3, 1280, 640, 1024, True), (8, 640, 320, 1024, True),
                                              (2, 1920, 0, 1000, False)])
def test_groupnorm_slab_in_registers(dt, nb, c0, c1, hw, silu):
    """csrc/norm.hip gn_slab_kernel: a workgroup = one image x 1 / 2 / 4 groups keeps its slab in registers (2 .. 22 vectors per thread: every
    instantiation is hit here -- the 1024-thread one for the 60- / 30-channel groups of the 32 x 32 level's skip concatenations too --, both element
    types, one and two sources, ragged row counts, grids that are and are not a multiple of the 8 XCDs) and reads the tensor once.  Against torch in fp64 (values whose mean is 10 x their spread), against the multi-pass kernels it
    replaces on the same input, bitwise repeatable; the statistics-only entry (`groupnorm_affine`) against the tables of the full op."""
    from mmgt_amd import hip
    C = c0 + c1
    x = rnd("slab.x", (nb, hw, C), 1.3, dt) + (10.0 * rnd("slab.mean", (C,), 1.0)).to(dt)
    x0, x1 = (x[..., :c0].contiguous(), x[..., c0:].contiguous()) if c1 else (x, None)
    g = rnd("g", (C,), 0.2) + 1.0
    b = rnd("b", (C,), 0.2)
    ref = F.group_norm(x.double().permute(0, 2, 1), 32, g.double(), b.double(), 1e-5).permute(0, 2, 1)
    plain = ref
    if silu:
        ref = F.silu(ref)
    n0 = hip.call_count("mmgt_groupnorm_nhwc")
    out = hip.groupnorm(x0, g, b, 32, 1e-5, silu=silu, x1=x1)
    assert hip.call_count("mmgt_groupnorm_nhwc") == n0 + 1
    t = dict(rtol=1e-3, atol=2e-4) if dt == torch.float32 else tol(dt)
    torch.testing.assert_close(out.double(), ref, **t)
    assert torch.equal(out, hip.groupnorm(x0, g, b, 32, 1e-5, silu=silu, x1=x1))
    sc, sh = hip.groupnorm_affine(x0, g, b, 32, 1e-5, x1=x1)
    hip.tune("gn_slab", 0)
    try:
        passes = hip.groupnorm(x0, g, b, 32, 1e-5, silu=silu, x1=x1)
        sc0, sh0 = hip.groupnorm_affine(x0, g, b, 32, 1e-5, x1=x1)
    finally:
        hip.tune("gn_slab", 1)
    torch.testing.assert_close(out.double(), passes.double(), **t)
    torch.testing.assert_close(sc, sc0, rtol=2e-5, atol=1e-6)
    torch.testing.assert_close(sh, sh0, rtol=2e-5, atol=2e-5)
    torch.testing.assert_close((x.double() * sc.double()[:, None] + sh.double()[:, None]), plain, rtol=1e-4, atol=2e-4)


@pytest.mark.parametrize("c0,c1,hw,eps", [(320, 0, 4096, 1e-5), (640, 320, 1024, 1e-5), (512, 0, 4096, 1e-6), (1280, 0, 64, 1e-6),
                                          (128, 0, 8192, 1e-6), (256, 0, 4096, 1e-6)])
def test_groupnorm_mean_much_larger_than_std_fp32(c0, c1, hw, eps):
    """Real SD-1.5 / VAE activations have channels whose |mean| is 30-100x their std: the statistics
    must not lose the variance to cancellation.  Per-channel means up to +-100 with unit-scale noise, a few whole groups
    sitting at mean 100 with std 1, eps down to the VAE's 1e-6; fp32 mode at the north-star tolerance."""
    from mmgt_amd import hip
    nb, C = 2, c0 + c1
    mean_c = rnd("gn.mean", (C,), 30.0)
    mean_c[: C // 32 * 3] = 100.0 + rnd("gn.mean2", (C // 32 * 3,), 0.5)      # three groups: |mean| = 100 std
    x = rnd("gn.x", (nb, hw, C), 1.7) + mean_c
    x0, x1 = (x[..., :c0].contiguous(), x[..., c0:].contiguous()) if c1 else (x, None)
    g = rnd("g", (C,), 0.2) + 1.0
    b = rnd("b", (C,), 0.2)
    ref = F.group_norm(x.double().permute(0, 2, 1), 32, g.double(), b.double(), eps).permute(0, 2, 1)
    out = hip.groupnorm(x0, g, b, 32, eps, silu=False, x1=x1)
    torch.testing.assert_close(out.double(), ref, rtol=1e-3, atol=1e-4)


@pytest.mark.parametrize("C", [320, 1280])
def test_layernorm_mean_much_larger_than_std_fp32(C):
    from mmgt_amd import hip
    rows = 517
    x = rnd("ln.x", (rows, C), 1.0) + 30.0 * rnd("ln.mean", (rows, 1), 3.0)       # row means up to +-90, unit-scale spread
    g = rnd("g", (C,), 0.2) + 1.0
    b = rnd("b", (C,), 0.2)
    ref = F.layer_norm(x.double(), (C,), g.double(), b.double(), 1e-5)
    torch.testing.assert_close(hip.layernorm(x, g, b).double(), ref, rtol=1e-3, atol=1e-4)


@pytest.mark.parametrize("dt", DT)
@pytest.mark.parametrize("C", [320, 640, 768, 1280])
def test_layernorm(dt, C):
    from mmgt_amd import hip
    rows = 4 * 6 * 5 + 3
    x = rnd("x", (rows, C), 2.0, dt) + 0.5
    g = rnd("g", (C,), 0.2) + 1.0
    b = rnd("b", (C,), 0.2)
    ref = F.layer_norm(x.double(), (C,), g.double(), b.double(), 1e-5)
    torch.testing.assert_close(hip.layernorm(x, g, b).double(), ref, **tol(dt))
    pe = rnd("pe", (6, C), 1.0)
    idx = (torch.arange(rows, device=dev()) // 5) % 6
    torch.testing.assert_close(hip.layernorm(x, g, b, pe=pe, pe_div=5, pe_mod=6).double(), ref + pe.double()[idx],
                               **tol(dt))
    # the folded form the motion modules use: beta is the (pe_mod, C) table beta + pe
    table = (b[None, :] + pe).contiguous()
    torch.testing.assert_close(hip.layernorm(x, g, table, pe_div=5, pe_mod=6).double(), ref + pe.double()[idx], **tol(dt))


def _ref_attn(q, k, v, scale):
    # q (B, H, Nq, d), k/v (B, H, Nk, d) doubles
    s = torch.einsum("bhqd,bhkd->bhqk", q, k) * scale
    return torch.einsum("bhqk,bhkd->bhqd", torch.softmax(s, -1), v)


@pytest.mark.parametrize("dt", DT)
@pytest.mark.parametrize("hd,nq,nk2", [(40, 64, 64), (40, 200, 200), (80, 16, 16), (160, 4, 4), (160, 1, 1),
                                       (80, 300, 0), (40, 1024, 1024), (80, 512, 256), (40, 256, 0), (40, 512, 64)])
@pytest.mark.parametrize("vt", [False, True])
def test_attention_spatial_with_bank(dt, hd, nq, nk2, vt):
    """Self-attention over nq tokens; the second half of the batch also attends to a per-CFG-row bank of nk2 keys."""
    from mmgt_amd import hip
    heads, frames = 8, 3
    B = 2 * frames
    inner = heads * hd
    q = rnd("q", (B, nq, inner), 1.0, dt)
    k = rnd("k", (B, nq, inner), 1.0, dt)
    v = rnd("v", (B, nq, inner), 1.0, dt)
    kb = rnd("kb", (2, max(nk2, 1), inner), 1.0, dt)
    vb = rnd("vb", (2, max(nk2, 1), inner), 1.0, dt)
    scale = hd ** -0.5
    split = lambda t: t.double().reshape(t.shape[0], t.shape[1], heads, hd).permute(0, 2, 1, 3)
    refs = []
    for b in range(B):
        kk, vv = k[b:b + 1], v[b:b + 1]
        if nk2 and b >= frames:
            kk = torch.cat([kk, kb[b // frames][None]], 1)
            vv = torch.cat([vv, vb[b // frames][None]], 1)
        refs.append(_ref_attn(split(q[b:b + 1]), split(kk), split(vv), scale))
    ref = torch.cat(refs).permute(0, 2, 1, 3).reshape(B, nq, inner)
    out = torch.empty_like(q)
    kw = {}
    if vt:
        pad = lambda n: (n + 7) // 8 * 8
        vT = torch.full((B, inner, pad(nq)), float("nan"), device=dev(), dtype=dt)
        vT[:, :, :nq] = v.transpose(1, 2)
        vbT = torch.full((2, inner, pad(max(nk2, 1))), float("nan"), device=dev(), dtype=dt)
        vbT[:, :, :max(nk2, 1)] = vb.transpose(1, 2)
        vv, v_str = vT, (vT.stride(0), 0, vT.stride(1))
        v2, v2_str = vbT, (vbT.stride(0), vbT.stride(1))
    else:
        vv, v_str = v, (nq * inner, 0, inner)
        v2, v2_str = vb, (vb.stride(0), inner)
    if nk2:
        kw = dict(k2=kb, v2=v2, k2_str=(kb.stride(0), inner), v2_str=v2_str, k2_bdiv=frames, nk2=nk2,
                  seg2_first_batch=frames)
    hip.attention(q, k, vv, out, batch=B, heads=heads, hd=hd, nq=nq, nk=nq, scale=scale, q_str=(nq * inner, 0, inner),
                  k_str=(nq * inner, 0, inner), v_str=v_str, o_str=(nq * inner, 0, inner), v_transposed=vt, **kw)
    torch.testing.assert_close(out.double(), ref, **tol(dt))


def _set_attn64(a64):
    """a64 = 0: the 32-queries-per-wave kernel of attention.hip; 1: attn64.hip's 64-queries-per-wave kernel (attn64d_kernel); None: default (1)."""
    from mmgt_amd import hip
    hip.tune("attn64", 1 if a64 is None else a64)


@pytest.mark.parametrize("hd,a64", [(40, 1), (40, 0), (80, 1)])
def test_attention_online_softmax_rescale_branch_is_forced(hd, a64):
    """(a64: the 64-queries-per-wave kernel of attn64.hip, the production path at head_dim 40, or off.)
    The rare, data-dependent branch of the online softmax (guide rule 26): one key row per later tile is spiked against one query
    so that query's running maximum jumps mid-sequence (own keys and bank keys); full-tensor check against fp64, bf16, V transposed,
    whole 64-key tiles -- the production configuration of the spatial self-attention."""
    from mmgt_amd import hip
    dt = torch.bfloat16
    heads, B, nq, nk2 = 2, 4, 512, 256
    inner = heads * hd
    q = rnd("q2", (B, nq, inner), 1.0, dt)
    k = rnd("k2", (B, nq, inner), 1.0, dt)
    v = rnd("v2", (B, nq, inner), 1.0, dt)
    kb = rnd("kb2", (2, nk2, inner), 1.0, dt)
    vb = rnd("vb2", (2, nk2, inner), 1.0, dt)
    for tile, qrow in ((1, 5), (3, 77), (6, 300)):          # spike: key 64 * tile + 9 aligned with query qrow, x 6
        k[:, 64 * tile + 9] = (q[:, qrow].float() * 6).to(dt)
    kb[:, 130] = (q[2, 411].float() * 8).to(dt)
    scale = hd ** -0.5
    split = lambda t: t.double().reshape(t.shape[0], t.shape[1], heads, hd).permute(0, 2, 1, 3)
    refs = []
    for b in range(B):
        kk, vv = k[b:b + 1], v[b:b + 1]
        if b >= 2:
            kk = torch.cat([kk, kb[(b - 0) // 2][None]], 1)
            vv = torch.cat([vv, vb[(b - 0) // 2][None]], 1)
        refs.append(_ref_attn(split(q[b:b + 1]), split(kk), split(vv), scale))
    ref = torch.cat(refs).permute(0, 2, 1, 3).reshape(B, nq, inner)
    vT, vbT = v.transpose(1, 2).contiguous(), vb.transpose(1, 2).contiguous()
    out = torch.empty_like(q)
    try:
        _set_attn64(a64)
        hip.attention(q, k, vT, out, batch=B, heads=heads, hd=hd, nq=nq, nk=nq, scale=scale, q_str=(nq * inner, 0, inner),
                      k_str=(nq * inner, 0, inner), v_str=(vT.stride(0), 0, vT.stride(1)), o_str=(nq * inner, 0, inner),
                      v_transposed=True, k2=kb, v2=vbT, k2_str=(kb.stride(0), inner), v2_str=(vbT.stride(0), vbT.stride(1)),
                      k2_bdiv=2, nk2=nk2, seg2_first_batch=2)
    finally:
        _set_attn64(None)
    torch.testing.assert_close(out.double(), ref, **tol(dt))


def test_attention_without_running_maximum_and_its_overflow_guard():
    """attn64d_kernel<FAST> (mmgt_tune "attn_nomax", the default): the softmax reference of a row stays what its first 32 keys set it to.  (1) On
    ordinary rows -- including rows whose maximum jumps mid-sequence by a few octaves -- the result meets the same fp64 gate as the kernel with the
    running maximum.  (2) A key whose score lies 2^130 above the reference overflows p = 2^(s - reference): the workgroup's guard (a denominator
    beyond 2^100) must send it through the tile loop again WITH the running maximum -- its rows are then bitwise those of attn_nomax = 0 --, no
    inf / NaN may reach the output, and workgroups without such a key keep their fast results."""
    from mmgt_amd import hip
    dt = torch.bfloat16
    heads, hd, B, nq, nk2 = 2, 40, 4, 512, 256
    inner = heads * hd
    q = rnd("qf", (B, nq, inner), 1.0, dt)
    k = rnd("kf", (B, nq, inner), 1.0, dt)
    v = rnd("vf", (B, nq, inner), 1.0, dt)
    kb = rnd("kbf", (2, nk2, inner), 1.0, dt)
    vb = rnd("vbf", (2, nk2, inner), 1.0, dt)
    for tile, qrow in ((1, 5), (3, 77), (6, 300)):          # moderate spikes: the reference of these rows ends ~2^15 below their maximum
        k[:, 64 * tile + 9] = (q[:, qrow].float() * 6).to(dt)
    k[1, 64 * 5 + 17] = (q[1, 130].float() * 60).to(dt)     # batch 1, query 130 (workgroup of queries 0 .. 255): 2^(~180) above the reference
    kb[1, 200] = (q[3, 411].float() * 70).to(dt)            # a bank key against batch 3, query 411 (bank row 3 // 2 = 1)
    scale = hd ** -0.5
    split = lambda t: t.double().reshape(t.shape[0], t.shape[1], heads, hd).permute(0, 2, 1, 3)
    refs = []
    for b in range(B):
        kk, vv = k[b:b + 1], v[b:b + 1]
        if b >= 2:
            kk = torch.cat([kk, kb[b // 2][None]], 1)
            vv = torch.cat([vv, vb[b // 2][None]], 1)
        refs.append(_ref_attn(split(q[b:b + 1]), split(kk), split(vv), scale))
    ref = torch.cat(refs).permute(0, 2, 1, 3).reshape(B, nq, inner)
    vT, vbT = v.transpose(1, 2).contiguous(), vb.transpose(1, 2).contiguous()

    def run(nomax):
        out = torch.full_like(q, float("nan"))
        hip.tune("attn_nomax", nomax)
        hip.attention(q, k, vT, out, batch=B, heads=heads, hd=hd, nq=nq, nk=nq, scale=scale, q_str=(nq * inner, 0, inner),
                      k_str=(nq * inner, 0, inner), v_str=(vT.stride(0), 0, vT.stride(1)), o_str=(nq * inner, 0, inner),
                      v_transposed=True, k2=kb, v2=vbT, k2_str=(kb.stride(0), inner), v2_str=(vbT.stride(0), vbT.stride(1)),
                      k2_bdiv=2, nk2=nk2, seg2_first_batch=2)
        return out
    try:
        exact, fast = run(0), run(1)
    finally:
        hip.tune("attn_nomax", 1)
    assert torch.isfinite(fast.float()).all()
    # (nearly one-hot rows of |v| <= 1 values: P and the output are bf16, 2^-9 relative each -- the same gate for both kernels)
    torch.testing.assert_close(exact.double(), ref, rtol=2e-2, atol=3e-2)
    torch.testing.assert_close(fast.double(), ref, rtol=2e-2, atol=3e-2)
    # the two workgroups that overflowed (256 queries x both heads... a workgroup = one (batch, head) pair x 256 queries) went through the guarded pass
    assert torch.equal(fast[1, :256], exact[1, :256]) and torch.equal(fast[3, 256:], exact[3, 256:])
    # ... the others did not: somewhere their rounding differs (another reference), far inside the gate
    assert not torch.equal(fast[0], exact[0])
    assert (fast[0].float() - exact[0].float()).abs().max() <= 2.0 ** -7
    assert torch.equal(run(1), fast)                        # repeatable


@pytest.mark.parametrize("heads,B,nq,nk2", [(2, 4, 512, 256), (8, 2, 256, 0), (3, 4, 256, 64), (8, 8, 1024, 1024)])
def test_attention_head_dim_80_dma_staged(heads, B, nq, nk2):
    """csrc/attn80.hip (head_dim 80, the 32 x 32 level: LDS-DMA staged tiles, eight 32-query waves per workgroup, no running maximum in the fast
    pass) against torch in fp64 and against attention.hip's register-staged kernel (mmgt_tune "attn80" = 0) on the same operands: with and
    without the bank segment, pair counts that are and are not a multiple of the 8 XCDs, rows whose maximum jumps mid-sequence, and a key 2^170
    above a row's reference (the guard re-runs that workgroup with the running maximum; no inf / NaN leaves the kernel); repeatable bitwise."""
    from mmgt_amd import hip
    dt = torch.bfloat16
    hd = 80
    inner = heads * hd
    q = rnd("q80", (B, nq, inner), 1.0, dt)
    k = rnd("k80", (B, nq, inner), 1.0, dt)
    v = rnd("v80", (B, nq, inner), 1.0, dt)
    kb = rnd("kb80", (2, max(nk2, 64), inner), 1.0, dt)
    vb = rnd("vb80", (2, max(nk2, 64), inner), 1.0, dt)
    for tile, qrow in ((1, 5), (2, 77), (3, 200)):
        k[:, 64 * tile + 9] = (q[:, qrow].float() * 4).to(dt)
    k[1, 64 * 3 + 17] = (q[1, 130].float() * 40).to(dt)     # batch 1, query 130: far beyond 2^100 above its reference
    scale = hd ** -0.5
    split = lambda t: t.double().reshape(t.shape[0], t.shape[1], heads, hd).permute(0, 2, 1, 3)
    bdiv2 = max(3 * B // 4, 1)                              # the kernel reads bank row b // k2_bdiv: the third quarter of the batch row 0, the last row 1
    refs = []
    for b in range(B):
        kk, vv = k[b:b + 1], v[b:b + 1]
        if nk2 and b >= B // 2:
            kk = torch.cat([kk, kb[b // bdiv2][None, :nk2]], 1)
            vv = torch.cat([vv, vb[b // bdiv2][None, :nk2]], 1)
        refs.append(_ref_attn(split(q[b:b + 1]), split(kk), split(vv), scale))
    ref = torch.cat(refs).permute(0, 2, 1, 3).reshape(B, nq, inner)
    vT, vbT = v.transpose(1, 2).contiguous(), vb.transpose(1, 2).contiguous()
    kw = dict(k2=kb, v2=vbT, k2_str=(kb.stride(0), inner), v2_str=(vbT.stride(0), vbT.stride(1)), k2_bdiv=bdiv2, nk2=nk2,
              seg2_first_batch=B // 2) if nk2 else {}

    def run(a80):
        out = torch.full_like(q, float("nan"))
        hip.tune("attn80", a80)
        hip.attention(q, k, vT, out, batch=B, heads=heads, hd=hd, nq=nq, nk=nq, scale=scale, q_str=(nq * inner, 0, inner),
                      k_str=(nq * inner, 0, inner), v_str=(vT.stride(0), 0, vT.stride(1)), o_str=(nq * inner, 0, inner), v_transposed=True, **kw)
        return out
    try:
        base, new = run(0), run(1)
        again = run(1)
    finally:
        hip.tune("attn80", 1)
    assert torch.isfinite(new.float()).all()
    torch.testing.assert_close(base.double(), ref, rtol=2e-2, atol=3e-2)
    torch.testing.assert_close(new.double(), ref, rtol=2e-2, atol=3e-2)
    assert torch.equal(new, again)
    err_new, err_base = (new.double() - ref).abs().mean().item(), (base.double() - ref).abs().mean().item()
    assert err_new <= 1.5 * err_base + 1e-5, (err_new, err_base)


@pytest.mark.parametrize("a64", [0, 1])
def test_attention_run_to_run_deterministic(a64):
    """Identical launches give bitwise identical results (a data race between the staging writes and the fragment reads of the
    flash attention kernels would show as a result that changes from launch to launch): spatial shape with bank, head_dim 40,
    the 64-queries-per-wave kernel (a64 = 1) and the 32-query one."""
    from mmgt_amd import hip
    dt = torch.bfloat16
    heads, hd, B, n = 8, 40, 8, 1024
    c = heads * hd
    qk = rnd("det.qk", (B * n, 2 * c), 1.0, dt)
    vt = rnd("det.vt", (B, c, n), 1.0, dt)
    kb, vbt = rnd("det.kb", (2, n, c), 1.0, dt), rnd("det.vbt", (2, c, n), 1.0, dt)
    o = torch.empty((B * n, c), device=dev(), dtype=dt)

    def run():
        hip.attention(qk, qk[:, c:], vt, o, batch=B, heads=heads, hd=hd, nq=n, nk=n, scale=hd ** -0.5, q_str=(n * 2 * c, 0, 2 * c),
                      k_str=(n * 2 * c, 0, 2 * c), v_str=(c * n, 0, n), o_str=(n * c, 0, c), v_transposed=True, k2=kb, v2=vbt,
                      k2_str=(kb.stride(0), kb.stride(1)), v2_str=(vbt.stride(0), vbt.stride(1)), k2_bdiv=B // 2, nk2=n,
                      seg2_first_batch=B // 2)
    try:
        _set_attn64(a64)
        run()
        first = o.clone()
        for _ in range(12):
            o.zero_()
            run()
            assert torch.equal(o, first)
    finally:
        _set_attn64(None)


@pytest.mark.parametrize("dt", DT)
@pytest.mark.parametrize("hd,frames,hw", [(40, 24, 16), (80, 8, 9), (160, 24, 4), (40, 32, 1)])
def test_attention_temporal_layout(dt, hd, frames, hw):
    """Sequences run over the frame axis of a ((b f), hw, C) token tensor: batch = (b, pixel), token stride hw*C."""
    from mmgt_amd import hip
    heads, b = 8, 2
    C = heads * hd
    qkv = rnd("qkv", (b * frames, hw, 3 * C), 1.0, dt)
    q, k, v = qkv[..., :C], qkv[..., C:2 * C], qkv[..., 2 * C:]
    seq = lambda t: t.double().reshape(b, frames, hw, heads, hd).permute(0, 2, 3, 1, 4).reshape(b * hw, heads, frames, hd)
    ref = _ref_attn(seq(q), seq(k), seq(v), hd ** -0.5)
    ref = ref.reshape(b, hw, heads, frames, hd).permute(0, 3, 1, 2, 4).reshape(b * frames, hw, C)
    out = torch.empty((b * frames, hw, C), device=dev(), dtype=dt)
    st = (frames * hw * 3 * C, 3 * C, hw * 3 * C)
    hip.attention(q, k, v, out, batch=b * hw, heads=heads, hd=hd, nq=frames, nk=frames, scale=hd ** -0.5, q_str=st,
                  k_str=st, v_str=st, o_str=(frames * hw * C, C, hw * C), bdiv=hw)
    torch.testing.assert_close(out.double(), ref, **tol(dt))


@pytest.mark.parametrize("nq,nk2,frames", [(256, 128, 2), (512, 256, 3)])
def test_attention_twin_output_is_the_own_key_attention(nq, nk2, frames, a64=1):
    """mmgt_attention_twin: one pass over [own keys | bank] per frame writes the attention over both segments AND, as the state after the
    last own-key tile, the attention over the own keys alone -- bitwise what two separate launches compute (the CFG pair of the first
    reference-attention reader: mmgt_amd/unet3d.py _spatial_transformer_twin)."""
    from mmgt_amd import hip
    dt, hd, heads = torch.bfloat16, 40, 8
    inner = heads * hd
    qk = rnd("tw.qk", (frames, nq, 2 * inner), 1.0, dt)
    v = rnd("tw.v", (frames, nq, inner), 1.0, dt)
    kbank = rnd("tw.kb", (1, nk2, inner), 1.0, dt)
    vbank = rnd("tw.vb", (1, nk2, inner), 1.0, dt)
    vt = v.transpose(1, 2).contiguous()
    v2t = vbank.transpose(1, 2).contiguous()
    common = dict(batch=frames, heads=heads, hd=hd, nq=nq, nk=nq, scale=hd ** -0.5, q_str=(nq * 2 * inner, 0, 2 * inner),
                  k_str=(nq * 2 * inner, 0, 2 * inner), v_str=(inner * nq, 0, nq), o_str=(nq * inner, 0, inner), v_transposed=True)
    seg2 = dict(k2=kbank, v2=v2t, k2_str=(kbank.stride(0), kbank.stride(1)), v2_str=(v2t.stride(0), v2t.stride(1)), k2_bdiv=frames, nk2=nk2,
                seg2_first_batch=0)
    both = torch.empty((frames, nq, inner), device=dev(), dtype=dt)
    twin = torch.empty_like(both)
    ref_both, ref_own = torch.empty_like(both), torch.empty_like(both)
    try:
        _set_attn64(a64)
        hip.attention(qk, qk[..., inner:], vt, both, twin_out=twin, **common, **seg2)
        hip.attention(qk, qk[..., inner:], vt, ref_both, **common, **seg2)
        hip.attention(qk, qk[..., inner:], vt, ref_own, **common)
    finally:
        _set_attn64(None)
    assert torch.equal(both, ref_both) and torch.equal(twin, ref_own)
    sp = lambda t, n: t.double().reshape(frames, n, heads, hd).permute(0, 2, 1, 3)
    want = _ref_attn(sp(qk[..., :inner], nq), sp(qk[..., inner:], nq), sp(v, nq), hd ** -0.5).permute(0, 2, 1, 3).reshape(frames, nq, inner)
    torch.testing.assert_close(twin.double(), want, **tol(dt))
    with pytest.raises(RuntimeError, match="attention_twin"):                 # no twin kernel for this shape: an error, not a fallback
        hip.attention(qk[:, :200], qk[:, :200, inner:], vt, both[:, :200], twin_out=twin[:, :200], **dict(common, nq=200), **seg2)


@pytest.mark.parametrize("dt", DT)
def test_attention_cross_audio_24_heads(dt):
    """MM-HAA: three cross-attention branches to 32 audio tokens run as one 24-head problem over fused projections."""
    from mmgt_amd import hip
    hd, nq, bf = 40, 64, 4
    inner = 8 * hd
    q3 = rnd("q3", (bf, nq, 3 * inner), 1.0, dt)
    kv = rnd("kv", (bf, 32, 6 * inner), 1.0, dt)
    out = torch.empty_like(q3)
    hip.attention(q3, kv, kv[..., 3 * inner:], out, batch=bf, heads=24, hd=hd, nq=nq, nk=32, scale=hd ** -0.5,
                  q_str=(nq * 3 * inner, 0, 3 * inner), k_str=(32 * 6 * inner, 0, 6 * inner),
                  v_str=(32 * 6 * inner, 0, 6 * inner), o_str=(nq * 3 * inner, 0, 3 * inner))
    sp = lambda t, n: t.double().reshape(bf, n, 24, hd).permute(0, 2, 1, 3)
    ref = _ref_attn(sp(q3, nq), sp(kv[..., :3 * inner], 32), sp(kv[..., 3 * inner:], 32), hd ** -0.5)
    torch.testing.assert_close(out.double(), ref.permute(0, 2, 1, 3).reshape(bf, nq, 3 * inner), **tol(dt))


@pytest.mark.parametrize("dt", DT)
@pytest.mark.parametrize("nq,bf", [(64, 3), (200, 3), (200, 8), (384, 16)])
def test_attention_scaled_rows_into_a_padded_operand(dt, nq, bf):
    """mmgt_attention_scaled: the three MM-HAA branches write mask_i * attn_i straight into the (rows, 3 inner + pad) operand of the merged
    out-projection (mmgt_amd/unet3d.py _audio_transformer); the pad columns are not touched."""
    from mmgt_amd import hip
    hd = 40
    inner = 8 * hd
    k3, kp = 3 * inner, 3 * inner + 64
    q3 = rnd("sq3", (bf, nq, k3), 1.0, dt)
    kv = rnd("skv", (bf, 32, 2 * k3), 1.0, dt)
    rs = rnd("srs", (3, bf * nq), 0.5) + 0.5
    out = torch.full((bf, nq, kp), 7.0, device=dev(), dtype=dt)
    hip.attention(q3, kv, kv[..., k3:], out, batch=bf, heads=24, hd=hd, nq=nq, nk=32, scale=hd ** -0.5,
                  q_str=(nq * k3, 0, k3), k_str=(32 * 2 * k3, 0, 2 * k3), v_str=(32 * 2 * k3, 0, 2 * k3), o_str=(nq * kp, 0, kp),
                  out_scale=rs, out_scale_heads=8)
    sp = lambda t, n: t.double().reshape(bf, n, 24, hd).permute(0, 2, 1, 3)
    ref = _ref_attn(sp(q3, nq), sp(kv[..., :k3], 32), sp(kv[..., k3:], 32), hd ** -0.5).permute(0, 2, 1, 3)      # (bf, nq, 24, hd)
    ref = ref * rs.double().reshape(3, bf, nq).permute(1, 2, 0).repeat_interleave(8, dim=2)[..., None]
    torch.testing.assert_close(out[..., :k3].double(), ref.reshape(bf, nq, k3), **tol(dt))
    assert (out[..., k3:] == 7.0).all()
    # bf % 8 == 0: the heads-inner workgroup order (all 24 heads of a query block back to back on one XCD) is a permutation of the grid
    hip.tune("attn_heads_inner", 0)
    try:
        plain = torch.full((bf, nq, kp), 7.0, device=dev(), dtype=dt)
        hip.attention(q3, kv, kv[..., k3:], plain, batch=bf, heads=24, hd=hd, nq=nq, nk=32, scale=hd ** -0.5,
                      q_str=(nq * k3, 0, k3), k_str=(32 * 2 * k3, 0, 2 * k3), v_str=(32 * 2 * k3, 0, 2 * k3), o_str=(nq * kp, 0, kp),
                      out_scale=rs, out_scale_heads=8)
    finally:
        hip.tune("attn_heads_inner", 1)
    assert torch.equal(out, plain)


def test_softmax_rows_and_plumbing():
    from mmgt_amd import hip
    for dt in DT:
        x = rnd("sm", (37, 1000), 3.0, dt)
        torch.testing.assert_close(hip.softmax_rows(x, 0.3).double(), torch.softmax(x.double() * 0.3, -1), **tol(dt))
        lat = rnd("lat", (2, 4, 3, 5, 5), 1.0)
        nhwc = hip.ncfhw_to_nhwc(lat, 64, dt)
        assert nhwc.shape == (6, 5, 5, 64) and nhwc[..., 4:].abs().max() == 0
        torch.testing.assert_close(nhwc[..., :4].float(), lat.permute(0, 2, 3, 4, 1).reshape(6, 5, 5, 4).to(dt).float())
        back = hip.nhwc_to_ncfhw(nhwc, 2, 4)
        torch.testing.assert_close(back, lat.to(dt).float())
        ts = torch.tensor([999.0, 3.0], device=dev())
        tf = hip.timestep_features(ts, 320, dt)
        fr = torch.exp(-math.log(10000.0) * torch.arange(160, device=dev()) / 160)
        ref = torch.cat([torch.cos(ts[:, None] * fr), torch.sin(ts[:, None] * fr)], -1)
        torch.testing.assert_close(tf.float(), ref, rtol=1e-3, atol=2e-3 if dt == torch.float32 else 1e-2)
        torch.testing.assert_close(hip.silu(x).double(), F.silu(x.double()), **tol(dt))


def test_cfg_ddim_and_window_accumulate():
    from mmgt_amd import hip
    Fr, C, hw = 6, 4, 3
    lat = rnd("lat", (1, C, Fr, hw, hw), 1.0)
    ps = torch.zeros((2, C, Fr, hw, hw), device=dev())
    cnt = torch.zeros((Fr,), device=dev())
    ref_ps = ps.clone()
    ref_cnt = cnt.clone()
    for wi, win in enumerate([[0, 1, 2, 3], [2, 3, 4, 5], [4, 5, 0, 1]]):
        pred = rnd(f"pred{wi}", (2 * len(win), hw, hw, 64), 1.0, torch.bfloat16)
        idx = torch.tensor(win, device=dev(), dtype=torch.int32)
        hip.accumulate_window(pred, ps, cnt, idx, C)
        p5 = pred[..., :C].float().reshape(2, len(win), hw, hw, C).permute(0, 4, 1, 2, 3)
        ref_ps[:, :, win] += p5
        ref_cnt[win] += 1
    torch.testing.assert_close(ps, ref_ps)
    torch.testing.assert_close(cnt, ref_cnt)
    # the window-parallel form: one CFG row at a time, sliced to the C valid channels (Cpad == C), counter bumped once per window
    ps2, cnt2 = torch.zeros_like(ps), torch.zeros_like(cnt)
    for wi, win in enumerate([[0, 1, 2, 3], [2, 3, 4, 5], [4, 5, 0, 1]]):
        pred = rnd(f"pred{wi}", (2 * len(win), hw, hw, 64), 1.0, torch.bfloat16)
        idx = torch.tensor(win, device=dev(), dtype=torch.int32)
        rows = pred[..., :C].float().contiguous().view(2, len(win), hw, hw, C)
        for row in (0, 1):
            hip.accumulate_window(rows[row].contiguous(), ps2, cnt2, idx, C, rows=1, row0=row, bump_counter=row == 0)
    assert torch.equal(ps2, ps) and torch.equal(cnt2, cnt)
    with pytest.raises(RuntimeError, match="CFG rows"):
        hip.accumulate_window(rows[0].contiguous(), ps2, cnt2, idx, C, rows=1, row0=2)
    g, sa_t, sb_t, sa_p, sb_p = 3.5, 0.6, 0.8, 0.9, math.sqrt(1 - 0.81)
    out = hip.cfg_ddim_step(ps, cnt, lat, g, sa_t, sb_t, sa_p, sb_p)
    eps = ps / cnt[None, None, :, None, None]
    v = eps[0:1] + g * (eps[1:2] - eps[0:1])
    x0 = sa_t * lat - sb_t * v
    e = sa_t * v + sb_t * lat
    torch.testing.assert_close(out, sa_p * x0 + sb_p * e, rtol=1e-5, atol=1e-6)


def test_errors_are_loud():
    from mmgt_amd import hip
    a = rnd("a", (8, 100), 1.0)
    w = rnd("w", (8, 100), 1.0)
    with pytest.raises(RuntimeError, match="multiple of 64"):
        hip.gemm(a, w)
    with pytest.raises(RuntimeError):
        hip.gemm(a.cpu(), w.cpu())
