"""GroupNorm + SiLU + conv3x3 in one launch (csrc/gnconv.hip, the VAE's 128-channel levels) against fp64 with the kernel's rounding points
(normalised activations and the result are bf16, every sum fp32) and against the two launches it replaces."""
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def dev():
    return torch.device("cuda:0")


def _case(nb, H, W, seed, res, bias=True, cin=128, cout=128):
    g = torch.Generator(device="cpu").manual_seed(seed)
    x = (torch.randn((nb, H, W, cin), generator=g) * 1.5 + 0.3).to(dev()).bfloat16()
    w = (torch.randn((cout, cin, 3, 3), generator=g) / math.sqrt(9 * cin)).to(dev())
    b = (torch.rand(cout, generator=g) - 0.5).to(dev()) if bias else None
    r = torch.randn((nb, H, W, cout), generator=g).to(dev()).bfloat16() if res else None
    scale = (0.5 + torch.rand((nb, cin), generator=g)).to(dev())
    shift = (torch.rand((nb, cin), generator=g) - 0.5).to(dev())
    return x, w, b, r, scale, shift


def _ref(x, w, b, r, scale, shift):
    """fp64 conv of the bf16-rounded activations (the fused kernel's rounding point) with the bf16-rounded weights"""
    t = torch.addcmul(shift[:, None, None, :], x.float(), scale[:, None, None, :])
    y = (t / (1 + torch.exp(-t))).bfloat16().double()
    o = F.conv2d(y.permute(0, 3, 1, 2), w.bfloat16().double(), None if b is None else b.double(), padding=1).permute(0, 2, 3, 1)
    if r is not None:
        o = o + r.double()
    return o


@pytest.mark.parametrize("nb,H,W,res", [(1, 16, 16, False), (3, 32, 48, True), (2, 64, 64, False), (5, 48, 16, True), (300, 16, 16, True)])
def test_gnconv_tables_vs_fp64(nb, H, W, res):
    """One tile per image (every halo pixel is padding), several tiles per image (halos cross tile borders), more tiles than workgroups
    (300 images: the weight stream and the halo prefetch run across tile boundaries), with and without residual."""
    from mmgt_amd import hip
    from mmgt_amd.packing import pack_gnconv
    x, w, b, r, scale, shift = _case(nb, H, W, 100 + nb, res)
    out = hip.gn_silu_conv3x3_tables(x, scale, shift, pack_gnconv(w), 128, b, r)
    ref = _ref(x, w, b, r, scale, shift)
    d = (out.double() - ref).abs()
    tol = 2.0 ** -8 * ref.abs() + 6e-3          # bf16 rounding of the result + one-ulp flips of activations (device exp against torch's)
    assert (d <= tol).all(), (d.max().item(), (d / tol).max().item(), (d > tol).sum().item())
    assert d.mean().item() < 2e-3
    out2 = hip.gn_silu_conv3x3_tables(x, scale, shift, pack_gnconv(w), 128, b, r)
    assert torch.equal(out, out2)                                                     # reproducible


@pytest.mark.parametrize("cin,cout,res", [(256, 128, False), (128, 64, False), (256, 256, True), (256, 128, True)])
@pytest.mark.parametrize("nb,H,W", [(1, 16, 16), (3, 32, 48), (300, 16, 16)])
def test_gnconv_other_shapes_vs_fp64(cin, cout, res, nb, H, W):
    """Cin = 256 (two phases per tile: the halves of the input channels take turns in the LDS halo and accumulate into the same registers) and
    Cout = 64 (8 x 1 wave grid), Cout = 256 (two launches on the halves of an output with 256 channels per pixel): the VAE's 256 -> 128 conv,
    conv_out and the 256 x 256 level."""
    from mmgt_amd import hip
    from mmgt_amd.packing import pack_gnconv
    x, w, b, r, scale, shift = _case(nb, H, W, 200 + nb + cin, res, cin=cin, cout=cout)
    out = hip.gn_silu_conv3x3_tables(x, scale, shift, pack_gnconv(w), cout, b, r)
    ref = _ref(x, w, b, r, scale, shift)
    d = (out.double() - ref).abs()
    tol = 2.0 ** -8 * ref.abs() + 6e-3
    assert (d <= tol).all(), (d.max().item(), (d / tol).max().item(), (d > tol).sum().item())
    assert d.mean().item() < 2e-3
    assert torch.equal(out, hip.gn_silu_conv3x3_tables(x, scale, shift, pack_gnconv(w), cout, b, r))


@pytest.mark.parametrize("with_stats", [False, True])
@pytest.mark.parametrize("cin,cout,nb,H,W", [(128, 128, 300, 16, 16), (256, 128, 300, 16, 16), (256, 256, 8, 64, 64), (128, 128, 4, 256, 256)])
def test_gnconv_residual_repeated_runs(cin, cout, nb, H, W, with_stats):
    """The residual paths (epilogue vectors requested under the last MFMAs for Cin = 128; accumulator initialisation for Cin = 256), with and
    without the statistics epilogue, over many tiles per workgroup, ten times into a sentinel-filled output: every run equals the launch
    without residual + the residual added outside to within the one bf16 rounding that differs, and all runs are bitwise equal.
    (This is the test that would have caught the store-data hazard of round 5 -- a 16-byte buffer store with a register soffset whose data
    registers the next instruction rewrites sends wrong values in lanes 12 .. 15 -- which failed one run in three at first and every run at
    256 x 256; tools/check_mfma_overlap.py now scans the device code for the pattern.)"""
    from mmgt_amd import hip
    from mmgt_amd.packing import pack_gnconv
    x, w, b, r, scale, shift = _case(nb, H, W, 300 + cin + cout, True, cin=cin, cout=cout)
    wimg = pack_gnconv(w)
    good = hip.gn_silu_conv3x3_tables(x, scale, shift, wimg, cout, b, None).float() + r.float()
    first = None
    st = torch.empty((nb * (H // 16) * (W // 16), cout // 4, 2), device=dev()) if with_stats else None
    for k in range(10):
        out = torch.full((nb, H, W, cout), 777.0, device=dev(), dtype=torch.bfloat16)
        hip.gn_silu_conv3x3_tables(x, scale, shift, wimg, cout, b, r, out=out, stats=st)
        d = (out.float() - good).abs()
        assert (d <= 2.0 ** -6 * good.abs() + 4e-2).all(), (k, d.max().item(), (d > 0.1).sum().item())
        first = out if first is None else first
        assert torch.equal(out, first), k


@pytest.mark.parametrize("cin,cout,res,nb,H,W", [(128, 128, True, 3, 64, 48), (256, 128, False, 2, 32, 32), (256, 256, True, 2, 48, 32), (128, 128, False, 300, 16, 16)])
def test_gnconv_statistics_of_the_output(cin, cout, res, nb, H, W):
    """The per-tile (sum, sum of squares) the launch writes beside its output, folded by mmgt_gn_stats_finalize, against the statistics pass
    over the stored tensor (hip.groupnorm_affine) and against fp64 GroupNorm statistics of it; bitwise repeatable; the output itself is the
    launch's without statistics."""
    from mmgt_amd import hip
    from mmgt_amd.packing import pack_gnconv
    x, w, b, r, scale, shift = _case(nb, H, W, 400 + cin + cout, res, cin=cin, cout=cout)
    x = x + 3.0                                                                          # a mean several sigma off zero: E[x^2] - mean^2 in fp32
    wimg = pack_gnconv(w)
    g = torch.Generator(device="cpu").manual_seed(9)
    gamma, beta = (0.5 + torch.rand(cout, generator=g)).to(dev()), (torch.rand(cout, generator=g) - 0.5).to(dev())
    tiles = (H // 16) * (W // 16)
    stats = torch.full((nb * tiles, cout // 4, 2), float("nan"), device=dev())
    out = hip.gn_silu_conv3x3_tables(x, scale, shift, wimg, cout, b, r, stats=stats)
    assert torch.equal(out, hip.gn_silu_conv3x3_tables(x, scale, shift, wimg, cout, b, r))
    assert torch.isfinite(stats).all()
    sc, sh = hip.gn_tables_from_stats(stats, gamma, beta, 32, 1e-6, nb, cout)
    if H * W > 256:                                                                      # (the statistics pass wants more than 256 pixels)
        sc2, sh2 = hip.groupnorm_affine(out.view(nb, H * W, cout), gamma, beta, 32, 1e-6)
        torch.testing.assert_close(sc, sc2, rtol=2e-4, atol=1e-6)
        torch.testing.assert_close(sh, sh2, rtol=2e-4, atol=2e-4)
    o = out.double().view(nb, H * W, 32, cout // 32)
    mean, var = o.mean(dim=(1, 3)), o.var(dim=(1, 3), unbiased=False)
    rstd = (var + 1e-6).rsqrt()
    want_sc = gamma.double().view(1, 32, -1) * rstd[:, :, None]
    want_sh = beta.double().view(1, 32, -1) - mean[:, :, None] * want_sc
    torch.testing.assert_close(sc.double().view(nb, 32, -1), want_sc, rtol=1e-4, atol=1e-6)
    torch.testing.assert_close(sh.double().view(nb, 32, -1), want_sh, rtol=1e-4, atol=1e-4)
    stats2 = torch.empty_like(stats)
    hip.gn_silu_conv3x3_tables(x, scale, shift, wimg, cout, b, r, stats=stats2)
    assert torch.equal(stats, stats2)


def test_gnconv_statistics_of_an_output_whose_mean_is_30_sigma():
    """the epilogue's tallies are UNSHIFTED fp32 sums per 16 x 16 tile and 4-channel quad.  The finalize kernel turns every partial into
    (count, mean, M2) and merges them with Chan's update, so the cancellation E[x^2] - mean^2 is confined to the 1024 values of one partial: at
    |mean| = 30 sigma (a conv bias of 30) the tables from the producing launch agree with fp64 to 1e-4 relative (measured 1e-5; the plain
    E[x^2] - mean^2 over the image was 1e-3), next to the pivot-shifted pass over the tensor (1e-6)."""
    from mmgt_amd import hip
    from mmgt_amd.packing import pack_gnconv
    nb, H, W, cout = 2, 64, 64, 128
    x, w, b, r, scale, shift = _case(nb, H, W, 77, False)
    b = b + 30.0 * _ref(x, w, None, None, scale, shift).std().float()
    g = torch.Generator(device="cpu").manual_seed(10)
    gamma, beta = (0.5 + torch.rand(cout, generator=g)).to(dev()), (torch.rand(cout, generator=g) - 0.5).to(dev())
    stats = torch.empty((nb * (H // 16) * (W // 16), cout // 4, 2), device=dev())
    out = hip.gn_silu_conv3x3_tables(x, scale, shift, pack_gnconv(w), cout, b, None, stats=stats)
    o = out.double().view(nb, H * W, 32, cout // 32)
    mean, var = o.mean(dim=(1, 3)), o.var(dim=(1, 3), unbiased=False)
    assert (mean.abs() / var.sqrt()).min() > 20
    sc, sh = hip.gn_tables_from_stats(stats, gamma, beta, 32, 1e-6, nb, cout)
    sc2, sh2 = hip.groupnorm_affine(out.view(nb, H * W, cout), gamma, beta, 32, 1e-6)
    want_sc = gamma.double().view(1, 32, -1) * (var + 1e-6).rsqrt()[:, :, None]
    e1 = ((sc.double().view(nb, 32, -1) - want_sc) / want_sc).abs().max().item()
    e2 = ((sc2.double().view(nb, 32, -1) - want_sc) / want_sc).abs().max().item()
    # normalised values (x * scale + shift) at the group's typical |x - mean| = sigma: the error a consumer sees, in units of sigma
    xs = mean[:, :, None] + var.sqrt()[:, :, None]
    y1 = xs * sc.double().view(nb, 32, -1) + sh.double().view(nb, 32, -1)
    y0 = xs * want_sc + (beta.double().view(1, 32, -1) - mean[:, :, None] * want_sc)
    ey = ((y1 - y0).abs() / gamma.double().view(1, 32, -1)).max().item()
    print(f"|mean| = {(mean.abs() / var.sqrt()).mean().item():.0f} sigma: relative scale error from the launch's tallies {e1:.2e}, from the shifted pass {e2:.2e}; "
          f"normalised-value error {ey:.2e} sigma")
    assert e1 < 1e-4 and ey < 1e-4 and e2 < 1e-4


def test_gnconv_no_bias_and_zero_padding():
    """shift != 0 makes silu(shift) != 0 at a zero INPUT: the padding must be zeros after the activation.  Constant input: every interior
    pixel sees nine taps, edges six, corners four."""
    from mmgt_amd import hip
    from mmgt_amd.packing import pack_gnconv
    nb, H, W = 2, 32, 32
    x = torch.zeros((nb, H, W, 128), device=dev(), dtype=torch.bfloat16)
    scale = torch.ones((nb, 128), device=dev())
    shift = torch.full((nb, 128), 2.0, device=dev())
    w = torch.zeros((128, 128, 3, 3), device=dev())
    w[:, 0] = 1.0                                                                     # every tap reads input channel 0
    out = hip.gn_silu_conv3x3_tables(x, scale, shift, pack_gnconv(w), 128).float()
    v = torch.tensor(2.0 / (1 + math.exp(-2.0))).bfloat16().float().item()
    cnt = F.conv2d(torch.ones((1, 1, H, W)), torch.ones((1, 1, 3, 3)), padding=1)[0, 0].to(dev())
    ref = (cnt * v)[None, :, :, None].expand(nb, H, W, 128)
    torch.testing.assert_close(out, ref.bfloat16().float(), rtol=2.0 ** -7, atol=0)


def test_gnconv_fused_groupnorm_vs_two_launches():
    """The product call (statistics pass + fused launch) against hip.groupnorm(silu) -> hip.conv3x3 at a VAE shape: both round the
    normalised activations to bf16; the table form x * scale + shift differs from ((x - mean) rstd) gamma + beta in the last fp32 bit."""
    from mmgt_amd import hip
    from mmgt_amd.packing import pack_conv3x3, pack_gnconv
    nb, H, W = 2, 128, 128
    g = torch.Generator(device="cpu").manual_seed(7)
    x = (torch.randn((nb, H, W, 128), generator=g) * 2 + 0.5).to(dev()).bfloat16()
    w = (torch.randn((128, 128, 3, 3), generator=g) / math.sqrt(9 * 128))
    gamma = (0.5 + torch.rand(128, generator=g)).to(dev())
    beta = (torch.rand(128, generator=g) - 0.5).to(dev())
    b = (torch.rand(128, generator=g) - 0.5).to(dev())
    r = torch.randn((nb, H, W, 128), generator=g).to(dev()).bfloat16()
    fused = hip.gn_silu_conv3x3(x, gamma, beta, 32, 1e-6, pack_gnconv(w.to(dev())), 128, b, r)
    y = hip.groupnorm(x.view(nb, H * W, 128), gamma, beta, 32, 1e-6, silu=True).view(nb, H, W, 128)
    two = hip.conv3x3(y, pack_conv3x3(w).to(dev()).bfloat16(), b, residual=r)
    d = (fused.float() - two.float()).abs()
    assert d.max().item() < 4e-2 and d.mean().item() < 1.5e-3, (d.max().item(), d.mean().item())
    # and against fp64 of the module (GroupNorm in fp64 from the bf16 input): the fused path must be no worse than the two launches
    xd = x.double().permute(0, 3, 1, 2)
    ref = F.conv2d(F.silu(F.group_norm(xd, 32, gamma.double(), beta.double(), 1e-6)), w.to(dev()).bfloat16().double(), b.double(), padding=1)
    ref = ref.permute(0, 2, 3, 1) + r.double()
    e_f, e_t = (fused.double() - ref).abs().mean().item(), (two.double() - ref).abs().mean().item()
    assert e_f <= 1.1 * e_t + 1e-5, (e_f, e_t)


def test_gnconv_rejects_unsupported():
    from mmgt_amd import hip
    x = torch.zeros((1, 24, 16, 128), device=dev(), dtype=torch.bfloat16)
    assert not hip.gn_silu_conv3x3_supported(torch.bfloat16, 128, 128, 24, 16)
    assert not hip.gn_silu_conv3x3_supported(torch.float32, 128, 128, 32, 32)
    assert not hip.gn_silu_conv3x3_supported(torch.bfloat16, 512, 256, 32, 32)
    assert not hip.gn_silu_conv3x3_supported(torch.bfloat16, 128, 64, 32, 32, residual=True)
    with pytest.raises(AssertionError):
        hip.gn_silu_conv3x3_tables(x, torch.ones((1, 128), device=dev()), torch.zeros((1, 128), device=dev()),
                                   torch.zeros(18 * 16384, device=dev(), dtype=torch.uint8), 128)
