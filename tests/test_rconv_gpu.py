"""GroupNorm-apply + SiLU + conv3x3 of the UNet's resnets in one launch (csrc/rconv.hip: mmgt_gn_silu_conv3x3_unet) against fp64 with the kernel's
rounding points (normalised activations and the result are bf16, every sum fp32), against the two launches it replaces, with two-source inputs
(the skip concatenation of unet_3d_blocks.py:941-969), the time-embedding rows and the residual of ResnetBlock3D.forward (resnet.py:217-247)."""
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def dev():
    return torch.device("cuda:0")


def _case(nb, H, W, c0, c1, cout, seed, res=False, bias=True, temb_rows=0):
    g = torch.Generator(device="cpu").manual_seed(seed)
    cin = c0 + c1
    x0 = (torch.randn((nb, H, W, c0), generator=g) * 1.5 + 0.3).to(dev()).bfloat16()
    x1 = (torch.randn((nb, H, W, c1), generator=g) * 0.7 - 0.2).to(dev()).bfloat16() if c1 else None
    w = (torch.randn((cout, cin, 3, 3), generator=g) / math.sqrt(9 * cin)).to(dev())
    b = (torch.rand(cout, generator=g) - 0.5).to(dev()) if bias else None
    temb = (torch.rand((temb_rows, cout), generator=g) - 0.5).to(dev()) if temb_rows else None
    r = torch.randn((nb, H, W, cout), generator=g).to(dev()).bfloat16() if res else None
    tab = torch.empty((2, nb, cin), device=dev())
    tab[0] = (0.5 + torch.rand((nb, cin), generator=g)).to(dev())
    tab[1] = (torch.rand((nb, cin), generator=g) - 0.5).to(dev())
    return x0, x1, w, b, temb, r, tab[0], tab[1]


def _ref(x0, x1, w, b, temb, b2_imgs, r, scale, shift):
    """fp64 conv of the bf16-rounded activations (the fused kernel's rounding point) with the bf16-rounded weights"""
    x = x0 if x1 is None else torch.cat([x0, x1], dim=3)
    t = torch.addcmul(shift[:, None, None, :], x.float(), scale[:, None, None, :])
    y = (t / (1 + torch.exp(-t))).bfloat16().double()
    o = F.conv2d(y.permute(0, 3, 1, 2), w.bfloat16().double(), None if b is None else b.double(), padding=1).permute(0, 2, 3, 1)
    if temb is not None:
        rows = torch.arange(x.shape[0], device=x.device) // b2_imgs
        o = o + temb.double()[rows][:, None, None, :]
    if r is not None:
        o = o + r.double()
    return o


def _check(out, ref, cin):
    # one output bf16 ulp + the fp32 accumulation bound of a 9 Cin-long sum of O(1) products
    d = (out.double() - ref).abs()
    tol = ref.abs() * 2.0 ** -8 + 9 * cin * 2.0 ** -22 * 4
    assert (d <= tol).all(), (d.max().item(), (d - tol).max().item())


@pytest.mark.parametrize("nb,H,W,c0,c1,cout,res,temb_rows,b2", [
    (1, 16, 16, 320, 0, 320, False, 0, 0),             # one tile: every halo pixel outside the image is padding
    (3, 32, 48, 320, 0, 320, True, 3, 1),              # edge / corner / interior tiles, a temb row per image, residual
    (2, 64, 64, 640, 320, 320, False, 1, 2),           # two sources (up_blocks.3: 640 | 320), one temb row for both images
    (4, 32, 32, 320, 320, 640, True, 2, 2),            # two output blocks
    (2, 16, 16, 1280, 640, 1280, True, 2, 1),          # 30 phases, four output blocks, the seam inside the walk
    (300, 16, 16, 320, 0, 320, True, 0, 0),            # more units than workgroups: many units per workgroup, epilogue -> next unit hand-over
])
def test_rconv_tables_vs_fp64(nb, H, W, c0, c1, cout, res, temb_rows, b2):
    """... in every cut of the workgroup (mmgt_tune "rconv_cb": blocks of 320 / 256 / 160 output channels = 4 x 2 waves of 10 or 8 column tiles, 8 x 1
    waves of 10) and in the one the dispatcher picks"""
    from mmgt_amd import hip
    from mmgt_amd.packing import pack_rconv
    x0, x1, w, b, temb, r, scale, shift = _case(nb, H, W, c0, c1, cout, 100 + nb + c0 + c1, res=res, temb_rows=temb_rows)
    ref = _ref(x0, x1, w, b, temb, max(b2, 1), r, scale, shift)
    wimg = pack_rconv(w)
    try:
        for cb in (0, 320, 256, 160):
            if cb and cout % cb:
                continue
            hip.tune("rconv_cb", cb)
            out = hip.gn_silu_conv3x3_unet(x0, scale, shift, wimg, cout, b, temb, b2, r, x1=x1)
            torch.cuda.synchronize()
            _check(out, ref, c0 + c1)
    finally:
        hip.tune("rconv_cb", 0)


def test_rconv_routing_closed_form():
    """Any pixel / channel / tap / block mix-up breaks this: input channel c of pixel (y, x) holds a value that identifies (c, y, x) after the
    activation is made the identity-like (scale tiny: silu(t) ~ t / 2), and every output channel reads ONE (tap, input channel)."""
    from mmgt_amd import hip
    from mmgt_amd.packing import pack_rconv
    nb, H, W, c0, c1, cout = 2, 32, 32, 320, 64, 320
    cin = c0 + c1
    g = torch.Generator(device="cpu").manual_seed(5)
    x = torch.randn((nb, H, W, cin), generator=g).to(dev()).bfloat16()
    tab = torch.zeros((2, nb, cin), device=dev())
    tab[0] = 1.0
    w = torch.zeros((cout, cin, 3, 3), device=dev())
    co = torch.arange(cout, device=dev())
    ci = (co * 7 + 3) % cin
    tap = (co * 5 + 1) % 9
    w[co, ci, tap // 3, tap % 3] = 1.0
    out = hip.gn_silu_conv3x3_unet(x[..., :c0].contiguous(), tab[0], tab[1], pack_rconv(w), cout, x1=x[..., c0:].contiguous()).float()
    t = x.float()
    y = (t / (1 + torch.exp(-t))).bfloat16().float()
    yp = F.pad(y.permute(0, 3, 1, 2), (1, 1, 1, 1))
    for c in range(0, cout, 13):
        ky, kx = int(tap[c]) // 3, int(tap[c]) % 3
        want = yp[:, int(ci[c]), ky:ky + H, kx:kx + W]
        assert torch.equal(out[..., c], want), c


def test_rconv_zero_padding_after_the_activation():
    """shift != 0 makes silu(shift) != 0 at a zero INPUT: the padding must be zeros after the activation.  Constant input: every interior
    pixel sees nine taps, edges six, corners four."""
    from mmgt_amd import hip
    from mmgt_amd.packing import pack_rconv
    nb, H, W, cin, cout = 2, 32, 32, 320, 320
    x = torch.zeros((nb, H, W, cin), device=dev(), dtype=torch.bfloat16)
    tab = torch.empty((2, nb, cin), device=dev())
    tab[0] = 1.0
    tab[1] = 2.0
    w = torch.zeros((cout, cin, 3, 3), device=dev())
    w[:, 0] = 1.0                                                                     # every tap reads input channel 0
    out = hip.gn_silu_conv3x3_unet(x, tab[0], tab[1], pack_rconv(w), cout).float()
    v = torch.tensor(2.0 / (1 + math.exp(-2.0))).bfloat16().float().item()
    cnt = F.conv2d(torch.ones((1, 1, H, W)), torch.ones((1, 1, 3, 3)), padding=1)[0, 0].to(dev())
    ref = (cnt * v)[None, :, :, None].expand(nb, H, W, cout)
    torch.testing.assert_close(out, ref.bfloat16().float(), rtol=2.0 ** -7, atol=0)


@pytest.mark.parametrize("c0,c1,cout,H", [(320, 0, 320, 64), (640, 320, 320, 64), (640, 640, 640, 32), (1280, 0, 1280, 16)])
def test_rconv_in_step_shapes_vs_two_launches_and_fp64(c0, c1, cout, H):
    """The in-step shapes (48 images) against fp64 AND against hip.groupnorm(silu) -> hip.conv3x3, the pair it replaces (the pair rounds the
    normalised tensor to bf16 at the same point; the table form x * scale + shift differs from ((x - mean) rstd) gamma + beta in the last
    fp32 bit); ten runs into sentinel-filled outputs are bitwise equal (the store-hazard test of csrc/gnconv.hip)."""
    from mmgt_amd import hip
    from mmgt_amd.packing import pack_conv3x3, pack_rconv
    nb, W = 48, H
    g = torch.Generator(device="cpu").manual_seed(7 + c0 + c1)
    cin = c0 + c1
    x0 = (torch.randn((nb, H, W, c0), generator=g) * 2 + 0.5).to(dev()).bfloat16()
    x1 = (torch.randn((nb, H, W, c1), generator=g) - 0.3).to(dev()).bfloat16() if c1 else None
    w = (torch.randn((cout, cin, 3, 3), generator=g) / math.sqrt(9 * cin))
    gamma, beta = (0.5 + torch.rand(cin, generator=g)).to(dev()), (torch.rand(cin, generator=g) - 0.5).to(dev())
    b = (torch.rand(cout, generator=g) - 0.5).to(dev())
    temb = (torch.rand((2, cout), generator=g) - 0.5).to(dev())
    r = torch.randn((nb, H, W, cout), generator=g).to(dev()).bfloat16()
    scale, shift = hip.groupnorm_affine(x0.view(nb, H * W, c0), gamma, beta, 32, 1e-5, x1=None if x1 is None else x1.view(nb, H * W, c1))
    wimg = pack_rconv(w.to(dev()))
    out = hip.gn_silu_conv3x3_unet(x0, scale, shift, wimg, cout, b, temb, nb // 2, r, x1=x1)
    _check(out, _ref(x0, x1, w.to(dev()), b, temb, nb // 2, r, scale, shift), cin)
    hdn = hip.groupnorm(x0.view(nb, H * W, c0), gamma, beta, 32, 1e-5, silu=True, x1=None if x1 is None else x1.view(nb, H * W, c1))
    two = hip.conv3x3(hdn.view(nb, H, W, cin), pack_conv3x3(w).to(dev()).bfloat16(), b, bias2=temb, bias2_rows=nb // 2 * H * W, residual=r)
    d = (out.float() - two.float()).abs()
    print(f"{c0}+{c1}->{cout} @ {H}: fused vs two launches max {d.max().item():.3e} mean {d.mean().item():.3e}")
    assert d.max() <= 0.08 and d.mean() <= 2e-3
    for _ in range(10):
        o2 = torch.full_like(out, float("nan"))
        hip.gn_silu_conv3x3_unet(x0, scale, shift, wimg, cout, b, temb, nb // 2, r, x1=x1, out=o2)
        assert torch.equal(o2, out)


@pytest.mark.parametrize("HW", [1024, 256, 64])
def test_groupnorm_affine_two_sources(HW):
    """The statistics pass over a channel concatenation whose groups straddle the seam (1280 | 640 in 32 groups of 60) against fp64; 256 and 64
    pixels: the small-image kernel's two passes with tables out."""
    from mmgt_amd import hip
    nb, c0, c1 = 3, 1280, 640
    g = torch.Generator(device="cpu").manual_seed(3)
    x0 = (torch.randn((nb, HW, c0), generator=g) * 1.3 + 4.0).to(dev()).bfloat16()
    x1 = (torch.randn((nb, HW, c1), generator=g) * 0.4 - 2.0).to(dev()).bfloat16()
    gamma, beta = (0.5 + torch.rand(c0 + c1, generator=g)).to(dev()), (torch.rand(c0 + c1, generator=g) - 0.5).to(dev())
    sc, sh = hip.groupnorm_affine(x0, gamma, beta, 32, 1e-5, x1=x1)
    x = torch.cat([x0, x1], dim=2).double().view(nb, HW, 32, -1)
    mean, var = x.mean(dim=(1, 3)), x.var(dim=(1, 3), unbiased=False)
    want_sc = gamma.double().view(1, 32, -1) * (var + 1e-5).rsqrt()[:, :, None]
    want_sh = beta.double().view(1, 32, -1) - mean[:, :, None] * want_sc
    torch.testing.assert_close(sc.double().view(nb, 32, -1), want_sc, rtol=1e-4, atol=1e-6)
    torch.testing.assert_close(sh.double().view(nb, 32, -1), want_sh, rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("nb,H,c0,c1,cout,res,mean", [(48, 64, 320, 0, 320, False, 0.0), (48, 32, 640, 320, 640, True, 2.0), (48, 16, 1280, 0, 1280, False, 0.5),
                                                      (24, 64, 320, 0, 320, True, 40.0), (4, 32, 320, 0, 320, False, -3.0)])
def test_rconv_statistics_of_the_output(nb, H, c0, c1, cout, res, mean):
    """The next GroupNorm's tables from the launch's epilogue (pivot-shifted partials per tile / wave row group / channel, folded by
    mmgt_gn_stats_finalize_unet with Chan's update) against the statistics pass over the stored tensor and against fp64 -- also for an output whose
    mean is 40 sigma (a conv bias of 40: an unshifted E[x^2] - mean^2 in fp32 would keep ~3 digits); bitwise repeatable; the output itself is the
    launch's without statistics; every cut of the workgroup."""
    from mmgt_amd import hip
    from mmgt_amd.packing import pack_rconv
    W = H
    x0, x1, w, b, temb, r, scale, shift = _case(nb, H, W, c0, c1, cout, 500 + nb + cout, res=res, temb_rows=2)
    b = b + mean
    g = torch.Generator(device="cpu").manual_seed(9)
    gamma, beta = (0.5 + torch.rand(cout, generator=g)).to(dev()), (torch.rand(cout, generator=g) - 0.5).to(dev())
    wimg = pack_rconv(w)
    try:
        for cb in (0, 320, 256, 160):
            if cb and cout % cb:
                continue
            hip.tune("rconv_cb", cb)
            out, (sc, sh) = hip.gn_silu_conv3x3_unet(x0, scale, shift, wimg, cout, b, temb, nb // 2, r, x1=x1, next_norm=(gamma, beta, 32, 1e-5))
            assert torch.equal(out, hip.gn_silu_conv3x3_unet(x0, scale, shift, wimg, cout, b, temb, nb // 2, r, x1=x1))
            sc2, sh2 = hip.groupnorm_affine(out.view(nb, H * W, cout), gamma, beta, 32, 1e-5)
            o = out.double().view(nb, H * W, 32, cout // 32)
            mu, var = o.mean(dim=(1, 3)), o.var(dim=(1, 3), unbiased=False)
            want_sc = gamma.double().view(1, 32, -1) * (var + 1e-5).rsqrt()[:, :, None]
            want_sh = beta.double().view(1, 32, -1) - mu[:, :, None] * want_sc
            # the error a consumer sees on a normalised value at |x - mean| = sigma, in units of sigma (its own bf16 rounding: 4e-3)
            xs = mu[:, :, None] + var.sqrt()[:, :, None]
            for name, (a_, b_) in (("epilogue", (sc, sh)), ("pass", (sc2, sh2))):
                e_sc = ((a_.double().view(nb, 32, -1) - want_sc) / want_sc).abs().max().item()
                ey = ((xs * a_.double().view(nb, 32, -1) + b_.double().view(nb, 32, -1) - (xs * want_sc + want_sh)).abs() / gamma.double().view(1, 32, -1)).max().item()
                print(f"cb {cb} |mean| / sigma {(mu.abs() / var.sqrt()).mean().item():.1f}: tables from the {name}: relative scale error {e_sc:.2e}, normalised-value error {ey:.2e} sigma")
                assert e_sc < 1e-4 and ey < 2e-4, (name, e_sc, ey)
            out2, (sc3, sh3) = hip.gn_silu_conv3x3_unet(x0, scale, shift, wimg, cout, b, temb, nb // 2, r, x1=x1, next_norm=(gamma, beta, 32, 1e-5))
            assert torch.equal(sc, sc3) and torch.equal(sh, sh3) and torch.equal(out, out2)
    finally:
        hip.tune("rconv_cb", 0)


def test_rconv_rejects_unsupported():
    from mmgt_amd import hip
    assert not hip.gn_silu_conv3x3_unet_supported(torch.float32, 320, 0, 320, 64, 64)
    assert not hip.gn_silu_conv3x3_unet_supported(torch.bfloat16, 320, 0, 320, 8, 8)
    assert not hip.gn_silu_conv3x3_unet_supported(torch.bfloat16, 320, 0, 64, 64, 64)
    assert hip.gn_silu_conv3x3_unet_supported(torch.bfloat16, 1280, 640, 1280, 16, 16)
    x = torch.zeros((1, 16, 16, 320), device=dev(), dtype=torch.bfloat16)
    tab = torch.zeros((2, 1, 320), device=dev())
    with pytest.raises(RuntimeError):
        hip._check(hip.lib().mmgt_gn_silu_conv3x3_unet(x.data_ptr(), 320, None, 0, tab.data_ptr(), x.data_ptr(), None, None, 0, None, x.data_ptr(), 1, 16, 16,
                                                      300, hip.BF16, None), "mmgt_gn_silu_conv3x3_unet")
