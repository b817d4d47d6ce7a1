"""torch.ops.mmgt_hip.* (mmgt_amd/torch_ops.py): dispatcher registration over the C ABI."""
import pytest
import torch
import torch.nn.functional as F

import mmgt_amd.torch_ops as T
from mmgt_amd.synthetic import hash_uniform


def test_ops_are_registered_with_schemas_and_have_no_cpu_kernel():
    for name in T.OPS:
        op = getattr(torch.ops.mmgt_hip, name)
        assert "mmgt_hip::" + name in str(op.default._schema)
    with pytest.raises(NotImplementedError):
        torch.ops.mmgt_hip.gemm(torch.zeros(4, 64), torch.zeros(8, 64), None, None, 0)


def test_fake_kernels_give_shapes():
    from torch._subclasses.fake_tensor import FakeTensorMode
    with FakeTensorMode():
        x = torch.empty(6, 8, 8, 64, device="cuda", dtype=torch.bfloat16)
        w = torch.empty(128, 9 * 64, device="cuda", dtype=torch.bfloat16)
        assert torch.ops.mmgt_hip.conv3x3_nhwc(x, w, None, None, 2, False).shape == (6, 4, 4, 128)
        assert torch.ops.mmgt_hip.conv3x3_nhwc(x, w, None, None, 1, True).shape == (6, 16, 16, 128)
        a = torch.empty(10, 64, device="cuda")
        assert torch.ops.mmgt_hip.gemm(a, torch.empty(128, 64, device="cuda"), None, None, 1).shape == (10, 64)


@pytest.mark.gpu
def test_ops_match_torch_math_fp32():
    dev = "cuda:0"
    r = lambda n, s, sc=1.0: hash_uniform(n, s, sc).to(dev)
    a, w, b, res = r("a", (200, 320)), r("w", (640, 320), 0.05), r("b", (640,)), r("res", (200, 640))
    torch.testing.assert_close(torch.ops.mmgt_hip.gemm(a, w, b, res, 0), F.linear(a, w, b) + res, rtol=1e-3, atol=1e-4)
    q, k, v = r("q", (3, 50, 320)), r("k", (3, 70, 320)), r("v", (3, 70, 320))
    sp = lambda t: t.view(3, -1, 8, 40).transpose(1, 2)
    want = F.scaled_dot_product_attention(sp(q), sp(k), sp(v)).transpose(1, 2).reshape(3, 50, 320)
    torch.testing.assert_close(torch.ops.mmgt_hip.attention(q, k, v, 8, 40 ** -0.5), want, rtol=1e-3, atol=1e-4)
    x = r("x", (2, 64, 320), 2.0)
    g, be = r("g", (320,), 0.2) + 1, r("be", (320,), 0.2)
    want = F.silu(F.group_norm(x.permute(0, 2, 1), 32, g, be, 1e-5)).permute(0, 2, 1)
    torch.testing.assert_close(torch.ops.mmgt_hip.groupnorm_silu(x, g, be, 32, 1e-5, True), want, rtol=1e-3, atol=1e-4)
    torch.testing.assert_close(torch.ops.mmgt_hip.layernorm(x.view(128, 320), g, be, 1e-5),
                               F.layer_norm(x.view(128, 320), (320,), g, be, 1e-5), rtol=1e-3, atol=1e-4)
    from mmgt_amd.packing import pack_conv3x3
    xi, wc = r("xi", (2, 8, 8, 64)), r("wc", (128, 64, 3, 3), 0.05)
    want = F.conv2d(xi.permute(0, 3, 1, 2), wc, None, stride=2, padding=1).permute(0, 2, 3, 1)
    got = torch.ops.mmgt_hip.conv3x3_nhwc(xi, pack_conv3x3(wc).to(dev), None, None, 2, False)
    torch.testing.assert_close(got, want, rtol=1e-3, atol=1e-4)


def test_fake_kernels_of_the_fused_ops_give_shapes():
    from torch._subclasses.fake_tensor import FakeTensorMode
    with FakeTensorMode():
        e = lambda *s, dt=torch.bfloat16: torch.empty(*s, device="cuda", dtype=dt)
        outs = torch.ops.mmgt_hip.attn_bank_fwd(e(4, 256, 640), e(4, 320, 256), e(2, 64, 320), e(2, 320, 64), 8, 40 ** -0.5, 2, 0, True)
        assert [o.shape for o in outs] == [(4, 256, 320)] * 2
        assert torch.ops.mmgt_hip.temporal_attn(e(16, 9, 960), 8, 8).shape == (16, 9, 320)
        assert torch.ops.mmgt_hip.mmhaa_cross(e(2, 64, 960), e(2, 32, 1920), e(3, 128, dt=torch.float32), 8).shape == (2, 64, 960)
        x = e(256, 320)
        assert torch.ops.mmgt_hip.ff_fused(x, e(320, dt=torch.float32), e(320, dt=torch.float32), e(10, dt=torch.uint8),
                                           e(320, dt=torch.float32), 1280, None, None, None).shape == (256, 320)
        assert torch.ops.mmgt_hip.rowgemm320(x, e(10, dt=torch.uint8), 960, None, None, None, None).shape == (256, 960)
        assert torch.ops.mmgt_hip.vae_decode(e(2, 4, 8, 8), [e(3)]).shape == (2, 3, 64, 64)
        assert torch.ops.mmgt_hip.temporal_leg(e(2 * 24 * 8, 320), e(320, dt=torch.float32), e(32, 320, dt=torch.float32), e(10, dt=torch.uint8),
                                               e(320, dt=torch.float32), 2, 24, 8).shape == (384, 320)
        assert torch.ops.mmgt_hip.gn_silu_conv3x3(e(2, 32, 32, 640), e(2, 32, 32, 320), e(960, dt=torch.float32), e(960, dt=torch.float32), 32, 1e-5,
                                                  e(10, dt=torch.uint8), 640, None, e(2, 640, dt=torch.float32), None).shape == (2, 32, 32, 640)


@pytest.mark.gpu
def test_attention_family_ops_match_torch_math():
    """attn_bank_fwd (read-mode reference attention, + its twin), temporal_attn, mmhaa_cross against SDPA restatements of the reference's
    call sites (mutual_self_attention.py:149-188, motion_module.py:351-388, attention.py:700-760)."""
    dev = "cuda:0"
    r = lambda n, s, sc=1.0: hash_uniform("to." + n, s, sc).to(dev)
    heads, hd, frames = 8, 40, 2
    inner = heads * hd
    sp = lambda t: t.reshape(t.shape[0], t.shape[1], heads, -1).transpose(1, 2)
    # ---- bank attention, fp32: images 2, 3 (the conditional row) also read the bank of row 1
    B, n, nb = 4, 64, 32
    qk, v = r("qk", (B, n, 2 * inner)), r("v", (B, n, inner))
    kb, vb = r("kb", (2, nb, inner)), r("vb", (2, nb, inner))
    out = torch.ops.mmgt_hip.attn_bank_fwd(qk, v.transpose(1, 2).contiguous(), kb, vb.transpose(1, 2).contiguous(), heads, hd ** -0.5, frames, 2, False)[0]
    for b in range(B):
        k, vv = qk[b:b + 1, :, inner:], v[b:b + 1]
        if b >= 2:
            k, vv = torch.cat([k, kb[b // frames][None]], 1), torch.cat([vv, vb[b // frames][None]], 1)
        want = F.scaled_dot_product_attention(sp(qk[b:b + 1, :, :inner]), sp(k), sp(vv)).transpose(1, 2).reshape(1, n, inner)
        torch.testing.assert_close(out[b:b + 1], want, rtol=1e-3, atol=1e-4)
    # ---- twin, bf16 (the production kernel): ONE pass gives [own + bank] and [own]
    n2 = 256
    qk16, v16 = r("qk16", (2, n2, 2 * inner)).bfloat16(), r("v16", (2, n2, inner)).bfloat16()
    kb16, vb16 = r("kb16", (1, 64, inner)).bfloat16(), r("vb16", (1, 64, inner)).bfloat16()
    both, own = torch.ops.mmgt_hip.attn_bank_fwd(qk16, v16.transpose(1, 2).contiguous(), kb16, vb16.transpose(1, 2).contiguous(), heads, hd ** -0.5,
                                                 2, 0, True)
    f64 = lambda t: t.double()
    k_all, v_all = torch.cat([qk16[..., inner:], kb16.expand(2, -1, -1)], 1), torch.cat([v16, vb16.expand(2, -1, -1)], 1)
    want_both = F.scaled_dot_product_attention(sp(f64(qk16[..., :inner])), sp(f64(k_all)), sp(f64(v_all))).transpose(1, 2).reshape(2, n2, inner)
    want_own = F.scaled_dot_product_attention(sp(f64(qk16[..., :inner])), sp(f64(qk16[..., inner:])), sp(f64(v16))).transpose(1, 2).reshape(2, n2, inner)
    torch.testing.assert_close(both.double(), want_both, rtol=0, atol=2e-2)
    torch.testing.assert_close(own.double(), want_own, rtol=0, atol=2e-2)
    # ---- temporal attention over the frame axis of ((b f), hw, 3C)
    b, f, hw = 2, 8, 9
    qkv = r("qkv", (b * f, hw, 3 * inner))
    seq = lambda t: t.reshape(b, f, hw, heads, hd).permute(0, 2, 3, 1, 4).reshape(b * hw, heads, f, hd)
    want = F.scaled_dot_product_attention(seq(qkv[..., :inner]), seq(qkv[..., inner:2 * inner]), seq(qkv[..., 2 * inner:]))
    want = want.reshape(b, hw, heads, f, hd).permute(0, 3, 1, 2, 4).reshape(b * f, hw, inner)
    torch.testing.assert_close(torch.ops.mmgt_hip.temporal_attn(qkv, f, heads), want, rtol=1e-3, atol=1e-4)
    # ---- the three masked audio cross-attentions
    B, n, la = 2, 64, 32
    q3, kv3 = r("q3", (B, n, 3 * inner)), r("kv3", (B, la, 6 * inner))
    ms = (r("ms", (3, B * n)).abs() * torch.tensor([1.0, 1.0, 2.0], device=dev)[:, None]).contiguous()
    got = torch.ops.mmgt_hip.mmhaa_cross(q3, kv3, ms, heads)
    for i in range(3):
        sl = slice(i * inner, (i + 1) * inner)
        a = F.scaled_dot_product_attention(sp(q3[..., sl]), sp(kv3[..., sl]), sp(kv3[..., 3 * inner:][..., sl])).transpose(1, 2).reshape(B, n, inner)
        torch.testing.assert_close(got[..., sl], a * ms[i].view(B, n, 1), rtol=1e-3, atol=1e-4)


@pytest.mark.gpu
def test_fused_projection_ops_match_torch_math_bf16():
    """ff_fused (+ proj_out), rowgemm320: bf16-only kernels against fp64 of the same bf16 operands (one output ulp + accumulation noise)."""
    from mmgt_amd.packing import pack_ff_fused, pack_ff_proj_out, pack_rowgemm
    dev = "cuda:0"
    r = lambda n, s, sc=1.0: hash_uniform("tf." + n, s, sc).to(dev)
    M, C, inner = 512, 320, 1280
    x = r("x", (M, C), 1.5).bfloat16()
    g, be = 1 + 0.2 * r("g", (C,)), 0.1 * r("be", (C,))
    w1, b1 = (r("w1", (2 * inner, C)) * C ** -0.5).bfloat16(), 0.1 * r("b1", (2 * inner,))
    w2, b2 = (r("w2", (C, inner)) * inner ** -0.5).bfloat16(), 0.1 * r("b2", (C,))
    ln = F.layer_norm(x.double(), (C,), g.double(), be.double(), 1e-5).bfloat16().double()
    hcat = ln @ w1.double().t() + b1.double()
    hid = (hcat[:, :inner] * F.gelu(hcat[:, inner:])).bfloat16().double()
    want = x.double() + hid @ w2.double().t() + b2.double()
    got = torch.ops.mmgt_hip.ff_fused(x, g, be, pack_ff_fused(w1, b1, w2), b2, inner, None, None, None)
    torch.testing.assert_close(got.double(), want, rtol=2 ** -7, atol=2e-2)
    wpo, bpo, res2 = (r("wpo", (C, C)) * C ** -0.5).bfloat16(), 0.1 * r("bpo", (C,)), r("res2", (M, C)).bfloat16()
    want_po = res2.double() + want.bfloat16().double() @ wpo.double().t() + bpo.double()
    got_po = torch.ops.mmgt_hip.ff_fused(x, g, be, pack_ff_fused(w1, b1, w2), b2, inner, pack_ff_proj_out(wpo), bpo, res2)
    torch.testing.assert_close(got_po.double(), want_po, rtol=2 ** -7, atol=3e-2)
    wq, bq = (r("wq", (960, C)) * C ** -0.5).bfloat16(), 0.1 * r("bq", (960,))
    want_q = ln @ wq.double().t() + bq.double()
    got_q = torch.ops.mmgt_hip.rowgemm320(x, pack_rowgemm(wq), 960, bq, g, be, None)
    torch.testing.assert_close(got_q.double(), want_q, rtol=2 ** -7, atol=2e-2)
    # temporal_leg: the op against the same fp64 restatement tests/test_tleg_gpu.py uses
    from mmgt_amd.packing import pack_tleg
    from tests import test_tleg_gpu as TT
    B, Fr, n = 2, 12, 16
    w = TT._weights("op", scale=2.0)
    xt = r("xt", (B * Fr * n, C), 1.5).bfloat16()
    got_t = torch.ops.mmgt_hip.temporal_leg(xt, w["g"], w["bpe"], pack_tleg(w["q"], w["k"], w["v"], w["o"]), w["bo"], B, Fr, 8)
    want_t = TT._ref(xt, w, B, Fr, n, 40 ** -0.5)
    assert ((got_t.double() - want_t).abs() <= 2.0 ** -8 * want_t.abs() + 8e-3).all()
    # gn_silu_conv3x3: GroupNorm -> SiLU -> conv3x3 (+ temb, + residual) over a two-source input, against fp64 with the bf16 rounding of the normalised tensor
    from mmgt_amd.packing import pack_rconv
    nb, hh, c0, c1, co = 2, 32, 320, 64, 320
    xa, xb = r("rc.x", (nb, hh, hh, c0), 1.5).bfloat16(), r("rc.s", (nb, hh, hh, c1)).bfloat16()
    gg, bb = 1 + 0.2 * r("rc.g", (c0 + c1,)), 0.1 * r("rc.b", (c0 + c1,))
    wc = r("rc.w", (co, c0 + c1, 3, 3)) * (9 * (c0 + c1)) ** -0.5
    bc, te, rs = 0.1 * r("rc.bc", (co,)), 0.3 * r("rc.te", (nb, co)), r("rc.r", (nb, hh, hh, co)).bfloat16()
    got_c = torch.ops.mmgt_hip.gn_silu_conv3x3(xa, xb, gg, bb, 32, 1e-5, pack_rconv(wc), co, bc, te, rs)
    xc = torch.cat([xa, xb], 3).double().permute(0, 3, 1, 2)
    nrm = F.silu(F.group_norm(xc, 32, gg.double(), bb.double(), 1e-5)).bfloat16().double()
    want_c = F.conv2d(nrm, wc.bfloat16().double(), bc.double(), padding=1).permute(0, 2, 3, 1) + te.double()[:, None, None, :] + rs.double()
    assert ((got_c.double() - want_c).abs() <= 2.0 ** -7 * want_c.abs() + 2e-2).all()


@pytest.mark.gpu
def test_vae_decode_op_matches_oracle():
    from mmgt_amd.synthetic import synth_state_dict
    from mmgt_amd.vae import vae_decoder_spec
    from oracle import vae_ref
    sd = synth_state_dict(vae_decoder_spec(), prefix="vae.", device="cpu")
    lat = hash_uniform("vae.lat", (1, 4, 2, 8, 8), 1.0)
    with torch.no_grad():
        ref = vae_ref.decode_latents(sd, lat)                                         # (1, 3, 2, 64, 64) in [0, 1]
    weights = [sd[k].cuda() for k in vae_decoder_spec()]
    z = (lat[0] / 0.18215).permute(1, 0, 2, 3).contiguous().cuda()
    out = torch.ops.mmgt_hip.vae_decode(z, weights)
    torch.testing.assert_close((out / 2 + 0.5).clamp(0, 1).cpu(), ref[0].permute(1, 0, 2, 3), rtol=1e-3, atol=1e-4)
    assert torch.equal(torch.ops.mmgt_hip.vae_decode(z, weights), out)                # the cached packing
