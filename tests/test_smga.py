"""SMGA (Stage-1 audio -> pose sampler): the oracle against goldens from the reference's own GestureDecoder /
GestureDiffusion (tools/refgen/gen_smga_golden.py), and the HIP path (mmgt_amd/smga.py) against the oracle and the goldens."""
import json
import os

import numpy as np
import pytest
import torch

from tests.oracle_cache import cached  # noqa: E402

from oracle import smga_ref as R
from tests import smga_cases as sc

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="module")
def gold():
    return {k: torch.from_numpy(v) for k, v in np.load(os.path.join(GOLD, "smga.npz")).items()}


@pytest.fixture(scope="module")
def spec():
    return {k: tuple(v) for k, v in json.load(open(os.path.join(GOLD, "smga_keys.json"))).items()}


def test_oracle_forward_matches_reference_golden(gold, spec):
    sd = sc.smga_state_dict(spec)
    inp = sc.smga_inputs()
    cfg = R.SMGAConfig()
    with torch.no_grad():
        for name, t in (("t999", 999), ("t19", 19)):
            times = torch.full((2,), t, dtype=torch.long)
            torch.testing.assert_close(R.forward(sd, cfg, inp["x"], inp["cond_frame"], inp["cond"], times, True),
                                       gold[f"cond_{name}"], rtol=1e-4, atol=2e-5)
            torch.testing.assert_close(R.forward(sd, cfg, inp["x"], inp["cond_frame"], inp["cond"], times, False),
                                       gold[f"null_{name}"], rtol=1e-4, atol=2e-5)
            torch.testing.assert_close(R.guided_forward(sd, cfg, inp["x"], inp["cond_frame"], inp["cond"], times, 2.0),
                                       gold[f"guided_{name}"], rtol=1e-4, atol=4e-5)


def test_oracle_schedule_and_sampler_match_reference_golden(gold, spec):
    cfg = R.SMGAConfig()
    torch.testing.assert_close(R.cosine_alphas_cumprod(1000), gold["alphas_cumprod"], rtol=0, atol=0)
    pairs = R.ddim_time_pairs(cfg)
    assert len(pairs) == 50 and pairs[0][0] == 999 and pairs[-1] == (19, -1)
    sd = sc.smga_state_dict(spec)
    inp = sc.smga_inputs()
    with torch.no_grad():
        out = R.ddim_sample(sd, cfg, inp["cond_frame"][:1], inp["cond"][:1], sc.sampler_noises())
    # 50 stochastic DDIM steps with x0 clipped to [-1, 1]: fp32 summation-order differences stay below the north-star tolerance
    torch.testing.assert_close(out, gold["ddim_sample"], rtol=1e-3, atol=1e-4)


# ------------------------------------------------------------------------------------------------ HIP path
def _hip_model(spec, dtype):
    from mmgt_amd.smga import GestureDecoder
    m = GestureDecoder(nfeats=402, seq_len=80, latent_dim=512, ff_size=1024, num_layers=8, num_heads=8, cond_feature_dim=1059,
                       device="cuda:0", dtype=dtype)
    m.load_state_dict(sc.smga_state_dict(spec))
    return m


@pytest.mark.gpu
def test_hip_decoder_fp32_matches_reference_golden_and_oracle(gold, spec):
    """GestureDecoder.forward / guided_forward through the C ABI, fp32-I/O mode, against the REFERENCE's outputs (rtol 1e-3 /
    atol 1e-4) at a noisy and a late timestep, both guidance branches."""
    m = _hip_model(spec, torch.float32)
    inp = {k: v.cuda() for k, v in sc.smga_inputs().items()}
    for name, t in (("t999", 999), ("t19", 19)):
        times = torch.full((2,), t, dtype=torch.long)
        torch.testing.assert_close(m(inp["x"], inp["cond_frame"], inp["cond"], times, cond_drop_prob=0.0).cpu(), gold[f"cond_{name}"],
                                   rtol=1e-3, atol=1e-4)
        torch.testing.assert_close(m(inp["x"], inp["cond_frame"], inp["cond"], times, cond_drop_prob=1.0).cpu(), gold[f"null_{name}"],
                                   rtol=1e-3, atol=1e-4)
        torch.testing.assert_close(m.guided_forward(inp["x"], inp["cond_frame"], inp["cond"], times, 2.0).cpu(),
                                   gold[f"guided_{name}"], rtol=1e-3, atol=2e-4)


def smga_bf16_floor(spec, gold):
    """|oracle sampler with bf16 weights / activations (fp32 sampler state: what the HIP bf16 mode does) - reference sample|."""
    inp = sc.smga_inputs()
    noises = sc.sampler_noises()
    sd16 = {k: (v.bfloat16() if v.is_floating_point() else v) for k, v in sc.smga_state_dict(spec).items()}
    cfg = R.SMGAConfig()
    orig = R.guided_forward
    try:
        R.guided_forward = lambda sd, c, x, cf, ce, tc, w: orig(sd16, c, x.bfloat16(), cf.bfloat16(), ce.bfloat16(), tc, w).float()
        with torch.no_grad():
            return (R.ddim_sample(None, cfg, inp["cond_frame"][:1], inp["cond"][:1], noises) - gold["ddim_sample"]).abs()
    finally:
        R.guided_forward = orig


@pytest.mark.gpu
def test_hip_ddim_sampler_matches_reference_golden(gold, spec):
    """The whole 50-step guided DDIM sampler (eta = 1, x0 clipped) on the reference's own noise draws: fp32 mode against the
    reference's sample; bf16 product mode reported and held to the error of the oracle run under CPU bf16 (x 1.5)."""
    from mmgt_amd.smga import GestureDiffusion
    inp = sc.smga_inputs()
    noises = sc.sampler_noises()
    m = _hip_model(spec, torch.float32)
    out = GestureDiffusion(m, 80, 402).ddim_sample((1, 80, 402), inp["cond_frame"][:1].cuda(), inp["cond"][:1].cuda(), noises=noises)
    d = (out.cpu() - gold["ddim_sample"]).abs()
    print(f"SMGA sampler fp32 mode vs reference: max|d| {d.max().item():.3e}")
    torch.testing.assert_close(out.cpu(), gold["ddim_sample"], rtol=1e-3, atol=1e-4)
    m16 = _hip_model(spec, torch.bfloat16)
    out16 = GestureDiffusion(m16, 80, 402).ddim_sample((1, 80, 402), inp["cond_frame"][:1].cuda(), inp["cond"][:1].cuda(), noises=noises)
    d16 = (out16.cpu() - gold["ddim_sample"]).abs()
    floor = cached("smga_sampler_bf16_floor", lambda: smga_bf16_floor(spec, gold))
    print(f"SMGA sampler bf16 mode: HIP max|d| {d16.max().item():.3e} mean {d16.mean().item():.3e}; CPU-bf16 floor max "
          f"{floor.max().item():.3e} mean {floor.mean().item():.3e}")
    assert torch.isfinite(out16).all() and d16.mean() <= 1.5 * floor.mean() + 1e-4
    # the maximum too: 50 clipped eta = 1 steps saturate a few coordinates at +-1, where one flipped clip decision is a full-range
    # error in either implementation, so the bound is the floor's own maximum x 1.5 (plus one bf16 ulp at 1.0)
    assert d16.max() <= 1.5 * floor.max() + 2 ** -7


@pytest.mark.gpu
def test_hip_ddim_sampler_graph_replay_equals_the_eager_loop(spec):
    """Without injected noises the 50-step loop is captured into a HIP graph (launch-bound: ~5000 small kernels) and replayed over static
    buffers: same kernels, same draw order -> bitwise the eager loop, also on a second replay with other inputs; and it is faster."""
    import time
    from mmgt_amd import hip
    from mmgt_amd.smga import GestureDiffusion
    inp = sc.smga_inputs()
    m = _hip_model(spec, torch.bfloat16)
    diff = GestureDiffusion(m, 80, 402)

    def sample(row, seed, eager):
        gen = torch.Generator(device="cuda").manual_seed(seed)
        if eager:
            hip.tune("smga_graph", 0)
        try:
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            out = diff.ddim_sample((1, 80, 402), inp["cond_frame"][row:row + 1].cuda(), inp["cond"][row:row + 1].cuda(), generator=gen)
            torch.cuda.synchronize()
            return out.cpu(), time.perf_counter() - t0
        finally:
            hip.tune("smga_graph", 1)
    e0, te = sample(0, 5, True)
    sample(0, 5, False)                                   # builds the graph
    g0, tg = sample(0, 5, False)
    assert torch.isfinite(g0).all() and torch.equal(g0, e0)
    e1, _ = sample(1, 9, True)
    g1, _ = sample(1, 9, False)
    assert torch.equal(g1, e1) and not torch.equal(g1, g0)
    print(f"SMGA 50-step sample: eager {te * 1e3:.1f} ms, graph replay {tg * 1e3:.1f} ms")
    assert tg < te


@pytest.mark.gpu
def test_hip_ddim_sampler_graph_is_rebuilt_after_a_weight_reload_or_a_constant_change(spec):
    """A captured graph holds raw pointers into the model's weight tensors and the loop's host constants: after
    load_state_dict with other weights, or a change of guidance_weight / eta, the replay path must equal the eager loop on the NEW
    state, not replay the stale capture."""
    from mmgt_amd import hip
    from mmgt_amd.smga import GestureDiffusion
    inp = sc.smga_inputs()
    m = _hip_model(spec, torch.bfloat16)
    diff = GestureDiffusion(m, 80, 402)
    cf, ce = inp["cond_frame"][:1].cuda(), inp["cond"][:1].cuda()

    def sample(eager):
        gen = torch.Generator(device="cuda").manual_seed(3)
        if eager:
            hip.tune("smga_graph", 0)
        try:
            return diff.ddim_sample((1, 80, 402), cf, ce, generator=gen).cpu()
        finally:
            hip.tune("smga_graph", 1)
    g0 = sample(False)
    assert torch.equal(sample(False), g0) and len(diff._graphs) == 1
    sd2 = {k: (v * 0.5 if v.is_floating_point() and v.dim() >= 2 else v) for k, v in sc.smga_state_dict(spec).items()}
    m.load_state_dict(sd2)
    torch.cuda.empty_cache()                              # the old weight blocks really go back to the driver
    junk = [torch.full((1 << 20,), float("nan"), device="cuda") for _ in range(64)]   # ... or are recycled with poison
    g1, e1 = sample(False), sample(True)
    assert torch.isfinite(g1).all() and torch.equal(g1, e1) and not torch.equal(g1, g0)
    assert len(diff._graphs) == 1                         # the stale entry is gone
    del junk
    diff.guidance_weight = 1.0
    g2, e2 = sample(False), sample(True)
    assert torch.equal(g2, e2) and not torch.equal(g2, g1)
    diff.eta = 0.5
    g3, e3 = sample(False), sample(True)
    assert torch.equal(g3, e3) and not torch.equal(g3, g2)


@pytest.mark.gpu
def test_hip_decoder_forward_does_not_reuse_a_recycled_condition(gold, spec):
    """forward() caches the condition-only state; a NEW condition allocated at the freed address of the previous one (what the
    caching allocator does with per-slice `.cuda()` tensors) or updated in place must not hit that cache."""
    inp = sc.smga_inputs()
    m = _hip_model(spec, torch.float32)
    x, times = inp["x"][:1].cuda(), torch.full((1,), 999, dtype=torch.long)
    cf, ce = inp["cond_frame"][:1].cuda(), inp["cond"][:1].cuda()
    a = m(x, cf, ce, times).clone()
    ptrs = (cf.data_ptr(), ce.data_ptr())
    ce.mul_(-0.5)                                   # in place: same address, new version
    b_ = m(x, cf, ce, times).clone()
    assert (a - b_).abs().max() > 1e-4
    ce2_host = (inp["cond"][:1] * 0.25 + 0.1)
    del ce
    torch.cuda.synchronize()
    ce2 = ce2_host.cuda()                           # very likely the recycled block; either way the result must be the new cond's
    c = m(x, cf, ce2, times).clone()
    m._prep = None
    c_ref = m(x, cf, ce2, times)
    torch.testing.assert_close(c, c_ref, rtol=0, atol=0)
    assert (c - b_).abs().max() > 1e-4
