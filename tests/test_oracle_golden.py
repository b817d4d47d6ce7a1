"""The oracle (oracle/*.py, CPU fp32) against golden vectors produced by the REFERENCE's own modules
(tools/refgen/gen_golden.py, run in the build container).  These pin the oracle; the HIP path is then checked
against the oracle in the -m gpu tests."""
import os

import numpy as np
import pytest
import torch

from mmgt_amd.synthetic import synth_state_dict, synth_tensor
from oracle import unet3d_ref as R
from tests import golden_cases as gc

TOL = dict(rtol=1e-4, atol=2e-5)   # fp32 summation-order noise only


def _load(golden_dir, name):
    return {k: torch.from_numpy(v) for k, v in np.load(os.path.join(golden_dir, name + ".npz")).items()}


def _prefixed(sd, prefix):
    return {k[len(prefix):]: v for k, v in sd.items() if k.startswith(prefix)}


# ----------------------------------------------------------------------------------------------- single blocks (G3)

def _resnet_spec(cin, cout):
    s = {"norm1.weight": (cin,), "norm1.bias": (cin,), "conv1.weight": (cout, cin, 3, 3), "conv1.bias": (cout,),
         "time_emb_proj.weight": (cout, 1280), "time_emb_proj.bias": (cout,), "norm2.weight": (cout,),
         "norm2.bias": (cout,), "conv2.weight": (cout, cout, 3, 3), "conv2.bias": (cout,)}
    if cin != cout:
        s.update({"conv_shortcut.weight": (cout, cin, 1, 1), "conv_shortcut.bias": (cout,)})
    return s


def _attn_spec(p, dim, inner, ctx):
    return {f"{p}.to_q.weight": (inner, dim), f"{p}.to_k.weight": (inner, ctx), f"{p}.to_v.weight": (inner, ctx),
            f"{p}.to_out.0.weight": (dim, inner), f"{p}.to_out.0.bias": (dim,)}


def _ff_spec(p, dim):
    return {f"{p}.net.0.proj.weight": (8 * dim, dim), f"{p}.net.0.proj.bias": (8 * dim,),
            f"{p}.net.2.weight": (dim, 4 * dim), f"{p}.net.2.bias": (dim,)}


def _norm_spec(p, c):
    return {f"{p}.weight": (c,), f"{p}.bias": (c,)}


def _spatial_spec(c, inner, audio=False):
    t = "transformer_blocks.0"
    s = {}
    s.update(_norm_spec("norm", c))
    s.update({"proj_in.weight": (inner, c, 1, 1), "proj_in.bias": (inner,), "proj_out.weight": (c, inner, 1, 1),
              "proj_out.bias": (c,)})
    for n in ("norm1", "norm2", "norm3"):
        s.update(_norm_spec(f"{t}.{n}", inner))
    s.update(_attn_spec(f"{t}.attn1", inner, inner, inner))
    s.update(_ff_spec(f"{t}.ff", inner))
    if audio:
        for i in range(3):
            s.update(_attn_spec(f"{t}.attn2_{i}", inner, inner, 768))
        for z in ("zero_conv_full", "zero_conv_face", "zero_conv_lip"):
            s.update({f"{t}.{z}.weight": (inner, inner, 1, 1), f"{t}.{z}.bias": (inner,)})
    else:
        s.update(_attn_spec(f"{t}.attn2", inner, inner, 768))
    return s


def _motion_spec(c):
    q, t = "temporal_transformer", "temporal_transformer.transformer_blocks.0"
    s = {}
    s.update(_norm_spec(f"{q}.norm", c))
    s.update({f"{q}.proj_in.weight": (c, c), f"{q}.proj_in.bias": (c,), f"{q}.proj_out.weight": (c, c),
              f"{q}.proj_out.bias": (c,)})
    for i in range(2):
        s.update(_attn_spec(f"{t}.attention_blocks.{i}", c, c, c))
        s[f"{t}.attention_blocks.{i}.pos_encoder.pe"] = (1, 32, c)
        s.update(_norm_spec(f"{t}.norms.{i}", c))
    s.update(_ff_spec(f"{t}.ff", c))
    s.update(_norm_spec(f"{t}.ff_norm", c))
    return s


def oracle_block(name):
    """Run the oracle on one G3 case; shared with the GPU parity tests."""
    c = gc.BLOCK_CASES[name]
    inp = gc.block_inputs(name)
    cfg = R.UNet3DConfig()
    kind, f = c["kind"], c["f"]
    x5 = inp["x"]
    b = x5.shape[0]
    x = x5.permute(0, 2, 1, 3, 4).reshape(b * f, x5.shape[1], *x5.shape[3:])
    back = lambda y: y.reshape(b, f, *y.shape[1:]).permute(0, 2, 1, 3, 4)
    outs = {}
    if kind == "resnet":
        sd = synth_state_dict(_resnet_spec(c["cin"], c["cout"]), prefix=name + ".")
        outs[name] = back(R.resnet_block(_dotted(sd), "r", x, inp["temb"], cfg, f))
    elif kind == "down":
        sd = synth_state_dict({"conv.weight": (c["c"], c["c"], 3, 3), "conv.bias": (c["c"],)}, prefix=name + ".")
        outs[name] = back(torch.nn.functional.conv2d(x, sd["conv.weight"], sd["conv.bias"], stride=2, padding=1))
    elif kind == "up":
        sd = synth_state_dict({"conv.weight": (c["c"], c["c"], 3, 3), "conv.bias": (c["c"],)}, prefix=name + ".")
        y = torch.nn.functional.interpolate(x, scale_factor=2.0, mode="nearest")
        outs[name] = back(torch.nn.functional.conv2d(y, sd["conv.weight"], sd["conv.bias"], padding=1))
    elif kind == "spatial":
        sd = synth_state_dict(_spatial_spec(c["c"], c["c"]), prefix=name + ".")
        outs[name] = back(R.spatial_transformer(_dotted(sd), "r", x, inp["ehs"], inp["bank"], cfg, f))
    elif kind == "audio":
        sd = synth_state_dict(_spatial_spec(c["c"], c["cin"], audio=True), prefix=name + ".")
        audio = inp["audio"].reshape(b * f, 32, 768)
        masks = (inp["full"], inp["face"], inp["lips"])
        outs[name] = back(R.audio_transformer(_dotted(sd), "r", x, audio, masks, c["depth"], inp["motion_scale"], cfg))
        outs[name + "_unweighted"] = back(R.audio_transformer(_dotted(sd), "r", x, audio, masks, c["depth"], None, cfg))
    elif kind == "motion":
        sd = synth_state_dict(_motion_spec(c["c"]), prefix=name + ".")
        outs[name] = back(R.motion_module(_dotted(sd), "r", x, cfg, f))
    return outs


def _dotted(sd):
    return {"r." + k: v for k, v in sd.items()}


@pytest.mark.parametrize("name", list(gc.BLOCK_CASES))
def test_block_matches_reference(golden_dir, name):
    g = _load(golden_dir, "blocks")
    for k, v in oracle_block(name).items():
        torch.testing.assert_close(v, g[k], **TOL)


# ----------------------------------------------------------------------------------------------- whole UNet (G2, config 1)

def unet_spec(cfg: R.UNet3DConfig):
    """Key -> shape table of UNet3DConditionModel for a config (1526 keys at full width)."""
    from mmgt_amd.unet3d_spec import unet3d_spec
    return unet3d_spec(cfg.block_out_channels, cfg.cross_attention_dim, cfg.audio_attention_dim)


def run_oracle_unet(case_name, weighted, sd=None):
    case = gc.UNET_CASES[case_name]
    cfg = R.UNet3DConfig(block_out_channels=case["block_out_channels"], cross_attention_dim=case["cross_attention_dim"],
                         audio_attention_dim=case["audio_attention_dim"])
    if sd is None:
        sd = synth_state_dict(unet_spec(cfg))
    inp = gc.unet_inputs(case)
    with torch.no_grad():
        return R.unet3d_forward(sd, cfg, inp["sample"], inp["timestep"], inp["ehs"], inp["audio"], inp["pose"],
                                inp["full"], inp["face"], inp["lips"], inp["motion_scale"], inp["banks"],
                                weighted=weighted)


@pytest.mark.parametrize("mode", ["script", "eval"])
def test_unet_tiny_matches_reference(golden_dir, mode):
    g = _load(golden_dir, "unet3d_tiny")
    out = run_oracle_unet("tiny", weighted=(mode == "script"))
    torch.testing.assert_close(out, g[mode], **TOL)


def test_unet_tiny_modes_differ(golden_dir):
    g = _load(golden_dir, "unet3d_tiny")
    assert (g["script"] - g["eval"]).abs().max() > 1e-4      # motion_scale=[1,1,2] must matter


def test_unet_full_width_config1_matches_reference(golden_dir):
    g = _load(golden_dir, "unet3d_full_cfg1")
    out = run_oracle_unet("full_cfg1", weighted=True)
    torch.testing.assert_close(out, g["script"], **TOL)


# ----------------------------------------------------------------------------------------------- side models (G6)

def test_pose_guider_and_audio_proj(golden_dir):
    g = _load(golden_dir, "side_models")
    inp = gc.side_inputs()
    pg_spec = {"conv_in.weight": (16, 3, 3, 3), "conv_in.bias": (16,)}
    chans = (16, 32, 96, 256)
    for i in range(3):
        pg_spec[f"blocks.{2 * i}.weight"] = (chans[i], chans[i], 3, 3)
        pg_spec[f"blocks.{2 * i}.bias"] = (chans[i],)
        pg_spec[f"blocks.{2 * i + 1}.weight"] = (chans[i + 1], chans[i], 3, 3)
        pg_spec[f"blocks.{2 * i + 1}.bias"] = (chans[i + 1],)
    pg_spec["conv_out.weight"] = (320, 256, 3, 3)
    pg_spec["conv_out.bias"] = (320,)
    sd = synth_state_dict(pg_spec, prefix="pose_guider.")
    torch.testing.assert_close(R.pose_guider_forward(sd, inp["pose_rgb"]), g["pose_guider"], **TOL)
    ap_spec = {"proj1.weight": (512, 46080), "proj1.bias": (512,), "proj2.weight": (512, 512), "proj2.bias": (512,),
               "proj3.weight": (32 * 768, 512), "proj3.bias": (32 * 768,), "norm.weight": (768,), "norm.bias": (768,)}
    sd = synth_state_dict(ap_spec, prefix="audioproj.")
    torch.testing.assert_close(R.audio_proj_forward(sd, inp["audio_feats"]), g["audio_proj"], **TOL)


def test_unet3d_spec_matches_reference_keys(golden_dir):
    """mmgt_amd.unet3d_spec reproduces the reference's 1526 state-dict keys and shapes."""
    import json
    from mmgt_amd.unet3d_spec import unet3d_spec
    ref = json.load(open(os.path.join(golden_dir, "unet3d_keys_full.json")))
    mine = unet3d_spec()
    assert len(ref) == 1526 and set(ref) == set(mine)
    assert all(tuple(ref[k]) == tuple(mine[k]) for k in ref)


# ----------------------------------------------------------------------------------------------- ReferenceNet (U10)

def _refnet_oracle(case_name):
    from mmgt_amd.unet3d_spec import unet2d_reference_spec
    case = gc.REFNET_CASES[case_name]
    cfg = R.UNet3DConfig(block_out_channels=case["block_out_channels"], cross_attention_dim=case["cross_attention_dim"])
    sd = synth_state_dict(unet2d_reference_spec(case["block_out_channels"], case["cross_attention_dim"]), prefix="refnet.")
    inp = gc.refnet_inputs(case)
    with torch.no_grad():
        return R.reference_net_banks(sd, cfg, inp["latents"], inp["timestep"], inp["ehs"])


@pytest.mark.parametrize("case_name", ["tiny", "full"])
def test_reference_net_banks_match_reference(golden_dir, case_name):
    g = _load(golden_dir, "refnet_" + case_name)
    banks, sample = _refnet_oracle(case_name)
    assert list(banks) == [k[len("bank."):] for k in g if k.startswith("bank.")]      # module order down -> up -> mid
    for k, v in banks.items():
        torch.testing.assert_close(v, g["bank." + k], **TOL)
    torch.testing.assert_close(sample, g["sample"], **TOL)


def test_unet2d_reference_spec_matches_reference_keys(golden_dir):
    import json
    from mmgt_amd.unet3d_spec import unet2d_reference_spec
    ref = json.load(open(os.path.join(golden_dir, "unet2d_keys_full.json")))
    mine = unet2d_reference_spec()
    assert len(ref) == 682 and set(ref) == set(mine) and all(tuple(ref[k]) == tuple(mine[k]) for k in ref)


def test_oracle_cache_entries_are_current_and_reproduce():
    """tests/golden/oracle_cache: every committed entry carries the key of the CURRENT oracle / case sources (a stale entry
    would only make the GPU suite slow again, never wrong: it is recomputed), and one entry is recomputed here and must match
    bit for bit (the entries are this oracle's outputs, nothing else)."""
    import glob
    import os
    from tests import oracle_cache as oc
    files = sorted(glob.glob(os.path.join(oc.CACHE_DIR, "*.pt")))
    assert len(files) >= 6
    for f in files:
        name = os.path.splitext(os.path.basename(f))[0]
        assert torch.load(f, map_location="cpu")["key"] == oc.entry_key(name), f"{name}: regenerate with tools/gen_oracle_cache.py"
    from tests import test_pipeline_gpu as TP
    got = TP.oracle_pipeline_fp32(TP.build_weights("cpu"), 8, 12, 4)
    ref = torch.load(os.path.join(oc.CACHE_DIR, "pipeline_fp32_8_12_4.pt"), map_location="cpu")["value"]
    assert torch.equal(got["want"], ref["want"]) and all(torch.equal(a, b) for a, b in zip(got["traj"], ref["traj"]))
