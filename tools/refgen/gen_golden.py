"""Generate tests/golden/*.npz by running the REFERENCE's own modules (build container only).

    python tools/refgen/gen_golden.py [--skip-full]

Inputs and weights are pure functions of names (mmgt_amd/synthetic.py), so the fixtures hold only the reference's
OUTPUTS (small tensors) plus the scalar parameters needed to regenerate the inputs.  No reference source or bytecode
is written anywhere (sys.dont_write_bytecode).  The case definitions live in tests/golden_cases.py so that the tests
regenerate exactly the same inputs.
"""
import argparse
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools", "refgen"))
import refload  # noqa: E402

refload.setup()
from tests import golden_cases as gc  # noqa: E402
from mmgt_amd.synthetic import synth_state_dict  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")


def save(name, **arrays):
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **{k: (v.detach().numpy() if torch.is_tensor(v) else np.asarray(v)) for k, v in arrays.items()})
    print(f"wrote {path} ({os.path.getsize(path)} B)")


def gen_context():
    from src.pipelines.context import uniform
    out = {}
    for (L, ctx, ov) in gc.CONTEXT_CASES:
        wins = list(uniform(0, 25, L, ctx, 1, ov))
        out[f"L{L}_c{ctx}_o{ov}"] = np.asarray(wins, dtype=np.int64)
    save("context_windows", **out)


def gen_unet(case_name):
    case = gc.UNET_CASES[case_name]
    from src.models.attention import TemporalBasicTransformerBlock
    from src.models.mutual_self_attention import ReferenceAttentionControl
    t0 = time.time()
    m = refload.build_reference_unet3d(block_out_channels=list(case["block_out_channels"]),
                                       cross_attention_dim=case["cross_attention_dim"],
                                       audio_attention_dim=case["audio_attention_dim"])
    sd = synth_state_dict(m.state_dict())
    m.load_state_dict(sd)
    del sd
    ReferenceAttentionControl(m, mode="read", do_classifier_free_guidance=True, batch_size=1, fusion_blocks="full")
    inp = gc.unet_inputs(case)
    for name, mod in m.named_modules():
        if isinstance(mod, TemporalBasicTransformerBlock):
            key = name.replace(".transformer_blocks.0", "")
            mod.bank = [inp["banks"][key].to(torch.float16)]          # what update(writer) stores: mutual_self_attention.py:340
    outs = {}
    for mode in (("script",) if case_name == "full_cfg2" else ("script", "eval")):
        if mode == "script":                                          # scripts/pose2vid.py:151-156,183-184
            m.train()
            m.enable_gradient_checkpointing()
        else:
            m.eval()
        with torch.no_grad():
            outs[mode] = m(inp["sample"], inp["timestep"], encoder_hidden_states=inp["ehs"],
                           audio_embedding=inp["audio"], pose_cond_fea=inp["pose"], full_mask=inp["full"],
                           face_mask=inp["face"], body_mask=inp["lips"], motion_scale=inp["motion_scale"],
                           return_dict=False)[0]
    print(case_name, "ref forward done in", round(time.time() - t0, 1), "s; mean|x|", outs["script"].abs().mean().item())
    if case_name == "full_cfg2":       # G4: the full tensor is 3 MB; commit the strided sub-sample + moments only
        full_path = os.environ.get("MMGT_G4_FULL")          # optional scratch copy (never committed) for oracle cross-checks
        if full_path:
            np.save(full_path, outs["script"].numpy())
        outs = gc.g4_summary(outs["script"])
    save("unet3d_" + case_name, **outs)


def gen_blocks():
    """Full-width single modules at reduced H, W, F (G3)."""
    from src.models.resnet import ResnetBlock3D, Downsample3D, Upsample3D
    from src.models.transformer_3d import Transformer3DModel
    from src.models.motion_module import get_motion_module
    from src.models.mutual_self_attention import ReferenceAttentionControl
    from src.models.attention import TemporalBasicTransformerBlock
    out = {}
    for name, c in gc.BLOCK_CASES.items():
        kind = c["kind"]
        inp = gc.block_inputs(name)
        if kind == "resnet":
            mod = ResnetBlock3D(in_channels=c["cin"], out_channels=c["cout"], temb_channels=1280, eps=1e-5, groups=32,
                                non_linearity="silu", use_inflated_groupnorm=True)
            mod.load_state_dict(synth_state_dict(mod.state_dict(), prefix=name + "."))
            with torch.no_grad():
                out[name] = mod(inp["x"], inp["temb"])
        elif kind == "down":
            mod = Downsample3D(c["c"], use_conv=True, out_channels=c["c"], padding=1, name="op")
            mod.load_state_dict(synth_state_dict(mod.state_dict(), prefix=name + "."))
            with torch.no_grad():
                out[name] = mod(inp["x"])
        elif kind == "up":
            mod = Upsample3D(c["c"], use_conv=True, out_channels=c["c"])
            mod.load_state_dict(synth_state_dict(mod.state_dict(), prefix=name + "."))
            with torch.no_grad():
                out[name] = mod(inp["x"])
        elif kind == "spatial":
            mod = Transformer3DModel(8, c["c"] // 8, in_channels=c["c"], num_layers=1, cross_attention_dim=768,
                                     norm_num_groups=32, unet_use_cross_frame_attention=False,
                                     unet_use_temporal_attention=False)
            mod.load_state_dict(synth_state_dict(mod.state_dict(), prefix=name + "."))

            class _U(torch.nn.Module):           # ReferenceAttentionControl only needs torch_dfs(unet)
                def __init__(self, m):
                    super().__init__()
                    self.m = m
            ReferenceAttentionControl(_U(mod), mode="read", do_classifier_free_guidance=True, batch_size=1,
                                      fusion_blocks="full")
            for m_ in mod.modules():
                if isinstance(m_, TemporalBasicTransformerBlock):
                    m_.bank = [inp["bank"].to(torch.float16)]
            with torch.no_grad():
                out[name] = mod(inp["x"], encoder_hidden_states=inp["ehs"]).sample
        elif kind == "audio":
            mod = Transformer3DModel(8, c["cin"] // 8, in_channels=c["c"], num_layers=1, cross_attention_dim=768,
                                     norm_num_groups=32, use_audio_module=True, depth=c["depth"],
                                     unet_block_name="down", stack_enable_blocks_name=["up", "down", "mid"],
                                     stack_enable_blocks_depth=[0, 1, 2, 3], unet_use_cross_frame_attention=False,
                                     unet_use_temporal_attention=False)
            mod.load_state_dict(synth_state_dict(mod.state_dict(), prefix=name + "."))
            with torch.no_grad():
                out[name] = mod(inp["x"], encoder_hidden_states=inp["audio"], full_mask=inp["full"],
                                face_mask=inp["face"], body_mask=inp["lips"], motion_scale=inp["motion_scale"],
                                return_dict=False)[0]
                out[name + "_unweighted"] = mod(inp["x"], encoder_hidden_states=inp["audio"], full_mask=inp["full"],
                                                face_mask=inp["face"], body_mask=inp["lips"], motion_scale=None,
                                                return_dict=False)[0]
        elif kind == "motion":
            mod = get_motion_module(c["c"], "Vanilla", refload.UNET_ADDITIONAL_KWARGS["motion_module_kwargs"])
            mod.load_state_dict(synth_state_dict(mod.state_dict(), prefix=name + "."))
            with torch.no_grad():
                out[name] = mod(inp["x"], None, None)
        print(name, tuple(out[name].shape), "mean|x|", out[name].abs().mean().item())
    save("blocks", **out)


def gen_side_models():
    from src.models.pose_guider import PoseGuider
    from src.models.audio_proj import AudioProjModel
    pg = PoseGuider(320, block_out_channels=(16, 32, 96, 256))
    pg.load_state_dict(synth_state_dict(pg.state_dict(), prefix="pose_guider."))
    inp = gc.side_inputs()
    with torch.no_grad():
        pose = pg(inp["pose_rgb"])
    ap = AudioProjModel(seq_len=5, blocks=12, channels=768, intermediate_dim=512, output_dim=768, context_tokens=32)
    ap.load_state_dict(synth_state_dict(ap.state_dict(), prefix="audioproj."))
    with torch.no_grad():
        aud = ap(inp["audio_feats"])
    print("pose", tuple(pose.shape), pose.abs().mean().item(), "audio", tuple(aud.shape), aud.abs().mean().item())
    save("side_models", pose_guider=pose, audio_proj=aud)


def gen_refnet(case_name):
    """ReferenceNet in write mode: the 16 banks ReferenceAttentionControl.update() would hand to the denoiser."""
    from src.models.unet_2d_condition import UNet2DConditionModel
    from src.models.mutual_self_attention import ReferenceAttentionControl
    from src.models.attention import BasicTransformerBlock
    case = gc.REFNET_CASES[case_name]
    cfg = dict(sample_size=64, in_channels=4, out_channels=4, center_input_sample=False, flip_sin_to_cos=True,
               freq_shift=0, down_block_types=["CrossAttnDownBlock2D"] * 3 + ["DownBlock2D"],
               mid_block_type="UNetMidBlock2DCrossAttn", up_block_types=["UpBlock2D"] + ["CrossAttnUpBlock2D"] * 3,
               block_out_channels=list(case["block_out_channels"]), layers_per_block=2, downsample_padding=1,
               mid_block_scale_factor=1, act_fn="silu", norm_num_groups=32, norm_eps=1e-5,
               cross_attention_dim=case["cross_attention_dim"], attention_head_dim=8)
    m = UNet2DConditionModel.from_config(cfg).eval()          # from_pretrained => eval (scripts/pose2vid.py:146-148)
    m.load_state_dict(synth_state_dict(m.state_dict(), prefix="refnet."))
    ReferenceAttentionControl(m, do_classifier_free_guidance=True, mode="write", batch_size=1, fusion_blocks="full")
    inp = gc.refnet_inputs(case)
    with torch.no_grad():
        out = m(inp["latents"], inp["timestep"], encoder_hidden_states=inp["ehs"], return_dict=False)[0]
    banks = {}
    for name, mod in m.named_modules():
        if isinstance(mod, BasicTransformerBlock):
            assert len(mod.bank) == 1
            banks[name.replace(".transformer_blocks.0", "")] = mod.bank[0]
    print(case_name, "refnet keys", len(m.state_dict()), "banks", len(banks), "out", tuple(out.shape))
    save("refnet_" + case_name, sample=out, **{"bank." + k: v for k, v in banks.items()})
    if case_name == "full":
        import json
        json.dump({k: list(v.shape) for k, v in m.state_dict().items()}, open(os.path.join(OUT, "unet2d_keys_full.json"), "w"))


def gen_interp():
    from src.pipelines.utils import linear, slerp
    inp = gc.interp_inputs()
    save("interp", linear=linear(inp["v0"], inp["v1"], 0.25), slerp=slerp(inp["v0"], inp["v1"], 0.25),
         slerp_parallel=slerp(inp["v0"], inp["v0"] * 1.0001, 0.25))


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--skip-full", action="store_true")
    ap.add_argument("--only", default=None)
    a = ap.parse_args()
    os.makedirs(OUT, exist_ok=True)
    torch.set_grad_enabled(True)
    steps = {"context": gen_context, "interp": gen_interp, "side": gen_side_models, "blocks": gen_blocks,
             "refnet_tiny": lambda: gen_refnet("tiny"), "refnet_full": lambda: gen_refnet("full"),
             "tiny": lambda: gen_unet("tiny"), "full": lambda: gen_unet("full_cfg1"),
             "full_cfg2": lambda: gen_unet("full_cfg2")}
    for k, fn in steps.items():
        if a.only and k != a.only:
            continue
        if k == "full_cfg2" and a.only != k:        # minutes of CPU and ~15 GB: only on request
            continue
        if k == "full" and a.skip_full:
            continue
        fn()
