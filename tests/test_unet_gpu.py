"""The HIP denoise-step operator (mmgt_amd.UNet3DConditionModel, through the C ABI) against the oracle on identical
weights, latents and conditioning.

  * fp32-I/O mode: rtol 1e-3 / atol 1e-4 (the north-star tolerance) at BASELINE config-1 geometry, full width.
  * bf16 mode: gated against the bf16 noise floor of this 1.4 B-parameter network (PyTorch's own CPU bf16 path differs
    from fp32 by max|d| 1.0e-2 on outputs of mean|x| 0.29): max|d| <= 6e-2, mean|d| <= 8e-3.
"""
import pytest
import torch

pytestmark = pytest.mark.gpu

from mmgt_amd.synthetic import synth_state_dict  # noqa: E402
from oracle import unet3d_ref as R  # noqa: E402
from tests import golden_cases as gc  # noqa: E402


@pytest.fixture(scope="module")
def full_sd():
    from mmgt_amd.unet3d_spec import unet3d_spec
    sd_gpu = synth_state_dict(unet3d_spec(), device="cuda:0")
    sd_cpu = {k: v.cpu() for k, v in sd_gpu.items()}
    return sd_gpu, sd_cpu


def _to_dev(inp):
    mv = lambda t: t.cuda() if torch.is_tensor(t) else t
    out = {k: ([mv(x) for x in v] if isinstance(v, list) and torch.is_tensor(v[0]) else mv(v)) for k, v in inp.items()
           if k != "banks"}
    out["banks"] = {k: v.cuda() for k, v in inp["banks"].items()}
    return out


def _run_hip(sd_gpu, case, dtype, weighted=True):
    from mmgt_amd.unet3d import UNet3DConditionModel
    m = UNet3DConditionModel(device="cuda:0", dtype=dtype)
    m.load_state_dict(sd_gpu)
    if weighted:
        m.enable_gradient_checkpointing()          # scripts/pose2vid.py:183-184
    else:
        m.eval()
    inp = _to_dev(gc.unet_inputs(case))
    m.set_banks(inp["banks"])
    out = m(inp["sample"], inp["timestep"], encoder_hidden_states=inp["ehs"], audio_embedding=inp["audio"],
            pose_cond_fea=inp["pose"], full_mask=inp["full"], face_mask=inp["face"], body_mask=inp["lips"],
            motion_scale=inp["motion_scale"], return_dict=False)[0]
    torch.cuda.synchronize()
    return out.float().cpu()


def _run_oracle(sd_cpu, case, weighted=True):
    cfg = R.UNet3DConfig()
    inp = gc.unet_inputs(case)
    with torch.no_grad():
        return R.unet3d_forward(sd_cpu, cfg, inp["sample"], inp["timestep"], inp["ehs"], inp["audio"], inp["pose"],
                                inp["full"], inp["face"], inp["lips"], inp["motion_scale"], inp["banks"],
                                weighted=weighted)


def test_unet_fp32_mode_matches_oracle_config1(full_sd):
    sd_gpu, sd_cpu = full_sd
    case = gc.UNET_CASES["full_cfg1"]
    ref = _run_oracle(sd_cpu, case)
    out = _run_hip(sd_gpu, case, torch.float32)
    print("fp32 mode: max|d|", (out - ref).abs().max().item(), "mean|x|", ref.abs().mean().item())
    torch.testing.assert_close(out, ref, rtol=1e-3, atol=1e-4)


def test_unet_fp32_mode_eval_semantics(full_sd):
    """eval() => motion_scale ignored."""
    sd_gpu, sd_cpu = full_sd
    case = gc.UNET_CASES["full_cfg1"]
    ref = _run_oracle(sd_cpu, case, weighted=False)
    out = _run_hip(sd_gpu, case, torch.float32, weighted=False)
    torch.testing.assert_close(out, ref, rtol=1e-3, atol=1e-4)


def test_unet_matches_reference_golden_config1(full_sd, golden_dir):
    """Straight against the REFERENCE's own output (tests/golden/unet3d_full_cfg1.npz)."""
    import numpy as np, os
    sd_gpu, _ = full_sd
    g = torch.from_numpy(np.load(os.path.join(golden_dir, "unet3d_full_cfg1.npz"))["script"])
    out = _run_hip(sd_gpu, gc.UNET_CASES["full_cfg1"], torch.float32)
    torch.testing.assert_close(out, g, rtol=1e-3, atol=1e-4)


def test_unet_bf16_mode_within_noise_floor(full_sd):
    sd_gpu, sd_cpu = full_sd
    case = gc.UNET_CASES["full_cfg1"]
    ref = _run_oracle(sd_cpu, case)
    out = _run_hip(sd_gpu, case, torch.bfloat16)
    d = (out - ref).abs()
    print("bf16 mode: max|d|", d.max().item(), "mean|d|", d.mean().item(), "mean|x|", ref.abs().mean().item())
    assert torch.isfinite(out).all()
    assert d.max() <= 6e-2 and d.mean() <= 8e-3


def test_unet_wider_geometry_fp32(full_sd):
    """16x16 latent (128x128 px), 5 frames: exercises multi-tile attention, ragged GEMM tiles and window length != 8."""
    sd_gpu, sd_cpu = full_sd
    case = dict(gc.UNET_CASES["full_cfg1"], frames=5, latent=16, timestep=21)
    ref = _run_oracle(sd_cpu, case)
    out = _run_hip(sd_gpu, case, torch.float32)
    torch.testing.assert_close(out, ref, rtol=1e-3, atol=1e-4)


def test_unet_full_resolution_512x512_six_frames(full_sd):
    """BASELINE config-2 spatial size (512x512 px = 64x64 latent, 4096 tokens and 4096 bank keys at level 0) on 6 frames:
    the shapes at which the production tile configurations are chosen (128x320 conv tile from 49152 output rows, 256x128
    for the GEGLU / long-K GEMMs, 64-key attention tiles without ragged-tail code).  fp32-I/O mode against the CPU oracle
    at the north-star tolerance, bf16 product mode against the same reference at the bf16 noise floor."""
    sd_gpu, sd_cpu = full_sd
    case = dict(gc.UNET_CASES["full_cfg1"], frames=6, latent=64, timestep=499)
    ref = _run_oracle(sd_cpu, case)
    out = _run_hip(sd_gpu, case, torch.float32)
    d = (out - ref).abs()
    print("512x512x6 fp32 mode: max|d|", d.max().item(), "mean|x|", ref.abs().mean().item())
    torch.testing.assert_close(out, ref, rtol=1e-3, atol=1e-4)
    out16 = _run_hip(sd_gpu, case, torch.bfloat16)
    d16 = (out16 - ref).abs()
    print("512x512x6 bf16 mode: max|d|", d16.max().item(), "mean|d|", d16.mean().item())
    assert torch.isfinite(out16).all()
    assert d16.max() <= 8e-2 and d16.mean() <= 8e-3
