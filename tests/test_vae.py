"""VAE decode leg (decode_latents): HIP decoder through the C ABI vs the oracle restatement of diffusers' AutoencoderKL
decoder, plus analytic anchors for the oracle itself (the reference holds no fixture for this third-party piece)."""
import pytest
import torch

from mmgt_amd.synthetic import hash_uniform, synth_state_dict
from oracle import vae_ref


def _sd(device="cpu"):
    from mmgt_amd.vae import vae_decoder_spec
    return synth_state_dict(vae_decoder_spec(), prefix="vae.", device=device)


def test_oracle_vae_shapes_and_frame_independence():
    sd = _sd()
    lat = hash_uniform("vae.lat", (1, 4, 3, 8, 8), 1.0)
    with torch.no_grad():
        v = vae_ref.decode_latents(sd, lat)
        v1 = vae_ref.decode_latents(sd, lat[:, :, 1:2])
    assert v.shape == (1, 3, 3, 64, 64) and v.min() >= 0 and v.max() <= 1
    torch.testing.assert_close(v[:, :, 1:2], v1)                       # decoding is per frame


def test_oracle_vae_known_answer_zero_weights():
    """All conv / linear weights zero: every resnet is the identity on its (zero) input, so the output is conv_out.bias
    everywhere -> decode_latents == clamp(bias / 2 + 0.5)."""
    sd = {k: torch.zeros_like(v) for k, v in _sd().items()}
    sd["decoder.conv_out.bias"] = torch.tensor([-2.0, 0.2, 3.0])
    with torch.no_grad():
        v = vae_ref.decode_latents(sd, torch.randn(1, 4, 2, 8, 8))
    want = torch.tensor([0.0, 0.6, 1.0]).view(1, 3, 1, 1, 1).expand_as(v)
    torch.testing.assert_close(v, want)


@pytest.mark.gpu
@pytest.mark.parametrize("dtype,tol", [(torch.float32, dict(rtol=1e-3, atol=1e-4)), (torch.bfloat16, dict(rtol=0, atol=4e-2))])
def test_hip_vae_matches_oracle(dtype, tol):
    from mmgt_amd.vae import AutoencoderKL
    sd = _sd()
    lat = hash_uniform("vae.lat", (1, 4, 3, 8, 8), 1.0)
    with torch.no_grad():
        ref = vae_ref.decode_latents(sd, lat)
    vae = AutoencoderKL(device="cuda:0", dtype=dtype)
    vae.load_state_dict(sd)
    out = vae.decode_video(lat.cuda(), frames_per_batch=2).cpu()
    assert out.shape == ref.shape
    print(dtype, "max|d|", (out - ref).abs().max().item())
    torch.testing.assert_close(out, ref, **tol)
    one = vae.decode((lat[0, :, 1:2] / 0.18215).permute(1, 0, 2, 3).cuda()).sample.cpu()    # diffusers-style call
    torch.testing.assert_close((one / 2 + 0.5).clamp(0, 1), ref[0, :, 1:2].permute(1, 0, 2, 3), **tol)


@pytest.mark.gpu
def test_hip_vae_full_resolution_frame_is_finite():
    """One 512x512 frame (64x64 latent) in bf16: the size the sampler decodes; checked through frame independence."""
    from mmgt_amd.vae import AutoencoderKL
    vae = AutoencoderKL(device="cuda:0", dtype=torch.bfloat16)
    vae.load_state_dict(_sd("cuda:0"))
    lat = hash_uniform("vae.lat512", (1, 4, 2, 64, 64), 1.0).cuda()
    both = vae.decode_video(lat, frames_per_batch=2)
    single = vae.decode_video(lat[:, :, 1:2].contiguous(), frames_per_batch=1)
    assert both.shape == (1, 3, 2, 512, 512) and torch.isfinite(both).all()
    torch.testing.assert_close(both[:, :, 1:2], single, rtol=0, atol=1e-6)
