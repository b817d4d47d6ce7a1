"""Inputs / weights of the SMGA (Stage-1 audio -> pose) parity cases, shared by tools/refgen/gen_smga_golden.py (which runs
the reference) and the tests: pure functions of names via mmgt_amd/synthetic.py."""
import torch

from mmgt_amd.synthetic import hash_uniform, synth_tensor

SAMPLER_SEED = 1234


def smga_state_dict(spec, device="cpu"):
    """Hash-seeded values for every key of GestureDecoder.state_dict(); `rotary.freqs` buffers keep their defined values."""
    sd = {}
    for k, shape in spec.items():
        if k.endswith("rotary.freqs"):
            dim = 2 * shape[0]
            sd[k] = (1.0 / (10000 ** (torch.arange(0, dim, 2)[: dim // 2].float() / dim))).to(device)
        elif k in ("null_cond_embed", "null_cond_hidden"):
            sd[k] = hash_uniform("smga." + k, shape, 1.0, device)
        else:
            sd[k] = synth_tensor("smga." + k, shape, device)
    return sd


def smga_inputs(batch=2):
    return dict(x=hash_uniform("smga.x", (batch, 80, 402), 1.0), cond_frame=hash_uniform("smga.cond_frame", (batch, 402), 0.8),
                cond=hash_uniform("smga.cond", (batch, 80, 1059), 1.0))


def sampler_noises(steps=50, shape=(1, 80, 402)):
    """The normal draws GestureDiffusion.ddim_sample makes on the CPU after torch.manual_seed(SAMPLER_SEED):
    randn(shape), then one randn_like per DDIM step except the last (diffusion.py:249,268)."""
    g = torch.Generator().manual_seed(SAMPLER_SEED)
    return [torch.randn(shape, generator=g) for _ in range(steps)]
