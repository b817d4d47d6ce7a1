"""CLIP vision tower (the reference's image_encoder): the oracle restatement against goldens produced by
transformers.CLIPVisionModelWithProjection itself (tests/golden/clip_vision.npz, tools/refgen/gen_clip_golden.py), and the
HIP implementation (through the C ABI) against both."""
import os

import numpy as np
import pytest
import torch

from oracle import clip_ref
from tests import golden_cases as gc


@pytest.fixture(scope="module")
def golden(golden_dir):
    return np.load(os.path.join(golden_dir, "clip_vision.npz"))


@pytest.mark.parametrize("name", list(gc.CLIP_CASES))
def test_oracle_clip_matches_transformers_golden(golden, name):
    case = gc.CLIP_CASES[name]
    sd = gc.clip_state_dict(case)
    with torch.no_grad():
        emb, last = clip_ref.clip_vision_forward(sd, gc.clip_pixels(case), case["num_attention_heads"])
    torch.testing.assert_close(emb, torch.from_numpy(golden[name + ".image_embeds"]), rtol=1e-4, atol=2e-5)
    torch.testing.assert_close(last[:, :, :64], torch.from_numpy(golden[name + ".last_hidden_state"]), rtol=1e-4, atol=5e-5)


def test_clip_spec_has_the_transformers_keys():
    from transformers import CLIPVisionConfig, CLIPVisionModelWithProjection
    from mmgt_amd.clip_vision import clip_vision_spec
    case = gc.CLIP_CASES["tiny"]
    m = CLIPVisionModelWithProjection(CLIPVisionConfig(**{k: v for k, v in case.items() if k != "batch"}))
    ref = {k: tuple(v.shape) for k, v in m.state_dict().items() if "position_ids" not in k}
    spec = {k: tuple(v) for k, v in clip_vision_spec(case["hidden_size"], case["intermediate_size"],
                                                     case["num_hidden_layers"], case["image_size"], case["patch_size"],
                                                     case["projection_dim"]).items()}
    assert spec == ref


def test_clip_fails_loudly_without_gpu():
    from mmgt_amd.clip_vision import CLIPVisionModelWithProjection
    m = CLIPVisionModelWithProjection.__new__(CLIPVisionModelWithProjection)
    m._loaded = True
    with pytest.raises(RuntimeError, match="GPU only"):
        m.forward(torch.zeros(1, 3, 224, 224))


@pytest.mark.gpu
@pytest.mark.parametrize("name", list(gc.CLIP_CASES))
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_hip_clip_matches_golden(golden, name, dtype):
    from mmgt_amd.clip_vision import CLIPVisionModelWithProjection
    case = gc.CLIP_CASES[name]
    m = CLIPVisionModelWithProjection(device="cuda:0", dtype=dtype, **{k: v for k, v in case.items() if k != "batch"})
    m.load_state_dict(gc.clip_state_dict(case))
    out = m(gc.clip_pixels(case).cuda())
    emb, last = out.image_embeds.cpu(), out.last_hidden_state.cpu()
    g_emb, g_last = torch.from_numpy(golden[name + ".image_embeds"]), torch.from_numpy(golden[name + ".last_hidden_state"])
    d = (emb - g_emb).abs()
    print(name, dtype, "image_embeds max|d|", d.max().item(), "mean|x|", g_emb.abs().mean().item())
    if dtype == torch.float32:
        torch.testing.assert_close(emb, g_emb, rtol=1e-3, atol=1e-4)          # the north-star tolerance, vs transformers
        torch.testing.assert_close(last[:, :, :64], g_last, rtol=1e-3, atol=1e-4)
    else:
        assert d.max() <= 6e-2 and d.mean() <= 1.2e-2                          # bf16 noise floor on embeddings of ~0.45
