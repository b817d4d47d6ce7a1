"""wav2vec2 audio features (src/dataset/audio_processor.py:76-131 -> src/models/wav2vec.py): the oracle against the outputs
of the REFERENCE's own Wav2VecModel class (tests/golden/wav2vec.npz, tools/refgen/gen_wav2vec_golden.py), the HIP model against both."""
import json
import os

import numpy as np
import pytest
import torch

from tests import wav2vec_cases as gc

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="module")
def gold():
    return {k: torch.from_numpy(v) for k, v in np.load(os.path.join(GOLD, "wav2vec.npz")).items() if k != "transformers_version"}


@pytest.fixture(scope="module")
def keys():
    return {k: tuple(v) for k, v in json.load(open(os.path.join(GOLD, "wav2vec_keys.json"))).items()}


def test_wav2vec_spec_matches_reference_keys(keys):
    from mmgt_amd.wav2vec import wav2vec_spec
    mine = wav2vec_spec(weight_norm_keys="parametrized")
    assert len(keys) == 211 and list(mine) == list(keys) and all(tuple(mine[k]) == keys[k] for k in keys)
    ck = wav2vec_spec(weight_norm_keys="checkpoint")      # the spelling facebook/wav2vec2-base-960h holds
    assert "encoder.pos_conv_embed.conv.weight_g" in ck and len(ck) == len(keys)


def test_oracle_matches_reference_golden(gold, keys):
    from oracle import wav2vec_ref as R
    sd = gc.wav2vec_state_dict(keys)
    with torch.no_grad():
        feats = R.feature_extract(sd, gc.wav2vec_wave(), gc.WAV2VEC_SEQ_LEN)
        emb = R.audio_emb(sd, gc.wav2vec_wave(), gc.WAV2VEC_SEQ_LEN)
    torch.testing.assert_close(feats[0], gold["features"], rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(emb, gold["audio_emb"], rtol=1e-4, atol=5e-5)
    assert emb.shape == (gc.WAV2VEC_SEQ_LEN, 12, 768) and emb.abs().mean() > 0.5


@pytest.mark.gpu
def test_hip_wav2vec_matches_reference_golden(gold, keys):
    """fp32-I/O mode at the north-star tolerance against the reference's outputs; bf16 product mode against the same, gated by the
    oracle run under CPU bf16 (x 1.5), both spellings of the weight-norm keys."""
    from mmgt_amd.wav2vec import Wav2VecModel
    from oracle import wav2vec_ref as R
    sd = gc.wav2vec_state_dict(keys)
    wave = gc.wav2vec_wave().cuda()
    m = Wav2VecModel(device="cuda:0", dtype=torch.float32)
    m.load_state_dict(sd)
    feats = m.feature_extract(wave, gc.WAV2VEC_SEQ_LEN)
    torch.testing.assert_close(feats[0].cpu(), gold["features"], rtol=1e-3, atol=1e-4)
    out = m(wave, seq_len=gc.WAV2VEC_SEQ_LEN, output_hidden_states=True)
    assert len(out.hidden_states) == 13
    torch.testing.assert_close(out.last_hidden_state[0].cpu(), gold["last_hidden_state"], rtol=1e-3, atol=2e-4)
    emb = m.audio_emb(wave, gc.WAV2VEC_SEQ_LEN)
    d = (emb.cpu() - gold["audio_emb"]).abs()
    print(f"wav2vec fp32 mode vs reference: max|d| {d.max().item():.3e}")
    torch.testing.assert_close(emb.cpu(), gold["audio_emb"], rtol=1e-3, atol=2e-4)
    # checkpoint spelling of the weight norm
    sd2 = {k.replace("parametrizations.weight.original0", "weight_g").replace("parametrizations.weight.original1", "weight_v"): v for k, v in sd.items()}
    m2 = Wav2VecModel(device="cuda:0", dtype=torch.float32)
    m2.load_state_dict(sd2)
    assert torch.equal(m2.audio_emb(wave, gc.WAV2VEC_SEQ_LEN), emb)
    # bf16 product mode
    m16 = Wav2VecModel(device="cuda:0", dtype=torch.bfloat16)
    m16.load_state_dict(sd)
    e16 = m16.audio_emb(wave, gc.WAV2VEC_SEQ_LEN).cpu()
    sd16 = {k: v.bfloat16() for k, v in sd.items()}
    with torch.no_grad():
        floor = (R.audio_emb(sd16, gc.wav2vec_wave().bfloat16(), gc.WAV2VEC_SEQ_LEN).float() - gold["audio_emb"]).abs()
    d16 = (e16 - gold["audio_emb"]).abs()
    print(f"wav2vec bf16 mode: HIP max|d| {d16.max().item():.3e} mean {d16.mean().item():.3e}; CPU-bf16 floor max {floor.max().item():.3e} mean {floor.mean().item():.3e}")
    assert torch.isfinite(e16).all() and d16.mean() <= 1.5 * floor.mean() and d16.max() <= 1.5 * floor.max()
