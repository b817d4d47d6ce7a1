"""Weights / waveform of the wav2vec2 parity case, shared by tools/refgen/gen_wav2vec_golden.py (which runs the reference's
Wav2VecModel) and tests/test_wav2vec.py: pure functions of names via mmgt_amd/synthetic.py."""

WAV2VEC_SEQ_LEN = 24            # frames at 25 fps
WAV2VEC_SAMPLES = 15360         # 0.96 s at 16 kHz


def wav2vec_state_dict(keys, device="cpu"):
    """Hash-seeded weights for every key of the reference's Wav2VecModel.state_dict() (wav2vec2-base geometry): weights
    U(+- 3 / sqrt(fan_in)) so that the signal survives the 7-layer GELU conv stack and the 12 post-LN layers, norm gains 1 +- 0.1,
    biases +- 0.05; the weight-norm gain of the positional conv 2 +- 0.5 (either key spelling: parametrizations.weight.original0 of
    current torch, weight_g of the checkpoint format the reference's transformers 4.30 reads)."""
    import math
    import torch
    from mmgt_amd.synthetic import hash_uniform
    sd = {}
    for k, shape in keys.items():
        shape = tuple(shape)
        name = "w2v." + k.replace("parametrizations.weight.original0", "weight_g").replace("parametrizations.weight.original1", "weight_v")
        if k.endswith("original0") or k.endswith("weight_g"):
            sd[k] = 2.0 + hash_uniform(name, shape, 0.5, device)
        elif len(shape) == 1 and ("norm" in k) and k.endswith("weight"):
            sd[k] = 1.0 + hash_uniform(name, shape, 0.1, device)
        elif len(shape) == 1:
            sd[k] = hash_uniform(name, shape, 0.05 if k.endswith("bias") else 1.0, device)
        else:
            fan_in = 1
            for s in shape[1:]:
                fan_in *= s
            sd[k] = hash_uniform(name, shape, 3.0 / math.sqrt(fan_in), device)
    return {k: v.to(torch.float32) for k, v in sd.items()}


def wav2vec_wave(device="cpu"):
    """(1, 15360) waveform, normalised as Wav2Vec2FeatureExtractor does (zero mean, unit variance: audio_processor.py:107)."""
    from mmgt_amd.synthetic import hash_uniform
    w = hash_uniform("w2v.wave", (1, WAV2VEC_SAMPLES), 1.0, device)
    return (w - w.mean()) / (w.var(unbiased=False) + 1e-7).sqrt()
