"""Generate tests/golden/wav2vec.npz + wav2vec_keys.json by running the REFERENCE's `Wav2VecModel` (src/models/wav2vec.py: a
transformers Wav2Vec2Model whose conv features are linearly interpolated to `seq_len` frames) on the transformers build of this
container, with the hash-seeded weights and waveform of tests/golden_cases.py (wav2vec2-base geometry: the audio_encoder of
src/dataset/audio_processor.py:76-131).  Only key names / shapes and outputs are stored.   python tools/refgen/gen_wav2vec_golden.py"""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(1, "/root/reference")
sys.dont_write_bytecode = True
from tests import wav2vec_cases as gc  # noqa: E402

import transformers  # noqa: E402
from transformers import Wav2Vec2Config  # noqa: E402
from src.models.wav2vec import Wav2VecModel  # noqa: E402  (the reference's class)

cfg = Wav2Vec2Config(attn_implementation="eager")                 # base geometry = facebook/wav2vec2-base-960h's config.json
assert cfg.hidden_size == 768 and cfg.num_hidden_layers == 12 and cfg.feat_extract_norm == "group" and not cfg.do_stable_layer_norm
m = Wav2VecModel(cfg).eval()
keys = {k: list(v.shape) for k, v in m.state_dict().items()}
json.dump(keys, open(os.path.join(ROOT, "tests", "golden", "wav2vec_keys.json"), "w"), indent=0)
sd = gc.wav2vec_state_dict(keys)
missing, unexpected = m.load_state_dict(sd, strict=True)
wave = gc.wav2vec_wave()
seq_len = gc.WAV2VEC_SEQ_LEN
with torch.no_grad():
    out = m(wave, seq_len=seq_len, output_hidden_states=True)
    feats = m.feature_extract(wave, seq_len)
emb = torch.stack(out.hidden_states[1:], dim=1).squeeze(0)                    # audio_processor.py:122-123: (12, S, 768)
emb = emb.permute(1, 0, 2).contiguous()                                       # "b s d -> s b d": (S, 12, 768)
res = {"transformers_version": np.array(transformers.__version__), "audio_emb": emb.numpy(), "features": feats[0].numpy(),
       "last_hidden_state": out.last_hidden_state[0].numpy()}
path = os.path.join(ROOT, "tests", "golden", "wav2vec.npz")
np.savez_compressed(path, **res)
print("wrote", path, os.path.getsize(path), "B;", emb.shape, float(emb.abs().mean()), len(keys), "keys")
