"""Generate tests/golden/clip_vision.npz from the transformers build installed in the build container:
`transformers.CLIPVisionModelWithProjection` (the class the reference instantiates, scripts/pose2vid.py:158-162) with the
hash-seeded weights and inputs of tests/golden_cases.py.  Only outputs are stored.   python tools/refgen/gen_clip_golden.py"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tests import golden_cases as gc  # noqa: E402

import transformers  # noqa: E402
from transformers import CLIPVisionConfig, CLIPVisionModelWithProjection  # noqa: E402

out = {"transformers_version": np.array(transformers.__version__)}
for name, case in gc.CLIP_CASES.items():
    cfg = CLIPVisionConfig(**{k: v for k, v in case.items() if k != "batch"})
    assert cfg.hidden_act == "quick_gelu" and cfg.layer_norm_eps == 1e-5
    m = CLIPVisionModelWithProjection(cfg).eval()
    sd = gc.clip_state_dict(case)
    missing, unexpected = m.load_state_dict(sd, strict=False)
    assert not unexpected and all("position_ids" in k for k in missing), (missing, unexpected)
    with torch.no_grad():
        r = m(pixel_values=gc.clip_pixels(case))
    out[name + ".image_embeds"] = r.image_embeds.numpy()
    out[name + ".last_hidden_state"] = r.last_hidden_state.numpy()[:, :, :64]      # a column slice keeps the file small
    print(name, r.image_embeds.shape, float(r.image_embeds.abs().mean()))
path = os.path.join(ROOT, "tests", "golden", "clip_vision.npz")
np.savez_compressed(path, **out)
print("wrote", path, os.path.getsize(path), "B")
