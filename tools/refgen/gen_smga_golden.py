"""Generate tests/golden/smga.npz by running the REFERENCE's own Stage-1 modules (build container only):
`GestureDecoder` (src/audio2pose_model/model.py:324-490) and `GestureDiffusion.ddim_sample`
(src/audio2pose_model/diffusion.py:241-274) in the configuration of the SMGA wrapper (src/audio2pose_model/SMGA.py:62-108).

    python tools/refgen/gen_smga_golden.py

The reference imports its package as `audio2pose_model.*` (model.py:9-10), so /root/reference/src goes on sys.path; the absent
`p_tqdm` (one unused import) gets a two-line stand-in.  Weights and inputs are pure functions of names
(mmgt_amd/synthetic.py); the fixture holds the reference's OUTPUTS only, plus tests/golden/smga_keys.json (state-dict key ->
shape), from which the tests rebuild the same weights.  No reference source or bytecode is written anywhere.
"""
import json
import os
import sys

import numpy as np
import torch

sys.dont_write_bytecode = True
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools", "refgen", "standin_smga"))
sys.path.insert(0, "/root/reference/src")

import torch.nn.functional as F  # noqa: E402
from audio2pose_model.diffusion import GestureDiffusion  # noqa: E402
from audio2pose_model.model import GestureDecoder  # noqa: E402

from tests import smga_cases as sc  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")


def main():
    model = GestureDecoder(nfeats=402, seq_len=80, latent_dim=512, ff_size=1024, num_layers=8, num_heads=8, dropout=0.1,
                           cond_feature_dim=1059, activation=F.gelu)                       # SMGA.py:83-93
    diffusion = GestureDiffusion(model, 80, 402, schedule="cosine", n_timestep=1000, predict_epsilon=False, loss_type="l2",
                                 use_p2=False, cond_drop_prob=0.25, guidance_weight=2)     # SMGA.py:95-106
    spec = {k: list(v.shape) for k, v in model.state_dict().items()}
    json.dump(spec, open(os.path.join(OUT, "smga_keys.json"), "w"), indent=0)
    model.load_state_dict(sc.smga_state_dict(spec))
    model.eval()
    diffusion.eval()
    inp = sc.smga_inputs()
    out = {}
    with torch.no_grad():
        for name, t in (("t999", 999), ("t19", 19)):
            times = torch.full((inp["x"].shape[0],), t, dtype=torch.long)
            out[f"cond_{name}"] = model(inp["x"], inp["cond_frame"], inp["cond"], times, cond_drop_prob=0.0)
            out[f"null_{name}"] = model(inp["x"], inp["cond_frame"], inp["cond"], times, cond_drop_prob=1.0)
            out[f"guided_{name}"] = model.guided_forward(inp["x"], inp["cond_frame"], inp["cond"], times, 2)
        # the whole 50-step sampler on the draws torch.manual_seed(SEED) gives the reference on the CPU
        torch.manual_seed(sc.SAMPLER_SEED)
        sample = diffusion.ddim_sample((1, 80, 402), inp["cond_frame"][:1], inp["cond"][:1])
        out["ddim_sample"] = sample
        out["alphas_cumprod"] = diffusion.alphas_cumprod
    for k, v in out.items():
        print(k, tuple(v.shape), float(v.abs().mean()))
    np.savez_compressed(os.path.join(OUT, "smga.npz"), **{k: v.numpy() for k, v in out.items()})
    print("wrote", os.path.join(OUT, "smga.npz"), os.path.getsize(os.path.join(OUT, "smga.npz")))


if __name__ == "__main__":
    main()
