"""bf16 noise floor of the operator at the benchmarked shape (BASELINE config 2, 512x512x24): the CPU oracle run with every
weight and activation in bfloat16 (PyTorch CPU kernels: fp32 accumulate, bf16 storage between ops -- what the reference's
`.to(dtype)` modules would do if its bank hand-off accepted bf16) against its own fp32 run on identical
inputs.  Minutes of CPU, ~20 GB: run in the build container; the summary is committed as
tests/golden/unet3d_full_cfg2_bf16floor.npz and read by tests/test_unet_gpu.py as the gate for the bf16 product mode.

    python tools/gen_bf16_floor.py [--fp32 cached_fp32_output.npy]
"""
import argparse
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mmgt_amd.synthetic import synth_state_dict  # noqa: E402
from mmgt_amd.unet3d_spec import unet3d_spec  # noqa: E402
from oracle import unet3d_ref as R  # noqa: E402
from tests import golden_cases as gc  # noqa: E402


def run(sd, inp, dt):
    c = lambda t: t.to(dt) if torch.is_tensor(t) and t.is_floating_point() else t
    s = {k: c(v) for k, v in sd.items()}
    with torch.no_grad():
        return R.unet3d_forward(s, R.UNet3DConfig(), c(inp["sample"]), inp["timestep"], c(inp["ehs"]), c(inp["audio"]),
                                c(inp["pose"]), [c(x) for x in inp["full"]], [c(x) for x in inp["face"]],
                                [c(x) for x in inp["lips"]], inp["motion_scale"],
                                {k: c(v) for k, v in inp["banks"].items()}).float()


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--fp32", default=None, help="cached fp32 oracle output (.npy) of the same case")
    ap.add_argument("--case", default="full_cfg2")
    a = ap.parse_args()
    case = gc.UNET_CASES[a.case]
    inp = gc.unet_inputs(case)
    sd = synth_state_dict(unet3d_spec())
    ref = torch.from_numpy(np.load(a.fp32)) if a.fp32 else run(sd, inp, torch.float32)
    t0 = time.time()
    low = run(sd, inp, torch.bfloat16)
    d = (low - ref).abs()
    print(f"bf16 oracle {time.time() - t0:.0f} s; floor max|d| {d.max().item():.4e} mean|d| {d.mean().item():.4e} "
          f"(ref mean|x| {ref.abs().mean().item():.4f})")
    out = os.path.join(ROOT, "tests", "golden", f"unet3d_{a.case}_bf16floor.npz")
    np.savez_compressed(out, max_abs=d.max().numpy(), mean_abs=d.mean().numpy(), p999_abs=d.flatten().kthvalue(int(d.numel() * 0.999)).values.numpy(),
                        ref_mean_abs=ref.abs().mean().numpy())
    print("wrote", out)
