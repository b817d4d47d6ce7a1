#!/usr/bin/env python3
"""Writes tests/golden/oracle_cache/*.pt: the ORACLE results of the slowest GPU parity tests, computed on the CPU by the very
functions those tests call (tests/oracle_cache.py explains the key that ties an entry to the sources it came from).

    python tools/gen_oracle_cache.py [name ...]        # all entries, or the named ones
"""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["MMGT_WRITE_ORACLE_CACHE"] = "1"
from tests.oracle_cache import CACHE_DIR, cached  # noqa: E402


def main():
    only = set(sys.argv[1:])
    want = lambda n: not only or n in only
    torch.set_num_threads(max(1, (os.cpu_count() or 2)))
    t0 = time.time()

    def emit(name, fn):
        if not want(name):
            return
        p = os.path.join(CACHE_DIR, name + ".pt")
        if os.path.exists(p):
            os.remove(p)
        t = time.time()
        cached(name, fn)
        print(f"{name}: {time.time() - t:.0f} s, {os.path.getsize(p) / 1024:.0f} KiB", flush=True)

    if not only or only & {"pipeline_fp32_8_12_4", "pipeline_fp32_14_8_2", "pipeline_bf16_floor", "long_video_96"}:
        from tests import test_pipeline_gpu as TP
        sds_cpu = TP.build_weights("cpu")
        emit("pipeline_fp32_8_12_4", lambda: TP.oracle_pipeline_fp32(sds_cpu, 8, 12, 4))
        emit("pipeline_fp32_14_8_2", lambda: TP.oracle_pipeline_fp32(sds_cpu, 14, 8, 2))
        emit("pipeline_bf16_floor", lambda: TP.oracle_pipeline_bf16_floor(sds_cpu))
        emit("long_video_96", lambda: TP.oracle_long_video(sds_cpu))
        del sds_cpu
    if want("unet_512x512_six_frames"):
        from mmgt_amd.synthetic import synth_state_dict
        from mmgt_amd.unet3d_spec import unet3d_spec
        from tests import golden_cases as gc
        from tests import test_unet_gpu as TU
        sd_cpu = synth_state_dict(unet3d_spec(), device="cpu")
        case = TU.SIX_FRAME_CASE
        emit("unet_512x512_six_frames", lambda: TU._run_oracle(sd_cpu, case))
        del sd_cpu
    if want("unet_512x512_twelve_frames"):
        from mmgt_amd.synthetic import synth_state_dict
        from mmgt_amd.unet3d_spec import unet3d_spec
        from tests import test_unet_gpu as TU
        sd_cpu = synth_state_dict(unet3d_spec(), device="cpu")
        emit("unet_512x512_twelve_frames", lambda: TU._run_oracle(sd_cpu, TU.TWELVE_FRAME_CASE))
        del sd_cpu
    if want("vae_decode_512x512_frame"):
        from tests import test_vae as TV
        emit("vae_decode_512x512_frame", TV.oracle_full_resolution_frame)
    if want("smga_sampler_bf16_floor"):
        from tests import test_smga as TS
        gold = {k: torch.from_numpy(v) for k, v in np.load(os.path.join(ROOT, "tests", "golden", "smga.npz")).items()}
        spec = {k: tuple(v) for k, v in json.load(open(os.path.join(ROOT, "tests", "golden", "smga_keys.json"))).items()}
        emit("smga_sampler_bf16_floor", lambda: TS.smga_bf16_floor(spec, gold))
    print(f"done in {time.time() - t0:.0f} s")


if __name__ == "__main__":
    main()
