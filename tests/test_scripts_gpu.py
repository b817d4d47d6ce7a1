"""The counterpart scripts of SURVEY 8c end to end on the device, at BASELINE config-1 size (64x64 px, 8 frames, few DDIM steps):
scripts/pose2vid.py --synthetic and scripts/audio2vid.py --synthetic (BASELINE config 3: SMGA audio->pose chained into Stage 2,
device-side conditioning, uint8 output path)."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(script, *args):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", script), *args], capture_output=True, text=True, cwd=ROOT,
                       timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    return json.loads(r.stdout.strip().splitlines()[-1])


def test_audio2vid_synthetic_chain_smga_into_stage2(tmp_path):
    rec = _run("audio2vid.py", "--synthetic", "-W", "64", "-H", "64", "-L", "8", "--steps", "2", "--dtype", "bf16", "--out_dir",
               str(tmp_path))
    assert rec["video"] == [1, 8, 64, 64, 3] and rec["video_dtype"] == "torch.uint8" and rec["keypoints_finite"] and rec["slices"] == 1
    assert rec["mask_levels"] == [[8, 64], [8, 16], [8, 4], [8, 1]]
    frames = np.load(rec["saved"])
    assert frames.shape == (8, 64, 64, 3) and frames.dtype == np.uint8 and frames.std() > 0


def test_audio2vid_config3_at_full_size_512x512x24(tmp_path):
    """BASELINE config 3 at its stated size: SMGA audio -> pose, device-side conditioning, Stage 2 at 512x512x24 bf16 (2 DDIM steps:
    the per-step cost is the bench's), VAE decode to uint8 (scripts/audio2vid.py:275-498)."""
    rec = _run("audio2vid.py", "--synthetic", "-W", "512", "-H", "512", "-L", "24", "--steps", "2", "--num_c", "24", "--dtype", "bf16",
               "--out_dir", str(tmp_path))
    print({k: rec[k] for k in ("build_s", "smga_s", "conditioning_s", "stage2_s")})
    assert rec["video"] == [1, 24, 512, 512, 3] and rec["video_dtype"] == "torch.uint8" and rec["keypoints_finite"] and rec["slices"] == 1
    assert rec["mask_levels"] == [[24, 4096], [24, 1024], [24, 256], [24, 64]]
    frames = np.load(rec["saved"])
    assert frames.shape == (24, 512, 512, 3) and frames.dtype == np.uint8 and frames.std() > 0


def test_pose2vid_synthetic_with_hands_mask(tmp_path):
    """--hands_mask_path: full = clamp(1 - face + lips + hands, 0, 1) per level (reference :239-271)."""
    rec = _run("pose2vid.py", "--synthetic", "-W", "64", "-H", "64", "-L", "8", "--steps", "2", "--hands_mask_path", "synthetic",
               "--out_dir", str(tmp_path))
    assert rec["video"] == [1, 3, 8, 64, 64] and rec["finite"]


def test_pose2vid_synthetic(tmp_path):
    rec = _run("pose2vid.py", "--synthetic", "-W", "64", "-H", "64", "-L", "8", "--steps", "2", "--out_dir", str(tmp_path))
    assert rec["video"] == [1, 3, 8, 64, 64] and rec["finite"]


def test_pose2vid_clip_parallel_single_rank_launch(tmp_path):
    """BASELINE config 4 plumbing on the one GPU of the test box: torch.distributed.run with one rank -- weights broadcast
    (RCCL), the rank's clip sampled, frames gathered on rank 0.  More ranks only add independent clips."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=1", "--master-addr", "127.0.0.1",
                        "--master-port", str(29600 + os.getpid() % 300), os.path.join(ROOT, "scripts", "pose2vid.py"), "--synthetic",
                        "--clip-parallel", "-W", "64", "-H", "64", "-L", "8", "--steps", "2", "--out_dir", str(tmp_path)],
                       capture_output=True, text=True, cwd=ROOT, timeout=900, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [json.loads(l) for l in r.stdout.strip().splitlines() if l.startswith("{")]
    assert lines[0] == {"clip_parallel_ranks": 1, "clips": 1} and lines[-1]["video"] == [1, 8, 64, 64, 3] and lines[-1]["finite"]   # uint8 frames travel
