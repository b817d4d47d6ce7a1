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


def test_audio2vid_with_a_wav_file_and_a_reference_image(tmp_path):
    """--audio_path (16-kHz PCM .wav through the wav2vec2 leg) and --image_path (PIL) replace the synthetic waveform / reference image."""
    import wave
    from PIL import Image
    t = np.arange(16000) / 16000.0
    pcm = (0.4 * np.sin(2 * np.pi * 220 * t) * 32767).astype("<i2")
    with wave.open(str(tmp_path / "a.wav"), "wb") as w:
        w.setnchannels(1); w.setsampwidth(2); w.setframerate(16000); w.writeframes(pcm.tobytes())
    Image.fromarray(np.random.default_rng(0).integers(0, 255, (80, 64, 3), dtype=np.uint8)).save(tmp_path / "ref.png")
    rec = _run("audio2vid.py", "--synthetic", "--audio_path", str(tmp_path / "a.wav"), "--image_path", str(tmp_path / "ref.png"), "-W", "64",
               "-H", "64", "-L", "8", "--steps", "2", "--out_dir", str(tmp_path))
    assert rec["video"] == [1, 8, 64, 64, 3] and rec["keypoints_finite"]
    with wave.open(str(tmp_path / "b.wav"), "wb") as w:
        w.setnchannels(1); w.setsampwidth(2); w.setframerate(44100); w.writeframes(pcm.tobytes())
    r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "audio2vid.py"), "--synthetic", "--audio_path", str(tmp_path / "b.wav"),
                        "-W", "64", "-H", "64", "-L", "8", "--steps", "1", "--out_dir", str(tmp_path)], capture_output=True, text=True, cwd=ROOT)
    assert r.returncode != 0 and "resample to 16 kHz" in r.stderr


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


def test_pose2vid_from_files(tmp_path):
    """The reference's single-sample mode (scripts/pose2vid.py:196-300) on FILES: reference image (PNG, through PIL), pose clip as a
    directory of PNGs, face / lips masks as .npy stacks, hands masks as an animated GIF; random-init weights (no checkpoint exists in
    this image); the clip comes back as the reference's save_videos_grid .gif branch writes it."""
    from PIL import Image
    rng = np.random.default_rng(0)
    L = 8
    Image.fromarray(rng.integers(0, 255, (96, 80, 3), dtype=np.uint8)).save(tmp_path / "ref.png")
    os.makedirs(tmp_path / "pose")
    for i in range(L + 2):                                              # more frames than L: the clip is cut to L
        Image.fromarray(rng.integers(0, 255, (72, 72, 3), dtype=np.uint8)).save(tmp_path / "pose" / f"{i:04d}.png")
    yy, xx = np.mgrid[0:128, 0:128]
    blob = lambda cx, cy, r: (((xx - cx) ** 2 + (yy - cy) ** 2) < r * r).astype(np.uint8) * 255
    np.save(tmp_path / "face.npy", np.stack([blob(64 + i, 50, 30) for i in range(L)]))
    np.save(tmp_path / "lips.npy", np.stack([blob(64 + i, 70, 8) for i in range(L)]))
    hands = [Image.fromarray(blob(20, 100 - i, 10)) for i in range(L)]
    hands[0].save(tmp_path / "hands.gif", save_all=True, append_images=hands[1:])
    rec = _run("pose2vid.py", "--random-weights", "--image_path", str(tmp_path / "ref.png"), "--pose_path", str(tmp_path / "pose"),
               "--face_mask_path", str(tmp_path / "face.npy"), "--lips_mask_path", str(tmp_path / "lips.npy"), "--hands_mask_path",
               str(tmp_path / "hands.gif"), "-W", "64", "-H", "64", "-L", str(L), "--num_c", "8", "--steps", "2", "--out_dir", str(tmp_path))
    assert rec["video"] == [1, L, 64, 64, 3] and rec["frames"] == L and rec["weights"] == "random"
    gif = Image.open(rec["saved"])
    assert gif.n_frames == L and gif.size == (64, 64)
    assert np.asarray(gif.convert("RGB")).std() > 0


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
