"""Output path of the sampler: the counterpart of `save_videos_grid` / `save_videos_from_pil`
(src/utils/util.py:76-107,148-165).  The uint8 conversion runs on the device (AutoencoderKL.decode_video_uint8,
csrc/conditioning.hip: mmgt_frames_to_u8); this module only lays frames out and writes them.  The reference's .mp4 branch
encodes with PyAV / libx264, which is not part of this build: .gif goes through PIL exactly as the reference's .gif branch does,
.npy stores the raw uint8 frames, .mp4 raises."""
import os
from pathlib import Path

import numpy as np
import torch


def frames_uint8(videos) -> np.ndarray:
    """(b, c, t, h, w) float in [0, 1] (Pose2VideoPipelineOutput.videos) or (b, t, h, w, 3) uint8 (output_type="uint8") ->
    (t, h, b * w, 3) uint8, clips side by side like make_grid(nrow = b) without padding for b == 1 (util.py:148-160)."""
    v = torch.as_tensor(videos)
    if v.dtype != torch.uint8:
        v = (v.permute(0, 2, 3, 4, 1) * 255).numpy().astype(np.uint8)           # (x * 255).numpy().astype(np.uint8)
        v = torch.from_numpy(v)
    return torch.cat(list(v), dim=2).numpy()


def save_videos_grid(videos, path: str, rescale=False, n_rows=6, fps=8):
    if rescale:
        videos = (torch.as_tensor(videos).float() + 1.0) / 2.0
    frames = frames_uint8(videos)
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    fmt = Path(path).suffix
    if fmt == ".gif":
        from PIL import Image
        pil = [Image.fromarray(f) for f in frames]
        pil[0].save(fp=path, format="GIF", append_images=pil[1:], save_all=True, duration=(1 / fps * 1000), loop=0)
    elif fmt == ".npy":
        np.save(path, frames)
    elif fmt == ".mp4":
        raise RuntimeError("mp4 output needs PyAV / libx264 (src/utils/util.py:83-97), which this build does not include: "
                           "write .gif or .npy, or hand frames_uint8() to your encoder")
    else:
        raise ValueError("Unsupported file type. Use .mp4 or .gif.")
