"""Output path of the sampler: the counterpart of `save_videos_grid` / `save_videos_from_pil`
(src/utils/util.py:76-107,148-165).  The uint8 conversion runs on the device (AutoencoderKL.decode_video_uint8,
csrc/conditioning.hip: mmgt_frames_to_u8); this module only lays frames out and writes them.  The reference's .mp4 branch
encodes with PyAV / libx264, which is not part of this build: .gif goes through PIL exactly as the reference's .gif branch does,
.npy stores the raw uint8 frames, .mp4 raises."""
import os
from pathlib import Path

import numpy as np
import torch


def frames_uint8(videos, n_rows=6) -> np.ndarray:
    """(b, c, t, h, w) float in [0, 1] (Pose2VideoPipelineOutput.videos) or (b, t, h, w, 3) uint8 (output_type="uint8") ->
    (t, H, W, 3) uint8 grid frames laid out as torchvision.utils.make_grid(x, nrow=n_rows) does it (util.py:148-160): one clip is
    returned as it is; b > 1 clips sit in rows of min(n_rows, b) cells of (h + 2) x (w + 2) with a 2-pixel zero border."""
    v = torch.as_tensor(videos)
    if v.dtype != torch.uint8:
        v = (v.permute(0, 2, 3, 4, 1) * 255).numpy().astype(np.uint8)           # (x * 255).numpy().astype(np.uint8)
        v = torch.from_numpy(v)
    v = v.numpy()
    b, t, h, w, c = v.shape
    if b == 1:
        return v[0]
    pad = 2
    xmaps = min(int(n_rows), b)
    ymaps = -(-b // xmaps)
    ch, cw = h + pad, w + pad
    grid = np.zeros((t, ch * ymaps + pad, cw * xmaps + pad, c), dtype=np.uint8)
    for k in range(b):
        y, x = divmod(k, xmaps)
        grid[:, y * ch + pad:y * ch + pad + h, x * cw + pad:x * cw + pad + w] = v[k]
    return grid


def save_videos_grid(videos, path: str, rescale=False, n_rows=6, fps=8):
    if rescale:
        if torch.as_tensor(videos).dtype == torch.uint8:
            raise ValueError("rescale=True maps [-1, 1] floats to [0, 1]; uint8 frames are already in their final range")
        if torch.as_tensor(videos).shape[0] != 1:
            raise NotImplementedError("rescale with more than one clip also rescales make_grid's zero padding (to 127) in the "
                                      "reference; not reproduced: rescale each clip before calling")
        videos = (torch.as_tensor(videos).float() + 1.0) / 2.0
    frames = frames_uint8(videos, n_rows)
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
