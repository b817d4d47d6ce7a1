"""ORACLE (test infrastructure): CPU restatements of the conditioning producers and the output conversion.

  * mask_pyramid_pil: ImageProcessor.preprocess_mov_mask (src/dataset/image_processor.py:75-102,311-333) with the very library the
    reference reaches through torchvision: `transforms.Resize` on a PIL image is `PIL.Image.resize(size, BILINEAR)`, `ToTensor`
    is / 255.  PINNED: this is the reference's own arithmetic (Pillow is installed here and on the GPU box).
  * blur_mask_ref: scripts/pose2vid.py:94-114 restated in numpy float arithmetic (cv2.resize INTER_LINEAR -> GaussianBlur with
    sigma = 0.3 ((k - 1) / 2 - 1) + 0.8, BORDER_REFLECT_101 -> NORM_MINMAX to 0..255).  cv2 is absent: PARITY UNPINNED; OpenCV's
    8-bit paths use fixed-point coefficients, so single-ulp (1 / 255) differences from the real cv2 are expected.
  * frames_to_uint8_ref: decode_latents' (x / 2 + 0.5).clamp(0, 1) (pipeline_pose2vid_long.py:121-123) followed by
    save_videos_grid's (x * 255).numpy().astype(np.uint8) (src/utils/util.py:148-160).
Only tests/ may import this file.
"""
import numpy as np
import torch
from PIL import Image


def mask_pyramid_pil(masks_u8: np.ndarray, img_size: int = 512):
    """masks_u8 (L, S, S) uint8 -> list[4] of float32 (L, (img_size/8/2^k)^2)."""
    out = []
    for k in range(4):
        d = img_size // 8 // (2 ** k)
        lv = [np.asarray(Image.fromarray(m, mode="L").resize((d, d), Image.BILINEAR), dtype=np.float32) / 255.0 for m in masks_u8]
        out.append(torch.from_numpy(np.stack(lv)).reshape(len(masks_u8), -1))
    return out


def blur_mask_ref(mask: np.ndarray, ksize: int, out: int = 64):
    h, w = mask.shape
    sx, sy = w / out, h / out
    xs = (np.arange(out) + 0.5) * sx - 0.5
    ys = (np.arange(out) + 0.5) * sy - 0.5

    def taps(f, n):
        i0 = np.floor(f).astype(int)
        a = f - i0
        lo = i0 < 0
        i0[lo], a[lo] = 0, 0.0
        hi = i0 >= n - 1
        i0[hi], a[hi] = max(n - 2, 0), 1.0 if n > 1 else 0.0
        return i0, np.minimum(i0 + 1, n - 1), a.astype(np.float32)
    x0, x1, ax = taps(xs.astype(np.float32), w)
    y0, y1, ay = taps(ys.astype(np.float32), h)
    m = mask.astype(np.float32)
    top = m[y0][:, x0] * (1 - ax)[None] + m[y0][:, x1] * ax[None]
    bot = m[y1][:, x0] * (1 - ax)[None] + m[y1][:, x1] * ax[None]
    r = np.rint(top * (1 - ay)[:, None] + bot * ay[:, None]).astype(np.float32)
    sigma = 0.3 * ((ksize - 1) * 0.5 - 1) + 0.8
    rad = ksize // 2
    g = np.exp(-0.5 * (np.arange(ksize) - rad) ** 2 / sigma ** 2).astype(np.float32)
    g /= g.sum()
    idx = np.arange(-rad, out + rad)
    idx = np.where(idx < 0, -idx, idx)
    idx = np.where(idx >= out, 2 * out - 2 - idx, idx)
    hp = np.stack([(r[:, idx[k:k + out]] * g[k]) for k in range(ksize)]).sum(0)
    vp = np.stack([(hp[idx[k:k + out], :] * g[k]) for k in range(ksize)]).sum(0)
    b = np.clip(np.rint(vp), 0, 255)
    mn, mx = b.min(), b.max()
    sc = 255.0 / (mx - mn) if mx > mn else 0.0
    return np.clip(np.rint((b - mn) * sc), 0, 255).astype(np.uint8)


def frames_to_uint8_ref(x: torch.Tensor):
    """x (N, H, W, 3) decoder output in [-1, 1] -> uint8."""
    return ((x.float() / 2 + 0.5).clamp(0, 1) * 255).numpy().astype(np.uint8)
