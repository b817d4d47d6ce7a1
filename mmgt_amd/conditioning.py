"""Host-side producers of the operator's conditioning layout: negligible work, they only
define what `Pose2VideoPipeline.__call__` is handed.

  * process_audio_emb   scripts/pose2vid.py:72-91 / scripts/audio2vid.py:111-130  (+-2-frame clamped window stack)
  * mask_pyramid        src/dataset/image_processor.py:311-333 (+ transforms :75-102): a 64x64 "L" mask per frame ->
                        four levels (H/8/2^k)^2, ToTensor() range [0,1], flattened to (L, N_k)
  * full_mask_from_lips scripts/audio2vid.py:470-476 convention: full[k] = 1 + lips[k]

The reference resizes PIL images with torchvision's `Resize` (bilinear, antialias per torchvision version) and blurs with
cv2; neither library is part of this build, so the resize is restated with `F.interpolate(mode="bilinear",
antialias=True)` (PIL-equivalent for down-scaling) and is NOT pinned against the reference's image stack.
"""
from typing import List

import torch
import torch.nn.functional as F


def process_audio_emb(audio_emb: torch.Tensor) -> torch.Tensor:
    """(L, ...) -> (L, 5, ...): frame i gets frames clamp(i-2 .. i+2, 0, L-1)."""
    n = audio_emb.shape[0]
    idx = (torch.arange(n)[:, None] + torch.arange(-2, 3)[None, :]).clamp_(0, n - 1)
    return audio_emb[idx]


def mask_pyramid(masks: torch.Tensor, img_size: int = 512) -> List[torch.Tensor]:
    """masks (L, h, w) uint8 or float in [0, 255] -> list[4] of (L, (img_size/8/2^k)^2) float32 in [0, 1]."""
    m = masks.to(torch.float32)
    if masks.dtype == torch.uint8 or m.max() > 1.0:
        m = m / 255.0                                            # ToTensor()
    m = m[:, None]
    out = []
    for k in range(4):
        s = img_size // 8 // (2 ** k)
        r = m if m.shape[-2:] == (s, s) else F.interpolate(m, size=(s, s), mode="bilinear", antialias=True,
                                                            align_corners=False)
        out.append(r.reshape(r.shape[0], -1).clamp_(0, 1))
    return out


def full_mask_from_lips(lips: List[torch.Tensor]) -> List[torch.Tensor]:
    return [1 + l for l in lips]
