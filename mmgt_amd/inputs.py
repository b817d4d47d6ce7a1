"""Caller-side inputs of the sampler from FILES (the callers either side of the hot path): the counterpart of
what `scripts/pose2vid.py:196-271` does between `read_frames(...)` and the pipeline call, with the per-frame mask arithmetic on the
device (mmgt_amd/conditioning.py) instead of cv2 / PIL on the host.

  read_frames        frames of a clip as PIL images (reference: src/utils/util.py read_frames, PyAV).  Containers that need a video
                     decoder (.mp4, .avi, .mov, .mkv, .webm) raise: PyAV / cv2 are not part of this build; a directory of images,
                     a .npy stack or an animated .gif / .png / .webp is read instead.
  pose_tensor        transforms.Resize((H, W)) + ToTensor of the pose frames -> (1, 3, L, H, W) float in [0, 1]
                     (scripts/pose2vid.py:231-236)
  motion_masks       face / lips / hands mask frames -> blur_mask (resize 64 x 64, Gaussian 31 / 21 / 21, min-max normalise) ->
                     the 64 / 32 / 16 / 8 pyramid -> full = clamp(1 - face + lips + hands, 0, 1) per level (:239-271)
  load_checkpoint    a state dict from .safetensors / .pth / .pt / .bin / .ckpt or a diffusers-style directory
  split_net_checkpoint   the reference's `Net` checkpoint (net-<num_c>.pth, scripts/pose2vid.py:41-67,186-190) -> per-module dicts
"""
import os
from pathlib import Path
from typing import Dict, List, Optional, Sequence

import numpy as np
import torch

VIDEO_CONTAINERS = {".mp4", ".avi", ".mov", ".mkv", ".webm", ".m4v"}
IMAGE_SUFFIXES = {".png", ".jpg", ".jpeg", ".bmp", ".webp", ".tif", ".tiff"}


def read_frames(path, limit: Optional[int] = None) -> list:
    """PIL frames of `path`: a directory of images (sorted by name), a .npy stack (L, H, W[, C]) uint8, an animated image
    (.gif / .png / .webp) or one still image."""
    from PIL import Image, ImageSequence
    p = Path(path)
    if not p.exists():
        raise FileNotFoundError(f"read_frames: {p} does not exist")
    if p.is_dir():
        files = sorted(f for f in p.iterdir() if f.is_file() and f.suffix.lower() in IMAGE_SUFFIXES)
        if not files:
            raise RuntimeError(f"read_frames: no image files ({', '.join(sorted(IMAGE_SUFFIXES))}) in {p}")
        return [Image.open(f).copy() for f in (files if limit is None else files[:limit])]
    suf = p.suffix.lower()
    if suf in VIDEO_CONTAINERS:
        raise RuntimeError(f"read_frames: {p.name} needs a video decoder (PyAV / cv2: src/utils/util.py read_frames), which this build "
                           f"does not include -- extract the frames into a directory of images or a .npy stack and pass that instead")
    if suf == ".npy":
        arr = np.load(p)
        if arr.dtype != np.uint8 or arr.ndim not in (3, 4):
            raise RuntimeError(f"read_frames: {p.name} must hold uint8 frames (L, H, W) or (L, H, W, C), got {arr.dtype} {arr.shape}")
        arr = arr if limit is None else arr[:limit]
        return [Image.fromarray(f) for f in arr]
    img = Image.open(p)
    frames = [f.copy() for f in ImageSequence.Iterator(img)]
    return frames if limit is None else frames[:limit]


def pose_tensor(frames: Sequence, width: int, height: int) -> torch.Tensor:
    """(1, 3, L, H, W) float32 in [0, 1]: torchvision's Resize((H, W)) (PIL bilinear with antialias) + ToTensor per frame."""
    from PIL import Image
    out = []
    for f in frames:
        f = f.convert("RGB")
        if f.size != (width, height):
            f = f.resize((width, height), Image.BILINEAR)
        out.append(torch.from_numpy(np.asarray(f, dtype=np.uint8).copy()))
    x = torch.stack(out).permute(3, 0, 1, 2).float().div_(255.0)      # (3, L, H, W)
    return x.unsqueeze(0)


def _mask_stack(frames: Sequence, length: int) -> torch.Tensor:
    """Mask frames -> (L, h, w) uint8 (single channel: the reference blurs the array as read and converts to "L" afterwards; mask
    videos are grey, so the first channel is the mask)."""
    arrs = []
    for f in frames[:length]:
        a = np.asarray(f)
        arrs.append(torch.from_numpy((a if a.ndim == 2 else a[..., 0]).astype(np.uint8).copy()))
    return torch.stack(arrs)


def motion_masks(face_frames: Sequence, lips_frames: Sequence, hands_frames: Optional[Sequence], length: int, device,
                 img_size: int = 512):
    """-> (full, face, lips): three lists of four (L, (64 / 2^k)^2) float tensors on the CPU, as the pipeline takes them."""
    from . import conditioning as C

    def pyramid(frames, ksize):
        u8 = _mask_stack(frames, length).to(device).contiguous()
        return [m.cpu() for m in C.mask_pyramid_device(C.blur_mask_device(u8, ksize), img_size)]
    face = pyramid(face_frames, 31)
    lips = pyramid(lips_frames, 21)
    if hands_frames is not None:
        hands = pyramid(hands_frames, 21)
    else:
        hands = [torch.zeros_like(m) for m in lips]                   # `Image.new("L", (64, 64), 0)` (:252)
    return C.full_mask_with_hands(face, lips, hands), face, lips


def load_checkpoint(path) -> Dict[str, torch.Tensor]:
    """State dict of a checkpoint file, or of a diffusers / transformers-style directory (first of diffusion_pytorch_model.safetensors,
    model.safetensors, diffusion_pytorch_model.bin, pytorch_model.bin)."""
    p = Path(path)
    if p.is_dir():
        for name in ("diffusion_pytorch_model.safetensors", "model.safetensors", "diffusion_pytorch_model.bin", "pytorch_model.bin"):
            if (p / name).is_file():
                p = p / name
                break
        else:
            raise FileNotFoundError(f"load_checkpoint: no weights file found in {p}")
    if not p.is_file():
        raise FileNotFoundError(f"load_checkpoint: {p} does not exist")
    if p.suffix == ".safetensors":
        from safetensors.torch import load_file
        return load_file(str(p), device="cpu")
    if p.suffix in (".pth", ".pt", ".bin", ".ckpt"):
        sd = torch.load(str(p), map_location="cpu", weights_only=True)
        return sd.get("state_dict", sd) if isinstance(sd, dict) else sd
    raise RuntimeError(f"load_checkpoint: unknown file format {p.suffix!r} ({p})")


NET_PREFIXES = ("reference_unet", "denoising_unet", "pose_guider", "audioproj")


def split_net_checkpoint(sd: Dict[str, torch.Tensor]) -> Dict[str, Dict[str, torch.Tensor]]:
    """The reference trains and stores its modules as one `Net` (scripts/pose2vid.py:41-67): keys `<module>.<key>`."""
    out = {p: {} for p in NET_PREFIXES}
    for k, v in sd.items():
        head, _, rest = k.partition(".")
        if head not in out:
            raise RuntimeError(f"split_net_checkpoint: unexpected key {k!r} (expected one of {NET_PREFIXES} as the first component)")
        out[head][rest] = v
    return out
