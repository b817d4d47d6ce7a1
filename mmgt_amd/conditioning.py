"""Host-side producers of the operator's conditioning layout: negligible work, they only
define what `Pose2VideoPipeline.__call__` is handed.

  * process_audio_emb   scripts/pose2vid.py:72-91 / scripts/audio2vid.py:111-130  (+-2-frame clamped window stack)
  * mask_pyramid        src/dataset/image_processor.py:311-333 (+ transforms :75-102): a 64x64 "L" mask per frame ->
                        four levels (H/8/2^k)^2, ToTensor() range [0,1], flattened to (L, N_k)
  * full_mask_from_lips scripts/audio2vid.py:470-476 convention: full[k] = 1 + lips[k]
  * full_mask_with_hands scripts/pose2vid.py:266-271: clamp(1 - face + lips + hands, 0, 1) per level

The host functions below (`process_audio_emb`, `mask_pyramid`) are the original plain-torch forms.  The `*_device` functions
run the same producers on the GPU through libmmgt_hip.so (csrc/conditioning.hip):

  * blur_mask_device        scripts/pose2vid.py:94-114 (cv2.resize -> cv2.GaussianBlur -> cv2.normalize), float arithmetic with
                            cv2's rounding points; cv2 is not part of this build, so this piece is parity-unpinned
  * mask_pyramid_device     torchvision `Resize` on a PIL "L" image IS `PIL.Image.resize(BILINEAR)`: the kernel runs PIL's own
                            two-pass fixed-point resampling from the integer coefficient tables computed here exactly as
                            Pillow's Resample.c does -- bit-exact with PIL (tests/test_conditioning.py)
  * process_audio_emb_device
  * pose_frames_device      data/extract_movment_mask_all.py:319-321 `pose_vid_generator` + src/dwpose: SMGA key points -> the pose
                            frames PoseGuider reads and the face / lips / hands mask frames, drawn on the device (csrc/dwpose.hip) instead of
                            four mp4 files written and read back (scripts/audio2vid.py:386,426-441)
"""
import math
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


def full_mask_with_hands(face: List[torch.Tensor], lips: List[torch.Tensor], hands: List[torch.Tensor]) -> List[torch.Tensor]:
    """`--hands_mask_path` of scripts/pose2vid.py (:239-271): full = clamp(1 - face + lips + hands, 0, 1).  The reference indexes
    its 4-level lists with the FRAME index there and fails on any real input; this is the evident intent,
    applied per pyramid level (the hands masks go through the same blur -> pyramid producers as face and lips)."""
    return [torch.clamp(1.0 - f.float() + l.float() + h.float(), 0.0, 1.0) for f, l, h in zip(face, lips, hands)]


# ------------------------------------------------------------------------------------------------ device producers

def pil_bilinear_tables(in_size: int, out_size: int):
    """Pillow's precompute_coeffs + normalize_coeffs_8bpc (src/libImaging/Resample.c) for the BILINEAR (triangle) filter:
    per output sample the first input tap, the tap count and the integer weights with 22 fractional bits.  Python floats are
    IEEE doubles like the C code's, so the integers equal Pillow's."""
    scale = in_size / out_size
    filterscale = max(scale, 1.0)
    support = 1.0 * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    bounds, kk = [], []
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        ss = 1.0 / filterscale
        xmin = max(int(center - support + 0.5), 0)
        xmax = min(int(center + support + 0.5), in_size) - xmin
        w = []
        for x in range(xmax):
            v = abs((x + xmin - center + 0.5) * ss)
            w.append(1.0 - v if v < 1.0 else 0.0)
        ww = sum(w)
        w = [(v / ww if ww != 0.0 else v) for v in w] + [0.0] * (ksize - xmax)
        bounds.append((xmin, xmax))
        kk.append([int(0.5 + v * (1 << 22)) if v >= 0 else int(-0.5 + v * (1 << 22)) for v in w])
    return torch.tensor(bounds, dtype=torch.int32), torch.tensor(kk, dtype=torch.int32)


_TABLES = {}


def mask_pyramid_device(masks_u8: torch.Tensor, img_size: int = 512) -> List[torch.Tensor]:
    """masks_u8 (L, S, S) uint8 on the GPU, S = img_size / 8 (the blurred 64 x 64 "L" masks) -> list[4] of (L, (S / 2^k)^2)
    float32 in [0, 1]: ImageProcessor.preprocess_mov_mask (image_processor.py:311-333)."""
    from . import hip
    S = masks_u8.shape[1]
    out = []
    for k in range(4):
        D = img_size // 8 // (2 ** k)
        if D == S:
            lvl = masks_u8.to(torch.float32) / 255.0                    # Resize to the same size is the identity; ToTensor
        else:
            key = (S, D, masks_u8.device)
            if key not in _TABLES:
                b, c = pil_bilinear_tables(S, D)
                _TABLES[key] = (b.to(masks_u8.device), c.to(masks_u8.device))
            lvl = hip.resample_u8(masks_u8, D, *_TABLES[key])
        out.append(lvl.reshape(lvl.shape[0], -1))
    return out


def blur_mask_device(masks_u8: torch.Tensor, kernel_size: int) -> torch.Tensor:
    """(L, H, W) uint8 mask frames on the GPU -> (L, 64, 64) uint8: blur_mask(resize_dim=(64, 64), kernel_size) of
    scripts/pose2vid.py:94-114 (31 for the face masks, 21 for the lips: scripts/audio2vid.py:455-462)."""
    from . import hip
    return hip.blur_mask_u8(masks_u8.contiguous(), kernel_size)


def process_audio_emb_device(audio_emb: torch.Tensor) -> torch.Tensor:
    from . import hip
    return hip.window_stack(audio_emb.to(torch.float32).contiguous(), 2)


def pose_frames_device(kp_normalised: torch.Tensor, height: int = 512, width: int = 512):
    """kp_normalised (L, 402) fp32 on the GPU: SMGA's output after the seam smoothing (scripts/audio2vid.py:351-376) ->
    (pose (1, 3, L, height, width) fp32 in [0, 1] = ToTensor of the drawn frames after transforms.Resize((height, width)), :436-441;
     face_u8, lips_u8, hands_u8 (L, 512, 512) uint8 mask frames for blur_mask_device).  The reference draws at 512 x 512; other square
    sizes go through PIL's bilinear resampling per channel (what torchvision's Resize on a PIL image is), bit-exact with PIL."""
    from . import hip
    kp = kp_normalised.to(torch.float32).reshape(kp_normalised.shape[0], 134, 3).contiguous()
    pose_u8, hands, lips, face = hip.dwpose_draw(kp)
    if (height, width) == (512, 512):
        pose = pose_u8.permute(3, 0, 1, 2)[None].to(torch.float32) / 255.0
    else:
        if height != width:
            raise NotImplementedError("pose_frames_device: square frames only (the reference's configs are 512 x 512)")
        key = (512, height, pose_u8.device)
        if key not in _TABLES:
            b, c = pil_bilinear_tables(512, height)
            _TABLES[key] = (b.to(pose_u8.device), c.to(pose_u8.device))
        chans = [hip.resample_u8(pose_u8[..., ch].contiguous(), height, *_TABLES[key]) for ch in range(3)]
        pose = torch.stack(chans, 0)[None]
    return pose.contiguous(), face, lips, hands
