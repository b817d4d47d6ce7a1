"""Frame interpolation of the sampler's latents (the reference's `interpolate_latents`, pipeline_pose2vid_long.py:292-335,
with the blends of src/pipelines/utils.py:15-29), batched: every in-between frame of every neighbouring pair is produced by
one tensor expression on the device the latents live on -- the reference walks the pairs and rates in Python loops.

Only interpolation_factor >= 2 reaches this module, which no reference script sets; it is host-side tensor plumbing around
the sampler, not one of the kernels.
"""
import torch

_SPHERICAL = False
_PARALLEL_DOT = 0.9995      # beyond this |cos| the spherical blend degenerates and falls back to the straight line


def set_tensor_interpolation_method(is_slerp):
    global _SPHERICAL
    _SPHERICAL = bool(is_slerp)


def get_tensor_interpolation_method():
    return slerp if _SPHERICAL else linear


def blend_pairs(a, b, rates, spherical=False):
    """a, b: (P, ...) stacks of P tensor pairs; rates: (R,) blend positions in (0, 1).  Returns (P, R, ...): for every pair
    and rate the straight-line blend, or the great-circle blend with the pair's own angle (each pair taken as ONE flat
    vector, as the reference does per frame)."""
    P = a.shape[0]
    r = torch.as_tensor(rates, dtype=a.dtype, device=a.device).view(1, -1, *([1] * (a.dim() - 1)))
    a_, b_ = a.unsqueeze(1), b.unsqueeze(1)
    line = (1.0 - r) * a_ + r * b_
    if not spherical:
        return line
    fa, fb = a.reshape(P, -1), b.reshape(P, -1)
    cos = ((fa / fa.norm(dim=1, keepdim=True)) * (fb / fb.norm(dim=1, keepdim=True))).sum(dim=1)
    ang = cos.clamp(-1.0, 1.0).acos().view(P, 1, *([1] * (a.dim() - 1)))
    arc = (((1.0 - r) * ang).sin() * a_ + (r * ang).sin() * b_) / ang.sin()
    near = (cos.abs() > _PARALLEL_DOT).view(P, 1, *([1] * (a.dim() - 1)))
    return torch.where(near, line, arc)


def linear(v1, v2, t):
    return blend_pairs(v1[None], v2[None], [t])[0, 0]


def slerp(v0, v1, t, DOT_THRESHOLD=_PARALLEL_DOT):
    if DOT_THRESHOLD != _PARALLEL_DOT:
        raise NotImplementedError("only the reference's default threshold is built")
    return blend_pairs(v0[None], v1[None], [t], spherical=True)[0, 0]


def interpolate_frames(latents, factor):
    """(b, c, f, h, w) -> (b, c, (f - 1) * factor + 1, h, w): factor - 1 blended frames between neighbours, originals kept."""
    if factor < 2:
        return latents
    b, c, f, h, w = latents.shape
    fr = latents.permute(2, 0, 1, 3, 4)                              # (f, b, c, h, w): a frame is one flat vector per pair
    rates = [i / factor for i in range(1, factor)]
    mid = blend_pairs(fr[:-1], fr[1:], rates, spherical=_SPHERICAL)  # (f - 1, factor - 1, b, c, h, w)
    seq = torch.cat([fr[:-1].unsqueeze(1), mid], dim=1).reshape((f - 1) * factor, b, c, h, w)
    return torch.cat([seq, fr[-1:]], dim=0).permute(1, 2, 0, 3, 4).contiguous()
