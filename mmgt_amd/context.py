"""Temporal sliding-window schedule of the long-video sampler (pure integers).

Host-side mirror of src/pipelines/context.py:7-49 (`uniform`, `ordered_halving`, `get_context_scheduler`): closed-loop
windows of `context_size` frames, stride context_size * 2^k - overlap, wrapping modulo the clip length.
"""
from typing import Callable, Iterator, List


def ordered_halving(val: int) -> float:
    """Bit-reversal of a 64-bit integer as a fraction in [0, 1)."""
    rev = 0
    v = int(val) & ((1 << 64) - 1)
    for _ in range(64):
        rev = (rev << 1) | (v & 1)
        v >>= 1
    return rev / (1 << 64)


def _ceil_log2_ratio(num: int, den: int) -> int:
    """ceil(log2(num / den)) for num > den > 0 in exact integer arithmetic."""
    k = 0
    while den << k < num:
        k += 1
    return k


def uniform(step: int, num_steps, num_frames: int, context_size: int, context_stride: int = 3,
            context_overlap: int = 4, closed_loop: bool = True) -> Iterator[List[int]]:
    if num_frames <= context_size:
        yield list(range(num_frames))
        return
    levels = min(context_stride, _ceil_log2_ratio(num_frames, context_size) + 1)
    frac = ordered_halving(step)
    pad = int(round(num_frames * frac))
    for level in range(levels):
        cstep = 1 << level
        stop = num_frames + pad + (0 if closed_loop else -context_overlap)
        for j in range(int(frac * cstep) + pad, stop, context_size * cstep - context_overlap):
            yield [e % num_frames for e in range(j, j + context_size * cstep, cstep)]


def get_context_scheduler(name: str) -> Callable:
    if name == "uniform":
        return uniform
    raise ValueError(f"Unknown context_overlap policy {name}")
