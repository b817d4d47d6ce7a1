"""ORACLE (test infrastructure): literal restatement of src/pipelines/context.py:7-49 (window index lists)."""
import numpy as np


def ordered_halving(val):            # context.py:7-12
    bin_str = f"{val:064b}"
    return int(bin_str[::-1], 2) / (1 << 64)


def uniform(step, num_steps, num_frames, context_size, context_stride=3, context_overlap=4, closed_loop=True):
    """context.py:15-42."""
    if num_frames <= context_size:
        yield list(range(num_frames))
        return
    context_stride = min(context_stride, int(np.ceil(np.log2(num_frames / context_size))) + 1)
    for context_step in 1 << np.arange(context_stride):
        pad = int(round(num_frames * ordered_halving(step)))
        for j in range(int(ordered_halving(step) * context_step) + pad,
                       num_frames + pad + (0 if closed_loop else -context_overlap),
                       (context_size * context_step - context_overlap)):
            yield [e % num_frames for e in range(j, j + context_size * context_step, context_step)]
