"""Stand-in for the absent `p_tqdm` package (the reference imports `p_map` at module level in
src/audio2pose_model/diffusion.py:11 but never calls it on the sampling path).  Build container only; never shipped."""


def p_map(fn, *iterables, **kw):
    return list(map(fn, *iterables))
