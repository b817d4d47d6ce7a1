from torch import nn


class AdaLayerNormSingle(nn.Module):  # PixArt only; never built on this path
    def __init__(self, *a, **k):
        raise NotImplementedError
