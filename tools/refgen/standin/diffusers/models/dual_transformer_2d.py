class DualTransformer2DModel:  # dual_cross_attention is False on this path
    def __init__(self, *a, **k):
        raise NotImplementedError
