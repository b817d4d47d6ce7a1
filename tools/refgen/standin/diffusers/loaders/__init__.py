class UNet2DConditionLoadersMixin:
    pass
