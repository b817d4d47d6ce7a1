"""Import the reference's own src.models against the diffusers stand-in (build container only)."""
import os
import sys

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"


def setup():
    if not os.path.isdir(REF):
        raise RuntimeError("reference checkout not present: goldens can only be generated in the build container")
    for p in (REF, os.path.join(HERE, "standin")):
        if p not in sys.path:
            sys.path.insert(0, p)


SD15_UNET_CONFIG = dict(
    sample_size=64, in_channels=4, out_channels=4, center_input_sample=False, flip_sin_to_cos=True, freq_shift=0,
    down_block_types=["CrossAttnDownBlock3D", "CrossAttnDownBlock3D", "CrossAttnDownBlock3D", "DownBlock3D"],
    mid_block_type="UNetMidBlock3DCrossAttn",
    up_block_types=["UpBlock3D", "CrossAttnUpBlock3D", "CrossAttnUpBlock3D", "CrossAttnUpBlock3D"],
    block_out_channels=[320, 640, 1280, 1280], layers_per_block=2, downsample_padding=1, mid_block_scale_factor=1,
    act_fn="silu", norm_num_groups=32, norm_eps=1e-5, cross_attention_dim=768, attention_head_dim=8,
)

UNET_ADDITIONAL_KWARGS = dict(  # config/prompts/animation.yaml:47-75
    use_inflated_groupnorm=True, unet_use_cross_frame_attention=False, unet_use_temporal_attention=False,
    use_motion_module=True, use_audio_module=True, motion_module_resolutions=[1, 2, 4, 8],
    motion_module_mid_block=True, motion_module_decoder_only=False, motion_module_type="Vanilla",
    motion_module_kwargs=dict(num_attention_heads=8, num_transformer_block=1,
                              attention_block_types=["Temporal_Self", "Temporal_Self"],
                              temporal_position_encoding=True, temporal_position_encoding_max_len=32,
                              temporal_attention_dim_div=1),
    audio_attention_dim=768, stack_enable_blocks_name=["up", "down", "mid"], stack_enable_blocks_depth=[0, 1, 2, 3],
)


def build_reference_unet3d(**overrides):
    setup()
    from src.models.unet_3d import UNet3DConditionModel
    cfg = dict(SD15_UNET_CONFIG)
    add = dict(UNET_ADDITIONAL_KWARGS)
    for k, v in overrides.items():
        (add if k in add else cfg)[k] = v
    return UNet3DConditionModel.from_config(cfg, **add)
