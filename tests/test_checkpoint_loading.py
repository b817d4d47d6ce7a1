"""`UNet3DConditionModel.from_pretrained_2d` (reference: src/models/unet_3d.py:627-718) on a synthetic on-disk checkpoint in
the layout the reference's scripts load (scripts/pose2vid.py:151-156): `<base>/unet/config.json` +
`diffusion_pytorch_model.safetensors` (the SD-1.5 2-D UNet: no motion / audio keys, fp16 like the published file) and a
motion-module `.pth`; the MM-HAA keys come from no file and keep the constructor's initialisation (strict=False)."""
import json
import os

import pytest
import torch

from mmgt_amd.unet3d import UNet3DConditionModel
from mmgt_amd.unet3d_spec import unet3d_spec

SD15_UNET_CONFIG_JSON = {  # runwayml/stable-diffusion-v1-5 unet/config.json (keys the reference's from_config receives)
    "_class_name": "UNet2DConditionModel", "_diffusers_version": "0.6.0", "act_fn": "silu", "attention_head_dim": 8,
    "block_out_channels": [320, 640, 1280, 1280], "center_input_sample": False, "cross_attention_dim": 768,
    "down_block_types": ["CrossAttnDownBlock2D", "CrossAttnDownBlock2D", "CrossAttnDownBlock2D", "DownBlock2D"],
    "downsample_padding": 1, "flip_sin_to_cos": True, "freq_shift": 0, "in_channels": 4, "layers_per_block": 2,
    "mid_block_scale_factor": 1, "norm_eps": 1e-05, "norm_num_groups": 32, "out_channels": 4, "sample_size": 64,
    "up_block_types": ["UpBlock2D", "CrossAttnUpBlock2D", "CrossAttnUpBlock2D", "CrossAttnUpBlock2D"]}

UNET_ADDITIONAL_KWARGS = dict(  # config/prompts/animation.yaml:47-75
    use_inflated_groupnorm=True, unet_use_cross_frame_attention=False, unet_use_temporal_attention=False,
    use_motion_module=True, use_audio_module=True, motion_module_resolutions=[1, 2, 4, 8], motion_module_mid_block=True,
    motion_module_decoder_only=False, motion_module_type="Vanilla",
    motion_module_kwargs=dict(num_attention_heads=8, num_transformer_block=1,
                              attention_block_types=["Temporal_Self", "Temporal_Self"], temporal_position_encoding=True,
                              temporal_position_encoding_max_len=32, temporal_attention_dim_div=1),
    audio_attention_dim=768, stack_enable_blocks_name=["up", "down", "mid"], stack_enable_blocks_depth=[0, 1, 2, 3])


def _write_config(d):
    os.makedirs(d, exist_ok=True)
    json.dump(SD15_UNET_CONFIG_JSON, open(os.path.join(d, "config.json"), "w"))


def test_from_pretrained_2d_error_paths(tmp_path):
    """unet_3d.py:645-646 (config missing), :682 (no weights file), :693-696 (unknown motion-module suffix)."""
    from safetensors.torch import save_file
    with pytest.raises(RuntimeError, match="does not exist or is not a file"):
        UNet3DConditionModel.from_pretrained_2d(tmp_path, tmp_path / "mm.pth", subfolder="unet")
    _write_config(tmp_path / "unet")
    with pytest.raises(FileNotFoundError, match="no weights file found"):
        UNet3DConditionModel.from_pretrained_2d(tmp_path, tmp_path / "mm.pth", subfolder="unet",
                                                unet_additional_kwargs=UNET_ADDITIONAL_KWARGS)
    save_file({"conv_in.bias": torch.zeros(320)}, str(tmp_path / "unet" / "diffusion_pytorch_model.safetensors"))
    (tmp_path / "mm.bin").write_bytes(b"x")
    with pytest.raises(RuntimeError, match="unknown file format for motion module weights"):
        UNet3DConditionModel.from_pretrained_2d(tmp_path, tmp_path / "mm.bin", subfolder="unet",
                                                unet_additional_kwargs=UNET_ADDITIONAL_KWARGS)


@pytest.mark.gpu
def test_from_pretrained_2d_loads_sd15_layout_and_motion_module(tmp_path):
    from safetensors.torch import save_file
    from mmgt_amd.synthetic import synth_state_dict
    from mmgt_amd.unet3d import default_init
    from oracle import unet3d_ref as R
    from tests import golden_cases as gc
    spec = unet3d_spec()
    full = synth_state_dict(spec, device="cuda:0")
    is_motion = lambda k: ".motion_modules." in k
    is_audio = lambda k: ".audio_modules." in k
    sd2d = {k: v.half().cpu().contiguous() for k, v in full.items() if not is_motion(k) and not is_audio(k)}
    sd2d["class_embedding.weight"] = torch.zeros(4, 4, dtype=torch.float16)        # an unexpected key: ignored (strict=False)
    motion = {k: v.cpu() for k, v in full.items() if is_motion(k)}
    assert len(sd2d) - 1 + len(motion) + sum(is_audio(k) for k in spec) == len(spec) == 1526
    _write_config(tmp_path / "unet")
    save_file(sd2d, str(tmp_path / "unet" / "diffusion_pytorch_model.safetensors"))
    torch.save(motion, tmp_path / "mm_sd_v15_v2.pth")
    del full

    m = UNet3DConditionModel.from_pretrained_2d(tmp_path, tmp_path / "mm_sd_v15_v2.pth", subfolder="unet",
                                                unet_additional_kwargs=UNET_ADDITIONAL_KWARGS, dtype=torch.float32)
    assert m.training and not m.gradient_checkpointing                    # from_config leaves train() mode (App. B-4)
    m.enable_gradient_checkpointing()

    # what the reference's model.load_state_dict(state_dict, strict=False) leaves behind: file values where present,
    # constructor initialisation elsewhere (zero-init zero_convs => the audio branches add nothing yet)
    merged = {k: v.float() for k, v in sd2d.items() if k in spec}
    merged.update(motion)
    missing = {k: spec[k] for k in spec if k not in merged}
    assert all(is_audio(k) for k in missing) and len(missing) > 0
    merged.update(default_init(missing))
    case = gc.UNET_CASES["full_cfg1"]
    inp = gc.unet_inputs(case)
    with torch.no_grad():
        ref = R.unet3d_forward(merged, R.UNet3DConfig(), inp["sample"], inp["timestep"], inp["ehs"], inp["audio"], inp["pose"],
                               inp["full"], inp["face"], inp["lips"], inp["motion_scale"], inp["banks"])
    mv = lambda t: t.cuda()
    m.set_banks({k: mv(v) for k, v in inp["banks"].items()})
    out = m(mv(inp["sample"]), inp["timestep"], encoder_hidden_states=mv(inp["ehs"]), audio_embedding=mv(inp["audio"]),
            pose_cond_fea=mv(inp["pose"]), full_mask=[mv(x) for x in inp["full"]], face_mask=[mv(x) for x in inp["face"]],
            body_mask=[mv(x) for x in inp["lips"]], motion_scale=inp["motion_scale"], return_dict=False)[0]
    torch.testing.assert_close(out.cpu(), ref, rtol=1e-3, atol=1e-4)

    # mm_zero_proj_out drops the motion modules' proj_out from the checkpoint => zero-initialised (unet_3d.py:697-705)
    m0 = UNet3DConditionModel.from_pretrained_2d(tmp_path, tmp_path / "mm_sd_v15_v2.pth", subfolder="unet",
                                                 unet_additional_kwargs=UNET_ADDITIONAL_KWARGS, mm_zero_proj_out=True,
                                                 dtype=torch.float32)
    assert float(m0.w["down_blocks.0.motion_modules.0.temporal_transformer.proj_out.w"].abs().sum()) == 0.0
