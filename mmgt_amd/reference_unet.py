"""ReferenceNet on MI355X: the reference's UNet2DConditionModel (src/models/unet_2d_condition.py:872-1308) run once
per clip in "write" mode, plus the ReferenceAttentionControl hand-off (src/models/mutual_self_attention.py:19-365).

It is the SD-1.5 UNet2D with conv_norm_out / conv_out removed (:645-653,1296-1299) = the UNet3D wiring at one frame
without motion and audio modules, so it reuses the UNet3D building blocks and HIP kernels; same 682 state-dict keys.
"""
from typing import Dict

import torch

from . import hip
from .unet3d import UNet3DConditionModel
from .unet3d_spec import unet2d_reference_spec


class UNet2DConditionOutput:
    def __init__(self, sample):
        self.sample = sample


class UNet2DConditionModel(UNet3DConditionModel):
    def __init__(self, device="cuda", dtype=torch.bfloat16, **config):
        super().__init__(device=device, dtype=dtype, **config)
        self.spec = unet2d_reference_spec(self.boc, self.config.cross_attention_dim, self.in_channels,
                                          self.config.layers_per_block)
        self.training = False                      # from_pretrained(...) leaves it in eval() (scripts/pose2vid.py:146-148)
        self.bank: Dict[str, torch.Tensor] = {}     # filled by forward(): {reader prefix: (b, N, C) fp32}

    def forward(self, sample, timestep, encoder_hidden_states, return_dict: bool = True, **unused):
        """sample (b, 4, h, w) -> (b, 320, h, w) (the output head is disabled in the reference); side effect: self.bank."""
        if not self._loaded:
            raise RuntimeError("UNet2DConditionModel.forward before load_state_dict")
        if not sample.is_cuda:
            raise RuntimeError("mmgt_amd.UNet2DConditionModel runs on the GPU only (no CPU path exists)")
        b, _, hh, ww = sample.shape
        lpb = self.config.layers_per_block
        temb = self._time_embedding(timestep, b)
        x = hip.ncfhw_to_nhwc(sample.to(torch.float32)[:, :, None].contiguous(), 64, self._dtype)   # f = 1
        x = hip.conv3x3(x, self.w["conv_in.w"], self.w["conv_in.bias"])
        ehs = encoder_hidden_states.to(self._device)
        bank: Dict[str, torch.Tensor] = {}
        skips = [x]
        for i in range(4):
            p = f"down_blocks.{i}"
            for j in range(lpb):
                x = self._resnet(f"{p}.resnets.{j}", x, temb)
                if i < 3:
                    x = self._spatial_transformer(f"{p}.attentions.{j}", x, ehs, 1, write=bank)
                skips.append(x)
            if i != 3:
                x = hip.conv3x3(x, self.w[f"{p}.downsamplers.0.conv.w"], self.w[f"{p}.downsamplers.0.conv.bias"], stride=2)
                skips.append(x)
        x = self._resnet("mid_block.resnets.0", x, temb)
        mid: Dict[str, torch.Tensor] = {}
        x = self._spatial_transformer("mid_block.attentions.0", x, ehs, 1, write=mid)
        x = self._resnet("mid_block.resnets.1", x, temb)
        for i in range(4):
            p = f"up_blocks.{i}"
            for j in range(lpb + 1):
                x = self._resnet(f"{p}.resnets.{j}", x, temb, skip=skips.pop())
                if i > 0:
                    x = self._spatial_transformer(f"{p}.attentions.{j}", x, ehs, 1, write=bank)
            if i != 3:
                w2 = self.w.get(f"{p}.upsamplers.0.conv.w2")          # (the four-phase form: unet3d.py `_pack`)
                if w2 is not None:
                    x = hip.conv3x3(x, w2, self.w[f"{p}.upsamplers.0.conv.bias"], upsample=2)
                else:
                    x = hip.conv3x3(x, self.w[f"{p}.upsamplers.0.conv.w"], self.w[f"{p}.upsamplers.0.conv.bias"], upsample=True)
        bank.update(mid)                              # module order down -> up -> mid
        self.bank = bank
        out = hip.nhwc_to_ncfhw(x, b, self.boc[0])[:, :, 0].to(sample.dtype)
        return UNet2DConditionOutput(out) if return_dict else (out,)

    __call__ = forward

    def write_banks(self, latents, timestep, encoder_hidden_states):
        self.forward(latents, timestep, encoder_hidden_states, return_dict=False)
        return self.bank

    def denoise_window(self, *a, **k):
        raise NotImplementedError("ReferenceNet has no denoise step")


class ReferenceAttentionControl:
    """src/models/mutual_self_attention.py:19-365.  The writer is the ReferenceNet, the reader the denoising UNet3D;
    `update` pairs their transformer blocks the way the reference does (stable sort on descending channel width over the
    module order down -> up -> mid) and hands the banks over rounded through `dtype` (fp16 by default, :304,340)."""

    def __init__(self, unet, mode="write", do_classifier_free_guidance=False, attention_auto_machine_weight=float("inf"),
                 gn_auto_machine_weight=1.0, style_fidelity=1.0, reference_attn=True, reference_adain=False,
                 fusion_blocks="midup", batch_size=1):
        assert mode in ("read", "write") and fusion_blocks in ("midup", "full")
        if reference_adain:
            raise NotImplementedError("reference_adain is not used by the reference scripts")
        self.unet, self.mode = unet, mode
        self.reference_attn = reference_attn
        self.fusion_blocks = fusion_blocks
        self.do_classifier_free_guidance = do_classifier_free_guidance

    def _modules(self, unet):
        keys = unet.bank_keys()                       # module order down -> up -> mid
        if self.fusion_blocks == "midup":
            keys = [k for k in keys if not k.startswith("down_blocks")]
            keys = [k for k in keys if k.startswith("mid_block")] + [k for k in keys if k.startswith("up_blocks")]
        width = lambda k: unet.spec[k + ".transformer_blocks.0.norm1.weight"][0]
        return sorted(keys, key=lambda k: -width(k))   # Python's sort is stable, like the reference's

    def update(self, writer, dtype=torch.float16):
        if not self.reference_attn:
            return
        readers, writers = self._modules(self.unet), self._modules(writer.unet)
        banks = {}
        for r, w in zip(readers, writers):
            if w in writer.unet.bank:
                banks[r] = writer.unet.bank[w].to(dtype)
        self.unet.bank_fp16_roundtrip = False          # rounding already applied with the caller's dtype
        self.unet.set_banks(banks)

    def clear(self):
        if self.mode == "read":
            self.unet.clear_banks()
        else:
            self.unet.bank = {}
