"""UNet3DConditionModel on MI355X: the denoise-step operator of MMGT's Stage-2 pipeline.

Host-side mirror of the reference class (src/models/unet_3d.py:33-718): same constructor config, same
`forward(sample, timestep, encoder_hidden_states, audio_embedding, ..., pose_cond_fea, full_mask, face_mask, body_mask,
motion_scale, return_dict)` signature and semantics, same state-dict key names.  All arithmetic runs in the HIP kernels
of libmmgt_hip.so (mmgt_amd/hip.py); this file only wires them: activations stay channels-last ((b f), h, w, c) from
conv_in to conv_out, so none of the reference's rearrange/contiguous round trips exist here.

Work the reference does and this implementation does not (results identical):
  * bank K/V are projected once per clip (set_banks), not per frame and step (mutual_self_attention.py:150-165);
  * the unconditional CFG half never attends to the bank (the reference computes it and overwrites it, :160-188);
  * cross-attention to the single CLIP token is softmax over one key == 1, so it collapses to a per-CFG-row vector
    to_out(to_v(e)) added in the attn1 output epilogue (attention.py:455-462).
"""
import math
from dataclasses import dataclass
from typing import Dict, List, Optional, Union

import torch

from . import hip
from .packing import pack_conv3x3, pack_conv3x3_up2, pack_conv_taps, pack_ff_fused, pack_ff_proj_out, pack_geglu, pack_rconv, pack_rowgemm, pack_tleg, pad_cols, pad_rows, round_up
from .unet3d_spec import unet3d_spec

SD15_CONFIG = dict(  # SD-1.5 unet/config.json + unet_3d.py:649-662 + config/prompts/animation.yaml:47-75
    in_channels=4, out_channels=4, block_out_channels=(320, 640, 1280, 1280), layers_per_block=2, attention_head_dim=8,
    cross_attention_dim=768, audio_attention_dim=768, norm_num_groups=32, norm_eps=1e-5, center_input_sample=False,
    use_motion_module=True, use_audio_module=True, use_inflated_groupnorm=True,
    motion_module_kwargs=dict(num_attention_heads=8, temporal_position_encoding_max_len=32),
)


@dataclass
class UNet3DConditionOutput:
    sample: torch.Tensor

    def __getitem__(self, i):
        return (self.sample,)[i]


class _Config(dict):
    __getattr__ = dict.__getitem__


def default_init(spec, seed=0):
    """The reference constructor's initialisation for keys no checkpoint provides: nn.Linear / nn.Conv2d default
    (kaiming-uniform(a=sqrt 5) == U(+-1/sqrt(fan_in)) for weight and bias), norm affine (1, 0), zero_module for the
    MM-HAA zero-convs and the motion modules' proj_out (attention.py:556-566, motion_module.py:72-75), sinusoid `pe`."""
    from .synthetic import sinusoid_pe
    g = torch.Generator().manual_seed(seed)
    out = {}
    fan = {}
    for k, shape in spec.items():
        if k.endswith(".weight") and len(shape) > 1:
            f = 1
            for s in shape[1:]:
                f *= s
            fan[k[:-len(".weight")]] = f
    for k, shape in spec.items():
        leaf = k.rsplit(".", 1)[-1]
        base = k[:-len(leaf) - 1]
        if leaf == "pe":
            out[k] = sinusoid_pe(shape[1], shape[2])
        elif any(z in k for z in ("zero_conv_", "temporal_transformer.proj_out")):
            out[k] = torch.zeros(shape)
        elif len(shape) == 1 and base not in fan:                       # norm affine
            out[k] = torch.ones(shape) if leaf == "weight" else torch.zeros(shape)
        else:
            bound = 1.0 / math.sqrt(fan.get(base, shape[0]))
            out[k] = (torch.rand(shape, generator=g) * 2 - 1) * bound
    return out


def mask_bias_columns(rs, ncols, dtype):
    """Operand columns of MM-HAA's merged out-projection that meet the merged biases in the weight image (`.oz3.w` / `.oz3.wb`):
    rs (3, m) fp32 = mask_i * motion_scale_i per token -> (m, ncols) `dtype`, per branch [m_hi | m_hi | m_lo] against the image's
    [b_hi | b_lo | b_hi], so that the product keeps fp32 accuracy in bf16 storage (lo x lo, 2^-16 relative, dropped)."""
    cols = torch.zeros((rs.shape[1], ncols), device=rs.device, dtype=dtype)
    rs_hi = rs.to(dtype)
    rs_lo = (rs - rs_hi.float()).to(dtype)
    for i in range(3):
        cols[:, 3 * i], cols[:, 3 * i + 1], cols[:, 3 * i + 2] = rs_hi[i], rs_hi[i], rs_lo[i]
    return cols


class UNet3DConditionModel:
    _supports_gradient_checkpointing = True

    def __init__(self, device="cuda", dtype=torch.bfloat16, **config):
        cfg = dict(SD15_CONFIG)
        unknown = set(config) - set(cfg) - {"sample_size", "motion_module_type", "motion_module_resolutions",
                                            "motion_module_mid_block", "motion_module_decoder_only",
                                            "unet_use_cross_frame_attention", "unet_use_temporal_attention",
                                            "stack_enable_blocks_name", "stack_enable_blocks_depth", "flip_sin_to_cos",
                                            "freq_shift", "down_block_types", "up_block_types", "mid_block_type",
                                            "act_fn", "downsample_padding", "mid_block_scale_factor", "_class_name",
                                            "only_cross_attention", "dual_cross_attention", "use_linear_projection",
                                            "class_embed_type", "num_class_embeds", "upcast_attention",
                                            "resnet_time_scale_shift", "task_type", "mode", "_diffusers_version"}
        if unknown:
            raise ValueError(f"UNet3DConditionModel: unsupported config keys {sorted(unknown)}")
        cfg.update(config)
        self.config = _Config(cfg)
        boc = tuple(cfg["block_out_channels"])
        if len(boc) != 4 or any(c % 64 for c in boc) or any((c // 8) not in (40, 80, 160) for c in boc):
            raise ValueError(f"block_out_channels {boc}: the HIP kernels cover head dims 40/80/160 (SD-1.5 widths)")
        self.boc = boc
        self.heads = 8
        self.in_channels = cfg["in_channels"]
        self.out_channels = cfg["out_channels"]
        self._device = torch.device(device)
        self._dtype = dtype
        hip.dtype_code(dtype)
        self.training = True                 # from_config leaves the module in train() mode
        self.gradient_checkpointing = False
        # A/B switches of the host side live in the library's mmgt_tune table (hip.tune("fused_ff", 0) / MMGT_TUNE="fused_ff=0"), read
        # once here: one state describes a run
        self._fuse_ff = bool(hip.tune_get("fused_ff"))            # 0: the three-launch FeedForward
        self._twin = bool(hip.tune_get("twin_attention"))         # one attention pass for both rows of the first reader
        self._share_rows = bool(hip.tune_get("shared_rows"))      # conv_in + first resnet once for both CFG rows
        self._fuse_oz = bool(hip.tune_get("oz3"))                 # 0: the three masked audio out-projections as separate launches
        self._fuse_ln = bool(hip.tune_get("rowgemm"))             # 0: LayerNorm and q / k / v GEMMs as separate launches
        self._ffpo_cat = bool(hip.tune_get("ffpo_cat"))           # ff2 + residual + proj_out + residual of a block as one two-source GEMM (0: two GEMMs)
        self._ffpo = {}
        self._sc_cat = bool(hip.tune_get("sc_cat"))               # a resnet's conv_shortcut over [x | skip] as one two-source launch (0: two GEMMs)
        self._up2 = bool(hip.tune_get("up2"))                     # the up-sampling convs as four 2 x 2 convs on the stored image (0: the 3 x 3 conv on the upsampled view)
        self._fuse_tleg = bool(hip.tune_get("tleg"))              # 0: a level-0 temporal-attention leg as three launches
        self._tables = None                                       # (reader key, data_ptr, images, (scale, shift)): GroupNorm tables handed from a conv's epilogue to the norm's reader
        self._rconv_stats = bool(hip.tune_get("rconv_stats"))     # a fused leg's output statistics from its epilogue instead of a pass over the tensor
        self._rconv = hip.tune_get("rconv")                        # resnets whose GroupNorm -> SiLU -> conv3x3 legs run as one launch (csrc/rconv.hip): a mask of
                                                                   # 1 the 320-wide level, 2 the 640-wide level, 4 the 1280-wide level (16 x 16 pixel tiles)
        self.spec = unet3d_spec(boc, cfg["cross_attention_dim"], cfg["audio_attention_dim"], self.in_channels,
                                self.out_channels, cfg["layers_per_block"],
                                cfg["motion_module_kwargs"].get("temporal_position_encoding_max_len", 32))
        self.w: Dict[str, torch.Tensor] = {}
        self._banks: Dict[str, tuple] = {}
        self.bank_fp16_roundtrip = dtype == torch.float32   # reference stores banks as fp16 (mutual_self_attention.py:340)
        self._loaded = False
        self._ehs_cache = None
        self._zbias = {}

    # ------------------------------------------------------------------------------------------ reference-style API
    @classmethod
    def from_config(cls, config, **kwargs):
        config = dict(config)
        config.update(kwargs)
        dev = config.pop("device", "cuda")
        dt = config.pop("dtype", torch.bfloat16)
        return cls(device=dev, dtype=dt, **config)

    @classmethod
    def from_pretrained_2d(cls, pretrained_model_path, motion_module_path, subfolder=None, unet_additional_kwargs=None,
                           mm_zero_proj_out=False, device="cuda", dtype=torch.bfloat16):
        """unet_3d.py:627-718: SD-1.5 `unet/` weights merged with the motion-module checkpoint, strict=False."""
        import json
        from pathlib import Path
        path = Path(pretrained_model_path)
        if subfolder is not None:
            path = path / subfolder
        cfg_file = path / "config.json"
        if not cfg_file.is_file():
            raise RuntimeError(f"{cfg_file} does not exist or is not a file")
        cfg = json.load(open(cfg_file))
        keep = {k: v for k, v in cfg.items() if k in SD15_CONFIG}
        model = cls(device=device, dtype=dtype, **keep, **{k: v for k, v in (unet_additional_kwargs or {}).items()})
        if (path / "diffusion_pytorch_model.safetensors").exists():
            from safetensors.torch import load_file
            sd = load_file(str(path / "diffusion_pytorch_model.safetensors"), device="cpu")
        elif (path / "diffusion_pytorch_model.bin").exists():
            sd = torch.load(path / "diffusion_pytorch_model.bin", map_location="cpu", weights_only=True)
        else:
            raise FileNotFoundError(f"no weights file found in {path}")
        mpath = Path(motion_module_path)
        if mpath.exists() and mpath.is_file():
            if mpath.suffix.lower() in (".pth", ".pt", ".ckpt"):
                msd = torch.load(mpath, map_location="cpu", weights_only=True)
            elif mpath.suffix.lower() == ".safetensors":
                from safetensors.torch import load_file
                msd = load_file(str(mpath), device="cpu")
            else:
                raise RuntimeError(f"unknown file format for motion module weights: {mpath.suffix}")
            if mm_zero_proj_out:
                msd = {k: v for k, v in msd.items() if "proj_out" not in k}
            sd.update(msd)
        model.load_state_dict(sd, strict=False)
        return model

    @property
    def dtype(self):
        return self._dtype

    @property
    def device(self):
        return self._device

    def to(self, device=None, dtype=None):
        if (device is not None and torch.device(device) != self._device) or (dtype is not None and dtype != self._dtype):
            raise RuntimeError("construct UNet3DConditionModel with the target device/dtype: packed weights are not re-cast")
        return self

    def train(self, mode=True):
        self.training = mode
        return self

    def eval(self):
        return self.train(False)

    def enable_gradient_checkpointing(self):
        """Only observable effect at inference: with train() mode it selects the motion_scale-weighted MM-HAA sum
        (unet_3d_blocks.py:539-572 vs :591-600)."""
        self.gradient_checkpointing = True

    def disable_gradient_checkpointing(self):
        self.gradient_checkpointing = False

    def state_dict_spec(self):
        return dict(self.spec)

    # ------------------------------------------------------------------------------------------ weights
    def load_state_dict(self, sd, strict=True):
        missing = [k for k in self.spec if k not in sd]
        unexpected = [k for k in sd if k not in self.spec]
        if strict and (missing or unexpected):
            raise RuntimeError(f"load_state_dict: missing {missing[:5]}... ({len(missing)}), unexpected "
                               f"{unexpected[:5]}... ({len(unexpected)})")
        for k in self.spec:
            if k in sd and tuple(sd[k].shape) != tuple(self.spec[k]):
                raise RuntimeError(f"load_state_dict: shape mismatch for {k}: {tuple(sd[k].shape)} vs {self.spec[k]}")
        if missing and not self._loaded:
            # strict=False on a fresh model (from_pretrained_2d, unet_3d.py:710): the reference keeps the constructor's
            # initialisation for keys the checkpoints do not hold (audio modules before stage-2 training, ...)
            sd = dict(sd)
            sd.update(default_init({k: self.spec[k] for k in missing}))
        elif missing:
            raise RuntimeError("partial load_state_dict on an already packed model is not supported: pass every key")
        self._pack(sd)
        self._loaded = True
        self._ehs_cache = None
        self._zbias = {}
        self._ffpo = {}
        self._banks = {}
        return missing, unexpected

    def _t(self, x):      # model-dtype device copy
        return x.to(device=self._device, dtype=self._dtype).contiguous()

    def _f(self, x):      # fp32 device copy (bias / norm affine)
        return x.to(device=self._device, dtype=torch.float32).contiguous()

    def _pack(self, sd):
        w = self.w
        has = lambda k: k in sd

        def norm(p):
            if has(p + ".weight"):
                w[p + ".g"], w[p + ".b"] = self._f(sd[p + ".weight"]), self._f(sd[p + ".bias"])

        def lin(p, key=None):
            key = key or p
            if has(p + ".weight"):
                w[key + ".w"] = self._t(sd[p + ".weight"].reshape(sd[p + ".weight"].shape[0], -1))
            if has(p + ".bias"):
                w[key + ".bias"] = self._f(sd[p + ".bias"])

        def ff(p):
            if has(p + ".net.0.proj.weight"):
                wp, bp = pack_geglu(sd[p + ".net.0.proj.weight"], sd[p + ".net.0.proj.bias"])
                w[p + ".ff1.w"], w[p + ".ff1.bias"] = self._t(wp), self._f(bp)
            lin(p + ".net.2", p + ".ff2")
            if has(p + ".net.0.proj.weight") and has(p + ".net.2.weight"):
                c, inner = sd[p + ".net.2.weight"].shape
                if self._fuse_ff and hip.ff_fused_supported(c, inner, self._dtype):
                    # the 320-channel level: LayerNorm -> ff1 -> GEGLU -> ff2 -> + residual as one launch (csrc/ffn.hip)
                    w[p + ".ffimg"] = pack_ff_fused(sd[p + ".net.0.proj.weight"].to(self._device), self._f(sd[p + ".net.0.proj.bias"]),
                                                    sd[p + ".net.2.weight"].to(self._device))

        def rowimg(key, ws):
            """The 320-channel level: LayerNorm -> Linear(s) as one launch that keeps the rows in registers (csrc/rowgemm.hip)."""
            wcat = torch.cat(ws, 0)
            if self._fuse_ln and hip.rowgemm320_supported(self._dtype, wcat.shape[1], wcat.shape[0]):
                w[key] = pack_rowgemm(wcat.to(self._device))

        def proj_in_img(p):      # GroupNorm -> proj_in of a transformer block as one launch (GroupNorm applied while the rows are loaded)
            if has(p + ".proj_in.weight"):
                wt = sd[p + ".proj_in.weight"]
                rowimg(p + ".proj_in.img", [wt.reshape(wt.shape[0], -1)])
            if has(p + ".proj_out.weight"):     # ... and proj_out + residual on the end of the fused FeedForward launch (csrc/ffn.hip)
                wt = sd[p + ".proj_out.weight"]
                wt = wt.reshape(wt.shape[0], -1)
                if self._fuse_ff and tuple(wt.shape) == (320, 320) and hip.ff_fused_supported(320, 1280, self._dtype):
                    w[p + ".proj_out.img"] = pack_ff_proj_out(wt.to(self._device))

        def self_attn(p):
            if has(p + ".to_q.weight"):
                w[p + ".qk.w"] = self._t(torch.cat([sd[p + ".to_q.weight"], sd[p + ".to_k.weight"]], 0))
                w[p + ".k.w"] = self._t(sd[p + ".to_k.weight"])
                w[p + ".v.w"] = self._t(sd[p + ".to_v.weight"])
                rowimg(p + ".qkv_img", [sd[p + ".to_q.weight"], sd[p + ".to_k.weight"], sd[p + ".to_v.weight"]])
            lin(p + ".to_out.0", p + ".o")

        def conv(p, cin_pad=None, cout_pad=None):
            if has(p + ".weight"):
                w[p + ".w"] = self._t(pack_conv3x3(sd[p + ".weight"], cin_pad, cout_pad))
            if has(p + ".bias"):
                w[p + ".bias"] = self._f(pad_rows(sd[p + ".bias"], cout_pad or sd[p + ".bias"].shape[0]))

        conv("conv_in", cin_pad=64)
        conv("conv_out", cout_pad=64)
        if self._dtype == torch.bfloat16 and hip.tune_get("conv_out_taps") and has("conv_out.weight") and tuple(sd["conv_out.weight"].shape) == (4, 320, 3, 3) \
                and hip.rowgemm320_supported(self._dtype, 320, 64):
            # conv_norm_out + SiLU + conv_out as ONE 36-column GEMM over the pixels (all nine taps) + a gather of the neighbours' products
            w["conv_out.timg"] = pack_conv_taps(sd["conv_out.weight"].to(self._device))
        norm("conv_norm_out")
        lin("time_embedding.linear_1")
        lin("time_embedding.linear_2")

        self._resnets = [k[: -len(".conv1.weight")] for k in self.spec if k.endswith(".conv1.weight")]
        temb_w, temb_b, off = [], [], 0
        self._temb_slices = {}
        for p in self._resnets:
            norm(p + ".norm1")
            norm(p + ".norm2")
            conv(p + ".conv1")
            conv(p + ".conv2")
            cout = self.spec[p + ".conv1.weight"][0]
            if self._dtype == torch.bfloat16 and (self._rconv & {320: 1, 640: 2, 1280: 4}.get(cout, 0)) and \
                    has(p + ".conv1.weight") and has(p + ".conv2.weight"):
                for cv in (".conv1", ".conv2"):                        # fragment-major images of the fused GroupNorm + SiLU + conv launch
                    wt = sd[p + cv + ".weight"]
                    if hip.gn_silu_conv3x3_unet_supported(self._dtype, wt.shape[1], 0, wt.shape[0], 16, 16):
                        w[p + cv + ".rimg"] = pack_rconv(wt.to(self._device))
            self._temb_slices[p] = (off, cout)
            off += cout
            if has(p + ".time_emb_proj.weight"):
                temb_w.append(sd[p + ".time_emb_proj.weight"])
                temb_b.append(sd[p + ".time_emb_proj.bias"])
            if (p + ".conv_shortcut.weight") in self.spec and has(p + ".conv_shortcut.weight"):
                ws = sd[p + ".conv_shortcut.weight"].reshape(cout, -1)
                w[p + ".sc.w"] = self._t(ws)
                w[p + ".sc.bias"] = self._f(sd[p + ".conv_shortcut.bias"])
        if temb_w:
            assert len(temb_w) == len(self._resnets), "time_emb_proj weights must be loaded together"
            w["temb_all.w"] = self._t(torch.cat(temb_w, 0))
            w["temb_all.bias"] = self._f(torch.cat(temb_b, 0))

        for k in self.spec:
            if k.endswith("samplers.0.conv.weight"):
                conv(k[: -len(".weight")])
                if ".upsamplers." in k and self._up2 and self._dtype == torch.bfloat16 and has(k) and (sd[k].shape[0] % 256 == 0 or sd[k].shape[0] % 320 == 0):
                    # the conv behind the nearest 2x upsampling (resnet.py:31-77, Upsample3D: F.interpolate(nearest, x2) -> conv) as four 2 x 2 convs on the stored image: 16 / 36 of the work
                    w[k[: -len(".weight")] + ".w2"] = self._t(pack_conv3x3_up2(sd[k]))

        self._spatial = [k[: -len(".transformer_blocks.0.attn2.to_q.weight")] for k in self.spec
                         if k.endswith(".transformer_blocks.0.attn2.to_q.weight")]
        for p in self._spatial:
            t = p + ".transformer_blocks.0"
            norm(p + ".norm")
            lin(p + ".proj_in")
            proj_in_img(p)
            lin(p + ".proj_out")
            for n in ("norm1", "norm2", "norm3"):
                norm(f"{t}.{n}")
            self_attn(t + ".attn1")
            lin(t + ".attn2.to_q", t + ".attn2.q")
            if has(t + ".attn2.to_k.weight"):
                w[t + ".attn2.kv.w"] = self._t(torch.cat([sd[t + ".attn2.to_k.weight"], sd[t + ".attn2.to_v.weight"]], 0))
                w[t + ".attn2.v.w"] = self._t(sd[t + ".attn2.to_v.weight"])
            lin(t + ".attn2.to_out.0", t + ".attn2.o")
            ff(t + ".ff")

        self._audio = [k[: -len(".transformer_blocks.0.attn2_0.to_q.weight")] for k in self.spec
                       if k.endswith(".transformer_blocks.0.attn2_0.to_q.weight")]
        for p in self._audio:
            t = p + ".transformer_blocks.0"
            norm(p + ".norm")
            lin(p + ".proj_in")
            proj_in_img(p)
            lin(p + ".proj_out")
            for n in ("norm1", "norm2", "norm3"):
                norm(f"{t}.{n}")
            self_attn(t + ".attn1")
            if has(f"{t}.attn2_0.to_q.weight"):
                w[t + ".q3.w"] = self._t(torch.cat([sd[f"{t}.attn2_{i}.to_q.weight"] for i in range(3)], 0))
                rowimg(t + ".q3_img", [sd[f"{t}.attn2_{i}.to_q.weight"] for i in range(3)])
                w[t + ".kv3.w"] = self._t(torch.cat([sd[f"{t}.attn2_{i}.to_k.weight"] for i in range(3)] +
                                                    [sd[f"{t}.attn2_{i}.to_v.weight"] for i in range(3)], 0))
            for i, z in enumerate(("zero_conv_full", "zero_conv_face", "zero_conv_lip")):
                lin(f"{t}.attn2_{i}.to_out.0", f"{t}.o{i}")
                lin(f"{t}.{z}", f"{t}.z{i}")
                if has(f"{t}.attn2_{i}.to_out.0.weight") and has(f"{t}.{z}.weight"):
                    # zero_conv_i(mask_i * to_out_i(a)) = mask_i * (a (Wz Wo)^T + Wz b_o) + b_z  (attention.py:730-760): the
                    # per-token mask is a row scalar, so the 1x1 conv folds into the Linear -- one GEMM per branch
                    # (Wz Wo and Wz b_o: fp32 GEMMs of the library itself, once per load_state_dict)
                    wz = self._f(sd[f"{t}.{z}.weight"].reshape(sd[f"{t}.{z}.weight"].shape[0], -1))
                    wo_t = self._f(sd[f"{t}.attn2_{i}.to_out.0.weight"].t())
                    bo = self._f(sd[f"{t}.attn2_{i}.to_out.0.bias"])[None, :].contiguous()
                    w[f"{t}.oz{i}.w"] = self._t(hip.gemm(wz, wo_t))
                    w[f"{t}.oz{i}.bias"] = hip.gemm(wz, bo).reshape(-1).contiguous()
            if all((f"{t}.oz{i}.w") in w for i in range(3)):
                # the three branches as ONE GEMM over the concatenated reduction (see _audio_transformer): [Wzo_0 | Wzo_1 | Wzo_2 | Wz_i b_o,i | 0]
                # The bias term mask_i s_i (Wz_i b_o,i) keeps fp32 accuracy on the bf16 path: both factors are split into a storage-
                # dtype head and tail, three operand columns [m_hi | m_hi | m_lo] per branch against [b_hi | b_lo | b_hi] (the lo x lo
                # term, 2^-16 relative, is dropped; in fp32 storage the tails are zero) -- the unfused path multiplies the fp32 mask
                # with the fp32 bias, and blurred masks are not 8-bit values
                inner3 = sum(w[f"{t}.oz{i}.w"].shape[1] for i in range(3))
                pad = torch.zeros((w[f"{t}.oz0.w"].shape[0], round_up(inner3 + 9, 64) - inner3), device=self._device, dtype=torch.float32)
                for i in range(3):
                    b = w[f"{t}.oz{i}.bias"]
                    b_hi = b.to(self._dtype).float()
                    pad[:, 3 * i], pad[:, 3 * i + 1], pad[:, 3 * i + 2] = b_hi, b - b_hi, b_hi
                w[t + ".oz3.w"] = torch.cat([w[f"{t}.oz{i}.w"] for i in range(3)] + [self._t(pad)], 1).contiguous()
                w[t + ".oz3.wb"] = self._t(pad).contiguous()          # the bias columns alone: rows whose attention output is zero
            ff(t + ".ff")

        self._motion = [k[: -len(".temporal_transformer.norm.weight")] for k in self.spec
                        if k.endswith(".temporal_transformer.norm.weight")]
        for p in self._motion:
            q = p + ".temporal_transformer"
            t = q + ".transformer_blocks.0"
            norm(q + ".norm")
            lin(q + ".proj_in")
            proj_in_img(q)
            lin(q + ".proj_out")
            for i in range(2):
                a = f"{t}.attention_blocks.{i}"
                if has(a + ".to_q.weight"):
                    w[a + ".qkv.w"] = self._t(torch.cat([sd[a + ".to_q.weight"], sd[a + ".to_k.weight"],
                                                         sd[a + ".to_v.weight"]], 0))
                    rowimg(a + ".qkv_img", [sd[a + ".to_q.weight"], sd[a + ".to_k.weight"], sd[a + ".to_v.weight"]])
                    if self._fuse_tleg and has(a + ".to_out.0.weight") and has(a + ".to_out.0.bias") and \
                            hip.temporal_leg320_supported(self._dtype, sd[a + ".to_q.weight"].shape[1], self.heads, 24, 4096):
                        # the whole leg (LayerNorm + pe -> q | k | v -> attention over the frames -> to_out + residual) as one launch (csrc/tleg.hip)
                        w[a + ".tleg_img"] = pack_tleg(*(sd[a + k].to(self._device) for k in (".to_q.weight", ".to_k.weight", ".to_v.weight",
                                                                                               ".to_out.0.weight")))
                lin(a + ".to_out.0", a + ".o")
                if has(a + ".pos_encoder.pe"):
                    w[a + ".pe"] = self._f(sd[a + ".pos_encoder.pe"][0])
                norm(f"{t}.norms.{i}")
                if (a + ".pe") in w and (f"{t}.norms.{i}.b") in w:
                    # LayerNorm bias + positional encoding as one (max_len, C) table (motion_module.py:359-366 adds pe
                    # right after the norm): the kernel then reads gamma and one table row instead of gamma, beta and pe
                    w[f"{t}.norms.{i}.bpe"] = (w[f"{t}.norms.{i}.b"][None, :] + w[a + ".pe"]).contiguous()
            norm(t + ".ff_norm")
            ff(t + ".ff")

    # ------------------------------------------------------------------------------------------ reference banks
    def bank_keys(self) -> List[str]:
        """The 16 reference-attention readers in the reference's module order down -> up -> mid."""
        down = [p for p in self._spatial if p.startswith("down_blocks")]
        up = [p for p in self._spatial if p.startswith("up_blocks")]
        mid = [p for p in self._spatial if p.startswith("mid_block")]
        return down + up + mid

    def set_banks(self, banks: Optional[Dict[str, torch.Tensor]]):
        """banks: {prefix: (2, N, C)} LayerNorm'd ReferenceNet features (what `ReferenceAttentionControl.update`
        hands over).  Projects K and V^T of every bank ONCE (they are step- and frame-invariant)."""
        self._banks = {}
        if not banks:
            return
        for p, bank in banks.items():
            if p not in self._spatial:
                raise KeyError(f"set_banks: {p} is not a reference-attention reader")
            t = p + ".transformer_blocks.0.attn1"
            bk = bank.to(self._device)
            if self.bank_fp16_roundtrip:
                bk = bk.to(torch.float16)
            bk = bk.to(self._dtype).contiguous()
            two, n, c = bk.shape
            kb = hip.gemm(bk.view(two * n, c), self.w[t + ".k.w"]).view(two, n, -1)
            vbt = torch.empty((two, kb.shape[2], round_up(n, 8)), device=self._device, dtype=self._dtype)
            hip.gemm_batched_wx(self.w[t + ".v.w"], bk, out=vbt)
            self._banks[p] = (kb, vbt, n)

    def clear_banks(self):
        self._banks = {}

    # ------------------------------------------------------------------------------------------ building blocks
    def _gn(self, p, x, eps, silu=False, x1=None):
        nb, h, ww, c = x.shape
        y = hip.groupnorm(x.view(nb, h * ww, c), self.w[p + ".g"], self.w[p + ".b"], 32, eps, silu=silu,
                          x1=None if x1 is None else x1.view(nb, h * ww, -1))
        return y.view(nb, h, ww, -1)

    def _ln(self, p, x, **kw):
        return hip.layernorm(x, self.w[p + ".g"], self.w[p + ".b"], 1e-5, **kw)

    def _lin(self, p, x, **kw):
        return hip.gemm(x, self.w[p + ".w"], self.w.get(p + ".bias"), **kw)

    def _ff(self, p, x, residual):
        g = hip.gemm(x, self.w[p + ".ff1.w"], self.w[p + ".ff1.bias"], act=hip.ACT_GEGLU)
        return hip.gemm(g, self.w[p + ".ff2.w"], self.w[p + ".ff2.bias"], residual=residual)

    def _norm_ff(self, p, norm, hid):
        """hid + FeedForward(LayerNorm(hid)) (attention.py:361,465,642,769; motion_module.py:253-254): one fused launch where the
        weight image exists (bf16, 320 channels), LayerNorm -> GEMM(GEGLU) -> GEMM(+ residual) otherwise."""
        img = self.w.get(p + ".ffimg")
        if img is not None:
            return hip.ff_fused(hid, self.w[norm + ".g"], self.w[norm + ".b"], img, self.w[p + ".ff2.bias"], hid,
                                self.w[p + ".ff2.w"].shape[1])
        return self._ff(p, self._ln(norm, hid), hid)

    def _norm_proj_in(self, p, x):
        """proj_in(GroupNorm(x)) of a transformer block (transformer_3d.py:174-188, motion_module.py:156-170) -> (nb*h*w, inner): at the
        320-channel level the statistics pass alone, then one launch that normalises the rows while it loads them (csrc/rowgemm.hip)."""
        nb, h, ww, c = x.shape
        n = h * ww
        img = self.w.get(p + ".proj_in.img")
        if img is not None and n > 256 and n % 128 == 0:
            kept, self._tables = self._tables, None
            if kept is not None and kept[0] == p + ".norm" and kept[1] == x.data_ptr() and kept[2] == nb:
                sc, sh = kept[3]          # from the epilogue of the conv that wrote x (`_resnet`, reader=)
            else:
                sc, sh = hip.groupnorm_affine(x.view(nb, n, c), self.w[p + ".norm.g"], self.w[p + ".norm.b"], 32, 1e-6)
            return hip.rowgemm320(x.view(nb * n, c), img, self.w[p + ".proj_in.w"].shape[0], self.w.get(p + ".proj_in.bias"),
                                  pre_scale=sc, pre_shift=sh, pre_rows=n)[0]
        xn = self._gn(p + ".norm", x, 1e-6)
        return self._lin(p + ".proj_in", xn.view(nb * n, c))

    def _norm_ff_proj_out(self, p, norm, hid, q, x_res):
        """proj_out(hid + FeedForward(LayerNorm(hid))) + x_res, the tail of a transformer block (transformer_3d.py:262-268,
        motion_module.py:178-182): one launch where both weight images exist (bf16, 320 channels), else FeedForward then GEMM."""
        img, po = self.w.get(p + ".ffimg"), self.w.get(q + ".proj_out.img")
        if img is not None and po is not None and (q + ".proj_out.bias") in self.w:
            return hip.ff_fused_po(hid, self.w[norm + ".g"], self.w[norm + ".b"], img, self.w[p + ".ff2.bias"], hid,
                                   self.w[p + ".ff2.w"].shape[1], po, self.w[q + ".proj_out.bias"], x_res)
        wcat = self._ffpo_weights(p, q) if img is None else None
        if wcat is not None and hid.shape[0] * max(wcat[0].shape[1] - hid.shape[1], hid.shape[1]) * 2 < hip.DMA_LIMIT:
            # proj_out(hid + b2 + W2 g) + x_res = [W_po W2 | W_po] . [g | hid] + (b_po + W_po b2) + x_res: the block's last two GEMMs as ONE over the
            # two sources g (GEGLU output) and hid -- the FeedForward's result is read by nothing else (transformer_3d.py:262-268, motion_module.py:178-182)
            g = hip.gemm(self._ln(norm, hid), self.w[p + ".ff1.w"], self.w[p + ".ff1.bias"], act=hip.ACT_GEGLU)
            return hip.conv1x1_cat(g, hid, wcat[0], wcat[1], residual=x_res)
        hid = self._norm_ff(p, norm, hid)
        return hip.gemm(hid, self.w[q + ".proj_out.w"], self.w.get(q + ".proj_out.bias"), residual=x_res)

    def _ffpo_weights(self, p, q):
        """([W_po W2 | W_po] (C, inner + C) in the model dtype, b_po + W_po b2 (fp32)) of a transformer block whose FeedForward output feeds proj_out
        alone; products in fp32 from the stored weights, rounded once.  None where the one-launch form does not apply."""
        key = p + ".ffpo"
        if key not in self._ffpo:
            w2, wpo = self.w.get(p + ".ff2.w"), self.w.get(q + ".proj_out.w")
            ok = self._ffpo_cat and w2 is not None and wpo is not None and (p + ".ff2.bias") in self.w and (q + ".proj_out.bias") in self.w and \
                wpo.shape[0] == wpo.shape[1] == w2.shape[0] and hip.conv1x1_cat_supported(self._dtype, w2.shape[1], w2.shape[0], wpo.shape[0])
            if ok:
                wpf = wpo.float()
                wc = torch.cat([wpf @ w2.float(), wpf], 1).to(self._dtype).contiguous()
                bc = (self.w[q + ".proj_out.bias"].float() + wpf @ self.w[p + ".ff2.bias"].float()).contiguous()
                self._ffpo[key] = (wc, bc)
            else:
                self._ffpo[key] = None
        return self._ffpo[key]

    def _resnet(self, p, x, temb, skip=None, out=None, reader=None):
        """ResnetBlock3D (resnet.py:217-247); `skip` = the UNet skip tensor that the reference concatenates first.  reader: key of the
        GroupNorm that reads the result next (a transformer block's `norm`): on the fused path its tables come out of conv2's epilogue
        (`_norm_proj_in` picks them up from self._tables)."""
        nb, h, ww, c0 = x.shape
        hw = h * ww
        cout = self.spec[p + ".conv1.weight"][0]
        c1 = 0 if skip is None else skip.shape[3]
        # GroupNorm-apply + SiLU + conv3x3 as ONE launch behind the statistics pass (csrc/rconv.hip): the normalised tensor is never written
        fused = (p + ".conv1.rimg") in self.w and (p + ".conv2.rimg") in self.w and \
            hip.gn_silu_conv3x3_unet_supported(self._dtype, c0, c1, cout, h, ww) and hip.gn_silu_conv3x3_unet_supported(self._dtype, cout, 0, cout, h, ww)
        if fused:
            eps = self.config.norm_eps
            sc, sh = hip.groupnorm_affine(x.view(nb, hw, c0), self.w[p + ".norm1.g"], self.w[p + ".norm1.b"], 32, eps,
                                          x1=None if skip is None else skip.view(nb, hw, c1))
            if self._rconv_stats:       # norm2's statistics from conv1's epilogue (resnet.py:231): no pass over the tensor
                hdn, (sc, sh) = hip.gn_silu_conv3x3_unet(x, sc, sh, self.w[p + ".conv1.rimg"], cout, self.w[p + ".conv1.bias"], temb[p], nb // temb[p].shape[0],
                                                         x1=skip, next_norm=(self.w[p + ".norm2.g"], self.w[p + ".norm2.b"], 32, eps))
            else:
                hdn = hip.gn_silu_conv3x3_unet(x, sc, sh, self.w[p + ".conv1.rimg"], cout, self.w[p + ".conv1.bias"], temb[p], nb // temb[p].shape[0], x1=skip)
                sc, sh = hip.groupnorm_affine(hdn.view(nb, hw, cout), self.w[p + ".norm2.g"], self.w[p + ".norm2.b"], 32, eps)
        else:
            hdn = self._gn(p + ".norm1", x, self.config.norm_eps, silu=True, x1=skip)
            b2rows = (nb // temb[p].shape[0]) * hw
            if temb[p].shape[0] == 1:
                b2rows = max(b2rows, 256)       # one row for every output row: any count >= M is the same sum, and >= 256 keeps the vectorised epilogue
            hdn = hip.conv3x3(hdn, self.w[p + ".conv1.w"], self.w[p + ".conv1.bias"], bias2=temb[p], bias2_rows=b2rows)
            hdn = self._gn(p + ".norm2", hdn, self.config.norm_eps, silu=True)
        if (p + ".sc.w") in self.w:
            wsc = self.w[p + ".sc.w"]
            if skip is None:
                res = hip.gemm(x.view(nb * hw, c0), wsc, self.w[p + ".sc.bias"])
            else:
                c1 = skip.shape[3]
                if self._sc_cat and nb * hw >= 6144 and hip.conv1x1_cat_supported(self._dtype, c0, c1, cout) and nb * hw * max(c0, c1) * 2 < hip.DMA_LIMIT:   # (3072 rows: 54 us against two GEMMs of 23)
                    # conv_shortcut over [x | skip] as one launch (two-source gather, one tap) instead of two GEMMs chained through a residual
                    res = hip.conv1x1_cat(x.view(nb * hw, c0), skip.view(nb * hw, c1), wsc, self.w[p + ".sc.bias"])
                else:
                    res = hip.gemm(x.view(nb * hw, c0), self._sc_split(p, c0, 0), self.w[p + ".sc.bias"])
                    res = hip.gemm(skip.view(nb * hw, c1), self._sc_split(p, c0, 1), None, residual=res)
            res = res.view(nb, h, ww, cout)
        else:
            assert skip is None
            res = x
        if fused:
            if reader is not None and self._rconv_stats and hw > 256 and (reader.rsplit(".", 1)[0] + ".proj_in.img") in self.w:
                y, tab = hip.gn_silu_conv3x3_unet(hdn, sc, sh, self.w[p + ".conv2.rimg"], cout, self.w[p + ".conv2.bias"], residual=res, out=out,
                                                  next_norm=(self.w[reader + ".g"], self.w[reader + ".b"], 32, 1e-6))
                self._tables = (reader, y.data_ptr(), nb, tab)
                return y
            return hip.gn_silu_conv3x3_unet(hdn, sc, sh, self.w[p + ".conv2.rimg"], cout, self.w[p + ".conv2.bias"], residual=res, out=out)
        return hip.conv3x3(hdn, self.w[p + ".conv2.w"], self.w[p + ".conv2.bias"], residual=res, out=out)

    def _sc_split(self, p, c0, which):
        key = f"{p}.sc.w.{which}"
        if key not in self.w:
            wsc = self.w[p + ".sc.w"]
            self.w[f"{p}.sc.w.0"] = wsc[:, :c0].contiguous()
            self.w[f"{p}.sc.w.1"] = wsc[:, c0:].contiguous()
        return self.w[key]

    def _self_attention(self, t, n1, nb, n, inner, bank=None, frames=1, cfg_row=None, norm=None):
        """attn1: q,k from one GEMM, V^T from a batched W.X^T GEMM, flash attention, returns (nb*n, inner).
        cfg_row: None = both CFG rows batched (the bank is read by the second half of the batch only); 0 / 1 = the batch
        holds the unconditional / conditional row alone (window-parallel sampling splits them over ranks).
        norm: `n1` is the UN-normalised hidden state and `norm` the key of the LayerNorm in front of the attention: LayerNorm, q | k
        and V^T come out of one launch (the 320-channel level, csrc/rowgemm.hip)."""
        hd = inner // self.heads
        npad = round_up(n, 8)
        vt = torch.empty((nb, inner, npad), device=self._device, dtype=self._dtype)
        if norm is not None:
            qk, _ = hip.rowgemm320(n1, self.w[t + ".qkv_img"], 3 * inner, ln_gamma=self.w[norm + ".g"], ln_beta=self.w[norm + ".b"],
                                   n1=2 * inner, n_tok=n, out_t=vt)
        else:
            qk = hip.gemm(n1, self.w[t + ".qk.w"])
            hip.gemm_batched_wx(self.w[t + ".v.w"], n1.view(nb, n, inner), out=vt)
        o = torch.empty((nb * n, inner), device=self._device, dtype=self._dtype)
        kw = {}
        if bank is not None and cfg_row != 0:
            kb, vbt, nkb = bank
            if cfg_row == 1:                 # only bank row 1 (the conditional row's features) is ever read: SURVEY App. C-6
                kb, vbt = kb[1:], vbt[1:]
            kw = dict(k2=kb, v2=vbt, k2_str=(kb.stride(0), kb.stride(1)), v2_str=(vbt.stride(0), vbt.stride(1)),
                      k2_bdiv=frames, nk2=nkb, seg2_first_batch=0 if cfg_row == 1 else nb // 2)
        hip.attention(qk, qk[:, inner:], vt, o, batch=nb, heads=self.heads, hd=hd, nq=n, nk=n, scale=hd ** -0.5,
                      q_str=(n * 2 * inner, 0, 2 * inner), k_str=(n * 2 * inner, 0, 2 * inner),
                      v_str=(inner * npad, 0, npad), o_str=(n * inner, 0, inner), v_transposed=True, **kw)
        return o

    def _spatial_transformer(self, p, x, ehs, frames, write=None, cfg_row=None):
        """Transformer3DModel + TemporalBasicTransformerBlock in bank-read mode (transformer_3d.py:139-268,
        mutual_self_attention.py:149-230)."""
        nb, h, ww, c = x.shape
        n = h * ww
        m = nb * n
        t = p + ".transformer_blocks.0"
        hid = self._norm_proj_in(p, x)
        inner = hid.shape[1]
        fuse = write is None and (t + ".attn1.qkv_img") in self.w and n % 128 == 0
        n1 = hid if fuse else self._ln(t + ".norm1", hid)
        if write is not None:            # ReferenceNet "write" mode: bank.append(norm_hidden_states) (mutual_self_attention.py:139-148)
            write[p] = n1.view(nb, n, inner).float()
        o = self._self_attention(t + ".attn1", n1, nb, n, inner, bank=None if write is not None else self._banks.get(p),
                                 frames=frames, cfg_row=cfg_row, norm=t + ".norm1" if fuse else None)
        if ehs.shape[1] == 1:
            # one key: softmax == 1, attn2 output is the per-CFG-row constant to_out(to_v(e))
            cvec = self._clip_vector(t, ehs)
            if cfg_row is not None:
                cvec = cvec[cfg_row:cfg_row + 1]
            rows = nb // cvec.shape[0] * n if cvec.shape[0] != nb else n
            hid = hip.gemm(o, self.w[t + ".attn1.o.w"], self.w[t + ".attn1.o.bias"], residual=hid,
                           bias2=cvec, bias2_rows=rows)
        else:
            hid = hip.gemm(o, self.w[t + ".attn1.o.w"], self.w[t + ".attn1.o.bias"], residual=hid)
            hid = self._cross_attention(t, hid, ehs if cfg_row is None else ehs[cfg_row:cfg_row + 1], nb, n, inner)
        out = self._norm_ff_proj_out(t + ".ff", t + ".norm3", hid, p, x.view(m, c))
        return out.view(nb, h, ww, c)

    def _spatial_transformer_twin(self, p, x2, ehs, frames):
        """The first reference-attention reader of a CFG pair whose rows entered with the same input (`cfg_rows_share_input`): x2 =
        ((2 f), h, w, c) with identical halves.  GroupNorm, proj_in, LayerNorm and q | k | V^T are computed for the f frames once; ONE
        attention pass over the frames' own keys serves both rows -- the unconditional row's output [x] is the state of the conditional
        row's [x | bank] pass after its last own-key tile (mmgt_attention_twin) -- and the rows part at the out-projection, which adds the
        per-row CLIP constant.  Returns None when the shape has no twin kernel (the caller then runs the batched block)."""
        nb, h, ww, c = x2.shape
        f, n = nb // 2, h * ww
        t = p + ".transformer_blocks.0"
        bank = self._banks.get(p)
        inner = self.w[t + ".attn1.o.w"].shape[1]
        hd = inner // self.heads
        if bank is None or (t + ".attn1.qkv_img") not in self.w or ehs.shape[1] != 1 or self._dtype != torch.bfloat16 or hd != 40 \
                or n % 256 or bank[2] % 64 or nb % 2:
            return None
        hid = self._norm_proj_in(p, x2[:f])                                              # (f n, inner)
        npad = round_up(n, 8)
        vt = torch.empty((f, inner, npad), device=self._device, dtype=self._dtype)
        qk, _ = hip.rowgemm320(hid, self.w[t + ".attn1.qkv_img"], 3 * inner, ln_gamma=self.w[t + ".norm1.g"], ln_beta=self.w[t + ".norm1.b"],
                               n1=2 * inner, n_tok=n, out_t=vt)
        kb, vbt, nkb = bank
        kb, vbt = kb[1:], vbt[1:]                       # the conditional row's reference features
        o = torch.empty((nb * n, inner), device=self._device, dtype=self._dtype)
        hip.attention(qk, qk[:, inner:], vt, o[f * n:], batch=f, heads=self.heads, hd=hd, nq=n, nk=n, scale=hd ** -0.5,
                      q_str=(n * 2 * inner, 0, 2 * inner), k_str=(n * 2 * inner, 0, 2 * inner), v_str=(inner * npad, 0, npad),
                      o_str=(n * inner, 0, inner), v_transposed=True, k2=kb, v2=vbt, k2_str=(kb.stride(0), kb.stride(1)),
                      v2_str=(vbt.stride(0), vbt.stride(1)), k2_bdiv=frames, nk2=nkb, seg2_first_batch=0, twin_out=o[:f * n])
        cvec = self._clip_vector(t, ehs)                                                 # (2, inner) fp32: row 0 unconditional
        hid2 = torch.empty((nb * n, inner), device=self._device, dtype=self._dtype)
        for row in (0, 1):
            sl = slice(row * f * n, (row + 1) * f * n)
            hip.gemm(o[sl], self.w[t + ".attn1.o.w"], self.w[t + ".attn1.o.bias"], residual=hid, bias2=cvec[row:row + 1],
                     bias2_rows=max(f * n, 256), out=hid2[sl])
        out = self._norm_ff_proj_out(t + ".ff", t + ".norm3", hid2, p, x2.view(nb * n, c))
        return out.view(nb, h, ww, c)

    def _clip_vector(self, t, ehs):
        """to_out(to_v(e)) of the one-token CLIP cross-attention, per CFG row, fp32.  It depends on the weights and on
        `encoder_hidden_states` only, which the sampler passes unchanged at every step: cached while the caller keeps
        handing in the same (unmodified) tensor object -- the reference held here keeps its storage from being reused."""
        c = self._ehs_cache
        if c is None or c[0] is not ehs or c[1] != ehs._version:
            c = self._ehs_cache = (ehs, ehs._version, {})
        if t not in c[2]:
            e = ehs.reshape(ehs.shape[0], -1).to(self._dtype).contiguous()
            c[2][t] = hip.gemm(hip.gemm(e, self.w[t + ".attn2.v.w"]), self.w[t + ".attn2.o.w"],
                               self.w[t + ".attn2.o.bias"]).float()
        return c[2][t]

    def _cross_attention(self, t, hid, ehs, nb, n, inner):
        """General attn2 (more than one context token): attention.py:448-462."""
        hd = inner // self.heads
        n2 = self._ln(t + ".norm2", hid)
        q = hip.gemm(n2, self.w[t + ".attn2.q.w"])
        e = ehs.to(self._dtype)
        if e.shape[0] != nb:
            e = e.repeat_interleave(nb // e.shape[0], dim=0)
        e = e.contiguous()
        l = e.shape[1]
        kv = hip.gemm(e.view(nb * l, -1), self.w[t + ".attn2.kv.w"])
        o = torch.empty_like(q)
        hip.attention(q, kv, kv[:, inner:], o, batch=nb, heads=self.heads, hd=hd, nq=n, nk=l, scale=hd ** -0.5,
                      q_str=(n * inner, 0, inner), k_str=(l * 2 * inner, 0, 2 * inner), v_str=(l * 2 * inner, 0, 2 * inner),
                      o_str=(n * inner, 0, inner))
        return hip.gemm(o, self.w[t + ".attn2.o.w"], self.w[t + ".attn2.o.bias"], residual=hid)

    def _audio_transformer(self, p, x, audio, masks, depth, motion_scale, ms_cache=None):
        """MM-HAA (attention.py:649-771): self-attention, three masked audio cross-attentions through zero-convs."""
        nb, h, ww, c = x.shape
        n = h * ww
        m = nb * n
        t = p + ".transformer_blocks.0"
        hid = self._norm_proj_in(p, x)
        inner = hid.shape[1]
        hd = inner // self.heads
        if (t + ".attn1.qkv_img") in self.w and n % 128 == 0:
            o = self._self_attention(t + ".attn1", hid, nb, n, inner, norm=t + ".norm1")
        else:
            o = self._self_attention(t + ".attn1", self._ln(t + ".norm1", hid), nb, n, inner)
        hid = hip.gemm(o, self.w[t + ".attn1.o.w"], self.w[t + ".attn1.o.bias"], residual=hid)
        la = audio.shape[1]
        state = ms_cache.get("state") if ms_cache is not None else None
        fused = self._fuse_oz and (t + ".oz3.w") in self.w and ms_cache is not None
        # images whose audio embedding is ALL ZERO (the unconditional CFG row: pipeline_pose2vid_long.py:484-485): to_k / to_v carry no bias,
        # so their keys and values are zero, every score is 0, the softmax is uniform and the attention output is exactly 0 -- their q
        # projection and attention are not computed, their rows of the merged operand stay zero (only the bias / mask terms remain)
        nb0 = min(nb, ms_cache.get("zero_images", 0)) if fused else 0
        m0 = nb0 * n
        q3 = kv3 = None
        if nb0 < nb:
            hid_c = hid[m0:]
            if (t + ".q3_img") in self.w:
                q3, _ = hip.rowgemm320(hid_c, self.w[t + ".q3_img"], 3 * inner, ln_gamma=self.w[t + ".norm2.g"], ln_beta=self.w[t + ".norm2.b"])
            else:
                q3 = hip.gemm(self._ln(t + ".norm2", hid_c), self.w[t + ".q3.w"])
            kv3 = state.get(("kv3", t, nb0)) if state is not None else None
            if kv3 is None:
                kv3 = hip.gemm(audio[nb0:].reshape((nb - nb0) * la, -1), self.w[t + ".kv3.w"])
                if state is not None:
                    state[("kv3", t, nb0)] = kv3
        scales = tuple(1.0 if motion_scale is None else float(motion_scale[i]) for i in range(3))
        if fused:
            # sum_i zero_conv_i(mask_i * to_out_i(a_i)) as ONE GEMM: the attention writes mask_i s_i a_i (fp32 multiplier in its softmax
            # normalisation), three extra columns of the operand hold mask_i s_i against the merged biases Wz_i b_o,i in the weight
            # image, and the residual stream is read and written once instead of three times (attention.py:730-760)
            k3 = 3 * inner
            kp = self.w[t + ".oz3.w"].shape[1]
            ck = (depth, k3)
            if ck not in ms_cache:                     # once per forward and level: the masks do not change between its modules
                kept = state.get(("mask_rows", depth, kp - k3, scales)) if state is not None else None
                if kept is None:
                    rows = []
                    for i in range(3):
                        mask = masks[i][depth].reshape(-1).to(device=self._device, dtype=torch.float32)
                        if mask.numel() != m:
                            raise RuntimeError(f"mask level {depth} has {mask.numel()} entries, block has {m} tokens")
                        rows.append(mask if scales[i] == 1.0 else mask * scales[i])
                    rs = torch.stack(rows).contiguous()                               # (3, m) fp32
                    cols = mask_bias_columns(rs, kp - k3, self._dtype)
                    kept = (rs, cols)
                    if state is not None:
                        state[("mask_rows", depth, kp - k3, scales)] = kept
                rs, cols = kept
                buf = torch.empty((m, kp), device=self._device, dtype=self._dtype)   # scratch operand: not kept between steps
                buf[:, k3:] = cols
                if m0:
                    buf[:m0, :k3] = 0
                ms_cache[ck] = (rs, buf)
            rs, a3 = ms_cache[ck]
            if nb0 < nb:
                hip.attention(q3, kv3, kv3[:, k3:], a3[m0:], batch=nb - nb0, heads=3 * self.heads, hd=hd, nq=n, nk=la, scale=hd ** -0.5,
                              q_str=(n * k3, 0, k3), k_str=(la * 2 * k3, 0, 2 * k3), v_str=(la * 2 * k3, 0, 2 * k3), o_str=(n * kp, 0, kp),
                              out_scale=rs[:, m0:], out_scale_heads=self.heads)
            key = (t + ".zsum", scales)
            if key not in self._zbias:
                self._zbias[key] = sum(self.w[f"{t}.z{i}.bias"] * scales[i] for i in range(3)).contiguous()
            if 0 < m0 < m:
                # zero-audio rows: their operand is zero up to the mask columns -- a K = 64 GEMM against the bias columns of the weight image
                out = torch.empty_like(hid)
                hip.gemm(a3[:m0, k3:], self.w[t + ".oz3.wb"], self._zbias[key], residual=hid[:m0], out=out[:m0])
                hip.gemm(a3[m0:], self.w[t + ".oz3.w"], self._zbias[key], residual=hid[m0:], out=out[m0:])
                hid = out
            elif m0 == m:
                hid = hip.gemm(a3[:, k3:], self.w[t + ".oz3.wb"], self._zbias[key], residual=hid)
            else:
                hid = hip.gemm(a3, self.w[t + ".oz3.w"], self._zbias[key], residual=hid)
        else:
            a3 = torch.empty_like(q3)
            hip.attention(q3, kv3, kv3[:, 3 * inner:], a3, batch=nb, heads=3 * self.heads, hd=hd, nq=n, nk=la,
                          scale=hd ** -0.5, q_str=(n * 3 * inner, 0, 3 * inner), k_str=(la * 6 * inner, 0, 6 * inner),
                          v_str=(la * 6 * inner, 0, 6 * inner), o_str=(n * 3 * inner, 0, 3 * inner))
            for i in range(3):
                mask = masks[i][depth].reshape(-1).to(device=self._device, dtype=torch.float32).contiguous()
                if mask.numel() != m:
                    raise RuntimeError(f"mask level {depth} has {mask.numel()} entries, block has {m} tokens")
                s = scales[i]
                key = (f"{t}.z{i}", s)
                if key not in self._zbias:
                    self._zbias[key] = (self.w[f"{t}.z{i}.bias"] * s).contiguous()
                hid = hip.gemm_post(a3[:, i * inner:(i + 1) * inner], self.w[f"{t}.oz{i}.w"], self.w[f"{t}.oz{i}.bias"], mask, s,
                                    self._zbias[key], hid)
        out = self._norm_ff_proj_out(t + ".ff", t + ".norm3", hid, p, x.view(m, c))
        return out.view(nb, h, ww, c)

    def _motion_module(self, p, x, frames):
        """VanillaTemporalModule (motion_module.py:146-182,236-259,351-388): attention runs over the frame axis in
        place on the ((b f), hw, c) layout (batch = (b, pixel), token stride hw*c)."""
        nb, h, ww, c = x.shape
        n = h * ww
        m = nb * n
        b = nb // frames
        hd = c // self.heads
        if frames > 32:
            raise RuntimeError("temporal attention window is limited to 32 frames (positional encoding max_len)")
        q = p + ".temporal_transformer"
        t = q + ".transformer_blocks.0"
        hid = self._norm_proj_in(q, x)
        for i in range(2):
            a = f"{t}.attention_blocks.{i}"
            if (a + ".tleg_img") in self.w and (f"{t}.norms.{i}.bpe") in self.w and hip.temporal_leg320_supported(self._dtype, c, self.heads, frames, n, b):
                hid = hip.temporal_leg320(hid, self.w[f"{t}.norms.{i}.g"], self.w[f"{t}.norms.{i}.bpe"], self.w[a + ".tleg_img"], self.w[a + ".o.bias"],
                                          b, frames, n, hd ** -0.5)
                continue
            if (a + ".qkv_img") in self.w and n % 128 == 0:
                qkv, _ = hip.rowgemm320(hid, self.w[a + ".qkv_img"], 3 * c, ln_gamma=self.w[f"{t}.norms.{i}.g"],
                                        ln_beta=self.w[f"{t}.norms.{i}.bpe"], pe_div=n, pe_mod=frames)
            else:
                nrm = hip.layernorm(hid, self.w[f"{t}.norms.{i}.g"], self.w[f"{t}.norms.{i}.bpe"], 1e-5, pe_div=n, pe_mod=frames)
                qkv = hip.gemm(nrm, self.w[a + ".qkv.w"])
            o = torch.empty((m, c), device=self._device, dtype=self._dtype)
            st = (frames * n * 3 * c, 3 * c, n * 3 * c)
            hip.attention(qkv, qkv[:, c:], qkv[:, 2 * c:], o, batch=b * n, heads=self.heads, hd=hd, nq=frames, nk=frames,
                          scale=hd ** -0.5, q_str=st, k_str=st, v_str=st, o_str=(frames * n * c, c, n * c), bdiv=n)
            hid = hip.gemm(o, self.w[a + ".o.w"], self.w[a + ".o.bias"], residual=hid)
        out = self._norm_ff_proj_out(t + ".ff", t + ".ff_norm", hid, q, x.view(m, c))
        return out.view(nb, h, ww, c)

    # ------------------------------------------------------------------------------------------ forward
    def _time_embedding(self, timestep, batch):
        if not torch.is_tensor(timestep):
            ts = torch.tensor([float(timestep)], dtype=torch.float32, device=self._device)
        else:
            ts = timestep.to(device=self._device, dtype=torch.float32).reshape(-1)
        if ts.numel() == 1:
            batch = 1                 # one timestep for every row (the sampler's case): one embedding row, and its per-resnet slices are views
        ts = ts.expand(batch).contiguous()
        feat = hip.timestep_features(ts, self.boc[0], self._dtype)
        e = self._lin("time_embedding.linear_1", feat, act=hip.ACT_SILU)
        e = self._lin("time_embedding.linear_2", e)
        # every ResnetBlock3D's time_emb_proj(silu(emb)) in one GEMM (resnet.py:225-226)
        all_t = hip.gemm(hip.silu(e), self.w["temb_all.w"], self.w["temb_all.bias"]).float()
        return {p: all_t[:, off:off + n].contiguous() for p, (off, n) in self._temb_slices.items()}

    def forward(self, sample, timestep, encoder_hidden_states, audio_embedding=None, class_labels=None,
                mask_cond_fea=None, pose_cond_fea=None, attention_mask=None, full_mask=None, face_mask=None,
                body_mask=None, motion_scale=None, down_block_additional_residuals=None,
                mid_block_additional_residual=None, return_dict: bool = True):
        """UNet3DConditionModel.forward (unet_3d.py:425-625): (b, 4, f, h, w) in, (b, 4, f, h, w) out."""
        if attention_mask is not None or class_labels is not None or down_block_additional_residuals is not None \
                or mid_block_additional_residual is not None:
            raise NotImplementedError("attention_mask / class_labels / additional residuals are not used on this path")
        x = self.denoise_window(sample, timestep, encoder_hidden_states, audio_embedding, pose_cond_fea, full_mask,
                                face_mask, body_mask, motion_scale)
        out = hip.nhwc_to_ncfhw(x, sample.shape[0], self.out_channels).to(sample.dtype)
        if not return_dict:
            return (out,)
        return UNet3DConditionOutput(sample=out)

    __call__ = forward

    def denoise_window(self, sample, timestep, encoder_hidden_states, audio_embedding=None, pose_cond_fea=None,
                       full_mask=None, face_mask=None, body_mask=None, motion_scale=None, cfg_row=None, window_state=None,
                       audio_zero_rows=0, cfg_rows_share_input=False):
        """The operator body; returns the prediction channels-last ((b f), h, w, 64) with the first 4 channels valid
        (what mmgt_accumulate_window consumes, so the sampler never converts layouts).
        window_state: a dict the CALLER owns, one per (window, CFG row) whose audio / masks / motion_scale do not change between
        DDIM steps (the sampler's windows never move, pipeline_pose2vid_long.py:534-543): the operator keeps what it derives from those
        inputs alone in it -- the audio K / V projections of the six audio modules, the mask rows of MM-HAA -- instead of recomputing
        them at every step.  None: nothing is kept.
        cfg_rows_share_input: the CALLER's statement that the two rows of `sample` (and of the pose features) are copies of each other, as
        the sampler builds them; conv_in and the first resnet then run once.
        audio_zero_rows: the CALLER's statement that the audio embedding of the first `audio_zero_rows` batch rows is all zero (the
        unconditional CFG row, pipeline_pose2vid_long.py:484-485); their audio cross-attention is exactly 0 and is not computed.
        cfg_row (0 or 1): `sample`, the audio, pose and masks hold ONE CFG row (b = 1) -- the unconditional row never reads
        the reference banks, the conditional row reads them in every frame; `encoder_hidden_states` stays the (2, 1, 768)
        pair.  The window-parallel sampler deals the two rows of a window to different GPUs."""
        if cfg_row is not None and (cfg_row not in (0, 1) or sample.shape[0] != 1 or encoder_hidden_states.shape[0] != 2):
            raise RuntimeError("cfg_row: one CFG row (b = 1) with the (2, ...) encoder_hidden_states pair")
        if not self._loaded:
            raise RuntimeError("UNet3DConditionModel.forward before load_state_dict")
        if not sample.is_cuda:
            raise RuntimeError("mmgt_amd.UNet3DConditionModel runs on the GPU only (no CPU path exists)")
        b, cin, f, hh, ww = sample.shape
        if hh % 8 or ww % 8:
            raise RuntimeError("latent height/width must be multiples of 8 (unet_3d.py:461-469)")
        lpb = self.config.layers_per_block
        # scripts leave the module in train() + gradient checkpointing => motion_scale applied
        ms = motion_scale if (self.training and self.gradient_checkpointing) else None
        x_in = sample.to(torch.float32).contiguous()
        if self.config.center_input_sample:
            x_in = 2 * x_in - 1.0
        temb = self._time_embedding(timestep, b)
        x = hip.ncfhw_to_nhwc(x_in, 64, self._dtype)
        pose = None
        if pose_cond_fea is not None:
            if pose_cond_fea.dim() == 4:      # already channels-last ((b f), h, w, 320) in the model dtype (PoseGuider.forward_nhwc)
                pose = pose_cond_fea.to(self._dtype).contiguous()
            else:
                pose = hip.ncfhw_to_nhwc(pose_cond_fea.to(torch.float32).contiguous(), self.boc[0], self._dtype)
        # The two CFG rows of a pair enter with the SAME latents, pose features and timestep (the caller says so: cfg_rows_share_input);
        # they stay identical until the first transformer reads the per-row conditioning, so conv_in and the first ResnetBlock3D are
        # computed once for the f frames and duplicated (pipeline_pose2vid_long.py:554-580: `latent_model_input = latents.repeat(2 ...)`).
        shared = bool(cfg_rows_share_input) and b == 2 and cfg_row is None and self._share_rows
        if shared:
            x2 = torch.empty((2 * f, hh, ww, self.boc[0]), device=self._device, dtype=self._dtype)
            x = hip.conv3x3(x[:f], self.w["conv_in.w"], self.w["conv_in.bias"], residual=None if pose is None else pose[:f], out=x2[:f])
            x2[f:].copy_(x)                             # (one half-tensor copy; torch.cat would read and write both halves)
        else:
            x = hip.conv3x3(x, self.w["conv_in.w"], self.w["conv_in.bias"], residual=pose)
        audio = None
        if audio_embedding is not None:
            audio = audio_embedding.to(device=self._device, dtype=self._dtype).reshape(b * f, *audio_embedding.shape[2:])
            audio = audio.contiguous()
        masks = (full_mask, face_mask, body_mask)
        if not 0 <= int(audio_zero_rows) <= b:
            raise RuntimeError(f"audio_zero_rows = {audio_zero_rows} outside 0..{b}")
        # per-forward mask rows / operand buffers of the audio modules, by level
        ms_cache = {"state": window_state, "zero_images": int(audio_zero_rows) * f}
        ehs = encoder_hidden_states.to(self._device)

        skips = [x2 if shared else x]
        for i in range(4):
            p = f"down_blocks.{i}"
            for j in range(lpb):
                if shared and i == 0 and j == 0:
                    r = f"{p}.resnets.0"
                    x2 = torch.empty((2 * f,) + tuple(x.shape[1:3]) + (self.spec[r + ".conv1.weight"][0],), device=self._device, dtype=self._dtype)
                    x = self._resnet(r, x, {r: temb[r][:1]}, out=x2[:f], reader=f"{p}.attentions.0.norm" if self._twin else None)
                    x2[f:].copy_(x)
                    x = x2
                else:
                    x = self._resnet(f"{p}.resnets.{j}", x, temb, reader=f"{p}.attentions.{j}.norm" if i < 3 else None)
                if i < 3:
                    y = self._spatial_transformer_twin(f"{p}.attentions.{j}", x, ehs, f) if (shared and i == 0 and j == 0 and self._twin) else None
                    x = y if y is not None else self._spatial_transformer(f"{p}.attentions.{j}", x, ehs, f, cfg_row=cfg_row)
                    if f"{p}.audio_modules.{j}" in self._audio:
                        x = self._audio_transformer(f"{p}.audio_modules.{j}", x, audio, masks, i, ms, ms_cache)
                x = self._motion_module(f"{p}.motion_modules.{j}", x, f)
                skips.append(x)
            if i != 3:
                x = hip.conv3x3(x, self.w[f"{p}.downsamplers.0.conv.w"], self.w[f"{p}.downsamplers.0.conv.bias"], stride=2)
                skips.append(x)

        x = self._resnet("mid_block.resnets.0", x, temb)
        x = self._spatial_transformer("mid_block.attentions.0", x, ehs, f, cfg_row=cfg_row)
        x = self._motion_module("mid_block.motion_modules.0", x, f)
        x = self._resnet("mid_block.resnets.1", x, temb)

        for i in range(4):
            p = f"up_blocks.{i}"
            for j in range(lpb + 1):
                x = self._resnet(f"{p}.resnets.{j}", x, temb, skip=skips.pop(), reader=f"{p}.attentions.{j}.norm" if i > 0 else None)
                if i > 0:
                    x = self._spatial_transformer(f"{p}.attentions.{j}", x, ehs, f, cfg_row=cfg_row)
                x = self._motion_module(f"{p}.motion_modules.{j}", x, f)
            if i != 3:
                w2 = self.w.get(f"{p}.upsamplers.0.conv.w2")
                if w2 is not None:
                    x = hip.conv3x3(x, w2, self.w[f"{p}.upsamplers.0.conv.bias"], upsample=2)
                else:
                    x = hip.conv3x3(x, self.w[f"{p}.upsamplers.0.conv.w"], self.w[f"{p}.upsamplers.0.conv.bias"], upsample=True)

        timg = self.w.get("conv_out.timg")
        nb, h, ww, c = x.shape
        if timg is not None and c == 320 and (h * ww) % 128 == 0:
            sc, sh = hip.groupnorm_affine(x.view(nb, h * ww, c), self.w["conv_norm_out.g"], self.w["conv_norm_out.b"], 32, self.config.norm_eps)
            y, _ = hip.rowgemm320(x.view(nb * h * ww, c), timg, 64, pre_scale=sc, pre_shift=sh, pre_rows=h * ww, pre_silu=True)
            return hip.conv_taps_gather(y, self.w["conv_out.bias"], nb, h, ww)
        x = self._gn("conv_norm_out", x, self.config.norm_eps, silu=True)
        x = hip.conv3x3(x, self.w["conv_out.w"], self.w["conv_out.bias"])
        return x
