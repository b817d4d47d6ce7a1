"""AutoencoderKL (sd-vae-ft-mse) on MI355X: the `decode_latents` leg of the Stage-2 sampler and the encoder that turns the
reference image into `ref_image_latents` once per clip (pipeline_pose2vid_long.py:427-434).

Mirrors what src/pipelines/pipeline_pose2vid_long.py:112-125 asks of diffusers' `AutoencoderKL.decode`, with the
diffusers 0.24.0 state-dict key names so `sd-vae-ft-mse` checkpoints load by name.  All frames of a clip are decoded as
one channels-last batch (the reference loops frame by frame; every op is per-frame, so results are identical) on the same
HIP kernels as the UNet: implicit-GEMM conv3x3 (fused nearest-2x upsample), GroupNorm+SiLU, GEMM; the single 512-wide
attention head of the mid block uses materialised scores (GEMM -> row softmax -> GEMM), 34 GFLOP per frame, whose logits keep
~17 bits in the bf16 model through hi / lo operand pieces (`_mid_attention_split`).
"""
from collections import OrderedDict

import torch

from . import hip
from .packing import pack_conv3x3, pack_conv3x3_up2, pack_gnconv, pack_rconv, pad_cols, pad_rows, round_up

_UP_CH = ((512, 512), (512, 512), (512, 256), (256, 128))     # (in, out) of decoder.up_blocks.0..3


def vae_decoder_spec():
    s = OrderedDict()

    def norm(p, c):
        s[p + ".weight"] = (c,)
        s[p + ".bias"] = (c,)

    def conv(p, cin, cout, k=3):
        s[p + ".weight"] = (cout, cin, k, k)
        s[p + ".bias"] = (cout,)

    def resnet(p, cin, cout):
        norm(p + ".norm1", cin)
        conv(p + ".conv1", cin, cout)
        norm(p + ".norm2", cout)
        conv(p + ".conv2", cout, cout)
        if cin != cout:
            conv(p + ".conv_shortcut", cin, cout, 1)

    conv("post_quant_conv", 4, 4, 1)
    conv("decoder.conv_in", 4, 512)
    resnet("decoder.mid_block.resnets.0", 512, 512)
    a = "decoder.mid_block.attentions.0"
    norm(a + ".group_norm", 512)
    for n in ("to_q", "to_k", "to_v", "to_out.0"):
        s[f"{a}.{n}.weight"] = (512, 512)
        s[f"{a}.{n}.bias"] = (512,)
    resnet("decoder.mid_block.resnets.1", 512, 512)
    for i, (cin, cout) in enumerate(_UP_CH):
        for j in range(3):
            resnet(f"decoder.up_blocks.{i}.resnets.{j}", cin if j == 0 else cout, cout)
        if i != 3:
            conv(f"decoder.up_blocks.{i}.upsamplers.0.conv", cout, cout)
    norm("decoder.conv_norm_out", 128)
    conv("decoder.conv_out", 128, 3)
    return s


_DOWN_CH = ((128, 128), (128, 256), (256, 512), (512, 512))   # (in, out) of encoder.down_blocks.0..3


def vae_encoder_spec():
    """diffusers 0.24.0 `AutoencoderKL` encoder + quant_conv key names (double_z: 8 moment channels)."""
    s = OrderedDict()

    def norm(p, c):
        s[p + ".weight"] = (c,)
        s[p + ".bias"] = (c,)

    def conv(p, cin, cout, k=3):
        s[p + ".weight"] = (cout, cin, k, k)
        s[p + ".bias"] = (cout,)

    def resnet(p, cin, cout):
        norm(p + ".norm1", cin)
        conv(p + ".conv1", cin, cout)
        norm(p + ".norm2", cout)
        conv(p + ".conv2", cout, cout)
        if cin != cout:
            conv(p + ".conv_shortcut", cin, cout, 1)

    conv("encoder.conv_in", 3, 128)
    for i, (cin, cout) in enumerate(_DOWN_CH):
        for j in range(2):
            resnet(f"encoder.down_blocks.{i}.resnets.{j}", cin if j == 0 else cout, cout)
        if i != 3:
            conv(f"encoder.down_blocks.{i}.downsamplers.0.conv", cout, cout)
    resnet("encoder.mid_block.resnets.0", 512, 512)
    a = "encoder.mid_block.attentions.0"
    norm(a + ".group_norm", 512)
    for n in ("to_q", "to_k", "to_v", "to_out.0"):
        s[f"{a}.{n}.weight"] = (512, 512)
        s[f"{a}.{n}.bias"] = (512,)
    resnet("encoder.mid_block.resnets.1", 512, 512)
    norm("encoder.conv_norm_out", 512)
    conv("encoder.conv_out", 512, 8)
    conv("quant_conv", 8, 8, 1)
    return s


class _Cfg:
    block_out_channels = (128, 256, 512, 512)
    scaling_factor = 0.18215


class DecoderOutput:
    def __init__(self, sample):
        self.sample = sample


class AutoencoderKL:
    """Decoder (every clip, all frames) and encoder (once per clip, the reference image: prologue).  The
    encoder keys are optional in `load_state_dict`: without them `encode_mean` raises."""
    config = _Cfg()

    def __init__(self, device="cuda", dtype=torch.bfloat16):
        self._device, self._dtype = torch.device(device), dtype
        hip.dtype_code(dtype)
        self.spec = vae_decoder_spec()
        self.encoder_spec = vae_encoder_spec()
        self.w = {}
        self._loaded = False
        self._has_encoder = False
        self._split_attention = True       # False: the fp32 kernels for the mid-block attention of a bf16 model too (A/B, tests)

    @property
    def dtype(self):
        return self._dtype

    @property
    def device(self):
        return self._device

    def to(self, *a, **k):
        return self

    def _t(self, x):
        return x.to(device=self._device, dtype=self._dtype).contiguous()

    def _f(self, x):
        return x.to(device=self._device, dtype=torch.float32).contiguous()

    def load_state_dict(self, sd, strict=True):
        missing = [k for k in self.spec if k not in sd]
        if missing:
            raise RuntimeError(f"AutoencoderKL.load_state_dict: missing {len(missing)} decoder keys, e.g. {missing[:3]}")
        w = self.w
        enc_present = [k for k in self.encoder_spec if k in sd]
        if enc_present and len(enc_present) != len(self.encoder_spec):
            raise RuntimeError(f"AutoencoderKL.load_state_dict: {len(self.encoder_spec) - len(enc_present)} encoder keys missing")
        spec = OrderedDict(self.spec)
        if enc_present:
            spec.update(self.encoder_spec)
        for k, shape in spec.items():
            if tuple(sd[k].shape) != tuple(shape):
                raise RuntimeError(f"shape mismatch for {k}: {tuple(sd[k].shape)} vs {shape}")
        for k in spec:
            if not k.endswith(".weight"):
                continue
            p = k[:-len(".weight")]
            wt, b = sd[k], sd[p + ".bias"]
            if wt.dim() == 1:                                        # GroupNorm affine
                w[p + ".g"], w[p + ".b"] = self._f(wt), self._f(b)
            elif wt.dim() == 4 and wt.shape[2] == 3:                 # conv3x3 (channels padded to multiples of 64)
                cin_pad, cout_pad = round_up(wt.shape[1], 64), round_up(wt.shape[0], 64)
                w[p + ".w"] = self._t(pack_conv3x3(wt, cin_pad, cout_pad))
                w[p + ".bias"] = self._f(pad_rows(b, cout_pad))
                if ".upsamplers." in p and self._dtype == torch.bfloat16 and hip.tune_get("up2") and wt.shape[0] % 256 == 0 and wt.shape[1] % 64 == 0:
                    # the conv behind the nearest 2x upsampling as four 2 x 2 convs on the stored image (packing.pack_conv3x3_up2): 16 / 36 of the work
                    w[p + ".w2"] = self._t(pack_conv3x3_up2(wt.to(self._device, torch.float32)))
                if self._dtype == torch.bfloat16 and (wt.shape[1], cout_pad) in ((128, 128), (256, 128), (256, 256), (128, 64)):
                    # the 512 x 512 and 256 x 256 levels: GroupNorm + SiLU + conv fused (csrc/gnconv.hip) wherever a GroupNorm feeds this conv
                    wpad = torch.zeros((cout_pad, wt.shape[1], 3, 3), device=self._device, dtype=torch.float32)
                    wpad[:wt.shape[0]] = wt.to(self._device, torch.float32)
                    w[p + ".gimg"] = pack_gnconv(wpad)
                elif self._dtype == torch.bfloat16 and wt.shape[1] % 64 == 0 and wt.shape[0] % 256 == 0:
                    # the 512-channel levels (and 512 -> 256): GroupNorm + SiLU + conv fused in 64-channel phases over 256-wide output blocks (csrc/rconv.hip)
                    w[p + ".rimg"] = pack_rconv(wt.to(self._device, torch.float32))
            elif wt.dim() == 4:                                      # 1x1 conv as GEMM
                m = wt.reshape(wt.shape[0], -1)
                w[p + ".w"] = self._t(pad_rows(pad_cols(m, round_up(m.shape[1], 64)), round_up(m.shape[0], 64)))
                w[p + ".bias"] = self._f(pad_rows(b, round_up(m.shape[0], 64)))
        for a in ["decoder.mid_block.attentions.0"] + (["encoder.mid_block.attentions.0"] if enc_present else []):
            # The single 512-wide head: with real sd-vae-ft-mse weights the logits reach hundreds, where a bf16 score -- or a score of
            # bf16-rounded q / k -- has an ulp of 1-2; the reference runs this block in fp16 / fp32.
            #   fp32 model: the fp32 instantiation of the kernels throughout (materialised scores, 34 GFLOP per frame);
            #   bf16 model: the logits keep ~17 bits through hi / lo bf16 operand pieces on the bf16 MFMA path (_mid_attention_split):
            #               W = W_hi + W_lo for the q / k projections, q = q_hi + q_lo and k likewise for the scores.
            w[a + ".q.w"], w[a + ".q.bias"] = self._f(sd[a + ".to_q.weight"]), self._f(sd[a + ".to_q.bias"])
            w[a + ".k.w"], w[a + ".k.bias"] = self._f(sd[a + ".to_k.weight"]), self._f(sd[a + ".to_k.bias"])
            w[a + ".v.w"] = self._f(sd[a + ".to_v.weight"])
            w[a + ".o.w"] = self._f(sd[a + ".to_out.0.weight"])
            # softmax rows sum to 1, so P (V + 1 b_v^T) = P V + b_v^T: the value bias moves into the output bias
            # (W_o b_v as an fp32 GEMM with one output column, on the device like every other product of this library)
            wob = hip.gemm(w[a + ".o.w"], self._f(sd[a + ".to_v.bias"])[None, :].contiguous()).reshape(-1)
            w[a + ".o.bias"] = (wob + self._f(sd[a + ".to_out.0.bias"])).contiguous()
            if self._dtype == torch.bfloat16:
                def hi_lo(m):
                    hi = m.to(torch.bfloat16)
                    return hi, (m - hi.float()).to(torch.bfloat16)
                qh, ql = hi_lo(w[a + ".q.w"])
                kh, kl = hi_lo(w[a + ".k.w"])
                w[a + ".qk.w4"] = torch.cat([qh, ql, kh, kl], 0).contiguous()            # (4C, C): rows = the four output blocks
                w[a + ".v.wb"] = w[a + ".v.w"].to(torch.bfloat16).contiguous()
                w[a + ".o.wb"] = w[a + ".o.w"].to(torch.bfloat16).contiguous()
        self._loaded = True
        self._has_encoder = bool(enc_present)
        return [], [k for k in sd if k not in spec]

    # ------------------------------------------------------------------------------------------------ blocks
    def _gn(self, p, x, silu):
        nb, h, ww, c = x.shape
        return hip.groupnorm(x.view(nb, h * ww, c), self.w[p + ".g"], self.w[p + ".b"], 32, 1e-6, silu=silu).view(nb, h, ww, c)

    def _gn_silu_conv(self, pn, pc, x, residual=None, stats=None, next_pn=None):
        """conv3x3(silu(GroupNorm(x))) (+ residual): diffusers `ResnetBlock2D.forward`'s norm -> nonlinearity -> conv.  In a bf16 model the convs
        of the 512 x 512 and 256 x 256 levels (128 / 256 input channels; conv_out padded to 64 outputs) run as ONE fused launch per 128 output
        channels (csrc/gnconv.hip) and those of the 512-channel levels as one launch in 64-channel phases (csrc/rconv.hip, round 6): the normalised
        tensor is never written.  The GroupNorm statistics come from `stats` -- what the launch that PRODUCED x left beside it: ("tiles", per-tile
        partial sums) of a gnconv launch, ("tables", (scale, shift)) of an rconv launch that was told the norm (`next_pn`) -- or from one pass over x.
        Everything else runs GroupNorm and conv as two launches.  Returns (out, stats of out or None)."""
        nb, h, ww, c = x.shape
        cout = self.w[pc + ".bias"].numel()
        mode = hip.tune_get("gnconv")
        chain = mode >= 2                                        # (1: every fused launch behind its own statistics pass)
        g, b = self.w[pn + ".g"], self.w[pn + ".b"]
        tables = None
        if stats is not None and chain:
            tables = stats[1] if stats[0] == "tables" else hip.gn_tables_from_stats(stats[1], g, b, 32, 1e-6, nb, c)
        if mode and (pc + ".gimg") in self.w and hip.gn_silu_conv3x3_supported(x.dtype, c, cout, h, ww, residual is not None) and h * ww > 256:
            r = hip.gn_silu_conv3x3(x, g, b, 32, 1e-6, self.w[pc + ".gimg"], cout, self.w[pc + ".bias"], residual, tables=tables, want_stats=chain)
            return (r[0], None if r[1] is None else ("tiles", r[1])) if chain else (r, None)
        if mode and (pc + ".rimg") in self.w and hip.gn_silu_conv3x3_unet_supported(x.dtype, c, 0, cout, h, ww) and \
                x.numel() * 2 <= hip.DMA_LIMIT and nb * h * ww * cout * 2 <= hip.DMA_LIMIT:
            sc, sh = tables if tables is not None else hip.groupnorm_affine(x.view(nb, h * ww, c), g, b, 32, 1e-6)
            nn = (self.w[next_pn + ".g"], self.w[next_pn + ".b"], 32, 1e-6) if next_pn and chain else None
            r = hip.gn_silu_conv3x3_unet(x, sc, sh, self.w[pc + ".rimg"], cout, self.w[pc + ".bias"], residual=residual, next_norm=nn)
            return (r[0], ("tables", r[1])) if nn else (r, None)
        return hip.conv3x3(self._gn(pn, x, True), self.w[pc + ".w"], self.w[pc + ".bias"], residual=residual), None

    def _resnet(self, p, x, stats=None, next_pn=None):
        """-> (out, statistics of out or None); `stats`: those of x, from the launch that produced it; next_pn: the GroupNorm that reads the result"""
        nb, h, ww, cin = x.shape
        hdn, st = self._gn_silu_conv(p + ".norm1", p + ".conv1", x, stats=stats, next_pn=p + ".norm2")
        res = x
        if (p + ".conv_shortcut.w") in self.w:
            res = hip.gemm(x.view(nb * h * ww, cin), self.w[p + ".conv_shortcut.w"], self.w[p + ".conv_shortcut.bias"])
            res = res.view(nb, h, ww, -1)
        return self._gn_silu_conv(p + ".norm2", p + ".conv2", hdn, residual=res, stats=st, next_pn=next_pn)

    def _mid_attention(self, x, a="decoder.mid_block.attentions.0"):
        if self._split_attention and self._dtype == torch.bfloat16 and x.shape[1] * x.shape[2] % 256 == 0 and x.shape[1] * x.shape[2] <= 8192:
            return self._mid_attention_split(x, a)
        nb, h, ww, c = x.shape
        n = h * ww
        t = self._gn(a + ".group_norm", x, False).view(nb * n, c).float()                    # fp32 from here (see load_state_dict)
        q = hip.gemm(t, self.w[a + ".q.w"], self.w[a + ".q.bias"]).view(nb, n, c)
        k = hip.gemm(t, self.w[a + ".k.w"], self.w[a + ".k.bias"]).view(nb, n, c)
        vt = torch.empty((nb, c, n), device=self._device, dtype=torch.float32)
        hip.gemm_batched_wx(self.w[a + ".v.w"], t.view(nb, n, c), out=vt)                    # V^T without its bias
        s_ = torch.empty((nb, n, n), device=self._device, dtype=torch.float32)
        hip.gemm_batched(q, k, out=s_)                                                        # scores, one frame per z
        hip.softmax_rows(s_.view(nb * n, n), c ** -0.5, out=s_.view(nb * n, n))
        o = torch.empty((nb, n, c), device=self._device, dtype=torch.float32)
        hip.gemm_batched(s_, vt, out=o)
        out = hip.gemm(o.view(nb * n, c), self.w[a + ".o.w"], self.w[a + ".o.bias"], residual=x.view(nb * n, c).float())
        return out.to(self._dtype).view(nb, h, ww, c)

    def _mid_attention_split(self, x, a):
        """The bf16 model's mid-block attention on the bf16 MFMA path.  Only the logits need more than a bf16 mantissa, and they get it
        from operand PIECES: t (bf16, exact) times [Wq_hi; Wq_lo; Wk_hi; Wk_lo] with fp32 accumulators out gives q and k to fp32 accuracy;
        q = q_hi + q_lo, k = k_hi + k_lo then give q . k = q_hi k_hi + q_hi k_lo + q_lo k_hi (+ O(2^-17)) as ONE bf16 GEMM over the
        concatenated reduction [q_hi | q_hi | q_lo] . [k_hi | k_lo | k_hi] (K = 3 C), fp32 logits out.  softmax -> bf16 probabilities in
        one pass; P V and the output projection are ordinary bf16 GEMMs (their operands are in [0, 1] / activations like any other layer's)."""
        nb, h, ww, c = x.shape
        n = h * ww
        w = self.w
        t = self._gn(a + ".group_norm", x, False).view(nb * n, c)
        qk = hip.gemm_bf16_f32(t, w[a + ".qk.w4"])                                             # (nb n, 4C) fp32
        Qp, Kp = hip.qk_split3(qk, w[a + ".q.bias"], w[a + ".k.bias"])                         # (nb n, 3C) bf16 each
        del qk
        vt = torch.empty((nb, c, n), device=self._device, dtype=torch.bfloat16)
        hip.gemm_batched_wx(w[a + ".v.wb"], t.view(nb, n, c), out=vt)                          # V^T without its bias
        p = torch.empty((nb, n, n), device=self._device, dtype=torch.bfloat16)
        s_ = torch.empty((n, n), device=self._device, dtype=torch.float32)
        for f in range(nb):                                                                    # 256 tiles of 256 x 256 per frame: one per CU
            hip.gemm_bf16_f32(Qp[f * n:(f + 1) * n], Kp[f * n:(f + 1) * n], out=s_)
            hip.softmax_rows_f32_bf16(s_, c ** -0.5, out=p[f])
        o = torch.empty((nb, n, c), device=self._device, dtype=torch.bfloat16)
        hip.gemm_batched(p, vt, out=o)
        out = hip.gemm(o.view(nb * n, c), w[a + ".o.wb"], w[a + ".o.bias"], residual=x.view(nb * n, c))
        return out.view(nb, h, ww, c)

    # ------------------------------------------------------------------------------------------------ API
    def decode_nhwc(self, z):
        """z: (nb, h, w, 64) channels-last latents (4 valid channels, already divided by 0.18215) -> (nb, 8h, 8w, 64)
        with 3 valid channels."""
        if not self._loaded:
            raise RuntimeError("AutoencoderKL.decode before load_state_dict")
        nb, h, ww, _ = z.shape
        x = hip.gemm(z.view(nb * h * ww, 64), self.w["post_quant_conv.w"], self.w["post_quant_conv.bias"]).view(nb, h, ww, 64)
        x = hip.conv3x3(x, self.w["decoder.conv_in.w"], self.w["decoder.conv_in.bias"])
        x, _ = self._resnet("decoder.mid_block.resnets.0", x)
        x = self._mid_attention(x)
        x, _ = self._resnet("decoder.mid_block.resnets.1", x)
        st = None                                            # statistics of x written by the launch that produced it (the fused launches only)
        for i in range(4):
            for j in range(3):
                nxt = f"decoder.up_blocks.{i}.resnets.{j + 1}.norm1" if j < 2 else "decoder.conv_norm_out" if i == 3 else None
                x, st = self._resnet(f"decoder.up_blocks.{i}.resnets.{j}", x, st, next_pn=nxt)
            if i != 3:
                p = f"decoder.up_blocks.{i}.upsamplers.0.conv"
                w2 = self.w.get(p + ".w2")
                if w2 is not None:
                    x, st = hip.conv3x3(x, w2, self.w[p + ".bias"], upsample=2), None
                else:
                    x, st = hip.conv3x3(x, self.w[p + ".w"], self.w[p + ".bias"], upsample=True), None
        return self._gn_silu_conv("decoder.conv_norm_out", "decoder.conv_out", x, stats=st)[0]

    def encode_nhwc(self, x):
        """x: (nb, H, W, 64) channels-last image in [-1, 1] (3 valid channels) -> (nb, H/8, W/8, 64) moments (channels
        0..3 = mean, 4..7 = logvar) after quant_conv.  diffusers 0.24.0 `Encoder.forward` + `quant_conv`."""
        if not self._has_encoder:
            raise RuntimeError("AutoencoderKL.encode: the encoder weights were not loaded")
        x = hip.conv3x3(x, self.w["encoder.conv_in.w"], self.w["encoder.conv_in.bias"])
        for i in range(4):
            for j in range(2):
                x, _ = self._resnet(f"encoder.down_blocks.{i}.resnets.{j}", x)
            if i != 3:
                p = f"encoder.down_blocks.{i}.downsamplers.0.conv"      # F.pad(x, (0, 1, 0, 1)) + stride-2 conv, padding 0
                x = hip.conv3x3(x, self.w[p + ".w"], self.w[p + ".bias"], stride=2, pad_high_only=True)
        x, _ = self._resnet("encoder.mid_block.resnets.0", x)
        x = self._mid_attention(x, "encoder.mid_block.attentions.0")
        x, _ = self._resnet("encoder.mid_block.resnets.1", x)
        x = self._gn("encoder.conv_norm_out", x, True)
        x = hip.conv3x3(x, self.w["encoder.conv_out.w"], self.w["encoder.conv_out.bias"])
        nb, h, ww, _ = x.shape
        return hip.gemm(x.view(nb * h * ww, 64), self.w["quant_conv.w"], self.w["quant_conv.bias"]).view(nb, h, ww, 64)

    def encode_mean(self, x):
        """`vae.encode(x).latent_dist.mean` (pipeline_pose2vid_long.py:427-434): x (n, 3, H, W) in [-1, 1] -> (n, 4, H/8, W/8)
        fp32 on the GPU (the caller multiplies by 0.18215)."""
        xx = x.to(self._device, torch.float32).contiguous()[:, :, None]                         # (n, 3, 1, H, W)
        n = xx.shape[0]
        img = hip.ncfhw_to_nhwc(xx.permute(2, 1, 0, 3, 4).contiguous(), 64, self._dtype)        # images as the f axis
        mom = self.encode_nhwc(img)
        out = hip.nhwc_to_ncfhw(mom, 1, 4)                                                       # (1, 4, n, h, w)
        return out[0].permute(1, 0, 2, 3).contiguous()

    def decode(self, z):
        """diffusers-style: z (n, 4, h, w) -> DecoderOutput(sample (n, 3, 8h, 8w))."""
        zz = z.to(self._device, torch.float32).contiguous()[:, :, None]                         # (n, 4, 1, h, w)
        n = zz.shape[0]
        x = hip.ncfhw_to_nhwc(zz.permute(2, 1, 0, 3, 4).contiguous(), 64, self._dtype)          # frames as the f axis
        y = self.decode_nhwc(x)
        out = hip.nhwc_to_ncfhw(y, 1, 3)                                                        # (1, 3, n, H, W)
        return DecoderOutput(out[0].permute(1, 0, 2, 3).to(z.dtype))

    def decode_video(self, latents, frames_per_batch=8):
        """Pose2VideoPipeline.decode_latents (pipeline_pose2vid_long.py:112-125): latents (b, 4, f, h, w) ->
        (b, 3, f, 8h, 8w) fp32 in [0, 1] on the GPU; z / 0.18215 and (x / 2 + 0.5).clamp(0, 1) are fused into the layout
        kernels."""
        lat = latents.to(self._device, torch.float32).contiguous()
        b, c, f, h, ww = lat.shape
        outs = []
        for f0 in range(0, f, frames_per_batch):
            chunk = lat[:, :, f0:f0 + frames_per_batch].contiguous()
            x = hip.ncfhw_to_nhwc(chunk, 64, self._dtype, scale=1.0 / 0.18215)
            y = self.decode_nhwc(x)
            outs.append(hip.nhwc_to_ncfhw(y, b, 3, scale=0.5, shift=0.5, clamp01=True))
        return torch.cat(outs, dim=2)

    def decode_video_uint8(self, latents, frames_per_batch=8):
        """The output path on the device: latents (b, 4, f, h, w) -> uint8 frames (b, f, 8h, 8w, 3) on the GPU =
        save_videos_grid's (x * 255).astype(uint8) (src/utils/util.py:148-160) of decode_latents' clamped frames, fused into
        one conversion kernel per batch of frames: 4x fewer bytes cross PCIe than the fp32 (b, 3, f, H, W) video."""
        lat = latents.to(self._device, torch.float32).contiguous()
        b, c, f, h, ww = lat.shape
        outs = []
        for f0 in range(0, f, frames_per_batch):
            chunk = lat[:, :, f0:f0 + frames_per_batch].contiguous()
            x = hip.ncfhw_to_nhwc(chunk, 64, self._dtype, scale=1.0 / 0.18215)
            y = hip.frames_to_u8(self.decode_nhwc(x), 0.5, 0.5)                    # ((b fc), H, W, 3)
            outs.append(y.view(b, -1, *y.shape[1:]))
        return torch.cat(outs, dim=1)
