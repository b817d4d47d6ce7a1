#!/usr/bin/env python3
"""Experiment ("What's weak" 1): what would an fp32 RESIDUAL STREAM buy the bf16 product mode?

The oracle (CPU restatement of the reference UNet3D) is run three ways on the same weights / inputs:
  fp32      the reference arithmetic;
  bf16      every matmul takes bf16-rounded operands and accumulates in fp32 (what the MFMA kernels do), every tensor written to memory is
            rounded to bf16 -- the current product mode (its error is the measured floor the HIP kernels are gated against);
  bf16+f32  the same matmuls, but the tensors on the residual path (conv_in / proj_in outputs, every `x + f(x)` sum, the skips) stay
            fp32; LayerNorm / GroupNorm read them in fp32 and write bf16 -- an UPPER BOUND on what fp32 residual epilogues can buy (the
            block-internal tensors between a matmul and the following norm are left in fp32 too).
Prints max / mean |error| of the predicted noise against the fp32 run.   python tools/exp_fp32_residual.py [latent frames]"""
import os
import sys

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mmgt_amd.synthetic import synth_state_dict  # noqa: E402
from mmgt_amd.unet3d_spec import unet3d_spec  # noqa: E402
from oracle import unet3d_ref as R  # noqa: E402
from tests import golden_cases as gc  # noqa: E402

MODE = {"stream_fp32": False}
bf = lambda t: t.to(torch.bfloat16).float()
outr = lambda t: t if MODE["stream_fp32"] else bf(t)


def _lin(sd, p, x):
    b = sd.get(p + ".bias")
    return outr(F.linear(bf(x), bf(sd[p + ".weight"]), None if b is None else b.float()))


def _conv(sd, p, x, stride=1, padding=1):
    b = sd.get(p + ".bias")
    return outr(F.conv2d(bf(x), bf(sd[p + ".weight"]), None if b is None else b.float(), stride=stride, padding=padding))


def _gn(sd, p, x, groups, eps):
    return bf(F.group_norm(x.float() if MODE["stream_fp32"] else bf(x), groups, sd[p + ".weight"].float(), sd[p + ".bias"].float(), eps))


def _ln(sd, p, x):
    return bf(F.layer_norm(x.float() if MODE["stream_fp32"] else bf(x), (x.shape[-1],), sd[p + ".weight"].float(), sd[p + ".bias"].float(), 1e-5))


def attention(sd, p, x, ctx, heads):
    b, n, _ = x.shape
    mm = lambda a, w: bf(F.linear(bf(a), bf(sd[w])))                 # q, k, v are written to memory as bf16 in both modes
    q, k, v = mm(x, p + ".to_q.weight"), mm(ctx, p + ".to_k.weight"), mm(ctx, p + ".to_v.weight")
    hd = q.shape[-1] // heads
    sp = lambda t: t.view(b, -1, heads, hd).transpose(1, 2)
    o = bf(F.scaled_dot_product_attention(sp(q), sp(k), sp(v)).transpose(1, 2).reshape(b, n, heads * hd))
    return outr(F.linear(o, bf(sd[p + ".to_out.0.weight"]), sd[p + ".to_out.0.bias"].float()))


def feed_forward(sd, p, x):
    h, gate = bf(F.linear(bf(x), bf(sd[p + ".net.0.proj.weight"]), sd[p + ".net.0.proj.bias"].float())).chunk(2, dim=-1)   # (kept in registers / bf16)
    return outr(F.linear(bf(h * F.gelu(gate)), bf(sd[p + ".net.2.weight"]), sd[p + ".net.2.bias"].float()))


def run(sd, case, emulate, stream_fp32=False):
    inp = gc.unet_inputs(case)
    saved = {n: getattr(R, n) for n in ("_lin", "_conv", "_gn", "_ln", "attention", "feed_forward")}
    if emulate:
        MODE["stream_fp32"] = stream_fp32
        for n, f in (("_lin", _lin), ("_conv", _conv), ("_gn", _gn), ("_ln", _ln), ("attention", attention), ("feed_forward", feed_forward)):
            setattr(R, n, f)
    try:
        with torch.no_grad():
            out = R.unet3d_forward(sd, R.UNet3DConfig(), inp["sample"], inp["timestep"], inp["ehs"], inp["audio"], inp["pose"], inp["full"],
                                   inp["face"], inp["lips"], inp["motion_scale"], inp["banks"], weighted=True)
            if emulate and not stream_fp32:
                pass                      # (adds of bf16-valued fp32 tensors are NOT re-rounded by torch: round the stream below)
        return out.float()
    finally:
        for n, f in saved.items():
            setattr(R, n, f)


def main():
    latent, frames = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (8, 8)
    torch.set_num_threads(os.cpu_count() or 8)
    case = dict(gc.UNET_CASES["full_cfg1"], latent=latent, frames=frames)
    sd = synth_state_dict(unet3d_spec(), device="cpu")
    ref = run(sd, case, emulate=False)
    # the all-bf16 mode rounds the residual sums as well: that is torch's own bf16 arithmetic on bf16 tensors
    sd16 = {k: (v.bfloat16() if v.is_floating_point() else v) for k, v in sd.items()}
    inp = gc.unet_inputs(case)
    c = lambda t: t.bfloat16() if torch.is_tensor(t) and t.is_floating_point() else t
    with torch.no_grad():
        b16 = R.unet3d_forward(sd16, R.UNet3DConfig(), c(inp["sample"]), inp["timestep"], c(inp["ehs"]), c(inp["audio"]), c(inp["pose"]),
                               [c(x) for x in inp["full"]], [c(x) for x in inp["face"]], [c(x) for x in inp["lips"]], inp["motion_scale"],
                               {k: c(v) for k, v in inp["banks"].items()}, weighted=True).float()
    f32s = run(sd, case, emulate=True, stream_fp32=True)
    print(f"geometry: {latent}x{latent} latent, {frames} frames, CFG batch 2, full width; mean|ref| {ref.abs().mean().item():.4f} max|ref| {ref.abs().max().item():.3f}")
    for name, o in (("bf16 everywhere (PyTorch CPU bf16: the gate's floor)", b16), ("bf16 matmuls + fp32 residual stream (upper bound)", f32s)):
        d = (o - ref).abs()
        print(f"  {name:58s} max|d| {d.max().item():.3e}  mean|d| {d.mean().item():.3e}  max rel to absmax {d.max().item() / ref.abs().max().item():.3e}")


if __name__ == "__main__":
    main()
