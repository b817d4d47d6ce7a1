#!/usr/bin/env python3
"""Per call-site profile of one 512x512x24 denoise step: wraps the mmgt_amd.hip entry points with HIP events and prints
time by (op, shape), so the shapes that matter are known before a kernel is tuned.   python tools/profile_step.py [steps]"""
import collections
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from mmgt_amd import hip  # noqa: E402

records = []


def key_of(name, args, kw):
    if name == "gemm":
        a, w = args[0], args[1]
        epi = ("geglu" if kw.get("act") == hip.ACT_GEGLU else "") + ("+res" if kw.get("residual") is not None else "") + \
              ("+b2" if kw.get("bias2") is not None else "") + ("+rs" if kw.get("row_scale") is not None else "")
        return f"gemm M={a.shape[0]} N={w.shape[0]} K={a.shape[1]} {epi}", 2 * a.shape[0] * w.shape[0] * a.shape[1]
    if name == "conv1x1_cat":
        x0, x1, w = args[0], args[1], args[2]
        res = kw.get("residual") is not None or (len(args) > 4 and args[4] is not None)
        return (f"conv1x1_cat M={x0.shape[0]} N={w.shape[0]} K={x0.shape[1]}+{x1.shape[1]} (two sources, one launch){' +res' if res else ''}",
                2 * x0.shape[0] * w.shape[0] * (x0.shape[1] + x1.shape[1]))
    if name == "conv_taps_gather":
        return f"conv_taps_gather nb={args[2]} h={args[3]} (conv_out: sum of the neighbours' products)", 0
    if name == "gemm_bf16_f32":
        a, w = args[0], args[1]
        return f"gemm_bf16_f32 M={a.shape[0]} N={w.shape[0]} K={a.shape[1]} (bf16 pieces -> fp32)", 2 * a.shape[0] * w.shape[0] * a.shape[1]
    if name == "gemm_post":
        a, w = args[0], args[1]
        return f"gemm_post M={a.shape[0]} N={w.shape[0]} K={a.shape[1]} +rs+post+res", 2 * a.shape[0] * w.shape[0] * a.shape[1]
    if name == "gemm_batched_wx":
        w, x = args[0], args[1]
        return f"gemm_wx B={x.shape[0]} R={w.shape[0]} ntok={x.shape[1]} K={x.shape[2]}", 2 * x.shape[0] * w.shape[0] * x.shape[1] * x.shape[2]
    if name == "gemm_batched":
        a, w = args[0], args[1]
        return f"gemm_b B={a.shape[0]} M={a.shape[1]} N={w.shape[1]} K={a.shape[2]}", 2 * a.shape[0] * a.shape[1] * w.shape[1] * a.shape[2]
    if name == "conv3x3":
        x, wp = args[0], args[1]
        st, up = kw.get("stride", 1), kw.get("upsample", False)
        if up is not True and up == 2:      # the four-phase image [4][Cout][2][2][Cin]; TF/s stays algorithmic (36 multiply-adds per stored pixel)
            cout, cin = wp.shape[1], wp.shape[4]
        else:
            cout = wp.shape[0]
            cin = wp.numel() // (9 * cout)
        oh = x.shape[1] * (2 if up else 1) // st
        return (f"conv nb={x.shape[0]} h={x.shape[1]} cin={cin} cout={cout} s={st} up={int(up)}" + (" (four 2x2 phase convs)" if int(up) == 2 else ""),
                2 * x.shape[0] * oh * oh * cout * 9 * cin)
    if name == "attention":
        # the second key segment (reference bank) is read by batches >= seg2_first_batch only (the conditional CFG half)
        nb2 = kw["batch"] - kw.get("seg2_first_batch", 0) if kw.get("nk2", 0) else 0
        fl = 4 * kw["heads"] * kw["hd"] * kw["nq"] * (kw["batch"] * kw["nk"] + nb2 * kw.get("nk2", 0))
        return (f"attn b={kw['batch']} h={kw['heads']} hd={kw['hd']} nq={kw['nq']} nk={kw['nk']} nk2={kw.get('nk2', 0)} "
                f"vt={int(kw.get('v_transposed', False))}", fl)
    if name == "groupnorm":
        x = args[0]
        c = x.shape[2] + (kw["x1"].shape[2] if kw.get("x1") is not None else 0)
        return f"groupnorm nb={x.shape[0]} hw={x.shape[1]} c={c} silu={int(kw.get('silu', False))}", 0
    if name == "layernorm":
        x = args[0]
        return f"layernorm rows={x.shape[0]} c={x.shape[1]} pe={int(kw.get('pe') is not None)}", 0
    if name == "rowgemm320":
        x, N = args[0], args[2]
        n1 = kw.get("n1")
        pro = "LN" if kw.get("ln_gamma") is not None else "GN" if kw.get("pre_scale") is not None else "-"
        return (f"rowgemm320 M={x.shape[0]} N={N} ({pro} prologue{', V^T' if n1 not in (None, N) else ''}{', +res' if kw.get('residual') is not None else ''})",
                2 * x.shape[0] * N * 320)
    if name == "gn_silu_conv3x3_unet":
        x, cout = args[0], args[4]
        x1 = kw.get("x1")
        cin = x.shape[3] + (0 if x1 is None else x1.shape[3])
        res = kw.get("residual") is not None or (len(args) > 8 and args[8] is not None)
        return (f"gn_silu_conv3x3_unet nb={x.shape[0]} h={x.shape[1]} cin={cin} cout={cout} (GroupNorm apply + SiLU + conv{' + res' if res else ''}, one launch)",
                2 * x.shape[0] * x.shape[1] * x.shape[2] * cout * 9 * cin)
    if name == "groupnorm_affine":
        x = args[0]
        return f"groupnorm_affine nb={x.shape[0]} hw={x.shape[1]} c={x.shape[2]} (statistics only)", 0
    if name == "gn_silu_conv3x3_tables":
        x, cout = args[0], args[4]
        res = (args[6] if len(args) > 6 else kw.get("residual")) is not None
        return (f"gn_silu_conv3x3 nb={x.shape[0]} h={x.shape[1]} cin={x.shape[3]} cout={cout} (GroupNorm apply + SiLU + conv{' + res' if res else ''}, one launch)",
                2 * x.shape[0] * x.shape[1] * x.shape[2] * cout * 9 * x.shape[3])
    if name == "temporal_leg320":
        x, B, F, n = args[0], args[5], args[6], args[7]
        return (f"temporal_leg320 B={B} F={F} n={n} (LN + pe -> q|k|v -> attention over the frames -> to_out + res)",
                2 * x.shape[0] * 320 * 4 * 320 + 4 * x.shape[0] * 320 * F)
    if name == "ff_fused_po":
        x = args[0]
        inner = args[6]
        return (f"ff_fused_po M={x.shape[0]} C={x.shape[1]} inner={inner} (LN + ff1 + GEGLU + ff2 + res + proj_out + res)",
                2 * x.shape[0] * (3 * inner + x.shape[1]) * x.shape[1])
    if name == "ff_fused":
        x = args[0]
        inner = args[6] if len(args) > 6 else kw["inner"]
        return f"ff_fused M={x.shape[0]} C={x.shape[1]} inner={inner} (LN + ff1 + GEGLU + ff2 + res)", 2 * x.shape[0] * 3 * inner * x.shape[1]
    t = next((a for a in args if torch.is_tensor(a)), None)
    return f"{name} {tuple(t.shape) if t is not None else ''}", 0


def wrap(name):
    fn = getattr(hip, name)

    def inner(*args, **kw):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        r = fn(*args, **kw)
        e1.record()
        records.append((key_of(name, args, kw), e0, e1))
        return r
    setattr(hip, name, inner)


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    if len(sys.argv) > 2:                                       # window length (the reference ships context_frames = 12)
        bench.FRAMES = int(sys.argv[2])
        bench.build_inputs.__defaults__ = (bench.FRAMES,) + bench.build_inputs.__defaults__[1:]
    for n in ["gemm", "gemm_post", "gemm_batched_wx", "gemm_batched", "ff_fused", "ff_fused_po", "temporal_leg320", "rowgemm320", "groupnorm_affine", "gn_silu_conv3x3_unet", "conv1x1_cat", "conv_taps_gather", "conv3x3", "groupnorm", "layernorm", "attention", "softmax_rows",
              "ncfhw_to_nhwc", "nhwc_to_ncfhw", "timestep_features", "silu", "cfg_ddim_step", "accumulate_window"]:
        wrap(n)
    sys.argv = ["bench.py", "--steps", str(steps), "--warmup", "1", "--no-cpu-baseline", "--no-extras", "--no-calib"]
    import io
    import contextlib
    # run bench.main() (its own warmup is recorded too: drop those records afterwards)
    buf = io.StringIO()
    marks = []                                                  # len(records) at the start of every denoise step
    from mmgt_amd.unet3d import UNet3DConditionModel
    orig_dw = UNet3DConditionModel.denoise_window

    def marked(self, *a, **k):
        marks.append(len(records))
        return orig_dw(self, *a, **k)
    UNet3DConditionModel.denoise_window = marked
    with contextlib.redirect_stdout(buf):
        bench.main()
    torch.cuda.synchronize()
    UNet3DConditionModel.denoise_window = orig_dw
    # the records in front of marks[0] are once-per-clip set-up (set_banks: 32 projections, ...); marks[1] is the first timed step
    assert len(marks) >= steps + 1, (len(marks), steps)
    recs = records[marks[-steps]:]                              # the timed steps only (warm-up and set-up dropped)
    agg = collections.OrderedDict()
    for (k, fl), e0, e1 in recs:
        t = e0.elapsed_time(e1) * 1e3
        a = agg.setdefault(k, [0, 0.0, 0])
        a[0] += 1
        a[1] += t
        a[2] += fl
    tot = sum(a[1] for a in agg.values())
    print(f"bench: {buf.getvalue().strip()[:200]}")
    print(f"total inside calls: {tot / steps / 1e3:.2f} ms/step, {len(recs) // steps} calls/step")
    fam = collections.Counter()
    for k, (n, t, fl) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        fam[k.split()[0]] += t
        tf = f"{fl / t / 1e6:7.0f} TF/s" if fl else " " * 12
        print(f"{t / steps / 1e3:8.3f} ms {100 * t / tot:5.1f}%  x{n // steps:4d}  {t / n:8.1f} us  {tf}  {k}")
    print({k: round(v / steps / 1e3, 2) for k, v in fam.most_common()})


if __name__ == "__main__":
    main()
