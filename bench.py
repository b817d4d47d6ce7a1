#!/usr/bin/env python3
"""bench.py — UNet3D denoise-steps/s of the MMGT Stage-2 path on MI355X (BASELINE.json metric, config 2).

One "step" = one DDIM iteration of Pose2VideoPipeline's loop over one 24-frame 512x512 clip (latents (1,4,24,64,64)):
CFG-batched UNet3D forward on (2,4,24,64,64) with pose features, motion masks, 32 audio tokens per frame and the 16
reference-attention banks, then window accumulate + CFG combine + DDIM update.  bf16 storage, fp32 accumulate.
All inputs are resident in HBM before the timed region.  N GPUs = N independent clips (clip-parallel, no collective in
the loop), so `value` is the aggregate over ranks and scaling is "weak".

    python bench.py [--gpus N] [--steps K] [--warmup W] [--no-cpu-baseline] [--full-cpu-baseline] [--no-extras]

With --gpus N > 1 and no RANK in the environment, bench.py starts `python -m torch.distributed.run --nproc-per-node N` on
itself as a CHILD process (before anything touches the GPU) and exits with its return code; under torch.distributed.run it
reads RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* as usual.

Besides the contract's line (metric / value / roofline / cpu_baseline) rank 0 at N = 1 reports, timed OUTSIDE the step loop
("exclude prologue + VAE, report them separately"): `vae_decode` (ms per 512x512 frame and its MFMA roofline
fraction at 2.51 TFLOP per frame), `prologue_ms` (CLIP ViT-L/14 + VAE encode + ReferenceNet + PoseGuider + bank projection,
once per clip), `max_abs_delta_vs_cpu` (bf16 HIP forward against the CPU fp32 oracle on the cpu_baseline's sample) and
`roofline_kernels` (live HIP-event timings of the step's heaviest kernels at their in-step shapes).
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

ALGO_TFLOP_PER_STEP = 62.0      # minimal algorithmic FLOPs of one CFG denoise step at 512x512x24
EXEC_TFLOP_PER_STEP = 71.6      # as executed by the reference's op graph (quoted alongside, never the numerator)
VAE_TFLOP_PER_FRAME = 2.51
PEAK_BF16_TFLOPS = 2500.0       # MI355X dense bf16 MFMA (MI355X_MICROARCH.md)
FRAMES, LATENT = 24, 64
FULL_CPU_STEP = "profiles/r3/bench_full_cpu_baseline_r3.json"   # one measured full CFG step of the oracle on the GPU box's host cores


def executed_tflop(hip):
    """TFLOP one step of THIS build executes: SURVEY 8d's 62.0 minus the exact skips of the CFG pair that are switched on (DESIGN section 4;
    each is work the reference performs on provably zero or identical operands).  Derived from the shapes, per switch of the mmgt_tune table."""
    f, n0 = FRAMES, LATENT * LATENT                       # frames per CFG row, tokens per frame at level 0
    t = ALGO_TFLOP_PER_STEP * 1e12
    if hip.tune_get("zero_audio_skip") and hip.tune_get("oz3"):
        # the unconditional row's audio cross-attention (zero keys / values -> output exactly 0): per MM-HAA module its q projection, the
        # 32-key attention of the three branches and the 3 x inner columns of the merged out-projection (inner, tokens per frame) per module
        for inner, n in ((320, n0), (320, n0), (320, n0 // 4), (640, n0 // 4), (640, n0 // 16), (1280, n0 // 16)):
            rows = f * n
            t -= 2.0 * rows * inner * 3 * inner + 4.0 * rows * 3 * inner * 32 + 2.0 * rows * inner * 3 * inner
    if hip.tune_get("shared_rows"):
        # conv_in and the first ResnetBlock3D on the f frames once (both rows enter with the same latents / pose / timestep)
        t -= 2.0 * f * n0 * 320 * 36 + 2 * (2.0 * f * n0 * 320 * 9 * 320)
        if hip.tune_get("twin_attention"):
            # the first reference reader: GroupNorm / proj_in / q | k | v once, ONE attention pass over the own keys for both rows
            t -= 4 * (2.0 * f * n0 * 320 * 320) + 4.0 * 8 * 40 * n0 * n0 * f
    if hip.tune_get("up2"):
        # the three convs behind a nearest 2x upsampling as four 2 x 2 convs on the stored image: 16 instead of 36 multiply-adds per stored pixel
        # and channel pair (weights of the taps that fall on one stored pixel summed on the host: the same function, packing.pack_conv3x3_up2)
        for c, n in ((1280, n0 // 64), (1280, n0 // 16), (640, n0 // 4)):
            t -= 2.0 * (2 * f) * n * c * c * (36 - 16)
    return t / 1e12


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--full-cpu-baseline", action="store_true",
                    help="time one FULL 24-frame CFG forward of the oracle on the host (minutes); the default on hosts with >= 64 cores")
    ap.add_argument("--sample-cpu-baseline", action="store_true",
                    help="time the 2-of-24-frame sample of the oracle only (the default on hosts with < 64 cores)")
    ap.add_argument("--no-extras", action="store_true", help="skip the VAE / prologue / per-kernel legs")
    ap.add_argument("--no-calib", action="store_true", help="skip box_calib (profiled runs: its kernels are not part of the step)")
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f32"])
    ap.add_argument("--master-port", type=int, default=0)
    return ap.parse_args()


def self_launch(a):
    """--gpus N > 1 without a launcher: re-run this file under torch.distributed.run as a child process.  Nothing in this
    process has touched the GPU yet (torch is not even imported), and the child is a plain subprocess, not an exec."""
    port = a.master_port or (29500 + os.getpid() % 2000)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.call(cmd, env=env)


def build_inputs(dev, frames=FRAMES, latent=LATENT, tag="bench"):
    import torch  # noqa: F401
    from mmgt_amd.synthetic import bank_spatial, hash_uniform, synth_masks
    lips = synth_masks(tag + ".lips", frames, latent)
    face = synth_masks(tag + ".face", frames, latent)
    full = [1 + l for l in lips]
    return dict(
        latents=hash_uniform(tag + ".latents", (1, 4, frames, latent, latent), 1.7).to(dev),
        clip=hash_uniform(tag + ".clip", (1, 768), 1.0).to(dev),
        audio=hash_uniform(tag + ".audio", (1, frames, 32, 768), 1.7).to(dev),
        pose=hash_uniform(tag + ".pose", (1, 320, frames, latent, latent), 0.5).to(dev),
        full=[m.to(dev) for m in full], face=[m.to(dev) for m in face], lips=[m.to(dev) for m in lips],
        banks={k: hash_uniform(tag + ".bank." + k, (2, n, c), 1.0).to(dev)
               for k, (n, c) in bank_spatial((320, 640, 1280, 1280), latent).items()},
        motion_scale=[1.0, 1.0, 2.0])


def operator_args(inp, dev):
    """The CFG-batched operator call for `inp` (what Pose2VideoPipeline.denoise feeds the UNet for one window)."""
    import torch
    cat2 = lambda L: [torch.cat([m] * 2).to(dev) for m in L]
    sample = inp["latents"].repeat(2, 1, 1, 1, 1).to(dev)
    ehs = torch.cat([torch.zeros(1, 1, 768), inp["clip"].reshape(1, 1, 768).cpu()]).to(dev)
    audio = torch.cat([torch.zeros_like(inp["audio"]), inp["audio"]]).to(dev)
    pose = inp["pose"].repeat(2, 1, 1, 1, 1).to(dev)
    return sample, ehs, audio, pose, cat2(inp["full"]), cat2(inp["face"]), cat2(inp["lips"])


def cpu_baseline(sd_cpu, frames_sample=2):
    """Oracle (CPU fp32 restatement of the reference) timed on this box's host cores on a bounded sample: one CFG
    denoise-step forward at 512x512 with `frames_sample` of the 24 frames; cost is linear in frames (spatial ops are
    per frame; temporal attention is <0.2% of the FLOPs), so steps/s = 1 / (t * 24 / frames_sample).  With
    frames_sample = 24 (--full-cpu-baseline) it is one measured step.  Returns (record, inputs, oracle output)."""
    import torch
    from oracle import unet3d_ref as R
    cores = max(1, (os.cpu_count() or 2) // 2)
    torch.set_num_threads(cores)
    inp = build_inputs("cpu", frames=frames_sample, tag="bench.cpu")
    sample, ehs, audio, pose, full, face, lips = operator_args(inp, "cpu")
    t0 = time.time()
    with torch.no_grad():
        out = R.unet3d_forward(sd_cpu, R.UNet3DConfig(), sample, torch.tensor(499), ehs, audio, pose, full, face, lips,
                               inp["motion_scale"], inp["banks"], weighted=True)
    dt = time.time() - t0
    assert torch.isfinite(out).all()
    what = (f"one CFG denoise-step forward of the oracle (PyTorch CPU fp32) at 512x512 on {frames_sample} of 24 frames: "
            f"{dt:.1f} s" + (f", scaled x{FRAMES // frames_sample} to 24 frames" if frames_sample != FRAMES else " (measured, not scaled)"))
    rec = {"value": 1.0 / (dt * FRAMES / frames_sample), "unit": "steps/s", "cores": cores, "kind": "port", "sample": what,
           "cpu": _cpu_name()}
    full = os.path.join(ROOT, FULL_CPU_STEP)
    if frames_sample != FRAMES and os.path.exists(full):
        # 128 cores are not filled by 2 frames, so the cost is NOT linear in frames: the stated baseline is the MEASURED full
        # step (same oracle, same host class, committed); today's bounded sample and its linear scaling are reported beside it
        m = json.load(open(full))["cpu_baseline"]
        rec = {"value": m["value"], "unit": "steps/s", "cores": m["cores"], "kind": f"port, measured {FULL_CPU_STEP}",
               "sample": m["sample"], "cpu": m["cpu"],
               "todays_sample": {"value_scaled_linearly": 1.0 / (dt * FRAMES / frames_sample), "cores": cores, "cpu": _cpu_name(), "sample": what}}
    return rec, inp, out


def pmc_traffic():
    """HBM-side bytes per denoise step from the committed rocprofv3 PMC passes (profiles/r*/pmc_traffic*.json: FETCH_SIZE and
    WRITE_SIZE collected in separate runs of this same command, gfx950 FETCH_SIZE x2 correction applied)."""
    import glob
    cur = os.path.join(ROOT, "profiles", "CURRENT_TRAFFIC")      # names the passes that belong to the current build
    files = []
    if os.path.exists(cur):
        f = os.path.join(ROOT, open(cur).read().strip())
        files = [f] if os.path.exists(f) else []
    if not files:
        files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", "pmc_traffic*.json")), key=os.path.getmtime)
    if not files:
        return None, None
    d = json.load(open(files[-1]))
    return d["total_GB_per_step"] * 1e9, os.path.relpath(files[-1], ROOT)


def pmc_mfma_busy():
    """Matrix-pipe busy fraction of the whole step from the committed PMC pass of this same command (profiles/r*/pmc_step_mfma*.json:
    rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE, tools/profile_bench.sh)."""
    import glob
    cur = os.path.join(ROOT, "profiles", "CURRENT_TRAFFIC")      # the matrix-pipe pass of the same profile run sits beside the traffic passes
    if os.path.exists(cur):
        sib = os.path.join(ROOT, open(cur).read().split()[0].replace("pmc_traffic_", "pmc_step_mfma_"))
        if os.path.exists(sib):
            return json.load(open(sib))["mfma_busy"], os.path.relpath(sib, ROOT)
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", "pmc_step_mfma*.json")))   # (by name: file times do not survive a checkout)
    if not files:
        return None, None
    return json.load(open(files[-1]))["mfma_busy"], os.path.relpath(files[-1], ROOT)


def _cpu_name():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def _time_ms(fn, reps=20, warm=3):
    """Average device time of fn() in ms, HIP events on the stream the kernels are launched on (torch's current stream)."""
    import torch
    for _ in range(warm):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def box_calib(dev, dtype):
    """What THIS box does on two fixed loads, so that lines from different boxes (+-4 % on identical binaries) can be read against each other:
    the in-kernel clock / rate of a bare 16x16x32 bf16 MFMA loop after 2 s of back-to-back launches (csrc/calib.hip) and the time of one
    fixed gemm16 launch, 8192^3 bf16 on random operands."""
    import torch
    from mmgt_amd import hip
    from mmgt_amd.synthetic import hash_uniform
    mhz, tf = hip.box_calib(2.0)
    rec = {"mfma_loop_mhz": mhz, "mfma_loop_tflops": tf, "what": "bare v_mfma_f32_16x16x32_bf16 loop, random operands, one wave per SIMD, "
           "in-kernel clock (s_memtime / s_memrealtime) after 2 s of back-to-back launches"}
    if dtype == torch.bfloat16:
        a = hash_uniform("calib.a", (8192, 8192), 1.0, dev).to(dtype)
        w = hash_uniform("calib.w", (8192, 8192), 1.0, dev).to(dtype)
        o = torch.empty((8192, 8192), device=dev, dtype=dtype)
        ms = _time_ms(lambda: hip.gemm(a, w, out=o), reps=20, warm=5)
        rec["gemm16_8192_us"] = ms * 1e3
        rec["gemm16_8192_tflops"] = 2.0 * 8192 ** 3 / ms / 1e9
    return rec


def kernel_rooflines(unet, dev):
    """Live HIP-event timings of the step's heaviest kernels at their in-step shapes (level 0: 48 frames x 4096 tokens x 320
    channels), each against the roofline that bounds it.  Launch counts per step are from profiles/ (opshapes)."""
    import torch
    from mmgt_amd import hip
    from mmgt_amd.synthetic import hash_uniform
    dt = unet.dtype
    out = []
    M, C = 48 * 4096, 320
    x = hash_uniform("k.x", (M, C), 1.0, dev).to(dt)
    t = "down_blocks.0.attentions.0.transformer_blocks.0"
    # GEGLU ff1 (N = 2560, K = 320) + ff2 (K = 1280)
    w1, b1 = unet.w[t + ".ff.ff1.w"], unet.w[t + ".ff.ff1.bias"]
    ms = _time_ms(lambda: hip.gemm(x, w1, b1, act=hip.ACT_GEGLU))
    fl = 2.0 * M * 2560 * 320
    out.append({"kernel": "gemm GEGLU ff1 M=196608 N=2560 K=320", "bound": "mfma", "ms": ms, "achieved": fl / ms / 1e9,
                "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s", "frac": fl / ms / 1e9 / PEAK_BF16_TFLOPS})
    # the whole FeedForward of a level-0 block as the step runs it: LayerNorm -> ff1 -> GEGLU -> ff2 -> + residual, one launch (csrc/ffn.hip)
    if (t + ".ff.ffimg") in unet.w:
        img, b2 = unet.w[t + ".ff.ffimg"], unet.w[t + ".ff.ff2.bias"]
        g3, b3 = unet.w[t + ".norm3.g"], unet.w[t + ".norm3.b"]
        ms = _time_ms(lambda: hip.ff_fused(x, g3, b3, img, b2, x, 1280))
        fl = 2.0 * M * 3 * 1280 * 320
        out.append({"kernel": "ff_fused M=196608 C=320 inner=1280 (LayerNorm + ff1 + GEGLU + ff2 + residual)", "bound": "mfma", "ms": ms,
                    "achieved": fl / ms / 1e9, "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s", "frac": fl / ms / 1e9 / PEAK_BF16_TFLOPS})
    # LayerNorm -> q | k (row-major) + V^T of a level-0 self-attention, one launch (csrc/rowgemm.hip): x once in, 3 x 126 MB out
    if (t + ".attn1.qkv_img") in unet.w:
        img, g1, b1n = unet.w[t + ".attn1.qkv_img"], unet.w[t + ".norm1.g"], unet.w[t + ".norm1.b"]
        vt_o = torch.empty((48, C, 4096), device=dev, dtype=dt)
        qk_o = torch.empty((M, 2 * C), device=dev, dtype=dt)
        ms = _time_ms(lambda: hip.rowgemm320(x, img, 3 * C, ln_gamma=g1, ln_beta=b1n, n1=2 * C, n_tok=4096, out=qk_o, out_t=vt_o))
        by = 4.0 * M * C * 2
        out.append({"kernel": "rowgemm320 M=196608 N=960 (LayerNorm + q|k + V^T)", "bound": "hbm", "ms": ms, "achieved": by / ms / 1e6,
                    "peak": 8000.0, "unit": "GB/s", "frac": by / ms / 1e6 / 8000.0})
        del vt_o, qk_o
    # spatial attention with bank (uncond half: 4096 keys, cond half: 8192 keys), hd 40
    n, heads, hd = 4096, 8, 40
    qk = hash_uniform("k.qk", (48 * n, 2 * C), 1.0, dev).to(dt)
    vt = hash_uniform("k.vt", (48, C, n), 1.0, dev).to(dt)
    kb = hash_uniform("k.kb", (2, n, C), 1.0, dev).to(dt)
    vbt = hash_uniform("k.vbt", (2, C, n), 1.0, dev).to(dt)
    o = torch.empty((48 * n, C), device=dev, dtype=dt)

    def attn():
        hip.attention(qk, qk[:, C:], vt, o, batch=48, heads=heads, hd=hd, nq=n, nk=n, scale=hd ** -0.5,
                      q_str=(n * 2 * C, 0, 2 * C), k_str=(n * 2 * C, 0, 2 * C), v_str=(C * n, 0, n), o_str=(n * C, 0, C),
                      v_transposed=True, k2=kb, v2=vbt, k2_str=(kb.stride(0), kb.stride(1)),
                      v2_str=(vbt.stride(0), vbt.stride(1)), k2_bdiv=24, nk2=n, seg2_first_batch=24)
    ms = _time_ms(attn)
    fl = 4.0 * heads * hd * n * (24 * n + 24 * 2 * n)          # the uncond half never reads the bank
    out.append({"kernel": "attention hd=40 nq=4096 nk=4096 (+4096 bank keys for the cond half)", "bound": "mfma", "ms": ms,
                "achieved": fl / ms / 1e9, "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s", "frac": fl / ms / 1e9 / PEAK_BF16_TFLOPS})
    # conv3x3 320 -> 320 at 64x64
    xi = hash_uniform("k.xi", (48, 64, 64, C), 1.0, dev).to(dt)
    wc, bc = unet.w["down_blocks.0.resnets.0.conv2.w"], unet.w["down_blocks.0.resnets.0.conv2.bias"]
    ms = _time_ms(lambda: hip.conv3x3(xi, wc, bc))
    fl = 2.0 * M * C * 9 * C
    out.append({"kernel": "conv3x3 48x64x64 320->320", "bound": "mfma", "ms": ms, "achieved": fl / ms / 1e9,
                "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s", "frac": fl / ms / 1e9 / PEAK_BF16_TFLOPS})
    # ... the same leg as the step runs it since round 6: GroupNorm apply + SiLU + conv3x3 + residual as ONE launch behind the statistics pass (csrc/rconv.hip)
    g, b = unet.w["down_blocks.0.resnets.0.norm1.g"], unet.w["down_blocks.0.resnets.0.norm1.b"]
    x3 = xi.view(48, 4096, C)
    if "down_blocks.0.resnets.0.conv2.rimg" in unet.w:
        rimg = unet.w["down_blocks.0.resnets.0.conv2.rimg"]
        sc_, sh_ = hip.groupnorm_affine(x3, g, b, 32, 1e-5)
        oc = torch.empty_like(xi)
        ms = _time_ms(lambda: hip.gn_silu_conv3x3_unet(xi, sc_, sh_, rimg, C, bc, residual=xi, out=oc))
        out.append({"kernel": "gn_silu_conv3x3_unet 48x64x64 320->320 (GroupNorm apply + SiLU + conv + residual, one launch)", "bound": "mfma", "ms": ms,
                    "achieved": fl / ms / 1e9, "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s", "frac": fl / ms / 1e9 / PEAK_BF16_TFLOPS})
        ms = _time_ms(lambda: hip.groupnorm_affine(x3, g, b, 32, 1e-5))
        by = 1.0 * M * C * 2
        out.append({"kernel": "groupnorm statistics pass 48x4096x320 (tables for the fused launch)", "bound": "hbm", "ms": ms, "achieved": by / ms / 1e6,
                    "peak": 8000.0, "unit": "GB/s", "frac": by / ms / 1e6 / 8000.0})
        del oc
    # GroupNorm + SiLU at L0 (HBM: 2 reads + 1 write of the tensor): the unfused pass (fp32-I/O mode, the 8 x 8 level)
    ms = _time_ms(lambda: hip.groupnorm(x3, g, b, 32, 1e-5, silu=True))
    by = 3.0 * M * C * 2
    out.append({"kernel": "groupnorm+silu 48x4096x320", "bound": "hbm", "ms": ms, "achieved": by / ms / 1e6, "peak": 8000.0,
                "unit": "GB/s", "frac": by / ms / 1e6 / 8000.0})
    del x3, xi
    # GroupNorm of a level-1 transformer block (48 x 1024 x 640): the slab of one image x two groups in registers, one read + one write (csrc/norm.hip gn_slab_kernel)
    x1 = hash_uniform("k.x1", (48, 1024, 640), 1.0, dev).to(dt)
    g1_, b1_ = torch.ones(640, device=dev), torch.zeros(640, device=dev)
    o1 = torch.empty_like(x1)
    ms = _time_ms(lambda: hip.groupnorm(x1, g1_, b1_, 32, 1e-6, out=o1))
    by = 2.0 * x1.numel() * 2
    out.append({"kernel": "groupnorm 48x1024x640 (slab in registers: one read, one write)", "bound": "hbm", "ms": ms, "achieved": by / ms / 1e6, "peak": 8000.0,
                "unit": "GB/s", "frac": by / ms / 1e6 / 8000.0})
    del x1, o1
    # spatial attention with bank at the 32 x 32 level, hd 80 (csrc/attn80.hip)
    n1, c1 = 1024, 640
    qk1 = hash_uniform("k.qk1", (48 * n1, 2 * c1), 1.0, dev).to(dt)
    vt1 = hash_uniform("k.vt1", (48, c1, n1), 1.0, dev).to(dt)
    kb1 = hash_uniform("k.kb1", (2, n1, c1), 1.0, dev).to(dt)
    vbt1 = hash_uniform("k.vbt1", (2, c1, n1), 1.0, dev).to(dt)
    oa = torch.empty((48 * n1, c1), device=dev, dtype=dt)
    ms = _time_ms(lambda: hip.attention(qk1, qk1[:, c1:], vt1, oa, batch=48, heads=8, hd=80, nq=n1, nk=n1, scale=80 ** -0.5, q_str=(n1 * 2 * c1, 0, 2 * c1),
                                        k_str=(n1 * 2 * c1, 0, 2 * c1), v_str=(c1 * n1, 0, n1), o_str=(n1 * c1, 0, c1), v_transposed=True, k2=kb1, v2=vbt1,
                                        k2_str=(kb1.stride(0), kb1.stride(1)), v2_str=(vbt1.stride(0), vbt1.stride(1)), k2_bdiv=24, nk2=n1, seg2_first_batch=24))
    fl = 4.0 * 8 * 80 * n1 * (24 * n1 + 24 * 2 * n1)
    out.append({"kernel": "attention hd=80 nq=1024 nk=1024 (+1024 bank keys for the cond half)", "bound": "mfma", "ms": ms, "achieved": fl / ms / 1e9,
                "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s", "frac": fl / ms / 1e9 / PEAK_BF16_TFLOPS})
    return out


def extras(pipe, unet, dev, dtype):
    """VAE decode, prologue -- the once-per-clip legs SURVEY 8d asks to report next to the step rate."""
    import torch
    from mmgt_amd.synthetic import build_synthetic_pipeline, hash_uniform
    res = {}
    full = build_synthetic_pipeline(dev, dtype, with_prologue=True)
    vae, clip, refnet, pg = full.vae, full.image_encoder, full.reference_unet, full.pose_guider
    del full
    # ---- VAE decode: 8 frames per call as decode_video batches them
    z = hash_uniform("bench.z", (1, 4, 8, LATENT, LATENT), 1.0).to(dev)
    ms = _time_ms(lambda: vae.decode_video(z), reps=2, warm=1) / 8
    tf = VAE_TFLOP_PER_FRAME / (ms / 1e3)
    res["vae_decode"] = {"ms_per_frame": ms, "ms_per_24_frame_clip": ms * 24, "achieved": tf, "peak": PEAK_BF16_TFLOPS,
                         "unit": "TFLOP/s", "frac": tf / PEAK_BF16_TFLOPS, "tflop_per_frame": VAE_TFLOP_PER_FRAME}
    # the decoder's HBM-bound kernel (GroupNorm + SiLU at 512 x 512 x 128: two reads + one write of a 537 MB tensor) and the split-operand
    # logits GEMM of its mid-block attention (q . k as three bf16 products over one 3C-long reduction, fp32 out)
    from mmgt_amd import hip
    x = hash_uniform("bench.vae_gn", (8, 512 * 512, 128), 1.0).to(dev).to(dtype)
    g, b = torch.ones(128, device=dev), torch.zeros(128, device=dev)
    o = torch.empty_like(x)
    gms = _time_ms(lambda: hip.groupnorm(x, g, b, 32, 1e-6, silu=True, out=o), reps=5, warm=2)
    by = 3.0 * x.numel() * 2
    kern = [{"kernel": "groupnorm+silu 8x262144x128 (VAE up_blocks.3)", "bound": "hbm", "ms": gms, "achieved": by / gms / 1e6, "peak": 8000.0,
             "unit": "GB/s", "frac": by / gms / 1e6 / 8000.0}]
    if dtype == torch.bfloat16:
        # the fused GroupNorm-apply + SiLU + conv3x3 128 -> 128 launch of the same level (csrc/gnconv.hip), against the MFMA roofline
        from mmgt_amd.packing import pack_gnconv
        xg = x.view(8, 512, 512, 128)
        sc_, sh_ = hip.groupnorm_affine(x, g, b, 32, 1e-6)
        wg = pack_gnconv(hash_uniform("bench.vae_gw", (128, 128, 3, 3), 0.03).to(dev))
        og = o.view(8, 512, 512, 128)
        cms = _time_ms(lambda: hip.gn_silu_conv3x3_tables(xg, sc_, sh_, wg, 128, None, None, out=og), reps=5, warm=2)
        fl = 2.0 * x.shape[0] * x.shape[1] * 128 * 9 * 128
        kern.append({"kernel": "gn_silu_conv3x3 8x512x512 128->128 (GroupNorm apply + SiLU + conv, one launch)", "bound": "mfma", "ms": cms,
                     "achieved": fl / cms / 1e9, "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s", "frac": fl / cms / 1e9 / PEAK_BF16_TFLOPS})
        del xg, og, wg
    del x, o
    if dtype == torch.bfloat16:
        qp = hash_uniform("bench.vae_qp", (4096, 1536), 1.0).to(dev).to(dtype)
        kp = hash_uniform("bench.vae_kp", (4096, 1536), 1.0).to(dev).to(dtype)
        so = torch.empty((4096, 4096), device=dev, dtype=torch.float32)
        lms = _time_ms(lambda: hip.gemm_bf16_f32(qp, kp, out=so), reps=5, warm=2)
        fl = 2.0 * 4096 * 4096 * 1536
        kern.append({"kernel": "gemm_bf16_f32 4096x4096x1536 (VAE attention logits, hi/lo pieces)", "bound": "mfma", "ms": lms,
                     "achieved": fl / lms / 1e9, "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s", "frac": fl / lms / 1e9 / PEAK_BF16_TFLOPS})
        del qp, kp, so
    res["vae_decode"]["kernels"] = kern
    # ---- prologue: CLIP embed, VAE encode of the reference image, ReferenceNet (banks), PoseGuider, bank K/V projection
    ref = hash_uniform("bench.ref", (1, 3, 512, 512), 1.0).to(dev)
    pix = hash_uniform("bench.clip_px", (1, 3, 224, 224), 1.5).to(dev)
    pose = (hash_uniform("bench.pose_rgb", (1, 3, FRAMES, 512, 512), 0.5) + 0.5).to(dev)

    def prologue():
        emb = clip(pix.to(clip.dtype)).image_embeds.float().reshape(1, 1, -1)
        ehs = torch.cat([torch.zeros_like(emb), emb])
        lat = vae.encode_mean(ref) * 0.18215
        banks = refnet.write_banks(lat.float().repeat(2, 1, 1, 1), 0, ehs)
        unet.set_banks(banks)
        return pg(pose)
    res["prologue_ms"] = _time_ms(prologue, reps=2, warm=1)
    del vae, clip, refnet, pg
    torch.cuda.empty_cache()
    res.update(other_configs(pipe, unet, dev, dtype))
    return res


def other_configs(pipe, unet, dev, dtype):
    """The denoise loop at the other BASELINE geometries, timed like the headline step (HIP events, inputs resident):
      config5_ms_per_step   one DDIM step of the 96-frame long video: context 24, overlap 8 => 6 windows (pipeline_pose2vid_long.py:522-635)
      ctx12_ms_per_window   the reference's SHIPPED window length, context_frames = 12 (pipeline_pose2vid_long.py:360-362,
                            scripts/pose2vid.py:317-321), on the 24-frame clip (overlap 4)
      smga_ms_per_slice     Stage 1 of config 3: one 80-frame slice of the SMGA sampler (50 DDIM steps x 2 guidance passes)"""
    import torch
    from mmgt_amd.context import uniform
    from mmgt_amd.synthetic import synth_tensor
    res = {}
    sched = pipe.scheduler
    dup = lambda ms: [torch.cat([m] * 2) for m in ms]

    def loop_ms(frames, ctx, ov, tag, nsteps=2):
        inp = build_inputs(dev, frames=frames, tag=tag)
        unet.set_banks(inp["banks"])
        ehs = torch.cat([torch.zeros(1, 1, 768, device=dev), inp["clip"].reshape(1, 1, 768)])
        audio_pre = torch.cat([torch.zeros_like(inp["audio"]), inp["audio"]])
        full, face, lips = dup(inp["full"]), dup(inp["face"]), dup(inp["lips"])
        nwin = len(list(uniform(0, 25, frames, ctx, 1, ov)))
        fn = lambda: pipe.denoise(inp["latents"], [sched.timesteps[3 + i] for i in range(nsteps)], ehs, inp["pose"], audio_pre, full, face,
                                  lips, 3.5, inp["motion_scale"], context_frames=ctx, context_stride=1, context_overlap=ov,
                                  num_inference_steps=25)
        out = fn()
        assert torch.isfinite(out).all()
        return _time_ms(fn, reps=1, warm=0) / nsteps, nwin

    ms, nwin = loop_ms(96, 24, 8, "bench.c5")
    res["config5_ms_per_step"] = {"ms": ms, "windows_per_step": nwin, "ms_per_window": ms / nwin,
                                  "what": "512x512x96, context 24, overlap 8, one GPU (BASELINE configs[4] geometry)"}
    torch.cuda.empty_cache()
    ms, nwin = loop_ms(FRAMES, 12, 4, "bench.c12")
    res["ctx12_ms_per_window"] = {"ms": ms / nwin, "windows_per_step": nwin, "ms_per_step": ms,
                                  "what": "512x512x24 clip sampled with the reference's shipped context_frames=12, overlap 4"}
    # ---- SMGA (Stage 1): one 3.2-second slice
    from mmgt_amd.smga import SMGA, smga_spec
    sd = {k: (synth_tensor("smga." + k, shp) if not k.endswith("rotary.freqs") else torch.zeros(shp)) for k, shp in smga_spec().items()}
    a2p = SMGA(feature_type="wavlm", device=dev, dtype=dtype, state_dict=sd)
    from mmgt_amd.synthetic import hash_uniform
    cond, init = hash_uniform("bench.smga.cond", (1, 80, 1059), 1.0), hash_uniform("bench.smga.init", (1, 402), 0.8)
    gen = torch.Generator(device=dev).manual_seed(0)
    res["smga_ms_per_slice"] = _time_ms(lambda: a2p.render_sample(cond_frame=init, cond=cond[0], generator=gen), reps=2, warm=1)
    return res


def main():
    a = parse_args()
    if a.gpus > 1 and "RANK" not in os.environ:
        raise SystemExit(self_launch(a))

    import torch
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {a.gpus}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the product has no CPU path")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist = None
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=dev)

    from mmgt_amd.pipeline import Pose2VideoPipeline
    from mmgt_amd.scheduler import DDIMScheduler
    from mmgt_amd.synthetic import synth_state_dict
    from mmgt_amd.unet3d import UNet3DConditionModel
    from mmgt_amd.unet3d_spec import unet3d_spec

    dtype = torch.bfloat16 if a.dtype == "bf16" else torch.float32
    sd = synth_state_dict(unet3d_spec(), device=dev)          # random-init weights of the reference architecture
    unet = UNet3DConditionModel(device=dev, dtype=dtype)
    unet.load_state_dict(sd)
    unet.enable_gradient_checkpointing()                      # script semantics: motion_scale applied
    sd_cpu = None
    if rank == 0 and world == 1 and not a.no_cpu_baseline:
        sd_cpu = {k: v.cpu() for k, v in sd.items()}
    del sd
    torch.cuda.empty_cache()

    sched = DDIMScheduler()
    sched.set_timesteps(25)
    pipe = Pose2VideoPipeline(vae=None, image_encoder=None, reference_unet=None, denoising_unet=unet, pose_guider=None,
                              scheduler=sched)
    inp = build_inputs(dev, tag=f"bench.rank{rank}")
    unet.set_banks(inp["banks"])
    ehs = torch.cat([torch.zeros(1, 1, 768, device=dev), inp["clip"].reshape(1, 1, 768)])
    audio_pre = torch.cat([torch.zeros_like(inp["audio"]), inp["audio"]])
    dup = lambda ms: [torch.cat([m] * 2) for m in ms]
    full, face, lips = dup(inp["full"]), dup(inp["face"]), dup(inp["lips"])

    def run(nsteps, start):
        ts = [sched.timesteps[(start + i) % 25] for i in range(nsteps)]
        return pipe.denoise(inp["latents"], ts, ehs, inp["pose"], audio_pre, full, face, lips, 3.5, inp["motion_scale"],
                            context_frames=FRAMES, context_stride=1, context_overlap=4, num_inference_steps=25)

    def fence():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    if a.warmup:
        run(a.warmup, 0)
    fence()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record()
    out = run(a.steps, a.warmup)
    ev1.record()
    fence()
    elapsed = time.perf_counter() - t0
    dev_ms = ev0.elapsed_time(ev1)
    assert torch.isfinite(out).all(), "non-finite latents"
    if dist is not None:
        tt = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())

    if rank == 0:
        per_gpu = a.steps / elapsed
        ach = ALGO_TFLOP_PER_STEP * a.steps / (dev_ms / 1e3)
        traffic, traffic_src = pmc_traffic()
        from mmgt_amd import hip
        exe = executed_tflop(hip) if a.dtype == "bf16" else ALGO_TFLOP_PER_STEP
        ach_exe = exe * a.steps / (dev_ms / 1e3)
        busy, busy_src = pmc_mfma_busy()
        res = {
            "metric": "UNet3D denoise-steps/sec at 512x512x24 bf16", "value": world * per_gpu, "unit": "steps/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": 1e3 * elapsed / a.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": a.dtype, "data": "synthetic",
            "config": {"workload": "BASELINE configs[1]: 512x512x24 pose2vid denoise step (CFG batch 2, UNet3D + MM-HAA, "
                                   "16 reference banks, 1 window of 24 frames, guidance 3.5, DDIM v-pred), random-init "
                                   "weights", "frames": FRAMES, "latent": [LATENT, LATENT], "parallelism": f"clip-parallel x{world}",
                       "device_ms_per_step_rank0": dev_ms / a.steps},
            "roofline": {"bound": "mfma", "achieved": ach, "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
                         "frac": ach / PEAK_BF16_TFLOPS, "traffic": traffic,
                         "executed_tflop": exe, "frac_executed": ach_exe / PEAK_BF16_TFLOPS, "mfma_busy": busy,
                         "note": f"whole denoise step on one GPU: {ALGO_TFLOP_PER_STEP} TFLOP algorithmic per step "
                                 f"(SURVEY 8d; {EXEC_TFLOP_PER_STEP} as executed by the reference) / HIP-event time; executed_tflop = "
                                 f"what this build runs per step after the exact skips of the CFG pair that are switched on; "
                                 f"traffic = bytes per step from {traffic_src}; mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x "
                                 f"kernel cycles) over the step's kernels from {busy_src} -- both are committed rocprofv3 PMC passes of this "
                                 f"command at --steps 3 --warmup 1 (per-step = total / 4), not measurements of this {a.steps}-step run"},
        }
        if not a.no_calib:
            res["box_calib"] = box_calib(dev, dtype)
        if world == 1 and not a.no_extras:
            res["roofline_kernels"] = kernel_rooflines(unet, dev)
            res.update(extras(pipe, unet, dev, dtype))
        if sd_cpu is not None:
            # a measured full step (about 160 s on the GPU box's 128 host cores) is what the default line carries; small hosts keep the sample
            full_cpu = a.full_cpu_baseline or (not a.sample_cpu_baseline and (os.cpu_count() or 1) >= 64)
            rec, cinp, cout = cpu_baseline(sd_cpu, FRAMES if full_cpu else 2)
            res["cpu_baseline"] = rec
            del sd_cpu
            # the same sample through the bf16 HIP operator: the metric's "max|delta| vs CPU ref" 
            sample, ehs_c, audio_c, pose_c, fm, fc, lp = operator_args(cinp, dev)
            unet.set_banks({k: v.to(dev) for k, v in cinp["banks"].items()})
            pred = unet.forward(sample, 499, ehs_c, audio_c, pose_cond_fea=pose_c, full_mask=fm, face_mask=fc, body_mask=lp,
                                motion_scale=cinp["motion_scale"], return_dict=False)[0].float().cpu()
            d = (pred - cout).abs()
            res["max_abs_delta_vs_cpu"] = {"max_abs": float(d.max()), "mean_abs": float(d.mean()),
                                           "max_rel_to_absmax": float(d.max() / cout.abs().max()),
                                           "ref_mean_abs": float(cout.abs().mean()),
                                           "what": f"{a.dtype} HIP UNet3D forward vs the CPU fp32 oracle on the cpu_baseline sample"}
        print(json.dumps(res))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
