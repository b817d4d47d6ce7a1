#!/usr/bin/env python3
"""bench.py — UNet3D denoise-steps/s of the MMGT Stage-2 path on MI355X (BASELINE.json metric, config 2).

One "step" = one DDIM iteration of Pose2VideoPipeline's loop over one 24-frame 512x512 clip (latents (1,4,24,64,64)):
CFG-batched UNet3D forward on (2,4,24,64,64) with pose features, motion masks, 32 audio tokens per frame and the 16
reference-attention banks, then window accumulate + CFG combine + DDIM update.  bf16 storage, fp32 accumulate.
All inputs are resident in HBM before the timed region.  N GPUs = N independent clips (clip-parallel, no collective in
the loop), so `value` is the aggregate over ranks and scaling is "weak".

    python bench.py [--gpus N] [--steps K] [--warmup W] [--no-cpu-baseline]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py --gpus N ...
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

ALGO_TFLOP_PER_STEP = 62.0      # minimal algorithmic FLOPs of one CFG denoise step at 512x512x24
EXEC_TFLOP_PER_STEP = 71.6      # as executed by the reference's op graph (quoted alongside, never the numerator)
PEAK_BF16_TFLOPS = 2500.0       # MI355X dense bf16 MFMA (MI355X_MICROARCH.md)
FRAMES, LATENT = 24, 64


def build_inputs(dev, frames=FRAMES, latent=LATENT, tag="bench"):
    from mmgt_amd.synthetic import hash_uniform, synth_masks
    from tests.golden_cases import bank_spatial
    case = dict(block_out_channels=(320, 640, 1280, 1280), latent=latent)
    lips = synth_masks(tag + ".lips", frames, latent)
    face = synth_masks(tag + ".face", frames, latent)
    full = [1 + l for l in lips]
    return dict(
        latents=hash_uniform(tag + ".latents", (1, 4, frames, latent, latent), 1.7).to(dev),
        clip=hash_uniform(tag + ".clip", (1, 768), 1.0).to(dev),
        audio=hash_uniform(tag + ".audio", (1, frames, 32, 768), 1.7).to(dev),
        pose=hash_uniform(tag + ".pose", (1, 320, frames, latent, latent), 0.5).to(dev),
        full=[m.to(dev) for m in full], face=[m.to(dev) for m in face], lips=[m.to(dev) for m in lips],
        banks={k: hash_uniform(tag + ".bank." + k, (2, n, c), 1.0).to(dev) for k, (n, c) in bank_spatial(case).items()},
        motion_scale=[1.0, 1.0, 2.0])


def cpu_baseline(sd_cpu, frames_sample=2):
    """Oracle (CPU fp32 restatement of the reference) timed on this box's host cores on a bounded sample: one CFG
    denoise-step forward at 512x512 with `frames_sample` of the 24 frames; cost is linear in frames (spatial ops are
    per frame; temporal attention is <0.2% of the FLOPs), so steps/s = 1 / (t * 24 / frames_sample)."""
    from oracle import unet3d_ref as R
    cores = max(1, (os.cpu_count() or 2) // 2)
    torch.set_num_threads(cores)
    inp = build_inputs("cpu", frames=frames_sample)
    cat2 = lambda L: [torch.cat([m] * 2) for m in L]
    sample = inp["latents"].repeat(2, 1, 1, 1, 1)
    ehs = torch.cat([torch.zeros(1, 1, 768), inp["clip"].reshape(1, 1, 768)])
    audio = torch.cat([torch.zeros_like(inp["audio"]), inp["audio"]])
    pose = inp["pose"].repeat(2, 1, 1, 1, 1)
    t0 = time.time()
    with torch.no_grad():
        out = R.unet3d_forward(sd_cpu, R.UNet3DConfig(), sample, torch.tensor(499), ehs, audio, pose, cat2(inp["full"]),
                               cat2(inp["face"]), cat2(inp["lips"]), inp["motion_scale"], inp["banks"], weighted=True)
    dt = time.time() - t0
    assert torch.isfinite(out).all()
    return {"value": 1.0 / (dt * FRAMES / frames_sample), "unit": "steps/s", "cores": cores, "kind": "port",
            "sample": f"one CFG denoise-step forward of the oracle (PyTorch CPU fp32) at 512x512 on {frames_sample} of 24 "
                      f"frames: {dt:.1f} s, scaled x{FRAMES // frames_sample} to 24 frames",
            "cpu": _cpu_name()}


def pmc_traffic():
    """HBM-side bytes per denoise step from the committed rocprofv3 PMC passes (profiles/r*/pmc_traffic*.json: FETCH_SIZE and
    WRITE_SIZE collected in separate runs of this same command, gfx950 FETCH_SIZE x2 correction applied)."""
    import glob
    latest = os.path.join(ROOT, "profiles", "r1", "pmc_traffic_s55.json")     # the passes that belong to the current build
    files = [latest] if os.path.exists(latest) else sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", "pmc_traffic*.json")))
    if not files:
        return None, None
    d = json.load(open(files[-1]))
    return d["total_GB_per_step"] * 1e9, os.path.relpath(files[-1], ROOT)


def _cpu_name():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f32"])
    a = ap.parse_args()

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
                         "note": f"whole denoise step on one GPU: {ALGO_TFLOP_PER_STEP} TFLOP algorithmic per step "
                                 f"(SURVEY 8d; {EXEC_TFLOP_PER_STEP} as executed by the reference) / HIP-event time; "
                                 f"traffic = bytes per step from {traffic_src}"},
        }
        if sd_cpu is not None:
            res["cpu_baseline"] = cpu_baseline(sd_cpu)
        print(json.dumps(res))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
