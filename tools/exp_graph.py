"""Experiment: capture one benchmarked UNet3D forward (512x512x24, CFG batch 2, bf16) into a HIP graph and compare the replay
time with the eager launch sequence (683 launches).  Reported in DESIGN.md; nothing in the product depends on this file."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

def main():
    dev = torch.device("cuda", 0)
    from mmgt_amd.synthetic import synth_state_dict
    from mmgt_amd.unet3d import UNet3DConditionModel
    from mmgt_amd.unet3d_spec import unet3d_spec
    unet = UNet3DConditionModel(device=dev, dtype=torch.bfloat16)
    unet.load_state_dict(synth_state_dict(unet3d_spec(), device=dev))
    unet.enable_gradient_checkpointing()
    inp = bench.build_inputs(dev)
    unet.set_banks(inp["banks"])
    sample, ehs, audio, pose, fm, fc, lp = bench.operator_args(inp, dev)

    tstep = torch.tensor([499.0], device=dev)

    def fwd():
        return unet.forward(sample, tstep, ehs, audio, pose_cond_fea=pose, full_mask=fm, face_mask=fc, body_mask=lp,
                            motion_scale=inp["motion_scale"], return_dict=False)[0]

    def timeit(fn, reps=10):
        for _ in range(2):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps, 1e3 * (time.perf_counter() - t0) / reps

    ref = fwd()
    print("eager   dev %.2f ms  wall %.2f ms" % timeit(fwd), flush=True)
    t0 = time.perf_counter()
    for _ in range(5):
        fwd()
    print("host-only launch time %.2f ms/forward" % (1e3 * (time.perf_counter() - t0) / 5), flush=True)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        fwd()
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    with torch.cuda.graph(g):
        out = fwd()
    g.replay()
    torch.cuda.synchronize()
    print("graph == eager:", bool((out == ref).all()), flush=True)
    print("graph   dev %.2f ms  wall %.2f ms" % timeit(g.replay), flush=True)

if __name__ == "__main__":
    main()
