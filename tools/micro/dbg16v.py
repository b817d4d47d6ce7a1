import sys, torch
sys.path.insert(0, '.')
from mmgt_amd import hip
dev = torch.device('cuda:0')
hip.tune('gemm_cfg', 16)
for (M, N, K) in [(256, 256, 64), (256, 256, 128), (256, 256, 192), (256, 256, 320), (512, 256, 128), (256, 512, 128), (2048, 2048, 256), (70000, 1280, 320)]:
    g = torch.Generator(device='cpu').manual_seed(1)
    a = torch.randint(-1, 2, (M, K), generator=g).to(dev).bfloat16()
    w = torch.randint(-1, 2, (N, K), generator=g).to(dev).bfloat16()
    a[:, 1] = (torch.arange(M, device=dev) % 7 - 3).bfloat16()
    w[:, 0] = (torch.arange(N, device=dev) % 5 - 2).bfloat16()
    ref = a.double() @ w.double().t()
    hip.tune('g16_ver', 3)
    out = hip.gemm(a, w).double()
    hip.tune('g16_ver', 1)
    bad = out != ref
    print(M, N, K, 'bad', int(bad.sum()), 'of', bad.numel(), end=' | ')
    if bad.any():
        rows = bad.any(1).nonzero().flatten(); cols = bad.any(0).nonzero().flatten()
        print('rows', rows[:8].tolist(), '..', int(rows[-1]), len(rows), 'cols', cols[:8].tolist(), '..', int(cols[-1]), len(cols))
        # per 16x16 tile of the first 256x256: fraction bad
        t = bad[:256, :256].reshape(16, 16, 16, 16).permute(0, 2, 1, 3).reshape(16, 16, 256).float().mean(-1)
        print((t > 0).int())
    else:
        print('ok')
