"""Device-side conditioning producers and output conversion against their oracles."""
import numpy as np
import pytest
import torch

from mmgt_amd import conditioning as C
from mmgt_amd.synthetic import hash_uniform
from oracle import conditioning_ref as R


def _masks(frames=5, size=64, tag="cm"):
    """Blob-like uint8 masks with hard edges and gradients."""
    u = hash_uniform(tag, (frames, size, size), 0.5) + 0.5
    yy, xx = torch.meshgrid(torch.arange(size), torch.arange(size), indexing="ij")
    blob = ((yy - size * 0.4) ** 2 + (xx - size * 0.55) ** 2 < (size * 0.22) ** 2).float()
    return ((0.7 * blob[None] + 0.3 * u) * 255).clamp(0, 255).to(torch.uint8)


def test_pil_tables_reproduce_pil_resize_bit_exactly_on_cpu():
    """The integer coefficient tables restated from Pillow's Resample.c, applied in plain integer arithmetic, equal
    PIL.Image.resize(BILINEAR) bit for bit: 64 -> 32, 16, 8 (the pyramid) and a non-integer ratio."""
    from PIL import Image
    m = _masks(3).numpy()
    for S, D in ((64, 32), (64, 16), (64, 8), (64, 24)):
        bounds, kk = C.pil_bilinear_tables(S, D)
        bounds, kk = bounds.numpy(), kk.numpy().astype(np.int64)
        for img in m:
            tmp = np.zeros((S, D), np.int64)
            for x in range(D):
                x0, n = bounds[x]
                tmp[:, x] = np.clip(((1 << 21) + (img[:, x0:x0 + n].astype(np.int64) * kk[x, :n]).sum(1)) >> 22, 0, 255)
            out = np.zeros((D, D), np.int64)
            for y in range(D):
                y0, n = bounds[y]
                out[y] = np.clip(((1 << 21) + (tmp[y0:y0 + n] * kk[y, :n, None]).sum(0)) >> 22, 0, 255)
            want = np.asarray(Image.fromarray(img, mode="L").resize((D, D), Image.BILINEAR))
            assert np.array_equal(out, want), (S, D)


def test_blur_mask_oracle_properties():
    """The unpinned blur_mask restatement: constant -> zeros (min == max), output spans 0..255, impulse -> separable Gaussian."""
    assert R.blur_mask_ref(np.full((128, 128), 77, np.uint8), 31).max() == 0
    m = np.zeros((64, 64), np.uint8)
    m[32, 32] = 255
    b = R.blur_mask_ref(m, 21).astype(np.float64)
    assert b.max() == 255 and b.min() == 0 and b[32, 32] == 255 and abs(b[32, 28] - b[28, 32]) <= 1 and b[32, 28] < b[32, 30]


@pytest.mark.gpu
def test_device_mask_pyramid_bit_exact_with_pil():
    m = _masks(7)
    got = C.mask_pyramid_device(m.cuda(), 512)
    want = R.mask_pyramid_pil(m.numpy(), 512)
    assert [tuple(g.shape) for g in got] == [(7, 4096), (7, 1024), (7, 256), (7, 64)]
    for g, w in zip(got, want):
        # the resampled 8-bit values are PIL's bit for bit; ToTensor's / 255 may differ in the last ulp between the host's and
        # the device's fp32 division
        assert torch.equal((g.cpu() * 255).round().to(torch.uint8), (w * 255).round().to(torch.uint8))
        torch.testing.assert_close(g.cpu(), w, rtol=0, atol=1e-7)
    from mmgt_amd import hip
    from PIL import Image
    b, c = C.pil_bilinear_tables(64, 24)                       # a non-integer ratio, raw uint8 out
    raw = hip.resample_u8(m.cuda(), 24, b.cuda(), c.cuda(), as_float=False).cpu().numpy()
    for i in range(7):
        assert np.array_equal(raw[i], np.asarray(Image.fromarray(m[i].numpy(), mode="L").resize((24, 24), Image.BILINEAR)))


@pytest.mark.gpu
@pytest.mark.parametrize("ksize,size", [(31, 512), (21, 512), (31, 64), (21, 200)])
def test_device_blur_mask_matches_restatement(ksize, size):
    m = _masks(4, size, tag=f"bm{size}")
    got = C.blur_mask_device(m.cuda(), ksize).cpu().numpy()
    for i in range(4):
        want = R.blur_mask_ref(m[i].numpy(), ksize)
        d = np.abs(got[i].astype(int) - want.astype(int))
        assert d.max() <= 1 and (d > 0).mean() < 0.02, (d.max(), (d > 0).mean())     # rounding ties of float32 vs float64 sums only


@pytest.mark.gpu
def test_device_audio_window_and_uint8_frames():
    from mmgt_amd import hip
    x = hash_uniform("aw", (9, 12, 768), 1.0)
    torch.testing.assert_close(C.process_audio_emb_device(x.cuda()).cpu(), C.process_audio_emb(x), rtol=0, atol=0)
    for dt in (torch.float32, torch.bfloat16):
        dec = (hash_uniform("u8", (3, 16, 16, 64), 1.3)).to(dt)
        got = hip.frames_to_u8(dec.cuda()).cpu().numpy()
        assert np.array_equal(got, R.frames_to_uint8_ref(dec[..., :3]))


def test_video_out_layout_and_gif(tmp_path):
    from mmgt_amd.video_out import frames_uint8, save_videos_grid
    v = hash_uniform("vo", (1, 3, 4, 8, 8), 0.5) + 0.5
    f = frames_uint8(v)
    assert f.shape == (4, 8, 8, 3) and f.dtype == np.uint8
    assert np.array_equal(f, (v[0].permute(1, 2, 3, 0) * 255).numpy().astype(np.uint8))
    u8 = torch.from_numpy(f)[None]
    assert np.array_equal(frames_uint8(u8), f)
    save_videos_grid(v, str(tmp_path / "a.gif"), fps=8)
    save_videos_grid(u8, str(tmp_path / "a.npy"))
    assert np.array_equal(np.load(tmp_path / "a.npy"), f)
    with pytest.raises(RuntimeError, match="PyAV"):
        save_videos_grid(v, str(tmp_path / "a.mp4"))
