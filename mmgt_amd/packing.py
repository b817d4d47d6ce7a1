"""Host-side weight repacking for the HIP kernels (done once per load_state_dict; plumbing, no hot-path arithmetic)."""
import torch


def pack_geglu(w, b):
    """GEGLU proj (8C, C): interleave per 32 output channels as [32 h rows | 32 gate rows] so that the GEMM epilogue finds
    h and gate of one output element in the same lane (include/mmgt_hip.h: MMGT_ACT_GEGLU)."""
    n2, k = w.shape
    n = n2 // 2
    assert n % 32 == 0
    wp = w.reshape(2, n // 32, 32, k).permute(1, 0, 2, 3).reshape(n2, k).contiguous()
    bp = None if b is None else b.reshape(2, n // 32, 32).permute(1, 0, 2).reshape(n2).contiguous()
    return wp, bp


def pack_conv3x3(w, cin_pad=None, cout_pad=None):
    """(Cout, Cin, 3, 3) -> [Cout][3][3][Cin] (K = (ky, kx, cin), cin fastest), optionally zero padded."""
    cout, cin = w.shape[:2]
    cin_pad = cin_pad or cin
    cout_pad = cout_pad or cout
    out = torch.zeros((cout_pad, 3, 3, cin_pad), device=w.device, dtype=w.dtype)
    out[:cout, :, :, :cin] = w.permute(0, 2, 3, 1)
    return out.contiguous()


def pad_rows(w, n_pad):
    """Zero-pad the output (row) dimension of a [N, K] weight / [N] bias."""
    if w is None or w.shape[0] == n_pad:
        return w
    out = torch.zeros((n_pad,) + tuple(w.shape[1:]), device=w.device, dtype=w.dtype)
    out[: w.shape[0]] = w
    return out


def pad_cols(w, k_pad):
    if w.shape[1] == k_pad:
        return w.contiguous()
    out = torch.zeros((w.shape[0], k_pad), device=w.device, dtype=w.dtype)
    out[:, : w.shape[1]] = w
    return out


def round_up(x, m):
    return (x + m - 1) // m * m
