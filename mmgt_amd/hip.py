"""ctypes binding of libmmgt_hip.so (include/mmgt_hip.h): the ONLY compute entry points of the product.

There is no fallback: if the library is missing or a call fails, a RuntimeError is raised.  torch is used for device
memory (the caching allocator) and the current HIP stream only.
"""
import ctypes
import os
from ctypes import c_float, c_int, c_long, c_void_p

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MMGT_LIB") or os.path.join(_HERE, "libmmgt_hip.so")    # MMGT_LIB: a diagnostic build (make trace), tools only

F32, BF16 = 0, 1
ACT_NONE, ACT_GEGLU, ACT_SILU, ACT_RELU, ACT_QUICK_GELU, ACT_GELU, ACT_MISH = 0, 1, 2, 3, 4, 5, 6

_lib = None
DMA_LIMIT = (1 << 31) - 1        # the kernels address every operand through 32-bit LDS-DMA offsets: 2 GiB per tensor

_SIGS = {
    "mmgt_abi_version": (c_int, []),
    "mmgt_last_error": (ctypes.c_char_p, []),
    "mmgt_tune": (c_int, [ctypes.c_char_p, c_int]),
    "mmgt_tune_get": (c_int, [ctypes.c_char_p, ctypes.POINTER(c_int)]),
    "mmgt_box_calib": (c_int, [c_float, ctypes.POINTER(c_float), ctypes.POINTER(c_float), c_void_p]),
    "mmgt_gemm16_set_trace": (None, [c_void_p]),
    "mmgt_ffn_set_trace": (None, [c_void_p]),
    "mmgt_rconv_set_trace": (None, [c_void_p]),
    "mmgt_gemm": (c_int, [c_void_p, c_long, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_float, c_void_p, c_long,
                          c_void_p, c_long, c_int, c_int, c_int, c_int, c_int, c_long, c_long, c_long, c_long, c_int,
                          c_void_p]),
    "mmgt_gemm_post": (c_int, [c_void_p, c_long, c_void_p, c_void_p, c_void_p, c_float, c_void_p, c_void_p, c_long, c_void_p,
                               c_long, c_int, c_int, c_int, c_int, c_void_p]),
    "mmgt_conv3x3_nhwc": (c_int, [c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p,
                                  c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p]),
    "mmgt_groupnorm_chunks": (c_int, [c_int]),
    "mmgt_groupnorm_nhwc": (c_int, [c_void_p, c_int, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int,
                                    c_int, c_int, c_float, c_int, c_int, c_void_p]),
    "mmgt_layernorm": (c_int, [c_void_p, c_long, c_void_p, c_void_p, c_float, c_void_p, c_int, c_int, c_void_p, c_long,
                               c_int, c_int, c_int, c_void_p]),
    "mmgt_attention": (c_int, [c_void_p, c_long, c_long, c_long, c_void_p, c_long, c_long, c_long, c_void_p, c_long,
                               c_long, c_long, c_void_p, c_long, c_long, c_long, c_int, c_void_p, c_void_p, c_long,
                               c_long, c_long, c_long, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_float,
                               c_int, c_int, c_void_p]),
    "mmgt_attention_scaled": (c_int, [c_void_p, c_long, c_long, c_long, c_void_p, c_long, c_long, c_long, c_void_p, c_long, c_long, c_long,
                                      c_void_p, c_long, c_long, c_long, c_void_p, c_long, c_int, c_int, c_int, c_int, c_int, c_int, c_float,
                                      c_int, c_void_p]),
    "mmgt_attention_twin": (c_int, [c_void_p, c_long, c_long, c_long, c_void_p, c_long, c_long, c_long, c_void_p, c_long, c_long, c_long,
                                    c_void_p, c_void_p, c_long, c_long, c_long, c_int, c_void_p, c_void_p, c_long, c_long, c_long, c_long,
                                    c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_float, c_int, c_void_p]),
    "mmgt_softmax_rows": (c_int, [c_void_p, c_long, c_void_p, c_long, c_int, c_int, c_float, c_int, c_void_p]),
    "mmgt_gemm_bf16_f32": (c_int, [c_void_p, c_long, c_void_p, c_long, c_void_p, c_long, c_int, c_int, c_int, c_void_p]),
    "mmgt_qk_split3": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_long, c_int, c_void_p]),
    "mmgt_softmax_rows_f32_bf16": (c_int, [c_void_p, c_long, c_void_p, c_long, c_int, c_int, c_float, c_void_p]),
    "mmgt_ncfhw_to_nhwc": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_float, c_int, c_void_p]),
    "mmgt_nhwc_to_ncfhw": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_float, c_float, c_int,
                                   c_int, c_void_p]),
    "mmgt_conv1x1_cat_nhwc": (c_int, [c_void_p, c_int, c_void_p, c_int, c_long, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p]),
    "mmgt_conv_taps_gather": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    "mmgt_timestep_features": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p]),
    "mmgt_ff_fused_image_bytes": (c_int, [c_int, c_int]),
    "mmgt_ff_fused": (c_int, [c_void_p, c_long, c_void_p, c_void_p, c_float, c_void_p, c_void_p, c_void_p, c_long, c_void_p, c_long,
                              c_int, c_int, c_int, c_int, c_void_p]),
    "mmgt_rowgemm320_image_bytes": (c_long, [c_int]),
    "mmgt_rowgemm_set_trace": (None, [c_void_p]),
    "mmgt_ff_fused_po": (c_int, [c_void_p, c_long, c_void_p, c_void_p, c_float, c_void_p, c_void_p, c_void_p, c_long, c_void_p, c_void_p,
                                 c_void_p, c_long, c_void_p, c_long, c_int, c_int, c_int, c_int, c_void_p]),
    "mmgt_groupnorm_affine": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_float, c_int,
                                      c_void_p]),
    "mmgt_rowgemm320": (c_int, [c_void_p, c_long, c_int, c_void_p, c_void_p, c_int, c_int, c_float, c_void_p, c_void_p, c_void_p, c_int, c_void_p,
                                c_long, c_void_p, c_long, c_int, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "mmgt_gn_silu_conv3x3_image_bytes": (c_long, [c_int, c_int]),
    "mmgt_gn_silu_conv3x3": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "mmgt_gn_silu_conv3x3_unet_image_bytes": (c_long, [c_int, c_int]),
    "mmgt_gn_silu_conv3x3_unet": (c_int, [c_void_p, c_int, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p,
                                          c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "mmgt_gn_silu_conv3x3_unet_stats_rows": (c_int, [c_int, c_int, c_int, c_int]),
    "mmgt_gn_silu_conv3x3_unet_stats": (c_int, [c_void_p, c_int, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p,
                                                c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "mmgt_gn_stats_finalize_unet": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_float, c_void_p]),
    "mmgt_groupnorm_affine2": (c_int, [c_void_p, c_int, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int,
                                       c_float, c_int, c_void_p]),
    "mmgt_gn_stats_finalize": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_float, c_void_p]),
    "mmgt_temporal_leg320_image_bytes": (c_long, []),
    "mmgt_temporal_leg320": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_int, c_int, c_int, c_float, c_float, c_int,
                                     c_void_p]),
    "mmgt_channel_norm_gelu": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_float, c_int, c_void_p]),
    "mmgt_lerp_rows": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    "mmgt_silu": (c_int, [c_void_p, c_void_p, c_long, c_int, c_void_p]),
    "mmgt_cfg_ddim_step": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_long, c_int, c_int, c_float, c_float,
                                   c_float, c_float, c_float, c_void_p]),
    "mmgt_accumulate_window": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int,
                                       c_void_p]),
    "mmgt_rotary": (c_int, [c_void_p, c_void_p, c_void_p, c_long, c_int, c_int, c_int, c_void_p]),
    "mmgt_film_residual": (c_int, [c_void_p, c_void_p, c_long, c_void_p, c_void_p, c_void_p, c_long, c_int, c_int, c_int, c_void_p]),
    "mmgt_mean_tokens": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    "mmgt_activation": (c_int, [c_void_p, c_void_p, c_long, c_int, c_int, c_void_p]),
    "mmgt_smga_ddim_step": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_long, c_float, c_float, c_float, c_float,
                                    c_float, c_float, c_int, c_int, c_void_p]),
    "mmgt_blur_mask_u8": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    "mmgt_resample_u8": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_int, c_void_p]),
    "mmgt_window_stack": (c_int, [c_void_p, c_void_p, c_int, c_long, c_int, c_void_p]),
    "mmgt_frames_to_u8": (c_int, [c_void_p, c_void_p, c_long, c_int, c_float, c_float, c_int, c_void_p]),
    "mmgt_dwpose_draw": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p]),
    "mmgt_accumulate_window_rows": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int,
                                            c_int, c_int, c_int, c_int, c_void_p]),
}

EXPORTS = tuple(_SIGS)


def lib():
    """Load libmmgt_hip.so once; raise loudly if it has not been built (python __graft_entry__.py / make -C mmgt_amd/csrc)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(f"{LIB_PATH} is missing: build it with `make -C mmgt_amd/csrc` (hipcc, gfx950). "
                               "mmgt_amd has no non-HIP compute path.")
        L = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in _SIGS.items():
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        _lib = L
        for kv in filter(None, os.environ.get("MMGT_TUNE", "").split(",")):     # A/B switches for the tools: MMGT_TUNE="splitk=0,attn64=0"
            k, v = kv.split("=")
            if L.mmgt_tune(k.strip().encode(), int(v)) != 0:
                raise RuntimeError(f"MMGT_TUNE: {L.mmgt_last_error().decode()}")
    return _lib


_calls = {}


def _check(rc, what):
    _calls[what] = _calls.get(what, 0) + 1
    if rc != 0:
        raise RuntimeError(f"{what} failed (rc={rc}): {lib().mmgt_last_error().decode()}")


def call_count(what):
    """How often the entry point `what` (an include/mmgt_hip.h name) has been called through this binding: lets a test assert WHICH kernel a
    dispatcher took."""
    return _calls.get(what, 0)


def tune(key, value):
    _check(lib().mmgt_tune(key.encode(), int(value)), "mmgt_tune")


def tune_get(key):
    """A host-side switch of the library's table (include/mmgt_hip.h: mmgt_tune_get)."""
    v = c_int(0)
    _check(lib().mmgt_tune_get(key.encode(), ctypes.byref(v)), "mmgt_tune_get")
    return v.value


def box_calib(warm_seconds=2.0):
    """(in-kernel MHz, TFLOP/s) of a bare 16x16x32 bf16 MFMA loop on random operands after `warm_seconds` of back-to-back launches: a
    measure of the BOX, quoted beside a step time so that lines from different boxes can be compared (include/mmgt_hip.h: mmgt_box_calib)."""
    mhz, tf = c_float(0), c_float(0)
    _check(lib().mmgt_box_calib(float(warm_seconds), ctypes.byref(mhz), ctypes.byref(tf), _stream()), "mmgt_box_calib")
    return mhz.value, tf.value


def dtype_code(dt):
    if dt == torch.bfloat16:
        return BF16
    if dt == torch.float32:
        return F32
    raise RuntimeError(f"mmgt_amd supports float32 and bfloat16 storage, got {dt}")


def _ptr(t):
    return None if t is None else t.data_ptr()


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _dev(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise RuntimeError("mmgt_amd.hip: tensors must live on the GPU (no CPU fallback exists)")


def _f32(t, name):
    if t is not None and (t.dtype != torch.float32 or not t.is_contiguous()):
        raise RuntimeError(f"{name} must be contiguous float32")
    return t


# ------------------------------------------------------------------------------------------------------------ GEMM

def gemm(a, w, bias=None, *, out=None, residual=None, bias2=None, bias2_rows=0, row_scale=None, alpha=1.0,
         act=ACT_NONE):
    """out[M, Nout] = epilogue(a[M, K] @ w[N, K]^T).  a may be a strided 2-D view (row stride, unit column stride)."""
    _dev(a, w, bias, out, residual, bias2, row_scale)
    assert a.dim() == 2 and w.dim() == 2 and a.stride(1) == 1 and w.is_contiguous() and a.dtype == w.dtype
    M, K = a.shape
    N = w.shape[0]
    assert w.shape[1] == K, (a.shape, w.shape)
    n_out = N // 2 if act == ACT_GEGLU else N
    if out is None:
        out = torch.empty((M, n_out), device=a.device, dtype=a.dtype)
    assert out.shape == (M, n_out) and out.stride(1) == 1 and out.dtype == a.dtype
    if residual is not None:
        assert residual.shape == (M, n_out) and residual.stride(1) == 1 and residual.dtype == a.dtype
    lim = DMA_LIMIT                    # an operand of 2 GiB or more is worked in runs of rows
    if M * a.stride(0) * a.element_size() > lim and M > 1:
        step = max(1, lim // (a.stride(0) * a.element_size()))
        # bias2 row groups: a run must start on a group boundary -- unless ONE row serves every output row (bias2_rows >= M: the
        # time embedding / the twin CLIP vector), which every run then takes as it is
        one_row = bias2 is not None and bias2_rows >= M
        if bias2 is not None and not one_row:
            step = step // bias2_rows * bias2_rows
            if step == 0:
                raise RuntimeError("gemm: a bias2 row group exceeds the 2 GiB range of the 32-bit LDS-DMA offsets")
        assert 0 < step < M
        for m0 in range(0, M, step):
            m1 = min(M, m0 + step)
            if bias2 is None:
                b2, b2r = None, 0
            elif one_row:
                b2, b2r = bias2[:1], max(m1 - m0, bias2_rows)
            else:
                b2, b2r = bias2[m0 // bias2_rows:(m1 + bias2_rows - 1) // bias2_rows], bias2_rows
            gemm(a[m0:m1], w, bias, out=out[m0:m1], residual=None if residual is None else residual[m0:m1], bias2=b2, bias2_rows=b2r,
                 row_scale=None if row_scale is None else row_scale[m0:m1], alpha=alpha, act=act)
        return out
    _check(lib().mmgt_gemm(_ptr(a), a.stride(0), _ptr(w), _ptr(_f32(bias, "bias")), _ptr(_f32(bias2, "bias2")),
                           bias2_rows, _ptr(_f32(row_scale, "row_scale")), alpha, _ptr(residual),
                           residual.stride(0) if residual is not None else 0, _ptr(out), out.stride(0), M, N, K, act, 1,
                           0, 0, 0, 0, dtype_code(a.dtype), _stream()), "mmgt_gemm")
    return out


def gemm_post(a, w, bias, row_scale, alpha, bias_post, residual, out=None):
    """out = (a @ w^T + bias) * row_scale[:, None] * alpha + bias_post + residual   (see mmgt_gemm_post)."""
    _dev(a, w, bias, row_scale, bias_post, residual, out)
    assert a.dim() == 2 and w.dim() == 2 and a.stride(1) == 1 and w.is_contiguous() and a.dtype == w.dtype
    M, K = a.shape
    N = w.shape[0]
    if out is None:
        out = torch.empty((M, N), device=a.device, dtype=a.dtype)
    assert out.shape == (M, N) and residual.shape == (M, N) and residual.stride(1) == 1 and out.stride(1) == 1
    _check(lib().mmgt_gemm_post(_ptr(a), a.stride(0), _ptr(w), _ptr(_f32(bias, "bias")), _ptr(_f32(row_scale, "row_scale")),
                                alpha, _ptr(_f32(bias_post, "bias_post")), _ptr(residual), residual.stride(0), _ptr(out),
                                out.stride(0), M, N, K, dtype_code(a.dtype), _stream()), "mmgt_gemm_post")
    return out


def gemm_batched_wx(w, x, bias_unused=None, *, out):
    """out[b] (R, ldo>=Ntok) = w[R, K] @ x[b][Ntok, K]^T for every batch b: the V^T projection (keys contiguous)."""
    _dev(w, x, out)
    B, ntok, K = x.shape
    R = w.shape[0]
    assert x.is_contiguous() and w.is_contiguous() and out.dim() == 3 and out.shape[0] == B and out.shape[1] == R
    assert out.stride(2) == 1
    _check(lib().mmgt_gemm(_ptr(w), K, _ptr(x), None, None, 0, None, 1.0, None, 0, _ptr(out), out.stride(1), R, ntok, K,
                           ACT_NONE, B, 0, x.stride(0), 0, out.stride(0), dtype_code(x.dtype), _stream()),
           "mmgt_gemm(batched)")
    return out


def gemm_batched(a, w, *, out):
    """out[b] (M, N) = a[b] (M, K) @ w[b] (N, K)^T for dense 3-D operands (one problem per grid.z)."""
    _dev(a, w, out)
    B, M, K = a.shape
    N = w.shape[1]
    assert a.is_contiguous() and w.is_contiguous() and out.is_contiguous() and w.shape == (B, N, K) and out.shape == (B, M, N)
    _check(lib().mmgt_gemm(_ptr(a), K, _ptr(w), None, None, 0, None, 1.0, None, 0, _ptr(out), N, M, N, K, ACT_NONE, B,
                           M * K, N * K, 0, M * N, dtype_code(a.dtype), _stream()), "mmgt_gemm(batched)")
    return out


def conv3x3(x, wp, bias=None, *, stride=1, upsample=False, bias2=None, bias2_rows=0, residual=None, act=ACT_NONE,
            x1=None, out=None, pad_high_only=False):
    """x (NB, H, W, C0) channels-last [+ x1 (NB, H, W, C1)], wp [Cout][3][3][C0+C1] -> (NB, OH, OW, Cout).
    pad_high_only (stride 2 only): zero padding after the last row / column only (the VAE encoder's downsampler).
    upsample = True: the conv sees the nearest-2x upsampled input; upsample = 2: the same result from the four-phase image
    packing.pack_conv3x3_up2(weight) ([4][Cout][2][2][C0]: four 2 x 2 convs on the stored image, 16 / 36 of the multiply-adds; bf16, bias only)."""
    _dev(x, wp, bias, bias2, residual, x1)
    assert x.dim() == 4 and x.is_contiguous() and wp.is_contiguous() and wp.dtype == x.dtype
    NB, IH, IW, C0 = x.shape
    C1 = 0 if x1 is None else x1.shape[3]
    if upsample is not True and upsample == 2:
        assert wp.dim() == 5 and wp.shape[0] == 4 and wp.shape[2:] == (2, 2, C0) and x1 is None and residual is None and bias2 is None
        assert x.dtype == torch.bfloat16 and act == ACT_NONE and stride == 1 and not pad_high_only
        cout = wp.shape[1]
        assert cout % 256 == 0 or cout % 320 == 0
    else:
        upsample = bool(upsample)
        cout = wp.shape[0]
        assert wp.numel() == cout * 9 * (C0 + C1)
    vh, vw = (IH * 2, IW * 2) if upsample else (IH, IW)
    if pad_high_only:
        assert stride == 2 and not upsample
        oh, ow = (vh - 2) // 2 + 1, (vw - 2) // 2 + 1
    else:
        oh, ow = (vh - 1) // stride + 1, (vw - 1) // stride + 1
    if out is None:
        out = torch.empty((NB, oh, ow, cout), device=x.device, dtype=x.dtype)
    assert out.shape == (NB, oh, ow, cout) and out.is_contiguous()
    if residual is not None:
        assert residual.shape == out.shape and residual.is_contiguous()
    # The kernels address every operand through 32-bit LDS-DMA offsets (2 GiB per tensor).  Larger tensors -- the PoseGuider and
    # VAE convs of a 96-frame 512 x 512 clip -- are worked in runs of whole images: images are independent in a conv.
    lim = DMA_LIMIT
    per_img = max(t.numel() // NB * t.element_size() for t in (x, out, x1, residual) if t is not None)
    if per_img * NB > lim and NB > 1:
        step = max(1, lim // per_img)
        for n0 in range(0, NB, step):
            n1 = min(NB, n0 + step)
            b2, b2r = None, bias2_rows
            if bias2 is not None:
                m0, m1 = n0 * oh * ow, n1 * oh * ow
                if bias2_rows >= NB * oh * ow:              # ONE row for every output row (see gemm)
                    b2, b2r = bias2[:1], max(m1 - m0, bias2_rows)
                elif m0 % bias2_rows:
                    raise RuntimeError("conv3x3: a batch split inside a bias2 row group is not supported")
                else:
                    b2 = bias2[m0 // bias2_rows:(m1 + bias2_rows - 1) // bias2_rows]
            conv3x3(x[n0:n1], wp, bias, stride=stride, upsample=upsample, bias2=b2, bias2_rows=b2r,
                    residual=None if residual is None else residual[n0:n1], act=act, x1=None if x1 is None else x1[n0:n1],
                    out=out[n0:n1], pad_high_only=pad_high_only)
        return out
    _check(lib().mmgt_conv3x3_nhwc(_ptr(x), C0, _ptr(x1), C1, NB, IH, IW, -2 if pad_high_only else stride, int(upsample), _ptr(wp),
                                   _ptr(_f32(bias, "bias")), _ptr(_f32(bias2, "bias2")), bias2_rows, _ptr(residual),
                                   _ptr(out), cout, act, dtype_code(x.dtype), _stream()), "mmgt_conv3x3_nhwc")
    return out


# ------------------------------------------------------------------------------------------------------------ norms

def groupnorm(x, gamma, beta, groups, eps, silu=False, x1=None, out=None):
    """x (NB, HW, C0) [+ x1 (NB, HW, C1)] -> (NB, HW, C0+C1), per-image GroupNorm (+ SiLU)."""
    _dev(x, gamma, beta, x1)
    assert x.dim() == 3 and x.is_contiguous()
    NB, HW, C0 = x.shape
    C1 = 0 if x1 is None else x1.shape[2]
    if x1 is not None:
        assert x1.is_contiguous() and x1.shape[:2] == x.shape[:2] and x1.dtype == x.dtype
    if out is None:
        out = torch.empty((NB, HW, C0 + C1), device=x.device, dtype=x.dtype)
    chunks = lib().mmgt_groupnorm_chunks(HW)
    ws = torch.empty((NB * chunks * groups * 2,), device=x.device, dtype=torch.float32)
    _check(lib().mmgt_groupnorm_nhwc(_ptr(x), C0, _ptr(x1), C1, _ptr(_f32(gamma, "gamma")), _ptr(_f32(beta, "beta")),
                                     _ptr(out), _ptr(ws), NB, HW, groups, eps, int(silu), dtype_code(x.dtype),
                                     _stream()), "mmgt_groupnorm_nhwc")
    return out


def groupnorm_affine(x, gamma, beta, groups, eps, x1=None):
    """x (NB, HW, C) [+ x1 (NB, HW, C1): the channel concatenation] -> (scale, shift), fp32 (NB, C + C1) each:
    GroupNorm(x | x1)[n, p, c] = (x | x1)[n, p, c] * scale[n, c] + shift[n, c].
    The statistics pass of `groupnorm` alone; `rowgemm320(pre_scale=, pre_shift=)` / `gn_silu_conv3x3_unet` apply the tables while they load x."""
    _dev(x, gamma, beta, x1)
    assert x.dim() == 3 and x.is_contiguous()
    NB, HW, C = x.shape
    C1 = 0
    if x1 is not None:
        assert x1.is_contiguous() and x1.shape[:2] == x.shape[:2] and x1.dtype == x.dtype
        C1 = x1.shape[2]
    chunks = lib().mmgt_groupnorm_chunks(HW)
    ws = torch.empty((NB * chunks * groups * 2,), device=x.device, dtype=torch.float32)
    tab = torch.empty((2, NB, C + C1), device=x.device, dtype=torch.float32)       # one allocation: the fused convs fetch scale | shift rows with one descriptor
    scale, shift = tab[0], tab[1]
    _check(lib().mmgt_groupnorm_affine2(_ptr(x), C, _ptr(x1), C1, _ptr(_f32(gamma, "gamma")), _ptr(_f32(beta, "beta")), _ptr(ws), _ptr(scale),
                                        _ptr(shift), NB, HW, groups, eps, dtype_code(x.dtype), _stream()), "mmgt_groupnorm_affine2")
    return scale, shift


def gn_silu_conv3x3_unet_supported(dtype, c0, c1, cout, H, W):
    """csrc/rconv.hip: bf16, 16 x 16 pixel tiles, 64-channel phases, output blocks of 320 / 256 / 160 channels"""
    return dtype == torch.bfloat16 and c0 % 64 == 0 and c1 % 64 == 0 and c0 > 0 and cout % 160 == 0 and H % 16 == 0 and W % 16 == 0


def gn_silu_conv3x3_unet(x, scale, shift, wimg, cout, bias=None, bias2=None, b2_imgs=0, residual=None, x1=None, out=None, next_norm=None):
    """bias + bias2[n // b2_imgs] + conv3x3(silu((x | x1) * scale[n, c] + shift[n, c])) (+ residual) in one launch (csrc/rconv.hip): x (NB, H, W, C0)
    [, x1 (NB, H, W, C1)] bf16 channels-last, (scale, shift) the tables of `groupnorm_affine` (one allocation), wimg = packing.pack_rconv(weight).
    next_norm = (gamma, beta, groups, eps) of the GroupNorm that reads the result: returns (out, (scale, shift) of THAT norm) -- its statistics
    come from the launch's epilogue and a small fold (mmgt_gn_stats_finalize_unet), the pass over the tensor is not needed."""
    _dev(x, scale, shift, wimg, bias, bias2, residual, x1, out)
    assert x.dim() == 4 and x.is_contiguous()
    NB, H, W, C0 = x.shape
    C1 = 0
    if x1 is not None:
        assert x1.is_contiguous() and x1.shape[:3] == x.shape[:3] and x1.dtype == x.dtype
        C1 = x1.shape[3]
    assert gn_silu_conv3x3_unet_supported(x.dtype, C0, C1, cout, H, W)
    C = C0 + C1
    assert scale.shape == (NB, C) and shift.shape == (NB, C) and scale.dtype == torch.float32 and scale.is_contiguous() and shift.is_contiguous()
    assert shift.data_ptr() == scale.data_ptr() + 4 * NB * C, "scale and shift must be the two halves of one allocation (groupnorm_affine)"
    assert wimg.dtype == torch.uint8 and wimg.is_contiguous() and wimg.numel() == lib().mmgt_gn_silu_conv3x3_unet_image_bytes(C, cout)
    if out is None:
        out = torch.empty((NB, H, W, cout), device=x.device, dtype=x.dtype)
    assert out.shape == (NB, H, W, cout) and out.is_contiguous() and out.dtype == x.dtype
    if residual is not None:
        assert residual.shape == out.shape and residual.is_contiguous() and residual.dtype == x.dtype
    if bias2 is not None:
        assert bias2.dim() == 2 and bias2.shape[1] == cout and bias2.is_contiguous() and b2_imgs > 0 and (NB + b2_imgs - 1) // b2_imgs <= bias2.shape[0]
    stats = None
    if next_norm is not None:
        rows = lib().mmgt_gn_silu_conv3x3_unet_stats_rows(NB, H, W, cout)
        assert rows in (2, 4)
        ppi = (H // 16) * (W // 16) * (16 // rows)
        stats = torch.empty((3, NB * ppi, cout), device=x.device, dtype=torch.float32)
    _check(lib().mmgt_gn_silu_conv3x3_unet_stats(_ptr(x), C0, _ptr(x1), C1, _ptr(scale), _ptr(wimg), _ptr(_f32(bias, "bias")), _ptr(_f32(bias2, "bias2")),
                                                 b2_imgs, _ptr(residual), _ptr(out), _ptr(stats), NB, H, W, cout, dtype_code(x.dtype), _stream()),
           "mmgt_gn_silu_conv3x3_unet")
    if next_norm is None:
        return out
    gamma, beta, groups, eps = next_norm
    tab = torch.empty((2, NB, cout), device=x.device, dtype=torch.float32)
    _check(lib().mmgt_gn_stats_finalize_unet(_ptr(stats), _ptr(_f32(gamma, "gamma")), _ptr(_f32(beta, "beta")), _ptr(tab), NB, ppi, 16 * rows, cout, groups,
                                             eps, _stream()), "mmgt_gn_stats_finalize_unet")
    return out, (tab[0], tab[1])


def gn_silu_conv3x3_supported(dtype, cin, cout, H, W, residual=False):
    """(Cin, Cout) = (128, 128), (256, 128), (256, 256: two launches on the halves of the output channels), (128, 64: no residual)"""
    return (dtype == torch.bfloat16 and ((cin, cout) in ((128, 128), (256, 128), (256, 256)) or ((cin, cout) == (128, 64) and not residual))
            and H % 16 == 0 and W % 16 == 0)


def gn_silu_conv3x3_tables(x, scale, shift, wimg, cout, bias=None, residual=None, out=None, stats=None):
    """bias + conv3x3(silu(x * scale[n, c] + shift[n, c])) (+ residual) (csrc/gnconv.hip): x (NB, H, W, Cin) bf16 channels-last, scale / shift
    (NB, Cin) fp32 (the tables of `groupnorm_affine`), wimg = packing.pack_gnconv(weight (cout, Cin, 3, 3)): one launch per 128 output channels.
    stats: or a (NB * (H / 16) * (W / 16), cout / 4, 2) fp32 tensor that receives the per-tile (sum, sum of squares) of the stored values
    (`gn_tables_from_stats` turns them into the tables of the GroupNorm that reads `out`)."""
    _dev(x, scale, shift, wimg, bias, residual, out, stats)
    assert x.dim() == 4 and x.is_contiguous()
    NB, H, W, C = x.shape
    assert gn_silu_conv3x3_supported(x.dtype, C, cout, H, W, residual is not None) and x.numel() * x.element_size() <= DMA_LIMIT
    assert scale.shape == (NB, C) and shift.shape == (NB, C) and scale.dtype == torch.float32 and shift.dtype == torch.float32
    assert scale.is_contiguous() and shift.is_contiguous()
    if shift.data_ptr() != scale.data_ptr() + 4 * NB * C:                     # the kernel wants the two tables in one allocation
        tab = torch.stack([scale, shift])
        scale, shift = tab[0], tab[1]
    cl = min(cout, 128)                                                # output channels per launch
    per = lib().mmgt_gn_silu_conv3x3_image_bytes(C, cl)
    assert wimg.dtype == torch.uint8 and wimg.is_contiguous() and wimg.numel() == per * (cout // cl)
    if out is None:
        out = torch.empty((NB, H, W, cout), device=x.device, dtype=x.dtype)
    assert out.shape == (NB, H, W, cout) and out.is_contiguous() and out.dtype == x.dtype and out.numel() * 2 <= DMA_LIMIT
    if residual is not None:
        assert residual.shape == out.shape and residual.is_contiguous() and residual.dtype == x.dtype
    if stats is not None:
        assert cl == 128 and stats.dtype == torch.float32 and stats.is_contiguous() and stats.shape == (NB * (H // 16) * (W // 16), cout // 4, 2)
    bias = _f32(bias, "bias")
    if bias is not None:
        assert bias.numel() >= cout
    for h in range(cout // cl):
        _check(lib().mmgt_gn_silu_conv3x3(_ptr(x), _ptr(scale), _ptr(shift), wimg.data_ptr() + h * per, None if bias is None else bias.data_ptr() + 4 * cl * h,
                                          None if residual is None else residual.data_ptr() + 2 * cl * h, out.data_ptr() + 2 * cl * h,
                                          None if stats is None else stats.data_ptr() + 4 * 64 * h,
                                          NB, H, W, C, cl, cout, dtype_code(x.dtype), _stream()), "mmgt_gn_silu_conv3x3")
    return out


def gn_tables_from_stats(stats, gamma, beta, groups, eps, NB, C):
    """(scale, shift) of GroupNorm(groups, gamma, beta, eps) over a tensor (NB, H, W, C) from the per-tile partial sums `gn_silu_conv3x3*` wrote
    while storing it (stats (NB * tiles, C / 4, 2)): one small launch instead of the statistics pass over the tensor."""
    _dev(stats, gamma, beta)
    assert stats.dim() == 3 and stats.shape[1] == C // 4 and stats.shape[2] == 2 and stats.shape[0] % NB == 0 and stats.is_contiguous()
    tab = torch.empty((2, NB, C), device=stats.device, dtype=torch.float32)
    _check(lib().mmgt_gn_stats_finalize(_ptr(stats), _ptr(_f32(gamma, "gamma")), _ptr(_f32(beta, "beta")), _ptr(tab[0]), _ptr(tab[1]), NB, stats.shape[0] // NB,
                                        C, groups, eps, _stream()), "mmgt_gn_stats_finalize")
    return tab[0], tab[1]


def gn_silu_conv3x3(x, gamma, beta, groups, eps, wimg, cout, bias=None, residual=None, out=None, tables=None, want_stats=False):
    """conv3x3(silu(GroupNorm(x))) (+ residual): the statistics from `tables` (scale, shift) if the caller has them (`gn_tables_from_stats` of the
    launch that produced x), else ONE pass over x (`groupnorm_affine`); then one fused launch per 128 output channels.  x (NB, H, W, Cin) bf16
    channels-last -> (NB, H, W, cout), or (out, stats) with want_stats (stats = None where the launch cannot produce them)."""
    assert x.dim() == 4 and x.is_contiguous()
    NB, H, W, C = x.shape
    if out is None:
        out = torch.empty((NB, H, W, cout), device=x.device, dtype=x.dtype)
    stats = None
    if want_stats and cout % 128 == 0:
        stats = torch.empty((NB * (H // 16) * (W // 16), cout // 4, 2), device=x.device, dtype=torch.float32)
    step = max(1, DMA_LIMIT // (H * W * max(C, cout) * x.element_size()))       # 32-bit buffer offsets: runs of whole images below 2 GiB
    tiles = (H // 16) * (W // 16)
    for n0 in range(0, NB, step):
        n1 = min(NB, n0 + step)
        if tables is None:
            scale, shift = groupnorm_affine(x[n0:n1].view(n1 - n0, H * W, C), gamma, beta, groups, eps)
        else:
            scale, shift = tables[0][n0:n1], tables[1][n0:n1]
        gn_silu_conv3x3_tables(x[n0:n1], scale, shift, wimg, cout, bias, None if residual is None else residual[n0:n1], out[n0:n1],
                               None if stats is None else stats[n0 * tiles:n1 * tiles])
    return (out, stats) if want_stats else out


def layernorm(x, gamma, beta, eps=1e-5, pe=None, pe_div=1, pe_mod=1, out=None):
    """x (rows, C) -> LayerNorm over C (+ pe[(row // pe_div) % pe_mod] added after the affine).  With pe=None and
    pe_mod > 1, `beta` is a (>= pe_mod, C) table of beta + pe rows indexed the same way."""
    _dev(x, gamma, beta, pe)
    assert x.dim() == 2 and x.stride(1) == 1
    rows, C = x.shape
    if pe is None and pe_mod > 1:
        assert beta.dim() == 2 and beta.shape[0] >= pe_mod and beta.shape[1] == C and beta.is_contiguous()
    else:
        assert beta.numel() == C
    if out is None:
        out = torch.empty((rows, C), device=x.device, dtype=x.dtype)
    _check(lib().mmgt_layernorm(_ptr(x), x.stride(0), _ptr(_f32(gamma, "gamma")), _ptr(_f32(beta, "beta")), eps,
                                _ptr(_f32(pe, "pe")), pe_div, pe_mod, _ptr(out), out.stride(0), rows, C,
                                dtype_code(x.dtype), _stream()), "mmgt_layernorm")
    return out


def channel_norm_gelu(x, gamma, beta, eps=1e-5):
    """(rows, C): per-channel normalisation over the rows + affine + GELU (wav2vec2's first conv layer)."""
    _dev(x, gamma, beta)
    assert x.dim() == 2 and x.is_contiguous()
    out = torch.empty_like(x)
    _check(lib().mmgt_channel_norm_gelu(_ptr(x), _ptr(_f32(gamma, "gamma")), _ptr(_f32(beta, "beta")), _ptr(out), x.shape[0], x.shape[1], eps,
                                        dtype_code(x.dtype), _stream()), "mmgt_channel_norm_gelu")
    return out


def lerp_rows(x, rows_out):
    """(rows_in, C) -> (rows_out, C): linear interpolation along the rows, align_corners=True."""
    _dev(x)
    assert x.dim() == 2 and x.is_contiguous()
    out = torch.empty((rows_out, x.shape[1]), device=x.device, dtype=x.dtype)
    _check(lib().mmgt_lerp_rows(_ptr(x), _ptr(out), x.shape[0], rows_out, x.shape[1], dtype_code(x.dtype), _stream()), "mmgt_lerp_rows")
    return out


def ff_fused_supported(C, inner, dtype):
    return dtype == torch.bfloat16 and lib().mmgt_ff_fused_image_bytes(C, inner) > 0


def ff_fused(x, ln_gamma, ln_beta, wimg, bias2, residual, inner, eps=1e-5, out=None):
    """out = residual + bias2 + W2 . GEGLU(W1 . LN(x) + b1) in one launch (x (M, 320) bf16; wimg = packing.pack_ff_fused(...);
    ln_gamma None: x is used as it is)."""
    _dev(x, ln_gamma, ln_beta, wimg, bias2, residual)
    assert x.dim() == 2 and x.stride(1) == 1 and residual.shape == x.shape and residual.stride(1) == 1 and residual.dtype == x.dtype
    M, C = x.shape
    assert wimg.dtype == torch.uint8 and wimg.is_contiguous() and wimg.numel() == lib().mmgt_ff_fused_image_bytes(C, inner)
    if out is None:
        out = torch.empty((M, C), device=x.device, dtype=x.dtype)
    assert out.shape == (M, C) and out.stride(1) == 1 and out.dtype == x.dtype
    _check(lib().mmgt_ff_fused(_ptr(x), x.stride(0), _ptr(_f32(ln_gamma, "ln_gamma")), _ptr(_f32(ln_beta, "ln_beta")), eps, _ptr(wimg),
                               _ptr(_f32(bias2, "bias2")), _ptr(residual), residual.stride(0), _ptr(out), out.stride(0), M, C, inner,
                               dtype_code(x.dtype), _stream()), "mmgt_ff_fused")
    return out


def ff_fused_po(x, ln_gamma, ln_beta, wimg, bias2, residual, inner, wpo, bias_po, residual2, eps=1e-5, out=None):
    """`ff_fused` with the transformer block's proj_out on the end of the same launch: out = residual2 + bias_po + Wpo . hidden,
    hidden = bf16(residual + bias2 + FeedForward(LN(x))) never stored (wpo = packing.pack_ff_proj_out(proj_out.weight))."""
    _dev(x, ln_gamma, ln_beta, wimg, bias2, residual, wpo, bias_po, residual2)
    assert x.dim() == 2 and x.stride(1) == 1 and residual.shape == x.shape and residual.stride(1) == 1 and residual.dtype == x.dtype
    assert residual2.shape == x.shape and residual2.stride(1) == 1 and residual2.dtype == x.dtype
    M, C = x.shape
    assert wimg.dtype == torch.uint8 and wimg.is_contiguous() and wimg.numel() == lib().mmgt_ff_fused_image_bytes(C, inner)
    assert wpo.dtype == torch.uint8 and wpo.is_contiguous() and wpo.numel() == C * C * 2
    if out is None:
        out = torch.empty((M, C), device=x.device, dtype=x.dtype)
    assert out.shape == (M, C) and out.stride(1) == 1 and out.dtype == x.dtype
    _check(lib().mmgt_ff_fused_po(_ptr(x), x.stride(0), _ptr(_f32(ln_gamma, "ln_gamma")), _ptr(_f32(ln_beta, "ln_beta")), eps, _ptr(wimg),
                                  _ptr(_f32(bias2, "bias2")), _ptr(residual), residual.stride(0), _ptr(wpo), _ptr(_f32(bias_po, "bias_po")),
                                  _ptr(residual2), residual2.stride(0), _ptr(out), out.stride(0), M, C, inner, dtype_code(x.dtype), _stream()),
           "mmgt_ff_fused_po")
    return out


def rowgemm320_supported(dtype, K, N):
    return dtype == torch.bfloat16 and K == 320 and lib().mmgt_rowgemm320_image_bytes(N) > 0


def rowgemm320(x, wimg, N, bias=None, *, ln_gamma=None, ln_beta=None, pe_div=0, pe_mod=0, eps=1e-5, residual=None, bias2=None,
               bias2_rows=0, n1=None, out=None, out_t=None, n_tok=0, pre_scale=None, pre_shift=None, pre_rows=0, pre_silu=False):
    """[LayerNorm ->] Linear of the 320-channel level in one launch that reads x once (csrc/rowgemm.hip).  x (M, 320) bf16, wimg =
    packing.pack_rowgemm(W (N, 320)).  Columns [0, n1) -> out (M, n1) row-major (+ residual, n1 == N only), columns [n1, N) ->
    out_t (M / n_tok, N - n1, npad) transposed per batch of n_tok rows (the V^T operand of `attention(v_transposed=True)`).
    ln_beta (pe_mod, 320) fp32: row (m / pe_div) % pe_mod.  pre_scale / pre_shift (M / pre_rows, 320) fp32 instead of a LayerNorm:
    x * scale[m / pre_rows] + shift[m / pre_rows] (`groupnorm_affine`; pre_silu: SiLU behind it).  Returns (out, out_t)."""
    _dev(x, wimg, bias, ln_gamma, ln_beta, residual, bias2, out, out_t, pre_scale, pre_shift)
    norm = 1 if ln_gamma is not None else 0
    if pre_scale is not None:
        assert ln_gamma is None and pre_rows > 0 and x.shape[0] % pre_rows == 0
        assert pre_scale.shape == pre_shift.shape == (x.shape[0] // pre_rows, 320) and pre_scale.is_contiguous() and pre_shift.is_contiguous()
        norm, ln_gamma, ln_beta, pe_div, pe_mod = (3 if pre_silu else 2), pre_scale, pre_shift, pre_rows, x.shape[0] // pre_rows
    assert x.dim() == 2 and x.shape[1] == 320 and x.stride(1) == 1 and x.dtype == torch.bfloat16
    M = x.shape[0]
    n1 = N if n1 is None else n1
    assert wimg.dtype == torch.uint8 and wimg.is_contiguous() and wimg.numel() == lib().mmgt_rowgemm320_image_bytes(N)
    if n1 and out is None:
        out = torch.empty((M, n1), device=x.device, dtype=x.dtype)
    npad = 0
    if n1 < N:
        assert n_tok > 0 and M % n_tok == 0
        if out_t is None:
            out_t = torch.empty((M // n_tok, N - n1, (n_tok + 7) // 8 * 8), device=x.device, dtype=x.dtype)
        assert out_t.dtype == x.dtype and out_t.is_contiguous() and out_t.shape[:2] == (M // n_tok, N - n1)
        npad = out_t.shape[2]
    if out is not None:
        assert out.shape == (M, n1) and out.stride(1) == 1 and out.dtype == x.dtype
    if residual is not None:
        assert residual.shape == (M, N) and residual.stride(1) == 1 and residual.dtype == x.dtype
    if ln_beta is not None:
        assert ln_beta.is_contiguous() and ln_beta.numel() >= 320 * max(pe_mod, 1)
    if bias2 is not None:
        assert bias2.dim() == 2 and bias2.shape[1] == N and bias2.is_contiguous() and (M + bias2_rows - 1) // bias2_rows <= bias2.shape[0]
    _check(lib().mmgt_rowgemm320(_ptr(x), x.stride(0), norm, _ptr(_f32(ln_gamma, "ln_gamma")), _ptr(_f32(ln_beta, "ln_beta")), pe_div, pe_mod,
                                 eps, _ptr(wimg), _ptr(_f32(bias, "bias")), _ptr(_f32(bias2, "bias2")), bias2_rows, _ptr(residual),
                                 0 if residual is None else residual.stride(0), _ptr(out), 0 if out is None else out.stride(0), n1,
                                 _ptr(out_t), n_tok, npad, M, N, dtype_code(x.dtype), _stream()), "mmgt_rowgemm320")
    return out, out_t


def temporal_leg320_supported(dtype, C, heads, frames, n_pix, batch=1):
    """What csrc/tleg.hip is built for: bf16, 8 heads of 40, windows of 24 or 12 frames, whole 4-wave tasks, a tensor a buffer resource
    can address (2 GiB) -- a caller with any other shape takes the three-launch path."""
    return dtype == torch.bfloat16 and C == 320 and heads == 8 and frames in (12, 24) and n_pix % (4 * 48 // frames) == 0 and \
        batch * frames * n_pix * 640 < (1 << 31)


def temporal_leg320(x, ln_gamma, beta_pe, wimg, bias_o, batch, frames, n_pix, scale, eps=1e-5, out=None):
    """x + to_out(temporal attention(LayerNorm(x) + pe)) of a level-0 motion-module attention block in ONE launch (csrc/tleg.hip).  x (batch *
    frames * n_pix, 320) bf16 rows (batch, frame, pixel); beta_pe (>= frames, 320) fp32 = LayerNorm bias + pe rows; wimg = packing.pack_tleg."""
    _dev(x, ln_gamma, beta_pe, wimg, bias_o, out)
    if not (x.dim() == 2 and temporal_leg320_supported(x.dtype, x.shape[1], 8, frames, n_pix, batch)):
        raise RuntimeError(f"temporal_leg320: unsupported shape (dtype {x.dtype}, {tuple(x.shape)}, {frames} frames x {n_pix} pixels x batch {batch}): "
                           "8 heads of 40 channels, 12 or 24 frames, bf16, < 2 GiB")
    assert x.dim() == 2 and x.shape == (batch * frames * n_pix, 320) and x.is_contiguous() and x.dtype == torch.bfloat16
    assert wimg.dtype == torch.uint8 and wimg.is_contiguous() and wimg.numel() == lib().mmgt_temporal_leg320_image_bytes()
    assert beta_pe.dim() == 2 and beta_pe.shape[1] == 320 and beta_pe.shape[0] >= frames
    if out is None:
        out = torch.empty_like(x)
    assert out.shape == x.shape and out.is_contiguous() and out.dtype == x.dtype
    _check(lib().mmgt_temporal_leg320(_ptr(x), _ptr(out), _ptr(_f32(ln_gamma, "ln_gamma")), _ptr(_f32(beta_pe, "beta_pe")), beta_pe.shape[0],
                                      _ptr(wimg), _ptr(_f32(bias_o, "bias_o")), batch, frames, n_pix, scale, eps, dtype_code(x.dtype), _stream()),
           "mmgt_temporal_leg320")
    return out


# ------------------------------------------------------------------------------------------------------------ attention

def attention(q, k, v, out, *, batch, heads, hd, nq, nk, scale, q_str, k_str, v_str, o_str, bdiv=1, v_transposed=False,
              k2=None, v2=None, k2_str=(0, 0), v2_str=(0, 0), k2_bdiv=1, nk2=0, seg2_first_batch=0, out_scale=None, out_scale_heads=0,
              twin_out=None):
    """Raw strided attention (see include/mmgt_hip.h).  *_str = (batch stride 0, batch stride 1, token stride).
    out_scale (groups, batch * nq) fp32 + out_scale_heads: the output rows of head group g = head // out_scale_heads are multiplied by
    out_scale[g] (mmgt_attention_scaled: single key segment, row-major V)."""
    _dev(q, k, v, out, k2, v2, out_scale, twin_out)
    if twin_out is not None:
        # every batch entry reads (k2, v2); twin_out (out's strides) receives the attention over (k, v) alone (mmgt_attention_twin)
        assert k2 is not None and v_transposed and seg2_first_batch == 0 and out_scale is None and twin_out.dtype == out.dtype
        _check(lib().mmgt_attention_twin(_ptr(q), q_str[0], q_str[1], q_str[2], _ptr(k), k_str[0], k_str[1], k_str[2], _ptr(v),
                                         v_str[0], v_str[1], v_str[2], _ptr(out), _ptr(twin_out), o_str[0], o_str[1], o_str[2], bdiv,
                                         _ptr(k2), _ptr(v2), k2_str[0], k2_str[1], v2_str[0], v2_str[1], k2_bdiv, nk2, batch, heads, hd,
                                         nq, nk, scale, dtype_code(q.dtype), _stream()), "mmgt_attention_twin")
        return out
    if out_scale is not None:
        assert k2 is None and not v_transposed and bdiv == 1 and out_scale_heads > 0 and heads % out_scale_heads == 0
        assert out_scale.dtype == torch.float32 and out_scale.dim() == 2 and out_scale.stride(1) == 1
        assert out_scale.shape == (heads // out_scale_heads, batch * nq)
        _check(lib().mmgt_attention_scaled(_ptr(q), q_str[0], q_str[1], q_str[2], _ptr(k), k_str[0], k_str[1], k_str[2], _ptr(v),
                                           v_str[0], v_str[1], v_str[2], _ptr(out), o_str[0], o_str[1], o_str[2], _ptr(out_scale),
                                           out_scale.stride(0), out_scale_heads, batch, heads, hd, nq, nk, scale, dtype_code(q.dtype),
                                           _stream()), "mmgt_attention_scaled")
        return out
    _check(lib().mmgt_attention(_ptr(q), q_str[0], q_str[1], q_str[2], _ptr(k), k_str[0], k_str[1], k_str[2], _ptr(v),
                                v_str[0], v_str[1], v_str[2], _ptr(out), o_str[0], o_str[1], o_str[2], bdiv, _ptr(k2),
                                _ptr(v2), k2_str[0], k2_str[1], v2_str[0], v2_str[1], k2_bdiv, nk2, seg2_first_batch,
                                batch, heads, hd, nq, nk, scale, int(v_transposed), dtype_code(q.dtype), _stream()),
           "mmgt_attention")
    return out


def softmax_rows(x, scale=1.0, out=None):
    _dev(x)
    assert x.dim() == 2 and x.stride(1) == 1
    if out is None:
        out = torch.empty_like(x)
    _check(lib().mmgt_softmax_rows(_ptr(x), x.stride(0), _ptr(out), out.stride(0), x.shape[0], x.shape[1], scale,
                                   dtype_code(x.dtype), _stream()), "mmgt_softmax_rows")
    return out


def softmax_rows_f32_bf16(x, scale, out):
    """fp32 logits (rows, cols) -> bf16 probabilities, one pass (cols a multiple of 256, <= 8192)."""
    _dev(x, out)
    assert x.dim() == 2 and x.stride(1) == 1 and x.dtype == torch.float32
    assert out.shape == x.shape and out.stride(1) == 1 and out.dtype == torch.bfloat16
    _check(lib().mmgt_softmax_rows_f32_bf16(_ptr(x), x.stride(0), _ptr(out), out.stride(0), x.shape[0], x.shape[1], scale, _stream()),
           "mmgt_softmax_rows_f32_bf16")
    return out


def gemm_bf16_f32(a, w, out=None):
    """out fp32 (M, N) = a bf16 (M, K) @ w bf16 (N, K)^T: the raw fp32 accumulators of the bf16 MFMA path (K % 64 == 0, N % 8 == 0)."""
    _dev(a, w, out)
    assert a.dim() == 2 and w.dim() == 2 and a.shape[1] == w.shape[1] and a.stride(1) == 1 and w.stride(1) == 1
    assert a.dtype == torch.bfloat16 and w.dtype == torch.bfloat16
    M, K = a.shape
    N = w.shape[0]
    if out is None:
        out = torch.empty((M, N), device=a.device, dtype=torch.float32)
    assert out.shape == (M, N) and out.dtype == torch.float32 and out.stride(1) == 1
    _check(lib().mmgt_gemm_bf16_f32(_ptr(a), a.stride(0), _ptr(w), w.stride(0), _ptr(out), out.stride(0), M, N, K, _stream()),
           "mmgt_gemm_bf16_f32")
    return out


def qk_split3(qk, bias_q, bias_k):
    """qk fp32 (rows, 4C) = [t Wq_hi^T | t Wq_lo^T | t Wk_hi^T | t Wk_lo^T] -> (Qp, Kp) bf16 (rows, 3C): [q_hi|q_hi|q_lo], [k_hi|k_lo|k_hi]
    with q = qk0 + qk1 + bias_q, k = qk2 + qk3 + bias_k (so that Qp . Kp = q . k to ~17 bits on the bf16 MFMA path)."""
    _dev(qk, bias_q, bias_k)
    assert qk.dim() == 2 and qk.is_contiguous() and qk.dtype == torch.float32 and qk.shape[1] % 16 == 0
    rows, C = qk.shape[0], qk.shape[1] // 4
    assert bias_q.shape == (C,) and bias_k.shape == (C,) and bias_q.dtype == torch.float32 and bias_k.dtype == torch.float32
    Qp = torch.empty((rows, 3 * C), device=qk.device, dtype=torch.bfloat16)
    Kp = torch.empty((rows, 3 * C), device=qk.device, dtype=torch.bfloat16)
    _check(lib().mmgt_qk_split3(_ptr(qk), _ptr(bias_q), _ptr(bias_k), _ptr(Qp), _ptr(Kp), rows, C, _stream()), "mmgt_qk_split3")
    return Qp, Kp


# ------------------------------------------------------------------------------------------------------------ plumbing

def ncfhw_to_nhwc(x, cpad, dtype, scale=1.0):
    """(B, C, F, H, W) fp32 -> (B*F, H, W, cpad) channels-last `dtype` (times `scale`), zero padded channels."""
    _dev(x)
    assert x.dim() == 5 and x.dtype == torch.float32 and x.is_contiguous()
    B, C, F, H, W = x.shape
    out = torch.empty((B * F, H, W, cpad), device=x.device, dtype=dtype)
    _check(lib().mmgt_ncfhw_to_nhwc(_ptr(x), _ptr(out), B, C, F, H, W, cpad, scale, dtype_code(dtype), _stream()),
           "mmgt_ncfhw_to_nhwc")
    return out


def nhwc_to_ncfhw(x, B, C, scale=1.0, shift=0.0, clamp01=False):
    """(B*F, H, W, cpad) -> (B, C, F, H, W) fp32 (first C channels), out = clamp(x * scale + shift)."""
    _dev(x)
    assert x.dim() == 4 and x.is_contiguous()
    BF, H, W, cpad = x.shape
    F = BF // B
    out = torch.empty((B, C, F, H, W), device=x.device, dtype=torch.float32)
    _check(lib().mmgt_nhwc_to_ncfhw(_ptr(x), _ptr(out), B, C, F, H, W, cpad, scale, shift, int(clamp01),
                                    dtype_code(x.dtype), _stream()),
           "mmgt_nhwc_to_ncfhw")
    return out


def conv1x1_cat_supported(dtype, c0, c1, cout):
    return dtype == torch.bfloat16 and c0 % 64 == 0 and c1 % 64 == 0 and c0 > 0 and c1 > 0 and (cout % 256 == 0 or cout % 320 == 0)


def conv1x1_cat(x0, x1, w, bias=None, residual=None, out=None):
    """bias + [x0 | x1] . w^T (+ residual): x0 (rows, C0), x1 (rows, C1) bf16, w (Cout, C0 + C1) -> (rows, Cout), one launch (csrc/gemm16.hip's conv
    gather over two sources with one tap): a resnet's conv_shortcut over the skip concatenation."""
    _dev(x0, x1, w, bias, residual, out)
    assert x0.dim() == 2 and x1.dim() == 2 and x0.shape[0] == x1.shape[0] and x0.is_contiguous() and x1.is_contiguous() and w.is_contiguous()
    rows, c0, c1, cout = x0.shape[0], x0.shape[1], x1.shape[1], w.shape[0]
    assert conv1x1_cat_supported(x0.dtype, c0, c1, cout) and w.shape == (cout, c0 + c1) and w.dtype == x0.dtype == x1.dtype
    if out is None:
        out = torch.empty((rows, cout), device=x0.device, dtype=x0.dtype)
    assert out.shape == (rows, cout) and out.is_contiguous()
    if residual is not None:
        assert residual.shape == out.shape and residual.is_contiguous() and residual.dtype == x0.dtype
    _check(lib().mmgt_conv1x1_cat_nhwc(_ptr(x0), c0, _ptr(x1), c1, rows, _ptr(w), _ptr(_f32(bias, "bias")), _ptr(residual), _ptr(out), cout,
                                       dtype_code(x0.dtype), _stream()), "mmgt_conv1x1_cat_nhwc")
    return out


def conv_taps_gather(y, bias, NB, H, W):
    """y (NB * H * W, ldY >= 36) bf16 = the per-pixel products of a 4-channel 3 x 3 conv (column 4 tap + o: packing.pack_conv_taps) -> (NB, H, W, 8) bf16:
    channels 0 .. 3 = bias + the sum over the taps of the neighbour pixels' products, 4 .. 7 zero."""
    _dev(y, bias)
    assert y.dim() == 2 and y.shape[0] == NB * H * W and y.shape[1] >= 36 and y.stride(1) == 1 and y.dtype == torch.bfloat16
    out = torch.empty((NB, H, W, 8), device=y.device, dtype=y.dtype)
    _check(lib().mmgt_conv_taps_gather(_ptr(y), y.stride(0), _ptr(_f32(bias, "bias")), _ptr(out), NB, H, W, dtype_code(y.dtype), _stream()),
           "mmgt_conv_taps_gather")
    return out


def timestep_features(timesteps, dim, dtype):
    _dev(timesteps)
    assert timesteps.dtype == torch.float32 and timesteps.dim() == 1
    out = torch.empty((timesteps.shape[0], dim), device=timesteps.device, dtype=dtype)
    _check(lib().mmgt_timestep_features(_ptr(timesteps), _ptr(out), timesteps.shape[0], dim, dtype_code(dtype),
                                        _stream()), "mmgt_timestep_features")
    return out


def silu(x):
    _dev(x)
    assert x.is_contiguous()
    out = torch.empty_like(x)
    _check(lib().mmgt_silu(_ptr(x), _ptr(out), x.numel(), dtype_code(x.dtype), _stream()), "mmgt_silu")
    return out


def cfg_ddim_step(pred_sum, counter, latents, guidance, sa_t, sb_t, sa_p, sb_p):
    """latents (1, C, F, H, W) fp32; pred_sum (2, C, F, H, W) fp32; counter (F,) fp32 -> new latents."""
    _dev(pred_sum, counter, latents)
    assert latents.dtype == torch.float32 and latents.is_contiguous() and pred_sum.is_contiguous()
    _, C, F, H, W = latents.shape
    out = torch.empty_like(latents)
    _check(lib().mmgt_cfg_ddim_step(_ptr(pred_sum), _ptr(_f32(counter, "counter")), _ptr(latents), _ptr(out),
                                    latents.numel(), F, H * W, guidance, sa_t, sb_t, sa_p, sb_p, _stream()),
           "mmgt_cfg_ddim_step")
    return out


def accumulate_window(pred, pred_sum, counter, idx, C, rows=2, row0=0, bump_counter=True):
    """pred ((rows*Fw), H, W, cpad) channels-last; pred_sum (2, C, F, H, W) fp32: CFG rows [row0, row0 + rows) += pred at
    frames idx (int32 device); counter[idx] += 1 when bump_counter."""
    _dev(pred, pred_sum, counter, idx)
    assert idx.dtype == torch.int32 and pred.is_contiguous() and pred_sum.is_contiguous()
    Fw = idx.numel()
    F = pred_sum.shape[2]
    hw = pred.shape[1] * pred.shape[2]
    assert pred.shape[0] == rows * Fw and pred_sum.shape[0] == 2 and pred_sum.shape[3] * pred_sum.shape[4] == hw
    _check(lib().mmgt_accumulate_window_rows(_ptr(pred), _ptr(pred_sum), _ptr(counter), _ptr(idx), Fw, F, C, pred.shape[3],
                                             hw, rows, row0, int(bump_counter), dtype_code(pred.dtype), _stream()),
           "mmgt_accumulate_window_rows")


# ------------------------------------------------------------------------------------------------------------ SMGA glue

def rotary(x, cos_sin, seq):
    """x (rows, dim) -> rotated pairs, angle table cos_sin (>= seq, dim/2, 2) fp32 indexed by row % seq."""
    _dev(x, cos_sin)
    assert x.dim() == 2 and x.is_contiguous() and cos_sin.is_contiguous() and cos_sin.shape[0] >= seq
    out = torch.empty_like(x)
    _check(lib().mmgt_rotary(_ptr(x), _ptr(_f32(cos_sin, "cos_sin")), _ptr(out), x.shape[0], x.shape[1], seq, dtype_code(x.dtype),
                             _stream()), "mmgt_rotary")
    return out


def film_residual(x, scale_shift, rows_per_batch, res=None, res2=None):
    """res (+ res2) + (scale + 1) * x + shift; scale_shift: fp32 (B, >= 2 dim) view with unit column stride, [scale | shift] per row."""
    _dev(x, scale_shift, res, res2)
    assert x.dim() == 2 and x.is_contiguous() and scale_shift.dtype == torch.float32 and scale_shift.stride(1) == 1
    assert scale_shift.shape[1] >= 2 * x.shape[1] and scale_shift.shape[0] * rows_per_batch >= x.shape[0]
    for r in (res, res2):
        assert r is None or (r.shape == x.shape and r.is_contiguous() and r.dtype == x.dtype)
    out = torch.empty_like(x)
    _check(lib().mmgt_film_residual(_ptr(x), _ptr(scale_shift), scale_shift.stride(0), _ptr(res), _ptr(res2), _ptr(out), x.shape[0],
                                    x.shape[1], rows_per_batch, dtype_code(x.dtype), _stream()), "mmgt_film_residual")
    return out


def mean_tokens(x):
    """(B, T, C) -> fp32 (B, C)."""
    _dev(x)
    assert x.dim() == 3 and x.is_contiguous()
    out = torch.empty((x.shape[0], x.shape[2]), device=x.device, dtype=torch.float32)
    _check(lib().mmgt_mean_tokens(_ptr(x), _ptr(out), x.shape[0], x.shape[1], x.shape[2], dtype_code(x.dtype), _stream()),
           "mmgt_mean_tokens")
    return out


def activation(x, act):
    _dev(x)
    assert x.is_contiguous()
    out = torch.empty_like(x)
    _check(lib().mmgt_activation(_ptr(x), _ptr(out), x.numel(), act, dtype_code(x.dtype), _stream()), "mmgt_activation")
    return out


def smga_ddim_step(pred_uncond, pred_cond, x, noise, guidance, sqrt_recip, sqrt_recipm1, sqrt_next, c, sigma, last):
    _dev(pred_uncond, pred_cond, x, noise)
    assert pred_uncond.is_contiguous() and pred_cond.is_contiguous() and pred_uncond.dtype == pred_cond.dtype
    assert x.dtype == torch.float32 and x.is_contiguous() and pred_uncond.numel() == x.numel() == pred_cond.numel()
    out = torch.empty_like(x)
    _check(lib().mmgt_smga_ddim_step(_ptr(pred_uncond), _ptr(pred_cond), _ptr(x), _ptr(_f32(noise, "noise")), _ptr(out), x.numel(),
                                     guidance, sqrt_recip, sqrt_recipm1, sqrt_next, c, sigma, int(last),
                                     dtype_code(pred_uncond.dtype), _stream()), "mmgt_smga_ddim_step")
    return out


# ------------------------------------------------------------------------------------------------------------ conditioning / output

def blur_mask_u8(masks, ksize):
    """(L, H, W) uint8 -> (L, 64, 64) uint8 (see mmgt_blur_mask_u8)."""
    _dev(masks)
    assert masks.dtype == torch.uint8 and masks.dim() == 3 and masks.is_contiguous()
    out = torch.empty((masks.shape[0], 64, 64), device=masks.device, dtype=torch.uint8)
    _check(lib().mmgt_blur_mask_u8(_ptr(masks), _ptr(out), masks.shape[0], masks.shape[1], masks.shape[2], ksize, _stream()),
           "mmgt_blur_mask_u8")
    return out


def resample_u8(x, D, bounds, coeffs, as_float=True):
    """(L, S, S) uint8 -> (L, D, D) float32 in [0, 1] (or uint8) with PIL's integer coefficient tables (int32 device tensors)."""
    _dev(x, bounds, coeffs)
    assert x.dtype == torch.uint8 and x.dim() == 3 and x.shape[1] == x.shape[2] and x.is_contiguous()
    assert bounds.dtype == torch.int32 and coeffs.dtype == torch.int32 and bounds.shape == (D, 2) and coeffs.shape[0] == D
    out = torch.empty((x.shape[0], D, D), device=x.device, dtype=torch.float32 if as_float else torch.uint8)
    _check(lib().mmgt_resample_u8(_ptr(x), _ptr(out) if as_float else None, None if as_float else _ptr(out), x.shape[0], x.shape[1], D,
                                  _ptr(bounds), _ptr(coeffs), coeffs.shape[1], _stream()), "mmgt_resample_u8")
    return out


def window_stack(x, half=2):
    """(L, ...) float32 -> (L, 2 half + 1, ...)."""
    _dev(x)
    assert x.dtype == torch.float32 and x.is_contiguous()
    L = x.shape[0]
    D = x.numel() // L
    out = torch.empty((L, 2 * half + 1) + tuple(x.shape[1:]), device=x.device, dtype=torch.float32)
    _check(lib().mmgt_window_stack(_ptr(x), _ptr(out), L, D, half, _stream()), "mmgt_window_stack")
    return out


def frames_to_u8(x, scale=0.5, shift=0.5):
    """channels-last (N, H, W, cpad) -> (N, H, W, 3) uint8 = trunc(clamp(x * scale + shift, 0, 1) * 255)."""
    _dev(x)
    assert x.dim() == 4 and x.is_contiguous()
    out = torch.empty(tuple(x.shape[:3]) + (3,), device=x.device, dtype=torch.uint8)
    _check(lib().mmgt_frames_to_u8(_ptr(x), _ptr(out), x.shape[0] * x.shape[1] * x.shape[2], x.shape[3], scale, shift,
                                   dtype_code(x.dtype), _stream()), "mmgt_frames_to_u8")
    return out



def dwpose_draw(kp, H=512, W=512):
    """SMGA's normalised key points (L, 402) or (L, 134, 3) fp32 -> (pose (L, H, W, 3), hands, lips, face (L, H, W)) uint8: the reference's
    pose_vid_generator frames drawn on the device (see mmgt_dwpose_draw)."""
    _dev(kp)
    kp = kp.reshape(kp.shape[0], 134, 3)
    assert kp.dtype == torch.float32 and kp.is_contiguous()
    L = kp.shape[0]
    pose = torch.empty((L, H, W, 3), device=kp.device, dtype=torch.uint8)
    hands, lips, face = (torch.empty((L, H, W), device=kp.device, dtype=torch.uint8) for _ in range(3))
    _check(lib().mmgt_dwpose_draw(_ptr(kp), _ptr(pose), _ptr(hands), _ptr(lips), _ptr(face), L, H, W, _stream()), "mmgt_dwpose_draw")
    return pose, hands, lips, face
