// Element-wise pieces of the wav2vec2 feature extractor (src/dataset/audio_processor.py:76-131 through
// src/models/wav2vec.py:42-127 = transformers Wav2Vec2Model) that are not GEMM / LayerNorm / attention (gfx950):
//
//  * mmgt_channel_norm_gelu   layer 0 of the conv feature extractor: GroupNorm(num_groups = C, C channels) over TIME -- every channel
//                             normalised on its own over the T frames -- affine, exact GELU ("feat_extract_norm": "group",
//                             Wav2Vec2GroupNormConvLayer), on the channels-last (T, C) tensor the conv GEMM wrote.
//  * mmgt_lerp_rows           `linear_interpolation` (src/models/wav2vec.py:196-209): F.interpolate(mode="linear",
//                             align_corners=True) of the (T, C) conv features to seq_len frames.
// Both are micro-second kernels on (3071 x 512) / (24 x 512) tensors; both storage types (the fp32 instantiation is the parity mode).
#include "common.h"
#include "mmgt_hip.h"

namespace {

template <typename T>
struct V16 {
  static constexpr int VEC = 16 / sizeof(T);
  static __device__ __forceinline__ void load(const T* p, float* f) {
    union { u32x4 u; T e[VEC]; } v;
    v.u = *reinterpret_cast<const u32x4*>(p);
#pragma unroll
    for (int i = 0; i < VEC; ++i) f[i] = Elem<T>::ld(&v.e[i]);
  }
  static __device__ __forceinline__ void store(T* p, const float* f) {
    union { u32x4 u; T e[VEC]; } v;
#pragma unroll
    for (int i = 0; i < VEC; ++i) Elem<T>::st(&v.e[i], f[i]);
    *reinterpret_cast<u32x4*>(p) = v.u;
  }
};

// one workgroup per VEC channels; three passes over the T rows (mean, variance about the mean, normalise): exact two-pass
// statistics with fixed-order reductions (bitwise reproducible)
template <typename T>
__global__ __launch_bounds__(256) void channel_norm_gelu_kernel(const T* __restrict__ x, const float* __restrict__ gamma,
                                                                const float* __restrict__ beta, T* __restrict__ out, int rows, int C,
                                                                float eps) {
  constexpr int VEC = V16<T>::VEC;
  __shared__ float red[256][VEC];
  const int c0 = blockIdx.x * VEC, tid = threadIdx.x;
  auto total = [&](const float (&acc)[VEC], float (&tot)[VEC]) {
#pragma unroll
    for (int e = 0; e < VEC; ++e) red[tid][e] = acc[e];
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
      if (tid < s)
#pragma unroll
        for (int e = 0; e < VEC; ++e) red[tid][e] += red[tid + s][e];
      __syncthreads();
    }
#pragma unroll
    for (int e = 0; e < VEC; ++e) tot[e] = red[0][e];
    __syncthreads();
  };
  float acc[VEC], mean[VEC], rstd[VEC], f[VEC];
#pragma unroll
  for (int e = 0; e < VEC; ++e) acc[e] = 0.f;
  for (int r = tid; r < rows; r += 256) {
    V16<T>::load(x + (long)r * C + c0, f);
#pragma unroll
    for (int e = 0; e < VEC; ++e) acc[e] += f[e];
  }
  total(acc, mean);
#pragma unroll
  for (int e = 0; e < VEC; ++e) { mean[e] /= (float)rows; acc[e] = 0.f; }
  for (int r = tid; r < rows; r += 256) {
    V16<T>::load(x + (long)r * C + c0, f);
#pragma unroll
    for (int e = 0; e < VEC; ++e) { const float d = f[e] - mean[e]; acc[e] += d * d; }
  }
  total(acc, rstd);
  float g[VEC], b[VEC];
#pragma unroll
  for (int e = 0; e < VEC; ++e) { rstd[e] = rsqrtf(rstd[e] / (float)rows + eps); g[e] = gamma[c0 + e]; b[e] = beta[c0 + e]; }
  for (int r = tid; r < rows; r += 256) {
    V16<T>::load(x + (long)r * C + c0, f);
#pragma unroll
    for (int e = 0; e < VEC; ++e) f[e] = gelu_erf_f((f[e] - mean[e]) * rstd[e] * g[e] + b[e]);
    V16<T>::store(out + (long)r * C + c0, f);
  }
}

// out[i] = w0 x[lo] + w1 x[hi],  src = i (rows_in - 1) / (rows_out - 1),  lo = floor(src),  w1 = src - lo  (area_pixel_compute_scale
// with align_corners = True, as ATen's upsample_linear1d)
template <typename T>
__global__ __launch_bounds__(256) void lerp_rows_kernel(const T* __restrict__ x, T* __restrict__ out, int rows_in, int rows_out, int C) {
  constexpr int VEC = V16<T>::VEC;
  const int nv = C / VEC;
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (long)rows_out * nv) return;
  const int i = (int)(idx / nv), c = (int)(idx - (long)i * nv) * VEC;
  const float scale = rows_out > 1 ? (float)(rows_in - 1) / (float)(rows_out - 1) : 0.f;
  const float src = scale * (float)i;
  int lo = (int)src;
  if (lo > rows_in - 1) lo = rows_in - 1;
  const int hi = lo + (lo < rows_in - 1 ? 1 : 0);
  const float w1 = src - (float)lo, w0 = 1.f - w1;
  float a[VEC], b[VEC];
  V16<T>::load(x + (long)lo * C + c, a);
  V16<T>::load(x + (long)hi * C + c, b);
#pragma unroll
  for (int e = 0; e < VEC; ++e) a[e] = w0 * a[e] + w1 * b[e];
  V16<T>::store(out + (long)i * C + c, a);
}

}  // namespace

extern "C" int mmgt_channel_norm_gelu(const void* x, const float* gamma, const float* beta, void* out, int rows, int C, float eps,
                                      int dtype, void* stream) {
  MMGT_CHECK(x && gamma && beta && out && rows > 0 && C > 0, "channel_norm_gelu: bad arguments");
  MMGT_CHECK(dtype == MMGT_F32 || dtype == MMGT_BF16, "channel_norm_gelu: bad dtype %d", dtype);
  const int vec = dtype == MMGT_BF16 ? 8 : 4;
  MMGT_CHECK(C % vec == 0 && (((uintptr_t)x | (uintptr_t)out) & 15) == 0, "channel_norm_gelu: C %% %d != 0 or unaligned tensors", vec);
  hipStream_t s = (hipStream_t)stream;
  if (dtype == MMGT_BF16)
    hipLaunchKernelGGL(channel_norm_gelu_kernel<bf16_t>, dim3(C / 8), dim3(256), 0, s, (const bf16_t*)x, gamma, beta, (bf16_t*)out, rows, C, eps);
  else
    hipLaunchKernelGGL(channel_norm_gelu_kernel<float>, dim3(C / 4), dim3(256), 0, s, (const float*)x, gamma, beta, (float*)out, rows, C, eps);
  MMGT_LAUNCH_CHECK();
  return 0;
}

extern "C" int mmgt_lerp_rows(const void* x, void* out, int rows_in, int rows_out, int C, int dtype, void* stream) {
  MMGT_CHECK(x && out && rows_in > 0 && rows_out > 0 && C > 0, "lerp_rows: bad arguments");
  MMGT_CHECK(dtype == MMGT_F32 || dtype == MMGT_BF16, "lerp_rows: bad dtype %d", dtype);
  const int vec = dtype == MMGT_BF16 ? 8 : 4;
  MMGT_CHECK(C % vec == 0 && (((uintptr_t)x | (uintptr_t)out) & 15) == 0, "lerp_rows: C %% %d != 0 or unaligned tensors", vec);
  const long n = (long)rows_out * (C / vec);
  hipStream_t s = (hipStream_t)stream;
  if (dtype == MMGT_BF16)
    hipLaunchKernelGGL(lerp_rows_kernel<bf16_t>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, (const bf16_t*)x, (bf16_t*)out, rows_in, rows_out, C);
  else
    hipLaunchKernelGGL(lerp_rows_kernel<float>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, (const float*)x, (float*)out, rows_in, rows_out, C);
  MMGT_LAUNCH_CHECK();
  return 0;
}
