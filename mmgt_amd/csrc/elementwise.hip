// Layout conversion, timestep features, CFG + DDIM update, window accumulation, row softmax (gfx950).  All HBM-bound
// grid-stride kernels.
#include <stdarg.h>
#include <stdio.h>

#include "common.h"
#include "mmgt_hip.h"

// ---------------------------------------------------------------------------------------------- error plumbing
static thread_local char g_err[512] = "";
extern "C" void mmgt_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
extern "C" const char* mmgt_last_error(void) { return g_err; }
extern "C" int mmgt_abi_version(void) { return 1; }

namespace {

inline int grid_for(long n, int block = 256) {
  long g = (n + block - 1) / block;
  return (int)(g < 1 ? 1 : (g > 2048 * 4 ? 2048 * 4 : g));
}

// One thread per 8 consecutive output channels of one pixel (one or two 16-byte stores, one index decomposition per 8
// elements in 32-bit arithmetic; the element-per-thread form with 64-bit div/mod took 300 us for the 25 MB latent tensor).
template <typename T>
__global__ void ncfhw_to_nhwc_kernel(const float* __restrict__ in, T* __restrict__ out, int B, int C, int F, int HW,
                                     int Cpad, float scale) {
  const unsigned cv_n = (unsigned)Cpad / 8u;
  const unsigned nvec = (unsigned)B * (unsigned)F * (unsigned)HW * cv_n;      // host checks < 2^31
  for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += gridDim.x * blockDim.x) {
    const unsigned cv = i % cv_n;
    unsigned t = i / cv_n;
    const unsigned p = t % (unsigned)HW;
    t /= (unsigned)HW;
    const unsigned f = t % (unsigned)F, b = t / (unsigned)F;
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const unsigned c = cv * 8 + e;
      v[e] = c < (unsigned)C ? in[(((long)b * C + c) * F + f) * HW + p] * scale : 0.f;
    }
    T* dst = out + (long)i * 8;
    if (sizeof(T) == 2) {
      *reinterpret_cast<u32x4*>(dst) = (u32x4){pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]), pack_bf16x2(v[4], v[5]),
                                               pack_bf16x2(v[6], v[7])};
    } else {
      reinterpret_cast<f32x4*>(dst)[0] = (f32x4){v[0], v[1], v[2], v[3]};
      reinterpret_cast<f32x4*>(dst)[1] = (f32x4){v[4], v[5], v[6], v[7]};
    }
  }
}

template <typename T>
__global__ void nhwc_to_ncfhw_kernel(const T* __restrict__ in, float* __restrict__ out, int B, int C, int F, int HW,
                                     int Cpad, float scale, float shift, int clamp01) {
  const long total = (long)B * C * F * HW;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int p = i % HW;
    long t = i / HW;
    const int f = t % F;
    t /= F;
    const int c = t % C;
    const int b = t / C;
    float v = Elem<T>::ld(in + (((long)b * F + f) * HW + p) * Cpad + c) * scale + shift;
    if (clamp01) v = fminf(fmaxf(v, 0.f), 1.f);
    out[i] = v;
  }
}

template <typename T>
__global__ void timestep_kernel(const float* __restrict__ ts, T* __restrict__ out, int B, int dim) {
  const int half = dim / 2;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < B * dim; i += gridDim.x * blockDim.x) {
    const int b = i / dim, j = i % dim;
    const int k = j < half ? j : j - half;
    const float freq = expf(-9.210340371976184f * (float)k / (float)half);  // ln(10000)
    const float a = ts[b] * freq;
    Elem<T>::st(out + i, j < half ? cosf(a) : sinf(a));
  }
}

template <typename T>
__global__ void silu_kernel(const T* __restrict__ x, T* __restrict__ out, long n) {
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
    Elem<T>::st(out + i, silu_f(Elem<T>::ld(x + i)));
}

__global__ void cfg_ddim_kernel(const float* __restrict__ ps, const float* __restrict__ counter,
                                const float* __restrict__ x, float* __restrict__ xo, long n, int F, int hw, float g,
                                float sa_t, float sb_t, float sa_p, float sb_p) {
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const int f = (int)((i / hw) % F);
    const float cnt = counter[f];
    const float eu = ps[i] / cnt, ec = ps[n + i] / cnt;
    const float v = eu + g * (ec - eu);
    const float xi = x[i];
    const float x0 = sa_t * xi - sb_t * v;
    const float e = sa_t * v + sb_t * xi;
    xo[i] = sa_p * x0 + sb_p * e;
  }
}

template <typename T>
__global__ void accumulate_window_kernel(const T* __restrict__ pred, float* __restrict__ ps, float* __restrict__ counter,
                                         const int* __restrict__ idx, int Fw, int F, int C, int Cpad, int hw, int rows,
                                         int row0, int bump) {
  const long total = (long)rows * C * Fw * hw;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int p = i % hw;
    long t = i / hw;
    const int j = t % Fw;
    t /= Fw;
    const int c = t % C;
    const int b = t / C;
    const float v = Elem<T>::ld(pred + (((long)b * Fw + j) * hw + p) * Cpad + c);
    ps[(((long)(row0 + b) * C + c) * F + idx[j]) * hw + p] += v;  // window indices are distinct: no two threads share a target
  }
  if (bump && blockIdx.x == 0 && threadIdx.x < Fw) counter[idx[threadIdx.x]] += 1.f;
}

template <typename T>
__global__ __launch_bounds__(256) void softmax_rows_kernel(const T* __restrict__ x, long ldx, T* __restrict__ out,
                                                           long ldo, int rows, int cols, float scale) {
  // one wave per row; three passes over a row that stays in L2 (VAE mid-block attention only)
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const long row = (long)blockIdx.x * 4 + wid;
  if (row >= rows) return;
  const T* xr = x + row * ldx;
  float m = -1e30f;
  for (int c = lane; c < cols; c += 64) m = fmaxf(m, Elem<T>::ld(xr + c) * scale);
  m = wave_max(m);
  float s = 0.f;
  for (int c = lane; c < cols; c += 64) s += __expf(Elem<T>::ld(xr + c) * scale - m);
  s = wave_sum(s);
  const float inv = 1.f / s;
  for (int c = lane; c < cols; c += 64) Elem<T>::st(out + row * ldo + c, __expf(Elem<T>::ld(xr + c) * scale - m) * inv);
}

// fp32 logits -> bf16 probabilities, one wave per row held in registers (cols = 256 * NV): one read, one write (the generic kernel above
// makes three scalar passes over a row).  The VAE mid-block attention: 32768 rows x 4096.
template <int NV>
__global__ __launch_bounds__(256) void softmax_rows_f32_bf16_kernel(const float* __restrict__ x, long ldx, bf16_t* __restrict__ out, long ldo,
                                                                    int rows, float scale) {
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const long row = (long)blockIdx.x * 4 + wid;
  if (row >= rows) return;
  const f32x4* xr = reinterpret_cast<const f32x4*>(x + row * ldx);
  f32x4 v[NV];
#pragma unroll
  for (int i = 0; i < NV; ++i) v[i] = xr[i * 64 + lane];
  float m = -1e30f;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    v[i] *= scale;
    m = fmaxf(m, fmaxf(fmaxf(v[i][0], v[i][1]), fmaxf(v[i][2], v[i][3])));
  }
  m = wave_max(m);
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
#pragma unroll
    for (int e = 0; e < 4; ++e) { v[i][e] = __expf(v[i][e] - m); s += v[i][e]; }
  }
  s = wave_sum(s);
  const float inv = 1.f / s;
  u32x2* orow = reinterpret_cast<u32x2*>(out + row * ldo);
#pragma unroll
  for (int i = 0; i < NV; ++i)
    orow[i * 64 + lane] = (u32x2){pack_bf16x2(v[i][0] * inv, v[i][1] * inv), pack_bf16x2(v[i][2] * inv, v[i][3] * inv)};
}

// q / k of the VAE attention as hi / lo bf16 pieces for a three-term product on the bf16 MFMA path:
//   qk [rows][4 C] fp32 = [t Wq_hi^T | t Wq_lo^T | t Wk_hi^T | t Wk_lo^T]  (mmgt_gemm_bf16_f32 against the stacked weight pieces)
//   q = qk[0] + qk[1] + bias_q, k = qk[2] + qk[3] + bias_k, x = hi + lo with hi = bf16(x), lo = bf16(x - hi)
//   Qp [rows][3 C] = [q_hi | q_hi | q_lo],  Kp [rows][3 C] = [k_hi | k_lo | k_hi]   ->   Qp . Kp = q_hi k_hi + q_hi k_lo + q_lo k_hi,
// i.e. q . k without the lo x lo term: relative error ~2^-17 per product against 2^-9 for plain bf16 operands.
__global__ __launch_bounds__(256) void qk_split3_kernel(const float* __restrict__ qk, const float* __restrict__ bias_q,
                                                        const float* __restrict__ bias_k, bf16_t* __restrict__ Qp, bf16_t* __restrict__ Kp,
                                                        long rows, int C) {
  const int nv = C / 4;
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= rows * nv) return;
  const long r = idx / nv;
  const int c = (int)(idx - r * nv) * 4;
  const float* src = qk + r * 4 * C + c;
  const f32x4 q = *reinterpret_cast<const f32x4*>(src) + *reinterpret_cast<const f32x4*>(src + C) + *reinterpret_cast<const f32x4*>(bias_q + c);
  const f32x4 k = *reinterpret_cast<const f32x4*>(src + 2 * C) + *reinterpret_cast<const f32x4*>(src + 3 * C) +
                  *reinterpret_cast<const f32x4*>(bias_k + c);
  auto split = [](const f32x4& x, u32x2& hi, u32x2& lo) {
    hi = (u32x2){pack_bf16x2(x[0], x[1]), pack_bf16x2(x[2], x[3])};
    const float h0 = __uint_as_float(hi[0] << 16), h1 = __uint_as_float(hi[0] & 0xffff0000u);
    const float h2 = __uint_as_float(hi[1] << 16), h3 = __uint_as_float(hi[1] & 0xffff0000u);
    lo = (u32x2){pack_bf16x2(x[0] - h0, x[1] - h1), pack_bf16x2(x[2] - h2, x[3] - h3)};
  };
  u32x2 qh, ql, kh, kl;
  split(q, qh, ql);
  split(k, kh, kl);
  bf16_t* qd = Qp + r * 3 * C + c;
  bf16_t* kd = Kp + r * 3 * C + c;
  *reinterpret_cast<u32x2*>(qd) = qh;
  *reinterpret_cast<u32x2*>(qd + C) = qh;
  *reinterpret_cast<u32x2*>(qd + 2 * C) = ql;
  *reinterpret_cast<u32x2*>(kd) = kh;
  *reinterpret_cast<u32x2*>(kd + C) = kl;
  *reinterpret_cast<u32x2*>(kd + 2 * C) = kh;
}

}  // namespace

extern "C" int mmgt_ncfhw_to_nhwc(const float* in, void* out, int B, int C, int F, int H, int W, int Cpad, float scale,
                                  int dtype, void* stream) {
  MMGT_CHECK(in && out && Cpad >= C && C > 0, "ncfhw_to_nhwc: bad arguments");
  MMGT_CHECK(dtype == MMGT_F32 || dtype == MMGT_BF16, "ncfhw_to_nhwc: bad dtype");
  MMGT_CHECK(Cpad % 8 == 0, "ncfhw_to_nhwc: the padded channel count must be a multiple of 8 (Cpad=%d)", Cpad);
  MMGT_CHECK((long)B * F * H * W * (Cpad / 8) < (1l << 31), "ncfhw_to_nhwc: tensor too large");
  const long total = (long)B * F * H * W * (Cpad / 8);
  if (dtype == MMGT_BF16)
    hipLaunchKernelGGL(ncfhw_to_nhwc_kernel<bf16_t>, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, in,
                       (bf16_t*)out, B, C, F, H * W, Cpad, scale);
  else
    hipLaunchKernelGGL(ncfhw_to_nhwc_kernel<float>, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, in,
                       (float*)out, B, C, F, H * W, Cpad, scale);
  MMGT_LAUNCH_CHECK();
  return 0;
}

extern "C" int mmgt_nhwc_to_ncfhw(const void* in, float* out, int B, int C, int F, int H, int W, int Cpad, float scale,
                                  float shift, int clamp01, int dtype, void* stream) {
  MMGT_CHECK(in && out && Cpad >= C && C > 0, "nhwc_to_ncfhw: bad arguments");
  MMGT_CHECK(dtype == MMGT_F32 || dtype == MMGT_BF16, "nhwc_to_ncfhw: bad dtype");
  const long total = (long)B * C * F * H * W;
  if (dtype == MMGT_BF16)
    hipLaunchKernelGGL(nhwc_to_ncfhw_kernel<bf16_t>, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream,
                       (const bf16_t*)in, out, B, C, F, H * W, Cpad, scale, shift, clamp01);
  else
    hipLaunchKernelGGL(nhwc_to_ncfhw_kernel<float>, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream,
                       (const float*)in, out, B, C, F, H * W, Cpad, scale, shift, clamp01);
  MMGT_LAUNCH_CHECK();
  return 0;
}

// A 3 x 3 conv with FOUR output channels as a GEMM + a gather: Y[pixel][4 tap + o] = W[o][tap] . x[pixel] for all nine taps at once (one 36-column
// GEMM over the 320 channels of every pixel: csrc/rowgemm.hip with the GroupNorm + SiLU of conv_norm_out in its prologue), then
// out[n][y][x][o] = bias[o] + sum over taps of Y[n][y + ky - 1][x + kx - 1][4 tap + o] (out-of-image neighbours contribute nothing: the zero padding).
// The implicit-GEMM form pads the four channels to a 64-column tile and gathers nine taps of 320 channels per output pixel: 156 us + 84 us of
// GroupNorm pass at 48 x 64 x 64 against ~45 us for the GEMM + this kernel's 25 MB.
namespace {
__global__ __launch_bounds__(256) void conv_taps_gather_kernel(const bf16_t* __restrict__ Y, int ldY, const float* __restrict__ bias, bf16_t* __restrict__ out,
                                                               int NB, int H, int W) {
  const long p = (long)blockIdx.x * 256 + threadIdx.x;
  if (p >= (long)NB * H * W) return;
  const int x = (int)(p % W), y = (int)((p / W) % H);
  float acc[4] = {bias ? bias[0] : 0.f, bias ? bias[1] : 0.f, bias ? bias[2] : 0.f, bias ? bias[3] : 0.f};
#pragma unroll
  for (int ky = 0; ky < 3; ++ky)
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) {
      const int yy = y + ky - 1, xx = x + kx - 1;
      if (yy >= 0 && yy < H && xx >= 0 && xx < W) {
        const u32x2 v = *reinterpret_cast<const u32x2*>(Y + (p + (long)(ky - 1) * W + (kx - 1)) * ldY + 4 * (ky * 3 + kx));
        acc[0] += __uint_as_float(v[0] << 16);
        acc[1] += __uint_as_float(v[0] & 0xffff0000u);
        acc[2] += __uint_as_float(v[1] << 16);
        acc[3] += __uint_as_float(v[1] & 0xffff0000u);
      }
    }
  *reinterpret_cast<u32x4*>(out + p * 8) = (u32x4){pack_bf16x2(acc[0], acc[1]), pack_bf16x2(acc[2], acc[3]), 0u, 0u};   // 4 channels + 4 zeros
}
}  // namespace

extern "C" int mmgt_conv_taps_gather(const void* Y, int ldY, const float* bias, void* out, int NB, int H, int W, int dtype, void* stream) {
  MMGT_CHECK(Y && out && NB > 0 && H > 0 && W > 0, "conv_taps_gather: bad arguments");
  MMGT_CHECK(dtype == MMGT_BF16, "conv_taps_gather: bf16 only");
  MMGT_CHECK(ldY >= 36 && ldY % 4 == 0 && ((uintptr_t)Y % 8) == 0 && ((uintptr_t)out % 16) == 0, "conv_taps_gather: Y needs >= 36 columns, 8-byte aligned rows");
  const long total = (long)NB * H * W;
  hipLaunchKernelGGL(conv_taps_gather_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)Y, ldY, bias,
                     (bf16_t*)out, NB, H, W);
  MMGT_LAUNCH_CHECK();
  return 0;
}

extern "C" int mmgt_timestep_features(const float* timesteps, void* out, int B, int dim, int dtype, void* stream) {
  MMGT_CHECK(timesteps && out && B > 0 && dim > 0 && dim % 2 == 0, "timestep_features: bad arguments");
  MMGT_CHECK(dtype == MMGT_F32 || dtype == MMGT_BF16, "timestep_features: bad dtype");
  if (dtype == MMGT_BF16)
    hipLaunchKernelGGL(timestep_kernel<bf16_t>, dim3(grid_for((long)B * dim)), dim3(256), 0, (hipStream_t)stream,
                       timesteps, (bf16_t*)out, B, dim);
  else
    hipLaunchKernelGGL(timestep_kernel<float>, dim3(grid_for((long)B * dim)), dim3(256), 0, (hipStream_t)stream,
                       timesteps, (float*)out, B, dim);
  MMGT_LAUNCH_CHECK();
  return 0;
}

extern "C" int mmgt_silu(const void* x, void* out, long n, int dtype, void* stream) {
  MMGT_CHECK(x && out && n > 0, "silu: bad arguments");
  MMGT_CHECK(dtype == MMGT_F32 || dtype == MMGT_BF16, "silu: bad dtype");
  if (dtype == MMGT_BF16)
    hipLaunchKernelGGL(silu_kernel<bf16_t>, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x,
                       (bf16_t*)out, n);
  else
    hipLaunchKernelGGL(silu_kernel<float>, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, (const float*)x,
                       (float*)out, n);
  MMGT_LAUNCH_CHECK();
  return 0;
}

extern "C" int mmgt_cfg_ddim_step(const float* pred_sum, const float* counter, const float* latents, float* latents_out,
                                  long n, int F, int hw, float guidance, float sa_t, float sb_t, float sa_p, float sb_p,
                                  void* stream) {
  MMGT_CHECK(pred_sum && counter && latents && latents_out && n > 0 && F > 0 && hw > 0, "cfg_ddim_step: bad arguments");
  hipLaunchKernelGGL(cfg_ddim_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, pred_sum, counter, latents,
                     latents_out, n, F, hw, guidance, sa_t, sb_t, sa_p, sb_p);
  MMGT_LAUNCH_CHECK();
  return 0;
}

extern "C" int mmgt_accumulate_window_rows(const void* pred, float* pred_sum, float* counter, const int* idx, int Fw, int F,
                                           int C, int Cpad, int hw, int rows, int row0, int bump_counter, int dtype,
                                           void* stream) {
  MMGT_CHECK(pred && pred_sum && counter && idx, "accumulate_window: null pointer");
  MMGT_CHECK(Fw > 0 && Fw <= 256 && F >= Fw && C > 0 && Cpad >= C && hw > 0, "accumulate_window: bad sizes");
  MMGT_CHECK(rows >= 1 && row0 >= 0 && row0 + rows <= 2, "accumulate_window: CFG rows [%d, %d) outside [0, 2)", row0, row0 + rows);
  MMGT_CHECK(dtype == MMGT_F32 || dtype == MMGT_BF16, "accumulate_window: bad dtype");
  const long total = (long)rows * C * Fw * hw;
  if (dtype == MMGT_BF16)
    hipLaunchKernelGGL(accumulate_window_kernel<bf16_t>, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream,
                       (const bf16_t*)pred, pred_sum, counter, idx, Fw, F, C, Cpad, hw, rows, row0, bump_counter);
  else
    hipLaunchKernelGGL(accumulate_window_kernel<float>, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream,
                       (const float*)pred, pred_sum, counter, idx, Fw, F, C, Cpad, hw, rows, row0, bump_counter);
  MMGT_LAUNCH_CHECK();
  return 0;
}

extern "C" int mmgt_accumulate_window(const void* pred, float* pred_sum, float* counter, const int* idx, int Fw, int F,
                                      int C, int Cpad, int hw, int dtype, void* stream) {
  return mmgt_accumulate_window_rows(pred, pred_sum, counter, idx, Fw, F, C, Cpad, hw, 2, 0, 1, dtype, stream);
}

extern "C" int mmgt_softmax_rows(const void* x, long ldx, void* out, long ldo, int rows, int cols, float scale, int dtype,
                                 void* stream) {
  MMGT_CHECK(x && out && rows > 0 && cols > 0, "softmax_rows: bad arguments");
  MMGT_CHECK(dtype == MMGT_F32 || dtype == MMGT_BF16, "softmax_rows: bad dtype");
  dim3 grid((rows + 3) / 4);
  if (dtype == MMGT_BF16)
    hipLaunchKernelGGL(softmax_rows_kernel<bf16_t>, grid, dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, ldx,
                       (bf16_t*)out, ldo, rows, cols, scale);
  else
    hipLaunchKernelGGL(softmax_rows_kernel<float>, grid, dim3(256), 0, (hipStream_t)stream, (const float*)x, ldx,
                       (float*)out, ldo, rows, cols, scale);
  MMGT_LAUNCH_CHECK();
  return 0;
}

extern "C" int mmgt_softmax_rows_f32_bf16(const float* x, long ldx, void* out, long ldo, int rows, int cols, float scale, void* stream) {
  MMGT_CHECK(x && out && rows > 0, "softmax_rows_f32_bf16: bad arguments");
  MMGT_CHECK(cols % 256 == 0 && cols >= 256 && cols <= 8192 && ldx % 4 == 0 && ldo % 4 == 0 && ((uintptr_t)x % 16) == 0 &&
                 ((uintptr_t)out % 8) == 0,
             "softmax_rows_f32_bf16: cols must be a multiple of 256 up to 8192 with aligned rows (cols=%d)", cols);
  dim3 grid((rows + 3) / 4);
  hipStream_t s = (hipStream_t)stream;
  bf16_t* o = reinterpret_cast<bf16_t*>(out);
  switch (cols / 256) {
#define SM_CASE(NV_) case NV_: hipLaunchKernelGGL(softmax_rows_f32_bf16_kernel<NV_>, grid, dim3(256), 0, s, x, ldx, o, ldo, rows, scale); break
    SM_CASE(1); SM_CASE(2); SM_CASE(3); SM_CASE(4); SM_CASE(6); SM_CASE(8); SM_CASE(12); SM_CASE(16); SM_CASE(24); SM_CASE(32);
#undef SM_CASE
    default: MMGT_CHECK(false, "softmax_rows_f32_bf16: cols / 256 = %d has no instantiation", cols / 256);
  }
  MMGT_LAUNCH_CHECK();
  return 0;
}

extern "C" int mmgt_qk_split3(const float* qk, const float* bias_q, const float* bias_k, void* Qp, void* Kp, long rows, int C, void* stream) {
  MMGT_CHECK(qk && bias_q && bias_k && Qp && Kp && rows > 0 && C > 0 && C % 4 == 0, "qk_split3: bad arguments");
  MMGT_CHECK(((uintptr_t)qk % 16) == 0 && ((uintptr_t)bias_q % 16) == 0 && ((uintptr_t)bias_k % 16) == 0 && ((uintptr_t)Qp % 8) == 0 &&
                 ((uintptr_t)Kp % 8) == 0, "qk_split3: misaligned pointer");
  const long total = rows * (C / 4);
  MMGT_CHECK(total < (1l << 31) * 256, "qk_split3: too large");
  hipLaunchKernelGGL(qk_split3_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, qk, bias_q, bias_k,
                     reinterpret_cast<bf16_t*>(Qp), reinterpret_cast<bf16_t*>(Kp), rows, C);
  MMGT_LAUNCH_CHECK();
  return 0;
}
