// Element-wise pieces of the Stage-1 SMGA audio -> pose sampler (gfx950): rotary position rotation, FiLM-modulated residual
// add, token mean and the guided x0-prediction DDIM update.  The matrix work of the sampler (Linear layers, attention,
// LayerNorm) runs on the kernels of gemm.hip / attention.hip / norm.hip; these are the HBM-bound glue between them.
//
// Replaces (paths relative to the reference checkout): src/audio2pose_model/rotary_embedding_torch.py:38-61,106-113
// (apply_rotary_emb on the attention inputs, model.py:121,267,298-299), model.py:44-64 (featurewise_affine of a DenseFiLM
// output + residual, :246-259), model.py:460 (mean pooling of the condition tokens), src/audio2pose_model/diffusion.py:149-156,
// 257-273 (guided x0 prediction, clip, predict_noise_from_start, DDIM update with eta = 1).
#include "common.h"
#include "mmgt_hip.h"

namespace {

inline int grid_for(long n, int block = 256) {
  long g = (n + block - 1) / block;
  return (int)(g < 1 ? 1 : (g > 8192 ? 8192 : g));
}

// out[r][2i], out[r][2i+1] = x0 c - x1 s, x1 c + x0 s with (c, s) = table[(r % seq)][i]  (interleaved pairs, freqs_for "lang")
template <typename T>
__global__ void rotary_kernel(const T* __restrict__ x, const float* __restrict__ cs, T* __restrict__ out, long rows, int dim,
                              int seq) {
  const long total = rows * (dim / 2);
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int p = (int)(i % (dim / 2));
    const long r = i / (dim / 2);
    const float c = cs[((r % seq) * (dim / 2) + p) * 2], s = cs[((r % seq) * (dim / 2) + p) * 2 + 1];
    const float x0 = Elem<T>::ld(x + r * dim + 2 * p), x1 = Elem<T>::ld(x + r * dim + 2 * p + 1);
    Elem<T>::st(out + r * dim + 2 * p, x0 * c - x1 * s);
    Elem<T>::st(out + r * dim + 2 * p + 1, x1 * c + x0 * s);
  }
}

// out[r][c] = res[r][c] + (ss[b][c] + 1) * x[r][c] + ss[b][dim + c],  b = r / rows_per_batch;  res2 (optional) is added too
template <typename T>
__global__ void film_residual_kernel(const T* __restrict__ x, const float* __restrict__ ss, long ld_ss,
                                     const T* __restrict__ res, const T* __restrict__ res2, T* __restrict__ out, long rows,
                                     int dim, int rows_per_batch) {
  const long total = rows * dim;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int c = (int)(i % dim);
    const long r = i / dim;
    const float* s = ss + (r / rows_per_batch) * ld_ss;
    float v = (s[c] + 1.f) * Elem<T>::ld(x + i) + s[dim + c];
    if (res) v += Elem<T>::ld(res + i);
    if (res2) v += Elem<T>::ld(res2 + i);
    Elem<T>::st(out + i, v);
  }
}

// out[b][c] = mean over the tokens of x[b][t][c]  (fp32 out)
template <typename T>
__global__ void mean_tokens_kernel(const T* __restrict__ x, float* __restrict__ out, int batch, int tokens, int dim) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < batch * dim; i += gridDim.x * blockDim.x) {
    const int b = i / dim, c = i - b * dim;
    float s = 0.f;
    for (int t = 0; t < tokens; ++t) s += Elem<T>::ld(x + ((long)b * tokens + t) * dim + c);
    out[i] = s / (float)tokens;
  }
}

// x0 = clamp(unc + w (cond - unc), -1, 1); eps = (x sqrt(1/a) - x0) / sqrt(1/a - 1);
// x' = last ? x0 : x0 sqrt(a') + c eps + sigma noise     (all fp32: the sampler state never leaves fp32)
template <typename T>
__global__ void smga_ddim_kernel(const T* __restrict__ unc, const T* __restrict__ cond, const float* __restrict__ x,
                                 const float* __restrict__ noise, float* __restrict__ out, long n, float w, float recip,
                                 float recipm1, float a_next_sqrt, float c, float sigma, int last) {
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const float u = Elem<T>::ld(unc + i), cd = Elem<T>::ld(cond + i);
    const float x0 = fminf(fmaxf(u + (cd - u) * w, -1.f), 1.f);
    const float eps = (recip * x[i] - x0) / recipm1;
    out[i] = last ? x0 : x0 * a_next_sqrt + c * eps + sigma * noise[i];
  }
}

template <typename T>
__global__ void activation_kernel(const T* __restrict__ x, T* __restrict__ out, long n, int act) {
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const float v = Elem<T>::ld(x + i);
    Elem<T>::st(out + i, act == 2 ? silu_f(v) : act == 5 ? gelu_erf_f(v) : mish_f(v));
  }
}

}  // namespace

#define SMGA_LAUNCH(T, KERN, GRIDN, ...)                                                                          \
  hipLaunchKernelGGL(KERN<T>, dim3(grid_for(GRIDN)), dim3(256), 0, (hipStream_t)stream, __VA_ARGS__)

extern "C" int mmgt_rotary(const void* x, const float* cos_sin, void* out, long rows, int dim, int seq, int dtype, void* stream) {
  MMGT_CHECK(x && cos_sin && out && rows > 0 && dim > 0 && dim % 2 == 0 && seq > 0, "rotary: bad arguments");
  MMGT_CHECK(dtype == MMGT_F32 || dtype == MMGT_BF16, "rotary: bad dtype");
  if (dtype == MMGT_BF16) SMGA_LAUNCH(bf16_t, rotary_kernel, rows * (dim / 2), (const bf16_t*)x, cos_sin, (bf16_t*)out, rows, dim, seq);
  else SMGA_LAUNCH(float, rotary_kernel, rows * (dim / 2), (const float*)x, cos_sin, (float*)out, rows, dim, seq);
  MMGT_LAUNCH_CHECK();
  return 0;
}

extern "C" int mmgt_film_residual(const void* x, const float* scale_shift, long ld_ss, const void* res, const void* res2, void* out,
                                  long rows, int dim, int rows_per_batch, int dtype, void* stream) {
  MMGT_CHECK(x && scale_shift && out && rows > 0 && dim > 0 && rows_per_batch > 0 && ld_ss >= 2 * dim, "film_residual: bad arguments");
  MMGT_CHECK(dtype == MMGT_F32 || dtype == MMGT_BF16, "film_residual: bad dtype");
  if (dtype == MMGT_BF16)
    SMGA_LAUNCH(bf16_t, film_residual_kernel, rows * dim, (const bf16_t*)x, scale_shift, ld_ss, (const bf16_t*)res, (const bf16_t*)res2,
                  (bf16_t*)out, rows, dim, rows_per_batch);
  else
    SMGA_LAUNCH(float, film_residual_kernel, rows * dim, (const float*)x, scale_shift, ld_ss, (const float*)res, (const float*)res2,
                  (float*)out, rows, dim, rows_per_batch);
  MMGT_LAUNCH_CHECK();
  return 0;
}

extern "C" int mmgt_mean_tokens(const void* x, float* out, int batch, int tokens, int dim, int dtype, void* stream) {
  MMGT_CHECK(x && out && batch > 0 && tokens > 0 && dim > 0, "mean_tokens: bad arguments");
  MMGT_CHECK(dtype == MMGT_F32 || dtype == MMGT_BF16, "mean_tokens: bad dtype");
  if (dtype == MMGT_BF16) SMGA_LAUNCH(bf16_t, mean_tokens_kernel, (long)batch * dim, (const bf16_t*)x, out, batch, tokens, dim);
  else SMGA_LAUNCH(float, mean_tokens_kernel, (long)batch * dim, (const float*)x, out, batch, tokens, dim);
  MMGT_LAUNCH_CHECK();
  return 0;
}

extern "C" int mmgt_smga_ddim_step(const void* pred_uncond, const void* pred_cond, const float* x, const float* noise, float* out,
                                   long n, float guidance, float sqrt_recip_acp, float sqrt_recipm1_acp, float sqrt_acp_next,
                                   float c, float sigma, int last, int dtype, void* stream) {
  MMGT_CHECK(pred_uncond && pred_cond && x && out && (last || noise) && n > 0, "smga_ddim_step: bad arguments");
  MMGT_CHECK(dtype == MMGT_F32 || dtype == MMGT_BF16, "smga_ddim_step: bad dtype");
  if (dtype == MMGT_BF16)
    SMGA_LAUNCH(bf16_t, smga_ddim_kernel, n, (const bf16_t*)pred_uncond, (const bf16_t*)pred_cond, x, noise, out, n, guidance, sqrt_recip_acp,
                  sqrt_recipm1_acp, sqrt_acp_next, c, sigma, last);
  else
    SMGA_LAUNCH(float, smga_ddim_kernel, n, (const float*)pred_uncond, (const float*)pred_cond, x, noise, out, n, guidance, sqrt_recip_acp,
                  sqrt_recipm1_acp, sqrt_acp_next, c, sigma, last);
  MMGT_LAUNCH_CHECK();
  return 0;
}

extern "C" int mmgt_activation(const void* x, void* out, long n, int act, int dtype, void* stream) {
  MMGT_CHECK(x && out && n > 0, "activation: bad arguments");
  MMGT_CHECK(act == MMGT_ACT_SILU || act == MMGT_ACT_GELU || act == MMGT_ACT_MISH, "activation: act %d unsupported (SiLU, GELU, Mish)", act);
  MMGT_CHECK(dtype == MMGT_F32 || dtype == MMGT_BF16, "activation: bad dtype");
  if (dtype == MMGT_BF16) SMGA_LAUNCH(bf16_t, activation_kernel, n, (const bf16_t*)x, (bf16_t*)out, n, act);
  else SMGA_LAUNCH(float, activation_kernel, n, (const float*)x, (float*)out, n, act);
  MMGT_LAUNCH_CHECK();
  return 0;
}
