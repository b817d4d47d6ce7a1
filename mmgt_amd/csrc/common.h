// Shared device helpers for the MMGT Stage-2 HIP kernels (gfx950 / CDNA4 only).
//
// Every matrix kernel is written once over a storage type T in {float, bf16}: the bf16 instantiation is the
// product path (v_mfma_f32_32x32x16_bf16, fp32 accumulate); the float instantiation is the fp32-I/O parity mode of the
// SAME kernel (v_mfma_f32_32x32x2_f32, bit-for-bit an fp32 fma chain) that the rtol 1e-3 / atol 1e-4 gate runs on.
// Both use one fragment convention: lane (r = lane & 31, h = lane >> 5) holds 8 consecutive k values k = 8h .. 8h+7 of
// row/column r for a K-step of 16.
#pragma once
// Cache policy of the kernels' OUTPUT stores (the `aux` field of a raw buffer store): 0 = default (the line stays dirty in the XCD's L2 until it is
// evicted or the kernel's end writes it back), 16 = sc1 (write-through).  Experiment knob (make wt).
#ifndef MMGT_ST_AUX
#define MMGT_ST_AUX 0
#endif
#include <hip/hip_runtime.h>
#include <hip/hip_bf16.h>
#include <stdint.h>

typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) short s16x8;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;

typedef unsigned short bf16_t;  // raw storage

__device__ __forceinline__ float bf16_to_f32(bf16_t v) { return __uint_as_float(((unsigned int)v) << 16); }
__device__ __forceinline__ bf16_t f32_to_bf16(float f) {
  // plain cast: hipcc emits v_cvt_pk_bf16_f32 (RNE, NaN-preserving)
  __hip_bfloat16 b = __float2bfloat16(f);
  return *reinterpret_cast<bf16_t*>(&b);
}

template <typename T> struct Elem;
template <> struct Elem<float> {
  static __device__ __forceinline__ float ld(const float* p) { return *p; }
  static __device__ __forceinline__ void st(float* p, float v) { *p = v; }
  static __device__ __forceinline__ float cvt(float v) { return v; }
};
template <> struct Elem<bf16_t> {
  static __device__ __forceinline__ float ld(const bf16_t* p) { return bf16_to_f32(*p); }
  static __device__ __forceinline__ void st(bf16_t* p, float v) { *p = f32_to_bf16(v); }
  static __device__ __forceinline__ float cvt(float v) { return bf16_to_f32(f32_to_bf16(v)); }
};

// ---- MFMA fragment: 8 consecutive-k elements of T per lane ---------------------------------------------------------
template <typename T> struct Frag;
template <> struct Frag<bf16_t> {
  s16x8 v;
  __device__ __forceinline__ void zero() { v = (s16x8)(0); }
  __device__ __forceinline__ void set(int j, float f) { v[j] = (short)f32_to_bf16(f); }
};
template <> struct Frag<float> {
  float v[8];
  __device__ __forceinline__ void zero() {
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = 0.f;
  }
  __device__ __forceinline__ void set(int j, float f) { v[j] = f; }
};

// acc(32x32) += A(32x16) * B(16x32); C/D layout: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5).
__device__ __forceinline__ void mma32(f32x16& acc, const Frag<bf16_t>& a, const Frag<bf16_t>& b) {
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.v, b.v, acc, 0, 0, 0);
}
__device__ __forceinline__ void mma32(f32x16& acc, const Frag<float>& a, const Frag<float>& b) {
  // K=2 per instruction: lanes 0-31 supply k = j, lanes 32-63 k = 8 + j; eight of them cover the 16-wide K-step.
#pragma unroll
  for (int j = 0; j < 8; ++j) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.v[j], b.v[j], acc, 0, 0, 0);
}

// Load a fragment (8 consecutive elements) from LDS / global at a 16-byte (bf16) or 32-byte (f32) aligned address.
__device__ __forceinline__ void frag_load(Frag<bf16_t>& f, const bf16_t* p) { f.v = *reinterpret_cast<const s16x8*>(p); }
__device__ __forceinline__ void frag_load(Frag<float>& f, const float* p) {
  f32x4 lo = *reinterpret_cast<const f32x4*>(p);
  f32x4 hi = *reinterpret_cast<const f32x4*>(p + 4);
#pragma unroll
  for (int j = 0; j < 4; ++j) { f.v[j] = lo[j]; f.v[4 + j] = hi[j]; }
}

// Fill a fragment from eight fp32 values.  bf16: four v_cvt_pk_bf16_f32 (RNE) -- element-wise set() costs a convert plus
// two or three pack instructions per value, which made the P^T fragments the largest VALU item of the attention loop.
typedef __attribute__((ext_vector_type(2))) float f32x2_;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_;
__device__ __forceinline__ unsigned pack_bf16x2(float a, float b) {
  bf16x2_ r = __builtin_convertvector((f32x2_){a, b}, bf16x2_);
  return *reinterpret_cast<unsigned*>(&r);
}
__device__ __forceinline__ void frag_set8(Frag<bf16_t>& f, const float (&v)[8]) {
  union { u32x4 u; s16x8 s; } cv;
  cv.u = (u32x4){pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]), pack_bf16x2(v[4], v[5]), pack_bf16x2(v[6], v[7])};
  f.v = cv.s;
}
__device__ __forceinline__ void frag_set8(Frag<float>& f, const float (&v)[8]) {
#pragma unroll
  for (int j = 0; j < 8; ++j) f.v[j] = v[j];
}

__device__ __forceinline__ float frag_get(const Frag<bf16_t>& f, int j) { return bf16_to_f32((bf16_t)f.v[j]); }
__device__ __forceinline__ float frag_get(const Frag<float>& f, int j) { return f.v[j]; }

__device__ __forceinline__ int acc_row(int reg, int lane) { return (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5); }

__device__ __forceinline__ float silu_f(float x) { return x / (1.f + __expf(-x)); }
// Mish (SMGA's FiLM generators and time MLP): x * tanh(softplus(x)) = x * (e^2 + 2 e) / (e^2 + 2 e + 2), e = exp(x)
__device__ __forceinline__ float mish_f(float x) {
  if (x > 20.f) return x;                                     // tanh(softplus(x)) == 1 to fp32 precision; avoids inf / inf
  const float e = __expf(x), n = e * (e + 2.f);
  return x * n / (n + 2.f);
}
// CLIP's activation (transformers `quick_gelu`): x * sigmoid(1.702 x)
__device__ __forceinline__ float quick_gelu_f(float x) { return x / (1.f + __expf(-1.702f * x)); }
// Exact-erf GELU, x * Phi(x), with erf from Abramowitz & Stegun 7.1.28:
//   erf(z) = 1 - (1 + a1 z + a2 z^2 + ... + a6 z^6)^-16,  |error| <= 3e-7  (fp32 evaluation: |gelu error| < 1e-6)
// i.e. six FMAs, four squarings and ONE reciprocal per value -- the libm erff costs ~4x more VALU issue slots, and
// 7.1.26 needs a reciprocal AND an exponential (both quarter rate).  With z = |x| / sqrt(2) folded into the coefficients
// and h = (1 - erf(z)) / 2 = 0.5 p^-16:   x Phi(x) = x (1 - h) for x >= 0, x h for x < 0  ==  max(x, 0) - |x| h,
// so the sign select disappears; the 0.5 rides in the polynomial (every coefficient times 2^(1/16), p'^16 = 2 p^16).
typedef __attribute__((ext_vector_type(2))) float f32x2;
#define MMGT_GELU_K  1.0442737824274138f   /* 2^(1/16) */
#define MMGT_GELU_C1 (0.04986734694f * MMGT_GELU_K)       /* 0.0705230784 / 2^(1/2) */
#define MMGT_GELU_C2 (0.02114100615f * MMGT_GELU_K)       /* 0.0422820123 / 2       */
#define MMGT_GELU_C3 (0.003277626324f * MMGT_GELU_K)      /* 0.0092705272 / 2^(3/2) */
#define MMGT_GELU_C4 (0.000038003575f * MMGT_GELU_K)      /* 0.0001520143 / 4       */
#define MMGT_GELU_C5 (0.000048890636f * MMGT_GELU_K)      /* 0.0002765672 / 2^(5/2) */
#define MMGT_GELU_C6 (0.00000538297500f * MMGT_GELU_K)    /* 0.0000430638 / 8       */
__device__ __forceinline__ float gelu_erf_f(float x) {
  const float z = fabsf(x);
  float p = fmaf(MMGT_GELU_C6, z, MMGT_GELU_C5);
  p = fmaf(p, z, MMGT_GELU_C4);
  p = fmaf(p, z, MMGT_GELU_C3);
  p = fmaf(p, z, MMGT_GELU_C2);
  p = fmaf(p, z, MMGT_GELU_C1);
  p = fmaf(p, z, MMGT_GELU_K);
  p *= p; p *= p; p *= p; p *= p;
  const float h = __builtin_amdgcn_rcpf(p);                  // (1 - erf(z)) / 2;  p = inf -> 0
  return fmaf(-z, h, fmaxf(x, 0.f));
}
// Two values per issue slot (v_pk_fma_f32 / v_pk_mul_f32): in the GEGLU epilogue, which does not run beside MFMAs, the
// packed form measured 13% faster on the whole ff1 GEMM than two scalar chains.
__device__ __forceinline__ f32x2 gelu_erf_f2(f32x2 x) {
  const f32x2 z = {fabsf(x[0]), fabsf(x[1])};
  f32x2 p = z * MMGT_GELU_C6 + MMGT_GELU_C5;
  p = p * z + MMGT_GELU_C4;
  p = p * z + MMGT_GELU_C3;
  p = p * z + MMGT_GELU_C2;
  p = p * z + MMGT_GELU_C1;
  p = p * z + MMGT_GELU_K;
  p *= p; p *= p; p *= p; p *= p;
  const f32x2 h = {__builtin_amdgcn_rcpf(p[0]), __builtin_amdgcn_rcpf(p[1])};
  const f32x2 r = {fmaxf(x[0], 0.f), fmaxf(x[1], 0.f)};
  return r - z * h;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
  return v;
}

// ---- host side error plumbing (no exceptions cross the C ABI) ------------------------------------------------------
#ifdef __cplusplus
extern "C" {
#endif
void mmgt_set_error(const char* fmt, ...);
#ifdef __cplusplus
}
#endif

#define MMGT_CHECK(cond, ...)      \
  do {                             \
    if (!(cond)) {                 \
      mmgt_set_error(__VA_ARGS__); \
      return 1;                    \
    }                              \
  } while (0)

#define MMGT_LAUNCH_CHECK()                                              \
  do {                                                                   \
    hipError_t e_ = hipGetLastError();                                   \
    if (e_ != hipSuccess) {                                              \
      mmgt_set_error("%s:%d: %s", __FILE__, __LINE__, hipGetErrorString(e_)); \
      return 2;                                                          \
    }                                                                    \
  } while (0)

// 16-byte output store through a pointer with the cache policy of MMGT_ST_AUX (the asm form is invisible to hipcc's vmcnt bookkeeping: the
// hardware still counts it, so compiler waits can only wait for more than they need)
__device__ __forceinline__ void st_out16(void* p, u32x4 v) {
#if MMGT_ST_AUX == 16
  asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
#else
  *reinterpret_cast<u32x4*>(p) = v;
#endif
}
