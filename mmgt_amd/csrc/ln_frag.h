// LayerNorm of rows that a wave holds as MFMA fragments (csrc/ffn.hip, csrc/rowgemm.hip): lane (r = lane & 31, hh = lane >> 5) holds
// channels 16 ks + 8 hh .. + 7 of row r in xf[ks]; the partner lane (r, 1 - hh) holds the other half of the row.
#pragma once
#include "common.h"

namespace {

// y = (x - mean) * rstd * gamma + beta, rounded to bf16 in place.  lgb (LDS): gamma[C] | beta[C] fp32.
// Statistics in ONE pass over the packed bf16 pairs with v_dot2c_f32_bf16 (sum x = dot(x2, (1, 1)), sum x^2 = dot(x2, x2): products of
// bf16 values are exact in fp32, four accumulators each), var = E[x^2] - mean^2 clamped at 0.  In fp32 its relative error is
// <= ~C 2^-24 (1 + mean^2 / var) -- 2e-5 (1 + mean^2 / var) for C = 320, under a tenth of a bf16 ulp on rstd up to |mean| = 15 std
// (tests/test_rowgemm_gpu.py holds a |mean| = 8 std case against fp64); the two passes over unpacked fp32 values that ln_kernel makes cost
// 4x the instructions, and in these kernels the LayerNorm is serial work in front of the first MFMA (in-kernel stamps: 23 000 of a
// workgroup's 99 000 cycles in rowgemm at N = 960 before, ~9 000 after).
// STATS = false: y = x * gamma + beta only -- the second pass of a GroupNorm whose per-(image, channel) scale / shift tables the
// caller put where gamma / beta are read (gn_apply_kernel's v = x * scale + shift, the same FMA: bit-identical).
// SILU (with STATS = false): y = silu(x * gamma + beta) -- GroupNorm + SiLU in front of a conv that runs as a GEMM (conv_out: csrc/rowgemm.hip, norm 3).
template <int KS, bool STATS = true, bool SILU = false>
__device__ __forceinline__ void layernorm_fragments(s16x8 (&xf)[KS], const float* lgb, int hh, float eps) {
  constexpr int C = 16 * KS;
  typedef __attribute__((ext_vector_type(2))) __bf16 bf2;
  const bf2 ones = {(__bf16)1.0f, (__bf16)1.0f};
  f32x2 r2 = {1.f, 1.f}, nm2 = {0.f, 0.f};
  if constexpr (STATS) {
    float s[4] = {0.f, 0.f, 0.f, 0.f}, q[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      union { s16x8 f; unsigned u[4]; } v;
      v.f = xf[ks];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const bf2 p = __builtin_bit_cast(bf2, v.u[j]);
        s[j] = __builtin_amdgcn_fdot2_f32_bf16(p, ones, s[j], false);
        q[j] = __builtin_amdgcn_fdot2_f32_bf16(p, p, q[j], false);
      }
    }
    float sum = (s[0] + s[1]) + (s[2] + s[3]), sq = (q[0] + q[1]) + (q[2] + q[3]);
    sum += __shfl_xor(sum, 32);
    sq += __shfl_xor(sq, 32);
    const float mean = sum * (1.f / (float)C);
    const float var = fmaxf(fmaf(-mean, mean, sq * (1.f / (float)C)), 0.f);
    const float rstd = rsqrtf(var + eps);
    r2 = (f32x2){rstd, rstd};
    nm2 = (f32x2){-mean * rstd, -mean * rstd};
  }
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) {
    const int c = 16 * ks + 8 * hh;
    union { s16x8 f; unsigned u[4]; } v;
    v.f = xf[ks];
    const f32x4 g0 = *reinterpret_cast<const f32x4*>(lgb + c), g1 = *reinterpret_cast<const f32x4*>(lgb + c + 4);
    const f32x4 b0 = *reinterpret_cast<const f32x4*>(lgb + C + c), b1 = *reinterpret_cast<const f32x4*>(lgb + C + c + 4);
    unsigned o[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const f32x2 x2 = {__uint_as_float(v.u[j] << 16), __uint_as_float(v.u[j] & 0xffff0000u)};
      const f32x2 z = STATS ? __builtin_elementwise_fma(x2, r2, nm2) : x2;
      const f32x2 g2 = j < 2 ? (f32x2){g0[2 * j], g0[2 * j + 1]} : (f32x2){g1[2 * j - 4], g1[2 * j - 3]};
      const f32x2 b2 = j < 2 ? (f32x2){b0[2 * j], b0[2 * j + 1]} : (f32x2){b1[2 * j - 4], b1[2 * j - 3]};
      f32x2 y = __builtin_elementwise_fma(z, g2, b2);
      if constexpr (SILU) y = (f32x2){silu_f(y[0]), silu_f(y[1])};
      o[j] = pack_bf16x2(y[0], y[1]);
    }
    union { u32x4 u; s16x8 s; } cv;
    cv.u = (u32x4){o[0], o[1], o[2], o[3]};
    xf[ks] = cv.s;
  }
}

}  // namespace
