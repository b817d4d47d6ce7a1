// LayerNorm -> GEGLU FeedForward -> + residual as ONE kernel for the 320-channel level (bf16, gfx950).
//
//   out[m, :] = res[m, :] + b2 + W2 . ( h (.) gelu(g) ),   [h | g] = W1 . LN(x[m, :]) + b1          (diffusers FeedForward,
//   activation_fn = "geglu": attention.py:361,465,642,769 call sites)
//
// Why.  At level 0 (196 608 tokens x 320 channels) the three launches LayerNorm -> ff1 (N = 2560, K = 320, GEGLU) -> ff2
// (K = 1280) + residual move 126 + 126 | 126 + 503 | 503 + 126 + 126 MB through HBM per block for 483 GFLOP, and the K = 320
// GEMM tiles spend more time filling and draining than multiplying (profiles/r2/gemm16_tile_trace_r2.txt).  Here a wave keeps
// its 32 token rows for the whole block: the normalised rows sit in 80 registers as the B operand of ff1, the hidden
// activations never leave the register file (the ff1 accumulator, GEGLU'd and packed to bf16, IS the B operand of ff2: "an
// accumulator tile as the next MFMA's operand", cdna_hip_programming.md section 3), the 32 x 320 output tile accumulates in 160
// registers over all 1280 hidden channels, and only the weights stream: 2.5 MB per block and layer, L2 / Infinity-Cache
// resident, through a 2-stage LDS ring filled by LDS-DMA.  HBM traffic: x once in (it is also the residual), out once.
//
// Orientation (v_mfma_f32_32x32x16_bf16, D = A . B, lane (r = lane & 31, hh = lane >> 5)):
//   ff1   H^T[hidden 32 x rows 32] += W1[hidden, k] . xn^T[k, rows]     A = W1 fragment (LDS), B = xn fragment (registers)
//   ff2   O^T[chan 32 x rows 32]   += W2[chan, hidden] . G^T[hidden, rows]   A = W2 fragment (LDS), B = G (registers)
// A D tile holds column (token row) r on the lane and rows 4 hh + (i & 3) + 8 (i >> 2) in register i, so registers 8 s .. 8 s + 7
// of the GEGLU'd tile, converted pairwise to bf16, are the B fragment of k-step s whose element j is hidden channel
// 16 s + 8 (j >> 2) + 4 hh + (j & 3): the W2 image is packed in exactly that k order (mmgt_amd/packing.py: pack_ff_fused).
//
// Weight image (one per layer, built once per load_state_dict): per sub-block of 32 hidden channels 61 KiB =
//   [20 k-steps][h | gate] 1-KiB ff1 fragments | [10 channel tiles][2 k-steps] 1-KiB ff2 fragments | 64 ff1 biases | pad,
// every fragment lane-linear (lane l's 16 bytes at l * 16), so the image is copied to LDS by 61 linear 1-KiB LDS-DMA pieces
// and every ds_read_b128 is base + lane * 16 + immediate: conflict-free, no swizzle, no address arithmetic.
// Workgroup = 4 waves (one per SIMD, up to 512 registers each) = 128 token rows; one sub-block = 60 MFMAs per wave (1920
// matrix-pipe cycles) against 60 fragment reads and 16 GEGLU evaluations per lane; the DMA of sub-block s + 1 is issued behind
// the barrier that opens sub-block s, so a whole sub-block (~1 us) covers its L2 latency.
#include "common.h"
#include "gemm_common.h"
#include "mmgt_hip.h"

namespace {

constexpr int FFC = 320, FF_KS = FFC / 16, FF_NU = FFC / 32;
constexpr int FF_W1 = FF_KS * 2 * 1024, FF_W2 = FF_NU * 2 * 1024, FF_B1 = FF_W1 + FF_W2, FF_STAGE = 61 * 1024, FF_NPIECE = 61;
static_assert(FF_B1 + 256 <= FF_STAGE, "stage");

__device__ __forceinline__ f32x16 mma32b(s16x8 a, s16x8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0); }

__device__ __forceinline__ s16x8 pack8(const float (&v)[8]) {
  union { u32x4 u; s16x8 s; } cv;
  cv.u = (u32x4){pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]), pack_bf16x2(v[4], v[5]), pack_bf16x2(v[6], v[7])};
  return cv.s;
}

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1)))
void ff_fused_kernel(const bf16_t* __restrict__ x, long ldx, const float* __restrict__ gamma, const float* __restrict__ beta,
                     float eps, const char* __restrict__ wimg, int nsb, const float* __restrict__ bias2,
                     const bf16_t* __restrict__ res, long ldr, bf16_t* __restrict__ out, long ldo, int M) {
  extern __shared__ __attribute__((aligned(1024))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31, hh = lane >> 5;
  const long row = (long)blockIdx.x * 128 + wid * 32 + r;
  const long rowc = row < M ? row : M - 1;

  const __amdgpu_buffer_rsrc_t rw = dma_rsrc(wimg);
  auto issue = [&](int sb, int stage) {   // the wave's pieces wid, wid + 4, ... of sub-block sb's 61-KiB image
    const int soff = sb * FF_STAGE + wid * 1024;
    char* dst = smem + stage * FF_STAGE + wid * 1024;
#pragma unroll
    for (int i = 0; i < 15; ++i) blds16(rw, (unsigned)lane * 16u, soff + i * 4096, dst + i * 4096);
    if (wid == 0) blds16(rw, (unsigned)lane * 16u, soff + 15 * 4096, dst + 15 * 4096);
  };

  // ---- the wave's 32 rows as ff1 B fragments: lane (r, hh) holds channels 16 ks + 8 hh .. + 7 of row r
  s16x8 xf[FF_KS];
  {
    const bf16_t* xr = x + rowc * ldx + 8 * hh;
#pragma unroll
    for (int ks = 0; ks < FF_KS; ++ks) xf[ks] = *reinterpret_cast<const s16x8*>(xr + 16 * ks);
    // gamma | beta go through LDS (behind the two stages): with an LDS-DMA in flight hipcc waits vmcnt(0) for every plain
    // global load, which serialised 40 L2 round trips here; so the weight DMA also starts only after these loads
    float* lgb = reinterpret_cast<float*>(smem + 2 * FF_STAGE);
    if (tid < 3 * FFC / 4 && (gamma || tid >= 2 * FFC / 4)) {   // gamma | beta | bias2: 3 x 80 vectors
      const float* src = tid < FFC / 4 ? gamma + 4 * tid : tid < 2 * FFC / 4 ? beta + 4 * (tid - FFC / 4) : bias2 + 4 * (tid - 2 * FFC / 4);
      *reinterpret_cast<f32x4*>(lgb + 4 * tid) = *reinterpret_cast<const f32x4*>(src);
    }
    __syncthreads();
    issue(0, 0);
    if (gamma) {   // LayerNorm (exact two-pass statistics in registers, as ln_kernel): y = (x - mean) * rstd * gamma + beta
      float sum = 0.f;
#pragma unroll
      for (int ks = 0; ks < FF_KS; ++ks)
#pragma unroll
        for (int j = 0; j < 8; ++j) sum += bf16_to_f32((bf16_t)xf[ks][j]);
      sum += __shfl_xor(sum, 32);
      const float mean = sum / (float)FFC;
      float sq = 0.f;
#pragma unroll
      for (int ks = 0; ks < FF_KS; ++ks)
#pragma unroll
        for (int j = 0; j < 8; ++j) { const float d = bf16_to_f32((bf16_t)xf[ks][j]) - mean; sq += d * d; }
      sq += __shfl_xor(sq, 32);
      const float rstd = rsqrtf(sq / (float)FFC + eps);
#pragma unroll
      for (int ks = 0; ks < FF_KS; ++ks) {
        const int c = 16 * ks + 8 * hh;
        const f32x4 g0 = *reinterpret_cast<const f32x4*>(lgb + c), g1 = *reinterpret_cast<const f32x4*>(lgb + c + 4);
        const f32x4 b0 = *reinterpret_cast<const f32x4*>(lgb + FFC + c), b1 = *reinterpret_cast<const f32x4*>(lgb + FFC + c + 4);
        float y[8];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          y[j] = (bf16_to_f32((bf16_t)xf[ks][j]) - mean) * rstd * g0[j] + b0[j];
          y[4 + j] = (bf16_to_f32((bf16_t)xf[ks][4 + j]) - mean) * rstd * g1[j] + b1[j];
        }
        xf[ks] = pack8(y);
      }
    }
  }

  f32x16 oacc[FF_NU];
#pragma unroll
  for (int u = 0; u < FF_NU; ++u) oacc[u] = (f32x16)(0.f);

  for (int sb = 0; sb < nsb; ++sb) {
    wait_vmcnt<0>();                    // this wave's pieces of sub-block sb have landed (issued one sub-block ago)
    __builtin_amdgcn_s_barrier();       // ... and everybody's; every wave has finished reading the other stage
    if (sb + 1 < nsb) issue(sb + 1, (sb + 1) & 1);
    const char* st = smem + (sb & 1) * FF_STAGE;
    const char* sl = st + lane * 16;
    // ---- ff1: accumulators start from the bias (register i <-> hidden 4 hh + (i & 3) + 8 (i >> 2))
    f32x16 hacc, gacc;
    {
      const float* bl = reinterpret_cast<const float*>(st + FF_B1) + 4 * hh;
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) {
        const f32x4 bh = *reinterpret_cast<const f32x4*>(bl + 8 * g4), bg = *reinterpret_cast<const f32x4*>(bl + 32 + 8 * g4);
#pragma unroll
        for (int e = 0; e < 4; ++e) { hacc[4 * g4 + e] = bh[e]; gacc[4 * g4 + e] = bg[e]; }
      }
    }
    // fragment reads run PF k-steps ahead of their MFMAs (a ring of PF + 1 register pairs); the sched_barriers pin that order --
    // left alone, hipcc issues each pair of reads right in front of the MFMAs that need them and waits lgkmcnt(0) every step
    constexpr int PF = 3;
    s16x8 fr[PF + 1][2];
#pragma unroll
    for (int i = 0; i < PF; ++i) {
      fr[i][0] = *reinterpret_cast<const s16x8*>(sl + (2 * i) * 1024);
      fr[i][1] = *reinterpret_cast<const s16x8*>(sl + (2 * i + 1) * 1024);
    }
#pragma unroll
    for (int ks = 0; ks < FF_KS; ++ks) {
      if (ks + PF < FF_KS) {
        fr[(ks + PF) % (PF + 1)][0] = *reinterpret_cast<const s16x8*>(sl + (2 * (ks + PF)) * 1024);
        fr[(ks + PF) % (PF + 1)][1] = *reinterpret_cast<const s16x8*>(sl + (2 * (ks + PF) + 1) * 1024);
      } else {   // the first ff2 fragments ride behind the last ff1 reads: they do not depend on the GEGLU
        const int u = ks + PF - FF_KS;
        fr[(ks + PF) % (PF + 1)][0] = *reinterpret_cast<const s16x8*>(sl + FF_W1 + (2 * u) * 1024);
        fr[(ks + PF) % (PF + 1)][1] = *reinterpret_cast<const s16x8*>(sl + FF_W1 + (2 * u + 1) * 1024);
      }
      __builtin_amdgcn_sched_barrier(0);
      hacc = mma32b(fr[ks % (PF + 1)][0], xf[ks], hacc);
      gacc = mma32b(fr[ks % (PF + 1)][1], xf[ks], gacc);
      __builtin_amdgcn_sched_barrier(0);
    }
    // ---- GEGLU in registers -> the two B fragments of ff2
    s16x8 gb[2];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      float v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = hacc[8 * s + j] * gelu_erf_f(gacc[8 * s + j]);
      gb[s] = pack8(v);
    }
    // ---- ff2
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int u = 0; u < FF_NU; ++u) {
      if (u + PF < FF_NU) {
        fr[(FF_KS + u + PF) % (PF + 1)][0] = *reinterpret_cast<const s16x8*>(sl + FF_W1 + (2 * (u + PF)) * 1024);
        fr[(FF_KS + u + PF) % (PF + 1)][1] = *reinterpret_cast<const s16x8*>(sl + FF_W1 + (2 * (u + PF) + 1) * 1024);
      }
      __builtin_amdgcn_sched_barrier(0);
      oacc[u] = mma32b(fr[(FF_KS + u) % (PF + 1)][0], gb[0], oacc[u]);
      oacc[u] = mma32b(fr[(FF_KS + u) % (PF + 1)][1], gb[1], oacc[u]);
      __builtin_amdgcn_sched_barrier(0);
    }
  }

  // ---- epilogue: + b2 + residual, bf16, 16-byte stores.  Register group k (registers 4 k .. 4 k + 3) of tile u is channels
  // 32 u + 8 k + 4 hh + (0..3); v_permlane32_swap of groups (k, k + 1) gives lane hh = 0 channels 32 u + 8 k .. + 7 and lane hh = 1
  // channels 32 u + 8 k + 8 .. + 15 (cdna_hip_programming.md T21).  All 20 residual vectors are requested first (the x fragments
  // are dead: their registers take them), the stores go through a buffer resource sized to the M valid rows, so rows beyond M
  // are dropped by the range check instead of a branch per store.
  {
    const bf16_t* rr = res + rowc * ldr + 8 * hh;
    u32x4 rv[2 * FF_NU];
#pragma unroll
    for (int i = 0; i < 2 * FF_NU; ++i) rv[i] = *reinterpret_cast<const u32x4*>(rr + 16 * i);   // channels 16 i + 8 hh .. + 7
    const __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc(out, 0, (int)((long)M * ldo * 2), 0x00020000);
    const unsigned obase = (unsigned)(row * ldo + 8 * hh) * 2u;          // (rows >= M: beyond num_records -> dropped)
    const float* lb2 = reinterpret_cast<const float*>(smem + 2 * FF_STAGE) + 2 * FFC + 8 * hh;
#pragma unroll
    for (int u = 0; u < FF_NU; ++u)
#pragma unroll
      for (int k = 0; k < 4; k += 2) {
        const int c = 32 * u + 8 * k;          // + 8 hh in the bases
        union { u32x4 q; bf16_t e[8]; } r8;
        r8.q = rv[2 * u + k / 2];
        const f32x4 b0 = *reinterpret_cast<const f32x4*>(lb2 + c), b1 = *reinterpret_cast<const f32x4*>(lb2 + c + 4);
        float o8[8];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(oacc[u][4 * k + e]), __float_as_uint(oacc[u][4 * k + 4 + e]), false, false);
          o8[e] = __uint_as_float(sw[0]) + b0[e] + bf16_to_f32(r8.e[e]);
          o8[4 + e] = __uint_as_float(sw[1]) + b1[e] + bf16_to_f32(r8.e[4 + e]);
        }
        const u32x4 pk = (u32x4){pack_bf16x2(o8[0], o8[1]), pack_bf16x2(o8[2], o8[3]), pack_bf16x2(o8[4], o8[5]), pack_bf16x2(o8[6], o8[7])};
        __builtin_amdgcn_raw_buffer_store_b128(pk, ro, (int)(obase + 2u * c), 0, 0);
      }
  }
}

}  // namespace

extern "C" int mmgt_ff_fused_image_bytes(int C, int inner) {
  if (C != FFC || inner <= 0 || inner % 32) return -1;
  return (inner / 32) * FF_STAGE;
}

extern "C" int mmgt_ff_fused(const void* x, long ldx, const float* ln_gamma, const float* ln_beta, float eps, const void* wimg,
                             const float* bias2, const void* residual, long ldr, void* out, long ldo, int M, int C, int inner,
                             int dtype, void* stream) {
  MMGT_CHECK(x && wimg && bias2 && residual && out, "ff_fused: null pointer");
  MMGT_CHECK(dtype == MMGT_BF16, "ff_fused: bf16 only (the fp32-I/O mode runs LayerNorm / GEMM / GEMM)");
  MMGT_CHECK(C == FFC && inner > 0 && inner % 32 == 0, "ff_fused: built for %d channels (got %d) and inner %% 32 == 0 (got %d)", FFC, C, inner);
  MMGT_CHECK((ln_gamma != nullptr) == (ln_beta != nullptr), "ff_fused: gamma / beta must come together");
  MMGT_CHECK((long)M * ldo * 2 < (1l << 31), "ff_fused: output beyond the 2 GiB range of a buffer resource (split the rows)");
  MMGT_CHECK(M > 0 && ldx >= C && ldr >= C && ldo >= C && ldx % 8 == 0 && ldr % 8 == 0 && ldo % 8 == 0, "ff_fused: bad M=%d or row strides", M);
  MMGT_CHECK((((uintptr_t)x | (uintptr_t)residual | (uintptr_t)out | (uintptr_t)wimg | (uintptr_t)bias2) & 15) == 0 &&
                 (!ln_gamma || (((uintptr_t)ln_gamma | (uintptr_t)ln_beta) & 15) == 0),
             "ff_fused: pointers must be 16-byte aligned");
  const size_t lds = 2 * FF_STAGE + 3 * FFC * sizeof(float);
  auto kern = ff_fused_kernel;
  static bool attr = false;
  if (!attr) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
      mmgt_set_error("ff_fused: cannot reserve %zu bytes of LDS", lds);
      return 2;
    }
    attr = true;
  }
  const unsigned grid = (unsigned)((M + 127) / 128);
  hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, (hipStream_t)stream, (const bf16_t*)x, ldx, ln_gamma, ln_beta, eps,
                     (const char*)wimg, inner / 32, bias2, (const bf16_t*)residual, ldr, (bf16_t*)out, ldo, M);
  MMGT_LAUNCH_CHECK();
  return 0;
}
