// LayerNorm -> GEGLU FeedForward -> + residual as ONE kernel for the 320-channel level (bf16, gfx950).
//
//   out[m, :] = res[m, :] + b2 + W2 . ( h (.) gelu(g) ),   [h | g] = W1 . LN(x[m, :]) + b1          (diffusers FeedForward,
//   activation_fn = "geglu": attention.py:361,465,642,769 call sites)
//
// Why.  At level 0 (196 608 tokens x 320 channels) the three launches LayerNorm -> ff1 (N = 2560, K = 320, GEGLU) -> ff2
// (K = 1280) + residual move 126 + 126 | 126 + 503 | 503 + 126 + 126 MB through HBM per block for 483 GFLOP, and the K = 320
// GEMM tiles spend more time filling and draining than multiplying (profiles/r2/gemm16_tile_trace_r2.txt).  Here the 32 token rows
// of a row group stay on one SIMD for the whole block: the normalised rows sit in 80 registers as the B operand of ff1, the
// hidden activations never reach memory (the ff1 accumulator, GEGLU'd and packed to bf16, IS the B operand of ff2: "an
// accumulator tile as the next MFMA's operand", cdna_hip_programming.md section 3), the 32 x 320 output tile accumulates in 160
// registers over all 1280 hidden channels, and only the weights stream: 2.5 MB per block and layer, L2 resident (measured: the
// weight stream alone runs at 20 TB/s chip-wide), through LDS rings filled by LDS-DMA.  HBM traffic: x once in (it is also the
// residual), out once.
//
// Orientation (v_mfma_f32_32x32x16_bf16, D = A . B, lane (r = lane & 31, hh = lane >> 5)):
//   ff1   H^T[hidden 32 x rows 32] += W1[hidden, k] . xn^T[k, rows]     A = W1 fragment (LDS), B = xn fragment (registers)
//   ff2   O^T[chan 32 x rows 32]   += W2[chan, hidden] . G^T[hidden, rows]   A = W2 fragment (LDS), B = G fragment
// A D tile holds column (token row) r on the lane and rows 4 hh + (i & 3) + 8 (i >> 2) in register i, so registers 8 s .. 8 s + 7
// of the GEGLU'd tile, converted pairwise to bf16, are the B fragment of k-step s whose element j is hidden channel
// 16 s + 8 (j >> 2) + 4 hh + (j & 3): the W2 image is packed in exactly that k order (mmgt_amd/packing.py: pack_ff_fused).
//
// Weight image (one per layer, built once per load_state_dict): per sub-block of 32 hidden channels 61 KiB =
//   [20 k-steps][h | gate] 1-KiB ff1 fragments | [10 channel tiles][2 k-steps] 1-KiB ff2 fragments | 64 ff1 biases | pad,
// every fragment lane-linear (lane l's 16 bytes at l * 16): the image is copied to LDS by linear 1-KiB LDS-DMA pieces and every
// ds_read_b128 is base + lane * 16 + immediate -- conflict-free, no swizzle, no address arithmetic.
//
// Structure: PRODUCER / CONSUMER waves.  A first version ran the whole chain in one wave per SIMD (512 registers): it was bound by
// instruction ISSUE, not by the matrix pipe -- a lone wave issues one instruction per ~5.5 cycles, and a sub-block needs 525 of
// them (60 MFMAs, 68 fragment reads, 16 erf-GELUs of 17 instructions, waits, DMA) = 2900 cycles against 1920 of MFMA (in-kernel
// stamps: tools/trace_ffn.py; 525 us against 640 us for the three launches).  Now a workgroup is 8 waves = 4 row groups of 32 rows
// x 2 roles, the two roles of a row group on one SIMD (waves w and w + 4):
//   A (waves 0-3)  x rows -> LayerNorm -> B fragments; per sub-block ff1 (40 MFMAs) interleaved with the GEGLU of the previous
//                  sub-block on the VALU; the packed G tile (2 KiB) goes to the partner through a 2-slot LDS ring;
//   B (waves 4-7)  per sub-block ff2 (20 MFMAs) into the 160 output registers, ALL the weight DMA (A's stream carries no memory
//                  instruction at all), and the epilogue (+ bias2 + residual, stores).
// Two instruction streams per SIMD: A's GELUs issue beside B's MFMAs and vice versa, and both fit 256 registers.  One barrier per
// sub-block orders everything:  iteration i:  A: ff1(i) || GEGLU(i-1) -> G(i-1)   B: ff2(i-2); DMA W1(i+1), W2(i-1)   | wait, barrier
// (W1 slot (i+1)&1 was last read by ff1(i-1), W2 slot (i-1)&1 by ff2(i-3), G slot (i-1)&1 by ff2(i-3): all one barrier back.)
// Iterations 0 .. nsb + 1; barrier 0 opens iteration 0 (W1(0) landed), barrier i + 1 closes iteration i <= nsb: nsb + 2 in all.
#include <type_traits>

#include "common.h"
#include "gemm_common.h"
#include "mmgt_hip.h"

namespace {

constexpr int FFC = 320, FF_KS = FFC / 16, FF_NU = FFC / 32;
// weight image per sub-block (61 KiB): [ff1 fragments 40 KiB][ff2 fragments 20 KiB][64 ff1 biases | pad: 1 KiB]
constexpr int FF_W1 = FF_KS * 2 * 1024, FF_W2 = FF_NU * 2 * 1024, FF_B1 = FF_W1 + FF_W2, FF_IMG = 61 * 1024;
// LDS: two ff1 slots | two ff2 slots | two G slots (4 row groups x 2 KiB) | gamma, beta, bias2 | the ff1 biases of ALL sub-blocks
constexpr int FF_L1 = 0, FF_L2 = 2 * FF_W1, FF_LGT = FF_L2 + 2 * FF_W2, FF_GSLOT = 4 * 2048, FF_LG = FF_LGT + 2 * FF_GSLOT,
              FF_LB = FF_LG + 3 * FFC * 4, FF_MAXSB = 64;
static_assert(FF_B1 + 256 <= FF_IMG && FF_LB + FF_MAXSB * 256 <= 160 * 1024, "layout");

__device__ __forceinline__ f32x16 mma32b(s16x8 a, s16x8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0); }

// Exact-erf GELU for two values as two interleaved dependency chains, in two halves.  x Phi(x) = max(x, 0) - |x| Phi(-|x|), and
// log2 Phi(-z) is a smooth function that a degree-5 polynomial in z = min(|x|, 7) follows to 1.1e-6 where z Phi(-z) is largest
// (weighted minimax fit, tools/fit_gelu.py; |gelu error| <= 7.7e-7 in fp32 evaluation, the same class as common.h's gelu_erf_f,
// |error| < 1e-6, which the other kernels use): 5 FMAs + one exp2 instead of 6 FMAs, 4 squarings and a reciprocal -- role A is bound
// by its instruction COUNT (10 vector instructions per value here against 15).  Beyond z = 7, Phi(-z) < 1.3e-12.
#define FF_G0 -1.000055242e+00f
#define FF_G1 -1.150636504e+00f
#define FF_G2 -4.603651887e-01f
#define FF_G3 -5.145699537e-02f
#define FF_G4 6.927462962e-03f
#define FF_G5 -4.497945628e-04f
__device__ __forceinline__ float relu1(float x) {   // one v_max (the builtin form is preceded by a canonicalising v_max)
  float r;
  asm("v_max_f32 %0, 0, %1" : "=v"(r) : "v"(x));
  return r;
}
__device__ __forceinline__ void gelu_poly2(float x0, float x1, float& p0, float& p1) {
  const float z0 = fminf(fabsf(x0), 7.f), z1 = fminf(fabsf(x1), 7.f);
  float a = fmaf(FF_G5, z0, FF_G4), b = fmaf(FF_G5, z1, FF_G4);
  a = fmaf(a, z0, FF_G3); b = fmaf(b, z1, FF_G3);
  a = fmaf(a, z0, FF_G2); b = fmaf(b, z1, FF_G2);
  a = fmaf(a, z0, FF_G1); b = fmaf(b, z1, FF_G1);
  p0 = fmaf(a, z0, FF_G0); p1 = fmaf(b, z1, FF_G0);
}
__device__ __forceinline__ void gelu_finish2(float x0, float x1, float p0, float p1, float h0, float h1, float& o0, float& o1) {
  const float r0 = __builtin_amdgcn_exp2f(p0), r1 = __builtin_amdgcn_exp2f(p1);     // Phi(-|x|)
  o0 = h0 * fmaf(-fabsf(x0), r0, relu1(x0));
  o1 = h1 * fmaf(-fabsf(x1), r1, relu1(x1));
}

__device__ __forceinline__ s16x8 pack8(const float (&v)[8]) {
  union { u32x4 u; s16x8 s; } cv;
  cv.u = (u32x4){pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]), pack_bf16x2(v[4], v[5]), pack_bf16x2(v[6], v[7])};
  return cv.s;
}

#ifndef MMGT_FFN_PACE
#define MMGT_FFN_PACE 0
#endif
__device__ __forceinline__ void mfma_pace() {     // ~24 cycles in which this wave asks nothing of the vector issue port
  __builtin_amdgcn_sched_barrier(0);
  if (MMGT_FFN_PACE == 1) asm volatile("s_nop 7");
  if (MMGT_FFN_PACE == 2) asm volatile("s_nop 7\n\ts_nop 1");
  if (MMGT_FFN_PACE == 3) asm volatile("s_nop 7\n\ts_nop 7");
  if (MMGT_FFN_PACE == 4) asm volatile("s_sleep 1");
  __builtin_amdgcn_sched_barrier(0);
}

__device__ __forceinline__ void lds_barrier() {   // this wave's LDS traffic has completed, then the workgroup barrier (LDS-DMA is NOT
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // drained here: the loader waves wait vmcnt themselves)
  __builtin_amdgcn_s_barrier();
}

// DBG (mmgt_tune("ffn_dbg", v), measurements only): 1 = every weight piece takes the poison offset (nothing is fetched: the
// compute streams alone), 2 = no MFMA / GELU (the weight stream alone), 3 = no GELU, 4 = no ff2 MFMAs; results are garbage.
template <int DBG>
__global__ __launch_bounds__(512, 2)
void ff_fused_kernel(const bf16_t* __restrict__ x, long ldx, const float* __restrict__ gamma, const float* __restrict__ beta,
                     float eps, const char* __restrict__ wimg, int nsb, const float* __restrict__ bias2,
                     const bf16_t* __restrict__ res, long ldr, bf16_t* __restrict__ out, long ldo, int M, unsigned long long* trace) {
  extern __shared__ __attribute__((aligned(1024))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31, hh = lane >> 5;
  const int rg = wid & 3;                         // row group: waves rg (role A) and rg + 4 (role B) share a SIMD
  const long row = (long)blockIdx.x * 128 + rg * 32 + r;
  const long rowc = row < M ? row : M - 1;
  int trace_n = 0;
  auto stamp = [&]() {   // debug (tools/trace_ffn.py): shader-clock stamps of waves 0 (A) and 4 (B) of every workgroup
    if (trace && rg == 0 && lane == 0 && trace_n < 32) trace[((long)blockIdx.x * 2 + (wid >> 2)) * 32 + trace_n++] = __builtin_amdgcn_s_memtime();
  };
  stamp();

  // ---- gamma | beta | bias2 and the ff1 biases of all sub-blocks -> LDS (all 512 threads)
  {
    float* lgb = reinterpret_cast<float*>(smem + FF_LG);
    if (tid < 3 * FFC / 4 && (gamma || tid >= 2 * FFC / 4)) {   // 3 x 80 vectors
      const float* src = tid < FFC / 4 ? gamma + 4 * tid : tid < 2 * FFC / 4 ? beta + 4 * (tid - FFC / 4) : bias2 + 4 * (tid - 2 * FFC / 4);
      *reinterpret_cast<f32x4*>(lgb + 4 * tid) = *reinterpret_cast<const f32x4*>(src);
    }
    for (int v = tid; v < nsb * 16; v += 512)                    // 16 vectors of 4 biases per sub-block, from the image
      *reinterpret_cast<f32x4*>(smem + FF_LB + v * 16) = *reinterpret_cast<const f32x4*>(wimg + (long)(v >> 4) * FF_IMG + FF_B1 + (v & 15) * 16);
  }
  using std::integral_constant;
  constexpr integral_constant<bool, true> T{};
  constexpr integral_constant<bool, false> F{};
  constexpr int PF = 3;                  // fragment reads run PF steps ahead of their MFMAs; sched_barriers pin that order (left alone,
                                         // hipcc reads right in front of each MFMA and waits lgkmcnt(0) every step)
  if (wid < 4) {
    // =========================================================================================== role A: ff1 + GEGLU
    s16x8 xf[FF_KS];                     // the 32 rows as ff1 B fragments: lane (r, hh) holds channels 16 ks + 8 hh .. + 7 of row r
    {
      const bf16_t* xr = x + rowc * ldx + 8 * hh;
#pragma unroll
      for (int ks = 0; ks < FF_KS; ++ks) xf[ks] = *reinterpret_cast<const s16x8*>(xr + 16 * ks);
    }
    __syncthreads();                     // (tables in LDS)
    if (gamma) {   // LayerNorm (exact two-pass statistics in registers, as ln_kernel): y = (x - mean) * rstd * gamma + beta
      const float* lgb = reinterpret_cast<const float*>(smem + FF_LG);
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
    stamp();
    s16x8 fr[PF + 1][2];
    // Iteration i of role A:  S(i) | GEGLU(i - 1) -> packed G tile -> LDS slot (i - 1) & 1, one VALU-only block | M(i) | ff1(i), one
    // MFMA-dense block (two fragment reads and two waits per MFMA pair, nothing else).  Meanwhile role B:  S(i) | ff2(i - 2), 20 dense
    // MFMAs | M(i) | the 15 LDS-DMA pieces of the next weights.  The matrix pipe of the SIMD is handed back and forth: B's MFMAs
    // run under A's GELUs, B's DMA issue (~90 cycles a piece) under A's MFMAs.  Interleaving GELU and MFMAs inside A instead
    // (~10 instructions between MFMAs) stretched every MFMA gap of A to ~50 cycles and B's MFMAs came on top (in-kernel stamps:
    // 2670 ticks per iteration against 1920 of matrix pipe), whatever the instruction count of the GELU was.
    auto glu = [&](int sb, const f32x16& hp, const f32x16& gp) {        // GEGLU of sub-block sb's (hp, gp) -> G slot sb & 1
      char* gs = smem + FF_LGT + (sb & 1) * FF_GSLOT + rg * 2048 + lane * 16;
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        float gv[8];
#pragma unroll
        for (int j = 0; j < 8; j += 2) {
          float p0, p1;
          gelu_poly2(gp[8 * s + j], gp[8 * s + j + 1], p0, p1);
          gelu_finish2(gp[8 * s + j], gp[8 * s + j + 1], p0, p1, hp[8 * s + j], hp[8 * s + j + 1], gv[j], gv[j + 1]);
        }
        *reinterpret_cast<s16x8*>(gs + s * 1024) = pack8(gv);
      }
    };
    auto ff1 = [&](int sb, f32x16& hn, f32x16& gn) {                     // ff1 of sub-block sb from W1 slot sb & 1, bias first
      const char* s1 = smem + FF_L1 + (sb & 1) * FF_W1 + lane * 16;
      const float* bl = reinterpret_cast<const float*>(smem + FF_LB + sb * 256) + 4 * hh;   // register i <-> hidden 4 hh + (i & 3) + 8 (i >> 2)
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) {
        const f32x4 bh = *reinterpret_cast<const f32x4*>(bl + 8 * g4), bg = *reinterpret_cast<const f32x4*>(bl + 32 + 8 * g4);
#pragma unroll
        for (int e = 0; e < 4; ++e) { hn[4 * g4 + e] = bh[e]; gn[4 * g4 + e] = bg[e]; }
      }
#pragma unroll
      for (int i = 0; i < PF; ++i) {
        fr[i][0] = *reinterpret_cast<const s16x8*>(s1 + (2 * i) * 1024);
        fr[i][1] = *reinterpret_cast<const s16x8*>(s1 + (2 * i + 1) * 1024);
      }
#pragma unroll
      for (int ks = 0; ks < FF_KS; ++ks) {
        if (ks + PF < FF_KS) {
          fr[(ks + PF) % (PF + 1)][0] = *reinterpret_cast<const s16x8*>(s1 + (2 * (ks + PF)) * 1024);
          fr[(ks + PF) % (PF + 1)][1] = *reinterpret_cast<const s16x8*>(s1 + (2 * (ks + PF) + 1) * 1024);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (DBG != 2) {
          hn = mma32b(fr[ks % (PF + 1)][0], xf[ks], hn);
          gn = mma32b(fr[ks % (PF + 1)][1], xf[ks], gn);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    };
    f32x16 hA, gA, hB, gB;
    lds_barrier();                                              // S(0)
    stamp();
    lds_barrier();                                              // M(0): W1(0) has landed
    ff1(0, hA, gA);
    auto iter = [&](int i, f32x16& hp, f32x16& gp, f32x16& hn, f32x16& gn) {   // 1 <= i < nsb
      lds_barrier();                                            // S(i)
      if (i < 6) stamp();
      glu(i - 1, hp, gp);
      if (i < 6) stamp();
      lds_barrier();                                            // M(i): W1(i) has landed
      ff1(i, hn, gn);
      if (i < 6) stamp();
    };
    int i = 1;
    for (; i + 1 < nsb; i += 2) {
      iter(i, hA, gA, hB, gB);
      iter(i + 1, hB, gB, hA, gA);
    }
    stamp();
    if (i < nsb) {                                              // nsb even: one more full iteration, the pending tile ends in (hB, gB)
      iter(i, hA, gA, hB, gB);
      lds_barrier();                                            // S(nsb)
      glu(nsb - 1, hB, gB);
    } else {
      lds_barrier();                                            // S(nsb)
      glu(nsb - 1, hA, gA);
    }
    lds_barrier();                                              // M(nsb)
    lds_barrier();                                              // S(nsb + 1): B's last ff2 follows
    stamp();
  } else {
    // =========================================================================================== role B: weight DMA, ff2, epilogue
    const int bw = wid - 4;
    const __amdgpu_buffer_rsrc_t rw = dma_rsrc(wimg);
    const unsigned lane16 = DBG == 1 ? DMA_POISON : (unsigned)lane * 16u;
    // this wave's pieces bw, bw + 4, ... of the ff1 part (40 pieces -> W1 slot) / ff2 part (20 pieces -> W2 slot) of sub-block sb
    auto issue1 = [&](int sb, int i) { blds16(rw, lane16, sb * FF_IMG + (bw + 4 * i) * 1024, smem + FF_L1 + (sb & 1) * FF_W1 + (bw + 4 * i) * 1024); };
    auto issue2 = [&](int sb, int i) { blds16(rw, lane16, sb * FF_IMG + FF_W1 + (bw + 4 * i) * 1024, smem + FF_L2 + (sb & 1) * FF_W2 + (bw + 4 * i) * 1024); };
    __syncthreads();                     // (tables in LDS: the plain loads above are done before the first DMA goes out)
#pragma unroll
    for (int i = 0; i < 10; ++i) issue1(0, i);
    f32x16 oacc[FF_NU];
#pragma unroll
    for (int u = 0; u < FF_NU; ++u) oacc[u] = (f32x16)(0.f);
    // ff2 of sub-block sb from G slot sb & 1 and W2 slot sb & 1: 20 dense MFMAs, two channel tiles per step
    auto ff2 = [&](int sb) {
      const char* s2 = smem + FF_L2 + (sb & 1) * FF_W2 + lane * 16;
      const char* gs = smem + FF_LGT + (sb & 1) * FF_GSLOT + rg * 2048 + lane * 16;
      s16x8 fb[3][4], gb[2];
      auto rd = [&](int st, s16x8 (&f)[4]) {
#pragma unroll
        for (int q = 0; q < 4; ++q) f[q] = *reinterpret_cast<const s16x8*>(s2 + (4 * st + q) * 1024);   // (tile 2 st + (q >> 1), k-step q & 1)
      };
      gb[0] = *reinterpret_cast<const s16x8*>(gs);
      gb[1] = *reinterpret_cast<const s16x8*>(gs + 1024);
      rd(0, fb[0]);
      rd(1, fb[1]);
#pragma unroll
      for (int st = 0; st < FF_NU / 2; ++st) {
        if (st + 2 < FF_NU / 2) rd(st + 2, fb[(st + 2) % 3]);
        __builtin_amdgcn_sched_barrier(0);
        if (DBG != 2 && DBG != 4) {
          // A wave that presents an MFMA to a busy matrix pipe blocks the SIMD's vector issue -- its partner's VALU included (tools/micro/
          // coexec.hip: a VALU-only wave beside an MFMA-only wave takes the SUM of their times; with ~24 cycles of s_nop behind each MFMA the
          // VALU wave disappears under the MFMA wave).  These 20 MFMAs run beside the partner's GELU block: pace them at the pipe's rate.
          const int u = 2 * st;
          oacc[u] = mma32b(fb[st % 3][0], gb[0], oacc[u]); mfma_pace();
          oacc[u + 1] = mma32b(fb[st % 3][2], gb[0], oacc[u + 1]); mfma_pace();
          oacc[u] = mma32b(fb[st % 3][1], gb[1], oacc[u]); mfma_pace();
          oacc[u + 1] = mma32b(fb[st % 3][3], gb[1], oacc[u + 1]); mfma_pace();
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    };
    // second half of iteration i: W2(i - 1) first (needed at S(i + 1)), then W1(i + 1) (needed at M(i + 1)), wait for the former
    auto dma = [&](int i) {
      const bool dv = i - 1 >= 0 && i - 1 < nsb, dw = i + 1 < nsb;
      if (dv) {
#pragma unroll
        for (int q = 0; q < 5; ++q) issue2(i - 1, q);
      }
      if (dw) {
#pragma unroll
        for (int q = 0; q < 10; ++q) issue1(i + 1, q);
        wait_vmcnt<10>();
      } else {
        wait_vmcnt<0>();
      }
    };
    stamp();
    lds_barrier();                                              // S(0)
    for (int i = 0; i <= nsb; ++i) {
      if (i >= 2 && i < 8) stamp();
      if (i >= 2) ff2(i - 2);
      if (i >= 2 && i < 8) stamp();
      wait_vmcnt<0>();                                          // W1(i) (issued one iteration ago) has landed
      lds_barrier();                                            // M(i)
      if (i >= 2 && i < 8) stamp();
      dma(i);
      if (i >= 2 && i < 8) stamp();
      lds_barrier();                                            // S(i + 1)
    }
    ff2(nsb - 1);
    stamp();
    // ---- epilogue: + b2 + residual, bf16, 16-byte stores.  Register group k (registers 4 k .. 4 k + 3) of tile u is channels
    // 32 u + 8 k + 4 hh + (0..3); v_permlane32_swap of groups (k, k + 1) gives lane hh = 0 channels 32 u + 8 k .. + 7 and lane hh = 1
    // channels 32 u + 8 k + 8 .. + 15 (cdna_hip_programming.md T21).  The stores go through a buffer resource sized to the M valid rows:
    // rows beyond M are dropped by the range check instead of a branch per store.
    {
      const bf16_t* rr = res + rowc * ldr + 8 * hh;
      const __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc(out, 0, (int)((long)M * ldo * 2), 0x00020000);
      const unsigned obase = (unsigned)(row * ldo + 8 * hh) * 2u;          // (rows >= M: beyond num_records -> dropped)
      const float* lb2 = reinterpret_cast<const float*>(smem + FF_LG) + 2 * FFC + 8 * hh;
#pragma unroll
      for (int half = 0; half < 2; ++half) {                               // the residual vectors of 5 tiles at a time (40 registers)
        u32x4 rv[FF_NU];
#pragma unroll
        for (int q = 0; q < FF_NU; ++q) rv[q] = *reinterpret_cast<const u32x4*>(rr + 16 * (FF_NU * half + q));   // channels 16 q' + 8 hh .. + 7
#pragma unroll
        for (int uu = 0; uu < FF_NU / 2; ++uu)
#pragma unroll
          for (int k = 0; k < 4; k += 2) {
            const int u = (FF_NU / 2) * half + uu;
            const int c = 32 * u + 8 * k;          // + 8 hh in the bases
            union { u32x4 q; bf16_t e[8]; } r8;
            r8.q = rv[2 * uu + k / 2];
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
    stamp();
  }
}

int g_ffn_dbg = 0;
unsigned long long* g_ffn_trace = nullptr;

}  // namespace

void mmgt_ffn_set_dbg(int v) { g_ffn_dbg = v; }
// Debug (tools/trace_ffn.py): device buffer of u64 [workgroups][2 roles][32] for the shader-clock stamps of waves 0 and 4; NULL = off.
extern "C" void mmgt_ffn_set_trace(void* p) { g_ffn_trace = reinterpret_cast<unsigned long long*>(p); }

extern "C" int mmgt_ff_fused_image_bytes(int C, int inner) {
  if (C != FFC || inner < 64 || inner % 32 || inner / 32 > FF_MAXSB) return -1;
  return (inner / 32) * FF_IMG;
}

extern "C" int mmgt_ff_fused(const void* x, long ldx, const float* ln_gamma, const float* ln_beta, float eps, const void* wimg,
                             const float* bias2, const void* residual, long ldr, void* out, long ldo, int M, int C, int inner,
                             int dtype, void* stream) {
  MMGT_CHECK(x && wimg && bias2 && residual && out, "ff_fused: null pointer");
  MMGT_CHECK(dtype == MMGT_BF16, "ff_fused: bf16 only (the fp32-I/O mode runs LayerNorm / GEMM / GEMM)");
  MMGT_CHECK(mmgt_ff_fused_image_bytes(C, inner) > 0, "ff_fused: built for %d channels (got %d) and inner = 64 .. %d in steps of 32 (got %d)",
             FFC, C, 32 * FF_MAXSB, inner);
  MMGT_CHECK((ln_gamma != nullptr) == (ln_beta != nullptr), "ff_fused: gamma / beta must come together");
  MMGT_CHECK((long)M * ldo * 2 < (1l << 31), "ff_fused: output beyond the 2 GiB range of a buffer resource (split the rows)");
  MMGT_CHECK(M > 0 && ldx >= C && ldr >= C && ldo >= C && ldx % 8 == 0 && ldr % 8 == 0 && ldo % 8 == 0, "ff_fused: bad M=%d or row strides", M);
  MMGT_CHECK((((uintptr_t)x | (uintptr_t)residual | (uintptr_t)out | (uintptr_t)wimg | (uintptr_t)bias2) & 15) == 0 &&
                 (!ln_gamma || (((uintptr_t)ln_gamma | (uintptr_t)ln_beta) & 15) == 0),
             "ff_fused: pointers must be 16-byte aligned");
  const size_t lds = FF_LB + (size_t)(inner / 32) * 256;
  auto kern = g_ffn_dbg == 1 ? ff_fused_kernel<1> : g_ffn_dbg == 2 ? ff_fused_kernel<2> : g_ffn_dbg == 3 ? ff_fused_kernel<3> : g_ffn_dbg == 4 ? ff_fused_kernel<4> : ff_fused_kernel<0>;
  static bool attr[5] = {false, false, false, false, false};
  if (!attr[g_ffn_dbg]) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, FF_LB + FF_MAXSB * 256) != hipSuccess) {
      mmgt_set_error("ff_fused: cannot reserve %d bytes of LDS", FF_LB + FF_MAXSB * 256);
      return 2;
    }
    attr[g_ffn_dbg] = true;
  }
  const unsigned grid = (unsigned)((M + 127) / 128);
  hipLaunchKernelGGL(kern, dim3(grid), dim3(512), lds, (hipStream_t)stream, (const bf16_t*)x, ldx, ln_gamma, ln_beta, eps,
                     (const char*)wimg, inner / 32, bias2, (const bf16_t*)residual, ldr, (bf16_t*)out, ldo, M, g_ffn_trace);
  MMGT_LAUNCH_CHECK();
  return 0;
}
