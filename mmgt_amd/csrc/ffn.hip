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
#include <type_traits>

#include "common.h"
#include "gemm_common.h"
#include "mmgt_hip.h"

namespace {

constexpr int FFC = 320, FF_KS = FFC / 16, FF_NU = FFC / 32;
// weight image per sub-block (61 KiB): [ff1 fragments 40 KiB][ff2 fragments 20 KiB][64 ff1 biases | pad: 1 KiB]
constexpr int FF_W1 = FF_KS * 2 * 1024, FF_W2 = FF_NU * 2 * 1024, FF_B1 = FF_W1 + FF_W2, FF_IMG = 61 * 1024;
constexpr int FF_P1 = FF_W1 / 4096, FF_P2 = FF_W2 / 4096;      // LDS-DMA pieces per wave and sub-block: 10 (ff1 part), 5 (ff2 part)
// LDS: two ff1 slots | two ff2 slots | gamma, beta, bias2 | the ff1 biases of ALL sub-blocks (256 B each, copied once)
constexpr int FF_L1 = 0, FF_L2 = 2 * FF_W1, FF_LG = FF_L2 + 2 * FF_W2, FF_LB = FF_LG + 3 * FFC * 4, FF_MAXSB = 128;
static_assert(FF_B1 + 256 <= FF_IMG && FF_LB + FF_MAXSB * 256 <= 160 * 1024, "layout");

__device__ __forceinline__ f32x16 mma32b(s16x8 a, s16x8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0); }

// gelu_erf_f of common.h in two halves (same arithmetic): p = poly(|x|);  x Phi(x) = max(x, 0) - |x| / (2 p^16)'s reciprocal form.
// v_med3 instead of fmaxf: no canonicalising v_max in front of it.
__device__ __forceinline__ float gelu_poly(float x) {
  const float z = fabsf(x);
  float p = fmaf(MMGT_GELU_C6, z, MMGT_GELU_C5);
  p = fmaf(p, z, MMGT_GELU_C4);
  p = fmaf(p, z, MMGT_GELU_C3);
  p = fmaf(p, z, MMGT_GELU_C2);
  p = fmaf(p, z, MMGT_GELU_C1);
  return fmaf(p, z, MMGT_GELU_K);
}
__device__ __forceinline__ float gelu_finish(float x, float p) {
  p *= p; p *= p; p *= p; p *= p;
  const float h = __builtin_amdgcn_rcpf(p);
  return fmaf(-fabsf(x), h, __builtin_amdgcn_fmed3f(x, 0.f, __builtin_inff()));
}

// the same for two values, statements interleaved (two independent dependency chains for the in-order issue)
__device__ __forceinline__ void gelu_poly2(float x0, float x1, float& p0, float& p1) {
  const float z0 = fabsf(x0), z1 = fabsf(x1);
  float a = fmaf(MMGT_GELU_C6, z0, MMGT_GELU_C5), b = fmaf(MMGT_GELU_C6, z1, MMGT_GELU_C5);
  a = fmaf(a, z0, MMGT_GELU_C4); b = fmaf(b, z1, MMGT_GELU_C4);
  a = fmaf(a, z0, MMGT_GELU_C3); b = fmaf(b, z1, MMGT_GELU_C3);
  a = fmaf(a, z0, MMGT_GELU_C2); b = fmaf(b, z1, MMGT_GELU_C2);
  a = fmaf(a, z0, MMGT_GELU_C1); b = fmaf(b, z1, MMGT_GELU_C1);
  p0 = fmaf(a, z0, MMGT_GELU_K); p1 = fmaf(b, z1, MMGT_GELU_K);
}
__device__ __forceinline__ void gelu_finish2(float x0, float x1, float p0, float p1, float h0, float h1, float& o0, float& o1) {
  p0 *= p0; p1 *= p1; p0 *= p0; p1 *= p1; p0 *= p0; p1 *= p1; p0 *= p0; p1 *= p1;
  const float r0 = __builtin_amdgcn_rcpf(p0), r1 = __builtin_amdgcn_rcpf(p1);
  const float m0 = __builtin_amdgcn_fmed3f(x0, 0.f, __builtin_inff()), m1 = __builtin_amdgcn_fmed3f(x1, 0.f, __builtin_inff());
  o0 = h0 * fmaf(-fabsf(x0), r0, m0);
  o1 = h1 * fmaf(-fabsf(x1), r1, m1);
}

__device__ __forceinline__ s16x8 pack8(const float (&v)[8]) {
  union { u32x4 u; s16x8 s; } cv;
  cv.u = (u32x4){pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]), pack_bf16x2(v[4], v[5]), pack_bf16x2(v[6], v[7])};
  return cv.s;
}

// Schedule of a wave (sub-blocks j = 0 .. nsb - 1 of 32 hidden channels; one wave per SIMD, so everything that can overlap has to
// be interleaved in this one instruction stream):
//   iteration j:  1. wait W1(j+1), barrier      2. phase A: ff1(j+1) MFMAs || GEGLU(j) on the VALU || DMA of W2(j+1)
//                 3. wait W2(j), barrier        4. phase B: ff2(j) MFMAs || DMA of W1(j+3)
// i.e. the 40 MFMAs of the NEXT sub-block's ff1 cover the 16 erf-GELUs of this one (the GEGLU'd tile is the B operand of ff2, so
// without the skew the matrix pipe idles through ~1000 cycles of VALU per sub-block: v1 of this kernel, 610 us against 650 for the
// three launches).  The ff1 and ff2 fragments live in separate 2-slot rings because they are freed at different times: the ff1
// slot of sub-block j+1 after phase A of iteration j, the ff2 slot of sub-block j after phase B.  DMA order W1(0) W1(1) | W2(0) W1(2)
// | W2(1) W1(3) | ...: the counted waits leave the two younger groups in flight (10 + 5 pieces per wave).  Barrier 1 also frees the ff2 slot that W2(j+1) overwrites (every wave is
// through phase B of j-1), barrier 3 the ff1 slot that W1(j+3) overwrites.
// DBG (mmgt_tune("ffn_dbg", v), measurements only): 1 = every weight piece takes the poison offset (nothing is fetched: the
// compute stream alone), 2 = the MFMA / GELU phases are skipped (the weight stream alone), 3 = no weight DMA instruction at all;
// results are garbage in all three.
template <int DBG>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1)))
void ff_fused_kernel(const bf16_t* __restrict__ x, long ldx, const float* __restrict__ gamma, const float* __restrict__ beta,
                     float eps, const char* __restrict__ wimg, int nsb, const float* __restrict__ bias2,
                     const bf16_t* __restrict__ res, long ldr, bf16_t* __restrict__ out, long ldo, int M, unsigned long long* trace) {
  extern __shared__ __attribute__((aligned(1024))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31, hh = lane >> 5;
  const long row = (long)blockIdx.x * 128 + wid * 32 + r;
  const long rowc = row < M ? row : M - 1;

  int trace_n = 0;
  auto stamp = [&]() {   // debug (tools/trace_ffn.py): shader-clock stamps of wave 0 of every workgroup at its phase boundaries
    if (trace && wid == 0 && lane == 0 && trace_n < 64) trace[(long)blockIdx.x * 64 + trace_n++] = __builtin_amdgcn_s_memtime();
  };
  stamp();
  const __amdgpu_buffer_rsrc_t rw = dma_rsrc(wimg);
  const unsigned lane16 = (unsigned)lane * 16u;
  // piece i (0 .. 9) of the wave's share of W1(sb) / piece i (0 .. 4) of W2(sb): pieces wid + 4 i of the part.  Sub-blocks beyond the
  // image are "loaded" too, with the poison offset (the range check returns zeros, nothing is fetched): every iteration then issues
  // the same number of pieces and the counted waits are compile-time constants all the way to the last sub-block.
  auto issue1 = [&](int sb, int i) {
    if (DBG == 3) return;
    blds16(rw, (DBG != 1 && sb < nsb) ? lane16 : DMA_POISON, sb * FF_IMG + (wid + 4 * i) * 1024, smem + FF_L1 + (sb & 1) * FF_W1 + (wid + 4 * i) * 1024);
  };
  auto issue2 = [&](int sb, int i) {
    if (DBG == 3) return;
    blds16(rw, (DBG != 1 && sb < nsb) ? lane16 : DMA_POISON, sb * FF_IMG + FF_W1 + (wid + 4 * i) * 1024, smem + FF_L2 + (sb & 1) * FF_W2 + (wid + 4 * i) * 1024);
  };

  // ---- the wave's 32 rows as ff1 B fragments: lane (r, hh) holds channels 16 ks + 8 hh .. + 7 of row r
  s16x8 xf[FF_KS];
  {
    const bf16_t* xr = x + rowc * ldx + 8 * hh;
#pragma unroll
    for (int ks = 0; ks < FF_KS; ++ks) xf[ks] = *reinterpret_cast<const s16x8*>(xr + 16 * ks);
    // gamma | beta | bias2 and the ff1 biases go through LDS: with an LDS-DMA in flight hipcc waits vmcnt(0) for every plain
    // global load, which serialised 40 L2 round trips here; so the weight DMA also starts only behind these loads
    float* lgb = reinterpret_cast<float*>(smem + FF_LG);
    if (tid < 3 * FFC / 4 && (gamma || tid >= 2 * FFC / 4)) {   // 3 x 80 vectors
      const float* src = tid < FFC / 4 ? gamma + 4 * tid : tid < 2 * FFC / 4 ? beta + 4 * (tid - FFC / 4) : bias2 + 4 * (tid - 2 * FFC / 4);
      *reinterpret_cast<f32x4*>(lgb + 4 * tid) = *reinterpret_cast<const f32x4*>(src);
    }
    for (int v = tid; v < nsb * 16; v += 256)                    // 16 vectors of 4 biases per sub-block, from the image
      *reinterpret_cast<f32x4*>(smem + FF_LB + v * 16) = *reinterpret_cast<const f32x4*>(wimg + (long)(v >> 4) * FF_IMG + FF_B1 + (v & 15) * 16);
    __syncthreads();
#pragma unroll
    for (int i = 0; i < FF_P1; ++i) issue1(0, i);
#pragma unroll
    for (int i = 0; i < FF_P1; ++i) issue1(1, i);
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
  stamp();

  constexpr int PF = 3;                  // fragment reads run PF steps ahead of their MFMAs (ring of PF + 1 register pairs); the
  s16x8 fr[PF + 1][2];                   // sched_barriers pin that order -- left alone, hipcc reads right in front of each MFMA pair
  using std::integral_constant;
  // Phase A.  FF1: ff1 of sub-block `sbn` into (hn, gn), which start from the ff1 bias, with the 5 DMA pieces of W2(sbn) on steps
  // 0, 2, .., 8; GLU: GEGLU of (hp, gp) into gb, one value per k-step.
  auto phaseA = [&](int sbn, auto FF1c, auto GLUc, f32x16& hn, f32x16& gn, const f32x16& hp, const f32x16& gp, s16x8 (&gb)[2]) {
    constexpr bool FF1 = decltype(FF1c)::value, GLU = decltype(GLUc)::value;
    const char* s1 = smem + FF_L1 + (sbn & 1) * FF_W1 + lane * 16;
    if constexpr (FF1) {
      // accumulators start from the ff1 bias (register i <-> hidden 4 hh + (i & 3) + 8 (i >> 2))
      const float* bl = reinterpret_cast<const float*>(smem + FF_LB + sbn * 256) + 4 * hh;
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
    }
    // GEGLU of the pending tile, two values per PAIR of k-steps as two independent chains (a lone wave issues a dependent VALU
    // chain at ~7 cycles per instruction, two interleaved chains at 4): the polynomial of both values rides behind the MFMAs of
    // the even step, the squarings / reciprocal / product behind those of the odd step.  Each half is pinned in place by an opaque
    // use (LLVM otherwise sinks the whole GELU to its consumer behind the barrier).
    float gv[16], pz[2];
#pragma unroll
    for (int ks = 0; ks < FF_KS; ++ks) {
      if constexpr (FF1) {
        if (ks + PF < FF_KS) {
          fr[(ks + PF) % (PF + 1)][0] = *reinterpret_cast<const s16x8*>(s1 + (2 * (ks + PF)) * 1024);
          fr[(ks + PF) % (PF + 1)][1] = *reinterpret_cast<const s16x8*>(s1 + (2 * (ks + PF) + 1) * 1024);
        }
        if ((ks & 1) == 0 && ks / 2 < FF_P2) issue2(sbn, ks / 2);
        __builtin_amdgcn_sched_barrier(0);
        if (DBG != 2) {
          hn = mma32b(fr[ks % (PF + 1)][0], xf[ks], hn);
          gn = mma32b(fr[ks % (PF + 1)][1], xf[ks], gn);
        }
      }
      if (GLU && ks < 16 && DBG != 2) {
        const int e0 = ks & ~1, e1 = e0 + 1;
        if ((ks & 1) == 0) {
          gelu_poly2(gp[e0], gp[e1], pz[0], pz[1]);
          asm volatile("" : "+v"(pz[0]), "+v"(pz[1]));
        } else {
          gelu_finish2(gp[e0], gp[e1], pz[0], pz[1], hp[e0], hp[e1], gv[e0], gv[e1]);
          asm volatile("" : "+v"(gv[e0]), "+v"(gv[e1]));
        }
      }
      if constexpr (FF1) {
        // an in-order wave stalls at the second MFMA until the matrix pipe takes it (32 cycles behind the first), and everything
        // behind it with it: half of the step's VALU goes BETWEEN the two MFMAs
        if (GLU && ks < 16 && DBG != 2) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x002, 7, 0);
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x002, 12, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    if constexpr (GLU && DBG != 2) {
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = gv[8 * s + j];
        gb[s] = pack8(v);
      }
    }
  };
  // Phase B.  ff2 of sub-block sb from gb, two channel tiles per step in the order (u, s0) (u+1, s0) (u, s1) (u+1, s1): the two
  // k-steps of a tile accumulate into the same registers, back to back they would run at the MFMA's latency, not its issue rate.
  // The 10 DMA pieces of W1(sb + 3) ride on the 5 steps (not behind the last sub-block).
  auto phaseB = [&](int sb, auto DMAc, const s16x8 (&gb)[2]) {
    const char* s2 = smem + FF_L2 + (sb & 1) * FF_W2 + lane * 16;
    s16x8 fb[3][4];                        // ring of 3 steps x (2 tiles x 2 k-steps); reads run 2 steps ahead
    auto rd = [&](int st, s16x8 (&f)[4]) {
#pragma unroll
      for (int q = 0; q < 4; ++q) f[q] = *reinterpret_cast<const s16x8*>(s2 + (4 * st + q) * 1024);   // (tile 2 st + (q >> 1), k-step q & 1)
    };
    rd(0, fb[0]);
    rd(1, fb[1]);
#pragma unroll
    for (int st = 0; st < FF_NU / 2; ++st) {
      if (st + 2 < FF_NU / 2) rd(st + 2, fb[(st + 2) % 3]);
      if constexpr (decltype(DMAc)::value) { issue1(sb + 3, 2 * st); issue1(sb + 3, 2 * st + 1); }
      __builtin_amdgcn_sched_barrier(0);
      if (DBG != 2) {
        const int u = 2 * st;
        oacc[u] = mma32b(fb[st % 3][0], gb[0], oacc[u]);
        oacc[u + 1] = mma32b(fb[st % 3][2], gb[0], oacc[u + 1]);
        oacc[u] = mma32b(fb[st % 3][1], gb[1], oacc[u]);
        oacc[u + 1] = mma32b(fb[st % 3][3], gb[1], oacc[u + 1]);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  constexpr integral_constant<bool, true> T{};
  constexpr integral_constant<bool, false> F{};

  f32x16 hA, gA, hB, gB;
  s16x8 gb[2] = {};
  // iteration -1: ff1(0) alone (W2(0) goes out with it), then W1(2)
  wait_vmcnt<FF_P1>();                       // W1(0) has landed; W1(1) in flight
  __builtin_amdgcn_s_barrier();
  phaseA(0, T, F, hA, gA, hA, gA, gb);
  __builtin_amdgcn_s_barrier();              // every wave is through ff1(0): its slot takes W1(2)
#pragma unroll
  for (int i = 0; i < FF_P1; ++i) issue1(2, i);
  // iteration j < nsb - 1:  W1(j+1) landed (younger: W2(j), W1(j+2)) | A: ff1(j+1) || GEGLU(j) || W2(j+1) out | W2(j) landed (younger:
  // W1(j+2), W2(j+1)) | B: ff2(j) || W1(j+3) out.       last iteration: GEGLU, drain, ff2.
  auto iteration = [&](int j, f32x16& hp, f32x16& gp, f32x16& hn, f32x16& gn) {
    wait_vmcnt<FF_P1 + FF_P2>();
    __builtin_amdgcn_s_barrier();
    if (j < 8) stamp();
    phaseA(j + 1, T, T, hn, gn, hp, gp, gb);
    if (j < 8) stamp();
    wait_vmcnt<FF_P1 + FF_P2>();
    __builtin_amdgcn_s_barrier();
    if (j < 8) stamp();
    phaseB(j, T, gb);
    if (j < 8) stamp();
  };
  auto last_iteration = [&](int j, f32x16& hp, f32x16& gp) {
    phaseA(j + 1, F, T, hp, gp, hp, gp, gb);
    wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();
    phaseB(j, F, gb);
  };
  int j = 0;
  for (; j + 2 < nsb; j += 2) {
    iteration(j, hA, gA, hB, gB);
    iteration(j + 1, hB, gB, hA, gA);
  }
  if (nsb - j == 2) {
    iteration(j, hA, gA, hB, gB);
    last_iteration(j + 1, hB, gB);
  } else {
    last_iteration(j, hA, gA);
  }

  stamp();
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
    const float* lb2 = reinterpret_cast<const float*>(smem + FF_LG) + 2 * FFC + 8 * hh;
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
  stamp();
}

int g_ffn_dbg = 0;
unsigned long long* g_ffn_trace = nullptr;

}  // namespace

void mmgt_ffn_set_dbg(int v) { g_ffn_dbg = v; }
// Debug (tools/trace_ffn.py): device buffer of u64 [workgroups][64] for the shader-clock stamps of wave 0; NULL switches them off.
extern "C" void mmgt_ffn_set_trace(void* p) { g_ffn_trace = reinterpret_cast<unsigned long long*>(p); }

extern "C" int mmgt_ff_fused_image_bytes(int C, int inner) {
  if (C != FFC || inner <= 0 || inner % 32) return -1;
  return (inner / 32) * FF_IMG;
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
  MMGT_CHECK(inner / 32 <= FF_MAXSB, "ff_fused: inner %d beyond %d", inner, 32 * FF_MAXSB);
  const size_t lds = FF_LB + (size_t)(inner / 32) * 256;
  auto kern = g_ffn_dbg == 1 ? ff_fused_kernel<1> : g_ffn_dbg == 2 ? ff_fused_kernel<2> : g_ffn_dbg == 3 ? ff_fused_kernel<3> : ff_fused_kernel<0>;
  static bool attr[4] = {false, false, false, false};
  if (!attr[g_ffn_dbg]) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, FF_LB + FF_MAXSB * 256) != hipSuccess) {
      mmgt_set_error("ff_fused: cannot reserve %d bytes of LDS", FF_LB + FF_MAXSB * 256);
      return 2;
    }
    attr[g_ffn_dbg] = true;
  }
  const unsigned grid = (unsigned)((M + 127) / 128);
  hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, (hipStream_t)stream, (const bf16_t*)x, ldx, ln_gamma, ln_beta, eps,
                     (const char*)wimg, inner / 32, bias2, (const bf16_t*)residual, ldr, (bf16_t*)out, ldo, M, g_ffn_trace);
  MMGT_LAUNCH_CHECK();
  return 0;
}
