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
#include "ln_frag.h"
#include "mmgt_hip.h"

namespace {

constexpr int FFC = 320, FF_KS = FFC / 16, FF_NU = FFC / 32;
constexpr int FF_P1 = 10, FF_P2 = 5;      // LDS-DMA pieces per wave and sub-block in the single-role kernel (4 waves): ff1 part, ff2 part
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

// DBG (mmgt_tune("ffn_dbg", v) of the -DMMGT_ABLATE build, measurements only): 1 = every weight piece takes the poison offset (nothing is fetched: the
// compute streams alone), 2 = no MFMA / GELU (the weight stream alone); results are garbage.

// ===================================================================================================================
// Single-role kernel: one wave per SIMD (4 waves = 128 rows per workgroup, up to 512 registers) runs the whole
// chain, and everything that has to overlap is interleaved in that one instruction stream at the granularity of ONE MFMA.
//
// What bounds it (in-kernel stamps, tools/trace_ffn.py): a lone wave issues ONE instruction per ~5.5 cycles -- of any kind: MFMA, VALU,
// ds_read, s_waitcnt and s_nop all cost an issue slot.  A sub-block of 32 hidden channels needs 60 MFMAs = 1920 matrix-pipe cycles, so
// the stream may carry at most ~5 instructions per MFMA; the first cut of this kernel carried 7.7 (460 per sub-block: 3450 ticks per
// iteration, no better than the two-role kernel).  This version spends the budget as follows, per sub-block:
//   60 MFMA | 60 fragment reads, ONE wait per four (the four fragments of the next MFMA group are requested right behind the wait for
//   the current group) | 15 LDS-DMA pieces, contiguous per wave, so four pieces share one M0 / soffset setting (instruction offsets
//   0 .. 3072) | GEGLU of 16 values in 160 VALU: the degree-5 polynomial and the final x Phi(x) as v_pk_fma_f32 on value PAIRS, the
//   ff1 bias added here (v_pk_add_f32) instead of preloading 32 accumulator registers (the first MFMA of a tile takes C = 0) | 8 bias
//   reads | 4 waits + 2 barriers.
// Schedule:  iteration j:  phase A: 40 MFMAs of ff1(j+1), behind each ~2.7 VALU of the second part of GEGLU(j), DMA of W2(j+1);
//                          phase B: 20 MFMAs of ff2(j), behind each ~2.7 VALU of the first part of GEGLU(j+1), DMA of W1(j+3).
// The hand-over between the phases sits at the START of a phase's last MFMA group: wait for this wave's LDS reads and for the DMA of
// the next phase's weights, barrier, request the next phase's first fragments -- they arrive behind the last four MFMAs, so neither
// phase starts with an empty pipe.  That barrier also frees the slot the next phase's DMA overwrites (every wave has completed its
// reads of it).
struct GluChain { f32x2 x, z, a; };
// GEGLU micro-steps of a value pair and their VALU instruction counts; a tile is 8 pairs = 160 instructions, dealt out evenly (by
// count) over the 60 MFMA slots of ff2 (20) and ff1 (40)
constexpr int FF_MS = 13, FF_MSW[FF_MS] = {2, 1, 2, 1, 1, 1, 1, 1, 2, 2, 1, 2, 3}, FF_PAIRW = 20, FF_TILEW = 8 * FF_PAIRW, FF_SLOTS = 60;
constexpr int ff_ms_weight_before(int m) {
  int w = (m / FF_MS) * FF_PAIRW;
  for (int k = 0; k < m % FF_MS; ++k) w += FF_MSW[k];
  return w;
}
constexpr int ff_ms_first(int slot) {     // first micro-step of MFMA slot `slot` (0 .. 60): the first one at or beyond the slot's share
  const int target = slot * FF_TILEW / FF_SLOTS;
  int m = 0;
  while (m < 8 * FF_MS && ff_ms_weight_before(m) < target) ++m;
  return m;
}
static_assert(ff_ms_first(0) == 0 && ff_ms_first(FF_SLOTS) == 8 * FF_MS, "micro-step schedule");

// PO: the transformer block's proj_out rides on the end -- out = res2 + bias_po + Wpo . bf16(hidden), hidden = res + b2 + FeedForward(...)
// as before but never stored: packed to bf16 it IS the B operand of the proj_out MFMAs (the accumulator registers' k order, matched by
// the weight image of packing.pack_ff_proj_out), whose 10 x 20 KiB weight tiles stream through the ff1 ring, two tiles (two
// independent accumulation chains) at a time; the 160 output registers are reused for the second GEMM.
template <int DBG, bool TRACE, bool PO>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1)))
void ff_fused1_kernel(const bf16_t* __restrict__ x, long ldx, const float* __restrict__ gamma, const float* __restrict__ beta,
                      float eps, const char* __restrict__ wimg, int nsb, const float* __restrict__ bias2,
                      const bf16_t* __restrict__ res, long ldr, bf16_t* __restrict__ out, long ldo, int M, unsigned long long* trace,
                      const char* __restrict__ wpo, const float* __restrict__ bias_po, const bf16_t* __restrict__ res2, long ldr2) {
  extern __shared__ __attribute__((aligned(1024))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31, hh = lane >> 5;
  const long row = (long)blockIdx.x * 128 + wid * 32 + r;
  const long rowc = row < M ? row : M - 1;
  int trace_n = 0;
  auto stamp = [&]() {   // TRACE build (tools/trace_ffn.py): shader-clock stamps of wave 0 of every workgroup at its phase boundaries
    if constexpr (TRACE) {
      if (trace && wid == 0 && lane == 0 && trace_n < 32) trace[(long)blockIdx.x * 64 + trace_n++] = __builtin_amdgcn_s_memtime();
    }
  };
  stamp();
  const __amdgpu_buffer_rsrc_t rw = dma_rsrc(wimg);
  const unsigned lane16 = (unsigned)lane * 16u;
  // This wave's share of W1(sb) is the 10 contiguous 1-KiB pieces 10 wid .. 10 wid + 9 of the ff1 part, of W2(sb) the 5 pieces 5 wid ..
  // 5 wid + 4 of the ff2 part.  Sub-blocks beyond the image are "loaded" too, with the poison offset (the range check returns zeros,
  // nothing is fetched): every iteration issues the same number of pieces and the counted waits are compile-time constants.
  auto piece = [&](auto Ic, unsigned voff, int src, char* dst, const __amdgpu_buffer_rsrc_t rs) {     // piece i of a run: group i / 4 (one M0 / soffset), instruction offset 1024 (i % 4)
    constexpr int i = decltype(Ic)::value;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(dst + (i / 4) * 4096), 16, (int)voff,
                                             src + (i / 4) * 4096, (i % 4) * 1024, 0);
  };
  auto issue1 = [&](int sb, auto Ic) {
    piece(Ic, (DBG != 1 && sb < nsb) ? lane16 : DMA_POISON, sb * FF_IMG + wid * (FF_P1 * 1024), smem + FF_L1 + (sb & 1) * FF_W1 + wid * (FF_P1 * 1024), rw);
  };
  auto issue2 = [&](int sb, auto Ic) {
    piece(Ic, (DBG != 1 && sb < nsb) ? lane16 : DMA_POISON, sb * FF_IMG + FF_W1 + wid * (FF_P2 * 1024), smem + FF_L2 + (sb & 1) * FF_W2 + wid * (FF_P2 * 1024), rw);
  };
  using std::integral_constant;
  constexpr integral_constant<bool, true> T{};
  constexpr integral_constant<bool, false> F{};
  auto for_range = [](auto LOc, auto HIc, auto&& fn) {            // fn(integral_constant<int, i>) for i in [lo, hi)
    constexpr int lo = decltype(LOc)::value, hi = decltype(HIc)::value;
    [&]<int... I>(std::integer_sequence<int, I...>) { (fn(integral_constant<int, lo + I>{}), ...); }(std::make_integer_sequence<int, (hi > lo ? hi - lo : 0)>{});
  };
#define FF_IC(v) integral_constant<int, (v)>{}

  // ---- prologue: everything the workgroup needs from memory is requested up front, in ONE latency -- the first two ff1 weight
  // blocks (LDS-DMA), the wave's 32 rows, gamma | beta | bias2 and the ff1 biases of all sub-blocks (into registers) -- and only then
  // used (hipcc waits vmcnt(0) at the first use of a plain load while an LDS-DMA is in flight: exactly what is wanted here; rows ->
  // tables (a load-store loop) -> barrier -> DMA paid the memory latency several times in a row).
  for_range(FF_IC(0), FF_IC(FF_P1), [&](auto i) { issue1(0, i); });
  for_range(FF_IC(0), FF_IC(FF_P1), [&](auto i) { issue1(1, i); });
  // the wave's 32 rows as ff1 B fragments: lane (r, hh) holds channels 16 ks + 8 hh .. + 7 of row r
  s16x8 xf[FF_KS];
  {
    const bf16_t* xr = x + rowc * ldx + 8 * hh;
#pragma unroll
    for (int ks = 0; ks < FF_KS; ++ks) xf[ks] = *reinterpret_cast<const s16x8*>(xr + 16 * ks);
    float* lgb = reinterpret_cast<float*>(smem + FF_LG);
    const bool has_tab = tid < 3 * FFC / 4 && (gamma || tid >= 2 * FFC / 4);     // 3 x 80 vectors
    f32x4 tv = (f32x4)(0.f), bv[FF_MAXSB * 16 / 256];
    if (has_tab) tv = *reinterpret_cast<const f32x4*>(tid < FFC / 4 ? gamma + 4 * tid : tid < 2 * FFC / 4 ? beta + 4 * (tid - FFC / 4) : bias2 + 4 * (tid - 2 * FFC / 4));
#pragma unroll
    for (int i = 0; i < FF_MAXSB * 16 / 256; ++i) {              // 16 vectors of 4 ff1 biases per sub-block, from the image
      const int v = tid + 256 * i;
      if (v < nsb * 16) bv[i] = *reinterpret_cast<const f32x4*>(wimg + (long)(v >> 4) * FF_IMG + FF_B1 + (v & 15) * 16);
    }
    f32x4 pv = (f32x4)(0.f);
    if (PO && tid < FFC / 4) pv = *reinterpret_cast<const f32x4*>(bias_po + 4 * tid);
    if (has_tab) *reinterpret_cast<f32x4*>(lgb + 4 * tid) = tv;
    if (PO && tid < FFC / 4) *reinterpret_cast<f32x4*>(smem + FF_LGT + 16 * tid) = pv;      // (the G slots of the two-role kernel: unused here)
#pragma unroll
    for (int i = 0; i < FF_MAXSB * 16 / 256; ++i) {
      const int v = tid + 256 * i;
      if (v < nsb * 16) *reinterpret_cast<f32x4*>(smem + FF_LB + v * 16) = bv[i];
    }
    __syncthreads();
    if (gamma) layernorm_fragments(xf, reinterpret_cast<const float*>(smem + FF_LG), hh, eps);
  }
  f32x16 oacc[FF_NU];
#pragma unroll
  for (int u = 0; u < FF_NU; ++u) oacc[u] = (f32x16)(0.f);
  stamp();

  // GEGLU micro-step m of the tile (hp, gp): step m % 13 of value pair m / 13 (accumulator registers 2 p, 2 p + 1 = two consecutive hidden
  // channels); the last step packs the pair into word p & 3 of the ff2 B fragment p >> 2.  The accumulators live in AGPRs (the kernel
  // needs > 256 registers): their reads are micro-steps of their own -- left to the compiler, all 32 v_accvgpr_read of a tile land in
  // front of the phase's first MFMA.  bh / bg: the ff1 biases of the tile for this lane (register i <-> hidden 4 hh + (i & 3) + 8 (i >> 2)).
  GluChain gc;
  f32x4 bh[4], bg[4];
  const float seven = 7.f;
  auto acc_read = [](float a) { float v; asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(v) : "a"(a)); return v; };
  auto min_abs = [&](float v) { float z; asm("v_min_f32_e64 %0, |%1|, %2" : "=v"(z) : "v"(v), "v"(seven)); return z; };
  // <2 x float> arithmetic: hipcc emits v_pk_fma_f32 / v_pk_add_f32 only where no MFMA is in flight and scalar pairs behind an MFMA
  // (its "unpack packed instructions overlapped by MFMAs" pass) -- rightly: forced to v_pk_* by inline asm, a packed instruction waits
  // for the matrix pipe and the iteration went from 3450 to 4320 ticks.
  auto pk_fma_vs = [](f32x2 a, f32x2 b, f32x2 c) { return __builtin_elementwise_fma(a, b, c); };
  auto pk_nfma = [](f32x2 a, f32x2 b, f32x2 c) { return __builtin_elementwise_fma(-a, b, c); };
  auto pk_add = [](f32x2 a, f32x2 b) { return a + b; };
  auto pk_mul = [](f32x2 a, f32x2 b) { return a * b; };
  const f32x2 g5v = (f32x2)(FF_G5);
  auto glu_ms = [&](auto Mc, const f32x16& hp, const f32x16& gp, u32x4 (&gbx)[2]) {
    constexpr int m = decltype(Mc)::value, p = m / FF_MS, k = m % FF_MS, e0 = 2 * p, e1 = 2 * p + 1;
    if constexpr (k == 0) gc.x = (f32x2){acc_read(gp[e0]), acc_read(gp[e1])};
    if constexpr (k == 1) gc.x = pk_add(gc.x, __builtin_shufflevector(bg[e0 >> 2], bg[e0 >> 2], e0 & 3, e1 & 3));
    if constexpr (k == 2) gc.z = (f32x2){min_abs(gc.x[0]), min_abs(gc.x[1])};
    if constexpr (k == 3) gc.a = pk_fma_vs(g5v, gc.z, (f32x2)(FF_G4));
    if constexpr (k == 4) gc.a = pk_fma_vs(gc.a, gc.z, (f32x2)(FF_G3));
    if constexpr (k == 5) gc.a = pk_fma_vs(gc.a, gc.z, (f32x2)(FF_G2));
    if constexpr (k == 6) gc.a = pk_fma_vs(gc.a, gc.z, (f32x2)(FF_G1));
    if constexpr (k == 7) gc.a = pk_fma_vs(gc.a, gc.z, (f32x2)(FF_G0));
    if constexpr (k == 8) gc.a = (f32x2){__builtin_amdgcn_exp2f(gc.a[0]), __builtin_amdgcn_exp2f(gc.a[1])};     // Phi(-|x|)
    if constexpr (k == 9) gc.x = (f32x2){relu1(gc.x[0]), relu1(gc.x[1])};
    if constexpr (k == 10) gc.a = pk_nfma(gc.z, gc.a, gc.x);        // x Phi(x) = max(x, 0) - |x| Phi(-|x|)   (|x| clamped at 7: Phi(-7) = 1.3e-12)
    if constexpr (k == 11) gc.z = (f32x2){acc_read(hp[e0]), acc_read(hp[e1])};
    if constexpr (k == 12) {
      const f32x2 o = pk_mul(pk_add(gc.z, __builtin_shufflevector(bh[e0 >> 2], bh[e0 >> 2], e0 & 3, e1 & 3)), gc.a);
      gbx[p >> 2][p & 3] = pack_bf16x2(o[0], o[1]);
    }
  };
  auto glu_slot = [&](auto Sc, const f32x16& hp, const f32x16& gp, u32x4 (&gbx)[2]) {      // the micro-steps that ride behind MFMA slot s (0 .. 59)
    constexpr int s = decltype(Sc)::value;
    for_range(FF_IC(ff_ms_first(s)), FF_IC(ff_ms_first(s + 1)), [&](auto mc) { glu_ms(mc, hp, gp, gbx); });
  };
  auto read_bias = [&](int sb) {
    const float* bl = reinterpret_cast<const float*>(smem + FF_LB + sb * 256) + 4 * hh;
#pragma unroll
    for (int g4 = 0; g4 < 4; ++g4) { bh[g4] = *reinterpret_cast<const f32x4*>(bl + 8 * g4); bg[g4] = *reinterpret_cast<const f32x4*>(bl + 32 + 8 * g4); }
  };
  auto as_frag = [](const u32x4& v) { union { u32x4 u; s16x8 s; } cv; cv.u = v; return cv.s; };

  s16x8 fa[2][4], fb[2][4];              // fragment rings of the two phases: [group parity][MFMA of the group]
  auto read_a = [&](int sb, auto Gc) {   // ff1 fragments of MFMA group g (k-steps 2 g, 2 g + 1: h, gate, h, gate) of sub-block sb
    constexpr int g = decltype(Gc)::value;
    const char* s1 = smem + FF_L1 + (sb & 1) * FF_W1 + lane * 16 + g * 4096;
#pragma unroll
    for (int q = 0; q < 4; ++q) fa[g & 1][q] = *reinterpret_cast<const s16x8*>(s1 + q * 1024);
  };
  auto read_b = [&](int sb, auto STc) {  // ff2 fragments of step st: (tile 2 st, k 0) (2 st, k 1) (2 st + 1, k 0) (2 st + 1, k 1)
    constexpr int st = decltype(STc)::value;
    const char* s2 = smem + FF_L2 + (sb & 1) * FF_W2 + lane * 16 + st * 4096;
#pragma unroll
    for (int q = 0; q < 4; ++q) fb[st & 1][q] = *reinterpret_cast<const s16x8*>(s2 + q * 1024);
  };
  // ONE wait per MFMA group: all fragment reads but the N youngest (the next group's, requested just before) have returned.  (The
  // builtin, not inline asm: the compiler's own wait insertion sees it and adds nothing per MFMA.)  gfx9 encoding: lgkmcnt in bits 11:8,
  // vmcnt / expcnt fields left at their maxima.
  auto group_wait = [](auto Nc) { __builtin_amdgcn_s_waitcnt(0xC07F | (decltype(Nc)::value << 8)); };
  // hand-over at the start of a phase's last group: this wave's LDS reads are complete and its share of the next phase's weights has
  // landed (`Younger` DMA pieces may stay in flight), then the workgroup barrier
  auto hand_over = [&](auto Yc) {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    wait_vmcnt<decltype(Yc)::value>();
    __builtin_amdgcn_s_barrier();
  };

  // Phase A.  FF1: ff1 of sub-block sbn into (hn, gn) (the first MFMA of each takes C = 0; the bias is added by the GEGLU), one DMA piece
  // per group (an LDS-DMA piece occupies the CU's one texture-address unit for >= 16 cycles and the four waves issue theirs at the same
  // time: bursts stall the wave): the second half of W1(sbn + 1) on groups 0 .. 4, W2(sbn) on groups 5 .. 9; GLU: the micro-steps of slots 20 .. 59 of the GEGLU of (hp, gp) into gbx; NEXT: the hand-over
  // to phase B of sub-block sbn - 1 in front of the last group (its first fragments are requested there).
  auto phaseA = [&](int sbn, auto FF1c, auto GLUc, auto NEXTc, auto DMA1c, f32x16& hn, f32x16& gn, const f32x16& hp, const f32x16& gp, u32x4 (&gbx)[2]) {
    constexpr bool FF1 = decltype(FF1c)::value, GLU = decltype(GLUc)::value && DBG != 2, NEXT = decltype(NEXTc)::value, DMA1 = decltype(DMA1c)::value;
    for_range(FF_IC(0), FF_IC(FF_KS / 2), [&](auto gcn) {
      constexpr int g = decltype(gcn)::value;
      if constexpr (NEXT && g == FF_KS / 2 - 1) {
        hand_over(FF_IC(FF_P1 + FF_P2 - 1));
        read_b(sbn - 1, FF_IC(0));
      }
      if constexpr (FF1 && g + 1 < FF_KS / 2) read_a(sbn, FF_IC(g + 1));
      if constexpr (FF1) group_wait(FF_IC((g + 1 < FF_KS / 2 ? 4 : 0) + (NEXT && g == FF_KS / 2 - 1 ? 4 : 0)));
      if constexpr (FF1 && DMA1 && g < 5) issue1(sbn + 1, FF_IC(5 + g));      // one DMA piece per group: W1(sbn + 1) second half, then W2(sbn)
      if constexpr (FF1 && g >= 5) issue2(sbn, FF_IC(g - 5));
      for_range(FF_IC(0), FF_IC(4), [&](auto qc) {
        constexpr int q = decltype(qc)::value, ks = 2 * g + (q >> 1);
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (FF1) {
          if (DBG != 2) {
            f32x16& acc = (q & 1) ? gn : hn;
            acc = mma32b(fa[g & 1][q], xf[ks], ks == 0 ? (f32x16)(0.f) : acc);
          }
        }
        if constexpr (GLU) glu_slot(FF_IC(20 + 4 * g + q), hp, gp, gbx);
      });
      __builtin_amdgcn_sched_barrier(0);
    });
  };
  // Phase B.  MM: ff2 of sub-block sb from the B fragments gbc, channel tiles in the order (u, k0) (u+1, k0) (u, k1) (u+1, k1) -- the two
  // k-steps of a tile accumulate into the same registers, back to back they would run at the MFMA's latency, not its issue rate; DMA:
  // the first half of W1(sb + 3), five pieces on steps 0 .. 3; GLU: the micro-steps of slots 0 .. 19 of the GEGLU of (hp, gp) into gbx (its biases
  // are read at the start); NEXT: the hand-over to phase A of sub-block sb + 2 in front of the last step.
  auto phaseB = [&](int sb, auto MMc, auto DMAc, auto GLUc, auto NEXTc, const u32x4 (&gbc)[2], const f32x16& hp, const f32x16& gp, u32x4 (&gbx)[2]) {
    constexpr bool MM = decltype(MMc)::value, DMA = decltype(DMAc)::value, GLU = decltype(GLUc)::value && DBG != 2, NEXT = decltype(NEXTc)::value;
    s16x8 g0, g1;
    if constexpr (MM) { g0 = as_frag(gbc[0]); g1 = as_frag(gbc[1]); }
    if constexpr (GLU) read_bias(sb + 1);
    for_range(FF_IC(0), FF_IC(FF_NU / 2), [&](auto stc) {
      constexpr int st = decltype(stc)::value, u = 2 * st;
      if constexpr (NEXT && st == FF_NU / 2 - 1) {
        hand_over(FF_IC(2 * FF_P2));
        read_a(sb + 2, FF_IC(0));
      }
      if constexpr (MM && st + 1 < FF_NU / 2) read_b(sb, FF_IC(st + 1));
      if constexpr (MM) group_wait(FF_IC((st + 1 < FF_NU / 2 ? 4 : 0) + (NEXT && st == FF_NU / 2 - 1 ? 4 : 0)));
      for_range(FF_IC(0), FF_IC(4), [&](auto qc) {
        constexpr int q = decltype(qc)::value;      // MFMA q of the step: tile u + (q & 1), k-step q >> 1  (fragment 2 (q & 1) + (q >> 1))
        if constexpr (DMA && st < 4 && (q == 0 || (st == 0 && q == 2))) issue1(sb + 3, FF_IC(st == 0 ? q / 2 : st + 1));     // W1(sb + 3) first half
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (MM) {
          if (DBG != 2) oacc[u + (q & 1)] = mma32b(fb[st & 1][2 * (q & 1) + (q >> 1)], (q >> 1) ? g1 : g0, oacc[u + (q & 1)]);
        }
        if constexpr (GLU) glu_slot(FF_IC(4 * st + q), hp, gp, gbx);
      });
      __builtin_amdgcn_sched_barrier(0);
    });
  };

  f32x16 hA, gA, hB, gB;
  u32x4 gbA[2] = {}, gbB[2] = {};
  // iteration -1: ff1(0) alone (W2(0) goes out with it), then the first part of GEGLU(0) alone with W1(2) going out
  wait_vmcnt<FF_P1>();                       // W1(0) has landed; W1(1) in flight
  __builtin_amdgcn_s_barrier();
  read_a(0, FF_IC(0));
  phaseA(0, T, F, F, F, hA, gA, hA, gA, gbA);
  __builtin_amdgcn_s_barrier();              // every wave is through ff1(0): its slot takes W1(2)
  phaseB(-1, F, T, T, T, gbA, hA, gA, gbA);  // ... and in front of its last step: W1(1) has landed (younger: W2(0), half of W1(2)), first fragments of ff1(1)
  // iteration j < nsb - 1:   A: ff1(j+1) || GEGLU(j) 2nd part || W1(j+2) 2nd half, W2(j+1) out; hand-over in front of the last group: W2(j)
  //                             landed (younger: both halves of W1(j+2), 4 pieces of W2(j+1))
  //                          B: ff2(j) || GEGLU(j+1) 1st part || W1(j+3) 1st half out; hand-over: W1(j+2) landed (younger: W2(j+1), half W1(j+3))
  // last iteration:          GEGLU 2nd part, drain, ff2.
  auto iteration = [&](int j, f32x16& hp, f32x16& gp, f32x16& hn, f32x16& gn, u32x4 (&gbp)[2], u32x4 (&gbn)[2]) {
    if (j < 6) stamp();
    phaseA(j + 1, T, T, T, T, hn, gn, hp, gp, gbp);
    if (j < 6) stamp();
    phaseB(j, T, T, T, T, gbp, hn, gn, gbn);
  };
  // proj_out weights: tile pair p (tiles 2 p, 2 p + 1, 40 KiB) -> half p & 1 of the ff1 ring; this wave's 10 contiguous pieces
  const __amdgpu_buffer_rsrc_t rpo = dma_rsrc(PO ? wpo : wimg);
  auto issue_po = [&](int pr, auto Ic) {
    piece(Ic, (DBG != 1 && pr < FF_NU / 2) ? lane16 : DMA_POISON, pr * FF_W1 + wid * (FF_P1 * 1024), smem + FF_L1 + (pr & 1) * FF_W1 + wid * (FF_P1 * 1024), rpo);
  };
  auto last_iteration = [&](int j, f32x16& hp, f32x16& gp, u32x4 (&gbp)[2]) {
    hand_over(FF_IC(0));
    if constexpr (PO) {     // the ff1 ring is idle from here on (every wave is through ff1 of the last sub-block): proj_out tile pairs 0 and 1
      for_range(FF_IC(0), FF_IC(FF_P1), [&](auto i) { issue_po(0, i); });
      for_range(FF_IC(0), FF_IC(FF_P1), [&](auto i) { issue_po(1, i); });
    }
    read_b(j, FF_IC(0));
    phaseA(j + 1, F, T, F, F, hp, gp, hp, gp, gbp);
    phaseB(j, T, F, F, F, gbp, hp, gp, gbp);
  };
  int j = 0;
  for (; j + 2 < nsb; j += 2) {
    iteration(j, hA, gA, hB, gB, gbA, gbB);
    iteration(j + 1, hB, gB, hA, gA, gbB, gbA);
  }
  if (nsb - j == 2) {
    iteration(j, hA, gA, hB, gB, gbA, gbB);
    last_iteration(j + 1, hB, gB, gbB);
  } else {
    last_iteration(j, hA, gA, gbA);
  }
  stamp();
  if constexpr (PO) {
    // ---- hidden = FeedForward + b2 + residual in the ACCUMULATOR layout, packed to bf16: the B fragments of proj_out.  A residual
    // vector holds channels c .. c + 7 (lane hh = 0) / c + 8 .. c + 15 (hh = 1) of c = 32 u + 16 (k / 2); the registers 4 k .. 4 k + 3 and
    // 4 k + 4 .. 4 k + 7 of the tile want c + 4 hh + (0..3) and c + 8 + 4 hh + (0..3): v_permlane32_swap of the vector's first half
    // with its second half (two packed dwords each) is exactly that exchange (the inverse of the store epilogue's).
    s16x8 hf[2 * FF_NU];
    {
      const bf16_t* rr = res + rowc * ldr + 8 * hh;
      u32x4 rv[2 * FF_NU];
#pragma unroll
      for (int i = 0; i < 2 * FF_NU; ++i) rv[i] = *reinterpret_cast<const u32x4*>(rr + 16 * i);
      const float* lb2a = reinterpret_cast<const float*>(smem + FF_LG) + 2 * FFC + 4 * hh;
#pragma unroll
      for (int u = 0; u < FF_NU; ++u) {
        float hv[16];
#pragma unroll
        for (int k = 0; k < 4; k += 2) {
          const u32x4 r4 = rv[2 * u + k / 2];
          const auto s0 = __builtin_amdgcn_permlane32_swap(r4[0], r4[2], false, false);
          const auto s1 = __builtin_amdgcn_permlane32_swap(r4[1], r4[3], false, false);
          const unsigned lo[2] = {s0[0], s1[0]}, hi[2] = {s0[1], s1[1]};      // packed residual of registers 4 k .. 4 k + 3 / 4 k + 4 .. 4 k + 7
          const f32x4 b0 = *reinterpret_cast<const f32x4*>(lb2a + 32 * u + 8 * k), b1 = *reinterpret_cast<const f32x4*>(lb2a + 32 * u + 8 * k + 8);
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const unsigned wl = lo[e >> 1], wh = hi[e >> 1];
            const float rl = __uint_as_float((e & 1) ? (wl & 0xffff0000u) : (wl << 16)), rh = __uint_as_float((e & 1) ? (wh & 0xffff0000u) : (wh << 16));
            hv[4 * k + e] = oacc[u][4 * k + e] + b0[e] + rl;
            hv[4 * k + 4 + e] = oacc[u][4 * k + 4 + e] + b1[e] + rh;
          }
        }
#pragma unroll
        for (int sx = 0; sx < 2; ++sx) {
          float v8[8];
#pragma unroll
          for (int jx = 0; jx < 8; ++jx) v8[jx] = hv[8 * sx + jx];
          hf[2 * u + sx] = pack8(v8);
        }
      }
    }
    // ---- proj_out: tile pair p = output channels 64 p .. 64 p + 63, 2 x 20 MFMAs into oacc[2 p], oacc[2 p + 1]
    s16x8 fp[2][4];                          // fragment ring: [group parity][tile of the pair x k-step of the group]
    for_range(FF_IC(0), FF_IC(FF_NU / 2), [&](auto pc) {
      constexpr int pr = decltype(pc)::value;
      const char* sp = smem + FF_L1 + (pr & 1) * FF_W1 + lane * 16;
      if constexpr (pr + 1 < FF_NU / 2) wait_vmcnt<FF_P1>(); else wait_vmcnt<0>();     // pair pr has landed (pair pr + 1 may be in flight)
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      auto read_p = [&](auto Gc) {           // group g: k-steps 2 g, 2 g + 1 of both tiles
        constexpr int g = decltype(Gc)::value;
#pragma unroll
        for (int q = 0; q < 4; ++q) fp[g & 1][q] = *reinterpret_cast<const s16x8*>(sp + (q >> 1) * (FF_KS * 1024) + (2 * g + (q & 1)) * 1024);
      };
      read_p(FF_IC(0));
      for_range(FF_IC(0), FF_IC(FF_KS / 2), [&](auto gcx) {
        constexpr int g = decltype(gcx)::value;
        if constexpr (g + 1 < FF_KS / 2) { read_p(FF_IC(g + 1)); __builtin_amdgcn_s_waitcnt(0xC07F | (4 << 8)); }
        else __builtin_amdgcn_s_waitcnt(0xC07F);
        __builtin_amdgcn_sched_barrier(0);
        if (DBG != 2) {
#pragma unroll
          for (int q = 0; q < 4; ++q) {      // order (tile 0, k0) (tile 1, k0) (tile 0, k1) (tile 1, k1)
            const int t = q & 1, kk = q >> 1;
            const f32x16 cin = (g == 0 && kk == 0) ? (f32x16)(0.f) : oacc[2 * pr + t];
            oacc[2 * pr + t] = mma32b(fp[g & 1][2 * t + kk], hf[2 * g + kk], cin);
          }
        }
        __builtin_amdgcn_sched_barrier(0);
      });
      if constexpr (pr + 2 < FF_NU / 2) {    // every wave is through pair pr: its half of the ring takes pair pr + 2
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        for_range(FF_IC(0), FF_IC(FF_P1), [&](auto i) { issue_po(pr + 2, i); });
      }
    });
  }
  // ---- epilogue (as the two-role kernel's): + b2 + residual, bf16, 16-byte stores through a buffer resource sized to the M valid rows
  {
    const bf16_t* rr = PO ? res2 + rowc * ldr2 + 8 * hh : res + rowc * ldr + 8 * hh;
    u32x4 rv[2 * FF_NU];
#pragma unroll
    for (int i = 0; i < 2 * FF_NU; ++i) rv[i] = *reinterpret_cast<const u32x4*>(rr + 16 * i);   // channels 16 i + 8 hh .. + 7
    const __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc(out, 0, (int)((long)M * ldo * 2), 0x00020000);
    const unsigned obase = (unsigned)(row * ldo + 8 * hh) * 2u;          // (rows >= M: beyond num_records -> dropped)
    const float* lb2 = (PO ? reinterpret_cast<const float*>(smem + FF_LGT) : reinterpret_cast<const float*>(smem + FF_LG) + 2 * FFC) + 8 * hh;
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
        __builtin_amdgcn_raw_buffer_store_b128(pk, ro, (int)(obase + 2u * c), 0, MMGT_ST_AUX);
      }
  }
  stamp();
#undef FF_IC
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

namespace {
// shared launcher: wpo == NULL: the FeedForward alone; else + proj_out (single-role kernel only)
int ff_fused_launch(const void* x, long ldx, const float* ln_gamma, const float* ln_beta, float eps, const void* wimg, const float* bias2,
                    const void* residual, long ldr, void* out, long ldo, int M, int C, int inner, int dtype, void* stream,
                    const void* wpo, const float* bias_po, const void* res2, long ldr2) {
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
  MMGT_CHECK(!wpo || (bias_po && res2 && ldr2 >= C && ldr2 % 8 == 0 && (((uintptr_t)wpo | (uintptr_t)bias_po | (uintptr_t)res2) & 15) == 0),
             "ff_fused_po: proj_out needs its bias, its residual (row stride >= %d, %% 8 == 0) and 16-byte aligned pointers", C);
  const size_t lds = FF_LB + (size_t)(inner / 32) * 256;
  const unsigned grid = (unsigned)((M + 127) / 128);
  auto reserve = [](const void* k, bool& done) {
    if (done) return true;
    if (hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, FF_LB + FF_MAXSB * 256) != hipSuccess) {
      mmgt_set_error("ff_fused: cannot reserve %d bytes of LDS", FF_LB + FF_MAXSB * 256);
      return false;
    }
    return done = true;
  };
  {
#ifdef MMGT_ABLATE   // timing ablations (results are garbage): only in libmmgt_hip_abl.so (`make abl`), never in the product library
    auto kern = wpo ? (g_ffn_dbg == 1 ? ff_fused1_kernel<1, false, true> : g_ffn_dbg == 2 ? ff_fused1_kernel<2, false, true>
                       : g_ffn_trace ? ff_fused1_kernel<0, true, true> : ff_fused1_kernel<0, false, true>)
                    : (g_ffn_dbg == 1 ? ff_fused1_kernel<1, false, false> : g_ffn_dbg == 2 ? ff_fused1_kernel<2, false, false>
                       : g_ffn_trace ? ff_fused1_kernel<0, true, false> : ff_fused1_kernel<0, false, false>);
    const int ai = g_ffn_dbg == 1 ? 1 : g_ffn_dbg == 2 ? 2 : g_ffn_trace ? 3 : 0;
#else
    auto kern = wpo ? (g_ffn_trace ? ff_fused1_kernel<0, true, true> : ff_fused1_kernel<0, false, true>)
                    : (g_ffn_trace ? ff_fused1_kernel<0, true, false> : ff_fused1_kernel<0, false, false>);
    const int ai = g_ffn_trace ? 3 : 0;
#endif
    static bool attr[2][4] = {};
    if (!reserve(reinterpret_cast<const void*>(kern), attr[wpo != nullptr][ai])) return 2;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, (hipStream_t)stream, (const bf16_t*)x, ldx, ln_gamma, ln_beta, eps,
                       (const char*)wimg, inner / 32, bias2, (const bf16_t*)residual, ldr, (bf16_t*)out, ldo, M, g_ffn_trace,
                       (const char*)wpo, bias_po, (const bf16_t*)res2, ldr2);
  }
  MMGT_LAUNCH_CHECK();
  return 0;
}
}  // namespace

extern "C" int mmgt_ff_fused(const void* x, long ldx, const float* ln_gamma, const float* ln_beta, float eps, const void* wimg,
                             const float* bias2, const void* residual, long ldr, void* out, long ldo, int M, int C, int inner,
                             int dtype, void* stream) {
  return ff_fused_launch(x, ldx, ln_gamma, ln_beta, eps, wimg, bias2, residual, ldr, out, ldo, M, C, inner, dtype, stream, nullptr, nullptr,
                         nullptr, 0);
}

extern "C" int mmgt_ff_fused_po(const void* x, long ldx, const float* ln_gamma, const float* ln_beta, float eps, const void* wimg,
                                const float* bias2, const void* residual, long ldr, const void* wpo, const float* bias_po,
                                const void* residual2, long ldr2, void* out, long ldo, int M, int C, int inner, int dtype, void* stream) {
  MMGT_CHECK(wpo, "ff_fused_po: null proj_out image");
  return ff_fused_launch(x, ldx, ln_gamma, ln_beta, eps, wimg, bias2, residual, ldr, out, ldo, M, C, inner, dtype, stream, wpo, bias_po,
                         residual2, ldr2);
}
