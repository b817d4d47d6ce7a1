// One launch per temporal-attention leg of a level-0 motion module (bf16, C = 320, 8 heads of 40, gfx950):
//
//   out = hid + to_out( softmax_f( q k^T / sqrt(40) ) v ),   q | k | v = (LayerNorm(hid) + pe[frame]) . Wqkv^T        per pixel, over the F frames
//
// Replaces the three launches of mmgt_amd/unet3d.py::_motion_module per attention block (reference: src/models/motion_module.py:236-259
// `TemporalTransformerBlock.forward`, :351-388 `VersatileAttention.forward`, :262-273 `PositionalEncoding`): rowgemm320 (LayerNorm + pe -> q|k|v,
// 126 MB in, 377 MB out), tattn_kernel (377 MB in, 126 MB out) and the out-projection GEMM + residual (252 MB in, 126 MB out) -- 1.38 GB of
// HBM traffic for 0.16 TFLOP at level 0 (389 us, 10 legs per step).  Here the residual stream is read twice and written once (378 MB) and
// q, k, v, the probabilities and the attention output never leave the registers.
//
// Work split.  The F frames of one pixel are F rows of the token matrix, n = h w rows apart.  A wave owns P = 48 / F pixels = 48 rows =
// three 16-row tiles of v_mfma_f32_16x16x32_bf16 (no padding rows for any F that divides 48; instantiated for 24 and 12), ordered pixel-major (rho = px F + f), so a 16 x 16
// score tile is either entirely inside the wave's pixels' diagonal band or skipped.  One wave per SIMD (the rows, the attention output of all
// heads and a head's q | k | v live in ~430 registers), four waves = 4 P adjacent pixels per workgroup, persistent workgroups (one per CU).
//
// Per task (192 rows):
//   1. 48 rows -> registers as MFMA fragments (lane (lm, lq) = row 16 rt + lm, channels 32 ks + 8 lq .. + 7), LayerNorm on the fragments
//      (one-pass statistics with v_dot2c_f32_bf16, ln_frag.h's bound), gamma and the beta + pe[f] table from LDS -> xn (120 registers), which
//      serves as the A and as the B operand (the two fragment layouts coincide).
//   2. per head: K^T = Wk xn^T and Q^T (D[ch][row]: the lane holds 4 channels of a row = half a fragment of the score product over channels),
//      V = xn Wv^T (D[row][ch]: the lane holds 4 frames of a channel = half a fragment of P.V over the keys).  Heads are worked in pairs whose 80
//      channels are five 16-channel tiles (the channels 32 .. 39 of both heads share one), so nothing is padded.  S^T = K Q^T (two MFMAs per
//      tile pair), softmax over the <= F valid keys (cross-lane part: two lane-group exchanges), O^T = V^T P^T, normalised, packed to bf16: the
//      B operand of the out-projection, whose reduction is the 320 channels in tile order.
//   3. out^T = Wo O^T + bias, + residual (re-read), 16-byte stores after v_permlane16_swap (gemm16.hip's epilogue idiom).
// Weights: a fragment-major image (mmgt_amd/packing.py::pack_tleg) of 12 chunks of 30 KiB and 12 of 20 KiB (head pair x {k, v, q} x {wide,
// narrow}) and 10 chunks of 20 KiB (32 output columns each) streams through a 3-slot LDS ring by LDS-DMA, two chunks ahead, one barrier per
// chunk; the stream runs across task boundaries.  Rounding points are those of the three launches it replaces (xn, q | k | v, p, the attention output and the result are bf16;
// every sum is fp32), so the two paths agree to the last bit or one bf16 ulp.
#include <type_traits>

#include "common.h"
#include "gemm_common.h"
#include "mmgt_hip.h"

namespace {

typedef __attribute__((ext_vector_type(4))) float acc4;

constexpr int TL_C = 320, TL_HEADS = 8, TL_HD = 40, TL_ROWS = 48, TL_NRT = 3, TL_NKS = 10, TL_NCT = 3;
// Heads come in pairs (A, B) whose 80 channels are laid out as five 16-channel tiles: A0 A1 | AB2 | B0 B1 with AB2 = [A's channels 32 .. 39 |
// B's channels 32 .. 39] -- no padding rows (three tiles per head would pad 40 to 48: +20 % of the q | k | v MFMAs and weight bytes).  A projection
// of a pair runs as two chunks: "wide" = A0 A1 AB2 (30 fragments, padded to 32 KiB in the image: 8 whole DMA pieces per wave), "narrow" = B0 B1
// (20 fragments, 5 pieces per wave); the out-projection's chunks (32 output columns) are narrow too.  Chunk order of a task: per pair
// k, v, q wide | k, v, q narrow, then the ten out-projection chunks.
constexpr int TL_WIDE = 32 * 1024, TL_NARROW = TL_NKS * 2 * 1024, TL_NPAIR = TL_HEADS / 2, TL_PAIR_BYTES = 3 * TL_WIDE + 3 * TL_NARROW;
constexpr int TL_NQKV = 6 * TL_NPAIR, TL_NOUT = TL_C / 32, TL_NCH = TL_NQKV + TL_NOUT, TL_IMG = TL_NPAIR * TL_PAIR_BYTES + TL_NOUT * TL_NARROW;
static_assert(TL_NKS * TL_NCT * 1024 <= TL_WIDE && TL_NARROW % (4 * 1024) == 0 && TL_WIDE % (4 * 1024) == 0, "chunk geometry");
constexpr int TL_NSLOT = 3, TL_SLOT = TL_WIDE;
constexpr int TL_BPE_STRIDE = TL_C * 4 + 16;                                        // beta + pe rows in LDS: 16 rows on 16 different bank groups
constexpr int TL_L_RING = 0, TL_L_GAMMA = TL_NSLOT * TL_SLOT, TL_L_BIAS = TL_L_GAMMA + TL_C * 4, TL_L_BPE = TL_L_BIAS + TL_C * 4;
constexpr int TL_MAXF = 24;
constexpr int TL_LDS = TL_L_BPE + TL_MAXF * TL_BPE_STRIDE;
static_assert(TL_LDS <= 160 * 1024, "LDS");

struct TlegArgs {
  const bf16_t* x;          // (B F n, 320) rows (b, f, pixel)
  bf16_t* out;              // same shape (may alias x: a wave reads its rows before it writes them)
  const float* gamma;       // LayerNorm weight (320)
  const float* bpe;         // (>= F, 320): LayerNorm bias + positional encoding of frame f
  const char* wimg;         // pack_tleg image
  const float* bias_o;      // out-projection bias (320)
  int n, ntasks, tasks_per_batch;
  float scale_log2e, eps;
};

__device__ __forceinline__ acc4 mma16(s16x8 a, s16x8 b, acc4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }
__device__ __forceinline__ s16x8 frag2(u32x2 lo, u32x2 hi) {
  union { u32x4 u; s16x8 s; } cv;
  cv.u = (u32x4){lo[0], lo[1], hi[0], hi[1]};
  return cv.s;
}
__device__ __forceinline__ u32x2 pack4(acc4 v) { return (u32x2){pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])}; }
__device__ __forceinline__ float bf_lo(unsigned u) { return __uint_as_float(u << 16); }
__device__ __forceinline__ float bf_hi(unsigned u) { return __uint_as_float(u & 0xffff0000u); }
// all-reduce over the four lanes (lm, lq = 0 .. 3) that share a row
__device__ __forceinline__ float quad_sum(float v) {
  const auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  v = __uint_as_float(a[0]) + __uint_as_float(a[1]);
  const auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return __uint_as_float(b[0]) + __uint_as_float(b[1]);
}
__device__ __forceinline__ float quad_max(float v) {
  const auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  v = fmaxf(__uint_as_float(a[0]), __uint_as_float(a[1]));
  const auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return fmaxf(__uint_as_float(b[0]), __uint_as_float(b[1]));
}

// do score tiles (key tile kt, query tile qt) hold any (key, query) pair of the same pixel?  rows 16 t .. 16 t + 15, pixel = row / F
template <int F> constexpr bool tl_pair(int kt, int qt) { return (16 * kt) / F <= (16 * qt + 15) / F && (16 * qt) / F <= (16 * kt + 15) / F; }
// ... and is every pair of them in the same pixel (no mask needed)?
template <int F> constexpr bool tl_all(int kt, int qt) { return (16 * kt) / F == (16 * kt + 15) / F && (16 * qt) / F == (16 * qt + 15) / F && (16 * kt) / F == (16 * qt) / F; }

// ABL (mmgt_tune("tleg_abl", bits) of the -DMMGT_ABLATE build, 24-frame kernel, timing only -- results are garbage): 1 no LayerNorm arithmetic, 2 no attention arithmetic,
// 4 no weight DMA after the first chunk, 8 no epilogue (residual loads, stores), 16 no projection MFMAs, 32 no hand-over wait / barrier, 64 no row loads
// after the first task, 128 the counted waits of the hand-over but no barrier
template <int F, int ABL>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1)))
void tleg320_kernel(const TlegArgs a) {
  static_assert(TL_ROWS % F == 0 && F <= TL_MAXF, "frames per pixel must divide 48");
  constexpr int P = TL_ROWS / F;                         // pixels per wave
  extern __shared__ __attribute__((aligned(1024))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lm = lane & 15, lq = lane >> 4;
  using std::integral_constant;
#define TL_IC(v) integral_constant<int, (v)>{}
  auto for_range = [](auto LOc, auto HIc, auto&& fn) {
    constexpr int lo = decltype(LOc)::value, hi = decltype(HIc)::value;
    [&]<int... I>(std::integer_sequence<int, I...>) { (fn(integral_constant<int, lo + I>{}), ...); }(std::make_integer_sequence<int, (hi > lo ? hi - lo : 0)>{});
  };

  const int G = gridDim.x;
  const int my_tasks = (a.ntasks - (int)blockIdx.x + G - 1) / G;
  const int total = my_tasks * TL_NCH;                   // chunks this workgroup consumes

  // ---- weight stream.  Chunk g (counted over the workgroup's tasks) is weight chunk g % 34 and lives in ring slot g % 3.  Behind the first
  // k-step of chunk g every wave issues its quarter of chunk g + 2 in one go (8 or 5 contiguous 1-KiB pieces, four to one M0 / scalar-offset
  // setting: ~14 instructions -- one piece per k-step with its own address arithmetic measured 59 us of a 261-us leg) into the slot chunk
  // g - 1 left; in front of the chunk's LAST k-step it waits for its pieces of chunk g + 1 (counted: the pieces of chunk g + 2 stay in
  // flight -- the stream moves ~1 GB per leg from L2 to LDS, a chunk needs about as long to land as one takes to multiply) and for its own
  // fragment reads of chunk g (all ten k-steps are in registers by then) and meets the others at ONE barrier: behind it chunk g + 1 has landed
  // for everybody and slot g % 3 is free for chunk g + 3, and the first fragments of chunk g + 1 are requested there, under the last k-step's
  // MFMAs.
  const __amdgpu_buffer_rsrc_t rw = dma_rsrc(a.wimg);
  const unsigned lane16 = (unsigned)lane * 16u;
  int cur_slot = 0;                                      // ring slot of the chunk being consumed
  int dma_c = 2, dma_g = 2, dma_slot = 2;                // the chunk whose pieces go out next: weight chunk index, global count, ring slot
  auto issue_pieces = [&](auto Nc, int src, char* dst) {  // N contiguous pieces from image offset src to dst
    for_range(TL_IC(0), Nc, [&](auto ic) {
      constexpr int i = decltype(ic)::value;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (__attribute__((address_space(3))) void*)(dst + (i / 4) * 4096), 16, (int)lane16,
                                               src + (i / 4) * 4096, (i % 4) * 1024, 0);
    });
  };
  auto issue_chunk = [&](int c, int slot) {               // this wave's quarter of weight chunk c into ring slot `slot`
    char* ring = smem + TL_L_RING + slot * TL_SLOT;
    const int pr = c / 6, r = c - pr * 6;                 // (c < TL_NQKV) pair, chunk of the pair: 0 .. 2 wide, 3 .. 5 narrow
    if (c < TL_NQKV && r < 3) {
      issue_pieces(TL_IC(TL_WIDE / 4096), pr * TL_PAIR_BYTES + r * TL_WIDE + wid * (TL_WIDE / 4), ring + wid * (TL_WIDE / 4));
    } else {
      const int src = c < TL_NQKV ? pr * TL_PAIR_BYTES + 3 * TL_WIDE + (r - 3) * TL_NARROW : TL_NPAIR * TL_PAIR_BYTES + (c - TL_NQKV) * TL_NARROW;
      issue_pieces(TL_IC(TL_NARROW / 4096), src + wid * (TL_NARROW / 4), ring + wid * (TL_NARROW / 4));
    }
  };
  auto chunk_next = [&]() {                              // the consumer moves on to the next chunk
    cur_slot = cur_slot == TL_NSLOT - 1 ? 0 : cur_slot + 1;
    dma_slot = dma_slot == TL_NSLOT - 1 ? 0 : dma_slot + 1;
    ++dma_g;
    dma_c = dma_c == TL_NCH - 1 ? 0 : dma_c + 1;
  };
  // ---- lane constant of the score mask: bit (3 qt + kt) <=> the lane's keys of tile kt (rows 16 kt + 4 lq .. + 3) and its query of tile qt
  // (row 16 qt + lm) belong to the same pixel.  One bit per tile pair: F is a multiple of 4, so a lane's four keys never straddle a pixel.
  static_assert(F % 4 == 0, "a lane's four key rows must belong to one pixel");
  unsigned tmask = 0;
#pragma unroll
  for (int qt = 0; qt < TL_NRT; ++qt)
#pragma unroll
    for (int kt = 0; kt < TL_NRT; ++kt)
      if ((16 * kt + 4 * lq) / F == (16 * qt + lm) / F) tmask |= 1u << (3 * qt + kt);
  // ---- prologue: the first weight chunk on its way, the tables into LDS
  issue_chunk(0, 0);
  if (total > 1) issue_chunk(1, 1);
  {
    float* lg = reinterpret_cast<float*>(smem + TL_L_GAMMA);
    float* lb = reinterpret_cast<float*>(smem + TL_L_BIAS);
    if (tid < TL_C / 4) *reinterpret_cast<f32x4*>(lg + 4 * tid) = *reinterpret_cast<const f32x4*>(a.gamma + 4 * tid);
    else if (tid < 2 * TL_C / 4) *reinterpret_cast<f32x4*>(lb + 4 * (tid - TL_C / 4)) = *reinterpret_cast<const f32x4*>(a.bias_o + 4 * (tid - TL_C / 4));
    for (int i = tid; i < F * (TL_C / 4); i += 256) {
      const int f = i / (TL_C / 4), c4 = i - f * (TL_C / 4);
      *reinterpret_cast<f32x4*>(smem + TL_L_BPE + f * TL_BPE_STRIDE + c4 * 16) = *reinterpret_cast<const f32x4*>(a.bpe + (long)f * TL_C + 4 * c4);
    }
  }
  wait_vmcnt<0>();
  __syncthreads();                                       // tables and chunks 0, 1 are in LDS
  s16x8 wf[2][TL_NCT];                                   // weight fragments: [k-step parity][tile]; a chunk's first ones are read under its predecessor's last k-step
#pragma unroll
  for (int f = 0; f < TL_NCT; ++f) wf[0][f] = *reinterpret_cast<const s16x8*>(smem + TL_L_RING + lane * 16 + f * 1024);

  const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(a.x), 0, (int)DMA_RANGE, 0x00020000);
  const __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc(a.out, 0, (int)DMA_RANGE, 0x00020000);
  const int wbase = lane * 16;                           // this lane's 16 bytes of a 1-KiB fragment

  // per-row constants of a task, recomputed where they are needed (not kept in registers across the heads): frame and byte offset of row 16 rt + lm
  auto row_consts = [&](int task, int (&rfr)[TL_NRT], unsigned (&rowoff)[TL_NRT]) {
    const int b = task / a.tasks_per_batch, pblk = task - b * a.tasks_per_batch;
    const int p0 = pblk * (4 * P) + wid * P;
    int lme = lm;
    asm volatile("" : "+v"(lme));
#pragma unroll
    for (int rt = 0; rt < TL_NRT; ++rt) {
      const int px = (16 * rt + lme) / F;                // pixel inside the wave
      rfr[rt] = (16 * rt + lme) - px * F;
      rowoff[rt] = (unsigned)(((b * F + rfr[rt]) * a.n + p0 + px) * (TL_C * 2));
    }
  };
  // The 48 rows of a task as MFMA fragments (lane (lm, lq): row 16 rt + lm, channels 32 ks + 8 lq .. + 7).  The rows of task t + 1 are
  // requested when task t's last projection has consumed xn -- in front of its last attention and its out-projection, ~20 000 cycles
  // before they are normalised -- so only the first task of a workgroup waits for HBM (all workgroups loading at once: ~11 B / cycle / CU).
  s16x8 xn[TL_NRT][TL_NKS];
  auto load_rows = [&](int task) {
    int rfr[TL_NRT];
    unsigned rowoff[TL_NRT];
    row_consts(task, rfr, rowoff);
#pragma unroll
    for (int rt = 0; rt < TL_NRT; ++rt) {
      for_range(TL_IC(0), TL_IC(TL_NKS), [&](auto kc) {
        constexpr int ks = decltype(kc)::value;
        union { u32x4 u; s16x8 s; } cv;
        cv.u = __builtin_amdgcn_raw_buffer_load_b128(rx, (int)(rowoff[rt] + 16 * lq + 64 * ks), 0, 0);
        xn[rt][ks] = cv.s;
      });
    }
  };
  if ((int)blockIdx.x < a.ntasks) load_rows(blockIdx.x);

  for (int task = blockIdx.x; task < a.ntasks; task += G) {
    int rfr[TL_NRT];
    unsigned rowoff[TL_NRT];
    row_consts(task, rfr, rowoff);

    // ---- 1. LayerNorm (+ pe) of the fragments
    if constexpr (!(ABL & 1)) {
      typedef __attribute__((ext_vector_type(2))) __bf16 bf2;
      const bf2 ones = {(__bf16)1.0f, (__bf16)1.0f};
      const float* lg = reinterpret_cast<const float*>(smem + TL_L_GAMMA);
      f32x2 r2[TL_NRT], nm2[TL_NRT];
#pragma unroll
      for (int rt = 0; rt < TL_NRT; ++rt) {
        float s[4] = {0.f, 0.f, 0.f, 0.f}, q[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < TL_NKS; ++ks) {
          union { s16x8 f; unsigned u[4]; } v;
          v.f = xn[rt][ks];
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const bf2 pr = __builtin_bit_cast(bf2, v.u[j]);
            s[j] = __builtin_amdgcn_fdot2_f32_bf16(pr, ones, s[j], false);
            q[j] = __builtin_amdgcn_fdot2_f32_bf16(pr, pr, q[j], false);
          }
        }
        const float sum = quad_sum((s[0] + s[1]) + (s[2] + s[3])), sq = quad_sum((q[0] + q[1]) + (q[2] + q[3]));
        const float mean = sum * (1.f / (float)TL_C);
        const float var = fmaxf(fmaf(-mean, mean, sq * (1.f / (float)TL_C)), 0.f);
        const float rstd = rsqrtf(var + a.eps);
        r2[rt] = (f32x2){rstd, rstd};
        nm2[rt] = (f32x2){-mean * rstd, -mean * rstd};
      }
#pragma unroll
      for (int ks = 0; ks < TL_NKS; ++ks) {
        const int c = 32 * ks + 8 * lq;
        const f32x4 g0 = *reinterpret_cast<const f32x4*>(lg + c), g1 = *reinterpret_cast<const f32x4*>(lg + c + 4);
#pragma unroll
        for (int rt = 0; rt < TL_NRT; ++rt) {
          const float* lb = reinterpret_cast<const float*>(smem + TL_L_BPE + rfr[rt] * TL_BPE_STRIDE) + c;
          const f32x4 b0 = *reinterpret_cast<const f32x4*>(lb), b1 = *reinterpret_cast<const f32x4*>(lb + 4);
          union { s16x8 f; unsigned u[4]; } v;
          v.f = xn[rt][ks];
          unsigned o[4];
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const f32x2 x2 = {bf_lo(v.u[j]), bf_hi(v.u[j])};
            const f32x2 z = __builtin_elementwise_fma(x2, r2[rt], nm2[rt]);
            const f32x2 g2 = j < 2 ? (f32x2){g0[2 * j], g0[2 * j + 1]} : (f32x2){g1[2 * j - 4], g1[2 * j - 3]};
            const f32x2 b2 = j < 2 ? (f32x2){b0[2 * j], b0[2 * j + 1]} : (f32x2){b1[2 * j - 4], b1[2 * j - 3]};
            const f32x2 y = __builtin_elementwise_fma(z, g2, b2);
            o[j] = pack_bf16x2(y[0], y[1]);
          }
          union { u32x4 u; s16x8 s; } cv;
          cv.u = (u32x4){o[0], o[1], o[2], o[3]};
          xn[rt][ks] = cv.s;
        }
      }
    }

    // ---- 2. heads
    u32x4 opk[TL_NRT][TL_NKS];                           // the attention output as B fragments of the out-projection: [row tile][k-step]

    // One chunk = ten k-steps of NF weight fragments (fragment ks NF + f of the slot).  The last k-step carries the hand-over described at the top.
    auto run_chunk = [&](auto NFc, auto&& mfmas) {
      constexpr int NF = decltype(NFc)::value;
      const char* base = smem + TL_L_RING + cur_slot * TL_SLOT + wbase;
      const char* nbase = smem + TL_L_RING + (cur_slot == TL_NSLOT - 1 ? 0 : cur_slot + 1) * TL_SLOT + wbase;
      const bool live = dma_g < total;
      for_range(TL_IC(0), TL_IC(TL_NKS), [&](auto kc) {
        constexpr int ks = decltype(kc)::value;
        if constexpr (ks == TL_NKS - 1) {
          if constexpr (!(ABL & 32)) {
            // this wave's pieces of the next chunk have landed: everything but the pieces of the chunk after next, the wave's youngest 8 or
            // 5 operations (the stores and residual loads of the out-projection phase are older than those)
            if (!live) wait_vmcnt<0>();
            else if (dma_c < TL_NQKV && (dma_c % 6) < 3) wait_vmcnt<TL_WIDE / 4096>();
            else wait_vmcnt<TL_NARROW / 4096>();
            __builtin_amdgcn_s_waitcnt(0xC07F);                      // ... and its reads of this chunk have returned
            if constexpr (!(ABL & 128)) __builtin_amdgcn_s_barrier();
          }
#pragma unroll
          for (int f = 0; f < TL_NCT; ++f) wf[0][f] = *reinterpret_cast<const s16x8*>(nbase + f * 1024);
        }
        // fragment group f: its MFMAs, then the request for the next k-step's fragment f -- the read goes out under the group's MFMAs instead
        // of in a block in front of the k-step (the matrix pipe drained for ~20 cycles there).  In front of every group NF reads are in
        // flight and the oldest is the one it needs: lgkmcnt(NF - 1)
        for_range(TL_IC(0), NFc, [&](auto fc) {
          constexpr int f = decltype(fc)::value;
          if constexpr (ks + 1 < TL_NKS) __builtin_amdgcn_s_waitcnt(0xC07F | ((NF - 1) << 8));
          __builtin_amdgcn_sched_barrier(0);
          if constexpr (!(ABL & 16)) mfmas(kc, fc);
          __builtin_amdgcn_sched_barrier(0);
          if constexpr (ks + 1 < TL_NKS) wf[(ks + 1) & 1][f] = *reinterpret_cast<const s16x8*>(base + ((ks + 1) * NF + f) * 1024);
        });
        // wave w issues its pieces behind k-step 2 w: issued by all four waves at once the pieces queue in the CU's one address path and the
        // last wave leaves its issue ~1000 cycles late -- the workgroup then waits for it at the hand-over barrier (22 us of a 227-us leg)
        if constexpr (ks % 2 == 0 && ks < 8 && !(ABL & 4)) {
          if (live && wid == ks / 2) issue_chunk(dma_c, dma_slot);
        }
      });
      chunk_next();
    };
    // a projection pass over NT channel tiles: acc[i][j] += W tile x row tile.  SW = false: D[ch][row] (acc[ct][rt]); SW = true: D[row][ch] (acc[rt][ct])
    auto project = [&](auto SWc, auto NTc, acc4 (&acc)[3][3]) {
      constexpr bool SW = decltype(SWc)::value;
      run_chunk(NTc, [&](auto kc, auto fc) {
        constexpr int ks = decltype(kc)::value, ct = decltype(fc)::value;
#pragma unroll
          for (int rt = 0; rt < TL_NRT; ++rt) {
            // (the first k-step takes C = 0 as an inline constant: no accumulator initialisation pass)
            if constexpr (SW) acc[rt][ct] = mma16(xn[rt][ks], wf[ks & 1][ct], ks == 0 ? (acc4)(0.f) : acc[rt][ct]);
            else acc[ct][rt] = mma16(wf[ks & 1][ct], xn[rt][ks], ks == 0 ? (acc4)(0.f) : acc[ct][rt]);
          }
      });
    };
    // k / v / q of a chunk into packed tiles: t[rt][j] = channel tile j of the chunk for row tile rt
    auto project_kvq = [&](auto NTc, u32x2 (&kt_)[TL_NRT][3], u32x2 (&vt_)[TL_NRT][3], u32x2 (&qt_)[TL_NRT][3]) {
      constexpr int NT = decltype(NTc)::value;
      {
        acc4 acc[3][3];
        project(std::false_type{}, NTc, acc);            // K^T: acc[ct][rt]
#pragma unroll
        for (int rt = 0; rt < TL_NRT; ++rt)
#pragma unroll
          for (int ct = 0; ct < NT; ++ct) kt_[rt][ct] = pack4(acc[ct][rt]);
      }
      {
        acc4 acc[3][3];
        project(std::true_type{}, NTc, acc);             // V: acc[rt][ct], lane = channel, registers = 4 key rows
#pragma unroll
        for (int rt = 0; rt < TL_NRT; ++rt)
#pragma unroll
          for (int ct = 0; ct < NT; ++ct) vt_[rt][ct] = pack4(acc[rt][ct]);
      }
      {
        acc4 acc[3][3];
        project(std::false_type{}, NTc, acc);            // Q^T
#pragma unroll
        for (int rt = 0; rt < TL_NRT; ++rt)
#pragma unroll
          for (int ct = 0; ct < NT; ++ct) qt_[rt][ct] = pack4(acc[ct][rt]);
      }
    };
    const u32x2 z2 = (u32x2){0u, 0u};
    // Attention of one head: k / v / q tiles 0, 1 are the head's own channels 0 .. 31, tile 2 is the pair's shared tile AB2, of which the head
    // owns the lanes lq < 2 (HI = false: head A) or lq >= 2 (HI = true: head B) -- q's other half is zeroed for the scores; O^T of tile 2 comes
    // out whole and the caller keeps the head's lanes.  otile[query row tile][tile], normalised and packed.
    auto attention = [&](auto HIc, const u32x2 (&kp)[TL_NRT][3], const u32x2 (&vp)[TL_NRT][3], const u32x2 (&qp)[TL_NRT][3], u32x2 (&otile)[TL_NRT][3]) {
      constexpr bool HI = decltype(HIc)::value;
      if constexpr (ABL & 2) {
#pragma unroll
        for (int rt = 0; rt < TL_NRT; ++rt)
#pragma unroll
          for (int ct = 0; ct < TL_NCT; ++ct) otile[rt][ct] = (u32x2){qp[rt][ct][0] ^ kp[rt][ct][1], vp[rt][ct][0]};
        return;
      }
      const bool mine = HI ? lq >= 2 : lq < 2;
      for_range(TL_IC(0), TL_IC(TL_NRT), [&](auto qc) {
        constexpr int qt = decltype(qc)::value;
        const u32x2 q2 = mine ? qp[qt][2] : z2;
        acc4 s[TL_NRT];
        float mx = -1e30f;
        for_range(TL_IC(0), TL_IC(TL_NRT), [&](auto kc) {
          constexpr int kt = decltype(kc)::value;
          if constexpr (tl_pair<F>(kt, qt)) {
            acc4 t = mma16(frag2(kp[kt][0], kp[kt][1]), frag2(qp[qt][0], qp[qt][1]), (acc4)(0.f));
            t = mma16(frag2(kp[kt][2], z2), frag2(q2, z2), t);
            if constexpr (!tl_all<F>(kt, qt)) {              // one lane condition per tile pair
              const bool ok = (tmask >> (3 * qt + kt)) & 1u;
#pragma unroll
              for (int e = 0; e < 4; ++e) t[e] = ok ? t[e] : -1e30f;
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) mx = fmaxf(mx, t[e]);
            s[kt] = t;
          }
        });
        mx = quad_max(mx);                                   // (of the raw scores: the scale is positive and rides in the exponent's FMA)
        const float nm = -mx * a.scale_log2e;
        float ls = 0.f;
        u32x2 pp[TL_NRT];
        for_range(TL_IC(0), TL_IC(TL_NRT), [&](auto kc) {
          constexpr int kt = decltype(kc)::value;
          if constexpr (tl_pair<F>(kt, qt)) {
            acc4 e4;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              e4[e] = __builtin_amdgcn_exp2f(fmaf(s[kt][e], a.scale_log2e, nm));
              ls += e4[e];
            }
            pp[kt] = pack4(e4);
          } else {
            pp[kt] = z2;
          }
        });
        const float inv = 1.f / quad_sum(ls);
        // O^T[ch][query] = sum over key tiles of V^T P^T, key tiles two at a time (a 32-deep reduction)
        constexpr bool v0 = tl_pair<F>(0, qt), v1 = tl_pair<F>(1, qt), v2 = tl_pair<F>(2, qt);
#pragma unroll
        for (int ct = 0; ct < TL_NCT; ++ct) {
          acc4 o = (acc4)(0.f);
          if constexpr (v0 || v1) o = mma16(frag2(vp[0][ct], vp[1][ct]), frag2(pp[0], pp[1]), o);
          if constexpr (v2) o = mma16(frag2(vp[2][ct], z2), frag2(pp[2], z2), o);
          otile[qt][ct] = pack4(o * inv);
        }
      });
    };

    for (int g = 0; g < TL_NPAIR; ++g) {
      u32x2 kp[TL_NRT][3], vp[TL_NRT][3], qp[TL_NRT][3];     // [row tile][tile]: A0, A1, AB2; later B0, B1 (AB2 stays)
      u32x2 oa[TL_NRT][3], ob[TL_NRT][3];
      project_kvq(TL_IC(3), kp, vp, qp);
      attention(std::false_type{}, kp, vp, qp, oa);          // head A = 2 g
      project_kvq(TL_IC(2), kp, vp, qp);                     // tiles 0, 1 <- B0, B1
      attention(std::true_type{}, kp, vp, qp, ob);           // head B = 2 g + 1
      // into the out-projection's operand: k-step 2 g = [A0 | A1], 2 g + 1 = [B0 | B1]; AB2 (lanes lq < 2: head A's channels 32 .. 39, lq >= 2:
      // head B's) is half of k-step 8 + g / 2
      for_range(TL_IC(0), TL_IC(TL_NPAIR), [&](auto gc) {
        constexpr int gg = decltype(gc)::value;
        if (g == gg) {
#pragma unroll
          for (int rt = 0; rt < TL_NRT; ++rt) {
            opk[rt][2 * gg] = (u32x4){oa[rt][0][0], oa[rt][0][1], oa[rt][1][0], oa[rt][1][1]};
            opk[rt][2 * gg + 1] = (u32x4){ob[rt][0][0], ob[rt][0][1], ob[rt][1][0], ob[rt][1][1]};
            const u32x2 m = lq < 2 ? oa[rt][2] : ob[rt][2];
            constexpr int ks = 8 + gg / 2;
            if constexpr (gg & 1) { opk[rt][ks][2] = m[0]; opk[rt][ks][3] = m[1]; }
            else { opk[rt][ks][0] = m[0]; opk[rt][ks][1] = m[1]; }
          }
        }
      });
    }

    // ---- 3. out-projection, 32 output columns per chunk, + bias + residual
    if (task + G < a.ntasks && !(ABL & 64)) load_rows(task + G);
    row_consts(task, rfr, rowoff);
    for (int oc = 0; oc < TL_NOUT; ++oc) {
      // residual vectors of this chunk's columns (the lane's 8 columns after the swap below)
      const int cofs = 16 * (lq & 1) + 8 * (lq >> 1);
      u32x4 rv[TL_NRT];
#pragma unroll
      for (int rt = 0; rt < TL_NRT; ++rt) rv[rt] = (ABL & 8) ? (u32x4)(0u) : __builtin_amdgcn_raw_buffer_load_b128(rx, (int)(rowoff[rt] + 2 * cofs), 64 * oc, 0);
      acc4 acc[2][TL_NRT];
      {
        const float* lb = reinterpret_cast<const float*>(smem + TL_L_BIAS) + 32 * oc + 4 * lq;
        const acc4 b0 = *reinterpret_cast<const acc4*>(lb), b1 = *reinterpret_cast<const acc4*>(lb + 16);
#pragma unroll
        for (int rt = 0; rt < TL_NRT; ++rt) { acc[0][rt] = b0; acc[1][rt] = b1; }
      }
      run_chunk(TL_IC(2), [&](auto kc, auto fc) {
        constexpr int ks = decltype(kc)::value, nt = decltype(fc)::value;
#pragma unroll
          for (int rt = 0; rt < TL_NRT; ++rt) {
            union { u32x4 u; s16x8 s; } ob;
            ob.u = opk[rt][ks];
            acc[nt][rt] = mma16(wf[ks & 1][nt], ob.s, acc[nt][rt]);
          }
      });
#pragma unroll
      for (int rt = 0; rt < TL_NRT; ++rt) {
        float o8[8];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const auto sw = __builtin_amdgcn_permlane16_swap(__float_as_uint(acc[0][rt][r]), __float_as_uint(acc[1][rt][r]), false, false);
          o8[r] = __uint_as_float(sw[0]);
          o8[4 + r] = __uint_as_float(sw[1]);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          o8[2 * j] += bf_lo(rv[rt][j]);
          o8[2 * j + 1] += bf_hi(rv[rt][j]);
        }
        const u32x4 pk = (u32x4){pack_bf16x2(o8[0], o8[1]), pack_bf16x2(o8[2], o8[3]), pack_bf16x2(o8[4], o8[5]), pack_bf16x2(o8[6], o8[7])};
        // (the chunk's column offset rides in the VECTOR offset: with a register in soffset hipcc does not keep the wait states between a 16-byte
        //  store and the next write of its data registers, and MI355X then stores the overwritten values in lanes 12 .. 15 of every 16-lane
        //  row -- found in csrc/gnconv.hip, round 5; tools/check_mfma_overlap.py scans every object of the library for the pattern)
        if (!(ABL & 8) || pk[0] == 0x12345678u) __builtin_amdgcn_raw_buffer_store_b128(pk, ro, (int)(rowoff[rt] + 2 * cofs + 64 * oc), 0, MMGT_ST_AUX);
      }
    }
  }
#undef TL_IC
}

int g_tleg_abl = 0;

}  // namespace

void mmgt_tleg_set_abl(int v) { g_tleg_abl = v; }

extern "C" long mmgt_temporal_leg320_image_bytes(void) { return TL_IMG; }

extern "C" int mmgt_temporal_leg320(const void* x, void* out, const float* ln_gamma, const float* beta_pe, int pe_rows, const void* wimg,
                                    const float* bias_o, int batch, int frames, int n_pix, float scale, float eps, int dtype, void* stream) {
  MMGT_CHECK(x && out && ln_gamma && beta_pe && wimg && bias_o, "temporal_leg320: null pointer");
  MMGT_CHECK(dtype == MMGT_BF16, "temporal_leg320: bf16 only (the fp32-I/O mode runs LayerNorm / GEMM / attention / GEMM)");
  MMGT_CHECK(frames == 24 || frames == 12, "temporal_leg320: built for windows of 24 (BASELINE config 2) or 12 frames (the reference's shipped context_frames), got %d", frames);
  MMGT_CHECK(pe_rows >= frames, "temporal_leg320: the beta + pe table has %d rows, the window %d frames", pe_rows, frames);
  const int P = TL_ROWS / frames;
  MMGT_CHECK(batch > 0 && n_pix > 0 && n_pix % (4 * P) == 0, "temporal_leg320: pixels per frame (%d) must be a multiple of %d", n_pix, 4 * P);
  MMGT_CHECK((long)batch * frames * n_pix * TL_C * 2 < (1l << 31), "temporal_leg320: tensor beyond the 2 GiB range of a buffer resource (split the batch)");
  MMGT_CHECK((((uintptr_t)x | (uintptr_t)out | (uintptr_t)ln_gamma | (uintptr_t)beta_pe | (uintptr_t)wimg | (uintptr_t)bias_o) & 15) == 0,
             "temporal_leg320: pointers must be 16-byte aligned");
  TlegArgs a;
  a.x = (const bf16_t*)x; a.out = (bf16_t*)out; a.gamma = ln_gamma; a.bpe = beta_pe; a.wimg = (const char*)wimg; a.bias_o = bias_o;
  a.n = n_pix; a.tasks_per_batch = n_pix / (4 * P); a.ntasks = batch * a.tasks_per_batch;
  a.scale_log2e = scale * 1.4426950408889634f; a.eps = eps;

  int dev = 0;
  static int ncu[16] = {};
  MMGT_CHECK(hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev < 16, "temporal_leg320: device query failed");
  if (!ncu[dev]) {
    hipDeviceProp_t prop;
    MMGT_CHECK(hipGetDeviceProperties(&prop, dev) == hipSuccess, "temporal_leg320: device query failed");
    ncu[dev] = prop.multiProcessorCount;
  }
  const unsigned grid = (unsigned)(a.ntasks < ncu[dev] ? a.ntasks : ncu[dev]);
  auto go = [&](auto kern, bool& attr) {
    if (!attr) {
      if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, TL_LDS) != hipSuccess) {
        mmgt_set_error("temporal_leg320: cannot reserve %d bytes of LDS", TL_LDS);
        return 2;
      }
      attr = true;
    }
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), (size_t)TL_LDS, (hipStream_t)stream, a);
    MMGT_LAUNCH_CHECK();
    return 0;
  };
  static bool attr[16][13] = {};
  if (frames == 24) {
#ifdef MMGT_ABLATE   // timing ablations (results are garbage): only in libmmgt_hip_abl.so (`make abl`), never in the product library
    switch (g_tleg_abl) {
      case 1: return go(tleg320_kernel<24, 1>, attr[dev][4]);
      case 2: return go(tleg320_kernel<24, 2>, attr[dev][5]);
      case 4: return go(tleg320_kernel<24, 4>, attr[dev][6]);
      case 8: return go(tleg320_kernel<24, 8>, attr[dev][7]);
      case 16: return go(tleg320_kernel<24, 16>, attr[dev][8]);
      case 32: return go(tleg320_kernel<24, 32>, attr[dev][9]);
      case 64: return go(tleg320_kernel<24, 64>, attr[dev][10]);
      case 36: return go(tleg320_kernel<24, 36>, attr[dev][11]);
      case 128: return go(tleg320_kernel<24, 128>, attr[dev][12]);
      default: break;
    }
#endif
    return go(tleg320_kernel<24, 0>, attr[dev][0]);
  }
  return go(tleg320_kernel<12, 0>, attr[dev][1]);
}
