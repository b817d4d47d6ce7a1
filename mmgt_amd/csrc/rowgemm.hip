// Row-stationary GEMM for the 320-channel level (bf16, gfx950):   out = [LayerNorm | GroupNorm affine](x) . W^T + bias [+ bias2[row group]] [+ residual]
//
// The projections around the level-0 attentions (attention.py:346-360,440-462,700-760: to_q / to_k / to_v / to_out; motion_module.py:
// 294-330) are M = 196 608 rows x K = 320: a 256 x 256 GEMM tile refills its pipeline every five K chunks and the LayerNorm in front
// of them is a pass of its own (profiles/r3/opshapes_r3a.txt: 435 - 740 TFLOP/s, and 19 ln_kernel launches of 55 us per step).  Here a
// wave keeps its 32 rows for the whole launch -- normalised in registers (80 registers, the B operand of every MFMA) -- and walks over
// the N output columns in tiles of 32: 20 MFMAs (32x32x16) per tile whose A operand, a 1-KiB weight fragment, comes from a 3-stage LDS
// ring filled by LDS-DMA from a fragment-major weight image (mmgt_amd/packing.py: pack_rowgemm; L2 resident, <= 600 KB).  x is read
// once, however wide N is; q / k and V^T of a self-attention come out of ONE launch: a tile is either "normal" (D = W . x^T: the lane
// holds a row, 16-byte stores into out[row][32 t ..]) or "transposed" (D = x . W^T, the same two fragments with the MFMA operands swapped:
// the lane holds a column and 8 consecutive rows, 16-byte stores into out_t[batch][column][token] -- the V^T layout of attn64.hip).
//
// Workgroup = 4 waves = 128 rows, <= 256 registers and 68 KB of LDS: TWO workgroups per CU, i.e. two independent instruction streams per
// SIMD (a lone wave issues one instruction per ~5.5 cycles, csrc/ffn.hip) whose load / LayerNorm / epilogue phases fall under each
// other's MFMA blocks.
// Per tile t:  MFMA groups 0 .. 2 (5 k-steps each; the fragments of group g + 1 are requested behind ONE wait for group g) | hand-over:
// this wave's reads of tile t are complete and its share of tile t + 1 has landed, barrier -> stage t % 3 is free: DMA of tile t + 3,
// first fragments of tile t + 1 | group 3 | epilogue (+ residual, bf16, two 16-byte stores per lane).
#include <type_traits>

#include "common.h"
#include "gemm_common.h"
#include "ln_frag.h"
#include "mmgt_hip.h"

namespace {

constexpr int RC = 320, R_KS = RC / 16, R_TILE = R_KS * 1024, R_NST = 3, R_PW = R_KS / 4;    // 5 DMA pieces per wave and tile
constexpr int R_LW = 0, R_LG = R_NST * R_TILE, R_LB = R_LG + 2 * RC * 4, R_MAXN = 1920, R_LDS = R_LB + R_MAXN * 4;
static_assert(R_LDS <= 80 * 1024, "two workgroups per CU");

__device__ __forceinline__ f32x16 rmma(s16x8 a, s16x8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0); }

struct RowGemmArgs {
  const bf16_t* x; long ldx;
  const float* gamma; const float* beta; int norm, pe_div, pe_mod; float eps;   // norm 1: LayerNorm, beta row (row / pe_div) % pe_mod; 2: x * gamma[g] + beta[g], both tables indexed so (3: + SiLU)
  const char* wimg; const float* bias; const float* bias2; int bias2_rows;
  const bf16_t* res; long ldr;
  bf16_t* out; long ldo; int n1;                                           // normal tiles: columns [0, n1)
  bf16_t* out_t; int n_tok, npad;                                          // transposed tiles: columns [n1, N) -> out_t[row / n_tok][c - n1][row % n_tok]
  int M, N;
  unsigned long long* trace;                                               // DBG 5: [workgroup][32] shader-clock stamps of wave 0
};

template <bool RES, int DBG>
__global__ __launch_bounds__(256, 2)
void rowgemm320_kernel(const RowGemmArgs a) {
  extern __shared__ __attribute__((aligned(1024))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31, hh = lane >> 5;
  const long row0 = (long)blockIdx.x * 128;
  const long row = row0 + wid * 32 + r;
  const long rowc = row < a.M ? row : a.M - 1;
  const int nt = a.N / 32, nt1 = a.n1 / 32;
  int trace_n = 0;
  auto stamp = [&]() {
    if constexpr (DBG == 5) {
      if (a.trace && wid == 0 && lane == 0 && trace_n < 32) a.trace[(long)blockIdx.x * 32 + trace_n++] = __builtin_amdgcn_s_memtime();
    }
  };
  stamp();
  using std::integral_constant;
#define R_IC(v) integral_constant<int, (v)>{}
  auto for_range = [](auto LOc, auto HIc, auto&& fn) {
    constexpr int lo = decltype(LOc)::value, hi = decltype(HIc)::value;
    [&]<int... I>(std::integer_sequence<int, I...>) { (fn(integral_constant<int, lo + I>{}), ...); }(std::make_integer_sequence<int, (hi > lo ? hi - lo : 0)>{});
  };
  const __amdgpu_buffer_rsrc_t rw = dma_rsrc(a.wimg);
  const unsigned lane16 = (unsigned)lane * 16u;
  // this wave's share of tile t: the 5 contiguous 1-KiB pieces 5 wid .. 5 wid + 4 (one M0 / soffset setting per 4: instruction offsets);
  // tiles beyond the image take the poison offset (nothing is fetched), so every iteration issues the same number of pieces
  auto dma_tile = [&](int t, int stage_off) {
    const unsigned voff = (DBG != 1 && t < nt) ? lane16 : DMA_POISON;
    const int src = t * R_TILE + wid * (R_PW * 1024);
    char* dst = smem + R_LW + stage_off + wid * (R_PW * 1024);
    for_range(R_IC(0), R_IC(R_PW), [&](auto ic) {
      constexpr int i = decltype(ic)::value;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (__attribute__((address_space(3))) void*)(dst + (i / 4) * 4096), 16, (int)voff,
                                               src + (i / 4) * 4096, (i % 4) * 1024, 0);
    });
  };

  // ---- prologue.  Everything the workgroup needs from memory is requested up front, in ONE latency: the first three weight tiles
  // (LDS-DMA), the wave's 32 rows, the norm tables and the bias vectors (into registers), and only then is anything used.  (hipcc
  // waits vmcnt(0) at the first use of a plain load while an LDS-DMA is in flight, which is exactly what is wanted here; the order
  // rows -> tables (a load-use-store loop) -> barrier -> DMA paid the HBM latency three to five times in a row: stamps, 31 000 cycles
  // in front of the first MFMA.)
  dma_tile(0, 0);
  dma_tile(1, R_TILE);
  dma_tile(2, 2 * R_TILE);
  // the wave's 32 rows as MFMA fragments: lane (r, hh) holds channels 16 ks + 8 hh .. + 7 of row r
  s16x8 xf[R_KS];
  {
    const bf16_t* xr = a.x + rowc * a.ldx + 8 * hh;
#pragma unroll
    for (int ks = 0; ks < R_KS; ++ks) xf[ks] = *reinterpret_cast<const s16x8*>(xr + 16 * ks);
    float* lgb = reinterpret_cast<float*>(smem + R_LG);
    float* lb = reinterpret_cast<float*>(smem + R_LB);
    f32x4 tn = (f32x4)(0.f), tb[2] = {(f32x4)(0.f), (f32x4)(0.f)}, tb2[2] = {(f32x4)(0.f), (f32x4)(0.f)};
    const bool has_norm = a.norm && tid < 2 * RC / 4;
    if (has_norm) {
      const long grp = a.pe_mod > 1 ? (long)(((unsigned)row0 / (unsigned)a.pe_div) % (unsigned)a.pe_mod) * RC : 0;
      const float* beta = a.beta + grp;
      const float* gamma = a.gamma + (a.norm >= 2 ? grp : 0);
      tn = *reinterpret_cast<const f32x4*>(tid < RC / 4 ? gamma + 4 * tid : beta + 4 * (tid - RC / 4));
    }
    const float* b2 = a.bias2 ? a.bias2 + (long)((unsigned)row0 / (unsigned)a.bias2_rows) * a.N : nullptr;
#pragma unroll
    for (int i = 0; i < 2; ++i) {            // N <= 1920: at most two vectors of four columns per thread
      const int c = 4 * (tid + 256 * i);
      if (c < a.N) {
        if (a.bias) tb[i] = *reinterpret_cast<const f32x4*>(a.bias + c);
        if (b2) tb2[i] = *reinterpret_cast<const f32x4*>(b2 + c);
      }
    }
    if (DBG == 5) { stamp(); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); stamp(); }
    if (has_norm) *reinterpret_cast<f32x4*>(lgb + 4 * tid) = tn;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int c = 4 * (tid + 256 * i);
      if (c < a.N) *reinterpret_cast<f32x4*>(lb + c) = tb[i] + tb2[i];
    }
    __syncthreads();
    if (DBG == 5) stamp();
    if (a.norm == 1) layernorm_fragments<R_KS, true>(xf, lgb, hh, a.eps);
    else if (a.norm == 2) layernorm_fragments<R_KS, false>(xf, lgb, hh, a.eps);
    else if (a.norm == 3) layernorm_fragments<R_KS, false, true>(xf, lgb, hh, a.eps);
  }

  // fragment ring: [group parity][k-step of the group]; 4 groups of 5 k-steps per tile (an EVEN number of groups: the first group of
  // the next tile, requested at the hand-over, does not land on the registers of the last group of this one)
  constexpr int GK = 5, NG = R_KS / GK;
  s16x8 fa[2][GK];
  auto read_group = [&](const char* stage, auto Gc) {     // stage: this lane's 16 bytes of fragment 0 of the tile's ring stage
    constexpr int g = decltype(Gc)::value;
#pragma unroll
    for (int q = 0; q < GK; ++q) fa[g & 1][q] = *reinterpret_cast<const s16x8*>(stage + (g * GK + q) * 1024);
  };
  // all fragment reads but the N youngest have returned (the builtin, so that the compiler's own wait insertion adds nothing per MFMA;
  // gfx9 encoding: lgkmcnt in bits 11:8, the vmcnt / expcnt fields at their maxima)
  auto group_wait = [](auto Nc) { __builtin_amdgcn_s_waitcnt(0xC07F | (decltype(Nc)::value << 8)); };
  // Counted wait for the LDS-DMA of tile t + 1 at the hand-over of tile t.  vmcnt retires in issue order, stores included (gfx9 has
  // no separate store counter), so the count is everything this wave issued BEHIND those five pieces: the two stores of tiles t - 2 and
  // t - 1, the residual loads of tiles t - 1 and t, the five pieces of tile t + 2 -- 9 + 4 RES in the steady state, less for the first
  // two tiles.  (A fixed count of 5 + 2 RES is correct too, but it waits for the acknowledgement of the stores of the previous tile,
  // ~1 us.)
  auto hand_over = [&](int t) {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (DBG == 4) return;                                    // (ablation: no DMA wait, no barrier)
    if (t >= 2) wait_vmcnt<R_PW + 4 + (RES ? 4 : 0)>();
    else if (t == 1) wait_vmcnt<R_PW + 2 + (RES ? 4 : 0)>();
    else wait_vmcnt<R_PW + (RES ? 2 : 0)>();
    __builtin_amdgcn_s_barrier();
  };
  const __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc(a.out, 0, (int)((long)a.M * a.ldo * 2), 0x00020000);
  // Store layout of a normal tile (round 5).  After the v_permlane32_swap of the epilogue lane (r, hh) owns 8 columns of ITS row: an instruction
  // then covers 32 rows x 32 bytes = 32 quarter lines, and the CU's address path -- which the weight DMA shares -- pays per line touched
  // (ablations: the launch at N = 960 is 167 us, 123 without its stores, 98 without its MFMAs: the stores and the MFMAs did not overlap).
  // v_permlane16_swap of the two 16-byte vectors of a tile hands lanes r >= 16 the SECOND vector of row r - 16 and lanes r < 16 the FIRST vector
  // of row r + 16: store A = rows 0 .. 15 x 64 bytes, store B = rows 16 .. 31 x 64 bytes -- 16 half lines per instruction.
  const unsigned obase = (unsigned)((row0 + wid * 32 + (r & 15)) * a.ldo + 16 * (r >> 4) + 8 * hh) * 2u;   // store A; B: + 16 rows (rows >= M: beyond num_records -> dropped)
  const unsigned ostep16 = (unsigned)(16 * a.ldo) * 2u;
  const bf16_t* rr = RES ? a.res + rowc * a.ldr + 8 * hh : nullptr;
  // transposed tiles: this lane's column is c = r, its 8-row groups start at token tok0 + 16 (k / 2) + 8 hh
  const long bt = (unsigned)row0 / (unsigned)a.n_tok;            // (M < 2^31: 32-bit division)
  const int tok0 = (int)(row0 - bt * a.n_tok) + wid * 32;
  bf16_t* ot = a.out_t ? a.out_t + (bt * (a.N - a.n1) + r) * (long)a.npad + tok0 + 8 * hh : nullptr;

  stamp();
  wait_vmcnt<2 * R_PW>();                    // tile 0 has landed (tiles 1, 2 in flight)
  __builtin_amdgcn_s_barrier();
  stamp();
  int so_cur = 0, so_nxt = R_TILE;           // ring stage (byte offset) of tile t / t + 1; tile t + 3 goes where tile t was
  read_group(smem + R_LW + lane * 16, R_IC(0));
  for (int t = 0; t < nt; ++t) {
    const char* pc = smem + R_LW + lane * 16 + so_cur;
    const char* pn = smem + R_LW + lane * 16 + so_nxt;
    u32x4 rv[2];
    if constexpr (RES) {
#pragma unroll
      for (int k = 0; k < 2; ++k) rv[k] = *reinterpret_cast<const u32x4*>(rr + 32 * t + 16 * k);
    }
    auto tile = [&](auto TRc) {
      constexpr bool TR = decltype(TRc)::value;
      // accumulator = bias: normal tile: register i <-> column 32 t + 8 (i >> 2) + 4 hh + (i & 3); transposed: the lane's column
      f32x16 acc;
      const float* lb = reinterpret_cast<const float*>(smem + R_LB) + 32 * t;
      if constexpr (!TR) {
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
          const f32x4 b = *reinterpret_cast<const f32x4*>(lb + 4 * hh + 8 * g4);
#pragma unroll
          for (int e = 0; e < 4; ++e) acc[4 * g4 + e] = b[e];
        }
      } else {
        acc = (f32x16)(lb[r]);
      }
      for_range(R_IC(0), R_IC(NG), [&](auto gc) {
        constexpr int g = decltype(gc)::value;
        if constexpr (g == NG - 1) {
          hand_over(t);
          dma_tile(t + 3, so_cur);
          read_group(pn, R_IC(0));             // (t + 1 == nt: reads whatever the stage holds, never used)
        } else {
          read_group(pc, R_IC(g + 1));
        }
        group_wait(R_IC(GK));
        __builtin_amdgcn_sched_barrier(0);
        if (DBG != 2) {
#pragma unroll
          for (int q = 0; q < GK; ++q) acc = TR ? rmma(xf[GK * g + q], fa[g & 1][q], acc) : rmma(fa[g & 1][q], xf[GK * g + q], acc);
        }
        __builtin_amdgcn_sched_barrier(0);
      });
      if constexpr (RES) {     // (pins the unpacking of the residual vectors down here: hoisted to their loads at the top of the tile, it would
#pragma unroll           //  wait for them -- and for every LDS-DMA in front of them -- before the first MFMA)
        for (int k = 0; k < 2; ++k) asm volatile("" : "+v"(rv[k]));
      }
      // ---- epilogue.  Register group k (registers 4 k .. 4 k + 3) is D rows 8 k + 4 hh + (0..3); v_permlane32_swap of groups (k, k + 1)
      // gives lane hh = 0 rows 8 k .. 8 k + 7 and lane hh = 1 rows 8 k + 8 .. + 15 (cdna_hip_programming.md T21): 16 bytes per store
      u32x4 pkv[2];
#pragma unroll
      for (int k = 0; k < 4; k += 2) {
        float o8[8];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(acc[4 * k + e]), __float_as_uint(acc[4 * k + 4 + e]), false, false);
          o8[e] = __uint_as_float(sw[0]);
          o8[4 + e] = __uint_as_float(sw[1]);
        }
        if constexpr (RES) {
          union { u32x4 q; bf16_t e[8]; } r8;
          r8.q = rv[k / 2];
#pragma unroll
          for (int e = 0; e < 8; ++e) o8[e] += bf16_to_f32(r8.e[e]);
        }
        pkv[k / 2] = (u32x4){pack_bf16x2(o8[0], o8[1]), pack_bf16x2(o8[2], o8[3]), pack_bf16x2(o8[4], o8[5]), pack_bf16x2(o8[6], o8[7])};
      }
      if (!(DBG == 3 && pkv[0][0] != 0x12345678u)) {             // (DBG 3: ablation without stores)
        if constexpr (!TR) {
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const auto sw = __builtin_amdgcn_permlane16_swap(pkv[0][e], pkv[1][e], false, false);
            pkv[0][e] = sw[0];
            pkv[1][e] = sw[1];
          }
          __builtin_amdgcn_raw_buffer_store_b128(pkv[0], ro, (int)(obase + 2u * (32 * t)), 0, MMGT_ST_AUX);
          __builtin_amdgcn_raw_buffer_store_b128(pkv[1], ro, (int)(obase + ostep16 + 2u * (32 * t)), 0, MMGT_ST_AUX);
        } else {
          *reinterpret_cast<u32x4*>(ot + (long)(32 * (t - nt1)) * a.npad) = pkv[0];
          *reinterpret_cast<u32x4*>(ot + (long)(32 * (t - nt1)) * a.npad + 16) = pkv[1];
        }
      }
    };
    if (t < nt1) tile(std::false_type{}); else tile(std::true_type{});
    so_cur = so_nxt;
    so_nxt = so_nxt == (R_NST - 1) * R_TILE ? 0 : so_nxt + R_TILE;
    if (t & 1) stamp();
  }
#undef R_IC
}

int g_rowgemm_dbg = 0;
unsigned long long* g_rowgemm_trace = nullptr;

}  // namespace

void mmgt_rowgemm_set_dbg(int v) { g_rowgemm_dbg = v; }
// Debug (tools/trace_rowgemm.py): device buffer of u64 [workgroups][32] for the stamps of the rowgemm_dbg = 5 build; NULL = off.
extern "C" void mmgt_rowgemm_set_trace(void* p) { g_rowgemm_trace = reinterpret_cast<unsigned long long*>(p); }

extern "C" long mmgt_rowgemm320_image_bytes(int N) {
  if (N <= 0 || N % 32 || N > R_MAXN) return -1;
  return (long)N * RC * 2;
}

extern "C" int mmgt_rowgemm320(const void* x, long ldx, int norm, const float* ln_gamma, const float* ln_beta, int pe_div, int pe_mod, float eps,
                               const void* wimg, const float* bias, const float* bias2, int bias2_rows, const void* residual, long ldr,
                               void* out, long ldo, int n1, void* out_t, int n_tok, int npad, int M, int N, int dtype, void* stream) {
  MMGT_CHECK(x && wimg, "rowgemm320: null pointer");
  MMGT_CHECK(dtype == MMGT_BF16, "rowgemm320: bf16 only (the fp32-I/O mode runs LayerNorm / GEMM)");
  MMGT_CHECK(mmgt_rowgemm320_image_bytes(N) > 0, "rowgemm320: N = 32 .. %d in steps of 32 (got %d)", R_MAXN, N);
  MMGT_CHECK(n1 >= 0 && n1 <= N && n1 % 32 == 0, "rowgemm320: n1 = %d must be a multiple of 32 within N = %d", n1, N);
  MMGT_CHECK(M > 0 && ldx >= RC && ldx % 8 == 0, "rowgemm320: bad M = %d or ldx = %ld", M, ldx);
  MMGT_CHECK(norm >= 0 && norm <= 3 && (norm != 0) == (ln_gamma != nullptr) && (norm != 0) == (ln_beta != nullptr),
             "rowgemm320: norm = %d (0 none, 1 LayerNorm, 2 scale / shift tables, 3 the tables + SiLU) and gamma / beta must come together", norm);
  MMGT_CHECK(!norm || pe_mod <= 1 || (pe_div > 0 && pe_div % 128 == 0), "rowgemm320: pe_div = %d must be a multiple of 128", pe_div);
  MMGT_CHECK(!bias2 || (bias2_rows > 0 && bias2_rows % 128 == 0), "rowgemm320: bias2_rows = %d must be a multiple of 128", bias2_rows);
  MMGT_CHECK(n1 == 0 || (out && ldo >= n1 && ldo % 8 == 0 && (long)M * ldo * 2 < (1l << 31)),
             "rowgemm320: normal output: null, bad ldo = %ld or beyond the 2 GiB range of a buffer resource (split the rows)", ldo);
  MMGT_CHECK(n1 == N || (out_t && n_tok > 0 && n_tok % 128 == 0 && M % n_tok == 0 && npad >= n_tok && npad % 8 == 0),
             "rowgemm320: transposed output needs n_tok %% 128 == 0, M %% n_tok == 0, npad >= n_tok, npad %% 8 == 0 (n_tok %d, npad %d)", n_tok, npad);
  MMGT_CHECK(!residual || (n1 == N && ldr >= N && ldr % 8 == 0), "rowgemm320: a residual needs normal tiles only and ldr >= N, ldr %% 8 == 0");
  MMGT_CHECK((((uintptr_t)x | (uintptr_t)residual | (uintptr_t)out | (uintptr_t)out_t | (uintptr_t)wimg | (uintptr_t)bias | (uintptr_t)bias2) & 15) == 0 &&
                 (!ln_gamma || (((uintptr_t)ln_gamma | (uintptr_t)ln_beta) & 15) == 0),
             "rowgemm320: pointers must be 16-byte aligned");
  RowGemmArgs a;
  a.x = (const bf16_t*)x; a.ldx = ldx;
  a.gamma = ln_gamma; a.beta = ln_beta; a.norm = norm; a.pe_div = pe_div > 0 ? pe_div : 1; a.pe_mod = pe_mod; a.eps = eps;
  a.wimg = (const char*)wimg; a.bias = bias; a.bias2 = bias2; a.bias2_rows = bias2_rows > 0 ? bias2_rows : 1;
  a.res = (const bf16_t*)residual; a.ldr = ldr;
  a.out = (bf16_t*)out; a.ldo = ldo; a.n1 = n1;
  a.out_t = (bf16_t*)out_t; a.n_tok = n_tok > 0 ? n_tok : 128; a.npad = npad;
  a.M = M; a.N = N;
  a.trace = g_rowgemm_trace;
#ifdef MMGT_ABLATE   // timing ablations 1 - 4 (results are garbage): only in libmmgt_hip_abl.so (`make abl`), never in the product library
  const int d = g_rowgemm_dbg;
  auto kern = residual ? (d == 1 ? rowgemm320_kernel<true, 1> : d == 2 ? rowgemm320_kernel<true, 2> : rowgemm320_kernel<true, 0>)
                       : (d == 1 ? rowgemm320_kernel<false, 1> : d == 2 ? rowgemm320_kernel<false, 2> : d == 3 ? rowgemm320_kernel<false, 3>
                          : d == 4 ? rowgemm320_kernel<false, 4> : d == 5 ? rowgemm320_kernel<false, 5> : rowgemm320_kernel<false, 0>);
#else                // (5 = the stamped build of tools/trace_rowgemm.py: correct results, needs the trace buffer)
  const int d = g_rowgemm_dbg == 5 && !residual && g_rowgemm_trace ? 5 : 0;
  auto kern = residual ? rowgemm320_kernel<true, 0> : d == 5 ? rowgemm320_kernel<false, 5> : rowgemm320_kernel<false, 0>;
#endif
  static bool attr[2][6] = {};
  if (!attr[residual != nullptr][d]) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, R_LDS) != hipSuccess) {
      mmgt_set_error("rowgemm320: cannot reserve %d bytes of LDS", R_LDS);
      return 2;
    }
    attr[residual != nullptr][d] = true;
  }
  const unsigned grid = (unsigned)((M + 127) / 128);
  hipLaunchKernelGGL(kern, dim3(grid), dim3(256), (size_t)(R_LB + N * 4), (hipStream_t)stream, a);
  MMGT_LAUNCH_CHECK();
  return 0;
}
