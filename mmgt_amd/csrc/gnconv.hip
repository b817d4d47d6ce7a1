// GroupNorm + SiLU + conv3x3 in ONE launch for the VAE's 128-channel levels (bf16, Cin = 128, Cout = 128, stride 1, padding 1, gfx950):
//
//   out[n, y, x, :] = bias + sum_{ky, kx} W[ky, kx] . silu( x[n, y + ky - 1, x + kx - 1, :] * scale[n, :] + shift[n, :] )   (+ residual)
//
// with (scale, shift) the per-(image, channel) tables of mmgt_groupnorm_affine (the statistics pass of GroupNorm alone).  Replaces the
// `hip.groupnorm(silu=True)` -> `hip.conv3x3` pair of mmgt_amd/vae.py::_resnet / decode_nhwc at decoder.up_blocks.3 and conv_norm_out
// (reference: diffusers `ResnetBlock2D.forward` = norm -> nonlinearity -> conv as called by `AutoencoderKL.decode` from
// src/pipelines/pipeline_pose2vid_long.py:112-125).  At 8 x 512 x 512 x 128 the pair cost 307 us (GroupNorm: 537 MB read twice, written once)
// + 735 us (implicit-GEMM conv on the 128 x 128 tile: with only 128 output columns the gathered A operand is staged 9 times per tile and
// is 2/3 of the L2 -> LDS traffic, 64 FLOP per staged byte).
//
// Here a workgroup owns a 16 x 16 pixel tile of one image.  Its 18 x 18 x 128 halo is read ONCE from HBM into registers, normalised, SiLU'd,
// rounded to bf16 (the rounding point of the unfused pair) and written to LDS (pixel-major, 272-byte pixel stride: the 16 lanes of a
// ds_read_b128 group then fall on 15 different 16-byte bank slots); out-of-image pixels are zeros AFTER the activation, as the conv's padding
// wants.  The nine taps' A fragments (v_mfma_f32_16x16x32_bf16: lane (lm, lq) = pixel lm of an image row, channels 32 ks + 8 lq .. + 7) are
// read straight from the halo with compile-time offsets -- no im2col staging at all.  The weights, a fragment-major image of 18 half-taps of
// 16 KiB (mmgt_amd/packing.py::pack_gnconv), stream through a 4-slot LDS ring by LDS-DMA three half-taps ahead, ONE barrier per half-tap
// (32 MFMAs per wave) placed between its two k-steps; the stream runs across tile boundaries (203 FLOP per staged byte).
// 8 waves = 4 (rows) x 2 (columns): a wave owns 4 image rows x 16 pixels x 64 output channels = 4 x 4 accumulator tiles.
// The halo of the NEXT tile is requested during half-tap 2 and normalised a vector per half-tap in the shadow of the MFMAs, so a tile
// boundary costs the epilogue, two barriers and the LDS writes.
#include <type_traits>
#include <utility>

#include "common.h"
#include "gemm_common.h"
#include "mmgt_hip.h"

namespace {

typedef __attribute__((ext_vector_type(4))) float acc4;

// A unit of work = one PHASE of one tile: 128 input channels of the tile's halo, 18 half-taps.  Cin = 128 NPH: the phases of a tile accumulate
// into the same accumulators (a 256-channel halo does not fit the LDS; its two halves take turns), the epilogue follows the last one.
constexpr int GC_PC = 128, GC_T = 16, GC_HP = GC_T + 2;                     // channels per phase, tile edge, halo edge
constexpr int GC_PSTR = GC_PC * 2 + 16;                                     // bytes per halo pixel in LDS
constexpr int GC_HALO = GC_HP * GC_HP * GC_PSTR;                            // 88 128
constexpr int GC_NV = GC_HP * GC_HP * (GC_PC / 8);                          // 16-byte vectors of a halo: 5184 = 10 x 512 + 64
constexpr int GC_NVL = (GC_NV + 511) / 512;                                 // vectors per lane (the 11th: lanes 0 .. 63 only)
constexpr int GC_NSLOT = 4, GC_NSTEP = 18;                                  // ring slots; half-taps (64 input channels x Cout) per phase
constexpr int GC_L_RING = (GC_HALO + 1023) / 1024 * 1024, GC_L_BIAS = GC_L_RING + GC_NSLOT * 16 * 1024, GC_L_TAB = GC_L_BIAS + 128 * 4, GC_L_STAT = GC_L_TAB + 2 * GC_PC * 4, GC_LDS = GC_L_STAT + 8 * 4 * 8 * 4;
static_assert(GC_LDS <= 160 * 1024 && 6 + GC_NVL <= GC_NSTEP, "LDS / schedule");

struct GcArgs {
  const bf16_t* x;       // (nb, H, W, Cin)
  const float* scale;    // (nb, Cin)
  const float* shift;    // (nb, Cin)
  const char* wimg;      // pack_gnconv image (NPH x 18 half-taps of 128 Cout bytes)
  const float* bias;     // (Cout) or null
  const bf16_t* res;     // (nb, H, W, Cout) or null
  bf16_t* out;           // (nb, H, W, Cout)
  int nb, H, W, tiles_x, tiles_per_img, ntiles;
  int ldo;               // channels per pixel of out / residual (>= Cout: a launch may write a 128-channel half of a wider tensor)
  float* stats;          // or null: per (tile, 4-channel quad of the launch's Cout) the pair (sum, sum of squares) of the STORED values, [tile][ldo / 4][2]
};

// (hipcc 7.2 rotates accumulator tiles between register ranges in front of the epilogue and then emits MFMAs whose vDst overlaps SrcC by two
// registers, e.g. v[120:123] <- ... + v[122:125]; tools/micro/mfma_overlap.hip shows the hardware computes those correctly -- the wrong outputs
// this file once produced came from the store hazard described at the epilogue's stores, not from them.)
__device__ __forceinline__ void gc_mma(acc4& c, s16x8 a, s16x8 b) { c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }
__device__ __forceinline__ float gc_lo(unsigned u) { return __uint_as_float(u << 16); }
__device__ __forceinline__ float gc_hi(unsigned u) { return __uint_as_float(u & 0xffff0000u); }

template <int LO, int... I, typename F>
__device__ __forceinline__ void gc_for_impl(std::integer_sequence<int, I...>, F&& fn) { (fn(std::integral_constant<int, LO + I>{}), ...); }
template <int LO, int HI, typename F>
__device__ __forceinline__ void gc_for(F&& fn) { gc_for_impl<LO>(std::make_integer_sequence<int, (HI > LO ? HI - LO : 0)>{}, static_cast<F&&>(fn)); }

// NPH = Cin / 128 (1, 2); COUT = 128 (8 waves = 4 x 2: 4 image rows x 64 channels each) or 64 (8 x 1: 2 image rows x 64 channels).
// ABL (mmgt_tune("gnconv_abl", bit) of the -DMMGT_ABLATE build, timing only -- results are garbage): 1 no MFMAs, 2 no weight DMA after the prologue, 4 no halo loads /
// normalisation / LDS writes after the first tile, 8 no hand-over wait, 16 no epilogue (residual loads, stores), 32 no hand-over barrier,
// 64 halo loads but no normalisation / LDS writes, 128 normalisation / LDS writes but no halo loads.
// ST: the epilogue also emits, per tile and 4-channel quad, (sum, sum of squares) of the values it stores -- the statistics of the NEXT GroupNorm
// (mmgt_gn_stats_finalize folds the tiles of an image in a fixed order), so that the pass over the tensor that recomputes them is not needed.
template <int NPH, int COUT, bool RES, int ABL, bool ST = false>
__global__ __launch_bounds__(512, 2) void gnconv_kernel(const GcArgs a) {
  static_assert(!ST || COUT == 128, "statistics: 128 output channels per launch");
  static_assert((NPH == 1 || NPH == 2) && (COUT == 64 || COUT == 128), "shape");
  constexpr int CIN = GC_PC * NPH, WN = COUT / 64, WM = 8 / WN, RT = GC_T / WM;   // waves along the channels / the rows; image rows per wave
  constexpr int NCT = COUT / 16, SLOT = 64 * COUT * 2, PPW = SLOT / 1024 / 8;     // 16-channel tiles; bytes per half-tap; DMA pieces per wave and half-tap
  extern __shared__ __attribute__((aligned(1024))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lm = lane & 15, lq = lane >> 4;
  const int wm = wid / WN, wn = wid % WN;                  // image rows RT wm .. of the tile, output channels 64 wn .. + 63
  const int G = gridDim.x;
  const int my_tiles = (a.ntiles - (int)blockIdx.x + G - 1) / G;
  const int my_units = my_tiles * NPH;
  const int total = my_units * GC_NSTEP;                   // half-taps this workgroup consumes

  // XCD-aware tile order (gemm.hip): XCD x = v & 7 walks a contiguous run of the tile sequence, so horizontally adjacent tiles -- which share
  // two halo columns -- are worked on one L2 at about the same time
  int t_cur = 0;                                           // position of the last decoded tile in the (image, row, column) order
  auto decode = [&](int v, int& n, int& ty, int& tx) {
    const int q = a.ntiles >> 3, r = a.ntiles & 7, xc = v & 7;
    const int t = (xc < r ? xc * (q + 1) : r * (q + 1) + (xc - r) * q) + (v >> 3);
    t_cur = t;
    n = t / a.tiles_per_img;
    const int rem = t - n * a.tiles_per_img;
    ty = rem / a.tiles_x;
    tx = rem - ty * a.tiles_x;
  };

  // ---- weight stream: half-tap g (counted over the workgroup's units) is image chunk g % (18 NPH) and lives in ring slot g % 4
  const __amdgpu_buffer_rsrc_t rW = dma_rsrc(a.wimg);
  const unsigned w_voff = (unsigned)(lane * 16 + wid * PPW * 1024);
  // Beyond the last half-tap the pieces still go out, against the poison offset (zeros into a slot nobody reads): every wait count below is
  // then a compile-time constant on every path, and hipcc's own waits for the halo registers see the same number of younger operations
  // whether or not the stream has ended (with conditional pieces it assumed none and waited for the newest pieces at every use).
  auto issue_w = [&](int g) {
    const int chunk = g % (GC_NSTEP * NPH), slot = g & (GC_NSLOT - 1);
    const unsigned vo = g < total ? w_voff : DMA_POISON;
#pragma unroll
    for (int u = 0; u < PPW; ++u)
      blds16(rW, vo, chunk * SLOT + u * 1024, smem + GC_L_RING + slot * SLOT + (wid * PPW + u) * 1024);
  };

  // ---- halo: vector v = tid + 512 i is pixel p = v >> 4 (row-major over the 18 x 18 halo), channel octet o = tid & 15 of the phase's 128
  const __amdgpu_buffer_rsrc_t rX = dma_rsrc(a.x), rO = dma_rsrc(a.out), rR = dma_rsrc(a.res ? a.res : a.x);
  const int oct = tid & 15;
  u32x4 hv[GC_NVL];
  unsigned hmask = 0;                                      // bit i: vector i lies inside the image
  // scale | shift of the (image, phase) of the halo in flight: ONE LDS-DMA piece of wave 0 (lanes 0 .. 31: 128 scales, lanes 32 .. 63: 128 shifts;
  // the two tables are one allocation, shift = scale + nb Cin), issued with the halo loads and published by the barrier of half-tap 5
  const __amdgpu_buffer_rsrc_t rT = dma_rsrc(a.scale);
  int iter_ = 0;
  auto load_halo = [&](int v, int ph) {
    int n, ty, tx;
    decode(v, n, ty, tx);
    if (wid == 0) blds16(rT, (unsigned)((((lane >> 5) * a.nb + n) * CIN + ph * GC_PC + (lane & 31) * 4) * 4), 0, smem + GC_L_TAB);
    hmask = 0;
    int p0 = tid >> 4, o16 = oct * 16 + ph * (GC_PC * 2);
    asm volatile("" : "+v"(p0), "+v"(o16));                 // opaque: the per-vector coordinates are recomputed here, not kept in 30 registers across the tile
#pragma unroll
    for (int i = 0; i < GC_NVL; ++i) {
      const int p = p0 + 32 * i;
      const int hy = (p * 3641) >> 16, hx = p - GC_HP * hy;               // p / 18, exact for p < 3 000
      const int y = ty * GC_T - 1 + hy, x = tx * GC_T - 1 + hx;
      const bool ok = p < GC_HP * GC_HP && y >= 0 && y < a.H && x >= 0 && x < a.W;
      const unsigned off = (unsigned)(((n * a.H + y) * a.W + x) * (CIN * 2) + o16);
      if (!(ABL & 128) || iter_ == 0) hv[i] = __builtin_amdgcn_raw_buffer_load_b128(rX, (int)(ok ? off : DMA_POISON), 0, 0);
      hmask |= ok ? 1u << i : 0u;
    }
  };
  // dword J of hv[I] <- bf16( silu( . * scale + shift ) ), zeros outside the image.  Branch-free (a select on the mask made hipcc branch around
  // the transcendentals, one basic block per pair, nothing interleaved with the MFMAs); the table is re-read per dword (two ds_read_b64)
  // instead of living in 16 registers.
  auto norm_dword = [&](auto Ic, auto Jc) {
    constexpr int I = decltype(Ic)::value, J = decltype(Jc)::value;
    const unsigned m = 0u - ((hmask >> I) & 1u);
    int tofs = GC_L_TAB + 32 * oct + 8 * J;
    asm volatile("" : "+v"(tofs));                          // (opaque: no common subexpressions across calls)
    const f32x2 sc = *reinterpret_cast<const f32x2*>(smem + tofs), sh = *reinterpret_cast<const f32x2*>(smem + tofs + GC_PC * 4);
    // silu(t) = t / (1 + e^-t) with v_exp_f32 / v_rcp_f32 (1 ulp) instead of the IEEE division sequence: 10 issue slots of 4 cycles per
    // element beside MFMAs that hold the SIMD's vector issue for 8 of their 16 cycles.  Scalar f32 arithmetic on purpose (the file is built
    // with -fno-slp-vectorize): packed f32 instructions cost ~25 cycles each beside MFMAs (MI355X_MICROARCH.md, per-instruction constants).
    const float t0 = fmaf(gc_lo(hv[I][J]), sc[0], sh[0]), t1 = fmaf(gc_hi(hv[I][J]), sc[1], sh[1]);
    const float v0 = t0 * __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(t0 * -1.4426950408889634f));
    const float v1 = t1 * __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(t1 * -1.4426950408889634f));
    unsigned pk = pack_bf16x2(v0, v1) & m;
    asm volatile("" : "+v"(pk));                            // pinned here (between two MFMA statements): hipcc otherwise sinks the arithmetic to the tile's end, where hv is stored
    hv[I][J] = pk;
  };
  auto norm_vec = [&](auto Ic) { gc_for<0, 4>([&](auto jc) { norm_dword(Ic, jc); }); };
  const int h_wbase = (tid >> 4) * GC_PSTR + oct * 16;
  auto write_halo = [&]() {
#pragma unroll
    for (int i = 0; i < GC_NVL; ++i)
      if (i + 1 < GC_NVL || tid < GC_NV - 512 * (GC_NVL - 1))
        *reinterpret_cast<u32x4*>(smem + h_wbase + i * 32 * GC_PSTR) = hv[i];
  };

  // ---- fragment addressing
  const int a_base = (RT * wm * GC_HP + lm) * GC_PSTR + lq * 16;         // + ((i + ky) 18 + kx) 272 + 128 kh + 64 ks2: compile-time
  const int w_base = GC_L_RING + wn * 4 * 1024 + lane * 16;              // + slot + (NCT ks2 + j) 1 KiB
  s16x8 fa[2][RT], fw[2][4];
  auto read_a = [&](auto Sc, auto Kc, s16x8 (&f)[RT]) {
    constexpr int S = decltype(Sc)::value, KS2 = decltype(Kc)::value, tap = S / 2, kh = S & 1, ky = tap / 3, kx = tap % 3;
#pragma unroll
    for (int i = 0; i < RT; ++i)
      f[i] = *reinterpret_cast<const s16x8*>(smem + a_base + ((i + ky) * GC_HP + kx) * GC_PSTR + kh * 128 + KS2 * 64);
  };
  auto read_w = [&](int slot, auto Kc, s16x8 (&f)[4]) {
    constexpr int KS2 = decltype(Kc)::value;
    const char* p = smem + w_base + slot * SLOT + KS2 * NCT * 1024;
#pragma unroll
    for (int j = 0; j < 4; ++j) f[j] = *reinterpret_cast<const s16x8*>(p + j * 1024);
  };

  // ---- prologue: bias -> LDS, the first three half-taps, the first halo
  if (tid < COUT) reinterpret_cast<float*>(smem + GC_L_BIAS)[tid] = a.bias ? a.bias[tid] : 0.f;
  int gstep = 0;                                           // half-taps consumed so far
  if (my_units > 0) {
    load_halo(blockIdx.x, 0);
    for (int g = 0; g < GC_NSLOT - 1; ++g) issue_w(g);
    wait_vmcnt<(GC_NSLOT - 1) * PPW>();                    // the table (and the halo) have landed
    __builtin_amdgcn_s_barrier();
    gc_for<0, GC_NVL>([&](auto ic) { norm_vec(ic); });
    write_halo();
    wait_vmcnt<(GC_NSLOT - 2) * PPW>();                    // half-tap 0 has landed
  }
  __builtin_amdgcn_s_waitcnt(0xC07F);                      // lgkmcnt(0): the LDS stores above
  __builtin_amdgcn_s_barrier();
  if (my_units > 0) {
    read_a(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{}, fa[0]);
    read_w(0, std::integral_constant<int, 0>{}, fw[0]);
  }

  int vt = blockIdx.x, ph = 0;
  acc4 acc[RT][4];
  for (int unit = 0; unit < my_units; ++unit) {
    int n, ty, tx;
    decode(vt, n, ty, tx);
    const int t_this = t_cur;
    const bool has_next = unit + 1 < my_units && !(ABL & 4);
    const bool last_ph = NPH == 1 || ph == NPH - 1, first_ph = NPH == 1 || ph == 0;
    const int nvt = last_ph ? vt + G : vt, nph = last_ph ? 0 : ph + 1;
    const bool stored = unit > 0 && first_ph;              // the previous unit ended in an epilogue: its stores (and residual loads) are in the vmcnt queue

    if (first_ph) {
      const acc4* lb = reinterpret_cast<const acc4*>(smem + GC_L_BIAS) + wn * 16 + lq;   // the lane's columns 16 j + 4 lq + r
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const acc4 b = lb[4 * j];
#pragma unroll
        for (int i = 0; i < RT; ++i) acc[i][j] = b;
      }
      if constexpr (RES && NPH > 1) {
        // two phases: the residual enters HERE, in the accumulators' own layout (4 channels = 8 bytes per tile), not in the epilogue -- there
        // is no register to spare for the epilogue's vectors while the accumulators live across the phase loop
        const unsigned roff = (unsigned)((((n * a.H + ty * GC_T + RT * wm) * a.W + tx * GC_T + lm) * a.ldo + wn * 64 + 4 * lq) * 2);
        const int rrow = a.W * a.ldo * 2;
#pragma unroll
        for (int i = 0; i < RT; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const u32x2 rr = __builtin_amdgcn_raw_buffer_load_b64(rR, (int)roff + 32 * j, i * rrow, 0);
            acc[i][j][0] += gc_lo(rr[0]);
            acc[i][j][1] += gc_hi(rr[0]);
            acc[i][j][2] += gc_lo(rr[1]);
            acc[i][j][3] += gc_hi(rr[1]);
          }
      }
    }

    // residual vectors of the epilogue (the lane's 8 channels after the swap there), requested under the tile's last MFMAs (earlier they do not fit the register file)
    const int cofs = wn * 64 + 16 * (lq & 1) + 8 * (lq >> 1);
    const unsigned eoff = (unsigned)((((n * a.H + ty * GC_T + RT * wm) * a.W + tx * GC_T + lm) * a.ldo + cofs) * 2);   // byte offset of (row 0, pair 0)
    const int erow = a.W * a.ldo * 2;                                                                                // bytes per image row
    u32x4 rv[RES ? RT : 1][2];

    gc_for<0, GC_NSTEP>([&](auto sc_) {
      constexpr int S = decltype(sc_)::value;
      const int slot = gstep & (GC_NSLOT - 1), nslot = (gstep + 1) & (GC_NSLOT - 1);
      // second k-step's fragments
      read_a(sc_, std::integral_constant<int, 1>{}, fa[1]);
      read_w(slot, std::integral_constant<int, 1>{}, fw[1]);
      // a vector of the next unit's halo per half-tap, half of it under each k-step's MFMAs (unconditional: without a next unit it works on
      // stale registers; a uniform branch would put it in a basic block of its own, in front of the MFMAs instead of between them)
      constexpr bool NORM = S >= 6 && S < 6 + GC_NVL && !(ABL & 64);
      constexpr int NV = NORM ? S - 6 : 0;                 // the halo vector of this half-tap
      // the MFMAs of a k-step, two dwords of the halo vector between them
      auto burst = [&](s16x8 (&fw_)[4], s16x8 (&fa_)[RT], auto J0c) {
        constexpr int J0 = decltype(J0c)::value;
        gc_for<0, RT>([&](auto ic) {
          constexpr int i = decltype(ic)::value;
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            if constexpr (!(ABL & 1)) gc_mma(acc[i][j], fw_[j], fa_[i]);
            else acc[i][j][0] += __builtin_bit_cast(float, (int)fw_[j][0] ^ (int)fa_[i][0]);
          }
          if constexpr (NORM && i == 0) norm_dword(std::integral_constant<int, NV>{}, std::integral_constant<int, J0>{});
          if constexpr (NORM && i == RT / 2) norm_dword(std::integral_constant<int, NV>{}, std::integral_constant<int, J0 + 1>{});
        });
      };
      // (scheduling hint: one MFMA, then a few of the normalisation's VALU instructions, ...; then the region ends)
      auto interleave = [&]() {
        if constexpr (NORM) {
#pragma unroll
          for (int k = 0; k < 4 * RT; ++k) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x002, 12 / RT, 0);
          }
        }
        __builtin_amdgcn_sched_barrier(0);
      };
      burst(fw[0], fa[0], std::integral_constant<int, 0>{});
      interleave();
      // ---- hand-over: half-tap gstep + 1 has landed (this wave's pieces; the barrier collects the others').  vmcnt retires in order, so the
      // count is the operations YOUNGER than those pieces: the pieces of half-tap gstep + 2, plus -- half-taps 0, 1 behind an epilogue -- its
      // stores (and the residual loads of half-tap 17), plus -- half-taps 3, 4 -- the halo loads issued in half-tap 2.
      constexpr int NST = 2 * RT, NHL = GC_NVL;   // (wave 0 also issued the table piece: it waits for one operation more than it must)
      if constexpr (ABL & (2 | 8)) {
      } else if constexpr (S <= 1) {
        constexpr int NRL = RES && NPH == 1 ? 2 * RT : 0;
        if (stored) wait_vmcnt<PPW + NST + NRL>(); else wait_vmcnt<PPW>();
      } else if constexpr (S == 3 || S == 4) {
        if (has_next) wait_vmcnt<PPW + NHL>(); else wait_vmcnt<PPW>();
      } else {
        wait_vmcnt<PPW>();
      }
      // (half-tap 5: the wait above has seen the halo loads and the table piece land; this barrier publishes the table)
      if constexpr (!(ABL & 32)) __builtin_amdgcn_s_barrier();
      if constexpr (!(ABL & 2)) issue_w(gstep + GC_NSLOT - 1);   // into the slot half-tap gstep - 1 left
      if constexpr (S == 2) {
        if (has_next) { iter_ = 1; load_halo(nvt, nph); }
      }
      if constexpr (S == GC_NSTEP - 1 && RES && NPH == 1) {
        if (last_ph) {
#pragma unroll
          for (int i = 0; i < RT; ++i)
#pragma unroll
            for (int jp = 0; jp < 2; ++jp) rv[i][jp] = __builtin_amdgcn_raw_buffer_load_b128(rR, (int)eoff + 64 * jp, i * erow, 0);
        }
      }
      // first k-step's fragments of the next half-tap (the next unit's A fragments wait for its halo)
      if constexpr (S + 1 < GC_NSTEP) read_a(std::integral_constant<int, S + 1>{}, std::integral_constant<int, 0>{}, fa[0]);
      if (gstep + 1 < total) read_w(nslot, std::integral_constant<int, 0>{}, fw[0]);
      burst(fw[1], fa[1], std::integral_constant<int, 2>{});
      interleave();
      ++gstep;
    });

    // ---- epilogue (gemm16.hip's idiom): lane (lm, lq) holds pixel lm of image row RT wm + i and, per tile j, channels 16 j + 4 lq + r;
    // v_permlane16_swap of tiles 2 jp, 2 jp + 1 -> 8 consecutive channels 32 jp + 16 (lq & 1) + 8 (lq >> 1) .. + 7: one 16-byte store.
    // (Measured and dropped: exchanging a pair between lanes lm and lm ^ 8 so that a store covers 8 whole 128-byte lines instead of 16 half
    // lines -- no gain, profiles/r5/bench_gnconv_fullline_r5.txt.)
    if (last_ph && (!(ABL & 16) || acc[0][0][0] == 1.2345e-30f)) {
      float st[2][2][2] = {};                              // [pair jp][quad of the lane's 8 channels][sum, sum of squares] over the wave's RT rows
      auto tally = [&](int jp, const u32x4& pk) {          // of the ROUNDED values: what the next GroupNorm will read
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float lo = gc_lo(pk[e]), hi = gc_hi(pk[e]);
          st[jp][e >> 1][0] += lo + hi;
          st[jp][e >> 1][1] = fmaf(lo, lo, fmaf(hi, hi, st[jp][e >> 1][1]));
        }
      };
#pragma unroll
      for (int i = 0; i < RT; ++i) {
#pragma unroll
        for (int jp = 0; jp < 2; ++jp) {
          const acc4 x = acc[i][2 * jp], y = acc[i][2 * jp + 1];
          u32x4 pk;
          if (!(RES && NPH == 1)) {
            const auto s01 = __builtin_amdgcn_permlane16_swap(pack_bf16x2(x[0], x[1]), pack_bf16x2(y[0], y[1]), false, false);
            const auto s23 = __builtin_amdgcn_permlane16_swap(pack_bf16x2(x[2], x[3]), pack_bf16x2(y[2], y[3]), false, false);
            pk = (u32x4){s01[0], s23[0], s01[1], s23[1]};
          } else {
            float o8[8];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const auto sw = __builtin_amdgcn_permlane16_swap(__float_as_uint(x[r]), __float_as_uint(y[r]), false, false);
              o8[r] = __uint_as_float(sw[0]);
              o8[4 + r] = __uint_as_float(sw[1]);
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              o8[2 * e] += gc_lo(rv[i][jp][e]);
              o8[2 * e + 1] += gc_hi(rv[i][jp][e]);
            }
            pk = (u32x4){pack_bf16x2(o8[0], o8[1]), pack_bf16x2(o8[2], o8[3]), pack_bf16x2(o8[4], o8[5]), pack_bf16x2(o8[6], o8[7])};
          }
          // (the row offset rides in the VECTOR offset, not in soffset: with a register in soffset hipcc's hazard recognizer does not keep the
          //  wait states between a 16-byte store and the next write of its data registers -- "this hazard only exists if the instruction is
          //  not using a register in the soffset field" --, and on MI355X the store then sends the overwritten values for lanes 12 .. 15 of
          //  every 16-lane row: the wrong outputs this file produced in about one run of three; tools/check_mfma_overlap.py scans for it)
          __builtin_amdgcn_raw_buffer_store_b128(pk, rO, (int)eoff + i * erow + 64 * jp, 0, 0);
          if constexpr (ST) tally(jp, pk);
        }
      }
      if constexpr (ST) {
        // sum over the 16 pixels of the row tile = the 16 lanes lm of a DPP row, in a fixed order (quad, quad pairs, halves, row); lane lm = 0 of
        // every lq then hands the wave's 4 quads x (sum, sum of squares) to the scratch [wave][lq][8]
        float* sc = reinterpret_cast<float*>(smem + GC_L_STAT) + (wid * 4 + lq) * 8;
#pragma unroll
        for (int jp = 0; jp < 2; ++jp)
#pragma unroll
          for (int k = 0; k < 2; ++k)
#pragma unroll
            for (int w = 0; w < 2; ++w) {
              float v = st[jp][k][w];
              v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xf, 0xf, false));    // quad_perm [1, 0, 3, 2]
              v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xf, 0xf, false));    // quad_perm [2, 3, 0, 1]
              v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xf, 0xf, false));   // row_half_mirror
              v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xf, 0xf, false));   // row_mirror
              if (lm == 0) sc[(jp * 2 + k) * 2 + w] = v;
            }
      }
    }
    if constexpr (ST) {
      if (last_ph) {
        // wave 0 folds the four row groups (wm = 0 .. 3, in that order) and writes the tile's 32 quads x 2 floats
        __builtin_amdgcn_s_waitcnt(0xC07F);
        __builtin_amdgcn_s_barrier();
        if (wid == 0) {
          const int q = lane >> 1, w = lane & 1;               // quad of the launch's 128 channels, sum / sum of squares
          const int wnq = q >> 4, c = 4 * q - 64 * wnq;        // wave column, channel inside its 64
          const int jp = c >> 5, i8 = (c & 31) >> 3, k = (c & 7) >> 2;
          const int lqq = (i8 >> 1) | ((i8 & 1) << 1);         // 8-channel run i8 = 2 (lq & 1) + (lq >> 1)
          const float* sc = reinterpret_cast<const float*>(smem + GC_L_STAT) + (wnq * 4 + lqq) * 8 + (jp * 2 + k) * 2 + w;
          float v = sc[0];
#pragma unroll
          for (int m = 1; m < 4; ++m) v += sc[m * WN * 4 * 8];
          __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), dma_rsrc(a.stats), ((t_this * (a.ldo / 4) + q) * 2 + w) * 4, 0, 0);
        }
      }
    }
    // ---- the next unit's halo takes the place of this one's
    if (has_next || ((ABL & 4) && unit + 1 < my_units)) {
      __builtin_amdgcn_s_barrier();                        // every wave has read its last A fragments
      if constexpr (!(ABL & (4 | 64))) write_halo();
      __builtin_amdgcn_s_waitcnt(0xC07F);
      __builtin_amdgcn_s_barrier();
      read_a(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{}, fa[0]);
    }
    vt = nvt;
    ph = nph;
  }
  wait_vmcnt<0>();                                         // (the poison pieces write LDS: none may be in flight when the workgroup's LDS is released)
}

// Statistics of a GroupNorm from the per-tile partials of the launch that produced its input: stats [nb * tiles][C / 4][2] (sum, sum of squares per
// 4-channel quad and 16 x 16 tile) -> the (scale, shift) tables.  One workgroup per image, thread = (group, run of tiles): every partial becomes
// (count, mean, M2 = sum of squared deviations from ITS mean) and the partials are merged with Chan's update -- mean and M2 of a union from those of
// its parts, every term of the order of the variance -- in a fixed order (8 interleaved runs of tiles per group, then the runs).  The cancellation
// E[x^2] - mean^2 is then confined to the 1024 values of one partial instead of the image's 10^5 .. 10^6 (|mean| >> sigma).
__global__ __launch_bounds__(256) void gn_stats_finalize_kernel(const float* __restrict__ stats, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                                float* __restrict__ scale, float* __restrict__ shift, int tiles, int C, int G, float eps,
                                                                float inv_count) {
  __shared__ float part[8][64][3];
  const int n = blockIdx.x, g = threadIdx.x % G, run = threadIdx.x / G, nrun = 256 / G;     // G = 32: 8 runs
  const int qpg = C / G / 4, nq = C / 4;
  const float cnt = 1024.f, icnt = 1.f / 1024.f;           // values per partial: 16 x 16 pixels x 4 channels
  float cn = 0.f, cm = 0.f, cq = 0.f;
  for (int t = run; t < tiles; t += nrun) {
    const float* p = stats + ((long)(n * tiles + t) * nq + g * qpg) * 2;
    for (int k = 0; k < qpg; ++k) {
      const float s1 = p[2 * k], mb = s1 * icnt, qb = fmaxf(fmaf(-s1, mb, p[2 * k + 1]), 0.f);
      const float tot = cn + cnt, d = mb - cm, f = cnt / tot;
      cm = fmaf(d, f, cm);
      cq += qb + d * d * cn * f;
      cn = tot;
    }
  }
  part[run][g][0] = cn;
  part[run][g][1] = cm;
  part[run][g][2] = cq;
  __syncthreads();
  if (run == 0) {
    for (int r = 1; r < nrun; ++r) {
      const float nb_ = part[r][g][0];
      if (nb_ > 0.f) {
        const float tot = cn + nb_, d = part[r][g][1] - cm, f = nb_ / tot;
        cm = fmaf(d, f, cm);
        cq += part[r][g][2] + d * d * cn * f;
        cn = tot;
      }
    }
    const float mean = cm;
    const float var = cq / cn;
    (void)inv_count;
    const float rstd = rsqrtf(var + eps);
    const int cpg = C / G;
    for (int c = g * cpg; c < (g + 1) * cpg; ++c) {
      const float sc = gamma[c] * rstd;
      scale[(long)n * C + c] = sc;
      shift[(long)n * C + c] = fmaf(-mean, sc, beta[c]);
    }
  }
}

int g_gnconv_abl = 0;

}  // namespace

void mmgt_gnconv_set_abl(int v) { g_gnconv_abl = v; }

// x (nb, H, W, Cin) bf16 channels-last, H and W multiples of 16; scale / shift (nb, Cin) fp32 (mmgt_groupnorm_affine); wimg = pack_gnconv
// image of the (Cout, Cin, 3, 3) weight; bias (Cout) fp32 or null; residual / out: (nb, H, W, ldo) bf16 tensors of which this launch reads /
// writes channels 0 .. Cout - 1 of the pointers it is given (ldo >= Cout, a multiple of 8: a 256-wide output runs as two launches on its
// 128-channel halves); residual may be null.  (Cin, Cout) = (128, 128), (256, 128), (128, 64); the residual with Cout = 128 only.
extern "C" int mmgt_gn_silu_conv3x3(const void* x, const float* scale, const float* shift, const void* wimg, const float* bias, const void* residual,
                                    void* out, float* stats, int nb, int H, int W, int cin, int cout, int ldo, int dtype, void* stream) {
  MMGT_CHECK(x && scale && shift && wimg && out && nb > 0 && H > 0 && W > 0, "gn_silu_conv3x3: bad arguments");
  MMGT_CHECK(dtype == MMGT_BF16, "gn_silu_conv3x3: bf16 only");
  const bool s128 = cin == 128 && cout == 128, s256 = cin == 256 && cout == 128, s64 = cin == 128 && cout == 64;
  MMGT_CHECK(s128 || s256 || (s64 && !residual), "gn_silu_conv3x3: (Cin, Cout) = (128, 128), (256, 128), or (128, 64) without residual (got %d -> %d)", cin, cout);
  MMGT_CHECK(shift == scale + (long)nb * cin, "gn_silu_conv3x3: scale and shift must be one allocation, shift = scale + nb * Cin (mmgt_amd/hip.py::groupnorm_affine)");
  MMGT_CHECK(!stats || (cout == 128 && ((uintptr_t)stats % 8) == 0 && (long)nb * (H / GC_T) * (W / GC_T) * (ldo / 4) * 8 < (1l << 31)),
             "gn_silu_conv3x3: statistics come with 128 output channels per launch (and a buffer below 2 GiB)");
  MMGT_CHECK(ldo >= cout && ldo % 8 == 0, "gn_silu_conv3x3: ldo = %d must be a multiple of 8 and >= Cout", ldo);
  MMGT_CHECK(H % GC_T == 0 && W % GC_T == 0, "gn_silu_conv3x3: H and W must be multiples of 16 (got %d x %d)", H, W);
  MMGT_CHECK((long)nb * H * W * cin * 2 < (1l << 31) && (long)nb * H * W * ldo * 2 < (1l << 31), "gn_silu_conv3x3: x, residual and out must be smaller than 2 GiB each");
  GcArgs a;
  a.x = reinterpret_cast<const bf16_t*>(x);
  a.scale = scale;
  a.shift = shift;
  a.wimg = reinterpret_cast<const char*>(wimg);
  a.bias = bias;
  a.res = reinterpret_cast<const bf16_t*>(residual);
  a.out = reinterpret_cast<bf16_t*>(out);
  a.nb = nb;
  a.H = H;
  a.W = W;
  a.tiles_x = W / GC_T;
  a.tiles_per_img = (H / GC_T) * (W / GC_T);
  a.ntiles = nb * a.tiles_per_img;
  a.ldo = ldo;
  a.stats = stats;
  int dev = 0;
  static int ncu[16] = {};
  MMGT_CHECK(hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev < 16, "gn_silu_conv3x3: device query failed");
  if (!ncu[dev]) {
    hipDeviceProp_t prop;
    MMGT_CHECK(hipGetDeviceProperties(&prop, dev) == hipSuccess, "gn_silu_conv3x3: device query failed");
    ncu[dev] = prop.multiProcessorCount;
  }
  int gx = ncu[dev] / 8 * 8;
  if (gx > a.ntiles) gx = a.ntiles;
  hipStream_t s = (hipStream_t)stream;
  static bool ready[16][32] = {};                          // LDS attribute set, per device and kernel instantiation (slot = the call site below)
  auto go = [&](void (*kern)(const GcArgs), int slot) -> int {
    if (!ready[dev][slot]) {
      if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, GC_LDS) != hipSuccess) {
        mmgt_set_error("gn_silu_conv3x3: cannot reserve %d bytes of LDS", GC_LDS);
        return 2;
      }
      ready[dev][slot] = true;
    }
    hipLaunchKernelGGL(kern, dim3(gx), dim3(512), GC_LDS, s, a);
    return 0;
  };
  int rc;
  if (stats && s256) rc = residual ? go(gnconv_kernel<2, 128, true, 0, true>, 15) : go(gnconv_kernel<2, 128, false, 0, true>, 16);
  else if (stats) rc = residual ? go(gnconv_kernel<1, 128, true, 0, true>, 17) : go(gnconv_kernel<1, 128, false, 0, true>, 18);
  else if (s256) rc = residual ? go(gnconv_kernel<2, 128, true, 0>, 0) : go(gnconv_kernel<2, 128, false, 0>, 1);
  else if (s64) rc = go(gnconv_kernel<1, 64, false, 0>, 2);
  else if (residual) rc = go(gnconv_kernel<1, 128, true, 0>, 3);
  else {
    rc = -1;
#ifdef MMGT_ABLATE   // timing ablations (results are garbage): only in libmmgt_hip_abl.so (`make abl`), never in the product library
    switch (g_gnconv_abl) {
      case 1: rc = go(gnconv_kernel<1, 128, false, 1>, 4); break;
      case 2: rc = go(gnconv_kernel<1, 128, false, 2>, 5); break;
      case 4: rc = go(gnconv_kernel<1, 128, false, 4>, 6); break;
      case 8: rc = go(gnconv_kernel<1, 128, false, 8>, 7); break;
      case 16: rc = go(gnconv_kernel<1, 128, false, 16>, 8); break;
      case 32: rc = go(gnconv_kernel<1, 128, false, 32>, 9); break;
      case 6: rc = go(gnconv_kernel<1, 128, false, 6>, 10); break;
      case 22: rc = go(gnconv_kernel<1, 128, false, 22>, 11); break;
      case 64: rc = go(gnconv_kernel<1, 128, false, 64>, 12); break;
      case 128: rc = go(gnconv_kernel<1, 128, false, 128>, 13); break;
      default: break;
    }
#endif
    if (rc < 0) rc = go(gnconv_kernel<1, 128, false, 0>, 14);
  }
  if (rc) return rc;
  MMGT_LAUNCH_CHECK();
  return 0;
}

// stats [nb * tiles][C / 4][2] as written by mmgt_gn_silu_conv3x3 (tiles = (H / 16) (W / 16) per image) -> scale | shift (nb, C) of GroupNorm(G groups,
// gamma, beta, eps) over the stored tensor; C / G a multiple of 4, G <= 64 and a divisor of 256.
extern "C" int mmgt_gn_stats_finalize(const float* stats, const float* gamma, const float* beta, float* scale, float* shift, int nb, int tiles, int C,
                                      int G, float eps, void* stream) {
  MMGT_CHECK(stats && gamma && beta && scale && shift && nb > 0 && tiles > 0, "gn_stats_finalize: bad arguments");
  MMGT_CHECK(G > 0 && G <= 64 && 256 % G == 0 && C % G == 0 && (C / G) % 4 == 0, "gn_stats_finalize: unsupported C = %d, G = %d", C, G);
  hipLaunchKernelGGL(gn_stats_finalize_kernel, dim3(nb), dim3(256), 0, (hipStream_t)stream, stats, gamma, beta, scale, shift, tiles, C, G, eps,
                     1.f / ((float)tiles * 256.f * (float)(C / G)));
  MMGT_LAUNCH_CHECK();
  return 0;
}

extern "C" long mmgt_gn_silu_conv3x3_image_bytes(int cin, int cout) {
  return ((cin == 128 || cin == 256) && cout == 128) || (cin == 128 && cout == 64) ? (long)(cin / 64) * 9 * 64 * cout * 2 : -1;
}
