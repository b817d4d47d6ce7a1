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
constexpr int GC_L_RING = (GC_HALO + 1023) / 1024 * 1024, GC_L_BIAS = GC_L_RING + GC_NSLOT * 16 * 1024, GC_L_TAB = GC_L_BIAS + 128 * 4, GC_LDS = GC_L_TAB + 2 * GC_PC * 4;
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
};

__device__ __forceinline__ acc4 gc_mma(s16x8 a, s16x8 b, acc4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }
__device__ __forceinline__ float gc_lo(unsigned u) { return __uint_as_float(u << 16); }
__device__ __forceinline__ float gc_hi(unsigned u) { return __uint_as_float(u & 0xffff0000u); }

template <int LO, int... I, typename F>
__device__ __forceinline__ void gc_for_impl(std::integer_sequence<int, I...>, F&& fn) { (fn(std::integral_constant<int, LO + I>{}), ...); }
template <int LO, int HI, typename F>
__device__ __forceinline__ void gc_for(F&& fn) { gc_for_impl<LO>(std::make_integer_sequence<int, (HI > LO ? HI - LO : 0)>{}, static_cast<F&&>(fn)); }

// NPH = Cin / 128 (1, 2); COUT = 128 (8 waves = 4 x 2: 4 image rows x 64 channels each) or 64 (8 x 1: 2 image rows x 64 channels).
// ABL (mmgt_tune("gnconv_abl", bit), timing only -- results are garbage): 1 no MFMAs, 2 no weight DMA after the prologue, 4 no halo loads /
// normalisation / LDS writes after the first tile, 8 no hand-over wait, 16 no epilogue (residual loads, stores), 32 no hand-over barrier,
// 64 halo loads but no normalisation / LDS writes, 128 normalisation / LDS writes but no halo loads
template <int NPH, int COUT, bool RES, int ABL>
__global__ __launch_bounds__(512, 2) void gnconv_kernel(const GcArgs a) {
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
  auto decode = [&](int v, int& n, int& ty, int& tx) {
    const int q = a.ntiles >> 3, r = a.ntiles & 7, xc = v & 7;
    const int t = (xc < r ? xc * (q + 1) : r * (q + 1) + (xc - r) * q) + (v >> 3);
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
  // dwords J0 .. J0 + NJ - 1 of hv[I] <- bf16( silu( . * scale + shift ) ), zeros outside the image.  Branch-free (a select on the mask made hipcc
  // branch around the transcendentals, one basic block per pair, nothing interleaved with the MFMAs)
  auto norm_part = [&](auto Ic, auto Jc, auto Nc) {
    constexpr int I = decltype(Ic)::value, J0 = decltype(Jc)::value, NJ = decltype(Nc)::value;
    static_assert(J0 % 2 == 0 && NJ % 2 == 0, "pairs of dwords");
    const unsigned m = 0u - ((hmask >> I) & 1u);
    int tofs = GC_L_TAB + 32 * oct;
    asm volatile("" : "+v"(tofs));                          // opaque: the table is re-read per call (4 ds_read_b128 per vector) instead of living in 16 registers
#pragma unroll
    for (int jj = J0; jj < J0 + NJ; jj += 2) {
      const f32x4 sc = *reinterpret_cast<const f32x4*>(smem + tofs + 8 * jj), sh = *reinterpret_cast<const f32x4*>(smem + tofs + GC_PC * 4 + 8 * jj);
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int j = jj + u;
        // silu(t) = t / (1 + e^-t) with v_exp_f32 / v_rcp_f32 (1 ulp) instead of the IEEE division sequence: 10 issue slots of 4 cycles per
        // element beside MFMAs that hold the SIMD's vector issue for 8 of their 16 cycles.  Scalar f32 arithmetic on purpose (the file is built
        // with -fno-slp-vectorize): packed f32 instructions cost ~25 cycles each beside MFMAs (MI355X_MICROARCH.md, per-instruction constants).
        const float t0 = fmaf(gc_lo(hv[I][j]), sc[2 * u], sh[2 * u]), t1 = fmaf(gc_hi(hv[I][j]), sc[2 * u + 1], sh[2 * u + 1]);
        const float v0 = t0 * __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(t0 * -1.4426950408889634f));
        const float v1 = t1 * __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(t1 * -1.4426950408889634f));
        unsigned pk = pack_bf16x2(v0, v1) & m;
        asm volatile("" : "+v"(pk));                        // pinned here: hipcc otherwise sinks the arithmetic to the tile's end, where hv is stored
        hv[I][j] = pk;
      }
    }
  };
  auto norm_vec = [&](auto Ic) { norm_part(Ic, std::integral_constant<int, 0>{}, std::integral_constant<int, 4>{}); };
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
      if constexpr (NORM) norm_part(std::integral_constant<int, S - 6>{}, std::integral_constant<int, 0>{}, std::integral_constant<int, 2>{});
#pragma unroll
      for (int i = 0; i < RT; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) if constexpr (!(ABL & 1)) acc[i][j] = gc_mma(fw[0][j], fa[0][i], acc[i][j]); else acc[i][j][0] += __builtin_bit_cast(float, (int)fw[0][j][0] ^ (int)fa[0][i][0]);
      if constexpr (NORM) {
#pragma unroll
        for (int k = 0; k < 4 * RT; ++k) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x002, 12 / RT, 0);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
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
      if constexpr (NORM) norm_part(std::integral_constant<int, S - 6>{}, std::integral_constant<int, 2>{}, std::integral_constant<int, 2>{});
#pragma unroll
      for (int i = 0; i < RT; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) if constexpr (!(ABL & 1)) acc[i][j] = gc_mma(fw[1][j], fa[1][i], acc[i][j]); else acc[i][j][0] += __builtin_bit_cast(float, (int)fw[1][j][0] ^ (int)fa[1][i][0]);
      if constexpr (NORM) {
#pragma unroll
        for (int k = 0; k < 4 * RT; ++k) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x002, 12 / RT, 0);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
      ++gstep;
    });

    // ---- epilogue (gemm16.hip's idiom): lane (lm, lq) holds pixel lm of image row RT wm + i and, per tile j, channels 16 j + 4 lq + r;
    // v_permlane16_swap of tiles 2 jp, 2 jp + 1 -> 8 consecutive channels 32 jp + 16 (lq & 1) + 8 (lq >> 1) .. + 7: one 16-byte store.
    // (Measured and dropped: exchanging a pair between lanes lm and lm ^ 8 so that a store covers 8 whole 128-byte lines instead of 16 half
    // lines -- no gain, profiles/r5/bench_gnconv_fullline_r5.txt.)
    if (last_ph && (!(ABL & 16) || acc[0][0][0] == 1.2345e-30f)) {
#pragma unroll
      for (int i = 0; i < RT; ++i) {
#pragma unroll
        for (int jp = 0; jp < 2; ++jp) {
          const acc4 x = acc[i][2 * jp], y = acc[i][2 * jp + 1];
          if (!(RES && NPH == 1)) {
            const auto s01 = __builtin_amdgcn_permlane16_swap(pack_bf16x2(x[0], x[1]), pack_bf16x2(y[0], y[1]), false, false);
            const auto s23 = __builtin_amdgcn_permlane16_swap(pack_bf16x2(x[2], x[3]), pack_bf16x2(y[2], y[3]), false, false);
            __builtin_amdgcn_raw_buffer_store_b128((u32x4){s01[0], s23[0], s01[1], s23[1]}, rO, (int)eoff + 64 * jp, i * erow, 0);
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
            __builtin_amdgcn_raw_buffer_store_b128((u32x4){pack_bf16x2(o8[0], o8[1]), pack_bf16x2(o8[2], o8[3]), pack_bf16x2(o8[4], o8[5]), pack_bf16x2(o8[6], o8[7])},
                                                   rO, (int)eoff + 64 * jp, i * erow, 0);
          }
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

int g_gnconv_abl = 0;

}  // namespace

void mmgt_gnconv_set_abl(int v) { g_gnconv_abl = v; }

// x (nb, H, W, Cin) bf16 channels-last, H and W multiples of 16; scale / shift (nb, Cin) fp32 (mmgt_groupnorm_affine); wimg = pack_gnconv
// image of the (Cout, Cin, 3, 3) weight; bias (Cout) fp32 or null; residual / out: (nb, H, W, ldo) bf16 tensors of which this launch reads /
// writes channels 0 .. Cout - 1 of the pointers it is given (ldo >= Cout, a multiple of 8: a 256-wide output runs as two launches on its
// 128-channel halves); residual may be null.  (Cin, Cout) = (128, 128), (256, 128), (128, 64); the residual with Cout = 128 only.
extern "C" int mmgt_gn_silu_conv3x3(const void* x, const float* scale, const float* shift, const void* wimg, const float* bias, const void* residual,
                                    void* out, int nb, int H, int W, int cin, int cout, int ldo, int dtype, void* stream) {
  MMGT_CHECK(x && scale && shift && wimg && out && nb > 0 && H > 0 && W > 0, "gn_silu_conv3x3: bad arguments");
  MMGT_CHECK(dtype == MMGT_BF16, "gn_silu_conv3x3: bf16 only");
  const bool s128 = cin == 128 && cout == 128, s256 = cin == 256 && cout == 128, s64 = cin == 128 && cout == 64;
  MMGT_CHECK(s128 || s256 || (s64 && !residual), "gn_silu_conv3x3: (Cin, Cout) = (128, 128), (256, 128), or (128, 64) without residual (got %d -> %d)", cin, cout);
  MMGT_CHECK(shift == scale + (long)nb * cin, "gn_silu_conv3x3: scale and shift must be one allocation, shift = scale + nb * Cin (mmgt_amd/hip.py::groupnorm_affine)");
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
  static int ncu = 0;
  if (!ncu) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) {
      mmgt_set_error("gn_silu_conv3x3: device query failed");
      return 2;
    }
    ncu = prop.multiProcessorCount;
  }
  int gx = ncu / 8 * 8;
  if (gx > a.ntiles) gx = a.ntiles;
  hipStream_t s = (hipStream_t)stream;
  static bool ready[32] = {};                              // LDS attribute set, per kernel instantiation (slot = the call site below)
  auto go = [&](void (*kern)(const GcArgs), int slot) -> int {
    if (!ready[slot]) {
      if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, GC_LDS) != hipSuccess) {
        mmgt_set_error("gn_silu_conv3x3: cannot reserve %d bytes of LDS", GC_LDS);
        return 2;
      }
      ready[slot] = true;
    }
    hipLaunchKernelGGL(kern, dim3(gx), dim3(512), GC_LDS, s, a);
    return 0;
  };
  int rc;
  if (s256) rc = residual ? go(gnconv_kernel<2, 128, true, 0>, 0) : go(gnconv_kernel<2, 128, false, 0>, 1);
  else if (s64) rc = go(gnconv_kernel<1, 64, false, 0>, 2);
  else if (residual) rc = go(gnconv_kernel<1, 128, true, 0>, 3);
  else switch (g_gnconv_abl) {
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
    default: rc = go(gnconv_kernel<1, 128, false, 0>, 14); break;
  }
  if (rc) return rc;
  MMGT_LAUNCH_CHECK();
  return 0;
}

extern "C" long mmgt_gn_silu_conv3x3_image_bytes(int cin, int cout) {
  return ((cin == 128 || cin == 256) && cout == 128) || (cin == 128 && cout == 64) ? (long)(cin / 64) * 9 * 64 * cout * 2 : -1;
}
