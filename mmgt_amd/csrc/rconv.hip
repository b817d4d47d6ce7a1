// GroupNorm-apply + SiLU + conv3x3 of the UNet's resnets in ONE launch (bf16, stride 1, padding 1, gfx950):
//
//   out[n, y, x, :] = bias + temb[n] + sum_{ky, kx} W[ky, kx] . silu( in[n, y + ky - 1, x + kx - 1, :] * scale[n, :] + shift[n, :] )   (+ residual)
//
// in = the channel concatenation of up to two channels-last tensors (hidden | skip: unet_3d_blocks.py:941-969 concatenates first), (scale, shift)
// the per-(image, channel) tables of the GroupNorm in front of the conv (mmgt_groupnorm_affine2: the statistics pass alone), temb the
// time-embedding row of the image's batch entry.  Replaces the `hip.groupnorm(silu=True)` -> `hip.conv3x3` pairs of
// mmgt_amd/unet3d.py::_resnet (reference: ResnetBlock3D.forward, /root/reference/src/models/resnet.py:217-247: norm1 -> nonlinearity -> conv1
// -> + temb -> norm2 -> nonlinearity -> conv2 -> + shortcut; InflatedGroupNorm / InflatedConv3d of :20-28, :156-196 are per-frame operators, so a
// frame is an image here).  The unfused pair wrote the normalised tensor to HBM and read it back nine times through gemm16's im2col gather.
//
// Structure (csrc/gnconv.hip's, re-cut for 320 .. 2560 input channels and 320-wide output blocks).  A workgroup owns a UNIT = a 16 x 16 pixel
// tile of one image x a block of CB = 32 NT output channels, and walks the input channels in PHASES of 64:
//   * the phase's 18 x 18 x 64-channel halo goes global -> LDS RAW by LDS-DMA (41 pieces of 1 KiB, no registers), one phase ahead, into the
//     second of two halo buffers, and is normalised + SiLU'd + rounded to bf16 IN PLACE by the thread that issued its pieces (so its own
//     vmcnt covers the hand-over; the table of the phase's 64 scales / shifts rides with it), a dword at a time between the MFMAs of the
//     running phase; out-of-image pixels are zeros AFTER the activation, as the conv's padding wants;
//   * LDS image of a halo: pixel R = 18 hy + hx is a 128-byte row, 16-byte chunk c of it at slot c ^ (R & 7) -- the DMA writes 64 lanes x 16 B
//     contiguously, so the swizzle sits in the per-lane SOURCE offset (chunk (lane & 7) ^ (lane >> 3) of pixel 8 piece + (lane >> 3): a lane
//     constant), and the sixteen lanes of every ds_read_b128 group of an A fragment (16 consecutive pixels, any start) hit sixteen bank groups;
//   * the nine taps' A fragments (v_mfma_f32_16x16x32_bf16: lane (lm, lq) = pixel lm of an image row, channels 32 ks + 8 lq .. + 7) are read
//     straight from the halo -- no im2col staging --, the weights (fragment-major image, mmgt_amd/packing.py::pack_rconv: per output block,
//     phase, tap and k-step of 32 channels 2 NT pieces of 1 KiB) stream through an NSLOT-deep LDS ring by LDS-DMA, ONE barrier per k-step
//     (4 NT MFMAs per wave) placed behind the MFMAs that still read the slot being freed;
//   * 8 waves = 4 (rows) x 2 (columns): a wave owns 4 image rows x 16 pixels x CB / 2 output channels = 4 x NT accumulator tiles (160 registers
//     at NT = 10), W fragments in a rolling window of four, the next k-step's A fragments read under the last MFMAs of the current one;
//   * the accumulators start from bias + temb row (LDS copy fetched one unit ahead); epilogue as gemm16's (v_permlane16_swap pairs -> 16-byte
//     stores, residual added in fp32).
// Staged bytes per MFMA: the weights only (64 CB bytes per 256 x CB x 32 MACs) -- 0.56 of gemm16's 256 x 320 conv tile.
#include <type_traits>
#include <utility>

#include "common.h"
#include "gemm_common.h"
#include "mmgt_hip.h"

namespace {

typedef __attribute__((ext_vector_type(4))) float acc4;

constexpr int RC_T = 16, RC_HP = RC_T + 2, RC_NPIX = RC_HP * RC_HP;          // tile edge, halo edge, halo pixels (324)
constexpr int RC_PC = 64;                                                   // channels per phase
constexpr int RC_HPIECES = (RC_NPIX * (RC_PC / 8) + 63) / 64;               // 1-KiB DMA pieces of a halo: 41 (the last one half used)
constexpr int RC_HALO = RC_HPIECES * 1024;                                  // bytes per halo buffer
constexpr int RC_NHV = (RC_HPIECES + 7) / 8;                                // pieces (= 16-byte vectors per lane) per wave and phase: 6
constexpr int RC_NSTEP = 18;                                                // k-steps (tap, 32-channel half) per phase
constexpr int RC_FWD = 3;                                                   // W fragments in flight per wave (4: 4 registers more -- the 320-wide cut then spills inside its k-steps)

struct RcArgs {
  const bf16_t* x0; const bf16_t* x1;   // (nb, H, W, C0) [, (nb, H, W, C1)]
  int C0, C1;
  const float* scale;                   // (2, nb, C0 + C1): scale | shift
  const char* wimg;                     // pack_rconv image
  const float* bias;                    // (Cout) or null
  const float* bias2;                   // (rows, Cout) or null: row n / b2_imgs is added to image n
  int b2_imgs;
  const bf16_t* res;                    // (nb, H, W, Cout) or null
  bf16_t* out;                          // (nb, H, W, Cout)
  int nb, H, W, tiles_x, tiles_per_img, ncb, nunits, nph, cout;
  float* stats;                         // or null: per (tile, wave row group, channel) the triple (pivot, sum, sum of squares of deviations) of the STORED values: [3][partials][cout]
  unsigned long long* trace;            // debug: [workgroup][512] 100-MHz stamps of wave 0 (k-step starts; 2 per epilogue), or null
  int trace_fine;                       // debug: four stamps per k-step (start, in front of the counted wait, in front of / behind the barrier)
  int stagger;                          // start delay of workgroup group (blockIdx.x >> 3) & 7, in units of 64 cycles per group index
  int abl;                              // -DMMGT_ABLATE builds only (timing ablations, results are garbage): 1 no weight data, 2 no halo data, 4 no normalisation, 8 no barriers, 16 no stores
};

// Accumulators live in the accumulation registers, tied in place (through the builtin hipcc rotates the 160 registers of a wave's tiles through
// other ranges with copies -- and spills: DESIGN 4, round 4, gemm16v).
// With two waves per SIMD hipcc splits the wave's 256 registers 128 | 128, so the first eight column tiles of a row (4 x 8 x 4 = 128 registers) sit in
// AGPRs and the rest in VGPRs, as in gemm16v.
template <bool AGPR>
__device__ __forceinline__ void rc_mma(acc4& c, s16x8 a, s16x8 b) {
  if constexpr (AGPR) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b));
  else asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b));
}
// An accumulator tile -> four VGPR floats, at THIS point of the program (left to hipcc, all 128 AGPR reads of an epilogue are hoisted to its top and
// the VGPR-resident tiles are spilled to make room -- straight behind the asm MFMAs whose result latency it does not know).
template <bool AGPR>
__device__ __forceinline__ acc4 rc_get(const acc4& c) {
  if constexpr (AGPR) {
    acc4 r;
    asm volatile("v_accvgpr_read_b32 %0, %4\n\tv_accvgpr_read_b32 %1, %5\n\tv_accvgpr_read_b32 %2, %6\n\tv_accvgpr_read_b32 %3, %7"
                 : "=&v"(r[0]), "=&v"(r[1]), "=&v"(r[2]), "=&v"(r[3]) : "a"(c[0]), "a"(c[1]), "a"(c[2]), "a"(c[3]));
    return r;
  } else {
    return c;
  }
}
__device__ __forceinline__ float rc_lo(unsigned u) { return __uint_as_float(u << 16); }
__device__ __forceinline__ float rc_hi(unsigned u) { return __uint_as_float(u & 0xffff0000u); }

// a wave-uniform 64-bit base in SGPRs (a SELECT between two descriptors would put every DMA into a waterfall loop).  A free function: a captureless
// helper lambda called from another lambda inside a __global__ function makes hipcc's host pass drop the kernel's stub.
__device__ __forceinline__ const void* rc_uni(const void* ptr) {
  const unsigned long long p = reinterpret_cast<unsigned long long>(ptr);
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)p), hi = __builtin_amdgcn_readfirstlane((unsigned)(p >> 32));
  return reinterpret_cast<const void*>(((unsigned long long)hi << 32) | lo);
}

// the lane id, read afresh (a free function: see rc_uni)
// (volatile: the builtin is loop-invariant, hipcc hoists it out of the phase loop and keeps what is derived from it in registers -- which it then spills)
__device__ __forceinline__ int rc_lane() {
  int l;
  asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
  return l;
}

template <int LO, int... I, typename F>
__device__ __forceinline__ void rc_for_impl(std::integer_sequence<int, I...>, F&& fn) { (fn(std::integral_constant<int, LO + I>{}), ...); }
template <int LO, int HI, typename F>
__device__ __forceinline__ void rc_for(F&& fn) { rc_for_impl<LO>(std::make_integer_sequence<int, (HI > LO ? HI - LO : 0)>{}, static_cast<F&&>(fn)); }

// Three cuts of a workgroup's 8 waves (rows x columns) over its unit of 256 pixels x CB output channels:
//   NT = 10, RT = 4 (4 x 2 waves, CB = 320): the 320-wide level (768 units = 3 rounds of 256 CUs at 48 x 64 x 64);
//   NT =  8, RT = 4 (4 x 2 waves, CB = 256): the 1280-wide level (48 tiles x 5 blocks = 240 units instead of 192);
//   NT = 10, RT = 2 (8 x 1 waves, CB = 160): the 640-wide level (192 tiles x 4 blocks = 768 units instead of 384 = 1.5 rounds) and 24-image launches.
template <int NT, int RT> struct RcCfg {
  static constexpr int WN = RT / 2, CB = 16 * NT * WN, SLOT = CB * 64, NPC = NT * WN;   // wave columns; output block; bytes and 1-KiB pieces per k-step
  // piece q of a k-step is issued by wave q % 8: every wave has at least PPW = NPC / 8 weight pieces per k-step in its in-order vmcnt queue, and the
  // counted waits below, written for PPW, only wait longer (never shorter) on the waves that have one more
  static constexpr int PPW = NPC / 8;
  static constexpr int NSLOT_FIT = (160 * 1024 - 2 * RC_HALO - 6144) / SLOT, NSLOT = NSLOT_FIT > 5 ? 5 : NSLOT_FIT;   // (<= 5: the late waves' normalisation must end in front of k-step 17)
  static constexpr int L_RING = 2 * RC_HALO, L_BIAS = L_RING + NSLOT * SLOT, L_TAB = L_BIAS + 4096, L_DUMMY = L_TAB + 1024, LDS = L_DUMMY + 1024;
  static_assert(LDS <= 160 * 1024 && NSLOT >= 3 && PPW >= 1 && NT % 2 == 0 && NT >= RC_FWD + 2 && (RC_NSTEP * NT) % RC_FWD == 0 && (RT == 2 || RT == 4), "LDS / shape");
};

// ST: the epilogue also emits the statistics of the GroupNorm that reads this launch's output (the next leg's norm2, resnet.py:231): per partial
// = (tile, wave row group: RT image rows x 16 pixels) and channel the pivot-shifted sums of the bf16 values it stores, folded over the partial's
// pixels in a fixed order; mmgt_gn_stats_finalize_unet combines the partials (Chan's update: no cancellation whatever the mean) into the
// (scale, shift) tables -- the statistics pass over the tensor is not needed.
template <int NT, int RT, bool RES, bool ST = false>
__global__ __launch_bounds__(512, 2) void rconv_kernel(const RcArgs a) {
  using Cfg = RcCfg<NT, RT>;
  constexpr int CB = Cfg::CB, SLOT = Cfg::SLOT, NPC = Cfg::NPC, PPW = Cfg::PPW, NSLOT = Cfg::NSLOT, WN = Cfg::WN;
  constexpr int NPAIR = NT / 2, NST = RT * NPAIR + (ST ? 6 * NPAIR : 0);      // tile pairs; 16-byte stores per wave and unit (ST: + 6 per pair)
  extern __shared__ __attribute__((aligned(1024))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lm = lane & 15, lq = lane >> 4;
  const int wm = wid / WN, wn = wid % WN;                  // image rows RT wm .. of the tile, output channels 16 NT wn .. of the block
#ifdef MMGT_ABLATE
  const int abl = a.abl;
#else
  constexpr int abl = 0;
#endif
  // debug (tools/trace_rconv.py): 100-MHz stamps of wave 0 -- in the -DMMGT_ABLATE library only: the four null-pointer tests per k-step, each a taken
  // scalar branch, were a tenth of the k-step's instruction stream
#ifdef MMGT_ABLATE
  int trace_n = 0;
  auto stamp = [&]() {
    if (a.trace) {
      if (tid == 0 && trace_n < 512) a.trace[(long)blockIdx.x * 512 + trace_n] = wall_clock64();
      ++trace_n;
    }
  };
  const bool trace_fine = a.trace_fine != 0;
#else
  auto stamp = [&]() {};
  constexpr bool trace_fine = false;
#endif
  const int G = gridDim.x;
  const int cin = a.C0 + a.C1;
  const int my_units = (a.nunits - (int)blockIdx.x + G - 1) / G;
  const int ksteps_per_unit = a.nph * RC_NSTEP;
  const int total = my_units * ksteps_per_unit;            // k-steps this workgroup consumes

  // XCD-aware unit order (gemm.hip): XCD x = v & 7 walks a contiguous run of the sequence s = tile * ncb + block, so the blocks of a tile and
  // horizontally adjacent tiles -- which share halo columns -- are worked on one L2 at about the same time
  auto decode = [&](int v, int& n, int& ty, int& tx, int& cb) {
    const int q = a.nunits >> 3, r = a.nunits & 7, xc = v & 7;
    const int s = (xc < r ? xc * (q + 1) : r * (q + 1) + (xc - r) * q) + (v >> 3);
    const int t = s / a.ncb;
    cb = s - t * a.ncb;
    n = t / a.tiles_per_img;
    const int rem = t - n * a.tiles_per_img;
    ty = rem / a.tiles_x;
    tx = rem - ty * a.tiles_x;
  };

  // ---- weight stream: k-step wg (counted over the workgroup's units) lives in ring slot wg % NSLOT.  Scalar state only, advanced without
  // branches: the image offset of the k-step (w_soff: + one k-step's row of the image; at a unit's end the next unit's block offset, decoded once per
  // unit at its first phase), the LDS offset of the slot, the k-steps left.  Piece q of a k-step is issued by wave q % 8.
  const __amdgpu_buffer_rsrc_t rW = dma_rsrc(a.wimg);
  const int w_row = a.cout * 64;                           // bytes of a k-step's weights over ALL output channels (the image is [k-step][cout / 16][1 KiB])
  int w_left = total, w_ch = 0, w_soff = 0, w_next = 0, w_lds = Cfg::L_RING;
  {
    int n, ty, tx, cb;
    decode(blockIdx.x < (unsigned)a.nunits ? (int)blockIdx.x : 0, n, ty, tx, cb);
    w_soff = cb * SLOT;
  }
  // Beyond the last k-step the pieces still go out against the poison offset (zeros into a slot nobody reads): every wait count below is a
  // compile-time constant on every path.
  auto issue_w = [&]() {
    const unsigned vo = (w_left > 0 && !(abl & 1)) ? (unsigned)(rc_lane() * 16 + wid * 1024) : DMA_POISON;
#pragma unroll
    for (int u = 0; u < (NPC + 7) / 8; ++u) {
      if (8 * u + 7 < NPC || wid + 8 * u < NPC) blds16(rW, vo, w_soff + u * 8192, smem + w_lds + (wid + 8 * u) * 1024);
    }
    --w_left;
    w_lds = w_lds == Cfg::L_RING + (NSLOT - 1) * SLOT ? Cfg::L_RING : w_lds + SLOT;
    ++w_ch;
    const bool wrap = w_ch == ksteps_per_unit;
    w_soff = wrap ? w_next : w_soff + w_row;
    w_ch = wrap ? 0 : w_ch;
  };

  // ---- halo stream: piece q = wid + 8 i of a phase holds vectors 64 q + lane = pixel R = 8 q + (lane >> 3), slot lane & 7, i.e. the source chunk
  // (lane & 7) ^ (R & 7) = (lane & 7) ^ (lane >> 3).  The thread that issues a vector normalises it (same buffer position), so `hmask` (bit i:
  // vector i lies inside the image) is the only state that crosses from the issue to the normalisation.
  unsigned hmask = 0;
  // (everything per lane below is derived from a lane id read AFRESH where it is used: values kept live across the k-steps were spilled, and a
  //  scratch reload inside the loop costs an s_waitcnt vmcnt(0) -- i.e. the latency of the weight and halo pieces just issued: 1 us per reload)
  // One vector (= one piece per wave) of the halo of phase (h_n, h_ty, h_tx; source h_src, h_cs channels per pixel, channel offset h_choff) per
  // hand-over of k-steps 0 .. 5: ~15 full-rate VALU (24-bit multiplies) and one DMA each, instead of ~150 instructions in one k-step of every
  // wave at once (stamps, tools/trace_rconv.py: k-step 0 of a phase took 3.2 us against 1.05).
  int h_n = 0, h_base = 0, h_ty = 0, h_tx = 0, h_cs2 = 0, h_choff2 = 0, h_c0 = 0;
  const bf16_t* h_src = a.x0;
  auto halo_target = [&](int v, int ph) {                    // scalars of the phase whose halo is fetched next
    int cb;
    decode(v, h_n, h_ty, h_tx, cb);
    const int c0 = ph * RC_PC;
    const bool second = c0 >= a.C0;
    h_c0 = c0;
    h_cs2 = (second ? a.C1 : a.C0) * 2;
    h_choff2 = (second ? c0 - a.C0 : c0) * 2;
    h_src = second ? a.x1 : a.x0;
    h_base = (h_n * a.H + h_ty * RC_T - 1) * a.W + h_tx * RC_T - 1;          // pixel index of halo pixel (0, 0) (may be negative: masked below)
  };
  auto issue_halo_vec = [&](auto Ic, int buf) {
    constexpr int i = decltype(Ic)::value;
    const __amdgpu_buffer_rsrc_t rX = dma_rsrc(rc_uni(h_src));
    const int ln = rc_lane(), hl_chunk = (ln & 7) ^ (ln >> 3);
    const int R = 8 * wid + (ln >> 3) + 64 * i;             // pixel of the lane's vector i
    const int hy = (int)__umul24(R, 3641) >> 16, hx = R - RC_HP * hy;           // R / 18, exact for R < 3 000
    const int y = h_ty * RC_T - 1 + hy, x = h_tx * RC_T - 1 + hx;
    const bool ok = R < RC_NPIX && y >= 0 && y < a.H && x >= 0 && x < a.W;
    const unsigned pix = (unsigned)(h_base + (int)__umul24(hy, a.W) + hx);
    const unsigned off = __umul24(pix, (unsigned)h_cs2) + (unsigned)(h_choff2 + hl_chunk * 16);
    const bool piece = wid + 8 * i < RC_HPIECES;                                // (wave-uniform: only wave 0 has a sixth piece)
    blds16(rX, (ok && !(abl & 2)) ? off : DMA_POISON, 0, piece ? smem + buf * RC_HALO + (wid + 8 * i) * 1024 : smem + Cfg::L_DUMMY);
    if constexpr (i == 0) hmask = 0;
    hmask |= ok ? 1u << i : 0u;
    // with the first vector: the phase's 64 scales (lanes 0 .. 15) | 64 shifts (lanes 16 .. 31), one piece of wave 0
    if constexpr (i == 0) {
      if (wid == 0)
        blds16(dma_rsrc(a.scale), ln < 32 ? (unsigned)((((ln >> 4) * a.nb + h_n) * cin + h_c0 + (ln & 15) * 4) * 4) : DMA_POISON, 0,
               smem + Cfg::L_TAB);
    }
  };
  // bias | temb row of unit v -> LDS (waves 0 .. 3: array wid >> 1, piece wid & 1 of 256 floats; null pointers read as zeros)
  auto issue_bias = [&](int v) {
    if (wid < 4) {
      int n, ty, tx, cb;
      decode(v, n, ty, tx, cb);
      const int arr = wid >> 1, pc = wid & 1, col = pc * 256 + rc_lane() * 4;
      const float* src = arr == 0 ? a.bias : a.bias2 ? a.bias2 + (long)(n / a.b2_imgs) * a.cout : nullptr;
      blds16(dma_rsrc(rc_uni(src ? src : a.scale)), (src && col < CB) ? (unsigned)((cb * CB + col) * 4) : DMA_POISON, 0, smem + Cfg::L_BIAS + arr * 2048 + pc * 1024);
    }
  };
  // Dwords J, J + 1 (four channels) of vector I of the halo in buffer `buf`: bf16 <- bf16( silu( . * scale + shift ) ), zeros outside the image, in five
  // STAGES that the k-step deals out behind its first five MFMA groups -- as one block its chain of dependent instructions (LDS read -> fma ->
  // exp -> rcp -> convert -> LDS write, ~300 cycles of latency) stood in front of the wave's next MFMAs (stamps: +0.15 us on a 0.9-us k-step).
  // Branch-free, scalar f32 arithmetic (csrc/gnconv.hip: packed f32 instructions cost ~25 cycles beside MFMAs); every stage's results are pinned
  // where the stage stands.
  u32x2 n_raw;
  f32x4 n_sc, n_sh, n_t, n_x;
  int n_pos = 0;
  auto norm_stage = [&](auto Kc, int buf, auto Ic, auto Jc) {
    constexpr int K = decltype(Kc)::value, I = decltype(Ic)::value, J = decltype(Jc)::value;
    if ((I == RC_NHV - 1 && wid != 0) || (abl & 4)) return;  // (only wave 0 has a sixth vector)
    if constexpr (K == 0) {
      const int ln = rc_lane();
      n_pos = buf * RC_HALO + (8 * wid + (ln >> 3)) * 128 + (ln & 7) * 16 + I * 8192 + 4 * J;   // vector I of this lane, dword J
      const int tofs = Cfg::L_TAB + ((ln & 7) ^ (ln >> 3)) * 32 + 8 * J;                          // its channel octet's scales
      n_raw = *reinterpret_cast<const u32x2*>(smem + n_pos);
      n_sc = *reinterpret_cast<const f32x4*>(smem + tofs);
      n_sh = *reinterpret_cast<const f32x4*>(smem + tofs + RC_PC * 4);
    } else if constexpr (K == 1) {
      n_t = (f32x4){fmaf(rc_lo(n_raw[0]), n_sc[0], n_sh[0]), fmaf(rc_hi(n_raw[0]), n_sc[1], n_sh[1]), fmaf(rc_lo(n_raw[1]), n_sc[2], n_sh[2]),
                    fmaf(rc_hi(n_raw[1]), n_sc[3], n_sh[3])};
#pragma unroll
      for (int e = 0; e < 4; ++e) { n_x[e] = n_t[e] * -1.4426950408889634f; asm volatile("" : "+v"(n_x[e]), "+v"(n_t[e])); }
    } else if constexpr (K == 2) {
#pragma unroll
      for (int e = 0; e < 4; ++e) { n_x[e] = __builtin_amdgcn_exp2f(n_x[e]); asm volatile("" : "+v"(n_x[e])); }
    } else if constexpr (K == 3) {
#pragma unroll
      for (int e = 0; e < 4; ++e) { n_x[e] = __builtin_amdgcn_rcpf(1.f + n_x[e]); asm volatile("" : "+v"(n_x[e])); }
    } else {
      const unsigned m = 0u - ((hmask >> I) & 1u);
      u32x2 pk = {pack_bf16x2(n_t[0] * n_x[0], n_t[1] * n_x[1]) & m, pack_bf16x2(n_t[2] * n_x[2], n_t[3] * n_x[3]) & m};
      asm volatile("" : "+v"(pk));
      *reinterpret_cast<u32x2*>(smem + n_pos) = pk;
    }
  };
  auto norm_dword = [&](int buf, auto Ic, auto Jc) {         // (prologue: the first halo, all at once; Jc even)
    rc_for<0, 5>([&](auto kc) { norm_stage(kc, buf, Ic, Jc); });
  };

  // ---- fragment addressing.  A: pixel R = rl + K with rl = 18 (4 wm) + lm and K = 18 (i + ky) + kx a compile-time constant: row R of the
  // buffer, slot (4 ks + lq) ^ (R & 7); R & 7 = (rl + (K & 7)) & 7, so eight per-lane offsets (one per K & 7) serve every tap.
  // computed where it is used from two registers (rl16: bits 6:4 = rl & 7; lq16) in three instructions per fragment -- eight offsets kept live cost
  // the 320-wide cut spills inside its k-steps.
  const int rl16 = (RC_HP * RT * wm + lm) * 16, lq16 = lq * 16;
  const int w_lane = wn * NT * 1024 + lane * 16;                           // + LDS offset of the slot + j 1 KiB
  s16x8 fa[RT], fw[RC_FWD];
  int a_rl = 0, a_base = 0;                                  // per k-step: an opaque copy of rl and the byte offset of the lane's row 0 in the halo buffer
  auto read_a_begin = [&](int buf) {                         // (opaque: the eight loop-invariant swizzles are not hoisted out of the phase loop and kept live)
    a_rl = rl16 >> 4;
    asm volatile("" : "+v"(a_rl));
    a_base = buf * RC_HALO + a_rl * 128;
  };
  auto read_a1 = [&](int buf, auto Sc, auto Ic) {             // A fragment of image row i of the wave for k-step S (after read_a_begin(buf))
    constexpr int S = decltype(Sc)::value, i = decltype(Ic)::value, tap = S >> 1, ks = S & 1, ky = tap / 3, kx = tap % 3, K = RC_HP * (i + ky) + kx;
    const int sw = ((a_rl + (K & 7)) & 7) * 16;               // ((rl + K) & 7) << 4: a multiple of 16, like everything below (one ds_read_b128)
    return *reinterpret_cast<const s16x8*>(smem + ((sw ^ (lq16 ^ (ks ? 64 : 0))) + a_base + K * 128));
  };
  auto read_w = [&](int slot_off, int j) { return *reinterpret_cast<const s16x8*>(smem + w_lane + slot_off + j * 1024); };

  // Start stagger.  Every unit of a launch is the same length, so the persistent workgroups of the whole chip reach their epilogues together: a
  // burst of nunits-per-round x 160 KB that HBM takes ~10 - 18 us to absorb while every wave's in-order vmcnt queue holds its next weight pieces
  // behind its stores.  Eight groups of workgroups start a few microseconds apart so that the bursts are spread over that time.
  for (int d = ((blockIdx.x >> 3) & 7) * a.stagger; d > 0; d -= 64) __builtin_amdgcn_s_sleep(64);

  // s_waitcnt vmcnt(W PPW + X): W k-steps of weight pieces + X other operations may stay in flight
#define RC_WAIT(W_, X_) wait_vmcnt<(W_) * PPW + (X_)>()

  // ---- prologue: bias of the first unit, its first halo (normalised here), the first NSLOT - 1 k-steps of weights
  int vt = blockIdx.x, ph = 0, fp = 0;                      // current unit, its phase, phases done (buffer parity)
  if (my_units > 0) {
    issue_bias(vt);
    halo_target(vt, 0);
    rc_for<0, RC_NHV>([&](auto ic) { issue_halo_vec(ic, 0); });
    for (int g = 0; g < NSLOT - 1; ++g) issue_w();
    RC_WAIT(NSLOT - 1, 0);                                  // bias, table, halo have landed (this wave's pieces)
    __builtin_amdgcn_s_barrier();                           // ... everybody's: the table is complete
    rc_for<0, RC_NHV>([&](auto ic) { norm_dword(0, ic, std::integral_constant<int, 0>{}); norm_dword(0, ic, std::integral_constant<int, 2>{}); });
    RC_WAIT(NSLOT - 2, 0);                                  // k-step 0 has landed
  }
  __builtin_amdgcn_s_waitcnt(0xC07F);                       // lgkmcnt(0): the LDS stores above
  __builtin_amdgcn_s_barrier();
  int c_slot = Cfg::L_RING;                                 // LDS offset of the ring slot of the k-step being multiplied
  if (my_units > 0) {
    read_a_begin(0);
    rc_for<0, RT>([&](auto ic) { fa[decltype(ic)::value] = read_a1(0, std::integral_constant<int, 0>{}, ic); });
#pragma unroll
    for (int j = 0; j < RC_FWD; ++j) fw[j] = read_w(c_slot, j);
  }

  const __amdgpu_buffer_rsrc_t rO = dma_rsrc(a.out), rR = dma_rsrc(a.res ? a.res : a.out);
  acc4 acc[RT][NT];
  bool stored = false;                                      // the previous phase ended in an epilogue: its stores are in the vmcnt queue
  while (vt < a.nunits) {
    const bool first_ph = ph == 0, last_ph = ph == a.nph - 1;
    const int nvt = last_ph ? vt + G : vt, nph = last_ph ? 0 : ph + 1;
    const bool has_next = nvt < a.nunits;
    const int buf = fp & 1, nbuf = buf ^ 1;
    if (has_next) halo_target(nvt, nph);
    if (first_ph && vt + G < a.nunits) {                    // the block offset of the weight stream's next unit (it enters that unit NSLOT - 1 k-steps early)
      int n, ty, tx, cb;
      decode(vt + G, n, ty, tx, cb);
      w_next = cb * SLOT;
    }

    if (first_ph) {
      const acc4* lb = reinterpret_cast<const acc4*>(smem + Cfg::L_BIAS) + wn * NT * 4 + lq;   // the lane's columns 16 j + 4 lq + r
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        const acc4 b = lb[4 * j] + lb[128 + 4 * j];               // bias + temb row (2 KiB apart)
#pragma unroll
        for (int i = 0; i < RT; ++i) acc[i][j] = b;
      }
      asm volatile("s_nop 7" ::: "memory");                   // (accumulator writes -> the first asm MFMA that reads them)
    }

    rc_for<0, RC_NSTEP>([&](auto sc_) {
      constexpr int S = decltype(sc_)::value;
      stamp();
      const int n_slot = c_slot == Cfg::L_RING + (NSLOT - 1) * SLOT ? Cfg::L_RING : c_slot + SLOT;
      // Two dwords of the next phase's halo per k-step, S = NSLOT .. NSLOT + 11.  The counted waits of the hand-overs leave NSLOT - 3 k-steps of
      // weight pieces and the halo pieces issued among them in flight, so the piece of hand-over h (vector h) -- and the table, issued with
      // vector 0 by wave 0 -- is only known to have landed behind hand-over h + NSLOT - 1: vector i is normalised in k-steps NSLOT + 2 i, + 1.
      // (With a six-deep ring a start at k-step 3 read LDS the DMA had not written yet: wrong rows, now and then.)
      constexpr bool NORM = S >= NSLOT && S < NSLOT + 2 * RC_NHV;
      constexpr int NV = NORM ? (S - NSLOT) >> 1 : 0, NJ = NORM ? 2 * ((S - NSLOT) & 1) : 0;
      static_assert(NSLOT + 2 * RC_NHV <= RC_NSTEP - 1, "the normalisation must end in front of the phase's last k-step (whose barrier the late waves take first)");
      auto fwi = [](int j) constexpr { return (S * NT + j) % RC_FWD; };      // the rolling window's register of W tile j of this k-step
      // ---- hand-over of k-step g: k-step g + 1 has landed (this wave's pieces; the barrier collects the others').  vmcnt retires in order, so
      // the count is the operations YOUNGER than those pieces: the pieces of the k-steps behind it, the halo pieces of the earlier hand-overs
      // (waves that issued a table / bias piece too wait for one operation more than they must), the stores of an epilogue in front of k-step 0.
      auto handover = [&]() {
        if (trace_fine) stamp();
        constexpr int HLO = S + 2 - NSLOT > 0 ? S + 2 - NSLOT : 0, HHI = S - 1 < RC_NHV - 1 ? S - 1 : RC_NHV - 1;
        constexpr int NH = HHI >= HLO ? HHI - HLO + 1 : 0;            // halo pieces issued behind the pieces waited for (one per hand-over 0 .. 5)
        constexpr bool EPI = S <= NSLOT - 3;                          // ... and the stores of an epilogue in front of k-step 0
        if constexpr (NH > 0 && EPI) {
          if (stored) { if (has_next) RC_WAIT(NSLOT - 3, NH + NST); else RC_WAIT(NSLOT - 3, NST); }
          else { if (has_next) RC_WAIT(NSLOT - 3, NH); else RC_WAIT(NSLOT - 3, 0); }
        } else if constexpr (NH > 0) {
          if (has_next) RC_WAIT(NSLOT - 3, NH); else RC_WAIT(NSLOT - 3, 0);
        } else if constexpr (EPI) {
          if (stored) RC_WAIT(NSLOT - 3, NST); else RC_WAIT(NSLOT - 3, 0);
        } else {
          RC_WAIT(NSLOT - 3, 0);
        }
        if constexpr (S == RC_NSTEP - 1) __builtin_amdgcn_s_waitcnt(0xC07F);   // lgkmcnt(0): this thread's normalised dwords are in LDS
        if (trace_fine) stamp();
        if (!(abl & 8)) __builtin_amdgcn_s_barrier();
        if (trace_fine) stamp();
        issue_w();                                          // into the slot k-step g - 1 left
        if constexpr (S < RC_NHV) {
          if (has_next) {
            issue_halo_vec(std::integral_constant<int, S>{}, nbuf);
            if constexpr (S == 0) { if (last_ph) issue_bias(nvt); }
          }
        }
      };
      // ---- W tiles 0 .. NT - 3: tile-major (a W fragment serves the wave's four image rows, then its register takes the tile four ahead)
      rc_for<0, NT - 2>([&](auto jc_) {
        constexpr int j = decltype(jc_)::value;
        rc_for<0, RT>([&](auto ic) { rc_mma<(j < 8)>(acc[decltype(ic)::value][j], fw[fwi(j)], fa[decltype(ic)::value]); });
        if constexpr (NORM && j < 5) norm_stage(std::integral_constant<int, j>{}, nbuf, std::integral_constant<int, NV>{}, std::integral_constant<int, NJ>{});
        if constexpr (j + RC_FWD < NT) fw[fwi(j)] = read_w(c_slot, j + RC_FWD);
        // ---- hand-over (see `handover` above), behind the last read of the current slot.  (Measured and dropped: waves 4 - 7 -- the SIMD partners
        // of waves 0 - 3 -- taking theirs five tile groups earlier, i.e. running half a k-step behind, MI355X_MICROARCH.md's stagger for "two waves
        // that run the same program with one barrier per block": 0 ... +3 % on the four in-step shapes, profiles/r6/bench_rconv_wstagger_r6.txt.)
        if constexpr (j + RC_FWD == NT) handover();
        if constexpr (j + RC_FWD >= NT) {
          // (no fragments are carried across an epilogue: it needs their 32 registers, and the unit's first reads cost one LDS round trip per ~100 000 cycles)
          if (S + 1 < RC_NSTEP || !last_ph) fw[fwi(j)] = read_w(n_slot, j + RC_FWD - NT);
        }
      });
      // ---- the last two W tiles row-major: an A fragment's register takes the next k-step's fragment as soon as its two MFMAs have issued
      // (k-step 17: from the next phase's halo, complete behind the barrier above)
      read_a_begin(S + 1 < RC_NSTEP ? buf : nbuf);
      rc_for<0, RT>([&](auto ic) {
        constexpr int i = decltype(ic)::value;
        rc_mma<(NT - 2 < 8)>(acc[i][NT - 2], fw[fwi(NT - 2)], fa[i]);
        rc_mma<(NT - 1 < 8)>(acc[i][NT - 1], fw[fwi(NT - 1)], fa[i]);
        if constexpr (S + 1 < RC_NSTEP) fa[i] = read_a1(buf, std::integral_constant<int, S + 1>{}, ic);
        else if (!last_ph) fa[i] = read_a1(nbuf, std::integral_constant<int, 0>{}, ic);
      });
      if (S + 1 < RC_NSTEP || !last_ph) {
        fw[fwi(NT - 2)] = read_w(n_slot, RC_FWD - 2);
        fw[fwi(NT - 1)] = read_w(n_slot, RC_FWD - 1);
      }
      __builtin_amdgcn_sched_barrier(0);
      c_slot = n_slot;
    });
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");        // (the asm MFMAs hide their result latency from hipcc: nothing below reads an accumulator earlier)
    stored = false;

    if (last_ph) {
      // ---- epilogue (gemm16.hip's idiom): lane (lm, lq) holds pixel lm of image row 4 wm + i and, per tile j, channels 16 j + 4 lq + r;
      // v_permlane16_swap of tiles 2 jp, 2 jp + 1 -> 8 consecutive channels 32 jp + 16 (lq & 1) + 8 (lq >> 1) .. + 7: one 16-byte store.
      stamp();
      int n, ty, tx, cb;
      decode(vt, n, ty, tx, cb);
      const int lne = rc_lane(), lme = lne & 15, lqe = lne >> 4;   // (nothing of this is hoisted above the main loop or kept live across it)
      const int cofs = cb * CB + wn * NT * 16 + 16 * (lqe & 1) + 8 * (lqe >> 1);
      const unsigned eoff = (unsigned)((((n * a.H + ty * RC_T + RT * wm) * a.W + tx * RC_T + lme) * a.cout + cofs) * 2);   // byte offset of (row 0, pair 0)
      const int erow = a.W * a.cout * 2;                      // bytes per image row
      auto out8 = [&](auto ic, auto jpc, const u32x4& rvv) {   // the lane's 8 output channels of image row i, tile pair jp, rounded and packed
        constexpr int i = decltype(ic)::value, jp = decltype(jpc)::value;
        const acc4 x = rc_get<(2 * jp < 8)>(acc[i][2 * jp]), y = rc_get<(2 * jp + 1 < 8)>(acc[i][2 * jp + 1]);
        if constexpr (!RES) {
          const auto s01 = __builtin_amdgcn_permlane16_swap(pack_bf16x2(x[0], x[1]), pack_bf16x2(y[0], y[1]), false, false);
          const auto s23 = __builtin_amdgcn_permlane16_swap(pack_bf16x2(x[2], x[3]), pack_bf16x2(y[2], y[3]), false, false);
          return (u32x4){s01[0], s23[0], s01[1], s23[1]};
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
            o8[2 * e] += rc_lo(rvv[e]);
            o8[2 * e + 1] += rc_hi(rvv[e]);
          }
          return (u32x4){pack_bf16x2(o8[0], o8[1]), pack_bf16x2(o8[2], o8[3]), pack_bf16x2(o8[4], o8[5]), pack_bf16x2(o8[6], o8[7])};
        }
      };
      if constexpr (ST) {
        // pair-major: a tile pair's 8 channels x RT rows are tallied in 16 registers, folded over the 16 pixels of the row tiles by four DPP steps
        // (fixed order) and written by lane lm = 0 of each lq; the pivot of a channel is the value of (row 0, pixel 0) of the partial
        const __amdgpu_buffer_rsrc_t rS = dma_rsrc(a.stats);
        const int part = ((n * a.tiles_per_img + ty * a.tiles_x + tx) * (RC_T / RT) + wm);
        const int nparts = a.nb * a.tiles_per_img * (RC_T / RT);
        u32x4 rv[2][RES ? RT : 1];
        if constexpr (RES) {
#pragma unroll
          for (int i = 0; i < RT; ++i) rv[0][i] = __builtin_amdgcn_raw_buffer_load_b128(rR, (int)eoff + i * erow, 0, 0);
        }
        rc_for<0, NPAIR>([&](auto jpc) {
          constexpr int jp = decltype(jpc)::value;
          if constexpr (RES && jp + 1 < NPAIR) {
#pragma unroll
            for (int i = 0; i < RT; ++i) rv[(jp + 1) & 1][i] = __builtin_amdgcn_raw_buffer_load_b128(rR, (int)eoff + i * erow + 64 * (jp + 1), 0, 0);
          }
          float piv[8], s1[8], s2[8];
          rc_for<0, RT>([&](auto ic) {
            constexpr int i = decltype(ic)::value;
            const u32x4 pk = out8(ic, jpc, rv[jp & 1][RES ? i : 0]);
            __builtin_amdgcn_raw_buffer_store_b128(pk, rO, (int)eoff + i * erow + 64 * jp, 0, MMGT_ST_AUX);
            float v[8];
#pragma unroll
            for (int e = 0; e < 4; ++e) { v[2 * e] = rc_lo(pk[e]); v[2 * e + 1] = rc_hi(pk[e]); }
            if constexpr (i == 0) {
#pragma unroll
              for (int e = 0; e < 8; ++e) {
                piv[e] = __uint_as_float((unsigned)__builtin_amdgcn_ds_bpermute((lne & 48) << 2, (int)__float_as_uint(v[e])));   // pixel 0 of the row tile
                s1[e] = 0.f;
                s2[e] = 0.f;
              }
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) { const float d = v[e] - piv[e]; s1[e] += d; s2[e] = fmaf(d, d, s2[e]); }
          });
          auto fold16 = [](float v) {                        // sum over the 16 lanes of a DPP row, fixed order (quad, quad pairs, halves, row)
            v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xf, 0xf, false));
            v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xf, 0xf, false));
            v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xf, 0xf, false));
            v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xf, 0xf, false));
            return v;
          };
#pragma unroll
          for (int e = 0; e < 8; ++e) { s1[e] = fold16(s1[e]); s2[e] = fold16(s2[e]); }
          // lane lm = 0 of each lq: [3][partials][cout] floats, 8 consecutive channels of each array = two 16-byte stores
          // (the array offset rides in the VECTOR offset too: no register in soffset -- the store-data hazard of csrc/gnconv.hip)
          const unsigned soff = (unsigned)((part * a.cout + cofs + 32 * jp) * 4), arr = (unsigned)(nparts * a.cout * 4);
          const unsigned vo0 = lme == 0 ? soff : DMA_POISON, vo1 = lme == 0 ? soff + arr : DMA_POISON, vo2 = lme == 0 ? soff + 2 * arr : DMA_POISON;   // (out of range: dropped)
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            __builtin_amdgcn_raw_buffer_store_b128((u32x4){__float_as_uint(piv[4 * h]), __float_as_uint(piv[4 * h + 1]), __float_as_uint(piv[4 * h + 2]), __float_as_uint(piv[4 * h + 3])}, rS, (int)vo0 + 16 * h, 0, 0);
            __builtin_amdgcn_raw_buffer_store_b128((u32x4){__float_as_uint(s1[4 * h]), __float_as_uint(s1[4 * h + 1]), __float_as_uint(s1[4 * h + 2]), __float_as_uint(s1[4 * h + 3])}, rS, (int)vo1 + 16 * h, 0, 0);
            __builtin_amdgcn_raw_buffer_store_b128((u32x4){__float_as_uint(s2[4 * h]), __float_as_uint(s2[4 * h + 1]), __float_as_uint(s2[4 * h + 2]), __float_as_uint(s2[4 * h + 3])}, rS, (int)vo2 + 16 * h, 0, 0);
          }
          __builtin_amdgcn_sched_barrier(0);
        });
      } else {
      u32x4 rv[2][RES ? NPAIR : 1];                           // residual vectors, one image row ahead of their use (one at a time: 0.8 us of latency each, 16 us per unit)
      if constexpr (RES) {
#pragma unroll
        for (int jp = 0; jp < NPAIR; ++jp) rv[0][jp] = __builtin_amdgcn_raw_buffer_load_b128(rR, (int)eoff + 64 * jp, 0, 0);
      }
      rc_for<0, RT * NPAIR>([&](auto qc) {
        constexpr int q = decltype(qc)::value, i = q / NPAIR, jp = q % NPAIR;
        if constexpr (RES && jp == 0 && i + 1 < RT) {
#pragma unroll
          for (int j2 = 0; j2 < NPAIR; ++j2) rv[(i + 1) & 1][j2] = __builtin_amdgcn_raw_buffer_load_b128(rR, (int)eoff + (i + 1) * erow + 64 * j2, 0, 0);
        }
        const u32x4 pk = out8(std::integral_constant<int, i>{}, std::integral_constant<int, jp>{}, rv[i & 1][RES ? jp : 0]);
        // (the row offset rides in the VECTOR offset, not in soffset: see csrc/gnconv.hip -- the store-data hazard tools/check_mfma_overlap.py scans for)
        if (!(abl & 16) || pk[0] == 0x12345678u) __builtin_amdgcn_raw_buffer_store_b128(pk, rO, (int)eoff + i * erow + 64 * jp, 0, MMGT_ST_AUX);
        __builtin_amdgcn_sched_barrier(0);
      });
      }
      // the next unit's first fragments (its halo and its first k-step are in LDS behind the barrier of k-step 17)
      if (has_next) {
        read_a_begin(nbuf);
        rc_for<0, RT>([&](auto ic) { fa[decltype(ic)::value] = read_a1(nbuf, std::integral_constant<int, 0>{}, ic); });
#pragma unroll
        for (int j = 0; j < RC_FWD; ++j) fw[j] = read_w(c_slot, j);
      }
      stored = true;
      stamp();
    }
    vt = nvt;
    ph = nph;
    ++fp;
  }
#undef RC_WAIT
  wait_vmcnt<0>();                                           // (the poison pieces write LDS: none may be in flight when the workgroup's LDS is released)
}

// The tables of a GroupNorm from the partials of the launch that produced its input: stats [3][parts][C] = (pivot, sum, sum of squares) of the
// deviations from the pivot over `cnt` values per (partial, channel).  One workgroup per image; thread (group g, run r of 256 / G) folds the partials
// r, r + runs, ... of its group's channels with Chan's pairwise update -- mean and M2 of a union from the means and M2s of its parts, every term of
// the order of the variance --, then the runs are folded in a fixed order.  Bitwise reproducible.
__global__ __launch_bounds__(256) void rconv_stats_finalize_kernel(const float* __restrict__ stats, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                                   float* __restrict__ scale, float* __restrict__ shift, int nb, int ppi, int C, int G, float cnt,
                                                                   float eps) {
  // a workgroup = one image x 8 groups, 32 runs per group (48 x 4 workgroups at the step's shapes: a fold is a chain of dependent updates, short chains)
  constexpr int GPB = 8, NRUN = 256 / GPB;
  __shared__ float pn[NRUN][GPB], pm[NRUN][GPB], pq[NRUN][GPB];
  const int n = blockIdx.x, gl = threadIdx.x % GPB, g = blockIdx.y * GPB + gl, run = threadIdx.x / GPB, cpg = C / G;
  const long arr = (long)nb * ppi * C;
  float cn = 0.f, cm = 0.f, cq = 0.f;                      // count, mean, sum of squared deviations from the mean
  const float icnt = 1.f / cnt;
  if (g < G)
    for (int p = run; p < ppi; p += NRUN) {
      const float* b = stats + ((long)n * ppi + p) * C + g * cpg;
      for (int k = 0; k < cpg; ++k) {
        const float s1 = b[arr + k], mb = b[k] + s1 * icnt, qb = fmaxf(b[2 * arr + k] - s1 * s1 * icnt, 0.f);
        const float tot = cn + cnt, d = mb - cm, f = cnt * __builtin_amdgcn_rcpf(tot);
        cm = fmaf(d, f, cm);
        cq += qb + d * d * cn * f;
        cn = tot;
      }
    }
  pn[run][gl] = cn;
  pm[run][gl] = cm;
  pq[run][gl] = cq;
  __syncthreads();
  if (run == 0 && g < G) {
    for (int r = 1; r < NRUN; ++r) {
      const float nb_ = pn[r][gl];
      if (nb_ > 0.f) {
        const float tot = cn + nb_, d = pm[r][gl] - cm, f = nb_ / tot;
        cm = fmaf(d, f, cm);
        cq += pq[r][gl] + d * d * cn * f;
        cn = tot;
      }
    }
    const float rstd = rsqrtf(cq / cn + eps);
    for (int c = g * cpg; c < (g + 1) * cpg; ++c) {
      const float sc = gamma[c] * rstd;
      scale[(long)n * C + c] = sc;
      shift[(long)n * C + c] = fmaf(-cm, sc, beta[c]);
    }
  }
}

int g_rconv_abl = 0, g_rconv_stagger = 0, g_rconv_cb = 0;
unsigned long long* g_rconv_trace = nullptr;
int g_rconv_fine = 0;                  // (mmgt_rconv_set_trace(buf | 1): four stamps per k-step)

}  // namespace

void mmgt_rconv_set_abl(int v) { g_rconv_abl = v; }
void mmgt_rconv_set_stagger(int v) { g_rconv_stagger = v; }
void mmgt_rconv_set_cb(int v) { g_rconv_cb = v; }
extern "C" void mmgt_rconv_set_trace(void* p) { g_rconv_trace = reinterpret_cast<unsigned long long*>((uintptr_t)p & ~(uintptr_t)1); g_rconv_fine = (int)((uintptr_t)p & 1); }

// x0 (nb, H, W, C0) [+ x1 (nb, H, W, C1)] bf16 channels-last, H and W multiples of 16; scale | shift (2, nb, C0 + C1) fp32 in one allocation;
// wimg = pack_rconv image of the (Cout, C0 + C1, 3, 3) weight; bias (Cout) / bias2 (rows, Cout) fp32 or null, image n takes row n / b2_imgs;
// residual / out (nb, H, W, Cout) bf16.  C0, C1 multiples of 64, Cout a multiple of 160.
namespace {
// the cut (block width) of a launch: the one that needs the fewest rounds of the persistent grid, weighted by what a unit costs -- its MFMA work grows
// with CB, its halo traffic, normalisation and epilogue hand-over do not (the 40).  mmgt_tune("rconv_cb", 320 | 256 | 160) forces one.
int rconv_cut(int nb, int H, int W, int cout, int cus) {
  const int tiles = nb * (H / RC_T) * (W / RC_T);
  int cb = 0;
  long best = 0;
  for (int c : {320, 256, 160}) {
    if (cout % c || (g_rconv_cb && g_rconv_cb != c)) continue;
    const long units = (long)tiles * (cout / c), score = (units + cus - 1) / cus * (c + 40);
    if (!cb || score < best) { cb = c; best = score; }
  }
  return cb;
}
int rconv_cus(int* out) {
  int dev = 0;
  static int ncu[16] = {};
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return -1;
  if (!ncu[dev]) {
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, dev) != hipSuccess) return -1;
    ncu[dev] = prop.multiProcessorCount;
  }
  *out = ncu[dev] / 8 * 8;
  return dev;
}
}  // namespace

// The image rows per statistics partial (4 or 2) of the cut mmgt_gn_silu_conv3x3_unet_stats will take for this shape: its `stats` buffer holds
// 3 * nb * (H / 16) * (W / 16) * (16 / rows) * Cout floats.
extern "C" int mmgt_gn_silu_conv3x3_unet_stats_rows(int nb, int H, int W, int cout) {
  int cus = 0;
  if (rconv_cus(&cus) < 0 || nb <= 0 || H <= 0 || W <= 0 || H % RC_T || W % RC_T) return -1;
  const int cb = rconv_cut(nb, H, W, cout, cus);
  return cb == 0 ? -1 : cb == 160 ? 2 : 4;
}

extern "C" int mmgt_gn_silu_conv3x3_unet_stats(const void* x0, int C0, const void* x1, int C1, const float* scale_shift, const void* wimg, const float* bias,
                                               const float* bias2, int b2_imgs, const void* residual, void* out, float* stats, int nb, int H, int W, int cout,
                                               int dtype, void* stream) {
  MMGT_CHECK(x0 && scale_shift && wimg && out && nb > 0 && H > 0 && W > 0, "gn_silu_conv3x3_unet: bad arguments");
  MMGT_CHECK(dtype == MMGT_BF16, "gn_silu_conv3x3_unet: bf16 only (the fp32-I/O mode runs GroupNorm / conv)");
  MMGT_CHECK((x1 != nullptr) == (C1 > 0) && C0 > 0 && C0 % RC_PC == 0 && C1 % RC_PC == 0, "gn_silu_conv3x3_unet: C0 = %d, C1 = %d must be multiples of %d", C0, C1, RC_PC);
  MMGT_CHECK(cout > 0 && cout % 160 == 0, "gn_silu_conv3x3_unet: Cout = %d must be a multiple of 160", cout);
  MMGT_CHECK(H % RC_T == 0 && W % RC_T == 0, "gn_silu_conv3x3_unet: H and W must be multiples of 16 (got %d x %d)", H, W);
  MMGT_CHECK(!bias2 || b2_imgs > 0, "gn_silu_conv3x3_unet: bias2 needs b2_imgs > 0");
  const long cin = (long)C0 + C1;
  MMGT_CHECK((long)nb * H * W * (C0 > C1 ? C0 : C1) * 2 < (1l << 31) && (long)nb * H * W * cout * 2 < (1l << 31) && (long)cout * cin * 18 < (1l << 31) &&
                 2l * nb * cin * 4 < (1l << 31),
             "gn_silu_conv3x3_unet: every operand must be smaller than 2 GiB");
  MMGT_CHECK((((uintptr_t)x0 | (uintptr_t)x1 | (uintptr_t)scale_shift | (uintptr_t)wimg | (uintptr_t)bias | (uintptr_t)bias2 | (uintptr_t)residual | (uintptr_t)out) & 15) == 0,
             "gn_silu_conv3x3_unet: pointers must be 16-byte aligned");
  RcArgs a{};
  a.x0 = reinterpret_cast<const bf16_t*>(x0);
  a.x1 = reinterpret_cast<const bf16_t*>(x1);
  a.C0 = C0;
  a.C1 = C1;
  a.scale = scale_shift;
  a.wimg = reinterpret_cast<const char*>(wimg);
  a.bias = bias;
  a.bias2 = bias2;
  a.b2_imgs = b2_imgs > 0 ? b2_imgs : 1;
  a.res = reinterpret_cast<const bf16_t*>(residual);
  a.out = reinterpret_cast<bf16_t*>(out);
  a.nb = nb;
  a.H = H;
  a.W = W;
  a.tiles_x = W / RC_T;
  a.tiles_per_img = (H / RC_T) * (W / RC_T);
  a.nph = (int)(cin / RC_PC);
  a.cout = cout;
  a.abl = g_rconv_abl;
  a.stagger = g_rconv_stagger;
  a.trace = g_rconv_trace;
  a.trace_fine = g_rconv_fine;
  int cus = 0;
  const int dev = rconv_cus(&cus);
  MMGT_CHECK(dev >= 0, "gn_silu_conv3x3_unet: device query failed");
  MMGT_CHECK(!stats || ((uintptr_t)stats % 16) == 0, "gn_silu_conv3x3_unet: stats must be 16-byte aligned");
  a.stats = stats;
  const int tiles = nb * (H / RC_T) * (W / RC_T);
  const int cb = rconv_cut(nb, H, W, cout, cus);
  MMGT_CHECK(cb, "gn_silu_conv3x3_unet: no block width for Cout = %d (rconv_cb = %d)", cout, g_rconv_cb);
  a.ncb = cout / cb;
  a.nunits = tiles * a.ncb;
  int gx = cus;
  if (gx > a.nunits) gx = a.nunits;
  static bool ready[16][12] = {};
  auto go = [&](void (*kern)(const RcArgs), int lds, int slot) -> int {
    if (!ready[dev][slot]) {
      if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess) {
        mmgt_set_error("gn_silu_conv3x3_unet: cannot reserve %d bytes of LDS", lds);
        return 2;
      }
      ready[dev][slot] = true;
    }
    hipLaunchKernelGGL(kern, dim3(gx), dim3(512), lds, (hipStream_t)stream, a);
    return 0;
  };
  MMGT_CHECK(!stats || 3l * tiles * (RC_T / (cb == 160 ? 2 : 4)) * cout * 4 < (1l << 31), "gn_silu_conv3x3_unet: statistics buffer beyond 2 GiB");
  int rc;
  if (stats) {
    if (cb == 320) rc = residual ? go(rconv_kernel<10, 4, true, true>, RcCfg<10, 4>::LDS, 7) : go(rconv_kernel<10, 4, false, true>, RcCfg<10, 4>::LDS, 6);
    else if (cb == 256) rc = residual ? go(rconv_kernel<8, 4, true, true>, RcCfg<8, 4>::LDS, 9) : go(rconv_kernel<8, 4, false, true>, RcCfg<8, 4>::LDS, 8);
    else rc = residual ? go(rconv_kernel<10, 2, true, true>, RcCfg<10, 2>::LDS, 11) : go(rconv_kernel<10, 2, false, true>, RcCfg<10, 2>::LDS, 10);
  } else if (cb == 320) rc = residual ? go(rconv_kernel<10, 4, true>, RcCfg<10, 4>::LDS, 1) : go(rconv_kernel<10, 4, false>, RcCfg<10, 4>::LDS, 0);
  else if (cb == 256) rc = residual ? go(rconv_kernel<8, 4, true>, RcCfg<8, 4>::LDS, 3) : go(rconv_kernel<8, 4, false>, RcCfg<8, 4>::LDS, 2);
  else rc = residual ? go(rconv_kernel<10, 2, true>, RcCfg<10, 2>::LDS, 5) : go(rconv_kernel<10, 2, false>, RcCfg<10, 2>::LDS, 4);
  if (rc) return rc;
  MMGT_LAUNCH_CHECK();
  return 0;
}

extern "C" int mmgt_gn_silu_conv3x3_unet(const void* x0, int C0, const void* x1, int C1, const float* scale_shift, const void* wimg, const float* bias,
                                         const float* bias2, int b2_imgs, const void* residual, void* out, int nb, int H, int W, int cout, int dtype,
                                         void* stream) {
  return mmgt_gn_silu_conv3x3_unet_stats(x0, C0, x1, C1, scale_shift, wimg, bias, bias2, b2_imgs, residual, out, nullptr, nb, H, W, cout, dtype, stream);
}

// stats [3][nb * parts_per_img][C] as written by mmgt_gn_silu_conv3x3_unet_stats (parts_per_img = (H / 16) (W / 16) (16 / rows), `count` = 16 rows values
// per partial and channel) -> scale | shift (2, nb, C) of GroupNorm(G groups, gamma, beta, eps) over the stored tensor.  G <= 64 and a divisor of 256.
extern "C" int mmgt_gn_stats_finalize_unet(const float* stats, const float* gamma, const float* beta, float* scale_shift, int nb, int parts_per_img, int count,
                                           int C, int G, float eps, void* stream) {
  MMGT_CHECK(stats && gamma && beta && scale_shift && nb > 0 && parts_per_img > 0 && count > 0, "gn_stats_finalize_unet: bad arguments");
  MMGT_CHECK(G > 0 && G <= 64 && C % G == 0, "gn_stats_finalize_unet: unsupported C = %d, G = %d", C, G);
  hipLaunchKernelGGL(rconv_stats_finalize_kernel, dim3(nb, (G + 7) / 8), dim3(256), 0, (hipStream_t)stream, stats, gamma, beta, scale_shift, scale_shift + (long)nb * C, nb,
                     parts_per_img, C, G, (float)count, eps);
  MMGT_LAUNCH_CHECK();
  return 0;
}

extern "C" long mmgt_gn_silu_conv3x3_unet_image_bytes(int cin, int cout) {
  return (cin > 0 && cin % RC_PC == 0 && cout > 0 && cout % 160 == 0) ? (long)cin * 9 * cout * 2 : -1;
}
