// GEMM / implicit-GEMM conv3x3 for the MMGT Stage-2 path (gfx950).
//
//   out[M, N] = epilogue( A[M, K] * W[N, K]^T )          W in torch nn.Linear layout ([out, in], K contiguous)
//
// A is either a dense row-major matrix (Linear, 1x1 conv on channels-last tokens) or the implicit im2col view of a
// channels-last (N, H, W, C) tensor for a 3x3 / pad 1 convolution (stride 1 or 2, optional fused nearest-2x upsample of
// the input, optional second source tensor = fused channel concat of the UNet skip connection).  K ordering of the conv
// view is (ky, kx, cin) with cin fastest, matching weights pre-packed as [Cout][3][3][Cin].
//
// Structure (one template over storage type, A view and tile shape):
//  * BM x BN tile per workgroup of 4 or 8 waves, K chunks of 128 bytes per row (64 bf16 / 32 fp32).
//  * Both operands go global -> LDS by LDS-DMA through buffer resources (buffer_load_dwordx4 ... offen lds: an SGPR
//    descriptor, one 32-bit VGPR offset per lane and an SGPR offset along K; no staging registers, no per-chunk vector
//    address arithmetic -- against global_load_lds with 64-bit per-lane addresses this measured -8..-18% on every shape).
//    LDS rows are exactly 128 B and lane-linear per wave instruction, so bank conflicts are removed by an XOR swizzle of
//    the 16-byte chunk index applied on the per-lane SOURCE offset and again on the fragment read
//    (chunk ^ ((row >> 1) & 7): the sixteen lanes of every ds_read_b128 group hit sixteen distinct 16-byte slots).
//    Conv zero padding = an out-of-range buffer offset, which the hardware reads as zero.
//  * NSTAGE-deep LDS ring, one raw s_barrier per chunk, counted s_waitcnt vmcnt so NSTAGE-2 chunks stay in flight
//    across the barrier; persistent grid whose chunk stream keeps rolling across tile boundaries.
//  * Epilogue from registers (MFMA operands swapped, permlane32 / DPP quad transposes) so that every lane loads/stores 8
//    consecutive columns (16 B bf16): bias, per-batch bias, SiLU / GEGLU, row scale, alpha, residual, store.
//  * XCD-aware tile order (n fastest inside an XCD's contiguous run) so the tiles sharing an A row panel hit one L2.
#include <string.h>

#include <type_traits>

#include "common.h"
#include "gemm_common.h"
#include "mmgt_hip.h"

// Main-loop variant (A/B-able by building a second library with -DMMGT_GEMM_VARIANT=n, tools/ab_gemm.py):
//   bit 0: fragments of K-step ks+1 are read while the MFMAs of K-step ks run (register double buffering) + s_setprio
//   bit 1: dense 8-wave tiles spread the next chunk's LDS-DMA issue between the MFMA groups instead of one burst
//   bit 2: the per-chunk barrier sits inside the chunk's MFMA work and the next chunk's first fragments are read early
//          (applied to the 320-column tiles, whose 2-deep ring depends on it; bit 3 forces it on every tile for A/B runs)
#ifndef MMGT_GEMM_VARIANT
#define MMGT_GEMM_VARIANT 7
#endif

namespace {

constexpr bool V_FRAGDB = (MMGT_GEMM_VARIANT & 1) != 0;
constexpr bool V_ILV = (MMGT_GEMM_VARIANT & 2) != 0;
constexpr bool V_LATE = (MMGT_GEMM_VARIANT & 4) != 0;
constexpr bool V_LATE_ALL = (MMGT_GEMM_VARIANT & 8) != 0;
constexpr bool V_NODMA = (MMGT_GEMM_VARIANT & 16) != 0;   // timing diagnostic only (wrong results): no LDS-DMA after the ring fill

// Transpose of 16-byte elements between the S (4 or 2) lanes r = lane & (S - 1) of a DPP quad and a lane's S registers:
// afterwards x[s] of lane r holds what x[r] of lane s held (an involution).  Butterfly of quad_perm moves: per exchanged
// register pair and dword one select for the value sent, one v_mov_dpp, two selects.
template <int CTRL>
__device__ __forceinline__ unsigned dpp_quad(unsigned v) {
  return (unsigned)__builtin_amdgcn_mov_dpp((int)v, CTRL, 0xf, 0xf, true);
}
template <int S>
__device__ __forceinline__ void quad_transpose(u32x4 (&x)[S], int lane) {
  static_assert(S == 2 || S == 4, "quad_transpose");
  unsigned a[S][4];   // scalars: element-wise selects on arrays of vectors are lowered through scratch
#pragma unroll
  for (int k = 0; k < S; ++k)
#pragma unroll
    for (int d = 0; d < 4; ++d) a[k][d] = x[k][d];
  {
    const bool bit = (lane & 1) != 0;
#pragma unroll
    for (int lo = 0; lo < S; lo += 2)
#pragma unroll
      for (int d = 0; d < 4; ++d) {
        const unsigned p = a[lo][d], q = a[lo + 1][d];
        const unsigned recv = dpp_quad<0xB1>(bit ? p : q);   // quad_perm [1, 0, 3, 2]
        a[lo][d] = bit ? recv : p;
        a[lo + 1][d] = bit ? q : recv;
      }
  }
  if (S == 4) {
    const bool bit = (lane & 2) != 0;
#pragma unroll
    for (int lo = 0; lo < 2; ++lo)
#pragma unroll
      for (int d = 0; d < 4; ++d) {
        const unsigned p = a[lo][d], q = a[lo + 2 < S ? lo + 2 : lo][d];
        const unsigned recv = dpp_quad<0x4E>(bit ? p : q);   // quad_perm [2, 3, 0, 1]
        a[lo][d] = bit ? recv : p;
        a[lo + 2 < S ? lo + 2 : lo][d] = bit ? q : recv;
      }
  }
#pragma unroll
  for (int k = 0; k < S; ++k) x[k] = (u32x4){a[k][0], a[k][1], a[k][2], a[k][3]};
}

// MODE 0 dense, 1 conv3x3.  WM x WN waves (4 or 8 per workgroup).
//
// The grid is persistent: workgroup b computes tiles b, b + gridDim.x, ... and the chunk stream of its LDS ring keeps
// rolling across tile boundaries -- while the last chunks of tile t are multiplied the first chunks of tile t + 1 are
// already in flight, and the epilogue of tile t (which borrows the ring stage consumed last) runs with them landing
// and its stores draining under the next main loop.  Measured before this: a 256 x 128 tile paid about 7 us of launch +
// first-load latency + store drain per tile, as much as a K = 640 main loop.
template <typename T, int MODE, int BM, int BN, int WM, int WN, int NSTAGE, int ROWB, int EPI>
__global__ __launch_bounds__(WM * WN * 64, (WM * WN == 4 && (BM / WM / 32) * (BN / WN / 32) <= 2) ? 3 : 2) void gemm_kernel(ADesc ad, const char* __restrict__ W, long bsw, Epi ep, int M, int N,
                                                   int K, int tiles_m, int tiles_n) {
  extern __shared__ __attribute__((aligned(1024))) char smem[];
  constexpr int ESZ = sizeof(T);
  constexpr int BK = ROWB / ESZ;                 // elements per chunk
  constexpr int KS = BK / 16;                    // MFMA K-steps per chunk
  constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
  constexpr int A_BYTES = BM * ROWB, B_BYTES = BN * ROWB, STAGE_BYTES = A_BYTES + B_BYTES;
  constexpr int NW = WM * WN;
  // ROWB = bytes of K per tile row per chunk (128, or 64 for the 256 x 256 tile whose 4-deep ring would not fit otherwise).
  // One LDS-DMA wave instruction fills 1 KiB = RPD consecutive rows of CPR 16-byte chunks.
  constexpr int CPR = ROWB / 16, RPD = 1024 / ROWB;
  constexpr int GA = BM / RPD / NW, GB = BN / RPD / NW;   // LDS-DMA groups per wave for A and B
  static_assert(ROWB == 128 || ROWB == 64, "ROWB");
  static_assert((NW == 4 || NW == 8) && BM % (WM * 32) == 0 && BN % (WN * 32) == 0 && BM % (RPD * NW) == 0 &&
                    BN % (RPD * NW) == 0 && KS >= 1, "tile");
  // XOR swizzle of the 16-byte chunk index by the row: the sixteen lanes of every ds_read_b128 service group (rows r, r+12,
  // r+20.. of one chunk column) must hit sixteen distinct 16-byte slots of the 256-byte bank period = 2 rows of 128 B or
  // 4 rows of 64 B.
  auto swz = [](int row) { return ROWB == 128 ? ((row >> 1) & 7) : ((row >> 2) & 3); };

  const int nwg = tiles_m * tiles_n;
  const int bz = blockIdx.z;
  const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wid / WN, wn = wid % WN;
  const int lr = lane & 31, lh = lane >> 5;

  // virtual tile id -> (tm, tn): XCD x (= id & 7, the hardware's round-robin placement) owns a contiguous run of tiles,
  // n fastest, so the tiles that share an A row panel are computed next to each other on one L2.
  auto decode = [&](int v, int& tm, int& tn) {
    const int q = nwg >> 3, r = nwg & 7, x = v & 7;
    const int t = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (v >> 3);
    tm = t / tiles_n;
    tn = t - tm * tiles_n;
  };

  // ---- LDS-DMA source addressing: wave `wid` fills 8-row groups g = wid * GA + i; lane -> (row l>>3, slot l&7) ----
  const int srow = lane / CPR, spos = lane % CPR;
  const T* a0 = reinterpret_cast<const T*>(ad.src0) + (long)bz * ad.bs0;
  const T* a1 = ad.src1 ? reinterpret_cast<const T*>(ad.src1) + (long)bz * ad.bs1 : nullptr;
  const T* wbase = reinterpret_cast<const T*>(W) + (long)bz * bsw;

  const __amdgpu_buffer_rsrc_t rA0 = dma_rsrc(a0), rA1 = dma_rsrc(a1 ? a1 : a0), rW = dma_rsrc(wbase);
  unsigned aoff[GA];         // byte offset of (row, swizzled chunk) at k = 0 (dense) / of the current tap's pixel (conv)
  int cn[GA], coy[GA], cox[GA], achunk[GA];
  unsigned woff[GB];
  // Chunks are prepared strictly in order, so the conv view keeps a running position (tap, channel): inside one tap and
  // one source tensor consecutive chunks are 128 B apart, and the (ky, kx) / padding / pixel address arithmetic is redone
  // only when the tap or the source changes (every Cin / 64 chunks instead of every chunk).
  int p_tap = 0, p_c = 0;
  int a_soff = 0;            // scalar byte offset of the chunk being issued, along K (dense) / inside the tap (conv)
  bool a_second = false;     // conv: the chunk comes from the second source tensor
  auto setup = [&](int tm, int tn) {   // operand addresses of tile (tm, tn), the tile the DMA stream is in
#pragma unroll
    for (int i = 0; i < GA; ++i) {
      const int row = (wid * GA + i) * RPD + srow;
      const int chunk = spos ^ swz(row);
      int m = tm * BM + row;
      if (m >= M) m = M - 1;
      if (MODE == 0) {
        aoff[i] = (unsigned)((long)m * ad.ld0 * ESZ) + chunk * 16;
      } else {
        const int hw = ad.OH * ad.OW;
        cn[i] = m / hw;
        const int rem = m - cn[i] * hw;
        coy[i] = rem / ad.OW;
        cox[i] = rem - coy[i] * ad.OW;
        achunk[i] = chunk * 16;
      }
    }
#pragma unroll
    for (int i = 0; i < GB; ++i) {
      const int row = (wid * GB + i) * RPD + srow;
      const int chunk = spos ^ swz(row);
      int n = tn * BN + row;
      if (n >= N) n = N - 1;
      woff[i] = (unsigned)((long)n * K * ESZ) + chunk * 16;
    }
    p_tap = 0;
    p_c = 0;
  };

  // LDS-DMA ops of one chunk are issued in slices spread between the MFMA groups of the previous chunk (a burst of 8
  // global_load_lds costs as many issue cycles as the chunk's 16 MFMAs).  `prep` resolves the per-lane source pointers
  // of the A operand once per chunk (conv: tap / channel decomposition + padding test), `issue` only launches DMAs.
  bool dma_on = true;
  auto prep = [&](int ch) {
    if (MODE == 0) {
      a_soff = ch * ROWB;   // byte offset along K
    } else {
      const int cin = ad.C0 + ad.C1;
      if (p_c == 0 || p_c == ad.C0) {
        const int ky = p_tap / 3, kx = p_tap - ky * 3;
        const int vh = ad.up == 1 ? ad.IH * 2 : ad.IH, vw = ad.up == 1 ? ad.IW * 2 : ad.IW;
        const bool second = p_c >= ad.C0 && ad.C1 > 0;
        const long cpp = second ? ad.C1 : ad.C0;
        a_second = second;
        a_soff = 0;
#pragma unroll
        for (int i = 0; i < GA; ++i) {
          const int iy = coy[i] * ad.stride + ky - ad.pad, ix = cox[i] * ad.stride + kx - ad.pad;
          const bool ok = iy >= 0 && iy < vh && ix >= 0 && ix < vw;
          const int sy = ad.up == 1 ? iy >> 1 : iy, sx = ad.up == 1 ? ix >> 1 : ix;
          const unsigned off = (unsigned)((((long)cn[i] * ad.IH + sy) * ad.IW + sx) * cpp * ESZ) + achunk[i];
          aoff[i] = ok ? off : DMA_POISON;   // padding: an out-of-range offset reads as zero
        }
      } else {
        a_soff += ROWB;
      }
      p_c += BK;
      if (p_c == cin) { p_c = 0; ++p_tap; }
    }
  };
  auto issue = [&](int stage, int ch, int part, int nparts) {  // chunk `ch` of the DMA-side tile -> LDS stage `stage`
    if (V_NODMA && !dma_on) return;
    char* st = smem + stage * STAGE_BYTES;
    const __amdgpu_buffer_rsrc_t rA = (MODE == 1 && a_second) ? rA1 : rA0;
#pragma unroll
    for (int i = 0; i < GA; ++i)
      if (i % nparts == part) blds16(rA, aoff[i], a_soff, st + (wid * GA + i) * 1024);
#pragma unroll
    for (int i = 0; i < GB; ++i)
      if (i % nparts == part) blds16(rW, woff[i], ch * ROWB, st + A_BYTES + (wid * GB + i) * 1024);
  };

  // fragment read addressing: row r of the tile, 16-byte chunk c  ->  r * 128 + ((c ^ ((r >> 1) & 7)) * 16)
  const int arow = wm * (BM / WM) + lr, brow = wn * (BN / WN) + lr;   // + 32 * tile index (keeps (row>>1)&7 pattern)

  // ---- epilogue geometry.  The MFMAs run with the operands swapped (D = W_frag x A_frag), so a lane holds ONE output row
  // m = lane & 31 and, per 32 x 32 accumulator tile, sixteen columns n = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5).
  // Two v_permlane32_swap per register pair turn that into two runs of 8 consecutive columns per lane ("octets": lane
  // half h owns columns 16 * g2 + 8 * h .. + 7); these are finished in fp32 and packed.  Stored like that, one store
  // instruction would touch 32 rows x 32 bytes, and the texture path spends ~5 cycles per 128-byte line touched
  // (measured with TA_BUSY: 166 cycles per 1 KiB store instruction against 19 for an LDS-DMA load) -- the store time of the
  // short-K GEMMs added to their load time.  So the S = 2 TN octets of a row block are transposed across the DPP quad
  // (quad_transpose): store s of lane (r, h) then covers row S * (lane >> 2 or 1) + s, columns of octet slot r, and one
  // instruction writes 64 / S rows of S * 32 contiguous bytes (full 128-byte lines for a 64-column wave tile).  No LDS
  // round trip, no barrier: waves drift into the next tile's chunks independently.
  static_assert(TN <= 6, "epilogue: column groups of two 32-column tiles, at most three of them");
  constexpr bool RES_PF = ESZ == 2 && TM * TN <= 4 && TN <= 2;   // wide wave tiles have no registers to spare for it
  // POST = the FULL epilogue: row scale / alpha / post-scale bias, SiLU / ReLU / quick-GELU, and the element-wise path for
  // ragged or unaligned problems.  The common instantiations (POST = false) carry bias, per-batch bias, GEGLU and residual on
  // the vectorised path only: every extra epilogue feature costs the hot kernels registers (the post-scale bias alone: 10%).
  // EPI: 0 common, 1 + row scale / alpha / post-scale bias (vectorised path only: MM-HAA's merged branch GEMMs), 2 FULL.
  constexpr bool POST_OK = EPI >= 1, FULL_OK = EPI >= 2;
  // ---- the chunk stream of this workgroup: tiles vt = blockIdx.x + k * gridDim.x, nchunks chunks each ----
  const int nchunks = K / BK;
  const int G = gridDim.x;
  const int my_tiles = (nwg - (int)blockIdx.x + G - 1) / G;
  const int total = my_tiles * nchunks;
  int vt_i = blockIdx.x, ich = 0, gi = 0, sl = 0;   // DMA side: tile, chunk in tile, chunks issued so far, stage to fill next
  int gc = 0, sc = 0;                               // MFMA side: chunks consumed so far, stage to read next
  {
    int tm, tn;
    decode(vt_i, tm, tn);
    setup(tm, tn);
  }
  auto advance_dma = [&]() {   // bookkeeping after chunk (vt_i, ich) has been issued
    ++gi;
    if (V_NODMA && gi >= NSTAGE) dma_on = false;
    sl = sl + 1 == NSTAGE ? 0 : sl + 1;
    if (++ich == nchunks) {
      ich = 0;
      vt_i += G;
      if (vt_i < nwg) {
        int tm, tn;
        decode(vt_i, tm, tn);
        setup(tm, tn);
      }
    }
  };
  // LATE (see the main loop): the ring runs one chunk fuller, the stage of chunk c is refilled inside chunk c's own work
  constexpr bool LATE = V_LATE && (BN == 320 || (BM == 256 && BN == 256) || V_LATE_ALL) && KS >= 2 && KS % 2 == 0;
#pragma unroll
  for (int s = 0; s < NSTAGE - (LATE ? 0 : 1); ++s)
    if (gi < total) { prep(ich); issue(sl, ich, 0, 1); advance_dma(); }
  // at most `younger` whole chunks of LDS-DMA (GA + GB ops per wave each) may stay in flight; vmcnt retires in order, so
  // allowing fewer than are really younger only waits longer
  auto wait_chunks = [&](int younger) {
    if (younger <= 0) wait_vmcnt<0>();
    else if (younger == 1) wait_vmcnt<(GA + GB)>();
    else if (younger == 2 || NSTAGE <= 3) wait_vmcnt<2 * (GA + GB)>();
    else wait_vmcnt<3 * (GA + GB)>();
  };

  if (LATE) {
    wait_chunks(gi - 1);
    __builtin_amdgcn_s_barrier();
  }
  for (int vt = blockIdx.x; vt < nwg; vt += G) {
    int tm, tn;
    decode(vt, tm, tn);
    const int row0 = tm * BM + wm * (BM / WM), col0 = tn * BN + wn * (BN / WN);

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) acc[i][j] = (f32x16)(0.f);

    // fragments of one K-step: lane (row lr [+32 i], k-half lh) reads its 8 consecutive k values from stage `st`
    auto load_frags = [&](const char* st, int ks, Frag<T>* pa, Frag<T>* pb) {
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        const int r = arow + 32 * i;
        const int sw = swz(r);
        if (ESZ == 2) {
          frag_load(pa[i], reinterpret_cast<const T*>(st + r * ROWB + (((2 * ks + lh) ^ sw) << 4)));
        } else {
          const f32x4 lo = *reinterpret_cast<const f32x4*>(st + r * ROWB + (((4 * ks + 2 * lh) ^ sw) << 4));
          const f32x4 hi = *reinterpret_cast<const f32x4*>(st + r * ROWB + (((4 * ks + 2 * lh + 1) ^ sw) << 4));
#pragma unroll
          for (int j = 0; j < 4; ++j) { pa[i].set(j, lo[j]); pa[i].set(4 + j, hi[j]); }
        }
      }
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int r = brow + 32 * j;
        const int sw = swz(r);
        const char* sb = st + A_BYTES;
        if (ESZ == 2) {
          frag_load(pb[j], reinterpret_cast<const T*>(sb + r * ROWB + (((2 * ks + lh) ^ sw) << 4)));
        } else {
          const f32x4 lo = *reinterpret_cast<const f32x4*>(sb + r * ROWB + (((4 * ks + 2 * lh) ^ sw) << 4));
          const f32x4 hi = *reinterpret_cast<const f32x4*>(sb + r * ROWB + (((4 * ks + 2 * lh + 1) ^ sw) << 4));
#pragma unroll
          for (int jj = 0; jj < 4; ++jj) { pb[j].set(jj, lo[jj]); pb[j].set(4 + jj, hi[jj]); }
        }
      }
    };
    Frag<T> fa[2][TM], fb[2][TN];
    // LATE: the tile's first chunk was waited for and published by the previous chunk's barrier (or the one in front of
    // the tile loop), before this tile's epilogue-operand loads and the previous tile's stores entered the vmcnt queue
    if (LATE) load_frags(smem + sc * STAGE_BYTES, 0, fa[0], fb[0]);

    // ---- epilogue operands requested before the main loop, so their latency hides under it:
    //  * residual (bf16 fast path): the lane's 16-byte vectors in the octet layout of the epilogue (measured, earlier
    //    layout: -35% on the L0 N = K = 320 projections);
    //  * bias[n] + bias2[batch row][n] of the lane's W row n = col0 + 32 j + (lane & 31), for the (at most two) bias2
    //    rows the tile touches.  They enter the accumulators as ONE more K-step after the main loop (see below).
    // vmcnt completes in order, so these older ops only make the counted waits of the ring conservative.
    const bool fast = ep.fast != 0;
    const bool geglu = ep.act == 1;
    const bool res_pf = RES_PF && ep.residual != nullptr && fast;
    // Store-layout coordinates of this lane (non-GEGLU: S = 2 TN slots per row block; see "epilogue geometry"):
    // vector s of row block i is row 32 i + S * (lr / S) + s, columns 32 * (r >> 1) + 16 * (r & 1) + 8 h with r = lr % S
    // (for S = 2: 16 r + 8 h).
    constexpr int SN = TN <= 2 ? 2 * TN : 2;   // residual prefetch exists for narrow wave tiles only (one column group)
    const int sr = lr & (SN - 1), srow = lr & ~(SN - 1);
    const int scol = (SN == 4 ? 32 * (sr >> 1) + 16 * (sr & 1) : 16 * sr) + 8 * lh;
    u32x4 rres[RES_PF ? TM : 1][RES_PF ? SN : 1];
    if (res_pf) {
      const T* resb = reinterpret_cast<const T*>(ep.residual) + (long)bz * ep.bsr;
#pragma unroll
      for (int i = 0; i < (RES_PF ? TM : 1); ++i)
#pragma unroll
        for (int sidx = 0; sidx < (RES_PF ? SN : 1); ++sidx) {
          const int m = row0 + 32 * i + srow + sidx, n = col0 + scol;
          rres[i][sidx] = (m < M && n < N && !geglu) ? *reinterpret_cast<const u32x4*>(resb + (long)m * ep.ldr + n)
                                                     : (u32x4)(0u);
        }
    }
    // bias2 rows of this tile: b2r0 (rows m with m / bias2_rows == b2r0) and b2r0 + 1; the host guarantees
    // bias2_rows >= BM on the fast path, so a tile touches at most two.
    const int b2div = ep.bias2 ? ep.bias2_rows : 0x7fffffff;
    const int b2r0 = (tm * BM) / b2div;
    int mlast = tm * BM + BM - 1;
    if (mlast >= M) mlast = M - 1;
    const bool b2two = mlast / b2div > b2r0;
    float bsum[TN][2];
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int n = col0 + 32 * j + lr;
      const bool ok = fast && n < N;
      const float b = (ok && ep.bias) ? ep.bias[n] : 0.f;
      bsum[j][0] = b + ((ok && ep.bias2) ? ep.bias2[(long)b2r0 * N + n] : 0.f);
      bsum[j][1] = b + ((ok && ep.bias2 && b2two) ? ep.bias2[(long)(b2r0 + 1) * N + n] : 0.f);
    }

    if (LATE) {
      // One barrier per chunk, placed INSIDE the chunk's MFMA work (after K-step KS-2 has been queued) instead of in
      // front of it: by then every read of this chunk has returned, so the barrier both frees its stage for the next
      // LDS-DMA and publishes the next chunk, whose first fragments are then read under the last K-step's MFMAs.  The
      // matrix pipe no longer idles through barrier skew + LDS read latency at every chunk start.
      // The refill of the freed stage is issued in KS slices (V_ILV): one right after the barrier, the others behind the
      // following MFMA groups (across the chunk boundary), so no wave pays a burst of LDS-DMA issue slots at once.
      constexpr bool ILVL = V_ILV;
      bool pendf = false;
      int pst = 0, pch = 0;
      for (int ch = 0; ch < nchunks; ++ch, ++gc) {
        const bool last = ch == nchunks - 1;
        const char* st = smem + sc * STAGE_BYTES;
        const int nsc = sc + 1 == NSTAGE ? 0 : sc + 1;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
          if (ks + 1 < KS) load_frags(st, ks + 1, fa[(ks + 1) & 1], fb[(ks + 1) & 1]);
          __builtin_amdgcn_s_setprio(1);
#pragma unroll
          for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) mma32(acc[i][j], fb[ks & 1][j], fa[ks & 1][i]);   // D[n][m]
          __builtin_amdgcn_s_setprio(0);
          if (ks == KS - 2) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // this wave's reads of the chunk have all returned
            if (gc + 1 < total) wait_chunks(gi - gc - 2);         // this wave's share of chunk gc + 1 has landed
            __builtin_amdgcn_s_barrier();
            if (!last) load_frags(smem + nsc * STAGE_BYTES, 0, fa[0], fb[0]);
            if (gi < total) {                                     // refill the stage just freed
              prep(ich);
              if (ILVL) {
                pendf = true; pst = sc; pch = ich;
                issue(pst, pch, 0, KS);
                if (KS == 1) { advance_dma(); pendf = false; }
              } else {
                issue(sc, ich, 0, 1);
                advance_dma();
              }
            }
          } else if (ILVL && pendf) {
            constexpr int KSm = KS > 0 ? KS : 1;
            const int part = (ks + 2) % KSm;
            issue(pst, pch, part, KS);
            if (part == KS - 1) { advance_dma(); pendf = false; }
          }
        }
        sc = nsc;
      }
      if (ILVL && pendf) {   // tile end: slices 0 (barrier) and 1 (last K-step) are out, the rest goes now
#pragma unroll
        for (int part = 2; part < KS; ++part) issue(pst, pch, part, KS);
        advance_dma();
        pendf = false;
      }
    } else {
    for (int ch = 0; ch < nchunks; ++ch, ++gc) {
      // chunk gc must have landed; up to NSTAGE-2 younger chunks may stay in flight (GA + GB LDS-DMA ops per wave each)
      wait_chunks(gi - gc - 1);
      __builtin_amdgcn_s_barrier();
      // Spreading the DMA issue between MFMA groups pays for the dense 8-wave tiles (+7% at 8192^3); for the conv gather
      // and the 4-wave tiles the burst right after the barrier measured faster.
      constexpr bool ILV = V_ILV && (NW == 8 && MODE == 0);
      const bool more = gi < total;
      const int st_i = sl, ch_i = ich;
      if (more) {
        prep(ich);
        if (!ILV) issue(st_i, ch_i, 0, 1);
      }

      const char* st = smem + sc * STAGE_BYTES;
      // fragments of K-step ks+1 are read from LDS while the MFMAs of K-step ks run (register double buffering)
      if (V_FRAGDB) load_frags(st, 0, fa[0], fb[0]);
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        if (V_FRAGDB) {
          if (ks + 1 < KS) load_frags(st, ks + 1, fa[(ks + 1) & 1], fb[(ks + 1) & 1]);
          __builtin_amdgcn_s_setprio(1);
        } else {
          load_frags(st, ks, fa[ks & 1], fb[ks & 1]);
        }
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j) mma32(acc[i][j], fb[ks & 1][j], fa[ks & 1][i]);   // D[n][m]
        if (V_FRAGDB) __builtin_amdgcn_s_setprio(0);
        if (ILV && more) issue(st_i, ch_i, ks, KS);   // this slice's LDS-DMA issues under the MFMAs just queued
      }
      if (more) advance_dma();
      sc = sc + 1 == NSTAGE ? 0 : sc + 1;
    }
    }

    // ---- bias as one more K-step (fast path):  D[n][m] += sum_c bsum[c][n] * sel[c][m], sel[c][m] = (row m belongs to
    // bias2 row b2r0 + c).  Each value is split into a storage-type head and tail (k = 2c, 2c + 1) so the bf16
    // instantiation adds the bias to ~2^-17 relative; for fp32 the tail is zero and the product is exact.
    if (fast && (ep.bias || ep.bias2)) {
      Frag<T> fbias[TN], fsel[TM];
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        fbias[j].zero();
        if (lh == 0) {
#pragma unroll
          for (int c = 0; c < 2; ++c) {
            const float hi = Elem<T>::cvt(bsum[j][c]);
            fbias[j].set(2 * c, hi);
            fbias[j].set(2 * c + 1, bsum[j][c] - hi);
          }
        }
      }
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        fsel[i].zero();
        int m = row0 + 32 * i + lr;
        if (m >= M) m = M - 1;
        const int c = m / b2div - b2r0;
        if (lh == 0) {
          fsel[i].set(0, c == 0 ? 1.f : 0.f);
          fsel[i].set(1, c == 0 ? 1.f : 0.f);
          fsel[i].set(2, c == 1 ? 1.f : 0.f);
          fsel[i].set(3, c == 1 ? 1.f : 0.f);
        }
      }
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) mma32(acc[i][j], fbias[j], fsel[i]);
    }

    // ---- epilogue from registers: lane = output row m, accumulator register = column (see "epilogue geometry") ----
    T* out = reinterpret_cast<T*>(ep.out) + (long)bz * ep.bso;
    const T* res = ep.residual ? reinterpret_cast<const T*>(ep.residual) + (long)bz * ep.bsr : nullptr;
    if (fast) {
      constexpr int OV = ESZ == 2 ? 1 : 2;       // 16-byte vectors per octet as stored
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        int m = row0 + 32 * i + lr;              // the row this lane holds in the accumulator layout
        asm volatile("" : "+v"(m));              // keeps the address arithmetic below out of the main loop's live ranges
        const bool mok = m < M;
        const bool scaled = POST_OK && (ep.row_scale != nullptr || ep.alpha != 1.f);   // uniform
        const float rs = POST_OK ? (ep.row_scale && mok ? ep.row_scale[m] : 1.f) * ep.alpha : 1.f;
        T* ob = out + (long)(row0 + 32 * i) * ep.ldo;
        // One column group = GN (1 or 2) adjacent 32-column accumulator tiles starting at tile JG: SNg = 2 GN octet slots
        // per row, transposed across the DPP quad into the store layout (see "epilogue geometry").
        auto group = [&](auto JGc, auto GNc) {
          constexpr int JG = decltype(JGc)::value, GN = decltype(GNc)::value, SNg = 2 * GN;
          const int srg = lr & (SNg - 1), srowg = lr & ~(SNg - 1);
          const int scolg = 32 * JG + (SNg == 4 ? 32 * (srg >> 1) + 16 * (srg & 1) : 16 * srg) + 8 * lh;
          // residual of this row block: prefetched (or loaded here) in the store layout, transposed back to "lane = row"
          u32x4 rv[OV][SNg];
          if (res) {
#pragma unroll
            for (int sidx = 0; sidx < SNg; ++sidx) {
              const int ms = row0 + 32 * i + srowg + sidx, ns = col0 + scolg;
              const bool inb = ms < M && ns < N;
              if (ESZ == 2) {
                rv[0][sidx] = res_pf ? rres[RES_PF ? i : 0][RES_PF ? sidx : 0]
                                     : (inb ? *reinterpret_cast<const u32x4*>(res + (long)ms * ep.ldr + ns) : (u32x4)(0u));
              } else {
                const u32x4* rp = reinterpret_cast<const u32x4*>(res + (long)ms * ep.ldr + ns);
                rv[0][sidx] = inb ? rp[0] : (u32x4)(0u);
                rv[OV - 1][sidx] = inb ? rp[1] : (u32x4)(0u);
              }
            }
#pragma unroll
            for (int q = 0; q < OV; ++q) quad_transpose<SNg>(rv[q], lane);
          }
          u32x4 pk[OV][SNg];                       // finished octets, slot k = 2 jj + g2
#pragma unroll
          for (int jj = 0; jj < GN; ++jj) {
            constexpr int JLAST = TN - 1;
            const int j = JG + jj;
            if (geglu && (j & 1)) continue;          // gate tiles are consumed with their h tile
            f32x16 v = acc[i][j];
            if (geglu) {
              // packed weights: tile j = 32 h channels, tile j + 1 = their 32 gates, same lane and register
              const f32x16 gt = acc[i][j < JLAST ? j + 1 : j];
#pragma unroll
              for (int r = 0; r < 16; r += 2) {
                const f32x2 gl = gelu_erf_f2((f32x2){gt[r], gt[r + 1]});
                v[r] *= gl[0];
                v[r + 1] *= gl[1];
              }
            } else {
              if (FULL_OK && ep.act == 2) {
#pragma unroll
                for (int r = 0; r < 16; ++r) v[r] = silu_f(v[r]);
              } else if (FULL_OK && ep.act == 3) {
#pragma unroll
                for (int r = 0; r < 16; ++r) v[r] = fmaxf(v[r], 0.f);
              } else if (FULL_OK && ep.act == 4) {
#pragma unroll
                for (int r = 0; r < 16; ++r) v[r] = quick_gelu_f(v[r]);
              } else if (FULL_OK && ep.act == 5) {
#pragma unroll
                for (int r = 0; r < 16; ++r) v[r] = gelu_erf_f(v[r]);
              } else if (FULL_OK && ep.act == 6) {
#pragma unroll
                for (int r = 0; r < 16; ++r) v[r] = mish_f(v[r]);
              }
              if (scaled) v *= rs;
            }
#pragma unroll
            for (int g2 = 0; g2 < 2; ++g2) {
              const int k = 2 * jj + g2;
              float o8[8];
#pragma unroll
              for (int e = 0; e < 4; ++e) {
                // a: n = 16 g2 + e (+4 in the upper lane half), b: n = 16 g2 + 8 + e (+4);  a.hi <-> b.lo
                const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(v[8 * g2 + e]),
                                                                 __float_as_uint(v[8 * g2 + 4 + e]), false, false);
                o8[e] = __uint_as_float(sw[0]);
                o8[4 + e] = __uint_as_float(sw[1]);
              }
              if (POST_OK && ep.bias_post) {   // uniform; the lane's 8 consecutive columns of this octet
                const float* pp = ep.bias_post + col0 + 32 * j + 16 * g2 + 8 * lh;
                const bool cok = col0 + 32 * j + 16 * g2 + 8 * lh < N;
                const f32x4 p0 = cok ? *reinterpret_cast<const f32x4*>(pp) : (f32x4)(0.f);
                const f32x4 p1 = cok ? *reinterpret_cast<const f32x4*>(pp + 4) : (f32x4)(0.f);
#pragma unroll
                for (int e = 0; e < 4; ++e) { o8[e] += p0[e]; o8[4 + e] += p1[e]; }
              }
              if (ESZ == 2) {
                if (res) {
                  union { u32x4 u; bf16_t e[8]; } r8;
                  r8.u = rv[0][k];
#pragma unroll
                  for (int e = 0; e < 8; ++e) o8[e] += bf16_to_f32(r8.e[e]);
                }
                union { bf16_t e[8]; u32x4 u; } p8;
#pragma unroll
                for (int e = 0; e < 8; ++e) p8.e[e] = f32_to_bf16(o8[e]);
                pk[0][k] = p8.u;
              } else {
                union { f32x4 f; u32x4 u; } c0, c1;
                c0.f = (f32x4){o8[0], o8[1], o8[2], o8[3]};
                c1.f = (f32x4){o8[4], o8[5], o8[6], o8[7]};
                if (res) {
                  union { u32x4 u; f32x4 f; } r0, r1;
                  r0.u = rv[0][k];
                  r1.u = rv[OV - 1][k];
                  c0.f += r0.f;
                  c1.f += r1.f;
                }
                pk[0][k] = c0.u;
                pk[OV - 1][k] = c1.u;
              }
            }
          }
          // ---- to the store layout and out
          if (!geglu) {
#pragma unroll
            for (int q = 0; q < OV; ++q) quad_transpose<SNg>(pk[q], lane);
#pragma unroll
            for (int sidx = 0; sidx < SNg; ++sidx) {
              const int ms = row0 + 32 * i + srowg + sidx, ns = col0 + scolg;
              if (ms >= M || ns >= N) continue;
              u32x4* op = reinterpret_cast<u32x4*>(ob + (long)(srowg + sidx) * ep.ldo + ns);
#pragma unroll
              for (int q = 0; q < OV; ++q) op[q] = pk[q][sidx];
            }
          } else if (GN == 2) {
            // GEGLU: the (h, gate) tile pair yields 32 output columns = 2 octets per row -> one 2 x 2 transpose
            const int gr = lr & 1, grow = lr & ~1;
#pragma unroll
            for (int q = 0; q < OV; ++q) {
              u32x4 t2[2] = {pk[q][0], pk[q][1]};
              quad_transpose<2>(t2, lane);
              pk[q][0] = t2[0];
              pk[q][1] = t2[1];
            }
#pragma unroll
            for (int sidx = 0; sidx < 2; ++sidx) {
              const int ms = row0 + 32 * i + grow + sidx;
              const int ns = col0 + 32 * JG + 16 * gr + 8 * lh;          // column in the accumulator's N space
              if (ms >= M || ns >= N) continue;
              const long ocol = (long)((col0 + 32 * JG) >> 1) + 16 * gr + 8 * lh;
              u32x4* op = reinterpret_cast<u32x4*>(ob + (long)(grow + sidx) * ep.ldo + ocol);
#pragma unroll
              for (int q = 0; q < OV; ++q) op[q] = pk[q][sidx];
            }
          }
        };
        using std::integral_constant;
        group(integral_constant<int, 0>{}, integral_constant<int, (TN >= 2 ? 2 : 1)>{});
        if constexpr (TN > 2) group(integral_constant<int, 2>{}, integral_constant<int, (TN >= 4 ? 2 : 1)>{});
        if constexpr (TN > 4) group(integral_constant<int, 4>{}, integral_constant<int, (TN >= 6 ? 2 : 1)>{});
      }
    } else if (FULL_OK) {
      // generic scalar path (ragged N, unaligned rows, bias2 blocks shorter than a tile): element by element
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        const int m = row0 + 32 * i + lr;
        if (m >= M) continue;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          if (geglu && (j & 1)) continue;
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int nl = acc_row(r, lane);
            const int ncol = col0 + 32 * j + nl;
            if (ncol >= N) continue;
            float x = acc[i][j][r];
            if (ep.bias) x += ep.bias[ncol];
            long ocol = ncol;
            if (geglu) {
              float gte = acc[i][(j + 1) < TN ? j + 1 : j][r];
              if (ep.bias) gte += ep.bias[ncol + 32];
              x *= gelu_erf_f(gte);
              ocol = (long)((col0 + 32 * j) >> 1) + nl;
            } else {
              if (ep.bias2) x += ep.bias2[(long)(m / ep.bias2_rows) * N + ncol];
              if (ep.act == 2) x = silu_f(x);
              if (ep.act == 3) x = fmaxf(x, 0.f);
              if (ep.act == 4) x = quick_gelu_f(x);
              if (ep.act == 5) x = gelu_erf_f(x);
              if (ep.act == 6) x = mish_f(x);
              x *= (ep.row_scale ? ep.row_scale[m] : 1.f) * ep.alpha;
              if (POST_OK && ep.bias_post) x += ep.bias_post[ncol];
            }
            if (res) x += Elem<T>::ld(res + (long)m * ep.ldr + ocol);
            Elem<T>::st(out + (long)m * ep.ldo + ocol, x);
          }
        }
      }
    }
  }
}

template <typename T, int MODE, int BM, int BN, int WM, int WN, int NSTAGE, int ROWB = 128, int EPI = 0>
int launch_cfg(const ADesc& ad, const void* W, long bsw, const Epi& ep, int M, int N, int K, int batch, hipStream_t s) {
  const int tiles_m = (M + BM - 1) / BM, tiles_n = (N + BN - 1) / BN;
  const size_t lds = (size_t)NSTAGE * (BM + BN) * ROWB;
  auto kern = gemm_kernel<T, MODE, BM, BN, WM, WN, NSTAGE, ROWB, EPI>;
  static int resident = 0;   // workgroups of this instantiation the whole device holds at once
  if (!resident) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) !=
        hipSuccess) {
      mmgt_set_error("gemm: cannot reserve %zu bytes of LDS", lds);
      return 2;
    }
    int per_cu = 0, dev = 0;
    hipDeviceProp_t prop;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kern, WM * WN * 64, lds) != hipSuccess || per_cu < 1 ||
        hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) {
      mmgt_set_error("gemm: occupancy query failed");
      return 2;
    }
    resident = per_cu * prop.multiProcessorCount;
  }
  // persistent grid: as many workgroups as stay resident (a multiple of 8 keeps id & 7 = XCD), each walks its tiles
  long gx = (resident + batch - 1) / batch;
  gx = (gx + 7) / 8 * 8;
  if (gx > (long)tiles_m * tiles_n) gx = (long)tiles_m * tiles_n;
  dim3 grid((unsigned)gx, 1, batch);
  hipLaunchKernelGGL(kern, grid, dim3(WM * WN * 64), lds, s, ad, reinterpret_cast<const char*>(W), bsw, ep, M, N, K, tiles_m,
                     tiles_n);
  MMGT_LAUNCH_CHECK();
  return 0;
}

int g_splitk = 1;     // mmgt_tune("splitk", 0 / 1): A/B switch of the split-K path
int g_tailsplit = 1;  // mmgt_tune("tailsplit", 0 / 1): A/B switch of the tail split (below)
int g_gemm_cfg = 0;   // 0 = heuristic; 1, 3, 6, 9, 12, 16, 17, 19, 20 force a tile configuration (mmgt_tune("gemm_cfg", v), benchmarking only)
int g_bm192 = 1;      // mmgt_tune("bm192", 0 / 1): A/B switch of the 192-row gemm16 tile

}  // namespace
// gemm16.hip: the bf16 256x256 8-phase core on 16x16x32 MFMAs (cfg 16)
int mmgt_gemm16_launch(int mode, int bn, const void* ad, const void* W, long bsw, const void* ep, int M, int N, int K,
                       int batch, void* stream);
int mmgt_gemm16_splitk(int mode, int bn, const void* ad, const void* W, const void* ep, int M, int N, int K, int S, void* stream, int m0);
namespace {

template <typename T, int MODE>
int launch(const ADesc& ad, const void* W, long bsw, const Epi& ep, int M, int N, int K, int batch, hipStream_t s) {
  const bool geglu = ep.act == 1;
  int cfg = g_gemm_cfg;
  // Split-K (gemm16.hip): a long reduction on a grid of at most half a tile per CU -- the 8x8-level convs (3072 rows x 1280 columns,
  // K = 11 520 / 23 040) and ff2 of that level (K = 5120).  S slices with >= 16 chunks of 64 each, S x tiles <= 256.
  if ((cfg == 0 || cfg == 18) && g_splitk && std::is_same<T, bf16_t>::value && batch == 1 && ep.act == 0 && ep.fast && !ep.row_scale &&
      ep.alpha == 1.f && !ep.bias_post && N % 8 == 0 && K >= 2560 && (N % 256 == 0 || N % 320 == 0) &&
      (((uintptr_t)ep.bias | (uintptr_t)ep.bias2) & 15) == 0) {
    const int bn = N % 256 == 0 ? 256 : 320;
    const long tiles = (long)((M + 255) / 256) * (N / bn);
    const int nch = K / 64;
    int S = 0;
    // (measured, tools/ab_cfg.py SET=ctx12 / step: at <= 64 tiles slices of >= 16 chunks pay -- 8x8 convs 126 -> 71 us, 248 -> 101 us, ff2
    // 59 -> 47 us at 12 frames --; between 65 and 128 tiles only long slices do: the 16x16 convs of a 12-frame window -5 %, their ff2 +10 %)
    for (int c = 8; c >= 2; --c)
      if (tiles * c <= 256 && nch % c == 0 && nch / c >= (tiles <= 64 ? 16 : 64)) { S = c; break; }
    // (dense shapes between 33 and 64 tiles -- ff2 of the 8x8 level at 24 frames, 3072 x 1280 x 5120 -- run 8 % faster on the 128 x 64 tile
    //  than split four ways: 58.1 against 63.0 us, profiles/r5/ab_cfg_deep_r5.txt; the convs of that level keep the split: 126 -> 71 us)
    if (S >= 2 && tiles <= 128 && (MODE == 1 || tiles <= 32 || tiles > 64)) return mmgt_gemm16_splitk(MODE, bn, &ad, W, &ep, M, N, K, S, s, 0);
  }
  if (cfg == 18) cfg = 0;
  if (cfg == 0) {
    // Measured on MI355X with tools/ab_gemm.py / tools/bench_kernels.py (one process, one device; re-swept after the
    // LDS-DMA moved to buffer addressing, which made 128x128 the better small tile for every dense shape):
    //   cfg 6  (256x128, 8 waves, 3-deep ring)    long reductions, GEGLU, and the widest L0 projections;
    //   cfg 1  (128x128, 2 workgroups / CU)       every other dense shape, and the conv gather on the small levels;
    //   cfg 12 (128x320, 8 waves, barrier inside the chunk)   conv outputs whose width is a multiple of 320 on the two
    //          large levels: the im2col gather is staged once per 320 columns instead of once per 128 (-5..-20%);
    //   cfg 3  (128x64, 3 workgroups / CU)        conv grids that would not fill the chip (the 8x8 level).
    const long tiles128 = (long)((M + 127) / 128) * ((N + 127) / 128) * batch;
    const long tiles256 = (long)((M + 255) / 256) * ((N + 127) / 128) * batch;
    //   cfg 9  (256x256, 8 waves, 2-deep ring, barrier inside the chunk)   long reductions whose grid still gives ~one tile
    //          per CU: the 16x16-level convs (-6%); dense shapes: below.
    const long tiles256sq = (long)((M + 255) / 256) * ((N + 255) / 256) * batch;
    const bool big_ok = N % 256 == 0 && K >= 2560 && tiles256sq >= 192 && tiles256sq <= 512;
    // dense: the 256x256 tile wherever N fills its columns to within 7% (N = 960, 1280, 1920, 2560, 3840, 5120, 10240) and
    // the grid still covers the chip -- with the slim common epilogue it no longer spills: GEGLU ff1 -11..-13%, q/k/v
    // projections of the motion modules -13%, the N = 1280 family -5..-10%.
    // (and M: the batched W.X^T projections have M = 320 / 640 rows per problem, 62 / 83 % of two / three 256-row tiles: -13..-17 % on the 128x128 tile)
    const bool sq_ok = ((N + 255) / 256) * 256l * 100 <= (long)N * 107 && ((M + 255) / 256) * 256l * 100 <= (long)M * 115 && tiles256sq >= 192;
    // cfg 16 (gemm16.hip: 256x256, 16x16x32 MFMAs, two wave groups in ping-pong; bf16 only) replaces cfg 9 wherever that ran and
    // takes the convs whose width fills 256-column tiles: measured (tools/ab_cfg.py, one process) -17..-19% on the 16x16-level
    // convs and the 16 -> 32 up-conv, -7..-10% on the long-K / wide dense shapes, -7% on the 32 -> 64 up-conv (N = 640: three
    // tiles with 17% padding still beat the 128x320 tile); K = 320 GEGLU too since the two wave groups run their epilogues side by
    // side (-9%); it loses on N = 320 / 640.
    // cfg 17 = the same core with a 256x320 tile (every width of the UNet is a multiple of 320): the L0 / L1 convs and the
    // 32 -> 64 / 16 -> 32 up-convs (-2..-22%), the dense N = 320 / 640 / 960 shapes of the 64x64 level (-6..-27%) and the long-K,
    // residual or N = 1280 shapes of the 32x32 level (-5..-8%).
    const bool b16 = std::is_same<T, bf16_t>::value;
    if (MODE == 1) {
      if (b16 && N % 320 == 0 && (M >= 49152 || (M >= 24576 && N >= 640))) cfg = 17;   // (12-frame windows: the 32x32-level convs, -7..-9 % against 128x128)
      else if (b16 && N % 256 == 0 && tiles256sq >= 192) cfg = 16;
      else cfg = (N % 320 == 0 && M >= 49152) ? 12 : big_ok ? 9 : (tiles128 < 512 || N <= 64) ? 3 : 1;   // (N <= 64: conv_out's 3 / 4 channels padded
                                                                                                         //  to 64 -- VAE 8 x 512^2 x 128 -> 64: 651 -> 503 us on the 128x64 tile)
    } else if (b16 && !geglu && N % 320 == 0 && N <= 960 && M >= 131072) cfg = 17;
    else if (b16 && !geglu && N % 320 == 0 && N <= 1280 && M >= 49152 && (K >= 1280 || ep.residual || N == 1280)) cfg = 17;   // 32x32 level
    else if (b16 && !geglu && N == 640 && M >= 24576 && K >= 2560 && g_bm192) cfg = 17;   // (12-frame windows: ff2 of the 32x32 level on the 192-row tile, 90.4 -> 82.3 us)
    else if (sq_ok) cfg = b16 ? 16 : 9;
    else if (geglu || K >= 1280) cfg = tiles256 >= 256 ? 6 : (!geglu && tiles128 <= 256) ? 3 : 1;   // (<= one 128x128 tile per CU: 128x64 tiles, -5..-7 % at M = 3072)
    else if (M >= 131072 && N >= 640) cfg = 6;
    else cfg = 1;
  }
  if (geglu && (cfg == 3 || cfg == 12)) cfg = 1;   // GEGLU pairs need 64-column wave tiles
  if (MODE == 1 && ad.up >= 2 && cfg != 16 && cfg != 17) cfg = N % 320 == 0 ? 17 : 16;   // the four-phase upsample conv exists in gemm16.hip only
  // cfg 19 = gemm16's 192 x 320 tile (round 5): taken where 256-row tiles leave the last round of the persistent grid half empty.  With T
  // tiles on 256 CUs a launch runs ceil(T / 256) rounds of one tile time; a 192-row tile takes ~0.88 of a 256-row tile's time (three MFMA row
  // tiles per four W fragment reads instead of four).  Measured (tools/ab_cfg.py SET=bm192, profiles/r5/ab_cfg_bm192_r5.txt): 49 152 x 640
  // (384 tiles = 2 rounds against 512 = 2 x 0.88): K = 640 + residual 66.1 -> 62.5 us, K = 2560 + residual 184.8 -> 170.6, K = 1280 89.9 ->
  // 83.2; conv 32 x 32, 320 -> 640 185 -> 174; 12-frame windows: 98 304 x 320 + residual 50.3 -> 46.2, 24 576 x 640 x 2560 88.4 -> 82.3, convs
  // 175 -> 167 and 190 -> 178.  It loses where the round count does not drop (196 608 rows, N = 1280) and against the tail split of the
  // long-reduction convs (48 x 32 x 32, 640 -> 640: 305 us against 322), which therefore keeps its shapes.
  bool bm192 = cfg == 19;
  if (cfg == 19) cfg = 17;
  if (cfg == 17 && !bm192 && g_gemm_cfg == 0 && g_bm192 && std::is_same<T, bf16_t>::value && batch == 1 && N % 320 == 0) {
    const long tn = N / 320;
    const long r256 = (((long)(M + 255) / 256) * tn + 255) / 256, r192 = (((long)(M + 191) / 192) * tn + 255) / 256;
    bm192 = (double)r192 * 0.88 < (double)r256 * 0.97;
  }
  if (cfg == 16 || cfg == 17) {   // gemm16.hip, 256 / 320 columns: bf16, plain vectorised epilogue only (GEGLU: 256); else fall back
    const bool post = ep.row_scale || ep.alpha != 1.f || ep.bias_post;   // row scale / alpha / post-scale bias: without GEGLU only
    if (std::is_same<T, bf16_t>::value && ep.fast && ep.act <= (cfg == 16 && !post ? 1 : 0) &&
        (((uintptr_t)ep.bias | (uintptr_t)ep.bias2 | (uintptr_t)ep.bias_post) & 15) == 0 && N % 4 == 0) {   // (bias vectors travel by 16-byte DMA)
      const int bn = cfg == 16 ? 256 : bm192 ? 192320 : 320;   // (192320: the 192 x 320 tile)
      // Tail split.  A persistent grid of T tiles runs ceil(T / 256) rounds of one tile per CU; with a long reduction and T = 256 q + r,
      // r <= 128 (the 32x32 level: 49 152 rows x 640 columns = 384 tiles), the last round keeps half the chip idle for a whole tile.  The
      // rows of the r tail tiles run as a second launch with the reduction split in two (2 r <= 256 half-tiles: one round of half the
      // length) + the fixed-order reduce.  Convs only (reductions of 2880 .. 17 280: 32x32 convs 640 -> 640 338 -> 310 us, 1280 -> 640
      // 631 -> 556, 1920 -> 640 923 -> 791): the fp32 partial slabs and the reduce cost ~25 us, which a dense K = 2560 tile does not repay.
      const int bnc = cfg == 16 ? 256 : 320;
      const long tiles_n = (N + bnc - 1) / bnc, tiles_m = (M + 255) / 256, ntile = tiles_m * tiles_n;
      const int nch = K / 64;
      if (g_tailsplit && (!bm192 || g_gemm_cfg == 0) && MODE == 1 && batch == 1 && !ad.ksplit && ep.act == 0 && !post && N % bnc == 0 && N % 8 == 0 && K >= 2560 && nch % 2 == 0 && ntile > 256) {
        const long rows_a = (256 * (ntile / 256) / tiles_n) * 256;                  // whole row tiles that fill ntile / 256 full rounds
        const long tail = (tiles_m - rows_a / 256) * tiles_n;
        const long img = MODE == 1 ? (long)ad.OH * ad.OW : 1;
        if (rows_a > 0 && rows_a < M && tail > 0 && 2 * tail <= 256 && 4 * tail >= 256 && rows_a % img == 0) {
          int rc = mmgt_gemm16_launch(MODE, bnc, &ad, W, bsw, &ep, (int)rows_a, N, K, 1, s);
          if (rc) return rc;
          ADesc adb = ad;
          Epi epb = ep;
          const long esz = sizeof(T);
          if (MODE == 1) {
            const long n0 = rows_a / img;
            adb.src0 = ad.src0 + n0 * ad.IH * ad.IW * ad.C0 * esz;
            if (ad.src1) adb.src1 = ad.src1 + n0 * ad.IH * ad.IW * ad.C1 * esz;
          } else {
            adb.src0 = ad.src0 + rows_a * ad.ld0 * esz;
          }
          if (ep.residual) epb.residual = ep.residual + rows_a * ep.ldr * esz;
          epb.out = ep.out + rows_a * ep.ldo * esz;
          return mmgt_gemm16_splitk(MODE, bnc, &adb, W, &epb, (int)(M - rows_a), N, K, 2, s, (int)rows_a);
        }
      }
      return mmgt_gemm16_launch(MODE, bn, &ad, W, bsw, &ep, M, N, K, batch, s);
    }
    if (MODE == 1 && ad.up >= 2) {
      mmgt_set_error("conv: the four-phase upsample form and the two-source 1 x 1 conv need the gemm16 path (bf16, 16-byte aligned bias, Cout %% 4 == 0)");
      return 1;
    }
    cfg = cfg == 16 ? 9 : geglu ? 1 : 12;   // (GEGLU pairs need 64-column wave tiles: not the 320-column tile)
  }
  if (cfg == 20) {   // gemm16.hip's 256 x 128 tile (round 5): the VAE's 128-wide levels
    if (std::is_same<T, bf16_t>::value && ep.fast && ep.act == 0 && (((uintptr_t)ep.bias | (uintptr_t)ep.bias2 | (uintptr_t)ep.bias_post) & 15) == 0 && N % 4 == 0)
      return mmgt_gemm16_launch(MODE, 128, &ad, W, bsw, &ep, M, N, K, batch, s);
    cfg = 1;
  }
  // anything beyond bias / per-batch bias / GEGLU / residual on the vectorised path runs the FULL instantiation (128x128 tile)
  if (!ep.fast || ep.act >= 2) return launch_cfg<T, MODE, 128, 128, 2, 2, 2, 128, 2>(ad, W, bsw, ep, M, N, K, batch, s);
  if (ep.row_scale || ep.alpha != 1.f || ep.bias_post) {
    if (MODE == 0 && !geglu) return launch_cfg<T, 0, 128, 128, 2, 2, 2, 128, 1>(ad, W, bsw, ep, M, N, K, batch, s);
    return launch_cfg<T, MODE, 128, 128, 2, 2, 2, 128, 2>(ad, W, bsw, ep, M, N, K, batch, s);
  }
#ifdef MMGT_GEMM_AB   // A/B builds (make ab) instantiate only the production tiles: seconds instead of minutes
  switch (cfg) {
    case 3: return launch_cfg<T, MODE, 128, 64, 2, 2, 2>(ad, W, bsw, ep, M, N, K, batch, s);
    case 6: return launch_cfg<T, MODE, 256, 128, 4, 2, 3>(ad, W, bsw, ep, M, N, K, batch, s);
    case 9: return launch_cfg<T, MODE, 256, 256, 2, 4, 2>(ad, W, bsw, ep, M, N, K, batch, s);
    case 12: return launch_cfg<T, MODE, 128, 320, 4, 2, 2>(ad, W, bsw, ep, M, N, K, batch, s);
    default: return launch_cfg<T, MODE, 128, 128, 2, 2, 2>(ad, W, bsw, ep, M, N, K, batch, s);
  }
#else
  // Tiles measured and dropped (tools/ab_gemm.py, one process, one device): 128x128 / 128x64 with a 3-deep ring (round 5 again, with 4-deep
  // rings too, on the M = 1536 / 3072 shapes of the 8x8 level: profiles/r5/ab_cfg_deep_r5.txt -- 128x64 x 3 gains 10 % at M = 1536 only), 64x64,
  // 256x128 with a 2-deep ring, 256x64, 256x256 with a 3- or 4-deep ring of 64-byte rows, 256x320 on 8 waves (accumulators +
  // double-buffered fragments spill inside the main loop; with single-buffered W fragments reloaded right after their last
  // MFMA it still spills 56-139 VGPRs and runs 1179 us at 8192^3 against 965) and 256x256 / 256x320 on 4 waves (one wave per SIMD, 512
  // registers: +4% at 8192^3, -9% on the 16x16 convs, 2-3x slower wherever the spilling epilogue matters).
  switch (cfg) {
    case 1: return launch_cfg<T, MODE, 128, 128, 2, 2, 2>(ad, W, bsw, ep, M, N, K, batch, s);
    case 3: return launch_cfg<T, MODE, 128, 64, 2, 2, 2>(ad, W, bsw, ep, M, N, K, batch, s);
    case 6: return launch_cfg<T, MODE, 256, 128, 4, 2, 3>(ad, W, bsw, ep, M, N, K, batch, s);
    case 9: return launch_cfg<T, MODE, 256, 256, 2, 4, 2>(ad, W, bsw, ep, M, N, K, batch, s);
    case 12: return launch_cfg<T, MODE, 128, 320, 4, 2, 2>(ad, W, bsw, ep, M, N, K, batch, s);
    default: return launch_cfg<T, MODE, 128, 128, 2, 2, 2>(ad, W, bsw, ep, M, N, K, batch, s);
  }
#endif
}

int epi_fast(const Epi& ep, int N, int n_out, int esz) {
  auto al16 = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
  bool ok = N % 8 == 0 && n_out % 8 == 0 && al16(ep.out) && (ep.ldo * esz) % 16 == 0 && (ep.bso * esz) % 16 == 0;
  ok = ok && (ep.bias2 == nullptr || ep.bias2_rows >= 256);   // a tile (<= 256 rows) touches at most two bias2 rows
  if (ep.residual) ok = ok && al16(ep.residual) && (ep.ldr * esz) % 16 == 0 && (ep.bsr * esz) % 16 == 0;
  return ok ? 1 : 0;
}

int check_common(int dtype, int M, int N, int K, int act) {
  MMGT_CHECK(dtype == MMGT_F32 || dtype == MMGT_BF16, "gemm: bad dtype %d", dtype);
  MMGT_CHECK(M > 0 && N > 0 && K > 0, "gemm: empty problem M=%d N=%d K=%d", M, N, K);
  MMGT_CHECK(K % 64 == 0, "gemm: K=%d must be a multiple of 64 (pad channels on the host)", K);
  MMGT_CHECK(act >= 0 && act <= 6, "gemm: bad act %d", act);
  MMGT_CHECK(act != 1 || N % 64 == 0, "gemm: GEGLU needs N %% 64 == 0 (N=%d)", N);
  return 0;
}

}  // namespace

void mmgt_attn_set64(int v);
void mmgt_gn_set_rows(int v);
void mmgt_attn_set_heads_inner(int v);
void mmgt_attn_set_attn80(int v);
void mmgt_attn_set_nomax(int v);
void mmgt_gemm16_set_pb(int v);
void mmgt_gemm16_set_stagger(int v);
void mmgt_gn_set_interleave(int v);
void mmgt_gn_set_lpr0(int v);
void mmgt_gn_set_narrow(int v);
void mmgt_gn_set_slab(int v);
void mmgt_ffn_set_dbg(int v);
void mmgt_rowgemm_set_dbg(int v);
void mmgt_tleg_set_abl(int v);
void mmgt_gnconv_set_abl(int v);
void mmgt_rconv_set_abl(int v);
void mmgt_rconv_set_stagger(int v);
void mmgt_rconv_set_cb(int v);
// Switches of the HOST side of the operator (mmgt_amd/unet3d.py, pipeline.py, smga.py read them through mmgt_tune_get): they live here,
// beside the kernel knobs, so that ONE state -- this table -- describes what a run executed (MMGT_TUNE="twin_attention=0,splitk=0").
namespace {
struct HostSwitch { const char* key; int value; };
HostSwitch g_host[] = {
    {"fused_ff", 1},         // LayerNorm -> FeedForward (-> proj_out) as one launch (0: three launches)
    {"twin_attention", 1},   // one attention pass for both CFG rows of the first reference reader
    {"shared_rows", 1},      // conv_in + first resnet once when the CFG rows share their input
    {"oz3", 1},              // the three masked audio out-projections as one GEMM
    {"rowgemm", 1},          // row-stationary LayerNorm / GroupNorm -> projection launches
    {"tleg", 1},             // a level-0 temporal-attention leg as one launch (csrc/tleg.hip)
    {"rconv", 5},            // the UNet resnets' GroupNorm + SiLU + conv3x3 legs as one launch (csrc/rconv.hip): a mask of 1 the 320-wide level, 2 the 640-wide, 4 the 1280-wide; 0 off
    {"conv_out_taps", 1},    // conv_norm_out + SiLU + conv_out (4 channels) as one 36-column GEMM over the pixels (GroupNorm + SiLU in its prologue) + a gather (0: GroupNorm pass + implicit-GEMM conv padded to 64 columns)
    {"ffpo_cat", 1},         // a transformer block's ff2 (+ residual) and proj_out (+ residual) as ONE two-source GEMM with the host-multiplied weight [W_po W2 | W_po] (0: two GEMMs)
    {"sc_cat", 1},           // the resnets' conv_shortcut over [x | skip] as one two-source 1 x 1 conv launch (0: two GEMMs chained through a residual)
    {"up2", 1},              // the convs behind a nearest 2x upsampling as four 2 x 2 convs on the stored image (packing.pack_conv3x3_up2; 0: 3 x 3 on the upsampled view)
    {"rconv_stats", 1},      // ... with the next GroupNorm's statistics from the launch's epilogue (0: a statistics pass over the tensor)
    {"gnconv", 2},           // the VAE's GroupNorm + SiLU + conv3x3 as one launch (csrc/gnconv.hip): 1 with a statistics pass, 2 statistics from the producing launch
    {"zero_audio_skip", 1},  // skip the audio cross-attention of an all-zero (unconditional) audio row
    {"window_state", 1},     // keep what a window's audio / masks determine across the steps of a clip
    {"smga_graph", 1},       // replay the SMGA sampler loop as a HIP graph
};
}  // namespace
extern "C" int mmgt_tune_get(const char* key, int* value) {
  if (key && value)
    for (const HostSwitch& h : g_host)
      if (!strcmp(key, h.key)) { *value = h.value; return 0; }
  mmgt_set_error("tune_get: unknown key");
  return 1;
}
extern "C" int mmgt_tune(const char* key, int value) {
  if (key)
    for (HostSwitch& h : g_host)
      if (!strcmp(key, h.key)) { h.value = value; return 0; }
  if (key && !strcmp(key, "gemm_cfg")) { g_gemm_cfg = value; return 0; }
  if (key && !strcmp(key, "attn64")) { mmgt_attn_set64(value); return 0; }
  if (key && !strcmp(key, "attn_nomax") && (value == 0 || value == 1)) { mmgt_attn_set_nomax(value); return 0; }
  if (key && !strcmp(key, "attn_heads_inner")) { mmgt_attn_set_heads_inner(value); return 0; }
  if (key && !strcmp(key, "attn80")) { mmgt_attn_set_attn80(value); return 0; }
  if (key && !strcmp(key, "g16_pb")) { mmgt_gemm16_set_pb(value); return 0; }
  if (key && !strcmp(key, "g16_stagger")) { mmgt_gemm16_set_stagger(value); return 0; }
  if (key && !strcmp(key, "gn_rows")) { mmgt_gn_set_rows(value); return 0; }
  if (key && !strcmp(key, "gn_interleave")) { mmgt_gn_set_interleave(value); return 0; }
  if (key && !strcmp(key, "gn_narrow")) { mmgt_gn_set_narrow(value); return 0; }
  if (key && !strcmp(key, "gn_slab")) { mmgt_gn_set_slab(value); return 0; }
  if (key && !strcmp(key, "gn_lpr0")) { if (value != 4 && value != 8 && value != 16) return -1; mmgt_gn_set_lpr0(value); return 0; }
  if (key && !strcmp(key, "splitk")) { g_splitk = value; return 0; }
  if (key && !strcmp(key, "bm192")) { g_bm192 = value; return 0; }
  if (key && !strcmp(key, "tailsplit")) { g_tailsplit = value; return 0; }
  if (key && !strcmp(key, "rconv_cb") && (value == 0 || value == 320 || value == 256 || value == 160)) { mmgt_rconv_set_cb(value); return 0; }
  if (key && !strcmp(key, "rconv_stagger") && value >= 0 && value <= 8192) { mmgt_rconv_set_stagger(value); return 0; }
#ifdef MMGT_ABLATE   // libmmgt_hip_abl.so (`make abl`): timing ablations whose RESULTS ARE GARBAGE -- the instruments under tools/ load that library explicitly
  if (key && !strcmp(key, "ffn_dbg") && value >= 0 && value <= 2) { mmgt_ffn_set_dbg(value); return 0; }
  if (key && !strcmp(key, "gnconv_abl") && value >= 0 && value <= 255) { mmgt_gnconv_set_abl(value); return 0; }
  if (key && !strcmp(key, "tleg_abl") && value >= 0 && value <= 128) { mmgt_tleg_set_abl(value); return 0; }
  if (key && !strcmp(key, "rconv_abl") && value >= 0 && value <= 31) { mmgt_rconv_set_abl(value); return 0; }
  if (key && !strcmp(key, "rowgemm_dbg") && value >= 0 && value <= 5) { mmgt_rowgemm_set_dbg(value); return 0; }
#else
  if (key && !strcmp(key, "rowgemm_dbg") && (value == 0 || value == 5)) { mmgt_rowgemm_set_dbg(value); return 0; }   // 5: the stamped (correct) build
  if (key && (!strcmp(key, "ffn_dbg") || !strcmp(key, "gnconv_abl") || !strcmp(key, "tleg_abl") || !strcmp(key, "rconv_abl") || !strcmp(key, "rowgemm_dbg"))) {
    mmgt_set_error("tune: '%s' selects a timing ablation whose results are garbage; the product library does not contain them (build libmmgt_hip_abl.so with `make -C mmgt_amd/csrc abl` and load it through MMGT_LIB)", key);
    return 1;
  }
#endif
  mmgt_set_error("tune: unknown key");
  return 1;
}

// out[m][n] = act(bias[n] + W[n] . A[m]) for a single row (the kernel handles up to 4; the dispatcher sends M == 1: the time-embedding MLP and the resnets' time_emb_proj of one timestep: unet_3d.py:
// 480-500, resnet.py:225-226): a wave per output column streams its weight row once with 16-byte loads against the rows of A (re-read from L2), sums
// across its lanes in a fixed order.  The tile kernels launch 128 x 128 tiles for these: 25 - 35 us per call against the few us the weights take to stream.
namespace {
template <typename T>
__global__ __launch_bounds__(256) void gemv_kernel(const T* __restrict__ A, long lda, const T* __restrict__ W, const float* __restrict__ bias,
                                                   T* __restrict__ out, long ldo, int M, int N, int K, int act) {
  constexpr int VEC = 16 / sizeof(T);
  const int lane = threadIdx.x & 63, n = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (n >= N) return;
  float acc[4] = {0.f, 0.f, 0.f, 0.f};
  const T* w = W + (long)n * K;
  for (int k = lane * VEC; k < K; k += 64 * VEC) {
    union { u32x4 u; T e[VEC]; } wv, av;
    wv.u = *reinterpret_cast<const u32x4*>(w + k);
#pragma unroll
    for (int m = 0; m < 4; ++m)
      if (m < M) {
        av.u = *reinterpret_cast<const u32x4*>(A + m * lda + k);
#pragma unroll
        for (int e = 0; e < VEC; ++e) acc[m] = fmaf(Elem<T>::ld(&wv.e[e]), Elem<T>::ld(&av.e[e]), acc[m]);
      }
  }
#pragma unroll
  for (int m = 0; m < 4; ++m) {
    float v = acc[m];
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    if (lane == 0 && m < M) {
      v += bias ? bias[n] : 0.f;
      if (act == 2) v = silu_f(v);
      Elem<T>::st(out + m * ldo + n, v);
    }
  }
}
}  // namespace

static int gemm_entry(const void* A, long lda, const void* W, const float* bias, const float* bias2, int bias2_rows,
                      const float* row_scale, float alpha, const float* bias_post, const void* residual, long ldr, void* out,
                      long ldo, int M, int N, int K, int act, int batch, long bsA, long bsW, long bsR, long bsO, int dtype,
                      void* stream) {
  if (check_common(dtype, M, N, K, act)) return 1;
  MMGT_CHECK(A && W && out, "gemm: null pointer");
  // (lda < K is allowed: A is only read, and overlapping rows are how a 1-D convolution's patches are laid out in a channels-last signal)
  MMGT_CHECK(lda >= 1 && batch >= 1, "gemm: lda %ld < 1 or batch %d < 1", lda, batch);
  MMGT_CHECK(!bias2 || bias2_rows > 0, "gemm: bias2_rows must be positive");
  const int esz = dtype == MMGT_BF16 ? 2 : 4;
  MMGT_CHECK(((uintptr_t)A % 16) == 0 && ((uintptr_t)W % 16) == 0 && (lda * esz) % 16 == 0 && (bsA * esz) % 16 == 0 &&
                 (bsW * esz) % 16 == 0,
             "gemm: A/W must be 16-byte aligned with 16-byte aligned rows");
  MMGT_CHECK(((long)(M - 1) * lda + K) * esz < (1l << 31) && (long)N * K * esz < (1l << 31),
             "gemm: an operand (per batch entry) exceeds the 2 GiB range of the 32-bit LDS-DMA offsets");
  if (M == 1 && batch == 1 && !bias2 && !row_scale && alpha == 1.f && !bias_post && !residual && (act == 0 || act == 2) && K % (16 / esz) == 0) {
    // (M == 1 only: a CFG row run alone halves M, and where that crossed the kernels' boundary -- 6 rows batched, 3 alone -- the fp32 parity mode's
    //  "one row alone == that row of the batch" gate saw two summation orders; one timestep is one row either way)
    if (dtype == MMGT_BF16)
      hipLaunchKernelGGL(gemv_kernel<bf16_t>, dim3((unsigned)((N + 3) / 4)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)A, lda, (const bf16_t*)W, bias,
                         (bf16_t*)out, ldo, M, N, K, act);
    else
      hipLaunchKernelGGL(gemv_kernel<float>, dim3((unsigned)((N + 3) / 4)), dim3(256), 0, (hipStream_t)stream, (const float*)A, lda, (const float*)W, bias,
                         (float*)out, ldo, M, N, K, act);
    MMGT_LAUNCH_CHECK();
    return 0;
  }
  ADesc ad{};
  ad.src0 = (const char*)A;
  ad.ld0 = lda;
  ad.bs0 = bsA;
  Epi ep{};
  ep.bias = bias; ep.bias2 = bias2; ep.bias2_rows = bias2_rows; ep.row_scale = row_scale; ep.alpha = alpha;
  ep.bias_post = bias_post;
  MMGT_CHECK(!bias_post || (act != 1 && ((uintptr_t)bias_post % 16) == 0), "gemm: bias_post needs a 16-byte aligned vector and no GEGLU");
  ep.residual = (const char*)residual; ep.ldr = ldr; ep.out = (char*)out; ep.ldo = ldo; ep.act = act;
  ep.bsr = bsR; ep.bso = bsO;
  ep.fast = epi_fast(ep, N, act == 1 ? N / 2 : N, esz);
  hipStream_t s = (hipStream_t)stream;
  return dtype == MMGT_BF16 ? launch<bf16_t, 0>(ad, W, bsW, ep, M, N, K, batch, s)
                            : launch<float, 0>(ad, W, bsW, ep, M, N, K, batch, s);
}

extern "C" int mmgt_gemm(const void* A, long lda, const void* W, const float* bias, const float* bias2, int bias2_rows,
                         const float* row_scale, float alpha, const void* residual, long ldr, void* out, long ldo, int M,
                         int N, int K, int act, int batch, long bsA, long bsW, long bsR, long bsO, int dtype,
                         void* stream) {
  return gemm_entry(A, lda, W, bias, bias2, bias2_rows, row_scale, alpha, nullptr, residual, ldr, out, ldo, M, N, K, act,
                    batch, bsA, bsW, bsR, bsO, dtype, stream);
}

extern "C" int mmgt_gemm_post(const void* A, long lda, const void* W, const float* bias, const float* row_scale, float alpha,
                              const float* bias_post, const void* residual, long ldr, void* out, long ldo, int M, int N,
                              int K, int dtype, void* stream) {
  return gemm_entry(A, lda, W, bias, nullptr, 0, row_scale, alpha, bias_post, residual, ldr, out, ldo, M, N, K, 0, 1, 0, 0, 0,
                    0, dtype, stream);
}

extern "C" int mmgt_conv3x3_nhwc(const void* x0, int C0, const void* x1, int C1, int NB, int IH, int IW, int stride,
                                 int upsample, const void* Wp, const float* bias, const float* bias2, int bias2_rows,
                                 const void* residual, void* out, int Cout, int act, int dtype, void* stream) {
  MMGT_CHECK(x0 && Wp && out, "conv3x3: null pointer");
  int pad_lo = 1;
  if (stride == -2) { stride = 2; pad_lo = 0; }   // diffusers Downsample2D(padding=0): F.pad(x, (0, 1, 0, 1)) + stride-2 conv
  MMGT_CHECK(stride == 1 || stride == 2, "conv3x3: stride %d", stride);
  MMGT_CHECK(!(upsample && stride != 1), "conv3x3: upsample requires stride 1");
  MMGT_CHECK(upsample >= 0 && upsample <= 2, "conv3x3: upsample = %d (0, 1, or 2 = the four-phase form with the packing.pack_conv3x3_up2 image)", upsample);
  MMGT_CHECK(C0 % 64 == 0 && C1 % 64 == 0 && (x1 != nullptr) == (C1 > 0),
             "conv3x3: channel counts must be multiples of 64 (C0=%d C1=%d)", C0, C1);
  MMGT_CHECK(act == 0 || act == 2 || act == 3, "conv3x3: act %d unsupported", act);
  MMGT_CHECK(((uintptr_t)x0 % 16) == 0 && ((uintptr_t)x1 % 16) == 0 && ((uintptr_t)Wp % 16) == 0,
             "conv3x3: pointers must be 16-byte aligned");
  // upsample == 2: the conv behind a nearest 2x upsampling as four 2 x 2 convs on the stored image (one per output phase = grid.z), K = 4 Cin each:
  // OH x OW is the PHASE grid (= the stored image), the kernel's epilogue scatters row (n, y, x) of phase (a, b) to pixel (2 y + a, 2 x + b)
  const bool up2 = upsample == 2;
  MMGT_CHECK(!up2 || (dtype == MMGT_BF16 && !residual && !bias2 && act == 0 && !x1 && (Cout % 256 == 0 || Cout % 320 == 0)),
             "conv3x3: the four-phase upsample form is bf16, one source, bias only, Cout a multiple of 256 or 320");
  const int VH = upsample == 1 ? IH * 2 : IH, VW = upsample == 1 ? IW * 2 : IW;
  const int OH = up2 ? IH : (VH + pad_lo + 1 - 3) / stride + 1, OW = up2 ? IW : (VW + pad_lo + 1 - 3) / stride + 1;
  const long M = (long)NB * OH * OW;
  MMGT_CHECK(M < (1l << 31) && (!up2 || 4 * M * Cout * 2 < (1l << 31) * 8), "conv3x3: too many output pixels");
  {
    const long esz_ = dtype == MMGT_BF16 ? 2 : 4, px = (long)NB * IH * IW;
    MMGT_CHECK(px * C0 * esz_ < (1l << 31) && px * C1 * esz_ < (1l << 31) && (long)Cout * 9 * (C0 + C1) * esz_ < (1l << 31),
               "conv3x3: a tensor exceeds the 2 GiB range of the 32-bit LDS-DMA offsets (split the batch)");
  }
  const int K = (up2 ? 4 : 9) * (C0 + C1);
  if (check_common(dtype, (int)M, Cout, K, act)) return 1;
  ADesc ad{};
  ad.src0 = (const char*)x0; ad.src1 = (const char*)x1; ad.C0 = C0; ad.C1 = C1; ad.IH = IH; ad.IW = IW; ad.OH = OH;
  ad.OW = OW; ad.stride = stride; ad.up = upsample; ad.pad = pad_lo;
  make_fastdiv((unsigned)(OH * OW), ad.fd_hw);
  make_fastdiv((unsigned)OW, ad.fd_ow);
  Epi ep{};
  ep.bias = bias; ep.bias2 = bias2; ep.bias2_rows = bias2_rows; ep.alpha = 1.f; ep.residual = (const char*)residual;
  ep.ldr = Cout; ep.out = (char*)out; ep.ldo = Cout; ep.act = act;
  ep.fast = epi_fast(ep, Cout, Cout, dtype == MMGT_BF16 ? 2 : 4);
  hipStream_t s = (hipStream_t)stream;
  if (up2) return launch<bf16_t, 1>(ad, Wp, (long)Cout * K, ep, (int)M, Cout, K, 4, s);
  return dtype == MMGT_BF16 ? launch<bf16_t, 1>(ad, Wp, 0, ep, (int)M, Cout, K, 1, s)
                            : launch<float, 1>(ad, Wp, 0, ep, (int)M, Cout, K, 1, s);
}

// 1 x 1 conv over the channel concatenation of two channels-last tensors: out[p][o] = bias[o] + W[o] . [x0[p] | x1[p]] (+ residual).  The resnets'
// conv_shortcut over [hidden | skip] (resnet.py:243-245) as ONE launch through the conv gather of csrc/gemm16.hip (two sources, one tap) instead of
// two dense GEMMs chained through a residual (the concatenation is never materialised either way).  bf16; C0, C1 multiples of 64; Wp [Cout][C0 + C1].
extern "C" int mmgt_conv1x1_cat_nhwc(const void* x0, int C0, const void* x1, int C1, long rows, const void* Wp, const float* bias, const void* residual,
                                     void* out, int Cout, int dtype, void* stream) {
  MMGT_CHECK(x0 && x1 && Wp && out && rows > 0, "conv1x1_cat: null pointer");
  MMGT_CHECK(dtype == MMGT_BF16, "conv1x1_cat: bf16 only");
  MMGT_CHECK(C0 > 0 && C1 > 0 && C0 % 64 == 0 && C1 % 64 == 0 && (Cout % 256 == 0 || Cout % 320 == 0), "conv1x1_cat: C0 = %d, C1 = %d must be multiples of 64, Cout = %d of 256 or 320",
             C0, C1, Cout);
  MMGT_CHECK(rows < (1l << 31) && rows * C0 * 2 < (1l << 31) && rows * C1 * 2 < (1l << 31) && (long)Cout * (C0 + C1) * 2 < (1l << 31), "conv1x1_cat: an operand exceeds 2 GiB");
  MMGT_CHECK((((uintptr_t)x0 | (uintptr_t)x1 | (uintptr_t)Wp | (uintptr_t)residual | (uintptr_t)out | (uintptr_t)bias) & 15) == 0, "conv1x1_cat: pointers must be 16-byte aligned");
  const int K = C0 + C1;
  if (check_common(dtype, (int)rows, Cout, K, 0)) return 1;
  ADesc ad{};
  // the rows as ONE image row of `rows` pixels: the centre tap of every pixel is the pixel itself, nothing reads a neighbour
  ad.src0 = (const char*)x0; ad.src1 = (const char*)x1; ad.C0 = C0; ad.C1 = C1; ad.IH = 1; ad.IW = (int)rows; ad.OH = 1; ad.OW = (int)rows;
  ad.stride = 1; ad.up = 3; ad.pad = 1;
  make_fastdiv((unsigned)rows, ad.fd_hw);
  make_fastdiv((unsigned)rows, ad.fd_ow);
  Epi ep{};
  ep.bias = bias; ep.alpha = 1.f; ep.residual = (const char*)residual; ep.ldr = Cout; ep.out = (char*)out; ep.ldo = Cout;
  ep.fast = epi_fast(ep, Cout, Cout, 2);
  return launch<bf16_t, 1>(ad, Wp, 0, ep, (int)rows, Cout, K, 1, (hipStream_t)stream);
}
