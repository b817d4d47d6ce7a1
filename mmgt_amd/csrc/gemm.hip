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
//  * BM x BN tile per 256-thread workgroup (4 waves), K chunks of 128 bytes per row (64 bf16 / 32 fp32).
//  * Both operands go global -> LDS by LDS-DMA (global_load_lds_dwordx4, no staging registers): LDS rows are exactly
//    128 B and lane-linear per wave instruction, so bank conflicts are removed by an XOR swizzle of the 16-byte chunk
//    index applied on the per-lane SOURCE address and again on the fragment read (chunk ^ ((row >> 1) & 7): the sixteen
//    lanes of every ds_read_b128 group hit sixteen distinct 16-byte slots).  Conv zero padding = the lane's source is a
//    zero page.
//  * NSTAGE-deep LDS ring, one raw s_barrier per chunk, counted s_waitcnt vmcnt so NSTAGE-2 chunks stay in flight
//    across the barrier.
//  * Epilogue through LDS: accumulators are transposed to row-major so that every lane loads/stores 8 consecutive
//    columns (16 B bf16): bias, per-batch bias, SiLU / GEGLU, row scale, alpha, residual, store.
//  * XCD-aware tile order (n fastest inside an XCD's contiguous run) so the tiles sharing an A row panel hit one L2.
#include <string.h>

#include "common.h"
#include "mmgt_hip.h"

// Main-loop variant (A/B-able by building a second library with -DMMGT_GEMM_VARIANT=n, tools/ab_gemm.py):
//   bit 0: fragments of K-step ks+1 are read while the MFMAs of K-step ks run (register double buffering) + s_setprio
//   bit 1: dense 8-wave tiles spread the next chunk's LDS-DMA issue between the MFMA groups instead of one burst
#ifndef MMGT_GEMM_VARIANT
#define MMGT_GEMM_VARIANT 3
#endif

namespace {

constexpr int ROWB = 128;  // bytes of K per tile row per chunk
constexpr bool V_FRAGDB = (MMGT_GEMM_VARIANT & 1) != 0;
constexpr bool V_ILV = (MMGT_GEMM_VARIANT & 2) != 0;

__device__ __attribute__((aligned(256))) unsigned int g_zero_page[64];

struct ADesc {
  const char* src0;
  const char* src1;
  long ld0;           // dense: row stride (elements)
  long bs0, bs1;      // batch (grid.z) stride in elements
  int C0, C1;         // conv: channels of the two sources (Cin = C0 + C1)
  int IH, IW, OH, OW; // conv: stored input dims and output dims
  int stride, up;     // conv: stride; up = 1 -> the conv sees the nearest-2x upsampled input
};

struct Epi {
  const float* bias;       // [N]
  const float* bias2;      // [ceil(M / bias2_rows)][N]   (time-embedding add: one row per CFG batch entry)
  const float* row_scale;  // [M]                         (motion-mask multiply)
  const char* residual;    // T [M][ldr]
  char* out;               // T [M][ldo]
  long ldr, ldo, bsr, bso; // strides in elements; bs* = grid.z strides
  int bias2_rows;
  float alpha;
  int act;                 // 0 none, 1 GEGLU (packed weights, out has N/2 columns), 2 SiLU, 3 ReLU
  int fast;                // 1: N % 8 == 0 and every row / pointer 16-byte aligned -> vectorised epilogue
};

__device__ __forceinline__ void glds16(const void* gsrc, void* lds_dst) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                   (__attribute__((address_space(3))) void*)lds_dst, 16, 0, 0);
}

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// MODE 0 dense, 1 conv3x3.  WM x WN waves (4 or 8 per workgroup).
//
// The grid is persistent: workgroup b computes tiles b, b + gridDim.x, ... and the chunk stream of its LDS ring keeps
// rolling across tile boundaries -- while the last chunks of tile t are multiplied the first chunks of tile t + 1 are
// already in flight, and the epilogue of tile t (which borrows the ring stage consumed last) runs with them landing
// and its stores draining under the next main loop.  Measured before this: a 256 x 128 tile paid about 7 us of launch +
// first-load latency + store drain per tile, as much as a K = 640 main loop.
template <typename T, int MODE, int BM, int BN, int WM, int WN, int NSTAGE>
__global__ __launch_bounds__(WM * WN * 64) void gemm_kernel(ADesc ad, const char* __restrict__ W, long bsw, Epi ep, int M, int N,
                                                   int K, int tiles_m, int tiles_n) {
  extern __shared__ __attribute__((aligned(1024))) char smem[];
  constexpr int ESZ = sizeof(T);
  constexpr int BK = ROWB / ESZ;                 // elements per chunk
  constexpr int KS = BK / 16;                    // MFMA K-steps per chunk
  constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
  constexpr int A_BYTES = BM * ROWB, B_BYTES = BN * ROWB, STAGE_BYTES = A_BYTES + B_BYTES;
  constexpr int NW = WM * WN;
  constexpr int GA = BM / 8 / NW, GB = BN / 8 / NW;   // 8-row LDS-DMA groups per wave for A and B
  static_assert((NW == 4 || NW == 8) && BM % (WM * 32) == 0 && BN % (WN * 32) == 0 && BM % (8 * NW) == 0 &&
                    BN % (8 * NW) == 0, "tile");

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
  const int srow = lane >> 3, spos = lane & 7;
  const T* a0 = reinterpret_cast<const T*>(ad.src0) + (long)bz * ad.bs0;
  const T* a1 = ad.src1 ? reinterpret_cast<const T*>(ad.src1) + (long)bz * ad.bs1 : nullptr;
  const T* wbase = reinterpret_cast<const T*>(W) + (long)bz * bsw;

  const char* aptr[GA];      // dense: pointer to (row, swizzled chunk) at k = 0
  int cn[GA], coy[GA], cox[GA], achunk[GA];
  const char* wptr[GB];
  // Chunks are prepared strictly in order, so the conv view keeps a running position (tap, channel): inside one tap and
  // one source tensor consecutive chunks are 128 B apart, and the (ky, kx) / padding / pixel address arithmetic is redone
  // only when the tap or the source changes (every Cin / 64 chunks instead of every chunk).
  int p_tap = 0, p_c = 0;
  unsigned okbits = 0;
  auto setup = [&](int tm, int tn) {   // operand addresses of tile (tm, tn), the tile the DMA stream is in
#pragma unroll
    for (int i = 0; i < GA; ++i) {
      const int row = (wid * GA + i) * 8 + srow;
      const int chunk = spos ^ ((row >> 1) & 7);
      int m = tm * BM + row;
      if (m >= M) m = M - 1;
      if (MODE == 0) {
        aptr[i] = reinterpret_cast<const char*>(a0 + (long)m * ad.ld0) + chunk * 16;
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
      const int row = (wid * GB + i) * 8 + srow;
      const int chunk = spos ^ ((row >> 1) & 7);
      int n = tn * BN + row;
      if (n >= N) n = N - 1;
      wptr[i] = reinterpret_cast<const char*>(wbase + (long)n * K) + chunk * 16;
    }
    p_tap = 0;
    p_c = 0;
  };

  // LDS-DMA ops of one chunk are issued in slices spread between the MFMA groups of the previous chunk (a burst of 8
  // global_load_lds costs as many issue cycles as the chunk's 16 MFMAs).  `prep` resolves the per-lane source pointers
  // of the A operand once per chunk (conv: tap / channel decomposition + padding test), `issue` only launches DMAs.
  const char* asrc[GA];
  auto prep = [&](int ch) {
    if (MODE == 0) {
      const long kb = (long)ch * ROWB;   // byte offset along K
#pragma unroll
      for (int i = 0; i < GA; ++i) asrc[i] = aptr[i] + kb;
    } else {
      const int cin = ad.C0 + ad.C1;
      if (p_c == 0 || p_c == ad.C0) {
        const int ky = p_tap / 3, kx = p_tap - ky * 3;
        const int vh = ad.up ? ad.IH * 2 : ad.IH, vw = ad.up ? ad.IW * 2 : ad.IW;
        const bool second = p_c >= ad.C0 && ad.C1 > 0;
        const T* base = second ? a1 : a0;
        const long cpp = second ? ad.C1 : ad.C0;
        okbits = 0;
#pragma unroll
        for (int i = 0; i < GA; ++i) {
          const int iy = coy[i] * ad.stride + ky - 1, ix = cox[i] * ad.stride + kx - 1;
          const bool ok = iy >= 0 && iy < vh && ix >= 0 && ix < vw;
          const int sy = ad.up ? iy >> 1 : iy, sx = ad.up ? ix >> 1 : ix;
          const char* p = reinterpret_cast<const char*>(base + (((long)cn[i] * ad.IH + sy) * ad.IW + sx) * cpp) + achunk[i];
          asrc[i] = ok ? p : reinterpret_cast<const char*>(g_zero_page) + spos * 16;
          okbits |= ok ? (1u << i) : 0u;
        }
      } else {
#pragma unroll
        for (int i = 0; i < GA; ++i) asrc[i] += (okbits >> i & 1u) ? ROWB : 0;
      }
      p_c += BK;
      if (p_c == cin) { p_c = 0; ++p_tap; }
    }
  };
  auto issue = [&](int stage, int ch, int part, int nparts) {  // chunk `ch` of the DMA-side tile -> LDS stage `stage`
    char* st = smem + stage * STAGE_BYTES;
    const long kb = (long)ch * ROWB;
#pragma unroll
    for (int i = 0; i < GA; ++i)
      if (i % nparts == part) glds16(asrc[i], st + (wid * GA + i) * 1024);
#pragma unroll
    for (int i = 0; i < GB; ++i)
      if (i % nparts == part) glds16(wptr[i] + kb, st + A_BYTES + (wid * GB + i) * 1024);
  };

  // fragment read addressing: row r of the tile, 16-byte chunk c  ->  r * 128 + ((c ^ ((r >> 1) & 7)) * 16)
  const int arow = wm * (BM / WM) + lr, brow = wn * (BN / WN) + lr;   // + 32 * tile index (keeps (row>>1)&7 pattern)

  // ---- epilogue geometry: per wave, SLAB rows at a time through LDS (row-major fp32, 4-float pad), then 8-column
  // vectors.  The transpose buffer of all waves must fit into ONE ring stage (the other stages hold the next tile).
  constexpr int WCOLS = 32 * TN;
  constexpr int ESTR = WCOLS + 4;
  constexpr int SLAB = (NW * 32 * ESTR * 4 <= STAGE_BYTES) ? 32 : 16;
  static_assert(NW * SLAB * ESTR * 4 <= STAGE_BYTES, "epilogue buffer must fit one stage");
  constexpr int NSLAB = TM * (32 / SLAB);
  constexpr int VPR = WCOLS / 8;                    // 8-column vectors per row of a wave's slab
  constexpr int ENV = SLAB * VPR / 64;              // vectors per lane per slab
  constexpr bool RES_PF = ESZ == 2 && TN <= 2;      // wide wave tiles have no registers to spare for it

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
#pragma unroll
  for (int s = 0; s < NSTAGE - 1; ++s)
    if (gi < total) { prep(ich); issue(sl, ich, 0, 1); advance_dma(); }

  for (int vt = blockIdx.x; vt < nwg; vt += G) {
    int tm, tn;
    decode(vt, tm, tn);
    const int row0 = tm * BM + wm * (BM / WM), col0 = tn * BN + wn * (BN / WN);

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) acc[i][j] = (f32x16)(0.f);

    // ---- residual prefetch (bf16 fast path): the epilogue's residual vectors are requested before the main loop, so
    // their HBM latency hides under it instead of being exposed once per slab (measured: -35% on the L0 N = K = 320
    // projections).  vmcnt completes in order, so older / younger extra ops only make the counted waits conservative.
    u32x4 rres[RES_PF ? NSLAB : 1][RES_PF ? ENV : 1];
    const bool res_pf = RES_PF && ep.residual != nullptr && ep.fast && ep.act != 1;
    if (res_pf) {
      const T* resb = reinterpret_cast<const T*>(ep.residual) + (long)bz * ep.bsr;
#pragma unroll
      for (int sb = 0; sb < (RES_PF ? NSLAB : 1); ++sb)
#pragma unroll
        for (int t = 0; t < (RES_PF ? ENV : 1); ++t) {
          const int v = lane + 64 * t;
          const int rr = v / VPR, hc = (v - rr * VPR) * 8;
          const int m = row0 + sb * SLAB + rr, n = col0 + hc;
          rres[sb][t] = (m < M && n < N) ? *reinterpret_cast<const u32x4*>(resb + (long)m * ep.ldr + n) : (u32x4)(0u);
        }
    }

    // ---- bias prefetch (8-wave tiles = one workgroup per CU, where nothing else hides an L2 round trip per slab): a
    // lane's vector slot t covers the same 8 columns in every slab, so the bias (and GEGLU gate bias) vectors are loaded
    // once per tile, before the main loop.  The 4-wave tiles keep the per-slab loads: they have no registers to spare
    // and two or three resident workgroups to cover the latency.
    constexpr bool BIAS_PF = NW == 8;
    f32x4 bvec[BIAS_PF ? ENV : 1][2], gvec[BIAS_PF ? ENV : 1][2];
    auto load_bias = [&]() {
      const float* zero = reinterpret_cast<const float*>(g_zero_page);
      const bool geglu = ep.act == 1;
#pragma unroll
      for (int t = 0; t < (BIAS_PF ? ENV : 1); ++t) {
        const int v = lane + 64 * t;
        int hcol;
        if (geglu) {
          const int per = VPR / 2, g = v % per;
          hcol = (g >> 2) * 64 + (g & 3) * 8;
        } else {
          hcol = (v % VPR) * 8;
        }
        const int ncol = col0 + hcol;
        const bool ok = ep.fast && ep.bias != nullptr && ncol < N;
        const float* bp = ok ? ep.bias + ncol : zero;
        const float* gp = ok && geglu ? ep.bias + ncol + 32 : zero;
        bvec[t][0] = *reinterpret_cast<const f32x4*>(bp);
        bvec[t][1] = *reinterpret_cast<const f32x4*>(bp + 4);
        gvec[t][0] = *reinterpret_cast<const f32x4*>(gp);
        gvec[t][1] = *reinterpret_cast<const f32x4*>(gp + 4);
      }
    };
    if (BIAS_PF) load_bias();

    for (int ch = 0; ch < nchunks; ++ch, ++gc) {
      // chunk gc must have landed; up to NSTAGE-2 younger chunks may stay in flight (GA + GB LDS-DMA ops per wave each)
      const int younger = total - 1 - gc;
      if (NSTAGE == 2 || younger < 1) wait_vmcnt<0>();
      else if (NSTAGE == 3 || younger < 2) wait_vmcnt<(GA + GB)>();
      else wait_vmcnt<2 * (GA + GB)>();
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
      Frag<T> fa[2][TM], fb[2][TN];
      auto load_frags = [&](int ks, Frag<T>* pa, Frag<T>* pb) {
#pragma unroll
        for (int i = 0; i < TM; ++i) {
          const int r = arow + 32 * i;
          const int sw = (r >> 1) & 7;
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
          const int sw = (r >> 1) & 7;
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
      if (V_FRAGDB) load_frags(0, fa[0], fb[0]);
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        if (V_FRAGDB) {
          if (ks + 1 < KS) load_frags(ks + 1, fa[(ks + 1) & 1], fb[(ks + 1) & 1]);
          __builtin_amdgcn_s_setprio(1);
        } else {
          load_frags(ks, fa[ks & 1], fb[ks & 1]);
        }
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j) mma32(acc[i][j], fa[ks & 1][i], fb[ks & 1][j]);
        if (V_FRAGDB) __builtin_amdgcn_s_setprio(0);
        if (ILV && more) issue(st_i, ch_i, ks, KS);   // this slice's LDS-DMA issues under the MFMAs just queued
      }
      if (more) advance_dma();
      sc = sc + 1 == NSTAGE ? 0 : sc + 1;
    }

    // ---- epilogue.  Every wave is done with the stage consumed last (raw barrier: the DMAs of the next tile stay in
    // flight); that stage is the transpose buffer, and the next DMA into it is issued behind the next tile's first
    // barrier, which every wave reaches only after its own LDS reads below have completed.
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    const int se = sc == 0 ? NSTAGE - 1 : sc - 1;
    float* ebuf = reinterpret_cast<float*>(smem + se * STAGE_BYTES) + wid * SLAB * ESTR;
    T* out = reinterpret_cast<T*>(ep.out) + (long)bz * ep.bso;
    const T* res = ep.residual ? reinterpret_cast<const T*>(ep.residual) + (long)bz * ep.bsr : nullptr;
    const bool geglu = ep.act == 1;
    const float* zero = reinterpret_cast<const float*>(g_zero_page);
#pragma unroll
    for (int sb = 0; sb < NSLAB; ++sb) {
      const int i = sb / (32 / SLAB), half = sb % (32 / SLAB);
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int r = 0; r < SLAB / 2; ++r) {
          const int reg = SLAB == 32 ? r : half * 8 + r;
          const int lrow = SLAB == 32 ? acc_row(reg, lane) : (r & 3) + 8 * (r >> 2) + 4 * lh;
          ebuf[lrow * ESTR + j * 32 + lr] = acc[i][j][reg];
        }
      __builtin_amdgcn_wave_barrier();
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      const int nvec = geglu ? SLAB * (VPR / 2) : SLAB * VPR;
#pragma unroll
      for (int t = 0; t < ENV; ++t) {
        const int v = lane + 64 * t;
        if (v >= nvec) continue;
        int rr, hcol, ncol;      // row in the slab, column of the (h) vector inside the wave tile, global column
        long ocol;
        if (geglu) {             // columns [0,32) of every 64 = h, [32,64) = gate of the same 32 output channels
          const int per = VPR / 2;
          rr = v / per;
          const int g = v - rr * per;
          const int blk = g >> 2, c8 = (g & 3) * 8;
          hcol = blk * 64 + c8;
          ncol = col0 + hcol;
          ocol = (long)((col0 + blk * 64) >> 1) + c8;
        } else {
          rr = v / VPR;
          hcol = (v - rr * VPR) * 8;
          ncol = col0 + hcol;
          ocol = ncol;
        }
        const int m = row0 + sb * SLAB + rr;
        if (m >= M || ncol >= N) continue;
        const float* hp = ebuf + rr * ESTR + hcol;
        float o8[8];
        {
          const f32x4 x0 = *reinterpret_cast<const f32x4*>(hp), x1 = *reinterpret_cast<const f32x4*>(hp + 4);
#pragma unroll
          for (int e = 0; e < 4; ++e) { o8[e] = x0[e]; o8[4 + e] = x1[e]; }
        }
        if (ep.fast) {
          // branch-free vector path: absent operands read a zero page
          const float* bp = ep.bias ? ep.bias + ncol : zero;
          const f32x4 b0 = BIAS_PF ? bvec[BIAS_PF ? t : 0][0] : *reinterpret_cast<const f32x4*>(bp);
          const f32x4 b1 = BIAS_PF ? bvec[BIAS_PF ? t : 0][1] : *reinterpret_cast<const f32x4*>(bp + 4);
#pragma unroll
          for (int e = 0; e < 4; ++e) { o8[e] += b0[e]; o8[4 + e] += b1[e]; }
          if (geglu) {
            const float* gp = ep.bias ? ep.bias + ncol + 32 : zero;
            const f32x4 g0 = BIAS_PF ? gvec[BIAS_PF ? t : 0][0] : *reinterpret_cast<const f32x4*>(gp);
            const f32x4 g1 = BIAS_PF ? gvec[BIAS_PF ? t : 0][1] : *reinterpret_cast<const f32x4*>(gp + 4);
            const f32x4 y0 = *reinterpret_cast<const f32x4*>(hp + 32), y1 = *reinterpret_cast<const f32x4*>(hp + 36);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              o8[e] *= gelu_erf_f(y0[e] + g0[e]);
              o8[4 + e] *= gelu_erf_f(y1[e] + g1[e]);
            }
          } else {
            const float* b2p = ep.bias2 ? ep.bias2 + (long)(m / ep.bias2_rows) * N + ncol : zero;
            const f32x4 c0 = *reinterpret_cast<const f32x4*>(b2p), c1 = *reinterpret_cast<const f32x4*>(b2p + 4);
            const float rs = (ep.row_scale ? ep.row_scale[m] : 1.f) * ep.alpha;
#pragma unroll
            for (int e = 0; e < 4; ++e) { o8[e] += c0[e]; o8[4 + e] += c1[e]; }
            if (ep.act == 2) {
#pragma unroll
              for (int e = 0; e < 8; ++e) o8[e] = silu_f(o8[e]);
            } else if (ep.act == 3) {
#pragma unroll
              for (int e = 0; e < 8; ++e) o8[e] = fmaxf(o8[e], 0.f);
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) o8[e] *= rs;
          }
          T* op = out + (long)m * ep.ldo + ocol;
          if (ESZ == 2) {
            union { u32x4 u; bf16_t e[8]; } rv;
            if (res_pf) rv.u = rres[RES_PF ? sb : 0][RES_PF ? t : 0];
            else rv.u = *reinterpret_cast<const u32x4*>(res ? reinterpret_cast<const char*>(res + (long)m * ep.ldr + ocol)
                                                            : reinterpret_cast<const char*>(zero));
            union { bf16_t e[8]; u32x4 u; } pk;
#pragma unroll
            for (int e = 0; e < 8; ++e) pk.e[e] = f32_to_bf16(o8[e] + bf16_to_f32(rv.e[e]));
            *reinterpret_cast<u32x4*>(op) = pk.u;
          } else {
            const float* rp = res ? reinterpret_cast<const float*>(res) + (long)m * ep.ldr + ocol : zero;
            const f32x4 r0 = *reinterpret_cast<const f32x4*>(rp), r1 = *reinterpret_cast<const f32x4*>(rp + 4);
            *reinterpret_cast<f32x4*>(op) = (f32x4){o8[0] + r0[0], o8[1] + r0[1], o8[2] + r0[2], o8[3] + r0[3]};
            *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(op) + 4) =
                (f32x4){o8[4] + r1[0], o8[5] + r1[1], o8[6] + r1[2], o8[7] + r1[3]};
          }
        } else {
          // generic scalar path (ragged N or unaligned rows)
          for (int e = 0; e < 8 && ncol + e < N; ++e) {
            float x = o8[e];
            if (ep.bias) x += ep.bias[ncol + e];
            if (geglu) {
              float gte = hp[32 + e];
              if (ep.bias) gte += ep.bias[ncol + 32 + e];
              x *= gelu_erf_f(gte);
            } else {
              if (ep.bias2) x += ep.bias2[(long)(m / ep.bias2_rows) * N + ncol + e];
              if (ep.act == 2) x = silu_f(x);
              if (ep.act == 3) x = fmaxf(x, 0.f);
              x *= (ep.row_scale ? ep.row_scale[m] : 1.f) * ep.alpha;
            }
            if (res) x += Elem<T>::ld(res + (long)m * ep.ldr + ocol + e);
            Elem<T>::st(out + (long)m * ep.ldo + ocol + e, x);
          }
        }
      }
      __builtin_amdgcn_wave_barrier();
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
  }
}

template <typename T, int MODE, int BM, int BN, int WM, int WN, int NSTAGE>
int launch_cfg(const ADesc& ad, const void* W, long bsw, const Epi& ep, int M, int N, int K, int batch, hipStream_t s) {
  const int tiles_m = (M + BM - 1) / BM, tiles_n = (N + BN - 1) / BN;
  const size_t lds = (size_t)NSTAGE * (BM + BN) * ROWB;
  auto kern = gemm_kernel<T, MODE, BM, BN, WM, WN, NSTAGE>;
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

int g_gemm_cfg = 0;   // 0 = heuristic; 1..5 force a tile configuration (mmgt_tune("gemm_cfg", v), benchmarking only)

template <typename T, int MODE>
int launch(const ADesc& ad, const void* W, long bsw, const Epi& ep, int M, int N, int K, int batch, hipStream_t s) {
  // Tile choice: 128 x 128 with a 3-deep ring for compute-bound shapes; for short reductions (the HBM-bound L0/L1
  // projections) smaller tiles with several resident workgroups per CU keep more loads in flight.
  const bool geglu = ep.act == 1;
  int cfg = g_gemm_cfg;
  if (cfg == 0) {
    // Measured on MI355X with tools/ab_gemm.py / tools/bench_kernels.py (one process, one device):
    //   cfg 6 (256x128, 8 waves, 3-deep ring)  long reductions, GEGLU, and the widest L0 projections;
    //   cfg 1 (128x128, 2 workgroups / CU)      the conv gather;
    //   cfg 3 (128x64, 3 workgroups / CU)       short reductions, narrow outputs, grids that would not fill the chip.
    const long tiles128 = (long)((M + 127) / 128) * ((N + 127) / 128) * batch;
    const long tiles256 = (long)((M + 255) / 256) * ((N + 127) / 128) * batch;
    if (MODE == 1) cfg = tiles128 < 512 ? 3 : 1;
    else if (geglu) cfg = tiles256 >= 256 ? 6 : 1;
    else if (K >= 1280) cfg = tiles256 >= 256 ? 6 : 3;
    else if (M >= 131072 && N >= 640) cfg = 6;
    else cfg = 3;
  }
  if (geglu && (cfg == 3 || cfg == 4 || cfg == 5 || cfg >= 8)) cfg = 1;
  switch (cfg) {
    case 1: return launch_cfg<T, MODE, 128, 128, 2, 2, 2>(ad, W, bsw, ep, M, N, K, batch, s);
    case 2: return launch_cfg<T, MODE, 128, 128, 2, 2, 3>(ad, W, bsw, ep, M, N, K, batch, s);
    case 3: return launch_cfg<T, MODE, 128, 64, 2, 2, 2>(ad, W, bsw, ep, M, N, K, batch, s);
    case 4: return launch_cfg<T, MODE, 128, 64, 2, 2, 3>(ad, W, bsw, ep, M, N, K, batch, s);
    case 5: return launch_cfg<T, MODE, 64, 64, 2, 2, 2>(ad, W, bsw, ep, M, N, K, batch, s);
    case 6: return launch_cfg<T, MODE, 256, 128, 4, 2, 3>(ad, W, bsw, ep, M, N, K, batch, s);
    case 7: return launch_cfg<T, MODE, 256, 128, 4, 2, 2>(ad, W, bsw, ep, M, N, K, batch, s);
    default: return launch_cfg<T, MODE, 256, 64, 4, 2, 3>(ad, W, bsw, ep, M, N, K, batch, s);
  }
}

int epi_fast(const Epi& ep, int N, int n_out, int esz) {
  auto al16 = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
  bool ok = N % 8 == 0 && n_out % 8 == 0 && al16(ep.out) && (ep.ldo * esz) % 16 == 0 && (ep.bso * esz) % 16 == 0;
  ok = ok && al16(ep.bias) && al16(ep.bias2);
  if (ep.residual) ok = ok && al16(ep.residual) && (ep.ldr * esz) % 16 == 0 && (ep.bsr * esz) % 16 == 0;
  return ok ? 1 : 0;
}

int check_common(int dtype, int M, int N, int K, int act) {
  MMGT_CHECK(dtype == MMGT_F32 || dtype == MMGT_BF16, "gemm: bad dtype %d", dtype);
  MMGT_CHECK(M > 0 && N > 0 && K > 0, "gemm: empty problem M=%d N=%d K=%d", M, N, K);
  MMGT_CHECK(K % 64 == 0, "gemm: K=%d must be a multiple of 64 (pad channels on the host)", K);
  MMGT_CHECK(act >= 0 && act <= 3, "gemm: bad act %d", act);
  MMGT_CHECK(act != 1 || N % 64 == 0, "gemm: GEGLU needs N %% 64 == 0 (N=%d)", N);
  return 0;
}

}  // namespace

extern "C" int mmgt_tune(const char* key, int value) {
  if (key && !strcmp(key, "gemm_cfg")) { g_gemm_cfg = value; return 0; }
  mmgt_set_error("tune: unknown key");
  return 1;
}

extern "C" int mmgt_gemm(const void* A, long lda, const void* W, const float* bias, const float* bias2, int bias2_rows,
                         const float* row_scale, float alpha, const void* residual, long ldr, void* out, long ldo, int M,
                         int N, int K, int act, int batch, long bsA, long bsW, long bsR, long bsO, int dtype,
                         void* stream) {
  if (check_common(dtype, M, N, K, act)) return 1;
  MMGT_CHECK(A && W && out, "gemm: null pointer");
  MMGT_CHECK(lda >= K && batch >= 1, "gemm: lda %ld < K %d or batch %d < 1", lda, K, batch);
  MMGT_CHECK(!bias2 || bias2_rows > 0, "gemm: bias2_rows must be positive");
  const int esz = dtype == MMGT_BF16 ? 2 : 4;
  MMGT_CHECK(((uintptr_t)A % 16) == 0 && ((uintptr_t)W % 16) == 0 && (lda * esz) % 16 == 0 && (bsA * esz) % 16 == 0 &&
                 (bsW * esz) % 16 == 0,
             "gemm: A/W must be 16-byte aligned with 16-byte aligned rows");
  ADesc ad{};
  ad.src0 = (const char*)A;
  ad.ld0 = lda;
  ad.bs0 = bsA;
  Epi ep{};
  ep.bias = bias; ep.bias2 = bias2; ep.bias2_rows = bias2_rows; ep.row_scale = row_scale; ep.alpha = alpha;
  ep.residual = (const char*)residual; ep.ldr = ldr; ep.out = (char*)out; ep.ldo = ldo; ep.act = act;
  ep.bsr = bsR; ep.bso = bsO;
  ep.fast = epi_fast(ep, N, act == 1 ? N / 2 : N, esz);
  hipStream_t s = (hipStream_t)stream;
  return dtype == MMGT_BF16 ? launch<bf16_t, 0>(ad, W, bsW, ep, M, N, K, batch, s)
                            : launch<float, 0>(ad, W, bsW, ep, M, N, K, batch, s);
}

extern "C" int mmgt_conv3x3_nhwc(const void* x0, int C0, const void* x1, int C1, int NB, int IH, int IW, int stride,
                                 int upsample, const void* Wp, const float* bias, const float* bias2, int bias2_rows,
                                 const void* residual, void* out, int Cout, int act, int dtype, void* stream) {
  MMGT_CHECK(x0 && Wp && out, "conv3x3: null pointer");
  MMGT_CHECK(stride == 1 || stride == 2, "conv3x3: stride %d", stride);
  MMGT_CHECK(!(upsample && stride != 1), "conv3x3: upsample requires stride 1");
  MMGT_CHECK(C0 % 64 == 0 && C1 % 64 == 0 && (x1 != nullptr) == (C1 > 0),
             "conv3x3: channel counts must be multiples of 64 (C0=%d C1=%d)", C0, C1);
  MMGT_CHECK(act == 0 || act == 2 || act == 3, "conv3x3: act %d unsupported", act);
  MMGT_CHECK(((uintptr_t)x0 % 16) == 0 && ((uintptr_t)x1 % 16) == 0 && ((uintptr_t)Wp % 16) == 0,
             "conv3x3: pointers must be 16-byte aligned");
  const int VH = upsample ? IH * 2 : IH, VW = upsample ? IW * 2 : IW;
  const int OH = (VH + 2 - 3) / stride + 1, OW = (VW + 2 - 3) / stride + 1;
  const long M = (long)NB * OH * OW;
  MMGT_CHECK(M < (1l << 31), "conv3x3: too many output pixels");
  const int K = 9 * (C0 + C1);
  if (check_common(dtype, (int)M, Cout, K, act)) return 1;
  ADesc ad{};
  ad.src0 = (const char*)x0; ad.src1 = (const char*)x1; ad.C0 = C0; ad.C1 = C1; ad.IH = IH; ad.IW = IW; ad.OH = OH;
  ad.OW = OW; ad.stride = stride; ad.up = upsample;
  Epi ep{};
  ep.bias = bias; ep.bias2 = bias2; ep.bias2_rows = bias2_rows; ep.alpha = 1.f; ep.residual = (const char*)residual;
  ep.ldr = Cout; ep.out = (char*)out; ep.ldo = Cout; ep.act = act;
  ep.fast = epi_fast(ep, Cout, Cout, dtype == MMGT_BF16 ? 2 : 4);
  hipStream_t s = (hipStream_t)stream;
  return dtype == MMGT_BF16 ? launch<bf16_t, 1>(ad, Wp, 0, ep, (int)M, Cout, K, 1, s)
                            : launch<float, 1>(ad, Wp, 0, ep, (int)M, Cout, K, 1, s);
}
