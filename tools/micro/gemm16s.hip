// gemm16's tile, fragments, LDS image and LDS-DMA streams with the two wave groups kept ONE CHUNK (+ one barrier) apart, so that a group's
// epilogue runs beside the other group's MFMAs (gfx950).
//
// What round 3 measured on gemm16_kernel (profiles/r3/gemm16_tile_trace_r3.txt, per-tile stamps): 30 % of a K = 640 tile (15 % at K = 1280) is
// not its main loop -- both wave groups run their epilogue at the same time with no MFMA beside it (2.6 / 4.0 us; 10 / 13 us on the 320-column
// residual tile, whose row tiles each wait for their own residual vectors), group 0 then waits 1.7 us at its first barrier for group 1.  Round 4's
// whole-step counters (profiles/r4/pmc_step_mfma_r4a.json): matrix pipe 41 % busy on gemm16<0,256>, 25 % on gemm16<0,320>.
//
// Structure.  A workgroup's time is cut into SLOTS of NPH = BN / 64 phases {load part | s_barrier | multiply part | s_barrier}.  A tile is
// nch = K / 64 main slots (one 64-deep chunk each, as in gemm16_kernel) followed by ONE epilogue slot of the same 2 NPH barriers, in which the
// group converts and stores its four row tiles in its load parts (= beside the other group's multiply parts).  Group 1 (waves 4-7, rows 128..255
// of the tile) runs the same sequence ONE SLOT (and the usual one barrier) behind group 0: while group 0 is in its epilogue slot group 1
// multiplies its last chunk, and while group 1 is in its epilogue slot group 0 multiplies the first chunk of the next tile.
//   * W is shared by the groups, which read it one chunk apart.  It lives in a RING of column pairs (a pair = the 2 x 32 W rows one phase
//     multiplies with: 8 KiB): 10 pairs at BN = 256, 11 at BN = 320 -- what the 160 KiB leave beside the A stages and the bias vectors.  The pair
//     group 0 reads in wall phase T is issued in wall phase T - LEADP, LEADP = ring - NPH - 1 = 5 phases (>= one chunk ahead, as in gemm16_kernel):
//     its ring slot was last read by group 1 NPH + 1 phases before that, behind a barrier every issuer has passed.  The W stream is timed by the
//     WALL phase (group 0's position), so every wave issues one W piece per phase whichever group it belongs to.
//   * A is private to a group (rows 32 w .. of wave w), two chunks ahead of the group's own position.
//   * THE ISSUE SCHEDULE IS STATIC: every wave issues the same vector-memory operations in every phase of every slot -- one W piece, the phase's
//     A pieces, in an epilogue slot the row tiles' stores.  Where the stream has nothing to fetch (the target
//     phase lies in an epilogue slot, or behind the last tile) the piece goes out with an out-of-range offset: it reads zeros, moves no memory,
//     and lands in LDS that is free by construction (the ring slot the next real pair will take; the A stage consumed in this slot).  So the
//     count for every s_waitcnt vmcnt is a compile-time constant of (slot kind, previous slot kind, phase) -- struct Sched -- and the scalar
//     bookkeeping per phase is a handful of instructions.  (The first version tracked issue sequence numbers at run time: 400-550 scalar
//     instructions per phase with the tile decode inlined twelve times, 2x slower than gemm16_kernel.)  The only run-time part: an epilogue's
//     stores are counted when the tile is full (every lane of every wave stores) and as zero otherwise -- under-counting only waits longer.
//     The bias pieces (three waves, once per tile) are never counted, for the same reason.
//   * Bias vectors: double-buffered per tile parity (group 1 initialises a tile one slot after group 0 fetched the next one).
//   * The tile order is decoded once per tile (the group's next tile, at its tile start); the streams take it from there when they cross
//     into the next tile (K >= 192: a stream is never more than one tile ahead of either group).
// Modes: dense / conv3x3 gather (MODE), BN 256 / 320, bias + per-batch bias, GEGLU (256), split-K slabs.  A residual, row scale / post-scale
// bias stay on gemm16_kernel.  Results: bit-identical to gemm16_kernel.
//
// STATUS (round 4): NOT the default -- mmgt_tune("g16_ver", 2) routes the shapes without a residual here.  Measured on
// the denoise step's shapes (profiles/r4/ab_gemm16s_r4.txt): bit-identical, 1.28-1.33x gemm16_kernel's TIME (conv 48 x 32^2 1920->640: 956 us
// against 746; GEGLU 49152 x 5120 x 640: 442 against 333).  (A residual variant -- the NEXT tile's residual vectors loaded in the epilogue slot
// into the registers the stored accumulators leave, the next tile's sum starting from them -- was written and dropped: inline-asm loads into
// accumulator registers made hipcc spill 24-130 of them, 1.5-2.2x gemm16_kernel's time and wrong integers on a ragged conv tile.)  What the
// ablation builds show (profiles/r4/abl_gemm16s_r4.txt): MFMAs + epilogue alone 193-272 us where gemm16_kernel takes 296-343 for everything;
// the fragment reads add 30-90 us, the LDS-DMA stream another 45-116 -- the counted waits themselves only 3-8 % -- i.e. the per-phase cadence
// of this stream (one W piece in EVERY load part, A pieces in three of four) costs more beside the other group's MFMAs than gemm16_kernel's
// (two pieces per phase, no DMA-dependent wait in three phases of four), and that outweighs what the hidden epilogue returns.  Two things the
// work established and gemm16_kernel's next revision can use: an instruction in a load part costs ~8 cycles beside a wave that issues MFMAs at
// raised priority (moving ~40 scalar instructions of bookkeeping behind the MFMAs took 8 % off), and a TAKEN scalar branch there costs tens.
#include <type_traits>

#include "common.h"
#include "gemm_common.h"
#include "mmgt_hip.h"

namespace {

typedef __attribute__((ext_vector_type(4))) float acc4;

__device__ __forceinline__ acc4 mma16s(s16x8 a, s16x8 b, acc4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }

// The static issue schedule of one wave and the s_waitcnt counts that follow from it.  Order inside a phase: W piece | A pieces | (epilogue slot)
// per finished row tile its stores, then its residual loads | s_waitcnt | barrier.
template <int BN, int EPI>
struct Sched {
  static constexpr int NPH = BN / 64, NT = BN / 32, NPAIR = NT / 2, RING = BN == 256 ? 10 : 11, LEADP = RING - NPH - 1, GA = 4;
  static constexpr int RPP = 1;                                                  // row tiles an epilogue phase finishes
  static constexpr int S = EPI == 1 ? NPAIR / 2 : EPI == 3 ? NT : NPAIR;         // stores per row tile
  static constexpr int R = 0;                                                    // (loads an epilogue phase issues behind its stores: none)
  static constexpr int a_lo(int P) { return BN == 256 ? (P <= 1 ? 0 : P) : P - 1; }       // first A piece of phase P ...
  static constexpr int a_n(int P) { return P == 0 ? 0 : (BN == 256 && P == 1) ? 2 : 1; }  // ... and how many (256: 0 2 1 1; 320: 0 1 1 1 1)
  static constexpr int rows_lo(int P) { return P * RPP < 4 ? P * RPP : 4; }
  static constexpr int rows_n(int P) { return rows_lo(P + 1) - rows_lo(P); }
  static constexpr int last_row_phase() { return (4 + RPP - 1) / RPP - 1; }
  static constexpr int extra(bool epi, int P, bool st) { return epi ? rows_n(P) * ((st ? S : 0) + R) : 0; }
  static constexpr int ops(bool epi, int P, bool st) { return 1 + a_n(P) + extra(epi, P, st); }
  // phases j = 0 .. NPH - 1: the previous slot (epilogue slot? pe), NPH .. 2 NPH - 1: the current one (ce); st: the epilogue's stores count
  static constexpr int opsj(int j, bool pe, bool ce, bool st) { return j < NPH ? ops(pe, j, st) : ops(ce, j - NPH, st); }
  // vmcnt in front of the first barrier of phase P: what may stay in flight
  static constexpr int waitn(int P, bool pe, bool ce, bool st) {
    // this wave's piece of the W pair group 0 reads in the next wall phase: the first operation of the phase LEADP - 1 back
    const int jw = NPH + P - (LEADP - 1);
    int n = opsj(jw, pe, ce, st) - 1;
    for (int j = jw + 1; j <= NPH + P; ++j) n += opsj(j, pe, ce, st);
    if (P == NPH - 1) {
      // the A chunk of the group's next main slot: its last piece went out in the previous slot's last phase
      int na = extra(pe, NPH - 1, st);
      for (int j = NPH; j < 2 * NPH; ++j) na += opsj(j, pe, ce, st);
      n = na < n ? na : n;
      if (ce && R) {                                   // the next tile's residual: the tile starts behind this phase's barriers
        int nr = 0;
        for (int j = last_row_phase() + 1; j < NPH; ++j) nr += ops(true, j, st);
        n = nr < n ? nr : n;
      }
    }
    return n;
  }
};

// EPI: 0 plain (bias in the accumulators' start), 1 GEGLU (BN = 256), 3 split-K fp32 slabs
template <int MODE, int BN, int EPI>
__global__ __launch_bounds__(512, 2) void gemm16s_kernel(ADesc ad, const char* __restrict__ W, long bsw, Epi ep, int M, int N, int K, int tiles_m,
                                                         int tiles_n, int pb
#if G16S_TRACE
                                                         , unsigned long long* trace, int dbg   // diagnostic build (make trace): phase stamps of workgroup 0; dbg: timing ablations, 1 no DMA, 2 no vmcnt waits, 4 no fragment reads, 8 no MFMAs (wrong results)
#endif
) {
  extern __shared__ __attribute__((aligned(1024))) char smem[];
#if !G16S_TRACE
#ifndef G16S_ABL
#define G16S_ABL 0
#endif
  constexpr int dbg = G16S_ABL;   // timing ablations (make abl): 1 no LDS-DMA, 2 no vmcnt waits, 4 no fragment reads, 8 no MFMAs, 16 no epilogue rows, 32 stores confined to 16 rows per workgroup, 64 epilogue arithmetic without stores, 128 nt stores
#endif
  typedef bf16_t T;
  typedef Sched<BN, EPI> SC;
  static_assert(BN == 256 || BN == 320, "BN");
  static_assert(EPI != 1 || BN == 256, "GEGLU: 256 columns");
  static_assert(EPI != 2, "no residual variant (see the header)");
  constexpr int ESZ = 2, BM = 256, ROWB = 128, BK = 64;
  constexpr int CPR = 8, RPD = 8, GA = SC::GA;
  constexpr int NPH = SC::NPH, NT = SC::NT, NPAIR = SC::NPAIR;
  constexpr int RING = SC::RING, LEADP = SC::LEADP;
  constexpr int A_STAGE = BM * ROWB, W_OFF = 2 * A_STAGE, PAIR_BYTES = 64 * ROWB;
  constexpr int BIAS_OFF = W_OFF + RING * PAIR_BYTES, BIAS_ARR = BN * 4, BIAS_SET = 3 * BIAS_ARR;   // per tile parity: bias | bias2 row 0 | bias2 row 1
  static_assert(BIAS_OFF + 2 * BIAS_SET <= 160 * 1024, "LDS");
  static_assert(LEADP >= NPH && LEADP <= NPH + 1, "the W stream runs one chunk (+ at most one phase) ahead");
  constexpr bool GEGLU = EPI == 1, SLAB = EPI == 3;

  const int nwg = tiles_m * tiles_n;
  const int bz = blockIdx.z;
  const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = wid >> 2;                     // wave group: waves 0-3 (rows 0..127) / 4-7 (rows 128..255); SIMD partners are w, w + 4
  const int wm = wid >> 1, wn = wid & 1;        // 4 x 2 wave grid: 64 rows x BN / 2 columns per wave
  const int lm = lane & 15, lq = lane >> 4;

  auto decode = [&](int v, int& tm, int& tn) {   // XCD-aware virtual tile order (see gemm16.hip)
    const int q = nwg >> 3, r = nwg & 7, x = v & 7;
    const int t = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (v >> 3);
    if (pb <= 1) {
      tm = t / tiles_n;
      tn = t - tm * tiles_n;
    } else {
      const int gsz = pb * tiles_n, g = t / gsz, w = t - g * gsz;
      const int pbe = min(pb, tiles_m - g * pb);
      tn = w / pbe;
      tm = g * pb + (w - tn * pbe);
    }
  };

  // ---- LDS-DMA source addressing.  A: wave `wid` fills the 8-row pieces 4 wid + i of the tile's 256 rows (its own group's half).
  // W: of column pair q (W rows 32 q .. + 31 of the tile's first column half, then the same of the second) wave `wid` fills the 8 rows
  // (wid >> 2) 32 + 8 (wid & 3) .. of the pair's ring slot: one piece per wave and pair.
  const int srow = lane / CPR, spos = lane % CPR;
  const int ldw = ad.ksplit ? ad.ldw : K;
  const long kstart = ad.ksplit ? (long)bz * K : 0;
  const T* a0 = reinterpret_cast<const T*>(ad.src0) + (ad.ksplit ? (MODE == 0 ? kstart : 0) : (long)bz * ad.bs0);
  const T* a1 = ad.src1 ? reinterpret_cast<const T*>(ad.src1) + (ad.ksplit ? 0 : (long)bz * ad.bs1) : nullptr;
  const T* wbase = reinterpret_cast<const T*>(W) + (ad.ksplit ? kstart : (long)bz * bsw);
  const __amdgpu_buffer_rsrc_t rA0 = dma_rsrc(a0), rA1 = dma_rsrc(a1 ? a1 : a0), rW = dma_rsrc(wbase);
  const int c0sw = spos ^ (srow >> 1);
  const int wrow_t = grp * (BN / 2) + 8 * (wid & 3);      // this wave's row of pair 0 in the W TILE (pair q: + 32 q)
  const int wrow_s = grp * 32 + 8 * (wid & 3);            // ... and inside a ring slot
  int limA = 0, limW = 0;
  unsigned aoff[MODE == 0 ? 2 : GA];
  unsigned woff = 0;
  unsigned am0 = 0;
  int a_step = 0, w_pair_step = 0;
  int p_tap = 0, p_c = 0, a_soff = 0;
  bool a_second = false, a_fresh = true;
  auto setupA = [&](int tm) {
    const unsigned row = (unsigned)(tm * BM + wid * GA * RPD + srow);
    if (MODE == 0) {
      a_step = (int)(RPD * ad.ld0 * ESZ);
      limA = M - (int)row;
#pragma unroll
      for (int q = 0; q < 2; ++q) aoff[q] = row * (unsigned)(ad.ld0 * ESZ) + (unsigned)((c0sw ^ (4 * ((q + wid * GA) & 1))) << 4);
    } else {
      am0 = row;
    }
    p_tap = MODE == 1 ? (int)(kstart / (ad.C0 + ad.C1)) : 0;
    p_c = MODE == 1 ? (int)(kstart - (long)p_tap * (ad.C0 + ad.C1)) : 0;
    a_fresh = true;
  };
  auto setupW = [&](int tn) {
    const unsigned row = (unsigned)(tn * BN + wrow_t + srow);
    w_pair_step = 32 * ldw * ESZ;
    limW = N - (int)row;
    // swizzle of a slot row 8 k + srow: chunk ^ ((4 k + (srow >> 1)) & 7); k = (wid >> 2) 4 + (wid & 3) has the parity of wid
    woff = row * (unsigned)(ldw * ESZ) + (unsigned)((c0sw ^ (4 * (wid & 1))) << 4);
  };
  auto prepA = [&](int ch) {   // source offsets of the A pieces of chunk `ch` of the A stream's tile (chunks strictly in order)
    if (MODE == 0) {
      a_soff = ch * ROWB;
    } else {
      const int cin = ad.C0 + ad.C1;
      if (p_c == 0 || p_c == ad.C0 || a_fresh) {
        a_fresh = false;
        const int ky = p_tap / 3, kx = p_tap - ky * 3;
        const int vh = ad.up ? ad.IH * 2 : ad.IH, vw = ad.up ? ad.IW * 2 : ad.IW;
        const bool second = p_c >= ad.C0 && ad.C1 > 0;
        const unsigned cpb = (unsigned)(second ? ad.C1 : ad.C0) * ESZ;
        a_second = second;
        a_soff = (p_c - (second ? ad.C0 : 0)) * ESZ;
#pragma unroll
        for (int i = 0; i < GA; ++i) {
          const unsigned m = am0 + RPD * i;
          const unsigned cn = fastdiv(m, ad.fd_hw), rem = m - cn * (unsigned)(ad.OH * ad.OW);
          const unsigned oy = fastdiv(rem, ad.fd_ow), ox = rem - oy * (unsigned)ad.OW;
          const int iy = (int)oy * ad.stride + ky - ad.pad, ix = (int)ox * ad.stride + kx - ad.pad;
          const bool ok = (int)m < M && iy >= 0 && iy < vh && ix >= 0 && ix < vw;
          const int sy = ad.up ? iy >> 1 : iy, sx = ad.up ? ix >> 1 : ix;
          const unsigned off = (cn * (unsigned)(ad.IH * ad.IW) + (unsigned)(sy * ad.IW + sx)) * cpb + (unsigned)((c0sw ^ (4 * ((i + wid * GA) & 1))) << 4);
          aoff[i] = ok ? off : DMA_POISON;
        }
      } else {
        a_soff += ROWB;
      }
      p_c += BK;
      if (p_c == cin) { p_c = 0; ++p_tap; }
    }
  };
  // (vector offset, scalar offset, second source?) of A piece I of the chunk prepared by prepA(): kept apart from the issue so that the last
  // piece of a chunk can go out one slot later, after the stream has moved on to the next chunk / tile
  auto pieceA = [&](int I, unsigned& voff, int& soff, bool& second) __attribute__((always_inline)) {
    if (MODE == 0) { voff = RPD * I < limA ? aoff[I & 1] : DMA_POISON; soff = a_soff + I * a_step; second = false; }
    else { voff = aoff[MODE == 0 ? 0 : I]; soff = a_soff; second = a_second; }
  };
  auto putA = [&](int stage, int I, unsigned voff, int soff, bool second) __attribute__((always_inline)) {
    if (MODE == 1 && second) blds16(rA1, voff, soff, smem + stage * A_STAGE + (wid * GA + I) * 1024);
    else blds16(rA0, voff, soff, smem + stage * A_STAGE + (wid * GA + I) * 1024);
  };
  const bool has_bias = ep.bias != nullptr || ep.bias2 != nullptr;
  const int b2div = ep.bias2 ? ep.bias2_rows : 0x7fffffff;
  // the bias vectors of tile `v` into the LDS set of parity `par` (waves 0-3: three arrays of BN floats = a 16-byte piece of 256 floats and, at
  // BN = 320, a 4-byte piece of 64); never counted by the waits: see Sched
  auto issue_bias = [&](int tm, int tn, int par) {
    int mlast = tm * BM + BM - 1;
    if (mlast >= M) mlast = M - 1;
    const int r0 = (tm * BM) / b2div, r1 = mlast / b2div;
    const int arr = wid;                                  // waves 0, 1, 2: bias, bias2 row r0, bias2 row r1
    const float* src = arr == 0 ? ep.bias : (arr < 3 && ep.bias2) ? ep.bias2 + (long)(arr == 1 ? r0 : r1) * N : nullptr;
    if (src && arr < 3) {
      const int ln = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
      char* dst = smem + BIAS_OFF + par * BIAS_SET + arr * BIAS_ARR;
      const __amdgpu_buffer_rsrc_t rb = dma_rsrc(src);
      const int col = tn * BN + ln * 4;
      blds16(rb, col < N ? (unsigned)col * 4u : DMA_POISON, 0, dst);
      if (BN == 320) {
        const int col1 = tn * BN + 256 + ln;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, (__attribute__((address_space(3))) void*)(dst + 1024), 4, col1 < N ? col1 * 4 : (int)DMA_POISON, 0, 0, 0);
      }
    }
  };


  // ---- fragment read addressing (as gemm16.hip)
  const int sw = (lm >> 1) & 7;
  const int roff0 = lm * ROWB + ((lq ^ sw) << 4), roff1 = lm * ROWB + (((4 + lq) ^ sw) << 4);
  const int a_base = wm * 64 * ROWB;
  const int b_base = W_OFF + wn * 32 * ROWB;              // this wave's 32 rows inside a ring slot

  const int nch = K / BK;                            // >= 3 (host)
  const int G = gridDim.x;
  const int my_tiles = (nwg - (int)blockIdx.x + G - 1) / G;
  const int SL = nch + 1;                            // slots per tile
  const int total_local = my_tiles * SL;             // slots of one group

  // ---- tiles: the group's current one and the next of this workgroup (decoded once per tile, at the group's tile start)
  int tm = 0, tn = 0, tmN = 0, tnN = 0;
  decode(blockIdx.x, tm, tn);
  tmN = tm;
  tnN = tn;

  // ---- stream state.  W: chunk ichW of its tile; the pair inside the chunk is the phase's compile-time target.  A: chunk ichA, stage stA.
  int ichW = 0, wchunk = 0;                          // wchunk = ichW * ROWB
  int wdst = W_OFF + wrow_s * ROWB;                  // LDS address of this wave's piece in the ring slot of the next real W pair
  int ichA = 0, stA = 0;
  setupA(tm);
  setupW(tn);

  // ---- THE LOAD PART HOLDS NO BOOKKEEPING.  Everything a phase's load part issues is prepared one phase earlier, in the multiply part
  // (behind the MFMAs, where the wave has the SIMD's priority and the other group is busy with its own load part): LDS read bases, M0 values,
  // scalar and vector offsets of the phase's DMA pieces.  The load part is then ds_reads, one or two instructions per DMA piece, s_waitcnt,
  // barrier -- it is what the other group's 16 MFMAs (256 cycles) have to cover, and beside a wave that issues MFMAs at raised priority every
  // instruction of it costs about 8 cycles (ablations, profiles/r4/abl_gemm16s_r4.txt: with ~60 instructions in the load part the phase took
  // 2 x 860 cycles against gemm16_kernel's 2 x 445).
  int p_rd = 0;                                      // W fragment reads: ring slot base (LDS byte offset, without the lane part)
  const char* p_ra = smem;                           // A fragment reads (phase 0): stage base
  unsigned p_wv = DMA_POISON;                        // W piece: vector offset, scalar offset, LDS destination
  int p_wso = 0, p_wm0 = 0;
  unsigned p_av[2] = {DMA_POISON, DMA_POISON};       // A pieces of the phase
  int p_aso[2] = {0, 0}, p_am0[2] = {0, 0};
  bool p_asec[2] = {false, false};
  int rdst = b_base;                                 // ring slot base of the next W pair this group reads

  // slot state (always that of the slot whose phases run next; advanced in the last multiply part of the slot before)
  int vt = blockIdx.x;                               // the group's current tile
  int pos = -grp;                                    // the group's slot inside its tile: 0 .. nch - 1 main, nch = epilogue (-1: group 1's idle first slot)
  int tix = 0;                                       // tiles this group has started (bias parity)
  int p0 = 0, r0 = my_tiles;                         // group 0's slot inside its tile (the wall position) and the tiles it has left, this one included
  int stM = 0;                                       // A stage of the group's next main slot
  bool w1 = false, w2 = false, a_now = false, mainslot = false, epislot = false, prev_epi = false, next_tile = false, bias_now = false;
  bool full_tile = false, prev_full = false, fs = false, late_e0 = false, late_e1 = false;
  int row0 = 0, col0 = 0;
  auto slot_flags = [&]() __attribute__((always_inline)) {
    const bool active = pos >= 0 && tix < my_tiles;
    mainslot = active && pos < nch;
    epislot = active && pos == nch;
    // W: is group 0's slot after this wall slot / the one after that a main slot of a tile it has?
    int q = p0 + 1, t = r0;
    if (q == SL) { q = 0; --t; }
    w1 = t > 0 && q != nch;
    q = p0 + 2; t = r0;
    if (q >= SL) { q -= SL; --t; }
    w2 = t > 0 && q != nch;
    // A: the chunk this group multiplies two slots from now
    q = pos + 2; t = my_tiles - tix;
    if (q >= SL) { q -= SL; --t; }
    a_now = t > 0 && q != nch;
    next_tile = vt + G < nwg;
    prev_epi = mainslot && pos == 0 && tix > 0;
    bias_now = grp == 0 && has_bias && pos == 1 && next_tile && mainslot;
    fs = epislot ? full_tile : prev_full;            // do the stores of the epilogue inside the waits' window count?
    late_e0 = prev_epi && !fs;
    late_e1 = prev_epi && fs;
  };

  acc4 acc[4][NT];
  // a tile starts (the group's tile vt, coordinates tmN / tnN).  Scalar part: the next tile's coordinates (runs in the last multiply part of the
  // epilogue slot); register part: accumulators = bias + per-batch bias (+ residual), at the top of the tile's first slot (ONE site in the code:
  // with a second one hipcc no longer keeps the accumulators in place and spills 100-400 registers)
  auto tile_coords = [&]() __attribute__((always_inline)) {
    tm = tmN;
    tn = tnN;
    if (vt + G < nwg) decode(vt + G, tmN, tnN);
    row0 = tm * BM + wm * 64;
    col0 = tn * BN + wn * (BN / 2);
    prev_full = full_tile;
    full_tile = !ad.ksplit && tm * BM + BM <= M && tn * BN + BN <= N;
  };
  auto acc_init = [&]() __attribute__((always_inline)) {
    const acc4* lb = reinterpret_cast<const acc4*>(smem + BIAS_OFF + (tix & 1) * BIAS_SET) + wn * (BN / 8) + lq;
    const int b2r0 = (tm * BM) / b2div;
    bool second[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      int m = row0 + 16 * i + lm;
      if (m >= M) m = M - 1;
      second[i] = ep.bias2 && m / b2div != b2r0;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        acc4 b = (has_bias && ep.bias) ? lb[4 * j] : (acc4)(0.f);
        if (has_bias && ep.bias2) b += second[i] ? lb[2 * BIAS_ARR / 16 + 4 * j] : lb[BIAS_ARR / 16 + 4 * j];
        acc[i][j] = b;
      }
    }
  };

  // what phase P of the slot described by the slot state issues and reads (runs one phase ahead of it)
  auto prep = [&](auto Pc) __attribute__((always_inline)) {
    constexpr int P = decltype(Pc)::value;
    // fragment reads
    p_rd = rdst;
    {
      const int nx = rdst + PAIR_BYTES == b_base + RING * PAIR_BYTES ? b_base : rdst + PAIR_BYTES;
      rdst = mainslot ? nx : rdst;
    }
    if (P == 0) p_ra = smem + stM * A_STAGE + a_base;
    // the W pair group 0 reads LEADP wall phases after phase P
    {
      constexpr int TP = P + (LEADP - NPH), Q = TP < NPH ? TP : TP - NPH;   // LEADP = NPH + 1 (BN 256): phase P + 1 of the next slot, or phase 0 of the one after
      const bool real = TP < NPH ? w1 : w2;
      p_wv = (real && 32 * Q < limW) ? woff : DMA_POISON;   // (rows beyond N: zeros, never read from memory)
      p_wso = wchunk + Q * w_pair_step;
      p_wm0 = wdst;
      const int nx = wdst + PAIR_BYTES == W_OFF + wrow_s * ROWB + RING * PAIR_BYTES ? W_OFF + wrow_s * ROWB : wdst + PAIR_BYTES;
      wdst = real ? nx : wdst;
      if (Q == NPH - 1) {
        wchunk = real ? wchunk + ROWB : wchunk;
        ichW = real ? ichW + 1 : ichW;
        if (__builtin_expect(ichW == nch, 0)) { ichW = 0; wchunk = 0; setupW(tnN); }
      }
    }
    // the phase's A pieces of the chunk this group multiplies two slots after this one
    if (SC::a_n(P) > 0) {
      if (P == 1) {
        if (MODE == 0) prepA(ichA);
        else if (__builtin_expect(a_now, 1)) prepA(ichA);
      }
#pragma unroll
      for (int k = 0; k < SC::a_n(P); ++k) {
        const int I = SC::a_lo(P) + k;
        unsigned voff; int soff; bool second;
        pieceA(I, voff, soff, second);
        p_av[k] = a_now ? voff : DMA_POISON;
        p_aso[k] = soff;
        p_asec[k] = second;
        p_am0[k] = stA * A_STAGE + (wid * GA + I) * 1024;
      }
      if (P == NPH - 1) {
        stA = a_now ? stA ^ 1 : stA;
        ichA = a_now ? ichA + 1 : ichA;
        if (__builtin_expect(ichA == nch, 0)) { ichA = 0; setupA(tmN); }
      }
    }
  };
  // ---- prologue: bias of the first tile, the W pairs of the first LEADP wall phases, the A chunks of the group's first main slots (group 0:
  // two; group 1: one -- its second goes out in its idle first slot, like every later one), the first tile's residual, the first tile's start
  if (total_local > 0) {
    if (grp == 0 && has_bias) issue_bias(tm, tn, 0);
    w1 = w2 = a_now = true;
    for (int c = 0; c < (grp == 0 ? 2 : 1); ++c) {
      prepA(ichA);
#pragma unroll
      for (int i = 0; i < GA; ++i) {
        unsigned voff; int soff; bool second;
        pieceA(i, voff, soff, second);
        putA(stA, i, voff, soff, second);
      }
      stA ^= 1;
      ++ichA;                                          // (nch >= 3: no tile change here)
    }
#pragma unroll
    for (int q = 0; q < LEADP; ++q) {
      blds16(rW, 32 * (q % NPH) < limW ? woff : DMA_POISON, wchunk + (q % NPH) * w_pair_step, smem + wdst);
      wdst += PAIR_BYTES;
      if (q % NPH == NPH - 1) { wchunk += ROWB; ++ichW; }
    }
  }
  wait_vmcnt<0>();
  __builtin_amdgcn_s_barrier();
  if (total_local > 0) tile_coords();
  slot_flags();
  prep(std::integral_constant<int, 0>{});
  if (grp == 1) __builtin_amdgcn_s_barrier();        // group 1 runs one barrier behind group 0 from here on

  s16x8 fa[4][2], fb[2][2];
#if G16S_TRACE
  // stamps of waves 0 and 4 of workgroup 0 into the 10 KiB of LDS behind the bias sets (ds_write: nothing on vmcnt), from the second tile on:
  // five per phase -- phase start | issue done | s_waitcnt vmcnt passed | first barrier passed | MFMAs issued
  constexpr int TR_OFF = BIAS_OFF + 2 * BIAS_SET, TR_CAP = 640;
  const bool tr_on = trace && blockIdx.x == 0 && bz == 0 && (wid & 3) == 0;
  int tr_n = 0;
  bool tr_rec = false;
  auto stamp = [&]() __attribute__((always_inline)) {
    if (tr_on && tr_rec && tr_n < TR_CAP) {
      unsigned long long t;
      asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
      if (lane == 0) reinterpret_cast<unsigned long long*>(smem + TR_OFF)[grp * TR_CAP + tr_n] = t;
      ++tr_n;
    }
  };
#else
  auto stamp = [&]() __attribute__((always_inline)) {};
#endif

  for (int ws = 0; ws <= total_local; ++ws) {
#if G16S_TRACE
    tr_rec = ws >= SL + grp;
#endif
    if (mainslot && pos == 0) acc_init();
    // store addressing of the epilogue slot (derived late: nothing of it is live in the main slots)
    int lme = lm, lqe = lq;
    if (epislot) asm volatile("" : "+v"(lme), "+v"(lqe));
    const int cofs = 16 * (lqe & 1) + 8 * (lqe >> 1);

    auto epi_row = [&](int i) __attribute__((always_inline)) {
      int m = row0 + 16 * i + lme;
      if (dbg & 32) m = (int)blockIdx.x * 16 + (m & 15);   // every workgroup rewrites its own 16 rows: the stores stay in L2
      if (SLAB) {
        float* pbs = reinterpret_cast<float*>(ep.out) + (long)bz * ep.bso + col0 + 4 * lqe + (long)m * ep.ldo;
#pragma unroll
        for (int j = 0; j < NT; ++j)
          if (m < M && col0 + 16 * j + 4 * lqe < N) *reinterpret_cast<acc4*>(pbs + 16 * j) = acc[i][j];
        return;
      }
      const int ncol = N - (col0 + cofs);
      T* orow = reinterpret_cast<T*>(ep.out) + (long)bz * ep.bso + (GEGLU ? (col0 >> 1) + cofs : col0 + cofs) + (long)m * ep.ldo;
#pragma unroll
      for (int jp = 0; jp < NPAIR; ++jp) {
        if (GEGLU && (jp & 1)) continue;
        acc4 x = acc[i][2 * jp], y = acc[i][2 * jp + 1];
        if (GEGLU) {
          const acc4 gx = acc[i][(2 * jp + 2) % NT], gy = acc[i][(2 * jp + 3) % NT];
          const f32x2 g0 = gelu_erf_f2((f32x2){gx[0], gx[1]}), g1 = gelu_erf_f2((f32x2){gx[2], gx[3]});
          const f32x2 g2 = gelu_erf_f2((f32x2){gy[0], gy[1]}), g3 = gelu_erf_f2((f32x2){gy[2], gy[3]});
          x[0] *= g0[0]; x[1] *= g0[1]; x[2] *= g1[0]; x[3] *= g1[1];
          y[0] *= g2[0]; y[1] *= g2[1]; y[2] *= g3[0]; y[3] *= g3[1];
        }
        const auto s01 = __builtin_amdgcn_permlane16_swap(pack_bf16x2(x[0], x[1]), pack_bf16x2(y[0], y[1]), false, false);
        const auto s23 = __builtin_amdgcn_permlane16_swap(pack_bf16x2(x[2], x[3]), pack_bf16x2(y[2], y[3]), false, false);
        const u32x4 ov = (u32x4){s01[0], s23[0], s01[1], s23[1]};
        if (dbg & 64) { asm volatile("" ::"v"(ov)); }
        else if (m < M && 32 * jp < ncol) *reinterpret_cast<u32x4*>(orow + (GEGLU ? 16 * jp : 32 * jp)) = ov;
      }
    };

    auto phase = [&](auto Pc, auto Kc) __attribute__((always_inline)) {
      constexpr int P = decltype(Pc)::value, KIND = decltype(Kc)::value;   // KIND: 0 main slot, 1 epilogue slot, 2 idle (outside this group's range)
      constexpr bool CE = KIND == 1;                                       // this slot is the group's epilogue slot (prev_epi: the previous one was)
      // ======== load part: what prep<P> laid out
      stamp();
      if (KIND == 0 && !(dbg & 4)) {
        const char* pw = smem + p_rd;
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          fb[t][0] = *reinterpret_cast<const s16x8*>(pw + 16 * t * ROWB + roff0);
          fb[t][1] = *reinterpret_cast<const s16x8*>(pw + 16 * t * ROWB + roff1);
        }
        if (P == 0) {
#pragma unroll
          for (int t = 0; t < 4; ++t) {
            fa[t][0] = *reinterpret_cast<const s16x8*>(p_ra + 16 * t * ROWB + roff0);
            fa[t][1] = *reinterpret_cast<const s16x8*>(p_ra + 16 * t * ROWB + roff1);
          }
        }
      }
      // ---- LDS-DMA, the same operations in every slot: one W piece, the phase's A pieces
      if (!(dbg & 1)) {
        blds16(rW, p_wv, p_wso, smem + p_wm0);
#pragma unroll
        for (int k = 0; k < SC::a_n(P); ++k) {
          if (MODE == 1 && p_asec[k]) blds16(rA1, p_av[k], p_aso[k], smem + p_am0[k]);
          else blds16(rA0, p_av[k], p_aso[k], smem + p_am0[k]);
        }
      }
      // the next tile's bias vectors: group 0, in its second slot of a tile (every wave has started the tile by then: group 1 one slot ago)
      if (KIND == 0 && P == 1 && __builtin_expect(bias_now, 0)) issue_bias(tmN, tnN, (tix + 1) & 1);
      // ---- the epilogue's row tiles of this phase
      if (CE && !(dbg & 16)) {
#pragma unroll
        for (int i = SC::rows_lo(P); i < SC::rows_lo(P) + SC::rows_n(P); ++i) epi_row(i);
      }
      // ---- what must have landed in front of the next barriers (Sched::waitn)
      stamp();
      {
        // (one body per slot kind: the variants differ in this immediate only, chosen by wave-uniform branches around the s_waitcnt)
        constexpr int n0 = SC::waitn(P, false, CE, false), n1 = SC::waitn(P, false, CE, true);
        constexpr int e0 = SC::waitn(P, true, false, false), e1 = SC::waitn(P, true, false, true);
        static_assert(n0 >= 0 && n1 >= n0 && n1 < 64 && e0 >= 0 && e1 >= e0 && e1 < 64, "vmcnt");
        if (dbg & 2) {
        } else if (KIND == 0) {                           // (the slot behind the group's epilogue slot: once per tile, laid out as unlikely)
          if (e1 != n0 && __builtin_expect(late_e1, 0)) wait_vmcnt<e1>();
          else if (e0 != n0 && __builtin_expect(late_e0, 0)) wait_vmcnt<e0>();
          else wait_vmcnt<n0>();
        } else {
          if (n0 == n1 || !fs) wait_vmcnt<n0>();
          else wait_vmcnt<n1>();
        }
      }
      stamp();
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
      stamp();
      // ======== multiply part; behind the MFMAs: the next phase's issue laid out, after a slot's last phase the next slot's state (and, behind
      // the epilogue slot, the next tile's start -- beside the other group's load part, not in front of this group's)
      if (KIND == 0 && !(dbg & 8)) {
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
          for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int u = 0; u < 2; ++u) acc[t][2 * P + u] = mma16s(fb[u][ks], fa[t][ks], acc[t][2 * P + u]);
      }
      if (P < NPH - 1) {
        prep(std::integral_constant<int, (P + 1) % NPH>{});
      } else {
        if (KIND == 0) stM ^= 1;
        if (++p0 == SL) { p0 = 0; --r0; }
        if (KIND == 2) {
          if (pos < 0) pos = 0;
        } else if (++pos == SL) { pos = 0; vt += G; ++tix; }
        if (KIND == 1 && tix < my_tiles) tile_coords();
        slot_flags();
        prep(std::integral_constant<int, 0>{});
      }
      if (KIND == 0 && !(dbg & 8)) __builtin_amdgcn_s_setprio(0);
      stamp();
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
    };
    auto run = [&](auto Kc) __attribute__((always_inline)) {
      [&]<int... I>(std::integer_sequence<int, I...>) { (phase(std::integral_constant<int, I>{}, Kc), ...); }(std::make_integer_sequence<int, NPH>{});
    };
    if (mainslot) run(std::integral_constant<int, 0>{});
    else if (epislot) run(std::integral_constant<int, 1>{});
    else run(std::integral_constant<int, 2>{});
  }
  if (grp == 0) __builtin_amdgcn_s_barrier();        // group 0 meets group 1's last barrier
#if G16S_TRACE
  if (tr_on) {
    for (int i = lane; i < TR_CAP; i += 64)
      trace[grp * TR_CAP + i] = i < tr_n ? reinterpret_cast<unsigned long long*>(smem + TR_OFF)[grp * TR_CAP + i] : 0ull;
  }
#endif
}

#if G16S_TRACE
unsigned long long* g_g16s_trace = nullptr;
int g_g16s_dbg = 0;
#endif

template <int MODE, int BN>
int launch16s(const ADesc& ad, const void* W, long bsw, const Epi& ep, int M, int N, int K, int batch, hipStream_t s, int pb_tune) {
  constexpr int BM = 256;
  const int tiles_m = (M + BM - 1) / BM, tiles_n = (N + BN - 1) / BN;
  size_t lds = (size_t)2 * BM * 128 + (size_t)(BN == 256 ? 10 : 11) * 8192 + 2 * 3 * BN * 4;
#if G16S_TRACE
  if (BN == 256) lds += 2 * 640 * 8;   // (the 320-column tile leaves no room for the stamps: not traced)
#endif
  static int resident = 0;
  if (!resident) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) {
      mmgt_set_error("gemm16s: device query failed");
      return 2;
    }
    resident = prop.multiProcessorCount;
  }
  long gx = (resident + batch - 1) / batch;
  gx = (gx + 7) / 8 * 8;
  if (gx > (long)tiles_m * tiles_n) gx = (long)tiles_m * tiles_n;
  dim3 grid((unsigned)gx, 1, batch);
  const int pb = pb_tune >= 0 ? pb_tune : (MODE == 0 && tiles_n >= 8 && tiles_m >= 8) ? 8 : 1;
  auto go = [&](auto kern) {
    static bool attr = false;
    if (!attr) {
      if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
        mmgt_set_error("gemm16s: cannot reserve %zu bytes of LDS", lds);
        return 2;
      }
      attr = true;
    }
    hipLaunchKernelGGL(kern, grid, dim3(512), lds, s, ad, reinterpret_cast<const char*>(W), bsw, ep, M, N, K, tiles_m, tiles_n, pb
#if G16S_TRACE
                       , BN == 256 ? g_g16s_trace : nullptr, g_g16s_dbg
#endif
    );
    MMGT_LAUNCH_CHECK();
    return 0;
  };
  if (ad.ksplit) return go(gemm16s_kernel<MODE, BN, 3>);
  if (ep.act == 1) {
    if constexpr (BN == 256) return go(gemm16s_kernel<MODE, 256, 1>);
    mmgt_set_error("gemm16s: GEGLU needs the 256-column tile");
    return 1;
  }
  if (ep.residual) {
    mmgt_set_error("gemm16s: no residual variant");
    return 1;
  }
  return go(gemm16s_kernel<MODE, BN, 0>);
}

}  // namespace

#if G16S_TRACE
extern "C" void mmgt_gemm16s_set_trace(void* p) { g_g16s_trace = reinterpret_cast<unsigned long long*>(p); }   // 2 x 640 stamps; nullptr: off
extern "C" void mmgt_gemm16s_set_dbg(int v) { g_g16s_dbg = v; }
#endif

// Entry for gemm16.hip's launcher: same preconditions as mmgt_gemm16_launch, and no row scale / alpha / post-scale bias, no GEGLU + residual.
int mmgt_gemm16s_launch(int mode, int bn, const void* adp, const void* W, long bsw, const void* epp, int M, int N, int K, int batch, void* stream,
                        int pb_tune) {
  const ADesc& ad = *reinterpret_cast<const ADesc*>(adp);
  const Epi& ep = *reinterpret_cast<const Epi*>(epp);
  hipStream_t s = (hipStream_t)stream;
  if (bn == 320) return mode == 0 ? launch16s<0, 320>(ad, W, bsw, ep, M, N, K, batch, s, pb_tune) : launch16s<1, 320>(ad, W, bsw, ep, M, N, K, batch, s, pb_tune);
  return mode == 0 ? launch16s<0, 256>(ad, W, bsw, ep, M, N, K, batch, s, pb_tune) : launch16s<1, 256>(ad, W, bsw, ep, M, N, K, batch, s, pb_tune);
}
