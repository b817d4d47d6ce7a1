// bf16 GEMM / implicit-GEMM conv3x3 core with ONE wave per SIMD: 256 x 256 tile, 4 waves, a 128 x 128 wave tile whose 64 accumulator tiles
// (v_mfma_f32_16x16x32_bf16) live in the 256 accumulation registers (gfx950).
//
//   out[M, N] = epilogue( A[M, K] * W[N, K]^T )      same operands, LDS image, DMA addressing and epilogue semantics as gemm16.hip
//
// Why a third core.  gemm16_kernel keeps two wave groups in ping-pong: 2 barriers per 16 MFMAs, load parts that the partner's MFMAs must
// cover, 24 KiB of LDS fragment reads per 64 MFMAs -- its matrix pipe is 41-55 % busy and nothing tried on that structure in round 4 moved
// it (gemm16s.hip's header).  tools/micro/lonewave.hip measured the alternative: ONE wave per SIMD that issues the 128 MFMAs of a 128 x 128
// wave tile per 64-deep chunk back to back, its 32 fragment reads, 16 LDS-DMA pieces and ONE barrier dealt out between them, keeps the pipe
// 89-98 % busy -- provided the accumulators stay in place, which hipcc only does when the MFMA is an inline-asm statement with a TIED
// accumulation-register operand ("+a"): through the builtin it rotates the 64 tiles through different AGPR ranges, copies them with
// v_accvgpr_mov and bunches the reads (59 % in the same micro-benchmark).  A 128 x 128 wave tile also reads 32 KiB of fragments per 128 MFMAs
// instead of 48.
//
// Schedule (per workgroup: a continuous stream of 64-deep chunks over its tiles, stage = chunk & 1, 64 KiB per stage: 256 A rows | 256 W rows of
// 128 B, XOR-swizzled as in gemm16.hip).  A chunk is two k-steps of 64 MFMAs; a k-step's fragments are read during the k-step before it:
//   k-step 0 of chunk g:  64 MFMAs on the k = 0..31 fragments | the 16 reads of the chunk's k = 32..63 fragments dealt out one per 4 MFMAs
//   s_waitcnt vmcnt(0) (chunk g + 1 has landed: issued a whole chunk ago), lgkmcnt(0), s_barrier   -- the ONE barrier of the chunk: behind it
//       every wave's reads of stage g & 1 are complete and chunk g + 1 is visible to every wave
//   k-step 1 of chunk g:  64 MFMAs | the 16 reads of chunk g + 1's k = 0..31 fragments (other stage) | the 16 DMA pieces of chunk g + 2
//       into the stage chunk g has just finished reading
// so the matrix pipe never waits for a fragment at a chunk or tile boundary, and the pieces issued in a tile's last k-step (the next
// tile's second chunk) are in flight under its epilogue.  The tile's bias vectors are fetched one tile ahead (k-step 1 of a tile's
// first chunk) and the accumulators start from them, as in gemm16.hip.
// Tile: 256 x 256 only (a 320-column tile would need 320 accumulation registers): N = 256 k shapes, and N = 640 / 960 / 1920 at 83-94 %.
#include <type_traits>

#include "common.h"
#include "gemm_common.h"
#include "mmgt_hip.h"

namespace {

typedef __attribute__((ext_vector_type(4))) float acc4;
#ifndef G16V_ABL
#define G16V_ABL 0   // timing ablations (make ablv; results wrong): 1 no LDS-DMA in the loop, 2 no vmcnt wait, 4 no barrier, 8 no per-row lgkmcnt waits, 16 every DMA piece with the poison offset (issued, reads zeros, moves no memory), 32 / 64 the A / the W pieces only
#endif

// D += Wfrag x Afrag with the accumulator tile tied to ONE accumulation-register quad for the whole kernel (see the header)
// (AGPR = true: accumulation-register file; false: the vector file.  hipcc cannot place 64 tiles in exactly 256 accumulation registers -- it
// then time-shares one quad between several tiles through scratch, and its copies read MFMA results without the wait states an inline-asm MFMA
// hides from it: wrong sums -- so the wave's last row tile lives in the vector file: 224 + 32.)
template <bool AGPR>
__device__ __forceinline__ void mma16v(acc4& c, s16x8 a, s16x8 b) {
  if constexpr (AGPR) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b));
  else asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b));
}
template <bool AGPR>
__device__ __forceinline__ void pin_acc(acc4& c) {
  if constexpr (AGPR) asm volatile("" : "+a"(c));
  else asm volatile("" : "+v"(c));
}

// EPI (ONE epilogue per instantiation: with all six in one kernel hipcc's allocation of the loop suffers -- the loop-invariant DMA offsets were
// spilled and reloaded per piece, the buffer descriptors moved to vector registers and every DMA wrapped in a waterfall loop):
// 0 plain, 1 GEGLU, 2 + residual, 3 GEGLU + residual, 4 row scale / alpha / post-scale bias, 5 the same + residual, 6 split-K fp32 slabs
template <int MODE, int EPI>
__global__ __launch_bounds__(256, 1) void gemm16v_kernel(ADesc ad, const char* __restrict__ W, long bsw, Epi ep, int M, int N, int K, int tiles_m,
                                                         int tiles_n, int pb) {
  extern __shared__ __attribute__((aligned(1024))) char smem[];
  typedef bf16_t T;
  constexpr int ESZ = 2, BM = 256, BN = 256, ROWB = 128, BK = 64, NW = 4;
  constexpr int A_BYTES = BM * ROWB, B_BYTES = BN * ROWB, STAGE_BYTES = A_BYTES + B_BYTES;
  constexpr int CPR = 8, RPD = 8;               // 16-byte chunks per row; rows per 1-KiB DMA piece
  constexpr int GA = BM / RPD / NW, GB = BN / RPD / NW;   // pieces per wave and chunk: 8 x A + 8 x W
  constexpr int RT = 8, NT = 8, RTA = 7;        // accumulator tiles per wave: 8 row tiles x 8 column tiles of 16 x 16; row tiles [0, RTA) in accumulation registers
  constexpr int BIAS_OFF = 2 * STAGE_BYTES, BIAS_ARR = 2048, NBP = 1;   // bias | bias2 row 0 | bias2 row 1 | post-scale bias
  const int nwg = tiles_m * tiles_n;
  const int bz = blockIdx.z;
  const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wid >> 1, wn = wid & 1;        // 2 x 2 wave grid: 128 rows x 128 columns per wave
  const int lm = lane & 15, lq = lane >> 4;

  auto decode = [&](int v, int& tm, int& tn) __attribute__((always_inline)) {
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

  // ---- LDS-DMA source addressing (as gemm.hip): wave `wid` fills the 8-row pieces g = wid * GA + i of each operand
  const int srow = lane / CPR, spos = lane % CPR;
  // split-K (ad.ksplit): slice bz of the reduction starts K * bz elements into every W row and, dense, into every A row; the conv
  // gather starts at that (tap, channel) position instead (setupA)
  const int ldw = ad.ksplit ? ad.ldw : K;
  const long kstart = ad.ksplit ? (long)bz * K : 0;
  const T* a0 = reinterpret_cast<const T*>(ad.src0) + (ad.ksplit ? (MODE == 0 ? kstart : 0) : (long)bz * ad.bs0);
  const T* a1 = ad.src1 ? reinterpret_cast<const T*>(ad.src1) + (ad.ksplit ? 0 : (long)bz * ad.bs1) : nullptr;
  const T* wbase = reinterpret_cast<const T*>(W) + (ad.ksplit ? kstart : (long)bz * bsw);
  // The pieces of a wave are 8 rows apart, which goes into the SCALAR offset of the DMA, so the per-lane state of a stream
  // is one offset per piece parity (the swizzled 16-byte chunk of piece i is c0 ^ 4 ((wid G + i) & 1): row = 8 (wid G + i) +
  // srow, so (row >> 1) & 7 = (4 (wid G + i) + (srow >> 1)) & 7 with srow >> 1 in 0..3) plus the number of valid rows from the lane's first row:
  // pieces beyond M / N get the poison offset in their VECTOR offset (the part the hardware range-checks) and read as zeros.
  // (descriptor bases 3 KiB in FRONT of the operands: a piece's immediate offset 1024 k, k = 0..3, is taken back in the scalar offset, which must
  // not go negative -- offsets are zero-extended)
  constexpr int RSRC_BACK = 3072;
  const __amdgpu_buffer_rsrc_t rA0 = dma_rsrc(reinterpret_cast<const char*>(a0) - RSRC_BACK),
                               rA1 = dma_rsrc(reinterpret_cast<const char*>(a1 ? a1 : a0) - RSRC_BACK), rW = dma_rsrc(reinterpret_cast<const char*>(wbase) - RSRC_BACK);
  const int c0sw = spos ^ (srow >> 1);
  // ONE vector offset per piece (row part + swizzled 16-byte chunk; rows beyond M / N: poison), so that a piece in the main loop is one
  // instruction: a lone wave issues an instruction every ~6 cycles and has ~190 slots per chunk beside its 128 MFMAs -- with the per-piece
  // compare / select / scalar-offset arithmetic of gemm16.hip's form (9 instructions per piece) the 16 pieces alone took 35 % of the kernel
  // (profiles/r4/abl_gemm16v_r4.txt).  conv: aoff = the current tap's pixel of every piece (zero padding = poison).
  unsigned aoff[GA];
  unsigned woff[GB];
  unsigned am0 = 0;                    // conv: output row of piece 0 (piece i: + 8 i)
  int p_tap = 0, p_c = 0, a_soff = 0;
  bool a_second = false, a_fresh = true;   // a_fresh: the first chunk of a tile computes its gather offsets wherever the slice starts
  // The A and the W stream run at different distances ahead of the MFMAs (see the schedule above), so each keeps its own
  // position: tile, chunk inside the tile, chunks issued so far.
  auto setupA = [&](int tm) __attribute__((always_inline)) {
    const unsigned row = (unsigned)(tm * BM + wid * GA * RPD + srow);
    if (MODE == 0) {
#pragma unroll
      for (int i = 0; i < GA; ++i) {
        const unsigned r = row + RPD * i;
        aoff[i] = (int)r < M ? r * (unsigned)(ad.ld0 * ESZ) + (unsigned)((c0sw ^ (4 * ((i + wid * GA) & 1))) << 4) : DMA_POISON;
      }
    } else {
      am0 = row;
    }
    p_tap = MODE == 1 ? (int)(kstart / (ad.C0 + ad.C1)) : 0;
    p_c = MODE == 1 ? (int)(kstart - (long)p_tap * (ad.C0 + ad.C1)) : 0;
    a_fresh = true;
  };
  auto setupW = [&](int tn) __attribute__((always_inline)) {
    const unsigned row = (unsigned)(tn * BN + wid * GB * RPD + srow);
#pragma unroll
    for (int i = 0; i < GB; ++i) {
      const unsigned r = row + RPD * i;
      woff[i] = (int)r < N ? r * (unsigned)(ldw * ESZ) + (unsigned)((c0sw ^ (4 * ((i + wid * GB) & 1))) << 4) : DMA_POISON;
    }
  };
  auto prepA = [&](int ch) __attribute__((always_inline)) {   // source offsets of the A pieces of chunk `ch` of the A stream's tile (chunks come strictly in order)
    if (MODE == 0) {
      a_soff = ch * ROWB;
    } else {
      const int cin = ad.C0 + ad.C1;
      if (p_c == 0 || p_c == ad.C0 || a_fresh) {
        a_fresh = false;
        const int ky = p_tap / 3, kx = p_tap - ky * 3;
        const int vh = ad.up ? ad.IH * 2 : ad.IH, vw = ad.up ? ad.IW * 2 : ad.IW;
        const bool second = p_c >= ad.C0 && ad.C1 > 0;
        const unsigned cpb = (unsigned)(second ? ad.C1 : ad.C0) * ESZ;   // bytes per pixel of the source tensor (< 2 GiB in all: host check)
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
  // piece P of a chunk: 0 .. GA - 1 = A, GA .. GA + GB - 1 = W.  Four consecutive pieces share one M0 (LDS base of the group; the
  // instruction's immediate offset 1024 k moves both the LDS and the memory address, so the scalar offset takes it back).
  auto issueP = [&](int stage, int soffA, int soffW, auto Pc) __attribute__((always_inline)) {
    constexpr int P = decltype(Pc)::value, I = P < GA ? P : P - GA, k = I & 3;
    char* base = smem + stage * STAGE_BYTES + (P < GA ? 0 : A_BYTES) + (wid * GA + (I & ~3)) * 1024;
    if constexpr (P < GA) {
      if (MODE == 1 && a_second) __builtin_amdgcn_raw_ptr_buffer_load_lds(rA1, (__attribute__((address_space(3))) void*)base, 16, (int)((G16V_ABL & (16 | 32)) ? DMA_POISON : aoff[I]), soffA + (RSRC_BACK - 1024 * k), 1024 * k, 0);
      else __builtin_amdgcn_raw_ptr_buffer_load_lds(rA0, (__attribute__((address_space(3))) void*)base, 16, (int)((G16V_ABL & (16 | 32)) ? DMA_POISON : aoff[I]), soffA + (RSRC_BACK - 1024 * k), 1024 * k, 0);
    } else {
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rW, (__attribute__((address_space(3))) void*)base, 16, (int)((G16V_ABL & (16 | 64)) ? DMA_POISON : woff[I]), soffW + (RSRC_BACK - 1024 * k), 1024 * k, 0);
    }
  };
  // Bias.  The accumulators of a tile START from bias[n] + bias2[batch row of m][n] instead of zero, read from an LDS copy of the
  // tile's bias vectors (bias | the at most two bias2 rows a tile touches: 512 floats each).  The copy of the NEXT tile is
  // fetched by DMA pieces of 256 floats that wave group 0 issues in phase 1 of a tile's first chunk: by then every wave has
  // initialised its accumulators from the current copy (group 1 has arrived at its phase-0 barrier), and the pieces are older
  // than the chunk's counted A pieces, so they have landed behind that chunk's last barrier.  (Columns >= N read as zeros.)
  const bool has_bias = ep.bias != nullptr || ep.bias2 != nullptr;
  const int b2div = ep.bias2 ? ep.bias2_rows : 0x7fffffff;
  // A fourth vector, the post-scale bias (added after the row scale: MM-HAA's merged out-projections), is consumed at the END of
  // a tile, so it is the CURRENT tile's (`vc`) that goes out at the same point.
  auto issue_bias = [&](int v, int vc) __attribute__((always_inline)) {
    int tm = 0, tn = 0, tnc;
    if (v >= 0) decode(v, tm, tn);
    { int tmc; decode(vc, tmc, tnc); }
    int mlast = tm * BM + BM - 1;
    if (mlast >= M) mlast = M - 1;
    const int r0 = (tm * BM) / b2div, r1 = mlast / b2div;
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int q = wid + 4 * u;
      if (q < 4 * NBP) {
        const int arr = q / NBP, pc = q - arr * NBP;
        const float* src = arr == 3 ? ep.bias_post : v < 0 ? nullptr : arr == 0 ? ep.bias : ep.bias2 ? ep.bias2 + (long)(arr == 1 ? r0 : r1) * N : nullptr;
        if (arr == 3) tn = tnc;
        if (src) {
          const int ln = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));   // lane id afresh: nothing kept live (or spilled) for this
          const int col = tn * BN + pc * 256 + ln * 4;
          blds16(dma_rsrc(src), col < N ? (unsigned)col * 4u : DMA_POISON, 0, smem + BIAS_OFF + arr * BIAS_ARR + pc * 1024);
        }
      }
    }
  };

  // ---- fragment read addressing (as gemm16.hip): lane (lm, lq) reads row base + 16 t + lm, 16-byte chunk (4 ks + lq) ^ swizzle(lm)
  const int sw = (lm >> 1) & 7;
  const int roff0 = lm * ROWB + ((lq ^ sw) << 4), roff1 = lm * ROWB + (((4 + lq) ^ sw) << 4);
  const int a_base = wm * 128 * ROWB;                      // this wave's 128 A rows
  const int b_base = A_BYTES + wn * 128 * ROWB;            // this wave's 128 W rows

  const int nchunks = K / BK;
  const int G = gridDim.x;
  const int my_tiles = (nwg - (int)blockIdx.x + G - 1) / G;
  const int total = my_tiles * nchunks;
  // ONE stream position for both operands (they run the same distance ahead): tile, chunk inside the tile, chunks issued so far
  int vtS = blockIdx.x, ichS = 0, giS = 0;
  {
    int tm, tn;
    decode(vtS, tm, tn);
    setupA(tm);
    setupW(tn);
  }
  auto advanceS = [&]() __attribute__((always_inline)) {
    ++giS;
    if (__builtin_expect(++ichS == nchunks, 0)) {
      ichS = 0;
      vtS += G;
      if (vtS < nwg) {
        int tm, tn;
        decode(vtS, tm, tn);
        setupA(tm);
        setupW(tn);
      }
    }
  };
  auto for_n = [&](auto Nc, auto&& f) __attribute__((always_inline)) {
    constexpr int N_ = decltype(Nc)::value;
    [&]<int... I>(std::integer_sequence<int, I...>) { (f(std::integral_constant<int, I>{}), ...); }(std::make_integer_sequence<int, N_>{});
  };
  using std::integral_constant;
  // ---- prologue: chunks 0 and 1 whole; the k = 0..31 fragments of chunk 0
  if (total > 0) {
    if (has_bias) issue_bias(blockIdx.x, blockIdx.x);
    for (int c = 0; c < 2 && giS < total; ++c) {
      prepA(ichS);
      const int stage = giS & 1, soA = a_soff, soW = ichS * ROWB;
      for_n(integral_constant<int, GA + GB>{}, [&](auto p) { issueP(stage, soA, soW, p); });
      advanceS();
    }
  }
  wait_vmcnt<0>();
  __builtin_amdgcn_s_barrier();

  // Fragments of k = 0..31 (f0*) / k = 32..63 (f1*): [row tile] of A, [column tile] of W.  Read by inline-asm ds_read_b128 into TIED registers
  // ("+v": through the builtin load hipcc gives every read a fresh register and copies 32 registers back at the loop head) with hand-counted
  // waits: a k-step's 16 reads go out during the k-step before it in the order W 0..7, A 0..7; LDS returns in order, so row tile i of the next
  // k-step may start behind s_waitcnt lgkmcnt(7 - i).
  s16x8 f0a[RT], f0b[NT], f1a[RT], f1b[NT];
#pragma unroll
  for (int q = 0; q < RT; ++q) { f0a[q] = (s16x8)((short)0); f0b[q] = (s16x8)((short)0); f1a[q] = (s16x8)((short)0); f1b[q] = (s16x8)((short)0); }
  const unsigned ad_a0 = (unsigned)(a_base + roff0), ad_a1 = (unsigned)(a_base + roff1), ad_b0 = (unsigned)(b_base + roff0), ad_b1 = (unsigned)(b_base + roff1);
  auto lds_read = [&](s16x8& dst, unsigned addr, auto Oc) __attribute__((always_inline)) {
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "+v"(dst) : "v"(addr), "n"(decltype(Oc)::value));
  };
  // read r = 0 .. 15 of a k-step's fragment set: W column tile r (r < 8), then A row tile r - 8
  auto read_frag = [&](unsigned addr_a, unsigned addr_b, auto Rc, s16x8 (&fa)[RT], s16x8 (&fb)[NT]) __attribute__((always_inline)) {
    constexpr int r = decltype(Rc)::value;
    if constexpr (r < NT) lds_read(fb[r], addr_b, std::integral_constant<int, 16 * r * ROWB>{});
    else lds_read(fa[r - NT], addr_a, std::integral_constant<int, 16 * (r - NT) * ROWB>{});
  };
  for_n(integral_constant<int, RT + NT>{}, [&](auto r) { read_frag(ad_a0, ad_b0, r, f0a, f0b); });   // chunk 0 (stage 0), k = 0..31

  int g = 0;                                         // chunks multiplied so far (stage = g & 1)
  acc4 acc[RT][NT];
  for (int vt = blockIdx.x; vt < nwg; vt += G) {
    int tm, tn;
    decode(vt, tm, tn);
    const int row0 = tm * BM + wm * 128, col0 = tn * BN + wn * 128;
    // ---- accumulators = bias + per-batch bias (fp32), from the LDS copy fetched one tile ahead
    if (has_bias) {
      const acc4* lb = reinterpret_cast<const acc4*>(smem + BIAS_OFF) + wn * (128 / 4) + lq;   // this lane's columns 16 j + 4 lq + r
      const int b2r0 = (tm * BM) / b2div;
      bool second[RT];
#pragma unroll
      for (int i = 0; i < RT; ++i) {
        int m = row0 + 16 * i + lm;
        if (m >= M) m = M - 1;
        second[i] = ep.bias2 && m / b2div != b2r0;
      }
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        const acc4 b = ep.bias ? lb[4 * j] : (acc4)(0.f);
        if (ep.bias2) {
          const acc4 r0v = lb[BIAS_ARR / 16 + 4 * j], r1v = lb[2 * BIAS_ARR / 16 + 4 * j];
#pragma unroll
          for (int i = 0; i < RT; ++i) { acc[i][j] = b + (second[i] ? r1v : r0v); if (i < RTA) pin_acc<true>(acc[i][j]); else pin_acc<false>(acc[i][j]); }
        } else {
#pragma unroll
          for (int i = 0; i < RT; ++i) { acc[i][j] = b; if (i < RTA) pin_acc<true>(acc[i][j]); else pin_acc<false>(acc[i][j]); }
        }
      }
    } else {
#pragma unroll
      for (int i = 0; i < RT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) { acc[i][j] = (acc4)(0.f); if (i < RTA) pin_acc<true>(acc[i][j]); else pin_acc<false>(acc[i][j]); }   // (straight into the accumulation registers: 256 start values do not fit beside the fragments)
    }

    for (int ch = 0; ch < nchunks; ++ch) {
      const unsigned so = (unsigned)((g & 1) * STAGE_BYTES), sn = (unsigned)(((g & 1) ^ 1) * STAGE_BYTES);
      // ---- k-step 0: this chunk's k = 32..63 fragments are read between the MFMAs
      {
        const unsigned ra = ad_a1 + so, rb = ad_b1 + so;
        for_n(integral_constant<int, RT * NT>{}, [&](auto Xc) __attribute__((always_inline)) {
          constexpr int i = decltype(Xc)::value / NT, j = decltype(Xc)::value % NT;
          if constexpr (j == 0 && !(G16V_ABL & 8)) asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(i == 0 ? RT - 1 : RT - 1 - i + 2 * i));   // W 0..7 and A 0..i of THIS k-step's set have landed (2 i reads of the next set are younger)
          mma16v<(i < RTA)>(acc[i][j], f0b[j], f0a[i]);
          if constexpr ((j & 3) == 3) read_frag(ra, rb, integral_constant<int, 2 * i + (j >> 2)>{}, f1a, f1b);
        });
      }
      // ---- the chunk's barrier
      if (!(G16V_ABL & 2)) wait_vmcnt<0>();
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      if (!(G16V_ABL & 4)) __builtin_amdgcn_s_barrier();
      // ---- k-step 1: the next chunk's k = 0..31 fragments (other stage) and the 16 DMA pieces of the chunk after it (into this stage)
      const bool more = giS < total;
      const int stS = giS & 1;                       // (== g & 1: the stream is two chunks ahead)
      if (more) prepA(ichS);
      const int soA = a_soff, soW = ichS * ROWB;
      if (ch == 0 && (ep.bias_post || (has_bias && vt + G < nwg))) issue_bias(has_bias && vt + G < nwg ? vt + G : -1, vt);
      {
        const unsigned ra = ad_a0 + sn, rb = ad_b0 + sn;
        for_n(integral_constant<int, RT * NT>{}, [&](auto Xc) __attribute__((always_inline)) {
          constexpr int i = decltype(Xc)::value / NT, j = decltype(Xc)::value % NT;
          mma16v<(i < RTA)>(acc[i][j], f1b[j], f1a[i]);       // (this set was waited for in front of the barrier)
          if constexpr ((j & 3) == 3) read_frag(ra, rb, integral_constant<int, 2 * i + (j >> 2)>{}, f0a, f0b);
          if constexpr ((j & 3) == 1) {
            if (more && !(G16V_ABL & 1)) issueP(stS, soA, soW, integral_constant<int, 2 * i + (j >> 2)>{});
          }
        });
      }
      if (more) advanceS();
      ++g;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // (the next tile's first fragments: in flight under nothing that reads them)

    // everything below derives its per-lane addressing from these opaque copies, so none of it is hoisted above the main loop
    int lme = lm, lqe = lq;
    asm volatile("" : "+v"(lme), "+v"(lqe));
    constexpr int NPAIR = NT / 2;
    auto epilogue = [&](auto Gc, auto Rc, auto Pc) __attribute__((always_inline)) {
      constexpr bool GEGLU = decltype(Gc)::value, RES = decltype(Rc)::value, POST = decltype(Pc)::value;   // POST: x * row_scale[m] * alpha + bias_post[n]   // GEGLU: BN = 256 only (host): a wave's 128 columns = two packed [32 h | 32 gate] groups
      constexpr int NBUF = 1;   // (no spare registers for a second residual buffer beside the fragments)
      const int cofs = 16 * (lqe & 1) + 8 * (lqe >> 1);
      const int ncol = N - (col0 + cofs);                   // accumulator columns left of N from this lane's first one
      T* obase = reinterpret_cast<T*>(ep.out) + (long)bz * ep.bso + (GEGLU ? (col0 >> 1) + cofs : col0 + cofs);
      const T* rbase = RES ? reinterpret_cast<const T*>(ep.residual) + (long)bz * ep.bsr + col0 + cofs : nullptr;
      u32x4 rv[RES ? NBUF : 1][RES ? NPAIR : 1];
      auto load_res = [&](int i, u32x4* dst) __attribute__((always_inline)) {
        const int m = row0 + 16 * i + lme;
        const T* rrow = rbase + (long)(m < M ? m : M - 1) * ep.ldr;
#pragma unroll
        for (int jp = 0; jp < NPAIR; ++jp) dst[jp] = *reinterpret_cast<const u32x4*>(rrow + (32 * jp < ncol ? 32 * jp : ncol - 8));   // (beyond N: column N - 8)
      };
      if (RES && NBUF == 2) load_res(0, rv[0]);
      float rsv[POST ? RT : 1];
      const acc4* lpost = reinterpret_cast<const acc4*>(smem + BIAS_OFF + 3 * BIAS_ARR) + ((wn * (BN / 2) + cofs) >> 2);
      if (POST) {
#pragma unroll
        for (int i = 0; i < RT; ++i) {
          const int m = row0 + 16 * i + lme;
          rsv[i] = (ep.row_scale ? ep.row_scale[m < M ? m : M - 1] : 1.f) * ep.alpha;
        }
      }
#pragma unroll
      for (int i = 0; i < RT; ++i) {
#pragma unroll
        for (int j = 0; j < NT; ++j) { if (i < RTA) pin_acc<true>(acc[i][j]); else pin_acc<false>(acc[i][j]); }   // the row tile leaves its registers HERE (hipcc would read all 256 up front and spill them)
        const int m = row0 + 16 * i + lme;
        T* orow = obase + (long)m * ep.ldo;
        if (RES) {
          if (NBUF == 2 ? i < RT - 1 : true) load_res(NBUF == 2 ? i + 1 : i, rv[NBUF == 2 ? (i + 1) & 1 : 0]);
        }
#pragma unroll
        for (int jp = 0; jp < NPAIR; ++jp) {
          if (GEGLU && (jp & 1)) continue;         // tile pairs 1, 3 are the gates of pairs 0, 2
          acc4 x = acc[i][2 * jp], y = acc[i][2 * jp + 1];
          if (GEGLU) {
            const acc4 gx = acc[i][(2 * jp + 2) % NT], gy = acc[i][(2 * jp + 3) % NT];
            const f32x2 g0 = gelu_erf_f2((f32x2){gx[0], gx[1]}), g1 = gelu_erf_f2((f32x2){gx[2], gx[3]});
            const f32x2 g2 = gelu_erf_f2((f32x2){gy[0], gy[1]}), g3 = gelu_erf_f2((f32x2){gy[2], gy[3]});
            x[0] *= g0[0]; x[1] *= g0[1]; x[2] *= g1[0]; x[3] *= g1[1];
            y[0] *= g2[0]; y[1] *= g2[1]; y[2] *= g3[0]; y[3] *= g3[1];
          }
          if (!RES && !POST) {   // nothing is added in the store layout: pack first, swap the two packed dwords per tile (half the swaps)
            const auto s01 = __builtin_amdgcn_permlane16_swap(pack_bf16x2(x[0], x[1]), pack_bf16x2(y[0], y[1]), false, false);
            const auto s23 = __builtin_amdgcn_permlane16_swap(pack_bf16x2(x[2], x[3]), pack_bf16x2(y[2], y[3]), false, false);
            if (m < M && 32 * jp < ncol)
              *reinterpret_cast<u32x4*>(orow + (GEGLU ? 16 * jp : 32 * jp)) = (u32x4){s01[0], s23[0], s01[1], s23[1]};
            continue;
          }
          float o8[8];
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const auto sw2 = __builtin_amdgcn_permlane16_swap(__float_as_uint(x[r]), __float_as_uint(y[r]), false, false);
            o8[r] = __uint_as_float(sw2[0]);
            o8[4 + r] = __uint_as_float(sw2[1]);
          }
          if (POST) {
            const acc4 pa = ep.bias_post ? lpost[8 * jp] : (acc4)(0.f), pb = ep.bias_post ? lpost[8 * jp + 1] : (acc4)(0.f);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              o8[e] = fmaf(o8[e], rsv[i], pa[e]);
              o8[4 + e] = fmaf(o8[4 + e], rsv[i], pb[e]);
            }
          }
          if (RES) {
            union { u32x4 u; bf16_t e[8]; } r8;
            r8.u = rv[NBUF == 2 ? i & 1 : 0][jp];
#pragma unroll
            for (int e = 0; e < 8; ++e) o8[e] += bf16_to_f32(r8.e[e]);
          }
          if (m < M && 32 * jp < ncol)
            *reinterpret_cast<u32x4*>(orow + (GEGLU ? 16 * jp : 32 * jp)) =
                (u32x4){pack_bf16x2(o8[0], o8[1]), pack_bf16x2(o8[2], o8[3]), pack_bf16x2(o8[4], o8[5]), pack_bf16x2(o8[6], o8[7])};
        }
      }
    };
    using std::true_type;
    using std::false_type;
    if constexpr (EPI == 6) {
      // split-K: the raw fp32 accumulators go to slab bz of the partial buffer (lane (lm, lq): row 16 i + lm, columns 16 j + 4 lq .. + 3
      // = one 16-byte store); mmgt_splitk_reduce sums the slabs in slice order and applies the epilogue
      float* pb = reinterpret_cast<float*>(ep.out) + (long)bz * ep.bso + col0 + 4 * lqe;
#pragma unroll
      for (int i = 0; i < RT; ++i) {
#pragma unroll
        for (int j = 0; j < NT; ++j) { if (i < RTA) pin_acc<true>(acc[i][j]); else pin_acc<false>(acc[i][j]); }   // the row tile leaves its registers HERE (hipcc would read all 256 up front and spill them)
        const int m = row0 + 16 * i + lme;
#pragma unroll
        for (int j = 0; j < NT; ++j)
          if (m < M && col0 + 16 * j + 4 * lqe < N) *reinterpret_cast<acc4*>(pb + (long)m * ep.ldo + 16 * j) = acc[i][j];
      }
    } else if constexpr (EPI == 1) epilogue(true_type{}, false_type{}, false_type{});
    else if constexpr (EPI == 3) epilogue(true_type{}, true_type{}, false_type{});
    else if constexpr (EPI == 4) epilogue(false_type{}, false_type{}, true_type{});
    else if constexpr (EPI == 5) epilogue(false_type{}, true_type{}, true_type{});
    else if constexpr (EPI == 2) epilogue(false_type{}, true_type{}, false_type{});
    else epilogue(false_type{}, false_type{}, false_type{});
  }
}

template <int MODE>
int launch16v(const ADesc& ad, const void* W, long bsw, const Epi& ep, int M, int N, int K, int batch, hipStream_t s, int pb_tune) {
  constexpr int BM = 256, BN = 256;
  const int tiles_m = (M + BM - 1) / BM, tiles_n = (N + BN - 1) / BN;
  const size_t lds = (size_t)2 * (BM + BN) * 128 + 4 * 2048;   // stages + the four bias vectors
  static int resident = 0;
  if (!resident) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) {
      mmgt_set_error("gemm16v: device query failed");
      return 2;
    }
    resident = prop.multiProcessorCount;
  }
  long gx = (resident + batch - 1) / batch;
  gx = (gx + 7) / 8 * 8;
  if (gx > (long)tiles_m * tiles_n) gx = (long)tiles_m * tiles_n;
  dim3 grid((unsigned)gx, 1, batch);
  const int pb = pb_tune >= 0 ? pb_tune : (MODE == 0 && tiles_n >= 8 && tiles_m >= 8) ? 8 : 1;
  auto go = [&](auto kern) __attribute__((always_inline)) {
    static bool attr = false;
    if (!attr) {
      if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
        mmgt_set_error("gemm16v: cannot reserve %zu bytes of LDS", lds);
        return 2;
      }
      attr = true;
    }
    hipLaunchKernelGGL(kern, grid, dim3(256), lds, s, ad, reinterpret_cast<const char*>(W), bsw, ep, M, N, K, tiles_m, tiles_n, pb);
    MMGT_LAUNCH_CHECK();
    return 0;
  };
  const bool post = ep.row_scale != nullptr || ep.alpha != 1.f || ep.bias_post != nullptr;
  if (ad.ksplit) return go(gemm16v_kernel<MODE, 6>);
  if (ep.act == 1) return ep.residual ? go(gemm16v_kernel<MODE, 3>) : go(gemm16v_kernel<MODE, 1>);
  if (post) return ep.residual ? go(gemm16v_kernel<MODE, 5>) : go(gemm16v_kernel<MODE, 4>);
  return ep.residual ? go(gemm16v_kernel<MODE, 2>) : go(gemm16v_kernel<MODE, 0>);
}

}  // namespace

// Entry for gemm16.hip's launcher: same preconditions as mmgt_gemm16_launch with bn = 256.
int mmgt_gemm16v_launch(int mode, const void* adp, const void* W, long bsw, const void* epp, int M, int N, int K, int batch, void* stream, int pb_tune) {
  const ADesc& ad = *reinterpret_cast<const ADesc*>(adp);
  const Epi& ep = *reinterpret_cast<const Epi*>(epp);
  hipStream_t s = (hipStream_t)stream;
  return mode == 0 ? launch16v<0>(ad, W, bsw, ep, M, N, K, batch, s, pb_tune) : launch16v<1>(ad, W, bsw, ep, M, N, K, batch, s, pb_tune);
}
