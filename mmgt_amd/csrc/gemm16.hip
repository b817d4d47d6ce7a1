// bf16 GEMM / implicit-GEMM conv3x3 core on 16x16x32 MFMAs: 256 x 256 or 256 x 320 tile, 8 waves, two wave groups in
// ping-pong (gfx950).
//
//   out[M, N] = epilogue( A[M, K] * W[N, K]^T )      same operands, LDS image and epilogue semantics as gemm.hip
//
// Why a second core.  gemm.hip's 8-wave tiles run their eight waves in lock step: every wave queues its LDS-DMA issue and
// its fragment reads in front of its own MFMAs, so the matrix pipe idles through every chunk's load phase (PMC, round 1:
// matrix pipe 55 % busy, 40 % of wave cycles stalled at instruction issue; 1.12 PFLOP/s at 8192^3).  Here the two waves
// that share a SIMD (wave w of rows 0..127 and wave w + 4 of rows 128..255) are kept ONE BARRIER APART for the whole
// kernel: a K chunk of 64 is worked in NPH = BN / 64 phases {load part | s_barrier | 16 MFMAs | s_barrier}, and while one
// group multiplies, its SIMD partners run their load part (fragment reads, two LDS-DMA pieces of the chunks ahead).  The
// 16x16x32 MFMA shape holds a higher clock than 32x32x16 at equal cycles per FLOP (MI355X_MICROARCH.md, DVFS give-back
// item 7) and gives 4-register accumulator tiles whose epilogue needs one v_permlane16_swap per register to turn two tiles
// into 16-byte row vectors.  8192^3: 1.38 PFLOP/s against 1.12 (random data, one process, tools/ab_cfg.py).
//
// Per wave (grid 4 x 2): output 64 x BN/2 = 4 x NT accumulator tiles (NT = BN / 32: 128 or 160 registers).  Phase p of a
// chunk multiplies the wave's four A row tiles (read once per chunk, in phase 0: 8 x ds_read_b128) with W column tiles
// 2p, 2p + 1 (4 x ds_read_b128 per phase): 16 MFMAs.  BN = 320 fits every width of the UNet (320 k) without padding.
// LDS: 2 stages x (256 + BN rows) x 128 B = 128 / 144 KiB; rows are 128 B, the 16-byte chunk index is XOR-swizzled by
// (row >> 1) & 7 on the per-lane DMA SOURCE offset and on the read (conflict-free for this fragment shape too: the sixteen
// lanes of a ds_read_b128 group cover rows r..r+3, r+12..r+15 of one chunk column and rows r+4..r+11 of the next).
// LDS-DMA (1-KiB pieces, 4 x A + NW x W per wave and chunk, NW = BN / 64): the A half of a stage is read in phase 0 only,
// the W half in every phase, so the W pieces of chunk c + 1 (other stage, idle since chunk c - 1) go out first and the
// A pieces of chunk c + 2 follow into the stage chunk c is still multiplying from: piece s of the sequence
// [W 0 .. W NW-1 | A 0 .. A 3] is issued in phase s / 2.  The last phase waits s_waitcnt vmcnt(4): chunk c + 1 has landed,
// the four A pieces of chunk c + 2 stay in flight (two to three half-tiles ahead of the MFMAs).
// Hazards.  WAR: every load part ends in s_waitcnt lgkmcnt(0) in front of its barrier, so a region's reads are retired when
// the barrier releases; group 1 runs one barrier behind group 0; the first A piece into a stage is issued in phase 2 (four
// barriers after group 0's phase-0 reads, three after group 1's), the first W piece in phase 0 of the next chunk (two / one
// barriers after the groups' last-phase reads).  RAW: chunk c + 1 is waited for (counted vmcnt) in the last phase's load part
// by every issuing wave, in front of a barrier both groups pass before their first read of that chunk.
// Tile boundary (what the per-tile stamps of tools/trace_gemm16.py showed, profiles/r2/gemm16_tile_trace_r2.txt): the streams roll on
// into the next tile, group 1 keeps the tile's last barrier for after its epilogue so that both groups' epilogues run side by
// side, the next tile's second W chunk goes out in front of the epilogue stores (vmcnt retires in order), the tile's bias vectors
// are fetched one tile ahead into LDS and the accumulators START from them, and the epilogue is instantiated per mode (GEGLU /
// residual / row scale + post-scale bias) with one base address per row tile.
#include <type_traits>

#include "common.h"
#include "gemm_common.h"
#include "mmgt_hip.h"

namespace {

typedef __attribute__((ext_vector_type(4))) float acc4;

__device__ __forceinline__ acc4 mma16(s16x8 a, s16x8 b, acc4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}

// BM = 192 (48 rows per wave = three 16-row tiles instead of four): the same core for shapes whose 256-row tile count leaves the last round of
// the persistent grid half empty -- 49 152 x 640 is 384 tiles of 256 x 320 = 1.5 rounds of 256 CUs, but 512 tiles of 192 x 320 = two rounds of
// 3/4 the length; 12 288 x 1280 is 192 tiles (a quarter of the chip idle throughout), but exactly 256 tiles of 192 x 320.
template <int MODE, int BN, int BM = 256>
__global__ __launch_bounds__(512, 2) void gemm16_kernel(ADesc ad, const char* __restrict__ W, long bsw, Epi ep, int M, int N,
                                                        int K, int tiles_m, int tiles_n, int pb_stg, unsigned long long* trace) {
  extern __shared__ __attribute__((aligned(1024))) char smem[];
  typedef bf16_t T;
  static_assert(BN == 128 || BN == 256 || BN == 320, "BN");
  static_assert(BM == 256 || BM == 192, "BM");
  constexpr int ESZ = 2, ROWB = 128, BK = 64, NW = 8;
  constexpr int WROWS = BM / 4, RT = WROWS / 16;   // rows and 16-row tiles per wave
  constexpr int A_BYTES = BM * ROWB, B_BYTES = BN * ROWB, STAGE_BYTES = A_BYTES + B_BYTES;
  constexpr int CPR = 8, RPD = 8;               // 16-byte chunks per row; rows per 1-KiB DMA piece
  constexpr int GA = BM / RPD / NW, GB = BN / RPD / NW;   // pieces per wave and chunk: 4 x A + 4 or 5 x W
  constexpr int NPH = BN / 64, NT = BN / 32;    // phases per chunk; accumulator tiles per row tile
  constexpr int BIAS_OFF = 2 * STAGE_BYTES, BIAS_ARR = 2048, NBP = BN == 320 ? 2 : 1;   // bias | bias2 row 0 | bias2 row 1 | post-scale bias: 512 floats each, NBP pieces
  auto swz = [](int row) { return (row >> 1) & 7; };

  const int nwg = tiles_m * tiles_n;
  const int pb = pb_stg & 255, stg = pb_stg >> 8;   // stg: start delay of every second CU's workgroup (tune key g16_stagger, units of 8128 clocks / 16)
  const int bz = blockIdx.z;
  const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wid >> 2;                      // wave group: waves 0-3 / 4-7 (SIMD partners w, w + 4 are in different groups)
  const int wm = wid >> 1, wn = wid & 1;        // 4 x 2 wave grid: 64 rows x BN / 2 columns per wave
  const int lm = lane & 15, lq = lane >> 4;
  int trace_n = 0;
  auto stamp = [&](int k) {   // debug: wave 0 / 4 of every workgroup log the 100-MHz clock at tile phase k
    if (trace && (wid & 3) == 0 && lane == 0 && trace_n < 32) trace[(((long)blockIdx.x * 32 + trace_n) * 2 + wr) * 4 + k] = wall_clock64();
  };

  // XCD-aware virtual tile order (see gemm.hip): XCD x = v & 7 walks a contiguous run of the tile sequence t.  The sequence itself is
  // row-major in groups of `pb` row panels that are walked COLUMN-major inside: the ~32 tiles an XCD works on at any moment then cover
  // pb panels x 32 / pb column tiles -- pb A panels + 32 / pb W panels through its 4 MB L2 instead of ~1 + min(32, tiles_n) (wide outputs:
  // N = 5120 / 10240 GEGLU, 3840 / 1920 q|k|v re-fetched W from the fabric once per row panel).  pb = 1: plain row-major.
  auto decode = [&](int v, int& tm, int& tn) {
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
  const __amdgpu_buffer_rsrc_t rA0 = dma_rsrc(a0), rA1 = dma_rsrc(a1 ? a1 : a0), rW = dma_rsrc(wbase);
  const int c0sw = spos ^ (srow >> 1);
  int limA = 0, limW = 0;
  unsigned aoff[MODE == 0 ? 2 : GA];   // dense: [piece parity]; conv: the current tap's pixel of every piece (zero padding = poison)
  unsigned woff[2];
  unsigned am0 = 0;                    // conv: output row of piece 0 (piece i: + 8 i)
  int a_step = 0, w_step = 0;          // scalar byte distance between consecutive pieces (dense A rows / W rows, 8 apart)
  int p_tap = 0, p_c = 0, a_soff = 0;
  bool a_second = false, a_fresh = true;   // a_fresh: the first chunk of a tile computes its gather offsets wherever the slice starts
  // The A and the W stream run at different distances ahead of the MFMAs (see the schedule above), so each keeps its own
  // position: tile, chunk inside the tile, chunks issued so far.
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
    const unsigned row = (unsigned)(tn * BN + wid * GB * RPD + srow);
    w_step = RPD * ldw * ESZ;
    limW = N - (int)row;
#pragma unroll
    for (int q = 0; q < 2; ++q) woff[q] = row * (unsigned)(ldw * ESZ) + (unsigned)((c0sw ^ (4 * ((q + wid * GB) & 1))) << 4);
  };
  auto prepA = [&](int ch) {   // source offsets of the A pieces of chunk `ch` of the A stream's tile (chunks come strictly in order)
    if (MODE == 0) {
      a_soff = ch * ROWB;
    } else {
      const int cin = ad.C0 + ad.C1;
      if (p_c == 0 || p_c == ad.C0 || a_fresh) {
        a_fresh = false;
        // up == 2: the conv behind a nearest 2x upsampling as FOUR 2 x 2 convs on the stored image, one per output phase (a, b) = grid.z:
        // output pixel (2 y + a, 2 x + b) sees input rows y + a - 1, y + a through the 3 x 3 weights summed per source pixel (the host's phase
        // image, packing.pack_conv3x3_up2): 16 instead of 36 multiply-adds per input pixel, the same sums
        // up == 3: a 1 x 1 conv over the two sources (the resnets' conv_shortcut over [x | skip]): one tap, the centre -- the gather's two-source
        // reduction without the nine taps (K = C0 + C1)
        const bool up2 = ad.up == 2;
        const int ky = up2 ? p_tap >> 1 : ad.up == 3 ? 1 : p_tap / 3, kx = up2 ? p_tap & 1 : ad.up == 3 ? 1 : p_tap - ky * 3;
        const int vh = ad.up == 1 ? ad.IH * 2 : ad.IH, vw = ad.up == 1 ? ad.IW * 2 : ad.IW;
        const bool second = p_c >= ad.C0 && ad.C1 > 0;
        const unsigned cpb = (unsigned)(second ? ad.C1 : ad.C0) * ESZ;   // bytes per pixel of the source tensor (< 2 GiB in all: host check)
        a_second = second;
        a_soff = (p_c - (second ? ad.C0 : 0)) * ESZ;
#pragma unroll
        for (int i = 0; i < GA; ++i) {
          const unsigned m = am0 + RPD * i;
          const unsigned cn = fastdiv(m, ad.fd_hw), rem = m - cn * (unsigned)(ad.OH * ad.OW);
          const unsigned oy = fastdiv(rem, ad.fd_ow), ox = rem - oy * (unsigned)ad.OW;
          const int iy = up2 ? (int)oy + ky + (bz >> 1) - 1 : (int)oy * ad.stride + ky - ad.pad;
          const int ix = up2 ? (int)ox + kx + (bz & 1) - 1 : (int)ox * ad.stride + kx - ad.pad;
          const bool ok = (int)m < M && iy >= 0 && iy < vh && ix >= 0 && ix < vw;
          const int sy = ad.up == 1 ? iy >> 1 : iy, sx = ad.up == 1 ? ix >> 1 : ix;
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
  auto issueA = [&](int stage, auto Ic) {     // A piece I of the chunk prepared by prepA()
    constexpr int I = decltype(Ic)::value;
    const __amdgpu_buffer_rsrc_t rA = (MODE == 1 && a_second) ? rA1 : rA0;
    if (MODE == 0) blds16(rA, RPD * I < limA ? aoff[I & 1] : DMA_POISON, a_soff + I * a_step, smem + stage * STAGE_BYTES + (wid * GA + I) * 1024);
    else blds16(rA, aoff[I], a_soff, smem + stage * STAGE_BYTES + (wid * GA + I) * 1024);
  };
  auto issueW = [&](int stage, int ch, auto Ic) {
    constexpr int I = decltype(Ic)::value;
    blds16(rW, RPD * I < limW ? woff[I & 1] : DMA_POISON, ch * ROWB + I * w_step, smem + stage * STAGE_BYTES + A_BYTES + (wid * GB + I) * 1024);
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
  auto issue_bias = [&](int v, int vc) {
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
  using std::integral_constant;

  // ---- fragment read addressing: lane (lm, lq) reads row base + 16 t + lm, 16-byte chunk (4 ks + lq) ^ swz; the swizzle
  // of a row depends on lm only ((16 t + lm) >> 1 & 7 == lm >> 1 & 7), so two per-lane offsets (ks = 0, 1) serve every tile
  const int sw = (lm >> 1) & 7;
  const int roff0 = lm * ROWB + ((lq ^ sw) << 4), roff1 = lm * ROWB + (((4 + lq) ^ sw) << 4);
  const int a_base = wm * WROWS * ROWB;                    // this wave's A rows
  const int b_base = A_BYTES + wn * (BN / 2) * ROWB;       // this wave's BN / 2 W rows

  const int nchunks = K / BK;
  const int G = gridDim.x;
  const int my_tiles = (nwg - (int)blockIdx.x + G - 1) / G;
  const int total = my_tiles * nchunks;
  int vtA = blockIdx.x, ichA = 0, giA = 0;          // A stream: tile, chunk in tile, chunks issued
  int vtW = blockIdx.x, ichW = 0, giW = 0;          // W stream
  int sc = 0;                                       // MFMA side: stage to read
  // Tile boundary: the W pieces of the next tile's chunk 1 go out in FRONT of a tile's epilogue (their stage, the last chunk's,
  // is idle from the last phase's barriers on), so that chunk 0 of the next tile has nothing to wait for that is younger than
  // the epilogue's stores: vmcnt retires in order, and waiting for a load issued behind the stores made every tile's first
  // chunk sit through the drain of 128-160 KiB of output (trace: 4.3-5.7 us against 1.7-2.0 for the other chunks).  That
  // chunk's counted wait then leaves the stores of a full tile in flight (w_relax); they have until the end of chunk 1.
  bool w_pre = false, w_relax = false;
  {
    int tm, tn;
    decode(vtA, tm, tn);
    setupA(tm);
    setupW(tn);
  }
  auto advanceA = [&]() {
    ++giA;
    if (++ichA == nchunks) {
      ichA = 0;
      vtA += G;
      if (vtA < nwg) {
        int tm, tn;
        decode(vtA, tm, tn);
        setupA(tm);
      }
    }
  };
  auto advanceW = [&]() {
    ++giW;
    if (++ichW == nchunks) {
      ichW = 0;
      vtW += G;
      if (vtW < nwg) {
        int tm, tn;
        decode(vtW, tm, tn);
        setupW(tn);
      }
    }
  };
  // static loops over compile-time indices
  auto for_n = [&](auto Nc, auto&& f) {
    constexpr int N_ = decltype(Nc)::value;
    [&]<int... I>(std::integer_sequence<int, I...>) { (f(integral_constant<int, I>{}), ...); }(std::make_integer_sequence<int, N_>{});
  };
  if (stg && ((blockIdx.x >> 3) & 1))
    for (int i = 0; i < stg; ++i) __builtin_amdgcn_s_sleep(8);
  // prologue: chunk 0 whole, the A half of chunk 1 (the A stream runs two chunks ahead of the MFMAs, the W stream one)
  if (total > 0) {
    if (wr == 0 && has_bias) issue_bias(blockIdx.x, blockIdx.x);   // (the post-scale bias of the first tile is issued again in its first chunk: harmless)
    prepA(ichA);
    for_n(integral_constant<int, GA>{}, [&](auto i) { issueA(0, i); });
    advanceA();
    for_n(integral_constant<int, GB>{}, [&](auto i) { issueW(0, ichW, i); });
    advanceW();
  }
  if (total > 1) {
    prepA(ichA);
    for_n(integral_constant<int, GA>{}, [&](auto i) { issueA(1, i); });
    advanceA();
    wait_vmcnt<GA>();
  } else {
    wait_vmcnt<0>();
  }
  __builtin_amdgcn_s_barrier();
  if (wr == 1) __builtin_amdgcn_s_barrier();   // the stagger: group 1 runs one barrier behind group 0 from here on

  for (int vt = blockIdx.x; vt < nwg; vt += G) {
    int tm, tn;
    decode(vt, tm, tn);
    const int row0 = tm * BM + wm * WROWS, col0 = tn * BN + wn * (BN / 2);

    acc4 acc[RT][NT];
    if (has_bias) {
      const acc4* lb = reinterpret_cast<const acc4*>(smem + BIAS_OFF) + wn * (BN / 8) + lq;   // this lane's columns 16 j + 4 lq + r
      const int b2r0 = (tm * BM) / b2div;
      bool second[RT];                     // row 16 i + lm of the wave belongs to the tile's second bias2 row
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
          for (int i = 0; i < RT; ++i) acc[i][j] = b + (second[i] ? r1v : r0v);
        } else {
#pragma unroll
          for (int i = 0; i < RT; ++i) acc[i][j] = b;
        }
      }
    } else {
#pragma unroll
      for (int i = 0; i < RT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = (acc4)(0.f);
    }

    stamp(0);
    s16x8 fa[RT][2], fb[2][2];   // [tile][ks]
    for (int ch = 0; ch < nchunks; ++ch) {
      if (trace && (wid & 3) == 0 && lane == 0 && trace_n < 8 && ch < 16)   // debug: chunk start stamps behind the tile stamps
        trace[65536 + (((long)blockIdx.x * 8 + trace_n) * 2 + wr) * 16 + ch] = wall_clock64();
      const char* st = smem + sc * STAGE_BYTES;
      const bool skipW = w_pre && ch == 0;                   // this chunk's W issue went out in front of the previous epilogue
      const bool moreW = !skipW && giW < total, moreA = giA < total;   // uniform: chunk c + 1 (W) / chunk c + 2 (A) exist
      const int stW = giW & 1, stA = giA & 1, chW = ichW;
      if (moreA) prepA(ichA);
      for_n(integral_constant<int, NPH>{}, [&](auto Pc) {
        constexpr int P = decltype(Pc)::value;
        // ---- load part
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          const char* pb = st + b_base + (32 * P + 16 * t) * ROWB;
          fb[t][0] = *reinterpret_cast<const s16x8*>(pb + roff0);
          fb[t][1] = *reinterpret_cast<const s16x8*>(pb + roff1);
        }
        if (P == 0) {
#pragma unroll
          for (int t = 0; t < RT; ++t) {
            const char* pa = st + a_base + 16 * t * ROWB;
            fa[t][0] = *reinterpret_cast<const s16x8*>(pa + roff0);
            fa[t][1] = *reinterpret_cast<const s16x8*>(pa + roff1);
          }
        }
        if (P == 1 && ch == 0 && wr == 0 && (ep.bias_post || (has_bias && vt + G < nwg))) issue_bias(has_bias && vt + G < nwg ? vt + G : -1, vt);
        // pieces 2 P, 2 P + 1 of [W 0 .. W GB-1 | A 0 .. A GA-1].  BN = 128 has two phases for its 2 + GA pieces: the W pieces in phase 0, ALL
        // A pieces in phase 1 -- an A piece lands in the stage the current chunk is multiplied from, whose A half group 1 reads one barrier
        // after group 0's phase-0 load part, so phase 1 is the first that may overwrite it.
        constexpr int PLO = BN == 128 ? (P == 0 ? 0 : GB) : 2 * P, PN = BN == 128 ? (P == 0 ? GB : GA) : 2;
        for_n(integral_constant<int, PN>{}, [&](auto Uc) {
          constexpr int S = PLO + decltype(Uc)::value;
          if constexpr (S < GB) {
            if (moreW) issueW(stW, chW, integral_constant<int, S>{});
            if constexpr (S == GB - 1) { if (moreW) advanceW(); }
          } else if constexpr (S < GB + GA) {
            if (moreA) issueA(stA, integral_constant<int, S - GB>{});
            if constexpr (S == GB + GA - 1) { if (moreA) advanceA(); }
          }
        });
        if (P == NPH - 1) {
          // chunk c + 1 must have landed before the barrier that precedes its first read; the GA A pieces of chunk c + 2
          // issued during this chunk are the only younger LDS-DMA (epilogue loads / stores of a tile end are older)
          constexpr int NST = RT * (NT / 2);                 // 16-byte stores of a full tile's epilogue per wave (GEGLU: half)
          if (skipW && w_relax) {
            if (ep.act == 1) { if (moreA) wait_vmcnt<GA + NST / 2>(); else wait_vmcnt<NST / 2>(); }
            else { if (moreA) wait_vmcnt<GA + NST>(); else wait_vmcnt<NST>(); }
          } else {
            if (moreA) wait_vmcnt<GA>(); else wait_vmcnt<0>();
          }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // this wave's LDS reads have returned: the barrier may free their region
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        // ---- multiply part
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
          for (int t = 0; t < RT; ++t)
#pragma unroll
            for (int u = 0; u < 2; ++u) acc[t][2 * P + u] = mma16(fb[u][ks], fa[t][ks], acc[t][2 * P + u]);
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
        // Group 1 keeps the tile's LAST barrier for after its epilogue: group 0 meets that barrier with its first one of the next
        // tile, after its own epilogue, so the two epilogues run side by side instead of one after the other (trace: the first
        // chunk of a tile took 4.3-5.7 us against 1.7-2.0 for the others while each group waited out the other's epilogue).
        if (!(P == NPH - 1 && wr == 1 && ch == nchunks - 1)) __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
      });
      sc ^= 1;
    }

    w_pre = giW < total;
    if (w_pre) {
      for_n(integral_constant<int, GB>{}, [&](auto i) { issueW(giW & 1, ichW, i); });
      advanceW();
    }
    w_relax = !ad.ksplit && nchunks >= 2 && tm * BM + BM <= M && tn * BN + BN <= N;   // every lane of every wave stores: the count above is exact
    // everything below derives its per-lane addressing from these opaque copies, so none of it is hoisted above the main loop
    // (where it would only lengthen live ranges: the loop runs at the register limit)
    int lme = lm, lqe = lq;
    asm volatile("" : "+v"(lme), "+v"(lqe));
    stamp(1);
    // ---- epilogue from registers.  D = W_frag x A_frag: lane (lm, lq) holds output row m = 16 i + lm and, per accumulator
    // tile j, the four columns 16 j + 4 lq + r.  v_permlane16_swap exchanges the odd 16-lane rows of tile j with the even rows
    // of tile j + 1, after which the lane owns 8 consecutive columns 16 j + 16 (lq & 1) + 8 (lq >> 1) .. + 7: one 16-byte
    // store per lane and tile pair, 64 contiguous bytes per output row and instruction.
    // One instantiation per mode (GEGLU / residual are compile-time here): a row tile's addresses are one 64-bit base per
    // operand plus constant column offsets, and no branch sits between the steps of the unrolled loops.
    // Residual vectors (clamped addresses).  256 columns: row tile i + 1's are requested before row tile i is worked, so the
    // tile end exposes one memory round trip; 320 columns: no registers to spare (more pressure here spills main-loop
    // state), each row tile waits for its own vectors.
    constexpr int NPAIR = NT / 2;
    stamp(2);
    auto epilogue = [&](auto Gc, auto Rc, auto Pc) {
      constexpr bool GEGLU = decltype(Gc)::value, RES = decltype(Rc)::value, POST = decltype(Pc)::value;   // POST: x * row_scale[m] * alpha + bias_post[n]   // GEGLU: BN = 256 only (host): a wave's 128 columns = two packed [32 h | 32 gate] groups
      constexpr int NBUF = (BN == 256 || (MODE == 0 && !POST)) ? 2 : 1;   // (the 320-column conv / row-scaled variants have no registers to spare)
      const int cofs = 16 * (lqe & 1) + 8 * (lqe >> 1);
      const int ncol = N - (col0 + cofs);                   // accumulator columns left of N from this lane's first one
      T* obase = reinterpret_cast<T*>(ep.out) + (long)bz * ep.bso + (GEGLU ? (col0 >> 1) + cofs : col0 + cofs);
      const T* rbase = RES ? reinterpret_cast<const T*>(ep.residual) + (long)bz * ep.bsr + col0 + cofs : nullptr;
      u32x4 rv[RES ? NBUF : 1][RES ? NPAIR : 1];
      auto load_res = [&](int i, u32x4* dst) {
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
        const int m = row0 + 16 * i + lme;
        T* orow = obase + (long)m * ep.ldo;
        if (MODE == 1 && ad.up == 2) {          // phase (a, b) = grid.z of the upsampled output: row (n, y, x) -> pixel (2 y + a, 2 x + b)
          const unsigned mm = (unsigned)(m < M ? m : M - 1);
          const unsigned cn = fastdiv(mm, ad.fd_hw), rem = mm - cn * (unsigned)(ad.OH * ad.OW);
          const unsigned oy = fastdiv(rem, ad.fd_ow), ox = rem - oy * (unsigned)ad.OW;
          orow = obase + ((long)(cn * 2u * (unsigned)ad.OH + 2u * oy + (unsigned)(bz >> 1)) * (2 * ad.OW) + 2 * ox + (bz & 1)) * ep.ldo;
        }
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
              st_out16(orow + (GEGLU ? 16 * jp : 32 * jp), (u32x4){s01[0], s23[0], s01[1], s23[1]});
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
            st_out16(orow + (GEGLU ? 16 * jp : 32 * jp),
                     (u32x4){pack_bf16x2(o8[0], o8[1]), pack_bf16x2(o8[2], o8[3]), pack_bf16x2(o8[4], o8[5]), pack_bf16x2(o8[6], o8[7])});
        }
      }
    };
    using std::true_type;
    using std::false_type;
    const bool post = ep.row_scale != nullptr || ep.alpha != 1.f || ep.bias_post != nullptr;
    if (ad.ksplit) {
      // split-K: the raw fp32 accumulators go to slab bz of the partial buffer (lane (lm, lq): row 16 i + lm, columns 16 j + 4 lq .. + 3
      // = one 16-byte store); mmgt_splitk_reduce sums the slabs in slice order and applies the epilogue
      float* pb = reinterpret_cast<float*>(ep.out) + (long)bz * ep.bso + col0 + 4 * lqe;
#pragma unroll
      for (int i = 0; i < RT; ++i) {
        const int m = row0 + 16 * i + lme;
#pragma unroll
        for (int j = 0; j < NT; ++j)
          if (m < M && col0 + 16 * j + 4 * lqe < N) *reinterpret_cast<acc4*>(pb + (long)m * ep.ldo + 16 * j) = acc[i][j];
      }
    } else if (BN == 256 && ep.act == 1) {
      if constexpr (BN == 256) {
        if (ep.residual) epilogue(true_type{}, true_type{}, false_type{}); else epilogue(true_type{}, false_type{}, false_type{});
      }
    } else if (post) {
      if (ep.residual) epilogue(false_type{}, true_type{}, true_type{}); else epilogue(false_type{}, false_type{}, true_type{});
    } else if (ep.residual) {
      epilogue(false_type{}, true_type{}, false_type{});
    } else {
      epilogue(false_type{}, false_type{}, false_type{});
    }
    stamp(3);
    ++trace_n;
    if (wr == 1) __builtin_amdgcn_s_barrier();   // (the barrier group 1 skipped in the tile's last phase)
  }
  if (wr == 0) __builtin_amdgcn_s_barrier();   // group 0 meets group 1's last barrier
}

unsigned long long* g_trace = nullptr;
int g_stg = -1;   // mmgt_tune("g16_stagger", v): start delay of every second CU's workgroup in units of 512 clocks (-1 = by shape)
int g_pb = -1;   // mmgt_tune("g16_pb", v): row panels per column-major group of the tile order (-1 = by shape, 1 = row-major)

template <int MODE, int BN, int BM = 256>
int launch16(const ADesc& ad, const void* W, long bsw, const Epi& ep, int M, int N, int K, int batch, hipStream_t s) {
  const int tiles_m = (M + BM - 1) / BM, tiles_n = (N + BN - 1) / BN;
  const size_t lds = (size_t)2 * (BM + BN) * 128 + 4 * 2048;   // stages + the four bias vectors
  auto kern = gemm16_kernel<MODE, BN, BM>;
  static int resident = 0;
  if (!resident) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) !=
        hipSuccess) {
      mmgt_set_error("gemm16: cannot reserve %zu bytes of LDS", lds);
      return 2;
    }
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) {
      mmgt_set_error("gemm16: device query failed");
      return 2;
    }
    resident = prop.multiProcessorCount;   // 128 / 144 KiB of LDS: one workgroup per CU
  }
  long gx = (resident + batch - 1) / batch;
  gx = (gx + 7) / 8 * 8;
  if (gx > (long)tiles_m * tiles_n) gx = (long)tiles_m * tiles_n;
  dim3 grid((unsigned)gx, 1, batch);
  // (measured, tools/ab_cfg.py G16_PB=1,4,8: see DESIGN.md; convs and narrow outputs keep the row-major order)
  const int pb = g_pb >= 0 ? g_pb : (MODE == 0 && tiles_n >= 8 && tiles_m >= 8) ? 8 : 1;
  // Short reductions with a residual epilogue spend as long in the tile end (residual in, tile out: HBM) as in the loop, and every workgroup of
  // the chip gets there at the same time: half of them start ~2.5 us late (profiles/r6/bench_shortk_r6.txt: -7 ... -10 % on K = 320 / 640)
  const int stg = g_stg >= 0 ? g_stg : (MODE == 0 && ep.residual && !ad.ksplit && K <= 1280) ? 10 : 0;
  hipLaunchKernelGGL(kern, grid, dim3(512), lds, s, ad, reinterpret_cast<const char*>(W), bsw, ep, M, N, K, tiles_m, tiles_n, pb | (stg << 8), g_trace);
  MMGT_LAUNCH_CHECK();
  return 0;
}

}  // namespace

// Debug (tools/trace_gemm16.py): a device buffer of [grid][32 tiles][2 groups][4 stamps] u64 receives 100-MHz time stamps.
extern "C" void mmgt_gemm16_set_trace(void* p) { g_trace = reinterpret_cast<unsigned long long*>(p); }
void mmgt_gemm16_set_pb(int v) { g_pb = v; }
void mmgt_gemm16_set_stagger(int v) { g_stg = v; }

namespace {

// ---- split-K for grids that cannot fill the chip (the 8x8-level convs: 3072 output rows = 60 tiles of 256 x 256 for 256 CUs, with
// reductions of 11 520 .. 23 040): S slices of the reduction run as S x tiles workgroups writing fp32 partial slabs, and this
// kernel sums the slabs in slice order (bitwise reproducible) and applies bias / per-batch bias / residual.  8 columns per thread.
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* __restrict__ part, int S, long slab, const float* __restrict__ bias,
                                                            const float* __restrict__ bias2, int b2rows, const bf16_t* __restrict__ res,
                                                            long ldr, bf16_t* __restrict__ out, long ldo, int M, int N, int m0) {
  const int nv = N / 8;
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (long)M * nv) return;
  const int m = (int)(idx / nv), c = (int)(idx - (long)m * nv) * 8;
  const float* p = part + (long)m * N + c;
  f32x4 a0 = *reinterpret_cast<const f32x4*>(p), a1 = *reinterpret_cast<const f32x4*>(p + 4);
  for (int z = 1; z < S; ++z) {
    a0 += *reinterpret_cast<const f32x4*>(p + z * slab);
    a1 += *reinterpret_cast<const f32x4*>(p + z * slab + 4);
  }
  if (bias) { a0 += *reinterpret_cast<const f32x4*>(bias + c); a1 += *reinterpret_cast<const f32x4*>(bias + c + 4); }
  if (bias2) {
    const float* b2 = bias2 + (long)((m + m0) / b2rows) * N + c;      // (m0: the launch covers rows m0 .. of a larger problem)
    a0 += *reinterpret_cast<const f32x4*>(b2);
    a1 += *reinterpret_cast<const f32x4*>(b2 + 4);
  }
  if (res) {
    union { u32x4 u; bf16_t e[8]; } r8;
    r8.u = *reinterpret_cast<const u32x4*>(res + (long)m * ldr + c);
#pragma unroll
    for (int e = 0; e < 4; ++e) { a0[e] += bf16_to_f32(r8.e[e]); a1[e] += bf16_to_f32(r8.e[4 + e]); }
  }
  *reinterpret_cast<u32x4*>(out + (long)m * ldo + c) =
      (u32x4){pack_bf16x2(a0[0], a0[1]), pack_bf16x2(a0[2], a0[3]), pack_bf16x2(a1[0], a1[1]), pack_bf16x2(a1[2], a1[3])};
}

// Partial-sum workspace of the split-K / tail-split launches: one buffer per device, owned by the library.  It grows only OUTSIDE a
// stream capture (hipMalloc / hipFree inside one would invalidate the capture), and a buffer that a capture has recorded is never
// freed: a graph replays with the pointer it captured, so a later growth retires that buffer instead of releasing it.  Growth is
// geometric and the retired list is bounded (every growth after a capture used to leak a slab without limit).
constexpr int MAX_DEV = 16, MAX_RETIRED = 8;
struct SplitkWs {
  float* p = nullptr; size_t bytes = 0; bool captured = false;
  float* retired[MAX_RETIRED] = {};          // buffers a graph captured before the workspace grew: kept alive for its replays, and
  size_t retired_bytes[MAX_RETIRED] = {};    // taken back into service when a later request fits one of them
  int nretired = 0;
};
SplitkWs g_splitk_ws[MAX_DEV];

float* splitk_workspace(size_t need, hipStream_t s) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= MAX_DEV) {
    mmgt_set_error("gemm16 split-K: device query failed");
    return nullptr;
  }
  SplitkWs& w = g_splitk_ws[dev];
  hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
  const bool capturing = hipStreamIsCapturing(s, &cs) == hipSuccess && cs != hipStreamCaptureStatusNone;
  if (need > w.bytes) {
    if (capturing) {
      mmgt_set_error("gemm16 split-K: the partial-sum workspace would have to grow to %zu bytes inside a stream capture; run the "
                     "shape once outside the capture (warm-up) first", need);
      return nullptr;
    }
    // Grow geometrically (at most log2 growths over a process) so that a sequence of slowly growing shapes cannot retire a slab each.
    size_t want = need;
    if (w.bytes && want < 2 * w.bytes) want = 2 * w.bytes;
    if (w.p && !w.captured) {
      (void)hipFree(w.p);                             // (synchronises the device: nothing is reading the old buffer afterwards)
    } else if (w.p) {
      if (w.nretired == MAX_RETIRED) {
        mmgt_set_error("gemm16 split-K: %d captured workspaces already retired; destroy the graphs that hold them and restart", MAX_RETIRED);
        return nullptr;
      }
      w.retired[w.nretired] = w.p;
      w.retired_bytes[w.nretired++] = w.bytes;
    }
    w.p = nullptr; w.bytes = 0; w.captured = false;
    if (hipMalloc(reinterpret_cast<void**>(&w.p), want) != hipSuccess) {
      if (want == need || hipMalloc(reinterpret_cast<void**>(&w.p), need) != hipSuccess) {
        w.p = nullptr;
        mmgt_set_error("gemm16 split-K: cannot allocate %zu bytes of partial sums", need);
        return nullptr;
      }
      want = need;
    }
    w.bytes = want;
  }
  if (capturing) w.captured = true;
  return w.p;
}

}  // namespace

// Split-K entry for gemm.hip's dispatcher: bf16, one problem (batch 1), no activation / row scale / post-scale bias, N % 8 == 0,
// (K / 64) % S == 0.  Returns 0 on success; the partial slabs live in the library-owned per-device buffer above (work on one stream
// at a time per device: the buffer is reused by the next call).
int mmgt_gemm16_splitk(int mode, int bn, const void* adp, const void* W, const void* epp, int M, int N, int K, int S, void* stream, int m0) {
  ADesc ad = *reinterpret_cast<const ADesc*>(adp);
  const Epi& ep = *reinterpret_cast<const Epi*>(epp);
  hipStream_t s = (hipStream_t)stream;
  const size_t need = (size_t)S * M * N * sizeof(float);
  float* ws = splitk_workspace(need, s);
  if (!ws) return 2;
  ad.ksplit = 1;
  ad.ldw = K;
  Epi pe{};
  pe.out = reinterpret_cast<char*>(ws);
  pe.ldo = N;
  pe.bso = (long)M * N;
  pe.alpha = 1.f;
  pe.fast = 1;
  const int ks = K / S;
  int rc;
  if (bn == 320) rc = mode == 0 ? launch16<0, 320>(ad, W, 0, pe, M, N, ks, S, s) : launch16<1, 320>(ad, W, 0, pe, M, N, ks, S, s);
  else rc = mode == 0 ? launch16<0, 256>(ad, W, 0, pe, M, N, ks, S, s) : launch16<1, 256>(ad, W, 0, pe, M, N, ks, S, s);
  if (rc) return rc;
  const long nthr = (long)M * (N / 8);
  hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)((nthr + 255) / 256)), dim3(256), 0, s, ws, S, (long)M * N, ep.bias, ep.bias2,
                     ep.bias2 ? ep.bias2_rows : 1, reinterpret_cast<const bf16_t*>(ep.residual), ep.ldr, reinterpret_cast<bf16_t*>(ep.out), ep.ldo, M, N, m0);
  MMGT_LAUNCH_CHECK();
  return 0;
}

// bf16 operands, RAW fp32 accumulators out: out[m][n] = sum_k A[m][k] * W[n][k], no epilogue -- the split-K slab writer with one slice that
// covers the whole reduction.  For products whose result must keep more than a bf16 mantissa (the VAE's 512-wide attention logits, fed
// with hi / lo split operands: mmgt_amd/vae.py).  K % 64 == 0, N % 8 == 0, 16-byte aligned rows.
extern "C" int mmgt_gemm_bf16_f32(const void* A, long lda, const void* W, long ldw, float* out, long ldo, int M, int N, int K, void* stream) {
  MMGT_CHECK(A && W && out && M > 0 && N > 0 && K > 0, "gemm_bf16_f32: bad arguments");
  MMGT_CHECK(K % 64 == 0 && N % 8 == 0 && lda >= K && ldw >= K && ldo >= N && lda % 8 == 0 && ldw % 8 == 0 && ldo % 4 == 0,
             "gemm_bf16_f32: unsupported shape M=%d N=%d K=%d lda=%ld ldw=%ld ldo=%ld", M, N, K, lda, ldw, ldo);
  MMGT_CHECK(((uintptr_t)A % 16) == 0 && ((uintptr_t)W % 16) == 0 && ((uintptr_t)out % 16) == 0, "gemm_bf16_f32: pointers must be 16-byte aligned");
  MMGT_CHECK(((long)(M - 1) * lda + K) * 2 < (1l << 31) && ((long)(N - 1) * ldw + K) * 2 < (1l << 31),
             "gemm_bf16_f32: an operand exceeds the 2 GiB range of the 32-bit LDS-DMA offsets");
  ADesc ad{};
  ad.src0 = reinterpret_cast<const char*>(A);
  ad.ld0 = lda;
  ad.ksplit = 1;
  ad.ldw = (int)ldw;
  Epi pe{};
  pe.out = reinterpret_cast<char*>(out);
  pe.ldo = ldo;
  pe.alpha = 1.f;
  pe.fast = 1;
  return launch16<0, 256>(ad, W, 0, pe, M, N, K, 1, (hipStream_t)stream);
}

// Entry for gemm.hip's dispatcher.  Preconditions (checked there): bf16, vectorised epilogue (ep.fast), act in {none, GEGLU}
// (GEGLU with bn = 256 only and without row scale / alpha / post-scale bias), 16-byte aligned bias vectors, K % 64 == 0.
int mmgt_gemm16_launch(int mode, int bn, const void* adp, const void* W, long bsw, const void* epp, int M, int N, int K,
                       int batch, void* stream) {
  const ADesc& ad = *reinterpret_cast<const ADesc*>(adp);
  const Epi& ep = *reinterpret_cast<const Epi*>(epp);
  hipStream_t s = (hipStream_t)stream;
  if (bn == 192320) return mode == 0 ? launch16<0, 320, 192>(ad, W, bsw, ep, M, N, K, batch, s) : launch16<1, 320, 192>(ad, W, bsw, ep, M, N, K, batch, s);   // 192 x 320 tile
  if (bn == 320) return mode == 0 ? launch16<0, 320>(ad, W, bsw, ep, M, N, K, batch, s) : launch16<1, 320>(ad, W, bsw, ep, M, N, K, batch, s);
  if (bn == 128) return mode == 0 ? launch16<0, 128>(ad, W, bsw, ep, M, N, K, batch, s) : launch16<1, 128>(ad, W, bsw, ep, M, N, K, batch, s);   // 256 x 128 tile (the VAE's 128-wide levels)
  return mode == 0 ? launch16<0, 256>(ad, W, bsw, ep, M, N, K, batch, s) : launch16<1, 256>(ad, W, bsw, ep, M, N, K, batch, s);
}
