// bf16 GEMM / implicit-GEMM conv3x3 core on 16x16x32 MFMAs: 256 x 256 tile, 8 waves, two wave groups in ping-pong (gfx950).
//
//   out[M, N] = epilogue( A[M, K] * W[N, K]^T )      same operands, LDS image and epilogue semantics as gemm.hip
//
// Why a second core.  gemm.hip's 8-wave tiles run their eight waves in lock step: every wave queues its LDS-DMA issue and
// its fragment reads in front of its own MFMAs, so the matrix pipe idles through every chunk's load phase (PMC, round 1:
// matrix pipe 55 % busy, 40 % of wave cycles stalled at instruction issue; 1.12 PFLOP/s at 8192^3).  Here the two waves
// that share a SIMD (wave w of rows 0..127 and wave w + 4 of rows 128..255) are kept ONE BARRIER APART for the whole
// kernel: a K chunk of 64 is worked in four phases {load part | s_barrier | 16 MFMAs | s_barrier}, and while one group
// multiplies, its SIMD partners run their load part (fragment reads of the next C quadrant, a slice of the next chunk's
// LDS-DMA).  The 16x16x32 MFMA shape holds a higher clock than 32x32x16 at equal cycles per FLOP
// (MI355X_MICROARCH.md, DVFS give-back item 7) and gives 4-register accumulator tiles whose epilogue needs one
// v_permlane16_swap per register to turn two tiles into 16-byte row vectors.
//
// Per wave: output 128 x 64 = 8 x 4 accumulator tiles (128 registers); per chunk and phase p one C quadrant of 64 x 32:
//   p0: read B0 (4 x ds_read_b128), A0 (8)   -> rows  0..63  x cols  0..31        DMA pieces [0, 2) of the next chunk
//   p1: read B1 (4)                          -> rows  0..63  x cols 32..63        DMA pieces [2, 5)
//   p2: read A1 (8)                          -> rows 64..127 x cols 32..63        DMA pieces [5, 8)
//   p3: (nothing to read: A1, B0 are live)   -> rows 64..127 x cols  0..31        s_waitcnt vmcnt(0): next chunk landed
// LDS: 2 stages x (256 + 256 rows) x 128 B = 128 KiB; rows are 128 B, the 16-byte chunk index is XOR-swizzled by
// (row >> 1) & 7 on the per-lane DMA SOURCE offset and on the read (conflict-free for this fragment shape too: the sixteen
// lanes of a ds_read_b128 group cover rows r..r+3, r+12..r+15 of one chunk column and rows r+4..r+11 of the next).
// Hazards: a stage is refilled in phases 0-2 of the chunk AFTER the one that read it last (its last reads, phase 2, are
// retired by an lgkmcnt(0) in front of that phase's barrier, three barriers earlier for either group); the refill is
// waited for (vmcnt) in phase 3's load part by every issuing wave, in front of a barrier both groups pass before their
// first read of it.
#include <type_traits>

#include "common.h"
#include "gemm_common.h"
#include "mmgt_hip.h"

namespace {

typedef __attribute__((ext_vector_type(4))) float acc4;

__device__ __forceinline__ acc4 mma16(s16x8 a, s16x8 b, acc4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}

// GLDS_SPLIT: how the 8 LDS-DMA pieces (4 x A, 4 x B, 1 KiB per wave each) of the next chunk are dealt to phases 0..2.
#ifndef MMGT_G16_SPLIT
#define MMGT_G16_SPLIT 0
#endif

template <int MODE>
__global__ __launch_bounds__(512, 2) void gemm16_kernel(ADesc ad, const char* __restrict__ W, long bsw, Epi ep, int M, int N,
                                                        int K, int tiles_m, int tiles_n) {
  extern __shared__ __attribute__((aligned(1024))) char smem[];
  typedef bf16_t T;
  constexpr int ESZ = 2, BM = 256, BN = 256, ROWB = 128, BK = 64, NW = 8, NSTAGE = 2;
  constexpr int A_BYTES = BM * ROWB, B_BYTES = BN * ROWB, STAGE_BYTES = A_BYTES + B_BYTES;
  constexpr int CPR = 8, RPD = 8;               // 16-byte chunks per row; rows per 1-KiB DMA piece
  constexpr int GA = BM / RPD / NW, GB = BN / RPD / NW;   // pieces per wave: 4 + 4
  constexpr int P0 = MMGT_G16_SPLIT == 1 ? 4 : 2, P1 = MMGT_G16_SPLIT == 1 ? 8 : 5;   // piece ranges [0,P0) [P0,P1) [P1,8)
  auto swz = [](int row) { return (row >> 1) & 7; };

  const int nwg = tiles_m * tiles_n;
  const int bz = blockIdx.z;
  const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wid >> 2, wc = wid & 3;        // wave group (M half) and N quarter
  const int lm = lane & 15, lq = lane >> 4;

  auto decode = [&](int v, int& tm, int& tn) {  // XCD-aware virtual tile order (see gemm.hip)
    const int q = nwg >> 3, r = nwg & 7, x = v & 7;
    const int t = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (v >> 3);
    tm = t / tiles_n;
    tn = t - tm * tiles_n;
  };

  // ---- LDS-DMA source addressing (as gemm.hip): wave `wid` fills the 8-row pieces g = wid * GA + i of each operand
  const int srow = lane / CPR, spos = lane % CPR;
  const T* a0 = reinterpret_cast<const T*>(ad.src0) + (long)bz * ad.bs0;
  const T* a1 = ad.src1 ? reinterpret_cast<const T*>(ad.src1) + (long)bz * ad.bs1 : nullptr;
  const T* wbase = reinterpret_cast<const T*>(W) + (long)bz * bsw;
  const __amdgpu_buffer_rsrc_t rA0 = dma_rsrc(a0), rA1 = dma_rsrc(a1 ? a1 : a0), rW = dma_rsrc(wbase);
  unsigned aoff[GA];
  // conv: per piece the image's first pixel index and the output pixel (y << 16 | x); the swizzled chunk of piece i is
  // (c0 ^ 4 (i & 1)): row = 32 wid + 8 i + srow, so (row >> 1) & 7 = (4 i + (srow >> 1)) & 7 with srow >> 1 in 0..3
  unsigned cbase[GA], cyx[GA];
  const int c0sw = spos ^ (srow >> 1);
  unsigned woff[GB];
  int p_tap = 0, p_c = 0, a_soff = 0;
  bool a_second = false;
  auto setup = [&](int tm, int tn) {
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
        const int cn = m / hw;
        const int rem = m - cn * hw;
        const int oy = rem / ad.OW;
        cbase[i] = (unsigned)cn * (unsigned)(ad.IH * ad.IW);
        cyx[i] = ((unsigned)oy << 16) | (unsigned)(rem - oy * ad.OW);
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
  auto prep = [&](int ch) {
    if (MODE == 0) {
      a_soff = ch * ROWB;
    } else {
      const int cin = ad.C0 + ad.C1;
      if (p_c == 0 || p_c == ad.C0) {
        const int ky = p_tap / 3, kx = p_tap - ky * 3;
        const int vh = ad.up ? ad.IH * 2 : ad.IH, vw = ad.up ? ad.IW * 2 : ad.IW;
        const bool second = p_c >= ad.C0 && ad.C1 > 0;
        const unsigned cpb = (unsigned)(second ? ad.C1 : ad.C0) * ESZ;   // bytes per pixel of the source tensor (< 2 GiB in all: host check)
        a_second = second;
        a_soff = 0;
#pragma unroll
        for (int i = 0; i < GA; ++i) {
          const int iy = (int)(cyx[i] >> 16) * ad.stride + ky - ad.pad, ix = (int)(cyx[i] & 0xffffu) * ad.stride + kx - ad.pad;
          const bool ok = iy >= 0 && iy < vh && ix >= 0 && ix < vw;
          const int sy = ad.up ? iy >> 1 : iy, sx = ad.up ? ix >> 1 : ix;
          const unsigned off = (cbase[i] + (unsigned)(sy * ad.IW + sx)) * cpb + (unsigned)((c0sw ^ (4 * (i & 1))) << 4);
          aoff[i] = ok ? off : DMA_POISON;
        }
      } else {
        a_soff += ROWB;
      }
      p_c += BK;
      if (p_c == cin) { p_c = 0; ++p_tap; }
    }
  };
  // pieces 0..3 = A, 4..7 = W of the chunk prepared by prep(); [LO, HI) go out now
  auto issue = [&](int stage, int ch, auto LOc, auto HIc) {
    constexpr int LO = decltype(LOc)::value, HI = decltype(HIc)::value;
    char* st = smem + stage * STAGE_BYTES;
    const __amdgpu_buffer_rsrc_t rA = (MODE == 1 && a_second) ? rA1 : rA0;
#pragma unroll
    for (int i = 0; i < GA; ++i)
      if (i >= LO && i < HI) blds16(rA, aoff[i], a_soff, st + (wid * GA + i) * 1024);
#pragma unroll
    for (int i = 0; i < GB; ++i)
      if (GA + i >= LO && GA + i < HI) blds16(rW, woff[i], ch * ROWB, st + A_BYTES + (wid * GB + i) * 1024);
  };
  using std::integral_constant;

  // ---- fragment read addressing: lane (lm, lq) reads row base + 16 t + lm, 16-byte chunk (4 ks + lq) ^ swz; the swizzle
  // of a row depends on lm only ((16 t + lm) >> 1 & 7 == lm >> 1 & 7), so two per-lane offsets (ks = 0, 1) serve every tile
  const int sw = (lm >> 1) & 7;
  const int roff0 = lm * ROWB + ((lq ^ sw) << 4), roff1 = lm * ROWB + (((4 + lq) ^ sw) << 4);
  const int a_base = wr * (BM / 2) * ROWB;                 // this wave's 128 A rows
  const int b_base = A_BYTES + wc * (BN / 4) * ROWB;       // this wave's 64 W rows

  const int nchunks = K / BK;
  const int G = gridDim.x;
  const int my_tiles = (nwg - (int)blockIdx.x + G - 1) / G;
  const int total = my_tiles * nchunks;
  int vt_i = blockIdx.x, ich = 0, gi = 0, sl = 0;   // DMA side
  int sc = 0;                                       // MFMA side: stage to read
  {
    int tm, tn;
    decode(vt_i, tm, tn);
    setup(tm, tn);
  }
  auto advance_dma = [&]() {
    ++gi;
    sl ^= 1;
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
  if (total > 0) {
    prep(ich);
    issue(sl, ich, integral_constant<int, 0>{}, integral_constant<int, 8>{});
    advance_dma();
  }
  wait_vmcnt<0>();
  __builtin_amdgcn_s_barrier();
  if (wr == 1) __builtin_amdgcn_s_barrier();   // the stagger: group 1 runs one barrier behind group 0 from here on

  for (int vt = blockIdx.x; vt < nwg; vt += G) {
    int tm, tn;
    decode(vt, tm, tn);
    const int row0 = tm * BM + wr * (BM / 2), col0 = tn * BN + wc * (BN / 4);

    acc4 acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = (acc4)(0.f);

    // bias[n] + bias2[batch row][n] of the lane's W row n = col0 + 16 j + lm (two bias2 rows at most per tile); they enter
    // the accumulators as one more MFMA after the main loop
    const int b2div = ep.bias2 ? ep.bias2_rows : 0x7fffffff;
    const int b2r0 = (tm * BM) / b2div;
    int mlast = tm * BM + BM - 1;
    if (mlast >= M) mlast = M - 1;
    const bool b2two = mlast / b2div > b2r0;
    float bsum[4][2];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int n = col0 + 16 * j + lm;
      const bool ok = n < N;
      const float b = (ok && ep.bias) ? ep.bias[n] : 0.f;
      bsum[j][0] = b + ((ok && ep.bias2) ? ep.bias2[(long)b2r0 * N + n] : 0.f);
      bsum[j][1] = b + ((ok && ep.bias2 && b2two) ? ep.bias2[(long)(b2r0 + 1) * N + n] : 0.f);
    }

    s16x8 fa[4][2], fb0[2][2], fb1[2][2];   // [tile][ks]
    for (int ch = 0; ch < nchunks; ++ch) {
      const char* st = smem + sc * STAGE_BYTES;
      const bool more = gi < total;
      const int st_i = sl, ch_i = ich;
      if (more) prep(ich);
      auto rdA = [&](int half) {
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          const char* p = st + a_base + (half * 64 + 16 * t) * ROWB;
          fa[t][0] = *reinterpret_cast<const s16x8*>(p + roff0);
          fa[t][1] = *reinterpret_cast<const s16x8*>(p + roff1);
        }
      };
      auto rdB = [&](int half, s16x8 (&f)[2][2]) {
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          const char* p = st + b_base + (half * 32 + 16 * t) * ROWB;
          f[t][0] = *reinterpret_cast<const s16x8*>(p + roff0);
          f[t][1] = *reinterpret_cast<const s16x8*>(p + roff1);
        }
      };
      auto mfmas = [&](auto MHc, auto NHc, s16x8 (&f)[2][2]) {
        constexpr int mh = decltype(MHc)::value, nh = decltype(NHc)::value;
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
          for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int u = 0; u < 2; ++u) acc[mh * 4 + t][nh * 2 + u] = mma16(f[u][ks], fa[t][ks], acc[mh * 4 + t][nh * 2 + u]);
        __builtin_amdgcn_s_setprio(0);
      };
      auto load_done = [&]() {   // this wave's LDS reads have returned: the phase's barrier may release their stage
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
      };
      auto mfma_done = [&]() {
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
      };
      // ---- phase 0
      rdB(0, fb0);
      rdA(0);
      if (more) issue(st_i, ch_i, integral_constant<int, 0>{}, integral_constant<int, P0>{});
      load_done();
      mfmas(integral_constant<int, 0>{}, integral_constant<int, 0>{}, fb0);
      mfma_done();
      // ---- phase 1
      rdB(1, fb1);
      if (more) issue(st_i, ch_i, integral_constant<int, P0>{}, integral_constant<int, P1>{});
      load_done();
      mfmas(integral_constant<int, 0>{}, integral_constant<int, 1>{}, fb1);
      mfma_done();
      // ---- phase 2
      rdA(1);
      if (more) issue(st_i, ch_i, integral_constant<int, P1>{}, integral_constant<int, 8>{});
      load_done();
      mfmas(integral_constant<int, 1>{}, integral_constant<int, 1>{}, fb1);
      mfma_done();
      // ---- phase 3: the next chunk must have landed before the barrier that precedes its first read
      if (more) advance_dma();
      wait_vmcnt<0>();
      load_done();
      mfmas(integral_constant<int, 1>{}, integral_constant<int, 0>{}, fb0);
      mfma_done();
      sc ^= 1;
    }

    // ---- bias as one more MFMA: D[n][m] += sum_c bsum[c][n] * sel[c][m]; values split into bf16 head + tail
    if (ep.bias || ep.bias2) {
      s16x8 fsel[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        int m = row0 + 16 * i + lm;
        if (m >= M) m = M - 1;
        const int c = m / b2div - b2r0;
        const short one = (short)0x3F80;
        const short s0 = (lq == 0 && c == 0) ? one : (short)0, s1 = (lq == 0 && c == 1) ? one : (short)0;
        fsel[i] = (s16x8){s0, s0, s1, s1, 0, 0, 0, 0};
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        s16x8 fbias = (s16x8)(0);
        if (lq == 0) {
#pragma unroll
          for (int c = 0; c < 2; ++c) {
            const bf16_t hi = f32_to_bf16(bsum[j][c]);
            fbias[2 * c] = (short)hi;
            fbias[2 * c + 1] = (short)f32_to_bf16(bsum[j][c] - bf16_to_f32(hi));
          }
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i][j] = mma16(fbias, fsel[i], acc[i][j]);
      }
    }

    // ---- epilogue from registers.  D = W_frag x A_frag: lane (lm, lq) holds output row m = 16 i + lm and, per accumulator
    // tile j, the four columns 16 j + 4 lq + r.  v_permlane16_swap exchanges the odd 16-lane rows of tile j with the even rows
    // of tile j + 1, after which the lane owns 8 consecutive columns 16 j + 16 (lq & 1) + 8 (lq >> 1) .. + 7: one 16-byte
    // store per lane and tile pair, 64 contiguous bytes per output row and instruction.
    T* out = reinterpret_cast<T*>(ep.out) + (long)bz * ep.bso;
    const T* res = ep.residual ? reinterpret_cast<const T*>(ep.residual) + (long)bz * ep.bsr : nullptr;
    const bool geglu = ep.act == 1;
    const int cofs = 16 * (lq & 1) + 8 * (lq >> 1);
    // two m-tiles at a time: their residual vectors are requested together (clamped addresses, no branches around the loads)
    // before any of them is consumed, so the loads overlap instead of paying one memory round trip each
#pragma unroll
    for (int ih = 0; ih < 4; ++ih) {
      u32x4 rv[2][2];
      if (res) {
#pragma unroll
        for (int i4 = 0; i4 < 2; ++i4)
#pragma unroll
          for (int jp = 0; jp < 2; ++jp) {
            int mc = row0 + 16 * (2 * ih + i4) + lm, nc = col0 + 32 * jp + cofs;
            mc = mc < M ? mc : M - 1;
            nc = nc < N ? nc : N - 8;
            rv[i4][jp] = *reinterpret_cast<const u32x4*>(res + (long)mc * ep.ldr + nc);
          }
      }
#pragma unroll
      for (int i4 = 0; i4 < 2; ++i4) {
        const int i = 2 * ih + i4;
        const int m = row0 + 16 * i + lm;
#pragma unroll
        for (int jp = 0; jp < 2; ++jp) {
          if (geglu && jp == 1) continue;        // tiles 2, 3 are the gates of tiles 0, 1 (packed weights: [32 h | 32 gate])
          acc4 x = acc[i][2 * jp], y = acc[i][2 * jp + 1];
          if (geglu) {
            const acc4 gx = acc[i][2], gy = acc[i][3];
            const f32x2 g0 = gelu_erf_f2((f32x2){gx[0], gx[1]}), g1 = gelu_erf_f2((f32x2){gx[2], gx[3]});
            const f32x2 g2 = gelu_erf_f2((f32x2){gy[0], gy[1]}), g3 = gelu_erf_f2((f32x2){gy[2], gy[3]});
            x[0] *= g0[0]; x[1] *= g0[1]; x[2] *= g1[0]; x[3] *= g1[1];
            y[0] *= g2[0]; y[1] *= g2[1]; y[2] *= g3[0]; y[3] *= g3[1];
          }
          float o8[8];
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const auto sw2 = __builtin_amdgcn_permlane16_swap(__float_as_uint(x[r]), __float_as_uint(y[r]), false, false);
            o8[r] = __uint_as_float(sw2[0]);
            o8[4 + r] = __uint_as_float(sw2[1]);
          }
          const int nacc = col0 + 32 * jp + cofs;                     // column in the accumulator's N space
          const long ocol = geglu ? (long)(col0 >> 1) + cofs : (long)nacc;
          if (res) {
            union { u32x4 u; bf16_t e[8]; } r8;
            r8.u = rv[i4][jp];
#pragma unroll
            for (int e = 0; e < 8; ++e) o8[e] += bf16_to_f32(r8.e[e]);
          }
          if (m < M && nacc < N)
            *reinterpret_cast<u32x4*>(out + (long)m * ep.ldo + ocol) =
                (u32x4){pack_bf16x2(o8[0], o8[1]), pack_bf16x2(o8[2], o8[3]), pack_bf16x2(o8[4], o8[5]), pack_bf16x2(o8[6], o8[7])};
        }
      }
    }
  }
  if (wr == 0) __builtin_amdgcn_s_barrier();   // group 0 meets group 1's last barrier
}

template <int MODE>
int launch16(const ADesc& ad, const void* W, long bsw, const Epi& ep, int M, int N, int K, int batch, hipStream_t s) {
  constexpr int BM = 256, BN = 256;
  const int tiles_m = (M + BM - 1) / BM, tiles_n = (N + BN - 1) / BN;
  const size_t lds = (size_t)2 * (BM + BN) * 128;
  auto kern = gemm16_kernel<MODE>;
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
    resident = prop.multiProcessorCount;   // 128 KiB of LDS: one workgroup per CU
  }
  long gx = (resident + batch - 1) / batch;
  gx = (gx + 7) / 8 * 8;
  if (gx > (long)tiles_m * tiles_n) gx = (long)tiles_m * tiles_n;
  dim3 grid((unsigned)gx, 1, batch);
  hipLaunchKernelGGL(kern, grid, dim3(512), lds, s, ad, reinterpret_cast<const char*>(W), bsw, ep, M, N, K, tiles_m, tiles_n);
  MMGT_LAUNCH_CHECK();
  return 0;
}

}  // namespace

// Entry for gemm.hip's dispatcher.  Preconditions (checked there): bf16, vectorised epilogue (ep.fast), act in {none, GEGLU},
// no row scale / alpha / post-scale bias, K % 64 == 0.
int mmgt_gemm16_launch(int mode, const void* adp, const void* W, long bsw, const void* epp, int M, int N, int K, int batch,
                       void* stream) {
  const ADesc& ad = *reinterpret_cast<const ADesc*>(adp);
  const Epi& ep = *reinterpret_cast<const Epi*>(epp);
  hipStream_t s = (hipStream_t)stream;
  return mode == 0 ? launch16<0>(ad, W, bsw, ep, M, N, K, batch, s) : launch16<1>(ad, W, bsw, ep, M, N, K, batch, s);
}
