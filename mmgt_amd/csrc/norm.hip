// GroupNorm (channels-last, per image) and LayerNorm for the MMGT Stage-2 path (gfx950).  Both are HBM-bound: 16-byte
// vector loads, one wave per row so that a lane always owns the same channels (register accumulators, no atomics:
// results are bitwise reproducible run to run).
#include "common.h"
#include "mmgt_hip.h"

namespace {

constexpr int GN_MAXC = 2560;

int g_gn_interleave = -1;   // mmgt_tune("gn_interleave", v): -1 = the default (interleaved), 0 / 1 = force the row mapping (GnRows below)
int g_gn_lpr0 = 8;          // mmgt_tune("gn_lpr0", v): smallest lanes-per-row tried (benchmarking only)
int g_gn_narrow = 2;        // mmgt_tune("gn_narrow", v): 0 = the general two-pass kernels for every shape, 1 = lane-per-vector kernels where the
                            // vectors of a row divide 64 (the VAE), 2 = also for 33..64 vectors, one row per wave (C = 320) (A/B)
int g_gn_slab = 1;        // mmgt_tune("gn_slab", 0 / 1): the register-resident single-read kernel where a slab fits (A/B)
int g_gn_rows = 0;   // mmgt_tune("gn_rows", v): force the rows per workgroup (0 = the measured choice below; benchmarking only)
// Upper bound of the chunk count for an image of HW pixels: what callers size the workspace with (mmgt_groupnorm_chunks).
inline int gn_chunks(int HW) {
  int c = HW >= 2048 ? (HW + 127) / 128 : (HW + 31) / 32;
  return c < 1 ? 1 : (c > 256 ? 256 : c);
}
// Rows per workgroup actually used.  A workgroup's fixed cost (pivot table, the xor tree over its 2 x 40 accumulators, four
// barriers) is as long as ~8 row steps of its loop, so more rows per workgroup amortise it until the grid no longer covers
// the chip -- measured (48 images, tools/bench_norm.py): 4096 x 320: 256 rows 84 us / 128 rows 92 / 64 rows 131 / 512 rows 119;
// 4096 x 640 and x 960: 128 rows best; 1024 x 640: 64 rows 51 us against 60 at 32.
inline int gn_chunks_used(int HW, int C, int NB) {
  int rows = g_gn_rows > 0 ? g_gn_rows : HW >= 2048 ? (C <= 320 ? 256 : 128) : 64;
  // Large images in small batches (the VAE decoder's 256 x 256 and 512 x 512 levels, 4-8 frames): ~1024 workgroups in all measured
  // best -- 8 x 262144 x 128: 2048 rows 395 us against 411 at 1024; 8 x 65536 x 256: 512 rows 210 us / 256 rows 230 / 1024 rows 262.
  if (g_gn_rows <= 0 && HW >= 65536) rows = max(rows, (int)(((long)HW * NB + 1023) / 1024));
  int c = (HW + rows - 1) / rows;
  const int cap = gn_chunks(HW);
  return c < 1 ? 1 : (c > cap ? cap : c);
}

template <typename T>
struct VecIO {
  static constexpr int VEC = 16 / sizeof(T);
  static __device__ __forceinline__ void load(const T* p, float* f) {
    union { u32x4 u; T e[VEC]; } v;
    v.u = *reinterpret_cast<const u32x4*>(p);
#pragma unroll
    for (int i = 0; i < VEC; ++i) f[i] = Elem<T>::ld(&v.e[i]);
  }
  static __device__ __forceinline__ void store(T* p, const float* f) {
    union { u32x4 u; T e[VEC]; } v;
#pragma unroll
    for (int i = 0; i < VEC; ++i) Elem<T>::st(&v.e[i], f[i]);
    *reinterpret_cast<u32x4*>(p) = v.u;
  }
};

// Which rows (pixels) of its image a workgroup of the two chunked passes walks: `wg_rows` rows per step of its four waves.
//   contiguous   rows [chunk * rows_per, (chunk + 1) * rows_per);
//   interleaved  steps chunk, chunk + chunks, chunk + 2 chunks, ... of `wg_rows` rows each: the image's workgroups together stream one
//                contiguous window forward instead of `chunks` streams a fixed distance apart.  Measured (tools/bench_norm.py, GN_IL=0,1):
//                -2..-5 % on every chunked shape (48 x 4096 x 320: 84 -> 81 us, x 960: 303 -> 294; 8 x 262144 x 256: 803 -> 779).
// `partial` holds one entry per (image, chunk) either way; both passes of a launch use the same mapping.
struct GnRows {
  int r0, r1, step;
  __device__ GnRows(int HW, int chunks, int chunk, int wg_rows, int interleave) {
    if (interleave) {
      r0 = chunk * wg_rows;
      r1 = HW;
      step = chunks * wg_rows;
    } else {
      const int rows_per = (HW + chunks - 1) / chunks;
      r0 = chunk * rows_per;
      r1 = min(HW, r0 + rows_per);
      step = wg_rows;
    }
  }
};

// ---- GroupNorm pass 1: per (image, row chunk) partial sum / sum of squares per group, of the values SHIFTED by a per-group
// pivot p_g = x[image][pixel 0][first channel of the group]: var = E[(x - p)^2] - (E[x - p])^2 cancels against (mean - p)^2,
// which is of the order of the variance itself for a pivot drawn from the data, instead of against mean^2 (real SD-1.5
// activations have channels with |mean| >> std; the unshifted single pass lost the digits the fp32 parity mode is meant
// to keep).  One pass, no extra HBM traffic: both kernels read the same pivot.
// LPR lanes cover one pixel row (64 for wide tensors; 16 / 32 for the VAE's 128 / 256-channel tensors so that a wave
// streams 64 / LPR rows at once with every lane busy); a lane always owns the same channels.
template <typename T>
__global__ __launch_bounds__(256) void gn_stats_kernel(const T* __restrict__ x0, int C0, const T* __restrict__ x1, int C1,
                                                       float* __restrict__ partial, int HW, int G, int chunks, int lpr, int interleave) {
  constexpr int VEC = VecIO<T>::VEC;
  constexpr int MAXS = GN_MAXC / (VEC * 64);
  __shared__ float ssum[GN_MAXC], ssq[GN_MAXC];
  const int C = C0 + C1, nvec = C / VEC;
  const int n = blockIdx.y, chunk = blockIdx.x;
  {
    const int cgp = C / G;
    for (int c = threadIdx.x; c < C; c += 256) {      // pivot of channel c's group (ssum doubles as the table)
      const int c0 = (c / cgp) * cgp;
      ssum[c] = c0 < C0 ? Elem<T>::ld(x0 + (long)n * HW * C0 + c0) : Elem<T>::ld(x1 + (long)n * HW * C1 + (c0 - C0));
    }
    __syncthreads();
  }
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int rpw = 64 / lpr, sub = lane / lpr, li = lane - sub * lpr;
  const GnRows rows(HW, chunks, chunk, 4 * rpw, interleave);
  const int r1 = rows.r1;

  float s[MAXS][VEC], q[MAXS][VEC];
#pragma unroll
  for (int i = 0; i < MAXS; ++i)
#pragma unroll
    for (int e = 0; e < VEC; ++e) s[i][e] = q[i][e] = 0.f;

  // the pivots stay in LDS (ssq doubles as their table until the reduction below) and are re-read per row: holding them in
  // registers next to the 2 x MAXS x VEC accumulators cost the kernel half its occupancy (174 VGPRs, 2 waves per SIMD)
  for (int c = threadIdx.x; c < C; c += 256) ssq[c] = ssum[c];
  __syncthreads();
  // two row steps per pass: ten 16-byte loads in flight per lane and ONE read of a vector's pivots for both rows (a pass per row kept five loads
  // in flight and read 32 bytes of pivots from LDS per 16 bytes loaded: 2.3 TB/s on the two-source statistics of the 64 x 64 level against 5.8 for
  // the lane-per-vector kernel of one source)
  for (int r = rows.r0 + wid * rpw + sub; r < r1; r += 2 * rows.step) {
    const long pix = (long)n * HW + r;
    const bool two = r + rows.step < r1;
    const long pix2 = two ? pix + rows.step : pix;
    int keep = 0;
    asm volatile("" : "+v"(keep));                    // opaque zero: keeps the pivot reads inside the row loop
    u32x4 raw[2][MAXS];
#pragma unroll
    for (int i = 0; i < MAXS; ++i) {
      const int vc = li + lpr * i;
      if (vc < nvec) {
        const int c = vc * VEC;
        const bool first = c < C0;
        const T* b0 = first ? x0 + c : x1 + (c - C0);
        const long cs = first ? C0 : C1;
        raw[0][i] = *reinterpret_cast<const u32x4*>(b0 + pix * cs);
        raw[1][i] = *reinterpret_cast<const u32x4*>(b0 + pix2 * cs);
      }
    }
#pragma unroll
    for (int i = 0; i < MAXS; ++i) {
      const int vc = li + lpr * i;
      if (vc < nvec) {
        const int c = vc * VEC;
        float pv[VEC];
#pragma unroll
        for (int e = 0; e < VEC; e += 4) *reinterpret_cast<f32x4*>(&pv[e]) = *reinterpret_cast<const f32x4*>(&ssq[c + e + keep]);
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          if (h == 1 && !two) continue;
          union { u32x4 u; T e[VEC]; } v;
          v.u = raw[h][i];
#pragma unroll
          for (int e = 0; e < VEC; ++e) { const float d = Elem<T>::ld(&v.e[e]) - pv[e]; s[i][e] += d; q[i][e] += d * d; }
        }
      }
    }
  }
  __syncthreads();                                    // every wave is done with the pivot table: ssq is reused below
  // fold the wave's row slots together (fixed xor tree), then the four waves in a fixed order
  for (int o = lpr; o < 64; o <<= 1) {
#pragma unroll
    for (int i = 0; i < MAXS; ++i)
#pragma unroll
      for (int e = 0; e < VEC; ++e) { s[i][e] += __shfl_xor(s[i][e], o); q[i][e] += __shfl_xor(q[i][e], o); }
  }
  for (int w = 0; w < 4; ++w) {
    if (wid == w && sub == 0) {
#pragma unroll
      for (int i = 0; i < MAXS; ++i) {
        const int vc = li + lpr * i;
        if (vc < nvec)
#pragma unroll
          for (int e = 0; e < VEC; ++e) {
            const int c = vc * VEC + e;
            if (w == 0) { ssum[c] = s[i][e]; ssq[c] = q[i][e]; }
            else { ssum[c] += s[i][e]; ssq[c] += q[i][e]; }
          }
      }
    }
    __syncthreads();
  }
  const int cg = C / G;
  if (threadIdx.x < G) {
    float a = 0.f, b = 0.f;
    for (int c = threadIdx.x * cg; c < (threadIdx.x + 1) * cg; ++c) { a += ssum[c]; b += ssq[c]; }
    float* dst = partial + (((long)n * chunks + chunk) * G + threadIdx.x) * 2;
    dst[0] = a;
    dst[1] = b;
  }
}

// Mean / rstd of every group of image n from the per-chunk partial sums: four adjacent lanes per group sum every fourth chunk each
// (independent loads, eight in flight) and are folded by a fixed xor tree -- the one-thread-per-group loop this replaces walked up to
// 256 dependent L2 round trips at the head of EVERY workgroup of the apply pass (~90 us of a 130 us workgroup on the VAE decoder's
// 512 x 512 levels).  Same order in every workgroup and every run: bitwise reproducible.  Needs G <= 64 and blockDim = 256.
__device__ __forceinline__ void gn_group_stats(const float* __restrict__ partial, int n, int chunks, int G, float cnt, float eps,
                                               float piv, float* smean, float* srstd) {
  const int g = threadIdx.x >> 2, j = threadIdx.x & 3;
  float a = 0.f, b = 0.f;
  if (g < G) {
    const float* src = partial + ((long)n * chunks * G + g) * 2;
#pragma unroll 8
    for (int ch = j; ch < chunks; ch += 4) {
      const f32x2 v = *reinterpret_cast<const f32x2*>(src + (long)ch * G * 2);
      a += v[0];
      b += v[1];
    }
  }
  a += __shfl_xor(a, 1); b += __shfl_xor(b, 1);
  a += __shfl_xor(a, 2); b += __shfl_xor(b, 2);
  if (g < G && j == 0) {
    const float dm = a / cnt;
    float var = b / cnt - dm * dm;
    var = var < 0.f ? 0.f : var;
    smean[g] = piv + dm;
    srstd[g] = rsqrtf(var + eps);
  }
}

// ---- GroupNorm pass 2: normalise (+ SiLU) ----
template <typename T>
__global__ __launch_bounds__(256) void gn_apply_kernel(const T* __restrict__ x0, int C0, const T* __restrict__ x1, int C1,
                                                       const float* __restrict__ gamma, const float* __restrict__ beta,
                                                       const float* __restrict__ partial, T* __restrict__ out, int HW,
                                                       int G, int chunks, float eps, int silu, int lpr, int interleave) {
  constexpr int VEC = VecIO<T>::VEC;
  constexpr int MAXS = GN_MAXC / (VEC * 64);
  __shared__ float sscale[GN_MAXC], sshift[GN_MAXC];
  __shared__ float smean[64], srstd[64];
  const int C = C0 + C1, nvec = C / VEC, cg = C / G;
  const int n = blockIdx.y, chunk = blockIdx.x;
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int rpw = 64 / lpr, sub = lane / lpr, li = lane - sub * lpr;
  const GnRows rows(HW, chunks, chunk, 4 * rpw, interleave);
  const int r1 = rows.r1;
  {
    const int g = threadIdx.x >> 2, c0 = (g < G ? g : 0) * cg;     // the pivot gn_stats_kernel shifted this group's values by
    const float piv = c0 < C0 ? Elem<T>::ld(x0 + (long)n * HW * C0 + c0) : Elem<T>::ld(x1 + (long)n * HW * C1 + (c0 - C0));
    gn_group_stats(partial, n, chunks, G, (float)HW * (float)cg, eps, piv, smean, srstd);
  }
  __syncthreads();
  for (int c = threadIdx.x; c < C; c += 256) {
    const int g = c / cg;
    const float sc = srstd[g] * gamma[c];
    sscale[c] = sc;
    sshift[c] = beta[c] - smean[g] * sc;
  }
  __syncthreads();
  for (int r = rows.r0 + wid * rpw + sub; r < r1; r += rows.step) {
    const long pix = (long)n * HW + r;
#pragma unroll
    for (int i = 0; i < MAXS; ++i) {
      const int vc = li + lpr * i;
      if (vc < nvec) {
        const int c = vc * VEC;
        const T* src = c < C0 ? x0 + pix * C0 + c : x1 + pix * C1 + (c - C0);
        float f[VEC];
        VecIO<T>::load(src, f);
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
          float v = f[e] * sscale[c + e] + sshift[c + e];
          f[e] = silu ? silu_f(v) : v;
        }
        VecIO<T>::store(out + pix * C + c, f);
      }
    }
  }
}

// ---- GroupNorm for NARROW rows: C / VEC in {8, 16, 32, 64} vectors per pixel (the VAE's 128 / 256 / 512-channel tensors, up to
// 1.1 GB each: nothing is served from the Infinity Cache, the passes run at what HBM streams).  A lane owns ONE 16-byte channel vector for
// the whole kernel -- its pivots (pass 1) or scale / shift pairs (pass 2) live in registers, 2 x VEC accumulators in all -- and keeps U
// row steps in flight: every load instruction of a wave covers 64 / nvec whole rows (1 KB contiguous), a workgroup step U x 4 of them.
// 40-50 VGPRs against the general kernels' 142 / 51 with their LDS table reads per element: 8 waves per SIMD, ~128 KB in flight per CU.
// Rows are interleaved over the image's workgroups as in GnRows.  Host side: HW % (4 * (64 / nvec) * U) == 0.
template <typename T, int U>
__global__ __launch_bounds__(256) void gn_stats_narrow_kernel(const T* __restrict__ x, float* __restrict__ partial, int C, int HW, int G,
                                                              int chunks) {
  constexpr int VEC = VecIO<T>::VEC;
  __shared__ float ssum[4][64 * VEC], ssq[4][64 * VEC];
  const int nvec = C / VEC, cg = C / G;
  const int n = blockIdx.y, chunk = blockIdx.x;
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int rpw = 64 / nvec, sub = lane / nvec, li = lane - sub * nvec, c = li * VEC;
  const bool act = sub < rpw;                           // nvec that does not divide 64 (C = 320: 40 vectors, one row per wave, 24 lanes idle)
  const T* img = x + (long)n * HW * C;
  float pv[VEC], s[VEC], q[VEC];
#pragma unroll
  for (int e = 0; e < VEC; ++e) {
    pv[e] = Elem<T>::ld(img + ((c + e) / cg) * cg);     // the group's pivot (see gn_stats_kernel)
    s[e] = q[e] = 0.f;
  }
  const int RS = 4 * rpw;
  for (int r = chunk * RS * U + wid * rpw + sub; r < HW; r += chunks * RS * U) {
    if (!act) continue;
    float f[U][VEC];
#pragma unroll
    for (int u = 0; u < U; ++u) VecIO<T>::load(img + (long)(r + u * RS) * C + c, f[u]);
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
      for (int e = 0; e < VEC; ++e) { const float d = f[u][e] - pv[e]; s[e] += d; q[e] += d * d; }
  }
  if (rpw > 1) {                                        // (then nvec divides 64)
    for (int o = nvec; o < 64; o <<= 1) {               // the wave's row slots (fixed xor tree)
#pragma unroll
      for (int e = 0; e < VEC; ++e) { s[e] += __shfl_xor(s[e], o); q[e] += __shfl_xor(q[e], o); }
    }
  }
  if (sub == 0) {
#pragma unroll
    for (int e = 0; e < VEC; ++e) { ssum[wid][c + e] = s[e]; ssq[wid][c + e] = q[e]; }
  }
  __syncthreads();
  for (int ch = threadIdx.x; ch < C; ch += 256) {       // the four waves in a fixed order
    ssum[0][ch] = (ssum[0][ch] + ssum[1][ch]) + (ssum[2][ch] + ssum[3][ch]);
    ssq[0][ch] = (ssq[0][ch] + ssq[1][ch]) + (ssq[2][ch] + ssq[3][ch]);
  }
  __syncthreads();
  if (threadIdx.x < G) {
    float a = 0.f, b = 0.f;
    for (int ch = threadIdx.x * cg; ch < (threadIdx.x + 1) * cg; ++ch) { a += ssum[0][ch]; b += ssq[0][ch]; }
    float* dst = partial + (((long)n * chunks + chunk) * G + threadIdx.x) * 2;
    dst[0] = a;
    dst[1] = b;
  }
}

template <typename T, int U>
__global__ __launch_bounds__(256) void gn_apply_narrow_kernel(const T* __restrict__ x, const float* __restrict__ gamma,
                                                              const float* __restrict__ beta, const float* __restrict__ partial,
                                                              T* __restrict__ out, int C, int HW, int G, int chunks, float eps, int silu) {
  constexpr int VEC = VecIO<T>::VEC;
  __shared__ float smean[64], srstd[64];
  const int nvec = C / VEC, cg = C / G;
  const int n = blockIdx.y, chunk = blockIdx.x;
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int rpw = 64 / nvec, sub = lane / nvec, li = lane - sub * nvec, c = li * VEC;
  const bool act = sub < rpw;
  const T* img = x + (long)n * HW * C;
  T* dst = out + (long)n * HW * C;
  {
    const int g = threadIdx.x >> 2;
    const float piv = Elem<T>::ld(img + (g < G ? g : 0) * cg);
    gn_group_stats(partial, n, chunks, G, (float)HW * (float)cg, eps, piv, smean, srstd);
  }
  __syncthreads();
  float sc[VEC], sh[VEC];
#pragma unroll
  for (int e = 0; e < VEC; ++e) {
    const int g = (c + e) / cg;
    sc[e] = srstd[g] * gamma[c + e];
    sh[e] = beta[c + e] - smean[g] * sc[e];
  }
  const int RS = 4 * rpw;
  for (int r = chunk * RS * U + wid * rpw + sub; r < HW; r += chunks * RS * U) {
    if (!act) continue;
    float f[U][VEC];
#pragma unroll
    for (int u = 0; u < U; ++u) VecIO<T>::load(img + (long)(r + u * RS) * C + c, f[u]);
#pragma unroll
    for (int u = 0; u < U; ++u) {
#pragma unroll
      for (int e = 0; e < VEC; ++e) {
        const float v = f[u][e] * sc[e] + sh[e];
        f[u][e] = silu ? silu_f(v) : v;
      }
      VecIO<T>::store(dst + (long)(r + u * RS) * C + c, f[u]);
    }
  }
}

// ---- GroupNorm pass 2 as TABLES: scale[n][c] = rstd * gamma, shift[n][c] = beta - mean * scale (the head of gn_apply_kernel, the same
// arithmetic), for a consumer that applies v = x * scale + shift itself while it loads x (csrc/rowgemm.hip, norm = 2).
template <typename T>
__global__ __launch_bounds__(256) void gn_affine_kernel(const T* __restrict__ x0, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                        const float* __restrict__ partial, float* __restrict__ scale, float* __restrict__ shift,
                                                        int C, int HW, int G, int chunks, float eps, const T* __restrict__ x1 = nullptr, int C1 = 0) {
  __shared__ float smean[64], srstd[64];
  const int n = blockIdx.x, cg = C / G, C0 = C - C1;
  {
    const int g = threadIdx.x >> 2, c0 = (g < G ? g : 0) * cg;                     // the pivot gn_stats_kernel shifted this group's values by
    const float piv = c0 < C0 ? Elem<T>::ld(x0 + (long)n * HW * C0 + c0) : Elem<T>::ld(x1 + (long)n * HW * C1 + (c0 - C0));
    gn_group_stats(partial, n, chunks, G, (float)HW * (float)cg, eps, piv, smean, srstd);
  }
  __syncthreads();
  for (int c = threadIdx.x; c < C; c += 256) {
    const int g = c / cg;
    const float sc = srstd[g] * gamma[c];
    scale[(long)n * C + c] = sc;
    shift[(long)n * C + c] = beta[c] - smean[g] * sc;
  }
}

// ---- GroupNorm for small images (HW <= 256: the 16x16 and 8x8 levels), one launch: a workgroup owns one image and GPW = 4
// groups (a slab of HW rows x 4 C/G channels, <= 164 KB, re-read from L2), and makes three passes over it -- sum, sum of squared
// deviations from the exact mean, normalise -- with fixed-order reductions through LDS (bitwise reproducible, no pivot needed).
// The two-kernel form above launches only HW / 32 x NB = 96 .. 384 workgroups twice and took 31 .. 85 us on tensors that stream
// in 2 .. 10 us: 1.9 ms of the step for 41 launches.
template <typename T, int GPW>
__global__ __launch_bounds__(256) void gn_small_kernel(const T* __restrict__ x0, int C0, const T* __restrict__ x1, int C1,
                                                       const float* __restrict__ gamma, const float* __restrict__ beta,
                                                       T* __restrict__ out, int HW, int G, float eps, int silu,
                                                       float* __restrict__ tscale = nullptr, float* __restrict__ tshift = nullptr) {
  // tscale / tshift (mmgt_groupnorm_affine2 on small images): the two statistics passes only, the per-(image, channel) tables out instead of the
  // normalised tensor -- for a consumer that applies x * scale + shift itself (csrc/rconv.hip).
  constexpr int VEC = VecIO<T>::VEC;
  __shared__ float part[256 * VEC];
  __shared__ float chan[320];               // per-channel sums of the slab (the dispatch guard admits cw = GPW * C / G <= 320 channels)
  __shared__ float gstat[2][GPW];
  const int C = C0 + C1, cg = C / G, cw = GPW * cg, nvw = cw / VEC;
  const int n = blockIdx.y, c_lo = blockIdx.x * cw;
  const int tid = threadIdx.x;
  const int rp = 256 / nvw;                 // rows per pass
  const int vc = tid % nvw, r0 = tid / nvw;
  const bool active = r0 < rp;
  const int c = c_lo + vc * VEC;            // first channel of this thread's vector (global channel index)
  const T* src = c < C0 ? x0 + (long)n * HW * C0 + c : x1 + (long)n * HW * C1 + (c - C0);
  const long rstride = c < C0 ? C0 : C1;
  int gid[VEC];
#pragma unroll
  for (int e = 0; e < VEC; ++e) gid[e] = (vc * VEC + e) / cg;

  auto reduce = [&](const float (&acc)[VEC], int which) {   // per-group total of acc over the slab -> gstat[which][0..3]
#pragma unroll
    for (int e = 0; e < VEC; ++e) part[tid * VEC + e] = active ? acc[e] : 0.f;
    __syncthreads();
    for (int ch = tid; ch < cw; ch += 256) {
      const int v = ch / VEC, e = ch % VEC;
      float s = 0.f;
      for (int r = 0; r < rp; ++r) s += part[(r * nvw + v) * VEC + e];
      chan[ch] = s;
    }
    __syncthreads();
    if (tid < GPW) {
      float s = 0.f;
      for (int k = 0; k < cg; ++k) s += chan[tid * cg + k];
      gstat[which][tid] = s;
    }
    __syncthreads();
  };

  float acc[VEC];
#pragma unroll
  for (int e = 0; e < VEC; ++e) acc[e] = 0.f;
  if (active)
#pragma unroll 4
    for (int r = r0; r < HW; r += rp) {
      float f[VEC];
      VecIO<T>::load(src + r * rstride, f);
#pragma unroll
      for (int e = 0; e < VEC; ++e) acc[e] += f[e];
    }
  reduce(acc, 0);
  const float cnt = (float)HW * (float)cg;
  float mean[VEC];
#pragma unroll
  for (int e = 0; e < VEC; ++e) { mean[e] = gstat[0][gid[e]] / cnt; acc[e] = 0.f; }
  if (active)
#pragma unroll 4
    for (int r = r0; r < HW; r += rp) {
      float f[VEC];
      VecIO<T>::load(src + r * rstride, f);
#pragma unroll
      for (int e = 0; e < VEC; ++e) { const float d = f[e] - mean[e]; acc[e] += d * d; }
    }
  reduce(acc, 1);
  if (active) {
    float sc[VEC], sh[VEC];
#pragma unroll
    for (int e = 0; e < VEC; ++e) {
      const float rstd = rsqrtf(gstat[1][gid[e]] / cnt + eps);
      sc[e] = rstd * gamma[c + e];
      sh[e] = beta[c + e] - mean[e] * sc[e];
    }
    if (tscale) {
      if (r0 == 0) {
#pragma unroll
        for (int e = 0; e < VEC; ++e) { tscale[(long)n * C + c + e] = sc[e]; tshift[(long)n * C + c + e] = sh[e]; }
      }
      return;
    }
    T* dst = out + (long)n * HW * C + c;
#pragma unroll 4
    for (int r = r0; r < HW; r += rp) {
      float f[VEC];
      VecIO<T>::load(src + r * rstride, f);
#pragma unroll
      for (int e = 0; e < VEC; ++e) {
        const float v = f[e] * sc[e] + sh[e];
        f[e] = silu ? silu_f(v) : v;
      }
      VecIO<T>::store(dst + (long)r * C, f);
    }
  }
}

// ---- GroupNorm with the slab in REGISTERS: gn_small_kernel's workgroup (one image x `gpw` groups), thread mapping, arithmetic and reduction
// orders -- bitwise its results -- but the slab is read from memory ONCE: thread (vector vc of a row, row r0 + i rp) keeps its <= NR vectors
// (raw, 4 registers each) and makes the three passes over them.  gn_small_kernel pays three dependent memory round trips per launch, and
// the two-kernel form (HW > 256) reads the tensor twice: 48 x 256 x 1280 took 31.6 us in the step and 48 x 1024 x 640 49.9 us, for 63 / 126 MB
// moved (profiles/r6/opshapes_r6b.txt).  Slabs of up to 22 vectors per thread: every single-source level of the UNet below 64 x 64.
template <typename T, int NR, int NTHR = 256>
__global__ __launch_bounds__(NTHR) void gn_slab_kernel(const T* __restrict__ x0, int C0, const T* __restrict__ x1, int C1,
                                                      const float* __restrict__ gamma, const float* __restrict__ beta,
                                                      T* __restrict__ out, int HW, int G, int gpw, float eps, int silu,
                                                      float* __restrict__ tscale, float* __restrict__ tshift) {
  constexpr int VEC = VecIO<T>::VEC;
  __shared__ float part[NTHR * VEC];
  __shared__ float chan[320];
  __shared__ float gstat[2][4];
  const int C = C0 + C1, cg = C / G, cw = gpw * cg, nvw = cw / VEC;
  // The slabs of one image share their 128-byte lines (a slab's pixel is cw * sizeof(T) = 40 .. 320 bytes of a C * sizeof(T) row): workgroups are
  // dealt to the XCDs round-robin, so the linear id is turned into a virtual one whose consecutive values sit on ONE XCD -- an image's slabs then
  // fetch each line into one L2, at about the same time (as csrc/gemm16.hip's tile order).
  int n = blockIdx.y, sl = blockIdx.x;
  {
    const int nsl = gridDim.x, tot = nsl * gridDim.y;
    if ((tot & 7) == 0) {
      const int lin = blockIdx.y * nsl + blockIdx.x, v = (lin & 7) * (tot >> 3) + (lin >> 3);
      n = v / nsl;
      sl = v - n * nsl;
    }
  }
  const int c_lo = sl * cw;
  const int tid = threadIdx.x;
  const int rp = NTHR / nvw;                // rows per pass
  const int vc = tid % nvw, r0 = tid / nvw;
  const bool active = r0 < rp;
  const int c = c_lo + vc * VEC;
  const T* src = c < C0 ? x0 + (long)n * HW * C0 + c : x1 + (long)n * HW * C1 + (c - C0);
  const long rstride = c < C0 ? C0 : C1;
  int gid[VEC];
#pragma unroll
  for (int e = 0; e < VEC; ++e) gid[e] = (vc * VEC + e) / cg;

  u32x4 raw[NR];
#pragma unroll
  for (int i = 0; i < NR; ++i) {
    const int r = r0 + i * rp;
    raw[i] = (active && r < HW) ? *reinterpret_cast<const u32x4*>(src + r * rstride) : (u32x4)(0u);
  }
  auto dec = [](const u32x4& u, float* f) {
    union { u32x4 u; T e[VEC]; } v;
    v.u = u;
#pragma unroll
    for (int e = 0; e < VEC; ++e) f[e] = Elem<T>::ld(&v.e[e]);
  };
  auto reduce = [&](const float (&acc)[VEC], int which) {   // (gn_small_kernel's, order for order)
#pragma unroll
    for (int e = 0; e < VEC; ++e) part[tid * VEC + e] = active ? acc[e] : 0.f;
    __syncthreads();
    for (int ch = tid; ch < cw; ch += NTHR) {
      const int v = ch / VEC, e = ch % VEC;
      float s = 0.f;
      for (int r = 0; r < rp; ++r) s += part[(r * nvw + v) * VEC + e];
      chan[ch] = s;
    }
    __syncthreads();
    if (tid < gpw) {
      float s = 0.f;
      for (int k = 0; k < cg; ++k) s += chan[tid * cg + k];
      gstat[which][tid] = s;
    }
    __syncthreads();
  };

  float acc[VEC];
#pragma unroll
  for (int e = 0; e < VEC; ++e) acc[e] = 0.f;
#pragma unroll
  for (int i = 0; i < NR; ++i)
    if (active && r0 + i * rp < HW) {
      float f[VEC];
      dec(raw[i], f);
#pragma unroll
      for (int e = 0; e < VEC; ++e) acc[e] += f[e];
    }
  reduce(acc, 0);
  const float cnt = (float)HW * (float)cg;
  float mean[VEC];
#pragma unroll
  for (int e = 0; e < VEC; ++e) { mean[e] = gstat[0][gid[e]] / cnt; acc[e] = 0.f; }
#pragma unroll
  for (int i = 0; i < NR; ++i)
    if (active && r0 + i * rp < HW) {
      float f[VEC];
      dec(raw[i], f);
#pragma unroll
      for (int e = 0; e < VEC; ++e) { const float d = f[e] - mean[e]; acc[e] += d * d; }
    }
  reduce(acc, 1);
  if (active) {
    float sc[VEC], sh[VEC];
#pragma unroll
    for (int e = 0; e < VEC; ++e) {
      const float rstd = rsqrtf(gstat[1][gid[e]] / cnt + eps);
      sc[e] = rstd * gamma[c + e];
      sh[e] = beta[c + e] - mean[e] * sc[e];
    }
    if (tscale) {
      if (r0 == 0) {
#pragma unroll
        for (int e = 0; e < VEC; ++e) { tscale[(long)n * C + c + e] = sc[e]; tshift[(long)n * C + c + e] = sh[e]; }
      }
      return;
    }
    T* dst = out + (long)n * HW * C + c;
#pragma unroll
    for (int i = 0; i < NR; ++i) {
      const int r = r0 + i * rp;
      if (r < HW) {
        float f[VEC];
        dec(raw[i], f);
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
          const float v = f[e] * sc[e] + sh[e];
          f[e] = silu ? silu_f(v) : v;
        }
        VecIO<T>::store(dst + (long)r * C, f);
      }
    }
  }
}

// the slab kernel's shape rule: whole vectors per slab, <= 22 vectors per thread; small slabs take as many groups as leave <= 6 vectors per thread
// (a workgroup's fixed cost -- two reductions, six barriers -- is as long as a few of its loads: 48 x 64 x 1280 as 1536 one-group workgroups
// 12.1 us, as 384 four-group ones see profiles/r6/bench_gn_slab_r6.txt), larger ones the fewest groups that fit; 0 = does not fit
inline int gn_slab_gpw(int C, int G, int HW, int vec, int* nr, int* nthr) {
  const int cg = C / G;
  auto need = [&](int gpw, int threads) {
    const int cw = gpw * cg;
    if (G % gpw || cw % vec || cw > 320 || cw / vec > 256) return 1 << 30;
    const int rp = threads / (cw / vec);
    return (HW + rp - 1) / rp;
  };
  int pick = 0;
  *nthr = 256;
  for (int gpw = 4; gpw >= 1 && !pick; gpw >>= 1)
    if (need(gpw, 256) <= 6) pick = gpw;
  for (int gpw = 1; gpw <= 4 && !pick; gpw <<= 1)
    if (need(gpw, 256) <= 22) pick = gpw;
  if (!pick) {
    // wide groups (the skip concatenations of the 32 x 32 level: 1920 / 960 channels = 60 / 30 per group, 15 vectors per slab row): 1024 threads hold
    // the slab in <= 16 vectors each (128 registers per lane at 16 waves per workgroup)
    for (int gpw = 1; gpw <= 4 && !pick; gpw <<= 1)
      if (need(gpw, 1024) <= 16) pick = gpw;
    if (!pick) return 0;
    *nthr = 1024;
    *nr = 16;
    return pick;
  }
  const int n = need(pick, 256);
  *nr = n <= 2 ? 2 : n <= 4 ? 4 : n <= 6 ? 6 : n <= 12 ? 12 : 22;
  return pick;
}
template <typename T>
void gn_slab_launch(int nr, int nthr, dim3 grid, hipStream_t s, const void* x0, int C0, const void* x1, int C1, const float* gamma, const float* beta, void* out,
                    int HW, int G, int gpw, float eps, int silu, float* tscale, float* tshift) {
#define GN_SLAB(NR_, NT_) hipLaunchKernelGGL((gn_slab_kernel<T, NR_, NT_>), grid, dim3(NT_), 0, s, (const T*)x0, C0, (const T*)x1, C1, gamma, beta, (T*)out, HW, G, gpw, \
                                             eps, silu, tscale, tshift)
  if (nthr == 1024) GN_SLAB(16, 1024);
  else if (nr == 2) GN_SLAB(2, 256); else if (nr == 4) GN_SLAB(4, 256); else if (nr == 6) GN_SLAB(6, 256); else if (nr == 12) GN_SLAB(12, 256); else GN_SLAB(22, 256);
#undef GN_SLAB
}

// ---- LayerNorm: LPR lanes per row (8 .. 64, so all 64 lanes stream 16-byte vectors even at C = 320), the row slice in
// registers, exact two-pass statistics, reductions by xor-shuffles inside the LPR-lane group ----
template <typename T>
__global__ __launch_bounds__(256) void ln_kernel(const T* __restrict__ x, long ldx, const float* __restrict__ gamma,
                                                 const float* __restrict__ beta, float eps, const float* __restrict__ pe,
                                                 int pe_div, int pe_mod, T* __restrict__ out, long ldo, int rows, int C,
                                                 int lpr, int vpl) {
  constexpr int VEC = VecIO<T>::VEC;
  constexpr int MAXV = 5;
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int rpw = 64 / lpr;                       // rows per wave
  const int sub = lane / lpr, li = lane - sub * lpr;
  const long row = ((long)blockIdx.x * 4 + wid) * rpw + sub;
  const bool ok = row < rows;
  const long rrow = ok ? row : rows - 1;
  float f[MAXV][VEC];
  float sum = 0.f;
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    if (i < vpl) {
      VecIO<T>::load(x + rrow * ldx + (li + lpr * i) * VEC, f[i]);
#pragma unroll
      for (int e = 0; e < VEC; ++e) sum += f[i][e];
    }
  }
  for (int o = lpr >> 1; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
  const float mean = sum / (float)C;
  float sq = 0.f;
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    if (i < vpl)
#pragma unroll
      for (int e = 0; e < VEC; ++e) { const float d = f[i][e] - mean; sq += d * d; }
  }
  for (int o = lpr >> 1; o > 0; o >>= 1) sq += __shfl_xor(sq, o);
  const float rstd = rsqrtf(sq / (float)C + eps);
  if (!ok) return;
  const float* perow = pe ? pe + (long)((row / pe_div) % pe_mod) * C : nullptr;
  // pe == nullptr with pe_mod > 1: beta is a [pe_mod][C] table (beta + pe folded once by the host), one vector load less
  if (!pe && pe_mod > 1) beta += (long)((row / pe_div) % pe_mod) * C;
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    if (i < vpl) {
      const int c = (li + lpr * i) * VEC;
      float g[VEC], b[VEC];
#pragma unroll
      for (int e = 0; e < VEC; e += 4) {
        *reinterpret_cast<f32x4*>(&g[e]) = *reinterpret_cast<const f32x4*>(gamma + c + e);
        *reinterpret_cast<f32x4*>(&b[e]) = *reinterpret_cast<const f32x4*>(beta + c + e);
      }
#pragma unroll
      for (int e = 0; e < VEC; ++e) {
        float v = (f[i][e] - mean) * rstd * g[e] + b[e];
        if (perow) v += perow[c + e];
        f[i][e] = v;
      }
      VecIO<T>::store(out + row * ldo + c, f[i]);
    }
  }
}

}  // namespace

void mmgt_gn_set_rows(int v) { g_gn_rows = v; }
void mmgt_gn_set_interleave(int v) { g_gn_interleave = v; }
void mmgt_gn_set_lpr0(int v) { g_gn_lpr0 = v; }
void mmgt_gn_set_narrow(int v) { g_gn_narrow = v; }
void mmgt_gn_set_slab(int v) { g_gn_slab = v; }

namespace {
constexpr int GN_NARROW_U = 4;
// the lane-per-vector kernels (gn_*_narrow_kernel): one source, nvec = C / VEC <= 64 vectors per row that either divide 64 or leave
// one row per wave (C = 320: 40 lanes of 64 -- g_gn_narrow >= 2), whole workgroup steps
inline bool gn_narrow_fits(int nvec, int C1, int HW) {
  return g_gn_narrow && C1 == 0 && nvec >= 8 && nvec <= 64 && (64 % nvec == 0 || (nvec > 32 && g_gn_narrow >= 2)) &&
         HW % (4 * (64 / nvec) * GN_NARROW_U) == 0;
}
inline int gn_narrow_chunks(int HW, int NB, int nvec) {
  const int step = 4 * (64 / nvec) * GN_NARROW_U;
  int rows = g_gn_rows > 0 ? g_gn_rows : 64;     // 8 x 4096 x 512: 64-128 rows 29 us, 256 rows 46 us; larger images: ~2048 workgroups in all
  if (g_gn_rows <= 0) rows = max(rows, (int)(((long)HW * NB + 2047) / 2048));
  rows = (rows + step - 1) / step * step;
  int chunks = (HW + rows - 1) / rows;
  const int cap = gn_chunks(HW);
  return chunks < 1 ? 1 : (chunks > cap ? cap : chunks);
}
}  // namespace
extern "C" int mmgt_groupnorm_chunks(int HW) { return gn_chunks(HW); }

extern "C" int mmgt_groupnorm_nhwc(const void* x0, int C0, const void* x1, int C1, const float* gamma, const float* beta,
                                   void* out, float* workspace, int NB, int HW, int G, float eps, int silu, int dtype,
                                   void* stream) {
  MMGT_CHECK(x0 && gamma && beta && out && workspace, "groupnorm: null pointer");
  MMGT_CHECK(dtype == MMGT_F32 || dtype == MMGT_BF16, "groupnorm: bad dtype %d", dtype);
  MMGT_CHECK((x1 != nullptr) == (C1 > 0), "groupnorm: x1/C1 mismatch");
  const int C = C0 + C1, vec = dtype == MMGT_BF16 ? 8 : 4;
  MMGT_CHECK(C <= GN_MAXC && C % G == 0 && G <= 64 && C0 % vec == 0 && C1 % vec == 0,
             "groupnorm: unsupported channels C0=%d C1=%d G=%d", C0, C1, G);
  MMGT_CHECK(NB > 0 && HW > 0 && NB <= 65535, "groupnorm: bad NB=%d HW=%d", NB, HW);
  hipStream_t s = (hipStream_t)stream;
  {
    int nr = 0, nthr = 256;
    const int gpw = g_gn_slab ? gn_slab_gpw(C, G, HW, vec, &nr, &nthr) : 0;
    if (gpw) {
      dim3 grid(G / gpw, NB);
      if (dtype == MMGT_BF16) gn_slab_launch<bf16_t>(nr, nthr, grid, s, x0, C0, x1, C1, gamma, beta, out, HW, G, gpw, eps, silu, nullptr, nullptr);
      else gn_slab_launch<float>(nr, nthr, grid, s, x0, C0, x1, C1, gamma, beta, out, HW, G, gpw, eps, silu, nullptr, nullptr);
      MMGT_LAUNCH_CHECK();
      return 0;
    }
  }
  {
    const int cg = C / G;
    // 4 groups per workgroup on the 8x8 level, 2 on the 16x16 level (more, shorter slabs: the passes are latency-bound)
    const int gpw = (HW > 64 && G % 2 == 0 && (2 * cg) % vec == 0) ? 2 : 4;
    const int cw = gpw * cg;
    if (HW <= 256 && G % gpw == 0 && cw % vec == 0 && cw <= 320 && cw / vec <= 256 && C0 % vec == 0) {
      dim3 grid(G / gpw, NB);
#define GN_SMALL(T_, GPW_) hipLaunchKernelGGL((gn_small_kernel<T_, GPW_>), grid, dim3(256), 0, s, (const T_*)x0, C0, (const T_*)x1, C1, gamma, \
                                              beta, (T_*)out, HW, G, eps, silu)
      if (dtype == MMGT_BF16) { if (gpw == 2) GN_SMALL(bf16_t, 2); else GN_SMALL(bf16_t, 4); }
      else { if (gpw == 2) GN_SMALL(float, 2); else GN_SMALL(float, 4); }
#undef GN_SMALL
      MMGT_LAUNCH_CHECK();
      return 0;
    }
  }
  const int nvec = C / vec;
  {
    // narrow rows (the VAE; C = 320 of the UNet): a lane per channel vector, see gn_stats_narrow_kernel
    constexpr int U = GN_NARROW_U;
    if (gn_narrow_fits(nvec, C1, HW)) {
      const int chunks = gn_narrow_chunks(HW, NB, nvec);
      dim3 grid(chunks, NB);
      if (dtype == MMGT_BF16) {
        hipLaunchKernelGGL((gn_stats_narrow_kernel<bf16_t, U>), grid, dim3(256), 0, s, (const bf16_t*)x0, workspace, C, HW, G, chunks);
        hipLaunchKernelGGL((gn_apply_narrow_kernel<bf16_t, U>), grid, dim3(256), 0, s, (const bf16_t*)x0, gamma, beta, workspace,
                           (bf16_t*)out, C, HW, G, chunks, eps, silu);
      } else {
        hipLaunchKernelGGL((gn_stats_narrow_kernel<float, U>), grid, dim3(256), 0, s, (const float*)x0, workspace, C, HW, G, chunks);
        hipLaunchKernelGGL((gn_apply_narrow_kernel<float, U>), grid, dim3(256), 0, s, (const float*)x0, gamma, beta, workspace,
                           (float*)out, C, HW, G, chunks, eps, silu);
      }
      MMGT_LAUNCH_CHECK();
      return 0;
    }
  }
  const int chunks = gn_chunks_used(HW, C, NB);
  dim3 grid(chunks, NB);
  // lanes per pixel row: the fewest of 8 / 16 / 32 / 64 whose lanes x MAXS vectors tile the row exactly, so that every lane
  // streams a vector in every load instruction: C = 320 -> 8 lanes x 5 vectors (8 rows per wave instruction), 640 -> 16,
  // 1280 -> 32, 2560 -> 64 (with 64 lanes per row the 40 vectors of a 320-channel row left 24 lanes idle).
  const int maxs = GN_MAXC / (vec * 64);
  int lpr = g_gn_lpr0;
  while (lpr < 64 && (lpr * maxs < nvec || nvec % lpr != 0)) lpr <<= 1;   // ragged rows (C = 960, 1920) measured faster at 64
  const int il = g_gn_interleave >= 0 ? g_gn_interleave : 1;
  if (dtype == MMGT_BF16) {
    hipLaunchKernelGGL(gn_stats_kernel<bf16_t>, grid, dim3(256), 0, s, (const bf16_t*)x0, C0, (const bf16_t*)x1, C1,
                       workspace, HW, G, chunks, lpr, il);
    hipLaunchKernelGGL(gn_apply_kernel<bf16_t>, grid, dim3(256), 0, s, (const bf16_t*)x0, C0, (const bf16_t*)x1, C1, gamma,
                       beta, workspace, (bf16_t*)out, HW, G, chunks, eps, silu, lpr, il);
  } else {
    hipLaunchKernelGGL(gn_stats_kernel<float>, grid, dim3(256), 0, s, (const float*)x0, C0, (const float*)x1, C1,
                       workspace, HW, G, chunks, lpr, il);
    hipLaunchKernelGGL(gn_apply_kernel<float>, grid, dim3(256), 0, s, (const float*)x0, C0, (const float*)x1, C1, gamma,
                       beta, workspace, (float*)out, HW, G, chunks, eps, silu, lpr, il);
  }
  MMGT_LAUNCH_CHECK();
  return 0;
}

// The same for the channel concatenation of TWO tensors (x0 | x1: a resnet's input behind a skip connection, unet_3d_blocks.py:941-969), whose
// groups may straddle the seam (1280 + 640 channels in 32 groups of 60); x1 == null: one tensor.
extern "C" int mmgt_groupnorm_affine2(const void* x, int C0, const void* x1, int C1, const float* gamma, const float* beta, float* workspace,
                                      float* scale, float* shift, int NB, int HW, int G, float eps, int dtype, void* stream) {
  MMGT_CHECK(x && gamma && beta && workspace && scale && shift, "groupnorm_affine: null pointer");
  MMGT_CHECK(dtype == MMGT_F32 || dtype == MMGT_BF16, "groupnorm_affine: bad dtype %d", dtype);
  MMGT_CHECK((x1 != nullptr) == (C1 > 0), "groupnorm_affine: x1 / C1 mismatch");
  const int vec = dtype == MMGT_BF16 ? 8 : 4, C = C0 + C1;
  MMGT_CHECK(C <= GN_MAXC && C % G == 0 && G <= 64 && C0 % vec == 0 && C1 % vec == 0, "groupnorm_affine: unsupported channels C=%d + %d G=%d", C0, C1, G);
  MMGT_CHECK(NB > 0 && HW > 0 && NB <= 65535, "groupnorm_affine: bad NB=%d HW=%d", NB, HW);
  hipStream_t s = (hipStream_t)stream;
  {
    int nr = 0, nthr = 256;
    const int gpw = g_gn_slab ? gn_slab_gpw(C, G, HW, vec, &nr, &nthr) : 0;      // the slab in registers: one read of the tensor, tables out
    if (gpw) {
      dim3 grid(G / gpw, NB);
      if (dtype == MMGT_BF16) gn_slab_launch<bf16_t>(nr, nthr, grid, s, x, C0, x1, C1, gamma, beta, nullptr, HW, G, gpw, eps, 0, scale, shift);
      else gn_slab_launch<float>(nr, nthr, grid, s, x, C0, x1, C1, gamma, beta, nullptr, HW, G, gpw, eps, 0, scale, shift);
      MMGT_LAUNCH_CHECK();
      return 0;
    }
  }
  {
    // small images (the 16 x 16 and 8 x 8 levels): the single-launch kernel's two statistics passes, tables out (see mmgt_groupnorm_nhwc)
    const int cg = C / G;
    const int gpw = (HW > 64 && G % 2 == 0 && (2 * cg) % vec == 0) ? 2 : 4;
    const int cw = gpw * cg;
    if (HW <= 256 && G % gpw == 0 && cw % vec == 0 && cw <= 320 && cw / vec <= 256 && C0 % vec == 0) {
      dim3 grid(G / gpw, NB);
#define GN_SMALL_T(T_, GPW_) hipLaunchKernelGGL((gn_small_kernel<T_, GPW_>), grid, dim3(256), 0, s, (const T_*)x, C0, (const T_*)x1, C1, gamma, beta, \
                                                (T_*)nullptr, HW, G, eps, 0, scale, shift)
      if (dtype == MMGT_BF16) { if (gpw == 2) GN_SMALL_T(bf16_t, 2); else GN_SMALL_T(bf16_t, 4); }
      else { if (gpw == 2) GN_SMALL_T(float, 2); else GN_SMALL_T(float, 4); }
#undef GN_SMALL_T
      MMGT_LAUNCH_CHECK();
      return 0;
    }
  }
  const int nvec = C / vec, maxs = GN_MAXC / (vec * 64);
  if (!x1 && gn_narrow_fits(nvec, 0, HW)) {
    const int chunks = gn_narrow_chunks(HW, NB, nvec);
    dim3 grid(chunks, NB);
    if (dtype == MMGT_BF16) {
      hipLaunchKernelGGL((gn_stats_narrow_kernel<bf16_t, GN_NARROW_U>), grid, dim3(256), 0, s, (const bf16_t*)x, workspace, C, HW, G, chunks);
      hipLaunchKernelGGL(gn_affine_kernel<bf16_t>, dim3(NB), dim3(256), 0, s, (const bf16_t*)x, gamma, beta, workspace, scale, shift, C, HW, G, chunks, eps);
    } else {
      hipLaunchKernelGGL((gn_stats_narrow_kernel<float, GN_NARROW_U>), grid, dim3(256), 0, s, (const float*)x, workspace, C, HW, G, chunks);
      hipLaunchKernelGGL(gn_affine_kernel<float>, dim3(NB), dim3(256), 0, s, (const float*)x, gamma, beta, workspace, scale, shift, C, HW, G, chunks, eps);
    }
    MMGT_LAUNCH_CHECK();
    return 0;
  }
  const int chunks = gn_chunks_used(HW, C, NB);
  dim3 grid(chunks, NB);
  int lpr = g_gn_lpr0;
  while (lpr < 64 && (lpr * maxs < nvec || nvec % lpr != 0)) lpr <<= 1;
  const int il = g_gn_interleave >= 0 ? g_gn_interleave : 1;
  if (dtype == MMGT_BF16) {
    hipLaunchKernelGGL(gn_stats_kernel<bf16_t>, grid, dim3(256), 0, s, (const bf16_t*)x, C0, (const bf16_t*)x1, C1, workspace, HW, G, chunks, lpr, il);
    hipLaunchKernelGGL(gn_affine_kernel<bf16_t>, dim3(NB), dim3(256), 0, s, (const bf16_t*)x, gamma, beta, workspace, scale, shift, C, HW, G, chunks, eps,
                       (const bf16_t*)x1, C1);
  } else {
    hipLaunchKernelGGL(gn_stats_kernel<float>, grid, dim3(256), 0, s, (const float*)x, C0, (const float*)x1, C1, workspace, HW, G, chunks, lpr, il);
    hipLaunchKernelGGL(gn_affine_kernel<float>, dim3(NB), dim3(256), 0, s, (const float*)x, gamma, beta, workspace, scale, shift, C, HW, G, chunks, eps,
                       (const float*)x1, C1);
  }
  MMGT_LAUNCH_CHECK();
  return 0;
}

extern "C" int mmgt_groupnorm_affine(const void* x, int C, const float* gamma, const float* beta, float* workspace, float* scale,
                                     float* shift, int NB, int HW, int G, float eps, int dtype, void* stream) {
  return mmgt_groupnorm_affine2(x, C, nullptr, 0, gamma, beta, workspace, scale, shift, NB, HW, G, eps, dtype, stream);
}

extern "C" int mmgt_layernorm(const void* x, long ldx, const float* gamma, const float* beta, float eps, const float* pe,
                              int pe_div, int pe_mod, void* out, long ldo, int rows, int C, int dtype, void* stream) {
  MMGT_CHECK(x && gamma && beta && out, "layernorm: null pointer");
  MMGT_CHECK(dtype == MMGT_F32 || dtype == MMGT_BF16, "layernorm: bad dtype %d", dtype);
  const int vec = dtype == MMGT_BF16 ? 8 : 4;
  MMGT_CHECK(C > 0 && C % vec == 0 && ldx % vec == 0 && ldo % vec == 0, "layernorm: unsupported C=%d", C);
  MMGT_CHECK(rows > 0, "layernorm: no rows");
  MMGT_CHECK(pe_div > 0 && pe_mod > 0, "layernorm: bad pe_div/pe_mod");
  MMGT_CHECK(((uintptr_t)x % 16) == 0 && ((uintptr_t)out % 16) == 0 && ((uintptr_t)gamma % 16) == 0 &&
                 ((uintptr_t)beta % 16) == 0, "layernorm: pointers must be 16-byte aligned");
  const int nvec = C / vec;
  int lpr = g_gn_lpr0;
  while (lpr < 64 && (nvec % lpr != 0 || nvec / lpr > 5)) lpr <<= 1;
  MMGT_CHECK(nvec % lpr == 0 && nvec / lpr <= 5, "layernorm: C=%d does not map onto 8..64 lanes x <=5 vectors", C);
  const int vpl = nvec / lpr, rows_per_block = 4 * (64 / lpr);
  dim3 grid((rows + rows_per_block - 1) / rows_per_block);
  hipStream_t s = (hipStream_t)stream;
  if (dtype == MMGT_BF16)
    hipLaunchKernelGGL(ln_kernel<bf16_t>, grid, dim3(256), 0, s, (const bf16_t*)x, ldx, gamma, beta, eps, pe, pe_div,
                       pe_mod, (bf16_t*)out, ldo, rows, C, lpr, vpl);
  else
    hipLaunchKernelGGL(ln_kernel<float>, grid, dim3(256), 0, s, (const float*)x, ldx, gamma, beta, eps, pe, pe_div, pe_mod,
                       (float*)out, ldo, rows, C, lpr, vpl);
  MMGT_LAUNCH_CHECK();
  return 0;
}
