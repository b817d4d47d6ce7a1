// GEMM / implicit-GEMM conv3x3 for the MMGT Stage-2 path (gfx950).
//
//   out[M, N] = epilogue( A[M, K] * W[N, K]^T )          W in torch nn.Linear layout ([out, in], K contiguous)
//
// A is either a dense row-major matrix (Linear, 1x1 conv on channels-last tokens) or the implicit im2col view of a
// channels-last (N, H, W, C) tensor for a 3x3 / pad 1 convolution (stride 1 or 2, optional fused nearest-2x upsample of
// the input, optional second source tensor = fused channel concat of the UNet skip connection).  K ordering of the conv
// view is (ky, kx, cin) with cin fastest, matching weights pre-packed as [Cout][3][3][Cin].
//
// Tile: 128 x 128 per 256-thread workgroup (4 waves as 2 x 2, 64 x 64 per wave = 2 x 2 MFMA 32x32 tiles), K chunk of
// 128 bytes per row (64 bf16 / 32 fp32), register-staged global -> LDS double buffering with one barrier per chunk,
// LDS rows padded to 144 B so the ds_read_b128 fragment reads are bank-conflict free.
#include "common.h"
#include "mmgt_hip.h"

namespace {

constexpr int BM = 128, BN = 128;
constexpr int ROWB = 128;          // bytes of K per tile row
constexpr int LSTR = ROWB + 16;    // LDS row stride in bytes
constexpr int TILE_BYTES = BM * LSTR;

struct ADesc {
  const char* src0;
  const char* src1;
  long ld0, ld1;      // dense: row stride (elements); conv: channels per pixel of each source
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
  int act;                 // 0 none, 1 GEGLU (packed weights, out has N/2 columns), 2 SiLU
};

template <typename T, int MODE>  // MODE 0 dense, 1 conv3x3
__global__ __launch_bounds__(256) void gemm_kernel(ADesc ad, const char* __restrict__ W, long bsw, Epi ep, int M, int N,
                                                   int K, int tiles_m, int tiles_n) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int ESZ = sizeof(T);
  constexpr int BK = ROWB / ESZ;        // elements per chunk
  constexpr int KS = BK / 16;           // MFMA K-steps per chunk
  auto lA = [&](int buf) -> char* { return smem + buf * 2 * TILE_BYTES; };
  auto lB = [&](int buf) -> char* { return smem + buf * 2 * TILE_BYTES + TILE_BYTES; };

  // XCD-aware tile order: hardware deals consecutive workgroup ids round-robin over the 8 XCDs; give each XCD a
  // contiguous run of logical tiles (n fastest) so the tiles sharing an A row panel hit one L2.
  const int nwg = tiles_m * tiles_n;
  int bid = blockIdx.x;
  {
    const int q = nwg >> 3, r = nwg & 7, x = bid & 7;
    bid = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (bid >> 3);
  }
  const int tm = bid / tiles_n, tn = bid % tiles_n;
  const int bz = blockIdx.z;

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wm = wid >> 1, wn = wid & 1;
  const int lr = lane & 31, lh = lane >> 5;

  // ---- per-thread staging assignment: 4 rows (tid/8 + 32 i), one 16-byte column (tid % 8) ----
  const int vcol = tid & 7;
  const int srow = tid >> 3;
  const T* a0 = reinterpret_cast<const T*>(ad.src0) + (long)bz * ad.bs0;
  const T* a1 = ad.src1 ? reinterpret_cast<const T*>(ad.src1) + (long)bz * ad.bs1 : nullptr;
  const T* wbase = reinterpret_cast<const T*>(W) + (long)bz * bsw;

  long arow_off[4];   // dense: element offset of the row; conv: unused
  int cn[4], coy[4], cox[4];
  bool arow_ok[4];
  const T* wrow[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    int m = tm * BM + srow + 32 * i;
    arow_ok[i] = m < M;
    if (m >= M) m = M - 1;
    if (MODE == 0) {
      arow_off[i] = (long)m * ad.ld0;
    } else {
      const int hw = ad.OH * ad.OW;
      cn[i] = m / hw;
      const int rem = m - cn[i] * hw;
      coy[i] = rem / ad.OW;
      cox[i] = rem - coy[i] * ad.OW;
    }
    int n = tn * BN + srow + 32 * i;
    if (n >= N) n = N - 1;
    wrow[i] = wbase + (long)n * K;
  }

  u32x4 ra[4], rb[4];
  auto load_chunk = [&](int kc) {  // kc: element offset into K
    if (MODE == 0) {
#pragma unroll
      for (int i = 0; i < 4; ++i)
        ra[i] = *reinterpret_cast<const u32x4*>(reinterpret_cast<const char*>(a0 + arow_off[i] + kc) + vcol * 16);
    } else {
      const int cin = ad.C0 + ad.C1;
      const int tap = kc / cin;
      const int c = kc - tap * cin;
      const int ky = tap / 3, kx = tap - ky * 3;
      const int vh = ad.up ? ad.IH * 2 : ad.IH, vw = ad.up ? ad.IW * 2 : ad.IW;
      const bool second = c >= ad.C0;
      const T* base = second ? a1 : a0;
      const long cpp = second ? ad.C1 : ad.C0;
      const int cc = second ? c - ad.C0 : c;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int iy = coy[i] * ad.stride + ky - 1, ix = cox[i] * ad.stride + kx - 1;
        const bool ok = iy >= 0 && iy < vh && ix >= 0 && ix < vw;
        const int sy = ad.up ? iy >> 1 : iy, sx = ad.up ? ix >> 1 : ix;
        u32x4 v = (u32x4)(0u);
        if (ok) {
          const T* p = base + (((long)cn[i] * ad.IH + sy) * ad.IW + sx) * cpp + cc;
          v = *reinterpret_cast<const u32x4*>(reinterpret_cast<const char*>(p) + vcol * 16);
        }
        ra[i] = v;
      }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
      rb[i] = *reinterpret_cast<const u32x4*>(reinterpret_cast<const char*>(wrow[i] + kc) + vcol * 16);
  };
  auto store_chunk = [&](int buf) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int off = (srow + 32 * i) * LSTR + vcol * 16;
      *reinterpret_cast<u32x4*>(lA(buf) + off) = ra[i];
      *reinterpret_cast<u32x4*>(lB(buf) + off) = rb[i];
    }
  };

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = (f32x16)(0.f);

  const int nchunks = K / BK;
  load_chunk(0);
  store_chunk(0);
  __syncthreads();
  for (int ch = 0; ch < nchunks; ++ch) {
    const int cur = ch & 1;
    if (ch + 1 < nchunks) load_chunk((ch + 1) * BK);
    const char* pa = lA(cur) + (wm * 64 + lr) * LSTR + lh * 8 * ESZ;
    const char* pb = lB(cur) + (wn * 64 + lr) * LSTR + lh * 8 * ESZ;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      Frag<T> fa[2], fb[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        frag_load(fa[i], reinterpret_cast<const T*>(pa + i * 32 * LSTR + ks * 16 * ESZ));
        frag_load(fb[i], reinterpret_cast<const T*>(pb + i * 32 * LSTR + ks * 16 * ESZ));
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) mma32(acc[i][j], fa[i], fb[j]);
    }
    if (ch + 1 < nchunks) store_chunk(cur ^ 1);
    __syncthreads();
  }

  // ---- epilogue ----
  T* out = reinterpret_cast<T*>(ep.out) + (long)bz * ep.bso;
  const T* res = ep.residual ? reinterpret_cast<const T*>(ep.residual) + (long)bz * ep.bsr : nullptr;
  const int row0 = tm * BM + wm * 64, col0 = tn * BN + wn * 64;
  if (ep.act == 1) {
    // GEGLU: MFMA column tile 0 = h, tile 1 = gate of the same 32 output channels (weights packed by the host).
    const int n = col0 + lr;               // packed column of h
    const int ocol = (col0 >> 1) + lr;     // output column
    const bool cok = (col0 + 32 + lr) < N;
    const float bh = (ep.bias && cok) ? ep.bias[n] : 0.f;
    const float bg = (ep.bias && cok) ? ep.bias[n + 32] : 0.f;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = row0 + i * 32 + acc_row(r, lane);
        if (m < M && cok) {
          float v = (acc[i][0][r] + bh) * gelu_erf_f(acc[i][1][r] + bg);
          if (res) v += Elem<T>::ld(res + (long)m * ep.ldr + ocol);
          Elem<T>::st(out + (long)m * ep.ldo + ocol, v);
        }
      }
    return;
  }
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int n = col0 + j * 32 + lr;
    const bool cok = n < N;
    const float b = (ep.bias && cok) ? ep.bias[n] : 0.f;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = row0 + i * 32 + acc_row(r, lane);
        if (m < M && cok) {
          float v = acc[i][j][r] + b;
          if (ep.bias2) v += ep.bias2[(long)(m / ep.bias2_rows) * N + n];
          if (ep.act == 2) v = silu_f(v);
          if (ep.row_scale) v *= ep.row_scale[m];
          v *= ep.alpha;
          if (res) v += Elem<T>::ld(res + (long)m * ep.ldr + n);
          Elem<T>::st(out + (long)m * ep.ldo + n, v);
        }
      }
  }
}

template <typename T, int MODE>
int launch(const ADesc& ad, const void* W, long bsw, const Epi& ep, int M, int N, int K, int batch, hipStream_t s) {
  const int tiles_m = (M + BM - 1) / BM, tiles_n = (N + BN - 1) / BN;
  dim3 grid(tiles_m * tiles_n, 1, batch);
  const size_t lds = 4 * TILE_BYTES;
  hipLaunchKernelGGL((gemm_kernel<T, MODE>), grid, dim3(256), lds, s, ad, reinterpret_cast<const char*>(W), bsw, ep, M, N,
                     K, tiles_m, tiles_n);
  MMGT_LAUNCH_CHECK();
  return 0;
}

int check_common(int dtype, int M, int N, int K, int act) {
  MMGT_CHECK(dtype == MMGT_F32 || dtype == MMGT_BF16, "gemm: bad dtype %d", dtype);
  MMGT_CHECK(M > 0 && N > 0 && K > 0, "gemm: empty problem M=%d N=%d K=%d", M, N, K);
  MMGT_CHECK(K % 64 == 0, "gemm: K=%d must be a multiple of 64 (pad channels on the host)", K);
  MMGT_CHECK(act >= 0 && act <= 2, "gemm: bad act %d", act);
  MMGT_CHECK(act != 1 || N % 64 == 0, "gemm: GEGLU needs N %% 64 == 0 (N=%d)", N);
  return 0;
}

}  // namespace

extern "C" int mmgt_gemm(const void* A, long lda, const void* W, const float* bias, const float* bias2, int bias2_rows,
                         const float* row_scale, float alpha, const void* residual, long ldr, void* out, long ldo, int M,
                         int N, int K, int act, int batch, long bsA, long bsW, long bsR, long bsO, int dtype,
                         void* stream) {
  if (check_common(dtype, M, N, K, act)) return 1;
  MMGT_CHECK(A && W && out, "gemm: null pointer");
  MMGT_CHECK(lda >= K && batch >= 1, "gemm: lda %ld < K %d or batch %d < 1", lda, K, batch);
  MMGT_CHECK(!bias2 || bias2_rows > 0, "gemm: bias2_rows must be positive");
  const int esz = dtype == MMGT_BF16 ? 2 : 4;
  MMGT_CHECK(((uintptr_t)A % 16) == 0 && ((uintptr_t)W % 16) == 0 && (lda * esz) % 16 == 0,
             "gemm: A/W must be 16-byte aligned with 16-byte aligned rows");
  ADesc ad{};
  ad.src0 = (const char*)A;
  ad.ld0 = lda;
  ad.bs0 = bsA;
  Epi ep{};
  ep.bias = bias; ep.bias2 = bias2; ep.bias2_rows = bias2_rows; ep.row_scale = row_scale; ep.alpha = alpha;
  ep.residual = (const char*)residual; ep.ldr = ldr; ep.out = (char*)out; ep.ldo = ldo; ep.act = act;
  ep.bsr = bsR; ep.bso = bsO;
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
  MMGT_CHECK(act == 0 || act == 2, "conv3x3: act %d unsupported", act);
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
  hipStream_t s = (hipStream_t)stream;
  return dtype == MMGT_BF16 ? launch<bf16_t, 1>(ad, Wp, 0, ep, (int)M, Cout, K, 1, s)
                            : launch<float, 1>(ad, Wp, 0, ep, (int)M, Cout, K, 1, s);
}
