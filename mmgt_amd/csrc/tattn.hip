// Temporal self-attention of the motion modules (<= 32 frames per sequence) for the MMGT Stage-2 path (gfx950).
//
// Replaces: VersatileAttention over the frame axis, src/models/motion_module.py:351-388 (the reference rearranges
// (b f) d c -> (b d) f c and runs diffusers' attention on 2 * h * w sequences of f frames).
//
// The data never leaves the sampler's token layout ((b f), h w, 3 C) (q | k | v of one pixel and frame are 3 x C contiguous
// values): a sequence is the f rows of ONE pixel, hw * 3C elements apart.  The generic flash kernel (attention.hip) runs
// it as one 64-thread workgroup per (pixel, head), each fetching 80-byte row pieces (HD = 40) and padding 24 keys to a
// 32-key tile: 31-36 TFLOP/s and 2.7 TB/s at level 0.  The arithmetic is negligible (6 GFLOP per launch at level 0), the
// bytes are not (503 MB), so this kernel is built around the memory stream:
//  * one workgroup per pixel and group of 320 / HD heads (8, 4, 2 for HD = 40, 80, 160): its q, k and v row pieces are
//    320 contiguous values (640 B bf16) per frame, fetched as 16-byte vectors by all 256 threads, every load of the
//    workgroup issued before the first wait, into three [frame][320] LDS images (row stride 640 + 16 B: the sixteen rows
//    of a ds_read_b128 group land on sixteen different 16-byte bank slots);
//  * one wave per head (two heads per wave at HD = 40): S^T = K . Q'^T on the matrix cores with the keys on the accumulator
//    registers (a lane owns one query frame), one softmax pass (the whole key set is one tile: no online rescaling),
//    O^T = V^T . P^T with V^T gathered from the row-major V image by 16-bit LDS reads;
//  * the normalised output goes back through LDS (over the dead Q image) and leaves as whole 640-byte rows.
#include "common.h"
#include "mmgt_hip.h"

namespace {

struct TAttnParams {
  const char *q, *k, *v;
  char* o;
  long q_bs0, q_bs1, q_ts, k_bs0, k_bs1, k_ts, v_bs0, v_bs1, v_ts, o_bs0, o_bs1, o_ts;   // element strides
  int bdiv, frames, heads;
  float scale_log2e;
};

constexpr int TA_W = 320;   // values of q (and k, v) per frame handled by one workgroup

template <typename T, int HD>
__global__ __launch_bounds__(256) void tattn_kernel(TAttnParams p) {
  constexpr int ESZ = sizeof(T), VEC = 16 / ESZ;
  constexpr int HPW = TA_W / HD;                 // heads per workgroup: 8, 4, 2
  constexpr int NWV = HPW < 4 ? HPW : 4;         // waves that compute
  constexpr int RS = TA_W * ESZ + 16;            // LDS row stride (bytes)
  constexpr int KSQ = (HD + 15) / 16;            // K-steps of the score product (zero padded)
  constexpr int DT = (HD + 31) / 32;             // 32-row tiles of O^T
  constexpr int NV = TA_W / VEC;                 // 16-byte vectors per row piece
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int F = p.frames;
  char* lQ = smem;
  char* lK = smem + F * RS;
  char* lV = smem + 2 * F * RS;

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int lr = lane & 31, lh = lane >> 5;
  const int groups = p.heads / HPW;
  const int b = blockIdx.x / groups, hg = blockIdx.x - b * groups;
  const int bo = b / p.bdiv, bi = b - bo * p.bdiv;
  const long col0 = (long)hg * TA_W;
  const T* qb = reinterpret_cast<const T*>(p.q) + bo * p.q_bs0 + bi * p.q_bs1 + col0;
  const T* kb = reinterpret_cast<const T*>(p.k) + bo * p.k_bs0 + bi * p.k_bs1 + col0;
  const T* vb = reinterpret_cast<const T*>(p.v) + bo * p.v_bs0 + bi * p.v_bs1 + col0;
  T* ob = reinterpret_cast<T*>(p.o) + bo * p.o_bs0 + bi * p.o_bs1 + col0;

  // ---- stage q | k | v: F rows x 3 pieces x NV vectors, all loads of a thread in flight together
  const int nvec = 3 * F * NV;
  constexpr int MAXL = (3 * 32 * NV + 255) / 256;   // F <= 32
  u32x4 r[MAXL];
#pragma unroll
  for (int i = 0; i < MAXL; ++i) {
    const int idx = tid + i * 256;
    if (idx < nvec) {
      const int piece = idx / (F * NV), rem = idx - piece * (F * NV);
      const int f = rem / NV, vc = rem - f * NV;
      const T* src = piece == 0 ? qb + (long)f * p.q_ts : piece == 1 ? kb + (long)f * p.k_ts : vb + (long)f * p.v_ts;
      r[i] = *reinterpret_cast<const u32x4*>(src + vc * VEC);
    }
  }
#pragma unroll
  for (int i = 0; i < MAXL; ++i) {
    const int idx = tid + i * 256;
    if (idx < nvec) {
      const int piece = idx / (F * NV), rem = idx - piece * (F * NV);
      const int f = rem / NV, vc = rem - f * NV;
      *reinterpret_cast<u32x4*>(smem + (piece * F + f) * RS + vc * 16) = r[i];
    }
  }
  __syncthreads();

  // rows past the last frame are clamped to it: their scores are masked, their probabilities exactly zero
  const int rowc = lr < F ? lr : F - 1;
  f32x16 oacc[(HPW + NWV - 1) / NWV][DT];
  float linv[(HPW + NWV - 1) / NWV];
  if (wid < NWV) {
#pragma unroll
    for (int hi = 0; hi < (HPW + NWV - 1) / NWV; ++hi) {
      const int hh = wid + hi * NWV;                 // head inside the workgroup's group
      // ---- S^T[key][q] = K . Q'^T  (scale * log2(e) folded into Q', padded reduction slots zeroed in both operands)
      f32x16 s = (f32x16)(0.f);
#pragma unroll
      for (int ks = 0; ks < KSQ; ++ks) {
        const int d = 16 * ks + 8 * lh;
        Frag<T> qf, kf;
        if (d < HD) {
          frag_load(qf, reinterpret_cast<const T*>(lQ + rowc * RS) + hh * HD + d);
          frag_load(kf, reinterpret_cast<const T*>(lK + rowc * RS) + hh * HD + d);
          float q8[8];
#pragma unroll
          for (int j = 0; j < 8; ++j) q8[j] = frag_get(qf, j) * p.scale_log2e;
          frag_set8(qf, q8);
        } else {
          qf.zero();
          kf.zero();
        }
        mma32(s, kf, qf);
      }
      // ---- softmax over the keys: registers r <-> key (r & 3) + 8 (r >> 2) + 4 lh, the other half by permlane32_swap
      float mx = -1e30f;
#pragma unroll
      for (int rr = 0; rr < 16; ++rr) {
        if (acc_row(rr, lane) >= F) s[rr] = -1e30f;
        mx = fmaxf(mx, s[rr]);
      }
      {
        const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(mx), __float_as_uint(mx), false, false);
        mx = fmaxf(__uint_as_float(sw[0]), __uint_as_float(sw[1]));
      }
      float ls = 0.f;
#pragma unroll
      for (int rr = 0; rr < 16; ++rr) {
        s[rr] = __builtin_amdgcn_exp2f(s[rr] - mx);
        ls += s[rr];
      }
      {
        const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(ls), __float_as_uint(ls), false, false);
        ls = __uint_as_float(sw[0]) + __uint_as_float(sw[1]);
      }
      linv[hi] = 1.f / ls;
      // ---- O^T[d][q] += V^T[d][key] . P^T[key][q]: the accumulator registers 8 s2 .. 8 s2 + 7 are the B fragment of K-step s2,
      // element j <-> key 16 s2 + 8 (j >> 2) + 4 lh + (j & 3) (guide section 3, "an accumulator tile as the next MFMA's operand")
#pragma unroll
      for (int dt = 0; dt < DT; ++dt) oacc[hi][dt] = (f32x16)(0.f);
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        Frag<T> pf;
        {
          float p8[8];
#pragma unroll
          for (int j = 0; j < 8; ++j) p8[j] = s[8 * s2 + j];
          frag_set8(pf, p8);
        }
        const int kbase = 16 * s2 + 4 * lh;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
          const int d = dt * 32 + lr;
          const int dc = d < HD ? d : HD - 1;       // rows past the head are computed on clamped data and never stored
          union { T e[8]; Frag<T> f; } cv;
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            int key = kbase + 8 * (j >> 2) + (j & 3);
            key = key < F ? key : F - 1;
            cv.e[j] = *(reinterpret_cast<const T*>(lV + key * RS) + hh * HD + dc);
          }
          mma32(oacc[hi][dt], cv.f, pf);
        }
      }
    }
  }
  __syncthreads();   // every wave has finished reading the Q image: it becomes the output image
  if (wid < NWV) {
#pragma unroll
    for (int hi = 0; hi < (HPW + NWV - 1) / NWV; ++hi) {
      const int hh = wid + hi * NWV;
      if (lr < F) {
#pragma unroll
        for (int dt = 0; dt < DT; ++dt)
#pragma unroll
          for (int rr = 0; rr < 16; ++rr) {
            const int d = dt * 32 + acc_row(rr, lane);
            if (d < HD) Elem<T>::st(reinterpret_cast<T*>(lQ + lr * RS) + hh * HD + d, oacc[hi][dt][rr] * linv[hi]);
          }
      }
    }
  }
  __syncthreads();
  for (int idx = tid; idx < F * NV; idx += 256) {
    const int f = idx / NV, vc = idx - f * NV;
    *reinterpret_cast<u32x4*>(ob + (long)f * p.o_ts + vc * VEC) = *reinterpret_cast<const u32x4*>(lQ + f * RS + vc * 16);
  }
}

template <typename T, int HD>
int launch(const TAttnParams& p, int batch, hipStream_t s) {
  const size_t lds = (size_t)3 * p.frames * (TA_W * sizeof(T) + 16);
  auto kern = tattn_kernel<T, HD>;
  if (lds > 64 * 1024) {
    static bool opted = false;
    if (!opted) {
      if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) {
        mmgt_set_error("temporal attention: cannot reserve %zu bytes of LDS", lds);
        return 2;
      }
      opted = true;
    }
  }
  const long grid = (long)batch * (p.heads / (TA_W / HD));
  hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(256), lds, s, p);
  MMGT_LAUNCH_CHECK();
  return 0;
}

}  // namespace

// Called by mmgt_attention (attention.hip) for the temporal pattern: nq == nk <= 32, no second key segment, V not transposed,
// heads * hd a multiple of 320, rows 16-byte aligned.  Returns -1 if the problem is not of that form (caller falls through).
int mmgt_tattn_try(const void* q, long q_bs0, long q_bs1, long q_ts, const void* k, long k_bs0, long k_bs1, long k_ts,
                   const void* v, long v_bs0, long v_bs1, long v_ts, void* o, long o_bs0, long o_bs1, long o_ts, int bdiv,
                   int batch, int heads, int hd, int frames, float scale, int dtype, void* stream) {
  if (!(hd == 40 || hd == 80 || hd == 160) || frames > 32 || frames < 1 || (heads * hd) % TA_W != 0) return -1;
  if ((long)batch * (heads / (TA_W / hd)) >= (1l << 31)) return -1;
  TAttnParams p{};
  p.q = (const char*)q; p.k = (const char*)k; p.v = (const char*)v; p.o = (char*)o;
  p.q_bs0 = q_bs0; p.q_bs1 = q_bs1; p.q_ts = q_ts; p.k_bs0 = k_bs0; p.k_bs1 = k_bs1; p.k_ts = k_ts;
  p.v_bs0 = v_bs0; p.v_bs1 = v_bs1; p.v_ts = v_ts; p.o_bs0 = o_bs0; p.o_bs1 = o_bs1; p.o_ts = o_ts;
  p.bdiv = bdiv; p.frames = frames; p.heads = heads;
  p.scale_log2e = scale * 1.4426950408889634f;
  hipStream_t s = (hipStream_t)stream;
  if (dtype == MMGT_BF16) {
    switch (hd) {
      case 40: return launch<bf16_t, 40>(p, batch, s);
      case 80: return launch<bf16_t, 80>(p, batch, s);
      default: return launch<bf16_t, 160>(p, batch, s);
    }
  }
  switch (hd) {
    case 40: return launch<float, 40>(p, batch, s);
    case 80: return launch<float, 80>(p, batch, s);
    default: return launch<float, 160>(p, batch, s);
  }
}
