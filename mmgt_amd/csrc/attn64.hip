// Spatial flash attention, head_dim 40, bf16, 64 queries per wave (gfx950).
//
// Same algorithm, LDS image and numerics as attn_kernel<bf16, 40, 4, true, 64, false> of attention.hip (transposed scores, -M
// folded into the score MFMA's zero padding, softmax denominator from a ones row of V^T, lazy rescale, key-permuted V^T rows,
// bank as a second key / value segment), but every wave carries TWO 32-query blocks through each 64-key tile: the K and V^T
// fragments of the tile are read from LDS once per wave and feed both blocks (14 ds_read_b128 per 64 x 64 wave tile instead of
// 28), the tile's staging, barriers and loop control are paid once per 256 queries instead of once per 128, and the two blocks'
// chains are independent (one block's exponentials can issue beside the other's MFMAs).  The price is two waves per SIMD
// instead of three (o and s for two blocks: 128 registers).  Dispatched for whole-tile key sets with nq % 256 == 0.
#include "common.h"
#include "attn_common.h"
#include "mmgt_hip.h"

namespace {

constexpr int HD = 40, KT = 64, NSUB = 2, HDK = 48, KSQ = 3, DT = 2, HDV = 64, QB = 2, NW = 4, NT = 256;
constexpr int RSK = HDK * 2 + 16, RSV = KT * 2 + 16, TILE_BYTES = KT * RSK + HDV * RSV;
constexpr int NVK = HD / 8, NVV = KT / 8;                       // 16-byte vectors per K row / per V^T row of a tile
constexpr int KVEC = (KT * NVK + NT - 1) / NT, VVEC = (HD * NVV + NT - 1) / NT;
constexpr float RESCALE_LAG = 8.f;                              // see attention.hip

// A wave that presents an MFMA to a busy matrix pipe blocks the SIMD's vector issue port, its partner's VALU included (tools/micro/
// coexec.hip); ATTN_PACE pads behind MFMAs with s_nop so that the partner's softmax can issue meanwhile (A/B: tools/ab_attn_pace.sh).
#ifndef ATTN_PACE
#define ATTN_PACE 0
#endif
__device__ __forceinline__ void pace_qk() {
  if (ATTN_PACE == 1 || ATTN_PACE == 2) { __builtin_amdgcn_sched_barrier(0); asm volatile("s_nop 7\n\ts_nop 1"); __builtin_amdgcn_sched_barrier(0); }
  if (ATTN_PACE == 3 || ATTN_PACE == 4) { __builtin_amdgcn_sched_barrier(0); asm volatile("s_nop 7"); __builtin_amdgcn_sched_barrier(0); }
}
__device__ __forceinline__ void pace_pv() {
  if (ATTN_PACE == 2) { __builtin_amdgcn_sched_barrier(0); asm volatile("s_nop 7\n\ts_nop 1"); __builtin_amdgcn_sched_barrier(0); }
  if (ATTN_PACE == 4) { __builtin_amdgcn_sched_barrier(0); asm volatile("s_nop 7"); __builtin_amdgcn_sched_barrier(0); }
}

// ATTN_TRACE (diagnostic build only: `make -C mmgt_amd/csrc trace` -> libmmgt_hip_trace.so, tools/trace_attn64.py): every wave sums the
// shader-clock time of the six segments of a tile iteration into scalar registers and stores the sums once at the end.  The stamps fence
// the scheduler (no overlap across segment borders), so the build's SHARES are what to read, not its run time.
#ifndef ATTN_TRACE
#define ATTN_TRACE 0
#endif
#if ATTN_TRACE
#define SEG(k) do { __builtin_amdgcn_sched_barrier(0); unsigned long long t_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); \
                    __builtin_amdgcn_sched_barrier(0); seg[k] += t_ - t_prev; t_prev = t_; } while (0)
#else
#define SEG(k) do { } while (0)
#endif

#if ATTN_TRACE
__global__ __launch_bounds__(NT, 2) void attn64_kernel(AttnParams p, unsigned long long* trace) {
#else
__global__ __launch_bounds__(NT, 2) void attn64_kernel(AttnParams p) {
#endif
  typedef bf16_t T;
  // (Double-buffered tiles -- tile t + 1 written at the end of tile t's work, one barrier per tile -- measured 1 % slower;
  // 128-key staged tiles worked as two 64-key blocks, half the barriers per key: +0.8 %.)
  constexpr int NBUF = 1;
  __shared__ __attribute__((aligned(16))) char smem[NBUF * TILE_BYTES];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int lr = lane & 31, lh = lane >> 5;
  int pair, qblk;
  {
    const int nqb = p.nqb, id = blockIdx.x;
    if ((p.npairs & 7) == 0) {   // all query blocks of a (batch, head) pair on one XCD (as attention.hip)
      const int xcd = id & 7, slot = id >> 3;
      pair = xcd + 8 * (slot / nqb);
      qblk = slot % nqb;
    } else {
      pair = id / nqb;
      qblk = id - pair * nqb;
    }
  }
  // longest first: the batches that also attend to the bank (b >= seg2_first_batch, twice the keys) are the LAST pairs, so the
  // grid is walked backwards and the tail of the launch is made of short workgroups
  pair = p.npairs - 1 - pair;
  const int b = pair / p.heads, head = pair - b * p.heads;
  const int bo = b / p.bdiv, bi = b - bo * p.bdiv;
  const int q0 = (qblk * NW + wid) * (32 * QB);
  const T* qb_ = reinterpret_cast<const T*>(p.q) + bo * p.q_bs0 + bi * p.q_bs1 + (long)head * HD;
  T* ob = reinterpret_cast<T*>(p.o) + bo * p.o_bs0 + bi * p.o_bs1 + (long)head * HD;

  // Q^T fragments: lane (q = lr, half lh) holds d = 16 ks + 8 lh + j, pre-multiplied by scale * log2(e)
  Frag<T> qf[QB][KSQ];
#pragma unroll
  for (int qb = 0; qb < QB; ++qb) {
    const T* qrow = qb_ + (long)(q0 + 32 * qb + lr) * p.q_ts;
#pragma unroll
    for (int ks = 0; ks < KSQ; ++ks) {
      const int d = 16 * ks + 8 * lh;
      if (d < HD) {
        frag_load(qf[qb][ks], qrow + d);
        float q8[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) q8[j] = frag_get(qf[qb][ks], j) * p.scale_log2e;
        frag_set8(qf[qb][ks], q8);
      } else {
        qf[qb][ks].zero();
      }
    }
  }
  f32x16 o[QB][DT];
#pragma unroll
  for (int qb = 0; qb < QB; ++qb)
#pragma unroll
    for (int i = 0; i < DT; ++i) o[qb][i] = (f32x16)(0.f);
  float m_run[QB] = {0.f, 0.f};

  // tile schedule: segment 0 = own keys, segment 1 = bank keys (conditional CFG half only)
  const bool has2 = p.k2 != nullptr && p.nk2 > 0 && b >= p.seg2_first_batch;
  const int nt0 = p.nk / KT;
  const int ntiles = nt0 + (has2 ? p.nk2 / KT : 0);
  const T* kb0 = reinterpret_cast<const T*>(p.k) + bo * p.k_bs0 + bi * p.k_bs1 + (long)head * HD;
  const T* vb0 = reinterpret_cast<const T*>(p.v) + bo * p.v_bs0 + bi * p.v_bs1;
  const int b2 = b / p.k2_bdiv;
  const T* kb1 = has2 ? reinterpret_cast<const T*>(p.k2) + b2 * p.k2_bs + (long)head * HD : kb0;
  const T* vb1 = has2 ? reinterpret_cast<const T*>(p.v2) + b2 * p.v2_bs : vb0;

  // LDS image: K rows [key][48 + pad], columns 40, 41 = 1 (against -M in Q'); V^T rows [d][64 keys permuted + pad], row 40 = 1
  for (int i = tid * 16; i < NBUF * TILE_BYTES; i += NT * 16) *reinterpret_cast<u32x4*>(smem + i) = (u32x4)(0u);
  __syncthreads();
  if (tid < NBUF * KT) {
    char* bt = smem + (tid / KT) * TILE_BYTES;
    const int r = tid % KT;
    Elem<T>::st(reinterpret_cast<T*>(bt + r * RSK) + HD, 1.f);
    Elem<T>::st(reinterpret_cast<T*>(bt + r * RSK) + HD + 1, 1.f);
    Elem<T>::st(reinterpret_cast<T*>(bt + KT * RSK + HD * RSV) + r, 1.f);
  }

  // issue-early / write-late staging with running per-thread pointers (full tiles only)
  u32x4 rk[KVEC], rv[VVEC];
  const T* pk[KVEC];
  const T* pv[VVEC];
  auto prefetch = [&](int it) {
    const bool s1 = it >= nt0;
    const int kt = (s1 ? it - nt0 : it) * KT;
    const long kts = s1 ? p.k2_ts : p.k_ts, vts = s1 ? p.v2_ts : p.v_ts;
    if (kt == 0) {
      const T* kb = s1 ? kb1 : kb0;
      const T* vb = s1 ? vb1 : vb0;
#pragma unroll
      for (int i = 0; i < KVEC; ++i) {
        const int idx = tid + i * NT, row = idx / NVK, vc = idx - row * NVK;
        pk[i] = kb + (long)row * kts + vc * 8;
      }
#pragma unroll
      for (int i = 0; i < VVEC; ++i) {
        const int idx = tid + i * NT, row = idx / NVV, vc = idx - row * NVV;
        pv[i] = vb + ((long)head * HD + row) * vts + vc * 8;
      }
    }
    const long kstep = (long)KT * kts;
#pragma unroll
    for (int i = 0; i < KVEC; ++i) {
      if ((i + 1) * NT <= KT * NVK || tid + i * NT < KT * NVK) rk[i] = *reinterpret_cast<const u32x4*>(pk[i]);
      pk[i] += kstep;
    }
#pragma unroll
    for (int i = 0; i < VVEC; ++i) {
      if ((i + 1) * NT <= HD * NVV || tid + i * NT < HD * NVV) rv[i] = *reinterpret_cast<const u32x4*>(pv[i]);
      pv[i] += KT;
    }
  };
  auto commit = [&](int buf) {
    char* bK = smem + buf * TILE_BYTES;
    char* bV = bK + KT * RSK;
#pragma unroll
    for (int i = 0; i < KVEC; ++i) {
      const int idx = tid + i * NT;
      if ((i + 1) * NT <= KT * NVK || idx < KT * NVK) {
        const int row = idx / NVK, vc = idx - row * NVK;
        *reinterpret_cast<u32x4*>(bK + row * RSK + vc * 16) = rk[i];
      }
    }
#pragma unroll
    for (int i = 0; i < VVEC; ++i) {
      const int idx = tid + i * NT;
      if ((i + 1) * NT <= HD * NVV || idx < HD * NVV) {
        const int row = idx / NVV, vc = idx - row * NVV;
        // vector vc = keys 8 vc .. 8 vc + 7: its halves go to 8-byte slots (vc & 1) and 2 + (vc & 1) of key group vc >> 1
        u32x2* dst = reinterpret_cast<u32x2*>(bV + row * RSV + (vc >> 1) * 32 + (vc & 1) * 8);
        dst[0] = (u32x2){rv[i][0], rv[i][1]};
        dst[2] = (u32x2){rv[i][2], rv[i][3]};
      }
    }
  };

  // ---- normalise and store: lane (q, half) owns d = 32 dt + 8 g + 4 half + (0..3); row 40 of O^T is the denominator ----
  auto store_out = [&](T* base) __attribute__((always_inline)) {
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) {
      constexpr int R = HD % 32, REG = (R & 3) + 4 * (R >> 3), LHS = (R >> 2) & 1;
      const float mine = o[qb][HD / 32][REG], other = __shfl_xor(mine, 32);
      const float inv = 1.f / (lh == LHS ? mine : other);
      T* orow = base + (long)(q0 + 32 * qb + lr) * p.o_ts;
#pragma unroll
      for (int dt = 0; dt < DT; ++dt)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int d = dt * 32 + 8 * g + 4 * lh;
          if (d < HD) {
            union { bf16_t e[4]; u32x2 u; } pk;
#pragma unroll
            for (int e = 0; e < 4; ++e) pk.e[e] = f32_to_bf16(o[qb][dt][4 * g + e] * inv);
            *reinterpret_cast<u32x2*>(orow + d) = pk.u;
          }
        }
    }
  };
  // mmgt_attention_twin: the state after the last tile of segment 0 IS the attention over the own keys alone -- what the batch entry's
  // twin without a second segment (the unconditional CFG row: same q, k, v) would compute; it is written there and the loop goes on
  T* ob_twin = p.o_twin ? reinterpret_cast<T*>(p.o_twin) + bo * p.o_bs0 + bi * p.o_bs1 + (long)head * HD : nullptr;

  const char* lK = smem;
  const char* lV = smem + KT * RSK;
  prefetch(0);
#if ATTN_TRACE
  unsigned long long seg[7] = {0, 0, 0, 0, 0, 0, 0}, t_prev, t_begin, r_begin;
  __builtin_amdgcn_sched_barrier(0);
  asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_begin), "=s"(r_begin) :: "memory");
  __builtin_amdgcn_sched_barrier(0);
  t_prev = t_begin;
#endif
  for (int it = 0; it < ntiles; ++it) {
    __syncthreads();   // every wave has finished reading the previous tile (and the padding constants are in place)
    SEG(0);            // [0] wait for the workgroup's slowest wave of the previous tile
    commit(0);
    __syncthreads();
    SEG(1);            // [1] vmcnt wait of the prefetched tile + its LDS writes + barrier
    if (it + 1 < ntiles) prefetch(it + 1);
    SEG(2);            // [2] issue of the next tile's global loads

    // ---- S^T - M = K . Q'^T for both query blocks: the tile's K fragments are read once ----
    f32x16 s[QB][NSUB];
    {
      Frag<T> kf[NSUB][KSQ];
#pragma unroll
      for (int sub = 0; sub < NSUB; ++sub)
#pragma unroll
        for (int ks = 0; ks < KSQ; ++ks)
          frag_load(kf[sub][ks], reinterpret_cast<const T*>(lK + (sub * 32 + lr) * RSK + lh * 16 + ks * 32));
#pragma unroll
      for (int qb = 0; qb < QB; ++qb)
#pragma unroll
        for (int sub = 0; sub < NSUB; ++sub) s[qb][sub] = (f32x16)(0.f);
#pragma unroll
      for (int ks = 0; ks < KSQ; ++ks)
#pragma unroll
        for (int sub = 0; sub < NSUB; ++sub)
#pragma unroll
          for (int qb = 0; qb < QB; ++qb) { mma32(s[qb][sub], kf[sub][ks], qf[qb][ks]); pace_qk(); }
    }
    SEG(3);            // [3] 6 K fragment reads + ISSUE of the 12 score MFMAs (their completion is waited for in [4])
    // ---- tile maxima of both query blocks, one (rare) rescale branch for the wave ----
    float mt[QB];
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) {
      float m1 = fmaxf(s[qb][0][0], s[qb][1][0]), m2 = fmaxf(s[qb][0][1], s[qb][1][1]);
#pragma unroll
      for (int r = 2; r < 16; r += 2) {
        m1 = fmaxf(fmaxf(m1, s[qb][0][r]), s[qb][1][r]);
        m2 = fmaxf(fmaxf(m2, s[qb][0][r + 1]), s[qb][1][r + 1]);
      }
      m1 = fmaxf(m1, m2);
      const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(m1), __float_as_uint(m1), false, false);
      mt[qb] = fmaxf(__uint_as_float(sw[0]), __uint_as_float(sw[1]));
    }
    if (it == 0 || __any(fmaxf(mt[0], mt[1]) > RESCALE_LAG)) {
#pragma unroll
      for (int qb = 0; qb < QB; ++qb) {
        float delta = it == 0 ? mt[qb] : fmaxf(mt[qb], 0.f);
        const float m_new = m_run[qb] + delta;
        const float hi = Elem<T>::cvt(m_new), lo = Elem<T>::cvt(m_new - hi);
        delta = (hi + lo) - m_run[qb];
        if (lh == 1) {   // lanes holding d = 40 .. 47 of the last K-step
          qf[qb][KSQ - 1].set(0, -hi);
          qf[qb][KSQ - 1].set(1, -lo);
        }
        const float alpha = __builtin_amdgcn_exp2f(-delta);
        m_run[qb] += delta;
#pragma unroll
        for (int i = 0; i < DT; ++i) o[qb][i] *= alpha;
#pragma unroll
        for (int sub = 0; sub < NSUB; ++sub) s[qb][sub] -= delta;
      }
    }
    SEG(4);            // [4] drain of the score MFMAs + 34 v_max3 + 2 permlane + decision (+ the rare rescale)
    // ---- O^T += V^T . P^T, 16 keys at a time: exponentials of the group -> P fragments of both blocks -> one read of each V^T
    // fragment feeding both.  One basic block: a group's MFMAs can run under the next group's exponentials.
#pragma unroll
    for (int sub = 0; sub < NSUB; ++sub)
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        Frag<T> pf[QB];
#pragma unroll
        for (int qb = 0; qb < QB; ++qb) {
          float p8[8];
#pragma unroll
          for (int j = 0; j < 8; ++j) p8[j] = __builtin_amdgcn_exp2f(s[qb][sub][8 * s2 + j]);
          frag_set8(pf[qb], p8);
        }
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
          union { u32x4 u; Frag<T> f; } cv;    // the lane's 8 keys are 16 contiguous bytes of the permuted row
          cv.u = *reinterpret_cast<const u32x4*>(lV + (dt * 32 + lr) * RSV + (sub * 2 + s2) * 32 + lh * 16);
#pragma unroll
          for (int qb = 0; qb < QB; ++qb) { mma32(o[qb][dt], cv.f, pf[qb]); pace_pv(); }
        }
      }
    SEG(5);            // [5] 64 v_exp + 32 cvt_pk + 8 V^T fragment reads + ISSUE of the 16 P.V MFMAs
    if (ob_twin && it == nt0 - 1) store_out(ob_twin);
  }
#if ATTN_TRACE
  {
    unsigned long long t_end, r_end;
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_end), "=s"(r_end) :: "memory");
    __builtin_amdgcn_sched_barrier(0);
    if (trace && lane == 0) {
      unsigned long long* dst = trace + ((long)blockIdx.x * NW + wid) * 16;
#pragma unroll
      for (int k = 0; k < 6; ++k) dst[k] = seg[k];
      dst[6] = t_end - t_begin;      // shader cycles of the whole loop
      dst[7] = r_end - r_begin;      // the same span on the 100-MHz clock
      dst[8] = (unsigned long long)ntiles;
      dst[9] = (unsigned long long)__builtin_amdgcn_s_getreg(((4 - 1) << 11) | (0 << 6) | 20);   // HW_REG_XCC_ID[3:0]
    }
  }
#endif

  store_out(ob);
}

}  // namespace

// attention.hip's dispatcher: bf16, head_dim 40, V transposed, nq % 256 == 0, nk % 64 == 0, nk2 % 64 == 0
int g_attn64_lds_pad = 0;   // mmgt_tune("attn64_pad", bytes): extra dynamic LDS per workgroup (experiment: 96 KiB forces one workgroup per CU)
void mmgt_attn64_set_pad(int v) { g_attn64_lds_pad = v; }

#if ATTN_TRACE
namespace { unsigned long long* g_attn64_trace = nullptr; }
extern "C" void mmgt_attn64_set_trace(void* p) { g_attn64_trace = reinterpret_cast<unsigned long long*>(p); }
#endif

int mmgt_attn64_launch(const void* params, int batch, int heads, void* stream) {
  AttnParams p = *reinterpret_cast<const AttnParams*>(params);
  p.heads = heads;
  p.npairs = batch * heads;
  p.nqb = p.nq / (32 * QB * NW);
  if (g_attn64_lds_pad > 65536 - 16384) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(attn64_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, g_attn64_lds_pad);
#if ATTN_TRACE
  hipLaunchKernelGGL(attn64_kernel, dim3((unsigned)((long)p.nqb * batch * heads)), dim3(NT), (size_t)g_attn64_lds_pad, (hipStream_t)stream, p, g_attn64_trace);
#else
  hipLaunchKernelGGL(attn64_kernel, dim3((unsigned)((long)p.nqb * batch * heads)), dim3(NT), (size_t)g_attn64_lds_pad, (hipStream_t)stream, p);
#endif
  MMGT_LAUNCH_CHECK();
  return 0;
}
