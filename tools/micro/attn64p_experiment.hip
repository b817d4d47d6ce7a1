// EXPERIMENT, NOT BUILT INTO libmmgt_hip.so (round 3): measured 3339 us against 1795 us for attn64.hip at the level-0 bank shape
// (48 x 8 heads, 4096 queries, 4096 + 4096 keys; bitwise identical results).  Ablations: without the exponentials 2793 us, without the
// PV MFMAs 1830 us.  Why it loses: one wave issues one instruction per ~5.5 cycles, a slot needs ~165 of them (the score tiles end up in
// the accumulation half of the register file, so every score is read back with v_accvgpr_read -- twice: maxima, exponentials), and
// the MFMAs of a group are presented back to back, so the wave stalls at the pipe instead of issuing the next group's exponentials.
// Two waves per SIMD with VGPR-form MFMAs (attn64.hip, <= 256 registers) have twice the issue slots and no read-backs.
// To try it again: add the file to csrc/Makefile, declare mmgt_attn64p_launch in attention.hip's dispatcher (nq % 512 == 0).
// Spatial flash attention, head_dim 40, bf16: ONE wave per SIMD, 128 queries per wave, software-pipelined (gfx950).
//
// Same algorithm, LDS image and numerics as attn64.hip (transposed scores, -M folded into the score MFMA's zero padding, softmax
// denominator from a ones row of V^T, lazy rescale, key-permuted V^T rows, bank as a second key / value segment); what changes is the
// schedule.  attn64 runs a tile as three phases per wave -- 12 score MFMAs, the maxima, exponentials + 16 PV MFMAs -- and relies on
// its partner wave on the SIMD to fill the gaps; but a wave that presents an MFMA to a busy matrix pipe blocks the SIMD's vector
// issue port, its partner's VALU included (tools/micro/coexec.hip), so the two waves' phases add up: matrix pipe 52 % busy, VALU
// 56 % (profiles/r2/pmc_attn_r2.txt).  A lone wave DOES overlap its own MFMAs with the vector instructions behind them, so here a
// wave owns the SIMD (up to 512 registers) and carries four 32-query blocks through a pipeline of SLOTS, one (tile, block) each:
//   slot n = 4 it + qb:   softmax + PV of (it, qb): maxima, (rare) rescale, 4 x [8 exponentials -> P fragment, 2 PV MFMAs]
//                         and, interleaved, the 6 score MFMAs of slot n + 1 into the other of two score buffers
// i.e. every slot is 14 MFMAs (448 matrix-pipe cycles) with ~65 vector instructions and 14 fragment reads spread between them, and
// only two 32 x 64 score tiles are alive at a time (the scores are vector operands too: with four of them next to the Q fragments the
// kernel did not fit the 256 architectural VGPRs; the 128 output registers live in the accumulation half of the file).  K and V^T tiles are double-buffered: tile it + 1 is
// written at the top of iteration it (two barriers per tile, as before, but with 14 MFMAs queued across each).
#include "common.h"
#include "attn_common.h"
#include "mmgt_hip.h"

namespace {

constexpr int HD = 40, KT = 64, NSUB = 2, HDK = 48, KSQ = 3, DT = 2, HDV = 64, QB = 4, NW = 4, NT = 256;
constexpr int RSK = HDK * 2 + 16, RSV = KT * 2 + 16, KBYTES = KT * RSK, VBYTES = HDV * RSV;
constexpr int NVK = HD / 8, NVV = KT / 8;                       // 16-byte vectors per K row / per V^T row of a tile
constexpr int KVEC = (KT * NVK + NT - 1) / NT, VVEC = (HD * NVV + NT - 1) / NT;
constexpr float RESCALE_LAG = 8.f;                              // see attention.hip

template <int DBG>
__global__ __launch_bounds__(NT) __attribute__((amdgpu_waves_per_eu(1, 1))) void attn64p_kernel(AttnParams p) {
  typedef bf16_t T;
  __shared__ __attribute__((aligned(16))) char smem[2 * KBYTES + 2 * VBYTES];      // K buffers 0, 1 | V^T buffers 0, 1
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
  float m_run[QB] = {0.f, 0.f, 0.f, 0.f};

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
  for (int i = tid * 16; i < 2 * KBYTES + 2 * VBYTES; i += NT * 16) *reinterpret_cast<u32x4*>(smem + i) = (u32x4)(0u);
  __syncthreads();
  if (tid < 2 * KT) {
    const int bf = tid / KT, r = tid % KT;
    Elem<T>::st(reinterpret_cast<T*>(smem + bf * KBYTES + r * RSK) + HD, 1.f);
    Elem<T>::st(reinterpret_cast<T*>(smem + bf * KBYTES + r * RSK) + HD + 1, 1.f);
    Elem<T>::st(reinterpret_cast<T*>(smem + 2 * KBYTES + bf * VBYTES + HD * RSV) + r, 1.f);
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
    char* bK = smem + buf * KBYTES;
    char* bV = smem + 2 * KBYTES + buf * VBYTES;
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

  f32x16 s[2][NSUB];                     // score tiles of the current slot (parity qb & 1) and of the next one
  // the 6 score MFMAs of (tile in K buffer kbuf, query block QN): S^T - M = K . Q'^T; fragment (sub, ks) of the tile is one ds_read_b128
  auto score_frag = [&](int kbuf, int f, Frag<T>& kf) {     // f = 3 sub + ks
    const int sub = f / KSQ, ks = f - sub * KSQ;
    frag_load(kf, reinterpret_cast<const T*>(smem + kbuf * KBYTES + (sub * 32 + lr) * RSK + lh * 16 + ks * 32));
  };
  using std::integral_constant;

  // One slot: softmax + PV of (tile in V buffer vbuf, block QBc) interleaved with the score MFMAs of the next slot's block QNc from
  // K buffer kbuf (if qk).
  auto slot = [&](auto QBc, auto QNc, bool first, bool qk, int vbuf, int kbuf) __attribute__((always_inline)) {
    constexpr int qb = decltype(QBc)::value, qn = decltype(QNc)::value, sc = qb & 1, sn = sc ^ 1;
    const char* lV = smem + 2 * KBYTES + vbuf * VBYTES;
    Frag<T> kf[NSUB * KSQ];
    if (qk) {
#pragma unroll
      for (int f = 0; f < NSUB * KSQ; ++f) score_frag(kbuf, f, kf[f]);
    }
    // first V^T fragments (group 0) ahead of their MFMAs
    Frag<T> vf[2][DT];
    auto read_v = [&](int g, Frag<T> (&dst)[DT]) {
#pragma unroll
      for (int dt = 0; dt < DT; ++dt) {
        union { u32x4 u; Frag<T> f; } cv;    // the lane's 8 keys are 16 contiguous bytes of the permuted row
        cv.u = *reinterpret_cast<const u32x4*>(lV + (dt * 32 + lr) * RSV + g * 32 + lh * 16);
        dst[dt] = cv.f;
      }
    };
    read_v(0, vf[0]);
    // two score MFMAs up front: the maxima below issue in their shadow
    if (qk) {
      s[sn][0] = (f32x16)(0.f);
      s[sn][1] = (f32x16)(0.f);
      mma32(s[sn][0], kf[0], qf[qn][0]);
      mma32(s[sn][1], kf[KSQ], qf[qn][0]);
    }
    // ---- tile maxima of the block, (rare) rescale
    {
      float m1 = fmaxf(s[sc][0][0], s[sc][1][0]), m2 = fmaxf(s[sc][0][1], s[sc][1][1]);
#pragma unroll
      for (int r = 2; r < 16; r += 2) {
        m1 = fmaxf(fmaxf(m1, s[sc][0][r]), s[sc][1][r]);
        m2 = fmaxf(fmaxf(m2, s[sc][0][r + 1]), s[sc][1][r + 1]);
      }
      m1 = fmaxf(m1, m2);
      const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(m1), __float_as_uint(m1), false, false);
      const float mt = fmaxf(__uint_as_float(sw[0]), __uint_as_float(sw[1]));
      if (first || __any(mt > RESCALE_LAG)) {
        float delta = first ? mt : fmaxf(mt, 0.f);
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
        for (int sub = 0; sub < NSUB; ++sub) s[sc][sub] -= delta;
      }
    }
    // ---- O^T += V^T . P^T, 16 keys at a time; the remaining four score MFMAs ride on groups 0 .. 3
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int sub = g >> 1, s2 = g & 1;
      if (g + 1 < 4) read_v(g + 1, vf[(g + 1) & 1]);
      __builtin_amdgcn_sched_barrier(0);
      Frag<T> pf;
      {
        float p8[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) p8[j] = DBG == 1 ? s[sc][sub][8 * s2 + j] : __builtin_amdgcn_exp2f(s[sc][sub][8 * s2 + j]);
        frag_set8(pf, p8);
      }
      if (qk) {     // MFMA g + 2 of the six: (sub', ks) = (0,1) (1,1) (0,2) (1,2)
        const int sq = g & 1, ks = 1 + (g >> 1);
        mma32(s[sn][sq], kf[sq * KSQ + ks], qf[qn][ks]);
      }
#pragma unroll
      for (int dt = 0; dt < DT; ++dt) { if (DBG != 2) mma32(o[qb][dt], vf[g & 1][dt], pf); else o[qb][dt][0] += __uint_as_float((unsigned)pf.v[0]); }
      __builtin_amdgcn_sched_barrier(0);
    }
  };
#define A_IC(v) integral_constant<int, (v)>{}

  // ---- prologue: tile 0 staged, the scores of slot 0
  prefetch(0);
  __syncthreads();                         // (zero fill and constants in place)
  commit(0);
  if (ntiles > 1) prefetch(1);
  __syncthreads();
  {
    Frag<T> kf[NSUB * KSQ];
#pragma unroll
    for (int f = 0; f < NSUB * KSQ; ++f) score_frag(0, f, kf[f]);
#pragma unroll
    for (int sub = 0; sub < NSUB; ++sub) s[0][sub] = (f32x16)(0.f);
#pragma unroll
    for (int ks = 0; ks < KSQ; ++ks)
#pragma unroll
      for (int sub = 0; sub < NSUB; ++sub) mma32(s[0][sub], kf[sub * KSQ + ks], qf[0][ks]);
  }
  for (int it = 0; it < ntiles; ++it) {
    const int cur = it & 1, nxt = cur ^ 1;
    const bool more = it + 1 < ntiles;
    if (it > 0) __syncthreads();           // every wave is through iteration it - 1: the buffers of tile it - 1 take tile it + 1
    if (more) commit(nxt);
    if (it + 2 < ntiles) prefetch(it + 2);
    slot(A_IC(0), A_IC(1), it == 0, true, cur, cur);        // scores of (it, 1 .. 3) from K tile it
    slot(A_IC(1), A_IC(2), it == 0, true, cur, cur);
    slot(A_IC(2), A_IC(3), it == 0, true, cur, cur);
    __syncthreads();                       // tile it + 1 is visible
    slot(A_IC(3), A_IC(0), it == 0, more, cur, nxt);        // scores of (it + 1, 0) from K tile it + 1
  }
#undef A_IC

  // ---- normalise and store: lane (q, half) owns d = 32 dt + 8 g + 4 half + (0..3); row 40 of O^T is the denominator ----
#pragma unroll
  for (int qb = 0; qb < QB; ++qb) {
    constexpr int R = HD % 32, REG = (R & 3) + 4 * (R >> 3), LHS = (R >> 2) & 1;
    const float mine = o[qb][HD / 32][REG], other = __shfl_xor(mine, 32);
    const float inv = 1.f / (lh == LHS ? mine : other);
    T* orow = ob + (long)(q0 + 32 * qb + lr) * p.o_ts;
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
}

}  // namespace

// attention.hip's dispatcher: bf16, head_dim 40, V transposed, nq % 512 == 0, nk % 64 == 0, nk2 % 64 == 0
int g_attn64p_dbg = 0;
void mmgt_attn64p_set_dbg(int v) { g_attn64p_dbg = v; }
int mmgt_attn64p_launch(const void* params, int batch, int heads, void* stream) {
  AttnParams p = *reinterpret_cast<const AttnParams*>(params);
  p.heads = heads;
  p.npairs = batch * heads;
  p.nqb = p.nq / (32 * QB * NW);
  auto kern = g_attn64p_dbg == 1 ? attn64p_kernel<1> : g_attn64p_dbg == 2 ? attn64p_kernel<2> : attn64p_kernel<0>;
  hipLaunchKernelGGL(kern, dim3((unsigned)((long)p.nqb * batch * heads)), dim3(NT), 0, (hipStream_t)stream, p);
  MMGT_LAUNCH_CHECK();
  return 0;
}
