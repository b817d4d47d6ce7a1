// attn64_kernel (round 2, phased; ATTN_TRACE / ATTN_PACE builds) and attn64p_kernel (round 4, software-pipelined over half tiles, with its
// ABL timing ablations), cut out of mmgt_amd/csrc/attn64.hip in round 5: attn64d_kernel is the one that ships (1742 vs 1795 / 1787 us on the bank
// launch, profiles/r4/bench_attn_r4.txt).  Record only: they compiled inside attn64.hip of commit d753722 (constants, launcher branches and the
// `make trace` target live there); tools/micro/trace_attn64.py and abl_attn64.py drove them.

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


// ---- attn64p_kernel (round 4): the same tile, LDS image, fragments and numerics, software-pipelined over 32-key HALF tiles ----------------
// What the stamps and the gap model of round 4 showed (profiles/r4/attn64_gap_trace.txt, gapfill_r4.txt): attn64_kernel's stream is PHASED -- 12
// score MFMAs with nothing to issue beside them, then the maxima, then 64 exponentials around 16 P.V MFMAs -- and two such waves on a SIMD take
// the SUM of their stand-alone times (3336 cycles per tile pair against 2 x 1542), while a stream that deals the same multiset out one gap at a
// time runs at ~76 cycles per MFMA pair for two waves (95 phased).  The exponentials that could fill the score MFMAs' gaps depend on them, so
// the fillers have to come from the PREVIOUS half tile:
//   body(j):  S(j + 1) = K(j + 1) . Q'^T  (6 MFMAs)   ||   P(j) = exp2(S(j)),  O += V(j) . P(j)  (32 exp, 16 cvt_pk, 8 MFMAs)   ||   max S(j + 1)
// with the score registers of half tile j + 1 in the 32 registers that half tile j - 1 left (64 in all, as before: two waves per SIMD).  The lazy
// rescale is decided per 32 keys at the end of a body (o, the new scores and the -M columns of Q' are adjusted together, exactly as before).
// LDS: a ring of three 64-key tiles, tile t + 2 written at the top of iteration t after ONE barrier per tile (its buffer was last read in
// iteration t - 1; its first reader is the second body of iteration t + 1, behind that iteration's barrier); global loads one tile further ahead.

// ABL (timing ablations, mmgt_tune("attn64_abl", bits); results are garbage for ABL != 0): 1 no barrier in the loop, 2 no commit / prefetch in
// the loop (32: no prefetch only, 64: no commit only), 4 v_mul in place of v_exp, 8 no maxima, 16 no LDS fragment reads in the loop.
template <int ABL, int NWV>
__global__ __launch_bounds__(NWV * 64, 2) void attn64p_kernel(AttnParams p) {
  typedef bf16_t T;
  constexpr int NB = 3, NT = NWV * 64, NW = NWV;   // (shadow the file's 4-wave constants: 8 waves = one workgroup per CU, a tile staged once per 512 queries)
  constexpr int KVEC = (KT * NVK + NT - 1) / NT, VVEC = (HD * NVV + NT - 1) / NT;
  __shared__ __attribute__((aligned(16))) char smem[NB * TILE_BYTES];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int lr = lane & 31, lh = lane >> 5;
  int pair, qblk;
  {
    const int nqb = p.nqb, id = blockIdx.x;
    if ((p.npairs & 7) == 0) {
      const int xcd = id & 7, slot = id >> 3;
      pair = xcd + 8 * (slot / nqb);
      qblk = slot % nqb;
    } else {
      pair = id / nqb;
      qblk = id - pair * nqb;
    }
  }
  pair = p.npairs - 1 - pair;   // longest first (see attn64_kernel)
  const int b = pair / p.heads, head = pair - b * p.heads;
  const int bo = b / p.bdiv, bi = b - bo * p.bdiv;
  const int q0 = (qblk * NW + wid) * (32 * QB);
  const T* qb_ = reinterpret_cast<const T*>(p.q) + bo * p.q_bs0 + bi * p.q_bs1 + (long)head * HD;
  T* ob = reinterpret_cast<T*>(p.o) + bo * p.o_bs0 + bi * p.o_bs1 + (long)head * HD;

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

  const bool has2 = p.k2 != nullptr && p.nk2 > 0 && b >= p.seg2_first_batch;
  const int nt0 = p.nk / KT;
  const int ntiles = nt0 + (has2 ? p.nk2 / KT : 0);
  const T* kb0 = reinterpret_cast<const T*>(p.k) + bo * p.k_bs0 + bi * p.k_bs1 + (long)head * HD;
  const T* vb0 = reinterpret_cast<const T*>(p.v) + bo * p.v_bs0 + bi * p.v_bs1;
  const int b2 = b / p.k2_bdiv;
  const T* kb1 = has2 ? reinterpret_cast<const T*>(p.k2) + b2 * p.k2_bs + (long)head * HD : kb0;
  const T* vb1 = has2 ? reinterpret_cast<const T*>(p.v2) + b2 * p.v2_bs : vb0;

  for (int i = tid * 16; i < NB * TILE_BYTES; i += NT * 16) *reinterpret_cast<u32x4*>(smem + i) = (u32x4)(0u);
  __syncthreads();
  if (tid < NB * KT) {
    char* bt = smem + (tid / KT) * TILE_BYTES;
    const int r = tid % KT;
    Elem<T>::st(reinterpret_cast<T*>(bt + r * RSK) + HD, 1.f);
    Elem<T>::st(reinterpret_cast<T*>(bt + r * RSK) + HD + 1, 1.f);
    Elem<T>::st(reinterpret_cast<T*>(bt + KT * RSK + HD * RSV) + r, 1.f);
  }

  u32x4 rk[KVEC], rv[VVEC];
  const T* pk[KVEC];
  const T* pv[VVEC];
  auto prefetch = [&](int it) {     // tiles strictly in order, one call per tile (running pointers)
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
        u32x2* dst = reinterpret_cast<u32x2*>(bV + row * RSV + (vc >> 1) * 32 + (vc & 1) * 8);
        dst[0] = (u32x2){rv[i][0], rv[i][1]};
        dst[2] = (u32x2){rv[i][2], rv[i][3]};
      }
    }
  };
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
            union { bf16_t e[4]; u32x2 u; } pk4;
#pragma unroll
            for (int e = 0; e < 4; ++e) pk4.e[e] = f32_to_bf16(o[qb][dt][4 * g + e] * inv);
            *reinterpret_cast<u32x2*>(orow + d) = pk4.u;
          }
        }
    }
  };
  T* ob_twin = p.o_twin ? reinterpret_cast<T*>(p.o_twin) + bo * p.o_bs0 + bi * p.o_bs1 + (long)head * HD : nullptr;

  // the (rare) rescale of a half tile's decision: o, the NEW scores and the -M columns of Q' move together
  auto rescale = [&](f32x16 (&sn)[QB], const float (&mt)[QB], bool first) __attribute__((always_inline)) {
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) {
      float delta = first ? mt[qb] : fmaxf(mt[qb], 0.f);
      const float m_new = m_run[qb] + delta;
      const float hi = Elem<T>::cvt(m_new), lo = Elem<T>::cvt(m_new - hi);
      delta = (hi + lo) - m_run[qb];
      if (lh == 1) {
        qf[qb][KSQ - 1].set(0, -hi);
        qf[qb][KSQ - 1].set(1, -lo);
      }
      const float alpha = __builtin_amdgcn_exp2f(-delta);
      m_run[qb] += delta;
#pragma unroll
      for (int i = 0; i < DT; ++i) o[qb][i] *= alpha;
      sn[qb] -= delta;
    }
  };
  auto tile_max = [&](const f32x16 (&sn)[QB], float (&mt)[QB]) __attribute__((always_inline)) {
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) {
      float m1 = fmaxf(fmaxf(sn[qb][0], sn[qb][1]), sn[qb][2]);
#pragma unroll
      for (int r = 3; r < 15; r += 2) m1 = fmaxf(fmaxf(m1, sn[qb][r]), sn[qb][r + 1]);
      m1 = fmaxf(m1, sn[qb][15]);
      const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(m1), __float_as_uint(m1), false, false);
      mt[qb] = fmaxf(__uint_as_float(sw[0]), __uint_as_float(sw[1]));
    }
  };

  // ---- prologue: tiles 0 and 1 into the ring, tile 2 into the staging registers, the scores of half tile 0 ----
  prefetch(0);
  commit(0);
  if (ntiles > 1) { prefetch(1); commit(1); }
  if (ntiles > 2) prefetch(2);
  __syncthreads();
  f32x16 sA[QB], sB[QB];
  {
    Frag<T> kf[KSQ];
#pragma unroll
    for (int ks = 0; ks < KSQ; ++ks) frag_load(kf[ks], reinterpret_cast<const T*>(smem + lr * RSK + lh * 16 + ks * 32));
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) sA[qb] = (f32x16)(0.f);
#pragma unroll
    for (int ks = 0; ks < KSQ; ++ks)
#pragma unroll
      for (int qb = 0; qb < QB; ++qb) mma32(sA[qb], kf[ks], qf[qb][ks]);
    float mt[QB];
    tile_max(sA, mt);
    rescale(sA, mt, true);
  }

  // One body: sp = scores of the half tile to exponentiate (V^T columns at vcol), sn = scores to compute from the K rows at krow.
  // decide: the half tile behind sn exists.  twin: the state after sp is the twin output (written in front of sn's decision).
  // Issue order.  hipcc's sched_group_barrier pipeline gave up on this block (all exponentials first, then runs of MFMAs), and a plain
  // sched_barrier(0) fences only what has side effects: pure instructions (v_exp, v_cvt_pk, MFMA) float to their first user when the
  // block is linearised.  So every chunk {1 MFMA + its fillers} pins its inputs and outputs through EMPTY volatile asm statements (no
  // instruction, an ordering edge) between two fences: the order below IS the issue order.
#define FENCE() __builtin_amdgcn_sched_barrier(0)
#define PIN(x) asm volatile("" : "+v"(x))
  auto body = [&](f32x16 (&sp)[QB], f32x16 (&sn)[QB], const char* krow, const char* vcol, bool decide, bool twin) __attribute__((always_inline)) {
    Frag<T> kf[KSQ];
    union VF { u32x4 u; Frag<T> f; } vf[2][DT];
    if (ABL & 16) {
#pragma unroll
      for (int ks = 0; ks < KSQ; ++ks) kf[ks] = qf[0][ks];
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) vf[s2][dt].f = qf[1][dt];
    } else {
#pragma unroll
    for (int ks = 0; ks < KSQ; ++ks) frag_load(kf[ks], reinterpret_cast<const T*>(krow + ks * 32));
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) vf[0][dt].u = *reinterpret_cast<const u32x4*>(vcol + dt * 32 * RSV);
    }
    float ex[2][16];                 // [query block][register of the half tile]: the exponentials
    Frag<T> pf[2][QB];               // [s2][query block]
    // exponentials e0 .. e1 - 1 of the 16 of group s2 (qb = e / 8), packs as they complete: inputs pinned in front, results behind (one
    // pin per exponential would put the one-wait-state TRANS hazard's s_nop behind every v_exp)
    auto exps = [&](int e0, int e1, int s2) __attribute__((always_inline)) {
#pragma unroll
      for (int e = e0; e < e1; ++e) PIN(sp[e >> 3][8 * s2 + (e & 7)]);
#pragma unroll
      for (int e = e0; e < e1; ++e)
        ex[e >> 3][8 * s2 + (e & 7)] = (ABL & 4) ? sp[e >> 3][8 * s2 + (e & 7)] * 1.5f : __builtin_amdgcn_exp2f(sp[e >> 3][8 * s2 + (e & 7)]);
#pragma unroll
      for (int e = e0; e < e1; ++e) {
        const int eq = e >> 3;
        if ((e & 7) == 7) {
          float p8[8];
#pragma unroll
          for (int j = 0; j < 8; ++j) p8[j] = ex[eq][8 * s2 + j];
          frag_set8(pf[s2][eq], p8);
          PIN(pf[s2][eq].v);
        } else if (e == e1 - 1 || (e & 7) < 6 || true) {
          PIN(ex[eq][8 * s2 + (e & 7)]);
        }
      }
    };
    FENCE();
    // chunks 0..5: the six score MFMAs of the NEXT half tile, the first 16 exponentials of this one (3 3 2 3 3 2) dealt out behind them
#pragma unroll
    for (int c = 0; c < 6; ++c) {
      const int ks = c >> 1, qb = c & 1;
      if (qb == 0) PIN(kf[ks].v);
      if (ks == 0) sn[qb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[0].v, qf[qb][0].v, (f32x16)(0.f), 0, 0, 0);
      else mma32(sn[qb], kf[ks], qf[qb][ks]);
      PIN(sn[qb]);
      exps((16 * c + 3) / 6, (16 * (c + 1) + 3) / 6, 0);
      FENCE();
    }
    if (!(ABL & 16)) {
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) vf[1][dt].u = *reinterpret_cast<const u32x4*>(vcol + dt * 32 * RSV + 32);
    }
    // chunks 6..9: P.V of the first 16 keys, the second 16 exponentials (4 each) behind them
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const int dt = c >> 1, qb = c & 1;
      mma32(o[qb][dt], vf[0][dt].f, pf[0][qb]);
      PIN(o[qb][dt]);
      exps(4 * c, 4 * c + 4, 1);
      FENCE();
    }
    // chunks 10..13: P.V of the second 16 keys beside the maxima of the new scores (one query block's chain per gap pair)
    float mt[QB];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const int dt = c >> 1, qb = c & 1;
      mma32(o[qb][dt], vf[1][dt].f, pf[1][qb]);
      PIN(o[qb][dt]);
      if (ABL & 8) {
        if (c < 2) mt[c] = sn[c][0];
      } else if (c < 2) {
        PIN(sn[c]);
        float m1 = fmaxf(fmaxf(sn[c][0], sn[c][1]), sn[c][2]);
#pragma unroll
        for (int r = 3; r < 15; r += 2) m1 = fmaxf(fmaxf(m1, sn[c][r]), sn[c][r + 1]);
        m1 = fmaxf(m1, sn[c][15]);
        const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(m1), __float_as_uint(m1), false, false);
        mt[c] = fmaxf(__uint_as_float(sw[0]), __uint_as_float(sw[1]));
        PIN(mt[c]);
      }
      FENCE();
    }
    if (twin) store_out(ob_twin);
    if (decide && __any(fmaxf(mt[0], mt[1]) > RESCALE_LAG)) rescale(sn, mt, false);
  };
#undef PIN
#undef FENCE

  int cur = 0;                                         // ring slot of tile `it`
  for (int it = 0; it < ntiles; ++it) {
    const int nxt = cur == NB - 1 ? 0 : cur + 1;
    if (it > 0 && !(ABL & 1)) __syncthreads();         // every wave has finished iteration it - 1: slot (it + 2) % 3 = (it - 1) % 3 is free
    if (!(ABL & 2)) {
      if (it + 2 < ntiles && !(ABL & 64)) commit(nxt == NB - 1 ? 0 : nxt + 1);
      if (it + 3 < ntiles && !(ABL & 32)) prefetch(it + 3);
    }
    const char* tK = smem + cur * TILE_BYTES + lr * RSK + lh * 16;
    const char* tV = smem + cur * TILE_BYTES + KT * RSK + lr * RSV + lh * 16;
    body(sA, sB, tK + 32 * RSK, tV, true, false);                                                        // S(it, 1) || P.V(it, 0)
    body(sB, sA, smem + nxt * TILE_BYTES + lr * RSK + lh * 16, tV + 64, it + 1 < ntiles, ob_twin && it == nt0 - 1);   // S(it + 1, 0) || P.V(it, 1)
    cur = nxt;
  }
  store_out(ob);
}


