// Spatial flash attention, head_dim 40, bf16, 64 queries per wave, K / V^T tiles staged by LDS-DMA (gfx950).
//
// Same algorithm and numerics as attn_kernel<bf16, 40, 4, true, 64, false> of attention.hip (transposed scores, -M folded into the
// score MFMA's zero padding, softmax denominator from a ones row of V^T, lazy rescale, bank as a second key / value segment), but
// every wave carries TWO 32-query blocks through each 64-key tile: the K and V^T fragments of the tile are read from LDS once per
// wave and feed both blocks, the tile's staging, barrier and loop control are paid once per 256 queries, and the two blocks' chains
// are independent (one block's exponentials issue beside the other's MFMAs).  The price is two waves per SIMD instead of three (o and
// s for two blocks: 128 registers).  Dispatched for whole-tile key sets with nq % 256 == 0.  The phased (round 2) and software-
// pipelined register-staged (round 4) predecessors are kept as records in tools/micro/attn64_variants.hip.
#include <type_traits>

#include "common.h"
#include "gemm_common.h"
#include "attn_common.h"
#include "mmgt_hip.h"

namespace {

constexpr int HD = 40, KT = 64, NSUB = 2, HDK = 48, KSQ = 3, DT = 2, HDV = 64, QB = 2, NW = 4, NT = 256;
constexpr int RSK = HDK * 2 + 16, RSV = KT * 2 + 16, TILE_BYTES = KT * RSK + HDV * RSV;
constexpr int NVK = HD / 8, NVV = KT / 8;                       // 16-byte vectors per K row / per V^T row of a tile
constexpr int KVEC = (KT * NVK + NT - 1) / NT, VVEC = (HD * NVV + NT - 1) / NT;
constexpr float RESCALE_LAG = 8.f;                              // see attention.hip

// ---- attn64d_kernel (round 4): attn64p's pipeline with the tiles staged by LDS-DMA ------------------------------------------------------
// The timing ablations of attn64p (tools/abl_attn64.py, profiles/r4/abl_attn64_r4.txt) price the register staging of a tile at 17 % of the
// kernel (the ds_write pass alone 10 %: 10 KB per workgroup and tile through the VGPR -> LDS store path, plus the vmcnt(0) in front of it) and
// the fragment reads at 15 %.  Here a tile goes global -> LDS by `buffer_load_dwordx4 ... lds` (ten 1-KiB pieces per tile, two or three per
// wave), TWO tiles ahead of its first reader, into a ring of four slots; no staging registers, no LDS stores, a counted vmcnt.
// LDS image of a slot (the DMA writes 64 lanes x 16 B contiguously, so every layout decision sits in the per-lane SOURCE offset):
//   K    64 rows x 80 B, packed.  Row i holds key pi(i) = i with bits 2 and 3 swapped: the score MFMA's output row (r & 3) + 8 (r >> 2) + 4 half
//        then puts keys 8 half .. 8 half + 7 of a 16-key step into registers 8 s2 .. 8 s2 + 7 -- the P fragment is contiguous in the KEY
//        order of V^T, which can therefore stay unpermuted.  The -M / zero padding of the third K-step (d = 40 .. 47) is one constant 16-byte
//        vector that every upper-half lane reads (broadcast).  80-byte rows: the sixteen lanes of a ds_read_b128 group hit sixteen granules.
//   V^T  40 rows x 128 B, 16-byte chunk c of row d at position c ^ ((d >> 1) & 7) (conflict-free for rows d = 32 dt + lane); the ones row
//        (d = 40, the softmax denominator) and the zero rows above it are two more constant vectors inside the slot.
constexpr int SLOT_K = KT * HD * 2, SLOT_V = HD * KT * 2, SLOT_CONST = SLOT_K + SLOT_V, SLOT_BYTES = SLOT_CONST + 64, NSLOT = 4;

#define FENCE() __builtin_amdgcn_sched_barrier(0)
#define PIN(x) asm volatile("" : "+v"(x))
// FAST: no running maximum.  The softmax reference of a query row stays what the first 32 keys set it to: any reference gives the same quotient
// sum(p v) / sum(p) as long as no p = 2^(s - reference) overflows (fp32 and bf16 share the exponent range, so a large p loses no relative
// precision), and the per-tile maxima, the lazy-rescale test and the reference update -- 22 of the ~74 vector instructions of a 32 x 64 block in a
// kernel that is bound by vector issue -- are not computed at all.  The guard: when a row's denominator ends beyond 2^100 (or not finite),
// the WORKGROUP runs the tile loop again with the running maximum (uniform decision through LDS): results are then bitwise those of FAST = false.
template <bool FAST>
__global__ __launch_bounds__(NT, 2) void attn64d_kernel(AttnParams p) {
  typedef bf16_t T;
  __shared__ __attribute__((aligned(1024))) char smem[NSLOT * SLOT_BYTES + 1024];
  __shared__ int redo_flag;
  const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
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
  const T* vb0 = reinterpret_cast<const T*>(p.v) + bo * p.v_bs0 + bi * p.v_bs1 + (long)head * HD * p.v_ts;
  const int b2 = b / p.k2_bdiv;
  const T* kb1 = has2 ? reinterpret_cast<const T*>(p.k2) + b2 * p.k2_bs + (long)head * HD : kb0;
  const T* vb1 = has2 ? reinterpret_cast<const T*>(p.v2) + b2 * p.v2_bs + (long)head * HD * p.v2_ts : vb0;
  // (descriptors are built at the point of use from wave-uniform 64-bit bases: a SELECT between two descriptors makes hipcc keep them in
  // memory and wrap every DMA in a waterfall loop with a full vmcnt(0) -- guide T20)
  auto uni = [](const T* ptr) {
    const unsigned long long a = reinterpret_cast<unsigned long long>(ptr);
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)a), hi = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
    return reinterpret_cast<const T*>(((unsigned long long)hi << 32) | lo);
  };
  const T* kseg = uni(kb0);
  const T* vseg = uni(vb0);
  int kts2 = __builtin_amdgcn_readfirstlane((int)p.k_ts * 2);     // bytes per key row of the current segment

  // constants of every slot: [1 1 0 0 0 0 0 0] (K padding against -M), eight ones (V^T row 40), zeros (rows 41 ..)
  if (tid < NSLOT * 12) {
    const int sl = tid / 12, w = tid % 12;
    unsigned v = 0;
    if (w == 0) v = 0x3F803F80u;                       // two bf16 ones
    if (w >= 4 && w < 8) v = 0x3F803F80u;
    *reinterpret_cast<unsigned*>(smem + sl * SLOT_BYTES + SLOT_CONST + 4 * w) = v;
  }

  // ---- DMA pieces of this wave: K piece `wid`, V^T piece `wid`, and the fifth K / V^T piece on waves 0 / 1.  Per-lane source offsets
  // (bytes from the segment's base; the tile's position goes into the scalar offset).
  unsigned offK = 0, offV = 0, offX = 0;
  auto seg_offsets = [&](bool s1) {
    const long kts = s1 ? p.k2_ts : p.k_ts, vts = s1 ? p.v2_ts : p.v_ts;
    auto koff = [&](int piece) {
      const int v = piece * 64 + lane, i = v / NVK, vc = v - i * NVK;
      const int key = (i & ~12) | ((i & 4) << 1) | ((i & 8) >> 1);
      return (unsigned)(key * kts * 2 + vc * 16);
    };
    auto voff = [&](int piece) {
      const int v = piece * 64 + lane, d = v >> 3, sl = v & 7, c = sl ^ ((d >> 1) & 7);
      return (unsigned)(d * vts * 2 + c * 16);
    };
    offK = koff(wid);
    offV = voff(wid);
    offX = wid == 0 ? koff(4) : voff(4);
  };
  seg_offsets(false);
  const int npieces = wid < 2 ? 3 : 2;
  auto issue_tile = [&](int it) {                      // the pieces of tile `it` into ring slot it % NSLOT (tiles strictly in order)
    if (it == nt0) {                                   // the bank segment starts: its bases, strides and per-lane offsets
      kseg = uni(kb1);
      vseg = uni(vb1);
      kts2 = __builtin_amdgcn_readfirstlane((int)p.k2_ts * 2);
      seg_offsets(true);
    }
    const int kt = it >= nt0 ? it - nt0 : it;
    char* slot = smem + (it & (NSLOT - 1)) * SLOT_BYTES;
    const int sK = kt * KT * kts2, sV = kt * KT * 2;
    const __amdgpu_buffer_rsrc_t rK = dma_rsrc(kseg), rV = dma_rsrc(vseg);
    blds16(rK, offK, sK, slot + wid * 1024);
    blds16(rV, offV, sV, slot + SLOT_K + wid * 1024);
    if (wid == 0) blds16(rK, offX, sK, slot + 4 * 1024);
    if (wid == 1) blds16(rV, offX, sV, slot + SLOT_K + 4 * 1024);
  };

  // ---- fragment read offsets inside a slot
  int kofs[KSQ];                                        // K rows of half tile 0 (half tile 1: + 32 rows)
#pragma unroll
  for (int ks = 0; ks < KSQ; ++ks) kofs[ks] = (ks == 2 && lh == 1) ? SLOT_CONST - 32 * HD * 2 * 0 : lr * (HD * 2) + ks * 32 + lh * 16;
  // (the padding vector is addressed slot-relative like the rows; body() adds the half tile's row offset only to real rows)
  const bool kpad = lh == 1;                            // lanes whose third K-step reads the constant vector
  int vofs[2][2][DT];                                   // [half tile][s2][dt]
#pragma unroll
  for (int hf = 0; hf < 2; ++hf)
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
      for (int dt = 0; dt < DT; ++dt) {
        const int d = 32 * dt + lr, c = 4 * hf + 2 * s2 + lh;
        vofs[hf][s2][dt] = d < HD ? SLOT_K + d * 128 + ((c ^ ((d >> 1) & 7)) << 4) : d == HD ? SLOT_CONST + 16 : SLOT_CONST + 32;
      }

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
  auto kfrag = [&](Frag<T>& f, const char* slot, int hf, int ks) __attribute__((always_inline)) {
    const int off = (ks == 2 && kpad) ? SLOT_CONST : kofs[ks] + hf * 32 * HD * 2;
    frag_load(f, reinterpret_cast<const T*>(slot + off));
  };

  // Fragment registers are double-buffered across bodies: a body multiplies with the K / first-V^T fragments its predecessor read for it and
  // reads its successor's behind its last score MFMA (the LDS round trip of five ds_read_b128 is off the wave's critical path; the
  // ablation that removed the fragment reads altogether ran 15 % faster).
  union VF { u32x4 u; Frag<T> f; };
  auto body = [&](auto fastc, f32x16 (&sp)[QB], f32x16 (&sn)[QB], Frag<T> (&kf)[KSQ], VF (&vf0)[DT], Frag<T> (&kfn)[KSQ], VF (&vf0n)[DT], const char* vslot, int vhf,
                  const char* kslot_n, int khf_n, const char* vslot_n, int vhf_n, bool decide, bool twin) __attribute__((always_inline)) {
    constexpr bool fast = decltype(fastc)::value;
    VF vf1[DT];
    float ex[2][16];
    Frag<T> pf[2][QB];
    auto exps = [&](int e0, int e1, int s2) __attribute__((always_inline)) {
#pragma unroll
      for (int e = e0; e < e1; ++e) PIN(sp[e >> 3][8 * s2 + (e & 7)]);
#pragma unroll
      for (int e = e0; e < e1; ++e) ex[e >> 3][8 * s2 + (e & 7)] = __builtin_amdgcn_exp2f(sp[e >> 3][8 * s2 + (e & 7)]);
#pragma unroll
      for (int e = e0; e < e1; ++e) {
        const int eq = e >> 3;
        if ((e & 7) == 7) {
          float p8[8];
#pragma unroll
          for (int j = 0; j < 8; ++j) p8[j] = ex[eq][8 * s2 + j];
          frag_set8(pf[s2][eq], p8);
          PIN(pf[s2][eq].v);
        } else {
          PIN(ex[eq][8 * s2 + (e & 7)]);
        }
      }
    };
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) vf1[dt].u = *reinterpret_cast<const u32x4*>(vslot + vofs[vhf][1][dt]);
    FENCE();
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
    // the successor's K fragments (this body's are consumed)
#pragma unroll
    for (int ks = 0; ks < KSQ; ++ks) kfrag(kfn[ks], kslot_n, khf_n, ks);
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const int dt = c >> 1, qb = c & 1;
      mma32(o[qb][dt], vf0[dt].f, pf[0][qb]);
      PIN(o[qb][dt]);
      exps(4 * c, 4 * c + 4, 1);
      FENCE();
    }
    // ... and its first V^T fragments
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) vf0n[dt].u = *reinterpret_cast<const u32x4*>(vslot_n + vofs[vhf_n][0][dt]);
    float mt[QB];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const int dt = c >> 1, qb = c & 1;
      mma32(o[qb][dt], vf1[dt].f, pf[1][qb]);
      PIN(o[qb][dt]);
      if (!fast && c < 2) {
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
    if constexpr (!fast) {
      if (decide && __any(fmaxf(mt[0], mt[1]) > RESCALE_LAG)) rescale(sn, mt, false);
    }
  };

  // One pass over the key tiles: prologue (tiles 0, 1, 2 on their way; the scores of half tile 0 set the softmax reference) + the tile loop.
  auto attend = [&](auto fastc) __attribute__((always_inline)) {
    issue_tile(0);
    if (ntiles > 1) issue_tile(1);
    if (ntiles > 2) issue_tile(2);
    if (ntiles > 2) { if (npieces == 3) wait_vmcnt<3>(); else wait_vmcnt<2>(); } else wait_vmcnt<0>();
    __syncthreads();
    f32x16 sA[QB], sB[QB];
    {
      Frag<T> kf[KSQ];
  #pragma unroll
      for (int ks = 0; ks < KSQ; ++ks) kfrag(kf[ks], smem, 0, ks);
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

    Frag<T> kfA[KSQ], kfB[KSQ];
    VF vfA[DT], vfB[DT];
  #pragma unroll
    for (int ks = 0; ks < KSQ; ++ks) kfrag(kfA[ks], smem, 1, ks);                                        // S(0, 1)
  #pragma unroll
    for (int dt = 0; dt < DT; ++dt) vfA[dt].u = *reinterpret_cast<const u32x4*>(smem + vofs[0][0][dt]);   // P.V(0, 0)
    for (int it = 0; it < ntiles; ++it) {
      if (it > 0) {
        // tile it + 1 (issued two iterations ago) has landed -- this wave's pieces: only those of tile it + 2 may still be in flight --
        // and, behind the barrier, everybody's; the barrier also frees slot (it + 3) % 4 = (it - 1) % 4, last read in iteration it - 1
        // (the fragments read ahead at the end of iteration it - 1 come from tile it's slot)
        if (it + 2 < ntiles) { if (npieces == 3) wait_vmcnt<3>(); else wait_vmcnt<2>(); } else wait_vmcnt<0>();
        __builtin_amdgcn_s_waitcnt(0xC07F);              // lgkmcnt(0): this wave's fragment reads have returned
        __builtin_amdgcn_s_barrier();
      }
      if (it + 3 < ntiles) issue_tile(it + 3);
      const char* cur = smem + (it & (NSLOT - 1)) * SLOT_BYTES;
      const char* nxt = smem + ((it + 1) & (NSLOT - 1)) * SLOT_BYTES;
      // S(it, 1) || P.V(it, 0); reads ahead: K(it + 1, 0), V(it, 1)
      body(fastc, sA, sB, kfA, vfA, kfB, vfB, cur, 0, nxt, 0, cur, 1, true, false);
      // S(it + 1, 0) || P.V(it, 1); reads ahead: K(it + 1, 1), V(it + 1, 0)
      body(fastc, sB, sA, kfB, vfB, kfA, vfA, cur, 1, nxt, 1, nxt, 0, it + 1 < ntiles, ob_twin && it == nt0 - 1);
    }
  };
  if constexpr (!FAST) {
    attend(std::false_type{});
  } else {
    if (tid == 0) redo_flag = 0;                           // (published by the first barrier of the pass)
    attend(std::true_type{});
    // the guard: a denominator beyond 2^100 (or inf / NaN: an exponent overflowed) in any row of the workgroup
    bool bad = false;
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) {
      constexpr int R = HD % 32, REG = (R & 3) + 4 * (R >> 3);
      bad |= !(fabsf(o[qb][HD / 32][REG]) < 1.2676506e30f);   // (lanes of the half that holds no denominator read a V^T row: finite too)
    }
    wait_vmcnt<0>();
    if (__any(bad) && lane == 0) redo_flag = 1;
    __syncthreads();
    if (redo_flag) {                                       // uniform over the workgroup: every wave has left the tile loop, no DMA is in flight
#pragma unroll
      for (int qb = 0; qb < QB; ++qb) {
        m_run[qb] = 0.f;
        if (lh == 1) {                                     // (the reference the first pass folded into the padding of the third K-step)
          qf[qb][KSQ - 1].set(0, 0.f);
          qf[qb][KSQ - 1].set(1, 0.f);
        }
#pragma unroll
        for (int i = 0; i < DT; ++i) o[qb][i] = (f32x16)(0.f);
      }
      kseg = uni(kb0);
      vseg = uni(vb0);
      kts2 = __builtin_amdgcn_readfirstlane((int)p.k_ts * 2);
      seg_offsets(false);
      attend(std::false_type{});
    }
  }
  store_out(ob);
}
#undef PIN
#undef FENCE

int g_attn_nomax = 1;   // mmgt_tune("attn_nomax", 0 / 1): the kernel without the running maximum (guarded), A/B switch

}  // namespace

void mmgt_attn_set_nomax(int v) { g_attn_nomax = v; }
int mmgt_attn_get_nomax() { return g_attn_nomax; }

// attention.hip's dispatcher: bf16, head_dim 40, V transposed, nq % 256 == 0, nk % 64 == 0, nk2 % 64 == 0
int mmgt_attn64_launch(const void* params, int batch, int heads, void* stream) {
  AttnParams p = *reinterpret_cast<const AttnParams*>(params);
  p.heads = heads;
  p.npairs = batch * heads;
  p.nqb = p.nq / (32 * QB * NW);
  if (g_attn_nomax) hipLaunchKernelGGL(attn64d_kernel<true>, dim3((unsigned)((long)p.nqb * batch * heads)), dim3(NT), 0, (hipStream_t)stream, p);
  else hipLaunchKernelGGL(attn64d_kernel<false>, dim3((unsigned)((long)p.nqb * batch * heads)), dim3(NT), 0, (hipStream_t)stream, p);
  MMGT_LAUNCH_CHECK();
  return 0;
}
