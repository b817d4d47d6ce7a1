// Flash-style attention for the MMGT Stage-2 path (gfx950): spatial self-attention with the ReferenceNet feature bank as
// a second key/value segment, MM-HAA audio cross-attention (32 keys) and temporal self-attention (<= 32 frames), all
// with head_dim in {40, 80, 160} (and 64 for the CLIP vision tower of the prologue).
//
// Structure per wave (32 queries): the score tile is computed TRANSPOSED, S^T[key][q] = K . Q^T, so a lane owns one
// query column: the online-softmax max / sum are in-register reductions plus one exchange between the two lane halves,
// and the exponentiated accumulator registers are directly the B operand of the second product
// O^T[d][q] += V^T[d][key] . P^T[key][q] (guide section 3, "an accumulator tile as the next MFMA's operand").
// K and V tiles (64 keys) are staged through LDS and shared by the workgroup's waves.
#include <type_traits>

#include "common.h"
#include "attn_common.h"
#include "mmgt_hip.h"

namespace {

// KT = keys per LDS tile: 64, or 32 for the short key sets (temporal attention over <= 32 frames, the 32 audio tokens),
// where a 64-key tile would spend half its MFMAs, exponentials and LDS on masked keys.
// RG = false: every key segment is a whole number of KT-key tiles (all the UNet's spatial shapes): the ragged-tail staging
// and masking code is not compiled in, so the hot loop has ONE staging path and no register shuffles where two would merge.
template <typename T, int HD, int NW, bool VT, int KT, bool RG = true>
__global__ __launch_bounds__(NW * 64, (NW == 4 && HD <= 40) ? 3 : 2) void attn_kernel(AttnParams p) {
  constexpr int NSUB = KT / 32;                 // 32-key score sub-tiles per LDS tile
  static_assert(KT == 32 || KT == 64, "KT");
  constexpr int ESZ = sizeof(T);
  constexpr int VEC = 16 / ESZ;                 // elements per 16-byte vector
  constexpr int HDK = (HD + 15) / 16 * 16;      // QK^T reduction length (zero padded)
  constexpr int KSQ = HDK / 16;
  constexpr int DT = (HD + 31) / 32;            // 32-row tiles of O^T
  constexpr int HDV = DT * 32;
  constexpr int RSK = HDK * ESZ + 16;           // K tile row stride (bytes): odd multiple of 16 -> conflict-free b128
  // bf16 V^T rows hold their keys PERMUTED inside every group of 16, [0-3, 8-11 | 4-7, 12-15]: the order the P fragment of a lane
  // half wants them, so a V fragment is ONE ds_read_b128.  (Two 8-byte reads 16 bytes apart are fused by hipcc into a
  // ds_read2_b64, which the LDS serves at a quarter of the b128 rate -- 16 LDS cycles per wave instruction against 4: with
  // twelve waves per CU the V^T fragment reads alone kept the LDS busy longer than a tile's MFMAs + softmax took, which is why
  // the matrix and vector pipes never overlapped in the round-1 counters.)  144-byte rows: an odd multiple of 16 bytes.
  constexpr bool VPERM = VT && ESZ == 2;
  constexpr int RSV = VT ? (KT * ESZ + 16) : (HDV * ESZ + 16);
  constexpr int VROWS = VT ? HDV : KT;
  constexpr int NT = NW * 64;
  // The kernel is VALU-bound at head_dim 40, so two per-score VALU operations ride in the MFMAs' zero padding instead:
  //  MFOLD: spare reduction slots of the score product (hd 40 -> 48) carry -M: Q'[q][HD..HD+1] = -(M_hi, M_lo) against
  //         K[key][HD..HD+1] = 1, so the MFMA starts from a literal-zero accumulator (no per-element v_mov of -M);
  //  LSUM:  a spare row of V^T (hd 40 -> 64, 80 -> 96) is all ones, so row HD of O^T accumulates the softmax denominator
  //         (of the probabilities as rounded for the P.V product) -- no per-element add.
  constexpr bool MFOLD = HDK - HD >= 2;
  constexpr bool LSUM = HDV > HD;
  // NBUF = 2 double-buffers the staged tiles (tile t+1 is written into the other buffer at the END of tile t's work, so
  // one barrier per tile both publishes it and retires the buffer it replaces).  Measured on MI355X: no gain over the
  // single buffer with two barriers (hd 40: 2816 vs 2812 us, hd 80: 290 vs 279 us) at twice the LDS, so it stays off.
  constexpr int TILE_BYTES = KT * RSK + VROWS * RSV;
  constexpr int NBUF = 1;
  __shared__ __attribute__((aligned(16))) char smem[NBUF * TILE_BYTES];

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int lr = lane & 31, lh = lane >> 5;
  // (Measured and rejected: software-pipelining the scores of tile t+1 under the softmax of tile t (K staged one tile ahead
  // of V, correct on the first run) -- with hipcc's schedule 2985 us at two waves per SIMD, 10.6 ms at three (116 spilled
  // VGPRs) against 2504 us for this loop at hd 40, 285 vs 262 us at hd 80; it needs a hand-placed instruction stream.)
  // (Measured and rejected: a start-up stagger of the co-resident workgroups by fractions of a tile -- s_sleep of
  // 128..1536 cycles keyed on several workgroup-id bit fields -- changed nothing, 2445 +- 10 us at hd 40: the resident
  // workgroups are not in lockstep.)
  // XCD-aware mapping (1-D grid): workgroup ids are dealt round-robin over the 8 XCDs, so id = slot * 8 + xcd.  All query
  // blocks of one (batch, head) pair are placed on ONE XCD (pair = xcd + 8 * (slot / nqb)): their K/V tiles are then
  // re-read from that XCD's L2 instead of crossing the fabric once per query block.
  int pair, qblk;
  {
    const int nqb = p.nqb, npairs = p.npairs;
    const int id = blockIdx.x;
    if (p.heads_inner && ((npairs / p.heads) & 7) == 0) {
      // few keys, heads packed in the row (MM-HAA's 24-head audio cross-attention: 80-byte head segments of 1920-byte rows): all heads of
      // one (batch entry, query block) run back to back on ONE XCD, so a row's cache lines cross the fabric once, not once per head
      const int xcd = id & 7, slot = id >> 3, per = nqb * p.heads;
      const int bb = xcd + 8 * (slot / per), r = slot - (slot / per) * per;
      qblk = r / p.heads;
      pair = bb * p.heads + (r - qblk * p.heads);
    } else if ((npairs & 7) == 0) {
      const int xcd = id & 7, slot = id >> 3;
      pair = xcd + 8 * (slot / nqb);
      qblk = slot % nqb;
    } else {
      pair = id / nqb;
      qblk = id - pair * nqb;
    }
  }
  const int b = pair / p.heads, head = pair - b * p.heads;
  const int bo = b / p.bdiv, bi = b - bo * p.bdiv;
  const int q0 = (qblk * NW + wid) * 32;

  const T* qb = reinterpret_cast<const T*>(p.q) + bo * p.q_bs0 + bi * p.q_bs1 + (long)head * HD;
  T* ob = reinterpret_cast<T*>(p.o) + bo * p.o_bs0 + bi * p.o_bs1 + (long)head * HD;

  // ---- Q^T fragments (B operand of S^T = K . Q^T): lane (q = lr, half lh) holds d = 16 ks + 8 lh + j ----
  Frag<T> qf[KSQ];
  {
    int qi = q0 + lr;
    if (qi >= p.nq) qi = p.nq - 1;
    if (qi < 0) qi = 0;
    const T* qrow = qb + (long)qi * p.q_ts;
#pragma unroll
    for (int ks = 0; ks < KSQ; ++ks) {
      const int d = 16 * ks + 8 * lh;
      if (d < HD) frag_load(qf[ks], qrow + d);
      else qf[ks].zero();
      // fold softmax scale * log2(e) into Q once: the MFMA then yields scores directly in the exp2 domain
      {
        float q8[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) q8[j] = frag_get(qf[ks], j) * p.scale_log2e;
        frag_set8(qf[ks], q8);
      }
    }
  }

  f32x16 o[DT];
#pragma unroll
  for (int i = 0; i < DT; ++i) o[i] = (f32x16)(0.f);
  // Online softmax in the log2 domain.  The running reference M enters the score MFMA chain as its initial accumulator
  // (-M per query column), so exp2 applies to the accumulator as it stands: no per-element subtract (the kernel is
  // VALU-bound).  When some query's tile maximum exceeds its reference (wave-uniform test) the tile takes the slow path:
  // shift the scores, move M, rescale O and l -- exactly the classic update.
  float m_run = 0.f, l_run = 0.f;

  // ---- tile schedule: segment 0 = own keys, segment 1 = bank keys (conditional CFG half only) ----
  const bool has2 = p.k2 != nullptr && p.nk2 > 0 && b >= p.seg2_first_batch;
  const int nt0 = (p.nk + KT - 1) / KT;
  const int ntiles = nt0 + (has2 ? (p.nk2 + KT - 1) / KT : 0);
  const T* kb0 = reinterpret_cast<const T*>(p.k) + bo * p.k_bs0 + bi * p.k_bs1 + (long)head * HD;
  const T* vb0 = reinterpret_cast<const T*>(p.v) + bo * p.v_bs0 + bi * p.v_bs1;
  const int b2 = b / p.k2_bdiv;
  const T* kb1 = has2 ? reinterpret_cast<const T*>(p.k2) + b2 * p.k2_bs + (long)head * HD : kb0;
  const T* vb1 = has2 ? reinterpret_cast<const T*>(p.v2) + b2 * p.v2_bs : vb0;

  // Only the HD valid columns (K) / rows (V^T) are ever staged; the zero padding up to HDK / HDV is written once.
  constexpr int NVK = HD / VEC;                          // valid 16-byte vectors per K row
  constexpr int NVV = VT ? KT / VEC : HD / VEC;          // vectors per V tile row
  constexpr int VR = VT ? HD : KT;                       // staged V tile rows
  constexpr bool PF = NW >= 4;                           // register prefetch only where a thread stages few vectors
  constexpr int KVEC = PF ? (KT * NVK + NT - 1) / NT : 1;
  constexpr int VVEC = PF ? (VR * NVV + NT - 1) / NT : 1;
  u32x4 rk[KVEC], rv[VVEC];
  for (int i = tid * 16; i < NBUF * TILE_BYTES; i += NT * 16) *reinterpret_cast<u32x4*>(smem + i) = (u32x4)(0u);
  if (MFOLD || LSUM) {
    __syncthreads();   // the padding constants below overwrite zeros written by other threads
    for (int r = tid; r < NBUF * KT; r += NT) {
      char* bK = smem + (r / KT) * TILE_BYTES;
      char* bV = bK + KT * RSK;
      const int rr = r % KT;
      if (MFOLD) {
        Elem<T>::st(reinterpret_cast<T*>(bK + rr * RSK) + HD, 1.f);
        Elem<T>::st(reinterpret_cast<T*>(bK + rr * RSK) + HD + 1, 1.f);
      }
      if (LSUM) {
        if (VT) Elem<T>::st(reinterpret_cast<T*>(bV + HD * RSV) + rr, 1.f);
        else Elem<T>::st(reinterpret_cast<T*>(bV + rr * RSV) + HD, 1.f);
      }
    }
  }

  struct TileSrc { const T *kb, *vb; long kts, vts; int nks, kt; };
  auto tile_src = [&](int it) {
    TileSrc t;
    const bool s1 = it >= nt0;
    t.kb = s1 ? kb1 : kb0;
    t.vb = s1 ? vb1 : vb0;
    t.kts = s1 ? p.k2_ts : p.k_ts;
    t.vts = s1 ? p.v2_ts : p.v_ts;
    t.nks = s1 ? p.nk2 : p.nk;
    t.kt = (s1 ? it - nt0 : it) * KT;
    return t;
  };
  // One staged vector of K / V.  Rows past the end of the segment are CLAMPED to the last valid row (finite data; their
  // scores are masked / their probabilities are exactly 0), so full tiles run without any per-vector branch.
  auto load_k = [&](const TileSrc& t, int idx) -> u32x4 {
    const int row = idx / NVK, vc = idx - row * NVK;
    int key = t.kt + row;
    key = key < t.nks ? key : t.nks - 1;
    return *reinterpret_cast<const u32x4*>(t.kb + (long)key * t.kts + vc * VEC);
  };
  auto load_v = [&](const TileSrc& t, int idx, bool full) -> u32x4 {
    const int row = idx / NVV, vc = idx - row * NVV;
    if (VT) {
      const int key0 = t.kt + vc * VEC;
      const T* src = t.vb + ((long)head * HD + row) * t.vts + key0;
      if (full || key0 + VEC <= t.nks) return *reinterpret_cast<const u32x4*>(src);
      union { u32x4 v; T e[VEC]; } val;       // ragged tail of the last tile: element-wise, zero filled
      val.v = (u32x4)(0u);
      for (int e = 0; e < VEC; ++e)
        if (key0 + e < t.nks) val.e[e] = src[e];
      return val.v;
    } else {
      int key = t.kt + row;
      key = key < t.nks ? key : t.nks - 1;
      return *reinterpret_cast<const u32x4*>(t.vb + (long)key * t.vts + (long)head * HD + vc * VEC);
    }
  };
  auto store_k = [&](char* lK, int idx, u32x4 v) {
    const int row = idx / NVK, vc = idx - row * NVK;
    *reinterpret_cast<u32x4*>(lK + row * RSK + vc * 16) = v;
  };
  auto store_v = [&](char* lV, int idx, u32x4 v) {
    const int row = idx / NVV, vc = idx - row * NVV;
    if (VPERM) {  // vector vc = keys 8 vc .. 8 vc + 7: its halves go to 8-byte slots (vc & 1) and 2 + (vc & 1) of key group vc >> 1
      u32x2* dst = reinterpret_cast<u32x2*>(lV + row * RSV + (vc >> 1) * 32 + (vc & 1) * 8);
      dst[0] = (u32x2){v[0], v[1]};
      dst[2] = (u32x2){v[2], v[3]};
    } else {
      *reinterpret_cast<u32x4*>(lV + row * RSV + vc * 16) = v;
    }
  };

  // issue-early / write-late staging (guide T14): the next tile's global loads fly under the current tile's MFMAs.
  // Full tiles use running per-thread pointers (two adds per vector per tile: the kernel is VALU-bound, so the address
  // arithmetic matters); the ragged last tile of a segment recomputes clamped / zero-filled addresses.
  const T* pk[KVEC];
  const T* pv[VVEC];
  auto seg_init = [&](const TileSrc& t) {
#pragma unroll
    for (int i = 0; i < KVEC; ++i) {
      const int idx = tid + i * NT;
      const int row = idx / NVK, vc = idx - row * NVK;
      pk[i] = t.kb + (long)row * t.kts + vc * VEC;
    }
#pragma unroll
    for (int i = 0; i < VVEC; ++i) {
      const int idx = tid + i * NT;
      const int row = idx / NVV, vc = idx - row * NVV;
      pv[i] = VT ? t.vb + ((long)head * HD + row) * t.vts + vc * VEC
                 : t.vb + (long)row * t.vts + (long)head * HD + vc * VEC;
    }
  };
  auto prefetch = [&](int it) {
    const TileSrc t = tile_src(it);
    if (HD <= 80 && t.kt == 0) seg_init(t);
    if (HD <= 80 && (!RG || t.kt + KT <= t.nks)) {      // wave-uniform: full tiles take the branch-free path
      const long kstep = (long)KT * t.kts, vstep = VT ? (long)KT : (long)KT * t.vts;
#pragma unroll
      for (int i = 0; i < KVEC; ++i) {
        const int idx = tid + i * NT;
        if ((i + 1) * NT <= KT * NVK || idx < KT * NVK) rk[i] = *reinterpret_cast<const u32x4*>(pk[i]);
        pk[i] += kstep;
      }
#pragma unroll
      for (int i = 0; i < VVEC; ++i) {
        const int idx = tid + i * NT;
        if ((i + 1) * NT <= VR * NVV || idx < VR * NVV) rv[i] = *reinterpret_cast<const u32x4*>(pv[i]);
        pv[i] += vstep;
      }
    } else {
#pragma unroll
      for (int i = 0; i < KVEC; ++i) {
        const int idx = tid + i * NT;
        if ((i + 1) * NT <= KT * NVK || idx < KT * NVK) rk[i] = load_k(t, idx);
      }
#pragma unroll
      for (int i = 0; i < VVEC; ++i) {
        const int idx = tid + i * NT;
        if ((i + 1) * NT <= VR * NVV || idx < VR * NVV) rv[i] = load_v(t, idx, HD > 80 && (!RG || t.kt + KT <= t.nks));
      }
    }
  };
  auto commit = [&](int buf) {
    char* bK = smem + buf * TILE_BYTES;
    char* bV = bK + KT * RSK;
#pragma unroll
    for (int i = 0; i < KVEC; ++i) {
      const int idx = tid + i * NT;
      if ((i + 1) * NT <= KT * NVK || idx < KT * NVK) store_k(bK, idx, rk[i]);
    }
#pragma unroll
    for (int i = 0; i < VVEC; ++i) {
      const int idx = tid + i * NT;
      if ((i + 1) * NT <= VR * NVV || idx < VR * NVV) store_v(bV, idx, rv[i]);
    }
  };
  // single-wave variant (temporal / tiny sequences): straight global -> LDS staging, rolled loops, no prefetch registers
  auto stage_direct = [&](int it) {
    const TileSrc t = tile_src(it);
    for (int idx = tid; idx < KT * NVK; idx += NT) store_k(smem, idx, load_k(t, idx));
    for (int idx = tid; idx < VR * NVV; idx += NT) store_v(smem + KT * RSK, idx, load_v(t, idx, false));
  };

  if (PF) prefetch(0);
  if (NBUF == 2) commit(0);
  for (int it = 0; it < ntiles; ++it) {
    const bool s1 = it >= nt0;
    const int nks = s1 ? p.nk2 : p.nk;
    const int kt = (s1 ? it - nt0 : it) * KT;
    const char* lK = smem + (NBUF == 2 ? (it & 1) * TILE_BYTES : 0);
    const char* lV = lK + KT * RSK;
    if (NBUF == 2) {
      __syncthreads();   // tile `it` is visible, and every wave has left the buffer tile it + 1 will replace
    } else {
      __syncthreads();   // every wave has finished reading the previous tile
      if (PF) commit(0);
      else stage_direct(it);
      __syncthreads();
    }
    if (PF && it + 1 < ntiles) prefetch(it + 1);

    // ---- S^T - M = K . Q'^T - M for the two 32-key sub-tiles ----
    f32x16 s[NSUB];
    {
      // all K fragments of the tile are requested before the first MFMA (where registers allow: hd <= 80), so the two
      // score chains wait for LDS once instead of once per K-step
      constexpr bool BATCH = HD <= 80;
      Frag<T> kf[BATCH ? NSUB : 1][BATCH ? KSQ : 1];
      if (BATCH) {
#pragma unroll
        for (int sub = 0; sub < NSUB; ++sub)
#pragma unroll
          for (int ks = 0; ks < KSQ; ++ks)
            frag_load(kf[BATCH ? sub : 0][BATCH ? ks : 0],
                      reinterpret_cast<const T*>(lK + (sub * 32 + lr) * RSK + lh * 8 * ESZ + ks * 16 * ESZ));
      }
#pragma unroll
      for (int sub = 0; sub < NSUB; ++sub) s[sub] = MFOLD ? (f32x16)(0.f) : (f32x16)(-m_run);
      // the two chains interleaved K-step by K-step: consecutive MFMAs are independent
#pragma unroll
      for (int ks = 0; ks < KSQ; ++ks)
#pragma unroll
        for (int sub = 0; sub < NSUB; ++sub) {
          if (BATCH) {
            mma32(s[sub], kf[BATCH ? sub : 0][BATCH ? ks : 0], qf[ks]);
          } else {
            Frag<T> k1;
            frag_load(k1, reinterpret_cast<const T*>(lK + (sub * 32 + lr) * RSK + lh * 8 * ESZ + ks * 16 * ESZ));
            mma32(s[sub], k1, qf[ks]);
          }
        }
      if (BATCH && ESZ == 2) {
        // keep the scheduler from re-serialising read -> wait -> MFMA pairs: all DS reads first, then the MFMAs
        __builtin_amdgcn_sched_group_barrier(0x100, NSUB * KSQ, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, NSUB * KSQ, 0);
      }
    }
    if (RG && kt + KT > nks) {   // ragged last tile only: mask the keys past the end
#pragma unroll
      for (int sub = 0; sub < NSUB; ++sub)
#pragma unroll
        for (int r = 0; r < 16; ++r)
          if (kt + sub * 32 + acc_row(r, lane) >= nks) s[sub][r] = -1e30f;
    }
    // two independent v_max3 chains (running, s0[r], s1[r]): 16 instructions for the 32 scores instead of 24
    float mt = fmaxf(s[0][0], s[NSUB - 1][0]), mt2 = fmaxf(s[0][1], s[NSUB - 1][1]);
#pragma unroll
    for (int r = 2; r < 16; r += 2) {
      mt = fmaxf(fmaxf(mt, s[0][r]), s[NSUB - 1][r]);
      mt2 = fmaxf(fmaxf(mt2, s[0][r + 1]), s[NSUB - 1][r + 1]);
    }
    mt = fmaxf(mt, mt2);
    {
      // the other lane half's maximum by v_permlane32_swap (a VALU move): __shfl_xor(mt, 32) is a ds_bpermute, an LDS
      // round trip in the middle of the tile's critical path.  swap(a, b) exchanges a's upper half with b's lower half,
      // so with a == b one result holds the own value and the other the partner's, in either lane half.
      const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(mt), __float_as_uint(mt), false, false);
      mt = fmaxf(__uint_as_float(sw[0]), __uint_as_float(sw[1]));
    }
    // Lazy rescale: the reference M of a row may lag its true maximum by up to RESCALE_LAG (log2 units): p = 2^(s - M) <= 2^LAG
    // is exact business for the fp32 row sums / accumulators and for the bf16 P operand alike (relative precision does not depend
    // on the scale), and the o *= alpha / s -= delta pass (80 vector instructions of a tile's ~130) runs on the few tiles where
    // some row's maximum jumps instead of on every tile in which any of the wave's 32 rows sets a new record (about half).
    constexpr float RESCALE_LAG = 8.f;
    if (it == 0 || __any(mt > RESCALE_LAG)) {
      // the first tile fixes M at the tile maximum (either sign); later tiles only ever raise it
      float delta = it == 0 ? mt : fmaxf(mt, 0.f);
      if (MFOLD) {
        // the reference the MFMA subtracts is M_hi + M_lo in the storage type: move M to the nearest such value
        const float m_new = m_run + delta;
        const float hi = Elem<T>::cvt(m_new), lo = Elem<T>::cvt(m_new - hi);
        delta = (hi + lo) - m_run;
        if (lh == 1) {   // lanes holding d = HD .. HD + 7 of the last K-step
          qf[KSQ - 1].set(0, -hi);
          qf[KSQ - 1].set(1, -lo);
        }
      }
      const float alpha = __builtin_amdgcn_exp2f(-delta);
      m_run += delta;
      if (!LSUM) l_run *= alpha;
#pragma unroll
      for (int i = 0; i < DT; ++i) o[i] *= alpha;
#pragma unroll
      for (int sub = 0; sub < NSUB; ++sub) s[sub] -= delta;
    }
    float ls = 0.f;
#pragma unroll
    for (int sub = 0; sub < NSUB; ++sub)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float pv = __builtin_amdgcn_exp2f(s[sub][r]);
        s[sub][r] = pv;
        if (!LSUM) ls += pv;
      }
    if (!LSUM) l_run += ls;

    // ---- O^T += V^T . P^T ----
#pragma unroll
    for (int sub = 0; sub < NSUB; ++sub)
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        Frag<T> pf;
        {
          float p8[8];
#pragma unroll
          for (int j = 0; j < 8; ++j) p8[j] = s[sub][8 * s2 + j];
          frag_set8(pf, p8);
        }
        // element j of this lane <-> key  sub*32 + 16*s2 + 8*(j>>2) + 4*lh + (j&3)
        const int kbase = sub * 32 + 16 * s2 + 4 * lh;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
          Frag<T> vf;
          const int d = dt * 32 + lr;
          if (VT) {
            const T* vp = reinterpret_cast<const T*>(lV + d * RSV) + kbase;
            if (ESZ == 2) {
              union { u32x4 u; Frag<T> f; } cv;    // the lane's 8 keys are 16 contiguous bytes of the permuted row
              cv.u = *reinterpret_cast<const u32x4*>(lV + d * RSV + (sub * 2 + s2) * 32 + lh * 16);
              vf = cv.f;
            } else {
              const f32x4 lo = *reinterpret_cast<const f32x4*>(vp);
              const f32x4 hi = *reinterpret_cast<const f32x4*>(vp + 8);
              union { float f[8]; Frag<T> fr; } cv;
#pragma unroll
              for (int j = 0; j < 4; ++j) { cv.f[j] = lo[j]; cv.f[4 + j] = hi[j]; }
              vf = cv.fr;
            }
          } else {
            union { T e[8]; Frag<T> f; } cv;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
              const int key = kbase + 8 * (j >> 2) + (j & 3);
              cv.e[j] = *(reinterpret_cast<const T*>(lV + key * RSV) + d);
            }
            vf = cv.f;
          }
          mma32(o[dt], vf, pf);
        }
      }
    if (NBUF == 2 && it + 1 < ntiles) commit((it + 1) & 1);
  }

  // ---- normalise and store: lane (q, half) owns d = 32 dt + 8 g + 4 half + (0..3) ----
  float l_tot;
  if (LSUM) {   // row HD of O^T: accumulator register and lane half that hold it
    constexpr int R = HD % 32, REG = (R & 3) + 4 * (R >> 3), LHS = (R >> 2) & 1;
    const float mine = o[HD / 32][REG], other = __shfl_xor(mine, 32);
    l_tot = lh == LHS ? mine : other;
  } else {
    l_tot = l_run + __shfl_xor(l_run, 32);
  }
  float inv = 1.f / l_tot;
  const int qi = q0 + lr;
  if (p.out_scale && qi < p.nq) inv *= p.out_scale[(long)(head / p.os_heads) * p.os_gs + (long)b * p.nq + qi];
  if (qi < p.nq) {
    T* orow = ob + (long)qi * p.o_ts;
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int d = dt * 32 + 8 * g + 4 * lh;
        if (d < HD) {
          if (ESZ == 2) {
            union { bf16_t e[4]; u32x2 u; } pk;
#pragma unroll
            for (int e = 0; e < 4; ++e) pk.e[e] = f32_to_bf16(o[dt][4 * g + e] * inv);
            *reinterpret_cast<u32x2*>(orow + d) = pk.u;
          } else {
            f32x4 pk;
#pragma unroll
            for (int e = 0; e < 4; ++e) pk[e] = o[dt][4 * g + e] * inv;
            *reinterpret_cast<f32x4*>(orow + d) = pk;
          }
        }
      }
  }
}

int g_heads_inner = 1;   // mmgt_tune("attn_heads_inner", 0 / 1): A/B switch of the heads-inner workgroup order (few keys, packed heads)
int g_attn64 = 1;   // mmgt_tune("attn64", 0 / 1): the 64-queries-per-wave kernel of attn64.hip (A/B switch)

}  // namespace
int mmgt_attn64_launch(const void* params, int batch, int heads, void* stream);
int mmgt_attn80_launch(const void* params, int batch, int heads, void* stream);   // attn80.hip (-1: switched off)
void mmgt_attn_set64(int v) { g_attn64 = v; }
void mmgt_attn_set_heads_inner(int v) { g_heads_inner = v; }
namespace {

template <typename T, int HD>
int launch_hd(AttnParams p, int batch, int heads, int vt, hipStream_t s) {

  // Short sequences (temporal attention, <= 32 frames) use one wave per workgroup; spatial sequences four.
  p.heads = heads;
  p.npairs = batch * heads;
  const bool short_keys = !vt && p.nk <= 32 && p.nk2 <= 32;   // every key segment fits one 32-key tile
  if (p.nq <= 32) {
    p.nqb = 1;
    dim3 grid(batch * heads);
    if (vt) hipLaunchKernelGGL((attn_kernel<T, HD, 1, true, 64>), grid, dim3(64), 0, s, p);
    else if (short_keys) hipLaunchKernelGGL((attn_kernel<T, HD, 1, false, 32>), grid, dim3(64), 0, s, p);
    else hipLaunchKernelGGL((attn_kernel<T, HD, 1, false, 64>), grid, dim3(64), 0, s, p);
  } else {
    p.nqb = (p.nq + 127) / 128;
    dim3 grid((unsigned)((long)p.nqb * batch * heads));
    const bool whole = p.nk % 64 == 0 && p.nk2 % 64 == 0;
    if (std::is_same<T, bf16_t>::value && HD == 40 && vt && whole && p.nq % 256 == 0 && g_attn64 && !p.out_scale)
      return mmgt_attn64_launch(&p, batch, heads, s);
    if (std::is_same<T, bf16_t>::value && HD == 80 && vt && whole && p.nq % 256 == 0 && !p.out_scale && !p.o_twin) {
      const int rc = mmgt_attn80_launch(&p, batch, heads, s);       // (-1: switched off)
      if (rc >= 0) return rc;
    }
    if (vt && whole) hipLaunchKernelGGL((attn_kernel<T, HD, 4, true, 64, false>), grid, dim3(256), 0, s, p);
    else if (vt) hipLaunchKernelGGL((attn_kernel<T, HD, 4, true, 64>), grid, dim3(256), 0, s, p);
    else if (short_keys) hipLaunchKernelGGL((attn_kernel<T, HD, 4, false, 32>), grid, dim3(256), 0, s, p);
    else hipLaunchKernelGGL((attn_kernel<T, HD, 4, false, 64>), grid, dim3(256), 0, s, p);
  }
  MMGT_LAUNCH_CHECK();
  return 0;
}

template <typename T>
int launch_t(const AttnParams& p, int batch, int heads, int hd, int vt, hipStream_t s) {
  switch (hd) {
    case 40: return launch_hd<T, 40>(p, batch, heads, vt, s);
    case 64: return launch_hd<T, 64>(p, batch, heads, vt, s);   // CLIP vision tower
    case 80: return launch_hd<T, 80>(p, batch, heads, vt, s);
    case 160: return launch_hd<T, 160>(p, batch, heads, vt, s);
    default: mmgt_set_error("attention: head_dim %d unsupported (40, 64, 80, 160)", hd); return 1;
  }
}

}  // namespace

// tattn.hip: the motion modules' temporal self-attention (one sequence = the <= 32 frames of one pixel)
int mmgt_tattn_try(const void* q, long q_bs0, long q_bs1, long q_ts, const void* k, long k_bs0, long k_bs1, long k_ts,
                   const void* v, long v_bs0, long v_bs1, long v_ts, void* o, long o_bs0, long o_bs1, long o_ts, int bdiv,
                   int batch, int heads, int hd, int frames, float scale, int dtype, void* stream);

static int attention_entry(const void* q, long q_bs0, long q_bs1, long q_ts, const void* k, long k_bs0, long k_bs1,
                           long k_ts, const void* v, long v_bs0, long v_bs1, long v_ts, void* o, long o_bs0,
                           long o_bs1, long o_ts, int bdiv, const void* k2, const void* v2, long k2_bs, long k2_ts,
                           long v2_bs, long v2_ts, int k2_bdiv, int nk2, int seg2_first_batch, int batch, int heads,
                           int hd, int nq, int nk, float scale, int v_transposed, const float* out_scale, long os_gs, int os_heads,
                           void* o_twin, int dtype, void* stream) {
  MMGT_CHECK(q && k && v && o, "attention: null pointer");
  MMGT_CHECK(dtype == MMGT_F32 || dtype == MMGT_BF16, "attention: bad dtype %d", dtype);
  MMGT_CHECK(batch > 0 && heads > 0 && nq > 0 && nk > 0 && bdiv > 0, "attention: empty problem");
  MMGT_CHECK((long)batch * heads * ((nq + 127) / 128) < (1l << 31), "attention: grid too large");
  MMGT_CHECK((k2 == nullptr) == (v2 == nullptr), "attention: k2/v2 must come together");
  MMGT_CHECK(!k2 || (k2_bdiv > 0 && nk2 >= 0), "attention: bad second segment");
  const int esz = dtype == MMGT_BF16 ? 2 : 4;
  const long vec = 16 / esz;
  MMGT_CHECK(q_ts % vec == 0 && k_ts % vec == 0 && o_ts % vec == 0 && q_bs0 % vec == 0 && q_bs1 % vec == 0 &&
                 k_bs0 % vec == 0 && k_bs1 % vec == 0 && v_bs0 % vec == 0 && v_bs1 % vec == 0 && v_ts % vec == 0,
             "attention: strides must keep 16-byte alignment");
  MMGT_CHECK(!k2 || (k2_ts % vec == 0 && k2_bs % vec == 0 && v2_ts % vec == 0 && v2_bs % vec == 0),
             "attention: segment-2 strides must keep 16-byte alignment");
  MMGT_CHECK(!out_scale || (os_heads > 0 && heads % os_heads == 0 && bdiv == 1 && !v_transposed),
             "attention: out_scale needs heads %% os_heads == 0, bdiv == 1 and row-major V");
  if (nq == nk && nk <= 32 && !k2 && !v_transposed && !out_scale) {   // temporal pattern: the memory-stream kernel of tattn.hip
    const int rc = mmgt_tattn_try(q, q_bs0, q_bs1, q_ts, k, k_bs0, k_bs1, k_ts, v, v_bs0, v_bs1, v_ts, o, o_bs0, o_bs1, o_ts, bdiv,
                                  batch, heads, hd, nq, scale, dtype, stream);
    if (rc >= 0) return rc;
  }
  AttnParams p{};
  p.q = (const char*)q; p.k = (const char*)k; p.v = (const char*)v; p.k2 = (const char*)k2; p.v2 = (const char*)v2;
  p.o = (char*)o;
  p.q_bs0 = q_bs0; p.q_bs1 = q_bs1; p.q_ts = q_ts; p.k_bs0 = k_bs0; p.k_bs1 = k_bs1; p.k_ts = k_ts;
  p.v_bs0 = v_bs0; p.v_bs1 = v_bs1; p.v_ts = v_ts; p.o_bs0 = o_bs0; p.o_bs1 = o_bs1; p.o_ts = o_ts;
  p.k2_bs = k2_bs; p.k2_ts = k2_ts; p.v2_bs = v2_bs; p.v2_ts = v2_ts;
  p.bdiv = bdiv; p.k2_bdiv = k2 ? k2_bdiv : 1; p.nk2 = k2 ? nk2 : 0; p.seg2_first_batch = seg2_first_batch;
  p.nq = nq; p.nk = nk;
  p.scale_log2e = scale * 1.4426950408889634f;
  p.out_scale = out_scale; p.os_gs = os_gs; p.os_heads = out_scale ? os_heads : 1;
  p.o_twin = (char*)o_twin;
  MMGT_CHECK(!o_twin || (k2 && nk2 > 0 && seg2_first_batch == 0 && !out_scale && dtype == MMGT_BF16 && hd == 40 && v_transposed && nq % 256 == 0 &&
                         nk % 64 == 0 && nk2 % 64 == 0 && nq > 32 && g_attn64),
             "attention_twin: needs the 64-queries-per-wave kernel (bf16, head_dim 40, V transposed, nq %% 256 == 0, whole 64-key tiles) and a second "
             "key segment read by every batch entry");
  p.heads_inner = g_heads_inner && !k2 && !v_transposed && nk <= 32 && nq > 32 && q_bs1 == 0 && o_bs1 == 0 && heads > 1;
  hipStream_t s = (hipStream_t)stream;
  return dtype == MMGT_BF16 ? launch_t<bf16_t>(p, batch, heads, hd, v_transposed, s)
                            : launch_t<float>(p, batch, heads, hd, v_transposed, s);
}

extern "C" int mmgt_attention(const void* q, long q_bs0, long q_bs1, long q_ts, const void* k, long k_bs0, long k_bs1,
                              long k_ts, const void* v, long v_bs0, long v_bs1, long v_ts, void* o, long o_bs0,
                              long o_bs1, long o_ts, int bdiv, const void* k2, const void* v2, long k2_bs, long k2_ts,
                              long v2_bs, long v2_ts, int k2_bdiv, int nk2, int seg2_first_batch, int batch, int heads,
                              int hd, int nq, int nk, float scale, int v_transposed, int dtype, void* stream) {
  return attention_entry(q, q_bs0, q_bs1, q_ts, k, k_bs0, k_bs1, k_ts, v, v_bs0, v_bs1, v_ts, o, o_bs0, o_bs1, o_ts, bdiv, k2, v2, k2_bs,
                         k2_ts, v2_bs, v2_ts, k2_bdiv, nk2, seg2_first_batch, batch, heads, hd, nq, nk, scale, v_transposed, nullptr, 0, 1,
                         nullptr, dtype, stream);
}

// The same with a per-row output multiplier per group of `os_heads` heads:  o[b][q][head] *= out_scale[(head / os_heads) * os_gs + b * nq + q]
// (applied in fp32 with the softmax normalisation).  MM-HAA's three masked audio cross-attentions (attention.py:730-760) write
// mask_i * attn2_i(x, audio_i) directly, so that the three  zero_conv_i(mask_i * to_out_i(.))  projections run as ONE GEMM over the
// concatenated reduction (mmgt_amd/unet3d.py).  Single key segment, row-major V, bdiv == 1.
extern "C" int mmgt_attention_scaled(const void* q, long q_bs0, long q_bs1, long q_ts, const void* k, long k_bs0, long k_bs1, long k_ts,
                                     const void* v, long v_bs0, long v_bs1, long v_ts, void* o, long o_bs0, long o_bs1, long o_ts,
                                     const float* out_scale, long os_group_stride, int os_heads, int batch, int heads, int hd, int nq, int nk,
                                     float scale, int dtype, void* stream) {
  MMGT_CHECK(out_scale, "attention_scaled: null out_scale");
  return attention_entry(q, q_bs0, q_bs1, q_ts, k, k_bs0, k_bs1, k_ts, v, v_bs0, v_bs1, v_ts, o, o_bs0, o_bs1, o_ts, 1, nullptr, nullptr, 0,
                         0, 0, 0, 1, 0, 0, batch, heads, hd, nq, nk, scale, 0, out_scale, os_group_stride, os_heads, nullptr, dtype, stream);
}

// mmgt_attention whose every batch entry reads a second key segment, with a TWIN output: the attention over the first segment alone
// (the state of the online softmax after its last tile) is written to o_twin with o's strides, the attention over both segments to o.
// The CFG pair of the first reference-attention reader (mutual_self_attention.py:160-230): both rows enter with the same hidden states, the
// conditional row attends [x | bank], the unconditional row [x] -- one pass over x serves both.  bf16, head_dim 40, V transposed,
// nq % 256 == 0, nk % 64 == 0, nk2 % 64 == 0 (the 64-queries-per-wave kernel of attn64.hip); anything else is an error.
extern "C" int mmgt_attention_twin(const void* q, long q_bs0, long q_bs1, long q_ts, const void* k, long k_bs0, long k_bs1, long k_ts,
                                   const void* v, long v_bs0, long v_bs1, long v_ts, void* o, void* o_twin, long o_bs0, long o_bs1, long o_ts,
                                   int bdiv, const void* k2, const void* v2, long k2_bs, long k2_ts, long v2_bs, long v2_ts, int k2_bdiv,
                                   int nk2, int batch, int heads, int hd, int nq, int nk, float scale, int dtype, void* stream) {
  MMGT_CHECK(o_twin, "attention_twin: null o_twin");
  return attention_entry(q, q_bs0, q_bs1, q_ts, k, k_bs0, k_bs1, k_ts, v, v_bs0, v_bs1, v_ts, o, o_bs0, o_bs1, o_ts, bdiv, k2, v2, k2_bs,
                         k2_ts, v2_bs, v2_ts, k2_bdiv, nk2, 0, batch, heads, hd, nq, nk, scale, 1, nullptr, 0, 1, o_twin, dtype, stream);
}
