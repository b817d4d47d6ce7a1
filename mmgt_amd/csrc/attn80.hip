// Spatial flash attention, head_dim 80 (the 32 x 32 level), bf16, K / V^T tiles staged by LDS-DMA (gfx950).
//
// attn64.hip's pipeline (attn64d_kernel: a ring of four 64-key slots filled by `buffer_load ... lds` two tiles ahead of their first reader, no
// staging registers, a counted vmcnt, one barrier per tile, fragments read one body ahead of their MFMAs, transposed scores, denominator from a
// ones row of V^T, no running maximum in the fast pass) for the next level of the UNet, where attention.hip's register-staged kernel kept
// the matrix pipe 35 % busy (768 TFLOP/s at 48 x 8 x 1024 x 2048: 22 MFMAs of a 32 x 64 block in ~2000 cycles of a SIMD).  What differs:
//   * one 32-query block per wave (o is 3 accumulator tiles: 80 rows + the ones row), four waves per workgroup = 128 queries on a ring of
//     THREE 20.5-KB slots (one tile ahead of the reader): two workgroups per CU, whose prologues, barriers and tails fall under each other's
//     tile loops.  (Eight waves on a four-slot ring, one workgroup per CU: -DATTN80_NW=8, 7 % / 11 % slower with / without the bank,
//     profiles/r6/bench_attn80_r6.txt.)
//   * head_dim 80 is five whole k-steps: no spare reduction slot carries the softmax reference, the score MFMA chain starts from an
//     accumulator of -reference instead (loop-invariant in the fast pass: 16 registers);
//   * K slot layout: per k-step a plane of 64 rows x 32 B (row i = key pi(i), the two 16-byte halves swapped in rows 8 .. 15 mod 16:
//     160-byte packed rows would put rows i and i + 8 on the same banks); V^T as in attn64.hip (128-byte rows, chunk c at c ^ ((d >> 1) & 7)).
#include <type_traits>

#include "common.h"
#include "gemm_common.h"
#include "attn_common.h"
#include "mmgt_hip.h"

namespace {

#ifndef ATTN80_NW
#define ATTN80_NW 4
#endif
constexpr int HD = 80, KT = 64, KSQ = 5, DT = 3, NW = ATTN80_NW, NT = NW * 64, NSLOT = NW == 8 ? 4 : 3, AHEAD = NSLOT - 1;   // tiles issued ahead of the one being read
constexpr int SLOT_K = KT * HD * 2, SLOT_V = HD * KT * 2, SLOT_CONST = SLOT_K + SLOT_V, SLOT_BYTES = SLOT_CONST + 64;
constexpr int NPK = SLOT_K / 1024, NPV = SLOT_V / 1024, PPW = (NPK + NPV + NW - 1) / NW;       // 10 + 10 one-KiB pieces per tile: piece ids wid + NW i
constexpr float RESCALE_LAG = 8.f;                            // see attention.hip
static_assert(NPK == 10 && NPV == 10 && NSLOT * SLOT_BYTES + 1024 <= 96 * 1024, "layout");

#define FENCE() __builtin_amdgcn_sched_barrier(0)
#define PIN(x) asm volatile("" : "+v"(x))

template <bool FAST>
__global__ __launch_bounds__(NT, 2) void attn80d_kernel(AttnParams p) {
  typedef bf16_t T;
  extern __shared__ __attribute__((aligned(1024))) char smem[];
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
  pair = p.npairs - 1 - pair;   // longest first (the batch entries behind seg2_first_batch also walk the bank)
  const int b = pair / p.heads, head = pair - b * p.heads;
  const int bo = b / p.bdiv, bi = b - bo * p.bdiv;
  const int q0 = (qblk * NW + wid) * 32;
  const T* qb_ = reinterpret_cast<const T*>(p.q) + bo * p.q_bs0 + bi * p.q_bs1 + (long)head * HD;
  T* ob = reinterpret_cast<T*>(p.o) + bo * p.o_bs0 + bi * p.o_bs1 + (long)head * HD;

  Frag<T> qf[KSQ];
  {
    const T* qrow = qb_ + (long)(q0 + lr) * p.q_ts;
#pragma unroll
    for (int ks = 0; ks < KSQ; ++ks) {
      frag_load(qf[ks], qrow + 16 * ks + 8 * lh);
      float q8[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) q8[j] = frag_get(qf[ks], j) * p.scale_log2e;
      frag_set8(qf[ks], q8);
    }
  }
  f32x16 o[DT];
#pragma unroll
  for (int i = 0; i < DT; ++i) o[i] = (f32x16)(0.f);
  float m_run = 0.f;

  const bool has2 = p.k2 != nullptr && p.nk2 > 0 && b >= p.seg2_first_batch;
  const int nt0 = p.nk / KT;
  const int ntiles = nt0 + (has2 ? p.nk2 / KT : 0);
  const T* kb0 = reinterpret_cast<const T*>(p.k) + bo * p.k_bs0 + bi * p.k_bs1 + (long)head * HD;
  const T* vb0 = reinterpret_cast<const T*>(p.v) + bo * p.v_bs0 + bi * p.v_bs1 + (long)head * HD * p.v_ts;
  const int b2 = b / p.k2_bdiv;
  const T* kb1 = has2 ? reinterpret_cast<const T*>(p.k2) + b2 * p.k2_bs + (long)head * HD : kb0;
  const T* vb1 = has2 ? reinterpret_cast<const T*>(p.v2) + b2 * p.v2_bs + (long)head * HD * p.v2_ts : vb0;
  auto uni = [](const T* ptr) {   // (wave-uniform 64-bit bases: see attn64.hip)
    const unsigned long long a = reinterpret_cast<unsigned long long>(ptr);
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)a), hi = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
    return reinterpret_cast<const T*>(((unsigned long long)hi << 32) | lo);
  };
  const T* kseg = uni(kb0);
  const T* vseg = uni(vb0);
  int kts2 = __builtin_amdgcn_readfirstlane((int)p.k_ts * 2);     // bytes per key row of the current segment

  // constants of every slot: eight ones (V^T row 80, the softmax denominator), zeros (rows 81 .. 95)
  if (tid < NSLOT * 8) {
    const int sl = tid / 8, w = tid % 8;
    *reinterpret_cast<unsigned*>(smem + sl * SLOT_BYTES + SLOT_CONST + 4 * w) = w < 4 ? 0x3F803F80u : 0u;
  }

  // ---- DMA pieces of this wave: ids wid + NW i of [K 0 .. 9 | V^T 0 .. 9]
  unsigned off[PPW];
  auto seg_offsets = [&](bool s1) {
    const long kts = s1 ? p.k2_ts : p.k_ts, vts = s1 ? p.v2_ts : p.v_ts;
    auto koff = [&](int piece) {      // granule g of the K planes: plane g / 128, row (g % 128) / 2, half (g & 1) ^ bit 3 of the row
      const int g = piece * 64 + lane, plane = g >> 7, row = (g & 127) >> 1, half = (g & 1) ^ ((row >> 3) & 1);
      const int key = (row & ~12) | ((row & 4) << 1) | ((row & 8) >> 1);
      return (unsigned)(key * kts * 2 + (16 * plane + 8 * half) * 2);
    };
    auto voff = [&](int piece) {
      const int v = piece * 64 + lane, d = v >> 3, sl = v & 7, c = sl ^ ((d >> 1) & 7);
      return (unsigned)(d * vts * 2 + c * 16);
    };
#pragma unroll
    for (int i = 0; i < PPW; ++i) {
      const int id = wid + NW * i;
      off[i] = id < NPK ? koff(id) : voff(id - NPK);
    }
  };
  seg_offsets(false);
  const int npieces = (NPK + NPV - wid + NW - 1) / NW;    // pieces this wave issues per tile
  auto issue_tile = [&](int it) {                      // the pieces of tile `it` into ring slot it % NSLOT (tiles strictly in order)
    if (it == nt0) {                                   // the bank segment starts
      kseg = uni(kb1);
      vseg = uni(vb1);
      kts2 = __builtin_amdgcn_readfirstlane((int)p.k2_ts * 2);
      seg_offsets(true);
    }
    const int kt = it >= nt0 ? it - nt0 : it;
    char* slot = smem + (it % NSLOT) * SLOT_BYTES;
    const int sK = kt * KT * kts2, sV = kt * KT * 2;
    const __amdgpu_buffer_rsrc_t rK = dma_rsrc(kseg), rV = dma_rsrc(vseg);
#pragma unroll
    for (int i = 0; i < PPW; ++i) {
      const int id = wid + NW * i;                     // (wave-uniform)
      if (id < NPK) blds16(rK, off[i], sK, slot + id * 1024);
      else if (id < NPK + NPV) blds16(rV, off[i], sV, slot + SLOT_K + (id - NPK) * 1024);
    }
  };
  // all but the pieces of the youngest `tiles` tiles this wave issued have landed
  auto wait_tiles = [&](int tiles) {
    const int n = tiles * npieces;                     // 0, 2, 3, 4, 5, 6
    if (n >= 6) wait_vmcnt<6>(); else if (n == 5) wait_vmcnt<5>(); else if (n == 4) wait_vmcnt<4>(); else if (n == 3) wait_vmcnt<3>();
    else if (n == 2) wait_vmcnt<2>(); else wait_vmcnt<0>();
  };

  // ---- fragment read offsets inside a slot
  int kofs[KSQ];                                        // K rows of half tile 0 (half tile 1: + 32 rows = 1024 bytes)
#pragma unroll
  for (int ks = 0; ks < KSQ; ++ks) kofs[ks] = ks * 2048 + lr * 32 + ((lh ^ ((lr >> 3) & 1)) << 4);
  int vofs[2][2][DT];                                   // [half tile][s2][dt]
#pragma unroll
  for (int hf = 0; hf < 2; ++hf)
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
      for (int dt = 0; dt < DT; ++dt) {
        const int d = 32 * dt + lr, c = 4 * hf + 2 * s2 + lh;
        vofs[hf][s2][dt] = d < HD ? SLOT_K + d * 128 + ((c ^ ((d >> 1) & 7)) << 4) : d == HD ? SLOT_CONST : SLOT_CONST + 16;
      }

  auto store_out = [&](T* base) __attribute__((always_inline)) {
    constexpr int R = HD % 32, REG = (R & 3) + 4 * (R >> 3), LHS = (R >> 2) & 1;
    const float mine = o[HD / 32][REG], other = __shfl_xor(mine, 32);
    const float inv = 1.f / (lh == LHS ? mine : other);
    T* orow = base + (long)(q0 + lr) * p.o_ts;
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int d = dt * 32 + 8 * g + 4 * lh;
        if (d < HD) {
          union { bf16_t e[4]; u32x2 u; } pk4;
#pragma unroll
          for (int e = 0; e < 4; ++e) pk4.e[e] = f32_to_bf16(o[dt][4 * g + e] * inv);
          *reinterpret_cast<u32x2*>(orow + d) = pk4.u;
        }
      }
  };

  auto tile_max = [&](const f32x16& sn) __attribute__((always_inline)) {
    float m1 = fmaxf(fmaxf(sn[0], sn[1]), sn[2]);
#pragma unroll
    for (int r = 3; r < 15; r += 2) m1 = fmaxf(fmaxf(m1, sn[r]), sn[r + 1]);
    m1 = fmaxf(m1, sn[15]);
    const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(m1), __float_as_uint(m1), false, false);
    return fmaxf(__uint_as_float(sw[0]), __uint_as_float(sw[1]));
  };
  // the reference moves by delta: the accumulators and the scores in flight follow
  auto rescale = [&](f32x16& sn, float delta) __attribute__((always_inline)) {
    const float alpha = __builtin_amdgcn_exp2f(-delta);
    m_run += delta;
#pragma unroll
    for (int i = 0; i < DT; ++i) o[i] *= alpha;
    sn -= delta;
  };
  auto kfrag = [&](Frag<T>& f, const char* slot, int hf, int ks) __attribute__((always_inline)) {
    frag_load(f, reinterpret_cast<const T*>(slot + kofs[ks] + hf * 1024));
  };

  // One body = the scores of the NEXT 32-key half tile (sn) beside the exponentials and P.V of the current one (sp); the fragments a body
  // multiplies with were read by its predecessor, it reads its successor's (see attn64.hip).
  union VF { u32x4 u; Frag<T> f; };
  auto body = [&](auto fastc, f32x16& sp, f32x16& sn, Frag<T> (&kf)[KSQ], VF (&vf0)[DT], Frag<T> (&kfn)[KSQ], VF (&vf0n)[DT], const char* vslot, int vhf,
                  const char* kslot_n, int khf_n, const char* vslot_n, int vhf_n, bool decide) __attribute__((always_inline)) {
    constexpr bool fast = decltype(fastc)::value;
    VF vf1[DT];
    float ex[16];
    Frag<T> pf[2];
    auto exps = [&](int e0, int e1, int s2) __attribute__((always_inline)) {
#pragma unroll
      for (int e = e0; e < e1; ++e) PIN(sp[8 * s2 + e]);
#pragma unroll
      for (int e = e0; e < e1; ++e) ex[8 * s2 + e] = __builtin_amdgcn_exp2f(sp[8 * s2 + e]);
#pragma unroll
      for (int e = e0; e < e1; ++e) {
        if (e == 7) {
          float p8[8];
#pragma unroll
          for (int j = 0; j < 8; ++j) p8[j] = ex[8 * s2 + j];
          frag_set8(pf[s2], p8);
          PIN(pf[s2].v);
        } else {
          PIN(ex[8 * s2 + e]);
        }
      }
    };
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) vf1[dt].u = *reinterpret_cast<const u32x4*>(vslot + vofs[vhf][1][dt]);
    FENCE();
    const f32x16 c0 = (f32x16)(-m_run);                  // (fast pass: loop-invariant)
#pragma unroll
    for (int c = 0; c < KSQ; ++c) {
      PIN(kf[c].v);
      if (c == 0) sn = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[0].v, qf[0].v, c0, 0, 0, 0);
      else mma32(sn, kf[c], qf[c]);
      PIN(sn);
      exps((8 * c + 2) / KSQ, (8 * (c + 1) + 2) / KSQ, 0);
      FENCE();
    }
    // the successor's K fragments (this body's are consumed)
#pragma unroll
    for (int ks = 0; ks < KSQ; ++ks) kfrag(kfn[ks], kslot_n, khf_n, ks);
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) {
      mma32(o[dt], vf0[dt].f, pf[0]);
      PIN(o[dt]);
      exps((8 * dt + 1) / DT, (8 * (dt + 1) + 1) / DT, 1);
      FENCE();
    }
    // ... and its first V^T fragments
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) vf0n[dt].u = *reinterpret_cast<const u32x4*>(vslot_n + vofs[vhf_n][0][dt]);
    float mt = 0.f;
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) {
      mma32(o[dt], vf1[dt].f, pf[1]);
      PIN(o[dt]);
      if (!fast && dt == 0) {
        PIN(sn);
        mt = tile_max(sn);
        PIN(mt);
      }
      FENCE();
    }
    if constexpr (!fast) {
      if (decide && __any(mt > RESCALE_LAG)) rescale(sn, fmaxf(mt, 0.f));
    }
  };

  // One pass over the key tiles: prologue (the first AHEAD tiles on their way; the scores of half tile 0 set the softmax reference) + the tile loop.
  auto attend = [&](auto fastc) __attribute__((always_inline)) {
#pragma unroll
    for (int t = 0; t < AHEAD; ++t)
      if (t < ntiles) issue_tile(t);
    wait_tiles(min(ntiles, AHEAD) - 1);                  // tile 0 has landed
    __syncthreads();
    f32x16 sA, sB;
    {
      Frag<T> kf[KSQ];
#pragma unroll
      for (int ks = 0; ks < KSQ; ++ks) kfrag(kf[ks], smem, 0, ks);
      sA = (f32x16)(0.f);
#pragma unroll
      for (int ks = 0; ks < KSQ; ++ks) mma32(sA, kf[ks], qf[ks]);
      m_run = 0.f;
      rescale(sA, tile_max(sA));                         // (o is zero: only the reference and sA move)
    }
    Frag<T> kfA[KSQ], kfB[KSQ];
    VF vfA[DT], vfB[DT];
#pragma unroll
    for (int ks = 0; ks < KSQ; ++ks) kfrag(kfA[ks], smem, 1, ks);                                        // S(0, 1)
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) vfA[dt].u = *reinterpret_cast<const u32x4*>(smem + vofs[0][0][dt]);   // P.V(0, 0)
    for (int it = 0; it < ntiles; ++it) {
      if (it > 0) {
        // tile it + 1 has landed -- this wave's pieces: only those of the tiles behind it (issued up to it + AHEAD - 1) may still be in flight --
        // and, behind the barrier, everybody's; the barrier also frees slot (it + AHEAD) % NSLOT = (it - 1) % NSLOT, last read in iteration it - 1
        wait_tiles(max(min(ntiles - 1, it + AHEAD - 1) - (it + 1), 0));
        __builtin_amdgcn_s_waitcnt(0xC07F);              // lgkmcnt(0): this wave's fragment reads have returned
        __builtin_amdgcn_s_barrier();
      }
      if (it + AHEAD < ntiles) issue_tile(it + AHEAD);
      const char* cur = smem + (it % NSLOT) * SLOT_BYTES;
      const char* nxt = smem + ((it + 1) % NSLOT) * SLOT_BYTES;
      // S(it, 1) || P.V(it, 0); reads ahead: K(it + 1, 0), V(it, 1)
      body(fastc, sA, sB, kfA, vfA, kfB, vfB, cur, 0, nxt, 0, cur, 1, true);
      // S(it + 1, 0) || P.V(it, 1); reads ahead: K(it + 1, 1), V(it + 1, 0)
      body(fastc, sB, sA, kfB, vfB, kfA, vfA, cur, 1, nxt, 1, nxt, 0, it + 1 < ntiles);
    }
  };
  if constexpr (!FAST) {
    attend(std::false_type{});
  } else {
    if (tid == 0) redo_flag = 0;                           // (published by the first barrier of the pass)
    attend(std::true_type{});
    // the guard: a denominator beyond 2^100 (or inf / NaN: an exponent overflowed) in any row of the workgroup -> the pass with the running maximum
    constexpr int R = HD % 32, REG = (R & 3) + 4 * (R >> 3);
    const bool bad = !(fabsf(o[HD / 32][REG]) < 1.2676506e30f);   // (lanes of the half that holds no denominator read a zero row: finite too)
    wait_vmcnt<0>();
    if (__any(bad) && lane == 0) redo_flag = 1;
    __syncthreads();
    if (redo_flag) {                                       // uniform over the workgroup: every wave has left the tile loop, no DMA is in flight
#pragma unroll
      for (int i = 0; i < DT; ++i) o[i] = (f32x16)(0.f);
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

int g_attn80 = 1;   // mmgt_tune("attn80", 0 / 1): head_dim 80 on this kernel (A/B switch against attention.hip's attn_kernel)

}  // namespace

void mmgt_attn_set_attn80(int v) { g_attn80 = v; }
int mmgt_attn_get_nomax();

// attention.hip's dispatcher: bf16, head_dim 80, V transposed, nq % 256 == 0 (a multiple of the 32 NW queries of a workgroup), nk % 64 == 0, nk2 % 64 == 0, no output scale, no twin output.
// Returns -1 when the kernel is switched off (the caller then takes attn_kernel).
int mmgt_attn80_launch(const void* params, int batch, int heads, void* stream) {
  if (!g_attn80) return -1;
  AttnParams p = *reinterpret_cast<const AttnParams*>(params);
  p.heads = heads;
  p.npairs = batch * heads;
  p.nqb = p.nq / (32 * NW);
  const size_t lds = (size_t)NSLOT * SLOT_BYTES + 1024;
  static bool ready[16][2] = {};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) { mmgt_set_error("attention: device query failed"); return 2; }
  const int fast = mmgt_attn_get_nomax() ? 1 : 0;
  const void* kern = fast ? reinterpret_cast<const void*>(attn80d_kernel<true>) : reinterpret_cast<const void*>(attn80d_kernel<false>);
  if (!ready[dev][fast]) {
    if (hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
      mmgt_set_error("attention: cannot reserve %d bytes of LDS", (int)lds);
      return 2;
    }
    ready[dev][fast] = true;
  }
  const dim3 grid((unsigned)((long)p.nqb * batch * heads));
  if (fast) hipLaunchKernelGGL(attn80d_kernel<true>, grid, dim3(NT), lds, (hipStream_t)stream, p);
  else hipLaunchKernelGGL(attn80d_kernel<false>, grid, dim3(NT), lds, (hipStream_t)stream, p);
  MMGT_LAUNCH_CHECK();
  return 0;
}
