// Round-3 producer / consumer fused FeedForward kernel (ffn_ver = 3), cut out of mmgt_amd/csrc/ffn.hip in round 5: it measured 503-537 us
// against 469-510 for the single-role kernel that ships.  Record only: it compiled inside ffn.hip of commit d753722 (helpers, constants and the
// launcher branch live there).

// DBG (mmgt_tune("ffn_dbg", v), measurements only): 1 = every weight piece takes the poison offset (nothing is fetched: the
// compute streams alone), 2 = no MFMA / GELU (the weight stream alone), 3 = no GELU, 4 = no ff2 MFMAs; results are garbage.
template <int DBG>
__global__ __launch_bounds__(512, 2)
void ff_fused_kernel(const bf16_t* __restrict__ x, long ldx, const float* __restrict__ gamma, const float* __restrict__ beta,
                     float eps, const char* __restrict__ wimg, int nsb, const float* __restrict__ bias2,
                     const bf16_t* __restrict__ res, long ldr, bf16_t* __restrict__ out, long ldo, int M, unsigned long long* trace) {
  extern __shared__ __attribute__((aligned(1024))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31, hh = lane >> 5;
  const int rg = wid & 3;                         // row group: waves rg (role A) and rg + 4 (role B) share a SIMD
  const long row = (long)blockIdx.x * 128 + rg * 32 + r;
  const long rowc = row < M ? row : M - 1;
  int trace_n = 0;
  auto stamp = [&]() {   // debug (tools/trace_ffn.py): shader-clock stamps of waves 0 (A) and 4 (B) of every workgroup
    if (trace && rg == 0 && lane == 0 && trace_n < 32) trace[((long)blockIdx.x * 2 + (wid >> 2)) * 32 + trace_n++] = __builtin_amdgcn_s_memtime();
  };
  stamp();

  // ---- gamma | beta | bias2 and the ff1 biases of all sub-blocks -> LDS (all 512 threads)
  {
    float* lgb = reinterpret_cast<float*>(smem + FF_LG);
    if (tid < 3 * FFC / 4 && (gamma || tid >= 2 * FFC / 4)) {   // 3 x 80 vectors
      const float* src = tid < FFC / 4 ? gamma + 4 * tid : tid < 2 * FFC / 4 ? beta + 4 * (tid - FFC / 4) : bias2 + 4 * (tid - 2 * FFC / 4);
      *reinterpret_cast<f32x4*>(lgb + 4 * tid) = *reinterpret_cast<const f32x4*>(src);
    }
    for (int v = tid; v < nsb * 16; v += 512)                    // 16 vectors of 4 biases per sub-block, from the image
      *reinterpret_cast<f32x4*>(smem + FF_LB + v * 16) = *reinterpret_cast<const f32x4*>(wimg + (long)(v >> 4) * FF_IMG + FF_B1 + (v & 15) * 16);
  }
  using std::integral_constant;
  constexpr integral_constant<bool, true> T{};
  constexpr integral_constant<bool, false> F{};
  constexpr int PF = 3;                  // fragment reads run PF steps ahead of their MFMAs; sched_barriers pin that order (left alone,
                                         // hipcc reads right in front of each MFMA and waits lgkmcnt(0) every step)
  if (wid < 4) {
    // =========================================================================================== role A: ff1 + GEGLU
    s16x8 xf[FF_KS];                     // the 32 rows as ff1 B fragments: lane (r, hh) holds channels 16 ks + 8 hh .. + 7 of row r
    {
      const bf16_t* xr = x + rowc * ldx + 8 * hh;
#pragma unroll
      for (int ks = 0; ks < FF_KS; ++ks) xf[ks] = *reinterpret_cast<const s16x8*>(xr + 16 * ks);
    }
    __syncthreads();                     // (tables in LDS)
    if (gamma) layernorm_fragments(xf, reinterpret_cast<const float*>(smem + FF_LG), hh, eps);
    stamp();
    s16x8 fr[PF + 1][2];
    // Iteration i of role A:  S(i) | GEGLU(i - 1) -> packed G tile -> LDS slot (i - 1) & 1, one VALU-only block | M(i) | ff1(i), one
    // MFMA-dense block (two fragment reads and two waits per MFMA pair, nothing else).  Meanwhile role B:  S(i) | ff2(i - 2), 20 dense
    // MFMAs | M(i) | the 15 LDS-DMA pieces of the next weights.  The matrix pipe of the SIMD is handed back and forth: B's MFMAs
    // run under A's GELUs, B's DMA issue (~90 cycles a piece) under A's MFMAs.  Interleaving GELU and MFMAs inside A instead
    // (~10 instructions between MFMAs) stretched every MFMA gap of A to ~50 cycles and B's MFMAs came on top (in-kernel stamps:
    // 2670 ticks per iteration against 1920 of matrix pipe), whatever the instruction count of the GELU was.
    auto glu = [&](int sb, const f32x16& hp, const f32x16& gp) {        // GEGLU of sub-block sb's (hp, gp) -> G slot sb & 1
      char* gs = smem + FF_LGT + (sb & 1) * FF_GSLOT + rg * 2048 + lane * 16;
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        float gv[8];
#pragma unroll
        for (int j = 0; j < 8; j += 2) {
          float p0, p1;
          gelu_poly2(gp[8 * s + j], gp[8 * s + j + 1], p0, p1);
          gelu_finish2(gp[8 * s + j], gp[8 * s + j + 1], p0, p1, hp[8 * s + j], hp[8 * s + j + 1], gv[j], gv[j + 1]);
        }
        *reinterpret_cast<s16x8*>(gs + s * 1024) = pack8(gv);
      }
    };
    auto ff1 = [&](int sb, f32x16& hn, f32x16& gn) {                     // ff1 of sub-block sb from W1 slot sb & 1, bias first
      const char* s1 = smem + FF_L1 + (sb & 1) * FF_W1 + lane * 16;
      const float* bl = reinterpret_cast<const float*>(smem + FF_LB + sb * 256) + 4 * hh;   // register i <-> hidden 4 hh + (i & 3) + 8 (i >> 2)
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) {
        const f32x4 bh = *reinterpret_cast<const f32x4*>(bl + 8 * g4), bg = *reinterpret_cast<const f32x4*>(bl + 32 + 8 * g4);
#pragma unroll
        for (int e = 0; e < 4; ++e) { hn[4 * g4 + e] = bh[e]; gn[4 * g4 + e] = bg[e]; }
      }
#pragma unroll
      for (int i = 0; i < PF; ++i) {
        fr[i][0] = *reinterpret_cast<const s16x8*>(s1 + (2 * i) * 1024);
        fr[i][1] = *reinterpret_cast<const s16x8*>(s1 + (2 * i + 1) * 1024);
      }
#pragma unroll
      for (int ks = 0; ks < FF_KS; ++ks) {
        if (ks + PF < FF_KS) {
          fr[(ks + PF) % (PF + 1)][0] = *reinterpret_cast<const s16x8*>(s1 + (2 * (ks + PF)) * 1024);
          fr[(ks + PF) % (PF + 1)][1] = *reinterpret_cast<const s16x8*>(s1 + (2 * (ks + PF) + 1) * 1024);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (DBG != 2) {
          hn = mma32b(fr[ks % (PF + 1)][0], xf[ks], hn);
          gn = mma32b(fr[ks % (PF + 1)][1], xf[ks], gn);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    };
    f32x16 hA, gA, hB, gB;
    lds_barrier();                                              // S(0)
    stamp();
    lds_barrier();                                              // M(0): W1(0) has landed
    ff1(0, hA, gA);
    auto iter = [&](int i, f32x16& hp, f32x16& gp, f32x16& hn, f32x16& gn) {   // 1 <= i < nsb
      lds_barrier();                                            // S(i)
      if (i < 6) stamp();
      glu(i - 1, hp, gp);
      if (i < 6) stamp();
      lds_barrier();                                            // M(i): W1(i) has landed
      ff1(i, hn, gn);
      if (i < 6) stamp();
    };
    int i = 1;
    for (; i + 1 < nsb; i += 2) {
      iter(i, hA, gA, hB, gB);
      iter(i + 1, hB, gB, hA, gA);
    }
    stamp();
    if (i < nsb) {                                              // nsb even: one more full iteration, the pending tile ends in (hB, gB)
      iter(i, hA, gA, hB, gB);
      lds_barrier();                                            // S(nsb)
      glu(nsb - 1, hB, gB);
    } else {
      lds_barrier();                                            // S(nsb)
      glu(nsb - 1, hA, gA);
    }
    lds_barrier();                                              // M(nsb)
    lds_barrier();                                              // S(nsb + 1): B's last ff2 follows
    stamp();
  } else {
    // =========================================================================================== role B: weight DMA, ff2, epilogue
    const int bw = wid - 4;
    const __amdgpu_buffer_rsrc_t rw = dma_rsrc(wimg);
    const unsigned lane16 = DBG == 1 ? DMA_POISON : (unsigned)lane * 16u;
    // this wave's pieces bw, bw + 4, ... of the ff1 part (40 pieces -> W1 slot) / ff2 part (20 pieces -> W2 slot) of sub-block sb
    auto issue1 = [&](int sb, int i) { blds16(rw, lane16, sb * FF_IMG + (bw + 4 * i) * 1024, smem + FF_L1 + (sb & 1) * FF_W1 + (bw + 4 * i) * 1024); };
    auto issue2 = [&](int sb, int i) { blds16(rw, lane16, sb * FF_IMG + FF_W1 + (bw + 4 * i) * 1024, smem + FF_L2 + (sb & 1) * FF_W2 + (bw + 4 * i) * 1024); };
    __syncthreads();                     // (tables in LDS: the plain loads above are done before the first DMA goes out)
#pragma unroll
    for (int i = 0; i < 10; ++i) issue1(0, i);
    f32x16 oacc[FF_NU];
#pragma unroll
    for (int u = 0; u < FF_NU; ++u) oacc[u] = (f32x16)(0.f);
    // ff2 of sub-block sb from G slot sb & 1 and W2 slot sb & 1: 20 dense MFMAs, two channel tiles per step
    auto ff2 = [&](int sb) {
      const char* s2 = smem + FF_L2 + (sb & 1) * FF_W2 + lane * 16;
      const char* gs = smem + FF_LGT + (sb & 1) * FF_GSLOT + rg * 2048 + lane * 16;
      s16x8 fb[3][4], gb[2];
      auto rd = [&](int st, s16x8 (&f)[4]) {
#pragma unroll
        for (int q = 0; q < 4; ++q) f[q] = *reinterpret_cast<const s16x8*>(s2 + (4 * st + q) * 1024);   // (tile 2 st + (q >> 1), k-step q & 1)
      };
      gb[0] = *reinterpret_cast<const s16x8*>(gs);
      gb[1] = *reinterpret_cast<const s16x8*>(gs + 1024);
      rd(0, fb[0]);
      rd(1, fb[1]);
#pragma unroll
      for (int st = 0; st < FF_NU / 2; ++st) {
        if (st + 2 < FF_NU / 2) rd(st + 2, fb[(st + 2) % 3]);
        __builtin_amdgcn_sched_barrier(0);
        if (DBG != 2 && DBG != 4) {
          // A wave that presents an MFMA to a busy matrix pipe blocks the SIMD's vector issue -- its partner's VALU included (tools/micro/
          // coexec.hip: a VALU-only wave beside an MFMA-only wave takes the SUM of their times; with ~24 cycles of s_nop behind each MFMA the
          // VALU wave disappears under the MFMA wave).  These 20 MFMAs run beside the partner's GELU block: pace them at the pipe's rate.
          const int u = 2 * st;
          oacc[u] = mma32b(fb[st % 3][0], gb[0], oacc[u]); mfma_pace();
          oacc[u + 1] = mma32b(fb[st % 3][2], gb[0], oacc[u + 1]); mfma_pace();
          oacc[u] = mma32b(fb[st % 3][1], gb[1], oacc[u]); mfma_pace();
          oacc[u + 1] = mma32b(fb[st % 3][3], gb[1], oacc[u + 1]); mfma_pace();
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    };
    // second half of iteration i: W2(i - 1) first (needed at S(i + 1)), then W1(i + 1) (needed at M(i + 1)), wait for the former
    auto dma = [&](int i) {
      const bool dv = i - 1 >= 0 && i - 1 < nsb, dw = i + 1 < nsb;
      if (dv) {
#pragma unroll
        for (int q = 0; q < 5; ++q) issue2(i - 1, q);
      }
      if (dw) {
#pragma unroll
        for (int q = 0; q < 10; ++q) issue1(i + 1, q);
        wait_vmcnt<10>();
      } else {
        wait_vmcnt<0>();
      }
    };
    stamp();
    lds_barrier();                                              // S(0)
    for (int i = 0; i <= nsb; ++i) {
      if (i >= 2 && i < 8) stamp();
      if (i >= 2) ff2(i - 2);
      if (i >= 2 && i < 8) stamp();
      wait_vmcnt<0>();                                          // W1(i) (issued one iteration ago) has landed
      lds_barrier();                                            // M(i)
      if (i >= 2 && i < 8) stamp();
      dma(i);
      if (i >= 2 && i < 8) stamp();
      lds_barrier();                                            // S(i + 1)
    }
    ff2(nsb - 1);
    stamp();
    // ---- epilogue: + b2 + residual, bf16, 16-byte stores.  Register group k (registers 4 k .. 4 k + 3) of tile u is channels
    // 32 u + 8 k + 4 hh + (0..3); v_permlane32_swap of groups (k, k + 1) gives lane hh = 0 channels 32 u + 8 k .. + 7 and lane hh = 1
    // channels 32 u + 8 k + 8 .. + 15 (cdna_hip_programming.md T21).  The stores go through a buffer resource sized to the M valid rows:
    // rows beyond M are dropped by the range check instead of a branch per store.
    {
      const bf16_t* rr = res + rowc * ldr + 8 * hh;
      const __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc(out, 0, (int)((long)M * ldo * 2), 0x00020000);
      const unsigned obase = (unsigned)(row * ldo + 8 * hh) * 2u;          // (rows >= M: beyond num_records -> dropped)
      const float* lb2 = reinterpret_cast<const float*>(smem + FF_LG) + 2 * FFC + 8 * hh;
#pragma unroll
      for (int half = 0; half < 2; ++half) {                               // the residual vectors of 5 tiles at a time (40 registers)
        u32x4 rv[FF_NU];
#pragma unroll
        for (int q = 0; q < FF_NU; ++q) rv[q] = *reinterpret_cast<const u32x4*>(rr + 16 * (FF_NU * half + q));   // channels 16 q' + 8 hh .. + 7
#pragma unroll
        for (int uu = 0; uu < FF_NU / 2; ++uu)
#pragma unroll
          for (int k = 0; k < 4; k += 2) {
            const int u = (FF_NU / 2) * half + uu;
            const int c = 32 * u + 8 * k;          // + 8 hh in the bases
            union { u32x4 q; bf16_t e[8]; } r8;
            r8.q = rv[2 * uu + k / 2];
            const f32x4 b0 = *reinterpret_cast<const f32x4*>(lb2 + c), b1 = *reinterpret_cast<const f32x4*>(lb2 + c + 4);
            float o8[8];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(oacc[u][4 * k + e]), __float_as_uint(oacc[u][4 * k + 4 + e]), false, false);
              o8[e] = __uint_as_float(sw[0]) + b0[e] + bf16_to_f32(r8.e[e]);
              o8[4 + e] = __uint_as_float(sw[1]) + b1[e] + bf16_to_f32(r8.e[4 + e]);
            }
            const u32x4 pk = (u32x4){pack_bf16x2(o8[0], o8[1]), pack_bf16x2(o8[2], o8[3]), pack_bf16x2(o8[4], o8[5]), pack_bf16x2(o8[6], o8[7])};
            __builtin_amdgcn_raw_buffer_store_b128(pk, ro, (int)(obase + 2u * c), 0, 0);
          }
      }
    }
    stamp();
  }
}


// ---- helpers the kernel used
#ifndef MMGT_FFN_PACE
#define MMGT_FFN_PACE 0
#endif
__device__ __forceinline__ void mfma_pace() {     // ~24 cycles in which this wave asks nothing of the vector issue port
  __builtin_amdgcn_sched_barrier(0);
  if (MMGT_FFN_PACE == 1) asm volatile("s_nop 7");
  if (MMGT_FFN_PACE == 2) asm volatile("s_nop 7\n\ts_nop 1");
  if (MMGT_FFN_PACE == 3) asm volatile("s_nop 7\n\ts_nop 7");
  if (MMGT_FFN_PACE == 4) asm volatile("s_sleep 1");
  __builtin_amdgcn_sched_barrier(0);
}

__device__ __forceinline__ void lds_barrier() {   // this wave's LDS traffic has completed, then the workgroup barrier (LDS-DMA is NOT
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // drained here: the loader waves wait vmcnt themselves)
  __builtin_amdgcn_s_barrier();
}

