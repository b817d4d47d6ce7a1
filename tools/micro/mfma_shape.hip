// Micro-benchmark for VERDICT r4 item 2 (MI355X_MICROARCH.md, DVFS give-back item 7): does a bf16 MFMA loop on RANDOM data hold a different
// clock -- and deliver different FLOP/s -- in the 16x16x32 shape than in the 32x32x16 shape on this box?  Bare loops, operands in registers,
// one wave per SIMD on every CU (96 KiB of dynamic LDS per workgroup: one workgroup per CU), the same FLOP per launch for both shapes, 2 s of
// back-to-back launches per shape before the stamped one; in-kernel clock = delta s_memtime / delta s_memrealtime x 100 MHz (median over the
// workgroups), rate from HIP events over the last batch.  A second pair of loops adds what attn64d / ff_fused / rowgemm do beside their MFMAs:
// one ds_read_b128 per MFMA-equivalent of 32 768 FLOP (the LDS holds random data).
//
//   hipcc --offload-arch=gfx950 -O3 tools/micro/mfma_shape.hip -o /tmp/mfma_shape && /tmp/mfma_shape > profiles/r5/mfma_shape_ab.txt
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <vector>

typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) short s16x8;

__device__ __forceinline__ short rnd_bf16(unsigned& h) {
  h = h * 1664525u + 1013904223u;
  return (short)(((h >> 16) & 0x8000u) | 0x3F00u | ((h >> 9) & 0xFFu));       // +-[0.5, 2)
}

// SHAPE 16: 16 accumulator tiles of 16x16 (64 registers), 16 MFMAs of 16x16x32 per iteration; SHAPE 32: 4 tiles of 32x32 (64 registers), 8 MFMAs of
// 32x32x16 per iteration -- 262 144 FLOP per iteration and wave either way.  LDS = 1: the A operands are re-read from LDS every iteration.
template <int SHAPE, int LDS>
__global__ __launch_bounds__(256) void k(unsigned long long* __restrict__ stamps, float* __restrict__ sink, int iters, unsigned seed) {
  extern __shared__ __attribute__((aligned(1024))) char smem[];
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  unsigned h = seed ^ (blockIdx.x * 2654435761u) ^ (threadIdx.x * 40503u);
  s16x8 a[4], b[4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) { a[i][j] = rnd_bf16(h); b[i][j] = rnd_bf16(h); }
  if (LDS) {
    for (int i = threadIdx.x; i < 16384; i += 256) {                          // 64 KiB of random bf16 pairs
      unsigned v = (unsigned)(unsigned short)rnd_bf16(h) | ((unsigned)(unsigned short)rnd_bf16(h) << 16);
      reinterpret_cast<unsigned*>(smem)[i] = v;
    }
    __syncthreads();
  }
  const char* rd = smem + wid * 16384 + lane * 16;
  unsigned long long c0, c1, r0, r1;
  float sum = 0.f;
  if constexpr (SHAPE == 16) {
    f32x4 acc[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = (f32x4)(0.f);
    asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(c0), "=s"(r0)::"memory");
    for (int it = 0; it < iters; ++it) {
      if (LDS) {
#pragma unroll
        for (int i = 0; i < 4; ++i) a[i] = *reinterpret_cast<const s16x8*>(rd + ((it & 3) * 4 + i) * 1024);
#pragma unroll
        for (int i = 0; i < 4; ++i) b[i] = *reinterpret_cast<const s16x8*>(rd + 8192 + ((it & 1) * 4 + i) * 1024);
      }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)   // (tied accumulation-register operand: through the builtin hipcc rotates the tiles through other AGPRs with
                                      //  v_accvgpr_mov copies and s_nops -- 27 cycles per MFMA instead of 16: tools/micro/lonewave.hip)
          asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc[4 * i + j]) : "v"(a[i]), "v"(b[j]));
    }
    asm volatile("s_nop 15\n\ts_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(c1), "=s"(r1)::"memory");
#pragma unroll
    for (int i = 0; i < 16; ++i) sum += acc[i][0] + acc[i][3];
  } else {
    f32x16 acc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[i] = (f32x16)(0.f);
    asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(c0), "=s"(r0)::"memory");
    for (int it = 0; it < iters; ++it) {
      if (LDS) {
#pragma unroll
        for (int i = 0; i < 4; ++i) a[i] = *reinterpret_cast<const s16x8*>(rd + ((it & 3) * 4 + i) * 1024);
#pragma unroll
        for (int i = 0; i < 4; ++i) b[i] = *reinterpret_cast<const s16x8*>(rd + 8192 + ((it & 1) * 4 + i) * 1024);
      }
      // 2 x 2 tiles, two k-steps of 16: a[0], a[1] / a[2], a[3] are the two k-steps of the two row tiles
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc[2 * i + j]) : "v"(a[2 * i + ks]), "v"(b[2 * j + ks]));
    }
    asm volatile("s_nop 15\n\ts_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(c1), "=s"(r1)::"memory");
#pragma unroll
    for (int i = 0; i < 4; ++i) sum += acc[i][0] + acc[i][15];
  }
  if (sum == 12345.678f) sink[threadIdx.x] = sum;
  if (threadIdx.x == 0) {
    stamps[2 * blockIdx.x] = c1 - c0;
    stamps[2 * blockIdx.x + 1] = r1 - r0;
  }
}

template <int SHAPE, int LDS>
void run(const char* what, unsigned long long* d_st, float* d_sink, int nwg) {
  constexpr int LDSB = 96 * 1024;
  const int iters = 20000;
  auto kern = k<SHAPE, LDS>;
  hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, LDSB);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  float spent = 0.f, last = 0.f;
  unsigned seed = 1;
  do {
    hipEventRecord(e0, 0);
    for (int q = 0; q < 8; ++q) hipLaunchKernelGGL(kern, dim3(nwg), dim3(256), LDSB, 0, d_st, d_sink, iters, seed++);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    hipEventElapsedTime(&last, e0, e1);
    spent += last;
  } while (spent < 2000.f);
  std::vector<unsigned long long> st(2 * nwg);
  hipMemcpy(st.data(), d_st, sizeof(unsigned long long) * 2 * nwg, hipMemcpyDeviceToHost);
  std::vector<double> mhz, cyc;
  for (int i = 0; i < nwg; ++i)
    if (st[2 * i + 1]) { mhz.push_back((double)st[2 * i] / (double)st[2 * i + 1] * 100.0); cyc.push_back((double)st[2 * i]); }
  std::sort(mhz.begin(), mhz.end());
  std::sort(cyc.begin(), cyc.end());
  const double flop = 8.0 * nwg * 4.0 * iters * 262144.0;
  printf("%-44s  %8.1f us/launch  %7.1f TFLOP/s  in-kernel clock %6.0f MHz (min %5.0f max %5.0f)  %6.2f cycles per 32768 FLOP\n", what,
         last / 8 * 1e3, flop / (last * 1e-3) / 1e12, mhz[mhz.size() / 2], mhz.front(), mhz.back(), cyc[cyc.size() / 2] / (iters * 8.0));
}

int main() {
  hipDeviceProp_t prop;
  hipGetDeviceProperties(&prop, 0);
  const int nwg = prop.multiProcessorCount;
  unsigned long long* d_st;
  float* d_sink;
  hipMalloc(&d_st, sizeof(unsigned long long) * 2 * nwg);
  hipMalloc(&d_sink, 1024);
  printf("# %s, %d CUs; one workgroup of 4 waves per CU, random bf16 operands, 2 s of launches per row\n", prop.name, nwg);
  for (int rep = 0; rep < 2; ++rep) {
    run<16, 0>("16x16x32, operands in registers", d_st, d_sink, nwg);
    run<32, 0>("32x32x16, operands in registers", d_st, d_sink, nwg);
    run<16, 1>("16x16x32, 8 ds_read_b128 per 16 MFMAs", d_st, d_sink, nwg);
    run<32, 1>("32x32x16, 8 ds_read_b128 per 8 MFMAs", d_st, d_sink, nwg);
  }
  return 0;
}
