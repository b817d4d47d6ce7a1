// Micro-benchmark: what an MFMA gap costs on a gfx950 SIMD as a function of what is issued inside it, for ONE and for TWO waves per SIMD.
//
// Every wave runs the same stream: repeat { 1 MFMA ; E x v_exp_f32 ; F x v_fma_f32 } ("interleaved"), or the same multiset as a run of
// G MFMAs followed by the G gaps' fillers ("phased": what attn64's QK^T run + softmax block is).  All operands are independent
// registers, the instructions are volatile asm statements (hipcc neither reorders nor pads them), the result is shader cycles per MFMA per
// wave from s_memtime around the loop, median over the workgroups, plus the clock the chip held (s_memrealtime).
//
//   hipcc --offload-arch=gfx950 -O3 tools/micro/gapfill.hip -o /tmp/gapfill && /tmp/gapfill > profiles/r4/gapfill.txt
//
// The question it answers: MI355X_MICROARCH.md's cycle-constants row says a gap runs max(pipe, 8 + sum of filler issue
// costs) for one wave's stream; DESIGN section 4 "fact 1" says that with two waves per SIMD MFMA time and VALU time ADD.  Both can hold.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <vector>

typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) short s16x8;

template <int E, int F>
__device__ __forceinline__ void fillers(float (&t)[4], float (&a)[8], float k) {
#pragma unroll
  for (int e = 0; e < E; ++e) asm volatile("v_exp_f32 %0, %0" : "+v"(t[e & 3]));
#pragma unroll
  for (int f = 0; f < F; ++f) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(a[f & 7]) : "v"(k));
}

// SHAPE 0: v_mfma_f32_32x32x16_bf16 (pipe 32), 1: v_mfma_f32_16x16x32_bf16 (pipe 16).  G = 1: interleaved; G > 1: phased runs of G.
template <int SHAPE, int E, int F, int G, int THREADS>
__global__ __launch_bounds__(THREADS, THREADS / 256) void k(unsigned long long* out, float* sink, int iters) {
  extern __shared__ char pad[];   // (dynamic LDS: 96 KiB forces one workgroup per CU)
  f32x16 c[4] = {(f32x16)(0.f), (f32x16)(0.f), (f32x16)(0.f), (f32x16)(0.f)};
  f32x4 d[4] = {(f32x4)(0.f), (f32x4)(0.f), (f32x4)(0.f), (f32x4)(0.f)};
  s16x8 x = (s16x8)((short)(threadIdx.x & 7)), y = (s16x8)((short)1);
  float t[4], a[8], kk = 1.0001f;
  for (int i = 0; i < 4; ++i) t[i] = -1.f - i;
  for (int i = 0; i < 8; ++i) a[i] = threadIdx.x * 1e-3f + i;
  unsigned long long t0, t1, r0, r1;
  __syncthreads();
  asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0), "=s"(r0)::"memory");
  for (int it = 0; it < iters; ++it) {
    {
      if (G == 1) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          if (SHAPE == 0) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(c[q]) : "v"(x), "v"(y));
          else asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(d[q]) : "v"(x), "v"(y));
          fillers<E, F>(t, a, kk);
        }
      } else {
#pragma unroll
        for (int q = 0; q < G; ++q) {
          if (SHAPE == 0) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(c[q & 3]) : "v"(x), "v"(y));
          else asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(d[q & 3]) : "v"(x), "v"(y));
        }
#pragma unroll
        for (int q = 0; q < G; ++q) fillers<E, F>(t, a, kk);
      }
    }
  }
  asm volatile("s_nop 15\n\ts_nop 15\n\ts_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1), "=s"(r1)::"memory");
  float s = 0;
  for (int i = 0; i < 4; ++i) s += c[i][0] + d[i][0] + t[i];
  for (int i = 0; i < 8; ++i) s += a[i];
  if (s == 12345.678f) sink[threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) {
    out[(blockIdx.x * (THREADS / 64) + threadIdx.x / 64) * 2] = t1 - t0;
    out[(blockIdx.x * (THREADS / 64) + threadIdx.x / 64) * 2 + 1] = r1 - r0;
  }
}

template <int SHAPE, int E, int F, int G, int THREADS>
void run(unsigned long long* dout, float* sink) {
  const int iters = 4000, nb = 256, waves = nb * THREADS / 64;
  const int per_it = G == 1 ? 4 : G;   // MFMAs per loop iteration
  auto kern = k<SHAPE, E, F, G, THREADS>;
  hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
  for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(kern, dim3(nb), dim3(THREADS), 96 * 1024, 0, dout, sink, iters);
  hipDeviceSynchronize();
  std::vector<unsigned long long> h(2 * waves);
  hipMemcpy(h.data(), dout, h.size() * 8, hipMemcpyDeviceToHost);
  std::vector<double> cyc(waves), clk(waves);
  for (int i = 0; i < waves; ++i) { cyc[i] = (double)h[2 * i] / ((double)iters * per_it); clk[i] = (double)h[2 * i] / (double)h[2 * i + 1] * 100.0; }
  std::sort(cyc.begin(), cyc.end());
  std::sort(clk.begin(), clk.end());
  const int pipe = SHAPE == 0 ? 32 : 16, wps = THREADS / 256, cost = 8 + 8 * E + 4 * F;
  const int model1 = std::max(pipe, cost);                       // one wave's stream alone (the guide's row)
  const int model2 = std::max(wps * pipe, wps * cost);           // wps waves sharing pipe and issue port perfectly
  printf("%-9s %s  E=%d F=%d  waves/SIMD=%d | cycles per MFMA per wave: median %6.1f  p10 %6.1f  p90 %6.1f | model: alone max(pipe, 8+cost) = %3d, "
         "perfect sharing = %3d, sum (pipe + VALU) x waves = %3d | clock %4.0f MHz\n",
         SHAPE == 0 ? "32x32x16" : "16x16x32", G == 1 ? "interleaved" : (G == 4 ? "runs of 4  " : "runs of 12 "), E, F, wps, cyc[waves / 2], cyc[waves / 10],
         cyc[waves * 9 / 10], model1, model2, wps * (pipe + 8 * E + 4 * F), clk[waves / 2]);
  fflush(stdout);
}

template <int SHAPE, int E, int F>
void both(unsigned long long* dout, float* sink) {
  run<SHAPE, E, F, 1, 256>(dout, sink);
  run<SHAPE, E, F, 1, 512>(dout, sink);
}

int main() {
  unsigned long long* dout;
  float* sink;
  hipMalloc(&dout, 256 * 8 * 2 * 8);
  hipMalloc(&sink, 4096);
  printf("# gfx950 MFMA gap filling: one stream per wave, 256 workgroups (one per CU), 16000 MFMAs per wave\n");
  both<0, 0, 0>(dout, sink);
  both<0, 0, 2>(dout, sink);
  both<0, 0, 4>(dout, sink);
  both<0, 0, 5>(dout, sink);
  both<0, 0, 6>(dout, sink);
  both<0, 0, 8>(dout, sink);
  both<0, 0, 12>(dout, sink);
  both<0, 1, 0>(dout, sink);
  both<0, 1, 2>(dout, sink);
  both<0, 1, 4>(dout, sink);
  both<0, 2, 0>(dout, sink);
  both<0, 2, 2>(dout, sink);
  both<0, 3, 0>(dout, sink);
  both<0, 3, 2>(dout, sink);
  both<0, 4, 0>(dout, sink);
  printf("# attention's multiset per 32x32x16 gap (64 v_exp + 72 other VALU per 28 MFMAs ~ E = 2, F = 3), interleaved against phased\n");
  run<0, 2, 3, 1, 256>(dout, sink);
  run<0, 2, 3, 1, 512>(dout, sink);
  run<0, 2, 3, 4, 256>(dout, sink);
  run<0, 2, 3, 4, 512>(dout, sink);
  run<0, 2, 3, 12, 256>(dout, sink);
  run<0, 2, 3, 12, 512>(dout, sink);
  printf("# 16x16x32 (pipe 16)\n");
  both<1, 0, 0>(dout, sink);
  both<1, 0, 1>(dout, sink);
  both<1, 0, 2>(dout, sink);
  both<1, 0, 3>(dout, sink);
  both<1, 0, 4>(dout, sink);
  both<1, 1, 0>(dout, sink);
  both<1, 1, 1>(dout, sink);
  both<1, 1, 2>(dout, sink);
  run<1, 1, 1, 4, 256>(dout, sink);
  run<1, 1, 1, 4, 512>(dout, sink);
  return 0;
}
