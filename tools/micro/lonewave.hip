// Micro-benchmark: can ONE wave per SIMD keep the gfx950 matrix pipe busy with v_mfma_f32_16x16x32_bf16 while it also issues the
// fragment reads, LDS-DMA pieces and the barrier of a 256 x 256 x 64 GEMM chunk (128 MFMAs, 32 ds_read_b128, 8 DMA pieces, 1 barrier per wave)?
// -- the structure of a 4-wave, 128 x 128-per-wave GEMM core (accumulators in 256 registers), against gemm16_kernel's two wave groups
// in ping-pong (2 barriers per 16 MFMAs).
//
//   hipcc --offload-arch=gfx950 -O3 tools/micro/lonewave.hip -o /tmp/lonewave && /tmp/lonewave > profiles/r4/lonewave_r4.txt
#include <hip/hip_runtime.h>
#ifndef LW_ASM
#define LW_ASM 1
#endif

#include <algorithm>
#include <cstdio>
#include <vector>

typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) short s16x8;

// MODE bits: 1 fragment reads (one ds_read_b128 per 4 MFMAs), 2 LDS-DMA pieces (one per 16 MFMAs) from a global buffer, 4 barrier per 128 MFMAs
template <int MODE, int WAVES>
__global__ __launch_bounds__(64 * WAVES, 1) void k(unsigned long long* out, const char* __restrict__ src, float* sink, int iters) {
  extern __shared__ __attribute__((aligned(1024))) char smem[];
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  f32x4 acc[64];
#pragma unroll
  for (int i = 0; i < 64; ++i) acc[i] = (f32x4)(0.f);
  s16x8 fa[8], fb[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) { fa[i] = (s16x8)((short)(lane & 3)); fb[i] = (s16x8)((short)1); }
  for (int i = threadIdx.x; i < 32768; i += blockDim.x) reinterpret_cast<int*>(smem)[i] = 0;
  __syncthreads();
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(src), 0, 0x40000000, 0x00020000);
  const char* rd = smem + lane * 16 + wid * 8192;
  unsigned voff = (unsigned)(blockIdx.x * 65536 + threadIdx.x * 16);
  unsigned long long t0, t1;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int s = 0; s < 2; ++s) {            // two k-steps of 32: 64 MFMAs each
      s16x8 na[8], nb[8];                     // the next k-step's fragments: 16 reads dealt out over this k-step's 64 MFMAs
#pragma unroll
      for (int i = 0; i < 8; ++i) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
#if LW_ASM
          asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc[8 * i + j]) : "v"(fa[i]), "v"(fb[j]));   // tied AGPR operand: the tile stays in place
#else
          acc[8 * i + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i], fb[j], acc[8 * i + j], 0, 0, 0);
#endif
          if ((MODE & 1) && (j & 3) == 3) {
            const int q = 2 * i + (j >> 2);
            const s16x8 v = *reinterpret_cast<const s16x8*>(rd + ((it * 2 + s) & 1) * 8192 + q * 512);
            if (q < 8) na[q] = v; else nb[q - 8] = v;
          }
          if ((MODE & 2) && j == 7 && (i & 1)) {   // 4 DMA pieces per 64 MFMAs
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(smem + 65536 + wid * 8192 + (i >> 1) * 1024), 16,
                                                     (int)voff, (it & 63) * 1024, 0, 0);
          }
        }
      }
      if (MODE & 1) {
#pragma unroll
        for (int i = 0; i < 8; ++i) { fa[i] = na[i]; fb[i] = nb[i]; }
      }
    }
    if (MODE & 2) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    if (MODE & 4) __builtin_amdgcn_s_barrier();
  }
  asm volatile("s_nop 15\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
  float sum = 0;
#pragma unroll
  for (int i = 0; i < 64; ++i) sum += acc[i][0];
  if (sum == 12345.f) sink[threadIdx.x] = sum;
  if (lane == 0) out[blockIdx.x * WAVES + wid] = t1 - t0;
}

template <int MODE, int WAVES>
void run(unsigned long long* dout, const char* src, float* sink, const char* what) {
  const int iters = 400, nb = 256;
  auto kern = k<MODE, WAVES>;
  hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
  for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(kern, dim3(nb), dim3(64 * WAVES), 128 * 1024, 0, dout, src, sink, iters);
  hipDeviceSynchronize();
  std::vector<unsigned long long> h(nb * WAVES);
  hipMemcpy(h.data(), dout, h.size() * 8, hipMemcpyDeviceToHost);
  std::vector<double> c(h.size());
  for (size_t i = 0; i < h.size(); ++i) c[i] = (double)h[i] / (iters * 128.0);
  std::sort(c.begin(), c.end());
  printf("%-58s waves/SIMD %d | cycles per MFMA per wave: median %5.1f  p10 %5.1f  p90 %5.1f  (pipe floor %d) -> pipe busy %4.0f %%\n", what, WAVES / 4, c[c.size() / 2],
         c[c.size() / 10], c[c.size() * 9 / 10], 16 * (WAVES / 4), 100.0 * 16 * (WAVES / 4) / c[c.size() / 2]);
  fflush(stdout);
}

int main() {
  unsigned long long* dout;
  char* src;
  float* sink;
  hipMalloc(&dout, 256 * 8 * 8);
  hipMalloc(&src, 64 << 20);
  hipMemset(src, 0, 64 << 20);
  hipMalloc(&sink, 4096);
  printf("# gfx950, 256 workgroups (one per CU), v_mfma_f32_16x16x32_bf16, 128 MFMAs per wave and iteration (a 128 x 128 wave tile, K = 64)\n");
  run<0, 4>(dout, src, sink, "MFMAs only");
  run<1, 4>(dout, src, sink, "+ 32 ds_read_b128 dealt out one per 4 MFMAs");
  run<3, 4>(dout, src, sink, "+ 8 LDS-DMA pieces (1 KiB each) + counted wait");
  run<7, 4>(dout, src, sink, "+ one s_barrier per 128 MFMAs");
  run<5, 4>(dout, src, sink, "reads + barrier, no DMA");
  return 0;
}
