// Micro-benchmark 2: which instruction classes of a wave are held up while its SIMD partner sits at an MFMA that the matrix pipe cannot
// take yet?  Waves 0-3 run one of {VALU, ds_read_b128, LDS-DMA buffer loads} loops, waves 4-7 an MFMA loop (32x32x16 or 16x16x32 bf16),
// optionally padded with s_nop to the pipe's cadence.  Prints per-launch times: A alone, MFMA alone, both.  (gfx950, 8 waves per CU.)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) short s16x8;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

template <int KIND, int SHAPE, int PAD>
__global__ __launch_bounds__(512, 2) void k(float* out, const char* src, int iters, int do_a, int do_mfma) {
  __shared__ __attribute__((aligned(1024))) char lds[65536];
  const int wid = threadIdx.x >> 6, lane = threadIdx.x & 63;
  for (int i = threadIdx.x * 16; i < 65536; i += 512 * 16) *reinterpret_cast<u32x4*>(lds + i) = (u32x4)(i);
  __syncthreads();
  if (wid < 4) {
    if (!do_a) return;
    if (KIND == 0) {            // VALU: 128 FMAs per iteration
      float a[8];
      for (int i = 0; i < 8; ++i) a[i] = threadIdx.x * 0.001f + i;
      for (int it = 0; it < iters; ++it)
#pragma unroll
        for (int r = 0; r < 16; ++r)
#pragma unroll
          for (int i = 0; i < 8; ++i) a[i] = fmaf(a[i], 1.0001f, 0.5f);
      float s = 0;
      for (int i = 0; i < 8; ++i) s += a[i];
      out[blockIdx.x * 512 + threadIdx.x] = s;
    } else if (KIND == 1) {     // LDS: 24 ds_read_b128 per iteration (the fragment reads of one gemm16 chunk)
      u32x4 acc = (u32x4)(0u);
      const char* p = lds + wid * 16384 + lane * 16;
      for (int it = 0; it < iters; ++it) {
        u32x4 v[8];
#pragma unroll
        for (int r = 0; r < 3; ++r) {
#pragma unroll
          for (int i = 0; i < 8; ++i) v[i] = *reinterpret_cast<const u32x4*>(p + ((i + 8 * r) & 15) * 1024);
#pragma unroll
          for (int i = 0; i < 8; ++i) asm volatile("" :: "v"(v[i]));
        }
        acc += v[0];
      }
      out[blockIdx.x * 512 + threadIdx.x] = acc[0];
    } else {                    // LDS-DMA: 8 buffer_load ... lds pieces per iteration (1 KiB each, L2-resident source)
      __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(src), 0, 1 << 30, 0x00020000);
      for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i)
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(lds + wid * 16384 + i * 1024), 16, lane * 16,
                                                   ((it * 8 + i) & 1023) * 1024 + wid * (1 << 20), 0, 0);
        asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      out[blockIdx.x * 512 + threadIdx.x] = lds[lane];
    }
  } else {
    if (!do_mfma) return;
    s16x8 x = (s16x8)(1), y = (s16x8)(2);
#define PADS() do { if (PAD == 1) asm volatile("s_nop 3"); if (PAD == 2) asm volatile("s_nop 7\n\ts_nop 1"); } while (0)
    if (SHAPE == 32) {
      f32x16 c0 = (f32x16)(0.f), c1 = c0, c2 = c0, c3 = c0;
      for (int it = 0; it < iters; ++it)
#pragma unroll
        for (int r = 0; r < 4; ++r) {            // 16 MFMAs = 512 pipe cycles per iteration
          c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, c0, 0, 0, 0); PADS();
          c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, c1, 0, 0, 0); PADS();
          c2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, c2, 0, 0, 0); PADS();
          c3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, c3, 0, 0, 0); PADS();
        }
      out[blockIdx.x * 512 + threadIdx.x] = c0[0] + c1[1] + c2[2] + c3[3];
    } else {
      f32x4 c[8];
      for (int i = 0; i < 8; ++i) c[i] = (f32x4)(0.f);
      for (int it = 0; it < iters; ++it)
#pragma unroll
        for (int r = 0; r < 4; ++r)              // 32 MFMAs = 512 pipe cycles per iteration
#pragma unroll
          for (int i = 0; i < 8; ++i) { c[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x, y, c[i], 0, 0, 0); PADS(); }
      float s = 0;
      for (int i = 0; i < 8; ++i) s += c[i][i & 3];
      out[blockIdx.x * 512 + threadIdx.x] = s;
    }
  }
}

template <int KIND, int SHAPE, int PAD>
float run(float* out, const char* src, int iters, int a, int m) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((k<KIND, SHAPE, PAD>), dim3(256), dim3(512), 0, 0, out, src, iters, a, m);
  hipEventRecord(e0);
  for (int i = 0; i < 5; ++i) hipLaunchKernelGGL((k<KIND, SHAPE, PAD>), dim3(256), dim3(512), 0, 0, out, src, iters, a, m);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  return ms / 5 * 1e3f;
}

#define ROW(K, S, P, what) printf("%-62s A alone %8.1f   MFMA alone %8.1f   both %8.1f\n", what, run<K, S, P>(out, src, iters, 1, 0), run<K, S, P>(out, src, iters, 0, 1), run<K, S, P>(out, src, iters, 1, 1))
int main() {
  float* out; char* src;
  hipMalloc(&out, 256 * 512 * 4);
  hipMalloc(&src, 8 << 20);
  hipMemset(src, 1, 8 << 20);
  const int iters = 2000;
  printf("per launch (us), %d iterations; A = waves 0-3, MFMA = waves 4-7 of one 8-wave workgroup per CU\n", iters);
  ROW(0, 32, 0, "A = 128 VALU        | 16 x mfma 32x32x16, back to back");
  ROW(0, 16, 0, "A = 128 VALU        | 32 x mfma 16x16x32, back to back");
  ROW(0, 16, 1, "A = 128 VALU        | 32 x mfma 16x16x32 + s_nop 3");
  ROW(1, 32, 0, "A = 24 ds_read_b128 | 16 x mfma 32x32x16, back to back");
  ROW(1, 16, 0, "A = 24 ds_read_b128 | 32 x mfma 16x16x32, back to back");
  ROW(1, 16, 1, "A = 24 ds_read_b128 | 32 x mfma 16x16x32 + s_nop 3");
  ROW(2, 32, 0, "A = 8 LDS-DMA pieces | 16 x mfma 32x32x16, back to back");
  ROW(2, 16, 0, "A = 8 LDS-DMA pieces | 32 x mfma 16x16x32, back to back");
  return 0;
}
