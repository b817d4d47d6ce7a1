// Micro-benchmark: does a wave's VALU work overlap with its SIMD partner's MFMAs?  (gfx950, two waves per SIMD, 512 threads per CU.)
// Waves 0-3 run a VALU-only loop (independent FMA chains), waves 4-7 an MFMA-only loop (v_mfma_f32_32x32x16_bf16 on 4 accumulators);
// the accumulators are arch VGPRs (what hipcc selects when a kernel fits 256 registers) or ACC registers (forced by an AGPR-constrained
// asm).  Prints the time of: VALU alone, MFMA alone, both together, for both forms.   hipcc --offload-arch=gfx950 -O3 coexec.hip -o coexec
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) short s16x8;

template <bool AGPR, int PAD>
__global__ __launch_bounds__(512, 2) void k(float* out, int iters, int do_valu, int do_mfma) {
  if (AGPR) { float t; asm volatile("" : "=a"(t)); (void)t; }
  const int wid = threadIdx.x >> 6;
  if (wid < 4) {
    if (!do_valu) return;
    float a[8];
    for (int i = 0; i < 8; ++i) a[i] = threadIdx.x * 0.001f + i;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int r = 0; r < 16; ++r)
#pragma unroll
        for (int i = 0; i < 8; ++i) a[i] = fmaf(a[i], 1.0001f, 0.5f);          // 128 independent-ish VALU per iteration
    }
    float s = 0;
    for (int i = 0; i < 8; ++i) s += a[i];
    out[blockIdx.x * 512 + threadIdx.x] = s;
  } else {
    if (!do_mfma) return;
    f32x16 c0 = (f32x16)(0.f), c1 = c0, c2 = c0, c3 = c0;
    s16x8 x = (s16x8)(1), y = (s16x8)(2);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {                                            // 16 MFMAs per iteration = 512 pipe cycles
#define PADS() do { if (PAD >= 1) asm volatile("s_nop 7"); if (PAD >= 2) asm volatile("s_nop 7"); if (PAD >= 3) asm volatile("s_nop 7"); if (PAD >= 4) asm volatile("s_nop 3"); } while (0)
        c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, c0, 0, 0, 0); PADS();
        c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, c1, 0, 0, 0); PADS();
        c2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, c2, 0, 0, 0); PADS();
        c3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, c3, 0, 0, 0); PADS();
      }
    }
    out[blockIdx.x * 512 + threadIdx.x] = c0[0] + c1[1] + c2[2] + c3[3];
  }
}

template <bool AGPR, int PAD>
float run(float* out, int iters, int v, int m) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((k<AGPR, PAD>), dim3(256), dim3(512), 0, 0, out, iters, v, m);
  hipEventRecord(e0);
  for (int i = 0; i < 5; ++i) hipLaunchKernelGGL((k<AGPR, PAD>), dim3(256), dim3(512), 0, 0, out, iters, v, m);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  return ms / 5 * 1e3f;
}

int main() {
  float* out;
  hipMalloc(&out, 256 * 512 * 4);
  const int iters = 2000;
  printf("per launch (us), %d iterations of {128 VALU | 16 MFMA 32x32x16}: one workgroup of 8 waves per CU\n", iters);
#define ROW(A, P, what) printf("%s:  VALU alone %8.1f   MFMA alone %8.1f   both %8.1f\n", what, run<A, P>(out, iters, 1, 0), run<A, P>(out, iters, 0, 1), run<A, P>(out, iters, 1, 1))
  ROW(false, 0, "VGPR-form MFMA, back to back         ");
  ROW(true, 0, "ACC-form  MFMA, back to back         ");
  ROW(false, 2, "VGPR-form MFMA + 16 cycles of s_nop  ");
  ROW(false, 3, "VGPR-form MFMA + 24 cycles of s_nop  ");
  ROW(false, 4, "VGPR-form MFMA + 28 cycles of s_nop  ");
  ROW(true, 3, "ACC-form  MFMA + 24 cycles of s_nop  ");
  ROW(true, 4, "ACC-form  MFMA + 28 cycles of s_nop  ");
  return 0;
}
