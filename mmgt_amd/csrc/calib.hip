// Box calibration for bench.py (`box_calib` in the bench line): MI355X boxes differ by several per cent in the clock they hold under an
// MFMA-dense load (MI355X_MICROARCH.md, DVFS give-back items 5 - 7), so a step time can only be compared across boxes beside a number
// that measures the box itself.  mmgt_box_calib runs a bare v_mfma_f32_16x16x32_bf16 loop (operands in registers, random data, one wave
// per SIMD, sixteen independent accumulator tiles) back to back for `warm_seconds`, then reads the in-kernel clock of one more launch:
// delta s_memtime (shader cycles) / delta s_memrealtime (100 MHz) per workgroup, median over the workgroups.
#include <algorithm>
#include <vector>

#include "common.h"
#include "mmgt_hip.h"

namespace {

typedef __attribute__((ext_vector_type(4))) float acc4;

__global__ __launch_bounds__(256) void calib_mfma_kernel(unsigned long long* __restrict__ stamps, float* __restrict__ sink, int iters, unsigned seed) {
  const int lane = threadIdx.x & 63;
  // pseudo-random bf16 operands: the clock a chip holds depends on the data (all-zero operands run ~20 % faster)
  unsigned h = seed ^ (blockIdx.x * 2654435761u) ^ (threadIdx.x * 40503u);
  auto rnd = [&]() { h = h * 1664525u + 1013904223u; return (short)(((h >> 16) & 0x8000u) | 0x3F00u | ((h >> 9) & 0xFFu)); };   // +-[0.5, 2)
  s16x8 a[4], b[4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) { a[i][j] = rnd(); b[i][j] = rnd(); }
  acc4 acc[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = (acc4)(0.f);
  unsigned long long c0, c1, r0, r1;
  asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(c0), "=s"(r0)::"memory");
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j)   // (tied accumulation-register operand: through the builtin hipcc copies the tiles between AGPR ranges and pads
                                    //  with s_nops -- 27 cycles per MFMA instead of 16, and the loop no longer loads the chip: tools/micro/lonewave.hip)
        asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc[4 * i + j]) : "v"(a[i]), "v"(b[j]));
  }
  asm volatile("s_nop 15\n\ts_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(c1), "=s"(r1)::"memory");
  float sum = 0.f;
#pragma unroll
  for (int i = 0; i < 16; ++i) sum += acc[i][0] + acc[i][3];
  if (sum == 12345.678f) sink[threadIdx.x] = sum;       // keeps the loop alive; never true in practice
  if (threadIdx.x == 0) {
    stamps[2 * blockIdx.x] = c1 - c0;
    stamps[2 * blockIdx.x + 1] = r1 - r0;
  }
  (void)lane;
}

}  // namespace

extern "C" int mmgt_box_calib(float warm_seconds, float* mfma_mhz, float* mfma_tflops, void* stream) {
  MMGT_CHECK(mfma_mhz && mfma_tflops && warm_seconds >= 0.f && warm_seconds <= 10.f, "box_calib: bad arguments");
  hipStream_t s = (hipStream_t)stream;
  int dev = 0;
  hipDeviceProp_t prop;
  MMGT_CHECK(hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess, "box_calib: device query failed");
  const int nwg = prop.multiProcessorCount, iters = 20000;      // 320 000 MFMAs per wave: ~2.3 ms per launch at 2.2 GHz
  // 96 KiB of (unused) dynamic LDS per workgroup: at most ONE workgroup per CU, so the nwg workgroups load every CU with one wave per SIMD
  // (without it the dispatcher doubles workgroups up on some CUs and leaves others idle: 1.46 PFLOP/s at 2.39 GHz on a round-5 box)
  constexpr int CALIB_LDS = 96 * 1024;
  static bool attr = false;
  if (!attr) {
    MMGT_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(calib_mfma_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, CALIB_LDS) == hipSuccess,
               "box_calib: cannot reserve LDS");
    attr = true;
  }
  unsigned long long* d_st = nullptr;
  float* d_sink = nullptr;
  MMGT_CHECK(hipMalloc(reinterpret_cast<void**>(&d_st), sizeof(unsigned long long) * 2 * nwg) == hipSuccess &&
                 hipMalloc(reinterpret_cast<void**>(&d_sink), sizeof(float) * 256) == hipSuccess,
             "box_calib: allocation failed");
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  float spent_ms = 0.f, last_ms = 0.f;
  unsigned seed = 1;
  do {                                                            // back-to-back launches until the chip has settled under the load
    (void)hipEventRecord(e0, s);
    for (int k = 0; k < 8; ++k) hipLaunchKernelGGL(calib_mfma_kernel, dim3(nwg), dim3(256), CALIB_LDS, s, d_st, d_sink, iters, seed++);
    (void)hipEventRecord(e1, s);
    (void)hipEventSynchronize(e1);
    (void)hipEventElapsedTime(&last_ms, e0, e1);
    spent_ms += last_ms;
  } while (spent_ms < warm_seconds * 1e3f);
  std::vector<unsigned long long> st(2 * nwg);
  const hipError_t ce = hipMemcpy(st.data(), d_st, sizeof(unsigned long long) * 2 * nwg, hipMemcpyDeviceToHost);
  (void)hipFree(d_st);
  (void)hipFree(d_sink);
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  MMGT_CHECK(ce == hipSuccess && hipGetLastError() == hipSuccess, "box_calib: the calibration launch failed");
  std::vector<double> mhz;
  for (int i = 0; i < nwg; ++i)
    if (st[2 * i + 1] > 0) mhz.push_back((double)st[2 * i] / (double)st[2 * i + 1] * 100.0);
  MMGT_CHECK(!mhz.empty(), "box_calib: no stamps came back");
  std::sort(mhz.begin(), mhz.end());
  *mfma_mhz = (float)mhz[mhz.size() / 2];
  // 8 launches x nwg workgroups x 4 waves x iters x 16 MFMAs x 16 384 FLOP over the last timed batch
  *mfma_tflops = (float)(8.0 * nwg * 4.0 * iters * 16.0 * 16384.0 / (last_ms * 1e-3) / 1e12);
  return 0;
}
