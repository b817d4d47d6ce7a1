// Does v_mfma_f32_16x16x32_bf16 tolerate a destination that PARTIALLY overlaps its accumulator input?  (round 5: hipcc 7.2 emits
// `v_mfma_f32_16x16x32_bf16 v[120:123], ..., v[122:125]` where it rotates accumulator tiles; tools/check_mfma_overlap.py flags the pattern.)
// One wave computes D = A B + C three ways: the builtin (reference), vDst = v[40:43] with SrcC = v[42:45] (overlap of two registers: tuples are 64-bit aligned), alone and at
// the end of a chain of 8 independent MFMAs (a busy matrix pipe), and two controls (tied, disjoint).
//   hipcc --offload-arch=gfx950 -O2 tools/micro/mfma_overlap.hip -o /tmp/mfma_overlap && /tmp/mfma_overlap
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
typedef __attribute__((ext_vector_type(8))) short s16x8;
typedef __attribute__((ext_vector_type(4))) float acc4;

// the same with a chain of independent MFMAs in front (they write v[48:51] / v[52:55] alternately)
__device__ acc4 run(int mode, s16x8 a, s16x8 b, acc4 c) {
  acc4 d = {0, 0, 0, 0};
  if (mode == 0) {          // overlap 2, idle pipe
    asm volatile("v_mov_b32 v42, %4\n\tv_mov_b32 v43, %5\n\tv_mov_b32 v44, %6\n\tv_mov_b32 v45, %7\n\ts_nop 4\n\t"
                 "v_mfma_f32_16x16x32_bf16 v[40:43], %8, %9, v[42:45]\n\ts_nop 7\n\ts_nop 7\n\ts_nop 7\n\t"
                 "v_mov_b32 %0, v40\n\tv_mov_b32 %1, v41\n\tv_mov_b32 %2, v42\n\tv_mov_b32 %3, v43"
                 : "=v"(d[0]), "=v"(d[1]), "=v"(d[2]), "=v"(d[3]) : "v"(c[0]), "v"(c[1]), "v"(c[2]), "v"(c[3]), "v"(a), "v"(b)
                 : "v40", "v41", "v42", "v43", "v44", "v45");
  } else if (mode == 1) {   // tied: vDst = SrcC = v[40:43] (control; VGPR tuples are 64-bit aligned on gfx950, so a partial overlap is always two registers)
    asm volatile("v_mov_b32 v40, %4\n\tv_mov_b32 v41, %5\n\tv_mov_b32 v42, %6\n\tv_mov_b32 v43, %7\n\ts_nop 4\n\t"
                 "v_mfma_f32_16x16x32_bf16 v[40:43], %8, %9, v[40:43]\n\ts_nop 7\n\ts_nop 7\n\ts_nop 7\n\t"
                 "v_mov_b32 %0, v40\n\tv_mov_b32 %1, v41\n\tv_mov_b32 %2, v42\n\tv_mov_b32 %3, v43"
                 : "=v"(d[0]), "=v"(d[1]), "=v"(d[2]), "=v"(d[3]) : "v"(c[0]), "v"(c[1]), "v"(c[2]), "v"(c[3]), "v"(a), "v"(b)
                 : "v40", "v41", "v42", "v43", "v44", "v45");
  } else if (mode == 2) {   // overlap 2 behind 8 independent MFMAs
    asm volatile("v_mov_b32 v42, %4\n\tv_mov_b32 v43, %5\n\tv_mov_b32 v44, %6\n\tv_mov_b32 v45, %7\n\t"
                 "v_mov_b32 v48, 0\n\tv_mov_b32 v49, 0\n\tv_mov_b32 v50, 0\n\tv_mov_b32 v51, 0\n\tv_mov_b32 v52, 0\n\tv_mov_b32 v53, 0\n\tv_mov_b32 v54, 0\n\tv_mov_b32 v55, 0\n\ts_nop 4\n\t"
                 "v_mfma_f32_16x16x32_bf16 v[48:51], %8, %9, v[48:51]\n\tv_mfma_f32_16x16x32_bf16 v[52:55], %8, %9, v[52:55]\n\t"
                 "v_mfma_f32_16x16x32_bf16 v[48:51], %8, %9, v[48:51]\n\tv_mfma_f32_16x16x32_bf16 v[52:55], %8, %9, v[52:55]\n\t"
                 "v_mfma_f32_16x16x32_bf16 v[48:51], %8, %9, v[48:51]\n\tv_mfma_f32_16x16x32_bf16 v[52:55], %8, %9, v[52:55]\n\t"
                 "v_mfma_f32_16x16x32_bf16 v[48:51], %8, %9, v[48:51]\n\tv_mfma_f32_16x16x32_bf16 v[52:55], %8, %9, v[52:55]\n\t"
                 "v_mfma_f32_16x16x32_bf16 v[40:43], %8, %9, v[42:45]\n\ts_nop 7\n\ts_nop 7\n\ts_nop 7\n\t"
                 "v_mov_b32 %0, v40\n\tv_mov_b32 %1, v41\n\tv_mov_b32 %2, v42\n\tv_mov_b32 %3, v43"
                 : "=v"(d[0]), "=v"(d[1]), "=v"(d[2]), "=v"(d[3]) : "v"(c[0]), "v"(c[1]), "v"(c[2]), "v"(c[3]), "v"(a), "v"(b)
                 : "v40", "v41", "v42", "v43", "v44", "v45", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55");
  } else {                  // no overlap (control): vDst = v[40:43], SrcC = v[44:47]
    asm volatile("v_mov_b32 v44, %4\n\tv_mov_b32 v45, %5\n\tv_mov_b32 v46, %6\n\tv_mov_b32 v47, %7\n\ts_nop 4\n\t"
                 "v_mfma_f32_16x16x32_bf16 v[40:43], %8, %9, v[44:47]\n\ts_nop 7\n\ts_nop 7\n\ts_nop 7\n\t"
                 "v_mov_b32 %0, v40\n\tv_mov_b32 %1, v41\n\tv_mov_b32 %2, v42\n\tv_mov_b32 %3, v43"
                 : "=v"(d[0]), "=v"(d[1]), "=v"(d[2]), "=v"(d[3]) : "v"(c[0]), "v"(c[1]), "v"(c[2]), "v"(c[3]), "v"(a), "v"(b)
                 : "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47");
  }
  return d;
}

__global__ void kern(const short* A, const short* B, const float* C, float* out) {
  const int lane = threadIdx.x;
  s16x8 a, b;
  for (int j = 0; j < 8; ++j) { a[j] = A[lane * 8 + j]; b[j] = B[lane * 8 + j]; }
  acc4 c;
  for (int j = 0; j < 4; ++j) c[j] = C[lane * 4 + j];
  const acc4 ref = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
  for (int j = 0; j < 4; ++j) out[lane * 4 + j] = ref[j];
  for (int m = 0; m < 4; ++m) {
    const acc4 d = run(m, a, b, c);
    for (int j = 0; j < 4; ++j) out[(m + 1) * 256 + lane * 4 + j] = d[j];
  }
}

int main() {
  std::vector<short> A(512), B(512);
  std::vector<float> C(256), out(5 * 256);
  srand(1);
  auto bf = [](float f) { unsigned u; memcpy(&u, &f, 4); return (short)(u >> 16); };
  for (auto& v : A) v = bf((rand() % 2001 - 1000) / 500.f);
  for (auto& v : B) v = bf((rand() % 2001 - 1000) / 500.f);
  for (auto& v : C) v = (rand() % 2001 - 1000) / 100.f;
  short *dA, *dB; float *dC, *dO;
  hipMalloc(&dA, 1024); hipMalloc(&dB, 1024); hipMalloc(&dC, 1024); hipMalloc(&dO, 5 * 1024);
  hipMemcpy(dA, A.data(), 1024, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), 1024, hipMemcpyHostToDevice); hipMemcpy(dC, C.data(), 1024, hipMemcpyHostToDevice);
  const char* names[4] = {"vDst v[40:43], SrcC v[42:45], idle pipe", "vDst = SrcC = v[40:43] (control)", "vDst v[40:43], SrcC v[42:45], behind 8 MFMAs", "vDst v[40:43], SrcC v[44:47] (control)"};
  for (int rep = 0; rep < 3; ++rep) {
    hipLaunchKernelGGL(kern, dim3(1), dim3(64), 0, 0, dA, dB, dC, dO);
    hipMemcpy(out.data(), dO, 5 * 1024, hipMemcpyDeviceToHost);
    for (int m = 0; m < 4; ++m) {
      int bad = 0, first = -1;
      for (int i = 0; i < 256; ++i) if (out[(m + 1) * 256 + i] != out[i]) { ++bad; if (first < 0) first = i; }
      printf("run %d: %-48s %3d of 256 results differ from the builtin's%s", rep, names[m], bad, bad ? "" : "\n");
      if (bad) printf(" (first: lane %d register %d: %g instead of %g)\n", first / 4, first % 4, out[(m + 1) * 256 + first], out[first]);
    }
  }
  return 0;
}
