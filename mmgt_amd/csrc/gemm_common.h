// Shared by the GEMM / implicit-GEMM conv kernels of libmmgt_hip.so (gemm.hip: 32x32x16 MFMA tiles, both storage types;
// gemm16.hip: the bf16 256x256 8-phase core on 16x16x32 MFMAs): operand / epilogue descriptors and the LDS-DMA helpers.
#pragma once
#include "common.h"

namespace {

struct ADesc {
  const char* src0;
  const char* src1;
  long ld0;           // dense: row stride (elements)
  long bs0, bs1;      // batch (grid.z) stride in elements
  int C0, C1;         // conv: channels of the two sources (Cin = C0 + C1)
  int IH, IW, OH, OW; // conv: stored input dims and output dims
  int stride, up;     // conv: stride; up = 1 -> the conv sees the nearest-2x upsampled input
  int pad;            // conv: zero rows / columns in front (1; 0 for the VAE encoder's (0, 1) padded downsample)
  unsigned fd_hw[3], fd_ow[3];   // conv: division of an output row index by OH * OW and by OW as multiply-high + shifts (fastdiv)
  int ksplit;         // gemm16 split-K: grid.z walks slices of the reduction (K = slice length, ldw = the full reduction length = row
  int ldw;            // stride of W in elements) and the epilogue writes fp32 partial slabs [z][M][N] to ep.out; 0: grid.z = batch
};

// Unsigned division by a launch-time constant as multiply-high and two shifts (Granlund & Montgomery), exact for every 32-bit
// numerator: the conv gather of gemm16.hip re-derives (image, y, x) of its output rows at every tap change instead of keeping
// them in registers.
inline void make_fastdiv(unsigned d, unsigned (&fd)[3]) {
  unsigned l = 0;
  while ((1ull << l) < d) ++l;
  fd[0] = (unsigned)((((1ull << l) - d) << 32) / d + 1);
  fd[1] = l < 1 ? l : 1;
  fd[2] = l > 0 ? l - 1 : 0;
}
__device__ __forceinline__ unsigned fastdiv(unsigned n, const unsigned (&fd)[3]) {
  const unsigned t = __umulhi(fd[0], n);
  return (t + ((n - t) >> fd[1])) >> fd[2];
}

struct Epi {
  const float* bias;       // [N]
  const float* bias2;      // [ceil(M / bias2_rows)][N]   (time-embedding add: one row per CFG batch entry)
  const float* row_scale;  // [M]                         (motion-mask multiply)
  const float* bias_post;  // [N]  added AFTER the row scale / alpha (the zero-conv bias of a merged out-proj . zero-conv)
  const char* residual;    // T [M][ldr]
  char* out;               // T [M][ldo]
  long ldr, ldo, bsr, bso; // strides in elements; bs* = grid.z strides
  int bias2_rows;
  float alpha;
  int act;                 // 0 none, 1 GEGLU (packed weights, out has N/2 columns), 2 SiLU, 3 ReLU, 4 quick-GELU, 5 GELU (erf), 6 Mish
  int fast;                // 1: N % 8 == 0 and every row / pointer 16-byte aligned -> vectorised epilogue
};

// LDS-DMA through a buffer resource (buffer_load_dwordx4 ... offen lds): SGPR descriptor + one 32-bit VGPR offset per lane
// + an SGPR offset for the position along K, instead of a 64-bit VGPR address per lane -- half the address registers, no
// per-chunk vector address arithmetic, and out-of-range offsets READ AS ZERO, which is the conv's zero padding.
// Operands are described as raw buffers of 2 GiB (the host checks the sizes); POISON is any offset beyond that.
constexpr unsigned DMA_RANGE = 0x80000000u, DMA_POISON = 0xC0000000u;
__device__ __forceinline__ __amdgpu_buffer_rsrc_t dma_rsrc(const void* base) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)DMA_RANGE, 0x00020000);
}
__device__ __forceinline__ void blds16(__amdgpu_buffer_rsrc_t r, unsigned voff, int soff, void* lds_dst) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)lds_dst, 16, (int)voff, soff, 0, 0);
}

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

}  // namespace
