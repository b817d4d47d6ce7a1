// Device side of the sampler's conditioning producers and of its output path (gfx950): SURVEY.md section 8f rows 3 and 4.
// The reference runs these on the host, frame by frame, through cv2 / PIL / torchvision / numpy; none of them is heavy, but
// at 80-frame clips x 3 mask videos they are Python loops in front of (and behind) a denoiser that now takes ~85 ms a step.
//
//  * mmgt_blur_mask_u8      scripts/pose2vid.py:94-114 / scripts/audio2vid.py:131-151 `blur_mask`: resize a mask frame to
//                           64 x 64 (cv2.resize, bilinear), Gaussian blur (cv2.GaussianBlur, ksize 31 or 21, sigma from the
//                           kernel size, reflect-101 border), min-max normalise to 0..255.
//  * mmgt_resample_u8       the four-level mask pyramid of src/dataset/image_processor.py:75-102,311-333: torchvision
//                           `Resize` on a PIL "L" image = PIL's two-pass fixed-point resampling with the triangle filter
//                           (antialiased when shrinking), then ToTensor (/ 255).  The host hands in PIL's own integer
//                           coefficient tables, so the result is BIT-EXACT with PIL (tests/test_conditioning.py).
//  * mmgt_window_stack      `process_audio_emb` (scripts/pose2vid.py:72-91): frame i -> frames clamp(i-2 .. i+2).
//  * mmgt_frames_to_u8      the output path, src/utils/util.py:148-160 `save_videos_grid`: (x * 255).astype(uint8) of the
//                           decoded frames, fused with decode_latents' (x / 2 + 0.5).clamp(0, 1)
//                           (pipeline_pose2vid_long.py:121-123) and the layout change to (frame, h, w, rgb).
#include "common.h"
#include "mmgt_hip.h"

namespace {

__device__ __forceinline__ int reflect101(int i, int n) {
  if (n == 1) return 0;
  while (i < 0 || i >= n) i = i < 0 ? -i : 2 * n - 2 - i;
  return i;
}

// one workgroup per frame; OUT x OUT (64 x 64) working image in LDS
template <int OUT>
__global__ __launch_bounds__(256) void blur_mask_kernel(const unsigned char* __restrict__ in, unsigned char* __restrict__ out,
                                                        int H, int W, int ksize, float sigma) {
  __shared__ float a[OUT * OUT], b[OUT * OUT];
  __shared__ float kern[64];
  __shared__ float red[2][256];
  const int f = blockIdx.x, tid = threadIdx.x;
  const unsigned char* src = in + (long)f * H * W;
  const int r = ksize / 2;
  if (tid == 0) {
    float s = 0.f;
    for (int i = 0; i < ksize; ++i) { kern[i] = expf(-0.5f * (i - r) * (i - r) / (sigma * sigma)); s += kern[i]; }
    for (int i = 0; i < ksize; ++i) kern[i] /= s;
  }
  // cv2.resize INTER_LINEAR: src = (dst + 0.5) * scale - 0.5, taps clamped into the image
  const float sx = (float)W / OUT, sy = (float)H / OUT;
  for (int i = tid; i < OUT * OUT; i += 256) {
    const int y = i / OUT, x = i - y * OUT;
    float fx = (x + 0.5f) * sx - 0.5f, fy = (y + 0.5f) * sy - 0.5f;
    int x0 = (int)floorf(fx), y0 = (int)floorf(fy);
    float ax = fx - x0, ay = fy - y0;
    if (x0 < 0) { x0 = 0; ax = 0.f; }
    if (x0 >= W - 1) { x0 = W - 1 > 0 ? W - 2 : 0; ax = W > 1 ? 1.f : 0.f; }
    if (y0 < 0) { y0 = 0; ay = 0.f; }
    if (y0 >= H - 1) { y0 = H - 1 > 0 ? H - 2 : 0; ay = H > 1 ? 1.f : 0.f; }
    const int x1 = W > 1 ? x0 + 1 : x0, y1 = H > 1 ? y0 + 1 : y0;
    const float p00 = src[(long)y0 * W + x0], p01 = src[(long)y0 * W + x1], p10 = src[(long)y1 * W + x0], p11 = src[(long)y1 * W + x1];
    a[i] = rintf((p00 * (1.f - ax) + p01 * ax) * (1.f - ay) + (p10 * (1.f - ax) + p11 * ax) * ay);   // cv2.resize returns uint8
  }
  __syncthreads();
  for (int i = tid; i < OUT * OUT; i += 256) {           // horizontal pass
    const int y = i / OUT, x = i - y * OUT;
    float s = 0.f;
    for (int k = 0; k < ksize; ++k) s += kern[k] * a[y * OUT + reflect101(x + k - r, OUT)];
    b[i] = s;
  }
  __syncthreads();
  float mn = 1e30f, mx = -1e30f;
  for (int i = tid; i < OUT * OUT; i += 256) {           // vertical pass, rounded to uint8 as cv2.GaussianBlur returns it
    const int y = i / OUT, x = i - y * OUT;
    float s = 0.f;
    for (int k = 0; k < ksize; ++k) s += kern[k] * b[reflect101(y + k - r, OUT) * OUT + x];
    s = fminf(fmaxf(rintf(s), 0.f), 255.f);
    a[i] = s;
    mn = fminf(mn, s);
    mx = fmaxf(mx, s);
  }
  red[0][tid] = mn;
  red[1][tid] = mx;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (tid < o) { red[0][tid] = fminf(red[0][tid], red[0][tid + o]); red[1][tid] = fmaxf(red[1][tid], red[1][tid + o]); }
    __syncthreads();
  }
  mn = red[0][0];
  mx = red[1][0];
  const float sc = mx > mn ? 255.f / (mx - mn) : 0.f;   // cv2.normalize(NORM_MINMAX, 0, 255)
  for (int i = tid; i < OUT * OUT; i += 256) out[(long)f * OUT * OUT + i] = (unsigned char)fminf(fmaxf(rintf((a[i] - mn) * sc), 0.f), 255.f);
}

// PIL ImagingResample for 8-bit single-band images, both passes with the host's coefficient tables (22 fractional bits):
// bounds[o] = (first tap, tap count), kk[o][ksize] = integer weights.  out = u8 / 255 as ToTensor() gives it, or raw u8.
__global__ __launch_bounds__(256) void resample_u8_kernel(const unsigned char* __restrict__ in, float* __restrict__ outf,
                                                          unsigned char* __restrict__ outu, int S, int D,
                                                          const int* __restrict__ bounds, const int* __restrict__ kk, int ksize) {
  extern __shared__ unsigned char lds[];
  unsigned char* tmp = lds;            // [S][D]: after the horizontal pass
  const int f = blockIdx.x, tid = threadIdx.x;
  const unsigned char* src = in + (long)f * S * S;
  for (int i = tid; i < S * D; i += 256) {
    const int y = i / D, x = i - y * D;
    const int x0 = bounds[2 * x], n = bounds[2 * x + 1];
    int acc = 1 << 21;
    for (int k = 0; k < n; ++k) acc += (int)src[y * S + x0 + k] * kk[x * ksize + k];
    acc >>= 22;
    tmp[i] = (unsigned char)(acc < 0 ? 0 : acc > 255 ? 255 : acc);
  }
  __syncthreads();
  for (int i = tid; i < D * D; i += 256) {
    const int y = i / D, x = i - y * D;
    const int y0 = bounds[2 * y], n = bounds[2 * y + 1];
    int acc = 1 << 21;
    for (int k = 0; k < n; ++k) acc += (int)tmp[(y0 + k) * D + x] * kk[y * ksize + k];
    acc >>= 22;
    const int v = acc < 0 ? 0 : acc > 255 ? 255 : acc;
    if (outf) outf[(long)f * D * D + i] = (float)v / 255.f;
    if (outu) outu[(long)f * D * D + i] = (unsigned char)v;
  }
}

__global__ void window_stack_kernel(const float* __restrict__ x, float* __restrict__ out, int L, long D, int half) {
  const int win = 2 * half + 1;
  const long total = (long)L * win * D;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long d = i % D;
    long t = i / D;
    const int j = (int)(t % win);
    const int f = (int)(t / win);
    int s = f + j - half;
    s = s < 0 ? 0 : s >= L ? L - 1 : s;
    out[i] = x[(long)s * D + d];
  }
}

template <typename T>
__global__ void frames_to_u8_kernel(const T* __restrict__ x, unsigned char* __restrict__ out, long npix, int cpad, float scale,
                                    float shift) {
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < npix * 3; i += (long)gridDim.x * blockDim.x) {
    const long p = i / 3;
    const int c = (int)(i - p * 3);
    float v = Elem<T>::ld(x + p * cpad + c) * scale + shift;
    v = fminf(fmaxf(v, 0.f), 1.f) * 255.f;
    out[i] = (unsigned char)v;                       // numpy astype(uint8): truncation
  }
}


inline int grid_for(long n) {
  long g = (n + 255) / 256;
  return (int)(g < 1 ? 1 : (g > 16384 ? 16384 : g));
}

}  // namespace

extern "C" int mmgt_blur_mask_u8(const unsigned char* masks, unsigned char* out, int frames, int H, int W, int ksize, void* stream) {
  MMGT_CHECK(masks && out && frames > 0 && H > 0 && W > 0, "blur_mask: bad arguments");
  MMGT_CHECK(ksize % 2 == 1 && ksize >= 3 && ksize <= 63, "blur_mask: ksize %d must be odd, 3..63", ksize);
  const float sigma = 0.3f * ((ksize - 1) * 0.5f - 1.f) + 0.8f;    // cv2.getGaussianKernel with sigma <= 0
  hipLaunchKernelGGL(blur_mask_kernel<64>, dim3(frames), dim3(256), 0, (hipStream_t)stream, masks, out, H, W, ksize, sigma);
  MMGT_LAUNCH_CHECK();
  return 0;
}

extern "C" int mmgt_resample_u8(const unsigned char* in, float* out_f32, unsigned char* out_u8, int frames, int S, int D,
                                const int* bounds, const int* coeffs, int ksize, void* stream) {
  MMGT_CHECK(in && (out_f32 || out_u8) && bounds && coeffs && frames > 0 && S > 0 && D > 0 && ksize > 0, "resample_u8: bad arguments");
  MMGT_CHECK((long)S * D <= 64 * 1024, "resample_u8: %d x %d intermediate exceeds the LDS budget", S, D);
  hipLaunchKernelGGL(resample_u8_kernel, dim3(frames), dim3(256), (size_t)S * D, (hipStream_t)stream, in, out_f32, out_u8, S, D,
                     bounds, coeffs, ksize);
  MMGT_LAUNCH_CHECK();
  return 0;
}

extern "C" int mmgt_window_stack(const float* x, float* out, int frames, long D, int half, void* stream) {
  MMGT_CHECK(x && out && frames > 0 && D > 0 && half >= 0, "window_stack: bad arguments");
  hipLaunchKernelGGL(window_stack_kernel, dim3(grid_for((long)frames * (2 * half + 1) * D)), dim3(256), 0, (hipStream_t)stream, x,
                     out, frames, D, half);
  MMGT_LAUNCH_CHECK();
  return 0;
}

extern "C" int mmgt_frames_to_u8(const void* x, unsigned char* out, long npix, int cpad, float scale, float shift, int dtype,
                                 void* stream) {
  MMGT_CHECK(x && out && npix > 0 && cpad >= 3, "frames_to_u8: bad arguments");
  MMGT_CHECK(dtype == MMGT_F32 || dtype == MMGT_BF16, "frames_to_u8: bad dtype");
  if (dtype == MMGT_BF16)
    hipLaunchKernelGGL(frames_to_u8_kernel<bf16_t>, dim3(grid_for(npix * 3)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x,
                       out, npix, cpad, scale, shift);
  else
    hipLaunchKernelGGL(frames_to_u8_kernel<float>, dim3(grid_for(npix * 3)), dim3(256), 0, (hipStream_t)stream, (const float*)x, out,
                       npix, cpad, scale, shift);
  MMGT_LAUNCH_CHECK();
  return 0;
}

