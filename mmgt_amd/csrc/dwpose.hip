// Key points -> pose / mask frames on the device (gfx950): the DWPose drawing code of the reference, SURVEY.md section 8f-1 (second half).
//
// The reference turns SMGA's key points into FOUR mp4 files on the host (data/extract_movment_mask_all.py:319-321 `pose_vid_generator` ->
// `process_keypoints` -> src/dwpose/__init__.py:220-283 `DWposeDetector_movment_mask.__call__`: per frame `draw_pose` + three box masks, all
// cv2 on a 512 x 512 canvas) and reads them back (scripts/audio2vid.py:386,426-430).  mmgt_dwpose_draw draws the same frames straight into
// device memory -- the Stage-2 conditioning producers (mask blur / pyramid, PoseGuider input) take them from there, no codec round trip.
//
//   draw_bodypose   src/dwpose/util.py:79-157    17 limbs = cv2.ellipse2Poly(centre, (len / 2, 4), angle) -> cv2.fillConvexPoly, canvas * 0.9, 18 joints r = 4
//   draw_handpose   src/dwpose/util.py:160-206   per hand 20 cv2.line(thickness 2) in HSV-wheel colours, 21 cv2.circle r = 4
//   draw_facepose   src/dwpose/util.py:291-302   68 cv2.circle r = 3, white
//   box masks       src/dwpose/util.py:208-230,349-388; __init__.py:147-196,266 (face = face box + hand boxes, uint8 wrap-around)
//
// cv2 is restated, not linked: OpenCV 4.x drawing.cpp's ellipse2Poly (integer-degree sine table, cvRound), FillConvexPoly (outline by Line /
// Line2, spans from two edge chains in 16.16 fixed point), the 8-connected LineIterator behind clipLine, ThickLine (quadrilateral + end discs)
// and the midpoint Circle.  Integer work throughout; the only floating point is the reference's own float32 arithmetic between the key points
// and the integer arguments (kept in the order numpy evaluates it, no contraction: -ffp-contract=off) and ellipse2Poly's double products.
// Bit-exact against oracle/dwpose_ref.py on uint8 (tests/test_dwpose.py); the oracle's cv2 restatement itself is unpinned (no cv2 here).
//
// One workgroup per frame.  Drawing is a painter's algorithm -- later primitives overwrite earlier ones -- so primitives of different colour are
// separated by a workgroup barrier (whose vmcnt(0) retires the earlier stores) and each primitive is drawn by all threads: one polygon vertex,
// one outline edge, one span row or one disc row per thread.  Nothing is read back from the canvas: the reference's `canvas * 0.9` after the limbs
// only ever sees limb colours or zero, so the limbs are drawn in their scaled colours.  A span row's two edge positions come from the closed
// form of FillConvexPoly's incremental walk (x = xs + dx (y - y_act) of the chain's edge that is active on that row), so rows are independent.
#include "common.h"
#include "dwpose_sintable.h"
#include "mmgt_hip.h"

namespace {

constexpr int XY_SHIFT = 16, XY_ONE = 1 << XY_SHIFT, CH = 512, CW = 512, NPT = 134, MAXV = 361;
typedef long long i64;
typedef unsigned char u8;

struct P2 { i64 x, y; };

__device__ __forceinline__ void put3(u8* img, int x, int y, unsigned c) {
  u8* p = img + ((long)y * CW + x) * 3;
  p[0] = (u8)c; p[1] = (u8)(c >> 8); p[2] = (u8)(c >> 16);
}
__device__ __forceinline__ void hline(u8* img, int y, int x0, int x1, unsigned c) {
  for (int x = x0; x <= x1; ++x) put3(img, x, y, c);
}

// cv::clipLine
__device__ bool clip_line(i64 width, i64 height, P2& a, P2& b) {
  const i64 right = width - 1, bottom = height - 1;
  i64 &x1 = a.x, &y1 = a.y, &x2 = b.x, &y2 = b.y;
  int c1 = (x1 < 0) + (x1 > right) * 2 + (y1 < 0) * 4 + (y1 > bottom) * 8;
  int c2 = (x2 < 0) + (x2 > right) * 2 + (y2 < 0) * 4 + (y2 > bottom) * 8;
  if ((c1 & c2) == 0 && (c1 | c2) != 0) {
    i64 t;
    if (c1 & 12) {
      t = c1 < 8 ? 0 : bottom;
      x1 += (i64)((double)(t - y1) * (double)(x2 - x1) / (double)(y2 - y1));
      y1 = t;
      c1 = (x1 < 0) + (x1 > right) * 2;
    }
    if (c2 & 12) {
      t = c2 < 8 ? 0 : bottom;
      x2 += (i64)((double)(t - y2) * (double)(x2 - x1) / (double)(y2 - y1));
      y2 = t;
      c2 = (x2 < 0) + (x2 > right) * 2;
    }
    if ((c1 & c2) == 0 && (c1 | c2) != 0) {
      if (c1) {
        t = c1 == 1 ? 0 : right;
        y1 += (i64)((double)(t - x1) * (double)(y2 - y1) / (double)(x2 - x1));
        x1 = t;
        c1 = 0;
      }
      if (c2) {
        t = c2 == 1 ? 0 : right;
        y2 += (i64)((double)(t - x2) * (double)(y2 - y1) / (double)(x2 - x1));
        x2 = t;
        c2 = 0;
      }
    }
  }
  return (c1 | c2) == 0;
}

// cv::Line, connectivity 8
__device__ void line8(u8* img, P2 a, P2 b, unsigned c) {
  if (!clip_line(CW, CH, a, b)) return;
  int dx = (int)(b.x - a.x), dy = (int)(b.y - a.y), sx = 1, sy = 1;
  if (dx < 0) { dx = -dx; sx = -1; }
  if (dy < 0) { dy = -dy; sy = -1; }
  const bool vert = dy > dx;
  if (vert) { const int t = dx; dx = dy; dy = t; }
  int err = dx - (dy + dy);
  const int plus = dx + dx, minus = -(dy + dy);
  int x = (int)a.x, y = (int)a.y;
  for (int i = 0; i <= dx; ++i) {
    put3(img, x, y, c);
    const bool m = err < 0;
    err += minus + (m ? plus : 0);
    if (vert) { y += sy; if (m) x += sx; } else { x += sx; if (m) y += sy; }
  }
}

__device__ __forceinline__ i64 cdiv(i64 a, i64 b) { return a / b; }   // C division truncates toward zero

// cv::Line2 (fixed-point end points)
__device__ void line2(u8* img, P2 a, P2 b, unsigned c) {
  if (!clip_line((i64)CW << XY_SHIFT, (i64)CH << XY_SHIFT, a, b)) return;
  i64 dx = b.x - a.x, dy = b.y - a.y;
  const i64 ax = dx < 0 ? -dx : dx, ay = dy < 0 ? -dy : dy;
  i64 x_step, y_step;
  int ecount;
  auto put = [&](i64 x, i64 y) { if (x >= 0 && x < CW && y >= 0 && y < CH) put3(img, (int)x, (int)y, c); };
  if (ax > ay) {
    if (dx < 0) { dy = -dy; const P2 t = a; a = b; b = t; }
    x_step = XY_ONE;
    y_step = cdiv(dy << XY_SHIFT, ax | 1);
    ecount = (int)((b.x - a.x) >> XY_SHIFT);
  } else {
    if (dy < 0) { dx = -dx; const P2 t = a; a = b; b = t; }
    x_step = cdiv(dx << XY_SHIFT, ay | 1);
    y_step = XY_ONE;
    ecount = (int)((b.y - a.y) >> XY_SHIFT);
  }
  a.x += XY_ONE >> 1;
  a.y += XY_ONE >> 1;
  put((b.x + (XY_ONE >> 1)) >> XY_SHIFT, (b.y + (XY_ONE >> 1)) >> XY_SHIFT);
  if (ax > ay) {
    a.x >>= XY_SHIFT;
    while (ecount >= 0) { put(a.x, a.y >> XY_SHIFT); a.x++; a.y += y_step; ecount--; }
  } else {
    a.y >>= XY_SHIFT;
    while (ecount >= 0) { put(a.x >> XY_SHIFT, a.y); a.x += x_step; a.y++; ecount--; }
  }
}

// Half width of a filled cv::Circle's row at vertical distance d (the midpoint loop: rows +-dy get dx, rows +-dx get dy)
__device__ int circle_half(int radius, int d) {
  int hw = -1, err = 0, dx = radius, dy = 0, plus = 1, minus = (radius << 1) - 1;
  while (dx >= dy) {
    if (dy == d && dx > hw) hw = dx;
    if (dx == d && dy > hw) hw = dy;
    dy++;
    err += plus;
    plus += 2;
    const int mask = (err <= 0) - 1;
    err -= minus & mask;
    dx += mask;
    minus -= mask & 2;
  }
  return hw;
}
// one row of a filled circle (row index j = 0 .. 2 radius)
__device__ void circle_row(u8* img, int cx, int cy, int radius, int j, unsigned c) {
  const int d = j - radius, y = cy + d;
  if (y < 0 || y >= CH) return;
  const int hw = circle_half(radius, d < 0 ? -d : d);
  const int x0 = cx - hw < 0 ? 0 : cx - hw, x1 = cx + hw > CW - 1 ? CW - 1 : cx + hw;
  if (hw >= 0 && x0 <= x1) hline(img, y, x0, x1, c);
}

// FillConvexPoly of the polygon v[0 .. n) (LDS), by all threads of the workgroup: outline edge per thread, span row per thread.
// shift = 0: integer vertices, outline by line8; shift = 16: fixed-point vertices, outline by line2.
__device__ void fill_convex_poly(u8* img, const P2* v, int n, unsigned c, int shift, int tid, int nthr) {
  const i64 delta = ((i64)1 << shift) >> 1;
  for (int e = tid; e < n; e += nthr) {
    const P2 p0 = v[e == 0 ? n - 1 : e - 1], p1 = v[e];
    if (shift == 0) line8(img, p0, p1, c);
    else line2(img, p0, p1, c);
  }
  // every thread derives the polygon's extent itself (n <= 361 reads of LDS: cheaper than a reduction + barrier)
  i64 xmin = v[0].x, xmax = v[0].x, ymin = v[0].y, ymax = v[0].y;
  int imin = 0;
  for (int i = 0; i < n; ++i) {
    const P2 p = v[i];
    if (p.y < ymin) { ymin = p.y; imin = i; }
    ymax = p.y > ymax ? p.y : ymax;
    xmax = p.x > xmax ? p.x : xmax;
    xmin = p.x < xmin ? p.x : xmin;
  }
  xmin = (xmin + delta) >> shift; xmax = (xmax + delta) >> shift;
  ymin = (ymin + delta) >> shift; ymax = (ymax + delta) >> shift;
  if (n < 3 || xmax < 0 || ymax < 0 || xmin >= CW || ymin >= CH) return;
  const int ytrue = (int)ymax, ylast = ytrue < CH - 1 ? ytrue : CH - 1;
  for (int y = (int)ymin + tid; y <= ylast; y += nthr) {
    if (y == ytrue || y < 0) continue;                 // the walk breaks in front of the polygon's last row; rows above the frame are skipped
    i64 ex[2];
#pragma unroll
    for (int ch = 0; ch < 2; ++ch) {
      const int di = ch == 0 ? 1 : n - 1;
      int idx0 = imin, idx = imin + di;
      if (idx >= n) idx -= n;
      // the chain's first vertex whose row lies below y; idx0 = its predecessor = where the active edge was entered
      for (int k = 0; k < n; ++k) {
        if ((int)((v[idx].y + delta) >> shift) > y) break;
        idx0 = idx;
        idx += di;
        if (idx >= n) idx -= n;
      }
      const int ty = (int)((v[idx].y + delta) >> shift), yact = (int)((v[idx0].y + delta) >> shift);
      i64 xs = v[idx0].x, xe = v[idx].x;
      if (shift != XY_SHIFT) { xs <<= XY_SHIFT - shift; xe <<= XY_SHIFT - shift; }
      const i64 dxe = cdiv((xe - xs) * 2 + (ty - yact), 2 * (i64)(ty - yact));
      ex[ch] = xs + dxe * (y - yact);
    }
    const i64 xl = ex[0] > ex[1] ? ex[1] : ex[0], xr = ex[0] > ex[1] ? ex[0] : ex[1];
    int xx1 = (int)((xl + (XY_ONE >> 1)) >> XY_SHIFT), xx2 = (int)((xr + (XY_ONE >> 1)) >> XY_SHIFT);
    if (xx2 >= 0 && xx1 < CW) {
      if (xx1 < 0) xx1 = 0;
      if (xx2 >= CW) xx2 = CW - 1;
      hline(img, y, xx1, xx2, c);
    }
  }
}

__device__ __forceinline__ unsigned rgb(int r, int g, int b) { return (unsigned)r | ((unsigned)g << 8) | ((unsigned)b << 16); }

__constant__ int LIMB[17][2] = {{2, 3}, {2, 6}, {3, 4}, {4, 5}, {6, 7}, {7, 8}, {2, 9}, {9, 10}, {10, 11}, {2, 12}, {12, 13}, {13, 14}, {2, 1}, {1, 15},
                                {15, 17}, {1, 16}, {16, 18}};
__constant__ unsigned char BODY_RGB[18][3] = {{255, 0, 0}, {255, 85, 0}, {255, 170, 0}, {255, 255, 0}, {170, 255, 0}, {85, 255, 0}, {0, 255, 0},
                                              {0, 255, 85}, {0, 255, 170}, {0, 255, 255}, {0, 170, 255}, {0, 85, 255}, {0, 0, 255}, {85, 0, 255},
                                              {170, 0, 255}, {255, 0, 255}, {255, 0, 170}, {255, 0, 85}};
__constant__ int HAND_EDGE[20][2] = {{0, 1}, {1, 2}, {2, 3}, {3, 4}, {0, 5}, {5, 6}, {6, 7}, {7, 8}, {0, 9}, {9, 10}, {10, 11}, {11, 12}, {0, 13},
                                     {13, 14}, {14, 15}, {15, 16}, {0, 17}, {17, 18}, {18, 19}, {19, 20}};
// matplotlib.colors.hsv_to_rgb([i / 20, 1, 1]) * 255, reversed, truncated to uint8 (util.py:178-183): written in the array's channel order
__constant__ unsigned char HAND_RGB[20][3] = {{0, 0, 255}, {0, 76, 255}, {0, 153, 255}, {0, 229, 255}, {0, 255, 203}, {0, 255, 127}, {0, 255, 51},
                                              {25, 255, 0}, {102, 255, 0}, {178, 255, 0}, {255, 255, 0}, {255, 178, 0}, {255, 102, 0}, {255, 25, 0},
                                              {255, 0, 50}, {255, 0, 127}, {255, 0, 204}, {229, 0, 255}, {152, 0, 255}, {76, 0, 255}};

// (the reference's float32 arithmetic, one rounding per operation as numpy evaluates it)
__device__ __forceinline__ float fadd(float a, float b) { return __fadd_rn(a, b); }
__device__ __forceinline__ float fmul(float a, float b) { return __fmul_rn(a, b); }
__device__ __forceinline__ float fdiv(float a, float b) { return __fdiv_rn(a, b); }

__global__ __launch_bounds__(256) void dwpose_draw_kernel(const float* __restrict__ kp, u8* __restrict__ pose, u8* __restrict__ hands_m,
                                                          u8* __restrict__ lips_m, u8* __restrict__ face_m) {
  __shared__ float cand[NPT][2];      // candidate / 512 with the invisible points at -1 (faces, hands, lips)
  __shared__ float body[18][2];       // the 18 body joints BEFORE the invisibility overwrite (__init__.py:230 copies them first)
  __shared__ int subset[18];          // joint index or -1
  __shared__ P2 pts[MAXV];
  __shared__ int box[4][4];           // face, lips, hand 0, hand 1: min_x, min_y, max_x, max_y (or an empty box)
  const int f = blockIdx.x, tid = threadIdx.x, nthr = blockDim.x;
  u8* img = pose + (long)f * CH * CW * 3;

  // ---- clear the canvas; key points: denormalize (extract_movment_mask_all.py:128-132), mask_leg (:66-89), / 512, visibility (__init__.py:222-247)
  for (int i = tid; i < CH * CW * 3 / 16; i += nthr) reinterpret_cast<u32x4*>(img)[i] = (u32x4)(0u);
  if (tid < NPT) {
    const float* k = kp + ((long)f * NPT + tid) * 3;
    float x = fadd(fmul(fdiv(fadd(k[0], 1.f), 2.f), 1000.f), -200.f);
    float y = fadd(fmul(fdiv(fadd(k[1], 1.f), 2.f), 1000.f), -200.f);
    float s = fadd(fmul(fdiv(fadd(k[2], 1.f), 2.f), 1000.f), -200.f);      // (the scores go through denormalize with everything else)
    if (tid == 9 || tid == 10 || tid == 12 || tid == 13) { x = 0.f; y = 0.f; s = 0.f; }
    x = fdiv(x, 512.f);
    y = fdiv(y, 512.f);
    if (tid < 18) { body[tid][0] = x; body[tid][1] = y; subset[tid] = s > 0.3f ? tid : -1; }
    const bool inv = s < 0.3f;
    cand[tid][0] = inv ? -1.f : x;
    cand[tid][1] = inv ? -1.f : y;
  }
  __syncthreads();

  // ---- draw_bodypose: limbs
  for (int li = 0; li < 17; ++li) {
    const int ja = LIMB[li][0] - 1, jb = LIMB[li][1] - 1;
    if (subset[ja] < 0 || subset[jb] < 0) continue;                       // (uniform)
    const float Y0 = fmul(body[ja][0], 512.f), Y1 = fmul(body[jb][0], 512.f);   // the reference's names: Y = x * W, X = y * H
    const float X0 = fmul(body[ja][1], 512.f), X1 = fmul(body[jb][1], 512.f);
    const float mX = fdiv(fadd(X0, X1), 2.f), mY = fdiv(fadd(Y0, Y1), 2.f);
    const float dX = fadd(X0, -X1), dY = fadd(Y0, -Y1);
    const float length = __fsqrt_rn(fadd(fmul(dX, dX), fmul(dY, dY)));
    const double ang_d = atan2((double)dX, (double)dY) * (180.0 / 3.141592653589793);   // math.degrees
    const int cx = (int)mY, cy = (int)mX, ax = (int)fdiv(length, 2.f), bx = 4;
    int angle = (int)ang_d;
    while (angle < 0) angle += 360;
    while (angle > 360) angle -= 360;
    const float alpha = DWPOSE_SIN_TABLE[450 - angle], beta = DWPOSE_SIN_TABLE[angle];
    for (int i = tid; i < MAXV; i += nthr) {
      const int a = i < 360 ? i : 360;
      const double x = (double)ax * (double)DWPOSE_SIN_TABLE[450 - a], y = (double)bx * (double)DWPOSE_SIN_TABLE[a];
      const double px = __dadd_rn(__dadd_rn((double)cx, __dmul_rn(x, (double)alpha)), -__dmul_rn(y, (double)beta));
      const double py = __dadd_rn(__dadd_rn((double)cy, __dmul_rn(x, (double)beta)), __dmul_rn(y, (double)alpha));
      pts[i].x = (i64)rint(px);
      pts[i].y = (i64)rint(py);
    }
    __syncthreads();
    const unsigned char* bc = BODY_RGB[li];
    // (canvas * 0.9).astype(uint8) of a limb colour: 255 -> 229, 170 -> 153, 85 -> 76
    auto dim = [](int c) { return (int)((double)c * 0.9); };
    fill_convex_poly(img, pts, MAXV, rgb(dim(bc[0]), dim(bc[1]), dim(bc[2])), 0, tid, nthr);
    __syncthreads();
  }
  // joints
  for (int j = 0; j < 18; ++j) {
    if (subset[j] < 0) continue;
    const int x = (int)fmul(body[j][0], 512.f), y = (int)fmul(body[j][1], 512.f);
    if (tid < 9) circle_row(img, x, y, 4, tid, rgb(BODY_RGB[j][0], BODY_RGB[j][1], BODY_RGB[j][2]));
    __syncthreads();
  }

  // ---- draw_handpose: per hand, 20 edges then 21 joints
  for (int hnd = 0; hnd < 2; ++hnd) {
    const float (*pk)[2] = cand + 92 + 21 * hnd;
    for (int e = 0; e < 20; ++e) {
      const int a = HAND_EDGE[e][0], b = HAND_EDGE[e][1];
      const int x1 = (int)fmul(pk[a][0], 512.f), y1 = (int)fmul(pk[a][1], 512.f);
      const int x2 = (int)fmul(pk[b][0], 512.f), y2 = (int)fmul(pk[b][1], 512.f);
      if (!(x1 > 0 && y1 > 0 && x2 > 0 && y2 > 0)) continue;              // (ints > 1e-5; uniform)
      const unsigned c = rgb(HAND_RGB[e][0], HAND_RGB[e][1], HAND_RGB[e][2]);
      // ThickLine, thickness 2: a quadrilateral in 16.16 fixed point and a disc of radius 1 at both ends
      const i64 p0x = (i64)x1 << XY_SHIFT, p0y = (i64)y1 << XY_SHIFT, p1x = (i64)x2 << XY_SHIFT, p1y = (i64)y2 << XY_SHIFT;
      const double dx = (double)(p0x - p1x) * (1.0 / XY_ONE), dy = (double)(p1y - p0y) * (1.0 / XY_ONE);
      double r = __dadd_rn(__dmul_rn(dx, dx), __dmul_rn(dy, dy));
      const i64 th = (i64)2 << (XY_SHIFT - 1);
      const bool quad = fabs(r) > 2.220446049250313e-16;
      if (quad && tid == 0) {
        r = (double)th / sqrt(r);
        const i64 dpx = (i64)rint(__dmul_rn(dy, r)), dpy = (i64)rint(__dmul_rn(dx, r));
        pts[0] = P2{p0x + dpx, p0y + dpy};
        pts[1] = P2{p0x - dpx, p0y - dpy};
        pts[2] = P2{p1x - dpx, p1y - dpy};
        pts[3] = P2{p1x + dpx, p1y + dpy};
      }
      __syncthreads();
      if (quad) fill_convex_poly(img, pts, 4, c, XY_SHIFT, tid, nthr);
      const int rad = (int)((th + (XY_ONE >> 1)) >> XY_SHIFT);
      if (tid >= 64 && tid < 64 + 2 * rad + 1) circle_row(img, x1, y1, rad, tid - 64, c);
      if (tid >= 128 && tid < 128 + 2 * rad + 1) circle_row(img, x2, y2, rad, tid - 128, c);
      __syncthreads();
    }
    for (int i = tid; i < 21 * 9; i += nthr) {
      const int j = i / 9;
      const int x = (int)fmul(pk[j][0], 512.f), y = (int)fmul(pk[j][1], 512.f);
      if (x > 0 && y > 0) circle_row(img, x, y, 4, i - 9 * j, rgb(0, 0, 255));
    }
    __syncthreads();
  }

  // ---- draw_facepose: 68 white discs of radius 3
  for (int i = tid; i < 68 * 7; i += nthr) {
    const int j = i / 7;
    const int x = (int)fmul(cand[24 + j][0], 512.f), y = (int)fmul(cand[24 + j][1], 512.f);
    if (x > 0 && y > 0) circle_row(img, x, y, 3, i - 7 * j, rgb(255, 255, 255));
  }

  // ---- the three box masks (util.py:208-230,349-388): face = 24..91, lips = 72..91, hands = 92..112 / 113..133
  if (tid < 4) {
    const int lo = tid == 0 ? 24 : tid == 1 ? 72 : tid == 2 ? 92 : 113, hi = tid == 0 ? 92 : tid == 1 ? 92 : tid == 2 ? 113 : 134;
    int x0 = CW, y0 = CH, x1 = 0, y1 = 0;
    for (int j = lo; j < hi; ++j) {
      const int x = (int)fmul(cand[j][0], 512.f), y = (int)fmul(cand[j][1], 512.f);
      if (x > 0 && y > 0) { x0 = x < x0 ? x : x0; y0 = y < y0 ? y : y0; x1 = x > x1 ? x : x1; y1 = y > y1 ? y : y1; }
    }
    const bool ok = x0 < x1 && y0 < y1;
    box[tid][0] = ok ? x0 : 0; box[tid][1] = ok ? y0 : 0; box[tid][2] = ok ? (x1 < CW ? x1 : CW) : 0; box[tid][3] = ok ? (y1 < CH ? y1 : CH) : 0;
  }
  __syncthreads();
  u8* hm = hands_m + (long)f * CH * CW;
  u8* lm = lips_m + (long)f * CH * CW;
  u8* fm = face_m + (long)f * CH * CW;
  for (int i = tid; i < CH * CW / 4; i += nthr) {
    const int y = (i * 4) / CW, x = (i * 4) - y * CW;
    unsigned vh = 0, vl = 0, vf = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      auto in = [&](int b) { return x + k >= box[b][0] && x + k < box[b][2] && y >= box[b][1] && y < box[b][3]; };
      const unsigned h = (in(2) || in(3)) ? 255u : 0u;
      vh |= h << (8 * k);
      vl |= (in(1) ? 255u : 0u) << (8 * k);
      vf |= (((in(0) ? 255u : 0u) + h) & 255u) << (8 * k);                // detected_map_face + detected_map_hands on uint8 (__init__.py:266)
    }
    reinterpret_cast<unsigned*>(hm)[i] = vh;
    reinterpret_cast<unsigned*>(lm)[i] = vl;
    reinterpret_cast<unsigned*>(fm)[i] = vf;
  }
}

}  // namespace

extern "C" int mmgt_dwpose_draw(const float* kp, unsigned char* pose, unsigned char* hands_mask, unsigned char* lips_mask, unsigned char* face_mask,
                                int frames, int H, int W, void* stream) {
  MMGT_CHECK(kp && pose && hands_mask && lips_mask && face_mask && frames > 0, "dwpose_draw: bad arguments");
  MMGT_CHECK(H == CH && W == CW, "dwpose_draw: the reference draws on a 512 x 512 canvas (got %d x %d); resample afterwards", H, W);
  MMGT_CHECK(((uintptr_t)pose % 16) == 0 && ((uintptr_t)hands_mask % 4) == 0 && ((uintptr_t)lips_mask % 4) == 0 && ((uintptr_t)face_mask % 4) == 0,
             "dwpose_draw: misaligned output");
  hipLaunchKernelGGL(dwpose_draw_kernel, dim3(frames), dim3(256), 0, (hipStream_t)stream, kp, pose, hands_mask, lips_mask, face_mask);
  MMGT_LAUNCH_CHECK();
  return 0;
}
