// kmath.hpp -- scalar device/host math shared by the HIP kernels of libmodsx.
//
// Every function states the reference lines whose arithmetic (operation order,
// f32/f64 placement, truncations) it reproduces bit for bit.  Compiled with
// -ffp-contract=off: no FMA contraction anywhere, IEEE f32/f64 div and sqrt.
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#define MX_HD __host__ __device__ __forceinline__
#define MX_D __device__ __forceinline__

namespace mx {

// atan2LUTff, detectors/helpers.cpp:160-207; L = 256-entry f64 table (:30-72).
// The reference is an eight-way branch on the signs of x, y and on |x| > |y|; every branch looks up
// L[(int)(255.f * small / big)] and returns (float)(c +- L) with c in {0, +-pi/2, +-pi} (f32 constants widened to f64).
// Written without branches -- one division, one look-up, selects -- because the eight paths of a wavefront would
// otherwise run one after the other.  (float)(-L) and (float)(L) are kept as plain negation / identity so that the sign
// of a zero result is the reference's.
// Split in two so that a kernel that only needs a FUNCTION of the angle can tabulate it: the result depends on (y, x) only
// through the table index, three sign / octant bits and one special case, 8 x 256 + 1 possible values in all.
//   atan2lut_case:  code = xp << 2 | yp << 1 | big, idx = clamped table index; returns true for the special case
//                   (x == 0 with y <= 0: the reference returns 0 there)
//   atan2lut_value: the angle of (code, idx)
MX_HD bool atan2lut_case(float y, float x, int &code, int &idx) {
  const float ax = fabsf(x), ay = fabsf(y);
  const bool xp = x > 0.f, yp = y > 0.f, big = ax > ay;
  const float num = big ? ay : ax, den = big ? ax : ay;
  const float q = 255.f * num / den;           // NaN only for x = y = 0, which is the special case
  int i = (int)q;
  idx = i < 0 ? 0 : (i > 255 ? 255 : i);
  code = (xp ? 4 : 0) | (yp ? 2 : 0) | (big ? 1 : 0);
  return !xp && !yp && !big && x == 0.f;
}
MX_HD float atan2lut_value(const double *L, int code, int idx) {
  const float PI_2f = 1.57079632679489661923f, PIf = 3.14159265358979323846f;
  const bool xp = code & 4, yp = code & 2, big = code & 1;
  const double Lv = L[idx];
  // x>0,y>0: L | pi/2 - L      x>0,y<=0: -L | -pi/2 + L      x<=0,y>0: pi - L | pi/2 + L      x<=0,y<=0: -pi + L | -pi/2 - L
  const bool neg = xp ? (yp ? !big : big) : (yp ? big : !big);
  const double sL = neg ? -Lv : Lv;
  double c;
  if (big) c = xp ? 0.0 : (yp ? (double)PIf : (double)(-PIf));
  else c = yp ? (double)PI_2f : (double)(-PI_2f);
  return (big && xp) ? (float)sL : (float)(c + sL);
}
MX_HD float atan2lut(const double *L, float y, float x) {
  int code, idx;
  const bool special = atan2lut_case(y, x, code, idx);
  const float r = atan2lut_value(L, code, idx);
  return special ? 0.f : r;
}

// solveLinear3x3, detectors/helpers.cpp:309-368
MX_HD void solve3(float *A, float *b) {
  int i = 0, pr = 0;
  float vp = fabsf(A[0]);
  float tmp = fabsf(A[3]);
  if (tmp > vp) { pr = 3; i = 1; vp = tmp; }
  if (fabsf(A[6]) > vp) { pr = 6; i = 2; }
  if (pr != 0) {
    float t;
    t = A[pr]; A[pr] = A[0]; A[0] = t;
    t = A[pr + 1]; A[pr + 1] = A[1]; A[1] = t;
    t = A[pr + 2]; A[pr + 2] = A[2]; A[2] = t;
    t = b[i]; b[i] = b[0]; b[0] = t;
  }
  vp = A[3] / A[0];
  A[4] -= vp * A[1]; A[5] -= vp * A[2]; b[1] -= vp * b[0];
  vp = A[6] / A[0];
  A[7] -= vp * A[1]; A[8] -= vp * A[2]; b[2] -= vp * b[0];
  if (fabsf(A[4]) < fabsf(A[7])) {
    float t;
    t = A[7]; A[7] = A[4]; A[4] = t;
    t = A[8]; A[8] = A[5]; A[5] = t;
    t = b[2]; b[2] = b[1]; b[1] = t;
  }
  vp = A[7] / A[4];
  A[8] -= vp * A[5]; b[2] -= vp * b[1];
  b[2] = (b[2]) / A[8];
  b[1] = (b[1] - A[5] * b[2]) / A[4];
  b[0] = (b[0] - A[2] * b[2] - A[1] * b[1]) / A[0];
}

// invSqrt, detectors/helpers.cpp:463-502 (Jacobi rotation in f64, f32 in/out)
MX_HD void inv_sqrt(float &a, float &b, float &c, float &l1, float &l2) {
  double t, r;
  if (b != 0) {
    r = double(c - a) / (2 * b);
    if (r >= 0) t = 1.0 / (r + sqrt(1 + r * r));
    else t = -1.0 / (-r + sqrt(1 + r * r));
    r = 1.0 / sqrt(1 + t * t);
    t = t * r;
  } else { r = 1; t = 0; }
  double x = 1.0 / sqrt(r * r * a - 2 * r * t * b + t * t * c);
  double z = 1.0 / sqrt(t * t * a + 2 * r * t * b + r * r * c);
  double d = sqrt(x * z);
  x /= d; z /= d;
  if (x < z) { l1 = float(z); l2 = float(x); } else { l1 = float(x); l2 = float(z); }
  a = float(r * r * x + t * t * z);
  b = float(-r * t * x + t * r * z);
  c = float(t * t * x + r * r * z);
}

// getEigenvalues, detectors/helpers.cpp:504-515
MX_HD bool eigenvalues(float a, float b, float c, float d, float &l1, float &l2) {
  float trace = a + d;
  float delta1 = (trace * trace - 4 * (a * d - b * c));
  if (delta1 < 0) return false;
  float delta = sqrtf(delta1);
  l1 = (trace + delta) / 2.0f;
  l2 = (trace - delta) / 2.0f;
  return true;
}

// interpolateCheckBorders, detectors/helpers.cpp:524-549: true when a corner of the sampled window has
// floor(x) <= 0 || floor(y) <= 0 || ceil(x) >= width || ceil(y) >= height.  For the integer-valued bounds these are
// x < 1, y < 1, x > width - 1, y > height - 1 (a NaN corner is false either way), so the four corners reduce to the extrema
// of their coordinates: no floor / ceil per corner.
MX_HD bool check_borders(int orig_w, int orig_h, float ofsx, float ofsy, float a11, float a12, float a21, float a22,
                         int res_w, int res_h) {
  const int width = orig_w - 2, height = orig_h - 2;
  const float hw = (float)ceil((double)(float)res_w / 2.0);
  const float hh = (float)ceil((double)(float)res_h / 2.0);
  float xmin = 0.f, xmax = 0.f, ymin = 0.f, ymax = 0.f;
#pragma unroll
  for (int i = 0; i < 4; i++) {
    const float xs = (i < 2) ? -hw : hw;
    const float ys = (i & 1) ? hh : -hh;
    const float imx = ofsx + xs * a11 + ys * a12;
    const float imy = ofsy + xs * a21 + ys * a22;
    if (i == 0) { xmin = xmax = imx; ymin = ymax = imy; }
    else { xmin = fminf(xmin, imx); xmax = fmaxf(xmax, imx); ymin = fminf(ymin, imy); ymax = fmaxf(ymax, imy); }
  }
  return xmin < 1.f || ymin < 1.f || xmax > (float)(width - 1) || ymax > (float)(height - 1);
}

#ifdef __HIPCC__
typedef const float __attribute__((address_space(1))) *gcfloat_p;
MX_HD gcfloat_p as_global(const float *p) { return (gcfloat_p)p; }
#endif

// one bilinear sample of interpolate(), detectors/helpers.cpp:566-616.
// touch = false: the (int) truncation branch; touch = true: floor + bounds branch (0 outside).
// PT is `const float *` or, in kernels, the same pointer cast to the global address space (as_global): image pointers
// that come out of job tables are generic to the compiler, which would otherwise emit FLAT loads for every tap.
// Addressing: the four neighbours are read at 32-bit BYTE offsets from the (wave-uniform) image base -- one 24-bit
// multiply, an add and a shift per tap, and the loads take the scalar base + vector offset form; the 64-bit
// multiply-add per row pointer this replaces was a third of the instructions of a tap.  An image level is far below
// 2^30 pixels (images are limited to 16384 px per side / 64 Mpx where they are created, capi.hip image_size_ok).

// Workgroups are dispatched round-robin over the 8 XCDs (block b runs on XCD b % 8 and is that XCD's (b / 8)-th block), each
// XCD with its own L2.  For a work list whose neighbours share input (jobs sorted by image, level, position): logical item
// (b & 7) * ceil(n / 8) + (b >> 3), i.e. XCD x takes the x-th CONTIGUOUS eighth of the list and its L2 holds one part of the
// images instead of all of them.  The grid is 8 * ceil(n / 8) blocks; results >= n have no work.  Placement only changes
// speed: the mapping is a bijection whatever XCD a block really runs on.
MX_D int xcd_chunk(int b, int n) {
  const int per = (n + 7) >> 3;
  return (b & 7) * per + (b >> 3);
}

template <class PT>
MX_D float bilinear_blend(PT im, int cols, int x, int y, float WX, float WY) {
  const unsigned off = (__umul24((unsigned)y, (unsigned)cols) + (unsigned)x) << 2;
  typedef const char __attribute__((address_space(1))) *gbyte_p;
  const PT R0 = (PT)((gbyte_p)im + off), R1 = (PT)((gbyte_p)(im + cols) + off);   // both bases are wave-uniform
  const float wx = WX - (float)x;
  const float I1 = wx * (R0[1] - R0[0]) + R0[0];
  return (WY - (float)y) * (wx * (R1[1] - R1[0]) + R1[0] - I1) + I1;
}
template <class PT>
MX_D float bilinear_tap(PT im, int rows, int cols, float WX, float WY, bool touch) {
  if (!touch) {
    int x = (int)WX, y = (int)WY;
    // the reference would fault on NaN / far out-of-range coordinates; keep device loads in bounds
    x = min(max(x, 0), cols - 2);
    y = min(max(y, 0), rows - 2);
    return bilinear_blend(im, cols, x, y, WX, WY);
  }
  const int x = (int)floorf(WX), y = (int)floorf(WY);
  if (WX >= 0 && WY >= 0 && x < cols - 1 && y < rows - 1) return bilinear_blend(im, cols, x, y, WX, WY);
  return 0.f;
}
// the same sample without a divergent branch (all the taps of a lane can then be issued together): out-of-image
// coordinates read a clamped address and select 0
template <class PT>
MX_D float bilinear_tap_touch_select(PT im, int rows, int cols, float WX, float WY) {
  const int x = (int)floorf(WX), y = (int)floorf(WY);
  const bool in = WX >= 0 && WY >= 0 && x < cols - 1 && y < rows - 1;
  const float v = bilinear_blend(im, cols, in ? x : 0, in ? y : 0, WX, WY);
  return in ? v : 0.f;
}

}  // namespace mx
