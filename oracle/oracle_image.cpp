/*
 * oracle_image.cpp -- TEST INFRASTRUCTURE ONLY (see oracle.h).
 *
 * CPU restatement of the image primitives the reference's hot path uses.
 * The OpenCV 2.4.9 routines (un-vendored dependency, README.md:11 of the
 * reference) are restated from their published algorithm:
 *   getGaussianKernel / GaussianBlur (separable, f32, BORDER_REPLICATE)
 *   resize(..., 0.5, 0.5, INTER_LINEAR)  == area-fast 2x2 path
 * Everything else follows the cited reference lines.
 */
#include "oracle_internal.hpp"

namespace orc {

/* ---- ATAN LUT (detectors/helpers.cpp:30-72) ---------------------------------
 * The reference table is atan(i/255) printed with 10 decimals, except three
 * entries that carry typos in the reference; those three known answers are
 * kept bug-for-bug.  tests/ pin the SHA-256 of the 256 doubles. */
static double g_atan_lut[256];
static bool g_atan_ready = false;
const double *atan_lut() {
  if (!g_atan_ready) {
    for (int i = 0; i < 256; i++) {
      char buf[64];
      snprintf(buf, sizeof buf, "%.10f", atan(i / 255.0));
      g_atan_lut[i] = strtod(buf, nullptr);
    }
    g_atan_lut[32] = 0.1248376255;
    g_atan_lut[83] = 0.3146752558;
    g_atan_lut[100] = 0.3737268255;
    g_atan_ready = true;
  }
  return g_atan_lut;
}

/* atan2LUTff, detectors/helpers.cpp:160-207 */
float atan2lut(float y, float x) {
  const double *L = atan_lut();
  const float PI_2f = 1.57079632679489661923f, PIf = 3.14159265358979323846f;
  if (x > 0.f) {
    if (y > 0.f) {
      if (x > y) return (float)L[(int)(255.f * y / x)];
      return (float)(PI_2f - L[(int)(255 * x / y)]);
    }
    float ay = fabsf(y);
    if (x > ay) return (float)(-L[(int)(255.f * ay / x)]);
    return (float)(-PI_2f + L[(int)(255.f * x / ay)]);
  }
  if (y > 0.f) {
    float ax = fabsf(x);
    if (ax > y) return (float)(PIf - L[(int)(255.f * y / ax)]);
    return (float)(PI_2f + L[(int)(255.f * ax / y)]);
  }
  float ax = fabsf(x), ay = fabsf(y);
  if (ax > ay) return (float)(-PIf + L[(int)(255.f * ay / ax)]);
  if (x == 0.f) return 0.f;
  return (float)(-PI_2f - L[(int)(255.f * ax / ay)]);
}

/* gray = (B+G+R)/3.0 as a cv::MatExpr, synth-detection.cpp:256-262.
 * OpenCV folds it to addWeighted(B+G, 1/3., R, 1/3., 0) evaluated in double
 * for CV_32F (arithm.cpp addWeighted32f = addWeighted_<float,double>). */
void gray_from_bgr(const uint8_t *bgr, int rows, int cols, float *out) {
  const double k = 1. / 3.0;
  for (size_t i = 0, n = (size_t)rows * cols; i < n; i++) {
    float t = (float)bgr[3 * i] + (float)bgr[3 * i + 1];
    out[i] = (float)((double)t * k + (double)(float)bgr[3 * i + 2] * k + 0.0);
  }
}

/* cv::getGaussianKernel(n, sigma, CV_32F), sigma > 0 */
std::vector<float> gaussian_kernel(int n, double sigma) {
  std::vector<float> k(n);
  double sigmaX = sigma > 0 ? sigma : ((n - 1) * 0.5 - 1) * 0.3 + 0.8;
  double scale2X = -0.5 / (sigmaX * sigmaX);
  double sum = 0;
  for (int i = 0; i < n; i++) {
    double x = i - (n - 1) * 0.5;
    double t = exp(scale2X * x * x);
    k[i] = (float)t;
    sum += k[i];
  }
  sum = 1. / sum;
  for (int i = 0; i < n; i++) k[i] = (float)(k[i] * sum);
  return k;
}

/* kernel size rule of gaussianBlur / gaussianBlurInplace, detectors/helpers.cpp:717-731 */
int blur_ksize(float sigma) {
  int size = (int)(2.0 * 3.0 * sigma + 1.0);
  if (size % 2 == 0) size++;
  return size;
}

static inline int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

/* cv::GaussianBlur(src, dst, Size(n,n), sigma, sigma, BORDER_REPLICATE) for CV_32F.
 * Row pass: ksize<=5 -> SymmRowSmallFilter (centre + symmetric pairs), otherwise
 * RowFilter (taps accumulated left to right).  Column pass: SymmColumnFilter
 * (centre, then pairs (below+above)*k).  Intermediate buffer is f32. */
void gaussian_blur(const Img &in, float sigma, Img &out) {
  const int n = blur_ksize(sigma);
  const int rows = in.rows, cols = in.cols;
  Img dst(rows, cols);
  if (n == 1) { dst.v = in.v; out = dst; return; }
  int nx = cols == 1 ? 1 : n, ny = rows == 1 ? 1 : n;
  std::vector<float> kx = gaussian_kernel(nx, sigma), ky = gaussian_kernel(ny, sigma);
  const int rx = nx / 2, ry = ny / 2;
  Img tmp(rows, cols);
  std::vector<float> ext(cols + 2 * rx);
  for (int r = 0; r < rows; r++) {
    const float *s = in.row(r);
    for (int c = -rx; c < cols + rx; c++) ext[c + rx] = s[clampi(c, 0, cols - 1)];
    float *d = tmp.row(r);
    if (nx <= 5) {
      const float *k = kx.data() + rx;
      for (int c = 0; c < cols; c++) {
        const float *S = ext.data() + c + rx;
        float v = S[0] * k[0];
        for (int j = 1; j <= rx; j++) v = v + (S[-j] + S[j]) * k[j];
        d[c] = v;
      }
    } else {
      for (int c = 0; c < cols; c++) {
        const float *S = ext.data() + c;
        float v = 0.f;
        for (int j = 0; j < nx; j++) v = v + S[j] * kx[j];
        d[c] = v;
      }
    }
  }
  const float *k = ky.data() + ry;
  for (int r = 0; r < rows; r++) {
    float *d = dst.row(r);
    const float *S0 = tmp.row(r);
    for (int c = 0; c < cols; c++) d[c] = k[0] * S0[c] + 0.f;
    for (int j = 1; j <= ry; j++) {
      const float *Sa = tmp.row(clampi(r + j, 0, rows - 1));
      const float *Sb = tmp.row(clampi(r - j, 0, rows - 1));
      for (int c = 0; c < cols; c++) d[c] = d[c] + k[j] * (Sa[c] + Sb[c]);
    }
  }
  out = dst;
}

/* cv::resize(src, dst, Size(0,0), 0.5, 0.5, INTER_LINEAR), affinedetectors/pyramid.cpp:520.
 * dsize = cvRound(dim*0.5) (round half to even); scale is exactly 2 so 2.4.9 switches
 * INTER_LINEAR to the area-fast path: full 2x2 blocks -> ((s00+s01)+s10)+s11)*0.25f,
 * partial blocks at an odd edge -> sum(available)/count. */
static int cv_round_half_even(double v) { return (int)lrint(v); }

void resize_half(const Img &in, Img &out) {
  const int sr = in.rows, sc = in.cols;
  const int dr = cv_round_half_even(sr * 0.5), dc = cv_round_half_even(sc * 0.5);
  Img dst(dr, dc);
  const float scale = 1.f / 4;
  const int dwidth1 = sc / 2;
  for (int dy = 0; dy < dr; dy++) {
    float *D = dst.row(dy);
    int sy0 = dy * 2;
    if (sy0 >= sr) { for (int dx = 0; dx < dc; dx++) D[dx] = 0; continue; }
    int w = (sy0 + 2 <= sr) ? dwidth1 : 0;
    int dx = 0;
    for (; dx < w; dx++) {
      const float *S = in.row(sy0) + dx * 2;
      const float *S1 = in.row(sy0 + 1) + dx * 2;
      float sum = 0;
      sum += S[0] + S[1] + S1[0] + S1[1];
      D[dx] = sum * scale;
    }
    for (; dx < dc; dx++) {
      float sum = 0; int count = 0; int sx0 = dx * 2;
      if (sx0 >= sc) D[dx] = 0;
      for (int sy = 0; sy < 2; sy++) {
        if (sy0 + sy >= sr) break;
        const float *S = in.row(sy0 + sy) + sx0;
        for (int sx = 0; sx < 2; sx++) {
          if (sx0 + sx >= sc) break;
          sum += S[sx]; count++;
        }
      }
      D[dx] = (float)sum / count;
    }
  }
  out = dst;
}

/* ScaleSpaceDetector::HessianResponse, affinedetectors/pyramid.cpp:223-281.
 * The 1-pixel frame is left uninitialised by the reference; the oracle writes 0. */
void hessian_response(const Img &in, float norm, Img &out) {
  const int rows = in.rows, cols = in.cols;
  Img dst(rows, cols);
  const float norm2 = norm * norm;
  for (int r = 1; r < rows - 1; r++) {
    const float *a = in.row(r - 1), *b = in.row(r), *c = in.row(r + 1);
    float *o = dst.row(r);
    for (int x = 1; x < cols - 1; x++) {
      float Lxx = (b[x - 1] - 2 * b[x] + b[x + 1]);
      float Lyy = (a[x] - 2 * b[x] + c[x]);
      float Lxy = (a[x + 1] - a[x - 1] + c[x - 1] - c[x + 1]) / 4.0f;
      o[x] = (Lxx * Lyy - Lxy * Lxy) * norm2;
    }
  }
  out = dst;
}

/* ScaleSpaceDetector::dogResponse, affinedetectors/pyramid.cpp:176-181: input - gaussianBlur(input, norm) -- the blur
 * sigma IS the norm argument (sigma^2 of the level at the call sites :475, :490). */
void dog_response(const Img &in, float norm, Img &out) {
  Img nb;
  gaussian_blur(in, norm, nb);
  Img dst(in.rows, in.cols);
  for (size_t i = 0; i < dst.v.size(); i++) dst.v[i] = in.v[i] - nb.v[i];
  out = dst;
}

/* computeGradient, detectors/helpers.cpp:779-797: central differences, one-sided at the frame */
static void compute_gradient(const Img &img, Img &gx, Img &gy) {
  const int width = img.cols, height = img.rows;
  gx = Img(height, width); gy = Img(height, width);
  for (int r = 0; r < height; r++)
    for (int c = 0; c < width; c++) {
      float xg, yg;
      if (width == 1) xg = 0.f;
      else if (c == 0) xg = img.row(r)[c + 1] - img.row(r)[c];
      else if (c == width - 1) xg = img.row(r)[c] - img.row(r)[c - 1];
      else xg = img.row(r)[c + 1] - img.row(r)[c - 1];
      if (height == 1) yg = 0.f;
      else if (r == 0) yg = img.row(r + 1)[c] - img.row(r)[c];
      else if (r == height - 1) yg = img.row(r)[c] - img.row(r - 1)[c];
      else yg = img.row(r + 1)[c] - img.row(r - 1)[c];
      gx.row(r)[c] = xg; gy.row(r)[c] = yg;
    }
}

/* ScaleSpaceDetector::HarrisResponse, affinedetectors/pyramid.cpp:283-305.  OpenCV 2.4 MatExpr arithmetic as read from the
 * published semantics (PARITY UNPINNED like every cv:: call of the detector): Mat::mul and Mat - Mat are plain f32
 * element operations, `sigmasq * Mat` is convertTo with the scale cast to float, `0.04 * A.mul(A)` folds the scale into
 * multiply(): (0.04f * a) * a. */
void harris_response(const Img &in, float norm, Img &out) {
  const float sigmasq = (float)(0.6 * norm);
  const float sigma = sqrtf(sigmasq);
  Img Lx, Ly;
  compute_gradient(in, Lx, Ly);
  Img xx(in.rows, in.cols), yy(in.rows, in.cols), xy(in.rows, in.cols);
  for (size_t i = 0; i < xx.v.size(); i++) { xx.v[i] = Lx.v[i] * Lx.v[i]; yy.v[i] = Ly.v[i] * Ly.v[i]; xy.v[i] = Lx.v[i] * Ly.v[i]; }
  Img bxx, byy, bxy;
  gaussian_blur(xx, sigma, bxx); gaussian_blur(yy, sigma, byy); gaussian_blur(xy, sigma, bxy);
  Img dst(in.rows, in.cols);
  const float k = 0.04f;
  for (size_t i = 0; i < dst.v.size(); i++) {
    const float dx2 = bxx.v[i] * sigmasq, dy2 = byy.v[i] * sigmasq, dxdy = bxy.v[i] * sigmasq;
    const float s = dx2 + dy2;
    const float a = dx2 * dy2, b = dxdy * dxdy, c = (k * s) * s;
    dst.v[i] = (a - b) - c;
  }
  out = dst;
}

/* interpolateCheckBorders, detectors/helpers.cpp:524-549 */
bool interpolate_check_borders(int orig_w, int orig_h, float ofsx, float ofsy, float a11, float a12,
                               float a21, float a22, int res_w, int res_h) {
  const int width = orig_w - 2, height = orig_h - 2;
  const float halfWidth = (float)ceil((float)res_w / 2.0);
  const float halfHeight = (float)ceil((float)res_h / 2.0);
  const float xs[4] = {-halfWidth, -halfWidth, +halfWidth, +halfWidth};
  const float ys[4] = {-halfHeight, +halfHeight, -halfHeight, +halfHeight};
  for (int i = 0; i < 4; i++) {
    float imx = ofsx + xs[i] * a11 + ys[i] * a12;
    float imy = ofsy + xs[i] * a21 + ys[i] * a22;
    if (floorf(imx) <= 0 || floorf(imy) <= 0 || ceilf(imx) >= width || ceilf(imy) >= height) return true;
  }
  return false;
}

/* interpolate, detectors/helpers.cpp:551-626: affine bilinear resampling with
 * incrementally accumulated f32 sample coordinates. */
bool interpolate(const Img &im, float ofsx, float ofsy, float a11, float a12, float a21, float a22, Img &res) {
  bool ret = false;
  const int width = im.cols - 1, height = im.rows - 1;
  const int halfWidth = res.cols >> 1, halfHeight = res.rows >> 1;
  float *out = res.v.data();
  float rx = ofsx - (float)halfHeight * a12;
  float ry = ofsy - (float)halfHeight * a22;
  const bool touch = interpolate_check_borders(im.cols, im.rows, ofsx, ofsy, a11, a12, a21, a22, res.cols, res.rows);
  for (int j = -halfHeight; j <= halfHeight; ++j) {
    float WX = rx - (float)halfWidth * a11;
    float WY = ry - (float)halfWidth * a21;
    for (int i = -halfWidth; i <= halfWidth; ++i) {
      if (!touch) {
        const int x = (int)WX, y = (int)WY;
        const float wx = WX - (float)x;
        const float *R0 = im.row(y), *R1 = im.row(y + 1);
        const float I1 = wx * (R0[x + 1] - R0[x]) + R0[x];
        *out++ = (WY - y) * (wx * (R1[x + 1] - R1[x]) + R1[x] - I1) + I1;
      } else {
        const int x = (int)floorf(WX), y = (int)floorf(WY);
        if (WX >= 0 && WY >= 0 && x < width && y < height) {
          const float wx = WX - x;
          const float *R0 = im.row(y), *R1 = im.row(y + 1);
          const float I1 = wx * (R0[x + 1] - R0[x]) + R0[x];
          *out++ = (WY - y) * (wx * (R1[x + 1] - R1[x]) + R1[x] - I1) + I1;
        } else {
          *out++ = 0;
          ret = true;
        }
      }
      WX += a11;
      WY += a21;
    }
    rx += a12;
    ry += a22;
  }
  return ret;
}

/* computeGaussMask, detectors/helpers.cpp:411-440 */
void gauss_mask(Img &mask) {
  int size = mask.cols, halfSize = size >> 1;
  float scale = float(halfSize) / 3.0f;
  float scale2 = -2.0f * scale * scale;
  std::vector<float> tmp(halfSize + 1);
  for (int i = 0; i <= halfSize; i++) tmp[i] = expf(float(i * i) / scale2);
  int endSize = int(ceilf(scale * 5.0f) - halfSize);
  for (int i = 1; i < endSize; i++) tmp[halfSize - i] += expf(float((i + halfSize) * (i + halfSize)) / scale2);
  for (int i = 0; i <= halfSize; i++)
    for (int j = 0; j <= halfSize; j++) {
      float v = tmp[i] * tmp[j];
      mask.at(i + halfSize, -j + halfSize) = v;
      mask.at(-i + halfSize, j + halfSize) = v;
      mask.at(i + halfSize, j + halfSize) = v;
      mask.at(-i + halfSize, -j + halfSize) = v;
    }
}

/* computeCircularGaussMask, detectors/helpers.cpp:442-461 */
void circular_gauss_mask(Img &mask, float sigma) {
  int halfSize = mask.cols >> 1;
  float r2 = float(halfSize * halfSize);
  float sigma2 = (sigma == 0) ? 0.9f * r2 : 2 * sigma * sigma;
  float *mp = mask.v.data();
  for (int i = 0; i < mask.rows; i++)
    for (int j = 0; j < mask.cols; j++) {
      float disq = float((i - halfSize) * (i - halfSize) + (j - halfSize) * (j - halfSize));
      *mp++ = (disq < r2) ? expf(-disq / sigma2) : 0;
    }
}

/* photometricallyNormalize, detectors/helpers.cpp:666-715 */
void photometrically_normalize(Img &image, const Img &mask) {
  const size_t n = image.v.size();
  float sum = 0, gsum = 0;
  for (size_t i = 0; i < n; i++)
    if (mask.v[i] > 0) { sum += image.v[i]; gsum++; }
  sum = sum / gsum;
  float var = 0;
  for (size_t i = 0; i < n; i++)
    if (mask.v[i] > 0) var += (sum - image.v[i]) * (sum - image.v[i]);
  var = sqrtf(var / gsum);
  if (var < 0.0001) return;
  float fac = 50.0f / var;
  for (size_t i = 0; i < n; i++) {
    float v = 128 + fac * (image.v[i] - sum);
    if (v > 255) v = 255;
    if (v < 0) v = 0;
    image.v[i] = v;
  }
}

}  // namespace orc

using namespace orc;
extern "C" {
void orc_gray_from_bgr_u8(const uint8_t *bgr, int rows, int cols, float *out) { gray_from_bgr(bgr, rows, cols, out); }
int orc_gaussian_kernel(int n, double sigma, float *out) {
  auto k = gaussian_kernel(n, sigma);
  memcpy(out, k.data(), n * sizeof(float));
  return n;
}
int orc_blur_ksize(float sigma) { return blur_ksize(sigma); }
void orc_gaussian_blur(const float *in, int rows, int cols, float sigma, float *out) {
  Img a(rows, cols, in), b;
  gaussian_blur(a, sigma, b);
  memcpy(out, b.v.data(), b.v.size() * sizeof(float));
}
void orc_resize_half(const float *in, int rows, int cols, float *out, int *orows, int *ocols) {
  Img a(rows, cols, in), b;
  resize_half(a, b);
  *orows = b.rows; *ocols = b.cols;
  if (out) memcpy(out, b.v.data(), b.v.size() * sizeof(float));
}
void orc_hessian_response(const float *in, int rows, int cols, float norm, float *out) {
  Img a(rows, cols, in), b;
  hessian_response(a, norm, b);
  memcpy(out, b.v.data(), b.v.size() * sizeof(float));
}
/* Response() of ScaleSpaceDetector for DetectorType 0 (Hessian), 1 (DoG), 2 (Harris): pyramid.cpp:132-175 */
void orc_response(const float *in, int rows, int cols, int detector_type, float norm, float *out) {
  Img a(rows, cols, in), b;
  if (detector_type == 1) dog_response(a, norm, b);
  else if (detector_type == 2) harris_response(a, norm, b);
  else hessian_response(a, norm, b);
  memcpy(out, b.v.data(), b.v.size() * sizeof(float));
}
int orc_interpolate(const float *im, int rows, int cols, float ofsx, float ofsy, float a11, float a12,
                    float a21, float a22, float *res, int rrows, int rcols) {
  Img a(rows, cols, im), r(rrows, rcols);
  bool t = interpolate(a, ofsx, ofsy, a11, a12, a21, a22, r);
  memcpy(res, r.v.data(), r.v.size() * sizeof(float));
  return t ? 1 : 0;
}
float orc_atan2lut(float y, float x) { return atan2lut(y, x); }
const double *orc_atan_lut(void) { return atan_lut(); }
}
