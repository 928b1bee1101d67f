// tables.cpp -- host-side precomputation for libmodsx (filter taps, masks, LUTs, small algebra).
// Transcendentals (exp, atan, pow, cos, sin) are evaluated on the host with the same libm the
// reference's CPU code calls; the device only ever sees the resulting tables.
#include <math.h>
#include <stdio.h>
#include "engine.hpp"

namespace mx {

// cv::getGaussianKernel(n, sigma, CV_32F) (OpenCV 2.4.9), the kernel gaussianBlur() asks for
// (detectors/helpers.cpp:717-731): taps from exp() in f64, stored f32, normalised by the f64 sum
// of the f32 taps.
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

// detectors/helpers.cpp:720-721
int blur_ksize(float sigma) {
  int size = (int)(2.0 * 3.0 * sigma + 1.0);
  if (size % 2 == 0) size++;
  return size;
}

// computeGaussMask, detectors/helpers.cpp:411-440
void gauss_mask(float *mask, int size) {
  int halfSize = size >> 1;
  float scale = float(halfSize) / 3.0f;
  float scale2 = -2.0f * scale * scale;
  std::vector<float> tmp(halfSize + 1);
  for (int i = 0; i <= halfSize; i++) tmp[i] = expf(float(i * i) / scale2);
  int endSize = int(ceilf(scale * 5.0f) - halfSize);
  for (int i = 1; i < endSize; i++) tmp[halfSize - i] += expf(float((i + halfSize) * (i + halfSize)) / scale2);
  for (int i = 0; i <= halfSize; i++)
    for (int j = 0; j <= halfSize; j++) {
      float v = tmp[i] * tmp[j];
      mask[(i + halfSize) * size + (-j + halfSize)] = v;
      mask[(-i + halfSize) * size + (j + halfSize)] = v;
      mask[(i + halfSize) * size + (j + halfSize)] = v;
      mask[(-i + halfSize) * size + (-j + halfSize)] = v;
    }
}

// computeCircularGaussMask, detectors/helpers.cpp:442-461
void circular_gauss_mask(float *mask, int size, float sigma) {
  int halfSize = size >> 1;
  float r2 = float(halfSize * halfSize);
  float sigma2 = (sigma == 0) ? 0.9f * r2 : 2 * sigma * sigma;
  for (int i = 0; i < size; i++)
    for (int j = 0; j < size; j++) {
      float disq = float((i - halfSize) * (i - halfSize) + (j - halfSize) * (j - halfSize));
      mask[i * size + j] = (disq < r2) ? expf(-disq / sigma2) : 0;
    }
}

// ATAN_LUT, detectors/helpers.cpp:30-72: atan(i/255) to 10 decimals; entries 32, 83 and 100 of the
// reference table deviate from that rule and are reproduced as they stand there.
const double *atan_lut_host() {
  static double lut[256];
  static bool ready = false;
  if (!ready) {
    for (int i = 0; i < 256; i++) {
      char buf[64];
      snprintf(buf, sizeof buf, "%.10f", atan(i / 255.0));
      lut[i] = strtod(buf, nullptr);
    }
    lut[32] = 0.1248376255;
    lut[83] = 0.3146752558;
    lut[100] = 0.3737268255;
    ready = true;
  }
  return lut;
}

// cv::invert(3x3, DECOMP_LU): closed-form adjugate / determinant (OpenCV 2.4.9 lapack.cpp)
bool invert3(const double *S, double *t) {
  double d = S[0] * (S[4] * S[8] - S[5] * S[7]) - S[1] * (S[3] * S[8] - S[5] * S[6]) + S[2] * (S[3] * S[7] - S[4] * S[6]);
  if (d == 0.) { for (int i = 0; i < 9; i++) t[i] = 0; return false; }
  d = 1. / d;
  t[0] = (S[4] * S[8] - S[5] * S[7]) * d; t[1] = (S[2] * S[7] - S[1] * S[8]) * d; t[2] = (S[1] * S[5] - S[2] * S[4]) * d;
  t[3] = (S[5] * S[6] - S[3] * S[8]) * d; t[4] = (S[0] * S[8] - S[2] * S[6]) * d; t[5] = (S[2] * S[3] - S[0] * S[5]) * d;
  t[6] = (S[3] * S[7] - S[4] * S[6]) * d; t[7] = (S[1] * S[6] - S[0] * S[7]) * d; t[8] = (S[0] * S[4] - S[1] * S[3]) * d;
  return true;
}

// rectifyTransformation, synth-detection.cpp:46-55
void rectify(double &a11, double &a12, double &a21, double &a22) {
  double a = a11, b = a12, c = a21, d = a22;
  double det = sqrt(fabs(a * d - b * c));
  double b2a2 = sqrt(b * b + a * a);
  a11 = b2a2 / det;
  a12 = 0;
  a21 = (d * b + c * a) / (b2a2 * det);
  a22 = det / b2a2;
}

}  // namespace mx
