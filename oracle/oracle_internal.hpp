/* oracle_internal.hpp -- TEST INFRASTRUCTURE ONLY (see oracle.h). */
#ifndef MODSX_ORACLE_INTERNAL_HPP
#define MODSX_ORACLE_INTERNAL_HPP
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <string>
#include <vector>
#include "oracle.h"

namespace orc {

struct Img {
  int rows = 0, cols = 0;
  std::vector<float> v;
  Img() {}
  Img(int r, int c) : rows(r), cols(c), v((size_t)r * c, 0.f) {}
  Img(int r, int c, const float *src) : rows(r), cols(c), v(src, src + (size_t)r * c) {}
  float *row(int r) { return v.data() + (size_t)r * cols; }
  const float *row(int r) const { return v.data() + (size_t)r * cols; }
  float &at(int r, int c) { return v[(size_t)r * cols + c]; }
  float at(int r, int c) const { return v[(size_t)r * cols + c]; }
};

const double *atan_lut();
float atan2lut(float y, float x);
void gray_from_bgr(const uint8_t *bgr, int rows, int cols, float *out);
std::vector<float> gaussian_kernel(int n, double sigma);
int blur_ksize(float sigma);
void gaussian_blur(const Img &in, float sigma, Img &out);
void resize_half(const Img &in, Img &out);
void hessian_response(const Img &in, float norm, Img &out);
void dog_response(const Img &in, float norm, Img &out);
void harris_response(const Img &in, float norm, Img &out);
bool interpolate_check_borders(int orig_w, int orig_h, float ofsx, float ofsy, float a11, float a12,
                               float a21, float a22, int res_w, int res_h);
bool interpolate(const Img &im, float ofsx, float ofsy, float a11, float a12, float a21, float a22, Img &res);
void gauss_mask(Img &mask);
void circular_gauss_mask(Img &mask, float sigma);
void photometrically_normalize(Img &image, const Img &mask);

}  // namespace orc
#endif
