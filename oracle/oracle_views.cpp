/*
 * oracle_views.cpp -- TEST INFRASTRUCTURE ONLY (see oracle.h).
 *
 * CPU restatement of view synthesis:
 *   GenerateSynthImageCorr   synth-detection.cpp:236-430
 *   SetVSPars                synth-detection.cpp:103-234
 * with the two OpenCV 2.4.9 routines it calls restated from their published algorithm:
 *   cv::warpAffine(INTER_LINEAR, BORDER_CONSTANT 128): forward matrix inverted in f64, source
 *     coordinates in 1/1024 fixed point (cvRound), rounded to 1/32 pixel, bilinear weights from
 *     the 32x32 float table (products of multiples of 1/32), 4 taps accumulated in f32;
 *   cv::GaussianBlur(Size(kx,ky), sx, sy) with the default BORDER_REFLECT_101.
 */
#include "oracle_internal.hpp"

namespace orc {

static int cv_round(double v) { return (int)lrint(v); }

void warp_affine(const Img &src, Img &dst, const double *Min, float cval) {
  double M[6];
  for (int i = 0; i < 6; i++) M[i] = Min[i];
  {
    double D = M[0] * M[4] - M[1] * M[3];
    D = D != 0 ? 1. / D : 0;
    double A11 = M[4] * D, A22 = M[0] * D;
    M[0] = A11; M[1] *= -D;
    M[3] *= -D; M[4] = A22;
    double b1 = -M[0] * M[2] - M[1] * M[5];
    double b2 = -M[3] * M[2] - M[4] * M[5];
    M[2] = b1; M[5] = b2;
  }
  const int AB_BITS = 10, AB_SCALE = 1 << AB_BITS, INTER_BITS = 5, TAB = 32;
  const int round_delta = AB_SCALE / TAB / 2;
  const int w = dst.cols, h = dst.rows, sw = src.cols, sh = src.rows;
  std::vector<int> adelta(w), bdelta(w);
  for (int x = 0; x < w; x++) { adelta[x] = cv_round(M[0] * x * AB_SCALE); bdelta[x] = cv_round(M[3] * x * AB_SCALE); }
  for (int y = 0; y < h; y++) {
    int X0 = cv_round((M[1] * y + M[2]) * AB_SCALE) + round_delta;
    int Y0 = cv_round((M[4] * y + M[5]) * AB_SCALE) + round_delta;
    float *D = dst.row(y);
    for (int x = 0; x < w; x++) {
      int X = (X0 + adelta[x]) >> (AB_BITS - INTER_BITS);
      int Y = (Y0 + bdelta[x]) >> (AB_BITS - INTER_BITS);
      int sx = X >> INTER_BITS, sy = Y >> INTER_BITS;
      sx = sx < -32768 ? -32768 : (sx > 32767 ? 32767 : sx);
      sy = sy < -32768 ? -32768 : (sy > 32767 ? 32767 : sy);
      const float fx = (X & (TAB - 1)) * (1.f / TAB), fy = (Y & (TAB - 1)) * (1.f / TAB);
      const float w0 = (1.f - fy) * (1.f - fx), w1 = (1.f - fy) * fx, w2 = fy * (1.f - fx), w3 = fy * fx;
      if ((unsigned)sx < (unsigned)std::max(sw - 1, 0) && (unsigned)sy < (unsigned)std::max(sh - 1, 0)) {
        const float *S = src.row(sy) + sx;
        D[x] = S[0] * w0 + S[1] * w1 + S[sw] * w2 + S[sw + 1] * w3;
      } else if (sx >= sw || sx + 1 < 0 || sy >= sh || sy + 1 < 0) {
        D[x] = cval;
      } else {
        auto tap = [&](int xx, int yy) { return (xx >= 0 && xx < sw && yy >= 0 && yy < sh) ? src.at(yy, xx) : cval; };
        float v0 = tap(sx, sy), v1 = tap(sx + 1, sy), v2 = tap(sx, sy + 1), v3 = tap(sx + 1, sy + 1);
        D[x] = v0 * w0 + v1 * w1 + v2 * w2 + v3 * w3;
      }
    }
  }
}

static int reflect101(int p, int n) {
  if (n == 1) return 0;
  while (p < 0 || p >= n) { if (p < 0) p = -p; else p = 2 * n - 2 - p; }
  return p;
}

void gaussian_blur_xy(const Img &in, int kx, int ky, double sx, double sy, Img &out) {
  const int rows = in.rows, cols = in.cols;
  if (rows == 1) ky = 1;
  if (cols == 1) kx = 1;
  Img dst(rows, cols);
  if (kx == 1 && ky == 1) { dst.v = in.v; out = dst; return; }
  sx = std::max(sx, 0.); sy = std::max(sy, 0.);
  std::vector<float> KX = gaussian_kernel(kx, sx);
  std::vector<float> KY = (ky == kx && fabs(sx - sy) < 2.220446049250313e-16) ? KX : gaussian_kernel(ky, sy);
  const int rx = kx / 2, ry = ky / 2;
  Img tmp(rows, cols);
  std::vector<float> ext(cols + 2 * rx);
  for (int r = 0; r < rows; r++) {
    const float *s = in.row(r);
    for (int c = -rx; c < cols + rx; c++) ext[c + rx] = s[reflect101(c, cols)];
    float *d = tmp.row(r);
    if (kx <= 5) {
      const float *k = KX.data() + rx;
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
        for (int j = 0; j < kx; j++) v = v + S[j] * KX[j];
        d[c] = v;
      }
    }
  }
  const float *k = KY.data() + ry;
  for (int r = 0; r < rows; r++) {
    float *d = dst.row(r);
    const float *S0 = tmp.row(r);
    for (int c = 0; c < cols; c++) d[c] = k[0] * S0[c] + 0.f;
    for (int j = 1; j <= ry; j++) {
      const float *Sa = tmp.row(reflect101(r + j, rows));
      const float *Sb = tmp.row(reflect101(r - j, rows));
      for (int c = 0; c < cols; c++) d[c] = d[c] + k[j] * (Sa[c] + Sb[c]);
    }
  }
  out = dst;
}

/* GenerateSynthImageCorr, synth-detection.cpp:236-430 (gray input, AREA_INTERP undefined) */
bool synth_view(const Img &gray, orc_view v, Img &out, double *H, bool sizeOnly) {
  double tilt = v.tilt;
  const double phi = v.phi, zoom = v.zoom, InitSigma = v.InitSigma;
  int zoomed = 0;
  bool vertical_tilt = false;
  if (tilt < 0) { tilt = -tilt; vertical_tilt = true; }
  if (fabs(zoom - 1.0f) >= 0.05) zoomed = 1;
  const int w = gray.cols, h = gray.rows;
  int wS1 = (int)(w * zoom), hS1 = (int)(h * zoom);
  if ((fabs(tilt - 1.) <= 0.1) && (fabs(phi) <= 0.2) && (fabs(zoom - 1.) <= 0.1)) {
    const double E[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    for (int i = 0; i < 9; i++) H[i] = E[i];
    if (sizeOnly) { out.rows = h; out.cols = w; } else out = gray;
    return true;
  }
  double d, d2, w_new, h_new;
  double kV = 1., kH = 1.;
  if (zoomed) { kV = (double)w / (double)wS1; kH = (double)h / (double)hS1; }
  const bool q1 = (phi >= 0) && (phi < M_PI / 2);
  if (vertical_tilt) {
    if (q1) {
      w_new = floor((0.5 + cos(phi) * w + sin(phi) * h) / (kH));
      h_new = floor((0.5 + sin(phi) * w + cos(phi) * h) / (tilt * kV));
      H[0] = cos(phi) / kH; H[1] = sin(phi) / kH; H[2] = 0;
      H[3] = -sin(phi) / (tilt * kV); H[4] = cos(phi) / (tilt * kV); H[5] = floor(0.5 + sin(phi) * w / (tilt * kV));
    } else {
      w_new = floor((0.5 - cos(phi) * w + sin(phi) * h) / (kH));
      h_new = floor((0.5 + sin(phi) * w - cos(phi) * h) / (tilt * kV));
      d = -floor(cos(phi) * w / kH);
      d2 = floor(0.5 + (sin(phi) * w - cos(phi) * h) / (tilt * kV));
      H[0] = cos(phi) / kH; H[1] = sin(phi) / kH; H[2] = d;
      H[3] = -sin(phi) / (tilt * kV); H[4] = cos(phi) / (tilt * kV); H[5] = d2;
    }
  } else {
    if (q1) {
      w_new = floor((0.5 + cos(phi) * w + sin(phi) * h) / (tilt * kH));
      h_new = floor((0.5 + sin(phi) * w + cos(phi) * h) / (kV));
      H[0] = cos(phi) / (tilt * kH); H[1] = sin(phi) / (tilt * kH); H[2] = 0;
      H[3] = -sin(phi) / kV; H[4] = cos(phi) / kV; H[5] = floor(0.5 + sin(phi) * w / kV);
    } else {
      w_new = floor((0.5 - cos(phi) * w + sin(phi) * h) / (tilt * kH));
      h_new = floor((0.5 + sin(phi) * w - cos(phi) * h) / (kV));
      d = -floor(cos(phi) * w / (tilt * kH));
      d2 = floor(0.5 + (sin(phi) * w - cos(phi) * h) / kV);
      H[0] = cos(phi) / (tilt * kH); H[1] = sin(phi) / (tilt * kH); H[2] = d;
      H[3] = -sin(phi) / kV; H[4] = cos(phi) / kV; H[5] = d2;
    }
  }
  H[6] = 0; H[7] = 0; H[8] = 1;
  if (sizeOnly) { out.rows = (int)h_new; out.cols = (int)w_new; return false; }
  double sigma_aa_2 = zoomed ? InitSigma / (4.0 * zoom) : InitSigma / 2.0;
  double sigma_aa = InitSigma * tilt / (2.0 * zoom);
  double sigma_x = vertical_tilt ? sigma_aa_2 : sigma_aa, sigma_y = vertical_tilt ? sigma_aa : sigma_aa_2;
  int w_rot, h_rot;
  double R[6];
  if (q1) {
    w_rot = floor((0.5 + cos(phi) * w + sin(phi) * h));
    h_rot = floor((0.5 + sin(phi) * w + cos(phi) * h));
    R[0] = cos(phi); R[1] = sin(phi); R[2] = 0;
    R[3] = -sin(phi); R[4] = cos(phi); R[5] = floor(0.5 + sin(phi) * w);
  } else {
    w_rot = floor((0.5 - cos(phi) * w + sin(phi) * h));
    h_rot = floor((0.5 + sin(phi) * w - cos(phi) * h));
    d = -floor(cos(phi) * w);
    d2 = floor(0.5 + (sin(phi) * w - cos(phi) * h));
    R[0] = cos(phi); R[1] = sin(phi); R[2] = d;
    R[3] = -sin(phi); R[4] = cos(phi); R[5] = d2;
  }
  Img temp(h_rot, w_rot);
  warp_affine(gray, temp, R, 128.f);
  if (v.doBlur) {
    int kx = floor(2.0 * 3.0 * sigma_x + 1.0);
    if (kx % 2 == 0) kx++;
    if (kx < 3) kx = 3;
    int ky = floor(2.0 * 3.0 * sigma_y + 1.0);
    if (ky % 2 == 0) ky++;
    if (ky < 3) ky = 3;
    Img t2;
    gaussian_blur_xy(temp, kx, ky, sigma_x, sigma_y, t2);
    temp = t2;
  }
  double Wz[6];
  if (vertical_tilt) { Wz[0] = 1.0 / kH; Wz[4] = 1.0 / (tilt * kV); }
  else { Wz[0] = 1.0 / (tilt * kH); Wz[4] = 1.0 / kV; }
  Wz[1] = 0; Wz[2] = 0; Wz[3] = 0; Wz[5] = 0;
  out = Img((int)h_new, (int)w_new);
  warp_affine(temp, out, Wz, 128.f);
  return false;
}

}  // namespace orc

using namespace orc;
extern "C" {

void orc_warp_affine(const float *src, int rows, int cols, const double *M6, float *dst, int drows, int dcols,
                     float border) {
  Img s(rows, cols, src), d(drows, dcols);
  warp_affine(s, d, M6, border);
  memcpy(dst, d.v.data(), d.v.size() * 4);
}

void orc_gaussian_blur_xy(const float *in, int rows, int cols, int kx, int ky, double sx, double sy, float *out) {
  Img a(rows, cols, in), b;
  gaussian_blur_xy(a, kx, ky, sx, sy, b);
  memcpy(out, b.v.data(), b.v.size() * 4);
}

int orc_synth_view(const float *gray, int rows, int cols, const orc_view *v, float *out, int *orows, int *ocols,
                   double *H9) {
  Img g(rows, cols, gray), o;
  bool ident = synth_view(g, *v, o, H9, out == nullptr);
  *orows = o.rows; *ocols = o.cols;
  if (out) memcpy(out, o.v.data(), o.v.size() * 4);
  return ident ? 1 : 0;
}

int orc_set_vs_pars(const double *scale_set, int ns, const double *tilt_set, int nt, double phi_base, double InitSigma,
                    int doBlur, orc_view *par, int cap, orc_view *prev, int *nprev, int cap_prev) {
  const double eps1 = 0.01;
  std::vector<orc_view> tmp;
  if (ns == 0 || nt == 0) {
    orc_view t; t.phi = 0; t.tilt = 0; t.zoom = 0; t.InitSigma = InitSigma; t.doBlur = 0;
    tmp.push_back(t);
  }
  for (int sc = 0; sc < ns; sc++)
    for (int t = 0; t < nt; t++) {
      if (fabs(tilt_set[t] - 1) > eps1) {
        int n_rot1 = floor(180.0 * tilt_set[t] / phi_base);
        double delta_phi = M_PI / n_rot1;
        if (n_rot1 < 0) {
          n_rot1 = 1; delta_phi = 0;
          orc_view v; v.phi = 0; v.tilt = -tilt_set[t]; v.zoom = scale_set[sc]; v.InitSigma = InitSigma; v.doBlur = doBlur;
          tmp.push_back(v);
        }
        for (int r = 0; r < n_rot1; r++) {
          orc_view v; v.phi = delta_phi * r; v.tilt = tilt_set[t]; v.zoom = scale_set[sc]; v.InitSigma = InitSigma;
          v.doBlur = doBlur;
          tmp.push_back(v);
        }
      } else {
        orc_view v; v.phi = 0; v.tilt = tilt_set[t]; v.zoom = scale_set[sc]; v.InitSigma = InitSigma; v.doBlur = doBlur;
        tmp.push_back(v);
      }
    }
  int n = 0;
  std::vector<orc_view> added;
  for (size_t i = 0; i < tmp.size(); i++) {
    bool uniq = true;
    for (int j = 0; j < *nprev; j++)
      if ((fabs(tmp[i].zoom - prev[j].zoom) <= eps1) && (fabs(tmp[i].tilt - prev[j].tilt) <= eps1) &&
          (fabs(tmp[i].phi - prev[j].phi) <= eps1)) { uniq = false; break; }
    if (uniq) { if (n < cap) par[n] = tmp[i]; n++; added.push_back(tmp[i]); }
  }
  for (const orc_view &v : added) if (*nprev < cap_prev) prev[(*nprev)++] = v;
  return n;
}
}
