/*
 * oracle_detect.cpp -- TEST INFRASTRUCTURE ONLY (see oracle.h).
 *
 * CPU restatement of the Hessian-Affine detector of the reference:
 *   ScaleSpaceDetector   detectors/affinedetectors/pyramid.{h,cpp}
 *   AffineShape          detectors/affinedetectors/affine.{h,cpp}
 *   AffineDetector       detectors/affinedetectors/scale-space-detector.{hpp,cpp}
 *   DetectAffineRegions  synth-detection.hpp:93-126
 */
#include "oracle_internal.hpp"

namespace orc {

/* solveLinear3x3, detectors/helpers.cpp:309-368 (f32, partial pivoting) */
static void solve3(float *A, float *b) {
  auto swp = [](float *p, float *q) { float t = *p; *p = *q; *q = t; };
  int i = 0;
  float *pr = A;
  float vp = fabsf(A[0]);
  float tmp = fabsf(A[3]);
  if (tmp > vp) { pr = A + 3; i = 1; vp = tmp; }
  if (fabsf(A[6]) > vp) { pr = A + 6; i = 2; }
  if (pr != A) { swp(pr, A); swp(pr + 1, A + 1); swp(pr + 2, A + 2); swp(b + i, b); }
  vp = A[3] / A[0];
  A[4] -= vp * A[1]; A[5] -= vp * A[2]; b[1] -= vp * b[0];
  vp = A[6] / A[0];
  A[7] -= vp * A[1]; A[8] -= vp * A[2]; b[2] -= vp * b[0];
  if (fabsf(A[4]) < fabsf(A[7])) { swp(A + 7, A + 4); swp(A + 8, A + 5); swp(b + 2, b + 1); }
  vp = A[7] / A[4];
  A[8] -= vp * A[5]; b[2] -= vp * b[1];
  b[2] = (b[2]) / A[8];
  b[1] = (b[1] - A[5] * b[2]) / A[4];
  b[0] = (b[0] - A[2] * b[2] - A[1] * b[1]) / A[0];
}

/* invSqrt, detectors/helpers.cpp:463-502 */
static void inv_sqrt(float &a, float &b, float &c, float &l1, float &l2) {
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

/* getEigenvalues, detectors/helpers.cpp:504-515 */
static bool eigenvalues(float a, float b, float c, float d, float &l1, float &l2) {
  float trace = a + d;
  float delta1 = (trace * trace - 4 * (a * d - b * c));
  if (delta1 < 0) return false;
  float delta = sqrtf(delta1);
  l1 = (trace + delta) / 2.0f;
  l2 = (trace - delta) / 2.0f;
  return true;
}

/* computeGradient, detectors/helpers.cpp:779-797 */
static void gradient(const Img &img, Img &gx, Img &gy) {
  const int w = img.cols, h = img.rows;
  for (int r = 0; r < h; ++r)
    for (int c = 0; c < w; ++c) {
      float xg, yg;
      if (c == 0) xg = img.at(r, c + 1) - img.at(r, c);
      else if (c == w - 1) xg = img.at(r, c) - img.at(r, c - 1);
      else xg = img.at(r, c + 1) - img.at(r, c - 1);
      if (r == 0) yg = img.at(r + 1, c) - img.at(r, c);
      else if (r == h - 1) yg = img.at(r, c) - img.at(r - 1, c);
      else yg = img.at(r + 1, c) - img.at(r - 1, c);
      gx.at(r, c) = xg;
      gy.at(r, c) = yg;
    }
}

/* AffineShape::findAffineShape (AFF_BMBRG_SMM branch), affinedetectors/affine.cpp:26-169.
 * Returns 1 and u[4] when the Baumberg iteration converges. */
int find_affine_shape(const Img &blur, const orc_hessaff_params &p, const Img &mask, float x, float y,
                      float s, float pixelDistance, float *u) {
  float era = 0.0f, erb = 0.0f;
  float u11 = 1.0f, u12 = 0.0f, u21 = 0.0f, u22 = 1.0f, l1 = 1.0f, l2 = 1.0f;
  float lx = x / pixelDistance, ly = y / pixelDistance;
  float ratio = s / (p.affInitialSigma * pixelDistance);
  if (!p.doBaumberg) { u[0] = u11; u[1] = u12; u[2] = u21; u[3] = u22; return 1; }
  const int W = p.smmWindowSize, maskPixels = W * W;
  Img img(W, W), fx(W, W), fy(W, W);
  for (int l = 0; l < p.maxIterations; l++) {
    float a = 0, b = 0, c = 0;
    interpolate(blur, lx, ly, u11 * ratio, u12 * ratio, u21 * ratio, u22 * ratio, img);
    gradient(img, fx, fy);
    for (int i = 0; i < maskPixels; ++i) {
      const float v = mask.v[i], gxx = fx.v[i], gyy = fy.v[i];
      const float gxy = gxx * gyy;
      a += gxx * gxx * v;
      b += gxy * v;
      c += gyy * gyy * v;
    }
    a /= maskPixels; b /= maskPixels; c /= maskPixels;
    inv_sqrt(a, b, c, l1, l2);
    if ((a != a) || (b != b) || (c != c)) break;
    erb = era;
    era = 1.0 - l2 / l1;
    float u11t = u11, u12t = u12;
    u11 = a * u11t + b * u21;
    u12 = a * u12t + b * u22;
    u21 = b * u11t + c * u21;
    u22 = b * u12t + c * u22;
    if (!eigenvalues(u11, u12, u21, u22, l1, l2)) break;
    if ((l1 / l2 > 6) || (l2 / l1 > 6)) break;
    if (era < p.convergenceThreshold && erb < p.convergenceThreshold) {
      u[0] = u11; u[1] = u12; u[2] = u21; u[3] = u22;
      return 1;
    }
  }
  return 0;
}

struct Detector {
  orc_hessaff_params par;
  /* thresholds, affinedetectors/pyramid.h:47-67 */
  double edgeScoreThreshold;
  float finalThreshold, positiveThreshold, negativeThreshold;
  Img octaveMap_;  /* 0/1 flags kept as float for brevity */
  std::vector<unsigned char> octaveMap;
  Img prevBlur, blur, low, cur, high;
  Img mask;
  int octaveIdx = 0, levelIdx = 0;
  std::vector<orc_sskp> sskps;
  std::vector<orc_keypoint> keys;
  bool runAffine = true;

  explicit Detector(const orc_hessaff_params &p) : par(p) {
    edgeScoreThreshold = (p.edgeEigenValueRatio + 1.0f) * (p.edgeEigenValueRatio + 1.0f) / p.edgeEigenValueRatio;
    finalThreshold = p.threshold;
    positiveThreshold = (float)(0.8 * finalThreshold);
    negativeThreshold = -positiveThreshold;
    if (p.detectorType == 0) finalThreshold = p.threshold * p.threshold; /* DET_HESSIAN only, pyramid.h:56-57 */
    if (p.mode != 0) finalThreshold = positiveThreshold = negativeThreshold = 0.0f;
    mask = Img(p.smmWindowSize, p.smmWindowSize);
    gauss_mask(mask);
  }

  static bool is_max(float val, const Img &pix, int row, int col) {
    for (int r = row - 1; r <= row + 1; r++)
      for (int c = col - 1; c <= col + 1; c++)
        if (pix.at(r, c) > val) return false;
    return true;
  }
  static bool is_min(float val, const Img &pix, int row, int col) {
    for (int r = row - 1; r <= row + 1; r++)
      for (int c = col - 1; c <= col + 1; c++)
        if (pix.at(r, c) < val) return false;
    return true;
  }

  /* localizeKeypoint, affinedetectors/pyramid.cpp:308-430 */
  void localize(int r, int c, float curScale, float pixelDistance) {
    const int cols = cur.cols, rows = cur.rows;
    const int r0 = r, c0 = c;
    float b[3] = {0, 0, 0};
    float val = 0;
    int nr = r, nc = c;
    for (int iter = 0; iter < 5; iter++) {
      r = nr; c = nc;
      const float *c0p = cur.row(r - 1), *c1p = cur.row(r), *c2p = cur.row(r + 1);
      const float *l0p = low.row(r - 1), *l1p = low.row(r), *l2p = low.row(r + 1);
      const float *h0p = high.row(r - 1), *h1p = high.row(r), *h2p = high.row(r + 1);
      float dxx = c1p[c - 1] - 2.0f * c1p[c] + c1p[c + 1];
      float dyy = c0p[c] - 2.0f * c1p[c] + c2p[c];
      float dss = l1p[c] - 2.0f * c1p[c] + h1p[c];
      float dxy = 0.25f * (c2p[c + 1] - c2p[c - 1] - c0p[c + 1] + c0p[c - 1]);
      if (iter == 0) {
        float edgeScore = (dxx + dyy) * (dxx + dyy) / (dxx * dyy - dxy * dxy);
        if (edgeScore >= edgeScoreThreshold || edgeScore < 0) return;
      }
      float dxs = 0.25f * (h1p[c + 1] - h1p[c - 1] - l1p[c + 1] + l1p[c - 1]);
      float dys = 0.25f * (h2p[c] - h0p[c] - l2p[c] + l0p[c]);
      float A[9] = {dxx, dxy, dxs, dxy, dyy, dys, dxs, dys, dss};
      float dx = 0.5f * (c1p[c + 1] - c1p[c - 1]);
      float dy = 0.5f * (c2p[c] - c0p[c]);
      float ds = 0.5f * (h1p[c] - l1p[c]);
      b[0] = -dx; b[1] = -dy; b[2] = -ds;
      solve3(A, b);
      if (std::isnan(b[0]) || std::isnan(b[1]) || std::isnan(b[2])) return;
      val = c1p[c] + 0.5f * (dx * b[0] + dy * b[1] + ds * b[2]);
      if (b[0] > 0.6) { if (c < cols - 3) nc++; else return; }
      if (b[1] > 0.6) { if (r < rows - 3) nr++; else return; }
      if (b[0] < -0.6) { if (c > 3) nc--; else return; }
      if (b[1] < -0.6) { if (r > 3) nr--; else return; }
      if (nr == r && nc == c) break;
    }
    if (fabs(b[0]) > 1.5 || fabs(b[1]) > 1.5 || fabs(b[2]) > 1.5 || fabsf(val) < finalThreshold ||
        octaveMap[(size_t)r * cols + c] > 0)
      return;
    octaveMap[(size_t)r * cols + c] = 1;
    float scale = curScale * powf(2.0f, b[2] / par.numberOfScales);
    /* getPointType, pyramid.cpp:66-130; DET_HESSIAN reads the detection-level blur */
    int type;
    if (par.detectorType == 1) type = val < 0 ? 11 /* DOG_BRIGHT */ : 10 /* DOG_DARK */;
    else if (par.detectorType == 2) type = val < 0 ? 31 /* HARRIS_BRIGHT */ : 30 /* HARRIS_DARK */;
    else if (val < 0) type = 2; /* HESSIAN_SADDLE */
    else {
      const float *ptr = blur.row(r) + c;
      float Lxx = (ptr[-1] - 2 * ptr[0] + ptr[1]);
      type = (Lxx < 0) ? 0 /* DARK */ : 1 /* BRIGHT */;
    }
    orc_sskp k;
    k.octave = octaveIdx; k.level = levelIdx; k.r0 = r0; k.c0 = c0; k.r = r; k.c = c; k.type = type; k.pad = 0;
    k.b0 = b[0]; k.b1 = b[1]; k.b2 = b[2]; k.val = val;
    k.x = pixelDistance * (c + b[0]);
    k.y = pixelDistance * (r + b[1]);
    k.s = pixelDistance * scale;
    k.pixelDistance = pixelDistance;
    sskps.push_back(k);
    if (runAffine) {
      /* onKeypointDetected(prevBlur, ...) -> findAffineShape -> keys.push_back,
         scale-space-detector.hpp:48-88 */
      float u[4];
      if (find_affine_shape(prevBlur, par, mask, k.x, k.y, k.s, pixelDistance, u)) {
        orc_keypoint kp;
        memset(&kp, 0, sizeof kp);
        kp.x = k.x; kp.y = k.y; kp.s = k.s;
        kp.a11 = u[0]; kp.a12 = u[1]; kp.a21 = u[2]; kp.a22 = u[3];
        kp.response = val;
        kp.sub_type = type;
        keys.push_back(kp);
      }
    }
  }

  /* findLevelKeypoints, pyramid.cpp:432-452 */
  void find_level(float curScale, float pixelDistance) {
    const int rows = cur.rows, cols = cur.cols, B = par.border;
    for (int r = B; r < rows - B; r++)
      for (int c = B; c < cols - B; c++) {
        const float val = cur.at(r, c);
        if ((val > positiveThreshold && (is_max(val, cur, r, c) && is_max(val, low, r, c) && is_max(val, high, r, c))) ||
            (val < negativeThreshold && (is_min(val, cur, r, c) && is_min(val, low, r, c) && is_min(val, high, r, c))))
          localize(r, c, curScale, pixelDistance);
      }
  }

  /* ScaleSpaceDetector::Response, pyramid.cpp:132-175 */
  void response(const Img &in, float norm, Img &out) {
    if (par.detectorType == 1) dog_response(in, norm, out);
    else if (par.detectorType == 2) harris_response(in, norm, out);
    else hessian_response(in, norm, out);
  }
  /* detectOctaveKeypoints, pyramid.cpp:455-538 */
  void octave(const Img &firstLevel, float pixelDistance, Img &next, float *dumpBlurs, float *dumpResps) {
    octaveMap.assign((size_t)firstLevel.rows * firstLevel.cols, 0);
    float sigmaStep = powf(2.0f, 1.0f / (float)par.numberOfScales);
    float curSigma = par.initialSigma;
    int numLevels = 1;
    blur = firstLevel;
    response(blur, curSigma * curSigma, cur);
    size_t npx = blur.v.size();
    if (dumpBlurs) memcpy(dumpBlurs, blur.v.data(), npx * 4);
    if (dumpResps) memcpy(dumpResps, cur.v.data(), npx * 4);
    for (int i = 1; i < par.numberOfScales + 2; i++) {
      float sigma = curSigma * sqrtf(sigmaStep * sigmaStep - 1.0f);
      Img nextBlur;
      gaussian_blur(blur, sigma, nextBlur);
      sigma = curSigma * sigmaStep;
      response(nextBlur, sigma * sigma, high);
      if (dumpBlurs) memcpy(dumpBlurs + i * npx, nextBlur.v.data(), npx * 4);
      if (dumpResps) memcpy(dumpResps + i * npx, high.v.data(), npx * 4);
      numLevels++;
      if (numLevels == 3) {
        levelIdx = i - 1;
        find_level(curSigma, pixelDistance);
        numLevels--;
      }
      if (i == par.numberOfScales) resize_half(nextBlur, next);
      prevBlur = blur;
      blur = nextBlur;
      low = cur;
      cur = high;
      curSigma *= sigmaStep;
    }
  }

  /* detectPyramidKeypoints, pyramid.cpp:540-573 */
  void detect(const Img &image) {
    float curSigma = 0.5f, pixelDistance = 1.0f;
    Img firstLevel = image;
    if (par.initialSigma > curSigma) {
      float sigma = sqrtf(par.initialSigma * par.initialSigma - curSigma * curSigma);
      Img t;
      gaussian_blur(firstLevel, sigma, t);
      firstLevel = t;
    }
    int minSize = 2 * par.border + 2;
    octaveIdx = 0;
    while (firstLevel.rows > minSize && firstLevel.cols > minSize) {
      Img next;
      octave(firstLevel, pixelDistance, next, nullptr, nullptr);
      pixelDistance *= 2.0;
      firstLevel = next;
      octaveIdx++;
    }
  }

  /* prepareKeysForExport, scale-space-detector.hpp:118-198 (std::sort is the
     same libstdc++ introsort the reference calls) */
  void prepare_export() {
    if (keys.empty()) return;
    if (par.mode == 0) return;
    std::sort(keys.begin(), keys.end(),
              [](orc_keypoint k1, orc_keypoint k2) { return fabs(k1.response) > fabs(k2.response); });
    double maxResponse = fabs(keys[0].response);
    int regNumber = (int)keys.size();
    auto cmp = [](const orc_keypoint &k1, const orc_keypoint &k2) { return fabs(k1.response) > fabs(k2.response); };
    switch (par.mode) {
      case 1: { /* RELATIVE_TH */
        orc_keypoint t = keys[0];
        t.response = (float)(maxResponse * par.rel_threshold);
        auto low = std::lower_bound(keys.begin(), keys.end(), t, cmp);
        keys.resize(low - keys.begin());
        break;
      }
      case 2: { /* FIXED_REG_NUMBER */
        int n = par.reg_number;
        if (par.doBaumberg) n = (int)floor(3.0 * (double)n);
        if ((n < regNumber) && (n >= 0)) keys.resize(n);
        break;
      }
      case 3: { /* RELATIVE_REG_NUMBER */
        int n = (int)floor(par.rel_reg_number * (double)keys.size());
        keys.resize(n);
        break;
      }
      case 4: { /* NOT_LESS_THAN_REGIONS */
        orc_keypoint t = keys[0];
        t.response = par.threshold;
        auto low = std::lower_bound(keys.begin(), keys.end(), t, cmp);
        int fix = (int)std::distance(keys.begin(), low);
        if (fix < par.reg_number) keys.resize(std::min(par.reg_number, regNumber));
        else keys.resize(std::min(fix, regNumber));
        break;
      }
      default: break;
    }
    if (par.mode == 2 && (int)keys.size() > par.reg_number) keys.resize(par.reg_number);
  }
};

/* rectifyTransformation, synth-detection.cpp:46-55 */
void rectify(double &a11, double &a12, double &a21, double &a22) {
  double a = a11, b = a12, c = a21, d = a22;
  double det = sqrt(fabs(a * d - b * c));
  double b2a2 = sqrt(b * b + a * a);
  a11 = b2a2 / det;
  a12 = 0;
  a21 = (d * b + c * a) / (b2a2 * det);
  a22 = det / b2a2;
}

}  // namespace orc

using namespace orc;
extern "C" {

void orc_default_hessaff_params(orc_hessaff_params *p) {
  /* build/config_iter_mods_cviu.ini:13-27 + structures.hpp:141-160 defaults */
  p->threshold = 5.3333f;
  p->mode = 0;
  p->reg_number = 2000;
  p->rel_threshold = -1;
  p->rel_reg_number = -1;
  p->numberOfScales = 3;
  p->initialSigma = 1.6f;
  p->edgeEigenValueRatio = 10.0;
  p->border = 5;
  p->maxIterations = 16;
  p->convergenceThreshold = 0.05f;
  p->smmWindowSize = 19;
  p->affInitialSigma = 1.6f;
  p->doBaumberg = 1;
  p->detectorType = 0;
}

void orc_octave_levels(const float *first, int rows, int cols, const orc_hessaff_params *p, float *blurs,
                       float *resps) {
  Detector d(*p);
  d.runAffine = false;
  Img f(rows, cols, first), next;
  d.octave(f, 1.0f, next, blurs, resps);
}

int orc_detect_scalespace(const float *img, int rows, int cols, const orc_hessaff_params *p, orc_sskp *out,
                          int cap) {
  Detector d(*p);
  d.runAffine = false;
  d.detect(Img(rows, cols, img));
  int n = (int)d.sskps.size();
  for (int i = 0; i < n && i < cap; i++) out[i] = d.sskps[i];
  return n;
}

/* DetectAffineKeypoints, scale-space-detector.cpp:43-85 */
int orc_detect_hessaff(const float *img, int rows, int cols, const orc_hessaff_params *p, double tilt,
                       double zoom, orc_keypoint *out, int cap) {
  orc_hessaff_params q = *p;
  if ((tilt > 2.0) || (zoom < 0.5)) q.reg_number = (int)floor(zoom * (double)q.reg_number / tilt);
  Detector d(q);
  d.detect(Img(rows, cols, img));
  d.prepare_export();
  int n = (int)d.keys.size();
  for (int i = 0; i < n && i < cap; i++) out[i] = d.keys[i];
  return n;
}

int orc_find_affine_shape(const float *blur, int rows, int cols, const orc_hessaff_params *p, float x, float y,
                          float s, float pixelDistance, float *u) {
  Img mask(p->smmWindowSize, p->smmWindowSize);
  gauss_mask(mask);
  return find_affine_shape(Img(rows, cols, blur), *p, mask, x, y, s, pixelDistance, u);
}

/* DetectAffineRegions<>, synth-detection.hpp:93-126 */
int orc_detect_affine_regions(const orc_keypoint *kps, int n, int img_id, int det_type, orc_region *out) {
  for (int i = 0; i < n; i++) {
    orc_keypoint k = kps[i];
    orc_region r;
    memset(&r, 0, sizeof r);
    r.img_id = img_id; r.img_reproj_id = 0; r.type = det_type; r.id = i;
    r.det_kp.s = k.s * sqrt(fabs(k.a11 * k.a22 - k.a12 * k.a21));
    rectify(k.a11, k.a12, k.a21, k.a22);
    r.det_kp.x = k.x; r.det_kp.y = k.y;
    r.det_kp.a11 = k.a11; r.det_kp.a12 = k.a12; r.det_kp.a21 = k.a21; r.det_kp.a22 = k.a22;
    r.det_kp.response = k.response;
    r.det_kp.sub_type = k.sub_type;
    out[i] = r;
  }
  return n;
}
}
