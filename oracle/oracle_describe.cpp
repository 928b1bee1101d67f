/*
 * oracle_describe.cpp -- TEST INFRASTRUCTURE ONLY (see oracle.h).
 *
 * CPU restatement of
 *   DetectOrientation / EstimateDominantAnglesFunctor   synth-detection.cpp:746-919
 *   ReprojectRegions (identity / affine H)               synth-detection.cpp:541-616
 *   DescribeRegions<>                                    synth-detection.hpp:169-255
 *   SIFTDescriptor (SIFT / RootSIFT)                     matching/siftdesc.{h,cpp}
 */
#include "oracle_internal.hpp"

namespace orc {

static const double K_SIGMA = 2 * 3.0 * sqrt(3.0); /* synth-detection.cpp:28 */

/* computeGradientMagnitudeAndOrientation, detectors/helpers.cpp:840-863 (interior only) */
static void grad_mag_ori(const Img &img, Img &mag, Img &ori) {
  for (int r = 1; r < img.rows - 1; ++r)
    for (int c = 1; c < img.cols - 1; ++c) {
      float xg = img.at(r, c + 1) - img.at(r, c - 1);
      float yg = img.at(r + 1, c) - img.at(r - 1, c);
      mag.at(r, c) = sqrtf(xg * xg + yg * yg);
      ori.at(r, c) = atan2lut(yg, xg);
    }
}

/* EstimateDominantAnglesFunctor::operator(), synth-detection.cpp:763-838 */
int dominant_angles(const Img &img, int doHalfSIFT, double max_th, int maxAngles, std::vector<float> &angles1) {
  angles1.clear();
  if (maxAngles == 0) return 0;
  const int pS = img.rows;
  const int bins = 36;
  float hist[bins + 1];
  std::vector<float> peak_values;
  for (int i = 0; i < bins; i++) hist[i] = 0.0f;
  hist[bins] = 0.0f; /* slot 36 is never read by the reference */
  Img gmag(pS, pS), gori(pS, pS), orimask(pS, pS);
  circular_gauss_mask(orimask, pS / 3.0f);
  grad_mag_ori(img, gmag, gori);
  const float *maskptr = orimask.row(1), *pmag = gmag.row(1), *pori = gori.row(1);
  const int maskPixels = pS * (pS - 2);
  for (int i = 0; i < maskPixels; ++i) {
    if (maskptr[i] > 0 && pmag[i] > 1.0) {
      int bin = (int)(bins * (pori[i] / float(M_PI) + 1.0f) / 2.0f);
      hist[bin] += pmag[i] * maskptr[i];
    }
  }
  for (int it = 0; it < 6; it++) { /* smoothCircularBuffer<36> */
    float first = hist[0], prev = hist[bins - 1];
    for (int i = 0; i < bins - 1; i++) {
      float curv = hist[i];
      hist[i] = prev + curv + hist[i + 1];
      prev = curv;
    }
    hist[bins - 1] = prev + hist[bins - 1] + first;
  }
  float thresh = 0.0;
  for (int i = 0; i < bins; i++) if (hist[i] > thresh) thresh = hist[i];
  thresh *= max_th;
  if (doHalfSIFT) {
    int hb = bins / 2;
    for (int i = 0; i < hb; i++) { hist[i] += hist[i + hb]; hist[i + hb] = 0; }
  }
  auto add_peak = [&](int a, int b, int c) {
    if (hist[b] >= thresh && hist[b] > hist[a] && hist[b] > hist[c]) {
      float pp = (hist[a] - hist[c]) / (hist[a] - 2.0f * hist[b] + hist[c]) / 2.0f;
      angles1.push_back(2.0f * float(M_PI) * (b + 0.5f + pp) / bins - float(M_PI));
      peak_values.push_back(hist[b]);
    }
  };
  add_peak(bins - 1, 0, 1);
  for (int i = 1; i < bins - 1; i++) add_peak(i - 1, i, i + 1);
  add_peak(bins - 2, bins - 1, 0);
  if (maxAngles == -1) maxAngles = 100000000;
  maxAngles = std::min(maxAngles, (int)peak_values.size());
  if (maxAngles > 0) {
    std::vector<float> tmp;
    for (int a = 0; a < maxAngles; a++) {
      if (peak_values[a] >= thresh) tmp.push_back(angles1[a]);
      else break;
    }
    angles1 = tmp;
  } else angles1.clear();
  return (int)angles1.size();
}

/* SIFTDescriptor: precomputeBinsAndWeights + gradients + samplePatch + norms,
   matching/siftdesc.cpp:22-131, 136-278, 290-379 */
struct Sift {
  int patchSize = 41, spatialBins = 4, orientationBins = 8;
  double maxBinValue = 0.2;
  std::vector<int> bin0, bin1;
  std::vector<double> w0, w1;
  Img mask;
  std::vector<double> vec;
  Sift(double maxBin) : maxBinValue(maxBin) {
    mask = Img(patchSize, patchSize);
    circular_gauss_mask(mask, 0);
    vec.resize(spatialBins * spatialBins * orientationBins);
    int halfSize = patchSize >> 1;
    float step = float(spatialBins + 1) / (2 * halfSize);
    bin0.resize(patchSize); bin1.resize(patchSize); w0.resize(patchSize); w1.resize(patchSize);
    for (int i = 0; i < patchSize; i++) {
      float x = step * i;
      int xi = (int)(x);
      bin0[i] = xi - 1;
      bin1[i] = xi;
      w1[i] = x - xi;
      w0[i] = 1.0f - w1[i];
      if (bin0[i] < 0) { bin0[i] = 0; w0[i] = 0; }
      if (bin0[i] >= spatialBins) { bin0[i] = spatialBins - 1; w0[i] = 0; }
      if (bin1[i] < 0) { bin1[i] = 0; w1[i] = 0; }
      if (bin1[i] >= spatialBins) { bin1[i] = spatialBins - 1; w1[i] = 0; }
      bin0[i] *= orientationBins;
      bin1[i] *= orientationBins;
    }
  }
  static double normalize(std::vector<double> &v) {
    double len = 0.0;
    for (size_t i = 0; i < v.size(); i += 4) {
      const double s0 = v[i] * v[i], s1 = v[i + 1] * v[i + 1], s2 = v[i + 2] * v[i + 2], s3 = v[i + 3] * v[i + 3];
      len += s0 + s1 + s2 + s3;
    }
    len = sqrt(len);
    const double fac = 1.0 / len;
    for (size_t i = 0; i < v.size(); i++) v[i] *= fac;
    return len;
  }
  void sample_patch(const Img &grad, const Img &ori) {
    const double TWO_PI = 6.28318530718; /* M_PI_DOUBLED, siftdesc.cpp:18 */
    for (int r = 0; r < patchSize; ++r) {
      const int br0 = spatialBins * bin0[r];
      const float wr0 = w0[r];
      const int br1 = spatialBins * bin1[r];
      const float wr1 = w1[r];
      for (int c = 0; c < patchSize; ++c) {
        float val = 0.0f * 1.0 + (1.0 - 0.0f) * mask.at(r, c) * grad.at(r, c);
        const int bc0 = bin0[c];
        const float wc0 = w0[c] * val;
        const int bc1 = bin1[c];
        const float wc1 = w1[c] * val;
        const float o = float(orientationBins) * (ori.at(r, c) + TWO_PI) / TWO_PI;
        int bo0 = (int)o;
        const float wo1 = o - bo0;
        bo0 %= orientationBins;
        int bo1 = (bo0 + 1) % orientationBins;
        const float wo0 = 1.0f - wo1;
        val = wr0 * wc0;
        if (val > 0) { vec[br0 + bc0 + bo0] += val * wo0; vec[br0 + bc0 + bo1] += val * wo1; }
        val = wr0 * wc1;
        if (val > 0) { vec[br0 + bc1 + bo0] += val * wo0; vec[br0 + bc1 + bo1] += val * wo1; }
        val = wr1 * wc0;
        if (val > 0) { vec[br1 + bc0 + bo0] += val * wo0; vec[br1 + bc0 + bo1] += val * wo1; }
        val = wr1 * wc1;
        if (val > 0) { vec[br1 + bc1 + bo0] += val * wo0; vec[br1 + bc1 + bo1] += val * wo1; }
      }
    }
  }
  /* type: 0 SIFT, 1 RootSIFT, 2 HalfSIFT, 3 HalfRootSIFT (operator(), siftdesc.cpp:399-442).  The half variants
   * skip the norm of the 128-bin histogram (doNorm = false), fold opposite orientation bins
   * (half[i*4+j] = vec[i*8+j] + vec[i*8+j+4]) and normalise the 64-vector; desc[64..127] are written 0.
   * NOTE: SIFTnorm's clip loop runs to the MEMBER vec.size() = 128 on the 64-element half vector
   * (siftdesc.cpp:250-254), an out-of-bounds read/write in the reference; the loop is restated over the 64 entries. */
  void compute(const Img &patch, int type, float *desc) {
    const bool rootsift = (type & 1) != 0, half = type >= 2;
    const int w = patch.cols, h = patch.rows;
    Img grad(h, w), ori(h, w);
    for (int r = 0; r < h; ++r)
      for (int c = 0; c < w; ++c) {
        float xg, yg;
        if (c == 0) xg = patch.at(r, c + 1) - patch.at(r, c);
        else if (c == w - 1) xg = patch.at(r, c) - patch.at(r, c - 1);
        else xg = patch.at(r, c + 1) - patch.at(r, c - 1);
        if (r == 0) yg = patch.at(r + 1, c) - patch.at(r, c);
        else if (r == h - 1) yg = patch.at(r, c) - patch.at(r - 1, c);
        else yg = patch.at(r + 1, c) - patch.at(r - 1, c);
        grad.at(r, c) = sqrtf(xg * xg + yg * yg);
        ori.at(r, c) = atan2lut(yg, xg);
      }
    for (auto &x : vec) x = 0;
    sample_patch(grad, ori);
    std::vector<double> full;
    if (half) {
      full = vec;
      std::vector<double> hv(64);
      for (int i = 0, b = 0; i < 16; i++)
        for (int j = 0; j < 4; j++) hv[b++] = full[i * 8 + j] + full[i * 8 + j + 4];
      vec = hv;
    }
    normalize(vec);
    bool changed = false;
    for (size_t i = 0; i < vec.size(); i++)
      if (vec[i] > maxBinValue) { vec[i] = maxBinValue; changed = true; }
    if (changed) normalize(vec);
    if (rootsift) { /* RootSIFTnorm(vector<double>&), siftdesc.cpp:199-221 */
      double sum = 0.;
      for (size_t i = 0; i < vec.size(); i++) sum += fabs(vec[i]);
      for (size_t i = 0; i < vec.size(); i++) vec[i] = sqrt(vec[i] / sum);
      for (size_t i = 0; i < vec.size(); i++) {
        int b = std::max(0, std::min((int)(512.0 * vec[i] + 0.5), 255));
        vec[i] = double(b);
      }
    } else { /* SIFTnorm(vector<double>&), siftdesc.cpp:247-262 */
      for (size_t i = 0; i < vec.size(); i++) {
        int b = std::max(0, std::min((int)(512.0f * vec[i] + 0.5), 255));
        vec[i] = double(b);
      }
    }
    for (size_t i = 0; i < vec.size(); i++) desc[i] = (float)vec[i];
    if (half) {
      for (int i = 64; i < 128; i++) desc[i] = 0.f;
      vec.assign(128, 0.0);
    }
  }
};

/* slow/accurate branch of DescribeRegions, synth-detection.hpp:186-224 */
void extract_patch(const Img &img, const orc_keypoint &kp, double mrSize, int patchSize, bool fast, Img &patch) {
  patch = Img(patchSize, patchSize);
  if (!fast) {
    float mrScale = (float)ceil(kp.s * mrSize);
    int patchImageSize = 2 * int(mrScale) + 1;
    float imageToPatchScale = float(patchImageSize) / float(patchSize);
    if (imageToPatchScale > 0.4) {
      patchImageSize += 2;
      Img smoothed(patchImageSize, patchImageSize);
      interpolate(img, (float)kp.x, (float)kp.y, (float)kp.a11, (float)kp.a12, (float)kp.a21, (float)kp.a22, smoothed);
      Img sm2;
      gaussian_blur(smoothed, 1.5f * imageToPatchScale, sm2);
      interpolate(sm2, (float)(patchImageSize >> 1), (float)(patchImageSize >> 1), imageToPatchScale, 0, 0,
                  imageToPatchScale, patch);
    } else {
      interpolate(img, (float)kp.x, (float)kp.y, (float)kp.a11 * imageToPatchScale, (float)kp.a12 * imageToPatchScale,
                  (float)kp.a21 * imageToPatchScale, (float)kp.a22 * imageToPatchScale, patch);
    }
  } else {
    double mrScale = (double)mrSize * kp.s;
    int patchImageSize = 2 * int(mrScale) + 1;
    double imageToPatchScale = double(patchImageSize) / (double)patchSize;
    float curr_sc = imageToPatchScale;
    interpolate(img, (float)kp.x, (float)kp.y, (float)kp.a11 * curr_sc, (float)kp.a12 * curr_sc,
                (float)kp.a21 * curr_sc, (float)kp.a22 * curr_sc, patch);
  }
}

}  // namespace orc

using namespace orc;
extern "C" {

int orc_dominant_angles(const float *patch, int patchSize, int doHalfSIFT, double th, int maxAngles, float *angles,
                        int cap) {
  std::vector<float> a;
  dominant_angles(Img(patchSize, patchSize, patch), doHalfSIFT, th, maxAngles, a);
  for (size_t i = 0; i < a.size() && (int)i < cap; i++) angles[i] = a[i];
  return (int)a.size();
}

/* DetectOrientation, synth-detection.cpp:841-919 (identity H: Hinv unused) */
int orc_detect_orientation(const float *img_, int rows, int cols, const orc_region *in, int n, double mrSize,
                           int patchSize, int doHalfSIFT, int maxAngNum, double th, int addUpRight, orc_region *out,
                           int cap) {
  Img img(rows, cols, img_);
  int count = 0;
  int nout = 0;
  double mrScale = (double)mrSize;
  int patchImageSize = 2 * int(mrScale) + 1;
  double imageToPatchScale = double(patchImageSize) / (double)patchSize;
  Img patch(patchSize, patchSize);
  std::vector<float> angles1;
  for (int i = 0; i < n; i++) {
    orc_region base = in[i];
    angles1.clear();
    float curr_sc = imageToPatchScale * base.det_kp.s;
    if (interpolate_check_borders(cols, rows, (float)in[i].det_kp.x, (float)in[i].det_kp.y, (float)in[i].det_kp.a11,
                                  (float)in[i].det_kp.a12, (float)in[i].det_kp.a21, (float)in[i].det_kp.a22,
                                  (int)(K_SIGMA * in[i].det_kp.s), (int)(K_SIGMA * in[i].det_kp.s)))
      continue;
    if (maxAngNum > 0) {
      base.id = count;
      interpolate(img, (float)base.det_kp.x, (float)base.det_kp.y, (float)base.det_kp.a11 * curr_sc,
                  (float)base.det_kp.a12 * curr_sc, (float)base.det_kp.a21 * curr_sc, (float)base.det_kp.a22 * curr_sc,
                  patch);
      dominant_angles(patch, doHalfSIFT, th, maxAngNum, angles1);
      for (size_t j = 0; j < angles1.size(); j++) {
        /* synth-detection.cpp:30 has `using namespace std`, so cos/sin of a float resolve to the
           f32 overloads (cosf/sinf); the result is then widened to double (:900-901) */
        double ci = cosf(-angles1[j]);
        double si = sinf(-angles1[j]);
        orc_region t = base;
        t.det_kp.a11 = base.det_kp.a11 * ci - base.det_kp.a12 * si;
        t.det_kp.a12 = base.det_kp.a11 * si + base.det_kp.a12 * ci;
        t.det_kp.a21 = base.det_kp.a21 * ci - base.det_kp.a22 * si;
        t.det_kp.a22 = base.det_kp.a21 * si + base.det_kp.a22 * ci;
        if (nout < cap) out[nout] = t;
        nout++;
      }
    }
    if (addUpRight) { if (nout < cap) out[nout] = base; nout++; }
  }
  return nout;
}

/* ReprojectRegions, synth-detection.cpp:541-616.  H is original->view; the
   inverse is cv::invert(DECOMP_LU) == closed-form 3x3 adjugate. */
static int reproject_regions_k(orc_region *regs, int n, const double *H, int orig_w, int orig_h, double boxk);
int orc_reproject_regions(orc_region *regs, int n, const double *H, int orig_w, int orig_h) {
  return reproject_regions_k(regs, n, H, orig_w, orig_h, K_SIGMA);
}
/* ReprojectRegionsAndRemoveTouchBoundary, synth-detection.cpp:63-102: the same with the box mrSize * s (default
   3 sqrt 3 = k_sigma / 2, synth-detection.hpp:68) -- the "None" list of imagerepresentation.cpp:1271-1272 */
int orc_reproject_regions_touch_boundary(orc_region *regs, int n, const double *H, int orig_w, int orig_h, double mrSize) {
  return reproject_regions_k(regs, n, H, orig_w, orig_h, mrSize);
}
static int reproject_regions_k(orc_region *regs, int n, const double *H, int orig_w, int orig_h, double boxk) {
  double eyeTest = fabs(H[0] - 1.0) + fabs(H[1]) + fabs(H[2]) + fabs(H[3]) + fabs(H[4] - 1.0) + fabs(H[5]) +
                   fabs(H[6]) + fabs(H[7]) + fabs(H[8] - 1.0);
  double Hi[9];
  {
    const double *S = H;
    double d = S[0] * (S[4] * S[8] - S[5] * S[7]) - S[1] * (S[3] * S[8] - S[5] * S[6]) + S[2] * (S[3] * S[7] - S[4] * S[6]);
    if (d != 0.) {
      d = 1. / d;
      Hi[0] = (S[4] * S[8] - S[5] * S[7]) * d; Hi[1] = (S[2] * S[7] - S[1] * S[8]) * d; Hi[2] = (S[1] * S[5] - S[2] * S[4]) * d;
      Hi[3] = (S[5] * S[6] - S[3] * S[8]) * d; Hi[4] = (S[0] * S[8] - S[2] * S[6]) * d; Hi[5] = (S[2] * S[3] - S[0] * S[5]) * d;
      Hi[6] = (S[3] * S[7] - S[4] * S[6]) * d; Hi[7] = (S[1] * S[6] - S[0] * S[7]) * d; Hi[8] = (S[0] * S[4] - S[1] * S[3]) * d;
    } else for (int i = 0; i < 9; i++) Hi[i] = 0;
  }
  for (int i = 0; i < n; i++) {
    regs[i].reproj_kp = regs[i].det_kp;
    if (!(eyeTest < 0.01)) { /* ReprojectByH, synth-detection.cpp:490-498 */
      const orc_keypoint k = regs[i].det_kp;
      orc_keypoint &o = regs[i].reproj_kp;
      o.x = (Hi[0] * k.x + Hi[1] * k.y + Hi[2]);
      o.y = (Hi[3] * k.x + Hi[4] * k.y + Hi[5]);
      o.a11 = (Hi[0] * k.a11 + Hi[1] * k.a21);
      o.a12 = (Hi[0] * k.a12 + Hi[1] * k.a22);
      o.a21 = (Hi[3] * k.a11 + Hi[4] * k.a21);
      o.a22 = (Hi[3] * k.a12 + Hi[4] * k.a22);
    }
  }
  int m = 0;
  for (int i = 0; i < n; i++) {
    const orc_keypoint &k = regs[i].reproj_kp;
    if ((k.x < orig_w) && (k.y < orig_h) && (k.x > 0) && (k.y > 0)) {
      if (!interpolate_check_borders(orig_w, orig_h, k.x, k.y, k.a11, k.a12, k.a21, k.a22, (int)(boxk * k.s),
                                     (int)(boxk * k.s)))
        regs[m++] = regs[i];
    }
  }
  return m;
}

void orc_extract_patch(const float *img_, int rows, int cols, const orc_region *reg, double mrSize, int patchSize,
                       float *patch) {
  Img img(rows, cols, img_), p;
  extract_patch(img, reg->det_kp, mrSize, patchSize, false, p);
  memcpy(patch, p.v.data(), p.v.size() * 4);
}

void orc_describe_patch(float *patch41, int photoNorm, int rootsift, double maxBinValue, float *desc) {
  Img p(41, 41, patch41);
  Sift sift(maxBinValue);
  if (photoNorm) photometrically_normalize(p, sift.mask);
  sift.compute(p, rootsift, desc);
  memcpy(patch41, p.v.data(), p.v.size() * 4);
}

void orc_describe_regions(const float *img_, int rows, int cols, const orc_region *regs, int n, double mrSize,
                          int patchSize, int fast, int photoNorm, int rootsift, double maxBinValue, float *desc) {
  Img img(rows, cols, img_), patch;
  Sift sift(maxBinValue);
  for (int i = 0; i < n; i++) {
    extract_patch(img, regs[i].det_kp, mrSize, patchSize, fast != 0, patch);
    if (photoNorm) photometrically_normalize(patch, sift.mask);
    sift.compute(patch, rootsift, desc + (size_t)i * 128);
  }
}
}
