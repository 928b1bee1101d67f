// filters.cpp -- host-side correspondence filters around the verifier.
//   DuplicateFiltering                       matching/matching.cpp:2983-3047
//   LORANSACFiltering (H branch, Sampson)    matching/matching.cpp:806-980
//   NaiveHCheck                              matching/matching.cpp:1171-1200
//   H_LAF_check                              matching/matching.cpp:251-309
#include <math.h>
#include <stdint.h>
#include <string.h>
#include <algorithm>
#include "engine_api.hpp"

namespace mx {

// The reference sorts the tentatives with std::sort (unstable) on |ratio| before the greedy O(T^2)
// pass; the same libstdc++ introsort on the same key sequence yields the same permutation.
int duplicate_filtering(const double *pts, const double *key, int T, double r, int do_sort, int *order,
                        unsigned char *keep) {
  if (r <= 0) {
    for (int i = 0; i < T; i++) { order[i] = i; keep[i] = 1; }
    return T;
  }
  // sorted position -> tentative.  The permutation std::sort produces depends only on the outcomes of its comparisons, so
  // the elements may be anything that compares the same way.  The ratios are f32 quotients widened to f64 (29 zero bits at
  // the bottom of the mantissa): |key| then fits, with the index below it, into one 64-bit word whose top 35 bits order
  // like the doubles do -- half the bytes to move and an integer comparison.  Anything else (a key with low mantissa bits,
  // a NaN -- for which the double comparison is false both ways) takes the 16-byte elements.
  struct E { double key; int i; };
  static thread_local std::vector<E> v;
  static thread_local std::vector<uint64_t> w;
  bool packed = do_sort && key && T < (1 << 29);
  if (packed) {
    w.resize(T);
    for (int i = 0; i < T; i++) {
      uint64_t bits;
      memcpy(&bits, &key[i], 8);
      bits &= ~(1ull << 63);                                   // fabs
      if ((bits & ((1ull << 29) - 1)) || bits > 0x7ff0000000000000ull) { packed = false; break; }
      w[i] = bits | (uint64_t)i;
    }
  }
  if (packed) {
    std::sort(w.begin(), w.end(), [](uint64_t a, uint64_t b) { return (a >> 29) < (b >> 29); });
    for (int i = 0; i < T; i++) order[i] = (int)(w[i] & ((1u << 29) - 1));
  } else {
    v.resize(T);
    for (int i = 0; i < T; i++) { v[i].key = key ? key[i] : 0; v[i].i = i; }
    if (do_sort) std::sort(v.begin(), v.end(), [](E a, E b) { return fabs(a.key) < fabs(b.key); });
    for (int i = 0; i < T; i++) order[i] = v[i].i;
  }
  const double r_sq = r * r;
  // Same greedy pass as the reference's O(T^2) double loop (matching.cpp:3017-3036), with a uniform grid
  // over the image-1 coordinates so that only tentatives within +-1 cell (cell = r) of i are visited;
  // the visiting order inside the j-loop is irrelevant to the result (j is only ever flagged, and a j is
  // flagged by the first unflagged i < j that is within r in both images -- the flag itself is all that
  // later iterations observe).
  static thread_local std::vector<double> P;
  static thread_local std::vector<char> uniq;
  P.resize((size_t)T * 4);
  double minx = 0, miny = 0, maxx = 0, maxy = 0;
  bool finite = true;
  for (int i = 0; i < T; i++) {
    const double *p = pts + 4 * order[i];
    for (int q = 0; q < 4; q++) { P[4 * (size_t)i + q] = p[q]; finite = finite && std::isfinite(p[q]); }
    if (i == 0) { minx = maxx = p[0]; miny = maxy = p[1]; }
    if (p[0] < minx) minx = p[0];
    if (p[1] < miny) miny = p[1];
    if (p[0] > maxx) maxx = p[0];
    if (p[1] > maxy) maxy = p[1];
  }
  uniq.assign(T, 1);
  // cell edge a little above r (two points within r of each other are then in the same or in adjacent cells whatever the
  // rounding of the cell index: the index only has to be monotone and to move by at most one over a distance r) and at most
  // 1024 cells per side.  Multi-view tentatives cluster -- a dozen per scene point -- so cells of the radius itself matter:
  // with 96 cells per side the pass was 2 ms of a 12 k-tentative pair
  const double cell_sz = std::max(r * 1.001, std::max(maxx - minx, maxy - miny) / 1024.0);
  const double inv_cell = 1.0 / cell_sz;
  const double gw = (maxx - minx) / cell_sz, gh = (maxy - miny) / cell_sz;
  if (!finite || T < 64 || gw > 8192 || gh > 8192) {
    for (int i = 0; i < T; i++) {
      if (!uniq[i]) continue;
      const double *p1 = &P[4 * (size_t)i];
      for (int j = i + 1; j < T; j++) {
        if (!uniq[j]) continue;
        const double *p2 = &P[4 * (size_t)j];
        double dx = p1[0] - p2[0], dy = p1[1] - p2[1];
        double d1 = dx * dx + dy * dy;
        if (d1 > r_sq) continue;
        dx = p1[2] - p2[2]; dy = p1[3] - p2[3];
        double d2 = dx * dx + dy * dy;
        if (d2 <= r_sq) uniq[j] = 0;
      }
    }
  } else {
    // The same outcome, one tentative at a time: j is dropped iff a KEPT i < j lies within r of it in both images (a dropped
    // i flags nothing, and whether i is kept is settled before any j > i is looked at).  So only kept tentatives enter the
    // grid -- a list per cell, threaded through next[] -- and the look-up of j stops at the first hit.
    const int GW = (int)gw + 3, GH = (int)gh + 3;
    static thread_local std::vector<int> head, next;
    head.assign((size_t)GW * GH, -1);
    next.resize(T);
    for (int j = 0; j < T; j++) {
      const double *p2 = &P[4 * (size_t)j];
      const int cx = (int)((p2[0] - minx) * inv_cell), cy = (int)((p2[1] - miny) * inv_cell);
      bool hit = false;
      for (int yy = std::max(0, cy - 1); yy <= std::min(GH - 1, cy + 1) && !hit; yy++)
        for (int xx = std::max(0, cx - 1); xx <= std::min(GW - 1, cx + 1) && !hit; xx++)
          for (int i = head[yy * GW + xx]; i >= 0; i = next[i]) {
            const double *p1 = &P[4 * (size_t)i];
            double dx = p1[0] - p2[0], dy = p1[1] - p2[1];
            const double d1 = dx * dx + dy * dy;
            if (d1 > r_sq) continue;
            dx = p1[2] - p2[2]; dy = p1[3] - p2[3];
            const double d2 = dx * dx + dy * dy;
            if (d2 <= r_sq) { hit = true; break; }
          }
      if (hit) uniq[j] = 0;
      else { const int c = cy * GW + cx; next[j] = head[c]; head[c] = j; }
    }
  }
  int kept = 0;
  for (int i = 0; i < T; i++) { keep[i] = uniq[i]; kept += uniq[i]; }
  return kept;
}

int loransac_h(const double *pts, const double *laf1, const double *laf2, int T, double err_threshold,
               double confidence, int max_samples_, int lo, double HLAFCoef, int doSymmCheck, unsigned seed, double *H,
               double *Hraw, unsigned char *inl, unsigned char *keep, int *data_out3, int error_type) {
  (void)lo;  // the reference ignores localOptimization on the H path (iter_type is the constant 4)
  for (int i = 0; i < T; i++) { inl[i] = 0; keep[i] = 0; }
  for (int i = 0; i < 9; i++) { H[i] = -1; Hraw[i] = 0; }
  data_out3[0] = data_out3[1] = data_out3[2] = 0;
  if (T < 8) return 0;  // MIN_POINTS, matching.hpp:27
  int max_samples = max_samples_;
  if (T <= 20) max_samples = 1000;
  std::vector<double> u2((size_t)T * 6);
  for (int i = 0; i < T; i++) {
    u2[6 * i] = pts[4 * i]; u2[6 * i + 1] = pts[4 * i + 1]; u2[6 * i + 2] = 1.;
    u2[6 * i + 3] = pts[4 * i + 2]; u2[6 * i + 4] = pts[4 * i + 3]; u2[6 * i + 5] = 1.;
  }
  double Hloran[9];
  ransac_h(u2.data(), T, err_threshold * err_threshold, confidence, max_samples, Hloran, inl, data_out3, 1, doSymmCheck,
           seed, nullptr, error_type);
  for (int i = 0; i < 9; i++) Hraw[i] = Hloran[i];
  double Ht[9] = {Hloran[0], Hloran[3], Hloran[6], Hloran[1], Hloran[4], Hloran[7], Hloran[2], Hloran[5], Hloran[8]};
  double Hinv[9];
  invert3(Ht, Hinv);
  bool nz = false;
  for (int i = 0; i < 9; i++) nz = nz || (Hinv[i] != 0.0);
  if (!nz) { for (int i = 0; i < T; i++) inl[i] = 0; return 0; }
  for (int i = 0; i < 9; i++) H[i] = Hinv[i];
  std::vector<int> ril;
  for (int i = 0; i < T; i++) if (inl[i]) ril.push_back(i);
  double Hi2[9];
  invert3(H, Hi2);
  int good = 0;
  for (int i : ril) {
    const double x1 = pts[4 * i], y1 = pts[4 * i + 1], x2 = pts[4 * i + 2], y2 = pts[4 * i + 3];
    double xa = (H[0] * x1 + H[1] * y1 + H[2]) / (H[6] * x1 + H[7] * y1 + H[8]);
    double ya = (H[3] * x1 + H[4] * y1 + H[5]) / (H[6] * x1 + H[7] * y1 + H[8]);
    double d1 = (x2 - xa) * (x2 - xa) + (y2 - ya) * (y2 - ya);
    xa = (Hi2[0] * x2 + Hi2[1] * y2 + Hi2[2]) / (Hi2[6] * x2 + Hi2[7] * y2 + Hi2[8]);
    ya = (Hi2[3] * x2 + Hi2[4] * y2 + Hi2[5]) / (Hi2[6] * x2 + Hi2[7] * y2 + Hi2[8]);
    double d2 = (x1 - xa) * (x1 - xa) + (y1 - ya) * (y1 - ya);
    if ((d1 <= 100.0) && (d2 <= 100.0)) good++;
  }
  if (good < 8) ril.clear();
  const double affErr = 3.0 * HLAFCoef * err_threshold;
  std::vector<int> kept;
  if (affErr > 0) {
    double HsT[9], Hs1[9];
    hds_sym_setup(Hloran, HsT, Hs1);
    for (int i : ril) {
      double u[18], err[3];
      const double *A = laf1 + 5 * i, *B = laf2 + 5 * i;
      u[0] = pts[4 * i]; u[1] = pts[4 * i + 1]; u[2] = 1.0;
      u[3] = pts[4 * i + 2]; u[4] = pts[4 * i + 3]; u[5] = 1.0;
      u[6] = u[0] + 3.0 * A[1] * A[4]; u[7] = u[1] + 3.0 * A[3] * A[4]; u[8] = 1.0;
      u[9] = u[3] + 3.0 * B[1] * B[4]; u[10] = u[4] + 3.0 * B[3] * B[4]; u[11] = 1.0;
      u[12] = u[0] + 3.0 * A[0] * A[4]; u[13] = u[1] + 3.0 * A[2] * A[4]; u[14] = 1.0;
      u[15] = u[3] + 3.0 * B[0] * B[4]; u[16] = u[4] + 3.0 * B[2] * B[4]; u[17] = 1.0;
      hds_sym_with(u, HsT, Hs1, err, 3, true);
      double sumErr = sqrt(err[0] + err[1] + err[2]);
      if (!(sumErr > affErr)) kept.push_back(i);
    }
  } else kept = ril;
  if ((int)kept.size() < 8) kept.clear();
  for (int i : kept) keep[i] = 1;
  return (int)kept.size();
}

}  // namespace mx
