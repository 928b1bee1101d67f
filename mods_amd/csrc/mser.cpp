// mser.cpp -- MSER+ / MSER- detection (host C++), SURVEY.md rows E1 / E2.
//
// Reference behaviour: DetectMSERs, detectors/mser/extrema/extrema.cpp:284-473 -> getRLEExtrema (libExtrema.cpp:539-560)
//   -> GetExtrema (getExtrema.cpp:390-437) + FastSetOptThresholds4StableRegion (optThresh.cpp:69-166)
//   -> RegionBoundaries (boundary.cpp:107-209) -> RLE2Ellipse (libExtrema.cpp:117-159) -> A = C^(1/2).
// The reference threads tagged pointers through a label image and packs the counters of small components into the
// pointer words; this implementation keeps the same component semantics in plain arrays:
//   * pixels enter in (grey level, raster) order; a union-find forest over pixel offsets, whose ROOT is always the first
//     pixel of the surviving component (the reference keeps the surviving label word);
//   * a component below min_size pixels only counts pixels and 4-connected perimeter ("border"); at min_size it is
//     promoted to a region with a per-level histogram of added pixels / perimeter, born at the current level;
//   * when components meet, the region that was largest at the PREVIOUS level survives (ties: first in the order up,
//     left, right, down; only a strictly larger size replaces the default first neighbour), the others are closed:
//     closed regions that lived for more than min_margin levels get their stable thresholds, the rest are dropped;
//   * stability of a closed region: for every level i the margin is the number of levels until the area has grown by
//     more than the perimeter at i; local maxima of the margin (non-descending runs) become thresholds at
//     pos + margin/2, overlapping ones are thinned and near-equal areas merged (optThresh.cpp:15-65);
//   * each (region, threshold) is the 4-connected component of {grey <= threshold} holding the region's first pixel;
//     its row runs give centroid and covariance, the keypoint is (centroid, sqrt of the covariance, s = 1).
// The sequential part is inherently ordered (the reference is single-threaded per view as well); the device only
// truncates the f32 view to u8 (k_trunc_u8) so that 1 byte per pixel crosses PCIe.
//
// Known deviations: the packed counters of the reference overflow their 15-bit size field when components of more than
// 32767 pixels are merged before promotion (impossible for min_size <= 8191); AffineKeypoint::octave_number and
// pyramid_scale are uninitialised stack values in the reference and 0 here; thresholds at level 255, for which the
// reference never builds a boundary and would dereference NULL, are skipped.
#include <math.h>
#include <stdint.h>
#include <string.h>
#include <algorithm>
#include <vector>
#include "engine_api.hpp"

namespace mx {

namespace {

struct StableLevel { int thresh, pos, margin; };

struct GrownRegion {
  int born, last;          // first / latest grey level with pixels of this region (minimum_int / maximum_int)
  int area, perimeter;     // running totals
  int seed;                // pixel offset of the component's first pixel (padded coordinates)
  bool kept;               // still in the output list
  std::vector<int> addArea, addPerim;   // per level increments, turned into cumulative sums when the region closes
  std::vector<StableLevel> levels;
};

struct Forest {
  int rows, cols, stride;
  const uint8_t *grey;                 // padded (rows + 2) x stride
  std::vector<int> parent;             // -1 = pixel not seen yet
  std::vector<int> smallArea, smallPerim, regionOf;   // per root: counters of a small component, or region index (>= 0)
  std::vector<GrownRegion> regions;    // promotion order = output order
  int minSize, promoteAt, maxSize;
  double minMargin;
  bool relative, inverted;

  int find(int p) {
    int r = p;
    while (parent[r] != r) r = parent[r];
    while (parent[p] != r) { const int n = parent[p]; parent[p] = r; p = n; }
    return r;
  }

  void thin_levels(GrownRegion &g) {
    std::vector<StableLevel> &t = g.levels;
    const std::vector<int> &cum = g.addArea;
    for (int i = 0; i < (int)t.size(); i++)
      while (i >= 0 && i + 1 < (int)t.size()) {
        const StableLevel &a = t[i], &b = t[i + 1];
        if (a.pos + a.margin < b.thresh && a.thresh < b.pos) break;        // disjoint
        if (b.margin <= a.margin) t.erase(t.begin() + i + 1);              // keep the wider margin
        else { t.erase(t.begin() + i); i--; break; }
      }
    for (int i = 0; i < (int)t.size(); i++)
      while (i + 1 < (int)t.size()) {
        StableLevel &a = t[i];
        const StableLevel &b = t[i + 1];
        if (a.pos + a.margin < b.pos) break;
        if (cum[b.thresh] - cum[a.thresh] > 0.1 * cum[a.thresh]) break;
        a.margin = b.pos - a.pos + b.margin;
        a.thresh = a.pos + a.margin / 2;
        t.erase(t.begin() + i + 1);
      }
  }

  void close_region(GrownRegion &g) {
    if (g.area < minSize) return;
    for (int l = g.born + 1; l <= g.last; l++) { g.addArea[l] += g.addArea[l - 1]; g.addPerim[l] += g.addPerim[l - 1]; }
    const std::vector<int> &A = g.addArea, &B = g.addPerim;
    const int invC = inverted ? 255 : 0, invM = inverted ? -1 : 1;
    int bestMargin = -1, bestPos = -1, up;
    auto flush = [&]() {
      const int th = bestPos + bestMargin / 2;
      if (A[th] <= maxSize && A[th] > minSize) g.levels.push_back({th, bestPos, bestMargin});
    };
    int i = g.born;
    do {
      up = (int)(i + minMargin);
      if (up > g.last) break;
      while (A[up] - A[i] < B[i] && up < g.last) up++;
      const int margin = up - i;
      double q = (double)margin;
      if (relative) q /= invC + invM * (i + margin / 2);
      if (q > minMargin && margin >= bestMargin) { bestMargin = margin; bestPos = i; }
      else {
        if (bestPos >= 0) { flush(); bestPos = -1; }
        bestMargin = margin;
      }
      i++;
    } while (up < g.last);
    if (bestPos >= 0) flush();
    thin_levels(g);
  }

  void promote(int root, int level) {
    GrownRegion g;
    g.born = g.last = level;
    g.area = smallArea[root]; g.perimeter = smallPerim[root];
    g.seed = root; g.kept = true;
    g.addArea.assign(256, 0); g.addPerim.assign(256, 0);
    g.addArea[level] = g.area; g.addPerim[level] = g.perimeter;
    regionOf[root] = (int)regions.size();
    regions.push_back(g);
  }

  void add_pixel(int root, int ofs, int level, int touching) {
    parent[ofs] = root;
    const int dPerim = 4 - 2 * touching;
    if (regionOf[root] < 0) {
      smallArea[root]++; smallPerim[root] += dPerim;
      if (smallArea[root] >= promoteAt) promote(root, level);
    } else {
      GrownRegion &g = regions[regionOf[root]];
      g.last = level; g.area++; g.perimeter += dPerim;
      g.addArea[level]++; g.addPerim[level] += dPerim;
    }
  }

  void run() {
    const size_t npx = (size_t)(rows + 2) * stride;
    parent.assign(npx, -1);
    smallArea.assign(npx, 0); smallPerim.assign(npx, 0); regionOf.assign(npx, -1);
    // bin sort: offsets per grey level in raster order (sortPixels.cpp:76-125)
    std::vector<int> start(257, 0);
    for (int r = 1; r <= rows; r++) for (int c = 1; c <= cols; c++) start[grey[(size_t)r * stride + c] + 1]++;
    for (int l = 0; l < 256; l++) start[l + 1] += start[l];
    std::vector<int> order((size_t)rows * cols), fill(start.begin(), start.end() - 1);
    for (int r = 1; r <= rows; r++) for (int c = 1; c <= cols; c++) { const int o = r * stride + c; order[fill[grey[o]]++] = o; }
    int lastRoot = -1;
    for (int level = 0; level < 256; level++)
      for (int k = start[level]; k < start[level + 1]; k++) {
        const int ofs = order[k];
        const int nb[4] = {ofs - stride, ofs - 1, ofs + 1, ofs + stride};
        int roots[4], nroots = 0, touching = 0;
        for (int q = 0; q < 4; q++) {
          if (parent[nb[q]] < 0) continue;
          touching++;
          const int r = find(nb[q]);
          bool dup = false;
          for (int z = 0; z < nroots; z++) dup = dup || roots[z] == r;
          if (!dup) roots[nroots++] = r;
        }
        if (nroots == 0) {                       // a new component: area 1, perimeter 4
          parent[ofs] = ofs; smallArea[ofs] = 1; smallPerim[ofs] = 4;
          lastRoot = ofs;
          continue;
        }
        int keep = roots[0];
        if (nroots > 1) {
          unsigned bestPrev = 0;
          int grown = 0;
          for (int z = 0; z < nroots; z++)
            if (regionOf[roots[z]] >= 0) {
              const GrownRegion &g = regions[regionOf[roots[z]]];
              const unsigned prev = (unsigned)(g.area - g.addArea[level]);   // its size one level below
              grown++;
              if (prev > bestPrev) { bestPrev = prev; keep = roots[z]; }
            }
          for (int z = 0; z < nroots; z++) {
            const int r = roots[z];
            if (r == keep) continue;
            parent[r] = keep;
            const int ri = regionOf[r];
            const int a = ri < 0 ? smallArea[r] : regions[ri].area, b = ri < 0 ? smallPerim[r] : regions[ri].perimeter;
            if (regionOf[keep] < 0) { smallArea[keep] += a; smallPerim[keep] += b; }
            else {
              GrownRegion &g = regions[regionOf[keep]];
              g.area += a; g.perimeter += b; g.addArea[level] += a; g.addPerim[level] += b;
            }
            if (ri >= 0 && grown) {
              GrownRegion &m = regions[ri];
              if (!relative && (level - m.born + 1) <= minMargin) m.kept = false;
              else {
                m.last = level;
                close_region(m);
                if (m.levels.empty()) m.kept = false;
              }
            }
          }
        }
        add_pixel(keep, ofs, level, touching);
        lastRoot = keep;
      }
    if (rows > 0 && cols > 0) {
      const int root = find(stride + 1);
      if (regionOf[root] >= 0) close_region(regions[regionOf[root]]);
    }
    (void)lastRoot;
  }
};

struct RowRun { int line, c0, c1; };

// row runs (raster order) of the 4-connected component of {grey <= level} that contains `seed`
void component_runs(const uint8_t *grey, int rows, int cols, int stride, int seed, int level, std::vector<uint8_t> &mark,
                    std::vector<int> &stack, std::vector<int> &pix, std::vector<RowRun> &runs) {
  pix.clear(); stack.clear(); runs.clear();
  auto ok = [&](int o) {
    const int r = o / stride, c = o - r * stride;
    return r >= 1 && r <= rows && c >= 1 && c <= cols && grey[o] <= level && !mark[o];
  };
  if (!ok(seed)) return;
  mark[seed] = 1; stack.push_back(seed);
  while (!stack.empty()) {
    const int o = stack.back(); stack.pop_back();
    pix.push_back(o);
    const int nb[4] = {o + stride, o - stride, o + 1, o - 1};
    for (int q : nb) if (ok(q)) { mark[q] = 1; stack.push_back(q); }
  }
  std::sort(pix.begin(), pix.end());
  for (size_t i = 0; i < pix.size();) {
    size_t j = i;
    while (j + 1 < pix.size() && pix[j + 1] == pix[j] + 1) j++;
    runs.push_back({pix[i] / stride - 1, pix[i] % stride - 1, pix[j] % stride - 1});
    i = j + 1;
  }
  for (int o : pix) mark[o] = 0;
}

// RLE2Ellipse, libExtrema.cpp:117-159: area moments of the unit-square pixels of the runs
void run_moments(const std::vector<RowRun> &runs, double &cx, double &cy, double &sxx, double &sxy, double &syy) {
  double area = 0, sx = 0, sy = 0;
  for (const RowRun &q : runs) {
    const double line = q.line, m = q.c0, n = 1 + q.c1;
    sx += (n * n - m * m) / 2;
    sy += (n - m) * (2 * line + 1) / 2;
    area += n - m;
  }
  cx = (double)sx / (double)area;
  cy = (double)sy / (double)area;
  sxx = syy = sxy = 0;
  for (const RowRun &q : runs) {
    const double line = q.line - cy, m = q.c0 - cx, n = 1 + q.c1 - cx;
    const double l2 = line * line, m2 = m * m, n2 = n * n;
    sxx += (n2 * n - m2 * m) / 3;
    syy += (n - m) * (3 * l2 + 3 * line + 1) / 3;
    sxy += -.25 * (m2 - n2) * (2 * line + 1);
  }
  sxx /= (double)area; syy /= (double)area; sxy /= (double)area;
}

// A = Q sqrt(T) Q^T with the Jacobi rotation of Matrix2::schur_sym (utls/matrix.cpp:185-217)
void sym_sqrt(double c00, double c01, double c11, double A[4]) {
  double t, r;
  if (c01 != 0) {
    r = double(c11 - c00) / (2 * c01);
    if (r >= 0) t = 1.0 / (r + ::sqrt(1 + r * r));
    else t = -1.0 / (-r + ::sqrt(1 + r * r));
    r = 1.0 / ::sqrt(1 + t * t);
    t = t * r;
  } else { r = 1; t = 0; }
  const double Q[2][2] = {{r, t}, {-t, r}}, C[2][2] = {{c00, c01}, {c01, c11}};
  double QtC[2][2], T[2][2], QS[2][2];
  for (int i = 0; i < 2; i++) for (int j = 0; j < 2; j++) QtC[i][j] = Q[0][i] * C[0][j] + Q[1][i] * C[1][j];
  for (int i = 0; i < 2; i++) for (int j = 0; j < 2; j++) T[i][j] = QtC[i][0] * Q[0][j] + QtC[i][1] * Q[1][j];
  const double S[2][2] = {{::sqrt(T[0][0]), 0.0}, {0.0, ::sqrt(T[1][1])}};
  for (int i = 0; i < 2; i++) for (int j = 0; j < 2; j++) QS[i][j] = Q[i][0] * S[0][j] + Q[i][1] * S[1][j];
  for (int i = 0; i < 2; i++) for (int j = 0; j < 2; j++) A[i * 2 + j] = QS[i][0] * Q[j][0] + QS[i][1] * Q[j][1];
}

}  // namespace

// u8: rows x cols grey values (already truncated from the f32 view).  Appends nothing to `out` beyond the keypoints of
// this view; returns MODSX_OK.
int detect_msers_host(const uint8_t *u8, int rows, int cols, const modsx_mser_params &par, double tilt, double zoom,
                      std::vector<modsx_keypoint> &out) {
  out.clear();
  if (rows <= 0 || cols <= 0) return MODSX_OK;
  int regNumber = par.reg_number;
  if ((tilt > 2.0) || (zoom < 0.5)) regNumber = (int)floor(zoom * 2.0 * regNumber / tilt);
  const double minMargin = par.mode != MODSX_FIXED_TH ? 1.0 : par.min_margin;
  const int stride = cols + 2;
  std::vector<uint8_t> grey((size_t)(rows + 2) * stride, 0), mark((size_t)(rows + 2) * stride, 0);
  for (int r = 0; r < rows; r++) memcpy(&grey[(size_t)(r + 1) * stride + 1], u8 + (size_t)r * cols, cols);
  std::vector<int> stack, pix;
  std::vector<RowRun> runs;
  for (int pol = 0; pol < 2; pol++) {
    if (pol == 1)
      for (int r = 1; r <= rows; r++) for (int c = 1; c <= cols; c++) { uint8_t &v = grey[(size_t)r * stride + c]; v = 255 - v; }
    Forest F;
    F.rows = rows; F.cols = cols; F.stride = stride; F.grey = grey.data();
    F.minSize = par.min_size; F.promoteAt = std::min(10000, par.min_size);
    F.maxSize = (int)((double)cols * rows * par.max_area);
    F.minMargin = par.relative ? minMargin / 100.0 : minMargin;
    F.relative = par.relative != 0; F.inverted = pol == 1;
    F.run();
    for (const GrownRegion &g : F.regions) {
      if (!g.kept) continue;
      for (const StableLevel &t : g.levels) {
        if (t.thresh >= 255) continue;
        component_runs(grey.data(), rows, cols, stride, g.seed, t.thresh, mark, stack, pix, runs);
        if (runs.empty()) continue;
        double cx, cy, sxx, sxy, syy, A[4];
        run_moments(runs, cx, cy, sxx, sxy, syy);
        sym_sqrt(sxx, sxy, syy, A);
        modsx_keypoint k;
        memset(&k, 0, sizeof k);
        k.x = cx; k.y = cy; k.a11 = A[0]; k.a12 = A[1]; k.a21 = A[2]; k.a22 = A[3];
        k.s = 1.0; k.response = t.margin; k.sub_type = pol == 0 ? 21 : 20;
        out.push_back(k);
      }
    }
  }
  // prepareKeysForExport, extrema.cpp:31-90 (same libstdc++ std::sort on the same sequence)
  if (!out.empty() && par.mode != MODSX_FIXED_TH) {
    auto byMargin = [](const modsx_keypoint &a, const modsx_keypoint &b) { return fabs(a.response) > fabs(b.response); };
    std::sort(out.begin(), out.end(), byMargin);
    const double top = fabs(out[0].response);
    const int have = (int)out.size();
    modsx_keypoint probe = out[0];
    switch (par.mode) {
      case MODSX_RELATIVE_TH:
        probe.response = top * par.rel_threshold;
        out.resize(std::lower_bound(out.begin(), out.end(), probe, byMargin) - out.begin());
        break;
      case MODSX_FIXED_REG_NUMBER:
        if (regNumber < have && regNumber >= 0) out.resize(regNumber);
        break;
      case MODSX_RELATIVE_REG_NUMBER:
        out.resize((size_t)std::max(0, (int)floor(par.rel_reg_number * (double)out.size())));
        break;
      case MODSX_NOT_LESS_THAN_REGIONS: {
        probe.response = minMargin;
        const int fixed = (int)(std::lower_bound(out.begin(), out.end(), probe, byMargin) - out.begin());
        out.resize((size_t)std::max(0, fixed < regNumber ? std::min(regNumber, have) : std::min(fixed, have)));
        break;
      }
      default: break;
    }
  }
  return MODSX_OK;
}

}  // namespace mx
