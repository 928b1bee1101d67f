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
#include <emmintrin.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <sys/mman.h>
#include <string.h>
#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <deque>
#include <functional>
#include <memory>
#include <mutex>
#include <thread>
#include <chrono>
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
  // parent: -1 = pixel not seen yet (the hot array: four neighbour look-ups per pixel); node: per ROOT the counters of a small
  // component, or the index of its region (>= 0) -- written when a pixel becomes a root, so it needs no clearing
  struct Node { int area, perim, region; };
  Node *node;
  int *parent;
  std::vector<GrownRegion> regions;    // promotion order = output order
  Forest(Node *n, int *par) : node(n), parent(par) {}
  int minSize, promoteAt, maxSize;
  double minMargin;
  bool relative, inverted;

  int find(int p) {
    while (parent[p] != p) { const int g = parent[parent[p]]; parent[p] = g; p = g; }   // path halving
    return p;
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
    g.area = node[root].area; g.perimeter = node[root].perim;
    g.seed = root; g.kept = true;
    g.addArea.assign(256, 0); g.addPerim.assign(256, 0);
    g.addArea[level] = g.area; g.addPerim[level] = g.perimeter;
    node[root].region = (int)regions.size();
    regions.push_back(std::move(g));
  }

  void add_pixel(int root, int ofs, int level, int touching) {
    parent[ofs] = root;
    const int dPerim = 4 - 2 * touching;
    Node &R = node[root];
    if (R.region < 0) {
      R.area++; R.perim += dPerim;
      if (R.area >= promoteAt) promote(root, level);
    } else {
      GrownRegion &g = regions[R.region];
      g.last = level; g.area++; g.perimeter += dPerim;
      g.addArea[level]++; g.addPerim[level] += dPerim;
    }
  }

  // order / start: the view's pixel offsets (padded coordinates) in (grey level, raster) order and the first place of every level
  // (sort_pixels below); inverted: the tree of 255 - grey walks the same buckets from level 255 down
  void run(const int *order, const int *start) {
    const size_t npx = (size_t)(rows + 2) * stride;
    memset(parent, 0xff, npx * sizeof(int));       // -1 = not seen
    parent[0] = 0;     // the sentinel chain of the neighbour look-ups below: offset 0 is a frame pixel, never visited
    static const int PF = getenv("MODSX_MSER_PF") ? atoi(getenv("MODSX_MSER_PF")) : 12;
    int lastRoot = -1;
    for (int level = 0; level < 256; level++) {
      const int bucket = inverted ? 255 - level : level, kEnd = start[bucket + 1];
      for (int k = start[bucket]; k < kEnd; k++) {
        const int ofs = order[k];
        if (k + PF < kEnd) { const int f = order[k + PF]; __builtin_prefetch(&parent[f - stride]); __builtin_prefetch(&parent[f]); __builtin_prefetch(&parent[f + stride]); }
        const int nb[4] = {ofs - stride, ofs - 1, ofs + 1, ofs + stride};
        int roots[4], nroots = 0, touching = 0;
        {
          // the common case without a data-dependent branch per neighbour: an unseen neighbour reads the sentinel chain 0 -> 0
          // (offset 0 is a frame pixel, never visited), a seen one is at most two steps from its root in a compressed forest
          const int p0 = parent[nb[0]], p1 = parent[nb[1]], p2 = parent[nb[2]], p3 = parent[nb[3]];
          touching = (p0 >= 0) + (p1 >= 0) + (p2 >= 0) + (p3 >= 0);
          const int a0 = p0 < 0 ? 0 : p0, a1 = p1 < 0 ? 0 : p1, a2 = p2 < 0 ? 0 : p2, a3 = p3 < 0 ? 0 : p3;
          const int r0 = parent[a0], r1 = parent[a1], r2 = parent[a2], r3 = parent[a3];
          const int m = std::max(std::max(r0, r1), std::max(r2, r3));
          const bool same = ((r0 == 0) | (r0 == m)) & ((r1 == 0) | (r1 == m)) & ((r2 == 0) | (r2 == m)) & ((r3 == 0) | (r3 == m));
          if (same && m > 0 && parent[m] == m) { roots[0] = m; nroots = 1; }
          else if (touching) {
            for (int q = 0; q < 4; q++) {
              const int pq = parent[nb[q]];
              if (pq < 0) continue;
              const int r = find(nb[q]);
              bool dup = false;
              for (int z = 0; z < nroots; z++) dup = dup || roots[z] == r;
              if (!dup) roots[nroots++] = r;
            }
          }
        }
        if (nroots == 0) {                       // a new component: area 1, perimeter 4
          parent[ofs] = ofs; node[ofs] = Node{1, 4, -1};
          lastRoot = ofs;
          continue;
        }
        int keep = roots[0];
        if (nroots > 1) {
          unsigned bestPrev = 0;
          int grown = 0;
          for (int z = 0; z < nroots; z++)
            if (node[roots[z]].region >= 0) {
              const GrownRegion &g = regions[node[roots[z]].region];
              const unsigned prev = (unsigned)(g.area - g.addArea[level]);   // its size one level below
              grown++;
              if (prev > bestPrev) { bestPrev = prev; keep = roots[z]; }
            }
          for (int z = 0; z < nroots; z++) {
            const int r = roots[z];
            if (r == keep) continue;
            parent[r] = keep;
            const int ri = node[r].region;
            const int a = ri < 0 ? node[r].area : regions[ri].area, b = ri < 0 ? node[r].perim : regions[ri].perimeter;
            if (node[keep].region < 0) { node[keep].area += a; node[keep].perim += b; }
            else {
              GrownRegion &g = regions[node[keep].region];
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
    }
    if (rows > 0 && cols > 0) {
      const int root = find(stride + 1);
      if (node[root].region >= 0) close_region(regions[node[root].region]);
    }
    (void)lastRoot;
  }
};

// The bin sort of a view (sortPixels.cpp:76-125): pixel offsets (padded coordinates, stride cols + 2) per grey level in raster order.
// Both polarities walk the same buckets (the inverted image's level l is this one's 255 - l, raster order inside a level is the
// same), so a view is sorted ONCE: every caller counts for itself (a third of a millisecond) and the two polarity tasks scatter one
// half of the rows each -- whoever comes first takes the other half too if nobody has.  The rows are cut into four strips that are
// counted, and two at a time scattered, side by side: neighbouring pixels mostly share their grey level, and one counter per level
// makes every step wait for the store of the step before it (a store-to-load forward per pixel); independent counters do not.
struct ViewSort {
  std::vector<int> order;
  std::atomic<int> part[2];     // the scatter of strips 0-1 / 2-3: 0 free, 1 taken, 2 done
  ViewSort() { part[0].store(0); part[1].store(0); }
};
void sort_pixels(const uint8_t *u8, int rows, int cols, ViewSort &V, int mine, int *start /* [257] */) {
  constexpr int NS = 4;
  const int stride = cols + 2;
  int cnt[NS * 256];
  memset(cnt, 0, sizeof cnt);
  int rb[NS + 1];
  for (int q = 0; q <= NS; q++) rb[q] = (int)((long)rows * q / NS);
  {
    int *c0 = &cnt[0], *c1 = &cnt[256], *c2 = &cnt[512], *c3 = &cnt[768];
    const int len = rb[1] - rb[0];                    // every strip has len or len + 1 rows; walk `len` rows of all four together
    for (int i = 0; i < len; i++) {
      const uint8_t *g0 = u8 + (size_t)(rb[0] + i) * cols, *g1 = u8 + (size_t)(rb[1] + i) * cols;
      const uint8_t *g2 = u8 + (size_t)(rb[2] + i) * cols, *g3 = u8 + (size_t)(rb[3] + i) * cols;
      for (int c = 0; c < cols; c++) { c0[g0[c]]++; c1[g1[c]]++; c2[g2[c]]++; c3[g3[c]]++; }
    }
    for (int q = 0; q < NS; q++)
      for (int r = rb[q] + len; r < rb[q + 1]; r++) { const uint8_t *g = u8 + (size_t)r * cols; int *cq = &cnt[q * 256]; for (int c = 0; c < cols; c++) cq[g[c]]++; }
  }
  start[0] = 0;
  for (int l = 0; l < 256; l++) {                     // level l: strip 0's pixels, then strip 1's, ... = raster order
    int at = start[l];
    for (int q = 0; q < NS; q++) { const int n = cnt[q * 256 + l]; cnt[q * 256 + l] = at; at += n; }
    start[l + 1] = at;
  }
  int *ord = V.order.data();
  for (int turn = 0; turn < 2; turn++) {
    const int h = turn ? 1 - mine : mine;
    int free0 = 0;
    if (!V.part[h].compare_exchange_strong(free0, 1, std::memory_order_acquire)) continue;
    int *ca = &cnt[(2 * h) * 256], *cb = &cnt[(2 * h + 1) * 256];
    const int ra = rb[2 * h], rbb = rb[2 * h + 1], na = rbb - ra, nb = rb[2 * h + 2] - rbb, len = na < nb ? na : nb;
    for (int i = 0; i < len; i++) {
      const uint8_t *ga = u8 + (size_t)(ra + i) * cols, *gb = u8 + (size_t)(rbb + i) * cols;
      const int oa = (ra + i + 1) * stride + 1, ob = (rbb + i + 1) * stride + 1;
      for (int c = 0; c < cols; c++) { ord[ca[ga[c]]++] = oa + c; ord[cb[gb[c]]++] = ob + c; }
    }
    for (int r = ra + len; r < rbb; r++) { const uint8_t *g = u8 + (size_t)r * cols; const int o = (r + 1) * stride + 1; for (int c = 0; c < cols; c++) ord[ca[g[c]]++] = o + c; }
    for (int r = rbb + len; r < rb[2 * h + 2]; r++) { const uint8_t *g = u8 + (size_t)r * cols; const int o = (r + 1) * stride + 1; for (int c = 0; c < cols; c++) ord[cb[g[c]]++] = o + c; }
    V.part[h].store(2, std::memory_order_release);
  }
  for (int spin = 0; V.part[0].load(std::memory_order_acquire) != 2 || V.part[1].load(std::memory_order_acquire) != 2; spin++)
    if (spin > 64) std::this_thread::yield();       // the other polarity's task is scattering its half right now
}

struct RowRun { int line, c0, c1; };


// row runs (raster order) of the 4-connected component of {grey <= level} that contains `seed`: a span fill -- a pixel
// taken off the stack is grown to its maximal horizontal span (which IS a row run of the component), the spans above and
// below are seeded from it; the runs are then sorted by (line, first column).  `mark` is all-zero on entry and on exit.
void component_runs(const uint8_t *grey, int stride, int seed, int level, std::vector<uint8_t> &mark, std::vector<int> &stack,
                    std::vector<RowRun> &runs) {
  stack.clear(); runs.clear();
  // the padding frame holds 255 in `fence` terms: the caller guarantees level < 255 and a frame of 255s (see mser_polarity)
  if (grey[seed] > level) return;
  stack.push_back(seed);
  while (!stack.empty()) {
    const int o = stack.back(); stack.pop_back();
    if (mark[o]) continue;
    int a = o, b = o;
    {
      // the maximal span around o, 16 pixels per look (the frame of 255s ends every row, so a look never matters past it)
      const __m128i lvl = _mm_set1_epi8((char)level), zero = _mm_setzero_si128();
      auto open16 = [&](int at) {
        const __m128i g = _mm_loadu_si128(reinterpret_cast<const __m128i *>(grey + at));
        const __m128i m = _mm_loadu_si128(reinterpret_cast<const __m128i *>(mark.data() + at));
        return (unsigned)_mm_movemask_epi8(_mm_and_si128(_mm_cmpeq_epi8(_mm_max_epu8(g, lvl), lvl), _mm_cmpeq_epi8(m, zero)));
      };
      for (;;) {
        const unsigned open = open16(b + 1);
        if (open == 0xffffu) { b += 16; continue; }
        b += __builtin_ctz(~open);
        break;
      }
      for (;;) {
        if (a < 16) { while (grey[a - 1] <= level && !mark[a - 1]) a--; break; }
        const unsigned open = open16(a - 16);          // bit i = pixel a - 16 + i
        if (open == 0xffffu) { a -= 16; continue; }
        a -= __builtin_clz((~open & 0xffffu) << 16);
        break;
      }
    }
    memset(&mark[a], 1, (size_t)(b - a + 1));
    const int line = a / stride;
    runs.push_back({line - 1, a - line * stride - 1, b - line * stride - 1});
    // the rows above and below: every start of a stretch of open pixels (grey <= level, not marked) under the span is a seed.
    // 16 pixels per step (the arrays are padded by 32 bytes for the reads past b): open = bytes that pass both tests, a start is
    // an open pixel whose left neighbour (the last pixel of the step before, for bit 0) is not.
    const __m128i lvl = _mm_set1_epi8((char)level), zero = _mm_setzero_si128();
    for (int dir = -1; dir <= 1; dir += 2) {
      const int base = dir * stride;
      unsigned carry = 0;
      for (int q = a; q <= b; q += 16) {
        const __m128i g = _mm_loadu_si128(reinterpret_cast<const __m128i *>(grey + q + base));
        const __m128i m = _mm_loadu_si128(reinterpret_cast<const __m128i *>(mark.data() + q + base));
        unsigned open = (unsigned)_mm_movemask_epi8(_mm_and_si128(_mm_cmpeq_epi8(_mm_max_epu8(g, lvl), lvl), _mm_cmpeq_epi8(m, zero)));
        const int left = b - q + 1;                    // pixels of the span in this step
        if (left < 16) open &= (1u << left) - 1u;
        unsigned starts = open & ~((open << 1) | carry);
        carry = (open >> 15) & 1u;
        while (starts) { const int bit = __builtin_ctz(starts); starts &= starts - 1; stack.push_back(q + bit + base); }
      }
    }
  }
  std::sort(runs.begin(), runs.end(), [](const RowRun &x, const RowRun &y) { return x.line != y.line ? x.line < y.line : x.c0 < y.c0; });
  for (const RowRun &q : runs) memset(&mark[(size_t)(q.line + 1) * stride + q.c0 + 1], 0, (size_t)(q.c1 - q.c0 + 1));
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

// per-thread scratch of the component tree (12-16 bytes per pixel): reused from call to call, so that a worker thread does
// not page in fresh memory for every view
// the forest's two arrays (4 + 12 bytes per pixel, read in grey-level order: every look-up a different page) on transparent huge
// pages where the system hands them out on request
struct HugeBuf {
  void *p = nullptr;
  size_t cap = 0;
  void *ensure(size_t bytes) {
    if (bytes <= cap) return p;
    if (p) free(p);
    const size_t want = (bytes + (2u << 20) - 1) & ~(size_t)((2u << 20) - 1);
    if (posix_memalign(&p, 2u << 20, want)) { p = nullptr; cap = 0; return nullptr; }
    static const bool off = getenv("MODSX_MSER_NOHUGE") != nullptr;
    if (!off) madvise(p, want, MADV_HUGEPAGE);
    cap = want;
    return p;
  }
  ~HugeBuf() { if (p) free(p); }
};
struct MserScratch {
  HugeBuf node, parent;
  std::vector<int> stack;
  std::vector<uint8_t> fence, mark;
  std::vector<RowRun> runs;
};
static thread_local MserScratch t_scratch;

// MSER+ (pol 0) or MSER- (pol 1: the inverted image, extremaInvertImage) of one view, in region / threshold order
// devOrder / devStart: the view's sorted offsets and level starts when the device has sorted it (kernels_pyramid.hip: k_mser_*), or null
static void mser_polarity(const uint8_t *u8, int rows, int cols, const modsx_mser_params &par, double minMargin, int pol, ViewSort &vs,
                          std::vector<modsx_keypoint> &out, const int *devOrder = nullptr, const int *devStart = nullptr) {
  MserScratch &S = t_scratch;
  const int stride = cols + 2;
  const size_t npx = (size_t)(rows + 2) * stride;
  // `fence`: the polarity's pixels inside a frame of 255, which stops the span fill of any threshold < 255 at the image border
  // (the tree itself only sees the sorted offsets)
  S.fence.assign(npx + 32, 255);                                   // (+ 32: the span fill reads 16 bytes at a time)
  if (S.mark.size() < npx + 32) S.mark.assign(npx + 32, 0);        // all-zero between calls: component_runs clears what it marks
  for (int r = 0; r < rows; r++) {
    uint8_t *f = &S.fence[(size_t)(r + 1) * stride + 1];
    const uint8_t *src = u8 + (size_t)r * cols;
    if (pol == 0) memcpy(f, src, cols);
    else for (int c = 0; c < cols; c++) f[c] = (uint8_t)(255 - src[c]);
  }
  int start[257];
  if (devOrder) memcpy(start, devStart, sizeof start);
  else sort_pixels(u8, rows, cols, vs, pol, start);
  Forest F((Forest::Node *)S.node.ensure(npx * sizeof(Forest::Node)), (int *)S.parent.ensure(npx * sizeof(int)));
  if (!F.node || !F.parent) return;              // (out of memory: no keys for this view)
  F.rows = rows; F.cols = cols; F.stride = stride;
  F.minSize = par.min_size; F.promoteAt = std::min(10000, par.min_size);
  F.maxSize = (int)((double)cols * rows * par.max_area);
  F.minMargin = par.relative ? minMargin / 100.0 : minMargin;
  F.relative = par.relative != 0; F.inverted = pol == 1;
  F.run(devOrder ? devOrder : vs.order.data(), start);
  for (const GrownRegion &g : F.regions) {
    if (!g.kept) continue;
    for (const StableLevel &t : g.levels) {
      if (t.thresh >= 255) continue;
      component_runs(S.fence.data(), stride, g.seed, t.thresh, S.mark, S.stack, S.runs);
      if (S.runs.empty()) continue;
      double cx, cy, sxx, sxy, syy, A[4];
      run_moments(S.runs, cx, cy, sxx, sxy, syy);
      sym_sqrt(sxx, sxy, syy, A);
      modsx_keypoint k;
      memset(&k, 0, sizeof k);
      k.x = cx; k.y = cy; k.a11 = A[0]; k.a12 = A[1]; k.a21 = A[2]; k.a22 = A[3];
      k.s = 1.0; k.response = t.margin; k.sub_type = pol == 0 ? 21 : 20;
      out.push_back(k);
    }
  }
}

// prepareKeysForExport, extrema.cpp:31-90 (same libstdc++ std::sort on the same sequence: MSER+ keys, then MSER-)
static void mser_export(std::vector<modsx_keypoint> &out, const modsx_mser_params &par, double minMargin, double tilt, double zoom) {
  int regNumber = par.reg_number;
  if ((tilt > 2.0) || (zoom < 0.5)) regNumber = (int)floor(zoom * 2.0 * regNumber / tilt);
  if (out.empty() || par.mode == MODSX_FIXED_TH) return;
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

// CPUs this process may use: the host's threads, capped by the container's allowance (cgroup v2 cpu.max, v1 cfs quota -- the GPU
// boxes show 256 logical CPUs and allow 16), divided by the ranks that share the host (torchrun / mpirun export the local world size)
int host_cpus_per_rank() {
  int n = (int)std::thread::hardware_concurrency();
  double quota = 0;
  if (FILE *f = fopen("/sys/fs/cgroup/cpu.max", "r")) {
    char q[64]; double per = 0;
    if (fscanf(f, "%63s %lf", q, &per) == 2 && strcmp(q, "max") && per > 0) quota = atof(q) / per;
    fclose(f);
  } else if (FILE *g = fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r")) {
    double qv = 0, per = 0;
    if (fscanf(g, "%lf", &qv) == 1 && qv > 0) {
      if (FILE *h = fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r")) { if (fscanf(h, "%lf", &per) == 1 && per > 0) quota = qv / per; fclose(h); }
    }
    fclose(g);
  }
  if (quota >= 1) n = std::min(n, (int)(quota + 0.5));
  for (const char *v : {"LOCAL_WORLD_SIZE", "OMPI_COMM_WORLD_LOCAL_SIZE", "MV2_COMM_WORLD_LOCAL_SIZE"})
    if (const char *e = getenv(v)) { const int w = atoi(e); if (w > 1) { n = std::max(1, n / w); break; } }
  return std::max(1, n);
}

// ---- host worker pool: the (view, polarity) component trees of a view set are independent (the reference runs one view
// per OpenMP thread, imagerepresentation.cpp:612-622, with threadprivate MSER globals) -------------------------------------
namespace {
class HostPool {
 public:
  static HostPool &get() { static HostPool p; return p; }
  // runs fn(0) .. fn(n - 1) on the workers and on the caller, returns when all are done.  Tasks are handed out by an atomic
  // counter; only as many workers as there are tasks are woken.
  void run(int n, const std::function<void(int)> &fn, bool mayHelp = false) {
    if (n <= 0) return;
    if (n == 1 || threads_.empty()) { for (int i = 0; i < n; i++) fn(i); return; }
    auto job = std::make_shared<Job>();
    job->n = n; job->fn = &fn;
    {
      std::lock_guard<std::mutex> lk(mu_);
      jobs_.push_back(job);
    }
    epoch_.fetch_add(1, std::memory_order_release);      // workers that are still looking around take it from here
    // only sleeping workers need a wake-up, and a futex wake is microseconds of the caller's time each: the short loops of a
    // launch set follow one another within tens of microseconds, which the workers bridge spinning (loop())
    const int wake = std::min(std::min(n - 1, (int)threads_.size()), sleepers_.load(std::memory_order_acquire));
    for (int k = 0; k < wake; k++) cv_.notify_one();
    work(*job);                    // the calling thread takes tasks too
    // Its last tasks may run for milliseconds on other threads (a component tree).  When the host is what the process is short of
    // (the flag-word wait is on, engine.hip) the caller of a long job takes tasks of the OTHER contexts' jobs meanwhile instead of
    // yielding in a loop (3.5 % of the ladder's host CPU was this loop)
    for (int spin = 0; job->done.load(std::memory_order_acquire) < n; spin++) {
      static const bool helpOff = getenv("MODSX_POOL_HELP") && !atoi(getenv("MODSX_POOL_HELP"));
      if (mayHelp && !helpOff && !host_wait_runtime()) {
        std::shared_ptr<Job> other;
        {
          std::lock_guard<std::mutex> lk(mu_);
          other = front_job();
        }
        if (other && other.get() != job.get()) { work(*other); continue; }
      }
      if (spin > 64) std::this_thread::yield();
    }
    std::lock_guard<std::mutex> lk(mu_);
    for (auto it = jobs_.begin(); it != jobs_.end(); ++it) if (it->get() == job.get()) { jobs_.erase(it); break; }
  }
 private:
  struct Job { int n = 0; std::atomic<int> next{0}, done{0}; const std::function<void(int)> *fn = nullptr; };
  std::mutex mu_;
  std::condition_variable cv_;
  std::deque<std::shared_ptr<Job>> jobs_;
  std::vector<std::thread> threads_;
  std::atomic<int> epoch_{0}, sleepers_{0};
  int spinUs_ = 0;     // MODSX_HOST_SPIN_US: measured on a lone 31-view pair, 150 us of watching bought nothing in the median and cost outliers
  bool stop_ = false;
  HostPool() {
    // default: the host's threads divided by the ranks that share it (torchrun / mpirun export the local world size: one
    // process per GPU, 8 per node), capped by OMP_NUM_THREADS when the launcher set one; MODSX_HOST_THREADS overrides
    // ... of the CPUs this process may actually use: a container's allowance (cgroup v2 cpu.max, v1 cfs quota) is what the
    // GPU boxes limit (256 logical CPUs visible, 16 allowed) -- threads beyond it only buy throttling
    int n = host_cpus_per_rank();
    if (const char *e = getenv("OMP_NUM_THREADS")) { const int o = atoi(e); if (o > 0) n = std::min(n, std::max(o, 4)); }
    if (const char *e = getenv("MODSX_HOST_THREADS")) n = atoi(e);
    if (const char *e = getenv("MODSX_HOST_SPIN_US")) spinUs_ = atoi(e);
    n = std::max(0, std::min(n, 64) - 1);
    for (int i = 0; i < n; i++) threads_.emplace_back([this] { loop(); });
  }
  ~HostPool() {
    { std::lock_guard<std::mutex> lk(mu_); stop_ = true; }
    cv_.notify_all();
    for (std::thread &t : threads_) t.join();
  }
  static void work(Job &j) {
    int mine = 0;
    for (;;) {
      const int i = j.next.fetch_add(1, std::memory_order_relaxed);
      if (i >= j.n) break;
      (*j.fn)(i);
      mine++;
    }
    if (mine) j.done.fetch_add(mine, std::memory_order_release);
  }
  // the first job that still has tasks to hand out (finished ones are dropped from the front); mu_ held
  std::shared_ptr<Job> front_job() {
    while (!jobs_.empty() && jobs_.front()->next.load(std::memory_order_relaxed) >= jobs_.front()->n) jobs_.pop_front();
    return jobs_.empty() ? nullptr : jobs_.front();
  }
  void loop() {
    for (;;) {
      std::shared_ptr<Job> j;
      const int seen = epoch_.load(std::memory_order_acquire);
      {
        std::lock_guard<std::mutex> lk(mu_);
        if (stop_) return;
        j = front_job();
      }
      if (j) { work(*j); continue; }
      // nothing to do: watch the epoch for a while before going to sleep (a job pushed after `seen` was read changes it; one
      // pushed before is found by the look under the lock below)
      bool fresh = false;
      if (spinUs_ > 0) {
        const auto until = std::chrono::steady_clock::now() + std::chrono::microseconds(spinUs_);
        for (int it = 0;; it++) {
          if (epoch_.load(std::memory_order_acquire) != seen) { fresh = true; break; }
          __builtin_ia32_pause();
          if ((it & 63) == 63 && std::chrono::steady_clock::now() >= until) break;
        }
      }
      if (fresh) continue;
      std::unique_lock<std::mutex> lk(mu_);
      sleepers_.fetch_add(1, std::memory_order_acq_rel);
      cv_.wait(lk, [&] { return stop_ || front_job() != nullptr; });
      sleepers_.fetch_sub(1, std::memory_order_acq_rel);
      if (stop_) return;
    }
  }
};
std::atomic<int> g_setsInFlight(0);
}  // namespace

// a launch set is being driven by the calling thread (detect_describe_views): how many are decides whether their short
// host loops may use the pool
// the calling thread drives a part of a lone pair (accumulate_views): its short loops use the pool whatever the count says
static thread_local bool t_lightPool = false;
void host_light_pool(bool on) { t_lightPool = on; }
void host_set_enter() { g_setsInFlight.fetch_add(1); }
void host_set_leave() { g_setsInFlight.fetch_sub(1); }


// fn(0) .. fn(n - 1) in parallel.  light = true marks the short per-view loops of a launch set (hundreds of microseconds of
// work): they go to the pool only while at most two launch sets are in flight, or when the caller says so (the parts of a lone pair, host_light_pool); with
// many contexts at work every host core already has a context's own loop to run and the loops stay where they are.
void host_parallel_for(int n, const std::function<void(int)> &fn, bool light) {
  if (light) {
    static const bool off = getenv("MODSX_HOST_SERIAL") != nullptr;
    static const int maxSets = getenv("MODSX_LIGHT_SETS") ? atoi(getenv("MODSX_LIGHT_SETS")) : 2;
    if ((g_setsInFlight.load() > maxSets && !t_lightPool) || off) { for (int i = 0; i < n; i++) fn(i); return; }
    HostPool::get().run(n, fn);
    return;
  }
  HostPool::get().run(n, fn, true);
}

// u8: rows x cols grey values (already truncated from the f32 view).  Appends nothing to `out` beyond the keypoints of
// this view; returns MODSX_OK.
int detect_msers_host(const uint8_t *u8, int rows, int cols, const modsx_mser_params &par, double tilt, double zoom,
                      std::vector<modsx_keypoint> &out) {
  out.clear();
  if (rows <= 0 || cols <= 0) return MODSX_OK;
  const double minMargin = par.mode != MODSX_FIXED_TH ? 1.0 : par.min_margin;
  ViewSort vs;
  vs.order.resize((size_t)rows * cols);
  for (int pol = 0; pol < 2; pol++) mser_polarity(u8, rows, cols, par, minMargin, pol, vs, out);
  mser_export(out, par, minMargin, tilt, zoom);
  return MODSX_OK;
}

// The views of a set: 2 n independent (view, polarity) trees on the host pool, largest first; out[i] = DetectMSERs of view i
int detect_msers_views(const uint8_t *const *u8, const int *rows, const int *cols, int n, const modsx_mser_params &par,
                       const double *tilts, const double *zooms, std::vector<modsx_keypoint> *out, const int *const *devOrder,
                       const int *const *devStart) {
  const double minMargin = par.mode != MODSX_FIXED_TH ? 1.0 : par.min_margin;
  std::vector<std::vector<modsx_keypoint>> part((size_t)2 * n);
  std::vector<ViewSort> sorts((size_t)n);         // one bin sort per view, shared by its two polarity tasks
  if (!devOrder) for (int v = 0; v < n; v++) if (rows[v] > 0 && cols[v] > 0) sorts[v].order.resize((size_t)rows[v] * cols[v]);
  std::vector<int> task((size_t)2 * n);
  for (int i = 0; i < 2 * n; i++) task[i] = i;
  std::stable_sort(task.begin(), task.end(), [&](int a, int b) { return (long)rows[a / 2] * cols[a / 2] > (long)rows[b / 2] * cols[b / 2]; });
  static const bool trace = getenv("MODSX_HOST_TIMING") && atoi(getenv("MODSX_HOST_TIMING")) >= 3;
  auto nowms = []() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
  std::vector<double> tb(trace ? 2 * n : 0), te(trace ? 2 * n : 0);
  std::vector<size_t> tid(trace ? 2 * n : 0);
  const double t00 = trace ? nowms() : 0;
  host_parallel_for(2 * n, [&](int k) {
    const int t = task[k], v = t / 2;
    if (trace) { tb[k] = nowms() - t00; tid[k] = std::hash<std::thread::id>()(std::this_thread::get_id()) % 1000; }
    if (rows[v] > 0 && cols[v] > 0)
      mser_polarity(u8[v], rows[v], cols[v], par, minMargin, t & 1, sorts[v], part[t], devOrder ? devOrder[v] : nullptr, devOrder ? devStart[v] : nullptr);
    if (trace) te[k] = nowms() - t00;
  });
  if (trace && nowms() - t00 > 30) {
    std::string line = "  slow mser job (" + std::to_string(2 * n) + " tasks, " + std::to_string(nowms() - t00) + " ms): task start-end@thread px:";
    for (int k = 0; k < 2 * n && k < 14; k++) {
      char b[96];
      snprintf(b, sizeof b, " %.1f-%.1f@%zu %dk", tb[k], te[k], tid[k], rows[task[k] / 2] * cols[task[k] / 2] / 1000);
      line += b;
    }
    fprintf(stderr, "%s\n", line.c_str());
  }
  for (int v = 0; v < n; v++) {
    out[v] = std::move(part[2 * v]);
    out[v].insert(out[v].end(), part[2 * v + 1].begin(), part[2 * v + 1].end());
    mser_export(out[v], par, minMargin, tilts[v], zooms[v]);
  }
  return MODSX_OK;
}

}  // namespace mx
