/*
 * oracle_match.cpp -- TEST INFRASTRUCTURE ONLY (see oracle.h).
 *
 * CPU restatement of
 *   MatchFlannFGINN with vector_matcher=linear   matching/matching.cpp:357-461
 *     (cv::flann linear index: squared L2 in f32, kNN sorted ascending, ties keep
 *      the lower train index -- KNNSimpleResultSet::addPoint)
 *   DuplicateFiltering                           matching/matching.cpp:2983-3047
 *   LORANSACFiltering (H branch)                 matching/matching.cpp:806-980
 *   NaiveHCheck / H_LAF_check                    matching/matching.cpp:1171-1200, 251-309
 * exp_ransacHcustom itself is NOT restated: it is the reference's own degensac
 * code compiled in place into oracle/_ref/libdegensac_ref.so (oracle/Makefile).
 */
#include <dlfcn.h>
#include "oracle_internal.hpp"

namespace orc {

/* linear kNN: nn best (dist, idx) ascending; equal distances keep insertion (index) order.  Queries are independent:
 * the loop runs under OpenMP (OMP_NUM_THREADS=1 for the single-thread baseline), the per-query arithmetic is unchanged. */
void knn_linear(const float *d1, int n1, const float *d2, int n2, int dim, int nn, int *idx, float *dist) {
#pragma omp parallel for schedule(dynamic, 16)
  for (int q = 0; q < n1; q++) {
    std::vector<float> bd(nn);
    std::vector<int> bi(nn);
    int count = 0;
    float worst = std::numeric_limits<float>::max();
    const float *a = d1 + (size_t)q * dim;
    for (int t = 0; t < n2; t++) {
      const float *b = d2 + (size_t)t * dim;
      float r = 0;
      for (int k = 0; k < dim; k += 4) {
        float e0 = a[k] - b[k], e1 = a[k + 1] - b[k + 1], e2 = a[k + 2] - b[k + 2], e3 = a[k + 3] - b[k + 3];
        r += e0 * e0 + e1 * e1 + e2 * e2 + e3 * e3;
      }
      if (r >= worst) continue;
      int i;
      for (i = count; i > 0; --i) {
        if (bd[i - 1] > r) {
          if (i < nn) { bd[i] = bd[i - 1]; bi[i] = bi[i - 1]; }
        } else break;
      }
      if (count < nn) ++count;
      bd[i] = r; bi[i] = t;
      if (count == nn) worst = bd[nn - 1];
    }
    for (int j = 0; j < nn; j++) {
      idx[(size_t)q * nn + j] = j < count ? bi[j] : -1;
      dist[(size_t)q * nn + j] = j < count ? bd[j] : std::numeric_limits<float>::max();
    }
  }
}

}  // namespace orc

using namespace orc;

/* ---- reference degensac (oracle/_ref) ------------------------------------- */
typedef struct { unsigned I; double J; } RefScore;
typedef void (*HDsPtr)(const double *, const double *, const double *, double *, int);
typedef void (*HDsiPtr)(const double *, const double *, const double *, double *, int, int *, int);
typedef void (*HDsidxPtr)(const double *, const double *, const double *, double *, int, int *, int);
typedef RefScore (*ransacH_fn)(double *u, int len, double th, double conf, int max_sam, double *H,
                               unsigned char *inl, int iter_type, int *data_out, int oriented_constraint,
                               unsigned inlLimit, double **resids, HDsPtr, HDsiPtr, HDsidxPtr, int doSymCheck);
typedef void (*FDsPtr)(const double *, const double *, double *, int);
typedef void (*exFDsPtr)(const double *, const double *, double *, double *, int);
typedef int (*ransacF_fn)(double *u, int len, double th, double conf, int max_sam, double *F, unsigned char *inl,
                          int *data_out, int do_lo, unsigned inlLimit, double **resids, double *H_best, int *Ih,
                          exFDsPtr, FDsPtr, int doSymCheck);
typedef void (*set_seed_fn)(unsigned);

static void *g_ref = nullptr;
static ransacH_fn g_ransacH = nullptr;
static HDsPtr g_HDs = nullptr, g_HDsSymMax = nullptr, g_HDsSym = nullptr;
static HDsiPtr g_HDsi = nullptr, g_HDsiSym = nullptr, g_HDsiSymMax = nullptr;
static HDsidxPtr g_HDsidx = nullptr, g_HDsSymidx = nullptr, g_HDsSymidxMax = nullptr;
static set_seed_fn g_set_seed = nullptr;
static ransacF_fn g_ransacF = nullptr;
static FDsPtr g_FDs = nullptr, g_FDsSym = nullptr;
static exFDsPtr g_exFDs = nullptr, g_exFDsSym = nullptr;

static bool load_ref() {
  if (g_ref) return true;
  Dl_info info;
  std::string dir = ".";
  if (dladdr((void *)&load_ref, &info) && info.dli_fname) {
    std::string p(info.dli_fname);
    size_t k = p.rfind('/');
    if (k != std::string::npos) dir = p.substr(0, k);
  }
  std::string path = dir + "/_ref/libdegensac_ref.so";
  g_ref = dlopen(path.c_str(), RTLD_NOW | RTLD_LOCAL);
  if (!g_ref) return false;
  g_ransacH = (ransacH_fn)dlsym(g_ref, "exp_ransacHcustom");
  g_HDs = (HDsPtr)dlsym(g_ref, "HDs");
  g_HDsi = (HDsiPtr)dlsym(g_ref, "HDsi");
  g_HDsidx = (HDsidxPtr)dlsym(g_ref, "HDsidx");
  g_HDsSymMax = (HDsPtr)dlsym(g_ref, "HDsSymMax");
  g_HDsSym = (HDsPtr)dlsym(g_ref, "HDsSym");
  g_HDsiSym = (HDsiPtr)dlsym(g_ref, "HDsiSym");
  g_HDsiSymMax = (HDsiPtr)dlsym(g_ref, "HDsiSymMax");
  g_HDsSymidx = (HDsidxPtr)dlsym(g_ref, "HDsSymidx");
  g_HDsSymidxMax = (HDsidxPtr)dlsym(g_ref, "HDsSymidxMax");
  g_set_seed = (set_seed_fn)dlsym(g_ref, "modsx_ref_set_seed");
  g_ransacF = (ransacF_fn)dlsym(g_ref, "exp_ransacFcustom");
  g_FDs = (FDsPtr)dlsym(g_ref, "FDs");
  g_FDsSym = (FDsPtr)dlsym(g_ref, "FDsSym");
  g_exFDs = (exFDsPtr)dlsym(g_ref, "exFDs");
  g_exFDsSym = (exFDsPtr)dlsym(g_ref, "exFDsSym");
  if (!g_ransacF || !g_FDs || !g_FDsSym || !g_exFDs || !g_exFDsSym || !g_ransacH || !g_HDs || !g_HDsi || !g_HDsidx || !g_HDsSymMax || !g_set_seed || !g_HDsSym || !g_HDsiSym ||
      !g_HDsiSymMax || !g_HDsSymidx || !g_HDsSymidxMax) {
    dlclose(g_ref);
    g_ref = nullptr;
    return false;
  }
  return true;
}

static bool invert3(const double *S, double *t) { /* cv::invert 3x3 DECOMP_LU closed form */
  double d = S[0] * (S[4] * S[8] - S[5] * S[7]) - S[1] * (S[3] * S[8] - S[5] * S[6]) + S[2] * (S[3] * S[7] - S[4] * S[6]);
  if (d == 0.) { for (int i = 0; i < 9; i++) t[i] = 0; return false; }
  d = 1. / d;
  t[0] = (S[4] * S[8] - S[5] * S[7]) * d; t[1] = (S[2] * S[7] - S[1] * S[8]) * d; t[2] = (S[1] * S[5] - S[2] * S[4]) * d;
  t[3] = (S[5] * S[6] - S[3] * S[8]) * d; t[4] = (S[0] * S[8] - S[2] * S[6]) * d; t[5] = (S[2] * S[3] - S[0] * S[5]) * d;
  t[6] = (S[3] * S[7] - S[4] * S[6]) * d; t[7] = (S[1] * S[6] - S[0] * S[7]) * d; t[8] = (S[0] * S[4] - S[1] * S[3]) * d;
  return true;
}

extern "C" {

int orc_knn_linear(const float *desc1, int n1, const float *desc2, int n2, int dim, int nn, int *idx, float *dist) {
  knn_linear(desc1, n1, desc2, n2, dim, nn, idx, dist);
  return n1;
}

/* MatchFlannFGINN, matching/matching.cpp:357-461 (sqminratio < 1 branch) */
int orc_match_fginn(const float *desc1, int n1, const float *desc2, int n2, int dim, const double *pos2, double ratioT,
                    double contradDist, int nn, orc_tentative *out, int cap) {
  if (n1 == 0 || n2 == 0) return 0;
  double sqminratio = ratioT * ratioT;
  double contrDistSq = contradDist * contradDist;
  std::vector<int> idx((size_t)n1 * nn);
  std::vector<float> dist((size_t)n1 * nn);
  knn_linear(desc1, n1, desc2, n2, dim, nn, idx.data(), dist.data());
  int matches = 0;
  for (int i = 0; i < n1; i++) {
    const int *ir = idx.data() + (size_t)i * nn;
    const float *dr = dist.data() + (size_t)i * nn;
    for (int j = 1; j < nn; j++) {
      if (ir[j] < 0) break; /* fewer than nn trains: the reference would read garbage */
      double ratio = dr[0] / dr[j];
      if (sqminratio >= 1.0) {
        /* "to get all points (for example, for calculating PDF)", matching.cpp:397-428: every query gives a record, closed by its
         * first contradictive neighbour or by the last one looked at; the ratio test is commented out there */
        double dx = pos2[2 * ir[0]] - pos2[2 * ir[j]], dy = pos2[2 * ir[0] + 1] - pos2[2 * ir[j] + 1];
        if (j == nn - 1 || dx * dx + dy * dy > contrDistSq) {
          if (matches < cap) {
            orc_tentative t;
            t.q = i; t.t0 = ir[0]; t.tj = ir[j]; t.t1 = ir[1];
            t.d1 = dr[0]; t.d2 = dr[j]; t.d2by2ndcl = dr[1];
            t.ratio = sqrt(ratio);
            out[matches] = t;
          }
          matches++;
          break;
        }
        continue;
      }
      if (ratio <= sqminratio) {
        if (matches < cap) {
          orc_tentative t;
          t.q = i; t.t0 = ir[0]; t.tj = ir[j]; t.t1 = ir[1];
          t.d1 = dr[0]; t.d2 = dr[j]; t.d2by2ndcl = dr[1];
          t.ratio = sqrt(ratio);
          out[matches] = t;
        }
        matches++;
        break;
      }
      double dx = pos2[2 * ir[0]] - pos2[2 * ir[j]], dy = pos2[2 * ir[0] + 1] - pos2[2 * ir[j] + 1];
      double dist1 = dx * dx + dy * dy;
      if (dist1 > contrDistSq) break;
    }
  }
  return matches;
}

/* DuplicateFiltering, matching/matching.cpp:2983-3047.  The sort key is
 * |ratio| (MODE_FGINN), |d1| or |scale|; std::sort is the same unstable introsort. */
int orc_duplicate_filtering(const double *pts, const double *key, int T, double r, int do_sort, int *order,
                            unsigned char *keep) {
  struct E { double key; int i; };
  std::vector<E> v(T);
  for (int i = 0; i < T; i++) { v[i].key = key ? key[i] : 0; v[i].i = i; }
  if (r <= 0) {
    for (int i = 0; i < T; i++) { order[i] = i; keep[i] = 1; }
    return T;
  }
  if (do_sort) std::sort(v.begin(), v.end(), [](E a, E b) { return fabs(a.key) < fabs(b.key); });
  double r_sq = r * r;
  std::vector<char> uniq(T, 1);
  for (int i = 0; i < T; i++) {
    if (!uniq[i]) continue;
    const double *p1 = pts + 4 * v[i].i;
    for (int j = i + 1; j < T; j++) {
      if (!uniq[j]) continue;
      const double *p2 = pts + 4 * v[j].i;
      double dx = p1[0] - p2[0], dy = p1[1] - p2[1];
      double d1 = dx * dx + dy * dy;
      if (d1 > r_sq) continue;
      dx = p1[2] - p2[2]; dy = p1[3] - p2[3];
      double d2 = dx * dx + dy * dy;
      if (d2 <= r_sq) uniq[j] = 0;
    }
  }
  int kept = 0;
  for (int i = 0; i < T; i++) { order[i] = v[i].i; keep[i] = uniq[i]; kept += uniq[i]; }
  return kept;
}

int orc_ref_available(void) { return load_ref() ? 1 : 0; }

/* LORANSACFiltering (useF = 0), matching/matching.cpp:806-980, calling the
 * reference's exp_ransacHcustom with a fixed seed (oracle/ref_shim.c replaces
 * time() for exp_ranH.c only).  errorType = SAMPSON (config_iter_mods_cviu.ini:164). */
int orc_loransac_h(const double *pts, const double *laf1, const double *laf2, int T, double err_threshold,
                   double confidence, int max_samples_, int lo, double HLAFCoef, int doSymmCheck, unsigned seed,
                   double *H, double *Hraw, unsigned char *inl, unsigned char *keep, int *data_out3, int error_type) {
  (void)lo;
  for (int i = 0; i < T; i++) { inl[i] = 0; keep[i] = 0; }
  for (int i = 0; i < 9; i++) { H[i] = -1; Hraw[i] = 0; }
  if (!load_ref()) return -1;
  if (T < 8) return 0;
  int max_samples = max_samples_;
  if (T <= 20) max_samples = 1000;
  std::vector<double> u2((size_t)T * 6);
  for (int i = 0; i < T; i++) {
    u2[6 * i] = pts[4 * i]; u2[6 * i + 1] = pts[4 * i + 1]; u2[6 * i + 2] = 1.;
    u2[6 * i + 3] = pts[4 * i + 2]; u2[6 * i + 4] = pts[4 * i + 3]; u2[6 * i + 5] = 1.;
  }
  std::vector<int> data_out((size_t)T * 18 + 18);
  double *resids = nullptr;
  double Hloran[9];
  g_set_seed(seed);
  g_ransacH(u2.data(), T, err_threshold * err_threshold, confidence, max_samples, Hloran, inl, 4, data_out.data(), 1,
            0, &resids, error_type == 0 ? g_HDs : error_type == 1 ? g_HDsSymMax : g_HDsSym,
            error_type == 0 ? g_HDsi : error_type == 1 ? g_HDsiSymMax : g_HDsiSym,
            error_type == 0 ? g_HDsidx : error_type == 1 ? g_HDsSymidxMax : g_HDsSymidx, doSymmCheck);   /* matching.cpp:821-846 */
  free(resids);
  data_out3[0] = data_out[0]; data_out3[1] = data_out[1]; data_out3[2] = data_out[2];
  for (int i = 0; i < 9; i++) Hraw[i] = Hloran[i];
  /* H = inv(Hloran^T) */
  double Ht[9] = {Hloran[0], Hloran[3], Hloran[6], Hloran[1], Hloran[4], Hloran[7], Hloran[2], Hloran[5], Hloran[8]};
  double Hinv[9];
  invert3(Ht, Hinv);
  bool nz = false;
  for (int i = 0; i < 9; i++) nz = nz || (Hinv[i] != 0.0);
  if (!nz) { for (int i = 0; i < T; i++) inl[i] = 0; return 0; }
  for (int i = 0; i < 9; i++) H[i] = Hinv[i];
  /* NaiveHCheck(ransac_corresp, H, 10.0) */
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
  /* H_LAF_check(list, Hloran, 3*HLAFCoef*err_threshold, HDsSymMax), k_sigma = 3 (matching.cpp:172) */
  const double affErr = 3.0 * HLAFCoef * err_threshold;
  std::vector<int> kept;
  if (affErr > 0) {
    for (int i : ril) {
      double u[18], err[3], lin[18];
      const double *A = laf1 + 5 * i, *B = laf2 + 5 * i;
      u[0] = pts[4 * i]; u[1] = pts[4 * i + 1]; u[2] = 1.0;
      u[3] = pts[4 * i + 2]; u[4] = pts[4 * i + 3]; u[5] = 1.0;
      u[6] = u[0] + 3.0 * A[1] * A[4]; u[7] = u[1] + 3.0 * A[3] * A[4]; u[8] = 1.0;
      u[9] = u[3] + 3.0 * B[1] * B[4]; u[10] = u[4] + 3.0 * B[3] * B[4]; u[11] = 1.0;
      u[12] = u[0] + 3.0 * A[0] * A[4]; u[13] = u[1] + 3.0 * A[2] * A[4]; u[14] = 1.0;
      u[15] = u[3] + 3.0 * B[0] * B[4]; u[16] = u[4] + 3.0 * B[2] * B[4]; u[17] = 1.0;
      g_HDsSymMax(lin, u, Hloran, err, 3);
      double sumErr = sqrt(err[0] + err[1] + err[2]);
      if (!(sumErr > affErr)) kept.push_back(i);
    }
  } else kept = ril;
  if ((int)kept.size() < 8) kept.clear();
  for (int i : kept) keep[i] = 1;
  return (int)kept.size();
}
/* LORANSACFiltering with useF = 1, matching/matching.cpp:806-980: the reference's exp_ransacFcustom (oracle/_ref,
 * fixed seed through ref_shim.c) followed by F_LAF_check (:193-250, k_sigma = 3).  error_type 0 = SAMPSON
 * (FDs / exFDs), otherwise FDsSym / exFDsSym (:821-846).  data_out3: samples, LO runs, unused. */
int orc_loransac_f(const double *pts, const double *laf1, const double *laf2, int T, double err_threshold,
                   double confidence, int max_samples, int lo, double LAFCoef, int doSymmCheck, int error_type,
                   unsigned seed, double *F, unsigned char *inl, unsigned char *keep, int *data_out3) {
  for (int i = 0; i < T; i++) { inl[i] = 0; keep[i] = 0; }
  for (int i = 0; i < 9; i++) F[i] = 0;
  data_out3[0] = data_out3[1] = data_out3[2] = 0;
  if (!load_ref()) return -1;
  if (T < 8) return 0;
  std::vector<double> u2((size_t)T * 6);
  for (int i = 0; i < T; i++) {
    u2[6 * i] = pts[4 * i]; u2[6 * i + 1] = pts[4 * i + 1]; u2[6 * i + 2] = 1.;
    u2[6 * i + 3] = pts[4 * i + 2]; u2[6 * i + 4] = pts[4 * i + 3]; u2[6 * i + 5] = 1.;
  }
  std::vector<int> data_out((size_t)T * 18 + 18);
  double *resids = nullptr;
  double HinF[9];
  int I_H = 0;
  FDsPtr fds = error_type == 0 ? g_FDs : g_FDsSym;
  exFDsPtr exfds = error_type == 0 ? g_exFDs : g_exFDsSym;
  g_set_seed(seed);
  g_ransacF(u2.data(), T, err_threshold * err_threshold, confidence, max_samples, F, inl, data_out.data(), lo, 0,
            &resids, HinF, &I_H, exfds, fds, doSymmCheck);
  free(resids);
  data_out3[0] = data_out[0]; data_out3[1] = data_out[1];
  const double affErr = LAFCoef * err_threshold;
  std::vector<int> kept;
  for (int i = 0; i < T; i++) {
    if (!inl[i]) continue;
    if (affErr > 0) {
      double u[18], err[3];
      const double *A = laf1 + 5 * i, *B = laf2 + 5 * i;
      u[0] = pts[4 * i]; u[1] = pts[4 * i + 1]; u[2] = 1.0;
      u[3] = pts[4 * i + 2]; u[4] = pts[4 * i + 3]; u[5] = 1.0;
      u[6] = u[0] + 3.0 * A[1] * A[4]; u[7] = u[1] + 3.0 * A[3] * A[4]; u[8] = 1.0;
      u[9] = u[3] + 3.0 * B[1] * B[4]; u[10] = u[4] + 3.0 * B[3] * B[4]; u[11] = 1.0;
      u[12] = u[0] + 3.0 * A[0] * A[4]; u[13] = u[1] + 3.0 * A[2] * A[4]; u[14] = 1.0;
      u[15] = u[3] + 3.0 * B[0] * B[4]; u[16] = u[4] + 3.0 * B[2] * B[4]; u[17] = 1.0;
      fds(u, F, err, 3);
      const double sumErr = sqrt(err[0]) + sqrt(err[1]) + sqrt(err[2]);
      if (sumErr > affErr) continue;
    }
    kept.push_back(i);
  }
  if ((int)kept.size() < 8) kept.clear();
  for (int i : kept) keep[i] = 1;
  return (int)kept.size();
}
}
