// ransac_shim.cpp -- the reference's own C entry points of the verification stage, signature for signature, so that an
// application that links libmodsx instead of libdegensac resolves the same symbols (link-time drop-in):
//
//   Score exp_ransacHcustom(double *u, int len, double th, double conf, int max_sam, double *H, unsigned char *inl,
//                           int iter_type, int *data_out, int oriented_constraint, unsigned inlLimit, double **resids,
//                           HDsPtr HDS1, HDsiPtr HDSi1, HDsidxPtr HDSidx1, int doSymCheck)      degensac/exp_ranH.h:29-36
//   int exp_ransacFcustom(double *u, int len, double th, double conf, int max_sam, double *F, unsigned char *inl,
//                         int *data_out, int do_lo, unsigned inlLimit, double **resids, double *H_best, int *Ih,
//                         exFDsPtr EXFDS1, FDsPtr FDS1, int doSymCheck)                         degensac/exp_ranF.h:67-72
//
// as LORANSACFiltering calls them (matching/matching.cpp:883, 891).  The error functions the caller passes are the
// library's own HDs / HDsi / HDsidx (Sampson; degensac/Htools.c:158-196, 284-320, 418-456), HDsSym / HDsiSym / HDsSymidx
// (SYMM_SUM, the RANSACPars default) or HDsSymMax / HDsiSymMax / HDsSymidxMax (SYMM_MAX) and FDs / exFDs or
// FDsSym / exFDsSym (Ftools.c:82-210), exported here with the reference's signatures and data layout; they select the
// error type of the restated RANSAC (ransac.cpp, ransac_f.cpp).  Foreign function pointers cannot be honoured by a
// restatement and are refused loudly (zero inliers + modsx_last_error()), as is an iter_type other than 4.
//   * `*resids` is a malloc'd block the caller frees (matching.cpp:884, 892 free it unread); it is zero-filled, the
//     per-iteration residual log of the reference is not reproduced.
//   * H_best is never written (the reference's own loop over it is `for (a = 0; a < 0; a++)`), *Ih is set to 0.
//   * the sample stream is seeded like srand(time(NULL)) (exp_ranH.c:823, exp_ranF.c:822) unless modsx_ransac_set_seed()
//     or the environment variable MODSX_RANSAC_SEED fixes it.
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <atomic>
#include "engine_api.hpp"

extern "C" {

typedef struct { unsigned I; double J; } Score;                                             /* degensac/rtools.h:17-24 */
typedef void (*HDsPtr)(const double *, const double *, const double *, double *, int);      /* degensac/Htools.h:1-3 */
typedef void (*HDsiPtr)(const double *, const double *, const double *, double *, int, int *, int);
typedef void (*HDsidxPtr)(const double *, const double *, const double *, double *, int, int *, int);
typedef void (*FDsPtr)(const double *, const double *, double *, int);                      /* degensac/Fcustomdef.h:3-4 */
typedef void (*exFDsPtr)(const double *, const double *, double *, double *, int);

static std::atomic<unsigned> g_seed(0);
static std::atomic<int> g_seed_set(0);
void modsx_ransac_set_seed(unsigned seed, int enable) { g_seed = seed; g_seed_set = enable; }
static unsigned shim_seed() {
  if (g_seed_set.load()) return g_seed.load();
  if (const char *e = getenv("MODSX_RANSAC_SEED")) return (unsigned)strtoul(e, nullptr, 10);
  return (unsigned)time(nullptr);
}

// Sampson error of one correspondence under H; `lin` in the reference's layout: entry j of row r of the linearisation at
// lin[r + j * 2 * len] (lin_hg, Htools.c:17-47)
static inline double sampson_h(const double *lin, int len, int i, const double *u, const double *H) {
  const int shift = 2 * len;
  const double *l = lin + 2 * i;
  double r1 = 0, r2 = 0;
  for (int j = 0; j < 9; j++) { r1 += H[j] * *l; r2 += H[j] * l[1]; l += shift; }
  const double a = H[0] - H[2] * u[0], b = H[3] - H[5] * u[0], c = -H[8] - H[2] * u[3] - H[5] * u[4];
  const double d = H[1] - H[2] * u[1], e = H[4] - H[5] * u[1];
  // pinvJ, Htools.c:126-156
  double pJ[8];
  const double a2 = a * a, b2 = b * b, c2 = c * c, d2 = d * d, e2 = e * e, c2pd2 = c2 + d2, ab = a * b, de = d * e;
  pJ[0] = -b * de + a * (c2 + e2);
  pJ[1] = b * c2pd2 - a * de;
  pJ[2] = c * (c2pd2 + e2);
  pJ[3] = -c * (a * d + b * e);
  pJ[4] = d * (b2 + c2) - ab * e;
  pJ[5] = -ab * d + e * (a2 + c2);
  pJ[6] = pJ[3];
  pJ[7] = c * (a2 + b2 + c2);
  const double N = a * pJ[0] + b * pJ[1] + c * pJ[2];
  for (int q = 0; q < 8; q++) pJ[q] /= N;
  double acc = 0;
  for (int j = 0; j < 4; j++) { const double t = pJ[j] * r1 + pJ[j + 4] * r2; acc += t * t; }
  return acc;
}
void HDs(const double *lin, const double *u, const double *H, double *p, int len) {
  for (int i = 0; i < len; i++) p[i] = sampson_h(lin, len, i, u + 6 * i, H);
}
void HDsi(const double *lin, const double *u6, const double *H, double *p, int len, int *pts, int ni) {
  for (int i = 0; i < ni; i++) p[i] = sampson_h(lin, len, pts[i], u6 + 6 * pts[i], H);
}
void HDsidx(const double *lin, const double *u6, const double *H, double *p, int len, int *idx, int siz) {
  for (int i = 0; i < siz; i++) p[i] = sampson_h(lin, len, idx[i], u6 + 6 * idx[i], H);
}

/* symmetric transfer error, sum and max of the two directions (Htools.c:199-279, 325-411, 458-535); `lin` is unused */
void HDsSym(const double *lin, const double *u, const double *H, double *p, int len) { (void)lin; mx::hds_sym(u, H, p, len, false); }
void HDsSymMax(const double *lin, const double *u, const double *H, double *p, int len) { (void)lin; mx::hds_sym(u, H, p, len, true); }
static void hds_sym_subset(const double *u6, const double *H, double *p, const int *pts, int ni, bool takeMax) {
  double HT[9], H1[9];
  mx::hds_sym_setup(H, HT, H1);
  for (int i = 0; i < ni; i++) mx::hds_sym_with(u6 + 6 * pts[i], HT, H1, p + i, 1, takeMax);
}
void HDsiSym(const double *lin, const double *u6, const double *H, double *p, int len, int *pts, int ni) { (void)lin; (void)len; hds_sym_subset(u6, H, p, pts, ni, false); }
void HDsiSymMax(const double *lin, const double *u6, const double *H, double *p, int len, int *pts, int ni) { (void)lin; (void)len; hds_sym_subset(u6, H, p, pts, ni, true); }
void HDsSymidx(const double *lin, const double *mu, const double *H, double *p, int len, int *idx, int siz) { (void)lin; (void)len; hds_sym_subset(mu, H, p, idx, siz, false); }
void HDsSymidxMax(const double *lin, const double *mu, const double *H, double *p, int len, int *idx, int siz) { (void)lin; (void)len; hds_sym_subset(mu, H, p, idx, siz, true); }

static inline void f_terms(const double *u, const double *F, double &r, double &a, double &b) {
  const double rxc = F[0] * u[3] + F[3] * u[4] + F[6], ryc = F[1] * u[3] + F[4] * u[4] + F[7];
  const double rwc = F[2] * u[3] + F[5] * u[4] + F[8];
  r = u[0] * rxc + u[1] * ryc + rwc;
  const double rx = F[0] * u[0] + F[1] * u[1] + F[2], ry = F[3] * u[0] + F[4] * u[1] + F[5];
  a = rxc * rxc + ryc * ryc;
  b = rx * rx + ry * ry;
}
void FDs(const double *u, const double *F, double *p, int len) {                   /* Ftools.c:82-107 */
  for (int i = 0; i < len; i++, u += 6) { double r, a, b; f_terms(u, F, r, a, b); p[i] = r * r / (a + b); }
}
void FDsSym(const double *u, const double *F, double *p, int len) {                /* Ftools.c:109-131 */
  for (int i = 0; i < len; i++, u += 6) { double r, a, b; f_terms(u, F, r, a, b); p[i] = r * r * (a + b) / (a * b); }
}
void exFDs(const double *u, const double *F, double *p, double *w, int len) {      /* Ftools.c:162-185 */
  for (int i = 0; i < len; i++, u += 6) { double r, a, b; f_terms(u, F, r, a, b); w[i] = a + b; p[i] = r * r / w[i]; w[i] = 1 / sqrt(w[i]); }
}
void exFDsSym(const double *u, const double *F, double *p, double *w, int len) {   /* Ftools.c:186-210 */
  for (int i = 0; i < len; i++, u += 6) {
    double r, a, b;
    f_terms(u, F, r, a, b);
    w[i] = (a * b) / (a + b);
    p[i] = r * r / w[i];
    w[i] = 1 / sqrt(w[i]);
  }
}

static double *alloc_resids(int len) {
  // RESIDS_M = 2 + RAN_REP * (1 + ILSQ_ITERS + 1) doubles per point and logged iteration (rtools.h:15); one block is enough
  // for a caller that only frees it
  return (double *)calloc((size_t)(len > 0 ? len : 1) * 64, sizeof(double));
}

Score exp_ransacHcustom(double *u, int len, double th, double conf, int max_sam, double *H, unsigned char *inl, int iter_type,
                        int *data_out, int oriented_constraint, unsigned inlLimit, double **resids, HDsPtr HDS1,
                        HDsiPtr HDSi1, HDsidxPtr HDSidx1, int doSymCheck) {
  Score S = {0, 0};
  if (resids) *resids = alloc_resids(len);
  if (!u || !H || !inl || !data_out || len < 4) { mx::set_error("exp_ransacHcustom: bad argument"); return S; }
  memset(inl, 0, (size_t)len);
  int error_type;   // the three triples LORANSACFiltering can pass (matching.cpp:821-846)
  if (HDS1 == &HDs && HDSi1 == &HDsi && HDSidx1 == &HDsidx) error_type = 0;
  else if (HDS1 == &HDsSymMax && HDSi1 == &HDsiSymMax && HDSidx1 == &HDsSymidxMax) error_type = 1;
  else if (HDS1 == &HDsSym && HDSi1 == &HDsiSym && HDSidx1 == &HDsSymidx) error_type = 2;
  else {
    mx::set_error("exp_ransacHcustom: only the library's own error functions are supported (HDs/HDsi/HDsidx, "
                  "HDsSym/HDsiSym/HDsSymidx, HDsSymMax/HDsiSymMax/HDsSymidxMax)");
    return S;
  }
  if (iter_type != 4 || inlLimit != 0) {
    mx::set_error("exp_ransacHcustom: only iter_type 4 with inlLimit 0 (LORANSACFiltering's call, matching.cpp:891) is supported");
    return S;
  }
  double J = 0;
  const int n = mx::ransac_h(u, len, th, conf, max_sam, H, inl, data_out, oriented_constraint, doSymCheck, shim_seed(), &J, error_type);
  S.I = n < 0 ? 0u : (unsigned)n;
  S.J = J;
  return S;
}

int exp_ransacFcustom(double *u, int len, double th, double conf, int max_sam, double *F, unsigned char *inl, int *data_out,
                      int do_lo, unsigned inlLimit, double **resids, double *H_best, int *Ih, exFDsPtr EXFDS1, FDsPtr FDS1,
                      int doSymCheck) {
  (void)H_best;
  if (resids) *resids = alloc_resids(len);
  if (Ih) *Ih = 0;
  if (!u || !F || !inl || !data_out || len < 8) { mx::set_error("exp_ransacFcustom: bad argument"); return 0; }
  memset(inl, 0, (size_t)len);
  int error_type;
  if (FDS1 == &FDs && EXFDS1 == &exFDs) error_type = 0;
  else if (FDS1 == &FDsSym && EXFDS1 == &exFDsSym) error_type = 1;
  else { mx::set_error("exp_ransacFcustom: only the library's FDs/exFDs or FDsSym/exFDsSym are supported"); return 0; }
  const int n = mx::ransac_f(u, len, th, conf, max_sam, F, inl, data_out, do_lo, inlLimit, error_type, doSymCheck, shim_seed());
  return n < 0 ? 0 : n;
}

}  // extern "C"
