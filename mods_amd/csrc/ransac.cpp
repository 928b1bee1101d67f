// ransac.cpp -- LO-RANSAC for homographies (host C++), the verifier the device path feeds.
//
// Restates exp_ransacHcustom (degensac/exp_ranH.c:796-1236) for the configuration the reference
// compiles and calls: iter_type 4, __D3__ (ratio 1 => LSQ on all inliers), __LSQ_BEFORE_LO__,
// __HASHING__, MSAC scoring (__SCORE__ == SC_M, MWM = 9/4 = 2 as an integer expression), Sampson
// error HDs.  Differences by design: the libc PRNG (srand/rand/random) is an explicit, re-entrant
// copy of glibc's TYPE_3 additive-feedback generator seeded by the caller instead of time(NULL);
// the inlier-set hash table is per call instead of the global HASH_TABLE; the 9x9 symmetric
// eigenproblem of the normalised DLT uses a cyclic Jacobi solver instead of LAPACK dsyev_
// (same eigenvector up to sign and ~1e-15 relative error).
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include "engine_api.hpp"

#include "ransac_common.hpp"

namespace mx {

// HDsSym / HDsSymMax, Htools.c:199-279.  The reference inverts H inside every call; hds_sym_setup + hds_sym_with are the same
// call split in two for callers that evaluate many small point sets against one H (the LAF check of the verifier: three
// points per call) -- the inverse of the same matrix is the same nine doubles every time.
void hds_sym_setup(const double *H, double *Hinv, double *H1) {
  const double Ht[9] = {H[0], H[3], H[6], H[1], H[4], H[7], H[2], H[5], H[8]};
  memcpy(Hinv, Ht, sizeof Ht);
  if (!inv3_lu(Hinv, H1)) memcpy(H1, Hinv, sizeof Ht);  // minv leaves the matrix on a singular pivot
}
void hds_sym_with(const double *u, const double *Hinv, const double *H1, double *p, int len, bool takeMax) {
  for (int i = 0; i < len; i++) {
    double a = H1[6] * u[0] + H1[7] * u[1] + H1[8];
    double b = Hinv[6] * u[3] + Hinv[7] * u[4] + Hinv[8];
    double xa = (H1[0] * u[0] + H1[1] * u[1] + H1[2]) / a;
    double ya = (H1[3] * u[0] + H1[4] * u[1] + H1[5]) / a;
    double xd = u[3] - xa, yd = u[4] - ya;
    double d1 = xd * xd + yd * yd;
    xa = (Hinv[0] * u[3] + Hinv[1] * u[4] + Hinv[2]) / b;
    ya = (Hinv[3] * u[3] + Hinv[4] * u[4] + Hinv[5]) / b;
    xd = u[0] - xa; yd = u[1] - ya;
    double d2 = xd * xd + yd * yd;
    *p++ = takeMax ? (d1 < d2 ? d2 : d1) : d1 + d2;
    u += 6;
  }
}
void hds_sym(const double *u, const double *H, double *p, int len, bool takeMax) {
  double Hinv[9], H1[9];
  hds_sym_setup(H, Hinv, H1);
  hds_sym_with(u, Hinv, H1, p, len, takeMax);
}

struct Ransac {
  const double *u;
  int len;
  std::vector<double> buffer, err;
  double *errs[5];
  HashTable ht;
  GlibcRandom rng;
  int errType = 0;   // the HDS1 the caller chose: 0 HDs (Sampson), 1 HDsSymMax, 2 HDsSym (matching.cpp:821-846)
  void errf(const double *h, double *d) {
    if (errType == 0) HDs(nullptr, u, h, d, len);
    else hds_sym(u, h, d, len, errType == 1);
  }

  // randsubset, rtools.c:27-41
  int *randsubset(int *pool, int max_sz, int siz) {
    for (int i = 0; i < siz; i++) {
      int s = (int)(rng.next() % (max_sz - i));
      int j = max_sz - i - 1;
      int q = pool[s]; pool[s] = pool[j]; pool[j] = q;
    }
    return pool + max_sz - siz;
  }

  // exp_iterHcustom, exp_ranH.c:617-737
  Score iterH(int *inliers, double th, double ths, int steps, double *H, int iterID) {
    double *d = errs[1];
    double h[9];
    Score maxS = {0, 0}, S = {0, 0}, Ss;
    double dth = (ths - th) / (steps);
    maxS = inlidxs(errs[4], len, th, inliers);
    if (maxS.I < 4) return S;
    S.I = inl_list(errs[4], len, th * 2, inliers);   // only the count and the indices of these wider sets are read
    memcpy(h, H, sizeof h);  // defined start value; overwritten below since S.I >= 4
    u2h(u, inliers, S.I, h, buffer.data());
    for (int it = 0; it < steps; it++) {
      errf(h, d);
      Ss = inlidxs(d, len, th, inliers);
      uint32_t hash = super_fast_hash((const char *)inliers, (int)(Ss.I * sizeof(int)));
      int ret = ht.contains(hash, (int)Ss.I, iterID);
      if (ret != -1 && ret != iterID) { S.I = 0; S.J = 0; return S; }
      if (ret == -1) ht.insert(hash, (int)Ss.I, iterID);
      S.I = inl_list(d, len, ths * 2, inliers);
      if (score_less(maxS, Ss)) {
        maxS = Ss;
        errs[1] = errs[0]; errs[0] = d; d = errs[1];
        memcpy(H, h, 9 * sizeof(double));
      }
      if (S.I < 4) return maxS;
      u2h(u, inliers, S.I, h, buffer.data());
      ths -= dth;
    }
    errf(h, d);
    S = inlidxs(d, len, th, inliers);
    if (score_less(maxS, S)) {
      maxS = S;
      errs[1] = errs[0]; errs[0] = d;
      memcpy(H, h, 9 * sizeof(double));
    }
    return maxS;
  }

  // exp_inHranicustom, exp_ranH.c:741-793
  Score inHrani(int *inliers, int ninl, double th, double *H, int rep, int *iterID) {
    Score S, maxS = {0, 0};
    double h[9];
    std::vector<int> intbuff(len);
    if (ninl < 8) return maxS;
    int ssiz = ninl / 2;
    if (ssiz > 12) ssiz = 12;
    double *d = errs[2]; errs[2] = errs[0]; errs[0] = d;
    memcpy(h, H, sizeof h);
    for (int i = 0; i < rep; i++) {
      int *sample = randsubset(inliers, ninl, ssiz);
      u2h(u, sample, ssiz, h, buffer.data());
      errf(h, errs[0]);
      errs[4] = errs[0];
      S = iterH(intbuff.data(), th, 4 * th, 4, h, ++*iterID);
      if (score_less(maxS, S)) {
        maxS = S;
        d = errs[2]; errs[2] = errs[0]; errs[0] = d;
        memcpy(H, h, 9 * sizeof(double));
      }
    }
    d = errs[2]; errs[2] = errs[0]; errs[0] = d;
    return maxS;
  }
};

// exp_ransacHcustom, exp_ranH.c:796-1236 (iter_type 4; HDS1 = HDs, HDsSymMax or HDsSym by error_type)
int ransac_h(const double *u, int len, double th, double conf, int max_sam, double *H, unsigned char *inl, int *data_out,
             int oriented_constraint, int doSymCheck, unsigned seed0, double *scoreJ, int error_type) {
  Ransac R;
  R.u = u; R.len = len; R.errType = error_type;
  std::vector<int> pool(len), inliers(len);
  double M[81], sol[81], *h = sol;
  memset(sol, 0, sizeof sol);
  Score maxS = {0, 0}, maxSs = {0, 0}, S = {0, 0};
  int iter_cnt = 0, no_rej = 0, iterID = 0, no_sam = 0;
  char new_max = 0;
  int bad_model = 0;
  int nb[18];
  for (int i = 0; i < 9; i++) H[i] = 0;
  R.rng.seed(seed0);
  for (int i = 0; i < len; i++) pool[i] = i;
  // the reference linearises all points once (lin_hg into Z) and reads Z for the samples and inside HDs; both re-form the few
  // products they need from u instead -- the same values -- which saves building and streaming 144 bytes per point
  // (u2h no longer builds the 2len x 9 matrix: no buffer)
  R.err.assign((size_t)len * 4, 0.0);
  std::vector<double> d_check(len);
  for (int i = 0; i < 4; i++) R.errs[i] = R.err.data() + (size_t)i * len;
  R.errs[4] = R.errs[3];
  unsigned seed = (unsigned)R.rng.next();
  int *samidx = pool.data() + len - 4;
  int bestsamidx[4] = {0, 0, 0, 0};
  (void)bestsamidx;
  double *d;

  auto score_of = [&](const double *dd) {
    Score s = {0, 0};
    double term[256];
    for (int j0 = 0; j0 < len; j0 += 256) {      // terms in a vectorisable loop, the f64 sum in point order
      const int m = len - j0 < 256 ? len - j0 : 256;
      for (int j = 0; j < m; j++) term[j] = trunc_quad(dd[j0 + j], th);
      for (int j = 0; j < m; j++) { if (dd[j0 + j] <= th) s.I++; s.J += term[j]; }
    }
    return s;
  };
  auto sym_bad = [&](const double *hh) {
    hds_sym(u, hh, d_check.data(), len, false);
    unsigned cnt = 0;
    const double th_check = 9.0 * th;
    for (int j = 0; j < len; j++) if (d_check[j] <= th_check) cnt++;
    return cnt <= 5 ? 1 : 0;
  };
  auto tol_of = [&](const double *hh) {
    double tol = hh[8];
    if (tol == 0) {
      for (int i = 0; i < 9; ++i) tol += hh[i] * hh[i];
      tol = sqrt(tol);
      tol *= 0.001;
    }
    return tol * tol * tol;
  };
  auto local_opt = [&]() {  // case 4 with __LSQ_BEFORE_LO__
    d = R.errs[0];
    S.I = inl_list(R.errs[4], len, 4 * th * 2, inliers.data());
    u2h(u, inliers.data(), S.I, h, R.buffer.data());
    R.errf(h, d);
    S.I = inl_list(d, len, th, inliers.data());
    S = R.inHrani(inliers.data(), (int)S.I, th, h, 10, &iterID);
  };

  while (no_sam < max_sam) {
    no_sam++;
    R.rng.seed(seed);
    // multirsampleT(Z, 9, 2, pool, 4, len, M), rtools.c:136-156 with sample() :14-25
    for (int i = 0; i < 4; i++) {
      int s = (int)(R.rng.next() % (len - i));
      int j = len - i - 1;
      int q = pool[s]; pool[s] = pool[j]; pool[j] = q;
      const double *sq = u + 6 * q;     // the row pair of point q, lin_hg (Htools.c:17-47)
      double *p = M + i * 18;
      for (int j = 0; j < 3; j++) {
        p[3 * j] = sq[3 + j]; p[3 * j + 1] = 0; p[3 * j + 2] = -sq[0] * sq[3 + j];
        p[9 + 3 * j] = 0; p[9 + 3 * j + 1] = sq[3 + j]; p[9 + 3 * j + 2] = -sq[1] * sq[3 + j];
      }
    }
    seed = (unsigned)R.rng.next();
    if (oriented_constraint && !all_hori_valid(u, samidx)) { no_rej++; continue; }
    for (int i = 72; i < 81; ++i) M[i] = 0.0;
    int nullsize = nullspace(M, sol, 9, nb);
    if (nullsize != 1) continue;
    double v = det3(h);
    double tol = tol_of(h);
    if (fabs(v / tol) < 10e-2) continue;
    d = R.errs[0];
    R.errf(h, d);
    S = score_of(d);
    int do_iterate;
    if (score_less(maxS, S)) {
      if (doSymCheck) bad_model = sym_bad(h);
      if (bad_model) continue;
      R.errs[0] = R.errs[3];
      R.errs[3] = d;
      maxS = S;
      new_max = 1;
      memcpy(H, h, 9 * sizeof(double));
    }
    if (score_less(maxSs, S)) {
      do_iterate = no_sam > 50;
      maxSs = S;
      R.errs[4] = d;
      memcpy(bestsamidx, samidx, 4 * sizeof(int));
    } else do_iterate = 0;
    if ((no_sam >= 50) && (iter_cnt == 0) && (maxSs.I > 4)) do_iterate = 1;
    if (do_iterate) {
      iter_cnt++;
      local_opt();
      double tl = tol_of(h);
      if (score_less(maxS, S) && (fabs(det3(h) / tl) > 10e-2)) {
        if (doSymCheck) bad_model = sym_bad(h);
        if (!bad_model) {
          d = R.errs[0]; R.errs[0] = R.errs[3]; R.errs[3] = d;
          maxS = S;
          new_max = 1;
          memcpy(H, h, 9 * sizeof(double));
        }
      }
    }
    if (new_max) {
      int new_sam = nsamples(maxS.I + 1, len, 4, conf);
      if (new_sam < max_sam) max_sam = new_sam;
      new_max = 0;
    }
  }
  if (iter_cnt == 0) {
    iter_cnt++;
    local_opt();
    double tl = tol_of(h);
    if (score_less(maxS, S) && (fabs(det3(h) / tl) > 10e-2)) {
      if (doSymCheck) bad_model = sym_bad(h);
      if (!bad_model) {
        d = R.errs[0]; R.errs[0] = R.errs[3]; R.errs[3] = d;
        maxS = S;
        memcpy(H, h, 9 * sizeof(double));
      }
    }
  }
  d = R.errs[3];
  int ninl = 0;
  for (int j = 0; j < len; j++) { inl[j] = d[j] <= th ? 1 : 0; ninl += inl[j]; }
  data_out[0] = no_sam; data_out[1] = iter_cnt; data_out[2] = no_rej;
  if (scoreJ) *scoreJ = maxS.J;
  return ninl;
}

}  // namespace mx
