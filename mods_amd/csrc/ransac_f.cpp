// ransac_f.cpp -- LO-RANSAC / DEGENSAC for the fundamental matrix (host C++).
//
// Restates exp_ransacFcustom (degensac/exp_ranF.c:795-1192) in the configuration the reference compiles:
// __DEGEN__ (plane-and-parallax test of every so-far-the-best sample), __D3__ with the inlLimit = 0 the
// caller passes (matching.cpp:883 => every local-optimisation LSQ runs on a random subset of 8 inliers),
// __LSQ_BEFORE_LO__, __HASHING__ (exp_ranF.c:21), oriented constraint on, MSAC scoring.  Helpers follow
// Ftools.c, DegUtils.c and ranH.c (file:line at each function).
//
// Differences by design (same as ransac.cpp): the libc PRNG is an explicit re-entrant copy of glibc's
// generator seeded by the caller instead of time(NULL); the hash table is per call; LAPACK dsyev_/dgesvd_
// and ccmath svduv are replaced by a cyclic Jacobi eigen-solver (null vectors and the rank-2 projection are
// unique up to sign, agreement ~1e-15).  Reads of uninitialised memory in the reference (u2f with fewer than
// 8 points, exp_ranF.c ALO branch indexing errs[i] past the loop) are replaced by the intended computation.
#include <algorithm>
#include <atomic>
#include <chrono>
#include <mutex>
#include "engine_api.hpp"
#include "ransac_common.hpp"

// -DMODSX_TRACE_RANSAC: the trajectory of exp_ransacFcustom on stderr, in the format of the tracing build of the reference
// (tools/trace_degensac.sh wraps the same call sites of exp_ranF.c through -D renames); for diffing the two, not shipped
#ifdef MODSX_TRACE_RANSAC
#include <cstdio>
#define RTRACE(...) fprintf(stderr, __VA_ARGS__)
#else
#define RTRACE(...) ((void)0)
#endif
namespace mx {
// host profile of the F verification (builds with -DMODSX_HPROF only; tools/hprof_f.py): nanoseconds and calls per section
#ifdef MODSX_HPROF
static std::atomic<long> g_hprof[2][24];
struct HProfScope {
  int id; std::chrono::steady_clock::time_point t0;
  explicit HProfScope(int i) : id(i), t0(std::chrono::steady_clock::now()) {}
  ~HProfScope() { g_hprof[0][id] += std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count(); g_hprof[1][id]++; }
};
#define HPROF(id) HProfScope hprofScope_##id(id)
#else
#define HPROF(id)
#endif

typedef void (*FdsFn)(const double *, const double *, double *, int);
typedef void (*ExFdsFn)(const double *, const double *, double *, double *, int);

// ---- symmetric Jacobi eigen-solver: jacobi_eig_rule of ransac_common.hpp (eigenvalues in ev, eigenvectors in the columns of V) ----
#ifdef MODSX_JACOBI_CHECK
static std::atomic<long> g_jc[8];
struct JcReport { ~JcReport() { fprintf(stderr, "jacobi check: calls %ld, rule1 differs %ld, rule2 differs %ld, sweeps old %ld rule1 %ld rule2 %ld\n", g_jc[0].load(), g_jc[1].load(), g_jc[2].load(), g_jc[3].load(), g_jc[4].load(), g_jc[5].load()); } } g_jcReport;
#endif
template <int N>
static void jacobi_eig_n(const double *C, double *ev, double *V) {
#ifdef MODSX_JACOBI_CHECK
  double e1[N], V1[N * N], e2[N], V2[N * N];
  const int s0 = jacobi_eig_rule<N, 0>(C, ev, V), s1 = jacobi_eig_rule<N, 1>(C, e1, V1), s2 = jacobi_eig_rule<N, 2>(C, e2, V2);
  int m0 = 0, m1 = 0, m2 = 0;
  for (int i = 1; i < N; i++) { if (ev[i] < ev[m0]) m0 = i; if (e1[i] < e1[m1]) m1 = i; if (e2[i] < e2[m2]) m2 = i; }
  bool d1 = m0 != m1 || ev[m0] != e1[m1], d2 = m0 != m2 || ev[m0] != e2[m2];
  for (int k = 0; k < N; k++) { d1 = d1 || V[k * N + m0] != V1[k * N + m1]; d2 = d2 || V[k * N + m0] != V2[k * N + m2]; }
  g_jc[0]++; g_jc[1] += d1; g_jc[2] += d2; g_jc[3] += s0; g_jc[4] += s1; g_jc[5] += s2;
#else
  jacobi_eig_rule<N, JACOBI_RULE>(C, ev, V);
#endif
}
static void jacobi_eig(const double *C, int n, double *ev, double *V) {   // the callers' sizes: 9 (u2f, left_null9) and 3 (singulF)
  if (n == 9) return jacobi_eig_n<9>(C, ev, V);
  if (n == 3) return jacobi_eig_n<3>(C, ev, V);
  abort();
}

static inline __host__ __device__ void cross3(double *o, const double *a, const double *b) {  // crossp, DegUtils.c:246-250
  o[0] = a[1] * b[2] - a[2] * b[1];
  o[1] = a[2] * b[0] - a[0] * b[2];
  o[2] = a[0] * b[1] - a[1] * b[0];
}
static inline __host__ __device__ void skew_sym(const double *a, double *ax) {  // DegUtils.c:212-222
  ax[0] = 0; ax[1] = -a[2]; ax[2] = a[1];
  ax[3] = a[2]; ax[4] = 0; ax[5] = -a[0];
  ax[6] = -a[1]; ax[7] = a[0]; ax[8] = 0;
}
static inline __host__ __device__ void mul3(double *o, const double *a, const double *b) {  // mmul, row-major 3x3
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) {
      double s = 0;
      for (int k = 0; k < 3; k++) s += a[i * 3 + k] * b[k * 3 + j];
      o[i * 3 + j] = s;
    }
}
static inline __host__ __device__ void tr3(double *o, const double *a) {
  for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) o[j * 3 + i] = a[i * 3 + j];
}

// Singular values and right singular vectors of a 3x3 exactly in the order ccmath's svduv leaves them
// (matutls/svduv.c + ldvmat.c + qrbdv.c: Householder bidiagonalisation, then implicit-shift QR sweeps on the
// bidiagonal).  The values are NOT sorted, and Hdetect (DegUtils.c:100-101) takes the third column of V whatever
// singular value it belongs to, so the order has to be reproduced; the left vectors are never used and the
// rotations that only touch them are dropped.
static void svd3_unsorted(const double *Ain, double *d, double *V) {
  double a[3][3], e[3] = {0, 0, 0};
  for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) a[i][j] = Ain[i * 3 + j];
  double beta = 0, g = 0;  // right reflector of step 0: I - beta (0,1,g)(0,1,g)^T
  for (int i = 0; i < 3; i++) {
    const int mm = 3 - i, nm = 2 - i;
    if (mm > 1) {  // left reflector annihilating a[i+1..][i]
      double w[3], ss = 0, h = 0;
      for (int j = 0; j < mm; j++) { w[j] = a[i + j][i]; ss += w[j] * w[j]; }
      if (ss > 0) {
        h = sqrt(ss);
        if (a[i][i] < 0) h = -h;
        ss += a[i][i] * h;
        ss = 1. / ss;
        w[0] += h;
        for (int k = 1; k < 3 - i; k++) {
          double r = 0;
          for (int j = 0; j < mm; j++) r += w[j] * a[i + j][i + k];
          r *= ss;
          for (int j = 0; j < mm; j++) a[i + j][i + k] -= r * w[j];
        }
      }
      d[i] = -h;
    } else d[i] = a[i][i];
    if (nm > 1) {  // right reflector annihilating a[i][i+2..] (only i == 0 for a 3x3)
      double ss = 0, h = 0, sv = 0;
      for (int j = 0; j < nm; j++) ss += a[i][i + 1 + j] * a[i][i + 1 + j];
      if (ss > 0) {
        h = sqrt(ss);
        if (a[i][i + 1] < 0) h = -h;
        sv = 1. + fabs(a[i][i + 1] / h);
        ss += a[i][i + 1] * h;
        ss = 1. / ss;
        const double t = 1. / (a[i][i + 1] += h);
        for (int r2 = i + 1; r2 < 3; r2++) {
          double r = 0;
          for (int j = 0; j < nm; j++) r += a[i][i + 1 + j] * a[r2][i + 1 + j];
          r *= ss;
          for (int j = 0; j < nm; j++) a[r2][i + 1 + j] -= r * a[i][i + 1 + j];
        }
        for (int j = 1; j < nm; j++) a[i][i + 1 + j] *= t;
      }
      beta = sv; g = a[i][i + 2]; e[i] = -h;
    } else if (nm == 1) e[i] = a[i][i + 1];
  }
  for (int i = 0; i < 9; i++) V[i] = 0;
  V[0] = 1;
  if (beta != 0.) {
    const double bg = beta * g;
    V[4] = 1. - beta; V[5] = -bg; V[7] = -bg; V[8] = 1. - bg * g;
  } else { V[4] = 1; V[8] = 1; }
  // QR sweeps on the bidiagonal (d, e); columns of V rotated along
  int m = 3;
  double t = fabs(d[0]);
  for (int j = 1; j < 3; ++j) { const double q = fabs(d[j]) + fabs(e[j - 1]); if (q > t) t = q; }
  t *= 1.e-15;
  for (int it = 0; m > 1 && it < 300; ++it) {
    int k;
    for (k = m - 1; k > 0; --k) {
      if (fabs(e[k - 1]) < t) break;
      if (fabs(d[k - 1]) < t) {
        double sn = 1., cs = 0.;
        for (int i = k; i < m; ++i) {
          const double aa = sn * e[i - 1], bb = d[i];
          e[i - 1] *= cs;
          const double uu = sqrt(aa * aa + bb * bb);
          d[i] = uu; sn = -aa / uu; cs = bb / uu;
        }
        break;
      }
    }
    double y = d[k], x = d[m - 1], u = e[m - 2];
    double aa = (y + x) * (y - x) - u * u, sn = y * e[k], bb = sn + sn, cs = 0;
    u = sqrt(aa * aa + bb * bb);
    if (u != 0.) {
      cs = sqrt((u + aa) / (u + u));
      if (cs != 0.) sn /= (cs * u);
      else sn = 1.;
      for (int i = k; i < m - 1; ++i) {
        bb = e[i];
        if (i > k) {
          aa = sn * e[i]; bb *= cs;
          e[i - 1] = u = sqrt(x * x + aa * aa);
          cs = x / u; sn = aa / u;
        }
        aa = cs * y + sn * bb; bb = cs * bb - sn * y;
        for (int r = 0; r < 3; ++r) {
          const double w = cs * V[r * 3 + i] + sn * V[r * 3 + i + 1];
          V[r * 3 + i + 1] = cs * V[r * 3 + i + 1] - sn * V[r * 3 + i];
          V[r * 3 + i] = w;
        }
        sn *= d[i + 1];
        d[i] = u = sqrt(aa * aa + sn * sn);
        y = cs * d[i + 1]; cs = aa / u; sn /= u;
        x = cs * bb + sn * y; y = cs * y - sn * bb;
      }
    }
    e[m - 2] = x; d[m - 1] = y;
    if (fabs(x) < t) --m;
    if (m == k + 1) --m;
  }
  for (int i = 0; i < 3; ++i)
    if (d[i] < 0.) {
      d[i] = -d[i];
      for (int r = 0; r < 3; ++r) V[r * 3 + i] = -V[r * 3 + i];
    }
}

// ---- Ftools.c ------------------------------------------------------------------------------------------
// lin_fm, Ftools.c:13-35: column-wise 9 x len, entry (3k+l, i) = x2_k * x1_l
static void lin_fm(const double *u, double *p, const int *inl, int len) {
  for (int i = 0; i < len; i++) {
    const double *s = u + 6 * inl[i];
    for (int k = 0; k < 3; k++)
      for (int l = 0; l < 3; l++) p[(size_t)(3 * k + l) * len + i] = s[k + 3] * s[l];
  }
}
// lin_fmN, Ftools.c:263-290: row-wise len x 9 of the normalised points
static void lin_fmN(const double *u, double *p, const int *inl, int len, const double *A1, const double *A2) {
  double a[3], b[3];
  a[2] = 1; b[2] = 1;
  for (int i = 0; i < len; i++) {
    const double *s = u + 6 * inl[i];
    a[0] = s[0] * A1[0] + A1[1];
    a[1] = s[1] * A1[0] + A1[2];
    b[0] = s[3] * A2[0] + A2[1];
    b[1] = s[4] * A2[0] + A2[2];
    for (int k = 0; k < 3; k++)
      for (int l = 0; l < 3; l++) *p++ = a[l] * b[k];
  }
}
// lin_fmN + (the weights of u2fw) + cov_mat in one pass, without the len x 9 matrix in memory: row i of it is
// z[3 k + l] = a[l] * b[k] with a = (a0, a1, 1), b = (b0, b1, 1) the normalised points (times w[inl[i]] when weighted), and the 45
// sums of Z^T Z take their terms in row order, four columns at a time as in cov_mat -- the same products, the same additions, the
// same order: the matrix is bit-identical to cov_mat(lin_fmN(...)).  (1 * x is x: the products with a[2] = b[2] = 1 are not formed.)
// The same on 512-bit vectors where the CPU has them (row i = z[i] * (z[0] .. z[7]), column 8 apart: 9 accumulators instead of 14,
// which no longer spill out of the 256-bit register file; the lanes are the same sums)
#if defined(__x86_64__)
__attribute__((target("avx512f"))) static void cov_fmN_fused_w8(double *Cv, const double *u, const int *inl, int len, const double *A1, const double *A2,
                                                                const double *w) {
  typedef double cv8 __attribute__((vector_size(64)));
  cv8 a[9];
  double a88 = 0;
  for (int i = 0; i < 9; i++) a[i] = (cv8){0, 0, 0, 0, 0, 0, 0, 0};
  const double s1 = A1[0], s2 = A2[0];
  for (int i = 0; i < len; i++) {
    const double *s = u + 6 * inl[i];
    const double x0 = s[0] * s1 + A1[1], x1 = s[1] * s1 + A1[2];
    const double y0 = s[3] * s2 + A2[1], y1 = s[4] * s2 + A2[2];
    cv8 v = {x0 * y0, x1 * y0, y0, x0 * y1, x1 * y1, y1, x0, x1};
    double z8 = 1;
    if (w) {
      const double m = w[inl[i]];
      v *= (cv8){m, m, m, m, m, m, m, m};
      z8 *= m;
    }
#pragma unroll
    for (int r = 0; r < 8; r++) { const double zr = v[r]; a[r] += (cv8){zr, zr, zr, zr, zr, zr, zr, zr} * v; }
    a[8] += (cv8){z8, z8, z8, z8, z8, z8, z8, z8} * v;
    a88 += z8 * z8;
  }
  for (int i = 0; i < 9; i++)
    for (int j = 0; j <= i; j++) { const double x = j < 8 ? a[i][j] : a88; Cv[9 * i + j] = x; Cv[i + 9 * j] = x; }
}
#endif
static void cov_fmN_fused(double *Cv, const double *u, const int *inl, int len, const double *A1, const double *A2, const double *w) {
#if defined(__x86_64__)
  static const bool wide = __builtin_cpu_supports("avx512f");
  if (wide) return cov_fmN_fused_w8(Cv, u, inl, len, A1, A2, w);
#endif
  typedef double cv4 __attribute__((vector_size(32)));
  cv4 a0[9], a1[5];
  double a88 = 0;
  for (int i = 0; i < 9; i++) a0[i] = (cv4){0, 0, 0, 0};
  for (int i = 0; i < 5; i++) a1[i] = (cv4){0, 0, 0, 0};
  const double s1 = A1[0], s2 = A2[0];
  for (int i = 0; i < len; i++) {
    const double *s = u + 6 * inl[i];
    const double x0 = s[0] * s1 + A1[1], x1 = s[1] * s1 + A1[2];
    const double y0 = s[3] * s2 + A2[1], y1 = s[4] * s2 + A2[2];
    cv4 v0 = {x0 * y0, x1 * y0, y0, x0 * y1}, v1 = {x1 * y1, y1, x0, x1};
    double z8 = 1;
    if (w) {
      const double m = w[inl[i]];
      const cv4 mm = {m, m, m, m};
      v0 *= mm; v1 *= mm; z8 *= m;
    }
#pragma unroll
    for (int r = 0; r < 4; r++) { const cv4 zi = {v0[r], v0[r], v0[r], v0[r]}; a0[r] += zi * v0; }
#pragma unroll
    for (int r = 0; r < 4; r++) { const cv4 zi = {v1[r], v1[r], v1[r], v1[r]}; a0[4 + r] += zi * v0; a1[r] += zi * v1; }
    const cv4 z8v = {z8, z8, z8, z8};
    a0[8] += z8v * v0; a1[4] += z8v * v1;
    a88 += z8 * z8;
  }
  double acc[9][9];
  for (int i = 0; i < 9; i++) {
    for (int j = 0; j < 4; j++) acc[i][j] = a0[i][j];
    if (i >= 4) for (int j = 0; j < 4; j++) acc[i][4 + j] = a1[i - 4][j];
  }
  acc[8][8] = a88;
  for (int i = 0; i < 9; i++)
    for (int j = 0; j <= i; j++) { Cv[9 * i + j] = acc[i][j]; Cv[i + 9 * j] = acc[i][j]; }
}
// slcm, Ftools.c:37-85: the cubic det(x A + (1 - x) B) = p0 x^3 + p1 x^2 + p2 x + p3; B leaves as A - B.
// The roots feed the orientation test (all_ori_valid), a sign decision that is ill-conditioned whenever an epipolar line of the
// sample is nearly horizontal: a relative 3e-13 in a root flipped it on a 31-tentative problem and sent the trajectory elsewhere.
// The coefficients are therefore summed in the reference's term order -- the expansion it spells out, kept here as data: a term is
// (+-c) x_i x_j [x_k] over x = (A row-wise, B row-wise), evaluated ((c x_i) x_j) x_k and added left to right; p1 and p2 end in
// bracketed six-term sums times one more entry.  (An equivalent determinant form agreed to 1e-16 in the coefficients only.)
struct CubicTerm { signed char c, i, j, k; };
static inline double cubic_sum(const double *x, const CubicTerm *t, int n) {
  double s = 0;
  for (int q = 0; q < n; q++) {
    const int c = t[q].c < 0 ? -t[q].c : t[q].c;
    double v = c == 1 ? x[t[q].i] : (double)c * x[t[q].i];
    v = v * x[t[q].j];
    if (t[q].k >= 0) v = v * x[t[q].k];
    if (q == 0) s = t[q].c < 0 ? -v : v;
    else s = t[q].c < 0 ? s - v : s + v;
  }
  return s;
}
static void slcm(const double *A, double *B, double *p) {
  static const CubicTerm DETB[6] = {{-1, 11, 13, 15}, {1, 10, 14, 15}, {1, 11, 12, 16}, {-1, 9, 14, 16}, {-1, 10, 12, 17}, {1, 9, 13, 17}};
  static const CubicTerm P1[18] = {{-1, 8, 10, 12}, {1, 7, 11, 12}, {1, 8, 9, 13}, {-1, 6, 11, 13}, {-1, 7, 9, 14}, {1, 6, 10, 14},
                                   {1, 5, 10, 15}, {-1, 4, 11, 15}, {-1, 2, 13, 15}, {3, 11, 13, 15}, {1, 1, 14, 15}, {-3, 10, 14, 15},
                                   {-1, 5, 9, 16}, {1, 3, 11, 16}, {1, 2, 12, 16}, {-3, 11, 12, 16}, {-1, 0, 14, 16}, {3, 9, 14, 16}};
  static const CubicTerm P1G[6] = {{1, 4, 9, -1}, {-1, 3, 10, -1}, {-1, 1, 12, -1}, {3, 10, 12, -1}, {1, 0, 13, -1}, {-3, 9, 13, -1}};   // ... x B33
  static const CubicTerm P2[24] = {{-1, 3, 8, 10}, {1, 3, 7, 11}, {1, 2, 7, 12}, {-1, 1, 8, 12}, {2, 8, 10, 12}, {-2, 7, 11, 12},
                                   {-1, 2, 6, 13}, {1, 0, 8, 13}, {-2, 8, 9, 13}, {2, 6, 11, 13}, {1, 1, 6, 14}, {-1, 0, 7, 14},
                                   {2, 7, 9, 14}, {-2, 6, 10, 14}, {2, 2, 13, 15}, {-3, 11, 13, 15}, {-2, 1, 14, 15}, {3, 10, 14, 15},
                                   {1, 2, 3, 16}, {-2, 3, 11, 16}, {-2, 2, 12, 16}, {3, 11, 12, 16}, {2, 0, 14, 16}, {-3, 9, 14, 16}};
  static const CubicTerm P2G1[6] = {{-1, 7, 9, -1}, {1, 6, 10, -1}, {1, 1, 15, -1}, {-2, 10, 15, -1}, {-1, 0, 16, -1}, {2, 9, 16, -1}};  // A23 x ...
  static const CubicTerm P2G2[6] = {{-1, 1, 3, -1}, {2, 3, 10, -1}, {2, 1, 12, -1}, {-3, 10, 12, -1}, {-2, 0, 13, -1}, {3, 9, 13, -1}};   // ... x B33
  static const CubicTerm P2G3[6] = {{1, 8, 9, -1}, {-1, 6, 11, -1}, {-1, 2, 15, -1}, {2, 11, 15, -1}, {1, 0, 17, -1}, {-2, 9, 17, -1}};   // A22 x ...
  double x[18];
  for (int i = 0; i < 9; i++) { x[i] = A[i]; x[9 + i] = B[i]; }
  p[0] = cubic_sum(x, DETB, 6);
  p[1] = cubic_sum(x, P1, 18) + cubic_sum(x, P1G, 6) * x[17];
  p[2] = cubic_sum(x, P2, 24) + x[5] * cubic_sum(x, P2G1, 6) + cubic_sum(x, P2G2, 6) * x[17] + x[4] * cubic_sum(x, P2G3, 6);
  for (int i = 0; i < 9; i++) { B[i] = A[i] - B[i]; x[9 + i] = B[i]; }
  p[3] = cubic_sum(x, DETB, 6);
}
// rroots3, Ftools.c:216-261: real roots of po[0] x^3 + po[1] x^2 + po[2] x + po[3]
static int rroots3(const double *po, double *r) {
  const double b = po[1] / po[0], c = po[2] / po[0];
  const double b2 = b * b, bt = b / 3;
  const double p = (3 * c - b2) / 9;
  const double q = ((2 * b2 * b) / 27 - b * c / 3 + po[3] / po[0]) / 2;
  const double D = q * q + p * p * p;
  if (D > 0) {
    const double A = sqrt(D) - q;
    if (A > 0) {
      const double v = pow(A, 1.0 / 3);
      r[0] = v - p / v - bt;
    } else {
      const double v = pow(-A, 1.0 / 3);
      r[0] = p / v - v - bt;
    }
    return 1;
  }
  const double e = q > 0 ? 1 : -1;
  const double R = e * sqrt(-p), R2 = R * 2;
  double cosphi = q / (R * R * R);
  if (cosphi > 1) cosphi = 1;
  else if (cosphi < -1) cosphi = -1;
  const double phit = acos(cosphi) / 3;
  const double pit = 3.14159265358979 / 3;
  r[0] = -R2 * cos(phit) - bt;
  r[1] = R2 * cos(pit - phit) - bt;
  r[2] = R2 * cos(pit + phit) - bt;
  return 3;
}
// FDs (Sampson), Ftools.c:87-107; FDsSym (symmetric epipolar distance), :109-131; exFDs, :162-185; exFDsSym, :186-210.
// Four points per step on 256-bit vectors, eight where the CPU has 512-bit ones (hds_ld4 / hds_ld8 of ransac_common.hpp gather a
// coordinate of consecutive points): every lane runs the scalar expression of its own point -- the same operations in the same
// order, IEEE mul / add / div / sqrt, no contraction -- so the values are those of the scalar loops that finish the last points.
#define FDS_COMMON(T, X1, Y1, X2, Y2) \
  const T rxc = F[0] * X2 + F[3] * Y2 + F[6]; \
  const T ryc = F[1] * X2 + F[4] * Y2 + F[7]; \
  const T rwc = F[2] * X2 + F[5] * Y2 + F[8]; \
  const T r = (X1 * rxc + Y1 * ryc + rwc); \
  const T rx = F[0] * X1 + F[1] * Y1 + F[2]; \
  const T ry = F[3] * X1 + F[4] * Y1 + F[5];
// KIND 0 FDs, 1 FDsSym, 2 exFDs, 3 exFDsSym
#define FDS_VECTOR_LOOP(V, W, LD, KIND) \
  for (; i + W <= len; i += W) { \
    const double *uu = u + (size_t)6 * i; \
    const V x1 = LD(uu, 6), y1 = LD(uu + 1, 6), x2 = LD(uu + 3, 6), y2 = LD(uu + 4, 6); \
    FDS_COMMON(V, x1, y1, x2, y2) \
    if (KIND == 0) { \
      const V o = r * r / (rxc * rxc + ryc * ryc + rx * rx + ry * ry); \
      for (int k = 0; k < W; k++) p[i + k] = o[k]; \
    } else if (KIND == 1) { \
      const V a = rxc * rxc + ryc * ryc, b = rx * rx + ry * ry; \
      const V o = r * r * (a + b) / (a * b); \
      for (int k = 0; k < W; k++) p[i + k] = o[k]; \
    } else if (KIND == 2) { \
      const V ww = rxc * rxc + ryc * ryc + rx * rx + ry * ry; \
      const V o = r * r / ww; \
      for (int k = 0; k < W; k++) { p[i + k] = o[k]; w[i + k] = 1 / sqrt(ww[k]); } \
    } else { \
      const V a = rxc * rxc + ryc * ryc, b = rx * rx + ry * ry; \
      const V ww = (a * b) / (a + b); \
      const V o = r * r / ww; \
      for (int k = 0; k < W; k++) { w[i + k] = ww[k]; p[i + k] = o[k]; } \
    } \
  }
#if defined(__x86_64__)
template <int KIND>
__attribute__((target("avx512f"))) static int fds_loop8(const double *u, const double *F, double *p, double *w, int len) {
  int i = 0;
  FDS_VECTOR_LOOP(hds_v8, 8, hds_ld8, KIND)
  return i;
}
#endif
template <int KIND>
static inline int fds_vector(const double *u, const double *F, double *p, double *w, int len) {
  int i = 0;
#if defined(__x86_64__)
  static const bool wide = __builtin_cpu_supports("avx512f");
  if (wide) i = fds_loop8<KIND>(u, F, p, w, len);
#endif
  FDS_VECTOR_LOOP(hds_v4, 4, hds_ld4, KIND)
  return i;
}
static void FDs(const double *u, const double *F, double *p, int len) {
  int i = fds_vector<0>(u, F, p, nullptr, len);
  for (u += (size_t)6 * i; i < len; i++, u += 6) {
    FDS_COMMON(double, u[0], u[1], u[3], u[4])
    p[i] = r * r / (rxc * rxc + ryc * ryc + rx * rx + ry * ry);
  }
}
static void FDsSym(const double *u, const double *F, double *p, int len) {
  int i = fds_vector<1>(u, F, p, nullptr, len);
  for (u += (size_t)6 * i; i < len; i++, u += 6) {
    FDS_COMMON(double, u[0], u[1], u[3], u[4])
    const double a = rxc * rxc + ryc * ryc;
    const double b = rx * rx + ry * ry;
    p[i] = r * r * (a + b) / (a * b);
  }
}
static void exFDs(const double *u, const double *F, double *p, double *w, int len) {
  int i = fds_vector<2>(u, F, p, w, len);
  for (u += (size_t)6 * i; i < len; i++, u += 6) {
    FDS_COMMON(double, u[0], u[1], u[3], u[4])
    w[i] = rxc * rxc + ryc * ryc + rx * rx + ry * ry;
    p[i] = r * r / w[i];
    w[i] = 1 / sqrt(w[i]);
  }
}
static void exFDsSym(const double *u, const double *F, double *p, double *w, int len) {
  int i = fds_vector<3>(u, F, p, w, len);
  for (u += (size_t)6 * i; i < len; i++, u += 6) {
    FDS_COMMON(double, u[0], u[1], u[3], u[4])
    const double a = rxc * rxc + ryc * ryc;
    const double b = rx * rx + ry * ry;
    w[i] = (a * b) / (a + b);
    p[i] = r * r / w[i];
  }
}
#undef FDS_COMMON
#undef FDS_VECTOR_LOOP
// singulF, Ftools.c:292-312: closest rank-2 matrix = F (I - v v^T), v the right singular vector of the
// smallest singular value (dgesvd_ + zeroed third singular value in the reference)
static void singulF(double *F) {
  double G[9], ev[3], V[9];
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) {
      double s = 0;
      for (int k = 0; k < 3; k++) s += F[k * 3 + i] * F[k * 3 + j];
      G[i * 3 + j] = s;
    }
  jacobi_eig(G, 3, ev, V);
  int m = 0;
  for (int i = 1; i < 3; i++) if (ev[i] < ev[m]) m = i;
  const double v[3] = {V[m], V[3 + m], V[6 + m]};
  for (int i = 0; i < 3; i++) {
    const double fv = F[i * 3] * v[0] + F[i * 3 + 1] * v[1] + F[i * 3 + 2] * v[2];
    for (int j = 0; j < 3; j++) F[i * 3 + j] -= fv * v[j];
  }
}
// denormF, utools.c:54-70
static void denormF(double *F, const double *A1, const double *A2) {
  double r = A2[0], x = A2[1], y = A2[2];
  F[6] += x * F[0] + y * F[3];
  F[7] += x * F[1] + y * F[4];
  F[8] += x * F[2] + y * F[5];
  F[0] *= r; F[1] *= r; F[2] *= r;
  F[3] *= r; F[4] *= r; F[5] *= r;
  r = A1[0]; x = A1[1]; y = A1[2];
  F[2] += x * F[0] + y * F[1];
  F[5] += x * F[3] + y * F[4];
  F[8] += x * F[6] + y * F[7];
  F[0] *= r; F[3] *= r; F[6] *= r;
  F[1] *= r; F[4] *= r; F[7] *= r;
}
// unit vector orthogonal to the `len` (<= 8) columns of the column-wise 9 x len matrix Z: the ninth left
// singular vector that svduv(D, Z, V, 9, U, 8) leaves in V[.][8] (Ftools.c:331-333)
static void left_null9(const double *Z, int len, double *f) {
  double G[81], ev[9], V[81];
  for (int i = 0; i < 9; i++)
    for (int j = 0; j <= i; j++) {
      double s = 0;
      for (int k = 0; k < len; k++) s += Z[(size_t)i * len + k] * Z[(size_t)j * len + k];
      G[i * 9 + j] = s; G[j * 9 + i] = s;
    }
  jacobi_eig(G, 9, ev, V);
  int m = 0;
  for (int i = 1; i < 9; i++) if (ev[i] < ev[m]) m = i;
  for (int k = 0; k < 9; k++) f[k] = V[k * 9 + m];
}
// u2f / u2fw, Ftools.c:315-345 / 347-398 (w == nullptr: unweighted)
static void u2fw(const double *u, const int *inl, const double *w, int len, double *F, double *buffer) {
  double A1[3], A2[3];
  double *Z = buffer;
  if (len > 8) {
    double V[81], ev[9], E[81];
    // lin_fmN(u, Z, inl, len, A1, A2); Z[i][.] *= w[inl[i]] when weighted; cov_mat(V, Z, len, 9) -- in one pass:
    { HPROF(8); normu(u, inl, len, A1, A2); }
    { HPROF(9); cov_fmN_fused(V, u, inl, len, A1, A2, w); }
    { HPROF(10); jacobi_eig(V, 9, ev, E); }
    int j = 0;
    for (int i = 1; i < 9; i++) if (ev[i] < ev[j]) j = i;
    for (int i = 0; i < 9; i++) F[i] = E[i * 9 + j];
  } else {
    lin_fm(u, Z, inl, len);
    if (w)
      // scalmul(Z + i, w[j], 9, 9), Ftools.c:372-376: the stride is 9 although the 9 x len matrix has row
      // length len (<= 8), so the weight of point i lands on the entries i, i + 9, i + 18, ... -- kept as is
      for (int i = 0; i < len; i++) {
        const double m = w[inl[i]];
        for (int c = 0; c < 9; c++)
          if (i + 9 * c < 9 * len) Z[i + 9 * c] *= m;
      }
    { HPROF(12); left_null9(Z, len, F); }
  }
  HPROF(11);
  singulF(F);
  if (len > 8) denormF(F, A1, A2);
}
static inline void u2f(const double *u, const int *inl, int len, double *F, double *buffer) {
  u2fw(u, inl, nullptr, len, F, buffer);
}
// epipole / getorisig / all_ori_valid, Ftools.c:400-436
static void epipole(double *ec, const double *F) {
  const double xeps = 1.9984e-15;
  cross3(ec, F, F + 6);
  for (int i = 0; i < 3; i++)
    if ((ec[i] > xeps) || (ec[i] < -xeps)) return;
  cross3(ec, F + 3, F + 6);
}
static inline double getorisig(const double *F, const double *ec, const double *u) {
  const double s1 = F[0] * u[3] + F[3] * u[4] + F[6] * u[5];
  const double s2 = ec[1] * u[2] - ec[2] * u[1];
  return s1 * s2;
}
static int all_ori_valid(const double *F, const double *us, const int *idx, int N) {
  double ec[3];
  epipole(ec, F);
  const double sig1 = getorisig(F, ec, us + 6 * idx[0]);
  for (int i = 1; i < N; i++) {
    const double sig = getorisig(F, ec, us + 6 * idx[i]);
    if (sig1 * sig < 0) return 0;
  }
  return 1;
}

// ---- homography helpers used by the degeneracy test -----------------------------------------------------
// row-major version of lin_hg (see ransac_common.hpp HDs)
static void lin_hg_rows(const double *u, double *lin, int len) {
  for (int i = 0; i < len; i++) {
    const double *s = u + 6 * i;
    double *r0 = lin + (size_t)18 * i, *r1 = r0 + 9;
    for (int j = 0; j < 3; j++) {
      r0[3 * j] = s[3 + j]; r0[3 * j + 1] = 0; r0[3 * j + 2] = -s[0] * s[3 + j];
      r1[3 * j] = 0; r1[3 * j + 1] = s[3 + j]; r1[3 * j + 2] = -s[1] * s[3 + j];
    }
  }
}
// dHDs, DegUtils.c:186-209
static void dHDs(const double *H, const double *u, int len, double *Ds, std::vector<double> &lin) {
  lin.resize((size_t)len * 18);
  lin_hg_rows(u, lin.data(), len);
  HDs(lin.data(), u, H, Ds, len);
}

static inline Score tr_inlidxs(const double *err, int len, double th, int *inl) {   // inlidxs as called from exp_ranF.c
  const Score s = inlidxs(err, len, th, inl);
  RTRACE("S %u %.17g th %.3g\n", s.I, s.J, th);
  return s;
}
// ---- rFtH's hypothesis loop on the device ---------------------------------------------------------------------------------
// rFtH (DegUtils.c:254-440) tries up to 2 x 10^4 epipoles, each from two off-plane correspondences, and counts for each the
// off-plane correspondences within 2 th of the F it gives with the plane's H; only a count above the best so far changes any
// state (innerFH, which also draws from the PRNG).  On a planar scene no hypothesis ever does and the loop is 15-20 ms of one
// host core per DEGENSAC sample.  Between two such events the sample stream is a function of the PRNG state alone, so the loop
// runs in batches: the host draws the next B samples from a COPY of the generator, one device thread per hypothesis forms the
// epipole and F and counts (the f64 operations of the host code in the same order: the counts are the host's), the host takes
// the first hypothesis whose count beats the best, replays the generator up to it and runs the reference's body for it.
struct RfthArgs { double Ht[9]; double th2; int nN, B; };
constexpr int RFTH_TILE = 512;     // off-plane correspondences staged in LDS at a time (24 KB)
// All four arrays are pinned host memory mapped into the device's address space: a batch is ONE operation on the stream (no
// copy in front of the kernel, none behind it) -- under a device saturated by 16 contexts every queued operation waits its turn,
// and three of them made a batch slower than the host loop it replaces.  The points are read once per workgroup (into LDS).
__global__ __launch_bounds__(256) void k_rfth_count(const double *__restrict__ us, const double *__restrict__ uN,
                                                    const unsigned *__restrict__ pairs, RfthArgs A, unsigned *__restrict__ cnt) {
  __shared__ double tile[RFTH_TILE * 6];
  const int j = blockIdx.x * 256 + threadIdx.x;
  const bool live = j < A.B;
  double F[9];
  if (live) {
    const double *a = us + 6 * (size_t)pairs[2 * j], *b = us + 6 * (size_t)pairs[2 * j + 1];
    double c1[3], c2[3], ec[3], aFt[9], aFtH[9];
    cross3(c1, a, a + 3);
    cross3(c2, b, b + 3);
    cross3(ec, c1, c2);
    const double nrm = sqrt(ec[0] * ec[0] + ec[1] * ec[1] + ec[2] * ec[2]);
    ec[0] = ec[0] / nrm; ec[1] = ec[1] / nrm; ec[2] = ec[2] / nrm;
    skew_sym(ec, aFt);
    mul3(aFtH, aFt, A.Ht);
    tr3(F, aFtH);
  }
  unsigned no_i = 0;
  for (int t0 = 0; t0 < A.nN; t0 += RFTH_TILE) {
    const int m = min(RFTH_TILE, A.nN - t0);
    __syncthreads();
    for (int k = threadIdx.x; k < 6 * m; k += 256) tile[k] = uN[6 * (size_t)t0 + k];
    __syncthreads();
    if (live)
      for (int i = 0; i < m; i++) {
        const double *u = tile + 6 * i;     // the same LDS address in every lane: a broadcast
        const double rxc = F[0] * u[3] + F[3] * u[4] + F[6];
        const double ryc = F[1] * u[3] + F[4] * u[4] + F[7];
        const double rwc = F[2] * u[3] + F[5] * u[4] + F[8];
        const double r = (u[0] * rxc + u[1] * ryc + rwc);
        const double rx = F[0] * u[0] + F[1] * u[1] + F[2];
        const double ry = F[3] * u[0] + F[4] * u[1] + F[5];
        const double d = r * r / (rxc * rxc + ryc * ryc + rx * rx + ry * ry);
        if (d < A.th2) ++no_i;
      }
  }
  if (live) cnt[j] = no_i;
}

// per host thread (the verifier runs on the contexts' helper threads): a stream and the few buffers of the loop, on the
// thread's current device.  No device (the CPU test container), MODSX_VERIFY_DEVICE=0 or any HIP error: the host loop.
static std::atomic<long> g_rfthStats[10];   // [6..9]: microseconds drawing samples ahead, waiting for the device, in the host phase, in event bodies
static inline long rfth_us(std::chrono::steady_clock::time_point a) { return std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - a).count(); }
//   // batches, hypotheses scored on the device, events, host / device disagreements, rFtH loops, their microseconds
struct RfthDevice {
  enum { BATCH = 20480 };   // a whole loop (2 x 10^4 hypotheses) in one round trip: speculation past an event costs the device nothing
  bool tried = false, ok = false;
  int dev = -1;                 // the device that was current when it was made: leased again only to threads on that device
  hipStream_t s = nullptr;
  double *hUs = nullptr, *hUn = nullptr, *dUs = nullptr, *dUn = nullptr;     // pinned + mapped: host pointer / device alias
  unsigned *hPairs = nullptr, *hCnt = nullptr, *dPairs = nullptr, *dCnt = nullptr;
  size_t capPts = 0;
  static bool pinned(void **h, void **d, size_t bytes) {
    return hipHostMalloc(h, bytes, hipHostMallocMapped) == hipSuccess && hipHostGetDevicePointer(d, *h, 0) == hipSuccess;
  }
  // failed initialisations / host-device disagreements in this process: after RFTH_GIVE_UP of them the loops stay on the host (a box
  // where the set-up keeps failing would otherwise pay stream + pinned allocations on every rFtH loop)
  static std::atomic<int> &failures() { static std::atomic<int> f(0); return f; }
  enum { RFTH_GIVE_UP = 8 };
  bool init() {
    if (tried) return ok;
    tried = true;
    const char *e = getenv("MODSX_VERIFY_DEVICE");
    if (e && atoi(e) == 0) return false;
    if (failures().load() >= RFTH_GIVE_UP) return false;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n < 1) { (void)hipGetLastError(); return false; }
    // the highest priority: a batch is a few wavefronts that must not queue behind the launch sets of 16 contexts
    int lo = 0, hi = 0;
    (void)hipDeviceGetStreamPriorityRange(&lo, &hi);
    if (hipStreamCreateWithPriority(&s, hipStreamNonBlocking, hi) != hipSuccess) { (void)hipGetLastError(); s = nullptr; return false; }
    if (!pinned((void **)&hPairs, (void **)&dPairs, BATCH * 8) || !pinned((void **)&hCnt, (void **)&dCnt, BATCH * 4)) {
      (void)hipGetLastError();
      release();       // a state that failed part-way keeps nothing
      return false;
    }
    ok = true;
    return true;
  }
  void release() {
    if (hPairs) hipHostFree(hPairs);
    if (hCnt) hipHostFree(hCnt);
    if (hUs) hipHostFree(hUs);
    if (hUn) hipHostFree(hUn);
    if (s) hipStreamDestroy(s);
    hPairs = hCnt = nullptr; hUs = hUn = nullptr; capPts = 0; s = nullptr;
    (void)hipGetLastError();
  }
  bool points(const double *us, const double *uN, size_t nN) {
    if (nN > capPts) {
      if (hUs) hipHostFree(hUs);
      if (hUn) hipHostFree(hUn);
      hUs = hUn = nullptr; capPts = 0;
      const size_t cap = nN + nN / 2 + 64;
      if (!pinned((void **)&hUs, (void **)&dUs, cap * 48) || !pinned((void **)&hUn, (void **)&dUn, cap * 48)) return false;
      capPts = cap;
    }
    memcpy(hUs, us, nN * 48);
    memcpy(hUn, uN, nN * 48);
    return true;
  }
  bool count(const RfthArgs &A) {     // hPairs -> hCnt
    hipLaunchKernelGGL(k_rfth_count, dim3((A.B + 255) / 256), dim3(256), 0, s, dUs, dUn, dPairs, A, dCnt);
    return hipStreamSynchronize(s) == hipSuccess && hipGetLastError() == hipSuccess;
  }
};
// The verifier's threads come and go (modsx_match_pairs starts its helpers per call), so the device state is not theirs: a
// thread borrows one from a process-wide pool for the duration of a loop and hands it back -- as many as ever verified at the
// same time exist, they are reused for the life of the process and never torn down (no HIP call at thread or process exit).
struct RfthLease {
  RfthDevice *d = nullptr;
  static std::mutex &mu() { static std::mutex m; return m; }
  static std::vector<RfthDevice *> &idle() { static std::vector<RfthDevice *> v; return v; }
  RfthLease() {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) { (void)hipGetLastError(); dev = -1; }
    {
      std::lock_guard<std::mutex> lk(mu());
      std::vector<RfthDevice *> &v = idle();
      for (size_t i = 0; i < v.size(); i++)
        if (v[i]->dev == dev) { d = v[i]; v.erase(v.begin() + i); break; }
    }
    if (!d) { d = new RfthDevice; d->dev = dev; }
  }
  ~RfthLease() {
    if (d->tried && !d->ok) {      // failed or disagreed: its resources go, and the next loop on this device starts a fresh state
      const char *e = getenv("MODSX_VERIFY_DEVICE");
      if (!(e && atoi(e) == 0) && d->dev >= 0) { RfthDevice::failures()++; d->release(); delete d; return; }
    }
    std::lock_guard<std::mutex> lk(mu());
    idle().push_back(d);
  }
};

struct RansacF {
  const double *u;
  int len;
  GlibcRandom rng;
  HashTable ht;
  std::vector<double> buffer, lin;
  FdsFn fds;
  ExFdsFn exfds;
  // Scratch of the nested local-optimisation routines, one set per routine (innerFH calls u2Fit, innerH calls inHrani calls iterH),
  // sized at first use and kept for the call: they used to be vectors constructed per invocation -- ~300 times per verification,
  // half a megabyte each at 6 k tentatives, i.e. mmap + page faults + munmap under the process-wide mm lock that 16 verifying
  // threads share.  Every buffer is written before it is read, except innerH's error planes, which are cleared as the vectors were.
  template <class T> static T *grab(std::vector<T> &v, size_t n) { if (v.size() < n) v.resize(n); return v.data(); }
  std::vector<double> sFitDs, sFitBuf, sFhUsam, sFhDs, sFhBuf, sIhErr, sIhZ;
  std::vector<int> sFitInl, sFhAll, sIhInliers, sHraniInt;
  std::vector<unsigned char> sFhV;
  std::vector<unsigned> sDualA, sDualB;

  int *randsubset(int *pool, int max_sz, int siz) {  // rtools.c:25-39
    for (int i = 0; i < siz; i++) {
      const int s = (int)(rng.next() % (max_sz - i));
      const int j = max_sz - i - 1;
      const int q = pool[s]; pool[s] = pool[j]; pool[j] = q;
    }
    return pool + max_sz - siz;
  }

  // iterH, ranH.c:18-84 (the non-hashing variant innerH uses; inlLimit = 10 there)
  Score iterH(const double *Z, int *inliers, double th, double ths, double *H, double **errs, unsigned inlLimit) {
    double *d = errs[1];
    double h[9];
    Score S = {0, 0}, Ss, maxS;
    const double dth = (ths - th) / 4;
    maxS = inlidxs(errs[4], len, th, inliers);
    if (maxS.I < 4) return S;
    memcpy(h, H, sizeof h);
    if (maxS.I <= inlLimit) u2h(u, inliers, (int)maxS.I, h, buffer.data());
    else u2h(u, randsubset(inliers, (int)maxS.I, (int)inlLimit), (int)inlLimit, h, buffer.data());
    for (int it = 0; it < 4; ++it) {
      HDs(Z, u, h, d, len);
      S = inlidxs(d, len, th, inliers);
      Ss = inlidxs(d, len, ths, inliers);
      if (score_less(maxS, S)) {
        maxS = S;
        errs[1] = errs[0]; errs[0] = d; d = errs[1];
        memcpy(H, h, 9 * sizeof(double));
      }
      if (Ss.I < 4) return maxS;
      if (Ss.I <= inlLimit) u2h(u, inliers, (int)Ss.I, h, buffer.data());
      else u2h(u, randsubset(inliers, (int)Ss.I, (int)inlLimit), (int)inlLimit, h, buffer.data());
      ths -= dth;
    }
    HDs(Z, u, h, d, len);
    S = inlidxs(d, len, th, inliers);
    if (score_less(maxS, S)) {
      maxS = S;
      errs[1] = errs[0]; errs[0] = d;
      memcpy(H, h, 9 * sizeof(double));
    }
    return maxS;
  }
  // inHrani, ranH.c:88-135
  Score inHrani(const double *Z, int *inliers, int ninl, double th, double **errs, double *H, unsigned inlLimit) {
    Score S, maxS = {0, 0};
    double h[9];
    if (ninl < 8) return maxS;
    int *intbuff = grab(sHraniInt, len);
    int ssiz = ninl / 2;
    if (ssiz > 12) ssiz = 12;
    double *d = errs[2]; errs[2] = errs[0]; errs[0] = d;
    memcpy(h, H, sizeof h);
    for (int i = 0; i < 10; ++i) {
      int *sample = randsubset(inliers, ninl, ssiz);
      u2h(u, sample, ssiz, h, buffer.data());
      HDs(Z, u, h, errs[0], len);
      errs[4] = errs[0];
      S = iterH(Z, intbuff, th, 4 * th, h, errs, inlLimit);
      if (score_less(maxS, S)) {
        maxS = S;
        d = errs[2]; errs[2] = errs[0]; errs[0] = d;
        memcpy(H, h, 9 * sizeof(double));
      }
    }
    d = errs[2]; errs[2] = errs[0]; errs[0] = d;
    return maxS;
  }
  // innerH, DegUtils.c:689-729
  unsigned innerH(double *H, double th, unsigned iters, unsigned char *inl) {
    double *err = grab(sIhErr, (size_t)len * 4), *Z = grab(sIhZ, (size_t)len * 18);
    std::fill(err, err + (size_t)len * 4, 0.0);
    int *inliers = grab(sIhInliers, len);
    double *errs[5];
    for (int i = 0; i < 4; i++) errs[i] = err + (size_t)i * len;
    errs[4] = errs[3];
    lin_hg_rows(u, Z, len);
    double *d = errs[0];
    HDs(Z, u, H, d, len);
    Score S = inlidxs(d, len, th, inliers);
    S = inHrani(Z, inliers, (int)S.I, th, errs, H, iters);
    d = errs[0];
    unsigned I = 0;
    for (int j = 0; j < len; j++) {
      if (d[j] <= th) { ++I; inl[j] = 1; }
      else inl[j] = 0;
    }
    return I;
  }

  // Hdetect, DegUtils.c:83-162: H from F and three correspondences (Hartley & Zisserman, result 13.6)
  void Hdetect(const double *F, const double *u7, const unsigned char *idx3, double *H) {
    double dsv[3], V[9], ec[3], Ex[9], Ft[9], A[9];
    // ec = third column of V of svduv(F) -- the epipole only when the smallest singular value comes out last
    svd3_unsorted(F, dsv, V);
    ec[0] = V[2]; ec[1] = V[5]; ec[2] = V[8];
    skew_sym(ec, Ex);
    tr3(Ft, F);
    mul3(A, Ex, Ft);
    double P1[3][3], P2[3][3];  // [point][coordinate]
    for (int i = 0; i < 3; i++)
      for (int j = 0; j < 3; j++) { P1[i][j] = u7[idx3[i] * 6 + j]; P2[i][j] = u7[idx3[i] * 6 + j + 3]; }
    double b[3];
    for (int i = 0; i < 3; i++) {
      double Ap2[3], p1c[3], p2c[3];
      for (int r = 0; r < 3; r++) Ap2[r] = A[r * 3] * P2[i][0] + A[r * 3 + 1] * P2[i][1] + A[r * 3 + 2] * P2[i][2];
      cross3(p1c, P1[i], Ap2);
      for (int r = 0; r < 3; r++)
        p2c[r] = -Ex[r * 3] * P1[i][0] + -Ex[r * 3 + 1] * P1[i][1] + -Ex[r * 3 + 2] * P1[i][2];
      b[i] = (p1c[0] * p2c[0] + p1c[1] * p2c[1] + p1c[2] * p2c[2]) / (p2c[0] * p2c[0] + p2c[1] * p2c[1] + p2c[2] * p2c[2]);
    }
    // M = rows p2_i; x = M^-1 b; H = A - ec x^T, stored column-wise
    double M[9], Mi[9];
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) M[i * 3 + j] = P2[i][j];
    const bool sing = !inv3_pivot(M, Mi);
    double x[3] = {0, 0, 0};
    if (!sing)
      for (int i = 0; i < 3; i++) x[i] = Mi[i * 3] * b[0] + Mi[i * 3 + 1] * b[1] + Mi[i * 3 + 2] * b[2];
    for (int i = 0; i < 3; i++)
      for (int j = 0; j < 3; j++) H[i + j * 3] = A[i * 3 + j] - ec[i] * x[j];
    if (std::isnan(H[0]) || std::isinf(H[0]) || sing) {
      H[1] = H[2] = H[3] = H[5] = H[6] = H[7] = 0;
      H[0] = H[4] = H[8] = 1;
    }
  }
  // minv (matutls/minv.c) reports a singular matrix when a pivot falls below 1e-15 of the largest pivot so far
  static bool inv3_pivot(const double *Min, double *Out) {
    double a[3][6];
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) { a[i][j] = Min[i * 3 + j]; a[i][3 + j] = (i == j); }
    double tq = 0;
    for (int c = 0; c < 3; c++) {
      int piv = c;
      for (int r = c + 1; r < 3; r++) if (fabs(a[r][c]) > fabs(a[piv][c])) piv = r;
      const double s = fabs(a[piv][c]);
      tq = tq > s ? tq : s;
      if (s < 1e-15 * tq || s == 0) return false;
      if (piv != c) for (int k = 0; k < 6; k++) { double t = a[c][k]; a[c][k] = a[piv][k]; a[piv][k] = t; }
      const double d = a[c][c];
      for (int k = 0; k < 6; k++) a[c][k] /= d;
      for (int r = 0; r < 3; r++) {
        if (r == c) continue;
        const double f = a[r][c];
        for (int k = 0; k < 6; k++) a[r][k] -= f * a[c][k];
      }
    }
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) Out[i * 3 + j] = a[i][3 + j];
    return true;
  }
  // checksample, DegUtils.c:37-80
  int checksample(const double *F, const double *u7, double th, double *H) {
    static const unsigned char IDXS[5][3] = {{0, 1, 2}, {3, 4, 5}, {0, 1, 6}, {3, 4, 6}, {2, 5, 6}};
    double Ds[7], sDs[7], buf[5 * 18];
    std::vector<double> l7;
    for (int i = 0; i < 5; ++i) {
      Hdetect(F, u7, IDXS[i], H);
      dHDs(H, u7, 7, Ds, l7);
      // sortDs, DegUtils.c:165-183
      unsigned char idx[7];
      memcpy(sDs, Ds, sizeof sDs);
      for (int a = 0; a < 7; ++a) idx[a] = (unsigned char)a;
      for (int a = 0; a < 7; ++a)
        for (int b = a + 1; b < 7; ++b)
          if (sDs[b] < sDs[a]) {
            const double t = sDs[b]; sDs[b] = sDs[a]; sDs[a] = t;
            const unsigned char ti = idx[b]; idx[b] = idx[a]; idx[a] = ti;
          }
      int inl5[5];
      for (int j = 0; j < 5; ++j) inl5[j] = idx[j];
      u2h(u7, inl5, 5, H, buf);
      dHDs(H, u7, 7, Ds, l7);
      int cnt = 0;
      for (int j = 0; j < 7; ++j) if (Ds[j] < th) ++cnt;
      if (cnt > 4) return 1;
    }
    return 0;
  }

  // u2Fit, DegUtils.c:631-686
  unsigned u2Fit(double *F, unsigned char *inl, double th, double ths, unsigned iters) {
    HPROF(2);
    const double dth = (ths - th) / (iters - 1);
    int *inlI = grab(sFitInl, len);
    double *Ds = grab(sFitDs, len), *buf = grab(sFitBuf, (size_t)9 * len);
    unsigned no_i;
    for (unsigned iter = 0; iter < iters; ++iter) {
      { HPROF(3); FDs(u, F, Ds, len); }
      no_i = 0;
      for (int i = 0; i < len; ++i) {
        if (Ds[i] < ths) { inl[i] = 1; ++no_i; }
        else inl[i] = 0;
      }
      if (no_i < 8) return no_i;
      no_i = 0;
      for (int i = 0; i < len; ++i) if (inl[i]) inlI[no_i++] = i;
      { HPROF(4); u2f(u, inlI, (int)no_i, F, buf); }
      ths -= dth;
    }
    FDs(u, F, Ds, len);
    no_i = 0;
    for (int i = 0; i < len; ++i) {
      if (Ds[i] < th) { inl[i] = 1; ++no_i; }
      else inl[i] = 0;
    }
    return no_i;
  }
  // dual_sample, DegUtils.c:592-628
  void dual_sample(const double *uA, unsigned lenA, unsigned sA, const double *uB, unsigned lenB, unsigned sB, double *usam) {
    unsigned *pA = grab(sDualA, lenA), *pB = grab(sDualB, lenB);
    for (unsigned i = 0; i < lenA; ++i) pA[i] = i;
    for (unsigned i = 0; i < lenB; ++i) pB[i] = i;
    for (unsigned pos = 0; pos < sA; ++pos) {
      const unsigned idx = (unsigned)(rng.next() % lenA);
      const unsigned t = pA[pos]; pA[pos] = pA[idx]; pA[idx] = t;
    }
    for (unsigned pos = 0; pos < sB; ++pos) {
      const unsigned idx = (unsigned)(rng.next() % lenB);
      const unsigned t = pB[pos]; pB[pos] = pB[idx]; pB[idx] = t;
    }
    for (unsigned i = 0; i < sA; ++i) memcpy(usam + 6 * i, uA + 6 * pA[i], 6 * sizeof(double));
    for (unsigned i = 0; i < sB; ++i) memcpy(usam + 6 * (i + sA), uB + 6 * pB[i], 6 * sizeof(double));
  }
  // innerFH, DegUtils.c:478-589
  void innerFH(const double *uH, unsigned lenH, const double *uO, unsigned lenO, double th, unsigned repCount,
               unsigned sH, unsigned sO, double *F, unsigned char *inl) {
    HPROF(6);
    double aF[9];
    unsigned char *v = grab(sFhV, len);
    double *usam = grab(sFhUsam, (size_t)6 * (sH + sO)), *Ds = grab(sFhDs, len), *buf = grab(sFhBuf, (size_t)9 * (sH + sO));
    int *all = grab(sFhAll, sH + sO);
    for (unsigned i = 0; i < sH + sO; ++i) all[i] = (int)i;
    for (int i = 0; i < 9; ++i) F[i] = 1;
    for (int i = 0; i < len; ++i) inl[i] = 0;
    unsigned max_i = 0, max_s = 0;
    for (unsigned rep = 0; rep < repCount; ++rep) {
      { HPROF(5); dual_sample(uH, lenH, sH, uO, lenO, sO, usam); }
      { HPROF(1); u2f(usam, all, (int)(sH + sO), aF, buf); }
      { HPROF(0); FDs(u, aF, Ds, len); }
      unsigned no_i = 0;
      for (int i = 0; i < len; ++i) {
        if (Ds[i] < th) { v[i] = 1; ++no_i; }
        else v[i] = 0;
      }
      if (max_i < no_i) { memcpy(inl, v, len); memcpy(F, aF, sizeof aF); max_i = no_i; }
      if (no_i > max_s) {
        max_s = no_i;
        no_i = u2Fit(aF, v, th, th * 3, 4);
        if (max_i < no_i) { memcpy(inl, v, len); memcpy(F, aF, sizeof aF); max_i = no_i; }
      }
    }
  }
  // rFtH, DegUtils.c:254-440: F from the plane homography H plus two off-plane correspondences
  unsigned rFtH(const unsigned char *hinl, double th, const double *H, double *F) {
    std::vector<double> Ds(len);
    dHDs(H, u, len, Ds.data(), lin);
    std::vector<unsigned char> nhinl(len), inl(len);
    unsigned nN = 0, nH = 0;
    for (int i = 0; i < len; ++i) {
      nhinl[i] = Ds[i] > 100 * th ? 1 : 0;
      nN += nhinl[i];
      if (hinl[i]) ++nH;
    }
    std::vector<double> uN((size_t)6 * nN + 6), us((size_t)6 * nN + 6), uV((size_t)6 * nN + 6), uH((size_t)6 * nH + 6);
    std::vector<double> DsN(nN + 1);
    std::vector<unsigned char> v(nN + 1);
    unsigned a = 0, b = 0;
    for (int i = 0; i < len; ++i) {
      if (nhinl[i]) {
        memcpy(&uN[6 * a], u + 6 * i, 6 * sizeof(double));
        memcpy(&us[6 * a], u + 6 * i, 3 * sizeof(double));
        us[6 * a + 3] = H[0] * u[6 * i + 3] + H[3] * u[6 * i + 4] + H[6] * u[6 * i + 5];
        us[6 * a + 4] = H[1] * u[6 * i + 3] + H[4] * u[6 * i + 4] + H[7] * u[6 * i + 5];
        us[6 * a + 5] = H[2] * u[6 * i + 3] + H[5] * u[6 * i + 4] + H[8] * u[6 * i + 5];
        ++a;
      }
      if (hinl[i]) { memcpy(&uH[6 * b], u + 6 * i, 6 * sizeof(double)); ++b; }
    }
    std::vector<unsigned> ptr(nN + 1);
    for (unsigned i = 0; i < nN; ++i) ptr[i] = i;
    unsigned max_i = 3, m_i = 4, max_sam = 10000;
    const double conf = .999;
    if (nN < 4 || nH < 6) return 0;
    double Ht[9];
    tr3(Ht, H);
    auto draw = [&](GlibcRandom &g, std::vector<unsigned> &p) {   // the two swaps of one iteration
      for (unsigned pos = 0; pos < 2; ++pos) {
        const unsigned idx = pos + 1 + (unsigned)(g.next() % (nN - pos - 1));
        const unsigned t = p[pos]; p[pos] = p[idx]; p[idx] = t;
      }
    };
    // the body of an iteration once its sample is in ptr[0], ptr[1]; returns the count of the hypothesis
    auto body = [&]() -> unsigned {
      double c1[3], c2[3], ec[3], aFt[9], aFtH[9], aF[9];
      cross3(c1, &us[6 * ptr[0]], &us[6 * ptr[0] + 3]);
      cross3(c2, &us[6 * ptr[1]], &us[6 * ptr[1] + 3]);
      cross3(ec, c1, c2);
      const double nrm = sqrt(ec[0] * ec[0] + ec[1] * ec[1] + ec[2] * ec[2]);
      ec[0] = ec[0] / nrm; ec[1] = ec[1] / nrm; ec[2] = ec[2] / nrm;
      skew_sym(ec, aFt);
      mul3(aFtH, aFt, Ht);
      tr3(aFt, aFtH);
      FDs(uN.data(), aFt, DsN.data(), (int)nN);
      unsigned no_i = 0;
      for (unsigned i = 0; i < nN; ++i) {
        if (DsN[i] < th * 2) { ++no_i; v[i] = 1; }
        else v[i] = 0;
      }
      const unsigned counted = no_i;
      if (no_i > m_i) {
        no_i = 0;
        for (unsigned i = 0; i < nN; ++i)
          if (v[i]) { memcpy(&uV[6 * no_i], &uN[6 * i], 6 * sizeof(double)); ++no_i; }
        m_i = no_i;
        innerFH(uH.data(), nH, uV.data(), no_i, th, 15, 6, 4, aF, inl.data());
        unsigned ninl = 0;
        for (int i = 0; i < len; ++i) if (inl[i]) ++ninl;
        if (ninl > max_i) {
          max_i = ninl;
          memcpy(F, aF, sizeof aF);
          unsigned maxni = 0;
          for (int i = 0; i < len; ++i) if (inl[i] && nhinl[i]) ++maxni;
          const unsigned ns = (unsigned)nsamples((int)maxni, (int)nN, 2, conf);
          max_sam = max_sam > ns ? ns : max_sam;
        }
      }
      return counted;
    };
    // the count of a hypothesis alone (the first half of body(), no state touched)
    auto count_only = [&](unsigned p0, unsigned p1) -> unsigned {
      double c1[3], c2[3], ec[3], aFt[9], aFtH[9];
      cross3(c1, &us[6 * p0], &us[6 * p0 + 3]);
      cross3(c2, &us[6 * p1], &us[6 * p1 + 3]);
      cross3(ec, c1, c2);
      const double nrm = sqrt(ec[0] * ec[0] + ec[1] * ec[1] + ec[2] * ec[2]);
      ec[0] = ec[0] / nrm; ec[1] = ec[1] / nrm; ec[2] = ec[2] / nrm;
      skew_sym(ec, aFt);
      mul3(aFtH, aFt, Ht);
      tr3(aFt, aFtH);
      FDs(uN.data(), aFt, DsN.data(), (int)nN);
      unsigned no_i = 0;
      for (unsigned i = 0; i < nN; ++i) if (DsN[i] < th * 2) ++no_i;
      return no_i;
    };
    const char *chk = getenv("MODSX_VERIFY_DEVICE_CHECK");
    const bool checkAll = chk && atoi(chk) != 0;
    unsigned no_sam = 1;
    struct LoopClock {
      std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
      ~LoopClock() { g_rfthStats[4]++; g_rfthStats[5] += std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - t0).count(); }
    } loopClock;
    RfthLease lease;
    RfthDevice &D = *lease.d;
    bool dev = D.init() && D.points(us.data(), uN.data(), nN);
    if (dev) {
      RfthArgs A;
      memcpy(A.Ht, Ht, sizeof Ht);
      A.th2 = th * 2; A.nN = (int)nN;
      std::vector<unsigned> ptr2;
      // Events (a count above the best so far) come in a burst at the start of a loop -- the best count starts at 4 -- and each one
      // invalidates the rest of a batch drawn ahead of it: a loop used to take ~5 round trips through a busy device.  So the loop
      // runs on the host until `quietMin` hypotheses in a row have changed nothing, and only then goes to the device in batches;
      // an event there sends it back to the host.  (Same samples, same order, same counts: the trajectory does not depend on where
      // a hypothesis was counted.)
      static const unsigned quietMin = getenv("MODSX_RFTH_QUIET") ? (unsigned)atoi(getenv("MODSX_RFTH_QUIET")) : 192;
      unsigned quiet = 0;
      while (dev && no_sam < 2 * max_sam) {
        if (quiet < quietMin) {
          const auto th0 = std::chrono::steady_clock::now();
          draw(rng, ptr);
          const unsigned before = m_i;
          body();
          ++no_sam;
          quiet = m_i != before ? 0 : quiet + 1;
          g_rfthStats[m_i != before ? 9 : 8] += rfth_us(th0);
          continue;
        }
        // the next B samples from a copy of the generator and of the permutation
        const unsigned B = std::min<unsigned>(RfthDevice::BATCH, 2 * max_sam - no_sam);
        const auto td0 = std::chrono::steady_clock::now();
        GlibcRandom g2 = rng;
        ptr2 = ptr;
        for (unsigned j = 0; j < B; ++j) { draw(g2, ptr2); D.hPairs[2 * j] = ptr2[0]; D.hPairs[2 * j + 1] = ptr2[1]; }
        A.B = (int)B;
        g_rfthStats[6] += rfth_us(td0);
        const auto tw0 = std::chrono::steady_clock::now();
        const bool okc = D.count(A);
        g_rfthStats[7] += rfth_us(tw0);
        if (!okc) { D.ok = false; dev = false; break; }      // a HIP error: the rest of the loop (and of the thread's calls) on the host
        g_rfthStats[0]++; g_rfthStats[1] += B;
        if (checkAll) {
          // MODSX_VERIFY_DEVICE_CHECK=1: the whole batch is counted again on the host -- a device UNDER-count would otherwise skip
          // a state change silently (only the hypotheses the device flags are re-run below)
          GlibcRandom g3 = rng;
          std::vector<unsigned> ptr3 = ptr;
          for (unsigned j = 0; j < B; ++j) {
            draw(g3, ptr3);
            if (count_only(ptr3[0], ptr3[1]) != D.hCnt[j]) g_rfthStats[3]++;
          }
        }
        unsigned hit = B;
        for (unsigned j = 0; j < B; ++j) if (D.hCnt[j] > m_i) { hit = j; break; }
        if (hit == B) { rng = g2; ptr.swap(ptr2); no_sam += B; continue; }      // nothing in the batch changes the state
        const auto te0 = std::chrono::steady_clock::now();
        for (unsigned j = 0; j <= hit; ++j) draw(rng, ptr);                       // replay up to the event, then the reference's body
        const unsigned counted = body();
        g_rfthStats[9] += rfth_us(te0);
        no_sam += hit + 1;
        quiet = 0;
        g_rfthStats[2]++;
        if (counted != D.hCnt[hit]) { g_rfthStats[3]++; D.ok = false; dev = false; }   // never seen; the host's count is what was acted on
      }
    }
    for (; no_sam < 2 * max_sam; ++no_sam) {      // the reference's loop (no device, or what is left after a device error)
      draw(rng, ptr);
      body();
    }
    return max_i;
  }

  // exp_iterFcustom, exp_ranF.c:616-733 (inlLimit = 0 => the LSQ subset has 8 points)
  Score iterF(int *inliers, double th, double ths, int iters, double *F, double **errs, int iterID, unsigned inlLimit) {
    double *d = errs[1];
    double f[9];
    Score S = {0, 0}, Ss, maxS;
    std::vector<double> w(len);
    const double dth = (ths - th) / 4;
    auto detached = [&](unsigned have) {
      unsigned dc = (unsigned)(int)(have * 1);
      if (dc > inlLimit) dc = inlLimit;
      if (dc < 8) dc = 8;
      return dc;
    };
    maxS = tr_inlidxs(errs[4], len, th, inliers);
    if (maxS.I < 8) return S;
    S = tr_inlidxs(errs[4], len, th * 2, inliers);
    unsigned dc = detached(S.I);
    if (dc >= S.I) u2f(u, inliers, (int)S.I, f, buffer.data());
    else u2f(u, randsubset(inliers, (int)S.I, (int)dc), (int)dc, f, buffer.data());
    for (int it = 0; it < iters; it++) {
      exfds(u, f, d, w.data(), len);
      S = tr_inlidxs(d, len, th, inliers);
      const uint32_t hash = super_fast_hash((const char *)inliers, (int)(S.I * sizeof(int)));
      const int ret = ht.contains(hash, (int)S.I, iterID);
      if (ret != -1 && ret != iterID) { S.I = 0; S.J = 0; return S; }
      if (ret == -1) ht.insert(hash, (int)S.I, iterID);
      if (score_less(maxS, S)) {
        maxS = S;
        errs[1] = errs[0]; errs[0] = d; d = errs[1];
        memcpy(F, f, sizeof f);
      }
      Ss = tr_inlidxs(d, len, ths * 2, inliers);
      if (Ss.I < 8) return maxS;
      dc = detached(Ss.I);
      if (dc >= Ss.I) u2fw(u, inliers, w.data(), (int)Ss.I, f, buffer.data());
      else u2fw(u, randsubset(inliers, (int)Ss.I, (int)dc), w.data(), (int)dc, f, buffer.data());
      ths -= dth;
    }
    fds(u, f, d, len);
    S = tr_inlidxs(d, len, th, inliers);
    if (score_less(maxS, S)) {
      maxS = S;
      errs[1] = errs[0]; errs[0] = d;
      memcpy(F, f, sizeof f);
    }
    return maxS;
  }
  // exp_inFranicustom, exp_ranF.c:736-792
  Score inFrani(int *inliers, int ninl, double th, double **errs, double *F, int *iterID, unsigned inlLimit) {
    Score S = {0, 0}, maxS = {0, 0};
    double f[9];
    if (ninl < 16) return maxS;
    std::vector<int> intbuff(len);
    unsigned ssiz = (unsigned)ninl / 2;
    if (ssiz > 14) ssiz = 14;
    double *d = errs[2]; errs[2] = errs[0]; errs[0] = d;
    for (int i = 0; i < 10; i++) {
      int *sample = randsubset(inliers, ninl, (int)ssiz);
      u2f(u, sample, (int)ssiz, f, buffer.data());
      fds(u, f, errs[0], len);
      errs[4] = errs[0];
      S = iterF(intbuff.data(), th, 4 * th, 4, f, errs, ++*iterID, inlLimit);
      if (score_less(maxS, S)) {
        maxS = S;
        d = errs[2]; errs[2] = errs[0]; errs[0] = d;
        memcpy(F, f, sizeof f);
      }
    }
    d = errs[2]; errs[2] = errs[0]; errs[0] = d;
    return maxS;
  }
};

// exp_ransacFcustom, exp_ranF.c:795-1192.  error_type 0: Sampson (FDs / exFDs), otherwise the symmetric
// epipolar distance (FDsSym / exFDsSym), matching.cpp:821-846.  data_out: samples, LO runs, DEGENSAC hits.
int ransac_f(const double *u, int len, double th, double conf, int max_sam, double *F, unsigned char *inl, int *data_out,
             int do_lo, unsigned inlLimit, int error_type, int doSymCheck, unsigned seed0) {
  RansacF R;
  R.u = u; R.len = len;
  R.fds = error_type == 0 ? FDs : FDsSym;
  R.exfds = error_type == 0 ? exFDs : exFDsSym;
  R.buffer.resize((size_t)len * 18 + 18 * 16);
  std::vector<int> pool(len), inliers(len);
  std::vector<double> Z((size_t)len * 9), err((size_t)len * 4, 0.0), errorsBest(len, 0.0), d_check(len), HDsv(len);
  double A[81], sol[81], u7[42], H[9], FBest[9], f[9], poly[4], roots[3];
  double *errs[5];
  int nb[18];
  Score maxS = {0, 0}, maxSs = {0, 0}, S = {0, 0};
  int samidxBest[7] = {0, 0, 0, 0, 0, 0, 0};
  int degen_cnt = 0, iter_cnt = 0, iterID = 0, no_sam = 0, bad_model = 0;
  unsigned non_degen = 0;
  for (int i = 0; i < 9; i++) { F[i] = 0; FBest[i] = 0; H[i] = 0; }
  for (int i = 0; i < len; i++) inl[i] = 0;
  R.rng.seed(seed0);
  for (int i = 0; i < len; i++) pool[i] = i;
  int *samidx = pool.data() + len - 7;
  lin_fm(u, Z.data(), pool.data(), len);
  for (int i = 0; i < 4; i++) errs[i] = err.data() + (size_t)i * len;
  errs[4] = errs[3];
  maxS.I = 8; maxSs.I = 8;
  double *f1 = sol, *f2 = sol + 9, *d = errs[3];
  unsigned seed = (unsigned)R.rng.next();
  int last_i = 0;  // the loop index the reference leaves behind for its ALO branch

  // the plane-and-parallax branch shared by the main loop and the final ALO step (exp_ranF.c:946-1001, 1066-1111)
  auto degenerate_update = [&](unsigned I, double *fcur, double *derr_alt, int &new_max) {
    if (I > 6) {
      { HPROF(15); I = R.rFtH(inl, th, H, fcur); }
      RTRACE("rFtH %u F0 %.17g\n", I, fcur[0]);
      if (I > maxS.I) {
        R.fds(u, fcur, errs[3], len);
        maxS.I = I;
        memcpy(F, fcur, 9 * sizeof(double));
        new_max = 1;
        d = errs[3];
      } else {
        R.fds(u, fcur, derr_alt, len);
        d = derr_alt;
      }
      double jj = 0;
      for (int j = 0; j < len; j++) jj += trunc_quad(d[j], th);
      if (new_max) maxS.J = jj;
      ++degen_cnt;
    }
  };
  auto lsq_and_lo = [&](const double *base, int *sidx) {  // __LSQ_BEFORE_LO__ + exp_inFranicustom
    HPROF(16);
    (void)sidx;
    d = errs[0];
    S = tr_inlidxs(base, len, 4 * th * 2, inliers.data());
    u2f(u, inliers.data(), (int)S.I, f, R.buffer.data());
    R.fds(u, f, d, len);
    S = tr_inlidxs(d, len, th, inliers.data());
    S = R.inFrani(inliers.data(), (int)S.I, th, errs, f, &iterID, inlLimit);
  };

  while (no_sam < max_sam) {
    no_sam++;
    R.rng.seed(seed);
    // rsampleT(Z, 9, pool, 7, len, A), rtools.c:74-92 with sample() :14-25
    for (int i = 0; i < 7; i++) {
      const int s = (int)(R.rng.next() % (len - i));
      const int j = len - i - 1;
      const int q = pool[s]; pool[s] = pool[j]; pool[j] = q;
      for (int c = 0; c < 9; c++) A[i * 9 + c] = Z[(size_t)c * len + q];
    }
    for (int i = 0; i < 7; i++) memcpy(u7 + 6 * i, u + 6 * samidx[i], 6 * sizeof(double));
    seed = (unsigned)R.rng.next();
    for (int i = 63; i < 81; ++i) A[i] = 0.0;
    memset(sol, 0, sizeof sol);
    const int nullsize = nullspace(A, f1, 9, nb);
    RTRACE("null %d\n", nullsize);
    if (nullsize != 2) { last_i = 3; continue; }
    slcm(f1, f2, poly);
    const int nsol = rroots3(poly, roots);
    RTRACE("roots %d %.17g\n", nsol, nsol ? roots[0] : 0.0);
    int new_max = 0, do_iterate = 0, LmaxI = 0, i;
    for (i = 0; i < nsol; i++) {
      for (int j = 0; j < 9; j++) f[j] = f1[j] * roots[i] + f2[j] * (1 - roots[i]);
      { const int ov = all_ori_valid(f, u, samidx, 7); RTRACE("ori %d\n", ov); if (!ov) continue; }
      d = errs[i];
      { HPROF(13); R.fds(u, f, d, len);
      S = tr_inlidxs(d, len, th, inliers.data()); }
      if ((int)S.I > LmaxI) LmaxI = (int)S.I;
      if (score_less(maxS, S)) {
        if (doSymCheck) {
          FDsSym(u, f, d_check.data(), len);
          unsigned cnt = 0;
          bad_model = 0;
          const int SI_min = (int)floor(0.6 * S.I);
          const double th_check = 4.0 * th;
          for (int j = 0; j < len; j++) if (d_check[j] <= th_check) cnt++;
          if ((int)cnt <= SI_min) bad_model = 1;
        }
        if (bad_model) continue;
        errs[i] = errs[3];
        errs[3] = d;
        maxS = S;
        memcpy(F, f, 9 * sizeof(double));
        new_max = 1;
      }
      if (score_less(maxSs, S)) {
        maxSs = S;
        const int cs = R.checksample(f, u7, 3 * th, H);
        RTRACE("cs %d H0 %.17g\n", cs, H[0]);
        if (cs) {
          dHDs(H, u, len, HDsv.data(), R.lin);
          unsigned I = 0;
          for (int j = 0; j < len; ++j) if (HDsv[j] < th * 3) ++I;
          if (I < 8) break;
          { HPROF(14); I = R.innerH(H, 16 * th, 10, inl); }
          RTRACE("innerH %u H0 %.17g\n", I, H[0]);
          degenerate_update(I, f, errs[i], new_max);
        } else {
          do_iterate = (do_lo > 0 && (no_sam > 50));
          errs[4] = d;
          non_degen++;
          memcpy(samidxBest, samidx, 7 * sizeof(int));
          memcpy(errorsBest.data(), d, len * sizeof(double));
          memcpy(FBest, f, 9 * sizeof(double));
        }
      }
    }
    last_i = i;
    if (do_lo > 0 && (no_sam == 50) && non_degen) do_iterate = 1;
    if (do_iterate) {
      iter_cnt++;
      lsq_and_lo(errs[4], samidx);
      if (score_less(maxS, S)) {
        d = errs[0]; errs[0] = errs[3]; errs[3] = d;
        maxS = S;
        memcpy(F, f, 9 * sizeof(double));
        new_max = 1;
      }
    }
    if (new_max) {
      const int new_sam = nsamples((int)maxS.I + 1, len, 7, conf);
      RTRACE("nsamples %d %d -> %d\n", (int)maxS.I + 1, len, new_sam);
      if (new_sam < max_sam) max_sam = new_sam;
    }
  }

  if (do_lo && (!iter_cnt && !degen_cnt) && non_degen) {
    for (int i = 0; i < 7; i++) memcpy(u7 + 6 * i, u + 6 * samidxBest[i], 6 * sizeof(double));
    const int csA = R.checksample(FBest, u7, 3 * th, H);
    RTRACE("cs %d H0 %.17g\n", csA, H[0]);
    if (csA) {
      dHDs(H, u, len, HDsv.data(), R.lin);
      unsigned I = 0;
      for (int j = 0; j < len; ++j) if (HDsv[j] < th * 3) ++I;
      if (I >= 8) { I = R.innerH(H, 16 * th, 10, inl); RTRACE("innerH %u H0 %.17g\n", I, H[0]); }
      int new_max = 0;
      degenerate_update(I, f, errs[last_i < 3 ? last_i : 3], new_max);
    } else {
      iter_cnt++;
      lsq_and_lo(errorsBest.data(), samidxBest);
      if (score_less(maxS, S)) {
        d = errs[0]; errs[0] = errs[3]; errs[3] = d;
        maxS = S;
        memcpy(F, f, 9 * sizeof(double));
      }
    }
  }
  d = errs[3];
  int ninl = 0;
  for (int j = 0; j < len; j++) { inl[j] = d[j] <= th ? 1 : 0; ninl += inl[j]; }
  data_out[0] = no_sam; data_out[1] = iter_cnt; data_out[2] = degen_cnt;
  return (int)maxS.I;
}

// LORANSACFiltering with useF = 1 (matching.cpp:806-980): exp_ransacFcustom on (x1 y1 1 x2 y2 1), then
// F_LAF_check (:193-250): the two extra points of each local affine frame (k_sigma = 3) must also satisfy F.
int loransac_f(const double *pts, const double *laf1, const double *laf2, int T, double err_threshold, double confidence,
               int max_samples, int lo, double LAFCoef, int doSymmCheck, int error_type, unsigned seed, double *F,
               unsigned char *inl, unsigned char *keep, int *data_out3) {
  for (int i = 0; i < T; i++) { inl[i] = 0; keep[i] = 0; }
  for (int i = 0; i < 9; i++) F[i] = 0;
  data_out3[0] = data_out3[1] = data_out3[2] = 0;
  if (T < 8) return 0;  // MIN_POINTS, matching.hpp:27
  std::vector<double> u2((size_t)T * 6);
  for (int i = 0; i < T; i++) {
    u2[6 * i] = pts[4 * i]; u2[6 * i + 1] = pts[4 * i + 1]; u2[6 * i + 2] = 1.;
    u2[6 * i + 3] = pts[4 * i + 2]; u2[6 * i + 4] = pts[4 * i + 3]; u2[6 * i + 5] = 1.;
  }
  { HPROF(17);
  ransac_f(u2.data(), T, err_threshold * err_threshold, confidence, max_samples, F, inl, data_out3, lo, 0, error_type,
           doSymmCheck, seed); }
  const FdsFn fds = error_type == 0 ? FDs : FDsSym;
  const double affErr = LAFCoef * err_threshold;
  int kept = 0;
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
    keep[i] = 1;
    kept++;
  }
  if (kept < 8) { for (int i = 0; i < T; i++) keep[i] = 0; kept = 0; }
  return kept;
}

}  // namespace mx

// out[0..4): device batches of rFtH's hypothesis loop, hypotheses counted on the device, state-changing hypotheses (each re-run on
// the host), host / device disagreements (never seen: one switches the calling thread back to the host loop).  Process-wide.
#ifdef MODSX_HPROF
extern "C" __attribute__((visibility("default"))) int modsx_debug_hprof(long *out, int reset) {
  for (int k = 0; k < 2; k++) for (int i = 0; i < 24; i++) { out[k * 24 + i] = mx::g_hprof[k][i].load(); if (reset) mx::g_hprof[k][i] = 0; }
  return 0;
}
#endif
extern "C" __attribute__((visibility("default"))) int modsx_verify_device_stats(long *out, int reset) {
  if (!out) return -1;
  for (int i = 0; i < 6; i++) { out[i] = mx::g_rfthStats[i].load(); if (reset) mx::g_rfthStats[i].store(0); }
  return 6;
}
// out[0..4): microseconds of the rFtH loops spent drawing samples ahead of a batch, waiting for the device, in the host phase
// (hypotheses that changed nothing), in the bodies of state-changing hypotheses.  Process-wide; reset clears.
extern "C" __attribute__((visibility("default"))) int modsx_verify_device_timing(long *out, int reset) {
  if (!out) return -1;
  for (int i = 0; i < 4; i++) { out[i] = mx::g_rfthStats[6 + i].load(); if (reset) mx::g_rfthStats[6 + i].store(0); }
  return 4;
}
