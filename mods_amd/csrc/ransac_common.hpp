// ransac_common.hpp -- numerical helpers shared by the homography and the epipolar LO-RANSAC (host C++).
// Each function names the reference routine it restates (degensac/*.c, matutls/*.c).
#pragma once
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

namespace mx {


// ---- glibc random_r TYPE_3 (srandom_r / random_r, stdlib/random_r.c) -------------------------
struct GlibcRandom {
  int32_t st[31];
  int f, r;
  void seed(unsigned s) {
    if (s == 0) s = 1;
    st[0] = (int32_t)s;
    int32_t word = (int32_t)s;
    for (int i = 1; i < 31; i++) {
      long hi = word / 127773, lo = word % 127773;
      long w = 16807 * lo - 2836 * hi;
      if (w < 0) w += 2147483647;
      word = (int32_t)w;
      st[i] = word;
    }
    f = 3; r = 0;
    for (int i = 0; i < 310; i++) next();
  }
  long next() {
    uint32_t val = (uint32_t)st[f] + (uint32_t)st[r];
    st[f] = (int32_t)val;
    long result = (long)(val >> 1);
    ++f;
    if (f >= 31) { f = 0; ++r; }
    else { ++r; if (r >= 31) r = 0; }
    return result;
  }
};

struct Score { unsigned I; double J; };

// rtools.c:228-236
static inline double trunc_quad(double epsilon, double thr) {
  if (thr == 0) return 0;
  if (epsilon >= thr * 9 / 4) return 0;
  return 1 - (epsilon / (thr * 9 / 4));
}
static inline bool score_less(const Score &a, const Score &b) { return a.J < b.J; }  // SC_M, rtools.c:238-250
// rtools.c:160-171
// The score is a running f64 sum in point order (kept); the terms are independent, so they are formed block-wise in a loop
// the compiler vectorises (one division per point) and added afterwards.
static inline Score inlidxs(const double *err, int len, double th, int *inl) {
  Score s = {0, 0};
  double term[256];
  unsigned cnt = 0;
  for (int i0 = 0; i0 < len; i0 += 256) {
    const int m = len - i0 < 256 ? len - i0 : 256;
    for (int i = 0; i < m; ++i) term[i] = trunc_quad(err[i0 + i], th);
    for (int i = 0; i < m; ++i) s.J += term[i];
    // the index list without a branch per point (a third of the points are outliers in no particular order): the slot is
    // written every time and kept when the point is an inlier; inl[] has len entries and cnt <= i0 + i
    for (int i = 0; i < m; ++i) { inl[cnt] = i0 + i; cnt += err[i0 + i] <= th; }
  }
  s.I = cnt;
  return s;
}
// the same list where the caller reads only the count and the indices (S.I of exp_iterHcustom's 2 * th sets): no score sum
static inline unsigned inl_list(const double *err, int len, double th, int *inl) {
  unsigned cnt = 0;
  for (int i = 0; i < len; ++i) { inl[cnt] = i; cnt += err[i] <= th; }
  return cnt;
}
// rtools.c:202-225
static inline int nsamples(int ninl, int ptNum, int samsiz, double conf) {
  const double EPS = 2.2204e-16;
  const int MAXS = 1000000;
  double a = 1, b = 1;
  for (int i = 0; i < samsiz; i++) { a *= ninl - i; b *= ptNum - i; }
  a = a / b;
  if (a < EPS) return MAXS;
  a = 1 - a;
  if (a < EPS) return 1;
  b = log(1 - conf) / log(a);
  if (b > MAXS) return MAXS;
  return (int)ceil(b);
}

// SuperFastHash, degensac/hash.c:4-54 (get16bits = little-endian 16-bit load)
static inline uint32_t super_fast_hash(const char *data, int len) {
  uint32_t hash = (uint32_t)len, tmp;
  if (len <= 0 || data == 0) return 0;
  auto g16 = [](const char *d) { return (uint32_t)(((uint32_t)((const uint8_t *)d)[1] << 8) + (uint32_t)((const uint8_t *)d)[0]); };
  int rem = len & 3;
  len >>= 2;
  for (; len > 0; len--) {
    hash += g16(data);
    tmp = (g16(data + 2) << 11) ^ hash;
    hash = (hash << 16) ^ tmp;
    data += 4;
    hash += hash >> 11;
  }
  switch (rem) {
    case 3: hash += g16(data); hash ^= hash << 16; hash ^= ((uint32_t)(int32_t)(signed char)data[2]) << 18; hash += hash >> 11; break;
    case 2: hash += g16(data); hash ^= hash << 11; hash += hash >> 17; break;
    case 1: hash += (uint32_t)(int32_t)(signed char)*data; hash ^= hash << 10; hash += hash >> 1;
  }
  hash ^= hash << 3; hash += hash >> 5; hash ^= hash << 4; hash += hash >> 17; hash ^= hash << 25; hash += hash >> 6;
  return hash;
}
// htInsert / htContains, degensac/hash.c:76-99 (64 chained buckets, newest first)
struct HashTable {
  struct Field { uint32_t hash; int length, iterID; };
  std::vector<Field> b[64];
  void insert(uint32_t hash, int length, int iterID) { b[hash % 64].push_back({hash, length, iterID}); }
  int contains(uint32_t hash, int length, int iterID) const {
    const std::vector<Field> &v = b[hash % 64];
    for (size_t i = v.size(); i-- > 0;)
      if (v[i].hash == hash && v[i].length == length && v[i].iterID == iterID) return iterID;
    for (size_t i = v.size(); i-- > 0;)
      if (v[i].hash == hash && v[i].length == length) return v[i].iterID;
    return -1;
  }
};

// det3, utools.c:196-202
static inline double det3(const double *A) {
  double r = (A[0] * A[4] * A[8] + A[2] * A[3] * A[7] + A[1] * A[5] * A[6]);
  r -= (A[2] * A[4] * A[6] + A[0] * A[5] * A[7] + A[1] * A[3] * A[8]);
  return r;
}

// nullspace (Gauss-Jordan with partial pivoting, row-wise matrix), utools.c:97-167
static inline int nullspace(double *matrix, double *ns, int n, int *buffer) {
  int *pnopivot = buffer, nonpivot = 0;
  int *ppivot = buffer + n;
  int i = 0, j, k, l, max;
  double pivot, t;
  const double tol = 1e-12;
  for (j = 0; j < n; j++) {
    pivot = fabs(matrix[n * i + j]); max = i;
    for (k = i + 1; k < n; k++) {
      t = fabs(matrix[n * k + j]);
      if (pivot < t) { pivot = t; max = k; }
    }
    if (pivot < tol) {
      *(pnopivot++) = j; nonpivot++;
      for (k = i; k < n; k++) matrix[n * k + j] = 0;
    } else {
      *(ppivot++) = j;
      for (k = j; k < n; k++) { t = matrix[i * n + k]; matrix[i * n + k] = matrix[max * n + k]; matrix[max * n + k] = t; }
      pivot = matrix[i * n + j];
      for (k = j; k < n; k++) matrix[i * n + k] /= pivot;
      for (k = 0; k < i; k++) {
        pivot = -matrix[k * n + j];
        for (l = j; l < n; l++) matrix[k * n + l] += pivot * matrix[i * n + l];
      }
      for (k = i + 1; k < n; k++) {
        pivot = matrix[k * n + j];
        for (l = j; l < n; l++) matrix[k * n + l] -= pivot * matrix[i * n + l];
      }
      i++;
    }
  }
  for (k = 0; k < nonpivot; k++) {
    j = buffer[k];
    for (l = 0; l < n - nonpivot; l++) ns[k * n + buffer[n + l]] = -matrix[l * n + j];
    for (l = 0; l < nonpivot; l++) ns[k * n + buffer[l]] = (j == buffer[l]) ? 1 : 0;
  }
  return nonpivot;
}

// lin_hg: column-wise 2len x 9 linearisation of u' = H u, Htools.c:17-54.  Column c of the row pair of
// point i: c = 3j -> (x'_j, 0), c = 3j+1 -> (0, x'_j), c = 3j+2 -> (-x x'_j, -y x'_j) with x' = (x2,y2,1),
// i.e. h is the column-major 3x3.
static inline void lin_hg(const double *u, double *dst, const int *inl, int len) {
  const size_t len2 = 2 * (size_t)len;
  for (int i = 0; i < len; i++) {
    const double *s = u + 6 * inl[i];
    double *r0 = dst + 2 * i, *r1 = dst + 2 * i + 1;
    for (int j = 0; j < 3; j++) {
      r0[(3 * j) * len2] = s[3 + j];
      r0[(3 * j + 1) * len2] = 0;
      r0[(3 * j + 2) * len2] = -s[0] * s[3 + j];
      r1[(3 * j) * len2] = 0;
      r1[(3 * j + 1) * len2] = s[3 + j];
      r1[(3 * j + 2) * len2] = -s[1] * s[3 + j];
    }
  }
}

// normu, utools.c:7-52.  The two sums of distances keep their order; the distances themselves (two square roots per point)
// are formed four points at a time in front of the additions.
#if defined(__x86_64__)
// the distance pass of normu eight points at a time (one 512-bit square root per image side); the sums keep their order.
// Returns the number of points done.
__attribute__((target("avx512f"))) static inline int normu_dist8(const double *u, const int *inl, int len, double m1x, double m1y, double m2x,
                                                                 double m2y, double *s1io, double *s2io) {
  typedef double nv8 __attribute__((vector_size(64)));
  const nv8 M1x = {m1x, m1x, m1x, m1x, m1x, m1x, m1x, m1x}, M1y = {m1y, m1y, m1y, m1y, m1y, m1y, m1y, m1y};
  const nv8 M2x = {m2x, m2x, m2x, m2x, m2x, m2x, m2x, m2x}, M2y = {m2y, m2y, m2y, m2y, m2y, m2y, m2y, m2y};
  double s1 = *s1io, s2 = *s2io;
  int j = 0;
  for (; j + 8 <= len; j += 8) {
    const double *p[8];
    for (int k = 0; k < 8; k++) p[k] = u + 6 * inl[j + k];
    const nv8 ax = (nv8){p[0][0], p[1][0], p[2][0], p[3][0], p[4][0], p[5][0], p[6][0], p[7][0]} - M1x;
    const nv8 ay = (nv8){p[0][1], p[1][1], p[2][1], p[3][1], p[4][1], p[5][1], p[6][1], p[7][1]} - M1y;
    const nv8 bx = (nv8){p[0][3], p[1][3], p[2][3], p[3][3], p[4][3], p[5][3], p[6][3], p[7][3]} - M2x;
    const nv8 by = (nv8){p[0][4], p[1][4], p[2][4], p[3][4], p[4][4], p[5][4], p[6][4], p[7][4]} - M2y;
    nv8 q1 = ax * ax + ay * ay, q2 = bx * bx + by * by;
    for (int k = 0; k < 8; k++) { q1[k] = sqrt(q1[k]); q2[k] = sqrt(q2[k]); }   // one vsqrtpd each
    for (int k = 0; k < 8; k++) { s1 += q1[k]; s2 += q2[k]; }
  }
  *s1io = s1; *s2io = s2;
  return j;
}
#endif
static inline void normu(const double *u, const int *inl, int len, double *A1, double *A2) {
  typedef double nv4 __attribute__((vector_size(32)));
  for (int j = 0; j < 3; j++) { A1[j] = 0; A2[j] = 0; }
  {
    double x1 = 0, y1 = 0, x2 = 0, y2 = 0;    // in registers: A1 / A2 may alias u to the compiler, a store and a load per addition
    for (int j = 0; j < len; j++) {
      const double *p = u + 6 * inl[j];
      x1 += p[0]; y1 += p[1];
      x2 += p[3]; y2 += p[4];
    }
    A1[1] = x1; A1[2] = y1; A2[1] = x2; A2[2] = y2;
  }
  if (len > 0)
    for (int i = 1; i < 3; i++) { A1[i] /= len; A2[i] /= len; }
  const nv4 m1x = {A1[1], A1[1], A1[1], A1[1]}, m1y = {A1[2], A1[2], A1[2], A1[2]};
  const nv4 m2x = {A2[1], A2[1], A2[1], A2[1]}, m2y = {A2[2], A2[2], A2[2], A2[2]};
  double s1 = A1[0], s2 = A2[0];
  int j = 0;
#if defined(__x86_64__)
  static const bool wide = __builtin_cpu_supports("avx512f");
  if (wide) j = normu_dist8(u, inl, len, A1[1], A1[2], A2[1], A2[2], &s1, &s2);
#endif
  for (; j + 4 <= len; j += 4) {
    const double *p0 = u + 6 * inl[j], *p1 = u + 6 * inl[j + 1], *p2 = u + 6 * inl[j + 2], *p3 = u + 6 * inl[j + 3];
    const nv4 ax = (nv4){p0[0], p1[0], p2[0], p3[0]} - m1x, ay = (nv4){p0[1], p1[1], p2[1], p3[1]} - m1y;
    const nv4 bx = (nv4){p0[3], p1[3], p2[3], p3[3]} - m2x, by = (nv4){p0[4], p1[4], p2[4], p3[4]} - m2y;
    nv4 q1 = ax * ax + ay * ay, q2 = bx * bx + by * by;
    for (int k = 0; k < 4; k++) { q1[k] = sqrt(q1[k]); q2[k] = sqrt(q2[k]); }   // one vsqrtpd each
    for (int k = 0; k < 4; k++) { s1 += q1[k]; s2 += q2[k]; }
  }
  for (; j < len; j++) {
    const double *p = u + 6 * inl[j];
    double a = p[0] - A1[1], b = p[1] - A1[2];
    s1 += sqrt(a * a + b * b);
    a = p[3] - A2[1]; b = p[4] - A2[2];
    s2 += sqrt(a * a + b * b);
  }
  A1[0] = s1; A2[0] = s2;
  if (A1[0] != 0) A1[0] = len * sqrt(2) / A1[0];
  if (A2[0] != 0) A2[0] = len * sqrt(2) / A2[0];
  A1[1] *= -A1[0]; A1[2] *= -A1[0];
  A2[1] *= -A2[0]; A2[2] *= -A2[0];
}

// lin_hgN: row-wise 2len x 9 normalised linearisation, Htools.c:56-96.  (lin_hgN and cov_mat_hgN are no longer called: u2h
// forms the same sums without the matrix, cov_hgN_fused below; they stay as the plain statement of what it computes.)
static inline void lin_hgN(const double *u, double *p, const int *inl, int len, const double *A1, const double *A2) {
  double a[3], b[3];
  a[2] = 1; b[2] = 1;
  for (int i = 0; i < len; i++) {
    const double *s = u + 6 * inl[i];
    a[0] = s[0] * A1[0] + A1[1];
    a[1] = s[1] * A1[0] + A1[2];
    b[0] = s[3] * A2[0] + A2[1];
    b[1] = s[4] * A2[0] + A2[2];
    double *r0 = p + (size_t)18 * i, *r1 = r0 + 9;
    for (int j = 0; j < 3; j++) {
      r0[3 * j] = b[j]; r0[3 * j + 1] = 0; r0[3 * j + 2] = -a[0] * b[j];
      r1[3 * j] = 0; r1[3 * j + 1] = b[j]; r1[3 * j + 2] = -a[1] * b[j];
    }
  }
}

// cov_mat, utools.c:170-183: Cv[i][j] = sum_k Z[k][i] * Z[k][j].  The row loop is hoisted outside so the 45
// running sums advance together; each individual sum still adds its terms in row order k = 0, 1, ...
// siz == 9 (every caller in the pipeline): row i of the lower triangle is z[i] * (z[0], ..., z[i]), formed four columns at a time
// (15 packed multiply + add pairs per row of Z instead of 45 scalar ones; the lanes beyond column i hold products nobody
// reads).  A lane is one sum: the same products added in the same order, so the matrix is the same bits.  (In the WxBS bench this
// routine and the linearisation in front of it were 44 % of the F verification's host time: tools/hprof_bench.py.)
static inline void cov_mat(double *Cv, const double *Z, int len, int siz) {
  if (siz == 9) {
    typedef double cv4 __attribute__((vector_size(32), aligned(8)));
    cv4 a0[9], a1[5], a2 = {0, 0, 0, 0};          // a0[i]: columns 0-3 of row i; a1[i - 4]: columns 4-7 of rows 4-8; a2: column 8 of row 8
    for (int i = 0; i < 9; i++) a0[i] = (cv4){0, 0, 0, 0};
    for (int i = 0; i < 5; i++) a1[i] = (cv4){0, 0, 0, 0};
    int k = 0;
    for (; k + 1 < len; k++) {        // the 12-wide loads of a row reach 3 doubles into the next one: the last row goes through the tail below
      const double *z = Z + (size_t)k * 9;
      const cv4 v0 = *reinterpret_cast<const cv4 *>(z), v1 = *reinterpret_cast<const cv4 *>(z + 4), v2 = *reinterpret_cast<const cv4 *>(z + 8);
#pragma unroll
      for (int i = 0; i < 9; i++) {
        const cv4 zi = {z[i], z[i], z[i], z[i]};
        a0[i] += zi * v0;
        if (i >= 4) a1[i - 4] += zi * v1;
        if (i == 8) a2 += zi * v2;
      }
    }
    double acc[9][9];
    for (int i = 0; i < 9; i++) {
      for (int j = 0; j < 4; j++) acc[i][j] = a0[i][j];
      if (i >= 4) for (int j = 0; j < 4; j++) acc[i][4 + j] = a1[i - 4][j];
    }
    acc[8][8] = a2[0];
    for (; k < len; k++) {
      const double *z = Z + (size_t)k * 9;
      for (int i = 0; i < 9; i++) {
        const double zi = z[i];
        for (int j = 0; j <= i; j++) acc[i][j] += zi * z[j];
      }
    }
    if (len <= 0) for (int i = 0; i < 9; i++) for (int j = 0; j <= i; j++) acc[i][j] = 0;
    for (int i = 0; i < 9; i++)
      for (int j = 0; j <= i; j++) { Cv[9 * i + j] = acc[i][j]; Cv[i + 9 * j] = acc[i][j]; }
    return;
  }
  double acc[9][9];
  for (int i = 0; i < 9; i++) for (int j = 0; j < 9; j++) acc[i][j] = 0;
  for (int k = 0; k < len; k++) {
    const double *z = Z + (size_t)k * siz;
    for (int i = 0; i < siz; i++) {
      const double zi = z[i];
      for (int j = 0; j <= i; j++) acc[i][j] += zi * z[j];
    }
  }
  for (int i = 0; i < siz; i++)
    for (int j = 0; j <= i; j++) { Cv[siz * i + j] = acc[i][j]; Cv[i + siz * j] = acc[i][j]; }
}

// cov_mat for the 2len x 9 matrix lin_hgN builds: row 2i is zero in columns 1, 4, 7 and row 2i + 1 in columns 0, 3, 6 (exact
// zeros by construction).  A product with such a zero is +-0 and adding +-0 to a running sum that starts at +0 never changes
// it (x + -x is +0 in round-to-nearest), so those 24 of the 45 products per row are left out; every sum still takes its
// remaining terms in row order.
static inline void cov_mat_hgN(double *Cv, const double *Z, int npts) {
  double acc[9][9];
  for (int i = 0; i < 9; i++) for (int j = 0; j < 9; j++) acc[i][j] = 0;
  static const int nz0[6] = {0, 2, 3, 5, 6, 8}, nz1[6] = {1, 2, 4, 5, 7, 8};
  for (int k = 0; k < npts; k++) {
    const double *z = Z + (size_t)18 * k;
    for (int a = 0; a < 6; a++) { const int i = nz0[a]; const double zi = z[i]; for (int b = 0; b <= a; b++) acc[i][nz0[b]] += zi * z[nz0[b]]; }
    z += 9;
    for (int a = 0; a < 6; a++) { const int i = nz1[a]; const double zi = z[i]; for (int b = 0; b <= a; b++) acc[i][nz1[b]] += zi * z[nz1[b]]; }
  }
  for (int i = 0; i < 9; i++)
    for (int j = 0; j <= i; j++) { Cv[9 * i + j] = acc[i][j]; Cv[i + 9 * j] = acc[i][j]; }
}

// ---- symmetric Jacobi eigen-solver, n <= 9: eigenvalues in ev, eigenvectors in the columns of V ----------
// Cyclic sweeps over (p, q), a rotation = a column pass, a row pass and the same pass over the eigenvectors.  The size is a
// template constant and the eigenvectors are kept transposed while the sweeps run, so two of the three passes of a rotation are
// over contiguous rows (4-wide vector code); every element still goes through the same multiplications and additions in the same
// order as in the plain three-loop form.
// The stopping rule.  RULE 0: `off <= 1e-32 diag` alone -- a threshold AT the rounding floor (eps^2), where the off-diagonal mass
// hovers: 15-28 sweeps for the 9 x 9 matrices of u2f / u2h, of which all but the first ~5 rotate by angles below an ulp and move
// nothing but off-diagonal dust.  RULE 2 (in use): additionally stop after the first sweep that leaves the diagonal AND every
// eigenvector component bit for bit where they were -- the later sweeps of RULE 0 only shrink the dust further, so the eigenpairs
// are RULE 0's bits (checked: -DMODSX_JACOBI_CHECK runs both on every call and counts smallest-eigenpair differences; 0 over the
// sweeps of tools/sweep_ransac_ref.py, near-degenerate planar scenes included) at 4.6 sweeps per call instead of 15.4.
// RULE 1 (one more sweep once off <= 1e-22 diag: "quadratic convergence") is NOT safe: with two close eigenvalues a tiny
// off-diagonal entry still means a large rotation, and 3 of 2 598 calls of one planar problem ended on another eigenvector --
// another F, 30 other inliers than the reference's degensac.  It stays here as the counter-example the check build measures.
constexpr double JACOBI_LAST = 1e-22;
#ifndef JACOBI_RULE
#define JACOBI_RULE 2
#endif
template <int N, int RULE>
static int jacobi_eig_rule(const double *C, double *ev, double *V) {
  alignas(32) double A[N * N], VT[N * N];
  for (int i = 0; i < N * N; i++) A[i] = C[i];
  for (int i = 0; i < N; i++) for (int j = 0; j < N; j++) VT[i * N + j] = (i == j);
  bool last = false;
  int sweep = 0;
  for (; sweep < 100; sweep++) {
    double off = 0, diag = 0;
    for (int i = 0; i < N; i++) for (int j = 0; j < N; j++) (i == j ? diag : off) += A[i * N + j] * A[i * N + j];
    if (off <= 1e-32 * diag || off == 0 || last) break;
    if (RULE == 1) last = off <= JACOBI_LAST * diag;
    bool changed = false;      // RULE 2: did this sweep move the diagonal or an eigenvector component at all?
    for (int p = 0; p < N - 1; p++)
      for (int q = p + 1; q < N; q++) {
        const double apq = A[p * N + q];
        if (apq == 0) continue;
        const double theta = (A[q * N + q] - A[p * N + p]) / (2 * apq);
        const double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1));
        const double cs = 1 / sqrt(t * t + 1), sn = t * cs;
        const double dp0 = A[p * N + p], dq0 = A[q * N + q];
#pragma unroll
        for (int k = 0; k < N; k++) {
          const double akp = A[k * N + p], akq = A[k * N + q];
          A[k * N + p] = cs * akp - sn * akq;
          A[k * N + q] = sn * akp + cs * akq;
        }
        double *Ap = A + p * N, *Aq = A + q * N, *Vp = VT + p * N, *Vq = VT + q * N;
#pragma unroll
        for (int k = 0; k < N; k++) {
          const double apk = Ap[k], aqk = Aq[k];
          Ap[k] = cs * apk - sn * aqk;
          Aq[k] = sn * apk + cs * aqk;
        }
        if (RULE == 2 && (A[p * N + p] != dp0 || A[q * N + q] != dq0)) changed = true;
#pragma unroll
        for (int k = 0; k < N; k++) {
          const double vkp = Vp[k], vkq = Vq[k];
          const double np_ = cs * vkp - sn * vkq, nq_ = sn * vkp + cs * vkq;
          if (RULE == 2 && (np_ != vkp || nq_ != vkq)) changed = true;
          Vp[k] = np_;
          Vq[k] = nq_;
        }
      }
    if (RULE == 2 && !changed) last = true;
  }
  for (int i = 0; i < N; i++) ev[i] = A[i * N + i];
  for (int i = 0; i < N; i++) for (int j = 0; j < N; j++) V[i * N + j] = VT[j * N + i];
  return sweep;
}

// eigenvector of the smallest eigenvalue of a symmetric 9x9 (stands in for lap_eig = dsyev_, whose
// first returned column is that vector: lapwrap.c:62-97, Htools.c:118-121)
static inline void smallest_eigvec9(const double *C, double *v) {
  double ev[9], V[81];
  jacobi_eig_rule<9, JACOBI_RULE>(C, ev, V);
  int best = 0;
  for (int i = 1; i < 9; i++) if (ev[i] < ev[best]) best = i;
  for (int k = 0; k < 9; k++) v[k] = V[k * 9 + best];
}

// denormH, utools.c:74-92 (F[] is the 3x3 in the _f1.._f9 order = F[0..8])
static inline void denormH(double *F, const double *A1, const double *A2) {
  double r = A2[0], x = A2[1], y = A2[2];
  F[6] += x * F[0] + y * F[3];
  F[7] += x * F[1] + y * F[4];
  F[8] += x * F[2] + y * F[5];
  F[0] *= r; F[1] *= r; F[2] *= r;
  F[3] *= r; F[4] *= r; F[5] *= r;
  r = 1 / A1[0]; x = -A1[1] * r; y = -A1[2] * r;
  for (int i = 0; i < 9; i += 3) {
    F[i] = r * F[i] + x * F[i + 2];
    F[i + 1] = r * F[i + 1] + y * F[i + 2];
  }
}

// lin_hgN + cov_mat_hgN in one pass, without the 2len x 9 matrix in memory: the 45 sums of Z^T Z, each with its terms in row
// order (row 2i of point i, then row 2i + 1).  With b = (b0, b1, 1) the normalised image-2 point and n0 = -a0, n1 = -a1 the
// negated normalised image-1 coordinates, row 2i holds b in columns 0, 3, 6 and n0 * b in columns 2, 5, 8; row 2i + 1 holds b
// in columns 1, 4, 7 and n1 * b in columns 2, 5, 8 (lin_hgN above; products with the structural zeros are left out as in
// cov_mat_hgN).  So
//   * the b x b block is the same sequence of terms for both rows: entries (0|3|6, 0|3|6) and (1|4|7, 1|4|7) are one sum,
//   * the cross blocks (2|5|8, 0|3|6) and (2|5|8, 1|4|7) are (n0 b) x b and (n1 b) x b,
//   * the (2|5|8, 2|5|8) block takes the row-2i term and then the row-(2i+1) term of every point.
// Each product is formed from the same two factors as Z[k][i] * Z[k][j] (multiplication commutes exactly), 1 * x and x * 1
// are x, and every sum adds its terms in the reference's order: the matrix is bit-identical to cov_mat(lin_hgN(...)).
typedef double cov_v4 __attribute__((vector_size(32)));
static inline void cov_hgN_fused(double *Cv, const double *u, const int *inl, int len, const double *A1, const double *A2) {
  const cov_v4 zero = {0, 0, 0, 0};
  cov_v4 BB = zero;                     // b0 b0, b1 b0, b1 b1, b0 (= 1 * b0)
  double sB1 = 0, sOne = 0;             // b1 (= 1 * b1), 1 * 1
  cov_v4 X0[3] = {zero, zero, zero};    // row 2i:     b0 * C0, b1 * C0, 1 * C0 with C0 = (n0 b0, n0 b1, n0, -)
  cov_v4 X1[3] = {zero, zero, zero};    // row 2i + 1: the same with n1
  cov_v4 S1 = zero;                     // (2,2) (5,2) (5,5) (8,2)
  double s85 = 0, s88 = 0;              // (8,5) (8,8)
  const double a1s = A1[0], a1x = A1[1], a1y = A1[2], a2s = A2[0], a2x = A2[1], a2y = A2[2];
  for (int i = 0; i < len; i++) {
    const double *s = u + 6 * inl[i];
    const double a0 = s[0] * a1s + a1x, a1 = s[1] * a1s + a1y;
    const double b0 = s[3] * a2s + a2x, b1 = s[4] * a2s + a2y;
    const double n0 = -a0, n1 = -a1;
    const cov_v4 Bv = {b0, b1, 1.0, 0.0};
    const cov_v4 C0 = (cov_v4){n0, n0, n0, 0.0} * Bv, C1 = (cov_v4){n1, n1, n1, 0.0} * Bv;   // n * 1 = n
    BB += (cov_v4){b0, b1, b1, 1.0} * (cov_v4){b0, b0, b1, b0};
    sB1 += b1; sOne += 1.0;
    const cov_v4 vb0 = {b0, b0, b0, b0}, vb1 = {b1, b1, b1, b1};
    X0[0] += vb0 * C0; X0[1] += vb1 * C0; X0[2] += C0;
    X1[0] += vb0 * C1; X1[1] += vb1 * C1; X1[2] += C1;
    // the shared block: z2 z2, z5 z2, z5 z5, z8 z2 | z8 z5, z8 z8 -- row 2i first
    const cov_v4 L0 = {C0[0], C0[1], C0[1], C0[2]}, R0 = {C0[0], C0[0], C0[1], C0[0]};
    const cov_v4 L1 = {C1[0], C1[1], C1[1], C1[2]}, R1 = {C1[0], C1[0], C1[1], C1[0]};
    S1 += L0 * R0;
    S1 += L1 * R1;
    s85 += C0[2] * C0[1]; s85 += C1[2] * C1[1];
    s88 += C0[2] * C0[2]; s88 += C1[2] * C1[2];
  }
  double acc[9][9];
  for (int i = 0; i < 9; i++) for (int j = 0; j < 9; j++) acc[i][j] = 0;
  for (int o = 0; o < 2; o++) {          // columns 0,3,6 (row 2i) and 1,4,7 (row 2i + 1)
    const int c0 = o, c1 = 3 + o, c2 = 6 + o;
    acc[c0][c0] = BB[0]; acc[c1][c0] = BB[1]; acc[c1][c1] = BB[2];
    acc[c2][c0] = BB[3]; acc[c2][c1] = sB1; acc[c2][c2] = sOne;
    const cov_v4 *X = o ? X1 : X0;
    // b_l * (n b_k): entry (max, min) of columns (2 | 5 | 8)[k] and c_l
    const int cc[3] = {2, 5, 8}, cb[3] = {c0, c1, c2};
    for (int l = 0; l < 3; l++)
      for (int k = 0; k < 3; k++) {
        const int i = cc[k] > cb[l] ? cc[k] : cb[l], j = cc[k] > cb[l] ? cb[l] : cc[k];
        acc[i][j] = X[l][k];
      }
  }
  acc[2][2] = S1[0]; acc[5][2] = S1[1]; acc[5][5] = S1[2]; acc[8][2] = S1[3]; acc[8][5] = s85; acc[8][8] = s88;
  for (int i = 0; i < 9; i++)
    for (int j = 0; j <= i; j++) { Cv[9 * i + j] = acc[i][j]; Cv[i + 9 * j] = acc[i][j]; }
}

// u2h, Htools.c:98-130
static inline void u2h(const double *u, const int *inl, int len, double *H, double *buffer) {
  if (len < 4) return;
  if (len == 4) {
    // The reference's 4-point branch (Htools.c:105-113) does NOT solve the 4-point problem: lin_hg fills its 9 x 9 buffer as
    // an 8-row column-wise matrix (72 entries, stride 8), trnm transposes it as a 9 x 9 (stride 9), the last row is zeroed
    // and the null space of THAT matrix becomes H; the nine entries lin_hg never wrote (Z2[72..80], which the transposition
    // moves into the last column) are uninitialised stack.  The computation is repeated here operation for operation with
    // those nine entries zero -- what the reference computes on a fresh stack.  The result is a homography unrelated to the
    // sample either way (its last column is then zero: H = e8, every point maps to the origin), so an inner sample of four
    // points -- a local optimisation that starts from 8 or 9 inliers -- never improves a model, in the reference as here.
    // (Until round 4 the intended null space was solved instead: a better answer, but not the reference's -- 3.7 % of small
    // pairs and 1 % of small epipolar problems then ended on another model.)
    double Z2[81], V[81];
    int nb[18];
    for (int i = 72; i < 81; ++i) Z2[i] = 0.0;   // the unwritten entries
    lin_hg(u, Z2, inl, len);                      // column c of the 8-row matrix at Z2 + 8 c
    for (int r = 0; r < 9; r++)                   // trnm(Z2, 9)
      for (int c = r + 1; c < 9; c++) { const double t = Z2[r * 9 + c]; Z2[r * 9 + c] = Z2[c * 9 + r]; Z2[c * 9 + r] = t; }
    for (int i = 72; i < 81; ++i) Z2[i] = 0.0;
    memset(V, 0, sizeof V);
    nullspace(Z2, V, 9, nb);
    memcpy(H, V, 9 * sizeof(double));
    return;
  }
  double A1[3], A2[3], V[81], ev[9];
  (void)buffer;
  normu(u, inl, len, A1, A2);
  cov_hgN_fused(V, u, inl, len, A1, A2);
  smallest_eigvec9(V, ev);
  memcpy(H, ev, 9 * sizeof(double));
  denormH(H, A1, A2);
}

// pinvJ + HDs (Sampson error), Htools.c:132-196
// `lin` (the reference's argument: the linearisation lin_hg() builds from the same u) is not read: its entries are
// re-formed from u -- identical products in identical order.
// Four points per step on 256-bit vectors (two 128-bit halves without AVX): every lane runs the scalar expression of its own
// point -- same operations, same order, IEEE add / mul / div, no contraction -- so the values are those of the scalar loop.
typedef double hds_v4 __attribute__((vector_size(32)));
typedef double hds_v8 __attribute__((vector_size(64)));
static inline hds_v4 hds_ld4(const double *p, size_t stride) { return (hds_v4){p[0], p[stride], p[2 * stride], p[3 * stride]}; }
__attribute__((target("avx512f"), always_inline)) static inline hds_v8 hds_ld8(const double *p, size_t stride) {
  return (hds_v8){p[0], p[stride], p[2 * stride], p[3 * stride], p[4 * stride], p[5 * stride], p[6 * stride], p[7 * stride]};
}
// the vector loop of HDs for W points per step: every lane runs the scalar expression of its own point.
// the linearisation of a point (lin_hg: x'_j, 0, -x x'_j / 0, x'_j, -y x'_j) is re-formed from u -- the same products,
// rounded the same way -- instead of being streamed from the 144-byte row of `lin` (the loop is memory-bound otherwise)
#define HDS_VECTOR_LOOP(V, W, LD, ZERO) \
  for (; i + W <= len; i += W) { \
    const double *uu = u + (size_t)6 * i; \
    const V u0 = LD(uu, 6), u1 = LD(uu + 1, 6), u3 = LD(uu + 3, 6), u4 = LD(uu + 4, 6), u5 = LD(uu + 5, 6); \
    const V zero = ZERO; \
    const V xs[3] = {u3, u4, u5}; \
    V r1 = zero, r2 = zero; \
    for (int j = 0; j < 3; j++) { \
      r1 += H[3 * j] * xs[j]; r1 += H[3 * j + 1] * zero; r1 += H[3 * j + 2] * (-u0 * xs[j]); \
      r2 += H[3 * j] * zero; r2 += H[3 * j + 1] * xs[j]; r2 += H[3 * j + 2] * (-u1 * xs[j]); \
    } \
    const V a = H[0] - H[2] * u0; \
    const V b = H[3] - H[5] * u0; \
    const V c = -H[8] - H[2] * u3 - H[5] * u4; \
    const V d = H[1] - H[2] * u1; \
    const V e = H[4] - H[5] * u1; \
    V pJ[8]; \
    { \
      const V a2 = a * a, b2 = b * b, c2 = c * c, d2 = d * d, e2 = e * e; \
      const V c2pd2 = c2 + d2, ab = a * b, de = d * e; \
      const V Q = c * (c2pd2 + e2); \
      pJ[0] = -b * de + a * (c2 + e2); \
      pJ[1] = b * c2pd2 - a * de; \
      pJ[2] = Q; \
      pJ[3] = -c * (a * d + b * e); \
      pJ[4] = d * (b2 + c2) - ab * e; \
      pJ[5] = -ab * d + e * (a2 + c2); \
      pJ[6] = pJ[3]; \
      pJ[7] = c * (a2 + b2 + c2); \
      const V N = a * pJ[0] + b * pJ[1] + c * pJ[2]; \
      for (int q = 0; q < 8; q++) pJ[q] /= N; \
    } \
    V acc = ZERO; \
    for (int j = 0; j < 4; j++) { \
      const V t = pJ[j] * r1 + pJ[j + 4] * r2; \
      acc += t * t; \
    } \
    for (int k = 0; k < W; k++) p[i + k] = acc[k]; \
  }
// eight points per step where the CPU has 512-bit vectors (the seven divisions per point are what the loop costs)
__attribute__((target("avx512f"))) static inline int hds_loop8(const double *u, const double *H, double *p, int len) {
  int i = 0;
  HDS_VECTOR_LOOP(hds_v8, 8, hds_ld8, ((hds_v8){0, 0, 0, 0, 0, 0, 0, 0}))
  return i;
}
static inline void HDs(const double *lin, const double *u, const double *H, double *p, int len) {
  (void)lin;
  int i = 0;
#if defined(__x86_64__)
  static const bool wide = __builtin_cpu_supports("avx512f");
  if (wide) i = hds_loop8(u, H, p, len);
#endif
  HDS_VECTOR_LOOP(hds_v4, 4, hds_ld4, ((hds_v4){0, 0, 0, 0}))
  u += (size_t)6 * i; p += i;
  for (; i < len; i++) {
    double r1 = 0, r2 = 0;
    for (int j = 0; j < 3; j++) {
      r1 += H[3 * j] * u[3 + j]; r1 += H[3 * j + 1] * 0.0; r1 += H[3 * j + 2] * (-u[0] * u[3 + j]);
      r2 += H[3 * j] * 0.0; r2 += H[3 * j + 1] * u[3 + j]; r2 += H[3 * j + 2] * (-u[1] * u[3 + j]);
    }
    double a = H[0] - H[2] * u[0];
    double b = H[3] - H[5] * u[0];
    double c = -H[8] - H[2] * u[3] - H[5] * u[4];
    double d = H[1] - H[2] * u[1];
    double e = H[4] - H[5] * u[1];
    double pJ[8];
    {
      double a2 = a * a, b2 = b * b, c2 = c * c, d2 = d * d, e2 = e * e;
      double c2pd2 = c2 + d2, ab = a * b, de = d * e;
      double Q = c * (c2pd2 + e2);
      pJ[0] = -b * de + a * (c2 + e2);
      pJ[1] = b * c2pd2 - a * de;
      pJ[2] = Q;
      pJ[3] = -c * (a * d + b * e);
      pJ[4] = d * (b2 + c2) - ab * e;
      pJ[5] = -ab * d + e * (a2 + c2);
      pJ[6] = pJ[3];
      pJ[7] = c * (a2 + b2 + c2);
      double N = a * pJ[0] + b * pJ[1] + c * pJ[2];
      for (int q = 0; q < 8; q++) pJ[q] /= N;
    }
    double acc = 0;
    for (int j = 0; j < 4; j++) {
      double t = pJ[j] * r1 + pJ[j + 4] * r2;
      acc += t * t;
    }
    *p++ = acc;
    u += 6;
  }
}

// minv for n = 3 (matutls/minv.c) is a generic LU inverse; the symmetric transfer check only needs a
// 3x3 inverse of H^T, computed here by the same LU-with-partial-pivoting scheme in closed form.
static inline bool inv3_lu(const double *M, double *Out) {
  double a[3][6];
  for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) { a[i][j] = M[i * 3 + j]; a[i][3 + j] = (i == j); }
  for (int c = 0; c < 3; c++) {
    int piv = c;
    for (int r = c + 1; r < 3; r++) if (fabs(a[r][c]) > fabs(a[piv][c])) piv = r;
    if (a[piv][c] == 0) return false;
    if (piv != c) for (int k = 0; k < 6; k++) { double t = a[c][k]; a[c][k] = a[piv][k]; a[piv][k] = t; }
    double d = a[c][c];
    for (int k = 0; k < 6; k++) a[c][k] /= d;
    for (int r = 0; r < 3; r++) {
      if (r == c) continue;
      double f = a[r][c];
      for (int k = 0; k < 6; k++) a[r][k] -= f * a[c][k];
    }
  }
  for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) Out[i * 3 + j] = a[i][3 + j];
  return true;
}

// all_Hori_valid, Htools.c:543-570
static inline int all_hori_valid(const double *us, const int *idx) {
  auto cross = [](double *o, const double *a, const double *b) {
    o[0] = a[1] * b[2] - a[2] * b[1];
    o[1] = a[2] * b[0] - a[0] * b[2];
    o[2] = a[0] * b[1] - a[1] * b[0];
  };
  const double *a = us + 6 * idx[0], *b = us + 6 * idx[1], *c = us + 6 * idx[2], *d = us + 6 * idx[3];
  double p[3], q[3];
  cross(p, a, b); cross(q, a + 3, b + 3);
  if ((p[0] * c[0] + p[1] * c[1] + p[2] * c[2]) * (q[0] * c[3] + q[1] * c[4] + q[2] * c[5]) < 0) return 0;
  if ((p[0] * d[0] + p[1] * d[1] + p[2] * d[2]) * (q[0] * d[3] + q[1] * d[4] + q[2] * d[5]) < 0) return 0;
  cross(p, c, d); cross(q, c + 3, d + 3);
  if ((p[0] * a[0] + p[1] * a[1] + p[2] * a[2]) * (q[0] * a[3] + q[1] * a[4] + q[2] * a[5]) < 0) return 0;
  if ((p[0] * b[0] + p[1] * b[1] + p[2] * b[2]) * (q[0] * b[3] + q[1] * b[4] + q[2] * b[5]) < 0) return 0;
  return 1;
}

}  // namespace mx
