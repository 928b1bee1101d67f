// kernels_match.hip -- brute-force 128-D squared-L2 matching with the FGINN ratio walk.
//
// Reference: MatchFlannFGINN (matching/matching.cpp:357-461) over an exact (linear) kNN:
//   squared L2 in f32 (exact: descriptors hold integers 0..255, sums < 2^24), neighbours sorted
//   ascending with ties by ascending train index, walk j = 1..nn-1:
//     accept at the first j with (float)d0/(float)dj <= ratio^2,
//     give up at the first j whose position is farther than contradDist from NN0's.
// The walk is restated as reductions over the N x M distance matrix (no top-50 sort):
//   sweep 1:  NN0 = lexicographic min (d, t)
//   Dmin(q)  = smallest integer distance that passes the ratio test against d0 (monotone)
//   sweep 2:  NN1 = lex-min over t != NN0;  NNj = lex-min over d >= Dmin;
//             nless = #{t != NN0 : d < Dmin};  nbad = #{those farther than contradDist from NN0}
//   accept  <=>  NNj exists, nbad == 0, nless <= nn-2      (rank of NNj is nless+1)
// Distances come from the int8 matrix cores: with a' = a-128, b' = b-128 (both in [-128,127])
// |a-b|^2 = |a'|^2 + |b'|^2 - 2 a'.b' exactly in int32;  a'.b' = v_mfma_i32_32x32x32_i8 over K = 128.
// One wavefront owns 32 queries and streams all trains in 32-wide tiles; lane l holds train column
// l&31 and query rows (reg&3)+8*(reg>>2)+4*(l>>5) of each 32x32 tile.
#include "engine.hpp"

namespace mx {

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

MX_D int sumsq_i8x16(v4i v) {
  int s = 0;
#pragma unroll
  for (int w = 0; w < 4; w++) {
    const int x = v[w];
#pragma unroll
    for (int b = 0; b < 4; b++) {
      const int e = (int)(signed char)((x >> (8 * b)) & 0xff);
      s += e * e;
    }
  }
  return s;
}

MX_D v4i load_frag(const uint8_t *base, int row, int nrows, int kb, int hi) {
  // 16 bytes [32*kb + 16*hi, +16) of descriptor `row`, converted u8 -> i8 by subtracting 128
  v4i v = {0, 0, 0, 0};
  if (row < nrows) {
    v = *reinterpret_cast<const v4i *>(base + (size_t)row * 128 + 32 * kb + 16 * hi);
    v[0] ^= 0x80808080; v[1] ^= 0x80808080; v[2] ^= 0x80808080; v[3] ^= 0x80808080;
  }
  return v;
}

MX_D bool ratio_pass(float d0, float d, double sqminratio) {
  const float r = d0 / d;            // f32 division as in `double ratio = distsRow[0]/distsRow[j]`
  return (double)r <= sqminratio;    // NaN (0/0) fails
}

// smallest integer D > d0 with ratio_pass(d0, D); the predicate is monotone in D
MX_D int ratio_dmin(int d0i, double sqminratio) {
  const float d0 = (float)d0i;
  double est = (double)d0i / sqminratio;
  int D = est > 2.0e9 ? 2000000000 : (int)est;
  if (D <= d0i) D = d0i + 1;
  while (D > d0i + 1 && ratio_pass(d0, (float)(D - 1), sqminratio)) D--;
  while (D < 2000000000 && !ratio_pass(d0, (float)D, sqminratio)) D++;
  return D;
}

MX_D bool lex_less(int da, int ia, int db, int ib) { return da < db || (da == db && ia < ib); }

constexpr int BIG = 0x7fffffff;

__global__ __launch_bounds__(256) void k_match_fginn(const uint8_t *d1, int n1, const uint8_t *d2, int n2,
                                                     const double *pos2, double sqminratio, double contrDistSq,
                                                     MatchRow *rows) {
  const int wave = (blockIdx.x * 256 + threadIdx.x) >> 6;
  const int lane = threadIdx.x & 63;
  const int q0 = wave * 32;
  if (q0 >= n1) return;
  const int col = lane & 31, hi = lane >> 5;
  // A fragments: query row q0 + (lane&31), k-blocks 0..3
  v4i a[4];
  int na_part = 0;
#pragma unroll
  for (int kb = 0; kb < 4; kb++) { a[kb] = load_frag(d1, q0 + col, n1, kb, hi); na_part += sumsq_i8x16(a[kb]); }
  const int na_row = na_part + __shfl_xor(na_part, 32);  // |a'|^2 of query q0 + (lane&31)
  int na[16];
#pragma unroll
  for (int r = 0; r < 16; r++) na[r] = __shfl(na_row, (r & 3) + 8 * (r >> 2) + 4 * hi);

  const int ntiles = (n2 + 31) >> 5;
  // ---------------- sweep 1: nearest neighbour ----------------
  int bd[16], bi[16];
#pragma unroll
  for (int r = 0; r < 16; r++) { bd[r] = BIG; bi[r] = BIG; }
  for (int t = 0; t < ntiles; t++) {
    const int trow = t * 32 + col;
    v16i acc = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    int nb_part = 0;
#pragma unroll
    for (int kb = 0; kb < 4; kb++) {
      const v4i b = load_frag(d2, trow, n2, kb, hi);
      nb_part += sumsq_i8x16(b);
      acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[kb], b, acc, 0, 0, 0);
    }
    const int nb = nb_part + __shfl_xor(nb_part, 32);
    if (trow < n2) {
#pragma unroll
      for (int r = 0; r < 16; r++) {
        const int d = na[r] + nb - 2 * acc[r];
        if (d < bd[r]) { bd[r] = d; bi[r] = trow; }  // tiles ascend, so a tie keeps the lower index
      }
    }
  }
  // reduce over the 32 lanes that share the same rows (same hi): lexicographic min
#pragma unroll
  for (int r = 0; r < 16; r++) {
#pragma unroll
    for (int m = 16; m >= 1; m >>= 1) {
      const int od = __shfl_xor(bd[r], m), oi = __shfl_xor(bi[r], m);
      if (lex_less(od, oi, bd[r], bi[r])) { bd[r] = od; bi[r] = oi; }
    }
  }
  // ---------------- per-row threshold and NN0 position ----------------
  int dmin[16];
#pragma unroll
  for (int r = 0; r < 16; r++) dmin[r] = (bi[r] == BIG) ? BIG : ratio_dmin(bd[r], sqminratio);
  // ---------------- sweep 2 ----------------
  int d1v[16], i1v[16], djv[16], ijv[16], nless[16], nbad[16];
#pragma unroll
  for (int r = 0; r < 16; r++) { d1v[r] = BIG; i1v[r] = BIG; djv[r] = BIG; ijv[r] = BIG; nless[r] = 0; nbad[r] = 0; }
  for (int t = 0; t < ntiles; t++) {
    const int trow = t * 32 + col;
    v16i acc = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    int nb_part = 0;
#pragma unroll
    for (int kb = 0; kb < 4; kb++) {
      const v4i b = load_frag(d2, trow, n2, kb, hi);
      nb_part += sumsq_i8x16(b);
      acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[kb], b, acc, 0, 0, 0);
    }
    const int nb = nb_part + __shfl_xor(nb_part, 32);
    if (trow < n2) {
#pragma unroll
      for (int r = 0; r < 16; r++) {
        const int d = na[r] + nb - 2 * acc[r];
        if (trow != bi[r]) {
          if (d < d1v[r]) { d1v[r] = d; i1v[r] = trow; }
          if (d < dmin[r]) {
            nless[r]++;
            // rare path: geometric consistency with NN0 (distanceSq, matching.cpp:174-179), f64
            const double dx = pos2[2 * bi[r]] - pos2[2 * trow], dy = pos2[2 * bi[r] + 1] - pos2[2 * trow + 1];
            if (dx * dx + dy * dy > contrDistSq) nbad[r]++;
          } else if (d < djv[r]) { djv[r] = d; ijv[r] = trow; }
        }
      }
    }
  }
#pragma unroll
  for (int r = 0; r < 16; r++) {
#pragma unroll
    for (int m = 16; m >= 1; m >>= 1) {
      int od = __shfl_xor(d1v[r], m), oi = __shfl_xor(i1v[r], m);
      if (lex_less(od, oi, d1v[r], i1v[r])) { d1v[r] = od; i1v[r] = oi; }
      od = __shfl_xor(djv[r], m); oi = __shfl_xor(ijv[r], m);
      if (lex_less(od, oi, djv[r], ijv[r])) { djv[r] = od; ijv[r] = oi; }
      nless[r] += __shfl_xor(nless[r], m);
      nbad[r] += __shfl_xor(nbad[r], m);
    }
  }
  if (col == 0) {
#pragma unroll
    for (int r = 0; r < 16; r++) {
      const int q = q0 + (r & 3) + 8 * (r >> 2) + 4 * hi;
      if (q < n1) {
        MatchRow o;
        o.t0 = bi[r] == BIG ? -1 : bi[r];
        o.t1 = i1v[r] == BIG ? -1 : i1v[r];
        o.tj = ijv[r] == BIG ? -1 : ijv[r];
        o.nless = nless[r]; o.nbad = nbad[r];
        o.d0 = (float)bd[r]; o.d1 = (float)d1v[r]; o.dj = (float)djv[r];
        rows[q] = o;
      }
    }
  }
}

void launch_match(hipStream_t s, const uint8_t *d1, int n1, const uint8_t *d2, int n2, const double *pos2,
                  double sqminratio, double contrDistSq, MatchRow *rows) {
  if (n1 <= 0 || n2 <= 0) return;
  const int waves = (n1 + 31) / 32;
  const int blocks = (waves + 3) / 4;
  hipLaunchKernelGGL(k_match_fginn, dim3(blocks), dim3(256), 0, s, d1, n1, d2, n2, pos2, sqminratio, contrDistSq, rows);
}

}  // namespace mx
