// kernels_match.hip -- brute-force 128-D squared-L2 matching with the FGINN ratio walk.
//
// Reference: MatchFlannFGINN (matching/matching.cpp:357-461) over an exact (linear) kNN:
//   squared L2 in f32 (exact: descriptors hold integers 0..255, sums < 2^24), neighbours sorted
//   ascending with ties by ascending train index, walk j = 1..nn-1:
//     accept at the first j with (float)d0/(float)dj <= ratio^2,
//     give up at the first j whose position is farther than contradDist from NN0's.
// Restated as reductions over the N x M distance matrix (no top-50 sort):
//   sweep 1   top-2 of (d, t) per query: NN0, NN1
//   decide    j = 1 of the walk needs only NN0/NN1: ratio(d0, d1) passes -> ACCEPT (NNj = NN1);
//             NN1 farther than contradDist from NN0 -> REJECT; otherwise UNDECIDED
//   sweep 2   only for UNDECIDED queries: Dmin = smallest integer distance passing the ratio test against
//             d0 (the predicate is monotone); NNj = lex-min over d >= Dmin; nless = #{t != NN0 : d < Dmin};
//             nbad = #{those farther than contradDist from NN0}
//   accept  <=>  NNj exists, nbad == 0, nless <= nn-2          (rank of NNj is nless+1)
// Distances come from the int8 matrix cores: with a' = a-128, b' = b-128 (both in [-128,127])
// |a-b|^2 = |a'|^2 + |b'|^2 - 2 a'.b' exactly in int32;  a'.b' = v_mfma_i32_32x32x32_i8 over K = 128.
//
// Work decomposition: a 256-thread workgroup owns 128 queries (one 32-query A fragment set per wave,
// resident in registers for the whole sweep) and one of S contiguous ranges of train tiles (M is split so
// that small N still fills the chip).  Train tiles (32 descriptors = 4 KB) are loaded with fully coalesced
// 16-byte-per-lane reads, staged in a double-buffered LDS tile shared by the 4 waves, and read back as MFMA
// B fragments with ds_read_b128.  Each lane keeps the running top-2 of its 16 accumulator rows as packed keys.  In
// sweep 1 the query fragment is negated (a'' = 127 - a), so acc = -(a'.b') - sum(b') and
// key = ((|b'|^2 - 2 a'.b') << 8) | local tile = (acc << 9) + column constant: one v_lshl_add, one v_min and one v_med3
// per matrix element, with the accumulators in VGPRs (-amdgpu-mfma-vgpr-form) and the four MFMAs of tile q issued
// between the quarters of the update of tile q - 1.  The query norm, constant per row, is added after the sweep.
#include "engine.hpp"

namespace mx {

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

constexpr int BIG = 0x7fffffff;
constexpr unsigned UBIG = 0xffffffffu;
constexpr int TILES_PER_SPLIT_MAX = 256;   // 8 bits of local tile index in the packed key of sweep 1
constexpr int TPS = 4;                     // train tiles staged per barrier (4 x 4 KB per LDS buffer)

// norms[i] = |d_i - 128|^2; normS[i] (optional) = |d_i - 128|^2 + 2 sum(d_i - 128), the column constant of sweep 1
__device__ __forceinline__ void norms_body(const uint8_t *d, int n, int *norms, int *normS) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const v4i *p = reinterpret_cast<const v4i *>(d + (size_t)i * 128);
  int s = 0, lin = 0;
#pragma unroll
  for (int q = 0; q < 8; q++) {
    const v4i v = p[q];
#pragma unroll
    for (int w = 0; w < 4; w++) {
      const int x = v[w] ^ 0x80808080;
#pragma unroll
      for (int b = 0; b < 4; b++) { const int e = (int)(signed char)((x >> (8 * b)) & 0xff); s += e * e; lin += e; }
    }
  }
  norms[i] = s;
  if (normS) normS[i] = s + 2 * lin;
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
// median of three (folds to v_med3_u32): with m1 <= m2 the new second-smallest after seeing k is med3(m1, m2, k)
MX_D unsigned umed3(unsigned a, unsigned b, unsigned c) { return min(max(a, b), max(min(a, b), c)); }
MX_D int imed3(int a, int b, int c) { return min(max(a, b), max(min(a, b), c)); }

// A fragment: 16 bytes [32*kb + 16*hi, +16) of a descriptor, u8 -> i8 (x - 128 == x ^ 0x80)
MX_D v4i load_a(const uint8_t *base, int row, int kb, int hi) {
  v4i v = *reinterpret_cast<const v4i *>(base + (size_t)row * 128 + 32 * kb + 16 * hi);
  v[0] ^= 0x80808080; v[1] ^= 0x80808080; v[2] ^= 0x80808080; v[3] ^= 0x80808080;
  return v;
}

// the query fragment of sweep 1: a'' = 127 - a = -(a - 128) - 1 (u8 -> i8 by x ^ 0x7f), so that
// a''.b' = -(a'.b') - sum(b') and the distance key needs no negation of the accumulator
MX_D v4i load_a_neg(const uint8_t *base, int row, int kb, int hi) {
  v4i v = *reinterpret_cast<const v4i *>(base + (size_t)row * 128 + 32 * kb + 16 * hi);
  v[0] ^= 0x7f7f7f7f; v[1] ^= 0x7f7f7f7f; v[2] ^= 0x7f7f7f7f; v[3] ^= 0x7f7f7f7f;
  return v;
}

struct MatchGeom {
  int n1, n2, S, tilesPerSplit;
};

// stage one 32-descriptor tile (4 KB) into LDS: thread t copies bytes [16 t, 16 t + 16) of the tile.
// fetch_tile issues the global load early (next tile, in flight during the MFMAs); put_tile parks it in LDS.
MX_D v4i fetch_tile(const uint8_t *d2, int n2, int tile, int tid) {
  const int row = tile * 32 + (tid >> 3);
  v4i v = {(int)0x80808080, (int)0x80808080, (int)0x80808080, (int)0x80808080};  // rows past the end: a' = 0
  if (row < n2) v = *reinterpret_cast<const v4i *>(d2 + (size_t)tile * 4096 + (size_t)tid * 16);
  return v;
}
MX_D void put_tile(v4i v, unsigned char *lds, int tid) {
  v[0] ^= 0x80808080; v[1] ^= 0x80808080; v[2] ^= 0x80808080; v[3] ^= 0x80808080;
  // LDS image: row r at r*128, XOR-swizzled in 16-byte slots so the b128 fragment reads spread over banks
  const int r = tid >> 3, slot = tid & 7;
  *reinterpret_cast<v4i *>(lds + r * 128 + ((slot ^ (r & 7)) << 4)) = v;
}
MX_D void stage_tile(const uint8_t *d2, int n2, int tile, unsigned char *lds, int tid) {
  put_tile(fetch_tile(d2, n2, tile, tid), lds, tid);
}
MX_D v4i read_b(const unsigned char *lds, int col, int kb, int hi) {
  const int slot = 2 * kb + hi;
  return *reinterpret_cast<const v4i *>(lds + col * 128 + ((slot ^ (col & 7)) << 4));
}

// ---------------- sweep 1: per (query, split) top-2 -------------------------------------------------------
__device__ __forceinline__ void sweep1_body(const uint8_t *d1, const int *norm1, const uint8_t *d2,
                                                      const int *normS2, MatchGeom g, int4 *partial) {
  __shared__ __attribute__((aligned(16))) unsigned char tileBuf[2][TPS * 4096];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int col = lane & 31, hi = lane >> 5;
  const int qb = blockIdx.x, sp = blockIdx.y;
  const int q0 = qb * 128 + wave * 32;
  const int qrow = min(q0 + col, g.n1 - 1);
  v4i a[4];
#pragma unroll
  for (int kb = 0; kb < 4; kb++) a[kb] = load_a_neg(d1, qrow, kb, hi);
  // Per row the query norm na is a constant, so the running top-2 is kept on
  //   key = ((nb - 2 a'.b') << 8) | tile = (acc << 9) + (((nb + 2 sum b') << 8) | tile)   with acc = a''.b',
  // i.e. ONE v_lshl_add, one signed min and one signed med3 per matrix element; |nb - 2 a'.b'| < 2^23, so the key
  // fits an int32 with 8 tile bits (tilesPerSplit <= 256).
  const int ntilesAll = (g.n2 + 31) >> 5;
  const int tBeg = sp * g.tilesPerSplit, tEnd = min(tBeg + g.tilesPerSplit, ntilesAll);
  int m1[16], m2[16];
#pragma unroll
  for (int r = 0; r < 16; r++) { m1[r] = BIG; m2[r] = BIG; }
  // tiles past the end of the split are staged as zeros (b' = 0 => acc = 0) and get the column constant BIG, so
  // their keys are BIG and the tile loop needs no branches
  const int NOTILE = 0x3fffffff >> 5;   // fetch_tile sees a row index >= n2 and returns zeros
  for (int q = 0; q < TPS; q++) stage_tile(d2, g.n2, tBeg + q < tEnd ? tBeg + q : NOTILE, tileBuf[0] + q * 4096, tid);
  __syncthreads();
  for (int tg = tBeg; tg < tEnd; tg += TPS) {
    const int cur = ((tg - tBeg) / TPS) & 1;
    v4i nxt[TPS];
#pragma unroll
    for (int q = 0; q < TPS; q++) nxt[q] = fetch_tile(d2, g.n2, tg + TPS + q < tEnd ? tg + TPS + q : NOTILE, tid);
    int tileC[TPS];
#pragma unroll
    for (int q = 0; q < TPS; q++) {
      const int t = tg + q, trow = t * 32 + col;
      tileC[q] = (t < tEnd && trow < g.n2) ? ((normS2[trow] << 8) | (t - tBeg)) : BIG;
    }
    // software pipeline inside a wave: the four MFMAs of tile q are issued between the four quarters of the top-2
    // update of tile q - 1, so the VALU work runs in the shadow of the matrix pipe
    v16i accP = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    {
      const unsigned char *tb = tileBuf[cur];
#pragma unroll
      for (int kb = 0; kb < 4; kb++) accP = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[kb], read_b(tb, col, kb, hi), accP, 0, 0, 0);
    }
#pragma unroll
    for (int q = 1; q < TPS; q++) {
      const unsigned char *tb = tileBuf[cur] + q * 4096;
      v4i bf[4];
#pragma unroll
      for (int kb = 0; kb < 4; kb++) bf[kb] = read_b(tb, col, kb, hi);
      v16i accN = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
      for (int kb = 0; kb < 4; kb++) {
        accN = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[kb], bf[kb], accN, 0, 0, 0);
#pragma unroll
        for (int r = 4 * kb; r < 4 * kb + 4; r++) {
          const int key = (accP[r] << 9) + tileC[q - 1];
          m2[r] = imed3(m1[r], m2[r], key);
          m1[r] = min(m1[r], key);
        }
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   // one MFMA ...
        __builtin_amdgcn_sched_group_barrier(0x002, 12, 0);  // ... then the 12 VALU ops of a quarter update
      }
      accP = accN;
    }
#pragma unroll
    for (int r = 0; r < 16; r++) {
      const int key = (accP[r] << 9) + tileC[TPS - 1];
      m2[r] = imed3(m1[r], m2[r], key);
      m1[r] = min(m1[r], key);
    }
#pragma unroll
    for (int q = 0; q < TPS; q++) put_tile(nxt[q], tileBuf[cur ^ 1] + q * 4096, tid);
    __syncthreads();
  }
  // unpack and merge the per-lane top-2 over the 32 lanes that hold the same rows
#pragma unroll
  for (int r = 0; r < 16; r++) {
    const int na = norm1[min(q0 + (r & 3) + 8 * (r >> 2) + 4 * hi, g.n1 - 1)];
    int d0 = m1[r] == BIG ? BIG : (m1[r] >> 8) + na, i0 = m1[r] == BIG ? BIG : (tBeg + (m1[r] & 255)) * 32 + col;
    int dd1 = m2[r] == BIG ? BIG : (m2[r] >> 8) + na, i1 = m2[r] == BIG ? BIG : (tBeg + (m2[r] & 255)) * 32 + col;
#pragma unroll
    for (int m = 16; m >= 1; m >>= 1) {
      const int od0 = __shfl_xor(d0, m), oi0 = __shfl_xor(i0, m), od1 = __shfl_xor(dd1, m), oi1 = __shfl_xor(i1, m);
      // merge two sorted pairs (d0,i0)<=(dd1,i1) and (od0,oi0)<=(od1,oi1)
      if (lex_less(od0, oi0, d0, i0)) {
        // other's best wins; second = min(mine best, other's second)
        if (lex_less(od1, oi1, d0, i0)) { dd1 = od1; i1 = oi1; } else { dd1 = d0; i1 = i0; }
        d0 = od0; i0 = oi0;
      } else {
        if (lex_less(od0, oi0, dd1, i1)) { dd1 = od0; i1 = oi0; }
      }
    }
    if (col == 0) {
      const int q = q0 + (r & 3) + 8 * (r >> 2) + 4 * hi;
      if (q < g.n1) partial[(size_t)q * g.S + sp] = make_int4(d0, i0, dd1, i1);
    }
  }
}

// ---------------- decide: merge splits, j = 1 of the walk, compact the undecided queries --------------------
__device__ __forceinline__ void decide_body(const int4 *partial, MatchGeom g, const double *pos2,
                                                      double sqminratio, double contrDistSq, MatchRow *rows, int *dmin,
                                                      int *undecided, int *nUndecided) {
  const int q = blockIdx.x * 256 + threadIdx.x;
  if (q >= g.n1) return;
  int d0 = BIG, i0 = BIG, d1 = BIG, i1 = BIG;
  for (int s = 0; s < g.S; s++) {
    const int4 p = partial[(size_t)q * g.S + s];
    if (lex_less(p.x, p.y, d0, i0)) {
      if (lex_less(p.z, p.w, d0, i0)) { d1 = p.z; i1 = p.w; } else { d1 = d0; i1 = i0; }
      d0 = p.x; i0 = p.y;
    } else if (lex_less(p.x, p.y, d1, i1)) { d1 = p.x; i1 = p.y; }
  }
  MatchRow o;
  o.t0 = i0 == BIG ? -1 : i0; o.t1 = i1 == BIG ? -1 : i1; o.tj = -1; o.nless = 0; o.nbad = 0;
  o.d0 = (float)d0; o.d1 = (float)d1; o.dj = 0.f;
  int dm = 0;
  if (i0 != BIG && i1 != BIG) {
    if (ratio_pass((float)d0, (float)d1, sqminratio)) { o.tj = i1; o.dj = (float)d1; }       // accepted at j = 1
    else {
      const double dx = pos2[2 * i0] - pos2[2 * i1], dy = pos2[2 * i0 + 1] - pos2[2 * i1 + 1];
      if (dx * dx + dy * dy > contrDistSq) o.nbad = 1;                                      // first contradictive
      else {
        dm = ratio_dmin(d0, sqminratio);
        const int slot = atomicAdd(nUndecided, 1);
        undecided[slot] = q;
        o.nless = -1;   // filled by sweep 2
      }
    }
  }
  rows[q] = o;
  dmin[q] = dm;
}

// ---------------- sweep 2 over the undecided queries ----------------------------------------------------------
__device__ __forceinline__ void sweep2_body(const uint8_t *d1, const int *norm1, const uint8_t *d2,
                                                      const int *norm2, MatchGeom g, const double *pos2,
                                                      double contrDistSq, const MatchRow *rows, const int *dmin,
                                                      const int *undecided, const int *nUndecided, int4 *partial2) {
  __shared__ __attribute__((aligned(16))) unsigned char tileBuf[2][TPS * 4096];
  const int nU = *nUndecided;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int col = lane & 31, hi = lane >> 5;
  const int qb = blockIdx.x, sp = blockIdx.y;
  if (qb * 128 >= nU) return;
  const int u0 = qb * 128 + wave * 32;
  const int qA = undecided[min(u0 + col, nU - 1)];
  v4i a[4];
#pragma unroll
  for (int kb = 0; kb < 4; kb++) a[kb] = load_a(d1, qA, kb, hi);
  int na[16], dm[16], t0[16];
#pragma unroll
  for (int r = 0; r < 16; r++) {
    const int q = undecided[min(u0 + (r & 3) + 8 * (r >> 2) + 4 * hi, nU - 1)];
    na[r] = norm1[q]; dm[r] = dmin[q]; t0[r] = rows[q].t0;
  }
  const int ntilesAll = (g.n2 + 31) >> 5;
  const int tBeg = sp * g.tilesPerSplit, tEnd = min(tBeg + g.tilesPerSplit, ntilesAll);
  unsigned mj[16];
  int nless[16], nbad[16];
#pragma unroll
  for (int r = 0; r < 16; r++) { mj[r] = UBIG; nless[r] = 0; nbad[r] = 0; }
  for (int q = 0; q < TPS; q++) if (tBeg + q < tEnd) stage_tile(d2, g.n2, tBeg + q, tileBuf[0] + q * 4096, tid);
  __syncthreads();
  for (int tg = tBeg; tg < tEnd; tg += TPS) {
    const int cur = ((tg - tBeg) / TPS) & 1;
    v4i nxt[TPS];
#pragma unroll
    for (int q = 0; q < TPS; q++) {
      nxt[q] = (v4i){0, 0, 0, 0};
      if (tg + TPS + q < tEnd) nxt[q] = fetch_tile(d2, g.n2, tg + TPS + q, tid);
    }
#pragma unroll
    for (int q = 0; q < TPS; q++) {
      const int t = tg + q;
      if (t >= tEnd) break;
      const unsigned char *tb = tileBuf[cur] + q * 4096;
      v16i acc = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
      for (int kb = 0; kb < 4; kb++) acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[kb], read_b(tb, col, kb, hi), acc, 0, 0, 0);
      const int trow = t * 32 + col;
      if (trow < g.n2) {
        const int nb = norm2[trow];
        const unsigned lt = (unsigned)(t - tBeg);
#pragma unroll
        for (int r = 0; r < 16; r++) {
          const int d = na[r] + nb - 2 * acc[r];
          if (d >= dm[r]) mj[r] = min(mj[r], ((unsigned)d << 9) | lt);
          else if (trow != t0[r]) {
            nless[r]++;
            // rare path: geometric consistency with NN0 (distanceSq, matching.cpp:174-179), f64
            const double dx = pos2[2 * t0[r]] - pos2[2 * trow], dy = pos2[2 * t0[r] + 1] - pos2[2 * trow + 1];
            if (dx * dx + dy * dy > contrDistSq) nbad[r]++;
          }
        }
      }
    }
#pragma unroll
    for (int q = 0; q < TPS; q++) if (tg + TPS + q < tEnd) put_tile(nxt[q], tileBuf[cur ^ 1] + q * 4096, tid);
    __syncthreads();
  }
#pragma unroll
  for (int r = 0; r < 16; r++) {
    int dj = mj[r] == UBIG ? BIG : (int)(mj[r] >> 9), ij = mj[r] == UBIG ? BIG : (tBeg + (int)(mj[r] & 511)) * 32 + col;
#pragma unroll
    for (int m = 16; m >= 1; m >>= 1) {
      const int od = __shfl_xor(dj, m), oi = __shfl_xor(ij, m);
      if (lex_less(od, oi, dj, ij)) { dj = od; ij = oi; }
      nless[r] += __shfl_xor(nless[r], m);
      nbad[r] += __shfl_xor(nbad[r], m);
    }
    if (col == 0) {
      const int u = u0 + (r & 3) + 8 * (r >> 2) + 4 * hi;
      if (u < nU) partial2[(size_t)u * g.S + sp] = make_int4(dj, ij, nless[r], nbad[r]);
    }
  }
}

__device__ __forceinline__ void finish_body(const int4 *partial2, MatchGeom g, const int *undecided,
                                                      const int *nUndecided, MatchRow *rows) {
  const int u = blockIdx.x * 256 + threadIdx.x;
  if (u >= *nUndecided) return;
  int dj = BIG, ij = BIG, nless = 0, nbad = 0;
  for (int s = 0; s < g.S; s++) {
    const int4 p = partial2[(size_t)u * g.S + s];
    if (lex_less(p.x, p.y, dj, ij)) { dj = p.x; ij = p.y; }
    nless += p.z; nbad += p.w;
  }
  const int q = undecided[u];
  MatchRow o = rows[q];
  o.tj = ij == BIG ? -1 : ij;
  o.dj = (float)dj;
  o.nless = nless; o.nbad = nbad;
  rows[q] = o;
}

size_t match_workspace_bytes(int n1, int n2, int *S_out, int *tilesPerSplit_out) {
  const int nQB = (n1 + 127) / 128;
  const int ntiles = (n2 + 31) / 32;
  int S = (768 + nQB - 1) / nQB;                 // aim at >= 3 workgroups per CU
  if (S > ntiles / 4) S = ntiles / 4;           // at least 4 tiles per split
  if (S < 1) S = 1;
  int tps = (ntiles + S - 1) / S;
  if (tps > TILES_PER_SPLIT_MAX) { tps = TILES_PER_SPLIT_MAX; }
  S = (ntiles + tps - 1) / tps;
  if (S < 1) S = 1;
  *S_out = S; *tilesPerSplit_out = tps;
  size_t bytes = 0;
  bytes += (size_t)(n1 + 2 * (size_t)n2) * 4 + 768;  // norms (+ the sweep-1 column constants of the trains)
  bytes += (size_t)n1 * S * 16 * 2 + 256;        // partial, partial2
  bytes += (size_t)n1 * 4 * 2 + 256;             // dmin, undecided
  bytes += 256;                                  // counter
  return bytes;
}

// ---- batched entry points: blockIdx.z selects one of up to MATCH_MAXB independent problems (the pairs of a launch set).
// At 2-3 k descriptors per image a problem is six launches of ~10-20 us of mostly latency, so the problems of a batch
// share the launches.
struct MatchProblem {
  const uint8_t *d1, *d2;
  const double *pos2;
  int *norm1, *norm2, *normS2, *dmin, *undecided, *counter;
  int4 *partial, *partial2;
  MatchRow *rows;
  MatchGeom g;
};
struct MatchBatch { MatchProblem p[MATCH_MAXB]; };

__global__ __launch_bounds__(256) void k_desc_norms(MatchBatch b) {
  const MatchProblem &P = b.p[blockIdx.z];
  if (blockIdx.y == 0) norms_body(P.d1, P.g.n1, P.norm1, (int *)nullptr);
  else norms_body(P.d2, P.g.n2, P.norm2, P.normS2);
}
__global__ __launch_bounds__(256) void k_match_sweep1(MatchBatch b) {
  const MatchProblem &P = b.p[blockIdx.z];
  if ((int)blockIdx.x * 128 >= P.g.n1 || (int)blockIdx.y >= P.g.S) return;
  sweep1_body(P.d1, P.norm1, P.d2, P.normS2, P.g, P.partial);
}
__global__ __launch_bounds__(256) void k_match_decide(MatchBatch b, double sqminratio, double contrDistSq) {
  const MatchProblem &P = b.p[blockIdx.z];
  decide_body(P.partial, P.g, P.pos2, sqminratio, contrDistSq, P.rows, P.dmin, P.undecided, P.counter);
}
__global__ __launch_bounds__(256) void k_match_sweep2(MatchBatch b, double contrDistSq) {
  const MatchProblem &P = b.p[blockIdx.z];
  if ((int)blockIdx.x * 128 >= P.g.n1 || (int)blockIdx.y >= P.g.S) return;
  sweep2_body(P.d1, P.norm1, P.d2, P.norm2, P.g, P.pos2, contrDistSq, P.rows, P.dmin, P.undecided, P.counter, P.partial2);
}
__global__ __launch_bounds__(256) void k_match_finish(MatchBatch b) {
  const MatchProblem &P = b.p[blockIdx.z];
  finish_body(P.partial2, P.g, P.undecided, P.counter, P.rows);
}

// Problems with n1 == 0 or n2 == 0 must be left out by the caller.  workspace[i] holds match_workspace_bytes(n1[i], n2[i]).
void launch_match_batch(hipStream_t s, int nb, const uint8_t *const *d1, const int *n1, const uint8_t *const *d2, const int *n2,
                        const double *const *pos2, double sqminratio, double contrDistSq, MatchRow *const *rows,
                        void *const *workspace) {
  if (nb <= 0) return;
  MatchBatch b;
  memset(&b, 0, sizeof b);
  int maxN = 0, maxN1 = 0, maxS = 0;
  for (int i = 0; i < nb; i++) {
    MatchProblem &P = b.p[i];
    P.g.n1 = n1[i]; P.g.n2 = n2[i];
    match_workspace_bytes(n1[i], n2[i], &P.g.S, &P.g.tilesPerSplit);
    char *w = (char *)workspace[i];
    auto take = [&](size_t bytes) { char *p = w; w += (bytes + 255) & ~(size_t)255; return p; };
    P.norm1 = (int *)take((size_t)n1[i] * 4); P.norm2 = (int *)take((size_t)n2[i] * 4); P.normS2 = (int *)take((size_t)n2[i] * 4);
    P.partial = (int4 *)take((size_t)n1[i] * P.g.S * 16); P.partial2 = (int4 *)take((size_t)n1[i] * P.g.S * 16);
    P.dmin = (int *)take((size_t)n1[i] * 4); P.undecided = (int *)take((size_t)n1[i] * 4);
    P.counter = (int *)take(64);
    P.d1 = d1[i]; P.d2 = d2[i]; P.pos2 = pos2[i]; P.rows = rows[i];
    hipMemsetAsync(P.counter, 0, 4, s);
    maxN = std::max(maxN, std::max(n1[i], n2[i])); maxN1 = std::max(maxN1, n1[i]); maxS = std::max(maxS, P.g.S);
  }
  hipLaunchKernelGGL(k_desc_norms, dim3((maxN + 255) / 256, 2, nb), dim3(256), 0, s, b);
  const dim3 grid((maxN1 + 127) / 128, maxS, nb), gridQ((maxN1 + 255) / 256, 1, nb);
  hipLaunchKernelGGL(k_match_sweep1, grid, dim3(256), 0, s, b);
  hipLaunchKernelGGL(k_match_decide, gridQ, dim3(256), 0, s, b, sqminratio, contrDistSq);
  hipLaunchKernelGGL(k_match_sweep2, grid, dim3(256), 0, s, b, contrDistSq);
  hipLaunchKernelGGL(k_match_finish, gridQ, dim3(256), 0, s, b);
}

void launch_match(hipStream_t s, const uint8_t *d1, int n1, const uint8_t *d2, int n2, const double *pos2,
                  double sqminratio, double contrDistSq, MatchRow *rows, void *workspace) {
  if (n1 <= 0 || n2 <= 0) return;
  launch_match_batch(s, 1, &d1, &n1, &d2, &n2, &pos2, sqminratio, contrDistSq, &rows, &workspace);
}

}  // namespace mx
