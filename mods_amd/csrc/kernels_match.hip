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
// Distances come from the int8 matrix cores: with a'' = 127 - a, b' = b - 128 (both in [-128,127])
//   |a-b|^2 - |a-128|^2 = (|b'|^2 + 2 sum b') + 2 a''.b'   exactly in int32;   a''.b' = v_mfma_i32_32x32x32_i8 over K = 128.
//
// Work decomposition (gfx950: a plain VALU instruction costs 4 cycles per wavefront, a 32x32x32 int8 MFMA 32, so the
// epilogue, not the matrix pipe, is what has to be made small):
//  * TRAIN descriptors are the MFMA rows (A operand, streamed), QUERIES the columns (B operand, resident in VGPRs):
//    a lane then owns ONE query per 32-query set and sees 16 trains (a "group": fixed tile, fixed lane half) per tile,
//    so its running state is two keys, not 16 x 2, and the merge at the end is one lane exchange.
//  * Per group the lane forms 16 keys (d - |a'|^2) << 8 | idx with one v_lshl_add each, reduces them with a v_min3 tree
//    (8 ops) and feeds ONLY the group minimum to the running top-2 (v_med3 + v_min): 26 VALU per 4 MFMA instead of 48.
//    The top-2 of group minima misses exactly one candidate -- the second-best INSIDE the group of the overall winner;
//    k_match_decide recomputes the 15 other distances of NN0's group (dot4) and folds that candidate in.
//  * idx = (tile in a 12-tile chunk + 1) << 4 | register: every 12 tiles the lane moves the indices of keys that
//    changed into two index registers and clears the low byte, so older entries keep winning ties (ascending train
//    index, as the reference's sort) and a split may be any number of tiles long.
//  * k_match_pack rewrites the trains once as 4 KB tiles (b - 128, 16-byte slots XOR-swizzled so that the ds_read_b128
//    fragment reads are conflict-free) plus the 32 per-train key constants of each tile; the sweeps then stage 4 tiles per
//    barrier with direct global->LDS loads (no staging registers), double-buffered; a wave holds 2 x 32 queries, so every
//    fragment read from LDS feeds two MFMA chains.
//  * sweep 2 is the same loop with another two-instruction update: d < Dmin <=> key < (Dmin - |a'|^2) << 8, so a group
//    without a sub-threshold train feeds its minimum to NNj, and a group WITH one is logged as a 4-byte event in the
//    lane's own slots (no atomics); k_match_events recomputes the 16 distances of every event group exactly for
//    nless / nbad / NNj (and falls back to an exact scan of all trains when a lane ran out of slots).
//  * inside a wave the four MFMAs of one (tile, query set) chain are issued between the quarters of the reduction of the
//    previous chain, so the matrix pipe and the vector ALU overlap without relying on other waves being out of phase.
#include <type_traits>
#include "engine.hpp"

namespace mx {

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

constexpr int BIG = 0x7fffffff;
constexpr int NONE = 0x7fffff00;           // empty slot of a running minimum: larger than every real key, low byte 0
constexpr int TPS = 4;                     // train tiles staged per barrier
constexpr int CHUNK = 12;                  // tiles per index chunk (3 stages); tilesPerSplit is a multiple of it
// 32-query sets per wave (QS, even): 2 for most problems -- 3 wavefronts per SIMD --, 4 when both sides hold >= 40 k descriptors:
// every LDS fragment read then feeds four MFMA chains (half the LDS bytes per matrix instruction) at 2 wavefronts per SIMD
// (round 3: 0.392 against 0.402 ms at 46 k x 45 k, 0.155 against 0.151 ms at 24 k x 24 k, -25 % at 10 k; selected per problem)
constexpr int sweep_wps(int qs) { return qs >= 4 ? 2 : 3; }   // waves per SIMD the sweeps are built for
constexpr int qpb_of(int qs) { return 4 * 32 * qs; }          // queries per 256-thread workgroup
constexpr int QPB_MAX = qpb_of(4), NW_MAX = 256 * 3;
static int match_qsets(int nb, int n1, int n2) {
  static const int forced = getenv("MODSX_MATCH_QSETS") ? atoi(getenv("MODSX_MATCH_QSETS")) : 0;
  if (forced == 2 || forced == 4) return forced;
  return (nb == 1 && n1 >= 40000 && n2 >= 40000) ? 4 : 2;
}
constexpr int TILE_B = 4096, STAGE_B = TPS * TILE_B + TPS * 128;
constexpr int MAXD = 128 * 255 * 255;      // largest possible squared distance

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
MX_D int imed3(int a, int b, int c) { return min(max(a, b), max(min(a, b), c)); }
MX_D int imin3(int a, int b, int c) { return min(min(a, b), c); }

struct MatchGeom {
  int n1, n2, S, tilesPerSplit, qs;
};

// Sweep 2 runs over the UNDECIDED queries only, whose number the host does not know at launch time.  With sweep 1's splits it
// would be a handful of query blocks x S long splits -- a sixth of the machine busy for as long as a whole sweep 1 workgroup
// takes (41 us of the 165 at 24 k x 24 k, where 15 % of the queries are undecided).  Every workgroup therefore derives the
// split geometry from the device-side count: the NW workgroups of the launch are dealt out as (query block, split) with as
// many splits as fill the machine once.  k_match_events uses the same function.
struct Sweep2Geom { int nQB, S, tilesPerSplit; };
MX_HD Sweep2Geom sweep2_geom(int nUnd, int n2, int qs) {
  Sweep2Geom G;
  const int ntiles = (n2 + 31) >> 5;
  const int QPB = qpb_of(qs), SWEEP2_NW = 256 * sweep_wps(qs);    // one round of workgroups
  G.nQB = (nUnd + QPB - 1) / QPB;
  int S = G.nQB > 0 ? SWEEP2_NW / G.nQB : 1;
  if (S > ntiles / CHUNK) S = ntiles / CHUNK;     // at least one index chunk per split
  if (S < 1) S = 1;
  int tps = (ntiles + S - 1) / S;
  tps = ((tps + CHUNK - 1) / CHUNK) * CHUNK;
  S = (ntiles + tps - 1) / tps;
  G.S = S < 1 ? 1 : S;
  G.tilesPerSplit = tps;
  return G;
}
// entries of the per-(undecided query, split) arrays of sweep 2, whatever the count turns out to be
static size_t sweep2_entries(int n1, int n2) {
  const int ntiles = (n2 + 31) >> 5;
  const size_t smax = (size_t)std::max(1, ntiles / CHUNK);
  const size_t a = std::max<size_t>((size_t)n1, (size_t)NW_MAX * QPB_MAX);   // nQB * S <= NW while S > 1; S = 1 beyond
  return std::min(a, (size_t)n1 * smax) + QPB_MAX;
}

// register r of the 32x32 accumulator of lane half `hi` holds MFMA row 8 (r >> 2) + 4 hi + (r & 3)
MX_D int row_of(int r, int hi) { return 8 * (r >> 2) + 4 * hi + (r & 3); }

// ---------------- pack: norms, swizzled tiles, key constants -------------------------------------------------------
// y = 0: norm1[i] = |q_i - 128|^2.   y = 1: one thread per train slot t < ntilesPadded * 32:
//   tiles[t >> 5] row t & 31 = b - 128 in 16-byte slots, slot s stored at s ^ ((row >> 1) & 7)
//   cst[t] = (|b'|^2 + 2 sum b') << 8 | ((tile % 12) + 1) << 4 | register of the row;   past the end: zeros / NONE | idx
//   norm2[t] = |b'|^2
__device__ __forceinline__ void pack_body(const uint8_t *d1, int n1, int *norm1, const uint8_t *d2, int n2, int slots,
                                          unsigned char *tiles, int *cst, int *norm2) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (blockIdx.y == 0) {
    if (i >= n1) return;
    const v4i *p = reinterpret_cast<const v4i *>(d1 + (size_t)i * 128);
    int s = 0;
#pragma unroll
    for (int q = 0; q < 8; q++) {
      const v4i v = p[q];
#pragma unroll
      for (int w = 0; w < 4; w++) {
        const int x = v[w] ^ 0x80808080;
#pragma unroll
        for (int b = 0; b < 4; b++) { const int e = (int)(signed char)((x >> (8 * b)) & 0xff); s += e * e; }
      }
    }
    norm1[i] = s;
    return;
  }
  if (i >= slots) return;
  const int tile = i >> 5, row = i & 31;
  const int idx = (((tile % CHUNK) + 1) << 4) | (4 * (row >> 3) + (row & 3));
  unsigned char *dst = tiles + (size_t)tile * TILE_B + row * 128;
  const int sw = (row >> 1) & 7;
  if (i >= n2) {
#pragma unroll
    for (int q = 0; q < 8; q++) *reinterpret_cast<v4i *>(dst + ((q ^ sw) << 4)) = (v4i){0, 0, 0, 0};
    cst[i] = NONE | idx;
    return;
  }
  const v4i *p = reinterpret_cast<const v4i *>(d2 + (size_t)i * 128);
  int s = 0, lin = 0;
#pragma unroll
  for (int q = 0; q < 8; q++) {
    v4i v = p[q];
#pragma unroll
    for (int w = 0; w < 4; w++) {
      v[w] ^= 0x80808080;
      const int x = v[w];
#pragma unroll
      for (int b = 0; b < 4; b++) { const int e = (int)(signed char)((x >> (8 * b)) & 0xff); s += e * e; lin += e; }
    }
    *reinterpret_cast<v4i *>(dst + ((q ^ sw) << 4)) = v;
  }
  norm2[i] = s;
  cst[i] = ((s + 2 * lin) << 8) | idx;
}

// exact |a - b|^2 of two 128-byte descriptors: na + nb - 2 (a-128).(b-128), dot4 on the signed bytes
MX_D int exact_dist(const uint8_t *a, int na, const uint8_t *b, int nb) {
  const v4i *pa = reinterpret_cast<const v4i *>(a), *pb = reinterpret_cast<const v4i *>(b);
  int dot = 0;
#pragma unroll
  for (int q = 0; q < 8; q++) {
    const v4i x = pa[q], y = pb[q];
#pragma unroll
    for (int w = 0; w < 4; w++) dot = __builtin_amdgcn_sdot4(x[w] ^ 0x80808080, y[w] ^ 0x80808080, dot, false);
  }
  return na + nb - 2 * dot;
}

// Exact distances of one group (the 16 rows of tile `tile` that lane half `hi` of the sweeps owns) from query `qd`, by the
// 16 lanes l = 0..15 of a quarter wave: lane l returns the distance of row row_of(l, hi).  Loads are coalesced: in step k
// the lanes l < 8 read the eight 16-byte slices of row k, the lanes l >= 8 those of row k + 8 (two whole 128-byte rows per
// step instead of sixteen scattered 16-byte pieces), partial dot products are summed over the eight lanes of a row.
MX_D int group_dist16(const uint8_t *qd, int na, const uint8_t *d2, const int *norm2, int n2, int tile, int hi, int l, int *t_out) {
  const int slice = l & 7, half = l >> 3;
  v4i q = reinterpret_cast<const v4i *>(qd)[slice];
  q[0] ^= 0x80808080; q[1] ^= 0x80808080; q[2] ^= 0x80808080; q[3] ^= 0x80808080;
  int part[8];
#pragma unroll
  for (int k = 0; k < 8; k++) {
    const int t = tile * 32 + row_of(k + 8 * half, hi);
    v4i y = {(int)0x80808080, (int)0x80808080, (int)0x80808080, (int)0x80808080};
    if (t < n2) y = reinterpret_cast<const v4i *>(d2 + (size_t)t * 128)[slice];
    int dot = 0;
#pragma unroll
    for (int c = 0; c < 4; c++) dot = __builtin_amdgcn_sdot4(q[c], y[c] ^ 0x80808080, dot, false);
    part[k] = dot;
  }
  // sum over the 8 lanes of a half row with three DPP adds (half-row mirror, quad reverse, quad pair swap): every lane ends
  // with the total; integer sums, so the order is immaterial
#pragma unroll
  for (int k = 0; k < 8; k++) {
    part[k] += __builtin_amdgcn_update_dpp(0, part[k], 0x141, 0xf, 0xf, false);
    part[k] += __builtin_amdgcn_update_dpp(0, part[k], 0x1B, 0xf, 0xf, false);
    part[k] += __builtin_amdgcn_update_dpp(0, part[k], 0xB1, 0xf, 0xf, false);
  }
  int dot = part[0];
#pragma unroll
  for (int k = 1; k < 8; k++) dot = slice == k ? part[k] : dot;
  const int t = tile * 32 + row_of(l, hi);
  *t_out = t;
  return t < n2 ? na + norm2[t] - 2 * dot : BIG;
}

// ---------------- staging: 4 tiles + their constants, global -> LDS directly ------------------------------------------
typedef const unsigned char __attribute__((address_space(1))) *gbptr;
typedef unsigned char __attribute__((address_space(3))) *lbptr;
MX_D void stage_group(const unsigned char *tiles, const int *cst, int g0, unsigned char *buf, int wave, int lane) {
  const unsigned char *src = tiles + (size_t)g0 * TILE_B + lane * 16;
#pragma unroll
  for (int i = 0; i < 4; i++) {
    const int chunk = wave + 4 * i;   // 1 KB per wave instruction
    __builtin_amdgcn_global_load_lds((gbptr)(src + chunk * 1024), (lbptr)(buf + chunk * 1024), 16, 0, 0);
  }
  if (wave < 2)
    __builtin_amdgcn_global_load_lds((gbptr)(reinterpret_cast<const unsigned char *>(cst + (size_t)g0 * 32) + wave * 256 + lane * 4),
                                     (lbptr)(buf + TPS * TILE_B + wave * 256), 4, 0, 0);
}
MX_D v4i read_a(const unsigned char *tile, int row, int kb, int hi) {
  const int slot = 2 * kb + hi;
  return *reinterpret_cast<const v4i *>(tile + row * 128 + ((slot ^ ((row >> 1) & 7)) << 4));
}
// the query fragment: a'' = 127 - a = -(a - 128) - 1 (u8 -> i8 by x ^ 0x7f), bytes [32 kb + 16 hi, +16)
MX_D v4i load_q(const uint8_t *base, int row, int kb, int hi) {
  v4i v = *reinterpret_cast<const v4i *>(base + (size_t)row * 128 + 32 * kb + 16 * hi);
  v[0] ^= 0x7f7f7f7f; v[1] ^= 0x7f7f7f7f; v[2] ^= 0x7f7f7f7f; v[3] ^= 0x7f7f7f7f;
  return v;
}
MX_D int tree_min16(const int *k) {
  const int t0 = imin3(k[0], k[1], k[2]), t1 = imin3(k[3], k[4], k[5]), t2 = imin3(k[6], k[7], k[8]);
  const int t3 = imin3(k[9], k[10], k[11]), t4 = imin3(k[12], k[13], k[14]);
  return min(imin3(t0, t1, t2), imin3(t3, t4, k[15]));
}
// train index of a key of the current chunk (low byte = (tile in chunk + 1) << 4 | register)
MX_D int decode_idx(int lb, int chunkTile0, int hi) {
  return (chunkTile0 + (lb >> 4) - 1) * 32 + row_of(lb & 15, hi);
}

// ---------------- the sweep: MODE 0 = per (query, split) top-2 of the group minima, MODE 1 = NNj + event groups ------------
// One instruction stream per wave keeps both pipes busy: while the four MFMAs of a (tile, query set) chain run, the wave
// reduces the accumulators of the previous chain (26 VALU), so the matrix pipe never waits for a whole wave to leave its
// epilogue.  Fragments and key constants of the next tile are read from LDS one tile ahead (two register sets).
constexpr int EVCAP = 16;                  // event slots per (undecided query, split, lane half); more -> exact fallback
struct SweepArgs {
  const uint8_t *d1;
  const int *norm1;
  const unsigned char *tiles;
  const int *cst;
  MatchGeom g;
  int4 *partial;          // MODE 0
  const int *dmin, *undecided, *nUndecided;   // MODE 1
  int2 *partial2;
  int *evCnt, *ev;
};

template <int MODE, int QSETS>
__device__ __forceinline__ void sweep_body(const SweepArgs &A) {
  constexpr int QPB = qpb_of(QSETS);
  __shared__ __attribute__((aligned(16))) unsigned char sm[2][STAGE_B];
  const MatchGeom g = A.g;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int col = lane & 31, hi = lane >> 5;
  const int nQ = MODE == 0 ? g.n1 : *A.nUndecided;
  int sp, qb, S, tilesPerSplit;
  if (MODE == 0) { sp = blockIdx.y; qb = blockIdx.x; S = g.S; tilesPerSplit = g.tilesPerSplit; }
  else {
    const Sweep2Geom G2 = sweep2_geom(nQ, g.n2, QSETS);
    S = G2.S; tilesPerSplit = G2.tilesPerSplit;
    qb = (int)blockIdx.x / S; sp = (int)blockIdx.x - qb * S;
  }
  if (qb * QPB >= nQ) return;
  const int q0 = qb * QPB + wave * (32 * QSETS);
  v4i bq[QSETS][4];
  int mA[QSETS], mB[QSETS], iA[QSETS], iB[QSETS];   // MODE 0: m1, m2, i1, i2;  MODE 1: mj, threshold key, ij, event count
  int qsel[QSETS];
#pragma unroll
  for (int s = 0; s < QSETS; s++) {
    const int u = min(q0 + 32 * s + col, nQ - 1);
    qsel[s] = MODE == 0 ? u : A.undecided[u];
#pragma unroll
    for (int kb = 0; kb < 4; kb++) bq[s][kb] = load_q(A.d1, qsel[s], kb, hi);
    mA[s] = NONE; iA[s] = -1;
    if (MODE == 0) { mB[s] = NONE; iB[s] = -1; }
    else { mB[s] = (A.dmin[qsel[s]] - A.norm1[qsel[s]]) << 8; iB[s] = 0; }   // d < Dmin  <=>  key < (Dmin - |a'|^2) << 8
  }
  const int ntiles4 = (((g.n2 + 31) >> 5) + TPS - 1) & ~(TPS - 1);
  const int tBeg = sp * tilesPerSplit, tEnd = min(tBeg + tilesPerSplit, ntiles4);
  auto flush = [&](int chunkTile0) {
#pragma unroll
    for (int s = 0; s < QSETS; s++) {
      if (MODE == 0) {
        const int lb1 = mA[s] & 255, lb2 = mB[s] & 255;
        const int n2i = lb2 ? decode_idx(lb2, chunkTile0, hi) : (lb1 ? iA[s] : iB[s]);
        const int n1i = lb1 ? decode_idx(lb1, chunkTile0, hi) : iA[s];
        iA[s] = n1i; iB[s] = n2i;
        mA[s] &= ~255; mB[s] &= ~255;
      } else {
        const int lb = mA[s] & 255;
        if (lb) iA[s] = decode_idx(lb, chunkTile0, hi);
        mA[s] &= ~255;
      }
    }
  };
  // reduce one accumulator: 16 keys, v_min3 tree, then the running state
  auto epilogue = [&](const v16i &acc, const int *C, int s, int tile) {
    int k[16];
#pragma unroll
    for (int r = 0; r < 16; r++) k[r] = (acc[r] << 9) + C[r];
    const int t = tree_min16(k);
    if (MODE == 0) {
      mB[s] = imed3(mA[s], mB[s], t);
      mA[s] = min(mA[s], t);
    } else if (t < mB[s]) {
      // a train of this group is closer than Dmin: k_match_events takes the whole group (its other rows included)
      const int u = q0 + 32 * s + col;
      if (u < nQ) {
        if (iB[s] < EVCAP) A.ev[(((size_t)u * S + sp) * 2 + hi) * EVCAP + iB[s]] = tile;
        iB[s]++;
      }
    } else mA[s] = min(mA[s], t);
  };
  auto load_af = [&](const unsigned char *buf, int q, v4i *af) {
#pragma unroll
    for (int kb = 0; kb < 4; kb++) af[kb] = read_a(buf + q * TILE_B, col, kb, hi);
  };
  auto load_c = [&](const unsigned char *buf, int q, int *C) {
#pragma unroll
    for (int gq = 0; gq < 4; gq++) {
      const v4i c4 = *reinterpret_cast<const v4i *>(buf + TPS * TILE_B + q * 128 + (8 * gq + 4 * hi) * 4);
      C[4 * gq] = c4[0]; C[4 * gq + 1] = c4[1]; C[4 * gq + 2] = c4[2]; C[4 * gq + 3] = c4[3];
    }
  };
  v4i af[2][4];
  int C[2][16];
  v16i acc[2];
  acc[1] = (v16i){0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  // the pipeline starts with a neutral pending chain: zero accumulator, NONE constants -> keys that change nothing
#pragma unroll
  for (int r = 0; r < 16; r++) C[1][r] = NONE;
  int pendTile = 0;
  if (tBeg < tEnd) stage_group(A.tiles, A.cst, tBeg, sm[0], wave, lane);
  int it = 0;
  for (int tg = tBeg; tg < tEnd; tg += TPS, it++) {
    const unsigned char *buf = sm[it & 1];
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tg + TPS < tEnd) stage_group(A.tiles, A.cst, tg + TPS, sm[(it & 1) ^ 1], wave, lane);
    load_af(buf, 0, af[0]);
    load_c(buf, 0, C[0]);
#pragma unroll
    for (int q = 0; q < TPS; q++) {
      const int cur = q & 1;
      // Phase s of a tile: the chain of (tile q, set s) beside the reduction of the previous chain -- (tile q, set s - 1),
      // or the last set of the previous tile.  QSETS is even, so the chains alternate between the two accumulators.
      // The fragment / constant reads of the NEXT tile are issued as a burst in front of phases 0 and 1 and fenced there
      // (sched_barrier): left to the scheduler they sink to just before their first use and every MFMA waits for LDS.
#pragma unroll
      for (int s = 0; s < QSETS; s++) {
        if (q + 1 < TPS) {
          if (s == 0) load_af(buf, q + 1, af[cur ^ 1]);
          if (s == 1) load_c(buf, q + 1, C[cur ^ 1]);     // C[cur ^ 1] was the pending chain's until phase 0 ended
        }
        __builtin_amdgcn_sched_barrier(0);
        v16i &an = acc[s & 1];
        an = (v16i){0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
        for (int kb = 0; kb < 4; kb++) an = __builtin_amdgcn_mfma_i32_32x32x32_i8(af[cur][kb], bq[s][kb], an, 0, 0, 0);
        if (s == 0) epilogue(acc[1], C[cur ^ 1], QSETS - 1, pendTile);
        else epilogue(acc[(s - 1) & 1], C[cur], s - 1, tg + q);
#pragma unroll
        for (int kb = 0; kb < 4; kb++) {   // one MFMA, then a quarter of the reduction
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x002, 7, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      pendTile = tg + q;
    }
    if ((it % (CHUNK / TPS)) == CHUNK / TPS - 1) {
      // end of an index chunk: drain the pending chain, then move the indices of changed keys out of the low byte
      epilogue(acc[1], C[1], QSETS - 1, pendTile);
      acc[1] = (v16i){0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
      for (int r = 0; r < 16; r++) C[1][r] = NONE;
      flush(tg + TPS - CHUNK);
    }
  }
  if (it % (CHUNK / TPS)) {
    epilogue(acc[1], C[1], QSETS - 1, pendTile);
    flush(tBeg + (it / (CHUNK / TPS)) * CHUNK);
  }
  // the two lane halves of a query saw different rows: merge, convert keys to distances, store
#pragma unroll
  for (int s = 0; s < QSETS; s++) {
    const int q = q0 + 32 * s + col;
    const int na = A.norm1[qsel[s]];
    if (MODE == 0) {
      int d0 = iA[s] < 0 ? BIG : (mA[s] >> 8) + na, j0 = iA[s] < 0 ? BIG : iA[s];
      int dd1 = iB[s] < 0 ? BIG : (mB[s] >> 8) + na, j1 = iB[s] < 0 ? BIG : iB[s];
      const int od0 = __shfl_xor(d0, 32), oj0 = __shfl_xor(j0, 32), od1 = __shfl_xor(dd1, 32), oj1 = __shfl_xor(j1, 32);
      if (lex_less(od0, oj0, d0, j0)) {
        if (lex_less(od1, oj1, d0, j0)) { dd1 = od1; j1 = oj1; } else { dd1 = d0; j1 = j0; }
        d0 = od0; j0 = oj0;
      } else if (lex_less(od0, oj0, dd1, j1)) { dd1 = od0; j1 = oj0; }
      if (hi == 0 && q < nQ) A.partial[(size_t)q * S + sp] = make_int4(d0, j0, dd1, j1);
    } else {
      int dj = iA[s] < 0 ? BIG : (mA[s] >> 8) + na, tj = iA[s] < 0 ? BIG : iA[s];
      const int od = __shfl_xor(dj, 32), ot = __shfl_xor(tj, 32);
      if (lex_less(od, ot, dj, tj)) { dj = od; tj = ot; }
      if (q < nQ) {
        A.evCnt[((size_t)q * S + sp) * 2 + hi] = iB[s];
        if (hi == 0) A.partial2[(size_t)q * S + sp] = make_int2(dj, tj);
      }
    }
  }
}

// ---------------- decide: merge splits, the hidden candidate of NN0's group, j = 1 of the walk ---------------------------
// 16 lanes per query.  sweep 1 ranks group minima, so the one candidate it cannot have seen is the second-best inside the
// group (same tile, same lane half) of the overall winner: lane l recomputes the distance of that group's row l exactly.
#ifndef MODSX_DECIDE_Q
#define MODSX_DECIDE_Q 64
#endif
constexpr int DECIDE_Q = MODSX_DECIDE_Q;   // queries per workgroup of k_match_decide (16 lanes each)
__device__ __forceinline__ void decide_body(const uint8_t *d1, const int *norm1, const uint8_t *d2, const int *norm2,
                                            const int4 *partial, MatchGeom g, const double *pos2, double sqminratio,
                                            double contrDistSq, MatchRow *rows, int *dmin, int *undecided, int *nUndecided) {
  // undecided queries are compacted with ONE global atomic per workgroup (a counter word takes ~90 atomics per us)
  __shared__ int sList[DECIDE_Q], sCount, sBase;
  if (threadIdx.x == 0) sCount = 0;
  __syncthreads();
  const int l = threadIdx.x & 15;
  const int q = blockIdx.x * DECIDE_Q + (threadIdx.x >> 4);
  const bool live = q < g.n1;
  const int qc = live ? q : g.n1 - 1;
  int d0 = BIG, i0 = BIG, dd1 = BIG, j1 = BIG;
  for (int s = l; s < g.S; s += 16) {
    const int4 p = partial[(size_t)qc * g.S + s];
    if (lex_less(p.x, p.y, d0, i0)) {
      if (lex_less(p.z, p.w, d0, i0)) { dd1 = p.z; j1 = p.w; } else { dd1 = d0; j1 = i0; }
      d0 = p.x; i0 = p.y;
    } else if (lex_less(p.x, p.y, dd1, j1)) { dd1 = p.x; j1 = p.y; }
  }
#pragma unroll
  for (int m = 8; m >= 1; m >>= 1) {
    const int od0 = __shfl_xor(d0, m), oi0 = __shfl_xor(i0, m), od1 = __shfl_xor(dd1, m), oj1 = __shfl_xor(j1, m);
    if (lex_less(od0, oi0, d0, i0)) {
      if (lex_less(od1, oj1, d0, i0)) { dd1 = od1; j1 = oj1; } else { dd1 = d0; j1 = i0; }
      d0 = od0; i0 = oi0;
    } else if (lex_less(od0, oi0, dd1, j1)) { dd1 = od0; j1 = oi0; }
  }
  if (i0 != BIG) {
    const int tile = i0 >> 5, hi = ((i0 & 31) >> 2) & 1;
    int ht;
    int hd = group_dist16(d1 + (size_t)qc * 128, norm1[qc], d2, norm2, g.n2, tile, hi, l, &ht);
    if (ht >= g.n2 || ht == i0) { hd = BIG; ht = BIG; }
#pragma unroll
    for (int m = 8; m >= 1; m >>= 1) {
      const int od = __shfl_xor(hd, m), ot = __shfl_xor(ht, m);
      if (lex_less(od, ot, hd, ht)) { hd = od; ht = ot; }
    }
    if (lex_less(hd, ht, dd1, j1)) { dd1 = hd; j1 = ht; }
  }
  if (live && l == 0) {
    MatchRow o;
    o.t0 = i0 == BIG ? -1 : i0; o.t1 = j1 == BIG ? -1 : j1; o.tj = -1; o.nless = 0; o.nbad = 0;
    o.d0 = (float)d0; o.d1 = (float)dd1; o.dj = 0.f;
    int dm = 0;
    if (i0 != BIG && j1 != BIG) {
      if (ratio_pass((float)d0, (float)dd1, sqminratio)) { o.tj = j1; o.dj = (float)dd1; }       // accepted at j = 1
      else {
        const double dx = pos2[2 * i0] - pos2[2 * j1], dy = pos2[2 * i0 + 1] - pos2[2 * j1 + 1];
        if (dx * dx + dy * dy > contrDistSq) o.nbad = 1;                                      // first contradictive
        else {
          dm = ratio_dmin(d0, sqminratio);
          if (dm <= MAXD) {          // otherwise no distance can pass the ratio test: the walk ends without a match
            sList[atomicAdd(&sCount, 1)] = q;
            o.nless = -1;   // filled by sweep 2
          }
        }
      }
    }
    rows[q] = o;
    dmin[q] = dm;
  }
  __syncthreads();
  if (threadIdx.x == 0 && sCount) sBase = atomicAdd(nUndecided, sCount);
  __syncthreads();
  if ((int)threadIdx.x < sCount) undecided[sBase + threadIdx.x] = sList[threadIdx.x];
}

// ---------------- events: the groups of sweep 2 that hold a train below Dmin, recomputed exactly ---------------------------
// One wave per undecided query; each 16-lane quarter walks the event lists of every fourth (split, lane half) stream, one
// group (16 rows) at a time.  Output per query: nless, nbad and the lex-smallest (d, t) with d >= Dmin among the rows of the
// event groups (the sweep left those groups out of its own minimum).  A stream that ran out of its EVCAP slots sends the
// query to the exact fallback: every train, one per lane (low-entropy inputs with thousands of near-duplicates).
__device__ __forceinline__ void events_body(const uint8_t *d1, const int *norm1, const uint8_t *d2, const int *norm2, MatchGeom g,
                                            const double *pos2, double contrDistSq, MatchRow *rows, const int *dmin,
                                            const int *undecided, const int *nUndecided, const int *evCnt, const int *ev,
                                            int nn, const int2 *partial2) {
  __shared__ int sRec[4][MATCH_NN_MAX];  // per wave: the event groups of its query as (tile << 1 | lane half); fewer than nn of them
  const int w = threadIdx.x >> 6;
  const int u = blockIdx.x * 4 + w;
  const int nUnd = *nUndecided;
  if (u >= nUnd) return;
  const int lane = threadIdx.x & 63, l = lane & 15, sub = lane >> 4;
  const Sweep2Geom G2 = sweep2_geom(nUnd, g.n2, g.qs);      // the splits sweep 2 chose for this count
  const int S2 = G2.S;
  const int nst = 2 * S2;
  // Every event group holds at least one train below Dmin and at most one of all those trains is NN0, so nn or more
  // groups mean nless > nn - 2: the walk gives up (matching.cpp:435-457) and nothing has to be recomputed.  (Look-alike
  // regions -- a thousand similar blobs -- produce exactly this, and would otherwise dominate the kernel.)
  int total = 0;
  for (int st = lane; st < nst; st += 64) total += evCnt[(size_t)u * nst + st];
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) total += __shfl_xor(total, m);
  const int q = undecided[u];
  // the last step of the walk for this query: NNj = the lex-smallest of the event groups' candidate and the splits' minima of
  // sweep 2, then the row gets tj / dj / nless / nbad (this used to be a launch of its own)
  auto finish = [&](int nlessF, int nbadF, int djF, int tjF) {
    for (int sp = lane; sp < S2; sp += 64) {
      const int2 p = partial2[(size_t)u * S2 + sp];
      if (lex_less(p.x, p.y, djF, tjF)) { djF = p.x; tjF = p.y; }
    }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
      const int od = __shfl_xor(djF, m), ot = __shfl_xor(tjF, m);
      if (lex_less(od, ot, djF, tjF)) { djF = od; tjF = ot; }
    }
    if (lane == 0) {
      MatchRow o = rows[q];
      o.tj = tjF == BIG ? -1 : tjF;
      o.dj = (float)djF;
      o.nless = nlessF; o.nbad = nbadF;
      rows[q] = o;
    }
  };
  if (total >= nn) { finish(nn, 0, BIG, BIG); return; }
  const int t0 = rows[q].t0, Dm = dmin[q], na = norm1[q];
  const double x0 = pos2[2 * t0], y0 = pos2[2 * t0 + 1];
  const uint8_t *qd = d1 + (size_t)q * 128;
  int nless = 0, nbad = 0, dj = BIG, tj = BIG;
  // one group (tile, lane half) per quarter wave: lane l gets the exact distance of the group's row l
  auto visit_group = [&](int tile, int hi, bool active) {
    int t;
    const int d = group_dist16(qd, na, d2, norm2, g.n2, active ? tile : 0, hi, l, &t);
    if (!active || t >= g.n2 || t == t0) return;
    if (d < Dm) {
      nless++;
      // geometric consistency with NN0 (distanceSq, matching.cpp:174-179), f64
      const double dx = x0 - pos2[2 * t], dy = y0 - pos2[2 * t + 1];
      if (dx * dx + dy * dy > contrDistSq) nbad++;
    } else if (lex_less(d, t, dj, tj)) { dj = d; tj = t; }
  };
  // gather the (fewer than nn <= MATCH_NN_MAX) logged groups of all streams into one list; a stream that ran out of slots is
  // rescanned exactly over its own tiles afterwards (its logged groups are then ignored); the groups it did not flag hold
  // no train below Dmin and their candidates d >= Dmin are already in the sweep's minimum
  int nrec = 0;
  bool anyOver = false;
  for (int s0 = 0; s0 < nst; s0 += 64) {
    const int st = s0 + lane;
    int c = st < nst ? evCnt[(size_t)u * nst + st] : 0;
    const bool over = c > EVCAP;
    anyOver = anyOver || __any(over);
    if (over) c = 0;
    int pre = c;   // inclusive scan over the lanes
#pragma unroll
    for (int m = 1; m < 64; m <<= 1) { const int v = __shfl_up(pre, m); if (lane >= m) pre += v; }
    const int base = nrec + pre - c;
    for (int e = 0; e < c; e++)
      if (base + e < MATCH_NN_MAX) sRec[w][base + e] = (ev[((size_t)u * nst + st) * EVCAP + e] << 1) | (st & 1);
    nrec += __shfl(pre, 63);
  }
  nrec = min(nrec, MATCH_NN_MAX);
  for (int b = 0; b < nrec; b += 4) {      // wave-uniform trip count: the quarter waves shuffle among their own lanes
    const int e = b + sub;
    const int rec = e < nrec ? sRec[w][e] : 0;
    visit_group(rec >> 1, rec & 1, e < nrec);
  }
  if (anyOver) {
    for (int st = 0; st < nst; st++) {
      if (evCnt[(size_t)u * nst + st] <= EVCAP) continue;
      const int tb = (st >> 1) * G2.tilesPerSplit, te = min(tb + G2.tilesPerSplit, (g.n2 + 31) >> 5);
      for (int tile = tb; tile < te; tile += 4) visit_group(tile + sub, st & 1, tile + sub < te);
    }
  }
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) {
    nless += __shfl_xor(nless, m);
    nbad += __shfl_xor(nbad, m);
    const int od = __shfl_xor(dj, m), ot = __shfl_xor(tj, m);
    if (lex_less(od, ot, dj, tj)) { dj = od; tj = ot; }
  }
  finish(nless, nbad, dj, tj);
}

// ---- workspace layout: ONE description used by the size query and by the launcher -------------------------------------------
struct MatchLayout {
  int S, tilesPerSplit, slots;
  size_t norm1, norm2, cst, tiles, partial, partial2, dmin, undecided, evCnt, ev, evRes, counter, bytes;
};
static MatchLayout match_layout(int n1, int n2, int qs) {
  MatchLayout L;
  const int QPB = qpb_of(qs);
  const int nQB = (n1 + QPB - 1) / QPB;
  const int ntiles = (n2 + 31) / 32;
  // one round of workgroups: 3 per CU (the sweeps hold ~150 VGPRs) x 256 CUs; a second, partly filled round costs as much
  // as the first.  Many query blocks (N > 196 k) simply take several rounds.
#ifdef MATCH_TRACE
  static const int nwEnv = getenv("MODSX_MATCH_NW") ? atoi(getenv("MODSX_MATCH_NW")) : 0;   // workgroups per round, to trace 1 / 2 / 3 per CU
  int S = (nwEnv > 0 ? nwEnv : 256 * sweep_wps(qs)) / nQB;
#else
  int S = (256 * sweep_wps(qs)) / nQB;
#endif
  if (S > ntiles / CHUNK) S = ntiles / CHUNK;     // at least one chunk per split
  if (S < 1) S = 1;
  int tps = (ntiles + S - 1) / S;
  tps = ((tps + CHUNK - 1) / CHUNK) * CHUNK;
  S = (ntiles + tps - 1) / tps;
  if (S < 1) S = 1;
  L.S = S; L.tilesPerSplit = tps; L.slots = S * tps * 32;
  size_t w = 0;
  auto take = [&](size_t bytes) { const size_t o = w; w += (bytes + 255) & ~(size_t)255; return o; };
  L.norm1 = take((size_t)n1 * 4);
  L.norm2 = take((size_t)L.slots * 4);
  L.cst = take((size_t)L.slots * 4);
  L.tiles = take((size_t)L.slots * 128);
  L.partial = take((size_t)n1 * S * 16);
  const size_t e2 = sweep2_entries(n1, n2);
  L.partial2 = take(e2 * 8);
  L.dmin = take((size_t)n1 * 4);
  L.undecided = take((size_t)n1 * 4);
  L.evCnt = take(e2 * 2 * 4);
  L.ev = take(e2 * 2 * EVCAP * 4);
  L.evRes = take((size_t)n1 * 16);
  L.counter = take(64);
  L.bytes = w;
  return L;
}
// the larger of the two geometries: the caller sizes the workspace before the launcher picks one
size_t match_workspace_bytes(int n1, int n2) { return std::max(match_layout(n1, n2, 2).bytes, match_layout(n1, n2, 4).bytes); }

// ---- batched entry points: blockIdx.z selects one of up to MATCH_MAXB independent problems (the pairs of a launch set).
struct MatchProblem {
  const uint8_t *d1, *d2;
  const double *pos2;
  int *norm1, *norm2, *cst, *dmin, *undecided, *counter, *evCnt, *ev;
  unsigned char *tiles;
  int4 *partial, *evRes;
  int2 *partial2;
  MatchRow *rows;
  MatchGeom g;
  int slots;
};
struct MatchBatch { MatchProblem p[MATCH_MAXB]; };

__global__ __launch_bounds__(256) void k_match_pack(MatchBatch b) {
  const MatchProblem &P = b.p[blockIdx.z];
  if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) *P.counter = 0;   // the undecided count (decide adds to it): no fill launch of its own
  pack_body(P.d1, P.g.n1, P.norm1, P.d2, P.g.n2, P.slots, P.tiles, P.cst, P.norm2);
}
#ifdef MATCH_TRACE
// debugging aid (tools/trace_match.py): when and where every workgroup of the last k_match_sweep1 launch ran
__device__ unsigned long long g_mtrace[16384][4];
#endif
template <int QS>
__global__ __launch_bounds__(256, sweep_wps(QS)) void k_match_sweep1(MatchBatch b) {
  const MatchProblem &P = b.p[blockIdx.z];
  if ((int)blockIdx.x * qpb_of(QS) >= P.g.n1 || (int)blockIdx.y >= P.g.S) return;
#ifdef MATCH_TRACE
  const int wg = blockIdx.x + gridDim.x * blockIdx.y;
  if (threadIdx.x == 0 && wg < 16384) {
    unsigned hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    g_mtrace[wg][0] = wall_clock64();
    g_mtrace[wg][2] = ((unsigned long long)xcc << 32) | hw;
    g_mtrace[wg][3] = __builtin_readcyclecounter();
  }
#endif
  SweepArgs A;
  A.d1 = P.d1; A.norm1 = P.norm1; A.tiles = P.tiles; A.cst = P.cst; A.g = P.g; A.partial = P.partial;
  sweep_body<0, QS>(A);
#ifdef MATCH_TRACE
  __syncthreads();
  if (threadIdx.x == 0 && wg < 16384) { g_mtrace[wg][1] = wall_clock64(); g_mtrace[wg][3] = __builtin_readcyclecounter() - g_mtrace[wg][3]; }
#endif
}
__global__ __launch_bounds__(16 * DECIDE_Q) void k_match_decide(MatchBatch b, double sqminratio, double contrDistSq) {
  const MatchProblem &P = b.p[blockIdx.z];
  if ((int)blockIdx.x * DECIDE_Q >= P.g.n1) return;
  decide_body(P.d1, P.norm1, P.d2, P.norm2, P.partial, P.g, P.pos2, sqminratio, contrDistSq, P.rows, P.dmin, P.undecided,
              P.counter);
}
template <int QS>
__global__ __launch_bounds__(256, sweep_wps(QS)) void k_match_sweep2(MatchBatch b) {
  const MatchProblem &P = b.p[blockIdx.z];
  SweepArgs A;
  A.d1 = P.d1; A.norm1 = P.norm1; A.tiles = P.tiles; A.cst = P.cst; A.g = P.g;
  A.dmin = P.dmin; A.undecided = P.undecided; A.nUndecided = P.counter; A.partial2 = P.partial2; A.evCnt = P.evCnt; A.ev = P.ev;
  sweep_body<1, QS>(A);
}
__global__ __launch_bounds__(256) void k_match_events(MatchBatch b, double contrDistSq, int nn) {
  const MatchProblem &P = b.p[blockIdx.z];
  if ((int)blockIdx.x * 4 >= P.g.n1) return;
  events_body(P.d1, P.norm1, P.d2, P.norm2, P.g, P.pos2, contrDistSq, P.rows, P.dmin, P.undecided, P.counter, P.evCnt, P.ev,
              nn, P.partial2);
}
// Problems with n1 == 0 or n2 == 0 must be left out by the caller.  workspace[i] holds match_workspace_bytes(n1[i], n2[i]).
void launch_match_batch(hipStream_t s, int nb, const uint8_t *const *d1, const int *n1, const uint8_t *const *d2, const int *n2,
                        const double *const *pos2, double sqminratio, double contrDistSq, int nn, MatchRow *const *rows,
                        void *const *workspace, hipEvent_t *evSweep1) {
  if (nb <= 0) return;
  MatchBatch b;
  memset(&b, 0, sizeof b);
  int maxN1 = 0, maxS = 0, maxSlots = 0;
  const int qs = match_qsets(nb, n1[0], n2[0]);
  for (int i = 0; i < nb; i++) {
    MatchProblem &P = b.p[i];
    const MatchLayout L = match_layout(n1[i], n2[i], qs);
    P.g.n1 = n1[i]; P.g.n2 = n2[i]; P.g.S = L.S; P.g.tilesPerSplit = L.tilesPerSplit; P.g.qs = qs;
    P.slots = L.slots;
    char *w = (char *)workspace[i];
    P.norm1 = (int *)(w + L.norm1); P.norm2 = (int *)(w + L.norm2); P.cst = (int *)(w + L.cst);
    P.tiles = (unsigned char *)(w + L.tiles);
    P.partial = (int4 *)(w + L.partial); P.partial2 = (int2 *)(w + L.partial2);
    P.dmin = (int *)(w + L.dmin); P.undecided = (int *)(w + L.undecided);
    P.evCnt = (int *)(w + L.evCnt); P.ev = (int *)(w + L.ev); P.evRes = (int4 *)(w + L.evRes);
    P.counter = (int *)(w + L.counter);
    P.d1 = d1[i]; P.d2 = d2[i]; P.pos2 = pos2[i]; P.rows = rows[i];
    maxN1 = std::max(maxN1, n1[i]); maxS = std::max(maxS, L.S); maxSlots = std::max(maxSlots, L.slots);
  }
  hipLaunchKernelGGL(k_match_pack, dim3((std::max(maxN1, maxSlots) + 255) / 256, 2, nb), dim3(256), 0, s, b);
  const int QPB = qpb_of(qs), NW2 = 256 * sweep_wps(qs);
  const dim3 grid((maxN1 + QPB - 1) / QPB, maxS, nb);
  if (evSweep1) hipEventRecord(evSweep1[0], s);
  if (qs == 4) hipLaunchKernelGGL(k_match_sweep1<4>, grid, dim3(256), 0, s, b);
  else hipLaunchKernelGGL(k_match_sweep1<2>, grid, dim3(256), 0, s, b);
  if (evSweep1) hipEventRecord(evSweep1[1], s);
  hipLaunchKernelGGL(k_match_decide, dim3((maxN1 + DECIDE_Q - 1) / DECIDE_Q, 1, nb), dim3(16 * DECIDE_Q), 0, s, b, sqminratio, contrDistSq);
  // sweep 2: one round of workgroups dealt out on the device as (undecided block, split); more only if there could be more
  // undecided query blocks than that
  const dim3 grid2(std::max(NW2, (maxN1 + QPB - 1) / QPB), 1, nb);
  if (qs == 4) hipLaunchKernelGGL(k_match_sweep2<4>, grid2, dim3(256), 0, s, b);
  else hipLaunchKernelGGL(k_match_sweep2<2>, grid2, dim3(256), 0, s, b);
  hipLaunchKernelGGL(k_match_events, dim3((maxN1 + 3) / 4, 1, nb), dim3(256), 0, s, b, contrDistSq, nn);
}

void launch_match(hipStream_t s, const uint8_t *d1, int n1, const uint8_t *d2, int n2, const double *pos2,
                  double sqminratio, double contrDistSq, int nn, MatchRow *rows, void *workspace) {
  if (n1 <= 0 || n2 <= 0) return;
  launch_match_batch(s, 1, &d1, &n1, &d2, &n2, &pos2, sqminratio, contrDistSq, nn, &rows, &workspace, nullptr);
}

}  // namespace mx

#ifdef MATCH_TRACE
extern "C" __attribute__((visibility("default"))) int modsx_debug_match_trace(unsigned long long *out, int n) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(mx::g_mtrace), (size_t)n * 32, 0, hipMemcpyDeviceToHost);
}
#endif
