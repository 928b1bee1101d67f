// kernels_match.hip -- brute-force 128-D squared-L2 matching with the FGINN ratio walk.
//
// Reference: MatchFlannFGINN (matching/matching.cpp:357-461) over an exact (linear) kNN:
//   squared L2 in f32 (exact: descriptors hold integers 0..255, sums < 2^24), neighbours sorted
//   ascending with ties by ascending train index, walk j = 1..nn-1:
//     accept at the first j with (float)d0/(float)dj <= ratio^2,
//     give up at the first j whose position is farther than contradDist from NN0's.
// Distances come from the int8 matrix cores: with a'' = 127 - a, b' = b - 128 (both in [-128,127])
//   |a-b|^2 - |a-128|^2 = (|b'|^2 + 2 sum b') + 2 a''.b' = c + 2 a''.b'   exactly in int32;   a''.b' = v_mfma_i32_32x32x32_i8 over K = 128.
//
// Round 5: the epilogue, not the matrix pipe, was what bound the sweeps (26 vector instructions per 4 MFMAs for the running
// top-2 of keys that carried their row index; a plain VALU instruction costs 4 cycles per wavefront, a 32x32x32 int8 MFMA 32).
// The sweeps now spend 13:
//  * c = 2 h + p.  The MFMA chain STARTS from h (its C operand is the per-row constant, read from LDS with the tile), so an
//    accumulator element is t = h + a''.b' and d - |a'|^2 = 2 t + p with no instruction per element.
//  * p, the parity of sum(b), is made a property of the TILE: k_match_pack partitions the trains by parity (stable) into two
//    regions of the tile array, each padded to whole stages of 4 tiles; the sweeps run over the VIRTUAL tile sequence
//    "even class, then odd class".  Equal distances are always in the same class, where slot order is train order, so
//    "first seen wins" still is "lowest train index wins".  The ranks come from a one-pass scan: every 256-train workgroup
//    publishes its two counts in an epoch-stamped status word and sums the words of the workgroups before it.
//  * TRAIN descriptors are the MFMA rows, QUERIES the columns: a lane owns ONE query per 32-query set and sees 16 trains (a
//    "group": fixed tile, fixed lane half) per tile.  It reduces the 16 elements with a v_min3 tree (8), forms ONE key
//    (2 t + p) << 8 | tile code (1: v_lshl_add with the tile's constant) and inserts it into its K = 4 smallest group keys
//    (3 v_med3 + v_min).  No row index anywhere in the loop.
//  * k_match_decide merges the streams (split x lane half) of a query into the K smallest groups G[0..K-1] by (d, tile, half),
//    recomputes groups EXACTLY (v_dot4 on the packed rows), one at a time as needed, and runs the reference's walk over the
//    rows that are CERTAIN: a recomputed row r is certain iff (d_r, tile_r) < (d, tile) of the first group not yet recomputed
//    -- every row outside the recomputed groups is at or after that group in (distance, slot) order.  With K >= 4, NN0 and
//    NN1 are always certain.  On multi-view descriptors 98.7 % of the queries end their walk inside the first three groups
//    (tests/match_model.py is the executable model of this logic, checked against the oracle on the CPU);
//  * only the rest go through k_match_resolve, a second sweep for them alone: Dmin = smallest integer distance passing the ratio
//    test against d0; groups whose minimum is below Dmin are logged (2 bytes in the stream's own LDS slots, no atomics), the
//    others feed the lane's running minimum (keys WITH the row there: the kernel is a hundredth of sweep 1's work); the
//    workgroup then recomputes its logged groups exactly for nless / nbad / NNj and adds them to the query's row.
//      accept  <=>  NNj exists, nbad == 0, nless <= nn-2          (rank of NNj is nless+1)
//  * staging: 4 tiles (4 KB each, 16-byte slots XOR-swizzled for conflict-free ds_read_b128) + their 128 row constants + 4
//    tile constants per barrier with direct global->LDS loads, double-buffered; a wave holds 2 x 32 queries, so every
//    fragment read from LDS feeds two MFMA chains; the four MFMAs of a chain are issued between the quarters of the
//    reduction of the previous chain.
#include <atomic>
#include <type_traits>
#include "engine.hpp"

namespace mx {

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));
typedef unsigned long long u64;

constexpr int BIG = 0x7fffffff;
constexpr int NONE_H = 0x3fffff;           // row constant of a padding row: t = NONE_H + 0
constexpr int NONE_KEY = NONE_H << 9;      // 0x7ffffe00: empty slot of a running minimum; every real key is smaller, every padding key larger
constexpr int TPS = 4;                     // train tiles staged per barrier
constexpr int CHUNK = 240;                 // tiles per index chunk (absolute tile numbers): the low byte of a key is tile % CHUNK + 1
constexpr int MINT = 12;                   // fewest tiles a split is made of
#ifndef MODSX_KTOP
#define MODSX_KTOP 4
#endif
constexpr int KTOP = MODSX_KTOP;           // group keys a stream keeps (>= 4: NN0 and NN1 are then always certain in k_match_decide)
constexpr int PB = 256;                    // trains per workgroup of k_match_pack
#ifndef MODSX_PACK_POLLS
#define MODSX_PACK_POLLS 256
#endif
constexpr int PACK_POLLS = MODSX_PACK_POLLS;   // looks at a predecessor's status word before its trains are counted here instead
// 32-query sets per wave (QS, even): 2 for most problems -- 3 wavefronts per SIMD --, 4 when both sides hold >= 40 k descriptors:
// every LDS fragment read then feeds four MFMA chains (half the LDS bytes per matrix instruction) at 2 wavefronts per SIMD
constexpr int sweep_wps(int qs) { return qs >= 4 ? 2 : 3; }   // waves per SIMD the sweeps are built for
constexpr int qpb_of(int qs) { return 4 * 32 * qs; }          // queries per 256-thread workgroup (k_match_resolve)
// Sweep 1 has two shapes.  THIN: 256-thread workgroups, three (QS = 2) or two (QS = 4) per CU.  FAT: ONE workgroup per CU that holds
// every wavefront the CU is meant to carry (12 / 8) and stages four groups of tiles per barrier.  With thin workgroups the SIMDs serve
// the oldest wavefront first -- the first workgroup of a CU leaves at 37 us, the last at 64, alone (tools/trace_sweep_phases.py) --
// while wavefronts that meet at the same barriers advance and leave together, and the staged tiles are fetched once per CU instead
// of three times: -3.7 % at 24 k x 24 k and 48 k x 47 k (62.9 against 65.4 us, 225 against 233); at 10 k x 10 k, where a launch is
// latency and not work, the fat shape has too few workgroups to hide behind and loses 1.2 us -- so it is taken from 16 k queries on.
constexpr int s1_waves(int qs, bool fat) { return fat ? 4 * sweep_wps(qs) : 4; }      // wavefronts per workgroup of k_match_sweep1
constexpr int s1_qpb(int qs, bool fat) { return s1_waves(qs, fat) * 32 * qs; }         // its queries
constexpr int s1_round(int qs, bool fat) { return fat ? 256 : 256 * sweep_wps(qs); }   // workgroups of one round
constexpr int s1_spb(bool fat) { return fat ? 4 : 1; }      // groups of 4 tiles a workgroup stages per barrier (2 x SPB x 16.5 KB of LDS)
static bool match_fat(int n1) {
  static const int forced = getenv("MODSX_SWEEP1_FAT") ? atoi(getenv("MODSX_SWEEP1_FAT")) : -1;    // 0 / 1: one shape for every size
  return forced >= 0 ? forced != 0 : n1 >= 16000;
}
static int match_qsets(int nb, int n1, int n2) {
  static const int forced = getenv("MODSX_MATCH_QSETS") ? atoi(getenv("MODSX_MATCH_QSETS")) : 0;
  if (forced == 2 || forced == 4) return forced;
  return (nb == 1 && n1 >= 40000 && n2 >= 40000) ? 4 : 2;
}
constexpr int TILE_B = 4096;
constexpr int HOFF = TPS * TILE_B, STAGE_B = HOFF + TPS * 128;
constexpr int MAXD = 128 * 255 * 255;      // largest possible squared distance

// Tile geometry of a problem.  Host side: the capacity of one class region (either class may hold every train) and an upper
// bound of the virtual tile count.  Device side (written by k_match_pack's last workgroup): the padded tile counts.
MX_HD int region_tiles(int n2) { return (((n2 + 31) / 32 + TPS - 1) & ~(TPS - 1)) + TPS; }
MX_HD int ntiles_ub(int n2) { return (((n2 + 31) / 32 + TPS - 1) & ~(TPS - 1)) + 2 * TPS; }
struct TileGeo { int TEp, TOp, ntilesV, pad; };   // even / odd class tiles (multiples of TPS), their sum
MX_D int phys_tile(int v, int TEp, int offT) { return v < TEp ? v : offT + v - TEp; }

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
MX_D u64 key64(int d, int lo) { return ((u64)(unsigned)d << 32) | (unsigned)lo; }   // d >= 0
MX_D u64 shfl_xor64(u64 v, int m) {
  const int lo = __shfl_xor((int)(unsigned)v, m), hi = __shfl_xor((int)(v >> 32), m);
  return ((u64)(unsigned)hi << 32) | (unsigned)lo;
}

// minimum over the 16 lanes of a DPP row, in every lane (row rotations: no LDS crossbar, one instruction per step)
MX_D int rowmin16(int v) {
  v = min(v, __builtin_amdgcn_update_dpp(v, v, 0x128, 0xf, 0xf, false));
  v = min(v, __builtin_amdgcn_update_dpp(v, v, 0x124, 0xf, 0xf, false));
  v = min(v, __builtin_amdgcn_update_dpp(v, v, 0x122, 0xf, 0xf, false));
  v = min(v, __builtin_amdgcn_update_dpp(v, v, 0x121, 0xf, 0xf, false));
  return v;
}

struct MatchGeom {
  int n1, n2, S, tilesPerSplit, qs, ntilesUB, offT;   // offT: first tile of the odd class region
};

// Sweep 2 runs over the UNDECIDED queries only, whose number the host does not know at launch time.  Every workgroup of a fixed
// one-round launch therefore derives the split geometry from the device-side count: the NW workgroups are dealt out as
// (query block, split) with as many splits as fill the machine once.
// workgroups of k_match_resolve that hold a (query block, split): one per CU while the undecided queries fill at most two
// blocks (the usual 1-3 %: more workgroups would only wait for LDS), a full round of the sweeps' size beyond that (inputs with
// many near-duplicates per query: shorter splits, fewer logged groups per stream)
MX_HD int resolve_nw(int nQB, int qs) { return nQB <= 2 ? 256 : 256 * sweep_wps(qs); }
struct Sweep2Geom { int nQB, S, tilesPerSplit; };
MX_HD Sweep2Geom sweep2_geom(int nUnd, int ntiles, int qs) {
  Sweep2Geom G;
  const int QPB = qpb_of(qs);
  G.nQB = (nUnd + QPB - 1) / QPB;
  const int SWEEP2_NW = resolve_nw(G.nQB, qs);
  int S = G.nQB > 0 ? SWEEP2_NW / G.nQB : 1;
  if (S > ntiles / MINT) S = ntiles / MINT;
  if (S < 1) S = 1;
  int tps = (ntiles + S - 1) / S;
  tps = (tps + TPS - 1) & ~(TPS - 1);
  S = (ntiles + tps - 1) / tps;
  G.S = S < 1 ? 1 : S;
  G.tilesPerSplit = tps;
  return G;
}
// register r of the 32x32 accumulator of lane half `hi` holds MFMA row 8 (r >> 2) + 4 hi + (r & 3)
MX_D int row_of(int r, int hi) { return 8 * (r >> 2) + 4 * hi + (r & 3); }

// ---------------- pack: norms, parity classes, swizzled tiles, row / tile constants ---------------------------------------
// y = 0: norm1[i] = |q_i - 128|^2, one thread per query.
// y = 1: one 1024-thread workgroup per block of PB trains; thread i takes the trains i and i + 1024 of the block.
//   slot of a train = block base + its rank among the block's trains of its class (even classes first, each class padded
//   to whole tiles);  tiles[slot >> 5] row slot & 31 = b - 128 in 16-byte slots, slot s stored at s ^ ((row >> 1) & 7)
//   hrow[slot] = (|b'|^2 + 2 sum b') >> 1,  norm2[slot] = |b'|^2,  perm[slot] = train index;  padding: zero row, NONE_H, -1
//   tk[tile] = parity << 8 | (tile % CHUNK + 1)
struct PackArgs {
  const uint8_t *d1, *d2;
  const double *pos2;
  int n1, n2, offT;
  unsigned epoch;
  int *norm1, *norm2, *hrow, *perm;
  double2 *pos2p;
  unsigned char *tiles;
  u64 *status;            // one word per pack workgroup: epoch << 32 | even trains << 16 | odd trains
  TileGeo *geo;
};
// sum over the 8 lanes of a half DPP row (half-row mirror, quad reverse, quad pair swap): every lane ends with the total
MX_D int sum8(int v) {
  v += __builtin_amdgcn_update_dpp(0, v, 0x141, 0xf, 0xf, false);
  v += __builtin_amdgcn_update_dpp(0, v, 0x1B, 0xf, 0xf, false);
  v += __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xf, 0xf, false);
  return v;
}
// y = 0: norm1[i] = |q_i - 128|^2, eight lanes per query (coalesced 128-byte rows).
// y = 1: one 256-thread workgroup per PB = 256 trains, eight lanes per train (one 16-byte slice each), 32 trains per pass:
//   row = pass * 32 + tid / 8 is the train's place in the workgroup, so (pass, wave, group in the wave) is train order.
//   slot of a train = its rank among ALL trains of its class: the workgroup's base (sum of the counts of the workgroups
//   before it -- they were dispatched before it, so waiting for their status words cannot deadlock) + its rank inside.
//   tiles[slot >> 5] row slot & 31 = b - 128 in 16-byte slots, slot s stored at s ^ ((row >> 1) & 7); the odd class starts
//   at tile offT.  hrow[slot] = (|b'|^2 + 2 sum b') >> 1, norm2[slot] = |b'|^2, perm[slot] = train, pos2p[slot] = position.
//   The last workgroup knows the totals: it writes the tile geometry and the padding rows (zero row, NONE_H, -1).
__device__ __forceinline__ void pack_body(const PackArgs &A) {
  const int tid = threadIdx.x;
  if (blockIdx.y == 0) {
    const int i = blockIdx.x * 32 + (tid >> 3);
    int s = 0;
    if (i < A.n1) {
      const v4i v = reinterpret_cast<const v4i *>(A.d1 + (size_t)i * 128)[tid & 7];
#pragma unroll
      for (int w = 0; w < 4; w++) { const int x = v[w] ^ 0x80808080; s = __builtin_amdgcn_sdot4(x, x, s, false); }
    }
    s = sum8(s);
    if (i < A.n1 && (tid & 7) == 0) A.norm1[i] = s;
    return;
  }
  const int nwg = (A.n2 + PB - 1) / PB;
  const int blk = blockIdx.x;
  if (blk >= nwg) return;
  __shared__ int sCnt[2][32];      // [class][pass * 4 + wave]: trains of the class, then their exclusive prefix
  __shared__ int sTot[2], sBase[2][4], sMiss[256], sNmiss;
  __shared__ int sSlot[PB], sNv[PB], sHv[PB];   // per train of the workgroup: slot, |b'|^2, row constant
  const int wave = tid >> 6, lane = tid & 63, slice = tid & 7, gw = lane >> 3;
  constexpr int NP = PB / 32;
  if (tid == 0) sNmiss = 0;
  v4i row[NP];
  int meta[NP];                    // valid | parity << 1 | rank among the wave's trains of the class << 2
#pragma unroll
  for (int p = 0; p < NP; p++) {   // all loads first: eight independent 16-byte loads in flight per lane
    const int t = blk * PB + p * 32 + (tid >> 3);
    row[p] = (v4i){(int)0x80808080, (int)0x80808080, (int)0x80808080, (int)0x80808080};
    if (t < A.n2) row[p] = reinterpret_cast<const v4i *>(A.d2 + (size_t)t * 128)[slice];
  }
#pragma unroll
  for (int p = 0; p < NP; p++) {
    const int t = blk * PB + p * 32 + (tid >> 3);
    const bool valid = t < A.n2;
    v4i v = row[p];
    int s = 0, lin = 0;
#pragma unroll
    for (int w = 0; w < 4; w++) {
      v[w] ^= 0x80808080;
      s = __builtin_amdgcn_sdot4(v[w], v[w], s, false);
      lin = __builtin_amdgcn_sdot4(v[w], 0x01010101, lin, false);
    }
    row[p] = v;
    s = sum8(s); lin = sum8(lin);
    const int par = lin & 1;
    const u64 balE = __ballot(valid && !par && slice == 0), balO = __ballot(valid && par && slice == 0);
    const u64 below = (1ull << (gw * 8)) - 1;
    const int rank = par ? __popcll(balO & below) : __popcll(balE & below);
    meta[p] = (valid ? 1 : 0) | (par << 1) | (rank << 2);
    if (slice == 0) { sNv[p * 32 + (tid >> 3)] = s; sHv[p * 32 + (tid >> 3)] = (s + 2 * lin) >> 1; }
    if (lane == 0) { sCnt[0][p * 4 + wave] = __popcll(balE); sCnt[1][p * 4 + wave] = __popcll(balO); }
  }
  __syncthreads();
  if (wave < 2) {
    // exclusive prefix of the 32 counts of class `wave`
    const int c = lane < 32 ? sCnt[wave][lane] : 0;
    int pre = c;
#pragma unroll
    for (int m = 1; m < 32; m <<= 1) { const int v = __shfl_up(pre, m); if (lane >= m) pre += v; }
    if (lane < 32) sCnt[wave][lane] = pre - c;
    if (lane == 31) sTot[wave] = pre;
  }
  __syncthreads();
  const int myE = sTot[0], myO = sTot[1];
  if (tid == 0)
    __hip_atomic_store(A.status + blk, ((u64)A.epoch << 32) | ((u64)myE << 16) | (u64)myO, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  // the counts of the workgroups before this one (a word is valid when it carries this launch's epoch and sums to the
  // workgroup's train count -- whatever an earlier launch or another use of the memory left there does not)
  // The wait is BOUNDED: a predecessor whose word has not come after PACK_POLLS looks (~0.2 ms; it normally comes within a few
  // microseconds) is counted by this workgroup itself, from its trains -- so the kernel ends whatever the order in which the
  // hardware starts workgroups and whatever else holds the CUs (the order argument above needs per-queue in-order dispatch; with
  // many streams packing at once, workgroups that spin could in principle hold every slot a late predecessor needs).
  int bE = 0, bO = 0;
  for (int j = tid; j < blk; j += 256) {
    u64 w = 0;
    bool ok = false;
    for (int poll = 0; poll < PACK_POLLS && !ok; poll++) {
      w = __hip_atomic_load(A.status + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      ok = (unsigned)(w >> 32) == A.epoch && (int)((w >> 16) & 0xffff) + (int)(w & 0xffff) == PB;
      if (!ok) __builtin_amdgcn_s_sleep(2);
    }
    if (ok) { bE += (int)((w >> 16) & 0xffff); bO += (int)(w & 0xffff); }
    else sMiss[atomicAdd(&sNmiss, 1) & 255] = j;       // (more than 256 at once: the list wraps; handled below)
  }
  __syncthreads();
  {
    int nmiss = sNmiss;
    const bool all = nmiss > 256;                      // the list wrapped (hundreds of predecessors silent at once): count EVERY block before
    if (all) { bE = 0; bO = 0; nmiss = blk; }          // this one here and use none of the words
    for (int mi = 0; mi < nmiss; mi++) {               // count block j ourselves: eight lanes per train, 32 trains per pass
      const int j = all ? mi : sMiss[mi];
      int odd = 0;
      for (int p = 0; p < NP; p++) {
        const v4i v = reinterpret_cast<const v4i *>(A.d2 + (size_t)(j * PB + p * 32 + (tid >> 3)) * 128)[slice];   // j < blk: a full block
        int lin = 0;
#pragma unroll
        for (int w4 = 0; w4 < 4; w4++) lin = __builtin_amdgcn_sdot4(v[w4] ^ 0x80808080, 0x01010101, lin, false);
        lin = sum8(lin);
        if (slice == 0) odd += lin & 1;
      }
      bO += odd; bE += slice == 0 ? NP - odd : 0;      // each train is counted once, by the lane with slice 0
    }
  }
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) { bE += __shfl_xor(bE, m); bO += __shfl_xor(bO, m); }
  if (lane == 0) { sBase[0][wave] = bE; sBase[1][wave] = bO; }
  __syncthreads();
  const int baseE = sBase[0][0] + sBase[0][1] + sBase[0][2] + sBase[0][3];
  const int baseO = sBase[1][0] + sBase[1][1] + sBase[1][2] + sBase[1][3];
  const int offS = A.offT * 32;
#pragma unroll
  for (int p = 0; p < NP; p++) {
    if (!(meta[p] & 1)) continue;
    const int par = (meta[p] >> 1) & 1, rank = meta[p] >> 2;
    const int slot = (par ? offS + baseO : baseE) + sCnt[par][p * 4 + wave] + rank;
    const int tile = slot >> 5, r = slot & 31, sw = (r >> 1) & 7;
    *reinterpret_cast<v4i *>(A.tiles + (size_t)tile * TILE_B + r * 128 + ((slice ^ sw) << 4)) = row[p];
    if (slice == 0) sSlot[p * 32 + (tid >> 3)] = slot;
  }
  __syncthreads();
  {
    // per-slot constants and positions (k_match_decide / k_match_resolve read the positions with the rows), a thread per train
    const int t = blk * PB + tid;
    if (t < A.n2) {
      const int slot = sSlot[tid];
      A.hrow[slot] = sHv[tid]; A.norm2[slot] = sNv[tid]; A.perm[slot] = t;
      A.pos2p[slot] = reinterpret_cast<const double2 *>(A.pos2)[t];
    }
  }
  if (blk == nwg - 1) {
    const int totE = baseE + myE, totO = baseO + myO;
    const int TEp = (((totE + 31) >> 5) + TPS - 1) & ~(TPS - 1), TOp = (((totO + 31) >> 5) + TPS - 1) & ~(TPS - 1);
    if (tid == 0) { TileGeo G; G.TEp = TEp; G.TOp = TOp; G.ntilesV = TEp + TOp; G.pad = 0; *A.geo = G; }
    const int padE = TEp * 32 - totE, npad = padE + TOp * 32 - totO;     // at most 2 * (TPS * 32 - 1) slots
    for (int k = tid >> 3; k < npad; k += 32) {
      const int slot = k < padE ? totE + k : offS + totO + (k - padE);
      *reinterpret_cast<v4i *>(A.tiles + (size_t)(slot >> 5) * TILE_B + (slot & 31) * 128 + (slice << 4)) = (v4i){0, 0, 0, 0};
      if (slice == 0) { A.hrow[slot] = NONE_H; A.norm2[slot] = 0; A.perm[slot] = -1; }
      if (slice == 1) A.pos2p[slot] = make_double2(0.0, 0.0);
    }
  }
}

// Exact distances of one group (the 16 rows of tile `vtile` that lane half `hi` of the sweeps owns; `ptile` is where the tile
// lives) from query `qd`, by the 16 lanes l = 0..15 of a quarter wave: lane l returns the distance of row row_of(l, hi) (BIG for a
// padding row), its VIRTUAL slot vtile * 32 + row (the order of the walk), its physical slot and its train index.
// Loads are coalesced: in step k the lanes l < 8 read the eight 16-byte slices of row k, the lanes l >= 8 those of row k + 8
// (two whole 128-byte rows per step), partial dot products are summed over the eight lanes of a row.
MX_D int group_dist16(const uint8_t *qd, int na, const unsigned char *tiles, const int *norm2, const int *perm, int ptile, int vtile,
                      int hi, int l, int *vslot_out, int *pslot_out, int *t_out) {
  const int slice = l & 7, half = l >> 3;
  v4i q = reinterpret_cast<const v4i *>(qd)[slice];
  q[0] ^= 0x80808080; q[1] ^= 0x80808080; q[2] ^= 0x80808080; q[3] ^= 0x80808080;
  int part[8];
  const unsigned char *tb = tiles + (size_t)ptile * TILE_B;
#pragma unroll
  for (int k = 0; k < 8; k++) {
    const int r = row_of(k + 8 * half, hi);
    const v4i y = *reinterpret_cast<const v4i *>(tb + r * 128 + ((slice ^ ((r >> 1) & 7)) << 4));
    int dot = 0;
#pragma unroll
    for (int c = 0; c < 4; c++) dot = __builtin_amdgcn_sdot4(q[c], y[c], dot, false);
    part[k] = dot;
  }
#pragma unroll
  for (int k = 0; k < 8; k++) part[k] = sum8(part[k]);
  int dot = part[0];
#pragma unroll
  for (int k = 1; k < 8; k++) dot = slice == k ? part[k] : dot;
  const int row = row_of(l, hi), pslot = ptile * 32 + row;
  *vslot_out = vtile * 32 + row;
  *pslot_out = pslot;
  const int t = perm[pslot];
  *t_out = t;
  return t >= 0 ? na + norm2[pslot] - 2 * dot : BIG;
}

// ---------------- staging: 4 tiles + their constants, global -> LDS directly ------------------------------------------
typedef const unsigned char __attribute__((address_space(1))) *gbptr;
typedef unsigned char __attribute__((address_space(3))) *lbptr;
template <int NW>      // wavefronts of the workgroup.  16 chunks of tile bytes (1 KB per wave instruction) + 2 of row constants
MX_D void stage_group(const unsigned char *tiles, const int *hrow, int p0, unsigned char *buf, int wave, int lane) {
  // a producing wavefront takes CPW CONSECUTIVE chunks: one address pair and one M0 for all of them, the instruction's immediate
  // offset (applied to the global and to the LDS address alike) steps through them
  constexpr int CPW = NW >= 8 ? 2 : 4, NPROD = 16 / CPW;
  if (wave < NPROD) {
    const unsigned char *src = tiles + (size_t)p0 * TILE_B + (size_t)wave * (CPW * 1024) + lane * 16;
    unsigned char *dst = buf + wave * (CPW * 1024);
    __builtin_amdgcn_global_load_lds((gbptr)src, (lbptr)dst, 16, 0, 0);
    __builtin_amdgcn_global_load_lds((gbptr)src, (lbptr)dst, 16, 1024, 0);
    if (CPW == 4) {
      __builtin_amdgcn_global_load_lds((gbptr)src, (lbptr)dst, 16, 2048, 0);
      __builtin_amdgcn_global_load_lds((gbptr)src, (lbptr)dst, 16, 3072, 0);
    }
  }
  const int k = NW - 1 - wave;        // the last two wavefronts bring the row constants
  if (k < 2)
    __builtin_amdgcn_global_load_lds((gbptr)(reinterpret_cast<const unsigned char *>(hrow + (size_t)p0 * 32) + k * 256 + lane * 4),
                                     (lbptr)(buf + HOFF + k * 256), 4, 0, 0);
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
MX_D int tree_min16(const v16i &k) {
  const int t0 = imin3(k[0], k[1], k[2]), t1 = imin3(k[3], k[4], k[5]), t2 = imin3(k[6], k[7], k[8]);
  const int t3 = imin3(k[9], k[10], k[11]), t4 = imin3(k[12], k[13], k[14]);
  return min(imin3(t0, t1, t2), imin3(t3, t4, k[15]));
}

// LDS reads of the sweep core as assembly (the waits are counted by hand there): the four fragment slices of tile Q of a stage,
// and its 16 row constants
template <int Q>
MX_D void lds_load_af(unsigned base, const unsigned (&aAddr)[4], v4i *af) {
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(af[0]) : "v"(base + aAddr[0]), "n"(Q * TILE_B));
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(af[1]) : "v"(base + aAddr[1]), "n"(Q * TILE_B));
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(af[2]) : "v"(base + aAddr[2]), "n"(Q * TILE_B));
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(af[3]) : "v"(base + aAddr[3]), "n"(Q * TILE_B));
}
template <int Q>
MX_D void lds_load_c(unsigned addr, v16i &C) {
  v4i c0, c1, c2, c3;
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(c0) : "v"(addr), "n"(Q * 128));
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(c1) : "v"(addr), "n"(Q * 128 + 32));
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(c2) : "v"(addr), "n"(Q * 128 + 64));
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(c3) : "v"(addr), "n"(Q * 128 + 96));
  C = __builtin_shufflevector(__builtin_shufflevector(c0, c1, 0, 1, 2, 3, 4, 5, 6, 7), __builtin_shufflevector(c2, c3, 0, 1, 2, 3, 4, 5, 6, 7),
                              0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15);
}

// ---------------- the sweep core: tiles through LDS, MFMA chains, an epilogue per chain ---------------------------------------
// One instruction stream per wave keeps both pipes busy: while the four MFMAs of a (tile, query set) chain run, the wave
// reduces the accumulators of the previous chain, so the matrix pipe never waits for a whole wave to leave its epilogue.
// Fragments and row constants of the next tile are read from LDS one tile ahead (two register sets).  The epilogue is a policy:
//   int  kv(v)                          the wave-uniform constant of virtual tile v
//   void chain(acc, kv, s, tile)        reduce one accumulator (query set s, virtual tile `tile`)
//   void flush(chunkTile0)              end of an index chunk (CH tiles, absolute tile numbers)
#ifdef SWEEP_PHASE_TRACE
// debugging aid (tools/trace_sweep_phases.py; tools/build_variant.sh ptrace "-DSWEEP_PHASE_TRACE"): per wavefront of the last
// k_match_sweep1 launch, shader-clock cycles spent waiting at the stage barriers / issuing the next stage's DMA / in the tiles, its
// total, stages, and the 100 MHz wall clock at its start and end.  (A stamp is an s_memtime + s_waitcnt: a few hundred cycles each.)
__device__ unsigned long long g_ptrace[16384][8];
#define PTRACE(x) x
#else
#define PTRACE(x)
#endif
// SPB: groups of TPS tiles per barrier.  A workgroup that is alone on its CU (sweep 1) has nobody to cover the bubble at a
// barrier -- every wavefront refills its pipeline at the same moment --, so it stages SPB groups at once and meets 1 / SPB as often.
template <int QSETS, int EPI_VALU, int NW, int SPB, class Epi>
__device__ __forceinline__ void sweep_core(const unsigned char *tiles, const int *hrow, int TEp, int offT, int tBeg, int tEnd,
                                           const v4i (&bq)[QSETS][4], unsigned char (&sm)[2 * SPB][STAGE_B], Epi &epi) {
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int col = lane & 31, hi = lane >> 5;
  // The LDS reads are written as assembly with the waits counted by hand: the compiler orders every wait behind the reads
  // of the NEXT tile it has just issued (`s_waitcnt lgkmcnt(0)` -- a pending global->LDS load makes it give up counting), which
  // exposed one LDS latency per tile.  Reads issue and return in order, so "all but the newest 4" is exactly "everything of
  // the current tile".  Per-lane addresses: fragment slot 2 kb + hi of row col (swizzled), row constants 16 hi.
  unsigned aAddr[4];
#pragma unroll
  for (int kb = 0; kb < 4; kb++) aAddr[kb] = (unsigned)(col * 128 + (((2 * kb + hi) ^ ((col >> 1) & 7)) << 4));
  const unsigned cAddr = (unsigned)(HOFF + 16 * hi);
  const unsigned smBase = (unsigned)(size_t)(lbptr)&sm[0][0];
  v4i af[2][4];
  v16i C[2];
  int kv[2];
  v16i acc[2];
  // the pipeline starts with a neutral pending chain: keys that change nothing
#pragma unroll
  for (int r = 0; r < 16; r++) acc[1][r] = NONE_H;
  int kvPend = 0, pendTile = 0;
#pragma unroll
  for (int j = 0; j < SPB; j++)
    if (tBeg + j * TPS < tEnd) stage_group<NW>(tiles, hrow, phys_tile(tBeg + j * TPS, TEp, offT), sm[j], wave, lane);
  int it = 0;
  PTRACE(unsigned long long pa0 = 0; unsigned long long pa1 = 0; unsigned long long pa2 = 0; unsigned long long pt0 = __builtin_readcyclecounter(); const unsigned long long pstart = pt0; const unsigned long long pwall = wall_clock64();)
  for (int tb = tBeg; tb < tEnd; tb += TPS * SPB, it++) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    PTRACE(const unsigned long long pt1 = __builtin_readcyclecounter();)
#pragma unroll
    for (int j = 0; j < SPB; j++) {
      const int t = tb + TPS * SPB + j * TPS;
      if (t < tEnd) stage_group<NW>(tiles, hrow, phys_tile(t, TEp, offT), sm[((it & 1) ^ 1) * SPB + j], wave, lane);
    }
    PTRACE(const unsigned long long pt2 = __builtin_readcyclecounter(); pa0 += pt1 - pt0; pa1 += pt2 - pt1;)
   for (int j = 0; j < SPB; j++) {
    const int tg = tb + j * TPS;
    if (tg >= tEnd) break;
    const unsigned base = smBase + ((it & 1) * SPB + j) * STAGE_B;
    lds_load_af<0>(base, aAddr, af[0]);
    lds_load_c<0>(base + cAddr, C[0]);
    kv[0] = epi.kv(tg);
    // one tile: phase s = the chain of (tile q, set s) beside the reduction of the previous chain -- (tile q, set s - 1), or the
    // last set of the previous tile.  QSETS is even, so the chains alternate between the two accumulators.  The fragment reads
    // of the NEXT tile are issued in front of phase 0, its row constants in front of phase 1.
    auto tile = [&](auto qc) {
      constexpr int q = decltype(qc)::value, cur = q & 1;
#pragma unroll
      for (int s = 0; s < QSETS; s++) {
        if (q + 1 < TPS) {
          if (s == 0) lds_load_af<(q + 1) % TPS>(base, aAddr, af[cur ^ 1]);
          if (s == 1) { lds_load_c<(q + 1) % TPS>(base + cAddr, C[cur ^ 1]); kv[cur ^ 1] = epi.kv(tg + q + 1); }
        }
        if (s == 0) {
          // everything of THIS tile has landed once at most the reads just issued (4, none in the last tile of a stage) are pending
          if (q + 1 < TPS)
            asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(af[cur][0]), "+v"(af[cur][1]), "+v"(af[cur][2]), "+v"(af[cur][3]), "+v"(C[cur]));
          else
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(af[cur][0]), "+v"(af[cur][1]), "+v"(af[cur][2]), "+v"(af[cur][3]), "+v"(C[cur]));
        }
        __builtin_amdgcn_sched_barrier(0);
        v16i &an = acc[s & 1];
        an = __builtin_amdgcn_mfma_i32_32x32x32_i8(af[cur][0], bq[s][0], C[cur], 0, 0, 0);
#pragma unroll
        for (int kb = 1; kb < 4; kb++) an = __builtin_amdgcn_mfma_i32_32x32x32_i8(af[cur][kb], bq[s][kb], an, 0, 0, 0);
        if (s == 0) epi.chain(acc[1], kvPend, QSETS - 1, pendTile);
        else epi.chain(acc[(s - 1) & 1], kv[cur], s - 1, tg + q);
        // one MFMA, then a quarter of the reduction
        constexpr int Q1 = (EPI_VALU + 3) / 4, Q2 = (EPI_VALU + 2) / 4, Q3 = (EPI_VALU + 1) / 4, Q4 = EPI_VALU / 4;
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, Q1, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, Q2, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, Q3, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, Q4, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
      pendTile = tg + q;
      kvPend = kv[cur];
    };
    tile(std::integral_constant<int, 0>{});
    tile(std::integral_constant<int, 1>{});
    tile(std::integral_constant<int, 2>{});
    tile(std::integral_constant<int, 3>{});
    static_assert(TPS == 4, "four tiles per stage");
    if ((tg + TPS) % Epi::CH == 0) {
      // end of an index chunk: drain the pending chain, then move the indices of new keys out of the low byte
      epi.chain(acc[1], kvPend, QSETS - 1, pendTile);
#pragma unroll
      for (int r = 0; r < 16; r++) acc[1][r] = NONE_H;
      kvPend = 0;
      epi.flush(tg + TPS - Epi::CH);
    }
   }
   PTRACE({ const unsigned long long pt3 = __builtin_readcyclecounter(); pa2 += pt3 - pt2; pt0 = pt3; })
  }
  if (tEnd > tBeg && tEnd % Epi::CH) {
    epi.chain(acc[1], kvPend, QSETS - 1, pendTile);
    epi.flush((tEnd / Epi::CH) * Epi::CH);
  }
#ifdef SWEEP_PHASE_TRACE
  if (Epi::CH == CHUNK && lane == 0) {       // sweep 1 only (k_match_resolve runs the same core)
    const int wg = blockIdx.x + gridDim.x * blockIdx.y;
    if (wg * NW + wave < 16384) {
      unsigned long long *o = g_ptrace[wg * NW + wave];
      o[0] = pa0; o[1] = pa1; o[2] = pa2; o[3] = __builtin_readcyclecounter() - pstart; o[4] = (unsigned long long)it * SPB; o[5] = wall_clock64();
      o[6] = (unsigned long long)QSETS; o[7] = pwall;
    }
  }
#endif
}

// ---------------- sweep 1: per (query, split, lane half) the KTOP smallest group keys ------------------------------------------
struct SweepArgs {
  const uint8_t *d1;
  const int *norm1;
  const unsigned char *tiles;
  const int *hrow;
  const TileGeo *geo;
  MatchGeom g;
  int2 *partial;          // [(q * S + split) * 2 + half][KTOP] (distance, virtual tile), ascending; empty = (BIG, -1)
};
// 8 (v_min3 tree) + 1 (key) + KTOP (insertion) vector instructions per chain
template <int QSETS>
struct TopKEpi {
  static constexpr int CH = CHUNK;
  int m[QSETS][KTOP], I[QSETS][KTOP];
  int TEp;
  MX_D int kv(int v) const { return ((v >= TEp ? 1 : 0) << 8) | (v % CHUNK + 1); }   // parity << 8 | tile code
  MX_D void chain(const v16i &acc, int kvv, int s, int) {
    const int key = (tree_min16(acc) << 9) + kvv;
#pragma unroll
    for (int k = KTOP - 1; k >= 1; k--) m[s][k] = imed3(m[s][k - 1], m[s][k], key);
    m[s][0] = min(m[s][0], key);
  }
  // the tile numbers of keys that are new in this chunk move to the index registers, the low byte is cleared; keys with a
  // cleared low byte are older entries and keep their order among themselves
  MX_D void flush(int chunkTile0) {
#pragma unroll
    for (int s = 0; s < QSETS; s++) {
      int o[KTOP];
#pragma unroll
      for (int k = 0; k < KTOP; k++) o[k] = I[s][k];
#pragma unroll
      for (int k = 0; k < KTOP; k++) {
        const int lb = m[s][k] & 255;
        I[s][k] = lb ? chunkTile0 + lb - 1 : o[0];
        if (!lb) {
#pragma unroll
          for (int j = 0; j + 1 < KTOP; j++) o[j] = o[j + 1];
        }
        m[s][k] &= ~255;
      }
    }
  }
};
template <int QSETS, bool FAT>
__device__ __forceinline__ void sweep_body(const SweepArgs &A) {
  constexpr int QPB = s1_qpb(QSETS, FAT);
  __shared__ __attribute__((aligned(16))) unsigned char sm[2 * s1_spb(FAT)][STAGE_B];
  const MatchGeom g = A.g;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int col = lane & 31, hi = lane >> 5;
  const int sp = blockIdx.y, qb = blockIdx.x, S = g.S;
  if (qb * QPB >= g.n1) return;
  const int q0 = qb * QPB + wave * (32 * QSETS);
  v4i bq[QSETS][4];
  TopKEpi<QSETS> epi;
  int qsel[QSETS];
#pragma unroll
  for (int s = 0; s < QSETS; s++) {
    qsel[s] = min(q0 + 32 * s + col, g.n1 - 1);
#pragma unroll
    for (int kb = 0; kb < 4; kb++) bq[s][kb] = load_q(A.d1, qsel[s], kb, hi);
#pragma unroll
    for (int k = 0; k < KTOP; k++) { epi.m[s][k] = NONE_KEY; epi.I[s][k] = -1; }
  }
  epi.TEp = A.geo->TEp;
  const int tBeg = sp * g.tilesPerSplit, tEnd = min(tBeg + g.tilesPerSplit, A.geo->ntilesV);   // virtual tiles: even class, then odd
  sweep_core<QSETS, 9 + KTOP, s1_waves(QSETS, FAT), s1_spb(FAT)>(A.tiles, A.hrow, epi.TEp, g.offT, tBeg, tEnd, bq, sm, epi);
  // store the stream's keys as (distance, tile)
#pragma unroll
  for (int s = 0; s < QSETS; s++) {
    const int q = q0 + 32 * s + col;
    if (q >= g.n1) continue;
    const int na = A.norm1[qsel[s]];
    int2 e[KTOP];
#pragma unroll
    for (int k = 0; k < KTOP; k++) e[k] = epi.m[s][k] >= NONE_KEY ? make_int2(BIG, -1) : make_int2((epi.m[s][k] >> 8) + na, epi.I[s][k]);
    int4 *dst = reinterpret_cast<int4 *>(A.partial + (((size_t)q * S + sp) * 2 + hi) * KTOP);
#pragma unroll
    for (int k = 0; k < KTOP; k += 2) dst[k >> 1] = make_int4(e[k].x, e[k].y, e[k + 1].x, e[k + 1].y);
  }
}

// ---------------- decide: merge the streams, recompute groups exactly, walk the certain rows ------------------------------
// 16 lanes per query (a DPP row); see the header and tests/match_model.py.
#ifndef MODSX_DECIDE_Q
#define MODSX_DECIDE_Q 16
#endif
constexpr int DECIDE_Q = MODSX_DECIDE_Q;   // queries per workgroup of k_match_decide (16 lanes each)
struct UndRec { int q, na, t0, dm; double x0, y0; };
static_assert(sizeof(UndRec) == 32, "undecided record");
struct DecideArgs {
  const uint8_t *d1;
  const int *norm1, *norm2, *perm;
  const unsigned char *tiles;
  const TileGeo *geo;
  const int2 *partial;
  MatchGeom g;
  const double2 *pos2p;
  double sqminratio, contrDistSq;
  int nn;
  MatchRow *rows;
  int *nUndecided;
  UndRec *und;            // everything k_match_resolve needs of an undecided query, in one record
};
__device__ __forceinline__ void decide_body(const DecideArgs &A) {
  // undecided queries are compacted with ONE global atomic per workgroup (a counter word takes ~90 atomics per us)
  __shared__ int sCount, sBase;
  __shared__ UndRec sRec[DECIDE_Q];
  if (threadIdx.x == 0) sCount = 0;
  __syncthreads();
  const MatchGeom g = A.g;
  const int l = threadIdx.x & 15;
  const int q = blockIdx.x * DECIDE_Q + (threadIdx.x >> 4);
  const bool live = q < g.n1;
  const int qc = live ? q : g.n1 - 1;
  // 1. this lane's streams -> its KTOP smallest (distance, group) keys in ascending order; group = tile * 2 + half; empty = (BIG, BIG)
  int Ld[KTOP], Lg[KTOP];
  const int nst = 2 * g.S;
  {
    int4 p[KTOP / 2];
#pragma unroll
    for (int k2 = 0; k2 < KTOP / 2; k2++) p[k2] = make_int4(BIG, -1, BIG, -1);
    if (l < nst) {
      const int4 *src = reinterpret_cast<const int4 *>(A.partial + ((size_t)qc * nst + l) * KTOP);
#pragma unroll
      for (int k2 = 0; k2 < KTOP / 2; k2++) p[k2] = src[k2];
    }
#pragma unroll
    for (int k2 = 0; k2 < KTOP / 2; k2++) {
      Ld[2 * k2] = p[k2].y < 0 ? BIG : p[k2].x; Lg[2 * k2] = p[k2].y < 0 ? BIG : p[k2].y * 2 + (l & 1);
      Ld[2 * k2 + 1] = p[k2].w < 0 ? BIG : p[k2].z; Lg[2 * k2 + 1] = p[k2].w < 0 ? BIG : p[k2].w * 2 + (l & 1);
    }
  }
  for (int st = l + 16; st < nst; st += 16) {        // more than 16 streams (few queries, many splits): insertion
    const int2 *src = A.partial + ((size_t)qc * nst + st) * KTOP;
    for (int e = 0; e < KTOP; e++) {
      const int2 pe = src[e];
      if (pe.y < 0) break;
      int ed = pe.x, eg = pe.y * 2 + (st & 1);
#pragma unroll
      for (int k = 0; k < KTOP; k++)
        if (lex_less(ed, eg, Ld[k], Lg[k])) { const int td = Ld[k], tg = Lg[k]; Ld[k] = ed; Lg[k] = eg; ed = td; eg = tg; }
    }
  }
  // 2. the KTOP smallest of the query: KTOP rounds of (lexicographic minimum of the heads over the 16 lanes, the owner pops)
  int Gd[KTOP], Gg[KTOP];
#pragma unroll
  for (int k = 0; k < KTOP; k++) {
    const int dm = rowmin16(Ld[0]);
    const int gm = rowmin16(Ld[0] == dm ? Lg[0] : BIG);
    Gd[k] = dm; Gg[k] = gm;
    if (Ld[0] == dm && Lg[0] == gm) {
#pragma unroll
      for (int jj = 0; jj + 1 < KTOP; jj++) { Ld[jj] = Ld[jj + 1]; Lg[jj] = Lg[jj + 1]; }
      Ld[KTOP - 1] = BIG; Lg[KTOP - 1] = BIG;
    }
  }
  int nG = 0;
#pragma unroll
  for (int k = 0; k < KTOP; k++) nG += Gd[k] != BIG;
  const int usable = nG < KTOP ? nG : KTOP - 1;
  // 3. recomputed rows: lane l holds row l of every recomputed group -- distance, slot, train index and position, all loaded
  //    with the group, so that the walk below is arithmetic and lane exchanges only
  const uint8_t *qd = A.d1 + (size_t)qc * 128;
  const int na = A.norm1[qc];
  int pd[KTOP - 1], ps[KTOP - 1], pt[KTOP - 1];  // distance (BIG: padding / used), slot, train
  double px[KTOP - 1], py[KTOP - 1];
#pragma unroll
  for (int k = 0; k < KTOP - 1; k++) { pd[k] = BIG; ps[k] = BIG; pt[k] = -1; px[k] = 0; py[k] = 0; }
  int mrec = 0;
  const int TEp = A.geo->TEp;
  auto recompute = [&](int k) {                // static k: the pool lives in registers
    int pslot;
    pd[k] = group_dist16(qd, na, A.tiles, A.norm2, A.perm, phys_tile(Gg[k] >> 1, TEp, g.offT), Gg[k] >> 1, Gg[k] & 1, l, &ps[k],
                         &pslot, &pt[k]);
    const double2 xy = A.pos2p[pslot];
    px[k] = xy.x; py[k] = xy.y;
  };
  if (usable > 0) { recompute(0); mrec = 1; }
  if (usable > 1) { recompute(1); mrec = 2; }
  // 4. the walk over the certain rows
  int j = 0, res = 0;                          // res: 0 running, 1 accept, 2 reject / ran off the list, 3 k_match_resolve
  int d0 = BIG, t0 = -1, dd1 = BIG, t1 = -1, dj = BIG, tj = -1;
  double x0 = 0, y0 = 0;
  const int grp = threadIdx.x & 48;            // first lane of this query's 16 within the wave
  for (int guard = 0; guard < 16 * KTOP + 8 && res == 0; guard++) {
    // the smallest unused recomputed row: (distance, slot), first in this lane, then over the 16 lanes
    int ld = BIG, ls = BIG;
#pragma unroll
    for (int k = 0; k < KTOP - 1; k++) if (pd[k] != BIG && lex_less(pd[k], ps[k], ld, ls)) { ld = pd[k]; ls = ps[k]; }
    const int cd = rowmin16(ld);
    const int cs = rowmin16(ld == cd ? ls : BIG);
    // the first group not recomputed bounds what is certain (none left and none dropped: everything is)
    int bd = BIG, bg = BIG;
#pragma unroll
    for (int k = 0; k < KTOP; k++) if (k == mrec && k < nG) { bd = Gd[k]; bg = Gg[k]; }
    const bool certain = cd != BIG && (bd == BIG || cd < bd || (cd == bd && (cs >> 5) < (bg >> 1)));
    if (certain) {
      // the owner lane (register index of the row inside its group) hands over train and position; it marks the row used
      int ct = pt[0];
      double cx = px[0], cy = py[0];
#pragma unroll
      for (int k = 1; k < KTOP - 1; k++) if (pd[k] == cd && ps[k] == cs) { ct = pt[k]; cx = px[k]; cy = py[k]; }
#pragma unroll
      for (int k = 0; k < KTOP - 1; k++) if (pd[k] == cd && ps[k] == cs) pd[k] = BIG;
      const int rowc = cs & 31, owner = grp + (((rowc >> 3) << 2) | (rowc & 3));
      ct = __shfl(ct, owner); cx = __shfl(cx, owner); cy = __shfl(cy, owner);
      if (j == 0) { d0 = cd; t0 = ct; x0 = cx; y0 = cy; }
      else {
        if (j == 1) { dd1 = cd; t1 = ct; }
        const double dx = x0 - cx, dy = y0 - cy;
        const bool far = dx * dx + dy * dy > A.contrDistSq;
        if (A.sqminratio >= 1.0) {
          // "to get all points" (matching.cpp:397-428): the record is closed by the first contradictive neighbour or by the last one
          if (far || j == A.nn - 1) { res = 1; dj = cd; tj = ct; }
        } else if (ratio_pass((float)d0, (float)cd, A.sqminratio)) { res = 1; dj = cd; tj = ct; }
        else if (far) res = 2;                                    // first contradictive
        else if (j >= A.nn - 1) res = 2;                          // the walk looks at nn - 1 neighbours
      }
      j++;
    } else if (mrec < usable) {
#pragma unroll
      for (int k = 2; k < KTOP - 1; k++) if (k == mrec) recompute(k);
      mrec++;
    } else res = nG < KTOP ? 2 : 3;
  }
  if (res == 0) res = 3;                       // not reached
  if (live && l == 0) {
    MatchRow o;
    o.t0 = t0; o.t1 = t1; o.tj = -1; o.nless = 0; o.nbad = 0;
    o.d0 = (float)d0; o.d1 = (float)dd1; o.dj = 0.f;
    int dm = 0;
    if (res == 1) { o.tj = tj; o.dj = (float)dj; o.nless = A.sqminratio >= 1.0 ? 0 : j - 2; }
    else if (res == 2) o.nbad = t1 < 0 ? 0 : 1;
    else if (t0 >= 0) {
      dm = A.sqminratio >= 1.0 ? 0 : ratio_dmin(d0, A.sqminratio);      // all-points mode: k_match_pdf redoes the query's walk
      if (dm <= MAXD) {          // otherwise no distance can pass the ratio test: the walk ends without a match
        const int k = atomicAdd(&sCount, 1);
        UndRec r; r.q = q; r.na = na; r.t0 = t0; r.dm = dm; r.x0 = x0; r.y0 = y0;
        sRec[k] = r;
        // k_match_resolve adds to nless / nbad and takes the minimum of (dj, tj) over its splits
        o.nless = 0; o.nbad = 0; o.tj = -1; o.dj = __int_as_float(0x7fffffff);
      }
    }
    A.rows[q] = o;
  }
  __syncthreads();
  if (threadIdx.x == 0 && sCount) sBase = atomicAdd(A.nUndecided, sCount);
  __syncthreads();
  if ((int)threadIdx.x < sCount) A.und[sBase + threadIdx.x] = sRec[threadIdx.x];
}

// ---------------- resolve: the second sweep and its logged groups in ONE launch -------------------------------------------------------
// For the queries k_match_decide could not finish (about 1 % on multi-view descriptors).  A workgroup takes (block of
// undecided queries, split of the virtual tiles), geometry from the device-side count as above, and
//  1. sweeps its tiles with keys that carry the row, (2 t + p) << 8 | (tile % 12 + 1) << 4 | register (32 VALU per chain --
//     this kernel is about a hundredth of sweep 1's work): a group whose minimum is below Dmin is logged in LDS, the
//     others feed the lane's running minimum, which is NNj's exact (distance, slot) candidate of the stream;
//  2. recomputes its logged groups exactly, one per quarter wave, for nless / nbad / the candidates >= Dmin inside them (a
//     stream that ran out of slots is rescanned tile by tile); a query with nn or more groups here is not recomputed -- that
//     many groups mean nless > nn - 2 whatever they hold (matching.cpp:435-457) -- and gets nn added to its nless;
//  3. adds its counts to the query's row and takes the minimum of (dj, tj) with one 64-bit atomic: equal distances are in
//     one parity class, where train order is slot order, so the minimum over (distance, train) is the one over
//     (distance, slot).  No merge pass: the rows are complete when the launch ends.
struct ResolveArgs {
  const uint8_t *d1;
  const int *norm2, *perm, *hrow;
  const unsigned char *tiles;
  const TileGeo *geo;
  MatchGeom g;
  const double2 *pos2p;
  double contrDistSq;
  int nn;
  const UndRec *und;
  const int *nUndecided;
  MatchRow *rows;
};
constexpr int RCHUNK = 12;                 // tiles per index chunk of the resolve sweep (4 bits of tile code beside 4 of register)
// keys that carry the row: 16 (keys) + 8 (tree) + 2 vector instructions per chain, and the rare branch of a logged group
template <int QSETS, int EVS>
struct ResolveEpi {
  static constexpr int CH = RCHUNK;
  int m1[QSETS], I1[QSETS], thr[QSETS], nev[QSETS];
  int TEp, hi, stream0;                    // stream0: stream index of this lane's query set 0 (sets are 64 streams apart)
  unsigned short (*evt)[EVS];
  MX_D int kv(int v) const { return ((v >= TEp ? 1 : 0) << 8) | ((v % RCHUNK + 1) << 4); }
  MX_D void chain(const v16i &acc, int kvb, int s, int tile) {
    v16i k;
#pragma unroll
    for (int r = 0; r < 16; r++) k[r] = (acc[r] << 9) + (kvb + r);
    const int t = tree_min16(k);
    if (t < thr[s]) {
      if (nev[s] < EVS) evt[stream0 + 64 * s][nev[s]] = (unsigned short)tile;
      nev[s]++;
    } else m1[s] = min(m1[s], t);
  }
  MX_D void flush(int chunkTile0) {
#pragma unroll
    for (int s = 0; s < QSETS; s++) {
      const int lb = m1[s] & 255;
      if (lb) I1[s] = (chunkTile0 + (lb >> 4) - 1) * 32 + row_of(lb & 15, hi);
      m1[s] &= ~255;
    }
  }
};

template <int QSETS>
__device__ __forceinline__ void resolve_body(const ResolveArgs &A) {
  constexpr int QPB = qpb_of(QSETS);
  constexpr int EVS = 16;                              // slots per stream: 16 / 32 KB of LDS (QSETS 2 / 4)
  constexpr int LCAP = 2 * STAGE_B / 4;                // entries of the work list (it lives in the staging buffers)
  __shared__ __attribute__((aligned(16))) unsigned char sm[2][STAGE_B];
  __shared__ unsigned short sEvt[2 * QPB][EVS];        // logged groups (virtual tiles < 65536: the launcher's limit) per stream = (query in block) * 2 + half
  __shared__ UndRec sU[QPB];
  __shared__ int sEv[QPB], sNless[QPB], sNbad[QPB];
  __shared__ u64 sCand[QPB];                           // (distance, train)
  __shared__ int sNrec, sNover;
  __shared__ unsigned short sOver[2 * QPB];
  const MatchGeom g = A.g;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int col = lane & 31, hi = lane >> 5;
  const int nQ = *A.nUndecided;
  const Sweep2Geom G2 = sweep2_geom(nQ, g.ntilesUB, QSETS);
  const int S = G2.S, tilesPerSplit = G2.tilesPerSplit;
  const int qb = (int)blockIdx.x / S, sp = (int)blockIdx.x - qb * S;
  if (qb * QPB >= nQ) return;
  constexpr u64 INF = ~0ull;
  for (int i = tid; i < QPB; i += 256) {
    const int u = qb * QPB + i;
    UndRec r; r.q = -1; r.na = 0; r.t0 = -1; r.dm = 0; r.x0 = 0; r.y0 = 0;
    if (u < nQ) r = A.und[u];
    sU[i] = r;
    sEv[i] = 0; sNless[i] = 0; sNbad[i] = 0; sCand[i] = INF;
  }
  if (tid == 0) { sNrec = 0; sNover = 0; }
  __syncthreads();
  const int ul0 = wave * (32 * QSETS);
  v4i bq[QSETS][4];
  ResolveEpi<QSETS, EVS> epi;
#pragma unroll
  for (int s = 0; s < QSETS; s++) {
    const int ul = ul0 + 32 * s + col;
    const int q = max(sU[ul].q, 0);
#pragma unroll
    for (int kb = 0; kb < 4; kb++) bq[s][kb] = load_q(A.d1, q, kb, hi);
    epi.m1[s] = NONE_KEY; epi.I1[s] = -1; epi.nev[s] = 0;
    epi.thr[s] = sU[ul].q < 0 ? (int)0x80000000 : (sU[ul].dm - sU[ul].na) << 8;   // d < Dmin  <=>  key < (Dmin - |a'|^2) << 8; nothing for a dead lane
  }
  const int TEp = A.geo->TEp, offT = g.offT, ntilesV = A.geo->ntilesV;
  epi.TEp = TEp; epi.hi = hi; epi.stream0 = (ul0 + col) * 2 + hi; epi.evt = sEvt;
  const int tBeg = sp * tilesPerSplit, tEnd = min(tBeg + tilesPerSplit, ntilesV);
  sweep_core<QSETS, 26, 4, 1>(A.tiles, A.hrow, TEp, offT, tBeg, tEnd, bq, sm, epi);
  int (&m1)[QSETS] = epi.m1, (&I1)[QSETS] = epi.I1, (&nev)[QSETS] = epi.nev;
  __syncthreads();                       // the staging buffers are free from here on: they hold the work list
  // ---- 2. the logged groups: counts per query, a flat work list, streams that ran out of slots
  int *sList = reinterpret_cast<int *>(&sm[0][0]);     // (query in block) << 23 | half << 22 | tile
#pragma unroll
  for (int s = 0; s < QSETS; s++) {
    const int ul = ul0 + 32 * s + col;
    if (nev[s]) atomicAdd(&sEv[ul], nev[s]);
    if (m1[s] < NONE_KEY && sU[ul].q >= 0) {
      const int slot = I1[s];
      const int t = A.perm[phys_tile(slot >> 5, TEp, offT) * 32 + (slot & 31)];
      atomicMin(&sCand[ul], key64((m1[s] >> 8) + sU[ul].na, t));
    }
  }
  __syncthreads();
#pragma unroll
  for (int s = 0; s < QSETS; s++) {
    const int ul = ul0 + 32 * s + col;
    if (nev[s] == 0 || sEv[ul] >= A.nn) continue;      // nn or more groups in this split alone: the walk gives up
    if (nev[s] > EVS) { sOver[atomicAdd(&sNover, 1)] = (unsigned short)(ul * 2 + hi); continue; }
    const int base = atomicAdd(&sNrec, nev[s]);
    if (base + nev[s] > LCAP) {      // the list is full: the stream is rescanned like one that ran out of slots; its reserved entries are void
      for (int e = base; e < LCAP; e++) sList[e] = -1;
      sOver[atomicAdd(&sNover, 1)] = (unsigned short)(ul * 2 + hi);
      continue;
    }
    for (int e = 0; e < nev[s]; e++) sList[base + e] = (ul << 23) | (hi << 22) | sEvt[ul * 2 + hi][e];
  }
  __syncthreads();
  {
    const int l = lane & 15, qw = tid >> 4;            // 16 quarter waves
    // a visit = the exact distances of one group (compute), then its totals into the query's cells (commit); two visits are
    // computed before either is committed, so that their loads overlap
    struct Vis { int nl, nb, md, mt; };
    auto compute = [&](int ul, int vtile, int h, bool active) {
      const UndRec r = sU[active ? ul : 0];
      int vslot, pslot, t;
      const int d = group_dist16(A.d1 + (size_t)max(r.q, 0) * 128, r.na, A.tiles, A.norm2, A.perm, phys_tile(active ? vtile : 0, TEp, offT),
                                 vtile, h, l, &vslot, &pslot, &t);
      int nl = 0, nb = 0, cd = BIG, ct = BIG;
      if (active && d != BIG && t != r.t0) {
        if (d < r.dm) {
          nl = 1;
          // geometric consistency with NN0 (distanceSq, matching.cpp:174-179), f64
          const double2 xy = A.pos2p[pslot];
          const double dx = r.x0 - xy.x, dy = r.y0 - xy.y;
          if (dx * dx + dy * dy > A.contrDistSq) nb = 1;
        } else { cd = d; ct = t; }
      }
      // totals of the group over its 16 lanes
      nl += __builtin_amdgcn_update_dpp(0, nl, 0x128, 0xf, 0xf, false); nb += __builtin_amdgcn_update_dpp(0, nb, 0x128, 0xf, 0xf, false);
      nl += __builtin_amdgcn_update_dpp(0, nl, 0x124, 0xf, 0xf, false); nb += __builtin_amdgcn_update_dpp(0, nb, 0x124, 0xf, 0xf, false);
      nl += __builtin_amdgcn_update_dpp(0, nl, 0x122, 0xf, 0xf, false); nb += __builtin_amdgcn_update_dpp(0, nb, 0x122, 0xf, 0xf, false);
      nl += __builtin_amdgcn_update_dpp(0, nl, 0x121, 0xf, 0xf, false); nb += __builtin_amdgcn_update_dpp(0, nb, 0x121, 0xf, 0xf, false);
      Vis v; v.nl = nl; v.nb = nb;
      v.md = rowmin16(cd); v.mt = rowmin16(cd == v.md ? ct : BIG);
      return v;
    };
    auto commit = [&](const Vis &v, int ul, bool active) {
      if (active && l == 0) {
        if (v.nl) atomicAdd(&sNless[ul], v.nl);
        if (v.nb) atomicAdd(&sNbad[ul], v.nb);
        if (v.md != BIG) atomicMin(&sCand[ul], key64(v.md, v.mt));
      }
    };
    const int nrec = min(sNrec, LCAP);
    for (int b = 0; b < nrec; b += 32) {               // workgroup-uniform trip count
      const int e0 = b + qw, e1 = b + 16 + qw;
      int r0 = e0 < nrec ? sList[e0] : -1, r1 = e1 < nrec ? sList[e1] : -1;
      const bool on0 = r0 != -1, on1 = r1 != -1;
      if (!on0) r0 = 0;
      if (!on1) r1 = 0;
      const Vis v0 = compute((unsigned)r0 >> 23, r0 & 0x3fffff, (r0 >> 22) & 1, on0);
      const Vis v1 = compute((unsigned)r1 >> 23, r1 & 0x3fffff, (r1 >> 22) & 1, on1);
      commit(v0, (unsigned)r0 >> 23, on0);
      commit(v1, (unsigned)r1 >> 23, on1);
    }
    const int nover = sNover;
    for (int o = 0; o < nover; o++) {                  // a stream with more than EVS groups: every group of its tiles
      const int st = sOver[o];
      for (int tile = tBeg; tile < tEnd; tile += 16) commit(compute(st >> 1, tile + qw, st & 1, tile + qw < tEnd), st >> 1, tile + qw < tEnd);
    }
  }
  __syncthreads();
  // ---- 3. this split's share of the rows
  for (int i = tid; i < QPB; i += 256) {
    const int q = sU[i].q;
    if (q < 0) continue;
    MatchRow *row = A.rows + q;
    const int nl = sEv[i] >= A.nn ? A.nn : sNless[i];
    if (nl) atomicAdd(&row->nless, nl);
    if (sNbad[i]) atomicAdd(&row->nbad, sNbad[i]);
    const u64 c = sCand[i];                            // distance (an integer below 2^24) -> the bits of the float the row holds
    if (c != INF) atomicMin(reinterpret_cast<u64 *>(&row->tj), ((u64)(unsigned)__float_as_int((float)(int)(c >> 32)) << 32) | (unsigned)c);
  }
}

// ---------------- the "all points" mode (ratio >= 1, matching.cpp:397-428) for the queries k_match_decide could not finish -----------
// Every query gives a record there, closed by its first contradictive neighbour or by neighbour nn - 1: a query whose first
// neighbours all sit at NN0's place needs its exact sorted list down to rank nn - 1, which no reduction of the sweeps holds.
// A niche mode ("for example, for calculating PDF"), so exactness comes before speed: a workgroup takes an undecided query,
// writes the exact distance of every slot to its scratch row (a quarter wave per group of 16 rows) and then extracts the
// neighbours one by one, each the smallest (distance, slot) above the last, until the walk's rule closes the record.
struct PdfArgs {
  const uint8_t *d1;
  const int *norm2, *perm;
  const unsigned char *tiles;
  const TileGeo *geo;
  MatchGeom g;
  const double2 *pos2p;
  double contrDistSq;
  int nn;
  const UndRec *und;
  const int *nUndecided;
  MatchRow *rows;
  int *scratch;           // PDF_NW rows of 2 * offT * 32 distances
};
constexpr int PDF_NW = 64;
__device__ __forceinline__ void pdf_body(const PdfArgs &A) {
  __shared__ u64 sRed[4];
  __shared__ int sStop;
  const MatchGeom g = A.g;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l = tid & 15, qw = tid >> 4;
  const int nQ = *A.nUndecided;
  const int TEp = A.geo->TEp, ntilesV = A.geo->ntilesV;
  int *dist = A.scratch + (size_t)blockIdx.x * (size_t)g.ntilesUB * 32;       // by VIRTUAL slot
  constexpr u64 INF = ~0ull;
  for (int u = blockIdx.x; u < nQ; u += PDF_NW) {
    const UndRec r = A.und[u];
    __syncthreads();                       // the scratch row of the previous query is no longer read
    for (int grp = qw; grp < 2 * ntilesV; grp += 16) {
      int vslot, pslot, t;
      const int d = group_dist16(A.d1 + (size_t)r.q * 128, r.na, A.tiles, A.norm2, A.perm, phys_tile(grp >> 1, TEp, g.offT), grp >> 1, grp & 1,
                                 l, &vslot, &pslot, &t);
      dist[vslot] = d;                     // BIG for a padding row
    }
    __threadfence_block();
    __syncthreads();
    // the walk: neighbour j = the smallest (distance, slot) above neighbour j - 1
    u64 last = 0;
    bool first = true;
    int d0 = 0, dj = 0, tj = -1, dd1 = 0, t1 = -1, t0 = -1, j = 0, closed = 0;
    double x0 = 0, y0 = 0;
    const int nslots = ntilesV * 32;
    for (j = 0; j < A.nn && !closed; j++) {
      u64 best = INF;
      for (int s = tid; s < nslots; s += 256) {
        const int d = dist[s];
        if (d == BIG) continue;
        const u64 k = key64(d, s);
        if ((first || k > last) && k < best) best = k;
      }
#pragma unroll
      for (int m = 32; m >= 1; m >>= 1) best = min(best, shfl_xor64(best, m));
      if (lane == 0) sRed[wave] = best;
      __syncthreads();
      best = min(min(sRed[0], sRed[1]), min(sRed[2], sRed[3]));
      __syncthreads();
      if (best == INF) { closed = 2; break; }            // fewer than nn trains: the reference would read past its list
      last = best; first = false;
      const int cd = (int)(best >> 32), cs = (int)(unsigned)best;
      const int pslot = phys_tile(cs >> 5, TEp, g.offT) * 32 + (cs & 31);
      const int ct = A.perm[pslot];
      const double2 xy = A.pos2p[pslot];
      if (j == 0) { d0 = cd; t0 = ct; x0 = xy.x; y0 = xy.y; }
      else {
        if (j == 1) { dd1 = cd; t1 = ct; }
        const double dx = x0 - xy.x, dy = y0 - xy.y;
        if (j == A.nn - 1 || dx * dx + dy * dy > A.contrDistSq) { closed = 1; dj = cd; tj = ct; }
      }
    }
    if (tid == 0) {
      MatchRow o;
      o.t0 = t0; o.t1 = t1; o.nless = 0; o.nbad = 0; o.tj = closed == 1 ? tj : -1;
      o.dj = closed == 1 ? (float)dj : 0.f; o.d0 = (float)d0; o.d1 = (float)dd1;
      A.rows[r.q] = o;
    }
    (void)sStop;
  }
}

// ---- workspace layout: ONE description used by the size query and by the launcher -------------------------------------------
struct MatchLayout {
  int S, tilesPerSplit, ntilesUB, offT;
  size_t norm1, norm2, hrow, perm, pos2p, tiles, status, geo, partial, und, pdf, counter, bytes;
};
static MatchLayout match_layout(int n1, int n2, int qs, bool fat) {
  MatchLayout L;
  const int QPB = s1_qpb(qs, fat);
  const int nQB = (n1 + QPB - 1) / QPB;
  const int ntiles = ntiles_ub(n2);
  // one round of workgroups (one per CU); a second, partly filled round costs as much as the first.  Many query blocks
  // (N > 196 k) simply take several rounds.
#ifdef MATCH_TRACE
  static const int nwEnv = getenv("MODSX_MATCH_NW") ? atoi(getenv("MODSX_MATCH_NW")) : 0;   // workgroups per round, to trace 1 / 2 / 3 per CU
  int S = (nwEnv > 0 ? nwEnv : s1_round(qs, fat)) / nQB;
#else
  int S = s1_round(qs, fat) / nQB;
#endif
  if (S > ntiles / MINT) S = ntiles / MINT;
  if (S < 1) S = 1;
  int tps = (ntiles + S - 1) / S;
  tps = (tps + TPS - 1) & ~(TPS - 1);
  S = (ntiles + tps - 1) / tps;
  if (S < 1) S = 1;
  L.S = S; L.tilesPerSplit = tps; L.ntilesUB = ntiles; L.offT = region_tiles(n2);
  const size_t slots = (size_t)2 * L.offT * 32;       // two class regions
  size_t w = 0;
  auto take = [&](size_t bytes) { const size_t o = w; w += (bytes + 255) & ~(size_t)255; return o; };
  L.norm1 = take((size_t)n1 * 4);
  L.norm2 = take(slots * 4);
  L.hrow = take(slots * 4);
  L.perm = take(slots * 4);
  L.pos2p = take(slots * 16);
  L.tiles = take(slots * 128);
  L.status = take((size_t)((n2 + PB - 1) / PB) * 8);
  L.geo = take(sizeof(TileGeo));
  L.partial = take((size_t)n1 * S * 2 * KTOP * 8);
  L.und = take((size_t)n1 * 32);
  L.pdf = take((size_t)PDF_NW * ntiles * 32 * 4);      // k_match_pdf's scratch rows (the ratio >= 1 mode)
  L.counter = take(64);
  L.bytes = w;
  return L;
}
// the larger of the two geometries: the caller sizes the workspace before the launcher picks one
size_t match_workspace_bytes(int n1, int n2) {     // whatever shape the launcher picks (a batch takes the shape of its first problem)
  size_t b = 0;
  for (int qs = 2; qs <= 4; qs += 2) for (int fat = 0; fat < 2; fat++) b = std::max(b, match_layout(n1, n2, qs, fat != 0).bytes);
  return b;
}

// ---- batched entry points: blockIdx.z selects one of up to MATCH_MAXB independent problems (the pairs of a launch set).
struct MatchProblem {
  const uint8_t *d1, *d2;
  const double *pos2;
  int *norm1, *norm2, *hrow, *perm, *counter, *pdf;
  double2 *pos2p;
  u64 *status;
  TileGeo *geo;
  unsigned char *tiles;
  int2 *partial;
  UndRec *und;
  MatchRow *rows;
  MatchGeom g;
};
struct MatchBatch { MatchProblem p[MATCH_MAXB]; unsigned epoch; };

__global__ __launch_bounds__(256) void k_match_pack(MatchBatch b) {
  const MatchProblem &P = b.p[blockIdx.z];
  if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) *P.counter = 0;   // the undecided count (decide adds to it): no fill launch of its own
  PackArgs A;
  A.d1 = P.d1; A.d2 = P.d2; A.pos2 = P.pos2; A.n1 = P.g.n1; A.n2 = P.g.n2; A.offT = P.g.offT; A.epoch = b.epoch;
  A.norm1 = P.norm1; A.norm2 = P.norm2; A.hrow = P.hrow; A.perm = P.perm; A.pos2p = P.pos2p; A.tiles = P.tiles;
  A.status = P.status; A.geo = P.geo;
  pack_body(A);
}
#ifdef MATCH_TRACE
// debugging aid (tools/trace_match.py): when and where every workgroup of the last k_match_sweep1 launch ran
__device__ unsigned long long g_mtrace[16384][4];
#endif
template <int QS, bool FAT>
__global__ __launch_bounds__(64 * s1_waves(QS, FAT), FAT ? 1 : sweep_wps(QS)) void k_match_sweep1(MatchBatch b) {
  const MatchProblem &P = b.p[blockIdx.z];
  if ((int)blockIdx.x * s1_qpb(QS, FAT) >= P.g.n1 || (int)blockIdx.y >= P.g.S) return;
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
  A.d1 = P.d1; A.norm1 = P.norm1; A.tiles = P.tiles; A.hrow = P.hrow; A.geo = P.geo; A.g = P.g; A.partial = P.partial;
  sweep_body<QS, FAT>(A);
#ifdef MATCH_TRACE
  __syncthreads();
  if (threadIdx.x == 0 && wg < 16384) { g_mtrace[wg][1] = wall_clock64(); g_mtrace[wg][3] = __builtin_readcyclecounter() - g_mtrace[wg][3]; }
#endif
}
__global__ __launch_bounds__(16 * DECIDE_Q) void k_match_decide(MatchBatch b, double sqminratio, double contrDistSq, int nn) {
  const MatchProblem &P = b.p[blockIdx.z];
  if ((int)blockIdx.x * DECIDE_Q >= P.g.n1) return;
  DecideArgs A;
  A.d1 = P.d1; A.norm1 = P.norm1; A.norm2 = P.norm2; A.perm = P.perm; A.tiles = P.tiles; A.geo = P.geo; A.partial = P.partial; A.g = P.g;
  A.pos2p = P.pos2p; A.sqminratio = sqminratio; A.contrDistSq = contrDistSq; A.nn = nn; A.rows = P.rows;
  A.nUndecided = P.counter; A.und = P.und;
  decide_body(A);
}
template <int QS>
__global__ __launch_bounds__(256) void k_match_resolve(MatchBatch b, double contrDistSq, int nn) {
  const MatchProblem &P = b.p[blockIdx.z];
  ResolveArgs A;
  A.d1 = P.d1; A.norm2 = P.norm2; A.perm = P.perm; A.hrow = P.hrow; A.tiles = P.tiles; A.geo = P.geo; A.g = P.g;
  A.pos2p = P.pos2p; A.contrDistSq = contrDistSq; A.nn = nn; A.und = P.und; A.nUndecided = P.counter; A.rows = P.rows;
  resolve_body<QS>(A);
}
__global__ __launch_bounds__(256) void k_match_pdf(MatchBatch b, double contrDistSq, int nn) {
  const MatchProblem &P = b.p[blockIdx.z];
  PdfArgs A;
  A.d1 = P.d1; A.norm2 = P.norm2; A.perm = P.perm; A.tiles = P.tiles; A.geo = P.geo; A.g = P.g; A.pos2p = P.pos2p;
  A.contrDistSq = contrDistSq; A.nn = nn; A.und = P.und; A.nUndecided = P.counter; A.rows = P.rows; A.scratch = P.pdf;
  pdf_body(A);
}
// Problems with n1 == 0 or n2 == 0 must be left out by the caller.  workspace[i] holds match_workspace_bytes(n1[i], n2[i]).
static void launch_match_batch_once(hipStream_t s, int nb, const uint8_t *const *d1, const int *n1, const uint8_t *const *d2, const int *n2,
                                    const double *const *pos2, double sqminratio, double contrDistSq, int nn, MatchRow *const *rows,
                                    void *const *workspace, hipEvent_t *evSweep1);
void launch_match_batch(hipStream_t s, int nb, const uint8_t *const *d1, const int *n1, const uint8_t *const *d2, const int *n2,
                        const double *const *pos2, double sqminratio, double contrDistSq, int nn, MatchRow *const *rows,
                        void *const *workspace, hipEvent_t *evSweep1) {
  // (the four launches of a batch start from the inputs every time: issuing them twice -- MX_DUP, an experiment aid -- changes nothing)
  MX_DUP(K_MATCH) launch_match_batch_once(s, nb, d1, n1, d2, n2, pos2, sqminratio, contrDistSq, nn, rows, workspace, evSweep1);
}
static void launch_match_batch_once(hipStream_t s, int nb, const uint8_t *const *d1, const int *n1, const uint8_t *const *d2, const int *n2,
                                    const double *const *pos2, double sqminratio, double contrDistSq, int nn, MatchRow *const *rows,
                                    void *const *workspace, hipEvent_t *evSweep1) {
  if (nb <= 0) return;
  MatchBatch b;
  memset(&b, 0, sizeof b);
  // the epoch stamps the status words of k_match_pack's scan (nothing has to be cleared between launches)
  static std::atomic<unsigned> epochCtr{0};
  unsigned ep = ++epochCtr;
  if (ep == 0) ep = ++epochCtr;
  b.epoch = ep;
  int maxN1 = 0, maxS = 0, maxWg = 0;
  const int qs = match_qsets(nb, n1[0], n2[0]);
  const bool fat = match_fat(n1[0]) && !gpu_shared();     // whole-CU workgroups only when no other context's launches want the CUs
  for (int i = 0; i < nb; i++) {
    MatchProblem &P = b.p[i];
    const MatchLayout L = match_layout(n1[i], n2[i], qs, fat);
    P.g.n1 = n1[i]; P.g.n2 = n2[i]; P.g.S = L.S; P.g.tilesPerSplit = L.tilesPerSplit; P.g.qs = qs; P.g.ntilesUB = L.ntilesUB;
    P.g.offT = L.offT;
    char *w = (char *)workspace[i];
    P.norm1 = (int *)(w + L.norm1); P.norm2 = (int *)(w + L.norm2); P.hrow = (int *)(w + L.hrow); P.perm = (int *)(w + L.perm);
    P.pos2p = (double2 *)(w + L.pos2p); P.status = (u64 *)(w + L.status); P.geo = (TileGeo *)(w + L.geo);
    P.tiles = (unsigned char *)(w + L.tiles);
    P.partial = (int2 *)(w + L.partial);
    P.counter = (int *)(w + L.counter); P.und = (UndRec *)(w + L.und); P.pdf = (int *)(w + L.pdf);
    P.d1 = d1[i]; P.d2 = d2[i]; P.pos2 = pos2[i]; P.rows = rows[i];
    maxN1 = std::max(maxN1, n1[i]); maxS = std::max(maxS, L.S); maxWg = std::max(maxWg, (n2[i] + PB - 1) / PB);
  }
  hipLaunchKernelGGL(k_match_pack, dim3(std::max((maxN1 + 31) / 32, maxWg), 2, nb), dim3(256), 0, s, b);
  const int QPB = qpb_of(qs), NW2 = 256 * sweep_wps(qs), QPB1 = s1_qpb(qs, fat);
  const dim3 grid((maxN1 + QPB1 - 1) / QPB1, maxS, nb), block1(64 * s1_waves(qs, fat));
  if (evSweep1) hipEventRecord(evSweep1[0], s);
  if (qs == 4 && fat) hipLaunchKernelGGL((k_match_sweep1<4, true>), grid, block1, 0, s, b);
  else if (qs == 4) hipLaunchKernelGGL((k_match_sweep1<4, false>), grid, block1, 0, s, b);
  else if (fat) hipLaunchKernelGGL((k_match_sweep1<2, true>), grid, block1, 0, s, b);
  else hipLaunchKernelGGL((k_match_sweep1<2, false>), grid, block1, 0, s, b);
  if (evSweep1) hipEventRecord(evSweep1[1], s);
  hipLaunchKernelGGL(k_match_decide, dim3((maxN1 + DECIDE_Q - 1) / DECIDE_Q, 1, nb), dim3(16 * DECIDE_Q), 0, s, b, sqminratio, contrDistSq, nn);
  // resolve: one round of workgroups dealt out on the device as (undecided block, split); more only if there could be more
  // undecided query blocks than that
  const dim3 grid2(std::max(NW2, (maxN1 + QPB - 1) / QPB), 1, nb);
  if (sqminratio >= 1.0) hipLaunchKernelGGL(k_match_pdf, dim3(PDF_NW, 1, nb), dim3(256), 0, s, b, contrDistSq, nn);
  else if (qs == 4) hipLaunchKernelGGL(k_match_resolve<4>, grid2, dim3(256), 0, s, b, contrDistSq, nn);
  else hipLaunchKernelGGL(k_match_resolve<2>, grid2, dim3(256), 0, s, b, contrDistSq, nn);
}

void launch_match(hipStream_t s, const uint8_t *d1, int n1, const uint8_t *d2, int n2, const double *pos2,
                  double sqminratio, double contrDistSq, int nn, MatchRow *rows, void *workspace) {
  if (n1 <= 0 || n2 <= 0) return;
  launch_match_batch(s, 1, &d1, &n1, &d2, &n2, &pos2, sqminratio, contrDistSq, nn, &rows, &workspace, nullptr);
}

}  // namespace mx

#ifdef SWEEP_PHASE_TRACE
extern "C" __attribute__((visibility("default"))) int modsx_debug_phase_trace(unsigned long long *out, int n) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(mx::g_ptrace), (size_t)n * 64, 0, hipMemcpyDeviceToHost);
}
#endif
#ifdef MATCH_TRACE
extern "C" __attribute__((visibility("default"))) int modsx_debug_match_trace(unsigned long long *out, int n) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(mx::g_mtrace), (size_t)n * 32, 0, hipMemcpyDeviceToHost);
}
#endif
