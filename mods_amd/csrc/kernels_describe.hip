// kernels_describe.hip -- affine-normalised patch extraction and SIFT / RootSIFT description.
//
// Reference: DescribeRegions<>, synth-detection.hpp:169-255 (slow/accurate and fast branches)
//   interpolate                  detectors/helpers.cpp:551-626
//   gaussianBlurInplace          detectors/helpers.cpp:726-731 (cv::GaussianBlur, BORDER_REPLICATE)
//   photometricallyNormalize     detectors/helpers.cpp:666-715
//   SIFTDescriptor               matching/siftdesc.cpp:22-131 (bins, samplePatch), 136-278 (norms),
//                                290-379 (gradients + atan2LUTff)
// Data layout: a region's P x P f32 window (P = 2*ceil(s*mrSize)+3) is sampled one row tile at a time into LDS and
// row-filtered there at the columns the resampling needs (k_sample_rows_lds -> arena B, P x NC), column-filtered at the
// needed rows (k_blur_cols_lds -> arena C, NC x NC; small windows: in the sampling kernel itself) and read once by
// k_describe, which resamples it to 41x41 in LDS.  Windows whose row tile does not fit LDS take k_patch_sample (arena A)
// and the global-memory filter k_patch_blur.
#include "engine.hpp"

namespace mx {

// Workgroups are dispatched round-robin over the 8 XCDs (block b -> XCD b % 8), each with its own L2.  Neighbouring
// tiles of one window share halo rows/columns, so runs of G consecutive logical tiles are mapped to the same XCD;
// the runs themselves stay interleaved over the XCDs so that windows of very different cost (ksize grows with the
// window) remain balanced.  Bijective on [0, n) for the full super-groups; the ragged tail is left as is.
MX_D int xcd_swizzle(int b, int n) {
  const int G = 8, SG = 8 * G;
  const int full = (n / SG) * SG;
  if (b >= full) return b;
  const int sg = b / SG, r = b - sg * SG;
  const int xcd = r & 7, k = r >> 3;      // this block runs on XCD `xcd`; it is that XCD's k-th block of the super-group
  return sg * SG + xcd * G + k;
}

// The sampling kernel is the one stage whose inputs (the images) are shared between regions.  Its tile list is sorted by
// (image, 64-px row band, x) on the host and XCD x takes the x-th CONTIGUOUS eighth of it (block b runs on XCD b % 8 and is
// that XCD's (b / 8)-th block), so an XCD's L2 holds only its part of the images instead of all of them: without this every
// L2 pulled its own copy of every image (FETCH_SIZE ~10x the image bytes).  The grid is 8 * ceil(n / 8) blocks.
// (xcd_chunk lives in kmath.hpp: the Baumberg and orientation kernels use it for their image-major job lists too)

// tile -> job table: a per-workgroup binary search over the prefix array costs ~12 dependent L2 round trips,
// longer than the useful work of a 256-element tile, so the prefix array is expanded once per launch set
__global__ __launch_bounds__(64) void k_expand_tiles(const int *prefix, int nJobs, int *tileJob) {
  const int j = blockIdx.x;
  if (j >= nJobs) return;
  const int b = prefix[j], e = prefix[j + 1];
  for (int t = b + threadIdx.x; t < e; t += 64) tileJob[t] = j;
}

// tile descriptors of the LDS blur kernels: everything a workgroup needs in one 48-byte scalar load, so that its first
// vector loads (the inputs it parks in LDS) are two dependent round trips from the launch instead of three
// (blockIdx.y = 0: the row tiles of the fused sampling kernel, 1: the tiles of the column filter, 2: the windows' row starts)
// One THREAD per job: a workgroup per job (round 5) was 2 x 12 k one-wave workgroups of a few stores each per chunk, 27 us of
// dispatch for 2 us of work.
__global__ __launch_bounds__(256) void k_expand_blur_tiles(const DescJob *jobs, const int *prefix0, const int *prefix1, int nJobs,
                                                           const int *needTab, BlurTile *tiles0, BlurTile *tiles1, float2 *rowStart) {
  const int j = blockIdx.x * 256 + threadIdx.x, pass = blockIdx.y;
  if (j >= nJobs) return;
  const int *prefix = pass == 1 ? prefix1 : prefix0;
  const int b = prefix[j], e = prefix[j + 1];
  if (pass == 2) {
    // row starts of interpolate() (rx += a12, ry += a22 per row, helpers.cpp:563-566): one serial chain per window, run
    // here once instead of by every row tile of the fused sampling kernel (a tile of a large window is a few rows only)
    if (b == e) return;
    const DescJob jb = jobs[j];
    float2 *rs = rowStart + jb.scratchOfs;
    const int half = jb.P >> 1;
    float rx = jb.x - (float)half * jb.a12, ry = jb.y - (float)half * jb.a22;
    for (int r = 0; r < jb.P; r++) { rs[r] = make_float2(rx, ry); rx += jb.a12; ry += jb.a22; }
    return;
  }
  if (b == e) return;
  BlurTile *tiles = pass ? tiles1 : tiles0;
  const DescJob jb = jobs[j];
  const int R = jb.ksize >> 1;
  BlurTile bt;
  bt.P = jb.P; bt.NC = jb.NC; bt.n = jb.ksize; bt.tapOfs = jb.tapOfs; bt.needOfs = jb.needOfs;
  bt.job = j; bt.pad = 0;
  { const int NP = (jb.NC + 1) >> 1; bt.magic = ((1 << 20) + NP - 1) / NP; }
  for (int t = b; t < e; t++) {
    if (pass == 0) {
      const int r0 = (t - b) * jb.rows0;
      bt.count = jb.P - r0 < jb.rows0 ? jb.P - r0 : jb.rows0;
      bt.first = r0; bt.lo = 0; bt.span = bt.count;
      bt.srcOfs = jb.scratchOfs + (size_t)r0 * jb.P;
      bt.dstOfs = jb.rowOfs + (size_t)r0 * jb.NC;
    } else {
      const int ro0 = (t - b) * jb.ro1;
      bt.count = jb.NC - ro0 < jb.ro1 ? jb.NC - ro0 : jb.ro1;
      bt.first = ro0;
      bt.lo = needTab[jb.needOfs + ro0] - R;
      bt.span = needTab[jb.needOfs + ro0 + bt.count - 1] + R - bt.lo + 1;
      bt.srcOfs = jb.rowOfs;
      bt.dstOfs = jb.gridOfs + (size_t)ro0 * jb.NC;
    }
    tiles[t] = bt;
  }
}

// --- stage 1: interpolate(img, x, y, A, smoothed(P x P)) ---------------------------------------
// One wavefront per tile of 64 rows x SAMPLE_COLS columns of one window.  The coordinates of a row are f32 running sums
// (rx += a12 per row, WX += a11 per column) -- cheap, but serial along the row -- so lane j walks row j and parks the
// (WX, WY) of SAMPLE_C columns in LDS; the bilinear taps are then taken with the lanes running ALONG the rows (16
// neighbouring samples of 4 rows per instruction), so that a gather touches a handful of cache lines instead of 64 and
// every lane works even when the tile has few rows; the stores are row-contiguous.  Splitting long rows into column
// tiles keeps the longest serial walk at SAMPLE_COLS steps instead of P (up to ~2000).

constexpr int SAMPLE_COLS = 128, SAMPLE_C = 16, SAMPLE_CP = SAMPLE_C + 1, SAMPLE_B = 8;

// the taps of one parked chunk: SAMPLE_B samples (4 gathers each) are in flight per lane; the border branch of
// interpolate() is hoisted out of the loop so that the loads of a batch can be issued together
template <bool TOUCH>
__device__ __forceinline__ void sample_chunk(const ImgRef &im, const float *cx, const float *cy, float *dst, int P, int tot, int nc,
                                             int lane) {
  for (int e0 = lane; e0 < tot; e0 += 64 * SAMPLE_B) {
    float v[SAMPLE_B];
#pragma unroll
    for (int u = 0; u < SAMPLE_B; u++) {
      const int e = e0 + 64 * u, r = e / SAMPLE_C, c = e - r * SAMPLE_C;
      v[u] = (e < tot && c < nc) ? bilinear_tap(as_global(im.d), im.rows, im.cols, cx[r * SAMPLE_CP + c], cy[r * SAMPLE_CP + c], TOUCH) : 0.f;
    }
#pragma unroll
    for (int u = 0; u < SAMPLE_B; u++) {
      const int e = e0 + 64 * u, r = e / SAMPLE_C, c = e - r * SAMPLE_C;
      if (e < tot && c < nc) dst[(size_t)r * P + c] = v[u];
    }
  }
}

__global__ __launch_bounds__(64) void k_patch_sample(const DescJob *jobs, const int *tilePrefix, const int *tileJob,
                                                     const ImgRef *imgs, float *scratch, int nTiles) {
  const int tile = xcd_chunk(blockIdx.x, nTiles);
  if (tile >= nTiles) return;
  const int jid = tileJob[tile];
  const DescJob jb = jobs[jid];
  const int P = jb.P;
  if (P <= 0) return;
  const int lane = threadIdx.x;
  const int local = tile - tilePrefix[jid];
  const int ncolTiles = (P + SAMPLE_COLS - 1) / SAMPLE_COLS;
  const int rowTile = local / ncolTiles, colTile = local - rowTile * ncolTiles;
  const int row0 = rowTile * 64, col0 = colTile * SAMPLE_COLS;
  const int colEnd = (col0 + SAMPLE_COLS) < P ? (col0 + SAMPLE_COLS) : P;
  const int row = row0 + lane;
  const ImgRef im = imgs[jb.img];
  __shared__ float cx[64 * SAMPLE_CP], cy[64 * SAMPLE_CP];
  const int half = P >> 1;
  const bool touch = check_borders(im.cols, im.rows, jb.x, jb.y, jb.a11, jb.a12, jb.a21, jb.a22, P, P);
  float rx = jb.x - (float)half * jb.a12;
  float ry = jb.y - (float)half * jb.a22;
  const int nsteps = row < P ? row : 0;
  for (int j = 0; j < nsteps; j++) { rx += jb.a12; ry += jb.a22; }
  float WX = rx - (float)half * jb.a11;
  float WY = ry - (float)half * jb.a21;
  for (int i = 0; i < col0; i++) { WX += jb.a11; WY += jb.a21; }
  float *dst = scratch + jb.scratchOfs;
  const int rowsHere = (P - row0) < 64 ? (P - row0) : 64;
  for (int c0 = col0; c0 < colEnd; c0 += SAMPLE_C) {
    const int nc = (colEnd - c0) < SAMPLE_C ? (colEnd - c0) : SAMPLE_C;
    if (row < P) {
#pragma unroll 4
      for (int i = 0; i < nc; i++) {
        cx[lane * SAMPLE_CP + i] = WX;
        cy[lane * SAMPLE_CP + i] = WY;
        WX += jb.a11;
        WY += jb.a21;
      }
    }
    __syncthreads();
    const int tot = rowsHere * SAMPLE_C;
    if (!touch) sample_chunk<false>(im, cx, cy, dst + (size_t)row0 * P + c0, P, tot, nc, lane);
    else sample_chunk<true>(im, cx, cy, dst + (size_t)row0 * P + c0, P, tot, nc, lane);
    __syncthreads();
  }
}

// --- stage 2: separable Gaussian blur of each window, replicate border -------------------------
// The 41x41 resampling of stage 3 reads the blurred window only at NC <= 82 columns and the same NC rows
// (x0_i, x0_i + 1 for the 41 sample coordinates), so
//   pass 0 filters the rows only at those columns:      P x NC outputs   (cv RowFilter order: taps left->right;
//                                                                         SymmRowSmallFilter when ksize <= 5)
//   pass 1 filters the columns only at those rows:      NC x NC outputs  (SymmColumnFilter: centre + (below+above)*k)
// Every output is the same sum of the same terms in the same order as in the full blur.
// k_patch_blur is the global-memory form of the two passes; since the LDS kernels below exist it only serves windows
// whose tiles do not fit LDS.  A workgroup forms BLUR_TILE outputs (4 per thread): the start of a workgroup is a chain
// of dependent loads (tile -> job -> needed column -> inputs) that lasts longer than the arithmetic of 256 outputs, so
// fewer, fatter workgroups run faster; the taps are parked in LDS so that a tap costs one vector load, not two.
constexpr int BLUR_TILE = 1024, BLUR_TAPS = 512;   // ksize <= 512 is enforced by the host (windows up to ~2300 px)

__global__ __launch_bounds__(256) void k_patch_blur(const DescJob *jobs, const int *tilePrefix, const int *tileJob,
                                                    const float *taps, const int *needTab, const float *src,
                                                    float *dst, int pass) {
  const int tile = xcd_swizzle(blockIdx.x, gridDim.x);
  const int jid = tileJob[tile];
  const DescJob jb = jobs[jid];
  const int P = jb.P, NC = jb.NC;
  if (P <= 0) return;
  const int n = jb.ksize, R = n >> 1;
  __shared__ float sk[BLUR_TAPS];
  __shared__ int sneed[96];
  const float *kg = taps + jb.tapOfs;
  for (int i = threadIdx.x; i < n; i += 256) sk[i] = kg[i];
  if (threadIdx.x < NC) sneed[threadIdx.x] = needTab[jb.needOfs + threadIdx.x];
  __syncthreads();
  const int e0 = (tile - tilePrefix[jid]) * BLUR_TILE + threadIdx.x;
  constexpr int NQ = BLUR_TILE / 256;   // outputs per thread, advanced in lock-step over the taps so that NQ (x unroll)
                                        // independent loads are in flight per thread: the kernel is bound by memory
                                        // latency (VALU 7 %, waitcnt 68 % of the wave cycles), not by arithmetic
  const float *kk = sk;
  if (pass == 0) {
    const float *A = src + jb.scratchOfs;
    const int total = P * NC;
    const float *row[NQ];
    int c[NQ];
    float v[NQ];
#pragma unroll
    for (int q = 0; q < NQ; q++) {
      int e = e0 + q * 256;
      e = e < total ? e : total - 1;          // out-of-range slots repeat the last output and are not stored
      const int r = e / NC, ci = e - r * NC;
      c[q] = sneed[ci];
      row[q] = A + (size_t)r * P;
    }
    if (n == 1) {
#pragma unroll
      for (int q = 0; q < NQ; q++) v[q] = row[q][c[q]];
    } else if (n <= 5) {
#pragma unroll
      for (int q = 0; q < NQ; q++) v[q] = row[q][c[q]] * kk[R];
      for (int j = 1; j <= R; j++) {
#pragma unroll
        for (int q = 0; q < NQ; q++) {
          const int cm = c[q] - j < 0 ? 0 : c[q] - j, cp = c[q] + j > P - 1 ? P - 1 : c[q] + j;
          v[q] = v[q] + (row[q][cm] + row[q][cp]) * kk[R + j];
        }
      }
    } else {
#pragma unroll
      for (int q = 0; q < NQ; q++) v[q] = 0.f;
#pragma unroll 4
      for (int j = 0; j < n; j++) {
        const float kj = kk[j];
#pragma unroll
        for (int q = 0; q < NQ; q++) {
          int cc = c[q] + j - R;
          cc = cc < 0 ? 0 : (cc > P - 1 ? P - 1 : cc);
          v[q] = v[q] + row[q][cc] * kj;
        }
      }
    }
#pragma unroll
    for (int q = 0; q < NQ; q++) {
      const int e = e0 + q * 256;
      if (e < total) dst[jb.rowOfs + e] = v[q];
    }
  } else {
    const float *S = src + jb.rowOfs;   // P x NC
    const int total = NC * NC;
    const float *col[NQ];
    int r[NQ];
    float v[NQ];
#pragma unroll
    for (int q = 0; q < NQ; q++) {
      int e = e0 + q * 256;
      e = e < total ? e : total - 1;
      const int ri = e / NC, ci = e - ri * NC;
      r[q] = sneed[ri];
      col[q] = S + ci;
    }
    if (n == 1) {
#pragma unroll
      for (int q = 0; q < NQ; q++) v[q] = col[q][(size_t)r[q] * NC];
    } else {
#pragma unroll
      for (int q = 0; q < NQ; q++) v[q] = kk[R] * col[q][(size_t)r[q] * NC] + 0.f;
#pragma unroll 2
      for (int j = 1; j <= R; j++) {
        const float kj = kk[R + j];
#pragma unroll
        for (int q = 0; q < NQ; q++) {
          const int rp = r[q] + j > P - 1 ? P - 1 : r[q] + j, rm = r[q] - j < 0 ? 0 : r[q] - j;
          v[q] = v[q] + kj * (col[q][(size_t)rp * NC] + col[q][(size_t)rm * NC]);
        }
      }
    }
#pragma unroll
    for (int q = 0; q < NQ; q++) {
      const int e = e0 + q * 256;
      if (e < total) dst[jb.gridOfs + e] = v[q];
    }
  }
}

// LDS variants of the two passes.  The global-memory kernel above spends most of its issue slots on addresses: per tap a
// clamp, a 64-bit address and the load for one multiply and one add, and the vector ALUs (not the memory system) are what
// saturates.  Here a workgroup first parks its inputs in LDS WITH the replicated border written out (the clamp is paid once
// per input, not once per tap), after which a tap is one ds_read at an immediate offset, one multiply and one add; the taps
// themselves are wave-uniform and come through the scalar cache.  Sums, terms and order are those of the kernel above.
//   rows:  a tile is jb.rows0 consecutive window rows, each stored as R + P + R floats
//   cols:  a tile is jb.ro1 consecutive needed rows; it parks the source rows need[first] - R .. need[last] + R
// Jobs whose tile does not fit BLUR_LDS floats have rows0 / ro1 = 0 and go through k_patch_blur.
constexpr int BLUR_T = 256, BLUR_W = BLUR_T / 64;   // threads / waves per workgroup of the LDS blur kernels
constexpr int BLUR_LDS = MODSX_SR_WIN;   // row tile of the fused sampling + row-filter kernel: 20 KB
constexpr int BLUR_LDS_C = MODSX_BLUR_LDS_C;                               // column filter: fatter tiles re-read fewer halo rows

// the row filter proper, on a tile parked in LDS as nr rows of R + P + R floats (replicated border written out)
// (dst = the tile's place in arena B, row stride NC; or, for the fully fused small windows, an LDS block of row stride ostride)
__device__ __forceinline__ void blur_rows_from_lds(const BlurTile &bt, const float *win, const int *sneed, const float *__restrict__ taps,
                                                   float *__restrict__ out, int ostride) {
  const int NC = bt.NC, n = bt.n, R = n >> 1, RW = bt.P + 2 * R, nr = bt.count;
  // A thread forms 4 PAIRS of horizontally adjacent outputs (needed columns 2m, 2m+1 -- the host checks that such pairs are
  // neighbours in the window, which the x0 / x0+1 construction gives): both members of a pair take tap j from adjacent LDS
  // words, so a tap of a pair is one 2-word read, one packed multiply and one packed add.
  const float *kg = taps + bt.tapOfs;
  const int NP = (NC + 1) >> 1, total = nr * NP;
  constexpr int NQ = 4;
  for (int base = threadIdx.x; base < total; base += BLUR_T * NQ) {
    int p[NQ];     // win[p[q] + j], win[p[q] + j + 1] = window columns need[2m] + j - R, + 1 of the pair's row
    float v0[NQ], v1[NQ];
#pragma unroll
    for (int q = 0; q < NQ; q++) {
      int e = base + q * BLUR_T;
      e = e < total ? e : total - 1;                                      // idle slots repeat the last pair, not stored
      // e / NP (exact for e < 4096, NP <= 96); every product here is 24 bit x 24 bit (e < 2^12, magic <= 2^20, rows and strides
      // < 2^12): v_mul_u32_u24 issues at the full rate, the 32-bit v_mul_lo_u32 the plain `*` compiles to at a quarter of it
      const int r = (int)(__umul24((unsigned)e, (unsigned)bt.magic) >> 20), m = e - (int)__umul24((unsigned)r, (unsigned)NP);
      p[q] = (int)__umul24((unsigned)r, (unsigned)RW) + sneed[2 * m];
    }
    if (n == 1) {
#pragma unroll
      for (int q = 0; q < NQ; q++) { v0[q] = win[p[q]]; v1[q] = win[p[q] + 1]; }
    } else if (n <= 5) {
#pragma unroll
      for (int q = 0; q < NQ; q++) { v0[q] = win[p[q] + R] * kg[R]; v1[q] = win[p[q] + R + 1] * kg[R]; }
      for (int j = 1; j <= R; j++) {
        const float kj = kg[R + j];
#pragma unroll
        for (int q = 0; q < NQ; q++) {
          v0[q] = v0[q] + (win[p[q] + R - j] + win[p[q] + R + j]) * kj;
          v1[q] = v1[q] + (win[p[q] + R - j + 1] + win[p[q] + R + j + 1]) * kj;
        }
      }
    } else {
#pragma unroll
      for (int q = 0; q < NQ; q++) v0[q] = v1[q] = 0.f;
#pragma unroll 8
      for (int j = 0; j < n; j++) {
        const float kj = kg[j];
#pragma unroll
        for (int q = 0; q < NQ; q++) {
          v0[q] = v0[q] + win[p[q] + j] * kj;
          v1[q] = v1[q] + win[p[q] + j + 1] * kj;
        }
      }
    }
#pragma unroll
    for (int q = 0; q < NQ; q++) {
      const int e = base + q * BLUR_T;
      if (e < total) {
        const int r = (int)(__umul24((unsigned)e, (unsigned)bt.magic) >> 20), m = e - (int)__umul24((unsigned)r, (unsigned)NP);
        float *o = out + (__umul24((unsigned)r, (unsigned)ostride) + 2u * (unsigned)m);
        o[0] = v0[q];
        if (2 * m + 1 < NC) o[1] = v1[q];
      }
    }
  }
}

// the column filter proper on source rows parked in LDS from row `lo` on (row stride LS, replicated border rows written
// out): needed rows ro0 .. ro0 + nro - 1, all NC columns; out = first output of the tile (row stride NC)
template <int LS>
__device__ __forceinline__ void blur_cols_from_lds(int NC, int n, int ro0, int nro, int lo, int magic, const float *win, const int *sneed,
                                                   const float *__restrict__ kg, float *__restrict__ out) {
  const int R = n >> 1;
  // pairs of horizontally adjacent outputs again: (ri, 2m) and (ri, 2m+1) read adjacent LDS words in every parked row
  const int NP = (NC + 1) >> 1, total = nro * NP;
  constexpr int NQ = 4;
  for (int base = threadIdx.x; base < total; base += BLUR_T * NQ) {
    int pc[NQ];
    float2 v[NQ];
#pragma unroll
    for (int q = 0; q < NQ; q++) {
      int e = base + q * BLUR_T;
      e = e < total ? e : total - 1;
      const int ri = (int)(__umul24((unsigned)e, (unsigned)magic) >> 20), m = e - (int)__umul24((unsigned)ri, (unsigned)NP);
      pc[q] = (int)__umul24((unsigned)(sneed[ro0 + ri] - lo), (unsigned)LS) + 2 * m;   // (parked row >= 0)
    }
    if (n == 1) {
#pragma unroll
      for (int q = 0; q < NQ; q++) v[q] = *(const float2 *)&win[pc[q]];
    } else {
      const float kc = kg[R];
#pragma unroll
      for (int q = 0; q < NQ; q++) {
        const float2 c = *(const float2 *)&win[pc[q]];
        v[q].x = kc * c.x + 0.f; v[q].y = kc * c.y + 0.f;
      }
#pragma unroll 4
      for (int j = 1; j <= R; j++) {
        const float kj = kg[R + j];
#pragma unroll
        for (int q = 0; q < NQ; q++) {
          const float2 a = *(const float2 *)&win[pc[q] + j * LS], b = *(const float2 *)&win[pc[q] - j * LS];
          v[q].x = v[q].x + kj * (a.x + b.x);
          v[q].y = v[q].y + kj * (a.y + b.y);
        }
      }
    }
#pragma unroll
    for (int q = 0; q < NQ; q++) {
      const int e = base + q * BLUR_T;
      if (e < total) {
        const int ri = (int)(__umul24((unsigned)e, (unsigned)magic) >> 20), m = e - (int)__umul24((unsigned)ri, (unsigned)NP);
        float *o = out + (__umul24((unsigned)ri, (unsigned)NC) + 2u * (unsigned)m);
        o[0] = v[q].x;
        if (2 * m + 1 < NC) o[1] = v[q].y;
      }
    }
  }
}

// Stage 1 + the row pass of stage 2 in one launch: a workgroup SAMPLES its tile of window rows (interpolate(), the f32
// running-sum coordinates of k_patch_sample) straight into the LDS tile of the row filter and filters it there, so the
// P x P window never exists in HBM (arena A was written once and read once per region: 2 x 4 P^2 bytes, the largest
// traffic item of the describe stage).  Lane j of a wave walks row j of the tile over the wave's QUARTER of the columns
// (its coordinates start with col0 dependent adds, as for the column tiles of k_patch_sample) and parks SR_C columns at a
// time; the taps are then taken with the lanes running along the rows.  Coordinates, taps, filter sums: term for term those
// of k_patch_sample + the row filter.
// A wave parks C columns of up to 64 rows at a time in its 64 x 9 words of coordinates: C = 8 for tiles of more than 32 rows,
// C = 16 for tiles of up to 32 rows (the host keeps row tiles out of the 33..48 range), so that a lane has 8 samples --
// 16 loads -- in flight either way.
constexpr int SR_CW = MODSX_SR_HALF ? 4 : 8;      // columns parked per pass for tiles of more than 32 rows (twice that up to 32 rows)
constexpr int SR_WORDS = 64 * (SR_CW + 1);
constexpr int FC_LS = 64, FC_ROWS = MODSX_FC_ROWS;   // fully fused small windows: NC <= FC_LS needed columns, P + 2 R <= FC_ROWS block rows
MX_D int wave_id() { return __builtin_amdgcn_readfirstlane(threadIdx.x >> 6); }

template <bool TOUCH, int C>
__device__ __forceinline__ void sample_chunk_lds(const ImgRef &im, const float *cx, const float *cy, float *dst, int RW, int tot, int nc,
                                                 int lane) {
  constexpr int CP = C + 1, PER = SR_CW;   // tot <= 64 * PER (64 rows x SR_CW columns, or 32 x 2 SR_CW)
  float v[PER];
#pragma unroll
  for (int u = 0; u < PER; u++) {
    const int e = lane + 64 * u, r = e / C, c = e - r * C;
    v[u] = (e < tot && c < nc) ? bilinear_tap(as_global(im.d), im.rows, im.cols, cx[r * CP + c], cy[r * CP + c], TOUCH) : 0.f;
  }
#pragma unroll
  for (int u = 0; u < PER; u++) {
    const int e = lane + 64 * u, r = e / C, c = e - r * C;
    if (e < tot && c < nc) dst[__umul24((unsigned)r, (unsigned)RW) + (unsigned)c] = v[u];
  }
}

template <int C>
__device__ __forceinline__ void sample_rows_tile(const BlurTile &bt, const DescJob &jb, const ImgRef &im, const float2 *rowStart,
                                                 float *win, float *cx, float *cy, int lane, int wave) {
  constexpr int CP = C + 1, RG = C == SR_CW ? 64 : 32;   // rows per pass of the wave
  const int P = bt.P, R = bt.n >> 1, RW = P + 2 * R, nr = bt.count, r0 = bt.first;
  const int half = P >> 1;
  const bool touch = check_borders(im.cols, im.rows, jb.x, jb.y, jb.a11, jb.a12, jb.a21, jb.a22, P, P);
  const int cper = (P + BLUR_W - 1) / BLUR_W, cb = wave * cper, ce = (cb + cper) < P ? (cb + cper) : P;
  for (int rb = 0; rb < nr; rb += RG) {
    const bool active = lane < RG && rb + lane < nr;
    const float2 rs = rowStart[active ? r0 + rb + lane : r0];   // the running sums of the window's rows (k_expand_blur_tiles)
    float WX = rs.x - (float)half * jb.a11;
    float WY = rs.y - (float)half * jb.a21;
    for (int i = 0; i < cb; i++) { WX += jb.a11; WY += jb.a21; }
    const int rowsHere = (nr - rb) < RG ? (nr - rb) : RG;
    for (int c0 = cb; c0 < ce; c0 += C) {
      const int nc = (ce - c0) < C ? (ce - c0) : C;
      if (active) {
#pragma unroll
        for (int i = 0; i < C; i++) {
          if (i < nc) {
            cx[lane * CP + i] = WX;
            cy[lane * CP + i] = WY;
            WX += jb.a11;
            WY += jb.a21;
          }
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      float *d = win + rb * RW + R + c0;
      if (!touch) sample_chunk_lds<false, C>(im, cx, cy, d, RW, rowsHere * C, nc, lane);
      else sample_chunk_lds<true, C>(im, cx, cy, d, RW, rowsHere * C, nc, lane);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
  }
}

__global__ __launch_bounds__(BLUR_T, MODSX_SR_WGS) void k_sample_rows_lds(const BlurTile *__restrict__ tiles, const DescJob *__restrict__ jobs,
                                                         const ImgRef *__restrict__ imgs, const float *__restrict__ taps,
                                                         const int *__restrict__ needTab, float *__restrict__ dst, int nTiles,
                                                         const float2 *__restrict__ rowStarts, float *__restrict__ dstGrid) {
  const int ti = xcd_chunk(blockIdx.x, nTiles);   // tiles follow the (image, row band, x) order of the jobs: one part of the images per XCD
  if (ti >= nTiles) return;
  const BlurTile bt = tiles[ti];
  const DescJob jb = jobs[bt.job];
  const int P = bt.P, NC = bt.NC;
  const int n = bt.n, R = n >> 1, RW = P + 2 * R;
#ifdef MODSX_SAMPLE_PAD
  __shared__ volatile char spad_[MODSX_SAMPLE_PAD];
  if (threadIdx.x == 0) spad_[MODSX_SAMPLE_PAD - 1] = 1;
#endif
  __shared__ float win[BLUR_LDS + 2];
  __shared__ int sneed[96];
  // coordinates of the sampling phase (2 x 4 waves x 64 x 9 words); afterwards, for a small window that is here as a whole
  // (DescJob::ro1 < 0), the row-filtered block with its replicated border rows, which the column filter then reads in place
  __shared__ __attribute__((aligned(16))) float aux[FC_ROWS * FC_LS];
  static_assert(FC_ROWS * FC_LS >= 2 * BLUR_W * SR_WORDS, "aux holds the sampling coordinates");
  float *const cxw = aux + wave_id() * SR_WORDS, *const cyw = aux + (BLUR_W + wave_id()) * SR_WORDS;
  const int nr = bt.count;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  for (int i = threadIdx.x; i < NC; i += BLUR_T) sneed[i] = needTab[bt.needOfs + i];
  const ImgRef im = imgs[jb.img];
  const float2 *rowStart = rowStarts + jb.scratchOfs;
  if (nr <= 32) sample_rows_tile<2 * SR_CW>(bt, jb, im, rowStart, win, cxw, cyw, lane, wave);
  else sample_rows_tile<SR_CW>(bt, jb, im, rowStart, win, cxw, cyw, lane, wave);
  __syncthreads();
  // replicated border: R copies of the first and of the last sample of every row
  // (2^s >= 2 R lanes per row, 64 >> s rows per wavefront and pass: no division by the run-time 2 R per element)
  if (R > 0) {
    const int twoR = 2 * R;
    int s = 32 - __clz(twoR - 1);
    s = s > 6 ? 6 : s;
    const int kl = lane & ((1 << s) - 1), rsub = lane >> s, rpp = 64 >> s;
    for (int ri = wave * rpp + rsub; ri < nr; ri += BLUR_W * rpp) {
      float *rowp = win + __umul24((unsigned)ri, (unsigned)RW);
      for (int k = kl; k < twoR; k += 1 << s) {
        if (k < R) rowp[k] = rowp[R];
        else rowp[P + k] = rowp[R + P - 1];
      }
    }
  }
  __syncthreads();
  if (jb.ro1 >= 0) { blur_rows_from_lds(bt, win, sneed, taps, dst + bt.dstOfs, NC); return; }
  // the whole (small) window is here: its row-filtered block stays in LDS and the column filter follows at once -- arena B
  // is not touched either
  blur_rows_from_lds(bt, win, sneed, taps, aux + R * FC_LS, FC_LS);
  __syncthreads();
  for (int i = threadIdx.x; i < R * FC_LS; i += BLUR_T) {   // replicated border rows above and below
    const int k = i / FC_LS, x = i - k * FC_LS;
    aux[k * FC_LS + x] = aux[R * FC_LS + x];
    aux[(R + P + k) * FC_LS + x] = aux[(R + P - 1) * FC_LS + x];
  }
  __syncthreads();
  blur_cols_from_lds<FC_LS>(NC, n, 0, NC, -R, bt.magic, aux, sneed, taps + bt.tapOfs, dstGrid + jb.gridOfs);
}

template <int LS>   // LDS row stride (floats), a compile-time constant so that tap j of a column is an immediate offset
__device__ __forceinline__ void blur_cols_tile(const BlurTile &bt, float *win, int *sneed, const float *__restrict__ taps,
                                               const int *__restrict__ needTab, const float *__restrict__ src,
                                               float *__restrict__ dst) {
  const int P = bt.P, NC = bt.NC;
  const int n = bt.n;
  const int ro0 = bt.first, nro = bt.count, lo = bt.lo, S = bt.span;
  const int lane = threadIdx.x & 63, wave = wave_id();   // (wave-uniform: a parked row's source address is scalar arithmetic)
  for (int i = threadIdx.x; i < NC; i += BLUR_T) sneed[i] = needTab[bt.needOfs + i];
  const float *T = src + bt.srcOfs;   // P x NC
  {   // direct global -> LDS loads, one parked row (64-column chunk) per instruction
    typedef const float __attribute__((address_space(1))) *gptr;
    typedef float __attribute__((address_space(3))) *lptr;
    for (int si = wave; si < S; si += BLUR_W) {
      int rr = lo + si;
      rr = rr < 0 ? 0 : (rr > P - 1 ? P - 1 : rr);
      const float *a = T + (size_t)rr * NC;
      for (int x0 = 0; x0 < NC; x0 += 64) {
        const int x = x0 + lane;
        if (x < NC) __builtin_amdgcn_global_load_lds((gptr)(a + x), (lptr)(win + si * LS + x0), 4, 0, 0);
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  __syncthreads();
  blur_cols_from_lds<LS>(NC, n, ro0, nro, lo, bt.magic, win, sneed, taps + bt.tapOfs, dst + bt.dstOfs);
}

__global__ __launch_bounds__(BLUR_T) void k_blur_cols_lds(const BlurTile *__restrict__ tiles, const float *__restrict__ taps,
                                                       const int *__restrict__ needTab, const float *__restrict__ src,
                                                       float *__restrict__ dst) {
  const BlurTile bt = tiles[xcd_swizzle(blockIdx.x, gridDim.x)];
  __shared__ __attribute__((aligned(16))) float win[BLUR_LDS_C];
  __shared__ int sneed[96];
  if (bt.NC <= 64) blur_cols_tile<64>(bt, win, sneed, taps, needTab, src, dst);
  else blur_cols_tile<96>(bt, win, sneed, taps, needTab, src, dst);
}

// --- stage 3: 41x41 patch, photometric normalisation, SIFT histogram ----------------------------
// One 128-thread workgroup per region.  Everything order-dependent in the reference is kept in its
// order but fed from LDS so that the serial chains are pure dependent adds:
//   * sample coordinates: 41 lanes run the f32 running sums of interpolate() and park (WX, WY) in LDS,
//     then all 128 threads take the 1681 bilinear taps in parallel;
//   * photometricallyNormalize: the masked pixels (a disc, 1257 of 1681) are compacted in raster order
//     and summed by one lane with 16-byte LDS reads (mean, then variance);
//   * samplePatch: thread t owns bin t = 32*rb + 8*cb + ob and walks the 16x16 pixel block that can
//     reach it (rows 8rb..8rb+15, cols 8cb..8cb+15) in raster order with an f64 accumulator.  For rows
//     8rb..8rb+7 the bin is the pixel's bin1 (weight w1), for rows 8rb+8..8rb+15 its bin0 (weight w0);
//     same for columns -- this is precomputeBinsAndWeights (siftdesc.cpp:22-71) for 4 spatial bins
//     and patch 41, where step = 5/40 makes xi = i/8.
constexpr int PS = 41, NPX = PS * PS;

struct SiftConst {
  int nmask;   // number of pixels with mask > 0
};

// DR regions per workgroup, 128 threads (two wavefronts) each.  The regions run side by side and meet at the same barriers;
// what they share is the instruction stream of the serial chains: the 2 x 1257 ordered adds of the photometric
// normalisation (and the 128 of the RootSIFT L1 norm) are run by lanes 0 .. DR-1 of wavefront 0, one region per lane, so a
// chain costs its issue slots once per workgroup instead of once per region (a quarter of the kernel's vector instructions
// were these one-lane chains).  LDS per region is unchanged, so the same number of regions is resident per CU.
#ifndef MODSX_DESCRIBE_REGIONS
#define MODSX_DESCRIBE_REGIONS 2
#endif
constexpr int DR = MODSX_DESCRIBE_REGIONS;

__global__ __launch_bounds__(128 * DR) void k_describe(const DescJob *jobs, int n, const ImgRef *imgs, const float *grid,
                                                  const int *needTab, const float *coordTab,
                                                  const float *mask, const unsigned short *maskIdx,
                                                  const float *oTab, const int *binTab, const double *wTab,
                                                  SiftConst sc, int photoNorm, int descTypes, int nOut, double maxBin,
                                                  DescOut outs) {
  const int tidw = threadIdx.x;                                    // thread of the workgroup
  const int reg = __builtin_amdgcn_readfirstlane(tidw >> 7);       // region of the workgroup (a wavefront is inside one region)
  const int tid = tidw & 127;                                      // thread of the region
  const int kreal = blockIdx.x * DR + reg;
  const bool alive = kreal < n;                                    // a region past the end repeats the last one and stores nothing
  const int k = alive ? kreal : n - 1;
  // 15.8 KB of LDS per region (it was 19.7: four workgroups of two regions per CU; now five, and the kernel's time follows its
  // residency -- its phases are chains of dependent LDS / global accesses, not issue-bound).  Two pixel arrays, both unpadded
  // (index p = 41 r + c):
  //   bufO: the patch; after the gradients the pixel's orientation o = 8 (ori + 2 pi) / (2 pi) (staged in registers first) -- its
  //         bin and the two orientation weights are formed from it where they are used (the gather's forming step);
  //   bufA: WX (direct branch) / the resampling table (grid branch), the compacted masked values, later val = mask * |grad|.
  constexpr int NPXP = (NPX + 3) & ~3;
  __shared__ __attribute__((aligned(16))) float buf_[DR][2][NPXP];   // [0] bufO, [1] bufA: contiguous (the order-free gather's
                                                                     // accumulators span both)
  float *const patch = buf_[reg][0], *const bufO = buf_[reg][0];
  constexpr int PER_T = (NPX + 127) / 128;
  float *const bufA = buf_[reg][1];
  __shared__ __attribute__((aligned(16))) double slut_[DR][256];   // the descriptor vector and its partial sums (2 KB)
  __shared__ __attribute__((aligned(16))) unsigned char sbs_[DR][4 * 64];   // gather: orientation bin bo0 % 8 of the step's 4 x 64 slots
  // precomputeBinsAndWeights (siftdesc.cpp:22-71) per row / column index i: {w0[i], w1[i], the smaller nonzero one of the two (1
  // if both are 0), bin0 | bin1 << 2 with the bins in 0..3}; one table for the workgroup
  __shared__ float4 stab[PS];
  // the resampling table of the grid branch (smap, sfr) lies in bufA, which that branch does not use before the normalisation
  int4 *const smap = reinterpret_cast<int4 *>(bufA);
  float *const sfr = bufA + 168;
  static_assert(PS * 16 <= 168 * 4 && (168 + PS) <= NPXP, "smap (41 x int4) and sfr (41 floats) fit in bufA");
  double *const vec = slut_[reg], *const part = slut_[reg] + 128;
  __shared__ float sstat_[DR][2];
  float *const sstat = sstat_[reg];
  __shared__ double sfac_[DR];
  __shared__ int schanged_[DR];
#ifdef MODSX_DESCRIBE_PAD
  __shared__ volatile char spad_[MODSX_DESCRIBE_PAD];
  if (tidw == 0) spad_[MODSX_DESCRIBE_PAD - 1] = 1;
#endif
  const DescJob jb = jobs[k];
  if (tidw < PS) {
    const float w0 = (float)wTab[tidw], w1 = (float)wTab[PS + tidw];   // const float wr0 = w0[r] (siftdesc.cpp:79-81)
    const int q = tidw >> 3, b0 = min(max(q - 1, 0), 3), b1 = min(q, 3);   // a bin outside 0..3 comes with weight 0
    const float wm = w0 > 0 ? (w1 > 0 ? fminf(w0, w1) : w0) : (w1 > 0 ? w1 : 1.f);
    stab[tidw] = make_float4(w0, w1, wm, __int_as_float(b0 | (b1 << 2)));
  }
  if (jb.P > 0) {
    // interpolate(smoothed, P/2, P/2, i2p, 0, 0, i2p, patch) (synth-detection.hpp:211-212) on the compact blurred grid:
    // the host ran the coordinate recurrences once per window size; sample (j, i) blends grid rows idx(y0_j), idx(y0_j+1)
    // and columns idx(x0_i), idx(x0_i+1) with wx = WX_i - x0_i, wy = WY_j - y0_j -- the expression of helpers.cpp:575-577.
    const float *G = grid + jb.gridOfs;
    const int NC = jb.NC;
    // the 41 x {idx0, idx1, x0, valid} table and the 41 coordinates are parked in LDS first, so that a sample's four grid
    // loads depend on nothing but the job; all PER_T samples of a thread are then in flight together
    if (tid < PS) {
      const int *mp = needTab + jb.needOfs + NC + 4 * tid;
      const int4 m = make_int4(mp[0], mp[1], mp[2], mp[3]);
      smap[tid] = m;
      sfr[tid] = coordTab[jb.coordOfs + tid] - (float)m.z;     // wx_i = WX_i - x0_i (= wy for rows)
    }
    __syncthreads();
    // two halves of seven samples: 28 grid values in flight per thread instead of 56 -- the registers of the second 28 were what
    // held the kernel at four wavefronts per SIMD (117 VGPRs) once its LDS allowed five workgroups per CU
    constexpr int HALF_T = PER_T / 2;
    static_assert(PER_T == 2 * HALF_T, "an even number of samples per thread");
    // Addresses: 32-bit BYTE offsets from the region's (wave-uniform) grid base -- one 24-bit multiply per grid row and an
    // add + shift per load, the loads in the scalar base + vector offset form (the size_t row products this replaces were
    // quarter-rate 32-bit multiplies and 64-bit adds: two thirds of the issue slots of the sampling).  An entry that is not valid
    // has idx0 = idx1 = 0 in the host's table (engine.hip), so every address is inside the grid whatever `valid` says and the
    // select happens once, on the value.  (r, c) of p = tid + 128 k advance by (3, 5): 128 = 3 * 41 + 5 -- no division per sample.
    typedef const char __attribute__((address_space(1))) *gbyte_p;
    const gbyte_p Gb = (gbyte_p)as_global(G);
    const unsigned NCu = (unsigned)NC;
    int rs = tid / PS, cs = tid - rs * PS;
#pragma unroll
    for (int h = 0; h < 2; h++) {
      float g00[HALF_T], g01[HALF_T], g10[HALF_T], g11[HALF_T];
      int rr[HALF_T], cc[HALF_T];
#pragma unroll
      for (int kk = 0; kk < HALF_T; kk++) {
        const int r = rs < PS ? rs : PS - 1, c = rs < PS ? cs : PS - 1;   // p >= NPX: the last pixel again, not stored
        rr[kk] = r; cc[kk] = c;
        const int4 mr = smap[r], mc = smap[c];
        const unsigned o0 = __umul24((unsigned)mr.x, NCu), o1 = __umul24((unsigned)mr.y, NCu);
        g00[kk] = *(gcfloat_p)(Gb + ((o0 + (unsigned)mc.x) << 2)); g01[kk] = *(gcfloat_p)(Gb + ((o0 + (unsigned)mc.y) << 2));
        g10[kk] = *(gcfloat_p)(Gb + ((o1 + (unsigned)mc.x) << 2)); g11[kk] = *(gcfloat_p)(Gb + ((o1 + (unsigned)mc.y) << 2));
        cs += 128 - 3 * PS; rs += 3;     // p += 128
        if (cs >= PS) { cs -= PS; rs++; }
      }
      if (h == 0) __syncthreads();   // (the direct branch of another region of the workgroup has a barrier here)
#pragma unroll
      for (int kk = 0; kk < HALF_T; kk++) {
        const int p = tid + 128 * (h * HALF_T + kk);
        if (p < NPX) {
          const int r = rr[kk], c = cc[kk];
          float v = 0.f;
          if (smap[r].w && smap[c].w) {
            const float wx = sfr[c], wyd = sfr[r];
            const float I1 = wx * (g01[kk] - g00[kk]) + g00[kk];
            v = wyd * (wx * (g11[kk] - g10[kk]) + g10[kk] - I1) + I1;
          }
          patch[p] = v;
        }
      }
    }
  } else {
    // -- direct branch (imageToPatchScale <= 0.4 or fast extraction): interpolate() straight from the view
    const ImgRef im = imgs[jb.img];
    const float *src = im.d;
    const int srows = im.rows, scols = im.cols;
    const float ox = jb.x, oy = jb.y, a11 = jb.a11, a12 = jb.a12, a21 = jb.a21, a22 = jb.a22;
    const bool touch = check_borders(scols, srows, ox, oy, a11, a12, a21, a22, PS, PS);
    if (tid < PS) {
      const int half = PS >> 1;
      float rx = ox - (float)half * a12;
      float ry = oy - (float)half * a22;
      for (int j = 0; j < tid; j++) { rx += a12; ry += a22; }
      float WX = rx - (float)half * a11;
      float WY = ry - (float)half * a21;
#pragma unroll 1
      for (int i = 0; i < PS; i++) {
        bufA[tid * PS + i] = WX;
        bufO[tid * PS + i] = WY;
        WX += a11;
        WY += a21;
      }
    }
    __syncthreads();
    float sv[PER_T];
#pragma unroll
    for (int k = 0; k < PER_T; k++) {
      const int p = tid + 128 * k;
      const int r = p / PS, c = p - r * PS;
      sv[k] = p < NPX ? bilinear_tap(as_global(src), srows, scols, bufA[r * PS + c], bufO[r * PS + c], touch) : 0.f;
    }
    __syncthreads();   // every WY coordinate has been consumed; the samples may overwrite them
#pragma unroll
    for (int k = 0; k < PER_T; k++) {
      const int p = tid + 128 * k;
      if (p < NPX) patch[p] = sv[k];
    }
  }
  __syncthreads();
  // -- photometricallyNormalize (helpers.cpp:666-715): f32 running sums over the masked pixels
  if (photoNorm) {
    const int nm = sc.nmask, nm4 = (nm + 3) & ~3;
    for (int i = tid; i < nm4; i += 128) bufA[i] = i < nm ? patch[maskIdx[i]] : 0.f;
    __syncthreads();
    if (tidw < DR) {   // lane q of wavefront 0 runs region q's chain
      float sum = 0.f;
      const float *vals = buf_[tidw][1];
      const float4 *v4 = reinterpret_cast<const float4 *>(vals);
      const int full = nm >> 2;
#pragma unroll 8
      for (int i = 0; i < full; i++) { const float4 v = v4[i]; sum += v.x; sum += v.y; sum += v.z; sum += v.w; }
      for (int i = full * 4; i < nm; i++) sum += vals[i];
      const float gsum = (float)nm;   // gsum++ per masked pixel: every partial count < 2^24 is exact in f32
      sstat_[tidw][0] = sum / gsum;
      sstat_[tidw][1] = gsum;
    }
    __syncthreads();
    const float mean = sstat[0];
    for (int i = tid; i < nm4; i += 128) { const float d = mean - bufA[i]; bufA[i] = i < nm ? d * d : 0.f; }
    __syncthreads();
    if (tidw < DR) {
      float var = 0.f;
      const float *vals = buf_[tidw][1];
      const float4 *v4 = reinterpret_cast<const float4 *>(vals);
      const int full = nm >> 2;
#pragma unroll 8
      for (int i = 0; i < full; i++) { const float4 v = v4[i]; var += v.x; var += v.y; var += v.z; var += v.w; }
      for (int i = full * 4; i < nm; i++) var += vals[i];
      sstat_[tidw][1] = sqrtf(var / sstat_[tidw][1]);
    }
    __syncthreads();
    const float var = sstat[1];
    if (!((double)var < 0.0001)) {
      const float fac = 50.0f / var;
      for (int i = tid; i < NPX; i += 128) {
        float v = 128.f + fac * (patch[i] - mean);
        if (v > 255.f) v = 255.f;
        if (v < 0.f) v = 0.f;
        patch[i] = v;
      }
    }
    __syncthreads();
  }
  // -- gradients, orientation, per-pixel weights (siftdesc.cpp:346-379, 73-131)
  float ov[PER_T], vv[PER_T];   // o and val = mask * |grad| of the thread's pixels p = tid + 128 k
#pragma unroll
  for (int k = 0; k < PER_T; k++) {
    const int p = tid + 128 * k;
    ov[k] = 0.f; vv[k] = 0.f;
    if (p >= NPX) continue;
    const int r = p / PS, c = p - r * PS;
    // one-sided differences on the patch's frame, central ones inside (siftdesc.cpp:346-360): the neighbour that does not
    // exist is the pixel itself -- no per-pixel branch
    const float xg = patch[c == PS - 1 ? p : p + 1] - patch[c == 0 ? p : p - 1];
    const float yg = patch[r == PS - 1 ? p : p + PS] - patch[r == 0 ? p : p - PS];
    const float g = sqrtf(xg * xg + yg * yg);
    // o = (float)(8 * (ori + 2 pi) / (2 pi)) of ori = atan2LUTff(yg, xg) (siftdesc.cpp:103-110): the angle takes one of 8 x 256 + 1
    // values, so o comes from a table built with that f64 expression (engine.hip: upload_tables) -- no f64 look-up, add and
    // division per pixel
    int code, idx;
    const bool special = atan2lut_case(yg, xg, code, idx);
    // val = (float)(0.0 + (1.0 * (double)mask) * (double)g) in the reference: the f64 product of two f32 values is exact (48
    // significant bits), so rounding it to f32 IS the f32 product; mask, g >= 0, so the + 0.0 changes nothing
    vv[k] = mask[p] * g;         // the column weights wc0 / wc1 = (float)(w[c] * val) are formed in the gather
    ov[k] = oTab[special ? 2048 : code * 256 + idx];   // bo0 = (int)o and wo1 = o - bo0 (siftdesc.cpp:111-117) are formed in the gather
  }
  // -- samplePatch, order-free form.  vec[bin] += val * wo (siftdesc.cpp:73-131) adds f32 terms >= +0 to an f64 accumulator in
  // raster order.  Scale the region by a power of two S (exact) so that every nonzero term is an INTEGER (x >= 2^23 makes an f32
  // one) and add the terms as 64-bit integers -- associative, so any order and any number of partial accumulators give the
  // exact sum X.  If X < 2^53, every partial sum of the reference's loop is an integer below 2^53 too, i.e. exactly
  // representable: no add of that loop ever rounds and its result is X / S.  Otherwise (a term below 2^23 after scaling, a bin
  // at 2^53 or more, no usable S) the workgroup runs the ordered gather below: 1 % of the regions -- a pixel pair one ulp apart
  // under a small orientation weight, against a bin 2^23 times larger.
  //   S = 2^(47 - E) for max val < 2^E: every term is below 2^47 (32 of them fit under the 2^52 conversion constant), 5 binades of room for a bin
  //   against the largest val (2^(E + 6) was never reached on 10^4 regions, 2^(E + 5) by 2 in 10^4), 23 binades below it for the
  //   smallest term.
  // Every thread adds the 8 terms of each of its 14 pixels straight to the bins with LDS 64-bit integer atomics (twice the
  // rate of f64 ones here: tools/ubench/lds_atomics.hip): 8 copies of the bins, one per column mod 8 and skewed by one 8-byte
  // bank (neighbouring pixels mostly share a bin), 9 orientation slots per cell so that bin b0 + 1 is always the next word
  // (slot 8 is slot 0's).  A quarter of the ordered gather's vector instructions, no barrier per step.
  constexpr int ACS = 145, NAC = 8 * ACS;                      // copy stride and total in 8-byte words
  static_assert(NAC * 2 <= 2 * NPXP, "8 skewed copies of 16 x 9 bins fit in bufO + bufA");
  unsigned long long *const acc = reinterpret_cast<unsigned long long *>(buf_[reg][0]);
  __shared__ unsigned smin_[DR][2], smax_[DR][2];
  {
    unsigned mb = 0;               // val >= +0: the bit patterns order like the values, and a NaN or an infinity (which end in
                                   // the ordered gather) above all finite ones
#pragma unroll
    for (int k = 0; k < PER_T; k++) mb = max(mb, __float_as_uint(vv[k]) & 0x7fffffffu);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) mb = max(mb, (unsigned)__shfl_xor((int)mb, off));
    if ((tid & 63) == 0) smax_[reg][tid >> 6] = mb;
  }
  __syncthreads();   // all gradients taken: the patch and bufA are dead
  {
    float4 *const z = reinterpret_cast<float4 *>(buf_[reg][0]);
    for (int i = tid; i < (NAC * 2 + 3) / 4; i += 128) z[i] = make_float4(0.f, 0.f, 0.f, 0.f);
  }
  const unsigned mxb = max(smax_[reg][0], smax_[reg][1]);
  const int E = (int)(mxb >> 23) - 126;                          // max val in [2^(E - 1), 2^E)
  const bool scalable = mxb >= 0x00800000u && mxb < 0x7f800000u && E >= -60 && E <= 100;   // (a region of zeros: ordered, it is rare)
  const float S = __uint_as_float((unsigned)(127 + 47 - (scalable ? E : 47)) << 23);
  __syncthreads();
  unsigned tmin = 0xffffffffu;     // (bits of the smallest nonzero scaled term) - 1: +0 wraps to the maximum and never wins
  {
    int r = tid / PS, c = tid - r * PS;
#pragma unroll
    for (int k = 0; k < PER_T; k++) {
      const float val = vv[k] * S;   // exact: a power of two, no overflow, and scaling up makes no denormal
      if (val > 0) {                 // not: outside the mask's disc (a quarter of the pixels), flat, past the patch, NaN
        const float o = ov[k];
        const int bo0 = (int)o;
        const float wo1 = o - (float)bo0, wo0 = 1.0f - wo1;
        const float4 tr = stab[r], tc = stab[c];
        const int pr = __float_as_int(tr.w), pc = __float_as_int(tc.w);
        // wc = (float)(w[c] * (double)val): the weights are multiples of 1/8, so the f64 product is exact and its rounding to f32
        // is the f32 product (checked where the table is built)
        const float wc0 = tc.x * val, wc1 = tc.y * val;
        // byte offsets: copy (c & 7), cell (row bin, column bin), slot b0 = bo0 % 8 = bo0 & 7 (o >= 4: ori >= -pi)
        const int ob = (c & 7) * (ACS * 8) + (bo0 & 7) * 8;
        const int r0 = (pr & 3) * 288, r1 = ((pr >> 2) & 3) * 288, c0 = (pc & 3) * 72 + ob, c1 = ((pc >> 2) & 3) * 72 + ob;   // (& 3: the bins are 0..3 -- a 24-bit multiply to the compiler)
        typedef float v2f __attribute__((ext_vector_type(2)));
        const v2f wcv = {wc0, wc1}, wov = {wo0, wo1};
        const v2f va = wcv * tr.x, vb = wcv * tr.y;                      // (row bin0 | bin1) x (column bin0, bin1)
        const v2f ts[4] = {wov * va.x, wov * va.y, wov * vb.x, wov * vb.y};   // the slot's terms for bins b0, b0 + 1
        const int bs[4] = {r0 + c0, r0 + c1, r1 + c0, r1 + c1};
#pragma unroll
        for (int q = 0; q < 4; q++) {
          // x + 2^52 of an integer x < 2^52 holds x in its low 52 bits.  The words are added as they are: a (copy, bin) word takes
          // at most one term of each of the 2 x 16 pixels of its column pair, 32 x 2^47 <= 2^52, so the low 52 bits of the word
          // are the sum of its terms and the exponent fields pile up above them, to be masked off by the merge
          const double d0 = (double)ts[q].x + 4503599627370496.0, d1 = (double)ts[q].y + 4503599627370496.0;
          unsigned long long *const a = reinterpret_cast<unsigned long long *>(reinterpret_cast<char *>(acc) + bs[q]);
          __hip_atomic_fetch_add(a, (unsigned long long)__double_as_longlong(d0), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          __hip_atomic_fetch_add(a + 1, (unsigned long long)__double_as_longlong(d1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        // the smallest nonzero term of the pixel: products of non-negative factors round monotonically, so it is the one of
        // the smallest nonzero row weight, column weight and orientation weight
        const float wom = wo1 > 0 ? fminf(wo0, wo1) : wo0;
        tmin = min(tmin, __float_as_uint((tr.z * (tc.z * val)) * wom) - 1u);
      }
      c += 128 - 3 * PS; r += 3;     // p += 128
      if (c >= PS) { c -= PS; r++; }
    }
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) tmin = min(tmin, (unsigned)__shfl_xor((int)tmin, off));
  if ((tid & 63) == 0) smin_[reg][tid >> 6] = tmin;
  __syncthreads();
  double binT;
  bool inexact;
  {
    const int cell = tid >> 3, slot = tid & 7;
    unsigned long long X = 0;
#pragma unroll
    for (int q = 0; q < 8; q++) {
      constexpr unsigned long long M52 = (1ull << 52) - 1ull;
      X += acc[q * ACS + cell * 9 + slot] & M52;
      if (slot == 0) X += acc[q * ACS + cell * 9 + 8] & M52;
    }
    const unsigned m = min(smin_[reg][0], smin_[reg][1]);   // 0xffffffff: no nonzero term at all (every bin is +0)
    inexact = !scalable || (m != 0xffffffffu && m + 1u < 0x4b000000u) || (X >> 53) != 0;
    // X < 2^53: the two halves convert exactly and so does their sum; the division by S is a change of exponent
    binT = ldexp((double)(unsigned)(X >> 32) * 4294967296.0 + (double)(unsigned)X, (scalable ? E : 47) - 47);
  }
#ifdef MODSX_DESC_ORDERED_ONLY   // test build: every workgroup takes the ordered gather
  inexact = true;
#endif
  const int ordered = __syncthreads_or(inexact);   // the regions of a workgroup share the barriers of the ordered gather
  if (ordered) {
#pragma unroll
    for (int k = 0; k < PER_T; k++) {
      const int p = tid + 128 * k;
      if (p < NPX) { bufA[p] = vv[k]; bufO[p] = ov[k]; }
    }
    __syncthreads();
  // -- samplePatch: a bin gathers its 16x16 pixel block in raster order.  ONE wavefront per region does it, a lane owning two
  // neighbouring orientation bins of a spatial cell: a pixel's orientation falls into bin b0 with weight 1 - wo1 and into
  // b0 + 1 with wo1, so the lane of bins (2p, 2p + 1) takes something from the pixels with b0 in {2p - 1, 2p, 2p + 1} -- the same
  // 256 visits as a lane with one bin, for two running sums.  The region's other wavefront only helps forming the weighted
  // values of a step and waits at the barriers, which costs no issue slots: a third fewer vector instructions for the gather
  // (two thirds of this kernel) than 128 single-bin lanes.  Every bin still adds its terms in raster order, +0.0 for pixels
  // that do not belong to it.
  {
    const bool gth = tid < 64;
    const int rb = (tid >> 4) & 3, cb = (tid >> 2) & 3, oa = 2 * (tid & 3), ob = oa + 1, oam = (oa + 7) & 7;
    double accA = 0.0, accB = 0.0;
    float(*const sv0)[64] = reinterpret_cast<float(*)[64]>(vec);       // v (1 - wo1) of the step's 4 rows x 64 column slots (the descriptor
                                                                       // vector itself is written when the loop is over)
    float(*const sv1)[64] = reinterpret_cast<float(*)[64]>(part);      // v wo1 (the norm's scratch is idle until the gather is over)
    unsigned char(*const sbs)[64] = reinterpret_cast<unsigned char(*)[64]>(sbs_[reg]);
    // Step rr touches rows 8 rb + rr (rb = 0..3) with one row weight each; a pixel's column weight is w1[c] in the block
    // whose first half holds column c and w0[c] in the block whose second half does.  The product
    // wr * (float)(w[c] * val) is the same for the orientation lanes of a bin block, so the 128 threads first form the
    // 4 rows x (w1: columns 0..31, w0: columns 8..39) values of the step once (clamped to 0 when not > 0: such a pixel
    // adds nothing in the reference, and +0.0 here), times the two orientation weights, and the bins then read them.
#pragma unroll 1
    for (int rr = 0; rr < 16; rr++) {
#pragma unroll
      for (int u = 0; u < 2; u++) {
        const int i = tid + 128 * u, rbi = i >> 6, vc = i & 63;
        const int col = vc < 32 ? vc : vc - 24, r = 8 * rbi + rr;
        const float wrr = rr < 8 ? stab[r].y : stab[r].x;
        // wc0 / wc1 = (float)(w[c] * (double)val) in the reference: the weights are multiples of 1/8 (exact in f32, checked
        // where the table is built), so the f64 product is exact and its rounding to f32 is the f32 product
        const float wcv = (vc < 32 ? stab[col].y : stab[col].x) * bufA[r * PS + col];
        const float v = wrr * wcv;
        const float vcl = v > 0 ? v : 0.f;
        const float o = bufO[r * PS + col];
        const int bo0 = (int)o;
        const float wo1 = o - (float)bo0;      // formed per use (2.4 times per pixel) instead of being kept per pixel
        sv0[rbi][vc] = vcl * (1.0f - wo1);
        sv1[rbi][vc] = vcl * wo1;
        sbs[rbi][vc] = (unsigned char)(bo0 & 7);   // o >= 4 (ori >= -pi), so bo0 % 8 = bo0 & 7
      }
      __syncthreads();
      if (gth) {
#pragma unroll
        for (int seg = 0; seg < 4; seg++) {
          const int sx = (seg < 2 ? 0 : 24) + 8 * cb + 4 * seg;
          const float4 p0v = *reinterpret_cast<const float4 *>(&sv0[rb][sx]);
          const float4 p1v = *reinterpret_cast<const float4 *>(&sv1[rb][sx]);
          const unsigned b4 = *reinterpret_cast<const unsigned *>(&sbs[rb][sx]);
          const float p0s[4] = {p0v.x, p0v.y, p0v.z, p0v.w};
          const float p1s[4] = {p1v.x, p1v.y, p1v.z, p1v.w};
#pragma unroll
          for (int e = 0; e < 4; e++) {
            const int b0 = (int)((b4 >> (8 * e)) & 0xff);
            // bin b0 takes v (1 - wo1), bin (b0 + 1) % 8 takes v wo1
            const bool isA = b0 == oa, isAm = b0 == oam, isB = b0 == ob;
            const float ta = isA ? p0s[e] : (isAm ? p1s[e] : 0.f);
            const float tb = isB ? p0s[e] : (isA ? p1s[e] : 0.f);
            accA += (double)ta;
            accB += (double)tb;
          }
        }
      }
      __syncthreads();
    }
    if (gth) { vec[(rb * 4 + cb) * 8 + oa] = accA; vec[(rb * 4 + cb) * 8 + ob] = accB; }
  }
  } else vec[tid] = binT;
  __syncthreads();
  // The descriptors of one step share everything up to here (SIFTDescriptor::operator(), siftdesc.cpp:401-442: RootSIFT and
  // HalfRootSIFT run the same computeRootSiftDescriptor on the same patch and differ in the fold and the norm), so a step
  // with several descriptor classes (iters_mods_cviu_wxbs.ini:35: RootSIFT, HalfRootSIFT) emits all of them from one
  // histogram: output t has type (descTypes >> 4 t) & 15 and goes to outs.u8 (t = 0) or outs.u8x[t - 1].
  const double rawBin = vec[tid];
#pragma unroll 1
  for (int ot = 0; ot < nOut; ot++) {
  const int descType = (descTypes >> (4 * ot)) & 15;
  if (ot) {
    __syncthreads();
    vec[tid] = rawBin;
    __syncthreads();
  }
  const bool rootsift = (descType & 1) != 0;
  if (descType >= 2) {
    // HalfSIFT / HalfRootSIFT (siftdesc.cpp:412-433): opposite orientation bins are folded before the norm.  The upper
    // 64 entries become +0, which every later sum, clip and quantisation leaves untouched.
    const int sb = tid >> 2, j = tid & 3;
    const double h = tid < 64 ? vec[sb * 8 + j] + vec[sb * 8 + j + 4] : 0.0;
    __syncthreads();
    vec[tid] = h;
    __syncthreads();
  }
  // -- normalize / clip / renormalize (siftdesc.cpp:136-158, 199-221, 247-262).  The reference leaves after the first pass
  // when nothing was clipped; here such a region sits out the second pass (the regions of a workgroup share the barriers).
  bool active = true;
  for (int pass = 0; pass < 2; pass++) {
    if (tid < 32 && active) {
      const double s0 = vec[4 * tid] * vec[4 * tid], s1 = vec[4 * tid + 1] * vec[4 * tid + 1],
                   s2 = vec[4 * tid + 2] * vec[4 * tid + 2], s3 = vec[4 * tid + 3] * vec[4 * tid + 3];
      part[tid] = s0 + s1 + s2 + s3;
    }
    __syncthreads();
    if (tid == 0 && active) {
      double len = 0.0;
#pragma unroll
      for (int i = 0; i < 32; i++) len += part[i];
      len = sqrt(len);
      sfac_[reg] = 1.0 / len;
      schanged_[reg] = 0;
    }
    __syncthreads();
    if (active) vec[tid] *= sfac_[reg];
    __syncthreads();
    if (pass == 0) {
      if (vec[tid] > maxBin) { vec[tid] = maxBin; schanged_[reg] = 1; }
      __syncthreads();
      if (!schanged_[reg]) active = false;
    }
  }
  __syncthreads();
  if (rootsift) {
    if (tidw < DR) {   // lane q of wavefront 0: the L1 norm of region q, in index order
      const double *vq = slut_[tidw];
      double sum = 0.;
#pragma unroll 16
      for (int i = 0; i < 128; i++) sum += fabs(vq[i]);
      sfac_[tidw] = sum;
    }
    __syncthreads();
    vec[tid] = sqrt(vec[tid] / sfac_[reg]);
  }
  {
    int b;
    if (rootsift) b = (int)(512.0 * vec[tid] + 0.5);
    else b = (int)((double)512.0f * vec[tid] + 0.5);
    b = b < 255 ? b : 255;
    b = b > 0 ? b : 0;
    if (alive) {
      if (ot == 0) {
        outs.f[jb.img][(size_t)jb.outIdx * 128 + tid] = (float)b;
        outs.u8[jb.img][(size_t)jb.outIdx * 128 + tid] = (uint8_t)b;
      } else
        outs.u8x[ot - 1][jb.img][(size_t)jb.outIdx * 128 + tid] = (uint8_t)b;
    }
  }
  }   // ot
}

void launch_expand_tiles(hipStream_t s, const int *prefix, int nJobs, int *tileJob) {
  if (nJobs <= 0) return;
  hipLaunchKernelGGL(k_expand_tiles, dim3(nJobs), dim3(64), 0, s, prefix, nJobs, tileJob);
}
void launch_patch_sample(hipStream_t s, const DescJob *jobs, const int *tilePrefix, const int *tileJob, int nTiles,
                         const ImgRef *imgs, float *scratch) {
  if (nTiles <= 0) return;
  hipLaunchKernelGGL(k_patch_sample, dim3(8 * ((nTiles + 7) / 8)), dim3(64), 0, s, jobs, tilePrefix, tileJob, imgs, scratch, nTiles);
}
void launch_expand_blur_tiles(hipStream_t s, const DescJob *jobs, const int *prefixRows, const int *prefixCols, int nJobs,
                              const int *needTab, BlurTile *tilesRows, BlurTile *tilesCols, float2 *rowStarts) {
  if (nJobs > 0) MX_DUP(K_PATCH_SAMPLE) hipLaunchKernelGGL(k_expand_blur_tiles, dim3((nJobs + 255) / 256, 3), dim3(256), 0, s, jobs, prefixRows, prefixCols, nJobs, needTab,
                                    tilesRows, tilesCols, rowStarts);
}
void launch_sample_rows(hipStream_t s, const DescJob *jobs, const BlurTile *tiles, int nTiles, const ImgRef *imgs, const float *taps,
                        const int *needTab, float *dst, const float2 *rowStarts, float *dstGrid) {
  if (nTiles <= 0) return;
  MX_DUP(K_BLUR_ROWS) hipLaunchKernelGGL(k_sample_rows_lds, dim3(8 * ((nTiles + 7) / 8)), dim3(BLUR_T), 0, s, tiles, jobs, imgs, taps, needTab, dst, nTiles,
                     rowStarts, dstGrid);
}
void launch_blur_cols(hipStream_t s, const BlurTile *tiles, int nTiles, const float *taps, const int *needTab, const float *src,
                      float *dst) {
  if (nTiles > 0) MX_DUP(K_BLUR_COLS) hipLaunchKernelGGL(k_blur_cols_lds, dim3(nTiles), dim3(BLUR_T), 0, s, tiles, taps, needTab, src, dst);
}
void launch_patch_blur(hipStream_t s, const DescJob *jobs, const int *tilePrefix, const int *tileJob, int nTiles,
                       const float *taps, const int *needTab, const float *src, float *dst, int pass) {
  if (nTiles <= 0) return;
  hipLaunchKernelGGL(k_patch_blur, dim3(nTiles), dim3(256), 0, s, jobs, tilePrefix, tileJob, taps, needTab, src, dst, pass);
}
void launch_describe(hipStream_t s, const DescJob *jobs, int n, const ImgRef *imgs, const float *grid,
                     const int *needTab, const float *coordTab, const float *mask, const unsigned short *maskIdx, int nmask, const float *oTab, const int *bins,
                     const double *wts, int photoNorm, int descTypes, int nOut, double maxBin, const DescOut &outs) {
  if (n <= 0) return;
  SiftConst sc;
  sc.nmask = nmask;
  MX_DUP(K_DESCRIBE) hipLaunchKernelGGL(k_describe, dim3((n + DR - 1) / DR), dim3(128 * DR), 0, s, jobs, n, imgs, grid, needTab, coordTab, mask, maskIdx, oTab, bins,
                     wts, sc,
                     photoNorm, descTypes, nOut, maxBin, outs);
}

}  // namespace mx
