// kernels_pyramid.hip -- scale-space pyramid, Hessian response, 3-D extrema + localisation.
//
// Reference: ScaleSpaceDetector, detectors/affinedetectors/pyramid.cpp
//   gaussianBlur (helpers.cpp:717-724) -> cv::GaussianBlur f32 separable, BORDER_REPLICATE
//   HessianResponse   pyramid.cpp:223-281
//   cv::resize 0.5    pyramid.cpp:520 (area-fast 2x2)
//   findLevelKeypoints + isMax/isMin   pyramid.cpp:432-452, 42-64
//   localizeKeypoint  pyramid.cpp:308-430 (everything except the octaveMap claim and the
//                     powf scale, which the host applies in detection order)
// All kernels take a small by-value batch of jobs; blockIdx.z selects the job, so the two
// images of a pair (or several views) run in one launch.
#include "engine.hpp"

namespace mx {

// ---------------------------------------------------------------------------------------
// Fused separable Gaussian blur (+ optional Hessian response of the blurred tile).
// Tile: 64 x 32 outputs per 256-thread workgroup, staged through LDS (R = ksize / 2 is a template
// parameter, so every tap loop is unrolled and the taps sit in SGPRs):
//   src  (36+2R) x (68+2R)   clamped (replicate) input tile, row stride 84
//   tmp  (36+2R) x 68        row-filtered
//   blr  36 x 68             blurred tile incl. the 1-px ring the 3x3 Hessian needs (aliases src)
// Each thread produces 4 neighbouring outputs per pass from one register window (4 + 2R inputs), which cuts
// the LDS reads per output from 2R+1 to (4+2R)/4.  Row pass accumulates taps left to right (cv RowFilter,
// ksize > 5) or centre + symmetric pairs (SymmRowSmallFilter, ksize <= 5); column pass is centre +
// (below + above) * k (SymmColumnFilter) -- per output the same f32 operation order as the CPU filter engine.
// ---------------------------------------------------------------------------------------
constexpr int TW = 64;
constexpr int TMP_W = TW + 4;                 // 66 needed columns rounded up to a multiple of 4
// The tile height is a template parameter: 32 rows for the large levels (less halo per output), 16 rows when a whole launch
// is under two rounds of 32-row tiles (small octaves: twice the workgroups, shorter per-workgroup chains; 6 % of the
// pyramid's time).  BLR_H = TH + 4: the TH + 2 needed rows rounded up to a multiple of 4.

// The images of a launch set differ in size (a view of tilt t is 1/t of the image): a grid over the largest image would
// launch four empty workgroups for every useful one.  The launch is a flat list of tiles; tile0[] says where each job's
// tiles start (wave-uniform scan over at most MAXB entries of the kernel arguments).
MX_D int find_job(const int *tile0, int nj, int t) {
  int j = 0;
  while (j + 1 < nj && t >= tile0[j + 1]) j++;
  return j;
}

typedef float f2 __attribute__((ext_vector_type(2)));   // two f32 lanes of the packed ALU (v_pk_mul_f32 / v_pk_add_f32)

// Vector issue is what the pipeline's concurrent streams compete for, so both filter passes run on the packed f32 ALU: two
// outputs per instruction, each with the operand order and rounding of the scalar form (no contraction, IEEE mul and add).
//   row pass     pairs the SAME column of two neighbouring ROWS: the input tile is parked row-pair interleaved
//                (src2[pair][x] = {row 2 pair, row 2 pair + 1}), so a thread's register window is already made of pairs;
//   column pass  pairs two neighbouring COLUMNS of the row-filtered tile (an aligned 8-byte LDS read per window element).
template <int R, int TH>
__global__ __launch_bounds__(256) void k_blur_hess(BlurBatch batch) {
  constexpr int BLR_H = TH + 4;
  const int ji = find_job(batch.tile0, batch.nj, blockIdx.x);
  const BlurJob jb = batch.j[ji];
  const int rows = jb.rows, cols = jb.cols;
  const int lt = blockIdx.x - batch.tile0[ji], ntx = (cols + TW - 1) / TW;
  const int tx0 = (lt % ntx) * TW, ty0 = (lt / ntx) * TH;
  constexpr int n = 2 * R + 1;
  constexpr int sh = BLR_H + 2 * R, sw = TMP_W + 2 * R;
  static_assert(sh % 2 == 0, "the input tile is parked as row pairs");
  // sized for this instantiation's halo: 25 KB (R = 4) .. 31.6 KB (R = 8), i.e. 6 workgroups per CU for the two
  // smallest kernels of an octave instead of 5 -- octave 0 of a batch of 8 images is 3072 workgroups, two full rounds
  constexpr int SW = (sw + 3) & ~3;   // pairs per pair-row: a pair-row is 2 SW floats, 16-byte aligned
  __shared__ __attribute__((aligned(16))) float src[sh * SW];   // (sh / 2) pair-rows x SW pairs; later reused as blr
  __shared__ __attribute__((aligned(16))) float tmp[sh * TMP_W];
  float *blr = src;
  const int tid = threadIdx.x;
  // stage 1: clamped input tile
  {   // all loads of a thread are issued before the first LDS write: one memory round trip for the tile
    constexpr int PER = (sh * sw + 255) / 256;
    float t[PER];
#pragma unroll
    for (int u = 0; u < PER; u++) {
      const int i = tid + 256 * u, ii = i < sh * sw ? i : sh * sw - 1;
      const int ly = ii / sw, lx = ii - ly * sw;
      int gy = ty0 - 1 - R + ly, gx = tx0 - 1 - R + lx;
      gy = gy < 0 ? 0 : (gy > rows - 1 ? rows - 1 : gy);
      gx = gx < 0 ? 0 : (gx > cols - 1 ? cols - 1 : gx);
      // 32-bit element offset from the job's (wave-uniform) plane: one 24-bit multiply-add, the load in the scalar base + vector
      // offset form (a level is far below 2^24 px per side and 2^30 px in all, capi.hip image_size_ok)
      t[u] = as_global(jb.src)[__umul24((unsigned)gy, (unsigned)cols) + (unsigned)gx];
    }
#pragma unroll
    for (int u = 0; u < PER; u++) {
      const int i = tid + 256 * u;
      if (i < sh * sw) { const int ly = i / sw, lx = i - ly * sw; src[(ly >> 1) * (2 * SW) + 2 * lx + (ly & 1)] = t[u]; }
    }
  }
  __syncthreads();
  // stage 2: row filter, 4 outputs of 2 rows per thread
  for (int i = tid; i < (sh / 2) * (TMP_W / 4); i += 256) {
    const int lp = i / (TMP_W / 4), g = i - lp * (TMP_W / 4);
    const f2 *S = reinterpret_cast<const f2 *>(src) + lp * SW + 4 * g;   // output x reads S[x .. x + 2R], centre S[x + R]
    f2 w[4 + 2 * R];
#pragma unroll
    for (int q = 0; q < (4 + 2 * R) / 2; q++) {
      const float4 t = *reinterpret_cast<const float4 *>(S + 2 * q);
      w[2 * q] = (f2){t.x, t.y}; w[2 * q + 1] = (f2){t.z, t.w};
    }
    f2 v[4];
#pragma unroll
    for (int o = 0; o < 4; o++) {
      if (n <= 5) {
        v[o] = w[o + R] * batch.k[R];
#pragma unroll
        for (int j = 1; j <= R; j++) v[o] = v[o] + (w[o + R - j] + w[o + R + j]) * batch.k[R + j];
      } else {
        v[o] = (f2){0.f, 0.f};
#pragma unroll
        for (int j = 0; j < n; j++) v[o] = v[o] + w[o + j] * batch.k[j];
      }
    }
    *reinterpret_cast<float4 *>(tmp + (2 * lp) * TMP_W + 4 * g) = make_float4(v[0].x, v[1].x, v[2].x, v[3].x);
    *reinterpret_cast<float4 *>(tmp + (2 * lp + 1) * TMP_W + 4 * g) = make_float4(v[0].y, v[1].y, v[2].y, v[3].y);
  }
  __syncthreads();
  // stage 3: column filter -> blurred tile with 1-px ring, 4 output rows of 2 columns per thread
  for (int i = tid; i < (BLR_H / 4) * (TMP_W / 2); i += 256) {
    const int gy = i / (TMP_W / 2), lx = 2 * (i - gy * (TMP_W / 2));
    const float *S = tmp + (4 * gy) * TMP_W + lx;   // output row y reads rows y .. y + 2R, centre y + R
    f2 w[4 + 2 * R];
#pragma unroll
    for (int q = 0; q < 4 + 2 * R; q++) w[q] = *reinterpret_cast<const f2 *>(S + q * TMP_W);
#pragma unroll
    for (int o = 0; o < 4; o++) {
      f2 v = w[o + R] * batch.k[R] + (f2){0.f, 0.f};
#pragma unroll
      for (int j = 1; j <= R; j++) v = v + (w[o + R + j] + w[o + R - j]) * batch.k[R + j];
      *reinterpret_cast<f2 *>(blr + (4 * gy + o) * TMP_W + lx) = v;
    }
  }
  __syncthreads();
  // stage 4: write blur and response
  const float norm2 = jb.norm * jb.norm;
  for (int i = tid; i < TH * TW; i += 256) {
    const int ly = i / TW, lx = i - ly * TW;
    const int gy = ty0 + ly, gx = tx0 + lx;
    if (gy >= rows || gx >= cols) continue;
    const float *B = blr + (ly + 1) * TMP_W + (lx + 1);
    const unsigned gofs = __umul24((unsigned)gy, (unsigned)cols) + (unsigned)gx;
    jb.blur[gofs] = B[0];
    if (jb.resp) {
      float o = 0.f;
      if (gy >= 1 && gy < rows - 1 && gx >= 1 && gx < cols - 1) {
        const float v11 = B[-TMP_W - 1], v12 = B[-TMP_W], v13 = B[-TMP_W + 1];
        const float v21 = B[-1], v22 = B[0], v23 = B[1];
        const float v31 = B[TMP_W - 1], v32 = B[TMP_W], v33 = B[TMP_W + 1];
        float Lxx = (v21 - 2 * v22 + v23);
        float Lyy = (v12 - 2 * v22 + v32);
        float Lxy = (v13 - v11 + v31 - v33) / 4.0f;
        o = (Lxx * Lyy - Lxy * Lxy) * norm2;
      }
      jb.resp[gofs] = o;
    }
  }
}

// HessianResponse of an existing level (first level of octaves >= 1), pyramid.cpp:223-281
__global__ __launch_bounds__(256) void k_hessian(BlurBatch batch) {
  const int ji = find_job(batch.tile0, batch.nj, blockIdx.x);
  const BlurJob jb = batch.j[ji];
  const int rows = jb.rows, cols = jb.cols;
  const int lt = blockIdx.x - batch.tile0[ji], ntx = (cols + 63) / 64;
  const int gx = (lt % ntx) * 64 + (threadIdx.x & 63), gy = (lt / ntx) * 4 + (threadIdx.x >> 6);
  if (gx >= cols || gy >= rows) return;
  float o = 0.f;
  if (gy >= 1 && gy < rows - 1 && gx >= 1 && gx < cols - 1) {
    const float *B = jb.src + (size_t)gy * cols + gx;
    const float v11 = B[-cols - 1], v12 = B[-cols], v13 = B[-cols + 1];
    const float v21 = B[-1], v22 = B[0], v23 = B[1];
    const float v31 = B[cols - 1], v32 = B[cols], v33 = B[cols + 1];
    float Lxx = (v21 - 2 * v22 + v23);
    float Lyy = (v12 - 2 * v22 + v32);
    float Lxy = (v13 - v11 + v31 - v33) / 4.0f;
    o = (Lxx * Lyy - Lxy * Lxy) * (jb.norm * jb.norm);
  }
  jb.resp[(size_t)gy * cols + gx] = o;
}

// cv::resize(src, dst, Size(0,0), 0.5, 0.5, INTER_LINEAR) == area-fast 2x2 (pyramid.cpp:520):
// full blocks ((s00+s01)+s10)+s11)*0.25f, partial blocks sum(available)/count.
MX_D float resize_half_px(const float *src, int sr, int sc, int dx, int dy) {
  const int sy0 = dy * 2, sx0 = dx * 2;
  float out;
  if (sy0 >= sr) out = 0.f;
  else if (sy0 + 2 <= sr && dx < sc / 2) {
    const float *S = src + (size_t)sy0 * sc + sx0;
    float sum = 0.f;
    sum = sum + (((S[0] + S[1]) + S[sc]) + S[sc + 1]);
    out = sum * 0.25f;
  } else {
    float sum = 0.f; int count = 0;
    for (int sy = 0; sy < 2; sy++) {
      if (sy0 + sy >= sr) break;
      for (int sx = 0; sx < 2; sx++) {
        if (sx0 + sx >= sc) break;
        sum += src[(size_t)(sy0 + sy) * sc + sx0 + sx];
        count++;
      }
    }
    out = (sx0 >= sc) ? 0.f : sum / (float)count;
  }
  return out;
}
// First level of an octave >= 1: the 2 x 2 resize and, in the same launch, the Hessian response of the resized level
// (HessianResponse, pyramid.cpp:223-281; the expression of k_hessian) -- a thread forms the 3 x 3 resized values its
// response needs itself (the same f32 values its neighbours store), which saves a launch per octave.
__global__ __launch_bounds__(256) void k_resize_half(ResizeBatch batch) {
  const int ji = find_job(batch.tile0, batch.nj, blockIdx.x);
  const ResizeJob jb = batch.j[ji];
  const int lt = blockIdx.x - batch.tile0[ji], ntx = (jb.dcols + 63) / 64;
  const int dx = (lt % ntx) * 64 + (threadIdx.x & 63), dy = (lt / ntx) * 4 + (threadIdx.x >> 6);
  if (dx >= jb.dcols || dy >= jb.drows) return;
  const int sr = jb.srows, sc = jb.scols, rows = jb.drows, cols = jb.dcols;
  const float v22 = resize_half_px(jb.src, sr, sc, dx, dy);
  jb.dst[(size_t)dy * cols + dx] = v22;
  if (!jb.resp) return;
  float o = 0.f;
  if (dy >= 1 && dy < rows - 1 && dx >= 1 && dx < cols - 1) {
    const float v11 = resize_half_px(jb.src, sr, sc, dx - 1, dy - 1), v12 = resize_half_px(jb.src, sr, sc, dx, dy - 1),
                v13 = resize_half_px(jb.src, sr, sc, dx + 1, dy - 1);
    const float v21 = resize_half_px(jb.src, sr, sc, dx - 1, dy), v23 = resize_half_px(jb.src, sr, sc, dx + 1, dy);
    const float v31 = resize_half_px(jb.src, sr, sc, dx - 1, dy + 1), v32 = resize_half_px(jb.src, sr, sc, dx, dy + 1),
                v33 = resize_half_px(jb.src, sr, sc, dx + 1, dy + 1);
    float Lxx = (v21 - 2 * v22 + v23);
    float Lyy = (v12 - 2 * v22 + v32);
    float Lxy = (v13 - v11 + v31 - v33) / 4.0f;
    o = (Lxx * Lyy - Lxy * Lxy) * (jb.norm * jb.norm);
  }
  jb.resp[(size_t)dy * cols + dx] = o;
}

// ---------------------------------------------------------------------------------------
// 3x3x3 extremum scan + sub-pixel localisation.  One thread per pixel of the scan window
// [border, dim-border); a thread that finds an extremum runs the <= 5 Newton steps itself
// and appends a candidate record.  The first-come-first-served octaveMap rule
// (pyramid.cpp:414-418) depends on scan order, so it is applied by the host after sorting
// the candidates by (image, octave, level, row, col).
// ---------------------------------------------------------------------------------------
MX_D bool is_max9(float val, const float *p, int cols) {
  bool ok = true;
#pragma unroll
  for (int dr = -1; dr <= 1; dr++)
#pragma unroll
    for (int dc = -1; dc <= 1; dc++) ok = ok && !(p[dr * cols + dc] > val);
  return ok;
}
MX_D bool is_min9(float val, const float *p, int cols) {
  bool ok = true;
#pragma unroll
  for (int dr = -1; dr <= 1; dr++)
#pragma unroll
    for (int dc = -1; dc <= 1; dc++) ok = ok && !(p[dr * cols + dc] < val);
  return ok;
}

// One launch scans every (image, octave, level) of a batch: the 64 x NMS_ROWS pixel tiles of all jobs are numbered
// consecutively (tilePrefix, expanded to a tile -> job table once per launch).  A candidate's path is a chain of ~10
// dependent memory round trips (3 x 9-neighbour tests, <= 5 Newton steps), which is what a launch lasts however few
// pixels it covers -- hence one launch instead of one per octave.  Nearly every pixel fails the threshold test on its own
// value, so the scan is a stream over the response planes: a thread first loads its NMS_ROWS / 4 pixels (independent
// loads), then follows up the few that pass.
constexpr int NMS_ROWS = 16, NMS_LW = 66, NMS_REFINE_BLOCKS = 16;   // refine blocks per sub-queue

// sub-pixel localisation of one 3x3x3 extremum (pyramid.cpp:341-419): <= 5 Newton steps, edge / value tests, record
// Returns true and fills `k` for an accepted extremum; the caller appends the records of a wavefront with one atomic.
MX_D bool nms_refine(const NmsBatch &batch, const NmsJob &jb, int r, int c, Candidate &k) {
  const int rows = jb.rows, cols = jb.cols;
  const int r0 = r, c0 = c;
  float b[3] = {0.f, 0.f, 0.f};
  float val = 0.f;
  int nr = r, nc = c;
  for (int iter = 0; iter < 5; iter++) {
    r = nr; c = nc;
    const float *c1 = jb.cur + (size_t)r * cols, *c0p = c1 - cols, *c2 = c1 + cols;
    const float *l1 = jb.low + (size_t)r * cols, *l0 = l1 - cols, *l2 = l1 + cols;
    const float *h1 = jb.high + (size_t)r * cols, *h0 = h1 - cols, *h2 = h1 + cols;
    float dxx = c1[c - 1] - 2.0f * c1[c] + c1[c + 1];
    float dyy = c0p[c] - 2.0f * c1[c] + c2[c];
    float dss = l1[c] - 2.0f * c1[c] + h1[c];
    float dxy = 0.25f * (c2[c + 1] - c2[c - 1] - c0p[c + 1] + c0p[c - 1]);
    if (iter == 0) {
      float edgeScore = (dxx + dyy) * (dxx + dyy) / (dxx * dyy - dxy * dxy);
      if ((double)edgeScore >= batch.edgeScoreThreshold || edgeScore < 0) return false;
    }
    float dxs = 0.25f * (h1[c + 1] - h1[c - 1] - l1[c + 1] + l1[c - 1]);
    float dys = 0.25f * (h2[c] - h0[c] - l2[c] + l0[c]);
    float A[9] = {dxx, dxy, dxs, dxy, dyy, dys, dxs, dys, dss};
    float dx = 0.5f * (c1[c + 1] - c1[c - 1]);
    float dy = 0.5f * (c2[c] - c0p[c]);
    float ds = 0.5f * (h1[c] - l1[c]);
    b[0] = -dx; b[1] = -dy; b[2] = -ds;
    solve3(A, b);
    if (isnan(b[0]) || isnan(b[1]) || isnan(b[2])) return false;
    val = c1[c] + 0.5f * (dx * b[0] + dy * b[1] + ds * b[2]);
    if ((double)b[0] > 0.6) { if (c < cols - 3) nc++; else return false; }
    if ((double)b[1] > 0.6) { if (r < rows - 3) nr++; else return false; }
    if ((double)b[0] < -0.6) { if (c > 3) nc--; else return false; }
    if ((double)b[1] < -0.6) { if (r > 3) nr--; else return false; }
    if (nr == r && nc == c) break;
  }
  if (fabsf(b[0]) > 1.5f || fabsf(b[1]) > 1.5f || fabsf(b[2]) > 1.5f || fabsf(val) < batch.finalTh) return false;
  int type;
  if (batch.detType == MODSX_DET_DOG) type = val < 0 ? 11 : 10;            // DOG_BRIGHT / DOG_DARK
  else if (batch.detType == MODSX_DET_HARRIS) type = val < 0 ? 31 : 30;    // HARRIS_BRIGHT / HARRIS_DARK
  else if (val < 0) type = 2;
  else {
    const float *p = jb.blur + (size_t)r * cols + c;
    float Lxx = (p[-1] - 2 * p[0] + p[1]);
    type = (Lxx < 0) ? 0 : 1;
  }
  k.img = jb.img; k.octave = jb.octave; k.level = jb.level; k.type = type;
  k.r0 = r0; k.c0 = c0; k.r = r; k.c = c;
  k.b0 = b[0]; k.b1 = b[1]; k.b2 = b[2]; k.val = val;
  return true;
}

__global__ __launch_bounds__(256) void k_nms_localize(NmsBatch batch, const NmsJob *__restrict__ jobs,
                                                      const int *__restrict__ tilePrefix, const int *__restrict__ tileJob,
                                                      int4 *queue, unsigned *qcount, unsigned qcap, unsigned *overflow) {
  const int tile = blockIdx.x;
  const int jid = tileJob[tile];
  const NmsJob jb = jobs[jid];
  const int rows = jb.rows, cols = jb.cols, B = batch.border;
  const int local = tile - tilePrefix[jid];
  const int tilesX = (cols - 2 * B + 63) / 64;
  const int by = local / tilesX, bx = local - by * tilesX;
  const int r0 = B + by * NMS_ROWS, c0 = B + bx * 64;   // first pixel of the tile
  // the tile and a one-pixel halo of the three response planes: LDS row i = image row r0 - 1 + i, column likewise
  __shared__ float sp[3][NMS_ROWS + 2][NMS_LW];
  {
    const float *planes[3] = {jb.cur, jb.low, jb.high};
    constexpr int NE = (NMS_ROWS + 2) * NMS_LW, PER = (NE + 255) / 256;
    float t[3][PER];
#pragma unroll
    for (int pl = 0; pl < 3; pl++)
#pragma unroll
      for (int u = 0; u < PER; u++) {
        const int idx = threadIdx.x + 256 * u, rr = idx / NMS_LW, cc = idx - rr * NMS_LW;
        int gr = r0 - 1 + rr, gc = c0 - 1 + cc;
        gr = gr < 0 ? 0 : (gr > rows - 1 ? rows - 1 : gr);     // clamped positions are never part of a tested neighbourhood
        gc = gc < 0 ? 0 : (gc > cols - 1 ? cols - 1 : gc);
        t[pl][u] = idx < NE ? planes[pl][(size_t)gr * cols + gc] : 0.f;
      }
#pragma unroll
    for (int pl = 0; pl < 3; pl++)
#pragma unroll
      for (int u = 0; u < PER; u++) {
        const int idx = threadIdx.x + 256 * u;
        if (idx < NE) (&sp[pl][0][0])[idx] = t[pl][u];
      }
  }
  __syncthreads();
  // thread (lc, g) tests rows 4g .. 4g+3 of column lc.  `!(x > v0)` for the 27 values of the 3x3x3 block is max <= v0
  // (v_max / v_min skip NaNs exactly like the failed comparisons do), and the 3x3 maxima of four consecutive rows share
  // their row maxima, so a pixel costs ~35 VALU operations and no divergent loads.
  const int lc = threadIdx.x & 63, g = threadIdx.x >> 6;
  float mx[4], mn[4];
#pragma unroll
  for (int j = 0; j < 4; j++) { mx[j] = -INFINITY; mn[j] = INFINITY; }
  // a wavefront none of whose 256 pixels passes the threshold on its own value (flat image areas) skips the stencil
  bool any = false;
#pragma unroll
  for (int j = 0; j < 4; j++) {
    const float v0 = sp[0][4 * g + j + 1][lc + 1];
    any = any || v0 > batch.posTh || v0 < batch.negTh;
  }
  if (__ballot(any))
#pragma unroll
  for (int pl = 0; pl < 3; pl++) {
    float hmax[6], hmin[6];
#pragma unroll
    for (int k = 0; k < 6; k++) {
      const float a = sp[pl][4 * g + k][lc], b = sp[pl][4 * g + k][lc + 1], d = sp[pl][4 * g + k][lc + 2];
      hmax[k] = fmaxf(fmaxf(a, b), d);
      hmin[k] = fminf(fminf(a, b), d);
    }
#pragma unroll
    for (int j = 0; j < 4; j++) {
      mx[j] = fmaxf(mx[j], fmaxf(fmaxf(hmax[j], hmax[j + 1]), hmax[j + 2]));
      mn[j] = fminf(mn[j], fminf(fminf(hmin[j], hmin[j + 1]), hmin[j + 2]));
    }
  }
  const int c = c0 + lc;
  // the tile's extrema are queued with ONE global atomic per workgroup (slots within the tile come from an LDS counter),
  // and the queue is split into NMS_QUEUES sub-queues with counters 128 bytes apart: atomics on one address are served
  // one after the other and would cost more than the scan itself
  __shared__ unsigned scount, sbase;
  if (threadIdx.x == 0) scount = 0;
  __syncthreads();
  unsigned slot[4];
#pragma unroll
  for (int j = 0; j < 4; j++) {
    const int r = r0 + 4 * g + j;
    const float v0 = sp[0][4 * g + j + 1][lc + 1];
    bool cand = false;
    if (v0 > batch.posTh) cand = mx[j] <= v0;
    else if (v0 < batch.negTh) cand = mn[j] >= v0;
    slot[j] = (cand && r < rows - B && c < cols - B) ? atomicAdd(&scount, 1u) : 0xffffffffu;
  }
  __syncthreads();
  const unsigned sq = blockIdx.x % NMS_QUEUES, qsub = qcap / NMS_QUEUES;
  if (threadIdx.x == 0) sbase = scount ? atomicAdd(qcount + 32 * sq, scount) : 0u;
  __syncthreads();
#pragma unroll
  for (int j = 0; j < 4; j++)
    if (slot[j] != 0xffffffffu) {
      if (sbase + slot[j] < qsub) queue[(size_t)sq * qsub + sbase + slot[j]] = make_int4(jid, r0 + 4 * g + j, c, 0);
      else atomicOr(overflow, 1u);   // the host turns this into an error: dropping extrema silently would change the result
    }
}

// The same scan with ONE pass per octave (findLevelKeypoints over levels 1 .. S of an octave, pyramid.cpp:432-452, 524-530):
// a tile parks the S + 2 response planes of the octave once and every level takes its 3 x 3 x 3 extremum test from the
// per-plane 3 x 3 maxima / minima, which are formed once per plane -- each response plane is fetched once instead of by three
// level passes (535 -> ~300 MB per 31-view launch), and the stencil runs over S + 2 planes instead of 3 S.  `tileJob[tile]` is
// the octave's entry in `first` / `tilePrefix`; its level jobs are jobs[first[o]] .. jobs[first[o] + S - 1] (consecutive
// levels of one octave, as the host builds them), and the queue records name the level job, as the per-level scan's do.
template <int S>
__global__ __launch_bounds__(256) void k_nms_localize_oct(NmsBatch batch, const NmsJob *__restrict__ jobs, const int *__restrict__ first,
                                                          const int *__restrict__ tilePrefix, const int *__restrict__ tileJob,
                                                          int4 *queue, unsigned *qcount, unsigned qcap, unsigned *overflow) {
  constexpr int NP = S + 2;
  const int tile = blockIdx.x;
  const int oj = tileJob[tile];
  const int jid0 = first[oj];
  const NmsJob jb = jobs[jid0];
  const int rows = jb.rows, cols = jb.cols, B = batch.border;
  const int local = tile - tilePrefix[oj];
  const int tilesX = (cols - 2 * B + 63) / 64;
  const int by = local / tilesX, bx = local - by * tilesX;
  const int r0 = B + by * NMS_ROWS, c0 = B + bx * 64;
  __shared__ float sp[NP][NMS_ROWS + 2][NMS_LW];
  {
    const float *planes[NP];
    planes[0] = jb.low; planes[1] = jb.cur;
#pragma unroll
    for (int l = 0; l < S; l++) planes[2 + l] = jobs[jid0 + l].high;
    constexpr int NE = (NMS_ROWS + 2) * NMS_LW, PER = (NE + 255) / 256;
    // the (clamped) offsets of a thread's elements are the same in every plane
    int ofs[PER];
#pragma unroll
    for (int u = 0; u < PER; u++) {
      const int idx = threadIdx.x + 256 * u, rr = idx / NMS_LW, cc = idx - rr * NMS_LW;
      int gr = r0 - 1 + rr, gc = c0 - 1 + cc;
      gr = gr < 0 ? 0 : (gr > rows - 1 ? rows - 1 : gr);     // clamped positions are never part of a tested neighbourhood
      gc = gc < 0 ? 0 : (gc > cols - 1 ? cols - 1 : gc);
      ofs[u] = gr * cols + gc;
    }
    float t[NP][PER];
#pragma unroll
    for (int pl = 0; pl < NP; pl++)
#pragma unroll
      for (int u = 0; u < PER; u++) t[pl][u] = threadIdx.x + 256 * u < NE ? planes[pl][ofs[u]] : 0.f;
#pragma unroll
    for (int pl = 0; pl < NP; pl++)
#pragma unroll
      for (int u = 0; u < PER; u++) {
        const int idx = threadIdx.x + 256 * u;
        if (idx < NE) (&sp[pl][0][0])[idx] = t[pl][u];
      }
  }
  __syncthreads();
  const int lc = threadIdx.x & 63, g = threadIdx.x >> 6;
  bool any = false;
#pragma unroll
  for (int l = 1; l <= S; l++)
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const float v0 = sp[l][4 * g + j + 1][lc + 1];
      any = any || v0 > batch.posTh || v0 < batch.negTh;
    }
  __shared__ unsigned scount, sbase;
  if (threadIdx.x == 0) scount = 0;
  unsigned slot[S][4];
#pragma unroll
  for (int l = 0; l < S; l++)
#pragma unroll
    for (int j = 0; j < 4; j++) slot[l][j] = 0xffffffffu;
  __syncthreads();
  if (__ballot(any)) {
    // 3 x 3 maximum / minimum of every plane at the thread's four pixels (rows 4g .. 4g + 3 of column lc)
    float pmx[NP][4], pmn[NP][4];
#pragma unroll
    for (int pl = 0; pl < NP; pl++) {
      float hmax[6], hmin[6];
#pragma unroll
      for (int k = 0; k < 6; k++) {
        const float a = sp[pl][4 * g + k][lc], b = sp[pl][4 * g + k][lc + 1], d = sp[pl][4 * g + k][lc + 2];
        hmax[k] = fmaxf(fmaxf(a, b), d);
        hmin[k] = fminf(fminf(a, b), d);
      }
#pragma unroll
      for (int j = 0; j < 4; j++) {
        pmx[pl][j] = fmaxf(fmaxf(hmax[j], hmax[j + 1]), hmax[j + 2]);
        pmn[pl][j] = fminf(fminf(hmin[j], hmin[j + 1]), hmin[j + 2]);
      }
    }
    const int c = c0 + lc;
#pragma unroll
    for (int l = 1; l <= S; l++)
#pragma unroll
      for (int j = 0; j < 4; j++) {
        const int r = r0 + 4 * g + j;
        const float v0 = sp[l][4 * g + j + 1][lc + 1];
        // `!(x > v0)` for the 27 values of the 3 x 3 x 3 block is max <= v0 (v_max / v_min skip NaNs exactly like the failed
        // comparisons do); the plane order of the maximum is irrelevant
        const float mx = fmaxf(fmaxf(pmx[l - 1][j], pmx[l][j]), pmx[l + 1][j]);
        const float mn = fminf(fminf(pmn[l - 1][j], pmn[l][j]), pmn[l + 1][j]);
        bool cand = false;
        if (v0 > batch.posTh) cand = mx <= v0;
        else if (v0 < batch.negTh) cand = mn >= v0;
        if (cand && r < rows - B && c < cols - B) slot[l - 1][j] = atomicAdd(&scount, 1u);
      }
  }
  __syncthreads();
  const unsigned sq = blockIdx.x % NMS_QUEUES, qsub = qcap / NMS_QUEUES;
  if (threadIdx.x == 0) sbase = scount ? atomicAdd(qcount + 32 * sq, scount) : 0u;
  __syncthreads();
#pragma unroll
  for (int l = 0; l < S; l++)
#pragma unroll
    for (int j = 0; j < 4; j++)
      if (slot[l][j] != 0xffffffffu) {
        if (sbase + slot[l][j] < qsub) queue[(size_t)sq * qsub + sbase + slot[l][j]] = make_int4(jid0 + l, r0 + 4 * g + j, c0 + lc, 0);
        else atomicOr(overflow, 1u);   // the host turns this into an error: dropping extrema silently would change the result
      }
}

// (B+G+R)/3 of GenerateSynthImageCorr (synth-detection.cpp:253-262): a cv::MatExpr that OpenCV
// evaluates as addWeighted(B+G, 1/3., R, 1/3., 0) in f64 for CV_32F.
__global__ void k_gray_u8(const uint8_t *src, float *dst, size_t n, int channels) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  if (channels == 1) { dst[i] = (float)src[i]; return; }
  const double k = 1. / 3.0;
  float t = (float)src[3 * i] + (float)src[3 * i + 1];
  dst[i] = (float)((double)t * k + (double)(float)src[3 * i + 2] * k + 0.0);
}
__global__ void k_gray_f32(const float *src, float *dst, size_t n, int channels) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  if (channels == 1) { dst[i] = src[i]; return; }
  const double k = 1. / 3.0;
  float t = src[3 * i] + src[3 * i + 1];
  dst[i] = (float)((double)t * k + (double)src[3 * i + 2] * k + 0.0);
}

static int fill_tiles(BlurBatch &b, int nj, int tw, int th) {
  b.nj = nj;
  int t = 0;
  for (int i = 0; i < nj; i++) { b.tile0[i] = t; t += ((b.j[i].cols + tw - 1) / tw) * ((b.j[i].rows + th - 1) / th); }
  b.tile0[nj] = t;
  return t;
}
template <int TH>
static void launch_blur_hess_th(hipStream_t s, const BlurBatch &b, int tiles) {
  dim3 grid(tiles);
  MX_DUP(K_BLUR_HESS) switch (b.n >> 1) {
    case 1: hipLaunchKernelGGL((k_blur_hess<1, TH>), grid, dim3(256), 0, s, b); break;
    case 2: hipLaunchKernelGGL((k_blur_hess<2, TH>), grid, dim3(256), 0, s, b); break;
    case 3: hipLaunchKernelGGL((k_blur_hess<3, TH>), grid, dim3(256), 0, s, b); break;
    case 4: hipLaunchKernelGGL((k_blur_hess<4, TH>), grid, dim3(256), 0, s, b); break;
    case 5: hipLaunchKernelGGL((k_blur_hess<5, TH>), grid, dim3(256), 0, s, b); break;
    case 6: hipLaunchKernelGGL((k_blur_hess<6, TH>), grid, dim3(256), 0, s, b); break;
    case 7: hipLaunchKernelGGL((k_blur_hess<7, TH>), grid, dim3(256), 0, s, b); break;
    default: hipLaunchKernelGGL((k_blur_hess<8, TH>), grid, dim3(256), 0, s, b); break;
  }
}
void launch_blur_hess(hipStream_t s, const BlurBatch &bin, int nj, int, int) {
  BlurBatch b = bin;
  int tiles = fill_tiles(b, nj, TW, 32);
  if (tiles <= 0) return;
  // under two rounds of workgroups (5-6 per CU x 256 CUs) the 16-row tile is faster
  if (tiles >= 2560) { launch_blur_hess_th<32>(s, b, tiles); return; }
  tiles = fill_tiles(b, nj, TW, 16);
  launch_blur_hess_th<16>(s, b, tiles);
}
void launch_hessian(hipStream_t s, const BlurBatch &bin, int nj, int, int) {
  BlurBatch b = bin;
  const int tiles = fill_tiles(b, nj, 64, 4);
  if (tiles > 0) hipLaunchKernelGGL(k_hessian, dim3(tiles), dim3(256), 0, s, b);
}
void launch_resize_half(hipStream_t s, const ResizeBatch &bin, int nj, int, int) {
  ResizeBatch b = bin;
  b.nj = nj;
  int t = 0;
  for (int i = 0; i < nj; i++) { b.tile0[i] = t; t += ((b.j[i].dcols + 63) / 64) * ((b.j[i].drows + 3) / 4); }
  b.tile0[nj] = t;
  if (t > 0) MX_DUP(K_RESIZE) hipLaunchKernelGGL(k_resize_half, dim3(t), dim3(256), 0, s, b);
}

// The extrema found by the scan are few and scattered (about one per wavefront), and each needs a chain of ~10 dependent
// round trips: run inside the scan they would hold nearly every wavefront for the whole chain with one lane working.
// The scan therefore only queues (job, row, column); this kernel refines one queued extremum per lane.
__global__ __launch_bounds__(256) void k_nms_refine(NmsBatch batch, const NmsJob *__restrict__ jobs, const int4 *__restrict__ queue,
                                                    const unsigned *__restrict__ qcount, unsigned qcap, Candidate *out,
                                                    unsigned *counter, unsigned cap) {
  const unsigned sq = blockIdx.y, qsub = qcap / NMS_QUEUES;
  unsigned n = qcount[32 * sq];
  n = n < qsub ? n : qsub;
  const int lane = threadIdx.x & 63;
  for (unsigned i0 = blockIdx.x * 256; i0 < n; i0 += gridDim.x * 256) {   // wave-uniform trip count
    const unsigned i = i0 + threadIdx.x;
    Candidate k;
    bool ok = false;
    if (i < n) {
      const int4 q = queue[(size_t)sq * qsub + i];
      ok = nms_refine(batch, jobs[q.x], q.y, q.z, k);
    }
    // one atomic per wavefront: the accepted records take consecutive slots
    const unsigned long long m = __ballot(ok);
    if (m) {
      unsigned base = 0;
      const int leader = __ffsll((long long)m) - 1;
      if (lane == leader) base = atomicAdd(counter, (unsigned)__popcll(m));
      base = __shfl(base, leader);
      const unsigned slot = base + (unsigned)__popcll(m & ((1ull << lane) - 1ull));
      if (ok && slot < cap) out[slot] = k;
    }
  }
}

void launch_nms(hipStream_t s, const NmsBatch &b, const NmsJob *jobs, const int *tilePrefix, const int *tileJob, int nj,
                int nTiles, int4 *queue, unsigned *qcount, unsigned qcap, Candidate *out, unsigned *counter, unsigned cap,
                const int *octFirst, int levelsPerOctave) {
  if (nj <= 0 || nTiles <= 0) return;
  // octFirst: the tiles number OCTAVES (tilePrefix / tileJob index `octFirst`), every octave holding levelsPerOctave consecutive
  // level jobs; the one-pass-per-octave scan exists for the shipped numberOfScales = 3
  if (octFirst && levelsPerOctave == 3)
    hipLaunchKernelGGL(k_nms_localize_oct<3>, dim3(nTiles), dim3(256), 0, s, b, jobs, octFirst, tilePrefix, tileJob, queue, qcount, qcap, counter + 1);
  else
  hipLaunchKernelGGL(k_nms_localize, dim3(nTiles), dim3(256), 0, s, b, jobs, tilePrefix, tileJob, queue, qcount, qcap, counter + 1);
  hipLaunchKernelGGL(k_nms_refine, dim3(NMS_REFINE_BLOCKS, NMS_QUEUES), dim3(256), 0, s, b, jobs, queue, qcount, qcap, out, counter, cap);
}
// *ptr = (unsigned char)*in_ptr of DetectMSERs (extrema.cpp:401-403): f32 -> u8 by truncation, 4 pixels per thread
__global__ void k_trunc_u8(const float *src, uint8_t *dst, size_t n) {
  const size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
  if (i + 3 < n) {
    const float4 v = *reinterpret_cast<const float4 *>(src + i);
    const unsigned p = (unsigned)(uint8_t)v.x | ((unsigned)(uint8_t)v.y << 8) | ((unsigned)(uint8_t)v.z << 16) | ((unsigned)(uint8_t)v.w << 24);
    *reinterpret_cast<unsigned *>(dst + i) = p;
  } else {
    for (size_t k = i; k < n; k++) dst[k] = (uint8_t)src[k];
  }
}
void launch_trunc_u8(hipStream_t s, const float *src, uint8_t *dst, size_t n) {
  if (!n) return;
  const size_t threads = (n + 3) / 4;
  hipLaunchKernelGGL(k_trunc_u8, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, s, src, dst, n);
}
// ---- the bin sort of the MSER views (sortPixels.cpp:76-125) on the device -------------------------------------------------------------
// The component tree takes a view's pixel offsets (padded coordinates, stride cols + 2) per grey level in raster order.  Counting
// sort in three launches over all views of a set: a block = MSB_ROWS consecutive rows of one view, one wavefront.
//   k_mser_hist     per block the 256 counts of its rows
//   k_mser_scan     per view: start[l] = pixels below level l; base[block][l] = start[l] + the counts of the view's earlier blocks
//   k_mser_scatter  a wavefront walks its rows 64 pixels at a time: a lane's rank among the lanes of its step that hold the same grey
//                   value (eight ballots: the lanes that agree in every bit) + the level's running count = its place; raster order
//                   inside a level is the order of (block, step, lane), so the result is the host's stable sort bit for bit
constexpr int MSB_ROWS = 8;
struct MserSortBatch {
  const uint8_t *u8[MAXB];
  int rows[MAXB], cols[MAXB], blk0[MAXB + 1];     // blk0: first block of every view
  unsigned long long ordOfs[MAXB];                // first place of the view in `order`
  int n;
};
MX_D int mser_view_of(const MserSortBatch &b, int blk) {
  int v = 0;
  while (v + 1 < b.n && blk >= b.blk0[v + 1]) v++;
  return v;
}
__global__ __launch_bounds__(64) void k_mser_hist(MserSortBatch b, int *blockHist) {
  __shared__ int h[256];
  const int blk = blockIdx.x, v = mser_view_of(b, blk), lane = threadIdx.x;
  for (int i = lane; i < 256; i += 64) h[i] = 0;
  __syncthreads();
  const int r0 = (blk - b.blk0[v]) * MSB_ROWS, r1 = min(r0 + MSB_ROWS, b.rows[v]), cols = b.cols[v];
  const uint8_t *src = b.u8[v];
  for (int r = r0; r < r1; r++)
    for (int c = lane; c < cols; c += 64) atomicAdd(&h[src[(size_t)r * cols + c]], 1);
  __syncthreads();
  for (int i = lane; i < 256; i += 64) blockHist[(size_t)blk * 256 + i] = h[i];
}
__global__ __launch_bounds__(256) void k_mser_scan(MserSortBatch b, int *blockHist, int *start) {
  __shared__ int tot[256];
  const int v = blockIdx.x, l = threadIdx.x;
  const int b0 = b.blk0[v], b1 = b.blk0[v + 1];
  int run = 0;
  for (int k = b0; k < b1; k++) { const int c = blockHist[(size_t)k * 256 + l]; blockHist[(size_t)k * 256 + l] = run; run += c; }
  tot[l] = run;
  __syncthreads();
  if (l == 0) { int at = 0; for (int i = 0; i < 256; i++) { const int c = tot[i]; tot[i] = at; at += c; } start[v * 257 + 256] = at; }
  __syncthreads();
  const int st = tot[l];
  start[v * 257 + l] = st;
  for (int k = b0; k < b1; k++) blockHist[(size_t)k * 256 + l] += st;
}
__global__ __launch_bounds__(64) void k_mser_scatter(MserSortBatch b, const int *blockBase, int *order) {
  __shared__ int cnt[256];
  const int blk = blockIdx.x, v = mser_view_of(b, blk), lane = threadIdx.x;
  for (int i = lane; i < 256; i += 64) cnt[i] = blockBase[(size_t)blk * 256 + i];
  __syncthreads();
  const int r0 = (blk - b.blk0[v]) * MSB_ROWS, r1 = min(r0 + MSB_ROWS, b.rows[v]), cols = b.cols[v], stride = cols + 2;
  const uint8_t *src = b.u8[v];
  int *ord = order + b.ordOfs[v];
  const unsigned long long below = (1ull << lane) - 1ull;
  for (int r = r0; r < r1; r++)
    for (int c0 = 0; c0 < cols; c0 += 64) {
      const int c = c0 + lane;
      const bool on = c < cols;
      const int g = on ? src[(size_t)r * cols + c] : 0;
      unsigned long long same = __ballot(on);
#pragma unroll
      for (int bit = 0; bit < 8; bit++) {
        const unsigned long long m = __ballot((g >> bit) & 1);
        same &= ((g >> bit) & 1) ? m : ~m;
      }
      const int rank = __popcll(same & below);
      int at = 0;
      if (on) at = cnt[g];
      if (on) ord[at + rank] = (r + 1) * stride + c + 1;
      if (on && rank == 0) cnt[g] = at + __popcll(same);      // (one lane per value; every lane of the step has read the count above)
    }
}
size_t mser_sort_blocks(const int *rows, int n) { size_t t = 0; for (int i = 0; i < n; i++) t += (size_t)(rows[i] + MSB_ROWS - 1) / MSB_ROWS; return t; }
// u8[i]: rows[i] x cols[i] bytes on the device; order: sum of rows x cols ints (view i from ordOfs[i]); start: n x 257 ints;
// blockHist: mser_sort_blocks(rows, n) x 256 ints of scratch
void launch_mser_sort(hipStream_t s, const uint8_t *const *u8, const int *rows, const int *cols, const size_t *ordOfs, int n, int *blockHist,
                      int *start, int *order) {
  if (n <= 0) return;
  MserSortBatch b;
  memset(&b, 0, sizeof b);
  b.n = n;
  int t = 0;
  for (int i = 0; i < n; i++) {
    b.u8[i] = u8[i]; b.rows[i] = rows[i]; b.cols[i] = cols[i]; b.ordOfs[i] = ordOfs[i];
    b.blk0[i] = t;
    t += (rows[i] + MSB_ROWS - 1) / MSB_ROWS;
  }
  b.blk0[n] = t;
  if (t <= 0) return;
  hipLaunchKernelGGL(k_mser_hist, dim3(t), dim3(64), 0, s, b, blockHist);
  hipLaunchKernelGGL(k_mser_scan, dim3(n), dim3(256), 0, s, b, blockHist, start);
  hipLaunchKernelGGL(k_mser_scatter, dim3(t), dim3(64), 0, s, b, (const int *)blockHist, order);
}
void launch_gray(hipStream_t s, const void *src, float *dst, size_t n, int channels, int dtype) {
  dim3 grid((unsigned)((n + 255) / 256));
  if (dtype == 0) hipLaunchKernelGGL(k_gray_u8, grid, dim3(256), 0, s, (const uint8_t *)src, dst, n, channels);
  else hipLaunchKernelGGL(k_gray_f32, grid, dim3(256), 0, s, (const float *)src, dst, n, channels);
}

}  // namespace mx
