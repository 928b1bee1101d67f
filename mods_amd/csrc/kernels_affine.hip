// kernels_affine.hip -- Baumberg affine shape adaptation, one wavefront per keypoint.
//
// Reference: AffineShape::findAffineShape (AFF_BMBRG_SMM), detectors/affinedetectors/affine.cpp:26-169
//   interpolate        detectors/helpers.cpp:551-626  (f32 coordinates accumulated incrementally)
//   computeGradient    detectors/helpers.cpp:779-797
//   invSqrt / getEigenvalues   detectors/helpers.cpp:463-515
// The 19x19 window lives in LDS.  The 64 lanes produce the warped samples and the
// gradient products in parallel; the three second-moment sums are accumulated by three
// lanes in raster order because the reference's f32 running sums are order dependent.
#include "engine.hpp"

namespace mx {

constexpr int AW_MAX = 19;

// invSqrt (kmath.hpp inv_sqrt, detectors/helpers.cpp:463-502) for a whole wavefront that holds the same (a, b, c) in every
// lane: the two independent 1/sqrt of the rotated diagonal, and the two divisions by their geometric mean, run in lanes
// 0 and 1 side by side instead of one after the other in every lane (f64 sqrt and division are ~25 half-rate
// instructions each).  Every operation has the operands and the order of the scalar form.
MX_D void inv_sqrt_wave(int lane, float &a, float &b, float &c, float &l1, float &l2) {
  double t, r;
  if (b != 0) {
    r = double(c - a) / (2 * b);
    if (r >= 0) t = 1.0 / (r + sqrt(1 + r * r));
    else t = -1.0 / (-r + sqrt(1 + r * r));
    r = 1.0 / sqrt(1 + t * t);
    t = t * r;
  } else { r = 1; t = 0; }
  const bool odd = lane & 1;
  // even lanes: r*r*a - 2*r*t*b + t*t*c       odd lanes: t*t*a + 2*r*t*b + r*r*c
  const double rr = r * r, tt = t * t, m = 2 * r * t * b;
  const double first = (odd ? tt : rr) * a, last = (odd ? rr : tt) * c;
  double w = 1.0 / sqrt((odd ? first + m : first - m) + last);
  double x = __shfl(w, 0), z = __shfl(w, 1);
  const double d = sqrt(x * z);
  w = (odd ? z : x) / d;
  x = __shfl(w, 0); z = __shfl(w, 1);
  if (x < z) { l1 = float(z); l2 = float(x); } else { l1 = float(x); l2 = float(z); }
  a = float(r * r * x + t * t * z);
  b = float(-r * t * x + t * r * z);
  c = float(t * t * x + r * r * z);
}

// WT = window size when known at compile time (19, the default smmWindowSize), 0 = use the argument: the row / column of a
// pixel is an integer division by W, done once per keypoint (the LDS offsets of a pixel's four gradient neighbours do not
// change between iterations)
template <int WT>
__global__ __launch_bounds__(64) void k_baumberg(const AffJob *jobs, AffOut *out, int n, const float *mask, int Warg,
                                                 int maxIter, float convTh, float affInitialSigma) {
  const int W = WT ? WT : Warg;
  const int k = blockIdx.x;
  if (k >= n) return;
  const int lane = threadIdx.x;
  // 4.4 KB of LDS per keypoint: the kernel is a chain of dependent phases per iteration (coordinates -> taps -> gradients ->
  // 361 ordered adds -> Jacobi) and it is bound by vector-instruction issue (VALU busy ~90 %), so what counts is the number
  // of instructions per iteration (~1750; it was ~2150 with per-pixel edge branches in the gradient, 64-bit tap addresses
  // and the Jacobi step's two 1/sqrt in sequence).  The window mask and the neighbour offsets stay in registers (126 VGPRs,
  // 4 waves per SIMD: capping the registers for more waves only adds spills and is slower), and the sampled window shares
  // its buffer with the third product array: the products are staged in registers and written after every lane has taken
  // its gradients.
  // Tried and dropped: G keypoints per workgroup with the serial phases (ordered sums, Jacobi) of all G run side by side in
  // the lanes of one wavefront -- half the instructions per keypoint, but a keypoint still owns a wavefront and 4.4 KB of
  // LDS, so no more keypoints are in flight per CU, the group iterates in lock-step to its slowest member, and the launch
  // was 5-25 % slower for G = 2..16.  Two keypoint slots per WAVEFRONT (k_baumberg_stream below) is what the default
  // window size runs: no faster as a launch, but 17 % fewer vector instructions.
  __shared__ __attribute__((aligned(16))) float pa[AW_MAX * AW_MAX + 3], pb[AW_MAX * AW_MAX + 3], pc[AW_MAX * AW_MAX + 3];
  float *const simg = pc;
  const AffJob jb = jobs[k];
  const int WW = W * W, half = W >> 1;
  constexpr int PERM = (AW_MAX * AW_MAX + 63) / 64;
  float vmask[PERM];
  int pp[PERM], oxp[PERM], oxm[PERM], oyp[PERM], oym[PERM];   // pixel (clamped to the window) and its gradient neighbours
#pragma unroll
  for (int u = 0; u < PERM; u++) {
    const int i = lane + 64 * u;
    vmask[u] = i < WW ? mask[i] : 0.f;
    const int p = i < WW ? i : WW - 1;
    const int r = p / W, c = p - r * W;
    pp[u] = p;
    // computeGradient (helpers.cpp:779-797): one-sided differences on the window's frame, central ones inside
    oxp[u] = c == W - 1 ? p : p + 1;
    oxm[u] = c == 0 ? p : p - 1;
    oyp[u] = r == W - 1 ? p : p + W;
    oym[u] = r == 0 ? p : p - W;
  }
  float u11 = 1.0f, u12 = 0.0f, u21 = 0.0f, u22 = 1.0f, l1 = 1.0f, l2 = 1.0f;
  float era = 0.0f, erb = 0.0f;
  const float lx = jb.x / jb.pixelDistance, ly = jb.y / jb.pixelDistance;
  const float ratio = jb.s / (affInitialSigma * jb.pixelDistance);
  const gcfloat_p img = as_global(jb.blur);
  int ok = 0, l = 0;
  __syncthreads();
  for (l = 0; l < maxIter; l++) {
    const float a11 = u11 * ratio, a12 = u12 * ratio, a21 = u21 * ratio, a22 = u22 * ratio;
    const bool touch = check_borders(jb.cols, jb.rows, lx, ly, a11, a12, a21, a22, W, W);
    // sample coordinates: lane j runs the f32 running sums of row j (helpers.cpp:563-585) into LDS,
    // then all lanes take the bilinear taps
    {
      // row starts: lane j takes j steps of the running sum (a12 / a22 are uniform, the loop is not divergent: a lane
      // that has its row start sits out the remaining steps), then walks its row
      float rx = lx - (float)half * a12, ry = ly - (float)half * a22;
      if (WT) {
#pragma unroll
        for (int j = 1; j < (WT ? WT : 1); j++)
          if (lane >= j) { rx += a12; ry += a22; }
      } else {
        for (int j = 0; j < lane && j < W; j++) { rx += a12; ry += a22; }
      }
      if (lane < W) {
        float WX = rx - (float)half * a11;
        float WY = ry - (float)half * a21;
        if (WT) {
#pragma unroll
          for (int i = 0; i < (WT ? WT : 1); i++) {
            pa[lane * W + i] = WX;
            pb[lane * W + i] = WY;
            WX += a11;
            WY += a21;
          }
        } else {
#pragma unroll 1
          for (int i = 0; i < W; i++) {
            pa[lane * W + i] = WX;
            pb[lane * W + i] = WY;
            WX += a11;
            WY += a21;
          }
        }
      }
    }
    __syncthreads();
    {
      // all six taps of a lane in flight together (no per-tap branch: slots past the window repeat its last pixel and
      // are not stored): one memory round trip per iteration instead of six
      float sv[PERM];
      if (!touch) {
#pragma unroll
        for (int u = 0; u < PERM; u++) sv[u] = bilinear_tap(img, jb.rows, jb.cols, pa[pp[u]], pb[pp[u]], false);
      } else {
#pragma unroll
        for (int u = 0; u < PERM; u++) sv[u] = bilinear_tap_touch_select(img, jb.rows, jb.cols, pa[pp[u]], pb[pp[u]]);
      }
#pragma unroll
      for (int u = 0; u < PERM; u++)
        if (lane + 64 * u < WW) simg[lane + 64 * u] = sv[u];
    }
    __syncthreads();
    {
      float qa[PERM], qb[PERM], qc[PERM];
#pragma unroll
      for (int u = 0; u < PERM; u++) {
        const float gx = simg[oxp[u]] - simg[oxm[u]];
        const float gy = simg[oyp[u]] - simg[oym[u]];
        const float v = vmask[u];
        const float gxy = gx * gy;
        qa[u] = gx * gx * v;
        qb[u] = gxy * v;
        qc[u] = gy * gy * v;
      }
      __syncthreads();   // every gradient is taken: pc may replace the window
#pragma unroll
      for (int u = 0; u < PERM; u++) {
        const int p = lane + 64 * u;
        if (p < WW) { pa[p] = qa[u]; pb[p] = qb[u]; pc[p] = qc[u]; }
      }
    }
    __syncthreads();
    float acc = 0.f;
    if (lane < 3) {
      const float *arr = lane == 0 ? pa : (lane == 1 ? pb : pc);
      const float4 *a4 = reinterpret_cast<const float4 *>(arr);
      const int full = WW >> 2;
#pragma unroll 6
      for (int i = 0; i < full; i++) { const float4 v = a4[i]; acc += v.x; acc += v.y; acc += v.z; acc += v.w; }
      for (int i = full * 4; i < WW; i++) acc += arr[i];
      acc /= (float)WW;
    }
    float a = __shfl(acc, 0), b = __shfl(acc, 1), c = __shfl(acc, 2);
    inv_sqrt_wave(lane, a, b, c, l1, l2);
    if ((a != a) || (b != b) || (c != c)) break;
    erb = era;
    era = (float)(1.0 - (double)(l2 / l1));
    const float u11t = u11, u12t = u12;
    u11 = a * u11t + b * u21;
    u12 = a * u12t + b * u22;
    u21 = b * u11t + c * u21;
    u22 = b * u12t + c * u22;
    if (!eigenvalues(u11, u12, u21, u22, l1, l2)) break;
    if ((l1 / l2 > 6) || (l2 / l1 > 6)) break;
    if (era < convTh && erb < convTh) { ok = 1; break; }
    __syncthreads();
  }
  if (lane == 0) {
    AffOut o;
    o.u11 = u11; o.u12 = u12; o.u21 = u21; o.u22 = u22; o.ok = ok; o.iters = l;
    out[k] = o;
  }
}

// ------------------------------------------------------------------------------------------------
// K keypoints in flight per wavefront (19 x 19 window), streamed from the wavefront's own chunk of the job list.
// An iteration of k_baumberg is ~1200 vector instructions of which ~640 are work for a handful of lanes -- the 3 x 361
// ordered adds of the second-moment sums (3 lanes) and the f64 Jacobi step with the convergence tests (1 lane's worth) --
// and an instruction costs the same issue slot with 3 active lanes as with 64.  Here a wavefront has K keypoint SLOTS: it
// samples, differentiates and multiplies the windows of the live slots one after the other (the lane-parallel phases,
// unchanged) and then runs their serial phases SIDE BY SIDE: lane 3q + c sums chain c of slot q, lanes 2q and 2q + 1 run
// slot q's Jacobi step (its two independent 1/sqrt and divisions one in each lane of the pair) and keep its state.  A slot
// whose keypoint stops writes the result and takes the next keypoint of the chunk, so the slots stay full although the
// iteration counts differ (2 .. 16); iterations of different keypoints are independent, so which slot or wavefront runs a
// keypoint changes nothing.  Operands and order of every f32 / f64 operation are those of k_baumberg.
// On its own the launch is no faster than k_baumberg (the dependent chains bound it); it issues 17 % fewer vector
// instructions, and vector issue is what the pipeline's concurrent streams compete for.
MX_D void wave_lds_sync() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }   // LDS hand-over inside one wavefront
#ifdef BAUM_TRACE
// debugging aid (tools/trace_baumberg.py): shader-clock time of a wavefront per phase of its iterations, summed over the launch
__device__ unsigned long long g_btrace[16];
#define BT(i) do { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); const unsigned long long t_ = __builtin_readcyclecounter(); bt_[i] += t_ - btl_; btl_ = t_; } while (0)
#else
#define BT(i) do { } while (0)
#endif

#ifndef MODSX_BAUMBERG_K
#define MODSX_BAUMBERG_K 2
#endif
template <int K>
__global__ __launch_bounds__(64) void k_baumberg_stream(const AffJob *jobs, AffOut *out, int n, const float *mask, int chunk,
                                                        int nchunks, int maxIter, float convTh, float affInitialSigma) {
  constexpr int W = AW_MAX, WW = W * W, half = W >> 1, CH = AW_MAX * AW_MAX + 3;
  constexpr int PERM = (WW + 63) / 64;
  __shared__ __attribute__((aligned(16))) float buf[K][3][CH];
  const int lane = threadIdx.x;
  float vmask[PERM];
  int pp[PERM], oxp[PERM], oxm[PERM], oyp[PERM], oym[PERM];   // pixel (clamped to the window) and its gradient neighbours
#pragma unroll
  for (int u = 0; u < PERM; u++) {
    const int i = lane + 64 * u;
    vmask[u] = i < WW ? mask[i] : 0.f;
    const int p = i < WW ? i : WW - 1;
    const int r = p / W, c = p - r * W;
    pp[u] = p;
    oxp[u] = c == W - 1 ? p : p + 1;
    oxm[u] = c == 0 ? p : p - 1;
    oyp[u] = r == W - 1 ? p : p + W;
    oym[u] = r == 0 ? p : p - W;
  }
  // serial side: lanes 2q, 2q + 1 keep the state of slot q
  const int kq = lane >> 1;
  const bool slotLane = lane < 2 * K;
  float u11 = 1.0f, u12 = 0.0f, u21 = 0.0f, u22 = 1.0f, l1 = 1.0f, l2 = 1.0f, era = 0.0f, erb = 0.0f, ratio = 0.f;
  float slx = 0.f, sly = 0.f;   // the keypoint's position in its pyramid level (x / pixelDistance): formed once per keypoint
  int scols = 4, srows = 4;     // size of that level (interpolateCheckBorders is evaluated in the slot's state lanes)
  int ok = 0, it = 0, kidx = -1;
  // the job list is image-major in detection order (octave, level, row, column): XCD x takes the x-th contiguous eighth of
  // the chunks, so that the blur planes a keypoint's windows read stay in ONE L2 (round 2: 1.03 GB fetched per launch, every
  // window a miss)
  int next = min(xcd_chunk(blockIdx.x, nchunks) * chunk, n);           // wave-uniform: next keypoint of the chunk
  const int end = min(next + chunk, n);
  bool live = false;
  if (maxIter <= 0) {   // no iteration at all: identity shape, not converged (the loop of the reference does not run)
    for (int i = next + lane; i < end; i += 64) { AffOut o; o.u11 = 1; o.u12 = 0; o.u21 = 0; o.u22 = 1; o.ok = 0; o.iters = 0; out[i] = o; }
    return;
  }
#ifdef BAUM_TRACE
  unsigned long long bt_[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, btl_ = __builtin_readcyclecounter();
#endif
  for (;;) {
    // refill: every idle slot takes the next keypoint of the chunk, in slot order
    {
      const bool want = slotLane && !live;
      const unsigned long long wm = __ballot(want && !(lane & 1));           // one bit per idle slot (its even lane)
      const int rank = __popcll(wm & ((1ull << (lane & ~1)) - 1ull));
      const int cand = next + rank;
      if (want && cand < end) {
        kidx = cand;
        const AffJob sj = jobs[cand];
        ratio = sj.s / (affInitialSigma * sj.pixelDistance);
        slx = sj.x / sj.pixelDistance; sly = sj.y / sj.pixelDistance;
        scols = sj.cols; srows = sj.rows;
        u11 = 1.0f; u12 = 0.0f; u21 = 0.0f; u22 = 1.0f; l1 = 1.0f; l2 = 1.0f; era = 0.0f; erb = 0.0f;
        ok = 0; it = 0;
        live = true;
      }
      next = min(next + __popcll(wm), end);
    }
    const unsigned long long liveMask = __ballot(live);
    if (!liveMask) break;
    BT(0);
    const float A11 = u11 * ratio, A12 = u12 * ratio, A21 = u21 * ratio, A22 = u22 * ratio;
    // interpolateCheckBorders (helpers.cpp:524-549) of every slot at once, each in its own state lanes: one evaluation per
    // iteration instead of one per slot behind six broadcasts of the slot's matrix and position
    const unsigned long long touchMask = __ballot(check_borders(scols, srows, slx, sly, A11, A12, A21, A22, W, W));
    if constexpr (K % 2 == 0) {
#pragma unroll
      for (int sp = 0; sp < K; sp += 2) {       // the slots two at a time (K = 2: once)
      // Sample coordinates of BOTH slots at once (helpers.cpp:563-585: f32 running sums down the rows, then along each row).
      // The four chains -- slot 0 x, slot 0 y, slot 1 x, slot 1 y -- take one DPP row of 16 lanes each.  Row starts: lane i of
      // a DPP row needs i steps of its chain, which `v = v[lane - 1] + step` (row_shr:1; lane 0 of a row has no source and is
      // left alone) delivers for all 64 lanes in 15 instructions: lane i's value is final after step i and recomputed to the
      // same number afterwards.  Rows 16..18 continue from lane 15.  Then every lane walks its row (19 columns), the lanes
      // 0..2 of a DPP row a second one (rows 16..18) beside it in the other half of a packed add; the other lanes park that
      // half in the slot's third array, which is free until the taps are stored.  ~50 vector instructions per iteration for
      // what took ~150 per slot when each slot ran its x and y chains in the same 19 lanes.
      const int dq = sp + (lane >> 5), dc = (lane >> 4) & 1, di = lane & 15;   // slot, x / y chain, lane of the DPP row
      const float s11 = __shfl(A11, 2 * dq), s12 = __shfl(A12, 2 * dq), s21 = __shfl(A21, 2 * dq), s22 = __shfl(A22, 2 * dq);
      const float sox = __shfl(slx, 2 * dq), soy = __shfl(sly, 2 * dq);   // (both by every lane: a shuffle reads nothing from an idle lane)
      const float so = dc ? soy : sox;
      const float stepR = dc ? s22 : s12, stepC = dc ? s21 : s11;
      float v = so - (float)half * stepR;
      // (2 wait states between a VALU write and a DPP read of the same register; the hazard recognizer does not look inside asm)
#define MX_ROWSTEP "s_nop 1\n\tv_add_f32_dpp %0, %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
      asm volatile("s_nop 4\n\t" MX_ROWSTEP MX_ROWSTEP MX_ROWSTEP MX_ROWSTEP MX_ROWSTEP MX_ROWSTEP MX_ROWSTEP MX_ROWSTEP MX_ROWSTEP MX_ROWSTEP
                   MX_ROWSTEP MX_ROWSTEP MX_ROWSTEP MX_ROWSTEP MX_ROWSTEP "s_nop 1"
                   : "+v"(v)
                   : "v"(stepR));
#undef MX_ROWSTEP
      const float e16 = __shfl(v, lane | 15) + stepR, e17 = e16 + stepR, e18 = e17 + stepR;
      const float vx = di == 0 ? e16 : (di == 1 ? e17 : e18);
      const float hc = (float)half * stepC;
      typedef float v2f __attribute__((ext_vector_type(2)));
      v2f wv = {v - hc, vx - hc};
      const v2f st = {stepC, stepC};
      float *const b1 = &buf[dq][dc][di * W];
      float *const b2 = di < 3 ? &buf[dq][dc][(16 + di) * W] : &buf[dq][2][0];
#pragma unroll
      for (int i = 0; i < W; i++) {
        b1[i] = wv.x;
        b2[i] = wv.y;
        wv += st;
      }
      }
    }
    BT(1);
#pragma unroll
    for (int q = 0; q < K; q++) {
      if (!((liveMask >> (2 * q)) & 1)) continue;
#ifdef MODSX_BAUM_VECTOR_JOB   // round-5 form: the job through a vector index (the level's base is then a VGPR pair per tap)
      const AffJob jb = jobs[__shfl(kidx, 2 * q)];
#else
      // the slot's job through a SCALAR index (v_readlane of the slot's even lane): its fields arrive by scalar loads, so the
      // level's base pointer, rows and cols are SGPRs and a tap's loads take the scalar base + 32-bit vector offset form --
      // no 64-bit vector address per tap row (two v_lshl_add_u64 + two moves of the 20 vector instructions of a tap)
      const AffJob jb = jobs[__builtin_amdgcn_readlane(kidx, 2 * q)];
#endif
      const gcfloat_p img = as_global(jb.blur);
      float *const pa = buf[q][0], *const pb = buf[q][1], *const pc = buf[q][2];
      float *const simg = pc;
      const bool touch = (touchMask >> (2 * q)) & 1;
#ifdef BAUM_TRACE
      const float a11 = __shfl(A11, 2 * q), a12 = __shfl(A12, 2 * q), a21 = __shfl(A21, 2 * q), a22 = __shfl(A22, 2 * q);
#endif
#ifdef BAUM_TRACE
      if (lane == 0) {     // extent of the sampled window in the level: what a staged source tile would have to hold
        const float bw = 18.f * (fabsf(a11) + fabsf(a12)) + 3.f, bh = 18.f * (fabsf(a21) + fabsf(a22)) + 3.f;
        const float m = fmaxf(bw, bh);
        const int b = m <= 16.f ? 0 : m <= 24.f ? 1 : m <= 32.f ? 2 : m <= 40.f ? 3 : m <= 48.f ? 4 : m <= 64.f ? 5 : 6;
        atomicAdd(&g_btrace[9 + b], 1ull);
      }
#endif
      if constexpr (K % 2 != 0) {
        // sample coordinates: lane j runs the f32 running sums of row j (helpers.cpp:563-585) into LDS
        const float lx = __shfl(slx, 2 * q), ly = __shfl(sly, 2 * q);
        const float a11 = __shfl(A11, 2 * q), a12 = __shfl(A12, 2 * q), a21 = __shfl(A21, 2 * q), a22 = __shfl(A22, 2 * q);
        float rx = lx - (float)half * a12, ry = ly - (float)half * a22;
#pragma unroll
        for (int j = 1; j < W; j++)
          if (lane >= j) { rx += a12; ry += a22; }
        if (lane < W) {
          float WX = rx - (float)half * a11;
          float WY = ry - (float)half * a21;
#pragma unroll
          for (int i = 0; i < W; i++) {
            pa[lane * W + i] = WX;
            pb[lane * W + i] = WY;
            WX += a11;
            WY += a21;
          }
        }
      }
      wave_lds_sync();
      {
        // all six taps of a lane in flight together; slots past the window repeat its last pixel and are not stored
        float sv[PERM];
        if (!touch) {
#pragma unroll
          for (int u = 0; u < PERM; u++) sv[u] = bilinear_tap(img, jb.rows, jb.cols, pa[pp[u]], pb[pp[u]], false);
        } else {
#pragma unroll
          for (int u = 0; u < PERM; u++) sv[u] = bilinear_tap_touch_select(img, jb.rows, jb.cols, pa[pp[u]], pb[pp[u]]);
        }
        wave_lds_sync();
#pragma unroll
        for (int u = 0; u < PERM; u++)
          if (lane + 64 * u < WW) simg[lane + 64 * u] = sv[u];
      }
      wave_lds_sync();
      BT(2 + 2 * (q & 1));
      {
        float qa[PERM], qb[PERM], qc[PERM];
#pragma unroll
        for (int u = 0; u < PERM; u++) {
          const float gx = simg[oxp[u]] - simg[oxm[u]];
          const float gy = simg[oyp[u]] - simg[oym[u]];
          const float v = vmask[u];
          const float gxy = gx * gy;
          qa[u] = gx * gx * v;
          qb[u] = gxy * v;
          qc[u] = gy * gy * v;
        }
        wave_lds_sync();   // every gradient is taken: pc may replace the window
#pragma unroll
        for (int u = 0; u < PERM; u++) {
          const int p = lane + 64 * u;
          if (p < WW) { pa[p] = qa[u]; pb[p] = qb[u]; pc[p] = qc[u]; }
        }
      }
      BT(3 + 2 * (q & 1));
    }
    wave_lds_sync();
    float acc = 0.f;
    if (lane < 3 * K) {
      const int q3 = lane / 3, ch = lane - 3 * q3;
      const float *arr = buf[q3][ch];
      const float4 *a4 = reinterpret_cast<const float4 *>(arr);
#pragma unroll 6
      for (int i = 0; i < WW / 4; i++) { const float4 v = a4[i]; acc += v.x; acc += v.y; acc += v.z; acc += v.w; }
      acc += arr[WW - 1];
      acc /= (float)WW;
    }
    float a = __shfl(acc, 3 * kq), b = __shfl(acc, 3 * kq + 1), c = __shfl(acc, 3 * kq + 2);
    BT(6);
    if (live) {
      {   // invSqrt, helpers.cpp:463-502 (inv_sqrt_wave with the partner lane in the place of lanes 0 / 1)
        double t, r;
        if (b != 0) {
          r = double(c - a) / (2 * b);
          if (r >= 0) t = 1.0 / (r + sqrt(1 + r * r));
          else t = -1.0 / (-r + sqrt(1 + r * r));
          r = 1.0 / sqrt(1 + t * t);
          t = t * r;
        } else { r = 1; t = 0; }
        const bool odd = lane & 1;
        const double rr = r * r, tt = t * t, m = 2 * r * t * b;
        const double first = (odd ? tt : rr) * a, last = (odd ? rr : tt) * c;
        double w = 1.0 / sqrt((odd ? first + m : first - m) + last);
        double x = __shfl(w, lane & ~1), z = __shfl(w, lane | 1);
        const double d = sqrt(x * z);
        w = (odd ? z : x) / d;
        x = __shfl(w, lane & ~1); z = __shfl(w, lane | 1);
        if (x < z) { l1 = float(z); l2 = float(x); } else { l1 = float(x); l2 = float(z); }
        a = float(r * r * x + t * t * z);
        b = float(-r * t * x + t * r * z);
        c = float(t * t * x + r * r * z);
      }
      bool stop = false;
      int iters = it;                                 // the value of the reference's loop counter at a break
      if ((a != a) || (b != b) || (c != c)) stop = true;
      else {
        erb = era;
        era = (float)(1.0 - (double)(l2 / l1));
        const float u11t = u11, u12t = u12;
        u11 = a * u11t + b * u21;
        u12 = a * u12t + b * u22;
        u21 = b * u11t + c * u21;
        u22 = b * u12t + c * u22;
        if (!eigenvalues(u11, u12, u21, u22, l1, l2)) stop = true;
        else if ((l1 / l2 > 6) || (l2 / l1 > 6)) stop = true;
        else if (era < convTh && erb < convTh) { ok = 1; stop = true; }
      }
      it++;
      if (!stop && it >= maxIter) { stop = true; iters = maxIter; }
      if (stop) {
        live = false;
        if (!(lane & 1)) {
          AffOut o;
          o.u11 = u11; o.u12 = u12; o.u21 = u21; o.u22 = u22; o.ok = ok; o.iters = iters;
          out[kidx] = o;
        }
      }
    }
    BT(7);
  }
#ifdef BAUM_TRACE
  if (lane == 0) {
    for (int i = 0; i < 8; i++) atomicAdd(&g_btrace[i], bt_[i]);
    atomicAdd(&g_btrace[8], 1ull);
  }
#endif
}

#ifndef MODSX_BAUMBERG_CHUNK
#define MODSX_BAUMBERG_CHUNK 8
#endif

void launch_baumberg(hipStream_t s, const AffJob *jobs, AffOut *out, int n, const float *mask, int W, int maxIter,
                     float convTh, float affInitialSigma) {
  if (n <= 0) return;
  constexpr int K = MODSX_BAUMBERG_K;
  if (W == 19 && K > 1) {
    // chunk per wavefront: long enough to keep the slots full across keypoints of different iteration counts, short enough
    // for >= 4 rounds of wavefronts over the chip (256 CUs x 16 resident) so that the tail of the launch stays short
    int chunk = n / 16384;
    chunk = chunk < K ? K : (chunk > MODSX_BAUMBERG_CHUNK ? MODSX_BAUMBERG_CHUNK : chunk);
    const int nchunks = (n + chunk - 1) / chunk;
    MX_DUP(K_BAUMBERG) hipLaunchKernelGGL(k_baumberg_stream<K>, dim3(8 * ((nchunks + 7) / 8)), dim3(64), 0, s, jobs, out, n, mask, chunk, nchunks, maxIter,
                       convTh, affInitialSigma);
  } else if (W == 19) hipLaunchKernelGGL(k_baumberg<19>, dim3(n), dim3(64), 0, s, jobs, out, n, mask, W, maxIter, convTh, affInitialSigma);
  else hipLaunchKernelGGL(k_baumberg<0>, dim3(n), dim3(64), 0, s, jobs, out, n, mask, W, maxIter, convTh, affInitialSigma);
}

}  // namespace mx

#ifdef BAUM_TRACE
extern "C" __attribute__((visibility("default"))) int modsx_debug_baum_trace(unsigned long long *out, int reset) {
  int rc = (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(mx::g_btrace), 16 * 8, 0, hipMemcpyDeviceToHost);
  if (reset) { unsigned long long z[16] = {0}; rc |= (int)hipMemcpyToSymbol(HIP_SYMBOL(mx::g_btrace), z, sizeof z, 0, hipMemcpyHostToDevice); }
  return rc;
}
#endif
