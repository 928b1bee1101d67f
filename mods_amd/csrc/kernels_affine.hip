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

// WT = window size when known at compile time (19, the default smmWindowSize), 0 = use the argument: the row / column of a
// pixel is an integer division by W in three loops of every iteration, ~25 instructions each unless W is a constant
template <int WT>
__global__ __launch_bounds__(64) void k_baumberg(const AffJob *jobs, AffOut *out, int n, const float *mask, int Warg,
                                                 int maxIter, float convTh, float affInitialSigma) {
  const int W = WT ? WT : Warg;
  const int k = blockIdx.x;
  if (k >= n) return;
  const int lane = threadIdx.x;
  // 4.4 KB of LDS per keypoint: the kernel is a chain of dependent phases per iteration (coordinates -> taps -> gradients ->
  // 361 ordered adds -> Jacobi), so what it needs is many keypoints in flight per CU.  The window mask stays in registers
  // (it is the same every iteration), and the sampled window shares its buffer with the third product array: the
  // products are staged in registers and written after every lane has taken its gradients.
  __shared__ __attribute__((aligned(16))) float pa[AW_MAX * AW_MAX + 3], pb[AW_MAX * AW_MAX + 3], pc[AW_MAX * AW_MAX + 3];
  float *const simg = pc;
  const AffJob jb = jobs[k];
  const int WW = W * W, half = W >> 1;
  constexpr int PERM = (AW_MAX * AW_MAX + 63) / 64;
  float vmask[PERM];
#pragma unroll
  for (int u = 0; u < PERM; u++) { const int i = lane + 64 * u; vmask[u] = i < WW ? mask[i] : 0.f; }
  float u11 = 1.0f, u12 = 0.0f, u21 = 0.0f, u22 = 1.0f, l1 = 1.0f, l2 = 1.0f;
  float era = 0.0f, erb = 0.0f;
  const float lx = jb.x / jb.pixelDistance, ly = jb.y / jb.pixelDistance;
  const float ratio = jb.s / (affInitialSigma * jb.pixelDistance);
  int ok = 0, l = 0;
  __syncthreads();
  for (l = 0; l < maxIter; l++) {
    const float a11 = u11 * ratio, a12 = u12 * ratio, a21 = u21 * ratio, a22 = u22 * ratio;
    const bool touch = check_borders(jb.cols, jb.rows, lx, ly, a11, a12, a21, a22, W, W);
    // sample coordinates: lane j runs the f32 running sums of row j (helpers.cpp:563-585) into LDS,
    // then all lanes take the bilinear taps
    {
      // row starts: ONE chain of running sums (the same for every lane, a12 / a22 are uniform), lane j keeps step j --
      // no divergent loop; then lane j walks its row with the loop unrolled when the window size is a constant
      float cxr = lx - (float)half * a12, cyr = ly - (float)half * a22;
      float rx = cxr, ry = cyr;
      if (WT) {
#pragma unroll
        for (int j = 1; j < (WT ? WT : 1); j++) {
          cxr += a12; cyr += a22;
          if (lane == j) { rx = cxr; ry = cyr; }
        }
      } else {
        rx = lx - (float)half * a12; ry = ly - (float)half * a22;
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
    if (WT) {
      // all six taps of a lane in flight together: one memory round trip per iteration instead of six
      constexpr int PER = ((WT ? WT : 1) * (WT ? WT : 1) + 63) / 64;
      float sv[PER];
      if (!touch) {
#pragma unroll
        for (int u = 0; u < PER; u++) {
          const int p = lane + 64 * u;
          sv[u] = p < WW ? bilinear_tap(as_global(jb.blur), jb.rows, jb.cols, pa[p], pb[p], false) : 0.f;
        }
      } else {
#pragma unroll
        for (int u = 0; u < PER; u++) {
          const int p = lane + 64 * u;
          sv[u] = p < WW ? bilinear_tap(as_global(jb.blur), jb.rows, jb.cols, pa[p], pb[p], true) : 0.f;
        }
      }
#pragma unroll
      for (int u = 0; u < PER; u++) {
        const int p = lane + 64 * u;
        if (p < WW) simg[p] = sv[u];
      }
    } else {
      for (int p = lane; p < WW; p += 64) simg[p] = bilinear_tap(as_global(jb.blur), jb.rows, jb.cols, pa[p], pb[p], touch);
    }
    __syncthreads();
    {
      float qa[PERM], qb[PERM], qc[PERM];
#pragma unroll
      for (int u = 0; u < PERM; u++) {
        const int p = lane + 64 * u;
        qa[u] = qb[u] = qc[u] = 0.f;
        if (p < WW) {
          const int r = p / W, c = p - r * W;
          float gx, gy;
          if (c == 0) gx = simg[p + 1] - simg[p];
          else if (c == W - 1) gx = simg[p] - simg[p - 1];
          else gx = simg[p + 1] - simg[p - 1];
          if (r == 0) gy = simg[p + W] - simg[p];
          else if (r == W - 1) gy = simg[p] - simg[p - W];
          else gy = simg[p + W] - simg[p - W];
          const float v = vmask[u];
          const float gxy = gx * gy;
          qa[u] = gx * gx * v;
          qb[u] = gxy * v;
          qc[u] = gy * gy * v;
        }
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
    }
    float a = __shfl(acc, 0), b = __shfl(acc, 1), c = __shfl(acc, 2);
    a /= (float)WW; b /= (float)WW; c /= (float)WW;
    inv_sqrt(a, b, c, l1, l2);
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

void launch_baumberg(hipStream_t s, const AffJob *jobs, AffOut *out, int n, const float *mask, int W, int maxIter,
                     float convTh, float affInitialSigma) {
  if (n <= 0) return;
  if (W == 19) hipLaunchKernelGGL(k_baumberg<19>, dim3(n), dim3(64), 0, s, jobs, out, n, mask, W, maxIter, convTh, affInitialSigma);
  else hipLaunchKernelGGL(k_baumberg<0>, dim3(n), dim3(64), 0, s, jobs, out, n, mask, W, maxIter, convTh, affInitialSigma);
}

}  // namespace mx
