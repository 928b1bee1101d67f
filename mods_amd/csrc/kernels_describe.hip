// kernels_describe.hip -- affine-normalised patch extraction and SIFT / RootSIFT description.
//
// Reference: DescribeRegions<>, synth-detection.hpp:169-255 (slow/accurate and fast branches)
//   interpolate                  detectors/helpers.cpp:551-626
//   gaussianBlurInplace          detectors/helpers.cpp:726-731 (cv::GaussianBlur, BORDER_REPLICATE)
//   photometricallyNormalize     detectors/helpers.cpp:666-715
//   SIFTDescriptor               matching/siftdesc.cpp:22-131 (bins, samplePatch), 136-278 (norms),
//                                290-379 (gradients + atan2LUTff)
// Data layout: every region owns a dense P x P f32 window in a scratch arena in HBM
// (P = 2*ceil(s*mrSize)+3), written once by k_patch_sample, blurred by two separable passes
// (arena A -> B -> A) and read once by k_describe, which resamples it to 41x41 in LDS.
#include "engine.hpp"

namespace mx {

MX_D int find_job(const int *prefix, int nJobs, int tile) {
  int lo = 0, hi = nJobs;  // prefix[lo] <= tile < prefix[hi]
  while (hi - lo > 1) {
    int mid = (lo + hi) >> 1;
    if (prefix[mid] <= tile) lo = mid; else hi = mid;
  }
  return lo;
}

// --- stage 1: interpolate(img, x, y, A, smoothed(P x P)) ---------------------------------------
// One wavefront per 64 rows of one window.  Lane j walks row j left to right (sample coordinates are
// f32 running sums); 32-column chunks are transposed through LDS so the stores are row-contiguous.
__global__ __launch_bounds__(64) void k_patch_sample(const DescJob *jobs, const int *tilePrefix, int nJobs,
                                                     const ImgRef *imgs, float *scratch) {
  const int tile = blockIdx.x;
  const int jid = find_job(tilePrefix, nJobs, tile);
  const DescJob jb = jobs[jid];
  const int P = jb.P;
  if (P <= 0) return;
  const int lane = threadIdx.x;
  const int row0 = (tile - tilePrefix[jid]) * 64;
  const int row = row0 + lane;
  const ImgRef im = imgs[jb.img];
  __shared__ float tbuf[64 * 33];
  const int half = P >> 1;
  const bool touch = check_borders(im.cols, im.rows, jb.x, jb.y, jb.a11, jb.a12, jb.a21, jb.a22, P, P);
  float rx = jb.x - (float)half * jb.a12;
  float ry = jb.y - (float)half * jb.a22;
  const int nsteps = row < P ? row : 0;
  for (int j = 0; j < nsteps; j++) { rx += jb.a12; ry += jb.a22; }
  float WX = rx - (float)half * jb.a11;
  float WY = ry - (float)half * jb.a21;
  float *dst = scratch + jb.scratchOfs;
  const int rowsHere = (P - row0) < 64 ? (P - row0) : 64;
  for (int c0 = 0; c0 < P; c0 += 32) {
    const int nc = (P - c0) < 32 ? (P - c0) : 32;
    if (row < P) {
      for (int i = 0; i < nc; i++) {
        tbuf[lane * 33 + i] = bilinear_tap(im.d, im.rows, im.cols, WX, WY, touch);
        WX += jb.a11;
        WY += jb.a21;
      }
    }
    __syncthreads();
    for (int e = lane; e < rowsHere * 32; e += 64) {
      const int r = e >> 5, c = e & 31;
      if (c < nc) dst[(size_t)(row0 + r) * P + c0 + c] = tbuf[r * 33 + c];
    }
    __syncthreads();
  }
}

// --- stage 2: separable Gaussian blur of each window, replicate border -------------------------
// pass 0: rows (cv RowFilter: taps left->right; SymmRowSmallFilter when ksize <= 5)
// pass 1: columns (SymmColumnFilter: centre + (below + above) * k)
__global__ __launch_bounds__(256) void k_patch_blur(const DescJob *jobs, const int *tilePrefix, int nJobs,
                                                    const float *taps, const float *src, float *dst, int pass) {
  const int tile = blockIdx.x;
  const int jid = find_job(tilePrefix, nJobs, tile);
  const DescJob jb = jobs[jid];
  const int P = jb.P;
  if (P <= 0) return;
  const int px = (tile - tilePrefix[jid]) * 256 + threadIdx.x;
  if (px >= P * P) return;
  const int r = px / P, c = px - r * P;
  const int n = jb.ksize, R = n >> 1;
  const float *k = taps + jb.tapOfs;
  const float *S = src + jb.scratchOfs;
  float v;
  if (n == 1) v = S[px];
  else if (pass == 0) {
    const float *row = S + (size_t)r * P;
    if (n <= 5) {
      v = row[c] * k[R];
      for (int j = 1; j <= R; j++) {
        int cm = c - j < 0 ? 0 : c - j, cp = c + j > P - 1 ? P - 1 : c + j;
        v = v + (row[cm] + row[cp]) * k[R + j];
      }
    } else {
      v = 0.f;
      for (int j = 0; j < n; j++) {
        int cc = c + j - R;
        cc = cc < 0 ? 0 : (cc > P - 1 ? P - 1 : cc);
        v = v + row[cc] * k[j];
      }
    }
  } else {
    v = k[R] * S[px] + 0.f;
    for (int j = 1; j <= R; j++) {
      int rp = r + j > P - 1 ? P - 1 : r + j, rm = r - j < 0 ? 0 : r - j;
      v = v + k[R + j] * (S[(size_t)rp * P + c] + S[(size_t)rm * P + c]);
    }
  }
  dst[jb.scratchOfs + px] = v;
}

// --- stage 3: 41x41 patch, photometric normalisation, SIFT histogram ----------------------------
constexpr int PS = 41, NPX = PS * PS;

struct SiftConst {
  int lo[4], hi[4];  // pixel index range touching spatial bin k (rows and columns alike)
};

__global__ __launch_bounds__(128) void k_describe(const DescJob *jobs, int n, const ImgRef *imgs, const float *scratch,
                                                  const float *mask, const double *atanLut, const int *binTab,
                                                  const double *wTab, SiftConst sc, int photoNorm, int rootsift,
                                                  double maxBin, float *descF, uint8_t *descU8) {
  const int k = blockIdx.x;
  if (k >= n) return;
  const int tid = threadIdx.x;
  __shared__ float patch[NPX];
  __shared__ float sval[NPX];   // mask * gradient magnitude
  __shared__ float swo1[NPX];   // orientation interpolation weight
  __shared__ unsigned char sbo0[NPX];
  __shared__ float smask[NPX];
  __shared__ int sbin0[PS], sbin1[PS];
  __shared__ double sw0[PS], sw1[PS];
  __shared__ double vec[128];
  __shared__ float sstat[2];
  const DescJob jb = jobs[k];
  for (int i = tid; i < NPX; i += 128) smask[i] = mask[i];
  if (tid < PS) { sbin0[tid] = binTab[tid]; sbin1[tid] = binTab[PS + tid]; sw0[tid] = wTab[tid]; sw1[tid] = wTab[PS + tid]; }
  // -- resample to 41x41: lane j walks row j
  if (tid < PS) {
    const float *src; int srows, scols; float ox, oy, a11, a12, a21, a22;
    if (jb.P > 0) {
      src = scratch + jb.scratchOfs; srows = jb.P; scols = jb.P;
      ox = (float)(jb.P >> 1); oy = ox;
      a11 = jb.i2p; a12 = 0.f; a21 = 0.f; a22 = jb.i2p;
    } else {
      const ImgRef im = imgs[jb.img];
      src = im.d; srows = im.rows; scols = im.cols;
      ox = jb.x; oy = jb.y; a11 = jb.a11; a12 = jb.a12; a21 = jb.a21; a22 = jb.a22;
    }
    const int half = PS >> 1;
    const bool touch = check_borders(scols, srows, ox, oy, a11, a12, a21, a22, PS, PS);
    float rx = ox - (float)half * a12;
    float ry = oy - (float)half * a22;
    for (int j = 0; j < tid; j++) { rx += a12; ry += a22; }
    float WX = rx - (float)half * a11;
    float WY = ry - (float)half * a21;
    for (int i = 0; i < PS; i++) {
      patch[tid * PS + i] = bilinear_tap(src, srows, scols, WX, WY, touch);
      WX += a11;
      WY += a21;
    }
  }
  __syncthreads();
  // -- photometricallyNormalize: sequential f32 sums over the masked pixels (order dependent)
  if (photoNorm) {
    if (tid == 0) {
      float sum = 0.f, gsum = 0.f;
      for (int i = 0; i < NPX; i++)
        if (smask[i] > 0) { sum += patch[i]; gsum += 1.0f; }
      sum = sum / gsum;
      float var = 0.f;
      for (int i = 0; i < NPX; i++)
        if (smask[i] > 0) var += (sum - patch[i]) * (sum - patch[i]);
      var = sqrtf(var / gsum);
      sstat[0] = sum; sstat[1] = var;
    }
    __syncthreads();
    const float sum = sstat[0], var = sstat[1];
    if (!((double)var < 0.0001)) {
      const float fac = 50.0f / var;
      for (int i = tid; i < NPX; i += 128) {
        float v = 128.f + fac * (patch[i] - sum);
        if (v > 255.f) v = 255.f;
        if (v < 0.f) v = 0.f;
        patch[i] = v;
      }
    }
    __syncthreads();
  }
  // -- gradients, orientation, per-pixel weights
  const double TWO_PI = 6.28318530718;
  for (int p = tid; p < NPX; p += 128) {
    const int r = p / PS, c = p - r * PS;
    float xg, yg;
    if (c == 0) xg = patch[p + 1] - patch[p];
    else if (c == PS - 1) xg = patch[p] - patch[p - 1];
    else xg = patch[p + 1] - patch[p - 1];
    if (r == 0) yg = patch[p + PS] - patch[p];
    else if (r == PS - 1) yg = patch[p] - patch[p - PS];
    else yg = patch[p + PS] - patch[p - PS];
    const float g = sqrtf(xg * xg + yg * yg);
    const float ori = atan2lut(atanLut, yg, xg);
    const float val = (float)(0.0 + (1.0 * (double)smask[p]) * (double)g);
    const float o = (float)((double)8.0f * ((double)ori + TWO_PI) / TWO_PI);
    int bo0 = (int)o;
    swo1[p] = o - (float)bo0;
    sbo0[p] = (unsigned char)(bo0 % 8);
    sval[p] = val;
  }
  if (tid < 128) vec[tid] = 0.0;
  __syncthreads();
  // -- samplePatch: thread = one of the 128 bins, gathers its pixels in raster order (f64 accumulator)
  {
    const int rb = tid >> 5, cb = (tid >> 3) & 3, ob = tid & 7;
    const int binR = rb * 8, binC = cb * 8;  // bin0/bin1 tables hold spatialBin*8
    double acc = 0.0;
    for (int r = sc.lo[rb]; r <= sc.hi[rb]; r++) {
      const float wr0 = (float)sw0[r], wr1 = (float)sw1[r];
      const bool r0m = sbin0[r] == binR, r1m = sbin1[r] == binR;
      if (!r0m && !r1m) continue;
      for (int c = sc.lo[cb]; c <= sc.hi[cb]; c++) {
        const bool c0m = sbin0[c] == binC, c1m = sbin1[c] == binC;
        if (!c0m && !c1m) continue;
        const int p = r * PS + c;
        const int bo0 = sbo0[p];
        const int bo1 = (bo0 + 1) % 8;
        if (bo0 != ob && bo1 != ob) continue;
        const float val = sval[p];
        const float wc0 = (float)(sw0[c] * (double)val);
        const float wc1 = (float)(sw1[c] * (double)val);
        const float wo1 = swo1[p];
        const float wo0 = 1.0f - wo1;
        const float wo = (bo0 == ob) ? wo0 : wo1;
        float v;
        if (r0m && c0m) { v = wr0 * wc0; if (v > 0) acc += (double)(v * wo); }
        if (r0m && c1m) { v = wr0 * wc1; if (v > 0) acc += (double)(v * wo); }
        if (r1m && c0m) { v = wr1 * wc0; if (v > 0) acc += (double)(v * wo); }
        if (r1m && c1m) { v = wr1 * wc1; if (v > 0) acc += (double)(v * wo); }
      }
    }
    vec[tid] = acc;
  }
  __syncthreads();
  // -- normalize / clip / renormalize / (RootSIFT) / quantise -- sequential f64 sums on one lane
  __shared__ double sfac;
  __shared__ int schanged;
  for (int pass = 0; pass < 2; pass++) {
    if (tid == 0) {
      double len = 0.0;
      for (int i = 0; i < 128; i += 4) {
        const double s0 = vec[i] * vec[i], s1 = vec[i + 1] * vec[i + 1], s2 = vec[i + 2] * vec[i + 2],
                     s3 = vec[i + 3] * vec[i + 3];
        len += s0 + s1 + s2 + s3;
      }
      len = sqrt(len);
      sfac = 1.0 / len;
      schanged = 0;
    }
    __syncthreads();
    vec[tid] *= sfac;
    __syncthreads();
    if (pass == 0) {
      if (vec[tid] > maxBin) { vec[tid] = maxBin; schanged = 1; }
      __syncthreads();
      if (!schanged) break;
    }
  }
  __syncthreads();
  if (rootsift) {
    if (tid == 0) {
      double sum = 0.;
      for (int i = 0; i < 128; i++) sum += fabs(vec[i]);
      sfac = sum;
    }
    __syncthreads();
    vec[tid] = sqrt(vec[tid] / sfac);
  }
  {
    int b;
    if (rootsift) b = (int)(512.0 * vec[tid] + 0.5);
    else b = (int)((double)512.0f * vec[tid] + 0.5);
    b = b < 255 ? b : 255;
    b = b > 0 ? b : 0;
    descF[(size_t)k * 128 + tid] = (float)b;
    descU8[(size_t)k * 128 + tid] = (uint8_t)b;
  }
}

void launch_patch_sample(hipStream_t s, const DescJob *jobs, const int *tilePrefix, int nJobs, int nTiles,
                         const ImgRef *imgs, float *scratch) {
  if (nTiles <= 0) return;
  hipLaunchKernelGGL(k_patch_sample, dim3(nTiles), dim3(64), 0, s, jobs, tilePrefix, nJobs, imgs, scratch);
}
void launch_patch_blur(hipStream_t s, const DescJob *jobs, const int *tilePrefix, int nJobs, int nTiles,
                       const float *taps, const float *src, float *dst, int pass) {
  if (nTiles <= 0) return;
  hipLaunchKernelGGL(k_patch_blur, dim3(nTiles), dim3(256), 0, s, jobs, tilePrefix, nJobs, taps, src, dst, pass);
}
void launch_describe(hipStream_t s, const DescJob *jobs, int n, const ImgRef *imgs, const float *scratch,
                     const float *mask, const double *atanLut, const int *bins, const double *wts, int photoNorm,
                     int rootsift, double maxBin, float *descF, uint8_t *descU8) {
  if (n <= 0) return;
  // spatial bin k receives pixels i with bin0[i]/8 == k or bin1[i]/8 == k; with step 5/40 that is the
  // contiguous range [8k-?, 8k+15]; computed by the host from the same tables (engine.cpp) -> here the
  // ranges are conservative supersets: rows whose bins do not match are skipped inside the kernel.
  SiftConst sc;
  for (int k = 0; k < 4; k++) { sc.lo[k] = 8 * k - 8 < 0 ? 0 : 8 * k - 8; sc.hi[k] = 8 * k + 16 > 40 ? 40 : 8 * k + 16; }
  sc.hi[3] = 40;
  hipLaunchKernelGGL(k_describe, dim3(n), dim3(128), 0, s, jobs, n, imgs, scratch, mask, atanLut, bins, wts, sc,
                     photoNorm, rootsift, maxBin, descF, descU8);
}

}  // namespace mx
