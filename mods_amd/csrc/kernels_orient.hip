// kernels_orient.hip -- dominant orientation of affine regions, one wavefront per region.
//
// Reference: DetectOrientation + EstimateDominantAnglesFunctor, synth-detection.cpp:746-919
//   interpolate (41x41 patch, A * curr_sc)                 detectors/helpers.cpp:551-626
//   computeGradientMagnitudeAndOrientation + atan2LUTff     detectors/helpers.cpp:840-863, 160-207
//   36-bin histogram of mag * circular Gauss mask, 6x circular smoothing, parabolic peaks,
//   first maxAngles peaks in bin order.
// Lane j walks row j of the patch (the f32 sample coordinates are running sums along a row);
// histogram bin b is accumulated by lane b scanning the pixels in raster order, which keeps
// the reference's f32 summation order per bin.
#include "engine.hpp"

namespace mx {

constexpr int PS = 41, PSP = 44;   // PSP: padded row stride so rows start 16-byte aligned in LDS

__global__ __launch_bounds__(64) void k_orientation(const OriJob *jobs, OriOut *out, int n, const ImgRef *imgs,
                                                    const float *orimask, const double *atanLut, int doHalf,
                                                    double th, int maxAngles) {
  const int k = blockIdx.x;
  if (k >= n) return;
  const int lane = threadIdx.x;
  __shared__ __attribute__((aligned(16))) float patch[PS * PS];
  __shared__ __attribute__((aligned(16))) float bufX[PS * PSP];   // WX, later the histogram weights
  __shared__ __attribute__((aligned(16))) float bufY[PS * PSP];   // WY
  __shared__ __attribute__((aligned(16))) unsigned char sbin[PS * PSP];
  __shared__ float hist[40];
  const OriJob jb = jobs[k];
  const ImgRef im = imgs[jb.img];
  const int half = PS >> 1;
  const bool touch = check_borders(im.cols, im.rows, jb.x, jb.y, jb.a11, jb.a12, jb.a21, jb.a22, PS, PS);
  // sample coordinates: the f32 running sums of interpolate() (helpers.cpp:563-585), one lane per row
  if (lane < PS) {
    float rx = jb.x - (float)half * jb.a12;
    float ry = jb.y - (float)half * jb.a22;
    for (int j = 0; j < lane; j++) { rx += jb.a12; ry += jb.a22; }
    float WX = rx - (float)half * jb.a11;
    float WY = ry - (float)half * jb.a21;
#pragma unroll 1
    for (int i = 0; i < PS; i++) {
      bufX[lane * PSP + i] = WX;
      bufY[lane * PSP + i] = WY;
      WX += jb.a11;
      WY += jb.a21;
    }
  }
  __syncthreads();
#pragma unroll 4
  for (int p = lane; p < PS * PS; p += 64) {
    const int r = p / PS, c = p - r * PS;
    patch[p] = bilinear_tap(im.d, im.rows, im.cols, bufX[r * PSP + c], bufY[r * PSP + c], touch);
  }
  __syncthreads();
  const float PIf = float(M_PI);
  for (int p = lane; p < PS * PSP; p += 64) {
    const int r = p / PSP, c = p - r * PSP;
    unsigned char bin = 255;
    float w = 0.f;
    if (r >= 1 && r < PS - 1 && c >= 1 && c < PS - 1) {
      const int q = r * PS + c;
      const float xg = patch[q + 1] - patch[q - 1];
      const float yg = patch[q + PS] - patch[q - PS];
      const float mag = sqrtf(xg * xg + yg * yg);
      const float ori = atan2lut(atanLut, yg, xg);
      const float m = orimask[q];
      if (m > 0 && mag > 1.0f) {
        bin = (unsigned char)(int)(36 * (ori / PIf + 1.0f) / 2.0f);
        w = mag * m;
      }
    }
    sbin[p] = bin;
    bufX[p] = w;
  }
  __syncthreads();
  // lane b owns histogram bin b and adds its pixels in raster order (f32 running sum of the reference)
  if (lane < 36) {
    float h = 0.f;
#pragma unroll 1
    for (int r = 1; r < PS - 1; r++) {
      const unsigned *b4 = reinterpret_cast<const unsigned *>(sbin + r * PSP);
      const float4 *w4 = reinterpret_cast<const float4 *>(bufX + r * PSP);
#pragma unroll
      for (int g = 0; g < PSP / 4; g++) {
        const unsigned b = b4[g];
        const float4 w = w4[g];
        if ((int)(b & 0xff) == lane) h += w.x;
        if ((int)((b >> 8) & 0xff) == lane) h += w.y;
        if ((int)((b >> 16) & 0xff) == lane) h += w.z;
        if ((int)(b >> 24) == lane) h += w.w;
      }
    }
    hist[lane] = h;
  }
  __syncthreads();
  if (lane == 0) {
    const int nb = 36;
    for (int it = 0; it < 6; it++) {
      float first = hist[0], prev = hist[nb - 1];
      for (int i = 0; i < nb - 1; i++) {
        float cur = hist[i];
        hist[i] = prev + cur + hist[i + 1];
        prev = cur;
      }
      hist[nb - 1] = prev + hist[nb - 1] + first;
    }
    float thresh = 0.0f;
    for (int i = 0; i < nb; i++) if (hist[i] > thresh) thresh = hist[i];
    thresh = (float)((double)thresh * th);
    if (doHalf) {
      for (int i = 0; i < nb / 2; i++) { hist[i] += hist[i + nb / 2]; hist[i + nb / 2] = 0; }
    }
    OriOut o;
    o.n = 0;
    int npeaks = 0;  // number of peaks found so far (peak_values.size())
    // peaks in the order (35,0,1), (i-1,i,i+1) for i = 1..34, (34,35,0); the reference keeps the first
    // min(maxAngles, #peaks) of them -- every recorded peak already satisfies hist[b] >= thresh.
    for (int q = 0; q < nb; q++) {
      const int b = q, a = (q == 0) ? nb - 1 : q - 1, c = (q == nb - 1) ? 0 : q + 1;
      if (hist[b] >= thresh && hist[b] > hist[a] && hist[b] > hist[c]) {
        if (npeaks < maxAngles && o.n < 7) {
          float pp = (hist[a] - hist[c]) / (hist[a] - 2.0f * hist[b] + hist[c]) / 2.0f;
          o.ang[o.n++] = 2.0f * PIf * ((float)b + 0.5f + pp) / (float)nb - PIf;
        }
        npeaks++;
      }
    }
    out[k] = o;
  }
}

void launch_orientation(hipStream_t s, const OriJob *jobs, OriOut *out, int n, const ImgRef *imgs,
                          const float *orimask, const double *atanLut, int doHalf, double th, int maxAngles) {
  if (n <= 0) return;
  hipLaunchKernelGGL(k_orientation, dim3(n), dim3(64), 0, s, jobs, out, n, imgs, orimask, atanLut, doHalf, th,
                     maxAngles);
}

}  // namespace mx
