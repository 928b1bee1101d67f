// kernels_views.hip -- view synthesis on device.
//
// Reference: GenerateSynthImageCorr, synth-detection.cpp:236-430: rotate (cv::warpAffine, INTER_LINEAR,
// BORDER_CONSTANT 128), anisotropic anti-alias blur (cv::GaussianBlur, default BORDER_REFLECT_101), tilt/zoom
// scale (cv::warpAffine again).  The OpenCV 2.4.9 arithmetic restated: inverse map in f64, source coordinates
// in 1/1024 fixed point (cvRound = round half to even), rounded to 1/32 px, weights = products of multiples
// of 1/32 (exact in f32), four taps accumulated left to right in f32.
#include "engine.hpp"

namespace mx {

// cv::warpAffine's fixed-point inverse map: adelta / bdelta depend on x only, X0 / Y0 on y only (OpenCV tabulates the
// former per column and forms the latter per row)
MX_D void warp_xterm(const double *M, int x, int &adelta, int &bdelta) {
  const int AB_SCALE = 1024;
  adelta = (int)rint(M[0] * x * AB_SCALE); bdelta = (int)rint(M[3] * x * AB_SCALE);
}
MX_D void warp_yterm(const double *M, int y, int &X0, int &Y0) {
  const int AB_SCALE = 1024;
  X0 = (int)rint((M[1] * y + M[2]) * AB_SCALE) + 16;
  Y0 = (int)rint((M[4] * y + M[5]) * AB_SCALE) + 16;
}
MX_D float warp_fetch(const float *src, int sh, int sw, float cval, int adelta, int bdelta, int X0, int Y0) {
  const int X = (X0 + adelta) >> 5, Y = (Y0 + bdelta) >> 5;
  int sx = X >> 5, sy = Y >> 5;
  sx = sx < -32768 ? -32768 : (sx > 32767 ? 32767 : sx);
  sy = sy < -32768 ? -32768 : (sy > 32767 ? 32767 : sy);
  const float fx = (float)(X & 31) * (1.f / 32), fy = (float)(Y & 31) * (1.f / 32);
  const float w0 = (1.f - fy) * (1.f - fx), w1 = (1.f - fy) * fx, w2 = fy * (1.f - fx), w3 = fy * fx;
  float out;
  if (sx >= 0 && sx < sw - 1 && sy >= 0 && sy < sh - 1) {
    const float *S = src + (size_t)sy * sw + sx;
    out = S[0] * w0 + S[1] * w1 + S[sw] * w2 + S[sw + 1] * w3;
  } else if (sx >= sw || sx + 1 < 0 || sy >= sh || sy + 1 < 0) {
    out = cval;
  } else {
    const bool x0 = sx >= 0 && sx < sw, x1 = sx + 1 >= 0 && sx + 1 < sw, y0 = sy >= 0 && sy < sh, y1 = sy + 1 >= 0 && sy + 1 < sh;
    const float v0 = (x0 && y0) ? src[(size_t)sy * sw + sx] : cval;
    const float v1 = (x1 && y0) ? src[(size_t)sy * sw + sx + 1] : cval;
    const float v2 = (x0 && y1) ? src[(size_t)(sy + 1) * sw + sx] : cval;
    const float v3 = (x1 && y1) ? src[(size_t)(sy + 1) * sw + sx + 1] : cval;
    out = v0 * w0 + v1 * w1 + v2 * w2 + v3 * w3;
  }
  return out;
}
MX_D float warp_value(const float *src, int sh, int sw, const double *M, float cval, int x, int y) {
  int ad, bd, X0, Y0;
  warp_xterm(M, x, ad, bd);
  warp_yterm(M, y, X0, Y0);
  return warp_fetch(src, sh, sw, cval, ad, bd, X0, Y0);
}
MX_D void warp_body(const WarpJob &jb, int x, int y) {
  if (x >= jb.dcols || y >= jb.drows) return;
  jb.dst[(size_t)y * jb.dcols + x] = warp_value(jb.src, jb.srows, jb.scols, jb.M, jb.cval, x, y);
}
__global__ __launch_bounds__(256) void k_warp_affine(WarpJob jb) {
  warp_body(jb, blockIdx.x * 64 + (threadIdx.x & 63), blockIdx.y * 4 + (threadIdx.x >> 6));
}

MX_D int reflect101(int p, int n) {
  if (n == 1) return 0;
  while (p < 0 || p >= n) { if (p < 0) p = -p; else p = 2 * n - 2 - p; }
  return p;
}

// one pass of a separable Gaussian with BORDER_REFLECT_101: pass 0 = rows (RowFilter / SymmRowSmallFilter order),
// pass 1 = columns (SymmColumnFilter order)
MX_D int border_idx(int p, int n, int border) {
  if (border) return p < 0 ? 0 : (p > n - 1 ? n - 1 : p);
  return reflect101(p, n);
}
MX_D void blur_body(const float *src, float *dst, int rows, int cols, const float *taps, int n, int pass, int x, int y, int border = 0) {
  if (x >= cols || y >= rows) return;
  const int R = n >> 1;
  float v;
  if (n == 1) v = src[(size_t)y * cols + x];
  else if (pass == 0) {
    const float *row = src + (size_t)y * cols;
    if (n <= 5) {
      v = row[x] * taps[R];
      for (int j = 1; j <= R; j++) v = v + (row[border_idx(x - j, cols, border)] + row[border_idx(x + j, cols, border)]) * taps[R + j];
    } else {
      v = 0.f;
      for (int j = 0; j < n; j++) v = v + row[border_idx(x + j - R, cols, border)] * taps[j];
    }
  } else {
    v = taps[R] * src[(size_t)y * cols + x] + 0.f;
    for (int j = 1; j <= R; j++)
      v = v + taps[R + j] * (src[(size_t)border_idx(y + j, rows, border) * cols + x] + src[(size_t)border_idx(y - j, rows, border) * cols + x]);
  }
  dst[(size_t)y * cols + x] = v;
}
__global__ __launch_bounds__(256) void k_blur_pass(const float *src, float *dst, int rows, int cols, const float *taps,
                                                   int n, int pass, int border) {
  blur_body(src, dst, rows, cols, taps, n, pass, blockIdx.x * 64 + (threadIdx.x & 63), blockIdx.y * 4 + (threadIdx.x >> 6), border);
}

// ---- detector responses other than the Hessian (ScaleSpaceDetector::Response, pyramid.cpp:132-175) -------------------------
// dogResponse (:176-181): in - blur(in).   computeGradient (helpers.cpp:779-797) + the three products of HarrisResponse
// (:283-305).   harris_combine: dx2 = sigmasq * blur(LxLx) ..., R = (dx2 dy2 - dxdy dxdy) - (0.04f (dx2 + dy2)) (dx2 + dy2).
__global__ __launch_bounds__(256) void k_sub(const float *a, const float *b, float *o, size_t n) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) o[i] = a[i] - b[i];
}
__global__ __launch_bounds__(256) void k_grad_products(const float *img, int rows, int cols, float *xx, float *yy, float *xy) {
  const int c = blockIdx.x * 64 + (threadIdx.x & 63), r = blockIdx.y * 4 + (threadIdx.x >> 6);
  if (c >= cols || r >= rows) return;
  const float *R = img + (size_t)r * cols;
  float xg, yg;
  if (cols == 1) xg = 0.f;
  else if (c == 0) xg = R[c + 1] - R[c];
  else if (c == cols - 1) xg = R[c] - R[c - 1];
  else xg = R[c + 1] - R[c - 1];
  if (rows == 1) yg = 0.f;
  else if (r == 0) yg = R[cols + c] - R[c];
  else if (r == rows - 1) yg = R[c] - R[c - cols];
  else yg = R[cols + c] - R[c - cols];
  const size_t i = (size_t)r * cols + c;
  xx[i] = xg * xg; yy[i] = yg * yg; xy[i] = xg * yg;
}
__global__ __launch_bounds__(256) void k_harris_combine(const float *bxx, const float *byy, const float *bxy, float sigmasq, float *o, size_t n) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const float dx2 = bxx[i] * sigmasq, dy2 = byy[i] * sigmasq, dxdy = bxy[i] * sigmasq;
  const float s = dx2 + dy2;
  const float a = dx2 * dy2, b = dxdy * dxdy, c = (0.04f * s) * s;
  o[i] = (a - b) - c;
}

// ---- batched form: all views of a launch set in four launches (warp, blur rows, blur columns, warp) ---------------------------
// A view of tilt t is 1/t of the image, 31 to 61 of them per image: launched one by one they are ~250 launches of 2-10 us
// per pair.  Every block takes one 64 x 4 tile of one view; the view is found from the tile prefix carried by the jobs.
MX_D int find_view(const ViewJob *jobs, int n, int tile, int stage) {
  int j = 0;
  while (j + 1 < n && tile >= (stage == 2 ? jobs[j + 1].tileF : (stage ? jobs[j + 1].tileB : jobs[j + 1].tileA))) j++;
  return j;
}
__global__ __launch_bounds__(256) void k_views_warp(const ViewJob *jobs, int n, int stage) {
  const int j = find_view(jobs, n, blockIdx.x, stage);
  const ViewJob &v = jobs[j];
  WarpJob jb;
  if (stage == 0) { jb.src = v.src; jb.dst = v.rot; jb.srows = v.srows; jb.scols = v.scols; jb.drows = v.rrows; jb.dcols = v.rcols; }
  else { jb.src = v.rot; jb.dst = v.dst; jb.srows = v.rrows; jb.scols = v.rcols; jb.drows = v.drows; jb.dcols = v.dcols; }
#pragma unroll
  for (int i = 0; i < 6; i++) jb.M[i] = stage ? v.W[i] : v.R[i];
  jb.cval = 128.f;
  const int t = blockIdx.x - (stage ? v.tileB : v.tileA), tx = (jb.dcols + 63) / 64;
  warp_body(jb, (t % tx) * 64 + (threadIdx.x & 63), (t / tx) * 4 + (threadIdx.x >> 6));
}
__global__ __launch_bounds__(256) void k_views_blur(const ViewJob *jobs, int n, const float *taps, int pass) {
  const int j = find_view(jobs, n, blockIdx.x, 0);
  const ViewJob &v = jobs[j];
  if (!v.doBlur) return;
  const int t = blockIdx.x - v.tileA, tx = (v.rcols + 63) / 64;
  const float *src = pass ? v.tmp : v.rot;
  float *dst = pass ? v.rot : v.tmp;
  blur_body(src, dst, v.rrows, v.rcols, taps + v.tapOfs + (pass ? v.kx : 0), pass ? v.ky : v.kx, pass, (t % tx) * 64 + (threadIdx.x & 63),
            (t / tx) * 4 + (threadIdx.x >> 6));
}

// the two filter passes of k_views_rotblur over its LDS tiles; N = compile-time tap count (0: runtime nx / ny)
template <int N>
MX_D void vf_rows(const float *Wt, float *Tt, const float *kxT, int WW, int Rx, int lane, int ty, int xOut, int yEnd, int nxRt) {
  constexpr int NK = N ? N : 1;
  float k[NK];
#pragma unroll
  for (int q = 0; q < NK; q++) k[q] = N ? kxT[q] : 0.f;
  for (int ly = ty; ly < yEnd; ly += 4) {
    const float *row = Wt + ly * WW + Rx;
    for (int lx = lane; lx < xOut; lx += 64) {
      float r;
      if (N == 1) r = row[lx];
      else if (N == 3 || N == 5) {
        constexpr int R = N >> 1;
        r = row[lx] * k[R];
#pragma unroll
        for (int q = 1; q <= R; q++) r = r + (row[lx - q] + row[lx + q]) * k[R + q];
      } else if (N > 5) {
        constexpr int R = N >> 1;
        r = 0.f;
#pragma unroll
        for (int q = 0; q < N; q++) r = r + row[lx + q - R] * k[q];
      } else {
        const int nx = nxRt;
        if (nx <= 5) {
          r = row[lx] * kxT[Rx];
          for (int q = 1; q <= Rx; q++) r = r + (row[lx - q] + row[lx + q]) * kxT[Rx + q];
        } else {
          r = 0.f;
          for (int q = 0; q < nx; q++) r = r + row[lx + q - Rx] * kxT[q];
        }
      }
      Tt[ly * VF_TW + lx] = r;
    }
  }
}
template <int N>
MX_D void vf_cols(const float *Tt, float *outBase, const float *kyT, int rcols, int Ry, int lane, int ty, int xOut, int yOut, int nyRt) {
  constexpr int NK = N ? N : 1;
  float k[NK];
#pragma unroll
  for (int q = 0; q < NK; q++) k[q] = N ? kyT[q] : 0.f;
  for (int ly = ty; ly < yOut; ly += 4) {
    const float *col = Tt + (ly + Ry) * VF_TW;
    float *out = outBase + (size_t)ly * rcols;
    for (int lx = lane; lx < xOut; lx += 64) {
      float r;
      if (N == 1) r = col[lx];
      else if (N > 1) {
        constexpr int R = N >> 1;
        r = k[R] * col[lx] + 0.f;
#pragma unroll
        for (int q = 1; q <= R; q++) r = r + k[R + q] * (col[lx + q * VF_TW] + col[lx - q * VF_TW]);
      } else {
        (void)nyRt;
        r = kyT[Ry] * col[lx] + 0.f;
        for (int q = 1; q <= Ry; q++) r = r + kyT[Ry + q] * (col[lx + q * VF_TW] + col[lx - q * VF_TW]);
      }
      out[lx] = r;
    }
  }
}

// Rotate + anti-alias blur of a view in ONE launch.  The three-launch form writes the rotated image, reads it for the row
// filter, writes the row-filtered image, reads it for the column filter and writes the result over the rotated image: four
// passes over 1-2 Mpx per view, 16-31 views per image, in kernels that wait on memory.  Here a workgroup owns a
// VF_TW x VF_TH tile of the blurred rotated image: it evaluates cv::warpAffine at the tile's pixels plus the halo the two
// filters reach (BORDER_REFLECT_101 of the ROTATED image: a halo pixel outside it is the rotation evaluated at the mirrored
// coordinate, the value the separate launches read back), filters the rows of all TH + 2 Ry lines in LDS, then the columns,
// and stores the tile.  Every value is the same f32 the separate launches pass through memory; sums, terms and order are
// those of blur_body.
__global__ __launch_bounds__(256) void k_views_rotblur(const ViewJob *jobs, int n, const float *taps, int wFloats) {
  extern __shared__ float smem[];
  const int j = find_view(jobs, n, blockIdx.x, 2);
  const ViewJob &v = jobs[j];
  const int Rx = v.kx >> 1, Ry = v.ky >> 1;
  const int t = blockIdx.x - v.tileF, tx = (v.rcols + VF_TW - 1) / VF_TW;
  const int x0 = (t % tx) * VF_TW, y0 = (t / tx) * VF_TH;
  const int WW = VF_TW + 2 * Rx;
  float *const Wt = smem, *const Tt = smem + wFloats;
  const int lane = threadIdx.x & 63, ty = threadIdx.x >> 6;
  const int xEnd = min(VF_TW, v.rcols - x0) + 2 * Rx, yEnd = min(VF_TH, v.rrows - y0) + 2 * Ry;   // what the tile's outputs reach
  {
    constexpr int NX = (VF_TW + 2 * VF_RX + 63) / 64;   // columns of the haloed tile per lane
    int ad[NX], bd[NX];
#pragma unroll
    for (int u = 0; u < NX; u++) warp_xterm(v.R, reflect101(x0 - Rx + lane + 64 * u, v.rcols), ad[u], bd[u]);
    for (int ly = ty; ly < yEnd; ly += 4) {
      int X0, Y0;
      warp_yterm(v.R, reflect101(y0 - Ry + ly, v.rrows), X0, Y0);
#pragma unroll
      for (int u = 0; u < NX; u++) {
        const int lx = lane + 64 * u;
        if (lx < xEnd) Wt[ly * WW + lx] = warp_fetch(v.src, v.srows, v.scols, 128.f, ad[u], bd[u], X0, Y0);
      }
    }
  }
  __syncthreads();
  const float *kxT = taps + v.tapOfs, *kyT = kxT + v.kx;
  const int nx = v.kx, ny = v.ky, xOut = xEnd - 2 * Rx, yOut = yEnd - 2 * Ry;
  // The filter sizes of a view are wave-uniform and small (3 .. 13 taps for every default view): one switch per pass picks a
  // fully unrolled body with the taps in registers; other sizes take the runtime loop.  Terms and order are blur_body's.
  switch (nx) {
    case 1: vf_rows<1>(Wt, Tt, kxT, WW, Rx, lane, ty, xOut, yEnd, nx); break;
    case 3: vf_rows<3>(Wt, Tt, kxT, WW, Rx, lane, ty, xOut, yEnd, nx); break;
    case 5: vf_rows<5>(Wt, Tt, kxT, WW, Rx, lane, ty, xOut, yEnd, nx); break;
    case 7: vf_rows<7>(Wt, Tt, kxT, WW, Rx, lane, ty, xOut, yEnd, nx); break;
    case 9: vf_rows<9>(Wt, Tt, kxT, WW, Rx, lane, ty, xOut, yEnd, nx); break;
    case 11: vf_rows<11>(Wt, Tt, kxT, WW, Rx, lane, ty, xOut, yEnd, nx); break;
    case 13: vf_rows<13>(Wt, Tt, kxT, WW, Rx, lane, ty, xOut, yEnd, nx); break;
    default: vf_rows<0>(Wt, Tt, kxT, WW, Rx, lane, ty, xOut, yEnd, nx); break;
  }
  __syncthreads();
  float *const outBase = v.rot + (size_t)y0 * v.rcols + x0;
  switch (ny) {
    case 1: vf_cols<1>(Tt, outBase, kyT, v.rcols, Ry, lane, ty, xOut, yOut, ny); break;
    case 3: vf_cols<3>(Tt, outBase, kyT, v.rcols, Ry, lane, ty, xOut, yOut, ny); break;
    case 5: vf_cols<5>(Tt, outBase, kyT, v.rcols, Ry, lane, ty, xOut, yOut, ny); break;
    default: vf_cols<0>(Tt, outBase, kyT, v.rcols, Ry, lane, ty, xOut, yOut, ny); break;
  }
}

void launch_warp_affine(hipStream_t s, const WarpJob &jb) {
  dim3 grid((jb.dcols + 63) / 64, (jb.drows + 3) / 4);
  hipLaunchKernelGGL(k_warp_affine, grid, dim3(256), 0, s, jb);
}
void launch_blur_pass(hipStream_t s, const float *src, float *dst, int rows, int cols, const float *taps, int n, int pass, int border) {
  dim3 grid((cols + 63) / 64, (rows + 3) / 4);
  hipLaunchKernelGGL(k_blur_pass, grid, dim3(256), 0, s, src, dst, rows, cols, taps, n, pass, border);
}
void launch_sub(hipStream_t s, const float *a, const float *b, float *o, size_t n) {
  hipLaunchKernelGGL(k_sub, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, a, b, o, n);
}
void launch_grad_products(hipStream_t s, const float *img, int rows, int cols, float *xx, float *yy, float *xy) {
  hipLaunchKernelGGL(k_grad_products, dim3((cols + 63) / 64, (rows + 3) / 4), dim3(256), 0, s, img, rows, cols, xx, yy, xy);
}
void launch_harris_combine(hipStream_t s, const float *bxx, const float *byy, const float *bxy, float sigmasq, float *o, size_t n) {
  hipLaunchKernelGGL(k_harris_combine, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, bxx, byy, bxy, sigmasq, o, n);
}
void launch_views_warp(hipStream_t s, const ViewJob *jobs, int n, int tiles, int stage) {
  if (tiles > 0) MX_DUP(K_WARP) hipLaunchKernelGGL(k_views_warp, dim3(tiles), dim3(256), 0, s, jobs, n, stage);
}
void launch_views_blur(hipStream_t s, const ViewJob *jobs, int n, int tiles, const float *taps, int pass) {
  if (tiles > 0) hipLaunchKernelGGL(k_views_blur, dim3(tiles), dim3(256), 0, s, jobs, n, taps, pass);
}
void launch_views_rotblur(hipStream_t s, const ViewJob *jobs, int n, int tiles, const float *taps, int maxRx, int maxRy) {
  if (tiles <= 0) return;
  const int wFloats = (VF_TH + 2 * maxRy) * (VF_TW + 2 * maxRx), tFloats = (VF_TH + 2 * maxRy) * VF_TW;
  MX_DUP(K_VIEW_BLUR) hipLaunchKernelGGL(k_views_rotblur, dim3(tiles), dim3(256), (size_t)(wFloats + tFloats) * 4, s, jobs, n, taps, wFloats);
}

}  // namespace mx
