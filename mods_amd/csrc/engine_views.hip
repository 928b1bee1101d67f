// engine_views.hip -- on-demand view synthesis and the per-view loop (host orchestration).
//
//   SetVSPars                      synth-detection.cpp:103-234   (view ladder of one step)
//   GenerateSynthImageCorr         synth-detection.cpp:236-430   (rotate, anti-alias blur, tilt/zoom)
//   SynthDetectDescribeKeypoints   imagerepresentation.cpp:603-2047, HessianAffine branch for one SIFT-family
//                                  descriptor: per view detect -> orient -> reproject -> describe, then
//                                  AddRegions in view order with id re-basing (:588-600, :2044-2045)
// Views are independent until that concatenation, so `view_begin/view_step` let a caller take every G-th view
// (the shard of one GPU); the caller gathers the per-view blocks afterwards.
#include <math.h>
#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <thread>
#include "engine_api.hpp"

namespace mx {

int set_vs_pars(const double *scale_set, int ns, const double *tilt_set, int nt, double phi_base, double InitSigma,
                int doBlur, modsx_view *par, int cap, modsx_view *prev, int *nprev, int cap_prev) {
  const double eps1 = 0.01;
  std::vector<modsx_view> tmp;
  auto mk = [&](double phi, double tilt, double zoom, int blur) {
    modsx_view v; v.phi = phi; v.tilt = tilt; v.zoom = zoom; v.InitSigma = InitSigma; v.doBlur = blur;
    return v;
  };
  if (ns == 0 || nt == 0) tmp.push_back(mk(0, 0, 0, 0));
  for (int sc = 0; sc < ns; sc++)
    for (int t = 0; t < nt; t++) {
      if (fabs(tilt_set[t] - 1) > eps1) {
        int n_rot1 = floor(180.0 * tilt_set[t] / phi_base);
        double delta_phi = M_PI / n_rot1;
        if (n_rot1 < 0) {  // "no rotation" mode: one vertical tilt
          n_rot1 = 1; delta_phi = 0;
          tmp.push_back(mk(0, -tilt_set[t], scale_set[sc], doBlur));
        }
        for (int r = 0; r < n_rot1; r++) tmp.push_back(mk(delta_phi * r, tilt_set[t], scale_set[sc], doBlur));
      } else tmp.push_back(mk(0, tilt_set[t], scale_set[sc], doBlur));
    }
  int n = 0;
  std::vector<modsx_view> added;
  for (size_t i = 0; i < tmp.size(); i++) {
    bool uniq = true;
    for (int j = 0; j < *nprev; j++)
      if ((fabs(tmp[i].zoom - prev[j].zoom) <= eps1) && (fabs(tmp[i].tilt - prev[j].tilt) <= eps1) &&
          (fabs(tmp[i].phi - prev[j].phi) <= eps1)) { uniq = false; break; }
    if (uniq) { if (n < cap) par[n] = tmp[i]; n++; added.push_back(tmp[i]); }
  }
  for (const modsx_view &v : added) if (*nprev < cap_prev) prev[(*nprev)++] = v;
  return n;
}

// cv::warpAffine inverts the forward 2x3 matrix in f64 before the pixel loop (imgwarp.cpp)
static void invert_affine(const double *Min, double *M) {
  for (int i = 0; i < 6; i++) M[i] = Min[i];
  double D = M[0] * M[4] - M[1] * M[3];
  D = D != 0 ? 1. / D : 0;
  double A11 = M[4] * D, A22 = M[0] * D;
  M[0] = A11; M[1] *= -D;
  M[3] *= -D; M[4] = A22;
  double b1 = -M[0] * M[2] - M[1] * M[5];
  double b2 = -M[3] * M[2] - M[4] * M[5];
  M[2] = b1; M[5] = b2;
}

// host half of GenerateSynthImageCorr (synth-detection.cpp:236-430) for one view: homography, sizes, the inverse maps of the
// two warps and the anti-alias taps
struct ViewPlan {
  int identity = 0;
  double H[9];
  int w_rot = 0, h_rot = 0, ow = 0, oh = 0, doBlur = 0, kx = 1, ky = 1;
  double Rinv[6], Winv[6];
  std::vector<float> taps;   // kx taps, then ky taps
};

static int plan_view(const modsx_image *gray, const modsx_view &v, ViewPlan &P) {
  double tilt = v.tilt;
  const double phi = v.phi, zoom = v.zoom, InitSigma = v.InitSigma;
  int zoomed = 0;
  bool vertical_tilt = false;
  if (tilt < 0) { tilt = -tilt; vertical_tilt = true; }
  if (fabs(zoom - 1.0f) >= 0.05) zoomed = 1;
  const int w = gray->cols, h = gray->rows;
  int wS1 = (int)(w * zoom), hS1 = (int)(h * zoom);
  double *H = P.H;
  if ((fabs(tilt - 1.) <= 0.1) && (fabs(phi) <= 0.2) && (fabs(zoom - 1.) <= 0.1)) {  // original image, :278-289
    const double E[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    for (int i = 0; i < 9; i++) H[i] = E[i];
    P.identity = 1;
    P.ow = w; P.oh = h;
    return MODSX_OK;
  }
  P.identity = 0;
  double d, d2, w_new, h_new;
  double kV = 1., kH = 1.;
  if (zoomed) { kV = (double)w / (double)wS1; kH = (double)h / (double)hS1; }
  const bool q1 = (phi >= 0) && (phi < M_PI / 2);
  if (vertical_tilt) {
    if (q1) {
      w_new = floor((0.5 + cos(phi) * w + sin(phi) * h) / (kH));
      h_new = floor((0.5 + sin(phi) * w + cos(phi) * h) / (tilt * kV));
      H[0] = cos(phi) / kH; H[1] = sin(phi) / kH; H[2] = 0;
      H[3] = -sin(phi) / (tilt * kV); H[4] = cos(phi) / (tilt * kV); H[5] = floor(0.5 + sin(phi) * w / (tilt * kV));
    } else {
      w_new = floor((0.5 - cos(phi) * w + sin(phi) * h) / (kH));
      h_new = floor((0.5 + sin(phi) * w - cos(phi) * h) / (tilt * kV));
      d = -floor(cos(phi) * w / kH);
      d2 = floor(0.5 + (sin(phi) * w - cos(phi) * h) / (tilt * kV));
      H[0] = cos(phi) / kH; H[1] = sin(phi) / kH; H[2] = d;
      H[3] = -sin(phi) / (tilt * kV); H[4] = cos(phi) / (tilt * kV); H[5] = d2;
    }
  } else {
    if (q1) {
      w_new = floor((0.5 + cos(phi) * w + sin(phi) * h) / (tilt * kH));
      h_new = floor((0.5 + sin(phi) * w + cos(phi) * h) / (kV));
      H[0] = cos(phi) / (tilt * kH); H[1] = sin(phi) / (tilt * kH); H[2] = 0;
      H[3] = -sin(phi) / kV; H[4] = cos(phi) / kV; H[5] = floor(0.5 + sin(phi) * w / kV);
    } else {
      w_new = floor((0.5 - cos(phi) * w + sin(phi) * h) / (tilt * kH));
      h_new = floor((0.5 + sin(phi) * w - cos(phi) * h) / (kV));
      d = -floor(cos(phi) * w / (tilt * kH));
      d2 = floor(0.5 + (sin(phi) * w - cos(phi) * h) / kV);
      H[0] = cos(phi) / (tilt * kH); H[1] = sin(phi) / (tilt * kH); H[2] = d;
      H[3] = -sin(phi) / kV; H[4] = cos(phi) / kV; H[5] = d2;
    }
  }
  H[6] = 0; H[7] = 0; H[8] = 1;
  double sigma_aa_2 = zoomed ? InitSigma / (4.0 * zoom) : InitSigma / 2.0;
  double sigma_aa = InitSigma * tilt / (2.0 * zoom);
  double sigma_x = vertical_tilt ? sigma_aa_2 : sigma_aa, sigma_y = vertical_tilt ? sigma_aa : sigma_aa_2;
  double R[6];
  if (q1) {
    P.w_rot = floor((0.5 + cos(phi) * w + sin(phi) * h));
    P.h_rot = floor((0.5 + sin(phi) * w + cos(phi) * h));
    R[0] = cos(phi); R[1] = sin(phi); R[2] = 0;
    R[3] = -sin(phi); R[4] = cos(phi); R[5] = floor(0.5 + sin(phi) * w);
  } else {
    P.w_rot = floor((0.5 - cos(phi) * w + sin(phi) * h));
    P.h_rot = floor((0.5 + sin(phi) * w - cos(phi) * h));
    d = -floor(cos(phi) * w);
    d2 = floor(0.5 + (sin(phi) * w - cos(phi) * h));
    R[0] = cos(phi); R[1] = sin(phi); R[2] = d;
    R[3] = -sin(phi); R[4] = cos(phi); R[5] = d2;
  }
  P.ow = (int)w_new; P.oh = (int)h_new;
  if (P.w_rot <= 0 || P.h_rot <= 0 || P.ow <= 0 || P.oh <= 0) { set_error("degenerate synthesised view"); return MODSX_ERR_ARG; }
  invert_affine(R, P.Rinv);
  P.doBlur = v.doBlur;
  P.taps.clear();
  if (v.doBlur) {
    int kx = floor(2.0 * 3.0 * sigma_x + 1.0);
    if (kx % 2 == 0) kx++;
    if (kx < 3) kx = 3;
    int ky = floor(2.0 * 3.0 * sigma_y + 1.0);
    if (ky % 2 == 0) ky++;
    if (ky < 3) ky = 3;
    if (P.h_rot == 1) ky = 1;
    if (P.w_rot == 1) kx = 1;
    std::vector<float> KX = gaussian_kernel(kx, std::max(sigma_x, 0.));
    std::vector<float> KY = (ky == kx && fabs(sigma_x - sigma_y) < 2.220446049250313e-16) ? KX : gaussian_kernel(ky, std::max(sigma_y, 0.));
    P.kx = kx; P.ky = ky;
    P.taps = KX;
    P.taps.insert(P.taps.end(), KY.begin(), KY.end());
  }
  double Wz[6] = {0, 0, 0, 0, 0, 0};
  if (vertical_tilt) { Wz[0] = 1.0 / kH; Wz[4] = 1.0 / (tilt * kV); }
  else { Wz[0] = 1.0 / (tilt * kH); Wz[4] = 1.0 / kV; }
  invert_affine(Wz, P.Winv);
  return MODSX_OK;
}

// Synthesises n views of `gray` in four launches (rotate all, blur rows all, blur columns all, tilt/zoom all).  Outputs go to
// dst[i] (device, P[i].ow x P[i].oh floats; ignored for identity views).  No host wait: the job table and the taps travel
// through pinned staging that is only rewritten after the caller's next synchronisation of the stream.
static int synth_views_batch(modsx_ctx *c, const modsx_image *const *grays, const ViewPlan *P, float *const *dst, int n) {
  hipStream_t s = c->stream;
  std::vector<ViewJob> jobs;
  std::vector<float> taps;
  size_t rotFloats = 0;
  int tilesA = 0, tilesB = 0, tilesF = 0, maxRx = 0, maxRy = 0;
  double wpx = 0, rpx = 0;
  for (int i = 0; i < n; i++) {
    if (P[i].identity) continue;
    ViewJob j;
    memset(&j, 0, sizeof j);
    const modsx_image *gray = grays[i];
    j.src = gray->d; j.srows = gray->rows; j.scols = gray->cols;
    j.rrows = P[i].h_rot; j.rcols = P[i].w_rot; j.drows = P[i].oh; j.dcols = P[i].ow;
    j.dst = dst[i];
    j.doBlur = P[i].doBlur; j.kx = P[i].kx; j.ky = P[i].ky; j.tapOfs = (int)taps.size();
    taps.insert(taps.end(), P[i].taps.begin(), P[i].taps.end());
    for (int q = 0; q < 6; q++) { j.R[q] = P[i].Rinv[q]; j.W[q] = P[i].Winv[q]; }
    j.tileA = tilesA; j.tileB = tilesB; j.tileF = tilesF;
    // rotate + blur in one launch when the halo of the two filters fits the fused kernel's tile (every default view does)
    j.fused = j.doBlur && (j.kx >> 1) <= VF_RX && (j.ky >> 1) <= VF_RY;
    if (j.fused) {
      tilesF += ((j.rcols + VF_TW - 1) / VF_TW) * ((j.rrows + VF_TH - 1) / VF_TH);
      maxRx = std::max(maxRx, j.kx >> 1); maxRy = std::max(maxRy, j.ky >> 1);
    } else
      tilesA += ((j.rcols + 63) / 64) * ((j.rrows + 3) / 4);
    tilesB += ((j.dcols + 63) / 64) * ((j.drows + 3) / 4);
    j.rot = (float *)(uintptr_t)rotFloats;            // offsets first; the bases are added once the buffers exist
    rotFloats += (size_t)j.rrows * j.rcols;
    wpx += (double)gray->rows * gray->cols + 2.0 * j.rrows * j.rcols + (double)j.drows * j.dcols;
    rpx += (double)j.rrows * j.rcols;
    jobs.push_back(j);
  }
  if (jobs.empty()) return MODSX_OK;
  const size_t jobB = jobs.size() * sizeof(ViewJob), tapB = std::max<size_t>(1, taps.size()) * 4;
  if (!c->viewTmp[0].ensure(rotFloats * 4) || !c->viewTmp[1].ensure(rotFloats * 4) || !c->viewJobs.ensure(jobB + tapB + 64) ||
      !c->hViewJobs.ensure(jobB + tapB + 64))
    return MODSX_ERR_NOMEM;
  for (ViewJob &j : jobs) {
    const size_t o = (size_t)(uintptr_t)j.rot;
    j.rot = (float *)c->viewTmp[0].p + o;
    j.tmp = (float *)c->viewTmp[1].p + o;
  }
  char *hb = (char *)c->hViewJobs.p;
  memcpy(hb, jobs.data(), jobB);
  if (!taps.empty()) memcpy(hb + jobB, taps.data(), taps.size() * 4);
  MX_HIP(ctx_copy(c, c->viewJobs.p, hb, jobB + tapB, hipMemcpyHostToDevice));
  const ViewJob *dj = (const ViewJob *)c->viewJobs.p;
  const float *dt = (const float *)((char *)c->viewJobs.p + jobB);
  size_t pslot;
  prof_begin(c, K_WARP, wpx * 4, &pslot);
  launch_views_warp(s, dj, (int)jobs.size(), tilesA, 0);
  prof_end(c, pslot);
  prof_begin(c, K_VIEW_BLUR, rpx * 16, &pslot);
  launch_views_rotblur(s, dj, (int)jobs.size(), tilesF, dt, maxRx, maxRy);
  launch_views_blur(s, dj, (int)jobs.size(), tilesA, dt, 0);
  launch_views_blur(s, dj, (int)jobs.size(), tilesA, dt, 1);
  prof_end(c, pslot);
  prof_begin(c, K_WARP, 0, &pslot);
  launch_views_warp(s, dj, (int)jobs.size(), tilesB, 1);
  prof_end(c, pslot);
  MX_HIP(hipGetLastError());
  return MODSX_OK;
}

// the public single-view form (modsx_synth_view): a fresh allocation owned by the returned image
int synth_view(modsx_ctx *c, const modsx_image *gray, const modsx_view &v, modsx_image **out, double *H, int *identity, int) {
  *out = nullptr;
  ViewPlan P;
  int rc = plan_view(gray, v, P);
  if (rc) return rc;
  for (int i = 0; i < 9; i++) H[i] = P.H[i];
  *identity = P.identity;
  modsx_image *im = new modsx_image();
  im->rows = P.oh; im->cols = P.ow;
  if (P.identity) { im->d = gray->d; im->owned = false; *out = im; return MODSX_OK; }
  im->owned = true; im->d = nullptr;
  if (hipMalloc(&im->d, (size_t)P.ow * P.oh * 4) != hipSuccess) { delete im; set_error("hipMalloc view"); return MODSX_ERR_NOMEM; }
  float *dst[1] = {im->d};
  rc = synth_views_batch(c, &gray, &P, dst, 1);
  if (!rc && ctx_sync(c) != hipSuccess) { set_error("view synthesis failed"); rc = MODSX_ERR_DEVICE; }
  if (rc) { hipFree(im->d); delete im; return rc; }
  *out = im;
  return MODSX_OK;
}

// The per-view loop over ITEMS = (source image, view) pairs, any mix of images in one launch set (the view-sharded path
// batches the views a rank owns of several images / pairs: at world 8 a rank holds ~4 views per image, and launch sets of 4
// views leave the device mostly idle).  Output regions carry img_id = view index (0 for the identity view) and ids local to
// their item's block; `itemCounts[k]` gets the number of described regions of item k.  Descriptors are written item block
// after item block at devU8/devF (device, capacity devCapRegions) and optionally copied to hostDesc.
int detect_describe_items(modsx_ctx *c, const modsx_image *const *itemImg, const int *itemView, int nitems, const modsx_view *views,
                          const modsx_pair_params &pp, std::vector<modsx_region> &regs,
                          float *devF, uint8_t *devU8, size_t devCapRegions, float *hostDesc, int *itemCounts,
                          const DescSet *dsIn, uint8_t *const *devU8x) {
  CtxBusy busy(c);
  regs.clear();
  // the step's descriptor classes: class 0 goes to devF / devU8 / hostDesc, class k >= 1 to devU8x[k - 1] (same capacity);
  // without devU8x only class 0 is produced -- the orientation mode still follows the whole list
  DescSet ds;
  if (dsIn) ds = *dsIn;
  else { const int rd = resolve_descs(pp, nullptr, ds); if (rd) return rd; }
  const int oriHalf = ds.half() ? 1 : 0;
  if (!devU8x) { ds.forceHalf = oriHalf != 0; ds.n = 1; }
  if (itemCounts) for (int k = 0; k < nitems; k++) itemCounts[k] = 0;
  struct SetScope { SetScope() { host_set_enter(); } ~SetScope() { host_set_leave(); } } setScope;
  std::vector<int> take(nitems);
  for (int k = 0; k < nitems; k++) take[k] = k;
  size_t total = 0;
  for (size_t g0 = 0; g0 < take.size(); g0 += MAXB) {
    const int n = (int)std::min<size_t>(MAXB, take.size() - g0);
    modsx_image *vimg[MAXB];
    const modsx_image *cimg[MAXB];
    double Hs[MAXB][9], tilts[MAXB], zooms[MAXB];
    int ident[MAXB];
    int rc = MODSX_OK;
    for (int i = 0; i < n; i++) vimg[i] = nullptr;
    ViewPlan plans[MAXB];
    float *vdst[MAXB];
    const modsx_image *gimg[MAXB];
    for (int i = 0; i < n && !rc; i++) {
      const modsx_image *gray = itemImg[take[g0 + i]];
      gimg[i] = gray;
      const modsx_view &v = views[itemView[take[g0 + i]]];
      rc = plan_view(gray, v, plans[i]);
      if (rc) break;
      for (int q = 0; q < 9; q++) Hs[i][q] = plans[i].H[q];
      ident[i] = plans[i].identity;
      modsx_image *im = new modsx_image();
      im->rows = plans[i].oh; im->cols = plans[i].ow; im->owned = false;
      if (ident[i]) im->d = gray->d;
      else if (!c->viewImg[i].ensure((size_t)im->rows * im->cols * 4)) { delete im; rc = MODSX_ERR_NOMEM; break; }
      else im->d = (float *)c->viewImg[i].p;
      vdst[i] = im->d;
      vimg[i] = im;
      cimg[i] = im;
      tilts[i] = ident[i] ? 1.0 : fabs(v.tilt);   // SynthImage::tilt / zoom as GenerateSynthImageCorr leaves them
      zooms[i] = ident[i] ? 1.0 : v.zoom;
    }
    auto tnow = []() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const bool tim = getenv("MODSX_HOST_TIMING") != nullptr;
    const bool tim2 = tim && atoi(getenv("MODSX_HOST_TIMING")) >= 2;
    double tq = 0;
    double t0 = tnow();
    if (!rc) rc = synth_views_batch(c, gimg, plans, vdst, n);
    if (tim) hipStreamSynchronize(c->stream);
    double t1 = tnow();
    std::vector<modsx_keypoint> kps[MAXB];
    std::vector<modsx_region> r0[MAXB], ro[MAXB];
    if (!rc) {
      if (pp.detector == MODSX_DET_MSER) {
        // DetectAffineRegions(temp_img1, temp_kp1, det_par.MSERParam, DET_MSER, DetectMSERs), imagerepresentation.cpp:1037:
        // u8 truncation of every view on the device, one D2H of bytes, the component trees on host threads' time
        // all views of the set in one buffer, ONE download; the 2 n (view, polarity) trees then run on the host pool
        // ... and the bin sort of every view (the pixel offsets per grey level in raster order that the component tree walks) on the
        // device too: bytes + sorted offsets + level starts in ONE block, one download (MODSX_MSER_DEVICE_SORT=0: the host sorts)
        static const bool devSort = !(getenv("MODSX_MSER_DEVICE_SORT") && !atoi(getenv("MODSX_MSER_DEVICE_SORT")));
        size_t ofs[MAXB + 1], tot = 0, ordOfs[MAXB], npxAll = 0;
        int vr[MAXB], vc[MAXB];
        for (int i = 0; i < n; i++) {
          ofs[i] = tot; tot += ((size_t)cimg[i]->rows * cimg[i]->cols + 63) & ~(size_t)63;
          vr[i] = cimg[i]->rows; vc[i] = cimg[i]->cols;
          ordOfs[i] = npxAll; npxAll += (size_t)vr[i] * vc[i];
        }
        // layout of the block: [u8 views | start: n x 257 ints | order: npxAll ints]; blockHist behind it on the device only
        const size_t oStart = (tot + 255) & ~(size_t)255, oOrder = oStart + (((size_t)n * 257 * 4 + 255) & ~(size_t)255);
        const size_t blockB = devSort ? oOrder + npxAll * 4 : tot;
        const size_t histB = devSort ? mser_sort_blocks(vr, n) * 256 * 4 : 0;
        if (!c->misc.ensure(((blockB + 255) & ~(size_t)255) + histB + 64) || !c->hMser.ensure(blockB + 64)) rc = MODSX_ERR_NOMEM;
        for (int i = 0; i < n && !rc; i++)
          launch_trunc_u8(c->stream, cimg[i]->d, (uint8_t *)c->misc.p + ofs[i], (size_t)vr[i] * vc[i]);
        if (!rc && devSort) {
          const uint8_t *du8[MAXB];
          for (int i = 0; i < n; i++) du8[i] = (const uint8_t *)c->misc.p + ofs[i];
          char *base = (char *)c->misc.p;
          launch_mser_sort(c->stream, du8, vr, vc, ordOfs, n, (int *)(base + ((blockB + 255) & ~(size_t)255)), (int *)(base + oStart), (int *)(base + oOrder));
        }
        if (!rc && (ctx_copy(c, c->hMser.p, c->misc.p, blockB, hipMemcpyDeviceToHost) != hipSuccess ||
                    ctx_sync(c) != hipSuccess)) { set_error("MSER view download failed"); rc = MODSX_ERR_DEVICE; }
        const double tdl = tnow();
        if (!rc) {
          const uint8_t *src[MAXB];
          const int *ho[MAXB], *hs[MAXB];
          for (int i = 0; i < n; i++) {
            src[i] = (const uint8_t *)c->hMser.p + ofs[i];
            ho[i] = (const int *)((const char *)c->hMser.p + oOrder) + ordOfs[i];
            hs[i] = (const int *)((const char *)c->hMser.p + oStart) + (size_t)i * 257;
          }
          rc = detect_msers_views(src, vr, vc, n, pp.mser, tilts, zooms, kps, devSort ? ho : nullptr, devSort ? hs : nullptr);
        }
        if (tim) fprintf(stderr, "  mser set of %d views: u8 + download %.2f ms, component trees %.2f ms\n", n, tdl - t1, tnow() - tdl);
      } else rc = detect_keypoints_batch(c, cimg, n, pp.det, tilts, zooms, kps);
    }
    double t2 = tnow(), t3 = t2, t4 = t2;
    if (!rc) {
      const int detType = pp.detector == MODSX_DET_MSER ? MODSX_DET_MSER : MODSX_DET_HESSIAN;
      host_parallel_light(n, [&](int i) {
        r0[i].resize(kps[i].size());
        detect_affine_regions(kps[i].data(), (int)kps[i].size(), ident[i] ? 0 : itemView[take[g0 + i]], detType, r0[i].data());
      });
      // DetectOrientation(..., HalfSIFT_like_desc, ...): one oriented list for every descriptor of the step
      // (imagerepresentation.cpp:1254-1268, 1288-1296)
      rc = detect_orientation_batch(c, cimg, n, r0, pp.ori_mrSize, pp.ori_patchSize, oriHalf, pp.ori_maxAngles, pp.ori_threshold,
                                    0, ro);
    }
    t3 = tnow();
    tq = t3;
    if (!rc) {
      float *dF[MAXB];
      uint8_t *dU[MAXB];
      uint8_t *dUx[3][MAXB];
      size_t ofs = total;
      host_parallel_light(n, [&](int i) {
        int m = reproject_regions(ro[i].data(), (int)ro[i].size(), Hs[i], gimg[i]->cols, gimg[i]->rows);
        ro[i].resize(m);
      });
      for (int i = 0; i < n; i++) {
        const int m = (int)ro[i].size();
        dF[i] = devF ? devF + ofs * 128 : nullptr;
        dU[i] = devU8 ? devU8 + ofs * 128 : nullptr;
        for (int k = 1; k < ds.n; k++) dUx[k - 1][i] = devU8x[k - 1] ? devU8x[k - 1] + ofs * 128 : nullptr;
        ofs += m;
      }
      if (tim2) { fprintf(stderr, "  host %-28s %.3f ms\n", "reproject", tnow() - tq); tq = tnow(); }
      if (ofs > devCapRegions && (devF || devU8)) { set_error("descriptor buffer too small"); rc = MODSX_ERR_CAPACITY; }
      if (!rc) {
        uint8_t *const *xs[3] = {ds.n > 1 ? dUx[0] : nullptr, ds.n > 2 ? dUx[1] : nullptr, ds.n > 3 ? dUx[2] : nullptr};
        rc = describe_batch(c, cimg, n, ro, pp.desc_mrSize, pp.desc_patchSize, 0, pp.desc_photoNorm, ds.type[0],
                            pp.desc_maxBinValue, nullptr, devF ? dF : nullptr, devU8 ? dU : nullptr, &ds, xs);
      }
      if (tim2) tq = tnow();
      if (!rc) {
        for (int i = 0; i < n; i++) {
          if (hostDesc && !ro[i].empty()) {
            if (devF) {
              hipMemcpyAsync(hostDesc + total * 128, dF[i], ro[i].size() * 512, hipMemcpyDeviceToHost, c->stream);
            } else {
              hipMemcpyAsync(hostDesc + total * 128, c->descF[i].p, ro[i].size() * 512, hipMemcpyDeviceToHost, c->stream);
            }
          }
          if (itemCounts) itemCounts[take[g0 + i]] = (int)ro[i].size();
        }
        {   // the set's regions behind the list, one copy task per view
          size_t at[MAXB + 1];
          at[0] = regs.size();
          for (int i = 0; i < n; i++) at[i + 1] = at[i] + ro[i].size();
          // appended view by view (resize + copy would zero-fill 200 bytes per region first)
          if (regs.capacity() < at[n]) regs.reserve(std::max(at[n], 2 * regs.capacity()));
          for (int i = 0; i < n; i++) regs.insert(regs.end(), ro[i].begin(), ro[i].end());
          total += at[n] - at[0];
        }
        if (hostDesc) hipStreamSynchronize(c->stream);    // (copies into the caller's pageable memory; everything else was waited for by describe_batch)
        if (tim2) fprintf(stderr, "  host %-28s %.3f ms\n", "region list append", tnow() - tq);
      }
    }
    t4 = tnow();
    if (tim) fprintf(stderr, "set of %d views: synth %.2f detect %.2f orient %.2f describe %.2f ms\n", n, t1 - t0, t2 - t1, t3 - t2, t4 - t3);
    for (int i = 0; i < n; i++)
      if (vimg[i]) { if (vimg[i]->owned && vimg[i]->d) hipFree(vimg[i]->d); delete vimg[i]; }
    if (rc) return rc;
  }
  return MODSX_OK;
}

// The per-view loop of ONE image for views view_begin, view_begin + view_step, ... < nv; `viewCounts[v]` gets the number of
// described regions of view v (0 for views not taken).
int detect_describe_views(modsx_ctx *c, const modsx_image *gray, const modsx_view *views, int nv,
                          const modsx_pair_params &pp, int view_begin, int view_step, std::vector<modsx_region> &regs,
                          float *devF, uint8_t *devU8, size_t devCapRegions, float *hostDesc, int *viewCounts,
                          const DescSet *dsIn, uint8_t *const *devU8x) {
  if (viewCounts) for (int v = 0; v < nv; v++) viewCounts[v] = 0;
  if (view_step < 1) view_step = 1;
  std::vector<const modsx_image *> im;
  std::vector<int> vw;
  for (int v = view_begin; v < nv; v += view_step) { im.push_back(gray); vw.push_back(v); }
  std::vector<int> cnt(std::max<size_t>(1, vw.size()), 0);
  const int rc = detect_describe_items(c, im.data(), vw.data(), (int)vw.size(), views, pp, regs, devF, devU8, devCapRegions, hostDesc,
                                       cnt.data(), dsIn, devU8x);
  if (viewCounts) for (size_t k = 0; k < vw.size(); k++) viewCounts[vw[k]] = cnt[k];
  return rc;
}

// AddRegionsToList (imagerepresentation.cpp:588-600): ids of each appended view block are shifted by the size
// of the list so far (`base` = regions already in the list from earlier ladder steps).  Block boundaries come from the
// per-view counts of the loop, not from runs of equal img_id (every identity-like view carries img_id 0).
void rebase_ids(std::vector<modsx_region> &regs, const int *viewCounts, int nv, size_t base) {
  size_t start = 0;
  for (int v = 0; v < nv; v++) {
    const size_t end = std::min(regs.size(), start + (size_t)viewCounts[v]);
    for (size_t i = start; i < end; i++) { regs[i].id += (int)(base + start); regs[i].parent_id += (int)(base + start); }
    start = end;
  }
}

static void release_result_arrays(modsx_pair_result *res) {
  free(res->tentatives); free(res->ransac_inlier); free(res->verified);
  res->tentatives = nullptr; res->ransac_inlier = nullptr; res->verified = nullptr;
}

// One (detector, descriptor) class of the pair -- RegionVectorMap[det][desc] of both ImageRepresentations and
// CorrespondencesMapMap[desc][det] (imagerepresentation.cpp:552-600, correspondencebank.cpp:180-218): its accumulated regions
// per image, the device buffers that hold their u8 descriptors, and the tentatives of its last match.
struct LadderClass {
  std::vector<modsx_region> regs[2];
  size_t cap[2] = {(size_t)1 << 16, (size_t)1 << 16};
  DevBuf *buf[2] = {nullptr, nullptr};
  std::vector<modsx_tentative> tents;
};

// Append the regions + u8 descriptors of one step's views to the accumulated lists of one image side
// (SynthDetectDescribeKeypoints + AddRegions, imagerepresentation.cpp:603-2047).  The accumulator is re-allocated
// (device-to-device copy) when the step does not fit.
// A peer / half context is driven by ONE host thread for as long as it lives: a thread spawned per call starts with a cold
// allocator arena and fresh thread_local scratch every time (page faults worth milliseconds on some calls of a lone pair).
struct CtxWorker {
  std::thread th;
  std::mutex mu;
  std::condition_variable cv;
  std::function<void()> task;
  bool has = false, done = true, stop = false;
};
static void ctx_worker_post(modsx_ctx *c, std::function<void()> fn) {
  if (!c->worker) {
    CtxWorker *w = new CtxWorker();
    c->worker = w;
    const int dev = c->dev;
    w->th = std::thread([w, dev] {
      hipSetDevice(dev);
      for (;;) {
        std::function<void()> f;
        {
          std::unique_lock<std::mutex> lk(w->mu);
          w->cv.wait(lk, [&] { return w->has || w->stop; });
          if (w->stop) return;
          f = std::move(w->task);
          w->has = false;
        }
        f();
        { std::lock_guard<std::mutex> lk(w->mu); w->done = true; }
        w->cv.notify_all();
      }
    });
  }
  CtxWorker *w = (CtxWorker *)c->worker;
  { std::lock_guard<std::mutex> lk(w->mu); w->task = std::move(fn); w->has = true; w->done = false; }
  w->cv.notify_all();
}
static void ctx_worker_wait(modsx_ctx *c) {
  CtxWorker *w = (CtxWorker *)c->worker;
  if (!w) return;
  std::unique_lock<std::mutex> lk(w->mu);
  w->cv.wait(lk, [&] { return w->done; });
}
void ctx_worker_stop(modsx_ctx *c) {
  CtxWorker *w = (CtxWorker *)c->worker;
  if (!w) return;
  { std::lock_guard<std::mutex> lk(w->mu); w->stop = true; }
  w->cv.notify_all();
  if (w->th.joinable()) w->th.join();
  delete w;
  c->worker = nullptr;
}

// split = true (a lone pair on an otherwise idle GPU): the views [m, nv) of the step run on a helper context -- its own
// stream, scratch and host thread -- beside the views [0, m) on c; m balances the view areas.  A view's regions and
// descriptors do not depend on which launch set it is part of, so the step is the concatenation of the two parts (the
// second part's descriptors are moved behind the first's once its size is known).
static int accumulate_views(modsx_ctx *c, LadderClass *const *ks, const DescSet &ds, int side, const modsx_image *img,
                            const modsx_view *views, int nv, const modsx_pair_params &pp, modsx_comm *cm, bool split = false) {
  // ks[j] = the class of the step's j-th descriptor: all of them receive the step's regions (one oriented list for the
  // whole step), each its own descriptors.  Their lists may differ in length (earlier steps may have carried other
  // descriptors), so every class has its own base.
  const int nd = ds.n;
  size_t base[MODSX_MAX_DESC];
  for (int j = 0; j < nd; j++) base[j] = ks[j]->regs[side].size();
  std::vector<modsx_region> step;
  std::vector<int> counts(std::max(1, nv), 0);
  auto grow_bufs = [&]() -> int {     // the accumulated descriptors of the classes: cap regions, earlier steps' kept
    for (int j = 0; j < nd; j++) {
      DevBuf &buf = *ks[j]->buf[side];
      const size_t cap = ks[j]->cap[side];
      if (buf.cap >= cap * 128) continue;
      DevBuf bigger;
      if (!bigger.ensure(cap * 128)) return MODSX_ERR_NOMEM;
      if (base[j]) {
        MX_HIP(hipMemcpyAsync(bigger.p, buf.p, base[j] * 128, hipMemcpyDeviceToDevice, c->stream));
        MX_HIP(ctx_sync(c));
      }
      buf.release();
      buf = bigger;
    }
    return MODSX_OK;
  };
  auto room = [&]() { size_t r = (size_t)-1; for (int j = 0; j < nd; j++) r = std::min(r, ks[j]->cap[side] - base[j]); return r; };
  auto dst = [&](int j, size_t row) { return (uint8_t *)ks[j]->buf[side]->p + (base[j] + row) * 128; };
  // the step's regions behind every class's list, ids re-based onto that list (AddRegionsToList)
  auto append = [&]() {
    for (int j = 0; j < nd; j++) {
      std::vector<modsx_region> &acc = ks[j]->regs[side];
      const size_t at = acc.size();
      acc.insert(acc.end(), step.begin(), step.end());
      size_t start = 0;
      for (int v = 0; v < nv; v++) {
        const size_t end = std::min(step.size(), start + (size_t)counts[v]);
        for (size_t i = start; i < end; i++) { acc[at + i].id += (int)(base[j] + start); acc[at + i].parent_id += (int)(base[j] + start); }
        start = end;
      }
    }
  };
  const bool noSplit = getenv("MODSX_PAIR_NOSPLIT") != nullptr;   // read per call: bench.py switches it on for its one-stream leg
  static const double splitBias = getenv("MODSX_SPLIT_BIAS") ? atof(getenv("MODSX_SPLIT_BIAS")) : 0.25;   // a view's fixed cost, in untilted-view areas
  static const int nParts = getenv("MODSX_PAIR_PARTS") ? std::max(1, std::min(8, atoi(getenv("MODSX_PAIR_PARTS")))) : 3;   // 31 views: 2 / 3 / 4 parts 13.7 / 12.8 / 13.1 ms per pair
  if (split && !cm && !noSplit && nParts > 1 && nv >= 6 * nParts) {
    { const int rg = grow_bufs(); if (rg) return rg; }
    const size_t cap = room();
    // part p = views [cut[p], cut[p + 1]): equal shares of the views' weights (area + a fixed cost per view)
    const int P = nParts;
    std::vector<double> w(nv);
    double tot = 0;
    for (int v = 0; v < nv; v++) { const double t = fabs(views[v].tilt) > 1e-9 ? fabs(views[v].tilt) : 1.0; w[v] = views[v].zoom * views[v].zoom / t + splitBias; tot += w[v]; }
    std::vector<int> cut(P + 1, nv);
    cut[0] = 0;
    {
      double run = 0;
      int p = 1;
      for (int v = 0; v < nv && p < P; v++) {
        run += w[v];
        if (run >= tot * p / P && v + 1 < nv - (P - 1 - p)) cut[p++] = v + 1;
      }
      for (; p < P; p++) cut[p] = std::max(cut[p - 1] + 1, nv - (P - p));
    }
    // parts 1 .. P - 1 run on the chain of helper contexts c->half, c->half->half, ...
    std::vector<modsx_ctx *> hs(P, nullptr);
    bool ready = true;
    {
      modsx_ctx *prev = c;
      for (int p = 1; p < P && ready; p++) {
        if (!prev->half) prev->half = ctx_create(c->dev);
        hs[p] = prev->half;
        ready = hs[p] != nullptr;
        for (int j = 0; j < nd && ready; j++) ready = hs[p]->halfDesc[j].ensure(cap * 128);
        prev = hs[p];
      }
    }
    if (ready) {
      std::vector<std::vector<modsx_region>> part(P);
      std::vector<std::vector<int>> cnt(P, std::vector<int>(nv, 0));
      std::vector<int> rcs(P, MODSX_OK);
      std::vector<std::string> errs(P);
      for (int p = 1; p < P; p++) {
        modsx_ctx *h = hs[p];
        prof_reset(h, c->prof.enabled);
        ctx_worker_post(h, [&, p, h]() {
          host_light_pool(true);
          uint8_t *xs[3] = {(uint8_t *)h->halfDesc[1].p, (uint8_t *)h->halfDesc[2].p, (uint8_t *)h->halfDesc[3].p};
          rcs[p] = detect_describe_views(h, img, views, cut[p + 1], pp, cut[p], 1, part[p], nullptr, (uint8_t *)h->halfDesc[0].p, cap, nullptr,
                                         cnt[p].data(), &ds, xs);
          if (rcs[p]) errs[p] = last_error();
        });
      }
      host_light_pool(true);
      {
        uint8_t *xs[3] = {nd > 1 ? dst(1, 0) : nullptr, nd > 2 ? dst(2, 0) : nullptr, nd > 3 ? dst(3, 0) : nullptr};
        rcs[0] = detect_describe_views(c, img, views, cut[1], pp, 0, 1, step, nullptr, dst(0, 0), cap, nullptr, counts.data(), &ds, xs);
      }
      host_light_pool(false);
      size_t total = step.size();
      bool ok = rcs[0] == MODSX_OK;
      for (int p = 1; p < P; p++) {
        ctx_worker_wait(hs[p]);
        if (c->prof.enabled) {
          prof_collect(hs[p]);
          for (int q = 0; q < K_NCLASS; q++) { c->prof.ms[q] += hs[p]->prof.ms[q]; c->prof.work[q] += hs[p]->prof.work[q]; c->prof.launches[q] += hs[p]->prof.launches[q]; }
        }
        ok = ok && rcs[p] == MODSX_OK;
        total += part[p].size();
      }
      if (ok && total <= cap) {
        size_t at = step.size();
        for (int p = 1; p < P; p++) {
          if (!part[p].empty())
            for (int j = 0; j < nd; j++)
              MX_HIP(hipMemcpyAsync(dst(j, at), hs[p]->halfDesc[j].p, part[p].size() * 128, hipMemcpyDeviceToDevice, c->stream));
          at += part[p].size();
          for (int v = cut[p]; v < cut[p + 1]; v++) counts[v] = cnt[p][v];
        }
        MX_HIP(ctx_sync(c));
        step.reserve(total);
        for (int p = 1; p < P; p++) step.insert(step.end(), part[p].begin(), part[p].end());
        append();
        return MODSX_OK;
      }
      for (int p = 0; p < P; p++)
        if (rcs[p] && rcs[p] != MODSX_ERR_CAPACITY) { if (p) set_error(errs[p]); return rcs[p]; }
      // a part did not fit: the one-context path below grows the buffers and runs the step again
      step.clear();
      std::fill(counts.begin(), counts.end(), 0);
    }
  }
  if (cm) {   // view-sharded: this rank runs its views, the exchange appends the whole step in reference order (ids re-based)
    DevBuf *accs[MODSX_MAX_DESC];
    for (int j = 0; j < nd; j++) accs[j] = ks[j]->buf[side];
    int rc = detect_describe_views_sharded(c, cm, img, views, nv, pp, ds, step, accs, base, counts.data());
    if (rc) return rc;
    append();
    return MODSX_OK;
  }
  for (;;) {
    { const int rg = grow_bufs(); if (rg) return rg; }
    uint8_t *xs[3] = {nd > 1 ? dst(1, 0) : nullptr, nd > 2 ? dst(2, 0) : nullptr, nd > 3 ? dst(3, 0) : nullptr};
    int rc = detect_describe_views(c, img, views, nv, pp, 0, 1, step, nullptr, dst(0, 0), room(), nullptr, counts.data(), &ds, xs);
    if (rc == MODSX_ERR_CAPACITY && ks[0]->cap[side] < ((size_t)1 << 24)) {   // only "buffer too small" grows them
      for (int j = 0; j < nd; j++) ks[j]->cap[side] *= 4;
      continue;
    }
    if (rc) return rc;
    break;
  }
  append();
  return MODSX_OK;
}

int match_pair_views(modsx_ctx *c, const modsx_image *img1, const modsx_image *img2, const modsx_view *views, int nv,
                     const modsx_pair_params &pp, modsx_pair_result *res, VerifyTask *defer) {
  modsx_ladder_step one;
  memset(&one, 0, sizeof one);
  one.views = views; one.nviews = nv; one.match_ratio = pp.match_ratio; one.detector = pp.detector;
  int done = 0;
  return match_ladder(c, img1, img2, &one, 1, 0x7fffffff, pp, res, &done, defer);
}

// The iteration loop of mods.cpp:229-415 (HessianAffine and MSER classes with SIFT-family descriptors, LO-RANSAC
// verification, duplicates filtered before RANSAC): every step adds its views' regions to both image representations,
// re-matches the (detector, descriptor) classes it extended, and the ladder stops once min_matches verified
// correspondences exist.
int match_ladder(modsx_ctx *c, const modsx_image *img1, const modsx_image *img2, const modsx_ladder_step *steps, int nsteps,
                 int min_matches, const modsx_pair_params &pp, modsx_pair_result *res, int *steps_done, VerifyTask *defer,
                 modsx_comm *cm, int owner) {
  CtxBusy busy(c);
  if (defer && nsteps != 1) { set_error("deferred verification needs a one-step ladder"); return MODSX_ERR_ARG; }
  memset(res, 0, sizeof *res);
  for (int i = 0; i < 9; i++) res->H[i] = -1;
  const modsx_image *imgs[2] = {img1, img2};
  // cls[det][type]: det 0 = HessianAffine, 1 = MSER; type = MODSX_DESC_*.  GetCorresponcesVector("All", "All") walks
  // descriptor names, then detector names (correspondencebank.cpp:117-179): types 3, 2, 1, 0, HessianAffine before MSER.
  LadderClass cls[2][4];
  for (int d = 0; d < 2; d++) for (int t = 0; t < 4; t++) for (int sd = 0; sd < 2; sd++) cls[d][t].buf[sd] = &c->descCls[d][t][sd];
  static std::atomic<int> active(0);
  struct Guard { std::atomic<int> &a; ~Guard() { a.fetch_sub(1); } } guard{active};
  const bool alone = active.fetch_add(1) == 0;
  int cur = 0, step = 0;
  auto tnowL = []() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
  const bool timL = getenv("MODSX_HOST_TIMING") != nullptr;
  for (; step < nsteps && cur < min_matches; step++) {
    const double tL0 = tnowL();
    const int det = steps[step].detector == MODSX_DET_MSER ? 1 : 0;
    DescSet ds;
    { const int rd = resolve_descs(pp, &steps[step], ds); if (rd) { release_result_arrays(res); return rd; } }
    LadderClass *ks[MODSX_MAX_DESC];
    for (int j = 0; j < ds.n; j++) ks[j] = &cls[det][ds.type[j]];
    modsx_pair_params ps = pp;
    ps.detector = det ? MODSX_DET_MSER : MODSX_DET_HESSIAN;
    {
      // The two images are independent until the match (mods.cpp:255-271 runs them in two OpenMP threads): image 2 goes
      // through a peer context (own stream, own scratch) on a second host thread, so the host bookkeeping of one image
      // overlaps the kernels of the other (31 views: 42 -> 32 ms per pair).  Only when this is the one ladder running in the
      // process: with many contexts at work the GPU is already full and the extra streams and scratch cost throughput
      // (16 workers: 123 -> 77 pairs/s).  MODSX_PAIR_SERIAL=1 keeps one stream (measurements).
      const bool serialEnv = getenv("MODSX_PAIR_SERIAL") != nullptr;   // read per call (see MODSX_PAIR_NOSPLIT)
      const bool serial = serialEnv || !alone || cm;   // a sharded ladder issues its collectives from one thread, in one order
      int rc0 = MODSX_OK, rc1 = MODSX_OK;
      std::string err1;
      if (!serial && !c->peer) c->peer = ctx_create(c->dev);
      if (!serial && c->peer) {
        modsx_ctx *pc = c->peer;
        prof_reset(pc, c->prof.enabled);
        ctx_worker_post(pc, [&]() {
          rc1 = accumulate_views(pc, ks, ds, 1, imgs[1], steps[step].views, steps[step].nviews, ps, nullptr, true);
          if (rc1) err1 = last_error();
        });
        rc0 = accumulate_views(c, ks, ds, 0, imgs[0], steps[step].views, steps[step].nviews, ps, nullptr, true);
        ctx_worker_wait(pc);
        if (c->prof.enabled) {   // the peer's kernels belong to this call
          prof_collect(pc);
          for (int q = 0; q < K_NCLASS; q++) { c->prof.ms[q] += pc->prof.ms[q]; c->prof.work[q] += pc->prof.work[q]; c->prof.launches[q] += pc->prof.launches[q]; }
        }
        if (!rc0 && rc1) set_error(err1);
      } else {
        rc0 = accumulate_views(c, ks, ds, 0, imgs[0], steps[step].views, steps[step].nviews, ps, cm);
        if (!rc0) rc1 = accumulate_views(c, ks, ds, 1, imgs[1], steps[step].views, steps[step].nviews, ps, cm);
      }
      if (rc0 || rc1) { release_result_arrays(res); return rc0 ? rc0 : rc1; }
    }
    const double tL1 = tnowL();
    // Tentatives.MatchImgReps (correspondencebank.cpp:291-345): clear and re-match every (this detector, descriptor) class of
    // the step, each with the FGINNThreshold of its descriptor
    for (int j = 0; j < ds.n; j++) {
      LadderClass &k = *ks[j];
      std::vector<double> pos2(k.regs[1].size() * 2 + 2);
      for (size_t i = 0; i < k.regs[1].size(); i++) { pos2[2 * i] = k.regs[1][i].reproj_kp.x; pos2[2 * i + 1] = k.regs[1][i].reproj_kp.y; }
      int rc = cm ? match_sharded(c, cm, (uint8_t *)k.buf[0]->p, (int)k.regs[0].size(), (uint8_t *)k.buf[1]->p, (int)k.regs[1].size(),
                                  pos2.data(), ds.ratio[j], pp.contradDist, pp.nn, k.tents)
                  : match_device(c, (uint8_t *)k.buf[0]->p, (int)k.regs[0].size(), (uint8_t *)k.buf[1]->p, (int)k.regs[1].size(),
                                 pos2.data(), ds.ratio[j], pp.contradDist, pp.nn, k.tents);
      if (rc) { release_result_arrays(res); return rc; }
    }
    const double tL2 = tnowL();
    // GetCorresponcesVector(): the classes in map order; indices re-based onto the concatenation of their region lists
    RegList l1, l2;
    std::vector<modsx_tentative> tents;
    int nonEmpty = 0;
    LadderClass *lastCls = nullptr;
    for (int t = 3; t >= 0; t--)
      for (int d = 0; d < 2; d++)
        if (!cls[d][t].regs[0].empty() || !cls[d][t].regs[1].empty()) { nonEmpty++; lastCls = &cls[d][t]; }
    if (nonEmpty == 1 && !defer) tents = lastCls->tents;   // one class: no copy of 12 k tentatives with re-based indices
    for (int t = 3; t >= 0; t--)
      for (int d = 0; d < 2; d++) {
        LadderClass &k = cls[d][t];
        if (k.regs[0].empty() && k.regs[1].empty()) continue;
        const int o1 = (int)l1.size(), o2 = (int)l2.size();
        l1.add(k.regs[0]); l2.add(k.regs[1]);
        if (nonEmpty == 1 && !defer) continue;
        for (modsx_tentative tt : k.tents) {
          tt.q += o1; tt.t0 += o2;
          if (tt.t1 >= 0) tt.t1 += o2;
          if (tt.tj >= 0) tt.tj += o2;
          tents.push_back(tt);
        }
      }
    release_result_arrays(res);
    memset(res, 0, sizeof *res);
    for (int i = 0; i < 9; i++) res->H[i] = -1;
    res->n_regions1 = (int)l1.size();
    res->n_regions2 = (int)l2.size();
    if (defer) {   // the caller runs DuplicateFiltering + LO-RANSAC elsewhere (modsx_match_pairs_views: helper threads)
      defer->l1.clear(); defer->l2.clear(); defer->own.clear();
      defer->own.reserve(16);
      for (int t = 3; t >= 0; t--)
        for (int d = 0; d < 2; d++) {
          LadderClass &k = cls[d][t];
          if (k.regs[0].empty() && k.regs[1].empty()) continue;
          defer->own.emplace_back(std::move(k.regs[0])); defer->l1.add(defer->own.back());
          defer->own.emplace_back(std::move(k.regs[1])); defer->l2.add(defer->own.back());
        }
      defer->tents = std::move(tents); defer->res = res; defer->dev = c->dev;
      step++;
      break;
    }
    // a sharded single-step call may leave verification to the owner rank; a sharded ladder verifies on every rank (same
    // tentatives, same seed => same count), which is how the ranks agree on the early exit without a collective
    const double tL3 = tnowL();
    res->n_tentatives = (int)tents.size();
    if (!cm || owner < 0 || owner == comm_rank(cm)) verify_tentatives(l1, l2, tents, pp, res);
    cur = res->n_verified;
    if (cm && owner < 0 && nsteps > 1) {
      // the exit of a sharded ladder is AGREED, not assumed: one 4-byte all-gather of the verified count per step.  The ranks
      // of a node share one host, so they compute the same count; a rank that did not (another CPU, another libm) would
      // otherwise leave the loop at another step and hang the rest in the next exchange
      const int ra = comm_same_value(c, cm, cur, "the verified count of a ladder step");
      if (ra) { release_result_arrays(res); return ra; }
    }
    if (timL) fprintf(stderr, "ladder step %d: views %.2f match %.2f lists %.2f verify %.2f ms\n", step, tL1 - tL0, tL2 - tL1, tL3 - tL2, tnowL() - tL3);
  }
  if (steps_done) *steps_done = step;
  prof_collect(c);
  return MODSX_OK;
}

}  // namespace mx
