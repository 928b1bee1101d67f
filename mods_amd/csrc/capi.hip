// capi.hip -- the extern "C" boundary declared in include/modsx.h.
#include <math.h>
#include <malloc.h>
#include <atomic>
#include <chrono>
#include <mutex>
#include <thread>
#include <condition_variable>
#include <deque>
#include "engine_api.hpp"

using namespace mx;

#define NEED(c)                                             \
  do {                                                      \
    if (!(c)) { mx::set_error("null argument: " #c); return MODSX_ERR_ARG; } \
  } while (0)

template <typename T>
static T *to_malloc(const std::vector<T> &v) {
  T *p = (T *)malloc(sizeof(T) * (v.size() ? v.size() : 1));
  if (p && !v.empty()) memcpy(p, v.data(), sizeof(T) * v.size());
  return p;
}

// host share of the last modsx_match_pairs call: summed wall time of DuplicateFiltering + LO-RANSAC over its pairs, and
// the number of helper threads that ran them
static std::atomic<long> g_verifyUs(0);
static std::atomic<int> g_verifyThreads(0), g_verifyPairs(0);
static std::mutex g_verifyEachMu;
static std::vector<int> g_verifyEachUs;    // verification time of every pair of the last batch call, in completion order
static void verify_timed(mx::VerifyTask &t, const modsx_pair_params &pp);
// CPU time (not wall) of the last batch call, by kind of thread: the contexts' own threads and the verification helpers
static std::atomic<long> g_cpuWorkerUs(0), g_cpuHelperUs(0);
// helper threads of a batch call: one per working context unless MODSX_VERIFY_HELPERS says otherwise
static int verify_helpers(int nw) {
  static const int cap = getenv("MODSX_VERIFY_HELPERS") ? atoi(getenv("MODSX_VERIFY_HELPERS")) : 0;
  return cap > 0 ? (cap < nw ? cap : nw) : nw;
}
static long thread_cpu_us() { timespec t; clock_gettime(CLOCK_THREAD_CPUTIME_ID, &t); return t.tv_sec * 1000000L + t.tv_nsec / 1000; }
struct ThreadCpu { std::atomic<long> &acc; long t0; explicit ThreadCpu(std::atomic<long> &a) : acc(a), t0(thread_cpu_us()) {} ~ThreadCpu() { acc += thread_cpu_us() - t0; } };

extern "C" {

int modsx_version(void) { return MODSX_VERSION; }
const char *modsx_last_error(void) { return mx::last_error(); }
void modsx_free(void *p) { free(p); }

// The host stages of a pair allocate and release ~100 MB of job tables, region lists and tentative arrays per call.  With
// glibc's defaults every block above 128 KB is its own mmap: fresh zero pages faulted in on first touch and unmapped on free --
// about 1 ms of page faults per 31-view pair (measured: 16.3 -> 15.3 ms).  Once per process the thresholds are raised so that
// such blocks come from, and return to, the heap and stay mapped.  This changes malloc for the whole host process, so it is
// OPT-IN: MODSX_MALLOC_TUNE=1 (bench.py and tools/latency.py set it; a host application decides for itself).
static void tune_host_allocator() {
#if defined(__GLIBC__)
  static std::once_flag once;
  std::call_once(once, [] {
    const char *e = getenv("MODSX_MALLOC_TUNE");
    if (!e || atoi(e) == 0) return;
    mallopt(M_MMAP_THRESHOLD, 1 << 30);
    mallopt(M_TRIM_THRESHOLD, 1 << 30);
    mallopt(M_TOP_PAD, 64 << 20);
  });
#endif
}

modsx_ctx *modsx_create(int device_id) {
  tune_host_allocator();
  return ctx_create(device_id);
}
void modsx_destroy(modsx_ctx *ctx) { ctx_destroy(ctx); }
int modsx_synchronize(modsx_ctx *ctx) {
  NEED(ctx);
  MX_HIP(hipStreamSynchronize(ctx->stream));
  return MODSX_OK;
}

void modsx_default_hessaff_params(modsx_hessaff_params *p) {
  // build/config_iter_mods_cviu.ini:13-27; structures.hpp:141-160; affine.h:47-58
  p->threshold = 5.3333f;
  p->mode = MODSX_FIXED_TH;
  p->reg_number = 2000;
  p->rel_threshold = -1;
  p->rel_reg_number = -1;
  p->numberOfScales = 3;
  p->initialSigma = 1.6f;
  p->edgeEigenValueRatio = 10.0;
  p->border = 5;
  p->maxIterations = 16;
  p->convergenceThreshold = 0.05f;
  p->smmWindowSize = 19;
  p->affInitialSigma = 1.6f;
  p->doBaumberg = 1;
  p->detectorType = MODSX_DET_HESSIAN;
}

void modsx_default_mser_params(modsx_mser_params *p);
void modsx_default_pair_params(modsx_pair_params *p) {
  memset(p, 0, sizeof *p);
  modsx_default_hessaff_params(&p->det);
  p->ori_mrSize = 1.0; p->ori_patchSize = 41; p->ori_maxAngles = 1; p->ori_threshold = 0.8;
  p->desc_mrSize = 5.1962; p->desc_patchSize = 41; p->desc_photoNorm = 1; p->desc_type = MODSX_DESC_ROOT_SIFT;
  p->desc_maxBinValue = 0.2;
  p->match_ratio = 0.8; p->contradDist = 30.0; p->nn = 50;
  p->duplicateDist = 2.0;
  p->err_threshold = 3.0; p->confidence = 0.99; p->max_samples = 100000; p->localOptimization = 1;
  p->HLAFCoef = 12.0; p->doSymmCheck = 1;
  p->ransac_seed = 1;
  p->useF = 0; p->LAFCoef = 2.0; p->errorType = 0;
  p->detector = MODSX_DET_HESSIAN;
  modsx_default_mser_params(&p->mser);
}

// the samplers address a pixel by a 24-bit row x column product and a 32-bit byte offset from the image base (kmath.hpp,
// bilinear_tap); views and pyramid levels are never larger than the image they come from by more than the rotation's
// bounding box (< 2x per side), so the bound is taken with that margin
static bool image_size_ok(int rows, int cols) {
  if (rows < 2 || cols < 2) {   // interpolate()'s clamp `min(max(x, 0), cols - 2)` needs two pixels per side (kmath.hpp bilinear_blend)
    mx::set_error("images of fewer than 2 rows or 2 columns are not supported");
    return false;
  }
  if (rows > 16384 || cols > 16384 || (long long)rows * cols > (1ll << 26)) {
    mx::set_error("image larger than 16384 px per side / 64 Mpx is not supported");
    return false;
  }
  return true;
}

modsx_image *modsx_image_upload(modsx_ctx *ctx, const void *pixels, int rows, int cols, int channels, int dtype) {
  if (!ctx || !pixels || rows <= 0 || cols <= 0 || (channels != 1 && channels != 3) || (dtype != 0 && dtype != 1)) {
    mx::set_error("modsx_image_upload: bad argument");
    return nullptr;
  }
  if (!image_size_ok(rows, cols)) return nullptr;
  hipSetDevice(ctx->dev);
  const size_t n = (size_t)rows * cols;
  const size_t inBytes = n * channels * (dtype == 0 ? 1 : 4);
  modsx_image *im = new modsx_image();
  im->rows = rows; im->cols = cols; im->owned = true; im->d = nullptr;
  if (hipMalloc(&im->d, n * 4) != hipSuccess) { mx::set_error("hipMalloc image"); delete im; return nullptr; }
  if (!ctx->misc.ensure(inBytes)) { hipFree(im->d); delete im; return nullptr; }
  if (hipMemcpyAsync(ctx->misc.p, pixels, inBytes, hipMemcpyHostToDevice, ctx->stream) != hipSuccess) {
    mx::set_error("image H2D copy failed"); hipFree(im->d); delete im; return nullptr;
  }
  launch_gray(ctx->stream, ctx->misc.p, im->d, n, channels, dtype);
  if (hipStreamSynchronize(ctx->stream) != hipSuccess) { mx::set_error("image upload sync failed"); hipFree(im->d); delete im; return nullptr; }
  return im;
}

int modsx_image_update(modsx_ctx *ctx, modsx_image *im, const void *pixels, int rows, int cols, int channels, int dtype) {
  NEED(ctx); NEED(im); NEED(pixels);
  if ((channels != 1 && channels != 3) || (dtype != 0 && dtype != 1) || !im->owned || rows != im->rows || cols != im->cols) {
    mx::set_error("modsx_image_update: bad argument (the image must be an uploaded one of the same size)");
    return MODSX_ERR_ARG;
  }
  hipSetDevice(ctx->dev);
  const size_t n = (size_t)rows * cols, inBytes = n * channels * (dtype == 0 ? 1 : 4);
  if (!ctx->misc.ensure(inBytes)) return MODSX_ERR_NOMEM;
  if (hipMemcpyAsync(ctx->misc.p, pixels, inBytes, hipMemcpyHostToDevice, ctx->stream) != hipSuccess) { mx::set_error("image H2D copy failed"); return MODSX_ERR_DEVICE; }
  launch_gray(ctx->stream, ctx->misc.p, im->d, n, channels, dtype);
  // the caller's buffer is its own again on return (a copy from pageable memory is staged by the runtime; this wait also covers pinned sources)
  if (hipStreamSynchronize(ctx->stream) != hipSuccess) { mx::set_error("image update sync failed"); return MODSX_ERR_DEVICE; }
  return MODSX_OK;
}

modsx_image *modsx_image_wrap_device(modsx_ctx *ctx, const float *dev_pixels, int rows, int cols) {
  if (!ctx || !dev_pixels || rows <= 0 || cols <= 0) { mx::set_error("modsx_image_wrap_device: bad argument"); return nullptr; }
  if (!image_size_ok(rows, cols)) return nullptr;
  modsx_image *im = new modsx_image();
  im->d = const_cast<float *>(dev_pixels); im->rows = rows; im->cols = cols; im->owned = false;
  return im;
}

void modsx_image_free(modsx_ctx *ctx, modsx_image *img) {
  if (!img) return;
  if (ctx) hipSetDevice(ctx->dev);
  if (img->owned && img->d) hipFree(img->d);
  delete img;
}

int modsx_image_download(modsx_ctx *ctx, const modsx_image *img, float *out) {
  NEED(ctx); NEED(img); NEED(out);
  MX_HIP(hipMemcpyAsync(out, img->d, (size_t)img->rows * img->cols * 4, hipMemcpyDeviceToHost, ctx->stream));
  MX_HIP(hipStreamSynchronize(ctx->stream));
  return MODSX_OK;
}

int modsx_detect_affine_keypoints(modsx_ctx *ctx, const modsx_image *img, const modsx_hessaff_params *par, double tilt,
                                  double zoom, modsx_keypoint **out) {
  NEED(ctx); NEED(img); NEED(par); NEED(out);
  hipSetDevice(ctx->dev);
  std::vector<modsx_keypoint> k[1];
  const modsx_image *imgs[1] = {img};
  int rc = detect_keypoints_batch(ctx, imgs, 1, *par, &tilt, &zoom, k);
  if (rc) return rc;
  *out = to_malloc(k[0]);
  return (int)k[0].size();
}

int modsx_detect_scalespace(modsx_ctx *ctx, const modsx_image *img, const modsx_hessaff_params *par, modsx_sskp **out) {
  NEED(ctx); NEED(img); NEED(par); NEED(out);
  hipSetDevice(ctx->dev);
  std::vector<modsx_sskp> k[1];
  const modsx_image *imgs[1] = {img};
  int rc = detect_scalespace_batch(ctx, imgs, 1, *par, k);
  if (rc) return rc;
  *out = to_malloc(k[0]);
  return (int)k[0].size();
}

int modsx_octave_levels(modsx_ctx *ctx, const modsx_image *img, const modsx_hessaff_params *par, float *blurs,
                        float *resps) {
  NEED(ctx); NEED(img); NEED(par); NEED(blurs); NEED(resps);
  hipSetDevice(ctx->dev);
  const modsx_image *imgs[1] = {img};
  int rc = build_pyramids(ctx, imgs, 1, *par, true);
  if (rc) return rc;
  const int L = par->numberOfScales + 2;
  const size_t npx = (size_t)img->rows * img->cols;
  const Octave &o = ctx->pyr[0].oct[0];
  for (int l = 0; l < L; l++) {
    MX_HIP(hipMemcpyAsync(blurs + l * npx, o.blur[l], npx * 4, hipMemcpyDeviceToHost, ctx->stream));
    MX_HIP(hipMemcpyAsync(resps + l * npx, o.resp[l], npx * 4, hipMemcpyDeviceToHost, ctx->stream));
  }
  MX_HIP(hipStreamSynchronize(ctx->stream));
  return L;
}

int modsx_gaussian_blur(modsx_ctx *ctx, const modsx_image *img, float sigma, float *out) {
  NEED(ctx); NEED(img); NEED(out);
  hipSetDevice(ctx->dev);
  const size_t npx = (size_t)img->rows * img->cols;
  if (!ctx->scratchA.ensure(npx * 4)) return MODSX_ERR_NOMEM;
  BlurBatch b;
  memset(&b, 0, sizeof b);
  int n = blur_ksize(sigma);
  if (n > MAX_TAPS) { mx::set_error("modsx_gaussian_blur: ksize > 17"); return MODSX_ERR_ARG; }
  if (n == 1) {
    MX_HIP(hipMemcpyAsync(out, img->d, npx * 4, hipMemcpyDeviceToHost, ctx->stream));
    MX_HIP(hipStreamSynchronize(ctx->stream));
    return MODSX_OK;
  }
  std::vector<float> k = gaussian_kernel(n, sigma);
  b.n = n;
  for (int i = 0; i < n; i++) b.k[i] = k[i];
  b.j[0].src = img->d; b.j[0].blur = (float *)ctx->scratchA.p; b.j[0].resp = nullptr; b.j[0].rows = img->rows;
  b.j[0].cols = img->cols; b.j[0].norm = 1.f;
  launch_blur_hess(ctx->stream, b, 1, img->rows, img->cols);
  MX_HIP(hipMemcpyAsync(out, ctx->scratchA.p, npx * 4, hipMemcpyDeviceToHost, ctx->stream));
  MX_HIP(hipStreamSynchronize(ctx->stream));
  MX_HIP(hipGetLastError());
  return MODSX_OK;
}

// gaussianBlur(helpers.cpp:717-724) for any sigma (the tiled pyramid kernel stops at 17 taps): two generic passes, replicate border
static int blur_any(modsx_ctx *c, const float *src, float *dst, float *tmp, float *dTaps, float *hTaps, int rows, int cols, float sigma) {
  const int n = blur_ksize(sigma);
  if (n == 1) { MX_HIP(hipMemcpyAsync(dst, src, (size_t)rows * cols * 4, hipMemcpyDeviceToDevice, c->stream)); return MODSX_OK; }
  const int nx = cols == 1 ? 1 : n, ny = rows == 1 ? 1 : n;
  std::vector<float> kx = gaussian_kernel(nx, sigma), ky = gaussian_kernel(ny, sigma);
  memcpy(hTaps, kx.data(), nx * 4); memcpy(hTaps + nx, ky.data(), ny * 4);
  MX_HIP(hipMemcpyAsync(dTaps, hTaps, (size_t)(nx + ny) * 4, hipMemcpyHostToDevice, c->stream));
  launch_blur_pass(c->stream, src, tmp, rows, cols, dTaps, nx, 0, 1);
  launch_blur_pass(c->stream, tmp, dst, rows, cols, dTaps + nx, ny, 1, 1);
  return MODSX_OK;
}

int modsx_response(modsx_ctx *ctx, const modsx_image *img, int detector_type, float norm, float *out) {
  NEED(ctx); NEED(img); NEED(out);
  if (detector_type != MODSX_DET_HESSIAN && detector_type != MODSX_DET_DOG && detector_type != MODSX_DET_HARRIS) {
    mx::set_error("modsx_response: detector type must be Hessian (0), DoG (1) or Harris (2)");
    return MODSX_ERR_ARG;
  }
  hipSetDevice(ctx->dev);
  hipStream_t s = ctx->stream;
  const int rows = img->rows, cols = img->cols;
  const size_t npx = (size_t)rows * cols;
  const float sigma = detector_type == MODSX_DET_DOG ? norm : sqrtf((float)(0.6 * norm));
  const int ntap = 2 * std::max(1, blur_ksize(sigma)) * 3 + 16;
  if (ntap > 3 * 4096) { mx::set_error("modsx_response: blur kernel too large"); return MODSX_ERR_ARG; }
  if (!ctx->scratchA.ensure(npx * 4 * 8) || !ctx->viewTaps.ensure((size_t)ntap * 4) || !ctx->hViewTaps.ensure((size_t)ntap * 4)) return MODSX_ERR_NOMEM;
  float *buf = (float *)ctx->scratchA.p, *res = buf, *tmp = buf + npx, *a = buf + 2 * npx, *b = buf + 3 * npx, *cc = buf + 4 * npx;
  float *ba = buf + 5 * npx, *bb = buf + 6 * npx, *bc = buf + 7 * npx;
  float *dT = (float *)ctx->viewTaps.p, *hT = (float *)ctx->hViewTaps.p;
  int rc = MODSX_OK;
  if (detector_type == MODSX_DET_HESSIAN) {
    BlurBatch bt;
    memset(&bt, 0, sizeof bt);
    bt.j[0].src = img->d; bt.j[0].resp = res; bt.j[0].rows = rows; bt.j[0].cols = cols; bt.j[0].norm = norm;
    launch_hessian(s, bt, 1, rows, cols);
  } else if (detector_type == MODSX_DET_DOG) {
    rc = blur_any(ctx, img->d, a, tmp, dT, hT, rows, cols, sigma);
    if (rc) return rc;
    launch_sub(s, img->d, a, res, npx);
  } else {
    launch_grad_products(s, img->d, rows, cols, a, b, cc);
    const int n = blur_ksize(sigma);
    // the three blurs share the taps: separate slices of the staging buffer so no upload waits for a previous one
    rc = blur_any(ctx, a, ba, tmp, dT, hT, rows, cols, sigma);
    if (!rc) rc = blur_any(ctx, b, bb, tmp, dT + 2 * n + 2, hT + 2 * n + 2, rows, cols, sigma);
    if (!rc) rc = blur_any(ctx, cc, bc, tmp, dT + 4 * n + 4, hT + 4 * n + 4, rows, cols, sigma);
    if (rc) return rc;
    launch_harris_combine(s, ba, bb, bc, (float)(0.6 * norm), res, npx);
  }
  MX_HIP(hipMemcpyAsync(out, res, npx * 4, hipMemcpyDeviceToHost, s));
  MX_HIP(hipStreamSynchronize(s));
  MX_HIP(hipGetLastError());
  return MODSX_OK;
}

int modsx_resize_half(modsx_ctx *ctx, const modsx_image *img, float *out, int *orows, int *ocols) {
  NEED(ctx); NEED(img); NEED(orows); NEED(ocols);
  hipSetDevice(ctx->dev);
  const int dr = (int)lrint(img->rows * 0.5), dc = (int)lrint(img->cols * 0.5);
  *orows = dr; *ocols = dc;
  if (!out) return MODSX_OK;
  if (!ctx->scratchA.ensure((size_t)dr * dc * 4 + 4)) return MODSX_ERR_NOMEM;
  ResizeBatch rb;
  memset(&rb, 0, sizeof rb);
  rb.j[0].src = img->d; rb.j[0].dst = (float *)ctx->scratchA.p;
  rb.j[0].srows = img->rows; rb.j[0].scols = img->cols; rb.j[0].drows = dr; rb.j[0].dcols = dc;
  launch_resize_half(ctx->stream, rb, 1, dr, dc);
  MX_HIP(hipMemcpyAsync(out, ctx->scratchA.p, (size_t)dr * dc * 4, hipMemcpyDeviceToHost, ctx->stream));
  MX_HIP(hipStreamSynchronize(ctx->stream));
  MX_HIP(hipGetLastError());
  return MODSX_OK;
}

int modsx_detect_affine_regions(const modsx_keypoint *kps, int n, int img_id, int det_type, modsx_region *out) {
  if (n < 0 || (n > 0 && (!kps || !out))) { mx::set_error("modsx_detect_affine_regions: bad argument"); return MODSX_ERR_ARG; }
  detect_affine_regions(kps, n, img_id, det_type, out);
  return n;
}

int modsx_detect_orientation(modsx_ctx *ctx, const modsx_image *img, const modsx_region *in, int n, double mrSize,
                             int patchSize, int doHalfSIFT, int maxAngNum, double th, int addUpRight,
                             modsx_region **out) {
  NEED(ctx); NEED(img); NEED(out);
  if (n < 0 || (n > 0 && !in)) { mx::set_error("modsx_detect_orientation: bad argument"); return MODSX_ERR_ARG; }
  hipSetDevice(ctx->dev);
  std::vector<modsx_region> vin[1], vout[1];
  vin[0].assign(in, in + n);
  const modsx_image *imgs[1] = {img};
  int rc = detect_orientation_batch(ctx, imgs, 1, vin, mrSize, patchSize, doHalfSIFT, maxAngNum, th, addUpRight, vout);
  if (rc) return rc;
  *out = to_malloc(vout[0]);
  return (int)vout[0].size();
}

int modsx_reproject_regions(modsx_region *regs, int n, const double *H, int orig_w, int orig_h) {
  if (n < 0 || (n > 0 && !regs) || !H) { mx::set_error("modsx_reproject_regions: bad argument"); return MODSX_ERR_ARG; }
  return reproject_regions(regs, n, H, orig_w, orig_h);
}

int modsx_reproject_regions_touch_boundary(modsx_region *regs, int n, const double *H, int orig_w, int orig_h, double mrSize) {
  if (n < 0 || (n > 0 && !regs) || !H) { mx::set_error("modsx_reproject_regions_touch_boundary: bad argument"); return MODSX_ERR_ARG; }
  return reproject_regions_box(regs, n, H, orig_w, orig_h, mrSize);
}

int modsx_describe_regions(modsx_ctx *ctx, const modsx_image *img, const modsx_region *regs, int n, double mrSize,
                           int patchSize, int fast_extraction, int photoNorm, int desc_type, double maxBinValue,
                           float *desc) {
  NEED(ctx); NEED(img);
  if (n < 0 || (n > 0 && (!regs || !desc))) { mx::set_error("modsx_describe_regions: bad argument"); return MODSX_ERR_ARG; }
  if (desc_type < MODSX_DESC_SIFT || desc_type > MODSX_DESC_HALF_ROOT_SIFT) { mx::set_error("descriptor type"); return MODSX_ERR_ARG; }
  hipSetDevice(ctx->dev);
  std::vector<modsx_region> v[1];
  v[0].assign(regs, regs + n);
  const modsx_image *imgs[1] = {img};
  float *hosts[1] = {desc};
  int rc = describe_batch(ctx, imgs, 1, v, mrSize, patchSize, fast_extraction, photoNorm, desc_type, maxBinValue, hosts,
                          nullptr, nullptr);
  if (rc) return rc;
  return n;
}

int modsx_match_fginn(modsx_ctx *ctx, const float *desc1, int n1, const float *desc2, int n2, const double *pos2,
                      double ratio, double contradDist, int nn, modsx_tentative **out) {
  NEED(ctx); NEED(out);
  if (n1 < 0 || n2 < 0 || (n1 > 0 && !desc1) || (n2 > 0 && (!desc2 || !pos2))) { mx::set_error("modsx_match_fginn: bad argument"); return MODSX_ERR_ARG; }
  hipSetDevice(ctx->dev);
  std::vector<modsx_tentative> t;
  int rc = match_host_desc(ctx, desc1, n1, desc2, n2, pos2, ratio, contradDist, nn, t);
  if (rc) return rc;
  *out = to_malloc(t);
  return (int)t.size();
}

int modsx_duplicate_filtering(const double *pts, const double *key, int T, double r, int do_sort, int *order,
                              unsigned char *keep) {
  if (T < 0 || (T > 0 && (!pts || !order || !keep))) { mx::set_error("modsx_duplicate_filtering: bad argument"); return MODSX_ERR_ARG; }
  return duplicate_filtering(pts, key, T, r, do_sort, order, keep);
}

int modsx_ransac_h(const double *u, int len, double th, double conf, int max_sam, double *H, unsigned char *inl,
                   int *data_out, int oriented_constraint, int doSymCheck, unsigned seed, double *score_J) {
  if (!u || !H || !inl || !data_out || len < 4) { mx::set_error("modsx_ransac_h: bad argument"); return MODSX_ERR_ARG; }
  return ransac_h(u, len, th, conf, max_sam, H, inl, data_out, oriented_constraint, doSymCheck, seed, score_J);
}

int modsx_ransac_h_errtype(const double *u, int len, double th, double conf, int max_sam, double *H, unsigned char *inl,
                           int *data_out, int oriented_constraint, int doSymCheck, int error_type, unsigned seed, double *score_J) {
  if (!u || !H || !inl || !data_out || len < 4 || error_type < 0 || error_type > 2) { mx::set_error("modsx_ransac_h_errtype: bad argument"); return MODSX_ERR_ARG; }
  return ransac_h(u, len, th, conf, max_sam, H, inl, data_out, oriented_constraint, doSymCheck, seed, score_J, error_type);
}

int modsx_ransac_f(const double *u, int len, double th, double conf, int max_sam, int do_lo, unsigned inl_limit,
                   int error_type, int doSymCheck, unsigned seed, double *F, unsigned char *inl, int *data_out) {
  if (!u || !F || !inl || !data_out || len < 8) { mx::set_error("modsx_ransac_f: bad argument"); return MODSX_ERR_ARG; }
  return ransac_f(u, len, th, conf, max_sam, F, inl, data_out, do_lo, inl_limit, error_type, doSymCheck, seed);
}

int modsx_loransac_f(const double *pts, const double *laf1, const double *laf2, int T, double err_threshold,
                     double confidence, int max_samples, int localOptimization, double LAFCoef, int doSymmCheck,
                     int error_type, unsigned seed, double *F, unsigned char *inl, unsigned char *keep, int *data_out) {
  if (!pts || !laf1 || !laf2 || !F || !inl || !keep || !data_out || T < 0) {
    mx::set_error("modsx_loransac_f: bad argument");
    return MODSX_ERR_ARG;
  }
  return loransac_f(pts, laf1, laf2, T, err_threshold, confidence, max_samples, localOptimization, LAFCoef, doSymmCheck,
                    error_type, seed, F, inl, keep, data_out);
}

int modsx_loransac_h(const double *pts, const double *laf1, const double *laf2, int T, double err_threshold,
                     double confidence, int max_samples, int localOptimization, double HLAFCoef, int doSymmCheck,
                     unsigned seed, double *H, double *Hraw, unsigned char *inl, unsigned char *keep, int *data_out) {
  if (T < 0 || !H || !Hraw || !data_out || (T > 0 && (!pts || !laf1 || !laf2 || !inl || !keep))) {
    mx::set_error("modsx_loransac_h: bad argument");
    return MODSX_ERR_ARG;
  }
  return loransac_h(pts, laf1, laf2, T, err_threshold, confidence, max_samples, localOptimization, HLAFCoef, doSymmCheck,
                    seed, H, Hraw, inl, keep, data_out);
}

int modsx_loransac_h_errtype(const double *pts, const double *laf1, const double *laf2, int T, double err_threshold,
                             double confidence, int max_samples, int localOptimization, double HLAFCoef, int doSymmCheck,
                             int error_type, unsigned seed, double *H, double *Hraw, unsigned char *inl, unsigned char *keep,
                             int *data_out) {
  if (T < 0 || !H || !Hraw || !data_out || (T > 0 && (!pts || !laf1 || !laf2 || !inl || !keep)) || error_type < 0 || error_type > 2) {
    mx::set_error("modsx_loransac_h_errtype: bad argument");
    return MODSX_ERR_ARG;
  }
  return loransac_h(pts, laf1, laf2, T, err_threshold, confidence, max_samples, localOptimization, HLAFCoef, doSymmCheck,
                    seed, H, Hraw, inl, keep, data_out, error_type);
}

static void verify_timed(mx::VerifyTask &t, const modsx_pair_params &pp) {
  const auto v0 = std::chrono::steady_clock::now();
  mx::verify_tentatives(t.l1, t.l2, t.tents, pp, t.res);
  const long us = std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - v0).count();
  g_verifyUs += us;
  g_verifyPairs++;
  std::lock_guard<std::mutex> lk(g_verifyEachMu);
  g_verifyEachUs.push_back((int)us);
}
// the caller's current HIP device is its own: worker(0) of the batch calls runs on the caller's thread and sets the contexts'
struct DeviceGuard {
  int dev = -1;
  DeviceGuard() { if (hipGetDevice(&dev) != hipSuccess) dev = -1; }
  ~DeviceGuard() { if (dev >= 0) hipSetDevice(dev); }
};

int modsx_match_pair(modsx_ctx *ctx, const modsx_image *img1, const modsx_image *img2, const modsx_pair_params *par,
                     modsx_pair_result *res) {
  NEED(ctx); NEED(img1); NEED(img2); NEED(par); NEED(res);
  hipSetDevice(ctx->dev);
  return match_pair(ctx, img1, img2, *par, res);
}

int modsx_match_pairs(modsx_ctx *const *ctxs, int n_ctx, const modsx_image *const *imgs1,
                      const modsx_image *const *imgs2, int n_pairs, const modsx_pair_params *par,
                      modsx_pair_result *results) {
  NEED(ctxs); NEED(par); NEED(results);
  if (n_ctx <= 0 || n_pairs < 0 || (n_pairs > 0 && (!imgs1 || !imgs2))) { mx::set_error("modsx_match_pairs: bad argument"); return MODSX_ERR_ARG; }
  for (int i = 0; i < n_ctx; i++) NEED(ctxs[i]);
  for (int i = 0; i < n_pairs; i++) memset(&results[i], 0, sizeof results[i]);   // no result owns arrays before its group ran
  std::atomic<int> next(0), failed(0);
  std::string firstErr;
  std::mutex *mu = new std::mutex();
  // every context takes up to MAXB / 2 pairs at a time and runs their 2g images as one batch; the group shrinks so
  // that all contexts get work when the batch is small
  int group = (n_pairs + n_ctx - 1) / n_ctx;
  if (group > mx::PAIR_GROUP) group = mx::PAIR_GROUP;
  if (group < 1) group = 1;
  // Verification (DuplicateFiltering + LO-RANSAC, ~1.4 ms of host time per pair against ~0.9 ms of device time) runs on
  // helper threads, one per working context: the context's own thread goes straight on to the next group, so its
  // stream is idle only while the host prepares job tables, not while it runs RANSAC.
  struct Queue {
    std::mutex m;
    std::condition_variable cv;
    std::deque<mx::VerifyTask> q;
    bool done = false;
  } vq;
  const modsx_pair_params pp = *par;
  auto helper = [&]() {
    ThreadCpu cpu(g_cpuHelperUs);
    for (;;) {
      mx::VerifyTask t;
      {
        std::unique_lock<std::mutex> lk(vq.m);
        vq.cv.wait(lk, [&] { return vq.done || !vq.q.empty(); });
        if (vq.q.empty()) return;
        t = std::move(vq.q.front());
        vq.q.pop_front();
      }
      hipSetDevice(t.dev);     // a new thread starts on device 0: the verifier's device-side loops belong on the producing context's GPU
      verify_timed(t, pp);
    }
  };
  auto worker = [&](int w) {
    modsx_ctx *c = ctxs[w];
    ThreadCpu cpu(g_cpuWorkerUs);
    hipSetDevice(c->dev);
    std::vector<mx::VerifyTask> tasks;
    for (;;) {
      if (failed.load()) break;   // another group failed: the batch is lost, stop feeding the stream
      int i = next.fetch_add(group);
      if (i >= n_pairs) break;
      const int g = (n_pairs - i) < group ? (n_pairs - i) : group;
      tasks.clear();
      int rc = match_pair_group(c, imgs1 + i, imgs2 + i, g, pp, &results[i], &tasks);
      if (rc) {
        std::lock_guard<std::mutex> g(*mu);
        if (!failed.exchange(rc)) firstErr = mx::last_error();
      }
      if (!tasks.empty()) {
        { std::lock_guard<std::mutex> lk(vq.m); for (auto &t : tasks) vq.q.push_back(std::move(t)); }
        vq.cv.notify_all();
      }
    }
    // out of pairs: the context's thread helps the verification helpers with what is still queued (the tail of a batch call is
    // the verification of its last pairs, with the device already idle)
    for (;;) {
      mx::VerifyTask t;
      {
        std::lock_guard<std::mutex> lk(vq.m);
        if (vq.q.empty()) break;
        t = std::move(vq.q.front());
        vq.q.pop_front();
      }
      hipSetDevice(t.dev);
      verify_timed(t, pp);
    }
  };
  std::vector<std::thread> th, hth;
  const int ngroups = (n_pairs + group - 1) / group;
  const int nw = n_ctx < ngroups ? n_ctx : ngroups;
  g_verifyUs = 0; g_verifyPairs = 0; g_verifyThreads = nw; g_cpuWorkerUs = 0; g_cpuHelperUs = 0;
  { std::lock_guard<std::mutex> lk(g_verifyEachMu); g_verifyEachUs.clear(); }
  DeviceGuard restoreCallerDevice;
  for (int w = 0; w < verify_helpers(nw); w++) hth.emplace_back(helper);
  for (int w = 1; w < nw; w++) th.emplace_back(worker, w);
  if (nw > 0) worker(0);
  for (auto &t : th) t.join();
  { std::lock_guard<std::mutex> lk(vq.m); vq.done = true; }
  vq.cv.notify_all();
  for (auto &t : hth) t.join();
  delete mu;
  if (failed.load()) {
    // a failed batch returns no results: release what the successful groups already own
    for (int i = 0; i < n_pairs; i++) modsx_pair_result_release(&results[i]);
    mx::set_error(firstErr);
    return failed.load();
  }
  return n_pairs;
}

// modsx_match_pairs for multi-view pairs: every context takes one pair at a time (the pairs come off a shared counter), runs
// the view loop of both images and the match, and hands the tentatives to a helper thread for DuplicateFiltering +
// LO-RANSAC while its own thread goes on to the next pair -- a 31-view pair carries ~12 k tentatives and ~10 ms of host
// verification, during which the context's stream would otherwise sit idle.
int modsx_match_pairs_views(modsx_ctx *const *ctxs, int n_ctx, const modsx_image *const *imgs1, const modsx_image *const *imgs2,
                            int n_pairs, const modsx_view *views, int n_views, const modsx_pair_params *par,
                            modsx_pair_result *results) {
  NEED(ctxs); NEED(par); NEED(results); NEED(views);
  if (n_ctx <= 0 || n_views <= 0 || n_pairs < 0 || (n_pairs > 0 && (!imgs1 || !imgs2))) {
    mx::set_error("modsx_match_pairs_views: bad argument");
    return MODSX_ERR_ARG;
  }
  for (int i = 0; i < n_ctx; i++) NEED(ctxs[i]);
  for (int i = 0; i < n_pairs; i++) memset(&results[i], 0, sizeof results[i]);
  std::atomic<int> next(0), failed(0);
  std::string firstErr;
  std::mutex mu;
  struct Queue {
    std::mutex m;
    std::condition_variable cv;
    std::deque<mx::VerifyTask> q;
    bool done = false;
  } vq;
  const modsx_pair_params pp = *par;
  auto helper = [&]() {
    ThreadCpu cpu(g_cpuHelperUs);
    for (;;) {
      mx::VerifyTask t;
      {
        std::unique_lock<std::mutex> lk(vq.m);
        vq.cv.wait(lk, [&] { return vq.done || !vq.q.empty(); });
        if (vq.q.empty()) return;
        t = std::move(vq.q.front());
        vq.q.pop_front();
      }
      hipSetDevice(t.dev);     // a new thread starts on device 0: the verifier's device-side loops belong on the producing context's GPU
      verify_timed(t, pp);
    }
  };
  auto worker = [&](int w) {
    modsx_ctx *c = ctxs[w];
    ThreadCpu cpu(g_cpuWorkerUs);
    hipSetDevice(c->dev);
    for (;;) {
      if (failed.load()) break;
      const int i = next.fetch_add(1);
      if (i >= n_pairs) break;
      mx::VerifyTask task;
      const int rc = mx::match_pair_views(c, imgs1[i], imgs2[i], views, n_views, pp, &results[i], &task);
      if (rc) {
        std::lock_guard<std::mutex> g(mu);
        if (!failed.exchange(rc)) firstErr = mx::last_error();
        continue;
      }
      { std::lock_guard<std::mutex> lk(vq.m); vq.q.push_back(std::move(task)); }
      vq.cv.notify_one();
    }
    for (;;) {      // out of pairs: help with the verifications still queued (see modsx_match_pairs)
      mx::VerifyTask t;
      {
        std::lock_guard<std::mutex> lk(vq.m);
        if (vq.q.empty()) break;
        t = std::move(vq.q.front());
        vq.q.pop_front();
      }
      hipSetDevice(t.dev);
      verify_timed(t, pp);
    }
  };
  std::vector<std::thread> th, hth;
  const int nw = n_ctx < n_pairs ? n_ctx : n_pairs;
  g_verifyUs = 0; g_verifyPairs = 0; g_verifyThreads = nw; g_cpuWorkerUs = 0; g_cpuHelperUs = 0;
  { std::lock_guard<std::mutex> lk(g_verifyEachMu); g_verifyEachUs.clear(); }
  DeviceGuard restoreCallerDevice;
  for (int w = 0; w < verify_helpers(nw); w++) hth.emplace_back(helper);
  for (int w = 1; w < nw; w++) th.emplace_back(worker, w);
  if (nw > 0) worker(0);
  for (auto &t : th) t.join();
  { std::lock_guard<std::mutex> lk(vq.m); vq.done = true; }
  vq.cv.notify_all();
  for (auto &t : hth) t.join();
  if (failed.load()) {
    for (int i = 0; i < n_pairs; i++) modsx_pair_result_release(&results[i]);
    mx::set_error(firstErr);
    return failed.load();
  }
  return n_pairs;
}

void modsx_pair_result_release(modsx_pair_result *res) {
  if (!res) return;
  free(res->tentatives); free(res->ransac_inlier); free(res->verified);
  res->tentatives = nullptr; res->ransac_inlier = nullptr; res->verified = nullptr;
}

int modsx_last_batch_verify(double *sum_ms, int *pairs, int *threads) {
  if (sum_ms) *sum_ms = g_verifyUs.load() / 1000.0;
  if (pairs) *pairs = g_verifyPairs.load();
  if (threads) *threads = g_verifyThreads.load();
  return MODSX_OK;
}

// verification time (DuplicateFiltering + LO-RANSAC + the LAF checks, wall of the verifying thread) of every pair of the last batch call
extern "C" __attribute__((visibility("default"))) int modsx_debug_last_batch_verify_each(double *ms, int cap) {
  std::lock_guard<std::mutex> lk(g_verifyEachMu);
  const int n = (int)g_verifyEachUs.size();
  for (int i = 0; i < n && i < cap; i++) ms[i] = g_verifyEachUs[i] / 1000.0;
  return n;
}

// CPU seconds (thread CPU clocks, not wall) the last batch call spent in the contexts' own threads and in its verification helpers
extern "C" __attribute__((visibility("default"))) int modsx_debug_last_batch_cpu(double *worker_s, double *helper_s) {
  if (worker_s) *worker_s = g_cpuWorkerUs.load() / 1e6;
  if (helper_s) *helper_s = g_cpuHelperUs.load() / 1e6;
  return MODSX_OK;
}

// 1: stage boundaries wait through the runtime (hipStreamSynchronize) right now, 0: through the context's flag word (engine.hip, MODSX_HOST_WAIT;
// `auto` follows the process' CPU load)
extern "C" __attribute__((visibility("default"))) int modsx_debug_host_wait_runtime() { return mx::host_wait_runtime() ? 1 : 0; }

int modsx_last_timings(modsx_ctx *ctx, double *ms6) {
  NEED(ctx); NEED(ms6);
  for (int i = 0; i < 6; i++) ms6[i] = ctx->timings[i];
  return MODSX_OK;
}

int modsx_set_vs_pars(const double *scale_set, int ns, const double *tilt_set, int nt, double phi_base,
                      double InitSigma, int doBlur, modsx_view *par, int cap, modsx_view *prev, int *nprev,
                      int cap_prev) {
  if (ns < 0 || nt < 0 || !par || !prev || !nprev || (ns > 0 && !scale_set) || (nt > 0 && !tilt_set)) {
    mx::set_error("modsx_set_vs_pars: bad argument");
    return MODSX_ERR_ARG;
  }
  return set_vs_pars(scale_set, ns, tilt_set, nt, phi_base, InitSigma, doBlur, par, cap, prev, nprev, cap_prev);
}

modsx_image *modsx_synth_view(modsx_ctx *ctx, const modsx_image *gray, const modsx_view *view, double *H9,
                              int *is_identity) {
  if (!ctx || !gray || !view || !H9 || !is_identity) { mx::set_error("modsx_synth_view: null argument"); return nullptr; }
  hipSetDevice(ctx->dev);
  modsx_image *out = nullptr;
  if (synth_view(ctx, gray, *view, &out, H9, is_identity) != MODSX_OK) return nullptr;
  return out;
}

int modsx_detect_describe_views(modsx_ctx *ctx, const modsx_image *img, const modsx_view *views, int nviews,
                                const modsx_pair_params *par, int view_begin, int view_step, modsx_region **regs,
                                float **desc, void *dev_desc_u8, long dev_cap, int *view_counts) {
  NEED(ctx); NEED(img); NEED(views); NEED(par); NEED(regs);
  if (nviews <= 0 || view_begin < 0) { mx::set_error("modsx_detect_describe_views: bad argument"); return MODSX_ERR_ARG; }
  hipSetDevice(ctx->dev);
  std::vector<modsx_region> r;
  size_t cap = dev_desc_u8 ? (size_t)dev_cap : ((size_t)1 << 16);
  std::vector<int> counts(nviews, 0);
  int rc;
  for (;;) {
    if (!ctx->descAllF[0].ensure(cap * 512)) return MODSX_ERR_NOMEM;
    uint8_t *du8 = (uint8_t *)dev_desc_u8;
    if (!du8) { if (!ctx->descAllU8[0].ensure(cap * 128)) return MODSX_ERR_NOMEM; du8 = (uint8_t *)ctx->descAllU8[0].p; }
    rc = detect_describe_views(ctx, img, views, nviews, *par, view_begin, view_step, r, (float *)ctx->descAllF[0].p, du8,
                               cap, nullptr, counts.data());
    if (rc == MODSX_ERR_CAPACITY && !dev_desc_u8 && cap < ((size_t)1 << 24)) { cap *= 4; continue; }
    break;
  }
  if (rc) return rc;
  if (view_counts) memcpy(view_counts, counts.data(), sizeof(int) * nviews);
  if (view_step <= 1 && view_begin == 0) rebase_ids(r, counts.data(), nviews, 0);
  if (desc) {
    *desc = (float *)malloc(std::max<size_t>(1, r.size()) * 512);
    if (!r.empty()) {
      MX_HIP(hipMemcpyAsync(*desc, ctx->descAllF[0].p, r.size() * 512, hipMemcpyDeviceToHost, ctx->stream));
      MX_HIP(hipStreamSynchronize(ctx->stream));
    }
  }
  *regs = to_malloc(r);
  return (int)r.size();
}

int modsx_match_fginn_device(modsx_ctx *ctx, const void *dev_desc1_u8, int n1, const void *dev_desc2_u8, int n2,
                             const double *pos2, double ratio, double contradDist, int nn, modsx_tentative **out) {
  NEED(ctx); NEED(out);
  if (n1 < 0 || n2 < 0 || (n1 > 0 && !dev_desc1_u8) || (n2 > 0 && (!dev_desc2_u8 || !pos2))) {
    mx::set_error("modsx_match_fginn_device: bad argument");
    return MODSX_ERR_ARG;
  }
  hipSetDevice(ctx->dev);
  std::vector<modsx_tentative> t;
  int rc = match_device(ctx, (const uint8_t *)dev_desc1_u8, n1, (const uint8_t *)dev_desc2_u8, n2, pos2, ratio,
                        contradDist, nn, t);
  if (rc) return rc;
  *out = to_malloc(t);
  return (int)t.size();
}

int modsx_match_ladder(modsx_ctx *ctx, const modsx_image *img1, const modsx_image *img2,
                       const modsx_ladder_step *steps, int nsteps, int min_matches, const modsx_pair_params *par,
                       modsx_pair_result *res, int *steps_done) {
  NEED(ctx); NEED(img1); NEED(img2); NEED(steps); NEED(par); NEED(res);
  if (nsteps <= 0) { mx::set_error("modsx_match_ladder: nsteps"); return MODSX_ERR_ARG; }
  for (int s = 0; s < nsteps; s++)
    if (!steps[s].views || steps[s].nviews <= 0) { mx::set_error("modsx_match_ladder: empty step"); return MODSX_ERR_ARG; }
  hipSetDevice(ctx->dev);
  return match_ladder(ctx, img1, img2, steps, nsteps, min_matches, *par, res, steps_done);
}

int modsx_match_pair_views(modsx_ctx *ctx, const modsx_image *img1, const modsx_image *img2,
                           const modsx_view *views, int nviews, const modsx_pair_params *par,
                           modsx_pair_result *res) {
  NEED(ctx); NEED(img1); NEED(img2); NEED(views); NEED(par); NEED(res);
  if (nviews <= 0) { mx::set_error("modsx_match_pair_views: nviews"); return MODSX_ERR_ARG; }
  hipSetDevice(ctx->dev);
  return match_pair_views(ctx, img1, img2, views, nviews, *par, res);
}

void modsx_default_mser_params(modsx_mser_params *p) {
  // build/config_iter_mods_cviu.ini:4-12; extremaParams.h:66-82
  memset(p, 0, sizeof *p);
  p->min_size = 30; p->max_area = 0.05; p->min_margin = 8; p->relative = 0;
  p->mode = MODSX_FIXED_TH; p->reg_number = 500; p->rel_threshold = -1; p->rel_reg_number = -1;
}

int modsx_detect_msers(modsx_ctx *ctx, const modsx_image *img, const modsx_mser_params *par, double tilt, double zoom,
                       modsx_keypoint **out) {
  NEED(ctx); NEED(img); NEED(par); NEED(out);
  if (par->min_size < 1 || !(par->max_area > 0)) { mx::set_error("modsx_detect_msers: parameters"); return MODSX_ERR_ARG; }
  hipSetDevice(ctx->dev);
  const size_t n = (size_t)img->rows * img->cols;
  if (!ctx->misc.ensure(n + 16)) return MODSX_ERR_NOMEM;
  launch_trunc_u8(ctx->stream, img->d, (uint8_t *)ctx->misc.p, n);
  std::vector<uint8_t> host(n);
  MX_HIP(hipMemcpyAsync(host.data(), ctx->misc.p, n, hipMemcpyDeviceToHost, ctx->stream));
  MX_HIP(hipStreamSynchronize(ctx->stream));
  std::vector<modsx_keypoint> k;
  int rc = detect_msers_host(host.data(), img->rows, img->cols, *par, tilt, zoom, k);
  if (rc) return rc;
  *out = (modsx_keypoint *)malloc(sizeof(modsx_keypoint) * std::max<size_t>(1, k.size()));
  if (!*out) { mx::set_error("out of memory"); return MODSX_ERR_NOMEM; }
  if (!k.empty()) memcpy(*out, k.data(), sizeof(modsx_keypoint) * k.size());
  return (int)k.size();
}

int modsx_detect_msers_u8(const unsigned char *gray, int rows, int cols, const modsx_mser_params *par, double tilt,
                          double zoom, modsx_keypoint **out) {
  if (!gray || !par || !out || rows <= 0 || cols <= 0 || par->min_size < 1 || !(par->max_area > 0)) {
    mx::set_error("modsx_detect_msers_u8: bad argument");
    return MODSX_ERR_ARG;
  }
  std::vector<modsx_keypoint> k;
  int rc = detect_msers_host(gray, rows, cols, *par, tilt, zoom, k);
  if (rc) return rc;
  *out = (modsx_keypoint *)malloc(sizeof(modsx_keypoint) * std::max<size_t>(1, k.size()));
  if (!*out) { mx::set_error("out of memory"); return MODSX_ERR_NOMEM; }
  if (!k.empty()) memcpy(*out, k.data(), sizeof(modsx_keypoint) * k.size());
  return (int)k.size();
}

// The host half of the MSER view loop without a device (test aid): DetectMSERs of n u8 views as the engine runs them -- 2 n (view,
// polarity) tasks on the host pool that share a view's bin sort.  counts[i] = keys of view i; *out = the views' keys one after the
// other (modsx_free).  Returns the total or a negative status.
extern "C" __attribute__((visibility("default"))) int modsx_debug_msers_views_u8(const unsigned char *const *gray, const int *rows, const int *cols,
                                                                                  int n, const modsx_mser_params *par, const double *tilts,
                                                                                  const double *zooms, int *counts, modsx_keypoint **out) {
  if (!gray || !rows || !cols || n <= 0 || n > mx::MAXB || !par || !tilts || !zooms || !counts || !out) { mx::set_error("modsx_debug_msers_views_u8: bad argument"); return MODSX_ERR_ARG; }
  std::vector<modsx_keypoint> k[mx::MAXB];
  int rc = mx::detect_msers_views(gray, rows, cols, n, *par, tilts, zooms, k);
  if (rc) return rc;
  size_t total = 0;
  for (int i = 0; i < n; i++) { counts[i] = (int)k[i].size(); total += k[i].size(); }
  *out = (modsx_keypoint *)malloc(sizeof(modsx_keypoint) * std::max<size_t>(1, total));
  if (!*out) { mx::set_error("out of memory"); return MODSX_ERR_NOMEM; }
  size_t at = 0;
  for (int i = 0; i < n; i++) { if (!k[i].empty()) memcpy(*out + at, k[i].data(), sizeof(modsx_keypoint) * k[i].size()); at += k[i].size(); }
  return (int)total;
}

int modsx_save_regions(const char *path, const modsx_region_class *classes, int nclasses) {
  if (!path || !classes || nclasses <= 0) { mx::set_error("modsx_save_regions: bad argument"); return MODSX_ERR_ARG; }
  for (int i = 0; i < nclasses; i++) {
    const modsx_region_class &c = classes[i];
    if (!c.det_name || !c.desc_name || c.n < 0 || c.dim < 0 || c.stride < c.dim || (c.n > 0 && (!c.regs || (c.dim > 0 && !c.desc)))) {
      mx::set_error("modsx_save_regions: bad class");
      return MODSX_ERR_ARG;
    }
  }
  return save_regions(path, classes, nclasses);
}

int modsx_load_regions(const char *path, const char *det_name, const char *desc_name, modsx_region **regs, float **desc,
                       int *dim, char *found_det, char *found_desc) {
  if (!path || !regs || !desc || !dim) { mx::set_error("modsx_load_regions: bad argument"); return MODSX_ERR_ARG; }
  std::vector<modsx_region> r;
  std::vector<float> d;
  std::string fd, fs;
  int rc;
  try { rc = load_regions(path, det_name, desc_name, r, d, dim, &fd, &fs); }
  catch (const std::exception &e) { mx::set_error(std::string("modsx_load_regions: ") + e.what()); return MODSX_ERR_NOMEM; }
  if (rc) return rc;
  *regs = (modsx_region *)malloc(sizeof(modsx_region) * std::max<size_t>(1, r.size()));
  *desc = (float *)malloc(sizeof(float) * std::max<size_t>(1, d.size()));
  if (!*regs || !*desc) { free(*regs); free(*desc); mx::set_error("out of memory"); return MODSX_ERR_NOMEM; }
  if (!r.empty()) memcpy(*regs, r.data(), sizeof(modsx_region) * r.size());
  if (!d.empty()) memcpy(*desc, d.data(), sizeof(float) * d.size());
  if (found_det) { strncpy(found_det, fd.c_str(), 63); found_det[63] = 0; }
  if (found_desc) { strncpy(found_desc, fs.c_str(), 63); found_desc[63] = 0; }
  return (int)r.size();
}

int modsx_profile(modsx_ctx *ctx, int enable) {
  NEED(ctx);
  prof_reset(ctx, enable != 0);
  return MODSX_OK;
}

int modsx_kernel_stats(modsx_ctx *ctx, double *ms, double *work, long *launches, int n) {
  NEED(ctx); NEED(ms); NEED(work); NEED(launches);
  prof_collect(ctx);
  for (int i = 0; i < n && i < K_NCLASS; i++) { ms[i] = ctx->prof.ms[i]; work[i] = ctx->prof.work[i]; launches[i] = ctx->prof.launches[i]; }
  return K_NCLASS;
}

}  // extern "C"
