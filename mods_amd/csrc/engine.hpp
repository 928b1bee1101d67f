// engine.hpp -- internal types of libmodsx (device job descriptors, context, host helpers).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <string>
#include <vector>
#include "../../include/modsx.h"
#include "kmath.hpp"

namespace mx {

// ---- device job descriptors (passed by value in kernel arguments) ----------------------
#ifndef MODSX_MAXB
#define MODSX_MAXB 32
#endif
constexpr int MAXB = MODSX_MAXB;   // images (views) per batched launch set: 8 / 16 / 32 measured 150 / 157 / 159 pairs/s at 31 views
constexpr int PAIR_GROUP = 4;      // identity-view pairs per launch set of modsx_match_pairs (8 images; 16 measured slower)
constexpr int NMS_MAXJ = 1024; // (image, octave, level) jobs per NMS launch (flushed when full)
constexpr int MAX_TAPS = 17;   // pyramid kernels: ksize <= 17

struct BlurJob {
  const float *src;
  float *blur;   // may alias nothing; nullptr for k_hessian
  float *resp;   // nullptr = no response
  int rows, cols;
  float norm;    // sigma^2 passed to HessianResponse
  int pad;
};
struct BlurBatch {
  int n;                 // ksize
  float k[MAX_TAPS];
  BlurJob j[MAXB];
  int nj;                // jobs in use; tile0[i] = first tile of job i in the flat tile list of the launch (filled by the launcher)
  int tile0[MAXB + 1];
};
struct ResizeJob {
  const float *src;
  float *dst;
  float *resp;   // not null: the Hessian response of the resized level is written too (norm = its sigma^2)
  int srows, scols, drows, dcols;
  float norm;
  int pad;
};
struct ResizeBatch { ResizeJob j[MAXB]; int nj; int tile0[MAXB + 1]; };
struct NmsJob {
  const float *low, *cur, *high, *blur;
  int rows, cols, img, octave, level, pad;
};
struct NmsBatch {   // thresholds of one scan; the (image, octave, level) jobs and their tile prefix live in device memory
  float posTh, negTh, finalTh;
  int border;
  double edgeScoreThreshold;
  int detType, pad;      // getPointType (pyramid.cpp:66-130): Hessian dark / bright / saddle, DoG 10 / 11, Harris 30 / 31
};
struct Candidate {
  int img, octave, level, type;
  int r0, c0, r, c;
  float b0, b1, b2, val;
};

// Baumberg: one wave per scale-space keypoint
struct AffJob {
  const float *blur;  // prevBlur level
  int rows, cols;
  float x, y, s, pixelDistance;
};
struct AffOut { float u11, u12, u21, u22; int ok; int iters; };

// orientation: one wave per region
struct OriJob {
  int img;
  float x, y, a11, a12, a21, a22;  // A * curr_sc, f32 (synth-detection.cpp:892-898)
};
// orientation results: (1 + maxA) 32-bit words per region -- the angle count, then the angles; maxA = min(maxAngles, 18), the
// most strict local maxima a 36-bin circular histogram can have (so "all peaks", maxAngles = -1, loses none)
constexpr int ORI_MAX_PEAKS = 18;

// describe
struct DescJob {
  int img;
  int P;              // patchImageSize (+2 when smoothed), 0 = direct mode
  int ksize;          // blur kernel size for 1.5*imageToPatchScale
  int tapOfs;         // offset into the tap table
  float x, y, a11, a12, a21, a22;   // direct mode: A * imageToPatchScale
  float i2p;          // imageToPatchScale
  int NC;             // number of window columns (= rows) the 41x41 resampling reads (<= 82)
  int needOfs;        // offset into the int table: NC needed indices, then 41 x {idx of x0, idx of x0+1, valid}
  int coordOfs;       // offset into the float table: 41 sample coordinates WX_i (= WY_j)
  int touch;          // interpolate()'s border branch for the 41x41 resampling
  int outIdx;         // index of the region in its image's descriptor buffers
  int rows0;          // window rows per workgroup of the LDS row filter, 0 = this job goes through k_patch_blur
  int ro1;            // needed rows per workgroup of the LDS column filter, 0 = k_patch_blur, -1 = the fused sampling kernel
                      // filters the columns too (small window: whole row tile + row-filtered block fit its LDS)
  unsigned long long scratchOfs;    // float offset of this region's P x P window (arena A); for windows that take the fused
                                    // sample + row-filter kernel: float2 offset of its P row starts
  unsigned long long rowOfs;        // float offset of its P x NC row-filtered block (arena B)
  unsigned long long gridOfs;       // float offset of its NC x NC blurred grid (arena C)
};
struct BlurTile {     // one workgroup of the LDS blur kernels
  unsigned long long srcOfs, dstOfs;   // float offsets: first input row / first output of the tile
  int P, NC, n, tapOfs, needOfs;
  int job, pad;                        // index of the tile's DescJob (the fused sample + row-filter kernel reads its affine frame)
  int first, count, lo, span, magic;  // rows pass: window rows [first, first+count); columns pass: needed rows, parked source rows [lo, lo+span); magic = ceil(2^20 / ceil(NC / 2))
};
struct ImgRef { const float *d; int rows, cols, pad; };
// per image of the batch: [n][128] f32 and u8 descriptors of the step's first descriptor class, u8 of up to three more
struct DescOut { float *f[MAXB]; uint8_t *u8[MAXB]; uint8_t *u8x[3][MAXB]; };

// view synthesis
struct WarpJob {
  const float *src;
  float *dst;
  int srows, scols, drows, dcols;
  double M[6];   // inverse map (dst -> src), already inverted on the host in f64
  float cval;
  int pad;
};

// one non-identity view of a batched synthesis (kernels_views.hip): src -R-> rot -blur (through tmp)-> rot -W-> dst
struct ViewJob {
  const float *src;
  float *rot, *tmp, *dst;
  int srows, scols, rrows, rcols, drows, dcols;
  int kx, ky, tapOfs, doBlur;
  int tileA, tileB;   // first 64 x 4 tile of this view in the rotated-image / output-image tile lists
  int tileF, fused;   // fused rotate + blur: first VF_TW x VF_TH tile of this view (views with a blur whose halo fits)
  double R[6], W[6];  // inverse maps of the two cv::warpAffine calls (f64, inverted on the host)
};

// matching
struct MatchRow {     // per-query result of the device matcher
  int t0, t1, nless, nbad;
  int tj;             // (tj, dj) is one 8-byte word (dj the high half): k_match_resolve takes the minimum of (distance, train)
  float dj;           // over the splits with a 64-bit atomic -- positive floats order like their bits
  float d0, d1;
};
static_assert(sizeof(MatchRow) == 32, "match rows are 32 bytes");

// ---- host side ---------------------------------------------------------------------------
struct DevBuf {
  void *p = nullptr;
  size_t cap = 0;
  bool ensure(size_t bytes);
  void release();
};
struct PinBuf {
  void *p = nullptr;
  size_t cap = 0;
  bool ensure(size_t bytes);
  void release();
};

struct Octave {
  int rows, cols;
  float pixelDistance;
  float *blur[8];
  float *resp[8];
};
struct Pyramid {
  int nOct = 0;
  Octave oct[24];
  DevBuf store;
};

void set_error(const std::string &s);
// How many contexts are inside a call that keeps the device busy (a pair, a view set, a matching problem).  One: the call has the GPU
// to itself and may launch shapes that take whole CUs (k_match_sweep1's fat workgroups: -4 % alone); more: its launches share the
// CUs with the other contexts' and stay good neighbours (the fat shape waits for a drained CU there: -0.4 % pairs/s under 16 streams).
bool gpu_shared();
struct CtxBusy { modsx_ctx *c; explicit CtxBusy(modsx_ctx *c); ~CtxBusy(); CtxBusy(const CtxBusy &) = delete; CtxBusy &operator=(const CtxBusy &) = delete; };
// stage boundaries without the runtime's copy / wait calls (engine.hip): a kernel copies between device and PINNED host memory,
// a one-lane kernel writes a sequence number into the context's pinned flag word, the host thread naps until it sees it
hipError_t ctx_copy(modsx_ctx *c, void *dst, const void *src, size_t bytes, hipMemcpyKind kind);
hipError_t ctx_sync(modsx_ctx *c);
unsigned ctx_mark(modsx_ctx *c);
bool ctx_mark_reached(modsx_ctx *c, unsigned seq);
hipError_t ctx_wait_mark(modsx_ctx *c, unsigned seq);
bool host_wait_runtime();
int host_cpus_per_rank();      // mser.cpp: CPUs this process may use (cgroup allowance / local ranks)
#define MX_HIP(expr)                                                                          \
  do {                                                                                        \
    hipError_t e_ = (expr);                                                                   \
    if (e_ != hipSuccess) {                                                                   \
      mx::set_error(std::string(#expr) + ": " + hipGetErrorString(e_));                       \
      return MODSX_ERR_DEVICE;                                                                \
    }                                                                                         \
  } while (0)

// tables.cpp (host precomputation, reference semantics)
std::vector<float> gaussian_kernel(int n, double sigma);
int blur_ksize(float sigma);
void gauss_mask(float *mask, int size);
void circular_gauss_mask(float *mask, int size, float sigma);
const double *atan_lut_host();
inline bool check_borders_host(int w, int h, float ofsx, float ofsy, float a11, float a12, float a21, float a22, int rw, int rh) {
  return check_borders(w, h, ofsx, ofsy, a11, a12, a21, a22, rw, rh);     // kmath.hpp (host build of the kernels' expression)
}
bool invert3(const double *S, double *t);
void rectify(double &a11, double &a12, double &a21, double &a22);

// kernel launchers
void launch_blur_hess(hipStream_t s, const BlurBatch &b, int nj, int maxRows, int maxCols);
void launch_hessian(hipStream_t s, const BlurBatch &b, int nj, int maxRows, int maxCols);
void launch_resize_half(hipStream_t s, const ResizeBatch &b, int nj, int maxRows, int maxCols);
void launch_nms(hipStream_t s, const NmsBatch &b, const NmsJob *jobs, const int *tilePrefix, const int *tileJob, int nj,
                int nTiles, int4 *queue, unsigned *qcount, unsigned qcap, Candidate *out, unsigned *counter, unsigned cap,
                const int *octFirst = nullptr, int levelsPerOctave = 0);
size_t cand_sort_temp_bytes(unsigned nsort);
int launch_cand_order(hipStream_t s, const Candidate *cand, const unsigned *counter, unsigned nsort, unsigned long long *keys,
                      unsigned long long *keys2, unsigned *idx, unsigned *idx2, void *temp, size_t tempBytes, unsigned long long *tabKey,
                      unsigned *tabRank, unsigned tabSize, unsigned *slotOf, Candidate *out, unsigned *outCount);
constexpr int NMS_QUEUES = 64;      // sub-queues of the NMS extremum queue; counter k lives at counter[32 * (k + 1)]
constexpr int NMS_TILE_ROWS = 16;   // rows of a 64-column NMS tile (kernels_pyramid.hip: NMS_ROWS)
void launch_gray(hipStream_t s, const void *src, float *dst, size_t n, int channels, int dtype);
void launch_baumberg(hipStream_t s, const AffJob *jobs, AffOut *out, int n, const float *mask, int W, int maxIter,
                     float convTh, float affInitialSigma);
constexpr int ATAN_CASES = 2048 + 64;   // 8 x 256 (sign / octant bits, table index) values of atan2LUTff's angle + the special case (entry 2048), padded
constexpr int ORI_NV = 1344;   // entries of the orientation kernel's voting-pixel list (1245 under the mask, padded to 64 lanes x 21)
void launch_orientation(hipStream_t s, const OriJob *jobs, float *out, int n, const ImgRef *imgs,
                        const unsigned short *maskIdx, const float *maskW,
                        const unsigned char *binTab, int doHalf, double th, int maxAngles);
void launch_trunc_u8(hipStream_t s, const float *src, uint8_t *dst, size_t n);
size_t mser_sort_blocks(const int *rows, int n);
void launch_mser_sort(hipStream_t s, const uint8_t *const *u8, const int *rows, const int *cols, const size_t *ordOfs, int n, int *blockHist,
                      int *start, int *order);
void launch_expand_blur_tiles(hipStream_t s, const DescJob *jobs, const int *prefixRows, const int *prefixCols, int nJobs,
                              const int *needTab, BlurTile *tilesRows, BlurTile *tilesCols, float2 *rowStarts);
void launch_sample_rows(hipStream_t s, const DescJob *jobs, const BlurTile *tiles, int nTiles, const ImgRef *imgs, const float *taps,
                        const int *needTab, float *dst, const float2 *rowStarts, float *dstGrid);
void launch_blur_cols(hipStream_t s, const BlurTile *tiles, int nTiles, const float *taps, const int *needTab, const float *src,
                      float *dst);
void launch_expand_tiles(hipStream_t s, const int *prefix, int nJobs, int *tileJob);
void launch_patch_sample(hipStream_t s, const DescJob *jobs, const int *tilePrefix, const int *tileJob, int nTiles,
                         const ImgRef *imgs, float *scratch);
void launch_patch_blur(hipStream_t s, const DescJob *jobs, const int *tilePrefix, const int *tileJob, int nTiles,
                       const float *taps, const int *needTab, const float *src, float *dst, int pass);
void launch_describe(hipStream_t s, const DescJob *jobs, int n, const ImgRef *imgs, const float *grid,
                     const int *needTab, const float *coordTab,
                     const float *mask, const unsigned short *maskIdx, int nmask, const float *oTab, const int *bins,
                     const double *wts, int photoNorm, int descTypes, int nOut, double maxBin, const DescOut &outs);
void launch_warp_affine(hipStream_t s, const WarpJob &jb);
void launch_blur_pass(hipStream_t s, const float *src, float *dst, int rows, int cols, const float *taps, int n, int pass, int border = 0);
void launch_sub(hipStream_t s, const float *a, const float *b, float *o, size_t n);
void launch_grad_products(hipStream_t s, const float *img, int rows, int cols, float *xx, float *yy, float *xy);
void launch_harris_combine(hipStream_t s, const float *bxx, const float *byy, const float *bxy, float sigmasq, float *o, size_t n);
void launch_views_warp(hipStream_t s, const ViewJob *jobs, int n, int tiles, int stage);
void launch_views_blur(hipStream_t s, const ViewJob *jobs, int n, int tiles, const float *taps, int pass);
constexpr int VF_TW = 128, VF_TH = 16, VF_RX = 32, VF_RY = 2;   // tile of the fused rotate + blur kernel and the largest halo it takes
void launch_views_rotblur(hipStream_t s, const ViewJob *jobs, int n, int tiles, const float *taps, int maxRx, int maxRy);
size_t match_workspace_bytes(int n1, int n2);
void launch_match(hipStream_t s, const uint8_t *d1, int n1, const uint8_t *d2, int n2, const double *pos2,
                  double sqminratio, double contrDistSq, int nn, MatchRow *rows, void *workspace);
// Shapes of the fused sampling + row-filter kernel (kernels_describe.hip k_sample_rows_lds; engine.hip sizes the tiles with the
// same numbers): LDS floats of the sampled row tile, block rows of a fully fused small window, whether a wave parks 4 / 8
// columns of coordinates at a time (1) or 8 / 16 (0), and the workgroups per CU the kernel is built for
// The kernel's time follows its residency far more than its tile shapes: with 4 workgroups per CU (20 KB tile + 20 KB of
// coordinates / fused block) it took 1.9 ms per 31-view pair on one stream, padded to 3 per CU 3.4 ms, and at 6 / 7 / 8 per CU
// 1.33 / 1.18 / 1.13 ms (175 -> 179 / 179.5 / 180.5 pairs/s), although fewer windows are then small enough for the fused
// column pass (the separate column filter grows from 0.69 to 0.86 ms).
#ifndef MODSX_SR_WIN
#define MODSX_SR_WIN 2400
#define MODSX_FC_ROWS 40
#define MODSX_SR_HALF 1
#define MODSX_SR_WGS 8
#endif
// LDS floats of a column-filter workgroup (kernels_describe.hip k_blur_cols_lds; engine.hip sizes the tiles with the same number)
#ifndef MODSX_BLUR_LDS_C
#define MODSX_BLUR_LDS_C 9984
#endif
constexpr int MATCH_NN_MAX = 256;   // largest `nn` of the FGINN walk (matching.hpp:268-269: 50): the event kernel lists fewer than nn groups per query
constexpr int MATCH_MAXB = 4;   // independent matching problems per launch set (blockIdx.z)
void launch_match_batch(hipStream_t s, int nb, const uint8_t *const *d1, const int *n1, const uint8_t *const *d2, const int *n2,
                        const double *const *pos2, double sqminratio, double contrDistSq, int nn, MatchRow *const *rows,
                        void *const *workspace, hipEvent_t *evSweep1 = nullptr);   // evSweep1: two events recorded around sweep 1

}  // namespace mx

namespace mx {
// experiment aid (builds with -DMODSX_DUP_BUILD only; tools/ab_dup.sh): MODSX_DUP=<bit mask of KClass> launches the kernels of those
// classes twice, which prices a kernel under the bench's 16 streams without changing any result (the launches are idempotent)
#ifdef MODSX_DUP_BUILD
int dup_count(int cls);
#define MX_DUP(cls) for (int dupLeft_ = mx::dup_count(cls); dupLeft_ > 0; dupLeft_--)
#else
#define MX_DUP(cls)
#endif
enum KClass { K_BLUR_HESS = 0, K_HESSIAN, K_RESIZE, K_NMS, K_BAUMBERG, K_ORIENT, K_PATCH_SAMPLE, K_BLUR_ROWS, K_DESCRIBE,
              K_MATCH, K_GRAY, K_WARP, K_VIEW_BLUR, K_BLUR_COLS, K_MATCH_SWEEP1, K_NCLASS };
struct Profiler {
  bool enabled = false;
  std::vector<hipEvent_t> evA, evB;
  std::vector<int> cls;
  size_t used = 0;
  double ms[K_NCLASS] = {0};
  double work[K_NCLASS] = {0};   // algorithmic bytes (flops for K_MATCH)
  long launches[K_NCLASS] = {0};
};
}  // namespace mx

struct modsx_image {
  float *d;
  int rows, cols;
  bool owned;
};

struct modsx_ctx {
  int dev;
  hipStream_t stream;
  mx::Pyramid pyr[mx::MAXB];
  mx::DevBuf candSort, candOut;   // device-side detection order: keys / indices / hash table / temp storage, and the ordered survivors
  size_t lastSurvivors = 0;
  mx::DevBuf nmsJobs, cand, counter, affJobs, affOut, oriJobs, oriOut, descJobs, tilePrefix, taps, imgRefs, scratchA, scratchB,
      descF[mx::MAXB], descU8[mx::MAXB], descAllF[2], descAllU8[2], descAllU8b[2], descCls[2][4][2], descU8x[3][mx::MAXB], shardLocal, pos2, matchRows, matchWork, misc, viewTmp[2], viewTaps, viewJobs, viewImg[mx::MAXB], scratchC, needTab, coordTab, tileJob, blurTiles, nmsQueue, rowStarts;
  mx::ImgRef imgRefsHost[mx::MAXB];   // what imgRefs holds on the device
  mx::PinBuf hDescB;           // second staging blob of describe_batch: chunk k + 1 is prepared while chunk k runs
  hipEvent_t descEv[2];
  bool descEvPending[2] = {false, false};
  bool descByEvent[2] = {true, true};
  unsigned descMark[2] = {0, 0};   // ... as sequence numbers of the context's flag word (ctx_mark) when the runtime's waits are not used
  unsigned *hFlag = nullptr;   // pinned word the stream's flag kernel writes (ctx_sync, engine.hip)
  unsigned flagSeq = 0;        // last sequence number issued
  bool waitRuntime = true;     // how the call in progress waits and copies (latched by CtxBusy from MODSX_HOST_WAIT / the CPU load)
  mx::PinBuf hCand, hAff, hOri, hDesc, hMisc, hNms, hMatch, hViewTaps, hViewJobs, hMser, hRefs;
  // constant tables on device
  float *dSmmMask = nullptr;   // 19x19 computeGaussMask
  int smmW = 0;
  std::vector<uint64_t> candKeys, candKeys2, candClaim;   // detection-order sort + octaveMap claim scratch (host)
  std::vector<uint32_t> candOrder, candOrder2;
  float *dOriMask = nullptr;   // values of the 41x41 circular mask (sigma = 41/3) at the ORI_NV listed pixels
  unsigned short *dOriIdx = nullptr;   // their index in the kernel's padded 44-column patch, raster order
  float *dSiftMask = nullptr;  // 41x41 circular mask, sigma2 = 0.9 r^2
  unsigned short *dSiftMaskIdx = nullptr;  // raster-ordered indices of the pixels with mask > 0
  int nSiftMask = 0;
  double *dAtan = nullptr;
  unsigned char *dOriBinTab = nullptr;   // histogram bin of every atan2LUT angle (ATAN_CASES entries)
  float *dSiftOTab = nullptr;            // fractional SIFT orientation bin of every atan2LUT angle
  int *dSiftBins = nullptr;    // bin0[41], bin1[41]
  double *dSiftW = nullptr;    // w0[41], w1[41]
  double timings[6];
  mx::Profiler prof;
  size_t lastCandCount = 0;    // scale-space candidates of the context's last launch set (sizes the speculative download)
  int busyDepth = 0;           // nesting of mx::CtxBusy on this context (its driving thread only)
  int shardLane = 0;           // lane of the rank's communicator this context issues its collectives on (engine_shard.hip)
  modsx_ctx *peer = nullptr;   // second stream + buffers, created on demand: the two images of a multi-view pair run side by side
  modsx_ctx *half = nullptr;   // a lone pair: the second part of an image's views runs here (accumulate_views)
  mx::DevBuf halfDesc[4];      // ... and writes its descriptors (one buffer per descriptor class of the step) here until the first part's count is known
  void *worker = nullptr;      // CtxWorker: the host thread that drives this context when it is a peer / half (engine_views.hip)
};
