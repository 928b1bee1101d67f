/*
 * modsx.h -- C ABI of the MI355X-native MODS inner loop (libmodsx.so).
 *
 * Every entry point replaces one interface of the reference (ducha-aiki/mods);
 * the file:line it replaces is cited on each declaration.  Plain pointers and
 * sizes only; no exceptions cross this boundary.  Conventions: an int return is
 * a count (>= 0) or a negative modsx_status; arrays returned through a `**`
 * parameter are malloc'd by the library and released with modsx_free(); device
 * state lives behind opaque handles; one modsx_ctx per calling thread (the
 * reference calls its detectors concurrently from nested OpenMP threads,
 * imagerepresentation.cpp:612-622 -- a ctx owns one HIP stream).
 *
 * The library needs a gfx950 device: modsx_create() fails (NULL +
 * modsx_last_error()) when none is present.  There is no CPU fallback.
 */
#ifndef MODSX_H
#define MODSX_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define MODSX_VERSION 100

typedef enum modsx_status {
  MODSX_OK = 0,
  MODSX_ERR_ARG = -1,
  MODSX_ERR_DEVICE = -2,
  MODSX_ERR_NOMEM = -3,
  MODSX_ERR_INTERNAL = -4,
  MODSX_ERR_CAPACITY = -5,  /* a caller-provided (or pooled) output buffer is too small: retry with a larger one */
  MODSX_ERR_TIMEOUT = -6    /* view-sharded path: a collective did not complete within the communicator's deadline (a peer rank
                               is missing or failed); the communicator is dead, destroy it and create a new one */
} modsx_status;

/* detection_mode_t, detectors/structures.hpp:11-15 */
enum { MODSX_FIXED_TH = 0, MODSX_RELATIVE_TH = 1, MODSX_FIXED_REG_NUMBER = 2, MODSX_RELATIVE_REG_NUMBER = 3,
       MODSX_NOT_LESS_THAN_REGIONS = 4 };
/* detector_type / descriptor_type, detectors/structures.hpp:17-38, 75-96 */
enum { MODSX_DET_HESSIAN = 0, MODSX_DET_DOG = 1, MODSX_DET_HARRIS = 2, MODSX_DET_MSER = 3 };   /* detector_type, detectors/structures.hpp:16-19 */
/* descriptor types of SIFTDescriptor::operator() (matching/siftdesc.cpp:399-442).  The half variants fold opposite
 * orientation bins into 64 values; their rows keep the 128 stride with entries 64..127 zero, which leaves every
 * L2 distance unchanged, so the matcher needs no second layout. */
enum { MODSX_DESC_SIFT = 0, MODSX_DESC_ROOT_SIFT = 1, MODSX_DESC_HALF_SIFT = 2, MODSX_DESC_HALF_ROOT_SIFT = 3 };
#define MODSX_MAX_DESC 4   /* descriptor classes one step can carry (the four SIFT-family types above) */
/* ScaleSpaceDetector point types, affinedetectors/pyramid.h:32-36 */
enum { MODSX_HESSIAN_DARK = 0, MODSX_HESSIAN_BRIGHT = 1, MODSX_HESSIAN_SADDLE = 2 };

/* == struct AffineKeypoint, detectors/structures.hpp:187-199 (same field order and layout) */
typedef struct modsx_keypoint {
  double x, y;
  double a11, a12, a21, a22;
  double s;
  double response;
  int octave_number;
  double pyramid_scale;
  int sub_type;
} modsx_keypoint;

/* == struct AffineRegion minus the heap-allocated Descriptor, detectors/structures.hpp:222-233.
 * Descriptors travel next to the region array as a dense [n][128] f32 matrix holding the
 * integers 0..255 the reference stores in Descriptor::vec (matching/siftdesc.cpp:218-274). */
typedef struct modsx_region {
  int img_id, img_reproj_id, id, parent_id, type;
  modsx_keypoint det_kp, reproj_kp;
} modsx_region;

/* PyramidParams + AffineShapeParams of ScaleSpaceDetectorParams,
 * detectors/structures.hpp:125-160, affinedetectors/affine.h:27-62, scale-space-detector.hpp:20-28 */
typedef struct modsx_hessaff_params {
  float threshold;
  int mode;
  int reg_number;
  float rel_threshold;
  float rel_reg_number;
  int numberOfScales;
  float initialSigma;
  double edgeEigenValueRatio;
  int border;
  int maxIterations;
  float convergenceThreshold;
  int smmWindowSize;
  float affInitialSigma;
  int doBaumberg;
  int detectorType;   /* PyramidParams::DetectorType (structures.hpp:148): MODSX_DET_HESSIAN (default), MODSX_DET_DOG or MODSX_DET_HARRIS --
                       * ScaleSpaceDetector::Response (pyramid.cpp:132-175), thresholds pyramid.h:47-67, point types pyramid.cpp:66-130 */
} modsx_hessaff_params;

/* scale-space keypoint before affine adaptation: the arguments of
 * KeypointCallback::onKeypointDetected, affinedetectors/pyramid.h:23-27 */
/* == struct extrema::ExtremaParams (the fields DetectMSERs reads), detectors/mser/extrema/extremaParams.h:49-82;
 * defaults of modsx_default_mser_params = [MSER] of build/config_iter_mods_cviu.ini:4-12 */
typedef struct modsx_mser_params {
  int min_size;          /* minimum region size in pixels */
  double max_area;       /* maximum region area relative to the image */
  double min_margin;     /* stability margin in grey levels (the detector threshold) */
  int relative;          /* margin relative to the grey level */
  int mode;              /* detection_mode_t, MODSX_FIXED_TH ... */
  int reg_number;
  float rel_threshold, rel_reg_number;
} modsx_mser_params;

typedef struct modsx_sskp {
  int octave, level, r0, c0, r, c, type, pad;
  float b0, b1, b2, val;
  float x, y, s, pixelDistance;
} modsx_sskp;

/* the fields MatchFlannFGINN fills in a TentativeCorrespExt (matching/matching.hpp:39-52):
 * first = list1[q], second = list2[t0], secondbad = list2[tj], secondbadby2ndcl = list2[t1] */
typedef struct modsx_tentative {
  int q, t0, tj, t1;
  double d1, d2, d2by2ndcl, ratio;
} modsx_tentative;

/* parameters of one identity-view HessAff -> SIFT -> FGINN -> LO-RANSAC pass over an image pair
 * (the body of mods.cpp:229-415 for one step); defaults = build/config_iter_mods_cviu.ini */
typedef struct modsx_pair_params {
  modsx_hessaff_params det;
  double ori_mrSize; int ori_patchSize; int ori_maxAngles; double ori_threshold; /* [DominantOrientation] :102-108 */
  double desc_mrSize; int desc_patchSize; int desc_photoNorm; int desc_type; double desc_maxBinValue; /* [SIFTDescriptor] :109-116 */
  double match_ratio; double contradDist; int nn;       /* FGINNThreshold, [Matching] contradDist :147 */
  double duplicateDist;                                  /* [DuplicateFiltering] :156-159, mode bestFGINN */
  double err_threshold, confidence; int max_samples; int localOptimization; double HLAFCoef; int doSymmCheck; /* [RANSAC] :162-170 */
  unsigned ransac_seed;
  int useF;            /* RANSACPars::useF (matching.hpp:148): 1 = epipolar verification, exp_ransacFcustom + F_LAF_check */
  double LAFCoef;      /* RANSACPars::LAFCoef, threshold of F_LAF_check = LAFCoef * err_threshold */
  int errorType;       /* RANSAC_error_t: 0 SAMPSON, 1 SYMM_MAX, 2 SYMM_SUM (F path: 1 and 2 both select FDsSym) */
  int detector;        /* MODSX_DET_HESSIAN (det) or MODSX_DET_MSER (mser): which detector the view loop runs */
  modsx_mser_params mser;
  /* The `Descriptors=` / `FGINNThreshold=` lists of one [DetectorN] section (iters_mods_cviu_wxbs.ini:35-36: RootSIFT,
   * HalfRootSIFT with 0.85, 0.8).  n_desc = 0: the one class {desc_type, match_ratio}.  n_desc = 1..4: the step carries these
   * descriptor classes, each matched with its own ratio; desc_type / match_ratio are then ignored.  See "descriptor classes"
   * at modsx_ladder_step. */
  int n_desc;
  int desc_types[MODSX_MAX_DESC];
  double desc_ratios[MODSX_MAX_DESC];
} modsx_pair_params;

typedef struct modsx_pair_result {
  int n_regions1, n_regions2;    /* described regions per image */
  int n_tentatives;              /* after MatchFlannFGINN */
  int n_unique;                  /* after DuplicateFiltering */
  int n_ransac_inliers;          /* exp_ransacHcustom inliers */
  int n_verified;                /* after NaiveHCheck + H_LAF_check */
  int ransac_samples, ransac_lo;
  double H[9];                   /* row-major, image 1 -> image 2 (LORANSACFiltering's H); with useF the
                                  * fundamental matrix as exp_ransacFcustom returns it */
  /* malloc'd arrays (modsx_free): tentatives after duplicate filtering in RANSAC order,
   * inlier flags from RANSAC, flags after the LAF check */
  modsx_tentative *tentatives;
  unsigned char *ransac_inlier;
  unsigned char *verified;
} modsx_pair_result;

/* == struct ViewSynthParameters (the geometric part), detectors/structures.hpp:201-214 */
typedef struct modsx_view {
  double zoom, tilt, phi, InitSigma;
  int doBlur;
} modsx_view;

typedef struct modsx_ctx modsx_ctx;
typedef struct modsx_image modsx_image;

int modsx_version(void);
const char *modsx_last_error(void);
void modsx_free(void *p);

/* With MODSX_MALLOC_TUNE=1 in the environment the first call also raises glibc's mmap / trim thresholds (mallopt) so that the
 * ~100 MB of host tables a pair allocates and frees stay mapped between calls (1 ms of page faults per 31-view pair otherwise).
 * Opt-in because it changes malloc for the whole host process. */
modsx_ctx *modsx_create(int device_id);
/* How the calling thread waits inside the calls that keep the device busy (MODSX_HOST_WAIT = runtime | flag | auto, default auto):
 * through hipStreamSynchronize, or by napping until a pinned flag word shows the sequence number that a one-lane kernel writes
 * behind the stage's launches (15 % less host CPU per pair, about 1 % fewer pairs/s when CPUs are plentiful, more pairs/s when they
 * are not).  `auto` takes the flag wait while the process uses more than 80 % of its CPU allowance (cgroup quota / local ranks).
 * A thread that naps there has its timer slack set to 2 us (prctl PR_SET_TIMERSLACK, once): its later sleeps wake on time. */
void modsx_destroy(modsx_ctx *ctx);
int modsx_synchronize(modsx_ctx *ctx);

void modsx_default_hessaff_params(modsx_hessaff_params *p);
void modsx_default_pair_params(modsx_pair_params *p);

/* Image upload.  dtype 0 = u8, 1 = f32; channels 1 or 3 (BGR as cv::imread gives).  3-channel input is
 * converted with the reference's (B+G+R)/3 rule: GenerateSynthImageCorr, synth-detection.cpp:253-262
 * (identity view: out_img.pixels = gray, :278-289). */
/* images are limited to 16384 px per side and 64 Mpx (32-bit pixel addressing in the samplers); larger ones are refused */
modsx_image *modsx_image_upload(modsx_ctx *ctx, const void *pixels, int rows, int cols, int channels, int dtype);
/* new pixels for an image made by modsx_image_upload, same size: no allocation (a caller that streams pairs through a context
 * keeps two images per context and refills them).  Synchronous like the upload: the buffer is the caller's again on return. */
int modsx_image_update(modsx_ctx *ctx, modsx_image *img, const void *pixels, int rows, int cols, int channels, int dtype);
/* wrap pixels that already live in HBM (f32, 1 channel, dense rows); not owned */
modsx_image *modsx_image_wrap_device(modsx_ctx *ctx, const float *dev_pixels, int rows, int cols);
void modsx_image_free(modsx_ctx *ctx, modsx_image *img);
int modsx_image_download(modsx_ctx *ctx, const modsx_image *img, float *out);

/* int DetectAffineKeypoints(cv::Mat &input, vector<AffineKeypoint> &out1, ScaleSpaceDetectorParams params,
 *                           ScalePyramid &scale_pyramid, const double tilt, const double zoom)
 * affinedetectors/scale-space-detector.hpp:231, .cpp:43-85 */
int modsx_detect_affine_keypoints(modsx_ctx *ctx, const modsx_image *img, const modsx_hessaff_params *par,
                                  double tilt, double zoom, modsx_keypoint **out);
/* stage tap: ScaleSpaceDetector::detectPyramidKeypoints without the affine callback (pyramid.cpp:540-573) */
int modsx_detect_scalespace(modsx_ctx *ctx, const modsx_image *img, const modsx_hessaff_params *par,
                            modsx_sskp **out);
/* stage tap: the 5 blur and 5 response levels of the first octave built from `img` as its first level
 * (ScaleSpaceDetector::detectOctaveKeypoints, pyramid.cpp:455-538); each output is 5*rows*cols f32 */
int modsx_octave_levels(modsx_ctx *ctx, const modsx_image *img, const modsx_hessaff_params *par, float *blurs,
                        float *resps);
/* stage taps for the two image primitives: gaussianBlur (detectors/helpers.cpp:717-724) and
 * cv::resize(.., 0.5, 0.5, INTER_LINEAR) (pyramid.cpp:520) */
int modsx_gaussian_blur(modsx_ctx *ctx, const modsx_image *img, float sigma, float *out);
int modsx_resize_half(modsx_ctx *ctx, const modsx_image *img, float *out, int *orows, int *ocols);

/* stage tap: Mat ScaleSpaceDetector::Response(const Mat &inputImage, float norm), affinedetectors/pyramid.cpp:132-175, for
 * DetectorType MODSX_DET_HESSIAN (HessianResponse :223-281), MODSX_DET_DOG (dogResponse :176-181: the input minus its
 * gaussianBlur with sigma = norm) and MODSX_DET_HARRIS (HarrisResponse :283-305: gradients, three blurred products,
 * det - 0.04 trace^2).  out: rows*cols f32.  The scale-space loop of modsx_detect_affine_keypoints itself runs the
 * Hessian response only (the shipped configurations use HessianAffine and MSER); the other two are provided as kernels. */
int modsx_response(modsx_ctx *ctx, const modsx_image *img, int detector_type, float norm, float *out);

/* template DetectAffineRegions<>: scale by sqrt|det A| and rectify, synth-detection.hpp:93-126 (host math) */
int modsx_detect_affine_regions(const modsx_keypoint *kps, int n, int img_id, int det_type, modsx_region *out);

/* int DetectMSERs(cv::Mat &input, std::vector<AffineKeypoint> &out1, extrema::ExtremaParams params,
 *                 ScalePyramid &scale_pyramid, const double tilt, const double zoom)
 *                                            detectors/mser/extrema/extrema.h:11, extrema.cpp:284-473 (doOnNormal)
 * MSER+ (sub_type 21) then MSER- (sub_type 20) of one view: x, y = centroid, A = covariance^(1/2), s = 1,
 * response = margin.  The view is truncated to u8 on the device; the component tree itself is sequential host code
 * (as in the reference).  PARITY UNPINNED (the reference's MSER sources cannot be built in this image): checked against
 * an independent CPU restatement only.  *out is malloc'd (modsx_free).  Returns the number of keypoints. */
void modsx_default_mser_params(modsx_mser_params *p);
int modsx_detect_msers(modsx_ctx *ctx, const modsx_image *img, const modsx_mser_params *par, double tilt, double zoom,
                       modsx_keypoint **out);
/* the same on a host u8 image (what DetectMSERs builds at extrema.cpp:395-403); needs no device */
int modsx_detect_msers_u8(const unsigned char *gray, int rows, int cols, const modsx_mser_params *par, double tilt,
                          double zoom, modsx_keypoint **out);

/* int DetectOrientation(AffineRegionList &in, AffineRegionList &out, SynthImage &img, double mrSize,
 *                       int patchSize, int doHalfSIFT, int maxAngNum, double th, bool addUpRight)
 * synth-detection.hpp:151-159, .cpp:841-919 */
int modsx_detect_orientation(modsx_ctx *ctx, const modsx_image *img, const modsx_region *in, int n, double mrSize,
                             int patchSize, int doHalfSIFT, int maxAngNum, double th, int addUpRight,
                             modsx_region **out);

/* int ReprojectRegions(AffineRegionList &keypoints, double *H, int orig_w, int orig_h)
 * synth-detection.cpp:541-616; filters in place, returns the new count (host math) */
int modsx_reproject_regions(modsx_region *regs, int n, const double *H, int orig_w, int orig_h);
/* int ReprojectRegionsAndRemoveTouchBoundary(AffineRegionList &keypoints, double *H, int orig_w, int orig_h,
 *                                            const double mrSize = 3.0*sqrt(3.0))        synth-detection.cpp:63-102, .hpp:68
 * The same with the box mrSize * s instead of k_sigma * s: the un-oriented "None" list SynthDetectDescribeKeypoints keeps
 * beside the described ones (imagerepresentation.cpp:1271-1272; what SaveRegions writes for regions without descriptors). */
int modsx_reproject_regions_touch_boundary(modsx_region *regs, int n, const double *H, int orig_w, int orig_h, double mrSize);

/* template DescribeRegions<SIFTDescriptor>(AffineRegionList&, SynthImage&, FuncType, double mrSize,
 *            int patchSize, bool fast_extraction, bool photoNorm)     synth-detection.hpp:169-255
 * with SIFTDescriptor::operator() matching/siftdesc.cpp:401-442.  desc: n*128 f32 (integers 0..255). */
int modsx_describe_regions(modsx_ctx *ctx, const modsx_image *img, const modsx_region *regs, int n, double mrSize,
                           int patchSize, int fast_extraction, int photoNorm, int desc_type, double maxBinValue,
                           float *desc);

/* int MatchFlannFGINN(const AffineRegionList &list1, const AffineRegionList &list2,
 *                     TentativeCorrespListExt &corresp, const MatchPars &par, const int nn = 50)
 * matching/matching.hpp:268-269, .cpp:357-461 with vector_matcher = linear, vector_dist = L2.
 * desc*: [n][128] f32 holding integers 0..255 (anything else -- fractions, out-of-range values, NaN -- is refused with
 * MODSX_ERR_ARG: the int8 matrix-core path is exact only on that domain); pos2: [n2][2] reproj_kp x,y of list2.
 * A ratio >= 1 takes the reference's "all points" branch (matching.cpp:397-428): a record per query, closed by its first
 * contradictive neighbour or by neighbour nn - 1 (no ratio test).
 * nn must lie in [2, 256] (the reference takes any nn, default 50; the device walk lists fewer than nn groups of trains per
 * query, in 256 slots): other values return MODSX_ERR_ARG on every match path, the sharded and fused ones included.
 * Images must have at least 2 rows and 2 columns (modsx_image_upload / modsx_image_wrap_device refuse smaller ones). */
int modsx_match_fginn(modsx_ctx *ctx, const float *desc1, int n1, const float *desc2, int n2, const double *pos2,
                      double ratio, double contradDist, int nn, modsx_tentative **out);

/* void DuplicateFiltering(TentativeCorrespListExt&, const double r, const int mode) matching/matching.hpp:300,
 * .cpp:2983-3047.  pts: [T][4] = x1 y1 x2 y2; key: |ratio| (bestFGINN), d1 or scale; order[] receives the
 * permutation of the sort, keep[] the survivor flags in sorted order.  Returns the survivor count. */
int modsx_duplicate_filtering(const double *pts, const double *key, int T, double r, int do_sort, int *order,
                              unsigned char *keep);

/* Score exp_ransacHcustom(double *u, int len, double th, double conf, int max_sam, double *H,
 *          unsigned char *inl, int iter_type, int *data_out, int oriented_constraint, unsigned inlLimit,
 *          double **resids, HDsPtr, HDsiPtr, HDsidxPtr, int doSymCheck)       degensac/exp_ranH.h:29-36
 * with the Sampson error functions (HDs/HDsi/HDsidx) and iter_type 4, plus an explicit seed in place of
 * srand(time(NULL)) (exp_ranH.c:823).  u: [len][6] = x1 y1 1 x2 y2 1; H: column-major-as-returned (maps
 * image 2 -> image 1 transposed, see matching.cpp:922-938); data_out[0..2] = samples, LO count, orientation
 * rejects.  Returns the inlier count. */
int modsx_ransac_h(const double *u, int len, double th, double conf, int max_sam, double *H, unsigned char *inl,
                   int *data_out, int oriented_constraint, int doSymCheck, unsigned seed, double *score_J);

/* int LORANSACFiltering(TentativeCorrespListExt &in, TentativeCorrespListExt &out, double *H,
 *                       const RANSACPars pars)   matching/matching.hpp:284-286, .cpp:806-980 (useF = 0,
 * errorType SAMPSON).  pts [T][4]; laf1/laf2 [T][5] = reproj a11 a12 a21 a22 s.  Outputs H (row-major
 * img1->img2), Hraw (as exp_ransacHcustom returned it), inl (RANSAC), keep (after NaiveHCheck + H_LAF_check).
 * Returns the verified count. */
int modsx_loransac_h(const double *pts, const double *laf1, const double *laf2, int T, double err_threshold,
                     double confidence, int max_samples, int localOptimization, double HLAFCoef, int doSymmCheck,
                     unsigned seed, double *H, double *Hraw, unsigned char *inl, unsigned char *keep,
                     int *data_out);

/* The same two calls with RANSACPars::errorType (matching.cpp:821-846 picks the HDS1 / HDSi1 / HDSidx1 triple the whole
 * of exp_ransacHcustom scores with): 0 SAMPSON = HDs, 1 SYMM_MAX = HDsSymMax, 2 SYMM_SUM = HDsSym (the RANSACPars
 * default, matching.hpp:138-171).  modsx_ransac_h / modsx_loransac_h are error_type 0; modsx_pair_params.errorType
 * selects it for the fused callers. */
int modsx_ransac_h_errtype(const double *u, int len, double th, double conf, int max_sam, double *H, unsigned char *inl,
                           int *data_out, int oriented_constraint, int doSymCheck, int error_type, unsigned seed,
                           double *score_J);
int modsx_loransac_h_errtype(const double *pts, const double *laf1, const double *laf2, int T, double err_threshold,
                             double confidence, int max_samples, int localOptimization, double HLAFCoef, int doSymmCheck,
                             int error_type, unsigned seed, double *H, double *Hraw, unsigned char *inl,
                             unsigned char *keep, int *data_out);

/* int exp_ransacFcustom(double *u, int len, double th, double conf, int max_sam, double *F, unsigned char *inl,
 *                       int *data_out, int do_lo, unsigned inlLimit, double **resids, double *H_best, int *Ih,
 *                       exFDsPtr EXFDS1, FDsPtr FDS1, int doSymCheck)      degensac/exp_ranF.c:795-1192
 * LO-RANSAC + DEGENSAC (plane-and-parallax) for the fundamental matrix; u [len][6] = x1 y1 1 x2 y2 1, th squared.
 * error_type 0 = Sampson (FDs/exFDs), 1 = symmetric epipolar distance (FDsSym/exFDsSym), matching.cpp:821-846.
 * F: 9 doubles as the reference returns them (x2^T F x1 = 0 with F read row-wise).  data_out: samples drawn,
 * local optimisations, degenerate (H-consistent) samples handled.  `seed` replaces srand(time(NULL)).
 * Returns the inlier count of the best model. */
int modsx_ransac_f(const double *u, int len, double th, double conf, int max_sam, int do_lo, unsigned inl_limit,
                   int error_type, int doSymCheck, unsigned seed, double *F, unsigned char *inl, int *data_out);

/* LORANSACFiltering with RANSACPars::useF = 1 (matching.cpp:806-980): exp_ransacFcustom + F_LAF_check (:193-250,
 * threshold LAFCoef * err_threshold).  Same argument layout as modsx_loransac_h.  Returns the verified count. */
int modsx_loransac_f(const double *pts, const double *laf1, const double *laf2, int T, double err_threshold,
                     double confidence, int max_samples, int localOptimization, double LAFCoef, int doSymmCheck,
                     int error_type, unsigned seed, double *F, unsigned char *inl, unsigned char *keep, int *data_out);
/* The hypothesis loop of DEGENSAC's plane-and-parallax step (rFtH, DegUtils.c:254-440: up to 2 x 10^4 two-point epipoles per
 * H-degenerate sample, each scored with FDs over the off-plane correspondences) runs on the device when the calling thread has
 * one: the host draws the samples of a batch from a copy of the PRNG, one device thread per hypothesis counts, the host acts on
 * the first count that changes the loop's state exactly as the reference does.  Same trajectory, same F (tests/test_gpu_verify.py
 * against the reference's compiled degensac).  MODSX_VERIFY_DEVICE=0 keeps the loop on the host; without a device it is there
 * anyway.  A device state that failed to set up, hit a HIP error or disagreed with the host's recount is torn down, the loop at
 * hand finishes on the host, and the next loop starts a fresh state; after 8 such failures in a process the loops stay on the host.  out[0..6): device batches, hypotheses counted on the device, state-changing hypotheses, host / device disagreements,
 * rFtH loops run (either way) and the microseconds they took (process-wide; reset != 0 clears them).  Returns 6. */
int modsx_verify_device_stats(long *out, int reset);
/* out[0..4): microseconds of those loops spent drawing samples ahead of a batch, waiting for the device, in the host phase
 * (hypotheses that changed nothing, counted on the host at the start of a loop and after every state change) and in the bodies of
 * state-changing hypotheses.  Process-wide; reset != 0 clears them.  Returns 4. */
int modsx_verify_device_timing(long *out, int reset);

/* One step of mods.cpp's iteration loop (:229-415) for an identity view: detect + orient + describe both
 * images, match, filter duplicates, verify.  Images and all intermediates stay in HBM between stages. */
int modsx_match_pair(modsx_ctx *ctx, const modsx_image *img1, const modsx_image *img2,
                     const modsx_pair_params *par, modsx_pair_result *res);
void modsx_pair_result_release(modsx_pair_result *res);

/* A batch of independent pairs, pipelined over n_ctx contexts (one host thread + one HIP stream each), the
 * counterpart of the reference's OpenMP parallelism over images / views (mods.cpp:255-271,
 * imagerepresentation.cpp:612-622): host-side bookkeeping of one pair overlaps device work of another.
 * Every context takes up to 4 pairs at a time and runs their 8 images as one launch set; DuplicateFiltering +
 * LO-RANSAC of a finished group run on helper threads (one per working context) while the context's thread feeds
 * the next group to its stream.  Results are those of n_pairs modsx_match_pair calls.
 * All contexts must live on the device that holds the images.  Returns n_pairs. */
int modsx_match_pairs(modsx_ctx *const *ctxs, int n_ctx, const modsx_image *const *imgs1,
                      const modsx_image *const *imgs2, int n_pairs, const modsx_pair_params *par,
                      modsx_pair_result *results);

/* The same for multi-view pairs (mods.cpp's loop over image pairs with one view list per image, as modsx_match_pair_views
 * per pair): every context takes one pair at a time off a shared counter; DuplicateFiltering + LO-RANSAC of a matched pair
 * run on helper threads while the context goes on to the next pair.  Results are those of n_pairs modsx_match_pair_views
 * calls.  Returns n_pairs. */
int modsx_match_pairs_views(modsx_ctx *const *ctxs, int n_ctx, const modsx_image *const *imgs1,
                            const modsx_image *const *imgs2, int n_pairs, const modsx_view *views, int n_views,
                            const modsx_pair_params *par, modsx_pair_result *results);

/* host share of the last modsx_match_pairs / modsx_match_pairs_views call: wall time of DuplicateFiltering + LO-RANSAC summed over its pairs (ms),
 * the pairs verified and the helper threads that ran them (measurement hook) */
int modsx_last_batch_verify(double *sum_ms, int *pairs, int *threads);
/* per-stage time of the last modsx_match_pair in ms: detect, orient, describe, match, verify, total */
int modsx_last_timings(modsx_ctx *ctx, double *ms6);

/* int SetVSPars(scale_set, tilt_set, phi_base, FGINNThreshold, DistanceThreshold, descriptors, par, prev_par,
 *               InitSigma, doBlur, dsplevels, minSigma, maxSigma)            synth-detection.cpp:103-234
 * The view ladder of one step, de-duplicated against the views of earlier steps (prev[0..*nprev) is extended).
 * Returns the number of new views (host only). */
int modsx_set_vs_pars(const double *scale_set, int ns, const double *tilt_set, int nt, double phi_base,
                      double InitSigma, int doBlur, modsx_view *par, int cap, modsx_view *prev, int *nprev,
                      int cap_prev);

/* void GenerateSynthImageCorr(const cv::Mat &in_img, SynthImage &out_img, name, tilt, phi, zoom, InitSigma,
 *                             doBlur, img_id, convert2gray)                    synth-detection.cpp:236-430
 * gray: an uploaded image (gray conversion already applied).  Returns a new image handle (modsx_image_free),
 * H9 = SynthImage::H (original -> view); *is_identity = 1 when the reference takes its "original image"
 * short-cut (the handle then aliases `gray`). */
modsx_image *modsx_synth_view(modsx_ctx *ctx, const modsx_image *gray, const modsx_view *view, double *H9,
                              int *is_identity);

/* The HessianAffine / SIFT-family branch of ImageRepresentation::SynthDetectDescribeKeypoints
 * (imagerepresentation.cpp:603-2047) for the views view_begin, view_begin+view_step, ...: synthesise, detect,
 * orient, reproject to the original frame, describe.  regs: malloc'd, in view order, ids re-based as
 * AddRegions does (:588-600, :2044-2045) when the call covers all views (view_step == 1), otherwise local to
 * each view block (img_id = view index) so that shards can be merged.  desc (optional): malloc'd [n][128] f32.
 * dev_desc_u8 (optional): device buffer of capacity dev_cap regions that receives the [n][128] u8 descriptors
 * (what the matcher consumes) without leaving HBM.  view_counts (optional): [nviews] regions per view. */
int modsx_detect_describe_views(modsx_ctx *ctx, const modsx_image *img, const modsx_view *views, int nviews,
                                const modsx_pair_params *par, int view_begin, int view_step, modsx_region **regs,
                                float **desc, void *dev_desc_u8, long dev_cap, int *view_counts);

/* MatchFlannFGINN on u8 descriptors that already live in HBM (e.g. after an all-gather over xGMI) */
int modsx_match_fginn_device(modsx_ctx *ctx, const void *dev_desc1_u8, int n1, const void *dev_desc2_u8, int n2,
                             const double *pos2, double ratio, double contradDist, int nn, modsx_tentative **out);

/* One step of mods.cpp:229-415 with a ladder of synthesised views per image (same views for both images,
 * as in iters_mods_cviu.ini). */
int modsx_match_pair_views(modsx_ctx *ctx, const modsx_image *img1, const modsx_image *img2,
                           const modsx_view *views, int nviews, const modsx_pair_params *par,
                           modsx_pair_result *res);

/* The iteration ladder of mods.cpp:229-415 (`for step < maxSteps && curr_matches < minMatches`), restricted to
 * the HessianAffine + SIFT-family class with LO-RANSAC homography verification and doBeforeRANSAC = 1
 * (config_iter_mods_cviu.ini).  Step k synthesises its views for both images, APPENDS the regions to the two
 * image representations (AddRegions, imagerepresentation.cpp:552-600), re-matches everything accumulated so far
 * (MatchImgReps clears and rebuilds the class' tentatives every step, correspondencebank.cpp:291-345) and
 * verifies; the loop ends once n_verified >= min_matches.  match_ratio of a step (FGINNThreshold of that
 * iteration's section) overrides par->match_ratio when > 0.  *steps_done = steps executed.
 * Two detector classes are kept apart as in the reference (separate_detectors): a step adds regions to, and re-matches,
 * only the class of its own detector; the other class keeps its tentatives (CorrespondenceBank, correspondencebank.cpp:
 * 180-218) and GetCorresponcesVector() concatenates HessianAffine before MSER (map order).
 *   Descriptor classes (the WxBS ladder, iters_mods_cviu_wxbs.ini:35-41,48-54,61-67).  A step may carry several descriptors.
 * The reference then keeps one region list per (detector, descriptor) -- RegionVectorMap[det][desc] -- and one tentative list
 * per (descriptor, detector), each matched separately with the FGINNThreshold of its descriptor (MatchImgReps,
 * correspondencebank.cpp:291-347).  All descriptors of a step are computed on ONE oriented region list: when any of them
 * is a Half type the dominant orientations come from DetectOrientation(..., doHalfSIFT = true, ...) (opposite histogram
 * bins folded, synth-detection.cpp:801-808) and every descriptor of the step -- RootSIFT included -- is described on that
 * list (imagerepresentation.cpp:693-706, 1259-1264, 1288-1296).  The fused callers do exactly that: one orientation pass
 * (Half-folded iff a Half type is present), one pass over the patches that emits every descriptor of the step.
 *   GetCorresponcesVector("All", "All") walks std::map<descriptor name, std::map<detector name, list>> (correspondencebank.cpp:
 * 117-179), i.e. the tentatives reach DuplicateFiltering / LO-RANSAC in the order HalfRootSIFT, HalfSIFT, RootSIFT, SIFT and,
 * inside a descriptor, HessianAffine before MSER.  Region indices of the result refer to the concatenation of the per-class
 * region lists of each image in that same order (with one descriptor class: [HessianAffine regions, MSER regions]). */
typedef struct modsx_ladder_step {
  const modsx_view *views;
  int nviews;
  double match_ratio;
  int detector;      /* MODSX_DET_HESSIAN (0) or MODSX_DET_MSER: the [HessianAffineN] / [MSERN] section this step comes from */
  /* the section's Descriptors / FGINNThreshold lists; n_desc = 0: par->n_desc / desc_types / desc_ratios (and, when those are
   * empty too, the one class {par->desc_type, match_ratio}) */
  int n_desc;
  int desc_types[MODSX_MAX_DESC];
  double desc_ratios[MODSX_MAX_DESC];
} modsx_ladder_step;
int modsx_match_ladder(modsx_ctx *ctx, const modsx_image *img1, const modsx_image *img2,
                       const modsx_ladder_step *steps, int nsteps, int min_matches, const modsx_pair_params *par,
                       modsx_pair_result *res, int *steps_done);

/* ---- view-sharded multi-GPU path (one process per GPU, RCCL over xGMI) ------------------------------------------------
 * The reference's unit of parallelism is the synthesised view (`#pragma omp parallel for` over views,
 * imagerepresentation.cpp:612-622); views meet only in AddRegions' ordered concatenation (:2044-2045, ids re-based by
 * AddRegionsToList :588-600).  View v belongs to rank v mod world.  A modsx_comm is ONE communicator per rank; the
 * 128-byte id comes from modsx_comm_unique_id() (RCCL, bound at run time with dlopen) or modsx_comm_loopback_id() on one
 * rank and reaches the others by any side channel (MPI, torch.distributed, a file, a shared variable).
 *   Lanes.  The contexts (host thread + stream) of a rank that use the communicator at the same time are its LANES:
 * modsx_comm_set_lanes(n) once, modsx_comm_attach(comm, ctx, lane) per context, one thread per lane.  Collectives are issued
 * in strict round-robin lane order, the same sequence on every rank: lane k of every rank must make the same sharded
 * calls in the same order, and every lane the same number of them per round (modsx_comm_lane_done() takes a lane that
 * has finished out of the rotation, modsx_comm_reset_lanes() puts all of them back at a quiescent point).
 *   Errors are collective: a failure on one rank (bad view, out of memory, ...) is carried in the exchanged headers and
 * every rank returns it from the same call; no rank is left waiting.  A collective that does not complete within the
 * deadline (modsx_comm_set_timeout, env MODSX_COMM_TIMEOUT_MS, default 30000) aborts the communicator: the call and all
 * later ones on it return MODSX_ERR_TIMEOUT.
 *   The loopback transport runs `world` ranks inside ONE process on ONE device (one context + host thread per rank): the
 * all-gather is `world` device-to-device copies behind an event handshake, everything above it is the code the RCCL
 * transport runs.  It exists so that world > 1 is testable on a single GPU. */
typedef struct modsx_comm modsx_comm;
int modsx_comm_unique_id(void *id128);
int modsx_comm_loopback_id(void *id128, int world);
modsx_comm *modsx_comm_create(modsx_ctx *ctx, const void *id128, int rank, int world);
void modsx_comm_destroy(modsx_comm *comm);
int modsx_comm_set_lanes(modsx_comm *comm, int nlanes);
int modsx_comm_attach(modsx_comm *comm, modsx_ctx *ctx, int lane);
int modsx_comm_lane_done(modsx_comm *comm, int lane);
int modsx_comm_reset_lanes(modsx_comm *comm);
int modsx_comm_set_timeout(modsx_comm *comm, int milliseconds);
int modsx_comm_info(const modsx_comm *comm, int *rank, int *world, int *rccl_version, long *bytes_gathered, long *collectives);
/* out[0..n): collectives issued, bytes gathered, block-size retries, agreement collectives, lanes, loopback (0/1), dead (0/1),
 * microseconds the lanes waited for their turn to issue, bytes received in owner-only exchanges, owner-only exchanges.
 * Returns the number of statistics available. */
int modsx_comm_stats(const modsx_comm *comm, long *out, int n);
/* How modsx_match_pairs_views_sharded (owner_base >= 0) moves the rows of an image side.  MODSX_EXCHANGE_ALL_GATHER (default, the
 * north star's "all-gather of regions + descriptors"): every rank receives every row.  MODSX_EXCHANGE_OWNER: the per-item counts go
 * to every rank (an all-gather of a few KB), the rows of pair g only to the rank that matches and verifies it (ncclSend / ncclRecv
 * in one group): 1 / world of the all-gather's bytes on the wire, exact sizes (no padded blocks, no retry), one more host wait per
 * call.  Set it alike on every rank before the first sharded call: it is part of the call's contract, like the view list.  The mode
 * travels in the header, and on the loopback transport (which checks sizes) ranks set differently stop with MODSX_ERR_TIMEOUT at the
 * first exchange; over RCCL they would issue all-gathers of different byte counts, which the collective library does not define
 * (a hang that ends in the communicator's deadline, or garbage that the header check then rejects) -- the library cannot detect the
 * mismatch there before the collective.  The calls that return lists to every rank keep the all-gather. */
#define MODSX_EXCHANGE_ALL_GATHER 0
#define MODSX_EXCHANGE_OWNER 1
int modsx_comm_set_exchange(modsx_comm *comm, int mode);
/* The owner-only exchange stated on the host (what the device path derives from the gathered headers; the gloo CPU test runs ranks
 * over it): item_counts[f] = regions of item f = image * nviews + view (item f belongs to rank f mod world, its rows sit in that
 * rank's local buffer in item order), image_owner[j] = the rank that reads image j.  For `rank`:
 *   sends[k] = {peer, image, first local row, rows}: the messages it sends, in issue order (images ascending);
 *   recvs[k] = {peer, image, first row of its receive buffer, rows}: the messages it receives (images ascending, then source ranks);
 *   jobs[k]  = {first row of the receive buffer, first row of the list, rows, 0}: where the rows of each item of an owned image go
 *              (list rows count ALL items: the lists keep the all-gather's positions).
 * n[0..5) = messages sent, messages received, jobs, rows of the receive buffer, list length.  cap = capacity of each array in
 * entries (4 ints each); MODSX_ERR_CAPACITY when one is too small (n[] is still filled).  Returns MODSX_OK. */
int modsx_shard_owner_plan(const int *item_counts, int nimages, int nviews, int world, int rank, const int *image_owner, int *sends,
                           int *recvs, int *jobs, int cap, long *n);
/* Where row j of the reference's list sits in the all-gathered buffer (rank r's padded block starts at r * maxrows):
 * counts[r * nviews + v] = regions of view v on rank r (0 unless r == v mod world).  Returns the list length (host only). */
int modsx_view_block_order(const int *counts, int world, int nviews, int *src, int cap, int *maxrows_out);
/* The wire format of one exchange, stated on the host -- what the pack / ordering kernels of the sharded calls do on the device
 * (the GPU tests compare the two byte for byte; the gloo CPU tests run world 2 and 3 over these functions without a device):
 *   block = header {magic "MXSH", rc, rows, items, counts[items]} padded to 64 B, then rows of R + 128 * ndesc bytes (the region
 *           part, then the region's descriptor of every class of the step), padded to block_rows rows.
 *   row_format MODSX_SHARD_ROW_REGION: R = sizeof(modsx_region) = 200, the whole region -- the calls that return region lists
 *           (modsx_detect_describe_views_sharded, the ladder);
 *   row_format MODSX_SHARD_ROW_KP: R = 56, the doubles x, y, a11, a12, a21, a22, s of the region's reproj_kp -- all that the
 *           matcher (positions) and DuplicateFiltering / LO-RANSAC read of a region: modsx_match_pairs_views_sharded, which
 *           returns pair results only (184 B per region and descriptor class on the wire instead of 328).
 * Items = (image, view) pairs, f = image * nviews + view; item f belongs to rank f mod world.
 * _bytes: size of a block.  _pack: this rank's block (counts[f] = 0 for items of other ranks; a block that is too small keeps
 * the true row count in its header).  _unpack: the reference's list from the `world` gathered blocks -- regs_out is
 * modsx_region[cap] (ROW_REGION) or double[cap][7] (ROW_KP); returns its length, MODSX_ERR_CAPACITY (*need_rows = the largest
 * row count) when a block was too small, or the rc a rank's header carries. */
#define MODSX_SHARD_ROW_REGION 0
#define MODSX_SHARD_ROW_KP 1
long modsx_shard_block_bytes(int items, int block_rows, int ndesc, int row_format);
long modsx_shard_block_pack(const modsx_region *regs, const unsigned char *const *desc, int ndesc, int n, const int *counts, int items,
                            int rc_local, int block_rows, int row_format, void *block);
long modsx_shard_blocks_unpack(const void *blocks, int world, int items, int block_rows, int ndesc, int row_format, void *regs_out,
                               unsigned char *const *desc_out, long cap, int *item_counts, int *need_rows);
/* test hooks (need a device): the same two steps through the device kernels, outputs copied back to the host */
long modsx_shard_device_pack(modsx_ctx *ctx, const modsx_region *regs, const unsigned char *const *desc, int ndesc, int n, int row_format,
                             void *rows_out);
long modsx_shard_device_unpack(modsx_ctx *ctx, const void *blocks, int world, int items, int block_rows, int ndesc, int row_format,
                               void *regs_out, unsigned char *const *desc_out, double *pos_out, long cap);
/* SynthDetectDescribeKeypoints with this rank taking views rank, rank + world, ...: one all-gather of padded blocks
 * (header with the per-view counts + rows of modsx_region + 128 u8 descriptor bytes = 328 B), device to device; the
 * reference's order is rebuilt on the device from the gathered headers; every rank returns
 * ALL regions in reference order with re-based ids.  *dev_desc_u8 = the [n][128] u8 descriptors in HBM (owned by ctx,
 * valid until its next sharded call). */
int modsx_detect_describe_views_sharded(modsx_ctx *ctx, modsx_comm *comm, const modsx_image *img, const modsx_view *views,
                                        int nviews, const modsx_pair_params *par, modsx_region **regs, void **dev_desc_u8,
                                        int *view_counts);
/* MatchFlannFGINN with the query rows split over the ranks (every rank holds all descriptors of both images): the
 * per-query result rows are all-gathered on the device, every rank returns the full tentative list in query order. */
int modsx_match_fginn_sharded(modsx_ctx *ctx, modsx_comm *comm, const void *dev_desc1_u8, int n1, const void *dev_desc2_u8,
                              int n2, const double *pos2, double ratio, double contradDist, int nn, modsx_tentative **out);
/* modsx_match_pair_views over the ranks: both images sharded by view, the match sharded by query row, DuplicateFiltering +
 * LO-RANSAC on rank `owner` only (owner < 0: on every rank); the other ranks fill the counters up to n_tentatives. */
int modsx_match_pair_views_sharded(modsx_ctx *ctx, modsx_comm *comm, const modsx_image *img1, const modsx_image *img2,
                                   const modsx_view *views, int nviews, const modsx_pair_params *par, int owner,
                                   modsx_pair_result *res);
/* n_pairs (1..16) multi-view pairs in ONE sharded call -- what keeps the fixed costs of the exchange from growing with the world
 * size: the views of all 2 n_pairs images form one item list (item f = image * nviews + view belongs to rank f mod world), so a
 * rank's launch sets hold ~2 n_pairs nviews / world views whatever the world size; ONE all-gather moves the rows of every image
 * side (MODSX_SHARD_ROW_KP: 56 B of geometry + the u8 descriptors of a region).  owner_base >= 0: pair g is matched AND verified by rank (owner_base + g) mod world -- the
 * exchange left all it needs there -- so the call holds ONE collective; results[g] is what modsx_match_pair_views returns for
 * pair g on that rank, the other ranks fill n_regions1 / n_regions2 only.  owner_base < 0: every rank returns every pair (the
 * query rows of the n_pairs problems of a descriptor class are split over the ranks, ONE all-gather of result rows per class).
 * A rank that fails after the exchange (out of memory in its own matcher) returns the error alone; its peers meet the
 * communicator's deadline at their next collective, as for a rank that died.  Returns n_pairs. */
int modsx_match_pairs_views_sharded(modsx_ctx *ctx, modsx_comm *comm, const modsx_image *const *imgs1, const modsx_image *const *imgs2,
                                    int n_pairs, const modsx_view *views, int nviews, const modsx_pair_params *par, int owner_base,
                                    modsx_pair_result *results);
/* modsx_match_ladder over the ranks (configs[3]: the iteration ladder with every step's views sharded): each step's
 * regions are exchanged and appended to the accumulated lists on every rank, the match is sharded by query row, and every
 * rank runs DuplicateFiltering + LO-RANSAC on the same tentatives with the same seed; the verified count that decides the
 * min_matches exit is all-gathered (4 bytes per rank and step) and the call fails on every rank (MODSX_ERR_INTERNAL) should
 * the ranks ever disagree, so no rank can leave the loop alone.  Results are identical on every rank and equal to
 * modsx_match_ladder's. */
int modsx_match_ladder_sharded(modsx_ctx *ctx, modsx_comm *comm, const modsx_image *img1, const modsx_image *img2,
                               const modsx_ladder_step *steps, int nsteps, int min_matches, const modsx_pair_params *par,
                               modsx_pair_result *res, int *steps_done);

/* Key files: void ImageRepresentation::SaveRegions(std::string fname, int mode) / LoadRegions(std::string fname)
 * (imagerepresentation.cpp:2139-2215; mods.cpp:236-241 reads them instead of detecting when read_pre_extracted is
 * set).  Text format, one (detector, descriptor) class after the other; files are byte-identical to the reference's
 * for the same lists.  A class = the AffineRegionVector of RegionVectorMap[det_name][desc_name]: n regions, their
 * descriptors as n rows of `stride` floats of which the first `dim` are written. */
typedef struct modsx_region_class {
  const char *det_name, *desc_name;
  const modsx_region *regs;
  const float *desc;
  int n, dim, stride;
} modsx_region_class;
int modsx_save_regions(const char *path, const modsx_region_class *classes, int nclasses);
/* Loads the class (det_name, desc_name) -- NULL / "" = the first class in the file.  regs and desc ([n][*dim]) are
 * malloc'd (modsx_free); found_det / found_desc (optional, >= 64 bytes) receive the names.  Returns n. */
int modsx_load_regions(const char *path, const char *det_name, const char *desc_name, modsx_region **regs, float **desc,
                       int *dim, char *found_det, char *found_desc);

/* Measurement hooks (no reference counterpart; the reference only keeps wall-clock TimeLog, structures.hpp:51-74).
 * modsx_profile(ctx, 1) brackets every kernel launch with HIP events on the ctx stream and accumulates, per kernel
 * class, GPU milliseconds, launch count and algorithmic work (bytes; flops for the matcher).  Classes in order:
 * blur_hess, hessian, resize, nms_localize (scan + refine), baumberg, orientation, patch_sample, blur_rows, describe,
 * match_fginn, gray, warp_affine, view_blur, blur_cols, match_sweep1 (the one k_match launch that carries the 2 N M 128
 * contraction; also part of match_fginn).  modsx_kernel_stats returns the number of classes.
 * Cost: with ROCm 7 a stream that has recorded timing events keeps per-dispatch completion signals on its queue -- every later
 * launch of that context costs its host thread more CPU (measured: 8.5 -> 29 ms per 1920x1080 pair of ~220 launches), also after
 * modsx_profile(ctx, 0).  Profile on contexts made for it, or after the throughput measurement (bench.py orders its legs so). */
int modsx_profile(modsx_ctx *ctx, int enable);
int modsx_kernel_stats(modsx_ctx *ctx, double *ms, double *work, long *launches, int n);

#ifdef __cplusplus
}
#endif
#endif
