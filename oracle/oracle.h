/*
 * oracle.h -- C interface of the CPU oracle.
 *
 * TEST INFRASTRUCTURE ONLY.  This is a CPU restatement of the reference's
 * (ducha-aiki/mods) algorithm for the hot path, written to be called from
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.  Nothing in
 * mods_amd/ (the product) may include, link or call it.
 *
 * Parity status: the reference holds NO golden vectors or unit tests for this
 * path; the only anchor is build/examples/cat.txt (the
 * ground-truth homography of the example pair).  OpenCV 2.4.9 (GaussianBlur,
 * resize, invert, flann) is an un-vendored dependency whose arithmetic is
 * restated here from its published algorithm.  => "parity unpinned" for the
 * detection/description stages; RANSAC is pinned against the reference's own
 * degensac sources compiled in place (oracle/_ref, see oracle/Makefile).
 */
#ifndef MODSX_ORACLE_H
#define MODSX_ORACLE_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* layout of AffineKeypoint, detectors/structures.hpp:187-199 */
typedef struct orc_keypoint {
  double x, y;
  double a11, a12, a21, a22;
  double s;
  double response;
  int octave_number;
  double pyramid_scale;
  int sub_type;
} orc_keypoint;

/* AffineRegion without the heap descriptor, detectors/structures.hpp:222-233 */
typedef struct orc_region {
  int img_id, img_reproj_id, id, parent_id, type;
  orc_keypoint det_kp, reproj_kp;
} orc_region;

/* PyramidParams + AffineShapeParams, detectors/structures.hpp:125-160, affinedetectors/affine.h:27-62 */
typedef struct orc_hessaff_params {
  float threshold;
  int mode;            /* detection_mode_t: 0 FIXED_TH .. 4 NOT_LESS_THAN_REGIONS */
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
  int detectorType;    /* detector_type, detectors/structures.hpp:16-19: 0 Hessian, 1 DoG, 2 Harris (ScaleSpaceDetector::Response) */
} orc_hessaff_params;

/* a scale-space keypoint before affine adaptation (debug / stage parity) */
typedef struct orc_sskp {
  int octave, level, r0, c0, r, c, type, pad;
  float b0, b1, b2, val;
  float x, y, s, pixelDistance;
} orc_sskp;

typedef struct orc_tentative {
  int q;        /* index in list1 (query)                       */
  int t0;       /* nearest neighbour in list2 ("second")         */
  int tj;       /* first ratio-passing neighbour ("secondbad")   */
  int t1;       /* 2nd closest ("secondbadby2ndcl")              */
  double d1, d2, d2by2ndcl, ratio;
} orc_tentative;

void orc_default_hessaff_params(orc_hessaff_params *p);

/* image helpers */
void orc_gray_from_bgr_u8(const uint8_t *bgr, int rows, int cols, float *out);
int  orc_gaussian_kernel(int n, double sigma, float *out);
int  orc_blur_ksize(float sigma);
void orc_gaussian_blur(const float *in, int rows, int cols, float sigma, float *out);
void orc_resize_half(const float *in, int rows, int cols, float *out, int *orows, int *ocols);
void orc_hessian_response(const float *in, int rows, int cols, float norm, float *out);
void orc_response(const float *in, int rows, int cols, int detector_type, float norm, float *out);
int  orc_interpolate(const float *im, int rows, int cols, float ofsx, float ofsy,
                     float a11, float a12, float a21, float a22, float *res, int rrows, int rcols);
float orc_atan2lut(float y, float x);
const double *orc_atan_lut(void);

/* view synthesis: ViewSynthParameters (detectors/structures.hpp:201-214) and GenerateSynthImageCorr */
typedef struct orc_view {
  double zoom, tilt, phi, InitSigma;
  int doBlur;
} orc_view;
/* cv::warpAffine(src, dst, M(2x3 forward), Size(dcols,drows), INTER_LINEAR, BORDER_CONSTANT, 128) */
void orc_warp_affine(const float *src, int rows, int cols, const double *M6, float *dst, int drows, int dcols,
                     float border);
/* cv::GaussianBlur(img, img, Size(kx,ky), sx, sy) with BORDER_REFLECT_101 */
void orc_gaussian_blur_xy(const float *in, int rows, int cols, int kx, int ky, double sx, double sy, float *out);
/* GenerateSynthImageCorr (synth-detection.cpp:236-430) on a gray f32 image: returns view size and H (orig->view);
 * pass out = NULL to query the size */
int orc_synth_view(const float *gray, int rows, int cols, const orc_view *v, float *out, int *orows, int *ocols,
                   double *H9);
/* SetVSPars (synth-detection.cpp:103-234): prev[] is updated in place (capacity cap_prev) */
int orc_set_vs_pars(const double *scale_set, int ns, const double *tilt_set, int nt, double phi_base,
                    double InitSigma, int doBlur, orc_view *par, int cap, orc_view *prev, int *nprev, int cap_prev);

/* pyramid of one octave: blurs[5], responses[5] (each rows*cols), returns next octave base */
void orc_octave_levels(const float *first, int rows, int cols, const orc_hessaff_params *p,
                       float *blurs, float *resps);

/* detection */
int orc_detect_scalespace(const float *img, int rows, int cols, const orc_hessaff_params *p,
                          orc_sskp *out, int cap);
int orc_detect_hessaff(const float *img, int rows, int cols, const orc_hessaff_params *p,
                       double tilt, double zoom, orc_keypoint *out, int cap);
int orc_find_affine_shape(const float *blur, int rows, int cols, const orc_hessaff_params *p,
                          float x, float y, float s, float pixelDistance, float *u /*4*/);
int orc_detect_affine_regions(const orc_keypoint *kps, int n, int img_id, int det_type, orc_region *out);
int orc_detect_orientation(const float *img, int rows, int cols, const orc_region *in, int n,
                           double mrSize, int patchSize, int doHalfSIFT, int maxAngNum, double th,
                           int addUpRight, orc_region *out, int cap);
int orc_dominant_angles(const float *patch, int patchSize, int doHalfSIFT, double th, int maxAngles,
                        float *angles, int cap);
int orc_reproject_regions(orc_region *regs, int n, const double *H, int orig_w, int orig_h);
int orc_reproject_regions_touch_boundary(orc_region *regs, int n, const double *H, int orig_w, int orig_h, double mrSize);
/* descriptor: type 0 SIFT, 1 RootSIFT; desc is n*128 floats holding integers 0..255 */
void orc_describe_regions(const float *img, int rows, int cols, const orc_region *regs, int n,
                          double mrSize, int patchSize, int fast, int photoNorm, int rootsift,
                          double maxBinValue, float *desc);
void orc_describe_patch(float *patch41, int photoNorm, int rootsift, double maxBinValue, float *desc);
void orc_extract_patch(const float *img, int rows, int cols, const orc_region *reg,
                       double mrSize, int patchSize, float *patch);

/* matching: linear kNN (squared L2, ties by ascending index) + FGINN walk */
int orc_match_fginn(const float *desc1, int n1, const float *desc2, int n2, int dim,
                    const double *pos2 /* n2*2 : reproj x,y */, double ratio, double contradDist,
                    int nn, orc_tentative *out, int cap);
int orc_knn_linear(const float *desc1, int n1, const float *desc2, int n2, int dim, int nn,
                   int *idx, float *dist);
/* duplicate filtering: pts = T*4 (x1,y1,x2,y2); key = sort key per mode; keep[] out; returns kept,
 * order[] = permutation applied by the (unstable) std::sort  */
int orc_duplicate_filtering(const double *pts, const double *key, int T, double r, int do_sort,
                            int *order, unsigned char *keep);

/* verification around exp_ransacHcustom (from oracle/_ref when loaded) */
int orc_ref_available(void);
int orc_loransac_h(const double *pts /*T*4*/, const double *laf1 /*T*5: a11 a12 a21 a22 s*/,
                   const double *laf2, int T, double err_threshold, double confidence, int max_samples,
                   int lo, double HLAFCoef, int doSymmCheck, unsigned seed,
                   double *H /*9 row-major img1->img2*/, double *Hraw /*9 as returned*/,
                   unsigned char *inl /*T ransac inliers*/, unsigned char *keep /*T after LAF check*/,
                   int *data_out /*3*/, int error_type /*0 SAMPSON, 1 SYMM_MAX, 2 SYMM_SUM: matching.cpp:821-846*/);
int orc_loransac_f(const double *pts, const double *laf1, const double *laf2, int T, double err_threshold,
                   double confidence, int max_samples, int lo, double LAFCoef, int doSymmCheck, int error_type,
                   unsigned seed, double *F /*9 as exp_ransacFcustom returns it*/, unsigned char *inl,
                   unsigned char *keep /*after F_LAF_check*/, int *data_out3);

/* DetectMSERs (6-arg overload, doOnNormal branch), detectors/mser/extrema/extrema.cpp:284-473 -- PARITY UNPINNED,
 * see oracle_mser.cpp.  mode: 0 FIXED_TH, 1 RELATIVE_TH, 2 FIXED_REG_NUMBER, 3 RELATIVE_REG_NUMBER,
 * 4 NOT_LESS_THAN_REGIONS.  Writes min(count, cap) keypoints, returns count. */
int orc_detect_msers(const float *img, int rows, int cols, int min_size, double max_area, double min_margin, int relative,
                     int mode, int reg_number, double rel_threshold, double rel_reg_number, double tilt, double zoom,
                     orc_keypoint *out, int cap);

#ifdef __cplusplus
}
#endif
#endif
