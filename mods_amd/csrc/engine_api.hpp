// engine_api.hpp -- functions shared between engine.hip, ransac.cpp, filters.cpp and capi.hip.
#pragma once
#include <functional>
#include <string>
#include <vector>
#include "engine.hpp"

struct modsx_comm;

namespace mx {
const char *last_error();
void prof_collect(modsx_ctx *c);
void prof_reset(modsx_ctx *c, bool enable);
modsx_ctx *ctx_create(int device_id);
void ctx_destroy(modsx_ctx *c);
int build_pyramids(modsx_ctx *c, const modsx_image *const *imgs, int n, const modsx_hessaff_params &p,
                   bool singleOctaveFromFirstLevel);
int detect_scalespace_batch(modsx_ctx *c, const modsx_image *const *imgs, int n, const modsx_hessaff_params &p,
                            std::vector<modsx_sskp> *out);
int detect_keypoints_batch(modsx_ctx *c, const modsx_image *const *imgs, int n, const modsx_hessaff_params &par,
                           const double *tilts, const double *zooms, std::vector<modsx_keypoint> *out);
void detect_affine_regions(const modsx_keypoint *kps, int n, int img_id, int det_type, modsx_region *out);
int detect_orientation_batch(modsx_ctx *c, const modsx_image *const *imgs, int n, const std::vector<modsx_region> *in,
                             double mrSize, int patchSize, int doHalfSIFT, int maxAngNum, double th, int addUpRight,
                             std::vector<modsx_region> *out);
int reproject_regions(modsx_region *regs, int n, const double *H, int orig_w, int orig_h);
int reproject_regions_box(modsx_region *regs, int n, const double *H, int orig_w, int orig_h, double boxk);
// The descriptor classes one step carries (modsx_pair_params / modsx_ladder_step n_desc, desc_types, desc_ratios resolved):
// `Descriptors=` and `FGINNThreshold=` of a [DetectorN] section.  half(): the step orients with doHalfSIFT = true
// (imagerepresentation.cpp:693-706, 1259-1264).
struct DescSet {
  int n = 1;
  int type[MODSX_MAX_DESC] = {MODSX_DESC_ROOT_SIFT, 0, 0, 0};
  double ratio[MODSX_MAX_DESC] = {0, 0, 0, 0};
  bool forceHalf = false;   // a caller that takes only the first class of a longer list keeps the list's orientation mode
  bool half() const { for (int i = 0; i < n; i++) if (type[i] >= 2) return true; return forceHalf; }
  int packed() const { int p = 0; for (int i = 0; i < n; i++) p |= (type[i] & 15) << (4 * i); return p; }
};
// rc = MODSX_ERR_ARG for a type outside 0..3, a type listed twice or n_desc outside 0..4
int resolve_descs(const modsx_pair_params &pp, const modsx_ladder_step *st, DescSet &ds);
void desc_class_order(const DescSet &ds, int *ord);   // the classes in descriptor-name order (GetCorresponcesVector)
// devU8x (optional): devU8x[k][i] receives the u8 descriptors of class k + 1 of image i (default: c->descU8x[k][i])
int describe_batch(modsx_ctx *c, const modsx_image *const *imgs, int n, const std::vector<modsx_region> *regs,
                   double mrSize, int patchSize, int fast, int photoNorm, int descType, double maxBin,
                   float *const *descHost, float *const *devF, uint8_t *const *devU8, const DescSet *ds = nullptr,
                   uint8_t *const *const *devU8x = nullptr);
struct ProfScopeFwd;
int set_vs_pars(const double *scale_set, int ns, const double *tilt_set, int nt, double phi_base, double InitSigma,
                int doBlur, modsx_view *par, int cap, modsx_view *prev, int *nprev, int cap_prev);
// slot < 0: the view is a fresh allocation owned by the returned image (public API).  slot in [0, MAXB): the view lives in
// the context's pooled buffer `slot`, nothing is allocated or freed and the call does not synchronise the stream (the
// per-view loop: hipMalloc / hipFree synchronise the whole device, i.e. every other context's stream as well).
int synth_view(modsx_ctx *c, const modsx_image *gray, const modsx_view &v, modsx_image **out, double *H, int *identity,
               int slot = -1);
// the per-view loop over (source image, view) items: one launch set may mix the views of several images
int detect_describe_items(modsx_ctx *c, const modsx_image *const *itemImg, const int *itemView, int nitems, const modsx_view *views,
                          const modsx_pair_params &pp, std::vector<modsx_region> &regs,
                          float *devF, uint8_t *devU8, size_t devCapRegions, float *hostDesc, int *itemCounts,
                          const DescSet *ds = nullptr, uint8_t *const *devU8x = nullptr);
int detect_describe_views(modsx_ctx *c, const modsx_image *gray, const modsx_view *views, int nv,
                          const modsx_pair_params &pp, int view_begin, int view_step, std::vector<modsx_region> &regs,
                          float *devF, uint8_t *devU8, size_t devCapRegions, float *hostDesc, int *viewCounts,
                          const DescSet *ds = nullptr, uint8_t *const *devU8x = nullptr);
void ctx_worker_stop(modsx_ctx *c);   // joins the context's host thread, if it has one (ctx_destroy)
void rebase_ids(std::vector<modsx_region> &regs, const int *viewCounts, int nv, size_t base);
struct VerifyTask;
// defer != nullptr (one-step ladders only): the tentatives are matched and handed back unverified
int match_ladder(modsx_ctx *c, const modsx_image *img1, const modsx_image *img2, const modsx_ladder_step *steps, int nsteps,
                 int min_matches, const modsx_pair_params &pp, modsx_pair_result *res, int *steps_done, VerifyTask *defer = nullptr,
                 modsx_comm *comm = nullptr, int owner = -1);
int match_pair_views(modsx_ctx *c, const modsx_image *img1, const modsx_image *img2, const modsx_view *views, int nv,
                     const modsx_pair_params &pp, modsx_pair_result *res, VerifyTask *defer = nullptr);
void prof_begin(modsx_ctx *c, int cls, double work, size_t *slot);
void prof_end(modsx_ctx *c, size_t slot);
bool prof_reserve(modsx_ctx *c, int cls, double work, hipEvent_t *ev2);
// a sharded match (engine_shard.hip): this rank owns the query rows [lo, lo + per) of n1_total
struct MatchShard { void *comm; int world, per, n1_total, lo; };
// the lane's (per + 1)-row blocks (header row + result rows), grown by agreement; then header + all-gather + download + wait
int match_shard_begin(modsx_ctx *c, const MatchShard &sh, mx::MatchRow **blk);
int match_shard_gather(modsx_ctx *c, const MatchShard &sh, int local_rc, mx::MatchRow *host);
int detect_describe_views_sharded(modsx_ctx *c, modsx_comm *cm, const modsx_image *img, const modsx_view *views, int nv,
                                  const modsx_pair_params &pp, const DescSet &ds, std::vector<modsx_region> &regs,
                                  DevBuf *const *descAcc, const size_t *base, int *viewCounts);
int detect_describe_items_sharded(modsx_ctx *c, modsx_comm *cm, const modsx_image *const *imgs, int nimg, const modsx_view *views,
                                  int nviews, const modsx_pair_params &pp, const DescSet &ds, std::vector<modsx_region> &regs,
                                  DevBuf *const *descAcc, const size_t *base, int *itemCounts, const unsigned char *wantImg = nullptr,
                                  std::vector<size_t> *regStart = nullptr, double **devPos = nullptr, std::vector<double> *kpRows = nullptr,
                                  const int *ownerImg = nullptr);   // ownerImg[j]: the one rank that reads image j's rows (owner-only exchange)
void rows_to_tentatives(const MatchRow *rows, int n1, int nn, std::vector<modsx_tentative> &o);
int comm_rank(const modsx_comm *cm);
int comm_world(const modsx_comm *cm);
int comm_same_value(modsx_ctx *c, modsx_comm *cm, int value, const char *what);   // one 4-byte all-gather; an error on every rank when they differ
int match_sharded(modsx_ctx *c, modsx_comm *cm, const uint8_t *d1, int n1, const uint8_t *d2, int n2, const double *pos2Host,
                  double ratioT, double contradDist, int nn, std::vector<modsx_tentative> &out);
int match_device_batch(modsx_ctx *c, int nb, const uint8_t *const *d1, const int *n1, const uint8_t *const *d2, const int *n2,
                       const double *const *pos2Host, double ratioT, double contradDist, int nn,
                       std::vector<modsx_tentative> *out, const MatchShard *shard, const double *const *pos2Dev = nullptr);
int match_device(modsx_ctx *c, const uint8_t *d1, int n1, const uint8_t *d2, int n2, const double *pos2Host,
                 double ratioT, double contradDist, int nn, std::vector<modsx_tentative> &out);
int match_host_desc(modsx_ctx *c, const float *desc1, int n1, const float *desc2, int n2, const double *pos2,
                    double ratioT, double contradDist, int nn, std::vector<modsx_tentative> &out);
// The region list a tentative's indices refer to: the per-class lists of one image in the order GetCorresponcesVector walks
// the classes (descriptor name, then detector name), as segments -- two descriptor classes of one detector share their
// regions, so the concatenation is never materialised.
struct RegList {
  const modsx_region *p[2 * MODSX_MAX_DESC];
  size_t end[2 * MODSX_MAX_DESC];   // running end index of every segment
  int n = 0;
  void clear() { n = 0; }
  void add(const modsx_region *q, size_t cnt) { p[n] = q; end[n] = (n ? end[n - 1] : 0) + cnt; n++; }
  void add(const std::vector<modsx_region> &v) { add(v.data(), v.size()); }
  size_t size() const { return n ? end[n - 1] : 0; }
  const modsx_region &operator[](size_t i) const {
    int k = 0;
    while (k < n - 1 && i >= end[k]) k++;     // an index past size() is a caller bug: it lands in the last segment and trips the check
    if (i >= end[k]) abort();
    return p[k][i - (k ? end[k - 1] : 0)];
  }
};
void verify_tentatives(const RegList &r1, const RegList &r2, const std::vector<modsx_tentative> &tents,
                       const modsx_pair_params &pp, modsx_pair_result *res);
void verify_tentatives(const std::vector<modsx_region> &r1, const std::vector<modsx_region> &r2,
                       const std::vector<modsx_tentative> &tents, const modsx_pair_params &pp, modsx_pair_result *res);
void verify_tentatives_kp(const double *kp1, size_t n1, const double *kp2, size_t n2, const std::vector<modsx_tentative> &tents,
                          const modsx_pair_params &pp, modsx_pair_result *res);
// A pair whose tentatives are matched but not yet verified: modsx_match_pairs hands these to helper threads so that a
// context's stream is fed the next group while the host runs DuplicateFiltering + LO-RANSAC of this one.  The task owns the
// region vectors its two lists point into (moving a vector keeps its heap block, so the segments stay valid).
struct VerifyTask {
  std::vector<std::vector<modsx_region>> own;
  RegList l1, l2;
  std::vector<modsx_tentative> tents;
  modsx_pair_result *res = nullptr;
  int dev = 0;        // the device of the context that produced the task: the verifier's device-side loops (rFtH counting) run there
  VerifyTask() = default;
  VerifyTask(VerifyTask &&) = default;
  VerifyTask &operator=(VerifyTask &&) = default;
  VerifyTask(const VerifyTask &) = delete;              // l1 / l2 point into `own`: a copy would dangle
  VerifyTask &operator=(const VerifyTask &) = delete;
};
// deferred == nullptr: verification runs inline, pair by pair.  Otherwise the G matching problems share the matcher's
// launches and one VerifyTask per pair is appended to *deferred instead of being verified.
int match_pair_group(modsx_ctx *c, const modsx_image *const *imgs1, const modsx_image *const *imgs2, int G,
                     const modsx_pair_params &pp, modsx_pair_result *res, std::vector<VerifyTask> *deferred = nullptr);
int match_pair(modsx_ctx *c, const modsx_image *img1, const modsx_image *img2, const modsx_pair_params &pp,
               modsx_pair_result *res);

// filters.cpp / ransac.cpp (host)
int duplicate_filtering(const double *pts, const double *key, int T, double r, int do_sort, int *order,
                        unsigned char *keep);
int ransac_h(const double *u, int len, double th, double conf, int max_sam, double *H, unsigned char *inl, int *data_out,
             int oriented_constraint, int doSymCheck, unsigned seed0, double *scoreJ, int error_type = 0);
void hds_sym(const double *u, const double *H, double *p, int len, bool takeMax);
void hds_sym_setup(const double *H, double *Hinv, double *H1);                 // hds_sym = setup + with
void hds_sym_with(const double *u, const double *Hinv, const double *H1, double *p, int len, bool takeMax);
int save_regions(const char *path, const modsx_region_class *classes, int nclasses);
int load_regions(const char *path, const char *det_name, const char *desc_name, std::vector<modsx_region> &regs,
                 std::vector<float> &desc, int *dim, std::string *found_det, std::string *found_desc);
int detect_msers_host(const uint8_t *u8, int rows, int cols, const modsx_mser_params &par, double tilt, double zoom,
                      std::vector<modsx_keypoint> &out);
// devOrder[i] / devStart[i]: view i's pixel offsets in (grey level, raster) order and the 257 level starts when the device sorted the
// view (launch_mser_sort), or null: the host sorts
int detect_msers_views(const uint8_t *const *u8, const int *rows, const int *cols, int n, const modsx_mser_params &par,
                       const double *tilts, const double *zooms, std::vector<modsx_keypoint> *out, const int *const *devOrder = nullptr,
                       const int *const *devStart = nullptr);
// fn(0) .. fn(n - 1) on the process-wide host worker pool (MODSX_HOST_THREADS, default min(hardware threads, 64)) + the caller
void host_parallel_for(int n, const std::function<void(int)> &fn, bool light = false);
void host_light_pool(bool on);   // this thread's short host loops may use the pool regardless of the sets in flight
void host_set_enter();
void host_set_leave();
inline void host_parallel_light(int n, const std::function<void(int)> &fn) { host_parallel_for(n, fn, true); }
int ransac_f(const double *u, int len, double th, double conf, int max_sam, double *F, unsigned char *inl, int *data_out,
             int do_lo, unsigned inlLimit, int error_type, int doSymCheck, unsigned seed0);
int loransac_f(const double *pts, const double *laf1, const double *laf2, int T, double err_threshold, double confidence,
               int max_samples, int lo, double LAFCoef, int doSymmCheck, int error_type, unsigned seed, double *F,
               unsigned char *inl, unsigned char *keep, int *data_out3);
int loransac_h(const double *pts, const double *laf1, const double *laf2, int T, double err_threshold,
               double confidence, int max_samples, int lo, double HLAFCoef, int doSymmCheck, unsigned seed, double *H,
               double *Hraw, unsigned char *inl, unsigned char *keep, int *data_out, int error_type = 0);
}  // namespace mx
