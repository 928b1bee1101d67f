// engine_shard.hip -- view-sharded multi-GPU path: one process per GPU, RCCL collectives on device buffers.
//
// Reference unit of parallelism: the synthesised views of an image are independent from GenerateSynthImageCorr through
// DescribeRegions (the `#pragma omp parallel for` over views, imagerepresentation.cpp:612-622) and meet only in the ordered
// concatenation of AddRegions (:2044-2045, ids re-based by AddRegionsToList :588-600).  Here view v belongs to rank
// v mod world; ONE exchange step per image side rebuilds the reference's list on every rank:
//   ncclAllGather of the per-view counts (nviews ints per rank), then ncclAllGather of the padded row blocks
//   (row = modsx_region, 200 B, + the 128 u8 descriptor bytes = 328 B), both on the context's stream, device to device
//   (xGMI between the GPUs of a node); a gather kernel then writes regions and descriptors in (view, detection) order.
// Matching splits the QUERY rows (rank r takes [r n1 / W, (r+1) n1 / W) against all of image 2); the per-query result rows
// of the device matcher (32 B each) are all-gathered before the host turns them into tentatives, so every rank ends with
// the full list in query order and no host object crosses ranks.  DuplicateFiltering + LO-RANSAC run on the owner rank.
//
// RCCL is bound at run time (dlopen; an already loaded librccl -- e.g. the one torch ships -- is reused), so libmodsx has
// no link-time dependency on it and single-GPU users never load it.
#include <dlfcn.h>
#include <algorithm>
#include <rccl/rccl.h>
#include "engine_api.hpp"

namespace mx {

struct RcclApi {
  void *h = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*GetVersion)(int *) = nullptr;
  const char *(*GetErrorString)(ncclResult_t) = nullptr;
};
static RcclApi g_rccl;
static bool rccl_load() {
  if (g_rccl.h) return true;
  const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
  void *h = nullptr;
  for (const char *n : names) if ((h = dlopen(n, RTLD_NOW | RTLD_NOLOAD | RTLD_GLOBAL))) break;   // reuse a loaded one
  if (!h) for (const char *n : names) if ((h = dlopen(n, RTLD_NOW | RTLD_GLOBAL))) break;
  if (!h) { set_error("RCCL (librccl.so.1) not found: the view-sharded path needs it"); return false; }
  RcclApi a;
  a.h = h;
  a.GetUniqueId = (decltype(a.GetUniqueId))dlsym(h, "ncclGetUniqueId");
  a.CommInitRank = (decltype(a.CommInitRank))dlsym(h, "ncclCommInitRank");
  a.CommDestroy = (decltype(a.CommDestroy))dlsym(h, "ncclCommDestroy");
  a.AllGather = (decltype(a.AllGather))dlsym(h, "ncclAllGather");
  a.GetVersion = (decltype(a.GetVersion))dlsym(h, "ncclGetVersion");
  a.GetErrorString = (decltype(a.GetErrorString))dlsym(h, "ncclGetErrorString");
  if (!a.GetUniqueId || !a.CommInitRank || !a.CommDestroy || !a.AllGather || !a.GetVersion || !a.GetErrorString) {
    set_error("librccl lacks an expected symbol");
    return false;
  }
  g_rccl = a;
  return true;
}
#define MX_NCCL(expr)                                                                                   \
  do {                                                                                                  \
    ncclResult_t r_ = (expr);                                                                           \
    if (r_ != ncclSuccess) {                                                                            \
      mx::set_error(std::string(#expr) + ": " + g_rccl.GetErrorString(r_));                            \
      return MODSX_ERR_DEVICE;                                                                          \
    }                                                                                                   \
  } while (0)

}  // namespace mx

struct modsx_comm {
  ncclComm_t comm = nullptr;
  int rank = 0, world = 1, version = 0;
  mx::DevBuf rowsLocal, rowsAll, regsDev, cntDev, matchAll;
  mx::PinBuf hRegs, hCnt;
  long bytes_gathered = 0, collectives = 0;
};

namespace mx {

constexpr int REG_B = (int)sizeof(modsx_region), ROW_B = REG_B + 128;
static_assert(sizeof(modsx_region) == 200, "region rows are 200 + 128 bytes on the wire");

// rows[i] = region i (REG_B bytes, 8-byte words) followed by its 128 descriptor bytes (16-byte words)
__global__ void k_pack_rows(const unsigned char *regs, const unsigned char *desc, int n, unsigned char *rows) {
  const int i = blockIdx.x * 8 + (threadIdx.x >> 5), l = threadIdx.x & 31;
  if (i >= n) return;
  unsigned char *dst = rows + (size_t)i * ROW_B;
  if (l < REG_B / 8) reinterpret_cast<uint64_t *>(dst)[l] = reinterpret_cast<const uint64_t *>(regs + (size_t)i * REG_B)[l];
  if (l < 16) reinterpret_cast<uint64_t *>(dst + REG_B)[l] = reinterpret_cast<const uint64_t *>(desc + (size_t)i * 128)[l];
}
// out row j comes from gathered row src[j]; split back into the region array and the descriptor matrix
__global__ void k_unpack_rows(const unsigned char *rowsAll, const int *src, int n, unsigned char *regs, unsigned char *desc) {
  const int j = blockIdx.x * 8 + (threadIdx.x >> 5), l = threadIdx.x & 31;
  if (j >= n) return;
  const unsigned char *s = rowsAll + (size_t)src[j] * ROW_B;
  if (l < REG_B / 8) reinterpret_cast<uint64_t *>(regs + (size_t)j * REG_B)[l] = reinterpret_cast<const uint64_t *>(s)[l];
  if (l < 16) reinterpret_cast<uint64_t *>(desc + (size_t)j * 128)[l] = reinterpret_cast<const uint64_t *>(s + REG_B)[l];
}

// Position of every row of the reference's list inside the all-gathered buffer: view v is rank v mod W's, at that rank's
// running offset; rank r's block starts at r * maxrows.  counts[r * nviews + v].  Returns the list length.
int view_block_order(const int *counts, int world, int nviews, int maxrows, std::vector<int> &src) {
  src.clear();
  std::vector<int> run(world, 0);
  for (int v = 0; v < nviews; v++) {
    const int r = v % world, c = counts[r * nviews + v];
    for (int k = 0; k < c; k++) src.push_back(r * maxrows + run[r] + k);
    run[r] += c;
  }
  return (int)src.size();
}

// The sharded SynthDetectDescribeKeypoints: regions of ALL views in reference order on every rank (ids re-based),
// u8 descriptors in `descOut` (device, grown as needed).
int detect_describe_views_sharded(modsx_ctx *c, modsx_comm *cm, const modsx_image *img, const modsx_view *views, int nv,
                                  const modsx_pair_params &pp, std::vector<modsx_region> &regs, DevBuf &descOut, int *viewCounts) {
  hipStream_t s = c->stream;
  const int W = cm->world, R = cm->rank;
  std::vector<modsx_region> local;
  std::vector<int> cnt(nv, 0);
  size_t cap = (size_t)1 << 15;
  for (;;) {
    if (!c->descAllU8b[0].ensure(cap * 128)) return MODSX_ERR_NOMEM;
    int rc = detect_describe_views(c, img, views, nv, pp, R, W, local, nullptr, (uint8_t *)c->descAllU8b[0].p, cap, nullptr,
                                   cnt.data());
    if (rc == MODSX_ERR_CAPACITY && cap < ((size_t)1 << 24)) { cap *= 4; continue; }
    if (rc) return rc;
    break;
  }
  const int nloc = (int)local.size();
  // 1. counts of every rank
  if (!cm->cntDev.ensure((size_t)(W + 1) * nv * 4) || !cm->hCnt.ensure((size_t)(W + 1) * nv * 4)) return MODSX_ERR_NOMEM;
  int *hc = (int *)cm->hCnt.p, *dc = (int *)cm->cntDev.p;
  memcpy(hc, cnt.data(), (size_t)nv * 4);
  MX_HIP(hipMemcpyAsync(dc, hc, (size_t)nv * 4, hipMemcpyHostToDevice, s));
  MX_NCCL(g_rccl.AllGather(dc, dc + nv, nv, ncclInt32, cm->comm, s));
  MX_HIP(hipMemcpyAsync(hc + nv, dc + nv, (size_t)W * nv * 4, hipMemcpyDeviceToHost, s));
  // 2. local rows = region + descriptor (regions go up once; descriptors never left the device)
  if (!cm->hRegs.ensure((size_t)std::max(1, nloc) * REG_B) || !cm->regsDev.ensure((size_t)std::max(1, nloc) * REG_B)) return MODSX_ERR_NOMEM;
  if (nloc) {
    memcpy(cm->hRegs.p, local.data(), (size_t)nloc * REG_B);
    MX_HIP(hipMemcpyAsync(cm->regsDev.p, cm->hRegs.p, (size_t)nloc * REG_B, hipMemcpyHostToDevice, s));
  }
  MX_HIP(hipStreamSynchronize(s));
  const int *all = hc + nv;
  int maxrows = 0;
  for (int r = 0; r < W; r++) {
    int t = 0;
    for (int v = 0; v < nv; v++) t += all[r * nv + v];
    maxrows = std::max(maxrows, t);
  }
  std::vector<int> src;
  const int N = view_block_order(all, W, nv, maxrows, src);
  if (viewCounts) for (int v = 0; v < nv; v++) viewCounts[v] = all[(v % W) * nv + v];
  regs.resize(N);
  if (!N) return MODSX_OK;
  if (!cm->rowsLocal.ensure((size_t)maxrows * ROW_B) || !cm->rowsAll.ensure((size_t)W * maxrows * ROW_B)) return MODSX_ERR_NOMEM;
  if (nloc) hipLaunchKernelGGL(k_pack_rows, dim3((nloc + 7) / 8), dim3(256), 0, s, (const unsigned char *)cm->regsDev.p,
                               (const unsigned char *)c->descAllU8b[0].p, nloc, (unsigned char *)cm->rowsLocal.p);
  // 3. the exchange: one all-gather of the padded blocks
  MX_NCCL(g_rccl.AllGather(cm->rowsLocal.p, cm->rowsAll.p, (size_t)maxrows * ROW_B, ncclUint8, cm->comm, s));
  cm->bytes_gathered += (long)W * maxrows * ROW_B; cm->collectives += 2;
  // 4. reference order on the device; regions come down, descriptors stay
  if (!c->misc.ensure((size_t)N * 4) || !descOut.ensure((size_t)N * 128) || !cm->regsDev.ensure((size_t)N * REG_B) ||
      !cm->hRegs.ensure((size_t)N * REG_B)) return MODSX_ERR_NOMEM;
  MX_HIP(hipMemcpyAsync(c->misc.p, src.data(), (size_t)N * 4, hipMemcpyHostToDevice, s));
  hipLaunchKernelGGL(k_unpack_rows, dim3((N + 7) / 8), dim3(256), 0, s, (const unsigned char *)cm->rowsAll.p, (const int *)c->misc.p, N,
                     (unsigned char *)cm->regsDev.p, (unsigned char *)descOut.p);
  MX_HIP(hipMemcpyAsync(cm->hRegs.p, cm->regsDev.p, (size_t)N * REG_B, hipMemcpyDeviceToHost, s));
  MX_HIP(hipStreamSynchronize(s));   // also keeps `src` alive until the upload is done
  MX_HIP(hipGetLastError());
  memcpy(regs.data(), cm->hRegs.p, (size_t)N * REG_B);
  std::vector<int> vc(nv);
  for (int v = 0; v < nv; v++) vc[v] = all[(v % W) * nv + v];
  rebase_ids(regs, vc.data(), nv, 0);
  return MODSX_OK;
}

// MatchFlannFGINN with the query rows split over the ranks (d1 / d2 hold ALL descriptors on every rank)
int match_sharded(modsx_ctx *c, modsx_comm *cm, const uint8_t *d1, int n1, const uint8_t *d2, int n2, const double *pos2Host,
                  double ratioT, double contradDist, int nn, std::vector<modsx_tentative> &out) {
  out.clear();
  if (n1 <= 0 || n2 <= 0) return MODSX_OK;
  const int W = cm->world, R = cm->rank;
  const int per = (n1 + W - 1) / W;                 // rows per rank (the last ranks may hold fewer or none)
  const int lo = std::min(n1, R * per), hi = std::min(n1, lo + per);
  MatchShard sh;
  sh.comm = cm; sh.per = per; sh.n1_total = n1; sh.lo = lo;
  return match_device_batch(c, 1, &d1, &n1, &d2, &n2, &pos2Host, ratioT, contradDist, nn, &out, &sh);
}

// called by match_device_batch between the matcher launches and the D2H of the result rows
int match_shard_gather(modsx_ctx *c, const MatchShard &sh, MatchRow *rowsLocal, MatchRow **rowsAll) {
  modsx_comm *cm = (modsx_comm *)sh.comm;
  const size_t blk = (size_t)sh.per * sizeof(MatchRow);
  if (!cm->matchAll.ensure(blk * cm->world)) return MODSX_ERR_NOMEM;
  MX_NCCL(g_rccl.AllGather(rowsLocal, cm->matchAll.p, blk, ncclUint8, cm->comm, c->stream));
  cm->bytes_gathered += (long)blk * cm->world; cm->collectives++;
  *rowsAll = (MatchRow *)cm->matchAll.p;
  return MODSX_OK;
}

int match_pair_views_sharded(modsx_ctx *c, modsx_comm *cm, const modsx_image *img1, const modsx_image *img2, const modsx_view *views,
                             int nv, const modsx_pair_params &pp, int owner, modsx_pair_result *res) {
  memset(res, 0, sizeof *res);
  for (int i = 0; i < 9; i++) res->H[i] = -1;
  std::vector<modsx_region> r1, r2;
  int rc = detect_describe_views_sharded(c, cm, img1, views, nv, pp, r1, c->descAllU8[0], nullptr);
  if (rc) return rc;
  rc = detect_describe_views_sharded(c, cm, img2, views, nv, pp, r2, c->descAllU8[1], nullptr);
  if (rc) return rc;
  res->n_regions1 = (int)r1.size(); res->n_regions2 = (int)r2.size();
  std::vector<double> pos2(r2.size() * 2 + 2);
  for (size_t i = 0; i < r2.size(); i++) { pos2[2 * i] = r2[i].reproj_kp.x; pos2[2 * i + 1] = r2[i].reproj_kp.y; }
  std::vector<modsx_tentative> tents;
  rc = match_sharded(c, cm, (uint8_t *)c->descAllU8[0].p, (int)r1.size(), (uint8_t *)c->descAllU8[1].p, (int)r2.size(), pos2.data(),
                     pp.match_ratio, pp.contradDist, pp.nn, tents);
  if (rc) return rc;
  res->n_tentatives = (int)tents.size();
  if (owner < 0 || owner == cm->rank) verify_tentatives(r1, r2, tents, pp, res);   // DuplicateFiltering + LO-RANSAC: sequential, tiny
  prof_collect(c);
  return MODSX_OK;
}

}  // namespace mx

using namespace mx;
extern "C" {

int modsx_comm_unique_id(void *id128) {
  if (!id128) { mx::set_error("modsx_comm_unique_id: null"); return MODSX_ERR_ARG; }
  if (!rccl_load()) return MODSX_ERR_DEVICE;
  ncclUniqueId id;
  MX_NCCL(g_rccl.GetUniqueId(&id));
  memcpy(id128, &id, sizeof id);
  return MODSX_OK;
}

modsx_comm *modsx_comm_create(modsx_ctx *ctx, const void *id128, int rank, int world) {
  if (!ctx || !id128 || world < 1 || rank < 0 || rank >= world) { mx::set_error("modsx_comm_create: bad argument"); return nullptr; }
  if (!rccl_load()) return nullptr;
  hipSetDevice(ctx->dev);
  modsx_comm *cm = new modsx_comm();
  cm->rank = rank; cm->world = world;
  ncclUniqueId id;
  memcpy(&id, id128, sizeof id);
  ncclResult_t r = g_rccl.CommInitRank(&cm->comm, world, id, rank);
  if (r != ncclSuccess) { mx::set_error(std::string("ncclCommInitRank: ") + g_rccl.GetErrorString(r)); delete cm; return nullptr; }
  g_rccl.GetVersion(&cm->version);
  return cm;
}

void modsx_comm_destroy(modsx_comm *cm) {
  if (!cm) return;
  if (cm->comm) g_rccl.CommDestroy(cm->comm);
  DevBuf *bufs[] = {&cm->rowsLocal, &cm->rowsAll, &cm->regsDev, &cm->cntDev, &cm->matchAll};
  for (DevBuf *b : bufs) b->release();
  cm->hRegs.release(); cm->hCnt.release();
  delete cm;
}

int modsx_comm_info(const modsx_comm *cm, int *rank, int *world, int *rccl_version, long *bytes_gathered, long *collectives) {
  if (!cm) { mx::set_error("modsx_comm_info: null"); return MODSX_ERR_ARG; }
  if (rank) *rank = cm->rank;
  if (world) *world = cm->world;
  if (rccl_version) *rccl_version = cm->version;
  if (bytes_gathered) *bytes_gathered = cm->bytes_gathered;
  if (collectives) *collectives = cm->collectives;
  return MODSX_OK;
}

int modsx_view_block_order(const int *counts, int world, int nviews, int *src, int cap, int *maxrows_out) {
  if (!counts || world < 1 || nviews < 1) { mx::set_error("modsx_view_block_order: bad argument"); return MODSX_ERR_ARG; }
  int maxrows = 0;
  for (int r = 0; r < world; r++) {
    int t = 0;
    for (int v = 0; v < nviews; v++) t += counts[r * nviews + v];
    maxrows = std::max(maxrows, t);
  }
  std::vector<int> s;
  const int n = view_block_order(counts, world, nviews, maxrows, s);
  if (maxrows_out) *maxrows_out = maxrows;
  if (src) for (int i = 0; i < n && i < cap; i++) src[i] = s[i];
  return n;
}

int modsx_detect_describe_views_sharded(modsx_ctx *ctx, modsx_comm *comm, const modsx_image *img, const modsx_view *views, int nviews,
                                        const modsx_pair_params *par, modsx_region **regs, void **dev_desc_u8, int *view_counts) {
  if (!ctx || !comm || !img || !views || !par || !regs || nviews <= 0) { mx::set_error("modsx_detect_describe_views_sharded: bad argument"); return MODSX_ERR_ARG; }
  hipSetDevice(ctx->dev);
  std::vector<modsx_region> r;
  int rc = detect_describe_views_sharded(ctx, comm, img, views, nviews, *par, r, ctx->descAllU8[0], view_counts);
  if (rc) return rc;
  if (dev_desc_u8) *dev_desc_u8 = ctx->descAllU8[0].p;
  modsx_region *p = (modsx_region *)malloc(sizeof(modsx_region) * std::max<size_t>(1, r.size()));
  if (!r.empty()) memcpy(p, r.data(), sizeof(modsx_region) * r.size());
  *regs = p;
  return (int)r.size();
}

int modsx_match_fginn_sharded(modsx_ctx *ctx, modsx_comm *comm, const void *dev_desc1_u8, int n1, const void *dev_desc2_u8, int n2,
                              const double *pos2, double ratio, double contradDist, int nn, modsx_tentative **out) {
  if (!ctx || !comm || !out || n1 < 0 || n2 < 0 || (n1 > 0 && !dev_desc1_u8) || (n2 > 0 && (!dev_desc2_u8 || !pos2))) {
    mx::set_error("modsx_match_fginn_sharded: bad argument");
    return MODSX_ERR_ARG;
  }
  hipSetDevice(ctx->dev);
  std::vector<modsx_tentative> t;
  int rc = match_sharded(ctx, comm, (const uint8_t *)dev_desc1_u8, n1, (const uint8_t *)dev_desc2_u8, n2, pos2, ratio, contradDist, nn, t);
  if (rc) return rc;
  modsx_tentative *p = (modsx_tentative *)malloc(sizeof(modsx_tentative) * std::max<size_t>(1, t.size()));
  if (!t.empty()) memcpy(p, t.data(), sizeof(modsx_tentative) * t.size());
  *out = p;
  return (int)t.size();
}

int modsx_match_pair_views_sharded(modsx_ctx *ctx, modsx_comm *comm, const modsx_image *img1, const modsx_image *img2,
                                   const modsx_view *views, int nviews, const modsx_pair_params *par, int owner, modsx_pair_result *res) {
  if (!ctx || !comm || !img1 || !img2 || !views || !par || !res || nviews <= 0) { mx::set_error("modsx_match_pair_views_sharded: bad argument"); return MODSX_ERR_ARG; }
  hipSetDevice(ctx->dev);
  return match_pair_views_sharded(ctx, comm, img1, img2, views, nviews, *par, owner, res);
}

}  // extern "C"
