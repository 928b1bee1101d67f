// engine.hip -- host orchestration of the device path (C++ above HIP, below the C ABI).
//
// Mirrors, stage by stage, what the reference does per synthesised view in
// ImageRepresentation::SynthDetectDescribeKeypoints (imagerepresentation.cpp:603-2047) and per pair
// in mods.cpp:229-415.  Dense per-pixel / per-patch work runs in the HIP kernels; the host keeps the
// order-dependent bookkeeping (detection order, first-come octaveMap claims, std::sort) and the few
// libm transcendentals (powf / cos / sin / exp) so they are evaluated by the same libm as on the CPU path.
#include <atomic>
#include <math.h>
#include <stdio.h>
#include <algorithm>
#include <chrono>
#include <map>
#include <unordered_set>
#include <mutex>
#include <dlfcn.h>
#include <string>
#include "engine_api.hpp"

#ifdef MODSX_DUP_BUILD
namespace mx {
int dup_count(int cls) {
  static const long mask = getenv("MODSX_DUP") ? strtol(getenv("MODSX_DUP"), nullptr, 0) : 0;
  return 1 + (int)((mask >> cls) & 1);
}
}  // namespace mx
#endif
namespace mx {

static thread_local std::string g_err;
void set_error(const std::string &s) { g_err = s; }
static std::atomic<int> g_busyContexts{0};
bool gpu_shared() { return g_busyContexts.load(std::memory_order_relaxed) > 1; }
CtxBusy::CtxBusy(modsx_ctx *ctx) : c(ctx) {
  if (c && c->busyDepth++ == 0) {
    g_busyContexts.fetch_add(1, std::memory_order_relaxed);
    // how this call waits at its stage boundaries is decided once, here (MODSX_HOST_WAIT=auto follows the process' CPU load): a
    // context that goes back from the flag word to the runtime's wait first lets the runtime catch up with its stream -- the
    // runtime has not been asked about that stream since the context left its wait, and the small copies it does with the CPU
    // (rocprofv3 run of the 16-stream bench: SIGSEGV inside hipMemcpyAsync) rest on its own picture of what has completed
    const bool rt = host_wait_runtime() || !c->hFlag;
    if (rt && !c->waitRuntime) hipStreamSynchronize(c->stream);
    c->waitRuntime = rt;
  }
}
CtxBusy::~CtxBusy() { if (c && --c->busyDepth == 0) g_busyContexts.fetch_sub(1, std::memory_order_relaxed); }
const char *last_error() { return g_err.c_str(); }

bool DevBuf::ensure(size_t bytes) {
  if (bytes <= cap) return true;
  release();
  size_t want = bytes + bytes / 4 + 256;
  if (hipMalloc(&p, want) != hipSuccess) { p = nullptr; cap = 0; set_error("hipMalloc failed"); return false; }
  cap = want;
  return true;
}
void DevBuf::release() { if (p) hipFree(p); p = nullptr; cap = 0; }
bool PinBuf::ensure(size_t bytes) {
  if (bytes <= cap) return true;
  release();
  size_t want = bytes + bytes / 4 + 256;
  if (hipHostMalloc(&p, want, hipHostMallocDefault) != hipSuccess) { p = nullptr; cap = 0; set_error("hipHostMalloc failed"); return false; }
  cap = want;
  return true;
}
void PinBuf::release() { if (p) hipHostFree(p); p = nullptr; cap = 0; }

// ---- stage boundaries without the runtime's wait machinery -------------------------------------------------------------------------
// What a stage boundary costs the HOST was measured in isolation (tools/ubench/host_sync.hip, 16 threads with a stream each, 8
// kernels + results + wait per iteration, waits of ~3 ms): with hipMemcpyAsync + hipStreamSynchronize 500 us of CPU per
// iteration (340 in the caller: the runtime watches the signal for 200 us before it sleeps in the kernel driver, and 160 in the
// runtime's own threads), 3 ms when an H2D hipMemcpyAsync is part of the iteration (the wait then spins to the end) -- against
// 73 us when the last kernel writes a sequence word into pinned memory and the thread looks at it between naps.  So the hot path
// neither copies nor waits through the runtime:
//   ctx_copy   device <-> PINNED host memory by a kernel on the context's stream (system-scope accesses on the host side);
//   ctx_sync   a one-lane kernel writes the context's next sequence number to its pinned flag word behind everything issued so
//              far; the thread spins for a few microseconds, then naps (10 us growing to 100 us, timer slack 2 us) until it
//              sees it.  After 20 s without it hipStreamSynchronize is asked (a faulted queue never writes the flag).
// In the pipeline the gain is smaller than in isolation (host_wait_runtime below says when it is taken): most of a pair's host CPU
// is the engine's own loops in the context threads (19-20 of 25 ms), not the waits.
}  // namespace mx
#include <sys/prctl.h>
#include <time.h>
namespace mx {
__global__ void k_flag(unsigned *flag, unsigned seq) {
  __threadfence_system();
  __hip_atomic_store(flag, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
// HOST_SRC: the source is pinned host memory -- a system-scope acquire in front of the loads drops whatever an earlier launch left
// in the caches of those addresses; otherwise the destination is, and a system-scope release follows the stores.  16 bytes per lane
// (1 KB per wave instruction: the PCIe link sees whole cache lines; 8-byte system-scope atomics per lane moved a 2 MB table at a
// fifth of the rate and cost the pipeline 10 %).
template <bool HOST_SRC>
__global__ __launch_bounds__(256) void k_copy_pinned(uint4 *dst, const uint4 *src, size_t n16, int tail) {
  if (HOST_SRC) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
  const size_t stride = (size_t)gridDim.x * 256;
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  for (; i + 3 * stride < n16; i += 4 * stride) {
    const uint4 a = src[i], b = src[i + stride], c = src[i + 2 * stride], d = src[i + 3 * stride];
    dst[i] = a; dst[i + stride] = b; dst[i + 2 * stride] = c; dst[i + 3 * stride] = d;
  }
  for (; i < n16; i += stride) dst[i] = src[i];
  if (tail && blockIdx.x == 0 && threadIdx.x < (unsigned)tail)
    reinterpret_cast<unsigned char *>(dst + n16)[threadIdx.x] = reinterpret_cast<const unsigned char *>(src + n16)[threadIdx.x];
  if (!HOST_SRC) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
}
// MODSX_HOST_WAIT = runtime | flag | auto (default).  Measured (16 contexts, one GPU, 16 CPUs): on the headline workload, which uses a
// third of the CPUs, the flag wait takes 0.0255 -> 0.0215 CPU-s per pair (the runtime's own threads 4.7 -> 2.1 ms, system time
// 3.7 -> 1.7 ms) and costs 1.2 % of the pairs/s (the naps wake tens of microseconds late at ~30 boundaries per pair); on the cviu
// ladder, whose MSER steps keep all 16 CPUs busy, it takes 0.244 -> 0.206 CPU-s per pair and GIVES 13 % (61.7 -> 70.0 pairs/s).
// So `auto` looks at what the process is short of: every 50 ms it compares the CPU time the process used with its allowance
// (cgroup quota / local ranks) -- above 80 % the host is the limit and the flag wait is taken, below 60 % the runtime's wait.
}  // namespace mx
#include <sys/resource.h>
namespace mx {
bool host_wait_runtime() {
  static const int forced = [] {
    const char *e = getenv("MODSX_HOST_WAIT");
    if (e && !strcmp(e, "runtime")) return 1;
    if (e && !strcmp(e, "flag")) return 0;
    if (e && !strcmp(e, "alternate")) return 2;      // test aid: every call of this function answers the other way (the transitions)
    return -1;
  }();
  if (forced == 2) { static std::atomic<unsigned> flip{0}; return (flip.fetch_add(1, std::memory_order_relaxed) & 1) != 0; }
  if (forced >= 0) return forced != 0;
  static const double allowance = (double)host_cpus_per_rank();
  static std::atomic<int> rt{allowance >= 8 ? 1 : 0};
  static std::atomic<long long> nextNs{0};
  static std::atomic<long long> lastWallNs{0}, lastCpuUs{0};
  const long long now = std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count();
  long long due = nextNs.load(std::memory_order_relaxed);
  if (now >= due && nextNs.compare_exchange_strong(due, now + 50000000LL, std::memory_order_relaxed)) {     // one thread per period
    rusage ru;
    getrusage(RUSAGE_SELF, &ru);
    const long long cpu = (ru.ru_utime.tv_sec + ru.ru_stime.tv_sec) * 1000000LL + ru.ru_utime.tv_usec + ru.ru_stime.tv_usec;
    const long long w0 = lastWallNs.exchange(now), c0 = lastCpuUs.exchange(cpu);
    if (w0 && now - w0 < 1000000000LL) {           // (a longer gap: the process was idle in between, the figure says nothing)
      const double load = (double)(cpu - c0) * 1e3 / (double)(now - w0) / allowance;
      if (load > 0.80) rt.store(0, std::memory_order_relaxed);
      else if (load < 0.60) rt.store(1, std::memory_order_relaxed);
    }
  }
  return rt.load(std::memory_order_relaxed) != 0;
}
hipError_t ctx_copy(modsx_ctx *c, void *dst, const void *src, size_t bytes, hipMemcpyKind kind) {
  if (!bytes) return hipSuccess;
  const bool h2d = kind == hipMemcpyHostToDevice;
  static const bool rtCopy = [] { const char *e = getenv("MODSX_HOST_COPY"); return e && !strcmp(e, "runtime"); }();
  // tables above 64 KB stay with the runtime's copy engines: as kernels on the stream they cost the pipeline 3 % (the describe
  // blobs and candidate lists are megabytes; a copy kernel holds the stream for tens of microseconds that the DMA engine overlaps)
  static const size_t rtAbove = getenv("MODSX_HOST_COPY_MAX") ? (size_t)atol(getenv("MODSX_HOST_COPY_MAX")) : 65536;
  if (c->waitRuntime || rtCopy || bytes > rtAbove || (!h2d && kind != hipMemcpyDeviceToHost) || (((uintptr_t)dst | (uintptr_t)src) & 15))
    return hipMemcpyAsync(dst, src, bytes, kind, c->stream);
  const size_t n16 = bytes >> 4;
  const int tail = (int)(bytes & 15);
  const unsigned grid = (unsigned)std::min<size_t>(std::max<size_t>((n16 + 1023) / 1024, 1), 512);
  if (h2d) hipLaunchKernelGGL(k_copy_pinned<true>, dim3(grid), dim3(256), 0, c->stream, (uint4 *)dst, (const uint4 *)src, n16, tail);
  else hipLaunchKernelGGL(k_copy_pinned<false>, dim3(grid), dim3(256), 0, c->stream, (uint4 *)dst, (const uint4 *)src, n16, tail);
  return hipGetLastError();
}
unsigned ctx_mark(modsx_ctx *c) {
  const unsigned seq = ++c->flagSeq;
  hipLaunchKernelGGL(k_flag, dim3(1), dim3(1), 0, c->stream, c->hFlag, seq);
  return seq;
}
bool ctx_mark_reached(modsx_ctx *c, unsigned seq) {
  return (int)(__atomic_load_n(c->hFlag, __ATOMIC_ACQUIRE) - seq) >= 0;
}
hipError_t ctx_wait_mark(modsx_ctx *c, unsigned seq) {
  static const int spinUs = getenv("MODSX_WAIT_SPIN_US") ? atoi(getenv("MODSX_WAIT_SPIN_US")) : 6;
  static const long napMax = getenv("MODSX_WAIT_NAP_MAX_US") ? atol(getenv("MODSX_WAIT_NAP_MAX_US")) * 1000 : 100000;
  if (ctx_mark_reached(c, seq)) return hipSuccess;
  const auto t0 = std::chrono::steady_clock::now();
  auto us_since = [&]() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count(); };
  for (int i = 0;; i++) {
    if (ctx_mark_reached(c, seq)) return hipSuccess;
    __builtin_ia32_pause();
    if ((i & 15) == 15 && us_since() >= spinUs) break;
  }
  static thread_local bool slack = false;
  if (!slack) { prctl(PR_SET_TIMERSLACK, 2000UL, 0, 0, 0); slack = true; }   // the naps below mean what they say (default slack: 50 us)
  // naps of 10 us growing to 100: 20-30 looks at a wait of 1-3 ms.  (Tried: one long nap sized from the last waits at the same call
  // site, then close looks -- under 16 streams a stage's wait varies too much, the thread woke late and the pairs/s fell by 3 %;
  // naps of 5 us throughout: the runtime's throughput for more CPU than the runtime's own wait takes.)
  long nap = std::min(10000L, napMax);
  for (int k = 0;; k++) {
    timespec ts = {0, nap};
    nanosleep(&ts, nullptr);
    if (ctx_mark_reached(c, seq)) return hipSuccess;
    if ((k & 1) && nap < napMax) nap = std::min(nap * 2, napMax);
    if ((k & 255) == 255 && us_since() > 20e6) break;
  }
  const hipError_t e = hipStreamSynchronize(c->stream);     // a queue that faulted never writes the flag: let the runtime say what happened
  if (e == hipSuccess && !ctx_mark_reached(c, seq)) return hipErrorUnknown;
  return e;
}
hipError_t ctx_sync(modsx_ctx *c) {
  if (c->waitRuntime || !c->hFlag) return hipStreamSynchronize(c->stream);
  const unsigned seq = ctx_mark(c);
  const hipError_t le = hipGetLastError();
  if (le != hipSuccess) return le;
  return ctx_wait_mark(c, seq);
}

// ---- per-kernel-class GPU timing with HIP events on the launch stream ------------------------------
struct ProfScope {
  modsx_ctx *c;
  bool on;
  size_t slot = 0;
  ProfScope(modsx_ctx *c_, int cls, double work) : c(c_), on(c_->prof.enabled) {
    if (!on) return;
    Profiler &p = c->prof;
    if (p.used == p.evA.size()) {
      hipEvent_t a, b;
      hipEventCreate(&a); hipEventCreate(&b);
      p.evA.push_back(a); p.evB.push_back(b); p.cls.push_back(0);
    }
    slot = p.used++;
    p.cls[slot] = cls;
    p.work[cls] += work;
    p.launches[cls]++;
    hipEventRecord(p.evA[slot], c->stream);
  }
  ~ProfScope() { if (on) hipEventRecord(c->prof.evB[slot], c->stream); }
};
void prof_begin(modsx_ctx *c, int cls, double work, size_t *slot) {
  *slot = (size_t)-1;
  Profiler &p = c->prof;
  if (!p.enabled) return;
  if (p.used == p.evA.size()) {
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    p.evA.push_back(a); p.evB.push_back(b); p.cls.push_back(0);
  }
  *slot = p.used++;
  p.cls[*slot] = cls;
  p.work[cls] += work;
  p.launches[cls]++;
  hipEventRecord(p.evA[*slot], c->stream);
}
void prof_end(modsx_ctx *c, size_t slot) { if (slot != (size_t)-1) hipEventRecord(c->prof.evB[slot], c->stream); }
// a slot whose two events the caller records itself (around one launch inside a launch helper); false when profiling is off
bool prof_reserve(modsx_ctx *c, int cls, double work, hipEvent_t *ev2) {
  Profiler &p = c->prof;
  if (!p.enabled) return false;
  if (p.used == p.evA.size()) {
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    p.evA.push_back(a); p.evB.push_back(b); p.cls.push_back(0);
  }
  const size_t slot = p.used++;
  p.cls[slot] = cls;
  p.work[cls] += work;
  p.launches[cls]++;
  ev2[0] = p.evA[slot]; ev2[1] = p.evB[slot];
  return true;
}
void prof_collect(modsx_ctx *c) {
  Profiler &p = c->prof;
  if (!p.enabled || !p.used) return;
  hipStreamSynchronize(c->stream);
  for (size_t i = 0; i < p.used; i++) {
    float ms = 0;
    if (hipEventElapsedTime(&ms, p.evA[i], p.evB[i]) == hipSuccess) p.ms[p.cls[i]] += ms;
  }
  p.used = 0;
}
void prof_reset(modsx_ctx *c, bool enable) {
  Profiler &p = c->prof;
  p.enabled = enable; p.used = 0;
  for (int i = 0; i < K_NCLASS; i++) { p.ms[i] = 0; p.work[i] = 0; p.launches[i] = 0; }
}

// MODSX_HOST_TIMING=2: wall time of the host phases between the launches of a set (stderr)
struct HostMark {
  bool on; double t;
  HostMark() : on(getenv("MODSX_HOST_TIMING") && atoi(getenv("MODSX_HOST_TIMING")) >= 2), t(0) { if (on) t = now(); }
  static double now() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
  void mark(const char *what) { if (!on) return; const double n = now(); fprintf(stderr, "  host %-28s %.3f ms\n", what, n - t); t = n; }
};
static double now_ms() {
  return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

// ------------------------------------------------------------------------------------------------
// context
// ------------------------------------------------------------------------------------------------
static int upload_tables(modsx_ctx *c) {
  const int PS = 41;
  std::vector<float> m(PS * PS);
  circular_gauss_mask(m.data(), PS, PS / 3.0f);  // EstimateDominantAnglesFunctor ctor, synth-detection.cpp:757-762
  {
    // the pixels that can vote (mask > 0; all lie inside the 1-pixel frame the gradient needs), raster order, as indices
    // into k_orientation's 44-column patch
    const int PSP = 44;
    std::vector<unsigned short> idx;
    std::vector<float> w;
    for (int r = 1; r < PS - 1; r++)
      for (int q = 1; q < PS - 1; q++)
        if (m[r * PS + q] > 0) { idx.push_back((unsigned short)(r * PSP + q)); w.push_back(m[r * PS + q]); }
    for (int r = 0; r < PS; r++)
      for (int q = 0; q < PS; q++)
        if ((r == 0 || q == 0 || r == PS - 1 || q == PS - 1) && m[r * PS + q] > 0) {
          set_error("orientation mask reaches the patch frame");
          return MODSX_ERR_ARG;
        }
    if ((int)idx.size() > ORI_NV) { set_error("orientation voting list exceeds ORI_NV"); return MODSX_ERR_ARG; }
    while ((int)idx.size() < ORI_NV) { idx.push_back((unsigned short)(PSP + 1)); w.push_back(0.f); }
    {   // k_orientation's lane l takes the CONTIGUOUS list elements [PER_L l, PER_L (l + 1)) and reads table entry lane + 64 q
      constexpr int PER_L = ORI_NV / 64;
      std::vector<unsigned short> idx2(ORI_NV);
      std::vector<float> w2(ORI_NV);
      for (int l = 0; l < 64; l++)
        for (int q = 0; q < PER_L; q++) { idx2[l + 64 * q] = idx[PER_L * l + q]; w2[l + 64 * q] = w[PER_L * l + q]; }
      idx.swap(idx2); w.swap(w2);
    }
    MX_HIP(hipMalloc(&c->dOriMask, ORI_NV * 4));
    MX_HIP(hipMalloc(&c->dOriIdx, ORI_NV * 2));
    MX_HIP(hipMemcpy(c->dOriMask, w.data(), ORI_NV * 4, hipMemcpyHostToDevice));
    MX_HIP(hipMemcpy(c->dOriIdx, idx.data(), ORI_NV * 2, hipMemcpyHostToDevice));
  }
  MX_HIP(hipMalloc(&c->dSiftMask, PS * PS * 4));
  circular_gauss_mask(m.data(), PS, 0);          // SIFTDescriptor ctor / DescribeRegions, siftdesc.h:87
  MX_HIP(hipMemcpy(c->dSiftMask, m.data(), PS * PS * 4, hipMemcpyHostToDevice));
  {
    std::vector<unsigned short> idx;
    for (int i = 0; i < PS * PS; i++) if (m[i] > 0) idx.push_back((unsigned short)i);
    c->nSiftMask = (int)idx.size();
    MX_HIP(hipMalloc(&c->dSiftMaskIdx, idx.size() * 2));
    MX_HIP(hipMemcpy(c->dSiftMaskIdx, idx.data(), idx.size() * 2, hipMemcpyHostToDevice));
  }
  MX_HIP(hipMalloc(&c->dAtan, 256 * 8));
  MX_HIP(hipMemcpy(c->dAtan, atan_lut_host(), 256 * 8, hipMemcpyHostToDevice));
  {
    // What the gradient stages need of atan2LUTff's angle is a function of it: the histogram bin (orientation) and the
    // fractional SIFT orientation bin (description).  The angle takes 8 x 256 values (kmath.hpp: atan2lut_case / _value) and
    // 0 in the special case (entry 2048), so both functions are tabulated here with the kernels' own expressions.
    std::vector<unsigned char> obin(ATAN_CASES);
    std::vector<float> so(ATAN_CASES);
    const double *L = atan_lut_host();
    const float PIf = float(M_PI);
    const double TWO_PI = 6.28318530718;
    for (int e = 0; e < ATAN_CASES; e++) {
      const float ori = e < 2048 ? atan2lut_value(L, e >> 8, e & 255) : 0.f;
      obin[e] = (unsigned char)(int)(36 * (ori / PIf + 1.0f) / 2.0f);          // synth-detection.cpp:781-786 as in k_orientation
      so[e] = (float)((double)8.0f * ((double)ori + TWO_PI) / TWO_PI);         // siftdesc.cpp:103-110 as in k_describe
    }
    MX_HIP(hipMalloc(&c->dOriBinTab, ATAN_CASES));
    MX_HIP(hipMemcpy(c->dOriBinTab, obin.data(), ATAN_CASES, hipMemcpyHostToDevice));
    MX_HIP(hipMalloc(&c->dSiftOTab, ATAN_CASES * 4));
    MX_HIP(hipMemcpy(c->dSiftOTab, so.data(), ATAN_CASES * 4, hipMemcpyHostToDevice));
  }
  // precomputeBinsAndWeights, matching/siftdesc.cpp:22-71 (spatialBins 4, orientationBins 8, patch 41)
  int bins[2 * PS];
  double w[2 * PS];
  const int spatialBins = 4, orientationBins = 8;
  int halfSize = PS >> 1;
  float step = float(spatialBins + 1) / (2 * halfSize);
  for (int i = 0; i < PS; i++) {
    float x = step * i;
    int xi = (int)(x);
    int b0 = xi - 1, b1 = xi;
    double w1 = x - xi;
    double w0 = 1.0f - w1;
    if (b0 < 0) { b0 = 0; w0 = 0; }
    if (b0 >= spatialBins) { b0 = spatialBins - 1; w0 = 0; }
    if (b1 < 0) { b1 = 0; w1 = 0; }
    if (b1 >= spatialBins) { b1 = spatialBins - 1; w1 = 0; }
    bins[i] = b0 * orientationBins; bins[PS + i] = b1 * orientationBins;
    w[i] = w0; w[PS + i] = w1;
    // k_describe forms (float)(w * (double)val) as an f32 product, which is the same number when w is an f32 value
    if ((double)(float)w0 != w0 || (double)(float)w1 != w1) { set_error("SIFT spatial weights are not f32 values"); return MODSX_ERR_ARG; }
  }
  MX_HIP(hipMalloc(&c->dSiftBins, sizeof bins));
  MX_HIP(hipMemcpy(c->dSiftBins, bins, sizeof bins, hipMemcpyHostToDevice));
  MX_HIP(hipMalloc(&c->dSiftW, sizeof w));
  MX_HIP(hipMemcpy(c->dSiftW, w, sizeof w, hipMemcpyHostToDevice));
  return MODSX_OK;
}

modsx_ctx *ctx_create(int device_id) {
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
    set_error("no HIP device visible: libmodsx needs an MI355X (gfx950); there is no CPU fallback");
    return nullptr;
  }
  if (device_id < 0 || device_id >= ndev) { set_error("device id out of range"); return nullptr; }
  if (hipSetDevice(device_id) != hipSuccess) { set_error("hipSetDevice failed"); return nullptr; }
  {
    // How a host thread waits for its stream.  The runtime's default spins on the completion signal: with one context per host
    // thread that is one busy core per context for as long as the stream has work -- 16 contexts burn 16 cores, which is the
    // whole CPU allowance of a container limited to 16 CPUs (the hosts this library is benchmarked on: cpu.max 1600000 100000),
    // and the verification / component-tree threads then run into the cgroup's throttle (stalls of 60-90 ms per 100 ms period).
    // MODSX_SYNC=block (default) waits on the interrupt instead; MODSX_SYNC=spin keeps the runtime's default.
    // The flag is per DEVICE (and changes how every user of that device in this process waits): it is set once for each device a
    // context is made on.  A failure is reported once and is not fatal -- the runtime then keeps spinning.
    static std::once_flag once[64];
    if (device_id < 64)
      std::call_once(once[device_id], [device_id] {
        const char *e = getenv("MODSX_SYNC");
        if (e && !strcmp(e, "spin")) return;
        const unsigned flag = (e && !strcmp(e, "yield")) ? hipDeviceScheduleYield : hipDeviceScheduleBlockingSync;
        if (hipSetDeviceFlags(flag) != hipSuccess) {
          (void)hipGetLastError();
          fprintf(stderr, "modsx: hipSetDeviceFlags(hipDeviceScheduleBlockingSync) failed on device %d; host threads will spin-wait\n", device_id);
        }
      });
  }
  modsx_ctx *c = new modsx_ctx();
  c->dev = device_id;
  if (hipStreamCreate(&c->stream) != hipSuccess) { set_error("hipStreamCreate failed"); delete c; return nullptr; }
  for (int i = 0; i < 2; i++) hipEventCreateWithFlags(&c->descEv[i], hipEventDisableTiming);
  if (hipHostMalloc((void **)&c->hFlag, 64, hipHostMallocDefault) != hipSuccess) { set_error("hipHostMalloc failed"); hipStreamDestroy(c->stream); delete c; return nullptr; }
  c->hFlag[0] = 0; c->flagSeq = 0;
  c->waitRuntime = host_wait_runtime();
  for (int i = 0; i < 6; i++) c->timings[i] = 0;
  if (upload_tables(c) != MODSX_OK) { delete c; return nullptr; }
  return c;
}

void ctx_destroy(modsx_ctx *c) {
  if (!c) return;
  ctx_worker_stop(c);
  if (c->peer) { ctx_destroy(c->peer); c->peer = nullptr; }
  if (c->half) { ctx_destroy(c->half); c->half = nullptr; }
  hipSetDevice(c->dev);
  hipStreamSynchronize(c->stream);
  for (int i = 0; i < MAXB; i++) c->pyr[i].store.release();
  DevBuf *bufs[] = {&c->nmsJobs, &c->cand, &c->counter, &c->affJobs, &c->affOut, &c->oriJobs, &c->oriOut, &c->descJobs, &c->tilePrefix,
                    &c->taps, &c->imgRefs, &c->scratchA, &c->scratchB, &c->descAllF[0], &c->descAllF[1], &c->descAllU8[0],
                    &c->descAllU8[1], &c->descAllU8b[0], &c->descAllU8b[1], &c->shardLocal, &c->pos2, &c->matchRows, &c->matchWork, &c->misc, &c->scratchC, &c->needTab, &c->coordTab, &c->tileJob, &c->blurTiles, &c->nmsQueue, &c->rowStarts, &c->viewTmp[0], &c->viewTmp[1], &c->viewTaps, &c->viewJobs};
  for (DevBuf *b : bufs) b->release();
  for (int i = 0; i < MAXB; i++) { c->descF[i].release(); c->descU8[i].release(); c->viewImg[i].release(); for (int k = 0; k < 3; k++) c->descU8x[k][i].release(); }
  for (int d = 0; d < 2; d++) for (int t = 0; t < 4; t++) for (int sd = 0; sd < 2; sd++) c->descCls[d][t][sd].release();
  for (int t = 0; t < 4; t++) c->halfDesc[t].release();
  c->candSort.release(); c->candOut.release();
  PinBuf *pins[] = {&c->hCand, &c->hAff, &c->hOri, &c->hDesc, &c->hMisc, &c->hNms, &c->hMatch, &c->hViewTaps, &c->hViewJobs, &c->hMser};
  for (PinBuf *b : pins) b->release();
  hipFree(c->dSmmMask); hipFree(c->dOriMask); hipFree(c->dOriIdx); hipFree(c->dSiftMask); hipFree(c->dSiftMaskIdx); hipFree(c->dAtan); hipFree(c->dOriBinTab); hipFree(c->dSiftOTab); hipFree(c->dSiftBins);
  hipFree(c->dSiftW);
  for (int i = 0; i < 2; i++) hipEventDestroy(c->descEv[i]);
  c->hDescB.release(); c->hRefs.release();
  hipStreamDestroy(c->stream);
  if (c->hFlag) hipHostFree(c->hFlag);
  delete c;
}

// ------------------------------------------------------------------------------------------------
// pyramid
// ------------------------------------------------------------------------------------------------
static int cv_round(double v) { return (int)lrint(v); }  // cvRound: round half to even

struct SigmaPlan {
  int levels;
  float sigmaStep;
  float curSigma[8];   // sigma of level i (pyramid.cpp:458-459, 532)
  float incSigma[8];   // blur applied to level i-1 to get level i (:483)
};

static SigmaPlan make_sigma_plan(const modsx_hessaff_params &p) {
  SigmaPlan s;
  s.levels = p.numberOfScales + 2;
  s.sigmaStep = powf(2.0f, 1.0f / (float)p.numberOfScales);
  float cur = p.initialSigma;
  s.curSigma[0] = cur; s.incSigma[0] = 0;
  for (int i = 1; i < s.levels; i++) {
    s.incSigma[i] = cur * sqrtf(s.sigmaStep * s.sigmaStep - 1.0f);
    cur *= s.sigmaStep;
    s.curSigma[i] = cur;
  }
  return s;
}

static int fill_taps(BlurBatch &b, float sigma) {
  int n = blur_ksize(sigma);
  if (n > MAX_TAPS) { set_error("pyramid blur kernel larger than 17 taps is not supported"); return MODSX_ERR_ARG; }
  std::vector<float> k = gaussian_kernel(n, sigma);
  b.n = n;
  for (int i = 0; i < n; i++) b.k[i] = k[i];
  return MODSX_OK;
}

// ScaleSpaceDetector::detectPyramidKeypoints / detectOctaveKeypoints (pyramid.cpp:455-573) for a batch
// of images: builds every blur and response level in HBM.  firstLevelGiven: the image IS the first level
// of a single octave (stage tap used by modsx_octave_levels).
int build_pyramids(modsx_ctx *c, const modsx_image *const *imgs, int n, const modsx_hessaff_params &p,
                   bool singleOctaveFromFirstLevel) {
  if (n <= 0 || n > MAXB) { set_error("batch size"); return MODSX_ERR_ARG; }
  if (p.numberOfScales < 1 || p.numberOfScales > 6) { set_error("numberOfScales"); return MODSX_ERR_ARG; }
  if (p.detectorType != MODSX_DET_HESSIAN && p.detectorType != MODSX_DET_DOG && p.detectorType != MODSX_DET_HARRIS) {
    set_error("detectorType must be Hessian (0), DoG (1) or Harris (2)");
    return MODSX_ERR_ARG;
  }
  const bool hess = p.detectorType == MODSX_DET_HESSIAN;    // its response is fused into the blur / resize kernels; the others follow below
  const SigmaPlan sp = make_sigma_plan(p);
  const int L = sp.levels;
  const int minSize = 2 * p.border + 2;
  int maxOct = 0;
  for (int i = 0; i < n; i++) {
    Pyramid &py = c->pyr[i];
    py.nOct = 0;
    int rows = imgs[i]->rows, cols = imgs[i]->cols;
    float pd = 1.0f;
    size_t total = 0;
    while (rows > minSize && cols > minSize && py.nOct < 24) {
      Octave &o = py.oct[py.nOct++];
      o.rows = rows; o.cols = cols; o.pixelDistance = pd;
      total += (size_t)2 * L * rows * cols;
      pd *= 2.0;
      rows = cv_round(rows * 0.5); cols = cv_round(cols * 0.5);
      if (singleOctaveFromFirstLevel) break;
    }
    if (!py.store.ensure(total * sizeof(float) + 64)) return MODSX_ERR_NOMEM;
    float *ptr = (float *)py.store.p;
    for (int o = 0; o < py.nOct; o++) {
      size_t npx = (size_t)py.oct[o].rows * py.oct[o].cols;
      for (int l = 0; l < L; l++) { py.oct[o].blur[l] = ptr; ptr += npx; }
      for (int l = 0; l < L; l++) { py.oct[o].resp[l] = ptr; ptr += npx; }
    }
    maxOct = std::max(maxOct, py.nOct);
  }
  hipStream_t s = c->stream;
  for (int o = 0; o < maxOct; o++) {
    // first level of the octave
    BlurBatch bb;
    memset(&bb, 0, sizeof bb);
    int nj = 0, mr = 0, mc = 0;
    if (o == 0) {
      const float curSigma0 = 0.5f;
      const bool preBlur = !singleOctaveFromFirstLevel && p.initialSigma > curSigma0;
      if (preBlur) {
        float sigma = sqrtf(p.initialSigma * p.initialSigma - curSigma0 * curSigma0);
        int rc = fill_taps(bb, sigma);
        if (rc) return rc;
      }
      for (int i = 0; i < n; i++) {
        if (c->pyr[i].nOct <= 0) continue;
        Octave &oc = c->pyr[i].oct[0];
        BlurJob &j = bb.j[nj++];
        j.src = imgs[i]->d; j.blur = oc.blur[0]; j.resp = hess ? oc.resp[0] : nullptr; j.rows = oc.rows; j.cols = oc.cols;
        j.norm = sp.curSigma[0] * sp.curSigma[0];
        mr = std::max(mr, oc.rows); mc = std::max(mc, oc.cols);
        if (!preBlur) {
          MX_HIP(hipMemcpyAsync(oc.blur[0], imgs[i]->d, (size_t)oc.rows * oc.cols * 4, hipMemcpyDeviceToDevice, s));
          j.src = oc.blur[0];
        }
      }
      double px = 0;
      for (int q = 0; q < nj; q++) px += (double)bb.j[q].rows * bb.j[q].cols;
      if (nj) {
        if (preBlur) { ProfScope ps(c, K_BLUR_HESS, px * 12); launch_blur_hess(s, bb, nj, mr, mc); }
        else if (hess) { ProfScope ps(c, K_HESSIAN, px * 8); launch_hessian(s, bb, nj, mr, mc); }
      }
    } else {
      ResizeBatch rb;
      memset(&rb, 0, sizeof rb);
      for (int i = 0; i < n; i++) {
        if (c->pyr[i].nOct <= o) continue;
        Octave &pv = c->pyr[i].oct[o - 1], &oc = c->pyr[i].oct[o];
        ResizeJob &r = rb.j[nj];
        r.src = pv.blur[p.numberOfScales]; r.dst = oc.blur[0];
        r.srows = pv.rows; r.scols = pv.cols; r.drows = oc.rows; r.dcols = oc.cols;
        r.resp = hess ? oc.resp[0] : nullptr; r.norm = sp.curSigma[0] * sp.curSigma[0];   // resize + Hessian of the new level in one launch
        BlurJob &j = bb.j[nj++];
        j.src = oc.blur[0]; j.blur = nullptr; j.resp = oc.resp[0]; j.rows = oc.rows; j.cols = oc.cols;
        j.norm = sp.curSigma[0] * sp.curSigma[0];
        mr = std::max(mr, oc.rows); mc = std::max(mc, oc.cols);
      }
      if (nj) {
        double spx = 0, dpx = 0;
        for (int q = 0; q < nj; q++) { spx += (double)rb.j[q].srows * rb.j[q].scols; dpx += (double)rb.j[q].drows * rb.j[q].dcols; }
        { ProfScope ps(c, K_RESIZE, (spx + dpx) * 4 + dpx * 4); launch_resize_half(s, rb, nj, mr, mc); }
      }
    }
    if (!nj) continue;
    for (int l = 1; l < L; l++) {
      BlurBatch b2;
      memset(&b2, 0, sizeof b2);
      int rc = fill_taps(b2, sp.incSigma[l]);
      if (rc) return rc;
      int k = 0;
      for (int i = 0; i < n; i++) {
        if (c->pyr[i].nOct <= o) continue;
        Octave &oc = c->pyr[i].oct[o];
        BlurJob &j = b2.j[k++];
        j.src = oc.blur[l - 1]; j.blur = oc.blur[l]; j.resp = hess ? oc.resp[l] : nullptr; j.rows = oc.rows; j.cols = oc.cols;
        j.norm = sp.curSigma[l] * sp.curSigma[l];
      }
      double px = 0;
      for (int q = 0; q < k; q++) px += (double)b2.j[q].rows * b2.j[q].cols;
      ProfScope ps(c, K_BLUR_HESS, px * 12);
      launch_blur_hess(s, b2, k, mr, mc);
    }
    if (!hess) {
      // DoG / Harris (ScaleSpaceDetector::dogResponse :176-181, HarrisResponse :283-305; norm = sigma^2 of the level, :475,490):
      // the response of every level from its blur, with the generic any-sigma filter passes (the DoG of a level is the level minus
      // its blur with sigma = norm; Harris blurs three gradient products with sqrt(0.6 norm)) -- separate launches per image and
      // level: these detectors are on no shipped configuration's path, the Hessian's fused kernels are untouched
      for (int l = 0; l < L; l++) {
        const float norm = sp.curSigma[l] * sp.curSigma[l];
        const float sigma = p.detectorType == MODSX_DET_DOG ? norm : sqrtf((float)(0.6 * norm));
        const int nt = blur_ksize(sigma);
        if (2 * nt > 2 * 4096) { set_error("response blur kernel too large"); return MODSX_ERR_ARG; }
        if (!c->viewTaps.ensure((size_t)L * 2 * 4100 * 4) || !c->hViewTaps.ensure((size_t)L * 2 * 4100 * 4)) return MODSX_ERR_NOMEM;
        // one slice of the staging buffers per level, so that no upload overwrites taps a queued launch still reads
        float *dT = (float *)c->viewTaps.p + (size_t)l * 2 * 4100, *hT = (float *)c->hViewTaps.p + (size_t)l * 2 * 4100;
        bool tapsUp = false;
        for (int i = 0; i < n; i++) {
          if (c->pyr[i].nOct <= o) continue;
          Octave &oc = c->pyr[i].oct[o];
          const int rows = oc.rows, cols = oc.cols;
          const size_t npx = (size_t)rows * cols;
          if (!c->scratchA.ensure(npx * 4 * 8)) return MODSX_ERR_NOMEM;
          float *buf = (float *)c->scratchA.p, *tmp = buf, *a = buf + npx, *b = buf + 2 * npx, *cc = buf + 3 * npx;
          float *ba = buf + 4 * npx, *bb2 = buf + 5 * npx, *bc = buf + 6 * npx;
          const int nx = cols == 1 ? 1 : nt, ny = rows == 1 ? 1 : nt;
          if (!tapsUp || nx != nt || ny != nt) {
            std::vector<float> kx = gaussian_kernel(nx, sigma), ky = gaussian_kernel(ny, sigma);
            if (tapsUp) MX_HIP(ctx_sync(c));      // a degenerate (one-row / one-column) level re-uses the slice
            memcpy(hT, kx.data(), nx * 4); memcpy(hT + nx, ky.data(), ny * 4);
            MX_HIP(ctx_copy(c, dT, hT, (size_t)(nx + ny) * 4, hipMemcpyHostToDevice));
            tapsUp = nx == nt && ny == nt;
          }
          auto blur = [&](const float *src, float *dst) {
            if (nt == 1) { MX_HIP(hipMemcpyAsync(dst, src, npx * 4, hipMemcpyDeviceToDevice, s)); return MODSX_OK; }
            launch_blur_pass(s, src, tmp, rows, cols, dT, nx, 0, 1);
            launch_blur_pass(s, tmp, dst, rows, cols, dT + nx, ny, 1, 1);
            return MODSX_OK;
          };
          if (p.detectorType == MODSX_DET_DOG) {
            int rc = blur(oc.blur[l], a);
            if (rc) return rc;
            launch_sub(s, oc.blur[l], a, oc.resp[l], npx);
          } else {
            launch_grad_products(s, oc.blur[l], rows, cols, a, b, cc);
            int rc = blur(a, ba);
            if (!rc) rc = blur(b, bb2);
            if (!rc) rc = blur(cc, bc);
            if (rc) return rc;
            launch_harris_combine(s, ba, bb2, bc, (float)(0.6 * norm), oc.resp[l], npx);
          }
        }
      }
    }
  }
  MX_HIP(hipGetLastError());
  return MODSX_OK;
}

// ------------------------------------------------------------------------------------------------
// detection: extrema + localisation on device, detection order + octaveMap + scale on host
// ------------------------------------------------------------------------------------------------
static const unsigned CAND_CAP = 1u << 21;

int detect_scalespace_batch(modsx_ctx *c, const modsx_image *const *imgs, int n, const modsx_hessaff_params &p,
                            std::vector<modsx_sskp> *out) {
  int rc = build_pyramids(c, imgs, n, p, false);
  if (rc) return rc;
  hipStream_t s = c->stream;
  if (!c->cand.ensure((size_t)CAND_CAP * sizeof(Candidate)) || !c->nmsQueue.ensure((size_t)CAND_CAP * 16)) return MODSX_ERR_NOMEM;
  if (!c->counter.ensure((NMS_QUEUES + 1) * 128)) return MODSX_ERR_NOMEM;
  // [0] accepted candidates, [1] extremum-queue overflow flag, from word 32 on the sub-queue counters: one fill for the lot
  MX_HIP(hipMemsetAsync(c->counter.p, 0, 128 + NMS_QUEUES * 128, s));
  bool queuesClean = true;   // the sub-queue counters are zero (no scan has run since the fill)
  // thresholds, affinedetectors/pyramid.h:47-67 (DET_HESSIAN)
  NmsBatch nb;
  memset(&nb, 0, sizeof nb);
  nb.edgeScoreThreshold = (p.edgeEigenValueRatio + 1.0f) * (p.edgeEigenValueRatio + 1.0f) / p.edgeEigenValueRatio;
  float finalTh = p.threshold;
  float posTh = (float)(0.8 * finalTh);
  float negTh = -posTh;
  if (p.detectorType == MODSX_DET_HESSIAN) finalTh = p.threshold * p.threshold;     // pyramid.h:56-57: squared for DET_HESSIAN only
  if (p.mode != MODSX_FIXED_TH) finalTh = posTh = negTh = 0.0f;
  nb.posTh = posTh; nb.negTh = negTh; nb.finalTh = finalTh; nb.border = p.border; nb.detType = p.detectorType;
  int maxOct = 0;
  for (int i = 0; i < n; i++) maxOct = std::max(maxOct, c->pyr[i].nOct);
  // all (image, octave, level) scans of the batch in one launch (NMS_MAXJ jobs at most per launch).  With the shipped
  // numberOfScales = 3 the tiles number OCTAVES and a tile scans the three levels of its octave in one pass over the five
  // response planes (k_nms_localize_oct); otherwise a tile belongs to one level
  const bool perOctave = p.numberOfScales == 3 && !getenv("MODSX_NMS_PER_LEVEL");
  std::vector<NmsJob> hjobs;
  std::vector<int> hpfx(1, 0), hfirst;
  double px = 0;
  auto flush = [&](bool last) -> int {
    const int nj = (int)hjobs.size();
    if (!nj) return MODSX_OK;
    const int np = (int)hpfx.size() - 1;     // tile groups: octaves or levels
    const size_t jobBytes = (size_t)nj * sizeof(NmsJob), pfxBytes = (size_t)(np + 1) * 4, firstBytes = (size_t)std::max<size_t>(1, hfirst.size()) * 4;
    if (!c->nmsJobs.ensure(jobBytes + pfxBytes + firstBytes + 64)) return MODSX_ERR_NOMEM;
    if (!c->hNms.ensure(jobBytes + pfxBytes + firstBytes)) return MODSX_ERR_NOMEM;   // jobs + prefix (+ first level job of every octave): one pinned blob, one copy
    memcpy(c->hNms.p, hjobs.data(), jobBytes);
    memcpy((char *)c->hNms.p + jobBytes, hpfx.data(), pfxBytes);
    if (!hfirst.empty()) memcpy((char *)c->hNms.p + jobBytes + pfxBytes, hfirst.data(), hfirst.size() * 4);
    MX_HIP(ctx_copy(c, c->nmsJobs.p, c->hNms.p, jobBytes + pfxBytes + firstBytes, hipMemcpyHostToDevice));
    if (!c->tileJob.ensure((size_t)hpfx.back() * 4 + 4)) return MODSX_ERR_NOMEM;
    const int *dPfx = (const int *)((char *)c->nmsJobs.p + jobBytes);
    launch_expand_tiles(s, dPfx, np, (int *)c->tileJob.p);
    {
      ProfScope ps(c, K_NMS, px * 12);
      if (!queuesClean) MX_HIP(hipMemsetAsync((unsigned *)c->counter.p + 32, 0, NMS_QUEUES * 128, s));
      queuesClean = false;
      launch_nms(s, nb, (const NmsJob *)c->nmsJobs.p, dPfx, (const int *)c->tileJob.p, nj,
                 hpfx.back(), (int4 *)c->nmsQueue.p, (unsigned *)c->counter.p + 32, CAND_CAP, (Candidate *)c->cand.p,
                 (unsigned *)c->counter.p, CAND_CAP, perOctave ? (const int *)((char *)c->nmsJobs.p + jobBytes + pfxBytes) : nullptr,
                 p.numberOfScales);
    }
    // the host tables are reused by the next flush; after the last one the counter read-back below waits for the launch
    if (!last) MX_HIP(ctx_sync(c));
    hjobs.clear(); hpfx.assign(1, 0); hfirst.clear(); px = 0;
    return MODSX_OK;
  };
  for (int o = 0; o < maxOct; o++)
    for (int i = 0; i < n; i++) {
      if (c->pyr[i].nOct <= o) continue;
      Octave &oc = c->pyr[i].oct[o];
      const int w = oc.cols - 2 * p.border, h = oc.rows - 2 * p.border;
      if (w <= 0 || h <= 0) continue;
      if ((int)hjobs.size() + p.numberOfScales > NMS_MAXJ) { int rcf = flush(false); if (rcf) return rcf; }
      const int tiles = ((w + 63) / 64) * ((h + NMS_TILE_ROWS - 1) / NMS_TILE_ROWS);
      if (perOctave) { hfirst.push_back((int)hjobs.size()); hpfx.push_back(hpfx.back() + tiles); }
      for (int l = 1; l <= p.numberOfScales; l++) {
        NmsJob j;
        j.low = oc.resp[l - 1]; j.cur = oc.resp[l]; j.high = oc.resp[l + 1]; j.blur = oc.blur[l];
        j.rows = oc.rows; j.cols = oc.cols; j.img = i; j.octave = o; j.level = l; j.pad = 0;
        hjobs.push_back(j);
        if (!perOctave) hpfx.push_back(hpfx.back() + tiles);
        px += (double)oc.rows * oc.cols;
      }
    }
  { int rcf = flush(true); if (rcf) return rcf; }
  // MODSX_DEVICE_ORDER=1: detection order and the octaveMap claim on the device (kernels_cand.hip) -- the host receives the
  // SURVIVING candidates in the reference's visiting order and only forms the scale (glibc powf) and the keypoint records.
  // Built and bit-exact (tests/test_gpu_parity.py), but NOT the default: measured in round 4 it costs the 31-view bench 9 %
  // (167 against 183.5 pairs/s) and a lone pair 0.5 ms (13.05 against 12.55 ms) -- key build, radix sort, two table fills, claim
  // and compaction are six more small launches on every launch set's stream, while the host's radix sort + hash claim run on
  // host cores that are otherwise idle and overlap the other contexts' device work.
  static const bool deviceOrder = getenv("MODSX_DEVICE_ORDER") != nullptr && atoi(getenv("MODSX_DEVICE_ORDER")) != 0;
  if (deviceOrder) {
    // the opt-in path lives in libmodsx_cand.so beside this library (kernels_cand.hip), loaded on first use
    typedef size_t (*temp_fn)(unsigned);
    typedef int (*order_fn)(hipStream_t, const Candidate *, const unsigned *, unsigned, unsigned long long *, unsigned long long *, unsigned *, unsigned *,
                            void *, size_t, unsigned long long *, unsigned *, unsigned, unsigned *, Candidate *, unsigned *);
    static temp_fn cand_sort_temp_bytes = nullptr;
    static order_fn launch_cand_order = nullptr;
    static std::once_flag candOnce;
    std::call_once(candOnce, [] {
      Dl_info di;
      std::string path = "libmodsx_cand.so";
      if (dladdr((const void *)&modsx_create, &di) && di.dli_fname) {
        const std::string self(di.dli_fname);
        const size_t sl = self.rfind('/');
        if (sl != std::string::npos) path = self.substr(0, sl + 1) + "libmodsx_cand.so";
      }
      if (void *h = dlopen(path.c_str(), RTLD_NOW | RTLD_LOCAL)) {
        cand_sort_temp_bytes = (temp_fn)dlsym(h, "modsx_cand_sort_temp_bytes");
        launch_cand_order = (order_fn)dlsym(h, "modsx_cand_order");
      }
    });
    if (!cand_sort_temp_bytes || !launch_cand_order) {
      set_error("MODSX_DEVICE_ORDER=1, but libmodsx_cand.so (make -C mods_amd/csrc cand) is not beside libmodsx.so");
      return MODSX_ERR_DEVICE;
    }
    if (!c->hMisc.ensure(64)) return MODSX_ERR_NOMEM;
    for (int attempt = 0;; attempt++) {
      // the sort runs over a host-chosen capacity (the device-side count is not known here): what the context's last set had
      // (+ 1/4); a set that holds more is ordered again with its real count -- one more wait, as for the old download
      const unsigned nsort = (unsigned)std::min<size_t>(CAND_CAP, attempt ? c->lastCandCount + 64 : c->lastCandCount + c->lastCandCount / 4 + 1024);
      unsigned tabSize = 1024;
      while (tabSize < 2 * nsort + 16) tabSize <<= 1;
      auto up = [](size_t b) { return (b + 255) & ~(size_t)255; };
      const size_t tempB = cand_sort_temp_bytes(nsort);
      const size_t oKeys = 0, oKeys2 = oKeys + up((size_t)nsort * 8), oIdx = oKeys2 + up((size_t)nsort * 8), oIdx2 = oIdx + up((size_t)nsort * 4),
                   oSlot = oIdx2 + up((size_t)nsort * 4), oTabK = oSlot + up((size_t)nsort * 4), oTabR = oTabK + up((size_t)tabSize * 8),
                   oTemp = oTabR + up((size_t)tabSize * 4), total = oTemp + up(tempB) + 256;
      if (!c->candSort.ensure(total) || !c->candOut.ensure((size_t)nsort * sizeof(Candidate) + 64)) return MODSX_ERR_NOMEM;
      char *w = (char *)c->candSort.p;
      unsigned *survivors = (unsigned *)c->counter.p + 2;     // word 2 of the counter block (zeroed with it)
      if (launch_cand_order(s, (const Candidate *)c->cand.p, (const unsigned *)c->counter.p, nsort, (unsigned long long *)(w + oKeys),
                            (unsigned long long *)(w + oKeys2), (unsigned *)(w + oIdx), (unsigned *)(w + oIdx2), w + oTemp, tempB,
                            (unsigned long long *)(w + oTabK), (unsigned *)(w + oTabR), tabSize, (unsigned *)(w + oSlot),
                            (Candidate *)c->candOut.p, survivors)) { set_error("device-side detection order failed"); return MODSX_ERR_DEVICE; }
      const size_t spec = std::min<size_t>(nsort, c->lastSurvivors + c->lastSurvivors / 4 + 1024);
      if (!c->hCand.ensure(std::max<size_t>(spec, 1) * sizeof(Candidate))) return MODSX_ERR_NOMEM;
      MX_HIP(ctx_copy(c, c->hMisc.p, c->counter.p, 12, hipMemcpyDeviceToHost));
      MX_HIP(ctx_copy(c, c->hCand.p, c->candOut.p, spec * sizeof(Candidate), hipMemcpyDeviceToHost));
      MX_HIP(ctx_sync(c));
      const unsigned cnt = ((unsigned *)c->hMisc.p)[0], nsurv = ((unsigned *)c->hMisc.p)[2];
      if (cnt > CAND_CAP || ((unsigned *)c->hMisc.p)[1]) { set_error("candidate buffer overflow"); return MODSX_ERR_NOMEM; }
      c->lastCandCount = cnt;
      if (cnt > nsort) {                       // the capacity was a guess and too small: once more with the count
        if (attempt) { set_error("device-side detection order: capacity does not converge"); return MODSX_ERR_INTERNAL; }
        continue;
      }
      c->lastSurvivors = nsurv;
      if (nsurv > spec) {
        if (!c->hCand.ensure((size_t)nsurv * sizeof(Candidate))) return MODSX_ERR_NOMEM;   // (re-allocation loses the first part: copy all)
        MX_HIP(ctx_copy(c, c->hCand.p, c->candOut.p, (size_t)nsurv * sizeof(Candidate), hipMemcpyDeviceToHost));
        MX_HIP(ctx_sync(c));
      }
      HostMark hm;
      const Candidate *cd = (const Candidate *)c->hCand.p;
      const SigmaPlan sp = make_sigma_plan(p);
      // image-major, then (octave, level, row, column): one linear pass
      std::vector<uint32_t> imgStart(n + 1, 0);
      for (unsigned k = 0; k < nsurv; k++) {
        if ((unsigned)cd[k].img >= (unsigned)n || (unsigned)cd[k].octave >= 32u) { set_error("candidate outside the image / octave range"); return MODSX_ERR_DEVICE; }
        imgStart[cd[k].img + 1]++;
      }
      for (int i = 0; i < n; i++) imgStart[i + 1] += imgStart[i];
      host_parallel_light(n, [&](int img) {
        std::vector<modsx_sskp> &dst = out[img];
        dst.clear();
        dst.reserve(imgStart[img + 1] - imgStart[img]);
        for (uint32_t k = imgStart[img]; k < imgStart[img + 1]; k++) {
          const Candidate &q = cd[k];
          const float pixelDistance = c->pyr[q.img].oct[q.octave].pixelDistance;
          const float curScale = sp.curSigma[q.level];
          float scale = curScale * powf(2.0f, q.b2 / p.numberOfScales);
          modsx_sskp kp;
          kp.octave = q.octave; kp.level = q.level; kp.r0 = q.r0; kp.c0 = q.c0; kp.r = q.r; kp.c = q.c; kp.type = q.type;
          kp.pad = 0;
          kp.b0 = q.b0; kp.b1 = q.b1; kp.b2 = q.b2; kp.val = q.val;
          kp.x = pixelDistance * (q.c + q.b0);
          kp.y = pixelDistance * (q.r + q.b1);
          kp.s = pixelDistance * scale;
          kp.pixelDistance = pixelDistance;
          dst.push_back(kp);
        }
      });
      hm.mark("sskp from ordered survivors");
      return MODSX_OK;
    }
  }
  // the count and the candidates come down behind ONE wait: the records are copied speculatively, as many as the context's
  // last set had (+ 1/4); a set that holds more costs a second copy for the rest
  if (!c->hMisc.ensure(64)) return MODSX_ERR_NOMEM;
  const size_t spec = std::min<size_t>(CAND_CAP, c->lastCandCount + c->lastCandCount / 4 + 1024);
  if (!c->hCand.ensure(spec * sizeof(Candidate))) return MODSX_ERR_NOMEM;
  MX_HIP(ctx_copy(c, c->hMisc.p, c->counter.p, 8, hipMemcpyDeviceToHost));
  MX_HIP(ctx_copy(c, c->hCand.p, c->cand.p, spec * sizeof(Candidate), hipMemcpyDeviceToHost));
  MX_HIP(ctx_sync(c));
  unsigned cnt = *(unsigned *)c->hMisc.p;
  if (cnt > CAND_CAP || ((unsigned *)c->hMisc.p)[1]) { set_error("candidate buffer overflow"); return MODSX_ERR_NOMEM; }
  c->lastCandCount = cnt;
  if (cnt > spec) {
    if (!c->hCand.ensure((size_t)cnt * sizeof(Candidate))) return MODSX_ERR_NOMEM;   // (re-allocation loses the first part: copy all)
    MX_HIP(ctx_copy(c, c->hCand.p, c->cand.p, (size_t)cnt * sizeof(Candidate), hipMemcpyDeviceToHost));
    MX_HIP(ctx_sync(c));
  }
  HostMark hm;
  const Candidate *cd = (const Candidate *)c->hCand.p;
  // The reference visits (octave, level, row, col) in this order per image (pyramid.cpp:438-451, 490-498, 564-571) and the
  // first candidate in that order that lands on a pixel of an octave claims it (octaveMap, :414-418).  Images are
  // independent: the candidates are bucketed by image once, then every image sorts one 64-bit key per candidate, applies the
  // claim through a small open-addressing table and builds its keypoints -- one task per image on the host pool.
  for (unsigned k = 0; k < cnt; k++) {
    const Candidate &q = cd[k];
    if ((unsigned)q.img >= (unsigned)n || (unsigned)q.octave >= 32u || (unsigned)q.level >= 32u || (unsigned)q.r0 >= (1u << 14) ||
        (unsigned)q.c0 >= (1u << 14) || (unsigned)q.r >= (1u << 24) || (unsigned)q.c >= (1u << 24)) {
      set_error("candidate outside the sort key's range");
      return MODSX_ERR_DEVICE;
    }
  }
  std::vector<uint32_t> &byImg = c->candOrder;
  std::vector<uint32_t> imgStart(n + 1, 0);
  byImg.resize(cnt);
  for (unsigned k = 0; k < cnt; k++) imgStart[cd[k].img + 1]++;
  for (int i = 0; i < n; i++) imgStart[i + 1] += imgStart[i];
  {
    std::vector<uint32_t> fill(imgStart.begin(), imgStart.end() - 1);
    for (unsigned k = 0; k < cnt; k++) byImg[fill[cd[k].img]++] = k;
  }
  hm.mark("cand bucket by image");
  const SigmaPlan sp = make_sigma_plan(p);
  host_parallel_light(n, [&](int img) {
    std::vector<modsx_sskp> &dst = out[img];
    dst.clear();
    const uint32_t *idx = byImg.data() + imgStart[img];
    const size_t m = imgStart[img + 1] - imgStart[img];
    if (!m) return;
    // keys are unique (one candidate per (octave, level, pixel)), so the order is the same whatever sorts them: an LSD radix
    // sort of (key, index), 11 bits per pass, passes whose digit is the same for every key left out
    // key and index share ONE 64-bit word (octave 5 | level 5 | row 14 | column 14 | index 21 bits: images are at most 16384 px per side,
    // a set holds at most 2^21 candidates): half the bytes per pass of the (key, index) pairs sorted until round 6
    static thread_local std::vector<uint64_t> order, order2;
    static_assert(CAND_CAP <= (1u << 21), "index field of the packed sort key");
    order.resize(m); order2.resize(m);
    for (size_t k = 0; k < m; k++) {
      const Candidate &q = cd[idx[k]];
      order[k] = ((uint64_t)q.octave << 54) | ((uint64_t)q.level << 49) | ((uint64_t)(q.r0 & 0x3fff) << 35) | ((uint64_t)(q.c0 & 0x3fff) << 21) | idx[k];
    }
    {
      constexpr int BITS = 11, NB = 1 << BITS;
      uint32_t hist[NB];
      for (int shift = 21; shift < 59; shift += BITS) {
        memset(hist, 0, sizeof hist);
        for (size_t k = 0; k < m; k++) hist[(order[k] >> shift) & (NB - 1)]++;
        if (hist[(order[0] >> shift) & (NB - 1)] == m) continue;
        uint32_t sum = 0;
        for (int b = 0; b < NB; b++) { const uint32_t h = hist[b]; hist[b] = sum; sum += h; }
        for (size_t k = 0; k < m; k++) order2[hist[(order[k] >> shift) & (NB - 1)]++] = order[k];
        order.swap(order2);
      }
    }
    size_t tabSize = 64;
    while (tabSize < m * 2 + 16) tabSize <<= 1;
    std::vector<uint64_t> claimed(tabSize, 0);  // key + 1, 0 = empty
    dst.reserve(m);
    for (size_t kk = 0; kk < m; kk++) {
      const Candidate &q = cd[order[kk] & 0x1fffffu];
      const uint64_t key = (((uint64_t)q.octave << 48) | ((uint64_t)q.r << 24) | (uint64_t)q.c) + 1;
      size_t h = (size_t)((key * 0x9E3779B97F4A7C15ull) >> 20) & (tabSize - 1);
      bool taken = false;
      while (claimed[h]) { if (claimed[h] == key) { taken = true; break; } h = (h + 1) & (tabSize - 1); }
      if (taken) continue;
      claimed[h] = key;
      const float pixelDistance = c->pyr[q.img].oct[q.octave].pixelDistance;
      const float curScale = sp.curSigma[q.level];
      float scale = curScale * powf(2.0f, q.b2 / p.numberOfScales);
      modsx_sskp kp;
      kp.octave = q.octave; kp.level = q.level; kp.r0 = q.r0; kp.c0 = q.c0; kp.r = q.r; kp.c = q.c; kp.type = q.type;
      kp.pad = 0;
      kp.b0 = q.b0; kp.b1 = q.b1; kp.b2 = q.b2; kp.val = q.val;
      kp.x = pixelDistance * (q.c + q.b0);
      kp.y = pixelDistance * (q.r + q.b1);
      kp.s = pixelDistance * scale;
      kp.pixelDistance = pixelDistance;
      dst.push_back(kp);
    }
  });
  hm.mark("octaveMap claim + sskp");
  return MODSX_OK;
}

static int ensure_smm_mask(modsx_ctx *c, int W) {
  if (c->smmW == W && c->dSmmMask) return MODSX_OK;
  if (W < 3 || W > 19 || !(W & 1)) { set_error("smmWindowSize must be odd and <= 19"); return MODSX_ERR_ARG; }
  if (c->dSmmMask) hipFree(c->dSmmMask);
  std::vector<float> m(W * W);
  gauss_mask(m.data(), W);
  MX_HIP(hipMalloc(&c->dSmmMask, W * W * 4));
  MX_HIP(hipMemcpy(c->dSmmMask, m.data(), W * W * 4, hipMemcpyHostToDevice));
  c->smmW = W;
  return MODSX_OK;
}

// AffineDetector::prepareKeysForExport, scale-space-detector.hpp:118-198
static void prepare_keys_for_export(std::vector<modsx_keypoint> &keys, const modsx_hessaff_params &p) {
  if (keys.empty() || p.mode == MODSX_FIXED_TH) return;
  auto cmpv = [](modsx_keypoint k1, modsx_keypoint k2) { return fabs(k1.response) > fabs(k2.response); };
  std::sort(keys.begin(), keys.end(), cmpv);
  double maxResponse = fabs(keys[0].response);
  int regNumber = (int)keys.size();
  auto cmp = [](const modsx_keypoint &k1, const modsx_keypoint &k2) { return fabs(k1.response) > fabs(k2.response); };
  switch (p.mode) {
    case MODSX_RELATIVE_TH: {
      modsx_keypoint t = keys[0];
      t.response = (float)(maxResponse * p.rel_threshold);
      keys.resize(std::lower_bound(keys.begin(), keys.end(), t, cmp) - keys.begin());
      break;
    }
    case MODSX_FIXED_REG_NUMBER: {
      int nn = p.reg_number;
      if (p.doBaumberg) nn = (int)floor(3.0 * (double)nn);
      if ((nn < regNumber) && (nn >= 0)) keys.resize(nn);
      break;
    }
    case MODSX_RELATIVE_REG_NUMBER: {
      keys.resize((int)floor(p.rel_reg_number * (double)keys.size()));
      break;
    }
    case MODSX_NOT_LESS_THAN_REGIONS: {
      modsx_keypoint t = keys[0];
      t.response = p.threshold;
      int fix = (int)(std::lower_bound(keys.begin(), keys.end(), t, cmp) - keys.begin());
      if (fix < p.reg_number) keys.resize(std::min(p.reg_number, regNumber));
      else keys.resize(std::min(fix, regNumber));
      break;
    }
    default: break;
  }
  if (p.mode == MODSX_FIXED_REG_NUMBER && (int)keys.size() > p.reg_number) keys.resize(p.reg_number);
}

// DetectAffineKeypoints (scale-space-detector.cpp:43-85) for a batch of images
int detect_keypoints_batch(modsx_ctx *c, const modsx_image *const *imgs, int n, const modsx_hessaff_params &par,
                           const double *tilts, const double *zooms, std::vector<modsx_keypoint> *out) {
  modsx_hessaff_params p = par;  // reg_number is rescaled per image just before the export step (it only matters there)
  std::vector<modsx_sskp> ss[MAXB];
  int rc = detect_scalespace_batch(c, imgs, n, p, ss);
  if (rc) return rc;
  HostMark hm;
  rc = ensure_smm_mask(c, p.smmWindowSize);
  if (rc) return rc;
  size_t total = 0;
  for (int i = 0; i < n; i++) total += ss[i].size();
  for (int i = 0; i < n; i++) out[i].clear();
  if (!total) return MODSX_OK;
  hipStream_t s = c->stream;
  if (!c->hAff.ensure(total * sizeof(AffJob) + total * sizeof(AffOut))) return MODSX_ERR_NOMEM;
  if (!c->affJobs.ensure(total * sizeof(AffJob)) || !c->affOut.ensure(total * sizeof(AffOut))) return MODSX_ERR_NOMEM;
  AffJob *hj = (AffJob *)c->hAff.p;
  AffOut *ho = (AffOut *)((char *)c->hAff.p + total * sizeof(AffJob));
  size_t first[MAXB + 1];   // image i's keypoints are jobs first[i] .. first[i + 1]
  first[0] = 0;
  for (int i = 0; i < n; i++) first[i + 1] = first[i] + ss[i].size();
  host_parallel_light(n, [&](int i) {
    size_t k = first[i];
    for (const modsx_sskp &q : ss[i]) {
      const Octave &oc = c->pyr[i].oct[q.octave];
      AffJob &j = hj[k++];
      j.blur = oc.blur[q.level - 1];  // prevBlur: one level below the detection level (pyramid.cpp:428-429)
      j.rows = oc.rows; j.cols = oc.cols;
      j.x = q.x; j.y = q.y; j.s = q.s; j.pixelDistance = q.pixelDistance;
    }
  });
  hm.mark("AffJob build");
  if (p.doBaumberg) {
    MX_HIP(ctx_copy(c, c->affJobs.p, hj, total * sizeof(AffJob), hipMemcpyHostToDevice));
    ProfScope ps(c, K_BAUMBERG, (double)total * 361 * 4 * 2);
    launch_baumberg(s, (AffJob *)c->affJobs.p, (AffOut *)c->affOut.p, (int)total, c->dSmmMask, p.smmWindowSize,
                    p.maxIterations, p.convergenceThreshold, p.affInitialSigma);
    MX_HIP(ctx_copy(c, ho, c->affOut.p, total * sizeof(AffOut), hipMemcpyDeviceToHost));
    MX_HIP(ctx_sync(c));
  } else {
    for (size_t i = 0; i < total; i++) { ho[i].u11 = 1; ho[i].u12 = 0; ho[i].u21 = 0; ho[i].u22 = 1; ho[i].ok = 1; ho[i].iters = 0; }
  }
  hm.mark("baumberg launch + wait");
  host_parallel_light(n, [&](int i) {
    size_t k = first[i];
    out[i].reserve(ss[i].size());
    for (const modsx_sskp &q : ss[i]) {
      const AffOut &a = ho[k++];
      if (!a.ok) continue;
      modsx_keypoint kp;
      memset(&kp, 0, sizeof kp);
      kp.x = q.x; kp.y = q.y; kp.s = q.s;
      kp.a11 = a.u11; kp.a12 = a.u12; kp.a21 = a.u21; kp.a22 = a.u22;
      kp.response = q.val;
      kp.sub_type = q.type;
      out[i].push_back(kp);
    }
    modsx_hessaff_params pe = p;
    const double tilt = tilts ? tilts[i] : 1.0, zoom = zooms ? zooms[i] : 1.0;
    if ((tilt > 2.0) || (zoom < 0.5)) pe.reg_number = (int)floor(zoom * (double)pe.reg_number / tilt);
    prepare_keys_for_export(out[i], pe);
  });
  hm.mark("keypoints + export");
  return MODSX_OK;
}

// DetectAffineRegions<>, synth-detection.hpp:93-126
void detect_affine_regions(const modsx_keypoint *kps, int n, int img_id, int det_type, modsx_region *out) {
  for (int i = 0; i < n; i++) {
    modsx_keypoint k = kps[i];
    modsx_region &r = out[i];         // built in place (a 200-byte record)
    memset(&r, 0, sizeof r);
    r.img_id = img_id; r.img_reproj_id = 0; r.type = det_type; r.id = i;
    r.det_kp.s = k.s * sqrt(fabs(k.a11 * k.a22 - k.a12 * k.a21));
    rectify(k.a11, k.a12, k.a21, k.a22);
    r.det_kp.x = k.x; r.det_kp.y = k.y;
    r.det_kp.a11 = k.a11; r.det_kp.a12 = k.a12; r.det_kp.a21 = k.a21; r.det_kp.a22 = k.a22;
    r.det_kp.response = k.response;
    r.det_kp.sub_type = k.sub_type;
  }
}

static const double K_SIGMA = 2 * 3.0 * sqrt(3.0);  // synth-detection.cpp:28

// `Descriptors=` / `FGINNThreshold=` of the step: the step's own list, else the parameter block's, else {desc_type, match_ratio}
// ord[0..n): the classes of ds in descriptor-NAME order, the outer key of CorrespondencesMapMap (correspondencebank.cpp:117-179):
// "HalfRootSIFT" (3) < "HalfSIFT" (2) < "RootSIFT" (1) < "SIFT" (0)
void desc_class_order(const DescSet &ds, int *ord) {
  int m = 0;
  for (int type = 3; type >= 0; type--)
    for (int i = 0; i < ds.n; i++) if (ds.type[i] == type) ord[m++] = i;
}
int resolve_descs(const modsx_pair_params &pp, const modsx_ladder_step *st, DescSet &ds) {
  const int ns = st ? st->n_desc : 0;
  if (ns < 0 || ns > MODSX_MAX_DESC || pp.n_desc < 0 || pp.n_desc > MODSX_MAX_DESC) { set_error("n_desc must be 0..4"); return MODSX_ERR_ARG; }
  if (ns > 0) { ds.n = ns; for (int i = 0; i < ns; i++) { ds.type[i] = st->desc_types[i]; ds.ratio[i] = st->desc_ratios[i]; } }
  else if (pp.n_desc > 0) { ds.n = pp.n_desc; for (int i = 0; i < ds.n; i++) { ds.type[i] = pp.desc_types[i]; ds.ratio[i] = pp.desc_ratios[i]; } }
  else { ds.n = 1; ds.type[0] = pp.desc_type; ds.ratio[0] = st && st->match_ratio > 0 ? st->match_ratio : pp.match_ratio; }
  for (int i = 0; i < ds.n; i++) {
    if (ds.type[i] < 0 || ds.type[i] > 3) { set_error("descriptor type must be 0..3 (SIFT, RootSIFT, HalfSIFT, HalfRootSIFT)"); return MODSX_ERR_ARG; }
    for (int j = 0; j < i; j++) if (ds.type[j] == ds.type[i]) { set_error("a descriptor type is listed twice in one step"); return MODSX_ERR_ARG; }
  }
  return MODSX_OK;
}

static int upload_img_refs(modsx_ctx *c, const modsx_image *const *imgs, int n) {
  const bool fresh = !c->imgRefs.p;
  if (!c->imgRefs.ensure(MAXB * sizeof(ImgRef))) return MODSX_ERR_NOMEM;
  ImgRef refs[MAXB];
  memset(refs, 0, sizeof refs);
  for (int i = 0; i < n; i++) { refs[i].d = imgs[i]->d; refs[i].rows = imgs[i]->rows; refs[i].cols = imgs[i]->cols; }
  // orientation and description of a launch set name the same images: the table on the device is already right
  if (!fresh && memcmp(refs, c->imgRefsHost, sizeof refs) == 0) return MODSX_OK;
  // through a pinned copy of its own, in stream order and without a wait: the stream is idle when a stage begins (every stage ends
  // in a synchronize), so the previous table's copy has long completed when this buffer is written again
  if (!c->hRefs.ensure(sizeof refs)) return MODSX_ERR_NOMEM;
  memcpy(c->hRefs.p, refs, sizeof refs);
  MX_HIP(ctx_copy(c, c->imgRefs.p, c->hRefs.p, sizeof refs, hipMemcpyHostToDevice));
  memcpy(c->imgRefsHost, refs, sizeof refs);
  return MODSX_OK;
}

// DetectOrientation, synth-detection.cpp:841-919, for a batch of (image, region list)
int detect_orientation_batch(modsx_ctx *c, const modsx_image *const *imgs, int n, const std::vector<modsx_region> *in,
                             double mrSize, int patchSize, int doHalfSIFT, int maxAngNum, double th, int addUpRight,
                             std::vector<modsx_region> *out) {
  if (patchSize != 41) { set_error("orientation patchSize must be 41"); return MODSX_ERR_ARG; }
  for (int i = 0; i < n; i++) out[i].clear();
  double mrScale = (double)mrSize;
  int patchImageSize = 2 * int(mrScale) + 1;
  double imageToPatchScale = double(patchImageSize) / (double)patchSize;
  HostMark hm;
  std::vector<OriJob> jobs;
  std::vector<char> passed[MAXB];
  std::vector<OriJob> jobsOf[MAXB];     // per image, then concatenated: the images are independent (one pool task each)
  host_parallel_light(n, [&](int i) {
    passed[i].assign(in[i].size(), 0);
    jobsOf[i].clear();
    jobsOf[i].reserve(in[i].size());
    for (size_t r = 0; r < in[i].size(); r++) {
      const modsx_keypoint &k = in[i][r].det_kp;
      if (check_borders_host(imgs[i]->cols, imgs[i]->rows, (float)k.x, (float)k.y, (float)k.a11, (float)k.a12,
                             (float)k.a21, (float)k.a22, (int)(K_SIGMA * k.s), (int)(K_SIGMA * k.s)))
        continue;
      passed[i][r] = 1;
      if (maxAngNum > 0) {
        float curr_sc = imageToPatchScale * k.s;
        OriJob j;
        j.img = i; j.x = (float)k.x; j.y = (float)k.y;
        j.a11 = (float)k.a11 * curr_sc; j.a12 = (float)k.a12 * curr_sc;
        j.a21 = (float)k.a21 * curr_sc; j.a22 = (float)k.a22 * curr_sc;
        jobsOf[i].push_back(j);
      }
    }
  });
  size_t jobStart[MAXB + 1];
  jobStart[0] = 0;
  for (int i = 0; i < n; i++) jobStart[i + 1] = jobStart[i] + jobsOf[i].size();
  jobs.resize(jobStart[n]);
  host_parallel_light(n, [&](int i) { if (!jobsOf[i].empty()) memcpy(jobs.data() + jobStart[i], jobsOf[i].data(), jobsOf[i].size() * sizeof(OriJob)); });
  hm.mark("orientation jobs");
  const float *res = nullptr;   // in the pinned staging buffer: (1 + maxA) words per job
  const int maxA = maxAngNum < 0 ? ORI_MAX_PEAKS : std::min(maxAngNum, ORI_MAX_PEAKS);
  const size_t oriB = (size_t)(1 + maxA) * 4;
  if (!jobs.empty()) {
    int rc = upload_img_refs(c, imgs, n);
    if (rc) return rc;
    hipStream_t s = c->stream;
    size_t nj = jobs.size();
    if (!c->oriJobs.ensure(nj * sizeof(OriJob)) || !c->oriOut.ensure(nj * oriB) ||
        !c->hOri.ensure(nj * (sizeof(OriJob) + oriB)))
      return MODSX_ERR_NOMEM;
    // jobs up and results down through pinned memory: pageable transfers are staged and serialised by the runtime
    memcpy(c->hOri.p, jobs.data(), nj * sizeof(OriJob));
    float *hres = (float *)((char *)c->hOri.p + nj * sizeof(OriJob));
    res = hres;
    MX_HIP(ctx_copy(c, c->oriJobs.p, c->hOri.p, nj * sizeof(OriJob), hipMemcpyHostToDevice));
    ProfScope ps(c, K_ORIENT, (double)nj * 41 * 41 * 4);
    launch_orientation(s, (OriJob *)c->oriJobs.p, (float *)c->oriOut.p, (int)nj, (ImgRef *)c->imgRefs.p, c->dOriIdx,
                       c->dOriMask, c->dOriBinTab, doHalfSIFT, th, maxA);
    MX_HIP(ctx_copy(c, hres, c->oriOut.p, nj * oriB, hipMemcpyDeviceToHost));
    MX_HIP(ctx_sync(c));
  }
  hm.mark("orientation launch + wait");
  host_parallel_light(n, [&](int i) {
    size_t jk = jobStart[i];
    out[i].clear();
    out[i].reserve(in[i].size() + in[i].size() / 4);
    for (size_t r = 0; r < in[i].size(); r++) {
      if (!passed[i][r]) continue;
      // (a region is a 200-byte record: it is copied once, into its place in the list, and edited there)
      const modsx_region &base = in[i][r];
      if (maxAngNum > 0) {
        const float *o = res + (jk++) * (size_t)(1 + maxA);
        int on;
        memcpy(&on, o, 4);
        const double b11 = base.det_kp.a11, b12 = base.det_kp.a12, b21 = base.det_kp.a21, b22 = base.det_kp.a22;
        for (int a = 0; a < on; a++) {
          // `using namespace std` in synth-detection.cpp:30 => cos/sin(float) are the f32 overloads
          double ci = cosf(-o[1 + a]);
          double si = sinf(-o[1 + a]);
          out[i].push_back(base);
          modsx_region &t = out[i].back();
          t.id = 0;  // const_temp_region.id = count, count is never incremented (synth-detection.cpp:854,889)
          t.det_kp.a11 = b11 * ci - b12 * si;
          t.det_kp.a12 = b11 * si + b12 * ci;
          t.det_kp.a21 = b21 * ci - b22 * si;
          t.det_kp.a22 = b21 * si + b22 * ci;
        }
      }
      if (addUpRight) { out[i].push_back(base); if (maxAngNum > 0) out[i].back().id = 0; }
    }
  });
  hm.mark("rotated regions");
  return MODSX_OK;
}

// ReprojectRegions, synth-detection.cpp:541-616 (box = k_sigma * s), and ReprojectRegionsAndRemoveTouchBoundary, :63-102
// (box = mrSize * s, default 3 sqrt 3: the "None" list of imagerepresentation.cpp:1271-1272)
int reproject_regions(modsx_region *regs, int n, const double *H, int orig_w, int orig_h) {
  return reproject_regions_box(regs, n, H, orig_w, orig_h, K_SIGMA);
}
int reproject_regions_box(modsx_region *regs, int n, const double *H, int orig_w, int orig_h, double boxk) {
  double eyeTest = fabs(H[0] - 1.0) + fabs(H[1]) + fabs(H[2]) + fabs(H[3]) + fabs(H[4] - 1.0) + fabs(H[5]) + fabs(H[6]) +
                   fabs(H[7]) + fabs(H[8] - 1.0);
  double Hi[9];
  invert3(H, Hi);
  for (int i = 0; i < n; i++) {
    regs[i].reproj_kp = regs[i].det_kp;
    if (!(eyeTest < 0.01)) {
      const modsx_keypoint k = regs[i].det_kp;
      modsx_keypoint &o = regs[i].reproj_kp;
      o.x = (Hi[0] * k.x + Hi[1] * k.y + Hi[2]);
      o.y = (Hi[3] * k.x + Hi[4] * k.y + Hi[5]);
      o.a11 = (Hi[0] * k.a11 + Hi[1] * k.a21);
      o.a12 = (Hi[0] * k.a12 + Hi[1] * k.a22);
      o.a21 = (Hi[3] * k.a11 + Hi[4] * k.a21);
      o.a22 = (Hi[3] * k.a12 + Hi[4] * k.a22);
    }
  }
  int m = 0;
  for (int i = 0; i < n; i++) {
    const modsx_keypoint &k = regs[i].reproj_kp;
    if ((k.x < orig_w) && (k.y < orig_h) && (k.x > 0) && (k.y > 0)) {
      if (!check_borders_host(orig_w, orig_h, (float)k.x, (float)k.y, (float)k.a11, (float)k.a12, (float)k.a21,
                              (float)k.a22, (int)(boxk * k.s), (int)(boxk * k.s)))
      { if (m != i) regs[m] = regs[i]; m++; }
    }
  }
  return m;
}

// DescribeRegions<SIFTDescriptor>, synth-detection.hpp:169-255, for a batch.  Descriptors stay in HBM
// (c->descF[i], c->descU8[i]); descHost[i] (optional) receives the f32 copy.
int describe_batch(modsx_ctx *c, const modsx_image *const *imgs, int n, const std::vector<modsx_region> *regs,
                   double mrSize, int patchSize, int fast, int photoNorm, int descType, double maxBin,
                   float *const *descHost, float *const *devF, uint8_t *const *devU8, const DescSet *ds,
                   uint8_t *const *const *devU8x) {
  if (patchSize != 41) { set_error("descriptor patchSize must be 41"); return MODSX_ERR_ARG; }
  DescSet one;
  if (!ds) { one.n = 1; one.type[0] = descType; ds = &one; }
  if (ds->n < 1 || ds->n > MODSX_MAX_DESC) { set_error("describe: 1..4 descriptor classes"); return MODSX_ERR_ARG; }
  if (n > MAXB) { set_error("describe batch too large"); return MODSX_ERR_ARG; }
  hipStream_t s = c->stream;
  int rc = upload_img_refs(c, imgs, n);
  if (rc) return rc;
  // Window arena per chunk.  Measured on MI355X with 16 contexts: throughput is flat up to 256 MiB per context and
  // halves from 384 MiB on (the three arenas of all contexts stop fitting the 256 MiB Infinity Cache working set),
  // so a batch is cut into chunks of 192 MiB of windows.  MODSX_ARENA_MB overrides it for experiments.
  static const size_t arenaMB = getenv("MODSX_ARENA_MB") ? (size_t)atol(getenv("MODSX_ARENA_MB")) : 192;
  const size_t ARENA_FLOATS = std::max<size_t>(arenaMB, 16) << 18;  // MiB -> floats
  DescOut outs;
  memset(&outs, 0, sizeof outs);
  for (int i = 0; i < n; i++) {
    const size_t nr = regs[i].size();
    float *outF = devF ? devF[i] : nullptr;
    uint8_t *outU8 = devU8 ? devU8[i] : nullptr;
    if (!outF) { if (!c->descF[i].ensure(std::max<size_t>(1, nr) * 128 * 4)) return MODSX_ERR_NOMEM; outF = (float *)c->descF[i].p; }
    if (!outU8) { if (!c->descU8[i].ensure(std::max<size_t>(1, nr) * 128)) return MODSX_ERR_NOMEM; outU8 = (uint8_t *)c->descU8[i].p; }
    outs.f[i] = outF; outs.u8[i] = outU8;
    for (int k = 1; k < ds->n; k++) {
      uint8_t *o = devU8x && devU8x[k - 1] ? devU8x[k - 1][i] : nullptr;
      if (!o) { if (!c->descU8x[k - 1][i].ensure(std::max<size_t>(1, nr) * 128)) return MODSX_ERR_NOMEM; o = (uint8_t *)c->descU8x[k - 1][i].p; }
      outs.u8x[k - 1][i] = o;
    }
  }
  // the regions of all images of the batch go through one launch set per chunk (a chunk ends when the window arena is
  // full); region order inside an image is kept, outIdx addresses the image's own descriptor buffer
  // The window size of every region, once for the whole batch (one pool task per image): P = patchImageSize + 2 of the
  // smoothed branch, 0 for the direct branch (imageToPatchScale <= 0.4, or fast extraction)
  HostMark hm;
  std::vector<int> winP[MAXB];
  host_parallel_light(n, [&](int i) {
    winP[i].resize(regs[i].size());
    for (size_t r = 0; r < regs[i].size(); r++) {
      int P = 0;
      if (!fast) {
        const modsx_keypoint &k = regs[i][r].det_kp;
        float mrScale = (float)ceil(k.s * mrSize);
        int patchImageSize = 2 * int(mrScale) + 1;
        float i2p = float(patchImageSize) / float(patchSize);
        if (i2p > 0.4) P = patchImageSize + 2;
      }
      winP[i][r] = P;
    }
  });
  int curImg = 0, chunkNo = 0;
  size_t curReg = 0;
  while (curImg < n && regs[curImg].empty()) curImg++;
  while (curImg < n) {
    {
      std::vector<DescJob> jobs;
      std::vector<int> pfxSample(1, 0), pfxRow(1, 0), pfxCol(1, 0), pfxRowL(1, 0), pfxColL(1, 0);
      std::vector<float> taps, coordTab;
      std::vector<int> needTab;
      struct PInfo { int tapOfs, ksize, needOfs, NC, coordOfs, touch, rows0, ro1; };
      std::map<int, PInfo> pinfo;  // per window size P
      size_t arenaA = 0, arenaB = 0, arenaC = 0;
      bool full = false;
      // the tables of a window size (taps, needed columns, sample coordinates, tile shapes), built when the size first appears
      auto make_pinfo = [&](int P, PInfo &pi) -> int {
        const float i2p = float(P - 2) / float(patchSize);
        float sigma = 1.5f * i2p;
        pi.ksize = blur_ksize(sigma);
        if (pi.ksize > 512) { set_error("descriptor window too large (blur kernel > 512 taps)"); return MODSX_ERR_ARG; }
        std::vector<float> kk = gaussian_kernel(pi.ksize, sigma);
        pi.tapOfs = (int)taps.size();
        taps.insert(taps.end(), kk.begin(), kk.end());
        // coordinates of interpolate(smoothed, P/2, P/2, i2p, 0, 0, i2p, patch41): f32 running sums
        // (helpers.cpp:563-585); rows and columns run the same recurrence (a12 = a21 = 0, ofsx = ofsy)
        const float o = (float)(P >> 1);
        pi.touch = check_borders_host(P, P, o, o, i2p, 0.f, 0.f, i2p, 41, 41) ? 1 : 0;
        float W[41];
        {
          float rx = o - (float)20 * 0.f;
          float WX = rx - (float)20 * i2p;
          for (int q = 0; q < 41; q++) { W[q] = WX; WX += i2p; }
        }
        int x0[41], valid[41];
        std::vector<int> need;
        for (int q = 0; q < 41; q++) {
          if (!pi.touch) {
            int x = (int)W[q];
            x = x < 0 ? 0 : (x > P - 2 ? P - 2 : x);
            x0[q] = x; valid[q] = 1;
          } else {
            int x = (int)floorf(W[q]);
            valid[q] = (W[q] >= 0 && x < P - 1) ? 1 : 0;
            x0[q] = valid[q] ? x : 0;
          }
          if (valid[q]) { need.push_back(x0[q]); need.push_back(x0[q] + 1); }
        }
        std::sort(need.begin(), need.end());
        need.erase(std::unique(need.begin(), need.end()), need.end());
        if (need.empty()) need.push_back(0);
        pi.NC = (int)need.size();
        pi.needOfs = (int)needTab.size();
        needTab.insert(needTab.end(), need.begin(), need.end());
        for (int q = 0; q < 41; q++) {
          int i0 = 0, i1 = 0;
          if (valid[q]) {
            i0 = (int)(std::lower_bound(need.begin(), need.end(), x0[q]) - need.begin());
            i1 = (int)(std::lower_bound(need.begin(), need.end(), x0[q] + 1) - need.begin());
          }
          needTab.push_back(i0); needTab.push_back(i1); needTab.push_back(x0[q]); needTab.push_back(valid[q]);
        }
        pi.coordOfs = (int)coordTab.size();
        coordTab.insert(coordTab.end(), W, W + 41);
        {  // tile shapes of the LDS blur kernels: <= BLUR_OUT outputs and <= BLUR_LDS floats per workgroup
          const int BLUR_LDS = MODSX_SR_WIN, BLUR_LDS_C = MODSX_BLUR_LDS_C, R = pi.ksize >> 1, NP2 = 2 * ((pi.NC + 1) / 2);
          const int cap = 2048 / NP2, capC = 4096 / NP2;
          // the row filter pairs needed columns (2m, 2m+1); they are neighbours in the window by construction
          // (x0, x0 + 1 of one sample, or a contiguous range) -- if ever not, the job takes the global-memory kernel
          bool pairs = true;
          for (int a = 0; a + 1 < pi.NC; a += 2) pairs = pairs && need[a + 1] == need[a] + 1;
          pi.rows0 = pairs ? std::min(cap, BLUR_LDS / (P + 2 * R)) : 0;
          if (pi.rows0 < 2) pi.rows0 = 0;
          if (pi.rows0 > 32 && pi.rows0 < 48 && pi.rows0 < P) pi.rows0 = 32;   // the fused sampling kernel parks 8 columns x <= 32 rows or 4 x <= 64 (MODSX_SR_HALF)
          pi.ro1 = 0;
          const int LS = pi.NC <= 64 ? 64 : 96;   // LDS row stride of the column filter
          for (int ro = std::min(capC, pi.NC); ro >= 2 && !pi.ro1 && pi.NC <= 96; ro--) {
            int span = 0;
            for (int a = 0; a < pi.NC; a += ro) span = std::max(span, need[std::min(a + ro, pi.NC) - 1] - need[a] + 2 * R + 1);
            if (span * LS <= BLUR_LDS_C) pi.ro1 = ro;
          }
          // a window that is one row tile, with <= 64 needed columns and <= 80 block rows (kernels_describe.hip: FC_LS,
          // FC_ROWS): the fused sampling kernel runs the column filter too
          if (pi.rows0 >= P && pi.NC <= 64 && P + 2 * R <= MODSX_FC_ROWS) pi.ro1 = -1;
        }
        return MODSX_OK;
      };
      // which regions this chunk takes (a light sequential walk over the window sizes), then the job records in parallel
      size_t beg[MAXB], end[MAXB], at[MAXB + 1];
      for (int q = 0; q < n; q++) { beg[q] = end[q] = 0; }
      int i = curImg;
      size_t r = curReg, count = 0;
      for (; i < n && !full; i++, r = 0) {
        beg[i] = r;
        for (; r < regs[i].size(); r++) {
          const int P = winP[i][r];
          if (P > 0) {
            if (pinfo.find(P) == pinfo.end()) {
              PInfo pi;
              const int prc = make_pinfo(P, pi);
              if (prc) return prc;
              pinfo.insert({P, pi});
            }
            const size_t needA = (size_t)P * P;
            if (arenaA + needA > ARENA_FLOATS && count) { full = true; break; }
            arenaA += needA;
          }
          count++;
        }
        end[i] = r;
        if (full) break;
      }
      at[0] = 0;
      for (int q = 0; q < n; q++) at[q + 1] = at[q] + (end[q] - beg[q]);
      jobs.resize(count);
      host_parallel_light(n, [&](int q) {
        for (size_t rr = beg[q]; rr < end[q]; rr++) {
          const modsx_keypoint &k = regs[q][rr].det_kp;
          DescJob j;
          memset(&j, 0, sizeof j);
          j.img = q;
          j.outIdx = (int)rr;
          j.x = (float)k.x; j.y = (float)k.y;
          if (!fast) {
            float mrScale = (float)ceil(k.s * mrSize);
            int patchImageSize = 2 * int(mrScale) + 1;
            float i2p = float(patchImageSize) / float(patchSize);
            j.i2p = i2p;
            const int P = winP[q][rr];
            if (P > 0) {
              const PInfo &pi = pinfo.find(P)->second;
              j.P = P;
              j.a11 = (float)k.a11; j.a12 = (float)k.a12; j.a21 = (float)k.a21; j.a22 = (float)k.a22;
              j.tapOfs = pi.tapOfs; j.ksize = pi.ksize; j.NC = pi.NC; j.needOfs = pi.needOfs; j.coordOfs = pi.coordOfs;
              j.touch = pi.touch; j.rows0 = pi.rows0; j.ro1 = pi.ro1;
            } else {
              j.P = 0;
              j.a11 = (float)k.a11 * i2p; j.a12 = (float)k.a12 * i2p; j.a21 = (float)k.a21 * i2p; j.a22 = (float)k.a22 * i2p;
            }
          } else {
            double mrScale = (double)mrSize * k.s;
            int patchImageSize = 2 * int(mrScale) + 1;
            double i2pd = double(patchImageSize) / (double)patchSize;
            float curr_sc = i2pd;
            j.P = 0; j.i2p = curr_sc;
            j.a11 = (float)k.a11 * curr_sc; j.a12 = (float)k.a12 * curr_sc; j.a21 = (float)k.a21 * curr_sc; j.a22 = (float)k.a22 * curr_sc;
          }
          jobs[at[q] + (rr - beg[q])] = j;
        }
      });
      // (i, r) = first region that did not fit, or i == n
      if (full) { curImg = i; curReg = r; } else { curImg = n; curReg = 0; }
      hm.mark("desc jobs");
      const size_t nj = jobs.size();
      // Launch order of the chunk: by image, then by 64-pixel row band, then by x.  The sampling kernel hands every XCD one
      // contiguous eighth of this order (kernels_describe.hip: xcd_chunk), i.e. one part of the images; outIdx keeps every
      // descriptor at its region's place, so the reference's list order is untouched.
      // (a stable LSD radix sort of one key per job: jobs are generated in (image, outIdx) order, which breaks the ties)
      {
        std::vector<uint64_t> key(nj), key2(nj);
        std::vector<uint32_t> ord(nj), ord2(nj);
        for (size_t q = 0; q < nj; q++) {
          const DescJob &a = jobs[q];
          // the order only places neighbouring windows on neighbouring workgroups (no result depends on it): whole pixels
          // are enough, and a 32-bit key is three passes instead of six
          const int xi = (int)a.x, yb = (int)a.y >> 6;
          const uint32_t x16 = (uint32_t)(xi < 0 ? 0 : (xi > 65535 ? 65535 : xi));
          const uint32_t band = (uint32_t)(yb < 0 ? 0 : (yb > 1023 ? 1023 : yb));
          key[q] = ((uint64_t)(uint32_t)a.img << 26) | ((uint64_t)band << 16) | x16;
          ord[q] = (uint32_t)q;
        }
        constexpr int BITS = 11, NB = 1 << BITS;
        uint32_t hist[NB];
        for (int shift = 0; shift < 33; shift += BITS) {
          memset(hist, 0, sizeof hist);
          for (size_t q = 0; q < nj; q++) hist[(key[q] >> shift) & (NB - 1)]++;
          if (nj && hist[(key[0] >> shift) & (NB - 1)] == nj) continue;
          uint32_t sum = 0;
          for (int b = 0; b < NB; b++) { const uint32_t h = hist[b]; hist[b] = sum; sum += h; }
          for (size_t q = 0; q < nj; q++) {
            const uint32_t d = hist[(key[q] >> shift) & (NB - 1)]++;
            key2[d] = key[q]; ord2[d] = ord[q];
          }
          key.swap(key2); ord.swap(ord2);
        }
        std::vector<DescJob> sorted(nj);
        for (size_t q = 0; q < nj; q++) sorted[q] = jobs[ord[q]];
        jobs.swap(sorted);
      }
      hm.mark("desc job sort");
      const size_t windowFloats = arenaA;   // all P x P windows of the chunk: its size limit and its algorithmic bytes
      arenaA = 0;                           // arena A itself only holds the windows that do not take the fused kernel
      size_t rowStarts = 0;                 // the fused ones get their P row starts (float2) instead
      for (DescJob &j : jobs) {
        if (j.P > 0) {
          j.rowOfs = arenaB; j.gridOfs = arenaC;
          if (!j.rows0) { j.scratchOfs = arenaA; arenaA += (size_t)j.P * j.P; }
          else { j.scratchOfs = rowStarts; rowStarts += (size_t)j.P; }
          arenaB += (size_t)j.P * j.NC; arenaC += (size_t)j.NC * j.NC;
        }
        // windows whose row tile fits LDS are sampled by the fused sample + row-filter kernel (arena A is not touched); the
        // others go through k_patch_sample (64 x SAMPLE_COLS tiles) and the global-memory row filter
        pfxSample.push_back(pfxSample.back() + (j.P > 0 && !j.rows0 ? ((j.P + 63) / 64) * ((j.P + 127) / 128) : 0));
        // the blur passes: LDS kernels where a tile fits, k_patch_blur (BLUR_TILE outputs per workgroup) otherwise
        pfxRowL.push_back(pfxRowL.back() + (j.P > 0 && j.rows0 ? (j.P + j.rows0 - 1) / j.rows0 : 0));
        pfxColL.push_back(pfxColL.back() + (j.P > 0 && j.ro1 > 0 ? (j.NC + j.ro1 - 1) / j.ro1 : 0));
        pfxRow.push_back(pfxRow.back() + (j.P > 0 && !j.rows0 ? (j.P * j.NC + 1023) / 1024 : 0));
        pfxCol.push_back(pfxCol.back() + (j.P > 0 && j.ro1 == 0 ? (j.NC * j.NC + 1023) / 1024 : 0));
      }
      // the job table, the five tile prefixes and the three small tables travel as ONE pinned blob and one copy: nine
      // separate uploads cost nine ~6 us copy kernels per chunk on the stream
      auto up16 = [](size_t b) { return (b + 15) & ~(size_t)15; };
      const size_t oJobs = 0, oPfx = up16(nj * sizeof(DescJob)), pfxB = up16((nj + 1) * 4);
      const size_t oTaps = oPfx + 5 * pfxB, oNeed = oTaps + up16(taps.size() * 4), oCoord = oNeed + up16(needTab.size() * 4);
      const size_t blobB = oCoord + up16(coordTab.size() * 4) + 16;
      // two staging blobs in turn: the copy of chunk k may still be in flight while chunk k + 1 is being prepared
      const int slot = chunkNo & 1;
      PinBuf &hblob = slot ? c->hDescB : c->hDesc;
      if (c->descEvPending[slot]) { MX_HIP(c->descByEvent[slot] ? hipEventSynchronize(c->descEv[slot]) : ctx_wait_mark(c, c->descMark[slot])); c->descEvPending[slot] = false; }
      if (!c->descJobs.ensure(blobB) || !hblob.ensure(blobB) ||
          !c->scratchA.ensure(std::max<size_t>(1, arenaA) * 4) || !c->scratchB.ensure(std::max<size_t>(1, arenaB) * 4) ||
          !c->scratchC.ensure(std::max<size_t>(1, arenaC) * 4) || !c->rowStarts.ensure(std::max<size_t>(1, rowStarts) * 8) ||
          !c->tileJob.ensure(((size_t)pfxSample.back() + pfxRow.back() + pfxCol.back() + 3) * 4))
        return MODSX_ERR_NOMEM;
      int *tjS = (int *)c->tileJob.p, *tjR = tjS + pfxSample.back(), *tjC = tjR + pfxRow.back();
      if (!c->blurTiles.ensure(((size_t)pfxRowL.back() + pfxColL.back() + 1) * sizeof(BlurTile))) return MODSX_ERR_NOMEM;
      BlurTile *btR = (BlurTile *)c->blurTiles.p, *btC = btR + pfxRowL.back();
      char *hb = (char *)hblob.p, *db = (char *)c->descJobs.p;
      memcpy(hb + oJobs, jobs.data(), nj * sizeof(DescJob));
      const std::vector<int> *pf[5] = {&pfxSample, &pfxRow, &pfxCol, &pfxRowL, &pfxColL};
      for (int q = 0; q < 5; q++) memcpy(hb + oPfx + q * pfxB, pf[q]->data(), (nj + 1) * 4);
      if (!taps.empty()) memcpy(hb + oTaps, taps.data(), taps.size() * 4);
      if (!needTab.empty()) memcpy(hb + oNeed, needTab.data(), needTab.size() * 4);
      if (!coordTab.empty()) memcpy(hb + oCoord, coordTab.data(), coordTab.size() * 4);
      hm.mark("desc tables + blob");
      MX_HIP(ctx_copy(c, db, hb, blobB, hipMemcpyHostToDevice));
      c->descByEvent[slot] = c->waitRuntime || !c->hFlag;
      if (c->descByEvent[slot]) MX_HIP(hipEventRecord(c->descEv[slot], s)); else c->descMark[slot] = ctx_mark(c);
      c->descEvPending[slot] = true;
      int *dPfxS = (int *)(db + oPfx), *dPfxR = (int *)(db + oPfx + pfxB), *dPfxC = (int *)(db + oPfx + 2 * pfxB);
      int *dPfxRL = (int *)(db + oPfx + 3 * pfxB), *dPfxCL = (int *)(db + oPfx + 4 * pfxB);
      float *dTaps = (float *)(db + oTaps), *dCoord = (float *)(db + oCoord);
      int *dNeed = (int *)(db + oNeed);
      const DescJob *dj = (const DescJob *)c->descJobs.p;
      // tile -> job tables of the global-memory fallbacks: most chunks have no such tiles at all
      if (pfxSample.back()) launch_expand_tiles(s, dPfxS, (int)nj, tjS);
      if (pfxRow.back()) launch_expand_tiles(s, dPfxR, (int)nj, tjR);
      if (pfxCol.back()) launch_expand_tiles(s, dPfxC, (int)nj, tjC);
      // algorithmic work of the describe STAGE per SURVEY section 8(d): the (P+2)^2 f32 window of every region read once
      // (booked here) + 128 B written per region (booked on k_describe); the arenas between the four kernels are an
      // artefact of the split and are not algorithmic bytes
      if (pfxRowL.back() + pfxColL.back())
        launch_expand_blur_tiles(s, dj, dPfxRL, dPfxCL, (int)nj, dNeed, btR, btC, (float2 *)c->rowStarts.p);
      { ProfScope ps(c, K_PATCH_SAMPLE, (double)windowFloats * 4);
        launch_sample_rows(s, dj, btR, pfxRowL.back(), (ImgRef *)c->imgRefs.p, dTaps, dNeed, (float *)c->scratchB.p,
                           (const float2 *)c->rowStarts.p, (float *)c->scratchC.p);
        launch_patch_sample(s, dj, dPfxS, tjS, pfxSample.back(), (ImgRef *)c->imgRefs.p, (float *)c->scratchA.p); }
      { ProfScope ps(c, K_BLUR_ROWS, 0.0);
        launch_patch_blur(s, dj, dPfxR, tjR, pfxRow.back(), dTaps, dNeed, (float *)c->scratchA.p,
                          (float *)c->scratchB.p, 0); }
      { ProfScope ps(c, K_BLUR_COLS, 0.0);
        launch_blur_cols(s, btC, pfxColL.back(), dTaps, dNeed, (float *)c->scratchB.p, (float *)c->scratchC.p);
        launch_patch_blur(s, dj, dPfxC, tjC, pfxCol.back(), dTaps, dNeed, (float *)c->scratchB.p,
                          (float *)c->scratchC.p, 1); }
      ProfScope psd(c, K_DESCRIBE, (double)nj * 128 * ds->n);
      launch_describe(s, dj, (int)nj, (ImgRef *)c->imgRefs.p, (float *)c->scratchC.p, dNeed, dCoord,
                      c->dSiftMask, c->dSiftMaskIdx, c->nSiftMask, c->dSiftOTab,
                      c->dSiftBins, c->dSiftW, photoNorm, ds->packed(), ds->n, maxBin, outs);
      chunkNo++;
    }
  }
  hm.mark("desc launches");
  MX_HIP(ctx_sync(c));   // callers read the descriptor buffers and reuse the staging blobs
  hm.mark("desc wait");
  c->descEvPending[0] = c->descEvPending[1] = false;
  if (descHost) {
    bool any = false;
    for (int i = 0; i < n; i++)
      if (descHost[i] && !regs[i].empty()) {
        MX_HIP(hipMemcpyAsync(descHost[i], outs.f[i], regs[i].size() * 128 * 4, hipMemcpyDeviceToHost, s));
        any = true;
      }
    if (any) MX_HIP(hipStreamSynchronize(s));
  }
  MX_HIP(hipGetLastError());
  return MODSX_OK;
}

// the host half of the matcher: per-query result rows -> TentativeCorrespExt records (matching.cpp:435-457)
void rows_to_tentatives(const MatchRow *rows, int n1, int nn, std::vector<modsx_tentative> &o) {
  o.reserve(n1 / 4 + 16);
  for (int q = 0; q < n1; q++) {
    const MatchRow &r = rows[q];
    // rank of the first ratio-passing neighbour is nless+1; it must be <= nn-1 and every neighbour
    // before it must lie within contradDist of NN0 (matching.cpp:435-457)
    if (r.t0 < 0 || r.tj < 0 || r.nbad != 0 || r.nless > nn - 2) continue;
    modsx_tentative t;
    t.q = q; t.t0 = r.t0; t.tj = r.tj;
    t.t1 = r.t1;
    t.d1 = r.d0; t.d2 = r.dj; t.d2by2ndcl = r.d1;
    double ratio = r.d0 / r.dj;  // f32 / f32, then widened (matching.cpp:437)
    t.ratio = sqrt(ratio);
    o.push_back(t);
  }
}

// MatchFlannFGINN (matching/matching.cpp:357-461, linear index) on descriptors resident in HBM, for nb <= MATCH_MAXB
// independent (query set, train set) problems that share the kernel launches (blockIdx.z) and one synchronisation
int match_device_batch(modsx_ctx *c, int nb, const uint8_t *const *d1, const int *n1, const uint8_t *const *d2, const int *n2,
                       const double *const *pos2Host, double ratioT, double contradDist, int nn,
                       std::vector<modsx_tentative> *out, const MatchShard *shard, const double *const *pos2Dev) {
  // pos2Dev (optional, unsharded branch): the positions already live on the device; pos2Host is then not read
  CtxBusy busy(c);
  if (nb < 1 || nb > MATCH_MAXB) { set_error("match_device_batch: batch size"); return MODSX_ERR_ARG; }
  hipStream_t s = c->stream;
  const double sqminratio = ratioT * ratioT, contrDistSq = contradDist * contradDist;
  if (!(sqminratio == sqminratio)) { set_error("match ratio is NaN"); return MODSX_ERR_ARG; }   // ratio >= 1: the "all points" branch (matching.cpp:397-428)
  // nn = neighbours the walk may look at (default 50, matching.hpp:268-269); the event lists of the device matcher hold up to MATCH_NN_MAX groups
  if (nn < 2 || nn > MATCH_NN_MAX) { set_error("match: nn must be in [2, 256]"); return MODSX_ERR_ARG; }
  for (int i = 0; i < nb; i++)      // the matcher logs train tiles as 16-bit numbers (kernels_match.hip k_match_resolve)
    if (n2[i] > 2000000) { set_error("match: more than 2 000 000 train descriptors in one problem"); return MODSX_ERR_ARG; }
  auto up = [](size_t b) { return (b + 255) & ~(size_t)255; };
  if (shard) {
    // view-sharded run (engine_shard.hip): this rank matches the query rows [lo, lo + per) of ONE problem; the result rows
    // of all ranks are all-gathered on the device and every rank builds the full tentative list
    if (nb != 1) { set_error("match_device_batch: a sharded match takes one problem"); return MODSX_ERR_ARG; }
    out[0].clear();
    const int N1 = shard->n1_total, M = n2[0], per = shard->per;
    const int lo = shard->lo, nloc = std::max(0, std::min(N1, lo + per) - lo);
    MatchRow *blk = nullptr;
    int rc = match_shard_begin(c, *shard, &blk);          // the lane's blocks (growth agreed by all ranks)
    if (rc) return rc;
    // from here to the all-gather nothing returns: a local failure travels in the block header
    int lrc = MODSX_OK;
    const int world = shard->world;
    const size_t posB = up((size_t)M * 16), allB = (size_t)world * (per + 1) * sizeof(MatchRow);
    if (!c->pos2.ensure(posB) || !c->hMatch.ensure(posB + up(allB)) || !c->matchWork.ensure(match_workspace_bytes(std::max(1, nloc), M)))
      lrc = MODSX_ERR_NOMEM;
    char *hpos = (char *)c->hMatch.p, *hrow = hpos ? hpos + posB : nullptr;
    if (!lrc) {
      memcpy(hpos, pos2Host[0], (size_t)M * 16);
      if (ctx_copy(c, c->pos2.p, hpos, (size_t)M * 16, hipMemcpyHostToDevice) != hipSuccess) { set_error("sharded match: upload failed"); lrc = MODSX_ERR_DEVICE; }
    }
    if (!lrc && nloc > 0) {
      ProfScope ps(c, K_MATCH, 2.0 * nloc * (double)M * 128);
      launch_match(s, d1[0] + (size_t)lo * 128, nloc, d2[0], M, (const double *)c->pos2.p, sqminratio, contrDistSq, nn, blk + 1, c->matchWork.p);
    }
    rc = match_shard_gather(c, *shard, lrc, lrc ? nullptr : (MatchRow *)hrow);
    if (rc) return rc;
    // rank r's rows sit behind its header row; the last ranks may hold fewer rows, or none
    std::vector<MatchRow> rows((size_t)N1);
    for (int r = 0; r < world; r++) {
      const int rlo = std::min(N1, r * per), rn = std::min(N1, rlo + per) - rlo;
      if (rn > 0) memcpy(rows.data() + rlo, (const MatchRow *)hrow + (size_t)r * (per + 1) + 1, (size_t)rn * sizeof(MatchRow));
    }
    rows_to_tentatives(rows.data(), N1, nn, out[0]);
    return MODSX_OK;
  }
  size_t posOfs[MATCH_MAXB], rowOfs[MATCH_MAXB], workOfs[MATCH_MAXB], posB = 0, rowB = 0, workB = 0;
  int live[MATCH_MAXB], nl = 0;
  for (int i = 0; i < nb; i++) {
    out[i].clear();
    if (n1[i] <= 0 || n2[i] <= 0) continue;
    posOfs[nl] = posB; posB += up((size_t)n2[i] * 16);
    rowOfs[nl] = rowB; rowB += up((size_t)n1[i] * sizeof(MatchRow));
    workOfs[nl] = workB; workB += up(match_workspace_bytes(n1[i], n2[i]));
    live[nl++] = i;
  }
  if (!nl) return MODSX_OK;
  if (!c->pos2.ensure(posB) || !c->matchRows.ensure(rowB) || !c->matchWork.ensure(workB) || !c->hMatch.ensure(posB + rowB))
    return MODSX_ERR_NOMEM;
  char *hpos = (char *)c->hMatch.p, *hrow = hpos + posB;   // pinned staging: positions up, rows down
  const uint8_t *pd1[MATCH_MAXB], *pd2[MATCH_MAXB];
  const double *ppos[MATCH_MAXB];
  MatchRow *prow[MATCH_MAXB];
  void *pwork[MATCH_MAXB];
  int pn1[MATCH_MAXB], pn2[MATCH_MAXB];
  double work = 0;
  for (int k = 0; k < nl; k++) {
    const int i = live[k];
    pd1[k] = d1[i]; pd2[k] = d2[i]; pn1[k] = n1[i]; pn2[k] = n2[i];
    ppos[k] = pos2Dev ? pos2Dev[i] : (const double *)((char *)c->pos2.p + posOfs[k]);
    prow[k] = (MatchRow *)((char *)c->matchRows.p + rowOfs[k]);
    pwork[k] = (char *)c->matchWork.p + workOfs[k];
    if (!pos2Dev) memcpy(hpos + posOfs[k], pos2Host[i], (size_t)n2[i] * 16);
    work += 2.0 * n1[i] * (double)n2[i] * 128;
  }
  if (!pos2Dev) MX_HIP(ctx_copy(c, c->pos2.p, hpos, posB, hipMemcpyHostToDevice));
  {
    // K_MATCH = every launch of the problem(s); K_MATCH_SWEEP1 = the one launch that carries the 2 N M 128 contraction
    hipEvent_t evS1[2];
    const bool tS1 = prof_reserve(c, K_MATCH_SWEEP1, work, evS1);
    ProfScope ps(c, K_MATCH, work);
    launch_match_batch(s, nl, pd1, pn1, pd2, pn2, ppos, sqminratio, contrDistSq, nn, prow, pwork, tS1 ? evS1 : nullptr);
  }
  MX_HIP(ctx_copy(c, hrow, c->matchRows.p, rowB, hipMemcpyDeviceToHost));
  MX_HIP(ctx_sync(c));
  MX_HIP(hipGetLastError());
  for (int k = 0; k < nl; k++) rows_to_tentatives((const MatchRow *)(hrow + rowOfs[k]), pn1[k], nn, out[live[k]]);
  return MODSX_OK;
}

int match_device(modsx_ctx *c, const uint8_t *d1, int n1, const uint8_t *d2, int n2, const double *pos2Host,
                 double ratioT, double contradDist, int nn, std::vector<modsx_tentative> &out) {
  return match_device_batch(c, 1, &d1, &n1, &d2, &n2, &pos2Host, ratioT, contradDist, nn, &out, nullptr);
}

// The matcher computes on u8: the reference's SIFT-family descriptors hold the integers 0..255
// ((int)(512 v + 0.5) clamped, siftdesc.cpp:218-274).  Anything else (fractions, values out of range, NaN) would be
// matched with different distances than FLANN's float L2, so it is refused instead of being truncated silently.
static bool desc_f32_to_u8(const float *f, size_t n, uint8_t *u) {
  bool ok = true;
  for (size_t i = 0; i < n; i++) {
    const float v = f[i];
    const int b = (v >= 0.f && v <= 255.f) ? (int)v : -1;
    ok = ok && b >= 0 && (float)b == v;
    u[i] = (uint8_t)(b < 0 ? 0 : b);
  }
  return ok;
}

int match_host_desc(modsx_ctx *c, const float *desc1, int n1, const float *desc2, int n2, const double *pos2,
                    double ratioT, double contradDist, int nn, std::vector<modsx_tentative> &out) {
  out.clear();
  if (n1 == 0 || n2 == 0) return MODSX_OK;
  std::vector<uint8_t> u1((size_t)n1 * 128), u2((size_t)n2 * 128);
  if (!desc_f32_to_u8(desc1, u1.size(), u1.data()) || !desc_f32_to_u8(desc2, u2.size(), u2.data())) {
    set_error("modsx_match_fginn: descriptors must hold the integers 0..255 (SIFT-family quantisation)");
    return MODSX_ERR_ARG;
  }
  if (!c->descU8[0].ensure(u1.size()) || !c->descU8[1].ensure(u2.size())) return MODSX_ERR_NOMEM;
  MX_HIP(hipMemcpyAsync(c->descU8[0].p, u1.data(), u1.size(), hipMemcpyHostToDevice, c->stream));
  MX_HIP(hipMemcpyAsync(c->descU8[1].p, u2.data(), u2.size(), hipMemcpyHostToDevice, c->stream));
  MX_HIP(hipStreamSynchronize(c->stream));
  return match_device(c, (uint8_t *)c->descU8[0].p, n1, (uint8_t *)c->descU8[1].p, n2, pos2, ratioT, contradDist, nn, out);
}

// ------------------------------------------------------------------------------------------------
// one step of the mods.cpp loop for an identity view
// ------------------------------------------------------------------------------------------------
// DuplicateFiltering + LORANSACFiltering on the tentatives of one pair (mods.cpp:300-342, doBeforeRANSAC = 1).
// Fills the counters, H and the three malloc'd arrays of `res` (which must not own arrays yet).
void verify_tentatives(const std::vector<modsx_region> &r1, const std::vector<modsx_region> &r2,
                       const std::vector<modsx_tentative> &tents, const modsx_pair_params &pp, modsx_pair_result *res) {
  RegList a, b;
  a.add(r1); b.add(r2);
  verify_tentatives(a, b, tents, pp, res);
}
// kp1(i) / kp2(i): the seven doubles x, y, a11, a12, a21, a22, s of region i's reproj_kp -- all that DuplicateFiltering and
// LO-RANSAC read of a region
template <class K1, class K2>
static void verify_core(const K1 &kp1, const K2 &kp2, const std::vector<modsx_tentative> &tents, const modsx_pair_params &pp,
                        modsx_pair_result *res) {
  HostMark hm;
  res->n_tentatives = (int)tents.size();
  const int T0 = (int)tents.size();
  std::vector<double> pts((size_t)T0 * 4 + 4), key(T0 + 1);
  for (int i = 0; i < T0; i++) {
    const double *a = kp1(tents[i].q), *b = kp2(tents[i].t0);
    pts[4 * i] = a[0]; pts[4 * i + 1] = a[1]; pts[4 * i + 2] = b[0]; pts[4 * i + 3] = b[1];
    key[i] = tents[i].ratio;
  }
  hm.mark("verify: points");
  std::vector<int> order(T0 + 1);
  std::vector<unsigned char> keepd(T0 + 1);
  duplicate_filtering(pts.data(), key.data(), T0, pp.duplicateDist, 1, order.data(), keepd.data());
  hm.mark("verify: duplicate filter");
  std::vector<modsx_tentative> uniq;
  for (int i = 0; i < T0; i++) if (keepd[i]) uniq.push_back(tents[order[i]]);
  const int T = (int)uniq.size();
  res->n_unique = T;
  std::vector<double> p2((size_t)T * 4 + 4), l1((size_t)T * 5 + 5), l2((size_t)T * 5 + 5);
  for (int i = 0; i < T; i++) {
    const double *a = kp1(uniq[i].q), *b = kp2(uniq[i].t0);
    p2[4 * i] = a[0]; p2[4 * i + 1] = a[1]; p2[4 * i + 2] = b[0]; p2[4 * i + 3] = b[1];
    for (int k = 0; k < 5; k++) { l1[5 * i + k] = a[2 + k]; l2[5 * i + k] = b[2 + k]; }
  }
  res->tentatives = (modsx_tentative *)malloc(sizeof(modsx_tentative) * std::max(1, T));
  res->ransac_inlier = (unsigned char *)calloc(std::max(1, T), 1);
  res->verified = (unsigned char *)calloc(std::max(1, T), 1);
  for (int i = 0; i < T; i++) res->tentatives[i] = uniq[i];
  hm.mark("verify: unique lists");
  double Hraw[9];
  int dout[3] = {0, 0, 0};
  int nv;
  if (pp.useF)
    nv = loransac_f(p2.data(), l1.data(), l2.data(), T, pp.err_threshold, pp.confidence, pp.max_samples,
                    pp.localOptimization, pp.LAFCoef, pp.doSymmCheck, pp.errorType, pp.ransac_seed, res->H,
                    res->ransac_inlier, res->verified, dout);
  else
    nv = loransac_h(p2.data(), l1.data(), l2.data(), T, pp.err_threshold, pp.confidence, pp.max_samples,
                    pp.localOptimization, pp.HLAFCoef, pp.doSymmCheck, pp.ransac_seed, res->H, Hraw, res->ransac_inlier,
                    res->verified, dout, pp.errorType);
  hm.mark("verify: lo-ransac + checks");
  res->n_verified = nv < 0 ? 0 : nv;
  res->n_ransac_inliers = 0;
  for (int i = 0; i < T; i++) res->n_ransac_inliers += res->ransac_inlier[i];
  res->ransac_samples = dout[0]; res->ransac_lo = dout[1];
}

void verify_tentatives(const RegList &r1, const RegList &r2, const std::vector<modsx_tentative> &tents,
                       const modsx_pair_params &pp, modsx_pair_result *res) {
  static_assert(offsetof(modsx_keypoint, x) == 0 && offsetof(modsx_keypoint, s) == 48, "x, y, a11, a12, a21, a22, s are seven consecutive doubles");
  verify_core([&](size_t i) { return &r1[i].reproj_kp.x; }, [&](size_t i) { return &r2[i].reproj_kp.x; }, tents, pp, res);
}
// The same on geometry rows (kp[7 i ..]: what the view-sharded pair call moves instead of whole regions).  The list of a step with
// several descriptor classes repeats its n regions once per class (index = class * n + region), as the RegList of such a step does.
void verify_tentatives_kp(const double *kp1, size_t n1, const double *kp2, size_t n2, const std::vector<modsx_tentative> &tents,
                          const modsx_pair_params &pp, modsx_pair_result *res) {
  verify_core([&](size_t i) { return kp1 + 7 * (n1 ? i % n1 : 0); }, [&](size_t i) { return kp2 + 7 * (n2 ? i % n2 : 0); }, tents, pp, res);
}

// One step of mods.cpp's loop (identity view) for G <= MAXB / 2 independent pairs at once: the 2G images go through
// detection, orientation and description as ONE batch (one launch set, blockIdx.z / job tables select the image), then
// each pair is matched and verified.  Results are those of G separate calls.
int match_pair_group(modsx_ctx *c, const modsx_image *const *imgs1, const modsx_image *const *imgs2, int G,
                     const modsx_pair_params &pp, modsx_pair_result *res, std::vector<VerifyTask> *deferred) {
  CtxBusy busy(c);
  if (G < 1 || G > PAIR_GROUP || 2 * G > MAXB) { set_error("match_pair_group: group size"); return MODSX_ERR_ARG; }
  for (int g = 0; g < G; g++) {
    memset(&res[g], 0, sizeof res[g]);
    for (int i = 0; i < 9; i++) res[g].H[i] = -1;
  }
  const int n = 2 * G;
  const modsx_image *imgs[MAXB];
  for (int g = 0; g < G; g++) { imgs[2 * g] = imgs1[g]; imgs[2 * g + 1] = imgs2[g]; }
  struct SetScope { SetScope() { host_set_enter(); } ~SetScope() { host_set_leave(); } } setScope;   // a launch set in flight (host pool policy)
  const double t0 = now_ms();
  std::vector<modsx_keypoint> kps[MAXB];
  int rc = detect_keypoints_batch(c, imgs, n, pp.det, nullptr, nullptr, kps);
  if (rc) return rc;
  std::vector<modsx_region> regs[MAXB], oriented[MAXB];
  for (int i = 0; i < n; i++) {
    regs[i].resize(kps[i].size());
    detect_affine_regions(kps[i].data(), (int)kps[i].size(), 0, MODSX_DET_HESSIAN, regs[i].data());
  }
  const double t1 = now_ms();
  // the step's descriptor classes: ONE oriented list (Half-folded orientation histogram iff a Half type is among them,
  // imagerepresentation.cpp:693-706, 1259-1264, 1288-1296), one pass over the patches for all of them
  DescSet ds;
  rc = resolve_descs(pp, nullptr, ds);
  if (rc) return rc;
  rc = detect_orientation_batch(c, imgs, n, regs, pp.ori_mrSize, pp.ori_patchSize, ds.half() ? 1 : 0, pp.ori_maxAngles, pp.ori_threshold,
                                0, oriented);
  if (rc) return rc;
  const double eye[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
  for (int i = 0; i < n; i++) {
    int m = reproject_regions(oriented[i].data(), (int)oriented[i].size(), eye, imgs[i]->cols, imgs[i]->rows);
    oriented[i].resize(m);
  }
  const double t2 = now_ms();
  rc = describe_batch(c, imgs, n, oriented, pp.desc_mrSize, pp.desc_patchSize, 0, pp.desc_photoNorm, ds.type[0],
                      pp.desc_maxBinValue, nullptr, nullptr, nullptr, &ds);
  if (rc) return rc;
  const double t3 = now_ms();
  double tMatch = 0, tVerify = 0;
  // classes in the order GetCorresponcesVector("All", "All") walks them: descriptor NAME order (correspondencebank.cpp:117-179)
  int ord[MODSX_MAX_DESC];
  desc_class_order(ds, ord);
  auto desc_of = [&](int k, int img) -> const uint8_t * { return (const uint8_t *)(k == 0 ? c->descU8[img].p : c->descU8x[k - 1][img].p); };
  std::vector<double> pos2v[MAXB / 2];
  std::vector<modsx_tentative> tentsv[MAXB / 2];
  const double m0 = now_ms();
  for (int g = 0; g < G; g++) {
    const std::vector<modsx_region> &rb = oriented[2 * g + 1];
    res[g].n_regions1 = (int)oriented[2 * g].size() * ds.n;
    res[g].n_regions2 = (int)rb.size() * ds.n;
    pos2v[g].resize(rb.size() * 2 + 2);
    for (size_t i = 0; i < rb.size(); i++) { pos2v[g][2 * i] = rb[i].reproj_kp.x; pos2v[g][2 * i + 1] = rb[i].reproj_kp.y; }
  }
  // the G matching problems of a class share the matcher's launches; each class has its own FGINN threshold
  for (int oi = 0; oi < ds.n; oi++) {
    const int k = ord[oi];
    const uint8_t *pd1[MAXB / 2], *pd2[MAXB / 2];
    const double *ppos[MAXB / 2];
    int pn1[MAXB / 2], pn2[MAXB / 2];
    std::vector<modsx_tentative> part[MAXB / 2];
    for (int g = 0; g < G; g++) {
      pd1[g] = desc_of(k, 2 * g); pd2[g] = desc_of(k, 2 * g + 1);
      pn1[g] = (int)oriented[2 * g].size(); pn2[g] = (int)oriented[2 * g + 1].size(); ppos[g] = pos2v[g].data();
    }
    for (int g0 = 0; g0 < G; g0 += MATCH_MAXB) {
      const int nbm = std::min(MATCH_MAXB, G - g0);
      rc = match_device_batch(c, nbm, pd1 + g0, pn1 + g0, pd2 + g0, pn2 + g0, ppos + g0, ds.ratio[k], pp.contradDist, pp.nn,
                              part + g0, nullptr);
      if (rc) return rc;
    }
    for (int g = 0; g < G; g++) {
      const int o1 = oi * pn1[g], o2 = oi * pn2[g];
      if (oi == 0) { tentsv[g].swap(part[g]); continue; }
      for (modsx_tentative t : part[g]) {
        t.q += o1; t.t0 += o2;
        if (t.t1 >= 0) t.t1 += o2;
        if (t.tj >= 0) t.tj += o2;
        tentsv[g].push_back(t);
      }
    }
  }
  tMatch = now_ms() - m0;
  for (int g = 0; g < G; g++) {
    if (deferred) {   // modsx_match_pairs: verification is handed to the caller's helper threads
      deferred->emplace_back();
      VerifyTask &t = deferred->back();
      t.own.resize(2);
      t.own[0].swap(oriented[2 * g]); t.own[1].swap(oriented[2 * g + 1]);
      for (int oi = 0; oi < ds.n; oi++) { t.l1.add(t.own[0]); t.l2.add(t.own[1]); }
      t.tents.swap(tentsv[g]); t.res = &res[g]; t.dev = c->dev;
    } else {
      const double m1 = now_ms();
      RegList l1, l2;
      for (int oi = 0; oi < ds.n; oi++) { l1.add(oriented[2 * g]); l2.add(oriented[2 * g + 1]); }
      const int nr1 = res[g].n_regions1, nr2 = res[g].n_regions2;
      verify_tentatives(l1, l2, tentsv[g], pp, &res[g]);
      res[g].n_regions1 = nr1; res[g].n_regions2 = nr2;
      tVerify += now_ms() - m1;
    }
  }
  const double t5 = now_ms();
  prof_collect(c);
  // per-stage wall time of the group, divided by the number of pairs it carried
  c->timings[0] = (t1 - t0) / G; c->timings[1] = (t2 - t1) / G; c->timings[2] = (t3 - t2) / G; c->timings[3] = tMatch / G;
  c->timings[4] = tVerify / G; c->timings[5] = (t5 - t0) / G;
  return MODSX_OK;
}

int match_pair(modsx_ctx *c, const modsx_image *img1, const modsx_image *img2, const modsx_pair_params &pp,
               modsx_pair_result *res) {
  return match_pair_group(c, &img1, &img2, 1, pp, res);
}

}  // namespace mx
