// engine_shard.hip -- view-sharded multi-GPU path: one process per GPU, RCCL collectives on device buffers.
//
// Reference unit of parallelism: the synthesised views of an image are independent from GenerateSynthImageCorr through
// DescribeRegions (the `#pragma omp parallel for` over views, imagerepresentation.cpp:612-622) and meet only in the ordered
// concatenation of AddRegions (:2044-2045, ids re-based by AddRegionsToList :588-600).  Here view v belongs to rank
// v mod world; ONE collective per image side rebuilds the reference's list on every rank:
//   every rank packs a BLOCK = header {magic, rc, rows, views, per-view counts} + its rows (row = modsx_region, 200 B, + the
//   128 u8 descriptor bytes = 328 B) padded to the lane's agreed block size; one all-gather of the blocks on the context's
//   stream, device to device (xGMI between the GPUs of a node); a kernel reads the W headers on the device and writes
//   regions and descriptors in (view, detection) order.  The host learns the counts from the same download that brings the
//   regions back.  A block that turns out too small (a rank had more rows than the agreed size) is seen by every rank in
//   the headers: all of them grow the block size alike and repeat the exchange.
// Matching splits the QUERY rows (rank r takes [r n1 / W, (r+1) n1 / W) against all of image 2); the per-query result rows
// of the device matcher (32 B each, behind a header row) are all-gathered before the host turns them into tentatives, so
// every rank ends with the full list in query order and no host object crosses ranks.  DuplicateFiltering + LO-RANSAC run on
// the owner rank (or on every rank: same input, same seed, same result -- how the sharded ladder agrees on its early exit).
//
// What keeps W ranks from hanging each other:
//  * ONE communicator per rank.  The contexts (host thread + stream) of a rank are LANES of that communicator; collectives
//    are issued in strict round-robin lane order (lane 0's k-th, lane 1's k-th, ...), which is the same sequence on every
//    rank whatever the thread timing -- collectives of one communicator must be issued in one global order, and several
//    communicators per device can deadlock in the hardware queues.
//  * Error agreement.  A rank-local failure before a collective travels IN the collective (the `rc` of the header); every
//    rank returns the same error at the same point and none is left waiting.  Buffers a rank needs in order to take part
//    (the blocks themselves) grow only at points every rank derives from the gathered headers, followed by a 4-byte
//    all-gather of the allocation results.
//  * A watchdog.  Every wait on a collective has a deadline (MODSX_COMM_TIMEOUT_MS, default 30 s): past it the
//    communicator is aborted (ncclCommAbort), marked dead, and every call on it returns MODSX_ERR_TIMEOUT.
//
// Transports: RCCL, bound at run time (dlopen; an already loaded librccl -- e.g. the one torch ships -- is reused), so
// libmodsx has no link-time dependency on it and single-GPU users never load it; and LOOPBACK: W ranks inside one process
// on one device (one host thread per rank), the all-gather being W device-to-device copies between the ranks' buffers
// behind an event handshake.  Everything above the transport -- blocks, headers, device-side ordering, id re-basing, the
// query-row split, lanes, agreement, watchdog -- is the same code, which is how W = 2, 3, 8 are tested on a 1-GPU box.
#include <dlfcn.h>
#include <unistd.h>
#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <map>
#include <memory>
#include <mutex>
#include <thread>
#include <rccl/rccl.h>
#include "engine_api.hpp"

namespace mx {

using Clock = std::chrono::steady_clock;

struct RcclApi {
  void *h = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*CommAbort)(ncclComm_t) = nullptr;
  ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*GetVersion)(int *) = nullptr;
  const char *(*GetErrorString)(ncclResult_t) = nullptr;
  // the owner-only exchange (optional: a librccl without them still serves the all-gather paths)
  ncclResult_t (*Send)(const void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Recv)(void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
};
static RcclApi g_rccl;
static std::mutex g_rcclMu;
static bool rccl_load() {
  std::lock_guard<std::mutex> lk(g_rcclMu);
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
  a.CommAbort = (decltype(a.CommAbort))dlsym(h, "ncclCommAbort");
  a.AllGather = (decltype(a.AllGather))dlsym(h, "ncclAllGather");
  a.GetVersion = (decltype(a.GetVersion))dlsym(h, "ncclGetVersion");
  a.GetErrorString = (decltype(a.GetErrorString))dlsym(h, "ncclGetErrorString");
  a.Send = (decltype(a.Send))dlsym(h, "ncclSend");
  a.Recv = (decltype(a.Recv))dlsym(h, "ncclRecv");
  a.GroupStart = (decltype(a.GroupStart))dlsym(h, "ncclGroupStart");
  a.GroupEnd = (decltype(a.GroupEnd))dlsym(h, "ncclGroupEnd");
  if (!a.GetUniqueId || !a.CommInitRank || !a.CommDestroy || !a.CommAbort || !a.AllGather || !a.GetVersion || !a.GetErrorString) {
    set_error("librccl lacks an expected symbol");
    return false;
  }
  g_rccl = a;
  return true;
}

// ---- loopback transport: W ranks of one process on one device ---------------------------------------------------------
constexpr char LOOP_MAGIC[8] = {'M', 'X', 'L', 'O', 'O', 'P', '0', '1'};
// one point-to-point transfer of an exchange: `bytes` at `ptr` to / from rank `peer`.  The k-th send of rank a to rank b meets the
// k-th receive of rank b from rank a (the order ncclSend / ncclRecv match in inside a group).
struct Xfer { int peer; void *ptr; size_t bytes; };
struct LoopGroup {
  int world = 0;
  std::mutex mu;
  std::condition_variable cv;
  int arrived = 0, joined = 0;
  long gen = 0;
  bool dead = false;
  // the events of a rank's slot belong to the GROUP: peers wait on them after the last barrier of a collective, so they are
  // destroyed with the group (when the last rank has left), never by the rank that recorded them
  struct Slot { const void *send = nullptr; void *recv = nullptr; size_t bytes = 0; hipEvent_t ready = nullptr, done = nullptr; bool taken = false;
                const std::vector<Xfer> *sends = nullptr; };   // sends: the rank's send list of a point-to-point exchange
  std::vector<Slot> slots;
  int dev = 0;
  ~LoopGroup() {
    hipSetDevice(dev);
    for (Slot &sl : slots) { if (sl.ready) hipEventDestroy(sl.ready); if (sl.done) hipEventDestroy(sl.done); }
  }
  // all ranks of the group meet here; false = a rank did not show up within the deadline (the group is dead from then on)
  bool barrier(int timeout_ms) {
    std::unique_lock<std::mutex> lk(mu);
    if (dead) return false;
    const long g0 = gen;
    if (++arrived == world) { arrived = 0; gen++; cv.notify_all(); return true; }
    const auto deadline = Clock::now() + std::chrono::milliseconds(timeout_ms);
    while (gen == g0 && !dead)
      if (cv.wait_until(lk, deadline) == std::cv_status::timeout && gen == g0) { dead = true; cv.notify_all(); }
    return gen != g0 && !dead;
  }
  void kill() { std::lock_guard<std::mutex> lk(mu); dead = true; cv.notify_all(); }
};
// The point-to-point exchange of W in-process ranks: every rank publishes its send list, copies what its receives name out of the
// peers' send buffers (device to device, behind the peers' `ready` events) and lets go of its own send buffers once every peer
// has copied (`done`).  0 = ok, 1 = a rank did not show up, 2 = the lists of two ranks do not match, 3 = a HIP call failed.
static int loop_exchange(LoopGroup &g, int R, const std::vector<Xfer> &sends, const std::vector<Xfer> &recvs, hipStream_t s, int timeout_ms) {
  const int W = g.world;
  LoopGroup::Slot &me = g.slots[R];
  me.sends = &sends;
  if (hipEventRecord(me.ready, s) != hipSuccess) return 3;
  if (!g.barrier(timeout_ms)) return 1;
  int bad = 0;
  std::vector<size_t> nextFrom(W, 0);      // how far into peer p's send list the search for "to me" has come
  std::vector<char> waited(W, 0);
  for (const Xfer &rv : recvs) {
    if (rv.peer < 0 || rv.peer >= W) { bad = 2; break; }
    const std::vector<Xfer> &ps = *g.slots[rv.peer].sends;
    size_t &k = nextFrom[rv.peer];
    while (k < ps.size() && ps[k].peer != R) k++;
    if (k == ps.size() || ps[k].bytes != rv.bytes) { bad = 2; break; }
    if (!waited[rv.peer]) { if (hipStreamWaitEvent(s, g.slots[rv.peer].ready, 0) != hipSuccess) { bad = 3; break; } waited[rv.peer] = 1; }
    if (rv.bytes && hipMemcpyAsync(rv.ptr, ps[k].ptr, rv.bytes, hipMemcpyDeviceToDevice, s) != hipSuccess) { bad = 3; break; }
    k++;
  }
  if (!bad)
    for (int p = 0; p < W && !bad; p++) {       // every send to me must have met a receive
      const std::vector<Xfer> &ps = *g.slots[p].sends;
      for (size_t k = nextFrom[p]; k < ps.size(); k++) if (ps[k].peer == R) { bad = 2; break; }
    }
  if (hipEventRecord(me.done, s) != hipSuccess) bad = bad ? bad : 3;
  if (!g.barrier(timeout_ms)) return 1;      // the send lists stay valid until every rank has read them
  if (bad) return bad;
  for (int p = 0; p < W; p++) if (hipStreamWaitEvent(s, g.slots[p].done, 0) != hipSuccess) return 3;
  return 0;
}
static std::mutex g_loopMu;
static std::map<uint64_t, std::shared_ptr<LoopGroup>> g_loopGroups;
static uint64_t g_loopNext = 1;

// ---- a stand-in for librccl inside this process (tests only: modsx_debug_mock_rccl) -----------------------------------------------
// RCCL with more than one rank cannot run on a one-GPU box.  The mock fills the SAME function table the real library fills, so
// W in-process ranks go through modsx_comm_create's RCCL branch and transport_all_gather's RCCL branch (issueMu, the abort
// rules, the per-communicator issue order) -- everything but librccl itself.  Its all-gather is the loopback algorithm: W
// device-to-device copies ordered by events, two host barriers.
struct MockComm { std::shared_ptr<LoopGroup> g; int rank; };
static std::mutex g_mockMu;
static std::map<uint64_t, std::shared_ptr<LoopGroup>> g_mockGroups;
static uint64_t g_mockNext = 1;
static RcclApi g_rcclSaved;
static bool g_mockOn = false;
static ncclResult_t mock_GetUniqueId(ncclUniqueId *id) {
  memset(id, 0, sizeof *id);
  std::lock_guard<std::mutex> lk(g_mockMu);
  const uint64_t key = g_mockNext++;
  memcpy(id->internal, "MXMOCK01", 8);
  memcpy(id->internal + 8, &key, 8);
  return ncclSuccess;
}
static ncclResult_t mock_CommInitRank(ncclComm_t *out, int world, ncclUniqueId id, int rank) {
  if (memcmp(id.internal, "MXMOCK01", 8)) return ncclInvalidArgument;
  uint64_t key;
  memcpy(&key, id.internal + 8, 8);
  std::lock_guard<std::mutex> lk(g_mockMu);
  std::shared_ptr<LoopGroup> &g = g_mockGroups[key];
  if (!g) { g = std::make_shared<LoopGroup>(); g->world = world; g->slots.resize(world); hipGetDevice(&g->dev); }
  if (g->world != world || rank < 0 || rank >= world || g->slots[rank].taken) return ncclInvalidArgument;
  LoopGroup::Slot &sl = g->slots[rank];
  if (hipEventCreateWithFlags(&sl.ready, hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&sl.done, hipEventDisableTiming) != hipSuccess)
    return ncclUnhandledCudaError;
  sl.taken = true;
  MockComm *mc = new MockComm{g, rank};
  if (++g->joined == world) g_mockGroups.erase(key);
  *out = reinterpret_cast<ncclComm_t>(mc);
  return ncclSuccess;
}
static ncclResult_t mock_CommDestroy(ncclComm_t c) { delete reinterpret_cast<MockComm *>(c); return ncclSuccess; }
static ncclResult_t mock_CommAbort(ncclComm_t c) { reinterpret_cast<MockComm *>(c)->g->kill(); return ncclSuccess; }
static ncclResult_t mock_GetVersion(int *v) { *v = 0; return ncclSuccess; }
static const char *mock_GetErrorString(ncclResult_t) { return "mock rccl error"; }
static ncclResult_t mock_AllGather(const void *send, void *recv, size_t bytes, ncclDataType_t, ncclComm_t c, hipStream_t s) {
  MockComm *mc = reinterpret_cast<MockComm *>(c);
  LoopGroup &g = *mc->g;
  const int W = g.world;
  LoopGroup::Slot &me = g.slots[mc->rank];
  me.send = send; me.recv = recv; me.bytes = bytes;
  if (hipEventRecord(me.ready, s) != hipSuccess) return ncclUnhandledCudaError;
  if (!g.barrier(20000)) return ncclSystemError;
  for (int p = 0; p < W; p++) if (g.slots[p].bytes != bytes) return ncclInvalidArgument;
  for (int p = 0; p < W; p++) {
    if (hipStreamWaitEvent(s, g.slots[p].ready, 0) != hipSuccess) return ncclUnhandledCudaError;
    if (bytes && hipMemcpyAsync((char *)recv + (size_t)p * bytes, g.slots[p].send, bytes, hipMemcpyDeviceToDevice, s) != hipSuccess) return ncclUnhandledCudaError;
  }
  if (hipEventRecord(me.done, s) != hipSuccess) return ncclUnhandledCudaError;
  if (!g.barrier(20000)) return ncclSystemError;
  for (int p = 0; p < W; p++) if (hipStreamWaitEvent(s, g.slots[p].done, 0) != hipSuccess) return ncclUnhandledCudaError;
  return ncclSuccess;
}
// ncclGroupStart .. ncclGroupEnd of the stand-in: the sends and receives of the calling thread are collected and run as ONE
// loop_exchange at the group's end (a rank is one host thread here, as it is one process with the real library)
struct MockPending { MockComm *mc = nullptr; hipStream_t s = nullptr; std::vector<Xfer> sends, recvs; int depth = 0; bool bad = false; };
static thread_local MockPending t_mockPending;
static ncclResult_t mock_GroupStart() { t_mockPending.depth++; return ncclSuccess; }
static ncclResult_t mock_p2p(bool send, void *ptr, size_t bytes, int peer, ncclComm_t c, hipStream_t s) {
  MockPending &P = t_mockPending;
  MockComm *mc = reinterpret_cast<MockComm *>(c);
  if (P.depth < 1) return ncclInvalidUsage;          // the library only issues them inside a group
  if (P.mc && (P.mc != mc || P.s != s)) P.bad = true;
  P.mc = mc; P.s = s;
  (send ? P.sends : P.recvs).push_back(Xfer{peer, ptr, bytes});
  return ncclSuccess;
}
static ncclResult_t mock_Send(const void *p, size_t n, ncclDataType_t, int peer, ncclComm_t c, hipStream_t s) { return mock_p2p(true, const_cast<void *>(p), n, peer, c, s); }
static ncclResult_t mock_Recv(void *p, size_t n, ncclDataType_t, int peer, ncclComm_t c, hipStream_t s) { return mock_p2p(false, p, n, peer, c, s); }
static ncclResult_t mock_GroupEnd() {
  MockPending &P = t_mockPending;
  if (P.depth < 1) return ncclInvalidUsage;
  if (--P.depth) return ncclSuccess;
  ncclResult_t r = ncclSuccess;
  if (P.bad) r = ncclInvalidUsage;
  else if (P.mc) {
    const int e = loop_exchange(*P.mc->g, P.mc->rank, P.sends, P.recvs, P.s, 20000);
    r = e == 0 ? ncclSuccess : e == 1 ? ncclSystemError : e == 2 ? ncclInvalidArgument : ncclUnhandledCudaError;
  }
  P = MockPending();
  return r;
}

struct ShardLane {
  DevBuf blkLocal, blkAll, regsIn, regsOut, mLocal, mAll, rcDev, order, posOut, ownSend, ownRecv, ownJobs;
  PinBuf hRegs, hHdr, hRc, hJobs;
  size_t capBlock = 0;    // agreed: bytes of one region block the lane's buffers hold (blkLocal; blkAll = world x)
  size_t capMatch = 0;    // agreed: bytes of one match block (mLocal; mAll = world x)
  size_t capOwnSend = 0, capOwnRecv = 0, capOwnList = 0;   // agreed (owner-only exchange): rows the send / receive buffers and the list buffers hold
  int guessRows = 0;      // agreed: rows per block of the next exchange (a function of the headers gathered so far)
  size_t lastN = 0;       // local: list length of the last exchange (sizes the speculative region download)
  size_t localCap = 0;    // local: region capacity the rank's own descriptor scratch needed last time (a too small first guess runs the views twice)
  bool retired = false;
};

}  // namespace mx

struct modsx_comm {
  // transport
  ncclComm_t nccl = nullptr;
  std::shared_ptr<mx::LoopGroup> loop;
  int rank = 0, world = 1, version = 0, dev = 0;
  std::atomic<int> timeout_ms{30000};       // deadline of a collective (the watchdog)
  std::atomic<int> turn_timeout_ms{300000}; // deadline of a lane waiting for ITS TURN: the lane ahead may be in a long host stage
                                            // (a full-size MSER step, a 10^6-sample RANSAC), which is not a hung collective
  // lanes: the contexts of this rank, collectives issued round-robin
  std::mutex mu;
  std::condition_variable cv;
  std::vector<mx::ShardLane> lanes;
  int turn = 0;
  std::atomic<bool> dead{false};            // read without the lock by every lane and by the waits
  std::string deadWhy;                      // written once, under mu, before dead is set
  std::mutex issueMu;                       // serialises enqueueing on the RCCL communicator with its abort
  bool aborted = false;                     // under issueMu: ncclCommAbort has run; nccl is never used again
  long bytes_gathered = 0, collectives = 0, retries = 0, agreements = 0;
  long bytes_received = 0, exchanges = 0;   // the owner-only exchange: bytes this rank received in point-to-point transfers, calls
  std::atomic<int> exchange_mode{MODSX_EXCHANGE_ALL_GATHER};   // modsx_comm_set_exchange: the same on every rank
  std::atomic<long> turn_wait_us{0};        // time the lanes spent waiting for their turn (modsx_comm_stats)
};

namespace mx {

constexpr int REG_B = (int)sizeof(modsx_region);
static_assert(sizeof(modsx_region) == 200, "region rows are 200 + 128 bytes per descriptor class on the wire");
// What of a region travels in a row: the whole modsx_region (the calls that return region lists), or the seven geometry doubles
// of its reproj_kp -- x, y, a11, a12, a21, a22, s: the matcher's positions and everything DuplicateFiltering / LO-RANSAC read
// (verify_tentatives) -- for the calls that return pair results only.  regLen is a multiple of 8.
struct RowFmt { int regOff, regLen, posOfs; };
constexpr int KP_B = 7 * (int)sizeof(double);
static_assert(offsetof(modsx_keypoint, s) == 6 * sizeof(double), "x, y, a11, a12, a21, a22, s lead modsx_keypoint");
static RowFmt row_fmt(int format) {
  if (format == MODSX_SHARD_ROW_KP) return RowFmt{(int)offsetof(modsx_region, reproj_kp), KP_B, 0};
  return RowFmt{0, REG_B, (int)offsetof(modsx_region, reproj_kp)};
}
static inline __host__ __device__ int row_bytes(int regLen, int nd) { return regLen + 128 * nd; }   // nd = descriptor classes of the step
struct DescPtrs { unsigned char *p[MODSX_MAX_DESC]; };
constexpr int HDR_MAGIC = 0x4D585348;   // "MXSH"
constexpr int HDR_MAGIC_OWNER = 0x4D58534F;   // "MXSO": a header of the owner-only exchange (no rows behind it)
constexpr int HDR_FIXED = 4;            // ints before the per-view counts: magic, rc, rows, views
static int hdr_bytes(int nv) { return ((HDR_FIXED + nv) * 4 + 63) & ~63; }

static void comm_kill(modsx_comm *cm, const std::string &why) {
  {
    std::lock_guard<std::mutex> lk(cm->mu);
    if (cm->dead.load()) return;
    cm->deadWhy = why;
    cm->dead.store(true);
    cm->cv.notify_all();
  }
  if (cm->loop) cm->loop->kill();
  else {
    // abort once, and never while another lane is inside ncclAllGather on this communicator (issueMu); the handle stays set
    // (modsx_comm_destroy frees what is left) but nothing enqueues on it again: transport_all_gather checks `aborted` under
    // the same lock
    std::lock_guard<std::mutex> lk(cm->issueMu);
    if (!cm->aborted && cm->nccl) { g_rccl.CommAbort(cm->nccl); cm->aborted = true; }
  }
}
static int comm_dead_rc(modsx_comm *cm) {
  std::string why;
  { std::lock_guard<std::mutex> lk(cm->mu); why = cm->deadWhy; }
  set_error("communicator is dead (" + why + "): a collective timed out or failed; destroy it and create a new one");
  return MODSX_ERR_TIMEOUT;
}

// round-robin issue order over the lanes that are still active
// MODSX_SHARD_FREE_ORDER=1 (a measurement aid, honoured at world 1 only, where no peer has to see the same order): lanes issue as
// they arrive -- the difference to the default run is what the fixed order costs
static bool free_order(const modsx_comm *cm) {
  static const bool on = getenv("MODSX_SHARD_FREE_ORDER") && atoi(getenv("MODSX_SHARD_FREE_ORDER")) != 0;
  return on && cm->world == 1;
}
static int turn_begin(modsx_comm *cm, int lane) {
  if (free_order(cm)) return cm->dead.load() ? comm_dead_rc(cm) : MODSX_OK;
  std::unique_lock<std::mutex> lk(cm->mu);
  const int tmo = std::max(cm->turn_timeout_ms.load(), cm->timeout_ms.load());
  const auto t0 = Clock::now();
  const auto deadline = t0 + std::chrono::milliseconds(tmo);
  struct Acc { modsx_comm *cm; Clock::time_point t0; ~Acc() { cm->turn_wait_us += std::chrono::duration_cast<std::chrono::microseconds>(Clock::now() - t0).count(); } } acc{cm, t0};
  while (!cm->dead.load() && cm->turn != lane) {
    if (cm->cv.wait_until(lk, deadline) == std::cv_status::timeout && cm->turn != lane && !cm->dead.load()) {
      const int ahead = cm->turn;
      lk.unlock();
      comm_kill(cm, "lane " + std::to_string(lane) + " waited " + std::to_string(tmo) + " ms for its turn (lane " +
                        std::to_string(ahead) + " never issued its collective)");
      return comm_dead_rc(cm);
    }
  }
  if (cm->dead.load()) { lk.unlock(); return comm_dead_rc(cm); }
  return MODSX_OK;
}
static void turn_advance_locked(modsx_comm *cm) {
  const int L = (int)cm->lanes.size();
  for (int k = 1; k <= L; k++) {
    const int t = (cm->turn + k) % L;
    if (!cm->lanes[t].retired) { cm->turn = t; break; }
  }
  cm->cv.notify_all();
}
static void turn_end(modsx_comm *cm) {
  if (free_order(cm)) return;
  std::lock_guard<std::mutex> lk(cm->mu);
  turn_advance_locked(cm);
}

// wait for the stream with the watchdog's deadline
static int comm_wait(modsx_comm *cm, hipStream_t s) {
  const auto t0 = Clock::now();
  const int tmo = cm->timeout_ms.load();
  const auto deadline = t0 + std::chrono::milliseconds(tmo);
  for (;;) {
    hipError_t e = hipStreamQuery(s);
    if (e == hipSuccess) return MODSX_OK;
    if (e != hipErrorNotReady) { set_error(std::string("sharded path: ") + hipGetErrorString(e)); comm_kill(cm, "device error"); return MODSX_ERR_DEVICE; }
    if (cm->dead.load()) return comm_dead_rc(cm);
    const auto now = Clock::now();
    if (now > deadline) {
      comm_kill(cm, "a collective did not complete within " + std::to_string(tmo) + " ms (a peer rank is missing)");
      return comm_dead_rc(cm);
    }
    if (now - t0 > std::chrono::microseconds(200)) usleep(50); else std::this_thread::yield();
  }
}

// the transport: enqueue an all-gather of `bytes` per rank on stream s (this rank's turn is held by the caller)
static int transport_all_gather(modsx_comm *cm, const void *send, void *recv, size_t bytes, hipStream_t s) {
  if (cm->dead.load()) return comm_dead_rc(cm);
  if (!cm->loop) {     // the transport is chosen by what the communicator was created as, never by a handle that may change
    ncclResult_t r;
    {
      std::lock_guard<std::mutex> lk(cm->issueMu);
      if (cm->aborted || !cm->nccl) return comm_dead_rc(cm);
      r = g_rccl.AllGather(send, recv, bytes, ncclUint8, cm->nccl, s);
    }
    if (r != ncclSuccess) {
      set_error(std::string("ncclAllGather: ") + g_rccl.GetErrorString(r));
      comm_kill(cm, "ncclAllGather failed");
      return MODSX_ERR_DEVICE;
    }
  } else {
    LoopGroup &g = *cm->loop;
    const int W = cm->world, R = cm->rank;
    LoopGroup::Slot &me = g.slots[R];
    me.send = send; me.recv = recv; me.bytes = bytes;
    MX_HIP(hipEventRecord(me.ready, s));
    if (!g.barrier(cm->timeout_ms.load())) { comm_kill(cm, "loopback: a rank did not reach the collective"); return comm_dead_rc(cm); }
    for (int p = 0; p < W; p++)
      if (g.slots[p].bytes != bytes) { comm_kill(cm, "loopback: ranks disagree on the size of a collective"); return comm_dead_rc(cm); }
    for (int p = 0; p < W; p++) {
      MX_HIP(hipStreamWaitEvent(s, g.slots[p].ready, 0));
      if (bytes) MX_HIP(hipMemcpyAsync((char *)recv + (size_t)p * bytes, g.slots[p].send, bytes, hipMemcpyDeviceToDevice, s));
    }
    MX_HIP(hipEventRecord(me.done, s));
    if (!g.barrier(cm->timeout_ms.load())) { comm_kill(cm, "loopback: a rank left the collective"); return comm_dead_rc(cm); }
    for (int p = 0; p < W; p++) MX_HIP(hipStreamWaitEvent(s, g.slots[p].done, 0));   // my send buffer is free once every peer copied it
  }
  cm->bytes_gathered += (long)bytes * cm->world;
  cm->collectives++;
  return MODSX_OK;
}
// the transport of the owner-only exchange: this rank's sends and receives as ONE group on stream s (the lane's turn is held by the
// caller).  Every rank derives its lists from the same gathered headers, so the k-th send of a to b has the size of the k-th receive
// of b from a.
static int transport_exchange(modsx_comm *cm, const std::vector<Xfer> &sends, const std::vector<Xfer> &recvs, hipStream_t s) {
  if (cm->dead.load()) return comm_dead_rc(cm);
  if (!cm->loop) {
    ncclResult_t r = ncclSuccess;
    {
      std::lock_guard<std::mutex> lk(cm->issueMu);
      if (cm->aborted || !cm->nccl) return comm_dead_rc(cm);
      if (!g_rccl.Send || !g_rccl.Recv || !g_rccl.GroupStart || !g_rccl.GroupEnd) { set_error("librccl lacks ncclSend / ncclRecv / ncclGroupStart / ncclGroupEnd"); return MODSX_ERR_DEVICE; }
      r = g_rccl.GroupStart();
      for (size_t k = 0; k < recvs.size() && r == ncclSuccess; k++) r = g_rccl.Recv(recvs[k].ptr, recvs[k].bytes, ncclUint8, recvs[k].peer, cm->nccl, s);
      for (size_t k = 0; k < sends.size() && r == ncclSuccess; k++) r = g_rccl.Send(sends[k].ptr, sends[k].bytes, ncclUint8, sends[k].peer, cm->nccl, s);
      const ncclResult_t e = g_rccl.GroupEnd();      // always closed, also after a failed call inside
      if (r == ncclSuccess) r = e;
    }
    if (r != ncclSuccess) {
      set_error(std::string("ncclSend / ncclRecv: ") + g_rccl.GetErrorString(r));
      comm_kill(cm, "the owner-only exchange failed");
      return MODSX_ERR_DEVICE;
    }
  } else {
    const int e = loop_exchange(*cm->loop, cm->rank, sends, recvs, s, cm->timeout_ms.load());
    if (e) {
      comm_kill(cm, e == 1 ? "loopback: a rank did not reach the exchange" : e == 2 ? "loopback: the ranks' send and receive lists do not match" : "loopback: a device call failed");
      return comm_dead_rc(cm);
    }
  }
  for (const Xfer &x : recvs) cm->bytes_received += (long)x.bytes;
  cm->exchanges++;
  return MODSX_OK;
}
static int ordered_all_gather(modsx_comm *cm, int lane, const void *send, void *recv, size_t bytes, hipStream_t s) {
  int rc = turn_begin(cm, lane);
  if (rc) return rc;
  rc = transport_all_gather(cm, send, recv, bytes, s);
  turn_end(cm);
  return rc;
}

// every rank contributes the result of a local step (an allocation); all of them get the first failure in rank order
static int comm_agree(modsx_comm *cm, int lane, hipStream_t s, int local_rc) {
  ShardLane &L = cm->lanes[lane];
  int *h = (int *)L.hRc.p, *d = (int *)L.rcDev.p;
  h[0] = local_rc;
  MX_HIP(hipMemcpyAsync(d, h, 4, hipMemcpyHostToDevice, s));
  int rc = ordered_all_gather(cm, lane, d, d + 16, 4, s);
  if (rc) return rc;
  MX_HIP(hipMemcpyAsync(h + 16, d + 16, (size_t)cm->world * 4, hipMemcpyDeviceToHost, s));
  rc = comm_wait(cm, s);
  if (rc) return rc;
  cm->agreements++;
  for (int r = 0; r < cm->world; r++)
    if (h[16 + r]) {
      if (r != cm->rank || !local_rc) set_error("rank " + std::to_string(r) + " of the sharded call failed (" + std::to_string(h[16 + r]) + ")");
      return h[16 + r];
    }
  return MODSX_OK;
}

// every rank contributes a value it must share with the others (the verified count that decides a ladder's early exit):
// MODSX_OK when all ranks hold the same one, MODSX_ERR_INTERNAL -- on EVERY rank, from the same gathered data -- otherwise
int comm_same_value(modsx_ctx *c, modsx_comm *cm, int value, const char *what) {
  const int lane = c->shardLane >= 0 && c->shardLane < (int)cm->lanes.size() ? c->shardLane : 0;
  ShardLane &L = cm->lanes[lane];
  hipStream_t s = c->stream;
  int *h = (int *)L.hRc.p, *d = (int *)L.rcDev.p;
  h[0] = value;
  MX_HIP(hipMemcpyAsync(d, h, 4, hipMemcpyHostToDevice, s));
  int rc = ordered_all_gather(cm, lane, d, d + 16, 4, s);
  if (rc) return rc;
  MX_HIP(hipMemcpyAsync(h + 16, d + 16, (size_t)cm->world * 4, hipMemcpyDeviceToHost, s));
  rc = comm_wait(cm, s);
  if (rc) return rc;
  cm->agreements++;
  for (int r = 0; r < cm->world; r++)
    if (h[16 + r] != h[16]) {
      set_error(std::string("ranks disagree on ") + what + " (rank 0: " + std::to_string(h[16]) + ", rank " + std::to_string(r) + ": " +
                std::to_string(h[16 + r]) + "): the hosts of the ranks do not compute the verification alike");
      return MODSX_ERR_INTERNAL;
    }
  return MODSX_OK;
}

// block = header + rows; rows[i] = region record i (regLen bytes of `regs`, 8-byte words) followed by its 128 descriptor bytes
__global__ void k_pack_rows(const unsigned char *regs, int regLen, DescPtrs desc, int nd, int n, unsigned char *rows) {
  const int i = blockIdx.x * 8 + (threadIdx.x >> 5), l = threadIdx.x & 31;
  if (i >= n) return;
  unsigned char *dst = rows + (size_t)i * row_bytes(regLen, nd);
  if (l < regLen / 8) reinterpret_cast<uint64_t *>(dst)[l] = reinterpret_cast<const uint64_t *>(regs + (size_t)i * regLen)[l];
  for (int k = 0; k < nd; k++)
    if (l < 16) reinterpret_cast<uint64_t *>(dst + regLen + 128 * k)[l] = reinterpret_cast<const uint64_t *>(desc.p[k] + (size_t)i * 128)[l];
}

// The reference's order from the gathered blocks, on the device: view v is rank v mod W's, at that rank's running offset;
// its rows go to the list position the views before it fill (AddRegions' concatenation).  One 32-lane group per row slot
// (rank r, row i < G); every workgroup rebuilds the two small prefix tables from the W headers in LDS.
constexpr int SHARD_MAXV = 1024, SHARD_MAXW = 64;
__global__ __launch_bounds__(256) void k_unpack_blocks(const unsigned char *all, int W, int nv, int G, size_t blockB, int hdrB,
                                                       unsigned char *regs, int regLen, DescPtrs desc, int nd, size_t cap, double *pos,
                                                       int posOfs) {
  __shared__ int viewStart[SHARD_MAXV], runStart[SHARD_MAXV], cnt[SHARD_MAXV], run[SHARD_MAXW];
  for (int v = threadIdx.x; v < nv; v += 256) {
    const int *h = reinterpret_cast<const int *>(all + (size_t)(v % W) * blockB);
    cnt[v] = h[HDR_FIXED + v];
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int r = 0; r < W; r++) run[r] = 0;
    int tot = 0;
    for (int v = 0; v < nv; v++) { const int r = v % W; viewStart[v] = tot; runStart[v] = run[r]; run[r] += cnt[v]; tot += cnt[v]; }
  }
  __syncthreads();
  const long slot = (long)blockIdx.x * 8 + (threadIdx.x >> 5);
  const int l = threadIdx.x & 31;
  const int r = (int)(slot / G), i = (int)(slot % G);
  if (r >= W || i >= min(run[r], G)) return;
  int v = r;
  while (v + W < nv && runStart[v + W] <= i) v += W;     // views of rank r in ascending order; an empty one shares its start with the next
  const size_t j = (size_t)viewStart[v] + (i - runStart[v]);
  if (j >= cap) return;
  const unsigned char *s = all + (size_t)r * blockB + hdrB + (size_t)i * row_bytes(regLen, nd);
  if (l < regLen / 8) reinterpret_cast<uint64_t *>(regs + j * regLen)[l] = reinterpret_cast<const uint64_t *>(s)[l];
  // reproj_kp.x, .y of every region as the matcher's pos2 array: ranks that do not verify a pair never bring its regions to the host
  if (pos && l < 2) pos[2 * j + l] = reinterpret_cast<const double *>(s + posOfs)[l];
  for (int k = 0; k < nd; k++)
    if (l < 16) reinterpret_cast<uint64_t *>(desc.p[k] + j * 128)[l] = reinterpret_cast<const uint64_t *>(s + regLen + 128 * k)[l];
}

// Position of every row of the reference's list inside the all-gathered buffer (the host statement of what
// k_unpack_blocks computes; tests compare the two): counts[r * nviews + v].  Returns the list length.
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

// ---- the owner-only exchange ---------------------------------------------------------------------------------------------------
// modsx_match_pairs_views_sharded verifies pair g on ONE rank, and only that rank reads the pair's rows: with
// modsx_comm_set_exchange(MODSX_EXCHANGE_OWNER) the rows of image j travel to owner[j] alone (ncclSend / ncclRecv inside one
// group) instead of to every rank -- 1 / world of the all-gather's volume on the wire, the same bytes into the owner.  The
// per-item counts still go to every rank (an all-gather of the headers, a few KB): from them every rank derives, alike,
//   * the messages: rank r sends the rows of its items of image j (consecutive in its local order) to owner[j], images ascending;
//     the owner's receive buffer holds them image by image, source ranks ascending;
//   * the unpack jobs of an owner: item f of an owned image = rows of the receive buffer -> rows [start[f], start[f] + count[f]) of
//     the list, where start[] counts ALL items -- a rank's lists keep the global positions, the slices of images it does not own
//     stay unwritten and unread;
//   * the capacities every rank's buffers need (the maxima over the ranks), so that growth happens at agreed points.
struct OwnerMsg { int peer, image, row0, rows; };   // row0: first LOCAL row (send) / first row of the receive buffer (recv)
struct OwnerJob { int src0, dst0, n, pad; };        // n rows from row src0 of the receive buffer to row dst0 of the list
struct OwnerPlan {
  std::vector<OwnerMsg> sends, recvs;
  std::vector<OwnerJob> jobs;
  size_t recvRows = 0, sendRows = 0, maxRecvRows = 0, maxSendRows = 0, N = 0;
};
static void owner_plan(const int *cnt, int nimg, int nviews, int W, int R, const int *owner, OwnerPlan &P) {
  P = OwnerPlan();
  const int nv = nimg * nviews;
  std::vector<size_t> start(nv + 1, 0), perRank(W, 0), perOwner(W, 0);
  for (int f = 0; f < nv; f++) { start[f + 1] = start[f] + (size_t)cnt[f]; perRank[f % W] += (size_t)cnt[f]; perOwner[owner[f / nviews]] += (size_t)cnt[f]; }
  P.N = start[nv];
  for (int r = 0; r < W; r++) { P.maxSendRows = std::max(P.maxSendRows, perRank[r]); P.maxRecvRows = std::max(P.maxRecvRows, perOwner[r]); }
  P.sendRows = perRank[R];
  size_t localRow = 0;
  std::vector<size_t> rowsOf(W);
  for (int j = 0; j < nimg; j++) {
    std::fill(rowsOf.begin(), rowsOf.end(), 0);
    for (int v = 0; v < nviews; v++) { const int f = j * nviews + v; rowsOf[f % W] += (size_t)cnt[f]; }
    if (rowsOf[R]) P.sends.push_back(OwnerMsg{owner[j], j, (int)localRow, (int)rowsOf[R]});
    localRow += rowsOf[R];
    if (owner[j] != R) continue;
    std::vector<size_t> seg(W);
    for (int r = 0; r < W; r++) {
      seg[r] = P.recvRows;
      if (rowsOf[r]) P.recvs.push_back(OwnerMsg{r, j, (int)P.recvRows, (int)rowsOf[r]});
      P.recvRows += rowsOf[r];
    }
    for (int v = 0; v < nviews; v++) {
      const int f = j * nviews + v, r = f % W;
      if (cnt[f]) P.jobs.push_back(OwnerJob{(int)seg[r], (int)start[f], cnt[f], 0});
      seg[r] += (size_t)cnt[f];
    }
  }
}
// rows of the receive buffer to their list positions: blockIdx.y = job, one 32-lane group per row
__global__ __launch_bounds__(256) void k_unpack_jobs(const OwnerJob *jobs, const unsigned char *recv, unsigned char *regs, int regLen, DescPtrs desc,
                                                     int nd, double *pos, int posOfs) {
  const OwnerJob jb = jobs[blockIdx.y];
  const int i = blockIdx.x * 8 + (threadIdx.x >> 5), l = threadIdx.x & 31;
  if (i >= jb.n) return;
  const unsigned char *s = recv + (size_t)(jb.src0 + i) * row_bytes(regLen, nd);
  const size_t j = (size_t)jb.dst0 + i;
  if (l < regLen / 8) reinterpret_cast<uint64_t *>(regs + j * regLen)[l] = reinterpret_cast<const uint64_t *>(s)[l];
  if (pos && l < 2) pos[2 * j + l] = reinterpret_cast<const double *>(s + posOfs)[l];
  for (int k = 0; k < nd; k++)
    if (l < 16) reinterpret_cast<uint64_t *>(desc.p[k] + j * 128)[l] = reinterpret_cast<const uint64_t *>(s + regLen + 128 * k)[l];
}

int comm_rank(const modsx_comm *cm) { return cm->rank; }
int comm_world(const modsx_comm *cm) { return cm->world; }

static int lane_of(modsx_ctx *c, modsx_comm *cm) {
  const int lane = c->shardLane;
  return lane >= 0 && lane < (int)cm->lanes.size() ? lane : 0;
}

// grow a device buffer keeping its first `keep` bytes (the accumulated descriptors of earlier ladder steps)
static int grow_keep(hipStream_t s, DevBuf &b, size_t keep, size_t bytes) {
  if (b.cap >= bytes) return MODSX_OK;
  DevBuf bigger;
  if (!bigger.ensure(bytes)) return MODSX_ERR_NOMEM;
  if (keep && b.p) {
    MX_HIP(hipMemcpyAsync(bigger.p, b.p, keep, hipMemcpyDeviceToDevice, s));
    MX_HIP(hipStreamSynchronize(s));
  }
  b.release();
  b = bigger;
  return MODSX_OK;
}

// The sharded SynthDetectDescribeKeypoints: the regions of ALL views of this step in reference order on every rank (ids local
// to their view block: the caller re-bases them onto its lists, AddRegionsToList), the u8 descriptors of the step's
// descriptor class k appended at row base[k] of *descAcc[k].
int detect_describe_views_sharded(modsx_ctx *c, modsx_comm *cm, const modsx_image *img, const modsx_view *views, int nv,
                                  const modsx_pair_params &pp, const DescSet &ds, std::vector<modsx_region> &regs,
                                  DevBuf *const *descAcc, const size_t *base, int *viewCounts) {
  return detect_describe_items_sharded(c, cm, &img, 1, views, nv, pp, ds, regs, descAcc, base, viewCounts, nullptr, nullptr, nullptr, nullptr);
}
// The same for the views of SEVERAL images in one exchange (the batched pair call: 2 k images of k pairs).  The (image, view)
// items are numbered f = image * nviews + view and item f belongs to rank f mod world; a rank runs all its items as one
// launch set sequence (detect_describe_items) and ONE all-gather moves every image side.  regs: all images in item order
// (image, view, detection); itemCounts[f] = regions of item f.
// wantImg (optional, nimg flags): only the regions of the flagged images come to the host -- `regs` is then the concatenation of
// THOSE images' slices and regStart[j] .. regStart[j + 1] the slice of image j in it (empty for the others); without it `regs`
// holds all images.  devPos (optional): receives the device array of (reproj x, y) of ALL regions in item order (the matcher's
// pos2), valid until the lane's next exchange.
// kpRows (optional): the rows carry the verification slice only (MODSX_SHARD_ROW_KP); `regs` stays empty and *kpRows receives what
// `regs` would hold, seven doubles per region.
int detect_describe_items_sharded(modsx_ctx *c, modsx_comm *cm, const modsx_image *const *imgs, int nimg, const modsx_view *views,
                                  int nviews, const modsx_pair_params &pp, const DescSet &ds, std::vector<modsx_region> &regs,
                                  DevBuf *const *descAcc, const size_t *base, int *viewCounts, const unsigned char *wantImg,
                                  std::vector<size_t> *regStart, double **devPos, std::vector<double> *kpRows, const int *ownerImg) {
  regs.clear();
  if (kpRows) kpRows->clear();
  const RowFmt rf = row_fmt(kpRows ? MODSX_SHARD_ROW_KP : MODSX_SHARD_ROW_REGION);
  const size_t RB = (size_t)rf.regLen;     // bytes of a region record on the device, in the staging buffers and on the wire
  const int nv = nimg * nviews;            // items: what the block headers call views
  const int nd = ds.n, ROW_B = row_bytes(rf.regLen, nd);
  if (cm->dead.load()) return comm_dead_rc(cm);
  if (nimg < 1 || nviews < 1 || nv > SHARD_MAXV || cm->world > SHARD_MAXW) { set_error("sharded path: at most 1024 (image, view) items per exchange and 64 ranks"); return MODSX_ERR_ARG; }
  hipStream_t s = c->stream;
  const int W = cm->world, R = cm->rank, lane = lane_of(c, cm);
  ShardLane &L = cm->lanes[lane];
  // 1. this rank's views.  A failure here does not return: it travels in the header, so that no rank waits for this one.
  std::vector<modsx_region> local;
  std::vector<int> cnt(nv, 0);
  int lrc = MODSX_OK;
  std::string lerr;
  size_t cap = std::max<size_t>((size_t)1 << 15, L.localCap);
  for (;;) {
    if (!c->shardLocal.ensure(cap * 128 * nd)) { lrc = MODSX_ERR_NOMEM; break; }   // class k at row k * cap
    uint8_t *xs[3] = {(uint8_t *)c->shardLocal.p + cap * 128, (uint8_t *)c->shardLocal.p + 2 * cap * 128, (uint8_t *)c->shardLocal.p + 3 * cap * 128};
    std::vector<const modsx_image *> myImg;
    std::vector<int> myView, myItem;
    for (int f = R; f < nv; f += W) { myImg.push_back(imgs[f / nviews]); myView.push_back(f % nviews); myItem.push_back(f); }
    std::vector<int> myCnt(std::max<size_t>(1, myItem.size()), 0);
    lrc = detect_describe_items(c, myImg.data(), myView.data(), (int)myItem.size(), views, pp, local, nullptr, (uint8_t *)c->shardLocal.p, cap,
                                nullptr, myCnt.data(), &ds, xs);
    std::fill(cnt.begin(), cnt.end(), 0);
    for (size_t k = 0; k < myItem.size(); k++) cnt[myItem[k]] = myCnt[k];
    if (lrc == MODSX_ERR_CAPACITY && cap < ((size_t)1 << 24)) { cap *= 4; continue; }
    break;
  }
  L.localCap = cap;
  if (lrc) { lerr = last_error(); local.clear(); std::fill(cnt.begin(), cnt.end(), 0); }
  const int nloc = (int)local.size();
  const int hdrB = hdr_bytes(nv);
  if (ownerImg && cm->exchange_mode.load() == MODSX_EXCHANGE_OWNER) {
    // ---- the owner-only exchange: headers to every rank, rows to the rank that reads them (see owner_plan) ----
    for (int j = 0; j < nimg; j++) if (ownerImg[j] < 0 || ownerImg[j] >= W) { set_error("sharded path: an image owner is not a rank"); return MODSX_ERR_ARG; }
    // a. the headers: an all-gather of hdrB bytes per rank; the mode travels in the magic, so ranks set differently stop here
    if ((size_t)hdrB > L.capBlock) {
      int arc = MODSX_OK;
      if (!L.blkLocal.ensure(hdrB) || !L.blkAll.ensure((size_t)hdrB * W)) arc = MODSX_ERR_NOMEM;
      arc = comm_agree(cm, lane, s, arc);
      if (arc) return arc;
      L.capBlock = hdrB;
    }
    if (!L.hHdr.ensure((size_t)hdrB * (W + 1))) { comm_kill(cm, "no pinned memory for a block header"); return MODSX_ERR_NOMEM; }
    int *hh = (int *)L.hHdr.p;
    memset(hh, 0, hdrB);
    hh[0] = HDR_MAGIC_OWNER; hh[1] = lrc; hh[2] = lrc ? 0 : nloc; hh[3] = nv;
    if (!lrc) memcpy(hh + HDR_FIXED, cnt.data(), (size_t)nv * 4);
    MX_HIP(hipMemcpyAsync(L.blkLocal.p, hh, hdrB, hipMemcpyHostToDevice, s));
    int rc = ordered_all_gather(cm, lane, L.blkLocal.p, L.blkAll.p, hdrB, s);
    if (rc) return rc;
    MX_HIP(hipMemcpyAsync((char *)L.hHdr.p + hdrB, L.blkAll.p, (size_t)hdrB * W, hipMemcpyDeviceToHost, s));
    rc = comm_wait(cm, s);
    if (rc) return rc;
    std::vector<int> vc(nv, 0);
    for (int r = 0; r < W; r++) {
      const int *h = (const int *)((char *)L.hHdr.p + (size_t)hdrB * (r + 1));
      if (h[0] != HDR_MAGIC_OWNER || h[3] != nv) { comm_kill(cm, "a gathered header is malformed (ranks out of step, or set to different exchange modes)"); return comm_dead_rc(cm); }
      if (h[1]) {
        if (r == R) set_error(lerr); else set_error("rank " + std::to_string(r) + " failed in the sharded detect / describe (" + std::to_string(h[1]) + ")");
        return h[1];
      }
      for (int v = r; v < nv; v += W) vc[v] = h[HDR_FIXED + v];
    }
    // b. the plan, and the buffers it needs: capacities are the maxima over the ranks, so every rank grows at the same calls
    OwnerPlan P;
    owner_plan(vc.data(), nimg, nviews, W, R, ownerImg, P);
    if ((size_t)nloc != P.sendRows) { comm_kill(cm, "a rank's header does not count its rows"); return comm_dead_rc(cm); }
    const size_t listRows = std::max<size_t>(P.N, 1);
    if (P.maxSendRows > L.capOwnSend || P.maxRecvRows > L.capOwnRecv || listRows > L.capOwnList) {
      const size_t ns = std::max(L.capOwnSend, P.maxSendRows + P.maxSendRows / 4 + 64), nr = std::max(L.capOwnRecv, P.maxRecvRows + P.maxRecvRows / 4 + 64);
      const size_t nl = std::max(L.capOwnList, listRows + listRows / 4 + 64);
      int arc = MODSX_OK;
      // capacities are agreed in ROWS, so the bytes behind them are those of the widest row any later call on this lane may
      // carry (full region record + MODSX_MAX_DESC descriptor classes): a call with more classes and row counts under the
      // caps does not pass through this branch again
      const size_t ROW_CAP = (size_t)row_bytes(REG_B, MODSX_MAX_DESC), RB_CAP = (size_t)REG_B;
      if (!L.ownSend.ensure(ns * ROW_CAP) || !L.ownRecv.ensure(nr * ROW_CAP) || !L.regsIn.ensure(ns * RB_CAP) || !L.hRegs.ensure(std::max(ns, nl) * RB_CAP) ||
          !L.regsOut.ensure(nl * RB_CAP) || !L.posOut.ensure(nl * 16))
        arc = MODSX_ERR_NOMEM;
      for (int k = 0; k < nd && !arc; k++)
        if (grow_keep(s, *descAcc[k], base[k] * 128, (base[k] + nl) * 128) != MODSX_OK) arc = MODSX_ERR_NOMEM;
      arc = comm_agree(cm, lane, s, arc);
      if (arc) return arc;
      L.capOwnSend = ns; L.capOwnRecv = nr; L.capOwnList = nl;
    }
    // the caller's descriptor lists are its own: they may be other buffers than at the last agreed growth
    for (int k = 0; k < nd; k++)
      if (grow_keep(s, *descAcc[k], base[k] * 128, (base[k] + L.capOwnList) * 128) != MODSX_OK) { comm_kill(cm, "out of memory for a descriptor list"); return MODSX_ERR_NOMEM; }
    const size_t jobB = std::max<size_t>(1, P.jobs.size()) * sizeof(OwnerJob);
    if (!L.hJobs.ensure(jobB) || !L.ownJobs.ensure(jobB)) { comm_kill(cm, "out of memory for the unpack jobs"); return MODSX_ERR_NOMEM; }
    // c. this rank's rows, packed in local (item) order
    if (nloc) {
      if (rf.regLen == REG_B) memcpy(L.hRegs.p, local.data(), (size_t)nloc * REG_B);
      else for (int i = 0; i < nloc; i++) memcpy((char *)L.hRegs.p + (size_t)i * RB, (const char *)&local[i] + rf.regOff, RB);
      MX_HIP(hipMemcpyAsync(L.regsIn.p, L.hRegs.p, (size_t)nloc * RB, hipMemcpyHostToDevice, s));
      DescPtrs dp;
      for (int k = 0; k < MODSX_MAX_DESC; k++) dp.p[k] = (unsigned char *)c->shardLocal.p + (size_t)k * cap * 128;
      hipLaunchKernelGGL(k_pack_rows, dim3((nloc + 7) / 8), dim3(256), 0, s, (const unsigned char *)L.regsIn.p, rf.regLen, dp, nd, nloc,
                         (unsigned char *)L.ownSend.p);
    }
    // d. the exchange: one group of sends and receives, in the lane's turn
    std::vector<Xfer> sends, recvs;
    for (const OwnerMsg &m : P.sends) sends.push_back(Xfer{m.peer, (char *)L.ownSend.p + (size_t)m.row0 * ROW_B, (size_t)m.rows * ROW_B});
    for (const OwnerMsg &m : P.recvs) recvs.push_back(Xfer{m.peer, (char *)L.ownRecv.p + (size_t)m.row0 * ROW_B, (size_t)m.rows * ROW_B});
    rc = turn_begin(cm, lane);
    if (rc) return rc;
    rc = transport_exchange(cm, sends, recvs, s);
    turn_end(cm);
    if (rc) return rc;
    // e. the owned images' rows to their places in the list
    if (!P.jobs.empty()) {
      memcpy(L.hJobs.p, P.jobs.data(), P.jobs.size() * sizeof(OwnerJob));
      MX_HIP(hipMemcpyAsync(L.ownJobs.p, L.hJobs.p, P.jobs.size() * sizeof(OwnerJob), hipMemcpyHostToDevice, s));
      int maxn = 0;
      for (const OwnerJob &jb : P.jobs) maxn = std::max(maxn, jb.n);
      DescPtrs dp;
      for (int k = 0; k < MODSX_MAX_DESC; k++) dp.p[k] = k < nd ? (unsigned char *)descAcc[k]->p + base[k] * 128 : nullptr;
      hipLaunchKernelGGL(k_unpack_jobs, dim3((maxn + 7) / 8, (unsigned)P.jobs.size()), dim3(256), 0, s, (const OwnerJob *)L.ownJobs.p,
                         (const unsigned char *)L.ownRecv.p, (unsigned char *)L.regsOut.p, rf.regLen, dp, nd, devPos ? (double *)L.posOut.p : nullptr, rf.posOfs);
    }
    L.lastN = P.N;
    if (viewCounts) for (int v = 0; v < nv; v++) viewCounts[v] = vc[v];
    if (devPos) *devPos = (double *)L.posOut.p;
    // f. the regions the caller reads on the host: the flagged images it owns (others were never received)
    std::vector<size_t> st(nimg + 1, 0), devStart(nimg + 1, 0);
    for (int j = 0; j < nimg; j++) {
      size_t n = 0;
      for (int v = 0; v < nviews; v++) n += (size_t)vc[(size_t)j * nviews + v];
      devStart[j + 1] = devStart[j] + n;
      st[j + 1] = st[j] + ((!wantImg || wantImg[j]) && ownerImg[j] == R ? n : 0);
    }
    for (int j = 0; j < nimg; j++)
      if (st[j + 1] > st[j])
        MX_HIP(hipMemcpyAsync((char *)L.hRegs.p + st[j] * RB, (char *)L.regsOut.p + devStart[j] * RB, (devStart[j + 1] - devStart[j]) * RB, hipMemcpyDeviceToHost, s));
    rc = comm_wait(cm, s);
    if (rc) return rc;
    MX_HIP(hipGetLastError());
    if (kpRows) { kpRows->resize(st[nimg] * 7); if (st[nimg]) memcpy(kpRows->data(), L.hRegs.p, st[nimg] * RB); }
    else { regs.resize(st[nimg]); if (st[nimg]) memcpy(regs.data(), L.hRegs.p, st[nimg] * REG_B); }
    if (regStart) *regStart = st;
    return MODSX_OK;
  }
  if (!L.guessRows) L.guessRows = 4096;
  for (int attempt = 0;; attempt++) {
    const int G = L.guessRows;
    const size_t blockB = (size_t)hdrB + (size_t)G * ROW_B;
    // 2. buffers a rank needs to take part grow at points every rank computes alike, and the outcome is agreed
    if (blockB > L.capBlock) {
      int arc = MODSX_OK;
      if (!L.blkLocal.ensure(blockB) || !L.blkAll.ensure(blockB * W)) arc = MODSX_ERR_NOMEM;
      arc = comm_agree(cm, lane, s, arc);
      if (arc) return arc;
      L.capBlock = blockB;
    }
    // 3. everything else this rank needs: a failure is reported through the header
    const size_t rowsCap = (size_t)W * G;
    if (!L.hHdr.ensure((size_t)hdrB * (W + 1))) { comm_kill(cm, "no pinned memory for a block header"); return MODSX_ERR_NOMEM; }
    if (!lrc) {
      if (!L.regsIn.ensure((size_t)std::max(1, nloc) * RB) || !L.hRegs.ensure(std::max((size_t)std::max(1, nloc), rowsCap) * RB) ||
          !L.regsOut.ensure(rowsCap * RB) || (devPos && !L.posOut.ensure(rowsCap * 16))) {
        lrc = MODSX_ERR_NOMEM; lerr = "sharded path: out of memory";
      }
      for (int k = 0; k < nd && !lrc; k++)
        if (grow_keep(s, *descAcc[k], base[k] * 128, (base[k] + rowsCap) * 128) != MODSX_OK) { lrc = MODSX_ERR_NOMEM; lerr = "sharded path: out of memory"; }
    }
    int *hh = (int *)L.hHdr.p;
    memset(hh, 0, hdrB);
    hh[0] = HDR_MAGIC; hh[1] = lrc; hh[2] = lrc ? 0 : nloc; hh[3] = nv;
    if (!lrc) memcpy(hh + HDR_FIXED, cnt.data(), (size_t)nv * 4);
    MX_HIP(hipMemcpyAsync(L.blkLocal.p, hh, hdrB, hipMemcpyHostToDevice, s));
    const int npack = lrc ? 0 : std::min(nloc, G);
    if (npack) {
      if (rf.regLen == REG_B) memcpy(L.hRegs.p, local.data(), (size_t)npack * REG_B);
      else for (int i = 0; i < npack; i++) memcpy((char *)L.hRegs.p + (size_t)i * RB, (const char *)&local[i] + rf.regOff, RB);
      MX_HIP(hipMemcpyAsync(L.regsIn.p, L.hRegs.p, (size_t)npack * RB, hipMemcpyHostToDevice, s));
      DescPtrs dp;
      for (int k = 0; k < MODSX_MAX_DESC; k++) dp.p[k] = (unsigned char *)c->shardLocal.p + (size_t)k * cap * 128;
      hipLaunchKernelGGL(k_pack_rows, dim3((npack + 7) / 8), dim3(256), 0, s, (const unsigned char *)L.regsIn.p, rf.regLen, dp, nd, npack,
                         (unsigned char *)L.blkLocal.p + hdrB);
    }
    // 4. the exchange: one all-gather of the blocks
    int rc = ordered_all_gather(cm, lane, L.blkLocal.p, L.blkAll.p, blockB, s);
    if (rc) return rc;
    // 5. headers down; reference order on the device; a first guess of the regions down with the same wait
    MX_HIP(hipMemcpy2DAsync((char *)L.hHdr.p + hdrB, hdrB, L.blkAll.p, blockB, hdrB, W, hipMemcpyDeviceToHost, s));
    size_t got = 0;
    if (!lrc) {
      DescPtrs dp;
      for (int k = 0; k < MODSX_MAX_DESC; k++) dp.p[k] = k < nd ? (unsigned char *)descAcc[k]->p + base[k] * 128 : nullptr;
      hipLaunchKernelGGL(k_unpack_blocks, dim3((unsigned)((rowsCap + 7) / 8)), dim3(256), 0, s, (const unsigned char *)L.blkAll.p, W, nv, G,
                         blockB, hdrB, (unsigned char *)L.regsOut.p, rf.regLen, dp, nd, rowsCap, devPos ? (double *)L.posOut.p : nullptr,
                         rf.posOfs);
      if (!wantImg) {     // all regions are wanted: a first guess comes down with the same wait as the headers
        got = std::min(rowsCap, L.lastN + L.lastN / 4 + 256);
        MX_HIP(hipMemcpyAsync(L.hRegs.p, L.regsOut.p, got * RB, hipMemcpyDeviceToHost, s));
      }
    }
    rc = comm_wait(cm, s);
    if (rc) return rc;
    MX_HIP(hipGetLastError());
    // 6. what every rank sees alike: a failed rank, or a block that was too small
    int maxrows = 0;
    size_t N = 0;
    std::vector<int> vc(nv, 0);
    for (int r = 0; r < W; r++) {
      const int *h = (const int *)((char *)L.hHdr.p + (size_t)hdrB * (r + 1));
      if (h[0] != HDR_MAGIC || h[3] != nv) { comm_kill(cm, "a gathered block header is malformed (ranks out of step)"); return comm_dead_rc(cm); }
      if (h[1]) {
        if (r == R) set_error(lerr); else set_error("rank " + std::to_string(r) + " failed in the sharded detect / describe (" + std::to_string(h[1]) + ")");
        return h[1];
      }
      maxrows = std::max(maxrows, h[2]);
      for (int v = r; v < nv; v += W) { vc[v] = h[HDR_FIXED + v]; N += (size_t)vc[v]; }
    }
    if (maxrows > G) {        // every rank takes this branch together
      L.guessRows = maxrows + maxrows / 4 + 64;
      cm->retries++;
      if (attempt > 2) { comm_kill(cm, "block size does not converge"); return comm_dead_rc(cm); }
      continue;
    }
    L.guessRows = std::max(G, maxrows + maxrows / 4 + 64);
    L.lastN = N;
    if (viewCounts) for (int v = 0; v < nv; v++) viewCounts[v] = vc[v];
    if (devPos) *devPos = (double *)L.posOut.p;
    if (wantImg) {
      // only the flagged images: their slices are known now (the counts came with the headers), one copy each, one more wait
      std::vector<size_t> st(nimg + 1, 0), devStart(nimg + 1, 0);
      for (int j = 0; j < nimg; j++) {
        size_t n = 0;
        for (int v = 0; v < nviews; v++) n += (size_t)vc[(size_t)j * nviews + v];
        devStart[j + 1] = devStart[j] + n;
        st[j + 1] = st[j] + (wantImg[j] ? n : 0);
      }
      bool any = false;
      for (int j = 0; j < nimg; j++)
        if (wantImg[j] && devStart[j + 1] > devStart[j]) {
          MX_HIP(hipMemcpyAsync((char *)L.hRegs.p + st[j] * RB, (char *)L.regsOut.p + devStart[j] * RB,
                                (devStart[j + 1] - devStart[j]) * RB, hipMemcpyDeviceToHost, s));
          any = true;
        }
      if (any) { rc = comm_wait(cm, s); if (rc) return rc; }
      if (kpRows) { kpRows->resize(st[nimg] * 7); if (st[nimg]) memcpy(kpRows->data(), L.hRegs.p, st[nimg] * RB); }
      else { regs.resize(st[nimg]); if (st[nimg]) memcpy(regs.data(), L.hRegs.p, st[nimg] * REG_B); }
      if (regStart) *regStart = st;
      return MODSX_OK;
    }
    if (N > got) {           // the speculative download was short (first call, or a much larger image)
      MX_HIP(hipMemcpyAsync((char *)L.hRegs.p + got * RB, (char *)L.regsOut.p + got * RB, (N - got) * RB, hipMemcpyDeviceToHost, s));
      rc = comm_wait(cm, s);
      if (rc) return rc;
    }
    if (kpRows) { kpRows->resize(N * 7); if (N) memcpy(kpRows->data(), L.hRegs.p, N * RB); }
    else { regs.resize(N); if (N) memcpy(regs.data(), L.hRegs.p, N * REG_B); }
    return MODSX_OK;
  }
}

// MatchFlannFGINN with the query rows split over the ranks (d1 / d2 hold ALL descriptors on every rank)
int match_sharded(modsx_ctx *c, modsx_comm *cm, const uint8_t *d1, int n1, const uint8_t *d2, int n2, const double *pos2Host,
                  double ratioT, double contradDist, int nn, std::vector<modsx_tentative> &out) {
  out.clear();
  if (cm->dead.load()) return comm_dead_rc(cm);
  if (n1 <= 0 || n2 <= 0) return MODSX_OK;   // the same on every rank: nobody issues a collective
  const int W = cm->world, R = cm->rank;
  const int per = (n1 + W - 1) / W;                 // rows per rank (the last ranks may hold fewer or none)
  const int lo = std::min(n1, R * per);
  MatchShard sh;
  sh.comm = cm; sh.world = W; sh.per = per; sh.n1_total = n1; sh.lo = lo;
  return match_device_batch(c, 1, &d1, &n1, &d2, &n2, &pos2Host, ratioT, contradDist, nn, &out, &sh);
}

// match_device_batch, sharded branch, step 1: the lane's blocks ((per + 1) result rows: a header row, then the rows of this
// rank's queries).  Growth is agreed; *blk = the local block.
int match_shard_begin(modsx_ctx *c, const MatchShard &sh, MatchRow **blk) {
  modsx_comm *cm = (modsx_comm *)sh.comm;
  const int lane = lane_of(c, cm);
  ShardLane &L = cm->lanes[lane];
  const size_t blockB = (size_t)(sh.per + 1) * sizeof(MatchRow);
  if (blockB > L.capMatch) {
    int arc = MODSX_OK;
    if (!L.mLocal.ensure(blockB) || !L.mAll.ensure(blockB * cm->world)) arc = MODSX_ERR_NOMEM;
    arc = comm_agree(cm, lane, c->stream, arc);
    if (arc) return arc;
    L.capMatch = blockB;
  }
  *blk = (MatchRow *)L.mLocal.p;
  return MODSX_OK;
}
// step 2, after the matcher launches: header (with this rank's local result), all-gather, the gathered blocks to `host`
// (world x (per + 1) rows), one wait.  Returns the agreed result: the first failure in rank order.
int match_shard_gather(modsx_ctx *c, const MatchShard &sh, int local_rc, MatchRow *host) {
  modsx_comm *cm = (modsx_comm *)sh.comm;
  const int lane = lane_of(c, cm), W = cm->world;
  ShardLane &L = cm->lanes[lane];
  hipStream_t s = c->stream;
  const size_t blockB = (size_t)(sh.per + 1) * sizeof(MatchRow);
  static_assert(sizeof(MatchRow) == 32, "match rows are 32 bytes on the wire");
  int *h = (int *)L.hRc.p + 64;
  h[0] = HDR_MAGIC; h[1] = local_rc; h[2] = sh.per; h[3] = sh.n1_total;
  MX_HIP(hipMemcpyAsync(L.mLocal.p, h, 16, hipMemcpyHostToDevice, s));
  int rc = ordered_all_gather(cm, lane, L.mLocal.p, L.mAll.p, blockB, s);
  if (rc) return rc;
  if (host) MX_HIP(hipMemcpyAsync(host, L.mAll.p, blockB * W, hipMemcpyDeviceToHost, s));
  else MX_HIP(hipMemcpy2DAsync((int *)L.hRc.p + 128, 16, L.mAll.p, blockB, 16, W, hipMemcpyDeviceToHost, s));
  rc = comm_wait(cm, s);
  if (rc) return rc;
  MX_HIP(hipGetLastError());
  for (int r = 0; r < W; r++) {
    const int *g = host ? (const int *)((const char *)host + blockB * r) : (const int *)L.hRc.p + 128 + 4 * r;
    if (g[0] != HDR_MAGIC || g[2] != sh.per || g[3] != sh.n1_total) { comm_kill(cm, "a gathered match block is malformed (ranks out of step)"); return comm_dead_rc(cm); }
    if (g[1]) {
      if (r != cm->rank) set_error("rank " + std::to_string(r) + " failed in the sharded match (" + std::to_string(g[1]) + ")");
      return g[1];
    }
  }
  return local_rc;
}

// MatchFlannFGINN for nb problems at once, each with its query rows split over the ranks: ONE all-gather moves the result rows of
// all of them (block = header row + sum of the per-problem row shares), the problems share the matcher's launches four at a time.
int match_sharded_batch(modsx_ctx *c, modsx_comm *cm, int nb, const uint8_t *const *d1, const int *n1, const uint8_t *const *d2,
                        const int *n2, const double *const *pos2Host, double ratioT, double contradDist, int nn,
                        std::vector<modsx_tentative> *out, const double *const *pos2Dev) {
  // pos2Dev (optional): the positions already live on the device (the exchange wrote them); pos2Host is then not read
  for (int g = 0; g < nb; g++) out[g].clear();
  if (cm->dead.load()) return comm_dead_rc(cm);
  const double sqminratio = ratioT * ratioT, contrDistSq = contradDist * contradDist;
  if (!(sqminratio == sqminratio)) { set_error("match ratio is NaN"); return MODSX_ERR_ARG; }   // ratio >= 1: the "all points" branch (matching.cpp:397-428)
  if (nn < 2 || nn > MATCH_NN_MAX) { set_error("match: nn must be in [2, 256]"); return MODSX_ERR_ARG; }
  const int W = cm->world, R = cm->rank;
  hipStream_t s = c->stream;
  auto up = [](size_t b) { return (b + 255) & ~(size_t)255; };
  // the row shares: every rank derives the same layout from the gathered counts
  std::vector<int> per(nb), lo(nb), nloc(nb), off(nb);
  int rowsTot = 0, n1Tot = 0;
  size_t posB = 0, workB = 0;
  std::vector<size_t> posOfs(nb), workOfs(nb);
  for (int g = 0; g < nb; g++) {
    const bool live = n1[g] > 0 && n2[g] > 0;
    per[g] = live ? (n1[g] + W - 1) / W : 0;
    lo[g] = std::min(n1[g], R * per[g]);
    nloc[g] = live ? std::max(0, std::min(n1[g], lo[g] + per[g]) - lo[g]) : 0;
    off[g] = 1 + rowsTot;
    rowsTot += per[g]; n1Tot += live ? n1[g] : 0;
    posOfs[g] = posB; posB += up((size_t)std::max(0, n2[g]) * 16);
    workOfs[g] = workB; workB += up(match_workspace_bytes(std::max(1, nloc[g]), std::max(1, n2[g])));
  }
  if (rowsTot == 0) return MODSX_OK;       // the same on every rank: nobody issues a collective
  MatchShard sh;
  sh.comm = cm; sh.world = W; sh.per = rowsTot; sh.n1_total = n1Tot; sh.lo = 0;
  MatchRow *blk = nullptr;
  int rc = match_shard_begin(c, sh, &blk);
  if (rc) return rc;
  int lrc = MODSX_OK;
  const size_t allB = (size_t)W * (rowsTot + 1) * sizeof(MatchRow);
  if (!c->pos2.ensure(posB + 256) || !c->hMatch.ensure(posB + up(allB) + 256) || !c->matchWork.ensure(workB + 256)) lrc = MODSX_ERR_NOMEM;
  char *hpos = (char *)c->hMatch.p, *hrow = hpos ? hpos + posB : nullptr;
  if (!lrc && !pos2Dev) {
    for (int g = 0; g < nb; g++) if (n2[g] > 0) memcpy(hpos + posOfs[g], pos2Host[g], (size_t)n2[g] * 16);
    if (hipMemcpyAsync(c->pos2.p, hpos, posB, hipMemcpyHostToDevice, s) != hipSuccess) { set_error("sharded match: upload failed"); lrc = MODSX_ERR_DEVICE; }
  }
  if (!lrc) {
    // live problems share the launches (blockIdx.z), MATCH_MAXB at a time
    const uint8_t *pd1[MATCH_MAXB], *pd2[MATCH_MAXB];
    const double *ppos[MATCH_MAXB];
    MatchRow *prow[MATCH_MAXB];
    void *pwork[MATCH_MAXB];
    int pn1[MATCH_MAXB], pn2[MATCH_MAXB], k = 0;
    double work = 0;
    auto flush = [&]() {
      if (!k) return;
      size_t pslot;
      prof_begin(c, K_MATCH, work, &pslot);
      launch_match_batch(s, k, pd1, pn1, pd2, pn2, ppos, sqminratio, contrDistSq, nn, prow, pwork);
      prof_end(c, pslot);
      k = 0; work = 0;
    };
    for (int g = 0; g < nb; g++) {
      if (nloc[g] <= 0) continue;
      pd1[k] = d1[g] + (size_t)lo[g] * 128; pn1[k] = nloc[g]; pd2[k] = d2[g]; pn2[k] = n2[g];
      ppos[k] = pos2Dev ? pos2Dev[g] : (const double *)((char *)c->pos2.p + posOfs[g]);
      prow[k] = blk + off[g]; pwork[k] = (char *)c->matchWork.p + workOfs[g];
      work += 2.0 * nloc[g] * (double)n2[g] * 128;
      if (++k == MATCH_MAXB) flush();
    }
    flush();
  }
  rc = match_shard_gather(c, sh, lrc, lrc ? nullptr : (MatchRow *)hrow);
  if (rc) return rc;
  std::vector<MatchRow> rows;
  for (int g = 0; g < nb; g++) {
    if (per[g] <= 0) continue;
    rows.resize((size_t)n1[g]);
    for (int r = 0; r < W; r++) {
      const int rlo = std::min(n1[g], r * per[g]), rn = std::min(n1[g], rlo + per[g]) - rlo;
      if (rn > 0) memcpy(rows.data() + rlo, (const MatchRow *)hrow + (size_t)r * (rowsTot + 1) + off[g], (size_t)rn * sizeof(MatchRow));
    }
    rows_to_tentatives(rows.data(), n1[g], nn, out[g]);
  }
  return MODSX_OK;
}

// modsx_match_pairs_views_sharded: np pairs in ONE sharded call.  The views of all 2 np images are one item list (a rank's
// launch sets are ~2 np nviews / world views again, whatever the world size), ONE exchange moves every image side, the np
// matching problems of a descriptor class share ONE result all-gather, and pair g is verified by rank (owner_base + g) mod
// world (owner_base < 0: by every rank).  Per pair the result is that of modsx_match_pair_views.
int match_pairs_views_sharded(modsx_ctx *c, modsx_comm *cm, const modsx_image *const *imgs1, const modsx_image *const *imgs2, int np,
                              const modsx_view *views, int nv, const modsx_pair_params &pp, int owner_base, modsx_pair_result *results) {
  for (int g = 0; g < np; g++) { memset(&results[g], 0, sizeof results[g]); for (int i = 0; i < 9; i++) results[g].H[i] = -1; }
  DescSet ds;
  int rc = resolve_descs(pp, nullptr, ds);
  if (rc) return rc;
  const int nimg = 2 * np;
  std::vector<const modsx_image *> imgs(nimg);
  for (int g = 0; g < np; g++) { imgs[2 * g] = imgs1[g]; imgs[2 * g + 1] = imgs2[g]; }
  std::vector<modsx_region> regs;
  std::vector<int> counts((size_t)nimg * nv, 0);
  DevBuf *acc[MODSX_MAX_DESC];
  size_t base0[MODSX_MAX_DESC] = {0, 0, 0, 0};
  for (int k = 0; k < ds.n; k++) acc[k] = &c->descCls[0][k][0];      // scratch of this call: class slot k of the list
  // the regions of a pair come to the host only on the rank that verifies it; the matcher's positions stay on the device
  std::vector<unsigned char> want(nimg, 0);
  for (int g = 0; g < np; g++) {
    const bool mine = owner_base < 0 || (owner_base + g) % cm->world == cm->rank;
    want[2 * g] = want[2 * g + 1] = mine ? 1 : 0;
  }
  std::vector<size_t> hs;      // slice of image j in `kp` (empty when not wanted)
  std::vector<double> kp;      // the call returns pair results, no region lists: a row carries the verification slice of a region only
  double *devPos = nullptr;
  // owner_base >= 0: only the owner of a pair reads its rows -- with MODSX_EXCHANGE_OWNER they travel to that rank alone
  std::vector<int> ownerImg(nimg, 0);
  for (int g = 0; g < np && owner_base >= 0; g++) ownerImg[2 * g] = ownerImg[2 * g + 1] = (owner_base + g) % cm->world;
  rc = detect_describe_items_sharded(c, cm, imgs.data(), nimg, views, nv, pp, ds, regs, acc, base0, counts.data(), want.data(), &hs, &devPos, &kp,
                                     owner_base >= 0 ? ownerImg.data() : nullptr);
  if (rc) return rc;
  // the slice of every image in the gathered (device) lists
  std::vector<size_t> start(nimg + 1, 0);
  for (int j = 0; j < nimg; j++) { size_t n = 0; for (int v = 0; v < nv; v++) n += (size_t)counts[(size_t)j * nv + v]; start[j + 1] = start[j] + n; }
  int ord[MODSX_MAX_DESC];
  desc_class_order(ds, ord);
  std::vector<std::vector<modsx_tentative>> tents(np);
  std::vector<int> n1(np), n2(np);
  std::vector<const double *> pdev(np);
  for (int g = 0; g < np; g++) {
    n1[g] = (int)(start[2 * g + 1] - start[2 * g]); n2[g] = (int)(start[2 * g + 2] - start[2 * g + 1]);
    pdev[g] = devPos + 2 * start[2 * g + 1];
  }
  for (int oi = 0; oi < ds.n; oi++) {
    const int k = ord[oi];
    std::vector<const uint8_t *> d1(np), d2(np);
    std::vector<std::vector<modsx_tentative>> part(np);
    for (int g = 0; g < np; g++) {
      d1[g] = (const uint8_t *)acc[k]->p + start[2 * g] * 128; d2[g] = (const uint8_t *)acc[k]->p + start[2 * g + 1] * 128;
    }
    if (owner_base < 0) {
      // every rank wants every pair: the query rows of each problem are split over the ranks, one all-gather of result rows
      rc = match_sharded_batch(c, cm, np, d1.data(), n1.data(), d2.data(), n2.data(), nullptr, ds.ratio[k], pp.contradDist, pp.nn, part.data(),
                               pdev.data());
      if (rc) return rc;
    } else {
      // a pair is matched where it is verified: the exchange left every descriptor and position of it on this rank, so the owner
      // runs the whole problem (with np = world pairs per call that is one problem per rank, the load the row split had) and NO
      // second collective follows -- a lane's turn comes once per call, and the lanes keep the stagger a ring order allows
      std::vector<int> mine;
      for (int g = 0; g < np; g++) if ((owner_base + g) % cm->world == cm->rank) mine.push_back(g);
      for (size_t m0 = 0; m0 < mine.size(); m0 += MATCH_MAXB) {
        const int nm = (int)std::min<size_t>(MATCH_MAXB, mine.size() - m0);
        const uint8_t *q1[MATCH_MAXB], *q2[MATCH_MAXB];
        const double *qp[MATCH_MAXB];
        int m1[MATCH_MAXB], m2[MATCH_MAXB];
        std::vector<modsx_tentative> tmp[MATCH_MAXB];
        for (int i = 0; i < nm; i++) { const int g = mine[m0 + i]; q1[i] = d1[g]; q2[i] = d2[g]; qp[i] = pdev[g]; m1[i] = n1[g]; m2[i] = n2[g]; }
        rc = match_device_batch(c, nm, q1, m1, q2, m2, nullptr, ds.ratio[k], pp.contradDist, pp.nn, tmp, nullptr, qp);
        if (rc) return rc;
        for (int i = 0; i < nm; i++) part[mine[m0 + i]].swap(tmp[i]);
      }
    }
    for (int g = 0; g < np; g++) {
      const int o1 = oi * n1[g], o2 = oi * n2[g];
      if (oi == 0) { tents[g].swap(part[g]); continue; }
      for (modsx_tentative t : part[g]) {
        t.q += o1; t.t0 += o2;
        if (t.t1 >= 0) t.t1 += o2;
        if (t.tj >= 0) t.tj += o2;
        tents[g].push_back(t);
      }
    }
  }
  for (int g = 0; g < np; g++) {
    modsx_pair_result *res = &results[g];
    res->n_regions1 = n1[g] * ds.n; res->n_regions2 = n2[g] * ds.n;
    res->n_tentatives = (int)tents[g].size();
    if (owner_base >= 0 && (owner_base + g) % cm->world != cm->rank) continue;
    verify_tentatives_kp(kp.data() + 7 * hs[2 * g], (size_t)n1[g], kp.data() + 7 * hs[2 * g + 1], (size_t)n2[g], tents[g], pp, res);
  }
  prof_collect(c);
  return MODSX_OK;
}

int match_pair_views_sharded(modsx_ctx *c, modsx_comm *cm, const modsx_image *img1, const modsx_image *img2, const modsx_view *views,
                             int nv, const modsx_pair_params &pp, int owner, modsx_pair_result *res) {
  modsx_ladder_step one;
  memset(&one, 0, sizeof one);
  one.views = views; one.nviews = nv; one.match_ratio = pp.match_ratio; one.detector = pp.detector;
  int done = 0;
  return match_ladder(c, img1, img2, &one, 1, 0x7fffffff, pp, res, &done, nullptr, cm, owner);
}

}  // namespace mx

using namespace mx;
extern "C" {

int modsx_comm_unique_id(void *id128) {
  if (!id128) { mx::set_error("modsx_comm_unique_id: null"); return MODSX_ERR_ARG; }
  if (!rccl_load()) return MODSX_ERR_DEVICE;
  ncclUniqueId id;
  ncclResult_t r = g_rccl.GetUniqueId(&id);
  if (r != ncclSuccess) { mx::set_error(std::string("ncclGetUniqueId: ") + g_rccl.GetErrorString(r)); return MODSX_ERR_DEVICE; }
  static_assert(sizeof id <= 128, "the id buffer of the C ABI is 128 bytes");
  memset(id128, 0, 128);
  memcpy(id128, &id, sizeof id);
  return MODSX_OK;
}

int modsx_comm_loopback_id(void *id128, int world) {
  if (!id128 || world < 1 || world > SHARD_MAXW) { mx::set_error("modsx_comm_loopback_id: bad argument"); return MODSX_ERR_ARG; }
  std::lock_guard<std::mutex> lk(g_loopMu);
  const uint64_t key = g_loopNext++;
  auto g = std::make_shared<LoopGroup>();
  g->world = world;
  g->slots.resize(world);
  g_loopGroups[key] = g;
  memset(id128, 0, 128);
  memcpy(id128, LOOP_MAGIC, 8);
  memcpy((char *)id128 + 8, &key, 8);
  return MODSX_OK;
}

static int comm_make_lanes(modsx_comm *cm, int n) {
  cm->lanes.clear();
  cm->lanes.resize(n);
  int guess = 0;   // MODSX_SHARD_BLOCK_ROWS: rows of the first exchange's blocks (tests use a small value to reach the retry path)
  if (const char *e = getenv("MODSX_SHARD_BLOCK_ROWS")) guess = std::max(1, atoi(e));
  for (ShardLane &L : cm->lanes) L.guessRows = guess;
  for (ShardLane &L : cm->lanes)
    if (!L.rcDev.ensure(4 * (16 + SHARD_MAXW)) || !L.hRc.ensure(4 * (128 + 4 * SHARD_MAXW))) return MODSX_ERR_NOMEM;
  cm->turn = 0;
  return MODSX_OK;
}

modsx_comm *modsx_comm_create(modsx_ctx *ctx, const void *id128, int rank, int world) {
  if (!ctx || !id128 || world < 1 || world > SHARD_MAXW || rank < 0 || rank >= world) { mx::set_error("modsx_comm_create: bad argument"); return nullptr; }
  hipSetDevice(ctx->dev);
  std::unique_ptr<modsx_comm> cm(new modsx_comm());
  cm->rank = rank; cm->world = world; cm->dev = ctx->dev;
  if (const char *e = getenv("MODSX_COMM_TIMEOUT_MS")) cm->timeout_ms.store(std::max(1, atoi(e)));
  cm->turn_timeout_ms.store(getenv("MODSX_COMM_TURN_TIMEOUT_MS") ? std::max(1, atoi(getenv("MODSX_COMM_TURN_TIMEOUT_MS"))) : 10 * cm->timeout_ms.load());
  if (!memcmp(id128, LOOP_MAGIC, 8)) {
    uint64_t key;
    memcpy(&key, (const char *)id128 + 8, 8);
    std::lock_guard<std::mutex> lk(g_loopMu);
    auto it = g_loopGroups.find(key);
    if (it == g_loopGroups.end() || it->second->world != world) { mx::set_error("modsx_comm_create: unknown loopback group or wrong world size"); return nullptr; }
    cm->loop = it->second;
    cm->loop->dev = ctx->dev;
    LoopGroup::Slot &sl = cm->loop->slots[rank];
    if (sl.taken) { mx::set_error("modsx_comm_create: loopback rank already taken"); return nullptr; }
    if ((!sl.ready && hipEventCreateWithFlags(&sl.ready, hipEventDisableTiming) != hipSuccess) ||
        (!sl.done && hipEventCreateWithFlags(&sl.done, hipEventDisableTiming) != hipSuccess)) {
      mx::set_error("modsx_comm_create: hipEventCreate failed");     // the slot stays free; what was created is the group's
      return nullptr;
    }
    if (comm_make_lanes(cm.get(), 1)) return nullptr;                 // before the slot is taken: a failure leaves it free
    sl.taken = true;
    if (++cm->loop->joined == world) g_loopGroups.erase(it);   // complete: the id cannot be joined again
  } else {
    if (!rccl_load()) return nullptr;
    ncclUniqueId id;
    memcpy(&id, id128, sizeof id);
    ncclResult_t r = g_rccl.CommInitRank(&cm->nccl, world, id, rank);
    if (r != ncclSuccess) { mx::set_error(std::string("ncclCommInitRank: ") + g_rccl.GetErrorString(r)); return nullptr; }
    g_rccl.GetVersion(&cm->version);
  }
  if (!cm->loop && comm_make_lanes(cm.get(), 1)) { if (cm->nccl) g_rccl.CommDestroy(cm->nccl); return nullptr; }
  ctx->shardLane = 0;
  return cm.release();
}

int modsx_comm_set_lanes(modsx_comm *cm, int nlanes) {
  if (!cm || nlanes < 1 || nlanes > 256) { mx::set_error("modsx_comm_set_lanes: bad argument"); return MODSX_ERR_ARG; }
  hipSetDevice(cm->dev);
  std::lock_guard<std::mutex> lk(cm->mu);
  for (ShardLane &L : cm->lanes) {
    DevBuf *bufs[] = {&L.blkLocal, &L.blkAll, &L.regsIn, &L.regsOut, &L.mLocal, &L.mAll, &L.rcDev, &L.order, &L.posOut, &L.ownSend, &L.ownRecv, &L.ownJobs};
    for (DevBuf *b : bufs) b->release();
    L.hRegs.release(); L.hHdr.release(); L.hRc.release(); L.hJobs.release();
  }
  return comm_make_lanes(cm, nlanes);
}

int modsx_comm_attach(modsx_comm *cm, modsx_ctx *ctx, int lane) {
  if (!cm || !ctx || lane < 0 || lane >= (int)cm->lanes.size()) { mx::set_error("modsx_comm_attach: bad argument"); return MODSX_ERR_ARG; }
  ctx->shardLane = lane;
  return MODSX_OK;
}

int modsx_comm_lane_done(modsx_comm *cm, int lane) {
  if (!cm || lane < 0 || lane >= (int)cm->lanes.size()) { mx::set_error("modsx_comm_lane_done: bad argument"); return MODSX_ERR_ARG; }
  std::lock_guard<std::mutex> lk(cm->mu);
  cm->lanes[lane].retired = true;
  if (cm->turn == lane) turn_advance_locked(cm);
  return MODSX_OK;
}

int modsx_comm_reset_lanes(modsx_comm *cm) {
  if (!cm) { mx::set_error("modsx_comm_reset_lanes: null"); return MODSX_ERR_ARG; }
  std::lock_guard<std::mutex> lk(cm->mu);
  for (ShardLane &L : cm->lanes) L.retired = false;
  cm->turn = 0;
  cm->cv.notify_all();
  return MODSX_OK;
}

int modsx_comm_set_timeout(modsx_comm *cm, int ms) {
  if (!cm || ms < 1) { mx::set_error("modsx_comm_set_timeout: bad argument"); return MODSX_ERR_ARG; }
  cm->timeout_ms.store(ms);
  // a lane may wait for its turn ten times as long as for a collective (a long host stage of the lane ahead is no hang);
  // MODSX_COMM_TURN_TIMEOUT_MS overrides
  cm->turn_timeout_ms.store(getenv("MODSX_COMM_TURN_TIMEOUT_MS") ? std::max(1, atoi(getenv("MODSX_COMM_TURN_TIMEOUT_MS"))) : (ms > 200000000 ? ms : 10 * ms));
  return MODSX_OK;
}

void modsx_comm_destroy(modsx_comm *cm) {
  if (!cm) return;
  hipSetDevice(cm->dev);
  if (cm->nccl) {
    std::lock_guard<std::mutex> lk(cm->issueMu);
    if (cm->aborted) { /* ncclCommAbort already released the communicator */ }
    else if (cm->dead.load()) g_rccl.CommAbort(cm->nccl);
    else g_rccl.CommDestroy(cm->nccl);
    cm->nccl = nullptr;
  }
  if (cm->loop) {
    // this rank's events stay with the group (peers may still be about to wait on them); the group's destructor releases them
    // when the last rank drops its reference
    hipDeviceSynchronize();
    cm->loop->slots[cm->rank].taken = false;
    cm->loop.reset();
  }
  for (ShardLane &L : cm->lanes) {
    DevBuf *bufs[] = {&L.blkLocal, &L.blkAll, &L.regsIn, &L.regsOut, &L.mLocal, &L.mAll, &L.rcDev, &L.order, &L.posOut, &L.ownSend, &L.ownRecv, &L.ownJobs};
    for (DevBuf *b : bufs) b->release();
    L.hRegs.release(); L.hHdr.release(); L.hRc.release(); L.hJobs.release();
  }
  delete cm;
}

int modsx_comm_set_exchange(modsx_comm *cm, int mode) {
  if (!cm || (mode != MODSX_EXCHANGE_ALL_GATHER && mode != MODSX_EXCHANGE_OWNER)) { mx::set_error("modsx_comm_set_exchange: bad argument"); return MODSX_ERR_ARG; }
  cm->exchange_mode.store(mode);
  return MODSX_OK;
}

int modsx_shard_owner_plan(const int *item_counts, int nimages, int nviews, int world, int rank, const int *image_owner, int *sends,
                           int *recvs, int *jobs, int cap, long *n) {
  if (!item_counts || !image_owner || !n || nimages < 1 || nviews < 1 || world < 1 || rank < 0 || rank >= world || cap < 0 ||
      (cap > 0 && (!sends || !recvs || !jobs))) { mx::set_error("modsx_shard_owner_plan: bad argument"); return MODSX_ERR_ARG; }
  for (int j = 0; j < nimages; j++) if (image_owner[j] < 0 || image_owner[j] >= world) { mx::set_error("modsx_shard_owner_plan: an owner is not a rank"); return MODSX_ERR_ARG; }
  for (int f = 0; f < nimages * nviews; f++) if (item_counts[f] < 0) { mx::set_error("modsx_shard_owner_plan: negative count"); return MODSX_ERR_ARG; }
  OwnerPlan P;
  owner_plan(item_counts, nimages, nviews, world, rank, image_owner, P);
  n[0] = (long)P.sends.size(); n[1] = (long)P.recvs.size(); n[2] = (long)P.jobs.size(); n[3] = (long)P.recvRows; n[4] = (long)P.N;
  if ((long)cap < n[0] || (long)cap < n[1] || (long)cap < n[2]) return MODSX_ERR_CAPACITY;
  for (size_t k = 0; k < P.sends.size(); k++) { const OwnerMsg &m = P.sends[k]; int *o = sends + 4 * k; o[0] = m.peer; o[1] = m.image; o[2] = m.row0; o[3] = m.rows; }
  for (size_t k = 0; k < P.recvs.size(); k++) { const OwnerMsg &m = P.recvs[k]; int *o = recvs + 4 * k; o[0] = m.peer; o[1] = m.image; o[2] = m.row0; o[3] = m.rows; }
  for (size_t k = 0; k < P.jobs.size(); k++) { const OwnerJob &jb = P.jobs[k]; int *o = jobs + 4 * k; o[0] = jb.src0; o[1] = jb.dst0; o[2] = jb.n; o[3] = 0; }
  return MODSX_OK;
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

int modsx_comm_stats(const modsx_comm *cm, long *out, int n) {
  if (!cm || !out) { mx::set_error("modsx_comm_stats: null"); return MODSX_ERR_ARG; }
  const long v[] = {cm->collectives, cm->bytes_gathered, cm->retries, cm->agreements, (long)cm->lanes.size(), cm->loop ? 1L : 0L, cm->dead.load() ? 1L : 0L, cm->turn_wait_us.load(),
                    cm->bytes_received, cm->exchanges};
  const int m = (int)(sizeof v / sizeof v[0]);
  for (int i = 0; i < n && i < m; i++) out[i] = v[i];
  return m;
}

/* The wire format of one exchange, stated on the host (what k_pack_rows / k_unpack_blocks do on the device; the GPU tests compare
 * the two byte for byte, the gloo CPU tests run ranks over it without a device):
 *   block = header {magic "MXSH", rc, rows, items, counts[items]} padded to 64 B, then `rows` records of
 *           R + 128 * ndesc bytes (the region part, then the region's descriptor of every class), then padding up to
 *           block_rows records.  row_format MODSX_SHARD_ROW_REGION: R = sizeof(modsx_region), the whole region (the calls that
 *           return region lists); MODSX_SHARD_ROW_KP: R = 56, the doubles x, y, a11, a12, a21, a22, s of its reproj_kp (all the
 *           matcher and the verification read: modsx_match_pairs_views_sharded).
 * modsx_shard_block_bytes: size of a block for `items` (image, view) items, block_rows rows and ndesc descriptor classes.
 * modsx_shard_block_pack:  this rank's block from its regions (item order), descriptors (desc[k]: [n][128] u8 of class k) and
 *                          per-item counts (counts[f] = 0 for items of other ranks); rows beyond block_rows are left out --
 *                          the header still carries the true row count, which is how every rank sees that the block was too
 *                          small.  Returns the bytes written.
 * modsx_shard_blocks_unpack: the reference's list from the `world` gathered blocks: item f sits in the block of rank f mod world
 *                          at that rank's running offset.  regs_out (modsx_region[cap], or double[cap][7] for ROW_KP) /
 *                          desc_out[k]: capacity `cap` regions; item_counts: [items].
 *                          Returns the list length, MODSX_ERR_CAPACITY (with *need_rows = the largest row count) when a block
 *                          was too small, a rank's rc when its header carries one, MODSX_ERR_ARG on a malformed header. */
static bool fmt_ok(int f) { return f == MODSX_SHARD_ROW_REGION || f == MODSX_SHARD_ROW_KP; }
long modsx_shard_block_bytes(int items, int block_rows, int ndesc, int row_format) {
  if (items < 1 || block_rows < 0 || ndesc < 1 || ndesc > MODSX_MAX_DESC || !fmt_ok(row_format)) return MODSX_ERR_ARG;
  return (long)hdr_bytes(items) + (long)block_rows * row_bytes(row_fmt(row_format).regLen, ndesc);
}
long modsx_shard_block_pack(const modsx_region *regs, const unsigned char *const *desc, int ndesc, int n, const int *counts, int items,
                            int rc_local, int block_rows, int row_format, void *block) {
  if (!block || !counts || items < 1 || n < 0 || ndesc < 1 || ndesc > MODSX_MAX_DESC || (n > 0 && (!regs || !desc)) || !fmt_ok(row_format)) {
    mx::set_error("modsx_shard_block_pack: bad argument");
    return MODSX_ERR_ARG;
  }
  const RowFmt rf = row_fmt(row_format);
  const int hdrB = hdr_bytes(items), ROW_B = row_bytes(rf.regLen, ndesc);
  unsigned char *b = (unsigned char *)block;
  memset(b, 0, (size_t)hdrB + (size_t)block_rows * ROW_B);
  int *h = (int *)b;
  h[0] = HDR_MAGIC; h[1] = rc_local; h[2] = rc_local ? 0 : n; h[3] = items;
  if (!rc_local) memcpy(h + HDR_FIXED, counts, (size_t)items * 4);
  const int npack = rc_local ? 0 : std::min(n, block_rows);
  for (int i = 0; i < npack; i++) {
    unsigned char *row = b + hdrB + (size_t)i * ROW_B;
    memcpy(row, (const char *)(regs + i) + rf.regOff, rf.regLen);
    for (int k = 0; k < ndesc; k++) memcpy(row + rf.regLen + 128 * k, desc[k] + (size_t)i * 128, 128);
  }
  return (long)hdrB + (long)block_rows * ROW_B;
}
long modsx_shard_blocks_unpack(const void *blocks, int world, int items, int block_rows, int ndesc, int row_format, void *regs_out,
                               unsigned char *const *desc_out, long cap, int *item_counts, int *need_rows) {
  if (!blocks || world < 1 || items < 1 || ndesc < 1 || ndesc > MODSX_MAX_DESC || !fmt_ok(row_format)) { mx::set_error("modsx_shard_blocks_unpack: bad argument"); return MODSX_ERR_ARG; }
  const RowFmt rf = row_fmt(row_format);
  const int hdrB = hdr_bytes(items), ROW_B = row_bytes(rf.regLen, ndesc);
  const size_t blockB = (size_t)hdrB + (size_t)block_rows * ROW_B;
  const unsigned char *all = (const unsigned char *)blocks;
  int maxrows = 0;
  for (int r = 0; r < world; r++) {
    const int *h = (const int *)(all + r * blockB);
    if (h[0] != HDR_MAGIC || h[3] != items) { mx::set_error("modsx_shard_blocks_unpack: malformed block header"); return MODSX_ERR_ARG; }
    if (h[1]) { mx::set_error("rank " + std::to_string(r) + " reports a failure in its block header"); return h[1]; }
    maxrows = std::max(maxrows, h[2]);
  }
  if (need_rows) *need_rows = maxrows;
  if (maxrows > block_rows) { mx::set_error("modsx_shard_blocks_unpack: a block was too small (every rank sees this and repeats the exchange)"); return MODSX_ERR_CAPACITY; }
  std::vector<int> run(world, 0);
  long j = 0;
  for (int f = 0; f < items; f++) {
    const int r = f % world;
    const int *h = (const int *)(all + r * blockB);
    const int cnt = h[HDR_FIXED + f];
    if (item_counts) item_counts[f] = cnt;
    for (int i = 0; i < cnt; i++, j++) {
      if (j >= cap) { mx::set_error("modsx_shard_blocks_unpack: output capacity"); return MODSX_ERR_CAPACITY; }
      const unsigned char *row = all + r * blockB + hdrB + (size_t)(run[r] + i) * ROW_B;
      if (regs_out) memcpy((char *)regs_out + (size_t)j * rf.regLen, row, rf.regLen);
      for (int k = 0; k < ndesc; k++) if (desc_out && desc_out[k]) memcpy(desc_out[k] + (size_t)j * 128, row + rf.regLen + 128 * k, 128);
    }
    run[r] += cnt;
  }
  return j;
}
/* test hook: the device kernels on host-provided data (needs a device): packs `n` regions + descriptors into rows with
 * k_pack_rows (fed the way the exchange feeds it: the region part of every record, contiguous), or orders `world` gathered blocks
 * with k_unpack_blocks; outputs are copied back to the host */
long modsx_shard_device_pack(modsx_ctx *ctx, const modsx_region *regs, const unsigned char *const *desc, int ndesc, int n, int row_format,
                             void *rows_out) {
  if (!ctx || !regs || !desc || !rows_out || n < 1 || ndesc < 1 || ndesc > MODSX_MAX_DESC || !fmt_ok(row_format)) { mx::set_error("modsx_shard_device_pack: bad argument"); return MODSX_ERR_ARG; }
  hipSetDevice(ctx->dev);
  hipStream_t s = ctx->stream;
  const RowFmt rf = row_fmt(row_format);
  DevBuf dr, dd, dout;
  const int ROW_B = row_bytes(rf.regLen, ndesc);
  if (!dr.ensure((size_t)n * rf.regLen) || !dd.ensure((size_t)n * 128 * ndesc) || !dout.ensure((size_t)n * ROW_B)) return MODSX_ERR_NOMEM;
  std::vector<unsigned char> part((size_t)n * rf.regLen);
  for (int i = 0; i < n; i++) memcpy(part.data() + (size_t)i * rf.regLen, (const char *)(regs + i) + rf.regOff, rf.regLen);
  MX_HIP(hipMemcpyAsync(dr.p, part.data(), part.size(), hipMemcpyHostToDevice, s));
  DescPtrs dp;
  for (int k = 0; k < MODSX_MAX_DESC; k++) dp.p[k] = nullptr;
  for (int k = 0; k < ndesc; k++) {
    dp.p[k] = (unsigned char *)dd.p + (size_t)k * n * 128;
    MX_HIP(hipMemcpyAsync(dp.p[k], desc[k], (size_t)n * 128, hipMemcpyHostToDevice, s));
  }
  hipLaunchKernelGGL(k_pack_rows, dim3((n + 7) / 8), dim3(256), 0, s, (const unsigned char *)dr.p, rf.regLen, dp, ndesc, n, (unsigned char *)dout.p);
  MX_HIP(hipMemcpyAsync(rows_out, dout.p, (size_t)n * ROW_B, hipMemcpyDeviceToHost, s));
  MX_HIP(hipStreamSynchronize(s));
  dr.release(); dd.release(); dout.release();
  return (long)n * ROW_B;
}
long modsx_shard_device_unpack(modsx_ctx *ctx, const void *blocks, int world, int items, int block_rows, int ndesc, int row_format,
                               void *regs_out, unsigned char *const *desc_out, double *pos_out, long cap) {
  if (!ctx || !blocks || !regs_out || !desc_out || world < 1 || world > SHARD_MAXW || items < 1 || items > SHARD_MAXV || ndesc < 1 || ndesc > MODSX_MAX_DESC ||
      !fmt_ok(row_format)) {
    mx::set_error("modsx_shard_device_unpack: bad argument");
    return MODSX_ERR_ARG;
  }
  hipSetDevice(ctx->dev);
  hipStream_t s = ctx->stream;
  const RowFmt rf = row_fmt(row_format);
  const size_t RB = (size_t)rf.regLen;
  const int hdrB = hdr_bytes(items), ROW_B = row_bytes(rf.regLen, ndesc);
  const size_t blockB = (size_t)hdrB + (size_t)block_rows * ROW_B, rowsCap = (size_t)world * block_rows;
  if ((size_t)cap < rowsCap) { mx::set_error("modsx_shard_device_unpack: cap must hold world * block_rows regions"); return MODSX_ERR_ARG; }
  DevBuf din, dregs, ddesc, dpos;
  if (!din.ensure(blockB * world) || !dregs.ensure(rowsCap * RB + 64) || !ddesc.ensure(rowsCap * 128 * ndesc + 64) || !dpos.ensure(rowsCap * 16 + 64)) return MODSX_ERR_NOMEM;
  MX_HIP(hipMemcpyAsync(din.p, blocks, blockB * world, hipMemcpyHostToDevice, s));
  MX_HIP(hipMemsetAsync(dregs.p, 0, rowsCap * RB, s));
  DescPtrs dp;
  for (int k = 0; k < MODSX_MAX_DESC; k++) dp.p[k] = k < ndesc ? (unsigned char *)ddesc.p + (size_t)k * rowsCap * 128 : nullptr;
  hipLaunchKernelGGL(k_unpack_blocks, dim3((unsigned)((rowsCap + 7) / 8)), dim3(256), 0, s, (const unsigned char *)din.p, world, items, block_rows, blockB,
                     hdrB, (unsigned char *)dregs.p, rf.regLen, dp, ndesc, rowsCap, (double *)dpos.p, rf.posOfs);
  MX_HIP(hipMemcpyAsync(regs_out, dregs.p, rowsCap * RB, hipMemcpyDeviceToHost, s));
  for (int k = 0; k < ndesc; k++) MX_HIP(hipMemcpyAsync(desc_out[k], dp.p[k], rowsCap * 128, hipMemcpyDeviceToHost, s));
  if (pos_out) MX_HIP(hipMemcpyAsync(pos_out, dpos.p, rowsCap * 16, hipMemcpyDeviceToHost, s));
  MX_HIP(hipStreamSynchronize(s));
  din.release(); dregs.release(); ddesc.release(); dpos.release();
  return (long)rowsCap;
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
  // the first descriptor class of par's list is the one returned (the orientation mode follows the whole list)
  DescSet ds;
  int rc = resolve_descs(*par, nullptr, ds);
  if (rc) return rc;
  ds.forceHalf = ds.half();
  ds.n = 1;
  DevBuf *acc[1] = {&ctx->descAllU8[0]};
  const size_t base0[1] = {0};
  std::vector<int> counts(nviews, 0);
  rc = detect_describe_views_sharded(ctx, comm, img, views, nviews, *par, ds, r, acc, base0, counts.data());
  if (rc) return rc;
  rebase_ids(r, counts.data(), nviews, 0);
  if (view_counts) memcpy(view_counts, counts.data(), sizeof(int) * nviews);
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

int modsx_match_pairs_views_sharded(modsx_ctx *ctx, modsx_comm *comm, const modsx_image *const *imgs1, const modsx_image *const *imgs2,
                                    int n_pairs, const modsx_view *views, int nviews, const modsx_pair_params *par, int owner_base,
                                    modsx_pair_result *results) {
  if (!ctx || !comm || !imgs1 || !imgs2 || !views || !par || !results || nviews <= 0 || n_pairs < 1 || n_pairs > 16) {
    mx::set_error("modsx_match_pairs_views_sharded: bad argument (1..16 pairs per call)");
    return MODSX_ERR_ARG;
  }
  for (int g = 0; g < n_pairs; g++) if (!imgs1[g] || !imgs2[g]) { mx::set_error("modsx_match_pairs_views_sharded: null image"); return MODSX_ERR_ARG; }
  hipSetDevice(ctx->dev);
  const int rc = match_pairs_views_sharded(ctx, comm, imgs1, imgs2, n_pairs, views, nviews, *par, owner_base, results);
  if (rc) for (int g = 0; g < n_pairs; g++) modsx_pair_result_release(&results[g]);
  return rc ? rc : n_pairs;
}

int modsx_match_ladder_sharded(modsx_ctx *ctx, modsx_comm *comm, const modsx_image *img1, const modsx_image *img2,
                               const modsx_ladder_step *steps, int nsteps, int min_matches, const modsx_pair_params *par,
                               modsx_pair_result *res, int *steps_done) {
  if (!ctx || !comm || !img1 || !img2 || !steps || nsteps < 1 || !par || !res) { mx::set_error("modsx_match_ladder_sharded: bad argument"); return MODSX_ERR_ARG; }
  for (int i = 0; i < nsteps; i++)
    if (!steps[i].views || steps[i].nviews < 1) { mx::set_error("modsx_match_ladder_sharded: a step without views"); return MODSX_ERR_ARG; }
  hipSetDevice(ctx->dev);
  // every rank verifies (owner -1): same tentatives, same seed, same result -- the early exit needs no collective
  return match_ladder(ctx, img1, img2, steps, nsteps, min_matches, *par, res, steps_done, nullptr, comm, -1);
}

}  // extern "C"

// Test hook: enable != 0 puts the in-process stand-in into librccl's function table (the real one, if it was loaded, is kept and
// comes back with enable == 0).  Communicators made under one table must be destroyed under it.
extern "C" __attribute__((visibility("default"))) int modsx_debug_mock_rccl(int enable) {
  using namespace mx;
  std::lock_guard<std::mutex> lk(g_rcclMu);
  if (enable && !g_mockOn) {
    g_rcclSaved = g_rccl;
    RcclApi a;
    a.h = (void *)1;
    a.GetUniqueId = mock_GetUniqueId; a.CommInitRank = mock_CommInitRank; a.CommDestroy = mock_CommDestroy; a.CommAbort = mock_CommAbort;
    a.AllGather = mock_AllGather; a.GetVersion = mock_GetVersion; a.GetErrorString = mock_GetErrorString;
    a.Send = mock_Send; a.Recv = mock_Recv; a.GroupStart = mock_GroupStart; a.GroupEnd = mock_GroupEnd;
    g_rccl = a;
    g_mockOn = true;
  } else if (!enable && g_mockOn) {
    g_rccl = g_rcclSaved;
    g_mockOn = false;
  }
  return MODSX_OK;
}
