// hostprof.cpp -- a sampling profile of the HOST side of the pipeline (debug aid, off unless started): where do the CPU-seconds per
// pair go (bench.py: host_cpu_s_per_pair_rank0)?  ITIMER_PROF counts the process' CPU time over all threads and delivers SIGPROF
// to a thread that is running; the handler keeps the call chain (backtrace) in a preallocated table.  modsx_debug_sampler(0, path)
// stops and writes, per sample bucket, "count module offset" lines for (a) the innermost frame and (b) the innermost frame inside
// libmodsx.so -- tools/host_sampler.py resolves the offsets with the libraries' symbol tables (nm) and prints the tables.
#include <dlfcn.h>
#include <execinfo.h>
#include <signal.h>
#include <stdio.h>
#include <string.h>
#include <sys/time.h>
#include <atomic>
#include <map>
#include <string>
#include <vector>

namespace {
constexpr int MAXS = 1 << 16, DEPTH = 64;
void **g_frames = nullptr;             // MAXS x DEPTH
unsigned char *g_depth = nullptr;
std::atomic<int> g_n(0);
struct sigaction g_old;
void on_prof(int) {
  const int i = g_n.fetch_add(1);
  if (i >= MAXS) return;
  const int d = backtrace(g_frames + (size_t)i * DEPTH, DEPTH);
  g_depth[i] = (unsigned char)d;
}
}  // namespace

extern "C" __attribute__((visibility("default"))) int modsx_debug_sampler(int start, const char *path, int period_us) {
  if (start) {
    if (!g_frames) { g_frames = (void **)calloc((size_t)MAXS * DEPTH, sizeof(void *)); g_depth = (unsigned char *)calloc(MAXS, 1); }
    if (!g_frames || !g_depth) return -1;
    void *warm[4];
    backtrace(warm, 4);                  // loads libgcc's unwinder outside the handler
    g_n = 0;
    struct sigaction sa;
    memset(&sa, 0, sizeof sa);
    sa.sa_handler = on_prof;
    sa.sa_flags = SA_RESTART;
    sigaction(SIGPROF, &sa, &g_old);
    itimerval tv;
    tv.it_interval.tv_sec = 0; tv.it_interval.tv_usec = period_us > 0 ? period_us : 1000;
    tv.it_value = tv.it_interval;
    return setitimer(ITIMER_PROF, &tv, nullptr);
  }
  itimerval off;
  memset(&off, 0, sizeof off);
  setitimer(ITIMER_PROF, &off, nullptr);
  sigaction(SIGPROF, &g_old, nullptr);
  const int n = g_n.load() < MAXS ? g_n.load() : MAXS;
  if (!path) return n;
  Dl_info self;
  const char *selfName = dladdr((void *)&modsx_debug_sampler, &self) ? self.dli_fname : "";
  std::map<std::string, long> leaf, inlib, mods, rt, chains;      // rt: innermost libmodsx frame of the samples whose innermost frame is NOT in libmodsx
  for (int i = 0; i < n; i++) {
    bool haveLeaf = false, haveLib = false, leafInLib = false;
    std::string leafMod, chainStr, lastMod;
    for (int f = 0; f < g_depth[i]; f++) {
      void *pc = g_frames[(size_t)i * DEPTH + f];
      Dl_info di;
      if (!dladdr(pc, &di) || !di.dli_fname) continue;
      if (f < 2) continue;               // [0] the handler, [1] libc's signal return trampoline, [2] the interrupted pc
      {   // the chain of modules from the innermost frame outwards (consecutive repeats folded): which kind of thread was this
        const char *b = strrchr(di.dli_fname, '/');
        const std::string mod = b ? b + 1 : di.dli_fname;
        if (mod != lastMod) { chainStr += (chainStr.empty() ? "" : " < ") + mod; lastMod = mod; }
      }
      char key[512];
      snprintf(key, sizeof key, "%s %lx", di.dli_fname, (unsigned long)((char *)pc - (char *)di.dli_fbase));
      if (!haveLeaf) { leaf[key]++; mods[di.dli_fname]++; haveLeaf = true; leafInLib = !strcmp(di.dli_fname, selfName); leafMod = di.dli_fname; }
      if (!haveLib && !strcmp(di.dli_fname, selfName)) {
        inlib[key]++; haveLib = true;
        if (!leafInLib) rt[std::string(key) + " <- " + leafMod.substr(leafMod.rfind('/') + 1)]++;
      }
    }
    chains[chainStr]++;
    if (!haveLib) { inlib["(outside-libmodsx) 0"]++; rt["(outside-libmodsx) 0 <- " + leafMod.substr(leafMod.rfind('/') + 1)]++; }
  }
  FILE *fp = fopen(path, "w");
  if (!fp) return -1;
  fprintf(fp, "samples %d\n", n);
  for (auto &m : mods) fprintf(fp, "M %ld %s\n", m.second, m.first.c_str());
  for (auto &m : leaf) fprintf(fp, "L %ld %s\n", m.second, m.first.c_str());
  for (auto &m : inlib) fprintf(fp, "I %ld %s\n", m.second, m.first.c_str());
  for (auto &m : rt) fprintf(fp, "R %ld %s\n", m.second, m.first.c_str());
  for (auto &m : chains) fprintf(fp, "C %ld %s\n", m.second, m.first.c_str());
  fclose(fp);
  return n;
}
