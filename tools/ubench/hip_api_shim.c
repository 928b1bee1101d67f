// hip_api_shim.c -- LD_PRELOAD shim: wall time and calls of the HIP entry points the engine uses on its hot path, summed over all
// threads, printed at exit (gcc -O2 -shared -fPIC hip_api_shim.c -o hip_api_shim.so -ldl; LD_PRELOAD=... python bench.py ...).
// What the calls cost the CALLING threads in wall time (lock waits included), which the thread CPU clocks do not show.
#define _GNU_SOURCE
#include <dlfcn.h>
#include <stdio.h>
#include <stdlib.h>
#include <time.h>
#include <stdatomic.h>
typedef int (*fn_launch)(const void *, unsigned long long, unsigned long long, unsigned long long, unsigned long long, void **, size_t, void *);
static _Atomic long ns[8], cnt[8];
static const char *names[8] = {"hipLaunchKernel", "hipMemcpyAsync", "hipStreamSynchronize", "hipMemsetAsync", "hipEventRecord", "hipEventSynchronize", "hipGetLastError", "hipExtLaunchKernel"};
static inline long now(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec * 1000000000L + t.tv_nsec; }
static void *sym(const char *n) {
  static void *h;
  if (!h) h = dlopen("libamdhip64.so", RTLD_LAZY | RTLD_GLOBAL);      // (the library comes in later than this shim: RTLD_NEXT does not see it)
  void *p = h ? dlvsym(h, n, "hip_4.2") : NULL;
  if (!p && h) p = dlsym(h, n);
  if (!p) { fprintf(stderr, "shim: no %s\n", n); abort(); }
  return p;
}
__attribute__((destructor)) static void report(void) {
  for (int i = 0; i < 8; i++) if (cnt[i]) fprintf(stderr, "hip_api_shim: %-22s %10ld calls %10.1f ms wall %8.2f us/call\n", names[i], (long)cnt[i], ns[i] / 1e6, ns[i] / 1e3 / cnt[i]);
}
typedef struct { unsigned x, y, z; } dim3_;
int hipLaunchKernel(const void *f, dim3_ g, dim3_ b, void **args, size_t shm, void *s) {
  static int (*real)(const void *, dim3_, dim3_, void **, size_t, void *);
  if (!real) real = sym("hipLaunchKernel");
  const long t = now(); const int r = real(f, g, b, args, shm, s); ns[0] += now() - t; cnt[0]++; return r;
}
int hipMemcpyAsync(void *d, const void *s_, size_t n, int k, void *s) {
  static int (*real)(void *, const void *, size_t, int, void *);
  if (!real) real = sym("hipMemcpyAsync");
  const long t = now(); const int r = real(d, s_, n, k, s); ns[1] += now() - t; cnt[1]++; return r;
}
int hipStreamSynchronize(void *s) {
  static int (*real)(void *);
  if (!real) real = sym("hipStreamSynchronize");
  const long t = now(); const int r = real(s); ns[2] += now() - t; cnt[2]++; return r;
}
int hipMemsetAsync(void *d, int v, size_t n, void *s) {
  static int (*real)(void *, int, size_t, void *);
  if (!real) real = sym("hipMemsetAsync");
  const long t = now(); const int r = real(d, v, n, s); ns[3] += now() - t; cnt[3]++; return r;
}
int hipEventRecord(void *e, void *s) {
  static int (*real)(void *, void *);
  if (!real) real = sym("hipEventRecord");
  const long t = now(); const int r = real(e, s); ns[4] += now() - t; cnt[4]++; return r;
}
int hipEventSynchronize(void *e) {
  static int (*real)(void *);
  if (!real) real = sym("hipEventSynchronize");
  const long t = now(); const int r = real(e); ns[5] += now() - t; cnt[5]++; return r;
}
