// graph_seq.hip -- a launch-bound sequence (the scale-space pyramid of a view set: ~30 dependent launches, many of a few microseconds)
// issued launch by launch against the same sequence as an instantiated hipGraph: wall time per sequence and CPU of the calling
// threads, for 1 and 16 threads with a stream each.   usage: graph_seq [kernels=30] [kernel_us=5] [iters=300]
#include <hip/hip_runtime.h>
#include <time.h>
#include <unistd.h>
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <vector>
struct Args { float v[256]; };     // ~1 KB of by-value arguments, as the batched pyramid kernels take
__global__ void k_work(unsigned *out, unsigned long long cycles, Args a) {
  const unsigned long long t0 = wall_clock64();
  while (wall_clock64() - t0 < cycles) {}
  if (threadIdx.x == 0 && blockIdx.x == 0) out[0] += (unsigned)a.v[3];
}
static double wall() { timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec + 1e-9 * ts.tv_nsec; }
static double thread_cpu() { timespec ts; clock_gettime(CLOCK_THREAD_CPUTIME_ID, &ts); return ts.tv_sec + 1e-9 * ts.tv_nsec; }
int main(int argc, char **argv) {
  const int K = argc > 1 ? atoi(argv[1]) : 30, iters = argc > 3 ? atoi(argv[3]) : 300;
  const double kus = argc > 2 ? atof(argv[2]) : 5.0;
  const unsigned long long cycles = (unsigned long long)(kus * 100.0);
  hipSetDeviceFlags(hipDeviceScheduleBlockingSync);
  hipFree(0);
  for (int T : {1, 16})
    for (int mode = 0; mode < 2; mode++) {
      std::vector<double> w(T), cpu(T);
      std::vector<std::thread> th;
      std::atomic<int> ready(0); std::atomic<bool> go(false);
      for (int t = 0; t < T; t++)
        th.emplace_back([&, t]() {
          hipStream_t s; hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
          unsigned *d; hipMalloc(&d, 4096); hipMemset(d, 0, 4096);
          Args a; for (int i = 0; i < 256; i++) a.v[i] = 1.f;
          auto seq = [&]() { for (int k = 0; k < K; k++) hipLaunchKernelGGL(k_work, dim3(64), dim3(256), 0, s, d, cycles, a); };
          hipGraph_t g = nullptr; hipGraphExec_t ge = nullptr;
          if (mode == 1) {
            hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal); seq(); hipStreamEndCapture(s, &g);
            if (hipGraphInstantiate(&ge, g, nullptr, nullptr, 0) != hipSuccess) { fprintf(stderr, "instantiate failed\n"); return; }
          }
          auto run = [&]() { if (mode == 1) hipGraphLaunch(ge, s); else seq(); hipStreamSynchronize(s); };
          run();
          ready++; while (!go.load()) usleep(100);
          const double a0 = wall(), c0 = thread_cpu();
          for (int it = 0; it < iters; it++) run();
          w[t] = wall() - a0; cpu[t] = thread_cpu() - c0;
        });
      while (ready.load() < T) usleep(1000);
      go = true;
      for (auto &x : th) x.join();
      double ws = 0, cs = 0; for (int t = 0; t < T; t++) { ws += w[t]; cs += cpu[t]; }
      printf("%2d thread(s), %s: %.1f us wall per sequence of %d kernels of %.0f us (%.1f us beyond the kernels), %.1f us CPU of the caller\n", T,
             mode ? "graph launch     " : "launch by launch ", ws / T / iters * 1e6, K, kus, ws / T / iters * 1e6 - K * kus, cs / T / iters * 1e6);
      fflush(stdout);
    }
  return 0;
}
