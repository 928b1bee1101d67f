// host_sync.hip -- what a stage boundary costs the HOST: T threads, each with its own stream, loop { K small kernels; results to
// the host; wait }.  Reports process CPU per iteration split into the T calling threads and "everything else" (the HIP / HSA
// runtime's own threads), for the ways of moving results and of waiting:
//   mode 0  hipMemcpyAsync D2H (pinned) + hipStreamSynchronize               (what the engine does at every stage boundary)
//   mode 1  the kernel writes its results into mapped pinned memory + hipStreamSynchronize
//   mode 2  mode 1, but the wait is a poll of a sequence word the last kernel writes (nanosleep between looks), no runtime call
//   mode 3  hipMemcpyAsync D2H + hipEventRecord + hipEventSynchronize
//   mode 4  H2D hipMemcpyAsync of a job table in front of the kernels, then mode 0
// usage: host_sync [threads=16] [iters=300] [kernels=8] [kernel_us=100]
#include <hip/hip_runtime.h>
#include <sys/resource.h>
#include <sys/syscall.h>
#include <time.h>
#include <unistd.h>
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <vector>

__global__ void k_spin(unsigned *out, unsigned long long cycles, unsigned seq, unsigned *flag) {
  const unsigned long long t0 = wall_clock64();
  while (wall_clock64() - t0 < cycles) {}
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    out[0] = seq;
    if (flag) { __threadfence_system(); __hip_atomic_store(flag, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM); }
  }
}
static double thread_cpu() { timespec ts; clock_gettime(CLOCK_THREAD_CPUTIME_ID, &ts); return ts.tv_sec + 1e-9 * ts.tv_nsec; }
static double proc_cpu() { timespec ts; clock_gettime(CLOCK_PROCESS_CPUTIME_ID, &ts); return ts.tv_sec + 1e-9 * ts.tv_nsec; }
static double wall() { timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec + 1e-9 * ts.tv_nsec; }

int main(int argc, char **argv) {
  const int T = argc > 1 ? atoi(argv[1]) : 16, iters = argc > 2 ? atoi(argv[2]) : 300, K = argc > 3 ? atoi(argv[3]) : 8;
  const double kus = argc > 4 ? atof(argv[4]) : 100.0;
  hipSetDeviceFlags(hipDeviceScheduleBlockingSync);
  hipFree(0);
  const unsigned long long cycles = (unsigned long long)(kus * 100.0);   // wall_clock64: 100 MHz
  {   // does an asynchronous copy call return while the stream is busy?  (wall time of the CALL behind a 2 ms kernel)
    hipStream_t s; hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    unsigned *d, *h, *big, *hbig; hipMalloc(&d, 4096); hipHostMalloc(&h, 4096, hipHostMallocDefault);
    hipMalloc(&big, 4 << 20); hipHostMalloc(&hbig, 4 << 20, hipHostMallocDefault);
    for (int rep = 0; rep < 3; rep++)
      for (size_t bytes : {(size_t)4096, (size_t)65536, (size_t)(2 << 20)}) {
        hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, s, d, 200000ull, 1u, (unsigned *)nullptr);
        double a = wall(); hipMemcpyAsync(big, hbig, bytes, hipMemcpyHostToDevice, s); double b = wall();
        hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, s, d, 100ull, 1u, (unsigned *)nullptr);
        double c = wall(); hipMemcpyAsync(hbig, big, bytes, hipMemcpyDeviceToHost, s); double e = wall();
        hipStreamSynchronize(s);
        if (rep == 2) printf("behind a 2 ms kernel: hipMemcpyAsync H2D of %zu bytes returns after %.1f us, D2H after %.1f us\n", bytes, (b - a) * 1e6, (e - c) * 1e6);
      }
  }
  for (int mode = 0; mode < 5; mode++) {
    std::vector<double> tcpu(T, 0.0);
    std::atomic<int> ready(0);
    std::atomic<bool> go(false);
    double c0 = 0, w0 = 0;
    std::vector<std::thread> th;
    for (int t = 0; t < T; t++)
      th.emplace_back([&, t]() {
        hipStream_t s; hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
        unsigned *d, *h, *hjobs, *djobs; hipMalloc(&d, 4096); hipHostMalloc(&h, 4096, hipHostMallocDefault);
        hipHostMalloc(&hjobs, 65536, hipHostMallocDefault); hipMalloc(&djobs, 65536);
        hipEvent_t ev; hipEventCreateWithFlags(&ev, hipEventDisableTiming | hipEventBlockingSync);
        h[0] = 0; h[16] = 0;
        unsigned *hd = nullptr; hipHostGetDevicePointer((void **)&hd, h, 0);
        // warm
        hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, s, d, cycles, 0u, (unsigned *)nullptr); hipStreamSynchronize(s);
        ready++;
        while (!go.load()) usleep(100);
        const double a = thread_cpu();
        for (int it = 1; it <= iters; it++) {
          if (mode == 4) hipMemcpyAsync(djobs, hjobs, 65536, hipMemcpyHostToDevice, s);
          for (int k = 0; k < K; k++) {
            const bool last = k == K - 1;
            unsigned *out = (last && (mode == 1 || mode == 2)) ? hd : d;
            hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, s, out, cycles, (unsigned)it, (last && mode == 2) ? hd + 16 : (unsigned *)nullptr);
          }
          if (mode == 0 || mode == 3 || mode == 4) hipMemcpyAsync(h, d, 4096, hipMemcpyDeviceToHost, s);
          if (mode == 3) { hipEventRecord(ev, s); hipEventSynchronize(ev); }
          else if (mode == 2) {
            volatile unsigned *f = h + 16;
            timespec nap = {0, 50000};
            while (*f != (unsigned)it) nanosleep(&nap, nullptr);
          } else hipStreamSynchronize(s);
          if (h[0] != (unsigned)it) { fprintf(stderr, "mode %d thread %d iter %d: got %u\n", mode, t, it, h[0]); break; }
        }
        tcpu[t] = thread_cpu() - a;
        hipStreamSynchronize(s);
      });
    while (ready.load() < T) usleep(1000);
    c0 = proc_cpu(); w0 = wall();
    go = true;
    for (auto &x : th) x.join();
    const double c1 = proc_cpu(), w1 = wall();
    double callers = 0; for (double x : tcpu) callers += x;
    const double n = (double)T * iters;
    printf("mode %d: wall %.3f s, %.1f us wall per iteration and thread; CPU per iteration: callers %.1f us, other threads %.1f us, total %.1f us\n",
           mode, w1 - w0, (w1 - w0) / iters * 1e6, callers / n * 1e6, (c1 - c0 - callers) / n * 1e6, (c1 - c0) / n * 1e6);
    fflush(stdout);
  }
  return 0;
}
