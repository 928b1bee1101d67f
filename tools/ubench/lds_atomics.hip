// lds_atomics.hip -- throughput of LDS atomic adds on gfx950 (per CU), by type and address pattern.
// Behind k_describe's order-free gather: 8 f64 adds per pixel into 128 bins.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s\n", hipGetErrorString(e_)); return 1; } } while (0)

template <typename T, int MODE>
__global__ __launch_bounds__(256) void k(const int *pat, T *out, int iters) {
  __shared__ T acc[8 * 129 + 64];
  for (int i = threadIdx.x; i < 8 * 129 + 64; i += 256) acc[i] = T(0);
  __syncthreads();
  int idx[8];
#pragma unroll
  for (int j = 0; j < 8; j++) idx[j] = pat[(blockIdx.x & 7) * 2048 + j * 256 + threadIdx.x];
  T v = T(threadIdx.x + 1);
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int j = 0; j < 8; j++) {
      if (MODE == 0) __hip_atomic_fetch_add(&acc[idx[j]], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      else { T old = __hip_atomic_fetch_add(&acc[idx[j]], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); v += old * T(0); }
    }
  }
  __syncthreads();
  out[blockIdx.x * 256 + threadIdx.x] = acc[threadIdx.x];
}

template <typename T, int MODE>
int run(const char *name, const int *dpat, const char *pname) {
  T *out;
  const int blocks = 256 * 8, iters = 2000;
  CK(hipMalloc(&out, blocks * 256 * sizeof(T)));
  hipEvent_t a, b;
  CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  hipLaunchKernelGGL((k<T, MODE>), dim3(blocks), dim3(256), 0, 0, dpat, out, 10);
  CK(hipEventRecord(a));
  hipLaunchKernelGGL((k<T, MODE>), dim3(blocks), dim3(256), 0, 0, dpat, out, iters);
  CK(hipEventRecord(b));
  CK(hipDeviceSynchronize());
  float ms;
  CK(hipEventElapsedTime(&ms, a, b));
  const double ops = (double)blocks * 256 * iters * 8;   // lane-ops
  // per CU and cycle at 2.1 GHz (nominal; the clock under this load is not measured here)
  printf("%-10s %-28s %8.3f ms  %7.2f G lane-ops/s  = %5.2f lane-ops / CU / ns\n", name, pname, ms, ops / ms * 1e-6, ops / ms * 1e-6 / 256);
  CK(hipFree(out));
  return 0;
}

int main() {
  std::vector<int> pat(8 * 2048);
  int *dpat;
  CK(hipMalloc(&dpat, pat.size() * 4));
  const char *names[] = {"distinct consecutive", "all lanes one address", "random bin, copy = lane & 7, skew", "random bin, one copy",
                         "8 runs of 8 same bin, copy=lane&7", "same bin per 8 lanes, one copy"};
  for (int p = 0; p < 6; p++) {
    unsigned s = 12345;
    for (int g = 0; g < 8; g++) for (int j = 0; j < 8; j++) for (int t = 0; t < 256; t++) {
      s = s * 1664525u + 1013904223u;
      int bin = (s >> 10) & 127, v;
      const int lane = t & 63;
      if (p == 0) v = t + (j & 1) * 256;
      else if (p == 1) v = j;
      else if (p == 2) v = (lane & 7) * 129 + bin;
      else if (p == 3) v = bin;
      else { unsigned h = (g * 131 + j * 17 + (t >> 3)) * 2654435761u; bin = (h >> 12) & 127; v = p == 4 ? (lane & 7) * 129 + bin : bin; }
      pat[g * 2048 + j * 256 + t] = v;
    }
    CK(hipMemcpy(dpat, pat.data(), pat.size() * 4, hipMemcpyHostToDevice));
    if (run<double, 0>("f64", dpat, names[p])) return 1;
    if (run<float, 0>("f32", dpat, names[p])) return 1;
    if (run<unsigned, 0>("u32", dpat, names[p])) return 1;
    if (run<unsigned long long, 0>("u64", dpat, names[p])) return 1;
  }
  return 0;
}
