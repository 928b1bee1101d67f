// micro-benchmark: v_mfma_i32_32x32x32_i8 in DEPENDENT chains (the four k-blocks of one 32x32x128 product accumulate into one
// register set) against two interleaved chains, with VPM plain VALU instructions per MFMA issued by the same wave.
//   MODE 0: chain of 4 on acc A, then chain of 4 on acc B (what a one-product-at-a-time sweep does)
//   MODE 1: two chains interleaved A B A B A B A B
//   MODE 2: four chains interleaved (16 MFMAs)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));
#define REP 1024

template <int MODE, int VPM>
__global__ __launch_bounds__(256) void k(int *out, int seed) {
  v4i a[4], b[4];
  for (int i = 0; i < 4; i++) { a[i] = (v4i){seed + i, seed + 1, seed + 2, seed + 3}; b[i] = (v4i){seed * 3, seed * 5 + i, seed * 7, seed * 11}; }
  v16i c[4];
  int m[8], acc = 0;
  for (int i = 0; i < 8; i++) m[i] = threadIdx.x + i;
  auto valu = [&](int n) {
#pragma unroll
    for (int v = 0; v < n; v++) {
      const int j = v & 7;
      if (v & 1) asm volatile("v_min3_i32 %0, %0, %1, %2" : "+v"(m[j]) : "v"(m[(j + 3) & 7]), "v"(m[(j + 5) & 7]));
      else asm volatile("v_lshl_add_u32 %0, %1, 9, %2" : "=v"(m[j]) : "v"(m[(j + 2) & 7]), "v"(m[(j + 5) & 7]));
    }
  };
  for (int i = 0; i < REP; i++) {
    // opaque to the compiler: the operands may have changed, so no product is loop-invariant
#pragma unroll
    for (int q = 0; q < 4; q++) { asm volatile("" : "+v"(a[q])); asm volatile("" : "+v"(b[q])); }
    constexpr int NCH = MODE == 0 ? 1 : (MODE == 1 ? 2 : 4);
#pragma unroll
    for (int rep = 0; rep < 4 / NCH; rep++) {
      asm volatile("" : "+v"(a[0]));   // every chain is a different product to the compiler
#pragma unroll
      for (int ch = 0; ch < NCH; ch++) c[ch] = (v16i){0};
#pragma unroll
      for (int kb = 0; kb < 4; kb++)
#pragma unroll
        for (int ch = 0; ch < NCH; ch++) {
          c[ch] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[kb], b[(kb + ch) & 3], c[ch], 0, 0, 0);
          valu(VPM);
        }
#pragma unroll
      for (int ch = 0; ch < NCH; ch++) { int t = c[ch][0]; for (int r = 1; r < 16; r++) t ^= c[ch][r]; acc += t; }
    }
  }
  int s = acc;
  for (int i = 0; i < 8; i++) s += m[i];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int MODE, int VPM> void run(int wgPerCU) {
  int *o; hipMalloc(&o, 256 * 256 * 8 * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int blocks = 256 * wgPerCU;
  hipLaunchKernelGGL((k<MODE, VPM>), dim3(blocks), dim3(256), 0, 0, o, 3);
  hipEventRecord(e0);
  hipLaunchKernelGGL((k<MODE, VPM>), dim3(blocks), dim3(256), 0, 0, o, 3);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double mfmas = (double)blocks * 4 * REP * 16;
  const double ops = mfmas * 2.0 * 32.0 * 32 * 32;
  printf("mode %d  VALU/MFMA %2d  waves/SIMD %d: %.3f ms  %.0f TOP/s  (%.1f cycles per MFMA per SIMD at 2.4 GHz)\n",
         MODE, VPM, wgPerCU, ms, ops / ms / 1e9, ms * 1e-3 * 2.4e9 / (mfmas / 1024.0));
  hipFree(o);
}
int main() {
  for (int w : {1, 2, 3}) { run<0, 0>(w); run<1, 0>(w); run<2, 0>(w); }
  for (int w : {1, 2, 3}) { run<0, 6>(w); run<1, 6>(w); run<2, 6>(w); run<0, 7>(w); run<1, 7>(w); }
  return 0;
}
