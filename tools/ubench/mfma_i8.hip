// micro-benchmark: int8 MFMA issue rate on gfx950 and how much plain VALU work hides beside it.
//   shape 0: v_mfma_i32_32x32x32_i8      shape 1: v_mfma_i32_16x16x64_i8
//   VPM = plain VALU instructions (v_lshl_add / v_min3 mix) issued per MFMA by the same wave
// Reports TOP/s (2 * M * N * K per MFMA) for 1, 2, 4 waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));
#define REP 2048

template <int SHAPE, int VPM>
__global__ __launch_bounds__(256) void k(int *out, int seed) {
  v4i a = {seed, seed + 1, seed + 2, seed + 3}, b = {seed * 3, seed * 5, seed * 7, seed * 11};
  v16i c0 = {0}, c1 = {0};
  v4i e0 = {0}, e1 = {0}, e2 = {0}, e3 = {0};
  int m[8];
  for (int i = 0; i < 8; i++) m[i] = threadIdx.x + i;
  for (int i = 0; i < REP; i++) {
    if (SHAPE == 0) {
      c0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, c0, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(b, a, c1, 0, 0, 0);
    } else {
      e0 = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, e0, 0, 0, 0);
      e1 = __builtin_amdgcn_mfma_i32_16x16x64_i8(b, a, e1, 0, 0, 0);
      e2 = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, a, e2, 0, 0, 0);
      e3 = __builtin_amdgcn_mfma_i32_16x16x64_i8(b, b, e3, 0, 0, 0);
    }
    // exactly VPM VALU instructions per MFMA, issued by the same wave, independent of the accumulators; asm volatile
    // keeps the compiler from re-associating them (each instruction reads two and writes one of 8 rotating registers)
    constexpr int NV = VPM * (SHAPE == 0 ? 2 : 4);
#pragma unroll
    for (int v = 0; v < NV; v++) {
      const int j = v & 7;
      if (v & 1) asm volatile("v_min3_i32 %0, %0, %1, %2" : "+v"(m[j]) : "v"(m[(j + 3) & 7]), "v"(m[(j + 5) & 7]));
      else asm volatile("v_lshl_add_u32 %0, %1, 9, %2" : "=v"(m[j]) : "v"(m[(j + 2) & 7]), "v"(m[(j + 5) & 7]));
    }
  }
  int s = 0;
  for (int i = 0; i < 16; i++) s += c0[i] + c1[i];
  for (int i = 0; i < 4; i++) s += e0[i] + e1[i] + e2[i] + e3[i];
  for (int i = 0; i < 8; i++) s += m[i];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int SHAPE, int VPM> void run(int wgPerCU) {
  int *o; hipMalloc(&o, 256 * 256 * 8 * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int blocks = 256 * wgPerCU;
  hipLaunchKernelGGL((k<SHAPE, VPM>), dim3(blocks), dim3(256), 0, 0, o, 3);
  hipEventRecord(e0);
  hipLaunchKernelGGL((k<SHAPE, VPM>), dim3(blocks), dim3(256), 0, 0, o, 3);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double mfmas = (double)blocks * 4 * REP * (SHAPE == 0 ? 2 : 4);
  const double ops = mfmas * 2.0 * (SHAPE == 0 ? 32.0 * 32 * 32 : 16.0 * 16 * 64);
  printf("shape %s  VALU/MFMA %2d  waves/SIMD %d: %.3f ms  %.0f TOP/s  (%.1f cycles per MFMA per SIMD at 2.4 GHz)\n",
         SHAPE == 0 ? "32x32x32" : "16x16x64", VPM, wgPerCU, ms, ops / ms / 1e9,
         ms * 1e-3 * 2.4e9 / (mfmas / 1024.0));
  hipFree(o);
}
int main() {
  for (int w : {1, 2, 4}) { run<0, 0>(w); run<1, 0>(w); }
  for (int w : {1, 2, 4}) { run<0, 4>(w); run<0, 6>(w); run<0, 7>(w); run<0, 8>(w); run<0, 12>(w); }
  for (int w : {2, 4}) { run<1, 2>(w); run<1, 3>(w); run<1, 4>(w); }
  return 0;
}
