// micro-benchmark: what a chain of DEPENDENT v_mfma_i32_32x32x32_i8 (same accumulator) costs on gfx950, back to back and with
// plain VALU instructions between the links, against two interleaved chains.  The loop body is inline assembly, so the
// instruction order is exactly the one written here.
//   T0: chain of 4 on c0, chain of 4 on c1                      (no VALU)
//   T1: two chains interleaved c0 c1 c0 c1 c0 c1 c0 c1            (no VALU)
//   T2: T0 with V independent VALU after every MFMA
//   T3: T1 with V independent VALU after every MFMA
//   T4: T0, all 8V VALU after the eight MFMAs
#include <hip/hip_runtime.h>
#include <cstdio>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));
#define REP 1024
#ifdef RANDOM_DATA
#ifdef SIFT_LIKE
#define RND (AB ^ (rnd() & 0x1f1f1f1f))
#else
#define RND rnd()
#endif
#else
#define RND (seed + 3)
#endif
#define MF0(c, a, b) asm volatile("v_mfma_i32_32x32x32_i8 %0, %1, %2, 0" : "=v"(c) : "v"(a), "v"(b))
#define MF(c, a, b) asm volatile("v_mfma_i32_32x32x32_i8 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b))

template <int V>
__device__ __forceinline__ void valu(int *m) {
#pragma unroll
  for (int v = 0; v < V; v++) {
    const int j = v & 7;
    if (v & 1) asm volatile("v_min3_i32 %0, %0, %1, %2" : "+v"(m[j]) : "v"(m[(j + 3) & 7]), "v"(m[(j + 5) & 7]));
    else asm volatile("v_lshl_add_u32 %0, %1, 9, %2" : "=v"(m[j]) : "v"(m[(j + 2) & 7]), "v"(m[(j + 5) & 7]));
  }
}

template <int T, int V>
__global__ __launch_bounds__(256) void k(int *out, int seed) {
  unsigned h = (threadIdx.x + 977u * blockIdx.x + seed) * 2654435761u; auto rnd = [&]() { h ^= h << 13; h ^= h >> 17; h ^= h << 5; return (int)h; };
  int AB = 0x60606060;
  v4i a = {RND, RND, RND, RND};
  AB = (int)0x80808080;
  v4i b = {RND, RND, RND, RND};
  v16i c0, c1;
  int m[8], acc = 0;
  for (int i = 0; i < 8; i++) m[i] = threadIdx.x + i;
  for (int i = 0; i < REP; i++) {
    if (T == 0 || T == 2 || T == 4) {
      MF0(c0, a, b); if (T == 2) valu<V>(m);
      MF(c0, b, a); if (T == 2) valu<V>(m);
      MF(c0, a, a); if (T == 2) valu<V>(m);
      MF(c0, b, b); if (T == 2) valu<V>(m);
      MF0(c1, a, b); if (T == 2) valu<V>(m);
      MF(c1, b, a); if (T == 2) valu<V>(m);
      MF(c1, a, a); if (T == 2) valu<V>(m);
      MF(c1, b, b); if (T == 2) valu<V>(m);
      if (T == 4) valu<8 * V>(m);
    } else {
      MF0(c0, a, b); if (T == 3) valu<V>(m);
      MF0(c1, a, b); if (T == 3) valu<V>(m);
      MF(c0, b, a); if (T == 3) valu<V>(m);
      MF(c1, b, a); if (T == 3) valu<V>(m);
      MF(c0, a, a); if (T == 3) valu<V>(m);
      MF(c1, a, a); if (T == 3) valu<V>(m);
      MF(c0, b, b); if (T == 3) valu<V>(m);
      MF(c1, b, b); if (T == 3) valu<V>(m);
    }
  }
  asm volatile("s_nop 15\n s_nop 15\n s_nop 15" ::: "memory");
  for (int r = 0; r < 16; r++) acc += c0[r] ^ c1[r];
  for (int i = 0; i < 8; i++) acc += m[i];
  out[blockIdx.x * 256 + threadIdx.x] = acc;
}

template <int T, int V> void run(int wgPerCU) {
  int *o; (void)hipMalloc(&o, 256 * 256 * 8 * 4);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  const int blocks = 256 * wgPerCU;
  hipLaunchKernelGGL((k<T, V>), dim3(blocks), dim3(256), 0, 0, o, 3);
  (void)hipEventRecord(e0);
  hipLaunchKernelGGL((k<T, V>), dim3(blocks), dim3(256), 0, 0, o, 3);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  const double mfmas = (double)blocks * 4 * REP * 8;
  const double ops = mfmas * 2.0 * 32.0 * 32 * 32;
  printf("T%d  VALU/MFMA %d  waves/SIMD %d: %.3f ms  %.0f TOP/s  (%.1f cycles per MFMA per SIMD at 2.4 GHz)\n",
         T, V, wgPerCU, ms, ops / ms / 1e9, ms * 1e-3 * 2.4e9 / (mfmas / 1024.0));
  (void)hipFree(o);
}
int main() {
  for (int w : {1, 2, 3}) { run<0, 0>(w); run<1, 0>(w); }
  for (int w : {1, 2, 3}) { run<2, 1>(w); run<2, 4>(w); run<2, 7>(w); run<3, 1>(w); run<3, 4>(w); run<3, 7>(w); run<4, 4>(w); run<4, 7>(w); }
  return 0;
}
