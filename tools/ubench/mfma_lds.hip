// micro-benchmark: what LDS reads cost beside v_mfma_i32_32x32x32_i8 on gfx950.  Loop body in inline assembly:
// 8 MFMAs (two chains of 4), V plain VALU after each, and L ds_read of width W issued in front of every group of 4 MFMAs
// (addresses conflict-free: lane * W bytes), waited for at the end of the iteration.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v2i __attribute__((ext_vector_type(2)));
typedef int v16i __attribute__((ext_vector_type(16)));
#define REP 1024
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
template <int W, int L>
__device__ __forceinline__ void lds(unsigned addr, v4i *r) {
#pragma unroll
  for (int i = 0; i < L; i++) {
    if (W == 16) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(r[i]) : "v"(addr), "n"(i * 1024));
    if (W == 8) asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(*(v2i *)&r[i]) : "v"(addr), "n"(i * 1024));
    if (W == 4) asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(r[i][0]) : "v"(addr), "n"(i * 1024));
  }
}

template <int V, int W, int L>
__global__ __launch_bounds__(256) void k(int *out, int seed) {
  __shared__ int sm[8192];
  for (int i = threadIdx.x; i < 8192; i += 256) sm[i] = i * seed;
  __syncthreads();
  v4i a = {seed, seed + 1, seed + 2, seed + 3}, b = {seed * 3, seed * 5, seed * 7, seed * 11};
  v16i c0, c1;
  v4i r0[4], r1[4];
  for (int i = 0; i < 4; i++) { r0[i] = a; r1[i] = b; }
  int m[8], acc = 0;
  for (int i = 0; i < 8; i++) m[i] = threadIdx.x + i;
  const unsigned addr = (threadIdx.x & 63) * W;
  for (int i = 0; i < REP; i++) {
    lds<W, L>(addr, r0);
    MF0(c0, a, b); valu<V>(m);
    MF(c0, b, a); valu<V>(m);
    MF(c0, a, a); valu<V>(m);
    MF(c0, b, b); valu<V>(m);
    lds<W, L>(addr + 4096, r1);
    MF0(c1, a, b); valu<V>(m);
    MF(c1, b, a); valu<V>(m);
    MF(c1, a, a); valu<V>(m);
    MF(c1, b, b); valu<V>(m);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
  asm volatile("s_nop 15\n s_nop 15\n s_nop 15" ::: "memory");
  for (int r = 0; r < 16; r++) acc += c0[r] ^ c1[r];
  for (int i = 0; i < 8; i++) acc += m[i];
  for (int i = 0; i < 4; i++) acc += r0[i][0] + r1[i][0];
  out[blockIdx.x * 256 + threadIdx.x] = acc;
}

template <int V, int W, int L> void run(int wgPerCU) {
  int *o; (void)hipMalloc(&o, 256 * 256 * 8 * 4);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  const int blocks = 256 * wgPerCU;
  hipLaunchKernelGGL((k<V, W, L>), dim3(blocks), dim3(256), 0, 0, o, 3);
  (void)hipEventRecord(e0);
  hipLaunchKernelGGL((k<V, W, L>), dim3(blocks), dim3(256), 0, 0, o, 3);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  const double mfmas = (double)blocks * 4 * REP * 8;
  const double ops = mfmas * 2.0 * 32.0 * 32 * 32;
  printf("VALU/MFMA %d  ds_read_b%-3d x %d per 4 MFMA  waves/SIMD %d: %.3f ms  %.0f TOP/s  (%.1f cycles per MFMA per SIMD at 2.4 GHz)\n",
         V, W * 8, L, wgPerCU, ms, ops / ms / 1e9, ms * 1e-3 * 2.4e9 / (mfmas / 1024.0));
  (void)hipFree(o);
}
int main() {
  for (int w : {1, 3}) {
    run<0, 16, 0>(w); run<0, 16, 1>(w); run<0, 16, 2>(w); run<0, 16, 4>(w);
    run<0, 8, 4>(w); run<0, 4, 4>(w);
    run<6, 16, 0>(w); run<6, 16, 2>(w); run<6, 16, 4>(w); run<6, 8, 4>(w); run<6, 4, 4>(w);
    run<3, 16, 2>(w); run<3, 16, 4>(w);
  }
  return 0;
}
