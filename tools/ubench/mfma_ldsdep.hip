// micro-benchmark: v_mfma_i32_32x32x32_i8 whose A operands come from LDS (ds_read_b128), as in a tile sweep.
// Per group: wait for the group's 4 fragments, issue the next group's reads (L of them, 4 are fragments, the rest dummies),
// 4 dependent MFMAs (+V VALU each).  DIST = how many groups ahead the reads are issued (1 or 2 register sets in flight).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));
#define REP 1024
#define MF0(c, a, b) asm volatile("v_mfma_i32_32x32x32_i8 %0, %1, %2, 0" : "=v"(c) : "v"(a), "v"(b))
#define MF(c, a, b) asm volatile("v_mfma_i32_32x32x32_i8 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b))
#define DSR(dst, addr, off) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(off))

template <int V>
__device__ __forceinline__ void valu(int *m) {
#pragma unroll
  for (int v = 0; v < V; v++) {
    const int j = v & 7;
    if (v & 1) asm volatile("v_min3_i32 %0, %0, %1, %2" : "+v"(m[j]) : "v"(m[(j + 3) & 7]), "v"(m[(j + 5) & 7]));
    else asm volatile("v_lshl_add_u32 %0, %1, 9, %2" : "=v"(m[j]) : "v"(m[(j + 2) & 7]), "v"(m[(j + 5) & 7]));
  }
}

template <int V, int L, int DIST>
__global__ __launch_bounds__(256) void k(int *out, int seed) {
  __shared__ int sm[8192];
  for (int i = threadIdx.x; i < 8192; i += 256) sm[i] = i * seed;
  __syncthreads();
  v4i b = {seed * 3, seed * 5, seed * 7, seed * 11};
  v16i c0, c1;
  v4i f[3][4], d[4];
  int m[8], acc = 0;
  for (int i = 0; i < 8; i++) m[i] = threadIdx.x + i;
  const unsigned addr = (threadIdx.x & 63) * 16;
  auto reads = [&](v4i *dst) {
    DSR(dst[0], addr, 0); DSR(dst[1], addr, 1024); DSR(dst[2], addr, 2048); DSR(dst[3], addr, 3072);
    if (L > 4) { DSR(d[0], addr, 4096); DSR(d[1], addr, 5120); DSR(d[2], addr, 6144); DSR(d[3], addr, 7168); }
  };
  reads(f[0]);
  if (DIST == 2) reads(f[1]);
  // 6 groups per iteration so that the register set index is static for DIST 1 (2 sets) and 2 (3 sets)
  for (int i = 0; i < REP; i++) {
#pragma unroll
    for (int g = 0; g < 6; g++) {
      constexpr int NS = DIST + 1;
      v4i *cur = f[g % NS], *nxt = f[(g + DIST) % NS];
      if (DIST == 1) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      else { if (L > 4) asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory"); else asm volatile("s_waitcnt lgkmcnt(4)" ::: "memory"); }
      reads(nxt);
      v16i &c = (g & 1) ? c1 : c0;
      MF0(c, cur[0], b); valu<V>(m);
      MF(c, cur[1], b); valu<V>(m);
      MF(c, cur[2], b); valu<V>(m);
      MF(c, cur[3], b); valu<V>(m);
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)\n s_nop 15\n s_nop 15\n s_nop 15" ::: "memory");
  for (int r = 0; r < 16; r++) acc += c0[r] ^ c1[r];
  for (int i = 0; i < 8; i++) acc += m[i];
  for (int i = 0; i < 4; i++) acc += d[i][0] + f[0][i][1] + f[1][i][2] + f[2][i][3];
  out[blockIdx.x * 256 + threadIdx.x] = acc;
}

template <int V, int L, int DIST> void run(int wgPerCU) {
  int *o; (void)hipMalloc(&o, 256 * 256 * 8 * 4);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  const int blocks = 256 * wgPerCU;
  hipLaunchKernelGGL((k<V, L, DIST>), dim3(blocks), dim3(256), 0, 0, o, 3);
  (void)hipEventRecord(e0);
  hipLaunchKernelGGL((k<V, L, DIST>), dim3(blocks), dim3(256), 0, 0, o, 3);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  const double mfmas = (double)blocks * 4 * REP * 24;
  const double ops = mfmas * 2.0 * 32.0 * 32 * 32;
  printf("VALU/MFMA %d  reads/group %d  distance %d  waves/SIMD %d: %.3f ms  %.0f TOP/s  (%.1f cycles per MFMA per SIMD at 2.4 GHz)\n",
         V, L, DIST, wgPerCU, ms, ops / ms / 1e9, ms * 1e-3 * 2.4e9 / (mfmas / 1024.0));
  (void)hipFree(o);
}
int main() {
  for (int w : {1, 2, 3}) {
    run<0, 4, 1>(w); run<0, 8, 1>(w); run<0, 4, 2>(w); run<0, 8, 2>(w);
    run<6, 4, 1>(w); run<6, 8, 1>(w); run<6, 4, 2>(w); run<6, 8, 2>(w);
  }
  return 0;
}
