// micro-benchmark: issue cost (cycles per wave64 instruction per SIMD) of the integer / address operations the compiler
// emits for index arithmetic: v_mul_lo_u32, v_mul_u32_u24, v_mad_u32_u24, v_mad_u64_u32, v_lshl_add_u64, v_mul_hi_u32,
// v_add_u32 -- which of them are full-rate decides how addresses in the hot loops are to be written
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP 4096
template <int OP>
__global__ __launch_bounds__(256) void k(unsigned *out, unsigned a, unsigned long long d) {
  unsigned x0 = a + threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3;
  unsigned long long y0 = d + threadIdx.x, y1 = y0 + 1, y2 = y0 + 2, y3 = y0 + 3;
  for (int i = 0; i < REP; i++) {
    if (OP == 0) { asm volatile("v_add_u32 %0, %0, %4\n\tv_add_u32 %1, %1, %4\n\tv_add_u32 %2, %2, %4\n\tv_add_u32 %3, %3, %4" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(a)); }
    if (OP == 1) { asm volatile("v_mul_lo_u32 %0, %0, %4\n\tv_mul_lo_u32 %1, %1, %4\n\tv_mul_lo_u32 %2, %2, %4\n\tv_mul_lo_u32 %3, %3, %4" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(a)); }
    if (OP == 2) { asm volatile("v_mul_u32_u24 %0, %0, %4\n\tv_mul_u32_u24 %1, %1, %4\n\tv_mul_u32_u24 %2, %2, %4\n\tv_mul_u32_u24 %3, %3, %4" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(a)); }
    if (OP == 3) { asm volatile("v_mad_u32_u24 %0, %0, %4, %1\n\tv_mad_u32_u24 %1, %1, %4, %2\n\tv_mad_u32_u24 %2, %2, %4, %3\n\tv_mad_u32_u24 %3, %3, %4, %0" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(a)); }
    if (OP == 4) { asm volatile("v_mad_u64_u32 %0, vcc, %2, %3, %0\n\tv_mad_u64_u32 %1, vcc, %2, %3, %1" : "+v"(y0), "+v"(y1) : "v"(x0), "v"(a) : "vcc"); }
    if (OP == 5) { asm volatile("v_lshl_add_u64 %0, %0, 2, %4\n\tv_lshl_add_u64 %1, %1, 2, %4\n\tv_lshl_add_u64 %2, %2, 2, %4\n\tv_lshl_add_u64 %3, %3, 2, %4" : "+v"(y0), "+v"(y1), "+v"(y2), "+v"(y3) : "v"(d)); }
    if (OP == 6) { asm volatile("v_mul_hi_u32 %0, %0, %4\n\tv_mul_hi_u32 %1, %1, %4\n\tv_mul_hi_u32 %2, %2, %4\n\tv_mul_hi_u32 %3, %3, %4" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(a)); }
    if (OP == 7) { asm volatile("v_add_lshl_u32 %0, %0, %4, 2\n\tv_add_lshl_u32 %1, %1, %4, 2\n\tv_add_lshl_u32 %2, %2, %4, 2\n\tv_add_lshl_u32 %3, %3, %4, 2" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(a)); }
    if (OP == 8) { asm volatile("v_cndmask_b32 %0, %0, %4, vcc\n\tv_cndmask_b32 %1, %1, %4, vcc\n\tv_cndmask_b32 %2, %2, %4, vcc\n\tv_cndmask_b32 %3, %3, %4, vcc" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(a) : "vcc"); }
    if (OP == 9) { asm volatile("v_med3_i32 %0, %0, 0, %4\n\tv_med3_i32 %1, %1, 0, %4\n\tv_med3_i32 %2, %2, 0, %4\n\tv_med3_i32 %3, %3, 0, %4" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(a)); }
  }
  out[blockIdx.x * 256 + threadIdx.x] = x0 + x1 + x2 + x3 + (unsigned)(y0 + y1 + y2 + y3);
}
template <int OP> void run(const char *name, int opsPerIter) {
  unsigned *o; hipMalloc(&o, 256 * 2048 * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int blocks = 256 * 8;  // 8 workgroups of 4 waves per CU: 8 waves per SIMD
  hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, o, 3u, 5ull);
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, o, 3u, 5ull);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  double cyc = ms * 1e-3 * 2.4e9 / (8.0 * REP * opsPerIter);
  printf("%-20s %.3f ms  ~%.2f cycles per wave-instruction per SIMD (assuming 2.4 GHz)\n", name, ms, cyc);
  hipFree(o);
}
int main() {
  run<0>("v_add_u32", 4); run<1>("v_mul_lo_u32", 4); run<2>("v_mul_u32_u24", 4); run<3>("v_mad_u32_u24", 4);
  run<4>("v_mad_u64_u32", 2); run<5>("v_lshl_add_u64", 4); run<6>("v_mul_hi_u32", 4); run<7>("v_add_lshl_u32", 4);
  run<8>("v_cndmask_b32", 4); run<9>("v_med3_i32", 4);
  return 0;
}
