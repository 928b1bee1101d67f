// micro-benchmark: issue cost (cycles per wave64 instruction per SIMD) of the VALU ops the descriptor kernels lean on
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define REP 4096
template <int OP>
__global__ __launch_bounds__(256) void k(float *out, float a, double d) {
  float x0 = a + threadIdx.x, x1 = x0 + 1.f, x2 = x0 + 2.f, x3 = x0 + 3.f;
  double y0 = d + threadIdx.x, y1 = y0 + 1.0, y2 = y0 + 2.0, y3 = y0 + 3.0;
  for (int i = 0; i < REP; i++) {
    if (OP == 0) { x0 += a; x1 += a; x2 += a; x3 += a; }                                    // v_add_f32
    if (OP == 1) { y0 += d; y1 += d; y2 += d; y3 += d; }                                    // v_add_f64
    if (OP == 2) { y0 *= d; y1 *= d; y2 *= d; y3 *= d; }                                    // v_mul_f64
    if (OP == 3) { y0 = (double)x0 + 1.0; x0 = (float)y0; y1 = (double)x1 + 1.0; x1 = (float)y1; }   // cvt pair + add64 (x2)
    if (OP == 4) { x0 = (float)(int)x0 + a; x1 = (float)(int)x1 + a; x2 = (float)(int)x2 + a; x3 = (float)(int)x3 + a; }  // 2 cvt + add
    if (OP == 5) { x0 = sqrtf(x0); x1 = sqrtf(x1); x2 = sqrtf(x2); x3 = sqrtf(x3); }
    if (OP == 6) { x0 = x0 / a; x1 = x1 / a; x2 = x2 / a; x3 = x3 / a; }
    if (OP == 7) { y0 = y0 / d; y1 = y1 / d; y2 = y2 / d; y3 = y3 / d; }
  }
  out[blockIdx.x * 256 + threadIdx.x] = x0 + x1 + x2 + x3 + (float)(y0 + y1 + y2 + y3);
}
template <int OP> void run(const char *name, int opsPerIter) {
  float *o; hipMalloc(&o, 256 * 2048 * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int blocks = 256 * 8;  // 8 workgroups of 4 waves per CU: 8 waves per SIMD
  hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, o, 1.0001f, 1.0000001);
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, o, 1.0001f, 1.0000001);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  // waves per SIMD = 8; instructions per wave = REP * opsPerIter; cycles at 2.4 GHz
  double cyc = ms * 1e-3 * 2.4e9 / (8.0 * REP * opsPerIter);
  printf("%-28s %.3f ms  ~%.2f cycles per wave-instruction per SIMD (assuming 2.4 GHz, %d ops/iter)\n", name, ms, cyc, opsPerIter);
  hipFree(o);
}
int main() {
  run<0>("v_add_f32 x4", 4); run<1>("v_add_f64 x4", 4); run<2>("v_mul_f64 x4", 4);
  run<3>("cvt64<-32,add64,cvt32<-64 x2", 6); run<4>("cvt i32<-f32, f32<-i32, add x4", 12);
  run<5>("sqrtf x4", 4); run<6>("f32 div x4", 4); run<7>("f64 div x4", 4);
  return 0;
}
