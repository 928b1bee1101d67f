// micro-benchmark: issue cost (cycles per wave64 instruction per SIMD) of VALU forms the compiler emits all over the
// descriptor kernels -- selects on VCC / on an SGPR pair, compares, SGPR operands, packed f32, conversions
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP 4096
#define A4(op) asm volatile(op(0) "\n\t" op(1) "\n\t" op(2) "\n\t" op(3) : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(a), "s"(sa), "s"(sm), "v"(pa), "s"(spa) : "vcc")
// operands: %0-%3 x, %4-%7 p (64-bit pairs), %8 a (vgpr), %9 sa (sgpr), %10 sm (sgpr pair), %11 pa (vgpr pair), %12 spa (sgpr pair)
#define OP_ADD_V(i) "v_add_f32 %" #i ", %" #i ", %8"
#define OP_ADD_S(i) "v_add_f32 %" #i ", %9, %" #i
#define OP_MUL_S(i) "v_mul_f32 %" #i ", %9, %" #i
#define OP_CND_VCC(i) "v_cndmask_b32 %" #i ", %" #i ", %8, vcc"
#define OP_CND_S(i) "v_cndmask_b32 %" #i ", %" #i ", %8, %10"
#define OP_CMP_VCC(i) "v_cmp_lt_f32 vcc, %" #i ", %8"
#define OP_CMP_S(i) "v_cmp_lt_f32 s[20:21], %" #i ", %8"
#define P(i) "%[p" #i "]"
template <int OP>
__global__ __launch_bounds__(256) void k(float *out, float a, unsigned long long m) {
  float x0 = a + threadIdx.x, x1 = x0 + 1.f, x2 = x0 + 2.f, x3 = x0 + 3.f;
  typedef float v2 __attribute__((ext_vector_type(2)));
  v2 p0 = {x0, x1}, p1 = {x1, x2}, p2 = {x2, x3}, p3 = {x3, x0};
  const v2 pa = {a, a};
  const float sa = __builtin_amdgcn_readfirstlane(a);
  const unsigned long long sm = __builtin_amdgcn_readfirstlane((unsigned)m) | ((unsigned long long)__builtin_amdgcn_readfirstlane((unsigned)(m >> 32)) << 32);
  const unsigned long long spa = sm;
  for (int i = 0; i < REP; i++) {
    if (OP == 0) A4(OP_ADD_V);
    if (OP == 1) A4(OP_ADD_S);
    if (OP == 2) A4(OP_MUL_S);
    if (OP == 3) A4(OP_CND_VCC);
    if (OP == 4) A4(OP_CND_S);
    if (OP == 5) A4(OP_CMP_VCC);
    if (OP == 6) asm volatile("v_cmp_lt_f32 s[20:21], %0, %4\n\tv_cmp_lt_f32 s[22:23], %1, %4\n\tv_cmp_lt_f32 s[24:25], %2, %4\n\tv_cmp_lt_f32 s[26:27], %3, %4" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(a) : "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27");
    if (OP == 7) asm volatile("v_pk_add_f32 %0, %0, %4\n\tv_pk_add_f32 %1, %1, %4\n\tv_pk_add_f32 %2, %2, %4\n\tv_pk_add_f32 %3, %3, %4" : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(pa));
    if (OP == 8) asm volatile("v_pk_mul_f32 %0, %4, %0\n\tv_pk_mul_f32 %1, %4, %1\n\tv_pk_mul_f32 %2, %4, %2\n\tv_pk_mul_f32 %3, %4, %3" : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "s"(spa));
    if (OP == 9) asm volatile("v_cvt_i32_f32 %0, %0\n\tv_cvt_i32_f32 %1, %1\n\tv_cvt_i32_f32 %2, %2\n\tv_cvt_i32_f32 %3, %3" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3));
    if (OP == 10) asm volatile("v_cvt_f64_f32 %0, %4\n\tv_cvt_f64_f32 %1, %5\n\tv_cvt_f64_f32 %2, %6\n\tv_cvt_f64_f32 %3, %7" : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(x0), "v"(x1), "v"(x2), "v"(x3));
    if (OP == 11) asm volatile("v_fma_f32 %0, %0, %4, %4\n\tv_fma_f32 %1, %1, %4, %4\n\tv_fma_f32 %2, %2, %4, %4\n\tv_fma_f32 %3, %3, %4, %4" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(a));
    if (OP == 12) asm volatile("v_sqrt_f32 %0, %0\n\tv_sqrt_f32 %1, %1\n\tv_sqrt_f32 %2, %2\n\tv_sqrt_f32 %3, %3" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3));
    if (OP == 13) asm volatile("v_rcp_f32 %0, %0\n\tv_rcp_f32 %1, %1\n\tv_rcp_f32 %2, %2\n\tv_rcp_f32 %3, %3" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3));
    if (OP == 14) asm volatile("v_cndmask_b32 %0, %0, %4, vcc\n\tv_add_f32 %1, %1, %4\n\tv_cndmask_b32 %2, %2, %4, vcc\n\tv_add_f32 %3, %3, %4" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(a) : "vcc");
    if (OP == 15) asm volatile("v_mov_b32 %0, %4\n\tv_mov_b32 %1, %4\n\tv_mov_b32 %2, %4\n\tv_mov_b32 %3, %4" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(a));
    if (OP == 16) asm volatile("v_add_f64 %0, %0, %4\n\tv_add_f64 %1, %1, %4\n\tv_add_f64 %2, %2, %4\n\tv_add_f64 %3, %3, %4" : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(pa));
    if (OP == 17) asm volatile("v_max_f32 %0, %0, %4\n\tv_min_f32 %1, %1, %4\n\tv_max_f32 %2, %2, %4\n\tv_min_f32 %3, %3, %4" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(a));
  }
  out[blockIdx.x * 256 + threadIdx.x] = x0 + x1 + x2 + x3 + p0.x + p1.y + p2.x + p3.y;
}
template <int OP> void run(const char *name) {
  float *o; (void)hipMalloc(&o, 256 * 2048 * 4);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  const int blocks = 256 * 8;  // 8 workgroups of 4 waves per CU: 8 waves per SIMD
  hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, o, 1.0001f, 0x5555aaaa3333ccccull);
  (void)hipEventRecord(e0);
  hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, o, 1.0001f, 0x5555aaaa3333ccccull);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  double cyc = ms * 1e-3 * 2.4e9 / (8.0 * REP * 4);
  printf("%-34s %.3f ms  ~%.2f cycles per wave-instruction per SIMD (assuming 2.4 GHz)\n", name, ms, cyc);
  (void)hipFree(o);
}
int main() {
  run<0>("v_add_f32 v,v,v"); run<1>("v_add_f32 v,s,v"); run<2>("v_mul_f32 v,s,v"); run<3>("v_cndmask_b32 vcc");
  run<4>("v_cndmask_b32_e64 sgpr pair"); run<5>("v_cmp_lt_f32 -> vcc"); run<6>("v_cmp_lt_f32 -> sgpr pairs");
  run<7>("v_pk_add_f32 v,v,v"); run<8>("v_pk_mul_f32 v,s,v"); run<9>("v_cvt_i32_f32"); run<10>("v_cvt_f64_f32");
  run<11>("v_fma_f32"); run<12>("v_sqrt_f32"); run<13>("v_rcp_f32"); run<14>("cndmask vcc / add alternating");
  run<15>("v_mov_b32"); run<16>("v_add_f64"); run<17>("v_max/min_f32");
  return 0;
}
