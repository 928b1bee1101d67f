// wave_simd.hip -- which SIMD does wave w of a workgroup land on?  (gfx950)
// A kernel whose serial chains all sit in wave 0 of a 4-wave workgroup is bound by ONE SIMD of the CU if the dispatcher
// always starts a workgroup on the same SIMD.  Launches workgroups of 2 / 4 / 8 waves that stay resident a while (so that
// several share a CU, as in the pipeline) and histograms HW_ID.simd_id per wave index.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__global__ void k_probe(unsigned *hist, int nw, int spin) {
  const int wave = threadIdx.x >> 6;
  unsigned hw;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
  const unsigned simd = (hw >> 4) & 3;
  float x = (float)threadIdx.x;
  for (int i = 0; i < spin; i++) x = x * 1.0001f + 0.5f;   // stay resident
  if ((threadIdx.x & 63) == 0) atomicAdd(&hist[wave * 4 + simd], 1u);
  if (x == 12345.678f) hist[0] = 0;
}

int main() {
  unsigned *d;
  hipMalloc(&d, 64 * 4 * sizeof(unsigned));
  for (int nw : {1, 2, 4, 8}) {
    for (int lds : {0, 16384}) {
      hipMemset(d, 0, 64 * 4 * sizeof(unsigned));
      hipLaunchKernelGGL(k_probe, dim3(256 * 20), dim3(64 * nw), lds, 0, d, nw, 20000);
      hipDeviceSynchronize();
      std::vector<unsigned> h(64 * 4);
      hipMemcpy(h.data(), d, h.size() * 4, hipMemcpyDeviceToHost);
      printf("waves/WG %d, dyn LDS %5d:", nw, lds);
      for (int w = 0; w < nw; w++) printf("  w%d[%u %u %u %u]", w, h[w * 4], h[w * 4 + 1], h[w * 4 + 2], h[w * 4 + 3]);
      printf("\n");
    }
  }
  return 0;
}
