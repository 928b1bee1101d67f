// kernels_orient.hip -- dominant orientation of affine regions, one wavefront per region.
//
// Reference: DetectOrientation + EstimateDominantAnglesFunctor, synth-detection.cpp:746-919
//   interpolate (41x41 patch, A * curr_sc)                 detectors/helpers.cpp:551-626
//   computeGradientMagnitudeAndOrientation + atan2LUTff     detectors/helpers.cpp:840-863, 160-207
//   36-bin histogram of mag * circular Gauss mask, 6x circular smoothing, parabolic peaks,
//   first maxAngles peaks in bin order.
// Lane j walks row j of the patch (the f32 sample coordinates are running sums along a row) and takes its taps;
// histogram bin b is accumulated by lane b scanning the pixels in raster order, which keeps
// the reference's f32 summation order per bin.
#include "engine.hpp"

namespace mx {

constexpr int PS = 41, PSP = 44;   // PSP: padded row stride so rows start 16-byte aligned in LDS
constexpr int ORI_B = 21;          // taps of a patch row in flight per lane

// LDS_TABLE: the launch is too small to fill the device (a lone pair of few views): every wavefront then runs alone on its SIMD,
// its 20 table look-ups per lane are trips to L2 one after the other, and a copy of the table in LDS is the faster choice
// (0.34 against 0.52 ms for the orientation stage of a 1-view pair).  Large launches read the table where it lies: 2.1 KB of LDS
// less per region is three more regions per CU.
template <bool LDS_TABLE>
__global__ __launch_bounds__(64) void k_orientation(const OriJob *jobs, float *out, int n, const ImgRef *imgs,
                                                    const unsigned short *maskIdx, const float *maskW,
                                                    const unsigned char *binTab, int doHalf, double th, int maxAngles) {
  const int k = xcd_chunk(blockIdx.x, n);   // regions are listed image by image: one part of the views per XCD's L2
  if (k >= n) return;
  const int lane = threadIdx.x;
  // 7.4 KB of LDS per region (21 regions resident per CU; it was 11.3 KB / 14 with a table copy and a separate bin array, and
  // the kernel is bound by the latency of its dependent phases, not by issue): the patch, later overwritten by the voting list
  __shared__ __attribute__((aligned(16))) float bufX[PS * PSP];
  // the bins of the compacted voting list live behind its weights in the patch's own buffer (the patch is dead by then: the
  // gradients are staged in registers): at most ORI_NV + 4 weights, then the bytes -- 7.4 KB of LDS per region instead of 9.2
  static_assert((ORI_NV + 4) * 4 + ORI_NV + 4 <= PS * PSP * 4 && (ORI_NV + 4) % 4 == 0, "the voting list fits in the patch buffer");
  unsigned char *const sbin = reinterpret_cast<unsigned char *>(bufX + ORI_NV + 4);
  // the 2 KB bin table is read where it lies (every wavefront of the launch hits the same 33 cache lines; a copy per
  // workgroup cost 2.1 of the 11.3 KB of LDS that bound the residency: 14 -> 17 regions per CU) unless the launch is small
  __shared__ __attribute__((aligned(16))) unsigned char sbtLds[LDS_TABLE ? ATAN_CASES : 16];
  const unsigned char *const sbt = LDS_TABLE ? sbtLds : binTab;
  __shared__ float hist[40];
  static_assert(ATAN_CASES == 33 * 64, "one 4-byte word per lane and step");
  if (LDS_TABLE) {   // the bin table: independent loads per lane, issued together
    unsigned t[9];
#pragma unroll
    for (int u = 0; u < 9; u++) t[u] = lane + 64 * u < ATAN_CASES / 4 ? reinterpret_cast<const unsigned *>(binTab)[lane + 64 * u] : 0u;
#pragma unroll
    for (int u = 0; u < 9; u++)
      if (lane + 64 * u < ATAN_CASES / 4) reinterpret_cast<unsigned *>(sbtLds)[lane + 64 * u] = t[u];
  }
  const OriJob jb = jobs[k];
  const ImgRef im = imgs[jb.img];
  const int half = PS >> 1;
  const bool touch = check_borders(im.cols, im.rows, jb.x, jb.y, jb.a11, jb.a12, jb.a21, jb.a22, PS, PS);
  // lane j owns row j of the patch: it runs the f32 running sums of interpolate() (helpers.cpp:563-585) along the row
  // and takes the taps as it goes, four columns in flight
  {
    // row starts: lane j takes j steps of the running sum (a12 / a22 are uniform: a lane that has its row start sits out
    // the remaining steps)
    float rx = jb.x - (float)half * jb.a12, ry = jb.y - (float)half * jb.a22;
#pragma unroll
    for (int j = 1; j < PS; j++)
      if (lane >= j) { rx += jb.a12; ry += jb.a22; }
    if (lane < PS) {
      float WX = rx - (float)half * jb.a11;
      float WY = ry - (float)half * jb.a21;
      float *row = bufX + lane * PSP;
      // ORI_B taps of the row in flight at a time (the coordinates are cheap dependent adds, the taps are not dependent)
      for (int i0 = 0; i0 < PS; i0 += ORI_B) {
        float t[ORI_B];
        if (!touch) {
#pragma unroll
          for (int u = 0; u < ORI_B; u++) {
            t[u] = i0 + u < PS ? bilinear_tap(as_global(im.d), im.rows, im.cols, WX, WY, false) : 0.f;
            WX += jb.a11; WY += jb.a21;
          }
        } else {
#pragma unroll
          for (int u = 0; u < ORI_B; u++) {
            t[u] = i0 + u < PS ? bilinear_tap(as_global(im.d), im.rows, im.cols, WX, WY, true) : 0.f;
            WX += jb.a11; WY += jb.a21;
          }
        }
#pragma unroll
        for (int u = 0; u < ORI_B; u++)
          if (i0 + u < PS) row[i0 + u] = t[u];
      }
    }
  }
  __syncthreads();
  const float PIf = float(M_PI);
  // gradients -> (bin, weight) per pixel, staged in registers so that the weights can take the patch's place.  Only the
  // pixels under the circular mask can vote (mask > 0: a disc of radius 20, 1245 of the 1681 pixels, all of them interior),
  // so the wave walks the host-built raster-ordered list of those pixels (padded LDS index + mask value, padded with
  // mask 0 to ORI_NV entries) instead of the whole 44 x 41 grid: 20 passes instead of 29 of the kernel's most expensive
  // per-pixel code (IEEE sqrt, two IEEE divisions, the f64 look-up).
  constexpr int PER_L = ORI_NV / 64;
  float wreg[PER_L];
  unsigned char breg[PER_L];
#pragma unroll
  for (int q = 0; q < PER_L; q++) {
    const int e = lane + 64 * q;
    const int p = maskIdx[e];
    const float m = maskW[e];
    const float xg = bufX[p + 1] - bufX[p - 1];
    const float yg = bufX[p + PSP] - bufX[p - PSP];
    const float mag = sqrtf(xg * xg + yg * yg);
    // bin = (int)(36 * (ori / pi + 1) / 2) of ori = atan2LUTff(yg, xg): the angle takes one of 8 x 256 + 1 values, so the
    // bin comes from a table built with that expression (engine.hip: upload_tables) -- no f64 look-up, no division by pi
    int code, idx;
    const bool special = atan2lut_case(yg, xg, code, idx);
    unsigned char bin = 255;
    float w = 0.f;
    if (m > 0 && mag > 1.0f) {
      bin = sbt[special ? 2048 : code * 256 + idx];
      w = mag * m;
    }
    wreg[q] = w; breg[q] = bin;
  }
  __syncthreads();
  // Only pixels with a gradient above 1 vote, so the (bin, weight) pairs are first compacted IN RASTER ORDER: list entry
  // e = lane + 64 q, so chunk q precedes chunk q + 1 and within a chunk lanes are ordered -- a ballot prefix keeps the order.
  int nv = 0;   // wave-uniform
#pragma unroll
  for (int q = 0; q < PER_L; q++) {
    const bool valid = breg[q] != 255;
    const unsigned long long m = __ballot(valid);
    const int pos = nv + __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
    if (valid) { bufX[pos] = wreg[q]; sbin[pos] = breg[q]; }
    nv += __popcll(m);
  }
  if (lane < 4) { bufX[nv + lane] = 0.f; sbin[nv + lane] = 255; }   // pad the last group of four
  __syncthreads();
  // lane b owns histogram bin b and adds its pixels in raster order (f32 running sum of the reference).  This loop is the
  // kernel's largest stage and is bound by VALU issue, so a vote is ONE vector instruction: the four bins of a group are
  // wave-uniform (every lane reads the same LDS word), so they go to a scalar register once per group and the scalar unit
  // -- which issues beside the vector unit -- opens EXEC for exactly lane `bin` (s_bfe + s_lshl of 1) before each add.
  // The 255 of padding selects lane 63, whose sum is never read.  All 64 lanes of the single wavefront are active here.
  {
    float h = 0.f;
    const unsigned *b4 = reinterpret_cast<const unsigned *>(sbin);
    const float4 *w4 = reinterpret_cast<const float4 *>(bufX);
    const int ng = (nv + 3) >> 2;
#pragma unroll 4
    for (int g = 0; g < ng; g++) {
      const unsigned b = __builtin_amdgcn_readfirstlane(b4[g]);
      const float4 w = w4[g];
      unsigned t;
      unsigned long long ex;   // the incoming EXEC is saved and restored (the block is entered with all 64 lanes of the
                               // launch's single wavefront active, but nothing here depends on that)
      asm volatile(
          "s_mov_b64 %2, exec\n\t"
          "s_bfe_u32 %1, %3, 0x60000\n\t"
          "s_lshl_b64 exec, 1, %1\n\t"
          "v_add_f32_e32 %0, %0, %4\n\t"
          "s_bfe_u32 %1, %3, 0x60008\n\t"
          "s_lshl_b64 exec, 1, %1\n\t"
          "v_add_f32_e32 %0, %0, %5\n\t"
          "s_bfe_u32 %1, %3, 0x60010\n\t"
          "s_lshl_b64 exec, 1, %1\n\t"
          "v_add_f32_e32 %0, %0, %6\n\t"
          "s_bfe_u32 %1, %3, 0x60018\n\t"
          "s_lshl_b64 exec, 1, %1\n\t"
          "v_add_f32_e32 %0, %0, %7\n\t"
          "s_mov_b64 exec, %2"
          : "+v"(h), "=&s"(t), "=&s"(ex)
          : "s"(b), "v"(w.x), "v"(w.y), "v"(w.z), "v"(w.w)
          : "scc", "memory");
    }
    if (lane < 36) hist[lane] = h;
  }
  __syncthreads();
  // 6 passes of the circular [1 1 1] smoothing: the in-place loop of the reference (synth-detection.cpp:795-809) only
  // ever reads not-yet-updated neighbours, i.e. new[i] = (old[i-1] + old[i]) + old[i+1]; one lane per bin.
  const int nb = 36;
  float hv = lane < nb ? hist[lane] : 0.f;
  for (int it = 0; it < 6; it++) {
    const float left = __shfl(hv, lane == 0 ? nb - 1 : lane - 1);
    const float right = __shfl(hv, lane == nb - 1 ? 0 : lane + 1);
    hv = left + hv + right;
  }
  float mx = lane < nb ? hv : 0.f;   // thresh starts at 0 and takes every larger bin: a maximum, order free
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off));
  const float thresh = (float)((double)(mx > 0.f ? mx : 0.f) * th);
  if (doHalf) {
    const float other = __shfl(hv, lane < nb / 2 ? lane + nb / 2 : lane);
    if (lane < nb / 2) hv += other;
    else if (lane < nb) hv = 0.f;
  }
  const float ha = __shfl(hv, lane == 0 ? nb - 1 : lane - 1);
  const float hc = __shfl(hv, lane == nb - 1 ? 0 : lane + 1);
  const bool peak = lane < nb && hv >= thresh && hv > ha && hv > hc;
  const unsigned long long peaks = __ballot(peak);
  // the reference keeps the first min(maxAngles, #peaks) peaks in bin order (35,0,1), (i-1,i,i+1), (34,35,0)
  const int rank = __popcll(peaks & ((1ull << lane) - 1ull));
  float *const o = out + (size_t)k * (1 + maxAngles);   // maxAngles <= ORI_MAX_PEAKS: the stride of the result records
  if (lane == 0) o[0] = __int_as_float(min(__popcll(peaks), maxAngles));
  if (peak && rank < maxAngles) {
    const float pp = (ha - hc) / (ha - 2.0f * hv + hc) / 2.0f;
    o[1 + rank] = 2.0f * PIf * ((float)lane + 0.5f + pp) / (float)nb - PIf;
  }
}

void launch_orientation(hipStream_t s, const OriJob *jobs, float *out, int n, const ImgRef *imgs,
                          const unsigned short *maskIdx, const float *maskW, const unsigned char *binTab, int doHalf, double th,
                          int maxAngles) {
  if (n <= 0) return;
  // fewer regions than two rounds of wavefronts over the chip: the launch is latency, not residency
  if (n < 2 * 256 * 16)
    MX_DUP(K_ORIENT) hipLaunchKernelGGL(k_orientation<true>, dim3(8 * ((n + 7) / 8)), dim3(64), 0, s, jobs, out, n, imgs, maskIdx, maskW, binTab, doHalf, th,
                       maxAngles);
  else
    MX_DUP(K_ORIENT) hipLaunchKernelGGL(k_orientation<false>, dim3(8 * ((n + 7) / 8)), dim3(64), 0, s, jobs, out, n, imgs, maskIdx, maskW, binTab, doHalf, th,
                       maxAngles);
}

}  // namespace mx
