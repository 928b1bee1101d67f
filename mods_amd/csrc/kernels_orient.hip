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
  // mask 0 to ORI_NV entries; table entry lane + 64 q = list element PER_L lane + q, see the histogram below) instead of the whole
  // 44 x 41 grid: 21 passes instead of 29 of the kernel's most expensive
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
  // The histogram: bin b is the f32 running sum of its votes in raster order (the reference's loop, synth-detection.cpp:771-787).
  // Until round 5 lane b added its votes one instruction per vote with EXEC opened for that lane alone -- 1245 single-lane adds, the
  // largest stage of the kernel.  Now the votes are first SORTED BY BIN, stably, and lane b then adds only its own: the list is
  // laid out so that lane l holds the CONTIGUOUS segment [PER_L l, PER_L (l + 1)) of the raster-ordered voting list in its
  // registers (the host permutes the pixel table: entry lane + 64 q = list element PER_L lane + q), so a counting sort needs no
  // lane exchange: (1) every lane counts its votes per bin in its own column of a [bin][lane] table of 16-bit counters (and keeps a
  // vote's rank among the lane's earlier votes of that bin), (2) lane b turns row b into exclusive prefix sums -- (bin, lane) order
  // IS raster order within a bin -- and the row totals into the bins' bases, (3) a vote's slot is base[bin] + row prefix + rank,
  // (4) the weights go to their slots, (5) lane b adds its slots in order.  ~550 vector instructions instead of ~2000, the same sums.
  // The counters live in the patch buffer (dead: the gradients are in registers), the sorted weights replace them.
  constexpr int NB = 37;                              // bins 0..36 (36: ori == pi, a slot of its own that nothing reads, as in the reference)
  constexpr int NROW = NB + 1;                        // + a row for the pixels that do not vote (no branch per pixel)
  constexpr int DUMP = ORI_NV;                        // their weights go to slots DUMP + lane, which nothing reads
  static_assert(NROW * 64 * 2 <= PS * PSP * 4 && (ORI_NV + 64) * 4 <= PS * PSP * 4, "counters and sorted weights fit in the patch buffer");
  unsigned short *const cnt = reinterpret_cast<unsigned short *>(bufX);
  __shared__ int sbase[64];
  {
    float4 *const z = reinterpret_cast<float4 *>(bufX);
    for (int i = lane; i < (NROW * 64 * 2 + 15) / 16; i += 64) z[i] = make_float4(0.f, 0.f, 0.f, 0.f);
  }
  __syncthreads();
  int slot[PER_L], crow[PER_L];                       // rank among the lane's earlier votes of the bin, later the vote's slot; counter index
#pragma unroll
  for (int q = 0; q < PER_L; q++) {
    const int bq = breg[q] == 255 ? NB : (int)breg[q];
    crow[q] = bq * 64 + lane;
    const unsigned short c = cnt[crow[q]];            // (LDS operations of a wavefront execute in order: a later vote of the same bin sees this increment)
    slot[q] = c;
    cnt[crow[q]] = (unsigned short)(c + 1);
  }
  __syncthreads();
  int nmine = 0;                                      // lane b < NB: votes of bin b
  if (lane < NB) {
    uint4 *const row = reinterpret_cast<uint4 *>(bufX) + lane * 8;   // 64 counters = 8 x 16 bytes
    unsigned run = 0;
    auto pre = [&run](unsigned w) { const unsigned lo = w & 0xffffu, hi = w >> 16; const unsigned o = run | ((run + lo) << 16); run += lo + hi; return o; };
#pragma unroll
    for (int i = 0; i < 8; i++) {
      uint4 w = row[i];
      w.x = pre(w.x); w.y = pre(w.y); w.z = pre(w.z); w.w = pre(w.w);
      row[i] = w;
    }
    nmine = (int)run;
  }
  int base;                                           // slot of bin b's first vote: exclusive prefix of the totals over the lanes (0 beyond NB)
  {
    int incl = nmine;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const int v = __shfl_up(incl, d); if (lane >= d) incl += v; }
    base = incl - nmine;
    sbase[lane] = base;
  }
  __syncthreads();
#pragma unroll
  for (int q = 0; q < PER_L; q++) {
    const int bq = crow[q] >> 6;
    const int sl = slot[q] + (int)cnt[crow[q]] + sbase[bq];
    slot[q] = bq == NB ? DUMP + lane : sl;
  }
  __syncthreads();                                    // every slot is known: the sorted weights may take the counters' place
#pragma unroll
  for (int q = 0; q < PER_L; q++) bufX[slot[q]] = wreg[q];
  __syncthreads();
  {
    int nmax = nmine;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) nmax = max(nmax, __shfl_xor(nmax, off));
    nmax = __builtin_amdgcn_readfirstlane(nmax);
    float h = 0.f;
    // eight slots per step, read together (a lane past its own votes reads its neighbours' slots and drops them: the address stays
    // inside the buffer), added in order
    for (int k0 = 0; k0 < nmax; k0 += 8) {
      const float *m8 = bufX + min(base + k0, ORI_NV + 56);
      const int rem = nmine - k0;
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; u++) v[u] = m8[u];
      asm volatile("" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]));   // (the loads stay together)
#pragma unroll
      for (int u = 0; u < 8; u++)
        if (rem > u) h += v[u];
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
