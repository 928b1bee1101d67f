// kernels_cand.hip -- detection order and the octaveMap claim on the device.
//
// Reference: ScaleSpaceDetector visits (octave, level, row, column) in this order per image (pyramid.cpp:438-451, 490-498,
// 564-571) and localizeKeypoint drops a candidate whose CONVERGED pixel of the octave was already claimed by an earlier one
// (octaveMap, :414-418).  The scan + refinement kernels leave the accepted candidates unordered; until round 4 the host
// sorted them (one 64-bit key each) and applied the claim through a hash table, per launch set.  Here:
//   k_cand_keys     key = image | octave | level | row | column of the extremum (unique per candidate); slots past the
//                   device-side count get the largest key, so the sort may run over a host-chosen capacity
//   rocprim::radix_sort_pairs (key, candidate index)
//   k_cand_claim    sorted position p claims (image, octave, converged row, converged column): open addressing on the
//                   64-bit cell key (atomicCAS), the cell keeps the smallest position (atomicMin) = the first visitor
//   k_cand_compact  the candidates whose position owns their cell, in sorted order (one workgroup: a launch set holds
//                   ~10^5 candidates), and their number
// The host then only forms scale = curSigma * powf(2, b2 / numberOfScales) (glibc's powf, which the device cannot
// reproduce bit for bit) and the keypoint records, in one linear pass.
#include <cstring>
#include <rocprim/device/device_radix_sort.hpp>
#include "engine.hpp"

namespace mx {

__global__ void k_cand_keys(const Candidate *cand, const unsigned *counter, unsigned nsort, unsigned long long *keys, unsigned *idx) {
  const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nsort) return;
  const unsigned cnt = min(counter[0], nsort);
  unsigned long long k = ~0ull;
  if (i < cnt) {
    const Candidate q = cand[i];
    k = ((unsigned long long)(unsigned)q.img << 56) | ((unsigned long long)(unsigned)q.octave << 51) | ((unsigned long long)(unsigned)q.level << 48) |
        ((unsigned long long)(unsigned)q.r0 << 24) | (unsigned long long)(unsigned)q.c0;
  }
  keys[i] = k;
  idx[i] = i;
}

__global__ void k_cand_claim(const Candidate *cand, const unsigned *counter, unsigned nsort, const unsigned *idxSorted,
                             unsigned long long *tabKey, unsigned *tabRank, unsigned tabMask, unsigned *slotOf) {
  const unsigned p = blockIdx.x * blockDim.x + threadIdx.x;
  const unsigned cnt = min(counter[0], nsort);
  if (p >= cnt) return;
  const Candidate q = cand[idxSorted[p]];
  const unsigned long long K = (((unsigned long long)(unsigned)q.img << 53) | ((unsigned long long)(unsigned)q.octave << 48) |
                                ((unsigned long long)(unsigned)q.r << 24) | (unsigned long long)(unsigned)q.c) + 1ull;
  unsigned h = (unsigned)((K * 0x9E3779B97F4A7C15ull) >> 20) & tabMask;
  for (;;) {
    const unsigned long long prev = atomicCAS(&tabKey[h], 0ull, K);
    if (prev == 0ull || prev == K) break;
    h = (h + 1) & tabMask;
  }
  atomicMin(&tabRank[h], p);
  slotOf[p] = h;
}

__global__ __launch_bounds__(1024) void k_cand_compact(const Candidate *cand, const unsigned *counter, unsigned nsort,
                                                       const unsigned *idxSorted, const unsigned *tabRank, const unsigned *slotOf,
                                                       Candidate *out, unsigned *outCount) {
  __shared__ unsigned part[1024];
  const unsigned cnt = min(counter[0], nsort);
  const unsigned per = (cnt + 1023) / 1024;
  const unsigned lo = min(cnt, threadIdx.x * per), hi = min(cnt, lo + per);
  unsigned mine = 0;
  for (unsigned p = lo; p < hi; p++) mine += tabRank[slotOf[p]] == p ? 1u : 0u;
  part[threadIdx.x] = mine;
  __syncthreads();
  for (unsigned d = 1; d < 1024; d <<= 1) {     // inclusive scan
    const unsigned v = threadIdx.x >= d ? part[threadIdx.x - d] : 0u;
    __syncthreads();
    part[threadIdx.x] += v;
    __syncthreads();
  }
  unsigned at = part[threadIdx.x] - mine;
  for (unsigned p = lo; p < hi; p++)
    if (tabRank[slotOf[p]] == p) out[at++] = cand[idxSorted[p]];
  if (threadIdx.x == 1023) *outCount = part[1023];
}

size_t cand_sort_temp_bytes(unsigned nsort) {
  size_t bytes = 0;
  rocprim::radix_sort_pairs(nullptr, bytes, (const unsigned long long *)nullptr, (unsigned long long *)nullptr, (const unsigned *)nullptr,
                            (unsigned *)nullptr, nsort, 0, 64, (hipStream_t)0);
  return bytes;
}

// order + claim of the first min(*counter, nsort) candidates of `cand`; survivors in order at `out`, their number at *outCount
int launch_cand_order(hipStream_t s, const Candidate *cand, const unsigned *counter, unsigned nsort, unsigned long long *keys,
                      unsigned long long *keys2, unsigned *idx, unsigned *idx2, void *temp, size_t tempBytes, unsigned long long *tabKey,
                      unsigned *tabRank, unsigned tabSize, unsigned *slotOf, Candidate *out, unsigned *outCount) {
  if (!nsort) return 0;
  hipLaunchKernelGGL(k_cand_keys, dim3((nsort + 255) / 256), dim3(256), 0, s, cand, counter, nsort, keys, idx);
  if (rocprim::radix_sort_pairs(temp, tempBytes, (const unsigned long long *)keys, keys2, (const unsigned *)idx, idx2, nsort, 0, 64, s) != hipSuccess)
    return -1;
  if (hipMemsetAsync(tabKey, 0, (size_t)tabSize * 8, s) != hipSuccess || hipMemsetAsync(tabRank, 0xff, (size_t)tabSize * 4, s) != hipSuccess) return -1;
  hipLaunchKernelGGL(k_cand_claim, dim3((nsort + 255) / 256), dim3(256), 0, s, cand, counter, nsort, (const unsigned *)idx2, tabKey, tabRank, tabSize - 1, slotOf);
  hipLaunchKernelGGL(k_cand_compact, dim3(1), dim3(1024), 0, s, cand, counter, nsort, (const unsigned *)idx2, (const unsigned *)tabRank,
                     (const unsigned *)slotOf, out, outCount);
  return 0;
}

}  // namespace mx

// The device-side order is an opt-in path (MODSX_DEVICE_ORDER=1) that measured 9 % slower than the host's order, and its rocPRIM
// radix / merge sort instantiations are 2.4 MB of code objects: it is built as a library of its own (libmodsx_cand.so, next to
// libmodsx.so) that engine.hip loads on first use, so the product library does not carry it.
extern "C" __attribute__((visibility("default"))) size_t modsx_cand_sort_temp_bytes(unsigned nsort) { return mx::cand_sort_temp_bytes(nsort); }
extern "C" __attribute__((visibility("default"))) int modsx_cand_order(hipStream_t s, const mx::Candidate *cand, const unsigned *counter, unsigned nsort,
                                                                        unsigned long long *keys, unsigned long long *keys2, unsigned *idx,
                                                                        unsigned *idx2, void *temp, size_t tempBytes, unsigned long long *tabKey,
                                                                        unsigned *tabRank, unsigned tabSize, unsigned *slotOf, mx::Candidate *out,
                                                                        unsigned *outCount) {
  return mx::launch_cand_order(s, cand, counter, nsort, keys, keys2, idx, idx2, temp, tempBytes, tabKey, tabRank, tabSize, slotOf, out, outCount);
}
