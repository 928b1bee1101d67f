// Measurement for VERDICT r5 item 5 (MSER on the device: "replace the estimate by a measurement"): the component tree of
// mods_amd/csrc/mser.cpp's Forest::run -- pixels in (grey level, raster) order, four neighbour look-ups in parent[], find with path
// halving, union under the "largest survives" rule, area / perimeter counters per root, promotion at min_size -- compiled as a device
// function, ONE LANE PER TREE, every tree with its own parent[] / node[] arrays in HBM (tree-major, as a (view, polarity) task owns
// them on the host) and its own image (the input shifted cyclically by 131 t pixels, so that the trees of a wavefront do not walk in
// lock-step through the same addresses).  The bin sort is done on the host (it is data parallel and not what is being measured).
// Left out: the per-level histograms of promoted regions and close_region (~15 % of a tree on the host) -- the figure is an UPPER
// bound of what a faithful device tree would deliver.
//   usage: mser_tree <image.u8> <rows> <cols> <trees> [trees ...]      -> trees/s per count
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>
#include <chrono>
struct Node { int area, perim, region; };
__device__ __forceinline__ int findp(int *parent, int p) {
  while (parent[p] != p) { const int g = parent[parent[p]]; parent[p] = g; p = g; }
  return p;
}
__global__ __launch_bounds__(64) void k_trees(const int *order, const uint8_t *grey, int *parentAll, Node *nodeAll, int rows, int cols,
                                              int ntrees, int promoteAt, unsigned long long *sink) {
  const int t = blockIdx.x * 64 + threadIdx.x;
  if (t >= ntrees) return;
  const int stride = cols + 2;
  const size_t npx = (size_t)(rows + 2) * stride, n = (size_t)rows * cols;
  int *parent = parentAll + (size_t)t * npx;
  Node *node = nodeAll + (size_t)t * npx;
  const int *ord = order + (size_t)t * n;
  unsigned long long acc = 0;
  for (size_t k = 0; k < n; k++) {
    const int ofs = ord[k];
    const int nb[4] = {ofs - stride, ofs - 1, ofs + 1, ofs + stride};
    int roots[4], nroots = 0, touching = 0;
#pragma unroll
    for (int q = 0; q < 4; q++) {
      const int pq = parent[nb[q]];
      if (pq < 0) continue;
      touching++;
      if (nroots && (pq == roots[nroots - 1] || parent[pq] == roots[nroots - 1]) && parent[roots[nroots - 1]] == roots[nroots - 1]) continue;
      const int r = findp(parent, nb[q]);
      bool dup = false;
      for (int z = 0; z < nroots; z++) dup = dup || roots[z] == r;
      if (!dup) roots[nroots++] = r;
    }
    if (nroots == 0) { parent[ofs] = ofs; node[ofs] = Node{1, 4, -1}; continue; }
    int keep = roots[0];
    if (nroots > 1) {
      int best = 0;
      for (int z = 0; z < nroots; z++) if (node[roots[z]].region >= 0 && node[roots[z]].area > best) { best = node[roots[z]].area; keep = roots[z]; }
      for (int z = 0; z < nroots; z++) {
        const int r = roots[z];
        if (r == keep) continue;
        parent[r] = keep;
        node[keep].area += node[r].area; node[keep].perim += node[r].perim;
      }
    }
    parent[ofs] = keep;
    Node R = node[keep];
    R.area++; R.perim += 4 - 2 * touching;
    if (R.region < 0 && R.area >= promoteAt) { R.region = 1; acc++; }
    node[keep] = R;
  }
  sink[t] = acc + (unsigned long long)node[findp(parent, stride + 1)].area;
}
int main(int argc, char **argv) {
  if (argc < 5) { printf("usage: mser_tree <image.u8> <rows> <cols> <trees> [...]\n"); return 1; }
  const int rows = atoi(argv[2]), cols = atoi(argv[3]), stride = cols + 2;
  const size_t n = (size_t)rows * cols, npx = (size_t)(rows + 2) * stride;
  std::vector<uint8_t> img(n);
  FILE *f = fopen(argv[1], "rb");
  if (!f || fread(img.data(), 1, n, f) != n) { printf("cannot read %s\n", argv[1]); return 1; }
  fclose(f);
  for (int ai = 4; ai < argc; ai++) {
    const int T = atoi(argv[ai]);
    std::vector<int> order((size_t)T * n);
    std::vector<uint8_t> g(n);
    double hostTree = 0;
    for (int t = 0; t < T; t++) {
      const size_t sh = ((size_t)131 * t * 7919) % n;
      for (size_t i = 0; i < n; i++) g[i] = img[(i + sh) % n];
      int start[257] = {0};
      for (size_t i = 0; i < n; i++) start[g[i] + 1]++;
      for (int l = 0; l < 256; l++) start[l + 1] += start[l];
      int *o = &order[(size_t)t * n];
      for (int r = 0; r < rows; r++) for (int c = 0; c < cols; c++) o[start[g[(size_t)r * cols + c]]++] = (r + 1) * stride + c + 1;
      if (t == 0) {   // the same loop on one host core, for the ratio
        std::vector<int> parent(npx, -1); std::vector<Node> node(npx);
        const auto h0 = std::chrono::steady_clock::now();
        for (size_t k = 0; k < n; k++) {
          const int ofs = o[k]; const int nb[4] = {ofs - stride, ofs - 1, ofs + 1, ofs + stride};
          int roots[4], nroots = 0, touching = 0;
          for (int q = 0; q < 4; q++) { const int pq = parent[nb[q]]; if (pq < 0) continue; touching++;
            int p = nb[q]; while (parent[p] != p) { const int gg = parent[parent[p]]; parent[p] = gg; p = gg; }
            bool dup = false; for (int z = 0; z < nroots; z++) dup = dup || roots[z] == p; if (!dup) roots[nroots++] = p; }
          if (!nroots) { parent[ofs] = ofs; node[ofs] = Node{1, 4, -1}; continue; }
          int keep = roots[0];
          for (int z = 1; z < nroots; z++) { parent[roots[z]] = keep; node[keep].area += node[roots[z]].area; node[keep].perim += node[roots[z]].perim; }
          parent[ofs] = keep; node[keep].area++; node[keep].perim += 4 - 2 * touching;
        }
        hostTree = std::chrono::duration<double>(std::chrono::steady_clock::now() - h0).count();
      }
    }
    int *dOrder, *dParent; Node *dNode; uint8_t *dGrey; unsigned long long *dSink;
    if (hipMalloc(&dOrder, order.size() * 4) != hipSuccess || hipMalloc(&dParent, (size_t)T * npx * 4) != hipSuccess ||
        hipMalloc(&dNode, (size_t)T * npx * sizeof(Node)) != hipSuccess || hipMalloc(&dGrey, n) != hipSuccess || hipMalloc(&dSink, (size_t)T * 8) != hipSuccess) {
      printf("trees %d: out of device memory\n", T); return 1;
    }
    (void)hipMemcpy(dOrder, order.data(), order.size() * 4, hipMemcpyHostToDevice);
    (void)hipMemset(dParent, 0xff, (size_t)T * npx * 4);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k_trees, dim3((T + 63) / 64), dim3(64), 0, 0, dOrder, dGrey, dParent, dNode, rows, cols, T, 30, dSink);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> sink(T);
    (void)hipMemcpy(sink.data(), dSink, (size_t)T * 8, hipMemcpyDeviceToHost);
    printf("trees %5d (%dx%d, one lane each, %d wavefronts): %.1f ms -> %.1f trees/s = %.2f pairs/s at 54 trees per pair (cviu MSER steps); "
           "one host core: %.1f ms per tree; check %llu\n", T, cols, rows, (T + 63) / 64, ms, T / (ms * 1e-3), T / (ms * 1e-3) / 54.0,
           hostTree * 1e3, sink[0]);
    (void)hipFree(dOrder); (void)hipFree(dParent); (void)hipFree(dNode); (void)hipFree(dGrey); (void)hipFree(dSink);
  }
  return 0;
}
