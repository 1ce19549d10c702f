// Does the host <-> device copy rate depend on HOW MUCH pinned memory the copies rotate over?  (IOMMU / translation reach)
// hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/microbench/pcie_ws.hip -o tools/microbench/pcie_ws
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1);} } while (0)
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
  const size_t chunk = 16u << 20;
  void* d; CK(hipMalloc(&d, chunk));
  hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  for (int nbuf : {1, 8, 32, 64, 128, 256}) {
    std::vector<void*> bufs(nbuf);
    for (auto& p : bufs) { CK(hipHostMalloc(&p, chunk, hipHostMallocPortable)); memset(p, 1, chunk); }
    for (int dir = 0; dir < 2; ++dir) {
      double best = 1e9;
      for (int rep = 0; rep < 3; ++rep) {
        CK(hipDeviceSynchronize());
        const double a = now();
        for (int i = 0; i < 64; ++i) {
          void* h = bufs[i % nbuf];
          if (dir == 0) CK(hipMemcpyAsync(d, h, chunk, hipMemcpyHostToDevice, s)); else CK(hipMemcpyAsync(h, d, chunk, hipMemcpyDeviceToHost, s));
        }
        CK(hipDeviceSynchronize());
        best = std::min(best, now() - a);
      }
      printf("%4d pinned buffers x 16 MiB (%5zu MiB working set)  %s  %.1f GB/s\n", nbuf, (size_t)nbuf * 16, dir ? "D2H" : "H2D", 64.0 * chunk / best * 1e-9);
    }
    for (auto& p : bufs) CK(hipHostFree(p));
  }
  return 0;
}
