// Host <-> MI355X copy rates that size the asynchronous host-buffer path (round 4):
// pinned vs pageable hipMemcpyAsync, both directions at once on two streams, the cost of hipHostRegister / hipHostMalloc,
// and what a host memcpy into a pinned staging ring delivers with 1 .. 16 threads.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -pthread tools/microbench/pcie_bw.hip -o tools/microbench/pcie_bw
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { \
  fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1);} } while (0)

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main() {
  const size_t sizes[] = {1u << 20, 8u << 20, 32u << 20, 128u << 20};
  void *d0, *d1;
  CK(hipMalloc(&d0, 128u << 20));
  CK(hipMalloc(&d1, 128u << 20));
  hipStream_t s0, s1;
  CK(hipStreamCreateWithFlags(&s0, hipStreamNonBlocking));
  CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
  double t0 = now();
  void *p0, *p1;
  CK(hipHostMalloc(&p0, 128u << 20, hipHostMallocDefault));
  CK(hipHostMalloc(&p1, 128u << 20, hipHostMallocDefault));
  printf("hipHostMalloc 2 x 128 MiB: %.2f ms\n", (now() - t0) * 1e3);
  memset(p0, 1, 128u << 20);
  memset(p1, 2, 128u << 20);
  char* pg = (char*)malloc(128u << 20);
  memset(pg, 3, 128u << 20);
  printf("%10s %12s %12s %12s %12s %12s\n", "bytes", "H2D pin GB/s", "D2H pin GB/s", "bidir GB/s", "H2D page", "D2H page");
  for (size_t n : sizes) {
    double r[5];
    for (int mode = 0; mode < 5; ++mode) {
      double best = 1e9;
      for (int rep = 0; rep < 6; ++rep) {
        CK(hipDeviceSynchronize());
        const double a = now();
        switch (mode) {
          case 0: CK(hipMemcpyAsync(d0, p0, n, hipMemcpyHostToDevice, s0)); break;
          case 1: CK(hipMemcpyAsync(p1, d1, n, hipMemcpyDeviceToHost, s1)); break;
          case 2: CK(hipMemcpyAsync(d0, p0, n, hipMemcpyHostToDevice, s0)); CK(hipMemcpyAsync(p1, d1, n, hipMemcpyDeviceToHost, s1)); break;
          case 3: CK(hipMemcpyAsync(d0, pg, n, hipMemcpyHostToDevice, s0)); break;
          case 4: CK(hipMemcpyAsync(pg, d1, n, hipMemcpyDeviceToHost, s1)); break;
        }
        CK(hipDeviceSynchronize());
        best = std::min(best, now() - a);
      }
      r[mode] = (mode == 2 ? 2.0 : 1.0) * n / best * 1e-9;
    }
    printf("%10zu %12.1f %12.1f %12.1f %12.1f %12.1f\n", n, r[0], r[1], r[2], r[3], r[4]);
  }
  // many small copies (one per array of a job) on one stream: per-copy overhead
  {
    const int k = 64;
    const size_t n = 256u << 10;
    double best = 1e9;
    for (int rep = 0; rep < 5; ++rep) {
      CK(hipDeviceSynchronize());
      const double a = now();
      for (int i = 0; i < k; ++i) CK(hipMemcpyAsync((char*)d0 + i * n, (char*)p0 + i * n, n, hipMemcpyHostToDevice, s0));
      const double enq = now() - a;
      CK(hipDeviceSynchronize());
      best = std::min(best, now() - a);
      if (rep == 4) printf("64 x 256 KiB pinned H2D on one stream: %.3f ms total (%.1f GB/s), enqueue %.3f ms\n", best * 1e3, k * n / best * 1e-9, enq * 1e3);
    }
  }
  // hipHostRegister of pageable memory
  for (size_t n : {size_t(8) << 20, size_t(128) << 20}) {
    double a = now();
    CK(hipHostRegister(pg, n, hipHostRegisterDefault));
    const double reg = now() - a;
    a = now();
    CK(hipMemcpyAsync(d0, pg, n, hipMemcpyHostToDevice, s0));
    CK(hipDeviceSynchronize());
    const double cp = now() - a;
    a = now();
    CK(hipHostUnregister(pg));
    printf("hipHostRegister %zu MiB: %.2f ms, copy %.1f GB/s, unregister %.2f ms\n", n >> 20, reg * 1e3, n / cp * 1e-9, (now() - a) * 1e3);
  }
  // host memcpy pageable -> pinned with T threads (the staging step for callers with ordinary buffers)
  for (int T : {1, 2, 4, 8, 16}) {
    const size_t n = 64u << 20;
    double best = 1e9;
    for (int rep = 0; rep < 4; ++rep) {
      const double a = now();
      std::vector<std::thread> th;
      for (int t = 0; t < T; ++t) th.emplace_back([&, t] { memcpy((char*)p0 + n / T * t, pg + n / T * t, n / T); });
      for (auto& x : th) x.join();
      best = std::min(best, now() - a);
    }
    printf("host memcpy pageable -> pinned, 64 MiB, %2d threads: %.1f GB/s\n", T, n / best * 1e-9);
  }
  printf("hardware threads: %u\n", std::thread::hardware_concurrency());
  return 0;
}
