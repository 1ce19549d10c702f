// Latency of dependent point operations for a lone wave (the regime of the Pippenger tail kernels) and for
// 2 / 4 waves per SIMD: ns per ge_double / ge_madd / ge_add_p3 in a dependent chain.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include "../../zkp_amd/csrc/dev_layout.h"
using namespace zkp;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)

template <int V>
__global__ void __launch_bounds__(64, 2) k_chain(const uint8_t* enc, dev_ext* out, int iters, int active_lanes) {
  if ((int)threadIdx.x >= active_lanes) return;
  uint32_t w[8];
  load_vec<2>(w, enc);
  asm volatile("" : "+v"(w[0]), "+v"(w[1]), "+v"(w[2]), "+v"(w[3]), "+v"(w[4]), "+v"(w[5]), "+v"(w[6]), "+v"(w[7]));   // keep the arithmetic on the VALU (uniform data would be scalarised)
  ge_p3 p, acc;
  ristretto_decode(p, w);
  ge_niels n; ge_affine_to_niels(n, p);
  acc = p;
#pragma unroll 1
  for (int it = 0; it < iters; ++it) {
    if (V == 0) ge_double<true>(acc, acc);
    else if (V == 1) ge_double<false>(acc, acc);
    else if (V == 2) ge_madd(acc, acc, n);
    else if (V == 3) ge_add_p3(acc, acc, p);
  }
  store_ext(out + blockIdx.x * 64 + threadIdx.x, acc);
}
template <int V>
void run(const char* name, const uint8_t* enc, dev_ext* out, int blocks, int lanes) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int iters = 2000;
  for (int rep = 0; rep < 2; ++rep) {
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(k_chain<V>, dim3(blocks), dim3(64), 0, 0, enc, out, iters, lanes);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    if (rep) printf("%-14s blocks %5d lanes %2d : %8.3f ms -> %8.1f ns per op\n", name, blocks, lanes, ms, ms * 1e6 / iters);
  }
}
int main() {
  uint8_t h[32] = {0xe2,0xf2,0xae,0x0a,0x6a,0xbc,0x4e,0x71,0xa8,0x84,0xa9,0x61,0xc5,0x00,0x51,0x5f,0x58,0xe3,0x0b,0x6a,0xa5,0x82,0xdd,0x8d,0xb6,0xa6,0x59,0x45,0xe0,0x8d,0x2d,0x76};
  uint8_t* enc; dev_ext* out; CK(hipMalloc(&enc, 32)); CK(hipMemcpy(enc, h, 32, hipMemcpyHostToDevice)); CK(hipMalloc(&out, sizeof(dev_ext) * 64 * 8192));
  for (int blocks : {1, 1024, 2048, 4096}) for (int lanes : {1, 64}) {
    if (blocks > 1 && lanes == 1) continue;
    run<0>("double<T>", enc, out, blocks, lanes); run<1>("double<noT>", enc, out, blocks, lanes);
    run<2>("madd", enc, out, blocks, lanes); run<3>("add_p3", enc, out, blocks, lanes);
  }
  return 0;
}
