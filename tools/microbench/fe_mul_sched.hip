// How fast does one wave get through dependent field multiplications, as a function of the order in
// which the 81 products are issued (column-major = each 64-bit accumulator is a dependent mad chain;
// row-major = consecutive mads hit 9 different accumulators) and of the number of waves per SIMD?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include "../../zkp_amd/csrc/fe25519.h"
using namespace zkp;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)

// row-major (operand scanning) variant of fe_mul
__device__ __forceinline__ void fe_mul_rows(fe& r, const fe& a, const fe& b) {
  uint64_t c[17];
#pragma unroll
  for (int k = 0; k < 17; ++k) c[k] = 0;
#pragma unroll
  for (int i = 0; i < 9; ++i) {
#pragma unroll
    for (int j = 0; j < 9; ++j) c[i + j] += (uint64_t)a.v[i] * b.v[j];
  }
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    c[k] += 1216ull * (uint32_t)c[k + 9];
    c[k + 1] += 9728ull * (uint32_t)(c[k + 9] >> 32);
  }
  fe_reduce_columns(r, c);
}

template <int VARIANT>
__global__ void __launch_bounds__(64) k_chain(uint32_t* io, int iters) {
  fe a, b;
  for (int i = 0; i < 9; ++i) { a.v[i] = io[threadIdx.x * 9 + i] & 0x1fffffff; b.v[i] = io[640 + threadIdx.x * 9 + i] & 0x1fffffff; }
  for (int it = 0; it < iters; ++it) {
    if (VARIANT == 0) fe_mul(a, a, b);
    else if (VARIANT == 1) fe_mul_rows(a, a, b);
    else if (VARIANT == 2) fe_sq(a, a);
  }
  for (int i = 0; i < 9; ++i) io[(blockIdx.x * 64 + threadIdx.x) * 9 + i] = a.v[i];
}

template <int V>
void run(const char* name, uint32_t* d, int waves_per_simd) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int iters = 20000;
  const int blocks = 256 * 4 * waves_per_simd;
  for (int rep = 0; rep < 2; ++rep) {
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(k_chain<V>, dim3(blocks), dim3(64), 0, 0, d, iters);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    if (rep) printf("%-10s waves/SIMD %d : %8.3f ms  -> %7.1f ns per dependent op per wave, %6.1f G field-ops/s chip\n", name, waves_per_simd, ms,
                    ms * 1e6 / iters, (double)blocks * 64 * iters / (ms * 1e-3) * 1e-9);
  }
}
int main() {
  uint32_t* d; CK(hipMalloc(&d, 256 * 4 * 8 * 64 * 9 * 4 + 65536)); CK(hipMemset(d, 0x5a, 65536));
  for (int w : {1, 2, 4, 8}) { run<0>("mul-col", d, w); run<1>("mul-row", d, w); run<2>("sq", d, w); }
  return 0;
}
