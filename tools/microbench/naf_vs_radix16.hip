// verify_compact's single-use points (verifier.rs:96-110: one variable-time scalar multiplication per left-hand side): is a sparse
// recoding (width-5 NAF, curve25519-dalek's choice for vartime Straus on a CPU: ~43 additions per 256-bit scalar) better than the
// signed radix-16 ladder (65 additions, fixed positions) on a 64-lane SIMT machine?  Every lane multiplies its own point by its own
// scalar; tables (8 entries of cached multiples: 1P..8P for radix 16, 1P,3P..15P for NAF-5) sit in global memory as in the product.
// A wavefront executes an addition at a bit position as soon as ANY lane has a non-zero digit there, so the sparse form only wins
// if lanes agree on positions.  Three scalar populations: random per lane; one scalar per group of 11 lanes (verify_compact's -c on the
// eleven left-hand sides of one CMZ proof); one scalar for the whole wavefront (best case for NAF).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <vector>
#include "../../zkp_amd/csrc/dev_layout.h"
namespace zkp {
__device__ __forceinline__ void sc_add_pattern(uint32_t e[8], uint32_t& top, const uint32_t s[8], uint32_t pattern) {
  uint64_t c = 0;
  for (int i = 0; i < 8; ++i) { c += (uint64_t)s[i] + pattern; e[i] = (uint32_t)c; c >>= 32; }
  top = (uint32_t)c;
}
}
#include "../../zkp_amd/csrc/hot_tables.h"
#include "../../zkp_amd/csrc/quad.h"
#include "../../zkp_amd/csrc/comb_tables.h"
using namespace zkp;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)

// tbl[lane][k] = (k + 1) P (radix 16) or (2 k + 1) P (NAF-5)
template <bool NAF>
__global__ void __launch_bounds__(256, 2) k_tables(uint32_t n, const uint8_t* enc, dev_ext* tbl) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  uint32_t w[8];
  load_vec<2>(w, enc + 32 * (size_t)(i % 64));
  ge_p3 P, P2, m;
  ristretto_decode(P, w);
  ge_cached c1, c2, c;
  ge_to_cached(c1, P);
  ge_double<true>(P2, P);
  ge_to_cached(c2, P2);
  dev_ext* t = tbl + (size_t)i * 8;
  store_comb_entry(t, c1);
  m = P;
  for (int k = 1; k < 8; ++k) {
    ge_add_cached(m, m, NAF ? c2 : c1);          // NAF: 3P, 5P, ...; radix 16: 2P, 3P, ...
    ge_to_cached(c, m);
    store_comb_entry(t + k, c);
  }
}
// signed radix 16: 64 windows x (4 doublings + 1 table addition), zero digits skipped (variable time)
__global__ void __launch_bounds__(256, 2) k_radix16(uint32_t n, const uint32_t* scalars, const dev_ext* tbl, dev_ext* out) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  uint32_t s[8], e[8], top;
  for (int k = 0; k < 8; ++k) s[k] = scalars[8 * (size_t)i + k];
  sc_add_pattern(e, top, s, 0x88888888u);
  const dev_ext* t = tbl + (size_t)i * 8;
  ge_p3 acc;
  ge_identity(acc);
#pragma unroll 1
  for (int j = 7; j >= 0; --j) {
    uint32_t cur = j == 7 ? e[7] : j == 6 ? e[6] : j == 5 ? e[5] : j == 4 ? e[4] : j == 3 ? e[3] : j == 2 ? e[2] : j == 1 ? e[1] : e[0];
#pragma unroll 1
    for (int k = 0; k < 8; ++k) {
      ge_double4(acc);
      const uint32_t nib = cur >> 28;
      cur <<= 4;
      const uint32_t neg = (uint32_t)(nib < 8u), mag = neg ? 8u - nib : nib - 8u;
      if (mag) {
        ge_cached sel;
        load_comb_entry(sel, t + (mag - 1));
        ge_cached_cneg(sel, neg);
        ge_add_cached(acc, acc, sel);
      }
    }
  }
  store_ext(out + i, acc);
}
// width-5 NAF: digits in {0, +-1, +-3, .., +-15}, recoded on the fly from the top (precomputed per lane into a byte string in global
// memory by the host would hide the recoding cost: it is precomputed here too, naf[i][256] signed bytes)
__global__ void __launch_bounds__(256, 2) k_naf5(uint32_t n, const int8_t* naf, const dev_ext* tbl, dev_ext* out) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int8_t* d = naf + 256 * (size_t)i;
  const dev_ext* t = tbl + (size_t)i * 8;
  ge_p3 acc;
  ge_identity(acc);
#pragma unroll 1
  for (int b = 255; b >= 0; --b) {
    ge_double<true>(acc, acc);
    const int v = d[b];
    if (v) {
      const uint32_t neg = (uint32_t)(v < 0), mag = (uint32_t)(v < 0 ? -v : v);      // odd, 1..15
      ge_cached sel;
      load_comb_entry(sel, t + (mag >> 1));
      ge_cached_cneg(sel, neg);
      ge_add_cached(acc, acc, sel);
    }
  }
  store_ext(out + i, acc);
}

static void naf5(int8_t out[256], const uint32_t s[8]) {
  uint64_t x[5] = {0, 0, 0, 0, 0};
  for (int i = 0; i < 4; ++i) x[i] = (uint64_t)s[2 * i] | ((uint64_t)s[2 * i + 1] << 32);
  for (int i = 0; i < 256; ++i) out[i] = 0;
  int pos = 0, carry = 0;
  while (pos < 256) {
    const int idx = pos / 64, bit = pos % 64;
    uint64_t buf = bit < 59 ? x[idx] >> bit : (x[idx] >> bit) | (x[idx + 1] << (64 - bit));
    const int window = carry + (int)(buf & 31);
    if ((window & 1) == 0) { ++pos; continue; }
    if (window < 16) { carry = 0; out[pos] = (int8_t)window; } else { carry = 1; out[pos] = (int8_t)(window - 32); }
    pos += 5;
  }
}

int main() {
  const uint32_t n = 256 * 1024;                       // 1024 blocks: 1 wavefront per SIMD x 4
  std::vector<uint8_t> enc(32 * 64);
  {   // 64 valid encodings: small multiples of the basepoint computed on the device would need more code; reuse the basepoint encoding
    const uint8_t b[32] = {0xe2,0xf2,0xae,0x0a,0x6a,0xbc,0x4e,0x71,0xa8,0x84,0xa9,0x61,0xc5,0x00,0x51,0x5f,0x58,0xe3,0x0b,0x6a,0xa5,0x82,0xdd,0x8d,0xb6,0xa6,0x59,0x45,0xe0,0x8d,0x2d,0x76};
    for (int i = 0; i < 64; ++i) for (int k = 0; k < 32; ++k) enc[32 * i + k] = b[k];
  }
  uint8_t* d_enc; dev_ext *d_t16, *d_tn, *d_out; uint32_t* d_sc; int8_t* d_naf;
  CK(hipMalloc(&d_enc, enc.size())); CK(hipMemcpy(d_enc, enc.data(), enc.size(), hipMemcpyHostToDevice));
  CK(hipMalloc(&d_t16, sizeof(dev_ext) * 8 * (size_t)n)); CK(hipMalloc(&d_tn, sizeof(dev_ext) * 8 * (size_t)n)); CK(hipMalloc(&d_out, sizeof(dev_ext) * (size_t)n));
  CK(hipMalloc(&d_sc, 32 * (size_t)n)); CK(hipMalloc(&d_naf, 256 * (size_t)n));
  hipLaunchKernelGGL(k_tables<false>, dim3(n / 256), dim3(256), 0, 0, n, d_enc, d_t16);
  hipLaunchKernelGGL(k_tables<true>, dim3(n / 256), dim3(256), 0, 0, n, d_enc, d_tn);
  CK(hipDeviceSynchronize());
  std::vector<uint32_t> sc(8 * (size_t)n);
  std::vector<int8_t> naf(256 * (size_t)n);
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  printf("# %u lanes (1024 blocks of 256), one 253-bit scalar multiplication per lane, table of 8 cached multiples per lane in HBM\n", n);
  for (int group : {1, 11, 64}) {
    srand(7);
    for (uint32_t i = 0; i < n; ++i) {
      const uint32_t g = i - i % group;
      for (int k = 0; k < 8; ++k) sc[8 * (size_t)i + k] = (i == g) ? ((uint32_t)rand() << 16) ^ (uint32_t)rand() : sc[8 * (size_t)g + k];
      if (i == g) sc[8 * (size_t)i + 7] &= 0x0fffffffu;
      naf5(&naf[256 * (size_t)i], &sc[8 * (size_t)i]);
    }
    size_t nz = 0;
    for (size_t q = 0; q < 256 * (size_t)4096; ++q) nz += naf[q] != 0;
    CK(hipMemcpy(d_sc, sc.data(), 32 * (size_t)n, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_naf, naf.data(), 256 * (size_t)n, hipMemcpyHostToDevice));
    float ms16 = 0, msn = 0;
    for (int rep = 0; rep < 3; ++rep) {
      CK(hipEventRecord(e0)); hipLaunchKernelGGL(k_radix16, dim3(n / 256), dim3(256), 0, 0, n, d_sc, d_t16, d_out); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      CK(hipEventElapsedTime(&ms16, e0, e1));
      CK(hipEventRecord(e0)); hipLaunchKernelGGL(k_naf5, dim3(n / 256), dim3(256), 0, 0, n, d_naf, d_tn, d_out); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      CK(hipEventElapsedTime(&msn, e0, e1));
    }
    printf("lanes sharing a scalar: %2d | signed radix 16: %7.3f ms | NAF-5 (%.1f non-zero digits per scalar): %7.3f ms | NAF / radix-16 = %.2f\n", group, ms16,
           (double)nz / 4096, msn, msn / ms16);
  }
  return 0;
}
