// gfx950: is ds_bpermute_b32 a constant-time table look-up?  (profiles/r05_bpermute_microbench.txt)
//
// Round 5 replaces the secret-indexed LDS reads of the prover's fixed-base and grouped comb walks by a lane crossbar: a wavefront keeps one
// table row in registers (entry e in lane e) and every lane fetches the entry its digit names with ds_bpermute_b32 (source lane = digit).  The
// instruction has no memory address -- but it runs on the LDS hardware, so before relying on it: does its time depend on WHICH lanes are read?
//   part 1  cycles per instruction for source patterns that would conflict under every bank model the LDS has for real addresses:
//           identity, broadcast, two sources 32 lanes apart (the 32-bank model of ds_read_b32: same bank, different address), four sources 16 apart,
//           sources restricted to lanes 0..31, random with repeats, random permutation; lone wavefront (latency) and 8 wavefronts per SIMD (throughput);
//   part 2  what 27 / 36 look-ups cost NEXT TO a point addition's worth of v_mad_u64_u32 (1,000 per look-up group), two wavefronts per SIMD as in k_terms_split.
// SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE of the same kernels: tools/bpermute_rate.sh.
//
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/microbench/bpermute_rate.hip -o tools/microbench/bpermute_rate
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1);} } while (0)
constexpr int ITER = 512, NB = 16;                 // NB independent look-ups per iteration

// pattern p: lane l reads lane pat[64 p + l]
template <int PAT>
__global__ void __launch_bounds__(256) k_bperm(const uint32_t* __restrict__ pat, uint32_t* __restrict__ out, uint64_t* __restrict__ cyc) {
  const uint32_t lane = threadIdx.x & 63u;
  const int addr = (int)(pat[64 * PAT + lane] * 4u);
  int x[NB];
#pragma unroll
  for (int k = 0; k < NB; ++k) x[k] = (int)(threadIdx.x * 2654435761u + k);
  const uint64_t t0 = __builtin_readcyclecounter();
#pragma unroll 1
  for (int i = 0; i < ITER; ++i) {
#pragma unroll
    for (int k = 0; k < NB; ++k) x[k] = __builtin_amdgcn_ds_bpermute(addr, x[k]);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
  const uint64_t t1 = __builtin_readcyclecounter();
  int r = 0;
#pragma unroll
  for (int k = 0; k < NB; ++k) r ^= x[k];
  out[blockIdx.x * blockDim.x + threadIdx.x] = (uint32_t)r;
  if (lane == 0) cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}

// part 2: LOOKUPS bpermutes, then 1,000 multiply-adds on four independent chains; LOOKUPS = 0: the multiply-adds alone
template <int LOOKUPS>
__global__ void __launch_bounds__(256, 2) k_mix(const uint32_t* __restrict__ pat, uint32_t* __restrict__ out, uint32_t a, uint32_t b, int pattern) {
  const uint32_t lane = threadIdx.x & 63u;
  const int addr = (int)(pat[64 * pattern + lane] * 4u);
  uint64_t y0 = threadIdx.x, y1 = y0 + 1, y2 = y0 + 2, y3 = y0 + 3;
  const uint32_t va = a + threadIdx.x, vb = b ^ threadIdx.x;
  int x[36];
#pragma unroll
  for (int k = 0; k < 36; ++k) x[k] = (int)(threadIdx.x * 2654435761u + k);
#pragma unroll 1
  for (int i = 0; i < 256; ++i) {
    if constexpr (LOOKUPS > 0) {
#pragma unroll
      for (int k = 0; k < LOOKUPS; ++k) x[k] = __builtin_amdgcn_ds_bpermute(addr, x[k] + (int)(uint32_t)y0);
    }
#pragma unroll 1
    for (int j = 0; j < 25; ++j) {
      asm volatile(
        "v_mad_u64_u32 %0, vcc, %4, %5, %0\n v_mad_u64_u32 %1, vcc, %4, %5, %1\n v_mad_u64_u32 %2, vcc, %4, %5, %2\n v_mad_u64_u32 %3, vcc, %4, %5, %3\n"
        "v_mad_u64_u32 %0, vcc, %4, %5, %0\n v_mad_u64_u32 %1, vcc, %4, %5, %1\n v_mad_u64_u32 %2, vcc, %4, %5, %2\n v_mad_u64_u32 %3, vcc, %4, %5, %3\n"
        "v_mad_u64_u32 %0, vcc, %4, %5, %0\n v_mad_u64_u32 %1, vcc, %4, %5, %1\n v_mad_u64_u32 %2, vcc, %4, %5, %2\n v_mad_u64_u32 %3, vcc, %4, %5, %3\n"
        "v_mad_u64_u32 %0, vcc, %4, %5, %0\n v_mad_u64_u32 %1, vcc, %4, %5, %1\n v_mad_u64_u32 %2, vcc, %4, %5, %2\n v_mad_u64_u32 %3, vcc, %4, %5, %3\n"
        "v_mad_u64_u32 %0, vcc, %4, %5, %0\n v_mad_u64_u32 %1, vcc, %4, %5, %1\n v_mad_u64_u32 %2, vcc, %4, %5, %2\n v_mad_u64_u32 %3, vcc, %4, %5, %3\n"
        "v_mad_u64_u32 %0, vcc, %4, %5, %0\n v_mad_u64_u32 %1, vcc, %4, %5, %1\n v_mad_u64_u32 %2, vcc, %4, %5, %2\n v_mad_u64_u32 %3, vcc, %4, %5, %3\n"
        "v_mad_u64_u32 %0, vcc, %4, %5, %0\n v_mad_u64_u32 %1, vcc, %4, %5, %1\n v_mad_u64_u32 %2, vcc, %4, %5, %2\n v_mad_u64_u32 %3, vcc, %4, %5, %3\n"
        "v_mad_u64_u32 %0, vcc, %4, %5, %0\n v_mad_u64_u32 %1, vcc, %4, %5, %1\n v_mad_u64_u32 %2, vcc, %4, %5, %2\n v_mad_u64_u32 %3, vcc, %4, %5, %3\n"
        "v_mad_u64_u32 %0, vcc, %4, %5, %0\n v_mad_u64_u32 %1, vcc, %4, %5, %1\n v_mad_u64_u32 %2, vcc, %4, %5, %2\n v_mad_u64_u32 %3, vcc, %4, %5, %3\n"
        "v_mad_u64_u32 %0, vcc, %4, %5, %0\n v_mad_u64_u32 %1, vcc, %4, %5, %1\n v_mad_u64_u32 %2, vcc, %4, %5, %2\n v_mad_u64_u32 %3, vcc, %4, %5, %3\n"
        : "+v"(y0), "+v"(y1), "+v"(y2), "+v"(y3) : "v"(va), "v"(vb) : "vcc");
    }
  }
  int r = 0;
#pragma unroll
  for (int k = 0; k < 36; ++k) r ^= x[k];
  const uint64_t s = y0 ^ y1 ^ y2 ^ y3;
  out[blockIdx.x * blockDim.x + threadIdx.x] = (uint32_t)r ^ (uint32_t)s ^ (uint32_t)(s >> 32);
}

typedef void (*kern_t)(const uint32_t*, uint32_t*, uint64_t*);
static const char* NAMES[] = {"identity", "broadcast lane 0", "lanes 0 / 32 alternating", "lanes 0,16,32,48 cycling", "l mod 32 (lower half only)",
                              "random with repeats", "random permutation", "l xor 32 (halves swapped)"};
constexpr int NPAT = 8;
template <int P> static kern_t kern() { return k_bperm<P>; }

int main() {
  hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  std::vector<uint32_t> pat(64 * NPAT);
  srand(12345);
  for (int l = 0; l < 64; ++l) {
    pat[0 * 64 + l] = l;
    pat[1 * 64 + l] = 0;
    pat[2 * 64 + l] = (l & 1) * 32;
    pat[3 * 64 + l] = (l & 3) * 16;
    pat[4 * 64 + l] = l & 31;
    pat[5 * 64 + l] = rand() & 63;
    pat[6 * 64 + l] = l;
    pat[7 * 64 + l] = l ^ 32;
  }
  for (int l = 63; l > 0; --l) { const int j = rand() % (l + 1); std::swap(pat[6 * 64 + l], pat[6 * 64 + j]); }
  uint32_t *d_pat, *out; uint64_t* cyc;
  const int max_blocks = cus * 8;
  CK(hipMalloc(&d_pat, pat.size() * 4)); CK(hipMemcpy(d_pat, pat.data(), pat.size() * 4, hipMemcpyHostToDevice));
  CK(hipMalloc(&out, (size_t)max_blocks * 256 * 4)); CK(hipMalloc(&cyc, (size_t)max_blocks * 4 * 8));
  kern_t ks[NPAT] = {kern<0>(), kern<1>(), kern<2>(), kern<3>(), kern<4>(), kern<5>(), kern<6>(), kern<7>()};
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  printf("# part 1: ds_bpermute_b32, %d independent look-ups per s_waitcnt, %d iterations (lone wavefront: s_memtime shader cycles of wavefront 0)\n", NB, ITER);
  printf("%-30s %26s %30s\n", "source pattern", "lone wavefront: cyc / look-up", "8 wavefronts / SIMD: look-ups / ns / CU");
  for (int p = 0; p < NPAT; ++p) {
    float lone = 0, full = 0;
    for (int rep = 0; rep < 3; ++rep) {
      CK(hipEventRecord(e0));
      hipLaunchKernelGGL(ks[p], dim3(1), dim3(64), 0, 0, d_pat, out, cyc);
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      uint64_t c; CK(hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost));
      lone = (float)c / (ITER * NB);
      CK(hipEventRecord(e0));
      hipLaunchKernelGGL(ks[p], dim3(max_blocks), dim3(256), 0, 0, d_pat, out, cyc);
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      CK(hipEventElapsedTime(&ms, e0, e1));
      full = (float)((double)ITER * NB * 32 / (ms * 1e6));   // 32 wavefronts per CU
    }
    printf("%-30s %26.3f %30.4f\n", NAMES[p], lone, full);
  }
  printf("# part 2: L look-ups + 1,000 v_mad_u64_u32 per iteration, 2 wavefronts / SIMD, 256 iterations: ms per launch (pattern: random with repeats | broadcast | 0 / 32 alternating)\n");
  auto run_mix = [&](void (*k)(const uint32_t*, uint32_t*, uint32_t, uint32_t, int), int pattern) {
    float best = 1e9f;
    for (int rep = 0; rep < 4; ++rep) {
      CK(hipEventRecord(e0));
      hipLaunchKernelGGL(k, dim3(cus * 2), dim3(256), 0, 0, d_pat, out, 3u, 5u, pattern);
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      if (rep && ms < best) best = ms;
    }
    return best;
  };
  printf("%-12s %10s %10s %10s\n", "look-ups", "random", "broadcast", "0/32");
  printf("%-12d %10.3f %10.3f %10.3f\n", 0, run_mix(k_mix<0>, 5), run_mix(k_mix<0>, 1), run_mix(k_mix<0>, 2));
  printf("%-12d %10.3f %10.3f %10.3f\n", 27, run_mix(k_mix<27>, 5), run_mix(k_mix<27>, 1), run_mix(k_mix<27>, 2));
  printf("%-12d %10.3f %10.3f %10.3f\n", 36, run_mix(k_mix<36>, 5), run_mix(k_mix<36>, 1), run_mix(k_mix<36>, 2));
  return 0;
}
