// gfx950: what does one Keccak-f[1600] cost a lone wavefront, and does spreading a STROBE state over more lanes help?  (profiles/r06_keccak_microbench.txt)
//
// The Merlin transcripts of a batch are hash CHAINS: 59 permutations per CMZ proof for the prover (prover.rs:76-112 over mod.rs:165-228), 24 dependent
// rounds each, and a batch of 4096 proofs is 128 wavefronts of the shipped kernel (a lane PAIR per proof, transcript_kernels.h) on a chip with 1024 SIMDs.
// VERDICT r5 item 1 asks for a "wavefront-cooperative" permutation: the state spread over more lanes.  This file measures the candidates before anything
// is built into the product:
//   pair      the shipped layout: lane h of a pair holds the h-th 32-bit half of all 25 words; the only cross-lane traffic is the partner's half for the
//             29 rotations of a round (one DPP quad_perm move each).  125 VALU instructions per round and lane.
//   pair_u    the same with the 24 rounds unrolled (round constants as literals, no s_load / loop branch).
//   word25    one WORD per lane (25 of 32 lanes, two states per wavefront): theta's column sums, the D fetch and pi + chi's three operand fetches are
//             ds_bpermute_b32 (gfx9 DPP moves stay inside 16-lane rows and cannot express a 5-cyclic shift, a 5 x 5 transpose or pi); 16 crossbar
//             moves + ~26 VALU instructions per round and lane -- a third of the VALU work, 16 x the lanes.
// Each variant runs CHAIN permutations back to back on (a) 128 wavefronts' worth of states per launch as in the product (4096 states; word25: 2048
// wavefronts), (b) 4 x that.  Reported: microseconds per permutation of the chain.  All variants must agree bit for bit on the final states.
//
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/microbench/keccak_lat.hip -o tools/microbench/keccak_lat
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1);} } while (0)

__device__ __constant__ uint64_t RC_D[24] = {
    0x0000000000000001ULL, 0x0000000000008082ULL, 0x800000000000808aULL, 0x8000000080008000ULL, 0x000000000000808bULL,
    0x0000000080000001ULL, 0x8000000080008081ULL, 0x8000000000008009ULL, 0x000000000000008aULL, 0x0000000000000088ULL,
    0x0000000080008009ULL, 0x000000008000000aULL, 0x000000008000808bULL, 0x800000000000008bULL, 0x8000000000008089ULL,
    0x8000000000008003ULL, 0x8000000000008002ULL, 0x8000000000000080ULL, 0x000000000000800aULL, 0x800000008000000aULL,
    0x8000000080008081ULL, 0x8000000000008080ULL, 0x0000000080000001ULL, 0x8000000080008008ULL};
static const uint64_t RC_H[24] = {
    0x0000000000000001ULL, 0x0000000000008082ULL, 0x800000000000808aULL, 0x8000000080008000ULL, 0x000000000000808bULL,
    0x0000000080000001ULL, 0x8000000080008081ULL, 0x8000000000008009ULL, 0x000000000000008aULL, 0x0000000000000088ULL,
    0x0000000080008009ULL, 0x000000008000000aULL, 0x000000008000808bULL, 0x800000000000008bULL, 0x8000000000008089ULL,
    0x8000000000008003ULL, 0x8000000000008002ULL, 0x8000000000000080ULL, 0x000000000000800aULL, 0x800000008000000aULL,
    0x8000000080008081ULL, 0x8000000000008080ULL, 0x0000000080000001ULL, 0x8000000080008008ULL};
#define RHO_INIT {0, 1, 62, 28, 27, 36, 44, 6, 55, 20, 3, 10, 43, 25, 39, 41, 45, 15, 21, 8, 18, 2, 61, 56, 14}

__device__ __forceinline__ uint32_t xor3(uint32_t a, uint32_t b, uint32_t c) { return (uint32_t)__builtin_amdgcn_bitop3_b32((int)a, (int)b, (int)c, 0x96); }
__device__ __forceinline__ uint32_t chi3(uint32_t a, uint32_t b, uint32_t c) { return (uint32_t)__builtin_amdgcn_bitop3_b32((int)a, (int)b, (int)c, 0xD2); }   // a ^ (~b & c)
__device__ __forceinline__ uint32_t pair_swap(uint32_t v) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xf, 0xf, true); }
__device__ __forceinline__ uint32_t rotl_half(uint32_t mine, uint32_t partner, int R) {
  const int n = R & 31;
  if (R & 32) return n ? __builtin_amdgcn_alignbit(partner, mine, 32 - n) : partner;
  return n ? __builtin_amdgcn_alignbit(mine, partner, 32 - n) : mine;
}

template <bool UNROLL>
__device__ __forceinline__ void keccak_pair(uint32_t a[25], uint32_t h) {
  constexpr int RHO[25] = RHO_INIT;
  auto round_fn = [&](uint64_t rc) {
    uint32_t c[5], b[25];
#pragma unroll
    for (int x = 0; x < 5; ++x) c[x] = xor3(xor3(a[x], a[x + 5], a[x + 10]), a[x + 15], a[x + 20]);
#pragma unroll
    for (int x = 0; x < 5; ++x) {
      const uint32_t cn = c[(x + 1) % 5];
      const uint32_t d = c[(x + 4) % 5] ^ rotl_half(cn, pair_swap(cn), 1);
#pragma unroll
      for (int y = 0; y < 5; ++y) a[x + 5 * y] ^= d;
    }
#pragma unroll
    for (int y = 0; y < 5; ++y)
#pragma unroll
      for (int x = 0; x < 5; ++x) {
        const uint32_t v = a[x + 5 * y];
        b[y + 5 * ((2 * x + 3 * y) % 5)] = RHO[x + 5 * y] ? rotl_half(v, pair_swap(v), RHO[x + 5 * y]) : v;
      }
#pragma unroll
    for (int y = 0; y < 5; ++y)
#pragma unroll
      for (int x = 0; x < 5; ++x) a[x + 5 * y] = chi3(b[x + 5 * y], b[(x + 1) % 5 + 5 * y], b[(x + 2) % 5 + 5 * y]);
    a[0] ^= h ? (uint32_t)(rc >> 32) : (uint32_t)rc;
  };
  if constexpr (UNROLL) {
#pragma unroll
    for (int r = 0; r < 24; ++r) round_fn(RC_D[r]);
  } else {
#pragma unroll 1
    for (int r = 0; r < 24; ++r) round_fn(RC_D[r]);
  }
}

// states: [n][25] uint64.  pair layout: 32 states per wavefront
template <bool UNROLL>
__global__ void __launch_bounds__(64) k_pair(uint64_t* __restrict__ st, uint32_t n, int chain, uint64_t* __restrict__ cyc) {
  const uint32_t lane = threadIdx.x, h = lane & 1, j = blockIdx.x * 32 + (lane >> 1);
  if (j >= n) return;
  uint32_t* w = reinterpret_cast<uint32_t*>(st + 25 * (size_t)j);
  uint32_t a[25];
#pragma unroll
  for (int i = 0; i < 25; ++i) a[i] = w[2 * i + h];
  const uint64_t t0 = __builtin_readcyclecounter();
#pragma unroll 1
  for (int k = 0; k < chain; ++k) keccak_pair<UNROLL>(a, h);
  const uint64_t t1 = __builtin_readcyclecounter();
#pragma unroll
  for (int i = 0; i < 25; ++i) w[2 * i + h] = a[i];
  if (lane == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

// one word per lane: lane l of a 32-lane half = word l (x = l % 5, y = l / 5), l < 25; lanes 25..31 idle but present (crossbar sources must be active)
__device__ __forceinline__ uint32_t bperm(int byte_addr, uint32_t v) { return (uint32_t)__builtin_amdgcn_ds_bpermute(byte_addr, (int)v); }
__global__ void __launch_bounds__(64) k_word25(uint64_t* __restrict__ st, uint32_t n, int chain, uint64_t* __restrict__ cyc) {
  constexpr int RHO[25] = RHO_INIT;
  const uint32_t lane = threadIdx.x, half = lane >> 5, l = lane & 31, base = half * 32;
  const uint32_t j = blockIdx.x * 2 + half;
  const bool live = l < 25 && j < n;
  const uint32_t x = l % 5, y = l / 5;
  uint32_t lo = 0, hi = 0;
  if (live) { const uint64_t v = st[25 * (size_t)j + l]; lo = (uint32_t)v; hi = (uint32_t)(v >> 32); }
  // crossbar source lanes (byte addresses), all inside this half
  auto L = [&](uint32_t xx, uint32_t yy) { return (int)(4 * (base + (xx % 5) + 5 * (yy % 5))); };
  const int up1 = L(x, y + 1), up2 = L(x, y + 2), up4 = L(x, y + 4), xm = L(x + 4, y), xp = L(x + 1, y);
  // after rho, lane (X, Y) needs B[X][Y], B[X+1][Y], B[X+2][Y] with B[X][Y] = rot(A[(X + 3 Y) % 5][X])
  const int s0 = L(x + 3 * y, x), s1 = L(x + 1 + 3 * y, x + 1), s2 = L(x + 2 + 3 * y, x + 2);
  uint32_t rho = 0;
#pragma unroll
  for (int i = 0; i < 25; ++i) rho = (l == (uint32_t)i) ? (uint32_t)RHO[i] : rho;
  const uint32_t m = (64 - rho) & 63;                      // rotl by rho = rotr by m
  const bool sw = (m & 32) != 0;
  const uint32_t ms = m & 31;
  const uint64_t t0 = __builtin_readcyclecounter();
#pragma unroll 1
  for (int k = 0; k < chain; ++k) {
#pragma unroll 1
    for (int r = 0; r < 24; ++r) {
      // theta: column parity in every lane of the column
      uint32_t a_lo = lo ^ bperm(up1, lo), a_hi = hi ^ bperm(up1, hi);
      const uint32_t q_lo = bperm(up4, lo), q_hi = bperm(up4, hi);
      uint32_t c_lo = xor3(a_lo, bperm(up2, a_lo), q_lo), c_hi = xor3(a_hi, bperm(up2, a_hi), q_hi);
      const uint32_t m_lo = bperm(xm, c_lo), m_hi = bperm(xm, c_hi), p_lo = bperm(xp, c_lo), p_hi = bperm(xp, c_hi);
      lo = xor3(lo, m_lo, __builtin_amdgcn_alignbit(p_lo, p_hi, 31));
      hi = xor3(hi, m_hi, __builtin_amdgcn_alignbit(p_hi, p_lo, 31));
      // rho: rotr by m (per-lane amount)
      const uint32_t u = sw ? hi : lo, v = sw ? lo : hi;   // (v:u) is the word after the optional 32-bit swap
      const uint32_t r_lo = __builtin_amdgcn_alignbit(v, u, ms), r_hi = __builtin_amdgcn_alignbit(u, v, ms);
      // pi + chi
      const uint32_t b0l = bperm(s0, r_lo), b0h = bperm(s0, r_hi), b1l = bperm(s1, r_lo), b1h = bperm(s1, r_hi), b2l = bperm(s2, r_lo), b2h = bperm(s2, r_hi);
      lo = chi3(b0l, b1l, b2l);
      hi = chi3(b0h, b1h, b2h);
      const uint64_t rc = RC_D[r];
      if (l == 0) { lo ^= (uint32_t)rc; hi ^= (uint32_t)(rc >> 32); }
    }
  }
  const uint64_t t1 = __builtin_readcyclecounter();
  if (live) st[25 * (size_t)j + l] = (uint64_t)hi << 32 | lo;
  if (lane == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

static void keccak_host(uint64_t a[25]) {
  static const int RHO[25] = RHO_INIT;
  for (int r = 0; r < 24; ++r) {
    uint64_t c[5], b[25];
    for (int x = 0; x < 5; ++x) c[x] = a[x] ^ a[x + 5] ^ a[x + 10] ^ a[x + 15] ^ a[x + 20];
    for (int x = 0; x < 5; ++x) {
      const uint64_t cn = c[(x + 1) % 5], d = c[(x + 4) % 5] ^ (cn << 1 | cn >> 63);
      for (int y = 0; y < 5; ++y) a[x + 5 * y] ^= d;
    }
    for (int y = 0; y < 5; ++y)
      for (int x = 0; x < 5; ++x) {
        const uint64_t v = a[x + 5 * y];
        const int R = RHO[x + 5 * y];
        b[y + 5 * ((2 * x + 3 * y) % 5)] = R ? (v << R | v >> (64 - R)) : v;
      }
    for (int y = 0; y < 5; ++y)
      for (int x = 0; x < 5; ++x) a[x + 5 * y] = b[x + 5 * y] ^ (~b[(x + 1) % 5 + 5 * y] & b[(x + 2) % 5 + 5 * y]);
    a[0] ^= RC_H[r];
  }
}

int main() {
  const int CHAIN = 64;
  for (uint32_t n : {4096u, 16384u}) {
    std::vector<uint64_t> init(25 * (size_t)n), ref(25 * (size_t)n);
    uint64_t s = 0x9e3779b97f4a7c15ull;
    for (auto& v : init) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; v = s; }
    ref = init;
    for (uint32_t j = 0; j < 8; ++j) for (int k = 0; k < CHAIN; ++k) keccak_host(&ref[25 * (size_t)j]);      // (the first eight states are checked against the host)
    uint64_t *d = nullptr, *dc = nullptr;
    CK(hipMalloc(&d, init.size() * 8));
    CK(hipMalloc(&dc, 8));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    std::vector<uint64_t> first;
    for (int variant = 0; variant < 3; ++variant) {
      float best = 1e9f;
      uint64_t cyc = 0;
      std::vector<uint64_t> out(init.size());
      for (int rep = 0; rep < 5; ++rep) {
        CK(hipMemcpy(d, init.data(), init.size() * 8, hipMemcpyHostToDevice));
        CK(hipEventRecord(e0));
        if (variant == 0) hipLaunchKernelGGL(k_pair<false>, dim3((n + 31) / 32), dim3(64), 0, 0, d, n, CHAIN, dc);
        if (variant == 1) hipLaunchKernelGGL(k_pair<true>, dim3((n + 31) / 32), dim3(64), 0, 0, d, n, CHAIN, dc);
        if (variant == 2) hipLaunchKernelGGL(k_word25, dim3((n + 1) / 2), dim3(64), 0, 0, d, n, CHAIN, dc);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms = 0;
        CK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
      }
      CK(hipMemcpy(out.data(), d, out.size() * 8, hipMemcpyDeviceToHost));
      CK(hipMemcpy(&cyc, dc, 8, hipMemcpyDeviceToHost));
      bool ok = memcmp(out.data(), ref.data(), 25 * 8 * 8) == 0;
      if (variant == 0) first = out; else ok = ok && out == first;
      static const char* names[3] = {"pair (shipped layout, rounds in a loop)", "pair_u (24 rounds unrolled)", "word25 (one word per lane, ds_bpermute_b32)"};
      printf("states %6u  %-46s %8.2f us per permutation (launch %8.3f ms / %d)   wavefront 0: %7.1f cycles per round   %s\n", n, names[variant], best * 1e3 / CHAIN, best, CHAIN,
             (double)cyc / (24.0 * CHAIN), ok ? "bytes ok" : "MISMATCH");
    }
    CK(hipFree(d)); CK(hipFree(dc));
  }
  return 0;
}
