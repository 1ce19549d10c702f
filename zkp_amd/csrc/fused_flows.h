// Fused statement flows: the whole life of a batch of proofs of ONE statement on the device --
// Merlin transcripts (merlin_prog.h), scalar arithmetic mod l (sc25519.h), operand assembly and the MSMs -- so that
// the host only uploads the inputs and downloads proofs / verdicts.  Included at the end of zkp_kernels.hip (one
// translation unit: shares zkp_ctx, the workspace carving and the MSM paths).
//
//   zkp_fused_prove           N x { macros.rs:206-258 build_prover ; prover.rs:76-112 prove_impl }
//   zkp_fused_verify_compact  N x { macros.rs:280-311 build_verifier ; verifier.rs:80-120 }
//   zkp_fused_batch_verify    macros.rs:336-370 ; batch_verifier.rs:67-235
#pragma once
#include "merlin_prog.h"
#include "transcript_kernels.h"

namespace zkp {

// Scalar::from_bytes_mod_order_wide over n 64-byte strings
__global__ void __launch_bounds__(256)
k_wide_reduce(uint32_t n, const uint8_t* __restrict__ wide, uint8_t* __restrict__ out) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  sc lo, hi, r;
  load_vec<2>(lo.v, wide + 64 * (size_t)i);
  load_vec<2>(hi.v, wide + 64 * (size_t)i + 32);
  sc_from_wide(r, lo, hi);
  store_vec<2>(out + 32 * (size_t)i, r.v);
}

// out[i] = -(in[i] mod l)
__global__ void __launch_bounds__(256)
k_neg_reduce(uint32_t n, const uint8_t* in, uint8_t* out) {       // in may equal out
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  sc a, r;
  load_vec<2>(a.v, in + 32 * (size_t)i);
  sc_reduce(r, a);
  sc_neg(r, r);
  store_vec<2>(out + 32 * (size_t)i, r.v);
}

// l <= the 256-bit little-endian value?  (Scalar::from_canonical_bytes / dalek's Deserialize accept only values < l.)
__device__ __forceinline__ uint32_t sc_not_canonical(const uint32_t v[8]) {
  uint64_t br = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) br = (((uint64_t)v[i] - sc_l(i) - br) >> 63) & 1u;
  return br ? 0u : 1u;                              // no borrow: v >= l
}

// The CSR multiscalar job of a batch of proofs of one statement: T terms per proof in nc MSMs.
//   term t of proof j:  scalar = tsc[t] == ~0 ? special[j] : vals[j][tsc[t]];   point = table index of point id tpt[t]
// (prover.rs:94-97 with vals = blindings; verifier.rs:97-106 with vals = responses, special = -c)
// (two kernels: the index half depends on the statement only and runs with the point phase, next to the transcripts)
__global__ void __launch_bounds__(256)
k_stmt_index(uint32_t N, uint32_t T, uint32_t nc, uint32_t ns, const uint32_t* __restrict__ toff, const uint32_t* __restrict__ tpt,
             uint32_t* __restrict__ off, uint32_t* __restrict__ pidx) {
  const size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (g < (size_t)N * nc) off[g] = (uint32_t)((g / nc) * T + toff[g % nc]);
  if (g == 0) off[(size_t)N * nc] = N * T;
  if (g >= (size_t)N * T) return;
  const uint32_t j = (uint32_t)(g / T), p = tpt[g % T];
  pidx[g] = p < ns ? p : ns + (p - ns) * N + j;
}
__global__ void __launch_bounds__(256)
k_stmt_scalars(uint32_t N, uint32_t T, uint32_t m, const uint32_t* __restrict__ tsc, const uint8_t* __restrict__ vals,
               const uint8_t* __restrict__ special, uint8_t* __restrict__ scalars, uint32_t halve_canonical) {
  const size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= (size_t)N * T) return;
  const uint32_t j = (uint32_t)(g / T), s = tsc[g % T];
  const uint8_t* src = s == 0xffffffffu ? special + 32 * (size_t)j : vals + 32 * ((size_t)j * m + s);
  sc a;
  load_vec<2>(a.v, src);
  if (halve_canonical) {           // the term path will encode 2 * H (batched encoder): hand it s / 2; vals are canonical (blindings)
    sc h;
    sc_halve_canonical(h, a);
    a = h;
  }
  store_vec<2>(scalars + 32 * g, a.v);
}
// The prover's version in one launch: blindings = the transcript rng's 64-byte strings mod l (prover.rs:86-92), written for
// k_responses, and the term operands from them.  A block takes P = 256 / max(m, T) whole proofs: first its lanes reduce the
// P m strings (into LDS and to `blind`), then they assemble the P T operands from LDS -- every string is reduced once.
__global__ void __launch_bounds__(256)
k_blind_scalars(uint32_t N, uint32_t T, uint32_t m, uint32_t P, const uint32_t* __restrict__ tsc, const uint8_t* __restrict__ wide, uint8_t* __restrict__ blind,
                uint8_t* __restrict__ scalars, uint32_t halve_canonical) {
  __shared__ uint32_t red[256][8];
  const uint32_t j0 = blockIdx.x * P, tid = threadIdx.x;
  const uint32_t np = min(P, N - j0);                              // proofs of this block (j0 < N by the grid size)
  if (tid < np * m) {
    const size_t v = (size_t)j0 * m + tid;
    sc lo, hi, r;
    load_vec<2>(lo.v, wide + 64 * v);
    load_vec<2>(hi.v, wide + 64 * v + 32);
    sc_from_wide(r, lo, hi);
    store_vec<2>(blind + 32 * v, r.v);
#pragma unroll
    for (int i = 0; i < 8; ++i) red[tid][i] = r.v[i];
  }
  __syncthreads();
  if (tid < np * T) {
    const uint32_t jl = tid / T, k = tid - jl * T;
    sc a;
#pragma unroll
    for (int i = 0; i < 8; ++i) a.v[i] = red[jl * m + tsc[k]][i];
    if (halve_canonical) {           // the term path will encode 2 * H (batched encoder): hand it s / 2
      sc h;
      sc_halve_canonical(h, a);
      a = h;
    }
    store_vec<2>(scalars + 32 * ((size_t)(j0 + jl) * T + k), a.v);
  }
}

// responses  s * c + b  (prover.rs:107-109); c = the 64 challenge bytes mod l (mod.rs:222-227).  A block takes P = 256 / m whole
// proofs: P lanes reduce the challenges (into LDS and to `chal`), then P m lanes compute the responses.
__global__ void __launch_bounds__(256)
k_responses(uint32_t N, uint32_t m, uint32_t P, const uint8_t* __restrict__ secrets, const uint8_t* __restrict__ wchal, uint8_t* __restrict__ chal,
            const uint8_t* __restrict__ blind, uint8_t* __restrict__ resp) {
  __shared__ uint32_t cs[256][8];
  const uint32_t j0 = blockIdx.x * P, tid = threadIdx.x;
  const uint32_t np = min(P, N - j0);
  if (tid < np) {
    const size_t j = (size_t)j0 + tid;
    sc lo, hi, c;
    load_vec<2>(lo.v, wchal + 64 * j);
    load_vec<2>(hi.v, wchal + 64 * j + 32);
    sc_from_wide(c, lo, hi);
    store_vec<2>(chal + 32 * j, c.v);
#pragma unroll
    for (int i = 0; i < 8; ++i) cs[tid][i] = c.v[i];
  }
  __syncthreads();
  if (tid >= np * m) return;
  const size_t g = (size_t)j0 * m + tid;
  sc s, c, b, r;
#pragma unroll
  for (int i = 0; i < 8; ++i) c.v[i] = cs[tid / m][i];
  load_vec<2>(s.v, secrets + 32 * g);
  load_vec<2>(b.v, blind + 32 * g);
  sc_mul(r, s, c);                      // s may be any 256-bit value (first operand), c and b are canonical
  sc_add(r, r, b);
  store_vec<2>(resp + 32 * g, r.v);
}

// (statements with more than 256 secrets: the challenges come reduced from k_wide_reduce)
__global__ void __launch_bounds__(256)
k_responses_wide(uint32_t N, uint32_t m, const uint8_t* __restrict__ secrets, const uint8_t* __restrict__ chal, const uint8_t* __restrict__ blind,
                 uint8_t* __restrict__ resp) {
  const size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= (size_t)N * m) return;
  sc s, c, b, r;
  load_vec<2>(s.v, secrets + 32 * g);
  load_vec<2>(c.v, chal + 32 * (g / m));
  load_vec<2>(b.v, blind + 32 * g);
  sc_mul(r, s, c);
  sc_add(r, r, b);
  store_vec<2>(resp + 32 * g, r.v);
}

// verify_compact verdicts (verifier.rs:87-92, :113-119): 0 = accepted
__global__ void __launch_bounds__(256)
k_verify_finish(uint32_t N, uint32_t nc, uint32_t ns, uint32_t m, const uint8_t* __restrict__ chal, const uint8_t* __restrict__ claimed,
                const uint8_t* __restrict__ responses, const uint8_t* __restrict__ status8, const uint32_t* __restrict__ failed,
                const dev_affine* __restrict__ pts, const uint32_t* __restrict__ unref, uint32_t n_unref, uint8_t* __restrict__ results) {
  const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= N) return;
  uint32_t bad = failed[j];
  for (uint32_t i = 0; i < m; ++i) {              // a response >= l never reaches the reference's verifier (serde rejects it, proofs.rs:14-20)
    uint32_t r[8];
    load_vec<2>(r, responses + 32 * ((size_t)j * m + i));
    bad |= sc_not_canonical(r);
  }
  for (uint32_t k = 0; k < nc; ++k) bad |= status8[(size_t)j * nc + k];
  for (uint32_t u = 0; u < n_unref; ++u) {
    const uint32_t p = unref[u];
    bad |= pts[p < ns ? p : ns + (p - ns) * N + j].valid == 0;
  }
  sc a, b;
  load_vec<2>(a.v, chal + 32 * (size_t)j);
  load_vec<2>(b.v, claimed + 32 * (size_t)j);     // compared as bytes: a non-canonical claim (c + l) is a different Scalar (verifier.rs:115)
  uint32_t d = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) d |= a.v[i] ^ b.v[i];
  results[j] = (bad || d) ? 1 : 0;
}

// commitment rows of the batch-verification operand list: rows[k][j] = commitments[j][k]  (batch_verifier.rs:208-212)
__global__ void __launch_bounds__(256)
k_transpose_commitments(uint32_t N, uint32_t nc, const uint8_t* __restrict__ coms, uint8_t* __restrict__ rows) {
  const size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= (size_t)N * nc) return;
  const size_t j = g / nc, k = g % nc;
  uint32_t w[8];
  load_vec<2>(w, coms + 32 * g);
  store_vec<2>(rows + 32 * (k * N + j), w);
}

// verify_batchable (verifier.rs:144-166), one MSM of K = np + nc terms per proof over  points || commitments:
//   coeffs[np + k] = -r_k ;  coeffs[lhs_k] += r_k * (-c) ;  coeffs[pt] += r_k * resp[sc]        (weights16 [N][nc][16], :153)
// Operand i of proof j: point id i for i < np (static ids first), commitment i - np after that.
__global__ void __launch_bounds__(256)
k_each_coeffs(uint32_t N, uint32_t m, uint32_t ns, uint32_t ni, uint32_t nc, const uint32_t* __restrict__ inc_off,
              const uint32_t* __restrict__ inc_k, const uint32_t* __restrict__ inc_sc, const uint8_t* __restrict__ minus_c,
              const uint8_t* __restrict__ responses, const uint8_t* __restrict__ weights16, uint32_t* __restrict__ off,
              uint8_t* __restrict__ scalars, uint32_t* __restrict__ pidx) {
  const uint32_t np = ns + ni, K = np + nc;
  const size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (g <= N) off[g] = (uint32_t)(g * K);
  if (g >= (size_t)N * K) return;
  const uint32_t j = (uint32_t)(g / K), i = (uint32_t)(g % K);
  sc acc;
  if (i < np) {
    coeff_of_point(acc, i, j, N, m, inc_off, inc_k, inc_sc, minus_c, responses, weights16, 1, nc);
  } else {
    sc r;
    sc_zero(r);
    load_vec<1>(r.v, weights16 + 16 * ((size_t)j * nc + (i - np)));
    sc_neg(acc, r);                                                // verifier.rs:154
  }
  store_vec<2>(scalars + 32 * g, acc.v);
  pidx[g] = i < ns ? i : (i < np ? ns + (i - ns) * N + j : ns + ni * N + j * nc + (i - np));
}
// verdicts of verify_batchable: accepted iff no rejection by the transcript protocol, every point decoded and the MSM
// is the identity (verifier.rs:134-140, :162-172)
__global__ void __launch_bounds__(256)
k_each_finish(uint32_t N, uint32_t m, const uint8_t* __restrict__ out, const uint8_t* __restrict__ status8, const uint32_t* __restrict__ failed,
              const uint8_t* __restrict__ responses, uint8_t* __restrict__ results) {
  const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= N) return;
  uint32_t w[8];
  load_vec<2>(w, out + 32 * (size_t)j);
  uint32_t d = failed[j] | status8[j];
  for (uint32_t i = 0; i < m; ++i) {              // canonical-scalar rule of the proof format (proofs.rs:27-32 under serde)
    uint32_t r[8];
    load_vec<2>(r, responses + 32 * ((size_t)j * m + i));
    d |= sc_not_canonical(r);
  }
#pragma unroll
  for (int i = 0; i < 8; ++i) d |= w[i];
  results[j] = d ? 1 : 0;
}

// ---- verify_batchable: one Straus MSM per proof (verifier.rs:162-166) ---------------------------------------------------------
// A proof's MSM has K = np + nc operands, almost all of them points nobody else multiplies (CMZ: 13 instance points + 11
// commitments, 24 of 36).  Lane-per-term ladders pay 256 doublings PER OPERAND (K x 321 point operations); Straus shares them:
//     acc = 16 acc ; acc += d_i * P_i  for every operand,   64 windows            256 doublings + K x ~60 additions per MSM
// over per-point tables {1 P .. 8 P} (signed radix-16 digits of the sign-folded coefficient: the -r weights of the commitment
// operands are 128-bit after folding, their upper 32 windows add nothing).  Variable time (public data): zero digits are skipped.
// A lone lane per proof is a 2,300-operation dependent chain and 4,096 proofs are 64 wavefronts on 1,024 SIMDs, so below 65,536 proofs
// a proof is split over lanes -- by WINDOWS (k_straus_each_win: 32 lanes of two windows each, all operands, 4 doublings per lane; a quad
// per proof then joins the 32 partial sums with the 252 doublings once), which keeps the doublings shared.  The earlier split by
// OPERANDS (k_straus_each: operand i on lane i mod L, every lane running its own 256 doublings; L = 1 is the lone lane) serves the
// large batches and statements of more than 64 operands.
// Tables: the walk is bound by its gathers (1,440 per CMZ proof; 1.6 GB of tables at 65,536 proofs, far beyond every cache), not by
// arithmetic, so an entry is ONE 128-byte cache line: the cached form (Y + X, Y - X, 2 Z, 2 d T) with every coordinate carried down to
// 256 bits and packed into 32 bytes (the 144-byte limb form straddles two or three lines: 8.96 -> 6.7 ms for the walk of 65,536 proofs).
// Affine entries (112 B, one multiplication less per addition) were measured too: their shared inversion per 256 points costs the table
// kernel 2.5 instead of 0.7 ms, more than the walk gains.
struct alignas(128) straus_entry { uint32_t w[32]; };     // YpX | YmX | Z2 | T2d, 8 words each
// value of a (limbs up to ~2^31) as 256 bits: three carry passes (the first two fold bits >= 255 back with 19), then 29-bit limbs -> words
__device__ __forceinline__ void fe_pack256(uint32_t w[8], const fe& a) {
  uint32_t t[9];
#pragma unroll
  for (int i = 0; i < 9; ++i) t[i] = a.v[i];
#pragma unroll
  for (int pass = 0; pass < 2; ++pass) {
#pragma unroll
    for (int i = 0; i < 8; ++i) { t[i + 1] += t[i] >> 29; t[i] &= FE_M29; }
    const uint32_t c = t[8] >> 23;
    t[8] &= FE_M23;
    t[0] += 19u * c;
  }
#pragma unroll
  for (int i = 0; i < 8; ++i) { t[i + 1] += t[i] >> 29; t[i] &= FE_M29; }       // t[0] + 19: at most one more unit travels up; t[8] <= 2^23
#pragma unroll
  for (int j = 0; j < 8; ++j) w[j] = 0;
#pragma unroll
  for (int i = 0; i < 9; ++i) {
    const int bit = 29 * i, wi = bit >> 5, sh = bit & 31;
    w[wi] |= t[i] << sh;
    if (sh > 3 && wi < 7) w[wi + 1] |= t[i] >> (32 - sh);
  }
}
__device__ __forceinline__ void fe_unpack256(fe& r, const uint32_t w[8]) {
#pragma unroll
  for (int i = 0; i < 9; ++i) {
    const int bit = 29 * i, wi = bit >> 5, sh = bit & 31;
    uint32_t x = w[wi] >> sh;
    if (sh > 3 && wi < 7) x |= w[wi + 1] << (32 - sh);
    r.v[i] = i == 8 ? x : (x & FE_M29);                    // the top limb keeps bit 255 (24 bits)
  }
  FE_TRACK(fe_set_ub_tight(r));
}
__device__ __forceinline__ void store_straus_entry(straus_entry* dst, const ge_cached& c) {
  uint32_t w[32];
  fe_pack256(w, c.YpX); fe_pack256(w + 8, c.YmX); fe_pack256(w + 16, c.Z2); fe_pack256(w + 24, c.T2d);
  store_vec<8>(dst, w);
}
__device__ __forceinline__ void load_straus_entry(ge_cached& c, const straus_entry* src) {
  uint32_t w[32];
  load_vec<8>(w, src);
  fe_unpack256(c.YpX, w); fe_unpack256(c.YmX, w + 8); fe_unpack256(c.Z2, w + 16); fe_unpack256(c.T2d, w + 24);
}
// k_straus_tables: lane per point of the call: tab[p][k] = (k + 1) P_p
__global__ void __launch_bounds__(256, 2)
k_straus_tables(uint32_t n_points, const dev_affine* __restrict__ pts, straus_entry* __restrict__ tab) {
  const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n_points) return;
  ge_p3 P, m2, m3, m4, m;
  ge_cached c1, c;
  load_affine(P, pts + p);                                           // (an undecodable point: garbage multiples, the proof is flagged by k_straus_finish)
  straus_entry* t = tab + (size_t)p * 8;
  ge_to_cached(c1, P);
  store_straus_entry(t + 0, c1);
  ge_double<true>(m2, P);
  ge_to_cached(c, m2); store_straus_entry(t + 1, c);
  ge_add_cached(m3, m2, c1);
  ge_to_cached(c, m3); store_straus_entry(t + 2, c);
  ge_double<true>(m4, m2);
  ge_to_cached(c, m4); store_straus_entry(t + 3, c);
  ge_add_cached(m, m4, c1);
  ge_to_cached(c, m); store_straus_entry(t + 4, c);
  ge_double<true>(m, m3);
  ge_to_cached(c, m); store_straus_entry(t + 5, c);
  ge_add_cached(m, m, c1);
  ge_to_cached(c, m); store_straus_entry(t + 6, c);
  ge_double<true>(m, m4);
  ge_to_cached(c, m); store_straus_entry(t + 7, c);
}
// lane (proof j, part l of L): operands i = l, l + L, ... < K of proof j (operand i: point id i for i < np, commitment i - np after that;
// table index as in k_each_coeffs).  digits = scratch [N K][9] words (recoded coefficient + sign), written and read by the owning lane
// only.  spart[j L + l] = the part's sum.  Dynamic LDS: ceil(K / L) words per lane (the digit words of the current 32-bit slice).
// The entry of the NEXT operand is requested before the current addition starts, the first entry of the next window before the four
// doublings: the gathers (a dependent HBM round trip each) overlap the arithmetic instead of serialising with it.
__global__ void __launch_bounds__(256, 2)
k_straus_each(uint32_t N, uint32_t K, uint32_t L, uint32_t ns, uint32_t ni, uint32_t nc, const uint8_t* __restrict__ scalars /*[N K][32]*/,
              const straus_entry* __restrict__ tab, uint32_t* __restrict__ digits, dev_ext* __restrict__ spart) {
  extern __shared__ uint32_t straus_lds[];
  uint32_t* col = straus_lds + threadIdx.x;                            // word of operand ordinal o at col[256 o]
  const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= N * L) return;
  const uint32_t j = g / L, l = g - j * L, np = ns + ni;
  const size_t base = (size_t)j * K;
  auto table_of = [&](uint32_t i) -> const straus_entry* {
    const size_t p = i < ns ? i : (i < np ? (size_t)ns + (size_t)(i - ns) * N + j : (size_t)ns + (size_t)ni * N + (size_t)j * nc + (i - np));
    return tab + p * 8;
  };
  ge_p3 acc;
  ge_identity(acc);
  bool started = false;                                                // (doubling the identity is skipped: short coefficients start late)
  uint64_t flipmask = 0;                                               // bit o: operand ordinal o was sign-folded (at most 60 operands per lane)
  uint32_t oo = 0;
  for (uint32_t i = l; i < K; i += L, ++oo) {                          // recode this lane's coefficients: signed radix-16, sign folded
    uint32_t sw[8], e[8], top;
    load_vec<2>(sw, scalars + 32 * (base + i));
    const uint32_t flip = sc_fold_sign(sw);                           // s P = (l - s)(-P); canonical s: the folded value is < 2^252
    sc_add_pattern(e, top, sw, 0x88888888u);
    uint32_t* d = digits + 9 * (base + i);
#pragma unroll
    for (int w = 0; w < 8; ++w) d[w] = e[w];
    d[8] = flip;
    flipmask |= (uint64_t)flip << oo;
    if (top) {                                                         // carry out of bit 255 (non-canonical inputs only): one more P at the top
      ge_cached q;
      load_straus_entry(q, table_of(i));
      ge_cached_cneg(q, flip);
      ge_add_cached(acc, acc, q);
      started = true;
    }
  }
  ge_cached qn;                                                        // the prefetched entry of the upcoming operand
  uint32_t magn = 0, negn = 0;
  auto fetch = [&](uint32_t i, uint32_t o, int k) {
    const uint32_t nib = (col[256 * o] >> (4 * k)) & 15u;
    const uint32_t neg = (uint32_t)(nib < 8u);
    magn = neg ? 8u - nib : nib - 8u;
    negn = neg ^ (uint32_t)((flipmask >> o) & 1u);
    if (magn) load_straus_entry(qn, table_of(i) + (magn - 1));
  };
#pragma unroll 1
  for (int w = 7; w >= 0; --w) {
    uint32_t o = 0;
    for (uint32_t i = l; i < K; i += L, ++o) col[256 * o] = digits[9 * (base + i) + w];
#pragma unroll 1
    for (int k = 7; k >= 0; --k) {
      fetch(l, 0, k);
      if (started) ge_double4(acc);
      o = 0;
#pragma unroll 1
      for (uint32_t i = l; i < K; i += L, ++o) {
        ge_cached q = qn;
        const uint32_t mag = magn, neg = negn;
        if (i + L < K) fetch(i + L, o + 1, k);
        if (mag) {
          ge_cached_cneg(q, neg);
          ge_add_cached(acc, acc, q);
          started = true;
        }
      }
    }
  }
  store_ext(spart + g, acc);
}
// The walk split by WINDOWS instead of by operands: part p of P adds, for ALL K operands of the proof, the digits of the 64 / P radix-16
// windows [p 64/P, (p+1) 64/P) -- 4 (64/P - 1) doublings instead of 252 per part, so splitting a proof over many lanes no longer
// multiplies the doublings -- and one quad per proof then joins the parts by Horner, sum_p 16^(p 64/P) S_p: the 252 doublings that no
// schedule can avoid, once per proof, on four lanes (k_straus_combine_quad).  digits[N K][9]: the recoded coefficient (8 words of
// signed radix-16 nibbles), word 8 = sign fold | carry out of bit 255 << 1 (k_straus_recode).  LDS: K words per lane.
__global__ void __launch_bounds__(256)
k_straus_recode(uint32_t NK, const uint8_t* __restrict__ scalars, uint32_t* __restrict__ digits) {
  const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= NK) return;
  uint32_t sw[8], e[8], top;
  load_vec<2>(sw, scalars + 32 * (size_t)g);
  const uint32_t flip = sc_fold_sign(sw);
  sc_add_pattern(e, top, sw, 0x88888888u);
  uint32_t* d = digits + 9 * (size_t)g;
#pragma unroll
  for (int w = 0; w < 8; ++w) d[w] = e[w];
  d[8] = flip | (top << 1);
}
__global__ void __launch_bounds__(256, 2)
k_straus_each_win(uint32_t N, uint32_t K, uint32_t P, uint32_t ns, uint32_t ni, uint32_t nc, const straus_entry* __restrict__ tab,
                  const uint32_t* __restrict__ digits, dev_ext* __restrict__ spart) {
  extern __shared__ uint32_t straus_lds[];
  uint32_t* col = straus_lds + threadIdx.x;                            // the current digit word of operand i at col[256 i]
  const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= N * P) return;
  const uint32_t j = g / P, p = g - j * P, np = ns + ni, Wp = 64u / P;
  const uint32_t* dj = digits + 9 * (size_t)j * K;
  auto table_of = [&](uint32_t i) -> const straus_entry* {
    const size_t t = i < ns ? i : (i < np ? (size_t)ns + (size_t)(i - ns) * N + j : (size_t)ns + (size_t)ni * N + (size_t)j * nc + (i - np));
    return tab + t * 8;
  };
  ge_p3 acc;
  ge_identity(acc);
  bool started = false;
  uint64_t flipmask = 0;                                               // (K <= 64: straus_win_parts)
  for (uint32_t i = 0; i < K; ++i) {
    const uint32_t f = dj[9 * i + 8];
    flipmask |= (uint64_t)(f & 1u) << i;
    if ((f >> 1) && p == P - 1) {                                      // carry out of bit 255 (non-canonical inputs only): one more P above the top window
      ge_cached q;
      load_straus_entry(q, table_of(i));
      ge_cached_cneg(q, f & 1u);
      ge_add_cached(acc, acc, q);
      started = true;
    }
  }
  ge_cached qn;
  uint32_t magn = 0, negn = 0;
  auto fetch = [&](uint32_t i, int k) {
    const uint32_t nib = (col[256 * i] >> (4 * k)) & 15u;
    const uint32_t neg = (uint32_t)(nib < 8u);
    magn = neg ? 8u - nib : nib - 8u;
    negn = neg ^ (uint32_t)((flipmask >> i) & 1u);
    if (magn) load_straus_entry(qn, table_of(i) + (magn - 1));
  };
  const int w_hi = (int)(Wp * (p + 1)) - 1, w_lo = (int)(Wp * p);
#pragma unroll 1
  for (int w = w_hi; w >= w_lo; --w) {
    const int k = w & 7;
    if (k == 7 || w == w_hi)
      for (uint32_t i = 0; i < K; ++i) col[256 * i] = dj[9 * i + (w >> 3)];
    fetch(0, k);
    if (started) ge_double4(acc);
#pragma unroll 1
    for (uint32_t i = 0; i < K; ++i) {
      ge_cached q = qn;
      const uint32_t mag = magn, neg = negn;
      if (i + 1 < K) fetch(i + 1, k);
      if (mag) {
        ge_cached_cneg(q, neg);
        ge_add_cached(acc, acc, q);
        started = true;
      }
    }
  }
  store_ext(spart + g, acc);
}
// quad per proof: out[j] = sum_p 16^(p Wp) spart[j P + p]
__global__ void __launch_bounds__(256)
k_straus_combine_quad(uint32_t N, uint32_t P, const dev_ext* __restrict__ spart, dev_ext* __restrict__ out) {
  const uint32_t gt = blockIdx.x * blockDim.x + threadIdx.x, j = gt >> 2;
  const int q = (int)(gt & 3u);
  if (j >= N) return;
  const dev_ext* s = spart + (size_t)j * P;
  const uint32_t dbl = 4u * (64u / P);
  qpt acc;
  q_load_ext(acc, s + (P - 1), q);
#pragma unroll 1
  for (int p = (int)P - 2; p >= 0; --p) {
#pragma unroll 1
    for (uint32_t d = 0; d < dbl; ++d) q_double(acc, acc, q);
    qpt t;
    q_load_ext(t, s + p, q);
    q_add(acc, acc, t, q);
  }
  q_store_ext(out + j, acc, q);
}
// lane per proof: sum of its L parts, decode status of its K operands, is the sum the identity (verifier.rs:162-168)
__global__ void __launch_bounds__(256, 2)
k_straus_finish(uint32_t N, uint32_t K, uint32_t L, const uint32_t* __restrict__ pidx, const dev_affine* __restrict__ pts, const dev_ext* __restrict__ spart,
                uint8_t* __restrict__ out, uint8_t* __restrict__ status8) {
  const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= N) return;
  ge_p3 acc;
  load_ext(acc, spart + (size_t)j * L);
  for (uint32_t l = 1; l < L; ++l) {
    ge_p3 q;
    load_ext(q, spart + (size_t)j * L + l);
    ge_add_p3(acc, acc, q);
  }
  uint32_t bad = 0;
  for (uint32_t i = 0; i < K; ++i) bad |= pts[pidx[(size_t)j * K + i]].valid ^ 1u;
  // the verdict needs is_identity() only (verifier.rs:166-168): a point is in the identity's coset iff X = 0 or Y = 0, which is when its
  // canonical encoding is 32 zero bytes -- no inversion.  out = those zero bytes, or a non-zero marker (k_each_finish tests for zero).
  uint32_t w[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) w[k] = 0;
  if (!bad) w[0] = (fe_iszero(acc.X) | fe_iszero(acc.Y)) ^ 1u;
  store_vec<2>(out + 32 * (size_t)j, w);
  status8[j] = (uint8_t)bad;
}

// Batch verification, after the transcripts, in one launch: any rejected proof of batch b -> any[b * any_stride] |= any_bit; -c mod l
// from the 64 challenge bytes; commitment rows of the operand list (rows[k][j] = commitments[j][k], batch_verifier.rs:208-212).
// A response that is not a canonical scalar rejects its batch too: the reference can only receive responses through serde,
// and dalek's Deserialize refuses s >= l (proofs.rs:14-32; SURVEY 8(a) semantic 5).  N = all proofs of the call, batches of N_each.
__global__ void __launch_bounds__(256)
k_batch_after_transcript(uint32_t N, uint32_t N_each, uint32_t nc, uint32_t m, const uint32_t* __restrict__ failed, uint32_t* __restrict__ any, uint32_t any_stride,
                         uint32_t any_bit, const uint8_t* __restrict__ wchal, uint8_t* __restrict__ minus_c, const uint8_t* __restrict__ coms,
                         uint8_t* __restrict__ rows, const uint8_t* __restrict__ responses) {
  const size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (g < N) {
    if (failed[g]) atomicOr(any + (g / N_each) * any_stride, any_bit);
    sc lo, hi, r;
    load_vec<2>(lo.v, wchal + 64 * g);
    load_vec<2>(hi.v, wchal + 64 * g + 32);
    sc_from_wide(r, lo, hi);
    sc_neg(r, r);
    store_vec<2>(minus_c + 32 * g, r.v);
  }
  if (coms && g < (size_t)N * nc) {                       // (coms == NULL: the latency schedule has transposed them before the transcripts, k_transpose_commitments)
    const size_t j = g / nc, k = g % nc;
    uint32_t w[8];
    load_vec<2>(w, coms + 32 * g);
    store_vec<2>(rows + 32 * (k * N + j), w);
  }
  if (g < (size_t)N * m) {
    uint32_t w[8];
    load_vec<2>(w, responses + 32 * g);
    if (sc_not_canonical(w)) atomicOr(any + ((g / m) / N_each) * any_stride, any_bit);
  }
}

__global__ void k_any_nonzero(uint32_t n, const uint32_t* __restrict__ flags, uint32_t* __restrict__ any) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n && flags[i]) *any = 1;
}

}  // namespace zkp

namespace {
using namespace zkp;

struct fused_shape {
  uint32_t m, ns, ni, nc, np, T;
  std::vector<uint32_t> inc_off, inc_k, inc_sc;          // per point id: (constraint, secret | ~0)
  std::vector<uint32_t> unref;                           // point ids no constraint mentions
};

int parse_shape(const zkp_batch_statement* st, fused_shape& s) {
  s.m = st->n_secrets; s.ns = st->n_static; s.ni = st->n_instance; s.nc = st->n_constraints; s.np = s.ns + s.ni;
  if (s.nc && (!st->cons_lhs || !st->cons_off)) return fail(ZKP_ERR_ARG, "statement without constraint arrays");
  s.T = s.nc ? st->cons_off[s.nc] : 0;
  if (s.T && (!st->cons_sc || !st->cons_pt)) return fail(ZKP_ERR_ARG, "statement without term arrays");
  std::vector<std::vector<std::pair<uint32_t, uint32_t>>> inc(s.np);
  std::vector<char> used(s.np, 0);
  for (uint32_t k = 0; k < s.nc; ++k) {
    if (st->cons_lhs[k] >= s.np) return fail(ZKP_ERR_ARG, "constraint lhs out of range");
    if (st->cons_off[k + 1] < st->cons_off[k]) return fail(ZKP_ERR_ARG, "cons_off must be non-decreasing");
    inc[st->cons_lhs[k]].emplace_back(k, 0xffffffffu);
    used[st->cons_lhs[k]] = 1;
    for (uint32_t q = st->cons_off[k]; q < st->cons_off[k + 1]; ++q) {
      if (st->cons_pt[q] >= s.np || st->cons_sc[q] >= s.m) return fail(ZKP_ERR_ARG, "constraint term out of range");
      inc[st->cons_pt[q]].emplace_back(k, st->cons_sc[q]);
      used[st->cons_pt[q]] = 1;
    }
  }
  s.inc_off.assign(s.np + 1, 0);
  for (uint32_t p = 0; p < s.np; ++p) {
    for (auto& e : inc[p]) { s.inc_k.push_back(e.first); s.inc_sc.push_back(e.second); }
    s.inc_off[p + 1] = (uint32_t)s.inc_k.size();
    if (!used[p]) s.unref.push_back(p);
  }
  return ZKP_OK;
}
std::vector<uint32_t> incidence_words(const fused_shape& s) {
  std::vector<uint32_t> w(s.inc_off);
  w.insert(w.end(), s.inc_k.begin(), s.inc_k.end());
  w.insert(w.end(), s.inc_sc.begin(), s.inc_sc.end());
  return w;
}

// batch_verifier.rs:173-206 on device buffers; d_inc = inc_off | inc_k | inc_sc.  K batches of N_each proofs (K = 1: one batch of N):
// d_sc = static coefficients [K][ns] || Matrix rows [ni + nc][K * N_each];  d_part holds ns * K * ceil(N_each / 256) partial sums
void launch_coeff_build(zkp_ctx* c, const fused_shape& s, uint32_t N_each, const uint32_t* d_inc, const uint8_t* d_mc,
                        const uint8_t* d_resp, const uint8_t* d_w, uint8_t* d_sc, uint32_t* d_part, uint32_t K = 1) {
  const uint32_t* d_inc_off = d_inc;
  const uint32_t* d_inc_k = d_inc + s.inc_off.size();
  const uint32_t* d_inc_sc = d_inc_k + s.inc_k.size();
  const uint32_t nblk = (N_each + 255) / 256, rows = s.ni + s.nc, N = K * N_each;
  if (N && (rows || s.ns)) {
    hipLaunchKernelGGL(k_coeff_build, dim3(K * nblk, rows + s.ns), dim3(256), 0, c->stream, N, N_each, nblk, K, s.m, s.ns, s.ni, s.nc, d_inc_off, d_inc_k, d_inc_sc, d_mc, d_resp, d_w,
                       d_sc, d_part);
    if (s.ns) hipLaunchKernelGGL(k_coeff_static_final, dim3(s.ns, K), dim3(64), 0, c->stream, nblk, s.ns, d_part, d_sc);
  } else if (s.ns) {
    hipMemsetAsync(d_sc, 0, (size_t)K * s.ns * 32, c->stream);
  }
}
size_t optional_ws(uint64_t total) {
  if (total <= kSmallOptional) return 1024 + (total + 1) * 4 + terms_path_ws((uint32_t)total, (uint32_t)total, 1);
  switch (pick_c(total)) {
    case 7: return pip_ws<7>(total);
    case 10: return pip_ws<10>(total);
    case 11: return pip_ws<11>(total);
    default: return pip_ws<16>(total);
  }
}

// ---- plans: everything about (flow, statement, N, STROBE position) that does not depend on the data -----------------
enum { SRC_TABLE = 0, SRC_SECRETS = 1, SRC_ENTROPY = 2, SRC_COMS = 3 };
enum { DST_WIDE = 0, DST_CHAL = 1 };
enum fused_flow : char { FLOW_PROVE = 'P', FLOW_VERIFY = 'V', FLOW_BATCH = 'B' };

// sd / steps: the same program in step form (assemble + chain, transcript_kernels.h); steps = false when the plan could not build it (a step with more
// than 64 PRF-output operations, more image words than a grid has rows) or the call is wide enough for the one-lane interpreter
struct prog_dev { const tr_op* ops = nullptr; uint32_t n = 0; uint32_t tail = 0; const uint64_t* tables = nullptr; tr_steps_dev sd; bool steps = false; };
struct fused_plan {
  fused_shape s;
  uint32_t N = 0, T1 = 0;          // T1 = terms per proof of the flow's CSR job
  char* d_block = nullptr;         // one allocation: programs | term arrays | incidence
  prog_dev a, b;
  const uint32_t* d_tarr = nullptr;   // toff[nc + 1] | tsc[T1] | tpt[T1] | unref[] | order[nc]
  const uint32_t* d_order = nullptr;  // constraints by descending number of terms (msm_map of the reduce / encode kernels)
  const uint32_t* d_inc = nullptr;
  std::vector<uint32_t> tpt;       // host copy of tpt[]: comb-table shape and table / ladder bounds of the flow's CSR job
  std::vector<uint32_t> pair;      // FLOW_VERIFY: stmt_job::pair (empty: nothing pairs), d_pair = its device copy
  const uint32_t* d_pair = nullptr;
  size_t img_bytes = 0;            // workspace of the step programs' images: max over the plan's programs of n_img * 21 words per proof (0 = no step form)
};

// Table / ladder bounds and comb shape of a CSR job whose proofs all multiply the point ids tpt[] (ids < ns: common to the
// batch; the others: one point per proof).  A common point that is registered for a fixed-base table leaves the cold
// classes at run time, which only lowers the counts.
terms_cfg cfg_from_terms(const uint32_t* tpt, uint32_t T1, uint32_t ns, uint32_t np, uint32_t N, uint32_t comb_min, const uint32_t* pair = nullptr, bool riders = false) {
  std::vector<uint64_t> u(np, 0), a(np, 0);
  for (uint32_t i = 0; i < T1; ++i) ++(stmt_absorbed(pair, i) ? a : u)[tpt[i]];    // (a term that rides on another's doubling chain is no use of its point)
  if (comb_min == 1) {
    // constant-time calls give single-use points a table only so that their 256 doublings run next to the table chains of
    // the shared points instead of inside the term kernel; a statement without shared points (DLEQ: B = x * H) has no
    // such chains, and the ladder (7 + 256 + 65 operations) is then cheaper than table + walk (256 + 7 TEETH + BITS - 4 + 65)
    bool shared = false;
    for (uint32_t p = ns; p < np; ++p) shared |= u[p] >= 2;      // (common points normally sit on fixed-base tables)
    if (!shared) comb_min = 2;
  }
  uint64_t n_tab = 0, n_lad = 0, tab_terms = 0;
  for (uint32_t p = 0; p < np; ++p) {
    const uint64_t mult = p < ns ? 1 : N, uses = p < ns ? u[p] * N : u[p];
    if (uses >= comb_min && uses) { n_tab += mult; tab_terms += uses * mult; } else if (uses == 1) n_lad += mult;
    // a table of multiples for a per-proof point all of whose terms ride (stmt_rider): counted like a comb table, its riders like table terms
    if (riders && stmt_rider(1u, p, ns, (uint32_t)std::min<uint64_t>(u[p], 0xffffffffu), (uint32_t)std::min<uint64_t>(a[p], 0xffffffffu))) { n_tab += mult; tab_terms += a[p] * mult; }
  }
  terms_cfg k;
  k.comb_min = comb_min;
  k.max_tables = (uint32_t)std::min<uint64_t>(n_tab, 0xffffffffu);
  k.max_ladder = (uint32_t)std::min<uint64_t>(n_lad, 0xffffffffu);
  k.teeth = pick_teeth(n_tab, tab_terms);
  return k;
}

int check_fused_statement(const zkp_fused_statement* st, fused_shape& s) {
  if (!st) return fail(ZKP_ERR_ARG, "statement is NULL");
  int rc = parse_shape(&st->shape, s);
  if (rc) return rc;
  if (!st->label || (s.m && !st->secret_labels) || (s.np && (!st->point_labels || !st->alloc_order)))
    return fail(ZKP_ERR_ARG, "statement labels missing");
  std::vector<char> seen(s.np, 0);
  for (uint32_t i = 0; i < s.np; ++i) {
    if (st->alloc_order[i] >= s.np || seen[st->alloc_order[i]]) return fail(ZKP_ERR_ARG, "alloc_order is not a permutation of the point ids");
    seen[st->alloc_order[i]] = 1;
  }
  if (st->alloc_seq) {                  // every secret once, the points in alloc_order's order
    std::vector<char> sec_seen(s.m, 0);
    uint32_t next_pt = 0;
    for (uint32_t a = 0; a < s.m + s.np; ++a) {
      const uint32_t e = st->alloc_seq[a];
      if (e & 0x80000000u) {
        const uint32_t i = e & 0x7fffffffu;
        if (i >= s.m || sec_seen[i]) return fail(ZKP_ERR_ARG, "alloc_seq: secret index out of range or repeated");
        sec_seen[i] = 1;
      } else {
        if (next_pt >= s.np || st->alloc_order[next_pt] != e) return fail(ZKP_ERR_ARG, "alloc_seq: points must follow alloc_order");
        ++next_pt;
      }
    }
  }
  return ZKP_OK;
}
// all transcripts of a fused call must stand at the same STROBE position (they do whenever they were built by the
// same sequence of appends with equal lengths); the caller falls back to the host pipeline otherwise
int common_tail(const uint8_t* ts, uint32_t N, uint32_t* pos) {
  for (uint32_t j = 1; j < N; ++j)
    if (memcmp(ts + 208 * (size_t)j + 200, ts + 200, 3) != 0) return fail(ZKP_ERR_ARG, "transcripts stand at different STROBE positions");
  *pos = ts[200] | (uint32_t)ts[201] << 8 | (uint32_t)ts[202] << 16;
  return ZKP_OK;
}

// Prover::new / Verifier::new + allocate_scalar + allocate_point in allocation order (prover.rs:41-73, verifier.rs:47-77).
// Every proof reads the table  common || inst[ni][N]; common points are variables of the program too (stride 0), so a
// compiled program does not depend on any value.
void compile_allocations(TrCompiler& tc, const zkp_fused_statement* st, const fused_shape& s, uint32_t N, bool validate) {
  tc.domain_sep(st->label);
  auto point = [&](uint32_t p) {
    tr_ref ref;
    if (p < s.ns) ref = tr_ref{SRC_TABLE, 0, 32ull * p};
    else ref = tr_ref{SRC_TABLE, 32, 32ull * (s.ns + (uint64_t)(p - s.ns) * N)};
    tc.append_point_var_var(st->point_labels[p], ref, validate);
  };
  if (st->alloc_seq) {                                   // the caller's interleaving of scalar and point allocations
    for (uint32_t a = 0; a < s.m + s.np; ++a) {
      const uint32_t e = st->alloc_seq[a];
      if (e & 0x80000000u) tc.append_scalar_var(st->secret_labels[e & 0x7fffffffu]);
      else point(e);
    }
    return;
  }
  for (uint32_t i = 0; i < s.m; ++i) tc.append_scalar_var(st->secret_labels[i]);
  for (uint32_t a = 0; a < s.np; ++a) point(st->alloc_order[a]);
}

std::string plan_key(char flow, const zkp_fused_statement* st, const fused_shape& s, uint32_t N, uint32_t pos) {
  std::string k(1, flow);
  auto u32 = [&](uint32_t v) { k.append(reinterpret_cast<const char*>(&v), 4); };
  auto str = [&](const char* p) { u32((uint32_t)strlen(p)); k.append(p); };
  u32(N); u32(pos); u32(s.m); u32(s.ns); u32(s.ni); u32(s.nc);
  str(st->label);
  for (uint32_t i = 0; i < s.m; ++i) str(st->secret_labels[i]);
  for (uint32_t i = 0; i < s.np; ++i) { str(st->point_labels[i]); u32(st->alloc_order[i]); }
  for (uint32_t i = 0; i < s.nc; ++i) { u32(st->shape.cons_lhs[i]); u32(st->shape.cons_off[i + 1]); }
  for (uint32_t i = 0; i < s.T; ++i) { u32(st->shape.cons_sc[i]); u32(st->shape.cons_pt[i]); }
  if (st->alloc_seq) for (uint32_t a = 0; a < s.m + s.np; ++a) u32(st->alloc_seq[a]);
  return k;
}

// ZKP_DEBUG_TRANSCRIPT=1: the compiled transcript program of a (flow, statement) is listed on stderr when its plan is
// built -- one line per operation, decoded: what each proof absorbs / keys from which input buffer at which stride and offset,
// what it emits, where it clones / restores the state, and the Keccak permutations in between.  Together with the host
// library's op log (host/merlin.cpp, same variable) this is what the reference's `debug-transcript` feature (Cargo.toml:35)
// gives: a challenge mismatch is found by diffing op sequences, not by staring at 32 wrong bytes.
bool debug_transcript_enabled() {
  static const bool on = [] { const char* e = getenv("ZKP_DEBUG_TRANSCRIPT"); return e && *e && *e != '0'; }();
  return on;
}
void dump_program(char flow, const char* which, const std::vector<tr_op>& ops, uint32_t N) {
  static const char* src_names[] = {"-", "points", "secrets", "entropy", "commitments"};
  static const char* dst_names[] = {"-", "wide(blindings)", "wide(challenge)"};
  fprintf(stderr, "[transcript program] flow=%c %s: %zu operations per proof (N = %u)\n", flow, which, ops.size(), N);
  uint32_t perms = 0;
  for (size_t i = 0; i < ops.size(); ++i) {
    const tr_fields f = tr_unpack(ops[i].ctl, ops[i].stride, ops[i].off);
    fprintf(stderr, "  %4zu:", i);
    if (f.flags & TR_RESTORE) fprintf(stderr, " RESTORE(state <- clone)");
    if (f.flags & TR_CHECK_NONZERO) fprintf(stderr, " REJECT-IDENTITY %s[j * %u + %llu .. +32]", src_names[f.src_buf < 5 ? f.src_buf : 0], f.stride, (unsigned long long)f.off);
    if (f.dst_buf) fprintf(stderr, " EMIT state.word[%u] bytes[%u..%u) -> %s[j * %u + %llu]", f.w, f.dlb, f.dlb + f.dnb, dst_names[f.dst_buf < 3 ? f.dst_buf : 0], f.stride, (unsigned long long)f.off);
    if (f.src_buf && !(f.flags & TR_CHECK_NONZERO))
      fprintf(stderr, " %s state.word[%u] bytes[%u..%u) <- %s[j * %u + %llu]", (f.flags & TR_OVERWRITE) ? "KEY" : "ABSORB", f.w, f.lb, f.lb + f.nb,
              src_names[f.src_buf < 5 ? f.src_buf : 0], f.stride, (unsigned long long)f.off);
    if (f.flags & TR_APPLY) fprintf(stderr, " APPLY constants[%llu] (labels, lengths, STROBE framing)%s", (unsigned long long)f.off, (f.flags & TR_PERMUTE) ? " + KECCAK-F" : "");
    if (f.flags & TR_SAVE) fprintf(stderr, " SAVE(clone <- state)");
    if (f.flags & TR_PERMUTE) ++perms;
    fputc('\n', stderr);
  }
  fprintf(stderr, "[transcript program] %u Keccak-f permutations per proof\n", perms);
}

int get_plan(zkp_ctx* c, char flow, const zkp_fused_statement* st, uint32_t N, uint32_t pos, fused_plan** out) {
  fused_shape s;
  int rc = check_fused_statement(st, s);
  if (rc) return rc;
  if ((pos & 0xff) >= 166) return fail(ZKP_ERR_ARG, "corrupt transcript blob (STROBE position out of range)");
  const std::string key = plan_key(flow, st, s, N, pos);
  auto it = c->fused_plans.find(key);
  if (it != c->fused_plans.end()) { *out = static_cast<fused_plan*>(it->second); return ZKP_OK; }
  if (c->capturing) return fail(ZKP_ERR_ARG, "graph capture: this statement has no compiled plan yet -- run the same call once before capturing it");
  std::unique_ptr<fused_plan> pl(new fused_plan());
  pl->s = s;
  pl->N = N;
  const uint32_t m = s.m, nc = s.nc;
  uint8_t tailA[3], tailB[3];
  std::vector<tr_op> pa, pb;
  std::vector<uint64_t> tbl_a, tbl_b;
  std::vector<uint32_t> tarr;
  TrCompiler ta((uint8_t)pos, (uint8_t)(pos >> 8), (uint8_t)(pos >> 16));
  if (flow == FLOW_PROVE) {
    // program A: allocations, then the blinding factors from a clone of the transcript (prover.rs:78-89)
    compile_allocations(ta, st, s, N, false);
    ta.save();
    for (uint32_t i = 0; i < m; ++i) ta.rng_rekey_with_witness_var("", tr_ref{SRC_SECRETS, 32 * m, 32ull * i}, 32);
    ta.rng_finalize_var(tr_ref{SRC_ENTROPY, 32, 0});
    for (uint32_t i = 0; i < m; ++i) ta.rng_fill_bytes(tr_ref{DST_WIDE, 64 * m, 64ull * i}, 64);
    ta.restore();
    pa = ta.finish(tailA);
    tbl_a = ta.tables();
    // program B: commitments, challenge (prover.rs:98-106)
    TrCompiler tb(tailA[0], tailA[1], tailA[2]);
    for (uint32_t k = 0; k < nc; ++k)
      tb.append_blinding_commitment_var(st->point_labels[st->shape.cons_lhs[k]], tr_ref{SRC_COMS, 32 * nc, 32ull * k}, false);
    tb.get_challenge_wide("chal", tr_ref{DST_CHAL, 64, 0});
    pb = tb.finish(tailB);
    tbl_b = tb.tables();
    // prover.rs:94-97 operand lists
    tarr.assign(nc + 1, 0);
    for (uint32_t k = 0; k < nc; ++k) tarr[k + 1] = st->shape.cons_off[k + 1];
    if (s.T) { tarr.insert(tarr.end(), st->shape.cons_sc, st->shape.cons_sc + s.T); tarr.insert(tarr.end(), st->shape.cons_pt, st->shape.cons_pt + s.T); }
    pl->T1 = s.T;
    pl->tpt.assign(st->shape.cons_pt, st->shape.cons_pt + s.T);
  } else if (flow == FLOW_VERIFY) {
    compile_allocations(ta, st, s, N, true);                            // verifier.rs:61-77 validating appends
    pa = ta.finish(tailA);
    tbl_a = ta.tables();
    TrCompiler tb(tailA[0], tailA[1], tailA[2]);
    for (uint32_t k = 0; k < nc; ++k)                                    // verifier.rs:108 (non-validating)
      tb.append_blinding_commitment_var(st->point_labels[st->shape.cons_lhs[k]], tr_ref{SRC_COMS, 32 * nc, 32ull * k}, false);
    tb.get_challenge_wide("chal", tr_ref{DST_CHAL, 64, 0});
    pb = tb.finish(tailB);
    tbl_b = tb.tables();
    // verifier.rs:95-106: per constraint the rhs terms with the responses, then (-c) on the lhs point
    std::vector<uint32_t> vsc, vpt;
    tarr.assign(nc + 1, 0);
    for (uint32_t k = 0; k < nc; ++k) {
      for (uint32_t q = st->shape.cons_off[k]; q < st->shape.cons_off[k + 1]; ++q) { vsc.push_back(st->shape.cons_sc[q]); vpt.push_back(st->shape.cons_pt[q]); }
      vsc.push_back(0xffffffffu);
      vpt.push_back(st->shape.cons_lhs[k]);
      tarr[k + 1] = (uint32_t)vsc.size();
    }
    pl->T1 = (uint32_t)vsc.size();
    pl->tpt = vpt;
    tarr.insert(tarr.end(), vsc.begin(), vsc.end());
    tarr.insert(tarr.end(), vpt.begin(), vpt.end());
    tarr.insert(tarr.end(), s.unref.begin(), s.unref.end());
    pl->pair = pair_terms(tarr.data(), vpt.data(), pl->T1, nc, s.ns, s.np);
  } else {
    compile_allocations(ta, st, s, N, true);                            // batch_verifier.rs:92-94, :105-107, :125-128
    for (uint32_t k = 0; k < nc; ++k)                                    // :152-160 validating
      ta.append_blinding_commitment_var(st->point_labels[st->shape.cons_lhs[k]], tr_ref{SRC_COMS, 32 * nc, 32ull * k}, true);
    ta.get_challenge_wide("chal", tr_ref{DST_CHAL, 64, 0});              // :163-167
    pa = ta.finish(tailA);
    tbl_a = ta.tables();
  }
  size_t order_at = 0, pair_at = 0;
  if (!pl->pair.empty()) { pair_at = tarr.size(); tarr.insert(tarr.end(), pl->pair.begin(), pl->pair.end()); }
  if (flow != FLOW_BATCH && nc) {       // tarr[0 .. nc] = term offsets of the flow's MSMs
    std::vector<uint32_t> order(nc);
    for (uint32_t k = 0; k < nc; ++k) order[k] = k;
    std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return tarr[a + 1] - tarr[a] > tarr[b + 1] - tarr[b]; });
    order_at = tarr.size();
    tarr.insert(tarr.end(), order.begin(), order.end());
  }
  if (debug_transcript_enabled()) {
    dump_program(flow, flow == FLOW_BATCH ? "program (allocations, commitments, challenge)" : "program A (allocations ...)", pa, N);
    if (!pb.empty()) dump_program(flow, "program B (commitments, challenge)", pb, N);
  }
  const std::vector<uint32_t> inc = incidence_words(s);
  // the step form of both programs (round 6): not for calls the one-lane interpreter serves (>= kVeryWideCallProofs proofs: their images would be gigabytes)
  const tr_step_prog spa = tr_steps_build(pa, tbl_a), spb = tr_steps_build(pb, tbl_b);
  auto steps_ok = [&](const tr_step_prog& sp) {
    if (sp.steps.empty() || N >= zkp_ctx::kVeryWideCallProofs || (size_t)sp.n_img * 3 + 1 > 65535) return false;
    for (const tr_step& t : sp.steps) if (t.emit_n > (uint32_t)TR_BLOCK) return false;
    return true;
  };
  const bool ok_a = steps_ok(spa), ok_b = steps_ok(spb);
  carve cv;
  struct step_off { size_t steps, emit, src, soff, cx, keep, chk; } so_a{}, so_b{};
  auto carve_steps = [&](const tr_step_prog& sp, step_off& o) {
    o.steps = cv.take(sp.steps.size() * sizeof(tr_step) + 64);
    o.emit = cv.take(sp.emit.size() * sizeof(tr_op) + 64);
    o.src = cv.take(sp.src.size() * sizeof(tr_op) + 64);
    o.soff = cv.take(sp.src_off.size() * 4 + 64);
    o.cx = cv.take(sp.cx.size() * 8 + 64);
    o.keep = cv.take(sp.keep.size() * 8 + 64);
    o.chk = cv.take(sp.chk.size() * sizeof(tr_op) + 64);
  };
  if (ok_a) carve_steps(spa, so_a);
  if (ok_b) carve_steps(spb, so_b);
  const size_t o_a = cv.take(pa.size() * sizeof(tr_op) + 64);
  const size_t o_b = cv.take(pb.size() * sizeof(tr_op) + 64);
  const size_t o_ta = cv.take(tbl_a.size() * 8 + 64);
  const size_t o_tb = cv.take(tbl_b.size() * 8 + 64);
  const size_t o_t = cv.take(tarr.size() * 4 + 64);
  const size_t o_i = cv.take(inc.size() * 4 + 64);
  HIP_TRY(hipSetDevice(c->device));
  HIP_TRY(hipMalloc(reinterpret_cast<void**>(&pl->d_block), cv.off));
  auto put = [&](size_t o, const void* src, size_t bytes) { return bytes ? hipMemcpy(pl->d_block + o, src, bytes, hipMemcpyHostToDevice) : hipSuccess; };
  hipError_t e = put(o_a, pa.data(), pa.size() * sizeof(tr_op));
  if (e == hipSuccess) e = put(o_b, pb.data(), pb.size() * sizeof(tr_op));
  if (e == hipSuccess) e = put(o_ta, tbl_a.data(), tbl_a.size() * 8);
  if (e == hipSuccess) e = put(o_tb, tbl_b.data(), tbl_b.size() * 8);
  if (e == hipSuccess) e = put(o_t, tarr.data(), tarr.size() * 4);
  if (e == hipSuccess) e = put(o_i, inc.data(), inc.size() * 4);
  auto put_steps = [&](const tr_step_prog& sp, const step_off& o, tr_steps_dev& d) {
    if (e == hipSuccess) e = put(o.steps, sp.steps.data(), sp.steps.size() * sizeof(tr_step));
    if (e == hipSuccess) e = put(o.emit, sp.emit.data(), sp.emit.size() * sizeof(tr_op));
    if (e == hipSuccess) e = put(o.src, sp.src.data(), sp.src.size() * sizeof(tr_op));
    if (e == hipSuccess) e = put(o.soff, sp.src_off.data(), sp.src_off.size() * 4);
    if (e == hipSuccess) e = put(o.cx, sp.cx.data(), sp.cx.size() * 8);
    if (e == hipSuccess) e = put(o.keep, sp.keep.data(), sp.keep.size() * 8);
    if (e == hipSuccess) e = put(o.chk, sp.chk.data(), sp.chk.size() * sizeof(tr_op));
    d.steps = reinterpret_cast<const tr_step*>(pl->d_block + o.steps);
    d.emit = reinterpret_cast<const tr_op*>(pl->d_block + o.emit);
    d.src = reinterpret_cast<const tr_op*>(pl->d_block + o.src);
    d.src_off = reinterpret_cast<const uint32_t*>(pl->d_block + o.soff);
    d.cx = reinterpret_cast<const uint64_t*>(pl->d_block + o.cx);
    d.keep32 = reinterpret_cast<const uint32_t*>(pl->d_block + o.keep);
    d.chk = reinterpret_cast<const tr_op*>(pl->d_block + o.chk);
    d.n_steps = (uint32_t)sp.steps.size(); d.n_img = sp.n_img; d.n_chk = (uint32_t)sp.chk.size();
  };
  tr_steps_dev sda, sdb;
  if (ok_a) put_steps(spa, so_a, sda);
  if (ok_b) put_steps(spb, so_b, sdb);
  if (e != hipSuccess) { hipFree(pl->d_block); return fail(ZKP_ERR_HIP, std::string("plan upload: ") + hipGetErrorString(e)); }
  pl->a = prog_dev{reinterpret_cast<const tr_op*>(pl->d_block + o_a), (uint32_t)pa.size(), tailA[0] | (uint32_t)tailA[1] << 8 | (uint32_t)tailA[2] << 16,
                    reinterpret_cast<const uint64_t*>(pl->d_block + o_ta)};
  pl->b = prog_dev{reinterpret_cast<const tr_op*>(pl->d_block + o_b), (uint32_t)pb.size(), tailB[0] | (uint32_t)tailB[1] << 8 | (uint32_t)tailB[2] << 16,
                    reinterpret_cast<const uint64_t*>(pl->d_block + o_tb)};
  pl->a.sd = sda; pl->a.steps = ok_a;
  pl->b.sd = sdb; pl->b.steps = ok_b;
  pl->img_bytes = (size_t)std::max(ok_a ? spa.n_img : 0u, ok_b ? spb.n_img : 0u) * 21 * 8 * N;
  pl->d_tarr = reinterpret_cast<const uint32_t*>(pl->d_block + o_t);
  pl->d_order = order_at ? pl->d_tarr + order_at : nullptr;
  pl->d_pair = pair_at ? pl->d_tarr + pair_at : nullptr;
  pl->d_inc = reinterpret_cast<const uint32_t*>(pl->d_block + o_i);
  if (c->fused_plans.size() >= 64) free_fused_plans(c);        // a bound on what a long-lived context keeps
  *out = pl.get();
  c->fused_plans[key] = pl.release();
  return ZKP_OK;
}

// throughput = the caller keeps calls in flight (_dev entry points): one lane per proof when the call is wide; otherwise a lane
// pair per proof
#ifdef ZKP_BUILD_TEST_HOOKS
__global__ void k_noop(uint32_t* p) { if (p) *p = 0; }
#endif
bool transcript_single_lane(const zkp_ctx* c, uint32_t N, bool throughput) {
  return c->tr_lanes < 0 ? (throughput && N >= zkp_ctx::kVeryWideCallProofs) : c->tr_lanes == 1;
}
// the step form (assemble + chain) serves every call the lane-pair interpreter would: ZKP_OPT_TRANSCRIPT_STEPS = 0 goes back to the interpreter
bool transcript_steps(const zkp_ctx* c, const prog_dev& p, uint32_t N, bool throughput, const uint64_t* d_img) {
  return c->tr_steps && p.steps && d_img && !transcript_single_lane(c, N, throughput);
}
// the wide half of a step program: image words + identity checks (nothing here depends on a transcript state)
void launch_assemble(zkp_ctx* c, const tr_steps_dev& sd, uint32_t N, const tr_bufs& bufs, uint64_t* d_img, uint32_t* d_failed) {
  const uint32_t rows = sd.n_img * 3u + ((sd.n_chk || (sd.tail >> 31)) ? 1u : 0u);
  if (rows) hipLaunchKernelGGL(k_transcript_assemble, dim3((N + TA_BLOCK - 1) / TA_BLOCK, rows), dim3(TA_BLOCK), 0, c->stream, sd, N, bufs, d_img, d_failed);
}
// phase (step programs only): 1 = the assemble pass, 2 = the chain, 3 = both
void run_program(zkp_ctx* c, const prog_dev& p_in, uint32_t N, const tr_bufs& bufs, uint8_t* d_ts, uint64_t* d_saved, uint32_t* d_failed, bool throughput,
                 uint64_t* d_img, bool owns_failed = false, int phase = 3) {
  if (!p_in.n) return;
  prog_dev p = p_in;
  if (owns_failed) p.tail |= 0x80000000u;          // the kernel writes every proof's rejection flag, 0 included
  if (transcript_steps(c, p, N, throughput, d_img)) {
    prof_note(c, ZKP_K_TRANSCRIPT, "zkp::k_transcript_chain");
    tr_steps_dev sd = p.sd;
    sd.tail = p.tail;
    if (phase & 1) launch_assemble(c, sd, N, bufs, d_img, d_failed);
    constexpr uint32_t per_block = TR_BLOCK / 2;
    if (phase & 2) {
      if (throughput)
        hipLaunchKernelGGL(k_transcript_chain<false>, dim3((N + per_block - 1) / per_block), dim3(TR_BLOCK), 0, c->stream, sd, reinterpret_cast<const uint32_t*>(d_img), N, bufs, d_ts,
                           reinterpret_cast<uint32_t*>(d_saved));
      else
        hipLaunchKernelGGL(k_transcript_chain<true>, dim3((N + per_block - 1) / per_block), dim3(TR_BLOCK), 0, c->stream, sd, reinterpret_cast<const uint32_t*>(d_img), N, bufs, d_ts,
                           reinterpret_cast<uint32_t*>(d_saved));
    }
    return;
  }
  prof_note(c, ZKP_K_TRANSCRIPT, transcript_single_lane(c, N, throughput) ? "zkp::k_transcript_run1" : "zkp::k_transcript_run");
  if (transcript_single_lane(c, N, throughput)) {
    hipLaunchKernelGGL(k_transcript_run1, dim3((N + TR_BLOCK - 1) / TR_BLOCK), dim3(TR_BLOCK), 0, c->stream, p.ops, p.n, p.tables, N, bufs, d_ts, d_saved, d_failed, p.tail);
  } else {
    constexpr uint32_t per_block = TR_BLOCK / 2;
    hipLaunchKernelGGL(k_transcript_run, dim3((N + per_block - 1) / per_block), dim3(TR_BLOCK), 0, c->stream, p.ops, p.n, p.tables, N, bufs, d_ts,
                       reinterpret_cast<uint32_t*>(d_saved), d_failed, p.tail);
  }
}
// Offer the program to the point phase of the term path that follows (it runs with the comb-table construction if there is
// one: k_tables_transcript); run_program_pending() afterwards runs it on its own if nobody took it.
void offer_program(zkp_ctx* c, const prog_dev& p, uint32_t N, const tr_bufs& bufs, uint8_t* d_ts, uint64_t* d_saved, uint32_t* d_failed, bool throughput, bool overlap,
                   uint64_t* d_img) {
  auto& t = c->pending_tr;
  const bool fuse = c->fuse_tables_transcript < 0 ? N < zkp_ctx::kVeryWideCallProofs : c->fuse_tables_transcript != 0;
  t.offered = t.active = fuse && p.n != 0 && throughput && !overlap && !transcript_single_lane(c, N, throughput);
  t.ops = p.ops; t.n_ops = p.n; t.tables = p.tables; t.N = N; t.bufs = bufs; t.ts = d_ts; t.saved = reinterpret_cast<uint32_t*>(d_saved); t.failed = d_failed; t.tail = p.tail;
  t.steps = transcript_steps(c, p, N, throughput, d_img);
  t.sd = p.sd; t.sd.tail = p.tail; t.img = d_img;
}
void run_program_pending(zkp_ctx* c, const prog_dev& p, uint32_t N, const tr_bufs& bufs, uint8_t* d_ts, uint64_t* d_saved, uint32_t* d_failed, bool throughput,
                         uint64_t* d_img) {
  const bool taken = c->pending_tr.offered && !c->pending_tr.active;      // the term path launched it with its tables
  c->pending_tr.offered = c->pending_tr.active = false;
  if (!taken) run_program(c, p, N, bufs, d_ts, d_saved, d_failed, throughput, d_img);
}

// ---- side stream: the scalar-independent half of path A runs next to the transcripts -------------------------------------
// overlap = false: the same work stays on the context's stream.  The synchronous host-pointer entry points overlap (one
// call in flight per context: latency matters, prove 1.71 -> 1.45 ms per 4096 CMZ proofs); the asynchronous _dev entry
// points do not (their callers pipeline many contexts, and cross-stream events between 2 x 16 streams cost more
// throughput than the overlap returns: 0.99 -> 1.9 ms per step).
// A graph with a forked branch, replayed by a process that owns ONE hardware queue (GPU_MAX_HW_QUEUES=1: what a profiler run that wants every
// kernel of a trace serialised sets), crashes inside hipGraphLaunch on ROCm 7.2.0.  While a capture is recording under that setting the work stays
// on the context's stream -- one queue would run the two branches one after the other anyway.  (capturing is constant between a flow's
// side_begin and its side_join, so the three calls agree.)
inline bool side_forks(const zkp_ctx* c, bool overlap) {
  static const bool one_queue = [] { const char* e = getenv("GPU_MAX_HW_QUEUES"); return e && atoi(e) == 1; }();
  return overlap && !(c->capturing && one_queue);
}
int side_begin(zkp_ctx* c, hipStream_t* main_out, bool overlap) {
  *main_out = c->stream;
  overlap = side_forks(c, overlap);
  if (!overlap) return ZKP_OK;
  if (!c->side_stream) {
    HIP_TRY(hipStreamCreateWithFlags(&c->side_stream, hipStreamNonBlocking));
    HIP_TRY(hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming));
    HIP_TRY(hipEventCreateWithFlags(&c->ev_join, hipEventDisableTiming));
  }
  HIP_TRY(hipEventRecord(c->ev_fork, c->stream));            // inputs (and everything queued before) are ready
  HIP_TRY(hipStreamWaitEvent(c->side_stream, c->ev_fork, 0));
  c->stream = c->side_stream;
  c->prof_suspended = true;
  return ZKP_OK;
}
int side_end(zkp_ctx* c, hipStream_t main, bool overlap) {
  if (!side_forks(c, overlap)) return ZKP_OK;
  c->prof_suspended = false;
  c->stream = main;
  HIP_TRY(hipEventRecord(c->ev_join, c->side_stream));
  return ZKP_OK;
}
int side_join(zkp_ctx* c, bool overlap) {
  if (side_forks(c, overlap)) HIP_TRY(hipStreamWaitEvent(c->stream, c->ev_join, 0));
  return ZKP_OK;
}

// ---- the flows on device buffers (asynchronous on the context's stream) ----------------------------------------------
struct ws_view {
  char* base;
  uint8_t* u8(size_t o) const { return reinterpret_cast<uint8_t*>(base + o); }
  uint32_t* u32(size_t o) const { return reinterpret_cast<uint32_t*>(base + o); }
};

struct prove_inter { size_t saved, failed, wide, blind, off, sc, pidx, wchal, img, end; };
prove_inter prove_carve(const fused_plan& pl, size_t start) {
  const size_t N = pl.N, m = pl.s.m, nc = pl.s.nc, T = pl.T1;
  carve cv;
  cv.off = start;
  prove_inter o;
  o.saved = cv.take(N * 25 * 8);
  o.failed = cv.take(N * 4 + 4);
  o.wide = cv.take(N * m * 64 + 64);
  o.blind = cv.take(N * m * 32 + 32);
  o.off = cv.take((N * nc + 1) * 4);
  o.sc = cv.take(N * T * 32 + 32);
  o.pidx = cv.take(N * T * 4 + 4);
  o.wchal = cv.take(N * 64 + 64);
  o.img = cv.take(pl.img_bytes);
  o.end = cv.off;
  return o;
}
int prove_core(zkp_ctx* c, const fused_plan& pl, const prove_inter& o, uint8_t* d_ts, const uint8_t* d_sec, const uint8_t* d_tbl,
               const uint8_t* d_ent, uint8_t* d_chal, uint8_t* d_resp, uint8_t* d_coms, uint8_t* d_st8, bool overlap, bool throughput,
               const std::function<int()>* late_inputs = nullptr, const std::function<int()>* early_outputs = nullptr, bool late_early = false) {
  const uint32_t N = pl.N, m = pl.s.m, nc = pl.s.nc, T = pl.T1, n_points = pl.s.ns + pl.s.ni * N;
  const ws_view w{static_cast<char*>(c->ws)};
  tr_bufs hb{};
  hb.src[SRC_TABLE] = d_tbl; hb.src[SRC_SECRETS] = d_sec; hb.src[SRC_ENTROPY] = d_ent; hb.src[SRC_COMS] = d_coms;
  hb.dst[DST_WIDE] = w.u8(o.wide); hb.dst[DST_CHAL] = w.u8(o.wchal);
  uint64_t* d_saved = reinterpret_cast<uint64_t*>(w.base + o.saved);
  prof_begin(c);
  const size_t lanes = std::max<size_t>((size_t)N * T, (size_t)N * nc) + 1;
  terms_cfg tk = cfg_from_terms(pl.tpt.data(), T, pl.s.ns, pl.s.np, N, c->ct_comb_min(throughput, (size_t)N * T));
  tk.throughput = throughput;
  if (pl.d_order) { tk.map.N = N; tk.map.nc = nc; tk.map.order = pl.d_order; }
  tk.stmt.toff = pl.d_tarr; tk.stmt.tpt = pl.d_tarr + nc + 1 + T; tk.stmt.N = N; tk.stmt.T = T; tk.stmt.nc = nc; tk.stmt.ns = pl.s.ns; tk.stmt.np = pl.s.np;
  tk.stmt.off = w.u32(o.off); tk.stmt.pidx = w.u32(o.pidx); tk.stmt.on = c->stmt_classify;
#ifdef ZKP_BUILD_TEST_HOOKS
  for (int q = 0; q < c->debug_dummy_launches; ++q) hipLaunchKernelGGL(k_noop, dim3(1), dim3(64), 0, c->stream, (uint32_t*)nullptr);   // (launch-count sensitivity probe)
#endif
  uint64_t* d_img = pl.img_bytes ? reinterpret_cast<uint64_t*>(w.base + o.img) : nullptr;
  offer_program(c, pl.a, N, hb, d_ts, d_saved, w.u32(o.failed), throughput, overlap, d_img);
  {   // side stream: operand indices, decode, classification, comb tables (nothing here depends on the blindings)
    hipStream_t main;
    int rc = side_begin(c, &main, overlap);
    if (rc) { c->pending_tr.offered = c->pending_tr.active = false; return rc; }
    // (late_early: those inputs sit in PINNED memory -- their copies are DMA that returns at once, so they are queued on the main stream now, behind the fork
    // event the side stream waits for and in front of the point phase's launches, instead of after them)
    if (late_inputs && late_early) {
      hipStream_t cur = c->stream;
      c->stream = main;
      rc = (*late_inputs)();
      c->stream = cur;
      late_inputs = nullptr;
      if (rc) { (void)side_end(c, main, overlap); c->pending_tr.offered = c->pending_tr.active = false; return rc; }
    }
    // (with constraints the term path's classifier writes the CSR offsets and point indices too: k_stmt_classify)
    if (!(nc && stmt_classify_applies(tk, N * T)))
      hipLaunchKernelGGL(k_stmt_index, grid1(lanes, 256), dim3(256), 0, c->stream, N, T, nc, pl.s.ns, pl.d_tarr, pl.d_tarr + nc + 1 + T, w.u32(o.off), w.u32(o.pidx));
    if (nc) rc = msm_terms_path(c, N * nc, w.u32(o.off), nullptr, w.u32(o.pidx), d_tbl, n_points, N * T, ZKP_CT, d_coms, d_st8, nullptr, o.end, false, PH_POINTS, tk);
    const int rc2 = side_end(c, main, overlap);
    if (rc || rc2) { c->pending_tr.offered = c->pending_tr.active = false; return rc ? rc : rc2; }
  }
  // (host-buffer calls on the latency schedule: the inputs only the transcripts read -- states, witnesses, entropy -- cross the link now, while the side
  // stream already decodes and builds tables from the points that went first)
  if (late_inputs) { const int rc = (*late_inputs)(); if (rc) return rc; }
  run_program_pending(c, pl.a, N, hb, d_ts, d_saved, w.u32(o.failed), throughput, d_img);
  prof_mark(c, ZKP_K_TRANSCRIPT);

  // the blindings are canonical (k_wide_reduce), so the halving the batched encoder wants is three instructions per limb here
  // instead of a kernel with a reduction of its own
  tk.prehalved = nc && terms_batched_encode(c, N * T, N * nc, tk.throughput);
  if (m && std::max(m, T) <= 256) {                     // (one launch; every 64-byte string reduced once)
    const uint32_t P = 256 / std::max(m, T);
    hipLaunchKernelGGL(k_blind_scalars, dim3((N + P - 1) / P), dim3(256), 0, c->stream, N, T, m, P, pl.d_tarr + nc + 1, w.u8(o.wide), w.u8(o.blind), w.u8(o.sc),
                       tk.prehalved ? 1u : 0u);
  } else if (m) {
    hipLaunchKernelGGL(k_wide_reduce, grid1((size_t)N * m, 256), dim3(256), 0, c->stream, N * m, w.u8(o.wide), w.u8(o.blind));
    if (T) hipLaunchKernelGGL(k_stmt_scalars, grid1((size_t)N * T, 256), dim3(256), 0, c->stream, N, T, m, pl.d_tarr + nc + 1, w.u8(o.blind), (const uint8_t*)nullptr, w.u8(o.sc),
                              tk.prehalved ? 1u : 0u);
  }
  prof_mark(c, ZKP_K_SCALARS);
  HIP_TRY(hipGetLastError());
  {
    int rc = side_join(c, overlap);
    if (rc) return rc;
    if (nc) rc = msm_terms_path(c, N * nc, w.u32(o.off), w.u8(o.sc), w.u32(o.pidx), d_tbl, n_points, N * T, ZKP_CT, d_coms, d_st8, nullptr, o.end, false, PH_SCALARS, tk);
    if (rc) return rc;
    if (early_outputs) { rc = (*early_outputs)(); if (rc) return rc; }      // (the commitments are final: their copy out starts under program B and the responses)
  }
  run_program(c, pl.b, N, hb, d_ts, d_saved, w.u32(o.failed), throughput, d_img);
  prof_mark(c, ZKP_K_TRANSCRIPT);
  if (m && m <= 256) {
    const uint32_t P = 256 / m;
    hipLaunchKernelGGL(k_responses, dim3((N + P - 1) / P), dim3(256), 0, c->stream, N, m, P, d_sec, w.u8(o.wchal), d_chal, w.u8(o.blind), d_resp);
  } else {
    hipLaunchKernelGGL(k_wide_reduce, grid1(N, 256), dim3(256), 0, c->stream, N, w.u8(o.wchal), d_chal);
    if (m) hipLaunchKernelGGL(k_responses_wide, grid1((size_t)N * m, 256), dim3(256), 0, c->stream, N, m, d_sec, d_chal, w.u8(o.blind), d_resp);
  }
  prof_mark(c, ZKP_K_SCALARS);
  HIP_TRY(hipGetLastError());
  return ZKP_OK;
}

struct verify_inter { size_t failed, mc, off, sc, pidx, coms, st8, wchal, chal, img, end; };
verify_inter verify_carve(const fused_plan& pl, size_t start) {
  const size_t N = pl.N, nc = pl.s.nc, T1 = pl.T1;
  carve cv;
  cv.off = start;
  verify_inter o;
  o.failed = cv.take(N * 4 + 4);
  o.mc = cv.take(N * 32 + 32);
  o.off = cv.take((N * nc + 1) * 4);
  o.sc = cv.take(N * T1 * 32 + 32);
  o.pidx = cv.take(N * T1 * 4 + 4);
  o.coms = cv.take(N * nc * 32 + 32);
  o.st8 = cv.take(N * nc + 4);
  o.wchal = cv.take(N * 64 + 64);
  o.chal = cv.take(N * 32 + 32);
  o.img = cv.take(pl.img_bytes);
  o.end = cv.off;
  return o;
}
// the verifier's CSR job: bounds, statement structure and -- where the statement classifier runs -- the pairs of terms that share a doubling chain
// riders: tables of multiples for points whose terms all ride (stmt_rider).  A table is a chain of 127 additions on one lane: worth it where calls are wide or
// overlap (throughput schedule, or >= kRiderLatencyProofs proofs) -- a lone synchronous call of 4096 CMZ proofs takes 1.89 ms with them and 1.55 ms without,
// 16,384 proofs 4.14 - 4.21 against 4.18 - 4.20 ms.  Workspace bounds are taken with riders = true (the larger layout).
constexpr uint32_t kRiderLatencyProofs = 16384;
terms_cfg verify_terms_cfg(const zkp_ctx* c, const fused_plan& pl, bool riders = true) {
  const uint32_t N = pl.N, nc = pl.s.nc, T1 = pl.T1;
  terms_cfg tk = cfg_from_terms(pl.tpt.data(), T1, pl.s.ns, pl.s.np, N, 2);
  tk.stmt.toff = pl.d_tarr; tk.stmt.tpt = pl.d_tarr + nc + 1 + T1; tk.stmt.N = N; tk.stmt.T = T1; tk.stmt.nc = nc; tk.stmt.ns = pl.s.ns; tk.stmt.np = pl.s.np;
  tk.stmt.on = c->stmt_classify;
  if (c->joint_ladder && pl.d_pair && stmt_classify_applies(tk, N * T1)) {
    riders = riders && c->rider_tables;
    const terms_cfg paired = cfg_from_terms(pl.tpt.data(), T1, pl.s.ns, pl.s.np, N, 2, pl.pair.data(), riders);
    tk.max_tables = paired.max_tables; tk.max_ladder = paired.max_ladder; tk.teeth = paired.teeth;
    tk.stmt.pair = pl.d_pair;
    tk.rider_tables = riders;                    // (the classifier hands them out only if the job's tables have 16 teeth: 129 entries hold 128 multiples)
  }
  return tk;
}
int verify_core(zkp_ctx* c, const fused_plan& pl, const verify_inter& o, uint8_t* d_ts, const uint8_t* d_tbl, const uint8_t* d_claim,
                const uint8_t* d_resp, uint8_t* d_results, bool overlap, bool throughput, const std::function<int()>* late_inputs = nullptr, bool late_early = false) {
  const uint32_t N = pl.N, m = pl.s.m, nc = pl.s.nc, T1 = pl.T1, n_points = pl.s.ns + pl.s.ni * N;
  const ws_view w{static_cast<char*>(c->ws)};
  tr_bufs hb{};
  hb.src[SRC_TABLE] = d_tbl; hb.src[SRC_COMS] = w.u8(o.coms);
  hb.dst[DST_CHAL] = w.u8(o.wchal);
  HIP_TRY(hipMemsetAsync(w.base + o.failed, 0, (size_t)N * 4, c->stream));
  prof_begin(c);
  const size_t lanes = std::max<size_t>((size_t)N * T1, (size_t)N * nc) + 1;
  terms_cfg tk = verify_terms_cfg(c, pl, throughput || N >= kRiderLatencyProofs);
  tk.throughput = throughput;
  if (pl.d_order) { tk.map.N = N; tk.map.nc = nc; tk.map.order = pl.d_order; }
  tk.stmt.off = w.u32(o.off); tk.stmt.pidx = w.u32(o.pidx);
  uint64_t* d_img = pl.img_bytes ? reinterpret_cast<uint64_t*>(w.base + o.img) : nullptr;
  offer_program(c, pl.a, N, hb, d_ts, nullptr, w.u32(o.failed), throughput, overlap, d_img);
  {   // side stream: operand indices and the point phase.  With no constraints there is no MSM, but every allocated
      // point must still decode (verifier.rs:87-92)
    hipStream_t main;
    int rc = side_begin(c, &main, overlap);
    if (rc) { c->pending_tr.offered = c->pending_tr.active = false; return rc; }
    if (late_inputs && late_early) {                     // (pinned inputs: see prove_core)
      hipStream_t cur = c->stream;
      c->stream = main;
      rc = (*late_inputs)();
      c->stream = cur;
      late_inputs = nullptr;
      if (rc) { (void)side_end(c, main, overlap); c->pending_tr.offered = c->pending_tr.active = false; return rc; }
    }
    if (!stmt_classify_applies(tk, N * T1))
      hipLaunchKernelGGL(k_stmt_index, grid1(lanes, 256), dim3(256), 0, c->stream, N, T1, nc, pl.s.ns, pl.d_tarr, pl.d_tarr + nc + 1 + T1, w.u32(o.off), w.u32(o.pidx));
    rc = msm_terms_path(c, N * nc, w.u32(o.off), nullptr, w.u32(o.pidx), d_tbl, n_points, N * T1, ZKP_VARTIME, w.u8(o.coms), w.u8(o.st8), nullptr, o.end, /*decode_all=*/true, PH_POINTS, tk);
    const int rc2 = side_end(c, main, overlap);
    if (rc || rc2) { c->pending_tr.offered = c->pending_tr.active = false; return rc ? rc : rc2; }
  }
  if (late_inputs) { const int rc = (*late_inputs)(); if (rc) return rc; }
  run_program_pending(c, pl.a, N, hb, d_ts, nullptr, w.u32(o.failed), throughput, d_img);
  prof_mark(c, ZKP_K_TRANSCRIPT);
  hipLaunchKernelGGL(k_neg_reduce, grid1(N, 256), dim3(256), 0, c->stream, N, d_claim, w.u8(o.mc));
  if (T1) hipLaunchKernelGGL(k_stmt_scalars, grid1((size_t)N * T1, 256), dim3(256), 0, c->stream, N, T1, m, pl.d_tarr + nc + 1, d_resp, w.u8(o.mc), w.u8(o.sc), 0u);
  prof_mark(c, ZKP_K_SCALARS);
  HIP_TRY(hipGetLastError());
  int rc = side_join(c, overlap);
  if (rc) return rc;
  rc = msm_terms_path(c, N * nc, w.u32(o.off), w.u8(o.sc), w.u32(o.pidx), d_tbl, n_points, N * T1, ZKP_VARTIME, w.u8(o.coms), w.u8(o.st8), nullptr, o.end, /*decode_all=*/true, PH_SCALARS, tk);
  if (rc) return rc;
  run_program(c, pl.b, N, hb, d_ts, nullptr, w.u32(o.failed), throughput, d_img);
  prof_mark(c, ZKP_K_TRANSCRIPT);
  hipLaunchKernelGGL(k_wide_reduce, grid1(N, 256), dim3(256), 0, c->stream, N, w.u8(o.wchal), w.u8(o.chal));
  // the decoded point table is the first thing msm_terms_path carves after its reserved prefix
  hipLaunchKernelGGL(k_verify_finish, grid1(N, 256), dim3(256), 0, c->stream, N, nc, pl.s.ns, m, w.u8(o.chal), d_claim, d_resp, w.u8(o.st8), w.u32(o.failed),
                     reinterpret_cast<const dev_affine*>(w.base + o.end), pl.d_tarr + nc + 1 + 2 * (size_t)T1, (uint32_t)pl.s.unref.size(), d_results);
  prof_mark(c, ZKP_K_SCALARS);
  HIP_TRY(hipGetLastError());
  return ZKP_OK;
}

// K batch verifications of N_each = pl.N / K proofs each in one pass (K = 1: zkp_fused_batch_verify).  The N = pl.N proofs of the
// call lie next to each other, batch b = proofs [b N_each, (b + 1) N_each).
// d_pts = [ns + (ni + nc) N][32] with static || instance rows filled in by the caller; the commitment rows are written here
struct batch_inter { size_t sc, failed, flags, wchal, mc, part, img, end; };
batch_inter batch_carve(const fused_plan& pl, size_t start, uint32_t K = 1) {
  const size_t N = pl.N, ns = pl.s.ns, N_each = N / (K ? K : 1), total = (size_t)K * ns + ((size_t)pl.s.ni + pl.s.nc) * N, nblk = (size_t)K * ((N_each + 255) / 256);
  carve cv;
  cv.off = start;
  batch_inter o;
  o.sc = cv.take(total * 32 + 32);
  o.failed = cv.take(N * 4 + 4);
  o.flags = cv.take((size_t)K * 4);
  o.wchal = cv.take(N * 64 + 64);
  o.mc = cv.take(N * 32 + 32);
  o.part = cv.take((ns ? ns : 1) * (nblk ? nblk : 1) * 32);
  o.img = cv.take(pl.img_bytes);
  o.end = cv.off;
  return o;
}
size_t optional_many_ws(uint64_t n_each, uint32_t K) {
  switch (pick_c(n_each)) {
    case 7: return pip_ws<7>(n_each, K);
    case 10: return pip_ws<10>(n_each, K);
    case 11: return pip_ws<11>(n_each, K);
    default: return pip_ws<16>(n_each, K);
  }
}
int batch_core(zkp_ctx* c, const fused_plan& pl, const batch_inter& o, uint8_t* d_ts, uint8_t* d_pts, const uint8_t* d_coms,
               const uint8_t* d_resp, const uint8_t* d_w, uint8_t* d_out /*[K][32]*/,
               uint32_t* d_status /*[K][2]: MSM decode failure | transcript rejection or non-canonical response*/, bool throughput, uint32_t K = 1,
               bool overlap = false, const std::function<int()>* late_ts = nullptr, const std::function<int()>* late_scalars = nullptr, bool late_early = false) {
  const uint32_t N = pl.N, nc = pl.s.nc, ns = pl.s.ns, ni = pl.s.ni, N_each = N / K;
  const size_t n_each = (size_t)ns + ((size_t)ni + nc) * N_each;          // terms of one batch's MSM (batch_verifier.rs:219-228)
  const ws_view w{static_cast<char*>(c->ws)};
  // Status words without memsets (K = 1): the transcript kernel zeroes the spare word behind its rejection flags, k_batch_after_transcript
  // sets bit 1 of it if a proof was rejected, the MSM sets bit 0 on a decode failure, and the MSM's last kernel writes both
  // status words (Pippenger sizes; tiny batches keep the two memsets).  K > 1: one flag word per batch, cleared by one memset.
  uint32_t* shared = nullptr;
  if (K > 1) {
    shared = w.u32(o.flags);
    HIP_TRY(hipMemsetAsync(shared, 0, (size_t)K * 4, c->stream));
  } else if (N && n_each > kSmallOptional && pl.a.n) {
    shared = w.u32(o.failed) + N;
  }
  if (!shared) HIP_TRY(hipMemsetAsync(d_status, 0, 8, c->stream));
  prof_begin(c);
  uint64_t* d_img = pl.img_bytes ? reinterpret_cast<uint64_t*>(w.base + o.img) : nullptr;
  // Latency schedule (one call in flight: the synchronous entry points, ZKP_OPT_DEV_OVERLAP = 2): the point half of the MSM's prepare step -- 24 N + 12
  // decompressions, batch_verifier.rs:226, which no scalar enters -- runs on the side stream next to the transcript chain (0.13 of a lone call's 0.85 ms
  // at 4096 proofs).  Needs the step form: its assemble pass clears the shared flag word BEFORE the fork, so the decoder's atomicOr cannot meet it.
  const bool split = overlap && K == 1 && N && shared && transcript_steps(c, pl.a, N, throughput, d_img);
  if (N) {
    tr_bufs hb{};
    hb.src[SRC_TABLE] = d_pts; hb.src[SRC_COMS] = d_coms;
    hb.dst[DST_CHAL] = w.u8(o.wchal);
    if (split) {
      run_program(c, pl.a, N, hb, d_ts, nullptr, w.u32(o.failed), throughput, d_img, /*owns_failed=*/true, /*phase=*/1);
      hipStream_t main;
      int rc = side_begin(c, &main, true);
      if (rc) return rc;
      if (late_early) {                                  // (pinned inputs: see prove_core)
        hipStream_t cur = c->stream;
        c->stream = main;
        if (late_ts) rc = (*late_ts)();
        if (!rc && late_scalars) rc = (*late_scalars)();
        c->stream = cur;
        late_ts = late_scalars = nullptr;
        if (rc) { (void)side_end(c, main, true); return rc; }
      }
      if (nc) hipLaunchKernelGGL(k_transpose_commitments, grid1((size_t)N * nc, 256), dim3(256), 0, c->stream, N, nc, d_coms, d_pts + 32 * ((size_t)ns + (size_t)ni * N));
      rc = msm_optional_impl(c, n_each, w.u8(o.sc), d_pts, d_out, d_status, o.end, shared, /*phases=*/1);
      const int rc2 = side_end(c, main, true);
      if (rc || rc2) return rc ? rc : rc2;
      if (late_ts) { rc = (*late_ts)(); if (rc) return rc; }             // (host-buffer calls: the transcript states cross the link while the side stream decodes)
      run_program(c, pl.a, N, hb, d_ts, nullptr, w.u32(o.failed), throughput, d_img, /*owns_failed=*/true, /*phase=*/2);
    } else {
      if (late_ts) { const int rc = (*late_ts)(); if (rc) return rc; }
      run_program(c, pl.a, N, hb, d_ts, nullptr, w.u32(o.failed), throughput, d_img, /*owns_failed=*/true);
    }
    prof_mark(c, ZKP_K_TRANSCRIPT);
    if (late_scalars) { const int rc = (*late_scalars)(); if (rc) return rc; }   // (responses and weights: nobody reads them before the chain is on its stream)
    const size_t lanes = std::max<size_t>(std::max<size_t>(N, (size_t)N * nc), (size_t)N * pl.s.m);
    hipLaunchKernelGGL(k_batch_after_transcript, grid1(lanes, 256), dim3(256), 0, c->stream, N, N_each, nc, pl.s.m, w.u32(o.failed),
                       shared ? shared : d_status + 1, 1u, shared ? 2u : 1u, w.u8(o.wchal), w.u8(o.mc), split ? (const uint8_t*)nullptr : d_coms,
                       d_pts + 32 * ((size_t)ns + (size_t)ni * N), d_resp);
  }
  launch_coeff_build(c, pl.s, N_each, pl.d_inc, w.u8(o.mc), d_resp, d_w, w.u8(o.sc), w.u32(o.part), K);
  prof_mark(c, ZKP_K_SCALARS);
  HIP_TRY(hipGetLastError());
  if (split) {
    const int rc = side_join(c, true);
    if (rc) return rc;
    return msm_optional_impl(c, n_each, w.u8(o.sc), d_pts, d_out, d_status, o.end, shared, /*phases=*/2);
  }
  if (K == 1) return msm_optional_impl(c, n_each, w.u8(o.sc), d_pts, d_out, d_status, o.end, shared);
  pip_seg seg;
  seg.K = K; seg.ns = ns; seg.N_each = N_each;
  switch (pick_c(n_each)) {
    case 7: return pip_run<7>(c, (uint32_t)n_each, w.u8(o.sc), d_pts, d_out, d_status, o.end, shared, seg);
    case 10: return pip_run<10>(c, (uint32_t)n_each, w.u8(o.sc), d_pts, d_out, d_status, o.end, shared, seg);
    case 11: return pip_run<11>(c, (uint32_t)n_each, w.u8(o.sc), d_pts, d_out, d_status, o.end, shared, seg);
    default: return pip_run<16>(c, (uint32_t)n_each, w.u8(o.sc), d_pts, d_out, d_status, o.end, shared, seg);
  }
}

// d_tbl = [ns + ni N + N nc][32]: common || instance rows || commitments [N][nc] (the last part doubles as the
// transcripts' commitment source)
struct each_inter { size_t failed, wchal, mc, off, sc, pidx, out, st8, img, end; };
each_inter each_carve(const fused_plan& pl, size_t start) {
  const size_t N = pl.N, K = (size_t)pl.s.np + pl.s.nc;
  carve cv;
  cv.off = start;
  each_inter o;
  o.failed = cv.take(N * 4 + 4);
  o.wchal = cv.take(N * 64 + 64);
  o.mc = cv.take(N * 32 + 32);
  o.off = cv.take((N + 1) * 4);
  o.sc = cv.take(N * K * 32 + 32);
  o.pidx = cv.take(N * K * 4 + 4);
  o.out = cv.take(N * 32 + 32);
  o.st8 = cv.take(N + 4);
  o.img = cv.take(pl.img_bytes);
  o.end = cv.off;
  return o;
}
// verify_batchable's MSM multiplies every point of a proof exactly once: per-proof points are single-use (ladder), a
// common point without a fixed-base table is shared by the N proofs (comb table)
terms_cfg each_terms_cfg(const fused_plan& pl) {
  terms_cfg k;
  k.max_tables = pl.N >= 2 ? pl.s.ns : 0;
  k.max_ladder = (uint32_t)std::min<uint64_t>(((uint64_t)pl.s.ni + pl.s.nc) * pl.N + pl.s.ns, 0xffffffffu);
  k.teeth = pl.N >= 6 ? 16 : 4;
  return k;
}
// The Straus path of verify_batchable (k_straus_each) and its workspace behind each_inter
struct straus_inter { size_t pts, tab, digits, spart, end; };
constexpr uint32_t kStrausMaxLanes = 8, kStrausMaxOpsPerLane = 60;      // (60 KB of dynamic LDS per block at most)
// Window parts per proof (k_straus_each_win), 0 = the walk split by operands.  Measured (profiles/r03_ab_verify_batchable_each.txt): 32 parts
// of two windows are the fastest split from 256 to 32,768 proofs (16 and 64 within 3 %); from 65,536 proofs on one lane per proof fills
// the chip by itself and the joining pass only adds work.
constexpr uint32_t kStrausWinParts = 32, kStrausWinMaxProofs = 65536;
inline uint32_t straus_win_parts(const zkp_ctx* c, const fused_plan& pl) {
  const uint32_t K = pl.s.np + pl.s.nc, N = pl.N;
  if (K > 64 || c->each_straus_lanes) return 0;                          // (K: one bit per operand in the lane's sign mask, K KB of LDS per block)
  if (c->each_straus_wins) return c->each_straus_wins;
  return N < kStrausWinMaxProofs ? kStrausWinParts : 0;
}
inline uint32_t straus_lanes(const zkp_ctx* c, const fused_plan& pl) {
  const uint32_t K = pl.s.np + pl.s.nc, N = pl.N;
  uint32_t L = c->each_straus_lanes ? std::min<uint32_t>(c->each_straus_lanes, kStrausMaxLanes) : (N >= 65536 ? 1u : (N >= 32768 ? 2u : (N >= 16384 ? 4u : 8u)));
  while (L < kStrausMaxLanes && (K + L - 1) / L > kStrausMaxOpsPerLane) L *= 2;
  // never more lanes than operands: a lane without an operand would still fetch "its" first table entry every window (an LDS column
  // nobody wrote, a gather past the proof's commitments)
  return std::max(1u, std::min(std::min(L, kStrausMaxLanes), K));
}
inline bool each_uses_straus(const zkp_ctx* c, const fused_plan& pl) {
  const uint64_t K = (uint64_t)pl.s.np + pl.s.nc;
  return c->each_straus && K >= 4 && (K + kStrausMaxLanes - 1) / kStrausMaxLanes <= kStrausMaxOpsPerLane;       // (tiny statements: nothing to share)
}
straus_inter straus_carve(const zkp_ctx* c, const fused_plan& pl, size_t start) {
  const size_t N = pl.N, K = (size_t)pl.s.np + pl.s.nc, n_points = (size_t)pl.s.ns + (size_t)pl.s.ni * N + N * pl.s.nc;
  carve cv;
  cv.off = start;
  straus_inter o;
  o.pts = cv.take(n_points * sizeof(dev_affine));
  o.tab = cv.take(n_points * 8 * sizeof(straus_entry) + 128);
  o.digits = cv.take(N * K * 9 * 4);
  o.spart = cv.take(N * (std::max(kStrausMaxLanes, straus_win_parts(c, pl)) + 1) * sizeof(dev_ext));
  o.end = cv.off;
  return o;
}
int each_core(zkp_ctx* c, const fused_plan& pl, const each_inter& o, uint8_t* d_ts, const uint8_t* d_tbl, const uint8_t* d_resp,
              const uint8_t* d_w, uint8_t* d_results, bool overlap) {
  const uint32_t N = pl.N, nc = pl.s.nc, ns = pl.s.ns, ni = pl.s.ni, K = pl.s.np + nc;
  const uint32_t n_points = ns + ni * N + N * nc;
  const ws_view w{static_cast<char*>(c->ws)};
  const uint8_t* d_coms = d_tbl + 32 * ((size_t)ns + (size_t)ni * N);
  tr_bufs hb{};
  hb.src[SRC_TABLE] = d_tbl; hb.src[SRC_COMS] = d_coms;
  hb.dst[DST_CHAL] = w.u8(o.wchal);
  HIP_TRY(hipMemsetAsync(w.base + o.failed, 0, (size_t)N * 4, c->stream));
  prof_begin(c);
  // one Straus MSM per proof (see k_straus_each): its points decode and its tables are built on the side stream, next to the transcripts
  // (overlap: the synchronous entry point, one call in flight per context, latency matters -- see side_begin; the _dev entry point
  // follows ZKP_OPT_DEV_OVERLAP like the other flows)
  const bool straus = each_uses_straus(c, pl);
  straus_inter so{};
  dev_affine* pts = nullptr;
  straus_entry* tab = nullptr;
  if (straus) {
    so = straus_carve(c, pl, o.end);
    if (so.end > c->ws_bytes) return fail(ZKP_ERR_ARG, "internal: workspace too small");
    pts = reinterpret_cast<dev_affine*>(w.base + so.pts);
    tab = reinterpret_cast<straus_entry*>(w.base + so.tab);
    hipStream_t main;
    int rc = side_begin(c, &main, overlap);
    if (rc) return rc;
    hipLaunchKernelGGL(k_decode_affine, grid1(n_points, 256), dim3(256), 0, c->stream, n_points, d_tbl, pts, (const uint32_t*)nullptr);
    hipLaunchKernelGGL(k_straus_tables, grid1(n_points, 256), dim3(256), 0, c->stream, n_points, pts, tab);
    rc = side_end(c, main, overlap);
    if (rc) return rc;
  }
  run_program(c, pl.a, N, hb, d_ts, nullptr, w.u32(o.failed), false, pl.img_bytes ? reinterpret_cast<uint64_t*>(w.base + o.img) : nullptr);
  prof_mark(c, ZKP_K_TRANSCRIPT);
  hipLaunchKernelGGL(k_wide_reduce, grid1(N, 256), dim3(256), 0, c->stream, N, w.u8(o.wchal), w.u8(o.mc));
  hipLaunchKernelGGL(k_neg_reduce, grid1(N, 256), dim3(256), 0, c->stream, N, w.u8(o.mc), w.u8(o.mc));
  const uint32_t* d_inc_k = pl.d_inc + pl.s.inc_off.size();
  hipLaunchKernelGGL(k_each_coeffs, grid1(std::max<size_t>((size_t)N * K, N) + 1, 256), dim3(256), 0, c->stream, N, pl.s.m, ns, ni, nc, pl.d_inc, d_inc_k,
                     d_inc_k + pl.s.inc_k.size(), w.u8(o.mc), d_resp, d_w, w.u32(o.off), w.u8(o.sc), w.u32(o.pidx));
  prof_mark(c, ZKP_K_SCALARS);
  HIP_TRY(hipGetLastError());
  if (straus) {
    dev_ext* spart = reinterpret_cast<dev_ext*>(w.base + so.spart);
    const uint32_t Wn = straus_win_parts(c, pl), L = Wn ? 1 : straus_lanes(c, pl);
    if (Wn) hipLaunchKernelGGL(k_straus_recode, grid1((size_t)N * K, 256), dim3(256), 0, c->stream, N * K, w.u8(o.sc), w.u32(so.digits));
    const int rcj = side_join(c, overlap);
    if (rcj) return rcj;
    prof_mark(c, ZKP_K_TABLES);                                        // (what the main stream waited for the side stream, if anything)
    if (Wn) {
      hipLaunchKernelGGL(k_straus_each_win, grid1((size_t)N * Wn, 256), dim3(256), (size_t)K * 1024, c->stream, N, K, Wn, ns, ni, nc, tab, w.u32(so.digits), spart + N);
      hipLaunchKernelGGL(k_straus_combine_quad, grid1((size_t)N * 4, 256), dim3(256), 0, c->stream, N, Wn, spart + N, spart);
    } else
      hipLaunchKernelGGL(k_straus_each, grid1((size_t)N * L, 256), dim3(256), (size_t)((K + L - 1) / L) * 1024, c->stream, N, K, L, ns, ni, nc, w.u8(o.sc), tab,
                         w.u32(so.digits), spart);
    prof_mark(c, ZKP_K_TERMS);
    hipLaunchKernelGGL(k_straus_finish, grid1(N, 256), dim3(256), 0, c->stream, N, K, L, w.u32(o.pidx), pts, spart, w.u8(o.out), w.u8(o.st8));
    prof_mark(c, ZKP_K_REDUCE);
    HIP_TRY(hipGetLastError());
  } else {
    const int rc = msm_terms_path(c, N, w.u32(o.off), w.u8(o.sc), w.u32(o.pidx), d_tbl, n_points, N * K, ZKP_VARTIME, w.u8(o.out), w.u8(o.st8), nullptr, o.end, /*decode_all=*/true,
                                  PH_ALL, each_terms_cfg(pl));
    if (rc) return rc;
  }
  hipLaunchKernelGGL(k_each_finish, grid1(N, 256), dim3(256), 0, c->stream, N, pl.s.m, w.u8(o.out), w.u8(o.st8), w.u32(o.failed), d_resp, d_results);
  prof_mark(c, ZKP_K_SCALARS);
  HIP_TRY(hipGetLastError());
  return ZKP_OK;
}

}  // namespace

void free_fused_plans(zkp_ctx* c) {
  if (!c->fused_plans.empty()) ++c->plans_generation;      // graphs captured over these plans' device blocks are stale from here on
  for (auto& kv : c->fused_plans) {
    fused_plan* pl = static_cast<fused_plan*>(kv.second);
    if (pl->d_block) hipFree(pl->d_block);
    delete pl;
  }
  c->fused_plans.clear();
}

extern "C" {

int zkp_batch_check(zkp_ctx* c, const zkp_batch_statement* st, uint32_t N, const uint8_t* minus_c, const uint8_t* responses,
                    const uint8_t* weights16, const uint8_t* static_points, const uint8_t* instance_points,
                    const uint8_t* commitments, uint8_t out_point[32], int* status, uint8_t* debug_scalars) {
  if (!c || !st || !out_point || !status) return fail(ZKP_ERR_ARG, "NULL pointer");
  fused_shape s;
  int rc = parse_shape(st, s);
  if (rc) return rc;
  const uint32_t m = s.m, ns = s.ns, ni = s.ni, nc = s.nc;
  if (N && ((!minus_c) || (m && !responses) || (nc && (!weights16 || !commitments)) || (ni && !instance_points)))
    return fail(ZKP_ERR_ARG, "NULL input pointer");
  if (ns && !static_points) return fail(ZKP_ERR_ARG, "NULL static points");
  HIP_TRY(hipSetDevice(c->device));
  const size_t rows = (size_t)ni + nc, total = (size_t)ns + rows * N;
  if (total > 0x7fffffffull) return fail(ZKP_ERR_ARG, "batch too large");
  const uint32_t nblk = (N + 255) / 256;
  const std::vector<uint32_t> inc = incidence_words(s);
  carve cv;
  const size_t o_sc = cv.take(total * 32 + 32);
  const size_t o_pts = cv.take(total * 32 + 32);
  const size_t o_out = cv.take(32);
  const size_t o_st = cv.take(4);
  const size_t o_mc = cv.take((size_t)N * 32);
  const size_t o_resp = cv.take((size_t)N * m * 32);
  const size_t o_w = cv.take((size_t)nc * N * 16);
  const size_t o_coms = cv.take((size_t)nc * N * 32);
  const size_t o_inc = cv.take(inc.size() * 4 + 16);
  const size_t o_part = cv.take((size_t)(ns ? ns : 1) * (nblk ? nblk : 1) * 32);
  const size_t reserved = cv.off;
  rc = ensure_ws(c, reserved + optional_ws(total));
  if (rc) return rc;
  char* base = static_cast<char*>(c->ws);
  uint8_t* d_sc = reinterpret_cast<uint8_t*>(base + o_sc);
  uint8_t* d_pts = reinterpret_cast<uint8_t*>(base + o_pts);
  // operands of batch_verifier.rs:219-228: points = static || instance rows || commitment rows (row = constraint)
  if (ns) HIP_TRY(hipMemcpyAsync(d_pts, static_points, (size_t)ns * 32, hipMemcpyHostToDevice, c->stream));
  if (ni && N) HIP_TRY(hipMemcpyAsync(d_pts + 32 * (size_t)ns, instance_points, (size_t)ni * N * 32, hipMemcpyHostToDevice, c->stream));
  if (nc && N) {
    HIP_TRY(hipMemcpyAsync(base + o_coms, commitments, (size_t)nc * N * 32, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipMemcpyAsync(base + o_w, weights16, (size_t)nc * N * 16, hipMemcpyHostToDevice, c->stream));
  }
  if (N) {
    HIP_TRY(hipMemcpyAsync(base + o_mc, minus_c, (size_t)N * 32, hipMemcpyHostToDevice, c->stream));
    if (m) HIP_TRY(hipMemcpyAsync(base + o_resp, responses, (size_t)N * m * 32, hipMemcpyHostToDevice, c->stream));
  }
  HIP_TRY(hipMemcpyAsync(base + o_inc, inc.data(), inc.size() * 4, hipMemcpyHostToDevice, c->stream));
  prof_begin(c);
  if (nc && N)
    hipLaunchKernelGGL(k_transpose_commitments, grid1((size_t)N * nc, 256), dim3(256), 0, c->stream, N, nc,
                       reinterpret_cast<const uint8_t*>(base + o_coms), d_pts + 32 * ((size_t)ns + (size_t)ni * N));
  launch_coeff_build(c, s, N, reinterpret_cast<uint32_t*>(base + o_inc), reinterpret_cast<uint8_t*>(base + o_mc),
                     reinterpret_cast<uint8_t*>(base + o_resp), reinterpret_cast<uint8_t*>(base + o_w), d_sc,
                     reinterpret_cast<uint32_t*>(base + o_part));
  prof_mark(c, ZKP_K_SCALARS);
  HIP_TRY(hipGetLastError());
  if (debug_scalars) HIP_TRY(hipMemcpyAsync(debug_scalars, d_sc, total * 32, hipMemcpyDeviceToHost, c->stream));
  rc = msm_optional_impl(c, total, d_sc, d_pts, reinterpret_cast<uint8_t*>(base + o_out), reinterpret_cast<uint32_t*>(base + o_st), reserved);
  if (rc) return rc;
  uint32_t stv = 1;
  HIP_TRY(hipMemcpyAsync(out_point, static_cast<char*>(c->ws) + o_out, 32, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipMemcpyAsync(&stv, static_cast<char*>(c->ws) + o_st, 4, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  *status = (int)stv;
  return ZKP_OK;
}

// (the host-pointer twins of the entry points below -- zkp_fused_prove, _verify_compact, _batch_verify[_many], _verify_batchable -- are
//  host_jobs.h: the same flows as asynchronous jobs, followed by zkp_ctx_job_wait)
// ---- prove ---------------------------------------------------------------------------------------------------------
int zkp_fused_prove_dev(zkp_ctx* c, const zkp_fused_statement* st, uint32_t N, uint32_t strobe_pos, uint8_t* d_transcripts,
                        const uint8_t* d_secrets, const uint8_t* d_table, const uint8_t* d_entropy, uint8_t* d_challenges,
                        uint8_t* d_responses, uint8_t* d_commitments, uint8_t* d_status) {
  if (!c) return fail(ZKP_ERR_ARG, "ctx is NULL");
  if (N == 0) return ZKP_OK;
  HIP_TRY(hipSetDevice(c->device));
  fused_plan* pl = nullptr;
  int rc = get_plan(c, FLOW_PROVE, st, N, strobe_pos, &pl);
  if (rc) return rc;
  const fused_shape& s = pl->s;
  if (!d_transcripts || !d_entropy || !d_challenges || (s.m && (!d_secrets || !d_responses)) || (s.nc && (!d_commitments || !d_status)) || (s.np && !d_table))
    return fail(ZKP_ERR_ARG, "NULL device pointer");
  if ((uint64_t)N * s.T > 0x7fffffffull || (uint64_t)s.ns + (uint64_t)s.ni * N > 0x7fffffffull) return fail(ZKP_ERR_ARG, "batch too large");
  const prove_inter o = prove_carve(*pl, 0);
  rc = ensure_ws(c, o.end + terms_path_ws(s.ns + s.ni * N, N * s.T, N * s.nc, cfg_from_terms(pl->tpt.data(), s.T, s.ns, s.np, N, c->ct_comb_min(!c->dev_latency, (size_t)N * s.T))));
  if (rc) return rc;
  return prove_core(c, *pl, o, d_transcripts, d_secrets, d_table, d_entropy, d_challenges, d_responses, d_commitments, d_status, /*overlap=*/c->dev_overlap, /*throughput=*/!c->dev_latency);
}

// ---- verify_compact --------------------------------------------------------------------------------------------------
int zkp_fused_verify_compact_dev(zkp_ctx* c, const zkp_fused_statement* st, uint32_t N, uint32_t strobe_pos, uint8_t* d_transcripts,
                                 const uint8_t* d_table, const uint8_t* d_challenges, const uint8_t* d_responses, uint8_t* d_results) {
  if (!c) return fail(ZKP_ERR_ARG, "ctx is NULL");
  if (N == 0) return ZKP_OK;
  HIP_TRY(hipSetDevice(c->device));
  fused_plan* pl = nullptr;
  int rc = get_plan(c, FLOW_VERIFY, st, N, strobe_pos, &pl);
  if (rc) return rc;
  const fused_shape& s = pl->s;
  if (!d_transcripts || !d_challenges || !d_results || (s.m && !d_responses) || (s.np && !d_table)) return fail(ZKP_ERR_ARG, "NULL device pointer");
  if ((uint64_t)N * pl->T1 > 0x7fffffffull || (uint64_t)s.ns + (uint64_t)s.ni * N > 0x7fffffffull) return fail(ZKP_ERR_ARG, "batch too large");
  const verify_inter o = verify_carve(*pl, 0);
  rc = ensure_ws(c, o.end + terms_path_ws(s.ns + s.ni * N, N * pl->T1, N * s.nc, verify_terms_cfg(c, *pl)));
  if (rc) return rc;
  return verify_core(c, *pl, o, d_transcripts, d_table, d_challenges, d_responses, d_results, /*overlap=*/c->dev_overlap, /*throughput=*/!c->dev_latency);
}

// ---- batch verification ----------------------------------------------------------------------------------------------
int zkp_fused_batch_verify_dev(zkp_ctx* c, const zkp_fused_statement* st, uint32_t N, uint32_t strobe_pos, uint8_t* d_transcripts,
                               uint8_t* d_points, const uint8_t* d_commitments, const uint8_t* d_responses, const uint8_t* d_weights16,
                               uint8_t* d_out_point, uint32_t* d_status) {
  if (!c) return fail(ZKP_ERR_ARG, "ctx is NULL");
  HIP_TRY(hipSetDevice(c->device));
  fused_plan* pl = nullptr;
  int rc = get_plan(c, FLOW_BATCH, st, N, N ? strobe_pos : 0, &pl);
  if (rc) return rc;
  const fused_shape& s = pl->s;
  if (!d_out_point || !d_status || (N && (!d_transcripts || (s.nc && (!d_commitments || !d_weights16)) || (s.m && !d_responses))) ||
      ((s.ns || (N && (s.ni || s.nc))) && !d_points))
    return fail(ZKP_ERR_ARG, "NULL device pointer");
  if (!aligned16(d_transcripts) || !aligned16(d_points) || !aligned16(d_commitments) || !aligned16(d_responses) || !aligned16(d_weights16) || !aligned16(d_out_point))
    return fail(ZKP_ERR_ARG, "device buffers must be 16-byte aligned");
  const size_t total = (size_t)s.ns + ((size_t)s.ni + s.nc) * N;
  if (total > 0x7fffffffull) return fail(ZKP_ERR_ARG, "batch too large");
  const batch_inter o = batch_carve(*pl, 0);
  rc = ensure_ws(c, o.end + optional_ws(total));
  if (rc) return rc;
  return batch_core(c, *pl, o, d_transcripts, d_points, d_commitments, d_responses, d_weights16, d_out_point, d_status, /*throughput=*/!c->dev_latency, 1, /*overlap=*/c->dev_latency);
}

// ---- K batch verifications in one pass ------------------------------------------------------------------------------
static int check_many(uint32_t K, uint32_t N_each, const fused_shape& s, uint64_t* n_each_out) {
  if (K == 0 || N_each == 0) return fail(ZKP_ERR_ARG, "n_batches and N_each must be positive");
  if ((uint64_t)K * N_each > 0x7fffffffull) return fail(ZKP_ERR_ARG, "batch too large");
  const uint64_t n_each = (uint64_t)s.ns + ((uint64_t)s.ni + s.nc) * N_each;
  // the segmented Pippenger launches grids with one y-coordinate per (batch, window): K * W1(c) of them for the window size c that
  // n_each selects (pip_cfg<c>::W1 = ceil(256 / c) + 1), and a grid's y dimension ends at 65,535
  const uint64_t W1 = [&]() -> uint64_t {
    switch (pick_c(n_each)) { case 7: return pip_cfg<7>::W1; case 10: return pip_cfg<10>::W1; case 11: return pip_cfg<11>::W1; default: return pip_cfg<16>::W1; }
  }();
  if (n_each * K > 0x7fffffffull || (uint64_t)K * W1 > 65535) return fail(ZKP_ERR_ARG, "batch too large (n_batches x windows exceeds the grid)");
  *n_each_out = n_each;
  return ZKP_OK;
}
int zkp_fused_batch_verify_many_dev(zkp_ctx* c, const zkp_fused_statement* st, uint32_t K, uint32_t N_each, uint32_t strobe_pos,
                                    uint8_t* d_transcripts, uint8_t* d_points, const uint8_t* d_commitments, const uint8_t* d_responses,
                                    const uint8_t* d_weights16, uint8_t* d_out_points, uint32_t* d_status) {
  if (!c) return fail(ZKP_ERR_ARG, "ctx is NULL");
  HIP_TRY(hipSetDevice(c->device));
  // the K-batch kernels read these with 16-byte vector loads (k_batch_after_transcript, k_coeff_build, k_pip_prepare)
  if (!aligned16(d_transcripts) || !aligned16(d_points) || !aligned16(d_commitments) || !aligned16(d_responses) || !aligned16(d_weights16) || !aligned16(d_out_points))
    return fail(ZKP_ERR_ARG, "device buffers must be 16-byte aligned");
  if (K == 1) return zkp_fused_batch_verify_dev(c, st, N_each, strobe_pos, d_transcripts, d_points, d_commitments, d_responses, d_weights16, d_out_points, d_status);
  fused_shape s0;
  int rc = check_fused_statement(st, s0);
  if (rc) return rc;
  uint64_t n_each = 0;
  rc = check_many(K, N_each, s0, &n_each);
  if (rc) return rc;
  fused_plan* pl = nullptr;
  rc = get_plan(c, FLOW_BATCH, st, K * N_each, strobe_pos, &pl);
  if (rc) return rc;
  const fused_shape& s = pl->s;
  if (!d_out_points || !d_status || !d_transcripts || (s.nc && (!d_commitments || !d_weights16)) || (s.m && !d_responses) || !d_points)
    return fail(ZKP_ERR_ARG, "NULL device pointer");
  const batch_inter o = batch_carve(*pl, 0, K);
  rc = ensure_ws(c, o.end + optional_many_ws(n_each, K));
  if (rc) return rc;
  return batch_core(c, *pl, o, d_transcripts, d_points, d_commitments, d_responses, d_weights16, d_out_points, d_status, /*throughput=*/!c->dev_latency, K, /*overlap=*/c->dev_latency);
}

// ---- verify_batchable, one verdict per proof -----------------------------------------------------------------------------
int zkp_fused_verify_batchable_dev(zkp_ctx* c, const zkp_fused_statement* st, uint32_t N, uint32_t strobe_pos, uint8_t* d_transcripts,
                                   const uint8_t* d_table, const uint8_t* d_responses, const uint8_t* d_weights16, uint8_t* d_results) {
  if (!c) return fail(ZKP_ERR_ARG, "ctx is NULL");
  if (N == 0) return ZKP_OK;
  HIP_TRY(hipSetDevice(c->device));
  fused_plan* pl = nullptr;
  int rc = get_plan(c, FLOW_BATCH, st, N, strobe_pos, &pl);
  if (rc) return rc;
  const fused_shape& s = pl->s;
  if (!d_transcripts || !d_results || (s.m && !d_responses) || (s.nc && !d_weights16) || ((s.np || s.nc) && !d_table)) return fail(ZKP_ERR_ARG, "NULL device pointer");
  if (!aligned16(d_transcripts) || !aligned16(d_table) || !aligned16(d_responses) || !aligned16(d_weights16))
    return fail(ZKP_ERR_ARG, "device buffers must be 16-byte aligned");
  const size_t n_points = (size_t)s.ns + (size_t)s.ni * N + (size_t)N * s.nc, K = (size_t)s.np + s.nc;
  if (n_points > 0x7fffffffull || (size_t)N * K > 0x7fffffffull) return fail(ZKP_ERR_ARG, "batch too large");
  const each_inter o = each_carve(*pl, 0);
  rc = ensure_ws(c, each_uses_straus(c, *pl) ? straus_carve(c, *pl, o.end).end : o.end + terms_path_ws((uint32_t)n_points, (uint32_t)(N * K), N, each_terms_cfg(*pl)));
  if (rc) return rc;
  return each_core(c, *pl, o, d_transcripts, d_table, d_responses, d_weights16, d_results, /*overlap=*/c->dev_overlap);
}

}  // extern "C"
