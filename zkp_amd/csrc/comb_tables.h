// Comb tables for the points that do NOT have a fixed-base table (per-proof points such as P, Q of the CMZ statement).
//
// A generic term s*P costs 256 doublings + 128 additions in the radix-4 ladder (term_generic).  Splitting the scalar
// into four 64-bit chunks, s = s0 + 2^64 s1 + 2^128 s2 + 2^192 s3, and tabulating  k * 2^(64 j) * P  (j < 4, k = 1..8)
// once per DISTINCT point turns it into a 4-way interleaved radix-16 walk: 16 windows x (4 doublings + 4 additions)
// = 64 doublings + 64 additions per term.  Building the table costs ~256 doublings per point, i.e. what ONE term used
// to cost -- and in the reference's statements a per-proof point is typically shared by many terms (CMZ: P appears in
// 10 of the 11 constraints of a proof, benches/zkp.rs:34-43), so the doublings are amortised.  Even a point used once
// breaks even (more doublings up front, half the additions).
//
// The table is built by a QUAD of lanes per point (quad.h), because it is a 260-long dependent chain.
// Layout per point: 33 entries of 144 B in the quad-cached order (Y-X, Y+X, 2Z, 2dT):
//     entry 8 j + (k-1) = k * 2^(64 j) * P   (j = 0..3, k = 1..8),      entry 32 = 2^256 * P  (carry window)
#pragma once
#include "quad.h"

namespace zkp {

constexpr int COMB_ENTRIES = 33;

__device__ __forceinline__ void q_store_cached(dev_ext* dst, const qcached& c, int q) {
  uint32_t* w = reinterpret_cast<uint32_t*>(dst) + 9 * q;
#pragma unroll
  for (int i = 0; i < 9; ++i) w[i] = c.c.v[i];
}

__global__ void __launch_bounds__(256, 2)
k_comb_tables(uint32_t n_points, const uint32_t* __restrict__ uses, uint32_t comb_min, const dev_affine* __restrict__ pts,
              dev_ext* __restrict__ comb) {
  const uint32_t gt = blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t pi = gt >> 2;
  const int q = (int)(gt & 3u);
  if (pi >= n_points || uses[pi] < comb_min) return;    // uniform within the quad
  qpt base;
  {
    const uint32_t* w = reinterpret_cast<const uint32_t*>(pts + pi);      // x[9] y[9] t[9] valid
    fe one;
    fe_1(one);
#pragma unroll
    for (int i = 0; i < 9; ++i) base.c.v[i] = q == 0 ? w[i] : (q == 1 ? w[9 + i] : (q == 3 ? w[18 + i] : one.v[i]));
  }
  dev_ext* tbl = comb + (size_t)pi * COMB_ENTRIES;
#pragma unroll 1
  for (int j = 0; j < 4; ++j) {
    qpt m2, m3, m4, m;
    qcached c1, c;
    q_to_cached(c1, base, q);
    q_store_cached(tbl + 8 * j + 0, c1, q);
    q_double(m2, base, q);
    q_to_cached(c, m2, q); q_store_cached(tbl + 8 * j + 1, c, q);
    q_add_cached(m3, m2, c1, q);
    q_to_cached(c, m3, q); q_store_cached(tbl + 8 * j + 2, c, q);
    q_double(m4, m2, q);
    q_to_cached(c, m4, q); q_store_cached(tbl + 8 * j + 3, c, q);
    q_add_cached(m, m4, c1, q);                                            // 5
    q_to_cached(c, m, q); q_store_cached(tbl + 8 * j + 4, c, q);
    q_double(m, m3, q);                                                    // 6
    q_to_cached(c, m, q); q_store_cached(tbl + 8 * j + 5, c, q);
    q_add_cached(m, m, c1, q);                                             // 7
    q_to_cached(c, m, q); q_store_cached(tbl + 8 * j + 6, c, q);
    q_double(base, m4, q);                                                 // 8
    q_to_cached(c, base, q); q_store_cached(tbl + 8 * j + 7, c, q);
#pragma unroll 1
    for (int d = 0; d < 61; ++d) q_double(base, base, q);                  // 8 * 2^61 = 2^64
  }
  qcached c;
  q_to_cached(c, base, q);                                                 // 2^256 * P
  q_store_cached(tbl + 32, c, q);
}

// The same table built by ONE lane per point: about half the instructions of the quad version (no DPP exchanges, no
// replicated additions) at four times its latency.  Used by the asynchronous (_dev) entry points, whose callers keep many
// batches in flight: there the chip is VALU-bound and the chain's latency is hidden by the other streams.
__device__ __forceinline__ void store_comb_entry(dev_ext* dst, const ge_cached& c) {
  uint32_t w[36];
  fe_get(w, c.YmX); fe_get(w + 9, c.YpX); fe_get(w + 18, c.Z2); fe_get(w + 27, c.T2d);
  store_vec<9>(dst, w);
}
__global__ void __launch_bounds__(256, 2)
k_comb_tables_lane(uint32_t n_points, const uint32_t* __restrict__ uses, uint32_t comb_min, const dev_affine* __restrict__ pts,
                   dev_ext* __restrict__ comb) {
  const uint32_t pi = blockIdx.x * blockDim.x + threadIdx.x;
  if (pi >= n_points || uses[pi] < comb_min) return;
  ge_p3 base;
  load_affine(base, pts + pi);
  dev_ext* tbl = comb + (size_t)pi * COMB_ENTRIES;
#pragma unroll 1
  for (int j = 0; j < 4; ++j) {
    ge_p3 m2, m3, m4, m;
    ge_cached c1, c;
    ge_to_cached(c1, base);
    store_comb_entry(tbl + 8 * j + 0, c1);
    ge_double<true>(m2, base);
    ge_to_cached(c, m2); store_comb_entry(tbl + 8 * j + 1, c);
    ge_add_cached(m3, m2, c1);
    ge_to_cached(c, m3); store_comb_entry(tbl + 8 * j + 2, c);
    ge_double<true>(m4, m2);
    ge_to_cached(c, m4); store_comb_entry(tbl + 8 * j + 3, c);
    ge_add_cached(m, m4, c1);                                              // 5
    ge_to_cached(c, m); store_comb_entry(tbl + 8 * j + 4, c);
    ge_double<true>(m, m3);                                                // 6
    ge_to_cached(c, m); store_comb_entry(tbl + 8 * j + 5, c);
    ge_add_cached(m, m, c1);                                               // 7
    ge_to_cached(c, m); store_comb_entry(tbl + 8 * j + 6, c);
    ge_double<true>(base, m4);                                             // 8
    ge_to_cached(c, base); store_comb_entry(tbl + 8 * j + 7, c);
#pragma unroll 1
    for (int d = 0; d < 60; ++d) ge_double<false>(base, base);
    ge_double<true>(base, base);                                           // 8 * 2^61 = 2^64
  }
  ge_cached c;
  ge_to_cached(c, base);                                                   // 2^256 * P
  store_comb_entry(tbl + 32, c);
}

__device__ __forceinline__ void load_comb_entry(ge_cached& c, const dev_ext* src) {
  uint32_t w[36];
  load_vec<9>(w, src);
  fe_set(c.YmX, w); fe_set(c.YpX, w + 9); fe_set(c.Z2, w + 18); fe_set(c.T2d, w + 27);
}

// partial[t] = scalars[t] * P through P's comb table.  CT: every window reads all 8 entries of each chunk row and
// picks with masks; the instruction stream and the addresses do not depend on the scalar.
template <bool CT>
__device__ __forceinline__ void term_comb(uint32_t t, const uint8_t* __restrict__ scalars, const dev_ext* __restrict__ tbl,
                                          dev_ext* __restrict__ partial) {
  uint32_t s[8], e[8], top;
  load_vec<2>(s, scalars + 32 * (size_t)t);
  sc_add_pattern(e, top, s, 0x88888888u);                       // signed radix-16 digits: nibble - 8 in [-8, 7]
  ge_p3 acc;
  ge_identity(acc);
#pragma unroll 1
  for (int w = 15; w >= 0; --w) {
    ge_double<false>(acc, acc);
    ge_double<false>(acc, acc);
    ge_double<false>(acc, acc);
    ge_double<true>(acc, acc);
#pragma unroll 1
    for (int j = 0; j < 4; ++j) {
      const uint32_t nib = (sel8(e, 2 * j + (w >> 3)) >> (4 * (w & 7))) & 15u;
      const uint32_t neg = (uint32_t)(nib < 8u);
      const uint32_t mag = neg ? 8u - nib : nib - 8u;           // 0..8
      const dev_ext* row = tbl + 8 * j;
      ge_cached sel;
      ge_cached_identity(sel);
      if (CT) {
        // masked scan of the 8-entry row, two entries (18 independent 16-byte loads) in flight at a time
#pragma unroll 1
        for (uint32_t h = 0; h < 4; ++h) {
          ge_cached c0, c1;
          load_comb_entry(c0, row + 2 * h + 0);
          load_comb_entry(c1, row + 2 * h + 1);
          ge_cached_cmov(sel, c0, (uint32_t)(mag == 2 * h + 1));
          ge_cached_cmov(sel, c1, (uint32_t)(mag == 2 * h + 2));
        }
      } else if (mag) {
        load_comb_entry(sel, row + (mag - 1));
      }
      ge_cached_cneg(sel, neg);
      ge_add_cached(acc, acc, sel);
    }
  }
  {
    ge_cached sel, c;
    ge_cached_identity(sel);
    load_comb_entry(c, tbl + 32);
    ge_cached_cmov(sel, c, top);
    ge_add_cached(acc, acc, sel);
  }
  store_ext(partial + t, acc);
}

// partial[t] = scalars[t] * P for a point that no other term of a variable-time call uses (a constraint's left-hand side
// in verify_compact): signed radix-16 ladder over P's own eight multiples, 256 doublings + 65 additions instead of the
// radix-4 ladder's 256 + 128.  The multiples live in the point's (otherwise unused) comb-table slot.  Addresses depend on
// the scalar: variable-time callers only.
__device__ __forceinline__ void term_ladder16(uint32_t t, const uint8_t* __restrict__ scalars, const dev_affine* __restrict__ pt,
                                              dev_ext* __restrict__ tbl, dev_ext* __restrict__ partial) {
  uint32_t s[8], e[8], top;
  load_vec<2>(s, scalars + 32 * (size_t)t);
  sc_add_pattern(e, top, s, 0x88888888u);                       // signed radix-16 digits: nibble - 8 in [-8, 7]
  ge_p3 acc;
  {
    ge_p3 P, m2, m3, m4, m;
    ge_cached c1, c;
    load_affine(P, pt);
    ge_to_cached(c1, P);
    store_comb_entry(tbl + 0, c1);
    ge_double<true>(m2, P);
    ge_to_cached(c, m2); store_comb_entry(tbl + 1, c);
    ge_add_cached(m3, m2, c1);
    ge_to_cached(c, m3); store_comb_entry(tbl + 2, c);
    ge_double<true>(m4, m2);
    ge_to_cached(c, m4); store_comb_entry(tbl + 3, c);
    ge_add_cached(m, m4, c1);
    ge_to_cached(c, m); store_comb_entry(tbl + 4, c);
    ge_double<true>(m, m3);
    ge_to_cached(c, m); store_comb_entry(tbl + 5, c);
    ge_add_cached(m, m, c1);
    ge_to_cached(c, m); store_comb_entry(tbl + 6, c);
    ge_double<true>(m, m4);
    ge_to_cached(c, m); store_comb_entry(tbl + 7, c);
    ge_cached sel;                                              // carry out of bit 255: one more P at the top
    ge_cached_identity(sel);
    ge_cached_cmov(sel, c1, top);
    ge_identity(acc);
    ge_add_cached(acc, acc, sel);
  }
#pragma unroll 1
  for (int w = 63; w >= 0; --w) {
    ge_double<false>(acc, acc);
    ge_double<false>(acc, acc);
    ge_double<false>(acc, acc);
    ge_double<true>(acc, acc);
    const uint32_t nib = (sel8(e, w >> 3) >> (4 * (w & 7))) & 15u;
    const uint32_t neg = (uint32_t)(nib < 8u);
    const uint32_t mag = neg ? 8u - nib : nib - 8u;             // 0..8
    ge_cached sel;
    ge_cached_identity(sel);
    if (mag) load_comb_entry(sel, tbl + (mag - 1));
    ge_cached_cneg(sel, neg);
    ge_add_cached(acc, acc, sel);
  }
  store_ext(partial + t, acc);
}

}  // namespace zkp
