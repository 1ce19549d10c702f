// Comb tables for the points that do NOT have a fixed-base table (per-proof points such as P, Q of the CMZ statement).
//
// A generic term s*P costs 256 doublings + 128 additions in the radix-4 ladder (term_generic).  Splitting the scalar
// into TEETH chunks of 256/TEETH bits, s = sum_j 2^(BITS j) s_j, and tabulating  k * 2^(BITS j) * P  (j < TEETH,
// k = 1..8) once per DISTINCT point turns it into a TEETH-way interleaved radix-16 walk: BITS/4 windows x (4 doublings +
// TEETH additions) = BITS doublings + 64 additions per term.  Building the table costs 256 doublings + 7 TEETH point
// operations per point -- and in the reference's statements a per-proof point is typically shared by many terms (CMZ: P
// appears in 10 of the 11 constraints of a proof, benches/zkp.rs:34-43), so it is amortised.  Point operations per
// point with u terms:   256 + 7 TEETH + u (256 / TEETH + 64),   minimal at TEETH ~ sqrt(256 u / 7):
//     TEETH = 4  (64 doublings per term, 33-entry table, 4.75 KB)   for points with a few uses,
//     TEETH = 16 (16 doublings per term, 129-entry table, 18.6 KB)  from about 6 uses per point on (CMZ's P: 1564 -> 1168).
// The first window's four doublings act on the identity and are skipped (12 doublings per term at TEETH = 16).
// Each row of the TEETH = 16 table is read by exactly one window of a term, so the constant-time scans stream the table
// once per term instead of cycling 4 rows 16 times through a cache they do not fit (round 1: 556 MB of fabric traffic
// per launch for 9.6 MB of algorithmic bytes).
//
// Only points with at least two cold uses get a table; tables are addressed through a compact slot index (slot_of[]),
// so the workspace holds  #table points x entries  rather than  #points x entries.  Single-use points go to a signed
// radix-16 ladder over their own eight multiples (term_ladder16: 256 doublings + 65 additions, against 256 + 7 TEETH +
// BITS + 65 for table + walk), constant-time (masked scans) or variable-time.
//
// Layout per table: 8 TEETH + 1 entries of 144 B in the quad-cached order (Y-X, Y+X, 2Z, 2dT):
//     entry 8 j + (k-1) = k * 2^(BITS j) * P   (j < TEETH, k = 1..8),      entry 8 TEETH = 2^256 * P  (carry window)
#pragma once
#include "quad.h"

namespace zkp {

template <int TEETH>
struct comb_cfg {
  static_assert(TEETH == 4 || TEETH == 8 || TEETH == 16 || TEETH == 32, "teeth");
  static constexpr int BITS = 256 / TEETH;          // scalar bits per tooth
  static constexpr int WINDOWS = BITS / 4;          // radix-16 windows per tooth
  static constexpr int ENTRIES = 8 * TEETH + 1;
};
constexpr int comb_entries(int teeth) { return 8 * teeth + 1; }
constexpr int LADDER_ENTRIES = 8;                   // per ladder term: 1 P .. 8 P

__device__ __forceinline__ void q_store_cached(dev_ext* dst, const qcached& c, int q) {
  uint32_t* w = reinterpret_cast<uint32_t*>(dst) + 9 * q;
#pragma unroll
  for (int i = 0; i < 9; ++i) w[i] = c.c.v[i];
}

// slot_of[p] = compact table index of every point with >= comb_min cold uses (wave-aggregated counter: one device-scope
// atomic per wavefront); slot_pt[slot] = p.  The order of the slots is irrelevant to the results.
__global__ void __launch_bounds__(256)
k_comb_slots(uint32_t n_points, const uint32_t* __restrict__ uses, uint32_t comb_min, uint32_t max_tables,
             uint32_t* __restrict__ counter, uint32_t* __restrict__ slot_of, uint32_t* __restrict__ slot_pt, dev_affine* __restrict__ pts) {
  const uint32_t pi = blockIdx.x * blockDim.x + threadIdx.x;
  const bool want = pi < n_points && uses[pi] >= comb_min;
  const uint64_t mask = __ballot(want);
  if (!mask) return;
  const uint32_t lane = threadIdx.x & 63u;
  const int leader = __ffsll((long long)mask) - 1;
  uint32_t base = 0;
  if ((int)lane == leader) base = atomicAdd(counter, (uint32_t)__popcll(mask));
  base = (uint32_t)__shfl((int)base, leader);
  if (want) {
    const uint32_t slot = base + (uint32_t)__popcll(mask & ((1ull << lane) - 1ull));
    // max_tables is an upper bound by construction (derived from the statement, or from the call's own index arrays); a
    // slot beyond it would be a bug in that bound.  Fail closed: the point keeps no table, its terms are skipped, and it is
    // marked undecodable so that every MSM that uses it reports status 1 instead of a wrong point.
    slot_of[pi] = slot < max_tables ? slot : 0xffffffffu;
    if (slot < max_tables) slot_pt[slot] = pi;
    else pts[pi].valid = 0;
  }
}

template <int TEETH>
__global__ void __launch_bounds__(256, 2)
k_comb_tables(const uint32_t* __restrict__ n_slots, uint32_t max_tables, const uint32_t* __restrict__ slot_pt,
              const dev_affine* __restrict__ pts, dev_ext* __restrict__ comb) {
  using cfg = comb_cfg<TEETH>;
  const uint32_t gt = blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t slot = gt >> 2;
  const int q = (int)(gt & 3u);
  const uint32_t ns = min(*n_slots, max_tables);
  if (slot >= ns) return;                                 // uniform within the quad
  const uint32_t pi = slot_pt[slot];
  qpt base;
  {
    const uint32_t* w = reinterpret_cast<const uint32_t*>(pts + pi);      // x[9] y[9] t[9] valid
    fe one;
    fe_1(one);
#pragma unroll
    for (int i = 0; i < 9; ++i) base.c.v[i] = q == 0 ? w[i] : (q == 1 ? w[9 + i] : (q == 3 ? w[18 + i] : one.v[i]));
  }
  dev_ext* tbl = comb + (size_t)slot * cfg::ENTRIES;
#pragma unroll 1
  for (int j = 0; j < TEETH; ++j) {
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
    for (int d = 0; d < cfg::BITS - 3; ++d) q_double(base, base, q);       // 8 * 2^(BITS-3) = 2^BITS
  }
  qcached c;
  q_to_cached(c, base, q);                                                 // 2^256 * P
  q_store_cached(tbl + 8 * TEETH, c, q);
}

// The same table built by ONE lane per point: about half the instructions of the quad version (no DPP exchanges, no
// replicated additions) at four times its latency.  Used by the asynchronous (_dev) entry points, whose callers keep many
// batches in flight: there the chip is VALU-bound and the chain's latency is hidden by the other streams.
__device__ __forceinline__ void store_comb_entry(dev_ext* dst, const ge_cached& c) {
  uint32_t w[36];
  fe_get(w, c.YmX); fe_get(w + 9, c.YpX); fe_get(w + 18, c.Z2); fe_get(w + 27, c.T2d);
  store_vec<9>(dst, w);
}
template <int TEETH>
__global__ void __launch_bounds__(256, 2)
k_comb_tables_lane(const uint32_t* __restrict__ n_slots, uint32_t max_tables, const uint32_t* __restrict__ slot_pt,
                   const dev_affine* __restrict__ pts, dev_ext* __restrict__ comb) {
  using cfg = comb_cfg<TEETH>;
  const uint32_t slot = blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t ns = min(*n_slots, max_tables);
  if (slot >= ns) return;
  ge_p3 base;
  load_affine(base, pts + slot_pt[slot]);
  dev_ext* tbl = comb + (size_t)slot * cfg::ENTRIES;
#pragma unroll 1
  for (int j = 0; j < TEETH; ++j) {
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
    for (int d = 0; d < cfg::BITS - 4; ++d) ge_double<false>(base, base);
    ge_double<true>(base, base);                                           // 8 * 2^(BITS-3) = 2^BITS
  }
  ge_cached c;
  ge_to_cached(c, base);                                                   // 2^256 * P
  store_comb_entry(tbl + 8 * TEETH, c);
}

__device__ __forceinline__ void load_comb_entry(ge_cached& c, const dev_ext* src) {
  uint32_t w[36];
  load_vec<9>(w, src);
  fe_set(c.YmX, w); fe_set(c.YpX, w + 9); fe_set(c.Z2, w + 18); fe_set(c.T2d, w + 27);
}

// sel = (mag ? row[mag - 1] : identity) for mag in 0..8.  CT: all 8 entries of the row are read and the entry is picked
// with masks, two entries (18 independent 16-byte loads) in flight at a time.
template <bool CT>
__device__ __forceinline__ void comb_select(ge_cached& sel, const dev_ext* __restrict__ row, uint32_t mag) {
  ge_cached_identity(sel);
  if (CT) {
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
}

// partial[t] = scalars[t] * P through P's comb table.  CT: no branch or address depends on the scalar.
template <bool CT, int TEETH>
__device__ __forceinline__ void term_comb(uint32_t t, const uint8_t* __restrict__ scalars, const dev_ext* __restrict__ tbl,
                                          dev_ext* __restrict__ partial, uint32_t* ecol) {
  using cfg = comb_cfg<TEETH>;
  uint32_t s[8], e[8], top;
  load_vec<2>(s, scalars + 32 * (size_t)t);
  sc_add_pattern(e, top, s, 0x88888888u);                       // signed radix-16 digits: nibble - 8 in [-8, 7]
  // the recoded scalar goes to this lane's LDS column (word j at ecol[256 j]): the walk picks nibbles in table order, i.e. by
  // a run-time word index (the compiler would do the same with a promoted private array, in LDS of its own on top of the
  // fixed-base rows')
#pragma unroll
  for (int j = 0; j < 8; ++j) ecol[256 * j] = e[j];
  ge_p3 acc;
  ge_identity(acc);
  if (CT) {
    // Constant-time walk, software-pipelined: the first two entries of the NEXT row are requested before the addition of the
    // current one, so that a quarter of every row's 72 loads is in flight during ~1,100 instructions of arithmetic instead
    // of being waited for (two wavefronts per SIMD do not hide the scans' latency on their own: 30 % of this kernel's
    // wavefront-cycles were waits).
    ge_cached n0, n1;
    load_comb_entry(n0, tbl + 0);
    load_comb_entry(n1, tbl + 1);
#pragma unroll 1
    for (int w = cfg::WINDOWS - 1; w >= 0; --w) {
      if (w != cfg::WINDOWS - 1) {                              // (the accumulator is still the identity in the first window)
        ge_double4(acc);
      }
#pragma unroll 1
      for (int j = 0; j < TEETH; ++j) {
        const int nidx = j * cfg::WINDOWS + w;                  // nibble number of tooth j, window w
        const uint32_t nib = (ecol[256 * (nidx >> 3)] >> (4 * (nidx & 7))) & 15u;
        const uint32_t neg = (uint32_t)(nib < 8u);
        const uint32_t mag = neg ? 8u - nib : nib - 8u;         // 0..8
        const dev_ext* row = tbl + 8 * j;
        ge_cached sel;
        ge_cached_identity(sel);
        ge_cached_cmov(sel, n0, (uint32_t)(mag == 1));
        ge_cached_cmov(sel, n1, (uint32_t)(mag == 2));
#pragma unroll 1
        for (uint32_t h = 1; h < 4; ++h) {
          ge_cached c0, c1;
          load_comb_entry(c0, row + 2 * h + 0);
          load_comb_entry(c1, row + 2 * h + 1);
          ge_cached_cmov(sel, c0, (uint32_t)(mag == 2 * h + 1));
          ge_cached_cmov(sel, c1, (uint32_t)(mag == 2 * h + 2));
        }
        const dev_ext* next = (j + 1 < TEETH) ? row + 8 : tbl;  // (after the last row of the last window: a harmless re-read of row 0)
        load_comb_entry(n0, next + 0);
        load_comb_entry(n1, next + 1);
        ge_cached_cneg(sel, neg);
        ge_add_cached(acc, acc, sel);
      }
    }
  } else {
#pragma unroll 1
    for (int w = cfg::WINDOWS - 1; w >= 0; --w) {
      if (w != cfg::WINDOWS - 1) {
        ge_double4(acc);
      }
#pragma unroll 1
      for (int j = 0; j < TEETH; ++j) {
        const int nidx = j * cfg::WINDOWS + w;
        const uint32_t nib = (ecol[256 * (nidx >> 3)] >> (4 * (nidx & 7))) & 15u;
        const uint32_t neg = (uint32_t)(nib < 8u);
        const uint32_t mag = neg ? 8u - nib : nib - 8u;
        ge_cached sel;
        comb_select<false>(sel, tbl + 8 * j, mag);
        ge_cached_cneg(sel, neg);
        ge_add_cached(acc, acc, sel);
      }
    }
  }
  {
    ge_cached sel, c;
    ge_cached_identity(sel);
    load_comb_entry(c, tbl + 8 * TEETH);
    ge_cached_cmov(sel, c, top);
    ge_add_cached(acc, acc, sel);
  }
  store_ext(partial + t, acc);
}

// partial[t] = scalars[t] * P for a point that no other cold term of the call uses (a constraint's left-hand side in
// verify_compact; CMZ's Q in the prover): signed radix-16 ladder over P's own eight multiples, 256 doublings + 65
// additions instead of the radix-4 ladder's 256 + 128.  The multiples live in the term's slot of the ladder scratch.
// CT: the eight multiples are scanned with masks (prover.rs:94 semantics); otherwise the entry is loaded directly.
template <bool CT>
__device__ __forceinline__ void term_ladder16(uint32_t t, const uint8_t* __restrict__ scalars, const dev_affine* __restrict__ pt,
                                              dev_ext* __restrict__ tbl, dev_ext* __restrict__ partial, uint32_t* ecol) {
  uint32_t s[8], e[8], top;
  load_vec<2>(s, scalars + 32 * (size_t)t);
  sc_add_pattern(e, top, s, 0x88888888u);                       // signed radix-16 digits: nibble - 8 in [-8, 7]
#pragma unroll
  for (int j = 0; j < 8; ++j) ecol[256 * j] = e[j];             // this lane's LDS column (see term_comb)
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
  for (int j = 7; j >= 0; --j) {
    uint32_t cur = ecol[256 * j];
#pragma unroll 1
    for (int k = 0; k < 8; ++k) {
      ge_double4(acc);
      const uint32_t nib = cur >> 28;
      cur <<= 4;
      const uint32_t neg = (uint32_t)(nib < 8u);
      const uint32_t mag = neg ? 8u - nib : nib - 8u;           // 0..8
      ge_cached sel;
      comb_select<CT>(sel, tbl, mag);
      ge_cached_cneg(sel, neg);
      ge_add_cached(acc, acc, sel);
    }
  }
  store_ext(partial + t, acc);
}

}  // namespace zkp
