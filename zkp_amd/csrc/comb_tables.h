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
// (The grouped walk through LDS, comb_group_block below, runs two accumulators instead: 16 doublings + 1 addition per term, each staged row serving two additions.)
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
#include "stmt_pairs.h"

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
// Points with >= group_min uses also reserve uses[p] consecutive places of the CLASS_GROUP list: group_start[p] (a wavefront
// scan of the use counts + one atomic per wavefront on counter[1]).
__global__ void __launch_bounds__(256)
k_comb_slots(uint32_t n_points, const uint32_t* __restrict__ uses, uint32_t comb_min, uint32_t group_min, uint32_t max_tables,
             uint32_t* __restrict__ counter, uint32_t* __restrict__ slot_of, uint32_t* __restrict__ slot_pt, uint32_t* __restrict__ group_start,
             dev_affine* __restrict__ pts) {
  const uint32_t pi = blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t u = pi < n_points ? uses[pi] : 0u;
  const bool want = pi < n_points && u >= comb_min;
  const uint64_t mask = __ballot(want);
  if (!mask) return;
  const uint32_t lane = threadIdx.x & 63u;
  const int leader = __ffsll((long long)mask) - 1;
  const uint32_t gu = (want && u >= group_min) ? u : 0u;
  uint32_t incl = gu;                                          // inclusive scan of the grouped use counts over the wavefront
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const uint32_t up = (uint32_t)__shfl_up((int)incl, d);
    if ((int)lane >= d) incl += up;
  }
  const uint32_t total = (uint32_t)__shfl((int)incl, 63);
  uint32_t base = 0, gbase = 0;
  if ((int)lane == leader) {
    base = atomicAdd(counter, (uint32_t)__popcll(mask));
    if (total) gbase = atomicAdd(counter + 1, total);
  }
  base = (uint32_t)__shfl((int)base, leader);
  gbase = (uint32_t)__shfl((int)gbase, leader);
  if (want) {
    const uint32_t slot = base + (uint32_t)__popcll(mask & ((1ull << lane) - 1ull));
    // max_tables is an upper bound by construction (derived from the statement, or from the call's own index arrays); a
    // slot beyond it would be a bug in that bound.  Fail closed: the point keeps no table, its terms are skipped, and it is
    // marked undecodable so that every MSM that uses it reports status 1 instead of a wrong point.
    slot_of[pi] = slot < max_tables ? slot : 0xffffffffu;
    if (slot < max_tables) slot_pt[slot] = pi;
    else pts[pi].valid = 0;
    if (gu) group_start[pi] = gbase + incl - gu;
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
  if (pi & STMT_ABSORBED) return;                         // (a table of multiples, not a comb: k_rider_tables)
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
// One entry of 64 lanes' tables through LDS (k_rider_tables): a lane's entry is 9 chunks of 16 bytes, and 64 lanes storing chunk q of 64 different
// tables touch 64 cache lines per instruction.  Transposed, an instruction stores the 9 chunks of 7 tables' entries: 7 runs of 144 bytes.  (The comb builder
// k_comb_tables_lane gains nothing from it -- 3.10 against 3.12 ms for 204,800 tables: its 368 dependent point operations are the time, not its stores.)
// stage = [36][64] words, live = [64] flags of this wavefront; every lane of the wavefront calls it.
__device__ __forceinline__ void store_entries_staged(uint32_t (*stage)[64], const uint32_t* live, uint32_t lane, dev_ext* __restrict__ tbl0 /* table of lane 0 */,
                                                     uint32_t entries_per_table, uint32_t e, const ge_cached& c) {
  uint32_t w[36];
  fe_get(w, c.YmX); fe_get(w + 9, c.YpX); fe_get(w + 18, c.Z2); fe_get(w + 27, c.T2d);
#pragma unroll
  for (int i = 0; i < 36; ++i) stage[i][lane] = w[i];
  __builtin_amdgcn_wave_barrier();
#pragma unroll
  for (int i = 0; i < 9; ++i) {
    const uint32_t ch = (uint32_t)i * 64u + lane, t = ch / 9u, q = ch - 9u * t;      // chunk q of the table of lane t
    if (live[t]) {
      const uint4 v = make_uint4(stage[4 * q][t], stage[4 * q + 1][t], stage[4 * q + 2][t], stage[4 * q + 3][t]);
      reinterpret_cast<uint4*>(tbl0 + (size_t)t * entries_per_table + e)[q] = v;
    }
  }
  __builtin_amdgcn_wave_barrier();
}
template <int TEETH>
__device__ __forceinline__ void comb_table_lane(uint32_t slot, const uint32_t* __restrict__ n_slots, uint32_t max_tables, const uint32_t* __restrict__ slot_pt,
                                                const dev_affine* __restrict__ pts, dev_ext* __restrict__ comb) {
  using cfg = comb_cfg<TEETH>;
  const uint32_t ns = min(*n_slots, max_tables);
  if (slot >= ns) return;
  if (slot_pt[slot] & STMT_ABSORBED) return;              // (a table of multiples, not a comb: k_rider_tables)
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
template <int TEETH>
__global__ void __launch_bounds__(256, 2)
k_comb_tables_lane(const uint32_t* __restrict__ n_slots, uint32_t max_tables, const uint32_t* __restrict__ slot_pt,
                   const dev_affine* __restrict__ pts, dev_ext* __restrict__ comb) {
  comb_table_lane<TEETH>(blockIdx.x * blockDim.x + threadIdx.x, n_slots, max_tables, slot_pt, pts, comb);
}


// The one-lane builder as a PRODUCER wavefront and a CONSUMER wavefront (round 5): a lone wavefront already issues a point operation's ~1,000 dependent
// instructions at 4.6 cycles each, so a table's 368 sequential point operations ARE the 0.9 ms of the launch -- no schedule inside the lane shortens them.  Two
// wavefronts on two SIMDs do: wavefront 0 runs the 256-doubling chain of 64 tables and hands the base of every tooth over through LDS, wavefront 1 computes
// the eight multiples of that tooth (7 operations + 8 conversions) while the producer doubles on: same instructions in total, chain 256 instead of 368
// operations long.  One barrier per tooth; the hand-over is double buffered (the producer writes buffer j & 1 after barrier j - 1, which the consumer only
// passes once it has read buffer (j - 2) & 1).  `hand` = LDS [2][36][64] words.
template <int TEETH>
__device__ __forceinline__ void comb_table_pc(uint32_t slot0, const uint32_t* __restrict__ n_slots, uint32_t max_tables, const uint32_t* __restrict__ slot_pt,
                                              const dev_affine* __restrict__ pts, dev_ext* __restrict__ comb, uint32_t* hand) {
  using cfg = comb_cfg<TEETH>;
  const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
  const uint32_t ns = min(*n_slots, max_tables);
  const uint32_t slot = slot0 + lane;
  const bool have = slot < ns && !(slot_pt[slot < ns ? slot : 0u] & STMT_ABSORBED);     // (not the tables of multiples: k_rider_tables)
  dev_ext* tbl = comb + (size_t)(have ? slot : 0u) * cfg::ENTRIES;
  if (wave == 0) {
    ge_p3 base;
    ge_identity(base);
    if (have) load_affine(base, pts + slot_pt[slot]);
#pragma unroll 1
    for (int j = 0; j < TEETH; ++j) {
      uint32_t* h = hand + (j & 1) * 36 * 64 + lane;
#pragma unroll
      for (int i = 0; i < 9; ++i) { h[64 * i] = base.X.v[i]; h[64 * (9 + i)] = base.Y.v[i]; h[64 * (18 + i)] = base.Z.v[i]; h[64 * (27 + i)] = base.T.v[i]; }
      __syncthreads();                                                       // tooth j's base is there (and the consumer is done with tooth j - 1)
#pragma unroll 1
      for (int d = 0; d < cfg::BITS - 1; ++d) ge_double<false>(base, base);
      ge_double<true>(base, base);                                           // 2^BITS x base
    }
    if (have) {
      ge_cached c;
      ge_to_cached(c, base);                                                 // 2^256 * P
      store_comb_entry(tbl + 8 * TEETH, c);
    }
  } else {
#pragma unroll 1
    for (int j = 0; j < TEETH; ++j) {
      __syncthreads();
      const uint32_t* h = hand + (j & 1) * 36 * 64 + lane;
      ge_p3 base, m2, m3, m4, m;
#pragma unroll
      for (int i = 0; i < 9; ++i) { base.X.v[i] = h[64 * i]; base.Y.v[i] = h[64 * (9 + i)]; base.Z.v[i] = h[64 * (18 + i)]; base.T.v[i] = h[64 * (27 + i)]; }
      FE_TRACK(fe_set_ub_tight(base.X); fe_set_ub_tight(base.Y); fe_set_ub_tight(base.Z); fe_set_ub_tight(base.T));
      ge_cached c1, c;
      ge_to_cached(c1, base);
      ge_double<true>(m2, base);
      ge_add_cached(m3, m2, c1);
      ge_double<true>(m4, m2);
      if (have) {
        store_comb_entry(tbl + 8 * j + 0, c1);
        ge_to_cached(c, m2); store_comb_entry(tbl + 8 * j + 1, c);
        ge_to_cached(c, m3); store_comb_entry(tbl + 8 * j + 2, c);
        ge_to_cached(c, m4); store_comb_entry(tbl + 8 * j + 3, c);
      }
      ge_add_cached(m, m4, c1);                                              // 5
      if (have) { ge_to_cached(c, m); store_comb_entry(tbl + 8 * j + 4, c); }
      ge_double<true>(m, m3);                                                // 6
      if (have) { ge_to_cached(c, m); store_comb_entry(tbl + 8 * j + 5, c); }
      ge_add_cached(m, m, c1);                                               // 7
      if (have) { ge_to_cached(c, m); store_comb_entry(tbl + 8 * j + 6, c); }
      ge_double<true>(m, m4);                                                // 8
      if (have) { ge_to_cached(c, m); store_comb_entry(tbl + 8 * j + 7, c); }
    }
  }
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
        ge_double4_flat(acc);
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

// The constant-time comb scan of ONE term spread over a QUAD of lanes, lane q walking window q (round 6; narrow calls on the latency schedule).  A point with
// fewer than GROUP_MIN_USES terms cannot join the grouped walk (its crossbar needs <= 4 tables per half wavefront), so each of its terms scans its own rows:
// 64 scanned additions + 12 doublings in one lane, 745 k cycles -- the pole of the term kernel of a 4096-proof call, whose grouped and fixed-base blocks take
// 485 k and 470 k (profiles/r06_constant_time_wave_cycles.txt).  Here lane q adds the 16 teeth of window q (16 scanned additions, no doubling in between), then
//     lanes 3, 1:  x 16          lane 2 += lane 3's, lane 0 += lane 1's          lane 2: x 256          lane 0 += lane 2's
// = 16 scanned additions + 12 doublings + 2 additions per lane: a third of the chain for 1.45 x the instructions.  Same table, same digits, same entries scanned
// completely (no address or branch depends on the scalar); the sum is the same group element, hence the same bytes.
__device__ __forceinline__ void ge_p3_dpp_from(ge_p3& r, const ge_p3& a, int ctrl_sel) {     // ctrl_sel: 0 = lanes take their odd neighbour's (1,1,3,3), 1 = lane 2's (2,2,2,2)
  if (ctrl_sel == 0) { fe_dpp<ZKP_QP(1, 1, 3, 3)>(r.X, a.X); fe_dpp<ZKP_QP(1, 1, 3, 3)>(r.Y, a.Y); fe_dpp<ZKP_QP(1, 1, 3, 3)>(r.Z, a.Z); fe_dpp<ZKP_QP(1, 1, 3, 3)>(r.T, a.T); }
  else { fe_dpp<ZKP_QP(2, 2, 2, 2)>(r.X, a.X); fe_dpp<ZKP_QP(2, 2, 2, 2)>(r.Y, a.Y); fe_dpp<ZKP_QP(2, 2, 2, 2)>(r.Z, a.Z); fe_dpp<ZKP_QP(2, 2, 2, 2)>(r.T, a.T); }
}
template <int TEETH>
__device__ __forceinline__ void term_comb_split4(uint32_t t, uint32_t q, const uint8_t* __restrict__ scalars, const dev_ext* __restrict__ comb, uint32_t slot,
                                                 dev_ext* __restrict__ partial, uint32_t* ecol) {
  using cfg = comb_cfg<TEETH>;
  static_assert(cfg::WINDOWS == 4, "one lane of the quad per window");
  // (the term and slot numbers wait in the lane's LDS column for the last step instead of in registers: the additions below need all of them)
  typedef __attribute__((address_space(3))) volatile uint32_t lds_word;
  lds_word* stash = (lds_word*)(ecol + 256 * 8);
  stash[0] = slot;
  stash[256] = t;
  const dev_ext* __restrict__ tbl = comb + (size_t)slot * cfg::ENTRIES;
  uint32_t s[8], e[8], top;
  load_vec<2>(s, scalars + 32 * (size_t)t);
  sc_add_pattern(e, top, s, 0x88888888u);
#pragma unroll
  for (int j = 0; j < 8; ++j) ecol[256 * j] = e[j];
  stash[512] = top;
  ge_p3 acc;
  ge_identity(acc);
  {
    ge_cached n0, n1;
    load_comb_entry(n0, tbl + 0);
    load_comb_entry(n1, tbl + 1);
#pragma unroll 1
    for (int j = 0; j < TEETH; ++j) {
      const uint32_t nidx = (uint32_t)j * cfg::WINDOWS + q;      // nibble number of tooth j, this lane's window
      const uint32_t nib = (ecol[256 * (nidx >> 3)] >> (4 * (nidx & 7))) & 15u;
      const uint32_t neg = (uint32_t)(nib < 8u);
      const uint32_t mag = neg ? 8u - nib : nib - 8u;
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
      const dev_ext* next = (j + 1 < TEETH) ? row + 8 : tbl;
      load_comb_entry(n0, next + 0);
      load_comb_entry(n1, next + 1);
      ge_cached_cneg(sel, neg);
      ge_add_cached(acc, acc, sel);
    }
  }
  // the quad's Horner: every lane runs every step (DPP sources must be active lanes); a lane keeps a step's result only where the schedule above says so
  {
    ge_p3 d = acc;
    ge_double4(d);                                                // 16 x (window 3's, window 1's sum)
    fe_pick(acc.X, d.X, (q & 1u) != 0); fe_pick(acc.Y, d.Y, (q & 1u) != 0); fe_pick(acc.Z, d.Z, (q & 1u) != 0); fe_pick(acc.T, d.T, (q & 1u) != 0);
    ge_p3 o;
    ge_p3_dpp_from(o, acc, 0);                                    // lanes 0 / 2 see lanes 1 / 3
    ge_cached c;
    ge_to_cached(c, o);
    ge_add_cached(d, acc, c);
    fe_pick(acc.X, d.X, (q & 1u) == 0); fe_pick(acc.Y, d.Y, (q & 1u) == 0); fe_pick(acc.Z, d.Z, (q & 1u) == 0); fe_pick(acc.T, d.T, (q & 1u) == 0);
    d = acc;
    ge_double4(d);
    ge_double4(d);                                                // 256 x (windows 3 and 2)
    fe_pick(acc.X, d.X, q == 2u); fe_pick(acc.Y, d.Y, q == 2u); fe_pick(acc.Z, d.Z, q == 2u); fe_pick(acc.T, d.T, q == 2u);
    ge_p3_dpp_from(o, acc, 1);                                    // every lane sees lane 2
    ge_to_cached(c, o);
    ge_add_cached(d, acc, c);
    fe_pick(acc.X, d.X, q == 0u); fe_pick(acc.Y, d.Y, q == 0u); fe_pick(acc.Z, d.Z, q == 0u); fe_pick(acc.T, d.T, q == 0u);
  }
  {
    ge_cached sel, c;
    ge_cached_identity(sel);
    load_comb_entry(c, comb + (size_t)stash[0] * cfg::ENTRIES + 8 * TEETH);      // carry out of bit 255: 2^256 * P
    ge_cached_cmov(sel, c, stash[512]);
    ge_add_cached(acc, acc, sel);
  }
  if (q == 0u) store_ext(partial + stash[256], acc);
}

// ---- grouped comb terms: the table rows pass through LDS, every lane reads the entry its digit names --------------------
// The constant-time walk above pays 72 loads + 288 v_cndmask per addition to hide WHICH of a row's 8 entries a lane wants.
// LDS can hide it for free (hot_tables.h): a ds_read_b128 is serviced in four groups of 16 lanes with distinct lane mod 16, so
// if lane l only ever touches "column" l mod 16 of the LDS (the 16-byte slot l mod 16 of every 256-byte bank row), no two lanes
// of a service group meet on a bank whatever addresses they use.  A column serves 16 lanes of a 256-lane block (tid mod 16
// equal).  The terms of a point with >= GROUP_MIN_USES = 8 cold uses are listed next to each other (CLASS_GROUP, k_comb_slots
// / k_class_scatter), and column c takes 16 CONSECUTIVE list entries: they belong to at most 3 points (a run strictly inside
// the 16 has >= 8 entries), so per (window, tooth) step the column holds that row of at most 3 tables: 3 x 8 entries x 144 B,
// x 16 columns = 54 KB.  The block fetches the rows (each 16-byte chunk ONCE per block and step instead of once per lane: 10 x
// less L1/L2 traffic for CMZ's P) straight into LDS, one step ahead, while the lanes add.
// Per addition and lane: 15 LDS-DMA loads + 9 LDS reads instead of 72 loads + 288 selects.
constexpr int GROUP_RUNS = 3, GROUP_ROW_CHUNKS = 8 * 9, GROUP_IDENT_ROW = GROUP_RUNS * GROUP_ROW_CHUNKS, GROUP_CHUNK_ROWS = GROUP_IDENT_ROW + 9;
constexpr int GROUP_LDS_UINT4 = GROUP_CHUNK_ROWS * 16 + (256 + 16 * GROUP_RUNS) / 4;      // rows | slots[256] | colslot[16][3]
static_assert(GROUP_MIN_USES >= 8, "a column of 16 consecutive grouped terms must span at most GROUP_RUNS points");      // (comb_group_block, ZKP_OPT_CT_LOOKUP = 2)

__device__ __forceinline__ void comb_group_block(uint32_t i0, uint32_t n_g, const uint32_t* __restrict__ list_g, const uint8_t* __restrict__ scalars,
                                                 const uint32_t* __restrict__ pidx, const uint32_t* __restrict__ slot_of,
                                                 const dev_ext* __restrict__ comb, dev_ext* __restrict__ partial, uint4* lds) {
  using cfg = comb_cfg<16>;
  constexpr uint32_t NONE = 0xffffffffu;
  const uint32_t tid = threadIdx.x, col = tid & 15u, k = tid >> 4;
  const uint32_t i = i0 + col * 16 + k;                            // column col takes list entries i0 + 16 col .. + 15
  const bool listed = i < n_g;
  uint32_t t = 0, slot = NONE;
  if (listed) { t = list_g[i]; slot = slot_of[pidx[t]]; }           // (a grouped term's point index is in range by construction)
  uint32_t* slots = reinterpret_cast<uint32_t*>(lds + GROUP_CHUNK_ROWS * 16);
  uint32_t* colslot = slots + 256;
  slots[k * 16 + col] = slot;                                      // (entry k of column col at 16 k + col: a wavefront's reads below meet no bank twice)
  if (k < (uint32_t)GROUP_RUNS) colslot[col * GROUP_RUNS + k] = NONE;
  if (k < 9) {                                                     // the identity in table form (Y-X, Y+X, 2Z, 2dT) = (1, 1, 2, 0), per column
    uint32_t w[36];
#pragma unroll
    for (int q = 0; q < 36; ++q) w[q] = (q == 0 || q == 9) ? 1u : (q == 18 ? 2u : 0u);
    uint4 v = make_uint4(0, 0, 0, 0);
#pragma unroll
    for (int q = 0; q < 9; ++q) if (k == (uint32_t)q) v = make_uint4(w[4 * q], w[4 * q + 1], w[4 * q + 2], w[4 * q + 3]);
    lds[(GROUP_IDENT_ROW + k) * 16 + col] = v;
  }
  __syncthreads();
  uint32_t run = 0;                                                // which of the column's runs of equal slots this lane is in
  bool first = true;
#pragma unroll
  for (uint32_t kk = 1; kk < 16; ++kk) {
    const bool change = slots[kk * 16 + col] != slots[(kk - 1) * 16 + col];
    if (kk <= k) { run += change ? 1u : 0u; if (kk == k) first = change; }
  }
  if (first && run < (uint32_t)GROUP_RUNS) colslot[col * GROUP_RUNS + run] = slot;
  __syncthreads();
  const bool live = listed && slot != NONE && run < (uint32_t)GROUP_RUNS;     // (run < 3 always: see above)
  // this lane's part of the staging: chunks k, k + 16, .. of the row of each of the column's tables
  const uint4* src[GROUP_RUNS];
  bool have[GROUP_RUNS];
#pragma unroll
  for (int r = 0; r < GROUP_RUNS; ++r) {
    const uint32_t sr = colslot[col * GROUP_RUNS + r];
    have[r] = sr != NONE;
    src[r] = reinterpret_cast<const uint4*>(comb + (size_t)(have[r] ? sr : 0u) * cfg::ENTRIES) + k;
  }
  const bool tail = k < (uint32_t)(GROUP_ROW_CHUNKS - 64);         // chunk k + 64 exists for k < 8
  // Rows travel global -> LDS directly (global_load_lds_dwordx4: no staging registers, no ds_write): a wavefront instruction
  // writes LDS at a wave-uniform base + 16 x lane, and with lane = 16 (k mod 4) + col the four chunk rows 4 wave .. 4 wave + 3
  // (+ 16 m) of the column-interleaved layout ARE contiguous in lane order.
  typedef __attribute__((address_space(1))) const void* gptr_t;
  typedef __attribute__((address_space(3))) void* lptr_t;
  const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(tid >> 6));
  auto issue = [&](uint32_t row) {                                 // row = tooth: its 8 entries are chunks 72 row .. 72 row + 71 of a table
#pragma unroll
    for (int r = 0; r < GROUP_RUNS; ++r) {
      const uint4* p = src[r] + (size_t)row * GROUP_ROW_CHUNKS;
      uint4* d = lds + (size_t)(r * GROUP_ROW_CHUNKS + 4 * wave) * 16;
#pragma unroll
      for (int m = 0; m < 4; ++m)
        if (have[r]) __builtin_amdgcn_global_load_lds((gptr_t)(p + 16 * m), (lptr_t)(d + 16 * 16 * m), 16, 0, 0);
      if (have[r] && tail) __builtin_amdgcn_global_load_lds((gptr_t)(p + 64), (lptr_t)(d + 16 * 64), 16, 0, 0);
    }
  };
  // the scalar: signed radix-16 digits nibble - 8; D[w] = the 16 nibbles window w of the 16 teeth, tooth 0 lowest
  uint32_t dlo[cfg::WINDOWS], dhi[cfg::WINDOWS], top = 0;
  {
    uint32_t sc[8], e[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) sc[q] = 0;
    if (live) load_vec<2>(sc, scalars + 32 * (size_t)t);
    sc_add_pattern(e, top, sc, 0x88888888u);
#pragma unroll
    for (int w = 0; w < cfg::WINDOWS; ++w) {
      uint32_t lo = 0, hi = 0;
#pragma unroll
      for (int m = 0; m < 4; ++m) {                                // word m holds teeth 2m (low half) and 2m + 1
        lo |= (((e[m] >> (4 * w)) & 0xfu) | ((e[m] >> (12 + 4 * w)) & 0xf0u)) << (8 * m);
        hi |= (((e[m + 4] >> (4 * w)) & 0xfu) | ((e[m + 4] >> (12 + 4 * w)) & 0xf0u)) << (8 * m);
      }
      dlo[w] = lo; dhi[w] = hi;
    }
  }
  // TWO accumulators (end of round 4): `acc` collects windows 3 and 2, `lo` windows 1 and 0, so that a staged row serves two additions and the table passes
  // through LDS twice instead of four times -- half the LDS-DMA loads, half the barriers, 42 % less HBM fetch in the term kernel (1,433 -> 834 MiB per launch of
  // 20,480 CMZ proofs) -- for 16 doublings + 1 addition per term instead of 12 doublings:
  //   pass 0: acc += T_j[d(j,3)], lo += T_j[d(j,1)] over the teeth j;  acc, lo <- 16 acc, 16 lo;  pass 1: acc += T_j[d(j,2)], lo += T_j[d(j,0)];  result = 256 acc + lo.
  // Window-major with one accumulator (rounds 2 - 4) issued 15 LDS-DMA loads and two barriers per addition: 21 % of a wavefront's walk was spent issuing them
  // (profiles/r04_ab_experiments.txt, block t).  The kernel needs 229 VGPRs this way (249 before: the compiler keeps fewer temporaries alive across the shorter
  // loop body), +2 % on the term kernel at K = 50, +1 - 3 % on the step.  Four accumulators (every row staged once) would need ~72 more registers.
  ge_p3 acc, lo;
  ge_identity(acc);
  ge_identity(lo);
  issue(0);
  auto pick = [&](uint32_t& c0, uint32_t& c1, ge_cached& sel, uint32_t& neg) {
    const uint32_t nib = c0 & 15u;
    c0 = __builtin_amdgcn_alignbit(c1, c0, 4);
    c1 >>= 4;
    neg = (uint32_t)(nib < 8u);
    const uint32_t mag = neg ? 8u - nib : nib - 8u;              // 0..8
    const uint32_t row = mag ? run * GROUP_ROW_CHUNKS + (mag - 1u) * 9u : (uint32_t)GROUP_IDENT_ROW;
    const uint4* ent = lds + (size_t)row * 16 + col;
    uint32_t wd[36];
#pragma unroll
    for (int q = 0; q < 9; ++q) {
      const uint4 x = ent[q * 16];
      wd[4 * q + 0] = x.x; wd[4 * q + 1] = x.y; wd[4 * q + 2] = x.z; wd[4 * q + 3] = x.w;
    }
    fe_set(sel.YmX, wd); fe_set(sel.YpX, wd + 9); fe_set(sel.Z2, wd + 18); fe_set(sel.T2d, wd + 27);
  };
#pragma unroll 1
  for (int pass = 0; pass < 2; ++pass) {
    if (live && pass) { ge_double4(acc); ge_double4(lo); }
    uint32_t alo = pass ? dlo[2] : dlo[3], ahi = pass ? dhi[2] : dhi[3];     // the 16 nibbles of the hi accumulator's window
    uint32_t blo = pass ? dlo[0] : dlo[1], bhi = pass ? dhi[0] : dhi[1];     // ... and of the lo accumulator's
#pragma unroll 1
    for (int j = 0; j < 16; ++j) {
      asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");      // this step's rows have landed, for every wavefront's part
      ge_cached sel;
      uint32_t neg = 0;
      if (live) {
        pick(alo, ahi, sel, neg);
        ge_cached_cneg(sel, neg);
        ge_add_cached(acc, acc, sel);                                      // (the rows stay: the second look-up reads them too)
        pick(blo, bhi, sel, neg);
      }
      lds_barrier();                                               // every lane holds its second entry: the rows may be replaced
      if (pass == 0 || j != 15) issue((uint32_t)((j + 1) & 15));   // next step's rows arrive during the second addition
      if (live) {
        ge_cached_cneg(sel, neg);
        ge_add_cached(lo, lo, sel);
      }
    }
  }
  if (live) {
    ge_double4(acc);
    ge_double4(acc);                                               // 256 hi
    ge_cached c;
    ge_to_cached(c, lo);
    ge_add_cached(acc, acc, c);
  }
  if (live) {
    ge_cached sel, c;
    ge_cached_identity(sel);
    load_comb_entry(c, comb + (size_t)slot * cfg::ENTRIES + 8 * 16);     // carry out of bit 255: 2^256 * P
    ge_cached_cmov(sel, c, top);
    ge_add_cached(acc, acc, sel);
    store_ext(partial + t, acc);
  }
}

// ---- grouped comb terms, look-up on the lane crossbar (round 5): constant time BY CONSTRUCTION ------------------------------------------
// comb_group_block reads LDS at an index the secret digit names and argues with the bank / service-group model that this takes the same
// time for every digit.  Here the digit never becomes an address.  A WAVEFRONT takes 64 consecutive entries of the grouped list and holds
// the row of a (window, tooth) step as 8 tables x 8 entries = 64 (table, entry) pairs: LANE 8 r + (k - 1) HOLDS entry k of the row of the
// wavefront's r-th table.  The pair travels global -> LDS by LDS-DMA into the holder lane's OWN 144-byte slot (addresses: lane number and
// table slot, both public), the holder reads its slot back with ds_read_b128 at that same public address, and the lane whose term is on
// table r with digit magnitude m fetches the 36 words from lane 8 r + m - 1 with ds_bpermute_b32 -- a lane number, not an address.
// What the crossbar still has is BANKS (hot_tables.h, tools/microbench/bpermute_rate.hip): an instruction is served in two groups of 32
// lanes and two lanes of a group whose sources are 32 apart cost an extra cycle.  Lanes of one group read from lanes 8 r .. 8 r + 7 of
// THEIR runs r, and sources 32 apart are entries of runs r and r + 4 -- so a service group must never hold terms of five tables.  Hence
// the shape: a half of the wavefront takes XBAR_HALF_TERMS = 31 consecutive list entries (lanes 31 and 63 carry no term; they are holders
// like every lane), and with GROUP_MIN_USES >= 10 (CMZ's P has exactly 10 terms per proof) 31 consecutive entries span at most
// 2 + floor(29 / 10) = 4 runs -- four CONSECUTIVE numbers, whose source lanes 8 (r mod 4) + k are 32 distinct banks whatever the digits are --
// and the wavefront's 62 entries at most 2 + floor(60 / 10) = 8 = XBAR_RUNS (two static_asserts).  SQ_LDS_BANK_CONFLICT of the walk is zero
// for every scalar (tools/ct_check.py).  A zero digit fetches entry 1 and is masked to the identity.  No block barrier is left: a wavefront
// waits only for its own DMA (vmcnt), so the four wavefronts of a block drift apart freely.  Per addition and wavefront: 4.5 LDS-DMA loads +
// 4.5 ds_read_b128 + 36 crossbar moves (comb_group_block: 7.5 + 9, two barriers per two additions); a table row is fetched once per WAVEFRONT
// that holds terms of its point and pass -- 2 x 1.15 times for CMZ's P against 2 x 1.7 times per 16-lane column before.
constexpr uint32_t XBAR_RUNS = 8;
constexpr uint32_t XBAR_HALF_TERMS = 31, XBAR_WAVE_TERMS = 2 * XBAR_HALF_TERMS, XBAR_BLOCK_TERMS = 4 * XBAR_WAVE_TERMS;      // list entries per half / wavefront / 256-lane block
constexpr int XBAR_WAVE_UINT4 = 9 * 64;                         // a wavefront's staged pairs: chunk q of lane l at q * 64 + l
constexpr int XBAR_LDS_UINT4 = 4 * XBAR_WAVE_UINT4;             // 36 KB per 256-lane block
static_assert(2 + (XBAR_WAVE_TERMS - 2) / GROUP_MIN_USES <= XBAR_RUNS, "a wavefront's consecutive grouped terms must span at most XBAR_RUNS points");
static_assert(2 + (XBAR_HALF_TERMS - 2) / GROUP_MIN_USES <= 4 && XBAR_HALF_TERMS <= 32,
              "the terms of one crossbar service group (a half of the wavefront) must span at most 4 points: sources 32 lanes apart never meet");

__device__ __forceinline__ void comb_group_xbar(uint32_t i0, uint32_t n_g, const uint32_t* __restrict__ list_g, const uint8_t* __restrict__ scalars,
                                                const uint32_t* __restrict__ pidx, const uint32_t* __restrict__ slot_of,
                                                const dev_ext* __restrict__ comb, dev_ext* __restrict__ partial, uint4* lds) {
  using cfg = comb_cfg<16>;
  constexpr uint32_t NONE = 0xffffffffu;
  const uint32_t tid = threadIdx.x, lane = tid & 63u;
  const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(tid >> 6));
  if (i0 + wave * XBAR_WAVE_TERMS >= n_g) return;                  // (a whole wavefront past the end of the list: nothing waits for it)
  // lane l < 31 takes entry l of the wavefront's 62, lane 32 + l entry 31 + l; lanes 31 and 63 take none
  const uint32_t hl = lane & 31u;
  const uint32_t i = i0 + wave * XBAR_WAVE_TERMS + (lane >> 5) * XBAR_HALF_TERMS + hl;
  const bool listed = hl < XBAR_HALF_TERMS && i < n_g;
  uint32_t t = 0, slot = NONE;
  if (listed) { t = list_g[i]; slot = slot_of[pidx[t]]; }           // (a grouped term's point index is in range by construction)
  // runs of equal table slots in the wavefront: lanes without a term (31, 63, past the end of the list) join the run before them -- lanes 31 and
  // 63 pass the slot of their left neighbour on, so that a table whose terms sit on both sides of them stays ONE run
  uint32_t cur = slot;
  const uint32_t left = (uint32_t)__shfl_up((int)cur, 1);
  if (hl == 31u) cur = left;
  const uint32_t before = (uint32_t)__shfl_up((int)cur, 1);
  const bool change = lane == 0u || (listed && slot != before);
  const uint64_t cm = __ballot(change);
  const uint32_t run = (uint32_t)__popcll(cm & ((2ull << lane) - 1ull)) - 1u;
  const bool live = listed && slot != NONE && run < XBAR_RUNS;      // (run < 8 always: see above)
  // holder role: this lane keeps entry (lane & 7) + 1 of the table of run lane >> 3
  int first = -1;
  {
    uint64_t mm = cm;
#pragma unroll
    for (uint32_t r = 0; r < XBAR_RUNS; ++r) {
      const int f = mm ? (int)__ffsll((long long)mm) - 1 : -1;
      if ((lane >> 3) == r) first = f;
      mm &= mm - 1ull;
    }
  }
  uint32_t hslot = (uint32_t)__shfl((int)slot, first < 0 ? 0 : first);
  if (first < 0) hslot = NONE;
  const bool have = hslot != NONE;
  const uint4* hsrc = reinterpret_cast<const uint4*>(comb + (size_t)(have ? hslot : 0u) * cfg::ENTRIES + (lane & 7u));
  typedef __attribute__((address_space(1))) const void* gptr_t;
  typedef __attribute__((address_space(3))) void* lptr_t;
  uint4* wlds = lds + (size_t)wave * XBAR_WAVE_UINT4;
  const uint4* mine = wlds + lane;
  auto issue = [&](uint32_t row) {                                 // row = tooth: its 8 entries are dev_ext 8 row .. 8 row + 7 of a table
    const uint4* p = hsrc + (size_t)row * (8 * 9);
    if (have) {
#pragma unroll
      for (int q = 0; q < 9; ++q) __builtin_amdgcn_global_load_lds((gptr_t)(p + q), (lptr_t)(wlds + 64 * q), 16, 0, 0);
    }
  };
  // the scalar: signed radix-16 digits nibble - 8; D[w] = the 16 nibbles window w of the 16 teeth, tooth 0 lowest
  uint32_t dlo[cfg::WINDOWS], dhi[cfg::WINDOWS], top = 0;
  {
    uint32_t sc[8], e[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) sc[q] = 0;
    if (live) load_vec<2>(sc, scalars + 32 * (size_t)t);
    sc_add_pattern(e, top, sc, 0x88888888u);
#pragma unroll
    for (int w = 0; w < cfg::WINDOWS; ++w) {
      uint32_t lo = 0, hi = 0;
#pragma unroll
      for (int m = 0; m < 4; ++m) {                                // word m holds teeth 2m (low half) and 2m + 1
        lo |= (((e[m] >> (4 * w)) & 0xfu) | ((e[m] >> (12 + 4 * w)) & 0xf0u)) << (8 * m);
        hi |= (((e[m + 4] >> (4 * w)) & 0xfu) | ((e[m + 4] >> (12 + 4 * w)) & 0xf0u)) << (8 * m);
      }
      dlo[w] = lo; dhi[w] = hi;
    }
  }
  // two accumulators as in comb_group_block: `acc` collects windows 3 and 2, `lo` windows 1 and 0; a staged row serves two additions
  ge_p3 acc, lo;
  ge_identity(acc);
  ge_identity(lo);
  issue(0);
  const uint32_t run8 = (run & (XBAR_RUNS - 1u)) * 8u;
  auto pick = [&](uint32_t& c0, uint32_t& c1, ge_cached& sel, uint32_t& neg) {      // EVERY lane runs this: a crossbar source must be an active lane
    const uint32_t nib = c0 & 15u;
    c0 = __builtin_amdgcn_alignbit(c1, c0, 4);
    c1 >>= 4;
    neg = (uint32_t)(nib < 8u);
    const uint32_t mag = neg ? 8u - nib : nib - 8u;              // 0..8
    const uint32_t nz = (uint32_t)(mag != 0u);
    const int src = (int)((run8 + mag - nz) << 2);               // lane 8 run + mag - 1 (entry 1 for a zero digit: masked below)
    uint32_t wd[36];
#pragma unroll
    for (int q = 0; q < 9; ++q) {
      const uint4 x = mine[q * 64];                              // (public address: this lane's own slot)
      wd[4 * q + 0] = xbar_fetch(src, x.x); wd[4 * q + 1] = xbar_fetch(src, x.y);
      wd[4 * q + 2] = xbar_fetch(src, x.z); wd[4 * q + 3] = xbar_fetch(src, x.w);
    }
    const uint32_t m = 0u - nz, id = nz ^ 1u;                      // zero digit: the identity in table form (Y-X, Y+X, 2Z, 2dT) = (1, 1, 2, 0)
#pragma unroll
    for (int q = 0; q < 36; ++q) wd[q] &= m;
    wd[0] |= id; wd[9] |= id; wd[18] |= id << 1;
    fe_set(sel.YmX, wd); fe_set(sel.YpX, wd + 9); fe_set(sel.Z2, wd + 18); fe_set(sel.T2d, wd + 27);
  };
#pragma unroll 1
  for (int pass = 0; pass < 2; ++pass) {
    if (pass) { ge_double4(acc); ge_double4(lo); }
    uint32_t alo = pass ? dlo[2] : dlo[3], ahi = pass ? dhi[2] : dhi[3];     // the 16 nibbles of the hi accumulator's window
    uint32_t blo = pass ? dlo[0] : dlo[1], bhi = pass ? dhi[0] : dhi[1];     // ... and of the lo accumulator's
#pragma unroll 1
    for (int j = 0; j < 16; ++j) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");               // this step's pairs have landed in this wavefront's slots
      ge_cached sel;
      uint32_t neg = 0;
      pick(alo, ahi, sel, neg);
      ge_cached_cneg(sel, neg);
      ge_add_cached(acc, acc, sel);
      pick(blo, bhi, sel, neg);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");             // every slot has been read: the next row may overwrite them
      if (pass == 0 || j != 15) issue((uint32_t)((j + 1) & 15));      // ... and arrives during the second addition
      ge_cached_cneg(sel, neg);
      ge_add_cached(lo, lo, sel);
    }
  }
  ge_double4(acc);
  ge_double4(acc);                                                 // 256 hi
  {
    ge_cached c;
    ge_to_cached(c, lo);
    ge_add_cached(acc, acc, c);
  }
  if (live) {
    ge_cached sel, c;
    ge_cached_identity(sel);
    load_comb_entry(c, comb + (size_t)slot * cfg::ENTRIES + 8 * 16);     // carry out of bit 255: 2^256 * P
    ge_cached_cmov(sel, c, top);
    ge_add_cached(acc, acc, sel);
    store_ext(partial + t, acc);
  }
}

// partial[t] = scalars[t] * P for a point that no other cold term of the call uses (a constraint's left-hand side in
// verify_compact; CMZ's Q in the prover): signed radix-16 ladder over P's own eight multiples, 256 doublings + 65
// additions instead of the radix-4 ladder's 256 + 128.  The multiples live in the term's slot of the ladder scratch.
// CT: the eight multiples are scanned with masks (prover.rs:94 semantics); otherwise the entry is loaded directly.
// Ladder tables are WAVE-INTERLEAVED (round 4): the 64 ladder lanes of a wavefront share one 72 KB group, chunk q of entry e of lane l at
// 16-byte slot (9 e + q) * 64 + l.  A wavefront's load of one chunk is then one contiguous KB (8 cache lines, every byte used) instead of 64
// lines of which 16 bytes each are used (lane-contiguous tables, 1,152 B apart): the texture addresser handles a quarter of the lines per
// constant-time scan (72 loads per addition) and the L1 sees every line once instead of eight times.  Variable-time look-ups (one entry per
// lane, the entry differing between lanes) touch at most the 64 lines they touched before.
constexpr uint32_t LADDER_GROUP_UINT4 = 64u * LADDER_ENTRIES * 9u;          // 16-byte slots per wavefront group
__device__ __forceinline__ void ladder_store_entry(uint4* __restrict__ g, int e, const ge_cached& c) {
  uint32_t w[36];
  fe_get(w, c.YmX); fe_get(w + 9, c.YpX); fe_get(w + 18, c.Z2); fe_get(w + 27, c.T2d);
#pragma unroll
  for (int q = 0; q < 9; ++q) g[(9 * e + q) * 64] = make_uint4(w[4 * q], w[4 * q + 1], w[4 * q + 2], w[4 * q + 3]);
}
__device__ __forceinline__ void ladder_load_entry(ge_cached& c, const uint4* __restrict__ g, uint32_t e) {
  uint32_t w[36];
#pragma unroll
  for (int q = 0; q < 9; ++q) {
    const uint4 v = g[(9u * e + (uint32_t)q) * 64u];
    w[4 * q] = v.x; w[4 * q + 1] = v.y; w[4 * q + 2] = v.z; w[4 * q + 3] = v.w;
  }
  fe_set(c.YmX, w); fe_set(c.YpX, w + 9); fe_set(c.Z2, w + 18); fe_set(c.T2d, w + 27);
}
template <bool CT>
__device__ __forceinline__ void ladder_select(ge_cached& sel, const uint4* __restrict__ g, uint32_t mag) {
  ge_cached_identity(sel);
  if (CT) {
#pragma unroll 1
    for (uint32_t h = 0; h < 4; ++h) {
      ge_cached c0, c1;
      ladder_load_entry(c0, g, 2 * h + 0);
      ladder_load_entry(c1, g, 2 * h + 1);
      ge_cached_cmov(sel, c0, (uint32_t)(mag == 2 * h + 1));
      ge_cached_cmov(sel, c1, (uint32_t)(mag == 2 * h + 2));
    }
  } else if (mag) {
    ladder_load_entry(sel, g, mag - 1);
  }
}

template <bool CT>
__device__ __forceinline__ void term_ladder16(uint32_t t, const uint8_t* __restrict__ scalars, const dev_affine* __restrict__ pt,
                                              uint4* __restrict__ tbl /* this lane's slot 0 of its wavefront group */, dev_ext* __restrict__ partial, uint32_t* ecol) {
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
    ladder_store_entry(tbl, 0, c1);
    ge_double<true>(m2, P);
    ge_to_cached(c, m2); ladder_store_entry(tbl, 1, c);
    ge_add_cached(m3, m2, c1);
    ge_to_cached(c, m3); ladder_store_entry(tbl, 2, c);
    ge_double<true>(m4, m2);
    ge_to_cached(c, m4); ladder_store_entry(tbl, 3, c);
    ge_add_cached(m, m4, c1);
    ge_to_cached(c, m); ladder_store_entry(tbl, 4, c);
    ge_double<true>(m, m3);
    ge_to_cached(c, m); ladder_store_entry(tbl, 5, c);
    ge_add_cached(m, m, c1);
    ge_to_cached(c, m); ladder_store_entry(tbl, 6, c);
    ge_double<true>(m, m4);
    ge_to_cached(c, m); ladder_store_entry(tbl, 7, c);
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
      ladder_select<CT>(sel, tbl, mag);
      ge_cached_cneg(sel, neg);
      ge_add_cached(acc, acc, sel);
    }
  }
  store_ext(partial + t, acc);
}

// Tables of the multiples 1 P .. 128 P for per-proof points whose terms all ride on other terms' doubling chains (stmt_pairs.h: stmt_rider), in the place a
// 16-teeth comb table would take (129 entries; entry k - 1 = k P in the cached form).  One lane per table: a chain of 127 additions.
__global__ void __launch_bounds__(256, 2)
k_rider_tables(const uint32_t* __restrict__ n_slots, uint32_t max_tables, const uint32_t* __restrict__ slot_pt, const dev_affine* __restrict__ pts,
               dev_ext* __restrict__ comb) {
  __shared__ uint32_t stage[4][36][64];                           // (stores through LDS: store_entries_staged; 204,800 tables 2.47 -> 2.02 ms)
  __shared__ uint32_t live[4][64];
  const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
  const uint32_t slot = blockIdx.x * blockDim.x + threadIdx.x, ns = min(*n_slots, max_tables);
  const uint32_t pi = slot < ns ? slot_pt[slot] : 0u;
  const bool have = slot < ns && (pi & STMT_ABSORBED);
  live[wave][lane] = have ? 1u : 0u;
  if (!__ballot(have)) return;                                    // (wave-uniform: nothing to build here)
  ge_p3 P, m;
  ge_identity(P);
  if (have) load_affine(P, pts + (pi & ~STMT_ABSORBED));
  dev_ext* tbl0 = comb + (size_t)(slot - lane) * comb_cfg<16>::ENTRIES;
  ge_cached c1, c;
  ge_to_cached(c1, P);
  c = c1;
  m = P;
#pragma unroll 1
  for (int k = 0; k < 128; ++k) {
    if (k == 1) ge_double<true>(m, P);
    else if (k > 1) ge_add_cached(m, m, c1);
    if (k) ge_to_cached(c, m);
    store_entries_staged(stage[wave], live[wave], lane, tbl0, comb_cfg<16>::ENTRIES, (uint32_t)k, c);
  }
}

// Two terms of one MSM on ONE chain of doublings (variable time; round 6): acc = s1 * P1 + s2 * P2 by interleaving (Straus) -- the second term costs its eight
// multiples and its 64 additions, no doublings, no comb table.  The verifier's constraints  commitment = sum s_i P_i - c * LHS  (verifier.rs:95-106) pair the
// left-hand side (one use per proof: a ladder anyway) with the constraint's per-proof point (CMZ: P, ten uses -- until round 6 a 16-teeth comb table per proof
// and a 64-addition, 16-doubling walk per term), and V with Q.  partial[t2] becomes the identity: the sum of the MSM is unchanged.
__device__ __forceinline__ void ladder_build8(uint4* __restrict__ tbl, ge_cached& c1, const dev_affine* __restrict__ pt) {
  ge_p3 P, m2, m3, m4, m;
  ge_cached c;
  load_affine(P, pt);
  ge_to_cached(c1, P);
  ladder_store_entry(tbl, 0, c1);
  ge_double<true>(m2, P);
  ge_to_cached(c, m2); ladder_store_entry(tbl, 1, c);
  ge_add_cached(m3, m2, c1);
  ge_to_cached(c, m3); ladder_store_entry(tbl, 2, c);
  ge_double<true>(m4, m2);
  ge_to_cached(c, m4); ladder_store_entry(tbl, 3, c);
  ge_add_cached(m, m4, c1);
  ge_to_cached(c, m); ladder_store_entry(tbl, 4, c);
  ge_double<true>(m, m3);
  ge_to_cached(c, m); ladder_store_entry(tbl, 5, c);
  ge_add_cached(m, m, c1);
  ge_to_cached(c, m); ladder_store_entry(tbl, 6, c);
  ge_double<true>(m, m4);
  ge_to_cached(c, m); ladder_store_entry(tbl, 7, c);
}
// rider (optional): the second point's table of multiples 1 .. 128 (k_rider_tables) -- its term then adds one signed 8-bit digit per byte (32 additions) and
// builds nothing; nullptr: eight multiples of its own in tbl2, one signed nibble digit per nibble (64 additions).
__device__ __forceinline__ void term_ladder16_joint(uint32_t t, uint32_t t2, const uint8_t* __restrict__ scalars, const dev_affine* __restrict__ pt,
                                                    const dev_affine* __restrict__ pt2, uint4* __restrict__ tbl, uint4* __restrict__ tbl2,
                                                    const dev_ext* __restrict__ rider, dev_ext* __restrict__ partial, uint32_t* ecol) {
  uint32_t top, top2;
  {
    uint32_t s[8], e[8];
    load_vec<2>(s, scalars + 32 * (size_t)t);
    sc_add_pattern(e, top, s, 0x88888888u);
#pragma unroll
    for (int j = 0; j < 8; ++j) ecol[256 * j] = e[j];
    load_vec<2>(s, scalars + 32 * (size_t)t2);
    sc_add_pattern(e, top2, s, rider ? 0x80808080u : 0x88888888u);    // digits byte - 128 in [-128, 127] / nibble - 8 in [-8, 7]
#pragma unroll
    for (int j = 0; j < 8; ++j) ecol[256 * (8 + j)] = e[j];
  }
  ge_p3 acc;
  ge_identity(acc);
  {
    ge_cached c1;
    ladder_build8(tbl, c1, pt);
    if (top) ge_add_cached(acc, acc, c1);                       // carry out of bit 255: one more P at the top
    if (rider) load_comb_entry(c1, rider);
    else ladder_build8(tbl2, c1, pt2);
    if (top2) ge_add_cached(acc, acc, c1);
  }
#pragma unroll 1
  for (int j = 7; j >= 0; --j) {
    uint32_t cur = ecol[256 * j], cur2 = ecol[256 * (8 + j)];
#pragma unroll 1
    for (int k = 0; k < 8; ++k) {
      ge_double4(acc);
      uint32_t nib = cur >> 28;
      cur <<= 4;
      uint32_t neg = (uint32_t)(nib < 8u);
      uint32_t mag = neg ? 8u - nib : nib - 8u;                 // 0..8
      ge_cached sel;
      ladder_select<false>(sel, tbl, mag);
      ge_cached_cneg(sel, neg);
      ge_add_cached(acc, acc, sel);
      if (rider) {
        if (k & 1) {                                             // a byte of the second scalar is complete: 256^b = 16^(2 b)
          const uint32_t v = cur2 >> 24;
          cur2 <<= 8;
          neg = (uint32_t)(v < 128u);
          mag = neg ? 128u - v : v - 128u;                      // 0..128
          ge_cached_identity(sel);
          if (mag) load_comb_entry(sel, rider + (mag - 1u));
          ge_cached_cneg(sel, neg);
          ge_add_cached(acc, acc, sel);
        }
      } else {
        nib = cur2 >> 28;
        cur2 <<= 4;
        neg = (uint32_t)(nib < 8u);
        mag = neg ? 8u - nib : nib - 8u;
        ladder_select<false>(sel, tbl2, mag);
        ge_cached_cneg(sel, neg);
        ge_add_cached(acc, acc, sel);
      }
    }
  }
  store_ext(partial + t, acc);
  ge_identity(acc);
  store_ext(partial + t2, acc);
}

}  // namespace zkp
