// Fixed-base window tables for points that very many terms share.
//
// In the reference's statements most right-hand-side terms multiply a point that is COMMON to the whole batch
// (define_proof!'s common variables, BatchVerifier's static points: CMZ'13 has X_1..X_10 and A in 20 of the 31
// prover terms, benches/zkp.rs:32-45).  For such a point P the table
//       T[w][k] = k * 2^(W w) * P      w = 0 .. HOT_WINDOWS-1, k = 0 .. 2^(W-1)      (affine niels form, 112 B each)
// turns s*P into HOT_WINDOWS mixed additions (7M each) and NO doublings, against 256 doublings + 128 additions on the
// generic path.  Tables live in HBM/L2 (64 slots), are built once per point and kept across calls.
// Signed radix-2^W digits come from the same carry-free offset recoding as everywhere else: e = s + sum_w 2^(W w + W - 1),
// digit_w = window_w(e) - 2^(W-1) in [-2^(W-1), 2^(W-1) - 1]; e < 2^(W * HOT_WINDOWS) for every 256-bit s, so there is no
// carry window.
//
// How a lane picks its entry (k_terms_split stages one table row per window in LDS; all 256 lanes of a block use the same
// table): the row is written to LDS in SIXTEEN copies, 16-byte chunk q of the row at uint4 index 16 q + copy, and lane l
// reads copy l mod 16 with ds_read_b128.  The LDS services a ds_read_b128 in four fixed groups of 16 lanes whose lane
// numbers are distinct mod 16 (MI355X_MICROARCH.md, LDS), and copy c occupies banks 4c .. 4c+3 of every 256-byte bank row:
// whatever entries the lanes ask for, no two lanes of a group meet on a bank.  The access therefore takes the same 4 LDS
// cycles for every scalar -- constant time WITHOUT the masked scan over all entries (curve25519-dalek's answer to cache
// timing, which costs 8 entries x 27 v_cndmask + 56 LDS reads per addition at radix 16) -- and because the look-up no longer
// grows with the row, the window can be 6 bits: 43 additions per term instead of 65.  tools/ct_check.py counts
// SQ_LDS_BANK_CONFLICT next to the instruction counters.
#pragma once
#include "dev_layout.h"
#include "stmt_pairs.h"

namespace zkp {

// 7 since round 5 (37 additions per term instead of 43; rows of 64 entries are held as two sets of 32 lanes, fixed_base_xbar).  -DZKP_HOT_W=6 builds the
// round 2 - 4 shape, the only one the look-ups 1 and 2 of ZKP_OPT_CT_LOOKUP exist for (A/B builds: profiles/r05_ab_experiments.txt).
#ifndef ZKP_HOT_W
#define ZKP_HOT_W 7
#endif
constexpr int HOT_W = ZKP_HOT_W;                                 // window width in bits: 6, or 7 (crossbar look-ups only: fixed_base_xbar)
constexpr int HOT_WINDOWS = (257 + HOT_W - 1) / HOT_W;           // 43, 37
constexpr int HOT_HALF = 1 << (HOT_W - 1);                       // digits in [-HOT_HALF, HOT_HALF - 1]
constexpr int HOT_ROW = HOT_HALF + 1;                            // entries per row: k * base for k = 0 (identity) .. HOT_HALF
constexpr int HOT_SLOTS = 64;
constexpr int HOT_CLASSES = HOT_SLOTS + 3;                       // class 64 = "cold" terms through a comb table, 65 = cold terms on a ladder,
constexpr int CLASS_COMB = HOT_SLOTS, CLASS_LADDER = HOT_SLOTS + 1, CLASS_GROUP = HOT_SLOTS + 2;   // 66 = comb terms listed point by point
constexpr uint32_t GROUP_MIN_USES = 10;                          // a point's terms form a group from this many cold uses on (CMZ's P has 10; comb_tables.h: 31 consecutive grouped terms then span <= 4 points, 62 <= 8)
constexpr size_t HOT_SLOT_NIELS = (size_t)HOT_WINDOWS * HOT_ROW;
constexpr int HOT_ROW_CHUNKS = HOT_ROW * (int)(sizeof(dev_niels) / 16);   // 16-byte chunks per row
constexpr int HOT_COPIES = 16;
static_assert(HOT_W >= 6 && HOT_W <= 7 && HOT_W * HOT_WINDOWS >= 258, "fixed-base window shape");
constexpr bool HOT_LDS_ROWS = HOT_ROW_CHUNKS <= 256;             // the replicated-LDS-row walk (fixed_base_block) stages one chunk per lane: W = 6 only

// word i of the recoding constant sum_w 2^(W w + W - 1)
constexpr uint32_t hot_pattern_word(int i) {
  uint32_t v = 0;
  for (int w = 0; w < HOT_WINDOWS; ++w) {
    const int bit = HOT_W * w + HOT_W - 1;
    if (bit / 32 == i) v |= 1u << (bit % 32);
  }
  return v;
}
// e = s + pattern over 288 bits
__device__ __forceinline__ void hot_recode(uint32_t e[9], const uint32_t s[8]) {
  constexpr uint32_t P[9] = {hot_pattern_word(0), hot_pattern_word(1), hot_pattern_word(2), hot_pattern_word(3), hot_pattern_word(4),
                             hot_pattern_word(5), hot_pattern_word(6), hot_pattern_word(7), hot_pattern_word(8)};
  uint64_t c = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    c += (uint64_t)s[i] + P[i];
    e[i] = (uint32_t)c;
    c >>= 32;
  }
  e[8] = (uint32_t)c + P[8];
}
// takes the lowest window off e: magnitude 0 .. HOT_HALF and sign of the digit
__device__ __forceinline__ void hot_next_digit(uint32_t e[9], uint32_t& mag, uint32_t& neg) {
  const uint32_t d = e[0] & ((1u << HOT_W) - 1u);
#pragma unroll
  for (int i = 0; i < 8; ++i) e[i] = __builtin_amdgcn_alignbit(e[i + 1], e[i], HOT_W);
  e[8] >>= HOT_W;
  neg = (uint32_t)(d < (uint32_t)HOT_HALF);
  mag = neg ? (uint32_t)HOT_HALF - d : d - (uint32_t)HOT_HALF;
}

// ---- table construction ---------------------------------------------------------------------------------------------
// bases[h][w] = 2^(W w) * P_h      (one lane per point: 256 sequential doublings, paid once per point)
__global__ void __launch_bounds__(64, 2)
k_hot_bases(uint32_t nh, const dev_affine* __restrict__ pts, dev_ext* __restrict__ bases) {
  const uint32_t h = blockIdx.x * blockDim.x + threadIdx.x;
  if (h >= nh) return;
  ge_p3 b;
  load_affine(b, pts + h);
  ge_pin_vgpr(b);
  store_ext(bases + (size_t)h * HOT_WINDOWS, b);
#pragma unroll 1
  for (int w = 1; w < HOT_WINDOWS; ++w) {
#pragma unroll 1
    for (int k = 0; k < HOT_W - 1; ++k) ge_double<false>(b, b);
    ge_double<true>(b, b);
    store_ext(bases + (size_t)h * HOT_WINDOWS + w, b);
  }
}

// one lane per (point, window): the HOT_HALF multiples of the window base, normalised to affine niels with ONE shared
// inversion (Montgomery's trick over their Z coordinates); entry 0 of the row is the identity.  `mult` = global scratch,
// HOT_HALF extended points per lane.
__global__ void __launch_bounds__(64, 2)
k_hot_rows(uint32_t nh, const uint32_t* __restrict__ slots, const dev_ext* __restrict__ bases, dev_ext* __restrict__ mult,
           dev_niels* __restrict__ tables) {
  const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= nh * HOT_WINDOWS) return;
  const uint32_t h = g / HOT_WINDOWS, w = g - h * HOT_WINDOWS;
  dev_ext* mine = mult + (size_t)g * HOT_HALF;
  ge_p3 b, m;
  load_ext(b, bases + (size_t)h * HOT_WINDOWS + w);
  ge_cached bc;
  ge_to_cached(bc, b);
  // pass 1: m_k = k * b, kept in scratch with Z replaced... (X, Y, Z, T) stored; T := prefix product Z_1 .. Z_k
  m = b;
  fe pre = b.Z;
#pragma unroll 1
  for (int k = 1; k <= HOT_HALF; ++k) {
    ge_p3 st = m;
    st.T = pre;
    store_ext(mine + (k - 1), st);
    if (k < HOT_HALF) {
      ge_add_cached(m, m, bc);
      fe_mul(pre, pre, m.Z);
    }
  }
  fe inv, t, d2;
  fe_invert(inv, pre);                                             // 1 / (Z_1 ... Z_HALF)
  fe_from_const(d2, FE_D2);
  dev_niels* row = tables + (size_t)slots[h] * HOT_SLOT_NIELS + (size_t)w * HOT_ROW;
  ge_niels q;
  ge_niels_identity(q);
  store_niels(row, q, 1u);
#pragma unroll 1
  for (int k = HOT_HALF; k >= 1; --k) {
    ge_p3 cur;
    load_ext(cur, mine + (k - 1));
    fe zinv;
    if (k > 1) {
      ge_p3 prev;
      load_ext(prev, mine + (k - 2));
      fe_mul(zinv, inv, prev.T);                                   // prefix product up to k - 1
      fe_mul(inv, inv, cur.Z);
    } else {
      zinv = inv;
    }
    fe x, y;
    fe_mul(x, cur.X, zinv);
    fe_mul(y, cur.Y, zinv);
    fe_add(t, y, x); fe_carry(q.ypx, t);
    fe_sub(t, y, x); fe_carry(q.ymx, t);
    fe_mul(t, x, y);
    fe_mul(q.xy2d, t, d2);
    store_niels(row + k, q, 1u);
  }
}

// hotmap[i] = slot of point i if its encoding is one of the nh registered encodings, else -1
__global__ void __launch_bounds__(256)
k_hot_match(uint32_t n_points, const uint8_t* __restrict__ points, uint32_t nreg, const uint32_t* __restrict__ reg_words /*[nreg][8]*/,
            const int32_t* __restrict__ reg_slot, int32_t* __restrict__ hotmap, uint32_t* __restrict__ any_hot) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_points) return;
  uint32_t w[8];
  load_vec<2>(w, points + 32 * (size_t)i);
  int32_t slot = -1;
  for (uint32_t r = 0; r < nreg; ++r) {
    const uint32_t* k = reg_words + 8 * r;
    if (k[0] != w[0]) continue;
    bool eq = true;
#pragma unroll
    for (int j = 1; j < 8; ++j) eq &= k[j] == w[j];
    if (eq) slot = reg_slot[r];
  }
  hotmap[i] = slot;
  if (slot >= 0) *any_hot = 1u;
}

// ---- term classification: class = table slot (0..63), 64 = per-point comb table, 65 = plain ladder; terms grouped by class --
// uses[p] = number of terms of this call on point p that are not on a fixed-base table.  A comb table costs about as
// much as 1.3 ladders and makes every term on its point 2.8x cheaper, so it pays from the second use on; a point used
// once (a constraint's left-hand side in verify_compact, verifier.rs:101-105) goes to the ladder when comb_min = 2.
// Constant-time calls keep comb_min = 1: every cold term has the same schedule.
__global__ void __launch_bounds__(256)
k_use_count(uint32_t n_terms, const uint32_t* __restrict__ pidx, uint32_t n_points, const int32_t* __restrict__ hotmap,
            uint32_t* __restrict__ uses) {
  // The block counts its 256 terms per distinct point in LDS first (open addressing, 512 slots) and sends ONE atomic per distinct point:
  // a common point that was not registered as a fixed base is named by millions of terms, and atomics on one address serialise
  // beyond the L2s at ~20 ns each (5.9 ms for 2.6 M terms, measured).
  __shared__ uint32_t key[512];
  __shared__ uint32_t cnt[512];
  for (uint32_t i = threadIdx.x; i < 512; i += blockDim.x) { key[i] = 0xffffffffu; cnt[i] = 0; }
  __syncthreads();
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t < n_terms) {
    const uint32_t pi = pidx[t];
    if (pi < n_points && hotmap[pi] < 0) {
      uint32_t slot = (pi * 2654435761u) >> 23;                     // 9 bits
      for (;;) {
        const uint32_t seen = atomicCAS(&key[slot], 0xffffffffu, pi);
        if (seen == 0xffffffffu || seen == pi) break;
        slot = (slot + 1) & 511u;                                   // (at most 256 keys in 512 slots: always terminates)
      }
      atomicAdd(&cnt[slot], 1u);
    }
  }
  __syncthreads();
  for (uint32_t i = threadIdx.x; i < 512; i += blockDim.x)
    if (cnt[i]) atomicAdd(&uses[key[i]], cnt[i]);
}
// group_min: from this many cold uses on, the terms of a point are listed together (CLASS_GROUP) and walk its table through
// LDS (comb_group_block); 0xffffffff = no such class in this call.
__device__ __forceinline__ uint32_t term_class(uint32_t t, const uint32_t* pidx, uint32_t n_points, const int32_t* hotmap,
                                               const uint32_t* uses, uint32_t comb_min, uint32_t group_min) {
  const uint32_t pi = pidx[t];
  if (pi >= n_points) return (uint32_t)CLASS_COMB;                  // out of range: flagged by k_reduce_encode
  const int32_t s = hotmap[pi];
  if (s >= 0) return (uint32_t)s;
  const uint32_t u = uses[pi];
  if (u >= group_min) return (uint32_t)CLASS_GROUP;
  return u >= comb_min ? (uint32_t)CLASS_COMB : (uint32_t)CLASS_LADDER;
}
__global__ void __launch_bounds__(256)
k_class_count(uint32_t n_terms, const uint32_t* __restrict__ pidx, uint32_t n_points, const int32_t* __restrict__ hotmap,
              const uint32_t* __restrict__ uses, uint32_t comb_min, uint32_t group_min, uint32_t* __restrict__ class_cnt) {
  __shared__ uint32_t h[HOT_CLASSES];
  if (threadIdx.x < HOT_CLASSES) h[threadIdx.x] = 0;
  __syncthreads();
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t < n_terms) atomicAdd(&h[term_class(t, pidx, n_points, hotmap, uses, comb_min, group_min)], 1u);
  __syncthreads();
  if (threadIdx.x < HOT_CLASSES && h[threadIdx.x]) atomicAdd(&class_cnt[threadIdx.x], h[threadIdx.x]);
}
// class_start[c] = first list position of class c; class_start[HOT_CLASSES] = n_terms; cursor = copy;
// blk_start[c] = first 256-lane block of fixed-base class c when every class starts a new block (k_terms_split stages one
// table per block in LDS); blk_start[HOT_SLOTS] = number of such blocks
__global__ void k_class_scan(const uint32_t* __restrict__ class_cnt, uint32_t* __restrict__ class_start, uint32_t* __restrict__ cursor,
                             uint32_t* __restrict__ blk_start) {
  if (threadIdx.x != 0) return;
  uint32_t run = 0, blk = 0;
  for (int c = 0; c < HOT_CLASSES; ++c) {
    class_start[c] = run; cursor[c] = run; run += class_cnt[c];
    if (c < HOT_SLOTS) { blk_start[c] = blk; blk += (class_cnt[c] + 255u) / 256u; }
  }
  class_start[HOT_CLASSES] = run;
  blk_start[HOT_SLOTS] = blk;
}
// group_start[p] = offset of point p's terms inside the CLASS_GROUP segment (k_comb_slots), group_fill[p] = 0 on entry: the
// terms of a grouped point land next to each other, in any order
__global__ void __launch_bounds__(256)
k_class_scatter(uint32_t n_terms, const uint32_t* __restrict__ pidx, uint32_t n_points, const int32_t* __restrict__ hotmap,
                const uint32_t* __restrict__ uses, uint32_t comb_min, uint32_t group_min, const uint32_t* __restrict__ class_start,
                const uint32_t* __restrict__ group_start, uint32_t* __restrict__ group_fill, uint32_t* __restrict__ cursor,
                uint32_t* __restrict__ list) {
  __shared__ uint32_t h[HOT_CLASSES];
  __shared__ uint32_t base[HOT_CLASSES];
  if (threadIdx.x < HOT_CLASSES) h[threadIdx.x] = 0;
  __syncthreads();
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  uint32_t c = 0, rank = 0;
  if (t < n_terms) {
    c = term_class(t, pidx, n_points, hotmap, uses, comb_min, group_min);
    if (c != (uint32_t)CLASS_GROUP) rank = atomicAdd(&h[c], 1u);
  }
  __syncthreads();
  if (threadIdx.x < HOT_CLASSES && h[threadIdx.x]) base[threadIdx.x] = atomicAdd(&cursor[threadIdx.x], h[threadIdx.x]);
  __syncthreads();
  if (t < n_terms) {
    if (c == (uint32_t)CLASS_GROUP) {
      const uint32_t pi = pidx[t];
      list[class_start[CLASS_GROUP] + group_start[pi] + atomicAdd(&group_fill[pi], 1u)] = t;
    } else {
      list[base[c] + rank] = t;
    }
  }
}

// ---- the classifier of the fused flows: one launch ------------------------------------------------------------------------
// When the caller is a statement flow (fused_flows.h) every proof of the batch has the same T terms on the same np point ids
// (ids < ns: common to the batch, the others one point per proof), so everything the kernels above find out with atomics is
// arithmetic: use counts, classes, class sizes and block starts, list positions (grouped classes point by point), table slots,
// and the CSR offsets / point indices themselves (k_stmt_index).  Only the fixed-base registry has to be looked at (which common
// points are hot).  Replaces memset + k_stmt_index + k_hot_match + k_use_count + k_class_count + k_class_scan + k_comb_slots +
// k_class_scatter: seven launches fewer per flow, each worth ~1.4 us of a pipelined caller's step.
// Per-proof points are never treated as hot here (one that happens to equal a registered encoding takes the comb path: same result).
constexpr uint32_t STMT_MAX_TERMS = 1024, STMT_MAX_POINTS = 1024;
struct stmt_job {                      // device arrays of the plan + the shape
  const uint32_t* toff = nullptr;      // [nc + 1] term offsets of the constraints
  const uint32_t* tpt = nullptr;       // [T] point id of statement term k
  uint32_t N = 0, T = 0, nc = 0, ns = 0, np = 0;
  uint32_t* off = nullptr;             // out: [N nc + 1]
  uint32_t* pidx = nullptr;            // out: [N T]
  bool on = false;
  // variable-time jobs (round 6): which terms share a chain of doublings (stmt_pairs.h); nullptr: no pairs
  const uint32_t* pair = nullptr;
};

__global__ void __launch_bounds__(256)
k_stmt_classify(const stmt_job sj, const uint8_t* __restrict__ points, uint32_t nreg, const uint32_t* __restrict__ reg_words, const int32_t* __restrict__ reg_slot,
                uint32_t comb_min, uint32_t group_min, uint32_t max_tables, uint32_t* __restrict__ uses, uint32_t* __restrict__ class_start,
                uint32_t* __restrict__ blk_start, uint32_t* __restrict__ n_slots, uint32_t* __restrict__ slot_of, uint32_t* __restrict__ slot_pt,
                uint32_t* __restrict__ list, uint32_t rider_ok) {
  __shared__ uint32_t cnt[STMT_MAX_POINTS];            // terms of the statement on point id p
  __shared__ uint32_t acnt[STMT_MAX_POINTS];           // ... that ride on another term's doubling chain (stmt_pairs.h)
  __shared__ int32_t cls_p[STMT_MAX_POINTS];           // class of the terms on p
  __shared__ uint32_t rank_p[STMT_MAX_POINTS];         // table rank (static: among static table points; instance: among instance ones), or NONE
  __shared__ uint32_t goff_p[STMT_MAX_POINTS];         // grouped points: offset of p's group (static: in the static part; instance: inside a proof's part)
  __shared__ uint32_t pos_k[STMT_MAX_TERMS];           // term k: rank among the statement terms of its class (grouped: among the terms of its point)
  __shared__ uint32_t seen_p[STMT_MAX_POINTS];         // (running count of the terms of p while the ranks are handed out)
  __shared__ uint32_t cstart[HOT_CLASSES + 1];
  __shared__ uint32_t cc[HOT_CLASSES];                 // statement terms per class
  __shared__ uint32_t sh[5];                           // n_static_tab | n_inst_tab | SG (static grouped terms per proof) | UG (instance grouped terms per proof) | ladder terms whose rider has a table
  constexpr uint32_t NONE = 0xffffffffu;
  const uint32_t tid = threadIdx.x, N = sj.N, T = sj.T, ns = sj.ns, np = sj.np;
  for (uint32_t p = tid; p < np; p += 256) { cnt[p] = 0; seen_p[p] = 0; acnt[p] = 0; }
  if (tid < HOT_CLASSES) cc[tid] = 0;
  __syncthreads();
  for (uint32_t k = tid; k < T; k += 256) atomicAdd(stmt_absorbed(sj.pair, k) ? &acnt[sj.tpt[k]] : &cnt[sj.tpt[k]], 1u);
  __syncthreads();
  for (uint32_t p = tid; p < np; p += 256) {           // class of every point id; common points: fixed-base registry first
    int32_t c = -1;
    if (p < ns && nreg) {
      uint32_t w[8];
      load_vec<2>(w, points + 32 * (size_t)p);
      for (uint32_t r = 0; r < nreg; ++r) {
        const uint32_t* q = reg_words + 8 * r;
        bool eq = true;
#pragma unroll
        for (int i = 0; i < 8; ++i) eq &= q[i] == w[i];
        if (eq) c = reg_slot[r];
      }
    }
    if (c < 0) {
      const uint64_t u = p < ns ? (uint64_t)cnt[p] * N : cnt[p];
      c = u >= group_min ? CLASS_GROUP : (u >= comb_min ? CLASS_COMB : CLASS_LADDER);
    }
    cls_p[p] = c;
  }
  __syncthreads();
  if (tid == 0) {                                      // the sequential part: ranks and offsets (np + T + 67 steps)
    uint32_t n_stab = 0, n_itab = 0, SG = 0, UG = 0;
    for (uint32_t p = 0; p < np; ++p) {
      const int32_t c = cls_p[p];
      // a per-proof point ALL of whose terms ride (CMZ's P in the verifier) gets a table of its multiples 1 .. 128 in a comb table's place (k_rider_tables): its
      // riders then add one signed 8-bit digit per byte instead of one nibble digit per nibble
      const bool table = (cnt[p] && (c == CLASS_GROUP || c == CLASS_COMB)) || stmt_rider(rider_ok, p, ns, cnt[p], acnt[p]);
      rank_p[p] = table ? (p < ns ? n_stab++ : n_itab++) : NONE;
      goff_p[p] = 0;
      if (cnt[p] && c == CLASS_GROUP) {
        if (p < ns) { goff_p[p] = SG; SG += cnt[p]; } else { goff_p[p] = UG; UG += cnt[p]; }
      }
    }
    sh[0] = n_stab; sh[1] = n_itab; sh[2] = SG; sh[3] = UG;
    // per-term ranks.  Ladder terms that carry a rider with a table come first, proof by proof, then the others: a wavefront runs one kind of chain
    uint32_t lad_a = 0, lad_b = 0;
    for (uint32_t k = 0; k < T; ++k) {
      if (stmt_absorbed(sj.pair, k)) continue;
      const uint32_t p = sj.tpt[k];
      const int32_t c = cls_p[p];
      if (c == CLASS_GROUP) {
        pos_k[k] = seen_p[p]++;                                      // rank among the terms of p
      } else if (c == CLASS_LADDER) {
        const uint32_t k2 = sj.pair ? sj.pair[k] : STMT_UNPAIRED;
        const bool a = k2 != STMT_UNPAIRED && stmt_rider(rider_ok, sj.tpt[k2], ns, cnt[sj.tpt[k2]], acnt[sj.tpt[k2]]);
        pos_k[k] = a ? (STMT_ABSORBED | lad_a++) : lad_b++;           // (the high bit marks the first segment)
      } else {
        pos_k[k] = cc[c];
      }
      ++cc[c];
    }
    sh[4] = lad_a;
    uint32_t run = 0, blk = 0;
    for (int c = 0; c < HOT_CLASSES; ++c) {
      cstart[c] = run;
      const uint32_t n_c = cc[c] * N;
      if (blockIdx.x == 0) {
        class_start[c] = run;
        if (c < HOT_SLOTS) { blk_start[c] = blk; blk += (n_c + 255u) / 256u; }
      }
      run += n_c;
    }
    cstart[HOT_CLASSES] = run;
    if (blockIdx.x == 0) {
      class_start[HOT_CLASSES] = run;
      blk_start[HOT_SLOTS] = blk;
      *n_slots = n_stab + n_itab * N;
    }
  }
  __syncthreads();
  const size_t g = (size_t)blockIdx.x * 256 + tid;
  const size_t n_points = (size_t)ns + (size_t)(np - ns) * N, n_terms = (size_t)N * T;
  if (g < n_points) {                                   // per point: cold use count (what k_decode_affine looks at), table slot
    const uint32_t p = g < ns ? (uint32_t)g : ns + (uint32_t)((g - ns) / N);
    const uint32_t j = g < ns ? 0u : (uint32_t)((g - ns) % N);
    const int32_t c = cls_p[p];
    const bool cold = c >= HOT_SLOTS;
    const uint64_t u = cold ? (p < ns ? (uint64_t)cnt[p] * N : cnt[p]) : 0;
    uses[g] = (uint32_t)(u > 0xffffffffull ? 0xffffffffull : u);
    uint32_t slot = NONE;
    if (rank_p[p] != NONE) slot = p < ns ? rank_p[p] : sh[0] + rank_p[p] * N + j;
    if (slot != NONE && slot >= max_tables) slot = NONE;   // (cannot happen: max_tables counts every common point as cold)
    const bool rider = slot != NONE && stmt_rider(rider_ok, p, ns, cnt[p], acnt[p]);
    slot_of[g] = rider ? (slot | STMT_ABSORBED) : slot;     // (the high bit: a table of multiples, not a comb table -- read by term_ladder16_joint only: no term of p is on a list)
    if (slot != NONE) slot_pt[slot] = rider ? ((uint32_t)g | STMT_ABSORBED) : (uint32_t)g;
  }
  if (g < (size_t)N * sj.nc) sj.off[g] = (uint32_t)((g / sj.nc) * T + sj.toff[g % sj.nc]);
  if (g == 0) sj.off[(size_t)N * sj.nc] = N * T;
  if (g < n_terms) {
    const uint32_t j = (uint32_t)(g / T), k = (uint32_t)(g % T);
    const uint32_t p = sj.tpt[k];
    sj.pidx[g] = p < ns ? p : ns + (p - ns) * N + j;
    if (stmt_absorbed(sj.pair, k)) return;
    const int32_t c = cls_p[p];
    size_t pos;
    if (c == CLASS_GROUP) {
      // common grouped points first (all N cnt[p] terms of a point together), then proof by proof
      pos = p < ns ? (size_t)goff_p[p] * N + (size_t)j * cnt[p] + pos_k[k]
                   : (size_t)sh[2] * N + (size_t)j * sh[3] + goff_p[p] + pos_k[k];
    } else if (c == CLASS_LADDER) {
      pos = (pos_k[k] & STMT_ABSORBED) ? (size_t)j * sh[4] + (pos_k[k] & ~STMT_ABSORBED) : (size_t)N * sh[4] + (size_t)j * (cc[c] - sh[4]) + pos_k[k];
    } else {
      pos = (size_t)j * cc[c] + pos_k[k];
    }
    list[cstart[c] + pos] = (uint32_t)g;
  }
}

// ---- the fixed-base terms of one block (256 lanes, one table) ----------------------------------------------------------
// rep = HOT_ROW_CHUNKS * HOT_COPIES uint4 of LDS; rows = the table in HBM/L2.  Per window: the row (fetched one window ahead,
// one chunk per lane) goes to LDS in HOT_COPIES copies, the lanes read their entry from their own copy, one mixed addition.
// Lane l writes copy (i + l) mod 16 at step i: a ds_write_b128 is serviced 8 consecutive lanes at a time, and those then hit
// 8 different 16-byte slots of the 128-byte write bank row.
// Barrier for LDS hand-over only: waits for this wavefront's LDS operations (lgkmcnt), NOT for its vector-memory loads --
// __syncthreads() would also wait for the next row's prefetch (vmcnt) and expose its latency once per window.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// SCAN = the safe mode of ZKP_OPT_CT_MASKED_SCANS: the lane reads EVERY entry of the row (from its own copy, at addresses that do not
// depend on the digit) and keeps the one it wants with v_cndmask -- curve25519-dalek's masked scan; 224 LDS reads + 864 selects per
// addition instead of 7 reads, for callers who do not want the secret-index look-up to rest on the LDS service-group / bank model.
template <bool SCAN = false>
__device__ __forceinline__ void fixed_base_block(ge_p3& acc, uint32_t e[9], bool live, const uint4* __restrict__ rows, uint4* rep) {
  const uint32_t tid = threadIdx.x, copy = tid & (HOT_COPIES - 1);
  const bool loader = tid < (uint32_t)HOT_ROW_CHUNKS;
  uint4 v = make_uint4(0, 0, 0, 0);
  if (loader) v = rows[tid];
#pragma unroll 1
  for (int w = 0; w < HOT_WINDOWS; ++w) {
    lds_barrier();                                               // every lane has read its entry of the previous window
    if (loader) {
#pragma unroll
      for (uint32_t i = 0; i < (uint32_t)HOT_COPIES; ++i) rep[tid * HOT_COPIES + ((i + tid) & (HOT_COPIES - 1))] = v;
      if (w + 1 < HOT_WINDOWS) v = rows[(size_t)(w + 1) * HOT_ROW_CHUNKS + tid];
    }
    lds_barrier();
    if (live) {
      uint32_t mag, neg;
      hot_next_digit(e, mag, neg);
      uint32_t wd[28];
      if constexpr (SCAN) {
        const uint4* ent = rep + copy;                               // entry 0 (the identity) first, then every other entry masked in
#pragma unroll
        for (int i = 0; i < 7; ++i) {
          const uint4 x = ent[i * HOT_COPIES];
          wd[4 * i + 0] = x.x; wd[4 * i + 1] = x.y; wd[4 * i + 2] = x.z; wd[4 * i + 3] = x.w;
        }
#pragma unroll 1
        for (uint32_t k = 1; k <= (uint32_t)HOT_HALF; ++k) {
          const uint4* ek = rep + (size_t)k * (sizeof(dev_niels) / 16) * HOT_COPIES + copy;
          const uint32_t m = 0u - (uint32_t)(k == mag);                // all-ones for the wanted entry
#pragma unroll
          for (int i = 0; i < 7; ++i) {
            uint4 x = ek[i * HOT_COPIES];
            // the read happens for EVERY entry: without this the compiler sinks it under "some lane wants entry k" (a branch on the
            // secret digits -- caught by tools/ct_check.py: 9 x the LDS instructions for random scalars)
            asm volatile("" : "+v"(x.x), "+v"(x.y), "+v"(x.z), "+v"(x.w));
            wd[4 * i + 0] = (wd[4 * i + 0] & ~m) | (x.x & m); wd[4 * i + 1] = (wd[4 * i + 1] & ~m) | (x.y & m);
            wd[4 * i + 2] = (wd[4 * i + 2] & ~m) | (x.z & m); wd[4 * i + 3] = (wd[4 * i + 3] & ~m) | (x.w & m);
          }
        }
      } else {
        const uint4* ent = rep + (size_t)mag * (sizeof(dev_niels) / 16) * HOT_COPIES + copy;
#pragma unroll
        for (int i = 0; i < 7; ++i) {
          const uint4 x = ent[i * HOT_COPIES];
          wd[4 * i + 0] = x.x; wd[4 * i + 1] = x.y; wd[4 * i + 2] = x.z; wd[4 * i + 3] = x.w;
        }
      }
      ge_niels q;
      fe_set(q.ypx, wd); fe_set(q.ymx, wd + 9); fe_set(q.xy2d, wd + 18);
      ge_niels_cneg(q, neg);
      ge_madd(acc, acc, q);
    }
  }
}

// ---- the same walk with the look-up on the lane crossbar (round 5): constant time BY CONSTRUCTION ------------------------------------
// The secret digit never becomes a memory address -- not even an LDS one.  A wavefront holds the window's row in REGISTERS, one entry per
// lane, and every lane fetches the entry its digit names with ds_bpermute_b32: the instruction moves VGPR data between lanes over the LDS
// crossbar and takes a source LANE NUMBER instead of an address.  27 (W = 7: 54) crossbar moves per addition replace the 7 secret-indexed
// ds_read_b128 of fixed_base_block<false> -- and the 16 ds_write_b128 per lane and the two block barriers per window that kept the
// replicated rows coherent: a wavefront loads its own row (from L2) and never waits for another one.
//
// The crossbar is not free of banks, though (tools/microbench/bpermute_rate.hip, profiles/r05_bpermute_microbench.txt): the hardware serves
// the instruction like a ds_read_b32 of address 4 x source lane -- two groups of 32 lanes, 32 banks, bank = source lane mod 32 -- and two
// lanes of a group that name sources 32 apart (same bank, different "address") cost an extra LDS cycle: SQ_LDS_BANK_CONFLICT counts 2 per
// instruction for random sources over all 64 lanes, and 0 whenever every source of an instruction lies in ONE half of the wavefront
// (identity, broadcast, l mod 32, l xor 32).  So the layouts below keep the sources of every single instruction inside one half:
//   a row is held as SETS of 32 entries; in a set, lane l < 32 holds words 0..15 of entry l + 1, lane l + 32 words 16..27 of the same entry
//   (16 VGPRs per set); words 0..15 come from lane m, words 16..26 from lane m + 32 (m = magnitude - 1 within the set): the 64 lanes of an
//   instruction all read lanes 0..31, or all read lanes 32..63 -- distinct sources are distinct banks, equal sources broadcast.
//   W = 6: one set (32 non-zero entries), 27 moves.  W = 7: two sets (entries 1..32 and 33..64), both are fetched and the lane keeps the one
//   its magnitude lies in (v_cndmask on a value that is secret but never an address or a branch): 54 moves + 27 selects, 37 additions
//   per term instead of 43.
// A zero digit reads entry 1 and is masked to the identity (1, 1, 0).  Every lane of the wavefront takes part whether it has a term or not
// (a crossbar source must be an active lane); lanes without a term walk the scalar 0 and store nothing.
__device__ __forceinline__ uint32_t xbar_fetch(int src_lane_x4, uint32_t v) { return (uint32_t)__builtin_amdgcn_ds_bpermute(src_lane_x4, (int)v); }

__device__ __forceinline__ void fixed_base_xbar(ge_p3& acc, uint32_t e[9], const uint4* __restrict__ rows) {
  const uint32_t lane = threadIdx.x & 63u;
  constexpr int SETS = HOT_HALF / 32;                                // 1 (W = 6) or 2 (W = 7)
  static_assert(HOT_HALF == 32 * SETS && SETS >= 1 && SETS <= 2, "the crossbar walk holds sets of 32 entries");
  // (entry 0 of a row in memory is the identity: skipped.)  Upper half: chunks 4, 5, 6 of the entry and chunk 6 once more (in bounds)
  const uint4* my = rows + (size_t)(1u + (lane & 31u)) * 7u + (lane >> 5) * 4u;
  const uint32_t last = lane < 32u ? 3u : 2u;
  uint4 r[SETS][4];
#pragma unroll
  for (int s = 0; s < SETS; ++s)
#pragma unroll
    for (int q = 0; q < 4; ++q) r[s][q] = my[(size_t)s * 32 * 7 + (q < 3 ? (uint32_t)q : last)];
#pragma unroll 1
  for (int w = 0; w < HOT_WINDOWS; ++w) {
    uint32_t mag, neg;
    hot_next_digit(e, mag, neg);
    const uint32_t nz = (uint32_t)(mag != 0u);
    const uint32_t m1 = mag - nz;                                     // magnitude - 1 (0 for a zero digit)
    const int src = (int)((m1 & 31u) << 2);                          // lane (magnitude - 1) mod 32, x 4: always in the lower half
    uint32_t wd[28];
#pragma unroll
    for (int s = 0; s < SETS; ++s) {
      uint32_t t[28];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        t[4 * q + 0] = xbar_fetch(src, r[s][q].x); t[4 * q + 1] = xbar_fetch(src, r[s][q].y);
        t[4 * q + 2] = xbar_fetch(src, r[s][q].z); t[4 * q + 3] = xbar_fetch(src, r[s][q].w);
      }
#pragma unroll
      for (int q = 0; q < 3; ++q) {                                   // (+ 128: the same lane number in the upper half)
        t[16 + 4 * q + 0] = xbar_fetch(src + 128, r[s][q].x); t[16 + 4 * q + 1] = xbar_fetch(src + 128, r[s][q].y);
        t[16 + 4 * q + 2] = xbar_fetch(src + 128, r[s][q].z);
        if (q < 2) t[16 + 4 * q + 3] = xbar_fetch(src + 128, r[s][q].w);       // (word 27 is the valid flag: not needed)
      }
      if (s == 0) {
#pragma unroll
        for (int i = 0; i < 27; ++i) wd[i] = t[i];
      } else {
        const bool hi = m1 >= 32u;                                    // the entry lies in the second set
#pragma unroll
        for (int i = 0; i < 27; ++i) wd[i] = hi ? t[i] : wd[i];
      }
    }
    if (w + 1 < HOT_WINDOWS) {                                        // the next row travels during the addition
      const uint4* nx = my + (size_t)(w + 1) * HOT_ROW_CHUNKS;
#pragma unroll
      for (int s = 0; s < SETS; ++s)
#pragma unroll
        for (int q = 0; q < 4; ++q) r[s][q] = nx[(size_t)s * 32 * 7 + (q < 3 ? (uint32_t)q : last)];
    }
    const uint32_t m = 0u - nz;                                       // zero digit: the identity (1, 1, 0)
#pragma unroll
    for (int i = 0; i < 27; ++i) wd[i] &= m;
    wd[0] |= nz ^ 1u;
    wd[9] |= nz ^ 1u;
    ge_niels q;
    fe_set(q.ypx, wd); fe_set(q.ymx, wd + 9); fe_set(q.xy2d, wd + 18);
    ge_niels_cneg(q, neg);
    ge_madd(acc, acc, q);
  }
}

}  // namespace zkp
