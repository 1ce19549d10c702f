// MI355X (gfx950) kernels and C ABI of the ristretto255 MSM engine (include/zkp_mi355x.h).
//
// Two execution paths, both integer-VALU bound (v_mad_u64_u32 chains, see fe25519.h):
//
//  (A) many small MSMs  (zkp_msm_many; reference prover.rs:94, verifier.rs:97)
//        k_decode_affine   1 lane / distinct point      encodings -> (x, y, t), valid flag
//        k_terms_r4        1 lane / (scalar, point) term fixed-schedule signed radix-4 scalar
//                                                       multiplication, table {P, 2P} in VGPRs
//        k_reduce_encode   1 lane / MSM                 sum of its partials + ristretto encode
//
//  (B) one large MSM    (zkp_msm_optional; reference verifier.rs:162, batch_verifier.rs:219)
//        k_pip_prepare<C>  1 lane / term                decode -> affine niels, signed radix-2^C
//                                                       digits, per-window histogram
//        k_pip_scan        1 block / window             exclusive scan of bucket sizes
//        k_pip_scatter     1 lane / (term, window)      counting-sort scatter of term indices
//        k_pip_bucket_part 1 lane / <=L bucket entries  mixed additions (buckets split for load balance)
//        k_pip_bucket_merge 1 lane / (window, bucket)   sum of the bucket's parts
//        k_pip_reduce_lvl  1 lane / 8 inputs            radix-8 tree evaluation of sum_b b*S_b
//        k_pip_combine     1 block                      Horner over windows + ristretto encode
//
// Point addition is commutative and the final encoding canonical, so the (non-deterministic)
// order in which the atomics of k_pip_scatter fill a bucket cannot change a single output bit.
#include <hip/hip_runtime.h>
#include <sys/syscall.h>
#include <unistd.h>
#include <cctype>
#include <cstdio>
#include <cstring>
#include <algorithm>
#include <atomic>
#include <functional>
#include <map>
#include <memory>
#include <mutex>
#include <set>
#include <string>
#include <vector>
#include "../../include/zkp_mi355x.h"
#include "dev_layout.h"

using namespace zkp;

// =============================================================================================
// scalar recoding
// =============================================================================================
// e = s + K where K has the bit pattern `pattern` in every word: signed-digit recoding without a
// sequential carry (digit_i = e_i - 2^(c-1)).  top receives the carry out of bit 255.
__device__ __forceinline__ void sc_add_pattern(uint32_t e[8], uint32_t& top, const uint32_t s[8], uint32_t pattern) {
  uint64_t c = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    c += (uint64_t)s[i] + pattern;
    e[i] = (uint32_t)c;
    c >>= 32;
  }
  top = (uint32_t)c;
}

// w[j] for a run-time j without dynamic register indexing (which would go to scratch)
__device__ __forceinline__ uint32_t sel8(const uint32_t w[8], int j) {
  uint32_t r = w[0];
#pragma unroll
  for (int i = 1; i < 8; ++i) r = (j == i) ? w[i] : r;
  return r;
}

#include "hot_tables.h"
#include "quad.h"
#include "rowfe.h"
#include "comb_tables.h"
#include "sc25519.h"
#include "transcript_kernels.h"

// =============================================================================================
// (A) small-MSM path
// =============================================================================================
__global__ void __launch_bounds__(256, 2)
k_decode_affine(uint32_t n, const uint8_t* __restrict__ enc, dev_affine* __restrict__ out, const uint32_t* __restrict__ only) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  // `only` (optional): decode just the points some term multiplies through their coordinates.  The others are either
  // on a fixed-base table (registered points decoded successfully, hot_tables.h) or referenced by no term of this call.
  if (only && !only[i]) { out[i].valid = 1; return; }
  uint32_t w[8];
  load_vec<2>(w, enc + 32 * (size_t)i);
  ge_p3 p;
  const uint32_t ok = ristretto_decode(p, w);
  store_affine(out + i, p, ok);
}

// partial[t] = scalars[t] * points[pidx[t]].  Fixed schedule: 128 windows of (2 doublings, 1 unified
// addition of a masked-selected table entry); no branch or address depends on the scalar, so the
// same kernel serves ZKP_CT and ZKP_VARTIME.
__device__ __forceinline__ void term_generic(uint32_t t, const uint8_t* __restrict__ scalars, const uint32_t* __restrict__ pidx,
                                             uint32_t n_points, const dev_affine* __restrict__ pts, dev_ext* __restrict__ partial) {
  uint32_t s[8], e[8], top;
  load_vec<2>(s, scalars + 32 * (size_t)t);
  sc_add_pattern(e, top, s, 0xAAAAAAAAu);           // digits e_i - 2 in {-2,-1,0,1}
  ge_p3 P, P2, acc;
  const uint32_t pi = pidx[t];
  load_affine(P, pts + (pi < n_points ? pi : 0u));   // out-of-range index: flagged by k_reduce_encode
  ge_cached c1, c2;
  ge_to_cached(c1, P);
  ge_double<true>(P2, P);
  ge_to_cached(c2, P2);
  ge_identity(acc);
  fe_cmov(acc.X, P.X, top);                          // carry out of bit 255: digit 128 in {0,1}
  fe_cmov(acc.Y, P.Y, top);
  fe_cmov(acc.T, P.T, top);
#pragma unroll 1
  for (int j = 7; j >= 0; --j) {
    uint32_t cur = sel8(e, j);
#pragma unroll 1
    for (int k = 0; k < 16; ++k) {
      const uint32_t d = cur >> 30;
      cur <<= 2;
      ge_double<false>(acc, acc);
      ge_double<true>(acc, acc);
      const uint32_t neg = (uint32_t)(d < 2);
      const uint32_t mag = neg ? 2u - d : d - 2u;
      ge_cached sel;
      ge_cached_identity(sel);
      ge_cached_cmov(sel, c1, (uint32_t)(mag == 1));
      ge_cached_cmov(sel, c2, (uint32_t)(mag == 2));
      ge_cached_cneg(sel, neg);
      ge_add_cached(acc, acc, sel);
    }
  }
  store_ext(partial + t, acc);
}

// every term through the generic path (no fixed-base point registered)
__global__ void __launch_bounds__(256, 2)
k_terms_r4(uint32_t n_terms, const uint8_t* __restrict__ scalars, const uint32_t* __restrict__ pidx,
           uint32_t n_points, const dev_affine* __restrict__ pts, dev_ext* __restrict__ partial) {
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t < n_terms) term_generic(t, scalars, pidx, n_points, pts, partial);
}

// Classified terms in ONE launch, longest first: ladder terms (points with a single cold use of a variable-time call: 321
// point operations per lane), terms on per-proof points with a comb table (BITS - 4 + 65 point operations; masked scans, or --
// points with many uses in large constant-time calls -- rows through LDS: comb_group_block), and the fixed-base terms (43 mixed
// additions, entries read from replicated LDS rows), which fill the SIMDs the others leave idle.
// LADDER = false (every cold point of the call has a table: constant-time calls) only leaves the ladder's code out.
// (Measured and dropped: the same kernel capped at 168 VGPRs for a third wavefront per SIMD needs 39 spills and is no
// faster, 4.51 vs 4.51 M proofs/s pipelined; three separate kernels -- comb 154, fixed-base 161 VGPRs without spills, 3 per
// SIMD -- lose more to serialisation on the stream than the occupancy returns, 4.38 M/s.)
#ifdef ZKP_BUILD_TEST_HOOKS
// per-wavefront cycle recorder of the term kernel (tools/ct_check.py): [block][wavefront] = class << 56 | s_memtime cycles
__device__ uint64_t* g_wave_cycles = nullptr;
__device__ uint32_t g_wave_cycles_cap = 0;
#define ZKP_WAVE_T0 const uint64_t wave_t0_ = __builtin_readcyclecounter();
#define ZKP_WAVE_T1(cls) do { const uint64_t dt_ = __builtin_readcyclecounter() - wave_t0_; const uint32_t wi_ = blockIdx.x * 4 + (threadIdx.x >> 6); \
  if (g_wave_cycles && (threadIdx.x & 63u) == 0 && wi_ < g_wave_cycles_cap) g_wave_cycles[wi_] = ((uint64_t)(cls) << 56) | (dt_ & 0x00ffffffffffffffull); } while (0)
#else
#define ZKP_WAVE_T0
#define ZKP_WAVE_T1(cls)
#endif

// LOOKUP: how the fixed-base blocks and the grouped comb blocks pick a table entry (ZKP_OPT_CT_LOOKUP).
//   LOOKUP_XBAR  the row sits in the wavefront's registers, the entry comes over the lane crossbar (ds_bpermute_b32): no address of any
//                kind depends on the digit -- constant time by construction; the default of every call, and the only walk of 7-bit windows
//   LOOKUP_SCAN  masked scans over whole rows (curve25519-dalek's LookupTable::select); the grouped class does not exist, comb terms scan
//   LOOKUP_LDS   rounds 2 - 4: rows replicated in LDS, ds_read_b128 at an index the digit names, conflict-free under the bank model
enum : int { LOOKUP_XBAR = 0, LOOKUP_SCAN = 1, LOOKUP_LDS = 2 };
template <int LOOKUP>
constexpr int terms_lds_uint4() {
  if (LOOKUP == LOOKUP_XBAR) return XBAR_LDS_UINT4 > 512 ? XBAR_LDS_UINT4 : 512;          // (512: the comb / ladder blocks' scalar columns, 8 words x 256 lanes)
  return HOT_ROW_CHUNKS * HOT_COPIES > GROUP_LDS_UINT4 ? HOT_ROW_CHUNKS * HOT_COPIES : GROUP_LDS_UINT4;
}
template <bool CT, int TEETH, bool LADDER, int LOOKUP = LOOKUP_XBAR, bool SPLIT = false>
__global__ void __launch_bounds__(256, 2)
k_terms_split(const uint8_t* __restrict__ scalars, const uint32_t* __restrict__ pidx, uint32_t n_points,
              const dev_ext* __restrict__ comb, const uint32_t* __restrict__ slot_of, const uint32_t* __restrict__ class_start,
              const uint32_t* __restrict__ blk_start, const uint32_t* __restrict__ list, const dev_niels* __restrict__ tables,
              const dev_affine* __restrict__ pts, dev_ext* __restrict__ ladder_rw, uint32_t max_ladder, dev_ext* __restrict__ partial,
              uint32_t ladder_stride, const uint32_t* __restrict__ pair, uint32_t stmt_T) {
  // A block of fixed-base terms serves ONE table; its rows pass through LDS one window at a time, in 16 copies, so that every
  // lane reads the entry its digit names from banks of its own (hot_tables.h): no masked scan, no bank conflict, the same
  // LDS cycles for every scalar.
  static_assert(LOOKUP == LOOKUP_XBAR || HOT_LDS_ROWS, "7-bit fixed-base windows exist on the crossbar only");
  __shared__ uint4 hot_lds[terms_lds_uint4<LOOKUP>()];
  uint32_t* ecol = reinterpret_cast<uint32_t*>(hot_lds) + threadIdx.x;      // (ladder and comb blocks: the recoded scalars)
  const uint32_t n_hot = class_start[CLASS_COMB], n_comb = class_start[CLASS_LADDER] - n_hot;
  const uint32_t n_ladder = LADDER ? class_start[CLASS_GROUP] - class_start[CLASS_LADDER] : 0u;
  const uint32_t n_group = (CT && TEETH == 16) ? class_start[HOT_CLASSES] - class_start[CLASS_GROUP] : 0u;
  const uint32_t ladder_blocks = (n_ladder + blockDim.x - 1) / blockDim.x;
  // comb_split (constant-time narrow calls on the latency schedule): a QUAD of lanes per comb-scan term, one window each (term_comb_split4)
  const uint32_t comb_lanes = SPLIT ? 4u * n_comb : n_comb;
  const uint32_t comb_blocks = (comb_lanes + blockDim.x - 1) / blockDim.x;
  const uint32_t group_terms = LOOKUP == LOOKUP_XBAR ? XBAR_BLOCK_TERMS : 256u;       // list entries a grouped block takes (crossbar: 31 per half of a wavefront)
  const uint32_t group_blocks = (n_group + group_terms - 1) / group_terms;
  // Logical block number (ladder blocks, comb blocks, grouped blocks, fixed-base blocks -- longest first).  ladder_stride > 1 spreads the
  // ladder blocks over the front of the grid (ladder block i sits at position i * stride) instead of starting them all at once: a ladder
  // lane scans its own 1.1 KB table 65 times, and only as many of those tables as are in flight together have to fit the L2s
  // (one block in `stride` instead of every block of the first wave of the launch).  The mapping depends on the launch shape only.
  uint32_t vb = blockIdx.x;
  if (LADDER && ladder_stride > 1) {
    const uint32_t q = blockIdx.x / ladder_stride, r = blockIdx.x - q * ladder_stride;
    if (r == 0 && q < ladder_blocks) vb = q;
    else vb = ladder_blocks + blockIdx.x - min((blockIdx.x + ladder_stride - 1) / ladder_stride, ladder_blocks);
  }
  ZKP_WAVE_T0
  if (LADDER && vb < ladder_blocks) {
    if constexpr (LADDER) {
      const uint32_t i = vb * blockDim.x + threadIdx.x;
      if (i < n_ladder && i < max_ladder) {                       // (max_ladder bounds n_ladder by construction)
        const uint32_t t = list[n_hot + n_comb + i];
        const uint32_t pi = pidx[t];                              // < n_points (out-of-range indices are classed with the comb terms)
        uint4* tbl = reinterpret_cast<uint4*>(ladder_rw) + (size_t)(i >> 6) * LADDER_GROUP_UINT4 + (i & 63u);
        bool joint = false;
        if constexpr (!CT && terms_lds_uint4<LOOKUP>() >= 1024) {  // (16 words of LDS per lane for the two recoded scalars)
          // pair (variable-time statement jobs): this term carries another term of its MSM through its doublings; the tables of the second points lie behind
          // those of the first ones
          const uint32_t k2 = pair ? pair[t % stmt_T] : STMT_UNPAIRED;
          if (k2 != STMT_UNPAIRED) {
            const uint32_t t2 = t - t % stmt_T + k2, i2 = i + ((max_ladder + 63u) & ~63u), pi2 = pidx[t2];
            const uint32_t s2 = slot_of[pi2];                       // (high bit: the second point has a table of its multiples 1 .. 128, k_rider_tables)
            const dev_ext* rider = (s2 != 0xffffffffu && (s2 & STMT_ABSORBED)) ? comb + (size_t)(s2 & ~STMT_ABSORBED) * comb_cfg<16>::ENTRIES : nullptr;
            term_ladder16_joint(t, t2, scalars, pts + pi, pts + pi2, tbl, reinterpret_cast<uint4*>(ladder_rw) + (size_t)(i2 >> 6) * LADDER_GROUP_UINT4 + (i2 & 63u), rider, partial, ecol);
            joint = true;
          }
        }
        if (!joint) term_ladder16<CT>(t, scalars, pts + pi, tbl, partial, ecol);
      }
    }
    ZKP_WAVE_T1(1);
  } else if (vb < ladder_blocks + comb_blocks) {
    const uint32_t i = (vb - ladder_blocks) * blockDim.x + threadIdx.x;
    if constexpr (SPLIT) {
      {
        const uint32_t ti = i >> 2;                                 // (a quad shares its term: the branches below are uniform inside it)
        if (ti < n_comb) {
          const uint32_t t = list[n_hot + ti];
          const uint32_t pi = pidx[t];
          if (pi < n_points) {
            const uint32_t slot = slot_of[pi];
            if (slot != 0xffffffffu) term_comb_split4<TEETH>(t, i & 3u, scalars, comb, slot, partial, ecol);
          }
        }
        ZKP_WAVE_T1(2);
        return;
      }
    }
    if (i < n_comb) {
      const uint32_t t = list[n_hot + i];
      const uint32_t pi = pidx[t];
      if (pi < n_points) {                                        // (out of range: flagged by k_reduce_encode)
        const uint32_t slot = slot_of[pi];
        if (slot != 0xffffffffu) term_comb<CT, TEETH>(t, scalars, comb + (size_t)slot * comb_cfg<TEETH>::ENTRIES, partial, ecol);
      }
    }
    ZKP_WAVE_T1(2);
  } else if (vb < ladder_blocks + comb_blocks + group_blocks) {
    if constexpr (CT && TEETH == 16) {                            // terms of points with many uses, listed point by point
      if constexpr (LOOKUP == LOOKUP_XBAR)
        comb_group_xbar((vb - ladder_blocks - comb_blocks) * XBAR_BLOCK_TERMS, n_group, list + class_start[CLASS_GROUP], scalars, pidx, slot_of, comb, partial, hot_lds);
      else
        comb_group_block((vb - ladder_blocks - comb_blocks) * 256u, n_group, list + class_start[CLASS_GROUP], scalars, pidx, slot_of, comb, partial, hot_lds);
    }
    ZKP_WAVE_T1(3);
  } else {
    const uint32_t hb = vb - ladder_blocks - comb_blocks - group_blocks;
    if (hb >= blk_start[HOT_SLOTS]) return;                       // (uniform in the block)
    uint32_t c = 0;
    while (blk_start[c + 1] <= hb) ++c;                           // class = table slot of this block
    const uint4* src = reinterpret_cast<const uint4*>(tables + (size_t)c * HOT_SLOT_NIELS);
    const uint32_t i = (hb - blk_start[c]) * 256 + threadIdx.x;
    const bool live = i < class_start[c + 1] - class_start[c];
    const uint32_t t = live ? list[class_start[c] + i] : 0u;
    uint32_t s[8], e[9];
    ge_p3 acc;
    ge_identity(acc);
    if constexpr (LOOKUP == LOOKUP_XBAR) {
      if (hb * 256u + (threadIdx.x & ~63u) - blk_start[c] * 256u >= class_start[c + 1] - class_start[c]) return;    // a wavefront without terms (nothing waits for it)
#pragma unroll
      for (int q = 0; q < 8; ++q) s[q] = 0;
      if (live) load_vec<2>(s, scalars + 32 * (size_t)t);
      hot_recode(e, s);                                            // (lanes without a term walk the scalar 0: they are crossbar sources)
      fixed_base_xbar(acc, e, src);
    } else {
      if (live) {
        load_vec<2>(s, scalars + 32 * (size_t)t);
        hot_recode(e, s);
      }
      fixed_base_block<LOOKUP == LOOKUP_SCAN>(acc, e, live, src, hot_lds);
    }
    if (live) store_ext(partial + t, acc);
    ZKP_WAVE_T1(4);
  }
}

// Which MSM a lane of the reduce / encode kernels works on.  The MSMs of a statement differ in length (CMZ: ten 2-term
// constraints and one 11-term constraint per proof), and a wavefront loops as long as its longest lane: taken in index order
// EVERY wavefront holds a few 11-term MSMs and runs 11 iterations with most lanes idle after 2 (18 k instead of 7 k
// instructions per wavefront in k_encode_prepare).  With the statement known (fused flows) lane p takes MSM
// (p mod N) * nc + order[p / N], order = the constraints by descending length: wavefronts are uniform in length.
struct msm_map {
  uint32_t N = 0, nc = 0;
  const uint32_t* order = nullptr;        // device [nc], or NULL: lane p works on MSM p
};
__device__ __forceinline__ uint32_t msm_of_lane(const msm_map& m, uint32_t p) {
  if (!m.order) return p;
  const uint32_t q = p / m.N;
  return (p - q * m.N) * m.nc + m.order[q];
}

template <typename STATUS_T>
__global__ void __launch_bounds__(256, 2)
k_reduce_encode(uint32_t n_msm, const uint32_t* __restrict__ off, const uint32_t* __restrict__ pidx,
                uint32_t n_points, const dev_affine* __restrict__ pts, const dev_ext* __restrict__ partial,
                uint8_t* __restrict__ out, STATUS_T* __restrict__ status, const msm_map map) {
  const uint32_t lane = blockIdx.x * blockDim.x + threadIdx.x;
  if (lane >= n_msm) return;
  const uint32_t i = msm_of_lane(map, lane);
  const uint32_t b = off[i], e = off[i + 1];
  ge_p3 acc;
  ge_identity(acc);
  uint32_t bad = 0;
#pragma unroll 1
  for (uint32_t t = b; t < e; ++t) {
    ge_p3 q;
    load_ext(q, partial + t);
    ge_add_p3(acc, acc, q);
    const uint32_t pi = pidx[t];
    bad |= pi < n_points ? (pts[pi].valid ^ 1u) : 1u;
  }
  uint32_t w[8];
  ristretto_encode(w, acc);
  if (bad) {
#pragma unroll
    for (int k = 0; k < 8; ++k) w[k] = 0;
  }
  store_vec<2>(out + 32 * (size_t)i, w);
  status[i] = (STATUS_T)bad;
}

// ---- batched encoding: H = sum (s_i / 2) P_i per MSM, output encode(2 H) from ONE field inversion per 65,536 outputs ----
// (ristretto_dc_* in ge25519.h; Montgomery's trick as a product tree: level 1 = blocks of 256 outputs in LDS, level 2 = the
// block products.)  k_reduce_encode, which pays an inverse square root per output, remains for calls below 1,024 terms.
__global__ void __launch_bounds__(256)
k_halve_scalars(uint32_t n, const uint8_t* __restrict__ in, uint8_t* __restrict__ out) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  sc a, r;
  load_vec<2>(a.v, in + 32 * (size_t)i);
  sc_halve(r, a);
  store_vec<2>(out + 32 * (size_t)i, r.v);
}

constexpr int ENC_BLOCK = 256;
struct enc_tree { uint32_t node[2 * ENC_BLOCK][9]; };        // heap layout: node 1 = root, leaves ENC_BLOCK .. 2 ENC_BLOCK - 1
__device__ __forceinline__ void tree_get(fe& r, const enc_tree& t, int n) {
#pragma unroll
  for (int i = 0; i < 9; ++i) r.v[i] = t.node[n][i];
  FE_TRACK(fe_set_ub_tight(r));
}
__device__ __forceinline__ void tree_put(enc_tree& t, int n, const fe& a) {
#pragma unroll
  for (int i = 0; i < 9; ++i) t.node[n][i] = a.v[i];
}
// node = product of its children, up to the root
__device__ __forceinline__ void tree_up(enc_tree& t, int tid) {
  for (int w = ENC_BLOCK / 2; w >= 1; w >>= 1) {
    if (tid < w) {
      fe a, b, c;
      tree_get(a, t, 2 * (w + tid));
      tree_get(b, t, 2 * (w + tid) + 1);
      fe_mul(c, a, b);
      tree_put(t, w + tid, c);
    }
    __syncthreads();
  }
}
// node 1 holds the inverse of the root product; afterwards every leaf holds the inverse of its value
__device__ __forceinline__ void tree_down(enc_tree& t, int tid) {
  for (int w = 1; w <= ENC_BLOCK / 2; w <<= 1) {
    if (tid < w) {
      fe p, a, b, ia, ib;
      tree_get(p, t, w + tid);
      tree_get(a, t, 2 * (w + tid));
      tree_get(b, t, 2 * (w + tid) + 1);
      fe_mul(ia, p, b);
      fe_mul(ib, p, a);
      tree_put(t, 2 * (w + tid), ia);
      tree_put(t, 2 * (w + tid) + 1, ib);
    }
    __syncthreads();
  }
}

__device__ __forceinline__ uint32_t msm_sum(ge_p3& acc, uint32_t i, const uint32_t* __restrict__ off, const uint32_t* __restrict__ pidx,
                                            uint32_t n_points, const dev_affine* __restrict__ pts, const dev_ext* __restrict__ partial) {
  const uint32_t b = off[i], e = off[i + 1];
  ge_identity(acc);
  uint32_t bad = 0;
#pragma unroll 1
  for (uint32_t t = b; t < e; ++t) {
    ge_p3 q;
    load_ext(q, partial + t);
    ge_add_p3(acc, acc, q);
    const uint32_t pi = pidx[t];
    bad |= pi < n_points ? (pts[pi].valid ^ 1u) : 1u;
  }
  return bad;
}

// pass 2: inverses of up to 256 block products per block (one field inversion each)
__device__ __forceinline__ void encode_invert_block(enc_tree& tree, int tid, uint32_t b, uint32_t n_blocks, const uint32_t* __restrict__ bprod,
                                                    uint32_t* __restrict__ binv) {
  fe x;
  fe_1(x);
  if (b < n_blocks) {
#pragma unroll
    for (int k = 0; k < 9; ++k) x.v[k] = bprod[(size_t)b * 9 + k];
    FE_TRACK(fe_set_ub_tight(x));
  }
  tree_put(tree, ENC_BLOCK + tid, x);
  __syncthreads();
  tree_up(tree, tid);
#ifdef ZKP_AB_LANE_INVERT
  if (tid == 0) {
    fe r, inv;
    tree_get(r, tree, 1);
    fe_pin_vgpr(r);            // one lane, one LDS address: left alone, hipcc moves the whole inversion to the scalar ALU (2.3 x slower, 71 SGPR spills)
    fe_invert(inv, r);
    tree_put(tree, 1, inv);
  }
#else
  if (tid < 64) {              // the block's one inversion, which everything after it waits for: a wavefront on it, one limb per lane (rowfe.h)
    rowctx rc;
    row_init(rc);
    const uint32_t z = rc.live ? tree.node[1][rc.k] : 0u;
    const uint32_t inv = row_invert(rc, z);
    if (rc.live && rc.r == 0u) tree.node[1][rc.k] = inv;
  }
#endif
  __syncthreads();
  tree_down(tree, tid);
  if (b < n_blocks) {
#pragma unroll
    for (int k = 0; k < 9; ++k) binv[(size_t)b * 9 + k] = tree.node[ENC_BLOCK + tid][k];
  }
}
// pass 1: per MSM the sum of its partials, the decode status, the encoding state; per block the product of the x's
template <typename STATUS_T>
__global__ void __launch_bounds__(ENC_BLOCK, 2)
k_encode_prepare(uint32_t n_msm, const uint32_t* __restrict__ off, const uint32_t* __restrict__ pidx, uint32_t n_points,
                 const dev_affine* __restrict__ pts, const dev_ext* __restrict__ partial, uint32_t* __restrict__ states /*[n][6][9]*/,
                 uint32_t* __restrict__ xs /*[n][9]*/, uint32_t* __restrict__ bprod /*[blocks][9]*/, uint8_t* __restrict__ zflag,
                 STATUS_T* __restrict__ status, const msm_map map) {
  __shared__ enc_tree tree;
  const int tid = threadIdx.x;
  const uint32_t i = blockIdx.x * ENC_BLOCK + tid;      // lane position: states / xs / zflag are indexed by it, status / out by the MSM
  fe x;
  fe_1(x);
  if (i < n_msm) {
    ge_p3 acc;
    const uint32_t g = msm_of_lane(map, i);
    const uint32_t bad = msm_sum(acc, g, off, pidx, n_points, pts, partial);
    status[g] = (STATUS_T)bad;
    ristretto_dc_state s;
    ristretto_dc_prepare(s, x, acc);
    const uint32_t zero = fe_iszero(x);
    zflag[i] = (uint8_t)zero;
    if (zero) fe_1(x);
    uint32_t* st = states + (size_t)i * 54;
#pragma unroll
    for (int k = 0; k < 9; ++k) {
      st[k] = s.e.v[k]; st[9 + k] = s.f.v[k]; st[18 + k] = s.g.v[k]; st[27 + k] = s.h.v[k]; st[36 + k] = s.eg.v[k]; st[45 + k] = s.fh.v[k];
      xs[(size_t)i * 9 + k] = x.v[k];
    }
  }
  tree_put(tree, ENC_BLOCK + tid, x);
  __syncthreads();
  tree_up(tree, tid);
  if (tid < 9) bprod[(size_t)blockIdx.x * 9 + tid] = tree.node[1][tid];
}

// pass 2 (a kernel of its own: letting the last block of pass 1 do it needs a device-scope fence per block, i.e. an L2
// write-back / invalidate across the eight XCDs, and cost 4 % of the pipelined step)
__global__ void __launch_bounds__(ENC_BLOCK)
k_encode_invert(uint32_t n_blocks, const uint32_t* __restrict__ bprod, uint32_t* __restrict__ binv) {
  __shared__ enc_tree tree;
  encode_invert_block(tree, threadIdx.x, blockIdx.x * ENC_BLOCK + threadIdx.x, n_blocks, bprod, binv);
}

// pass 3: per-output inverses from the block inverse, then the 32 bytes
__global__ void __launch_bounds__(ENC_BLOCK, 2)
k_encode_finish(uint32_t n_msm, const uint32_t* __restrict__ off, const uint32_t* __restrict__ pidx, uint32_t n_points,
                const dev_affine* __restrict__ pts, const dev_ext* __restrict__ partial, const uint32_t* __restrict__ states,
                const uint32_t* __restrict__ xs, const uint32_t* __restrict__ binv, const uint8_t* __restrict__ zflag,
                const uint8_t* __restrict__ status8, const uint32_t* __restrict__ status32, uint8_t* __restrict__ out, const msm_map map) {
  __shared__ enc_tree tree;
  const int tid = threadIdx.x;
  const uint32_t i = blockIdx.x * ENC_BLOCK + tid;
  fe x;
  fe_1(x);
  if (i < n_msm) {
#pragma unroll
    for (int k = 0; k < 9; ++k) x.v[k] = xs[(size_t)i * 9 + k];
    FE_TRACK(fe_set_ub_tight(x));
  }
  tree_put(tree, ENC_BLOCK + tid, x);
  __syncthreads();
  tree_up(tree, tid);
  if (tid < 9) tree.node[1][tid] = binv[(size_t)blockIdx.x * 9 + tid];
  __syncthreads();
  tree_down(tree, tid);
  if (i >= n_msm) return;
  const uint32_t g = msm_of_lane(map, i);
  uint32_t w[8];
  if (zflag[i]) {
    // 2 H lies in the identity coset (or H is one of the few points where e g f h = 0): the general encoder
    ge_p3 acc;
    msm_sum(acc, g, off, pidx, n_points, pts, partial);
    ge_double<true>(acc, acc);
    ristretto_encode(w, acc);
  } else {
    ristretto_dc_state s;
    const uint32_t* st = states + (size_t)i * 54;
#pragma unroll
    for (int k = 0; k < 9; ++k) {
      s.e.v[k] = st[k]; s.f.v[k] = st[9 + k]; s.g.v[k] = st[18 + k]; s.h.v[k] = st[27 + k]; s.eg.v[k] = st[36 + k]; s.fh.v[k] = st[45 + k];
    }
    fe inv;
    tree_get(inv, tree, ENC_BLOCK + tid);
    ristretto_dc_finish(w, s, inv);
  }
  const uint32_t bad = status8 ? (uint32_t)status8[g] : status32[g];
  if (bad) {
#pragma unroll
    for (int k = 0; k < 8; ++k) w[k] = 0;
  }
  store_vec<2>(out + 32 * (size_t)g, w);
}

__global__ void k_iota_single_msm(uint32_t n, uint32_t* __restrict__ pidx, uint32_t* __restrict__ off) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) pidx[i] = i;
  if (i == 0) { off[0] = 0; off[1] = n; }
}

// =============================================================================================
// (B) Pippenger path
// =============================================================================================
template <int C>
struct pip_cfg {
  static constexpr int W = (256 + C - 1) / C;        // windows carrying the -2^(C-1) offset
  static constexpr int W1 = W + 1;                   // + the carry window (digit in {0,1}, no offset)
  static constexpr uint32_t B = 1u << (C - 1);       // bucket indices 1..B; index 0 unused
  static constexpr uint32_t B1 = B + 1;
  // word j of K = sum_{w<W} 2^(C-1 + C w)
  static constexpr uint32_t kword(int j) {
    uint32_t r = 0;
    for (int w = 0; w < W; ++w) {
      const int bit = C - 1 + C * w;
      if (bit / 32 == j) r |= 1u << (bit % 32);
    }
    return r;
  }
};

// K independent MSMs of n terms each in one run ("segmented" Pippenger: the batch index is one more window coordinate of
// every kernel below -- window id = b * W1 + w).  K = 1: one MSM over scalars[n] / points[n] (zkp_msm_optional).  K > 1: the
// operand list of zkp_fused_batch_verify_many -- K batch verifications of N_each proofs whose proofs lie next to each other,
//   scalars = static coefficients [K][ns] || Matrix rows [rows][K * N_each]      points = static [ns] || rows [rows][K * N_each]
// term l of MSM b:  l < ns: static point l with coefficient (b, l);  else  row r = (l - ns) / N_each, proof j = b N_each + (l - ns) % N_each.
struct pip_seg {
  uint32_t K = 1, ns = 0, N_each = 0;
};
__device__ __forceinline__ void pip_seg_src(const pip_seg& seg, uint32_t b, uint32_t l, size_t& si, size_t& pi) {
  si = pi = l;
  if (seg.K > 1) {
    if (l < seg.ns) {
      si = (size_t)b * seg.ns + l;
    } else {
      const uint32_t q = l - seg.ns, r = q / seg.N_each, j = q - r * seg.N_each;
      const size_t col = ((size_t)r * seg.K + b) * seg.N_each + j;
      si = (size_t)seg.K * seg.ns + col;
      pi = (size_t)seg.ns + col;
    }
  }
}

// PHASE: 3 = both halves; 1 = the points only (decode -> niels: nothing here depends on a scalar, so the latency schedule of the batch verifier runs it on the
// side stream next to the transcripts); 2 = the digits only
template <int C, int PHASE = 3>
__global__ void __launch_bounds__(256, 2)
k_pip_prepare(uint32_t n, const pip_seg seg, const uint8_t* __restrict__ scalars, const uint8_t* __restrict__ points,
              dev_niels* __restrict__ niels, uint32_t* __restrict__ digits, uint32_t* __restrict__ invalid) {
  using cfg = pip_cfg<C>;
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x, b = blockIdx.y;
  if (i >= n) return;
  size_t si, pi;
  pip_seg_src(seg, b, i, si, pi);
  niels += (size_t)b * n;
  digits += (size_t)b * cfg::W1 * n;
  if constexpr (PHASE & 1) {
    uint32_t w[8];
    load_vec<2>(w, points + 32 * pi);
    ge_p3 p;
    const uint32_t ok = ristretto_decode(p, w);
    ge_niels q;
    ge_affine_to_niels(q, p);
    store_niels(niels + i, q, ok);
    if (!ok) atomicOr(invalid + b, 1u);
  }
  if constexpr (!(PHASE & 2)) return;
  uint32_t s[8], e[10];
  load_vec<2>(s, scalars + 32 * si);
  const uint32_t flip = sc_fold_sign(s);      // s * P = (l - s) * (-P): the point's sign moves into the digits
  {
    uint64_t c = 0;
#pragma unroll
    for (int j = 0; j < 9; ++j) {
      c += (uint64_t)(j < 8 ? s[j] : 0u) + cfg::kword(j);
      e[j] = (uint32_t)c;
      c >>= 32;
    }
    e[9] = 0;
  }
#pragma unroll
  for (int w = 0; w < cfg::W1; ++w) {
    const int pos = C * w, lo = pos >> 5, sh = pos & 31;
    uint32_t v = e[lo] >> sh;
    if (sh + C > 32) v |= e[lo + 1] << (32 - sh);
    v &= (1u << C) - 1u;
    uint32_t mag, neg;
    if (w < cfg::W) {
      neg = (uint32_t)(v < cfg::B);
      mag = neg ? cfg::B - v : v - cfg::B;
    } else {
      neg = 0;
      mag = v;
    }
    digits[(size_t)w * n + i] = mag | ((neg ^ flip) << 31);
  }
}

// Entries per part of a bucket with cnt entries: about sqrt(cnt) -- a part is one lane's sequential chain, the merge of a bucket's
// parts one quad's -- but never more than 4 L.  A bucket may be HUGE: the 128-bit weights of a batch verification
// (batch_verifier.rs:179) leave a signed-digit carry in the window above their top one, so half of all commitment operands of the
// call meet in bucket 1 of that window (2.9 M entries at 524,288 CMZ proofs; sqrt would be a 1,700-addition chain per lane and
// another in the merge).  Such a bucket gets many parts of 4 L entries and k_pip_bucket_merge sums them with a whole block.
__device__ __forceinline__ uint32_t part_len(uint32_t cnt, uint32_t L) {
  const uint32_t r = (uint32_t)ceilf(sqrtf((float)cnt));
  return r > 4 * L ? 4 * L : (r > L ? r : L);
}
__device__ __forceinline__ uint32_t part_count(uint32_t cnt, uint32_t L) {
  const uint32_t pl = part_len(cnt, L);
  return (cnt + pl - 1) / pl;
}

// exclusive scans over the buckets of one window (one block per window):
//   start[w][b]  = sum_{b' < b} hist[w][b']            position of bucket b in the window's sorted list
//   vstart[w][b] = sum_{b' < b} part_count(hist[w][b'])  first "virtual lane" of bucket b (bucket split into parts)
//   vstart[w][bins] = total number of virtual lanes of the window
__global__ void __launch_bounds__(256)
k_pip_scan(uint32_t bins, uint32_t L, const uint32_t* __restrict__ hist, uint32_t* __restrict__ start,
           uint32_t* __restrict__ cursor, uint32_t* __restrict__ vstart) {
  __shared__ uint32_t part[256];
  __shared__ uint32_t vpart[256];
  const uint32_t w = blockIdx.x, tid = threadIdx.x;
  const uint32_t chunk = (bins + 255) / 256;
  const uint32_t lo = min(tid * chunk, bins), hi = min(lo + chunk, bins);
  const uint32_t* h = hist + (size_t)w * bins;
  uint32_t sum = 0, vsum = 0;
  for (uint32_t b = lo; b < hi; ++b) { sum += h[b]; vsum += part_count(h[b], L); }
  part[tid] = sum;
  vpart[tid] = vsum;
  __syncthreads();
  for (uint32_t d = 1; d < 256; d <<= 1) {
    const uint32_t v = tid >= d ? part[tid - d] : 0;
    const uint32_t vv = tid >= d ? vpart[tid - d] : 0;
    __syncthreads();
    part[tid] += v;
    vpart[tid] += vv;
    __syncthreads();
  }
  uint32_t run = part[tid] - sum, vrun = vpart[tid] - vsum;
  for (uint32_t b = lo; b < hi; ++b) {
    start[(size_t)w * bins + b] = run;
    cursor[(size_t)w * bins + b] = run;
    vstart[(size_t)w * (bins + 1) + b] = vrun;
    run += h[b];
    vrun += part_count(h[b], L);
  }
  if (tid == 255) vstart[(size_t)w * (bins + 1) + bins] = vpart[255];
}

// Counting sort of (term, window) keys by bucket WITHOUT device-scope atomics (measured at only ~8 G/s on MI355X:
// they execute beyond the per-XCD L2s).  Each block owns a tile of terms of one window and counts / ranks in LDS;
// tile histograms go through HBM:  hist pass -> per-bucket totals -> scans -> tile base offsets -> scatter pass.
template <int C>
struct sort_cfg {
  static constexpr int THREADS = C >= 16 ? 1024 : 256;
  static constexpr uint32_t TILE = C >= 16 ? 65536u : 2048u;      // LDS = B1 * 4 bytes: 16 KiB (c = 13) ... 128 KiB (c = 16)
};

template <int C>
__global__ void __launch_bounds__(sort_cfg<C>::THREADS)
k_pip_tile_hist(uint32_t n, const uint32_t* __restrict__ digits, uint32_t* __restrict__ tilehist) {
  using cfg = pip_cfg<C>;
  __shared__ uint32_t h[cfg::B1];
  const uint32_t w = blockIdx.y, t = blockIdx.x, tiles = gridDim.x;
  for (uint32_t b = threadIdx.x; b < cfg::B1; b += blockDim.x) h[b] = 0;
  __syncthreads();
  const uint32_t lo = t * sort_cfg<C>::TILE, hi = min(n, lo + sort_cfg<C>::TILE);
  const uint32_t* d = digits + (size_t)w * n;
  for (uint32_t i = lo + threadIdx.x; i < hi; i += blockDim.x) {
    const uint32_t mag = d[i] & 0x7fffffffu;
    if (mag) atomicAdd(&h[mag], 1u);
  }
  __syncthreads();
  uint32_t* out = tilehist + ((size_t)w * tiles + t) * cfg::B1;
  for (uint32_t b = threadIdx.x; b < cfg::B1; b += blockDim.x) out[b] = h[b];
}

// hist[w][b] = sum over tiles
__global__ void __launch_bounds__(256)
k_pip_tile_total(uint32_t bins, uint32_t tiles, uint32_t total, const uint32_t* __restrict__ tilehist, uint32_t* __restrict__ hist) {
  const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= total) return;
  const uint32_t w = g / bins, b = g - w * bins;
  const uint32_t* p = tilehist + (size_t)w * tiles * bins + b;
  uint32_t sum = 0, t = 0;
  for (; t + 8 <= tiles; t += 8) {                   // eight loads in flight (a column of up to ~1,000 tiles, one lane)
    uint32_t v[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) v[k] = p[(size_t)(t + k) * bins];
#pragma unroll
    for (int k = 0; k < 8; ++k) sum += v[k];
  }
  for (; t < tiles; ++t) sum += p[(size_t)t * bins];
  hist[g] = sum;
}
// tilehist[w][t][b] <- first slot of (tile t, bucket b) in the window's sorted list
__global__ void __launch_bounds__(256)
k_pip_tile_base(uint32_t bins, uint32_t tiles, uint32_t total, const uint32_t* __restrict__ start, uint32_t* __restrict__ tilehist) {
  const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= total) return;
  const uint32_t w = g / bins, b = g - w * bins;
  uint32_t* p = tilehist + (size_t)w * tiles * bins + b;
  uint32_t run = start[g], t = 0;
  for (; t + 8 <= tiles; t += 8) {
    uint32_t v[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) v[k] = p[(size_t)(t + k) * bins];
#pragma unroll
    for (int k = 0; k < 8; ++k) { p[(size_t)(t + k) * bins] = run; run += v[k]; }
  }
  for (; t < tiles; ++t) {
    const uint32_t c = p[(size_t)t * bins];
    p[(size_t)t * bins] = run;
    run += c;
  }
}

template <int C>
__global__ void __launch_bounds__(sort_cfg<C>::THREADS)
k_pip_tile_scatter(uint32_t n, const uint32_t* __restrict__ digits, const uint32_t* __restrict__ tilehist,
                   uint32_t* __restrict__ sorted) {
  using cfg = pip_cfg<C>;
  __shared__ uint32_t base[cfg::B1];
  // Every tile of a window scatters 4-byte entries all over that window's sorted list.  Workgroups go to the 8 XCDs round robin by their linear index, so
  // with (tile, window) = (x, y) the tiles of one window sit on different XCDs and every 128-byte line of the list is written in pieces from several L2s.
  // Windows are dealt out in groups of 8 instead: linear index i -> window 8 (i / (8 tiles)) + i % 8, i.e. window w lives on XCD w % 8 with all its tiles
  // and its list is assembled in ONE L2 (the windows past the last full group of 8 keep the plain order).
  const uint32_t tiles = gridDim.x, WK = gridDim.y, lin = blockIdx.x + blockIdx.y * tiles, full = (WK & ~7u) * tiles;
  uint32_t w, t;
#ifdef ZKP_AB_PLAIN_SCATTER
  if (false) {
#else
  if (lin < full) {
#endif
    const uint32_t grp = lin / (8u * tiles), r = lin - grp * 8u * tiles;
    w = grp * 8u + (r & 7u);
    t = r >> 3;
  } else {
    w = blockIdx.y;
    t = blockIdx.x;
  }
  const uint32_t* in = tilehist + ((size_t)w * tiles + t) * cfg::B1;
  for (uint32_t b = threadIdx.x; b < cfg::B1; b += blockDim.x) base[b] = in[b];
  __syncthreads();
  const uint32_t lo = t * sort_cfg<C>::TILE, hi = min(n, lo + sort_cfg<C>::TILE);
  const uint32_t* d = digits + (size_t)w * n;
  uint32_t* out = sorted + (size_t)w * n;
  for (uint32_t i = lo + threadIdx.x; i < hi; i += blockDim.x) {
    const uint32_t key = d[i], mag = key & 0x7fffffffu;
    if (mag) out[atomicAdd(&base[mag], 1u)] = i | (key & 0x80000000u);
  }
}

// Bucket accumulation, load balanced: virtual lane v of window w sums one part (<= L entries) of one
// bucket.  Buckets are split because digit distributions are NOT uniform in practice: canonical scalars
// are < 2^253, so the top window only ever uses a few dozen buckets, each holding n/32 .. n/64 points.
// vmap[w][v] = bucket of virtual lane v (replaces a 13-step binary search over vstart: 13 dependent global loads)
__global__ void __launch_bounds__(256)
k_pip_vmap(uint32_t bins, uint32_t total, uint32_t vmax, const uint32_t* __restrict__ vstart, uint32_t* __restrict__ vmap) {
  const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
  uint32_t w = 0, b = 0, v0 = 0, v1 = 0;
  if (g < total) {
    w = g / bins; b = g - w * bins;
    const uint32_t* vs = vstart + (size_t)w * (bins + 1);
    v0 = vs[b]; v1 = vs[b + 1];
  }
  // a huge bucket (part_len) has thousands of parts: its entries are written by the whole wavefront
  uint64_t bigmask = __ballot(v1 - v0 > 64u);
  while (bigmask) {
    const int src = __ffsll((unsigned long long)bigmask) - 1;
    bigmask &= bigmask - 1;
    const uint32_t bw = (uint32_t)__shfl((int)w, src), bb = (uint32_t)__shfl((int)b, src), b0 = (uint32_t)__shfl((int)v0, src), b1 = (uint32_t)__shfl((int)v1, src);
    for (uint32_t v = b0 + (threadIdx.x & 63u); v < b1; v += 64) vmap[(size_t)bw * vmax + v] = bb;
  }
  if (v1 - v0 <= 64u)
    for (uint32_t v = v0; v < v1; ++v) vmap[(size_t)w * vmax + v] = b;
}

__global__ void __launch_bounds__(256, 2)
k_pip_bucket_part(uint32_t n, uint32_t W1, uint32_t bins, uint32_t L, uint32_t vmax, const uint32_t* __restrict__ start,
                  const uint32_t* __restrict__ hist, const uint32_t* __restrict__ vstart, const uint32_t* __restrict__ vmap,
                  const uint32_t* __restrict__ sorted, const dev_niels* __restrict__ niels,
                  dev_ext* __restrict__ parts) {
  // (plain (part block, window) order: dealing the windows out in contiguous runs per XCD, so that an XCD's L2 sees at most two batches' operand arrays, lost
  // 10 % on this kernel and 6 % on config 3 -- profiles/r05_ab_experiments.txt block i; the 0.9 GB of gathers per K = 5 launch stay)
  const uint32_t v = blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t w = blockIdx.y;
  const uint32_t* vs = vstart + (size_t)w * (bins + 1);
  if (v >= vs[bins]) return;
  const uint32_t b = vmap[(size_t)w * vmax + v], g = w * bins + b;
  const uint32_t pl = part_len(hist[g], L);
  const uint32_t first = (v - vs[b]) * pl;
  const uint32_t cnt = min(pl, hist[g] - first);
  const uint32_t* lst = sorted + (size_t)w * n + start[g] + first;
  niels += (size_t)(w / W1) * n;                      // window id = batch * W1 + window: the sorted lists hold indices within the batch
  ge_p3 acc;
  ge_identity(acc);
  // ping-pong software pipeline: the gather of the next entry is in flight while the current one is added, and the
  // two buffers are distinct registers so that no copy (and therefore no early wait) sits between load and use
  ge_niels qa, qb;
  uint32_t ia = 0, ib = 0;
  if (cnt) { ia = lst[0]; load_niels(qa, niels + (ia & 0x7fffffffu)); }
#pragma unroll 1
  for (uint32_t k = 0; k < cnt; k += 2) {
    if (k + 1 < cnt) { ib = lst[k + 1]; load_niels(qb, niels + (ib & 0x7fffffffu)); }
    ge_niels_cneg(qa, ia >> 31);
    ge_madd(acc, acc, qa);
    if (k + 1 < cnt) {
      if (k + 2 < cnt) { ia = lst[k + 2]; load_niels(qa, niels + (ia & 0x7fffffffu)); }
      ge_niels_cneg(qb, ib >> 31);
      ge_madd(acc, acc, qb);
    }
  }
  store_ext(parts + (size_t)w * vmax + v, acc);
}

// one QUAD of lanes per bucket (quad.h) sums the bucket's parts in sequence (a latency-bound chain: the buckets of a short top
// window have sqrt(cnt) parts each -- a batch of 32,768 CMZ proofs has 64 such buckets of 75 parts side by side).  The few buckets
// with more than kMergeSeqParts parts -- the huge ones, see part_len -- are then summed by the whole block, one after the other:
// quad i adds up parts i, i + 64, ..., and a tree over the 64 quads (through LDS) finishes.
constexpr uint32_t kMergeSeqParts = 128;
__global__ void __launch_bounds__(256, 2)
k_pip_bucket_merge(uint32_t bins, uint32_t total, uint32_t vmax, const uint32_t* __restrict__ vstart,
                   const dev_ext* __restrict__ parts, dev_ext* __restrict__ buckets) {
  __shared__ uint32_t big[64];
  __shared__ uint32_t n_big;
  __shared__ uint32_t red[64][4][9];
  const uint32_t gt = blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t g = gt >> 2, qi = threadIdx.x >> 2;
  const int q = (int)(gt & 3u);
  if (threadIdx.x == 0) n_big = 0;
  __syncthreads();
  if (g < total) {
    const uint32_t w = g / bins, b = g - w * bins;
    const uint32_t* vs = vstart + (size_t)w * (bins + 1);
    const uint32_t v0 = vs[b], v1 = vs[b + 1];
    const dev_ext* p = parts + (size_t)w * vmax;
    if (v1 - v0 > kMergeSeqParts) {
      if (q == 0) big[atomicAdd(&n_big, 1u)] = g;
    } else {
      qpt acc;
      if (v1 == v0) {
        q_identity(acc, q);
      } else {
        q_load_ext(acc, p + v0, q);
#pragma unroll 1
        for (uint32_t v = v0 + 1; v < v1; ++v) {
          qpt t;
          q_load_ext(t, p + v, q);
          q_add(acc, acc, t, q);
        }
      }
      q_store_ext(buckets + g, acc, q);
    }
  }
  __syncthreads();
  const uint32_t nb = n_big;                                           // (the same for every lane of the block)
#pragma unroll 1
  for (uint32_t i = 0; i < nb; ++i) {
    const uint32_t gb = big[i], w = gb / bins, b = gb - w * bins;
    const uint32_t* vs = vstart + (size_t)w * (bins + 1);
    const uint32_t v0 = vs[b], v1 = vs[b + 1];
    const dev_ext* p = parts + (size_t)w * vmax;
    qpt acc;
    q_identity(acc, q);                                                // (33 .. 63 parts leave some quads empty)
    if (v0 + qi < v1) q_load_ext(acc, p + v0 + qi, q);
#pragma unroll 1
    for (uint32_t v = v0 + qi + 64; v < v1; v += 64) {
      qpt t;
      q_load_ext(t, p + v, q);
      q_add(acc, acc, t, q);
    }
#pragma unroll
    for (int k = 0; k < 9; ++k) red[qi][q][k] = acc.c.v[k];
    __syncthreads();
#pragma unroll 1
    for (uint32_t d = 32; d >= 1; d >>= 1) {
      if (qi < d) {
        qpt t;
#pragma unroll
        for (int k = 0; k < 9; ++k) t.c.v[k] = red[qi + d][q][k];
        q_add(acc, acc, t, q);
#pragma unroll
        for (int k = 0; k < 9; ++k) red[qi][q][k] = acc.c.v[k];
      }
      __syncthreads();
    }
    if (qi == 0) q_store_ext(buckets + gb, acc, q);
    __syncthreads();
  }
}

// One level of the tree evaluation of T = sum_g g * S_g  (g = 0 .. B-1, B a power of two).
// Invariant before a level:  T = sum_j ( A_j + M * j * R_j ),  A absent (= 0) and M = 1 at level 0.
// A quad folds m <= 8 consecutive inputs j = m j' + i:   R' = sum_i R_i,  A' = sum_i A_i + M * sum_i i R_i,  M' = M m.
// (M = 2^shift; chunk sizes: 8 everywhere except a first level of 2 or 4 when log2 B is not a multiple of 3.)
__device__ __forceinline__ void pip_reduce_quad(uint32_t g, int q, uint32_t n_out, uint32_t in_stride, uint32_t out_stride, int level, int m, int shift,
                                                const dev_ext* __restrict__ A_in, const dev_ext* __restrict__ R_in,
                                                dev_ext* __restrict__ A_out, dev_ext* __restrict__ R_out) {
  const uint32_t w = g / n_out, j = g - w * n_out;
  const dev_ext* rin = R_in + (size_t)w * in_stride + (size_t)m * j;
  qpt run, U, t;
  q_identity(run, q);
  q_identity(U, q);
  if (level == 0 && j == n_out - 1) {
    // bucket index B (digit -2^(C-1)) sits one past the tree's range: treat it as local index m of the last chunk,
    // i.e. seed the running sums with it (it is then counted m times in U and once in R')
    q_load_ext(run, rin + m, q);
    U = run;
  }
#pragma unroll 1
  for (int i = m - 1; i >= 1; --i) {
    q_load_ext(t, rin + i, q);
    q_add(run, run, t, q);
    q_add(U, U, run, q);
  }
  q_load_ext(t, rin, q);
  q_add(run, run, t, q);
#pragma unroll 1
  for (int k = 0; k < shift; ++k) q_double(U, U, q);
  if (level > 0) {
    const dev_ext* ain = A_in + (size_t)w * in_stride + (size_t)m * j;
#pragma unroll 1
    for (int i = 0; i < m; ++i) {
      q_load_ext(t, ain + i, q);
      q_add(U, U, t, q);
    }
  }
  q_store_ext(A_out + (size_t)w * out_stride + j, U, q);
  if (R_out) q_store_ext(R_out + (size_t)w * out_stride + j, run, q);
}
__global__ void __launch_bounds__(256, 2)
k_pip_reduce_lvl(uint32_t n_out, uint32_t total, uint32_t in_stride, uint32_t out_stride, int level, int m, int shift,
                 const dev_ext* __restrict__ A_in, const dev_ext* __restrict__ R_in,
                 dev_ext* __restrict__ A_out, dev_ext* __restrict__ R_out) {
  const uint32_t gt = blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t g = gt >> 2;                        // one quad of lanes per output (quad.h)
  const int q = (int)(gt & 3u);
  if (g >= total) return;
  pip_reduce_quad(g, q, n_out, in_stride, out_stride, level, m, shift, A_in, R_in, A_out, R_out);
}

// The last tree level (one output per window: W1 quads of this block), then
// result = sum_w 2^(C w) T_w  (Horner: 253 inherently sequential doublings, W1 - 1 additions);  encode (generic MSM) or identity test (fused
// batch verification: shared_flags).  One block per MSM of the run.  The chain runs on ONE wavefront with one limb per lane (rowfe.h): ~0.45 us
// per doubling against ~1.05 us for a quad of lanes (354 -> 144 us per launch, profiles/r05_ab_experiments.txt blocks h, i); what can be done beside the chain is done before it -- the W1 quads turn their window sums
// into cached operands (one multiplication level each) so that an addition of the chain is two levels, not three.
__global__ void __launch_bounds__(256)
k_pip_combine(int W1, int C, uint32_t in_stride, int level, int m, int shift, const dev_ext* __restrict__ A_in, const dev_ext* __restrict__ R_in,
              dev_ext* __restrict__ T_all, const uint32_t* __restrict__ invalid, uint8_t* __restrict__ out_point, uint32_t* __restrict__ status,
              uint32_t shared_flags /*1: invalid[b] also carries bit 1 = "a transcript rejected a proof" -> status[2 b + 1]*/) {
  const uint32_t b = blockIdx.x;
  dev_ext* T = T_all + (size_t)b * W1;             // this MSM's window sums
  const uint32_t g = threadIdx.x >> 2;
  const int q = (int)(threadIdx.x & 3u);
  if (g < (uint32_t)W1) pip_reduce_quad(b * (uint32_t)W1 + g, q, 1u, in_stride, 1u, level, m, shift, A_in, R_in, T_all, nullptr);
  __syncthreads();                                 // (T is written and read by this block only)
  // LDS: window w's operand at 36 w + 9 (coordinate) + limb -- the top window as the point itself (the chain starts from it), the others cached
  __shared__ uint32_t Tl[64 * 36];                 // (W1 <= 64: pip_run)
  if (g < (uint32_t)W1) {
    qpt t;
    q_load_ext(t, T + g, q);
    if (g + 1 < (uint32_t)W1) {
      qcached cc;
      q_to_cached(cc, t, q);
      t.c = cc.c;
    }
#pragma unroll
    for (int i = 0; i < 9; ++i) Tl[36 * g + 9 * q + i] = t.c.v[i];
  }
  __syncthreads();
  if (threadIdx.x < 64) {
    rowctx rc;
    row_init(rc);
    const uint32_t at = 9u * rc.r + (rc.live ? rc.k : 0u);
    uint32_t acc = rc.live ? Tl[36 * (W1 - 1) + at] : 0u;
#pragma unroll 1
    for (int k = W1 - 2; k >= 0; --k) {
#pragma unroll 1
      for (int d = 0; d < C; ++d) acc = row_double(rc, acc);
      const uint32_t t = rc.live ? Tl[36 * k + at] : 0u;
      acc = row_add_cached(rc, acc, t);
    }
    if (rc.live) Tl[at] = acc;                     // (window 0's operand has been consumed: every lane of this wavefront is past its read)
  }
  __syncthreads();
  if (threadIdx.x >= 4) return;                    // one quad of lanes: lane q now holds coordinate q
  qpt acc;
#pragma unroll
  for (int i = 0; i < 9; ++i) acc.c.v[i] = Tl[9 * q + i];
  ge_p3 full;
  q_gather(full, acc);
  uint32_t o[8];
  if (shared_flags) {
    // the fused batch verifications read a verdict off these 32 bytes, nothing else (zkp_mi355x.h: "the batch verifies iff ... d_out_point holds 32 zero
    // bytes"): a point lies in the identity's coset iff X = 0 or Y = 0, which is when its canonical encoding is 32 zero bytes (batch_verifier.rs:230-234
    // is_identity()) -- no inverse square root at the end of the chain.  out = those zero bytes, or a non-zero marker.
#pragma unroll
    for (int k = 0; k < 8; ++k) o[k] = 0;
    o[0] = (fe_iszero(full.X) | fe_iszero(full.Y)) ^ 1u;
  } else {
    ristretto_encode(o, full);
  }
  if (q != 0) return;
  const uint32_t flags = invalid[b];
  const uint32_t bad = flags & 1u;
  if (bad) {
#pragma unroll
    for (int k = 0; k < 8; ++k) o[k] = 0;
  }
  store_vec<2>(out_point + 32 * (size_t)b, o);
  if (shared_flags) { status[2 * b] = bad; status[2 * b + 1] = (flags >> 1) & 1u; }
  else status[b] = bad;
}

#ifdef ZKP_BUILD_TEST_HOOKS
// self-test hook for the 4-lane cooperative arithmetic: for pair i (P, Q): out[i] = enc(2P), enc(P+Q), enc(P+Q) via the
// niels form of Q, enc(P-Q) via the negated niels form
__global__ void __launch_bounds__(256, 2)
k_debug_quad(uint32_t n, const uint8_t* __restrict__ enc, uint8_t* __restrict__ out) {
  const uint32_t gt = blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t i = gt >> 2;
  const int q = (int)(gt & 3u);
  if (i >= n) return;
  uint32_t w[8];
  ge_p3 P, Q, R;
  load_vec<2>(w, enc + 64 * (size_t)i);
  ristretto_decode(P, w);
  load_vec<2>(w, enc + 64 * (size_t)i + 32);
  ristretto_decode(Q, w);
  // stage the operands through LDS-free scratch: build quad forms from the full points of this lane
  qpt p, s, r;
  p.c = q == 0 ? P.X : (q == 1 ? P.Y : (q == 2 ? P.Z : P.T));
  s.c = q == 0 ? Q.X : (q == 1 ? Q.Y : (q == 2 ? Q.Z : Q.T));
  __shared__ dev_niels nl[64];
  ge_niels qn;
  ge_affine_to_niels(qn, Q);
  if (q == 0) store_niels(&nl[threadIdx.x >> 2], qn, 1u);
  __syncthreads();
  uint32_t o[8];
  for (int op = 0; op < 4; ++op) {
    if (op == 0) q_double(r, p, q);
    else if (op == 1) q_add(r, p, s, q);
    else { qcached c; q_load_niels(c, &nl[threadIdx.x >> 2], q, op == 3 ? 1u : 0u); q_add_cached(r, p, c, q); }
    q_gather(R, r);
    ristretto_encode(o, R);
    if (q == 0) store_vec<2>(out + 128 * (size_t)i + 32 * op, o);
  }
}
// self-test hook for the one-limb-per-lane arithmetic (rowfe.h), one wavefront per pair (P, Q): out[i] = enc(2P), enc(P+Q), enc(2^11 P + Q)
__global__ void __launch_bounds__(64)
k_debug_row(uint32_t n, const uint8_t* __restrict__ enc, uint8_t* __restrict__ out) {
  const uint32_t i = blockIdx.x;
  if (i >= n) return;
  __shared__ uint32_t S[3 * 36];                   // P (X, Y, Z, T), Q cached (Y-X, Y+X, 2Z, 2dT), result
  if (threadIdx.x == 0) {
    uint32_t w[8];
    ge_p3 P, Q;
    load_vec<2>(w, enc + 64 * (size_t)i);
    ristretto_decode(P, w);
    load_vec<2>(w, enc + 64 * (size_t)i + 32);
    ristretto_decode(Q, w);
    fe c[4], d2;
    fe_sub(c[0], Q.Y, Q.X);
    fe_carry(c[0], c[0]);
    fe_add(c[1], Q.Y, Q.X);
    fe_carry(c[1], c[1]);
    fe_add(c[2], Q.Z, Q.Z);
    fe_carry(c[2], c[2]);
    fe_from_const(d2, FE_D2);
    fe_mul(c[3], Q.T, d2);
    const fe* pc[4] = {&P.X, &P.Y, &P.Z, &P.T};
    for (int r = 0; r < 4; ++r)
      for (int k = 0; k < 9; ++k) { S[9 * r + k] = pc[r]->v[k]; S[36 + 9 * r + k] = c[r].v[k]; }
  }
  __syncthreads();
  rowctx rc;
  row_init(rc);
  const uint32_t at = 9u * rc.r + (rc.live ? rc.k : 0u);
  const uint32_t p = rc.live ? S[at] : 0u, qc = rc.live ? S[36 + at] : 0u;
  for (int op = 0; op < 3; ++op) {
    uint32_t r;
    if (op == 0) r = row_double(rc, p);
    else if (op == 1) r = row_add_cached(rc, p, qc);
    else {
      r = p;
#pragma unroll 1
      for (int d = 0; d < 11; ++d) r = row_double(rc, r);
      r = row_add_cached(rc, r, qc);
    }
    __syncthreads();
    if (rc.live) S[72 + at] = r;
    __syncthreads();
    if (threadIdx.x == 0) {
      ge_p3 R;
      for (int k = 0; k < 9; ++k) { R.X.v[k] = S[72 + k]; R.Y.v[k] = S[81 + k]; R.Z.v[k] = S[90 + k]; R.T.v[k] = S[99 + k]; }
      uint32_t o[8];
      ristretto_encode(o, R);
      store_vec<2>(out + 96 * (size_t)i + 32 * op, o);
    }
  }
}
#endif  // ZKP_BUILD_TEST_HOOKS

// =============================================================================================
// batch-verification coefficient build (batch_verifier.rs:173-206) with scalar arithmetic mod l on the device
// =============================================================================================
// incidence of point id p: entries inc_off[p] .. inc_off[p+1]) of (constraint k, secret index or ~0 for "p is the lhs")
__device__ __forceinline__ void coeff_of_point(sc& acc, uint32_t p, uint32_t j, uint32_t N, uint32_t m,
                                               const uint32_t* __restrict__ inc_off, const uint32_t* __restrict__ inc_k,
                                               const uint32_t* __restrict__ inc_sc, const uint8_t* __restrict__ minus_c,
                                               const uint8_t* __restrict__ responses, const uint8_t* __restrict__ weights16,
                                               size_t wk = 0, size_t wj = 1) {
  if (wk == 0) wk = N;                                             // default layout [n_constraints][N][16] (batch_verifier.rs:179)
  sc_zero(acc);
  for (uint32_t e = inc_off[p]; e < inc_off[p + 1]; ++e) {
    const uint32_t k = inc_k[e], svar = inc_sc[e];
    sc r, rm, x, t;
    sc_zero(r);
    load_vec<1>(r.v, weights16 + 16 * ((size_t)k * wk + (size_t)j * wj));   // Scalar::from(u128)
    sc_to_mont(rm, r);
    const uint8_t* src = svar == 0xffffffffu ? minus_c + 32 * (size_t)j : responses + 32 * ((size_t)j * m + svar);
    load_vec<2>(x.v, src);
    sc_mont(t, x, rm);                                             // x * r mod l (x may be any 256-bit value)
    sc_add(acc, acc, t);
  }
}

// The whole coefficient build in one launch, grid (K * ceil(N_each / 256), ni + nc + ns) for K batches of N_each proofs that lie
// next to each other (N = K * N_each proofs; K = 1: one batch):
//   blockIdx.y <  ni + nc : instance rows and commitment rows of the coefficient matrix, lane (row, proof)
//   blockIdx.y >= ni + nc : static coefficients -- block-level partial sums over the proofs OF ONE BATCH (a block never straddles
//                           two batches; k_coeff_static_final adds them up per batch: batch_verifier.rs:187, :198 sum over the batch)
// scalars = static coefficients [K][ns] || Matrix rows [ni + nc][N]
__global__ void __launch_bounds__(256)
k_coeff_build(uint32_t N, uint32_t N_each, uint32_t nblk_each, uint32_t K, uint32_t m, uint32_t ns, uint32_t ni, uint32_t nc, const uint32_t* __restrict__ inc_off,
              const uint32_t* __restrict__ inc_k, const uint32_t* __restrict__ inc_sc, const uint8_t* __restrict__ minus_c,
              const uint8_t* __restrict__ responses, const uint8_t* __restrict__ weights16, uint8_t* __restrict__ scalars,
              uint32_t* __restrict__ partial /*[ns][gridDim.x][8]*/) {
  __shared__ uint32_t red[256][8];
  const uint32_t b = blockIdx.x / nblk_each, jl = (blockIdx.x - b * nblk_each) * blockDim.x + threadIdx.x;
  const bool live = jl < N_each;
  const uint32_t j = b * N_each + jl;
  const uint32_t row = blockIdx.y;
  if (row < ni + nc) {
    if (!live) return;
    sc acc;
    if (row < ni) {
      coeff_of_point(acc, ns + row, j, N, m, inc_off, inc_k, inc_sc, minus_c, responses, weights16);
    } else {
      sc r;
      sc_zero(r);
      load_vec<1>(r.v, weights16 + 16 * ((size_t)(row - ni) * N + j));
      sc_neg(acc, r);                                                // batch_verifier.rs:183
    }
    store_vec<2>(scalars + 32 * ((size_t)K * ns + (size_t)row * N + j), acc.v);
    return;
  }
  const uint32_t s = row - ni - nc;
  sc acc;
  sc_zero(acc);
  if (live) coeff_of_point(acc, s, j, N, m, inc_off, inc_k, inc_sc, minus_c, responses, weights16);
#pragma unroll
  for (int i = 0; i < 8; ++i) red[threadIdx.x][i] = acc.v[i];
  __syncthreads();
  for (uint32_t d = 128; d >= 1; d >>= 1) {
    if (threadIdx.x < d) {
      sc a, bb;
#pragma unroll
      for (int i = 0; i < 8; ++i) { a.v[i] = red[threadIdx.x][i]; bb.v[i] = red[threadIdx.x + d][i]; }
      sc_add(a, a, bb);
#pragma unroll
      for (int i = 0; i < 8; ++i) red[threadIdx.x][i] = a.v[i];
    }
    __syncthreads();
  }
  if (threadIdx.x < 8) partial[((size_t)s * gridDim.x + blockIdx.x) * 8 + threadIdx.x] = red[0][threadIdx.x];
}
// grid (ns, K), 64 lanes: static coefficient s of batch b = sum of its nblk_each block partials (2,048 of them at 524,288 proofs:
// a lone lane adding them one after the other took 0.6 ms)
__global__ void __launch_bounds__(64)
k_coeff_static_final(uint32_t nblk_each, uint32_t ns, const uint32_t* __restrict__ partial, uint8_t* __restrict__ scalars) {
  __shared__ uint32_t red[64][8];
  const uint32_t s = blockIdx.x, b = blockIdx.y, nblocks = nblk_each * gridDim.y;
  sc acc, t;
  sc_zero(acc);
  for (uint32_t q = threadIdx.x; q < nblk_each; q += 64) {
#pragma unroll
    for (int i = 0; i < 8; ++i) t.v[i] = partial[((size_t)s * nblocks + (size_t)b * nblk_each + q) * 8 + i];
    sc_add(acc, acc, t);
  }
#pragma unroll
  for (int i = 0; i < 8; ++i) red[threadIdx.x][i] = acc.v[i];
  __syncthreads();
  for (uint32_t d = 32; d >= 1; d >>= 1) {
    if (threadIdx.x < d) {
#pragma unroll
      for (int i = 0; i < 8; ++i) t.v[i] = red[threadIdx.x + d][i];
      sc_add(acc, acc, t);
#pragma unroll
      for (int i = 0; i < 8; ++i) red[threadIdx.x][i] = acc.v[i];
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) store_vec<2>(scalars + 32 * ((size_t)b * ns + s), acc.v);
}

// =============================================================================================
// stand-alone codec kernels
// =============================================================================================
__global__ void __launch_bounds__(256, 2)
k_decode_check(uint32_t n, const uint8_t* __restrict__ enc, uint8_t* __restrict__ status, uint8_t* __restrict__ xyzt) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  uint32_t w[8];
  load_vec<2>(w, enc + 32 * (size_t)i);
  ge_p3 p;
  const uint32_t ok = ristretto_decode(p, w);
  status[i] = (uint8_t)(ok ^ 1u);
  if (xyzt) {
    uint32_t o[32];
    fe_towords(o, p.X); fe_towords(o + 8, p.Y); fe_towords(o + 16, p.Z); fe_towords(o + 24, p.T);
    store_vec<8>(xyzt + 128 * (size_t)i, o);
  }
}

__global__ void __launch_bounds__(256, 2)
k_encode_many(uint32_t n, const uint8_t* __restrict__ xyzt, uint8_t* __restrict__ out) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  uint32_t w[32];
  load_vec<8>(w, xyzt + 128 * (size_t)i);
  ge_p3 p;
  fe_fromwords(p.X, w); fe_fromwords(p.Y, w + 8); fe_fromwords(p.Z, w + 16); fe_fromwords(p.T, w + 24);
  uint32_t o[8];
  ristretto_encode(o, p);
  store_vec<2>(out + 32 * (size_t)i, o);
}

// =============================================================================================
// host side: context, workspace, C ABI
// =============================================================================================
static thread_local std::string g_last_error = "";
struct zkp_ctx;
// contexts that are alive (a graph may outlive its context: launching it then is an error, not a use after free)
static std::mutex g_ctx_mu;
static std::set<zkp_ctx*> g_live_ctx;
// every context gets a process-unique id: a graph remembers the id, not just the address, so a context that happens to be
// allocated where a destroyed one lived (both generations still 0) cannot replay the dead context's recording
static std::atomic<uint64_t> g_ctx_next_id{1};

static int fail(int code, const std::string& msg) {
  g_last_error = msg;
  return code;
}
#define HIP_TRY(expr)                                                                          \
  do {                                                                                         \
    hipError_t e_ = (expr);                                                                    \
    if (e_ != hipSuccess)                                                                      \
      return fail(e_ == hipErrorOutOfMemory ? ZKP_ERR_OOM : ZKP_ERR_HIP,                       \
                  std::string(#expr) + ": " + hipGetErrorString(e_));                          \
  } while (0)

constexpr int kMaxEvents = 32;     // timing marks per call (a fused flow records more than one per kind)
struct zkp_ctx {
  uint64_t uid = 0;                        // process-unique (g_ctx_next_id)
  int device = 0;
  hipStream_t own_stream = nullptr;
  hipStream_t stream = nullptr;
  hipStream_t side_stream = nullptr;       // fused flows: point phase of the MSM next to the transcripts
  hipEvent_t ev_fork = nullptr, ev_join = nullptr;
  bool prof_suspended = false;
  bool capturing = false;                  // between zkp_ctx_capture_begin / _end: nothing may allocate or synchronise
  uint64_t batch_encode_min = 65536;       // ZKP_OPT_BATCH_ENCODE_MIN
  bool batch_encode_user = false;          // set explicitly: applies to every entry point as is
  int comb_teeth = 4;                      // ZKP_OPT_COMB_TEETH (generic _dev entry point; the other callers derive it)
  // ZKP_OPT_CT_SINGLE_USE_TABLES.  A table costs 256 + 7 TEETH point operations and then BITS - 4 + 65 per term, a ladder 7 + 256 +
  // 65: by instruction count a single-use point belongs on the ladder (328 against 445 at 16 teeth).  But a ladder lane is a
  // 321-operation dependent chain inside the term kernel (0.94 ms at 4096 proofs, against 0.41 ms for everything else in it),
  // while the same doublings spent on a table run next to the other tables' chains.  So, for constant-time calls (variable-time
  // ones, whose ladder has no masked scans, always use it): the synchronous entry points, where one call's latency is what the
  // caller sees, give single-use points a table; the asynchronous _dev entry points, whose callers keep many calls in flight
  // and are bound by instruction issue, take the ladder.  -1 = that rule (default), 0 = ladder, 1 = tables.
  int ct_single_use_tables = -1;
  // a transcript program the flow wants run next to the comb-table construction of the term path's point phase (one launch:
  // k_tables_transcript); consumed by msm_terms_path if it builds tables with one lane per point, else run by the flow itself
  struct { bool offered = false, active = false; const tr_op* ops = nullptr; uint32_t n_ops = 0; const uint64_t* tables = nullptr; uint32_t N = 0; tr_bufs bufs{};
           uint8_t* ts = nullptr; uint32_t* saved = nullptr; uint32_t* failed = nullptr; uint32_t tail = 0;
           bool steps = false; tr_steps_dev sd; uint64_t* img = nullptr; } pending_tr;       // steps: the program in step form (assemble + chain)
  int debug_dummy_launches = 0;      // ZKP_TESTOPT_DUMMY_LAUNCHES (test-hook builds only): empty kernels added to every prove call
  bool stmt_classify = true;         // the fused flows' one-launch term classifier (ZKP_TESTOPT_GENERIC_CLASSIFIER of test-hook builds turns it off)
  int fuse_tables_transcript = -1;       // ZKP_OPT_FUSE_TABLES_TRANSCRIPT: -1 = asynchronous _dev calls below kVeryWideCallProofs proofs, 0 = never, 1 = always
  int tables_lane = -1;              // ZKP_OPT_TABLES_LANE: comb tables built by one lane per point (1) or by a quad (0); -1 = by entry point
  int comb_split = -1;               // ZKP_OPT_COMB_SPLIT: -1 = narrow constant-time calls on the latency schedule, 0 = never, 1 = always (with the grouped walk's own rule)
  int grouped_comb = -1;             // ZKP_OPT_GROUPED_COMB: -1 = calls of kGroupedCombTerms terms or more, 0 = never, 1 = always
  int ladder_interleave = -1;        // ZKP_OPT_LADDER_INTERLEAVE: the term kernel's ladder blocks spread over the front of its grid instead of all first
                                     // (-1 = by size: from kInterleaveLadderBlocks ladder blocks up, where the later start of the last one no longer shows)
  static constexpr uint32_t kInterleaveLadderBlocks = 256;
  bool each_straus = true;           // ZKP_OPT_EACH_STRAUS: verify_batchable's per-proof MSMs as one Straus walk per proof (0: one ladder per operand)
  uint32_t each_straus_lanes = 0;    //   ... with this many lanes per proof (0 = by batch size)
  uint32_t each_straus_wins = 0;     //   ... or with this many window parts per proof (0 = by batch size)
  int ct_lookup = 0;                 // ZKP_OPT_CT_LOOKUP: how constant-time calls pick a table entry: 0 lane crossbar (default), 1 masked scans, 2 LDS rows read at the digit's index (rounds 2 - 4)
#ifdef ZKP_BUILD_TEST_HOOKS
  uint64_t* wave_cycles = nullptr;   // ZKP_TESTOPT_WAVE_CYCLES: per-wavefront cycle recorder of the term kernel
  static constexpr uint32_t kWaveCyclesCap = 1u << 20;
#endif
  static constexpr size_t kGroupedCombTerms = 400000;
  static constexpr size_t kSplitCombTerms = 8192;   // narrow constant-time calls on the latency schedule from this many terms on: grouped walk + quad-split scans (ZKP_OPT_COMB_SPLIT)
  bool dev_overlap = false;          // ZKP_OPT_DEV_OVERLAP: the _dev flows fork their scalar-independent half onto the side stream too
  bool dev_latency = false;          // ZKP_OPT_DEV_OVERLAP = 2: the _dev flows run the whole latency schedule of the synchronous entry points (a lone caller's choice)
  bool rider_tables = true;          // ZKP_OPT_JOINT_LADDER = 2 turns the riders' shared tables of multiples off (pairs stay)
  bool joint_ladder = true;          // ZKP_OPT_JOINT_LADDER: variable-time statement flows carry a second term of a constraint on a single-use point's doubling chain (1, default) or not (0)
  bool tr_steps = true;              // ZKP_OPT_TRANSCRIPT_STEPS: lane-pair transcripts as assemble + chain (1, default) or by the word-operation interpreter (0)
  int tr_lanes = -1;                 // ZKP_OPT_TRANSCRIPT_LANES: -1 = by entry point, 1 = one lane per proof, 2 = a lane pair per proof
  // The instruction-saving variants of the asynchronous entry points (ladder for single-use points, one transcript lane per
  // proof) lengthen a call's narrow kernels; they pay once a single call fills the chip.  Measured on CMZ batches
  // (profiles/r02_ab_experiments.txt): 4096 proofs (127 k terms) -1 % / 0 %, 16384 proofs (508 k terms) +7 % / +2.5 %.
  static constexpr size_t kWideCallTerms = 250000, kWideCallProofs = 8192;
  // Round 3: callers hand over several batches per call and keep only a few calls in flight.  Between 8,192 and 65,536 proofs a call's
  // transcript and table kernels are still a few hundred wavefronts of long dependent chains: there the lane-pair transcript (half the
  // latency) running inside the comb tables' launch wins (20 steps of 4,096 proofs as 4 calls of 5 batches: 5.7 -> 6.15 M proofs/s;
  // 1000 steps, 10 batches per call: +1 %); from 65,536 proofs on every kernel fills the chip and the instruction-saving forms win
  // (one transcript lane per proof, separate launches: fusing costs 1.5 - 3 % at 20 - 50 batches per call).  profiles/r03_ab_experiments.txt
  static constexpr size_t kVeryWideCallProofs = 65536;
  uint32_t ct_comb_min(bool throughput, size_t n_terms) const {
    return ct_single_use_tables < 0 ? (throughput && n_terms >= kWideCallTerms ? 2u : 1u) : (ct_single_use_tables ? 1u : 2u);
  }
  void* ws = nullptr;
  size_t ws_bytes = 0;
  // Captured graphs hold raw pointers into the workspace and into the fused plans' device blocks: each is stamped with these
  // generation numbers, and zkp_graph_launch refuses a graph whose stamps are stale (the workspace was reallocated by a larger
  // call, the plan cache was flushed) instead of running recorded kernels on freed memory.
  uint64_t ws_generation = 0, plans_generation = 0;
  bool profiling = false;
  hipEvent_t ev[kMaxEvents] = {};
  int ev_kind[kMaxEvents] = {};
  int n_ev = 0;
  float kernel_ms[ZKP_K_COUNT] = {};
  float total_ms = 0;
  std::string kernel_names[ZKP_K_COUNT];   // profiling: which variant of a kind's size- / option-dependent kernel the last call launched (zkp_ctx_last_kernels)
  // fixed-base tables (hot_tables.h): HOT_SLOTS slots, LRU
  dev_niels* hot_tables = nullptr;
  uint32_t* hot_reg_words = nullptr;       // device [HOT_SLOTS][8]: encodings of the registered points, densely packed
  int32_t* hot_reg_slot = nullptr;         // device [HOT_SLOTS]
  char* hot_scratch = nullptr;             // device: encodings, decoded points, window bases of points being added
  std::string hot_key[HOT_SLOTS];          // host: encoding held by each slot ("" = free)
  uint64_t hot_used[HOT_SLOTS] = {};
  uint64_t hot_tick = 0;
  uint32_t hot_nreg = 0;
  // fused flows (fused_flows.h): compiled transcript programs and operand templates per (flow, statement, N, position)
  std::map<std::string, void*> fused_plans;
  // asynchronous host-buffer jobs (host_jobs.h): at most one in flight per context
  struct job_t {
    char kind = 0;                     // 0 = none pending; 'P' prove, 'V' verify_compact, 'E' verify_batchable (each), 'B' batch verification(s)
    hipEvent_t done = nullptr;         // recorded behind the job's last copy
    uint8_t* pin = nullptr;            // pinned scratch for the few words zkp_ctx_job_wait turns into verdicts (MSM outputs, status words)
    size_t pin_bytes = 0;
    uint32_t K = 0;
    int* verdicts = nullptr;           // 'B': [K], the caller's
    int* invalid_point = nullptr;      // 'P': the caller's
    uint8_t* results = nullptr;        // 'V' / 'E': [n_results], the caller's -- set to 1 (rejected) whenever the job fails, at submit or at wait
    size_t n_results = 0;
    hipError_t copy_err = hipSuccess;  // what issuing the deferred copies out returned (zkp_ctx_job_poll may be the one that issues them; zkp_ctx_job_wait reports it)
    hipEvent_t tev[4] = {};            // profiling: job start | inputs on the device | flow done | outputs on the host
    bool timed = false;
    float ms[3] = {};                  // host -> device copies, kernels, device -> host copies of the last job (zkp_ctx_job_timing)
    // Device -> host copies are issued by zkp_ctx_job_wait once the kernels are done, not queued behind them at submit: a copy that sits in
    // the copy engine's queue waiting for its kernels blocks every later copy of every other context behind it (host_jobs.h).
    struct out_copy { void* dst; const void* src; size_t bytes; };
    std::vector<out_copy> outs;
    hipEvent_t copied = nullptr;       // recorded behind the deferred copies once they are issued
    bool copies_issued = false;
    bool inline_out = false;           // a synchronous call: its copies out are queued with its kernels (job_defers)
  } job;
  int defer_d2h = -1;                  // ZKP_OPT_JOB_DEFER_D2H: -1 = default (1), 0 = queue the copies out at submit
  bool sync_throughput = false;        // ZKP_OPT_SYNC_SCHEDULE: the synchronous host-pointer entry points run the jobs' throughput schedule (callers with a thread per context)
  size_t ws_limit = 0;                 // ZKP_OPT_WS_LIMIT_BYTES: a call that would need a larger workspace fails with ZKP_ERR_OOM (0 = no cap)
  bool hot_registry_uploaded = false;  // the device copy of the fixed-base registry matches hot_key[]
};
void free_fused_plans(zkp_ctx* c);

namespace {

struct carve {
  size_t off = 0;
  size_t take(size_t bytes) {
    const size_t o = off;
    off += (bytes + 255) & ~(size_t)255;
    return o;
  }
};

int ensure_ws(zkp_ctx* c, size_t bytes) {
  // every flow passes here before it touches the workspace: a submitted host-buffer job still owns it
  if (c->job.kind) return fail(ZKP_ERR_ARG, "a submitted job is pending on this context: zkp_ctx_job_wait first");
  if (bytes <= c->ws_bytes) return ZKP_OK;
  if (c->ws_limit && bytes + bytes / 8 > c->ws_limit)
    return fail(ZKP_ERR_OOM, "the call needs a workspace of " + std::to_string(bytes + bytes / 8) + " bytes, ZKP_OPT_WS_LIMIT_BYTES allows " + std::to_string(c->ws_limit));
  if (c->capturing) return fail(ZKP_ERR_ARG, "graph capture: the workspace would grow -- run the same calls once before capturing them");
  if (c->ws) {
    HIP_TRY(hipStreamSynchronize(c->stream));
    HIP_TRY(hipFree(c->ws));
    c->ws = nullptr;
    c->ws_bytes = 0;
    ++c->ws_generation;                          // graphs captured over the old workspace are stale from here on
  }
  const size_t want = bytes + bytes / 8;
  HIP_TRY(hipMalloc(&c->ws, want));
  c->ws_bytes = want;
  return ZKP_OK;
}

// profiling: the name (as rocprofv3 prints it) of a kernel whose variant is picked at run time, filed under its timing kind
void prof_note(zkp_ctx* c, int kind, const std::string& name) {
  if (!c->profiling || c->capturing || c->prof_suspended) return;
  std::string& s = c->kernel_names[kind];
  if ((";" + s + ";").find(";" + name + ";") == std::string::npos) s += (s.empty() ? "" : ";") + name;
}
void prof_begin(zkp_ctx* c) {
  c->n_ev = 0;
  if (c->profiling && !c->capturing) for (auto& s : c->kernel_names) s.clear();
  if (c->profiling && !c->capturing) { hipEventRecord(c->ev[0], c->stream); c->ev_kind[0] = -1; c->n_ev = 1; }
}
void prof_mark(zkp_ctx* c, int kind) {
  if (c->profiling && !c->capturing && !c->prof_suspended && c->n_ev < kMaxEvents) {
    hipEventRecord(c->ev[c->n_ev], c->stream);
    c->ev_kind[c->n_ev] = kind;
    c->n_ev++;
  }
}

inline dim3 grid1(size_t n, int block) { return dim3((unsigned)((n + block - 1) / block)); }
inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

// How a call of the term path is tuned.  None of it changes a result.
//   teeth       comb-table shape for points with >= 2 cold uses (comb_tables.h): 4, or 16 when such points carry ~6+ terms
//   max_tables  upper bound of the number of such points,  max_ladder  of the points with exactly one cold use: they size
//               the workspace.  Callers that know the statement pass tight bounds; the generic bounds always hold.
//   throughput  the caller keeps many calls in flight (asynchronous _dev entry points): pick the variant with the fewest
//               instructions (one lane per comb table, batched encoder from 2,048 outputs) over the lowest latency
//   comb_min    cold uses from which a point gets a comb table (fewer: ladder).  2, or 1 for constant-time calls (see
//               zkp_ctx::ct_single_use_tables)
struct terms_cfg {
  int teeth = 4;
  uint32_t max_tables = 0xffffffffu, max_ladder = 0xffffffffu;
  bool throughput = false;
  uint32_t comb_min = 2;
  msm_map map;                     // lane -> MSM assignment of the reduce / encode kernels (fused flows: constraints by length)
  bool prehalved = false;          // the caller already wrote s / 2 mod l where the batched encoder is used (terms_batched_encode)
  stmt_job stmt;                   // fused flows: the statement's term structure (one-launch classifier, k_stmt_classify)
  bool rider_tables = false;       // stmt.pair: max_tables counts a table of multiples for every per-proof point whose terms all ride (stmt_pairs.h: stmt_rider)
};
inline bool stmt_classify_applies(const terms_cfg& k, uint32_t n_terms) {
  return k.stmt.on && n_terms >= 1024 && k.stmt.T <= STMT_MAX_TERMS && k.stmt.np <= STMT_MAX_POINTS && k.stmt.T && k.stmt.N;
}
// (whether a constant-time call lists the terms of points with >= GROUP_MIN_USES uses together and walks their tables through
//  LDS is the context's choice: zkp_ctx::grouped_comb)
inline terms_cfg terms_cfg_clamped(terms_cfg k, uint32_t n_points, uint32_t n_terms) {
  if (k.comb_min != 1) k.comb_min = 2;
  k.max_tables = std::min(k.max_tables, std::min(n_points, n_terms / k.comb_min));
  k.max_ladder = k.comb_min == 1 ? 0u : std::min(k.max_ladder, std::min(n_points, n_terms));
  if (k.teeth != 16) k.teeth = 4;
  return k;
}
constexpr uint64_t kThroughputEncodeMin = 2048;
// Point operations of n_tab table points carrying tab_terms terms: n_tab (256 + 7 TEETH) + tab_terms (256 / TEETH - 4 + 65).
// TEETH = 16 beats 4 when 48 tab_terms > 84 n_tab, i.e. from about two terms per table point on -- unless its 4 x larger
// tables (18.6 KB per point) would not fit a sensible workspace.
inline int pick_teeth(uint64_t n_tab, uint64_t tab_terms) {
  if (!n_tab || tab_terms < 2 * n_tab) return 4;
  return n_tab * comb_entries(16) * sizeof(dev_ext) <= (64ull << 30) ? 16 : 4;
}

struct terms_layout { size_t pts, part, hot, cls, list, needs, slot_of, slot_pt, gstart, gfill, comb, ladder, half, states, xs, bprod, zflag, end; };
terms_layout terms_carve(size_t start, uint32_t n_points, uint32_t n_terms, uint32_t n_msm, const terms_cfg& k) {
  carve cv;
  cv.off = start;
  terms_layout o;
  const bool split = n_terms >= 1024;
  o.pts = cv.take((size_t)n_points * sizeof(dev_affine));
  o.part = cv.take((size_t)n_terms * sizeof(dev_ext));
  o.hot = cv.take((size_t)n_points * 4);
  o.cls = cv.take(512 * 4);                                        // cls | needs | gfill are cleared by ONE memset per call
  o.needs = cv.take((size_t)n_points * 4);
  o.gfill = cv.take(split ? (size_t)n_points * 4 : 0);             // grouped comb terms: fill cursor of a point's list range
  o.list = cv.take((size_t)n_terms * 4);
  o.slot_of = cv.take(split ? (size_t)n_points * 4 : 0);
  o.slot_pt = cv.take(split ? (size_t)k.max_tables * 4 : 0);
  o.gstart = cv.take(split ? (size_t)n_points * 4 : 0);            // grouped comb terms: list range of a point
  o.comb = cv.take(split ? (size_t)k.max_tables * comb_entries(k.teeth) * sizeof(dev_ext) : 0);
  o.ladder = cv.take(split ? (((size_t)k.max_ladder + 63) / 64) * (k.stmt.pair ? 2 : 1) * LADDER_GROUP_UINT4 * sizeof(uint4) : 0);   // wave-interleaved groups of 64 tables (comb_tables.h)
  const uint32_t enc_blocks = (n_msm + ENC_BLOCK - 1) / ENC_BLOCK;
  o.half = cv.take(split ? (size_t)n_terms * 32 : 0);            // (batched encoder: reserved whenever it could be chosen)
  o.states = cv.take(split ? (size_t)n_msm * 54 * 4 : 0);
  o.xs = cv.take(split ? (size_t)n_msm * 9 * 4 : 0);
  o.bprod = cv.take(split ? (size_t)enc_blocks * 9 * 4 * 2 : 0);
  o.zflag = cv.take(split ? (size_t)n_msm : 0);
  o.end = cv.off;
  return o;
}
size_t terms_path_ws(uint32_t n_points, uint32_t n_terms, uint32_t n_msm, const terms_cfg& k = terms_cfg()) {
  return terms_carve(0, n_points, n_terms, n_msm, terms_cfg_clamped(k, n_points, n_terms)).end;
}

// whether a call of the term path encodes its outputs as 2 * H with H = sum (s_i / 2) P_i (k_encode_*), i.e. works on halved scalars
inline bool terms_batched_encode(const zkp_ctx* c, uint32_t n_terms, uint32_t n_msm, bool throughput) {
  return n_terms >= 1024 && (uint64_t)n_msm >= ((throughput && !c->batch_encode_user) ? kThroughputEncodeMin : c->batch_encode_min);
}

template <bool CT, int TEETH, int LOOKUP>
void launch_terms_split(zkp_ctx* c, dim3 grid, bool ladder, const uint8_t* d_scalars, const uint32_t* d_pidx, uint32_t n_points, const dev_ext* comb,
                        const uint32_t* slot_of, const uint32_t* class_start, const uint32_t* blk_start, const uint32_t* list,
                        const dev_affine* pts, dev_ext* ladder_rw, uint32_t max_ladder, dev_ext* part, uint32_t comb_split, const uint32_t* pair, uint32_t stmt_T) {
  // ladder blocks spread over the first half of the grid (ZKP_OPT_LADDER_INTERLEAVE), or all at the front
  uint32_t stride = 0;
  const uint32_t lb = (max_ladder + 255) / 256;
  if (ladder && lb && (c->ladder_interleave < 0 ? lb >= zkp_ctx::kInterleaveLadderBlocks : c->ladder_interleave != 0)) {
    stride = (grid.x / 2) / lb;
    // ODD: workgroups go to the 8 XCDs round robin by their index, so an even stride puts every ladder block -- the long ones, 1.65 M cycles each -- on half, a
    // stride of 16 on ONE of the XCDs, and the launch ends in a tail on 32 CUs (round 5: 524,288 CMZ proofs 34 -> 87 ms when a 3 % larger grid moved the stride
    // from 15 to 16; profiles/r05_ab_experiments.txt block e)
    if (stride % 2 == 0 && stride) --stride;
    if (stride < 2) stride = 0;
  }
  const bool split_kernel = CT && TEETH == 16 && LOOKUP == LOOKUP_XBAR && comb_split && !ladder;
  prof_note(c, ZKP_K_TERMS, std::string("k_terms_split<") + (CT ? "true" : "false") + ", " + std::to_string(TEETH) + ", " + (ladder ? "true" : "false") + ", " + std::to_string(LOOKUP) + ", " +
                            (split_kernel ? "true" : "false") + ">");
  if constexpr (CT && TEETH == 16 && LOOKUP == LOOKUP_XBAR) {
    if (comb_split && !ladder) {                                   // (the caller asks for it on narrow calls only: they have no ladder class)
      hipLaunchKernelGGL((k_terms_split<CT, TEETH, false, LOOKUP, true>), grid, dim3(256), 0, c->stream, d_scalars, d_pidx, n_points, comb, slot_of, class_start, blk_start, list,
                         c->hot_tables, pts, ladder_rw, max_ladder, part, stride, pair, stmt_T);
      return;
    }
  }
  if (ladder)
    hipLaunchKernelGGL((k_terms_split<CT, TEETH, true, LOOKUP>), grid, dim3(256), 0, c->stream, d_scalars, d_pidx, n_points, comb, slot_of, class_start, blk_start, list,
                       c->hot_tables, pts, ladder_rw, max_ladder, part, stride, pair, stmt_T);
  else
    hipLaunchKernelGGL((k_terms_split<CT, TEETH, false, LOOKUP>), grid, dim3(256), 0, c->stream, d_scalars, d_pidx, n_points, comb, slot_of, class_start, blk_start, list,
                       c->hot_tables, pts, ladder_rw, max_ladder, part, stride, pair, stmt_T);
}

// phase: everything (default), or only the part that does not look at the scalars (decode, classification, comb tables:
// the fused flows run it on the context's side stream next to the transcripts that produce the scalars), or the rest.
enum : int { PH_POINTS = 1, PH_SCALARS = 2, PH_ALL = 3 };
int msm_terms_path(zkp_ctx* c, uint32_t n_msm, const uint32_t* d_off, const uint8_t* d_scalars,
                   const uint32_t* d_pidx, const uint8_t* d_points, uint32_t n_points, uint32_t n_terms, int flags,
                   uint8_t* d_out, uint8_t* d_status8, uint32_t* d_status32, size_t ws_reserved, bool decode_all = false,
                   int phase = PH_ALL, const terms_cfg& cfg_in = terms_cfg()) {
  const terms_cfg k = terms_cfg_clamped(cfg_in, n_points, n_terms);
  const terms_layout o = terms_carve(ws_reserved, n_points, n_terms, n_msm, k);
  const bool batched_encode = terms_batched_encode(c, n_terms, n_msm, k.throughput);
  const uint32_t enc_blocks = (n_msm + ENC_BLOCK - 1) / ENC_BLOCK;
  // ensure_ws was done by the caller for ws_reserved + this much; recompute defensively
  if (o.end > c->ws_bytes) return fail(ZKP_ERR_ARG, "internal: workspace too small");
  char* base = static_cast<char*>(c->ws);
  dev_affine* pts = reinterpret_cast<dev_affine*>(base + o.pts);
  dev_ext* part = reinterpret_cast<dev_ext*>(base + o.part);
  if (n_terms >= 1024) {
    // split the terms: those on a registered fixed-base point (grouped by table) / those on a point with a comb table
    // built here (two or more cold uses) / single-use points on a ladder
    int32_t* hotmap = reinterpret_cast<int32_t*>(base + o.hot);
    uint32_t* cls = reinterpret_cast<uint32_t*>(base + o.cls);     // cnt[66] | start[67] | cursor[66] | any | table counter | block starts[65]
    uint32_t* class_cnt = cls, *class_start = cls + 80, *cursor = cls + 160, *any_hot = cls + 240, *n_slots = cls + 241, *blk_start = cls + 256;
    uint32_t* list = reinterpret_cast<uint32_t*>(base + o.list);
    uint32_t* needs = reinterpret_cast<uint32_t*>(base + o.needs);
    uint32_t* slot_of = reinterpret_cast<uint32_t*>(base + o.slot_of);
    uint32_t* slot_pt = reinterpret_cast<uint32_t*>(base + o.slot_pt);
    dev_ext* comb = reinterpret_cast<dev_ext*>(base + o.comb);
    dev_ext* ladder = reinterpret_cast<dev_ext*>(base + o.ladder);
    const uint32_t comb_min = k.comb_min;
    // (the LDS walk has fewer instructions but less independent work per lane than the masked scans: it wins once the call
    //  keeps every SIMD busy -- single kernel, 4096 CMZ proofs 590 vs 370 us, 8192: 860 vs 690, 16384: 1300 vs 1390; pipelined
    //  step: 4096 proofs -1 %, 8192 +1.6 %, 16384 +7 %, 524,288 +4.7 %)
    // Round 6, narrow calls on the latency schedule (one call in flight): the scans of a point that cannot join the grouped walk are split over a quad of lanes
    // (term_comb_split4: a third of the chain), which removes what kept the grouped walk away from such calls -- CMZ's single-use Q alone in the scan class was the
    // pole of the kernel (4096 proofs: 580 against 380 us).  With both: P on the grouped walk, Q on quads, the fixed-base blocks are the longest class (280 us).
    const bool lat_split = c->comb_split < 0 ? (!k.throughput && flags == ZKP_CT && k.teeth == 16 && n_terms >= zkp_ctx::kSplitCombTerms && n_terms < zkp_ctx::kGroupedCombTerms) : c->comb_split != 0;
    const bool group_on = (HOT_LDS_ROWS && c->ct_lookup == LOOKUP_SCAN) ? false
                          : c->grouped_comb < 0 ? (n_terms >= (k.throughput ? zkp_ctx::kWideCallTerms : zkp_ctx::kGroupedCombTerms) || lat_split) : c->grouped_comb != 0;
    const bool comb_split = lat_split && flags == ZKP_CT && k.teeth == 16 && !(HOT_LDS_ROWS && c->ct_lookup != LOOKUP_XBAR);
    const uint32_t group_min = (flags == ZKP_CT && k.teeth == 16 && group_on && k.max_tables) ? GROUP_MIN_USES : 0xffffffffu;
    // (a caller that pairs terms sized its bounds for the pairs: it only does so where the statement classifier runs)
    const bool pair_on = k.stmt.pair && flags == ZKP_VARTIME && stmt_classify_applies(k, n_terms);
    if (k.stmt.pair && !pair_on) return fail(ZKP_ERR_ARG, "paired terms outside the statement classifier");
    // (a point whose terms all ride has no use the decoder would see: such jobs decode every point, as the verifiers do anyway -- verifier.rs:87-92)
    if (pair_on && !decode_all) return fail(ZKP_ERR_ARG, "paired terms need decode_all");
    // riders whose point has no other term get a table of multiples in a 16-teeth comb table's place (stmt_pairs.h: stmt_rider; the plan's bounds count them)
    const bool rider_ok = pair_on && k.teeth == 16 && k.rider_tables;
    uint32_t* gstart = reinterpret_cast<uint32_t*>(base + o.gstart);
    uint32_t* gfill = reinterpret_cast<uint32_t*>(base + o.gfill);
    if ((phase & PH_POINTS) && stmt_classify_applies(k, n_terms)) {
      // the statement's structure gives classes, list positions and table slots arithmetically: one launch
      const size_t lanes = std::max<size_t>(std::max<size_t>(n_terms, n_points), (size_t)k.stmt.N * k.stmt.nc + 1);
      hipLaunchKernelGGL(k_stmt_classify, grid1(lanes, 256), dim3(256), 0, c->stream, k.stmt, d_points, c->hot_nreg, c->hot_reg_words, c->hot_reg_slot, comb_min, group_min,
                         k.max_tables, needs, class_start, blk_start, n_slots, slot_of, slot_pt, list, rider_ok ? 1u : 0u);
      prof_mark(c, ZKP_K_SORT);
      hipLaunchKernelGGL(k_decode_affine, grid1(n_points, 256), dim3(256), 0, c->stream, n_points, d_points, pts, decode_all ? (const uint32_t*)nullptr : needs);
      prof_mark(c, ZKP_K_DECODE);
    } else if (phase & PH_POINTS) {
    HIP_TRY(hipMemsetAsync(cls, 0, o.list - o.cls, c->stream));     // class counters, use counts, group cursors
    if (c->hot_nreg)
      hipLaunchKernelGGL(k_hot_match, grid1(n_points, 256), dim3(256), 0, c->stream, n_points, d_points, c->hot_nreg, c->hot_reg_words, c->hot_reg_slot, hotmap, any_hot);
    else
      HIP_TRY(hipMemsetAsync(hotmap, 0xff, (size_t)n_points * 4, c->stream));
    hipLaunchKernelGGL(k_use_count, grid1(n_terms, 256), dim3(256), 0, c->stream, n_terms, d_pidx, n_points, hotmap, needs);
    hipLaunchKernelGGL(k_class_count, grid1(n_terms, 256), dim3(256), 0, c->stream, n_terms, d_pidx, n_points, hotmap, needs, comb_min, group_min, class_cnt);
    prof_mark(c, ZKP_K_SORT);
    // decode: every point when the caller's semantics ask for it (a verifier rejects any allocated point that does not
    // decompress, verifier.rs:87-92), otherwise only the points whose coordinates this call uses
    hipLaunchKernelGGL(k_decode_affine, grid1(n_points, 256), dim3(256), 0, c->stream, n_points, d_points, pts, decode_all ? (const uint32_t*)nullptr : needs);
    prof_mark(c, ZKP_K_DECODE);
    hipLaunchKernelGGL(k_class_scan, dim3(1), dim3(64), 0, c->stream, class_cnt, class_start, cursor, blk_start);
    if (k.max_tables)                  // table slots, and the list ranges of the grouped points (before the scatter that fills them)
      hipLaunchKernelGGL(k_comb_slots, grid1(n_points, 256), dim3(256), 0, c->stream, n_points, needs, comb_min, group_min, k.max_tables, n_slots, slot_of, slot_pt, gstart, pts);
    hipLaunchKernelGGL(k_class_scatter, grid1(n_terms, 256), dim3(256), 0, c->stream, n_terms, d_pidx, n_points, hotmap, needs, comb_min, group_min, class_start, gstart, gfill,
                       cursor, list);
    prof_mark(c, ZKP_K_SORT);          // path A: term classification
    }
    if ((phase & PH_POINTS) && k.max_tables) {
      if (c->tables_lane < 0 ? k.throughput : c->tables_lane != 0) {
        if (k.teeth == 16 && c->pending_tr.active) {               // the flow's transcript program shares the launch
          const auto& t = c->pending_tr;
          const uint32_t tr_blocks = (t.N + TR_BLOCK / 2 - 1) / (TR_BLOCK / 2);
          // tables as producer / consumer wavefronts (comb_table_pc): workgroups of two wavefronts = two transcript blocks, or 64 tables
          if (t.steps) {
            const uint32_t rows = t.sd.n_img * 3u + ((t.sd.n_chk || (t.sd.tail >> 31)) ? 1u : 0u);
            if (rows) hipLaunchKernelGGL(k_transcript_assemble, dim3((t.N + TA_BLOCK - 1) / TA_BLOCK, rows), dim3(TA_BLOCK), 0, c->stream, t.sd, t.N, t.bufs, t.img, t.failed);
            hipLaunchKernelGGL(k_tables_chain_pc<16>, dim3((tr_blocks + 1) / 2 + (k.max_tables + 63) / 64), dim3(2 * TR_BLOCK), 0, c->stream, tr_blocks, t.sd,
                               reinterpret_cast<const uint32_t*>(t.img), t.N, t.bufs, t.ts, t.saved, n_slots, k.max_tables, slot_pt, pts, comb);
          } else
            hipLaunchKernelGGL(k_tables_transcript_pc<16>, dim3((tr_blocks + 1) / 2 + (k.max_tables + 63) / 64), dim3(2 * TR_BLOCK), 0, c->stream, tr_blocks, t.ops, t.n_ops,
                               t.tables, t.N, t.bufs, t.ts, t.saved, t.failed, t.tail, n_slots, k.max_tables, slot_pt, pts, comb);
          c->pending_tr.active = false;
          prof_note(c, ZKP_K_TABLES, t.steps ? "zkp::k_tables_chain_pc<16>" : "zkp::k_tables_transcript_pc<16>");
        } else if (k.teeth == 16) hipLaunchKernelGGL(k_comb_tables_lane<16>, grid1(k.max_tables, 256), dim3(256), 0, c->stream, n_slots, k.max_tables, slot_pt, pts, comb);
        else hipLaunchKernelGGL(k_comb_tables_lane<4>, grid1(k.max_tables, 256), dim3(256), 0, c->stream, n_slots, k.max_tables, slot_pt, pts, comb);
      } else {
        if (k.teeth == 16) hipLaunchKernelGGL(k_comb_tables<16>, grid1((size_t)k.max_tables * 4, 256), dim3(256), 0, c->stream, n_slots, k.max_tables, slot_pt, pts, comb);
        else hipLaunchKernelGGL(k_comb_tables<4>, grid1((size_t)k.max_tables * 4, 256), dim3(256), 0, c->stream, n_slots, k.max_tables, slot_pt, pts, comb);
      }
    }
    if ((phase & PH_POINTS) && k.max_tables && rider_ok)
      hipLaunchKernelGGL(k_rider_tables, grid1(k.max_tables, 256), dim3(256), 0, c->stream, n_slots, k.max_tables, slot_pt, pts, comb);
    if ((phase & PH_POINTS) && k.max_tables && c->kernel_names[ZKP_K_TABLES].find("k_tables_transcript") == std::string::npos && c->kernel_names[ZKP_K_TABLES].find("k_tables_chain") == std::string::npos)
      prof_note(c, ZKP_K_TABLES, std::string((c->tables_lane < 0 ? k.throughput : c->tables_lane != 0) ? "zkp::k_comb_tables_lane<" : "zkp::k_comb_tables<") + (k.teeth == 16 ? "16>" : "4>"));
    if (phase & PH_POINTS) prof_mark(c, ZKP_K_TABLES);        // path A: comb-table construction
    // every class starts a new block (grouped blocks take 248 terms); with comb_split the scan class has four lanes per term
    const dim3 grid((unsigned)((n_terms + XBAR_BLOCK_TERMS - 1) / XBAR_BLOCK_TERMS + 4 + HOT_SLOTS + (comb_split ? 3 * ((n_terms + 255) / 256) + 1 : 0)));
    // the outputs are encoded as 2 * H (batched encoder below): the term kernels work on s / 2 mod l
    if (batched_encode) {
      uint8_t* d_half = reinterpret_cast<uint8_t*>(base + o.half);
      if (!k.prehalved) {
        if (phase & PH_SCALARS) hipLaunchKernelGGL(k_halve_scalars, grid1(n_terms, 256), dim3(256), 0, c->stream, n_terms, d_scalars, d_half);
        d_scalars = d_half;
      }
    }
    if (phase & PH_SCALARS) {
      // (vartime calls have nothing to hide: they never scan)
      const int lookup = (!HOT_LDS_ROWS || c->ct_lookup == LOOKUP_XBAR) ? LOOKUP_XBAR : (flags == ZKP_CT ? c->ct_lookup : LOOKUP_LDS);
#define ZKP_LAUNCH_TERMS(CT_, TEETH_, LK_) launch_terms_split<CT_, TEETH_, LK_>(c, grid, k.max_ladder != 0, d_scalars, d_pidx, n_points, comb, slot_of, class_start, blk_start, list, pts, ladder, k.max_ladder, part, comb_split ? 1u : 0u, pair_on ? k.stmt.pair : (const uint32_t*)nullptr, k.stmt.T)
      if (lookup == LOOKUP_XBAR) {
        if (flags == ZKP_CT) { if (k.teeth == 16) ZKP_LAUNCH_TERMS(true, 16, LOOKUP_XBAR); else ZKP_LAUNCH_TERMS(true, 4, LOOKUP_XBAR); }
        else { if (k.teeth == 16) ZKP_LAUNCH_TERMS(false, 16, LOOKUP_XBAR); else ZKP_LAUNCH_TERMS(false, 4, LOOKUP_XBAR); }
      }
#if ZKP_HOT_W <= 6
      else if (lookup == LOOKUP_SCAN) {
        if (k.teeth == 16) ZKP_LAUNCH_TERMS(true, 16, LOOKUP_SCAN); else ZKP_LAUNCH_TERMS(true, 4, LOOKUP_SCAN);
      } else {
        if (flags == ZKP_CT) { if (k.teeth == 16) ZKP_LAUNCH_TERMS(true, 16, LOOKUP_LDS); else ZKP_LAUNCH_TERMS(true, 4, LOOKUP_LDS); }
        else { if (k.teeth == 16) ZKP_LAUNCH_TERMS(false, 16, LOOKUP_LDS); else ZKP_LAUNCH_TERMS(false, 4, LOOKUP_LDS); }
      }
#endif
#undef ZKP_LAUNCH_TERMS
    }
  } else {
    if (n_points && (phase & PH_POINTS)) hipLaunchKernelGGL(k_decode_affine, grid1(n_points, 256), dim3(256), 0, c->stream, n_points, d_points, pts, (const uint32_t*)nullptr);
    prof_mark(c, ZKP_K_DECODE);
    if (n_terms && (phase & PH_SCALARS)) hipLaunchKernelGGL(k_terms_r4, grid1(n_terms, 256), dim3(256), 0, c->stream, n_terms, d_scalars, d_pidx, n_points, pts, part);
  }
  if (!(phase & PH_SCALARS)) { HIP_TRY(hipGetLastError()); return ZKP_OK; }
  prof_mark(c, ZKP_K_TERMS);
  if (n_msm && batched_encode) {
    uint32_t* states = reinterpret_cast<uint32_t*>(base + o.states);
    uint32_t* xs = reinterpret_cast<uint32_t*>(base + o.xs);
    uint32_t* bprod = reinterpret_cast<uint32_t*>(base + o.bprod);
    uint32_t* binv = bprod + (size_t)enc_blocks * 9;
    uint8_t* zflag = reinterpret_cast<uint8_t*>(base + o.zflag);
    if (d_status8)
      hipLaunchKernelGGL(k_encode_prepare<uint8_t>, dim3(enc_blocks), dim3(ENC_BLOCK), 0, c->stream, n_msm, d_off, d_pidx, n_points, pts, part, states, xs, bprod, zflag, d_status8, k.map);
    else
      hipLaunchKernelGGL(k_encode_prepare<uint32_t>, dim3(enc_blocks), dim3(ENC_BLOCK), 0, c->stream, n_msm, d_off, d_pidx, n_points, pts, part, states, xs, bprod, zflag, d_status32, k.map);
    hipLaunchKernelGGL(k_encode_invert, dim3((enc_blocks + ENC_BLOCK - 1) / ENC_BLOCK), dim3(ENC_BLOCK), 0, c->stream, enc_blocks, bprod, binv);
    hipLaunchKernelGGL(k_encode_finish, dim3(enc_blocks), dim3(ENC_BLOCK), 0, c->stream, n_msm, d_off, d_pidx, n_points, pts, part, states, xs, binv, zflag,
                       (const uint8_t*)d_status8, (const uint32_t*)d_status32, d_out, k.map);
  } else if (n_msm) {
    if (d_status8)
      hipLaunchKernelGGL(k_reduce_encode<uint8_t>, grid1(n_msm, 256), dim3(256), 0, c->stream, n_msm, d_off, d_pidx, n_points, pts, part, d_out, d_status8, k.map);
    else
      hipLaunchKernelGGL(k_reduce_encode<uint32_t>, grid1(n_msm, 256), dim3(256), 0, c->stream, n_msm, d_off, d_pidx, n_points, pts, part, d_out, d_status32, k.map);
  }
  prof_mark(c, ZKP_K_REDUCE);
  HIP_TRY(hipGetLastError());
  return ZKP_OK;
}

// entries per virtual lane of the bucket accumulation, and the resulting upper bound of lanes per window
inline uint32_t pip_part_len(uint64_t n) { return n < (1u << 18) ? 16u : (n < (1u << 21) ? 32u : 64u); }
template <int C>
size_t pip_vmax(uint64_t n) { return (size_t)(n / pip_part_len(n)) + pip_cfg<C>::B1 + 1; }

template <int C>
uint32_t pip_tiles(uint64_t n) { return (uint32_t)((n + sort_cfg<C>::TILE - 1) / sort_cfg<C>::TILE); }

// K = number of independent MSMs of n terms each that share the run (pip_seg)
template <int C>
size_t pip_ws(uint64_t n, uint32_t K = 1) {
  using cfg = pip_cfg<C>;
  const size_t WK = (size_t)cfg::W1 * K;         // (batch, window) pairs
  carve cv;
  cv.take((size_t)K * n * sizeof(dev_niels));
  cv.take(WK * n * 4);                           // digits
  cv.take(WK * n * 4);                           // sorted
  cv.take(WK * cfg::B1 * 4 * 3);                 // hist, start, cursor
  cv.take(WK * (cfg::B1 + 1) * 4);               // vstart
  cv.take(WK * pip_tiles<C>(n) * cfg::B1 * 4);   // tile histograms / tile base offsets
  cv.take(WK * pip_vmax<C>(n) * sizeof(dev_ext)); // bucket parts
  cv.take(WK * pip_vmax<C>(n) * 4);               // vmap
  cv.take((size_t)K * 4 + 256);                  // invalid flags
  cv.take(WK * cfg::B1 * sizeof(dev_ext));       // buckets
  cv.take(WK * (cfg::B + 8) * sizeof(dev_ext) * 2);  // reduction levels (A and R)
  return cv.off;
}

// d_out [K][32]; d_status [K] words, or [K][2] with shared_flags (the caller's K zeroed flag words: bit 0 ours, bit 1 the caller's)
template <int C>
int pip_run(zkp_ctx* c, uint32_t n, const uint8_t* d_scalars, const uint8_t* d_points, uint8_t* d_out,
            uint32_t* d_status, size_t ws_reserved, uint32_t* shared_flags, const pip_seg seg = pip_seg(), int phases = 3) {
  using cfg = pip_cfg<C>;
  const uint32_t K = seg.K;
  const size_t WK = (size_t)cfg::W1 * K;
  if (WK > 65535) return fail(ZKP_ERR_ARG, "too many batches in one call");
  carve cv;
  cv.off = ws_reserved;
  char* base = static_cast<char*>(c->ws);
  dev_niels* niels = reinterpret_cast<dev_niels*>(base + cv.take((size_t)K * n * sizeof(dev_niels)));
  uint32_t* digits = reinterpret_cast<uint32_t*>(base + cv.take(WK * n * 4));
  uint32_t* sorted = reinterpret_cast<uint32_t*>(base + cv.take(WK * n * 4));
  const size_t nb = WK * cfg::B1;
  if (nb * 4 > 0xffffffffull) return fail(ZKP_ERR_ARG, "too many batches in one call");
  uint32_t* hist = reinterpret_cast<uint32_t*>(base + cv.take(nb * 4 * 3));
  uint32_t* start = hist + nb;
  uint32_t* cursor = start + nb;
  uint32_t* vstart = reinterpret_cast<uint32_t*>(base + cv.take(WK * (cfg::B1 + 1) * 4));
  const uint32_t tiles = pip_tiles<C>(n);
  uint32_t* tilehist = reinterpret_cast<uint32_t*>(base + cv.take(WK * tiles * cfg::B1 * 4));
  const uint32_t L = pip_part_len(n);
  const size_t vmax = pip_vmax<C>(n);
  dev_ext* parts = reinterpret_cast<dev_ext*>(base + cv.take(WK * vmax * sizeof(dev_ext)));
  uint32_t* vmap = reinterpret_cast<uint32_t*>(base + cv.take(WK * vmax * 4));
  uint32_t* invalid = reinterpret_cast<uint32_t*>(base + cv.take((size_t)K * 4 + 256));
  if (shared_flags) invalid = shared_flags;          // the caller's flag words, already zero (bit 0: ours; bit 1: the caller's, for status[2 b + 1])
  dev_ext* buckets = reinterpret_cast<dev_ext*>(base + cv.take(nb * sizeof(dev_ext)));
  const size_t lvl_cap = WK * (cfg::B + 8);
  dev_ext* lvlA = reinterpret_cast<dev_ext*>(base + cv.take(lvl_cap * sizeof(dev_ext) * 2));
  dev_ext* lvlR = lvlA + lvl_cap;
  if (cv.off > c->ws_bytes) return fail(ZKP_ERR_ARG, "internal: workspace too small");

  // phases: 1 = the point half of the prepare step only (the caller runs it on a side stream; needs shared_flags, zeroed before), 2 = everything after it
  if (phases == 1) {
    if (!shared_flags) return fail(ZKP_ERR_ARG, "internal: split prepare without the caller's flag words");
    hipLaunchKernelGGL((k_pip_prepare<C, 1>), dim3((n + 255) / 256, K), dim3(256), 0, c->stream, n, seg, d_scalars, d_points, niels, digits, invalid);
    HIP_TRY(hipGetLastError());
    return ZKP_OK;
  }
  if (!shared_flags) HIP_TRY(hipMemsetAsync(invalid, 0, (size_t)K * 4, c->stream));
  prof_note(c, ZKP_K_DECODE, "k_pip_prepare<" + std::to_string(C) + ">");
  if (phases == 2) hipLaunchKernelGGL((k_pip_prepare<C, 2>), dim3((n + 255) / 256, K), dim3(256), 0, c->stream, n, seg, d_scalars, d_points, niels, digits, invalid);
  else hipLaunchKernelGGL((k_pip_prepare<C, 3>), dim3((n + 255) / 256, K), dim3(256), 0, c->stream, n, seg, d_scalars, d_points, niels, digits, invalid);
  prof_mark(c, ZKP_K_DECODE);
  hipLaunchKernelGGL(k_pip_tile_hist<C>, dim3(tiles, (unsigned)WK), dim3(sort_cfg<C>::THREADS), 0, c->stream, n, digits, tilehist);
  hipLaunchKernelGGL(k_pip_tile_total, grid1(nb, 256), dim3(256), 0, c->stream, cfg::B1, tiles, (uint32_t)nb, tilehist, hist);
  hipLaunchKernelGGL(k_pip_scan, dim3((unsigned)WK), dim3(256), 0, c->stream, cfg::B1, L, hist, start, cursor, vstart);
  hipLaunchKernelGGL(k_pip_tile_base, grid1(nb, 256), dim3(256), 0, c->stream, cfg::B1, tiles, (uint32_t)nb, start, tilehist);
  hipLaunchKernelGGL(k_pip_tile_scatter<C>, dim3(tiles, (unsigned)WK), dim3(sort_cfg<C>::THREADS), 0, c->stream, n, digits, tilehist, sorted);
  prof_mark(c, ZKP_K_SORT);
  hipLaunchKernelGGL(k_pip_vmap, grid1(nb, 256), dim3(256), 0, c->stream, cfg::B1, (uint32_t)nb, (uint32_t)vmax, vstart, vmap);
  hipLaunchKernelGGL(k_pip_bucket_part, dim3((unsigned)((vmax + 255) / 256), (unsigned)WK), dim3(256), 0, c->stream, n, (uint32_t)cfg::W1, cfg::B1, L, (uint32_t)vmax,
                     start, hist, vstart, vmap, sorted, niels, parts);
  hipLaunchKernelGGL(k_pip_bucket_merge, grid1(nb * 4, 256), dim3(256), 0, c->stream, cfg::B1, (uint32_t)nb, (uint32_t)vmax, vstart, parts, buckets);
  prof_mark(c, ZKP_K_BUCKET);
  // radix-8 tree over bucket indices 0 .. B-1 (bucket B is added in k_pip_combine)
  const dev_ext* Ain = nullptr;
  const dev_ext* Rin = buckets;
  uint32_t in_stride = cfg::B1, n_in = cfg::B;
  size_t lvl_off = 0;
  int level = 0;
  int shift = 0;
  static_assert(cfg::W1 <= 64, "k_pip_combine runs the last tree level with one quad per window in one 256-lane block");
  while (n_in > 1) {
    int lg = 0;
    while ((1u << lg) < n_in) ++lg;
    const int mbits = (level == 0 && lg % 3) ? lg % 3 : 3;            // first level absorbs the odd bits
    const uint32_t m = 1u << mbits, n_out = n_in >> mbits;
    dev_ext* Aout = lvlA + lvl_off;
    dev_ext* Rout = lvlR + lvl_off;
    const uint32_t total = (uint32_t)WK * n_out;
    if (n_out == 1) {                                                 // the last level shares the launch of the Horner tails (one block per MSM)
      hipLaunchKernelGGL(k_pip_combine, dim3(K), dim3(256), 0, c->stream, cfg::W1, C, in_stride, level, (int)m, shift, Ain, Rin, Aout, invalid, d_out, d_status, shared_flags ? 1u : 0u);
      break;
    }
    hipLaunchKernelGGL(k_pip_reduce_lvl, grid1((size_t)total * 4, 256), dim3(256), 0, c->stream, n_out, total, in_stride, n_out, level, (int)m, shift,
                       Ain, Rin, Aout, Rout);
    Ain = Aout;
    Rin = Rout;
    in_stride = n_out;
    n_in = n_out;
    lvl_off += total;
    shift += mbits;
    ++level;
  }
  prof_mark(c, ZKP_K_COMBINE);
  HIP_TRY(hipGetLastError());
  return ZKP_OK;
}

// window size by problem size (bucket work ~ n * 257/c madds, reduction work ~ 2^c * 257/c adds)
int pick_c(uint64_t n) {
  // Canonical scalars are < l ~ 2^252, so the windows that matter are those below bit 253.  c = 11 divides 253 exactly
  // (23 full windows, nothing spills into a carry window); c = 13 would leave a 6-bit top window whose ~32 buckets hold
  // n/32 points each (a 2 x sqrt(n/32)-long dependent chain), c = 10 a 3-bit one.  c = 16 leaves 13 bits: fine.
  if (n < (1u << 12)) return 7;
  if (n < (1u << 13)) return 10;
  if (n < (1u << 21)) return 11;
  return 16;
}
constexpr uint64_t kSmallOptional = 192;   // below this, zkp_msm_optional uses the per-term path

// Exact table / ladder counts of a host-side CSR job: uses per point, minus the points served by a fixed-base table.
terms_cfg host_terms_cfg(const zkp_ctx* c, uint32_t n_terms, const uint32_t* pidx, const uint8_t* points, uint32_t n_points, uint32_t comb_min) {
  std::vector<uint32_t> uses(n_points, 0);
  for (uint32_t t = 0; t < n_terms; ++t) ++uses[pidx[t]];
  std::vector<std::pair<uint64_t, int>> hot;                 // (first 8 bytes, slot) of the registered encodings
  for (int sl = 0; sl < HOT_SLOTS; ++sl)
    if (!c->hot_key[sl].empty()) { uint64_t w; memcpy(&w, c->hot_key[sl].data(), 8); hot.emplace_back(w, sl); }
  std::sort(hot.begin(), hot.end());
  std::vector<char> is_hot(n_points, 0);
  bool shared = false;
  for (uint32_t p = 0; p < n_points; ++p) {
    if (!uses[p] || hot.empty()) continue;
    uint64_t w;
    memcpy(&w, points + 32 * (size_t)p, 8);
    for (auto it = std::lower_bound(hot.begin(), hot.end(), std::make_pair(w, -1)); it != hot.end() && it->first == w; ++it)
      if (memcmp(c->hot_key[it->second].data(), points + 32 * (size_t)p, 32) == 0) { is_hot[p] = 1; break; }
  }
  for (uint32_t p = 0; p < n_points; ++p) shared |= !is_hot[p] && uses[p] >= 2;
  if (comb_min == 1 && !shared) comb_min = 2;      // no shared cold point: single-use points walk the ladder (see cfg_from_terms)
  uint64_t n_tab = 0, n_lad = 0, tab_terms = 0;
  for (uint32_t p = 0; p < n_points; ++p) {
    if (!uses[p] || is_hot[p]) continue;
    if (uses[p] >= comb_min) { ++n_tab; tab_terms += uses[p]; } else ++n_lad;
  }
  terms_cfg k;
  k.comb_min = comb_min;
  k.max_tables = (uint32_t)n_tab;
  k.max_ladder = (uint32_t)n_lad;
  k.teeth = pick_teeth(n_tab, tab_terms);
  return k;
}

}  // namespace

extern "C" {

const char* zkp_last_error(void) { return g_last_error.c_str(); }
#ifdef ZKP_BUILD_TEST_HOOKS
#define ZKP_VERSION_TEXT_(w) "zkp-mi355x 0.5 gfx950 (9x29-bit limbs, v_mad_u64_u32; " #w "-bit fixed-base windows)"
#define ZKP_VERSION_TEXT(w) ZKP_VERSION_TEXT_(w)
const char* zkp_version(void) { return ZKP_VERSION_TEXT(ZKP_HOT_W) " +test-hooks"; }
#else
#define ZKP_VERSION_TEXT_(w) "zkp-mi355x 0.5 gfx950 (9x29-bit limbs, v_mad_u64_u32; " #w "-bit fixed-base windows)"
#define ZKP_VERSION_TEXT(w) ZKP_VERSION_TEXT_(w)
const char* zkp_version(void) { return ZKP_VERSION_TEXT(ZKP_HOT_W); }
#endif

int zkp_ctx_create(zkp_ctx** out, int device_id) {
  if (!out) return fail(ZKP_ERR_ARG, "zkp_ctx_create: out is NULL");
  *out = nullptr;
  int count = 0;
  if (hipGetDeviceCount(&count) != hipSuccess || count == 0) return fail(ZKP_ERR_NO_DEVICE, "no HIP device visible");
  if (device_id < 0 || device_id >= count) return fail(ZKP_ERR_ARG, "device_id out of range");
  HIP_TRY(hipSetDevice(device_id));
  hipDeviceProp_t prop;
  HIP_TRY(hipGetDeviceProperties(&prop, device_id));
  if (std::string(prop.gcnArchName).rfind("gfx950", 0) != 0)
    return fail(ZKP_ERR_NO_DEVICE, std::string("device is ") + prop.gcnArchName + ", this library is built for gfx950 only");
  zkp_ctx* c = new zkp_ctx();
  c->uid = g_ctx_next_id.fetch_add(1);
  c->device = device_id;
  if (hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking) != hipSuccess) { delete c; return fail(ZKP_ERR_HIP, "hipStreamCreate failed"); }
  c->stream = c->own_stream;
  for (auto& e : c->ev)
    if (hipEventCreate(&e) != hipSuccess) { delete c; return fail(ZKP_ERR_HIP, "hipEventCreate failed"); }
  { std::lock_guard<std::mutex> lk(g_ctx_mu); g_live_ctx.insert(c); }
  *out = c;
  return ZKP_OK;
}

void zkp_ctx_destroy(zkp_ctx* c) {
  if (!c) return;
  { std::lock_guard<std::mutex> lk(g_ctx_mu); g_live_ctx.erase(c); }
  hipSetDevice(c->device);
  if (c->capturing) { hipGraph_t g = nullptr; hipStreamEndCapture(c->stream, &g); if (g) hipGraphDestroy(g); c->capturing = false; }
  hipStreamSynchronize(c->stream);
  if (c->job.done) hipEventDestroy(c->job.done);
  if (c->job.copied) hipEventDestroy(c->job.copied);
  for (auto& e : c->job.tev) if (e) hipEventDestroy(e);
  if (c->job.pin) hipHostFree(c->job.pin);
  if (c->ws) hipFree(c->ws);
  if (c->hot_tables) hipFree(c->hot_tables);
  if (c->hot_reg_words) hipFree(c->hot_reg_words);
  if (c->hot_reg_slot) hipFree(c->hot_reg_slot);
  if (c->hot_scratch) hipFree(c->hot_scratch);
#ifdef ZKP_BUILD_TEST_HOOKS
  if (c->wave_cycles) { uint64_t* z = nullptr; (void)hipMemcpyToSymbol(HIP_SYMBOL(g_wave_cycles), &z, sizeof(z)); hipFree(c->wave_cycles); }
#endif
  free_fused_plans(c);
  if (c->side_stream) hipStreamDestroy(c->side_stream);
  if (c->ev_fork) hipEventDestroy(c->ev_fork);
  if (c->ev_join) hipEventDestroy(c->ev_join);
  for (auto& e : c->ev) if (e) hipEventDestroy(e);
  if (c->own_stream) hipStreamDestroy(c->own_stream);
  delete c;
}

int zkp_ctx_set_stream(zkp_ctx* c, void* s) {
  if (!c) return fail(ZKP_ERR_ARG, "ctx is NULL");
  c->stream = s ? static_cast<hipStream_t>(s) : c->own_stream;
  return ZKP_OK;
}
int zkp_ctx_set_option(zkp_ctx* c, int option, uint64_t value) {
  if (!c) return fail(ZKP_ERR_ARG, "ctx is NULL");
  switch (option) {
    case ZKP_OPT_BATCH_ENCODE_MIN: c->batch_encode_min = value; c->batch_encode_user = true; return ZKP_OK;
    case ZKP_OPT_CT_SINGLE_USE_TABLES: c->ct_single_use_tables = value == ~0ull ? -1 : (value ? 1 : 0); return ZKP_OK;
    case ZKP_OPT_DEV_OVERLAP: c->dev_overlap = value != 0 && value != ~0ull; c->dev_latency = value == 2; return ZKP_OK;
#ifdef ZKP_BUILD_TEST_HOOKS
    case ZKP_TESTOPT_GENERIC_CLASSIFIER: c->stmt_classify = value == 0; return ZKP_OK;
    case ZKP_TESTOPT_DUMMY_LAUNCHES: c->debug_dummy_launches = (int)std::min<uint64_t>(value, 1000); return ZKP_OK;
    case ZKP_TESTOPT_WAVE_CYCLES: {
      HIP_TRY(hipSetDevice(c->device));
      HIP_TRY(hipStreamSynchronize(c->stream));
      uint64_t* p = nullptr;
      uint32_t cap = 0;
      if (value) {
        if (!c->wave_cycles) HIP_TRY(hipMalloc(&c->wave_cycles, sizeof(uint64_t) * zkp_ctx::kWaveCyclesCap));
        HIP_TRY(hipMemset(c->wave_cycles, 0, sizeof(uint64_t) * zkp_ctx::kWaveCyclesCap));
        p = c->wave_cycles;
        cap = zkp_ctx::kWaveCyclesCap;
      }
      HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(g_wave_cycles), &p, sizeof(p)));
      HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(g_wave_cycles_cap), &cap, sizeof(cap)));
      return ZKP_OK;
    }
#endif
    case ZKP_OPT_FUSE_TABLES_TRANSCRIPT: c->fuse_tables_transcript = value == ~0ull ? -1 : (value ? 1 : 0); return ZKP_OK;
    case ZKP_OPT_TABLES_LANE: c->tables_lane = value == ~0ull ? -1 : (value ? 1 : 0); return ZKP_OK;
    case ZKP_OPT_GROUPED_COMB: c->grouped_comb = value == ~0ull ? -1 : (value ? 1 : 0); return ZKP_OK;
    case ZKP_OPT_JOINT_LADDER: c->joint_ladder = value != 0; c->rider_tables = value != 2; return ZKP_OK;
    case ZKP_OPT_COMB_SPLIT: c->comb_split = value == ~0ull ? -1 : (value ? 1 : 0); return ZKP_OK;
    case ZKP_OPT_CT_LOOKUP:
      if (value == ~0ull) value = 0;
      if (value > 2 || (value && !HOT_LDS_ROWS)) return fail(ZKP_ERR_ARG, "ZKP_OPT_CT_LOOKUP: 0 (lane crossbar), 1 (masked scans), 2 (LDS rows); 1 and 2 exist for 6-bit fixed-base windows only");
      c->ct_lookup = (int)value; return ZKP_OK;
    case ZKP_OPT_LADDER_INTERLEAVE: c->ladder_interleave = value == ~0ull ? -1 : value != 0; return ZKP_OK;
    case ZKP_OPT_JOB_DEFER_D2H: c->defer_d2h = value == ~0ull ? -1 : (value ? 1 : 0); return ZKP_OK;
    case ZKP_OPT_SYNC_SCHEDULE: c->sync_throughput = value != 0 && value != ~0ull; return ZKP_OK;
    case ZKP_OPT_WS_LIMIT_BYTES: c->ws_limit = value == ~0ull ? 0 : (size_t)value; return ZKP_OK;
    case ZKP_OPT_EACH_STRAUS:
    {
      const bool wins = value > 0x200 && value <= 0x200 + 64 && ((value - 0x200) & (value - 0x201)) == 0;
      if (value != ~0ull && value > 8 && !wins)
        return fail(ZKP_ERR_ARG, "ZKP_OPT_EACH_STRAUS: 0 (off), 1 .. 8 lanes per proof, 0x200 + (1, 2, 4 .. 64) window parts per proof, or UINT64_MAX (default)");
      c->each_straus = value != 0;
      c->each_straus_lanes = (value == ~0ull || value == 0 || value > 8) ? 0u : (uint32_t)value;
      c->each_straus_wins = wins ? (uint32_t)(value - 0x200) : 0u;
      return ZKP_OK;
    }
    case ZKP_OPT_TRANSCRIPT_STEPS: c->tr_steps = value != 0; return ZKP_OK;
    case ZKP_OPT_TRANSCRIPT_LANES:
      if (value != ~0ull && value != 1 && value != 2) return fail(ZKP_ERR_ARG, "ZKP_OPT_TRANSCRIPT_LANES: 1, 2 or UINT64_MAX");
      c->tr_lanes = value == ~0ull ? -1 : (int)value;
      return ZKP_OK;
    case ZKP_OPT_COMB_TEETH:
      if (value != 4 && value != 16) return fail(ZKP_ERR_ARG, "ZKP_OPT_COMB_TEETH must be 4 or 16");
      c->comb_teeth = (int)value;
      return ZKP_OK;
    default: return fail(ZKP_ERR_ARG, "unknown option");
  }
}
int zkp_ctx_synchronize(zkp_ctx* c) {
  if (!c) return fail(ZKP_ERR_ARG, "ctx is NULL");
  HIP_TRY(hipStreamSynchronize(c->stream));
  return ZKP_OK;
}
int zkp_ctx_set_profiling(zkp_ctx* c, int enabled) {
  if (!c) return fail(ZKP_ERR_ARG, "ctx is NULL");
  c->profiling = enabled != 0;
  return ZKP_OK;
}
struct zkp_graph {
  hipGraph_t graph = nullptr;
  hipGraphExec_t exec = nullptr;
  int device = 0;
  zkp_ctx* ctx = nullptr;                        // the context whose workspace / plans the recorded kernels point into
  uint64_t ctx_uid = 0;
  uint64_t ws_generation = 0, plans_generation = 0;
};

int zkp_ctx_capture_begin(zkp_ctx* c) {
  if (!c) return fail(ZKP_ERR_ARG, "ctx is NULL");
  if (c->capturing) return fail(ZKP_ERR_ARG, "capture already in progress");
  HIP_TRY(hipSetDevice(c->device));
  HIP_TRY(hipStreamBeginCapture(c->stream, hipStreamCaptureModeThreadLocal));
  c->capturing = true;
  return ZKP_OK;
}
int zkp_ctx_capture_end(zkp_ctx* c, zkp_graph** out) {
  if (!c || !out) return fail(ZKP_ERR_ARG, "NULL pointer");
  *out = nullptr;
  if (!c->capturing) return fail(ZKP_ERR_ARG, "no capture in progress");
  c->capturing = false;
  hipGraph_t g = nullptr;
  HIP_TRY(hipStreamEndCapture(c->stream, &g));
  if (!g) return fail(ZKP_ERR_HIP, "hipStreamEndCapture returned no graph (a call inside the capture failed)");
  hipGraphExec_t e = nullptr;
  const hipError_t rc = hipGraphInstantiate(&e, g, nullptr, nullptr, 0);
  if (rc != hipSuccess) { hipGraphDestroy(g); return fail(ZKP_ERR_HIP, std::string("hipGraphInstantiate: ") + hipGetErrorString(rc)); }
  zkp_graph* zg = new zkp_graph();
  zg->graph = g;
  zg->exec = e;
  zg->device = c->device;
  zg->ctx = c;
  zg->ctx_uid = c->uid;
  zg->ws_generation = c->ws_generation;
  zg->plans_generation = c->plans_generation;
  *out = zg;
  return ZKP_OK;
}
int zkp_ctx_capture_abort(zkp_ctx* c) {
  if (!c) return fail(ZKP_ERR_ARG, "ctx is NULL");
  if (!c->capturing) return ZKP_OK;
  c->capturing = false;
  c->pending_tr.offered = c->pending_tr.active = false;
  hipGraph_t g = nullptr;
  const hipError_t rc = hipStreamEndCapture(c->stream, &g);       // (an invalidated capture reports an error here: the stream leaves capture mode either way)
  if (g) hipGraphDestroy(g);
  (void)hipGetLastError();
  (void)rc;
  return ZKP_OK;
}
int zkp_graph_launch(zkp_graph* g, zkp_ctx* c) {
  if (!g || !c) return fail(ZKP_ERR_ARG, "NULL pointer");
  if (g->ctx != c) return fail(ZKP_ERR_ARG, "a graph can only be launched on the context it was captured on (it records that context's workspace addresses)");
  { std::lock_guard<std::mutex> lk(g_ctx_mu); if (!g_live_ctx.count(c)) return fail(ZKP_ERR_ARG, "the graph's context has been destroyed"); }
  if (g->ctx_uid != c->uid) return fail(ZKP_ERR_ARG, "the graph's context has been destroyed (another context now lives at its address)");
  if (g->ws_generation != c->ws_generation)
    return fail(ZKP_ERR_ARG, "stale graph: the context's workspace was reallocated by a larger call after the capture -- capture again");
  if (g->plans_generation != c->plans_generation)
    return fail(ZKP_ERR_ARG, "stale graph: the context's statement plans were flushed after the capture -- capture again");
  if (c->capturing) return fail(ZKP_ERR_ARG, "a capture is in progress on this context");
  HIP_TRY(hipSetDevice(c->device));
  HIP_TRY(hipGraphLaunch(g->exec, c->stream));
  return ZKP_OK;
}
void zkp_graph_destroy(zkp_graph* g) {
  if (!g) return;
  hipSetDevice(g->device);
  if (g->exec) hipGraphExecDestroy(g->exec);
  if (g->graph) hipGraphDestroy(g->graph);
  delete g;
}

int zkp_ctx_last_timing(zkp_ctx* c, float* kernel_ms, float* total_ms) {
  if (!c) return fail(ZKP_ERR_ARG, "ctx is NULL");
  for (float& k : c->kernel_ms) k = 0;
  c->total_ms = 0;
  if (c->n_ev >= 2) {
    HIP_TRY(hipEventSynchronize(c->ev[c->n_ev - 1]));
    for (int i = 1; i < c->n_ev; ++i) {
      float ms = 0;
      HIP_TRY(hipEventElapsedTime(&ms, c->ev[i - 1], c->ev[i]));
      if (c->ev_kind[i] >= 0) c->kernel_ms[c->ev_kind[i]] += ms;
    }
    HIP_TRY(hipEventElapsedTime(&c->total_ms, c->ev[0], c->ev[c->n_ev - 1]));
  }
  if (kernel_ms) memcpy(kernel_ms, c->kernel_ms, sizeof(c->kernel_ms));
  if (total_ms) *total_ms = c->total_ms;
  return ZKP_K_COUNT;
}

int zkp_ctx_last_kernels(zkp_ctx* c, int kind, char* buf, size_t cap) {
  if (!c || !buf || !cap || kind < 0 || kind >= ZKP_K_COUNT) return fail(ZKP_ERR_ARG, "bad argument");
  const std::string& s = c->kernel_names[kind];
  const size_t n = std::min(s.size(), cap - 1);
  memcpy(buf, s.data(), n);
  buf[n] = 0;
  return (int)s.size();
}

int zkp_ctx_prepare_fixed_points(zkp_ctx* c, uint32_t n, const uint8_t* encodings) {
  if (!c) return fail(ZKP_ERR_ARG, "ctx is NULL");
  if (n == 0) return ZKP_OK;
  if (!encodings) return fail(ZKP_ERR_ARG, "NULL pointer");
  HIP_TRY(hipSetDevice(c->device));
  if (!c->hot_tables) {
    HIP_TRY(hipMalloc(&c->hot_tables, sizeof(dev_niels) * HOT_SLOT_NIELS * HOT_SLOTS));
    HIP_TRY(hipMalloc(&c->hot_reg_words, 32 * HOT_SLOTS));
    HIP_TRY(hipMalloc(&c->hot_reg_slot, 4 * HOT_SLOTS));
    HIP_TRY(hipMalloc(&c->hot_scratch, (size_t)HOT_SLOTS * (32 + sizeof(dev_affine) + 4 + sizeof(dev_ext) * HOT_WINDOWS)));
  }
  // which encodings are new?  (the last HOT_SLOTS distinct ones win if more are given)
  ++c->hot_tick;
  std::vector<std::string> fresh;
  for (uint32_t i = 0; i < n; ++i) {
    const std::string key(reinterpret_cast<const char*>(encodings + 32 * (size_t)i), 32);
    bool found = false;
    for (int sl = 0; sl < HOT_SLOTS; ++sl)
      if (c->hot_key[sl] == key) { c->hot_used[sl] = c->hot_tick; found = true; break; }
    if (!found && std::find(fresh.begin(), fresh.end(), key) == fresh.end()) fresh.push_back(key);
  }
  if (fresh.size() > (size_t)HOT_SLOTS) fresh.resize(HOT_SLOTS);
  if (fresh.empty() && c->hot_registry_uploaded) return ZKP_OK;      // every point has its table already: nothing to launch, nothing to wait for
  if (c->job.kind) return fail(ZKP_ERR_ARG, "a submitted job is pending on this context: zkp_ctx_job_wait first");
  c->hot_registry_uploaded = false;
  if (!fresh.empty()) {
    const uint32_t nh = (uint32_t)fresh.size();
    char* sc = c->hot_scratch;
    uint8_t* d_enc = reinterpret_cast<uint8_t*>(sc);
    dev_affine* d_aff = reinterpret_cast<dev_affine*>(sc + 32 * HOT_SLOTS);
    uint32_t* d_slots = reinterpret_cast<uint32_t*>(sc + (32 + sizeof(dev_affine)) * HOT_SLOTS);
    dev_ext* d_bases = reinterpret_cast<dev_ext*>(sc + (32 + sizeof(dev_affine) + 4) * HOT_SLOTS);
    std::vector<uint8_t> h_enc(32 * (size_t)nh);
    for (uint32_t i = 0; i < nh; ++i) memcpy(h_enc.data() + 32 * (size_t)i, fresh[i].data(), 32);
    HIP_TRY(hipMemcpyAsync(d_enc, h_enc.data(), h_enc.size(), hipMemcpyHostToDevice, c->stream));
    hipLaunchKernelGGL(k_decode_affine, grid1(nh, 256), dim3(256), 0, c->stream, nh, d_enc, d_aff, (const uint32_t*)nullptr);
    std::vector<dev_affine> h_aff(nh);
    HIP_TRY(hipMemcpyAsync(h_aff.data(), d_aff, sizeof(dev_affine) * nh, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    // slots: free first, then least recently used; undecodable encodings get no table (the generic path reports them)
    std::vector<uint32_t> h_slots(nh, 0);
    std::vector<uint32_t> keep;
    for (uint32_t i = 0; i < nh; ++i) {
      if (!h_aff[i].valid) continue;
      int best = -1;
      for (int sl = 0; sl < HOT_SLOTS; ++sl) {
        if (c->hot_used[sl] == c->hot_tick) continue;                  // touched by this very call
        if (c->hot_key[sl].empty()) { best = sl; break; }
        if (best < 0 || c->hot_used[sl] < c->hot_used[best]) best = sl;
      }
      if (best < 0) break;
      c->hot_key[best] = fresh[i];
      c->hot_used[best] = c->hot_tick;
      h_slots[i] = (uint32_t)best;
      keep.push_back(i);
    }
    if (!keep.empty()) {
      // compact the kept points to the front (decoded form + slot)
      std::vector<dev_affine> k_aff(keep.size());
      std::vector<uint32_t> k_slots(keep.size());
      for (size_t j = 0; j < keep.size(); ++j) { k_aff[j] = h_aff[keep[j]]; k_slots[j] = h_slots[keep[j]]; }
      const uint32_t nk = (uint32_t)keep.size();
      HIP_TRY(hipMemcpyAsync(d_aff, k_aff.data(), sizeof(dev_affine) * nk, hipMemcpyHostToDevice, c->stream));
      HIP_TRY(hipMemcpyAsync(d_slots, k_slots.data(), 4 * nk, hipMemcpyHostToDevice, c->stream));
      hipLaunchKernelGGL(k_hot_bases, grid1(nk, 64), dim3(64), 0, c->stream, nk, d_aff, d_bases);
      dev_ext* d_mult = nullptr;                                       // the multiples before normalisation (k_hot_rows)
      HIP_TRY(hipMalloc(&d_mult, sizeof(dev_ext) * (size_t)nk * HOT_WINDOWS * HOT_HALF));
      hipLaunchKernelGGL(k_hot_rows, grid1((size_t)nk * HOT_WINDOWS, 64), dim3(64), 0, c->stream, nk, d_slots, d_bases, d_mult, c->hot_tables);
      const hipError_t launch_err = hipGetLastError();
      const hipError_t sync_err = hipStreamSynchronize(c->stream);
      (void)hipFree(d_mult);
      HIP_TRY(launch_err);
      HIP_TRY(sync_err);
    }
  }
  // densely packed registry for k_hot_match
  std::vector<uint32_t> words;
  std::vector<int32_t> slots;
  for (int sl = 0; sl < HOT_SLOTS; ++sl)
    if (!c->hot_key[sl].empty()) {
      uint32_t w[8];
      memcpy(w, c->hot_key[sl].data(), 32);
      words.insert(words.end(), w, w + 8);
      slots.push_back(sl);
    }
  c->hot_nreg = (uint32_t)slots.size();
  if (c->hot_nreg) {
    HIP_TRY(hipMemcpyAsync(c->hot_reg_words, words.data(), 32 * (size_t)c->hot_nreg, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipMemcpyAsync(c->hot_reg_slot, slots.data(), 4 * (size_t)c->hot_nreg, hipMemcpyHostToDevice, c->stream));
  }
  HIP_TRY(hipStreamSynchronize(c->stream));
  c->hot_registry_uploaded = true;
  return ZKP_OK;
}

int zkp_msm_many_dev(zkp_ctx* c, uint32_t n_msm, const uint32_t* d_off, const uint8_t* d_scalars,
                     const uint32_t* d_pidx, const uint8_t* d_points, uint32_t n_points, uint32_t n_terms,
                     int flags, uint8_t* d_out, uint8_t* d_status) {
  if (!c) return fail(ZKP_ERR_ARG, "ctx is NULL");
  if (flags != ZKP_CT && flags != ZKP_VARTIME) return fail(ZKP_ERR_ARG, "flags must be ZKP_CT or ZKP_VARTIME");
  if (n_msm == 0) return ZKP_OK;
  if (!d_off || !d_out || !d_status) return fail(ZKP_ERR_ARG, "NULL device pointer");
  if (n_terms && (!d_scalars || !d_pidx || !d_points || n_points == 0)) return fail(ZKP_ERR_ARG, "terms without scalars/points");
  if (!aligned16(d_scalars) || !aligned16(d_points) || !aligned16(d_out)) return fail(ZKP_ERR_ARG, "device buffers must be 16-byte aligned");
  HIP_TRY(hipSetDevice(c->device));
  terms_cfg k;                               // the device arrays are not inspected on the host: generic bounds
  k.teeth = c->comb_teeth;
  k.throughput = true;
  k.comb_min = flags == ZKP_CT ? c->ct_comb_min(true, n_terms) : 2u;
  const int rc = ensure_ws(c, terms_path_ws(n_points, n_terms, n_msm, k));
  if (rc) return rc;
  prof_begin(c);
  return msm_terms_path(c, n_msm, d_off, d_scalars, d_pidx, d_points, n_points, n_terms, flags, d_out, d_status, nullptr, 0, false, PH_ALL, k);
}

int zkp_msm_many(zkp_ctx* c, uint32_t n_msm, const uint32_t* off, const uint8_t* scalars, const uint32_t* pidx,
                 const uint8_t* points, uint32_t n_points, int flags, uint8_t* out, uint8_t* status) {
  if (!c) return fail(ZKP_ERR_ARG, "ctx is NULL");
  if (flags != ZKP_CT && flags != ZKP_VARTIME) return fail(ZKP_ERR_ARG, "flags must be ZKP_CT or ZKP_VARTIME");
  if (n_msm == 0) return ZKP_OK;
  if (!off || !out || !status) return fail(ZKP_ERR_ARG, "NULL pointer");
  if (off[0] != 0) return fail(ZKP_ERR_ARG, "off[0] must be 0");
  for (uint32_t i = 0; i < n_msm; ++i)
    if (off[i + 1] < off[i]) return fail(ZKP_ERR_ARG, "off must be non-decreasing");
  const uint32_t n_terms = off[n_msm];
  if (n_terms && (!scalars || !pidx || !points || n_points == 0)) return fail(ZKP_ERR_ARG, "terms without scalars/points");
  for (uint32_t t = 0; t < n_terms; ++t)
    if (pidx[t] >= n_points) return fail(ZKP_ERR_ARG, "pidx out of range");
  HIP_TRY(hipSetDevice(c->device));
  const terms_cfg k = n_terms >= 1024 ? host_terms_cfg(c, n_terms, pidx, points, n_points, flags == ZKP_CT ? c->ct_comb_min(false, n_terms) : 2u) : terms_cfg();
  carve cv;
  const size_t o_off = cv.take((size_t)(n_msm + 1) * 4);
  const size_t o_sc = cv.take((size_t)n_terms * 32);
  const size_t o_pidx = cv.take((size_t)n_terms * 4);
  const size_t o_pts = cv.take((size_t)n_points * 32);
  const size_t o_out = cv.take((size_t)n_msm * 32);
  const size_t o_st = cv.take((size_t)n_msm);
  const size_t reserved = cv.off;
  int rc = ensure_ws(c, reserved + terms_path_ws(n_points, n_terms, n_msm, k));
  if (rc) return rc;
  char* base = static_cast<char*>(c->ws);
  HIP_TRY(hipMemcpyAsync(base + o_off, off, (size_t)(n_msm + 1) * 4, hipMemcpyHostToDevice, c->stream));
  if (n_terms) {
    HIP_TRY(hipMemcpyAsync(base + o_sc, scalars, (size_t)n_terms * 32, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipMemcpyAsync(base + o_pidx, pidx, (size_t)n_terms * 4, hipMemcpyHostToDevice, c->stream));
  }
  if (n_points) HIP_TRY(hipMemcpyAsync(base + o_pts, points, (size_t)n_points * 32, hipMemcpyHostToDevice, c->stream));
  prof_begin(c);
  rc = msm_terms_path(c, n_msm, reinterpret_cast<uint32_t*>(base + o_off), reinterpret_cast<uint8_t*>(base + o_sc),
                      reinterpret_cast<uint32_t*>(base + o_pidx), reinterpret_cast<uint8_t*>(base + o_pts), n_points,
                      n_terms, flags, reinterpret_cast<uint8_t*>(base + o_out), reinterpret_cast<uint8_t*>(base + o_st), nullptr, reserved, false, PH_ALL, k);
  if (rc) return rc;
  HIP_TRY(hipMemcpyAsync(out, base + o_out, (size_t)n_msm * 32, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipMemcpyAsync(status, base + o_st, (size_t)n_msm, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  return ZKP_OK;
}

// shared_flags (optional, Pippenger sizes only): a zeroed device word that replaces the path's own decode-failure flag (bit 0) and
// whose bit 1 the caller may have set; the last kernel then writes status[0] = bit 0 and status[1] = bit 1 (no memsets)
static int msm_optional_impl(zkp_ctx* c, uint64_t n, const uint8_t* d_scalars, const uint8_t* d_points,
                             uint8_t* d_out, uint32_t* d_status, size_t reserved, uint32_t* shared_flags = nullptr, int phases = 3) {
  if (n <= kSmallOptional) {
    if (shared_flags) return fail(ZKP_ERR_ARG, "internal: shared flags with a small MSM");
    carve cv;
    cv.off = reserved;
    const size_t o_pidx = cv.take((size_t)(n + 1) * 4);
    const size_t o_off = cv.take(256);
    const size_t inner = cv.off;
    int rc = ensure_ws(c, inner + terms_path_ws((uint32_t)n, (uint32_t)n, 1));
    if (rc) return rc;
    char* base = static_cast<char*>(c->ws);
    uint32_t* pidx = reinterpret_cast<uint32_t*>(base + o_pidx);
    uint32_t* off = reinterpret_cast<uint32_t*>(base + o_off);
    hipLaunchKernelGGL(k_iota_single_msm, grid1(n + 1, 256), dim3(256), 0, c->stream, (uint32_t)n, pidx, off);
    return msm_terms_path(c, 1, off, d_scalars, pidx, d_points, (uint32_t)n, (uint32_t)n, ZKP_VARTIME, d_out, nullptr, d_status, inner);
  }
  if (n > 0x7fffffffull) return fail(ZKP_ERR_ARG, "n too large (max 2^31-1 terms per call)");
  const int cbits = pick_c(n);
  size_t need = 0;
  switch (cbits) {
    case 7: need = pip_ws<7>(n); break;
    case 10: need = pip_ws<10>(n); break;
    case 11: need = pip_ws<11>(n); break;
    default: need = pip_ws<16>(n); break;
  }
  int rc = ensure_ws(c, reserved + need);
  if (rc) return rc;
  switch (cbits) {
    case 7: return pip_run<7>(c, (uint32_t)n, d_scalars, d_points, d_out, d_status, reserved, shared_flags, pip_seg(), phases);
    case 10: return pip_run<10>(c, (uint32_t)n, d_scalars, d_points, d_out, d_status, reserved, shared_flags, pip_seg(), phases);
    case 11: return pip_run<11>(c, (uint32_t)n, d_scalars, d_points, d_out, d_status, reserved, shared_flags, pip_seg(), phases);
    default: return pip_run<16>(c, (uint32_t)n, d_scalars, d_points, d_out, d_status, reserved, shared_flags, pip_seg(), phases);
  }
}

int zkp_msm_optional_dev(zkp_ctx* c, uint64_t n, const uint8_t* d_scalars, const uint8_t* d_points,
                         uint8_t* d_out_point, uint32_t* d_status) {
  if (!c) return fail(ZKP_ERR_ARG, "ctx is NULL");
  if (!d_out_point || !d_status) return fail(ZKP_ERR_ARG, "NULL output pointer");
  if (n && (!d_scalars || !d_points)) return fail(ZKP_ERR_ARG, "NULL input pointer");
  if (!aligned16(d_scalars) || !aligned16(d_points) || !aligned16(d_out_point)) return fail(ZKP_ERR_ARG, "device buffers must be 16-byte aligned");
  HIP_TRY(hipSetDevice(c->device));
  prof_begin(c);
  return msm_optional_impl(c, n, d_scalars, d_points, d_out_point, d_status, 0);
}

int zkp_msm_optional(zkp_ctx* c, uint64_t n, const uint8_t* scalars, const uint8_t* points, uint8_t out_point[32], int* status) {
  if (!c) return fail(ZKP_ERR_ARG, "ctx is NULL");
  if (!out_point || !status) return fail(ZKP_ERR_ARG, "NULL output pointer");
  if (n && (!scalars || !points)) return fail(ZKP_ERR_ARG, "NULL input pointer");
  HIP_TRY(hipSetDevice(c->device));
  carve cv;
  const size_t o_sc = cv.take((size_t)n * 32 + 32);
  const size_t o_pts = cv.take((size_t)n * 32 + 32);
  const size_t o_out = cv.take(32);
  const size_t o_st = cv.take(4);
  const size_t reserved = cv.off;
  // size the workspace once, up front (inputs are copied before the kernels are enqueued)
  size_t need = 0;
  if (n <= kSmallOptional) {
    need = 1024 + (n + 1) * 4 + terms_path_ws((uint32_t)n, (uint32_t)n, 1);
  } else {
    switch (pick_c(n)) {
      case 7: need = pip_ws<7>(n); break;
      case 10: need = pip_ws<10>(n); break;
      case 11: need = pip_ws<11>(n); break;
        default: need = pip_ws<16>(n); break;
    }
  }
  int rc = ensure_ws(c, reserved + need);
  if (rc) return rc;
  char* base = static_cast<char*>(c->ws);
  if (n) {
    HIP_TRY(hipMemcpyAsync(base + o_sc, scalars, (size_t)n * 32, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipMemcpyAsync(base + o_pts, points, (size_t)n * 32, hipMemcpyHostToDevice, c->stream));
  }
  prof_begin(c);
  rc = msm_optional_impl(c, n, reinterpret_cast<uint8_t*>(base + o_sc), reinterpret_cast<uint8_t*>(base + o_pts),
                         reinterpret_cast<uint8_t*>(base + o_out), reinterpret_cast<uint32_t*>(base + o_st), reserved);
  if (rc) return rc;
  uint32_t st = 1;
  HIP_TRY(hipMemcpyAsync(out_point, static_cast<char*>(c->ws) + o_out, 32, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipMemcpyAsync(&st, static_cast<char*>(c->ws) + o_st, 4, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  *status = (int)st;
  return ZKP_OK;
}

int zkp_decode_check(zkp_ctx* c, uint64_t n, const uint8_t* points, uint8_t* status, uint8_t* xyzt) {
  if (!c) return fail(ZKP_ERR_ARG, "ctx is NULL");
  if (n == 0) return ZKP_OK;
  if (!points || !status) return fail(ZKP_ERR_ARG, "NULL pointer");
  if (n > 0x7fffffffull) return fail(ZKP_ERR_ARG, "n too large");
  HIP_TRY(hipSetDevice(c->device));
  carve cv;
  const size_t o_pts = cv.take((size_t)n * 32);
  const size_t o_st = cv.take((size_t)n);
  const size_t o_xyzt = cv.take(xyzt ? (size_t)n * 128 : 0);
  const int rc = ensure_ws(c, cv.off);
  if (rc) return rc;
  char* base = static_cast<char*>(c->ws);
  HIP_TRY(hipMemcpyAsync(base + o_pts, points, (size_t)n * 32, hipMemcpyHostToDevice, c->stream));
  prof_begin(c);
  hipLaunchKernelGGL(k_decode_check, grid1(n, 256), dim3(256), 0, c->stream, (uint32_t)n, reinterpret_cast<uint8_t*>(base + o_pts),
                     reinterpret_cast<uint8_t*>(base + o_st), xyzt ? reinterpret_cast<uint8_t*>(base + o_xyzt) : nullptr);
  prof_mark(c, ZKP_K_DECODE);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipMemcpyAsync(status, base + o_st, (size_t)n, hipMemcpyDeviceToHost, c->stream));
  if (xyzt) HIP_TRY(hipMemcpyAsync(xyzt, base + o_xyzt, (size_t)n * 128, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  return ZKP_OK;
}

#ifdef ZKP_BUILD_TEST_HOOKS
int zkp_debug_quad_selftest(zkp_ctx* c, uint32_t n, const uint8_t* pairs, uint8_t* out) {
  if (!c || !pairs || !out) return fail(ZKP_ERR_ARG, "NULL pointer");
  if (n == 0) return ZKP_OK;
  HIP_TRY(hipSetDevice(c->device));
  carve cv;
  const size_t o_in = cv.take((size_t)n * 64);
  const size_t o_out = cv.take((size_t)n * 128);
  const int rc = ensure_ws(c, cv.off);
  if (rc) return rc;
  char* base = static_cast<char*>(c->ws);
  HIP_TRY(hipMemcpyAsync(base + o_in, pairs, (size_t)n * 64, hipMemcpyHostToDevice, c->stream));
  hipLaunchKernelGGL(k_debug_quad, grid1((size_t)n * 4, 256), dim3(256), 0, c->stream, n, reinterpret_cast<uint8_t*>(base + o_in), reinterpret_cast<uint8_t*>(base + o_out));
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipMemcpyAsync(out, base + o_out, (size_t)n * 128, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  return ZKP_OK;
}
#endif  // ZKP_BUILD_TEST_HOOKS

#ifdef ZKP_BUILD_TEST_HOOKS
int zkp_debug_row_selftest(zkp_ctx* c, uint32_t n, const uint8_t* pairs, uint8_t* out) {
  if (!c || !pairs || !out) return fail(ZKP_ERR_ARG, "NULL pointer");
  if (n == 0) return ZKP_OK;
  HIP_TRY(hipSetDevice(c->device));
  carve cv;
  const size_t o_in = cv.take((size_t)n * 64);
  const size_t o_out = cv.take((size_t)n * 96);
  const int rc = ensure_ws(c, cv.off);
  if (rc) return rc;
  char* base = static_cast<char*>(c->ws);
  HIP_TRY(hipMemcpyAsync(base + o_in, pairs, (size_t)n * 64, hipMemcpyHostToDevice, c->stream));
  hipLaunchKernelGGL(k_debug_row, dim3(n), dim3(64), 0, c->stream, n, reinterpret_cast<uint8_t*>(base + o_in), reinterpret_cast<uint8_t*>(base + o_out));
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipMemcpyAsync(out, base + o_out, (size_t)n * 96, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  return ZKP_OK;
}
#endif  // ZKP_BUILD_TEST_HOOKS

#ifdef ZKP_BUILD_TEST_HOOKS
int zkp_debug_wave_cycles(zkp_ctx* c, uint64_t* out, uint32_t cap) {
  if (!c || !out) return fail(ZKP_ERR_ARG, "NULL pointer");
  if (!c->wave_cycles) return fail(ZKP_ERR_ARG, "ZKP_TESTOPT_WAVE_CYCLES is off");
  HIP_TRY(hipSetDevice(c->device));
  HIP_TRY(hipStreamSynchronize(c->stream));
  const uint32_t n = std::min(cap, zkp_ctx::kWaveCyclesCap);
  HIP_TRY(hipMemcpy(out, c->wave_cycles, sizeof(uint64_t) * n, hipMemcpyDeviceToHost));
  HIP_TRY(hipMemset(c->wave_cycles, 0, sizeof(uint64_t) * zkp_ctx::kWaveCyclesCap));
  return (int)n;
}
#endif

int zkp_encode_many(zkp_ctx* c, uint64_t n, const uint8_t* xyzt, uint8_t* out) {
  if (!c) return fail(ZKP_ERR_ARG, "ctx is NULL");
  if (n == 0) return ZKP_OK;
  if (!xyzt || !out) return fail(ZKP_ERR_ARG, "NULL pointer");
  if (n > 0x7fffffffull) return fail(ZKP_ERR_ARG, "n too large");
  HIP_TRY(hipSetDevice(c->device));
  carve cv;
  const size_t o_in = cv.take((size_t)n * 128);
  const size_t o_out = cv.take((size_t)n * 32);
  const int rc = ensure_ws(c, cv.off);
  if (rc) return rc;
  char* base = static_cast<char*>(c->ws);
  HIP_TRY(hipMemcpyAsync(base + o_in, xyzt, (size_t)n * 128, hipMemcpyHostToDevice, c->stream));
  prof_begin(c);
  hipLaunchKernelGGL(k_encode_many, grid1(n, 256), dim3(256), 0, c->stream, (uint32_t)n, reinterpret_cast<uint8_t*>(base + o_in), reinterpret_cast<uint8_t*>(base + o_out));
  prof_mark(c, ZKP_K_REDUCE);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipMemcpyAsync(out, base + o_out, (size_t)n * 32, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  return ZKP_OK;
}

}  // extern "C"

#include "fused_flows.h"
#include "host_jobs.h"
