// Fixed-base window tables for points that very many terms share.
//
// In the reference's statements most right-hand-side terms multiply a point that is COMMON to the whole batch
// (define_proof!'s common variables, BatchVerifier's static points: CMZ'13 has X_1..X_10 and A in 20 of the 31
// prover terms, benches/zkp.rs:32-45).  For such a point P the table
//       T[w][k-1] = k * 16^w * P      w = 0..64, k = 1..8        (affine niels form, 112 B each, 58 KB per point)
// turns s*P into 65 mixed additions (7M each) and NO doublings, against 256 doublings + 128 additions on the generic
// path.  Tables live in HBM/L2 (64 slots = 3.7 MB), are built once per point and kept across calls.
// Signed radix-16 digits come from the same carry-free offset recoding as everywhere else: e = s + 0x88..8,
// digit_w = nibble_w(e) - 8 in [-8, 7]; a carry out of bit 255 (non-canonical scalars only) selects T[64][0].
// ZKP_CT: all 8 entries of the row are read and the wanted one is picked by masks (addresses do not depend on the
// scalar).  ZKP_VARTIME: the entry is loaded directly.
#pragma once
#include "dev_layout.h"

namespace zkp {

constexpr int HOT_WINDOWS = 65;
constexpr int HOT_ENTRIES = 8;
constexpr int HOT_SLOTS = 64;
constexpr int HOT_CLASSES = HOT_SLOTS + 2;                       // class 64 = "cold" terms through a comb table, 65 = cold terms on a ladder
constexpr int CLASS_COMB = HOT_SLOTS, CLASS_LADDER = HOT_SLOTS + 1;
constexpr size_t HOT_SLOT_NIELS = (size_t)HOT_WINDOWS * HOT_ENTRIES;

// ---- table construction ---------------------------------------------------------------------------------------------
// bases[h][w] = 16^w * P_h      (one lane per point: 256 sequential doublings, paid once per point)
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
    ge_double<false>(b, b);
    ge_double<false>(b, b);
    ge_double<false>(b, b);
    ge_double<true>(b, b);
    store_ext(bases + (size_t)h * HOT_WINDOWS + w, b);
  }
}

// one lane per (point, window): the 8 multiples of the window base, normalised to affine niels with ONE shared
// inversion (Montgomery's trick over the 8 Z coordinates)
__global__ void __launch_bounds__(64, 2)
k_hot_rows(uint32_t nh, const uint32_t* __restrict__ slots, const dev_ext* __restrict__ bases,
           dev_niels* __restrict__ tables) {
  const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= nh * HOT_WINDOWS) return;
  const uint32_t h = g / HOT_WINDOWS, w = g - h * HOT_WINDOWS;
  ge_p3 m[8];
  load_ext(m[0], bases + (size_t)h * HOT_WINDOWS + w);
  ge_double<true>(m[1], m[0]);          // 2
  ge_add_p3(m[2], m[1], m[0]);          // 3
  ge_double<true>(m[3], m[1]);          // 4
  ge_add_p3(m[4], m[3], m[0]);          // 5
  ge_double<true>(m[5], m[2]);          // 6
  ge_add_p3(m[6], m[5], m[0]);          // 7
  ge_double<true>(m[7], m[3]);          // 8
  fe pre[8], inv, t, d2;
  pre[0] = m[0].Z;
#pragma unroll
  for (int k = 1; k < 8; ++k) fe_mul(pre[k], pre[k - 1], m[k].Z);
  fe_invert(inv, pre[7]);
  fe_from_const(d2, FE_D2);
  dev_niels* row = tables + (size_t)slots[h] * HOT_SLOT_NIELS + (size_t)w * HOT_ENTRIES;
#pragma unroll
  for (int k = 7; k >= 0; --k) {
    fe zinv;
    if (k > 0) { fe_mul(zinv, inv, pre[k - 1]); fe_mul(inv, inv, m[k].Z); } else zinv = inv;
    fe x, y;
    fe_mul(x, m[k].X, zinv);
    fe_mul(y, m[k].Y, zinv);
    ge_niels q;
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
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n_terms) return;
  const uint32_t pi = pidx[t];
  if (pi < n_points && hotmap[pi] < 0) atomicAdd(&uses[pi], 1u);
}
__device__ __forceinline__ uint32_t term_class(uint32_t t, const uint32_t* pidx, uint32_t n_points, const int32_t* hotmap,
                                               const uint32_t* uses, uint32_t comb_min) {
  const uint32_t pi = pidx[t];
  if (pi >= n_points) return (uint32_t)CLASS_COMB;                  // out of range: flagged by k_reduce_encode
  const int32_t s = hotmap[pi];
  if (s >= 0) return (uint32_t)s;
  return uses[pi] >= comb_min ? (uint32_t)CLASS_COMB : (uint32_t)CLASS_LADDER;
}
__global__ void __launch_bounds__(256)
k_class_count(uint32_t n_terms, const uint32_t* __restrict__ pidx, uint32_t n_points, const int32_t* __restrict__ hotmap,
              const uint32_t* __restrict__ uses, uint32_t comb_min, uint32_t* __restrict__ class_cnt) {
  __shared__ uint32_t h[HOT_CLASSES];
  if (threadIdx.x < HOT_CLASSES) h[threadIdx.x] = 0;
  __syncthreads();
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t < n_terms) atomicAdd(&h[term_class(t, pidx, n_points, hotmap, uses, comb_min)], 1u);
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
__global__ void __launch_bounds__(256)
k_class_scatter(uint32_t n_terms, const uint32_t* __restrict__ pidx, uint32_t n_points, const int32_t* __restrict__ hotmap,
                const uint32_t* __restrict__ uses, uint32_t comb_min, uint32_t* __restrict__ cursor, uint32_t* __restrict__ list) {
  __shared__ uint32_t h[HOT_CLASSES];
  __shared__ uint32_t base[HOT_CLASSES];
  if (threadIdx.x < HOT_CLASSES) h[threadIdx.x] = 0;
  __syncthreads();
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  uint32_t c = 0, rank = 0;
  if (t < n_terms) { c = term_class(t, pidx, n_points, hotmap, uses, comb_min); rank = atomicAdd(&h[c], 1u); }
  __syncthreads();
  if (threadIdx.x < HOT_CLASSES && h[threadIdx.x]) base[threadIdx.x] = atomicAdd(&cursor[threadIdx.x], h[threadIdx.x]);
  __syncthreads();
  if (t < n_terms) list[base[c] + rank] = t;
}

// ---- the fixed-base term kernel -----------------------------------------------------------------------------------------
// acc += sum over the windows 8 j0 .. 8 j1 - 1 of digit_w * 16^w * P;  rows = the table row of window 8 j0 (HBM or LDS)
template <bool CT>
__device__ __forceinline__ void fixed_base_windows(ge_p3& acc, const uint32_t e[8], const dev_niels* __restrict__ rows, int j0, int j1) {
#pragma unroll 1
  for (int j = j0; j < j1; ++j) {
    uint32_t cur = sel8(e, j);
#pragma unroll 1
    for (int k = 0; k < 8; ++k) {
      const uint32_t nib = cur & 15u;
      cur >>= 4;
      const uint32_t neg = (uint32_t)(nib < 8u);
      const uint32_t mag = neg ? 8u - nib : nib - 8u;             // 0..8
      const dev_niels* row = rows + (size_t)(8 * (j - j0) + k) * HOT_ENTRIES;
      ge_niels q;
      if (CT) {
        ge_niels_identity(q);
        // masked scan, four entries (28 independent 16-byte loads) in flight at a time
#pragma unroll 1
        for (uint32_t h = 0; h < 2; ++h) {
          ge_niels c0, c1, c2, c3;
          load_niels(c0, row + 4 * h + 0);
          load_niels(c1, row + 4 * h + 1);
          load_niels(c2, row + 4 * h + 2);
          load_niels(c3, row + 4 * h + 3);
          const ge_niels* cs[4] = {&c0, &c1, &c2, &c3};
#pragma unroll
          for (uint32_t m = 0; m < 4; ++m) {
            const uint32_t hit = (uint32_t)(mag == 4 * h + m + 1);
            fe_cmov(q.ypx, cs[m]->ypx, hit);
            fe_cmov(q.ymx, cs[m]->ymx, hit);
            fe_cmov(q.xy2d, cs[m]->xy2d, hit);
          }
        }
      } else {
        ge_niels_identity(q);
        if (mag) load_niels(q, row + (mag - 1));
      }
      ge_niels_cneg(q, neg);
      ge_madd(acc, acc, q);
    }
  }
}
// the carry window (a scalar >= 2^256 - 0x88..8): digit in {0, 1};  row64 = the table row of window 64
__device__ __forceinline__ void fixed_base_carry(ge_p3& acc, uint32_t top, const dev_niels* __restrict__ row64) {
  ge_niels q, c;
  ge_niels_identity(q);
  load_niels(c, row64);
  fe_cmov(q.ypx, c.ypx, top);
  fe_cmov(q.ymx, c.ymx, top);
  fe_cmov(q.xy2d, c.xy2d, top);
  ge_madd(acc, acc, q);
}

template <bool CT>
__device__ __forceinline__ void term_fixed_base(uint32_t t, const uint8_t* __restrict__ scalars, const dev_niels* __restrict__ T,
                                                dev_ext* __restrict__ partial) {
  uint32_t s[8], e[8], top;
  load_vec<2>(s, scalars + 32 * (size_t)t);
  sc_add_pattern(e, top, s, 0x88888888u);                          // digits nibble - 8 in [-8, 7]
  ge_p3 acc;
  ge_identity(acc);
  fixed_base_windows<CT>(acc, e, T, 0, 8);
  fixed_base_carry(acc, top, T + (size_t)64 * HOT_ENTRIES);
  store_ext(partial + t, acc);
}

}  // namespace zkp
