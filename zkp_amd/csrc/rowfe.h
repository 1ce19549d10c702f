// One limb per lane: point arithmetic for the one chain of the library that nothing can run beside -- the Horner tail
// sum_w 2^(C w) T_w of a Pippenger run (k_pip_combine: 253 doublings and W1 - 1 additions, each waiting for the one before).
//
// quad.h spreads a point over 4 lanes (a coordinate each): the 4 multiplications of a level run side by side, but every lane still
// walks all 81 + 17 products of its multiplication and 9 instructions for every limb-wise addition, selection and DPP move -- a
// lone wavefront issues one instruction per ~4.5 cycles whatever its 60 idle lanes do, so a quad doubling is ~470 instructions =
// 1.0 us.  Here a wavefront holds ONE point as 4 DPP rows of 16 lanes, row r = coordinate r (X, Y, Z, T), lane k < 9 of a row =
// limb k (fe25519.h's 9 x 29-bit limbs), lanes 9 .. 15 of a row = 0:
//   * limb-wise work (add, sub, select, carry) is ONE instruction per field operation;
//   * a multiplication is column k in lane k: 17 v_mad_u64_u32 on operands moved inside the row by DPP -- row_shr:j brings
//     a[k - j] (zero fill below lane 0), row_shl:(9 - j) brings a[k - j + 9] (the zero lanes 9 .. 15 and the row's end blank what
//     does not wrap), row_newbcast:j brings b[j] to the whole row -- full-rate VALU moves, no LDS: 9 + 16 moves, 17 + 2 mads and a
//     carry pass of ~20 instructions, ~65 in all against ~140 per lane in fe_mul;
//   * only movement BETWEEN coordinates leaves the row: ds_bpermute_b32, three stages per doubling.
// A doubling is ~160 instructions + 3 LDS-crossbar latencies instead of ~470 instructions.
//
// tools/model/rowfe_model.py is this file instruction for instruction over Python integers: values against big-integer
// arithmetic and every intermediate against its register width at the top of the limb classes used here (tests/test_rowfe_model.py);
// on the GPU: zkp_debug_row_selftest (test-hook build) against the oracle, and every MSM of the suite ends in this chain.
//
// Limb classes are fe25519.h's: row_mul / row_carry return "tight"; row_mul takes a x b with max(a) * max(b) * 9 + 2^46 < 2^64.
#pragma once
#include "quad.h"

namespace zkp {

struct rowctx {
  uint32_t k, r;          // limb index inside the row (9 .. 15: idle lanes), row = coordinate
  uint32_t mask;          // 2^29 - 1, limb 8: 2^23 - 1, idle lanes: 0
  uint32_t sh;            // 29, limb 8: 23
  uint32_t bias2p, bias4p;
  uint32_t fx;            // lane 0: 1216 (2^261 mod p), lane 1: 19, else 0 -- what row_shl:7 of the second-order carries is worth
  uint32_t gy;            // lane 0: 19, else 0                             -- what row_shl:8 of limb 8's carry is worth
  uint32_t live;          // k < 9
  uint32_t self4;         // byte address of this lane for ds_bpermute_b32
};

__device__ __forceinline__ void row_init(rowctx& c) {
  const uint32_t lane = threadIdx.x & 63u;
  c.k = lane & 15u;
  c.r = lane >> 4;
  c.live = c.k < 9u ? 1u : 0u;
  c.mask = c.k < 8u ? FE_M29 : (c.k == 8u ? FE_M23 : 0u);
  c.sh = c.k == 8u ? 23u : 29u;
  c.bias2p = c.k == 0u ? 0x3fffffdau : (c.k < 8u ? 0x3ffffffeu : (c.k == 8u ? 0x00fffffeu : 0u));
  c.bias4p = 2u * c.bias2p;
  c.fx = c.k == 0u ? 1216u : (c.k == 1u ? 19u : 0u);
  c.gy = c.k == 0u ? 19u : 0u;
  c.self4 = lane << 2;
}

template <int CTRL>
__device__ __forceinline__ uint32_t row_dpp(uint32_t x) {
  return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, CTRL, 0xf, 0xf, true);
}
#define ZKP_ROW_SHR(n) (0x110 + (n))
#define ZKP_ROW_SHL(n) (0x100 + (n))
#define ZKP_ROW_BCAST(n) (0x150 + (n))

// the lane of the same limb in row MAP[r]; MAP packs four 2-bit row numbers, row 0's lowest
template <int M0, int M1, int M2, int M3>
__device__ __forceinline__ uint32_t row_pull(const rowctx& c, uint32_t x) {
  constexpr uint32_t MAP = (uint32_t)(M0 | (M1 << 2) | (M2 << 4) | (M3 << 6));
  const uint32_t src = (((MAP >> (2u * c.r)) & 3u) << 6) | (c.k << 2);
  return (uint32_t)__builtin_amdgcn_ds_bpermute((int)src, (int)x);
}

// 64-bit columns -> tight limbs: two parallel carry passes.  Pass 1 splits a column (weight 2^(29 k)) into l (29 bits; limb 8: 23), m (the
// next 29) and h (the rest, < 2^12): m belongs to limb k + 1, h to limb k + 2; limb 8's m is worth 19 at limb 0 and its h 19 at limb 1,
// limb 7's h is worth 2^261 = 1216 at limb 0.  Pass 2 moves what that left above each limb (a few bits) one lane up, limb 8's to limb 0 x 19.
__device__ __forceinline__ uint32_t row_reduce(const rowctx& c, uint64_t col) {
  const uint32_t l = (uint32_t)col & c.mask;
  const uint64_t t = col >> c.sh;
  const uint32_t m = (uint32_t)t & FE_M29;
  const uint32_t h = (uint32_t)(t >> 29);
  const uint32_t m1 = row_dpp<ZKP_ROW_SHR(1)>(m), h2 = row_dpp<ZKP_ROW_SHR(2)>(h);
  const uint32_t X = row_dpp<ZKP_ROW_SHL(7)>(h), Y = row_dpp<ZKP_ROW_SHL(8)>(m);
  uint64_t n = (uint64_t)(l + m1 + h2);
  n += (uint64_t)c.fx * X;
  n += (uint64_t)c.gy * Y;
  const uint32_t cy = (uint32_t)(n >> c.sh);
  const uint32_t r = (uint32_t)n & c.mask;
  const uint32_t c1 = row_dpp<ZKP_ROW_SHR(1)>(cy), c8 = row_dpp<ZKP_ROW_SHL(8)>(cy);
  const uint32_t o = r + c1 + __umul24(c.gy, c8);
  return c.live ? o : 0u;
}

// weak reduction of 32-bit limbs (fe_carry): one pass
__device__ __forceinline__ uint32_t row_carry(const rowctx& c, uint32_t v) {
  const uint32_t cy = v >> c.sh, r = v & c.mask;
  const uint32_t c1 = row_dpp<ZKP_ROW_SHR(1)>(cy), c8 = row_dpp<ZKP_ROW_SHL(8)>(cy);
  const uint32_t o = r + c1 + __umul24(c.gy, c8);
  return c.live ? o : 0u;
}

// a: zero in the idle lanes (every value this file produces is).  b: only its lanes 0 .. 8 are read.
#define ZKP_ROW_MUL_STEP(j)                                                             \
  {                                                                                     \
    const uint32_t bj = row_dpp<ZKP_ROW_BCAST(j)>(b);                                   \
    lo += (uint64_t)row_dpp<ZKP_ROW_SHR(j)>(a) * bj;                                    \
    hi += (uint64_t)row_dpp<ZKP_ROW_SHL(9 - (j))>(a) * bj;                              \
  }
__device__ __forceinline__ uint32_t row_mul(const rowctx& c, uint32_t a, uint32_t b) {
  uint64_t lo = (uint64_t)a * row_dpp<ZKP_ROW_BCAST(0)>(b), hi = 0;
  ZKP_ROW_MUL_STEP(1) ZKP_ROW_MUL_STEP(2) ZKP_ROW_MUL_STEP(3) ZKP_ROW_MUL_STEP(4)
  ZKP_ROW_MUL_STEP(5) ZKP_ROW_MUL_STEP(6) ZKP_ROW_MUL_STEP(7) ZKP_ROW_MUL_STEP(8)
  // columns 9 .. 16 fold as in fe_mul, in 32-bit halves: 2^(29 (k + 9)) = 1216 * 2^(29 k), 2^(29 (k + 9) + 32) = 9728 * 2^(29 (k + 1))
  const uint32_t hi32 = row_dpp<ZKP_ROW_SHR(1)>((uint32_t)(hi >> 32));
  uint64_t col = lo + 1216ull * (uint32_t)hi;
  col += 9728ull * hi32;
  return row_reduce(c, col);
}
#undef ZKP_ROW_MUL_STEP

// p = 2 p   (rows X, Y, Z, T, tight)
__device__ __forceinline__ uint32_t row_double(const rowctx& c, uint32_t p) {
  const uint32_t x = row_pull<0, 0, 0, 0>(c, p), y = row_pull<1, 1, 1, 1>(c, p);
  const uint32_t t = c.r == 3u ? x + y : p;                         // row 3 squares X + Y instead of T
  const uint32_t s = row_mul(c, t, t);                              // XX, YY, ZZ, (X + Y)^2
  const uint32_t a = row_pull<0, 0, 0, 0>(c, s), b = row_pull<1, 1, 1, 1>(c, s);
  const uint32_t h = b + a;                                         // H = YY + XX
  const uint32_t g = b + (c.bias2p - a);                            // G = YY - XX
  const uint32_t e = s + (c.bias4p - h);                            // row 3: E = (X + Y)^2 - H
  const uint32_t f = (s + s) + (c.bias4p - g);                      // row 2: F = 2 ZZ - G
  const uint32_t v = row_carry(c, c.r == 3u ? e : f);
  const uint32_t ee = row_pull<3, 3, 3, 3>(c, v), ff = row_pull<2, 2, 2, 2>(c, v);
  const uint32_t m1 = c.r == 1u ? g : (c.r == 2u ? ff : ee);       // E, G, F, E
  const uint32_t m2 = c.r == 0u ? ff : (c.r == 2u ? g : h);         // F, H, G, H
  return row_mul(c, m1, m2);                                        // X3 = E F, Y3 = G H, Z3 = F G, T3 = E H
}

// p + q, q in cached form: rows Y2 - X2, Y2 + X2, 2 Z2, 2 d T2 (tight)
__device__ __forceinline__ uint32_t row_add_cached(const rowctx& c, uint32_t p, uint32_t q) {
  uint32_t o = row_pull<1, 0, 2, 3>(c, p);
  const uint32_t t = c.r == 0u ? o + (c.bias2p - p) : (c.r == 1u ? p + o : p);      // Y1 - X1, Y1 + X1, Z1, T1
  const uint32_t u = row_mul(c, t, q);                                               // A, B, D, C
  o = row_pull<1, 0, 3, 2>(c, u);
  const uint32_t w = (c.r & 1u) ? u + o : (c.r == 0u ? o + (c.bias2p - u) : u + (c.bias2p - o));
  const uint32_t v = row_carry(c, w);                                                // E = B - A, H = B + A, F = D - C, G = D + C
  const uint32_t m2 = row_pull<2, 3, 3, 1>(c, v);                                    // F, G, G, H
  const uint32_t e0 = row_pull<0, 0, 0, 0>(c, v);
  const uint32_t m1 = c.r == 3u ? e0 : v;                                            // E, H, F, E
  return row_mul(c, m1, m2);                                                         // X3 = E F, Y3 = H G, Z3 = F G, T3 = E H
}

}  // namespace zkp
