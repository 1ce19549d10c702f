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
//   * movement BETWEEN coordinates: v_permlane32_swap / v_permlane16_swap (gfx950), three instructions per stage, three stages per doubling.
// A doubling is ~190 instructions instead of ~470.
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
}

template <int CTRL>
__device__ __forceinline__ uint32_t row_dpp(uint32_t x) {
  return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, CTRL, 0xf, 0xf, true);
}
#define ZKP_ROW_SHR(n) (0x110 + (n))
#define ZKP_ROW_SHL(n) (0x100 + (n))
#define ZKP_ROW_BCAST(n) (0x150 + (n))

// Movement BETWEEN rows.  gfx950 has two full-rate VALU swaps for it: v_permlane32_swap_b32 (lanes 32 .. 63 of its first operand <-> lanes 0 .. 31 of its
// second) and v_permlane16_swap_b32 (the odd 16-lane rows of the first <-> the even rows of the second).  Fed the same value twice they leave that value's
// lower / upper half in both halves of the two results, then row 0 / row 1 (2 / 3) in every row: three instructions put each of a point's four coordinates
// in front of every row, where ds_bpermute_b32 pays an LDS-crossbar round trip (~110 cycles of a chain that has nothing else to issue) per stage.
// -DZKP_AB_ROW_BPERMUTE keeps the crossbar (the A/B of profiles/r05_ab_experiments.txt block i).
struct rowpair { uint32_t a, b; };
struct rowquad { uint32_t r0, r1, r2, r3; };
#ifndef ZKP_AB_ROW_BPERMUTE
__device__ __forceinline__ rowpair row_halves(uint32_t x) {          // a: rows (0, 1, 0, 1) of x; b: rows (2, 3, 2, 3)
  const auto h = __builtin_amdgcn_permlane32_swap(x, x, false, false);
  return {h[0], h[1]};
}
__device__ __forceinline__ rowpair row_split16(uint32_t y) {         // y = rows (p, q, p, q)  ->  a: p everywhere; b: q everywhere
  const auto h = __builtin_amdgcn_permlane16_swap(y, y, false, false);
  return {h[0], h[1]};
}
__device__ __forceinline__ rowpair row_bcast01(const rowctx&, uint32_t x) { return row_split16(row_halves(x).a); }
__device__ __forceinline__ rowpair row_bcast23(const rowctx&, uint32_t x) { return row_split16(row_halves(x).b); }
__device__ __forceinline__ rowquad row_bcast_all(const rowctx&, uint32_t x) {
  const rowpair h = row_halves(x);
  const rowpair lo = row_split16(h.a), hi = row_split16(h.b);
  return {lo.a, lo.b, hi.a, hi.b};
}
#else
__device__ __forceinline__ uint32_t row_from(const rowctx& c, uint32_t x, uint32_t row) {
  return (uint32_t)__builtin_amdgcn_ds_bpermute((int)((row << 6) | (c.k << 2)), (int)x);
}
__device__ __forceinline__ rowpair row_bcast01(const rowctx& c, uint32_t x) { return {row_from(c, x, 0), row_from(c, x, 1)}; }
__device__ __forceinline__ rowpair row_bcast23(const rowctx& c, uint32_t x) { return {row_from(c, x, 2), row_from(c, x, 3)}; }
__device__ __forceinline__ rowquad row_bcast_all(const rowctx& c, uint32_t x) { return {row_from(c, x, 0), row_from(c, x, 1), row_from(c, x, 2), row_from(c, x, 3)}; }
#endif

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
  const rowpair xy = row_bcast01(c, p);
  const uint32_t t = c.r == 3u ? xy.a + xy.b : p;                   // row 3 squares X + Y instead of T
  const uint32_t s = row_mul(c, t, t);                              // XX, YY, ZZ, (X + Y)^2
  const rowpair ab = row_bcast01(c, s);
  const uint32_t h = ab.b + ab.a;                                   // H = YY + XX
  const uint32_t g = ab.b + (c.bias2p - ab.a);                      // G = YY - XX
  const uint32_t e = s + (c.bias4p - h);                            // row 3: E = (X + Y)^2 - H
  const uint32_t f = (s + s) + (c.bias4p - g);                      // row 2: F = 2 ZZ - G
  const uint32_t v = row_carry(c, c.r == 3u ? e : f);
  const rowpair fe_ = row_bcast23(c, v);                            // a: F, b: E
  const uint32_t m1 = c.r == 1u ? g : (c.r == 2u ? fe_.a : fe_.b);  // E, G, F, E
  const uint32_t m2 = c.r == 0u ? fe_.a : (c.r == 2u ? g : h);      // F, H, G, H
  return row_mul(c, m1, m2);                                        // X3 = E F, Y3 = G H, Z3 = F G, T3 = E H
}

// p + q, q in cached form: rows Y2 - X2, Y2 + X2, 2 Z2, 2 d T2 (tight)
__device__ __forceinline__ uint32_t row_add_cached(const rowctx& c, uint32_t p, uint32_t q) {
  const rowpair xy = row_bcast01(c, p);
  const uint32_t t = c.r == 0u ? xy.b + (c.bias2p - p) : (c.r == 1u ? p + xy.a : p);   // Y1 - X1, Y1 + X1, Z1, T1
  const uint32_t u = row_mul(c, t, q);                                                // A, B, D, C
  const rowquad uu = row_bcast_all(c, u);
  const uint32_t o = c.r == 0u ? uu.r1 : (c.r == 1u ? uu.r0 : (c.r == 2u ? uu.r3 : uu.r2));   // the other of (A, B), of (D, C)
  const uint32_t w = (c.r & 1u) ? u + o : (c.r == 0u ? o + (c.bias2p - u) : u + (c.bias2p - o));
  const uint32_t v = row_carry(c, w);                                                // E = B - A, H = B + A, F = D - C, G = D + C
  const rowquad vv = row_bcast_all(c, v);
  const uint32_t m2 = c.r == 0u ? vv.r2 : (c.r == 3u ? vv.r1 : vv.r3);               // F, G, G, H
  const uint32_t m1 = c.r == 3u ? vv.r0 : v;                                         // E, H, F, E
  return row_mul(c, m1, m2);                                                         // X3 = E F, Y3 = H G, Z3 = F G, T3 = E H
}

// a^(2^n)
__device__ __forceinline__ uint32_t row_sqn(const rowctx& c, uint32_t a, int n) {
#pragma unroll 1
  for (int i = 0; i < n; ++i) a = row_mul(c, a, a);
  return a;
}

// 1 / z = z^(p - 2) in every row (0 -> 0): fe_invert's addition chain (254 squarings, 12 multiplications), k_encode_invert 94 -> 54 us -- the one inversion
// behind a block of batched encodings, which every output of a prove call waits for
__device__ __forceinline__ uint32_t row_invert(const rowctx& c, uint32_t z) {
  uint32_t t0 = row_mul(c, z, z);                  // 2
  uint32_t t1 = row_sqn(c, t0, 2);                 // 8
  t1 = row_mul(c, z, t1);                          // 9
  t0 = row_mul(c, t0, t1);                         // 11
  const uint32_t z11 = t0;
  t0 = row_mul(c, t0, t0);                         // 22
  t0 = row_mul(c, t1, t0);                         // 2^5 - 1
  t1 = row_sqn(c, t0, 5);
  t0 = row_mul(c, t1, t0);                         // 2^10 - 1
  t1 = row_sqn(c, t0, 10);
  t1 = row_mul(c, t1, t0);                         // 2^20 - 1
  uint32_t t2 = row_sqn(c, t1, 20);
  t1 = row_mul(c, t2, t1);                         // 2^40 - 1
  t1 = row_sqn(c, t1, 10);
  t0 = row_mul(c, t1, t0);                         // 2^50 - 1
  t1 = row_sqn(c, t0, 50);
  t1 = row_mul(c, t1, t0);                         // 2^100 - 1
  t2 = row_sqn(c, t1, 100);
  t1 = row_mul(c, t2, t1);                         // 2^200 - 1
  t1 = row_sqn(c, t1, 50);
  t0 = row_mul(c, t1, t0);                         // 2^250 - 1
  t0 = row_sqn(c, t0, 5);                          // 2^255 - 32
  return row_mul(c, t0, z11);                      // 2^255 - 21 = p - 2
}

}  // namespace zkp
