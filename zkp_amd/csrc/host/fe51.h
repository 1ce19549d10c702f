// GF(2^255-19) for the HOST backend: 5 unsigned limbs of 51 bits in 64-bit registers, products in unsigned __int128.
//
// fe25519.h is shaped for gfx950 (9 x 29-bit limbs in 32-bit VGPRs, v_mad_u64_u32 columns); compiled for an x86-64 core it needs 81 + 17
// 32 x 32 products per multiplication where 25 64 x 64 products do.  The host backend (host/host_backend.cpp: tiny calls and calls without
// a GPU, BASELINE configs[0]) therefore includes THIS header through fe25519.h (-DZKP_HOST_FE51): the same function names and the same
// laziness contract, so that ge25519.h -- the point formulas and the ristretto255 codec the kernels are compiled from -- is used unchanged.
// Only host/host_backend.cpp may define ZKP_HOST_FE51: two translation units of one library with different `zkp::fe` would break the ODR.
//
// The contract ge25519.h was written against (fe25519.h, "limb-bound vocabulary"), scaled from 29 to 51 bits:
//   tight : output of fe_mul / fe_sq / fe_carry / fe_fromwords          limbs < 2^51 + 2^14
//   sum   : tight + tight                                               < 2^52 + 2^15
//   diff  : tight + BIAS2P - tight (fe_sub: subtrahend limbs <= 2^52 - 38)   < 3 * 2^51
//   fe_sub4: subtrahend limbs <= 2^53 - 76 (sum / diff class)
//   fe_mul / fe_sq accept limbs < 2^55 (every operand fe25519.h's rule  max(a) max(b) 8 < 2^61  admits is < 2^54 here); additions of
//   anything the 32-bit version could hold (<= 8 x tight) stay below 2^55.
// Replaces curve25519-dalek's `FieldElement51` (backend::serial::u64::field) for this path; checked against the oracle and against the
// GPU route byte for byte (tests/test_host_backend.py).
#pragma once
#include <stdint.h>

#define FE_TRACK(stmt) do { } while (0)

namespace zkp {

typedef unsigned __int128 fe_u128;
constexpr uint64_t FE_M51 = (1ull << 51) - 1;

struct fe { uint64_t v[5]; };

ZKP_HD void fe_0(fe& r) { for (int i = 0; i < 5; ++i) r.v[i] = 0; }
ZKP_HD void fe_1(fe& r) { fe_0(r); r.v[0] = 1; }
ZKP_HD void fe_copy(fe& r, const fe& a) { r = a; }

// 2p and 4p with every limb >= the limbs they are subtracted from
ZKP_HD uint64_t fe_bias2p(int i) { return i == 0 ? 0xfffffffffffdaull : 0xffffffffffffeull; }       // 2^52 - 38, 2^52 - 2
ZKP_HD uint64_t fe_bias4p(int i) { return i == 0 ? 0x1fffffffffffb4ull : 0x1ffffffffffffcull; }     // 2^53 - 76, 2^53 - 4

ZKP_HD void fe_add(fe& r, const fe& a, const fe& b) { for (int i = 0; i < 5; ++i) r.v[i] = a.v[i] + b.v[i]; }
ZKP_HD void fe_sub(fe& r, const fe& a, const fe& b) { for (int i = 0; i < 5; ++i) r.v[i] = a.v[i] + (fe_bias2p(i) - b.v[i]); }
ZKP_HD void fe_sub4(fe& r, const fe& a, const fe& b) { for (int i = 0; i < 5; ++i) r.v[i] = a.v[i] + (fe_bias4p(i) - b.v[i]); }
ZKP_HD void fe_neg(fe& r, const fe& a) { for (int i = 0; i < 5; ++i) r.v[i] = fe_bias2p(i) - a.v[i]; }

// weak reduction, all limbs in parallel: result tight
ZKP_HD void fe_carry(fe& r, const fe& a) {
  uint64_t c[5];
  for (int i = 0; i < 5; ++i) c[i] = a.v[i] >> 51;
  r.v[0] = (a.v[0] & FE_M51) + 19 * c[4];
  for (int i = 1; i < 5; ++i) r.v[i] = (a.v[i] & FE_M51) + c[i - 1];
}

ZKP_HD void fe_cmov(fe& r, const fe& b, uint32_t flag) {
  const uint64_t m = 0ull - (uint64_t)(flag & 1u);
  for (int i = 0; i < 5; ++i) r.v[i] ^= m & (r.v[i] ^ b.v[i]);
}
ZKP_HD void fe_cswap(fe& a, fe& b, uint32_t flag) {
  const uint64_t m = 0ull - (uint64_t)(flag & 1u);
  for (int i = 0; i < 5; ++i) { const uint64_t t = m & (a.v[i] ^ b.v[i]); a.v[i] ^= t; b.v[i] ^= t; }
}

// five 128-bit columns -> tight limbs.  c[4] holds no wrapped product, so its carry times 19 is added in 128 bits.
ZKP_HD void fe_reduce_columns(fe& r, fe_u128 c[5]) {
  c[1] += c[0] >> 51;
  c[2] += c[1] >> 51;
  c[3] += c[2] >> 51;
  c[4] += c[3] >> 51;
  const fe_u128 t0 = (fe_u128)((uint64_t)c[0] & FE_M51) + (c[4] >> 51) * 19;
  r.v[0] = (uint64_t)t0 & FE_M51;
  r.v[1] = ((uint64_t)c[1] & FE_M51) + (uint64_t)(t0 >> 51);
  r.v[2] = (uint64_t)c[2] & FE_M51;
  r.v[3] = (uint64_t)c[3] & FE_M51;
  r.v[4] = (uint64_t)c[4] & FE_M51;
}

ZKP_HD void fe_mul(fe& r, const fe& a, const fe& b) {
  const uint64_t a0 = a.v[0], a1 = a.v[1], a2 = a.v[2], a3 = a.v[3], a4 = a.v[4];
  const uint64_t b0 = b.v[0], b1 = b.v[1], b2 = b.v[2], b3 = b.v[3], b4 = b.v[4];
  const uint64_t b1_19 = 19 * b1, b2_19 = 19 * b2, b3_19 = 19 * b3, b4_19 = 19 * b4;        // < 2^55 * 19 < 2^60
  fe_u128 c[5];
  c[0] = (fe_u128)a0 * b0 + (fe_u128)a4 * b1_19 + (fe_u128)a3 * b2_19 + (fe_u128)a2 * b3_19 + (fe_u128)a1 * b4_19;
  c[1] = (fe_u128)a1 * b0 + (fe_u128)a0 * b1 + (fe_u128)a4 * b2_19 + (fe_u128)a3 * b3_19 + (fe_u128)a2 * b4_19;
  c[2] = (fe_u128)a2 * b0 + (fe_u128)a1 * b1 + (fe_u128)a0 * b2 + (fe_u128)a4 * b3_19 + (fe_u128)a3 * b4_19;
  c[3] = (fe_u128)a3 * b0 + (fe_u128)a2 * b1 + (fe_u128)a1 * b2 + (fe_u128)a0 * b3 + (fe_u128)a4 * b4_19;
  c[4] = (fe_u128)a4 * b0 + (fe_u128)a3 * b1 + (fe_u128)a2 * b2 + (fe_u128)a1 * b3 + (fe_u128)a0 * b4;
  fe_reduce_columns(r, c);
}

ZKP_HD void fe_sq(fe& r, const fe& a) {
  const uint64_t a0 = a.v[0], a1 = a.v[1], a2 = a.v[2], a3 = a.v[3], a4 = a.v[4];
  const uint64_t a3_19 = 19 * a3, a4_19 = 19 * a4;
  const uint64_t d0 = 2 * a0, d1 = 2 * a1, d2 = 2 * a2;
  fe_u128 c[5];
  c[0] = (fe_u128)a0 * a0 + (fe_u128)d1 * a4_19 + (fe_u128)d2 * a3_19;
  c[1] = (fe_u128)d0 * a1 + (fe_u128)d2 * a4_19 + (fe_u128)a3 * a3_19;
  c[2] = (fe_u128)d0 * a2 + (fe_u128)a1 * a1 + (fe_u128)(2 * a3) * a4_19;
  c[3] = (fe_u128)d0 * a3 + (fe_u128)d1 * a2 + (fe_u128)a4 * a4_19;
  c[4] = (fe_u128)d0 * a4 + (fe_u128)d1 * a3 + (fe_u128)a2 * a2;
  fe_reduce_columns(r, c);
}

ZKP_HD void fe_sqn(fe& r, const fe& a, int n) {
  r = a;
  for (int i = 0; i < n; ++i) fe_sq(r, r);
}

// ---- bytes <-> limbs: w[0..7] = the 32 bytes as little-endian 32-bit words; bit 255 is IGNORED -----------------------------------
ZKP_HD void fe_fromwords(fe& r, const uint32_t w[8]) {
  uint64_t q[4];
  for (int i = 0; i < 4; ++i) q[i] = (uint64_t)w[2 * i] | (uint64_t)w[2 * i + 1] << 32;
  r.v[0] = q[0] & FE_M51;
  r.v[1] = (q[0] >> 51 | q[1] << 13) & FE_M51;
  r.v[2] = (q[1] >> 38 | q[2] << 26) & FE_M51;
  r.v[3] = (q[2] >> 25 | q[3] << 39) & FE_M51;
  r.v[4] = (q[3] >> 12) & FE_M51;
}

// 1 iff the 256-bit little-endian integer in w is < p (so also bit 255 clear)
ZKP_HD uint32_t fe_words_canonical(const uint32_t w[8]) {
  const uint32_t all_ones = w[1] & w[2] & w[3] & w[4] & w[5] & w[6];
  const uint32_t ge_p = (uint32_t)(w[7] == 0x7fffffffu) & (uint32_t)(all_ones == 0xffffffffu) & (uint32_t)(w[0] >= 0xffffffedu);
  return (uint32_t)((w[7] >> 31) == 0) & (ge_p ^ 1u);
}

// canonical little-endian words of a (any limbs < 2^63)
ZKP_HD void fe_towords(uint32_t w[8], const fe& a) {
  uint64_t t[5];
  for (int i = 0; i < 5; ++i) t[i] = a.v[i];
  for (int pass = 0; pass < 2; ++pass) {                     // two sequential weak passes: value < 2^255 + 19 * 2^12, then < 2^255 + 19
    for (int i = 0; i < 4; ++i) { t[i + 1] += t[i] >> 51; t[i] &= FE_M51; }
    const uint64_t c = t[4] >> 51;
    t[4] &= FE_M51;
    t[0] += 19 * c;
  }
  // V < 2p (t[0] may hold a pending carry of a few units).  V >= p  <=>  u = V + 19 >= 2^255, and then V - p = u - 2^255.
  uint64_t u[5], s[5];
  u[0] = t[0] + 19;
  s[0] = t[0];
  for (int i = 0; i < 4; ++i) {
    u[i + 1] = t[i + 1] + (u[i] >> 51); u[i] &= FE_M51;
    s[i + 1] = t[i + 1] + (s[i] >> 51); s[i] &= FE_M51;
  }
  const uint64_t m = 0ull - (u[4] >> 51);                     // all ones iff V >= p
  u[4] &= FE_M51;
  for (int i = 0; i < 5; ++i) t[i] = (u[i] & m) | (s[i] & ~m);
  const uint64_t q0 = t[0] | t[1] << 51, q1 = t[1] >> 13 | t[2] << 38, q2 = t[2] >> 26 | t[3] << 25, q3 = t[3] >> 39 | t[4] << 12;
  w[0] = (uint32_t)q0; w[1] = (uint32_t)(q0 >> 32); w[2] = (uint32_t)q1; w[3] = (uint32_t)(q1 >> 32);
  w[4] = (uint32_t)q2; w[5] = (uint32_t)(q2 >> 32); w[6] = (uint32_t)q3; w[7] = (uint32_t)(q3 >> 32);
}

ZKP_HD uint32_t fe_isnegative(const fe& a) {
  uint32_t w[8];
  fe_towords(w, a);
  return w[0] & 1u;
}
ZKP_HD uint32_t fe_iszero(const fe& a) {
  uint32_t w[8];
  fe_towords(w, a);
  uint32_t x = 0;
  for (int i = 0; i < 8; ++i) x |= w[i];
  return (uint32_t)(x == 0);
}
ZKP_HD uint32_t fe_equal(const fe& a, const fe& b) {
  uint32_t wa[8], wb[8];
  fe_towords(wa, a);
  fe_towords(wb, b);
  uint32_t x = 0;
  for (int i = 0; i < 8; ++i) x |= wa[i] ^ wb[i];
  return (uint32_t)(x == 0);
}
// r = |a| (RFC 9496 CT_ABS): negate when the canonical value is odd.  a: limbs <= bias2p.
ZKP_HD void fe_abs(fe& r, const fe& a) {
  fe n;
  fe_neg(n, a);
  const uint32_t neg = fe_isnegative(a);
  r = a;
  fe_cmov(r, n, neg);
}

// z^(2^252 - 3) = z^((p-5)/8): 251 squarings + 11 multiplications (the chain of fe25519.h)
ZKP_HD void fe_pow22523(fe& out, const fe& z) {
  fe t0, t1, t2;
  fe_sq(t0, z);                 // 2
  fe_sqn(t1, t0, 2);            // 8
  fe_mul(t1, z, t1);            // 9
  fe_mul(t0, t0, t1);           // 11
  fe_sq(t0, t0);                // 22
  fe_mul(t0, t1, t0);           // 31 = 2^5 - 1
  fe_sqn(t1, t0, 5);
  fe_mul(t0, t1, t0);           // 2^10 - 1
  fe_sqn(t1, t0, 10);
  fe_mul(t1, t1, t0);           // 2^20 - 1
  fe_sqn(t2, t1, 20);
  fe_mul(t1, t2, t1);           // 2^40 - 1
  fe_sqn(t1, t1, 10);
  fe_mul(t0, t1, t0);           // 2^50 - 1
  fe_sqn(t1, t0, 50);
  fe_mul(t1, t1, t0);           // 2^100 - 1
  fe_sqn(t2, t1, 100);
  fe_mul(t1, t2, t1);           // 2^200 - 1
  fe_sqn(t1, t1, 50);
  fe_mul(t0, t1, t0);           // 2^250 - 1
  fe_sqn(t0, t0, 2);            // 2^252 - 4
  fe_mul(out, t0, z);           // 2^252 - 3
}

// r = 1/z = z^(p-2) = (z^(2^252-3))^8 * z^3   (0 -> 0)
ZKP_HD void fe_invert(fe& r, const fe& z) {
  fe t, z3;
  fe_pow22523(t, z);
  fe_sqn(t, t, 3);
  fe_sq(z3, z);
  fe_mul(z3, z3, z);
  fe_mul(r, t, z3);
}

// field constants are generated as 9 x 29-bit limbs (fe_constants.h, tools/gen_constants.py): repack
struct fe_const { uint32_t v[9]; };
ZKP_HD void fe_from_const(fe& r, const fe_const& c) {
  uint64_t q[5] = {0, 0, 0, 0, 0};
  for (int i = 0; i < 9; ++i) {
    const int bit = 29 * i, wi = bit >> 6, sh = bit & 63;
    q[wi] |= (uint64_t)c.v[i] << sh;
    if (sh + 29 > 64) q[wi + 1] |= (uint64_t)c.v[i] >> (64 - sh);
  }
  r.v[0] = q[0] & FE_M51;
  r.v[1] = (q[0] >> 51 | q[1] << 13) & FE_M51;
  r.v[2] = (q[1] >> 38 | q[2] << 26) & FE_M51;
  r.v[3] = (q[2] >> 25 | q[3] << 39) & FE_M51;
  r.v[4] = (q[3] >> 12) & FE_M51;
}

}  // namespace zkp
