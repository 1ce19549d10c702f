// Scalars modulo l = 2^252 + 27742317777372353535851937790883648493 on the GPU: eight 32-bit limbs, Montgomery
// multiplication (R = 2^256) built from the same full-rate v_mad_u64_u32 the field code uses.
// Needed by the batch-verification coefficient build (reference src/toolbox/batch_verifier.rs:173-206:
// `random_factor * minus_c[j]`, `random_factor * resp`, accumulation into the coefficient matrix).
// Host + device header, tested on the CPU against Python integers (tests/test_host_field.py).
#pragma once
#include <stdint.h>
#include "fe25519.h"   // ZKP_HD

namespace zkp {

struct sc { uint32_t v[8]; };     // little-endian limbs, always < l unless stated otherwise

ZKP_HD uint32_t sc_l(int i) {
  return i == 0 ? 0x5cf5d3edu : i == 1 ? 0x5812631au : i == 2 ? 0xa2f79cd6u : i == 3 ? 0x14def9deu : i == 4 ? 0x00000000u : i == 5 ? 0x00000000u : i == 6 ? 0x00000000u : 0x10000000u;
}
// R^2 mod l, R = 2^256 (generated: pow(2, 512, l))
ZKP_HD uint32_t sc_rr(int i) {
  return i == 0 ? 0x449c0f01u : i == 1 ? 0xa40611e3u : i == 2 ? 0x68859347u : i == 3 ? 0xd00e1ba7u : i == 4 ? 0x17f5be65u : i == 5 ? 0xceec73d2u : i == 6 ? 0x7c309a3du : 0x0399411bu;
}
// R mod l
ZKP_HD uint32_t sc_r1(int i) {
  return i == 0 ? 0x8d98951du : i == 1 ? 0xd6ec3174u : i == 2 ? 0x737dcf70u : i == 3 ? 0xc6ef5bf4u : i == 4 ? 0xfffffffeu : i == 5 ? 0xffffffffu : i == 6 ? 0xffffffffu : 0x0fffffffu;
}
// (l - 1) / 2
ZKP_HD uint32_t sc_half(int i) {
  return i == 0 ? 0x2e7ae9f6u : i == 1 ? 0x2c09318du : i == 2 ? 0x517bce6bu : i == 3 ? 0x0a6f7cefu : i == 7 ? 0x08000000u : 0u;
}
constexpr uint32_t SC_N0INV = 0x12547e1bu;      // -l^-1 mod 2^32

ZKP_HD void sc_zero(sc& r) {
#pragma unroll
  for (int i = 0; i < 8; ++i) r.v[i] = 0;
}
// a -= l if a >= l   (a < 2 l)
ZKP_HD void sc_cond_sub_l(sc& a) {
  uint32_t d[8];
  uint64_t br = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const uint64_t t = (uint64_t)a.v[i] - sc_l(i) - br;
    d[i] = (uint32_t)t;
    br = (t >> 63) & 1u;
  }
  const bool ge = br == 0;                    // no borrow: a >= l
#pragma unroll
  for (int i = 0; i < 8; ++i) a.v[i] = ge ? d[i] : a.v[i];
}
// r = a + b mod l   (a, b < l)
ZKP_HD void sc_add(sc& r, const sc& a, const sc& b) {
  uint64_t c = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    c += (uint64_t)a.v[i] + b.v[i];
    r.v[i] = (uint32_t)c;
    c >>= 32;
  }
  sc_cond_sub_l(r);                           // a + b < 2 l < 2^254: no carry out
}
// r = -a mod l   (a < l)
ZKP_HD void sc_neg(sc& r, const sc& a) {
  uint32_t nz = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) nz |= a.v[i];
  const uint32_t m = 0u - (uint32_t)(nz != 0);
  uint64_t br = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const uint64_t t = (uint64_t)sc_l(i) - a.v[i] - br;
    r.v[i] = (uint32_t)t & m;
    br = (t >> 63) & 1u;
  }
}
// Montgomery product a * b * 2^-256 mod l.  b < l; a any 256-bit value.  Result < l.
ZKP_HD void sc_mont(sc& r, const sc& a, const sc& b) {
  uint32_t t[10];
#pragma unroll
  for (int i = 0; i < 10; ++i) t[i] = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    uint64_t c = 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      c += (uint64_t)a.v[i] * b.v[j] + t[j];
      t[j] = (uint32_t)c;
      c >>= 32;
    }
    c += t[8];
    t[8] = (uint32_t)c;
    t[9] = (uint32_t)(c >> 32);
    const uint32_t m = t[0] * SC_N0INV;
    c = (uint64_t)m * sc_l(0) + t[0];
    c >>= 32;
#pragma unroll
    for (int j = 1; j < 8; ++j) {
      c += (uint64_t)m * sc_l(j) + t[j];
      t[j - 1] = (uint32_t)c;
      c >>= 32;
    }
    c += t[8];
    t[7] = (uint32_t)c;
    t[8] = t[9] + (uint32_t)(c >> 32);
  }
#pragma unroll
  for (int i = 0; i < 8; ++i) r.v[i] = t[i];
  sc_cond_sub_l(r);                           // < 2 l before
}
// r = a * b mod l   (two Montgomery products)
ZKP_HD void sc_mul(sc& r, const sc& a, const sc& b) {
  sc t, rr;
#pragma unroll
  for (int i = 0; i < 8; ++i) rr.v[i] = sc_rr(i);
  sc_mont(t, a, b);
  sc_mont(r, t, rr);
}
// to Montgomery form: a * R mod l  (so that sc_mont(to_mont(a), b) = a * b)
ZKP_HD void sc_to_mont(sc& r, const sc& a) {
  sc rr;
#pragma unroll
  for (int i = 0; i < 8; ++i) rr.v[i] = sc_rr(i);
  sc_mont(r, a, rr);
}
// any 256-bit value -> canonical representative
ZKP_HD void sc_reduce(sc& r, const sc& a) {
  sc one;
  sc_zero(one);
  one.v[0] = 1;
  sc t;
  sc_to_mont(t, a);          // a R mod l   (a may be >= l: sc_mont allows it in the first operand)
  sc_mont(r, t, one);        // a
}

// Sign folding for multiscalar multiplication: s * P = (l - s) * (-P).  If (l-1)/2 < s <= l, replaces s by l - s and
// returns 1 (the caller negates the point or the digits); otherwise leaves s alone (also when s > l: non-canonical
// scalars keep their value).  The folded scalar is < 2^252, and the negated 128-bit weights of the batch verifier
// (batch_verifier.rs:183: l - r) come out as r: 128 bits instead of 253, with none of the all-ones digit runs.
ZKP_HD uint32_t sc_fold_sign(uint32_t s[8]) {
  uint32_t d[8];
  uint64_t b1 = 0, b2 = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const uint64_t t1 = (uint64_t)sc_half(i) - s[i] - b1;      // borrow out  <=>  s > half
    b1 = (t1 >> 63) & 1u;
    const uint64_t t2 = (uint64_t)sc_l(i) - s[i] - b2;         // l - s, borrow out  <=>  s > l
    d[i] = (uint32_t)t2;
    b2 = (t2 >> 63) & 1u;
  }
  const bool fold = b1 && !b2;
#pragma unroll
  for (int i = 0; i < 8; ++i) s[i] = fold ? d[i] : s[i];
  return fold ? 1u : 0u;
}

// r = a / 2 mod l for any 256-bit a (reduced first): (a + (a odd ? l : 0)) >> 1
ZKP_HD void sc_halve(sc& r, const sc& a) {
  sc t;
  sc_reduce(t, a);
  const uint32_t odd = 0u - (t.v[0] & 1u);
  uint64_t c = 0;
  uint32_t w[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    c += (uint64_t)t.v[i] + (sc_l(i) & odd);
    w[i] = (uint32_t)c;
    c >>= 32;
  }
#pragma unroll
  for (int i = 0; i < 7; ++i) r.v[i] = (w[i] >> 1) | (w[i + 1] << 31);
  r.v[7] = w[7] >> 1;                         // t + l < 2^254: no carry out
}

// the same for an input that is already canonical (< l): no reduction
ZKP_HD void sc_halve_canonical(sc& r, const sc& t) {
  const uint32_t odd = 0u - (t.v[0] & 1u);
  uint64_t c = 0;
  uint32_t w[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    c += (uint64_t)t.v[i] + (sc_l(i) & odd);
    w[i] = (uint32_t)c;
    c >>= 32;
  }
#pragma unroll
  for (int i = 0; i < 7; ++i) r.v[i] = (w[i] >> 1) | (w[i + 1] << 31);
  r.v[7] = w[7] >> 1;
}

// 512-bit little-endian value (lo + hi * 2^256) -> canonical scalar: Scalar::from_bytes_mod_order_wide
ZKP_HD void sc_from_wide(sc& r, const sc& lo, const sc& hi) {
  sc r1, rr, a, b;
#pragma unroll
  for (int i = 0; i < 8; ++i) { r1.v[i] = sc_r1(i); rr.v[i] = sc_rr(i); }
  sc_mont(a, lo, r1);        // lo * R / R = lo mod l
  sc_mont(b, hi, rr);        // hi * R^2 / R = hi * 2^256 mod l
  sc_add(r, a, b);
}

}  // namespace zkp
