// GF(2^255-19) for gfx950: 9 unsigned limbs of 29 bits (limb 8 nominally 23 bits) in 32-bit VGPRs.
//
// Why this shape (profiles/r01_valu_rates_microbench.txt): on gfx950 v_mad_u64_u32 (32x32+64 -> 64)
// issues at the same rate as v_add_co_u32 / v_addc_co_u32, and fp64 FMA is no faster.  With 29-bit
// limbs every 64-bit column accumulator holds the whole column sum  sum_{i+j=k} a_i*b_j  (8 large
// products < 2^61 each) WITHOUT carry flags, so a field multiplication is a pure chain of
// v_mad_u64_u32 (81 products + 17 fold mads) followed by one short carry chain.  Additions and
// subtractions are limb-wise and lazy (no carry); a "carry" (weak reduction) is inserted only
// where interval analysis says the next multiplication would overflow 64 bits.
//
// Replaces curve25519-dalek 2.x `FieldElement51` (backend::serial::u64::field, not vendored in the
// reference; reference call sites: src/toolbox/verifier.rs:90,97,162, prover.rs:94,
// batch_verifier.rs:219 through RistrettoPoint / CompressedRistretto).
//
// Limb-bound vocabulary used in comments:
//   tight : output of fe_mul / fe_sq / fe_carry.  v[0..7] < 2^29 + 2^18, v[8] < 2^23 + 2^4
//   sum   : tight + tight                      (< 2^30 + ...)
//   diff  : tight + BIAS2P - tight              (< 3 * 2^29)
// fe_mul(a, b) requires  max(a) * max(b) * 8 + 2^46 < 2^64   (e.g. diff x sum, tight x anything).
//
// Compile with -DZKP_FE_TRACK (host only) to carry per-limb upper bounds through every operation
// and assert the requirement above -- tests/host/fe_host_test.cpp does that for every formula used.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define ZKP_HD __host__ __device__ __forceinline__
#else
#define ZKP_HD inline
#endif

#if defined(ZKP_HOST_FE51) && !defined(__HIPCC__)
// the host backend's translation unit: same interface over 5 x 51-bit limbs (host/fe51.h says why and what the contract is)
#include "host/fe51.h"
#else

#ifdef ZKP_FE_TRACK
#include <cassert>
#include <cstdio>
#include <cstdlib>
#endif

namespace zkp {

constexpr uint32_t FE_M29 = (1u << 29) - 1;
constexpr uint32_t FE_M23 = (1u << 23) - 1;

struct fe {
  uint32_t v[9];
#ifdef ZKP_FE_TRACK
  uint64_t ub[9];   // inclusive upper bound of v[i] over all inputs (interval arithmetic)
#endif
};

#ifdef ZKP_FE_TRACK
#define FE_TRACK(stmt) do { stmt; } while (0)
inline void fe_track_fail(const char* what) { fprintf(stderr, "fe bound violation: %s\n", what); abort(); }
inline void fe_set_ub_tight(fe& r) {
  for (int i = 0; i < 8; ++i) r.ub[i] = (1ull << 29) + (1ull << 18);
  r.ub[8] = (1ull << 23) + 16;
}
inline void fe_set_ub_exact(fe& r) { for (int i = 0; i < 9; ++i) r.ub[i] = r.v[i]; }
#else
#define FE_TRACK(stmt) do { } while (0)
#endif

// ---------------------------------------------------------------------------------------------
// constants
// ---------------------------------------------------------------------------------------------
ZKP_HD void fe_0(fe& r) {
#pragma unroll
  for (int i = 0; i < 9; ++i) r.v[i] = 0;
  FE_TRACK(fe_set_ub_exact(r));
}
ZKP_HD void fe_1(fe& r) {
  fe_0(r);
  r.v[0] = 1;
  FE_TRACK(fe_set_ub_exact(r));
}
ZKP_HD void fe_copy(fe& r, const fe& a) { r = a; }

// 2p = 2^256 - 38 written with every limb >= any tight limb:  l0 = 2^30-38, l1..7 = 2^30-2, l8 = 2^24-2
// 4p likewise with limbs ~2^31 (for subtracting "sum"/"diff" class values).
ZKP_HD uint32_t fe_bias2p(int i) { return i == 0 ? 0x3fffffdau : (i == 8 ? 0x00fffffeu : 0x3ffffffeu); }
ZKP_HD uint32_t fe_bias4p(int i) { return i == 0 ? 0x7fffffb4u : (i == 8 ? 0x01fffffcu : 0x7ffffffcu); }

// ---------------------------------------------------------------------------------------------
// lazy add / sub / neg
// ---------------------------------------------------------------------------------------------
ZKP_HD void fe_add(fe& r, const fe& a, const fe& b) {
#pragma unroll
  for (int i = 0; i < 9; ++i) {
    FE_TRACK(if (a.ub[i] + b.ub[i] > 0xffffffffull) fe_track_fail("fe_add overflow"); r.ub[i] = a.ub[i] + b.ub[i]);
    r.v[i] = a.v[i] + b.v[i];
  }
}

// r = a - b, b of class tight (or anything with limbs <= bias2p)
ZKP_HD void fe_sub(fe& r, const fe& a, const fe& b) {
#pragma unroll
  for (int i = 0; i < 9; ++i) {
    FE_TRACK(if (b.ub[i] > fe_bias2p(i)) fe_track_fail("fe_sub: subtrahend exceeds 2p bias");
             if (a.ub[i] + fe_bias2p(i) > 0xffffffffull) fe_track_fail("fe_sub overflow");
             r.ub[i] = a.ub[i] + fe_bias2p(i));
    r.v[i] = a.v[i] + (fe_bias2p(i) - b.v[i]);
  }
}

// r = a - b, b with limbs <= bias4p (sum / diff class); result must be carried before a multiplication
ZKP_HD void fe_sub4(fe& r, const fe& a, const fe& b) {
#pragma unroll
  for (int i = 0; i < 9; ++i) {
    FE_TRACK(if (b.ub[i] > fe_bias4p(i)) fe_track_fail("fe_sub4: subtrahend exceeds 4p bias");
             if (a.ub[i] + fe_bias4p(i) > 0xffffffffull) fe_track_fail("fe_sub4 overflow");
             r.ub[i] = a.ub[i] + fe_bias4p(i));
    r.v[i] = a.v[i] + (fe_bias4p(i) - b.v[i]);
  }
}

ZKP_HD void fe_neg(fe& r, const fe& a) {
#pragma unroll
  for (int i = 0; i < 9; ++i) {
    FE_TRACK(if (a.ub[i] > fe_bias2p(i)) fe_track_fail("fe_neg: operand exceeds 2p bias"); r.ub[i] = fe_bias2p(i));
    r.v[i] = fe_bias2p(i) - a.v[i];
  }
}

// weak reduction, all limbs in parallel (27 cheap 32-bit VALU ops, no carry chain):
// result limbs < 2^29 + 2^14  (tight class)
ZKP_HD void fe_carry(fe& r, const fe& a) {
  uint32_t c[9];
#pragma unroll
  for (int i = 0; i < 8; ++i) c[i] = a.v[i] >> 29;
  c[8] = a.v[8] >> 23;
  r.v[0] = (a.v[0] & FE_M29) + 19u * c[8];
#pragma unroll
  for (int i = 1; i < 8; ++i) r.v[i] = (a.v[i] & FE_M29) + c[i - 1];
  r.v[8] = (a.v[8] & FE_M23) + c[7];
  FE_TRACK(fe_set_ub_tight(r));
}

// select: r = flag ? b : r   (flag is 0/1).  Written as a per-limb select so that gfx950 gets one v_cndmask_b32 per
// limb (the masked-xor form costs three VALU ops per limb, and the table scans of the term kernels are made of
// these); v_cndmask has no data-dependent timing, and no branch is involved.
ZKP_HD void fe_cmov(fe& r, const fe& b, uint32_t flag) {
  const bool f = flag != 0;
#pragma unroll
  for (int i = 0; i < 9; ++i) {
    FE_TRACK(if (b.ub[i] > r.ub[i]) r.ub[i] = b.ub[i]);
    r.v[i] = f ? b.v[i] : r.v[i];
  }
}
ZKP_HD void fe_cswap(fe& a, fe& b, uint32_t flag) {
  const bool f = flag != 0;
#pragma unroll
  for (int i = 0; i < 9; ++i) {
    FE_TRACK(uint64_t mx = a.ub[i] > b.ub[i] ? a.ub[i] : b.ub[i]; a.ub[i] = mx; b.ub[i] = mx);
    const uint32_t x = a.v[i], y = b.v[i];
    a.v[i] = f ? y : x;
    b.v[i] = f ? x : y;
  }
}

// ---------------------------------------------------------------------------------------------
// multiplication: 81 + 17 v_mad_u64_u32, one 9-step carry chain
// ---------------------------------------------------------------------------------------------
// shared tail: columns c[0..8] (already containing the folded high half) -> tight limbs
ZKP_HD void fe_reduce_columns(fe& r, uint64_t c[9]) {
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    c[k + 1] += c[k] >> 29;
    r.v[k] = (uint32_t)c[k] & FE_M29;
  }
  r.v[8] = (uint32_t)c[8] & FE_M23;
  const uint64_t t = c[8] >> 23;                       // < 2^41, weight 2^255 == 19
  const uint64_t c0 = (uint64_t)r.v[0] + 19ull * (uint32_t)t;   // < 2^37
  r.v[0] = (uint32_t)c0 & FE_M29;
  r.v[1] += (uint32_t)(c0 >> 29) + 152u * (uint32_t)(t >> 32);   // 2^32 * 19 = 152 * 2^29 * ... limb 1
  FE_TRACK(fe_set_ub_tight(r));
}

#ifdef ZKP_FE_TRACK
inline void fe_track_mul(const fe& a, const fe& b) {
  // full 17-column bound, then the fold, in 128-bit arithmetic
  unsigned __int128 col[17];
  for (int k = 0; k < 17; ++k) col[k] = 0;
  for (int i = 0; i < 9; ++i) if (a.ub[i] >= (1ull << 31) || b.ub[i] >= (1ull << 31)) fe_track_fail("fe_mul: limb >= 2^31");
  for (int i = 0; i < 9; ++i)
    for (int j = 0; j < 9; ++j) col[i + j] += (unsigned __int128)a.ub[i] * b.ub[j];
  const unsigned __int128 lim = ((unsigned __int128)1 << 64) - 1;
  for (int k = 9; k < 17; ++k) if (col[k] > lim) fe_track_fail("fe_mul: high column overflows 64 bits");
  for (int k = 0; k < 9; ++k) {
    unsigned __int128 t = col[k];
    if (k + 9 <= 16) t += (unsigned __int128)1216 * 0xffffffffull;
    if (k + 8 >= 9) t += (unsigned __int128)9728 * 0xffffffffull;
    t += (unsigned __int128)1 << 36;   // incoming carry
    if (t > lim) fe_track_fail("fe_mul: low column overflows 64 bits");
  }
}
#endif

ZKP_HD void fe_mul(fe& r, const fe& a, const fe& b) {
  FE_TRACK(fe_track_mul(a, b));
  // Row-major (operand scanning) issue order: consecutive v_mad_u64_u32 hit nine different 64-bit
  // accumulators, so no mad waits on the previous one.  Measured on MI355X
  // (tools/microbench/fe_mul_sched.hip): 234 ns vs 298 ns per dependent multiplication for a lone wave,
  // 300 vs 293 G mul/s chip-wide at 8 waves/SIMD, against the column-major order.
  uint64_t c[17];
#pragma unroll
  for (int k = 0; k < 17; ++k) c[k] = 0;
#pragma unroll
  for (int i = 0; i < 9; ++i) {
#pragma unroll
    for (int j = 0; j < 9; ++j) c[i + j] += (uint64_t)a.v[i] * b.v[j];
  }
  // fold the high columns 9..16, split into 32-bit halves, with 2^261 == 1216 (mod p):
  //   2^(29k)       == 1216 * 2^(29(k-9))
  //   2^(29k + 32)  == 9728 * 2^(29(k-8))
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    c[k] += 1216ull * (uint32_t)c[k + 9];
    c[k + 1] += 9728ull * (uint32_t)(c[k + 9] >> 32);
  }
  fe_reduce_columns(r, c);
}

ZKP_HD void fe_sq(fe& r, const fe& a) {
  FE_TRACK(fe_track_mul(a, a));
  uint32_t a2[9];
#pragma unroll
  for (int i = 0; i < 9; ++i) a2[i] = a.v[i] << 1;    // requires a.v[i] < 2^31 (checked by fe_track_mul)
  uint64_t c[9];
  uint64_t h[8];
#pragma unroll
  for (int k = 9; k < 17; ++k) {
    uint64_t acc = 0;
#pragma unroll
    for (int i = k - 8; 2 * i < k; ++i) acc += (uint64_t)a2[i] * a.v[k - i];
    if ((k & 1) == 0) acc += (uint64_t)a.v[k / 2] * a.v[k / 2];
    h[k - 9] = acc;
  }
#pragma unroll
  for (int k = 0; k < 9; ++k) {
    uint64_t acc = 0;
#pragma unroll
    for (int i = 0; 2 * i < k; ++i) acc += (uint64_t)a2[i] * a.v[k - i];
    if ((k & 1) == 0) acc += (uint64_t)a.v[k / 2] * a.v[k / 2];
    if (k <= 7) acc += 1216ull * (uint32_t)h[k];
    if (k >= 1) acc += 9728ull * (uint32_t)(h[k - 1] >> 32);
    c[k] = acc;
  }
  fe_reduce_columns(r, c);
}

// r = a^(2^n), n >= 1; a real loop (not unrolled) so the inverse-square-root chain stays small in I-cache
ZKP_HD void fe_sqn(fe& r, const fe& a, int n) {
  r = a;
#pragma unroll 1
  for (int i = 0; i < n; ++i) fe_sq(r, r);
}

// ---------------------------------------------------------------------------------------------
// bytes <-> limbs
// ---------------------------------------------------------------------------------------------
// w[0..7] = the 32 bytes as little-endian 32-bit words; bit 255 is IGNORED (callers that need
// canonicity use fe_words_canonical).
ZKP_HD void fe_fromwords(fe& r, const uint32_t w[8]) {
#pragma unroll
  for (int i = 0; i < 9; ++i) {
    const int bit = 29 * i, wi = bit >> 5, sh = bit & 31;
    uint32_t x = w[wi] >> sh;
    if (sh > 3 && wi < 7) x |= w[wi + 1] << (32 - sh);
    r.v[i] = x & (i == 8 ? FE_M23 : FE_M29);
  }
  FE_TRACK(fe_set_ub_tight(r));
}

// 1 iff the 256-bit little-endian integer in w is < p (so also bit 255 clear)
ZKP_HD uint32_t fe_words_canonical(const uint32_t w[8]) {
  uint32_t all_ones = w[1] & w[2] & w[3] & w[4] & w[5] & w[6];
  const uint32_t ge_p = (w[7] == 0x7fffffffu) & (all_ones == 0xffffffffu) & (w[0] >= 0xffffffedu);
  return (uint32_t)((w[7] >> 31) == 0) & (ge_p ^ 1u);
}

// canonical little-endian words of a (any limbs < 2^32)
ZKP_HD void fe_towords(uint32_t w[8], const fe& a) {
  uint32_t t[9];
#pragma unroll
  for (int i = 0; i < 9; ++i) {
    FE_TRACK(if (a.ub[i] > 0xfffffff0ull) fe_track_fail("fe_towords: limb too large"));
    t[i] = a.v[i];
  }
  // two sequential weak passes bring the value into [0, 2^255 + 19*2^10)
#pragma unroll
  for (int pass = 0; pass < 2; ++pass) {
#pragma unroll
    for (int i = 0; i < 8; ++i) { t[i + 1] += t[i] >> 29; t[i] &= FE_M29; }
    const uint32_t c = t[8] >> 23;
    t[8] &= FE_M23;
    t[0] += 19u * c;
  }
  // now V = sum t[i] 2^(29i) < 2^255 + 19 < 2p (t[0] may hold a pending carry of a few units).
  // V >= p  <=>  u = V + 19 >= 2^255, and then V - p = u - 2^255.
  {
    uint32_t u[9], s[9];
    u[0] = t[0] + 19u;
    s[0] = t[0];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      u[i + 1] = t[i + 1] + (u[i] >> 29); u[i] &= FE_M29;
      s[i + 1] = t[i + 1] + (s[i] >> 29); s[i] &= FE_M29;
    }
    const uint32_t q = u[8] >> 23;              // 0 or 1
    u[8] &= FE_M23;
    const uint32_t m = 0u - q;
#pragma unroll
    for (int i = 0; i < 9; ++i) t[i] = (u[i] & m) | (s[i] & ~m);
  }
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    // word j = bits [32j, 32j+32)
    const int lo = (32 * j) / 29, sh = 32 * j - 29 * lo;
    uint32_t x = t[lo] >> sh;
    x |= t[lo + 1] << (29 - sh);
    if (29 - sh + 29 < 32 && lo + 2 < 9) x |= t[lo + 2] << (58 - sh);
    w[j] = x;
  }
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
#pragma unroll
  for (int i = 0; i < 8; ++i) x |= w[i];
  return (uint32_t)(x == 0);
}
ZKP_HD uint32_t fe_equal(const fe& a, const fe& b) {
  uint32_t wa[8], wb[8];
  fe_towords(wa, a);
  fe_towords(wb, b);
  uint32_t x = 0;
#pragma unroll
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

// ---------------------------------------------------------------------------------------------
// z^(2^252 - 3) = z^((p-5)/8): 251 squarings + 11 multiplications
// ---------------------------------------------------------------------------------------------
ZKP_HD void fe_pow22523(fe& out, const fe& z) {
  fe t0, t1, t2, t3;
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
  (void)t3;
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

// field constants as limbs (generated by tools/gen_constants.py; checked by tests/host)
struct fe_const { uint32_t v[9]; };
ZKP_HD void fe_from_const(fe& r, const fe_const& c) {
#pragma unroll
  for (int i = 0; i < 9; ++i) r.v[i] = c.v[i];
  FE_TRACK(fe_set_ub_exact(r));
}

}  // namespace zkp

#endif  // ZKP_HOST_FE51
