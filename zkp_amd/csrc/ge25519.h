// Edwards25519 (a = -1) group arithmetic in extended coordinates and the ristretto255 codec on top
// of fe25519.h.  Formulas: RFC 9496 section 4.3 (decode/encode) and the Hisil-Wong-Carter-Dawson
// unified addition / dedicated doubling.  Every fe_carry below is there because the interval
// tracker (ZKP_FE_TRACK, tests/host) says the following multiplication would otherwise overflow.
//
// Replaces curve25519-dalek 2.x EdwardsPoint / RistrettoPoint / CompressedRistretto as reached from
// the reference at src/toolbox/verifier.rs:90 (decompress), mod.rs:180,204 (compress),
// prover.rs:94, verifier.rs:97,162, batch_verifier.rs:219 (point additions inside the MSMs).
#pragma once
#include "fe25519.h"
#include "fe_constants.h"

namespace zkp {

struct ge_p3 { fe X, Y, Z, T; };             // extended: x = X/Z, y = Y/Z, T = XY/Z
struct ge_cached { fe YpX, YmX, Z2, T2d; };  // (Y+X, Y-X, 2Z, 2dT), all tight
struct ge_niels { fe ypx, ymx, xy2d; };      // affine (Z = 1): (y+x, y-x, 2dxy), all tight

ZKP_HD void ge_identity(ge_p3& r) {
  fe_0(r.X);
  fe_1(r.Y);
  fe_1(r.Z);
  fe_0(r.T);
}
ZKP_HD void ge_cached_identity(ge_cached& r) {
  fe_1(r.YpX);
  fe_1(r.YmX);
  fe_0(r.Z2);
  r.Z2.v[0] = 2;
  FE_TRACK(fe_set_ub_exact(r.Z2));
  fe_0(r.T2d);
}
ZKP_HD void ge_niels_identity(ge_niels& r) {
  fe_1(r.ypx);
  fe_1(r.ymx);
  fe_0(r.xy2d);
}

ZKP_HD void ge_to_cached(ge_cached& r, const ge_p3& p) {
  fe t, d2;
  fe_add(t, p.Y, p.X);
  fe_carry(r.YpX, t);
  fe_sub(t, p.Y, p.X);
  fe_carry(r.YmX, t);
  fe_add(t, p.Z, p.Z);
  fe_carry(r.Z2, t);
  fe_from_const(d2, FE_D2);
  fe_mul(r.T2d, p.T, d2);
}

// p must have Z = 1 (freshly decoded point)
ZKP_HD void ge_affine_to_niels(ge_niels& r, const ge_p3& p) {
  fe t, d2;
  fe_add(t, p.Y, p.X);
  fe_carry(r.ypx, t);
  fe_sub(t, p.Y, p.X);
  fe_carry(r.ymx, t);
  fe_from_const(d2, FE_D2);
  fe_mul(r.xy2d, p.T, d2);
}

// r = p + q (8M + 1 carry).  p: coordinates tight.
ZKP_HD void ge_add_cached(ge_p3& r, const ge_p3& p, const ge_cached& q) {
  fe a, b, c, d, e, f, g, h, t;
  fe_sub(t, p.Y, p.X);
  fe_mul(a, t, q.YmX);
  fe_add(t, p.Y, p.X);
  fe_mul(b, t, q.YpX);
  fe_mul(c, p.T, q.T2d);
  fe_mul(d, p.Z, q.Z2);
  fe_sub(e, b, a);          // diff
  fe_sub(t, d, c);          // diff
  fe_carry(f, t);           // tight, so that e*f (diff x diff otherwise) fits
  fe_add(g, d, c);          // sum
  fe_add(h, b, a);          // sum
  fe_mul(r.X, e, f);
  fe_mul(r.Y, g, h);
  fe_mul(r.Z, f, g);
  fe_mul(r.T, e, h);
}

// r = p - q
ZKP_HD void ge_sub_cached(ge_p3& r, const ge_p3& p, const ge_cached& q) {
  fe a, b, c, d, e, f, g, h, t;
  fe_sub(t, p.Y, p.X);
  fe_mul(a, t, q.YpX);
  fe_add(t, p.Y, p.X);
  fe_mul(b, t, q.YmX);
  fe_mul(c, p.T, q.T2d);
  fe_mul(d, p.Z, q.Z2);
  fe_sub(e, b, a);
  fe_add(t, d, c);          // sum
  fe_carry(f, t);
  fe_sub(g, d, c);          // diff
  fe_add(h, b, a);
  fe_mul(r.X, e, f);
  fe_mul(r.Y, g, h);
  fe_mul(r.Z, f, g);
  fe_mul(r.T, e, h);
}

// r = p + q, q affine niels (7M + 1 carry)
ZKP_HD void ge_madd(ge_p3& r, const ge_p3& p, const ge_niels& q) {
  fe a, b, c, d, e, f, g, h, t;
  fe_sub(t, p.Y, p.X);
  fe_mul(a, t, q.ymx);
  fe_add(t, p.Y, p.X);
  fe_mul(b, t, q.ypx);
  fe_mul(c, p.T, q.xy2d);
  fe_add(d, p.Z, p.Z);      // sum
  fe_sub(e, b, a);
  fe_sub(t, d, c);
  fe_carry(f, t);
  fe_add(g, d, c);          // sum + tight  (3 * 2^29): g*h and f*g still fit
  fe_add(h, b, a);
  fe_mul(r.X, e, f);
  fe_mul(r.Y, g, h);
  fe_mul(r.Z, f, g);
  fe_mul(r.T, e, h);
}

// r = p - q, q affine niels
ZKP_HD void ge_msub(ge_p3& r, const ge_p3& p, const ge_niels& q) {
  fe a, b, c, d, e, f, g, h, t;
  fe_sub(t, p.Y, p.X);
  fe_mul(a, t, q.ypx);
  fe_add(t, p.Y, p.X);
  fe_mul(b, t, q.ymx);
  fe_mul(c, p.T, q.xy2d);
  fe_add(d, p.Z, p.Z);
  fe_sub(e, b, a);
  fe_add(t, d, c);
  fe_carry(f, t);
  fe_sub(t, d, c);
  fe_carry(g, t);
  fe_add(h, b, a);
  fe_mul(r.X, e, f);
  fe_mul(r.Y, g, h);
  fe_mul(r.Z, f, g);
  fe_mul(r.T, e, h);
}

// r = 2p (4S + 4M + 2 carries); only X, Y, Z of p are read.  WITH_T = false skips T (3M) when the
// result feeds another doubling.
template <bool WITH_T = true>
ZKP_HD void ge_double(ge_p3& r, const ge_p3& p) {
  fe xx, yy, zz2, xpy, e, g, f, h, t;
  fe_sq(xx, p.X);
  fe_sq(yy, p.Y);
  fe_sq(t, p.Z);
  fe_add(zz2, t, t);        // sum
  fe_add(t, p.X, p.Y);      // sum
  fe_sq(xpy, t);
  fe_add(h, yy, xx);        // sum   (completed Y)
  fe_sub(g, yy, xx);        // diff  (completed Z)
  fe_sub4(t, xpy, h);       // tight + 4p - sum
  fe_carry(e, t);           // completed X
  fe_sub4(t, zz2, g);       // sum + 4p - diff
  fe_carry(f, t);           // completed T
  fe_mul(r.X, e, f);
  fe_mul(r.Y, h, g);
  fe_mul(r.Z, g, f);
  if (WITH_T) fe_mul(r.T, e, h);
}

// acc = 16 acc as a ROLLED loop: one doubling body (8 KB of code) instead of four.  The walks that use it share the CU's
// instruction cache with every other kernel in flight.
ZKP_HD void ge_double4(ge_p3& acc) {
#pragma unroll 1
  for (int k = 0; k < 3; ++k) ge_double<false>(acc, acc);
  ge_double<true>(acc, acc);
}

ZKP_HD void ge_double4_flat(ge_p3& acc) {
  ge_double<false>(acc, acc);
  ge_double<false>(acc, acc);
  ge_double<false>(acc, acc);
  ge_double<true>(acc, acc);
}

ZKP_HD void ge_neg(ge_p3& r, const ge_p3& p) {
  fe t;
  fe_neg(t, p.X);
  fe_carry(r.X, t);
  r.Y = p.Y;
  r.Z = p.Z;
  fe_neg(t, p.T);
  fe_carry(r.T, t);
}

// conditional negation of cached / niels forms: swap (Y+X, Y-X), negate the T term.  flag in {0,1}.
// The negated term is NOT carried (round 5): 2p - x has limbs <= 2^30, and the one place it goes is the product with the accumulator's tight T
// (c = T * T2d / T * xy2d), whose columns stay below 2^64 with a 2p-class operand -- the interval tracker checks exactly these compositions
// (tests/host/fe_host_lib.cpp: t_point_op 6 and 7).  27 instructions per table addition less than with the carry rounds 1 - 4 paid.
ZKP_HD void ge_cached_cneg(ge_cached& q, uint32_t flag) {
  fe_cswap(q.YpX, q.YmX, flag);
  fe n;
  fe_neg(n, q.T2d);
  fe_cmov(q.T2d, n, flag);
}
ZKP_HD void ge_niels_cneg(ge_niels& q, uint32_t flag) {
  fe_cswap(q.ypx, q.ymx, flag);
  fe n;
  fe_neg(n, q.xy2d);
  fe_cmov(q.xy2d, n, flag);
}
ZKP_HD void ge_cached_cmov(ge_cached& r, const ge_cached& q, uint32_t flag) {
  fe_cmov(r.YpX, q.YpX, flag);
  fe_cmov(r.YmX, q.YmX, flag);
  fe_cmov(r.Z2, q.Z2, flag);
  fe_cmov(r.T2d, q.T2d, flag);
}

// ---------------------------------------------------------------------------------------------
// SQRT_RATIO_M1(1, v) (RFC 9496 section 4.2 with u = 1 -- the only form decode and encode need).
// v tight.  Returns was_square; r = |1/sqrt(v)| or |1/sqrt(i v)|.
// ---------------------------------------------------------------------------------------------
ZKP_HD uint32_t fe_words_eq(const uint32_t w[8], const fe_words& c) {
  uint32_t x = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) x |= w[i] ^ c.w[i];
  return (uint32_t)(x == 0);
}

ZKP_HD uint32_t fe_invsqrt(fe& r, const fe& v) {
  fe t, v3, c;
  fe_sq(t, v);
  fe_mul(v3, t, v);              // v^3
  fe_sq(t, v3);
  fe_mul(t, t, v);               // v^7
  fe_pow22523(t, t);             // (v^7)^((p-5)/8)
  fe_mul(r, t, v3);              // r = v^3 (v^7)^((p-5)/8)
  fe_sq(t, r);
  fe_mul(t, t, v);               // check = v r^2
  uint32_t cw[8];
  fe_towords(cw, t);
  const uint32_t correct = fe_words_eq(cw, FE_W_ONE);
  const uint32_t flipped = fe_words_eq(cw, FE_W_MINUS_ONE);
  const uint32_t flipped_i = fe_words_eq(cw, FE_W_MINUS_SQRT_M1);
  fe_from_const(c, FE_SQRT_M1);
  fe_mul(t, r, c);
  fe_cmov(r, t, flipped | flipped_i);
  fe_abs(r, r);
  return correct | flipped;
}

// ---------------------------------------------------------------------------------------------
// ristretto255 decode: 32 bytes (as 8 LE words) -> extended point with Z = 1; returns 1 if valid.
// On failure the output is the identity (so downstream arithmetic stays defined).
// ---------------------------------------------------------------------------------------------
ZKP_HD uint32_t ristretto_decode(ge_p3& r, const uint32_t w[8]) {
  fe s, ss, u1, u2, u2s, v, t, one, I, dx, dy, d;
  const uint32_t canonical = fe_words_canonical(w) & ((w[0] & 1u) ^ 1u);
  fe_fromwords(s, w);
  fe_1(one);
  fe_sq(ss, s);
  fe_sub(u1, one, ss);           // 1 - s^2   (diff)
  fe_add(u2, one, ss);           // 1 + s^2   (sum)
  fe_sq(u2s, u2);
  fe_carry(t, u1);
  fe_sq(v, t);                   // u1^2
  fe_from_const(d, FE_D);
  fe_mul(v, v, d);               // d u1^2
  fe_neg(t, v);
  fe_sub(t, t, u2s);             // -d u1^2 - u2^2  (< 2^31)
  fe_carry(v, t);
  fe_mul(t, v, u2s);
  const uint32_t ok = fe_invsqrt(I, t);
  fe_mul(dx, I, u2);             // den_x
  fe_mul(dy, I, dx);
  fe_mul(dy, dy, v);             // den_y
  fe_add(t, s, s);
  fe_mul(t, t, dx);
  fe_abs(r.X, t);                // x = |2 s den_x|    (limbs <= bias2p)
  fe_carry(r.X, r.X);
  fe_mul(r.Y, u1, dy);           // y = u1 den_y
  fe_1(r.Z);
  fe_mul(r.T, r.X, r.Y);
  const uint32_t valid = canonical & ok & (fe_isnegative(r.T) ^ 1u) & (fe_iszero(r.Y) ^ 1u);
  ge_p3 id;
  ge_identity(id);
  fe_cmov(r.X, id.X, valid ^ 1u);
  fe_cmov(r.Y, id.Y, valid ^ 1u);
  fe_cmov(r.T, id.T, valid ^ 1u);
  return valid;
}

// ristretto255 encode: extended point -> canonical 32 bytes (8 LE words)
ZKP_HD void ristretto_encode(uint32_t w[8], const ge_p3& p) {
  fe u1, u2, t, t2, I, den1, den2, zinv, ix, iy, ench, x, y, dinv, sqrt_m1, c;
  fe_add(t, p.Z, p.Y);
  fe_sub(t2, p.Z, p.Y);
  fe_mul(u1, t, t2);             // (Z+Y)(Z-Y)
  fe_mul(u2, p.X, p.Y);
  fe_sq(t, u2);
  fe_mul(t, t, u1);
  fe_invsqrt(I, t);              // invsqrt(u1 u2^2)
  fe_mul(den1, I, u1);
  fe_mul(den2, I, u2);
  fe_mul(t, den1, den2);
  fe_mul(zinv, t, p.T);
  fe_from_const(sqrt_m1, FE_SQRT_M1);
  fe_mul(ix, p.X, sqrt_m1);
  fe_mul(iy, p.Y, sqrt_m1);
  fe_from_const(c, FE_INVSQRT_A_MINUS_D);
  fe_mul(ench, den1, c);
  fe_mul(t, p.T, zinv);
  const uint32_t rotate = fe_isnegative(t);
  x = p.X;
  y = p.Y;
  dinv = den2;
  fe_cmov(x, iy, rotate);
  fe_cmov(y, ix, rotate);
  fe_cmov(dinv, ench, rotate);
  fe_mul(t, x, zinv);
  const uint32_t neg_y = fe_isnegative(t);
  fe_neg(t, y);
  fe_carry(t, t);
  fe_cmov(y, t, neg_y);
  fe_sub(t, p.Z, y);
  fe_mul(t, dinv, t);
  fe_abs(t, t);
  fe_towords(w, t);
}

// ---- encode(2 P) with a plain inversion instead of an inverse square root ---------------------------------------------------
// The encoding of a DOUBLED point needs no square root: with e = 2XY, f = Z^2 + dT^2, g = Y^2 + X^2, h = Z^2 - dT^2 (the
// numerators / denominators of the doubling) everything reduces to 1 / (e g f h), and inversions batch (Montgomery's
// trick) where inverse square roots do not.  curve25519-dalek uses the same identity in
// RistrettoPoint::double_and_compress_batch.  The MSM kernels therefore compute H = sum (s_i / 2) P_i and emit encode(2 H):
// ~25 field multiplications per output instead of ~265.
//   ristretto_dc_prepare : state of P and x = e g f h   (x = 0 iff 2P is in the identity coset: caller falls back)
//   ristretto_dc_finish  : the 32 bytes from the state and 1 / x
struct ristretto_dc_state { fe e, f, g, h, eg, fh; };

ZKP_HD void ristretto_dc_prepare(ristretto_dc_state& s, fe& x, const ge_p3& p) {
  fe xx, yy, zz, dtt, t, d;
  fe_sq(xx, p.X);
  fe_sq(yy, p.Y);
  fe_sq(zz, p.Z);
  fe_sq(t, p.T);
  fe_from_const(d, FE_D);
  fe_mul(dtt, t, d);
  fe_add(t, p.Y, p.Y);
  fe_mul(s.e, p.X, t);           // 2 X Y
  fe_add(t, zz, dtt);
  fe_carry(s.f, t);              // Z^2 + d T^2
  fe_add(t, yy, xx);
  fe_carry(s.g, t);              // Y^2 - a X^2
  fe_sub(t, zz, dtt);
  fe_carry(s.h, t);              // Z^2 - d T^2
  fe_mul(s.eg, s.e, s.g);
  fe_mul(s.fh, s.f, s.h);
  fe_mul(x, s.eg, s.fh);
}

ZKP_HD void ristretto_dc_finish(uint32_t w[8], const ristretto_dc_state& s, const fe& inv) {
  fe zinv, tinv, magic, sqrt_m1, t, e, g, h, me, fs, mg;
  fe_mul(zinv, s.eg, inv);       // 1 / (f h)
  fe_mul(tinv, s.fh, inv);       // 1 / (e g)
  fe_mul(t, s.eg, zinv);
  const uint32_t rotate = fe_isnegative(t);
  e = s.e;
  g = s.g;
  h = s.h;
  fe_neg(t, s.e);
  fe_carry(me, t);
  fe_from_const(sqrt_m1, FE_SQRT_M1);
  fe_mul(fs, s.f, sqrt_m1);
  fe_from_const(magic, FE_INVSQRT_A_MINUS_D);
  fe_cmov(e, s.g, rotate);
  fe_cmov(g, me, rotate);
  fe_cmov(h, fs, rotate);
  fe_cmov(magic, sqrt_m1, rotate);
  fe_mul(t, h, e);
  fe_mul(t, t, zinv);
  const uint32_t neg_g = fe_isnegative(t);
  fe_neg(t, g);
  fe_carry(t, t);
  fe_cmov(g, t, neg_g);
  fe_mul(mg, g, tinv);
  fe_mul(mg, magic, mg);
  fe_sub(t, h, g);
  fe_mul(t, t, mg);
  fe_abs(t, t);
  fe_towords(w, t);
}

}  // namespace zkp
