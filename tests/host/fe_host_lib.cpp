// Host build of the DEVICE field/group headers (same source the HIP kernels compile), exposed through
// a tiny C interface so pytest can compare every primitive with the big-integer oracle on CPU.
// Built three times by tests/test_host_field.py: plain, with -DZKP_FE_TRACK (interval bound tracker:
// aborts if any lazy add/sub chain could overflow a 64-bit column or a 32-bit limb), and with -DZKP_HOST_FE51 (the same point
// formulas and codec over the host backend's 5 x 51-bit field, zkp_amd/csrc/host/fe51.h).
#include "../../zkp_amd/csrc/ge25519.h"
#include "../../zkp_amd/csrc/sc25519.h"
#include <cstring>
using namespace zkp;

static void load(fe& r, const uint8_t* b) { uint32_t w[8]; memcpy(w, b, 32); fe_fromwords(r, w); }
static void store(uint8_t* b, const fe& a) { uint32_t w[8]; fe_towords(w, a); memcpy(b, w, 32); }

extern "C" {
// op: 0 mul, 1 sq, 2 add, 3 sub, 4 neg, 5 (a-b)*(a+b) lazy chain, 6 pow22523, 7 carry(a+b+a)
void t_fe_binop(int op, const uint8_t* a, const uint8_t* b, uint8_t* out) {
  fe x, y, r, t, u;
  load(x, a); load(y, b);
  switch (op) {
    case 0: fe_mul(r, x, y); break;
    case 1: fe_sq(r, x); break;
    case 2: fe_add(r, x, y); break;
    case 3: fe_sub(r, x, y); break;
    case 4: fe_neg(r, x); break;
    case 5: fe_sub(t, x, y); fe_add(u, x, y); fe_mul(r, t, u); break;
    case 6: fe_pow22523(r, x); break;
    case 7: fe_add(t, x, y); fe_add(t, t, x); fe_carry(r, t); break;
    default: fe_0(r);
  }
  store(out, r);
}
int t_fe_canonical(const uint8_t* a) { uint32_t w[8]; memcpy(w, a, 32); return (int)fe_words_canonical(w); }
// raw limbs in, canonical bytes out: exercises fe_towords on non-normalised limbs
#ifdef ZKP_HOST_FE51
int t_is_fe51(void) { return 1; }
void t_fe51_towords_raw(const uint64_t* limbs, uint8_t* out) {          // the host backend's field: five limbs, any value below 2^63 each
  fe x; for (int i = 0; i < 5; ++i) x.v[i] = limbs[i];
  store(out, x);
}
// constants travel as 9 x 29-bit limbs (fe_constants.h): the repacking, byte for byte
void t_fe51_const(int which, uint8_t* out) {
  fe x;
  fe_from_const(x, which == 0 ? FE_D : which == 1 ? FE_D2 : which == 2 ? FE_SQRT_M1 : FE_INVSQRT_A_MINUS_D);
  store(out, x);
}
#else
int t_is_fe51(void) { return 0; }
void t_fe_towords_raw(const uint32_t* limbs, uint8_t* out) {
  fe x; for (int i = 0; i < 9; ++i) x.v[i] = limbs[i];
#ifdef ZKP_FE_TRACK
  for (int i = 0; i < 9; ++i) x.ub[i] = limbs[i];
#endif
  store(out, x);
}
#endif
int t_decode(const uint8_t* enc, uint8_t* xyzt /*4x32*/) {
  uint32_t w[8]; memcpy(w, enc, 32);
  ge_p3 p; const int ok = (int)ristretto_decode(p, w);
  store(xyzt, p.X); store(xyzt + 32, p.Y); store(xyzt + 64, p.Z); store(xyzt + 96, p.T);
  return ok;
}
static int dec(ge_p3& p, const uint8_t* enc) { uint32_t w[8]; memcpy(w, enc, 32); return (int)ristretto_decode(p, w); }
static void enc(uint8_t* out, const ge_p3& p) { uint32_t w[8]; ristretto_encode(w, p); memcpy(out, w, 32); }
int t_recode(const uint8_t* in, uint8_t* out) { ge_p3 p; int ok = dec(p, in); enc(out, p); return ok; }
// out = encode(op(P, Q)): 0 add via cached, 1 sub via cached, 2 madd via niels(Q), 3 msub, 4 double(P), 5 double x3 (no-T chain) + add
int t_point_op(int op, const uint8_t* pe, const uint8_t* qe, uint8_t* out) {
  ge_p3 p, q, r; int ok = dec(p, pe) & dec(q, qe);
  ge_cached c; ge_niels n;
  switch (op) {
    case 0: ge_to_cached(c, q); ge_add_cached(r, p, c); break;
    case 1: ge_to_cached(c, q); ge_sub_cached(r, p, c); break;
    case 2: ge_affine_to_niels(n, q); ge_madd(r, p, n); break;
    case 3: ge_affine_to_niels(n, q); ge_msub(r, p, n); break;
    case 4: ge_double<true>(r, p); break;
    case 5: ge_double<false>(r, p); ge_double<false>(r, r); ge_double<true>(r, r); ge_to_cached(c, q); ge_add_cached(r, r, c); break;
    case 6: ge_to_cached(c, q); ge_cached_cneg(c, 1); ge_add_cached(r, p, c); break;
    case 7: ge_affine_to_niels(n, q); ge_niels_cneg(n, 1); ge_madd(r, p, n); ge_madd(r, r, n); ge_msub(r, r, n); break;
    default: ge_identity(r);
  }
  enc(out, r);
  return ok;
}
// out = encode(2 P) through the inversion-only path (ristretto_dc_*), the point optionally rescaled by z first;
// returns 2 if x = e g f h was zero (no output), else the decode flag
int t_double_compress(const uint8_t* pe, const uint8_t* z, uint8_t* out) {
  ge_p3 p; const int ok = dec(p, pe);
  fe zz; load(zz, z);
  fe_mul(p.X, p.X, zz); fe_mul(p.Y, p.Y, zz); fe_mul(p.Z, p.Z, zz); fe_mul(p.T, p.T, zz);
  ristretto_dc_state s; fe x, inv;
  ristretto_dc_prepare(s, x, p);
  if (fe_iszero(x)) return 2;
  fe_invert(inv, x);
  uint32_t w[8];
  ristretto_dc_finish(w, s, inv);
  memcpy(out, w, 32);
  return ok;
}
// plain double-and-add scalar multiplication with the device formulas (long dependent chains)
int t_scalarmult(const uint8_t* s, const uint8_t* pe, uint8_t* out) {
  ge_p3 p, acc; int ok = dec(p, pe);
  ge_cached c; ge_to_cached(c, p);
  ge_identity(acc);
  for (int i = 255; i >= 0; --i) {
    ge_double<true>(acc, acc);
    if ((s[i >> 3] >> (i & 7)) & 1) ge_add_cached(acc, acc, c);
  }
  enc(out, acc);
  return ok;
}
// scalar arithmetic mod l (device header sc25519.h): op 0 mul, 1 add, 2 neg(a), 3 reduce(a), 4 mont(to_mont(a), b), 5 from_wide(lo=a, hi=b)
void t_sc_op(int op, const uint8_t* a, const uint8_t* b, uint8_t* out) {
  sc x, y, r;
  memcpy(x.v, a, 32); memcpy(y.v, b, 32);
  switch (op) {
    case 0: sc_mul(r, x, y); break;
    case 1: sc_add(r, x, y); break;
    case 2: sc_neg(r, x); break;
    case 3: sc_reduce(r, x); break;
    case 4: { sc t; sc_to_mont(t, x); sc_mont(r, t, y); break; }
    case 5: sc_from_wide(r, x, y); break;      // a + b * 2^256
    case 6: { r = x; const uint32_t f = sc_fold_sign(r.v); r.v[7] |= f << 31; break; }   // folded scalar, flag in bit 255
    case 7: sc_halve(r, x); break;             // a / 2 mod l
    default: sc_zero(r);
  }
  memcpy(out, r.v, 32);
}
}
