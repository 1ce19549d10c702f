/* ORACLE (test infrastructure): GF(2^255-19) with 5 x 51-bit limbs and unsigned __int128 products --
 * the representation of curve25519-dalek 2.x `backend::serial::u64::field::FieldElement51` (not
 * vendored; reached from the reference at src/toolbox/verifier.rs:90, mod.rs:180,204) -- plus
 * extended twisted-Edwards arithmetic and the RFC 9496 ristretto255 codec. */
#include <string.h>
#include "oracle.h"

typedef unsigned __int128 u128;
#define M51 ((1ULL << 51) - 1)

static const fe51 FE_ZERO = {{0, 0, 0, 0, 0}};
static const fe51 FE_ONE = {{1, 0, 0, 0, 0}};
/* d = -121665/121666, 2d, sqrt(-1), 1/sqrt(a-d), and the Elligator constants of RFC 9496 section 4.1,
 * as 51-bit limbs (derived with python from the decimal values in the RFC; checked in tests) */
static fe51 C_D, C_D2, C_SQRT_M1, C_INVSQRT_A_MINUS_D, C_SQRT_AD_MINUS_ONE, C_ONE_MINUS_D_SQ, C_D_MINUS_ONE_SQ;
static int consts_ready = 0;

static void fe_frombytes(fe51* r, const uint8_t s[32]) {
  uint64_t w[4];
  memcpy(w, s, 32);
  r->v[0] = w[0] & M51;
  r->v[1] = ((w[0] >> 51) | (w[1] << 13)) & M51;
  r->v[2] = ((w[1] >> 38) | (w[2] << 26)) & M51;
  r->v[3] = ((w[2] >> 25) | (w[3] << 39)) & M51;
  r->v[4] = (w[3] >> 12) & M51;          /* bit 255 ignored, as dalek does */
}

static void fe_weak_reduce(fe51* r) {
  uint64_t c;
  c = r->v[0] >> 51; r->v[0] &= M51; r->v[1] += c;
  c = r->v[1] >> 51; r->v[1] &= M51; r->v[2] += c;
  c = r->v[2] >> 51; r->v[2] &= M51; r->v[3] += c;
  c = r->v[3] >> 51; r->v[3] &= M51; r->v[4] += c;
  c = r->v[4] >> 51; r->v[4] &= M51; r->v[0] += 19 * c;
}

static void fe_tobytes(uint8_t s[32], const fe51* a) {
  fe51 t = *a;
  fe_weak_reduce(&t);
  fe_weak_reduce(&t);
  /* q = 1 iff t >= p */
  uint64_t q = (t.v[0] + 19) >> 51;
  q = (t.v[1] + q) >> 51;
  q = (t.v[2] + q) >> 51;
  q = (t.v[3] + q) >> 51;
  q = (t.v[4] + q) >> 51;
  t.v[0] += 19 * q;
  uint64_t c;
  c = t.v[0] >> 51; t.v[0] &= M51; t.v[1] += c;
  c = t.v[1] >> 51; t.v[1] &= M51; t.v[2] += c;
  c = t.v[2] >> 51; t.v[2] &= M51; t.v[3] += c;
  c = t.v[3] >> 51; t.v[3] &= M51; t.v[4] += c;
  t.v[4] &= M51;
  uint64_t w[4];
  w[0] = t.v[0] | (t.v[1] << 51);
  w[1] = (t.v[1] >> 13) | (t.v[2] << 38);
  w[2] = (t.v[2] >> 26) | (t.v[3] << 25);
  w[3] = (t.v[3] >> 39) | (t.v[4] << 12);
  memcpy(s, w, 32);
}

static void fe_add(fe51* r, const fe51* a, const fe51* b) {
  for (int i = 0; i < 5; ++i) r->v[i] = a->v[i] + b->v[i];
}
/* a - b with 16p added first (dalek's sub); result weakly reduced */
static void fe_sub(fe51* r, const fe51* a, const fe51* b) {
  r->v[0] = a->v[0] + 36028797018963664ULL - b->v[0];
  r->v[1] = a->v[1] + 36028797018963952ULL - b->v[1];
  r->v[2] = a->v[2] + 36028797018963952ULL - b->v[2];
  r->v[3] = a->v[3] + 36028797018963952ULL - b->v[3];
  r->v[4] = a->v[4] + 36028797018963952ULL - b->v[4];
  fe_weak_reduce(r);
}
static void fe_neg(fe51* r, const fe51* a) { fe_sub(r, &FE_ZERO, a); }

static void fe_mul(fe51* r, const fe51* a, const fe51* b) {
  const uint64_t a0 = a->v[0], a1 = a->v[1], a2 = a->v[2], a3 = a->v[3], a4 = a->v[4];
  const uint64_t b0 = b->v[0], b1 = b->v[1], b2 = b->v[2], b3 = b->v[3], b4 = b->v[4];
  const uint64_t b1_19 = b1 * 19, b2_19 = b2 * 19, b3_19 = b3 * 19, b4_19 = b4 * 19;
  u128 c0 = (u128)a0 * b0 + (u128)a4 * b1_19 + (u128)a3 * b2_19 + (u128)a2 * b3_19 + (u128)a1 * b4_19;
  u128 c1 = (u128)a1 * b0 + (u128)a0 * b1 + (u128)a4 * b2_19 + (u128)a3 * b3_19 + (u128)a2 * b4_19;
  u128 c2 = (u128)a2 * b0 + (u128)a1 * b1 + (u128)a0 * b2 + (u128)a4 * b3_19 + (u128)a3 * b4_19;
  u128 c3 = (u128)a3 * b0 + (u128)a2 * b1 + (u128)a1 * b2 + (u128)a0 * b3 + (u128)a4 * b4_19;
  u128 c4 = (u128)a4 * b0 + (u128)a3 * b1 + (u128)a2 * b2 + (u128)a1 * b3 + (u128)a0 * b4;
  c1 += (uint64_t)(c0 >> 51); r->v[0] = (uint64_t)c0 & M51;
  c2 += (uint64_t)(c1 >> 51); r->v[1] = (uint64_t)c1 & M51;
  c3 += (uint64_t)(c2 >> 51); r->v[2] = (uint64_t)c2 & M51;
  c4 += (uint64_t)(c3 >> 51); r->v[3] = (uint64_t)c3 & M51;
  const uint64_t carry = (uint64_t)(c4 >> 51);
  r->v[4] = (uint64_t)c4 & M51;
  r->v[0] += carry * 19;
  r->v[1] += r->v[0] >> 51;
  r->v[0] &= M51;
}
static void fe_sq(fe51* r, const fe51* a) { fe_mul(r, a, a); }
static void fe_sqn(fe51* r, const fe51* a, int n) {
  fe_sq(r, a);
  for (int i = 1; i < n; ++i) fe_sq(r, r);
}

static void fe_pow22523(fe51* out, const fe51* z) {
  fe51 t0, t1, t2;
  fe_sq(&t0, z);
  fe_sqn(&t1, &t0, 2);
  fe_mul(&t1, z, &t1);
  fe_mul(&t0, &t0, &t1);
  fe_sq(&t0, &t0);
  fe_mul(&t0, &t1, &t0);
  fe_sqn(&t1, &t0, 5);   fe_mul(&t0, &t1, &t0);
  fe_sqn(&t1, &t0, 10);  fe_mul(&t1, &t1, &t0);
  fe_sqn(&t2, &t1, 20);  fe_mul(&t1, &t2, &t1);
  fe_sqn(&t1, &t1, 10);  fe_mul(&t0, &t1, &t0);
  fe_sqn(&t1, &t0, 50);  fe_mul(&t1, &t1, &t0);
  fe_sqn(&t2, &t1, 100); fe_mul(&t1, &t2, &t1);
  fe_sqn(&t1, &t1, 50);  fe_mul(&t0, &t1, &t0);
  fe_sqn(&t0, &t0, 2);
  fe_mul(out, &t0, z);
}

static int fe_is_negative(const fe51* a) { uint8_t s[32]; fe_tobytes(s, a); return s[0] & 1; }
static int fe_is_zero(const fe51* a) {
  uint8_t s[32]; fe_tobytes(s, a);
  uint8_t x = 0; for (int i = 0; i < 32; ++i) x |= s[i];
  return x == 0;
}
static int fe_eq(const fe51* a, const fe51* b) {
  uint8_t s[32], t[32]; fe_tobytes(s, a); fe_tobytes(t, b);
  return memcmp(s, t, 32) == 0;
}
static void fe_cneg(fe51* r, int flag) { if (flag) { fe51 n; fe_neg(&n, r); *r = n; } }
static void fe_abs(fe51* r) { fe_cneg(r, fe_is_negative(r)); }

/* FieldElement::sqrt_ratio_i (RFC 9496 SQRT_RATIO_M1) */
static int fe_sqrt_ratio_i(fe51* r, const fe51* u, const fe51* v) {
  fe51 v3, v7, t, check, neg_u, neg_u_i;
  fe_sq(&t, v); fe_mul(&v3, &t, v);
  fe_sq(&t, &v3); fe_mul(&v7, &t, v);
  fe_mul(&t, u, &v7);
  fe_pow22523(&t, &t);
  fe_mul(&t, &t, &v3);
  fe_mul(r, &t, u);
  fe_sq(&t, r); fe_mul(&check, &t, v);
  fe_neg(&neg_u, u);
  fe_mul(&neg_u_i, &neg_u, &C_SQRT_M1);
  const int correct = fe_eq(&check, u), flipped = fe_eq(&check, &neg_u), flipped_i = fe_eq(&check, &neg_u_i);
  if (flipped || flipped_i) { fe_mul(&t, r, &C_SQRT_M1); *r = t; }
  fe_abs(r);
  return correct | flipped;
}

static void fe_from_dec_limbs(fe51* r, const uint8_t le32[32]) { fe_frombytes(r, le32); }

static void init_consts(void) {
  if (consts_ready) return;
  /* little-endian bytes of the constants (hex of the RFC 9496 decimals) */
  static const uint8_t D[32] = {0xa3,0x78,0x59,0x13,0xca,0x4d,0xeb,0x75,0xab,0xd8,0x41,0x41,0x4d,0x0a,0x70,0x00,0x98,0xe8,0x79,0x77,0x79,0x40,0xc7,0x8c,0x73,0xfe,0x6f,0x2b,0xee,0x6c,0x03,0x52};
  static const uint8_t SQRT_M1[32] = {0xb0,0xa0,0x0e,0x4a,0x27,0x1b,0xee,0xc4,0x78,0xe4,0x2f,0xad,0x06,0x18,0x43,0x2f,0xa7,0xd7,0xfb,0x3d,0x99,0x00,0x4d,0x2b,0x0b,0xdf,0xc1,0x4f,0x80,0x24,0x83,0x2b};
  fe_from_dec_limbs(&C_D, D);
  fe_add(&C_D2, &C_D, &C_D); fe_weak_reduce(&C_D2);
  fe_from_dec_limbs(&C_SQRT_M1, SQRT_M1);
  consts_ready = 1;   /* sqrt_ratio below needs SQRT_M1 */
  fe51 t, one = FE_ONE, m1;
  fe_neg(&m1, &one);
  fe_sub(&t, &m1, &C_D);                         /* a - d = -1 - d */
  fe_sqrt_ratio_i(&C_INVSQRT_A_MINUS_D, &one, &t);
  fe_sub(&t, &m1, &C_D);                         /* a d - 1 = -d - 1 */
  fe_sqrt_ratio_i(&C_SQRT_AD_MINUS_ONE, &t, &one);
  fe_neg(&C_SQRT_AD_MINUS_ONE, &C_SQRT_AD_MINUS_ONE);   /* RFC lists the odd root */
  fe_sq(&t, &C_D); fe_sub(&C_ONE_MINUS_D_SQ, &one, &t);
  fe_sub(&t, &C_D, &one); fe_sq(&C_D_MINUS_ONE_SQ, &t);
}

/* ---- extended points ------------------------------------------------------------------------ */
void orc_ge_identity(ge_ext* r) { r->X = FE_ZERO; r->Y = FE_ONE; r->Z = FE_ONE; r->T = FE_ZERO; }

void orc_ge_add(ge_ext* r, const ge_ext* p, const ge_ext* q) {
  init_consts();
  fe51 a, b, c, d, e, f, g, h, t, u;
  fe_sub(&t, &p->Y, &p->X); fe_sub(&u, &q->Y, &q->X); fe_mul(&a, &t, &u);
  fe_add(&t, &p->Y, &p->X); fe_add(&u, &q->Y, &q->X); fe_mul(&b, &t, &u);
  fe_mul(&t, &p->T, &q->T); fe_mul(&c, &t, &C_D2);
  fe_mul(&t, &p->Z, &q->Z); fe_add(&d, &t, &t);
  fe_sub(&e, &b, &a); fe_sub(&f, &d, &c); fe_add(&g, &d, &c); fe_add(&h, &b, &a);
  fe_weak_reduce(&g); fe_weak_reduce(&h);
  fe_mul(&r->X, &e, &f); fe_mul(&r->Y, &g, &h); fe_mul(&r->Z, &f, &g); fe_mul(&r->T, &e, &h);
}
static void ge_neg(ge_ext* r, const ge_ext* p) { fe_neg(&r->X, &p->X); r->Y = p->Y; r->Z = p->Z; fe_neg(&r->T, &p->T); }
static void ge_sub(ge_ext* r, const ge_ext* p, const ge_ext* q) { ge_ext n; ge_neg(&n, q); orc_ge_add(r, p, &n); }

void orc_ge_double(ge_ext* r, const ge_ext* p) {
  fe51 xx, yy, zz2, xpy, e, g, f, h, t;
  fe_sq(&xx, &p->X); fe_sq(&yy, &p->Y);
  fe_sq(&t, &p->Z); fe_add(&zz2, &t, &t);
  fe_add(&t, &p->X, &p->Y); fe_sq(&xpy, &t);
  fe_add(&h, &yy, &xx); fe_weak_reduce(&h);
  fe_sub(&g, &yy, &xx);
  fe_sub(&e, &xpy, &h);
  fe_sub(&f, &zz2, &g);
  fe_mul(&r->X, &e, &f); fe_mul(&r->Y, &h, &g); fe_mul(&r->Z, &g, &f); fe_mul(&r->T, &e, &h);
}

/* ---- ristretto255 (RFC 9496 section 4.3) ---------------------------------------------------- */
int orc_ristretto_decode(ge_ext* r, const uint8_t enc[32]) {
  init_consts();
  fe51 s, ss, u1, u2, u2s, v, t, I, dx, dy, one = FE_ONE;
  uint8_t chk[32];
  fe_frombytes(&s, enc);
  fe_tobytes(chk, &s);
  const int canonical = memcmp(chk, enc, 32) == 0;      /* dalek: s_bytes_check == as_bytes */
  const int s_neg = enc[0] & 1;
  fe_sq(&ss, &s);
  fe_sub(&u1, &one, &ss);
  fe_add(&u2, &one, &ss); fe_weak_reduce(&u2);
  fe_sq(&u2s, &u2);
  fe_sq(&t, &u1); fe_mul(&v, &t, &C_D); fe_neg(&v, &v); fe_sub(&v, &v, &u2s);
  fe_mul(&t, &v, &u2s);
  const int ok = fe_sqrt_ratio_i(&I, &one, &t);
  fe_mul(&dx, &I, &u2);
  fe_mul(&dy, &I, &dx); fe_mul(&dy, &dy, &v);
  fe_add(&t, &s, &s); fe_weak_reduce(&t); fe_mul(&r->X, &t, &dx); fe_abs(&r->X);
  fe_mul(&r->Y, &u1, &dy);
  r->Z = FE_ONE;
  fe_mul(&r->T, &r->X, &r->Y);
  if (!canonical || s_neg || !ok || fe_is_negative(&r->T) || fe_is_zero(&r->Y)) { orc_ge_identity(r); return 0; }
  return 1;
}

void orc_ristretto_encode(uint8_t enc[32], const ge_ext* p) {
  init_consts();
  fe51 u1, u2, t, t2, I, den1, den2, zinv, ix, iy, ench, x, y, dinv, one = FE_ONE;
  fe_add(&t, &p->Z, &p->Y); fe_weak_reduce(&t); fe_sub(&t2, &p->Z, &p->Y); fe_mul(&u1, &t, &t2);
  fe_mul(&u2, &p->X, &p->Y);
  fe_sq(&t, &u2); fe_mul(&t, &t, &u1);
  fe_sqrt_ratio_i(&I, &one, &t);
  fe_mul(&den1, &I, &u1); fe_mul(&den2, &I, &u2);
  fe_mul(&t, &den1, &den2); fe_mul(&zinv, &t, &p->T);
  fe_mul(&ix, &p->X, &C_SQRT_M1); fe_mul(&iy, &p->Y, &C_SQRT_M1);
  fe_mul(&ench, &den1, &C_INVSQRT_A_MINUS_D);
  fe_mul(&t, &p->T, &zinv);
  if (fe_is_negative(&t)) { x = iy; y = ix; dinv = ench; } else { x = p->X; y = p->Y; dinv = den2; }
  fe_mul(&t, &x, &zinv);
  fe_cneg(&y, fe_is_negative(&t));
  fe_sub(&t, &p->Z, &y); fe_mul(&t, &dinv, &t); fe_abs(&t);
  fe_tobytes(enc, &t);
}

static void elligator(ge_ext* r, const fe51* r0) {
  fe51 rr, u, v, s, sp, c, n, w0, w1, w2, w3, t, one = FE_ONE, m1;
  fe_neg(&m1, &one);
  fe_sq(&t, r0); fe_mul(&rr, &t, &C_SQRT_M1);
  fe_add(&t, &rr, &one); fe_weak_reduce(&t); fe_mul(&u, &t, &C_ONE_MINUS_D_SQ);
  fe_mul(&t, &rr, &C_D); fe_sub(&t, &m1, &t);
  fe_add(&v, &rr, &C_D); fe_weak_reduce(&v); fe_mul(&v, &t, &v);
  const int was_square = fe_sqrt_ratio_i(&s, &u, &v);
  fe_mul(&sp, &s, r0); fe_abs(&sp); fe_neg(&sp, &sp);
  if (!was_square) s = sp;
  c = was_square ? m1 : rr;
  fe_sub(&t, &rr, &one); fe_mul(&n, &c, &t); fe_mul(&n, &n, &C_D_MINUS_ONE_SQ); fe_sub(&n, &n, &v);
  fe_mul(&t, &s, &v); fe_add(&w0, &t, &t); fe_weak_reduce(&w0);
  fe_mul(&w1, &n, &C_SQRT_AD_MINUS_ONE);
  fe_sq(&t, &s); fe_sub(&w2, &one, &t); fe_add(&w3, &one, &t); fe_weak_reduce(&w3);
  fe_mul(&r->X, &w0, &w3); fe_mul(&r->Y, &w2, &w1); fe_mul(&r->Z, &w1, &w3); fe_mul(&r->T, &w0, &w2);
}

void orc_ristretto_from_uniform_bytes(ge_ext* r, const uint8_t b[64]) {
  init_consts();
  fe51 r1, r2; ge_ext p1, p2;
  fe_frombytes(&r1, b); fe_frombytes(&r2, b + 32);
  elligator(&p1, &r1); elligator(&p2, &r2);
  orc_ge_add(r, &p1, &p2);
}

void orc_ge_basepoint(ge_ext* r) {
  static const uint8_t B[32] = {0xe2,0xf2,0xae,0x0a,0x6a,0xbc,0x4e,0x71,0xa8,0x84,0xa9,0x61,0xc5,0x00,0x51,0x5f,0x58,0xe3,0x0b,0x6a,0xa5,0x82,0xdd,0x8d,0xb6,0xa6,0x59,0x45,0xe0,0x8d,0x2d,0x76};
  orc_ristretto_decode(r, B);
}

int orc_decode_check(uint64_t n, const uint8_t* points, uint8_t* status, uint8_t* xyzt) {
  for (uint64_t i = 0; i < n; ++i) {
    ge_ext p;
    status[i] = orc_ristretto_decode(&p, points + 32 * i) ? 0 : 1;
    if (xyzt) { fe_tobytes(xyzt + 128 * i, &p.X); fe_tobytes(xyzt + 128 * i + 32, &p.Y); fe_tobytes(xyzt + 128 * i + 64, &p.Z); fe_tobytes(xyzt + 128 * i + 96, &p.T); }
  }
  return 0;
}
int orc_encode_many(uint64_t n, const uint8_t* xyzt, uint8_t* out) {
  for (uint64_t i = 0; i < n; ++i) {
    ge_ext p;
    fe_frombytes(&p.X, xyzt + 128 * i); fe_frombytes(&p.Y, xyzt + 128 * i + 32);
    fe_frombytes(&p.Z, xyzt + 128 * i + 64); fe_frombytes(&p.T, xyzt + 128 * i + 96);
    orc_ristretto_encode(out + 32 * i, &p);
  }
  return 0;
}

/* helpers shared with msm.c */
void orc__ge_neg(ge_ext* r, const ge_ext* p) { ge_neg(r, p); }
void orc__ge_sub(ge_ext* r, const ge_ext* p, const ge_ext* q) { ge_sub(r, p, q); }
