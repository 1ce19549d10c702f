/* ORACLE (test infrastructure): arithmetic modulo l = 2^252 + 27742317777372353535851937790883648493,
 * the curve25519-dalek `Scalar` operations the toolbox uses (reference: mod.rs:226
 * from_bytes_mod_order_wide, prover.rs:108 s*c+b, verifier.rs:95,142 negation,
 * verifier.rs:155-158 / batch_verifier.rs:183-201 random-weight products).
 * Reduction folds at bit 252 with 2^252 == -c (mod l) three times; plain little-endian 64-bit limbs. */
#include <string.h>
#include "oracle.h"

typedef unsigned __int128 u128;

static const uint64_t L_LIMBS[4] = {0x5812631a5cf5d3edULL, 0x14def9dea2f79cd6ULL, 0, 0x1000000000000000ULL};
static const uint64_t C_LIMBS[2] = {0x5812631a5cf5d3edULL, 0x14def9dea2f79cd6ULL};   /* c = l - 2^252 */

/* out[na+nb] = a * b */
static void bn_mul(uint64_t* out, const uint64_t* a, int na, const uint64_t* b, int nb) {
  for (int i = 0; i < na + nb; ++i) out[i] = 0;
  for (int i = 0; i < na; ++i) {
    uint64_t carry = 0;
    for (int j = 0; j < nb; ++j) {
      const u128 t = (u128)a[i] * b[j] + out[i + j] + carry;
      out[i + j] = (uint64_t)t;
      carry = (uint64_t)(t >> 64);
    }
    out[i + nb] = carry;
  }
}
/* a += b (n limbs); returns carry */
static uint64_t bn_add(uint64_t* a, const uint64_t* b, int n) {
  uint64_t c = 0;
  for (int i = 0; i < n; ++i) { const u128 t = (u128)a[i] + b[i] + c; a[i] = (uint64_t)t; c = (uint64_t)(t >> 64); }
  return c;
}
/* a -= b (n limbs); returns borrow */
static uint64_t bn_sub(uint64_t* a, const uint64_t* b, int n) {
  uint64_t br = 0;
  for (int i = 0; i < n; ++i) {
    const u128 t = (u128)a[i] - b[i] - br;
    a[i] = (uint64_t)t;
    br = (uint64_t)(t >> 64) & 1;
  }
  return br;
}
static int bn_ge(const uint64_t* a, const uint64_t* b, int n) {
  for (int i = n - 1; i >= 0; --i) { if (a[i] > b[i]) return 1; if (a[i] < b[i]) return 0; }
  return 1;
}
/* split x (n limbs) at bit 252: lo[4] = x mod 2^252, hi[n-3] = x >> 252 */
static void split252(uint64_t lo[4], uint64_t* hi, const uint64_t* x, int n) {
  lo[0] = x[0]; lo[1] = x[1]; lo[2] = x[2]; lo[3] = x[3] & 0x0fffffffffffffffULL;
  for (int i = 0; i < n - 3; ++i) {
    const uint64_t a = x[i + 3] >> 60;
    const uint64_t b = (i + 4 < n) ? (x[i + 4] << 4) : 0;
    hi[i] = a | b;
  }
}

/* out[4] = x[8] mod l */
static void sc_reduce512(uint64_t out[4], const uint64_t x[8]) {
  uint64_t r1[4], q1[5], y[7], r2[4], q2[4], z[6], r3[4], q3[3], w[5];
  split252(r1, q1, x, 8);                 /* x = q1 2^252 + r1,  q1 < 2^260                    */
  bn_mul(y, q1, 5, C_LIMBS, 2);           /* y = q1 c < 2^385                                   */
  split252(r2, q2, y, 7);                 /* q2 < 2^133                                          */
  bn_mul(z, q2, 4, C_LIMBS, 2);           /* z = q2 c < 2^258                                    */
  split252(r3, q3, z, 6);                 /* q3 < 2^6                                            */
  bn_mul(w, q3, 3, C_LIMBS, 2);           /* w = q3 c < 2^131                                    */
  /* x == r1 - r2 + r3 - w  (mod l);  S = r1 + r3 + 4l - r2 - w  in [0, 2^255) */
  uint64_t S[4] = {r1[0], r1[1], r1[2], r1[3]};
  bn_add(S, r3, 4);
  for (int k = 0; k < 4; ++k) bn_add(S, L_LIMBS, 4);
  bn_sub(S, r2, 4);
  bn_sub(S, w, 4);
  while (bn_ge(S, L_LIMBS, 4)) bn_sub(S, L_LIMBS, 4);
  memcpy(out, S, 32);
}

void orc_sc_from_wide(uint8_t out[32], const uint8_t in[64]) {
  uint64_t x[8], r[4];
  memcpy(x, in, 64);
  sc_reduce512(r, x);
  memcpy(out, r, 32);
}
/* Scalar::from_canonical_bytes / dalek's serde Deserialize for Scalar [RECALL: curve25519-dalek 2.x src/scalar.rs]: a 32-byte
 * string is a Scalar only if its value is < l.  The reference's proofs reach a verifier through serde (proofs.rs:14-32,
 * tests/zkp.rs:53-54), so a challenge or response >= l can never arrive there.  1 = canonical. */
int orc_sc_is_canonical(const uint8_t in[32]) {
  uint64_t x[4];
  memcpy(x, in, 32);
  return bn_ge(x, L_LIMBS, 4) ? 0 : 1;
}
void orc_sc_reduce32(uint8_t out[32], const uint8_t in[32]) {
  uint8_t wide[64] = {0};
  memcpy(wide, in, 32);
  orc_sc_from_wide(out, wide);
}
void orc_sc_muladd(uint8_t out[32], const uint8_t a[32], const uint8_t b[32], const uint8_t c[32]) {
  uint64_t x[4], y[4], z[4], p[8], cc[8] = {0};
  memcpy(x, a, 32); memcpy(y, b, 32); memcpy(z, c, 32);
  bn_mul(p, x, 4, y, 4);
  memcpy(cc, z, 32);
  bn_add(p, cc, 8);                        /* < 2^512: a, b, c < 2^256 gives at most 2^512 - 2^257 + ... fits */
  uint64_t r[4];
  sc_reduce512(r, p);
  memcpy(out, r, 32);
}
void orc_sc_add(uint8_t out[32], const uint8_t a[32], const uint8_t b[32]) {
  static const uint8_t one[32] = {1};
  orc_sc_muladd(out, a, one, b);
}
void orc_sc_neg(uint8_t out[32], const uint8_t a[32]) {
  uint8_t r[32];
  orc_sc_reduce32(r, a);
  uint64_t x[4], l[4];
  memcpy(x, r, 32);
  memcpy(l, L_LIMBS, 32);
  if ((x[0] | x[1] | x[2] | x[3]) == 0) { memset(out, 0, 32); return; }
  bn_sub(l, x, 4);
  memcpy(out, l, 32);
}
void orc_sc_sub(uint8_t out[32], const uint8_t a[32], const uint8_t b[32]) {
  uint8_t nb[32];
  orc_sc_neg(nb, b);
  orc_sc_add(out, a, nb);
}
