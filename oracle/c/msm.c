/* ORACLE (test infrastructure): the three multiscalar-multiplication algorithms of curve25519-dalek 2.x
 * (`backend::serial::scalar_mul::{straus, pippenger}`, not vendored) that the reference reaches at
 *   prover.rs:94          RistrettoPoint::multiscalar_mul           -> constant-time Straus, radix 16
 *   verifier.rs:97,162    vartime_ / optional_multiscalar_mul       -> Straus, width-5 NAF   (n < 190)
 *   batch_verifier.rs:219 optional_multiscalar_mul                  -> Pippenger w = 6/7/8   (n >= 190)
 * restated from their published descriptions, plus CPU versions of the product's C-ABI contracts.
 * This file is #included by oracle_all.c after group.c (it uses that file's static field helpers). */
#include <stdlib.h>
#include <string.h>
#include "oracle.h"

typedef struct { fe51 YpX, YmX, Z, T2d; } ge_pniels;   /* dalek ProjectiveNielsPoint */

static void to_pniels(ge_pniels* r, const ge_ext* p) {
  init_consts();
  fe_add(&r->YpX, &p->Y, &p->X); fe_weak_reduce(&r->YpX);
  fe_sub(&r->YmX, &p->Y, &p->X);
  r->Z = p->Z;
  fe_mul(&r->T2d, &p->T, &C_D2);
}
static void pniels_identity(ge_pniels* r) { r->YpX = FE_ONE; r->YmX = FE_ONE; r->Z = FE_ONE; r->T2d = FE_ZERO; }

/* r = p + q (sign = 0) or p - q (sign = 1): 8M */
static void add_pniels(ge_ext* r, const ge_ext* p, const ge_pniels* q, int sign) {
  fe51 a, b, c, d, e, f, g, h, t;
  const fe51* qp = sign ? &q->YmX : &q->YpX;
  const fe51* qm = sign ? &q->YpX : &q->YmX;
  fe_add(&t, &p->Y, &p->X); fe_mul(&b, &t, qp);
  fe_sub(&t, &p->Y, &p->X); fe_mul(&a, &t, qm);
  fe_mul(&c, &p->T, &q->T2d);
  fe_mul(&d, &p->Z, &q->Z); fe_add(&d, &d, &d);
  fe_sub(&e, &b, &a); fe_add(&h, &b, &a); fe_weak_reduce(&h);
  if (sign) { fe_add(&f, &d, &c); fe_weak_reduce(&f); fe_sub(&g, &d, &c); }
  else      { fe_sub(&f, &d, &c); fe_add(&g, &d, &c); fe_weak_reduce(&g); }
  fe_mul(&r->X, &e, &f); fe_mul(&r->Y, &g, &h); fe_mul(&r->Z, &f, &g); fe_mul(&r->T, &e, &h);
}
static void pniels_cmov(ge_pniels* r, const ge_pniels* q, uint64_t mask) {
  uint64_t* a = (uint64_t*)r; const uint64_t* b = (const uint64_t*)q;
  for (size_t i = 0; i < sizeof(ge_pniels) / 8; ++i) a[i] ^= mask & (a[i] ^ b[i]);
}
static void mul_by_pow_2(ge_ext* r, int k) { for (int i = 0; i < k; ++i) orc_ge_double(r, r); }

/* ---- Scalar::to_radix_16: 64 signed digits in [-8, 8) (needs s < 2^255) -------------------- */
static void to_radix_16(int8_t out[64], const uint8_t s[32]) {
  for (int i = 0; i < 32; ++i) { out[2 * i] = s[i] & 15; out[2 * i + 1] = (s[i] >> 4) & 15; }
  for (int i = 0; i < 63; ++i) {
    const int8_t carry = (int8_t)((out[i] + 8) >> 4);
    out[i] -= (int8_t)(carry << 4);
    out[i + 1] += carry;
  }
}
/* ---- Scalar::non_adjacent_form(w) ----------------------------------------------------------- */
static void non_adjacent_form(int8_t naf[256], const uint8_t s[32], int w) {
  uint64_t x[5] = {0};
  memcpy(x, s, 32);
  memset(naf, 0, 256);
  const uint64_t width = 1ULL << w, mask = width - 1;
  uint64_t carry = 0;
  int pos = 0;
  while (pos < 256) {
    const int idx = pos / 64, bit = pos % 64;
    const uint64_t buf = bit < 64 - w ? x[idx] >> bit : (x[idx] >> bit) | (x[idx + 1] << (64 - bit));
    const uint64_t window = carry + (buf & mask);
    if ((window & 1) == 0) { pos += 1; continue; }
    if (window < width / 2) { carry = 0; naf[pos] = (int8_t)window; }
    else { carry = 1; naf[pos] = (int8_t)((int64_t)window - (int64_t)width); }
    pos += w;
  }
}
/* ---- Scalar::to_radix_2w(w), w in 4..8; returns number of digits ---------------------------- */
static int to_radix_2w(int8_t* digits /*>= 65*/, const uint8_t s[32], int w) {
  uint64_t x[5] = {0};
  memcpy(x, s, 32);
  const uint64_t radix = 1ULL << w, mask = radix - 1;
  const int count = (256 + w - 1) / w;
  uint64_t carry = 0;
  for (int i = 0; i <= count; ++i) digits[i] = 0;
  for (int i = 0; i < count; ++i) {
    const int off = i * w, idx = off / 64, bit = off % 64;
    const uint64_t buf = bit < 64 - w || idx == 3 ? x[idx] >> bit : (x[idx] >> bit) | (x[idx + 1] << (64 - bit));
    const uint64_t coef = carry + (buf & mask);
    carry = (coef + radix / 2) >> w;
    digits[i] = (int8_t)((int64_t)coef - (int64_t)(carry << w));
  }
  if (w == 8) { digits[count] += (int8_t)carry; return count + 1; }
  digits[count - 1] += (int8_t)(carry << w);
  return count;
}

/* the same three algorithms on AVX-512 IFMA or AVX2 vectors (dalek's simd_backend design), selected by orc_set_simd() */
#include "simd_ifma.c"

/* ---- constant-time Straus (dalek straus.rs, MultiscalarMul) --------------------------------- */
void orc_msm_straus_ct(ge_ext* r, size_t n, const uint8_t* scalars, const ge_ext* points) {
#if ORC_HAVE_IFMA
  if (g_orc_simd == 1) { simd_straus_ct_ifma(r, n, scalars, points); return; }
#endif
#if ORC_HAVE_AVX2
  if (g_orc_simd == 2) { simd_straus_ct_avx2(r, n, scalars, points); return; }
  if (g_orc_simd == 3) { simd_straus_ct_avx2p(r, n, scalars, points); return; }
#endif
  ge_pniels* tables = (ge_pniels*)malloc(sizeof(ge_pniels) * 8 * (n ? n : 1));
  int8_t* digits = (int8_t*)malloc(64 * (n ? n : 1));
  for (size_t i = 0; i < n; ++i) {            /* LookupTable [1P .. 8P] */
    ge_ext m = points[i];
    to_pniels(&tables[8 * i], &m);
    for (int j = 1; j < 8; ++j) { add_pniels(&m, &points[i], &tables[8 * i + j - 1], 0); to_pniels(&tables[8 * i + j], &m); }
    uint8_t red[32];
    orc_sc_reduce32(red, scalars + 32 * i);   /* dalek Scalars are always reduced */
    to_radix_16(digits + 64 * i, red);
  }
  ge_ext q;
  orc_ge_identity(&q);
  for (int j = 63; j >= 0; --j) {
    mul_by_pow_2(&q, 4);
    for (size_t i = 0; i < n; ++i) {
      const int8_t d = digits[64 * i + j];
      const int sign = d < 0;
      const int mag = sign ? -d : d;
      ge_pniels sel;
      pniels_identity(&sel);
      for (int k = 1; k <= 8; ++k) pniels_cmov(&sel, &tables[8 * i + k - 1], (uint64_t)0 - (uint64_t)(mag == k));   /* masked scan */
      add_pniels(&q, &q, &sel, sign);
    }
  }
  *r = q;
  free(tables); free(digits);
}

/* ---- vartime Straus, NAF-5 (dalek straus.rs, VartimeMultiscalarMul) ------------------------- */
void orc_msm_straus_vartime(ge_ext* r, size_t n, const uint8_t* scalars, const ge_ext* points) {
#if ORC_HAVE_IFMA
  if (g_orc_simd == 1) { simd_straus_vartime_ifma(r, n, scalars, points); return; }
#endif
#if ORC_HAVE_AVX2
  if (g_orc_simd == 2) { simd_straus_vartime_avx2(r, n, scalars, points); return; }
  if (g_orc_simd == 3) { simd_straus_vartime_avx2p(r, n, scalars, points); return; }
#endif
  ge_pniels* tables = (ge_pniels*)malloc(sizeof(ge_pniels) * 8 * (n ? n : 1));
  int8_t* nafs = (int8_t*)malloc(256 * (n ? n : 1));
  for (size_t i = 0; i < n; ++i) {            /* NafLookupTable5 [P, 3P, .., 15P] */
    ge_ext p2, m = points[i];
    orc_ge_double(&p2, &points[i]);
    ge_pniels p2n;
    to_pniels(&p2n, &p2);
    to_pniels(&tables[8 * i], &m);
    for (int j = 1; j < 8; ++j) { add_pniels(&m, &m, &p2n, 0); to_pniels(&tables[8 * i + j], &m); }
    uint8_t red[32];
    orc_sc_reduce32(red, scalars + 32 * i);
    non_adjacent_form(nafs + 256 * i, red, 5);
  }
  ge_ext q;
  orc_ge_identity(&q);
  for (int i = 255; i >= 0; --i) {
    orc_ge_double(&q, &q);
    for (size_t k = 0; k < n; ++k) {
      const int8_t d = nafs[256 * k + i];
      if (d > 0) add_pniels(&q, &q, &tables[8 * k + d / 2], 0);
      else if (d < 0) add_pniels(&q, &q, &tables[8 * k + (-d) / 2], 1);
    }
  }
  *r = q;
  free(tables); free(nafs);
}

/* ---- vartime Pippenger (dalek pippenger.rs) -------------------------------------------------- */
void orc_msm_pippenger(ge_ext* r, size_t n, const uint8_t* scalars, const ge_ext* points) {
#if ORC_HAVE_IFMA
  if (g_orc_simd == 1) { simd_pippenger_ifma(r, n, scalars, points); return; }
#endif
#if ORC_HAVE_AVX2
  if (g_orc_simd == 2) { simd_pippenger_avx2(r, n, scalars, points); return; }
  if (g_orc_simd == 3) { simd_pippenger_avx2p(r, n, scalars, points); return; }
#endif
  const int w = n < 500 ? 6 : (n < 800 ? 7 : 8);
  const int buckets_count = (1 << w) / 2;
  int8_t* digits = (int8_t*)malloc(66 * (n ? n : 1));
  ge_pniels* pn = (ge_pniels*)malloc(sizeof(ge_pniels) * (n ? n : 1));
  ge_ext* buckets = (ge_ext*)malloc(sizeof(ge_ext) * buckets_count);
  int digits_count = (256 + w - 1) / w + (w == 8 ? 1 : 0);
  for (size_t i = 0; i < n; ++i) {
    uint8_t red[32];
    orc_sc_reduce32(red, scalars + 32 * i);
    to_radix_2w(digits + 66 * i, red, w);
    to_pniels(&pn[i], &points[i]);
  }
  ge_ext total;
  orc_ge_identity(&total);
  for (int di = digits_count - 1; di >= 0; --di) {
    for (int b = 0; b < buckets_count; ++b) orc_ge_identity(&buckets[b]);
    for (size_t i = 0; i < n; ++i) {
      const int d = digits[66 * i + di];
      if (d > 0) add_pniels(&buckets[d - 1], &buckets[d - 1], &pn[i], 0);
      else if (d < 0) add_pniels(&buckets[-d - 1], &buckets[-d - 1], &pn[i], 1);
    }
    ge_ext inter = buckets[buckets_count - 1], sum = buckets[buckets_count - 1];
    for (int b = buckets_count - 2; b >= 0; --b) { orc_ge_add(&inter, &inter, &buckets[b]); orc_ge_add(&sum, &sum, &inter); }
    if (di != digits_count - 1) mul_by_pow_2(&total, w);
    orc_ge_add(&total, &total, &sum);
  }
  *r = total;
  free(digits); free(pn); free(buckets);
}

void orc_msm_vartime(ge_ext* r, size_t n, const uint8_t* scalars, const ge_ext* points) {
  if (n < 190) orc_msm_straus_vartime(r, n, scalars, points);
  else orc_msm_pippenger(r, n, scalars, points);
}

/* ---- CPU versions of the product's C-ABI contracts (include/zkp_mi355x.h) ------------------- */
int orc_msm_many(uint32_t n_msm, const uint32_t* off, const uint8_t* scalars, const uint32_t* pidx,
                 const uint8_t* points, uint32_t n_points, int flags, uint8_t* out, uint8_t* status) {
  ge_ext* dec = (ge_ext*)malloc(sizeof(ge_ext) * (n_points ? n_points : 1));
  uint8_t* ok = (uint8_t*)malloc(n_points ? n_points : 1);
  for (uint32_t i = 0; i < n_points; ++i) ok[i] = (uint8_t)orc_ristretto_decode(&dec[i], points + 32 * (size_t)i);
  size_t maxk = 1;
  for (uint32_t m = 0; m < n_msm; ++m) if (off[m + 1] - off[m] > maxk) maxk = off[m + 1] - off[m];
  ge_ext* pts = (ge_ext*)malloc(sizeof(ge_ext) * maxk);
  for (uint32_t m = 0; m < n_msm; ++m) {
    const uint32_t b = off[m], k = off[m + 1] - off[m];
    int bad = 0;
    for (uint32_t t = 0; t < k; ++t) {
      const uint32_t pi = pidx[b + t];
      if (pi >= n_points || !ok[pi]) { bad = 1; orc_ge_identity(&pts[t]); } else pts[t] = dec[pi];
    }
    ge_ext r;
    if (flags == 1) orc_msm_straus_ct(&r, k, scalars + 32 * (size_t)b, pts);
    else orc_msm_vartime(&r, k, scalars + 32 * (size_t)b, pts);
    if (bad) memset(out + 32 * (size_t)m, 0, 32); else orc_ristretto_encode(out + 32 * (size_t)m, &r);
    status[m] = (uint8_t)bad;
  }
  free(dec); free(ok); free(pts);
  return 0;
}

int orc_msm_optional(uint64_t n, const uint8_t* scalars, const uint8_t* points, uint8_t out_point[32], int* status) {
  ge_ext* pts = (ge_ext*)malloc(sizeof(ge_ext) * (n ? n : 1));
  int bad = 0;
  for (uint64_t i = 0; i < n; ++i) if (!orc_ristretto_decode(&pts[i], points + 32 * i)) bad = 1;   /* decompress() -> None */
  memset(out_point, 0, 32);
  if (!bad) { ge_ext r; orc_msm_vartime(&r, n, scalars, pts); orc_ristretto_encode(out_point, &r); }
  *status = bad;
  free(pts);
  return 0;
}
