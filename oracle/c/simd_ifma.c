/* ORACLE (test infrastructure) -- the CPU baseline's MSM inner loops on AVX-512 IFMA, 4 x 64-bit lanes = the four coordinates of ONE point.
 *
 * north_star asks for the reference's `simd_backend` next to the GPU numbers; the Rust crate cannot be built here (Cargo.toml:37-40 selects
 * curve25519-dalek's avx2 / ifma backends by feature; no toolchain, dependency not vendored).  This file restates that backend's published
 * design so that bench.py can print a MEASURED `cpu_baseline.simd` instead of a "1.5 - 2 x [RECALL]" note:
 *   - a vector of four field elements in radix 2^51 (dalek backend/vector/ifma/field.rs `F51x4Reduced`): products by vpmadd52luq / vpmadd52huq,
 *     the high halves entering one limb up with weight 2 (2^52 = 2 * 2^51);
 *   - point addition and doubling in the parallel formulas of Hisil-Wong-Carter-Dawson 2008 section 4 (dalek backend/vector/ifma/edwards.rs):
 *     two rounds of four simultaneous multiplications with lane shuffles in between;
 *   - on top of them the same three algorithms as the scalar port (msm.c): constant-time radix-16 Straus, width-5 NAF Straus, Pippenger w = 6/7/8.
 * Decompression / compression (one inverse square root chain per point) stay scalar, as in dalek.
 * Results are KAT-equal to the scalar port (tests/test_oracle_c.py::test_simd_msm_equals_scalar_port).  Selected at run time by orc_set_simd(1);
 * compiled only when the compiler targets AVX-512 IFMA + VL (oracle/Makefile: -march=native), refused at run time on a CPU without them. */
#if defined(__AVX512IFMA__) && defined(__AVX512VL__)
#include <immintrin.h>
#define ORC_HAVE_IFMA 1

typedef struct { __m256i v[5]; } f4;                 /* lane k of every limb = field element k; limbs < 2^52 wherever a product is taken */

static inline __m256i bc(uint64_t x) { return _mm256_set1_epi64x((long long)x); }
/* lane shuffles: imm = l0 | l1 << 2 | l2 << 4 | l3 << 6 -- output lane k takes input lane l_k */
#define SH(l0, l1, l2, l3) ((l0) | ((l1) << 2) | ((l2) << 4) | ((l3) << 6))
enum { SH_1132 = SH(1, 1, 3, 2), SH_0000 = SH(0, 0, 0, 0), SH_1331 = SH(1, 3, 3, 1), SH_0220 = SH(0, 2, 2, 0), SH_3131 = SH(3, 1, 3, 1), SH_2020 = SH(2, 0, 2, 0),
       SH_0120 = SH(0, 1, 2, 0), SH_1111 = SH(1, 1, 1, 1), SH_3003 = SH(3, 0, 0, 3), SH_1032 = SH(1, 0, 2, 3) };
static inline f4 shuf(const f4* a, const int imm) {
  f4 r;
  switch (imm) {
#define C(I) case I: for (int i = 0; i < 5; ++i) r.v[i] = _mm256_permute4x64_epi64(a->v[i], I); break;
    C(SH_1132) C(SH_0000) C(SH_1331) C(SH_0220) C(SH_3131) C(SH_2020) C(SH_0120) C(SH_1111) C(SH_3003) C(SH_1032)
#undef C
    default: r = *a;
  }
  return r;
}
/* per-lane select: mask bit k (0..3) set -> lane k from b */
static inline f4 blend(const f4* a, const f4* b, const int lanes) {
  f4 r;
  const __m256i m = _mm256_set_epi64x((lanes & 8) ? -1 : 0, (lanes & 4) ? -1 : 0, (lanes & 2) ? -1 : 0, (lanes & 1) ? -1 : 0);
  for (int i = 0; i < 5; ++i) r.v[i] = _mm256_blendv_epi8(a->v[i], b->v[i], m);
  return r;
}
static inline f4 f4_addraw(const f4* a, const f4* b) { f4 r; for (int i = 0; i < 5; ++i) r.v[i] = _mm256_add_epi64(a->v[i], b->v[i]); return r; }
/* carry pass, all limbs at once (dalek's F51x4Unreduced -> F51x4Reduced): every limb gives its bits above 51 to the next one, the top limb's
 * wrap to limb 0 with weight 19.  Inputs < 2^63 -> outputs < 2^51 + 2^17 < 2^52, which is all a product needs; not a canonical form. */
static inline void f4_reduce(f4* a) {
  const __m256i M = bc(M51);
  const __m256i c0 = _mm256_srli_epi64(a->v[0], 51), c1 = _mm256_srli_epi64(a->v[1], 51), c2 = _mm256_srli_epi64(a->v[2], 51),
                c3 = _mm256_srli_epi64(a->v[3], 51), c4 = _mm256_srli_epi64(a->v[4], 51);
  const __m256i c4_19 = _mm256_add_epi64(_mm256_slli_epi64(c4, 4), _mm256_add_epi64(_mm256_slli_epi64(c4, 1), c4));
  a->v[0] = _mm256_add_epi64(_mm256_and_si256(a->v[0], M), c4_19);
  a->v[1] = _mm256_add_epi64(_mm256_and_si256(a->v[1], M), c0);
  a->v[2] = _mm256_add_epi64(_mm256_and_si256(a->v[2], M), c1);
  a->v[3] = _mm256_add_epi64(_mm256_and_si256(a->v[3], M), c2);
  a->v[4] = _mm256_add_epi64(_mm256_and_si256(a->v[4], M), c3);
}
static inline f4 f4_add(const f4* a, const f4* b) { f4 r = f4_addraw(a, b); f4_reduce(&r); return r; }
/* a - b + 16 p (the scalar port's bias), reduced */
static inline f4 f4_sub(const f4* a, const f4* b) {
  f4 r;
  r.v[0] = _mm256_sub_epi64(_mm256_add_epi64(a->v[0], bc(36028797018963664ULL)), b->v[0]);
  for (int i = 1; i < 5; ++i) r.v[i] = _mm256_sub_epi64(_mm256_add_epi64(a->v[i], bc(36028797018963952ULL)), b->v[i]);
  f4_reduce(&r);
  return r;
}
/* four products at once; inputs: every limb < 2^52 */
static inline f4 f4_mul(const f4* a, const f4* b) {
  __m256i L[9], H[9];
  const __m256i z = _mm256_setzero_si256();
  for (int k = 0; k < 9; ++k) { L[k] = z; H[k] = z; }
  for (int i = 0; i < 5; ++i)
    for (int j = 0; j < 5; ++j) {
      L[i + j] = _mm256_madd52lo_epu64(L[i + j], a->v[i], b->v[j]);
      H[i + j] = _mm256_madd52hi_epu64(H[i + j], a->v[i], b->v[j]);
    }
  /* column k (weight 2^(51 k)) = L[k] + 2 H[k - 1]; columns 5 .. 9 fold with 2^255 = 19 */
  __m256i c[10];
  c[0] = L[0];
  for (int k = 1; k < 9; ++k) c[k] = _mm256_add_epi64(L[k], _mm256_slli_epi64(H[k - 1], 1));
  c[9] = _mm256_slli_epi64(H[8], 1);
  f4 r;
  for (int k = 0; k < 5; ++k) {
    const __m256i t = c[k + 5];
    r.v[k] = _mm256_add_epi64(c[k], _mm256_add_epi64(_mm256_slli_epi64(t, 4), _mm256_add_epi64(_mm256_slli_epi64(t, 1), t)));
  }
  f4_reduce(&r);
  return r;
}

/* ---- points: extended (X, Y, Z, T) in lanes 0..3; cached = (Y - X, Y + X, 2 d T, Z) -------------------------------------------------- */
static f4 F4_CACHE_CONST;       /* (1, 1, 2 d, 1) */
static int f4_consts_ready = 0;
static inline f4 f4_pack(const fe51* a, const fe51* b, const fe51* c, const fe51* d) {
  f4 r;
  for (int i = 0; i < 5; ++i) r.v[i] = _mm256_set_epi64x((long long)d->v[i], (long long)c->v[i], (long long)b->v[i], (long long)a->v[i]);
  return r;
}
static inline void f4_unpack(fe51 out[4], const f4* a) {
  uint64_t t[4];
  for (int i = 0; i < 5; ++i) { _mm256_storeu_si256((__m256i*)t, a->v[i]); for (int k = 0; k < 4; ++k) out[k].v[i] = t[k]; }
  for (int k = 0; k < 4; ++k) fe_weak_reduce(&out[k]);
}
static void f4_init(void) {
  if (f4_consts_ready) return;
  init_consts();
  F4_CACHE_CONST = f4_pack(&FE_ONE, &FE_ONE, &C_D2, &FE_ONE);
  f4_consts_ready = 1;
}
static inline f4 x4_from_ext(const ge_ext* p) {
  ge_ext q = *p;
  fe_weak_reduce(&q.X); fe_weak_reduce(&q.Y); fe_weak_reduce(&q.Z); fe_weak_reduce(&q.T);
  return f4_pack(&q.X, &q.Y, &q.Z, &q.T);
}
static inline void x4_to_ext(ge_ext* r, const f4* p) { fe51 o[4]; f4_unpack(o, p); r->X = o[0]; r->Y = o[1]; r->Z = o[2]; r->T = o[3]; }
static inline f4 x4_identity(void) { return f4_pack(&FE_ZERO, &FE_ONE, &FE_ONE, &FE_ZERO); }
static inline f4 x4_cached_identity(void) { return f4_pack(&FE_ONE, &FE_ONE, &FE_ZERO, &FE_ONE); }
/* (X, Y, Z, T) -> (Y - X, Y + X, 2 d T, Z) */
static inline f4 x4_to_cached(const f4* p) {
  const f4 s1 = shuf(p, SH_1132);                   /* (Y, Y, T, Z) */
  const f4 s2 = shuf(p, SH_0000);                   /* (X, X, X, X) */
  const f4 d = f4_sub(&s1, &s2), a = f4_add(&s1, &s2);
  f4 t = blend(&s1, &d, 1);                         /* lane 0: Y - X */
  t = blend(&t, &a, 2);                             /* lane 1: Y + X */
  return f4_mul(&t, &F4_CACHE_CONST);
}
/* cached form of -Q: swap (Y - X, Y + X), negate 2 d T */
static inline f4 x4_cached_neg(const f4* q) {
  const f4 sw = shuf(q, SH_1032);
  const f4 zero = f4_pack(&FE_ZERO, &FE_ZERO, &FE_ZERO, &FE_ZERO);
  const f4 n = f4_sub(&zero, &sw);
  return blend(&sw, &n, 4);
}
/* P + Q, Q cached: 2 x 4 multiplications */
static inline f4 x4_add_cached(const f4* p, const f4* q) {
  const f4 s1 = shuf(p, SH_1132), s2 = shuf(p, SH_0000);
  const f4 d = f4_sub(&s1, &s2), a = f4_add(&s1, &s2);
  f4 A = blend(&s1, &d, 1);
  A = blend(&A, &a, 2);                             /* (Y - X, Y + X, T, Z) */
  const f4 M = f4_mul(&A, q);                       /* (a, b, c, d0) */
  f4 u1 = shuf(&M, SH_1331), u2 = shuf(&M, SH_0220);   /* (b, d0, d0, b), (a, c, c, a) */
  f4 v1 = shuf(&M, SH_3131), v2 = shuf(&M, SH_2020);   /* (d0, b, d0, b), (c, a, c, a) */
  const f4 u1d = f4_addraw(&u1, &u1), v1d = f4_addraw(&v1, &v1);
  u1 = blend(&u1, &u1d, 2 | 4);                     /* (b, 2 d0, 2 d0, b) */
  v1 = blend(&v1, &v1d, 1 | 4);                     /* (2 d0, b, 2 d0, b) */
  const f4 us = f4_sub(&u1, &u2), ua = f4_add(&u1, &u2);
  const f4 vs = f4_sub(&v1, &v2), va = f4_add(&v1, &v2);
  const f4 E1 = blend(&us, &ua, 2);                 /* (e, g, f, e) */
  const f4 E2 = blend(&va, &vs, 1);                 /* (f, h, g, h) */
  return f4_mul(&E1, &E2);                          /* (e f, g h, f g, e h) = (X3, Y3, Z3, T3) */
}
static inline f4 x4_double(const f4* p) {
  f4 A = shuf(p, SH_0120);                          /* (X, Y, Z, X) */
  const f4 y = shuf(p, SH_1111);
  const f4 s = f4_add(&A, &y);
  A = blend(&A, &s, 8);                             /* (X, Y, Z, X + Y) */
  const f4 S = f4_mul(&A, &A);                      /* (xx, yy, zz, (x + y)^2) */
  const f4 yy = shuf(&S, SH_1111), xx = shuf(&S, SH_0000);
  const f4 h = f4_add(&yy, &xx), g = f4_sub(&yy, &xx);
  const f4 Sd = f4_addraw(&S, &S);
  const f4 S2 = blend(&S, &Sd, 4);                  /* lane 2: 2 zz */
  const f4 gh = blend(&g, &h, 8);                   /* (g, g, g, h) */
  const f4 W = f4_sub(&S2, &gh);                    /* lane 2: f = 2 zz - g, lane 3: e = (x + y)^2 - h */
  f4 E1 = shuf(&W, SH_3003);                        /* (e, ., ., e) */
  E1 = blend(&E1, &h, 2);
  E1 = blend(&E1, &g, 4);                           /* (e, h, g, e) */
  f4 E2 = shuf(&W, SH_2020);                        /* (f, ., f, .) */
  E2 = blend(&E2, &g, 2);
  E2 = blend(&E2, &h, 8);                           /* (f, g, f, h) */
  return f4_mul(&E1, &E2);                          /* (e f, h g, g f, e h) */
}
static inline f4 x4_cmov(const f4* a, const f4* b, uint64_t mask) {
  f4 r;
  const __m256i m = bc(mask);
  for (int i = 0; i < 5; ++i) r.v[i] = _mm256_blendv_epi8(a->v[i], b->v[i], m);
  return r;
}

/* ---- the three algorithms of msm.c on the vector backend ------------------------------------------------------------------------------ */
static void simd_straus_ct(ge_ext* r, size_t n, const uint8_t* scalars, const ge_ext* points) {
  f4_init();
  f4* tables = (f4*)aligned_alloc(32, sizeof(f4) * 8 * (n ? n : 1));
  int8_t* digits = (int8_t*)malloc(64 * (n ? n : 1));
  for (size_t i = 0; i < n; ++i) {
    const f4 p = x4_from_ext(&points[i]);
    f4 m = p;
    tables[8 * i] = x4_to_cached(&p);
    for (int j = 1; j < 8; ++j) { m = x4_add_cached(&m, &tables[8 * i]); tables[8 * i + j] = x4_to_cached(&m); }
    uint8_t red[32];
    orc_sc_reduce32(red, scalars + 32 * i);
    to_radix_16(digits + 64 * i, red);
  }
  f4 q = x4_identity();
  for (int j = 63; j >= 0; --j) {
    for (int k = 0; k < 4; ++k) q = x4_double(&q);
    for (size_t i = 0; i < n; ++i) {
      const int8_t d = digits[64 * i + j];
      const int sign = d < 0;
      const int mag = sign ? -d : d;
      f4 sel = x4_cached_identity();
      for (int k = 1; k <= 8; ++k) sel = x4_cmov(&sel, &tables[8 * i + k - 1], (uint64_t)0 - (uint64_t)(mag == k));      /* masked scan */
      const f4 neg = x4_cached_neg(&sel);
      sel = x4_cmov(&sel, &neg, (uint64_t)0 - (uint64_t)sign);
      q = x4_add_cached(&q, &sel);
    }
  }
  x4_to_ext(r, &q);
  free(tables); free(digits);
}

static void simd_straus_vartime(ge_ext* r, size_t n, const uint8_t* scalars, const ge_ext* points) {
  f4_init();
  f4* tables = (f4*)aligned_alloc(32, sizeof(f4) * 16 * (n ? n : 1));      /* [P, 3P .. 15P] and their negatives */
  int8_t* nafs = (int8_t*)malloc(256 * (n ? n : 1));
  for (size_t i = 0; i < n; ++i) {
    const f4 p = x4_from_ext(&points[i]);
    const f4 p2 = x4_double(&p);
    const f4 p2c = x4_to_cached(&p2);
    f4 m = p;
    for (int j = 0; j < 8; ++j) {
      tables[16 * i + j] = x4_to_cached(&m);
      tables[16 * i + 8 + j] = x4_cached_neg(&tables[16 * i + j]);
      m = x4_add_cached(&m, &p2c);
    }
    uint8_t red[32];
    orc_sc_reduce32(red, scalars + 32 * i);
    non_adjacent_form(nafs + 256 * i, red, 5);
  }
  f4 q = x4_identity();
  for (int i = 255; i >= 0; --i) {
    q = x4_double(&q);
    for (size_t k = 0; k < n; ++k) {
      const int8_t d = nafs[256 * k + i];
      if (d > 0) q = x4_add_cached(&q, &tables[16 * k + d / 2]);
      else if (d < 0) q = x4_add_cached(&q, &tables[16 * k + 8 + (-d) / 2]);
    }
  }
  x4_to_ext(r, &q);
  free(tables); free(nafs);
}

static void simd_pippenger(ge_ext* r, size_t n, const uint8_t* scalars, const ge_ext* points) {
  f4_init();
  const int w = n < 500 ? 6 : (n < 800 ? 7 : 8);
  const int buckets_count = (1 << w) / 2;
  int8_t* digits = (int8_t*)malloc(66 * (n ? n : 1));
  f4* pc = (f4*)aligned_alloc(32, sizeof(f4) * 2 * (n ? n : 1));          /* cached P and cached -P */
  f4* buckets = (f4*)aligned_alloc(32, sizeof(f4) * buckets_count);
  int digits_count = (256 + w - 1) / w + (w == 8 ? 1 : 0);
  for (size_t i = 0; i < n; ++i) {
    uint8_t red[32];
    orc_sc_reduce32(red, scalars + 32 * i);
    to_radix_2w(digits + 66 * i, red, w);
    const f4 p = x4_from_ext(&points[i]);
    pc[2 * i] = x4_to_cached(&p);
    pc[2 * i + 1] = x4_cached_neg(&pc[2 * i]);
  }
  f4 total = x4_identity();
  for (int di = digits_count - 1; di >= 0; --di) {
    for (int b = 0; b < buckets_count; ++b) buckets[b] = x4_identity();
    for (size_t i = 0; i < n; ++i) {
      const int d = digits[66 * i + di];
      if (d > 0) buckets[d - 1] = x4_add_cached(&buckets[d - 1], &pc[2 * i]);
      else if (d < 0) buckets[-d - 1] = x4_add_cached(&buckets[-d - 1], &pc[2 * i + 1]);
    }
    f4 inter = buckets[buckets_count - 1], sum = buckets[buckets_count - 1];
    for (int b = buckets_count - 2; b >= 0; --b) {
      const f4 bc_ = x4_to_cached(&buckets[b]);
      inter = x4_add_cached(&inter, &bc_);
      const f4 ic = x4_to_cached(&inter);
      sum = x4_add_cached(&sum, &ic);
    }
    if (di != digits_count - 1) for (int k = 0; k < w; ++k) total = x4_double(&total);
    const f4 sc_ = x4_to_cached(&sum);
    total = x4_add_cached(&total, &sc_);
  }
  x4_to_ext(r, &total);
  free(digits); free(pc); free(buckets);
}
#else
#define ORC_HAVE_IFMA 0
#endif

static int g_orc_simd = 0;
int orc_simd_available(void) {
#if ORC_HAVE_IFMA
  return __builtin_cpu_supports("avx512ifma") && __builtin_cpu_supports("avx512vl") ? 1 : 0;
#else
  return 0;
#endif
}
/* 1 = the MSM entry points of msm.c run on the vector backend (refused -> returns 0 -> when the CPU or the build lacks AVX-512 IFMA) */
int orc_set_simd(int on) {
  g_orc_simd = (on && orc_simd_available()) ? 1 : 0;
  return g_orc_simd;
}
