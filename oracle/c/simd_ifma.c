/* ORACLE (test infrastructure) -- the CPU baseline's MSM inner loops on AVX-512 IFMA or AVX2, 4 x 64-bit lanes = the four coordinates of ONE point.
 *
 * north_star asks for the reference's `simd_backend` next to the GPU numbers; the Rust crate cannot be built here (Cargo.toml:37-40 selects
 * curve25519-dalek's avx2 / ifma backends by feature; no toolchain, dependency not vendored).  This file restates that backend's published
 * design so that bench.py can print a MEASURED `cpu_baseline.simd` instead of a "1.5 - 2 x [RECALL]" note:
 *   - a vector of four field elements in radix 2^51 (dalek backend/vector/ifma/field.rs `F51x4Reduced`): products by vpmadd52luq / vpmadd52huq,
 *     the high halves entering one limb up with weight 2 (2^52 = 2 * 2^51);
 *   - point addition and doubling in the parallel formulas of Hisil-Wong-Carter-Dawson 2008 section 4 (dalek backend/vector/ifma/edwards.rs):
 *     two rounds of four simultaneous multiplications with lane shuffles in between;
 *   - on top of them the same three algorithms as the scalar port (msm.c): constant-time radix-16 Straus, width-5 NAF Straus, Pippenger w = 6/7/8.
 *   - the same on AVX2 for hosts without IFMA (dalek backend/vector/avx2/field.rs works in radix 2^25.5 with vpmuludq; here: ten limbs of 26 / 25
 *     bits, one limb per vector, the ref10 product table with 100 vpmuludq per four multiplications);
 * Decompression / compression (one inverse square root chain per point) stay scalar, as in dalek.
 * Results are KAT-equal to the scalar port (tests/test_oracle_c.py::test_simd_msm_equals_scalar_port, both instruction sets).  Selected at run time
 * by orc_set_simd(1) (best the CPU has) or orc_set_simd(2) (AVX2 even where IFMA exists); each section is compiled only when the compiler targets
 * its instruction set (oracle/Makefile: -march=native) and refused at run time on a CPU without it.  The point formulas and the three algorithms
 * are in simd_x4.inc, included once per instruction set. */
#if defined(__AVX2__)
#include <immintrin.h>
static inline __m256i bc(uint64_t x) { return _mm256_set1_epi64x((long long)x); }
/* lane shuffles: imm = l0 | l1 << 2 | l2 << 4 | l3 << 6 -- output lane k takes input lane l_k */
#define SH(l0, l1, l2, l3) ((l0) | ((l1) << 2) | ((l2) << 4) | ((l3) << 6))
enum { SH_1132 = SH(1, 1, 3, 2), SH_0000 = SH(0, 0, 0, 0), SH_1331 = SH(1, 3, 3, 1), SH_0220 = SH(0, 2, 2, 0), SH_3131 = SH(3, 1, 3, 1), SH_2020 = SH(2, 0, 2, 0),
       SH_0120 = SH(0, 1, 2, 0), SH_1111 = SH(1, 1, 1, 1), SH_3003 = SH(3, 0, 0, 3), SH_1032 = SH(1, 0, 2, 3) };
#endif

/* ======================================================= AVX-512 IFMA: radix 2^51, five limbs ======================================================= */
#if defined(__AVX512IFMA__) && defined(__AVX512VL__)
#define ORC_HAVE_IFMA 1
#define F4N 5
#define X4(n) n##_ifma
typedef struct { __m256i v[5]; } f4_ifma;            /* lane k of every limb = field element k; limbs < 2^52 wherever a product is taken */
/* carry pass, all limbs at once (dalek's F51x4Unreduced -> F51x4Reduced): every limb gives its bits above 51 to the next one, the top limb's
 * wrap to limb 0 with weight 19.  Inputs < 2^63 -> outputs < 2^51 + 2^17 < 2^52, which is all a product needs; not a canonical form. */
static inline void f4_reduce_ifma(f4_ifma* a) {
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
static inline f4_ifma f4_add_ifma(const f4_ifma* a, const f4_ifma* b) {
  f4_ifma r;
  for (int i = 0; i < 5; ++i) r.v[i] = _mm256_add_epi64(a->v[i], b->v[i]);
  f4_reduce_ifma(&r);
  return r;
}
/* a - b + 16 p (the scalar port's bias), reduced */
static inline f4_ifma f4_sub_ifma(const f4_ifma* a, const f4_ifma* b) {
  f4_ifma r;
  r.v[0] = _mm256_sub_epi64(_mm256_add_epi64(a->v[0], bc(36028797018963664ULL)), b->v[0]);
  for (int i = 1; i < 5; ++i) r.v[i] = _mm256_sub_epi64(_mm256_add_epi64(a->v[i], bc(36028797018963952ULL)), b->v[i]);
  f4_reduce_ifma(&r);
  return r;
}
/* four products at once; inputs: every limb < 2^52 */
static inline f4_ifma f4_mul_ifma(const f4_ifma* a, const f4_ifma* b) {
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
  f4_ifma r;
  for (int k = 0; k < 5; ++k) {
    const __m256i t = c[k + 5];
    r.v[k] = _mm256_add_epi64(c[k], _mm256_add_epi64(_mm256_slli_epi64(t, 4), _mm256_add_epi64(_mm256_slli_epi64(t, 1), t)));
  }
  f4_reduce_ifma(&r);
  return r;
}
static inline f4_ifma f4_sq_ifma(const f4_ifma* a) { return f4_mul_ifma(a, a); }
static inline f4_ifma f4_pack_ifma(const fe51* a, const fe51* b, const fe51* c, const fe51* d) {
  f4_ifma r;
  for (int i = 0; i < 5; ++i) r.v[i] = _mm256_set_epi64x((long long)d->v[i], (long long)c->v[i], (long long)b->v[i], (long long)a->v[i]);
  return r;
}
static inline void f4_unpack_ifma(fe51 out[4], const f4_ifma* a) {
  uint64_t t[4];
  for (int i = 0; i < 5; ++i) { _mm256_storeu_si256((__m256i*)t, a->v[i]); for (int k = 0; k < 4; ++k) out[k].v[i] = t[k]; }
  for (int k = 0; k < 4; ++k) fe_weak_reduce(&out[k]);
}
#include "simd_x4.inc"
#undef X4
#undef F4N
#else
#define ORC_HAVE_IFMA 0
#endif

/* ============================================ AVX2: radix 2^25.5, ten limbs of 26 / 25 bits, vpmuludq ============================================ */
#if defined(__AVX2__)
#define ORC_HAVE_AVX2 1
#define F4N 10
#define X4(n) n##_avx2
typedef struct { __m256i v[10]; } f4_avx2;           /* limb i (26 bits for even i, 25 for odd) in the low half of each 64-bit lane */
/* one carry pass, all limbs at once: limb i gives its bits above 26 / 25 to limb i + 1, limb 9 wraps to limb 0 with weight 19.
 * Inputs < 2^63.  After sums and differences of operands one pass leaves limbs < 2^26 + 2^7 / 2^25 + 2^7; a product's columns (< 2^61) take two. */
static inline void f4_reduce_avx2(f4_avx2* a) {
  const __m256i M26 = bc((1ull << 26) - 1), M25 = bc((1ull << 25) - 1);
  __m256i c[10];
  for (int i = 0; i < 10; ++i) c[i] = _mm256_srli_epi64(a->v[i], (i & 1) ? 25 : 26);
  const __m256i c9_19 = _mm256_add_epi64(_mm256_slli_epi64(c[9], 4), _mm256_add_epi64(_mm256_slli_epi64(c[9], 1), c[9]));
  a->v[0] = _mm256_add_epi64(_mm256_and_si256(a->v[0], M26), c9_19);
  for (int i = 1; i < 10; ++i) a->v[i] = _mm256_add_epi64(_mm256_and_si256(a->v[i], (i & 1) ? M25 : M26), c[i - 1]);
}
static inline f4_avx2 f4_add_avx2(const f4_avx2* a, const f4_avx2* b) {
  f4_avx2 r;
  for (int i = 0; i < 10; ++i) r.v[i] = _mm256_add_epi64(a->v[i], b->v[i]);
  f4_reduce_avx2(&r);
  return r;
}
/* a - b + 4 p, reduced; b: limbs <= 2^28 - 76 / 2^27 - 4 (anything add, sub or mul returned) */
static inline f4_avx2 f4_sub_avx2(const f4_avx2* a, const f4_avx2* b) {
  f4_avx2 r;
  r.v[0] = _mm256_sub_epi64(_mm256_add_epi64(a->v[0], bc((1ull << 28) - 76)), b->v[0]);
  for (int i = 1; i < 10; ++i) r.v[i] = _mm256_sub_epi64(_mm256_add_epi64(a->v[i], bc((i & 1) ? (1ull << 27) - 4 : (1ull << 28) - 4)), b->v[i]);
  f4_reduce_avx2(&r);
  return r;
}
/* four products at once, the ref10 table: h_k = sum_{i + j = k} f_i g_j [x 2 if i and j are odd] + 19 sum_{i + j = k + 10} f_i g_j [x 2 likewise].
 * Inputs: limbs < 2^26.1 / 2^25.1, so that 19 g_j and 2 f_i fit the 32 bits vpmuludq reads and a column stays below 2^61. */
static inline f4_avx2 f4_mul_avx2(const f4_avx2* a, const f4_avx2* b) {
  __m256i g19[10], f2[10], h[10];
  const __m256i k19 = bc(19);
  for (int j = 0; j < 10; ++j) g19[j] = _mm256_mul_epu32(b->v[j], k19);
  for (int i = 0; i < 10; ++i) f2[i] = _mm256_add_epi64(a->v[i], a->v[i]);
  for (int k = 0; k < 10; ++k) h[k] = _mm256_setzero_si256();
#pragma GCC unroll 10
  for (int i = 0; i < 10; ++i) {
#pragma GCC unroll 10
    for (int j = 0; j < 10; ++j) {
      const int k = i + j;
      const __m256i x = ((i & 1) && (j & 1)) ? f2[i] : a->v[i];
      const __m256i y = k >= 10 ? g19[j] : b->v[j];
      h[k >= 10 ? k - 10 : k] = _mm256_add_epi64(h[k >= 10 ? k - 10 : k], _mm256_mul_epu32(x, y));
    }
  }
  f4_avx2 r;
  for (int k = 0; k < 10; ++k) r.v[k] = h[k];
  f4_reduce_avx2(&r);
  f4_reduce_avx2(&r);
  return r;
}
/* four squares at once: 55 products instead of 100 (the ref10 squaring table: cross terms doubled, odd-odd terms doubled again, wrapped terms x 19) */
static inline f4_avx2 f4_sq_avx2(const f4_avx2* a) {
  __m256i f2[10], f19[10], f38[10], h[10];
  const __m256i k19 = bc(19);
  for (int i = 0; i < 10; ++i) {
    f2[i] = _mm256_add_epi64(a->v[i], a->v[i]);
    f19[i] = _mm256_mul_epu32(a->v[i], k19);
    f38[i] = _mm256_add_epi64(f19[i], f19[i]);
  }
  for (int k = 0; k < 10; ++k) h[k] = _mm256_setzero_si256();
#pragma GCC unroll 10
  for (int i = 0; i < 10; ++i) {
#pragma GCC unroll 10
    for (int j = i; j < 10; ++j) {
      const int k = i + j, wrap = k >= 10, c = (i == j ? 1 : 2) * (((i & 1) && (j & 1)) ? 2 : 1);      /* 1, 2 or 4 */
      const __m256i x = c >= 2 ? f2[i] : a->v[i];
      const __m256i y = wrap ? (c == 4 ? f38[j] : f19[j]) : (c == 4 ? f2[j] : a->v[j]);
      h[wrap ? k - 10 : k] = _mm256_add_epi64(h[wrap ? k - 10 : k], _mm256_mul_epu32(x, y));
    }
  }
  f4_avx2 r;
  for (int k = 0; k < 10; ++k) r.v[k] = h[k];
  f4_reduce_avx2(&r);
  f4_reduce_avx2(&r);
  return r;
}
/* a 51-bit limb is a 26-bit limb and the 25-bit limb above it */
static inline f4_avx2 f4_pack_avx2(const fe51* a, const fe51* b, const fe51* c, const fe51* d) {
  f4_avx2 r;
  for (int i = 0; i < 5; ++i) {
    const __m256i v = _mm256_set_epi64x((long long)d->v[i], (long long)c->v[i], (long long)b->v[i], (long long)a->v[i]);
    r.v[2 * i] = _mm256_and_si256(v, bc((1ull << 26) - 1));
    r.v[2 * i + 1] = _mm256_srli_epi64(v, 26);
  }
  return r;
}
static inline void f4_unpack_avx2(fe51 out[4], const f4_avx2* a) {
  uint64_t lo[4], hi[4];
  for (int i = 0; i < 5; ++i) {
    _mm256_storeu_si256((__m256i*)lo, a->v[2 * i]);
    _mm256_storeu_si256((__m256i*)hi, a->v[2 * i + 1]);
    for (int k = 0; k < 4; ++k) out[k].v[i] = lo[k] + (hi[k] << 26);
  }
  for (int k = 0; k < 4; ++k) fe_weak_reduce(&out[k]);
}
#include "simd_x4.inc"
#undef X4
#undef F4N

/* ---- AVX2, PACKED: curve25519-dalek's FieldElement2625x4 layout (backend/vector/avx2/field.rs) ----------------------------------------------
 * Five vectors of eight 32-bit lanes hold four field elements A, B, C, D:  vector i = (A_2i, B_2i, A_2i+1, B_2i+1 | C_2i, D_2i, C_2i+1, D_2i+1),
 * limbs of 26 / 25 bits.  A product unpacks both operands into the ten one-limb-per-vector operands of the section above (vpunpckl/hdq against zero),
 * runs the same 100 vpmuludq and repacks -- exactly what dalek's Mul does -- so the multiplications cost what they cost above; what the packing
 * halves is everything around them: additions, subtractions and their carry passes (5 vectors instead of 10), the lane shuffles and blends of the
 * 4-way point formulas (vpermd on 5 vectors), table entries (160 instead of 320 bytes) and the register pressure between two products. */
#define F4N 5
#define X4(n) n##_avx2p
#define X4_OWN_LANES 1
typedef struct { __m256i v[5]; } f4_avx2p;
/* position of (element e, limb parity p) inside a packed vector */
#define PL(e, p) (((e) & 1) + 2 * (p) + 4 * ((e) >> 1))
static inline __m256i pk_idx(int l0, int l1, int l2, int l3) {         /* vpermd index: output element k takes input element l_k, both limb parities */
  int ix[8];
  const int l[4] = {l0, l1, l2, l3};
  for (int k = 0; k < 4; ++k) for (int p = 0; p < 2; ++p) ix[PL(k, p)] = PL(l[k], p);
  return _mm256_setr_epi32(ix[0], ix[1], ix[2], ix[3], ix[4], ix[5], ix[6], ix[7]);
}
static inline f4_avx2p shuf_avx2p(const f4_avx2p* a, const int imm) {
  const __m256i ix = pk_idx(imm & 3, (imm >> 2) & 3, (imm >> 4) & 3, (imm >> 6) & 3);
  f4_avx2p r;
  for (int i = 0; i < 5; ++i) r.v[i] = _mm256_permutevar8x32_epi32(a->v[i], ix);
  return r;
}
static inline f4_avx2p blend_avx2p(const f4_avx2p* a, const f4_avx2p* b, const int lanes) {
  int m[8];
  for (int k = 0; k < 4; ++k) for (int p = 0; p < 2; ++p) m[PL(k, p)] = (lanes >> k) & 1 ? -1 : 0;
  const __m256i mv = _mm256_setr_epi32(m[0], m[1], m[2], m[3], m[4], m[5], m[6], m[7]);
  f4_avx2p r;
  for (int i = 0; i < 5; ++i) r.v[i] = _mm256_blendv_epi8(a->v[i], b->v[i], mv);
  return r;
}
static inline f4_avx2p f4_addraw_avx2p(const f4_avx2p* a, const f4_avx2p* b) { f4_avx2p r; for (int i = 0; i < 5; ++i) r.v[i] = _mm256_add_epi32(a->v[i], b->v[i]); return r; }
/* one carry pass over the packed form: the even limbs of a vector (lanes 0, 1, 4, 5) carry into its odd limbs (lanes 2, 3, 6, 7), the odd limbs into
 * the even limbs of the NEXT vector, limb 9 wraps to limb 0 with weight 19.  Inputs < 2^31 per lane. */
static inline void f4_reduce_avx2p(f4_avx2p* a) {
  const __m256i sh = _mm256_setr_epi32(26, 26, 25, 25, 26, 26, 25, 25);
  const __m256i mk = _mm256_setr_epi32((1 << 26) - 1, (1 << 26) - 1, (1 << 25) - 1, (1 << 25) - 1, (1 << 26) - 1, (1 << 26) - 1, (1 << 25) - 1, (1 << 25) - 1);
  __m256i rot[5];
  for (int i = 0; i < 5; ++i) rot[i] = _mm256_shuffle_epi32(_mm256_srlv_epi32(a->v[i], sh), _MM_SHUFFLE(1, 0, 3, 2));      /* (odd carries | even carries) per half */
  const __m256i w = rot[4];
  const __m256i w19 = _mm256_add_epi32(_mm256_slli_epi32(w, 4), _mm256_add_epi32(_mm256_slli_epi32(w, 1), w));
  for (int i = 0; i < 5; ++i) {
    const __m256i from_prev = i ? rot[i - 1] : w19;                      /* lanes 0, 1, 4, 5: the odd-limb carries of the vector before */
    a->v[i] = _mm256_add_epi32(_mm256_and_si256(a->v[i], mk), _mm256_blend_epi32(from_prev, rot[i], 0xCC));
  }
}
static inline f4_avx2p f4_add_avx2p(const f4_avx2p* a, const f4_avx2p* b) {
  f4_avx2p r = f4_addraw_avx2p(a, b);
  f4_reduce_avx2p(&r);
  return r;
}
/* a - b + 4 p, reduced (same bias as the unpacked section) */
static inline f4_avx2p f4_sub_avx2p(const f4_avx2p* a, const f4_avx2p* b) {
  const int E = (1 << 28) - 4, O = (1 << 27) - 4;
  const __m256i bias = _mm256_setr_epi32(E, E, O, O, E, E, O, O), bias0 = _mm256_setr_epi32(E - 72, E - 72, O, O, E - 72, E - 72, O, O);
  f4_avx2p r;
  for (int i = 0; i < 5; ++i) r.v[i] = _mm256_sub_epi32(_mm256_add_epi32(a->v[i], i ? bias : bias0), b->v[i]);
  f4_reduce_avx2p(&r);
  return r;
}
static inline f4_avx2 pk_unpack(const f4_avx2p* a) {                     /* -> one limb per vector, 64-bit lanes (A, B | C, D) */
  const __m256i z = _mm256_setzero_si256();
  f4_avx2 r;
  for (int i = 0; i < 5; ++i) { r.v[2 * i] = _mm256_unpacklo_epi32(a->v[i], z); r.v[2 * i + 1] = _mm256_unpackhi_epi32(a->v[i], z); }
  return r;
}
static inline f4_avx2p pk_pack(const f4_avx2* a) {                       /* limbs < 2^32 */
  f4_avx2p r;
  for (int i = 0; i < 5; ++i)
    r.v[i] = _mm256_or_si256(_mm256_shuffle_epi32(a->v[2 * i], _MM_SHUFFLE(3, 1, 2, 0)), _mm256_shuffle_epi32(a->v[2 * i + 1], _MM_SHUFFLE(2, 0, 3, 1)));
  return r;
}
static inline f4_avx2p f4_mul_avx2p(const f4_avx2p* a, const f4_avx2p* b) {
  const f4_avx2 x = pk_unpack(a), y = pk_unpack(b);
  const f4_avx2 h = f4_mul_avx2(&x, &y);
  return pk_pack(&h);
}
static inline f4_avx2p f4_sq_avx2p(const f4_avx2p* a) {
  const f4_avx2 x = pk_unpack(a);
  const f4_avx2 h = f4_sq_avx2(&x);
  return pk_pack(&h);
}
static inline f4_avx2p f4_pack_avx2p(const fe51* a, const fe51* b, const fe51* c, const fe51* d) {
  const f4_avx2 u = f4_pack_avx2(a, b, c, d);
  return pk_pack(&u);
}
static inline void f4_unpack_avx2p(fe51 out[4], const f4_avx2p* a) {
  const f4_avx2 u = pk_unpack(a);
  f4_unpack_avx2(out, &u);
}
#include "simd_x4.inc"
#undef PL
#undef X4_OWN_LANES
#undef X4
#undef F4N
#else
#define ORC_HAVE_AVX2 0
#endif

/* 0 = scalar port, 1 = AVX-512 IFMA, 2 = AVX2 (one limb per vector), 3 = AVX2 packed (dalek's FieldElement2625x4 layout) */
static int g_orc_simd = 0;
/* best vector instruction set the build AND this CPU have: 1 = AVX-512 IFMA + VL, 2 = AVX2, 0 = none */
int orc_simd_available(void) {
#if ORC_HAVE_IFMA
  if (__builtin_cpu_supports("avx512ifma") && __builtin_cpu_supports("avx512vl")) return 1;
#endif
#if ORC_HAVE_AVX2
  if (__builtin_cpu_supports("avx2")) return 2;
#endif
  return 0;
}
/* mode 0: scalar port; 1: the best vector backend there is; 2: AVX2 even where IFMA exists.  Returns what the MSM entry points of msm.c run on
 * from now on (0 = scalar: the request was refused because the CPU or the build lacks the instruction set) */
int orc_set_simd(int mode) {
  const int best = orc_simd_available();
  g_orc_simd = 0;
  if (mode == 1) g_orc_simd = best;
#if ORC_HAVE_AVX2
  if ((mode == 2 || mode == 3) && best) g_orc_simd = mode;      /* (IFMA implies AVX2) */
#endif
  return g_orc_simd;
}
int orc_simd_mode(void) { return g_orc_simd; }
