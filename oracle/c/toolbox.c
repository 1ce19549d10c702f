/* ORACLE (test infrastructure): the toolbox flow of the reference, followed step by step:
 *   TranscriptProtocol   src/toolbox/mod.rs:165-228
 *   Prover::prove_impl   src/toolbox/prover.rs:76-112   (+ allocate_* :53-73, new :41-50)
 *   Verifier             src/toolbox/verifier.rs:47-173
 *   BatchVerifier        src/toolbox/batch_verifier.rs:67-235, Matrix layout src/util.rs:21-37
 * with the macro's allocation order (src/macros.rs:206-258: secrets, instance points, common points).
 * Randomness the reference draws from thread_rng is passed in, so results are reproducible. */
#include <stdlib.h>
#include <string.h>
#include "oracle.h"

static const uint8_t ZERO32[32] = {0};

static void domain_sep(orc_transcript* t, const char* label) {                      /* mod.rs:166-169 */
  orc_transcript_append(t, "dom-sep", (const uint8_t*)"schnorrzkp/1.0/ristretto255", 27);
  orc_transcript_append(t, "dom-sep", (const uint8_t*)label, strlen(label));
}
static void append_scalar_var(orc_transcript* t, const char* label) {               /* mod.rs:171-173 */
  orc_transcript_append(t, "scvar", (const uint8_t*)label, strlen(label));
}
static void append_point(orc_transcript* t, const char* kind, const char* label, const uint8_t enc[32]) {
  orc_transcript_append(t, kind, (const uint8_t*)label, strlen(label));             /* mod.rs:181-182 etc. */
  orc_transcript_append(t, "val", enc, 32);
}
static int validate_and_append(orc_transcript* t, const char* kind, const char* label, const uint8_t enc[32]) {
  if (memcmp(enc, ZERO32, 32) == 0) return 1;                                       /* mod.rs:191-193, 215-217 */
  append_point(t, kind, label, enc);
  return 0;
}
static void get_challenge(orc_transcript* t, uint8_t out[32]) {                     /* mod.rs:223-227 */
  uint8_t wide[64];
  orc_transcript_challenge(t, "chal", wide, 64);
  orc_sc_from_wide(out, wide);
}

int orc_prove(const orc_statement* st, const uint8_t* tl, size_t tl_len, const uint8_t* secrets,
              const uint8_t* points, const uint8_t entropy32[32], uint8_t challenge[32], uint8_t* responses,
              uint8_t* commitments, uint8_t* blindings_out) {
  const uint32_t m = st->n_secrets, np = st->n_inst + st->n_common;
  orc_transcript t;
  orc_transcript_init(&t, tl, tl_len);
  domain_sep(&t, st->label);                                                        /* prover.rs:42 */
  for (uint32_t i = 0; i < m; ++i) append_scalar_var(&t, st->secret_names[i]);      /* prover.rs:54 */
  ge_ext* P = (ge_ext*)malloc(sizeof(ge_ext) * (np ? np : 1));
  for (uint32_t i = 0; i < np; ++i) {                                               /* prover.rs:69: compress + append */
    if (!orc_ristretto_decode(&P[i], points + 32 * i)) { free(P); return -1; }
    uint8_t enc[32];
    orc_ristretto_encode(enc, &P[i]);
    append_point(&t, "ptvar", st->point_names[i], enc);
  }
  /* prover.rs:78-82: transcript rng keyed with every secret, then with external entropy */
  orc_strobe rng = t.s;
  for (uint32_t i = 0; i < m; ++i) rng_rekey(&rng, "", secrets + 32 * i, 32);
  rng_finalize(&rng, entropy32);
  uint8_t* b = (uint8_t*)malloc(32 * (m ? m : 1));
  for (uint32_t i = 0; i < m; ++i) {                                                /* prover.rs:85-89 */
    uint8_t wide[64];
    rng_fill(&rng, wide, 64);
    orc_sc_from_wide(b + 32 * i, wide);
  }
  if (blindings_out) memcpy(blindings_out, b, 32 * m);
  /* prover.rs:92-103: one constant-time MSM per constraint */
  for (uint32_t k = 0; k < st->n_cons; ++k) {
    const uint32_t lo = st->cons_off[k], cnt = st->cons_off[k + 1] - lo;
    uint8_t* sc = (uint8_t*)malloc(32 * (cnt ? cnt : 1));
    ge_ext* pt = (ge_ext*)malloc(sizeof(ge_ext) * (cnt ? cnt : 1));
    for (uint32_t j = 0; j < cnt; ++j) { memcpy(sc + 32 * j, b + 32 * st->cons_sc[lo + j], 32); pt[j] = P[st->cons_pt[lo + j]]; }
    ge_ext com;
    orc_msm_straus_ct(&com, cnt, sc, pt);
    orc_ristretto_encode(commitments + 32 * k, &com);
    append_point(&t, "blindcom", st->point_names[st->cons_lhs[k]], commitments + 32 * k);
    free(sc); free(pt);
  }
  get_challenge(&t, challenge);                                                     /* prover.rs:106 */
  for (uint32_t i = 0; i < m; ++i) orc_sc_muladd(responses + 32 * i, secrets + 32 * i, challenge, b + 32 * i);   /* :107-109 */
  free(P); free(b);
  return 0;
}

static int build_verifier(orc_transcript* t, const orc_statement* st, const uint8_t* tl, size_t tl_len, const uint8_t* points) {
  orc_transcript_init(t, tl, tl_len);
  domain_sep(t, st->label);                                                         /* verifier.rs:48 */
  for (uint32_t i = 0; i < st->n_secrets; ++i) append_scalar_var(t, st->secret_names[i]);
  for (uint32_t i = 0; i < st->n_inst + st->n_common; ++i)                          /* verifier.rs:72-73 */
    if (validate_and_append(t, "ptvar", st->point_names[i], points + 32 * i)) return 1;
  return 0;
}

int orc_verify_compact(const orc_statement* st, const uint8_t* tl, size_t tl_len, const uint8_t* points,
                       const uint8_t challenge[32], const uint8_t* responses) {
  const uint32_t np = st->n_inst + st->n_common;
  if (!orc_sc_is_canonical(challenge)) return 1;                                    /* proofs.rs:15-20 through serde: Scalars are canonical */
  for (uint32_t i = 0; i < st->n_secrets; ++i) if (!orc_sc_is_canonical(responses + 32 * i)) return 1;
  orc_transcript t;
  if (build_verifier(&t, st, tl, tl_len, points)) return 1;
  ge_ext* P = (ge_ext*)malloc(sizeof(ge_ext) * (np ? np : 1));
  for (uint32_t i = 0; i < np; ++i)                                                 /* verifier.rs:87-92 */
    if (!orc_ristretto_decode(&P[i], points + 32 * i)) { free(P); return 1; }
  uint8_t minus_c[32];
  orc_sc_neg(minus_c, challenge);                                                   /* verifier.rs:95 */
  for (uint32_t k = 0; k < st->n_cons; ++k) {                                       /* verifier.rs:96-110 */
    const uint32_t lo = st->cons_off[k], cnt = st->cons_off[k + 1] - lo;
    uint8_t* sc = (uint8_t*)malloc(32 * (cnt + 1));
    ge_ext* pt = (ge_ext*)malloc(sizeof(ge_ext) * (cnt + 1));
    for (uint32_t j = 0; j < cnt; ++j) { memcpy(sc + 32 * j, responses + 32 * st->cons_sc[lo + j], 32); pt[j] = P[st->cons_pt[lo + j]]; }
    memcpy(sc + 32 * cnt, minus_c, 32);
    pt[cnt] = P[st->cons_lhs[k]];
    ge_ext com;
    orc_msm_vartime(&com, cnt + 1, sc, pt);
    uint8_t enc[32];
    orc_ristretto_encode(enc, &com);
    append_point(&t, "blindcom", st->point_names[st->cons_lhs[k]], enc);            /* non-validating append, :108 */
    free(sc); free(pt);
  }
  uint8_t c2[32], cred[32];
  get_challenge(&t, c2);
  orc_sc_reduce32(cred, challenge);
  free(P);
  return memcmp(c2, cred, 32) == 0 ? 0 : 1;                                         /* verifier.rs:113-119 */
}

static void sc_from_u128(uint8_t out[32], const uint8_t w16[16]) { memset(out, 0, 32); memcpy(out, w16, 16); }   /* Scalar::from(u128) */

int orc_verify_batchable(const orc_statement* st, const uint8_t* tl, size_t tl_len, const uint8_t* points,
                         const uint8_t* commitments, const uint8_t* responses, const uint8_t* weights16) {
  const uint32_t np = st->n_inst + st->n_common, nc = st->n_cons;
  for (uint32_t i = 0; i < st->n_secrets; ++i) if (!orc_sc_is_canonical(responses + 32 * i)) return 1;   /* proofs.rs:27-32 through serde */
  orc_transcript t;
  if (build_verifier(&t, st, tl, tl_len, points)) return 1;
  for (uint32_t k = 0; k < nc; ++k)                                                 /* verifier.rs:134-140 */
    if (validate_and_append(&t, "blindcom", st->point_names[st->cons_lhs[k]], commitments + 32 * k)) return 1;
  uint8_t c[32], minus_c[32];
  get_challenge(&t, c);
  orc_sc_neg(minus_c, c);                                                           /* verifier.rs:142 */
  uint8_t* coeffs = (uint8_t*)calloc(np + nc, 32);
  uint8_t* encs = (uint8_t*)malloc(32 * (np + nc));
  memcpy(encs, points, 32 * np);
  memcpy(encs + 32 * np, commitments, 32 * nc);
  for (uint32_t k = 0; k < nc; ++k) {                                               /* verifier.rs:151-160 */
    uint8_t r[32], tmp[32];
    sc_from_u128(r, weights16 + 16 * k);
    orc_sc_sub(coeffs + 32 * (np + k), coeffs + 32 * (np + k), r);
    uint8_t* lhs = coeffs + 32 * st->cons_lhs[k];
    orc_sc_muladd(tmp, r, minus_c, lhs); memcpy(lhs, tmp, 32);
    for (uint32_t j = st->cons_off[k]; j < st->cons_off[k + 1]; ++j) {
      uint8_t* cp = coeffs + 32 * st->cons_pt[j];
      orc_sc_muladd(tmp, r, responses + 32 * st->cons_sc[j], cp); memcpy(cp, tmp, 32);
    }
  }
  uint8_t out[32];
  int status = 1;
  orc_msm_optional(np + nc, coeffs, encs, out, &status);                            /* verifier.rs:162-166 */
  free(coeffs); free(encs);
  if (status) return 1;
  return memcmp(out, ZERO32, 32) == 0 ? 0 : 1;                                      /* verifier.rs:168 */
}

int orc_batch_verify(const orc_statement* st, const uint8_t* tl, size_t tl_len, uint32_t N,
                     const uint8_t* inst_points, const uint8_t* common_points, const uint8_t* commitments,
                     const uint8_t* responses, const uint8_t* weights16, uint8_t* msm_scalars, uint8_t* msm_points) {
  const uint32_t ni = st->n_inst, ns = st->n_common, nc = st->n_cons, m = st->n_secrets, np = ni + ns;
  for (size_t i = 0; i < (size_t)N * m; ++i) if (!orc_sc_is_canonical(responses + 32 * i)) return 1;   /* proofs.rs:27-32 through serde */
  /* kind / rank of every point variable in allocation order */
  uint8_t* is_common = (uint8_t*)malloc(np ? np : 1);
  uint32_t* rank = (uint32_t*)malloc(4 * (np ? np : 1));
  { uint32_t ri = 0, rs = 0;
    for (uint32_t p = 0; p < np; ++p) {
      is_common[p] = st->point_is_common ? st->point_is_common[p] : (uint8_t)(p >= ni);
      rank[p] = is_common[p] ? rs++ : ri++;
    }
    if (ri != ni || rs != ns) { free(is_common); free(rank); return 2; } }
  orc_transcript* ts = (orc_transcript*)malloc(sizeof(orc_transcript) * (N ? N : 1));
  int rc = 0;
  for (uint32_t j = 0; j < N; ++j) { orc_transcript_init(&ts[j], tl, tl_len); domain_sep(&ts[j], st->label); }   /* :75-77 */
  for (uint32_t i = 0; i < m; ++i) for (uint32_t j = 0; j < N; ++j) append_scalar_var(&ts[j], st->secret_names[i]);   /* :92-94 */
  for (uint32_t p = 0; p < np && !rc; ++p)                                          /* :105-107 static, :125-128 instance */
    for (uint32_t j = 0; j < N; ++j) {
      const uint8_t* enc = is_common[p] ? common_points + 32 * (size_t)rank[p] : inst_points + 32 * ((size_t)rank[p] * N + j);
      if (validate_and_append(&ts[j], "ptvar", st->point_names[p], enc)) { rc = 1; break; }
    }
  for (uint32_t j = 0; j < N && !rc; ++j)                                           /* :152-160 */
    for (uint32_t k = 0; k < nc; ++k)
      if (validate_and_append(&ts[j], "blindcom", st->point_names[st->cons_lhs[k]], commitments + 32 * ((size_t)j * nc + k))) { rc = 1; break; }
  if (rc) { free(ts); free(is_common); free(rank); return 1; }
  uint8_t* minus_c = (uint8_t*)malloc(32 * (size_t)(N ? N : 1));
  for (uint32_t j = 0; j < N; ++j) { uint8_t c[32]; get_challenge(&ts[j], c); orc_sc_neg(minus_c + 32 * j, c); }   /* :163-167 */
  free(ts);
  const size_t rows = ni + nc, total = ns + rows * N;
  uint8_t* sc = (uint8_t*)calloc(total ? total : 1, 32);       /* static_coeffs || Matrix(rows, N) row-major (util.rs:24) */
  uint8_t* statics = sc;
  uint8_t* inst = sc + 32 * (size_t)ns;
  for (uint32_t k = 0; k < nc; ++k) {                                               /* :176-206 */
    const uint32_t lhs = st->cons_lhs[k];
    for (uint32_t j = 0; j < N; ++j) {
      uint8_t r[32], tmp[32];
      sc_from_u128(r, weights16 + 16 * ((size_t)k * N + j));
      uint8_t* e = inst + 32 * ((size_t)(ni + k) * N + j);
      orc_sc_sub(e, e, r);                                                          /* :183 */
      uint8_t* dst = !is_common[lhs] ? inst + 32 * ((size_t)rank[lhs] * N + j) : statics + 32 * (size_t)rank[lhs];
      orc_sc_muladd(tmp, r, minus_c + 32 * j, dst); memcpy(dst, tmp, 32);           /* :185-192 */
      for (uint32_t q = st->cons_off[k]; q < st->cons_off[k + 1]; ++q) {            /* :194-204 */
        const uint32_t pv = st->cons_pt[q];
        uint8_t* d2 = !is_common[pv] ? inst + 32 * ((size_t)rank[pv] * N + j) : statics + 32 * (size_t)rank[pv];
        orc_sc_muladd(tmp, r, responses + 32 * ((size_t)j * m + st->cons_sc[q]), d2); memcpy(d2, tmp, 32);
      }
    }
  }
  uint8_t* pts = (uint8_t*)malloc(32 * (total ? total : 1));                        /* :208-217, :224-225 */
  memcpy(pts, common_points, 32 * (size_t)ns);
  memcpy(pts + 32 * (size_t)ns, inst_points, 32 * (size_t)ni * N);
  for (uint32_t k = 0; k < nc; ++k)
    for (uint32_t j = 0; j < N; ++j)
      memcpy(pts + 32 * ((size_t)ns + (size_t)(ni + k) * N + j), commitments + 32 * ((size_t)j * nc + k), 32);
  free(minus_c); free(is_common); free(rank);
  if (msm_scalars && msm_points) {
    memcpy(msm_scalars, sc, 32 * total);
    memcpy(msm_points, pts, 32 * total);
    free(sc); free(pts);
    return 0;
  }
  uint8_t out[32];
  int status = 1;
  orc_msm_optional(total, sc, pts, out, &status);                                   /* :219-228 */
  free(sc); free(pts);
  if (status) return 1;
  return memcmp(out, ZERO32, 32) == 0 ? 0 : 1;                                      /* :230-234 */
}
