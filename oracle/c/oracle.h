/* ORACLE -- test infrastructure, NOT product code.
 *
 * Plain-C CPU restatement of the hot path of dalek-cryptography/zkp and of the curve25519-dalek 2.x
 * (u64 backend) / merlin 2.x algorithms behind it.  Only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may load this library; the product (zkp_amd/, libzkp_mi355x.so) never
 * links, loads or calls it.
 *
 * curve25519-dalek `= "2"` and merlin `= "2"` are NOT vendored under /root/reference (Cargo.toml:21,27;
 * no lockfile).  Their published algorithms are restated here: FieldElement51 (5 x 51-bit limbs,
 * u128 products), Scalar arithmetic mod l, RFC 9496 ristretto255, constant-time radix-16 Straus
 * (`multiscalar_mul`, reference call site prover.rs:94), width-5 NAF Straus and Pippenger with
 * w = 6/7/8 behind the 190-term threshold (`vartime_multiscalar_mul` / `optional_multiscalar_mul`,
 * call sites verifier.rs:97,162 and batch_verifier.rs:219), STROBE-128 / Merlin, and the toolbox flow
 * of src/toolbox/{mod,prover,verifier,batch_verifier}.rs.
 *
 * PIN STATUS -- parity unpinned against the reference binary (no Rust toolchain, dependencies not vendored: no oracle/_ref;
 * RFC 9496 A.1-A.3, Merlin's KAT and libsodium pin the layers underneath): the reference holds no golden bytes for this path (all proofs are randomised,
 * prover.rs:82).  This library is pinned by tests/test_oracle_c.py against the RFC 9496 vectors,
 * Merlin's known-answer test, the big-integer model oracle/model.py (itself cross-checked against
 * libsodium 1.0.18) and the committed fixtures in tests/golden/.
 * What would pin it against the crate is staged: tests/golden/interop/ + rust/interop/ (the real crate verifies this side's proofs and emits
 * its own for tests/test_oracle_interop.py / tests/test_gpu_interop.py); until somebody runs it with cargo the status above stands.
 */
#ifndef ZKP_ORACLE_H
#define ZKP_ORACLE_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- field / group / scalars ------------------------------------------------------------- */
typedef struct { uint64_t v[5]; } fe51;
typedef struct { fe51 X, Y, Z, T; } ge_ext;

int  orc_ristretto_decode(ge_ext* r, const uint8_t enc[32]);           /* 1 = valid */
void orc_ristretto_encode(uint8_t enc[32], const ge_ext* p);
void orc_ristretto_from_uniform_bytes(ge_ext* r, const uint8_t b[64]);
void orc_ge_add(ge_ext* r, const ge_ext* p, const ge_ext* q);
void orc_ge_double(ge_ext* r, const ge_ext* p);
void orc_ge_identity(ge_ext* r);
void orc_ge_basepoint(ge_ext* r);

void orc_sc_from_wide(uint8_t out[32], const uint8_t in[64]);           /* Scalar::from_bytes_mod_order_wide */
void orc_sc_reduce32(uint8_t out[32], const uint8_t in[32]);
int  orc_sc_is_canonical(const uint8_t in[32]);                            /* 1 = value < l (what serde lets through as a Scalar) */
void orc_sc_muladd(uint8_t out[32], const uint8_t a[32], const uint8_t b[32], const uint8_t c[32]); /* a*b+c */
void orc_sc_neg(uint8_t out[32], const uint8_t a[32]);
void orc_sc_add(uint8_t out[32], const uint8_t a[32], const uint8_t b[32]);
void orc_sc_sub(uint8_t out[32], const uint8_t a[32], const uint8_t b[32]);

/* The MSM entry points below on vectors (simd_ifma.c + simd_x4.inc: dalek's simd_backend design, 4 lanes = the 4 coordinates of a point).
 * orc_simd_available(): 1 = AVX-512 IFMA + VL, 2 = AVX2 only, 0 = neither (build or CPU).  orc_set_simd(1) switches to the best of them,
 * orc_set_simd(2) to AVX2 even where IFMA exists, orc_set_simd(0) back to the scalar port; the return value (and orc_simd_mode()) is what
 * runs from now on: 0 scalar, 1 IFMA, 2 AVX2 -- 0 after a request the CPU or the build cannot serve (nothing changes then). */
int orc_simd_available(void);
int orc_set_simd(int mode);
int orc_simd_mode(void);

/* ---- the three dalek MSM entry points ---------------------------------------------------- */
/* constant-time Straus, radix 16  (RistrettoPoint::multiscalar_mul) */
void orc_msm_straus_ct(ge_ext* r, size_t n, const uint8_t* scalars, const ge_ext* points);
/* vartime Straus, NAF width 5     (optional_multiscalar_mul, size < 190) */
void orc_msm_straus_vartime(ge_ext* r, size_t n, const uint8_t* scalars, const ge_ext* points);
/* vartime Pippenger, w = 6 | 7 | 8 (optional_multiscalar_mul, size >= 190) */
void orc_msm_pippenger(ge_ext* r, size_t n, const uint8_t* scalars, const ge_ext* points);
/* dalek's dispatch: Straus below 190 terms, Pippenger from 190 */
void orc_msm_vartime(ge_ext* r, size_t n, const uint8_t* scalars, const ge_ext* points);

/* Same contracts as the product C ABI (include/zkp_mi355x.h), computed on the CPU: used as the
 * expected value in parity tests and as the timed CPU baseline. */
int orc_msm_many(uint32_t n_msm, const uint32_t* off, const uint8_t* scalars, const uint32_t* pidx,
                 const uint8_t* points, uint32_t n_points, int flags, uint8_t* out, uint8_t* status);
int orc_msm_optional(uint64_t n, const uint8_t* scalars, const uint8_t* points, uint8_t out_point[32], int* status);
int orc_decode_check(uint64_t n, const uint8_t* points, uint8_t* status, uint8_t* xyzt);
int orc_encode_many(uint64_t n, const uint8_t* xyzt, uint8_t* out);

/* ---- Merlin ------------------------------------------------------------------------------- */
typedef struct { uint8_t st[200]; uint8_t pos, pos_begin, cur_flags; } orc_strobe;
typedef struct { orc_strobe s; } orc_transcript;
void orc_transcript_init(orc_transcript* t, const uint8_t* label, size_t len);
void orc_transcript_append(orc_transcript* t, const char* label, const uint8_t* msg, size_t len);
void orc_transcript_challenge(orc_transcript* t, const char* label, uint8_t* out, size_t len);
uint64_t orc_keccak_count(void);

/* ---- toolbox: statement descriptor + prover / verifier / batch verifier ------------------ */
/* A statement fixes what define_proof! fixes (macros.rs:124-138,206-258): proof label, secret names,
 * point names (instance first, then common), constraints as CSR over (secret idx, point idx).
 * Point indices are in allocation order.  The macro allocates instance points first, then common
 * points (point_is_common = NULL); hand-written statements (benches/dleq.rs:188-241 allocates the
 * static G, H before the instance A, B) give the kind of every point explicitly. */
typedef struct {
  const char* label;
  uint32_t n_secrets, n_inst, n_common, n_cons;
  const char* const* secret_names;
  const char* const* point_names;        /* n_inst + n_common, in ALLOCATION order           */
  const uint8_t* point_is_common;        /* [n_inst+n_common] 1 = common/static point, or NULL */
                                         /*  = macro order (instance points first, then common) */
  const uint32_t* cons_lhs;              /* [n_cons] point index of the left-hand side      */
  const uint32_t* cons_off;              /* [n_cons+1] */
  const uint32_t* cons_sc;               /* [T] secret index  */
  const uint32_t* cons_pt;               /* [T] point index   */
} orc_statement;

/* Prover::prove_impl (prover.rs:76-112) for ONE proof.  points = encodings of all n_inst+n_common
 * points (as the macro's CompressedPoints); entropy32 replaces the 32 bytes drawn from thread_rng
 * at prover.rs:82.  Outputs: challenge[32], responses[n_secrets][32], commitments[n_cons][32].
 * blindings_out (optional, [n_secrets][32]) exposes the blinding scalars for cross-checks. */
int orc_prove(const orc_statement* st, const uint8_t* transcript_label, size_t tl_len,
              const uint8_t* secrets, const uint8_t* points, const uint8_t entropy32[32],
              uint8_t challenge[32], uint8_t* responses, uint8_t* commitments, uint8_t* blindings_out);
/* The verifier entry points take proofs as BYTES, i.e. where the reference has `bincode::deserialize` (tests/zkp.rs:54, :97)
 * in front of the verifier: a challenge or response that is not a canonical scalar (>= l) fails deserialisation there
 * (proofs.rs:14-32 over dalek's Deserialize [RECALL]) and is a VerificationFailure (1) here, for the proof / the whole batch. */
/* Verifier::verify_compact (verifier.rs:80-120): 0 = Ok, 1 = VerificationFailure */
int orc_verify_compact(const orc_statement* st, const uint8_t* transcript_label, size_t tl_len,
                       const uint8_t* points, const uint8_t challenge[32], const uint8_t* responses);
/* Verifier::verify_batchable (verifier.rs:123-173); weights[n_cons][16] replace the u128 draws */
int orc_verify_batchable(const orc_statement* st, const uint8_t* transcript_label, size_t tl_len,
                         const uint8_t* points, const uint8_t* commitments, const uint8_t* responses,
                         const uint8_t* weights16);
/* BatchVerifier::verify_batchable (batch_verifier.rs:137-235) for N proofs.
 * inst_points[n_inst][N][32] (row = variable, column = proof, as allocate_instance_point receives
 * them), common_points[n_common][32], commitments[N][n_cons][32], responses[N][n_secrets][32],
 * weights16[n_cons][N][16].  If msm_scalars/msm_points are non-NULL they receive the exact
 * (scalar, encoding) sequence chained into optional_multiscalar_mul at :219-228
 * (n_common + (n_inst+n_cons)*N entries) and the MSM itself is skipped (returns 0 / 1 only for the
 * host-side checks).  Returns 0 Ok, 1 VerificationFailure, 2 BatchSizeMismatch. */
int orc_batch_verify(const orc_statement* st, const uint8_t* transcript_label, size_t tl_len, uint32_t N,
                     const uint8_t* inst_points, const uint8_t* common_points, const uint8_t* commitments,
                     const uint8_t* responses, const uint8_t* weights16, uint8_t* msm_scalars, uint8_t* msm_points);

#ifdef __cplusplus
}
#endif
#endif
