/* zkp_toolbox.h -- C ABI of the HOST side of the MI355X zkp engine: Merlin transcripts and the
 * Prover / Verifier / BatchVerifier flows of dalek-cryptography/zkp (src/toolbox/), batched so that the
 * GPU is entered once per phase for a whole batch of proofs instead of once per 2-term MSM.
 *
 * Layering:  this library (libzkp_toolbox.so, C++/g++)  ->  zkp_mi355x.h (libzkp_mi355x.so, HIP).
 * Group arithmetic goes through the zkp_mi355x.h entry points, except for TINY calls and calls without a context (ctx == NULL), which
 * run the same field / group headers on the host (zkp_toolbox_set_host_max_terms below) -- a backend choice by size, never a silent
 * fallback: a call that needs the GPU and cannot have it fails with a negative code.  Host work (STROBE/Keccak transcripts, scalar arithmetic mod l, coefficient build) runs on
 * `n_threads` host threads (0 = all hardware threads).
 *
 * Reference mapping (what each call replaces, for N proofs of ONE statement at a time):
 *   zkp_prove_batch            N x { macros.rs:206-258 build_prover ; prover.rs:76-112 prove_impl }
 *   zkp_verify_compact_batch   N x { macros.rs:280-311 build_verifier ; verifier.rs:80-120 }
 *   zkp_verify_batchable_each  N x { build_verifier ; verifier.rs:123-173 }
 *   zkp_batch_verify           macros.rs:336-370 batch_verify ; batch_verifier.rs:67-235
 * Return codes: 0 = done (per-proof verdicts in `results`); ZKP_TB_VERIFICATION_FAILURE /
 * ZKP_TB_BATCH_SIZE_MISMATCH mirror src/errors.rs:4-11; negative = infrastructure failure (see
 * zkp_mi355x.h) -- never treat a non-zero code as "verified".
 */
#ifndef ZKP_TOOLBOX_H
#define ZKP_TOOLBOX_H

#include <stddef.h>
#include <stdint.h>
#include "zkp_mi355x.h"

#ifdef __cplusplus
extern "C" {
#endif

#define ZKP_TB_OK 0
#define ZKP_TB_VERIFICATION_FAILURE 1   /* ProofError::VerificationFailure (errors.rs:6)  */
#define ZKP_TB_BATCH_SIZE_MISMATCH 2    /* ProofError::BatchSizeMismatch   (errors.rs:9)  */
#define ZKP_TB_BAD_STATEMENT (-10)      /* malformed statement descriptor / NULL argument  */
#define ZKP_TB_INVALID_POINT (-11)      /* prover was handed an encoding that does not decode */
#define ZKP_TB_NO_ENTROPY (-12)         /* entropy / weights16 == NULL and the operating system's getrandom() failed */
#define ZKP_TB_TOO_LONG (-14)           /* a transcript message, output or label longer than UINT32_MAX bytes (merlin asserts) */

/* ---- Merlin transcripts (merlin::Transcript, re-exported by the reference at lib.rs:35) ----------- */
#define ZKP_TRANSCRIPT_BYTES 208        /* opaque, plain-old-data: memcpy = Clone */
void zkp_transcript_init(uint8_t t[ZKP_TRANSCRIPT_BYTES], const uint8_t* label, size_t label_len);
/* Merlin frames every message / output with its length as a u32 and asserts that the length fits (merlin 2.x
 * `encode_usize_as_u32`; the reference's tests/sig_and_vrf_example.rs:224-241 is the `#[ignore]`d > 4 GiB case): a `len`
 * above UINT32_MAX -- or a label longer than that -- is ZKP_TB_TOO_LONG and leaves the transcript untouched, never a
 * silently truncated length prefix.  Returns ZKP_TB_OK otherwise. */
int zkp_transcript_append_message(uint8_t t[ZKP_TRANSCRIPT_BYTES], const char* label, const uint8_t* msg, size_t len);
int zkp_transcript_challenge_bytes(uint8_t t[ZKP_TRANSCRIPT_BYTES], const char* label, uint8_t* out, size_t len);

/* ---- scalars mod l (curve25519_dalek::scalar::Scalar) -------------------------------------------- */
void zkp_scalar_from_wide(uint8_t out[32], const uint8_t in[64]);   /* from_bytes_mod_order_wide */
void zkp_scalar_muladd(uint8_t out[32], const uint8_t a[32], const uint8_t b[32], const uint8_t c[32]);   /* a*b + c */
void zkp_scalar_neg(uint8_t out[32], const uint8_t a[32]);

/* ---- statement descriptor: what define_proof! fixes (macros.rs:124-138, 159-170) ------------------ */
typedef struct zkp_statement zkp_statement;
zkp_statement* zkp_statement_new(const char* proof_label);
void zkp_statement_free(zkp_statement* st);
/* Variables are numbered in ALLOCATION order, which is part of the statement (it fixes the transcript).
 * The macro allocates all secrets, then instance points, then common points (macros.rs:215-242). */
int zkp_statement_add_secret(zkp_statement* st, const char* name);                 /* -> secret index  */
int zkp_statement_add_point(zkp_statement* st, const char* name, int is_common);   /* -> point index   */
/* lhs = sum_i secrets[i] * points[i]   (SchnorrCS::constrain, toolbox/mod.rs:86-98) */
int zkp_statement_constrain(zkp_statement* st, uint32_t lhs_point, uint32_t n_terms, const uint32_t* secrets,
                            const uint32_t* points);
uint32_t zkp_statement_num_secrets(const zkp_statement* st);
uint32_t zkp_statement_num_instance(const zkp_statement* st);
uint32_t zkp_statement_num_common(const zkp_statement* st);
uint32_t zkp_statement_num_constraints(const zkp_statement* st);

/* Data layout shared by all batch calls (N = batch size, m = #secrets, ni / ns = #instance / #common
 * points, nc = #constraints; "rank" = position among the points of the same kind, in allocation order):
 *   transcripts   [N][ZKP_TRANSCRIPT_BYTES]  in/out: advanced exactly as the reference advances them
 *   secrets       [N][m][32]
 *   inst_points   [ni][N][32]   row = variable, column = proof (what allocate_instance_point receives,
 *                               batch_verifier.rs:115-134; Matrix layout util.rs:21-37)
 *   common_points [ns][32]
 *   challenges    [N][32]   responses [N][m][32]   commitments [N][nc][32]
 *   results       [N]       0 = Ok(()), 1 = Err(VerificationFailure)
 * Canonical-scalar rule: the verify calls take proofs as raw 32-byte fields, i.e. where the reference has
 * `bincode::deserialize` in front of its verifiers (tests/zkp.rs:54, :97).  dalek's Deserialize refuses a Scalar whose
 * value is >= l (proofs.rs:14-32), so such a proof never verifies there; here a response (or compact challenge) >= l is
 * Err(VerificationFailure) for that proof -- for the whole batch in zkp_batch_verify*.  (The wire codec below applies the
 * same rule when it parses.)
 */

/* Prove N statements.  entropy = [N][32] bytes replacing the thread_rng contribution of prover.rs:82, or
 * NULL to draw them from the OS.  Produces BOTH proof formats' fields: compact = (challenge, responses)
 * (proofs.rs:15-20), batchable = (commitments, responses) (proofs.rs:27-32). */
int zkp_prove_batch(zkp_ctx* ctx, const zkp_statement* st, uint32_t N, uint8_t* transcripts, const uint8_t* secrets,
                    const uint8_t* inst_points, const uint8_t* common_points, const uint8_t* entropy, int n_threads,
                    uint8_t* challenges, uint8_t* responses, uint8_t* commitments);

int zkp_verify_compact_batch(zkp_ctx* ctx, const zkp_statement* st, uint32_t N, uint8_t* transcripts,
                             const uint8_t* inst_points, const uint8_t* common_points, const uint8_t* challenges,
                             const uint8_t* responses, int n_threads, uint8_t* results);

/* weights16 = [N][nc][16] replacing the u128 draws of verifier.rs:153, or NULL for OS randomness */
int zkp_verify_batchable_each(zkp_ctx* ctx, const zkp_statement* st, uint32_t N, uint8_t* transcripts,
                              const uint8_t* inst_points, const uint8_t* common_points, const uint8_t* commitments,
                              const uint8_t* responses, const uint8_t* weights16, int n_threads, uint8_t* results);

/* One verdict for the whole batch (batch_verifier.rs:230-234): returns ZKP_TB_OK or
 * ZKP_TB_VERIFICATION_FAILURE.  n_transcripts must equal N (else ZKP_TB_BATCH_SIZE_MISMATCH,
 * batch_verifier.rs:72-74).  weights16 = [nc][N][16] replacing batch_verifier.rs:179, or NULL. */
int zkp_batch_verify(zkp_ctx* ctx, const zkp_statement* st, uint32_t N, uint32_t n_transcripts, uint8_t* transcripts,
                     const uint8_t* inst_points, const uint8_t* common_points, const uint8_t* commitments,
                     const uint8_t* responses, const uint8_t* weights16, int n_threads);

/* K batch verifications in one call: the N = n_batches * N_each proofs lie next to each other in every array (layouts as in
 * zkp_batch_verify with that N), batch b = proofs [b * N_each, (b + 1) * N_each); verdicts [n_batches]: ZKP_TB_OK or
 * ZKP_TB_VERIFICATION_FAILURE per batch, each exactly what zkp_batch_verify returns for that batch alone (its own weights,
 * its own sums of the static coefficients, its own MSM: K x batch_verifier.rs:137-235).  On the device this is one
 * transcript launch, one coefficient grid and one segmented Pippenger for all batches (zkp_fused_batch_verify_many);
 * batches whose transcripts do not stand at one STROBE position, or below the fused threshold, are verified one by one.
 * Returns ZKP_TB_OK when every verdict was computed (look at verdicts[]), ZKP_TB_BATCH_SIZE_MISMATCH if n_transcripts !=
 * N, negative on infrastructure failure. */
int zkp_batch_verify_many(zkp_ctx* ctx, const zkp_statement* st, uint32_t n_batches, uint32_t N_each, uint32_t n_transcripts,
                          uint8_t* transcripts, const uint8_t* inst_points, const uint8_t* common_points, const uint8_t* commitments,
                          const uint8_t* responses, const uint8_t* weights16, int n_threads, int* verdicts);

/* zkp_batch_verify with bad-proof localisation.  The reference's batch verifier can only say that
 * SOME proof of the batch is wrong (batch_verifier.rs:233); finding it means verifying one by one.  This call runs the batch
 * check and, only if it fails, re-verifies every proof on its own (verifier.rs:123-173 semantics, from copies of the incoming
 * transcripts) and writes results[N]: 0 = that proof verifies, 1 = it does not.  Returns ZKP_TB_OK (results all 0) or
 * ZKP_TB_VERIFICATION_FAILURE (results say which).  A batch can also fail as a whole without any single proof failing
 * only with negligible probability (the random linear combination), so results then are all 0 and the code is still
 * ZKP_TB_VERIFICATION_FAILURE.  The transcripts are left as the batch check leaves them. */
int zkp_batch_verify_locate(zkp_ctx* ctx, const zkp_statement* st, uint32_t N, uint32_t n_transcripts, uint8_t* transcripts,
                            const uint8_t* inst_points, const uint8_t* common_points, const uint8_t* commitments,
                            const uint8_t* responses, const uint8_t* weights16, int n_threads, uint8_t* results);

/* zkp_batch_verify, additionally returning the coefficient vector the GPU built (zkp_batch_check's debug_scalars:
 * ns + (ni + nc) * N scalars in the operand order of batch_verifier.rs:219-223), so tests can compare it with the
 * host/oracle restatement of batch_verifier.rs:173-206. */
int zkp_batch_verify_coeffs(zkp_ctx* ctx, const zkp_statement* st, uint32_t N, uint32_t n_transcripts, uint8_t* transcripts,
                            const uint8_t* inst_points, const uint8_t* common_points, const uint8_t* commitments,
                            const uint8_t* responses, const uint8_t* weights16, int n_threads, uint8_t* coeffs);

/* ---- pipelines and device groups (round 4) ---------------------------------------------------------------------------
 * A zkp_pipe owns `contexts_per_device` engine contexts on each of the listed GPUs (HIP ordinals; an ordinal may be listed more than once)
 * and offers the calls above in two forms:
 *
 * (a) ASYNCHRONOUS JOBS -- zkp_*_submit puts one call on the next free context and returns a zkp_job; zkp_job_wait returns what the
 *     synchronous call would have returned and fills the outputs.  With >= 3 jobs in flight the host <-> device copies of one job overlap
 *     the kernels of the others: this is the throughput a caller with host buffers gets (bench.py: e2e_host_buffers.pipelined).
 *       - every buffer named by a submit stays valid and untouched until zkp_job_wait returns (which also frees the job);
 *       - buffers in pinned memory (zkp_host_alloc / zkp_host_register of zkp_mi355x.h) go to the DMA engines as they are; others are
 *         staged through pinned rings the pipe owns (one host memcpy each way);
 *       - flags = ZKP_JOB_SHARED_TRANSCRIPT: `transcripts` is ONE blob every proof starts from (`Transcript::new(label)` per proof, as the
 *         reference's callers write); transcripts_out = NULL or [N][208] for the advanced states;
 *       - inst_stride / weights_stride = proofs per row of the caller's [.][stride][32|16] arrays (>= N): a proof range of a larger batch
 *         is passed by pointer offset, nothing is gathered by the caller;
 *       - entropy == NULL / weights16 == NULL: 40 bytes of getrandom() per job key a ChaCha20 stream that is expanded ON THE DEVICE
 *         (the reference's thread_rng(), prover.rs:82 / verifier.rs:153 / batch_verifier.rs:179); getrandom failing = ZKP_TB_NO_ENTROPY;
 *       - all contexts busy = ZKP_TB_PIPE_FULL (wait for the oldest job); jobs complete independently: a failing job -- a rejected batch,
 *         ZKP_ERR_OOM on one context, a device fault -- leaves the verdicts of the others intact, and its own outputs fail closed
 *         (results / verdicts set to "rejected", negative return code from zkp_job_wait);
 *       - batches below the fused threshold or with ragged transcripts run the synchronous call inside submit (same results).
 *     A pipe and its jobs belong to one host thread at a time.
 *     SUBMITTER THREADS (round 5): a pipe over more than one entry of the device list carries out its submits on one host thread per entry --
 *     zkp_*_submit reserves a context, queues the call for that device's thread and returns; the thread stages the buffers (its pinned rings
 *     live on the GPU's NUMA node: zkp_host_alloc_on), enqueues the job, polls the contexts of ITS device so that a finished job's copies
 *     out start at once, and retires finished jobs (staged outputs copied back, verdicts written); zkp_job_wait picks the result up.  A
 *     single caller thread no longer pays 0.1 - 0.4 ms of host work per job per GPU (profiles/r05_pipe_host_scaling.txt: one submitting
 *     thread is host-bound near three GPUs).  Consequences for the caller: errors the submit itself would have returned (a malformed call,
 *     ZKP_ERR_OOM) arrive from zkp_job_wait; the STATEMENT, like every buffer, must stay alive until zkp_job_wait.  zkp_pipe_set_submit_threads
 *     forces the mode (1 = threads also for one device, 0 = the caller's thread does everything, -1 = default) while no job is in flight.
 *
 * (b) SYNCHRONOUS CALLS OVER ALL CONTEXTS -- zkp_pipe_prove_batch, _verify_compact_batch, _verify_batchable_each, _batch_verify[_many],
 *     _batch_verify_locate: same arguments and results as the single-context calls, the N proofs sharded as contiguous ranges
 *     [g N / G, (g + 1) N / G) over the G = min(contexts, N) contexts, one host thread per GPU, common points replicated.  Proving and
 *     per-proof verification give the bytes / verdicts of the single-context call.  zkp_pipe_batch_verify gives ONE verdict: each range
 *     is a batch check of its own (own weights, own static-coefficient sums: batch_verifier.rs:173-206 per range) and the batch verifies
 *     iff every range does -- the AND is taken on the host, no collective, no other process.  zkp_pipe_batch_verify_many
 *     hands whole batches to the contexts.  This is how one process uses the 8 GPUs of a node.
 */
#define ZKP_TB_PIPE_FULL 3              /* every context of the pipe has a job in flight */
typedef struct zkp_pipe zkp_pipe;
typedef struct zkp_job zkp_job;
int zkp_pipe_create(zkp_pipe** out, const int* device_ids, int n_devices, int contexts_per_device);
void zkp_pipe_destroy(zkp_pipe* pipe);                    /* jobs nobody waited for are DISCARDED: kernels waited for, nothing written to caller memory.  A handle
                                                           * that outlives its pipe: one made by a SUBMITTER THREAD may still go to zkp_job_wait (error code, the
                                                           * handle is freed -- otherwise it leaks); any other handle must not be touched after zkp_pipe_destroy */
int zkp_pipe_num_contexts(const zkp_pipe* pipe);
int zkp_pipe_num_devices(const zkp_pipe* pipe);
zkp_ctx* zkp_pipe_context(zkp_pipe* pipe, int i);        /* context i (tuning options, ZKP_OPT_WS_LIMIT_BYTES); do not destroy it */
int zkp_pipe_context_device(const zkp_pipe* pipe, int i);
/* The partition the synchronous calls of (b) use, as plain arithmetic (no pipe, no GPU): n_items proofs -- or whole batches of `unit` proofs each for
 * zkp_pipe_batch_verify_many -- over at most n_contexts contexts, every range at least fused_min_batch proofs wide; writes the G <= n_contexts non-empty
 * ranges [lo[g], hi[g]) (lo / hi: n_contexts words each) and returns G.  Range g runs on context g. */
uint32_t zkp_pipe_shard_plan(uint32_t n_items, uint32_t unit, uint32_t n_contexts, uint32_t fused_min_batch, uint32_t* lo, uint32_t* hi);
int zkp_pipe_jobs_in_flight(const zkp_pipe* pipe);
int zkp_pipe_set_submit_threads(zkp_pipe* pipe, int on);  /* -1 default (threads when the device list has more than one entry), 0 off, 1 on */
const char* zkp_pipe_last_error(const zkp_pipe* pipe);    /* text of the last failure of a pipe call (never NULL) */

int zkp_prove_batch_submit(zkp_pipe* pipe, const zkp_statement* st, uint32_t N, uint32_t flags, const uint8_t* transcripts,
                           const uint8_t* secrets, const uint8_t* inst_points, uint32_t inst_stride, const uint8_t* common_points,
                           const uint8_t* entropy, uint8_t* transcripts_out, uint8_t* challenges, uint8_t* responses,
                           uint8_t* commitments, zkp_job** job);
int zkp_verify_compact_batch_submit(zkp_pipe* pipe, const zkp_statement* st, uint32_t N, uint32_t flags, const uint8_t* transcripts,
                                    const uint8_t* inst_points, uint32_t inst_stride, const uint8_t* common_points,
                                    const uint8_t* challenges, const uint8_t* responses, uint8_t* transcripts_out, uint8_t* results,
                                    zkp_job** job);
int zkp_verify_batchable_each_submit(zkp_pipe* pipe, const zkp_statement* st, uint32_t N, uint32_t flags, const uint8_t* transcripts,
                                     const uint8_t* inst_points, uint32_t inst_stride, const uint8_t* common_points,
                                     const uint8_t* commitments, const uint8_t* responses, const uint8_t* weights16 /*[N][nc][16]*/,
                                     uint8_t* transcripts_out, uint8_t* results, zkp_job** job);
int zkp_batch_verify_many_submit(zkp_pipe* pipe, const zkp_statement* st, uint32_t n_batches, uint32_t N_each, uint32_t flags,
                                 const uint8_t* transcripts, const uint8_t* inst_points, uint32_t inst_stride,
                                 const uint8_t* common_points, const uint8_t* commitments, const uint8_t* responses,
                                 const uint8_t* weights16 /*[nc][weights_stride][16]*/, uint32_t weights_stride, uint8_t* transcripts_out,
                                 int* verdicts /*[n_batches]: ZKP_TB_OK | ZKP_TB_VERIFICATION_FAILURE after zkp_job_wait*/, zkp_job** job);
int zkp_job_context_index(const zkp_job* job);   /* which context of the pipe carries the job (zkp_pipe_context; profiling: zkp_ctx_job_timing) */
int zkp_job_done(const zkp_job* job);    /* 1 = zkp_job_wait would not block */
int zkp_job_wait(zkp_job* job);          /* the synchronous call's return code; frees the job */

int zkp_pipe_prove_batch(zkp_pipe* pipe, const zkp_statement* st, uint32_t N, uint8_t* transcripts, const uint8_t* secrets,
                         const uint8_t* inst_points, const uint8_t* common_points, const uint8_t* entropy, uint8_t* challenges,
                         uint8_t* responses, uint8_t* commitments);
int zkp_pipe_verify_compact_batch(zkp_pipe* pipe, const zkp_statement* st, uint32_t N, uint8_t* transcripts, const uint8_t* inst_points,
                                  const uint8_t* common_points, const uint8_t* challenges, const uint8_t* responses, uint8_t* results);
int zkp_pipe_verify_batchable_each(zkp_pipe* pipe, const zkp_statement* st, uint32_t N, uint8_t* transcripts, const uint8_t* inst_points,
                                   const uint8_t* common_points, const uint8_t* commitments, const uint8_t* responses,
                                   const uint8_t* weights16, uint8_t* results);
int zkp_pipe_batch_verify(zkp_pipe* pipe, const zkp_statement* st, uint32_t N, uint32_t n_transcripts, uint8_t* transcripts,
                          const uint8_t* inst_points, const uint8_t* common_points, const uint8_t* commitments, const uint8_t* responses,
                          const uint8_t* weights16);
int zkp_pipe_batch_verify_many(zkp_pipe* pipe, const zkp_statement* st, uint32_t n_batches, uint32_t N_each, uint32_t n_transcripts,
                               uint8_t* transcripts, const uint8_t* inst_points, const uint8_t* common_points, const uint8_t* commitments,
                               const uint8_t* responses, const uint8_t* weights16, int* verdicts);
int zkp_pipe_batch_verify_locate(zkp_pipe* pipe, const zkp_statement* st, uint32_t N, uint32_t n_transcripts, uint8_t* transcripts,
                                 const uint8_t* inst_points, const uint8_t* common_points, const uint8_t* commitments,
                                 const uint8_t* responses, const uint8_t* weights16, uint8_t* results);

/* Batches of at least this many proofs whose transcripts stand at one STROBE position run entirely on the device
 * (zkp_mi355x.h section 2c: transcripts, scalars and MSMs); smaller or ragged batches hash their transcripts on the
 * host threads and use the GPU for the group arithmetic only.  Both routes produce the same bytes.  Default 32 (the measured crossover for the CMZ statement);
 * 0 = always fused, UINT32_MAX = never. */
void zkp_toolbox_set_fused_min_batch(uint32_t n);
/* Host backend (zkp_amd/csrc/host/host_backend.cpp: the kernels' own point formulas and ristretto codec compiled for the host over a
 * 5 x 51-bit field, host/fe51.h).  Every call above accepts
 * ctx == NULL: the whole call then runs on the host cores -- no GPU needed (BASELINE configs[0]: "DLEQ proof single prove + verify on CPU").
 * With a context, calls whose group arithmetic is at most `n` (scalar, point) terms do the same, because a GPU call is a ~1 ms chain of
 * launches whatever its size and a 2-term multiscalar multiplication is ~0.1 ms on one core.  Default 16 (a single DLEQ proof: 2 terms to
 * prove, 4 to verify); 0 = never with a context.  Same bytes either way (canonical encodings); the prover's multiplications stay constant
 * time on the host (masked table look-ups, no skipped digits). */
void zkp_toolbox_set_host_max_terms(uint32_t n);
uint32_t zkp_toolbox_get_host_max_terms(void);
uint32_t zkp_toolbox_get_fused_min_batch(void);

/* The ChaCha20 block function (RFC 8439 section 2.3; state words 12-13 = counter, 14-15 = nonce) behind the default
 * entropy / weights of the calls above (`entropy == NULL`, `weights16 == NULL`): like the reference's `thread_rng()`, a
 * ChaCha stream keyed from the operating system.  Exposed for the known-answer test. */
void zkp_chacha20_block(const uint8_t key[32], uint64_t counter, uint64_t nonce, uint8_t out[64]);

/* ---- proof wire format (src/proofs.rs:14-32 under `bincode::serialize`, tests/zkp.rs:53-54, :96-97) ---------------
 * The reference derives serde's Serialize / Deserialize and its tests move proofs through bincode 1.x's top-level
 * functions: fixed-width little-endian integers, `u64` sequence lengths, a `Scalar` / `CompressedRistretto` as its 32
 * bytes (serde tuples carry no length):
 *   CompactProof   = challenge[32] | u64 m | m x response[32]                         (40 + 32 m bytes)
 *   BatchableProof = u64 nc | nc x commitment[32] | u64 m | m x response[32]          (16 + 32 (nc + m) bytes)
 * Decoding applies what curve25519-dalek's Deserialize applies: every scalar must be canonical (< l) -- a non-canonical
 * challenge or response is ZKP_TB_BAD_ENCODING -- while a CompressedRistretto is any 32 bytes (validity is decided by
 * decompress() during verification).  A length prefix larger than the bytes that follow is ZKP_TB_BAD_ENCODING.
 * *consumed receives the bytes used; bincode's top-level deserialize ignores trailing bytes, so does this decoder
 * (strict callers compare *consumed with len).  [No golden bytes exist in the reference: its tests only round-trip.]
 * Sizes in elements; the decoders never write more than max_* elements (ZKP_TB_BAD_STATEMENT if the proof holds more). */
#define ZKP_TB_BAD_ENCODING (-13)
size_t zkp_proof_compact_size(uint32_t m);
size_t zkp_proof_batchable_size(uint32_t nc, uint32_t m);
int zkp_proof_compact_encode(const uint8_t challenge[32], const uint8_t* responses /*[m][32]*/, uint32_t m, uint8_t* out,
                             size_t out_len);
int zkp_proof_compact_decode(const uint8_t* in, size_t len, uint8_t challenge[32], uint8_t* responses /*[max_m][32]*/,
                             uint32_t max_m, uint32_t* m, size_t* consumed);
int zkp_proof_batchable_encode(const uint8_t* commitments /*[nc][32]*/, uint32_t nc, const uint8_t* responses /*[m][32]*/,
                               uint32_t m, uint8_t* out, size_t out_len);
int zkp_proof_batchable_decode(const uint8_t* in, size_t len, uint8_t* commitments /*[max_nc][32]*/, uint32_t max_nc,
                               uint32_t* nc, uint8_t* responses /*[max_m][32]*/, uint32_t max_m, uint32_t* m, size_t* consumed);

/* ---- host-only halves, exposed so the host logic can be tested without a GPU ----------------------- */
/* Everything of zkp_batch_verify up to (not including) the MSM: writes the exact operand sequence of
 * batch_verifier.rs:219-228, ns + (ni + nc) * N scalars and encodings.  Returns 0 or the error the
 * reference would have returned before the MSM. */
int zkp_batch_verify_build(const zkp_statement* st, uint32_t N, uint32_t n_transcripts, uint8_t* transcripts,
                           const uint8_t* inst_points, const uint8_t* common_points, const uint8_t* commitments,
                           const uint8_t* responses, const uint8_t* weights16, int n_threads, uint8_t* msm_scalars,
                           uint8_t* msm_points);
/* Prover phase A (prover.rs:78-97 without the MSM): transcripts absorb the public points, blindings are
 * derived; writes blindings [N][m][32] and the CSR multiscalar job for zkp_msm_many:
 *   off [N*nc + 1], scalars [N*T][32], pidx [N*T] into the point table common_points || inst_points. */
int zkp_prove_phase_a(const zkp_statement* st, uint32_t N, uint8_t* transcripts, const uint8_t* secrets,
                      const uint8_t* inst_points, const uint8_t* common_points, const uint8_t* entropy, int n_threads,
                      uint8_t* blindings, uint32_t* off, uint8_t* scalars, uint32_t* pidx);
/* Prover phase B (prover.rs:98-109): absorb the commitments, derive challenges, compute responses. */
int zkp_prove_phase_b(const zkp_statement* st, uint32_t N, uint8_t* transcripts, const uint8_t* secrets,
                      const uint8_t* blindings, const uint8_t* commitments, int n_threads, uint8_t* challenges,
                      uint8_t* responses);
/* total number of constraint terms T = sum |rhs| */
uint32_t zkp_statement_num_terms(const zkp_statement* st);

#ifdef __cplusplus
}
#endif
#endif /* ZKP_TOOLBOX_H */
