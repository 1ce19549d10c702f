/* zkp_mi355x.h -- C ABI of the MI355X (gfx950) ristretto255 MSM / codec engine.
 *
 * This is the drop-in boundary for the hot path of dalek-cryptography/zkp.
 * The reference has no FFI: the path sits behind three curve25519-dalek trait methods plus the
 * point codec.  Each entry point below names the reference call site(s) it replaces; the Rust
 * `-sys` binding a maintainer would add is shown in INTEGRATION.md.
 *
 * Conventions
 *   scalar  : 32 bytes, little-endian integer.  The MSM entry points (1), (2), (2b) accept any 256-bit value and use
 *             it as an integer multiplier (so non-canonical scalars give the same group element dalek's
 *             `Scalar` would after reduction mod l).  The fused VERIFY flows (2c) take proofs, and there the
 *             reference's rule applies: a response >= l (which serde would refuse to deserialise into a
 *             `Scalar`, proofs.rs:14-32) is a verification failure for that proof / batch, and a compact proof's
 *             challenge is compared byte for byte with the recomputed canonical one (verifier.rs:115).
 *   point   : 32 bytes, ristretto255 encoding (RFC 9496).  Decoding rejects exactly what
 *             `CompressedRistretto::decompress` rejects (non-canonical, negative s, non-square,
 *             negative t, y == 0).
 *   buffers : caller owned, not retained after return.  `_dev` variants take DEVICE pointers
 *             (hipMalloc / torch CUDA tensors), enqueue on the context's stream and do not
 *             synchronise; plain variants take HOST pointers, copy in/out and synchronise.
 *   return  : 0 = computed.  < 0 = infrastructure failure (see ZKP_ERR_*): NO output may be
 *             trusted, callers must fail closed (reference analogue: ProofError::VerificationFailure,
 *             src/errors.rs:6).  There is no CPU fallback inside this library.
 *   threads : one context per host thread / per GPU; a context is not re-entrant.
 */
#ifndef ZKP_MI355X_H
#define ZKP_MI355X_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ZKP_OK 0
#define ZKP_ERR_HIP (-1)        /* a HIP runtime call failed (zkp_last_error has the text)        */
#define ZKP_ERR_ARG (-2)        /* NULL pointer / inconsistent sizes / index out of range         */
#define ZKP_ERR_NO_DEVICE (-3)  /* no gfx950 device visible                                       */
#define ZKP_ERR_OOM (-4)        /* device allocation failed                                       */

/* flags for zkp_msm_many */
#define ZKP_VARTIME 0  /* replaces RistrettoPoint::vartime_multiscalar_mul (verifier.rs:97)        */
#define ZKP_CT 1       /* replaces RistrettoPoint::multiscalar_mul (prover.rs:94): the instruction */
                       /* stream and every address are independent of the scalars                  */

typedef struct zkp_ctx zkp_ctx;

/* One context per GPU (one process per GPU in multi-GPU jobs).  device_id is the HIP ordinal. */
int zkp_ctx_create(zkp_ctx** out, int device_id);
void zkp_ctx_destroy(zkp_ctx* ctx);
/* Use an externally owned hipStream_t (e.g. torch.cuda.current_stream().cuda_stream); NULL = the
 * context's own stream. */
int zkp_ctx_set_stream(zkp_ctx* ctx, void* hip_stream);
int zkp_ctx_synchronize(zkp_ctx* ctx);
/* Text of the last error on this thread (never NULL). */
const char* zkp_last_error(void);
/* Library / build identification, e.g. "zkp-mi355x 0.1 gfx950". */
const char* zkp_version(void);

/* Tuning knobs (results never depend on them).
 *   ZKP_OPT_BATCH_ENCODE_MIN: calls of zkp_msm_many / the fused flows with at least this many outputs encode them as
 *     2 * H with H = sum (s_i / 2) P_i, which needs one field inversion per 65,536 outputs instead of an inverse square
 *     root per output (the identity behind curve25519-dalek's double_and_compress_batch): 10 x fewer instructions in the
 *     encoding step, three short kernels instead of one (default 65536: below that the extra latency outweighs the
 *     instructions saved; 0 = always, UINT64_MAX = never).  Unless the option is set, the asynchronous _dev
 *     entry points, whose callers keep many calls in flight, use 2048.
 *   ZKP_OPT_COMB_TEETH: comb-table shape zkp_msm_many_dev uses for points that two or more terms of a call multiply
 *     (4 or 16; default 4).  16 suits calls whose shared points carry about six or more terms each (16 instead of 64
 *     doublings per term, a 4 x larger table per point).  Every other entry point derives the shape from its inputs.
 *   ZKP_OPT_CT_SINGLE_USE_TABLES: how ZKP_CT calls serve a point that a single term multiplies.  1 = a comb table like the shared
 *     points (the 256 doublings run next to the other tables' chains: shortest call), 0 = a constant-time radix-16 ladder over
 *     the point's own eight multiples (26 % fewer instructions for that term, but a 321-operation dependent chain inside the
 *     term kernel), UINT64_MAX = default: the ladder in asynchronous _dev calls of 250,000 terms or more (a call that fills the
 *     chip on its own), tables otherwise.
 *   ZKP_OPT_DEV_OVERLAP: 1 = zkp_fused_prove_dev / _verify_compact_dev / _verify_batchable_dev run the half of their work that does not depend on the
 *     transcripts (decoding, classification, comb tables) on a second stream of the context, as the synchronous entry points
 *     always do: a shorter call, more cross-stream dependencies.  Default 0.  (Measured again in round 4 on a LONE call chain: 2.87 -> 3.17 ms per
 *     prove call of 20,480 proofs -- no gain there either.)  In a process that owns one hardware queue (GPU_MAX_HW_QUEUES=1) a capture records the
 *     flow without the fork: ROCm 7.2.0 crashes in hipGraphLaunch on a forked graph under that setting.
 *     2 (round 6) = the whole LATENCY SCHEDULE of the synchronous entry points on device buffers, for a caller that keeps ONE call in flight: the second
 *     stream as above -- for zkp_fused_batch_verify[_many]_dev too: the decompressions of the batch MSM (batch_verifier.rs:226) run next to the transcript
 *     chain --, comb tables built by four lanes per point, tables for single-use points, and transcript-chain wavefronts that own their SIMD
 *     (k_transcript_chain<true>).  One batch of 4096 CMZ proofs proven and batch-verified per call: 2.53 M proofs/s against 2.30 M on the default schedule
 *     and 2.05 M in round 5 (bench.py: lone_call).
 *   ZKP_OPT_GROUPED_COMB: 1 = constant-time calls list the terms of every point with 10 or more uses next to each other; a wavefront
 *     then holds the rows of the (at most 8) 16-teeth comb tables its 62 terms need and every lane takes the entry its digit names over
 *     the lane crossbar (ZKP_OPT_CT_LOOKUP: no masked scan, 20 % fewer instructions per addition, a row fetched once per wavefront instead
 *     of once per lane); 0 = every comb term scans its rows with masks; UINT64_MAX = default: 1 for calls of 400,000 terms or more
 *     (asynchronous _dev calls: 250,000).
 *   ZKP_OPT_TABLES_LANE: comb tables are built by one lane per point (1: half the instructions) or by four lanes per point
 *     (0: a quarter of the latency); UINT64_MAX = default: 1 in the asynchronous _dev entry points, 0 in the synchronous ones.
 *   ZKP_OPT_FUSE_TABLES_TRANSCRIPT: 1 = zkp_fused_prove_dev / _verify_compact_dev run their first transcript program in the same
 *     launch as the comb-table construction (both are long dependent chains on few wavefronts, independent of each other): a
 *     lone call of 4096 CMZ proofs takes 1.37 instead of 1.92 ms (one such call after the other, prove + batch verify: 1.36 -> 1.68 M
 *     proofs/s; four call chains in flight: 3.27 -> 3.70 M), while callers that keep 25 such calls in flight lose 9 % (the transcript
 *     wavefronts then carry the table builder's 212 registers instead of their own 118) -- they are better served by one wide call
 *     (zkp_fused_prove_dev over K * 4096 proofs + zkp_fused_batch_verify_many_dev: 6.1 - 6.9 M proofs/s).  0 = never; UINT64_MAX =
 *     default: in calls of fewer than 65,536 proofs (from there on separate launches are 1.5 - 3 % faster).
 *   ZKP_OPT_TRANSCRIPT_LANES: lanes per proof in the Merlin transcript kernel of the fused flows.  2 = a lane pair per proof
 *     (each lane holds one 32-bit half of every STROBE word: half the latency), 1 = one lane per proof (23 % fewer
 *     instructions per Keccak-f), UINT64_MAX = default: 1 in asynchronous _dev calls of 65,536 proofs or more, 2 otherwise.
 *   ZKP_OPT_CT_LOOKUP: how a ZKP_CT call picks the table entry a secret digit names.
 *     0 / UINT64_MAX (default since round 5) = LANE CROSSBAR: constant time by construction.  A wavefront holds the row in registers, one
 *       entry per lane (fixed-base rows: 32 entries of a 6-bit window; grouped comb rows: 8 tables x 8 entries), and every lane fetches its
 *       entry with ds_bpermute_b32 -- the digit selects a source LANE, never a memory address, and the layouts keep the sources of every
 *       instruction inside one 32-lane half, where the crossbar has no bank conflicts whatever the digits are (hot_tables.h, comb_tables.h;
 *       tools/microbench/bpermute_rate.hip).  Rows of 8 entries that belong to ONE lane (comb tables of points with fewer than 11 uses,
 *       ladder tables) are scanned completely and the entry kept with v_cndmask, as curve25519-dalek's LookupTable::select does.  This is
 *       what `RistrettoPoint::multiscalar_mul` promises at prover.rs:94, at the speed of the look-up of value 2 (profiles/r05_ab_experiments.txt).
 *     1 = MASKED SCANS everywhere (fixed-base rows: 32 entries x 7 LDS reads + 864 selects per addition; no grouped comb walk): the
 *       round-3 "safe mode"; same bytes, -25 % throughput.  Kept as the reference point of the other two.
 *     2 = the default of rounds 2 - 4: rows replicated in LDS and read at an index derived from the digit, from banks that no other lane of
 *       the ds_read_b128 service group touches -- constant time under the LDS bank / service-group model of the hardware guide (checked
 *       with PMC counters and per-wavefront cycle counts), not by construction.
 *     Same bytes out for every value (tests/test_gpu_device_entry.py); 1 and 2 exist for the 6-bit fixed-base window only.
 *   ZKP_OPT_CT_MASKED_SCANS: the name ZKP_OPT_CT_LOOKUP had in rounds 3 - 4 (same id; value 1 still selects the masked scans, value 0 is now the crossbar).
 *   ZKP_OPT_EACH_STRAUS: how zkp_fused_verify_batchable[_dev] computes a proof's MSM over its points and commitments (verifier.rs:162-166).
 *     UINT64_MAX = default: one Straus walk per proof -- 256 shared doublings and one table addition per operand and window; below
 *     65,536 proofs split over 32 lanes per proof by windows (each lane: two windows of every operand; one quad of lanes per proof then
 *     joins the partial sums), from 65,536 proofs on one lane per proof.  0x200 + P (P = 1, 2, 4 .. 64) = P window parts per proof;
 *     1 .. 8 = split by operands over that many lanes per proof (every lane runs the 256 doublings); 0 = the round-2 schedule
 *     (every single-use point on a ladder of its own: 256 doublings per operand).
 *   ZKP_OPT_LADDER_INTERLEAVE: 1 = the term kernel's ladder blocks (single-use points: CMZ's Q) are spread over the first half of its grid
 *     instead of all starting first (fewer per-lane ladder tables in flight together: -20 % HBM fetch in that kernel); 0 = all first;
 *     UINT64_MAX = default: spread when the launch has 256 or more ladder blocks (65,536 single-use points), where it also is ~1 % faster --
 *     in a lone smaller launch the later start of the last ladder block lengthens the kernel (profiles/r03_ab_experiments.txt, block l).
 *   ZKP_OPT_JOB_DEFER_D2H: when the device -> host copies of a host-buffer job (section 2d) are issued.  1 (default) = by zkp_ctx_job_poll /
 *     zkp_ctx_job_wait once the job's kernels are finished; 0 = queued behind the kernels at submit.  The copy engines serve ONE queue in order:
 *     a copy that waits in it for its job's kernels (milliseconds) holds up the copies IN of every job submitted after it -- measured with six
 *     contexts, a job's inputs take 3.7 ms to arrive instead of 1.4, and the pipelined rate through pinned buffers is 4.5 instead of 5.4 M proofs/s
 *     (profiles/r04_ab_experiments.txt, block e).
 *   ZKP_OPT_SYNC_SCHEDULE: which schedule the SYNCHRONOUS host-pointer entry points (section 2c) run.  0 / UINT64_MAX (default) = the low-latency one (second
 *     stream for the point phase, comb tables built by four lanes per point, single-use points on tables): one call at a time per process, the caller sees
 *     the call's duration.  1 = the throughput schedule of the asynchronous jobs (section 2d): for callers that issue synchronous calls from several host
 *     threads at once, one context per thread -- every call is a little longer, the chip does less work per proof (profiles/r04_ab_experiments.txt, block r).
 *   ZKP_OPT_WS_LIMIT_BYTES: the largest device workspace this context may allocate (it grows with the largest call it has served: ~85 KB per CMZ
 *     proof of a prove call, 9 KB of it the transcript images).  A call that would need more returns ZKP_ERR_OOM instead of allocating -- the way to keep several contexts of a
 *     zkp_pipe (zkp_toolbox.h) inside one GPU's memory.  0 / UINT64_MAX = no cap (default).
 *   ZKP_OPT_TRANSCRIPT_STEPS: how the lane-pair Merlin transcripts of the fused flows run (round 6).  1 (default) = assemble + chain: one wide kernel builds every
 *     proof's per-block absorb image (all loads independent; identity checks too), then a chain kernel does nothing but state = (state & KEEP) ^ image and
 *     Keccak-f[1600] per block, state in registers, the next image prefetched under the permutation -- 5.6 us per permutation on a lone wavefront instead of the
 *     9 - 14 us of the word-operation interpreter, whose every operation is a dependent global load (profiles/r06_keccak_microbench.txt, r06_ab_experiments.txt).
 *     0 = the interpreter of rounds 2 - 5 (same bytes).  Calls wide enough for one transcript lane per proof (ZKP_OPT_TRANSCRIPT_LANES) keep the interpreter.
 *   ZKP_OPT_COMB_SPLIT: constant-time calls, the terms of a per-call point with 2 - 9 uses (or one use and a table: CMZ's Q) scan their 8-entry rows completely; one
 *     lane per term is a chain of 64 scanned additions.  1 = a QUAD of lanes per such term, one window each (16 scanned additions per lane, then a 12-doubling
 *     Horner inside the quad: a third of the chain for 1.45 x the instructions), together with the grouped walk for the points that qualify.  Default
 *     (UINT64_MAX): calls of 8,192 .. 400,000 terms on the latency schedule (synchronous entry points, ZKP_OPT_DEV_OVERLAP = 2); 0 = never.  Same bytes.
 *   ZKP_OPT_JOINT_LADDER: variable-time statement flows (zkp_fused_verify_compact*: commitment = sum s_i P_i - c LHS per constraint, verifier.rs:95-106).  The
 *     left-hand side is multiplied once per proof -- a chain of 252 doublings of its own.  1 (default): that chain also carries ONE other per-proof term of the
 *     constraint (Straus interleaving: eight multiples + 64 additions, no doublings, no comb table) -- CMZ: P joins C_i's chain in ten constraints and needs no
 *     table, Q joins V's: 11 chains instead of 12 and no 16-teeth table per proof.  A per-proof point ALL of whose terms ride (P) gets a table of its multiples
 *     1 .. 128 where its comb table was: its riders add one signed 8-bit digit per byte (32 additions instead of 64) and build no multiples of their own;
 *     (a table is a chain of 127 additions: built on the throughput schedule and in calls of >= 16,384 proofs, not in a lone synchronous call of fewer);
 *     2 = pairs without these tables.  0 = every term on its own (rounds 2 - 5).  Same bytes.
 *   This enum is the whole option surface of the shipped library; measurement hooks live in test-hook builds only (end of file). */
enum { ZKP_OPT_BATCH_ENCODE_MIN = 1, ZKP_OPT_COMB_TEETH = 2, ZKP_OPT_CT_SINGLE_USE_TABLES = 3, ZKP_OPT_TRANSCRIPT_LANES = 4, ZKP_OPT_DEV_OVERLAP = 5, ZKP_OPT_GROUPED_COMB = 6, ZKP_OPT_TABLES_LANE = 7,
       ZKP_OPT_FUSE_TABLES_TRANSCRIPT = 8, ZKP_OPT_CT_LOOKUP = 9, ZKP_OPT_CT_MASKED_SCANS = 9 /* round-3 name: value 1 still selects the scans */, ZKP_OPT_EACH_STRAUS = 10, ZKP_OPT_LADDER_INTERLEAVE = 11, ZKP_OPT_WS_LIMIT_BYTES = 12, ZKP_OPT_JOB_DEFER_D2H = 13, ZKP_OPT_SYNC_SCHEDULE = 14, ZKP_OPT_TRANSCRIPT_STEPS = 15, ZKP_OPT_COMB_SPLIT = 16, ZKP_OPT_JOINT_LADDER = 17 };
int zkp_ctx_set_option(zkp_ctx* ctx, int option, uint64_t value);

/* HIP graphs.  A batch of proofs is a chain of ~35 short kernels (75 in round 1); enqueueing them one by one costs the host ~0.1 ms per
 * batch (measured: 0.14-0.17 ms against 0.9 ms of GPU time per pipelined batch; 0.025 ms as a graph), which matters for
 * short runs and for hosts busier than a benchmark loop.  Everything enqueued on the context's stream between _begin and
 * _end -- *_dev calls of this library and
 * the caller's own asynchronous copies on that stream -- is recorded instead of executed; zkp_graph_launch then replays
 * the recording with ONE host call.  Replays read and write the same device addresses, so the buffers must stay alive
 * and are reused by every replay.  Preconditions: the same calls ran once before on this context with the same shapes
 * (statement plans compiled, workspace sized, fixed points registered) -- otherwise ZKP_ERR_ARG -- and profiling is off.
 * Lifetime rules.  A graph belongs to the context it was captured on: the recorded kernels hold raw addresses inside that
 * context's workspace and statement plans.  Each graph is stamped with the context's workspace / plan generation;
 * zkp_graph_launch returns ZKP_ERR_ARG ("stale graph") -- it never runs the recording -- when, after the capture, a larger
 * call made the workspace grow (it is reallocated), the plan cache was flushed (more than 64 distinct (flow, statement, N)
 * plans on one context), the context was destroyed, or when the graph is offered to another context.  Capture again then.
 * If a call between _begin and _end fails, the capture is poisoned: call zkp_ctx_capture_abort (ends and discards it; the
 * context is usable again) -- zkp_ctx_capture_end on a poisoned capture returns an error and also leaves capture mode. */
typedef struct zkp_graph zkp_graph;
int zkp_ctx_capture_begin(zkp_ctx* ctx);
int zkp_ctx_capture_end(zkp_ctx* ctx, zkp_graph** out);
int zkp_ctx_capture_abort(zkp_ctx* ctx);                /* no capture in progress: no-op */
int zkp_graph_launch(zkp_graph* graph, zkp_ctx* ctx);   /* asynchronous, on the context's current stream */
void zkp_graph_destroy(zkp_graph* graph);

/* Performance hint, never changes a result: declare points that very many terms of later zkp_msm_many calls
 * will reference -- in the reference's vocabulary the statement's COMMON variables (define_proof!,
 * macros.rs:84,236-242) / BatchVerifier's static points (batch_verifier.rs:100-112), e.g. the issuer
 * parameters X_1..X_10, A of the CMZ'13 statement.  The engine builds fixed-base window tables for them once
 * (64 slots, least-recently-used replacement) and then serves every term on such a point with 43 mixed
 * additions and no doublings (154 KB of table per point).  encodings = HOST pointer [n][32]; synchronous. */
int zkp_ctx_prepare_fixed_points(zkp_ctx* ctx, uint32_t n, const uint8_t* encodings);

/* (1) Many small multiscalar multiplications in CSR form, fused with compression.
 *     Replaces, for a whole batch of proofs at once:
 *       prover.rs:94-97   RistrettoPoint::multiscalar_mul(..)            (flags = ZKP_CT)
 *       verifier.rs:97-106 RistrettoPoint::vartime_multiscalar_mul(..)   (flags = ZKP_VARTIME)
 *     and the `point.compress()` of toolbox/mod.rs:204 applied to each result.
 *       out[i]    = encode( sum_{t in [off[i], off[i+1])} scalars[t] * decode(points[pidx[t]]) )
 *       status[i] = 0 ok | 1 some referenced point failed to decode (Rust: decompress() == None;
 *                   out[i] is then 32 zero bytes and must not be used)
 *     off has n_msm+1 entries, off[0] = 0, non-decreasing; T = off[n_msm] terms. An empty range
 *     yields the identity encoding (32 zero bytes), like an empty dalek MSM. */
int zkp_msm_many(zkp_ctx* ctx, uint32_t n_msm, const uint32_t* off, const uint8_t* scalars /*[T][32]*/,
                 const uint32_t* pidx /*[T]*/, const uint8_t* points /*[n_points][32]*/, uint32_t n_points,
                 int flags, uint8_t* out /*[n_msm][32]*/, uint8_t* status /*[n_msm]*/);
int zkp_msm_many_dev(zkp_ctx* ctx, uint32_t n_msm, const uint32_t* d_off, const uint8_t* d_scalars,
                     const uint32_t* d_pidx, const uint8_t* d_points, uint32_t n_points, uint32_t n_terms,
                     int flags, uint8_t* d_out, uint8_t* d_status);

/* (2) One large multiscalar multiplication with decode-or-None.
 *     Replaces RistrettoPoint::optional_multiscalar_mul(scalars, points.map(decompress)) at
 *       verifier.rs:162-166 and batch_verifier.rs:219-228 (the batch-verification random linear
 *       combination of size num_s + (num_i + num_c) * N).
 *     *status = 0: Some(P), out_point = encode(P) (callers test for 32 zero bytes = identity,
 *                  verifier.rs:168 / batch_verifier.rs:230)
 *             = 1: None (some point failed to decode); out_point is zeroed and meaningless. */
int zkp_msm_optional(zkp_ctx* ctx, uint64_t n, const uint8_t* scalars /*[n][32]*/,
                     const uint8_t* points /*[n][32]*/, uint8_t out_point[32], int* status);
int zkp_msm_optional_dev(zkp_ctx* ctx, uint64_t n, const uint8_t* d_scalars, const uint8_t* d_points,
                         uint8_t* d_out_point /*[32]*/, uint32_t* d_status /*[1]*/);

/* (2b) Batch verification with the coefficient build ON THE GPU.
 *     Replaces batch_verifier.rs:173-234 in one call: the random-linear-combination coefficients
 *       static_coeffs[s]        = sum_j sum_{constraint i touching s} r_ij * (minus_c_j | resp_j[sc])      (:185-204)
 *       Matrix[(var, j)]        likewise for instance variables, and  Matrix[(n_i + i, j)] = -r_ij        (:183)
 *     are computed with on-device arithmetic mod l, laid out as the reference chains them (:219-223: static
 *     coefficients, then the matrix row-major), and fed to the same MSM as zkp_msm_optional together with
 *     static_points || instance_points || commitment rows.  Point ids: 0 .. n_static-1 = static points,
 *     n_static .. n_static+n_instance-1 = instance points.  All pointers are HOST pointers.
 *     minus_c [N][32] = the negated per-proof challenges (:163-167, computed by the caller's transcripts);
 *     responses [N][n_secrets][32]; weights16 [n_constraints][N][16] = the u128 random factors (:179), little endian;
 *     instance_points [n_instance][N][32]; commitments [N][n_constraints][32].
 *     *status as in zkp_msm_optional.  debug_scalars (NULL or [n_static + (n_instance+n_constraints)*N][32]) receives
 *     the coefficient vector for tests. */
typedef struct {
  uint32_t n_secrets, n_static, n_instance, n_constraints;
  const uint32_t* cons_lhs;   /* [n_constraints] point id of the left-hand side          */
  const uint32_t* cons_off;   /* [n_constraints + 1] offsets into cons_sc / cons_pt      */
  const uint32_t* cons_sc;    /* secret index of each right-hand-side term               */
  const uint32_t* cons_pt;    /* point id of each right-hand-side term                   */
} zkp_batch_statement;
int zkp_batch_check(zkp_ctx* ctx, const zkp_batch_statement* st, uint32_t N, const uint8_t* minus_c,
                    const uint8_t* responses, const uint8_t* weights16, const uint8_t* static_points,
                    const uint8_t* instance_points, const uint8_t* commitments, uint8_t out_point[32], int* status,
                    uint8_t* debug_scalars);

/* (2c) Fused statement flows: a whole batch of proofs of ONE statement handled
 *     on the device -- Merlin/STROBE transcripts (src/toolbox/mod.rs:165-228 over merlin 2.x), blinding factors
 *     (prover.rs:78-89), scalar arithmetic mod l (prover.rs:107-109, batch_verifier.rs:173-206), MSM operand assembly
 *     and the MSMs themselves.  The host uploads inputs and downloads proofs or verdicts; nothing else crosses PCIe.
 *     `transcripts` = [N][208] transcript blobs (layout of zkp_toolbox.h: 200 bytes of STROBE state, then pos,
 *     pos_begin, cur_flags), updated in place exactly as the reference updates its `&mut Transcript`s.  All N blobs
 *     must stand at the same STROBE position (same pos / pos_begin / cur_flags bytes), which holds whenever they were
 *     produced by the same sequence of appends with equal lengths; otherwise ZKP_ERR_ARG (callers then use the
 *     per-proof host transcripts of zkp_toolbox.h).  Layouts: secrets / responses [N][n_secrets][32];
 *     inst [n_instance][N][32]; common [n_static][32]; commitments [N][n_constraints][32]; entropy [N][32] (the 32
 *     bytes the external RNG contributes to each proof's TranscriptRng, prover.rs:82). */
typedef struct {
  zkp_batch_statement shape;
  const char* label;                  /* the proof label of `domain_sep` (mod.rs:166-169)                                  */
  const char* const* secret_labels;   /* [n_secrets]                                                                      */
  const char* const* point_labels;    /* [n_static + n_instance], indexed by point id                                     */
  const uint32_t* alloc_order;        /* [n_static + n_instance] point ids in allocation (= transcript) order             */
  const uint32_t* alloc_seq;          /* NULL: every secret is allocated before the first point (define_proof!'s order,   */
                                      /* macros.rs:215-242).  Otherwise [n_secrets + n_static + n_instance] entries, the  */
                                      /* caller's allocate_scalar / allocate_point calls in order: 0x80000000 | secret     */
                                      /* index, or a point id (the points must come in alloc_order's order).              */
} zkp_fused_statement;
/* N x { build_prover (macros.rs:206-258) ; Prover::prove_impl (prover.rs:76-112) }.  *invalid_point = 1 if some input
 * encoding did not decode (the reference prover holds decoded points, so this is a caller bug there). */
int zkp_fused_prove(zkp_ctx* ctx, const zkp_fused_statement* st, uint32_t N, uint8_t* transcripts, const uint8_t* secrets,
                    const uint8_t* inst, const uint8_t* common, const uint8_t* entropy, uint8_t* challenges,
                    uint8_t* responses, uint8_t* commitments, int* invalid_point);
/* The same with the prover's entropy drawn ON THE DEVICE (round 6): seed = 40 bytes the caller took from the operating system; proof j's 32 bytes of
 * `thread_rng()` (prover.rs:82) are bytes [32 j, 32 j + 32) of the ChaCha20 stream keyed with seed[0..32), nonce seed[32..40) -- what the asynchronous jobs of
 * section 2d do.  Drawing 32 N bytes on the host costs a synchronous call of 4096 proofs ~0.1 ms, the batch verifier's 16 N n_constraints bytes ~0.2 ms. */
int zkp_fused_prove_seeded(zkp_ctx* ctx, const zkp_fused_statement* st, uint32_t N, uint8_t* transcripts, const uint8_t* secrets,
                           const uint8_t* inst, const uint8_t* common, const uint8_t seed[40], uint8_t* challenges,
                           uint8_t* responses, uint8_t* commitments, int* invalid_point);
/* N x { build_verifier (macros.rs:280-311) ; Verifier::verify_compact (verifier.rs:80-120) }.  results[j]: 0 = accepted. */
int zkp_fused_verify_compact(zkp_ctx* ctx, const zkp_fused_statement* st, uint32_t N, uint8_t* transcripts,
                             const uint8_t* inst, const uint8_t* common, const uint8_t* challenges,
                             const uint8_t* responses, uint8_t* results);
/* batch_verify (macros.rs:336-370 ; batch_verifier.rs:67-235) without the batch-shape checks, which stay with the
 * caller.  weights16 [n_constraints][N][16].  *verdict: 0 = the batch verifies.  debug_scalars as in zkp_batch_check. */
int zkp_fused_batch_verify(zkp_ctx* ctx, const zkp_fused_statement* st, uint32_t N, uint8_t* transcripts,
                           const uint8_t* inst, const uint8_t* common, const uint8_t* commitments,
                           const uint8_t* responses, const uint8_t* weights16, int* verdict, uint8_t* debug_scalars);

/* K independent batch verifications in ONE pass (K x batch_verifier.rs:67-235), for servers that collect more proofs than
 * they want to tie to one verdict: the K * N_each proofs lie next to each other in every array, batch b = proofs
 * [b * N_each, (b + 1) * N_each).  Each batch gets what BatchVerifier::verify_batchable gives it: its own weights, its own
 * sums of the static-point coefficients (:187, :198), its own MSM of n_static + (n_instance + n_constraints) * N_each terms
 * (:219-228) and its own verdict -- a bad proof, a rejected point or an undecodable point in batch b changes verdict b only.
 * One transcript launch, one coefficient grid and one "segmented" Pippenger (sort key = (batch, window, digit)) serve all K
 * batches, so the narrow tails of a single batch check (bucket tree, Horner) are K times wider.
 * Layouts as in zkp_fused_batch_verify with N = K * N_each: transcripts [N][208], inst [n_instance][N][32], commitments
 * [N][n_constraints][32], responses [N][n_secrets][32], weights16 [n_constraints][N][16].  verdicts [K]: 0 = batch b verifies.
 * debug_scalars (NULL or [K * n_static + (n_instance + n_constraints) * N][32]): static coefficients batch by batch, then the
 * coefficient matrix row-major over all N proofs.  K = 1 is zkp_fused_batch_verify. */
int zkp_fused_batch_verify_many(zkp_ctx* ctx, const zkp_fused_statement* st, uint32_t n_batches, uint32_t N_each, uint8_t* transcripts,
                                const uint8_t* inst, const uint8_t* common, const uint8_t* commitments, const uint8_t* responses,
                                const uint8_t* weights16, int* verdicts, uint8_t* debug_scalars);
/* The same with the weights of batch_verifier.rs:179 drawn on the device from the ChaCha20 stream of a 40-byte seed (see zkp_fused_prove_seeded): weight
 * (constraint k, proof j) = bytes [16 (k N + j), + 16) of the stream. */
int zkp_fused_batch_verify_many_seeded(zkp_ctx* ctx, const zkp_fused_statement* st, uint32_t n_batches, uint32_t N_each, uint8_t* transcripts,
                                       const uint8_t* inst, const uint8_t* common, const uint8_t* commitments, const uint8_t* responses,
                                       const uint8_t seed[40], int* verdicts);

/* N x { build_verifier ; Verifier::verify_batchable (verifier.rs:123-173) }: one MSM of (points + commitments) terms per
 * proof, folded with the 128-bit weights16 [N][n_constraints][16] (verifier.rs:153).  results[j]: 0 = accepted.  This is
 * the per-proof check that localises a bad proof after a failed batch verification. */
int zkp_fused_verify_batchable(zkp_ctx* ctx, const zkp_fused_statement* st, uint32_t N, uint8_t* transcripts,
                               const uint8_t* inst, const uint8_t* common, const uint8_t* commitments,
                               const uint8_t* responses, const uint8_t* weights16, uint8_t* results);
/* The same, additionally returning the per-proof coefficient vectors the device folded (verifier.rs:144-160): debug_scalars
 * = NULL or [N][n_static + n_instance + n_constraints][32], per proof in the operand order of verifier.rs:162-166 (the
 * points by point id, then the commitments), so tests can compare them with a restatement of the fold. */
int zkp_fused_verify_batchable_coeffs(zkp_ctx* ctx, const zkp_fused_statement* st, uint32_t N, uint8_t* transcripts,
                                      const uint8_t* inst, const uint8_t* common, const uint8_t* commitments,
                                      const uint8_t* responses, const uint8_t* weights16, uint8_t* results,
                                      uint8_t* debug_scalars);

/*     Device-resident variants: every buffer is a device pointer (16-byte aligned), nothing is copied and the call
 *     returns as soon as the work is queued on the context's stream (zkp_ctx_synchronize to wait).  strobe_pos =
 *     pos | pos_begin << 8 | cur_flags << 16, the three trailing bytes every one of the N transcript blobs holds.
 *     d_table = common points followed by the instance rows: [n_static + n_instance * N][32].
 *     zkp_fused_prove_dev: d_status [N * n_constraints] bytes, non-zero where an input point failed to decode.
 *     zkp_fused_batch_verify_dev: d_points [n_static + (n_instance + n_constraints) * N][32] with the static points
 *     and instance rows filled in (the commitment rows are written by the call); d_status [2] words: decode failure
 *     in the MSM | a point or commitment rejected by the transcript protocol (identity encoding); the batch verifies
 *     iff both are 0 and d_out_point holds 32 zero bytes (= the canonical encoding of the identity; for any other sum the 32 bytes are a
 *     non-zero marker, not its encoding: batch_verifier.rs:230-234 asks is_identity() and nothing else, and the test -- X = 0 or Y = 0 --
 *     needs no inverse square root at the end of the MSM's 253-doubling chain). */
int zkp_fused_prove_dev(zkp_ctx* ctx, const zkp_fused_statement* st, uint32_t N, uint32_t strobe_pos, uint8_t* d_transcripts,
                        const uint8_t* d_secrets, const uint8_t* d_table, const uint8_t* d_entropy, uint8_t* d_challenges,
                        uint8_t* d_responses, uint8_t* d_commitments, uint8_t* d_status);
int zkp_fused_verify_compact_dev(zkp_ctx* ctx, const zkp_fused_statement* st, uint32_t N, uint32_t strobe_pos,
                                 uint8_t* d_transcripts, const uint8_t* d_table, const uint8_t* d_challenges,
                                 const uint8_t* d_responses, uint8_t* d_results);
int zkp_fused_batch_verify_dev(zkp_ctx* ctx, const zkp_fused_statement* st, uint32_t N, uint32_t strobe_pos,
                               uint8_t* d_transcripts, uint8_t* d_points, const uint8_t* d_commitments,
                               const uint8_t* d_responses, const uint8_t* d_weights16, uint8_t* d_out_point,
                               uint32_t* d_status);
/*     zkp_fused_verify_batchable_dev (verifier.rs:123-173 for N proofs, one verdict per proof): d_table = common points, instance rows and
 *     then the proofs' commitments, [n_static + n_instance * N][32] || [N][n_constraints][32]; d_weights16 [N][n_constraints][16] (the
 *     factors verifier.rs:153 draws per proof); d_results [N] bytes, 0 = verified. */
int zkp_fused_verify_batchable_dev(zkp_ctx* ctx, const zkp_fused_statement* st, uint32_t N, uint32_t strobe_pos,
                                   uint8_t* d_transcripts, const uint8_t* d_table, const uint8_t* d_responses,
                                   const uint8_t* d_weights16, uint8_t* d_results);

/* zkp_fused_batch_verify_many on device buffers: d_points [n_static + (n_instance + n_constraints) * N][32] (N = n_batches *
 * N_each; static points and instance rows filled in, commitment rows written by the call), d_out_points [n_batches][32],
 * d_status [n_batches][2] words (decode failure in batch b's MSM | a point, commitment or response of batch b rejected):
 * batch b verifies iff both are 0 and d_out_points[b] is 32 zero bytes (otherwise a non-zero marker, as above). */
int zkp_fused_batch_verify_many_dev(zkp_ctx* ctx, const zkp_fused_statement* st, uint32_t n_batches, uint32_t N_each, uint32_t strobe_pos,
                                    uint8_t* d_transcripts, uint8_t* d_points, const uint8_t* d_commitments, const uint8_t* d_responses,
                                    const uint8_t* d_weights16, uint8_t* d_out_points, uint32_t* d_status);

/* (2d) The fused flows on HOST buffers as asynchronous jobs (round 4).  One job per context at a time:
 *       zkp_fused_*_submit   queues host -> device copies, the flow and device -> host copies on the context's stream and returns;
 *       zkp_ctx_job_wait     blocks until that job is done and writes what needs the host (verdicts, invalid_point).
 *     The synchronous host-pointer calls of (2c) are exactly submit + wait.  A caller that wants throughput keeps SEVERAL contexts busy --
 *     on one GPU or on several (zkp_pipe of zkp_toolbox.h does that): the copies of one job overlap the kernels of the others.
 *     Host buffers: every input and output buffer named by a submit must stay valid and untouched until zkp_ctx_job_wait returns.  Pinned
 *     memory (zkp_host_alloc / zkp_host_register) makes the copies truly asynchronous; with ordinary memory HIP stages each copy and the
 *     submit call blocks for its duration (same results).
 *     flags: ZKP_JOB_SHARED_TRANSCRIPT -- `transcripts` is ONE 208-byte blob that every proof starts from (the reference's callers write
 *       `Transcript::new(label)` per proof: tests/zkp.rs:44, benches/zkp.rs:60) instead of [N][208].
 *     transcripts_out: NULL, or [N][208] receiving the advanced transcripts (what the reference leaves in its `&mut Transcript`s).
 *     inst_stride: proofs per row of the caller's `inst` array, >= N: row r of this job's instance points starts at inst + 32 * r * inst_stride.
 *       A job over the proof range [j0, j0 + N) of a larger batch passes inst + 32 * j0 and the batch size (weights_stride likewise for
 *       the [n_constraints][.][16] weights of the batch verifier) -- no gathering of columns on the host.
 *     entropy / weights16 == NULL: drawn ON THE DEVICE from the ChaCha20 stream (RFC 8439 block function, 64-bit counter from 0, 64-bit nonce)
 *       keyed with rng_seed[40] = key[32] || nonce[8], which the caller takes from the operating system per job -- what `thread_rng()` is
 *       to the reference (prover.rs:82, verifier.rs:153, batch_verifier.rs:179: rand 0.7's ThreadRng is a ChaCha stream keyed from the OS);
 *       entropy[j] = stream bytes [32 j, 32 j + 32), weights16 = the first 16 * n_constraints * N stream bytes in the array's own order.
 *     Errors (fail closed): the verdict words of a job -- results[N] of the two per-proof verifiers, verdicts[n_batches] of the batch verifier --
 *       are set to 1 (rejected) by the submit call itself as soon as its pointers have been checked, and only a job that ran to its end
 *       overwrites them: a failed submit (which leaves no job pending and has waited for whatever it had queued: the caller's buffers are free
 *       again) and a failed zkp_ctx_job_wait (< 0: the device failed underneath the job, or a copy out could not be issued by an earlier
 *       zkp_ctx_job_poll) both leave every verdict at "rejected", *invalid_point at 1, and no output buffer may be used.  The pinned words a
 *       wait derives verdicts from are poisoned at every submit, so nothing an earlier job left behind can read as "verified".
 *       Any other call on a context with a pending job returns ZKP_ERR_ARG.
 *     zkp_ctx_job_discard: for an owner that goes away with a job in flight (zkp_pipe_destroy): waits for the job's kernels, issues no copy
 *       out that was not already queued (with the default of ZKP_OPT_JOB_DEFER_D2H -- deferred copies --: none) and writes nothing to the caller's memory. */
#define ZKP_JOB_SHARED_TRANSCRIPT 1u
int zkp_fused_prove_submit(zkp_ctx* ctx, const zkp_fused_statement* st, uint32_t N, uint32_t flags, const uint8_t* transcripts,
                           const uint8_t* secrets, const uint8_t* inst, uint32_t inst_stride, const uint8_t* common,
                           const uint8_t* entropy, const uint8_t* rng_seed /*[40], used when entropy == NULL*/, uint8_t* transcripts_out,
                           uint8_t* challenges, uint8_t* responses, uint8_t* commitments, int* invalid_point);
int zkp_fused_verify_compact_submit(zkp_ctx* ctx, const zkp_fused_statement* st, uint32_t N, uint32_t flags, const uint8_t* transcripts,
                                    const uint8_t* inst, uint32_t inst_stride, const uint8_t* common, const uint8_t* challenges,
                                    const uint8_t* responses, uint8_t* transcripts_out, uint8_t* results /*[N]*/);
int zkp_fused_batch_verify_many_submit(zkp_ctx* ctx, const zkp_fused_statement* st, uint32_t n_batches, uint32_t N_each, uint32_t flags,
                                       const uint8_t* transcripts, const uint8_t* inst, uint32_t inst_stride, const uint8_t* common,
                                       const uint8_t* commitments, const uint8_t* responses, const uint8_t* weights16,
                                       uint32_t weights_stride, const uint8_t* rng_seed, uint8_t* transcripts_out, int* verdicts /*[n_batches]*/);
int zkp_fused_verify_batchable_submit(zkp_ctx* ctx, const zkp_fused_statement* st, uint32_t N, uint32_t flags, const uint8_t* transcripts,
                                      const uint8_t* inst, uint32_t inst_stride, const uint8_t* common, const uint8_t* commitments,
                                      const uint8_t* responses, const uint8_t* weights16 /*[N][n_constraints][16]*/, const uint8_t* rng_seed,
                                      uint8_t* transcripts_out, uint8_t* results /*[N]*/);
int zkp_ctx_job_wait(zkp_ctx* ctx);      /* no job pending: ZKP_OK at once */
int zkp_ctx_job_poll(zkp_ctx* ctx);      /* 1 = zkp_ctx_job_wait would not block, 0 = still running.  Polling also moves the job along: its copies
                                          * out are issued by the first poll (or wait) that finds its kernels finished -- see ZKP_OPT_JOB_DEFER_D2H */
int zkp_ctx_job_pending(zkp_ctx* ctx);   /* 1 = a job was submitted and not yet waited for */
int zkp_ctx_job_discard(zkp_ctx* ctx);   /* forget the pending job without touching the caller's memory (see Errors above) */
/* With zkp_ctx_set_profiling(ctx, 1): where the last finished job spent its time ON ITS STREAM, from HIP events recorded on that stream --
 * ms[0] host -> device copies (+ on-device randomness), ms[1] the flow's kernels, ms[2] device -> host copies.  Under load these include the
 * time the stream waited for the chip / the copy engines, which is what a pipelining caller wants to see. */
int zkp_ctx_job_timing(zkp_ctx* ctx, float ms[3]);

/* Pinned host memory for the jobs above (hipHostMalloc / hipHostRegister, visible to every GPU of the process).  zkp_host_is_pinned: 1 if p
 * points into such memory.  Registering costs ~12 us per MiB (profiles/r04_pcie_copy_rates.txt): register long-lived buffers once. */
int zkp_host_alloc(void** out, size_t bytes);
/* The same on the NUMA node of GPU `device` (round 5): on a two-socket host a staging buffer on the far socket sends every copied byte over the
 * socket interconnect first.  The runtime places pinned memory next to the calling thread's CURRENT device; zkp_host_alloc_on makes `device` current
 * for the allocation and restores the thread's device.  zkp_host_numa_node: the node the device hangs off (hipDeviceAttributeHostNumaId, else sysfs),
 * -1 = unknown; zkp_host_node_of: the node the page at p lives on (get_mempolicy), -1 = unknown / not permitted.  zkp_pipe's staging rings are
 * allocated this way (zkp_toolbox.h). */
int zkp_host_alloc_on(void** out, size_t bytes, int device);
int zkp_host_numa_node(int device);
int zkp_host_node_of(const void* p);
void zkp_host_free(void* p);
int zkp_host_register(void* p, size_t bytes);
int zkp_host_unregister(void* p);
int zkp_host_is_pinned(const void* p);

/* The device-side generator behind `entropy == NULL` / `weights16 == NULL`, exposed for the known-answer test: bytes (a multiple of 64)
 * of the ChaCha20 stream, block b = zkp_chacha20_block(key, first_block + b, nonce) of zkp_toolbox.h, into d_out (device, 16-byte aligned);
 * asynchronous on the context's stream. */
int zkp_chacha20_fill_dev(zkp_ctx* ctx, const uint8_t key[32], uint64_t nonce, uint64_t first_block, uint8_t* d_out, size_t bytes);

/* (3) Stand-alone decode / validity check, batched.  Replaces the
 *     `.map(|pt| pt.decompress()).collect::<Option<Vec<_>>>()` of verifier.rs:87-92.
 *     status[i] = 0 valid | 1 decompress() would return None.  If xyzt != NULL it receives the
 *     affine extended coordinates (X, Y, Z = 1, T) as 4 x 32-byte little-endian field elements. */
int zkp_decode_check(zkp_ctx* ctx, uint64_t n, const uint8_t* points /*[n][32]*/, uint8_t* status /*[n]*/,
                     uint8_t* xyzt /*[n][128] or NULL*/);

/* (4) Stand-alone encode.  Replaces `point.compress()` of mod.rs:180 for provers that hold
 *     uncompressed points (prover.rs:64-73).  Input: extended coordinates (X, Y, Z, T), each a
 *     32-byte little-endian field element (need not be reduced below p, bit 255 ignored). */
int zkp_encode_many(zkp_ctx* ctx, uint64_t n, const uint8_t* xyzt /*[n][128]*/, uint8_t* out /*[n][32]*/);

/* Timing of the last *_dev / host call on this context, measured with HIP events on the stream the
 * kernels were launched on.  kernel_ms[] is indexed by ZKP_K_*; returns the number of entries. */
enum {
  ZKP_K_DECODE = 0,      /* ristretto decode (+ affine-niels conversion, digit extraction)       */
  ZKP_K_TERMS = 1,       /* per-term scalar multiplication (small-MSM path)                       */
  ZKP_K_REDUCE = 2,      /* per-MSM sum of partials + compress                                    */
  ZKP_K_SORT = 3,        /* Pippenger: histogram + scan + scatter; small-MSM path: term classification    */
  ZKP_K_BUCKET = 4,      /* Pippenger: bucket accumulation                                        */
  ZKP_K_COMBINE = 5,     /* Pippenger: bucket reduction + window combination + compress           */
  ZKP_K_TRANSCRIPT = 6,  /* fused flows: batched Merlin/STROBE transcript programs                 */
  ZKP_K_SCALARS = 7,     /* fused flows: scalar arithmetic mod l (blindings, responses, coefficients, operand assembly) */
  ZKP_K_TABLES = 8,      /* small-MSM path: comb tables of the per-proof points (k_comb_tables*)                      */
  ZKP_K_COUNT = 9
};
int zkp_ctx_last_timing(zkp_ctx* ctx, float* kernel_ms /*[ZKP_K_COUNT]*/, float* total_ms);
/* With profiling on: which VARIANT of a kind's kernel the last call launched, by the name rocprofv3 prints -- the kernels whose template
 * arguments depend on the call's size, flags or options (ZKP_K_TERMS: "k_terms_split<true, 16, true, false>", ZKP_K_TABLES:
 * "zkp::k_comb_tables_lane<16>" / "zkp::k_tables_transcript_pc<16>", ZKP_K_TRANSCRIPT: "zkp::k_transcript_run" / "...run1", ZKP_K_DECODE of the
 * large-MSM path: "k_pip_prepare<11>"); several names are joined with ';', kinds whose kernels never vary give "".  Writes a NUL-terminated
 * string of at most cap - 1 characters and returns the untruncated length.  Profiles and benchmarks label kernels from THIS, not from a
 * copy of the dispatch thresholds. */
int zkp_ctx_last_kernels(zkp_ctx* ctx, int kind, char* buf, size_t cap);
/* Enable (1) / disable (0) per-kernel event timing (off by default: events add launch gaps). */
int zkp_ctx_set_profiling(zkp_ctx* ctx, int enabled);

#ifdef ZKP_BUILD_TEST_HOOKS
/* ---- test-hook builds only (zkp_amd/libzkp_mi355x_testhooks.so, compiled with -DZKP_BUILD_TEST_HOOKS; the shipped library
 *      has none of this: no extra options, no k_noop / k_debug_quad / k_debug_row kernels, no scratch in its code object) ----------------
 * zkp_debug_quad_selftest: exercises the 4-lane cooperative point arithmetic used by the latency-bound kernels.  pairs =
 *   [n][2][32] encodings (P, Q); out = [n][4][32] = enc(2P), enc(P+Q), enc(P+Q) through Q's niels form, enc(P-Q) through the
 *   negated niels form.
 * zkp_debug_row_selftest: the same for the one-limb-per-lane arithmetic of the Horner tail (zkp_amd/csrc/rowfe.h): out = [n][3][32] = enc(2P),
 *   enc(P+Q), enc(2^11 P + Q).
 * zkp_ctx_set_option extras (measurement, profiles/r02_ab_experiments.txt blocks o and p):
 *   ZKP_TESTOPT_DUMMY_LAUNCHES = n empty kernels added to every zkp_fused_prove_dev call (what a launch costs a pipelined caller);
 *   ZKP_TESTOPT_GENERIC_CLASSIFIER = 1 sends the fused flows through the generic six-kernel term classifier instead of
 *   k_stmt_classify;
 *   ZKP_TESTOPT_WAVE_CYCLES = 1 switches on a per-wavefront cycle recorder in the term kernel (s_memtime at entry and exit);
 * zkp_debug_wave_cycles copies out (and clears) up to cap records, [block][wavefront 0..3] = block class << 56 | cycles (class 1 =
 *   ladder, 2 = comb scan, 3 = grouped comb walk, 4 = fixed-base; 0 = no record): the timing side of the constant-time evidence. */
int zkp_debug_quad_selftest(zkp_ctx* ctx, uint32_t n, const uint8_t* pairs /*[n][64]*/, uint8_t* out /*[n][128]*/);
int zkp_debug_row_selftest(zkp_ctx* ctx, uint32_t n, const uint8_t* pairs /*[n][64]*/, uint8_t* out /*[n][96]*/);
int zkp_debug_wave_cycles(zkp_ctx* ctx, uint64_t* out, uint32_t cap);     /* returns the number of records copied */
enum { ZKP_TESTOPT_DUMMY_LAUNCHES = 1001, ZKP_TESTOPT_GENERIC_CLASSIFIER = 1002, ZKP_TESTOPT_WAVE_CYCLES = 1003 };
#endif

#ifdef __cplusplus
}
#endif
#endif /* ZKP_MI355X_H */
