//! UNBUILT SOURCE -- see ../../README.md.  Never compiled: no Rust toolchain exists in the build image.
//!
//! `extern "C"` declarations of `include/zkp_mi355x.h` (the HIP engine, libzkp_mi355x.so) and `include/zkp_toolbox.h`
//! (the host toolbox, libzkp_toolbox.so), one declaration per exported symbol, in header order.  Conventions (from the
//! headers): scalars and points are 32-byte strings; buffers are caller-owned and not retained; return 0 = computed,
//! negative = infrastructure failure (NO output may be trusted -- callers must fail closed); `_dev` variants take device
//! pointers and do not synchronise.
#![allow(non_camel_case_types)]
use std::os::raw::{c_char, c_int, c_void};

#[repr(C)]
pub struct zkp_ctx {
    _private: [u8; 0],
}
#[repr(C)]
pub struct zkp_graph {
    _private: [u8; 0],
}
#[repr(C)]
pub struct zkp_statement {
    _private: [u8; 0],
}
#[repr(C)]
pub struct zkp_pipe {
    _private: [u8; 0],
}
#[repr(C)]
pub struct zkp_job {
    _private: [u8; 0],
}

pub const ZKP_OK: c_int = 0;
pub const ZKP_ERR_HIP: c_int = -1;
pub const ZKP_ERR_ARG: c_int = -2;
pub const ZKP_ERR_NO_DEVICE: c_int = -3;
pub const ZKP_ERR_OOM: c_int = -4;
pub const ZKP_VARTIME: c_int = 0; // RistrettoPoint::vartime_multiscalar_mul (verifier.rs:97)
pub const ZKP_CT: c_int = 1; //      RistrettoPoint::multiscalar_mul (prover.rs:94)
pub const ZKP_OPT_BATCH_ENCODE_MIN: c_int = 1;
pub const ZKP_OPT_COMB_TEETH: c_int = 2;
pub const ZKP_OPT_CT_SINGLE_USE_TABLES: c_int = 3;
pub const ZKP_OPT_TRANSCRIPT_LANES: c_int = 4;
pub const ZKP_OPT_DEV_OVERLAP: c_int = 5;
pub const ZKP_OPT_GROUPED_COMB: c_int = 6;
pub const ZKP_OPT_TABLES_LANE: c_int = 7;
pub const ZKP_OPT_FUSE_TABLES_TRANSCRIPT: c_int = 8;
pub const ZKP_OPT_CT_LOOKUP: c_int = 9;            // 0 lane crossbar (default), 1 masked scans, 2 LDS rows at the digit's index
pub const ZKP_OPT_CT_MASKED_SCANS: c_int = 9;      // the round-3 name of the same option
pub const ZKP_OPT_EACH_STRAUS: c_int = 10;
pub const ZKP_OPT_LADDER_INTERLEAVE: c_int = 11;
pub const ZKP_OPT_WS_LIMIT_BYTES: c_int = 12;
pub const ZKP_OPT_JOB_DEFER_D2H: c_int = 13;
pub const ZKP_OPT_SYNC_SCHEDULE: c_int = 14;
pub const ZKP_OPT_TRANSCRIPT_STEPS: c_int = 15;
pub const ZKP_OPT_COMB_SPLIT: c_int = 16;
pub const ZKP_OPT_JOINT_LADDER: c_int = 17;
pub const ZKP_JOB_SHARED_TRANSCRIPT: u32 = 1;

pub const ZKP_TB_OK: c_int = 0;
pub const ZKP_TB_VERIFICATION_FAILURE: c_int = 1; // ProofError::VerificationFailure (errors.rs:6)
pub const ZKP_TB_BATCH_SIZE_MISMATCH: c_int = 2; //  ProofError::BatchSizeMismatch   (errors.rs:9)
pub const ZKP_TB_PIPE_FULL: c_int = 3; //             every context of a zkp_pipe has a job in flight
pub const ZKP_TB_BAD_STATEMENT: c_int = -10;
pub const ZKP_TB_INVALID_POINT: c_int = -11;
pub const ZKP_TB_NO_ENTROPY: c_int = -12;
pub const ZKP_TB_BAD_ENCODING: c_int = -13;
pub const ZKP_TRANSCRIPT_BYTES: usize = 208;

/// `zkp_batch_statement`: the statement in point-id form (static ids first, then instance ids).
#[repr(C)]
pub struct zkp_batch_statement {
    pub n_secrets: u32,
    pub n_static: u32,
    pub n_instance: u32,
    pub n_constraints: u32,
    pub cons_lhs: *const u32,
    pub cons_off: *const u32,
    pub cons_sc: *const u32,
    pub cons_pt: *const u32,
}
/// `zkp_fused_statement`: shape + the transcript labels and the allocation order.
#[repr(C)]
pub struct zkp_fused_statement {
    pub shape: zkp_batch_statement,
    pub label: *const c_char,
    pub secret_labels: *const *const c_char,
    pub point_labels: *const *const c_char,
    pub alloc_order: *const u32,
    pub alloc_seq: *const u32, // NULL = define_proof!'s order (all secrets first)
}

extern "C" {
    // ---- zkp_mi355x.h: context ----------------------------------------------------------------------------------
    pub fn zkp_ctx_create(out: *mut *mut zkp_ctx, device_id: c_int) -> c_int;
    pub fn zkp_ctx_destroy(ctx: *mut zkp_ctx);
    pub fn zkp_ctx_set_stream(ctx: *mut zkp_ctx, hip_stream: *mut c_void) -> c_int;
    pub fn zkp_ctx_synchronize(ctx: *mut zkp_ctx) -> c_int;
    pub fn zkp_last_error() -> *const c_char;
    pub fn zkp_version() -> *const c_char;
    pub fn zkp_ctx_set_option(ctx: *mut zkp_ctx, option: c_int, value: u64) -> c_int;
    pub fn zkp_ctx_capture_begin(ctx: *mut zkp_ctx) -> c_int;
    pub fn zkp_ctx_capture_end(ctx: *mut zkp_ctx, out: *mut *mut zkp_graph) -> c_int;
    pub fn zkp_ctx_capture_abort(ctx: *mut zkp_ctx) -> c_int;
    pub fn zkp_graph_launch(graph: *mut zkp_graph, ctx: *mut zkp_ctx) -> c_int;
    pub fn zkp_graph_destroy(graph: *mut zkp_graph);
    pub fn zkp_ctx_prepare_fixed_points(ctx: *mut zkp_ctx, n: u32, encodings: *const u8) -> c_int;
    // ---- (1) many small MSMs: prover.rs:94-97 (ZKP_CT), verifier.rs:97-106 (ZKP_VARTIME), + compress (mod.rs:204) ----
    pub fn zkp_msm_many(ctx: *mut zkp_ctx, n_msm: u32, off: *const u32, scalars: *const u8, pidx: *const u32, points: *const u8,
                        n_points: u32, flags: c_int, out: *mut u8, status: *mut u8) -> c_int;
    pub fn zkp_msm_many_dev(ctx: *mut zkp_ctx, n_msm: u32, d_off: *const u32, d_scalars: *const u8, d_pidx: *const u32,
                            d_points: *const u8, n_points: u32, n_terms: u32, flags: c_int, d_out: *mut u8, d_status: *mut u8) -> c_int;
    // ---- (2) one large MSM with decode-or-None: verifier.rs:162-166, batch_verifier.rs:219-228 ---------------------
    pub fn zkp_msm_optional(ctx: *mut zkp_ctx, n: u64, scalars: *const u8, points: *const u8, out_point: *mut u8, status: *mut c_int) -> c_int;
    pub fn zkp_msm_optional_dev(ctx: *mut zkp_ctx, n: u64, d_scalars: *const u8, d_points: *const u8, d_out_point: *mut u8,
                                d_status: *mut u32) -> c_int;
    // ---- (2b) batch_verifier.rs:173-234 with the coefficient build on the GPU ----------------------------------------
    pub fn zkp_batch_check(ctx: *mut zkp_ctx, st: *const zkp_batch_statement, n: u32, minus_c: *const u8, responses: *const u8,
                           weights16: *const u8, static_points: *const u8, instance_points: *const u8, commitments: *const u8,
                           out_point: *mut u8, status: *mut c_int, debug_scalars: *mut u8) -> c_int;
    // ---- (2c) fused statement flows ------------------------------------------------------------------------------------
    pub fn zkp_fused_prove(ctx: *mut zkp_ctx, st: *const zkp_fused_statement, n: u32, transcripts: *mut u8, secrets: *const u8,
                           inst: *const u8, common: *const u8, entropy: *const u8, challenges: *mut u8, responses: *mut u8,
                           commitments: *mut u8, invalid_point: *mut c_int) -> c_int;
    pub fn zkp_fused_prove_seeded(ctx: *mut zkp_ctx, st: *const zkp_fused_statement, n: u32, transcripts: *mut u8, secrets: *const u8, inst: *const u8,
                                  common: *const u8, seed: *const u8, challenges: *mut u8, responses: *mut u8, commitments: *mut u8, invalid_point: *mut c_int) -> c_int;
    pub fn zkp_fused_batch_verify_many_seeded(ctx: *mut zkp_ctx, st: *const zkp_fused_statement, n_batches: u32, n_each: u32, transcripts: *mut u8, inst: *const u8,
                                              common: *const u8, commitments: *const u8, responses: *const u8, seed: *const u8, verdicts: *mut c_int) -> c_int;
    pub fn zkp_fused_verify_compact(ctx: *mut zkp_ctx, st: *const zkp_fused_statement, n: u32, transcripts: *mut u8, inst: *const u8,
                                    common: *const u8, challenges: *const u8, responses: *const u8, results: *mut u8) -> c_int;
    pub fn zkp_fused_batch_verify(ctx: *mut zkp_ctx, st: *const zkp_fused_statement, n: u32, transcripts: *mut u8, inst: *const u8,
                                  common: *const u8, commitments: *const u8, responses: *const u8, weights16: *const u8,
                                  verdict: *mut c_int, debug_scalars: *mut u8) -> c_int;
    pub fn zkp_fused_batch_verify_many(ctx: *mut zkp_ctx, st: *const zkp_fused_statement, n_batches: u32, n_each: u32, transcripts: *mut u8,
                                       inst: *const u8, common: *const u8, commitments: *const u8, responses: *const u8,
                                       weights16: *const u8, verdicts: *mut c_int, debug_scalars: *mut u8) -> c_int;
    pub fn zkp_fused_verify_batchable(ctx: *mut zkp_ctx, st: *const zkp_fused_statement, n: u32, transcripts: *mut u8, inst: *const u8,
                                      common: *const u8, commitments: *const u8, responses: *const u8, weights16: *const u8,
                                      results: *mut u8) -> c_int;
    pub fn zkp_fused_verify_batchable_coeffs(ctx: *mut zkp_ctx, st: *const zkp_fused_statement, n: u32, transcripts: *mut u8,
                                             inst: *const u8, common: *const u8, commitments: *const u8, responses: *const u8,
                                             weights16: *const u8, results: *mut u8, debug_scalars: *mut u8) -> c_int;
    pub fn zkp_fused_prove_dev(ctx: *mut zkp_ctx, st: *const zkp_fused_statement, n: u32, strobe_pos: u32, d_transcripts: *mut u8,
                               d_secrets: *const u8, d_table: *const u8, d_entropy: *const u8, d_challenges: *mut u8,
                               d_responses: *mut u8, d_commitments: *mut u8, d_status: *mut u8) -> c_int;
    pub fn zkp_fused_verify_compact_dev(ctx: *mut zkp_ctx, st: *const zkp_fused_statement, n: u32, strobe_pos: u32,
                                        d_transcripts: *mut u8, d_table: *const u8, d_challenges: *const u8, d_responses: *const u8,
                                        d_results: *mut u8) -> c_int;
    pub fn zkp_fused_verify_batchable_dev(ctx: *mut zkp_ctx, st: *const zkp_fused_statement, n: u32, strobe_pos: u32,
                                          d_transcripts: *mut u8, d_table: *const u8, d_responses: *const u8,
                                          d_weights16: *const u8, d_results: *mut u8) -> c_int;
    pub fn zkp_fused_batch_verify_dev(ctx: *mut zkp_ctx, st: *const zkp_fused_statement, n: u32, strobe_pos: u32,
                                      d_transcripts: *mut u8, d_points: *mut u8, d_commitments: *const u8, d_responses: *const u8,
                                      d_weights16: *const u8, d_out_point: *mut u8, d_status: *mut u32) -> c_int;
    pub fn zkp_fused_batch_verify_many_dev(ctx: *mut zkp_ctx, st: *const zkp_fused_statement, n_batches: u32, n_each: u32, strobe_pos: u32,
                                           d_transcripts: *mut u8, d_points: *mut u8, d_commitments: *const u8, d_responses: *const u8,
                                           d_weights16: *const u8, d_out_points: *mut u8, d_status: *mut u32) -> c_int;
    // ---- (2d) the fused flows on host buffers as asynchronous jobs (one per context), pinned memory, device ChaCha20 ------
    pub fn zkp_fused_prove_submit(ctx: *mut zkp_ctx, st: *const zkp_fused_statement, n: u32, flags: u32, transcripts: *const u8, secrets: *const u8,
                                  inst: *const u8, inst_stride: u32, common: *const u8, entropy: *const u8, rng_seed: *const u8,
                                  transcripts_out: *mut u8, challenges: *mut u8, responses: *mut u8, commitments: *mut u8,
                                  invalid_point: *mut c_int) -> c_int;
    pub fn zkp_fused_verify_compact_submit(ctx: *mut zkp_ctx, st: *const zkp_fused_statement, n: u32, flags: u32, transcripts: *const u8,
                                           inst: *const u8, inst_stride: u32, common: *const u8, challenges: *const u8, responses: *const u8,
                                           transcripts_out: *mut u8, results: *mut u8) -> c_int;
    pub fn zkp_fused_batch_verify_many_submit(ctx: *mut zkp_ctx, st: *const zkp_fused_statement, n_batches: u32, n_each: u32, flags: u32,
                                              transcripts: *const u8, inst: *const u8, inst_stride: u32, common: *const u8,
                                              commitments: *const u8, responses: *const u8, weights16: *const u8, weights_stride: u32,
                                              rng_seed: *const u8, transcripts_out: *mut u8, verdicts: *mut c_int) -> c_int;
    pub fn zkp_fused_verify_batchable_submit(ctx: *mut zkp_ctx, st: *const zkp_fused_statement, n: u32, flags: u32, transcripts: *const u8,
                                             inst: *const u8, inst_stride: u32, common: *const u8, commitments: *const u8,
                                             responses: *const u8, weights16: *const u8, rng_seed: *const u8, transcripts_out: *mut u8,
                                             results: *mut u8) -> c_int;
    pub fn zkp_ctx_job_wait(ctx: *mut zkp_ctx) -> c_int;
    pub fn zkp_ctx_job_poll(ctx: *mut zkp_ctx) -> c_int;
    pub fn zkp_ctx_job_pending(ctx: *mut zkp_ctx) -> c_int;
    pub fn zkp_ctx_job_discard(ctx: *mut zkp_ctx) -> c_int;
    pub fn zkp_ctx_job_timing(ctx: *mut zkp_ctx, ms: *mut f32) -> c_int;
    pub fn zkp_host_alloc(out: *mut *mut c_void, bytes: usize) -> c_int;
    pub fn zkp_host_alloc_on(out: *mut *mut c_void, bytes: usize, device: c_int) -> c_int;
    pub fn zkp_host_numa_node(device: c_int) -> c_int;
    pub fn zkp_host_node_of(p: *const c_void) -> c_int;
    pub fn zkp_host_free(p: *mut c_void);
    pub fn zkp_host_register(p: *mut c_void, bytes: usize) -> c_int;
    pub fn zkp_host_unregister(p: *mut c_void) -> c_int;
    pub fn zkp_host_is_pinned(p: *const c_void) -> c_int;
    pub fn zkp_chacha20_fill_dev(ctx: *mut zkp_ctx, key: *const u8, nonce: u64, first_block: u64, d_out: *mut u8, bytes: usize) -> c_int;
    // ---- (3), (4) stand-alone codec: verifier.rs:87-92, mod.rs:180 -------------------------------------------------------
    pub fn zkp_decode_check(ctx: *mut zkp_ctx, n: u64, points: *const u8, status: *mut u8, xyzt: *mut u8) -> c_int;
    pub fn zkp_encode_many(ctx: *mut zkp_ctx, n: u64, xyzt: *const u8, out: *mut u8) -> c_int;
    pub fn zkp_ctx_last_timing(ctx: *mut zkp_ctx, kernel_ms: *mut f32, total_ms: *mut f32) -> c_int;
    pub fn zkp_ctx_last_kernels(ctx: *mut zkp_ctx, kind: c_int, buf: *mut c_char, cap: usize) -> c_int;
    pub fn zkp_ctx_set_profiling(ctx: *mut zkp_ctx, enabled: c_int) -> c_int;

    // ---- zkp_toolbox.h: Merlin transcripts, scalars ---------------------------------------------------------------------
    pub fn zkp_transcript_init(t: *mut u8, label: *const u8, label_len: usize);
    pub fn zkp_transcript_append_message(t: *mut u8, label: *const c_char, msg: *const u8, len: usize) -> c_int;
    pub fn zkp_transcript_challenge_bytes(t: *mut u8, label: *const c_char, out: *mut u8, len: usize) -> c_int;
    pub fn zkp_scalar_from_wide(out: *mut u8, input: *const u8);
    pub fn zkp_scalar_muladd(out: *mut u8, a: *const u8, b: *const u8, c: *const u8);
    pub fn zkp_scalar_neg(out: *mut u8, a: *const u8);
    // ---- statements (what define_proof! fixes: macros.rs:124-138, 159-170) ----------------------------------------------
    pub fn zkp_statement_new(proof_label: *const c_char) -> *mut zkp_statement;
    pub fn zkp_statement_free(st: *mut zkp_statement);
    pub fn zkp_statement_add_secret(st: *mut zkp_statement, name: *const c_char) -> c_int;
    pub fn zkp_statement_add_point(st: *mut zkp_statement, name: *const c_char, is_common: c_int) -> c_int;
    pub fn zkp_statement_constrain(st: *mut zkp_statement, lhs_point: u32, n_terms: u32, secrets: *const u32, points: *const u32) -> c_int;
    pub fn zkp_statement_num_secrets(st: *const zkp_statement) -> u32;
    pub fn zkp_statement_num_instance(st: *const zkp_statement) -> u32;
    pub fn zkp_statement_num_common(st: *const zkp_statement) -> u32;
    pub fn zkp_statement_num_constraints(st: *const zkp_statement) -> u32;
    pub fn zkp_statement_num_terms(st: *const zkp_statement) -> u32;
    // ---- batched flows -------------------------------------------------------------------------------------------------
    pub fn zkp_prove_batch(ctx: *mut zkp_ctx, st: *const zkp_statement, n: u32, transcripts: *mut u8, secrets: *const u8,
                           inst_points: *const u8, common_points: *const u8, entropy: *const u8, n_threads: c_int,
                           challenges: *mut u8, responses: *mut u8, commitments: *mut u8) -> c_int;
    pub fn zkp_verify_compact_batch(ctx: *mut zkp_ctx, st: *const zkp_statement, n: u32, transcripts: *mut u8, inst_points: *const u8,
                                    common_points: *const u8, challenges: *const u8, responses: *const u8, n_threads: c_int,
                                    results: *mut u8) -> c_int;
    pub fn zkp_verify_batchable_each(ctx: *mut zkp_ctx, st: *const zkp_statement, n: u32, transcripts: *mut u8, inst_points: *const u8,
                                     common_points: *const u8, commitments: *const u8, responses: *const u8, weights16: *const u8,
                                     n_threads: c_int, results: *mut u8) -> c_int;
    pub fn zkp_batch_verify(ctx: *mut zkp_ctx, st: *const zkp_statement, n: u32, n_transcripts: u32, transcripts: *mut u8,
                            inst_points: *const u8, common_points: *const u8, commitments: *const u8, responses: *const u8,
                            weights16: *const u8, n_threads: c_int) -> c_int;
    pub fn zkp_batch_verify_coeffs(ctx: *mut zkp_ctx, st: *const zkp_statement, n: u32, n_transcripts: u32, transcripts: *mut u8,
                                   inst_points: *const u8, common_points: *const u8, commitments: *const u8, responses: *const u8,
                                   weights16: *const u8, n_threads: c_int, coeffs: *mut u8) -> c_int;
    pub fn zkp_batch_verify_many(ctx: *mut zkp_ctx, st: *const zkp_statement, n_batches: u32, n_each: u32, n_transcripts: u32,
                                 transcripts: *mut u8, inst_points: *const u8, common_points: *const u8, commitments: *const u8,
                                 responses: *const u8, weights16: *const u8, n_threads: c_int, verdicts: *mut c_int) -> c_int;
    pub fn zkp_batch_verify_locate(ctx: *mut zkp_ctx, st: *const zkp_statement, n: u32, n_transcripts: u32, transcripts: *mut u8,
                                   inst_points: *const u8, common_points: *const u8, commitments: *const u8, responses: *const u8,
                                   weights16: *const u8, n_threads: c_int, results: *mut u8) -> c_int;
    // ---- pipelines and device groups: contexts over one or several GPUs, asynchronous jobs, sharded synchronous calls ------
    pub fn zkp_pipe_create(out: *mut *mut zkp_pipe, device_ids: *const c_int, n_devices: c_int, contexts_per_device: c_int) -> c_int;
    pub fn zkp_pipe_destroy(pipe: *mut zkp_pipe);
    pub fn zkp_pipe_num_contexts(pipe: *const zkp_pipe) -> c_int;
    pub fn zkp_pipe_num_devices(pipe: *const zkp_pipe) -> c_int;
    pub fn zkp_pipe_context(pipe: *mut zkp_pipe, i: c_int) -> *mut zkp_ctx;
    pub fn zkp_pipe_context_device(pipe: *const zkp_pipe, i: c_int) -> c_int;
    pub fn zkp_pipe_shard_plan(n_items: u32, unit: u32, n_contexts: u32, fused_min_batch: u32, lo: *mut u32, hi: *mut u32) -> u32;
    pub fn zkp_pipe_jobs_in_flight(pipe: *const zkp_pipe) -> c_int;
    pub fn zkp_pipe_set_submit_threads(pipe: *mut zkp_pipe, on: c_int) -> c_int;
    pub fn zkp_pipe_last_error(pipe: *const zkp_pipe) -> *const c_char;
    pub fn zkp_prove_batch_submit(pipe: *mut zkp_pipe, st: *const zkp_statement, n: u32, flags: u32, transcripts: *const u8, secrets: *const u8,
                                  inst_points: *const u8, inst_stride: u32, common_points: *const u8, entropy: *const u8,
                                  transcripts_out: *mut u8, challenges: *mut u8, responses: *mut u8, commitments: *mut u8,
                                  job: *mut *mut zkp_job) -> c_int;
    pub fn zkp_verify_compact_batch_submit(pipe: *mut zkp_pipe, st: *const zkp_statement, n: u32, flags: u32, transcripts: *const u8,
                                           inst_points: *const u8, inst_stride: u32, common_points: *const u8, challenges: *const u8,
                                           responses: *const u8, transcripts_out: *mut u8, results: *mut u8, job: *mut *mut zkp_job) -> c_int;
    pub fn zkp_verify_batchable_each_submit(pipe: *mut zkp_pipe, st: *const zkp_statement, n: u32, flags: u32, transcripts: *const u8,
                                            inst_points: *const u8, inst_stride: u32, common_points: *const u8, commitments: *const u8,
                                            responses: *const u8, weights16: *const u8, transcripts_out: *mut u8, results: *mut u8,
                                            job: *mut *mut zkp_job) -> c_int;
    pub fn zkp_batch_verify_many_submit(pipe: *mut zkp_pipe, st: *const zkp_statement, n_batches: u32, n_each: u32, flags: u32,
                                        transcripts: *const u8, inst_points: *const u8, inst_stride: u32, common_points: *const u8,
                                        commitments: *const u8, responses: *const u8, weights16: *const u8, weights_stride: u32,
                                        transcripts_out: *mut u8, verdicts: *mut c_int, job: *mut *mut zkp_job) -> c_int;
    pub fn zkp_job_context_index(job: *const zkp_job) -> c_int;
    pub fn zkp_job_done(job: *const zkp_job) -> c_int;
    pub fn zkp_job_wait(job: *mut zkp_job) -> c_int;
    pub fn zkp_pipe_prove_batch(pipe: *mut zkp_pipe, st: *const zkp_statement, n: u32, transcripts: *mut u8, secrets: *const u8,
                                inst_points: *const u8, common_points: *const u8, entropy: *const u8, challenges: *mut u8,
                                responses: *mut u8, commitments: *mut u8) -> c_int;
    pub fn zkp_pipe_verify_compact_batch(pipe: *mut zkp_pipe, st: *const zkp_statement, n: u32, transcripts: *mut u8, inst_points: *const u8,
                                         common_points: *const u8, challenges: *const u8, responses: *const u8, results: *mut u8) -> c_int;
    pub fn zkp_pipe_verify_batchable_each(pipe: *mut zkp_pipe, st: *const zkp_statement, n: u32, transcripts: *mut u8, inst_points: *const u8,
                                          common_points: *const u8, commitments: *const u8, responses: *const u8, weights16: *const u8,
                                          results: *mut u8) -> c_int;
    pub fn zkp_pipe_batch_verify(pipe: *mut zkp_pipe, st: *const zkp_statement, n: u32, n_transcripts: u32, transcripts: *mut u8,
                                 inst_points: *const u8, common_points: *const u8, commitments: *const u8, responses: *const u8,
                                 weights16: *const u8) -> c_int;
    pub fn zkp_pipe_batch_verify_many(pipe: *mut zkp_pipe, st: *const zkp_statement, n_batches: u32, n_each: u32, n_transcripts: u32,
                                      transcripts: *mut u8, inst_points: *const u8, common_points: *const u8, commitments: *const u8,
                                      responses: *const u8, weights16: *const u8, verdicts: *mut c_int) -> c_int;
    pub fn zkp_pipe_batch_verify_locate(pipe: *mut zkp_pipe, st: *const zkp_statement, n: u32, n_transcripts: u32, transcripts: *mut u8,
                                        inst_points: *const u8, common_points: *const u8, commitments: *const u8, responses: *const u8,
                                        weights16: *const u8, results: *mut u8) -> c_int;
    pub fn zkp_toolbox_set_host_max_terms(n: u32);
    pub fn zkp_toolbox_get_host_max_terms() -> u32;
    pub fn zkp_toolbox_set_fused_min_batch(n: u32);
    pub fn zkp_toolbox_get_fused_min_batch() -> u32;
    pub fn zkp_chacha20_block(key: *const u8, counter: u64, nonce: u64, out: *mut u8);
    // ---- proof wire format (proofs.rs:14-32 under bincode 1.x) ------------------------------------------------------------
    pub fn zkp_proof_compact_size(m: u32) -> usize;
    pub fn zkp_proof_batchable_size(nc: u32, m: u32) -> usize;
    pub fn zkp_proof_compact_encode(challenge: *const u8, responses: *const u8, m: u32, out: *mut u8, out_len: usize) -> c_int;
    pub fn zkp_proof_compact_decode(input: *const u8, len: usize, challenge: *mut u8, responses: *mut u8, max_m: u32, m: *mut u32,
                                    consumed: *mut usize) -> c_int;
    pub fn zkp_proof_batchable_encode(commitments: *const u8, nc: u32, responses: *const u8, m: u32, out: *mut u8, out_len: usize) -> c_int;
    pub fn zkp_proof_batchable_decode(input: *const u8, len: usize, commitments: *mut u8, max_nc: u32, nc: *mut u32, responses: *mut u8,
                                      max_m: u32, m: *mut u32, consumed: *mut usize) -> c_int;
    // ---- host-only halves (tests) -----------------------------------------------------------------------------------------
    pub fn zkp_batch_verify_build(st: *const zkp_statement, n: u32, n_transcripts: u32, transcripts: *mut u8, inst_points: *const u8,
                                  common_points: *const u8, commitments: *const u8, responses: *const u8, weights16: *const u8,
                                  n_threads: c_int, msm_scalars: *mut u8, msm_points: *mut u8) -> c_int;
    pub fn zkp_prove_phase_a(st: *const zkp_statement, n: u32, transcripts: *mut u8, secrets: *const u8, inst_points: *const u8,
                             common_points: *const u8, entropy: *const u8, n_threads: c_int, blindings: *mut u8, off: *mut u32,
                             scalars: *mut u8, pidx: *mut u32) -> c_int;
    pub fn zkp_prove_phase_b(st: *const zkp_statement, n: u32, transcripts: *mut u8, secrets: *const u8, blindings: *const u8,
                             commitments: *const u8, n_threads: c_int, challenges: *mut u8, responses: *mut u8) -> c_int;
}
