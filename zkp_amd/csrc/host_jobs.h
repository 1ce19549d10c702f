// host_jobs.h -- the fused flows on HOST buffers, asynchronously (round 4; included by zkp_kernels.hip after fused_flows.h).
//
// A Rust caller of the reference hands over host memory (`&[Scalar]`, `&[CompressedRistretto]`, `&mut Transcript`) and gets host
// memory back.  The synchronous host-pointer entry points of round 3 ran ONE call chain at a time with its copies in front of and
// behind it: 1.08 M proofs/s at 4,096 proofs per call against 6.1 M/s for the same flows on resident buffers.  The link is not the
// limit (profiles/r04_pcie_copy_rates.txt: 57 GB/s per direction, 97 GB/s both ways) -- the lone chain is.  Here a call is split
// into
//     zkp_fused_*_submit   enqueue host -> device copies, the flow, device -> host copies on the context's stream; return
//     zkp_ctx_job_wait     wait for that stream position, derive verdicts from the few status words the job left in pinned memory
// so that a caller (include/zkp_toolbox.h: zkp_pipe) keeps several contexts -- on one GPU or on several -- busy at once: the copies
// of one job run on the SDMA engines while the kernels of the others fill the chip.  One job in flight per context.
//
// What else changes against the synchronous calls, all of it what the reference does per call on the host:
//   * entropy (prover.rs:82 `thread_rng()`) and the batch weights (batch_verifier.rs:179) can be drawn ON THE DEVICE: a ChaCha20
//     stream (RFC 8439 block function, the generator behind rand 0.7's ThreadRng) keyed with 40 bytes the caller took from the
//     operating system -- 32 + 176 bytes per CMZ proof that never cross the link, no host thread hashing them;
//   * ZKP_JOB_SHARED_TRANSCRIPT: every proof starts from ONE transcript state (the reference's callers write
//     `Transcript::new(b"...")` per proof): 208 bytes instead of 208 N; transcripts_out == NULL: the advanced states stay on the
//     device (the reference's callers drop them);
//   * inst / weights rows may be slices of a longer row (`inst_stride` proofs per row in the caller's array): a proof range
//     [j0, j0 + N) of a batch that is sharded over contexts / GPUs is passed without gathering its columns on the host.
// The synchronous entry points of zkp_mi355x.h (2c) are these jobs followed by zkp_ctx_job_wait.
#pragma once

namespace zkp {

// RFC 8439 section 2.3 block function; block b of the stream = counter first_block + b, 64-bit nonce in words 14-15 (the layout
// of host/toolbox.cpp's chacha20_block, which the known-answer test pins)
struct chacha_key { uint32_t k[8]; };
__device__ __forceinline__ uint32_t cc_rotl(uint32_t v, int n) { return __builtin_amdgcn_alignbit(v, v, 32 - n); }
__global__ void __launch_bounds__(256)
k_chacha20_fill(chacha_key key, uint64_t nonce, uint64_t first_block, uint64_t n_blocks, uint32_t* __restrict__ out) {
  const uint64_t b = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= n_blocks) return;
  const uint64_t ctr = first_block + b;
  uint32_t s[16] = {0x61707865u, 0x3320646eu, 0x79622d32u, 0x6b206574u, key.k[0], key.k[1], key.k[2], key.k[3], key.k[4], key.k[5], key.k[6], key.k[7],
                    (uint32_t)ctr, (uint32_t)(ctr >> 32), (uint32_t)nonce, (uint32_t)(nonce >> 32)};
  uint32_t x[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) x[i] = s[i];
#define ZKP_CC_QR(a, b, c, d)                                                                          \
  x[a] += x[b]; x[d] = cc_rotl(x[d] ^ x[a], 16); x[c] += x[d]; x[b] = cc_rotl(x[b] ^ x[c], 12);        \
  x[a] += x[b]; x[d] = cc_rotl(x[d] ^ x[a], 8);  x[c] += x[d]; x[b] = cc_rotl(x[b] ^ x[c], 7);
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    ZKP_CC_QR(0, 4, 8, 12) ZKP_CC_QR(1, 5, 9, 13) ZKP_CC_QR(2, 6, 10, 14) ZKP_CC_QR(3, 7, 11, 15)
    ZKP_CC_QR(0, 5, 10, 15) ZKP_CC_QR(1, 6, 11, 12) ZKP_CC_QR(2, 7, 8, 13) ZKP_CC_QR(3, 4, 9, 14)
  }
#undef ZKP_CC_QR
  uint4* o = reinterpret_cast<uint4*>(out + 16 * b);
#pragma unroll
  for (int q = 0; q < 4; ++q) o[q] = make_uint4(x[4 * q] + s[4 * q], x[4 * q + 1] + s[4 * q + 1], x[4 * q + 2] + s[4 * q + 2], x[4 * q + 3] + s[4 * q + 3]);
}

// every proof starts from the same 208-byte transcript state (13 chunks of 16 bytes)
__global__ void __launch_bounds__(256)
k_broadcast_transcript(uint32_t N, const uint4* __restrict__ blob, uint4* __restrict__ ts) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < (size_t)N * 13) ts[i] = blob[i % 13];
}

__global__ void __launch_bounds__(256)
k_any_nonzero_bytes(size_t n, const uint8_t* __restrict__ flags, uint32_t* __restrict__ any) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n && flags[i]) *any = 1;
}

}  // namespace zkp

namespace {

// ---- pinned scratch and the pending-job record of a context -------------------------------------------------------------
// (a SYNCHRONOUS call is the only job of its context and its caller waits right behind it: its copies out are queued with the kernels -- deferring them would
// only add a host round trip between the last kernel and the first copy)
inline bool job_defers(const zkp_ctx* c) { return c->defer_d2h != 0 && !c->job.inline_out; }
int job_begin(zkp_ctx* c, size_t pin_bytes) {
  if (c->job.kind) return fail(ZKP_ERR_ARG, "a submitted job is pending on this context: zkp_ctx_job_wait first");
  if (c->capturing) return fail(ZKP_ERR_ARG, "graph capture: host-buffer jobs cannot be recorded (their copies name the caller's buffers)");
  if (!c->job.done) HIP_TRY(hipEventCreateWithFlags(&c->job.done, hipEventDisableTiming));
  if (pin_bytes > c->job.pin_bytes) {
    if (c->job.pin) (void)hipHostFree(c->job.pin);
    c->job.pin = nullptr;
    c->job.pin_bytes = 0;
    void* p = nullptr;
    HIP_TRY(hipHostMalloc(&p, pin_bytes + 256, hipHostMallocDefault));
    c->job.pin = static_cast<uint8_t*>(p);
    c->job.pin_bytes = pin_bytes + 256;
  }
  c->job.outs.clear();
  c->job.inline_out = false;
  c->job.copies_issued = false;
  c->job.copy_err = hipSuccess;
  c->job.results = nullptr;
  c->job.n_results = 0;
  c->job.verdicts = nullptr;
  c->job.invalid_point = nullptr;
  // the words zkp_ctx_job_wait turns into verdicts start out as "rejected" / "invalid": whatever an earlier job left here can never read as "verified"
  if (c->job.pin && pin_bytes) memset(c->job.pin, 0xff, pin_bytes);
  if (!c->job.copied) HIP_TRY(hipEventCreateWithFlags(&c->job.copied, hipEventDisableTiming));
  c->job.timed = c->profiling;
  if (c->job.timed) {
    for (auto& e : c->job.tev) if (!e) HIP_TRY(hipEventCreate(&e));
    HIP_TRY(hipEventRecord(c->job.tev[0], c->stream));
  }
  return ZKP_OK;
}
// profiling (zkp_ctx_set_profiling): where a job's time on its stream goes -- copies in, kernels, copies out
void job_mark(zkp_ctx* c, int i) { if (c->job.timed) (void)hipEventRecord(c->job.tev[i], c->stream); }
// the job is on the stream: remember what zkp_ctx_job_wait has to derive from the pinned words
int job_commit(zkp_ctx* c, char kind, uint32_t K, int* verdicts, int* invalid_point) {
  if (!job_defers(c)) job_mark(c, 3);
  HIP_TRY(hipEventRecord(c->job.done, c->stream));      // deferred copies out: this marks the end of the KERNELS
  c->job.kind = kind;
  c->job.K = K;
  c->job.verdicts = verdicts;
  c->job.invalid_point = invalid_point;
  return ZKP_OK;
}
// the caller's per-proof verdict array of a 'V' / 'E' job, registered right after job_begin: every failure path sets it to "rejected"
void job_results(zkp_ctx* c, uint8_t* results, size_t n) { c->job.results = results; c->job.n_results = n; }
// fail closed: nothing a failed job left in the caller's verdict words may read as "verified" / "valid"
void job_reject_all(zkp_ctx* c, char kind) {
  if (c->job.results && c->job.n_results) memset(c->job.results, 1, c->job.n_results);
  if (kind == 'B' && c->job.verdicts) for (uint32_t b = 0; b < c->job.K; ++b) c->job.verdicts[b] = 1;
  if (c->job.invalid_point) *c->job.invalid_point = 1;
}
// A failure after part of a job was queued: the queued copies still name the caller's buffers, so drain the stream before the
// error goes back (the caller may free them), and leave no job pending.  Outputs are undefined, the code says so (fail closed).
int job_abort(zkp_ctx* c, int rc) {
  const std::string msg = g_last_error;
  (void)hipStreamSynchronize(c->stream);
  (void)hipGetLastError();
  c->pending_tr.offered = c->pending_tr.active = false;
  job_reject_all(c, 0);                                  // (the 'B' verdicts were preset to 1 by the submitter before anything was queued)
  c->job.kind = 0;
  c->job.outs.clear();
  c->job.results = nullptr;
  c->job.n_results = 0;
  g_last_error = msg;
  return rc;
}

// rows of `row_bytes` each: contiguous on the device, `src_pitch` bytes apart in the caller's array
int h2d_rows(zkp_ctx* c, void* dst, const void* src, size_t rows, size_t row_bytes, size_t src_pitch) {
  if (!rows || !row_bytes) return ZKP_OK;
  if (rows == 1 || src_pitch == row_bytes) HIP_TRY(hipMemcpyAsync(dst, src, rows * row_bytes, hipMemcpyHostToDevice, c->stream));
  else HIP_TRY(hipMemcpy2DAsync(dst, row_bytes, src, src_pitch, row_bytes, rows, hipMemcpyHostToDevice, c->stream));
  return ZKP_OK;
}
int h2d(zkp_ctx* c, void* dst, const void* src, size_t bytes) { return h2d_rows(c, dst, src, 1, bytes, bytes); }
int d2h(zkp_ctx* c, void* dst, const void* src, size_t bytes) {
  if (!bytes) return ZKP_OK;
  if (job_defers(c)) { c->job.outs.push_back({dst, src, bytes}); return ZKP_OK; }       // issued by zkp_ctx_job_wait, after the kernels
  HIP_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, c->stream));
  return ZKP_OK;
}
// `bytes` of the ChaCha20 stream (key = seed[0..32), nonce = seed[32..40) little endian) into a 64-byte-granular device buffer
int chacha_fill(zkp_ctx* c, const uint8_t seed[40], uint64_t first_block, void* d_out, size_t bytes) {
  if (!bytes) return ZKP_OK;
  chacha_key k;
  memcpy(k.k, seed, 32);
  uint64_t nonce;
  memcpy(&nonce, seed + 32, 8);
  const uint64_t blocks = (bytes + 63) / 64;
  hipLaunchKernelGGL(k_chacha20_fill, grid1(blocks, 256), dim3(256), 0, c->stream, k, nonce, first_block, blocks, static_cast<uint32_t*>(d_out));
  HIP_TRY(hipGetLastError());
  return ZKP_OK;
}
// the N transcript blobs of a job on the device: the caller's N blobs, or N copies of its one blob
int put_transcripts(zkp_ctx* c, uint32_t N, bool shared, const uint8_t* transcripts, uint8_t* d_ts, uint8_t* d_blob) {
  if (!shared) return h2d(c, d_ts, transcripts, (size_t)N * 208);
  const int rc = h2d(c, d_blob, transcripts, 208);
  if (rc) return rc;
  hipLaunchKernelGGL(k_broadcast_transcript, grid1((size_t)N * 13, 256), dim3(256), 0, c->stream, N, reinterpret_cast<const uint4*>(d_blob), reinterpret_cast<uint4*>(d_ts));
  HIP_TRY(hipGetLastError());
  return ZKP_OK;
}
int job_tail(const uint8_t* transcripts, uint32_t N, bool shared, uint32_t* pos) {
  if (shared) { *pos = transcripts[200] | (uint32_t)transcripts[201] << 8 | (uint32_t)transcripts[202] << 16; return ZKP_OK; }
  return common_tail(transcripts, N, pos);
}
#define ZKP_JOB_TRY(expr) do { const int rc_ = (expr); if (rc_) return job_abort(c, rc_); } while (0)

// ---- prove ------------------------------------------------------------------------------------------------------------------
// sync_variant: the schedule of the synchronous entry point (side stream, quad tables: one call's latency is what the caller sees)
int prove_job(zkp_ctx* c, const zkp_fused_statement* st, uint32_t N, uint32_t flags, const uint8_t* transcripts, const uint8_t* secrets,
              const uint8_t* inst, uint32_t inst_stride, const uint8_t* common, const uint8_t* entropy, const uint8_t* rng_seed,
              uint8_t* transcripts_out, uint8_t* challenges, uint8_t* responses, uint8_t* commitments, int* invalid_point, bool sync_variant) {
  if (!c) return fail(ZKP_ERR_ARG, "ctx is NULL");
  if (N == 0) { if (invalid_point) *invalid_point = 0; return ZKP_OK; }
  if (!transcripts) return fail(ZKP_ERR_ARG, "NULL pointer");
  const bool shared = (flags & ZKP_JOB_SHARED_TRANSCRIPT) != 0;
  uint32_t pos = 0;
  int rc = job_tail(transcripts, N, shared, &pos);
  if (rc) return rc;
  HIP_TRY(hipSetDevice(c->device));
  fused_plan* pl = nullptr;
  rc = get_plan(c, FLOW_PROVE, st, N, pos, &pl);
  if (rc) return rc;
  const fused_shape& s = pl->s;
  if ((!entropy && !rng_seed) || !challenges || !invalid_point || (s.m && (!secrets || !responses)) || (s.nc && !commitments) || (s.ni && !inst) || (s.ns && !common))
    return fail(ZKP_ERR_ARG, "NULL pointer");
  if (s.ni && inst_stride < N) return fail(ZKP_ERR_ARG, "inst_stride is smaller than N");
  if ((uint64_t)N * s.T > 0x7fffffffull || (uint64_t)s.ns + (uint64_t)s.ni * N > 0x7fffffffull) return fail(ZKP_ERR_ARG, "batch too large");
  const uint32_t m = s.m, nc = s.nc, n_points = s.ns + s.ni * N;
  carve cv;
  const size_t o_ts = cv.take((size_t)N * 208);
  const size_t o_sec = cv.take((size_t)N * m * 32 + 32);
  const size_t o_tbl = cv.take((size_t)n_points * 32 + 32);
  const size_t o_ent = cv.take((size_t)N * 32 + 64);
  const size_t o_coms = cv.take((size_t)N * nc * 32 + 32);
  const size_t o_st = cv.take((size_t)N * nc + 4);
  const size_t o_chal = cv.take((size_t)N * 32);
  const size_t o_resp = cv.take((size_t)N * m * 32 + 32);
  const size_t o_blob = cv.take(256);
  const size_t o_flag = cv.take(16);
  const prove_inter o = prove_carve(*pl, cv.off);
  rc = ensure_ws(c, o.end + terms_path_ws(n_points, N * s.T, N * s.nc, cfg_from_terms(pl->tpt.data(), s.T, s.ns, s.np, N, c->ct_comb_min(!sync_variant, (size_t)N * s.T))));
  if (rc) return rc;
  rc = job_begin(c, 16);
  if (rc) return rc;
  const ws_view w{static_cast<char*>(c->ws)};
  c->job.inline_out = sync_variant;
  // The points go first: decode, classification and comb tables (side stream of the latency schedule) need nothing else.  What only the transcripts read --
  // states, witnesses, entropy, two thirds of the bytes -- follows once that work is queued (a copy from pageable memory holds the host thread until it is
  // staged), so it crosses the link under those kernels.  The asynchronous jobs keep everything in front: their copies ride under OTHER jobs' kernels.
  if (s.ns) ZKP_JOB_TRY(h2d(c, w.base + o_tbl, common, (size_t)s.ns * 32));
  ZKP_JOB_TRY(h2d_rows(c, w.base + o_tbl + 32 * (size_t)s.ns, inst, s.ni, (size_t)N * 32, (size_t)inst_stride * 32));
  const std::function<int()> rest = [&]() -> int {
    int r = put_transcripts(c, N, shared, transcripts, w.u8(o_ts), w.u8(o_blob));
    if (!r && m) r = h2d(c, w.base + o_sec, secrets, (size_t)N * m * 32);
    if (!r) r = entropy ? h2d(c, w.base + o_ent, entropy, (size_t)N * 32) : chacha_fill(c, rng_seed, 0, w.base + o_ent, (size_t)N * 32);
    return r;
  };
  bool coms_out = false;
  const std::function<int()> early = [&]() -> int {          // (pageable destinations hold the host until the copy is done: only pinned ones may leave early)
    if (!nc || !zkp_host_is_pinned(commitments)) return ZKP_OK;
    coms_out = true;
    return d2h(c, commitments, w.base + o_coms, (size_t)N * nc * 32);
  };
  if (!sync_variant) ZKP_JOB_TRY(rest());
  job_mark(c, 1);
  ZKP_JOB_TRY(prove_core(c, *pl, o, w.u8(o_ts), w.u8(o_sec), w.u8(o_tbl), w.u8(o_ent), w.u8(o_chal), w.u8(o_resp), w.u8(o_coms), w.u8(o_st), /*overlap=*/sync_variant,
                         /*throughput=*/!sync_variant, sync_variant ? &rest : nullptr, sync_variant ? &early : nullptr,
                         /*late_early=*/zkp_host_is_pinned(transcripts) && (!m || zkp_host_is_pinned(secrets)) && (!entropy || zkp_host_is_pinned(entropy))));
  job_mark(c, 2);
  if (hipMemsetAsync(w.base + o_flag, 0, 4, c->stream) != hipSuccess) return job_abort(c, fail(ZKP_ERR_HIP, "hipMemsetAsync failed"));
  if (nc) hipLaunchKernelGGL(k_any_nonzero_bytes, grid1((size_t)N * nc, 256), dim3(256), 0, c->stream, (size_t)N * nc, w.u8(o_st), w.u32(o_flag));
  ZKP_JOB_TRY(d2h(c, challenges, w.base + o_chal, (size_t)N * 32));
  if (m) ZKP_JOB_TRY(d2h(c, responses, w.base + o_resp, (size_t)N * m * 32));
  if (nc && !coms_out) ZKP_JOB_TRY(d2h(c, commitments, w.base + o_coms, (size_t)N * nc * 32));
  if (transcripts_out) ZKP_JOB_TRY(d2h(c, transcripts_out, w.base + o_ts, (size_t)N * 208));
  ZKP_JOB_TRY(d2h(c, c->job.pin, w.base + o_flag, 4));
  ZKP_JOB_TRY(job_commit(c, 'P', 0, nullptr, invalid_point));
  return ZKP_OK;
}

// ---- verify_compact -----------------------------------------------------------------------------------------------------------
int verify_compact_job(zkp_ctx* c, const zkp_fused_statement* st, uint32_t N, uint32_t flags, const uint8_t* transcripts, const uint8_t* inst, uint32_t inst_stride,
                       const uint8_t* common, const uint8_t* challenges, const uint8_t* responses, uint8_t* transcripts_out, uint8_t* results, bool sync_variant) {
  if (!c) return fail(ZKP_ERR_ARG, "ctx is NULL");
  if (N == 0) return ZKP_OK;
  if (!transcripts) return fail(ZKP_ERR_ARG, "NULL pointer");
  const bool shared = (flags & ZKP_JOB_SHARED_TRANSCRIPT) != 0;
  uint32_t pos = 0;
  int rc = job_tail(transcripts, N, shared, &pos);
  if (rc) return rc;
  HIP_TRY(hipSetDevice(c->device));
  fused_plan* pl = nullptr;
  rc = get_plan(c, FLOW_VERIFY, st, N, pos, &pl);
  if (rc) return rc;
  const fused_shape& s = pl->s;
  if (!challenges || !results || (s.m && !responses) || (s.ni && !inst) || (s.ns && !common)) return fail(ZKP_ERR_ARG, "NULL pointer");
  memset(results, 1, N);                                             // rejected until the job's copy out says otherwise (whatever fails from here on)
  if (s.ni && inst_stride < N) return fail(ZKP_ERR_ARG, "inst_stride is smaller than N");
  const uint32_t m = s.m, n_points = s.ns + s.ni * N;
  if ((uint64_t)N * pl->T1 > 0x7fffffffull || (uint64_t)s.ns + (uint64_t)s.ni * N > 0x7fffffffull) return fail(ZKP_ERR_ARG, "batch too large");
  carve cv;
  const size_t o_ts = cv.take((size_t)N * 208);
  const size_t o_tbl = cv.take((size_t)n_points * 32 + 32);
  const size_t o_claim = cv.take((size_t)N * 32);
  const size_t o_resp = cv.take((size_t)N * m * 32 + 32);
  const size_t o_res = cv.take((size_t)N);
  const size_t o_blob = cv.take(256);
  const verify_inter o = verify_carve(*pl, cv.off);
  rc = ensure_ws(c, o.end + terms_path_ws(n_points, N * pl->T1, N * s.nc, verify_terms_cfg(c, *pl)));
  if (rc) return rc;
  rc = job_begin(c, 16);
  if (rc) return rc;
  job_results(c, results, N);
  const ws_view w{static_cast<char*>(c->ws)};
  c->job.inline_out = sync_variant;
  if (s.ns) ZKP_JOB_TRY(h2d(c, w.base + o_tbl, common, (size_t)s.ns * 32));
  ZKP_JOB_TRY(h2d_rows(c, w.base + o_tbl + 32 * (size_t)s.ns, inst, s.ni, (size_t)N * 32, (size_t)inst_stride * 32));
  const std::function<int()> rest = [&]() -> int {            // (see prove_job: the points first, the rest under the side stream's kernels)
    int r = put_transcripts(c, N, shared, transcripts, w.u8(o_ts), w.u8(o_blob));
    if (!r) r = h2d(c, w.base + o_claim, challenges, (size_t)N * 32);
    if (!r && m) r = h2d(c, w.base + o_resp, responses, (size_t)N * m * 32);
    return r;
  };
  if (!sync_variant) ZKP_JOB_TRY(rest());
  job_mark(c, 1);
  ZKP_JOB_TRY(verify_core(c, *pl, o, w.u8(o_ts), w.u8(o_tbl), w.u8(o_claim), w.u8(o_resp), w.u8(o_res), /*overlap=*/sync_variant, /*throughput=*/!sync_variant,
                          sync_variant ? &rest : nullptr, /*late_early=*/zkp_host_is_pinned(transcripts) && zkp_host_is_pinned(challenges) && (!m || zkp_host_is_pinned(responses))));
  job_mark(c, 2);
  ZKP_JOB_TRY(d2h(c, results, w.base + o_res, (size_t)N));
  if (transcripts_out) ZKP_JOB_TRY(d2h(c, transcripts_out, w.base + o_ts, (size_t)N * 208));
  ZKP_JOB_TRY(job_commit(c, 'V', 0, nullptr, nullptr));
  return ZKP_OK;
}

// ---- K batch verifications --------------------------------------------------------------------------------------------------
int batch_verify_job(zkp_ctx* c, const zkp_fused_statement* st, uint32_t K, uint32_t N_each, uint32_t flags, const uint8_t* transcripts, const uint8_t* inst,
                     uint32_t inst_stride, const uint8_t* common, const uint8_t* commitments, const uint8_t* responses, const uint8_t* weights16, uint32_t w_stride,
                     const uint8_t* rng_seed, uint8_t* transcripts_out, int* verdicts, uint8_t* debug_scalars, bool sync_variant) {
  if (!c) return fail(ZKP_ERR_ARG, "ctx is NULL");
  if (!verdicts) return fail(ZKP_ERR_ARG, "NULL pointer");
  fused_shape s0;
  int rc = check_fused_statement(st, s0);
  if (rc) return rc;
  uint64_t n_each = (uint64_t)s0.ns + ((uint64_t)s0.ni + s0.nc) * N_each;
  if (K != 1) { rc = check_many(K, N_each, s0, &n_each); if (rc) return rc; }
  else if (n_each > 0x7fffffffull) return fail(ZKP_ERR_ARG, "batch too large");
  const uint32_t N = K * N_each;
  if (N && !transcripts) return fail(ZKP_ERR_ARG, "NULL pointer");
  const bool shared = (flags & ZKP_JOB_SHARED_TRANSCRIPT) != 0;
  uint32_t pos = 0;
  rc = N ? job_tail(transcripts, N, shared, &pos) : ZKP_OK;
  if (rc) return rc;
  HIP_TRY(hipSetDevice(c->device));
  fused_plan* pl = nullptr;
  rc = get_plan(c, FLOW_BATCH, st, N, pos, &pl);
  if (rc) return rc;
  const fused_shape& s = pl->s;
  if (N && ((s.nc && (!commitments || (!weights16 && !rng_seed))) || (s.m && !responses) || (s.ni && !inst))) return fail(ZKP_ERR_ARG, "NULL pointer");
  if (s.ns && !common) return fail(ZKP_ERR_ARG, "NULL pointer");
  if (N && ((s.ni && inst_stride < N) || (s.nc && weights16 && w_stride < N))) return fail(ZKP_ERR_ARG, "row stride is smaller than the number of proofs");
  const uint32_t m = s.m, nc = s.nc, ns = s.ns, ni = s.ni;
  const size_t n_pts = (size_t)ns + ((size_t)ni + nc) * N, n_sc = (size_t)K * ns + ((size_t)ni + nc) * N;
  carve cv;
  const size_t o_pts = cv.take(n_pts * 32 + 32);
  const size_t o_out = cv.take((size_t)K * 32);
  const size_t o_st = cv.take((size_t)K * 8 + 8);
  const size_t o_ts = cv.take((size_t)N * 208);
  const size_t o_coms = cv.take((size_t)N * nc * 32 + 32);
  const size_t o_resp = cv.take((size_t)N * m * 32 + 32);
  const size_t o_w = cv.take((size_t)nc * N * 16 + 64);
  const size_t o_blob = cv.take(256);
  const batch_inter o = batch_carve(*pl, cv.off, K);
  rc = ensure_ws(c, o.end + (K == 1 ? optional_ws(n_each) : optional_many_ws(n_each, K)));
  if (rc) return rc;
  rc = job_begin(c, (size_t)K * 40);
  if (rc) return rc;
  const ws_view w{static_cast<char*>(c->ws)};
  c->job.inline_out = sync_variant;
  // points and commitments first (the assemble pass and the decoder on the side stream read them); transcript states once that work is queued; responses and
  // weights (nobody reads them before the coefficient build) once the chain is on its stream -- see prove_job
  if (ns) ZKP_JOB_TRY(h2d(c, w.base + o_pts, common, (size_t)ns * 32));
  if (N) {
    ZKP_JOB_TRY(h2d_rows(c, w.base + o_pts + 32 * (size_t)ns, inst, ni, (size_t)N * 32, (size_t)inst_stride * 32));
    if (nc) ZKP_JOB_TRY(h2d(c, w.base + o_coms, commitments, (size_t)N * nc * 32));
  }
  const std::function<int()> late_ts = [&]() -> int { return put_transcripts(c, N, shared, transcripts, w.u8(o_ts), w.u8(o_blob)); };
  const std::function<int()> late_sc = [&]() -> int {
    int r = ZKP_OK;
    if (nc) r = weights16 ? h2d_rows(c, w.base + o_w, weights16, nc, (size_t)N * 16, (size_t)w_stride * 16) : chacha_fill(c, rng_seed, 0, w.base + o_w, (size_t)nc * N * 16);
    if (!r && m) r = h2d(c, w.base + o_resp, responses, (size_t)N * m * 32);
    return r;
  };
  if (N && !sync_variant) { ZKP_JOB_TRY(late_ts()); ZKP_JOB_TRY(late_sc()); }
  job_mark(c, 1);
  ZKP_JOB_TRY(batch_core(c, *pl, o, w.u8(o_ts), w.u8(o_pts), w.u8(o_coms), w.u8(o_resp), w.u8(o_w), w.u8(o_out), w.u32(o_st), /*throughput=*/!sync_variant, K, /*overlap=*/sync_variant,
                         sync_variant ? &late_ts : nullptr, sync_variant ? &late_sc : nullptr,
                         /*late_early=*/N && zkp_host_is_pinned(transcripts) && (!m || zkp_host_is_pinned(responses)) && (!nc || !weights16 || zkp_host_is_pinned(weights16))));
  job_mark(c, 2);
  if (debug_scalars) ZKP_JOB_TRY(d2h(c, debug_scalars, w.base + o.sc, n_sc * 32));
  ZKP_JOB_TRY(d2h(c, c->job.pin, w.base + o_out, (size_t)K * 32));
  ZKP_JOB_TRY(d2h(c, c->job.pin + (size_t)K * 32, w.base + o_st, (size_t)K * 8));
  if (N && transcripts_out) ZKP_JOB_TRY(d2h(c, transcripts_out, w.base + o_ts, (size_t)N * 208));
  for (uint32_t b = 0; b < K; ++b) verdicts[b] = 1;                  // until zkp_ctx_job_wait says otherwise
  ZKP_JOB_TRY(job_commit(c, 'B', K, verdicts, nullptr));
  return ZKP_OK;
}

// ---- verify_batchable, one verdict per proof ----------------------------------------------------------------------------------
int verify_batchable_job(zkp_ctx* c, const zkp_fused_statement* st, uint32_t N, uint32_t flags, const uint8_t* transcripts, const uint8_t* inst, uint32_t inst_stride,
                         const uint8_t* common, const uint8_t* commitments, const uint8_t* responses, const uint8_t* weights16, const uint8_t* rng_seed,
                         uint8_t* transcripts_out, uint8_t* results, uint8_t* debug_scalars, bool sync_variant) {
  if (!c) return fail(ZKP_ERR_ARG, "ctx is NULL");
  if (N == 0) return ZKP_OK;
  if (!transcripts || !results) return fail(ZKP_ERR_ARG, "NULL pointer");
  const bool shared = (flags & ZKP_JOB_SHARED_TRANSCRIPT) != 0;
  uint32_t pos = 0;
  int rc = job_tail(transcripts, N, shared, &pos);
  if (rc) return rc;
  HIP_TRY(hipSetDevice(c->device));
  fused_plan* pl = nullptr;
  rc = get_plan(c, FLOW_BATCH, st, N, pos, &pl);          // same transcript program as the batch verifier (:134-142 = :152-167)
  if (rc) return rc;
  const fused_shape& s = pl->s;
  if ((s.nc && (!commitments || (!weights16 && !rng_seed))) || (s.m && !responses) || (s.ni && !inst) || (s.ns && !common)) return fail(ZKP_ERR_ARG, "NULL pointer");
  memset(results, 1, N);                                             // rejected until the job's copy out says otherwise
  if (s.ni && inst_stride < N) return fail(ZKP_ERR_ARG, "inst_stride is smaller than N");
  const uint32_t m = s.m, nc = s.nc, ns = s.ns, ni = s.ni;
  const size_t n_points = (size_t)ns + (size_t)ni * N + (size_t)N * nc, K = (size_t)s.np + nc;
  if (n_points > 0x7fffffffull || (size_t)N * K > 0x7fffffffull) return fail(ZKP_ERR_ARG, "batch too large");
  carve cv;
  const size_t o_ts = cv.take((size_t)N * 208);
  const size_t o_tbl = cv.take(n_points * 32 + 32);
  const size_t o_resp = cv.take((size_t)N * m * 32 + 32);
  const size_t o_w = cv.take((size_t)N * nc * 16 + 64);
  const size_t o_res = cv.take((size_t)N + 4);
  const size_t o_blob = cv.take(256);
  const each_inter o = each_carve(*pl, cv.off);
  rc = ensure_ws(c, each_uses_straus(c, *pl) ? straus_carve(c, *pl, o.end).end : o.end + terms_path_ws((uint32_t)n_points, (uint32_t)(N * K), N, each_terms_cfg(*pl)));
  if (rc) return rc;
  rc = job_begin(c, 16);
  if (rc) return rc;
  job_results(c, results, N);
  const ws_view w{static_cast<char*>(c->ws)};
  ZKP_JOB_TRY(put_transcripts(c, N, shared, transcripts, w.u8(o_ts), w.u8(o_blob)));
  if (ns) ZKP_JOB_TRY(h2d(c, w.base + o_tbl, common, (size_t)ns * 32));
  ZKP_JOB_TRY(h2d_rows(c, w.base + o_tbl + 32 * (size_t)ns, inst, ni, (size_t)N * 32, (size_t)inst_stride * 32));
  if (nc) {
    ZKP_JOB_TRY(h2d(c, w.base + o_tbl + 32 * ((size_t)ns + (size_t)ni * N), commitments, (size_t)N * nc * 32));
    if (weights16) ZKP_JOB_TRY(h2d(c, w.base + o_w, weights16, (size_t)N * nc * 16));
    else ZKP_JOB_TRY(chacha_fill(c, rng_seed, 0, w.base + o_w, (size_t)N * nc * 16));
  }
  if (m) ZKP_JOB_TRY(h2d(c, w.base + o_resp, responses, (size_t)N * m * 32));
  job_mark(c, 1);
  ZKP_JOB_TRY(each_core(c, *pl, o, w.u8(o_ts), w.u8(o_tbl), w.u8(o_resp), w.u8(o_w), w.u8(o_res), /*overlap=*/sync_variant || c->dev_overlap));
  job_mark(c, 2);
  if (debug_scalars) ZKP_JOB_TRY(d2h(c, debug_scalars, w.base + o.sc, (size_t)N * K * 32));
  ZKP_JOB_TRY(d2h(c, results, w.base + o_res, (size_t)N));
  if (transcripts_out) ZKP_JOB_TRY(d2h(c, transcripts_out, w.base + o_ts, (size_t)N * 208));
  ZKP_JOB_TRY(job_commit(c, 'E', 0, nullptr, nullptr));
  return ZKP_OK;
}
#undef ZKP_JOB_TRY

}  // namespace

extern "C" {

// ---- pinned host memory -----------------------------------------------------------------------------------------------------
int zkp_host_alloc(void** out, size_t bytes) {
  if (!out) return fail(ZKP_ERR_ARG, "NULL pointer");
  *out = nullptr;
  if (!bytes) return ZKP_OK;
  int count = 0;
  if (hipGetDeviceCount(&count) != hipSuccess || count == 0) return fail(ZKP_ERR_NO_DEVICE, "no HIP device visible");
  HIP_TRY(hipHostMalloc(out, bytes, hipHostMallocPortable));
  return ZKP_OK;
}
// The NUMA node a GPU hangs off (its PCIe root complex): hipDeviceAttributeHostNumaId, else /sys/bus/pci/devices/<domain:bus:device.function>/numa_node.  -1 = unknown
// (no such file, a single-node host, a container that hides it).
int zkp_host_numa_node(int device) {
  int v = -1;
  if (hipDeviceGetAttribute(&v, hipDeviceAttributeHostNumaId, device) == hipSuccess && v >= 0) return v;       // (the runtime's own answer first)
  (void)hipGetLastError();
  char bdf[64] = {0};
  if (hipDeviceGetPCIBusId(bdf, (int)sizeof(bdf) - 1, device) != hipSuccess) { (void)hipGetLastError(); return -1; }
  for (char* q = bdf; *q; ++q) *q = (char)tolower((unsigned char)*q);
  const std::string path = std::string("/sys/bus/pci/devices/") + bdf + "/numa_node";
  FILE* f = fopen(path.c_str(), "r");
  if (!f) return -1;
  int node = -1;
  if (fscanf(f, "%d", &node) != 1) node = -1;
  fclose(f);
  return node;
}
// Pinned memory on the NUMA node of `device`: on a two-socket host a staging ring on the far socket sends every byte of every copy over
// the socket interconnect first.  The HIP runtime places a hipHostMalloc on the node closest to the CALLING THREAD'S CURRENT DEVICE (unless
// hipHostMallocNumaUser hands the policy to the user), so the allocation is made with `device` current and the thread's device restored
// afterwards -- no memory-policy system calls, nothing a container's seccomp filter can refuse.  zkp_host_node_of says where a page landed.
int zkp_host_alloc_on(void** out, size_t bytes, int device) {
  if (!out) return fail(ZKP_ERR_ARG, "NULL pointer");
  *out = nullptr;
  if (!bytes) return ZKP_OK;
  int count = 0;
  if (hipGetDeviceCount(&count) != hipSuccess || count == 0) return fail(ZKP_ERR_NO_DEVICE, "no HIP device visible");
  if (device < 0 || device >= count) return fail(ZKP_ERR_ARG, "zkp_host_alloc_on: no such device");
  int before = -1;
  if (hipGetDevice(&before) != hipSuccess) { (void)hipGetLastError(); before = -1; }
  HIP_TRY(hipSetDevice(device));
  const hipError_t e = hipHostMalloc(out, bytes, hipHostMallocPortable);
  if (before >= 0 && before != device) (void)hipSetDevice(before);
  if (e != hipSuccess) { *out = nullptr; return fail(ZKP_ERR_HIP, std::string("hipHostMalloc: ") + hipGetErrorString(e)); }
  return ZKP_OK;
}
// the NUMA node the page at p lives on (-1: unknown / not permitted): get_mempolicy(MPOL_F_NODE | MPOL_F_ADDR)
int zkp_host_node_of(const void* p) {
#if defined(__linux__) && defined(SYS_get_mempolicy)
  int node = -1;
  if (p && syscall(SYS_get_mempolicy, &node, nullptr, 0ul, const_cast<void*>(p), 3ul /* MPOL_F_NODE | MPOL_F_ADDR */) == 0) return node;
#endif
  (void)p;
  return -1;
}
void zkp_host_free(void* p) { if (p) (void)hipHostFree(p); }
int zkp_host_register(void* p, size_t bytes) {
  if (!p || !bytes) return fail(ZKP_ERR_ARG, "NULL pointer");
  HIP_TRY(hipHostRegister(p, bytes, hipHostRegisterPortable));
  return ZKP_OK;
}
int zkp_host_unregister(void* p) {
  if (!p) return fail(ZKP_ERR_ARG, "NULL pointer");
  HIP_TRY(hipHostUnregister(p));
  return ZKP_OK;
}
int zkp_host_is_pinned(const void* p) {
  if (!p) return 0;
  hipPointerAttribute_t a;
  if (hipPointerGetAttributes(&a, p) != hipSuccess) { (void)hipGetLastError(); return 0; }
  return a.type == hipMemoryTypeHost ? 1 : 0;
}

int zkp_chacha20_fill_dev(zkp_ctx* c, const uint8_t key[32], uint64_t nonce, uint64_t first_block, uint8_t* d_out, size_t bytes) {
  if (!c || !key || (bytes && !d_out)) return fail(ZKP_ERR_ARG, "NULL pointer");
  if (!aligned16(d_out) || (bytes & 63)) return fail(ZKP_ERR_ARG, "d_out must be 16-byte aligned and bytes a multiple of 64");
  HIP_TRY(hipSetDevice(c->device));
  uint8_t seed[40];
  memcpy(seed, key, 32);
  memcpy(seed + 32, &nonce, 8);
  return chacha_fill(c, seed, first_block, d_out, bytes);
}

// ---- jobs ---------------------------------------------------------------------------------------------------------------------
int zkp_fused_prove_submit(zkp_ctx* c, const zkp_fused_statement* st, uint32_t N, uint32_t flags, const uint8_t* transcripts, const uint8_t* secrets,
                           const uint8_t* inst, uint32_t inst_stride, const uint8_t* common, const uint8_t* entropy, const uint8_t* rng_seed,
                           uint8_t* transcripts_out, uint8_t* challenges, uint8_t* responses, uint8_t* commitments, int* invalid_point) {
  return prove_job(c, st, N, flags, transcripts, secrets, inst, inst_stride, common, entropy, rng_seed, transcripts_out, challenges, responses, commitments, invalid_point, false);
}
int zkp_fused_verify_compact_submit(zkp_ctx* c, const zkp_fused_statement* st, uint32_t N, uint32_t flags, const uint8_t* transcripts, const uint8_t* inst,
                                    uint32_t inst_stride, const uint8_t* common, const uint8_t* challenges, const uint8_t* responses, uint8_t* transcripts_out,
                                    uint8_t* results) {
  return verify_compact_job(c, st, N, flags, transcripts, inst, inst_stride, common, challenges, responses, transcripts_out, results, false);
}
int zkp_fused_batch_verify_many_submit(zkp_ctx* c, const zkp_fused_statement* st, uint32_t n_batches, uint32_t N_each, uint32_t flags, const uint8_t* transcripts,
                                       const uint8_t* inst, uint32_t inst_stride, const uint8_t* common, const uint8_t* commitments, const uint8_t* responses,
                                       const uint8_t* weights16, uint32_t weights_stride, const uint8_t* rng_seed, uint8_t* transcripts_out, int* verdicts) {
  if (n_batches == 0 || N_each == 0) return fail(ZKP_ERR_ARG, "n_batches and N_each must be positive");
  return batch_verify_job(c, st, n_batches, N_each, flags, transcripts, inst, inst_stride, common, commitments, responses, weights16, weights_stride, rng_seed, transcripts_out,
                          verdicts, nullptr, false);
}
int zkp_fused_verify_batchable_submit(zkp_ctx* c, const zkp_fused_statement* st, uint32_t N, uint32_t flags, const uint8_t* transcripts, const uint8_t* inst,
                                      uint32_t inst_stride, const uint8_t* common, const uint8_t* commitments, const uint8_t* responses, const uint8_t* weights16,
                                      const uint8_t* rng_seed, uint8_t* transcripts_out, uint8_t* results) {
  return verify_batchable_job(c, st, N, flags, transcripts, inst, inst_stride, common, commitments, responses, weights16, rng_seed, transcripts_out, results, nullptr, false);
}

// ---- the synchronous host-pointer entry points of zkp_mi355x.h (2c): the same jobs with the low-latency schedule, then wait ----------
static int job_finish(zkp_ctx* c, int rc) { return rc ? rc : zkp_ctx_job_wait(c); }
// ZKP_OPT_SYNC_SCHEDULE: 0 (default) = a synchronous call runs the low-latency schedule (the caller sees one call's duration); 1 = the jobs' throughput schedule
static bool sync_latency(const zkp_ctx* c) { return !(c && c->sync_throughput); }
int zkp_fused_prove(zkp_ctx* c, const zkp_fused_statement* st, uint32_t N, uint8_t* transcripts, const uint8_t* secrets, const uint8_t* inst, const uint8_t* common,
                    const uint8_t* entropy, uint8_t* challenges, uint8_t* responses, uint8_t* commitments, int* invalid_point) {
  if (c && N && !entropy) return fail(ZKP_ERR_ARG, "NULL pointer");
  return job_finish(c, prove_job(c, st, N, 0, transcripts, secrets, inst, N, common, entropy, nullptr, transcripts, challenges, responses, commitments, invalid_point, sync_latency(c)));
}
int zkp_fused_prove_seeded(zkp_ctx* c, const zkp_fused_statement* st, uint32_t N, uint8_t* transcripts, const uint8_t* secrets, const uint8_t* inst, const uint8_t* common,
                           const uint8_t seed[40], uint8_t* challenges, uint8_t* responses, uint8_t* commitments, int* invalid_point) {
  if (c && N && !seed) return fail(ZKP_ERR_ARG, "NULL pointer");
  return job_finish(c, prove_job(c, st, N, 0, transcripts, secrets, inst, N, common, nullptr, seed, transcripts, challenges, responses, commitments, invalid_point, sync_latency(c)));
}
int zkp_fused_verify_compact(zkp_ctx* c, const zkp_fused_statement* st, uint32_t N, uint8_t* transcripts, const uint8_t* inst, const uint8_t* common,
                             const uint8_t* challenges, const uint8_t* responses, uint8_t* results) {
  return job_finish(c, verify_compact_job(c, st, N, 0, transcripts, inst, N, common, challenges, responses, transcripts, results, sync_latency(c)));
}
int zkp_fused_batch_verify_many(zkp_ctx* c, const zkp_fused_statement* st, uint32_t K, uint32_t N_each, uint8_t* transcripts, const uint8_t* inst, const uint8_t* common,
                                const uint8_t* commitments, const uint8_t* responses, const uint8_t* weights16, int* verdicts, uint8_t* debug_scalars) {
  if (c && (K == 0 || (K != 1 && N_each == 0))) return fail(ZKP_ERR_ARG, "n_batches and N_each must be positive");
  if (c && K * N_each && st && st->shape.n_constraints && !weights16) return fail(ZKP_ERR_ARG, "NULL pointer");
  return job_finish(c, batch_verify_job(c, st, K, N_each, 0, transcripts, inst, K * N_each, common, commitments, responses, weights16, K * N_each, nullptr, transcripts, verdicts,
                                        debug_scalars, sync_latency(c)));
}
int zkp_fused_batch_verify_many_seeded(zkp_ctx* c, const zkp_fused_statement* st, uint32_t K, uint32_t N_each, uint8_t* transcripts, const uint8_t* inst, const uint8_t* common,
                                       const uint8_t* commitments, const uint8_t* responses, const uint8_t seed[40], int* verdicts) {
  if (c && (K == 0 || N_each == 0)) return fail(ZKP_ERR_ARG, "n_batches and N_each must be positive");
  if (c && !seed) return fail(ZKP_ERR_ARG, "NULL pointer");
  return job_finish(c, batch_verify_job(c, st, K, N_each, 0, transcripts, inst, K * N_each, common, commitments, responses, nullptr, K * N_each, seed, transcripts, verdicts,
                                        nullptr, sync_latency(c)));
}
int zkp_fused_batch_verify(zkp_ctx* c, const zkp_fused_statement* st, uint32_t N, uint8_t* transcripts, const uint8_t* inst, const uint8_t* common,
                           const uint8_t* commitments, const uint8_t* responses, const uint8_t* weights16, int* verdict, uint8_t* debug_scalars) {
  return zkp_fused_batch_verify_many(c, st, 1, N, transcripts, inst, common, commitments, responses, weights16, verdict, debug_scalars);
}
int zkp_fused_verify_batchable_coeffs(zkp_ctx* c, const zkp_fused_statement* st, uint32_t N, uint8_t* transcripts, const uint8_t* inst, const uint8_t* common,
                                      const uint8_t* commitments, const uint8_t* responses, const uint8_t* weights16, uint8_t* results, uint8_t* debug_scalars) {
  if (c && N && st && st->shape.n_constraints && !weights16) return fail(ZKP_ERR_ARG, "NULL pointer");
  return job_finish(c, verify_batchable_job(c, st, N, 0, transcripts, inst, N, common, commitments, responses, weights16, nullptr, transcripts, results, debug_scalars, sync_latency(c)));
}
int zkp_fused_verify_batchable(zkp_ctx* c, const zkp_fused_statement* st, uint32_t N, uint8_t* transcripts, const uint8_t* inst, const uint8_t* common,
                               const uint8_t* commitments, const uint8_t* responses, const uint8_t* weights16, uint8_t* results) {
  return zkp_fused_verify_batchable_coeffs(c, st, N, transcripts, inst, common, commitments, responses, weights16, results, nullptr);
}

int zkp_ctx_job_timing(zkp_ctx* c, float ms[3]) {
  if (!c || !ms) return fail(ZKP_ERR_ARG, "NULL pointer");
  memcpy(ms, c->job.ms, sizeof(c->job.ms));
  return ZKP_OK;
}
int zkp_ctx_job_pending(zkp_ctx* c) { return c && c->job.kind ? 1 : 0; }
// the kernels of the pending job are done: issue its copies out (asynchronously) and mark their end
static hipError_t job_issue_copies(zkp_ctx* c) {
  hipError_t e = hipSuccess;
  for (const auto& o : c->job.outs)
    if (e == hipSuccess) e = hipMemcpyAsync(o.dst, o.src, o.bytes, hipMemcpyDeviceToHost, c->stream);
  if (c->job.timed) (void)hipEventRecord(c->job.tev[3], c->stream);
  if (e == hipSuccess) e = hipEventRecord(c->job.copied, c->stream);
  // `copied` stands for THIS job's copies only if it was recorded now: after a failure the event still carries an earlier job's record, and waiting on it
  // would "succeed" with nothing copied.  The error is kept for zkp_ctx_job_wait, which reports it and rejects every verdict.
  c->job.copies_issued = e == hipSuccess;
  c->job.copy_err = e;
  return e;
}
int zkp_ctx_job_poll(zkp_ctx* c) {
  if (!c) return fail(ZKP_ERR_ARG, "ctx is NULL");
  if (!c->job.kind) return 1;
  (void)hipSetDevice(c->device);
  if (c->job.copy_err != hipSuccess) return 1;                         // failed: zkp_ctx_job_wait reports it
  if (!c->job.copies_issued) {
    const hipError_t e = hipEventQuery(c->job.done);
    if (e == hipErrorNotReady) { (void)hipGetLastError(); return 0; }
    if (e != hipSuccess || c->job.outs.empty()) return 1;              // failed (zkp_ctx_job_wait reports it), or nothing to copy: done
    if (job_issue_copies(c) != hipSuccess) return 1;
    return 0;                                                          // the copies out are on their way
  }
  const hipError_t e = hipEventQuery(c->job.copied);
  if (e == hipErrorNotReady) { (void)hipGetLastError(); return 0; }
  return 1;                                                            // done, or failed: zkp_ctx_job_wait reports which
}
static int job_retire(zkp_ctx* c, bool discard) {
  if (!c) return fail(ZKP_ERR_ARG, "ctx is NULL");
  if (!c->job.kind) return ZKP_OK;
  const char kind = c->job.kind;
  (void)hipSetDevice(c->device);
  hipError_t e = c->job.copy_err;                                      // (a poll may already have failed to issue the copies)
  if (e == hipSuccess && !c->job.copies_issued) {
    e = hipEventSynchronize(c->job.done);
    // the kernels are done: now the copies out -- nothing waits inside the copy engine's queue (see job_t::outs)
    if (e == hipSuccess && !c->job.outs.empty() && !discard) e = job_issue_copies(c);
  }
  if (e == hipSuccess && c->job.copies_issued) e = hipEventSynchronize(c->job.copied);
  c->job.outs.clear();
  c->job.kind = 0;
  if (discard) {                                                       // the caller's memory may be gone: nothing is written to it
    c->job.results = nullptr; c->job.n_results = 0; c->job.verdicts = nullptr; c->job.invalid_point = nullptr;
    (void)hipGetLastError();
    return ZKP_OK;
  }
  if (e != hipSuccess) {
    // fail closed: whatever reached the caller's buffers must not be read as "verified" / "proven"
    (void)hipStreamSynchronize(c->stream);                             // (copies that WERE queued still name the caller's buffers)
    (void)hipGetLastError();
    job_reject_all(c, kind);
    return fail(ZKP_ERR_HIP, std::string("job failed on the device: ") + hipGetErrorString(e));
  }
  if (c->job.timed)
    for (int i = 0; i < 3; ++i) (void)hipEventElapsedTime(&c->job.ms[i], c->job.tev[i], c->job.tev[i + 1]);
  if (kind == 'P') {
    uint32_t any;
    memcpy(&any, c->job.pin, 4);
    *c->job.invalid_point = any ? 1 : 0;
  } else if (kind == 'B') {
    static const uint8_t zero[32] = {0};
    const uint32_t K = c->job.K;
    for (uint32_t b = 0; b < K; ++b) {                               // batch_verifier.rs:230-234, once per batch
      uint32_t stv[2];
      memcpy(stv, c->job.pin + (size_t)K * 32 + 8 * (size_t)b, 8);
      c->job.verdicts[b] = (stv[0] == 0 && stv[1] == 0 && memcmp(c->job.pin + 32 * (size_t)b, zero, 32) == 0) ? 0 : 1;
    }
  }
  return ZKP_OK;
}
int zkp_ctx_job_wait(zkp_ctx* c) { return job_retire(c, false); }
// Forget the pending job: waits for its kernels (and for copies out that were already queued), issues no further copy and writes nothing to
// the caller's verdict words.  For owners that go away with jobs in flight (zkp_pipe_destroy): with deferred copies out (the default) the
// caller's output buffers are never touched after this call was entered.
int zkp_ctx_job_discard(zkp_ctx* c) { return job_retire(c, true); }

}  // extern "C"
