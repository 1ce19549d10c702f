//! UNBUILT SOURCE -- see ../../README.md.  Never compiled: no Rust toolchain exists in the build image.
//!
//! Safe layer over `zkp-mi355x-sys`:
//!  * [`Engine`]: one context per GPU, error mapping (every non-zero code is an `Err`: fail closed);
//!  * [`multiscalar`]: replacements for the three dalek trait calls the reference's toolbox makes (the per-call route);
//!  * [`Statement`] + [`prove_batch`] / [`verify_compact_batch`] / [`batch_verify`]: the batched route, where transcripts,
//!    scalar arithmetic and MSMs of a whole batch of proofs of one statement run on the device;
//!  * [`Pipe`]: engine contexts over one or several GPUs -- asynchronous jobs on host buffers ([`Pipe::submit_prove`],
//!    [`Pipe::submit_batch_verify_many`], [`Job::wait`]) and the synchronous calls sharded over every context
//!    ([`Pipe::prove_batch`], [`Pipe::batch_verify`]: one verdict, the AND of the per-GPU batch checks).
//! Layouts are the ones `include/zkp_toolbox.h` documents: secrets / responses `[N][m][32]`, instance points
//! `[n_inst][N][32]` (row = variable, column = proof, like `BatchVerifier::allocate_instance_point`), common points
//! `[n_common][32]`, commitments `[N][n_constraints][32]`, transcripts `[N][208]`.
use std::ffi::{CStr, CString};
use std::os::raw::c_int;
use std::ptr;

use curve25519_dalek::ristretto::CompressedRistretto;
use curve25519_dalek::scalar::Scalar;
use zkp_mi355x_sys as sys;

/// Mirrors `zkp::ProofError` (src/errors.rs:4-11) plus the infrastructure failures of the C ABI.
#[derive(Debug)]
pub enum Error {
    VerificationFailure,
    BatchSizeMismatch,
    /// negative return code of the C ABI with `zkp_last_error()`; no output may be trusted
    Backend(c_int, String),
    /// a slice handed to a safe wrapper does not have the length the statement / batch size implies (checked before any FFI call)
    Shape(&'static str),
}
fn check(rc: c_int) -> Result<(), Error> {
    match rc {
        0 => Ok(()),
        sys::ZKP_TB_VERIFICATION_FAILURE => Err(Error::VerificationFailure),
        sys::ZKP_TB_BATCH_SIZE_MISMATCH => Err(Error::BatchSizeMismatch),
        rc => {
            let msg = unsafe { CStr::from_ptr(sys::zkp_last_error()) }.to_string_lossy().into_owned();
            Err(Error::Backend(rc, msg))
        }
    }
}

/// One engine context = one GPU (one process per GPU in multi-GPU jobs).  Not `Sync`: a context is not re-entrant.  It is `Send`: a thread pool keeps one
/// context per worker thread (contexts are independent of each other, as the reference's Prover / Verifier / BatchVerifier objects are).
pub struct Engine(*mut sys::zkp_ctx);
unsafe impl Send for Engine {}
impl Engine {
    pub fn new(device_id: i32) -> Result<Engine, Error> {
        let mut ctx = ptr::null_mut();
        check(unsafe { sys::zkp_ctx_create(&mut ctx, device_id) })?;
        Ok(Engine(ctx))
    }
    /// A context for one worker of a thread pool: the synchronous calls run the throughput schedule of the asynchronous jobs (ZKP_OPT_SYNC_SCHEDULE = 1)
    /// instead of the low-latency one a lone caller wants.  Six such workers reach the rate of `Pipe`'s jobs ("A thread pool of synchronous calls").
    pub fn for_thread_pool(device_id: i32) -> Result<Engine, Error> {
        let e = Engine::new(device_id)?;
        check(unsafe { sys::zkp_ctx_set_option(e.0, sys::ZKP_OPT_SYNC_SCHEDULE, 1) })?;
        Ok(e)
    }
    /// Hint: the statement's common points (define_proof!'s third list / BatchVerifier's static points) get fixed-base tables.
    pub fn prepare_fixed_points(&self, points: &[CompressedRistretto]) -> Result<(), Error> {
        let flat: Vec<u8> = points.iter().flat_map(|p| p.as_bytes().iter().copied()).collect();
        check(unsafe { sys::zkp_ctx_prepare_fixed_points(self.0, points.len() as u32, flat.as_ptr()) })
    }
}
impl Drop for Engine {
    fn drop(&mut self) {
        unsafe { sys::zkp_ctx_destroy(self.0) }
    }
}

/// The per-call route: what `prover.rs:94`, `verifier.rs:97`, `verifier.rs:162` and `batch_verifier.rs:219` call.
pub mod multiscalar {
    use super::*;

    fn flat_scalars<'a, I: IntoIterator<Item = &'a Scalar>>(it: I) -> Vec<u8> {
        it.into_iter().flat_map(|s| s.as_bytes().iter().copied()).collect()
    }
    fn flat_points<'a, I: IntoIterator<Item = &'a CompressedRistretto>>(it: I) -> Vec<u8> {
        it.into_iter().flat_map(|p| p.as_bytes().iter().copied()).collect()
    }

    /// `RistrettoPoint::optional_multiscalar_mul(scalars, points.map(|p| p.decompress()))` followed by `.compress()`:
    /// `Ok(None)` = some point failed to decompress (the reference maps it to `VerificationFailure`).
    pub fn optional_multiscalar_mul<'a, S, P>(eng: &Engine, scalars: S, points: P) -> Result<Option<CompressedRistretto>, Error>
    where
        S: IntoIterator<Item = &'a Scalar>,
        P: IntoIterator<Item = &'a CompressedRistretto>,
    {
        let (s, p) = (flat_scalars(scalars), flat_points(points));
        assert_eq!(s.len(), p.len()); // dalek asserts equal lengths too
        let (mut out, mut status) = ([0u8; 32], 1 as c_int);
        check(unsafe { sys::zkp_msm_optional(eng.0, (s.len() / 32) as u64, s.as_ptr(), p.as_ptr(), out.as_mut_ptr(), &mut status) })?;
        Ok(if status == 0 { Some(CompressedRistretto(out)) } else { None })
    }

    /// Many `multiscalar_mul` (constant_time = true, prover.rs:94) / `vartime_multiscalar_mul` (verifier.rs:97) calls at
    /// once, each followed by `.compress()` (mod.rs:204): MSM i multiplies `scalars[off[i]..off[i+1]]` with
    /// `table[pidx[..]]`.  `None` = a referenced point failed to decompress.
    pub fn multiscalar_mul_many(eng: &Engine, off: &[u32], scalars: &[Scalar], pidx: &[u32], table: &[CompressedRistretto],
                                constant_time: bool) -> Result<Vec<Option<CompressedRistretto>>, Error> {
        if off.is_empty() || off[0] != 0 || off.windows(2).any(|w| w[1] < w[0]) {
            return Err(Error::Shape("off must be non-empty, start at 0 and be non-decreasing"));
        }
        let n = off.len() - 1;
        let n_terms = off[n] as usize;
        if scalars.len() != n_terms || pidx.len() != n_terms || pidx.iter().any(|&i| i as usize >= table.len()) {
            return Err(Error::Shape("scalars / pidx must have off[n] entries and index into table"));
        }
        let (s, p) = (flat_scalars(scalars.iter()), flat_points(table.iter()));
        let mut out = vec![0u8; 32 * n];
        let mut status = vec![0u8; n];
        let flags = if constant_time { sys::ZKP_CT } else { sys::ZKP_VARTIME };
        check(unsafe {
            sys::zkp_msm_many(eng.0, n as u32, off.as_ptr(), s.as_ptr(), pidx.as_ptr(), p.as_ptr(), table.len() as u32, flags,
                              out.as_mut_ptr(), status.as_mut_ptr())
        })?;
        Ok((0..n).map(|i| if status[i] == 0 { let mut b = [0u8; 32]; b.copy_from_slice(&out[32 * i..32 * i + 32]); Some(CompressedRistretto(b)) } else { None }).collect())
    }
}

/// A Merlin transcript as the 208-byte state the C ABI works on (merlin keeps its fields private, so the state machine
/// lives on the C side: `zkp_transcript_*` are byte-identical to merlin 2.x -- Merlin's published test vector is a test).
#[derive(Clone)]
pub struct Transcript(pub [u8; sys::ZKP_TRANSCRIPT_BYTES]);
impl Transcript {
    pub fn new(label: &'static [u8]) -> Transcript {
        let mut t = [0u8; sys::ZKP_TRANSCRIPT_BYTES];
        unsafe { sys::zkp_transcript_init(t.as_mut_ptr(), label.as_ptr(), label.len()) };
        Transcript(t)
    }
    /// Panics (like merlin's `encode_usize_as_u32` assert) when the message is longer than u32::MAX bytes: the C side returns
    /// ZKP_TB_TOO_LONG and leaves the transcript untouched.
    pub fn append_message(&mut self, label: &'static [u8], message: &[u8]) {
        let l = CString::new(label).expect("labels are NUL-free");
        let rc = unsafe { sys::zkp_transcript_append_message(self.0.as_mut_ptr(), l.as_ptr(), message.as_ptr(), message.len()) };
        assert_eq!(rc, 0, "zkp_transcript_append_message: code {} (message longer than u32::MAX bytes?)", rc);
    }
    pub fn challenge_bytes(&mut self, label: &'static [u8], dest: &mut [u8]) {
        let l = CString::new(label).expect("labels are NUL-free");
        let rc = unsafe { sys::zkp_transcript_challenge_bytes(self.0.as_mut_ptr(), l.as_ptr(), dest.as_mut_ptr(), dest.len()) };
        assert_eq!(rc, 0, "zkp_transcript_challenge_bytes: code {}", rc);
    }
}

#[derive(Copy, Clone)]
pub struct ScalarVar(pub u32);
#[derive(Copy, Clone)]
pub struct PointVar(pub u32);

/// What `define_proof!` fixes: labels, allocation order (every `allocate_*` call appends to the transcript when it is made),
/// constraints (`SchnorrCS::constrain`, toolbox/mod.rs:86-98).
pub struct Statement(*mut sys::zkp_statement);
impl Statement {
    pub fn new(proof_label: &'static [u8]) -> Statement {
        let l = CString::new(proof_label).unwrap();
        let h = unsafe { sys::zkp_statement_new(l.as_ptr()) };
        assert!(!h.is_null(), "zkp_statement_new returned NULL");
        Statement(h)
    }
    pub fn allocate_scalar(&mut self, label: &'static [u8]) -> ScalarVar {
        let l = CString::new(label).unwrap();
        ScalarVar(unsafe { sys::zkp_statement_add_secret(self.0, l.as_ptr()) } as u32)
    }
    /// `common` = true for define_proof!'s common points / `BatchVerifier::allocate_static_point`.
    pub fn allocate_point(&mut self, label: &'static [u8], common: bool) -> PointVar {
        let l = CString::new(label).unwrap();
        PointVar(unsafe { sys::zkp_statement_add_point(self.0, l.as_ptr(), common as c_int) } as u32)
    }
    pub fn constrain(&mut self, lhs: PointVar, linear_combination: &[(ScalarVar, PointVar)]) {
        let s: Vec<u32> = linear_combination.iter().map(|t| (t.0).0).collect();
        let p: Vec<u32> = linear_combination.iter().map(|t| (t.1).0).collect();
        let rc = unsafe { sys::zkp_statement_constrain(self.0, lhs.0, s.len() as u32, s.as_ptr(), p.as_ptr()) };
        assert_eq!(rc, 0, "constraint refers to an unallocated variable");
    }
    fn m(&self) -> usize { unsafe { sys::zkp_statement_num_secrets(self.0) as usize } }
    fn nc(&self) -> usize { unsafe { sys::zkp_statement_num_constraints(self.0) as usize } }
    fn ni(&self) -> usize { unsafe { sys::zkp_statement_num_instance(self.0) as usize } }
    fn ns(&self) -> usize { unsafe { sys::zkp_statement_num_common(self.0) as usize } }
    /// The C side reads [ni][N][32], [ns][32], [N][m][32], [N][nc][32] straight from the slices: any other length would be an
    /// out-of-bounds read from a safe function, so every length is checked against the statement's shape before the FFI call.
    fn check_shapes(&self, n: usize, inst: usize, common: usize, per_proof: &[(&str, usize, usize)]) -> Result<(), Error> {
        if inst != self.ni() * n || common != self.ns() {
            return Err(Error::Shape("instance / common points do not match the statement and batch size"));
        }
        for (what, len, per) in per_proof {
            if *len != n * per {
                let _ = what;
                return Err(Error::Shape("a per-proof array does not match the statement and batch size"));
            }
        }
        Ok(())
    }
}
impl Drop for Statement {
    fn drop(&mut self) {
        unsafe { sys::zkp_statement_free(self.0) }
    }
}

pub struct Proofs {
    pub challenges: Vec<u8>,  // [N][32]         -> CompactProof.challenge
    pub responses: Vec<u8>,   // [N][m][32]      -> CompactProof.responses / BatchableProof.responses
    pub commitments: Vec<u8>, // [N][nc][32]     -> BatchableProof.commitments
}

/// N x { build_prover ; Prover::prove_impl } (macros.rs:206-258, prover.rs:76-112).  The 32 bytes per proof that
/// `thread_rng()` contributes at prover.rs:82 are drawn inside the library from the operating system (pass them explicitly
/// through the raw call for deterministic tests).
pub fn prove_batch(eng: &Engine, st: &Statement, transcripts: &mut [Transcript], secrets: &[Scalar], inst_points: &[CompressedRistretto],
                   common_points: &[CompressedRistretto]) -> Result<Proofs, Error> {
    let n = transcripts.len();
    st.check_shapes(n, inst_points.len(), common_points.len(), &[("secrets", secrets.len(), st.m())])?;
    let mut ts: Vec<u8> = transcripts.iter().flat_map(|t| t.0.iter().copied()).collect();
    let sec: Vec<u8> = secrets.iter().flat_map(|s| s.as_bytes().iter().copied()).collect();
    let inst: Vec<u8> = inst_points.iter().flat_map(|p| p.as_bytes().iter().copied()).collect();
    let com: Vec<u8> = common_points.iter().flat_map(|p| p.as_bytes().iter().copied()).collect();
    let mut out = Proofs { challenges: vec![0; 32 * n], responses: vec![0; 32 * n * st.m()], commitments: vec![0; 32 * n * st.nc()] };
    check(unsafe {
        sys::zkp_prove_batch(eng.0, st.0, n as u32, ts.as_mut_ptr(), sec.as_ptr(), inst.as_ptr(), com.as_ptr(), ptr::null(), 0,
                             out.challenges.as_mut_ptr(), out.responses.as_mut_ptr(), out.commitments.as_mut_ptr())
    })?;
    for (t, chunk) in transcripts.iter_mut().zip(ts.chunks(sys::ZKP_TRANSCRIPT_BYTES)) {
        t.0.copy_from_slice(chunk);
    }
    Ok(out)
}

/// N x { build_verifier ; Verifier::verify_compact } (macros.rs:280-311, verifier.rs:80-120): one `Result` per proof.
pub fn verify_compact_batch(eng: &Engine, st: &Statement, transcripts: &mut [Transcript], inst_points: &[CompressedRistretto],
                            common_points: &[CompressedRistretto], challenges: &[u8], responses: &[u8]) -> Result<Vec<Result<(), Error>>, Error> {
    let n = transcripts.len();
    st.check_shapes(n, inst_points.len(), common_points.len(), &[("challenges", challenges.len(), 32), ("responses", responses.len(), 32 * st.m())])?;
    let mut ts: Vec<u8> = transcripts.iter().flat_map(|t| t.0.iter().copied()).collect();
    let inst: Vec<u8> = inst_points.iter().flat_map(|p| p.as_bytes().iter().copied()).collect();
    let com: Vec<u8> = common_points.iter().flat_map(|p| p.as_bytes().iter().copied()).collect();
    let mut results = vec![1u8; n];
    check(unsafe {
        sys::zkp_verify_compact_batch(eng.0, st.0, n as u32, ts.as_mut_ptr(), inst.as_ptr(), com.as_ptr(), challenges.as_ptr(),
                                      responses.as_ptr(), 0, results.as_mut_ptr())
    })?;
    for (t, chunk) in transcripts.iter_mut().zip(ts.chunks(sys::ZKP_TRANSCRIPT_BYTES)) {
        t.0.copy_from_slice(chunk);
    }
    Ok(results.into_iter().map(|r| if r == 0 { Ok(()) } else { Err(Error::VerificationFailure) }).collect())
}

/// `BatchVerifier::verify_batchable` (batch_verifier.rs:67-235): one verdict for the whole batch.  The u128 factors of
/// batch_verifier.rs:179 are drawn inside the library from a ChaCha20 stream keyed by the operating system.
pub fn batch_verify(eng: &Engine, st: &Statement, transcripts: &mut [Transcript], inst_points: &[CompressedRistretto],
                    common_points: &[CompressedRistretto], commitments: &[u8], responses: &[u8]) -> Result<(), Error> {
    let n = (commitments.len() / 32).checked_div(st.nc()).unwrap_or(transcripts.len());
    // (a transcript count different from n is the reference's BatchSizeMismatch, reported by the C side: batch_verifier.rs:72-74)
    st.check_shapes(n, inst_points.len(), common_points.len(), &[("commitments", commitments.len(), 32 * st.nc()), ("responses", responses.len(), 32 * st.m())])?;
    let mut ts: Vec<u8> = transcripts.iter().flat_map(|t| t.0.iter().copied()).collect();
    let inst: Vec<u8> = inst_points.iter().flat_map(|p| p.as_bytes().iter().copied()).collect();
    let com: Vec<u8> = common_points.iter().flat_map(|p| p.as_bytes().iter().copied()).collect();
    let rc = unsafe {
        sys::zkp_batch_verify(eng.0, st.0, n as u32, transcripts.len() as u32, ts.as_mut_ptr(), inst.as_ptr(), com.as_ptr(),
                              commitments.as_ptr(), responses.as_ptr(), ptr::null(), 0)
    };
    for (t, chunk) in transcripts.iter_mut().zip(ts.chunks(sys::ZKP_TRANSCRIPT_BYTES)) {
        t.0.copy_from_slice(chunk);
    }
    check(rc)
}

/// N x { build_verifier ; Verifier::verify_batchable } (verifier.rs:123-173): one `Result` per proof -- what a caller runs after
/// `batch_verify` failed and it has to know WHICH proofs are bad.  The per-proof u128 factors of verifier.rs:153 are drawn inside
/// the library.
pub fn verify_batchable_each(eng: &Engine, st: &Statement, transcripts: &mut [Transcript], inst_points: &[CompressedRistretto],
                             common_points: &[CompressedRistretto], commitments: &[u8], responses: &[u8]) -> Result<Vec<Result<(), Error>>, Error> {
    let n = transcripts.len();
    st.check_shapes(n, inst_points.len(), common_points.len(), &[("commitments", commitments.len(), 32 * st.nc()), ("responses", responses.len(), 32 * st.m())])?;
    let mut ts: Vec<u8> = transcripts.iter().flat_map(|t| t.0.iter().copied()).collect();
    let inst: Vec<u8> = inst_points.iter().flat_map(|p| p.as_bytes().iter().copied()).collect();
    let com: Vec<u8> = common_points.iter().flat_map(|p| p.as_bytes().iter().copied()).collect();
    let mut results = vec![1u8; n];
    check(unsafe {
        sys::zkp_verify_batchable_each(eng.0, st.0, n as u32, ts.as_mut_ptr(), inst.as_ptr(), com.as_ptr(), commitments.as_ptr(),
                                       responses.as_ptr(), ptr::null(), 0, results.as_mut_ptr())
    })?;
    for (t, chunk) in transcripts.iter_mut().zip(ts.chunks(sys::ZKP_TRANSCRIPT_BYTES)) {
        t.0.copy_from_slice(chunk);
    }
    Ok(results.into_iter().map(|r| if r == 0 { Ok(()) } else { Err(Error::VerificationFailure) }).collect())
}

/// K independent `BatchVerifier::verify_batchable` runs (batch_verifier.rs:137-235 each) over K * n_each proofs that lie next to
/// each other, in one pass over the GPU: one verdict per batch, each with its own weights and static-coefficient sums.
/// `inst_points` is [ni][K * n_each], everything else per proof in batch order.
pub fn batch_verify_many(eng: &Engine, st: &Statement, n_batches: usize, transcripts: &mut [Transcript], inst_points: &[CompressedRistretto],
                         common_points: &[CompressedRistretto], commitments: &[u8], responses: &[u8]) -> Result<Vec<Result<(), Error>>, Error> {
    let n = transcripts.len();
    if n_batches == 0 || n % n_batches != 0 {
        return Err(Error::Shape("the transcripts do not split into n_batches equal batches"));
    }
    st.check_shapes(n, inst_points.len(), common_points.len(), &[("commitments", commitments.len(), 32 * st.nc()), ("responses", responses.len(), 32 * st.m())])?;
    let mut ts: Vec<u8> = transcripts.iter().flat_map(|t| t.0.iter().copied()).collect();
    let inst: Vec<u8> = inst_points.iter().flat_map(|p| p.as_bytes().iter().copied()).collect();
    let com: Vec<u8> = common_points.iter().flat_map(|p| p.as_bytes().iter().copied()).collect();
    let mut verdicts = vec![1 as std::os::raw::c_int; n_batches];
    let rc = unsafe {
        sys::zkp_batch_verify_many(eng.0, st.0, n_batches as u32, (n / n_batches) as u32, n as u32, ts.as_mut_ptr(), inst.as_ptr(), com.as_ptr(),
                                   commitments.as_ptr(), responses.as_ptr(), ptr::null(), 0, verdicts.as_mut_ptr())
    };
    for (t, chunk) in transcripts.iter_mut().zip(ts.chunks(sys::ZKP_TRANSCRIPT_BYTES)) {
        t.0.copy_from_slice(chunk);
    }
    check(rc)?; // 0 = every verdict was computed; BatchSizeMismatch / negative = none was
    Ok(verdicts.into_iter().map(|v| if v == 0 { Ok(()) } else { Err(Error::VerificationFailure) }).collect())
}

/// `batch_verify`, and -- only when the batch check fails -- the per-proof verdicts of `verify_batchable_each` in the same call
/// (batch_verifier.rs:233 can only say that SOME proof is wrong).  `Ok(Ok(()))`: the batch verified; `Ok(Err(bad))`: it did not, and
/// `bad[j]` says whether proof j failed on its own (all `false` is possible: fail closed on the batch verdict, never on `bad`).
pub fn batch_verify_locate(eng: &Engine, st: &Statement, transcripts: &mut [Transcript], inst_points: &[CompressedRistretto],
                           common_points: &[CompressedRistretto], commitments: &[u8], responses: &[u8]) -> Result<Result<(), Vec<bool>>, Error> {
    let n = transcripts.len();
    st.check_shapes(n, inst_points.len(), common_points.len(), &[("commitments", commitments.len(), 32 * st.nc()), ("responses", responses.len(), 32 * st.m())])?;
    let mut ts: Vec<u8> = transcripts.iter().flat_map(|t| t.0.iter().copied()).collect();
    let inst: Vec<u8> = inst_points.iter().flat_map(|p| p.as_bytes().iter().copied()).collect();
    let com: Vec<u8> = common_points.iter().flat_map(|p| p.as_bytes().iter().copied()).collect();
    let mut results = vec![0u8; n];
    let rc = unsafe {
        sys::zkp_batch_verify_locate(eng.0, st.0, n as u32, n as u32, ts.as_mut_ptr(), inst.as_ptr(), com.as_ptr(), commitments.as_ptr(),
                                     responses.as_ptr(), ptr::null(), 0, results.as_mut_ptr())
    };
    for (t, chunk) in transcripts.iter_mut().zip(ts.chunks(sys::ZKP_TRANSCRIPT_BYTES)) {
        t.0.copy_from_slice(chunk);
    }
    match rc {
        0 => Ok(Ok(())),
        sys::ZKP_TB_VERIFICATION_FAILURE => Ok(Err(results.into_iter().map(|r| r != 0).collect())),
        _ => check(rc).map(|_| Ok(())),
    }
}

/// `zkp_pipe` (include/zkp_toolbox.h): `contexts_per_device` engine contexts on each listed GPU.  One process drives the GPUs of a
/// node: no process group, no collective -- the only cross-GPU exchange of this path is the AND of verdict bits, taken on the host.
/// Not `Sync`: a pipe and its jobs belong to one thread at a time.
pub struct Pipe(*mut sys::zkp_pipe);
impl Pipe {
    pub fn new(device_ids: &[i32], contexts_per_device: usize) -> Result<Pipe, Error> {
        let mut p = ptr::null_mut();
        check(unsafe { sys::zkp_pipe_create(&mut p, device_ids.as_ptr(), device_ids.len() as c_int, contexts_per_device as c_int) })?;
        Ok(Pipe(p))
    }
    pub fn jobs_in_flight(&self) -> usize { unsafe { sys::zkp_pipe_jobs_in_flight(self.0) as usize } }
    /// Asynchronous submits on one host thread per entry of the device list (`Some(true)`), on the caller's thread (`Some(false)`), or the
    /// default (`None`: threads when the pipe spans more than one entry).  Only while no job is in flight.  With threads, errors the submit
    /// itself would return arrive from `Job::wait`; the `Statement` is borrowed by the job for that reason.
    pub fn set_submit_threads(&self, on: Option<bool>) -> Result<(), Error> {
        let rc = unsafe { sys::zkp_pipe_set_submit_threads(self.0, match on { None => -1, Some(false) => 0, Some(true) => 1 }) };
        if rc != 0 { return Err(self.err(rc)); }
        Ok(())
    }
    fn err(&self, rc: c_int) -> Error {
        match rc {
            sys::ZKP_TB_VERIFICATION_FAILURE => Error::VerificationFailure,
            sys::ZKP_TB_BATCH_SIZE_MISMATCH => Error::BatchSizeMismatch,
            rc => Error::Backend(rc, unsafe { CStr::from_ptr(sys::zkp_pipe_last_error(self.0)) }.to_string_lossy().into_owned()),
        }
    }

    /// `prove_batch` over every context: contiguous proof ranges, one host thread per listed GPU; the bytes of the single-context call.
    pub fn prove_batch(&self, st: &Statement, transcripts: &mut [Transcript], secrets: &[Scalar], inst_points: &[CompressedRistretto],
                       common_points: &[CompressedRistretto]) -> Result<Proofs, Error> {
        let n = transcripts.len();
        st.check_shapes(n, inst_points.len(), common_points.len(), &[("secrets", secrets.len(), st.m())])?;
        let mut ts: Vec<u8> = transcripts.iter().flat_map(|t| t.0.iter().copied()).collect();
        let sec: Vec<u8> = secrets.iter().flat_map(|s| s.as_bytes().iter().copied()).collect();
        let inst: Vec<u8> = inst_points.iter().flat_map(|p| p.as_bytes().iter().copied()).collect();
        let com: Vec<u8> = common_points.iter().flat_map(|p| p.as_bytes().iter().copied()).collect();
        let mut out = Proofs { challenges: vec![0; 32 * n], responses: vec![0; 32 * n * st.m()], commitments: vec![0; 32 * n * st.nc()] };
        let rc = unsafe {
            sys::zkp_pipe_prove_batch(self.0, st.0, n as u32, ts.as_mut_ptr(), sec.as_ptr(), inst.as_ptr(), com.as_ptr(), ptr::null(),
                                      out.challenges.as_mut_ptr(), out.responses.as_mut_ptr(), out.commitments.as_mut_ptr())
        };
        if rc != 0 { return Err(self.err(rc)); }
        for (t, chunk) in transcripts.iter_mut().zip(ts.chunks(sys::ZKP_TRANSCRIPT_BYTES)) { t.0.copy_from_slice(chunk); }
        Ok(out)
    }

    /// `BatchVerifier::verify_batchable` over every context: each contiguous range is a batch check of its own (own weights, own sums of
    /// the static coefficients, batch_verifier.rs:173-206 per range); `Ok(())` iff every range verifies.
    pub fn batch_verify(&self, st: &Statement, transcripts: &mut [Transcript], inst_points: &[CompressedRistretto],
                        common_points: &[CompressedRistretto], commitments: &[u8], responses: &[u8]) -> Result<(), Error> {
        let n = (commitments.len() / 32).checked_div(st.nc()).unwrap_or(transcripts.len());
        st.check_shapes(n, inst_points.len(), common_points.len(), &[("commitments", commitments.len(), 32 * st.nc()), ("responses", responses.len(), 32 * st.m())])?;
        let mut ts: Vec<u8> = transcripts.iter().flat_map(|t| t.0.iter().copied()).collect();
        let inst: Vec<u8> = inst_points.iter().flat_map(|p| p.as_bytes().iter().copied()).collect();
        let com: Vec<u8> = common_points.iter().flat_map(|p| p.as_bytes().iter().copied()).collect();
        let rc = unsafe {
            sys::zkp_pipe_batch_verify(self.0, st.0, n as u32, transcripts.len() as u32, ts.as_mut_ptr(), inst.as_ptr(), com.as_ptr(),
                                       commitments.as_ptr(), responses.as_ptr(), ptr::null())
        };
        for (t, chunk) in transcripts.iter_mut().zip(ts.chunks(sys::ZKP_TRANSCRIPT_BYTES)) { t.0.copy_from_slice(chunk); }
        if rc != 0 { Err(self.err(rc)) } else { Ok(()) }
    }

    /// Asynchronous `prove_batch`: every proof starts from `start` (the reference's callers write `Transcript::new(label)` per proof); the
    /// per-proof entropy of prover.rs:82 is a ChaCha20 stream keyed from the OS and expanded on the device.  The buffers are borrowed until
    /// [`Job::wait`]; `Err(PipeFull)`-style back-pressure is `Error::Backend(ZKP_TB_PIPE_FULL, ..)`: wait for the oldest job first.
    pub fn submit_prove<'a>(&'a self, st: &'a Statement, n: usize, start: &'a Transcript, secrets: &'a [u8], inst_points: &'a [u8], common_points: &'a [u8],
                            out: &'a mut Proofs) -> Result<Job<'a>, Error> {
        st.check_shapes(n, inst_points.len() / 32, common_points.len() / 32, &[("secrets", secrets.len(), 32 * st.m())])?;
        if out.challenges.len() != 32 * n || out.responses.len() != 32 * n * st.m() || out.commitments.len() != 32 * n * st.nc() {
            return Err(Error::Shape("output buffers do not match the statement and batch size"));
        }
        let mut job = ptr::null_mut();
        let rc = unsafe {
            sys::zkp_prove_batch_submit(self.0, st.0, n as u32, sys::ZKP_JOB_SHARED_TRANSCRIPT, start.0.as_ptr(), secrets.as_ptr(), inst_points.as_ptr(), n as u32,
                                        common_points.as_ptr(), ptr::null(), ptr::null_mut(), out.challenges.as_mut_ptr(), out.responses.as_mut_ptr(),
                                        out.commitments.as_mut_ptr(), &mut job)
        };
        if rc != 0 { return Err(self.err(rc)); }
        Ok(Job { job, pipe: self, verdicts: None })
    }

    /// Asynchronous K batch verifications (K x batch_verifier.rs:67-235) over proofs lying next to each other; weights drawn on the device.
    pub fn submit_batch_verify_many<'a>(&'a self, st: &'a Statement, n_batches: usize, n_each: usize, start: &'a Transcript, inst_points: &'a [u8],
                                        common_points: &'a [u8], commitments: &'a [u8], responses: &'a [u8]) -> Result<Job<'a>, Error> {
        let n = n_batches * n_each;
        st.check_shapes(n, inst_points.len() / 32, common_points.len() / 32, &[("commitments", commitments.len(), 32 * st.nc()), ("responses", responses.len(), 32 * st.m())])?;
        let mut verdicts = vec![1 as c_int; n_batches].into_boxed_slice();      // boxed: the C side keeps the address until the job is waited for
        let mut job = ptr::null_mut();
        let rc = unsafe {
            sys::zkp_batch_verify_many_submit(self.0, st.0, n_batches as u32, n_each as u32, sys::ZKP_JOB_SHARED_TRANSCRIPT, start.0.as_ptr(), inst_points.as_ptr(),
                                              n as u32, common_points.as_ptr(), commitments.as_ptr(), responses.as_ptr(), ptr::null(), n as u32, ptr::null_mut(),
                                              verdicts.as_mut_ptr(), &mut job)
        };
        if rc != 0 { return Err(self.err(rc)); }
        Ok(Job { job, pipe: self, verdicts: Some(verdicts) })
    }
}
impl Drop for Pipe {
    fn drop(&mut self) {
        unsafe { sys::zkp_pipe_destroy(self.0) }      // (every Job<'_> borrows the pipe: none is alive here; a forgotten one is discarded, not written)
    }
}

/// A submitted call.  The C side holds pointers into the job's buffers -- the borrowed inputs and outputs, and the boxed `verdicts` --
/// until `zkp_job_wait` has returned, so a job that is dropped without `wait()` waits in `Drop` (its result is discarded): no pointer the
/// library holds ever outlives the memory it names.  `mem::forget`-ing a job is NOT covered by that: the box leaks, the context stays busy, and with
/// submitter threads the device's thread may still retire the job on its own -- copy staged outputs, write verdicts -- into buffers whose borrow has
/// ended.  `zkp_pipe_destroy` discards what is still pending (kernels waited for, nothing written), which bounds the damage to the pipe's lifetime;
/// do not forget jobs.
pub struct Job<'a> {
    job: *mut sys::zkp_job,
    pipe: &'a Pipe,
    verdicts: Option<Box<[c_int]>>,
}
impl<'a> Job<'a> {
    pub fn done(&self) -> bool { unsafe { sys::zkp_job_done(self.job) != 0 } }
    /// What the synchronous call would have returned; for a batch-verification job the per-batch verdicts.
    pub fn wait(mut self) -> Result<Vec<Result<(), Error>>, Error> {
        let rc = unsafe { sys::zkp_job_wait(self.job) };
        self.job = ptr::null_mut();
        if rc != 0 { return Err(self.pipe.err(rc)); }      // negative: the job failed closed, nothing it wrote may be used
        Ok(self.verdicts.take().map(|v| v.iter().map(|&x| if x == 0 { Ok(()) } else { Err(Error::VerificationFailure) }).collect()).unwrap_or_default())
    }
}
impl<'a> Drop for Job<'a> {
    fn drop(&mut self) {
        if !self.job.is_null() {
            unsafe { sys::zkp_job_wait(self.job) };        // buffers and `verdicts` are still alive here; the verdicts are dropped unread
            self.job = ptr::null_mut();
        }
    }
}

/// `bincode::serialize(&CompactProof)` / `deserialize` (tests/zkp.rs:53-54) through the C codec.
pub fn compact_proof_to_bytes(challenge: &Scalar, responses: &[Scalar]) -> Vec<u8> {
    let r: Vec<u8> = responses.iter().flat_map(|s| s.as_bytes().iter().copied()).collect();
    let mut out = vec![0u8; unsafe { sys::zkp_proof_compact_size(responses.len() as u32) }];
    let rc = unsafe { sys::zkp_proof_compact_encode(challenge.as_bytes().as_ptr(), r.as_ptr(), responses.len() as u32, out.as_mut_ptr(), out.len()) };
    assert_eq!(rc, 0);
    out
}
pub fn compact_proof_from_bytes(bytes: &[u8]) -> Option<(Scalar, Vec<Scalar>)> {
    let cap = bytes.len() / 32 + 1;
    let (mut c, mut r, mut m, mut used) = ([0u8; 32], vec![0u8; 32 * cap], 0u32, 0usize);
    let rc = unsafe { sys::zkp_proof_compact_decode(bytes.as_ptr(), bytes.len(), c.as_mut_ptr(), r.as_mut_ptr(), cap as u32, &mut m, &mut used) };
    if rc != 0 {
        return None; // truncated, or a scalar was not canonical (dalek's Deserialize refuses it too)
    }
    let to_scalar = |b: &[u8]| { let mut a = [0u8; 32]; a.copy_from_slice(b); Scalar::from_canonical_bytes(a) };
    let resp: Option<Vec<Scalar>> = r[..32 * m as usize].chunks(32).map(to_scalar).collect();
    Some((to_scalar(&c)?, resp?))
}
