"""Python binding of the host toolbox (include/zkp_toolbox.h -> libzkp_toolbox.so), shaped like the
reference's API so tests read like the reference's tests:

    reference (Rust)                                   here
    -------------------------------------------------  ----------------------------------------------
    Transcript::new(b"..")                             Transcript(b"..")
    toolbox::prover::Prover::new(label, &mut t)        Prover(label, t, engine)
    prover.allocate_scalar / allocate_point / constrain / prove_compact / prove_batchable
    toolbox::verifier::Verifier                        Verifier(label, t, engine)
    toolbox::batch_verifier::BatchVerifier             BatchVerifier(label, n, transcripts, engine)
    define_proof! { name, "label", (secrets), (instance), (common) : constraints }
                                                       define_proof(name, label, secrets, instance, common, constraints)

All arithmetic happens in the C++ host library and, below it, in the HIP library; this file only
marshals buffers.  Points are 32-byte ristretto255 encodings, scalars 32-byte little-endian strings
(ints are accepted and reduced mod l for convenience).
"""
from __future__ import annotations

import ctypes
import weakref
import os
from dataclasses import dataclass
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

from .engine import Engine, load_library

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libzkp_toolbox.so")
TRANSCRIPT_BYTES = 208
L = 2**252 + 27742317777372353535851937790883648493

EXPORTS = (
    "zkp_transcript_init", "zkp_transcript_append_message", "zkp_transcript_challenge_bytes", "zkp_scalar_from_wide",
    "zkp_scalar_muladd", "zkp_scalar_neg", "zkp_statement_new", "zkp_statement_free", "zkp_statement_add_secret",
    "zkp_statement_add_point", "zkp_statement_constrain", "zkp_statement_num_secrets", "zkp_statement_num_instance",
    "zkp_statement_num_common", "zkp_statement_num_constraints", "zkp_statement_num_terms", "zkp_prove_batch",
    "zkp_verify_compact_batch", "zkp_verify_batchable_each", "zkp_batch_verify", "zkp_batch_verify_coeffs", "zkp_batch_verify_build",
    "zkp_prove_phase_a", "zkp_prove_phase_b", "zkp_toolbox_set_fused_min_batch", "zkp_toolbox_get_fused_min_batch", "zkp_chacha20_block",
    "zkp_proof_compact_size", "zkp_proof_batchable_size", "zkp_proof_compact_encode", "zkp_proof_compact_decode",
    "zkp_proof_batchable_encode", "zkp_proof_batchable_decode", "zkp_batch_verify_locate", "zkp_batch_verify_many",
    "zkp_pipe_create", "zkp_pipe_destroy", "zkp_pipe_num_contexts", "zkp_pipe_num_devices", "zkp_pipe_context", "zkp_pipe_context_device", "zkp_pipe_shard_plan",
    "zkp_pipe_jobs_in_flight", "zkp_pipe_set_submit_threads", "zkp_pipe_last_error", "zkp_prove_batch_submit", "zkp_verify_compact_batch_submit",
    "zkp_verify_batchable_each_submit", "zkp_batch_verify_many_submit", "zkp_job_done", "zkp_job_wait", "zkp_job_context_index", "zkp_pipe_prove_batch",
    "zkp_pipe_verify_compact_batch", "zkp_pipe_verify_batchable_each", "zkp_pipe_batch_verify", "zkp_pipe_batch_verify_many",
    "zkp_pipe_batch_verify_locate", "zkp_toolbox_set_host_max_terms", "zkp_toolbox_get_host_max_terms",
)
ZKP_JOB_SHARED_TRANSCRIPT = 1
ZKP_TB_PIPE_FULL = 3


class ProofError(Exception):
    """src/errors.rs"""


class VerificationFailure(ProofError):
    pass


class BatchSizeMismatch(ProofError):
    pass


_lib = None


def lib() -> ctypes.CDLL:
    global _lib
    if _lib is None:
        load_library()          # libzkp_mi355x.so must exist: no CPU fallback
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(f"{LIB_PATH} is missing: run __graft_entry__.build()")
        _lib = ctypes.CDLL(LIB_PATH)
        _lib.zkp_statement_new.restype = ctypes.c_void_p
        _lib.zkp_statement_new.argtypes = [ctypes.c_char_p]
        _lib.zkp_statement_free.argtypes = [ctypes.c_void_p]
        _lib.zkp_statement_free.restype = None
        _lib.zkp_statement_add_secret.argtypes = [ctypes.c_void_p, ctypes.c_char_p]
        _lib.zkp_statement_add_point.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_int]
        _lib.zkp_statement_constrain.argtypes = [ctypes.c_void_p, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_void_p, ctypes.c_void_p]
        for f in ("num_secrets", "num_instance", "num_common", "num_constraints", "num_terms"):
            getattr(_lib, "zkp_statement_" + f).argtypes = [ctypes.c_void_p]
            getattr(_lib, "zkp_statement_" + f).restype = ctypes.c_uint32
        vp, u32, sz = ctypes.c_void_p, ctypes.c_uint32, ctypes.c_size_t
        _lib.zkp_proof_compact_size.restype = sz
        _lib.zkp_proof_compact_size.argtypes = [u32]
        _lib.zkp_proof_batchable_size.restype = sz
        _lib.zkp_proof_batchable_size.argtypes = [u32, u32]
        _lib.zkp_proof_compact_encode.argtypes = [ctypes.c_char_p, vp, u32, ctypes.c_char_p, sz]
        _lib.zkp_proof_compact_decode.argtypes = [ctypes.c_char_p, sz, vp, vp, u32, ctypes.POINTER(u32), ctypes.POINTER(sz)]
        _lib.zkp_proof_batchable_encode.argtypes = [vp, u32, vp, u32, ctypes.c_char_p, sz]
        _lib.zkp_proof_batchable_decode.argtypes = [ctypes.c_char_p, sz, vp, u32, ctypes.POINTER(u32), vp, u32, ctypes.POINTER(u32), ctypes.POINTER(sz)]
        i32 = ctypes.c_int
        _lib.zkp_pipe_create.argtypes = [ctypes.POINTER(vp), ctypes.POINTER(i32), i32, i32]
        _lib.zkp_pipe_destroy.argtypes = [vp]
        _lib.zkp_pipe_destroy.restype = None
        for f in ("num_contexts", "num_devices", "jobs_in_flight"):
            getattr(_lib, "zkp_pipe_" + f).argtypes = [vp]
        _lib.zkp_pipe_context.argtypes = [vp, i32]
        _lib.zkp_pipe_context.restype = vp
        _lib.zkp_pipe_context_device.argtypes = [vp, i32]
        _lib.zkp_pipe_shard_plan.argtypes = [u32, u32, u32, u32, vp, vp]
        _lib.zkp_pipe_shard_plan.restype = u32
        _lib.zkp_pipe_set_submit_threads.argtypes = [vp, i32]
        _lib.zkp_pipe_last_error.argtypes = [vp]
        _lib.zkp_pipe_last_error.restype = ctypes.c_char_p
        pj = ctypes.POINTER(vp)
        _lib.zkp_prove_batch_submit.argtypes = [vp, vp, u32, u32, vp, vp, vp, u32, vp, vp, vp, vp, vp, vp, pj]
        _lib.zkp_verify_compact_batch_submit.argtypes = [vp, vp, u32, u32, vp, vp, u32, vp, vp, vp, vp, vp, pj]
        _lib.zkp_verify_batchable_each_submit.argtypes = [vp, vp, u32, u32, vp, vp, u32, vp, vp, vp, vp, vp, vp, pj]
        _lib.zkp_batch_verify_many_submit.argtypes = [vp, vp, u32, u32, u32, vp, vp, u32, vp, vp, vp, vp, u32, vp, vp, pj]
        _lib.zkp_job_done.argtypes = [vp]
        _lib.zkp_job_context_index.argtypes = [vp]
        _lib.zkp_job_wait.argtypes = [vp]
        _lib.zkp_pipe_prove_batch.argtypes = [vp, vp, u32, vp, vp, vp, vp, vp, vp, vp, vp]
        _lib.zkp_pipe_verify_compact_batch.argtypes = [vp, vp, u32, vp, vp, vp, vp, vp, vp]
        _lib.zkp_pipe_verify_batchable_each.argtypes = [vp, vp, u32, vp, vp, vp, vp, vp, vp, vp]
        _lib.zkp_pipe_batch_verify.argtypes = [vp, vp, u32, u32, vp, vp, vp, vp, vp, vp]
        _lib.zkp_pipe_batch_verify_many.argtypes = [vp, vp, u32, u32, u32, vp, vp, vp, vp, vp, vp, vp]
        _lib.zkp_pipe_batch_verify_locate.argtypes = [vp, vp, u32, u32, vp, vp, vp, vp, vp, vp, vp]
    return _lib


def set_fused_min_batch(n: int) -> None:
    """Batches of >= n proofs run transcripts, scalars and MSMs on the device; smaller ones hash on the host threads."""
    lib().zkp_toolbox_set_fused_min_batch(ctypes.c_uint32(n))


def set_host_max_terms(n: int) -> None:
    """Calls of at most n (scalar, point) terms run on the host backend even with a GPU context (0 = never); ctx = None always does."""
    lib().zkp_toolbox_set_host_max_terms(ctypes.c_uint32(n))


def get_host_max_terms() -> int:
    f = lib().zkp_toolbox_get_host_max_terms
    f.restype = ctypes.c_uint32
    return int(f())


class HostEngine:
    """Stands where an Engine stands in the calls of this module and selects the host backend (ctx == NULL in the C ABI): no GPU needed."""
    _h = None

    def close(self):
        pass


def get_fused_min_batch() -> int:
    f = lib().zkp_toolbox_get_fused_min_batch
    f.restype = ctypes.c_uint32
    return int(f())


def pipe_shard_plan(n_items: int, n_contexts: int, unit: int = 1, fused_min_batch: Optional[int] = None):
    """The contiguous ranges zkp_pipe_prove_batch / zkp_pipe_batch_verify[_many] ... cut n_items proofs (batches of `unit` proofs) into, one per
    context: [(lo, hi)] for contexts 0 .. G - 1 (zkp_pipe_shard_plan; plain arithmetic, no GPU)."""
    lo = np.zeros(max(1, n_contexts), np.uint32)
    hi = np.zeros(max(1, n_contexts), np.uint32)
    g = lib().zkp_pipe_shard_plan(n_items, unit, n_contexts, get_fused_min_batch() if fused_min_batch is None else fused_min_batch, _p(lo), _p(hi))
    return [(int(lo[i]), int(hi[i])) for i in range(int(g))]


def _p(a):
    return None if a is None else a.ctypes.data_as(ctypes.c_void_p)


def scalar_bytes(x) -> bytes:
    if isinstance(x, (bytes, bytearray)):
        assert len(x) == 32
        return bytes(x)
    if isinstance(x, np.ndarray):
        return x.tobytes()
    return (int(x) % L).to_bytes(32, "little")


def _raise(rc: int, what: str):
    if rc == 1:
        raise VerificationFailure()
    if rc == 2:
        raise BatchSizeMismatch()
    if rc != 0:
        from .engine import ZkpError
        raise ZkpError(f"{what} failed with code {rc}: {load_library().zkp_last_error().decode()}")


# ---------------------------------------------------------------------------------------------------
class Transcript:
    """merlin::Transcript (plain 208-byte state; clone() is a copy)."""

    def __init__(self, label: bytes = b"", _state: Optional[np.ndarray] = None):
        if _state is not None:
            self.state = _state.copy()
        else:
            self.state = np.zeros(TRANSCRIPT_BYTES, np.uint8)
            lib().zkp_transcript_init(_p(self.state), label, ctypes.c_size_t(len(label)))

    def clone(self) -> "Transcript":
        return Transcript(_state=self.state)

    def append_message(self, label: bytes, message: bytes) -> None:
        rc = lib().zkp_transcript_append_message(_p(self.state), label, message, ctypes.c_size_t(len(message)))
        if rc != 0:
            raise ValueError("zkp_transcript_append_message: code %d (messages and labels are limited to 2^32 - 1 bytes, as in merlin)" % rc)

    def challenge_bytes(self, label: bytes, n: int) -> bytes:
        out = ctypes.create_string_buffer(n)
        rc = lib().zkp_transcript_challenge_bytes(_p(self.state), label, out, ctypes.c_size_t(n))
        if rc != 0:
            raise ValueError("zkp_transcript_challenge_bytes: code %d" % rc)
        return out.raw


def _scalar_canonical(b: bytes) -> bool:
    return len(b) == 32 and int.from_bytes(b, "little") < L


def _wire_decode(kind: str, buf: bytes, allow_trailing: bool):
    """The C codec of include/zkp_toolbox.h (zkp_proof_*_decode): -> (challenge | commitments, responses)."""
    buf = bytes(buf)
    cap = len(buf) // 32 + 1
    a = np.zeros((cap, 32), np.uint8)
    r = np.zeros((cap, 32), np.uint8)
    na, nr, used = ctypes.c_uint32(0), ctypes.c_uint32(0), ctypes.c_size_t(0)
    if kind == "compact":
        rc = lib().zkp_proof_compact_decode(buf, ctypes.c_size_t(len(buf)), _p(a), _p(r), ctypes.c_uint32(cap), ctypes.byref(nr), ctypes.byref(used))
    else:
        rc = lib().zkp_proof_batchable_decode(buf, ctypes.c_size_t(len(buf)), _p(a), ctypes.c_uint32(cap), ctypes.byref(na), _p(r), ctypes.c_uint32(cap),
                                              ctypes.byref(nr), ctypes.byref(used))
    if rc != 0:
        raise ValueError("malformed proof (truncated, or a scalar was not canonically encoded): code %d" % rc)
    if not allow_trailing and used.value != len(buf):
        raise ValueError("trailing bytes")
    return ([x.tobytes() for x in a[:na.value]] if kind != "compact" else a[0].tobytes()), [x.tobytes() for x in r[:nr.value]]


@dataclass
class CompactProof:            # src/proofs.rs:15-20
    challenge: bytes
    responses: List[bytes]

    # Wire format = what `bincode::serialize` (bincode 1.x top-level functions: fixed-width little-endian integers, u64
    # sequence lengths) produces for the serde-derived struct (proofs.rs:14): Scalar serialises as a 32-byte tuple (no
    # length), Vec<Scalar> as u64 length + elements.  The codec itself is C (include/zkp_toolbox.h, zkp_proof_*): this
    # class only marshals.  The reference's tests only round-trip it (tests/zkp.rs:53-54), so there are no golden bytes
    # to pin; dalek rejects non-canonical scalars on deserialisation, and so does the decoder.  allow_trailing = True is
    # bincode's own behaviour (its top-level deserialize ignores what follows the value); the default here is strict.
    def to_bytes(self) -> bytes:
        m = len(self.responses)
        out = ctypes.create_string_buffer(lib().zkp_proof_compact_size(ctypes.c_uint32(m)))
        resp = np.frombuffer(b"".join(self.responses), np.uint8) if m else np.zeros(0, np.uint8)
        rc = lib().zkp_proof_compact_encode(bytes(self.challenge), _p(resp), ctypes.c_uint32(m), out, ctypes.c_size_t(len(out)))
        if rc != 0:
            raise ValueError("zkp_proof_compact_encode: code %d" % rc)
        return out.raw

    @classmethod
    def from_bytes(cls, buf: bytes, allow_trailing: bool = False) -> "CompactProof":
        c, r = _wire_decode("compact", buf, allow_trailing)
        return cls(c, r)


@dataclass
class BatchableProof:          # src/proofs.rs:27-32
    commitments: List[bytes]
    responses: List[bytes]

    def to_bytes(self) -> bytes:
        nc, m = len(self.commitments), len(self.responses)
        out = ctypes.create_string_buffer(lib().zkp_proof_batchable_size(ctypes.c_uint32(nc), ctypes.c_uint32(m)))
        coms = np.frombuffer(b"".join(self.commitments), np.uint8) if nc else np.zeros(0, np.uint8)
        resp = np.frombuffer(b"".join(self.responses), np.uint8) if m else np.zeros(0, np.uint8)
        rc = lib().zkp_proof_batchable_encode(_p(coms), ctypes.c_uint32(nc), _p(resp), ctypes.c_uint32(m), out, ctypes.c_size_t(len(out)))
        if rc != 0:
            raise ValueError("zkp_proof_batchable_encode: code %d" % rc)
        return out.raw

    @classmethod
    def from_bytes(cls, buf: bytes, allow_trailing: bool = False) -> "BatchableProof":
        k, r = _wire_decode("batchable", buf, allow_trailing)
        return cls(k, r)      # CompressedRistretto deserialises from any 32 bytes; validity is checked at decompress


class Statement:
    """Owns a zkp_statement handle.  Variables are registered in allocation order."""

    def __init__(self, proof_label: bytes):
        self.proof_label = proof_label
        self._h = ctypes.c_void_p(lib().zkp_statement_new(proof_label))
        self.secrets: List[bytes] = []
        self.points: List[Tuple[bytes, bool]] = []
        self.constraints: List[Tuple[int, List[Tuple[int, int]]]] = []

    def __del__(self):
        try:
            if self._h:
                lib().zkp_statement_free(self._h)
                self._h = None
        except Exception:
            pass

    def add_secret(self, name: bytes) -> int:
        self.secrets.append(name)
        return lib().zkp_statement_add_secret(self._h, name)

    def add_point(self, name: bytes, common: bool) -> int:
        self.points.append((name, common))
        return lib().zkp_statement_add_point(self._h, name, int(common))

    def constrain(self, lhs: int, lc: Sequence[Tuple[int, int]]) -> None:
        sc = np.array([s for s, _ in lc], dtype=np.uint32)
        pt = np.array([p for _, p in lc], dtype=np.uint32)
        rc = lib().zkp_statement_constrain(self._h, lhs, len(lc), _p(sc), _p(pt))
        if rc != 0:
            raise ValueError("bad constraint")
        self.constraints.append((lhs, list(lc)))

    @property
    def m(self): return len(self.secrets)
    @property
    def ni(self): return sum(1 for _, c in self.points if not c)
    @property
    def ns(self): return sum(1 for _, c in self.points if c)
    @property
    def nc(self): return len(self.constraints)
    @property
    def terms(self): return sum(len(lc) for _, lc in self.constraints)


def _transcripts_array(transcripts: Sequence[Transcript]) -> np.ndarray:
    return np.stack([t.state for t in transcripts]) if transcripts else np.zeros((0, TRANSCRIPT_BYTES), np.uint8)


def _store_transcripts(transcripts: Sequence[Transcript], arr: np.ndarray) -> None:
    for t, row in zip(transcripts, arr):
        t.state[:] = row


# ---- batched entry points (numpy arrays in the layouts of include/zkp_toolbox.h) -------------------
def prove_batch(eng: Engine, st: Statement, transcripts: np.ndarray, secrets: np.ndarray, inst: np.ndarray,
                common: np.ndarray, entropy: Optional[np.ndarray] = None, threads: int = 0, out=None):
    """-> (challenges[N][32], responses[N][m][32], commitments[N][nc][32]); transcripts advanced in place.
    out = (challenges, responses, commitments) to fill (C-contiguous uint8 arrays of those shapes, e.g. from pinned_empty: the engine's copies
    out are then true DMA that leaves under the call's last kernels); default: fresh arrays."""
    n = len(transcripts)
    _check_batch_shapes(st, n, inst, common, None, secrets)
    if entropy is not None and tuple(np.shape(entropy)) != (n, 32):
        raise ValueError("entropy must have shape (%d, 32)" % n)
    if out is None:
        chal = np.zeros((n, 32), np.uint8)
        resp = np.zeros((n, st.m, 32), np.uint8)
        coms = np.zeros((n, st.nc, 32), np.uint8)
    else:
        chal, resp, coms = out
        for name, a, shape in (("challenges", chal, (n, 32)), ("responses", resp, (n, st.m, 32)), ("commitments", coms, (n, st.nc, 32))):
            if tuple(np.shape(a)) != shape or a.dtype != np.uint8 or not a.flags["C_CONTIGUOUS"]:
                raise ValueError("out: %s must be a C-contiguous uint8 array of shape %r" % (name, shape))
    rc = lib().zkp_prove_batch(eng._h, st._h, ctypes.c_uint32(n), _p(transcripts), _p(np.ascontiguousarray(secrets)),
                               _p(np.ascontiguousarray(inst)), _p(np.ascontiguousarray(common)),
                               _p(None if entropy is None else np.ascontiguousarray(entropy)), threads, _p(chal), _p(resp), _p(coms))
    _raise(rc, "zkp_prove_batch")
    return chal, resp, coms


def verify_compact_batch(eng, st, transcripts, inst, common, challenges, responses, threads: int = 0) -> np.ndarray:
    n = len(transcripts)
    _check_batch_shapes(st, n, inst, common, None, responses)
    if tuple(np.shape(challenges)) != (n, 32):
        raise ValueError("challenges must have shape (%d, 32)" % n)
    res = np.ones(n, np.uint8)
    rc = lib().zkp_verify_compact_batch(eng._h, st._h, ctypes.c_uint32(n), _p(transcripts), _p(np.ascontiguousarray(inst)),
                                        _p(np.ascontiguousarray(common)), _p(np.ascontiguousarray(challenges)),
                                        _p(np.ascontiguousarray(responses)), threads, _p(res))
    _raise(rc, "zkp_verify_compact_batch")
    return res


def verify_batchable_each(eng, st, transcripts, inst, common, commitments, responses, weights16=None, threads: int = 0) -> np.ndarray:
    n = len(transcripts)
    _check_batch_shapes(st, n, inst, common, commitments, responses, weights16, per_proof_weights=True)
    res = np.ones(n, np.uint8)
    rc = lib().zkp_verify_batchable_each(eng._h, st._h, ctypes.c_uint32(n), _p(transcripts), _p(np.ascontiguousarray(inst)),
                                         _p(np.ascontiguousarray(common)), _p(np.ascontiguousarray(commitments)),
                                         _p(np.ascontiguousarray(responses)),
                                         _p(None if weights16 is None else np.ascontiguousarray(weights16)), threads, _p(res))
    _raise(rc, "zkp_verify_batchable_each")
    return res


def _check_batch_shapes(st, n, inst, common, commitments, responses, weights16=None, per_proof_weights=False) -> None:
    """The C side reads [ni][N][32], [ns][32], [N][nc][32], [N][m][32] and [nc][N][16] (or [N][nc][16]) straight from these
    buffers: anything else would be an out-of-bounds read, so it is refused here."""
    def want(name, a, shape):
        if a is not None and (tuple(np.shape(a)) != shape or np.asarray(a).dtype != np.uint8):
            raise ValueError("%s must be a uint8 array of shape %r, got %r" % (name, shape, tuple(np.shape(a))))
    want("inst", inst, (st.ni, n, 32))
    want("common", common, (st.ns, 32))
    want("commitments", commitments, (n, st.nc, 32))
    want("responses", responses, (n, st.m, 32))
    want("weights16", weights16, (n, st.nc, 16) if per_proof_weights else (st.nc, n, 16))


def batch_verify(eng, st, transcripts, inst, common, commitments, responses, weights16=None, threads: int = 0,
                 batch_size: Optional[int] = None) -> None:
    """Raises VerificationFailure / BatchSizeMismatch like BatchVerifier::verify_batchable."""
    n = len(commitments)
    _check_batch_shapes(st, n, inst, common, commitments, responses, weights16)
    rc = lib().zkp_batch_verify(eng._h, st._h, ctypes.c_uint32(n if batch_size is None else batch_size),
                                ctypes.c_uint32(len(transcripts)), _p(transcripts),
                                _p(np.ascontiguousarray(inst)), _p(np.ascontiguousarray(common)),
                                _p(np.ascontiguousarray(commitments)), _p(np.ascontiguousarray(responses)),
                                _p(None if weights16 is None else np.ascontiguousarray(weights16)), threads)
    _raise(rc, "zkp_batch_verify")


def batch_verify_many(eng, st, n_batches: int, transcripts, inst, common, commitments, responses, weights16=None, threads: int = 0) -> np.ndarray:
    """zkp_batch_verify_many: n_batches independent BatchVerifier::verify_batchable runs over consecutive ranges of the
    len(commitments) proofs -> verdicts[n_batches] (0 = Ok, 1 = VerificationFailure)."""
    n = len(commitments)
    if n_batches <= 0 or n % n_batches:
        raise ValueError("the number of proofs must be a positive multiple of n_batches")
    _check_batch_shapes(st, n, inst, common, commitments, responses, weights16)
    verdicts = (ctypes.c_int * n_batches)(*([1] * n_batches))
    rc = lib().zkp_batch_verify_many(eng._h, st._h, ctypes.c_uint32(n_batches), ctypes.c_uint32(n // n_batches), ctypes.c_uint32(len(transcripts)),
                                     _p(transcripts), _p(np.ascontiguousarray(inst)), _p(np.ascontiguousarray(common)),
                                     _p(np.ascontiguousarray(commitments)), _p(np.ascontiguousarray(responses)),
                                     _p(None if weights16 is None else np.ascontiguousarray(weights16)), threads, verdicts)
    _raise(rc, "zkp_batch_verify_many")
    return np.array(list(verdicts), np.int32)


def batch_verify_locate(eng, st, transcripts, inst, common, commitments, responses, weights16=None, threads: int = 0):
    """zkp_batch_verify_locate: the batch check and, if it fails, which proofs fail -> (ok, results[N]): ok = the batch
    verified (results then all 0); otherwise results[j] = 1 for the proofs that fail on their own.  A failed batch whose
    per-proof pass finds nothing (the C contract allows it) is still ok = False: never read success from results alone."""
    n = len(commitments)
    _check_batch_shapes(st, n, inst, common, commitments, responses, weights16)
    res = np.ones(n, np.uint8)
    rc = lib().zkp_batch_verify_locate(eng._h, st._h, ctypes.c_uint32(n), ctypes.c_uint32(len(transcripts)), _p(transcripts),
                                       _p(np.ascontiguousarray(inst)), _p(np.ascontiguousarray(common)),
                                       _p(np.ascontiguousarray(commitments)), _p(np.ascontiguousarray(responses)),
                                       _p(None if weights16 is None else np.ascontiguousarray(weights16)), threads, _p(res))
    if rc not in (0, 1):
        _raise(rc, "zkp_batch_verify_locate")
    return rc == 0, res


def batch_verify_coeffs(eng, st, transcripts, inst, common, commitments, responses, weights16, threads: int = 0):
    """batch_verify that also returns the coefficient vector built on the GPU; (ok: bool, coeffs [total][32])."""
    n = len(commitments)
    _check_batch_shapes(st, n, inst, common, commitments, responses, weights16)
    total = st.ns + (st.ni + st.nc) * n
    co = np.zeros((total, 32), np.uint8)
    rc = lib().zkp_batch_verify_coeffs(eng._h, st._h, ctypes.c_uint32(n), ctypes.c_uint32(len(transcripts)), _p(transcripts),
                                       _p(np.ascontiguousarray(inst)), _p(np.ascontiguousarray(common)),
                                       _p(np.ascontiguousarray(commitments)), _p(np.ascontiguousarray(responses)),
                                       _p(np.ascontiguousarray(weights16)), threads, _p(co))
    if rc not in (0, 1):
        _raise(rc, "zkp_batch_verify_coeffs")
    return rc == 0, co


def batch_verify_build(st, transcripts, inst, common, commitments, responses, weights16, threads: int = 0):
    """Host-only half of batch_verify: the exact MSM operands (no GPU needed)."""
    n = len(commitments)
    _check_batch_shapes(st, n, inst, common, commitments, responses, weights16)
    total = st.ns + (st.ni + st.nc) * n
    ms = np.zeros((total, 32), np.uint8)
    mp = np.zeros((total, 32), np.uint8)
    rc = lib().zkp_batch_verify_build(st._h, ctypes.c_uint32(n), ctypes.c_uint32(len(transcripts)), _p(transcripts),
                                      _p(np.ascontiguousarray(inst)), _p(np.ascontiguousarray(common)),
                                      _p(np.ascontiguousarray(commitments)), _p(np.ascontiguousarray(responses)),
                                      _p(np.ascontiguousarray(weights16)), threads, _p(ms), _p(mp))
    _raise(rc, "zkp_batch_verify_build")
    return ms, mp



# ---- pipelines and device groups (include/zkp_toolbox.h, round 4) ------------------------------------------------------
def pinned_empty(shape, dtype=np.uint8) -> np.ndarray:
    """A numpy array in pinned host memory (zkp_host_alloc): jobs copy from / to it without staging.  The memory is freed when
    the array (and every view of it) is gone."""
    hip = load_library()
    hip.zkp_host_alloc.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_size_t]
    hip.zkp_host_free.argtypes = [ctypes.c_void_p]
    hip.zkp_host_free.restype = None
    nbytes = int(np.prod(shape)) * np.dtype(dtype).itemsize
    ptr = ctypes.c_void_p()
    rc = hip.zkp_host_alloc(ctypes.byref(ptr), ctypes.c_size_t(max(nbytes, 1)))
    if rc != 0:
        _raise(rc, "zkp_host_alloc")

    class _Owner:
        def __init__(self, p):
            self.p = p

        def __del__(self):
            try:
                hip.zkp_host_free(self.p)
            except Exception:
                pass
    buf = (ctypes.c_uint8 * max(nbytes, 1)).from_address(ptr.value)
    buf._owner = _Owner(ptr)                                     # keeps the allocation alive as long as the ctypes buffer is
    return np.frombuffer(buf, dtype=np.uint8, count=nbytes).view(dtype).reshape(shape)


def pinned_copy(a) -> np.ndarray:
    a = np.ascontiguousarray(a)
    out = pinned_empty(a.shape, a.dtype)
    out[...] = a
    return out


class Job:
    """A submitted call of a Pipe.  wait() returns what the synchronous call returns; the arrays handed to submit (kept alive
    here) hold the outputs afterwards.  The C side holds pointers into those arrays until the job has been waited for, so a Job
    that is dropped un-waited waits in __del__, and Pipe.close() waits for every job it still has in flight first."""

    def __init__(self, handle, keep, outputs, kind, pipe=None):
        self._h, self._keep, self.outputs, self.kind = handle, keep, outputs, kind
        self.context = int(lib().zkp_job_context_index(handle))
        self._pipe = pipe                                   # (keeps the pipe alive as long as the job is)
        if pipe is not None:
            pipe._jobs.add(self)

    def done(self) -> bool:
        return self._h is None or bool(lib().zkp_job_done(self._h))

    def wait(self, raise_on_failure: bool = True):
        if self._h is None:
            raise RuntimeError("job already waited for")
        rc = lib().zkp_job_wait(self._h)
        self._h = None
        self.rc = rc
        if rc < 0 or (raise_on_failure and rc != 0):
            _raise(rc, "zkp_job_wait")
        return self.outputs

    def _retire(self) -> None:
        """wait without raising (drop / pipe shutdown): afterwards nothing on the C side names this job's arrays"""
        if self._h is not None:
            h, self._h = self._h, None
            self.rc = lib().zkp_job_wait(h)

    def __del__(self):
        try:
            if self._pipe is None or getattr(self._pipe, "_h", None):
                self._retire()
        except Exception:
            pass


class Pipe:
    """zkp_pipe: `contexts_per_device` engine contexts on each listed GPU; asynchronous jobs (submit_* -> Job) and synchronous
    calls sharded over all contexts (prove_batch, batch_verify, ...)."""

    def __init__(self, devices=(0,), contexts_per_device: int = 3):
        devs = (ctypes.c_int * len(devices))(*devices)
        h = ctypes.c_void_p()
        rc = lib().zkp_pipe_create(ctypes.byref(h), devs, len(devices), contexts_per_device)
        if rc != 0:
            _raise(rc, "zkp_pipe_create")
        self._h = h
        self._jobs = weakref.WeakSet()                      # submitted and not yet waited for

    def close(self):
        if getattr(self, "_h", None):
            for j in list(self._jobs):                      # their arrays are still named by the contexts' pending copies
                j._retire()
            lib().zkp_pipe_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
        return False

    @property
    def num_contexts(self) -> int:
        return int(lib().zkp_pipe_num_contexts(self._h))

    @property
    def jobs_in_flight(self) -> int:
        return int(lib().zkp_pipe_jobs_in_flight(self._h))

    def last_error(self) -> str:
        return lib().zkp_pipe_last_error(self._h).decode()

    def set_submit_threads(self, on: int) -> None:
        """zkp_pipe_set_submit_threads: 1 = asynchronous submits are carried out by one host thread per entry of the device list (the default of a
        pipe over several entries), 0 = by the caller's thread, -1 = default.  Only while no job is in flight."""
        rc = lib().zkp_pipe_set_submit_threads(self._h, int(on))
        if rc != 0:
            _raise(rc, "zkp_pipe_set_submit_threads")

    def set_profiling(self, on: bool) -> None:
        hip = load_library()
        for i in range(self.num_contexts):
            hip.zkp_ctx_set_profiling(lib().zkp_pipe_context(self._h, i), 1 if on else 0)

    def job_timing(self, context: int):
        """(ms of host -> device copies, kernels, device -> host copies) of the last job that finished on that context (profiling on)"""
        hip = load_library()
        hip.zkp_ctx_job_timing.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_float)]
        ms = (ctypes.c_float * 3)()
        hip.zkp_ctx_job_timing(lib().zkp_pipe_context(self._h, context), ms)
        return tuple(float(x) for x in ms)

    def set_option(self, option: int, value: int, context: Optional[int] = None) -> None:
        """zkp_ctx_set_option on one context of the pipe, or on all of them"""
        hip = load_library()
        hip.zkp_ctx_set_option.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_uint64]
        for i in ([context] if context is not None else range(self.num_contexts)):
            rc = hip.zkp_ctx_set_option(lib().zkp_pipe_context(self._h, i), option, ctypes.c_uint64(value))
            if rc != 0:
                _raise(rc, "zkp_ctx_set_option")

    # -- asynchronous jobs ------------------------------------------------------------------------------------------
    @staticmethod
    def _ts(transcripts, n):
        """-> (array, flags): one 208-byte blob (shared) or [n][208]"""
        t = np.ascontiguousarray(transcripts, dtype=np.uint8)
        if t.shape == (TRANSCRIPT_BYTES,):
            return t, ZKP_JOB_SHARED_TRANSCRIPT
        if t.shape != (n, TRANSCRIPT_BYTES):
            raise ValueError("transcripts must be one blob [208] or [%d][208]" % n)
        return t, 0

    @staticmethod
    def _need(what, a, n_elems):
        """the C side reads n_elems elements: a shorter array would be an out-of-bounds read, not an error code"""
        if a is not None and np.asarray(a).size < n_elems:
            raise ValueError("%s has %d elements, the call reads %d" % (what, np.asarray(a).size, n_elems))

    def _submit(self, rc, h, what):
        if rc == ZKP_TB_PIPE_FULL:
            raise BlockingIOError("every context of the pipe has a job in flight")
        if rc != 0:
            _raise(rc, what)
        return h

    def submit_prove(self, st: Statement, n: int, transcripts, secrets, inst, common, entropy=None, want_transcripts=False, out=None,
                     inst_stride: Optional[int] = None) -> Job:
        """zkp_prove_batch_submit.  transcripts: one blob [208] (every proof starts from it) or [n][208].  out = optional dict of
        preallocated (e.g. pinned) arrays chal / resp / coms / ts.  -> Job whose outputs are (chal, resp, coms[, ts])."""
        ts, flags = self._ts(transcripts, n)
        self._need("secrets", secrets, n * st.m * 32)
        if not inst_stride:                       # (with a stride the caller passes a window into a larger array: its extent is the caller's business)
            self._need("inst", inst, st.ni * n * 32)
        self._need("common", common, st.ns * 32); self._need("entropy", entropy, n * 32)
        out = out or {}
        chal = out.get("chal") if out.get("chal") is not None else np.zeros((n, 32), np.uint8)
        resp = out.get("resp") if out.get("resp") is not None else np.zeros((n, st.m, 32), np.uint8)
        coms = out.get("coms") if out.get("coms") is not None else np.zeros((n, st.nc, 32), np.uint8)
        ts_out = (out.get("ts") if out.get("ts") is not None else np.zeros((n, TRANSCRIPT_BYTES), np.uint8)) if want_transcripts else None
        keep = [ts, np.ascontiguousarray(secrets), inst if inst_stride else np.ascontiguousarray(inst), np.ascontiguousarray(common),
                None if entropy is None else np.ascontiguousarray(entropy), chal, resp, coms, ts_out]
        h = ctypes.c_void_p()
        rc = lib().zkp_prove_batch_submit(self._h, st._h, n, flags, _p(keep[0]), _p(keep[1]), _p(keep[2]), inst_stride or n, _p(keep[3]), _p(keep[4]),
                                          _p(ts_out), _p(chal), _p(resp), _p(coms), ctypes.byref(h))
        self._submit(rc, h, "zkp_prove_batch_submit")
        return Job(h, keep + [st], (chal, resp, coms) + ((ts_out,) if want_transcripts else ()), "prove", self)

    def submit_batch_verify_many(self, st: Statement, n_batches: int, n_each: int, transcripts, inst, common, commitments, responses, weights16=None,
                                 want_transcripts=False, inst_stride: Optional[int] = None, weights_stride: Optional[int] = None) -> Job:
        """zkp_batch_verify_many_submit -> Job whose outputs are (verdicts[n_batches] (0 = Ok, 1 = VerificationFailure)[, ts])."""
        n = n_batches * n_each
        ts, flags = self._ts(transcripts, n)
        if not inst_stride:
            self._need("inst", inst, st.ni * n * 32)
        self._need("common", common, st.ns * 32); self._need("commitments", commitments, n * st.nc * 32); self._need("responses", responses, n * st.m * 32)
        if not weights_stride:
            self._need("weights16", weights16, st.nc * n * 16)
        verdicts = np.ones(n_batches, np.int32)
        ts_out = np.zeros((n, TRANSCRIPT_BYTES), np.uint8) if want_transcripts else None
        keep = [ts, inst if inst_stride else np.ascontiguousarray(inst), np.ascontiguousarray(common), np.ascontiguousarray(commitments),
                np.ascontiguousarray(responses), None if weights16 is None else (weights16 if weights_stride else np.ascontiguousarray(weights16)), verdicts, ts_out]
        h = ctypes.c_void_p()
        rc = lib().zkp_batch_verify_many_submit(self._h, st._h, n_batches, n_each, flags, _p(keep[0]), _p(keep[1]), inst_stride or n, _p(keep[2]), _p(keep[3]),
                                                _p(keep[4]), _p(keep[5]), weights_stride or n, _p(ts_out), _p(verdicts), ctypes.byref(h))
        self._submit(rc, h, "zkp_batch_verify_many_submit")
        return Job(h, keep + [st], (verdicts,) + ((ts_out,) if want_transcripts else ()), "batch_verify_many", self)

    def submit_verify_compact(self, st: Statement, n: int, transcripts, inst, common, challenges, responses, want_transcripts=False,
                              inst_stride: Optional[int] = None) -> Job:
        ts, flags = self._ts(transcripts, n)
        if not inst_stride:
            self._need("inst", inst, st.ni * n * 32)
        self._need("common", common, st.ns * 32); self._need("challenges", challenges, n * 32); self._need("responses", responses, n * st.m * 32)
        res = np.ones(n, np.uint8)
        ts_out = np.zeros((n, TRANSCRIPT_BYTES), np.uint8) if want_transcripts else None
        keep = [ts, inst if inst_stride else np.ascontiguousarray(inst), np.ascontiguousarray(common), np.ascontiguousarray(challenges),
                np.ascontiguousarray(responses), res, ts_out]
        h = ctypes.c_void_p()
        rc = lib().zkp_verify_compact_batch_submit(self._h, st._h, n, flags, _p(keep[0]), _p(keep[1]), inst_stride or n, _p(keep[2]), _p(keep[3]), _p(keep[4]),
                                                   _p(ts_out), _p(res), ctypes.byref(h))
        self._submit(rc, h, "zkp_verify_compact_batch_submit")
        return Job(h, keep + [st], (res,) + ((ts_out,) if want_transcripts else ()), "verify_compact", self)

    def submit_verify_batchable_each(self, st: Statement, n: int, transcripts, inst, common, commitments, responses, weights16=None,
                                     want_transcripts=False, inst_stride: Optional[int] = None) -> Job:
        ts, flags = self._ts(transcripts, n)
        if not inst_stride:
            self._need("inst", inst, st.ni * n * 32)
        self._need("common", common, st.ns * 32); self._need("commitments", commitments, n * st.nc * 32); self._need("responses", responses, n * st.m * 32)
        self._need("weights16", weights16, n * st.nc * 16)
        res = np.ones(n, np.uint8)
        ts_out = np.zeros((n, TRANSCRIPT_BYTES), np.uint8) if want_transcripts else None
        keep = [ts, inst if inst_stride else np.ascontiguousarray(inst), np.ascontiguousarray(common), np.ascontiguousarray(commitments),
                np.ascontiguousarray(responses), None if weights16 is None else np.ascontiguousarray(weights16), res, ts_out]
        h = ctypes.c_void_p()
        rc = lib().zkp_verify_batchable_each_submit(self._h, st._h, n, flags, _p(keep[0]), _p(keep[1]), inst_stride or n, _p(keep[2]), _p(keep[3]), _p(keep[4]),
                                                    _p(keep[5]), _p(ts_out), _p(res), ctypes.byref(h))
        self._submit(rc, h, "zkp_verify_batchable_each_submit")
        return Job(h, keep + [st], (res,) + ((ts_out,) if want_transcripts else ()), "verify_batchable_each", self)

    # -- synchronous calls sharded over every context (one host thread per listed device) ------------------------------
    def prove_batch(self, st: Statement, transcripts: np.ndarray, secrets, inst, common, entropy=None):
        n = len(transcripts)
        _check_batch_shapes(st, n, inst, common, None, secrets)
        chal = np.zeros((n, 32), np.uint8)
        resp = np.zeros((n, st.m, 32), np.uint8)
        coms = np.zeros((n, st.nc, 32), np.uint8)
        rc = lib().zkp_pipe_prove_batch(self._h, st._h, n, _p(transcripts), _p(np.ascontiguousarray(secrets)), _p(np.ascontiguousarray(inst)),
                                        _p(np.ascontiguousarray(common)), _p(None if entropy is None else np.ascontiguousarray(entropy)), _p(chal), _p(resp), _p(coms))
        _raise(rc, "zkp_pipe_prove_batch")
        return chal, resp, coms

    def verify_compact_batch(self, st, transcripts, inst, common, challenges, responses) -> np.ndarray:
        n = len(transcripts)
        _check_batch_shapes(st, n, inst, common, None, responses)
        res = np.ones(n, np.uint8)
        rc = lib().zkp_pipe_verify_compact_batch(self._h, st._h, n, _p(transcripts), _p(np.ascontiguousarray(inst)), _p(np.ascontiguousarray(common)),
                                                 _p(np.ascontiguousarray(challenges)), _p(np.ascontiguousarray(responses)), _p(res))
        _raise(rc, "zkp_pipe_verify_compact_batch")
        return res

    def verify_batchable_each(self, st, transcripts, inst, common, commitments, responses, weights16=None) -> np.ndarray:
        n = len(transcripts)
        _check_batch_shapes(st, n, inst, common, commitments, responses, weights16, per_proof_weights=True)
        res = np.ones(n, np.uint8)
        rc = lib().zkp_pipe_verify_batchable_each(self._h, st._h, n, _p(transcripts), _p(np.ascontiguousarray(inst)), _p(np.ascontiguousarray(common)),
                                                  _p(np.ascontiguousarray(commitments)), _p(np.ascontiguousarray(responses)),
                                                  _p(None if weights16 is None else np.ascontiguousarray(weights16)), _p(res))
        _raise(rc, "zkp_pipe_verify_batchable_each")
        return res

    def batch_verify(self, st, transcripts, inst, common, commitments, responses, weights16=None) -> None:
        """Raises VerificationFailure unless every range of the batch verifies (host AND of the per-context verdicts)."""
        n = len(commitments)
        _check_batch_shapes(st, n, inst, common, commitments, responses, weights16)
        rc = lib().zkp_pipe_batch_verify(self._h, st._h, n, len(transcripts), _p(transcripts), _p(np.ascontiguousarray(inst)), _p(np.ascontiguousarray(common)),
                                         _p(np.ascontiguousarray(commitments)), _p(np.ascontiguousarray(responses)),
                                         _p(None if weights16 is None else np.ascontiguousarray(weights16)))
        _raise(rc, "zkp_pipe_batch_verify")

    def batch_verify_many(self, st, n_batches: int, transcripts, inst, common, commitments, responses, weights16=None) -> np.ndarray:
        n = len(commitments)
        if n_batches <= 0 or n % n_batches:
            raise ValueError("the number of proofs must be a positive multiple of n_batches")
        _check_batch_shapes(st, n, inst, common, commitments, responses, weights16)
        verdicts = np.ones(n_batches, np.int32)
        rc = lib().zkp_pipe_batch_verify_many(self._h, st._h, n_batches, n // n_batches, len(transcripts), _p(transcripts), _p(np.ascontiguousarray(inst)),
                                              _p(np.ascontiguousarray(common)), _p(np.ascontiguousarray(commitments)), _p(np.ascontiguousarray(responses)),
                                              _p(None if weights16 is None else np.ascontiguousarray(weights16)), _p(verdicts))
        _raise(rc, "zkp_pipe_batch_verify_many")
        return verdicts

    def batch_verify_locate(self, st, transcripts, inst, common, commitments, responses, weights16=None):
        n = len(commitments)
        _check_batch_shapes(st, n, inst, common, commitments, responses, weights16)
        res = np.ones(n, np.uint8)
        rc = lib().zkp_pipe_batch_verify_locate(self._h, st._h, n, len(transcripts), _p(transcripts), _p(np.ascontiguousarray(inst)), _p(np.ascontiguousarray(common)),
                                                _p(np.ascontiguousarray(commitments)), _p(np.ascontiguousarray(responses)),
                                                _p(None if weights16 is None else np.ascontiguousarray(weights16)), _p(res))
        if rc not in (0, 1):
            _raise(rc, "zkp_pipe_batch_verify_locate")
        return rc == 0, res


def prove_phase_a(st, transcripts, secrets, inst, common, entropy, threads: int = 0):
    n = len(transcripts)
    blind = np.zeros((n, st.m, 32), np.uint8)
    off = np.zeros(n * st.nc + 1, np.uint32)
    sc = np.zeros((n * st.terms, 32), np.uint8)
    pidx = np.zeros(n * st.terms, np.uint32)
    rc = lib().zkp_prove_phase_a(st._h, ctypes.c_uint32(n), _p(transcripts), _p(np.ascontiguousarray(secrets)),
                                 _p(np.ascontiguousarray(inst)), _p(np.ascontiguousarray(common)),
                                 _p(np.ascontiguousarray(entropy)), threads, _p(blind), _p(off), _p(sc), _p(pidx))
    _raise(rc, "zkp_prove_phase_a")
    return blind, off, sc, pidx


def prove_phase_b(st, transcripts, secrets, blindings, commitments, threads: int = 0):
    n = len(transcripts)
    chal = np.zeros((n, 32), np.uint8)
    resp = np.zeros((n, st.m, 32), np.uint8)
    rc = lib().zkp_prove_phase_b(st._h, ctypes.c_uint32(n), _p(transcripts), _p(np.ascontiguousarray(secrets)),
                                 _p(np.ascontiguousarray(blindings)), _p(np.ascontiguousarray(commitments)), threads, _p(chal), _p(resp))
    _raise(rc, "zkp_prove_phase_b")
    return chal, resp


# ---- the reference's object API (one proof at a time; the batch functions do the work) ------------
@dataclass(frozen=True)
class ScalarVar:
    idx: int


@dataclass(frozen=True)
class PointVar:
    idx: int


def _enc(e) -> bytes:
    e = bytes(e) if not isinstance(e, np.ndarray) else e.tobytes()
    assert len(e) == 32
    return e


class Prover:
    """toolbox::prover::Prover (src/toolbox/prover.rs).  Points are supplied as encodings."""

    def __init__(self, proof_label: bytes, transcript: Transcript, engine: Engine):
        self._st = Statement(proof_label)
        self._t = transcript
        self._eng = engine
        self._scalars: List[bytes] = []
        self._points: List[bytes] = []

    def allocate_scalar(self, label: bytes, assignment) -> ScalarVar:
        self._scalars.append(scalar_bytes(assignment))
        return ScalarVar(self._st.add_secret(label))

    def allocate_point(self, label: bytes, assignment) -> Tuple[PointVar, bytes]:
        e = _enc(assignment)
        self._points.append(e)
        return PointVar(self._st.add_point(label, False)), e

    def constrain(self, lhs: PointVar, linear_combination: Sequence[Tuple[ScalarVar, PointVar]]) -> None:
        self._st.constrain(lhs.idx, [(s.idx, p.idx) for s, p in linear_combination])

    def _prove(self, entropy: Optional[bytes]):
        ts = self._t.state.reshape(1, -1).copy()
        secrets = np.frombuffer(b"".join(self._scalars), np.uint8).reshape(1, -1, 32) if self._scalars else np.zeros((1, 0, 32), np.uint8)
        inst = np.frombuffer(b"".join(self._points), np.uint8).reshape(-1, 1, 32) if self._points else np.zeros((0, 1, 32), np.uint8)
        ent = None if entropy is None else np.frombuffer(entropy, np.uint8).reshape(1, 32)
        chal, resp, coms = prove_batch(self._eng, self._st, ts, secrets, inst, np.zeros((0, 32), np.uint8), ent, threads=1)
        self._t.state[:] = ts[0]
        return chal[0].tobytes(), [r.tobytes() for r in resp[0]], [c.tobytes() for c in coms[0]]

    def prove_compact(self, entropy: Optional[bytes] = None) -> CompactProof:
        c, r, _ = self._prove(entropy)
        return CompactProof(c, r)

    def prove_batchable(self, entropy: Optional[bytes] = None) -> BatchableProof:
        _, r, k = self._prove(entropy)
        return BatchableProof(k, r)


class Verifier:
    """toolbox::verifier::Verifier (src/toolbox/verifier.rs)."""

    def __init__(self, proof_label: bytes, transcript: Transcript, engine: Engine):
        self._st = Statement(proof_label)
        self._t = transcript
        self._eng = engine
        self._points: List[bytes] = []

    def allocate_scalar(self, label: bytes) -> ScalarVar:
        return ScalarVar(self._st.add_secret(label))

    def allocate_point(self, label: bytes, assignment) -> PointVar:
        e = _enc(assignment)
        if e == bytes(32):                      # validate_and_append_point_var, mod.rs:191-193
            raise VerificationFailure()
        self._points.append(e)
        return PointVar(self._st.add_point(label, False))

    def constrain(self, lhs: PointVar, linear_combination) -> None:
        self._st.constrain(lhs.idx, [(s.idx, p.idx) for s, p in linear_combination])

    def _inst(self):
        return np.frombuffer(b"".join(self._points), np.uint8).reshape(-1, 1, 32) if self._points else np.zeros((0, 1, 32), np.uint8)

    def verify_compact(self, proof: CompactProof) -> None:
        if len(proof.responses) != self._st.m:                          # verifier.rs:82-84
            raise VerificationFailure()
        ts = self._t.state.reshape(1, -1).copy()
        resp = np.frombuffer(b"".join(proof.responses), np.uint8).reshape(1, -1, 32) if proof.responses else np.zeros((1, 0, 32), np.uint8)
        res = verify_compact_batch(self._eng, self._st, ts, self._inst(), np.zeros((0, 32), np.uint8),
                                   np.frombuffer(proof.challenge, np.uint8).reshape(1, 32), resp, threads=1)
        self._t.state[:] = ts[0]
        if res[0]:
            raise VerificationFailure()

    def verify_batchable(self, proof: BatchableProof, weights: Optional[Sequence[int]] = None) -> None:
        if len(proof.responses) != self._st.m or len(proof.commitments) != self._st.nc:     # verifier.rs:125-131
            raise VerificationFailure()
        ts = self._t.state.reshape(1, -1).copy()
        resp = np.frombuffer(b"".join(proof.responses), np.uint8).reshape(1, -1, 32) if proof.responses else np.zeros((1, 0, 32), np.uint8)
        coms = np.frombuffer(b"".join(proof.commitments), np.uint8).reshape(1, -1, 32) if proof.commitments else np.zeros((1, 0, 32), np.uint8)
        w = None if weights is None else np.frombuffer(b"".join(int(x).to_bytes(16, "little") for x in weights), np.uint8).reshape(1, -1, 16)
        res = verify_batchable_each(self._eng, self._st, ts, self._inst(), np.zeros((0, 32), np.uint8), coms, resp, w, threads=1)
        self._t.state[:] = ts[0]
        if res[0]:
            raise VerificationFailure()


class BatchVerifier:
    """toolbox::batch_verifier::BatchVerifier (src/toolbox/batch_verifier.rs)."""

    def __init__(self, proof_label: bytes, batch_size: int, transcripts: Sequence[Transcript], engine: Engine):
        if len(transcripts) != batch_size:                              # batch_verifier.rs:72-74
            raise BatchSizeMismatch()
        self._st = Statement(proof_label)
        self._n = batch_size
        self._ts = list(transcripts)
        self._eng = engine
        self._static: List[bytes] = []
        self._instance: List[List[bytes]] = []

    def allocate_scalar(self, label: bytes) -> ScalarVar:
        return ScalarVar(self._st.add_secret(label))

    def allocate_static_point(self, label: bytes, assignment) -> PointVar:
        e = _enc(assignment)
        if e == bytes(32):
            raise VerificationFailure()
        self._static.append(e)
        return PointVar(self._st.add_point(label, True))

    def allocate_instance_point(self, label: bytes, assignments: Sequence) -> PointVar:
        if len(assignments) != self._n:                                 # batch_verifier.rs:120-122
            raise BatchSizeMismatch()
        es = [_enc(a) for a in assignments]
        if any(e == bytes(32) for e in es):
            raise VerificationFailure()
        self._instance.append(es)
        return PointVar(self._st.add_point(label, False))

    def constrain(self, lhs: PointVar, linear_combination) -> None:
        self._st.constrain(lhs.idx, [(s.idx, p.idx) for s, p in linear_combination])

    def verify_batchable(self, proofs: Sequence[BatchableProof], weights: Optional[Sequence[Sequence[int]]] = None) -> None:
        if len(proofs) != self._n:                                      # batch_verifier.rs:138-140
            raise BatchSizeMismatch()
        for p in proofs:                                                # batch_verifier.rs:142-149
            if len(p.commitments) != self._st.nc or len(p.responses) != self._st.m:
                raise VerificationFailure()
        n = self._n
        ts = _transcripts_array(self._ts)
        inst = np.frombuffer(b"".join(e for row in self._instance for e in row), np.uint8).reshape(-1, n, 32) if self._instance else np.zeros((0, n, 32), np.uint8)
        common = np.frombuffer(b"".join(self._static), np.uint8).reshape(-1, 32) if self._static else np.zeros((0, 32), np.uint8)
        coms = np.frombuffer(b"".join(c for p in proofs for c in p.commitments), np.uint8).reshape(n, -1, 32) if n and self._st.nc else np.zeros((n, 0, 32), np.uint8)
        resp = np.frombuffer(b"".join(r for p in proofs for r in p.responses), np.uint8).reshape(n, -1, 32) if n and self._st.m else np.zeros((n, 0, 32), np.uint8)
        w = None if weights is None else np.frombuffer(b"".join(int(x).to_bytes(16, "little") for row in weights for x in row), np.uint8).reshape(-1, n, 16)
        try:
            batch_verify(self._eng, self._st, ts, inst, common, coms, resp, w, batch_size=n)
        finally:
            _store_transcripts(self._ts, ts)


# ---- define_proof! -----------------------------------------------------------------------------------
class ProofModule:
    """What `define_proof!` generates (src/macros.rs:124-370): a statement with fixed labels and allocation
    order (secrets, instance points, common points) and the five entry points."""

    def __init__(self, name: str, label: bytes, secrets: Sequence[str], instance: Sequence[str], common: Sequence[str],
                 constraints: Sequence[Tuple[str, Sequence[Tuple[str, str]]]]):
        self.name, self.label = name, label
        self.secrets, self.instance, self.common = list(secrets), list(instance), list(common)
        self.constraints = [(l, list(lc)) for l, lc in constraints]
        self.statement = Statement(label)
        sv = {n: self.statement.add_secret(n.encode()) for n in self.secrets}                # macros.rs:215-222
        pv = {n: self.statement.add_point(n.encode(), False) for n in self.instance}         # macros.rs:229-235
        pv.update({n: self.statement.add_point(n.encode(), True) for n in self.common})      # macros.rs:236-242
        for lhs, lc in self.constraints:                                                    # macros.rs:159-170
            self.statement.constrain(pv[lhs], [(sv[s], pv[p]) for s, p in lc])

    # -- array helpers ----------------------------------------------------------------------------
    def pack(self, secrets: Sequence[Dict[str, object]], points: Sequence[Dict[str, bytes]]):
        """list of per-proof dicts -> (secrets[N][m][32], inst[ni][N][32], common[ns][32])"""
        n = len(points)
        sec = np.frombuffer(b"".join(scalar_bytes(s[k]) for s in secrets for k in self.secrets), np.uint8).reshape(n, len(self.secrets), 32) if secrets else None
        inst = np.frombuffer(b"".join(_enc(p[k]) for k in self.instance for p in points), np.uint8).reshape(len(self.instance), n, 32)
        common = np.frombuffer(b"".join(_enc(points[0][k]) for k in self.common), np.uint8).reshape(len(self.common), 32)
        return sec, inst, common

    def prove_compact(self, eng, transcript: Transcript, secrets: Dict[str, object], points: Dict[str, bytes], entropy=None) -> CompactProof:
        c, r, _ = self._prove(eng, transcript, secrets, points, entropy)
        return CompactProof(c, r)

    def prove_batchable(self, eng, transcript: Transcript, secrets, points, entropy=None) -> BatchableProof:
        _, r, k = self._prove(eng, transcript, secrets, points, entropy)
        return BatchableProof(k, r)

    def _prove(self, eng, transcript, secrets, points, entropy):
        sec, inst, common = self.pack([secrets], [points])
        ts = transcript.state.reshape(1, -1).copy()
        ent = None if entropy is None else np.frombuffer(entropy, np.uint8).reshape(1, 32)
        chal, resp, coms = prove_batch(eng, self.statement, ts, sec, inst, common, ent, threads=1)
        transcript.state[:] = ts[0]
        return chal[0].tobytes(), [x.tobytes() for x in resp[0]], [x.tobytes() for x in coms[0]]

    def verify_compact(self, eng, proof: CompactProof, transcript: Transcript, points: Dict[str, bytes]) -> None:
        if any(_enc(points[k]) == bytes(32) for k in self.instance + self.common) or len(proof.responses) != len(self.secrets):
            raise VerificationFailure()
        _, inst, common = self.pack([], [points])
        ts = transcript.state.reshape(1, -1).copy()
        res = verify_compact_batch(eng, self.statement, ts, inst, common, np.frombuffer(proof.challenge, np.uint8).reshape(1, 32),
                                   np.frombuffer(b"".join(proof.responses), np.uint8).reshape(1, -1, 32), threads=1)
        transcript.state[:] = ts[0]
        if res[0]:
            raise VerificationFailure()

    def verify_batchable(self, eng, proof: BatchableProof, transcript: Transcript, points: Dict[str, bytes], weights=None) -> None:
        if len(proof.responses) != len(self.secrets) or len(proof.commitments) != len(self.constraints):
            raise VerificationFailure()
        _, inst, common = self.pack([], [points])
        ts = transcript.state.reshape(1, -1).copy()
        w = None if weights is None else np.frombuffer(b"".join(int(x).to_bytes(16, "little") for x in weights), np.uint8).reshape(1, -1, 16)
        res = verify_batchable_each(eng, self.statement, ts, inst, common,
                                    np.frombuffer(b"".join(proof.commitments), np.uint8).reshape(1, -1, 32),
                                    np.frombuffer(b"".join(proof.responses), np.uint8).reshape(1, -1, 32), w, threads=1)
        transcript.state[:] = ts[0]
        if res[0]:
            raise VerificationFailure()

    def batch_verify(self, eng, proofs: Sequence[BatchableProof], transcripts: Sequence[Transcript],
                     instance_points: Dict[str, Sequence[bytes]], common_points: Dict[str, bytes], weights=None, threads: int = 0) -> None:
        n = len(proofs)
        if len(transcripts) != n:
            raise BatchSizeMismatch()
        for k in self.instance:
            if len(instance_points[k]) != n:
                raise BatchSizeMismatch()
        for p in proofs:                                                # batch_verifier.rs:142-149
            if len(p.commitments) != len(self.constraints) or len(p.responses) != len(self.secrets):
                raise VerificationFailure()
        ts = _transcripts_array(transcripts)
        inst = np.frombuffer(b"".join(_enc(e) for k in self.instance for e in instance_points[k]), np.uint8).reshape(len(self.instance), n, 32)
        common = np.frombuffer(b"".join(_enc(common_points[k]) for k in self.common), np.uint8).reshape(len(self.common), 32)
        coms = np.frombuffer(b"".join(c for p in proofs for c in p.commitments), np.uint8).reshape(n, len(self.constraints), 32)
        resp = np.frombuffer(b"".join(r for p in proofs for r in p.responses), np.uint8).reshape(n, len(self.secrets), 32)
        w = None if weights is None else np.frombuffer(b"".join(int(x).to_bytes(16, "little") for row in weights for x in row), np.uint8).reshape(len(self.constraints), n, 16)
        try:
            batch_verify(eng, self.statement, ts, inst, common, coms, resp, w, threads=threads, batch_size=n)
        finally:
            _store_transcripts(transcripts, ts)


def define_proof(name, label, secrets, instance, common, constraints) -> ProofModule:
    return ProofModule(name, label if isinstance(label, bytes) else label.encode(), secrets, instance, common, constraints)


def dleq_module() -> ProofModule:
    """define_proof! {dleq, "DLEQ proof", (x), (A, B, H), (G) : A = (x * G), B = (x * H)}  (benches/zkp.rs:49)"""
    return define_proof("dleq", b"DLEQ proof", ["x"], ["A", "B", "H"], ["G"], [("A", [("x", "G")]), ("B", [("x", "H")])])


def cmz_module(n: int = 10) -> ProofModule:
    """cred_show_10 (benches/zkp.rs:27-46)."""
    ms = [f"m_{i}" for i in range(1, n + 1)]
    zs = [f"z_{i}" for i in range(1, n + 1)]
    cs = [f"C_{i}" for i in range(1, n + 1)]
    xs = [f"X_{i}" for i in range(1, n + 1)]
    cons = [(cs[i], [(ms[i], "P"), (zs[i], "A")]) for i in range(n)]
    cons.append(("V", [(ms[i], xs[i]) for i in range(n)] + [("minus_z_Q", "Q")]))
    return define_proof(f"cred_show_{n}", f"CMZ cred show n={n}".encode(), ms + zs + ["minus_z_Q"], cs + ["P", "Q", "V"], xs + ["A", "B"], cons)
