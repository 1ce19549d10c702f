"""ORACLE (test infrastructure): ctypes binding of oracle/liboracle.so (the plain-C CPU restatement,
oracle/c/oracle.h).  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this.
Parity unpinned against the reference binary (see oracle/c/oracle.h); pinned to RFC 9496 / Merlin / libsodium vectors."""
from __future__ import annotations

import ctypes
import os
import subprocess
from typing import List, Optional, Sequence, Tuple

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "liboracle.so")
_lib = None


def _lib_path() -> str:
    # ORACLE_SANITIZE=1: load the AddressSanitizer / UBSan build (oracle/Makefile: liboracle_asan.so); the process must have
    # been started with LD_PRELOAD of the sanitizer runtimes (tests/test_oracle_c.py::test_oracle_is_clean_under_sanitizers)
    return os.path.join(_HERE, "liboracle_asan.so") if os.environ.get("ORACLE_SANITIZE") else LIB_PATH


def build(force: bool = False) -> str:
    srcs = [os.path.join(_HERE, "c", f) for f in os.listdir(os.path.join(_HERE, "c"))]
    stale = (not os.path.exists(LIB_PATH)) or any(os.path.getmtime(s) > os.path.getmtime(LIB_PATH) for s in srcs)
    if force or stale:
        subprocess.check_call(["make", "-C", _HERE, "-B", "liboracle.so"], stdout=subprocess.DEVNULL)
    return LIB_PATH


class _Statement(ctypes.Structure):
    _fields_ = [("label", ctypes.c_char_p), ("n_secrets", ctypes.c_uint32), ("n_inst", ctypes.c_uint32),
                ("n_common", ctypes.c_uint32), ("n_cons", ctypes.c_uint32),
                ("secret_names", ctypes.POINTER(ctypes.c_char_p)), ("point_names", ctypes.POINTER(ctypes.c_char_p)),
                ("point_is_common", ctypes.c_void_p), ("cons_lhs", ctypes.c_void_p), ("cons_off", ctypes.c_void_p),
                ("cons_sc", ctypes.c_void_p), ("cons_pt", ctypes.c_void_p)]


def lib() -> ctypes.CDLL:
    global _lib
    if _lib is None:
        path = _lib_path()
        if path != LIB_PATH:
            subprocess.check_call(["make", "-C", _HERE, "liboracle_asan.so"], stdout=subprocess.DEVNULL)
        elif not os.path.exists(LIB_PATH):
            build()
        _lib = ctypes.CDLL(path)
        _lib.orc_keccak_count.restype = ctypes.c_uint64
    return _lib


SIMD_ISA = {0: None, 1: "avx512ifma", 2: "avx2", 3: "avx2p"}


def simd_available() -> bool:
    """the build and this CPU have a vector instruction set the backend of oracle/c/simd_ifma.c runs on (AVX-512 IFMA + VL, or AVX2)"""
    return bool(lib().orc_simd_available())


def simd_isas():
    """the instruction sets set_simd() accepts here, best first"""
    best = int(lib().orc_simd_available())
    return {0: [], 1: ["avx512ifma", "avx2p", "avx2"], 2: ["avx2p", "avx2"]}[best]


def set_simd(on) -> bool:
    """MSM inner loops on the vector backend (dalek's simd_backend design): True = the best instruction set this CPU has, "avx2" = AVX2 even
    where IFMA exists, "avx512ifma", False = the scalar port.  Returns whether a vector backend is on now (False: refused)."""
    if on in (False, None, 0):
        lib().orc_set_simd(0)
        return False
    if on == "avx2":
        return int(lib().orc_set_simd(2)) == 2
    if on == "avx2p":                                      # AVX2 in curve25519-dalek's packed layout (FieldElement2625x4: five vectors of eight 32-bit lanes)
        return int(lib().orc_set_simd(3)) == 3
    got = int(lib().orc_set_simd(1))
    if on == "avx512ifma" and got != 1:
        lib().orc_set_simd(0)
        return False
    return got != 0


def simd_mode():
    """which backend the MSM entry points run on right now: None (scalar port), "avx512ifma" or "avx2" """
    return SIMD_ISA[int(lib().orc_simd_mode())]


def _p(a):
    return None if a is None else a.ctypes.data_as(ctypes.c_void_p)


def _u8(a, last=None) -> np.ndarray:
    a = np.ascontiguousarray(a, dtype=np.uint8)
    if last is not None and a.size:
        a = a.reshape(-1, last)
    return a


class Statement:
    """Marshals a statement descriptor (secret names; point names in allocation order; which points are
    common; constraints) into oracle.h's orc_statement."""

    def __init__(self, label: bytes, secrets: Sequence[str], points: Sequence[Tuple[str, bool]],
                 constraints: Sequence[Tuple[str, Sequence[Tuple[str, str]]]]):
        self.label = label
        self.secrets = list(secrets)
        self.points = [n for n, _ in points]
        self.is_common = [bool(c) for _, c in points]
        self.constraints = [(lhs, list(lc)) for lhs, lc in constraints]
        si = {n: i for i, n in enumerate(self.secrets)}
        pi = {n: i for i, n in enumerate(self.points)}
        self.n_inst = sum(1 for c in self.is_common if not c)
        self.n_common = sum(1 for c in self.is_common if c)
        self._sn = (ctypes.c_char_p * max(1, len(self.secrets)))(*[s.encode() for s in self.secrets])
        self._pn = (ctypes.c_char_p * max(1, len(self.points)))(*[s.encode() for s in self.points])
        self._isc = np.array(self.is_common, dtype=np.uint8)
        self._lhs = np.array([pi[l] for l, _ in self.constraints], dtype=np.uint32)
        off = [0]
        sc, pt = [], []
        for _, lc in self.constraints:
            for s, p in lc:
                sc.append(si[s])
                pt.append(pi[p])
            off.append(len(sc))
        self._off = np.array(off, dtype=np.uint32)
        self._sc = np.array(sc, dtype=np.uint32)
        self._pt = np.array(pt, dtype=np.uint32)
        self.c = _Statement(label, len(self.secrets), self.n_inst, self.n_common, len(self.constraints),
                            self._sn, self._pn, self._isc.ctypes.data, self._lhs.ctypes.data, self._off.ctypes.data,
                            self._sc.ctypes.data, self._pt.ctypes.data)

    @classmethod
    def from_model(cls, st) -> "Statement":
        """From oracle.model.Statement (macro order: instance points, then common points)."""
        pts = [(n, False) for n in st.instance] + [(n, True) for n in st.common]
        return cls(st.label, st.secrets, pts, st.constraints)


# ---- thin wrappers ------------------------------------------------------------------------------
def decode_check(points, want_coords=False):
    points = _u8(points, 32)
    n = len(points)
    status = np.zeros(n, np.uint8)
    xyzt = np.zeros((n, 128), np.uint8) if want_coords else None
    lib().orc_decode_check(ctypes.c_uint64(n), _p(points), _p(status), _p(xyzt))
    return (status, xyzt) if want_coords else status


def encode_many(xyzt):
    xyzt = _u8(xyzt, 128)
    out = np.zeros((len(xyzt), 32), np.uint8)
    lib().orc_encode_many(ctypes.c_uint64(len(xyzt)), _p(xyzt), _p(out))
    return out


def from_uniform_bytes(b64: bytes) -> bytes:
    ge = (ctypes.c_uint64 * 20)()
    out = ctypes.create_string_buffer(32)
    lib().orc_ristretto_from_uniform_bytes(ge, b64)
    lib().orc_ristretto_encode(out, ge)
    return out.raw


def msm_many(off, scalars, pidx, points, flags=0):
    off = np.ascontiguousarray(off, dtype=np.uint32)
    pidx = np.ascontiguousarray(pidx, dtype=np.uint32)
    scalars = _u8(scalars, 32)
    points = _u8(points, 32)
    n_msm = len(off) - 1
    out = np.zeros((n_msm, 32), np.uint8)
    status = np.zeros(n_msm, np.uint8)
    lib().orc_msm_many(ctypes.c_uint32(n_msm), _p(off), _p(scalars), _p(pidx), _p(points), ctypes.c_uint32(len(points)),
                       ctypes.c_int(flags), _p(out), _p(status))
    return out, status


def msm_optional(scalars, points) -> Optional[bytes]:
    scalars = _u8(scalars, 32)
    points = _u8(points, 32)
    out = np.zeros(32, np.uint8)
    st = ctypes.c_int(1)
    lib().orc_msm_optional(ctypes.c_uint64(len(points)), _p(scalars), _p(points), _p(out), ctypes.byref(st))
    return None if st.value else out.tobytes()


def msm_algo(which: str, scalars, points_enc) -> bytes:
    """Run ONE named dalek algorithm ('straus_ct' | 'straus_vartime' | 'pippenger') on decoded points."""
    scalars = _u8(scalars, 32)
    points_enc = _u8(points_enc, 32)
    n = len(points_enc)
    ges = (ctypes.c_uint64 * (20 * max(n, 1)))()
    for i in range(n):
        ok = lib().orc_ristretto_decode(ctypes.byref(ges, 160 * i), points_enc[i].tobytes())
        assert ok == 1
    r = (ctypes.c_uint64 * 20)()
    getattr(lib(), "orc_msm_" + which)(r, ctypes.c_size_t(n), _p(scalars), ges)
    out = ctypes.create_string_buffer(32)
    lib().orc_ristretto_encode(out, r)
    return out.raw


def sc_from_wide(b64: bytes) -> bytes:
    out = ctypes.create_string_buffer(32)
    lib().orc_sc_from_wide(out, b64)
    return out.raw


def sc_muladd(a: bytes, b: bytes, c: bytes) -> bytes:
    out = ctypes.create_string_buffer(32)
    lib().orc_sc_muladd(out, a, b, c)
    return out.raw


def sc_neg(a: bytes) -> bytes:
    out = ctypes.create_string_buffer(32)
    lib().orc_sc_neg(out, a)
    return out.raw


def merlin_challenge(label: bytes, appends: Sequence[Tuple[bytes, bytes]], chal_label: bytes, n: int) -> bytes:
    t = ctypes.create_string_buffer(256)
    lib().orc_transcript_init(t, label, ctypes.c_size_t(len(label)))
    for l, m in appends:
        lib().orc_transcript_append(t, l, m, ctypes.c_size_t(len(m)))
    out = ctypes.create_string_buffer(n)
    lib().orc_transcript_challenge(t, chal_label, out, ctypes.c_size_t(n))
    return out.raw


def prove(st: Statement, transcript_label: bytes, secrets, points, entropy32: bytes):
    """-> (challenge[32], responses[m][32], commitments[n_cons][32], blindings[m][32])"""
    secrets = _u8(secrets, 32)
    points = _u8(points, 32)
    m, nc = len(st.secrets), len(st.constraints)
    chal = np.zeros(32, np.uint8)
    resp = np.zeros((m, 32), np.uint8)
    coms = np.zeros((nc, 32), np.uint8)
    blind = np.zeros((m, 32), np.uint8)
    rc = lib().orc_prove(ctypes.byref(st.c), transcript_label, ctypes.c_size_t(len(transcript_label)), _p(secrets),
                         _p(points), entropy32, _p(chal), _p(resp), _p(coms), _p(blind))
    if rc != 0:
        raise ValueError("orc_prove: a public point failed to decode")
    return chal, resp, coms, blind


def verify_compact(st: Statement, transcript_label: bytes, points, challenge, responses) -> int:
    return lib().orc_verify_compact(ctypes.byref(st.c), transcript_label, ctypes.c_size_t(len(transcript_label)),
                                    _p(_u8(points, 32)), _p(_u8(challenge)), _p(_u8(responses, 32)))


def verify_batchable(st: Statement, transcript_label: bytes, points, commitments, responses, weights16) -> int:
    return lib().orc_verify_batchable(ctypes.byref(st.c), transcript_label, ctypes.c_size_t(len(transcript_label)),
                                      _p(_u8(points, 32)), _p(_u8(commitments, 32)), _p(_u8(responses, 32)),
                                      _p(_u8(weights16, 16)))


def batch_verify(st: Statement, transcript_label: bytes, n: int, inst_points, common_points, commitments, responses,
                 weights16, want_msm_inputs: bool = False):
    """-> rc, or (rc, msm_scalars, msm_points) when want_msm_inputs (then the MSM itself is skipped)."""
    total = st.n_common + (st.n_inst + len(st.constraints)) * n
    ms = np.zeros((total, 32), np.uint8) if want_msm_inputs else None
    mp = np.zeros((total, 32), np.uint8) if want_msm_inputs else None
    rc = lib().orc_batch_verify(ctypes.byref(st.c), transcript_label, ctypes.c_size_t(len(transcript_label)),
                                ctypes.c_uint32(n), _p(_u8(inst_points)), _p(_u8(common_points)), _p(_u8(commitments)),
                                _p(_u8(responses)), _p(_u8(weights16)), _p(ms), _p(mp))
    return (rc, ms, mp) if want_msm_inputs else rc
