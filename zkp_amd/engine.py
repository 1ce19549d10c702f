"""ctypes binding of the C-ABI library (include/zkp_mi355x.h -> zkp_amd/libzkp_mi355x.so).

This is plumbing only: every function forwards to the HIP library and raises if the library or a
GPU is missing -- there is no CPU fallback anywhere in the product path.
"""
from __future__ import annotations

import ctypes
import os
from typing import Optional, Sequence, Tuple

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libzkp_mi355x.so")
TESTHOOKS_LIB_PATH = os.path.join(_HERE, "libzkp_mi355x_testhooks.so")     # -DZKP_BUILD_TEST_HOOKS build (tests / A-B tools)
ZKP_TESTOPT_DUMMY_LAUNCHES, ZKP_TESTOPT_GENERIC_CLASSIFIER, ZKP_TESTOPT_WAVE_CYCLES = 1001, 1002, 1003
ZKP_OPT_CT_LOOKUP, ZKP_OPT_EACH_STRAUS, ZKP_OPT_LADDER_INTERLEAVE = 9, 10, 11
ZKP_OPT_CT_MASKED_SCANS = ZKP_OPT_CT_LOOKUP           # the round-3 name (value 1 = masked scans)
ZKP_CT_LOOKUP_XBAR, ZKP_CT_LOOKUP_SCAN, ZKP_CT_LOOKUP_LDS = 0, 1, 2
ZKP_OPT_WS_LIMIT_BYTES, ZKP_OPT_JOB_DEFER_D2H, ZKP_OPT_SYNC_SCHEDULE, ZKP_OPT_TRANSCRIPT_STEPS, ZKP_OPT_COMB_SPLIT, ZKP_OPT_JOINT_LADDER = 12, 13, 14, 15, 16, 17

ZKP_VARTIME = 0
ZKP_CT = 1
ZKP_OPT_BATCH_ENCODE_MIN = 1
K_NAMES = ("decode", "terms", "reduce", "sort", "bucket", "combine", "transcript", "scalars", "tables")

EXPORTS = (
    "zkp_ctx_create", "zkp_ctx_destroy", "zkp_ctx_set_stream", "zkp_ctx_synchronize", "zkp_last_error",
    "zkp_version", "zkp_ctx_set_option", "zkp_msm_many", "zkp_msm_many_dev", "zkp_msm_optional", "zkp_msm_optional_dev",
    "zkp_decode_check", "zkp_encode_many", "zkp_ctx_last_timing", "zkp_ctx_set_profiling",
    "zkp_ctx_prepare_fixed_points", "zkp_batch_check", "zkp_fused_prove", "zkp_fused_verify_compact", "zkp_fused_batch_verify",
    "zkp_fused_verify_batchable", "zkp_fused_prove_dev", "zkp_fused_verify_compact_dev", "zkp_fused_verify_batchable_dev", "zkp_fused_batch_verify_dev",
    "zkp_fused_verify_batchable_coeffs", "zkp_fused_batch_verify_many", "zkp_fused_batch_verify_many_dev", "zkp_ctx_capture_begin", "zkp_ctx_capture_end", "zkp_ctx_capture_abort", "zkp_graph_launch", "zkp_graph_destroy",
    "zkp_fused_prove_submit", "zkp_fused_verify_compact_submit", "zkp_fused_batch_verify_many_submit", "zkp_fused_verify_batchable_submit",
    "zkp_fused_prove_seeded", "zkp_fused_batch_verify_many_seeded", "zkp_ctx_job_wait", "zkp_ctx_job_poll", "zkp_ctx_job_pending", "zkp_ctx_job_discard", "zkp_ctx_job_timing", "zkp_ctx_last_kernels", "zkp_host_alloc", "zkp_host_alloc_on", "zkp_host_numa_node", "zkp_host_node_of", "zkp_host_free", "zkp_host_register", "zkp_host_unregister",
    "zkp_host_is_pinned", "zkp_chacha20_fill_dev",
)
TEST_HOOK_EXPORTS = ("zkp_debug_quad_selftest", "zkp_debug_row_selftest", "zkp_debug_wave_cycles")      # only in libzkp_mi355x_testhooks.so


class ZkpError(RuntimeError):
    pass


_lib = None
_hooks_lib = None


def load_library(test_hooks: bool = False) -> ctypes.CDLL:
    """dlopen the HIP library; raises (never falls back) when it has not been built.  test_hooks = the -DZKP_BUILD_TEST_HOOKS
    build of the same sources (a second, independent copy of the library: its contexts must not be handed to libzkp_toolbox.so)."""
    global _lib, _hooks_lib
    if test_hooks and _hooks_lib is not None:
        return _hooks_lib
    if not test_hooks and _lib is not None:
        return _lib
    path = TESTHOOKS_LIB_PATH if test_hooks else LIB_PATH
    if not os.path.exists(path):
        raise ZkpError(f"{path} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                       "(hipcc --offload-arch=gfx950); there is no CPU fallback")
    lib = ctypes.CDLL(path)
    vp, u8p, u32p, i32 = ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int
    lib.zkp_ctx_create.argtypes = [ctypes.POINTER(vp), i32]
    lib.zkp_ctx_destroy.argtypes = [vp]
    lib.zkp_ctx_destroy.restype = None
    lib.zkp_ctx_set_stream.argtypes = [vp, vp]
    lib.zkp_ctx_synchronize.argtypes = [vp]
    lib.zkp_last_error.restype = ctypes.c_char_p
    lib.zkp_version.restype = ctypes.c_char_p
    lib.zkp_msm_many.argtypes = [vp, ctypes.c_uint32, u32p, u8p, u32p, u8p, ctypes.c_uint32, i32, u8p, u8p]
    lib.zkp_msm_many_dev.argtypes = [vp, ctypes.c_uint32, u32p, u8p, u32p, u8p, ctypes.c_uint32, ctypes.c_uint32, i32, u8p, u8p]
    lib.zkp_msm_optional.argtypes = [vp, ctypes.c_uint64, u8p, u8p, u8p, ctypes.POINTER(i32)]
    lib.zkp_msm_optional_dev.argtypes = [vp, ctypes.c_uint64, u8p, u8p, u8p, u32p]
    lib.zkp_decode_check.argtypes = [vp, ctypes.c_uint64, u8p, u8p, u8p]
    lib.zkp_encode_many.argtypes = [vp, ctypes.c_uint64, u8p, u8p]
    lib.zkp_ctx_last_timing.argtypes = [vp, ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_float)]
    lib.zkp_ctx_set_profiling.argtypes = [vp, i32]
    lib.zkp_ctx_capture_begin.argtypes = [vp]
    lib.zkp_ctx_capture_end.argtypes = [vp, ctypes.POINTER(vp)]
    lib.zkp_graph_launch.argtypes = [vp, vp]
    lib.zkp_graph_destroy.argtypes = [vp]
    lib.zkp_graph_destroy.restype = None
    lib.zkp_ctx_prepare_fixed_points.argtypes = [vp, ctypes.c_uint32, u8p]
    lib.zkp_ctx_capture_abort.argtypes = [vp]
    if test_hooks:
        lib.zkp_debug_quad_selftest.argtypes = [vp, ctypes.c_uint32, u8p, u8p]
        lib.zkp_debug_row_selftest.argtypes = [vp, ctypes.c_uint32, u8p, u8p]
        lib.zkp_debug_wave_cycles.argtypes = [vp, ctypes.c_void_p, ctypes.c_uint32]
        _hooks_lib = lib
    else:
        _lib = lib
    return lib


def _check(rc: int, what: str) -> None:
    if rc != 0:
        raise ZkpError(f"{what} failed with code {rc}: {load_library().zkp_last_error().decode()}")


def _u8(a, shape_last: int) -> np.ndarray:
    a = np.ascontiguousarray(a, dtype=np.uint8)
    if a.ndim != 2 or a.shape[1] != shape_last:
        raise ValueError(f"expected uint8 array of shape [n][{shape_last}], got {a.shape}")
    return a


def _ptr(a: Optional[np.ndarray]):
    return None if a is None else a.ctypes.data_as(ctypes.c_void_p)


class Engine:
    """One context on one GPU (HIP device ordinal `device`)."""

    def __init__(self, device: int = 0, test_hooks: bool = False):
        self._lib = load_library(test_hooks)
        self.test_hooks = test_hooks
        h = ctypes.c_void_p()
        _check(self._lib.zkp_ctx_create(ctypes.byref(h), device), "zkp_ctx_create")
        self._h = h
        self.device = device

    def close(self) -> None:
        if getattr(self, "_h", None):
            self._lib.zkp_ctx_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def version(self) -> str:
        return self._lib.zkp_version().decode()

    # ---- host-buffer entry points (numpy) ---------------------------------------------------
    def msm_many(self, off: Sequence[int], scalars, pidx: Sequence[int], points, flags: int = ZKP_VARTIME
                 ) -> Tuple[np.ndarray, np.ndarray]:
        """CSR batch of small MSMs -> (out[n_msm][32], status[n_msm])."""
        off = np.ascontiguousarray(off, dtype=np.uint32)
        n_msm = len(off) - 1
        pidx = np.ascontiguousarray(pidx, dtype=np.uint32)
        n_terms = int(off[-1]) if n_msm >= 0 and len(off) else 0
        scalars = _u8(scalars, 32) if n_terms else np.zeros((0, 32), np.uint8)
        points = _u8(points, 32) if len(points) else np.zeros((0, 32), np.uint8)
        if len(scalars) != n_terms or len(pidx) != n_terms:
            raise ValueError("scalars / pidx length must equal off[-1]")
        out = np.zeros((max(n_msm, 0), 32), np.uint8)
        status = np.zeros(max(n_msm, 0), np.uint8)
        _check(self._lib.zkp_msm_many(self._h, n_msm, _ptr(off), _ptr(scalars), _ptr(pidx), _ptr(points),
                                      len(points), flags, _ptr(out), _ptr(status)), "zkp_msm_many")
        return out, status

    def msm_optional(self, scalars, points) -> Optional[bytes]:
        """optional_multiscalar_mul: encoding of sum s_i * decode(P_i), or None if a decode fails."""
        scalars = _u8(scalars, 32) if len(scalars) else np.zeros((0, 32), np.uint8)
        points = _u8(points, 32) if len(points) else np.zeros((0, 32), np.uint8)
        if len(scalars) != len(points):
            raise ValueError("scalars and points must have equal length")
        out = np.zeros(32, np.uint8)
        st = ctypes.c_int(1)
        _check(self._lib.zkp_msm_optional(self._h, len(scalars), _ptr(scalars), _ptr(points), _ptr(out), ctypes.byref(st)),
               "zkp_msm_optional")
        return None if st.value else out.tobytes()

    def decode_check(self, points, want_coords: bool = False):
        points = _u8(points, 32) if len(points) else np.zeros((0, 32), np.uint8)
        status = np.zeros(len(points), np.uint8)
        xyzt = np.zeros((len(points), 128), np.uint8) if want_coords else None
        _check(self._lib.zkp_decode_check(self._h, len(points), _ptr(points), _ptr(status), _ptr(xyzt)), "zkp_decode_check")
        return (status, xyzt) if want_coords else status

    def encode_many(self, xyzt) -> np.ndarray:
        xyzt = _u8(xyzt, 128) if len(xyzt) else np.zeros((0, 128), np.uint8)
        out = np.zeros((len(xyzt), 32), np.uint8)
        _check(self._lib.zkp_encode_many(self._h, len(xyzt), _ptr(xyzt), _ptr(out)), "zkp_encode_many")
        return out

    def debug_quad_selftest(self, pairs) -> np.ndarray:
        pairs = _u8(pairs, 64)
        out = np.zeros((len(pairs), 4, 32), np.uint8)
        if not self.test_hooks:
            raise ZkpError("zkp_debug_quad_selftest exists in the test-hook build only: Engine(device, test_hooks=True)")
        _check(self._lib.zkp_debug_quad_selftest(self._h, len(pairs), _ptr(pairs), _ptr(out)), "zkp_debug_quad_selftest")
        return out

    def debug_row_selftest(self, pairs) -> np.ndarray:
        """(test-hook build) one-limb-per-lane point arithmetic (csrc/rowfe.h): [n][64] encodings (P, Q) -> [n][3][32] = enc(2P), enc(P+Q), enc(2^11 P + Q)"""
        pairs = _u8(pairs, 64)
        out = np.zeros((len(pairs), 3, 32), np.uint8)
        if not self.test_hooks:
            raise ZkpError("zkp_debug_row_selftest exists in the test-hook build only: Engine(device, test_hooks=True)")
        _check(self._lib.zkp_debug_row_selftest(self._h, len(pairs), _ptr(pairs), _ptr(out)), "zkp_debug_row_selftest")
        return out

    def debug_wave_cycles(self, cap: int = 1 << 20):
        """(test-hook build, after set_option(ZKP_TESTOPT_WAVE_CYCLES, 1)) -> (class[n], cycles[n]) of the term kernel's wavefronts since the last read"""
        if not self.test_hooks:
            raise ZkpError("zkp_debug_wave_cycles exists in the test-hook build only: Engine(device, test_hooks=True)")
        buf = np.zeros(cap, np.uint64)
        n = self._lib.zkp_debug_wave_cycles(self._h, _ptr(buf), ctypes.c_uint32(cap))
        if n < 0:
            _check(n, "zkp_debug_wave_cycles")
        buf = buf[:n]
        buf = buf[buf != 0]
        return (buf >> np.uint64(56)).astype(np.int64), (buf & np.uint64((1 << 56) - 1)).astype(np.int64)

    def prepare_fixed_points(self, encodings) -> None:
        """Hint: these points (the statement's common / static points) will be referenced by many terms."""
        encodings = _u8(encodings, 32) if len(encodings) else np.zeros((0, 32), np.uint8)
        _check(self._lib.zkp_ctx_prepare_fixed_points(self._h, len(encodings), _ptr(encodings)), "zkp_ctx_prepare_fixed_points")

    # ---- device-buffer entry points (raw device pointers, e.g. torch tensor .data_ptr()) ----
    def set_stream(self, hip_stream: int) -> None:
        _check(self._lib.zkp_ctx_set_stream(self._h, ctypes.c_void_p(hip_stream)), "zkp_ctx_set_stream")

    def synchronize(self) -> None:
        _check(self._lib.zkp_ctx_synchronize(self._h), "zkp_ctx_synchronize")

    def msm_many_dev(self, n_msm, d_off, d_scalars, d_pidx, d_points, n_points, n_terms, flags, d_out, d_status) -> None:
        _check(self._lib.zkp_msm_many_dev(self._h, n_msm, d_off, d_scalars, d_pidx, d_points, n_points, n_terms,
                                          flags, d_out, d_status), "zkp_msm_many_dev")

    def msm_optional_dev(self, n, d_scalars, d_points, d_out, d_status) -> None:
        _check(self._lib.zkp_msm_optional_dev(self._h, n, d_scalars, d_points, d_out, d_status), "zkp_msm_optional_dev")

    # ---- fused statement flows on device-resident buffers (include/zkp_mi355x.h section 2c) -------------
    def fused_prove_dev(self, fst: "FusedStatement", n, strobe_pos, d_ts, d_secrets, d_table, d_entropy, d_chal, d_resp, d_coms, d_status) -> None:
        _check(self._lib.zkp_fused_prove_dev(self._h, ctypes.byref(fst.c), ctypes.c_uint32(n), ctypes.c_uint32(strobe_pos),
                                             *[ctypes.c_void_p(x) for x in (d_ts, d_secrets, d_table, d_entropy, d_chal, d_resp, d_coms, d_status)]),
               "zkp_fused_prove_dev")

    def fused_verify_batchable_dev(self, fst: "FusedStatement", n, strobe_pos, d_ts, d_table, d_resp, d_w, d_results) -> None:
        """d_table = common || instance rows || commitments [n][nc]; d_w [n][nc][16]; d_results [n] bytes (0 = verified)"""
        _check(self._lib.zkp_fused_verify_batchable_dev(self._h, ctypes.byref(fst.c), ctypes.c_uint32(n), ctypes.c_uint32(strobe_pos),
                                                        *[ctypes.c_void_p(x) for x in (d_ts, d_table, d_resp, d_w, d_results)]),
               "zkp_fused_verify_batchable_dev")

    def fused_verify_compact_dev(self, fst: "FusedStatement", n, strobe_pos, d_ts, d_table, d_chal, d_resp, d_results) -> None:
        _check(self._lib.zkp_fused_verify_compact_dev(self._h, ctypes.byref(fst.c), ctypes.c_uint32(n), ctypes.c_uint32(strobe_pos),
                                                      *[ctypes.c_void_p(x) for x in (d_ts, d_table, d_chal, d_resp, d_results)]),
               "zkp_fused_verify_compact_dev")

    def fused_batch_verify_dev(self, fst: "FusedStatement", n, strobe_pos, d_ts, d_points, d_coms, d_resp, d_w, d_out, d_status) -> None:
        _check(self._lib.zkp_fused_batch_verify_dev(self._h, ctypes.byref(fst.c), ctypes.c_uint32(n), ctypes.c_uint32(strobe_pos),
                                                    *[ctypes.c_void_p(x) for x in (d_ts, d_points, d_coms, d_resp, d_w, d_out, d_status)]),
               "zkp_fused_batch_verify_dev")

    def fused_batch_verify_many_dev(self, fst: "FusedStatement", n_batches, n_each, strobe_pos, d_ts, d_points, d_coms, d_resp, d_w, d_out, d_status) -> None:
        """K batch verifications of n_each proofs in one pass; d_out [K][32], d_status [K][2] int32 (zkp_fused_batch_verify_many_dev)."""
        _check(self._lib.zkp_fused_batch_verify_many_dev(self._h, ctypes.byref(fst.c), ctypes.c_uint32(n_batches), ctypes.c_uint32(n_each), ctypes.c_uint32(strobe_pos),
                                                         *[ctypes.c_void_p(x) for x in (d_ts, d_points, d_coms, d_resp, d_w, d_out, d_status)]),
               "zkp_fused_batch_verify_many_dev")

    def fused_batch_verify_many(self, fst: "FusedStatement", n_batches: int, transcripts, inst, common, commitments, responses, weights16, want_coeffs: bool = False):
        """zkp_fused_batch_verify_many on host arrays: the len(transcripts) = n_batches * N_each proofs lie next to each other, batch b =
        proofs [b N_each, (b + 1) N_each) -> verdicts[n_batches] (0 = the batch verifies) [, the coefficient vector the device built];
        the transcripts [N][208] are advanced in place."""
        n = len(transcripts)
        if n_batches <= 0 or n % n_batches:
            raise ValueError("the number of proofs must be a multiple of n_batches")
        n_each = n // n_batches
        k = n_batches * fst.n_static + (fst.n_instance + len(fst._lhs)) * n
        verdicts = (ctypes.c_int * n_batches)(*([1] * n_batches))
        co = np.zeros((k, 32), np.uint8) if want_coeffs else None
        arrs = [np.ascontiguousarray(a, dtype=np.uint8) for a in (transcripts, inst, common, commitments, responses, weights16)]
        if arrs[1].shape != (fst.n_instance, n, 32) or arrs[2].shape != (fst.n_static, 32) or arrs[3].shape != (n, len(fst._lhs), 32) or \
                arrs[4].shape[:1] != (n,) or arrs[5].shape != (len(fst._lhs), n, 16) or arrs[0].shape != (n, 208):
            raise ValueError("array shapes do not match the statement / batch sizes")
        ptr = lambda a: a.ctypes.data_as(ctypes.c_void_p)
        _check(self._lib.zkp_fused_batch_verify_many(self._h, ctypes.byref(fst.c), ctypes.c_uint32(n_batches), ctypes.c_uint32(n_each), ptr(arrs[0]), ptr(arrs[1]),
                                                     ptr(arrs[2]), ptr(arrs[3]), ptr(arrs[4]), ptr(arrs[5]), verdicts, _ptr(co)), "zkp_fused_batch_verify_many")
        transcripts[...] = arrs[0]
        v = np.array(list(verdicts), np.int32)
        return (v, co) if want_coeffs else v

    def fused_verify_batchable_coeffs(self, fst: "FusedStatement", transcripts, inst, common, commitments, responses, weights16):
        """zkp_fused_verify_batchable_coeffs on host arrays -> (results[N], coefficient vectors [N][np + nc][32]); the
        transcripts [N][208] are advanced in place."""
        n = len(transcripts)
        k = fst.n_static + fst.n_instance + len(fst._lhs)
        res = np.ones(n, np.uint8)
        co = np.zeros((n, k, 32), np.uint8)
        arrs = [np.ascontiguousarray(a, dtype=np.uint8) for a in (transcripts, inst, common, commitments, responses, weights16)]
        ptr = lambda a: a.ctypes.data_as(ctypes.c_void_p)
        _check(self._lib.zkp_fused_verify_batchable_coeffs(self._h, ctypes.byref(fst.c), ctypes.c_uint32(n), ptr(arrs[0]), ptr(arrs[1]), ptr(arrs[2]),
                                                           ptr(arrs[3]), ptr(arrs[4]), ptr(arrs[5]), ptr(res), ptr(co)), "zkp_fused_verify_batchable_coeffs")
        transcripts[...] = arrs[0]
        return res, co

    @property
    def ct_lookups(self):
        """the values of ZKP_OPT_CT_LOOKUP this build of the library has: (0,) = lane crossbar only (7-bit fixed-base windows, the shipped shape);
        (0, 1, 2) in a -DZKP_HOT_W=6 build (masked scans and the LDS rows of rounds 2 - 4 exist for the 6-bit window only)"""
        return (0, 1, 2) if "6-bit" in self.version else (0,)

    def set_option(self, option: int, value: int) -> None:
        """Tuning knobs of include/zkp_mi355x.h (ZKP_OPT_*); results never depend on them."""
        _check(self._lib.zkp_ctx_set_option(self._h, ctypes.c_int(option), ctypes.c_uint64(value)), "zkp_ctx_set_option")

    def capture_begin(self) -> None:
        """Start recording what is enqueued on this context's stream into a HIP graph (zkp_ctx_capture_begin)."""
        _check(self._lib.zkp_ctx_capture_begin(self._h), "zkp_ctx_capture_begin")

    def capture_end(self) -> "Graph":
        g = ctypes.c_void_p()
        _check(self._lib.zkp_ctx_capture_end(self._h, ctypes.byref(g)), "zkp_ctx_capture_end")
        return Graph(self, g)

    def capture_abort(self) -> None:
        """End and discard a capture (after a failed call inside it); the context is usable again."""
        _check(self._lib.zkp_ctx_capture_abort(self._h), "zkp_ctx_capture_abort")

    def capture(self):
        """Context manager around capture_begin / capture_end: `with eng.capture() as cap: ...calls...` then `cap.graph`.  An
        exception inside the block aborts the capture (zkp_ctx_capture_abort) instead of leaving the stream in capture mode."""
        return _Capture(self)

    def set_profiling(self, enabled: bool) -> None:
        _check(self._lib.zkp_ctx_set_profiling(self._h, int(enabled)), "zkp_ctx_set_profiling")

    def last_timing(self):
        arr = (ctypes.c_float * len(K_NAMES))()
        tot = ctypes.c_float()
        rc = self._lib.zkp_ctx_last_timing(self._h, arr, ctypes.byref(tot))
        if rc < 0:
            _check(rc, "zkp_ctx_last_timing")
        return {k: float(arr[i]) for i, k in enumerate(K_NAMES)}, float(tot.value)

    def last_kernels(self):
        """{timing kind: [kernel names as rocprofv3 prints them]} for the kinds whose kernel variant the last call picked at run time"""
        out = {}
        buf = ctypes.create_string_buffer(512)
        self._lib.zkp_ctx_last_kernels.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_char_p, ctypes.c_size_t]
        for i, k in enumerate(K_NAMES):
            if self._lib.zkp_ctx_last_kernels(self._h, i, buf, 512) > 0:
                out[k] = buf.value.decode().split(";")
        return out


class _Capture:
    def __init__(self, eng: "Engine"):
        self._eng, self.graph = eng, None

    def __enter__(self):
        self._eng.capture_begin()
        return self

    def __exit__(self, exc_type, exc, tb):
        if exc_type is not None:
            self._eng.capture_abort()
            return False
        self.graph = self._eng.capture_end()
        return False


class Graph:
    """zkp_graph: a recorded chain of *_dev calls, replayed with one host call.  launch() raises ZkpError ("stale graph") when
    the context's workspace or plans changed after the capture (see zkp_mi355x.h, HIP graphs: lifetime rules)."""

    def __init__(self, eng: "Engine", handle):
        self._eng, self._h = eng, handle

    def launch(self) -> None:
        _check(self._eng._lib.zkp_graph_launch(self._h, self._eng._h), "zkp_graph_launch")

    def close(self) -> None:
        if self._h:
            self._eng._lib.zkp_graph_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class _BatchStatementC(ctypes.Structure):
    _fields_ = [("n_secrets", ctypes.c_uint32), ("n_static", ctypes.c_uint32), ("n_instance", ctypes.c_uint32),
                ("n_constraints", ctypes.c_uint32), ("cons_lhs", ctypes.c_void_p), ("cons_off", ctypes.c_void_p),
                ("cons_sc", ctypes.c_void_p), ("cons_pt", ctypes.c_void_p)]


class _FusedStatementC(ctypes.Structure):
    _fields_ = [("shape", _BatchStatementC), ("label", ctypes.c_char_p), ("secret_labels", ctypes.POINTER(ctypes.c_char_p)),
                ("point_labels", ctypes.POINTER(ctypes.c_char_p)), ("alloc_order", ctypes.c_void_p), ("alloc_seq", ctypes.c_void_p)]


class FusedStatement:
    """zkp_fused_statement (include/zkp_mi355x.h): the statement in point-id form (static ids first, then instance ids)
    with its transcript labels.  points = [(label, is_common)] in allocation order; constraints = [(lhs, [(secret, point)])]
    with indices into `secrets` / `points`, exactly as Prover/Verifier::constrain receives them.  alloc_seq (optional) =
    the caller's allocation calls in order, [("s", secret index) | ("p", point index)], when scalars and points are
    interleaved; default: every secret before the first point (define_proof!'s order)."""

    def __init__(self, proof_label: bytes, secrets, points, constraints, alloc_seq=None):
        ns = sum(1 for _, c in points if c)
        rank, k_c, k_i = [], 0, 0
        for _, c in points:
            if c:
                rank.append(k_c); k_c += 1
            else:
                rank.append(ns + k_i); k_i += 1
        self.n_static, self.n_instance = k_c, k_i
        self._lhs = np.array([rank[l] for l, _ in constraints], np.uint32)
        off, sc, pt = [0], [], []
        for _, lc in constraints:
            for s_, p_ in lc:
                sc.append(s_); pt.append(rank[p_])
            off.append(len(sc))
        self._off = np.array(off, np.uint32)
        self._sc = np.array(sc, np.uint32)
        self._pt = np.array(pt, np.uint32)
        self._order = np.array(rank, np.uint32)
        plabels = [None] * len(points)
        for (name, _), r in zip(points, rank):
            plabels[r] = bytes(name)
        self._sl = (ctypes.c_char_p * max(1, len(secrets)))(*[bytes(x) for x in secrets])
        self._pl = (ctypes.c_char_p * max(1, len(points)))(*plabels)
        self._label = bytes(proof_label)
        vp = lambda a: a.ctypes.data_as(ctypes.c_void_p)
        self.c = _FusedStatementC(_BatchStatementC(len(secrets), k_c, k_i, len(constraints), vp(self._lhs), vp(self._off), vp(self._sc), vp(self._pt)),
                                  self._label, self._sl, self._pl, vp(self._order), None)
        if alloc_seq is not None:
            self._seq = np.array([(0x80000000 | i) if kind == "s" else rank[i] for kind, i in alloc_seq], np.uint32)
            self.c.alloc_seq = vp(self._seq)
