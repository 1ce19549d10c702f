#!/usr/bin/env python3
"""BASELINE.json configs[0]: one DLEQ proof (benches/dleq.rs:49-90: constraint-API form, G = basepoint, H = hash(G), x = 89327492234) -- create_compact_dleq and
verify_compact_dleq -- through the C ABI: on the host backend (ctx == NULL, no GPU touched), through the GPU (if one is there), and by the oracle's C port.
Median of `reps` after warm-up; one host thread."""
import os, statistics, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from zkp_amd import toolbox as T
from tests import test_gpu_toolbox as R
from oracle import cbind as C

C.build()
x, A, B, G, H = R._capi_points()


def prove(eng):
    p = T.Prover(b"DLEQProof", T.Transcript(b"DLEQTest"), eng)
    vx = p.allocate_scalar(b"x", x)
    vB, _ = p.allocate_point(b"B", B); vH, _ = p.allocate_point(b"H", H); vA, cA = p.allocate_point(b"A", A); vG, cG = p.allocate_point(b"G", G)
    R.dleq_statement(p, vx, vA, vG, vB, vH)
    return p.prove_compact(), cA, cG


def verify(eng, proof, cA, cG):
    v = T.Verifier(b"DLEQProof", T.Transcript(b"DLEQTest"), eng)
    vx = v.allocate_scalar(b"x")
    vB, vH, vA, vG = v.allocate_point(b"B", B), v.allocate_point(b"H", H), v.allocate_point(b"A", cA), v.allocate_point(b"G", cG)
    R.dleq_statement(v, vx, vA, vG, vB, vH)
    v.verify_compact(proof)


def med(f, reps=200):
    for _ in range(10):
        f()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); f(); ts.append(time.perf_counter() - t0)
    return statistics.median(ts) * 1e6


rows = []
host = T.HostEngine()
pr = prove(host)
rows.append(("host backend (ctx = NULL; device headers compiled by g++)", med(lambda: prove(host)), med(lambda: verify(host, *pr))))
cst = C.Statement(b"DLEQProof", ["x"], [("B", False), ("H", False), ("A", False), ("G", False)], [("A", [("x", "B")]), ("G", [("x", "H")])])
sx = np.frombuffer(R.sc(x), np.uint8).reshape(1, 32); pts = np.frombuffer(B + H + A + G, np.uint8).reshape(4, 32)
ec, er, ek, _ = C.prove(cst, b"DLEQTest", sx, pts, bytes(32))
rows.append(("oracle C port (5 x 51-bit limbs, dalek's algorithms), prove / verify_compact", med(lambda: C.prove(cst, b"DLEQTest", sx, pts, bytes(32))),
             med(lambda: C.verify_compact(cst, b"DLEQTest", pts, ec, er)) if hasattr(C, "verify_compact") else float("nan")))
if os.path.exists("/dev/kfd"):
    from zkp_amd.engine import Engine
    eng = Engine(0)
    T.set_host_max_terms(0)
    rows.append(("GPU route (host transcripts, zkp_msm_many on the device)", med(lambda: prove(eng), 100), med(lambda: verify(eng, *pr), 100)))
    T.set_host_max_terms(16)
    rows.append(("default with a context: 2 - 4 terms <= host_max_terms = 16 -> host backend", med(lambda: prove(eng)), med(lambda: verify(eng, *pr))))
print("# BASELINE configs[0]: single DLEQ proof (benches/dleq.rs:49-90), microseconds per call, median; the Python object layer (Prover / Verifier mirrors,")
print("# ctypes) is inside every row except the oracle's.  Reference on its own hardware: ~100 us per proof / verification [SURVEY.md section 6, ESTIMATE].")
print("%-86s %12s %14s" % ("path", "prove_us", "verify_us"))
for name, a, b in rows:
    print("%-86s %12.1f %14.1f" % (name, a, b))
