#!/usr/bin/env python3
"""Generates tests/golden/ristretto_msm.json.  Run in the BUILD container only (it cross-checks the
big-integer model against libsodium 1.0.18 at /opt/conda/lib/libsodium.so, which the GPU box must not
depend on).  Everything is derived from fixed seeds; the output is data (inputs + expected outputs).

    python tests/golden/make_fixtures.py
"""
import ctypes
import hashlib
import json
import os
import random
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import model as M  # noqa: E402

so = ctypes.CDLL("/opt/conda/lib/libsodium.so")
so.sodium_init()


def sodium_msm(scalars, encs):
    """independent evaluation with libsodium: sum of crypto_scalarmult_ristretto255 results"""
    acc = bytes(32)
    for s, e in zip(scalars, encs):
        s %= M.L
        if s == 0:
            continue
        out = ctypes.create_string_buffer(32)
        assert so.crypto_scalarmult_ristretto255(out, s.to_bytes(32, "little"), e) == 0
        acc2 = ctypes.create_string_buffer(32)
        assert so.crypto_core_ristretto255_add(acc2, acc, out.raw) == 0
        acc = acc2.raw
    return acc


rng = random.Random(20260927)
fx = {"generator": "tests/golden/make_fixtures.py", "seed": 20260927, "msm": [], "proofs": [], "decode": []}


def rpoint():
    h = bytes(rng.randrange(256) for _ in range(64))
    out = ctypes.create_string_buffer(32)
    so.crypto_core_ristretto255_from_hash(out, h)
    assert out.raw == M.ristretto_encode(M.ristretto_from_uniform_bytes(h))
    return out.raw


for n in [0, 1, 1, 2, 3, 5, 12, 36, 36, 65, 200]:
    encs = [rpoint() for _ in range(n)]
    scalars = [rng.randrange(M.L) for _ in range(n)]
    if n >= 3:
        scalars[0], scalars[1] = 0, M.L - 1
    exp = M.msm_optional([s.to_bytes(32, "little") for s in scalars], encs)
    assert exp == sodium_msm(scalars, encs), n
    fx["msm"].append({"scalars": [s.to_bytes(32, "little").hex() for s in scalars], "points": [e.hex() for e in encs], "expect": exp.hex()})
# cancellation -> identity encoding
e = rpoint()
fx["msm"].append({"scalars": [(5).to_bytes(32, "little").hex(), (M.L - 5).to_bytes(32, "little").hex()], "points": [e.hex(), e.hex()], "expect": bytes(32).hex()})
# a point that fails to decode -> None
bad = bytes.fromhex("0100000000000000000000000000000000000000000000000000000000000000")
assert M.ristretto_decode(bad) is None and so.crypto_core_ristretto255_is_valid_point(bad) == 0
fx["msm"].append({"scalars": [(1).to_bytes(32, "little").hex()] * 2, "points": [e.hex(), bad.hex()], "expect": None})

# decode decisions on random strings (libsodium ignores bit 255, dalek / RFC 9496 reject it: only compare with the bit clear)
for _ in range(64):
    b = bytes(rng.randrange(256) for _ in range(31)) + bytes([rng.randrange(128)])
    ok = M.ristretto_decode(b) is not None
    assert ok == (so.crypto_core_ristretto255_is_valid_point(b) == 1)
    fx["decode"].append({"enc": b.hex(), "valid": ok})
b = bytes(31) + b"\x80"
fx["decode"].append({"enc": b.hex(), "valid": False})

# whole proofs with injected entropy.  DLEQ uses the reference tests' inputs (tests/zkp.rs:34-39).
G = M.BASEPOINT
H = M.ristretto_hash_from_bytes_sha512(b"A VRF input, for instance")
x = pow(89327492234, M.L - 2, M.L)
st = M.dleq_statement()
pts = {"A": M.pt_mul(x, G), "B": M.pt_mul(x, H), "H": H, "G": G}
for entropy in (bytes(32), bytes(range(32))):
    pr, encs = st.build_prover(M.Transcript(b"DLEQTest"), {"x": x}, pts)
    c, resp, coms, _ = pr._prove_impl(entropy)
    st.build_verifier(M.Transcript(b"DLEQTest"), encs).verify_compact(M.CompactProof(c, resp))
    st.build_verifier(M.Transcript(b"DLEQTest"), encs).verify_batchable(M.BatchableProof(coms, resp), [3, 5])
    fx["proofs"].append({"statement": "dleq", "label": b"DLEQTest".hex(), "secrets": [x.to_bytes(32, "little").hex()],
                         "points": [encs[n].hex() for n in st.instance + st.common], "entropy": entropy.hex(),
                         "challenge": c.to_bytes(32, "little").hex(), "responses": [r.to_bytes(32, "little").hex() for r in resp],
                         "commitments": [k.hex() for k in coms]})
# CMZ'13 n = 10 (benches/zkp.rs:27-46), valid by construction
st = M.cmz_statement(10)
sec = {n: rng.randrange(M.L) for n in st.secrets}
P = {n: M.ristretto_decode(rpoint()) for n in ["P", "Q", "A", "B"] + [f"X_{i}" for i in range(1, 11)]}
for i in range(1, 11):
    P[f"C_{i}"] = M.pt_add(M.pt_mul(sec[f"m_{i}"], P["P"]), M.pt_mul(sec[f"z_{i}"], P["A"]))
P["V"] = M.msm_points([sec[f"m_{i}"] for i in range(1, 11)] + [sec["minus_z_Q"]], [P[f"X_{i}"] for i in range(1, 11)] + [P["Q"]])
entropy = hashlib.sha256(b"cmz fixture").digest()
pr, encs = st.build_prover(M.Transcript(b"Benchmark"), sec, P)
c, resp, coms, _ = pr._prove_impl(entropy)
st.build_verifier(M.Transcript(b"Benchmark"), encs).verify_compact(M.CompactProof(c, resp))
fx["proofs"].append({"statement": "cmz", "label": b"Benchmark".hex(), "secrets": [sec[n].to_bytes(32, "little").hex() for n in st.secrets],
                     "points": [encs[n].hex() for n in st.instance + st.common], "entropy": entropy.hex(),
                     "challenge": c.to_bytes(32, "little").hex(), "responses": [r.to_bytes(32, "little").hex() for r in resp],
                     "commitments": [k.hex() for k in coms]})

out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "ristretto_msm.json")
json.dump(fx, open(out, "w"), indent=1)
print("wrote", out, os.path.getsize(out), "bytes")
