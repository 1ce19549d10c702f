"""The host backend of the toolbox (zkp_amd/csrc/host/host_backend.cpp): the kernels' own field / group headers compiled for the host,
behind the same zkp_toolbox.h calls, selected by ctx == NULL (no GPU needed) or by size (zkp_toolbox_set_host_max_terms).

BASELINE.json configs[0] as written -- "DLEQ proof (benches/dleq.rs) single prove + verify on CPU (plumbing, no GPU)" -- runs here
through the C ABI with no GPU visible: bytes equal to the oracle's, the reference's own tests (tests/zkp.rs,
tests/dleq_using_constraint_api.rs, tests/sig_and_vrf_example.rs) pass on it, the committed golden proofs are reproduced, rejected
proofs are rejected.  The -m gpu test at the end checks host == device byte for byte."""
import numpy as np
import pytest

from oracle import cbind as C
from oracle import model as M
from zkp_amd import toolbox as T
from tests import test_gpu_toolbox as R          # the mirrors of the reference's tests take the engine as an argument

LABEL = b"DLEQTest"


@pytest.fixture(scope="module")
def host():
    C.build()
    return T.HostEngine()


def test_config1_dleq_prove_and_verify_on_the_host_equals_oracle(host):
    """benches/dleq.rs:51-56 / tests/dleq_using_constraint_api.rs:41-56: G = basepoint, H = hash(G), x = 89327492234, labels DLEQTest / DLEQProof"""
    x, A, B, G, H = R._capi_points()
    entropy = bytes(range(32))
    prover = T.Prover(b"DLEQProof", T.Transcript(LABEL), host)
    var_x = prover.allocate_scalar(b"x", x)
    var_B, _ = prover.allocate_point(b"B", B)
    var_H, _ = prover.allocate_point(b"H", H)
    var_A, cmpr_A = prover.allocate_point(b"A", A)
    var_G, cmpr_G = prover.allocate_point(b"G", G)
    R.dleq_statement(prover, var_x, var_A, var_G, var_B, var_H)
    proof = prover.prove_compact(entropy=entropy)
    # the oracle's prover on the same statement (allocation order x, B, H, A, G), secrets and entropy
    cst = C.Statement(b"DLEQProof", ["x"], [("B", False), ("H", False), ("A", False), ("G", False)], [("A", [("x", "B")]), ("G", [("x", "H")])])
    ec, er, ek, _ = C.prove(cst, LABEL, np.frombuffer(R.sc(x), np.uint8).reshape(1, 32), np.frombuffer(B + H + A + G, np.uint8).reshape(4, 32), entropy)
    assert proof.challenge == ec.tobytes() and proof.responses[0] == er[0].tobytes()
    verifier = T.Verifier(b"DLEQProof", T.Transcript(LABEL), host)
    var_x = verifier.allocate_scalar(b"x")
    var_B = verifier.allocate_point(b"B", B)
    var_H = verifier.allocate_point(b"H", H)
    var_A = verifier.allocate_point(b"A", cmpr_A)
    var_G = verifier.allocate_point(b"G", cmpr_G)
    R.dleq_statement(verifier, var_x, var_A, var_G, var_B, var_H)
    verifier.verify_compact(proof)
    bad = T.CompactProof(proof.challenge, [R.sc(int.from_bytes(proof.responses[0], "little") + 1)])
    v2 = T.Verifier(b"DLEQProof", T.Transcript(LABEL), host)
    vx = v2.allocate_scalar(b"x")
    vB, vH, vA, vG = v2.allocate_point(b"B", B), v2.allocate_point(b"H", H), v2.allocate_point(b"A", cmpr_A), v2.allocate_point(b"G", cmpr_G)
    R.dleq_statement(v2, vx, vA, vG, vB, vH)
    with pytest.raises(T.VerificationFailure):
        v2.verify_compact(bad)


@pytest.mark.parametrize("name", ["test_create_and_verify_compact", "test_create_and_verify_batchable", "test_create_batch_and_batch_verify",
                                  "test_create_and_verify_compact_dleq", "test_create_and_verify_batchable_dleq", "test_create_batch_and_batch_verify_dleq",
                                  "test_create_and_verify_sig", "test_counterparty_signature_chain", "test_create_and_verify_vrf",
                                  "test_non_canonical_challenge_is_rejected", "test_unreferenced_point_must_still_decode"])
def test_the_reference_tests_pass_on_the_host_backend(host, name):
    """every mirror of the reference's tests in tests/test_gpu_toolbox.py, with the host backend where the GPU engine stood"""
    getattr(R, name)(host)


def test_golden_proofs_byte_exact_on_the_host(host):
    """tests/golden/ristretto_msm.json (made with libsodium's independent ristretto255): whole DLEQ / CMZ proofs with injected entropy"""
    import json
    fx = json.load(open(R.GOLDEN))
    for case in fx["proofs"]:
        mod = T.dleq_module() if case["statement"] == "dleq" else T.cmz_module(10)
        names = mod.instance + mod.common
        points = {n: bytes.fromhex(p) for n, p in zip(names, case["points"])}
        secrets = {n: bytes.fromhex(s) for n, s in zip(mod.secrets, case["secrets"])}
        label = bytes.fromhex(case["label"])
        proof = mod.prove_compact(host, T.Transcript(label), secrets, points, entropy=bytes.fromhex(case["entropy"]))
        assert proof.challenge.hex() == case["challenge"] and [r.hex() for r in proof.responses] == case["responses"]
        bp = mod.prove_batchable(host, T.Transcript(label), secrets, points, entropy=bytes.fromhex(case["entropy"]))
        assert [c.hex() for c in bp.commitments] == case["commitments"]
        mod.verify_compact(host, proof, T.Transcript(label), points)
        mod.verify_batchable(host, bp, T.Transcript(label), points)


def _cmz(n, seed):
    mod, secrets, inst, common = R._cmz_batch(n, seed)
    rng = np.random.default_rng(seed + 1)
    return mod, secrets, inst, common, rng.integers(0, 256, size=(n, 32), dtype=np.uint8), rng.integers(0, 256, size=(mod.statement.nc, n, 16), dtype=np.uint8)


def test_cmz_batch_calls_on_the_host_equal_the_oracle(host):
    n = 3
    mod, secrets, inst, common, entropy, w = _cmz(n, 5)
    st, label = mod.statement, b"Benchmark"
    t0 = lambda: np.stack([T.Transcript(label).state] * n)      # noqa: E731
    chal, resp, coms = T.prove_batch(host, st, t0(), secrets, inst, common, entropy)
    cst = C.Statement.from_model(M.cmz_statement(10))
    for j in range(n):
        ec, er, ek, _ = C.prove(cst, label, secrets[j], np.concatenate([inst[:, j], common]), entropy[j].tobytes())
        assert chal[j].tobytes() == ec.tobytes() and (resp[j] == er).all() and (coms[j] == ek).all()
    assert not T.verify_compact_batch(host, st, t0(), inst, common, chal, resp).any()
    assert not T.verify_batchable_each(host, st, t0(), inst, common, coms, resp).any()
    T.batch_verify(host, st, t0(), inst, common, coms, resp, w)
    assert C.batch_verify(cst, label, n, inst, common, coms, resp, w) == 0
    ok, co = T.batch_verify_coeffs(host, st, t0(), inst, common, coms, resp, w)
    ms, mp = T.batch_verify_build(st, t0(), inst, common, coms, resp, w)
    assert ok and (co == ms).all()
    # rejected proofs are rejected: a flipped response bit, an undecodable point, the identity, a non-canonical response
    junk = np.frombuffer(bytes([1] + [0] * 31), np.uint8)
    bad = resp.copy(); bad[1, 4, 0] ^= 1
    assert list(T.verify_compact_batch(host, st, t0(), inst, common, chal, bad)) == [0, 1, 0]
    assert list(T.verify_batchable_each(host, st, t0(), inst, common, coms, bad)) == [0, 1, 0]
    with pytest.raises(T.VerificationFailure):
        T.batch_verify(host, st, t0(), inst, common, coms, bad, w)
    ok, res = T.batch_verify_locate(host, st, t0(), inst, common, coms, bad)
    assert not ok and list(res) == [0, 1, 0]
    ji = inst.copy(); ji[11, 2] = junk
    assert list(T.verify_compact_batch(host, st, t0(), ji, common, chal, resp)) == [0, 0, 1]
    with pytest.raises(T.VerificationFailure):
        T.batch_verify(host, st, t0(), ji, common, coms, resp, w)
    zc = coms.copy(); zc[0, 3] = 0
    assert list(T.verify_batchable_each(host, st, t0(), inst, common, zc, resp)) == [1, 0, 0]
    big = resp.copy()
    big[2, 0] = np.frombuffer((int.from_bytes(resp[2, 0].tobytes(), "little") + M.L).to_bytes(32, "little"), np.uint8)
    assert list(T.verify_compact_batch(host, st, t0(), inst, common, chal, big)) == [0, 0, 1]
    v = T.batch_verify_many(host, st, 3, t0(), inst, common, coms, bad, w)
    assert list(v) == [0, 1, 0]


def test_host_msm_matches_the_golden_msm_and_decode_vectors(host):
    """the fixtures' MSMs (libsodium) through the host backend's batch verifier path: a one-constraint statement per vector is overkill --
    the backend's arithmetic is pinned through whole proofs above; here the RFC 9496 A.2 invalid encodings must be refused as points"""
    import json
    fx = json.load(open(R.GOLDEN))
    mod = T.dleq_module()
    x, points = R._dleq_assignments()
    proof = R.dleq.prove_compact(host, T.Transcript(LABEL), {"x": x}, points)
    for c in fx["decode"]:
        if c["valid"]:
            continue
        pts = dict(points, H=bytes.fromhex(c["enc"]))
        with pytest.raises(T.VerificationFailure):
            R.dleq.verify_compact(host, proof, T.Transcript(LABEL), pts)
    assert mod is not None


@pytest.mark.gpu
def test_host_backend_equals_device_path():
    from zkp_amd.engine import Engine
    eng = Engine(0)
    host = T.HostEngine()
    try:
        for n, seed in ((1, 3), (2, 4)):
            mod, secrets, inst, common, entropy, w = _cmz(n, seed)
            st, label = mod.statement, b"Benchmark"
            t0 = lambda: np.stack([T.Transcript(label).state] * n)      # noqa: E731
            T.set_host_max_terms(0)
            a = T.prove_batch(eng, st, t0(), secrets, inst, common, entropy)
            T.set_host_max_terms(1 << 20)                               # by size, with a context
            b = T.prove_batch(eng, st, t0(), secrets, inst, common, entropy)
            c_ = T.prove_batch(host, st, t0(), secrets, inst, common, entropy)
            for u, v, z in zip(a, b, c_):
                assert (u == v).all() and (u == z).all()
            for e_ in (eng, host):
                assert not T.verify_compact_batch(e_, st, t0(), inst, common, a[0], a[1]).any()
                T.batch_verify(e_, st, t0(), inst, common, a[2], a[1], w)
    finally:
        T.set_host_max_terms(16)
        eng.close()


def test_c_example_links_the_libraries_and_matches_the_python_layer(host, tmp_path):
    """examples/dleq_c_abi.c: plain C99 against include/zkp_toolbox.h and the two shared libraries (what a `-sys` crate links), host backend: builds
    warning-free, every check of the example passes, and the proof it prints is the proof of the Python object layer (= the oracle's, first test of this
    file) for the same secret, points and entropy."""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "dleq_c_abi")
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Wextra", "-Werror", "-pedantic", "-I", os.path.join(root, "include"), os.path.join(root, "examples", "dleq_c_abi.c"),
                           "-L", os.path.join(root, "zkp_amd"), "-lzkp_toolbox", "-lzkp_mi355x", "-Wl,-rpath," + os.path.join(root, "zkp_amd"), "-o", exe])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0 and "all checks passed (host backend, N = 1)" in out.stdout, (out.stdout, out.stderr)
    printed = dict(l.split(" ", 1) for l in out.stdout.strip().splitlines() if " " in l and not l.startswith("all"))
    x, A, B, G, H = R._capi_points()
    prover = T.Prover(b"DLEQProof", T.Transcript(LABEL), host)
    var_x = prover.allocate_scalar(b"x", x)
    var_B, _ = prover.allocate_point(b"B", B)
    var_H, _ = prover.allocate_point(b"H", H)
    var_A, _ = prover.allocate_point(b"A", A)
    var_G, _ = prover.allocate_point(b"G", G)
    R.dleq_statement(prover, var_x, var_A, var_G, var_B, var_H)
    proof = prover.prove_compact(entropy=bytes(range(32)))
    assert printed["challenge"].strip() == proof.challenge.hex() and printed["response"].strip() == proof.responses[0].hex()


# ---- statement shapes the reference's tests never build (tests/statement_shapes.py), on the host backend ---------------------------------
from tests import statement_shapes as SH


@pytest.mark.parametrize("name", SH.SHAPES)
def test_unusual_statement_shapes_on_the_host_backend_vs_oracle(host, name):
    """A static point as left-hand side, one point as left-hand side of two constraints, a left-hand side that is a right-hand side
    elsewhere, a repeated term, a statement without instance points: every flow of the toolbox on the host backend against the oracle
    (proofs byte for byte, the batch verifier's operand scalars incl. static_coeffs).  The -m gpu twin runs the same checks on the device."""
    n = 6
    rng = np.random.default_rng(sum(name.encode()) + n)
    shape, secrets_int, dlog = SH._shape_case(name, n, rng)
    secrets, inst, common = SH._materialise(shape, n, secrets_int, dlog)
    SH._check_all_flows(host, shape, n, secrets, inst, common, seed=n)


def test_64_constraint_statement_on_the_host_backend_vs_oracle(host):
    n = 4
    shape, secrets, inst, common = SH.w64_constraints_case(n, 2, np.random.default_rng(66))
    SH._check_all_flows(host, shape, n, secrets, inst, common, seed=2)
