"""GPU tests of the whole drop-in path (Python binding -> C++ host toolbox -> C ABI -> HIP kernels), written
after the reference's own integration tests:
  tests/zkp.rs                          create_and_verify_compact / _batchable / create_batch_and_batch_verify
  tests/dleq_using_constraint_api.rs    the same through Prover / Verifier / BatchVerifier
  tests/sig_and_vrf_example.rs          accept / reject pattern, stateful transcript chaining
plus what the reference lacks: byte-exact proofs against golden fixtures and the oracle (injected
entropy), a bad proof inside a batch, malformed points, and the CMZ'13 workload at BASELINE sizes."""
import hashlib
import json
import os
import random

import numpy as np
import pytest

from oracle import cbind as C
from oracle import model as M
from zkp_amd import toolbox as T

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ristretto_msm.json")
BASEPOINT = bytes.fromhex("e2f2ae0a6abc4e71a884a961c500515f58e30b6aa582dd8db6a65945e08d2d76")


@pytest.fixture(scope="module")
def eng():
    from zkp_amd.engine import Engine
    e = Engine(0)
    yield e
    e.close()


def sc(x):
    return (x % M.L).to_bytes(32, "little")


def mul(x: int, enc: bytes) -> bytes:
    """test-input helper: x * P on the CPU (oracle)"""
    return C.msm_algo("straus_vartime", np.frombuffer(sc(x), np.uint8).reshape(1, 32), np.frombuffer(enc, np.uint8).reshape(1, 32))


def hash_to_point(msg: bytes) -> bytes:
    return C.from_uniform_bytes(hashlib.sha512(msg).digest())       # RistrettoPoint::hash_from_bytes::<Sha512>


# ---- tests/zkp.rs -----------------------------------------------------------------------------------------
dleq = T.define_proof("dleq", b"DLEQ Example Proof", ["x"], ["A", "B", "H"], ["G"], [("A", [("x", "G")]), ("B", [("x", "H")])])


def _dleq_assignments():
    H = hash_to_point(b"A VRF input, for instance")
    x = pow(89327492234, M.L - 2, M.L)                              # Scalar::from(89327492234u64).invert()
    return x, {"A": mul(x, BASEPOINT), "B": mul(x, H), "G": BASEPOINT, "H": H}


def test_create_and_verify_compact(eng):
    x, points = _dleq_assignments()
    transcript = T.Transcript(b"DLEQTest")
    proof = dleq.prove_compact(eng, transcript, {"x": x}, points)
    # Serialize and parse the bincode representation (tests/zkp.rs:53-54)
    proof_bytes = proof.to_bytes()
    assert len(proof_bytes) == 32 + 8 + 32
    parsed_proof = T.CompactProof.from_bytes(proof_bytes)
    transcript = T.Transcript(b"DLEQTest")
    dleq.verify_compact(eng, parsed_proof, transcript, points)      # is_ok()


def test_create_and_verify_batchable(eng):
    x, points = _dleq_assignments()
    proof = dleq.prove_batchable(eng, T.Transcript(b"DLEQTest"), {"x": x}, points)
    proof = T.BatchableProof.from_bytes(proof.to_bytes())           # tests/zkp.rs:96-97
    dleq.verify_batchable(eng, proof, T.Transcript(b"DLEQTest"), points)


def test_create_batch_and_batch_verify(eng):
    messages = [b"One message", b"Another message", b"A third message", b"A fourth message"]
    proofs, pubkeys, vrf_outputs, Hs = [], [], [], []
    for i, message in enumerate(messages):
        H = hash_to_point(message)
        x = 89327492234 * (i + 1)
        A, B = mul(x, BASEPOINT), mul(x, H)
        proofs.append(dleq.prove_batchable(eng, T.Transcript(b"DLEQTest"), {"x": x}, {"A": A, "B": B, "G": BASEPOINT, "H": H}))
        pubkeys.append(A); vrf_outputs.append(B); Hs.append(H)
    transcripts = [T.Transcript(b"DLEQTest") for _ in messages]
    dleq.batch_verify(eng, proofs, transcripts, {"A": pubkeys, "B": vrf_outputs, "H": Hs}, {"G": BASEPOINT})
    # (not in the reference) a single bad response poisons the batch
    bad = [T.BatchableProof(list(p.commitments), list(p.responses)) for p in proofs]
    bad[2].responses[0] = sc(int.from_bytes(bad[2].responses[0], "little") + 1)
    with pytest.raises(T.VerificationFailure):
        dleq.batch_verify(eng, bad, [T.Transcript(b"DLEQTest") for _ in messages], {"A": pubkeys, "B": vrf_outputs, "H": Hs}, {"G": BASEPOINT})
    with pytest.raises(T.BatchSizeMismatch):
        dleq.batch_verify(eng, proofs, transcripts[:3], {"A": pubkeys, "B": vrf_outputs, "H": Hs}, {"G": BASEPOINT})


# ---- tests/dleq_using_constraint_api.rs -------------------------------------------------------------------
def dleq_statement(cs, x, A, G, B, H):
    cs.constrain(A, [(x, B)])
    cs.constrain(G, [(x, H)])


def _capi_points():
    B = BASEPOINT
    H = hash_to_point(B)
    x = 89327492234
    return x, mul(x, B), B, mul(x, H), H         # x, A, B, G, H  with A = x B, G = x H


def test_create_and_verify_compact_dleq(eng):
    x, A, B, G, H = _capi_points()
    transcript = T.Transcript(b"DLEQTest")
    prover = T.Prover(b"DLEQProof", transcript, eng)
    var_x = prover.allocate_scalar(b"x", x)
    var_B, _ = prover.allocate_point(b"B", B)
    var_H, _ = prover.allocate_point(b"H", H)
    var_A, cmpr_A = prover.allocate_point(b"A", A)
    var_G, cmpr_G = prover.allocate_point(b"G", G)
    dleq_statement(prover, var_x, var_A, var_G, var_B, var_H)
    proof = prover.prove_compact()

    transcript = T.Transcript(b"DLEQTest")
    verifier = T.Verifier(b"DLEQProof", transcript, eng)
    var_x = verifier.allocate_scalar(b"x")
    var_B = verifier.allocate_point(b"B", B)
    var_H = verifier.allocate_point(b"H", H)
    var_A = verifier.allocate_point(b"A", cmpr_A)
    var_G = verifier.allocate_point(b"G", cmpr_G)
    dleq_statement(verifier, var_x, var_A, var_G, var_B, var_H)
    verifier.verify_compact(proof)


def test_create_and_verify_batchable_dleq(eng):
    x, A, B, G, H = _capi_points()
    prover = T.Prover(b"DLEQProof", T.Transcript(b"DLEQTest"), eng)
    var_x = prover.allocate_scalar(b"x", x)
    var_B, _ = prover.allocate_point(b"B", B)
    var_H, _ = prover.allocate_point(b"H", H)
    var_A, _ = prover.allocate_point(b"A", A)
    var_G, _ = prover.allocate_point(b"G", G)
    dleq_statement(prover, var_x, var_A, var_G, var_B, var_H)
    proof = prover.prove_batchable()
    verifier = T.Verifier(b"DLEQProof", T.Transcript(b"DLEQTest"), eng)
    var_x = verifier.allocate_scalar(b"x")
    var_B = verifier.allocate_point(b"B", B)
    var_H = verifier.allocate_point(b"H", H)
    var_A = verifier.allocate_point(b"A", A)
    var_G = verifier.allocate_point(b"G", G)
    dleq_statement(verifier, var_x, var_A, var_G, var_B, var_H)
    verifier.verify_batchable(proof)
    # identity public point is refused at allocation (mod.rs:191-193)
    v2 = T.Verifier(b"DLEQProof", T.Transcript(b"DLEQTest"), eng)
    with pytest.raises(T.VerificationFailure):
        v2.allocate_point(b"B", bytes(32))


def test_create_batch_and_batch_verify_dleq(eng):
    B = BASEPOINT
    H = hash_to_point(B)
    batch_size = 16
    proofs, cmpr_As, cmpr_Gs = [], [], []
    for j in range(batch_size):
        x = 89327492234 + j
        A, G = mul(x, B), mul(x, H)
        prover = T.Prover(b"DLEQProof", T.Transcript(b"DLEQBatchTest"), eng)
        var_x = prover.allocate_scalar(b"x", x)
        var_B, _ = prover.allocate_point(b"B", B)
        var_H, _ = prover.allocate_point(b"H", H)
        var_A, cmpr_A = prover.allocate_point(b"A", A)
        var_G, cmpr_G = prover.allocate_point(b"G", G)
        dleq_statement(prover, var_x, var_A, var_G, var_B, var_H)
        proofs.append(prover.prove_batchable())
        cmpr_As.append(cmpr_A); cmpr_Gs.append(cmpr_G)
    transcripts = [T.Transcript(b"DLEQBatchTest") for _ in range(batch_size)]
    verifier = T.BatchVerifier(b"DLEQProof", batch_size, transcripts, eng)
    var_x = verifier.allocate_scalar(b"x")
    var_B = verifier.allocate_static_point(b"B", B)
    var_H = verifier.allocate_static_point(b"H", H)
    var_A = verifier.allocate_instance_point(b"A", cmpr_As)
    var_G = verifier.allocate_instance_point(b"G", cmpr_Gs)
    dleq_statement(verifier, var_x, var_A, var_G, var_B, var_H)
    verifier.verify_batchable(proofs)
    with pytest.raises(T.BatchSizeMismatch):
        T.BatchVerifier(b"DLEQProof", batch_size, transcripts[:-1], eng)


# ---- tests/sig_and_vrf_example.rs ---------------------------------------------------------------------------
sig_proof = T.define_proof("sig_proof", b"Sig", ["x"], ["A"], ["B"], [("A", [("x", "B")])])


def sign(eng, sk: int, pk: bytes, message: bytes, transcript):
    transcript.append_message(b"msg", message)
    return sig_proof.prove_batchable(eng, transcript, {"x": sk}, {"A": pk, "B": BASEPOINT})


def verify(eng, sig, message: bytes, pk: bytes, transcript):
    transcript.append_message(b"msg", message)
    sig_proof.verify_batchable(eng, sig, transcript, {"A": pk, "B": BASEPOINT})


def test_create_and_verify_sig(eng):
    rng = random.Random(3)
    domain_sep, msg1, msg2 = b"My Sig Application", b"Test Message 1", b"Test Message 2"
    sk1, sk2 = rng.randrange(1, M.L), rng.randrange(1, M.L)
    pk1, pk2 = mul(sk1, BASEPOINT), mul(sk2, BASEPOINT)
    sig1 = sign(eng, sk1, pk1, msg1, T.Transcript(domain_sep))
    sig2 = sign(eng, sk2, pk2, msg2, T.Transcript(domain_sep))
    verify(eng, sig1, msg1, pk1, T.Transcript(domain_sep))
    verify(eng, sig2, msg2, pk2, T.Transcript(domain_sep))
    for sig, msg, pk, dom in [(sig1, msg1, pk2, domain_sep), (sig2, msg2, pk1, domain_sep),       # wrong pubkey
                              (sig1, msg2, pk1, domain_sep), (sig2, msg1, pk2, domain_sep),       # wrong message
                              (sig1, msg1, pk1, b"Wrong"), (sig2, msg2, pk2, b"Wrong")]:          # wrong domain separator
        with pytest.raises(T.VerificationFailure):
            verify(eng, sig, msg, pk, T.Transcript(dom))


def test_counterparty_signature_chain(eng):
    rng = random.Random(4)
    sk1, sk2 = rng.randrange(1, M.L), rng.randrange(1, M.L)
    pk1, pk2 = mul(sk1, BASEPOINT), mul(sk2, BASEPOINT)
    trans1, trans2 = T.Transcript(b"Counterparty Example"), T.Transcript(b"Counterparty Example")
    msgs = [b"In this test, two counterparties exchange signatures.", b"However, the counterparties sign and verify messages",
            b"using stateful transcript objects.", b"When party 1 signs, the party 1 transcript changes;",
            b"when party 2 verifies, the party 2 transcript syncs.", b"In this way, the transcript states ratchet stateful signatures."]
    for rnd in range(3):
        s = sign(eng, sk1, pk1, msgs[2 * rnd], trans1)
        verify(eng, s, msgs[2 * rnd], pk1, trans2)
        s = sign(eng, sk2, pk2, msgs[2 * rnd + 1], trans2)
        verify(eng, s, msgs[2 * rnd + 1], pk2, trans1)
    assert (trans1.state == trans2.state).all()


# ---- beyond the reference: byte-exact proofs ----------------------------------------------------------------
def test_golden_proofs_byte_exact(eng):
    """Injected entropy makes proofs deterministic: the GPU path must reproduce the committed fixtures."""
    fx = json.load(open(GOLDEN))
    for case in fx["proofs"]:
        mod = T.dleq_module() if case["statement"] == "dleq" else T.cmz_module(10)
        names = mod.instance + mod.common
        points = {n: bytes.fromhex(p) for n, p in zip(names, case["points"])}
        secrets = {n: bytes.fromhex(s) for n, s in zip(mod.secrets, case["secrets"])}
        label = bytes.fromhex(case["label"])
        proof = mod.prove_compact(eng, T.Transcript(label), secrets, points, entropy=bytes.fromhex(case["entropy"]))
        assert proof.challenge.hex() == case["challenge"]
        assert [r.hex() for r in proof.responses] == case["responses"]
        bp = mod.prove_batchable(eng, T.Transcript(label), secrets, points, entropy=bytes.fromhex(case["entropy"]))
        assert [c.hex() for c in bp.commitments] == case["commitments"]
        mod.verify_compact(eng, proof, T.Transcript(label), points)
        mod.verify_batchable(eng, bp, T.Transcript(label), points)
    for case in fx["msm"]:
        s = np.frombuffer(b"".join(bytes.fromhex(x) for x in case["scalars"]), np.uint8).reshape(-1, 32)
        p = np.frombuffer(b"".join(bytes.fromhex(x) for x in case["points"]), np.uint8).reshape(-1, 32)
        got = eng.msm_optional(s, p)
        assert (got.hex() if got is not None else None) == case["expect"]
    st = eng.decode_check(np.frombuffer(b"".join(bytes.fromhex(c["enc"]) for c in fx["decode"]), np.uint8).reshape(-1, 32))
    assert [int(x) == 0 for x in st] == [c["valid"] for c in fx["decode"]]


def _cmz_batch(n, seed):
    """n valid CMZ'13 presentations sharing the issuer parameters; inputs made with the oracle's arithmetic."""
    rng = np.random.default_rng(seed)
    mod = T.cmz_module(10)

    def rs(k):
        s = rng.integers(0, 256, size=(k, 32), dtype=np.uint8)
        s[:, 31] &= 0x0f
        return s

    base = np.frombuffer(BASEPOINT, np.uint8).reshape(1, 32)
    common_sc = rs(12)
    common, _ = C.msm_many(np.arange(13, dtype=np.uint32), common_sc, np.zeros(12, np.uint32), base, 0)   # X_1..X_10, A, B
    secrets = rs(n * 21).reshape(n, 21, 32)                        # m_1..m_10, z_1..z_10, minus_z_Q
    pq, _ = C.msm_many(np.arange(2 * n + 1, dtype=np.uint32), rs(2 * n), np.zeros(2 * n, np.uint32), base, 0)
    P, Q = pq[:n], pq[n:]
    # C_i = m_i P + z_i A ; V = sum m_i X_i + minus_z_Q Q      (benches/zkp.rs:34-45)
    table = np.concatenate([common, P, Q])
    off, scal, pidx = [0], [], []
    for j in range(n):
        for i in range(10):
            scal += [secrets[j, i], secrets[j, 10 + i]]
            pidx += [12 + j, 10]
            off.append(len(pidx))
        for i in range(10):
            scal.append(secrets[j, i]); pidx.append(i)
        scal.append(secrets[j, 20]); pidx.append(12 + n + j)
        off.append(len(pidx))
    cv, st = C.msm_many(np.array(off, np.uint32), np.stack(scal), np.array(pidx, np.uint32), table, 0)
    assert not st.any()
    cv = cv.reshape(n, 11, 32)
    inst = np.concatenate([cv[:, :10].transpose(1, 0, 2), P[None], Q[None], cv[:, 10][None]])      # C_1..C_10, P, Q, V
    return mod, secrets, np.ascontiguousarray(inst), common


@pytest.mark.parametrize("n", [64, 4096])
def test_cmz_batch_prove_verify(eng, n):
    """BASELINE configs[1] (n = 4096): prove -> verify_compact (every proof) -> batch_verify, and oracle parity."""
    mod, secrets, inst, common = _cmz_batch(n, 11)
    label = b"Benchmark"
    rng = np.random.default_rng(12)
    entropy = rng.integers(0, 256, size=(n, 32), dtype=np.uint8)
    ts = np.stack([T.Transcript(label).state] * n)
    chal, resp, coms = T.prove_batch(eng, mod.statement, ts, secrets, inst, common, entropy)
    # the C oracle's prover must produce the same bytes for EVERY proof of the batch (4096 proofs: ~3 s of CPU), and its batch
    # verifier must accept what the GPU made
    cst = C.Statement.from_model(M.cmz_statement(10))
    for j in range(n):
        pts = np.concatenate([inst[:, j], common])
        ec, er, ek, _ = C.prove(cst, label, secrets[j], pts, entropy[j].tobytes())
        assert chal[j].tobytes() == ec.tobytes() and (resp[j] == er).all() and (coms[j] == ek).all(), j
    w = rng.integers(0, 256, size=(11, n, 16), dtype=np.uint8)
    assert C.batch_verify(cst, label, n, inst, common, coms, resp, w) == 0
    ts = np.stack([T.Transcript(label).state] * n)
    res = T.verify_compact_batch(eng, mod.statement, ts, inst, common, chal, resp)
    assert not res.any()
    ts = np.stack([T.Transcript(label).state] * n)
    res = T.verify_batchable_each(eng, mod.statement, ts, inst, common, coms, resp)
    assert not res.any()
    ts = np.stack([T.Transcript(label).state] * n)
    T.batch_verify(eng, mod.statement, ts, inst, common, coms, resp)
    # a bad proof inside the batch: batch fails, per-proof verification localises it
    bad = resp.copy()
    k = n // 3
    bad[k, 7, 0] ^= 1
    ts = np.stack([T.Transcript(label).state] * n)
    with pytest.raises(T.VerificationFailure):
        T.batch_verify(eng, mod.statement, ts, inst, common, coms, bad)
    ts = np.stack([T.Transcript(label).state] * n)
    res = T.verify_batchable_each(eng, mod.statement, ts, inst, common, coms, bad)
    assert res[k] == 1 and res.sum() == 1
    ts = np.stack([T.Transcript(label).state] * n)
    res = T.verify_compact_batch(eng, mod.statement, ts, inst, common, chal, bad)
    assert res[k] == 1 and res.sum() == 1
    # a malformed instance point (does not decode): that proof fails, and the batch fails as a whole
    badinst = inst.copy()
    badinst[3, k] = np.frombuffer(bytes.fromhex("0100000000000000000000000000000000000000000000000000000000000000"), np.uint8)
    ts = np.stack([T.Transcript(label).state] * n)
    res = T.verify_compact_batch(eng, mod.statement, ts, badinst, common, chal, resp)
    assert res[k] == 1 and res.sum() == 1
    ts = np.stack([T.Transcript(label).state] * n)
    with pytest.raises(T.VerificationFailure):
        T.batch_verify(eng, mod.statement, ts, badinst, common, coms, resp)


def test_unreferenced_point_must_still_decode(eng):
    """CMZ's common point `B` is in no constraint, yet every verifier decompresses it (verifier.rs:87-92,
    :162-166; batch_verifier.rs:224-226).  The prover only hashes its encoding, so a proof made over an
    undecodable `B` has consistent transcripts and still must be rejected."""
    mod, secrets, inst, common = _cmz_batch(4, 21)
    label = b"Benchmark"
    badc = common.copy()
    badc[11] = np.frombuffer(bytes.fromhex("0100000000000000000000000000000000000000000000000000000000000000"), np.uint8)
    ts = np.stack([T.Transcript(label).state] * 4)
    chal, resp, coms = T.prove_batch(eng, mod.statement, ts, secrets, inst, badc)
    ts = np.stack([T.Transcript(label).state] * 4)
    assert T.verify_compact_batch(eng, mod.statement, ts, inst, badc, chal, resp).all()
    ts = np.stack([T.Transcript(label).state] * 4)
    assert T.verify_batchable_each(eng, mod.statement, ts, inst, badc, coms, resp).all()
    ts = np.stack([T.Transcript(label).state] * 4)
    with pytest.raises(T.VerificationFailure):
        T.batch_verify(eng, mod.statement, ts, inst, badc, coms, resp)


@pytest.mark.parametrize("n", [1, 3, 300, 4096])
def test_gpu_coefficient_build_matches_host_and_oracle(eng, n):
    """batch_verifier.rs:173-206 computed on the GPU (scalar arithmetic mod l in sc25519.h) gives,
    bit for bit, the coefficient vector of the host restatement (zkp_batch_verify_build, itself checked against the
    oracle in test_host_toolbox.py) -- for CMZ (static + instance points) and DLEQ (shared secret, two constraints)."""
    rng = np.random.default_rng(100 + n)
    mod, secrets, inst, common = _cmz_batch(n, 31)
    label = b"Benchmark"
    entropy = rng.integers(0, 256, size=(n, 32), dtype=np.uint8)
    ts = np.stack([T.Transcript(label).state] * n)
    chal, resp, coms = T.prove_batch(eng, mod.statement, ts, secrets, inst, common, entropy)
    w = rng.integers(0, 256, size=(mod.statement.nc, n, 16), dtype=np.uint8)
    w[0, 0] = 0                                                      # a zero weight: -0 must stay 0, not l
    w[1 % mod.statement.nc, n - 1] = 255                             # the largest u128
    ts = np.stack([T.Transcript(label).state] * n)
    want_sc, _ = T.batch_verify_build(mod.statement, ts, inst, common, coms, resp, w)
    ts = np.stack([T.Transcript(label).state] * n)
    ok, got = T.batch_verify_coeffs(eng, mod.statement, ts, inst, common, coms, resp, w)
    assert ok
    assert (got == want_sc).all()
    if n <= 300:                                                     # the oracle's own build of the same operands
        cst = C.Statement.from_model(M.cmz_statement(10))
        rc, osc, _ = C.batch_verify(cst, label, n, inst, common, coms, resp, w, want_msm_inputs=True)
        assert rc == 0 and (osc == got).all()
    # a non-canonical response (s + l: the same residue, another byte string) never reaches the reference's verifier -- serde
    # refuses it (proofs.rs:27-32 over dalek's Deserialize) -- so the batch fails here too, on both routes and in the oracle
    big = resp.copy()
    big[0, 0] = np.frombuffer(((int.from_bytes(resp[0, 0].tobytes(), "little") + M.L)).to_bytes(32, "little"), np.uint8)
    for thr in (0xFFFFFFFF, 0):
        T.set_fused_min_batch(thr)
        try:
            ts = np.stack([T.Transcript(label).state] * n)
            ok, _ = T.batch_verify_coeffs(eng, mod.statement, ts, inst, common, coms, big, w)
            assert not ok
            ts = np.stack([T.Transcript(label).state] * n)
            with pytest.raises(T.VerificationFailure):
                T.batch_verify(eng, mod.statement, ts, inst, common, coms, big, w)
        finally:
            T.set_fused_min_batch(32)
    if n <= 300:
        assert C.batch_verify(C.Statement.from_model(M.cmz_statement(10)), label, n, inst, common, coms, big, w) == 1
    ts = np.stack([T.Transcript(label).state] * n)
    with pytest.raises(T.VerificationFailure):
        T.batch_verify_build(mod.statement, ts, inst, common, coms, big, w)
    # tampered response: coefficients still match the host's, verdict is failure
    bad = resp.copy()
    bad[n // 2, 3, 1] ^= 4
    ts = np.stack([T.Transcript(label).state] * n)
    want_sc, _ = T.batch_verify_build(mod.statement, ts, inst, common, coms, bad, w)
    ts = np.stack([T.Transcript(label).state] * n)
    ok, got = T.batch_verify_coeffs(eng, mod.statement, ts, inst, common, coms, bad, w)
    assert not ok and (got == want_sc).all()


# ---- tests/sig_and_vrf_example.rs:283-384  create_and_verify_vrf -----------------------------------------------
vrf_proof = T.define_proof("vrf_proof", b"VRF", ["x"], ["A", "G", "H"], ["B"], [("A", [("x", "B")]), ("G", [("x", "H")])])


def hash_to_group(transcript) -> bytes:
    """TranscriptProtocol::hash_to_group of the example (sig_and_vrf_example.rs:25-29): 64 challenge bytes -> from_uniform_bytes"""
    return C.from_uniform_bytes(transcript.challenge_bytes(b"output", 64))


def vrf(eng, sk: int, pk: bytes, function_transcript, message: bytes, proof_transcript):
    function_transcript.append_message(b"msg", message)
    H = hash_to_group(function_transcript)
    G = mul(sk, H)
    proof = vrf_proof.prove_compact(eng, proof_transcript, {"x": sk}, {"A": pk, "B": BASEPOINT, "G": G, "H": H})
    return G, proof


def vrf_verify(eng, output: bytes, function_transcript, message: bytes, pk: bytes, proof_transcript, proof):
    function_transcript.append_message(b"msg", message)
    H = hash_to_group(function_transcript)
    vrf_proof.verify_compact(eng, proof, proof_transcript, {"A": pk, "B": BASEPOINT, "G": output, "H": H})


def test_create_and_verify_vrf(eng):
    rng = random.Random(5)
    domain_sep, msg1, msg2 = b"My VRF Application", b"Test Message 1", b"Test Message 2"
    sk1, sk2 = rng.randrange(1, M.L), rng.randrange(1, M.L)
    pk1, pk2 = mul(sk1, BASEPOINT), mul(sk2, BASEPOINT)
    output1, proof1 = vrf(eng, sk1, pk1, T.Transcript(domain_sep), msg1, T.Transcript(domain_sep))
    output2, proof2 = vrf(eng, sk2, pk2, T.Transcript(domain_sep), msg2, T.Transcript(domain_sep))
    # each VRF output was correctly produced
    vrf_verify(eng, output1, T.Transcript(domain_sep), msg1, pk1, T.Transcript(domain_sep), proof1)
    vrf_verify(eng, output2, T.Transcript(domain_sep), msg2, pk2, T.Transcript(domain_sep), proof2)
    other = b"A different application"
    for out, msg, pk, dom, proof in [(output1, msg1, pk2, domain_sep, proof1), (output2, msg2, pk1, domain_sep, proof2),      # wrong pubkey
                                     (output2, msg1, pk1, domain_sep, proof1), (output1, msg2, pk2, domain_sep, proof2),      # wrong output
                                     (output1, msg1, pk1, other, proof1), (output2, msg2, pk2, other, proof2)]:               # wrong domain separator
        with pytest.raises(T.VerificationFailure):
            vrf_verify(eng, out, T.Transcript(domain_sep), msg, pk, T.Transcript(dom), proof)


# ---- beyond the reference: malleability and allocation order -----------------------------------------------------
def test_non_canonical_challenge_is_rejected(eng):
    """verifier.rs:115 compares the recomputed (canonical) challenge with the proof's Scalar; c + l is a different
    value there and does not even deserialise.  Both routes of verify_compact must refuse it."""
    x, points = _dleq_assignments()
    proof = dleq.prove_compact(eng, T.Transcript(b"DLEQTest"), {"x": x}, points)
    dleq.verify_compact(eng, proof, T.Transcript(b"DLEQTest"), points)
    shifted = (int.from_bytes(proof.challenge, "little") + M.L).to_bytes(32, "little")
    with pytest.raises(T.VerificationFailure):                                  # host-transcript route (N = 1)
        dleq.verify_compact(eng, T.CompactProof(shifted, proof.responses), T.Transcript(b"DLEQTest"), points)
    n = 64                                                                      # fused route (N >= 32)
    _, inst, common = dleq.pack([], [points] * n)
    chal = np.stack([np.frombuffer(proof.challenge, np.uint8)] * n).copy()
    resp = np.stack([np.frombuffer(b"".join(proof.responses), np.uint8).reshape(-1, 32)] * n)
    chal[17] = np.frombuffer(shifted, np.uint8)
    ts = np.stack([T.Transcript(b"DLEQTest").state] * n)
    res = T.verify_compact_batch(eng, dleq.statement, ts, inst, common, chal, resp)
    assert res[17] == 1 and res.sum() == 1
    old = T.lib().zkp_toolbox_get_fused_min_batch()
    try:                                                                        # and the host route at the same size
        T.lib().zkp_toolbox_set_fused_min_batch(0xffffffff)
        ts = np.stack([T.Transcript(b"DLEQTest").state] * n)
        res = T.verify_compact_batch(eng, dleq.statement, ts, inst, common, chal, resp)
        assert res[17] == 1 and res.sum() == 1
    finally:
        T.lib().zkp_toolbox_set_fused_min_batch(old)


@pytest.mark.parametrize("n", [1, 48])
def test_interleaved_allocation_order_matches_model(eng, n):
    """The reference appends to the transcript at every allocate_* call, in the caller's order (prover.rs:52-73):
    point, scalar, point, scalar, point, point must give the model's bytes on the host route (n = 1) and the fused one."""
    rng = random.Random(21)
    Bp, Hp = M.BASEPOINT, M.ristretto_hash_from_bytes_sha512(b"interleaved")
    label = b"InterleaveTest"

    def build(cs, x, y, encs_or_pts):
        vB = cs.allocate_point(b"B", encs_or_pts["B"])
        vx = cs.allocate_scalar(b"x", x) if x is not None else cs.allocate_scalar(b"x")
        vH = cs.allocate_point(b"H", encs_or_pts["H"])
        vy = cs.allocate_scalar(b"y", y) if y is not None else cs.allocate_scalar(b"y")
        vA = cs.allocate_point(b"A", encs_or_pts["A"])
        vG = cs.allocate_point(b"G", encs_or_pts["G"])
        first = lambda v: v[0] if isinstance(v, tuple) else v
        cs.constrain(first(vA), [(vx, first(vB)), (vy, first(vH))])
        cs.constrain(first(vG), [(vy, first(vB))])

    proofs, all_encs, ents = [], [], []
    for j in range(n):
        x, y = rng.randrange(1, M.L), rng.randrange(1, M.L)
        pts = {"B": Bp, "H": Hp, "A": M.pt_add(M.pt_mul(x, Bp), M.pt_mul(y, Hp)), "G": M.pt_mul(y, Bp)}
        encs = {k: M.ristretto_encode(v) for k, v in pts.items()}
        ent = bytes(rng.randrange(256) for _ in range(32))
        mp = M.Prover(b"Interleaved", M.Transcript(label))
        build(mp, x, y, pts)
        want = mp.prove_compact(ent)
        proofs.append((x, y, want)); all_encs.append(encs); ents.append(ent)
    if n == 1:
        x, y, want = proofs[0]
        pr = T.Prover(b"Interleaved", T.Transcript(label), eng)
        build(pr, x, y, all_encs[0])
        got = pr.prove_compact(ents[0])
        assert got.challenge == M.sc_to_bytes(want.challenge) and got.responses == [M.sc_to_bytes(r) for r in want.responses]
        ve = T.Verifier(b"Interleaved", T.Transcript(label), eng)
        build(ve, None, None, all_encs[0])
        ve.verify_compact(got)
        return
    # a batch through the statement API: the same allocation sequence, every point an instance point
    st = T.Statement(b"Interleaved")
    vB = st.add_point(b"B", False); vx = st.add_secret(b"x"); vH = st.add_point(b"H", False); vy = st.add_secret(b"y")
    vA = st.add_point(b"A", False); vG = st.add_point(b"G", False)
    st.constrain(vA, [(vx, vB), (vy, vH)])
    st.constrain(vG, [(vy, vB)])
    secrets = np.frombuffer(b"".join(sc(x) + sc(y) for x, y, _ in proofs), np.uint8).reshape(n, 2, 32)
    inst = np.frombuffer(b"".join(e[k] for k in ("B", "H", "A", "G") for e in all_encs), np.uint8).reshape(4, n, 32)
    entropy = np.frombuffer(b"".join(ents), np.uint8).reshape(n, 32)
    ts = np.stack([T.Transcript(label).state] * n)
    chal, resp, coms = T.prove_batch(eng, st, ts, secrets, inst, np.zeros((0, 32), np.uint8), entropy)      # fused (n >= 32)
    for j, (_, _, want) in enumerate(proofs):
        assert chal[j].tobytes() == M.sc_to_bytes(want.challenge), j
        assert [r.tobytes() for r in resp[j]] == [M.sc_to_bytes(r) for r in want.responses], j
    ts = np.stack([T.Transcript(label).state] * n)
    assert not T.verify_compact_batch(eng, st, ts, inst, np.zeros((0, 32), np.uint8), chal, resp).any()
    ts = np.stack([T.Transcript(label).state] * n)
    T.batch_verify(eng, st, ts, inst, np.zeros((0, 32), np.uint8), coms, resp)
    old = T.lib().zkp_toolbox_get_fused_min_batch()
    try:                                                                        # host-transcript route: same bytes
        T.lib().zkp_toolbox_set_fused_min_batch(0xffffffff)
        ts = np.stack([T.Transcript(label).state] * n)
        chal2, resp2, coms2 = T.prove_batch(eng, st, ts, secrets, inst, np.zeros((0, 32), np.uint8), entropy)
        assert (chal2 == chal).all() and (resp2 == resp).all() and (coms2 == coms).all()
    finally:
        T.lib().zkp_toolbox_set_fused_min_batch(old)


def test_batch_verify_locate_names_the_bad_proofs(eng):
    """batch_verifier.rs:233 can only say that some proof is wrong; zkp_batch_verify_locate runs the batch
    check and, when it fails, the per-proof check: two tampered proofs out of 300 are named, a clean batch costs one call."""
    n = 300
    mod, secrets, inst, common = _cmz_batch(n, 21)
    label = b"locate"
    entropy = np.random.default_rng(22).integers(0, 256, size=(n, 32), dtype=np.uint8)
    ts = np.stack([T.Transcript(label).state] * n)
    chal, resp, coms = T.prove_batch(eng, mod.statement, ts, secrets, inst, common, entropy)
    ts = np.stack([T.Transcript(label).state] * n)
    ok, res = T.batch_verify_locate(eng, mod.statement, ts, inst, common, coms, resp)
    assert ok and not res.any()
    after_ok = ts.copy()
    bad_resp, bad_coms = resp.copy(), coms.copy()
    bad_resp[41, 3, 7] ^= 0x10
    bad_coms[207, 10] = coms[206, 10]                              # a valid point, the wrong commitment
    ts = np.stack([T.Transcript(label).state] * n)
    ok, res = T.batch_verify_locate(eng, mod.statement, ts, inst, common, bad_coms, bad_resp)
    assert not ok and sorted(np.nonzero(res)[0].tolist()) == [41, 207]
    ts2 = np.stack([T.Transcript(label).state] * n)
    with pytest.raises(T.VerificationFailure):
        T.batch_verify(eng, mod.statement, ts2, inst, common, bad_coms, bad_resp)
    assert (ts[:, :203] == ts2[:, :203]).all()                     # transcripts are left as the batch check leaves them
    with pytest.raises(T.BatchSizeMismatch):
        T.batch_verify_locate(eng, mod.statement, ts[:-1], inst, common, coms, resp)
    assert after_ok.shape == ts.shape


@pytest.mark.parametrize("grouped", [0, 1])
def test_statement_with_more_common_points_than_table_slots(eng, grouped):
    """70 common generators: 64 get fixed-base tables, six stay cold and are shared by every proof of the batch -- comb tables
    (masked scans), or with ZKP_OPT_GROUPED_COMB their terms walk through LDS as one group of N terms per point.  The fused
    route (statement-aware one-launch classifier, k_stmt_classify) must produce the host-transcript route's bytes (generic
    classifier), for proving and for both verifications."""
    ng, n = 70, 48
    xs = ["x_%d" % i for i in range(ng)]
    gs = ["G_%d" % i for i in range(ng)]
    wide = T.define_proof("wide70", b"W70", xs, ["Q"], gs, [("Q", [(x, g) for x, g in zip(xs, gs)])])
    rng = np.random.default_rng(70)

    def rs(k):
        s = rng.integers(0, 256, size=(k, 32), dtype=np.uint8)
        s[:, 31] &= 0x0f
        return s

    base = np.frombuffer(BASEPOINT, np.uint8).reshape(1, 32)
    common, _ = C.msm_many(np.arange(ng + 1, dtype=np.uint32), rs(ng), np.zeros(ng, np.uint32), base, 0)
    secrets = rs(n * ng).reshape(n, ng, 32)
    off = (ng * np.arange(n + 1)).astype(np.uint32)
    q, st = C.msm_many(off, secrets.reshape(-1, 32), np.tile(np.arange(ng, dtype=np.uint32), n), common, 0)
    assert not st.any()
    inst = np.ascontiguousarray(q[None])                              # [1 instance point][n][32]
    entropy = rng.integers(0, 256, size=(n, 32), dtype=np.uint8)
    label = b"W70 test"
    fresh = lambda: np.stack([T.Transcript(label).state] * n)
    eng.set_option(6, grouped)
    try:
        results = []
        for min_batch in (0, 10**9):                                  # fused route, then host transcripts + generic classifier
            T.set_fused_min_batch(min_batch)
            ts = fresh()
            chal, resp, coms = T.prove_batch(eng, wide.statement, ts, secrets, inst, common, entropy)
            res = T.verify_compact_batch(eng, wide.statement, fresh(), inst, common, chal, resp)
            assert not res.any()
            T.batch_verify(eng, wide.statement, fresh(), inst, common, coms, resp)
            bad = resp.copy()
            bad[5, 69, 0] ^= 1
            res = T.verify_compact_batch(eng, wide.statement, fresh(), inst, common, chal, bad)
            assert res[5] == 1 and res.sum() == 1
            results.append((chal, resp, coms, ts))
        for a, b in zip(results[0], results[1]):
            assert (a == b).all()
    finally:
        T.set_fused_min_batch(32)
        eng.set_option(6, 2**64 - 1)


@pytest.mark.parametrize("n", [1, 1000])
def test_c_example_on_the_gpu(tmp_path, n):
    """examples/dleq_c_abi.c `gpu N`: a C99 program linked against the two shared libraries alone drives GPU 0 -- prove, both wire formats, the three
    verifiers, a tampered proof refused and located (no Python, no torch in that process)."""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "dleq_c_abi")
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Wextra", "-Werror", "-pedantic", "-I", os.path.join(root, "include"), os.path.join(root, "examples", "dleq_c_abi.c"),
                           "-L", os.path.join(root, "zkp_amd"), "-lzkp_toolbox", "-lzkp_mi355x", "-Wl,-rpath," + os.path.join(root, "zkp_amd"), "-o", exe])
    out = subprocess.run([exe, "gpu", str(n)], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "all checks passed (GPU 0, N = %d)" % n in out.stdout, (out.stdout, out.stderr)
    host = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert host.returncode == 0 and host.stdout.splitlines()[:3] == out.stdout.splitlines()[:3]       # proof 0: the bytes of the host backend


@pytest.mark.gpu
@pytest.mark.parametrize("pinned", [False, True])
def test_seeded_synchronous_calls_draw_the_pinned_chacha_stream(eng, pinned):
    """zkp_fused_prove_seeded / zkp_fused_batch_verify_many_seeded (round 6: what zkp_prove_batch / zkp_batch_verify call when the caller gives no entropy /
    no weights): proof j is the proof zkp_fused_prove makes from bytes [32 j, 32 j + 32) of the ChaCha20 stream of the seed, byte for byte -- on ordinary and
    on pinned buffers (pinned inputs are queued right behind the fork, the commitments leave early) -- and the seeded batch verification accepts the batch and
    rejects it after a flipped response bit."""
    import ctypes
    from zkp_amd.engine import load_library
    from tests.test_gpu_device_entry import _cmz_fused_statement
    hip = load_library()
    n = 700
    mod, secrets, inst, common = _cmz_batch(n, 41)
    fst = _cmz_fused_statement()
    seed = bytes(range(7, 47))
    T.lib().zkp_chacha20_block.argtypes = [ctypes.c_char_p, ctypes.c_uint64, ctypes.c_uint64, ctypes.c_char_p]
    out, stream = ctypes.create_string_buffer(64), b""
    for b in range((32 * n + 63) // 64):
        T.lib().zkp_chacha20_block(seed[:32], b, int.from_bytes(seed[32:], "little"), out)
        stream += out.raw
    entropy = np.frombuffer(stream[: 32 * n], np.uint8).reshape(n, 32)
    t0 = np.stack([T.Transcript(b"seeded").state] * n)
    T.set_fused_min_batch(0)
    try:
        chal, resp, coms = T.prove_batch(eng, mod.statement, t0.copy(), secrets, inst, common, entropy)       # zkp_fused_prove with the stream as entropy
    finally:
        T.set_fused_min_batch(32)
    mk = T.pinned_copy if pinned else (lambda a: np.array(a, copy=True))      # (the calls advance the transcripts in place)
    ts, sec, ins, com = mk(t0), mk(secrets), mk(inst), mk(common)
    alloc = T.pinned_empty if pinned else (lambda shape: np.zeros(shape, np.uint8))
    c2, r2, k2 = alloc((n, 32)), alloc((n, 21, 32)), alloc((n, 11, 32))
    invalid = ctypes.c_int(1)
    p = lambda a: a.ctypes.data_as(ctypes.c_void_p)
    eng.prepare_fixed_points(common)
    rc = hip.zkp_fused_prove_seeded(eng._h, ctypes.byref(fst.c), ctypes.c_uint32(n), p(ts), p(sec), p(ins), p(com), seed, p(c2), p(r2), p(k2), ctypes.byref(invalid))
    assert rc == 0 and invalid.value == 0
    assert (c2 == chal).all() and (r2 == resp).all() and (k2 == coms).all()
    for flip in (False, True):
        rr = r2.copy() if not pinned else T.pinned_copy(np.asarray(r2))
        if flip:
            rr[n // 2, 3, 0] ^= 1
        ts2 = mk(t0)
        verdicts = (ctypes.c_int * 1)(7)
        rc = hip.zkp_fused_batch_verify_many_seeded(eng._h, ctypes.byref(fst.c), ctypes.c_uint32(1), ctypes.c_uint32(n), p(ts2), p(ins), p(com), p(k2), p(rr), seed, verdicts)
        assert rc == 0 and verdicts[0] == (1 if flip else 0)
