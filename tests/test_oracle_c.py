"""Pins the C oracle (oracle/c, liboracle.so) before anything trusts it: RFC 9496 vectors, Merlin's
published known-answer test, the reference tests' deterministic public inputs, and agreement with the
big-integer model (oracle/model.py) for every primitive, every dalek MSM algorithm and the whole
toolbox flow (proof bytes, accept/reject decisions, batch-verification MSM inputs)."""
import hashlib
import json
import os
import random

import numpy as np
import pytest

from oracle import cbind as C
from oracle import model as M
from tests.test_host_field import BAD_ENCODINGS, GENERATOR_MULTIPLES

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="module", autouse=True)
def _built():
    C.build()


def sc(x):
    return (x % (1 << 256)).to_bytes(32, "little")


def arr(rows, width=32):
    return np.frombuffer(b"".join(rows), np.uint8).reshape(-1, width) if rows else np.zeros((0, width), np.uint8)


def test_rfc9496_vectors():
    st = C.decode_check(arr([bytes.fromhex(h) for h in BAD_ENCODINGS]))
    assert st.all()
    st = C.decode_check(arr([bytes.fromhex(h) for h in GENERATOR_MULTIPLES]))
    assert not st.any()
    # k * B through each MSM algorithm
    B = bytes.fromhex(GENERATOR_MULTIPLES[1])
    for k, h in enumerate(GENERATOR_MULTIPLES):
        for algo in ("straus_ct", "straus_vartime"):
            assert C.msm_algo(algo, arr([sc(k)]), arr([B])).hex() == h


def test_codec_matches_model():
    rng = random.Random(21)
    cases = [bytes(rng.randrange(256) for _ in range(32)) for _ in range(400)] + [bytes(32)]
    st, xyzt = C.decode_check(arr(cases), want_coords=True)
    for i, c in enumerate(cases):
        p = M.ristretto_decode(c)
        assert st[i] == (0 if p is not None else 1)
        if p is not None:
            got = [int.from_bytes(xyzt[i, 32 * k:32 * k + 32].tobytes(), "little") for k in range(4)]
            assert got == [p[0], p[1], 1, p[3]]
    pts = [M.pt_mul(rng.randrange(M.L), M.BASEPOINT) for _ in range(30)] + [M.IDENTITY]
    rows = []
    for p in pts:
        z = rng.randrange(1, M.P)
        rows.append(b"".join((c * z % M.P).to_bytes(32, "little") for c in p))
    out = C.encode_many(arr(rows, 128))
    for i, p in enumerate(pts):
        assert out[i].tobytes() == M.ristretto_encode(p)
    for _ in range(50):
        u = bytes(rng.randrange(256) for _ in range(64))
        assert C.from_uniform_bytes(u) == M.ristretto_encode(M.ristretto_from_uniform_bytes(u))


def test_reference_test_inputs():
    """Deterministic public inputs of the reference's own tests."""
    h1 = C.from_uniform_bytes(hashlib.sha512(b"A VRF input, for instance").digest())       # tests/zkp.rs:34
    assert h1.hex() == "8062d869a1a967d6a60604a3ec8d0316cef712e094f4cb991de60a9f52555068"
    B = bytes.fromhex(GENERATOR_MULTIPLES[1])
    h2 = C.from_uniform_bytes(hashlib.sha512(B).digest())          # tests/dleq_using_constraint_api.rs:43
    assert h2.hex() == "90ca11cd6c6227cb0abc39e2710c444ae6617ea81898e716353f3410d9656605"
    x = 89327492234
    assert C.msm_algo("straus_vartime", arr([sc(x)]), arr([B])).hex() == "241dacbe397b94c04f21eaa1f1df11878ae3b9833351d8cc957c9c7c13a36800"
    assert C.msm_algo("straus_ct", arr([sc(x)]), arr([h2])).hex() == "9ea41491e551a36b01db7bcf332ec44633d747e0a9ef73b5da77be43bbe5687e"
    xinv = pow(x, M.L - 2, M.L)                                                              # tests/zkp.rs:35
    assert C.msm_algo("straus_ct", arr([sc(xinv)]), arr([B])).hex() == "7af0b00ee8b188c437ed7a2ec4b679bb062c81fdc561f8e33296a4802e738643"
    assert C.msm_algo("straus_vartime", arr([sc(xinv)]), arr([h1])).hex() == "76f4e263598bc40ee74ebf628aeb3b9ff8c9d33c860df353eafdf6eeb977c742"


def test_scalars():
    rng = random.Random(22)
    for _ in range(300):
        w = bytes(rng.randrange(256) for _ in range(64))
        assert C.sc_from_wide(w) == sc(int.from_bytes(w, "little") % M.L)
        a, b, c = (rng.randrange(1 << 256) for _ in range(3))
        assert C.sc_muladd(sc(a), sc(b), sc(c)) == sc((a * b + c) % M.L)
        assert C.sc_neg(sc(a)) == sc((-a) % M.L)
    assert C.sc_from_wide(b"\xff" * 64) == sc(((1 << 512) - 1) % M.L)
    assert C.sc_neg(sc(0)) == sc(0) and C.sc_neg(sc(M.L)) == sc(0)


def test_merlin_kat():
    got = C.merlin_challenge(b"test protocol", [(b"some label", b"some data")], b"challenge", 32)
    assert got.hex() == "d5a21972d0d5fe320c0d263fac7fffb8145aa640af6e9bca177c03c7efcf0615"
    # longer traffic incl. messages crossing the 166-byte STROBE rate, against the model
    rng = random.Random(23)
    appends = [(b"l%d" % i, bytes(rng.randrange(256) for _ in range(n))) for i, n in enumerate([0, 1, 165, 166, 167, 400, 32])]
    t = M.Transcript(b"proto")
    for l, m in appends:
        t.append_message(l, m)
    assert C.merlin_challenge(b"proto", appends, b"c", 200) == t.challenge_bytes(b"c", 200)


@pytest.mark.parametrize("algo,n", [("straus_ct", 0), ("straus_ct", 1), ("straus_ct", 2), ("straus_ct", 11),
                                    ("straus_vartime", 1), ("straus_vartime", 3), ("straus_vartime", 36), ("straus_vartime", 189),
                                    ("pippenger", 190), ("pippenger", 499), ("pippenger", 500), ("pippenger", 799),
                                    ("pippenger", 800), ("pippenger", 1500)])
def test_msm_algorithms(algo, n):
    """Each dalek algorithm (and each Pippenger window size w = 6, 7, 8) against the definition."""
    rng = random.Random(100 + n)
    logs = [rng.randrange(1, M.L) for _ in range(16)]
    encs = [M.ristretto_encode(M.pt_mul(k, M.BASEPOINT)) for k in logs]
    idx = [rng.randrange(16) for _ in range(n)]
    scal = [rng.randrange(M.L) for _ in range(n)]
    for k, v in enumerate([0, 1, M.L - 1, (1 << 252) - 1, 1 << 252]):
        if k < n:
            scal[k] = v
    got = C.msm_algo(algo, arr([sc(s) for s in scal]), arr([encs[j] for j in idx]))
    dlog = sum(s * logs[j] for s, j in zip(scal, idx)) % M.L
    assert got == M.ristretto_encode(M.pt_mul(dlog, M.BASEPOINT))


def test_msm_contracts_small():
    rng = random.Random(31)
    pts = [M.ristretto_from_uniform_bytes(bytes(rng.randrange(256) for _ in range(64))) for _ in range(4)]
    encs = [M.ristretto_encode(p) for p in pts] + [bytes.fromhex(BAD_ENCODINGS[4])]
    off, scal, pidx = [0, 0, 2, 5, 6], [], []
    for _ in range(5):
        scal.append(rng.randrange(1 << 256))
        pidx.append(rng.randrange(4))
    scal.append(7)
    pidx.append(4)
    for flags in (0, 1):
        out, status = C.msm_many(off, arr([sc(s) for s in scal]), pidx, arr(encs), flags)
        assert list(status) == [0, 0, 0, 1]
        assert out[0].tobytes() == bytes(32) and out[3].tobytes() == bytes(32)
        for m in (1, 2):
            ts = range(off[m], off[m + 1])
            assert out[m].tobytes() == M.ristretto_encode(M.msm_points([scal[t] for t in ts], [pts[pidx[t]] for t in ts]))
    assert C.msm_optional(arr([sc(1)] * 2), arr([encs[0], encs[4]])) is None
    assert C.msm_optional(arr([sc(s) for s in scal[:3]]), arr([encs[p] for p in pidx[:3]])) == \
        M.msm_optional([sc(s) for s in scal[:3]], [encs[p] for p in pidx[:3]])


# ---- toolbox flow: C oracle == model, byte for byte ---------------------------------------------
def _dleq_instance(rng):
    x = rng.randrange(1, M.L)
    G = M.BASEPOINT
    H = M.ristretto_from_uniform_bytes(bytes(rng.randrange(256) for _ in range(64)))
    return {"x": x}, {"A": M.pt_mul(x, G), "B": M.pt_mul(x, H), "H": H, "G": G}


def _cmz_instance(rng, n=10):
    sec = {f"m_{i}": rng.randrange(M.L) for i in range(1, n + 1)}
    sec.update({f"z_{i}": rng.randrange(M.L) for i in range(1, n + 1)})
    sec["minus_z_Q"] = rng.randrange(M.L)
    rp = lambda: M.pt_mul(rng.randrange(1, M.L), M.BASEPOINT)
    pts = {f"X_{i}": rp() for i in range(1, n + 1)}
    pts.update({"A": rp(), "B": rp(), "P": rp(), "Q": rp()})
    for i in range(1, n + 1):
        pts[f"C_{i}"] = M.pt_add(M.pt_mul(sec[f"m_{i}"], pts["P"]), M.pt_mul(sec[f"z_{i}"], pts["A"]))
    pts["V"] = M.msm_points([sec[f"m_{i}"] for i in range(1, n + 1)] + [sec["minus_z_Q"]],
                            [pts[f"X_{i}"] for i in range(1, n + 1)] + [pts["Q"]])
    return sec, pts


@pytest.mark.parametrize("which", ["dleq", "cmz"])
def test_prove_matches_model_and_verifies(which):
    rng = random.Random(41)
    mst = M.dleq_statement() if which == "dleq" else M.cmz_statement(10)
    sec, pts = _dleq_instance(rng) if which == "dleq" else _cmz_instance(rng)
    cst = C.Statement.from_model(mst)
    entropy = bytes(rng.randrange(256) for _ in range(32))
    label = b"Benchmark"
    pr, encs = mst.build_prover(M.Transcript(label), sec, pts)
    mc, mresp, mcoms, mblind = pr._prove_impl(entropy)
    enc_rows = [encs[n] for n in cst.points]
    chal, resp, coms, blind = C.prove(cst, label, arr([sc(sec[n]) for n in cst.secrets]), arr(enc_rows), entropy)
    assert chal.tobytes() == sc(mc)
    assert [r.tobytes() for r in resp] == [sc(r) for r in mresp]
    assert [c.tobytes() for c in coms] == mcoms
    assert [b.tobytes() for b in blind] == [sc(b) for b in mblind]
    # accept
    assert C.verify_compact(cst, label, arr(enc_rows), chal, resp) == 0
    w = arr([rng.randrange(1 << 128).to_bytes(16, "little") for _ in cst.constraints], 16)
    assert C.verify_batchable(cst, label, arr(enc_rows), coms, resp, w) == 0
    mst.build_verifier(M.Transcript(label), encs).verify_compact(M.CompactProof(mc, mresp))
    # reject: wrong transcript label, tampered response, tampered commitment, swapped public point (tests/sig_and_vrf_example.rs pattern)
    assert C.verify_compact(cst, b"Benchmarl", arr(enc_rows), chal, resp) == 1
    bad = resp.copy()
    bad[0, 0] ^= 1
    assert C.verify_compact(cst, label, arr(enc_rows), chal, bad) == 1
    assert C.verify_batchable(cst, label, arr(enc_rows), coms, bad, w) == 1
    badc = coms.copy()
    badc[0] = np.frombuffer(enc_rows[0], np.uint8)
    assert C.verify_batchable(cst, label, arr(enc_rows), badc, resp, w) == 1
    swapped = list(enc_rows)
    swapped[0], swapped[1] = swapped[1], swapped[0]
    assert C.verify_compact(cst, label, arr(swapped), chal, resp) == 1
    # identity public point / identity commitment are rejected before any arithmetic (mod.rs:191,215)
    ident = list(enc_rows)
    ident[1] = bytes(32)
    assert C.verify_compact(cst, label, arr(ident), chal, resp) == 1
    zc = coms.copy()
    zc[0] = 0
    assert C.verify_batchable(cst, label, arr(enc_rows), zc, resp, w) == 1


def test_batch_verify_matches_model():
    rng = random.Random(43)
    mst = M.dleq_statement()
    cst = C.Statement.from_model(mst)
    n, label = 5, b"DLEQBatchTest"
    proofs, encs_all = [], []
    for _ in range(n):
        sec, pts = _dleq_instance(rng)
        pts["H"] = pts["H"]
        pr, encs = mst.build_prover(M.Transcript(label), sec, pts)
        proofs.append(pr.prove_batchable(bytes(rng.randrange(256) for _ in range(32))))
        encs_all.append(encs)
    weights = [[rng.randrange(1 << 128) for _ in range(n)] for _ in mst.constraints]
    inst = {k: [e[k] for e in encs_all] for k in mst.instance}
    common = {k: encs_all[0][k] for k in mst.common}
    bv = mst.build_batch_verifier([M.Transcript(label) for _ in range(n)], inst, common)
    m_scalars, m_points = bv.coefficient_build(proofs, weights)
    inst_rows = arr([e for k in mst.instance for e in inst[k]])
    common_rows = arr([common[k] for k in mst.common])
    coms = arr([c for p in proofs for c in p.commitments])
    resp = arr([sc(r) for p in proofs for r in p.responses])
    w16 = arr([w.to_bytes(16, "little") for row in weights for w in row], 16)
    rc, ms, mp = C.batch_verify(cst, label, n, inst_rows, common_rows, coms, resp, w16, want_msm_inputs=True)
    assert rc == 0
    assert [r.tobytes() for r in ms] == [sc(s) for s in m_scalars]
    assert [r.tobytes() for r in mp] == m_points
    assert C.batch_verify(cst, label, n, inst_rows, common_rows, coms, resp, w16) == 0
    # one bad proof poisons the batch (the reference has no such test; batch_verifier.rs:230-234)
    bad = resp.copy()
    bad[3, 5] ^= 0x10
    assert C.batch_verify(cst, label, n, inst_rows, common_rows, coms, bad, w16) == 1


def test_golden_fixtures():
    """tests/golden/*.json were produced by tests/golden/make_fixtures.py from the big-integer model with
    libsodium cross-checks in this container; the C oracle must reproduce them."""
    path = os.path.join(GOLDEN, "ristretto_msm.json")
    fx = json.load(open(path))
    for case in fx["msm"]:
        got = C.msm_optional(arr([bytes.fromhex(s) for s in case["scalars"]]), arr([bytes.fromhex(p) for p in case["points"]]))
        assert (got.hex() if got is not None else None) == case["expect"]
    for case in fx["proofs"]:
        mst = M.dleq_statement() if case["statement"] == "dleq" else M.cmz_statement(10)
        cst = C.Statement.from_model(mst)
        chal, resp, coms, _ = C.prove(cst, bytes.fromhex(case["label"]), arr([bytes.fromhex(s) for s in case["secrets"]]),
                                      arr([bytes.fromhex(p) for p in case["points"]]), bytes.fromhex(case["entropy"]))
        assert chal.tobytes().hex() == case["challenge"]
        assert [r.tobytes().hex() for r in resp] == case["responses"]
        assert [c.tobytes().hex() for c in coms] == case["commitments"]


# RFC 9496 appendix A.3: SHA-512 of each label -> from_uniform_bytes (the one-way map) -> encoding.  These are the vectors
# curve25519-dalek carries for RistrettoPoint::from_uniform_bytes / hash_from_bytes, which the reference's tests use to
# make H (tests/zkp.rs:35, tests/dleq_using_constraint_api.rs:44, tests/sig_and_vrf_example.rs:28).
RFC9496_A3 = [
    (b"Ristretto is traditionally a short shot of espresso coffee", "3066f82a1a747d45120d1740f14358531a8f04bbffe6a819f86dfe50f44a0a46"),
    (b"made with the normal amount of ground coffee but extracted with", "f26e5b6f7d362d2d2a94c5d0e7602cb4773c95a2e5c31a64f133189fa76ed61b"),
    (b"about half the amount of water in the same amount of time", "006ccd2a9e6867e6a2c5cea83d3302cc9de128dd2a9a57dd8ee7b9d7ffe02826"),
    (b"by using a finer grind.", "f8f0c87cf237953c5890aec3998169005dae3eca1fbb04548c635953c817f92a"),
    (b"This produces a concentrated shot of coffee per volume.", "ae81e7dedf20a497e10c304a765c1767a42d6e06029758d2d7e8ef7cc4c41179"),
    (b"Just pulling a normal shot short will produce a weaker shot", "e2705652ff9f5e44d3e841bf1c251cf7dddb77d140870d1ab2ed64f1a9ce8628"),
    (b"and is not a Ristretto as some believe.", "80bd07262511cdde4863f8a7434cef696750681cb9510eea557088f76d9e5065"),
]


def test_rfc9496_a3_hash_to_group_vectors():
    for label, want in RFC9496_A3:
        wide = hashlib.sha512(label).digest()
        assert C.from_uniform_bytes(wide).hex() == want
        assert M.ristretto_encode(M.ristretto_from_uniform_bytes(wide)).hex() == want
        assert M.ristretto_encode(M.ristretto_hash_from_bytes_sha512(label)).hex() == want


def test_from_bytes_mod_order_wide_edge_vectors():
    """Scalar::from_bytes_mod_order_wide (mod.rs:226 turns 64 challenge bytes into the challenge): 0, l - 1, l, the
    largest 512-bit value, 2^256 and friends -- against plain integer arithmetic."""
    L = M.L
    cases = [0, 1, L - 1, L, L + 1, 2 * L - 1, (1 << 252), (1 << 255) - 19, (1 << 256) - 1, 1 << 256, (1 << 256) + L, L << 256,
             (L << 256) - 1, (1 << 512) - 1, ((1 << 512) - 1) // L * L, ((1 << 512) - 1) // L * L - 1]
    for v in cases:
        b = v.to_bytes(64, "little")
        assert C.sc_from_wide(b) == (v % L).to_bytes(32, "little"), hex(v)
        assert M.sc_from_bytes_mod_order_wide(b) == v % L
    # the constant dalek's own tests carry for 2^256 - 1 mod l (scalar.rs CANONICAL_2_256_MINUS_1)
    want = bytes([28, 149, 152, 141, 116, 49, 236, 214, 112, 207, 125, 115, 244, 91, 239, 198] + [254] + [255] * 14 + [15])
    assert C.sc_from_wide(((1 << 256) - 1).to_bytes(64, "little")) == want


def test_merlin_block_crossings_c_oracle_equals_model():
    """STROBE's rate is 166 bytes: labels, messages and challenge outputs that end exactly at, one before, one after and
    several blocks beyond a block boundary must frame identically in the C oracle and the model."""
    rng = random.Random(77)
    for trial in range(12):
        appends = []
        for i in range(6):
            ll = rng.choice([0, 1, 5, 160, 165, 166, 167, 340])
            ml = rng.choice([0, 1, 31, 32, 164, 165, 166, 167, 331, 332, 333, 700])
            appends.append((bytes(rng.randrange(1, 256) for _ in range(ll)), bytes(rng.randrange(256) for _ in range(ml))))
        n = rng.choice([1, 32, 64, 165, 166, 167, 400])
        t = M.Transcript(b"crossing %d" % trial)
        for lab, msg in appends:
            t.append_message(lab, msg)
        assert C.merlin_challenge(b"crossing %d" % trial, appends, b"out", n) == t.challenge_bytes(b"out", n), trial


def test_oracle_is_clean_under_sanitizers():
    """the CPU restatement built with -fsanitize=address,undefined (oracle/Makefile: liboracle_asan.so)
    and driven over its whole surface -- codec on valid, invalid and random encodings, every MSM algorithm incl. empty and
    single-term inputs, scalar edge values, Merlin across block boundaries, DLEQ and CMZ prove / verify_compact /
    verify_batchable / batch_verify incl. rejected proofs -- must finish without a sanitizer report."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    asan = subprocess.check_output(["gcc", "-print-file-name=libasan.so"], text=True).strip()
    ubsan = subprocess.check_output(["gcc", "-print-file-name=libubsan.so"], text=True).strip()
    if not (os.path.isabs(asan) and os.path.exists(asan)):
        pytest.skip("no AddressSanitizer runtime next to gcc")
    code = r'''
import random, hashlib
import numpy as np
from oracle import cbind as C
from oracle import model as M
rng = random.Random(99)
arr = lambda rows, w=32: np.frombuffer(b"".join(rows), np.uint8).reshape(-1, w) if rows else np.zeros((0, w), np.uint8)
sc = lambda x: (x % (1 << 256)).to_bytes(32, "little")
encs = [bytes(rng.randrange(256) for _ in range(32)) for _ in range(64)] + [bytes(32), b"\xff" * 32]
st, xyzt = C.decode_check(arr(encs), want_coords=True)
B = M.ristretto_encode(M.BASEPOINT)
for algo in ("straus_ct", "straus_vartime", "pippenger"):
    for n in (1, 2, 17, 300):
        C.msm_algo(algo, arr([sc(rng.randrange(1 << 256)) for _ in range(n)]), arr([B] * n))
assert C.msm_optional(np.zeros((0, 32), np.uint8), np.zeros((0, 32), np.uint8)) == bytes(32)
assert C.msm_optional(arr([sc(1)]), arr([b"\x01" + bytes(31)])) is None
for v in (0, M.L - 1, M.L, (1 << 512) - 1):
    C.sc_from_wide(v.to_bytes(64, "little"))
C.sc_muladd(sc((1 << 256) - 1), sc((1 << 256) - 1), sc((1 << 256) - 1)); C.sc_neg(sc(0)); C.from_uniform_bytes(hashlib.sha512(b"x").digest())
C.merlin_challenge(b"t", [(b"a" * 170, b"m" * 700), (b"", b"")], b"c", 400)
from tests.test_oracle_c import _dleq_instance, _cmz_instance
for which in ("dleq", "cmz"):
    mst = M.dleq_statement() if which == "dleq" else M.cmz_statement(10)
    cst = C.Statement.from_model(mst)
    n = 3
    secs, pts = zip(*[(_dleq_instance(rng) if which == "dleq" else _cmz_instance(rng)) for _ in range(n)])
    common = {k: pts[0][k] for k in mst.common}
    coms, resps, insts = [], [], []
    for j in range(n):
        p = dict(pts[j]); p.update(common)
        if which == "cmz":
            for i in range(1, 11):
                p[f"C_{i}"] = M.pt_add(M.pt_mul(secs[j][f"m_{i}"], p["P"]), M.pt_mul(secs[j][f"z_{i}"], p["A"]))
            p["V"] = M.msm_points([secs[j][f"m_{i}"] for i in range(1, 11)] + [secs[j]["minus_z_Q"]], [p[f"X_{i}"] for i in range(1, 11)] + [p["Q"]])
        else:
            p["A"] = M.pt_mul(secs[j]["x"], p["G"]); p["B"] = M.pt_mul(secs[j]["x"], p["H"])
        enc = {k: M.ristretto_encode(v) for k, v in p.items()}
        ec, er, ek, _ = C.prove(cst, b"asan", arr([sc(secs[j][k]) for k in cst.secrets]), arr([enc[k] for k in cst.points]), bytes([j]) * 32)
        assert C.verify_compact(cst, b"asan", arr([enc[k] for k in cst.points]), ec, er) == 0
        w = np.frombuffer(bytes(rng.randrange(256) for _ in range(16 * len(cst.constraints))), np.uint8).reshape(-1, 16)
        assert C.verify_batchable(cst, b"asan", arr([enc[k] for k in cst.points]), ek, er, w) == 0
        bad = er.copy(); bad[0, 0] ^= 1
        assert C.verify_compact(cst, b"asan", arr([enc[k] for k in cst.points]), ec, bad) != 0
        coms.append(ek); resps.append(er); insts.append([enc[k] for k in mst.instance])
    inst = np.stack([arr([insts[j][i] for j in range(n)]) for i in range(len(mst.instance))])
    com_arr = arr([M.ristretto_encode(common[k]) for k in mst.common])
    w = np.frombuffer(bytes(rng.randrange(256) for _ in range(16 * len(cst.constraints) * n)), np.uint8).reshape(-1, n, 16)
    assert C.batch_verify(cst, b"asan", n, inst, com_arr, np.stack(coms), np.stack(resps), w) == 0
    badr = np.stack(resps).copy(); badr[1, 0, 3] ^= 2
    assert C.batch_verify(cst, b"asan", n, inst, com_arr, np.stack(coms), badr, w) != 0
print("sanitizer run complete")
'''
    env = dict(os.environ, ORACLE_SANITIZE="1", LD_PRELOAD=asan + ":" + ubsan, ASAN_OPTIONS="detect_leaks=0:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")
    r = subprocess.run([sys.executable, "-c", code], cwd=root, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "sanitizer run complete" in r.stdout, (r.stdout[-500:], r.stderr[-3000:])
    assert "AddressSanitizer" not in r.stderr and "runtime error" not in r.stderr, r.stderr[-3000:]


@pytest.mark.parametrize("isa", ["avx512ifma", "avx2", "avx2p"])
def test_simd_msm_equals_scalar_port(isa):
    """oracle/c/simd_ifma.c + simd_x4.inc (the four coordinates of a point in four lanes -- the design of curve25519-dalek's simd_backend -- on
    AVX-512 IFMA in radix 2^51, on AVX2 in radix 2^25.5 with one limb per vector, and on AVX2 in dalek's packed FieldElement2625x4 layout): the three MSM algorithms, whole proofs and a batch verification must give the bytes
    of the scalar 5 x 51 port.  Skipped where the build or the CPU lacks the instruction set (the vector code is then not even compiled /
    refused at run time)."""
    if isa not in C.simd_isas():
        assert C.set_simd(isa) is False and C.simd_mode() is None
        pytest.skip("no %s on this CPU / in this build" % isa)
    rng = np.random.default_rng(2024)
    base = np.frombuffer(bytes.fromhex("e2f2ae0a6abc4e71a884a961c500515f58e30b6aa582dd8db6a65945e08d2d76"), np.uint8).reshape(1, 32)
    try:
        for n, algos in ((1, ("straus_ct", "straus_vartime")), (2, ("straus_ct", "straus_vartime")), (11, ("straus_ct", "straus_vartime", "pippenger")),
                         (37, ("straus_ct", "straus_vartime", "pippenger")), (600, ("pippenger",)), (900, ("pippenger",))):
            ks = rng.integers(0, 256, size=(n, 32), dtype=np.uint8)
            ks[:, 31] &= 0x0f
            C.set_simd(False)
            pts, st = C.msm_many(np.arange(n + 1, dtype=np.uint32), ks, np.zeros(n, np.uint32), base, 0)
            assert not st.any()
            sc = rng.integers(0, 256, size=(n, 32), dtype=np.uint8)
            sc[0] = 0                                                   # a zero scalar
            if n > 1:
                sc[1] = np.frombuffer((M.L - 1).to_bytes(32, "little"), np.uint8)
            if n > 2:
                sc[2] = 0xff                                            # 2^256 - 1: reduced mod l by both backends
            for algo in algos:
                C.set_simd(False)
                want = C.msm_algo(algo, sc, pts)
                assert C.set_simd(isa) and C.simd_mode() == isa
                got = C.msm_algo(algo, sc, pts)
                assert got == want, (n, algo)
        # whole flows: prover (constant-time Straus), verifiers (NAF Straus), batch verifier (Pippenger)
        from tests.test_gpu_toolbox import _cmz_batch                 # (CPU only: the instance is made with the oracle's own arithmetic)
        cst = C.Statement.from_model(M.cmz_statement(10))
        C.set_simd(False)
        n = 40
        mod, secrets, inst, common = _cmz_batch(n, 77)
        ent = rng.integers(0, 256, size=(n, 32), dtype=np.uint8)
        w = rng.integers(0, 256, size=(11, n, 16), dtype=np.uint8)
        proofs = {}
        for simd in (False, isa):
            assert C.set_simd(simd) == bool(simd)
            out = [C.prove(cst, b"simd", secrets[j], np.concatenate([inst[:, j], common]), ent[j].tobytes())[:3] for j in range(n)]
            proofs[simd] = out
            coms = np.stack([o[2] for o in out])
            resp = np.stack([o[1] for o in out])
            assert C.batch_verify(cst, b"simd", n, inst, common, coms, resp, w) == 0                    # 12 + 24 * 40 = 972 terms: Pippenger w = 8
            bad = resp.copy()
            bad[7, 3, 0] ^= 1
            assert C.batch_verify(cst, b"simd", n, inst, common, coms, bad, w) == 1
            assert C.verify_compact(cst, b"simd", np.concatenate([inst[:, 3], common]), out[3][0], out[3][1]) == 0
        for a, b in zip(proofs[False], proofs[isa]):
            assert a[0].tobytes() == b[0].tobytes() and (a[1] == b[1]).all() and (a[2] == b[2]).all()
    finally:
        C.set_simd(False)
