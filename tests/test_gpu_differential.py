"""Differential verdict test: more than a thousand MUTATED proofs, and for every one of them the verdict of the product
(HIP path, both routes: transcripts on the host threads / everything fused on the device) must equal the verdict of the
oracle's restatement of the reference verifier --

    Verifier::verify_compact      verifier.rs:80-120      -> oracle.cbind.verify_compact
    Verifier::verify_batchable    verifier.rs:123-173     -> oracle.cbind.verify_batchable
    BatchVerifier::verify_batchable  batch_verifier.rs:137-235 -> oracle.cbind.batch_verify   (a batch of valid proofs + the mutant)

-- instead of an expectation written down by hand.  Mutations: a flipped bit in every field (challenge, each response, each
commitment, each instance point); each of the 28 invalid encodings of RFC 9496 appendix A.2 in EVERY point slot and every
commitment slot; the identity encoding in every point / commitment slot (rejected by the verifier-side appends mod.rs:186-221
for batchable commitments, hashed like any other commitment in compact proofs -- there are none on the wire); s + l for every
response and for the challenge (serde would not deserialise them: proofs.rs:14-32); valid-but-wrong points.  Torsion-shifted
representatives of an internal point must encode to the same bytes (RFC 9496 4.3.2): checked through zkp_encode_many."""
import random

import numpy as np
import pytest

from oracle import cbind as C
from oracle import model as M
from zkp_amd import toolbox as T
from tests.test_host_field import BAD_ENCODINGS
from tests.test_gpu_toolbox import _cmz_batch
from tests.test_gpu_fused import _dleq_batch

pytestmark = pytest.mark.gpu
NEVER = 0xFFFFFFFF
BAD = [np.frombuffer(bytes.fromhex(h), np.uint8) for h in BAD_ENCODINGS]
BASE = bytes.fromhex("e2f2ae0a6abc4e71a884a961c500515f58e30b6aa582dd8db6a65945e08d2d76")


@pytest.fixture(scope="module")
def eng():
    from zkp_amd.engine import Engine
    e = Engine(0)
    yield e
    e.close()
    T.set_fused_min_batch(32)


def _plus_l(s: np.ndarray) -> np.ndarray:
    return np.frombuffer((int.from_bytes(s.tobytes(), "little") + M.L).to_bytes(32, "little"), np.uint8)


def _mutants(rng, n0, ni, m, nc, inst, chal, resp, coms):
    """-> list of (description, base proof, field, index, new 32 bytes)"""
    out = []
    pick = lambda: rng.randrange(n0)

    def flip(a):
        b = a.copy()
        b[rng.randrange(32)] ^= 1 << rng.randrange(8)
        return b
    for _ in range(6):
        j = pick(); out.append(("bit flip in the challenge", j, "chal", 0, flip(chal[j])))
    for i in range(m):
        j = pick(); out.append(("bit flip in response %d" % i, j, "resp", i, flip(resp[j, i])))
        j = pick(); out.append(("response %d + l" % i, j, "resp", i, _plus_l(resp[j, i])))
    j = pick(); out.append(("challenge + l", j, "chal", 0, _plus_l(chal[j])))
    for k in range(nc):
        j = pick(); out.append(("bit flip in commitment %d" % k, j, "coms", k, flip(coms[j, k])))
        j = pick(); out.append(("identity commitment %d" % k, j, "coms", k, np.zeros(32, np.uint8)))
        j = pick(); out.append(("another proof's commitment %d" % k, j, "coms", k, coms[(j + 1) % n0, k]))
        for b, enc in enumerate(BAD):
            j = pick(); out.append(("invalid encoding %d as commitment %d" % (b, k), j, "coms", k, enc))
    for p in range(ni):
        j = pick(); out.append(("bit flip in instance point %d" % p, j, "inst", p, flip(inst[p, j])))
        j = pick(); out.append(("identity as instance point %d" % p, j, "inst", p, np.zeros(32, np.uint8)))
        j = pick(); out.append(("another proof's instance point %d" % p, j, "inst", p, inst[p, (j + 1) % n0]))
        for b, enc in enumerate(BAD):
            j = pick(); out.append(("invalid encoding %d as instance point %d" % (b, p), j, "inst", p, enc))
    for j in range(n0):
        out.append(("untouched", j, None, 0, None))
    return out


def _assemble(muts, inst, chal, resp, coms):
    n = len(muts)
    base = np.array([mu[1] for mu in muts])
    a = dict(inst=np.ascontiguousarray(inst[:, base]), chal=chal[base].copy(), resp=resp[base].copy(), coms=coms[base].copy())
    for q, (_, _, field, idx, val) in enumerate(muts):
        if field == "chal":
            a["chal"][q] = val
        elif field == "resp":
            a["resp"][q, idx] = val
        elif field == "coms":
            a["coms"][q, idx] = val
        elif field == "inst":
            a["inst"][idx, q] = val
    assert len(a["chal"]) == n
    return a


def _differential(eng, mod, cst, label, secrets, inst, common, seed):
    st = mod.statement
    n0 = len(secrets)
    rng = random.Random(seed)
    nrng = np.random.default_rng(seed)
    entropy = nrng.integers(0, 256, size=(n0, 32), dtype=np.uint8)
    ts = np.stack([T.Transcript(label).state] * n0)
    chal, resp, coms = T.prove_batch(eng, st, ts, secrets, inst, common, entropy)
    muts = _mutants(rng, n0, st.ni, st.m, st.nc, inst, chal, resp, coms)
    a = _assemble(muts, inst, chal, resp, coms)
    n = len(muts)
    w_each = nrng.integers(0, 256, size=(n, st.nc, 16), dtype=np.uint8)
    # ---- the oracle's verdicts --------------------------------------------------------------------------------------
    want_c = np.zeros(n, np.uint8)
    want_b = np.zeros(n, np.uint8)
    for q in range(n):
        pts = np.concatenate([a["inst"][:, q], common])              # define_proof!'s allocation order: instance, then common
        want_c[q] = C.verify_compact(cst, label, pts, a["chal"][q], a["resp"][q])
        want_b[q] = C.verify_batchable(cst, label, pts, a["coms"][q], a["resp"][q], w_each[q])
    untouched = np.array([mu[2] is None for mu in muts])
    assert not want_c[untouched].any() and not want_b[untouched].any()
    # the mutation set exercises both outcomes of each verifier (e.g. commitments are not part of a compact proof)
    assert want_c.sum() > n // 3 and (want_c == 0).sum() > n // 4 and want_b.sum() > n // 2
    # ---- the product, both routes ---------------------------------------------------------------------------------------
    got = {}
    for route, thr in (("host", NEVER), ("fused", 0)):
        T.set_fused_min_batch(thr)
        try:
            ts = np.stack([T.Transcript(label).state] * n)
            got[route, "compact"] = T.verify_compact_batch(eng, st, ts, a["inst"], common, a["chal"], a["resp"])
            ts = np.stack([T.Transcript(label).state] * n)
            got[route, "batchable"] = T.verify_batchable_each(eng, st, ts, a["inst"], common, a["coms"], a["resp"], w_each)
        finally:
            T.set_fused_min_batch(32)
    for (route, kind), res in got.items():
        want = want_c if kind == "compact" else want_b
        diff = np.nonzero(res != want)[0]
        assert len(diff) == 0, "%s %s: %s" % (route, kind, [(muts[q][0], int(want[q]), int(res[q])) for q in diff[:8]])
    # ---- batch verification: every 3rd mutant inside a batch of valid proofs, all batches in one many-batch call ---------------
    sel = list(range(0, n, 3))
    K, n_each = len(sel), 4
    cols = []
    for q in sel:
        others = [(muts[q][1] + d) % n0 for d in (1, 2, 3)]
        cols.append([("base", others[0]), ("base", others[1]), ("mut", q), ("base", others[2])])
    b_inst = np.zeros((st.ni, K * n_each, 32), np.uint8)
    b_resp = np.zeros((K * n_each, st.m, 32), np.uint8)
    b_coms = np.zeros((K * n_each, st.nc, 32), np.uint8)
    for b, col in enumerate(cols):
        for i, (kind, idx) in enumerate(col):
            g = b * n_each + i
            if kind == "base":
                b_inst[:, g], b_resp[g], b_coms[g] = inst[:, idx], resp[idx], coms[idx]
            else:
                b_inst[:, g], b_resp[g], b_coms[g] = a["inst"][:, idx], a["resp"][idx], a["coms"][idx]
    w = nrng.integers(0, 256, size=(st.nc, K * n_each, 16), dtype=np.uint8)
    want_batch = np.zeros(K, np.int32)
    for b in range(K):
        sl = slice(b * n_each, (b + 1) * n_each)
        want_batch[b] = C.batch_verify(cst, label, n_each, np.ascontiguousarray(b_inst[:, sl]), common, b_coms[sl], b_resp[sl], np.ascontiguousarray(w[:, sl]))
    assert want_batch.sum() > K // 2 and (want_batch == 0).sum() >= 3
    # a mutant that verify_batchable accepts on its own must leave its batch valid and vice versa
    assert (want_batch == want_b[sel]).all()
    for route, thr in (("host", NEVER), ("fused", 0)):
        T.set_fused_min_batch(thr)
        try:
            ts = np.stack([T.Transcript(label).state] * (K * n_each))
            v = T.batch_verify_many(eng, st, K, ts, b_inst, common, b_coms, b_resp, w)
        finally:
            T.set_fused_min_batch(32)
        diff = np.nonzero(v != want_batch)[0]
        assert len(diff) == 0, "%s batch: %s" % (route, [(muts[sel[b]][0], int(want_batch[b]), int(v[b])) for b in diff[:8]])
    return n, K


def test_cmz_mutated_proofs_verdicts_equal_the_oracles(eng):
    mod, secrets, inst, common = _cmz_batch(6, 77)
    n, K = _differential(eng, mod, C.Statement.from_model(M.cmz_statement(10)), b"Benchmark", secrets, inst, common, 78)
    assert n >= 750 and K >= 250


def test_dleq_mutated_proofs_verdicts_equal_the_oracles(eng):
    n0 = 5
    mod, x, A, B, H = _dleq_batch(n0, 79)
    inst = np.ascontiguousarray(np.stack([A, B, H]))
    common = np.frombuffer(BASE, np.uint8).reshape(1, 32).copy()
    n, K = _differential(eng, mod, C.Statement.from_model(M.dleq_statement()), b"DLEQTest", x, inst, common, 80)
    assert n >= 150


def test_torsion_shifted_representatives_encode_identically(eng):
    """An internal point and its translates by the four 4-torsion points of the Edwards curve are ONE ristretto255 element:
    zkp_encode_many (mod.rs:180 `compress`) must give the same 32 bytes for all of them, with any Z."""
    rng = random.Random(81)
    i = M.SQRT_M1
    torsion = [(0, 1, 1, 0), (0, M.P - 1, 1, 0), (i, 0, 1, 0), (M.P - i, 0, 1, 0)]
    rows, want = [], []
    for _ in range(24):
        p = M.pt_mul(rng.randrange(1, M.L), M.BASEPOINT)
        e = M.ristretto_encode(p)
        for t4 in torsion:
            q = M.pt_add(p, t4)
            z = rng.randrange(1, M.P)
            q = tuple(c * z % M.P for c in q)
            rows.append(b"".join(c.to_bytes(32, "little") for c in q))
            want.append(e)
    out = eng.encode_many(np.frombuffer(b"".join(rows), np.uint8).reshape(-1, 128))
    assert [o.tobytes() for o in out] == want
    oracle_out = C.encode_many(np.frombuffer(b"".join(rows), np.uint8).reshape(-1, 128))
    assert (oracle_out == out).all()
