"""Fused device flows (include/zkp_mi355x.h section 2c: Merlin transcripts, scalars mod l and MSMs all on the GPU)
against the host-transcript route of the same toolbox calls: proofs, verdicts and the transcripts left behind must be
identical byte for byte, for every batch size (the route is chosen by zkp_toolbox_set_fused_min_batch), and both
agree with the C oracle on sampled proofs.  Reference behaviour covered: prover.rs:76-112, verifier.rs:80-120,
batch_verifier.rs:67-235, mod.rs:165-228 (identity rejection included)."""
import random

import numpy as np
import pytest

from oracle import cbind as C
from oracle import model as M
from zkp_amd import toolbox as T
from tests.test_gpu_toolbox import BASEPOINT, _cmz_batch

pytestmark = pytest.mark.gpu
NEVER = 0xFFFFFFFF


@pytest.fixture(scope="module")
def eng():
    from zkp_amd.engine import Engine
    e = Engine(0)
    yield e
    e.close()
    T.set_fused_min_batch(32)


def _fresh(label, n, extra=None):
    t = T.Transcript(label)
    if extra is not None:
        t.append_message(b"ctx", extra)
    return np.stack([t.state] * n)


def _dleq_batch(n, seed):
    rng = np.random.default_rng(seed)
    mod = T.dleq_module()
    x = rng.integers(0, 256, size=(n, 1, 32), dtype=np.uint8)
    x[:, :, 31] &= 0x0f
    base = np.frombuffer(BASEPOINT, np.uint8).reshape(1, 32)
    hs = rng.integers(0, 256, size=(n, 32), dtype=np.uint8)
    hs[:, 31] &= 0x0f
    H, _ = C.msm_many(np.arange(n + 1, dtype=np.uint32), hs, np.zeros(n, np.uint32), base, 0)
    A, _ = C.msm_many(np.arange(n + 1, dtype=np.uint32), x[:, 0], np.zeros(n, np.uint32), base, 0)
    B, _ = C.msm_many(np.arange(n + 1, dtype=np.uint32), x[:, 0], np.arange(n, dtype=np.uint32), H, 0)
    return mod, x, A, B, H


@pytest.mark.parametrize("n", [1, 5, 64, 333, 4096])
def test_cmz_fused_equals_host_route(eng, n):
    mod, secrets, inst, common = _cmz_batch(n, 41)
    st = mod.statement
    label = b"Benchmark"
    rng = np.random.default_rng(n)
    entropy = rng.integers(0, 256, size=(n, 32), dtype=np.uint8)
    w = rng.integers(0, 256, size=(st.nc, n, 16), dtype=np.uint8)
    out = {}
    for route, thr in (("host", NEVER), ("fused", 0)):
        T.set_fused_min_batch(thr)
        assert T.get_fused_min_batch() == thr
        ts = _fresh(label, n, b"some context")
        chal, resp, coms = T.prove_batch(eng, st, ts, secrets, inst, common, entropy)
        ts_p = ts.copy()
        ts = _fresh(label, n, b"some context")
        res = T.verify_compact_batch(eng, st, ts, inst, common, chal, resp)
        ts_v = ts.copy()
        ts = _fresh(label, n, b"some context")
        ok, coeffs = T.batch_verify_coeffs(eng, st, ts, inst, common, coms, resp, w)
        ts_b = ts.copy()
        w_each = np.ascontiguousarray(w.transpose(1, 0, 2))                  # verifier.rs:153 draws per proof: [N][nc][16]
        ts = _fresh(label, n, b"some context")
        e0 = T.verify_batchable_each(eng, st, ts, inst, common, coms, resp, w_each)
        ts_e = ts.copy()
        # a wrong response, a wrong challenge, an identity instance point, an undecodable instance point
        k = n // 2
        bad_resp = resp.copy(); bad_resp[k, 3, 0] ^= 1
        bad_chal = chal.copy(); bad_chal[k, 5] ^= 8
        ts = _fresh(label, n, b"some context")
        r1 = T.verify_compact_batch(eng, st, ts, inst, common, chal, bad_resp)
        ts = _fresh(label, n, b"some context")
        r2 = T.verify_compact_batch(eng, st, ts, inst, common, bad_chal, resp)
        ident = inst.copy(); ident[2, k] = 0
        ts = _fresh(label, n, b"some context")
        r3 = T.verify_compact_batch(eng, st, ts, ident, common, chal, resp)
        junk = inst.copy(); junk[4, k] = np.frombuffer(bytes([1] + [0] * 31), np.uint8)
        ts = _fresh(label, n, b"some context")
        r4 = T.verify_compact_batch(eng, st, ts, junk, common, chal, resp)
        ts = _fresh(label, n, b"some context")
        ok_bad, _ = T.batch_verify_coeffs(eng, st, ts, inst, common, coms, bad_resp, w)
        ts = _fresh(label, n, b"some context")
        ok_ident, _ = T.batch_verify_coeffs(eng, st, ts, ident, common, coms, resp, w)
        zc = coms.copy(); zc[k, 0] = 0
        ts = _fresh(label, n, b"some context")
        ok_zc, _ = T.batch_verify_coeffs(eng, st, ts, inst, common, zc, resp, w)
        ts = _fresh(label, n, b"some context")
        e1 = T.verify_batchable_each(eng, st, ts, inst, common, coms, bad_resp, w_each)
        ts = _fresh(label, n, b"some context")
        e2 = T.verify_batchable_each(eng, st, ts, ident, common, coms, resp, w_each)
        ts = _fresh(label, n, b"some context")
        e3 = T.verify_batchable_each(eng, st, ts, junk, common, coms, resp, w_each)
        ts = _fresh(label, n, b"some context")
        e4 = T.verify_batchable_each(eng, st, ts, inst, common, zc, resp, w_each)
        badc = common.copy(); badc[11] = np.frombuffer(bytes([1] + [0] * 31), np.uint8)    # `B`: in no constraint, must still decode
        ts = _fresh(label, n, b"some context")
        e5 = T.verify_batchable_each(eng, st, ts, inst, badc, coms, resp, w_each)
        out[route] = dict(chal=chal, resp=resp, coms=coms, ts_p=ts_p, res=res, ts_v=ts_v, ok=ok, coeffs=coeffs, ts_b=ts_b,
                          r1=r1, r2=r2, r3=r3, r4=r4, ok_bad=ok_bad, ok_ident=ok_ident, ok_zc=ok_zc,
                          e0=e0, e1=e1, e2=e2, e3=e3, e4=e4, e5=e5, ts_e=ts_e)
    T.set_fused_min_batch(32)
    h, f = out["host"], out["fused"]
    for key in ("chal", "resp", "coms", "res", "coeffs", "r1", "r2", "r3", "r4", "e0", "e1", "e2", "e3", "e4", "e5"):
        assert (h[key] == f[key]).all(), key
    assert not f["e0"].any() and f["e5"].all()
    for key in ("e1", "e2", "e3", "e4"):
        assert f[key][n // 2] == 1 and f[key].sum() == 1, key
    for key in ("ts_p", "ts_v", "ts_b", "ts_e"):                               # 200 state bytes + pos, pos_begin, cur_flags
        assert (h[key][:, :203] == f[key][:, :203]).all(), key
    assert h["ok"] and f["ok"] and not f["res"].any()
    for key in ("r1", "r2", "r3", "r4"):
        assert f[key][n // 2] == 1 and f[key].sum() == 1, key
    assert not f["ok_bad"] and not f["ok_ident"] and not f["ok_zc"]
    assert not h["ok_bad"] and not h["ok_ident"] and not h["ok_zc"]


@pytest.mark.parametrize("n", [2, 700])
def test_dleq_fused_equals_host_route_and_oracle(eng, n):
    mod, x, A, B, H = _dleq_batch(n, 7)
    st = mod.statement
    label = b"DLEQBatch"
    inst = np.ascontiguousarray(np.stack([A, B, H]))
    common = np.frombuffer(BASEPOINT, np.uint8).reshape(1, 32).copy()
    rng = np.random.default_rng(5)
    entropy = rng.integers(0, 256, size=(n, 32), dtype=np.uint8)
    got = {}
    for route, thr in (("host", NEVER), ("fused", 0)):
        T.set_fused_min_batch(thr)
        ts = _fresh(label, n)
        chal, resp, coms = T.prove_batch(eng, st, ts, x, inst, common, entropy)
        ts2 = _fresh(label, n)
        res = T.verify_compact_batch(eng, st, ts2, inst, common, chal, resp)
        ts3 = _fresh(label, n)
        T.batch_verify(eng, st, ts3, inst, common, coms, resp)
        got[route] = (chal, resp, coms, ts, res, ts2, ts3)
    T.set_fused_min_batch(32)
    for a, b in zip(got["host"], got["fused"]):
        a, b = (a[:, :203], b[:, :203]) if a.shape[-1] == 208 else (a, b)
        assert (a == b).all()
    assert not got["fused"][4].any()
    cst = C.Statement.from_model(M.dleq_statement())
    for j in (0, n - 1):
        pts = np.concatenate([inst[:, j], common])
        ec, er, ek, _ = C.prove(cst, label, x[j], pts, entropy[j].tobytes())
        assert got["fused"][0][j].tobytes() == ec.tobytes() and (got["fused"][1][j] == er).all() and (got["fused"][2][j] == ek).all()


def test_ragged_transcripts_use_the_host_route(eng):
    """Transcripts at different STROBE positions cannot share a device program: the toolbox hashes them on the host
    (same results as proving each one alone); the raw fused entry point refuses them."""
    n = 300
    mod, x, A, B, H = _dleq_batch(n, 9)
    st = mod.statement
    inst = np.ascontiguousarray(np.stack([A, B, H]))
    common = np.frombuffer(BASEPOINT, np.uint8).reshape(1, 32).copy()
    entropy = np.random.default_rng(1).integers(0, 256, size=(n, 32), dtype=np.uint8)
    states = []
    for j in range(n):
        t = T.Transcript(b"ragged")
        t.append_message(b"m", b"x" * (j % 7))
        states.append(t.state)
    T.set_fused_min_batch(0)
    ts = np.stack(states)
    chal, resp, coms = T.prove_batch(eng, st, ts, x, inst, common, entropy)
    T.set_fused_min_batch(NEVER)
    ts2 = np.stack(states)
    chal2, resp2, coms2 = T.prove_batch(eng, st, ts2, x, inst, common, entropy)
    T.set_fused_min_batch(32)
    assert (chal == chal2).all() and (resp == resp2).all() and (coms == coms2).all() and (ts[:, :203] == ts2[:, :203]).all()
    ts3 = np.stack(states)
    res = T.verify_compact_batch(eng, st, ts3, inst, common, chal, resp)
    assert not res.any()


def _random_points(k, rng):
    base = np.frombuffer(BASEPOINT, np.uint8).reshape(1, 32)
    s = rng.integers(0, 256, size=(k, 32), dtype=np.uint8)
    s[:, 31] &= 0x0f
    pts, st = C.msm_many(np.arange(k + 1, dtype=np.uint32), s, np.zeros(k, np.uint32), base, 0)
    assert not st.any()
    return pts, s


@pytest.mark.parametrize("n", [3, 400])
def test_constraint_api_allocation_order_fused(eng, n):
    """benches/dleq.rs:188-241: the static points G, H are allocated BEFORE the instance points A, B, so their
    encodings enter the transcript first; the fused route takes the order from alloc_order.  Proofs must equal the
    oracle's (same allocation order) byte for byte."""
    rng = np.random.default_rng(77)
    label = b"DLEQBatchTest"
    cst = C.Statement(b"DLEQProof", ["x"], [("G", True), ("H", True), ("A", False), ("B", False)],
                      [("A", [("x", "G")]), ("B", [("x", "H")])])
    st = T.Statement(b"DLEQProof")
    x = st.add_secret(b"x")
    g, h = st.add_point(b"G", True), st.add_point(b"H", True)
    a, b = st.add_point(b"A", False), st.add_point(b"B", False)
    st.constrain(a, [(x, g)])
    st.constrain(b, [(x, h)])
    gh, _ = _random_points(2, rng)
    xs = rng.integers(0, 256, size=(n, 1, 32), dtype=np.uint8)
    xs[:, :, 31] &= 0x0f
    A, _ = C.msm_many(np.arange(n + 1, dtype=np.uint32), xs[:, 0], np.zeros(n, np.uint32), gh[0:1], 0)
    B, _ = C.msm_many(np.arange(n + 1, dtype=np.uint32), xs[:, 0], np.zeros(n, np.uint32), gh[1:2], 0)
    inst = np.ascontiguousarray(np.stack([A, B]))
    entropy = rng.integers(0, 256, size=(n, 32), dtype=np.uint8)
    T.set_fused_min_batch(0)
    try:
        ts = _fresh(label, n)
        chal, resp, coms = T.prove_batch(eng, st, ts, xs, inst, gh, entropy)
        for j in sorted(set([0, n // 2, n - 1])):
            ec, er, ek, _ = C.prove(cst, label, xs[j], np.stack([gh[0], gh[1], A[j], B[j]]), entropy[j].tobytes())
            assert chal[j].tobytes() == ec.tobytes() and (resp[j] == er).all() and (coms[j] == ek).all()
        ts = _fresh(label, n)
        assert not T.verify_compact_batch(eng, st, ts, inst, gh, chal, resp).any()
        ts = _fresh(label, n)
        T.batch_verify(eng, st, ts, inst, gh, coms, resp)
        swapped = np.ascontiguousarray(inst[::-1])                      # A and B exchanged: every proof fails
        ts = _fresh(label, n)
        assert T.verify_compact_batch(eng, st, ts, swapped, gh, chal, resp).all()
        ts = _fresh(label, n)
        with pytest.raises(T.VerificationFailure):
            T.batch_verify(eng, st, ts, swapped, gh, coms, resp)
    finally:
        T.set_fused_min_batch(32)


def test_w64_statement_fused_equals_host_route(eng):
    """BASELINE config 5's wide statement Q = sum_{i<64} x_i G_i: 64 secrets, 64 common points, one 64-term constraint
    (the transcript program has 64 rekeys + 64 fills: 150+ Keccak permutations per proof)."""
    n = 300
    rng = np.random.default_rng(64)
    names = [f"x_{i}" for i in range(64)]
    gens = [f"G_{i}" for i in range(64)]
    mod = T.define_proof("w64", b"W64", names, ["Q"], gens, [("Q", [(names[i], gens[i]) for i in range(64)])])
    st = mod.statement
    G, _ = _random_points(64, rng)
    xs = rng.integers(0, 256, size=(n, 64, 32), dtype=np.uint8)
    xs[:, :, 31] &= 0x0f
    off = (np.arange(n + 1, dtype=np.uint64) * 64).astype(np.uint32)
    Q, stq = C.msm_many(off, xs.reshape(-1, 32), np.tile(np.arange(64, dtype=np.uint32), n), G, 0)
    assert not stq.any()
    inst = np.ascontiguousarray(Q[None])
    entropy = rng.integers(0, 256, size=(n, 32), dtype=np.uint8)
    w = rng.integers(0, 256, size=(1, n, 16), dtype=np.uint8)
    out = {}
    for route, thr in (("host", NEVER), ("fused", 0)):
        T.set_fused_min_batch(thr)
        ts = _fresh(b"wide", n)
        chal, resp, coms = T.prove_batch(eng, st, ts, xs, inst, G, entropy)
        ts2 = _fresh(b"wide", n)
        res = T.verify_compact_batch(eng, st, ts2, inst, G, chal, resp)
        ts3 = _fresh(b"wide", n)
        ok, coeffs = T.batch_verify_coeffs(eng, st, ts3, inst, G, coms, resp, w)
        # verify_batchable per proof: 66 operands per MSM -- more than the window-split Straus walk takes (64), so the operand split runs
        bad = resp.copy(); bad[7, 63, 0] ^= 1
        each = T.verify_batchable_each(eng, st, _fresh(b"wide", n), inst, G, coms, bad, np.ascontiguousarray(w.transpose(1, 0, 2)))
        out[route] = (chal, resp, coms, ts[:, :203], res, ts2[:, :203], coeffs, ts3[:, :203], each)
        assert ok and not res.any() and each[7] == 1 and each.sum() == 1
    T.set_fused_min_batch(32)
    for a, b in zip(out["host"], out["fused"]):
        assert (a == b).all()


def test_fused_entry_points_reject_bad_arguments(eng):
    """The raw C entry points fail closed: negative return code, message in zkp_last_error, nothing marked verified."""
    import ctypes
    from zkp_amd.engine import FusedStatement, load_library
    lib = load_library()
    fst = FusedStatement(b"DLEQ proof", [b"x"], [(b"A", False), (b"B", False), (b"H", False), (b"G", True)],
                         [(0, [(0, 3)]), (1, [(0, 2)])])
    n = 4
    ts = np.stack([T.Transcript(b"t").state] * n)
    z = lambda *s: np.zeros(s, np.uint8)
    verdict = ctypes.c_int(0)
    vp = lambda a: a.ctypes.data_as(ctypes.c_void_p)
    # NULL inputs
    rc = lib.zkp_fused_batch_verify(eng._h, ctypes.byref(fst.c), ctypes.c_uint32(n), vp(ts), None, None, None, None, None, ctypes.byref(verdict), None)
    assert rc < 0 and b"NULL" in lib.zkp_last_error()
    # transcripts at different STROBE positions
    t2 = T.Transcript(b"t")
    t2.append_message(b"m", b"xyz")
    ragged = ts.copy()
    ragged[2] = t2.state
    res = z(n)
    rc = lib.zkp_fused_verify_compact(eng._h, ctypes.byref(fst.c), ctypes.c_uint32(n), vp(ragged), vp(z(3, n, 32)), vp(z(1, 32)), vp(z(n, 32)),
                                      vp(z(n, 1, 32)), vp(res))
    assert rc < 0 and b"STROBE" in lib.zkp_last_error()
    # a statement whose constraint names a point that does not exist
    bad = FusedStatement(b"DLEQ proof", [b"x"], [(b"A", False), (b"G", True)], [(0, [(0, 1)])])
    bad._pt[0] = 7
    rc = lib.zkp_fused_verify_compact(eng._h, ctypes.byref(bad.c), ctypes.c_uint32(n), vp(ts), vp(z(1, n, 32)), vp(z(1, 32)), vp(z(n, 32)),
                                      vp(z(n, 1, 32)), vp(res))
    assert rc < 0 and b"out of range" in lib.zkp_last_error()
    # all-zero (identity) encodings everywhere: the call works and every proof is rejected
    res[:] = 0
    rc = lib.zkp_fused_verify_compact(eng._h, ctypes.byref(fst.c), ctypes.c_uint32(n), vp(ts.copy()), vp(z(3, n, 32)), vp(z(1, 32)), vp(z(n, 32)),
                                      vp(z(n, 1, 32)), vp(res))
    assert rc == 0 and res.all()


@pytest.mark.parametrize("shape", ["no_constraints", "lhs_only", "shared_secret_many_points"])
def test_degenerate_statements_fused_equals_host_route(eng, shape):
    """Statements at the edges of what the constraint API accepts: points but no constraints (nothing to prove, the
    transcript still binds the points); a constraint with an empty right-hand side (a false statement: must be rejected
    by every verifier); one secret over many points."""
    n = 70
    rng = np.random.default_rng(5)
    pts, _ = _random_points(6, rng)
    st = T.Statement(b"edge")
    if shape == "no_constraints":
        st.add_secret(b"x")
        st.add_point(b"P", False)
        st.add_point(b"G", True)
        inst, common = np.ascontiguousarray(np.repeat(pts[0:1][None], n, 1)), pts[1:2].copy()
        secrets = rng.integers(0, 256, size=(n, 1, 32), dtype=np.uint8); secrets[:, :, 31] &= 0x0f
    elif shape == "lhs_only":
        p = st.add_point(b"P", False)
        st.constrain(p, [])
        inst, common = np.ascontiguousarray(np.repeat(pts[0:1][None], n, 1)), np.zeros((0, 32), np.uint8)
        secrets = np.zeros((n, 0, 32), np.uint8)
    else:
        x = st.add_secret(b"x")
        gs = [st.add_point(b"G%d" % i, True) for i in range(5)]
        q = st.add_point(b"Q", False)
        st.constrain(q, [(x, g) for g in gs])
        secrets = rng.integers(0, 256, size=(n, 1, 32), dtype=np.uint8); secrets[:, :, 31] &= 0x0f
        common = pts[:5].copy()
        off = (np.arange(n + 1, dtype=np.uint64) * 5).astype(np.uint32)
        Q, _ = C.msm_many(off, np.repeat(secrets[:, 0], 5, axis=0), np.tile(np.arange(5, dtype=np.uint32), n), common, 0)
        inst = np.ascontiguousarray(Q[None])
    entropy = rng.integers(0, 256, size=(n, 32), dtype=np.uint8)
    out = {}
    for route, thr in (("host", NEVER), ("fused", 0)):
        T.set_fused_min_batch(thr)
        ts = _fresh(b"edge-case", n)
        chal, resp, coms = T.prove_batch(eng, st, ts, secrets, inst, common, entropy)
        ts2 = _fresh(b"edge-case", n)
        res = T.verify_compact_batch(eng, st, ts2, inst, common, chal, resp)
        w = rng.integers(0, 256, size=(max(st.nc, 1), n, 16), dtype=np.uint8)[: st.nc]
        ts3 = _fresh(b"edge-case", n)
        try:
            T.batch_verify(eng, st, ts3, inst, common, coms, resp, np.ascontiguousarray(w))
            ok = True
        except T.VerificationFailure:
            ok = False
        out[route] = (chal, resp, coms, ts[:, :203], res, ts2[:, :203], np.array([ok]))
    T.set_fused_min_batch(32)
    for a, b in zip(out["host"], out["fused"]):
        assert a.shape == b.shape and (a == b).all()
    # "P = (empty sum)" is a false statement for P != identity (and an identity P is refused by the verifier's transcript):
    # the prover's commitment is the identity, the verifier recomputes (-c) P; every route must reject
    assert out["fused"][4].all() if shape == "lhs_only" else not out["fused"][4].any()
    assert out["fused"][6][0] == (shape != "lhs_only")


@pytest.mark.parametrize("seed", range(14))
def test_random_statements_fused_equals_host_route_and_oracle(eng, seed):
    """Randomly shaped statements through both routes and the oracle: 1-6 secrets, 0-4 common points, free and derived
    instance points ALLOCATED IN SHUFFLED ORDER (common and instance interleaved, like the constraint API allows),
    1-4 constraints with 1-5 terms each, secrets and points reused across constraints, a point that no constraint uses."""
    rng = np.random.default_rng(9000 + seed)
    # (seeds 8 ..: batches large enough for the one-launch statement classifier -- 1,024 terms -- which is also what pairs the verifier's terms onto shared
    #  doubling chains, ZKP_OPT_JOINT_LADDER: left-hand sides with free points of one or several uses, with each other, or with nothing)
    n = int(rng.integers(33, 120)) if seed < 8 else int(rng.integers(260, 420))
    m = int(rng.integers(1, 7))
    n_common = int(rng.integers(0, 5))
    n_free = int(rng.integers(1, 4))                # instance points that are given (random per proof)
    nc = int(rng.integers(1, 5))
    if n_common + n_free == 0:
        n_free = 1
    names = [("K%d" % i, True) for i in range(n_common)] + [("F%d" % i, False) for i in range(n_free)] + [("L%d" % k, False) for k in range(nc)]
    names.append(("U", bool(rng.integers(0, 2))))   # used by no constraint, must still decode
    order = rng.permutation(len(names))
    rhs_pool = list(range(n_common + n_free))
    cons = []
    for k in range(nc):
        terms = [(int(rng.integers(0, m)), int(rng.choice(rhs_pool))) for _ in range(int(rng.integers(1, 6)))]
        cons.append((n_common + n_free + k, terms))
    # values
    secrets = rng.integers(0, 256, size=(n, m, 32), dtype=np.uint8)
    secrets[:, :, 31] &= 0x0f
    commonv, _ = _random_points(n_common + 1, rng)                       # + U if common
    freev = [_random_points(n, rng)[0] for _ in range(n_free + 1)]       # per-proof values (+ U if instance)
    val = {}                                                             # name index -> [n][32] or [32]
    for i in range(n_common):
        val[i] = commonv[i]
    for i in range(n_free):
        val[n_common + i] = freev[i]
    for k, (lhs, terms) in enumerate(cons):
        table = np.concatenate([commonv[:n_common]] + [freev[i] for i in range(n_free)])      # common, then free rows [n_free][n]
        off = (np.arange(n + 1, dtype=np.uint64) * len(terms)).astype(np.uint32)
        sc = np.stack([secrets[:, s] for s, _ in terms], axis=1).reshape(n * len(terms), 32)
        pidx = np.array([[p if p < n_common else n_common + (p - n_common) * n + j for _, p in terms] for j in range(n)], np.uint32).reshape(-1)
        L, st_ = C.msm_many(off, sc, pidx, table, 0)
        assert not st_.any()
        val[lhs] = L
    u_idx = len(names) - 1
    val[u_idx] = commonv[n_common] if names[u_idx][1] else freev[n_free]
    # statements in the shuffled allocation order
    st = T.Statement(b"random statement %d" % seed)
    svars = [st.add_secret(b"s%d" % i) for i in range(m)]
    pvar = {}
    for a in order:
        pvar[int(a)] = st.add_point(names[a][0].encode(), names[a][1])
    for lhs, terms in cons:
        st.constrain(pvar[lhs], [(svars[s], pvar[p]) for s, p in terms])
    cst = C.Statement(b"random statement %d" % seed, ["s%d" % i for i in range(m)], [names[a] for a in order],
                      [(names[lhs][0], [("s%d" % s, names[p][0]) for s, p in terms]) for lhs, terms in cons])
    inst_rows = [val[int(a)] for a in order if not names[a][1]]
    common_rows = [val[int(a)] for a in order if names[a][1]]
    inst = np.ascontiguousarray(np.stack(inst_rows))
    common = np.ascontiguousarray(np.stack(common_rows)) if common_rows else np.zeros((0, 32), np.uint8)
    entropy = rng.integers(0, 256, size=(n, 32), dtype=np.uint8)
    w = rng.integers(0, 256, size=(nc, n, 16), dtype=np.uint8)
    label = b"fuzz"
    out = {}
    for route, thr in (("host", NEVER), ("fused", 0)):
        T.set_fused_min_batch(thr)
        ts = _fresh(label, n)
        chal, resp, coms = T.prove_batch(eng, st, ts, secrets, inst, common, entropy)
        ts2 = _fresh(label, n)
        res = T.verify_compact_batch(eng, st, ts2, inst, common, chal, resp)
        ts3 = _fresh(label, n)
        ok, coeffs = T.batch_verify_coeffs(eng, st, ts3, inst, common, coms, resp, w)
        ts4 = _fresh(label, n)
        each = T.verify_batchable_each(eng, st, ts4, inst, common, coms, resp, np.ascontiguousarray(w.transpose(1, 0, 2)))
        bad = resp.copy()
        bad[n // 2, cons[0][1][0][0], 0] ^= 1                             # (a response some constraint uses: an unused one changes no commitment)
        ts5 = _fresh(label, n)
        res_bad = T.verify_compact_batch(eng, st, ts5, inst, common, chal, bad)
        out[route] = (chal, resp, coms, ts[:, :203], res, ts2[:, :203], coeffs, ts3[:, :203], each, res_bad)
        assert ok and not res.any() and not each.any()
        assert res_bad[n // 2] == 1 and res_bad.sum() == 1
    T.set_fused_min_batch(32)
    for a, b in zip(out["host"], out["fused"]):
        assert (a == b).all()
    # the oracle, in the same allocation order (its point list is the allocation order)
    for j in (0, n - 1):
        pts_j = np.stack([val[int(a)] if names[a][1] else val[int(a)][j] for a in order])
        ec, er, ek, _ = C.prove(cst, label, secrets[j], pts_j, entropy[j].tobytes())
        assert out["fused"][0][j].tobytes() == ec.tobytes() and (out["fused"][1][j] == er).all() and (out["fused"][2][j] == ek).all()


def test_verify_batchable_coefficient_fold_matches_model(eng):
    """Row a5 (verifier.rs:144-160): the per-proof coefficient vector the DEVICE folds -- coeffs[np + k] = -r_k,
    coeffs[lhs_k] += r_k (-c), coeffs[pt] += r_k resp[sc] -- against the model's restatement, scalar for scalar, in the operand
    order of verifier.rs:162-166 (points, then commitments).  The verdicts alone would not notice a fold that is wrong
    in a way the MSM forgives."""
    from zkp_amd.engine import FusedStatement
    n = 40
    rng = random.Random(55)
    nrng = np.random.default_rng(55)
    mst, mod = M.cmz_statement(10), T.cmz_module(10)
    from tests.test_gpu_toolbox import _cmz_batch
    _, secrets, inst, common = _cmz_batch(n, 56)
    label = b"FoldTest"
    entropy = nrng.integers(0, 256, size=(n, 32), dtype=np.uint8)
    ts = _fresh(label, n)
    chal, resp, coms = T.prove_batch(eng, mod.statement, ts, secrets, inst, common, entropy)
    w16 = nrng.integers(0, 256, size=(n, 11, 16), dtype=np.uint8)
    w16[3] = 0xff                                                   # u128::MAX weights
    w16[4] = 0                                                      # zero weights
    # the statement in the fused ABI's form: secrets, points (instance first, then common: define_proof!'s order), constraints
    names = mod.instance + mod.common
    fst = FusedStatement(mod.label, [s.encode() for s in mod.secrets], [(k.encode(), k in mod.common) for k in names],
                         [(names.index(l), [(mod.secrets.index(s), names.index(p)) for s, p in lc]) for l, lc in mod.constraints])
    ts = _fresh(label, n)
    res, co = eng.fused_verify_batchable_coeffs(fst, ts, inst, common, coms, resp, w16)
    assert not res.any()
    for j in (0, 3, 4, 17, n - 1):
        ver = mst.build_verifier(M.Transcript(label), {k: (inst[mod.instance.index(k), j] if k in mod.instance else common[mod.common.index(k)]).tobytes()
                                                        for k in names})
        proof = M.BatchableProof([c.tobytes() for c in coms[j]], [int.from_bytes(r.tobytes(), "little") for r in resp[j]])
        weights = [int.from_bytes(w16[j, k].tobytes(), "little") for k in range(11)]
        for i, com in enumerate(proof.commitments):                 # verifier.rs:134-142 (what verify_batchable does before the fold)
            ver.transcript.validate_and_append_blinding_commitment(ver.point_labels[ver.constraints[i][0]], com)
        minus_c = (-ver.transcript.get_challenge(b"chal")) % M.L
        want = [0] * (len(ver.points) + 11)
        for i, (lhs, lc) in enumerate(ver.constraints):             # verifier.rs:151-160
            r = weights[i]
            want[len(ver.points) + i] = (want[len(ver.points) + i] - r) % M.L
            want[lhs] = (want[lhs] + r * minus_c) % M.L
            for sv, pv in lc:
                want[pv] = (want[pv] + r * proof.responses[sv]) % M.L
        # the model numbers points in allocation order (instance, then common); the device by point id (common first)
        ns, ni = len(mod.common), len(mod.instance)
        got = [int.from_bytes(co[j, i].tobytes(), "little") for i in range(ns + ni + 11)]
        by_alloc = got[ns:ns + ni] + got[:ns] + got[ns + ni:]
        assert by_alloc == want, j
        ver2 = mst.build_verifier(M.Transcript(label), {k: (inst[mod.instance.index(k), j] if k in mod.instance else common[mod.common.index(k)]).tobytes()
                                                         for k in names})
        ver2.verify_batchable(proof, weights)                        # and the model accepts the same proof with the same weights


def test_long_labels_cross_strobe_blocks_fused_host_oracle_model(eng):
    """Labels of 165 / 166 / 167 / 400 bytes put label and length prefix on either side of a 166-byte STROBE block: the
    GPU transcript programs, the host Merlin, the C oracle and the model must frame them identically (4-way)."""
    n = 40
    rng = random.Random(66)
    nrng = np.random.default_rng(66)
    lab = lambda k, c: (c * k).encode()
    proof_label = lab(300, "L")
    sec_names, pt_names = [lab(165, "x").decode(), lab(167, "y").decode()], [lab(166, "A").decode(), lab(400, "B").decode(), "G", lab(1, "H").decode()]
    G, H = M.BASEPOINT, M.ristretto_hash_from_bytes_sha512(b"long labels")
    mod = T.define_proof("long", proof_label, sec_names, pt_names[:2], pt_names[2:],
                         [(pt_names[0], [(sec_names[0], "G"), (sec_names[1], pt_names[3])]), (pt_names[1], [(sec_names[1], "G")])])
    cst = C.Statement(proof_label, sec_names, [(pt_names[0], False), (pt_names[1], False), ("G", True), (pt_names[3], True)],
                      [(pt_names[0], [(sec_names[0], "G"), (sec_names[1], pt_names[3])]), (pt_names[1], [(sec_names[1], "G")])])
    xs = [(rng.randrange(1, M.L), rng.randrange(1, M.L)) for _ in range(n)]
    Ge, He = M.ristretto_encode(G), M.ristretto_encode(H)
    A = [M.ristretto_encode(M.pt_add(M.pt_mul(x, G), M.pt_mul(y, H))) for x, y in xs]
    B = [M.ristretto_encode(M.pt_mul(y, G)) for _, y in xs]
    secrets = np.frombuffer(b"".join(x.to_bytes(32, "little") + y.to_bytes(32, "little") for x, y in xs), np.uint8).reshape(n, 2, 32)
    inst = np.frombuffer(b"".join(A + B), np.uint8).reshape(2, n, 32)
    common = np.frombuffer(Ge + He, np.uint8).reshape(2, 32)
    entropy = nrng.integers(0, 256, size=(n, 32), dtype=np.uint8)
    label = b"T" * 170                                               # the transcript label crosses a block as well
    ts = _fresh(label, n)
    chal, resp, coms = T.prove_batch(eng, mod.statement, ts, secrets, inst, common, entropy)        # fused route
    old = T.lib().zkp_toolbox_get_fused_min_batch()
    try:
        T.lib().zkp_toolbox_set_fused_min_batch(0xffffffff)
        ts_h = _fresh(label, n)
        chal_h, resp_h, coms_h = T.prove_batch(eng, mod.statement, ts_h, secrets, inst, common, entropy)   # host Merlin
    finally:
        T.lib().zkp_toolbox_set_fused_min_batch(old)
    assert (chal == chal_h).all() and (resp == resp_h).all() and (coms == coms_h).all() and (ts[:, :203] == ts_h[:, :203]).all()
    for j in (0, 1, n - 1):                                          # C oracle
        ec, er, ek, _ = C.prove(cst, label, secrets[j], np.stack([inst[0, j], inst[1, j], common[0], common[1]]), entropy[j].tobytes())
        assert chal[j].tobytes() == ec.tobytes() and (resp[j] == er).all() and (coms[j] == ek).all(), j
    # model (pure Python), one proof
    mp = M.Prover(proof_label, M.Transcript(label))
    vx, vy = mp.allocate_scalar(sec_names[0].encode(), xs[0][0]), mp.allocate_scalar(sec_names[1].encode(), xs[0][1])
    vA, _ = mp.allocate_point(pt_names[0].encode(), M.ristretto_decode(A[0])); vB, _ = mp.allocate_point(pt_names[1].encode(), M.ristretto_decode(B[0]))
    vG, _ = mp.allocate_point(b"G", G); vH, _ = mp.allocate_point(pt_names[3].encode(), H)
    mp.constrain(vA, [(vx, vG), (vy, vH)])
    mp.constrain(vB, [(vy, vG)])
    want = mp.prove_compact(entropy[0].tobytes())
    assert chal[0].tobytes() == M.sc_to_bytes(want.challenge) and [r.tobytes() for r in resp[0]] == [M.sc_to_bytes(r) for r in want.responses]
    ts = _fresh(label, n)
    assert not T.verify_compact_batch(eng, mod.statement, ts, inst, common, chal, resp).any()
    ts = _fresh(label, n)
    T.batch_verify(eng, mod.statement, ts, inst, common, coms, resp)


def test_debug_transcript_env_lists_the_compiled_program():
    """ZKP_DEBUG_TRANSCRIPT=1 on the fused route: the compiled transcript program of the flow is decoded to stderr when its
    plan is built (absorbs / keys with their input buffers and strides, emits, clone / restore, Keccak permutations)."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import numpy as np\n"
            "from zkp_amd import toolbox as T\n"
            "from zkp_amd.engine import Engine\n"
            "from tests.test_gpu_fused import _dleq_batch, _fresh\n"
            "eng = Engine(0)\n"
            "n = 40\n"
            "mod, x, A, B, H = _dleq_batch(n, 3)\n"
            "inst = np.ascontiguousarray(np.stack([A, B, H]))\n"
            "base = np.frombuffer(bytes.fromhex('e2f2ae0a6abc4e71a884a961c500515f58e30b6aa582dd8db6a65945e08d2d76'), np.uint8).reshape(1, 32)\n"
            "ts = _fresh(b'dbg', n)\n"
            "T.prove_batch(eng, mod.statement, ts, x, inst, base, np.zeros((n, 32), np.uint8))\n")
    r = subprocess.run([sys.executable, "-c", code], cwd=root, env=dict(os.environ, ZKP_DEBUG_TRANSCRIPT="1"), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    err = r.stderr
    assert "[transcript program] flow=P program A" in err and "[transcript program] flow=P program B" in err
    assert "KEY state.word[" in err and "<- secrets[j * 32 + 0]" in err            # the witness rekeys the RNG clone (prover.rs:80)
    assert "<- entropy[j * 32 + 0]" in err and "EMIT state.word[" in err and "-> wide(blindings)" in err and "-> wide(challenge)" in err
    assert "SAVE(clone <- state)" in err and "RESTORE(state <- clone)" in err and "KECCAK-F" in err
    assert "6 Keccak-f permutations per proof" in err or "Keccak-f permutations per proof" in err


@pytest.mark.parametrize("n", [3, 300])
def test_verify_batchable_straus_lane_counts_agree(n):
    """ZKP_OPT_EACH_STRAUS: verify_batchable's per-proof MSM (verifier.rs:162-166) as one Straus walk per proof with 1 .. 8 lanes
    per proof (operand split) or split into 1 .. 64 window parts per proof (the default below 65,536 proofs), and the round-2 schedule (a ladder per operand, option 0): the same verdicts for valid proofs, tampered responses,
    wrong / identity / undecodable points and commitments, a non-canonical response; equal to the oracle's."""
    from zkp_amd.engine import Engine
    mod, secrets, inst, common = _cmz_batch(n, 91)
    st = mod.statement
    label = b"straus"
    rng = np.random.default_rng(92)
    e0 = Engine(0)
    ts = _fresh(label, n)
    chal, resp, coms = T.prove_batch(e0, st, ts, secrets, inst, common, rng.integers(0, 256, size=(n, 32), dtype=np.uint8))
    e0.close()
    w = rng.integers(0, 256, size=(n, st.nc, 16), dtype=np.uint8)
    resp, coms, inst = resp.copy(), coms.copy(), inst.copy()
    bad = set()
    if n >= 300:
        resp[5, 2, 0] ^= 1; bad.add(5)
        coms[17, 3] = coms[18, 3]; bad.add(17)
        coms[40, 0] = 0; bad.add(40)
        coms[41, 10] = np.frombuffer(bytes([1] + [0] * 31), np.uint8); bad.add(41)
        inst[12, 77] = np.frombuffer(bytes([1] + [0] * 31), np.uint8); bad.add(77)
        inst[0, 78] = inst[0, 79]; bad.add(78)
        resp[100, 20] = np.frombuffer((int.from_bytes(resp[100, 20].tobytes(), "little") + M.L).to_bytes(32, "little"), np.uint8); bad.add(100)
    else:
        resp[1, 0, 0] ^= 1; bad.add(1)
    want = np.array([1 if j in bad else 0 for j in range(n)], np.uint8)
    cst = C.Statement.from_model(M.cmz_statement(10))
    for j in sorted(bad) + [0]:
        assert C.verify_batchable(cst, label, np.concatenate([inst[:, j], common]), coms[j], resp[j], w[j]) == want[j]
    T.set_fused_min_batch(0)
    try:
        for opt in (2**64 - 1, 0, 1, 2, 3, 4, 8, 0x201, 0x204, 0x210, 0x240):       # 0x200 + P: the walk split into P window parts per proof (default: 32)
            e = Engine(0)
            e.set_option(10, opt)
            ts = _fresh(label, n)
            got = T.verify_batchable_each(e, st, ts, inst, common, coms, resp, w)
            e.close()
            assert (got == want).all(), (opt, np.nonzero(got != want)[0][:8])
    finally:
        T.set_fused_min_batch(32)


def test_verify_batchable_straus_more_lanes_than_operands_dleq():
    """ZKP_OPT_EACH_STRAUS = 8 lanes per proof on a DLEQ-sized statement (K = 4 points + 2 commitments = 6 operands): lanes
    without an operand used to fetch an unwritten LDS word and gather past the proof's table; the lane count is clamped to K now.  Every
    option value gives the oracle's verdicts, for valid and for tampered proofs."""
    from zkp_amd.engine import Engine
    n = 200
    mod, x, A, B, H = _dleq_batch(n, 17)
    st = mod.statement
    inst = np.stack([A, B, H])
    G = np.frombuffer(BASEPOINT, np.uint8).reshape(1, 32)
    label = b"straus-dleq"
    rng = np.random.default_rng(18)
    e0 = Engine(0)
    chal, resp, coms = T.prove_batch(e0, st, _fresh(label, n), x, inst, G, rng.integers(0, 256, size=(n, 32), dtype=np.uint8))
    e0.close()
    w = rng.integers(0, 256, size=(n, st.nc, 16), dtype=np.uint8)
    resp, coms = resp.copy(), coms.copy()
    resp[3, 0, 1] ^= 2
    coms[50, 1] = coms[51, 1]
    coms[199, 0] = 0
    want = np.zeros(n, np.uint8)
    want[[3, 50, 199]] = 1
    cst = C.Statement.from_model(M.dleq_statement())
    for j in (0, 3, 50, 199):
        assert C.verify_batchable(cst, label, np.concatenate([inst[:, j], G]), coms[j], resp[j], w[j]) == want[j]
    T.set_fused_min_batch(0)
    try:
        for opt in (2**64 - 1, 0, 1, 5, 6, 7, 8, 0x202, 0x240):
            e = Engine(0)
            e.set_option(10, opt)
            got = T.verify_batchable_each(e, st, _fresh(label, n), inst, G, coms, resp, w)
            e.close()
            assert (got == want).all(), (opt, np.nonzero(got != want)[0][:8])
    finally:
        T.set_fused_min_batch(32)


@pytest.mark.gpu
def test_verify_compact_joint_ladder_equals_separate_terms():
    """ZKP_OPT_JOINT_LADDER (round 6): in the verifier's constraints  commitment = sum s_i P_i - c LHS  (verifier.rs:95-106) the left-hand side's doubling chain
    also carries one other per-proof term of the constraint (CMZ: P in the ten C_i constraints -- which then needs no comb table --, Q next to V; DLEQ with a
    per-proof H: H next to B).  Both settings must accept exactly the proofs the oracle's verifier accepts: all of a valid batch (every recomputed commitment
    bit-exact, or the challenge would differ), and none of the proofs with a tampered response of a PAIRED term, a tampered challenge, a tampered left-hand
    side, an undecodable partner point.  Sizes: the statement classifier (what pairs the terms) runs from 1,024 terms on; 20 proofs stay below it."""
    from zkp_amd.engine import Engine
    rng = np.random.default_rng(77)
    cases = []
    for n in (20, 333, 1500):
        mod, secrets, inst, common = _cmz_batch(n, 91 + n)
        cases.append((b"joint-cmz", mod.statement, C.Statement.from_model(M.cmz_statement(10)), secrets, inst, common, n))
    n = 700
    mod, x, A, B, H = _dleq_batch(n, 23)
    cases.append((b"joint-dleq", mod.statement, C.Statement.from_model(M.dleq_statement()), x, np.stack([A, B, H]), np.frombuffer(BASEPOINT, np.uint8).reshape(1, 32), n))
    T.set_fused_min_batch(0)
    try:
        for label, st, cst, secrets, inst, common, n in cases:
            e0 = Engine(0)
            chal, resp, coms = T.prove_batch(e0, st, _fresh(label, n), secrets, inst, common, rng.integers(0, 256, size=(n, 32), dtype=np.uint8))
            e0.close()
            ni = inst.shape[0]
            bad_resp, bad_chal, bad_inst = resp.copy(), chal.copy(), inst.copy()
            want = np.zeros(n, np.uint8)
            for i in range(resp.shape[1]):                               # one proof per secret: its response off by one bit
                bad_resp[(3 + i) % n, i, (5 * i) % 31] ^= 1 << (i % 8)
                want[(3 + i) % n] = 1
            results = {}
            # pairs + tables of multiples for points whose terms all ride / pairs only / separate terms.  The tables are for wide or overlapping calls (a chain of
            # 127 additions per table): synchronous calls below 16,384 proofs leave them out unless they run on the throughput schedule (ZKP_OPT_SYNC_SCHEDULE = 1)
            for opt, sched in ((1, 1), (1, 0), (2, 1), (0, 0)):
                e = Engine(0)
                e.set_option(17, opt)
                e.set_option(14, sched)
                ok = T.verify_compact_batch(e, st, _fresh(label, n), inst, common, chal, resp)
                r1 = T.verify_compact_batch(e, st, _fresh(label, n), inst, common, chal, bad_resp)
                bc = chal.copy(); bc[n - 1, 0] ^= 1
                r2 = T.verify_compact_batch(e, st, _fresh(label, n), inst, common, bc, resp)
                r3 = []
                for p in range(ni):                                      # every instance point in turn: another proof's point / not a point at all
                    bi = inst.copy()
                    bi[p, 1] = inst[p, 0]
                    bi[p, n - 2] = np.frombuffer(bytes([1] + [0] * 31), np.uint8)
                    r3.append(T.verify_compact_batch(e, st, _fresh(label, n), bi, common, chal, resp))
                e.close()
                assert not ok.any(), (label, n, opt, sched, np.nonzero(ok)[0][:8])
                assert (r1 == want).all(), (label, n, opt, sched, np.nonzero(r1 != want)[0][:8])
                assert r2[n - 1] == 1 and r2.sum() == 1
                for p, r in enumerate(r3):
                    assert r[1] == 1 and r[n - 2] == 1 and r.sum() == 2, (label, n, opt, sched, p, np.nonzero(r)[0][:8])
                results[opt, sched] = (ok, r1, r2, r3)
            for j in (0, 3, 4, n - 1):                                   # the oracle's verifier on the same bytes (define_proof!'s order: instance, then common)
                pts = np.concatenate([inst[:, j], common])
                assert C.verify_compact(cst, label, pts, chal[j], resp[j]) == 0
                assert C.verify_compact(cst, label, pts, chal[j], bad_resp[j]) == want[j]
    finally:
        T.set_fused_min_batch(32)
