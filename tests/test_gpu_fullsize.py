"""Parity at the FULL sizes BASELINE.json names (per-GPU shares of configs 3, 4, 5), through properties that do
not need the oracle to redo the whole computation:

* every point has a known discrete log (P_j = k_j * B), so an MSM result must equal (sum s_i k_i mod l) * B --
  one oracle scalar multiplication for millions of terms;
* checksum of checksums: the sum of ALL outputs of a zkp_msm_many call (taken with a second GPU MSM with unit
  scalars) must equal the point predicted from the discrete logs;
* a random sample of individual outputs is recomputed by the C oracle bit for bit;
* linearity: MSM(s, P) + MSM(s', P) = MSM(s + s', P).
"""
import numpy as np
import pytest

from oracle import cbind as C
from oracle import model as M

pytestmark = pytest.mark.gpu
BASE = np.frombuffer(bytes.fromhex("e2f2ae0a6abc4e71a884a961c500515f58e30b6aa582dd8db6a65945e08d2d76"), np.uint8).reshape(1, 32)


@pytest.fixture(scope="module")
def eng():
    from zkp_amd.engine import Engine
    e = Engine(0)
    yield e
    e.close()


def rand_scalars(rng, k):
    s = rng.integers(0, 256, size=(k, 32), dtype=np.uint8)
    s[:, 31] &= 0x0f
    return s


def to_ints(arr):
    """[n][32] uint8 little-endian -> python ints (vectorised through 4 uint64 words)"""
    w = arr.view(np.uint64).reshape(-1, 4)
    return [int(a) | (int(b) << 64) | (int(c) << 128) | (int(d) << 192) for a, b, c, d in w]


def expected_point(total_log):
    return M.ristretto_encode(M.pt_mul(total_log % M.L, M.BASEPOINT))


def make_points(eng, rng, n):
    """n valid points k_j * B made by the engine (checked against the oracle on a sample), with their logs."""
    ks = rand_scalars(rng, n)
    pts, st = eng.msm_many(np.arange(n + 1, dtype=np.uint32), ks, np.zeros(n, np.uint32), BASE, 0)
    assert not st.any()
    sample = rng.integers(0, n, size=8)
    exp, _ = C.msm_many(np.arange(9, dtype=np.uint32), ks[sample], np.zeros(8, np.uint32), BASE, 0)
    assert (pts[sample] == exp).all()
    return pts, to_ints(ks)


def test_config3_batch_msm_2p20_dleq(eng):
    """BatchVerifier MSM over 2^20 macro-DLEQ proofs: 1 + 5 * 2^20 = 5,242,881 terms (batch_verifier.rs:219)."""
    rng = np.random.default_rng(303)
    n = 1 + 5 * (1 << 20)
    distinct = 1 << 16
    pts, logs = make_points(eng, rng, distinct)
    idx = rng.integers(0, distinct, size=n)
    scal = rand_scalars(rng, n)
    got = eng.msm_optional(scal, pts[idx])
    logs_arr = np.array(logs, dtype=object)
    total = int(np.dot(np.array(to_ints(scal), dtype=object), logs_arr[idx]))
    assert got == expected_point(total)
    # linearity at full size: MSM(s) + MSM(s') == MSM(s + s')  (scalars < 2^252 so s + s' needs no reduction below 2^256)
    scal2 = rand_scalars(rng, n)
    got2 = eng.msm_optional(scal2, pts[idx])
    both = np.concatenate([np.frombuffer(got, np.uint8), np.frombuffer(got2, np.uint8)]).reshape(2, 32)
    one = np.zeros((2, 32), np.uint8)
    one[:, 0] = 1
    lhs = eng.msm_optional(one, both)
    total2 = total + int(np.dot(np.array(to_ints(scal2), dtype=object), logs_arr[idx]))
    assert lhs == expected_point(total2)
    # a single undecodable point anywhere -> None
    bad = pts[idx].copy()
    bad[n // 3] = np.frombuffer(bytes.fromhex("0100000000000000000000000000000000000000000000000000000000000000"), np.uint8)
    assert eng.msm_optional(scal, bad) is None


def _cmz_csr(n, n_common=11):
    """CSR of the CMZ commitment MSMs (as in bench.py): points [X_1..X_10, A, P_0, Q_0, P_1, Q_1, ...]"""
    import bench
    return bench.cmz_shape(n)


def test_config4_share_cmz_prove_524288(eng):
    """Per-GPU share of config 4: 524,288 CMZ proofs -> 5,767,168 commitment MSMs / 16,252,928 terms, constant-time
    schedule, with the common points on the fixed-base path."""
    rng = np.random.default_rng(404)
    n = 524288
    off, pidx, n_pts = _cmz_csr(n)
    pts, logs = make_points(eng, rng, n_pts)
    eng.prepare_fixed_points(pts[:11])
    blind = rand_scalars(rng, 31 * n)
    out, st = eng.msm_many(off, blind, pidx, pts, 1)
    assert not st.any()
    # sampled outputs against the C oracle, bit for bit (both MSM shapes: 2 terms and 11 terms)
    sample = np.concatenate([rng.integers(0, 11 * n, size=24), np.array([10, 21, 11 * n - 1])])
    for m in sample:
        lo, hi = int(off[m]), int(off[m + 1])
        exp, est = C.msm_many(np.array([0, hi - lo], np.uint32), blind[lo:hi], np.arange(hi - lo, dtype=np.uint32), pts[pidx[lo:hi]], 1)
        assert est[0] == 0 and (out[m] == exp[0]).all(), m
    # checksum of checksums: sum of all 5.7M outputs == (sum of all scalar * log) * B
    ones = np.zeros((11 * n, 32), np.uint8)
    ones[:, 0] = 1
    total_pt = eng.msm_optional(ones, out)
    logs_arr = np.array(logs, dtype=object)
    total = 0
    step = 1 << 20
    for a in range(0, 31 * n, step):                       # chunked exact big-integer dot product
        total += int(np.dot(np.array(to_ints(blind[a:a + step]), dtype=object), logs_arr[pidx[a:a + step]]))
    assert total_pt == expected_point(total)


def test_config5_share_w64_prove_32768(eng):
    """Per-GPU share of config 5: the wide statement Q = sum_{i<64} x_i G_i (64-term MSM per proof), 32,768 proofs,
    all 64 generators common (fixed-base path), and its batch-verification MSM of 64 + 2N terms."""
    rng = np.random.default_rng(505)
    n = 32768
    gens, glogs = make_points(eng, rng, 64)
    eng.prepare_fixed_points(gens)
    scal = rand_scalars(rng, 64 * n)
    off = (np.arange(n + 1, dtype=np.uint64) * 64).astype(np.uint32)
    pidx = np.tile(np.arange(64, dtype=np.uint32), n)
    out, st = eng.msm_many(off, scal, pidx, gens, 1)
    assert not st.any()
    ints = np.array(to_ints(scal), dtype=object).reshape(n, 64)
    gl = np.array(glogs, dtype=object)
    for m in list(rng.integers(0, n, size=6)) + [0, n - 1]:
        assert out[m].tobytes() == expected_point(int(np.dot(ints[m], gl)))
    ones = np.zeros((n, 32), np.uint8)
    ones[:, 0] = 1
    assert eng.msm_optional(ones, out) == expected_point(int(np.dot(ints.sum(axis=0), gl)))
    # the same products through the vartime schedule must give the same bytes
    out_v, st_v = eng.msm_many(off, scal, pidx, gens, 0)
    assert not st_v.any() and (out_v == out).all()
    # batch-verification shape: 64 static + (Q_j, commitment_j) rows -> 64 + 2N terms
    nb = 64 + 2 * n
    bsc = rand_scalars(rng, nb)
    bpts = np.concatenate([gens, out, out[::-1]])
    blogs = gl.tolist() + [int(np.dot(ints[m], gl)) % M.L for m in range(n)]
    blogs = np.array(blogs + blogs[64:][::-1], dtype=object)
    got = eng.msm_optional(bsc, bpts)
    assert got == expected_point(int(np.dot(np.array(to_ints(bsc), dtype=object), blogs)))


def test_config4_share_complete_flows_524288(eng):
    """Per-GPU share of config 4 through the COMPLETE flows (transcripts, scalars and MSMs on the device): 524,288 CMZ
    proofs proven, every one verified (verify_compact), the batch verified; a flipped bit anywhere fails the batch and is
    localised by verify_compact; sampled proofs equal the C oracle's byte for byte (same entropy)."""
    import bench
    from oracle import model as M
    from zkp_amd import toolbox as T
    n = 524288
    rng = np.random.default_rng(44)
    secrets, inst, common = bench.cmz_instance(eng, n, rng)
    mod = T.cmz_module(10)
    st = mod.statement
    label = b"config-4"
    entropy = rng.integers(0, 256, size=(n, 32), dtype=np.uint8)
    t0 = T.Transcript(label).state
    ts = np.repeat(t0[None], n, axis=0)
    chal, resp, coms = T.prove_batch(eng, st, ts, secrets, inst, common, entropy)
    cst = C.Statement.from_model(M.cmz_statement(10))
    for j in (0, 1, n // 2 + 7, n - 1):
        ec, er, ek, _ = C.prove(cst, label, secrets[j], np.concatenate([inst[:, j], common]), entropy[j].tobytes())
        assert chal[j].tobytes() == ec.tobytes() and (resp[j] == er).all() and (coms[j] == ek).all(), j
    ts = np.repeat(t0[None], n, axis=0)
    res = T.verify_compact_batch(eng, st, ts, inst, common, chal, resp)
    assert not res.any()
    ts = np.repeat(t0[None], n, axis=0)
    T.batch_verify(eng, st, ts, inst, common, coms, resp)
    k = 424242
    resp[k, 20, 31] ^= 0x01
    ts = np.repeat(t0[None], n, axis=0)
    with pytest.raises(T.VerificationFailure):
        T.batch_verify(eng, st, ts, inst, common, coms, resp)
    ts = np.repeat(t0[None], n, axis=0)
    res = T.verify_compact_batch(eng, st, ts, inst, common, chal, resp)
    assert res[k] == 1 and int(res.sum()) == 1


def test_config3_complete_flows_dleq_2p20(eng):
    """Config 3 through the complete flows: 2^20 DLEQ proofs (benches/zkp.rs:49: A = x G, B = x H with a per-proof H)
    proven and batch-verified on the device (the batch MSM has 1 + 5 N = 5,242,881 terms); a flipped bit fails the
    batch; sampled proofs equal the C oracle's byte for byte."""
    from zkp_amd import toolbox as T
    n = 1 << 20
    rng = np.random.default_rng(33)
    base = np.frombuffer(bytes.fromhex("e2f2ae0a6abc4e71a884a961c500515f58e30b6aa582dd8db6a65945e08d2d76"), np.uint8).reshape(1, 32)
    eng.prepare_fixed_points(base)
    x = rand_scalars(rng, n)
    iota = np.arange(n + 1, dtype=np.uint32)
    H, s1 = eng.msm_many(iota, rand_scalars(rng, n), np.zeros(n, np.uint32), base, 1)
    A, s2 = eng.msm_many(iota, x, np.zeros(n, np.uint32), base, 1)
    B, s3 = eng.msm_many(iota, x, np.arange(n, dtype=np.uint32), H, 1)
    assert not (s1.any() or s2.any() or s3.any())
    inst = np.ascontiguousarray(np.stack([A, B, H]))
    mod = T.dleq_module()
    st = mod.statement
    label = b"config-3"
    entropy = rng.integers(0, 256, size=(n, 32), dtype=np.uint8)
    t0 = T.Transcript(label).state
    ts = np.repeat(t0[None], n, axis=0)
    secrets = np.ascontiguousarray(x.reshape(n, 1, 32))
    chal, resp, coms = T.prove_batch(eng, st, ts, secrets, inst, base.copy(), entropy)
    cst = C.Statement.from_model(M.dleq_statement())
    for j in (0, 77777, n - 1):
        ec, er, ek, _ = C.prove(cst, label, secrets[j], np.concatenate([inst[:, j], base]), entropy[j].tobytes())
        assert chal[j].tobytes() == ec.tobytes() and (resp[j] == er).all() and (coms[j] == ek).all(), j
    ts = np.repeat(t0[None], n, axis=0)
    T.batch_verify(eng, st, ts, inst, base.copy(), coms, resp)
    coms[n // 3, 1, 5] ^= 0x40                      # most likely no longer a valid encoding -> failure either way
    ts = np.repeat(t0[None], n, axis=0)
    with pytest.raises(T.VerificationFailure):
        T.batch_verify(eng, st, ts, inst, base.copy(), coms, resp)


def test_config3_constraint_api_form_2p20(eng):
    """Config 3's other half at full size: the constraint-API form of benches/dleq.rs:188-241 -- the static points G, H are
    allocated BEFORE the instance points A, B, and H is shared by the whole batch, so the batch MSM has 2 + 4 N =
    4,194,306 terms.  2^20 proofs proven and batch-verified on the device; sampled proofs equal the C oracle's byte for byte;
    the GPU-built coefficient vector satisfies the checksum the statement implies; one flipped bit fails the batch."""
    from zkp_amd import toolbox as T
    n = 1 << 20
    rng = np.random.default_rng(34)
    base = np.frombuffer(bytes.fromhex("e2f2ae0a6abc4e71a884a961c500515f58e30b6aa582dd8db6a65945e08d2d76"), np.uint8).reshape(1, 32)
    hk = rand_scalars(rng, 1)
    Hs, s0 = eng.msm_many(np.arange(2, dtype=np.uint32), hk, np.zeros(1, np.uint32), base, 1)
    common = np.ascontiguousarray(np.concatenate([base, Hs]))            # G, H
    eng.prepare_fixed_points(common)
    x = rand_scalars(rng, n)
    iota = np.arange(n + 1, dtype=np.uint32)
    A, s1 = eng.msm_many(iota, x, np.zeros(n, np.uint32), common, 1)     # benches/dleq.rs:198-199: A = x G, B = x H
    B, s2 = eng.msm_many(iota, x, np.ones(n, np.uint32), common, 1)
    assert not (s0.any() or s1.any() or s2.any())
    st = T.Statement(b"DLEQProof")                                       # dleq.rs:206-216: scalar, static G, H, then instance A, B
    vx = st.add_secret(b"x")
    vg, vh = st.add_point(b"G", True), st.add_point(b"H", True)
    va, vb = st.add_point(b"A", False), st.add_point(b"B", False)
    st.constrain(va, [(vx, vg)])
    st.constrain(vb, [(vx, vh)])
    inst = np.ascontiguousarray(np.stack([A, B]))
    label = b"DLEQBatchTest"
    entropy = rng.integers(0, 256, size=(n, 32), dtype=np.uint8)
    t0 = T.Transcript(label).state
    ts = np.repeat(t0[None], n, axis=0)
    secrets = np.ascontiguousarray(x.reshape(n, 1, 32))
    chal, resp, coms = T.prove_batch(eng, st, ts, secrets, inst, common, entropy)
    cst = C.Statement(b"DLEQProof", ["x"], [("G", True), ("H", True), ("A", False), ("B", False)], [("A", [("x", "G")]), ("B", [("x", "H")])])
    for j in (0, 424242, n - 1):
        ec, er, ek, _ = C.prove(cst, label, secrets[j], np.stack([common[0], common[1], A[j], B[j]]), entropy[j].tobytes())
        assert chal[j].tobytes() == ec.tobytes() and (resp[j] == er).all() and (coms[j] == ek).all(), j
    w16 = rng.integers(0, 256, size=(2, n, 16), dtype=np.uint8)
    ts = np.repeat(t0[None], n, axis=0)
    ok, co = T.batch_verify_coeffs(eng, st, ts, inst, common, coms, resp, w16)
    assert ok and co.shape == (2 + 4 * n, 32)
    # size-independent property of the coefficient vector (batch_verifier.rs:173-206): the commitment rows are -r_ij, and the
    # static coefficients are  sum_j r_0j resp_j  (G)  and  sum_j r_1j resp_j  (H)  mod l
    ints = lambda a: np.array(to_ints(np.ascontiguousarray(a)), dtype=object)
    r0, r1 = ints(np.pad(w16[0], ((0, 0), (0, 16)))), ints(np.pad(w16[1], ((0, 0), (0, 16))))
    rs_ = ints(resp[:, 0])
    assert int.from_bytes(co[0].tobytes(), "little") == int(np.dot(r0, rs_)) % M.L
    assert int.from_bytes(co[1].tobytes(), "little") == int(np.dot(r1, rs_)) % M.L
    sample = rng.integers(0, n, size=64)
    for j in sample:
        assert int.from_bytes(co[2 + 2 * n + j].tobytes(), "little") == (-int(r0[j])) % M.L          # commitment row of constraint 0
        assert int.from_bytes(co[2 + 3 * n + j].tobytes(), "little") == (-int(r1[j])) % M.L
    resp[n // 5, 0, 31] ^= 1
    ts = np.repeat(t0[None], n, axis=0)
    with pytest.raises(T.VerificationFailure):
        T.batch_verify(eng, st, ts, inst, common, coms, resp)


def test_config5_share_complete_flows_w64_32768(eng):
    """Per-GPU share of config 5 through the complete flows: the wide statement Q = sum_{i<64} x_i G_i, 32,768 proofs
    proven, verified one by one and batch-verified on the device; sampled proofs equal the C oracle's."""
    from zkp_amd import toolbox as T
    n = 32768
    rng = np.random.default_rng(55)
    gens, _ = make_points(eng, rng, 64)
    eng.prepare_fixed_points(gens)
    names = [f"x_{i}" for i in range(64)]
    gnames = [f"G_{i}" for i in range(64)]
    cons = [("Q", [(names[i], gnames[i]) for i in range(64)])]
    mod = T.define_proof("w64", b"W64", names, ["Q"], gnames, cons)
    st = mod.statement
    xs = rand_scalars(rng, 64 * n)
    off = (np.arange(n + 1, dtype=np.uint64) * 64).astype(np.uint32)
    Q, sq = eng.msm_many(off, xs, np.tile(np.arange(64, dtype=np.uint32), n), gens, 1)
    assert not sq.any()
    inst = np.ascontiguousarray(Q[None])
    secrets = np.ascontiguousarray(xs.reshape(n, 64, 32))
    entropy = rng.integers(0, 256, size=(n, 32), dtype=np.uint8)
    label = b"config-5"
    t0 = T.Transcript(label).state
    ts = np.repeat(t0[None], n, axis=0)
    chal, resp, coms = T.prove_batch(eng, st, ts, secrets, inst, gens, entropy)
    cst = C.Statement.from_model(M.Statement(b"W64", names, ["Q"], gnames, cons))
    for j in (0, n - 1):
        ec, er, ek, _ = C.prove(cst, label, secrets[j], np.concatenate([inst[:, j], gens]), entropy[j].tobytes())
        assert chal[j].tobytes() == ec.tobytes() and (resp[j] == er).all() and (coms[j] == ek).all(), j
    ts = np.repeat(t0[None], n, axis=0)
    assert not T.verify_compact_batch(eng, st, ts, inst, gens, chal, resp).any()
    ts = np.repeat(t0[None], n, axis=0)
    T.batch_verify(eng, st, ts, inst, gens, coms, resp)
    resp[n - 5, 63, 0] ^= 2
    ts = np.repeat(t0[None], n, axis=0)
    with pytest.raises(T.VerificationFailure):
        T.batch_verify(eng, st, ts, inst, gens, coms, resp)


def test_config2_bench_shape_many_batches_per_call(eng):
    """The shape bench.py times since round 3: K = 8 batches of 4096 CMZ proofs (BASELINE configs[1]) proven by ONE wide call and
    verified by ONE zkp_fused_batch_verify_many call.  Every batch verifies; the coefficient vector of a batch equals that of a
    call of its own; tampering with proofs of two batches fails exactly those two; sampled proofs equal the oracle's."""
    from zkp_amd.engine import FusedStatement
    from zkp_amd import toolbox as T
    import bench
    K, n_each = 8, 4096
    n = K * n_each
    rng = np.random.default_rng(2026)
    st3 = bench.cmz_statement()
    secrets, inst, common = bench.make_instance(eng, st3, n, rng)
    mod = T.cmz_module(10)
    st = mod.statement
    fst = FusedStatement(b"CMZ cred show n=10", *st3)
    label = b"Benchmark"
    entropy = rng.integers(0, 256, size=(n, 32), dtype=np.uint8)
    eng.prepare_fixed_points(common)
    ts = np.stack([T.Transcript(label).state] * n)
    chal, resp, coms = T.prove_batch(eng, st, ts, secrets, inst, common, entropy)
    cst = C.Statement.from_model(M.cmz_statement(10))
    for j in (0, n_each - 1, n_each, 5 * n_each + 17, n - 1):
        ec, er, ek, _ = C.prove(cst, label, secrets[j], np.concatenate([inst[:, j], common]), entropy[j].tobytes())
        assert chal[j].tobytes() == ec.tobytes() and (resp[j] == er).all() and (coms[j] == ek).all(), j
    w = rng.integers(0, 256, size=(st.nc, n, 16), dtype=np.uint8)
    ts = np.stack([T.Transcript(label).state] * n)
    v, co = eng.fused_batch_verify_many(fst, K, ts, inst, common, coms, resp, w, want_coeffs=True)
    assert not v.any()
    b = 3
    sl = slice(b * n_each, (b + 1) * n_each)
    ts1 = np.stack([T.Transcript(label).state] * n_each)
    ok, co1 = T.batch_verify_coeffs(eng, st, ts1, np.ascontiguousarray(inst[:, sl]), common, coms[sl], resp[sl], np.ascontiguousarray(w[:, sl]))
    assert ok
    rows = st.ni + st.nc
    assert (co[b * st.ns:(b + 1) * st.ns] == co1[:st.ns]).all()
    assert (co[K * st.ns:].reshape(rows, n, 32)[:, sl] == co1[st.ns:].reshape(rows, n_each, 32)).all()
    assert C.batch_verify(cst, label, n_each, np.ascontiguousarray(inst[:, sl]), common, coms[sl], resp[sl], np.ascontiguousarray(w[:, sl])) == 0
    bad = resp.copy()
    bad[1 * n_each + 5, 0, 0] ^= 1
    bad[6 * n_each + 4000, 20, 31] ^= 8
    ts = np.stack([T.Transcript(label).state] * n)
    v = T.batch_verify_many(eng, st, K, ts, inst, common, coms, bad, w)
    assert v.tolist() == [0, 1, 0, 0, 0, 0, 1, 0]


@pytest.mark.parametrize("form", ["constraints", "constraints2"])
def test_config5_share_64_constraint_reading_complete_flows_32768(eng, form):
    """The other reading of BASELINE configs[4] ("64-constraint Schnorr"): 64 constraints Q_i = x_i G_i (+ y_i G_(i+1)) per proof, one GPU's share of
    2^18 proofs.  Complete flows on the device: 32,768 proofs proven (2,097,152 commitment MSMs), verified one by one and batch-verified (one MSM of
    64 + 128 N terms); sampled proofs equal the C oracle's; checksum of checksums over ALL commitments: sum_j sum_i K_ij = (sum of blindings x logs) B,
    checked through the responses -- sum_i K_ij = sum_i (s_i - c x_i) G_i per proof, i.e. the batch check with all weights 1 is a size-independent
    identity the oracle's verifier also computes; a tampered response fails the batch and verify_compact names the proof."""
    import bench
    from zkp_amd import toolbox as T
    n = 32768
    rng = np.random.default_rng(640 + len(form))
    st3 = bench.W64_FORMS[form]()
    secrets_l, points, cons = st3
    secrets, inst, common = bench.make_instance(eng, st3, n, rng)
    eng.prepare_fixed_points(common)
    names = [nm.decode() for nm, _ in points]
    st = T.Statement(b"W64")
    sv = [st.add_secret(s) for s in secrets_l]
    pv = [st.add_point(nm, c) for nm, c in points]
    for lhs, lc in cons:
        st.constrain(pv[lhs], [(sv[s], pv[p]) for s, p in lc])
    cst = C.Statement(b"W64", [s.decode() for s in secrets_l], [(nm.decode(), c) for nm, c in points],
                      [(names[l], [(secrets_l[s].decode(), names[q]) for s, q in lc]) for l, lc in cons])
    entropy = rng.integers(0, 256, size=(n, 32), dtype=np.uint8)
    label = b"config-5c"
    t0 = T.Transcript(label).state
    chal, resp, coms = T.prove_batch(eng, st, np.repeat(t0[None], n, axis=0), secrets, inst, common, entropy)
    assert coms.shape == (n, 64, 32)
    for j in (0, n // 2 + 1, n - 1):                         # allocation order: the 64 instance points, then the 64 generators
        ec, er, ek, _ = C.prove(cst, label, secrets[j], np.concatenate([inst[:, j], common]), entropy[j].tobytes())
        assert chal[j].tobytes() == ec.tobytes() and (resp[j] == er).all() and (coms[j] == ek).all(), j
    assert not T.verify_compact_batch(eng, st, np.repeat(t0[None], n, axis=0), inst, common, chal, resp).any()
    # the batch check, once with random weights and once with all weights 1 (the plain sum over all 2 M commitments: checksum of checksums)
    w = rng.integers(0, 256, size=(64, n, 16), dtype=np.uint8)
    T.batch_verify(eng, st, np.repeat(t0[None], n, axis=0), inst, common, coms, resp, w)
    ones = np.zeros((64, n, 16), np.uint8)
    ones[:, :, 0] = 1
    T.batch_verify(eng, st, np.repeat(t0[None], n, axis=0), inst, common, coms, resp, ones)
    bad = resp.copy()
    bad[n - 7, 40, 3] ^= 4
    with pytest.raises(T.VerificationFailure):
        T.batch_verify(eng, st, np.repeat(t0[None], n, axis=0), inst, common, coms, bad, ones)
    res = T.verify_compact_batch(eng, st, np.repeat(t0[None], n, axis=0), inst, common, chal, bad)
    assert res[n - 7] == 1 and res.sum() == 1
    # per-proof batchable verification on a slice (192 operands per proof: the operand-split Straus walk)
    sl = slice(1000, 1000 + 2048)
    each = T.verify_batchable_each(eng, st, np.repeat(t0[None], 2048, axis=0), np.ascontiguousarray(inst[:, sl]), common, coms[sl], bad[sl] if False else resp[sl])
    assert not each.any()
