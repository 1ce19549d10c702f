"""GPU parity tests proper: the HIP path, called through the C ABI (ctypes), against the oracle on the
same seeded inputs.  Bit-exact: every output is a canonical 32-byte ristretto255 encoding.

Large sizes use points with known discrete logs (P_j = p_j * B): the expected MSM result is then
(sum_i a_i * p_{j_i} mod l) * B -- one oracle scalar multiplication -- so the oracle side stays in
seconds while the GPU side runs at the sizes BASELINE.json names.
"""
import random

import numpy as np
import pytest

from oracle import model as M

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng():
    from zkp_amd.engine import Engine
    e = Engine(0)
    yield e
    e.close()


@pytest.fixture(scope="module")
def base_points():
    """64 points with known discrete logs, plus their encodings."""
    rng = random.Random(1234)
    logs = [rng.randrange(1, M.L) for _ in range(64)]
    logs[0] = 1
    encs = [M.ristretto_encode(M.pt_mul(k, M.BASEPOINT)) for k in logs]
    return logs, encs


def sc(x):
    return np.frombuffer((x % (1 << 256)).to_bytes(32, "little"), np.uint8)


def enc_arr(encs):
    return np.frombuffer(b"".join(encs), np.uint8).reshape(-1, 32)


def test_version_and_symbols(eng):
    assert "gfx950" in eng.version


def test_decode_check_matches_oracle(eng, base_points):
    from tests.test_host_field import BAD_ENCODINGS
    rng = random.Random(5)
    _, encs = base_points
    cases = list(encs[:20]) + [bytes(32)] + [bytes.fromhex(h) for h in BAD_ENCODINGS]
    cases += [bytes(rng.randrange(256) for _ in range(32)) for _ in range(500)]
    status, xyzt = eng.decode_check(enc_arr(cases), want_coords=True)
    for i, c in enumerate(cases):
        p = M.ristretto_decode(c)
        assert status[i] == (0 if p is not None else 1), c.hex()
        if p is not None:
            got = [int.from_bytes(xyzt[i, 32 * k:32 * k + 32].tobytes(), "little") for k in range(4)]
            assert got == [p[0], p[1], 1, p[3]]


def test_encode_many_matches_oracle(eng):
    rng = random.Random(6)
    pts = [M.IDENTITY] + [M.pt_mul(rng.randrange(1, M.L), M.BASEPOINT) for _ in range(40)]
    # projective representatives with random Z, and the 4-torsion-shifted coset members must encode identically
    rows = []
    for p in pts:
        z = rng.randrange(1, M.P)
        q = (p[0] * z % M.P, p[1] * z % M.P, p[2] * z % M.P, p[3] * z % M.P)
        rows.append(b"".join(c.to_bytes(32, "little") for c in q))
    out = eng.encode_many(np.frombuffer(b"".join(rows), np.uint8).reshape(-1, 128))
    for i, p in enumerate(pts):
        assert out[i].tobytes() == M.ristretto_encode(p)


def test_msm_many_small_against_oracle(eng, base_points):
    """Ragged CSR incl. empty MSMs, repeated points, zero / maximal / non-canonical scalars."""
    rng = random.Random(7)
    logs, encs = base_points
    npts = 12
    special = [0, 1, M.L - 1, M.L, M.L + 5, (1 << 256) - 1, (1 << 255), 0xAAAA << 240, 2, (1 << 252)]
    off, scalars, pidx = [0], [], []
    for m in range(40):
        k = [0, 1, 2, 3, 11, 12][m % 6]
        for _ in range(k):
            s = special[rng.randrange(len(special))] if rng.random() < 0.3 else rng.randrange(1 << 256)
            scalars.append(s)
            pidx.append(rng.randrange(npts))
        off.append(len(scalars))
    sc_arr = np.stack([sc(s) for s in scalars])
    for flags in (0, 1):
        out, status = eng.msm_many(off, sc_arr, pidx, enc_arr(encs[:npts]), flags)
        assert not status.any()
        for m in range(len(off) - 1):
            dlog = sum(scalars[t] * logs[pidx[t]] for t in range(off[m], off[m + 1])) % M.L
            assert out[m].tobytes() == M.ristretto_encode(M.pt_mul(dlog, M.BASEPOINT)), (m, flags)


def test_msm_many_direct_definition(eng):
    """Against the plain definition enc(sum s_i * dec(P_i)) with hash-derived points (no known logs)."""
    rng = random.Random(8)
    pts = [M.ristretto_from_uniform_bytes(bytes(rng.randrange(256) for _ in range(64))) for _ in range(5)]
    encs = [M.ristretto_encode(p) for p in pts]
    off, scalars, pidx = [0], [], []
    for k in (1, 2, 3, 5):
        for _ in range(k):
            scalars.append(rng.randrange(M.L))
            pidx.append(rng.randrange(5))
        off.append(len(scalars))
    out, status = eng.msm_many(off, np.stack([sc(s) for s in scalars]), pidx, enc_arr(encs), 1)
    assert not status.any()
    for m in range(len(off) - 1):
        ts = range(off[m], off[m + 1])
        exp = M.msm_points([scalars[t] for t in ts], [M.ristretto_decode(encs[pidx[t]]) for t in ts])
        assert out[m].tobytes() == M.ristretto_encode(exp)


def test_msm_many_invalid_point_sets_status(eng, base_points):
    _, encs = base_points
    bad = bytes.fromhex("0100000000000000000000000000000000000000000000000000000000000000")
    points = enc_arr([encs[0], bad, encs[1]])
    off = [0, 2, 4, 5]
    pidx = [0, 2, 0, 1, 1]
    out, status = eng.msm_many(off, np.stack([sc(3)] * 5), pidx, points, 0)
    assert list(status) == [0, 1, 1]
    assert out[1].tobytes() == bytes(32) and out[2].tobytes() == bytes(32)
    assert out[0].tobytes() != bytes(32)


@pytest.mark.parametrize("n", [0, 1, 2, 36, 191, 192, 193, 300, 4095, 4096, 5000, 8191, 8192, 33000, 70000])
def test_msm_optional_sizes(eng, base_points, n):
    """Covers the per-term path (n <= 192) and the Pippenger window sizes c = 7, 10, 11 (c = 16: tests/test_gpu_fullsize.py)."""
    rng = random.Random(1000 + n)
    logs, encs = base_points
    idx = [rng.randrange(64) for _ in range(n)]
    scalars = [rng.randrange(M.L) for _ in range(n)]
    if n >= 2:
        scalars[0], scalars[1] = 0, M.L - 1
    pts = enc_arr([encs[j] for j in idx]) if n else np.zeros((0, 32), np.uint8)
    sca = np.stack([sc(s) for s in scalars]) if n else np.zeros((0, 32), np.uint8)
    got = eng.msm_optional(sca, pts)
    dlog = sum(s * logs[j] for s, j in zip(scalars, idx)) % M.L
    assert got == M.ristretto_encode(M.pt_mul(dlog, M.BASEPOINT))


@pytest.mark.parametrize("n,kind", [(5000, "ones"), (5000, "weights"), (70000, "ones"), (70000, "weights"), (300000, "weights"), (70000, "two-values")])
def test_msm_optional_skewed_digits(eng, base_points, n, kind):
    """Pippenger with HUGE buckets.  A batch verification multiplies its commitments by -r, r a 128-bit weight
    (batch_verifier.rs:179-183): sign-folded, half of them carry a signed digit +1 into the window above bit 127, all in one
    bucket.  "weights": 60 % of the scalars are l - r; "ones": every scalar is 1 (one bucket holds the whole MSM: thousands of
    parts for the block-wide merge of k_pip_bucket_merge and the wavefront-wide writes of k_pip_vmap); "two-values": two scalars
    whose digits fill two buckets per window with 33 .. 63 parts or more."""
    rng = random.Random(4000 + n)
    logs, encs = base_points
    idx = [rng.randrange(64) for _ in range(n)]
    if kind == "ones":
        scalars = [1] * n
    elif kind == "weights":
        scalars = [(M.L - rng.randrange(1 << 128)) if rng.random() < 0.6 else rng.randrange(M.L) for _ in range(n)]
    else:
        a, b = rng.randrange(M.L), rng.randrange(M.L)
        scalars = [a if rng.random() < 0.97 else b for _ in range(n)]
    pts = enc_arr([encs[j] for j in idx])
    sca = np.stack([sc(s) for s in scalars])
    got = eng.msm_optional(sca, pts)
    dlog = sum(s * logs[j] for s, j in zip(scalars, idx)) % M.L
    assert got == M.ristretto_encode(M.pt_mul(dlog, M.BASEPOINT))


@pytest.mark.parametrize("n", [5, 250, 6000])
def test_msm_optional_none_on_bad_point(eng, base_points, n):
    rng = random.Random(77 + n)
    _, encs = base_points
    rows = [encs[rng.randrange(64)] for _ in range(n)]
    rows[n // 2] = bytes([2]) + bytes(31)          # s = 2: canonical, even, but not on the curve? decided by the oracle
    if M.ristretto_decode(rows[n // 2]) is not None:
        rows[n // 2] = bytes.fromhex("ffffffffffffffffffffffffffffffffffffffffffffffffffffffffffffff7f")
    assert M.ristretto_decode(rows[n // 2]) is None
    sca = np.stack([sc(rng.randrange(M.L)) for _ in range(n)])
    assert eng.msm_optional(sca, enc_arr(rows)) is None


def test_msm_optional_identity_and_cancellation(eng, base_points):
    """sum = identity must encode as 32 zero bytes (what verifier.rs:168 / batch_verifier.rs:230 test)."""
    rng = random.Random(99)
    logs, encs = base_points
    for n in (2, 40, 600):
        idx = [rng.randrange(1, 64) for _ in range(n - 1)]
        scalars = [rng.randrange(M.L) for _ in range(n - 1)]
        dlog = sum(s * logs[j] for s, j in zip(scalars, idx)) % M.L
        # last term cancels everything: (-dlog) * B  (base point has log 1 at index 0)
        idx.append(0)
        scalars.append((-dlog) % M.L)
        got = eng.msm_optional(np.stack([sc(s) for s in scalars]), enc_arr([encs[j] for j in idx]))
        assert got == bytes(32)


@pytest.mark.parametrize("flags", [0, 1])
def test_msm_many_fixed_base_tables(base_points, flags):
    """zkp_ctx_prepare_fixed_points is a pure performance hint: identical bytes with and without it, for both
    schedules, including non-canonical scalars (carry window), zero digits, and points that are NOT registered."""
    from zkp_amd.engine import Engine
    rng = random.Random(4242 + flags)
    logs, encs = base_points
    n_msm = 700                                   # > 1024 terms so the classified path is taken
    special = [0, 1, 8, M.L - 1, M.L, (1 << 256) - 1, 1 << 255, int("8" * 64, 16), int("7" * 64, 16), (1 << 253) - 1]
    off, scalars, pidx = [0], [], []
    for m in range(n_msm):
        for _ in range([2, 11, 1, 3][m % 4]):
            scalars.append(special[rng.randrange(len(special))] if rng.random() < 0.2 else rng.randrange(1 << 256))
            pidx.append(rng.randrange(40))
        off.append(len(scalars))
    sc_arr = np.stack([sc(s) for s in scalars])
    pts = enc_arr(encs[:40])
    plain = Engine(0)
    ref_out, ref_st = plain.msm_many(off, sc_arr, pidx, pts, flags)
    plain.close()
    hot = Engine(0)
    bad = bytes.fromhex("0100000000000000000000000000000000000000000000000000000000000000")
    hot.prepare_fixed_points(enc_arr(encs[:12] + [bad] + [bytes(32)]))      # 12 of the 40 points + junk + identity
    out, st = hot.msm_many(off, sc_arr, pidx, pts, flags)
    assert not st.any() and (out == ref_out).all()
    for m in range(0, n_msm, 37):                                          # and against the oracle definition
        dlog = sum(scalars[t] * logs[pidx[t]] for t in range(off[m], off[m + 1])) % M.L
        assert out[m].tobytes() == M.ristretto_encode(M.pt_mul(dlog, M.BASEPOINT))
    # more registrations than slots: least recently used tables are replaced, results stay exact
    rp = [M.ristretto_encode(M.pt_mul(rng.randrange(1, M.L), M.BASEPOINT)) for _ in range(70)]
    hot.prepare_fixed_points(enc_arr(rp))
    hot.prepare_fixed_points(enc_arr(encs[20:40]))
    out2, st2 = hot.msm_many(off, sc_arr, pidx, pts, flags)
    assert not st2.any() and (out2 == ref_out).all()
    # an invalid point in the table still flags exactly the MSMs that reference it
    pts_bad = pts.copy()
    pts_bad[5] = np.frombuffer(bad, np.uint8)
    out3, st3 = hot.msm_many(off, sc_arr, pidx, pts_bad, flags)
    for m in range(n_msm):
        uses = any(pidx[t] == 5 for t in range(off[m], off[m + 1]))
        assert st3[m] == int(uses)
        if not uses:
            assert (out3[m] == ref_out[m]).all()
    hot.close()


def test_quad_cooperative_point_ops(base_points):
    """4-lane cooperative doubling / addition / mixed addition (quad.h) against the oracle, incl. identity operands,
    P = Q (the unified formulas must double) and P = -Q (sum = identity).  The self-test kernel lives in the test-hook
    build of the library only (-DZKP_BUILD_TEST_HOOKS); the shipped library does not carry it."""
    from zkp_amd.engine import Engine, ZkpError
    with pytest.raises((ZkpError, AttributeError)):
        plain = Engine(0)
        try:
            plain.debug_quad_selftest(np.zeros((1, 64), np.uint8))
        finally:
            plain.close()
    eng = Engine(0, test_hooks=True)
    assert "+test-hooks" in eng.version
    rng = random.Random(31337)
    _, encs = base_points
    ident = bytes(32)
    negs = [M.ristretto_encode(M.pt_neg(M.ristretto_decode(e))) for e in encs[:4]]
    pairs = [(ident, ident), (encs[0], ident), (ident, encs[1]), (encs[2], encs[2]), (encs[3], negs[3])]
    pairs += [(encs[rng.randrange(64)], encs[rng.randrange(64)]) for _ in range(300)]
    arr = np.frombuffer(b"".join(p + q for p, q in pairs), np.uint8).reshape(-1, 64)
    out = eng.debug_quad_selftest(arr)
    eng.close()
    for i, (pe, qe) in enumerate(pairs):
        P, Q = M.ristretto_decode(pe), M.ristretto_decode(qe)
        exp = [M.pt_double(P), M.pt_add(P, Q), M.pt_add(P, Q), M.pt_add(P, M.pt_neg(Q))]
        for k in range(4):
            assert out[i, k].tobytes() == M.ristretto_encode(exp[k]), (i, k)


def test_row_cooperative_point_ops(base_points):
    """one-limb-per-lane doubling / cached addition (rowfe.h: the Horner tail of every Pippenger run) against the oracle: 2P, P + Q and the
    chain 2^11 P + Q, incl. identity operands, P = Q and P = -Q.  Test-hook build only, like the quad self-test."""
    from zkp_amd.engine import Engine
    eng = Engine(0, test_hooks=True)
    rng = random.Random(4242)
    _, encs = base_points
    ident = bytes(32)
    negs = [M.ristretto_encode(M.pt_neg(M.ristretto_decode(e))) for e in encs[:4]]
    pairs = [(ident, ident), (encs[0], ident), (ident, encs[1]), (encs[2], encs[2]), (encs[3], negs[3])]
    pairs += [(encs[rng.randrange(64)], encs[rng.randrange(64)]) for _ in range(200)]
    arr = np.frombuffer(b"".join(p + q for p, q in pairs), np.uint8).reshape(-1, 64)
    out = eng.debug_row_selftest(arr)
    eng.close()
    for i, (pe, qe) in enumerate(pairs):
        P, Q = M.ristretto_decode(pe), M.ristretto_decode(qe)
        chain = P
        for _ in range(11):
            chain = M.pt_double(chain)
        exp = [M.pt_double(P), M.pt_add(P, Q), M.pt_add(chain, Q)]
        for k in range(3):
            assert out[i, k].tobytes() == M.ristretto_encode(exp[k]), (i, k)


@pytest.mark.parametrize("flags", [0, 1])
def test_batched_encoder_equals_per_output_encoder(base_points, flags):
    """ZKP_OPT_BATCH_ENCODE_MIN: outputs encoded as 2 * sum (s/2) P with one shared inversion (ristretto_dc_*, the identity
    behind dalek's double_and_compress_batch) must be the bytes of the per-output encoder -- including empty MSMs and MSMs
    that cancel to the identity (x = 0 in the batch: general encoder), non-canonical scalars, invalid points (zeroed output,
    status 1), with block sizes that leave the last product-tree block partly empty."""
    from zkp_amd.engine import Engine, ZKP_OPT_BATCH_ENCODE_MIN
    rng = random.Random(99 + flags)
    logs, encs = base_points
    special = [0, 1, 2, M.L - 1, M.L, M.L + 1, (1 << 256) - 1, 1 << 255, (1 << 252) + 7]
    for n_msm in (300, 513, 1400):
        off, scalars, pidx = [0], [], []
        for m in range(n_msm):
            kind = m % 9
            if kind == 0:
                pass                                                    # empty MSM -> identity
            elif kind == 1:
                p = rng.randrange(40); s = rng.randrange(1, M.L)        # s P + (l - s) P = identity
                scalars += [s, M.L - s]; pidx += [p, p]
            elif kind == 2:
                scalars += [0, 0, 0]; pidx += [rng.randrange(40) for _ in range(3)]
            else:
                for _ in range([1, 2, 11, 5][m % 4]):
                    scalars.append(special[rng.randrange(len(special))] if rng.random() < 0.25 else rng.randrange(1 << 256))
                    pidx.append(rng.randrange(40))
            off.append(len(scalars))
        while len(scalars) < 1100:                                      # the classified path needs >= 1024 terms
            scalars.append(rng.randrange(1 << 256)); pidx.append(rng.randrange(40)); off[-1] += 1
        sc_arr = np.stack([sc(s) for s in scalars])
        pts = enc_arr(encs[:40])
        pts_bad = pts.copy()
        pts_bad[7] = np.frombuffer(bytes.fromhex("01" + "00" * 31), np.uint8)
        results = {}
        for mode, thr in (("per-output", 0xFFFFFFFFFFFFFFFF), ("batched", 0)):
            e = Engine(0)
            e.set_option(ZKP_OPT_BATCH_ENCODE_MIN, thr)
            e.prepare_fixed_points(enc_arr(encs[:9]))
            results[mode] = [e.msm_many(off, sc_arr, pidx, pts, flags), e.msm_many(off, sc_arr, pidx, pts_bad, flags)]
            e.close()
        for (o1, s1), (o2, s2) in zip(results["per-output"], results["batched"]):
            assert (s1 == s2).all() and (o1 == o2).all()
        out, st = results["batched"][0]
        assert not st.any()
        for m in range(0, n_msm, 9):
            assert out[m].tobytes() == bytes(32) and out[m + 1].tobytes() == bytes(32) if m + 1 < n_msm else True
        for m in range(3, n_msm, 41):                                   # and against the big-integer definition
            dlog = sum(scalars[t] * logs[pidx[t]] for t in range(off[m], off[m + 1])) % M.L
            assert out[m].tobytes() == M.ristretto_encode(M.pt_mul(dlog, M.BASEPOINT))
        assert results["batched"][1][1].any()
