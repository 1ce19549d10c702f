"""(helper module of tests/test_gpu_statement_shapes.py and tests/test_host_backend.py)
Statement shapes the constraint API allows and the reference's own tests never build (missing 4): a STATIC point as a
constraint's left-hand side (batch_verifier.rs:156-160, 186-188: its r * (-c) is summed over the batch into static_coeffs), one point as
left-hand side of two constraints, a point that is left-hand side of one constraint and right-hand side of another, a repeated term, and
statements of 64 constraints (the second reading of BASELINE configs[4]).  Each through prove / verify_compact / verify_batchable (per
proof) / batch_verify / batch_verify_many, on the fused device route and the host-transcript route, against the oracle's restatement of
prover.rs / verifier.rs / batch_verifier.rs -- proofs byte for byte, the batch verifier's operand scalars (static_coeffs first) scalar for
scalar."""
import numpy as np
import pytest

from oracle import cbind as C
from oracle import model as M
from zkp_amd import toolbox as T
from tests.test_gpu_toolbox import BASEPOINT

NEVER = 0xFFFFFFFF
L = M.L


def _sc(v: int) -> np.ndarray:
    return np.frombuffer((v % L).to_bytes(32, "little"), np.uint8)


def _rand_scalars(rng, k):
    s = rng.integers(0, 256, size=(k, 32), dtype=np.uint8)
    s[:, 31] &= 0x0f
    return s


def _ints(a):
    return [int.from_bytes(x.tobytes(), "little") for x in a.reshape(-1, 32)]


def _mul_base(scalars_int):
    """[k] ints -> [k][32] encodings of s * B"""
    base = np.frombuffer(BASEPOINT, np.uint8).reshape(1, 32)
    k = len(scalars_int)
    sc = np.stack([_sc(v) for v in scalars_int])
    pts, st = C.msm_many(np.arange(k + 1, dtype=np.uint32), sc, np.zeros(k, np.uint32), base, 0)
    assert not st.any()
    return pts


class Shape:
    """points: [(name, is_common)] in allocation order; cons: [(lhs name, [(secret name, point name)])];
    every point is d * B for a known discrete log d (common: one int, instance: [n] ints), so any left-hand side can be computed"""

    def __init__(self, label, secret_names, points, cons):
        self.label, self.secret_names, self.points, self.cons = label, secret_names, points, cons

    def build(self):
        st = T.Statement(self.label)
        sv = {s: st.add_secret(s.encode()) for s in self.secret_names}
        pv = {p: st.add_point(p.encode(), c) for p, c in self.points}
        for lhs, lc in self.cons:
            st.constrain(pv[lhs], [(sv[s], pv[p]) for s, p in lc])
        cst = C.Statement(self.label, self.secret_names, self.points, self.cons)
        return st, cst


def _materialise(shape, n, secrets_int, dlog):
    """secrets_int: name -> [n] ints; dlog: point name -> int (common) or [n] ints (instance), left-hand sides included"""
    m = len(shape.secret_names)
    secrets = np.zeros((n, m, 32), np.uint8)
    for i, s in enumerate(shape.secret_names):
        secrets[:, i] = np.stack([_sc(v) for v in secrets_int[s]])
    inst_names = [p for p, c in shape.points if not c]
    common_names = [p for p, c in shape.points if c]
    common = _mul_base([dlog[p] for p in common_names]) if common_names else np.zeros((0, 32), np.uint8)
    inst = np.stack([_mul_base(list(dlog[p])) for p in inst_names]) if inst_names else np.zeros((0, n, 32), np.uint8)
    # the statement is true: every constraint holds in the exponent
    for lhs, lc in shape.cons:
        for j in (0, n - 1):
            want = sum(secrets_int[s][j] * (dlog[p] if isinstance(dlog[p], int) else dlog[p][j]) for s, p in lc) % L
            have = dlog[lhs] if isinstance(dlog[lhs], int) else dlog[lhs][j]
            assert want == have % L, (lhs, j)
    return secrets, np.ascontiguousarray(inst), np.ascontiguousarray(common)


def _fresh(label, n):
    return np.stack([T.Transcript(label).state] * n)


def _check_all_flows(eng, shape, n, secrets, inst, common, seed, oracle_samples=(0, -1)):
    st, cst = shape.build()
    rng = np.random.default_rng(seed)
    entropy = rng.integers(0, 256, size=(n, 32), dtype=np.uint8)
    nc = st.nc
    w = rng.integers(0, 256, size=(nc, n, 16), dtype=np.uint8)
    tl = b"shapes"
    out = {}
    for route, thr in (("host", NEVER), ("fused", 0)):
        T.set_fused_min_batch(thr)
        try:
            ts = _fresh(tl, n)
            chal, resp, coms = T.prove_batch(eng, st, ts, secrets, inst, common, entropy)
            ts2 = _fresh(tl, n)
            res = T.verify_compact_batch(eng, st, ts2, inst, common, chal, resp)
            ts3 = _fresh(tl, n)
            ok, coeffs = T.batch_verify_coeffs(eng, st, ts3, inst, common, coms, resp, w)
            ts4 = _fresh(tl, n)
            each = T.verify_batchable_each(eng, st, ts4, inst, common, coms, resp, np.ascontiguousarray(w.transpose(1, 0, 2)))
            bad = resp.copy()
            bad[n // 2, 0, 0] ^= 1
            ts5 = _fresh(tl, n)
            res_bad = T.verify_compact_batch(eng, st, ts5, inst, common, chal, bad)
            ts6 = _fresh(tl, n)
            each_bad = T.verify_batchable_each(eng, st, ts6, inst, common, coms, bad, np.ascontiguousarray(w.transpose(1, 0, 2)))
            verd = verd_bad = None
            if n % 2 == 0 and n >= 2:
                ts7 = _fresh(tl, n)
                verd = T.batch_verify_many(eng, st, 2, ts7, inst, common, coms, resp, w)
                ts8 = _fresh(tl, n)
                verd_bad = T.batch_verify_many(eng, st, 2, ts8, inst, common, coms, bad, w)
            ts9 = _fresh(tl, n)
            T.batch_verify(eng, st, ts9, inst, common, coms, resp, w)                       # raises on failure
            ts10 = _fresh(tl, n)
            with pytest.raises(T.VerificationFailure):
                T.batch_verify(eng, st, ts10, inst, common, coms, bad, w)
        finally:
            T.set_fused_min_batch(32)
        assert ok and not res.any() and not each.any(), route
        assert res_bad[n // 2] == 1 and res_bad.sum() == 1 and each_bad[n // 2] == 1 and each_bad.sum() == 1, route
        if verd is not None:
            assert verd.tolist() == [0, 0] and verd_bad.tolist() == [0, 1], route
        out[route] = (chal, resp, coms, ts[:, :203], ts2[:, :203], coeffs, ts3[:, :203], ts4[:, :203])
    for a, b in zip(out["host"], out["fused"]):
        assert a.shape == b.shape and (a == b).all()
    chal, resp, coms = out["fused"][:3]
    # the oracle: proofs byte for byte (its point list is the allocation order) ...
    names = [p for p, _ in shape.points]
    inst_names = [p for p, c in shape.points if not c]
    common_names = [p for p, c in shape.points if c]
    for j in oracle_samples:
        j = j % n
        pts_j = np.stack([common[common_names.index(p)] if c else inst[inst_names.index(p)][j] for p, c in shape.points])
        ec, er, ek, _ = C.prove(cst, tl, secrets[j], pts_j, entropy[j].tobytes())
        assert chal[j].tobytes() == ec.tobytes() and (resp[j] == er).all() and (coms[j] == ek).all(), j
        assert C.verify_compact(cst, tl, pts_j, chal[j], resp[j]) == 0
    # ... and the batch verifier's operand list: static_coeffs (batch_verifier.rs:186-188, 198) first, then the row-major matrix
    if n <= 512:
        rc, osc, _ = C.batch_verify(cst, tl, n, inst, common, coms, resp, w, want_msm_inputs=True)
        assert rc == 0 and (osc == out["fused"][5]).all()
        assert C.batch_verify(cst, tl, n, inst, common, coms, resp, w) == 0
    return out["fused"]


def _shape_case(name, n, rng):
    """-> (Shape, secrets_int, dlog)"""
    r = lambda: int.from_bytes(rng.bytes(32), "little") % L
    rn = lambda: [r() for _ in range(n)]
    g, h = r(), r()
    if name == "static_lhs_unused_on_rhs":
        # K = k G with K, G static: the same statement in every proof of the batch (an issuer key); K is on no right-hand side
        k, x = r(), rn()
        shape = Shape(b"static lhs", ["k", "x"], [("G", True), ("H", True), ("K", True), ("A", False)], [("K", [("k", "G")]), ("A", [("x", "H")])])
        return shape, {"k": [k] * n, "x": x}, {"G": g, "H": h, "K": k * g % L, "A": [xi * h % L for xi in x]}
    if name == "static_lhs_used_on_rhs":
        # ... and the static left-hand side is the base of the per-proof constraint; statics allocated AFTER the instance point
        k, x = r(), rn()
        shape = Shape(b"static lhs 2", ["x", "k"], [("A", False), ("G", True), ("K", True)], [("A", [("x", "K")]), ("K", [("k", "G")])])
        return shape, {"k": [k] * n, "x": x}, {"G": g, "K": k * g % L, "A": [xi * k * g % L for xi in x]}
    if name == "static_lhs_twice":
        # one static point as left-hand side of two constraints: K = k G = k' H
        k = r()
        k2 = k * g * pow(h, -1, L) % L
        x = rn()
        shape = Shape(b"static lhs twice", ["k", "k2", "x"], [("G", True), ("H", True), ("K", True), ("A", False)],
                      [("K", [("k", "G")]), ("A", [("x", "G"), ("k", "H")]), ("K", [("k2", "H")])])
        return shape, {"k": [k] * n, "k2": [k2] * n, "x": x}, {"G": g, "H": h, "K": k * g % L, "A": [(xi * g + k * h) % L for xi in x]}
    if name == "instance_lhs_twice":
        # A = x G and A = y H for one instance point A
        x = rn()
        y = [xi * g * pow(h, -1, L) % L for xi in x]
        shape = Shape(b"lhs twice", ["x", "y"], [("A", False), ("G", True), ("H", True)], [("A", [("x", "G")]), ("A", [("y", "H")])])
        return shape, {"x": x, "y": y}, {"G": g, "H": h, "A": [xi * g % L for xi in x]}
    if name == "lhs_is_rhs_elsewhere":
        # A = x G ; B = y A + x G   (A: left-hand side of the first, right-hand side of the second)
        x, y = rn(), rn()
        a = [xi * g % L for xi in x]
        shape = Shape(b"lhs and rhs", ["x", "y"], [("G", True), ("A", False), ("B", False)], [("A", [("x", "G")]), ("B", [("y", "A"), ("x", "G")])])
        return shape, {"x": x, "y": y}, {"G": g, "A": a, "B": [(yi * ai + xi * g) % L for xi, yi, ai in zip(x, y, a)]}
    if name == "repeated_term":
        # the same (scalar, point) term twice in one constraint, and a constraint whose terms all name one point: A = x G + x G + y G
        x, y = rn(), rn()
        shape = Shape(b"repeated term", ["x", "y"], [("A", False), ("G", True)], [("A", [("x", "G"), ("x", "G"), ("y", "G")])])
        return shape, {"x": x, "y": y}, {"G": g, "A": [(2 * xi + yi) * g % L for xi, yi in zip(x, y)]}
    if name == "only_static_points":
        # no instance point at all: every proof of the batch proves the same two static relations (ni = 0: the matrix has commitment rows only)
        k, k2 = r(), r()
        shape = Shape(b"only statics", ["k", "k2"], [("G", True), ("H", True), ("K", True), ("K2", True)], [("K", [("k", "G")]), ("K2", [("k2", "H"), ("k", "K")])])
        return shape, {"k": [k] * n, "k2": [k2] * n}, {"G": g, "H": h, "K": k * g % L, "K2": (k2 * h + k * k * g) % L}
    raise KeyError(name)


SHAPES = ["static_lhs_unused_on_rhs", "static_lhs_used_on_rhs", "static_lhs_twice", "instance_lhs_twice", "lhs_is_rhs_elsewhere", "repeated_term", "only_static_points"]


def _w64_constraints_shape(terms_per_constraint):
    """the second reading of BASELINE configs[4] ("64-constraint Schnorr"): 64 constraints Q_i = x_i G_i (+ y_i H), generators static"""
    xs = ["x_%d" % i for i in range(64)]
    ys = ["y_%d" % i for i in range(64)] if terms_per_constraint == 2 else []
    points = [("Q_%d" % i, False) for i in range(64)] + [("G_%d" % i, True) for i in range(64)] + ([("H", True)] if ys else [])
    cons = [("Q_%d" % i, [("x_%d" % i, "G_%d" % i)] + ([("y_%d" % i, "H")] if ys else [])) for i in range(64)]
    return Shape(b"W64 constraints", xs + ys, points, cons)


def w64_constraints_case(n, tpc, rng):
    """-> (shape, secrets, inst, common) of n proofs of the 64-constraint statement with tpc terms per constraint"""
    shape = _w64_constraints_shape(tpc)
    r = lambda: int.from_bytes(rng.bytes(32), "little") % L
    g = [r() for _ in range(64)]
    h = r()
    secrets_int = {"x_%d" % i: [r() for _ in range(n)] for i in range(64)}
    dlog = {"G_%d" % i: g[i] for i in range(64)}
    if tpc == 2:
        secrets_int.update({"y_%d" % i: [r() for _ in range(n)] for i in range(64)})
        dlog["H"] = h
    for i in range(64):
        dlog["Q_%d" % i] = [(secrets_int["x_%d" % i][j] * g[i] + (secrets_int["y_%d" % i][j] * h if tpc == 2 else 0)) % L for j in range(n)]
    return (shape,) + _materialise(shape, n, secrets_int, dlog)
