"""Statement shapes the constraint API allows and the reference's own tests never build (missing 4), on the GPU: see
tests/statement_shapes.py for the shapes and the checks (every flow, fused device route and host-transcript route, against the oracle)."""
import numpy as np
import pytest

from oracle import cbind as C
from zkp_amd import toolbox as T
from tests.statement_shapes import (L, NEVER, SHAPES, _check_all_flows, _fresh, _materialise, _mul_base, _sc, _shape_case, w64_constraints_case)

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng():
    from zkp_amd.engine import Engine
    e = Engine(0)
    yield e
    e.close()
    T.set_fused_min_batch(32)


@pytest.mark.parametrize("n", [6, 200])
@pytest.mark.parametrize("name", SHAPES)
def test_unusual_statement_shapes_all_flows_both_routes_vs_oracle(eng, name, n):
    rng = np.random.default_rng(sum(name.encode()) + n)
    shape, secrets_int, dlog = _shape_case(name, n, rng)
    secrets, inst, common = _materialise(shape, n, secrets_int, dlog)
    _check_all_flows(eng, shape, n, secrets, inst, common, seed=n)


def test_static_lhs_false_statement_is_rejected_everywhere(eng):
    """K != k G for the STATIC left-hand side: every verifier must reject every proof of the batch (the static coefficient of K carries
    the sum over the batch of r (-c): batch_verifier.rs:186-188), on both routes and in the oracle."""
    n = 64
    rng = np.random.default_rng(5)
    shape, secrets_int, dlog = _shape_case("static_lhs_unused_on_rhs", n, rng)
    dlog["K"] = (dlog["K"] + 1) % L
    st, cst = shape.build()
    m = len(shape.secret_names)
    secrets = np.zeros((n, m, 32), np.uint8)
    for i, s in enumerate(shape.secret_names):
        secrets[:, i] = np.stack([_sc(v) for v in secrets_int[s]])
    common = _mul_base([dlog[p] for p, c in shape.points if c])
    inst = np.stack([_mul_base(list(dlog[p])) for p, c in shape.points if not c])
    entropy = rng.integers(0, 256, size=(n, 32), dtype=np.uint8)
    w = rng.integers(0, 256, size=(st.nc, n, 16), dtype=np.uint8)
    for thr in (NEVER, 0):
        T.set_fused_min_batch(thr)
        try:
            ts = _fresh(b"false", n)
            chal, resp, coms = T.prove_batch(eng, st, ts, secrets, inst, common, entropy)
            assert T.verify_compact_batch(eng, st, _fresh(b"false", n), inst, common, chal, resp).all()
            assert T.verify_batchable_each(eng, st, _fresh(b"false", n), inst, common, coms, resp, np.ascontiguousarray(w.transpose(1, 0, 2))).all()
            with pytest.raises(T.VerificationFailure):
                T.batch_verify(eng, st, _fresh(b"false", n), inst, common, coms, resp, w)
        finally:
            T.set_fused_min_batch(32)
    assert C.batch_verify(cst, b"false", n, inst, common, coms, resp, w) == 1


@pytest.mark.parametrize("tpc", [1, 2])
def test_64_constraint_statement_small_batches_all_flows_vs_oracle(eng, tpc):
    """64 constraints of 1 and of 2 terms (64 / 128 secrets, 64 instance left-hand sides, 64 - 65 static generators: 192 - 193 operands
    per proof in verify_batchable, beyond the 64-operand window split of the Straus walk), every flow, both routes, the oracle."""
    n = 48
    shape, secrets, inst, common = w64_constraints_case(n, tpc, np.random.default_rng(64 + tpc))
    _check_all_flows(eng, shape, n, secrets, inst, common, seed=tpc)
