"""The device-pointer (_dev) entry points of include/zkp_mi355x.h -- asynchronous, operands resident in HBM, what
bench.py measures -- against the host-pointer entry points and the oracle on the same inputs, byte for byte.
torch is used for nothing but device buffers."""
import numpy as np
import pytest

from oracle import cbind as C
from zkp_amd import toolbox as T
from tests.test_gpu_toolbox import BASEPOINT, _cmz_batch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng():
    from zkp_amd.engine import Engine
    e = Engine(0)
    yield e
    e.close()


def _torch():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("torch cannot see the GPU in this process (its HIP runtime must initialise before libzkp_mi355x.so: run with -m gpu)")
    return torch


def _dev(a):
    torch = _torch()
    return torch.from_numpy(np.ascontiguousarray(a)).to("cuda:0")


def _cmz_fused_statement():
    import bench
    from zkp_amd.engine import FusedStatement
    return FusedStatement(b"CMZ cred show n=10", *bench.cmz_statement())


@pytest.mark.parametrize("flags", [0, 1])
def test_msm_many_dev_equals_host_entry_and_oracle(eng, flags):
    torch = _torch()
    rng = np.random.default_rng(3)
    n = 300
    mod, secrets, inst, common = _cmz_batch(n, 5)
    import bench
    off, pidx, n_pts = bench.cmz_shape(n)
    pts = np.concatenate([common[:11], np.stack([inst[10], inst[11]], axis=1).reshape(-1, 32)])     # X_1..X_10, A, then P_j, Q_j
    assert len(pts) == n_pts
    sc = rng.integers(0, 256, size=(31 * n, 32), dtype=np.uint8)
    sc[:, 31] &= 0x0f
    eng.prepare_fixed_points(pts[:11])
    want, wst = eng.msm_many(off, sc, pidx, pts, flags)
    d_out = torch.zeros((11 * n, 32), dtype=torch.uint8, device="cuda:0")
    d_st = torch.ones(11 * n, dtype=torch.uint8, device="cuda:0")
    d_off, d_sc, d_pidx, d_pts = _dev(off.view(np.int32)), _dev(sc), _dev(pidx.view(np.int32)), _dev(pts)
    torch.cuda.synchronize()
    eng.msm_many_dev(11 * n, d_off.data_ptr(), d_sc.data_ptr(), d_pidx.data_ptr(), d_pts.data_ptr(), n_pts, 31 * n, flags,
                     d_out.data_ptr(), d_st.data_ptr())
    eng.synchronize()
    assert not wst.any() and not d_st.cpu().numpy().any()
    assert (d_out.cpu().numpy() == want).all()
    exp, est = C.msm_many(off, sc, pidx, pts, flags)
    assert (want == exp).all()


@pytest.mark.parametrize("n", [40, 1500])
def test_fused_dev_flows_equal_host_pointer_flows(eng, n):
    """zkp_fused_prove_dev / _verify_compact_dev / _batch_verify_dev against zkp_fused_prove / ... (through the toolbox)."""
    torch = _torch()
    mod, secrets, inst, common = _cmz_batch(n, 17)
    st = mod.statement
    fst = _cmz_fused_statement()
    rng = np.random.default_rng(n)
    entropy = rng.integers(0, 256, size=(n, 32), dtype=np.uint8)
    w = rng.integers(0, 256, size=(11, n, 16), dtype=np.uint8)
    t0 = T.Transcript(b"dev-entry").state
    pos = int(t0[200]) | int(t0[201]) << 8 | int(t0[202]) << 16
    ts0 = np.stack([t0] * n)
    # reference results through the host-pointer flows
    T.set_fused_min_batch(0)
    try:
        ts = ts0.copy()
        chal, resp, coms = T.prove_batch(eng, st, ts, secrets, inst, common, entropy)
        ts_after_prove = ts.copy()
        ts = ts0.copy()
        ok, coeffs = T.batch_verify_coeffs(eng, st, ts, inst, common, coms, resp, w)
        assert ok
    finally:
        T.set_fused_min_batch(32)
    eng.prepare_fixed_points(common)
    table = np.concatenate([common, inst.reshape(-1, 32)])
    d_ts, d_sec, d_tbl, d_ent = _dev(ts0), _dev(secrets), _dev(table), _dev(entropy)
    z = lambda *s: torch.zeros(s, dtype=torch.uint8, device="cuda:0")
    d_chal, d_resp, d_coms, d_st = z(n, 32), z(n, 21, 32), z(n, 11, 32), z(11 * n)
    torch.cuda.synchronize()
    eng.fused_prove_dev(fst, n, pos, d_ts.data_ptr(), d_sec.data_ptr(), d_tbl.data_ptr(), d_ent.data_ptr(), d_chal.data_ptr(),
                        d_resp.data_ptr(), d_coms.data_ptr(), d_st.data_ptr())
    eng.synchronize()
    assert not d_st.cpu().numpy().any()
    assert (d_chal.cpu().numpy() == chal).all() and (d_resp.cpu().numpy() == resp).all() and (d_coms.cpu().numpy() == coms).all()
    assert (d_ts.cpu().numpy()[:, :203] == ts_after_prove[:, :203]).all()
    # verify_compact on the device-resident proofs; then with one response corrupted
    d_ts2, d_res = _dev(ts0), z(n) + 1
    eng.fused_verify_compact_dev(fst, n, pos, d_ts2.data_ptr(), d_tbl.data_ptr(), d_chal.data_ptr(), d_resp.data_ptr(), d_res.data_ptr())
    eng.synchronize()
    assert not d_res.cpu().numpy().any()
    d_bad = d_resp.clone()
    d_bad[n // 3, 2, 0] ^= 1
    d_ts2 = _dev(ts0)
    eng.fused_verify_compact_dev(fst, n, pos, d_ts2.data_ptr(), d_tbl.data_ptr(), d_chal.data_ptr(), d_bad.data_ptr(), d_res.data_ptr())
    eng.synchronize()
    r = d_res.cpu().numpy()
    assert r[n // 3] == 1 and r.sum() == 1
    # batch verification: operand array with static points and instance rows filled in
    d_pts = z(12 + 24 * n, 32)
    d_pts[: 12 + 13 * n] = d_tbl
    d_ts3, d_w, d_out = _dev(ts0), _dev(w), z(32) + 1
    d_bst = torch.ones(2, dtype=torch.int32, device="cuda:0")
    torch.cuda.synchronize()
    eng.fused_batch_verify_dev(fst, n, pos, d_ts3.data_ptr(), d_pts.data_ptr(), d_coms.data_ptr(), d_resp.data_ptr(), d_w.data_ptr(),
                               d_out.data_ptr(), d_bst.data_ptr())
    eng.synchronize()
    assert not d_out.cpu().numpy().any() and not d_bst.cpu().numpy().any()
    d_ts3 = _dev(ts0)
    eng.fused_batch_verify_dev(fst, n, pos, d_ts3.data_ptr(), d_pts.data_ptr(), d_coms.data_ptr(), d_bad.data_ptr(), d_w.data_ptr(),
                               d_out.data_ptr(), d_bst.data_ptr())
    eng.synchronize()
    assert d_out.cpu().numpy().any() and not d_bst.cpu().numpy().any()
    # an identity commitment is refused by the transcript protocol: second status word
    d_zc = d_coms.clone()
    d_zc[n // 2, 4] = 0
    d_ts3 = _dev(ts0)
    eng.fused_batch_verify_dev(fst, n, pos, d_ts3.data_ptr(), d_pts.data_ptr(), d_zc.data_ptr(), d_resp.data_ptr(), d_w.data_ptr(),
                               d_out.data_ptr(), d_bst.data_ptr())
    eng.synchronize()
    assert d_bst.cpu().numpy()[1] == 1
