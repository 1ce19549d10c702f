"""The device-pointer (_dev) entry points of include/zkp_mi355x.h -- asynchronous, operands resident in HBM, what
bench.py measures -- against the host-pointer entry points and the oracle on the same inputs, byte for byte.
torch is used for nothing but device buffers."""
import ctypes

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


@pytest.mark.parametrize("n,wide", [(40, False), (1500, False), (40, True), (1500, True), (40, "fuse"), (1500, "fuse"), (40, "interp"), (1500, "interp"),
                                    (1500, "fuse+interp"), (40, "latency"), (1500, "latency"), (4096, "latency"), (1500, "fuse+wide"), (40, "wide+interp")])
def test_fused_dev_flows_equal_host_pointer_flows(eng, n, wide):
    """zkp_fused_prove_dev / _verify_compact_dev / _batch_verify_dev / _verify_batchable_dev against zkp_fused_prove / ... (through the toolbox).
    wide: with the variants the _dev entry points pick for calls that fill the chip on their own (one transcript lane per
    proof, constant-time ladder for single-use points), forced here at a small batch size; "fuse": with
    ZKP_OPT_FUSE_TABLES_TRANSCRIPT; "interp": ZKP_OPT_TRANSCRIPT_STEPS = 0, the word-operation interpreter of rounds 2 - 5 instead of assemble + chain
    (the default, which every other case runs); "latency": ZKP_OPT_DEV_OVERLAP = 2, the synchronous calls' schedule on device buffers (side stream for the
    point phases incl. the batch MSM's decompressions, chain wavefronts that own their SIMD)."""
    if "fuse" in str(wide):
        eng.set_option(8, 1)            # ZKP_OPT_FUSE_TABLES_TRANSCRIPT: program A in the comb tables' launch
    if "interp" in str(wide):
        eng.set_option(15, 0)           # ZKP_OPT_TRANSCRIPT_STEPS
    if wide == "latency":
        eng.set_option(5, 2)            # ZKP_OPT_DEV_OVERLAP
    elif wide is True or "wide" in str(wide):      # ("fuse+wide": the one-lane chain inside the comb tables' launch; "wide+interp": the one-lane interpreter)
        eng.set_option(4, 1)            # ZKP_OPT_TRANSCRIPT_LANES
        eng.set_option(3, 0)            # ZKP_OPT_CT_SINGLE_USE_TABLES
    try:
        _fused_dev_flows(eng, n)
    finally:
        eng.set_option(4, 2**64 - 1)
        eng.set_option(3, 2**64 - 1)
        eng.set_option(8, 2**64 - 1)
        eng.set_option(15, 1)
        eng.set_option(5, 0)


def _fused_dev_flows(eng, n):
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
    # verify_batchable per proof on device buffers (table = common || instance rows || commitments [n][11]; weights [n][11][16]):
    # the verdicts of the host-pointer flow, for valid proofs, a corrupted response, an identity commitment
    w_each = np.ascontiguousarray(w.transpose(1, 0, 2))
    for d_r, d_c, bad_at in ((d_resp, d_coms, None), (d_bad, d_coms, n // 3), (d_resp, d_zc, n // 2)):
        want = T.verify_batchable_each(eng, st, ts0.copy(), inst, common, d_c.cpu().numpy(), d_r.cpu().numpy(), w_each)
        d_tbl_each = torch.cat([d_tbl, d_c.reshape(-1, 32)])
        d_ts4, d_we, d_res4 = _dev(ts0), _dev(w_each), z(n) + 7
        torch.cuda.synchronize()
        eng.fused_verify_batchable_dev(fst, n, pos, d_ts4.data_ptr(), d_tbl_each.data_ptr(), d_r.data_ptr(), d_we.data_ptr(), d_res4.data_ptr())
        eng.synchronize()
        got = d_res4.cpu().numpy()
        assert (got == want).all() and (got.sum() == 0 if bad_at is None else (got[bad_at] == 1 and got.sum() == 1))


def test_hip_graph_replay_equals_direct_calls(eng):
    """zkp_ctx_capture_begin/_end + zkp_graph_launch: the recorded chain (fresh-transcript copies, fused prove, fused batch
    verification) replayed on NEW inputs placed in the same device buffers gives the bytes of the direct calls; capturing a
    call whose plan was never compiled is refused instead of allocating inside the capture."""
    torch = _torch()
    from zkp_amd.engine import Engine, ZkpError
    n = 96
    mod, secrets, inst, common = _cmz_batch(n, 8)
    fst = _cmz_fused_statement()
    e = Engine(0)
    stream = torch.cuda.Stream()
    e.set_stream(stream.cuda_stream)
    e.prepare_fixed_points(common)
    t0 = T.Transcript(b"graph").state
    pos = int(t0[200]) | int(t0[201]) << 8 | int(t0[202]) << 16
    rng = np.random.default_rng(4)
    d_ts0 = _dev(np.stack([t0] * n))
    d_ent, d_w = _dev(rng.integers(0, 256, size=(n, 32), dtype=np.uint8)), _dev(rng.integers(0, 256, size=(11, n, 16), dtype=np.uint8))
    d_sec, d_tbl = _dev(secrets), _dev(np.concatenate([common, inst.reshape(-1, 32)]))
    z8 = lambda *s: torch.zeros(s, dtype=torch.uint8, device="cuda:0")
    ts, ts2, chal, resp, coms, st = z8(n, 208), z8(n, 208), z8(n, 32), z8(n, 21, 32), z8(n, 11, 32), z8(11 * n)
    pts, out, bst = z8(12 + 24 * n, 32), z8(32), torch.ones(2, dtype=torch.int32, device="cuda:0")

    def chain():
        with torch.cuda.stream(stream):
            ts.copy_(d_ts0, non_blocking=True)
            ts2.copy_(d_ts0, non_blocking=True)
            pts[: 12 + 13 * n].copy_(d_tbl, non_blocking=True)
            e.fused_prove_dev(fst, n, pos, ts.data_ptr(), d_sec.data_ptr(), d_tbl.data_ptr(), d_ent.data_ptr(), chal.data_ptr(), resp.data_ptr(),
                              coms.data_ptr(), st.data_ptr())
            e.fused_batch_verify_dev(fst, n, pos, ts2.data_ptr(), pts.data_ptr(), coms.data_ptr(), resp.data_ptr(), d_w.data_ptr(), out.data_ptr(), bst.data_ptr())

    torch.cuda.synchronize()
    e.capture_begin()                       # nothing ran yet on this context: the plan does not exist -> refused (no allocation
    with pytest.raises(ZkpError):           # or synchronous copy may happen inside a capture)
        chain()
    e.capture_end().close()                 # what was recorded before the refusal (three copies) is discarded
    chain()                                 # direct: compiles the plans, sizes the workspace
    e.synchronize(); torch.cuda.synchronize()
    want = [x.clone() for x in (chal, resp, coms, ts, ts2)]
    assert int(bst.abs().sum().item()) == 0 and not bool(out.any().item())
    e.capture_begin()
    chain()
    g = e.capture_end()
    for x in (chal, resp, coms, ts, ts2):
        x.zero_()
    g.launch()
    e.synchronize(); torch.cuda.synchronize()
    assert all(bool((a == b).all().item()) for a, b in zip(want, (chal, resp, coms, ts, ts2)))
    # new prover randomness in the SAME buffer: the replay must follow the data, not the recording
    entropy2 = rng.integers(0, 256, size=(n, 32), dtype=np.uint8)
    d_ent.copy_(_dev(entropy2))
    torch.cuda.synchronize()
    g.launch()
    e.synchronize(); torch.cuda.synchronize()
    got = (chal.cpu().numpy(), resp.cpu().numpy(), coms.cpu().numpy())
    assert int(bst.abs().sum().item()) == 0 and not bool(out.any().item())
    ts_h = np.stack([t0] * n)
    chal_h, resp_h, coms_h = T.prove_batch(eng, mod.statement, ts_h, secrets, inst, common, entropy2)
    assert (got[0] == chal_h).all() and (got[1] == resp_h).all() and (got[2] == coms_h).all()
    assert not (got[0] == want[0].cpu().numpy()).all()
    # ---- lifetime rules (zkp_mi355x.h, HIP graphs) -------------------------------------------------------------------------------
    # a failing call inside a capture: the context manager aborts the capture and the context stays usable
    import bench
    from zkp_amd.engine import FusedStatement
    other = FusedStatement(b"another statement", *bench.dleq_macro_statement())
    with pytest.raises(ZkpError):
        with e.capture():
            with torch.cuda.stream(stream):
                e.fused_prove_dev(other, n, pos, ts.data_ptr(), d_sec.data_ptr(), d_tbl.data_ptr(), d_ent.data_ptr(), chal.data_ptr(), resp.data_ptr(),
                                  coms.data_ptr(), st.data_ptr())            # no plan for this statement yet: refused inside a capture
    e.capture_abort()                                                         # (no capture in progress any more: a no-op)
    chain()
    e.synchronize(); torch.cuda.synchronize()
    assert int(bst.abs().sum().item()) == 0 and not bool(out.any().item())
    g.launch()                                                                # the old graph is still valid: nothing it points into has moved
    e.synchronize()
    # a graph belongs to its context
    e2 = Engine(0)
    assert e._lib.zkp_graph_launch(g._h, e2._h) == -2 and b"context it was captured on" in e._lib.zkp_last_error()     # ZKP_ERR_ARG
    e2.close()
    # a larger call on the same context makes the workspace grow (it is reallocated): the graph is stale and refuses to replay
    n_big = 4 * n
    mod2, secrets2, inst2, common2 = _cmz_batch(n_big, 9)
    ts_big = np.stack([t0] * n_big)
    T.set_fused_min_batch(0)
    try:
        T.prove_batch(e, mod2.statement, ts_big, secrets2, inst2, common2, rng.integers(0, 256, size=(n_big, 32), dtype=np.uint8))
    finally:
        T.set_fused_min_batch(32)
    with pytest.raises(ZkpError, match="stale graph"):
        g.launch()
    with e.capture() as cap:                                                  # capture again: fine
        chain()
    cap.graph.launch()
    e.synchronize(); torch.cuda.synchronize()
    assert int(bst.abs().sum().item()) == 0 and not bool(out.any().item())
    # a graph must not outlive its context
    stale, lib, h_old = cap.graph, e._lib, e._h.value
    e.close()
    assert lib.zkp_graph_launch(stale._h, ctypes.c_void_p(h_old)) == -2 and b"destroyed" in lib.zkp_last_error()
    g.close()
    stale.close()


@pytest.mark.parametrize("single_use_tables,grouped", [(0, 0), (1, 0), (0, 1), (1, 1), (0, "masked"), (1, "masked"), (0, "lds"), (1, "lds"), (0, "interleave"), (0, "split"), (1, "split")])
def test_constant_time_single_use_schedules_agree_with_oracle(eng, single_use_tables, grouped):
    """ZKP_OPT_CT_SINGLE_USE_TABLES: a constant-time call serves a point that only one term multiplies either through a comb
    table (default when the call has shared points) or through the masked radix-16 ladder; both must give the oracle's bytes.
    CMZ shape (Q is the single-use point) and a DLEQ-like shape without any shared point (always the ladder).
    grouped = ZKP_OPT_GROUPED_COMB: the terms of points with 9 or more uses are listed point by point and a wavefront walks the rows of its
    (at most 8) tables together (the default of large calls), forced here at sizes that leave blocks partly filled and wavefronts with few and
    with many tables.  "masked" / "lds" = ZKP_OPT_CT_LOOKUP 1 / 2: the other two ways to pick a table entry (default 0: the lane crossbar)."""
    from zkp_amd.engine import Engine
    import bench
    rng = np.random.default_rng(12)
    e = Engine(0)
    if grouped in ("masked", "lds") and len(e.ct_lookups) == 1:
        e.close()
        pytest.skip("ZKP_OPT_CT_LOOKUP 1 / 2 exist in -DZKP_HOT_W=6 builds only (the shipped library: 7-bit windows, lane crossbar)")
    e.set_option(3, single_use_tables)
    if grouped == "masked":
        # ZKP_OPT_CT_MASKED_SCANS: the safe mode -- every table look-up of the constant-time call is a masked scan over the whole
        # row (fixed-base rows too), and the grouped walk stays off even when asked for
        e.set_option(9, 1)
        e.set_option(6, 1)
    elif grouped == "lds":
        # ZKP_OPT_CT_LOOKUP = 2: the look-up of rounds 2 - 4 (rows replicated in LDS, read at the digit's index), grouped walk through LDS
        e.set_option(9, 2)
        e.set_option(6, 1)
    elif grouped == "split":
        # ZKP_OPT_COMB_SPLIT: a quad of lanes per masked comb scan, one 64-bit window each, joined over DPP, next to the grouped walk (round 6: the default of
        # constant-time calls of 8,192 .. 400,000 terms on the latency schedule); calls with a ladder class keep one lane per scan
        e.set_option(16, 1)
    elif grouped == "interleave":
        # ZKP_OPT_LADDER_INTERLEAVE: the ladder blocks spread over the front of the term kernel's grid (the default of launches with
        # 65,536 single-use points or more), forced here on grids of a few blocks: strides 0 (too few blocks), 2 and more
        e.set_option(11, 1)
        e.set_option(6, 1)
    else:
        e.set_option(6, grouped)
    n = 80
    off, pidx, n_pts = bench.cmz_shape(n)
    ks = rng.integers(0, 256, size=(n_pts, 32), dtype=np.uint8)
    ks[:, 31] &= 0x0f
    base = np.frombuffer(BASEPOINT, np.uint8).reshape(1, 32)
    pts, _ = C.msm_many(np.arange(n_pts + 1, dtype=np.uint32), ks, np.zeros(n_pts, np.uint32), base, 0)
    e.prepare_fixed_points(pts[:11])
    sc = rng.integers(0, 256, size=(31 * n, 32), dtype=np.uint8)
    sc[:, 31] &= 0x0f
    sc[5] = 0; sc[6] = 0xff                                           # zero and a non-canonical 2^256 - 1 scalar
    got, st = e.msm_many(off, sc, pidx, pts, 1)
    want, wst = C.msm_many(off, sc, pidx, pts, 1)
    assert (st == wst).all() and (got == want).all()
    # the same with two proofs whose P does not decode: the ten MSMs on it report status 1, the others are unaffected
    bad = pts.copy()
    p_rows = sorted({int(v) for v in pidx if np.count_nonzero(pidx == v) >= 10 and v >= 11})
    assert len(p_rows) == n                                           # the per-proof P of every proof
    bad[p_rows[3]] = 0xff; bad[p_rows[41], 0] ^= 1
    got, st = e.msm_many(off, sc, pidx, bad, 1)
    want, wst = C.msm_many(off, sc, pidx, bad, 1)
    assert wst.sum() >= 10 and (st == wst).all() and (got[wst == 0] == want[wst == 0]).all()
    # 1,200 two-term MSMs  x G + y H_j : G shared by all (comb table), every H_j used once
    m = 1200
    km = rng.integers(0, 256, size=(m + 1, 32), dtype=np.uint8)
    km[:, 31] &= 0x0f
    table, _ = C.msm_many(np.arange(m + 2, dtype=np.uint32), km, np.zeros(m + 1, np.uint32), base, 0)   # point 0 = G' (not a registered fixed-base point), then H_0 .. H_{m-1}
    off2 = (2 * np.arange(m + 1)).astype(np.uint32)
    pidx2 = np.stack([np.zeros(m, np.uint32), 1 + np.arange(m, dtype=np.uint32)], axis=1).reshape(-1)
    sc2 = rng.integers(0, 256, size=(2 * m, 32), dtype=np.uint8)
    sc2[:, 31] &= 0x0f
    got, st = e.msm_many(off2, sc2, pidx2, table, 1)
    want, wst = C.msm_many(off2, sc2, pidx2, table, 1)
    assert (st == wst).all() and (got == want).all()
    # ... and with no shared point at all: 1,100 one-term MSMs on distinct points
    off3 = np.arange(1101, dtype=np.uint32)
    got, st = e.msm_many(off3, sc2[:1100], np.arange(1100, dtype=np.uint32), table[1:1101], 1)
    want, wst = C.msm_many(off3, sc2[:1100], np.arange(1100, dtype=np.uint32), table[1:1101], 1)
    assert (st == wst).all() and (got == want).all()
    e.close()


@pytest.mark.parametrize("single_use_tables", [0, 1])
def test_grouped_comb_walk_with_mixed_group_sizes(single_use_tables):
    """ZKP_OPT_GROUPED_COMB at its edges: points with exactly 8 and 9 uses (three tables in one 16-lane column), large groups that
    span columns and blocks, points below the threshold (2..7 uses: masked scans) and single-use points in the same constant-time
    call, an undecodable point inside a group, terms in shuffled order -- against the oracle, byte for byte."""
    from zkp_amd.engine import Engine
    rng = np.random.default_rng(77)
    uses = np.array([8, 9, 10, 23, 1, 2, 7, 8, 16, 8, 8, 9, 300, 1, 5] * 24, np.int64)
    n_pts = len(uses)
    ks = rng.integers(0, 256, size=(n_pts, 32), dtype=np.uint8)
    ks[:, 31] &= 0x0f
    base = np.frombuffer(BASEPOINT, np.uint8).reshape(1, 32)
    pts, _ = C.msm_many(np.arange(n_pts + 1, dtype=np.uint32), ks, np.zeros(n_pts, np.uint32), base, 0)
    pts = pts.copy()
    pts[7] = 0xff                                                        # a point of a group of 8 that does not decode
    pidx = np.repeat(np.arange(n_pts, dtype=np.uint32), uses)
    rng.shuffle(pidx)
    n_terms = len(pidx)
    cuts = np.sort(rng.choice(np.arange(1, n_terms), size=n_terms // 3, replace=False))
    off = np.concatenate([[0], cuts, [n_terms]]).astype(np.uint32)       # MSMs of 1 .. ~10 terms
    sc = rng.integers(0, 256, size=(n_terms, 32), dtype=np.uint8)
    sc[:, 31] &= 0x0f
    sc[3] = 0; sc[4] = 0xff
    want, wst = C.msm_many(off, sc, pidx, pts, 1)
    assert wst.any() and not wst.all()
    e = Engine(0)
    e.prepare_fixed_points(pts[[3, 12]])                                 # two of the points are fixed-base points as well
    e.set_option(3, single_use_tables)
    for grouped, masked in ((1, 0), (0, 0), (1, 1), (1, 2), (0, 2)):
        if masked not in e.ct_lookups:
            continue
        e.set_option(6, grouped)
        e.set_option(9, masked)                                          # ZKP_OPT_CT_LOOKUP
        got, st = e.msm_many(off, sc, pidx, pts, 1)
        assert (st == wst).all() and (got[wst == 0] == want[wst == 0]).all(), (grouped, masked)
    e.close()


@pytest.mark.parametrize("opts", [{}, {8: 0}, {4: 1}, {11: 1}])
def test_mid_size_dev_call_variants_give_the_same_proofs(eng, opts):
    """Asynchronous _dev calls of 8,192 .. 65,535 proofs run the lane-pair transcript inside the comb tables' launch by default
    (k_tables_transcript; round-3 wide-call rule).  Default, ZKP_OPT_FUSE_TABLES_TRANSCRIPT = 0 and one transcript lane per proof
    must all give the bytes of the synchronous host-pointer flow, and the proofs batch-verify in two batches of one pass.
    {11: 1} = ZKP_OPT_LADDER_INTERLEAVE forced on (33 ladder blocks spread over the first half of the term kernel's grid)."""
    torch = _torch()
    from zkp_amd.engine import Engine
    n = 8256                                                             # 129 x 64 proofs
    mod, secrets, inst, common = _cmz_batch(64, 35)
    reps = n // 64
    secrets = np.ascontiguousarray(np.tile(secrets, (reps, 1, 1)))
    inst = np.ascontiguousarray(np.tile(inst, (1, reps, 1)))
    entropy = np.random.default_rng(36).integers(0, 256, size=(n, 32), dtype=np.uint8)
    t0 = T.Transcript(b"mid-size").state
    pos = int(t0[200]) | int(t0[201]) << 8 | int(t0[202]) << 16
    ts = np.stack([t0] * n)
    chal, resp, coms = T.prove_batch(eng, mod.statement, ts, secrets, inst, common, entropy)      # synchronous reference
    e = Engine(0)
    for k, v in opts.items():
        e.set_option(k, v)
    e.prepare_fixed_points(common)
    fst = _cmz_fused_statement()
    z = lambda *s: torch.zeros(s, dtype=torch.uint8, device="cuda:0")
    d_ts, d_sec, d_tbl, d_ent = _dev(np.stack([t0] * n)), _dev(secrets), _dev(np.concatenate([common, inst.reshape(-1, 32)])), _dev(entropy)
    d_chal, d_resp, d_coms, d_st = z(n, 32), z(n, 21, 32), z(n, 11, 32), z(11 * n)
    torch.cuda.synchronize()
    e.fused_prove_dev(fst, n, pos, d_ts.data_ptr(), d_sec.data_ptr(), d_tbl.data_ptr(), d_ent.data_ptr(), d_chal.data_ptr(), d_resp.data_ptr(),
                      d_coms.data_ptr(), d_st.data_ptr())
    e.synchronize()
    assert not d_st.cpu().numpy().any()
    assert (d_chal.cpu().numpy() == chal).all() and (d_resp.cpu().numpy() == resp).all() and (d_coms.cpu().numpy() == coms).all()
    assert (d_ts.cpu().numpy()[:, :203] == ts[:, :203]).all()
    w = np.random.default_rng(37).integers(0, 256, size=(11, n, 16), dtype=np.uint8)
    d_ts2, d_w = _dev(np.stack([t0] * n)), _dev(w)
    d_pts = z(12 + 24 * n, 32)
    d_pts[: 12 + 13 * n] = d_tbl
    d_out, d_bst = torch.ones((2, 32), dtype=torch.uint8, device="cuda:0"), torch.ones((2, 2), dtype=torch.int32, device="cuda:0")
    torch.cuda.synchronize()
    e.fused_batch_verify_many_dev(fst, 2, n // 2, pos, d_ts2.data_ptr(), d_pts.data_ptr(), d_coms.data_ptr(), d_resp.data_ptr(), d_w.data_ptr(),
                                  d_out.data_ptr(), d_bst.data_ptr())
    e.synchronize()
    assert not d_out.cpu().numpy().any() and not d_bst.cpu().numpy().any()
    e.close()


def test_masked_scan_safe_mode_gives_the_same_proofs(eng):
    """ZKP_OPT_CT_LOOKUP (0 lane crossbar, 1 masked scans, 2 LDS rows) through the fused prover at the size where the wide-call variants (grouped walk, ladder) switch on
    by themselves: byte-identical proofs, and the batch of them verifies."""
    from zkp_amd.engine import Engine
    n = 13000                                                            # 403,000 terms: a wide call also for the synchronous entry points
    mod, secrets, inst, common = _cmz_batch(64, 33)
    reps = (n + 63) // 64
    secrets = np.ascontiguousarray(np.tile(secrets, (reps, 1, 1))[:n])
    inst = np.ascontiguousarray(np.tile(inst, (1, reps, 1))[:, :n])
    entropy = np.random.default_rng(34).integers(0, 256, size=(n, 32), dtype=np.uint8)
    out = {}
    for masked in eng.ct_lookups:
        e = Engine(0)
        e.set_option(9, masked)
        ts = np.stack([T.Transcript(b"safe-mode").state] * n)
        out[masked] = T.prove_batch(e, mod.statement, ts, secrets, inst, common, entropy)
        e.close()
    if len(out) == 1:                                                    # the shipped build: nothing to compare with, but unsupported values must be refused
        e = Engine(0)
        for v in (1, 2, 3):
            with pytest.raises(Exception):
                e.set_option(9, v)
        e.close()
    for other in [k for k in out if k != 0]:
        for a, b in zip(out[0], out[other]):
            assert (a == b).all(), other
    ts = np.stack([T.Transcript(b"safe-mode").state] * n)
    T.batch_verify(eng, mod.statement, ts, inst, common, out[0][2], out[0][1])


_ONE_QUEUE_SCRIPT = r"""
import sys
sys.path.insert(0, %(root)r)
import numpy as np
import torch
torch.cuda.init()
from zkp_amd.engine import Engine
from zkp_amd import toolbox as T
from tests.test_gpu_toolbox import _cmz_batch
from tests.test_gpu_device_entry import _cmz_fused_statement, _dev
n = 96
mod, secrets, inst, common = _cmz_batch(n, 8)
fst = _cmz_fused_statement()
e = Engine(0)
e.set_option(5, 1)                                  # ZKP_OPT_DEV_OVERLAP: the _dev flows fork onto the side stream
stream = torch.cuda.Stream()
e.set_stream(stream.cuda_stream)
e.prepare_fixed_points(common)
t0 = T.Transcript(b"one-queue").state
pos = int(t0[200]) | int(t0[201]) << 8 | int(t0[202]) << 16
rng = np.random.default_rng(5)
entropy = rng.integers(0, 256, size=(n, 32), dtype=np.uint8)
d_ts0, d_ent, d_sec = _dev(np.stack([t0] * n)), _dev(entropy), _dev(secrets)
d_tbl = _dev(np.concatenate([common, inst.reshape(-1, 32)]))
z8 = lambda *s: torch.zeros(s, dtype=torch.uint8, device="cuda:0")
ts, chal, resp, coms, st = z8(n, 208), z8(n, 32), z8(n, 21, 32), z8(n, 11, 32), z8(11 * n)
def chain():
    with torch.cuda.stream(stream):
        ts.copy_(d_ts0, non_blocking=True)
        e.fused_prove_dev(fst, n, pos, ts.data_ptr(), d_sec.data_ptr(), d_tbl.data_ptr(), d_ent.data_ptr(), chal.data_ptr(), resp.data_ptr(), coms.data_ptr(), st.data_ptr())
chain()
e.synchronize(); torch.cuda.synchronize()
want = [x.clone() for x in (chal, resp, coms)]
with e.capture() as cap:
    chain()
for x in (chal, resp, coms):
    x.zero_()
for _ in range(3):
    cap.graph.launch()
e.synchronize(); torch.cuda.synchronize()
assert all(bool((a == b).all().item()) for a, b in zip(want, (chal, resp, coms)))
ts_h = np.stack([t0] * n)
chal_h, resp_h, coms_h = T.prove_batch(e, mod.statement, ts_h, secrets, inst, common, entropy)
assert (chal.cpu().numpy() == chal_h).all() and (resp.cpu().numpy() == resp_h).all() and (coms.cpu().numpy() == coms_h).all()
print("ONE-QUEUE-OK")
"""


def test_forked_flow_in_a_graph_with_one_hardware_queue():
    """GPU_MAX_HW_QUEUES=1 (what a one-stream profiler run sets) + ZKP_OPT_DEV_OVERLAP + graph replay: ROCm 7.2.0 crashes in hipGraphLaunch on a
    graph with a forked branch when the process owns a single hardware queue, so a capture recorded under that setting keeps the flow on the
    context's stream (fused_flows.h: side_forks).  Own process: the variable is read when the HIP runtime initialises."""
    import os
    import subprocess
    import sys
    _torch()
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, GPU_MAX_HW_QUEUES="1")
    r = subprocess.run([sys.executable, "-c", _ONE_QUEUE_SCRIPT % {"root": root}], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "ONE-QUEUE-OK" in r.stdout, (r.returncode, r.stdout[-2000:], r.stderr[-2000:])
