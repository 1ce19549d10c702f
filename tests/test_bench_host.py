"""bench.py's host-side logic without a GPU: the statement descriptors of the four BASELINE workloads, the generic
instance maker (here driven by the C oracle's MSM instead of the engine's), the stream picker and the source hash."""
import numpy as np
import pytest

import bench
from oracle import cbind as C


class _OracleEngine:
    """stand-in with the one method make_instance uses (the engine's msm_many has the same signature)"""

    def msm_many(self, off, scalars, pidx, points, flags=0):
        return C.msm_many(off, scalars, pidx, points, flags)


@pytest.mark.parametrize("st_fn,label", [(bench.cmz_statement, b"CMZ cred show n=10"), (bench.dleq_macro_statement, b"DLEQ proof"),
                                         (bench.dleq_capi_statement, b"DLEQProof"), (bench.w64_statement, b"W64"),
                                         (bench.W64_FORMS["constraints"], b"W64"), (bench.W64_FORMS["constraints2"], b"W64")])
def test_make_instance_gives_provable_statements(st_fn, label):
    C.build()
    st = st_fn()
    secrets_l, points, cons = st
    n = 3
    rng = np.random.default_rng(5)
    secrets, inst, common = bench.make_instance(_OracleEngine(), st, n, rng)
    ns = sum(1 for _, c in points if c)
    assert secrets.shape == (n, len(secrets_l), 32) and inst.shape == (len(points) - ns, n, 32) and common.shape == (ns, 32)
    names = [nm.decode() for nm, _ in points]
    cst = C.Statement(label, [s.decode() for s in secrets_l], [(nm.decode(), c) for nm, c in points],
                      [(names[l], [(secrets_l[s].decode(), names[q]) for s, q in lc]) for l, lc in cons])
    com_rank, inst_rank = {}, {}
    for i, (_, c) in enumerate(points):
        (com_rank if c else inst_rank)[i] = len(com_rank if c else inst_rank)
    coms, resp = [], []
    for j in range(n):
        pts = np.stack([common[com_rank[i]] if i in com_rank else inst[inst_rank[i], j] for i in range(len(points))])
        _, er, ek, _ = C.prove(cst, b"Benchmark", secrets[j], pts, bytes([j + 1]) * 32)
        coms.append(ek); resp.append(er)
    w = rng.integers(0, 256, size=(len(cons), n, 16), dtype=np.uint8)
    assert C.batch_verify(cst, b"Benchmark", n, inst, common, np.stack(coms), np.stack(resp), w) == 0
    bad = np.stack(resp).copy()
    bad[1, 0, 0] ^= 1
    assert C.batch_verify(cst, b"Benchmark", n, inst, common, np.stack(coms), bad, w) != 0


def test_stream_picker_and_source_hash():
    assert [bench.pick_streams(k) for k in (1, 5, 20, 25)] == [1, 5, 20, 25]
    assert bench.pick_streams(200) == 25
    for k in (26, 32, 50, 100, 200, 1000):
        s = bench.pick_streams(k)
        assert 12 <= s <= 25 and (-k) % s == min((-k) % t for t in range(12, 26))
    h = bench.source_sha256()
    assert len(h) == 64 and h == bench.source_sha256()


def test_call_shape_picker():
    """--config 2 packs K batches into one call chain: K divides --steps (exactly --steps batches are timed)"""
    for steps in (1, 4, 7, 20, 40, 100, 1000):
        k, s = bench.pick_call_shape(steps)
        assert steps % k == 0 and 1 <= s <= steps // k and k <= bench.MAX_BATCHES_PER_CALL
    assert bench.pick_call_shape(20) == (5, 4) and bench.pick_call_shape(1000) == (50, 4) and bench.pick_call_shape(4) == (1, 4)
    assert bench.pick_call_shape(20, want_k=5, want_streams=3) == (5, 3)
    assert bench.pick_call_shape(20, want_k=1) == (1, 20)
    with pytest.raises(SystemExit):
        bench.pick_call_shape(20, want_k=3)


def test_workload_table_is_consistent():
    for name, (desc, parts, batch, streams, steps) in bench.WORKLOADS.items():
        assert "%d" in desc and abs(sum(share for _, _, share, _ in parts) - 1.0) < 1e-9 and batch > 0 and steps > 0
        for label, st_fn, share, flows in parts:
            secrets, points, cons = st_fn()
            assert all(f in ("prove", "batch_verify") for f in flows) and len(cons) >= 1
            assert sum(1 for _, c in points if c) <= 64          # every common point gets a fixed-base table slot
