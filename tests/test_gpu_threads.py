"""Per-context thread safety of the C ABI ("thread-safe per ctx"; the reference's Prover / Verifier / BatchVerifier objects are
single-threaded, different objects are independent -- prover.rs:24, batch_verifier.rs:32): several host threads, each with its OWN context on the same GPU,
issue synchronous toolbox calls at the same time (ctypes releases the GIL inside a call).  Every thread must get the bytes a lone thread gets, and an error
raised in one thread (its message lives in thread-local storage: zkp_last_error) must not leak into the others."""
import threading

import numpy as np
import pytest

from oracle import cbind as C
from oracle import model as M
from zkp_amd import toolbox as T
from tests.test_gpu_toolbox import BASEPOINT, _cmz_batch
from tests.test_gpu_fused import _dleq_batch

pytestmark = pytest.mark.gpu
LABEL = b"threads"


def _t0(n):
    return np.stack([T.Transcript(LABEL).state] * n)


@pytest.mark.parametrize("n_threads,sync_schedule", [(4, 0), (8, 0), (6, 1)])
def test_concurrent_contexts_give_the_bytes_of_a_lone_thread(n_threads, sync_schedule):
    from zkp_amd.engine import Engine, ZkpError
    n = 160
    jobs = []
    for t in range(n_threads):                      # thread t: its own statement instance, entropy and weights (CMZ for even t, DLEQ for odd t)
        rng = np.random.default_rng(100 + t)
        if t % 2 == 0:
            mod, secrets, inst, common = _cmz_batch(n, 900 + t)
        else:
            mod, secrets, A, B, H = _dleq_batch(n, 900 + t)
            inst = np.ascontiguousarray(np.stack([A, B, H]))
            common = np.frombuffer(BASEPOINT, np.uint8).reshape(1, 32).copy()
        st = mod.statement
        jobs.append(dict(st=st, secrets=secrets, inst=inst, common=common, entropy=rng.integers(0, 256, size=(n, 32), dtype=np.uint8),
                         w=rng.integers(0, 256, size=(st.nc, n, 16), dtype=np.uint8)))

    def flows(e, j, rounds):
        out = []
        for _ in range(rounds):
            ts = _t0(n)
            chal, resp, coms = T.prove_batch(e, j["st"], ts, j["secrets"], j["inst"], j["common"], j["entropy"])
            res = T.verify_compact_batch(e, j["st"], _t0(n), j["inst"], j["common"], chal, resp)
            T.batch_verify(e, j["st"], _t0(n), j["inst"], j["common"], coms, resp, j["w"])
            bad = resp.copy()
            bad[n // 2, 0, 0] ^= 1
            ok, each = T.batch_verify_locate(e, j["st"], _t0(n), j["inst"], j["common"], coms, bad, j["w"])
            out.append((chal, resp, coms, ts, res, ok, each))
        return out

    # what a lone thread gets
    e0 = Engine(0)
    want = [flows(e0, j, 1)[0] for j in jobs]
    e0.close()
    for w in want:
        assert not w[4].any() and not w[5] and w[6][n // 2] == 1 and w[6].sum() == 1
    # proofs of thread 0 against the oracle (the rest of the chain is pinned by the other test files)
    cst = C.Statement.from_model(M.cmz_statement(10))
    for jx in (0, n - 1):
        pts = np.concatenate([jobs[0]["inst"][:, jx], jobs[0]["common"]])
        ec, er, ek, _ = C.prove(cst, LABEL, jobs[0]["secrets"][jx], pts, jobs[0]["entropy"][jx].tobytes())
        assert bytes(want[0][0][jx]) == bytes(ec) and want[0][1][jx].tobytes() == bytes(er) and want[0][2][jx].tobytes() == bytes(ek)

    engines = [Engine(0) for _ in range(n_threads)]
    for e in engines:
        e.set_option(14, sync_schedule)             # ZKP_OPT_SYNC_SCHEDULE: 1 = the synchronous calls run the jobs' throughput schedule (same bytes)
    got, errors = [None] * n_threads, []
    start = threading.Barrier(n_threads + 1)

    def worker(t):
        try:
            start.wait()
            got[t] = flows(engines[t], jobs[t], 3)
        except Exception as ex:          # noqa: BLE001
            errors.append((t, ex))

    def offender():
        # calls that FAIL, in a loop, on a context of its own while the others work: its error text stays in this thread
        e = Engine(0)
        start.wait()
        try:
            for _ in range(50):
                with pytest.raises(ZkpError, match="NULL"):
                    e.msm_optional_dev(5, 0, 0, 0, 0)          # NULL device pointers: ZKP_ERR_ARG
        except Exception as ex:          # noqa: BLE001
            errors.append(("offender", ex))
        finally:
            e.close()

    th = [threading.Thread(target=worker, args=(t,)) for t in range(n_threads)] + [threading.Thread(target=offender)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    for e in engines:
        e.close()
    assert not errors, errors
    for t in range(n_threads):
        for r in got[t]:
            for a, b in zip(r, want[t]):
                assert (np.asarray(a) == np.asarray(b)).all(), "thread %d differs from the lone thread" % t
