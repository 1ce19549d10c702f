"""zkp_pipe (include/zkp_toolbox.h, "pipelines and device groups"): asynchronous jobs on host buffers and the synchronous calls
sharded over several contexts.

What must hold:
  * async == sync == oracle, byte for byte, for K in {1, 5} batches per job -- with explicit entropy / weights, with one shared
    start transcript or N blobs, from ordinary and from pinned buffers, for a proof range passed by row stride;
  * jobs are independent: a rejected batch in one job leaves the verdicts of the others intact; an infrastructure failure of one
    job (here: ZKP_ERR_OOM through ZKP_OPT_WS_LIMIT_BYTES on one context) is a negative code for that job only;
  * the device-side randomness (entropy / weights == NULL) is the ChaCha20 stream the host KAT pins, and two jobs never share it;
  * n = 2, 3, 8 contexts (here all on GPU 0; on a node: one per GPU): proofs and per-proof verdicts equal the single-context call's,
    a bad proof in shard g fails the AND and `locate` names it, uneven ranges, more contexts than proofs.
"""
import ctypes

import numpy as np
import pytest

from oracle import cbind as C
from oracle import model as M
from zkp_amd import toolbox as T
from tests.test_gpu_toolbox import _cmz_batch
from tests.test_gpu_fused import _dleq_batch

pytestmark = pytest.mark.gpu
LABEL = b"Benchmark"
ZKP_OPT_WS_LIMIT_BYTES = 12


@pytest.fixture(scope="module")
def eng():
    from zkp_amd.engine import Engine
    e = Engine(0)
    yield e
    e.close()
    T.set_fused_min_batch(32)


@pytest.fixture(scope="module")
def cmz():
    n = 5 * 96
    mod, secrets, inst, common = _cmz_batch(n, 4242)
    rng = np.random.default_rng(7)
    entropy = rng.integers(0, 256, size=(n, 32), dtype=np.uint8)
    w = rng.integers(0, 256, size=(mod.statement.nc, n, 16), dtype=np.uint8)
    return mod, secrets, inst, common, entropy, w


def _t0(n=None):
    s = T.Transcript(LABEL).state
    return s if n is None else np.stack([s] * n)


def _oracle_proofs(secrets, inst, common, entropy, idx):
    cst = C.Statement.from_model(M.cmz_statement(10))
    out = []
    for j in idx:
        pts = np.concatenate([inst[:, j], common])
        ec, er, ek, _ = C.prove(cst, LABEL, secrets[j], pts, entropy[j].tobytes())
        out.append((ec, er, ek))
    return out


@pytest.mark.parametrize("K", [1, 5])
def test_async_equals_sync_equals_oracle(eng, cmz, K):
    mod, secrets, inst, common, entropy, w = cmz
    st = mod.statement
    n_each = 96
    n = K * n_each
    sl = slice(0, n)
    inst_n = np.ascontiguousarray(inst[:, sl])
    w_n = np.ascontiguousarray(w[:, sl])
    # synchronous single-context reference
    ts_sync = _t0(n)
    chal_s, resp_s, coms_s = T.prove_batch(eng, st, ts_sync, secrets[sl], inst_n, common, entropy[sl])
    ts_v = _t0(n)
    v_sync = T.batch_verify_many(eng, st, K, ts_v, inst_n, common, coms_s, resp_s, w_n)
    assert not v_sync.any()
    with T.Pipe((0,), 3) as pipe:
        assert pipe.num_contexts == 3
        # (a) shared start transcript, (b) N blobs, (c) pinned buffers, (d) a column range of the big arrays passed by stride
        ja = pipe.submit_prove(st, n, _t0(), secrets[sl], inst_n, common, entropy[sl], want_transcripts=True)
        jb = pipe.submit_prove(st, n, _t0(n), secrets[sl], inst_n, common, entropy[sl])
        pin = dict(chal=T.pinned_empty((n, 32)), resp=T.pinned_empty((n, st.m, 32)), coms=T.pinned_empty((n, st.nc, 32)), ts=T.pinned_empty((n, 208)))
        jc = pipe.submit_prove(st, n, T.pinned_copy(_t0()), T.pinned_copy(secrets[sl]), T.pinned_copy(inst_n), T.pinned_copy(common), T.pinned_copy(entropy[sl]),
                               want_transcripts=True, out=pin)
        assert pipe.jobs_in_flight == 3
        with pytest.raises(BlockingIOError):
            pipe.submit_prove(st, n, _t0(), secrets[sl], inst_n, common, entropy[sl])
        ca, ra, ka, tsa = ja.wait()
        cb, rb, kb = jb.wait()
        cc, rc_, kc, tsc = jc.wait()
        assert cc is pin["chal"] and pipe.jobs_in_flight == 0
        for c_, r_, k_ in ((ca, ra, ka), (cb, rb, kb), (cc, rc_, kc)):
            assert (c_ == chal_s).all() and (r_ == resp_s).all() and (k_ == coms_s).all()
        assert (tsa[:, :203] == ts_sync[:, :203]).all() and (tsc[:, :203] == ts_sync[:, :203]).all()
        lo = 37                                             # proofs [lo, lo + n) ... of a batch that has more columns than the job takes
        if lo + n <= inst.shape[1]:
            jd = pipe.submit_prove(st, n, _t0(), secrets[lo:lo + n], inst[:, lo:], common, entropy[lo:lo + n], inst_stride=inst.shape[1])
            cd, rd, kd = jd.wait()
            ts2 = _t0(n)
            c2, r2, k2 = T.prove_batch(eng, st, ts2, secrets[lo:lo + n], np.ascontiguousarray(inst[:, lo:lo + n]), common, entropy[lo:lo + n])
            assert (cd == c2).all() and (rd == r2).all() and (kd == k2).all()
        # oracle: every 7th proof of the job
        idx = list(range(0, n, 7))
        for j, (ec, er, ek) in zip(idx, _oracle_proofs(secrets, inst, common, entropy, idx)):
            assert ca[j].tobytes() == ec.tobytes() and (ra[j] == er).all() and (ka[j] == ek).all(), j
        # batch verification jobs: K verdicts per job, transcripts as the synchronous call leaves them
        jv = pipe.submit_batch_verify_many(st, K, n_each, _t0(), inst_n, common, coms_s, resp_s, w_n, want_transcripts=True)
        jw = pipe.submit_batch_verify_many(st, K, n_each, _t0(n), inst_n, common, T.pinned_copy(coms_s), T.pinned_copy(resp_s), T.pinned_copy(w_n))
        verd, tsv = jv.wait()
        (verd2,) = jw.wait()
        assert (verd == 0).all() and (verd2 == 0).all()
        assert (tsv[:, :203] == ts_v[:, :203]).all()
        cst = C.Statement.from_model(M.cmz_statement(10))
        for b in range(K):
            s2 = slice(b * n_each, (b + 1) * n_each)
            assert C.batch_verify(cst, LABEL, n_each, np.ascontiguousarray(inst_n[:, s2]), common, coms_s[s2], resp_s[s2], np.ascontiguousarray(w_n[:, s2])) == 0
        # per-proof verifiers as jobs
        jx = pipe.submit_verify_compact(st, n, _t0(), inst_n, common, chal_s, resp_s)
        jy = pipe.submit_verify_batchable_each(st, n, _t0(), inst_n, common, coms_s, resp_s)
        bad = resp_s.copy()
        bad[n // 2, 3, 1] ^= 4
        jz = pipe.submit_verify_compact(st, n, _t0(), inst_n, common, chal_s, bad)
        (rx,), (ry,), (rz,) = jx.wait(), jy.wait(), jz.wait()
        assert not rx.any() and not ry.any()
        assert rz[n // 2] == 1 and rz.sum() == 1


def test_failing_job_leaves_the_others_intact(eng, cmz):
    mod, secrets, inst, common, entropy, w = cmz
    st = mod.statement
    K, n_each = 5, 96
    n = K * n_each
    ts = _t0(n)
    chal, resp, coms = T.prove_batch(eng, st, ts, secrets, inst, common, entropy)
    bad = resp.copy()
    j = 3 * n_each + 11
    bad[j, 20, 5] ^= 1
    cst = C.Statement.from_model(M.cmz_statement(10))
    with T.Pipe((0,), 4) as pipe:
        jobs = [pipe.submit_batch_verify_many(st, K, n_each, _t0(), inst, common, coms, resp if i != 1 else bad, w) for i in range(3)]
        verdicts = [jb.wait()[0] for jb in jobs]
        assert (verdicts[0] == 0).all() and (verdicts[2] == 0).all()
        expect = np.zeros(K, np.int32)
        expect[3] = 1
        assert (verdicts[1] == expect).all()
        for b in range(K):                              # the oracle's BatchVerifier, batch by batch, on the bad job's inputs
            s2 = slice(b * n_each, (b + 1) * n_each)
            assert C.batch_verify(cst, LABEL, n_each, np.ascontiguousarray(inst[:, s2]), common, coms[s2], bad[s2], np.ascontiguousarray(w[:, s2])) == expect[b]
        # an infrastructure failure on ONE context (its workspace may not grow): a negative code for that job, and only for it
        pipe.set_option(ZKP_OPT_WS_LIMIT_BYTES, 1 << 20, context=0)
        big_n = 2048                                    # needs far more than 1 MiB of workspace; contexts 1.. serve it, context 0 refuses
        mod2, sec2, inst2, com2 = _cmz_batch(big_n, 99)
        ent2 = np.random.default_rng(3).integers(0, 256, size=(big_n, 32), dtype=np.uint8)
        results = []
        for i in range(4):
            try:
                results.append(pipe.submit_prove(mod2.statement, big_n, _t0(), sec2, inst2, com2, ent2))
            except Exception as e:                      # noqa: BLE001
                results.append(e)
        failed = [r for r in results if isinstance(r, Exception)]
        ok = [r for r in results if not isinstance(r, Exception)]
        assert len(failed) == 1 and "code -4" in str(failed[0]), results            # ZKP_ERR_OOM, from the capped context
        assert "ZKP_OPT_WS_LIMIT_BYTES" in pipe.last_error()
        outs = [jb.wait() for jb in ok]
        for c_, r_, k_ in outs[1:]:
            assert (c_ == outs[0][0]).all() and (r_ == outs[0][1]).all() and (k_ == outs[0][2]).all()
        ts2 = _t0(big_n)
        T.batch_verify(eng, mod2.statement, ts2, inst2, com2, outs[0][2], outs[0][1])
        assert pipe.jobs_in_flight == 0
        pipe.set_option(ZKP_OPT_WS_LIMIT_BYTES, 0, context=0)
        again = [pipe.submit_prove(mod2.statement, big_n, _t0(), sec2, inst2, com2, ent2) for _ in range(4)]     # every context, the capped one included
        for jb in again:
            assert (jb.wait()[0] == outs[0][0]).all()


def test_device_randomness_is_the_pinned_chacha_stream(eng):
    import torch
    from zkp_amd.engine import load_library
    hip = load_library()
    hip.zkp_chacha20_fill_dev.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_uint64, ctypes.c_uint64, ctypes.c_void_p, ctypes.c_size_t]
    key = bytes(range(32))
    nonce, first = 0x0123456789ABCDEF, 0xFFFFFFFE                  # the counter crosses 32 bits inside the run
    blocks = 300
    d = torch.zeros(blocks * 64, dtype=torch.uint8, device="cuda:0")
    assert hip.zkp_chacha20_fill_dev(eng._h, key, nonce, first, d.data_ptr(), blocks * 64) == 0
    eng.synchronize()
    got = d.cpu().numpy().tobytes()
    want = b""
    out = ctypes.create_string_buffer(64)
    T.lib().zkp_chacha20_block.argtypes = [ctypes.c_char_p, ctypes.c_uint64, ctypes.c_uint64, ctypes.c_char_p]
    for b in range(blocks):
        T.lib().zkp_chacha20_block(key, first + b, nonce, out)
        want += out.raw
    assert got == want
    # RFC 8439 section 2.3.2 through the same host function (counter 1, nonce words 00:00:00:09 | 00:00:00:4a | 00:00:00:00 do not fit a
    # 64-bit-counter layout, so this is the all-zero-key block 0 of the original ChaCha20 instead: 76 b8 e0 ad a0 f1 3d 90 ...)
    T.lib().zkp_chacha20_block(bytes(32), 0, 0, out)
    assert out.raw[:8].hex() == "76b8e0ada0f13d90"


def test_entropy_and_weights_from_the_device(eng, cmz):
    mod, secrets, inst, common, entropy, w = cmz
    st = mod.statement
    n = 192
    sl = slice(0, n)
    inst_n = np.ascontiguousarray(inst[:, sl])
    with T.Pipe((0,), 3) as pipe:
        j1 = pipe.submit_prove(st, n, _t0(), secrets[sl], inst_n, common)           # entropy = None: getrandom() seed, ChaCha20 on the device
        j2 = pipe.submit_prove(st, n, _t0(), secrets[sl], inst_n, common)
        c1, r1, k1 = j1.wait()
        c2, r2, k2 = j2.wait()
        assert not (k1 == k2).all(axis=2).any()                                     # fresh blindings in every job and every proof
        assert len({k1[j].tobytes() for j in range(n)}) == n
        for r_, k_, c_ in ((r1, k1, c1), (r2, k2, c2)):
            (v,) = pipe.submit_batch_verify_many(st, 2, n // 2, _t0(), inst_n, common, k_, r_).wait()      # weights = None: on the device too
            assert (v == 0).all()
            (res,) = pipe.submit_verify_compact(st, n, _t0(), inst_n, common, c_, r_).wait()
            assert not res.any()
            (res,) = pipe.submit_verify_batchable_each(st, n, _t0(), inst_n, common, k_, r_).wait()
            assert not res.any()
        bad = r1.copy()
        bad[100, 0, 0] ^= 1
        (v,) = pipe.submit_batch_verify_many(st, 2, n // 2, _t0(), inst_n, common, k1, bad).wait()
        assert list(v) == [0, 1]
        # the oracle accepts what the device-seeded prover made
        cst = C.Statement.from_model(M.cmz_statement(10))
        assert C.batch_verify(cst, LABEL, n, inst_n, common, k1, r1, np.ascontiguousarray(w[:, sl])) == 0


def test_small_and_ragged_batches_take_the_host_route_inside_submit(eng):
    mod, x, A, B, H = _dleq_batch(7, 5)
    st = mod.statement
    inst = np.stack([A, B, H])
    G = np.frombuffer(bytes.fromhex("e2f2ae0a6abc4e71a884a961c500515f58e30b6aa582dd8db6a65945e08d2d76"), np.uint8).reshape(1, 32)
    ent = np.random.default_rng(1).integers(0, 256, size=(7, 32), dtype=np.uint8)
    ts = np.stack([T.Transcript(b"t%d" % (j % 3) * (1 + j % 2)).state for j in range(7)])      # ragged: different labels, different lengths
    ts_ref = ts.copy()
    want = T.prove_batch(eng, st, ts_ref, x, inst, G, ent)
    with T.Pipe((0,), 2) as pipe:
        job = pipe.submit_prove(st, 7, ts.copy(), x, inst, G, ent, want_transcripts=True)
        assert job.done()
        c_, r_, k_, tso = job.wait()
        assert (c_ == want[0]).all() and (r_ == want[1]).all() and (k_ == want[2]).all() and (tso[:, :203] == ts_ref[:, :203]).all()
        (res,) = pipe.submit_verify_compact(st, 7, ts.copy(), inst, G, c_, r_).wait()
        assert not res.any()
        (v,) = pipe.submit_batch_verify_many(st, 1, 7, ts.copy(), inst, G, k_, r_).wait()
        assert list(v) == [0]


@pytest.mark.parametrize("n_ctx,n", [(2, 480), (3, 100), (8, 333), (8, 5)])
def test_sharded_calls_equal_the_single_context_calls(eng, cmz, n_ctx, n):
    """zkp_pipe_* over n_ctx contexts (all on GPU 0 here, one host thread per listed device): uneven ranges, more contexts than proofs"""
    mod, secrets, inst, common, entropy, w = cmz
    st = mod.statement
    sl = slice(0, n)
    inst_n = np.ascontiguousarray(inst[:, sl])
    ts1 = _t0(n)
    chal, resp, coms = T.prove_batch(eng, st, ts1, secrets[sl], inst_n, common, entropy[sl])
    with T.Pipe((0,) * n_ctx, 1) as pipe:
        assert pipe.num_contexts == n_ctx
        ts = _t0(n)
        c_, r_, k_ = pipe.prove_batch(st, ts, secrets[sl], inst_n, common, entropy[sl])
        assert (c_ == chal).all() and (r_ == resp).all() and (k_ == coms).all() and (ts[:, :203] == ts1[:, :203]).all()
        pipe.batch_verify(st, _t0(n), inst_n, common, coms, resp)                                   # weights from the device, per range
        pipe.batch_verify(st, _t0(n), inst_n, common, coms, resp, np.ascontiguousarray(w[:, sl]))
        assert not pipe.verify_compact_batch(st, _t0(n), inst_n, common, chal, resp).any()
        assert not pipe.verify_batchable_each(st, _t0(n), inst_n, common, coms, resp).any()
        ok, res = pipe.batch_verify_locate(st, _t0(n), inst_n, common, coms, resp)
        assert ok and not res.any()
        # a bad proof in every shard in turn: the AND fails and locate names exactly that proof
        G = min(n_ctx, n)
        for g in sorted({0, G // 2, G - 1}):
            lo, hi = g * n // G, (g + 1) * n // G
            j = (lo + hi) // 2
            bad = resp.copy()
            bad[j, 9, 2] ^= 0x10
            with pytest.raises(T.VerificationFailure):
                pipe.batch_verify(st, _t0(n), inst_n, common, coms, bad)
            ok, res = pipe.batch_verify_locate(st, _t0(n), inst_n, common, coms, bad)
            assert not ok and res[j] == 1 and res.sum() == 1
            rc_ = pipe.verify_compact_batch(st, _t0(n), inst_n, common, chal, bad)
            assert rc_[j] == 1 and rc_.sum() == 1
        with pytest.raises(T.BatchSizeMismatch):
            pipe.batch_verify(st, _t0(n - 1) if n > 1 else _t0(2), inst_n, common, coms, resp)
        # K whole batches over the contexts
        if n % 5 == 0:
            v = pipe.batch_verify_many(st, 5, _t0(n), inst_n, common, coms, resp, np.ascontiguousarray(w[:, sl]))
            assert (v == 0).all()
            bad = resp.copy()
            bad[n // 5 * 2 + (n // 5) // 2, 0, 0] ^= 1
            v = pipe.batch_verify_many(st, 5, _t0(n), inst_n, common, coms, bad)
            assert list(v) == [0, 0, 1, 0, 0]


def test_context_with_a_pending_job_refuses_other_calls(eng, cmz):
    mod, secrets, inst, common, entropy, w = cmz
    st = mod.statement
    n = 96
    from zkp_amd.engine import FusedStatement, load_library
    hip = load_library()
    fst = FusedStatement(st.proof_label, st.secrets, st.points, st.constraints)
    inst_n = np.ascontiguousarray(inst[:, :n])
    chal = np.zeros((n, 32), np.uint8); resp = np.zeros((n, st.m, 32), np.uint8); coms = np.zeros((n, st.nc, 32), np.uint8)
    inv = ctypes.c_int(-1)
    t0 = _t0()
    args = [eng._h, ctypes.cast(ctypes.byref(fst.c), ctypes.c_void_p), n, 1, T._p(t0), T._p(np.ascontiguousarray(secrets[:n])), T._p(inst_n), n, T._p(common), T._p(np.ascontiguousarray(entropy[:n])), None, None,
            T._p(chal), T._p(resp), T._p(coms), ctypes.byref(inv)]
    hip.zkp_fused_prove_submit.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint32, ctypes.c_uint32] + [ctypes.c_void_p] * 3 + [ctypes.c_uint32] + [ctypes.c_void_p] * 8
    assert hip.zkp_fused_prove_submit(*args) == 0
    assert hip.zkp_ctx_job_pending(eng._h) == 1
    assert hip.zkp_fused_prove_submit(*args) == -2                      # ZKP_ERR_ARG: one job per context
    assert b"pending" in hip.zkp_last_error()
    with pytest.raises(Exception):
        eng.msm_many(np.array([0, 1], np.uint32), np.zeros((1, 32), np.uint8), np.zeros(1, np.uint32), common[:1])
    assert hip.zkp_ctx_job_wait(eng._h) == 0 and inv.value == 0 and hip.zkp_ctx_job_pending(eng._h) == 0
    ts = _t0(n)
    c2, r2, k2 = T.prove_batch(eng, st, ts, secrets[:n], inst_n, common, entropy[:n])
    assert (chal == c2).all() and (resp == r2).all() and (coms == k2).all()


def test_verdict_words_are_rejected_until_the_job_has_run_and_discard_writes_nothing(eng, cmz):
    """Fail closed: results[N] of a per-proof verification job read "rejected" from the moment submit has checked its pointers
    until the job's own copy out overwrites them (deferred: at wait), so no failure path in between can leave a zero-initialised buffer
    reading as "verified".  zkp_ctx_job_discard retires a job without issuing its copies out and without touching caller memory."""
    mod, secrets, inst, common, entropy, w = cmz
    st = mod.statement
    n = 96
    from zkp_amd.engine import FusedStatement, load_library
    hip = load_library()
    fst = FusedStatement(st.proof_label, st.secrets, st.points, st.constraints)
    inst_n = np.ascontiguousarray(inst[:, :n])
    chal, resp, coms = T.prove_batch(eng, st, _t0(n), secrets[:n], inst_n, common, entropy[:n])
    t0 = _t0()
    results = np.zeros(n, np.uint8)                                     # what a careless caller hands over
    hip.zkp_fused_verify_compact_submit.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint32, ctypes.c_uint32] + [ctypes.c_void_p] * 2 + [ctypes.c_uint32] + [ctypes.c_void_p] * 5
    args = [eng._h, ctypes.cast(ctypes.byref(fst.c), ctypes.c_void_p), n, 1, T._p(t0), T._p(inst_n), n, T._p(common), T._p(chal), T._p(resp), None, T._p(results)]
    assert hip.zkp_fused_verify_compact_submit(*args) == 0
    assert results.all()                                                # rejected until the job says otherwise
    assert hip.zkp_ctx_job_wait(eng._h) == 0 and not results.any()
    # a submit that fails after its pointers were checked (stride smaller than N) leaves "rejected" behind, and no job
    results[:] = 0
    bad_args = list(args)
    bad_args[6] = n - 1
    assert hip.zkp_fused_verify_compact_submit(*bad_args) == -2 and results.all() and hip.zkp_ctx_job_pending(eng._h) == 0
    # discard: the job's kernels run, its outputs never arrive, the verdict words keep what submit put there
    results[:] = 0
    assert hip.zkp_fused_verify_compact_submit(*args) == 0 and hip.zkp_ctx_job_pending(eng._h) == 1
    hip.zkp_ctx_job_discard.argtypes = [ctypes.c_void_p]
    assert hip.zkp_ctx_job_discard(eng._h) == 0 and hip.zkp_ctx_job_pending(eng._h) == 0
    assert results.all()
    assert hip.zkp_ctx_job_wait(eng._h) == 0                            # nothing pending: OK at once
    # the context serves the next call
    assert not T.verify_compact_batch(eng, st, _t0(n), inst_n, common, chal, resp).any()


def test_dropped_jobs_and_closed_pipes_leave_no_dangling_pointers(cmz):
    """A Job dropped without wait() retires itself (the C side holds pointers into its arrays until zkp_job_wait), and Pipe.close() retires
    whatever is still in flight before the contexts go away."""
    import gc
    mod, secrets, inst, common, entropy, w = cmz
    st = mod.statement
    n = 256
    inst_n = np.ascontiguousarray(inst[:, :n])
    pipe = T.Pipe((0,), 3)
    job = pipe.submit_prove(st, n, _t0(), secrets[:n], inst_n, common, entropy[:n])
    assert pipe.jobs_in_flight == 1
    del job
    gc.collect()
    assert pipe.jobs_in_flight == 0                                     # __del__ waited
    j1 = pipe.submit_prove(st, n, _t0(), secrets[:n], inst_n, common, entropy[:n])
    chal, resp, coms = j1.wait()
    j2 = pipe.submit_batch_verify_many(st, 1, n, _t0(), inst_n, common, coms, resp)
    j3 = pipe.submit_verify_compact(st, n, _t0(), inst_n, common, chal, resp)
    assert pipe.jobs_in_flight == 2
    pipe.close()                                                        # retires j2 and j3 first
    assert j2._h is None and j3._h is None and j2.rc == 0 and j3.rc == 0
    assert j2.outputs[0].tolist() == [0] and not j3.outputs[0].any()
    with pytest.raises(RuntimeError):
        j2.wait()


def test_pinned_rings_on_the_gpus_numa_node():
    """zkp_host_alloc_on: pinned memory placed on the NUMA node of the GPU it feeds -- where the host tells us the node and
    lets us set a memory policy; everywhere else the call is zkp_host_alloc.  Always: usable pinned memory, visible as such."""
    from zkp_amd.engine import load_library
    hip = load_library()
    hip.zkp_host_alloc_on.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_size_t, ctypes.c_int]
    hip.zkp_host_node_of.argtypes = [ctypes.c_void_p]
    hip.zkp_host_free.argtypes = [ctypes.c_void_p]
    hip.zkp_host_is_pinned.argtypes = [ctypes.c_void_p]
    node = hip.zkp_host_numa_node(0)
    assert node >= -1
    p = ctypes.c_void_p()
    assert hip.zkp_host_alloc_on(ctypes.byref(p), 8 << 20, 0) == 0 and p.value
    assert hip.zkp_host_is_pinned(p) == 1
    buf = (ctypes.c_uint8 * (8 << 20)).from_address(p.value)
    ctypes.memset(p, 0x5a, 8 << 20)                                   # every page touched
    assert buf[0] == 0x5a and buf[(8 << 20) - 1] == 0x5a
    where = hip.zkp_host_node_of(p)
    print("GPU 0 hangs off NUMA node %d; the ring's first page is on node %d" % (node, where))
    if node >= 0 and where >= 0:
        assert where == node, "the pinned ring did not land on the GPU's node"
    hip.zkp_host_free(p)
    assert hip.zkp_host_alloc_on(ctypes.byref(p), 4096, 1 << 20) == -2      # no such device: ZKP_ERR_ARG


@pytest.mark.parametrize("pinned", [False, True])
def test_submitter_threads_give_the_bytes_of_the_callers_thread(cmz, pinned):
    """a pipe over several entries of the device list carries out its asynchronous submits on one host thread per entry
    (here: GPU 0 listed three times, two contexts each).  Every byte and verdict equals the single-thread pipe's; errors of the submit
    itself arrive from wait(); a corrupted proof fails exactly its batch; dropping the pipe with jobs in flight is safe."""
    mod, secrets, inst, common, entropy, w = cmz
    st = mod.statement
    n, K = 96, 5
    nn = n * K
    assert len(secrets) >= nn
    mk = T.pinned_copy if pinned else np.ascontiguousarray
    a_sec, a_inst, a_com, a_ent = mk(secrets[:nn]), mk(inst[:, :nn]), mk(common), mk(entropy[:nn])
    ref = T.Pipe((0,), 2)
    chal0, resp0, coms0 = ref.submit_prove(st, nn, _t0(), a_sec, a_inst, a_com, a_ent).wait()
    ref.close()
    with T.Pipe((0, 0, 0), 2) as pipe:
        assert pipe.num_contexts == 6
        jobs = [pipe.submit_prove(st, nn, _t0(), a_sec, a_inst, a_com, a_ent) for _ in range(6)]
        with pytest.raises(BlockingIOError):
            pipe.submit_prove(st, nn, _t0(), a_sec, a_inst, a_com, a_ent)            # every context reserved
        assert sorted(j.context for j in jobs) == list(range(6))
        outs = [j.wait() for j in reversed(jobs)]                                    # retired in another order than submitted
        for chal, resp, coms in outs:
            assert (chal == chal0).all() and (resp == resp0).all() and (coms == coms0).all()
        assert pipe.jobs_in_flight == 0
        bad = resp0.copy()
        bad[2 * n + 5, 1, 0] ^= 1
        vj = [pipe.submit_batch_verify_many(st, K, n, _t0(), a_inst, a_com, coms0, r) for r in (resp0, bad, resp0)]
        cj = pipe.submit_verify_compact(st, nn, _t0(), a_inst, a_com, chal0, bad)
        ej = pipe.submit_verify_batchable_each(st, nn, _t0(), a_inst, a_com, coms0, resp0)
        while not all(j.done() for j in vj):                                         # done() never blocks and never touches a context
            pass
        assert [j.wait()[0].tolist() for j in vj] == [[0] * K, [0, 0, 1, 0, 0], [0] * K]
        res = cj.wait()[0]
        assert res[2 * n + 5] == 1 and res.sum() == 1 and not ej.wait()[0].any()
        # an error the submit itself finds (a stride smaller than the proof count) arrives from wait(), the context is free again afterwards
        j = pipe.submit_prove(st, nn, _t0(), a_sec, a_inst, a_com, a_ent, inst_stride=nn - 1)
        with pytest.raises(Exception):
            j.wait()
        assert pipe.jobs_in_flight == 0
        # small batches run inside the submit (host-transcript route) -- on the device's thread now
        small = pipe.submit_prove(st, 5, _t0(5), a_sec[:5], np.ascontiguousarray(a_inst[:, :5]), a_com, a_ent[:5]).wait()
        assert (small[0] == chal0[:5]).all() and (small[1] == resp0[:5]).all()
        # the synchronous sharded calls still work next to the threads
        c2, r2, k2 = pipe.prove_batch(st, _t0(nn), a_sec, a_inst, a_com, a_ent)
        assert (c2 == chal0).all() and (r2 == resp0).all() and (k2 == coms0).all()
        # jobs left in flight when the pipe goes away
        pipe.submit_prove(st, nn, _t0(), a_sec, a_inst, a_com, a_ent)
        pipe.submit_batch_verify_many(st, K, n, _t0(), a_inst, a_com, coms0, resp0)
    # the same pipe shape on the caller's thread: identical results
    with T.Pipe((0, 0, 0), 2) as pipe:
        pipe.set_submit_threads(0)
        chal, resp, coms = pipe.submit_prove(st, nn, _t0(), a_sec, a_inst, a_com, a_ent).wait()
        assert (chal == chal0).all() and (resp == resp0).all() and (coms == coms0).all()


@pytest.mark.parametrize("threads", [0, 1])
def test_destroying_a_pipe_discards_abandoned_jobs(cmz, threads):
    """zkp_pipe_destroy under jobs nobody waited for (a forgotten handle in Rust, a crashed caller): kernels are waited for, queued submits dropped, no copy out is
    issued -- the outputs are either untouched (discarded) or, with submitter threads, complete (the device's thread had retired the job already); never half written,
    and nothing is written after destroy returns."""
    mod, secrets, inst, common, entropy, w = cmz
    st = mod.statement
    n = 480
    inst_n = np.ascontiguousarray(inst[:, :n])
    ref = T.Pipe((0,), 1)
    want = ref.submit_prove(st, n, _t0(), secrets[:n], inst_n, common, entropy[:n]).wait()
    ref.close()
    pipe = T.Pipe((0, 0), 2)
    pipe.set_submit_threads(threads)
    jobs = [pipe.submit_prove(st, n, _t0(), secrets[:n], inst_n, common, entropy[:n]) for _ in range(4)]
    handles = [(j._h, j.outputs, j._keep) for j in jobs]                 # keep every array alive ourselves ...
    for j in jobs:
        j._h = None                                                      # ... and abandon the handles: nobody will wait
    pipe._jobs.clear()
    T.lib().zkp_pipe_destroy(pipe._h)
    pipe._h = None
    snap = [[a.copy() for a in outs] for _, outs, _ in handles]
    for (h, outs, keep), before in zip(handles, snap):
        untouched = all(not a.any() for a in outs)
        complete = all((a == b).all() for a, b in zip(outs, want))
        assert untouched or complete
        assert all((a == b).all() for a, b in zip(outs, before))
    if not threads:
        assert all(not a.any() for _, outs, _ in handles for a in outs)  # the caller's thread never issued a copy out
