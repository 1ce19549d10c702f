"""Randomised soak of the round-4 host-buffer path on the GPU (run by hand: python tools/pipe_soak.py SECONDS [SEED]):
zkp_pipe with 1 - 8 contexts on GPU 0, random mixes of asynchronous jobs -- prove over a proof range, K-batch verification, verify_compact,
verify_batchable per proof -- from ordinary and pinned buffers, waited for in random order, with random tampering; every output byte and
every verdict is compared with the synchronous single-context calls on the same inputs (which tests/ pin against the oracle)."""
import os, sys, time, random
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from zkp_amd.engine import Engine
from zkp_amd import toolbox as T
import bench

LABEL = b"soak"


def main():
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else int(time.time())
    rng = random.Random(seed)
    nrng = np.random.default_rng(seed)
    print("seed", seed)
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
    eng = Engine(0)
    T.set_host_max_terms(0)
    st = T.cmz_module(10).statement
    t0 = T.Transcript(LABEL).state
    t_end = time.time() + budget
    rounds = jobs_done = 0
    while time.time() < t_end:
        rounds += 1
        N = rng.choice([64, 256, 1024, 4096])
        secrets, inst, common = bench.make_instance(eng, bench.cmz_statement(), N, nrng)
        eng.prepare_fixed_points(common)
        entropy = nrng.integers(0, 256, size=(N, 32), dtype=np.uint8)
        chal, resp, coms = T.prove_batch(eng, st, np.stack([t0] * N), secrets, inst, common, entropy)
        # tampered copies and what each proof deserves (synchronous, single context)
        resp_t, coms_t, chal_t = resp.copy(), coms.copy(), chal.copy()
        for _ in range(rng.randrange(0, 5)):
            j = rng.randrange(N)
            what = rng.randrange(3)
            if what == 0:
                resp_t[j, rng.randrange(st.m), rng.randrange(31)] ^= 1 << rng.randrange(8)
            elif what == 1:
                coms_t[j, rng.randrange(st.nc)] = coms[(j + 1) % N, rng.randrange(st.nc)]
                chal_t[j, rng.randrange(16)] ^= 1
            else:
                coms_t[j, rng.randrange(st.nc)] = 0xff                     # not a ristretto encoding
                chal_t[j, 3] ^= 0x40
        w = nrng.integers(0, 256, size=(st.nc, N, 16), dtype=np.uint8)
        w_each = np.ascontiguousarray(w.transpose(1, 0, 2))
        bad_b = T.verify_batchable_each(eng, st, np.stack([t0] * N), inst, common, coms_t, resp_t, w_each).astype(bool)
        bad_c = T.verify_compact_batch(eng, st, np.stack([t0] * N), inst, common, chal_t, resp_t).astype(bool)
        assert bad_b.sum() <= 4 and bad_c.sum() <= 4
        maybe_pin = lambda a: T.pinned_copy(a) if rng.random() < 0.5 else np.ascontiguousarray(a)
        # the device list: GPU 0 once (submits on the caller's thread) or several times (one submitter thread per entry, round 5), sometimes with the mode forced the other way
        n_dev = rng.choice([1, 1, 2, 3, 4])
        with T.Pipe((0,) * n_dev, rng.choice([1, 2, 3, 6, 8]) if n_dev == 1 else rng.choice([1, 2, 3])) as pipe:
            if rng.random() < 0.25:
                pipe.set_submit_threads(rng.choice([0, 1]))
            pending = []

            def retire(k):
                nonlocal jobs_done
                job, check = pending.pop(k)
                check(job.wait(raise_on_failure=False), job.rc)
                jobs_done += 1

            for _ in range(rng.randrange(4, 24)):
                kind = rng.choice(["prove", "many", "compact", "each"])
                if kind == "many":
                    K = rng.choice([1, 2, 4, 8])
                    ne = rng.choice([x for x in (8, 32, 128, 512) if x * K <= N])
                    a = rng.randrange(0, N - K * ne + 1)
                else:
                    K, ne = 1, rng.randrange(1, min(N, 700) + 1)
                    a = rng.randrange(0, N - ne + 1)
                b = a + K * ne
                sl = slice(a, b)
                inst_r = maybe_pin(inst[:, sl])
                shared = rng.random() < 0.5
                ts = t0 if shared else maybe_pin(np.stack([t0] * (b - a)))
                while True:
                    try:
                        if kind == "prove":
                            job = pipe.submit_prove(st, b - a, ts, maybe_pin(secrets[sl]), inst_r, common, maybe_pin(entropy[sl]))
                            def check(out, rc, sl=sl):
                                assert rc == 0 and (out[0] == chal[sl]).all() and (out[1] == resp[sl]).all() and (out[2] == coms[sl]).all(), ("prove", sl)
                        elif kind == "many":
                            job = pipe.submit_batch_verify_many(st, K, ne, ts, inst_r, common, maybe_pin(coms_t[sl]), maybe_pin(resp_t[sl]), maybe_pin(w[:, sl]))
                            def check(out, rc, sl=sl, K=K, ne=ne):
                                want = bad_b[sl].reshape(K, ne).any(axis=1)
                                assert rc in (0, 1) and [bool(v) for v in out[0]] == [bool(x) for x in want], ("many", sl, K, out[0], want)
                        elif kind == "compact":
                            job = pipe.submit_verify_compact(st, b - a, ts, inst_r, common, maybe_pin(chal_t[sl]), maybe_pin(resp_t[sl]))
                            def check(out, rc, sl=sl):
                                assert rc in (0, 1) and (out[0].astype(bool) == bad_c[sl]).all(), ("compact", sl)
                        else:
                            job = pipe.submit_verify_batchable_each(st, b - a, ts, inst_r, common, maybe_pin(coms_t[sl]), maybe_pin(resp_t[sl]), maybe_pin(w_each[sl]))
                            def check(out, rc, sl=sl):
                                assert rc in (0, 1) and (out[0].astype(bool) == bad_b[sl]).all(), ("each", sl)
                        pending.append((job, check))
                        break
                    except BlockingIOError:                               # every context busy: retire one, any one
                        retire(rng.randrange(len(pending)))
                if pending and rng.random() < 0.3:
                    retire(rng.randrange(len(pending)))
            while pending:
                retire(rng.randrange(len(pending)))
    eng.close()
    print("pipe soak ok:", rounds, "rounds,", jobs_done, "jobs")


if __name__ == "__main__":
    main()
