"""Randomised soak of the round-3 code paths on the GPU (run by hand: python tools/soak.py SECONDS):
 - zkp_msm_optional with skewed digit distributions (huge buckets, block-wide merge) against the discrete-log oracle;
 - verify_batchable per proof, fused route (window-split Straus walk) against the host-transcript route, with random tampering;
 - K-batch verification against per-batch verdicts."""
import os, sys, time, random
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import model as M
from zkp_amd.engine import Engine
from zkp_amd import toolbox as T
import bench


def sc(x):
    return np.frombuffer((x % (1 << 256)).to_bytes(32, "little"), np.uint8)


def main():
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else int(time.time())
    rng = random.Random(seed)
    nrng = np.random.default_rng(seed)
    print("seed", seed)
    eng = Engine(0)
    logs = [rng.randrange(1, M.L) for _ in range(64)]
    encs = [M.ristretto_encode(M.pt_mul(k, M.BASEPOINT)) for k in logs]
    enc_np = np.frombuffer(b"".join(encs), np.uint8).reshape(-1, 32)
    mod = T.cmz_module(10)
    st = mod.statement
    t_end = time.time() + budget
    it = 0
    while time.time() < t_end:
        it += 1
        # 1. skewed MSM
        n = rng.choice([200, 3000, 5000, 9000, 40000, 120000, 300000])
        kind = rng.choice(["ones", "weights", "two", "few-digits"])
        idx = nrng.integers(0, 64, size=n)
        if kind == "ones":
            vals = [rng.randrange(1, 9)] * n
        elif kind == "weights":
            frac = rng.random()
            vals = [(M.L - rng.randrange(1 << 128)) if rng.random() < frac else rng.randrange(M.L) for _ in range(n)]
        elif kind == "two":
            a, b = rng.randrange(M.L), rng.randrange(M.L)
            vals = [a if rng.random() < 0.9 else b for _ in range(n)]
        else:
            pool = [rng.randrange(M.L) for _ in range(rng.choice([3, 17, 200]))]
            vals = [rng.choice(pool) for _ in range(n)]
        got = eng.msm_optional(np.stack([sc(v) for v in vals]), np.ascontiguousarray(enc_np[idx]))
        dlog = sum(v * logs[j] for v, j in zip(vals, idx.tolist())) % M.L
        assert got == M.ristretto_encode(M.pt_mul(dlog, M.BASEPOINT)), ("msm", n, kind)
        # 2. verify_batchable per proof: fused vs host route, random tampering
        N = rng.choice([1, 7, 64, 300, 1000, 4096])
        secrets, inst, common = bench.make_instance(eng, bench.cmz_statement(), N, nrng)
        eng.prepare_fixed_points(common)
        fresh = lambda: np.stack([T.Transcript(b"soak").state] * N)
        chal, resp, coms = T.prove_batch(eng, st, fresh(), secrets, inst, common, nrng.integers(0, 256, size=(N, 32), dtype=np.uint8))
        resp, coms, inst2 = resp.copy(), coms.copy(), inst.copy()
        for _ in range(rng.randrange(0, 4)):
            j = rng.randrange(N)
            what = rng.randrange(4)
            if what == 0:
                resp[j, rng.randrange(resp.shape[1]), rng.randrange(31)] ^= 1 << rng.randrange(8)
            elif what == 1:
                coms[j, rng.randrange(coms.shape[1])] = coms[rng.randrange(N), rng.randrange(coms.shape[1])]
            elif what == 2:
                inst2[rng.randrange(inst2.shape[0]), j] = inst2[rng.randrange(inst2.shape[0]), rng.randrange(N)]
            else:
                coms[j, rng.randrange(coms.shape[1])] = 0
        w = nrng.integers(0, 256, size=(N, st.nc, 16), dtype=np.uint8)
        res = {}
        for route, thr in (("host", 1 << 30), ("fused", 0)):
            T.set_fused_min_batch(thr)
            res[route] = T.verify_batchable_each(eng, st, fresh(), inst2, common, coms, resp, w)
        T.set_fused_min_batch(32)
        assert (res["host"] == res["fused"]).all(), ("each", N, np.nonzero(res["host"] != res["fused"])[0][:5])
        # 3. K batches in one call: verdict k == any bad proof in batch k
        if N >= 64:
            K = rng.choice([2, 4, 8])
            ne = N // K
            n_use = ne * K
            T.set_fused_min_batch(0)
            verdicts = T.batch_verify_many(eng, st, K, fresh()[:n_use], np.ascontiguousarray(inst2[:, :n_use]), common, coms[:n_use], resp[:n_use])
            T.set_fused_min_batch(32)
            bad = res["host"][:n_use].reshape(K, ne).any(axis=1)
            assert [bool(v) for v in verdicts] == [bool(b) for b in bad], ("many", N, K, verdicts, bad)
    eng.close()
    print("soak ok:", it, "iterations")


if __name__ == "__main__":
    main()
