"""A/B timing of the per-proof verifiers through the host toolbox (fused route, host buffers in/out, best of 3):
zkp_verify_batchable_each (verifier.rs:123-173; bad-proof localisation) and zkp_verify_compact_batch (verifier.rs:80-120).
    python tools/ab_each.py [N ...]        -> one line per N; device-side kernel time from zkp_ctx_last_timing next to the wall time"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np


def main():
    import bench
    from zkp_amd.engine import Engine
    from zkp_amd import toolbox as T
    opts = [a for a in sys.argv[1:] if "=" in a]                       # ID=VALUE: zkp_ctx_set_option (e.g. 10=0: the round-2 ladders)
    ns = [int(a) for a in sys.argv[1:] if "=" not in a] or [4096, 65536]
    eng = Engine(0)
    for kv in opts:
        eng.set_option(int(kv.split("=")[0]), int(kv.split("=")[1]))
    print("# options:", opts or "defaults")
    mod = T.cmz_module(10)
    st = mod.statement
    rng = np.random.default_rng(3)
    for n in ns:
        secrets, inst, common = bench.make_instance(eng, bench.cmz_statement(), n, rng)
        eng.prepare_fixed_points(common)
        ts = np.stack([T.Transcript(b"ab").state] * n)
        chal, resp, coms = T.prove_batch(eng, st, ts, secrets, inst, common, rng.integers(0, 256, size=(n, 32), dtype=np.uint8))
        w = rng.integers(0, 256, size=(n, st.nc, 16), dtype=np.uint8)
        out = {}
        for name, fn in (("verify_batchable_each", lambda ts: T.verify_batchable_each(eng, st, ts, inst, common, coms, resp, w)),
                         ("verify_compact_batch", lambda ts: T.verify_compact_batch(eng, st, ts, inst, common, chal, resp))):
            best, dev = None, None
            for rep in range(4):
                ts = np.stack([T.Transcript(b"ab").state] * n)
                eng.set_profiling(rep == 3)
                t0 = time.perf_counter()
                res = fn(ts)
                dt = time.perf_counter() - t0
                assert not res.any()
                if rep == 3:
                    km, tot = eng.last_timing()
                    dev = (tot, {k: round(v, 3) for k, v in km.items() if v > 0})
                elif best is None or dt < best:
                    best = dt
            eng.set_profiling(False)
            out[name] = (best, dev)
        for name, (best, dev) in out.items():
            print("N = %6d  %-22s wall %8.3f ms = %6.2f M proofs/s | device %8.3f ms = %6.2f M proofs/s  %s" % (n, name, best * 1e3, n / best / 1e6, dev[0], n / (dev[0] * 1e-3) / 1e6, dev[1]))
    eng.close()


if __name__ == "__main__":
    main()
