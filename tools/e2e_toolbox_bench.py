"""End-to-end rates THROUGH the host toolbox (host Merlin transcripts + scalar arithmetic + PCIe copies + GPU),
i.e. what an application calling zkp_prove_batch / zkp_batch_verify sees.  Not the bench.py metric (that one keeps
inputs resident in HBM); reported in DESIGN.md next to it."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from zkp_amd.engine import Engine
from zkp_amd import toolbox as T
from tests.test_gpu_toolbox import _cmz_batch

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
eng = Engine(0)
mod, secrets, inst, common = _cmz_batch(n, 11)
label = b"Benchmark"
entropy = np.random.default_rng(1).integers(0, 256, size=(n, 32), dtype=np.uint8)
for threads in (1, 16, 64, 0):
    best = {}
    for rep in range(3):
        ts = np.stack([T.Transcript(label).state] * n)
        t0 = time.perf_counter(); chal, resp, coms = T.prove_batch(eng, mod.statement, ts, secrets, inst, common, entropy, threads=threads); t1 = time.perf_counter()
        ts = np.stack([T.Transcript(label).state] * n)
        t2 = time.perf_counter(); T.batch_verify(eng, mod.statement, ts, inst, common, coms, resp, threads=threads); t3 = time.perf_counter()
        ts = np.stack([T.Transcript(label).state] * n)
        t4 = time.perf_counter(); res = T.verify_compact_batch(eng, mod.statement, ts, inst, common, chal, resp, threads=threads); t5 = time.perf_counter()
        assert not res.any()
        for k, v in (("prove", t1 - t0), ("batch_verify", t3 - t2), ("verify_compact", t5 - t4)):
            best[k] = min(best.get(k, 1e9), v)
    print("threads %3s | " % (threads or "all") + " | ".join("%s %7.2f ms = %9.0f proofs/s" % (k, v * 1e3, n / v) for k, v in best.items()))
