"""End-to-end rates THROUGH the host toolbox (host buffers in, host buffers out: PCIe copies included), i.e. what an
application calling zkp_prove_batch / zkp_batch_verify / zkp_verify_compact_batch sees, on both routes:
  host   = Merlin transcripts + scalar arithmetic on the host threads, group arithmetic on the GPU
  fused  = everything on the GPU (zkp_mi355x.h section 2c)
Not the bench.py metric (that one keeps inputs resident in HBM); reported in DESIGN.md next to it.

    python tools/e2e_toolbox_bench.py [N ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from zkp_amd.engine import Engine
from zkp_amd import toolbox as T
from tests.test_gpu_toolbox import _cmz_batch

sizes = [int(a) for a in sys.argv[1:]] or [4096]
eng = Engine(0)
label = b"Benchmark"
for n in sizes:
    mod, secrets, inst, common = _cmz_batch(n, 11)
    entropy = np.random.default_rng(1).integers(0, 256, size=(n, 32), dtype=np.uint8)
    for route, thr, threads in (("host/1 thread", 0xFFFFFFFF, 1), ("host/all threads", 0xFFFFFFFF, 0), ("fused", 0, 0)):
        if n > 16384 and threads == 1:
            continue
        T.set_fused_min_batch(thr)
        best, km = {}, {}
        for rep in range(4):
            eng.set_profiling(rep == 3)
            ts = np.stack([T.Transcript(label).state] * n)
            t0 = time.perf_counter(); chal, resp, coms = T.prove_batch(eng, mod.statement, ts, secrets, inst, common, entropy, threads=threads); t1 = time.perf_counter()
            if rep == 3: km["prove"] = eng.last_timing()
            ts = np.stack([T.Transcript(label).state] * n)
            t2 = time.perf_counter(); T.batch_verify(eng, mod.statement, ts, inst, common, coms, resp, threads=threads); t3 = time.perf_counter()
            if rep == 3: km["batch_verify"] = eng.last_timing()
            ts = np.stack([T.Transcript(label).state] * n)
            t4 = time.perf_counter(); res = T.verify_compact_batch(eng, mod.statement, ts, inst, common, chal, resp, threads=threads); t5 = time.perf_counter()
            if rep == 3: km["verify_compact"] = eng.last_timing()
            assert not res.any()
            if rep < 3:
                for k, v in (("prove", t1 - t0), ("batch_verify", t3 - t2), ("verify_compact", t5 - t4)):
                    best[k] = min(best.get(k, 1e9), v)
        eng.set_profiling(False)
        print("N %7d %-16s | " % (n, route) + " | ".join("%s %8.2f ms = %9.0f proofs/s" % (k, v * 1e3, n / v) for k, v in best.items()))
        if route == "fused":
            for k, (d, tot) in km.items():
                print("           device %-14s %7.3f ms: " % (k, tot) + ", ".join("%s %.3f" % (a, b) for a, b in d.items() if b > 0))
T.set_fused_min_batch(32)
# round 4: the same synchronous calls over a zkp_pipe of C contexts on this GPU (contiguous proof ranges, copies of one range under the
# kernels of the others; on a node: zkp_pipe over the 8 device ids)
for n in sizes:
    if n < 4096:
        continue
    mod, secrets, inst, common = _cmz_batch(n, 11)
    entropy = np.random.default_rng(1).integers(0, 256, size=(n, 32), dtype=np.uint8)
    for C in (2, 4, 8):
        if n // C < 1024:
            continue
        with T.Pipe((0,), C) as pipe:
            best = {}
            for rep in range(4):
                ts = np.stack([T.Transcript(label).state] * n)
                t0 = time.perf_counter(); chal, resp, coms = pipe.prove_batch(mod.statement, ts, secrets, inst, common, entropy); t1 = time.perf_counter()
                ts = np.stack([T.Transcript(label).state] * n)
                t2 = time.perf_counter(); pipe.batch_verify(mod.statement, ts, inst, common, coms, resp); t3 = time.perf_counter()
                ts = np.stack([T.Transcript(label).state] * n)
                t4 = time.perf_counter(); res = pipe.verify_compact_batch(mod.statement, ts, inst, common, chal, resp); t5 = time.perf_counter()
                assert not res.any()
                if rep:
                    for k, v in (("prove", t1 - t0), ("batch_verify", t3 - t2), ("verify_compact", t5 - t4)):
                        best[k] = min(best.get(k, 1e9), v)
        print("N %7d %-16s | " % (n, "pipe x %d ctx" % C) + " | ".join("%s %8.2f ms = %9.0f proofs/s" % (k, v * 1e3, n / v) for k, v in best.items()))
