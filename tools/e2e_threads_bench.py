#!/usr/bin/env python3
"""Throughput of PLAIN SYNCHRONOUS toolbox calls issued by several host threads, each with its own context (what a Rust service with a thread pool would
do without touching zkp_pipe): zkp_prove_batch + zkp_batch_verify_many on host buffers, OS entropy and weights inside the calls.
    python tools/e2e_threads_bench.py [--threads 1,2,4,6,8] [--k 1,5,10] [--jobs 24]"""
import argparse, json, os, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
import numpy as np
import torch  # noqa: F401  (torch's HIP runtime first, as in bench.py)
torch.cuda.init()
import bench
from zkp_amd import toolbox as T
from zkp_amd.engine import Engine

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=4096)
ap.add_argument("--threads", default="1,2,4,6,8")
ap.add_argument("--k", default="1,5,10")
ap.add_argument("--jobs", type=int, default=24)
a = ap.parse_args()
mod = T.cmz_module(10)
st = mod.statement
t0s = T.Transcript(bench.LABEL).state
for K in [int(x) for x in a.k.split(",")]:
    nn = a.n * K
    e0 = Engine(0)
    secrets, inst, common = bench.make_instance(e0, bench.cmz_statement(), nn, np.random.default_rng(78))
    e0.close()
    ts0 = np.stack([t0s] * nn)
    for nt in [int(x) for x in a.threads.split(",")]:
        engines = [Engine(0) for _ in range(nt)]
        todo = {"left": 0}
        lock = threading.Lock()
        errors = []

        def worker(e):
            try:
                while True:
                    with lock:
                        if todo["left"] <= 0:
                            return
                        todo["left"] -= 1
                    ts = ts0.copy()
                    chal, resp, coms = T.prove_batch(e, st, ts, secrets, inst, common)
                    ts = ts0.copy()
                    v = T.batch_verify_many(e, st, K, ts, inst, common, coms, resp)
                    if v.any():
                        raise AssertionError("a batch of fresh proofs did not verify")
            except Exception as ex:          # noqa: BLE001
                errors.append(ex)

        def run(jobs):
            todo["left"] = jobs
            th = [threading.Thread(target=worker, args=(e,)) for e in engines]
            t0 = time.perf_counter()
            for t in th:
                t.start()
            for t in th:
                t.join()
            return time.perf_counter() - t0

        run(2 * nt)                                     # plans, workspaces, tables
        el = run(a.jobs)
        for e in engines:
            e.close()
        if errors:
            raise errors[0]
        print(json.dumps({"threads": nt, "batches_per_call": K, "proofs_per_call": nn, "jobs": a.jobs, "proofs_per_s": round(a.jobs * nn / el), "ms_per_job": round(el / a.jobs * 1e3, 3)}), flush=True)
