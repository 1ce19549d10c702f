#!/usr/bin/env python3
"""Experiment: the RESIDENT flows (_dev entry points, torch buffers) driven like the pipe drives its jobs -- separate prove / verify jobs on C contexts,
refilled by the host when one finishes -- to separate the cost of host-driven refill from the cost of the copies."""
import collections, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import numpy as np
import torch
torch.cuda.init()
import bench
from zkp_amd.engine import Engine, FusedStatement
from zkp_amd import toolbox as T

def run(n=4096, K=5, C=6, jobs=48, copies=False):
    dev = torch.device("cuda", 0)
    st = bench.cmz_statement()
    fst = FusedStatement(bench.LABEL if False else b"cred_show_10", *st) if False else None
    p_st = st
    eng0 = Engine(0)
    nn = n * K
    secrets, inst, common = bench.make_instance(eng0, st, nn, np.random.default_rng(78))
    from bench import WORKLOADS
    label = WORKLOADS["2"][1][0][0]
    fst = FusedStatement(label, *st)
    m, nc, ns, ni = len(st[0]), len(st[2]), len(common), len(inst)
    t0s = T.Transcript(bench.LABEL).state
    pos = int(t0s[200]) | int(t0s[201]) << 8 | int(t0s[202]) << 16
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    z8 = lambda *s: torch.zeros(s, dtype=torch.uint8, device=dev)
    d_ts0 = t(np.stack([t0s] * nn)); d_sec = t(secrets); d_tbl = t(np.concatenate([common, inst.reshape(-1, 32)]))
    rng = np.random.default_rng(1)
    d_ent = t(rng.integers(0, 256, size=(nn, 32), dtype=np.uint8)); d_w = t(rng.integers(0, 256, size=(nc, nn, 16), dtype=np.uint8))
    n_pts = ns + (ni + nc) * nn
    engines, streams, bufs, events = [], [], [], []
    for k in range(C):
        e = Engine(0); s = torch.cuda.Stream(device=dev); e.set_stream(s.cuda_stream); e.prepare_fixed_points(common)
        engines.append(e); streams.append(s); events.append(torch.cuda.Event())
    outsets = collections.deque()
    for _ in range(2 * C + 2):
        b = dict(ts=z8(nn, 208), chal=z8(nn, 32), resp=z8(nn, m, 32), coms=z8(nn, nc, 32), st=z8(nn * nc), pts=z8(n_pts, 32), out=z8(K, 32), bst=torch.ones((K, 2), dtype=torch.int32, device=dev))
        b["pts"][: ns + ni * nn] = d_tbl
        outsets.append(b)
    torch.cuda.synchronize()
    free_ctx = collections.deque(range(C))
    def submit(kind, b):
        k = free_ctx.popleft()
        with torch.cuda.stream(streams[k]):
            b["ts"].copy_(d_ts0, non_blocking=True)
            if kind == "P":
                engines[k].fused_prove_dev(fst, nn, pos, b["ts"].data_ptr(), d_sec.data_ptr(), d_tbl.data_ptr(), d_ent.data_ptr(), b["chal"].data_ptr(), b["resp"].data_ptr(), b["coms"].data_ptr(), b["st"].data_ptr())
            else:
                engines[k].fused_batch_verify_many_dev(fst, K, n, pos, b["ts"].data_ptr(), b["pts"].data_ptr(), b["coms"].data_ptr(), b["resp"].data_ptr(), d_w.data_ptr(), b["out"].data_ptr(), b["bst"].data_ptr())
            ev = torch.cuda.Event(); ev.record(streams[k])
        return k, ev
    def loop(n_jobs):
        pending, to_verify = [], collections.deque(); sub = ver = 0
        while ver < n_jobs:
            if to_verify and free_ctx:
                b = to_verify.popleft(); k, ev = submit("V", b); pending.append(("V", b, k, ev))
            elif sub < n_jobs and free_ctx and outsets:
                b = outsets.popleft(); k, ev = submit("P", b); pending.append(("P", b, k, ev)); sub += 1
            else:
                idx = next((i for i, e in enumerate(pending) if e[3].query()), 0)
                kind, b, k, ev = pending.pop(idx); ev.synchronize(); free_ctx.append(k)
                if kind == "P": to_verify.append(b)
                else: outsets.append(b); ver += 1
    loop(2 * C + 2)
    t0 = time.perf_counter(); loop(jobs); el = time.perf_counter() - t0
    for e in engines: e.close()
    eng0.close()
    return {"resident_refill_proofs_per_s": round(jobs * nn / el), "K": K, "contexts": C}

if __name__ == "__main__":
    for K, C in ((5, 4), (5, 6), (5, 8), (5, 12), (10, 6)):
        print(json.dumps(run(K=K, C=C)), flush=True)
