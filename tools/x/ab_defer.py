#!/usr/bin/env python3
"""A/B of ZKP_OPT_JOB_DEFER_D2H inside ONE process (boxes differ by 20 %): alternate the setting, several repetitions per shape."""
import json, os, sys, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import torch
torch.cuda.init()
import bench
res = {}
bench.e2e_pipelined(n=4096, K=5, contexts=6, jobs=24, pinned=True)     # warm the process
for rep in range(4):
    for pinned in (True, False):
        for C in (6, 8):
            for d in (0, 1):
                os.environ["ZKP_X_DEFER"] = str(d)
                r = bench.e2e_pipelined(n=4096, K=5, contexts=C, jobs=72, pinned=pinned)
                res.setdefault((pinned, C, d), []).append(r["proofs_per_s"] / 1e6)
for k, v in sorted(res.items()):
    print("pinned=%s contexts=%d defer=%d  M proofs/s: %s  median %.2f" % (k[0], k[1], k[2], " ".join("%.2f" % x for x in v), statistics.median(v)))
