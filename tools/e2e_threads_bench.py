#!/usr/bin/env python3
"""Throughput of PLAIN SYNCHRONOUS toolbox calls issued by several host threads, each with its own context (what a service with a thread pool does
without touching zkp_pipe): bench.e2e_threads for a few shapes.
    python tools/e2e_threads_bench.py [--threads 1,2,4,6,8] [--k 1,5,10] [--mem fresh|pageable|pinned] [--opt 14=1]"""
import argparse, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
import torch  # noqa: E402,F401  (torch's HIP runtime first, as in bench.py)
torch.cuda.init()
import bench  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=4096)
ap.add_argument("--threads", default="1,2,4,6,8")
ap.add_argument("--k", default="1,5,10")
ap.add_argument("--jobs", type=int, default=24)
ap.add_argument("--mem", default="pageable", choices=["fresh", "pageable", "pinned"], help="fresh = the Python wrappers (outputs allocated per call); pageable / pinned = "
                "buffers of that kind allocated once per thread and passed to the C ABI directly")
ap.add_argument("--opt", action="append", default=[], metavar="ID=VALUE", help="zkp_ctx_set_option on every context (e.g. 14=1: ZKP_OPT_SYNC_SCHEDULE = throughput)")
a = ap.parse_args()
opts = tuple((int(kv.split("=")[0]), int(kv.split("=")[1])) for kv in a.opt)
for K in [int(x) for x in a.k.split(",")]:
    for nt in [int(x) for x in a.threads.split(",")]:
        r = bench.e2e_threads(n=a.n, K=K, threads=nt, jobs=a.jobs, mem=a.mem, opts=opts)
        print(json.dumps({k: (round(v, 3) if isinstance(v, float) else v) for k, v in r.items() if k != "note"}), flush=True)
