#!/usr/bin/env python3
"""Throughput through the host-buffer boundary (zkp_pipe) for a few shapes: python tools/e2e_pipe_bench.py [--n 4096] """
import argparse, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import torch  # noqa: E402,F401  (torch's HIP runtime first, as in bench.py)
torch.cuda.init()
import bench  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=4096)
ap.add_argument("--shapes", default="5x3,5x4,5x6,5x8,10x4,10x6,1x8,1x16,20x4")
ap.add_argument("--jobs", type=int, default=24)
ap.add_argument("--order", default="pinned,staged")
a = ap.parse_args()
for shape in a.shapes.split(","):
    K, C = (int(x) for x in shape.split("x"))
    for pinned in [x == "pinned" for x in a.order.split(",")]:
        r = bench.e2e_pipelined(n=a.n, K=K, contexts=C, jobs=a.jobs, pinned=pinned)
        print(json.dumps({k: (round(v, 3) if isinstance(v, float) else v) for k, v in r.items() if k not in ("bytes_per_proof", "devices")}), flush=True)
