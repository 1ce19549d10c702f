#!/usr/bin/env python3
"""Host side of an 8-GPU run, measured on ONE GPU: a zkp_pipe over D "devices" x C contexts that are all GPU 0, jobs of
K x 4096 CMZ proofs (prove, then batch-verify what came back), pinned caller buffers and ordinary memory staged through the pipe's rings, with
the asynchronous submits on the caller's thread and on one submitter thread per device.  The GPU is the same one chip in every row, so the
proofs/s column only says that the pipe keeps it busy; what scales with the number of GPUs is the HOST time per job:

    caller_ms_per_job = (time inside submit_* + time inside wait()) / jobs         on the one thread that drives the pipe
    cores for G GPUs at R proofs/s each = caller_ms_per_job x G x R / (proofs per job) / 1000

-> profiles/r05_pipe_host_scaling.txt       (GPU box: python tools/pipe_host_scaling.py [--devices 8] [--contexts 6] [--batches 10] [--jobs 96])"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--devices", type=int, default=8)
    ap.add_argument("--contexts", type=int, default=6)
    ap.add_argument("--batches", type=int, default=10)
    ap.add_argument("--jobs", type=int, default=96)
    ap.add_argument("--rate", type=float, default=6.0e6, help="proofs/s per GPU the host has to feed")
    args = ap.parse_args()
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
    import bench
    n, K = 4096, args.batches
    per_job = n * K
    print("# zkp_pipe host cost per job of %d x %d CMZ proofs (prove + batch verification of the proofs it returned), every context on GPU 0" % (K, n))
    print("# %-34s %-8s %-9s %12s %16s %16s %22s" % ("pipe", "buffers", "submits", "M proofs/s", "submit ms / job", "wait ms / job", "cores for 8 GPUs @ %.0f M" % (args.rate / 1e6)))
    for devs, ctxs in ((1, args.contexts), (args.devices, args.contexts)):
        for pinned in (True, False):
            for threads in ((0,) if devs == 1 else (0, 1)):
                t0 = time.perf_counter()
                r = bench.e2e_pipelined(n=n, K=K, contexts=ctxs, jobs=args.jobs, pinned=pinned, devices=(0,) * devs, submit_threads=threads)
                jobs2 = 2 * args.jobs                     # every prove job is followed by a verification job
                sub, wait = r["host_ms_in_submit"] / jobs2, r["host_ms_in_wait"] / jobs2
                # a prove job + its verification job handle per_job proofs: caller time per proof, times the proofs 8 GPUs need per second
                cores = (r["host_ms_in_submit"] * 1e-3 / args.jobs) * 8 * args.rate / per_job
                print("%-36s %-8s %-9s %12.2f %16.3f %16.3f %22.2f   (%.0f s)" % (
                    "%d device%s x %d contexts" % (devs, "" if devs == 1 else "s", ctxs), "pinned" if pinned else "staged", "threads" if threads else "caller",
                    r["proofs_per_s"] / 1e6, sub, wait, cores, time.perf_counter() - t0), flush=True)
    print("# submit ms / job: wall time the CALLER's thread spends inside submit_* (staging memcpy + ~40 enqueue calls on its own thread, or a queue push with submitter threads);")
    print("# wait ms / job: wall time inside done() / wait() -- mostly blocking on the GPU, which is one chip here and eight in the real run;")
    print("# cores for 8 GPUs: caller-thread seconds of submit work per second of wall clock if every one of 8 GPUs has to be fed %.0f M proofs/s -- above ~1.0 a single" % (args.rate / 1e6))
    print("# submitting thread cannot keep up (the submitter threads then carry that work, one per GPU, next to the staging copies of their own device).")


if __name__ == "__main__":
    main()
