#!/usr/bin/env python3
"""Where the zkp_pipe loses against the resident loop: summary of a `rocprofv3 --hip-trace --kernel-trace --memory-copy-trace` of
tools/e2e_pipe_bench.py (host buffers, asynchronous jobs) next to one of bench.py's resident loop.

    python tools/pipe_gap_trace.py DIR_PIPE DIR_RESIDENT      (directories with *_kernel_trace.csv, *_memory_copy_trace.csv, *_hip_api_trace.csv)

Prints, for the busiest window of each trace: wall time, summed kernel time by kernel group (the flow's own kernels, ChaCha20 entropy / weights, copy kernels),
the share of the window in which NO kernel ran, copy-engine busy time per direction, host time inside the HIP calls of the submitting thread."""
import csv, glob, sys


def load(d):
    ks, cs, api = [], [], []
    for f in glob.glob(d + "/*kernel_trace.csv"):
        for r in csv.DictReader(open(f)):
            ks.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", "")))
    for f in glob.glob(d + "/*memory_copy_trace.csv"):
        for r in csv.DictReader(open(f)):
            cs.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Direction", "")))
    for f in glob.glob(d + "/*hip_api_trace.csv"):
        for r in csv.DictReader(open(f)):
            api.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Function"], r.get("Thread_Id", "")))
    return sorted(ks), sorted(cs), sorted(api)


def union(iv):
    tot, cur_s, cur_e = 0, None, None
    for s, e in sorted(iv):
        if cur_e is None or s > cur_e:
            if cur_e is not None:
                tot += cur_e - cur_s
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    if cur_e is not None:
        tot += cur_e - cur_s
    return tot


def summarise(name, d, proofs_per_unit, last=None):
    ks, cs, api = load(d)
    # the measured window: from the first to the last k_terms_split<true (prover term kernel) of the LAST 60 % of the launches (warm-up jobs dropped), or of
    # the last `last` launches (bench.py: the timed calls are the last prove calls of the process)
    tk = [k for k in ks if k[2].startswith("k_terms_split<true")]
    tk = tk[-last:] if last else tk[int(len(tk) * 0.4):]
    t0, t1 = tk[0][0], tk[-1][1]
    win = [k for k in ks if k[0] >= t0 and k[1] <= t1]
    wall = (t1 - t0) / 1e6
    groups = {}
    for s, e, n in win:
        g = "chacha20 (entropy / weights on the device)" if "chacha" in n else ("copy / fill kernels" if "rocclr" in n or n.startswith("at::") else ("k_broadcast_transcript / any_nonzero (job plumbing)" if ("broadcast" in n or "any_nonzero" in n) else "flow kernels"))
        groups[g] = groups.get(g, 0) + (e - s)
    busy = union([(s, e) for s, e, _ in win]) / 1e6
    print("== %s: window %.2f ms, %d prover term-kernel launches (%d proofs each) -> %.2f M proofs/s" % (name, wall, len(tk), proofs_per_unit, len(tk) * proofs_per_unit / wall / 1e3))
    print("   some kernel running: %.1f %% of the window; summed kernel time / window = %.2f (average kernels in flight)" % (100 * busy / wall, sum(groups.values()) / 1e6 / wall))
    for g, v in sorted(groups.items(), key=lambda x: -x[1]):
        print("   %-58s %8.2f ms summed  (%.1f %% of all kernel time)" % (g, v / 1e6, 100.0 * v / sum(groups.values())))
    flow = {}
    for s, e, n in win:
        flow[n] = flow.get(n, 0) + (e - s)
    top = sorted(flow.items(), key=lambda x: -x[1])[:8]
    print("   per term-kernel launch: " + ", ".join("%s %.0f us" % (n[:28], v / 1e3 / len(tk)) for n, v in top))
    for direction in sorted({c[2] for c in cs}):
        cw = [(s, e) for s, e, dd in cs if dd == direction and s >= t0 and e <= t1]
        if cw:
            print("   copies %-28s %5d, engine busy %.1f %% of the window" % (direction, len(cw), 100 * union(cw) / 1e6 / wall))
    threads = {}
    for s, e, fn, th in api:
        if s >= t0 and e <= t1:
            threads.setdefault(th, {}).setdefault(fn, [0, 0])
            threads[th][fn][0] += 1
            threads[th][fn][1] += e - s
    for th, fns in threads.items():
        tot = sum(v[1] for v in fns.values()) / 1e6
        print("   host thread %s: %.1f %% of the window inside HIP calls: " % (th, 100 * tot / wall) + ", ".join("%s x%d %.1f ms" % (fn, v[0], v[1] / 1e6) for fn, v in sorted(fns.items(), key=lambda x: -x[1][1])[:6]))


if __name__ == "__main__":
    summarise("zkp_pipe, pinned host buffers, jobs of 10 batches x 4096 proofs on 6 contexts", sys.argv[1], 40960)
    summarise("resident loop (bench.py --steps 200 --batches-per-call 5 --streams 4: HIP graphs, the 40 timed calls)", sys.argv[2], 20480, last=40)
