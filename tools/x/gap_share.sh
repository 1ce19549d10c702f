#!/bin/bash
# How much of a lone call chain is idle time BETWEEN kernels?  rocprofv3 kernel trace of a one-stream run (graph replay), per call: wall = last end - first start, busy = sum of durations.
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
for K in 1 5; do
  rm -rf gpurun_out/gap_prof
  rocprofv3 --kernel-trace --output-format csv -d gpurun_out/gap_prof -o kt -- python bench.py --config 2 --steps 20 --warmup 2 --batches-per-call $K --streams 1 --max-hw-queues 1 --no-cpu-baseline --no-flow-lines > gpurun_out/gap_k$K.log 2>&1
  python - <<PY
import csv, glob
f = glob.glob("gpurun_out/gap_prof/**/kt_kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
ev = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows]
# the timed loop = the last 20 // K calls; a call starts with the classifier of the prove flow
K = $K
starts = [i for i, e in enumerate(ev) if "k_stmt_classify" in e[2]]
calls = 20 // K
sel = starts[-calls:]
tot_wall = tot_busy = n = 0
gaps = []
for a, b in zip(sel, sel[1:] + [len(ev)]):
    seg = [e for e in ev[a:b] if not e[2].startswith("at::") ]
    seg = ev[a:b]
    wall = seg[-1][1] - seg[0][0]
    busy = sum(e[1] - e[0] for e in seg)
    tot_wall += wall; tot_busy += busy; n += len(seg)
    gaps += [max(0, seg[i + 1][0] - seg[i][1]) for i in range(len(seg) - 1)]
gaps.sort()
print("K = %d: %d calls, %.1f kernels per call, wall %.3f ms per call, kernels %.3f ms (%.1f %%), gaps %.3f ms; median gap %.1f us, p90 %.1f us" % (
    K, len(sel), n / len(sel), tot_wall / len(sel) / 1e6, tot_busy / len(sel) / 1e6, 100.0 * tot_busy / tot_wall, (tot_wall - tot_busy) / len(sel) / 1e6,
    gaps[len(gaps) // 2] / 1e3, gaps[int(len(gaps) * 0.9)] / 1e3))
PY
done
rm -rf gpurun_out/gap_prof
