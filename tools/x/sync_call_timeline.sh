#!/bin/bash
# host + device timeline of ONE synchronous prove and ONE synchronous batch-verify call of 4096 CMZ proofs on ordinary host buffers (hip api + kernels + copies)
export TMPDIR=/tmp
O=gpurun_out; mkdir -p $O
cat > /tmp/sync_once.py <<'PY'
import sys, time, numpy as np
sys.path.insert(0, ".")
from zkp_amd.engine import Engine
from zkp_amd import toolbox as T
from tests.test_gpu_toolbox import _cmz_batch
n = 4096
eng = Engine(0)
mod, secrets, inst, common = _cmz_batch(n, 11)
ent = np.random.default_rng(1).integers(0, 256, size=(n, 32), dtype=np.uint8)
T.set_fused_min_batch(0)
for rep in range(6):
    ts = np.stack([T.Transcript(b"Benchmark").state] * n)
    t0 = time.perf_counter(); chal, resp, coms = T.prove_batch(eng, mod.statement, ts, secrets, inst, common, ent); t1 = time.perf_counter()
    ts = np.stack([T.Transcript(b"Benchmark").state] * n)
    t2 = time.perf_counter(); T.batch_verify(eng, mod.statement, ts, inst, common, coms, resp); t3 = time.perf_counter()
    print("# rep", rep, "zkp_prove_batch %.3f ms   zkp_batch_verify %.3f ms (wall, under the tracer)" % ((t1 - t0) * 1e3, (t3 - t2) * 1e3))
PY
rocprofv3 --hip-trace --kernel-trace --memory-copy-trace --output-format csv -d $O/r06_sync_prof -o s -- python /tmp/sync_once.py 2>/dev/null | grep "^# rep"
python - <<'PY'
import csv, glob
d = "gpurun_out/r06_sync_prof/"
ev = []
for f in glob.glob(d + "*hip_api_trace.csv"):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "api", r["Function"]))
for f in glob.glob(d + "*kernel_trace.csv"):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "KERNEL", r["Kernel_Name"].split("(")[0].replace("void ", "")[:50]))
for f in glob.glob(d + "*memory_copy_trace.csv"):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "COPY", r.get("Direction", "")))
ev.sort()
ks = [i for i, e in enumerate(ev) if e[2] == "KERNEL" and "k_stmt_classify" in e[3]]
t_k = ev[ks[-1]][0]
win = [e for e in ev if e[0] >= t_k - 300_000 and e[0] <= t_k + 3_300_000]
t0 = win[0][0]
print("# the last prove call and the batch verification behind it: start -> end (us), duration; api rows = calls of the submitting thread that took >= 3 us, copies, synchronisations")
for s, e, kind, name in win:
    if kind == "api" and (e - s) < 3000 and not name.startswith("hipMemcpy") and "Synchronize" not in name:
        continue
    print("%9.1f -> %9.1f %8.1f  %-6s %s" % ((s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3, kind, name))
PY
rm -rf $O/r06_sync_prof
