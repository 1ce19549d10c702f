#!/bin/bash
# Round 5, fourth GPU call: pipe changes (submitter threads, NUMA rings, fail-closed jobs) -- tests, soak, host scaling; then the driver's own command.
export TMPDIR=/tmp
R="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$R"
O=gpurun_out
mkdir -p $O
echo "== 1. pipe / job / thread tests"
python -m pytest tests/test_gpu_pipe.py tests/test_gpu_threads.py tests/test_gpu_sharding.py -m gpu -x -q -s 2>&1 | grep -v "^$" | tail -12
echo "== 2. soak of the pipe (device lists of several entries run the submitter threads)"
timeout 200 python tools/pipe_soak.py 60 777 2>&1 | tail -3
echo "== 3. host scaling"
lscpu | grep -E "^CPU\(s\)|Model name|NUMA node|Socket" | head -8
nproc
timeout 900 python tools/pipe_host_scaling.py > $O/r05_pipe_host_scaling.txt 2>&1; tail -25 $O/r05_pipe_host_scaling.txt
echo "== 4. the driver's command"
time python bench.py --steps 20 --warmup 5 > $O/r05_bench_steps20_pre.json 2> $O/r05_bench_steps20_pre.err
tail -3 $O/r05_bench_steps20_pre.err
python - <<'PY'
import json
j = json.loads(open("gpurun_out/r05_bench_steps20_pre.json").read().strip().splitlines()[-1])
print("value %.3f M  ms/step %.3f  ct_schedule %s" % (j["value"] / 1e6, j["ms_per_step"], j["config"]["ct_schedule"][:60]))
print("sustained", {k: v for k, v in j.get("sustained", {}).items() if k != "note"})
print("value_ct_by_construction", j.get("value_ct_by_construction"), "ct", j.get("ct"))
e = j.get("e2e_host_buffers", {})
print("e2e", round(e.get("proofs_per_s", 0)), round(e.get("pipelined", {}).get("proofs_per_s", 0)), round(e.get("pipelined_staged", {}).get("proofs_per_s", 0)), round(e.get("threads", {}).get("proofs_per_s", 0)))
print("cpu", j.get("cpu_baseline", {}).get("value"), "parity", j.get("parity_checked", {}).get("equal"), j.get("parity_checked", {}).get("proofs"))
print("flows", {k: round(v) for k, v in j["pipelined_proofs_per_s"].items()})
PY
