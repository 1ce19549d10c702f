#!/bin/bash
# scratch job script for gpurun (edited per call)
set -u
export TMPDIR=/tmp
O=gpurun_out
TAG=${1:-r02f}
mkdir -p $O
timeout 1200 python -m pytest tests -m gpu -x -q --durations=0 > $O/${TAG}_pytest.log 2>&1; echo "pytest rc=$?" >> $O/${TAG}_pytest.log
grep -E "passed|failed|rc=|Error" $O/${TAG}_pytest.log | tail -5
bash tools/collect_profiles.sh $TAG > $O/${TAG}_collect.log 2>&1; tail -12 $O/${TAG}_collect.log | cut -c1-250
cp $O/${TAG}_pmc_counters.json profiles/r02_pmc_counters.json   # so that the bench runs below find counters keyed to these sources
run() { # name, args...
  n=$1; shift
  t0=$SECONDS; timeout 600 python bench.py "$@" > $O/${TAG}_$n.json 2> $O/${TAG}_$n.err; echo "$n wall $((SECONDS-t0)) s rc=$?"; tail -2 $O/${TAG}_$n.err | cut -c1-300
  python - <<PY
import json
try:
    j=json.loads(open("$O/${TAG}_$n.json").read().strip().splitlines()[-1])
    print("$n", "%.0f"%j["value"], "ms/step %.4f"%j["ms_per_step"], "streams", j["config"]["streams"], "pipelined", {k:round(v) for k,v in j["pipelined_proofs_per_s"].items()}, "single", {k:round(v) for k,v in j["single_stream_proofs_per_s"].items()})
    print("   roofline:", j["roofline"]["kernel"], "frac %.5f"%j["roofline"]["frac"], "traffic", j["roofline"]["traffic"], "| step_valu", (j.get("step_valu") or {}).get("frac"), "| cpu", (j.get("cpu_baseline") or {}).get("value"))
    for f,d in j["kernel_ms"].items(): print("   ", f, {k:round(v,3) for k,v in d.items() if v})
except Exception as e: print("$n","failed",e)
PY
}
run d20 --steps 20 --warmup 5
run d200
run c3 --config 3
run c4 --config 4share
run c5 --config 5share
