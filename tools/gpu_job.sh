#!/bin/bash
# scratch job script for gpurun (edited per call)
set -u
export TMPDIR=/tmp
O=gpurun_out
TAG=${1:-r02k}
mkdir -p $O
timeout 1200 python -m pytest tests -m gpu -x -q > $O/${TAG}_pytest.log 2>&1; echo "pytest rc=$?" >> $O/${TAG}_pytest.log
grep -E "passed|failed|rc=|Error" $O/${TAG}_pytest.log | tail -5
run() { # name, args...
  n=$1; shift
  timeout 600 python bench.py --no-cpu-baseline "$@" > $O/${TAG}_bench_$n.json 2> $O/${TAG}_bench_$n.err; grep -v amdgpu.ids $O/${TAG}_bench_$n.err | tail -2 | cut -c1-300
  python - <<PY
import json
try:
    j=json.loads(open("$O/${TAG}_bench_$n.json").read().strip().splitlines()[-1])
    print("$n", "%.0f"%j["value"], "ms/step %.4f"%j["ms_per_step"], "pipelined", {k:round(v) for k,v in j["pipelined_proofs_per_s"].items()}, "prove", {k:round(v,3) for k,v in j["kernel_ms"]["prove"].items() if v})
except Exception as e: print("$n","failed",e)
PY
}
run s20 --steps 20 --warmup 5
run s200
run t20 --steps 20 --warmup 5 --engine-opt 3=1
run t200 --engine-opt 3=1
run s200b
run t200b --engine-opt 3=1
run c4 --config 4share
