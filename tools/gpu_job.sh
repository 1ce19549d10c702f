#!/bin/bash
set -u
export TMPDIR=/tmp
O=gpurun_out
TAG=${1:-r02}
mkdir -p $O
timeout 1200 python -m pytest tests -m gpu -x -q > $O/${TAG}_pytest.txt 2>&1; echo "pytest rc=$?" >> $O/${TAG}_pytest.txt
grep -E "passed|failed|rc=|Error|assert" $O/${TAG}_pytest.txt | tail -8
run() { # name, args...
  n=$1; shift
  t0=$SECONDS; timeout 600 python bench.py "$@" > $O/${TAG}_bench_$n.json 2> $O/${TAG}_bench_$n.err; echo "$n wall $((SECONDS-t0)) s rc=$?"; grep -v amdgpu.ids $O/${TAG}_bench_$n.err | tail -2 | cut -c1-300
  python - <<PY
import json
try:
    j=json.loads(open("$O/${TAG}_bench_$n.json").read().strip().splitlines()[-1])
    print("$n", "%.0f"%j["value"], "ms/step %.4f"%j["ms_per_step"], "streams", j["config"]["streams"], "single", {k:round(v) for k,v in j["single_stream_proofs_per_s"].items()}, "prove sort", round(j["kernel_ms"]["prove"]["sort"],3))
except Exception as e: print("$n","failed",e)
PY
}
A="--no-cpu-baseline --no-flow-lines"
for i in 1 2 3; do
run stmt_$i $A
run generic_$i $A --engine-opt 10=0
done
run stmt20 $A --steps 20 --warmup 5
run generic20 $A --steps 20 --warmup 5 --engine-opt 10=0
run stmt_c4 $A --config 4share
run generic_c4 $A --config 4share --engine-opt 10=0
