#!/bin/bash
# scratch job script for gpurun (edited per call)
set -u
export TMPDIR=/tmp
O=gpurun_out
TAG=${1:-r02}
mkdir -p $O
run() { # name, args...
  n=$1; shift
  t0=$SECONDS; timeout 600 python bench.py "$@" > $O/${TAG}_bench_$n.json 2> $O/${TAG}_bench_$n.err; echo "$n wall $((SECONDS-t0)) s rc=$?"; grep -v amdgpu.ids $O/${TAG}_bench_$n.err | tail -2 | cut -c1-300
  python - <<PY
import json
try:
    j=json.loads(open("$O/${TAG}_bench_$n.json").read().strip().splitlines()[-1])
    print("$n", "%.0f"%j["value"], "ms/step %.4f"%j["ms_per_step"], "streams", j["config"]["streams"], "pipelined", {k:round(v) for k,v in j["pipelined_proofs_per_s"].items()}, "single", {k:round(v) for k,v in j["single_stream_proofs_per_s"].items()})
    print("   prove ms", {k:round(v,3) for k,v in j["kernel_ms"]["prove"].items() if v})
except Exception as e: print("$n","failed",e)
PY
}
run new_a --no-cpu-baseline --no-flow-lines
run new_20 --no-cpu-baseline --no-flow-lines --steps 20 --warmup 5
run new_c4 --config 4share --no-cpu-baseline --no-flow-lines
cp zkp_amd/libzkp_mi355x.so /tmp/new.so; cp tools/ab/prev.so zkp_amd/libzkp_mi355x.so
run prev_a --no-cpu-baseline --no-flow-lines
run prev_20 --no-cpu-baseline --no-flow-lines --steps 20 --warmup 5
run prev_c4 --config 4share --no-cpu-baseline --no-flow-lines
run prev_b --no-cpu-baseline --no-flow-lines
cp /tmp/new.so zkp_amd/libzkp_mi355x.so
run new_b --no-cpu-baseline --no-flow-lines
timeout 600 python -m pytest tests -m gpu -x -q > $O/${TAG}_pytest.txt 2>&1; echo "pytest rc=$?" >> $O/${TAG}_pytest.txt
grep -E "passed|failed|rc=|Error|assert" $O/${TAG}_pytest.txt | tail -8
