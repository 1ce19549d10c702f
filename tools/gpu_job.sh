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
    print("$n", "%.0f"%j["value"], "ms/step %.4f"%j["ms_per_step"], "streams", j["config"]["streams"], "single", {k:round(v) for k,v in j["single_stream_proofs_per_s"].items()})
except Exception as e: print("$n","failed",e)
PY
}
A="--no-cpu-baseline --no-flow-lines"
cp zkp_amd/libzkp_mi355x.so /tmp/new.so
for i in 1 2 3; do
cp /tmp/new.so zkp_amd/libzkp_mi355x.so
run new_$i $A
cp tools/ab/prev3.so zkp_amd/libzkp_mi355x.so
run prev_$i $A
done
cp /tmp/new.so zkp_amd/libzkp_mi355x.so
run new20 $A --steps 20 --warmup 5
