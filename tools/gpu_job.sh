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
A="--no-cpu-baseline --no-flow-lines"
for i in 1 2 3; do
run base_20_$i $A --steps 20 --warmup 5
run ovl_20_$i $A --steps 20 --warmup 5 --engine-opt 5=1
done
for i in 1 2; do
run base_1000_$i $A --steps 1000
run ovl_1000_$i $A --steps 1000 --engine-opt 5=1
done
