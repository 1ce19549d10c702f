#!/bin/bash
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
    print("$n", "%.0f"%j["value"], "ms/step %.4f"%j["ms_per_step"], "streams", j["config"]["streams"], "single", {k:round(v) for k,v in j["single_stream_proofs_per_s"].items()}, "prove ms", {k:round(v,3) for k,v in j["kernel_ms"]["prove"].items() if v})
except Exception as e: print("$n","failed",e)
PY
}
A="--no-cpu-baseline --no-flow-lines"
cp zkp_amd/libzkp_mi355x.so /tmp/new.so
for i in 1 2; do
cp /tmp/new.so zkp_amd/libzkp_mi355x.so
run flat_$i $A
run flat20_$i $A --steps 20 --warmup 5
cp tools/ab/prev3.so zkp_amd/libzkp_mi355x.so
run rolled_$i $A
run rolled20_$i $A --steps 20 --warmup 5
done
cp /tmp/new.so zkp_amd/libzkp_mi355x.so
run flat_c4 $A --config 4share
cp tools/ab/prev3.so zkp_amd/libzkp_mi355x.so
run rolled_c4 $A --config 4share
cp /tmp/new.so zkp_amd/libzkp_mi355x.so
