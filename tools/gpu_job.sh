#!/bin/bash
set -u
export TMPDIR=/tmp
O=gpurun_out
TAG=${1:-r02}
mkdir -p $O
timeout 1200 python -m pytest tests -m gpu -x -q > $O/${TAG}_pytest.txt 2>&1; echo "pytest rc=$?" >> $O/${TAG}_pytest.txt
grep -E "passed|failed|rc=|Error|assert" $O/${TAG}_pytest.txt | tail -8
timeout 900 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_WAVES --output-format csv -d $O/${TAG}_ct -o ct -- python tools/ct_check.py > $O/${TAG}_ct.log 2>&1
timeout 900 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL --output-format csv -d $O/${TAG}_ct -o lds -- python tools/ct_check.py > $O/${TAG}_lds.log 2>&1
python tools/ct_check.py --summarise $O/${TAG}_ct/ct_counter_collection.csv $O/${TAG}_ct/lds_counter_collection.csv > $O/${TAG}_constant_time_counters.txt 2>&1; tail -3 $O/${TAG}_constant_time_counters.txt; grep -v IDENTICAL $O/${TAG}_constant_time_counters.txt | head; rm -rf $O/${TAG}_ct
run() { # name, args...
  n=$1; shift
  t0=$SECONDS; timeout 600 python bench.py "$@" > $O/${TAG}_bench_$n.json 2> $O/${TAG}_bench_$n.err; echo "$n wall $((SECONDS-t0)) s rc=$?"; grep -v amdgpu.ids $O/${TAG}_bench_$n.err | tail -2 | cut -c1-300
  python - <<PY
import json
try:
    j=json.loads(open("$O/${TAG}_bench_$n.json").read().strip().splitlines()[-1])
    print("$n", "%.0f"%j["value"], "ms/step %.4f"%j["ms_per_step"], "streams", j["config"]["streams"], "pipelined", {k:round(v) for k,v in j["pipelined_proofs_per_s"].items()}, "single", {k:round(v) for k,v in j["single_stream_proofs_per_s"].items()})
except Exception as e: print("$n","failed",e)
PY
}
A="--no-cpu-baseline --no-flow-lines"
run b16k_grp $A --batch 16384 --steps 50 --engine-opt 6=1
run b16k_scan $A --batch 16384 --steps 50 --engine-opt 6=0
run b8k_grp $A --batch 8192 --steps 100 --engine-opt 6=1
run b8k_scan $A --batch 8192 --steps 100 --engine-opt 6=0
run c4 $A --config 4share
run d $A
