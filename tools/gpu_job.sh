#!/bin/bash
# scratch job script for gpurun (edited per call)
set -u
export TMPDIR=/tmp
O=gpurun_out
TAG=${1:-r02}
mkdir -p $O
timeout 1200 python -m pytest tests -m gpu -x -q --durations=0 > $O/${TAG}_pytest_gpu_durations.txt 2>&1; echo "pytest rc=$?" >> $O/${TAG}_pytest_gpu_durations.txt
grep -E "passed|failed|rc=|Error" $O/${TAG}_pytest_gpu_durations.txt | tail -5
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
bash tools/collect_profiles.sh $TAG > $O/${TAG}_collect.log 2>&1; tail -6 $O/${TAG}_collect.log | cut -c1-250
cp $O/${TAG}_pmc_counters.json profiles/r02_pmc_counters.json   # so that the bench runs below find counters keyed to these sources
run() { # name, args...
  n=$1; shift
  t0=$SECONDS; timeout 600 python bench.py "$@" > $O/${TAG}_bench_$n.json 2> $O/${TAG}_bench_$n.err; echo "$n wall $((SECONDS-t0)) s rc=$?"; grep -v amdgpu.ids $O/${TAG}_bench_$n.err | tail -2 | cut -c1-300
  python - <<PY
import json
try:
    j=json.loads(open("$O/${TAG}_bench_$n.json").read().strip().splitlines()[-1])
    print("$n", "%.0f"%j["value"], "ms/step %.4f"%j["ms_per_step"], "streams", j["config"]["streams"], "pipelined", {k:round(v) for k,v in j["pipelined_proofs_per_s"].items()}, "single", {k:round(v) for k,v in j["single_stream_proofs_per_s"].items()})
    print("   roofline:", j["roofline"]["kernel"], "frac %.5f"%j["roofline"]["frac"], "traffic", j["roofline"]["traffic"], "| step_valu", (j.get("step_valu") or {}).get("frac"), "| cpu", (j.get("cpu_baseline") or {}).get("value"))
except Exception as e: print("$n","failed",e)
PY
}
run 1gpu_steps20 --steps 20 --warmup 5
run 1gpu
run cfg3 --config 3
run cfg4share --config 4share
run cfg5share --config 5share
timeout 900 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_WAVES --output-format csv -d $O/${TAG}_ct -o ct -- python tools/ct_check.py > $O/${TAG}_ct.log 2>&1
timeout 900 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL --output-format csv -d $O/${TAG}_ct -o lds -- python tools/ct_check.py > $O/${TAG}_lds.log 2>&1
python tools/ct_check.py --summarise $O/${TAG}_ct/ct_counter_collection.csv $O/${TAG}_ct/lds_counter_collection.csv > $O/${TAG}_constant_time_counters.txt 2>&1; tail -2 $O/${TAG}_constant_time_counters.txt; rm -rf $O/${TAG}_ct
timeout 900 python tools/e2e_toolbox_bench.py 4096 32768 131072 > $O/${TAG}_e2e_toolbox_host_included.txt 2>&1; tail -4 $O/${TAG}_e2e_toolbox_host_included.txt | cut -c1-250
