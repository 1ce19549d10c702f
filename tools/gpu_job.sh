#!/bin/bash
# scratch job script for gpurun (edited per call)
set -u
export TMPDIR=/tmp
O=gpurun_out
TAG=${1:-r02b}
mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q > $O/${TAG}_pytest.log 2>&1; echo "pytest rc=$?" >> $O/${TAG}_pytest.log
tail -5 $O/${TAG}_pytest.log
run() { # name, args...
  n=$1; shift
  timeout 300 python bench.py --no-cpu-baseline "$@" > $O/${TAG}_$n.json 2> $O/${TAG}_$n.err
  python - <<PY
import json
try:
    j=json.loads(open("$O/${TAG}_$n.json").read().strip().splitlines()[-1])
    print("$n", "%.0f"%j["value"], "%.4f"%j["ms_per_step"], "enq %.3f"%j.get("host_enqueue_ms_per_step",0), j["config"]["streams"], "prove", {k:round(v,3) for k,v in j["kernel_ms"]["prove"].items()}, "bv", {k:round(v,3) for k,v in j["kernel_ms"]["batch_verify"].items()})
except Exception as e: print("$n","failed",e)
PY
}
run t20 --steps 20 --warmup 5
run t200 --steps 200 --warmup 3
run e20 --steps 20 --warmup 5 --engine-opt 4=1
run e200 --steps 200 --warmup 3 --engine-opt 4=1
run t200b --steps 200 --warmup 3
run e200b --steps 200 --warmup 3 --engine-opt 4=1
B="python bench.py --steps 20 --warmup 1 --no-cpu-baseline"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_prof -o kt -- $B > $O/${TAG}_kt.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $O/${TAG}_prof -o sq -- $B > $O/${TAG}_sq.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_INST_CYCLES_VMEM_RD --output-format csv -d $O/${TAG}_prof -o sq2 -- $B > $O/${TAG}_sq2.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/${TAG}_prof -o fetch -- $B > $O/${TAG}_fetch.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/${TAG}_prof -o write -- $B > $O/${TAG}_write.log 2>&1
rocprofv3 --list-avail > $O/${TAG}_counters_avail.txt 2>&1
python - <<PY > $O/${TAG}_kernel_stats.txt
import csv
rows = list(csv.DictReader(open("$O/${TAG}_prof/kt_kernel_stats.csv")))
print("# rocprofv3 --kernel-trace --stats -- $B   (MI355X)")
print("%-46s %6s %12s %12s %12s %8s" % ("kernel", "calls", "avg_us", "min_us", "max_us", "pct"))
for r in rows:
    print("%-46s %6s %12.1f %12.1f %12.1f %8s" % (r["Name"].split("(")[0].replace("void ", "")[:46], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3, r["Percentage"]))
PY
python tools/pmc_summary.py $O/${TAG}_pmc.json $O/${TAG}_prof/fetch_counter_collection.csv $O/${TAG}_prof/write_counter_collection.csv $O/${TAG}_prof/sq_counter_collection.csv $O/${TAG}_prof/sq2_counter_collection.csv > $O/${TAG}_pmc.txt
head -16 $O/${TAG}_kernel_stats.txt; grep -E "k_terms|_step_totals|comb_tables|bucket_part|pip_prepare" $O/${TAG}_pmc.txt | cut -c1-600
tail -3 $O/${TAG}_sq2.log
rm -rf $O/${TAG}_prof/*/
