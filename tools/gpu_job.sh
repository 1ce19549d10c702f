#!/bin/bash
# scratch job script for gpurun (edited per call)
set -u
export TMPDIR=/tmp
O=gpurun_out
TAG=${1:-r02a}
mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q --durations=15 > $O/${TAG}_pytest.log 2>&1; echo "pytest rc=$?" >> $O/${TAG}_pytest.log
tail -25 $O/${TAG}_pytest.log
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/${TAG}_bench20.json 2> $O/${TAG}_bench20.err; tail -c 600 $O/${TAG}_bench20.err
timeout 300 python bench.py --steps 200 --warmup 3 --no-cpu-baseline > $O/${TAG}_bench200.json 2> $O/${TAG}_bench200.err
python - <<PY
import json
for f in ("bench20","bench200"):
    try:
        j=json.loads(open("$O/${TAG}_%s.json"%f).read().strip().splitlines()[-1])
        print(f, j["value"], j["ms_per_step"], j["config"]["streams"], json.dumps(j["kernel_ms"]))
    except Exception as e: print(f,"failed",e)
PY
B="python bench.py --steps 20 --warmup 1 --no-cpu-baseline"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_prof -o kt -- $B > $O/${TAG}_kt.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $O/${TAG}_prof -o sq -- $B > $O/${TAG}_sq.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/${TAG}_prof -o fetch -- $B > $O/${TAG}_fetch.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/${TAG}_prof -o write -- $B > $O/${TAG}_write.log 2>&1
python - <<PY > $O/${TAG}_kernel_stats.txt
import csv
rows = list(csv.DictReader(open("$O/${TAG}_prof/kt_kernel_stats.csv")))
print("# rocprofv3 --kernel-trace --stats -- $B   (MI355X)")
print("%-46s %6s %12s %12s %12s %8s" % ("kernel", "calls", "avg_us", "min_us", "max_us", "pct"))
for r in rows:
    print("%-46s %6s %12.1f %12.1f %12.1f %8s" % (r["Name"].split("(")[0].replace("void ", "")[:46], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3, r["Percentage"]))
PY
python tools/pmc_summary.py $O/${TAG}_pmc.json $O/${TAG}_prof/fetch_counter_collection.csv $O/${TAG}_prof/write_counter_collection.csv $O/${TAG}_prof/sq_counter_collection.csv > $O/${TAG}_pmc.txt
head -16 $O/${TAG}_kernel_stats.txt; grep -E "k_terms|_step_totals|comb_tables|encode" $O/${TAG}_pmc.txt | cut -c1-330
rm -rf $O/${TAG}_prof/*/  # keep only the csv summaries
ls $O/${TAG}_prof | head
