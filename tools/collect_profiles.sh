#!/bin/bash
# Run on the GPU box (through gpurun): bench + rocprofv3 kernel trace + PMC passes -> gpurun_out/$1_*
# usage: tools/collect_profiles.sh r02        (then copy the summaries you want judged into profiles/)
# PMC passes are separate runs with --kernel-trace only (MI355X_MICROARCH.md: never combine --pmc with the trace domains).
set -u
TAG=${1:-r02}
export TMPDIR=/tmp
O=gpurun_out
mkdir -p $O
B="python bench.py --steps 20 --warmup 1 --no-cpu-baseline --no-flow-lines"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_prof -o kt -- $B > $O/${TAG}_kt.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/${TAG}_prof -o fetch -- $B > $O/${TAG}_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/${TAG}_prof -o write -- $B > $O/${TAG}_write.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $O/${TAG}_prof -o sq -- $B > $O/${TAG}_sq.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_INST_ANY --output-format csv -d $O/${TAG}_prof -o sq2 -- $B > $O/${TAG}_sq2.log 2>&1
python - <<PY > $O/${TAG}_kernel_stats.txt
import csv
rows = list(csv.DictReader(open("$O/${TAG}_prof/kt_kernel_stats.csv")))
print("# rocprofv3 --kernel-trace --stats -- $B   (MI355X)")
print("%-46s %6s %12s %12s %12s %8s" % ("kernel", "calls", "avg_us", "min_us", "max_us", "pct"))
for r in rows:
    print("%-46s %6s %12.1f %12.1f %12.1f %8s" % (r["Name"].split("(")[0].replace("void ", "")[:46], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3, r["Percentage"]))
PY
python tools/pmc_summary.py $O/${TAG}_pmc_counters.json $O/${TAG}_prof/fetch_counter_collection.csv $O/${TAG}_prof/write_counter_collection.csv $O/${TAG}_prof/sq_counter_collection.csv $O/${TAG}_prof/sq2_counter_collection.csv > $O/${TAG}_pmc_counters.txt
rm -rf $O/${TAG}_prof
head -14 $O/${TAG}_kernel_stats.txt; grep -E "k_terms|_step_totals|k_pip_prep|k_pip_bucket_part" $O/${TAG}_pmc_counters.txt | cut -c1-300
