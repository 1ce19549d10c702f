#!/bin/bash
# Run on the GPU box (through gpurun): bench + rocprofv3 kernel trace + PMC passes -> gpurun_out/${TAG}_*_cfg${CFG}*
# usage: tools/collect_profiles.sh r03 [CONFIG=2] [STEPS] [extra bench.py arguments ...]   (then copy the summaries you want judged into profiles/)
# PMC passes are separate runs with --kernel-trace only (MI355X_MICROARCH.md: never combine --pmc with the trace domains).
set -u
TAG=${1:-r06}
CFG=${2:-2}
case $CFG in 2) DEF=20;; 3) DEF=2;; 4share) DEF=2;; 5share) DEF=8;; *) DEF=4;; esac
STEPS=${3:-$DEF}
shift; shift; shift
export TMPDIR=/tmp
O=gpurun_out
S=${TAG}_cfg${CFG}
mkdir -p $O
B="python bench.py --config $CFG --steps $STEPS --warmup 1 --no-cpu-baseline --no-flow-lines $*"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/${S}_prof -o kt -- $B > $O/${S}_kt.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/${S}_prof -o fetch -- $B > $O/${S}_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/${S}_prof -o write -- $B > $O/${S}_write.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU2 SQ_INSTS_VALU SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $O/${S}_prof -o sq -- $B > $O/${S}_sq.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_INST_ANY --output-format csv -d $O/${S}_prof -o sq2 -- $B > $O/${S}_sq2.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_INT32 MeanOccupancyPerCU --output-format csv -d $O/${S}_prof -o sq3 -- $B > $O/${S}_sq3.log 2>&1
# round 6: does the term kernel's traffic cost it anything?  wavefront cycles parked at s_waitcnt / barriers, and the L2's hit rate (separate passes;
# a counter this rocprofv3 does not know makes its pass produce no file, which the summary tolerates)
rocprofv3 --kernel-trace --pmc SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $O/${S}_prof -o sq4 -- $B > $O/${S}_sq4.log 2>&1
rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $O/${S}_prof -o tcc -- $B > $O/${S}_tcc.log 2>&1
rocprofv3 --kernel-trace --pmc TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum --output-format csv -d $O/${S}_prof -o tcp -- $B > $O/${S}_tcp.log 2>&1
# the same command on ONE stream: a kernel's own duration (nothing overlaps it) -- the time base of the time-share column and of roofline.launch_ms_rocprof_one_stream
rocprofv3 --kernel-trace --stats --output-format csv -d $O/${S}_prof -o kt1 -- $B --streams 1 --max-hw-queues 1 > $O/${S}_kt1.log 2>&1
python - <<PY > $O/${S}_kernel_stats.txt
import csv
rows = list(csv.DictReader(open("$O/${S}_prof/kt_kernel_stats.csv")))
import sys
sys.path.insert(0, ".")
import bench
print("# rocprofv3 --kernel-trace --stats -- $B   (MI355X; kernel sources sha256 %s)" % bench.source_sha256()[:16])
print("%-46s %6s %12s %12s %12s %8s" % ("kernel", "calls", "avg_us", "min_us", "max_us", "pct"))
for r in rows:
    print("%-46s %6s %12.1f %12.1f %12.1f %8s" % (r["Name"].split("(")[0].replace("void ", "")[:46], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3, r["Percentage"]))
PY
KPC=$(python -c "import json,sys; print(json.loads([l for l in open('$O/${S}_kt.log').read().splitlines() if l.startswith('{')][-1])['config']['batches_per_call'])")
EXTRA=$(ls $O/${S}_prof/sq4_counter_collection.csv $O/${S}_prof/tcc_counter_collection.csv $O/${S}_prof/tcp_counter_collection.csv 2>/dev/null)
python tools/pmc_summary.py $O/${TAG}_pmc_counters_cfg${CFG}_k${KPC}.json $O/${S}_kt.log $O/${S}_prof/fetch_counter_collection.csv $O/${S}_prof/write_counter_collection.csv $O/${S}_prof/sq_counter_collection.csv $O/${S}_prof/sq2_counter_collection.csv $O/${S}_prof/sq3_counter_collection.csv $EXTRA $O/${S}_prof/kt1_kernel_trace.csv > $O/${TAG}_pmc_counters_cfg${CFG}_k${KPC}.txt
python - <<PY > $O/${TAG}_kernel_stats_cfg${CFG}_k${KPC}_one_stream.txt
import csv, sys
sys.path.insert(0, ".")
import bench
rows = list(csv.DictReader(open("$O/${S}_prof/kt1_kernel_stats.csv")))
print("# rocprofv3 --kernel-trace --stats -- $B --streams 1 --max-hw-queues 1   (MI355X; kernel sources sha256 %s)" % bench.source_sha256()[:16])
print("%-46s %6s %12s %12s %12s %8s" % ("kernel", "calls", "avg_us", "min_us", "max_us", "pct"))
for r in rows:
    print("%-46s %6s %12.1f %12.1f %12.1f %8s" % (r["Name"].split("(")[0].replace("void ", "")[:46], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3, r["Percentage"]))
PY
mv $O/${S}_kernel_stats.txt $O/${TAG}_kernel_stats_cfg${CFG}_k${KPC}.txt
rm -rf $O/${S}_prof
head -14 $O/${TAG}_kernel_stats_cfg${CFG}_k${KPC}.txt; grep -E "k_terms|_step_totals|k_pip_prep|k_pip_bucket_part" $O/${TAG}_pmc_counters_cfg${CFG}_k${KPC}.txt | cut -c1-300
