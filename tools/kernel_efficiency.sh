#!/bin/bash
# Per-kernel efficiency of one bench configuration on ONE stream (kernels do not overlap, so a kernel's own duration is meaningful):
# duration, VALU wave-instructions, fraction of the integer-VALU issue peak (34.5e12 lane-instr/s), mean occupancy per CU.
# A kernel with a long tail shows up as low issue fraction AND low occupancy (how the huge Pippenger bucket was found).
# usage (GPU box): tools/kernel_efficiency.sh OUTFILE [bench.py arguments ...]   -- three rocprofv3 runs (trace; two PMC passes)
set -u
export TMPDIR=/tmp
OUT=$1; shift
D=$(mktemp -d /tmp/keff.XXXX)
B="python bench.py --no-cpu-baseline --no-flow-lines --streams 1 --max-hw-queues 1 $*"
rocprofv3 --kernel-trace --output-format csv -d $D -o kt -- $B > $D/kt.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_WAVES --output-format csv -d $D -o valu -- $B > $D/valu.log 2>&1
rocprofv3 --kernel-trace --pmc MeanOccupancyPerCU --output-format csv -d $D -o occ -- $B > $D/occ.log 2>&1
python - <<PY > $OUT
import csv, collections
def name(r): return r["Kernel_Name"].split("(")[0].replace("void ", "")
dur = collections.defaultdict(list)
for r in csv.DictReader(open("$D/kt_kernel_trace.csv")):
    dur[name(r)].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
cnt = collections.defaultdict(lambda: collections.defaultdict(list))
for f in ("valu", "occ"):
    try:
        for r in csv.DictReader(open("$D/%s_counter_collection.csv" % f)):
            cnt[name(r)][r["Counter_Name"]].append(float(r["Counter_Value"]))
    except FileNotFoundError:
        pass
tot = sum(sum(v) for v in dur.values())
print("# $B")
print("# one stream; issue %% = SQ_INSTS_VALU x 64 lanes / duration / 34.5e12; occupancy = MeanOccupancyPerCU (wavefronts per CU, 32 = full)")
print("# note: the trace covers the whole process, so the kernels of the bench's UNTIMED setup are listed too (instance construction through zkp_msm_many")
print("#       without registered fixed bases, fixed-base table building): k_use_count, k_class_*, k_comb_tables<16>, k_hot_*, k_terms_split<false,...>, k_terms_r4")
print("%-50s %6s %10s %7s %12s %8s %8s %9s" % ("kernel", "calls", "avg_us", "time%", "valu_instr", "issue%", "occ/CU", "waves"))
for k, v in sorted(dur.items(), key=lambda kv: -sum(kv[1])):
    avg = sum(v) / len(v)
    c = cnt.get(k, {})
    valu = sum(c["SQ_INSTS_VALU"]) / len(c["SQ_INSTS_VALU"]) if c.get("SQ_INSTS_VALU") else float("nan")
    occ = sum(c["MeanOccupancyPerCU"]) / len(c["MeanOccupancyPerCU"]) if c.get("MeanOccupancyPerCU") else float("nan")
    wv = sum(c["SQ_WAVES"]) / len(c["SQ_WAVES"]) if c.get("SQ_WAVES") else float("nan")
    print("%-50s %6d %10.1f %7.2f %12.4g %8.1f %8.2f %9.0f" % (k[:50], len(v), avg / 1e3, 100.0 * sum(v) / tot, valu, 100.0 * valu * 64 / (avg * 1e-9) / 34.5e12, occ, wv))
PY
rm -rf $D
head -30 $OUT
