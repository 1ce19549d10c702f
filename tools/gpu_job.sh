#!/bin/bash
set -u
export TMPDIR=/tmp
O=gpurun_out
TAG=${1:-r02}
mkdir -p $O
for o3 in 1 0; do for n in 1024 4096; do
  PROBE_OPT3=$o3 PROBE_NOCHECK=1 rocprofv3 --kernel-trace --output-format csv -d $O/${TAG}_p -o kt_${o3}_$n -- python tools/microbench/group_walk_probe.py $n > $O/${TAG}_kt.log 2>&1
  python - <<PY
import csv
rows=[r for r in csv.DictReader(open("$O/${TAG}_p/kt_${o3}_${n}_kernel_trace.csv")) if "k_terms_split" in r["Kernel_Name"]]
print("single_use_tables=$o3 n=$n", rows[-1]["Kernel_Name"][:36], [round((int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3) for r in rows[-6:]], "us (grouped, scan alternating)")
PY
done; done
rm -rf $O/${TAG}_p
