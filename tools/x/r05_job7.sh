#!/bin/bash
# Where does the term kernel's HBM traffic come from after the crossbar walks?  FETCH_SIZE / L2 hit rate of k_terms_split (K = 5 launch) for: the 6-bit window
# with the LDS look-up of rounds 2-4 (lookup 2), the 6-bit window with the crossbar walks (lookup 0), the shipped 7-bit crossbar library; 4 streams and 1 stream.
export TMPDIR=/tmp
R="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$R"
O=gpurun_out
cp zkp_amd/libzkp_mi355x.so /tmp/shipped.so
B="python bench.py --no-cpu-baseline --no-flow-lines --steps 20 --warmup 1"
run() {   # $1 = tag, rest = bench args
  tag=$1; shift
  for pmc in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum"; do
    rm -rf $O/tr_prof
    timeout 200 rocprofv3 --kernel-trace --pmc $pmc --output-format csv -d $O/tr_prof -o f -- $B "$@" > $O/tr_last.log 2>&1
    python - "$tag" <<'PY'
import csv, glob, sys, collections
f = glob.glob("gpurun_out/tr_prof/**/f_counter_collection.csv", recursive=True)
if not f:
    print(sys.argv[1], "no counter file"); sys.exit()
acc = collections.defaultdict(list)
for r in csv.DictReader(open(f[0])):
    if "k_terms_split<true" in r["Kernel_Name"]:
        acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
print("%-34s" % sys.argv[1], "  ".join("%s = %.4g (n = %d)" % (k, sum(v) / len(v), len(v)) for k, v in acc.items()), flush=True)
PY
  done
}
cp tools/x/variants/libzkp_w6.so zkp_amd/libzkp_mi355x.so
run "w6 lookup 2 (LDS rows), 4 streams" --engine-opt 9=2
run "w6 lookup 0 (crossbar), 4 streams" --engine-opt 9=0
run "w6 lookup 0 (crossbar), 1 stream" --engine-opt 9=0 --streams 1 --max-hw-queues 1
cp /tmp/shipped.so zkp_amd/libzkp_mi355x.so
run "w7 crossbar (shipped), 4 streams"
run "w7 crossbar (shipped), 1 stream" --streams 1 --max-hw-queues 1
run "w7 crossbar, no grouped walk" --engine-opt 6=0
run "w7 crossbar, tables for Q" --engine-opt 3=1
rm -rf $O/tr_prof
