#!/bin/bash
# A/B of the two-accumulator grouped walk (-DZKP_GROUP_TWO_ACC build in tools/x/variants/libzkp_two_acc.so) against the shipped library: parity tests first, then bench, then FETCH_SIZE
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
cp zkp_amd/libzkp_mi355x.so /tmp/shipped.so
cp tools/x/variants/libzkp_two_acc.so zkp_amd/libzkp_mi355x.so
echo "== parity with the variant library"
python -m pytest tests/test_gpu_toolbox.py tests/test_gpu_device_entry.py tests/test_gpu_fused.py tests/test_gpu_fullsize.py -m gpu -x -q 2>&1 | tail -3
B="python bench.py --no-cpu-baseline --no-flow-lines"
val() { $B "$@" 2>/dev/null | tail -1 | python -c "import json,sys; j=json.loads(sys.stdin.read()); k=j['kernel_ms_per_call']['prove']; print('%.3f M/s   lone prove call: terms %.3f ms, total %.3f ms' % (j['value']/1e6, k['terms'], k['total']))"; }
for r in 1 2 3; do
  cp /tmp/shipped.so zkp_amd/libzkp_mi355x.so;                       echo "shipped  20: $(val --steps 20 --warmup 5)";  echo "shipped 200: $(val --steps 200)"
  cp tools/x/variants/libzkp_two_acc.so zkp_amd/libzkp_mi355x.so;    echo "two_acc  20: $(val --steps 20 --warmup 5)";  echo "two_acc 200: $(val --steps 200)"
done
echo "== FETCH_SIZE of the term kernel (K = 5 launch)"
for v in shipped two_acc; do
  if [ $v = shipped ]; then cp /tmp/shipped.so zkp_amd/libzkp_mi355x.so; else cp tools/x/variants/libzkp_two_acc.so zkp_amd/libzkp_mi355x.so; fi
  rm -rf gpurun_out/ta_prof
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/ta_prof -o f -- $B --steps 20 --warmup 1 > /dev/null 2>&1
  python - <<PY
import csv, glob
f = glob.glob("gpurun_out/ta_prof/**/f_counter_collection.csv", recursive=True)[0]
v = [float(r["Counter_Value"]) for r in csv.DictReader(open(f)) if "k_terms_split<true" in r["Kernel_Name"] and r["Counter_Name"] == "FETCH_SIZE"]
print("$v: FETCH_SIZE of k_terms_split<true,...> per launch: %.1f MiB (%d launches)" % (sum(v) / len(v) / 1024.0, len(v)))
PY
done
rm -rf gpurun_out/ta_prof
cp /tmp/shipped.so zkp_amd/libzkp_mi355x.so
