#!/bin/bash
# Round 5, first GPU call: (1) is ds_bpermute_b32 a constant-time look-up (microbenchmark + LDS counters), (2) parity of the crossbar walks,
# (3) A/B of the three look-ups of ZKP_OPT_CT_LOOKUP on one box, (4) the 7-bit fixed-base window (tools/x/variants/libzkp_w7.so), (5) wavefront cycles per scalar pattern.
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out
mkdir -p $O
echo "== 1. bpermute microbenchmark"
tools/microbench/bpermute_rate > $O/r05_bpermute_microbench.txt 2>&1
cat $O/r05_bpermute_microbench.txt
( cd /tmp && rm -rf $OLDPWD/$O/bperm_pmc && timeout 120 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS --output-format csv -d $OLDPWD/$O/bperm_pmc -o b -- $OLDPWD/tools/microbench/bpermute_rate > /dev/null 2>&1 )
python - <<'PY' >> gpurun_out/r05_bpermute_microbench.txt
import csv, glob, collections
f = glob.glob("gpurun_out/bperm_pmc/**/b_counter_collection.csv", recursive=True)
if f:
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f[0])):
        acc[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    print("# rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS, full-chip launches (the largest SQ_INSTS_LDS of each kernel):")
    for k, v in acc.items():
        i = max(range(len(v["SQ_INSTS_LDS"])), key=lambda j: v["SQ_INSTS_LDS"][j])
        print("%-40s INSTS_LDS %12.0f  IDX_ACTIVE %12.0f  BANK_CONFLICT %10.0f" % (k.replace("void ", "")[:40], v["SQ_INSTS_LDS"][i], v["SQ_LDS_IDX_ACTIVE"][i], v["SQ_LDS_BANK_CONFLICT"][i]))
PY
tail -12 $O/r05_bpermute_microbench.txt
echo "== 2. parity with the crossbar walks (default) and the other look-ups"
python -m pytest tests/test_gpu_device_entry.py tests/test_gpu_toolbox.py tests/test_gpu_parity.py tests/test_gpu_fused.py tests/test_gpu_statement_shapes.py -m gpu -x -q 2>&1 | tail -5
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
echo "== 3. A/B of ZKP_OPT_CT_LOOKUP: 0 crossbar, 2 LDS rows at the digit's index (rounds 2-4), 1 masked scans"
B="python bench.py --no-cpu-baseline --no-flow-lines"
val() { $B "$@" 2>/dev/null | tail -1 | python -c "import json,sys; j=json.loads(sys.stdin.read()); k=j['kernel_ms_per_call']['prove']; print('%.3f M/s   lone prove call: tables %.3f terms %.3f reduce %.3f total %.3f ms' % (j['value']/1e6, k['tables'], k['terms'], k['reduce'], k['total']))"; }
for r in 1 2; do
  for lk in 0 2 1; do
    echo "lookup $lk   20: $(val --steps 20 --warmup 5 --engine-opt 9=$lk)"
    echo "lookup $lk  200: $(val --steps 200 --engine-opt 9=$lk)"
  done
done
echo "lookup 0 1000: $(val --engine-opt 9=0)"
echo "lookup 2 1000: $(val --engine-opt 9=2)"
for cfg in 4share 5share; do for lk in 0 2; do echo "lookup $lk cfg $cfg: $(val --config $cfg --engine-opt 9=$lk)"; done; done
echo "== 4. 7-bit fixed-base window"
cp zkp_amd/libzkp_mi355x.so /tmp/shipped.so
cp tools/x/variants/libzkp_w7.so zkp_amd/libzkp_mi355x.so
python -m pytest tests/test_gpu_toolbox.py tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -3
for r in 1 2; do
  echo "w7   20: $(val --steps 20 --warmup 5)"
  echo "w7  200: $(val --steps 200)"
done
echo "w7 1000: $(val)"
echo "w7 cfg 5share: $(val --config 5share)"
cp /tmp/shipped.so zkp_amd/libzkp_mi355x.so
echo "== 5. wavefront cycles per scalar pattern"
python tools/ct_check.py --cycles > $O/r05_constant_time_wave_cycles.txt 2>&1
cat $O/r05_constant_time_wave_cycles.txt
