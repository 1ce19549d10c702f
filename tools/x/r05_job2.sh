#!/bin/bash
# Round 5, second GPU call: the crossbar walks re-laid for bank-conflict freedom (sources of every ds_bpermute_b32 inside one half of the wavefront):
# (1) the whole GPU suite, (2) constant-time evidence -- PMC counters per scalar pattern (instructions, LDS bank conflicts) and wavefront cycles,
# (3) 6-bit against 7-bit (two sets) fixed-base window on one box, (4) soak of the job code (fail-closed changes).
export TMPDIR=/tmp
R="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$R"
O=gpurun_out
mkdir -p $O
echo "== 1. GPU suite"
python -m pytest tests -m gpu -x -q 2>&1 | tail -4
echo "== 2. constant-time evidence"
( cd /tmp && rm -rf $R/$O/ct_prof
  timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_WAVES --output-format csv -d $R/$O/ct_prof -o ct -- python $R/tools/ct_check.py > /dev/null 2>&1
  timeout 300 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL --output-format csv -d $R/$O/ct_prof -o lds -- python $R/tools/ct_check.py > /dev/null 2>&1 )
python tools/ct_check.py --summarise $(find $O/ct_prof -name "ct_counter_collection.csv") $(find $O/ct_prof -name "lds_counter_collection.csv") > $O/r05_constant_time_counters.txt 2>&1
grep -v IDENTICAL $O/r05_constant_time_counters.txt | head -40
grep -c IDENTICAL $O/r05_constant_time_counters.txt
grep "k_terms_split<true, 16, true, 0>.*BANK_CONFLICT" $O/r05_constant_time_counters.txt | cut -c1-300
python tools/ct_check.py --cycles > $O/r05_constant_time_wave_cycles.txt 2>&1
cat $O/r05_constant_time_wave_cycles.txt
rm -rf $O/ct_prof
echo "== 3. 6-bit window (shipped) against the 7-bit window in two sets of 32 entries"
B="python bench.py --no-cpu-baseline --no-flow-lines"
val() { $B "$@" 2>/dev/null | tail -1 | python -c "import json,sys; j=json.loads(sys.stdin.read()); k=j['kernel_ms_per_call']['prove']; print('%.3f M/s   lone prove call: tables %.3f terms %.3f reduce %.3f total %.3f ms' % (j['value']/1e6, k['tables'], k['terms'], k['reduce'], k['total']))"; }
cp zkp_amd/libzkp_mi355x.so /tmp/shipped.so
for r in 1 2 3; do
  cp /tmp/shipped.so zkp_amd/libzkp_mi355x.so
  echo "w6   20: $(val --steps 20 --warmup 5)"; echo "w6  200: $(val --steps 200)"
  cp tools/x/variants/libzkp_w7.so zkp_amd/libzkp_mi355x.so
  echo "w7   20: $(val --steps 20 --warmup 5)"; echo "w7  200: $(val --steps 200)"
done
cp /tmp/shipped.so zkp_amd/libzkp_mi355x.so
echo "w6 1000: $(val)"; echo "w6 5share: $(val --config 5share)"; echo "w6 4share: $(val --config 4share)"
cp tools/x/variants/libzkp_w7.so zkp_amd/libzkp_mi355x.so
python -m pytest tests/test_gpu_toolbox.py tests/test_gpu_parity.py tests/test_gpu_device_entry.py -m gpu -x -q 2>&1 | tail -2
echo "w7 1000: $(val)"; echo "w7 5share: $(val --config 5share)"; echo "w7 4share: $(val --config 4share)"
cp /tmp/shipped.so zkp_amd/libzkp_mi355x.so
echo "== 4. soak"
timeout 100 python tools/pipe_soak.py 45 555 2>&1 | tail -3
timeout 100 python tools/soak.py 30 555 2>&1 | tail -3
