#!/bin/bash
# Round 5: host scaling, constant-time evidence and soaks once more, on the final sources (the collection of tools/x/r05_job17.sh is from the same build)
export TMPDIR=/tmp
R="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$R"
O=gpurun_out
mkdir -p $O
echo "== 1. host scaling"
timeout 600 python tools/pipe_host_scaling.py > $O/r05_pipe_host_scaling.txt 2>&1; tail -14 $O/r05_pipe_host_scaling.txt | cut -c1-200
echo "== 2. constant-time evidence"
( cd /tmp && rm -rf $R/$O/ct_prof
  timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_WAVES --output-format csv -d $R/$O/ct_prof -o ct -- python $R/tools/ct_check.py > /dev/null 2>&1
  timeout 300 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL --output-format csv -d $R/$O/ct_prof -o lds -- python $R/tools/ct_check.py > /dev/null 2>&1 )
python tools/ct_check.py --summarise $(find $O/ct_prof -name "ct_counter_collection.csv") $(find $O/ct_prof -name "lds_counter_collection.csv") > $O/r05_constant_time_counters.txt 2>&1
grep -v IDENTICAL $O/r05_constant_time_counters.txt | head; grep -c IDENTICAL $O/r05_constant_time_counters.txt
python tools/ct_check.py --cycles > $O/r05_constant_time_wave_cycles.txt 2>&1
tail -12 $O/r05_constant_time_wave_cycles.txt | cut -c1-250
rm -rf $O/ct_prof
echo "== 3. soaks"
timeout 200 python tools/pipe_soak.py 120 2718 2>&1 | tail -2
timeout 200 python tools/soak.py 120 2718 2>&1 | tail -2
