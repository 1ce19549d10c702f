#!/bin/bash
# Run on the GPU box: the VALU microbenchmark (rates per second) and one PMC pass over it (issue cycles in the chip's own clock)
# -> gpurun_out/r04_valu_rates_microbench.txt, gpurun_out/r04_valu_issue_cycles_pmc.txt
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out
mkdir -p $O
$R/tools/microbench/valu_rates > $O/r04_valu_rates_microbench.txt
cd /tmp
rm -rf $O/valu_cal
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU2 SQ_ACTIVE_INST_VALU SQ_WAVES --output-format csv -d $O/valu_cal -o cal -- $R/tools/microbench/valu_rates > $O/valu_cal.log 2>&1
cd $R
python tools/valu_cal_summary.py $(find $O/valu_cal -name cal_counter_collection.csv) > $O/r04_valu_issue_cycles_pmc.txt
cat $O/r04_valu_rates_microbench.txt $O/r04_valu_issue_cycles_pmc.txt
