#!/bin/bash
# Round 5, third GPU call: the library as it ships (7-bit fixed-base windows, crossbar look-ups only): whole GPU suite incl. the new statement-shape and
# 64-constraint tests, constant-time evidence with the extra pattern, the driver's own command, the three readings of configs[4].
export TMPDIR=/tmp
R="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$R"
O=gpurun_out
mkdir -p $O
echo "== 1. GPU suite"
python -m pytest tests -m gpu -x -q --durations=12 2>&1 | tail -22
echo "== 2. constant-time evidence"
( cd /tmp && rm -rf $R/$O/ct_prof
  timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_WAVES --output-format csv -d $R/$O/ct_prof -o ct -- python $R/tools/ct_check.py > /dev/null 2>&1
  timeout 300 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL --output-format csv -d $R/$O/ct_prof -o lds -- python $R/tools/ct_check.py > /dev/null 2>&1 )
python tools/ct_check.py --summarise $(find $O/ct_prof -name "ct_counter_collection.csv") $(find $O/ct_prof -name "lds_counter_collection.csv") > $O/r05_constant_time_counters.txt 2>&1
grep -v IDENTICAL $O/r05_constant_time_counters.txt | head -20
grep -c IDENTICAL $O/r05_constant_time_counters.txt
grep "k_terms_split.*BANK_CONFLICT" $O/r05_constant_time_counters.txt | cut -c1-200
python tools/ct_check.py --cycles > $O/r05_constant_time_wave_cycles.txt 2>&1
cat $O/r05_constant_time_wave_cycles.txt
rm -rf $O/ct_prof
echo "== 3. the driver's command"
/usr/bin/time -v python bench.py --steps 20 --warmup 5 > $O/r05_bench_steps20_pre.json 2> $O/r05_bench_steps20_pre.err
grep -E "Elapsed|Maximum resident" $O/r05_bench_steps20_pre.err
python - <<'PY'
import json
j = json.loads(open("gpurun_out/r05_bench_steps20_pre.json").read().strip().splitlines()[-1])
print("value %.3f M  ms/step %.3f  ct_schedule %s" % (j["value"] / 1e6, j["ms_per_step"], j["config"]["ct_schedule"][:60]))
print("sustained", {k: v for k, v in j.get("sustained", {}).items() if k != "note"})
print("value_ct_by_construction", j.get("value_ct_by_construction"), "ct", j.get("ct"))
e = j.get("e2e_host_buffers", {})
print("e2e", round(e.get("proofs_per_s", 0)), round(e.get("pipelined", {}).get("proofs_per_s", 0)), round(e.get("pipelined_staged", {}).get("proofs_per_s", 0)), round(e.get("threads", {}).get("proofs_per_s", 0)))
print("cpu", j.get("cpu_baseline", {}).get("value"), "parity", j.get("parity_checked", {}).get("equal"), j.get("parity_checked", {}).get("proofs"))
print("flows", {k: round(v) for k, v in j["pipelined_proofs_per_s"].items()})
PY
echo "== 4. the readings of configs[4]"
B="python bench.py --no-cpu-baseline --no-flow-lines --config 5share"
for f in terms constraints constraints2; do
  $B --w64-form $f > $O/r05_bench_cfg5share_$f.json 2> $O/r05_bench_cfg5share_$f.err || tail -3 $O/r05_bench_cfg5share_$f.err
  python -c "
import json
j = json.loads(open('gpurun_out/r05_bench_cfg5share_$f.json').read().strip().splitlines()[-1])
k = j['kernel_ms_per_call']
print('$f', 'value %.3f M/s  ms/step %.3f' % (j['value'] / 1e6, j['ms_per_step']), 'lone prove', {a: round(b, 3) for a, b in k['prove'].items()}, 'lone batch verify total', round(k['batch_verify']['total'], 3))
"
done
