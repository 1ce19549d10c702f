#!/bin/bash
# One gpurun call: A/B of the 20-step schedule (stream priorities, staggered starts, call shapes), of compiler scheduling strategies for the
# kernel library (tools/x/variants/*.so, built by tools/x/build_variants.sh) and of the zkp_pipe job shape.  Results -> gpurun_out/exp_schedules.txt
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
OUT=gpurun_out/exp_schedules.txt
: > $OUT
B="python bench.py --no-cpu-baseline --no-flow-lines"
one() {   # label, args...
  local label="$1"; shift
  local line
  line=$(timeout 300 $B "$@" 2>gpurun_out/exp_err.log | tail -1 | python -c "
import json,sys
try:
    j=json.loads(sys.stdin.read())
    print('%.3f M/s  %.4f ms/step  K=%s calls=%s streams=%s' % (j['value']/1e6, j['ms_per_step'], j['config'].get('batches_per_call'), j['config'].get('calls'), j['config'].get('streams')))
except Exception as e:
    print('ERR', e)
")
  echo "$label | $* | $line" | tee -a $OUT
  if echo "$line" | grep -q ERR; then tail -3 gpurun_out/exp_err.log | tee -a $OUT; fi
}
S20="--steps 20 --warmup 5"
echo "## 20 steps (the driver's shape): schedules, alternated, 2 rounds" | tee -a $OUT
for r in 1 2; do
  one base          $S20
  one prio_h000     $S20 --stream-priorities=-1,0,0,0
  one prio_h00l     $S20 --stream-priorities=-1,0,0,1
  one prio_hh0l     $S20 --stream-priorities=-1,-1,0,1
  one prio_h0ll     $S20 --stream-priorities=-1,0,1,1
  one stagger_0.2   $S20 --stagger-ms 0.2
  one stagger_0.4   $S20 --stagger-ms 0.4
  one stagger_0.8   $S20 --stagger-ms 0.8
  one k4s5          $S20 --batches-per-call 4 --streams 5
  one k2s10         $S20 --batches-per-call 2 --streams 10 --max-hw-queues 10
  one k10s2         $S20 --batches-per-call 10 --streams 2
  one nograph       $S20 --no-graphs
  one nograph_prio  $S20 --no-graphs --stream-priorities=-1,0,0,1
done
echo "## 200 steps K = 50" | tee -a $OUT
for r in 1 2; do
  one base200       --steps 200
  one prio200       --steps 200 --stream-priorities=-1,0,0,1
done
echo "## kernel library built with other scheduling strategies (same sources)" | tee -a $OUT
cp zkp_amd/libzkp_mi355x.so gpurun_out/lib_shipped.so
for r in 1 2; do
  for v in base max-ilp; do
    if [ -f tools/x/variants/libzkp_$v.so ]; then
      cp tools/x/variants/libzkp_$v.so zkp_amd/libzkp_mi355x.so
      one lib_$v $S20
      one lib_${v}_200 --steps 200
    fi
  done
done
cp gpurun_out/lib_shipped.so zkp_amd/libzkp_mi355x.so
rm -f gpurun_out/lib_shipped.so
echo "## zkp_pipe job shapes (K x contexts), pinned / staged / pinned" | tee -a $OUT
timeout 400 python tools/e2e_pipe_bench.py --shapes 5x6,10x6,10x8,20x6,5x6 --order pinned,staged 2>&1 | grep -v "^$" | cut -c1-400 | tee -a $OUT
