#!/bin/bash
# usage (on the GPU box): bash tools/x/runlines.sh LINES.txt OUT.txt [rounds]    -- each line "label | bench.py arguments"; one bench.py process per line, alternated over `rounds`
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
IN=$1; OUT=$2; R=${3:-2}
: > $OUT
for r in $(seq 1 $R); do
  while IFS='|' read -r label args; do
    [ -z "$label" ] && continue
    case "$label" in \#*) continue;; esac
    line=$(timeout 300 python bench.py --no-cpu-baseline --no-flow-lines $args 2>gpurun_out/runlines_err.log | tail -1 | python -c "
import json,sys
try:
    j=json.loads(sys.stdin.read())
    k=j.get('kernel_ms_per_call',{})
    tot=' '.join('%s=%.3f'%(f,k[f]['total']) for f in sorted(k))
    print('%.3f M/s  %.4f ms/step  K=%s calls=%s streams=%s  lone-call ms: %s' % (j['value']/1e6, j['ms_per_step'], j['config'].get('batches_per_call'), j['config'].get('calls'), j['config'].get('streams'), tot))
except Exception as e:
    print('ERR', e)
")
    echo "$label |$args | $line" | tee -a $OUT
    if echo "$line" | grep -q ERR; then tail -3 gpurun_out/runlines_err.log | tee -a $OUT; fi
  done < $IN
done
