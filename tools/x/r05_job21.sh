#!/bin/bash
# Round 5: call shapes for the driver's 20 steps once more, now that the narrow phases of a call are shorter (K batches per call x S streams; default 5 x 4)
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
B="python bench.py --no-cpu-baseline --no-flow-lines --no-sustained --steps 20 --warmup 5"
val() { $B "$@" 2>/dev/null | tail -1 | python -c "import json,sys; j=json.loads(sys.stdin.read()); print('%.3f' % (j['value']/1e6))"; }
for r in 1 2 3; do
  echo "round $r:  5x4 $(val)   4x5 $(val --batches-per-call 4 --streams 5)   2x10 $(val --batches-per-call 2 --streams 10)   2x5 $(val --batches-per-call 2 --streams 5)   10x2 $(val --batches-per-call 10 --streams 2)   5x2 $(val --batches-per-call 5 --streams 2)   4x3 $(val --batches-per-call 4 --streams 3)   1x8 $(val --batches-per-call 1 --streams 8)"
done
