export TMPDIR=/tmp
export GPU_MAX_HW_QUEUES=4
P='import json,sys
j=json.loads(sys.stdin.read()); k=j["kernel_ms_per_call"]; print("%.3f M/s" % (j["value"]/1e6), {f: round(k[f]["total"],3) for f in k})'
for r in 1 2; do
for a in "--streams 1" "--streams 1 --engine-opt 5=1" "--streams 1 --batches-per-call 1" "--streams 1 --batches-per-call 1 --engine-opt 5=1"; do
  echo "$a: $(timeout 120 python bench.py --no-cpu-baseline --no-flow-lines --steps 20 --warmup 5 $a 2>/dev/null | tail -1 | python -c "$P")"
done; done
