#!/bin/bash
# Round 5: the pair transcript interpreter with prefetched source words against the loader of rounds 2 - 4 (tools/x/variants/libzkp_v_noprefetch.so = -DZKP_AB_TR_NO_PREFETCH)
export TMPDIR=/tmp
R="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$R"
O=gpurun_out
V=tools/x/variants
echo "== 1. GPU suite"
python -m pytest tests -m gpu -x -q 2>&1 | tail -4
echo "== 2. A/B"
B="python bench.py --no-cpu-baseline --no-flow-lines"
val() { $B "$@" 2>/dev/null | tail -1 | python -c "import json,sys; j=json.loads(sys.stdin.read()); k=j['kernel_ms_per_call']; print('%.3f M/s   lone calls: prove %.3f ms (tables %.3f transcript %.3f), batch verify %.3f ms (transcript %.3f)' % (j['value']/1e6, k['prove']['total'], k['prove'].get('tables', -1), k['prove'].get('transcript', -1), k['batch_verify']['total'], k['batch_verify'].get('transcript', -1)))"; }
cp zkp_amd/libzkp_mi355x.so /tmp/shipped.so
for r in 1 2; do
  for v in shipped noprefetch; do
    if [ $v = shipped ]; then cp /tmp/shipped.so zkp_amd/libzkp_mi355x.so; else cp $V/libzkp_v_$v.so zkp_amd/libzkp_mi355x.so; fi
    echo "$v  20: $(val --steps 20 --warmup 5)"; echo "$v 200: $(val --steps 200)"
    echo "$v K=1 x 1 stream: $(val --steps 20 --no-sustained --batches-per-call 1 --streams 1 | cut -c1-12)   K=5 x 1: $(val --steps 20 --no-sustained --batches-per-call 5 --streams 1 | cut -c1-12)   K=1 x 4: $(val --steps 20 --no-sustained --batches-per-call 1 --streams 4 | cut -c1-12)"
  done
done
echo "== 3. kernel clock (one stream, K = 1 and K = 5)"
for v in shipped noprefetch; do
  if [ $v = shipped ]; then cp /tmp/shipped.so zkp_amd/libzkp_mi355x.so; else cp $V/libzkp_v_$v.so zkp_amd/libzkp_mi355x.so; fi
  for K in 1 5; do
    ( cd /tmp && rm -rf $R/$O/tprof && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/tprof -o t -- python $R/bench.py --no-cpu-baseline --no-flow-lines --no-sustained --steps 20 --warmup 5 --streams 1 --batches-per-call $K > /dev/null 2>&1 )
    python - "$v K=$K" <<'PY'
import csv, glob, sys
for f in glob.glob("gpurun_out/tprof/**/t_kernel_stats.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if any(s in row["Name"] for s in ("k_transcript_run", "k_tables_transcript")):
            print(sys.argv[1], row["Name"][:40], row["Calls"], "avg us %.1f" % (float(row["AverageNs"]) / 1e3), "min %.1f" % (float(row["MinNs"]) / 1e3))
PY
    rm -rf $O/tprof
  done
done
cp /tmp/shipped.so zkp_amd/libzkp_mi355x.so
