#!/bin/bash
# Round 5: the Horner tail of k_pip_combine on one limb per lane (rowfe.h) against the 4-lane quads (tools/x/variants/libzkp_quad_horner.so = the build before):
# (1) GPU suite incl. the new self-test, (2) A/B at the driver's shape, at 200 steps and for lone calls, (3) the per-kernel clock of the lone batch verification
export TMPDIR=/tmp
R="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$R"
O=gpurun_out
echo "== 1. GPU suite"
python -m pytest tests -m gpu -x -q 2>&1 | tail -4
echo "== 2. A/B"
B="python bench.py --no-cpu-baseline --no-flow-lines"
val() { $B "$@" 2>/dev/null | tail -1 | python -c "import json,sys; j=json.loads(sys.stdin.read()); k=j['kernel_ms_per_call']; print('%.3f M/s   lone calls: prove %.3f ms, batch verify %.3f ms (combine %.3f)' % (j['value']/1e6, k['prove']['total'], k['batch_verify']['total'], k['batch_verify'].get('combine', -1)))"; }
cp zkp_amd/libzkp_mi355x.so /tmp/shipped.so
for r in 1 2; do
  cp tools/x/variants/libzkp_quad_horner.so zkp_amd/libzkp_mi355x.so
  echo "quad  20: $(val --steps 20 --warmup 5)"; echo "quad 200: $(val --steps 200)"
  echo "quad K=1 x 1 stream: $(val --steps 20 --no-sustained --batches-per-call 1 --streams 1 | cut -c1-12)   K=5 x 1: $(val --steps 20 --no-sustained --batches-per-call 5 --streams 1 | cut -c1-12)   K=1 x 4: $(val --steps 20 --no-sustained --batches-per-call 1 --streams 4 | cut -c1-12)"
  cp /tmp/shipped.so zkp_amd/libzkp_mi355x.so
  echo "row   20: $(val --steps 20 --warmup 5)"; echo "row  200: $(val --steps 200)"
  echo "row  K=1 x 1 stream: $(val --steps 20 --no-sustained --batches-per-call 1 --streams 1 | cut -c1-12)   K=5 x 1: $(val --steps 20 --no-sustained --batches-per-call 5 --streams 1 | cut -c1-12)   K=1 x 4: $(val --steps 20 --no-sustained --batches-per-call 1 --streams 4 | cut -c1-12)"
done
echo "== 3. kernel clock"
( cd /tmp && rm -rf $R/$O/rowprof && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/rowprof -o row -- python $R/bench.py --no-cpu-baseline --no-flow-lines --no-sustained --steps 20 --warmup 5 --streams 1 > /dev/null 2>&1 )
python - <<'PY'
import csv, glob
for f in glob.glob("gpurun_out/rowprof/**/row_kernel_stats.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if any(s in row["Name"] for s in ("k_pip_combine", "k_encode_invert", "k_pip_reduce_lvl")):
            print(row["Name"][:40], row["Calls"], "avg ns", row["AverageNs"], "min", row["MinNs"])
PY
rm -rf $O/rowprof
