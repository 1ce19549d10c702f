#!/bin/bash
# Round 5: three more latency cuts, each against the build without it (tools/x/variants/libzkp_v_*.so differ from the shipped build by ONE -DZKP_AB_* switch):
#   bperm        rowfe.h moves coordinates between rows with ds_bpermute_b32 instead of v_permlane32_swap / v_permlane16_swap
#   plainscatter k_pip_tile_scatter with the plain (tile, window) block order instead of whole windows per XCD
#   laneinvert   k_encode_invert's inversion on one lane instead of one limb per lane
export TMPDIR=/tmp
R="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$R"
O=gpurun_out
V=tools/x/variants
echo "== 0. what the gfx950 lane swaps do"
hipcc --offload-arch=gfx950 -O2 tools/microbench/permlane_probe.hip -o /tmp/permlane_probe 2>&1 | tail -3
/tmp/permlane_probe | tee $O/r05_permlane_probe.txt
if ! ( grep -q "swap16(h0,h0)\[0\] *rows from: 0 0 0 0 " $O/r05_permlane_probe.txt && grep -q "swap16(h0,h0)\[1\] *rows from: 1 1 1 1 " $O/r05_permlane_probe.txt && grep -q "swap16(h1,h1)\[0\] *rows from: 2 2 2 2 " $O/r05_permlane_probe.txt && grep -q "swap16(h1,h1)\[1\] *rows from: 3 3 3 3 " $O/r05_permlane_probe.txt ); then
  echo "!! the swaps do not do what rowfe.h assumes: continuing on the ds_bpermute build"
  cp $V/libzkp_v_bperm.so zkp_amd/libzkp_mi355x.so; cp $V/libzkp_v_bperm_hooks.so zkp_amd/libzkp_mi355x_testhooks.so
fi
echo "== 1. GPU suite"
python -m pytest tests -m gpu -x -q 2>&1 | tail -4
echo "== 2. A/B"
B="python bench.py --no-cpu-baseline --no-flow-lines"
val() { $B "$@" 2>/dev/null | tail -1 | python -c "import json,sys; j=json.loads(sys.stdin.read()); k=j['kernel_ms_per_call']; print('%.3f M/s   lone calls: prove %.3f ms (reduce %.3f), batch verify %.3f ms (sort %.3f combine %.3f)' % (j['value']/1e6, k['prove']['total'], k['prove'].get('reduce', -1), k['batch_verify']['total'], k['batch_verify'].get('sort', -1), k['batch_verify'].get('combine', -1)))"; }
cp zkp_amd/libzkp_mi355x.so /tmp/shipped.so
for r in 1 2; do
  for v in shipped bperm plainscatter laneinvert; do
    if [ $v = shipped ]; then cp /tmp/shipped.so zkp_amd/libzkp_mi355x.so; else cp $V/libzkp_v_$v.so zkp_amd/libzkp_mi355x.so; fi
    echo "$v  20: $(val --steps 20 --warmup 5)"; echo "$v 200: $(val --steps 200)"
    echo "$v K=1 x 1 stream: $(val --steps 20 --no-sustained --batches-per-call 1 --streams 1 | cut -c1-12)   K=5 x 1: $(val --steps 20 --no-sustained --batches-per-call 5 --streams 1 | cut -c1-12)"
  done
done
cp /tmp/shipped.so zkp_amd/libzkp_mi355x.so
echo "== 3. kernel clock (one stream)"
for v in shipped plainscatter; do
  if [ $v = shipped ]; then cp /tmp/shipped.so zkp_amd/libzkp_mi355x.so; else cp $V/libzkp_v_$v.so zkp_amd/libzkp_mi355x.so; fi
  ( cd /tmp && rm -rf $R/$O/rowprof && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/rowprof -o row -- python $R/bench.py --no-cpu-baseline --no-flow-lines --no-sustained --steps 20 --warmup 5 --streams 1 > /dev/null 2>&1 )
  python - "$v" <<'PY'
import csv, glob, sys
for f in glob.glob("gpurun_out/rowprof/**/row_kernel_stats.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if any(s in row["Name"] for s in ("k_pip_combine", "k_encode_invert", "k_pip_tile_scatter", "k_pip_bucket_part")):
            print(sys.argv[1], row["Name"][:40], row["Calls"], "avg ns", row["AverageNs"], "min", row["MinNs"])
PY
  rm -rf $O/rowprof
done
cp /tmp/shipped.so zkp_amd/libzkp_mi355x.so
