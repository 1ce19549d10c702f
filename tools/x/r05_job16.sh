#!/bin/bash
# Round 5: k_pip_bucket_part with whole runs of windows per XCD against the plain block order (tools/x/variants/libzkp_v_plainbuckets.so), HBM bytes of both,
# and the kernel list of a lone K = 1 call
export TMPDIR=/tmp
R="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$R"
O=gpurun_out
V=tools/x/variants
echo "== 1. GPU suite"
python -m pytest tests -m gpu -x -q 2>&1 | tail -4
echo "== 2. A/B"
B="python bench.py --no-cpu-baseline --no-flow-lines"
val() { $B "$@" 2>/dev/null | tail -1 | python -c "import json,sys; j=json.loads(sys.stdin.read()); k=j['kernel_ms_per_call']; print('%.3f M/s   lone calls: prove %.3f ms, batch verify %.3f ms (bucket %.3f sort %.3f combine %.3f)' % (j['value']/1e6, k['prove']['total'], k['batch_verify']['total'], k['batch_verify'].get('bucket', -1), k['batch_verify'].get('sort', -1), k['batch_verify'].get('combine', -1)))"; }
cp zkp_amd/libzkp_mi355x.so /tmp/shipped.so
for r in 1 2 3; do
  for v in shipped plainbuckets; do
    if [ $v = shipped ]; then cp /tmp/shipped.so zkp_amd/libzkp_mi355x.so; else cp $V/libzkp_v_$v.so zkp_amd/libzkp_mi355x.so; fi
    echo "$v  20: $(val --steps 20 --warmup 5)"; echo "$v 200: $(val --steps 200)"
  done
done
for v in shipped plainbuckets; do
  if [ $v = shipped ]; then cp /tmp/shipped.so zkp_amd/libzkp_mi355x.so; else cp $V/libzkp_v_$v.so zkp_amd/libzkp_mi355x.so; fi
  echo "$v K=1 x 1 stream: $(val --steps 20 --no-sustained --batches-per-call 1 --streams 1 | cut -c1-12)   K=5 x 1: $(val --steps 20 --no-sustained --batches-per-call 5 --streams 1 | cut -c1-12)   config 3: $($B --config 3 2>/dev/null | tail -1 | python -c "import json,sys; print('%.1f M terms/s' % (json.loads(sys.stdin.read())['value']/1e6))")"
done
echo "== 3. HBM bytes of the bucket kernel (one stream, K = 5)"
for v in shipped plainbuckets; do
  if [ $v = shipped ]; then cp /tmp/shipped.so zkp_amd/libzkp_mi355x.so; else cp $V/libzkp_v_$v.so zkp_amd/libzkp_mi355x.so; fi
  ( cd /tmp && rm -rf $R/$O/bprof && timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE WRITE_SIZE --output-format csv -d $R/$O/bprof -o b -- python $R/bench.py --no-cpu-baseline --no-flow-lines --no-sustained --steps 20 --warmup 5 --streams 1 > /dev/null 2>&1 )
  python - "$v" <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(lambda: [0, 0.0, 0.0])
for f in glob.glob("gpurun_out/bprof/**/b_counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        n = row["Kernel_Name"][:34]
        if any(s in n for s in ("k_pip_bucket_part", "k_pip_tile_scatter", "k_terms_split<true")):
            a = acc[n]
            if row["Counter_Name"] == "FETCH_SIZE": a[0] += 1; a[1] += float(row["Counter_Value"])
            if row["Counter_Name"] == "WRITE_SIZE": a[2] += float(row["Counter_Value"])
for n, a in acc.items():
    print(sys.argv[1], n, "launches", a[0], "FETCH_SIZE per launch %.1f MB (raw counter x 1 KB... as rocprofv3 reports it: %.4g)" % (a[1] / max(a[0], 1) / 1e3, a[1] / max(a[0], 1)), "WRITE_SIZE %.4g" % (a[2] / max(a[0], 1)))
PY
  rm -rf $O/bprof
done
cp /tmp/shipped.so zkp_amd/libzkp_mi355x.so
echo "== 4. kernels of a lone K = 1 call (one stream)"
( cd /tmp && rm -rf $R/$O/kprof && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/kprof -o k1 -- python $R/bench.py --no-cpu-baseline --no-flow-lines --no-sustained --steps 20 --warmup 5 --streams 1 --batches-per-call 1 > /dev/null 2>&1 )
python - <<'PY'
import csv, glob
for f in glob.glob("gpurun_out/kprof/**/k1_kernel_stats.csv", recursive=True):
    rows = sorted(csv.DictReader(open(f)), key=lambda r: -float(r["TotalDurationNs"]))
    for row in rows[:22]:
        print("%-44s calls %4s  avg us %8.1f  total %% %s" % (row["Name"][:44], row["Calls"], float(row["AverageNs"]) / 1e3, row["Percentage"]))
PY
rm -rf $O/kprof
