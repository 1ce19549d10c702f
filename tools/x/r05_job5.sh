#!/bin/bash
# Round 5, fifth GPU call: pipe tests after the fixes, the producer / consumer table builder (tools/x/variants/libzkp_pc.so, -DZKP_TABLES_PC=1) against the shipped one.
export TMPDIR=/tmp
R="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$R"
O=gpurun_out
mkdir -p $O
echo "== 1. pipe / job / thread tests"
python -m pytest tests/test_gpu_pipe.py tests/test_gpu_threads.py tests/test_gpu_sharding.py -m gpu -x -q -s 2>&1 | grep -v "^$" | tail -6
echo "== 2. producer / consumer table builder"
B="python bench.py --no-cpu-baseline --no-flow-lines"
val() { $B "$@" 2>/dev/null | tail -1 | python -c "import json,sys; j=json.loads(sys.stdin.read()); k=j['kernel_ms_per_call']['prove']; print('%.3f M/s   lone prove call: tables %.3f transcript %.3f terms %.3f reduce %.3f total %.3f ms; lone batch verify %.3f ms' % (j['value']/1e6, k['tables'], k.get('transcript', 0), k['terms'], k['reduce'], k['total'], j['kernel_ms_per_call']['batch_verify']['total']))"; }
cp zkp_amd/libzkp_mi355x.so /tmp/shipped.so
cp tools/x/variants/libzkp_pc.so zkp_amd/libzkp_mi355x.so
python -m pytest tests/test_gpu_toolbox.py tests/test_gpu_fused.py tests/test_gpu_device_entry.py -m gpu -x -q 2>&1 | tail -2
for r in 1 2 3; do
  cp /tmp/shipped.so zkp_amd/libzkp_mi355x.so
  echo "shipped   20: $(val --steps 20 --warmup 5)"; echo "shipped  200: $(val --steps 200)"
  cp tools/x/variants/libzkp_pc.so zkp_amd/libzkp_mi355x.so
  echo "pc        20: $(val --steps 20 --warmup 5)"; echo "pc       200: $(val --steps 200)"
done
for lib in /tmp/shipped.so tools/x/variants/libzkp_pc.so; do
  cp $lib zkp_amd/libzkp_mi355x.so
  echo "$(basename $lib) 1000: $(val)"
  echo "$(basename $lib) lone chains: K=1 x 1 stream: $(val --steps 20 --batches-per-call 1 --streams 1 | cut -c1-12)  K=5 x 1 stream: $(val --steps 20 --batches-per-call 5 --streams 1 | cut -c1-12)  K=1 x 4 streams: $(val --steps 20 --batches-per-call 1 --streams 4 | cut -c1-12)"
done
cp /tmp/shipped.so zkp_amd/libzkp_mi355x.so
