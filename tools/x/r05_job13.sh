#!/bin/bash
# Round 5: the conditional negation without its carry (ge25519.h) against the build before (tools/x/variants/libzkp_cneg_carry.so), then suite + collection on the new sources
export TMPDIR=/tmp
R="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$R"
O=gpurun_out
B="python bench.py --no-cpu-baseline --no-flow-lines"
val() { $B "$@" 2>/dev/null | tail -1 | python -c "import json,sys; j=json.loads(sys.stdin.read()); k=j['kernel_ms_per_call']['prove']; print('%.3f M/s   lone prove call: terms %.3f total %.3f ms' % (j['value']/1e6, k['terms'], k['total']))"; }
cp zkp_amd/libzkp_mi355x.so /tmp/shipped.so
for r in 1 2 3; do
  cp tools/x/variants/libzkp_cneg_carry.so zkp_amd/libzkp_mi355x.so
  echo "carry     20: $(val --steps 20 --warmup 5)"; echo "carry    200: $(val --steps 200)"
  cp /tmp/shipped.so zkp_amd/libzkp_mi355x.so
  echo "no carry  20: $(val --steps 20 --warmup 5)"; echo "no carry 200: $(val --steps 200)"
done
bash tools/x/r05_job9.sh
