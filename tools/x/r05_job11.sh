#!/bin/bash
# A/B: k_tables_transcript_pc capped at 168 / 128 VGPRs (3 / 4 wavefronts per SIMD, with spills) against the shipped 247-VGPR build
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
B="python bench.py --no-cpu-baseline --no-flow-lines"
val() { $B "$@" 2>/dev/null | tail -1 | python -c "import json,sys; j=json.loads(sys.stdin.read()); k=j['kernel_ms_per_call']['prove']; print('%.3f M/s   lone prove call: tables %.3f terms %.3f total %.3f ms' % (j['value']/1e6, k['tables'], k['terms'], k['total']))"; }
cp zkp_amd/libzkp_mi355x.so /tmp/shipped.so
for r in 1 2 3; do
  for v in shipped tt3 tt4; do
    if [ $v = shipped ]; then cp /tmp/shipped.so zkp_amd/libzkp_mi355x.so; else cp tools/x/variants/libzkp_$v.so zkp_amd/libzkp_mi355x.so; fi
    echo "$v   20: $(val --steps 20 --warmup 5)"
  done
done
for v in shipped tt3; do
  if [ $v = shipped ]; then cp /tmp/shipped.so zkp_amd/libzkp_mi355x.so; else cp tools/x/variants/libzkp_$v.so zkp_amd/libzkp_mi355x.so; fi
  echo "$v  200: $(val --steps 200)"
  echo "$v K=1 x 4 streams: $(val --steps 20 --batches-per-call 1 --streams 4 | cut -c1-12)   K=1 x 1: $(val --steps 20 --batches-per-call 1 --streams 1 | cut -c1-12)"
done
cp /tmp/shipped.so zkp_amd/libzkp_mi355x.so
