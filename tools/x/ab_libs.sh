#!/bin/bash
# bench.py under two builds of the kernel library, alternated on ONE box: usage tools/x/ab_libs.sh A.so B.so [rounds]   (prints value and the lone prove call's kinds)
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
cp zkp_amd/libzkp_mi355x.so /tmp/shipped.so
val() { python bench.py --no-cpu-baseline --no-flow-lines "$@" 2>/dev/null | tail -1 | python -c "import json,sys; j=json.loads(sys.stdin.read()); k=j['kernel_ms_per_call']['prove']; print('%.3f M/s   lone prove call: tables %.3f terms %.3f reduce %.3f total %.3f ms' % (j['value']/1e6, k['tables'], k['terms'], k['reduce'], k['total']))"; }
for r in $(seq 1 ${3:-2}); do
  for lib in "$1" "$2"; do
    cp "$lib" zkp_amd/libzkp_mi355x.so
    echo "$(basename $lib)  20: $(val --steps 20 --warmup 5)"
    echo "$(basename $lib) 200: $(val --steps 200)"
  done
done
cp /tmp/shipped.so zkp_amd/libzkp_mi355x.so
