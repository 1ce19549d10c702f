#!/bin/bash
# Round 5: the driver's command and the default run once more on whatever box this call gets (the box-to-box spread of the final build): $1 = tag of the output files
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out
T=${1:-box_c}
python bench.py --steps 20 --warmup 5 > $O/r05_bench_1gpu_steps20_$T.json 2> $O/err_$T.log
python bench.py --no-cpu-baseline > $O/r05_bench_1gpu_$T.json 2>> $O/err_$T.log
python - "$T" <<'PY'
import json, sys
t = sys.argv[1]
for f in ("1gpu_steps20_" + t, "1gpu_" + t):
    j = json.loads(open("gpurun_out/r05_bench_%s.json" % f).read().strip().splitlines()[-1])
    print(f, "value %.3f M" % (j["value"] / 1e6), "ms/step %.4f" % j["ms_per_step"], "step_valu", (j.get("step_valu") or {}).get("frac"), "sustained %.3f M" % (j["sustained"]["value"] / 1e6) if "sustained" in j else "")
PY
