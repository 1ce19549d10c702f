#!/bin/bash
# Round 5, tenth GPU call: the whole GPU suite after the pipe's discard-on-destroy, the smoke entry, the wavefront-cycle file with its final commentary, a bench line.
export TMPDIR=/tmp
R="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$R"
O=gpurun_out
mkdir -p $O
python -m pytest tests -m gpu -x -q 2>&1 | tail -4
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
python tools/ct_check.py --cycles > $O/r05_constant_time_wave_cycles.txt 2>&1; tail -14 $O/r05_constant_time_wave_cycles.txt | cut -c1-200
python bench.py --steps 20 --warmup 5 > $O/r05_bench_check.json 2> $O/r05_bench_check.err; tail -2 $O/r05_bench_check.err
python - <<'PY'
import json
j = json.loads(open("gpurun_out/r05_bench_check.json").read().strip().splitlines()[-1])
print("value %.3f M  sustained %.3f M  pmc %s  parity %s" % (j["value"] / 1e6, j["sustained"]["value"] / 1e6, bool(j["pmc_source"]), j["parity_checked"]["equal"]))
print("simd", j["cpu_baseline"]["simd"]["value"], (j["cpu_baseline"]["simd"].get("avx2") or {}).get("value"), (j["cpu_baseline"]["simd"].get("avx2") or {}).get("one_limb_per_vector"))
PY
