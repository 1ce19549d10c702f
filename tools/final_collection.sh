#!/bin/bash
# One gpurun call = one box: the profiles of every bench configuration (kernel traces, PMC passes, one-stream traces: tools/collect_profiles.sh), then the five
# bench lines that read them -> gpurun_out/r05_*; copy what is to be judged into profiles/.  Run tools/opcode_mix.py first (it keys the static mix to the sources).
export TMPDIR=/tmp
mkdir -p gpurun_out
bash tools/collect_profiles.sh r05 2 20 > gpurun_out/r05_collect_cfg2_k5.log 2>&1
bash tools/collect_profiles.sh r05 2 200 > gpurun_out/r05_collect_cfg2_k50.log 2>&1
bash tools/collect_profiles.sh r05 3 2 > gpurun_out/r05_collect_cfg3.log 2>&1
bash tools/collect_profiles.sh r05 4share 2 > gpurun_out/r05_collect_cfg4share.log 2>&1
bash tools/collect_profiles.sh r05 5share 8 > gpurun_out/r05_collect_cfg5share.log 2>&1
cp gpurun_out/r05_pmc_counters_cfg*.json profiles/
python bench.py --steps 20 --warmup 5 > gpurun_out/r05_bench_1gpu_steps20.json 2> gpurun_out/err1.log
python bench.py > gpurun_out/r05_bench_1gpu.json 2> gpurun_out/err2.log
python bench.py --config 3 > gpurun_out/r05_bench_cfg3.json 2> gpurun_out/err3.log
python bench.py --config 4share > gpurun_out/r05_bench_cfg4share.json 2> gpurun_out/err4.log
python bench.py --config 5share > gpurun_out/r05_bench_cfg5share.json 2> gpurun_out/err5.log
python bench.py --config 5share --w64-form constraints --no-cpu-baseline > gpurun_out/r05_bench_cfg5share_constraints.json 2> gpurun_out/err6.log
python bench.py --config 5share --w64-form constraints2 --no-cpu-baseline > gpurun_out/r05_bench_cfg5share_constraints2.json 2> gpurun_out/err7.log
tail -2 gpurun_out/err*.log | grep -v "^$" | head -20
python - <<'PY'
import json
for f in ("1gpu_steps20","1gpu","cfg3","cfg4share","cfg5share","cfg5share_constraints","cfg5share_constraints2"):
    try:
        j=json.loads(open("gpurun_out/r05_bench_%s.json"%f).read().strip().splitlines()[-1])
        sv=j.get("step_valu") or {}
        print(f, "value",round(j["value"]), "ms/step",round(j["ms_per_step"],3), "step_valu.frac", sv.get("frac"), "dom", j["roofline"]["kernel"][:50], "busy", j["roofline"].get("dominant_kernel_valu_busy"), "traffic", j["roofline"].get("traffic"), "pmc", bool(j["pmc_source"]))
        if "sustained" in j: print("   sustained", round(j["sustained"]["value"]), "ms/step", round(j["sustained"]["ms_per_step"], 4), "step_valu", (j["sustained"].get("step_valu") or {}).get("frac"), "ct_schedule:", j["config"]["ct_schedule"][:40])
        if "e2e_host_buffers" in j: print("   e2e", round(j["e2e_host_buffers"]["proofs_per_s"]), round(j["e2e_host_buffers"]["pipelined"]["proofs_per_s"]), round(j["e2e_host_buffers"]["pipelined_staged"]["proofs_per_s"]), "threads", round(j["e2e_host_buffers"].get("threads", {}).get("proofs_per_s", 0)))
    except Exception as e: print(f,"ERR",e)
PY
