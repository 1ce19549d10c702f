#!/bin/bash
# One gpurun call = one box (round 6): the profiles of every bench configuration (kernel traces, PMC passes, one-stream traces: tools/collect_profiles.sh), the lone-call
# (K = 1) traces on both schedules, the micro-benchmark and traces behind this round's decisions, then the bench lines that read the counter files -> gpurun_out/r06_*;
# copy what is to be judged into profiles/.  Run tools/opcode_mix.py first (it keys the static mix to the sources).
export TMPDIR=/tmp
O=gpurun_out
mkdir -p $O
bash tools/collect_profiles.sh r06 2 20 > $O/r06_collect_cfg2_k5.log 2>&1
bash tools/collect_profiles.sh r06 2 200 > $O/r06_collect_cfg2_k50.log 2>&1
bash tools/collect_profiles.sh r06 3 2 > $O/r06_collect_cfg3.log 2>&1
bash tools/collect_profiles.sh r06 4share 2 > $O/r06_collect_cfg4share.log 2>&1
bash tools/collect_profiles.sh r06 5share 8 > $O/r06_collect_cfg5share.log 2>&1
cp $O/r06_pmc_counters_cfg*.json profiles/
# the reference-shaped call: ONE batch of 4096 proofs per call on one stream, both schedules (one hardware queue: every kernel of the trace is serialised, so the
# latency schedule's two streams show as one chain -- its concurrent timing is lone_call.latency_schedule of the bench line)
for sched in throughput latency; do
  opt=""; [ $sched = latency ] && opt="--engine-opt 5=2"
  B="python bench.py --config 2 --steps 20 --warmup 2 --no-cpu-baseline --no-flow-lines --no-sustained --streams 1 --max-hw-queues 1 --batches-per-call 1 $opt"
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/r06_k1_prof -o k1 -- $B > $O/r06_k1_$sched.log 2>&1
  python - <<PY > $O/r06_kernel_stats_cfg2_k1_${sched}_schedule_one_stream.txt
import csv, sys
sys.path.insert(0, ".")
import bench
rows = list(csv.DictReader(open("$O/r06_k1_prof/k1_kernel_stats.csv")))
print("# rocprofv3 --kernel-trace --stats -- $B   (MI355X; kernel sources sha256 %s)" % bench.source_sha256()[:16])
print("%-46s %6s %12s %12s %12s %8s" % ("kernel", "calls", "avg_us", "min_us", "max_us", "pct"))
for r in rows:
    print("%-46s %6s %12.1f %12.1f %12.1f %8s" % (r["Name"].split("(")[0].replace("void ", "")[:46], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3, r["Percentage"]))
PY
  rm -rf $O/r06_k1_prof
done
# a synchronous call on host buffers: host and device timeline (hip api + kernels + copies)
bash tools/x/sync_call_timeline.sh > $O/r06_sync_call_timeline.txt 2>&1
# the pipe against the resident loop
rocprofv3 --hip-trace --kernel-trace --memory-copy-trace --output-format csv -d $O/r06_gap_pipe -o p -- python tools/e2e_pipe_bench.py --shapes 10x6 --order pinned --jobs 36 > $O/r06_gap_pipe.log 2>&1
rocprofv3 --hip-trace --kernel-trace --memory-copy-trace --output-format csv -d $O/r06_gap_res -o r -- python bench.py --steps 200 --batches-per-call 5 --streams 4 --warmup 2 --no-cpu-baseline --no-flow-lines --no-sustained > $O/r06_gap_res.log 2>&1
python tools/pipe_gap_trace.py $O/r06_gap_pipe $O/r06_gap_res > $O/r06_pipe_gap_trace.txt 2>&1
rm -rf $O/r06_gap_pipe $O/r06_gap_res
# Keccak-f[1600] on a lone wavefront: the shipped lane-pair layout, unrolled, and one word per lane over the crossbar
hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/microbench/keccak_lat.hip -o tools/microbench/keccak_lat 2>/dev/null
tools/microbench/keccak_lat > $O/r06_keccak_microbench.txt 2>&1
# constant-time evidence for the whole prover flow (secrets = witnesses and entropy)
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_WAVES --output-format csv -d $O/r06_ctf -o flow -- python tools/ct_check_flow.py > $O/r06_ctf1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_INSTS_SMEM SQ_INSTS_FLAT --output-format csv -d $O/r06_ctf -o flow2 -- python tools/ct_check_flow.py > $O/r06_ctf2.log 2>&1
python tools/ct_check_flow.py --summarise $(ls $O/r06_ctf/*flow_counter_collection.csv $O/r06_ctf/*flow2_counter_collection.csv 2>/dev/null) > $O/r06_constant_time_flow_counters.txt
rm -rf $O/r06_ctf
# verify_compact alone: where its time goes, kernel by kernel, in issue slots
rocprofv3 --kernel-trace --output-format csv -d $O/r06_vc -o vc -- python tools/verify_compact_profile.py > $O/r06_vc1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_VALU_INT64 SQ_ACTIVE_INST_VALU2 SQ_WAVES GRBM_GUI_ACTIVE --output-format csv -d $O/r06_vc -o vcp -- python tools/verify_compact_profile.py > $O/r06_vc2.log 2>&1
python tools/verify_compact_profile.py --summarise $(ls $O/r06_vc/*vc_kernel_trace.csv | head -1) $(ls $O/r06_vc/*vcp_counter_collection.csv | head -1) > $O/r06_verify_compact_accounting.txt 2>&1
rm -rf $O/r06_vc
python tools/e2e_toolbox_bench.py 4096 16384 > $O/r06_e2e_toolbox_host_included.txt 2>&1
python bench.py --steps 20 --warmup 5 > $O/r06_bench_1gpu_steps20.json 2> $O/err1.log
python bench.py > $O/r06_bench_1gpu.json 2> $O/err2.log
python bench.py --config 3 > $O/r06_bench_cfg3.json 2> $O/err3.log
python bench.py --config 4share > $O/r06_bench_cfg4share.json 2> $O/err4.log
python bench.py --config 5share > $O/r06_bench_cfg5share.json 2> $O/err5.log
python bench.py --config 5share --w64-form constraints --no-cpu-baseline > $O/r06_bench_cfg5share_constraints.json 2> $O/err6.log
tail -2 $O/err*.log | grep -v "^$" | head -20
python - <<'PY'
import json
for f in ("1gpu_steps20","1gpu","cfg3","cfg4share","cfg5share","cfg5share_constraints"):
    try:
        j=json.loads(open("gpurun_out/r06_bench_%s.json"%f).read().strip().splitlines()[-1])
        sv=j.get("step_valu") or {}
        print(f, "value",round(j["value"]), "ms/step",round(j["ms_per_step"],3), "step_valu.frac", sv.get("frac"), "dom", j["roofline"]["kernel"][:50], "busy", j["roofline"].get("dominant_kernel_valu_busy"), "traffic", j["roofline"].get("traffic"), "pmc", bool(j["pmc_source"]))
        if "roofline_valu" in j: print("   roofline_valu", {k: (round(v, 4) if isinstance(v, float) else v) for k, v in j["roofline_valu"].items() if k not in ("note", "algorithm_by_phase", "floor_by_flow")})
        if "sustained" in j: print("   sustained", round(j["sustained"]["value"]), "ms/step", round(j["sustained"]["ms_per_step"], 4), "step_valu", (j["sustained"].get("step_valu") or {}).get("frac"))
        if "lone_call" in j: print("   lone_call", j["lone_call"].get("throughput_schedule"), j["lone_call"].get("latency_schedule"))
        if "e2e_host_buffers" in j: print("   e2e", round(j["e2e_host_buffers"]["proofs_per_s"]), round(j["e2e_host_buffers"]["pipelined"]["proofs_per_s"]), round(j["e2e_host_buffers"]["pipelined_staged"]["proofs_per_s"]), "threads", round(j["e2e_host_buffers"].get("threads", {}).get("proofs_per_s", 0)))
    except Exception as e: print(f,"ERR",e)
PY
