#!/usr/bin/env python3
"""Per-kernel averages of rocprofv3 --pmc counter_collection CSVs (one or more passes) -> text + json.

    python tools/pmc_summary.py OUT.json BENCH_LINE.json pass1_counter_collection.csv [pass2_counter_collection.csv ...]

BENCH_LINE.json = the JSON line the profiled `python bench.py ...` command printed (workload shape: config, batch, batches per
call, steps).  HBM traffic per launch follows MI355X_MICROARCH.md section HBM: FETCH_SIZE / WRITE_SIZE are in KiB of fabric
requests; on gfx950 FETCH_SIZE reports half of the bytes of wide (16 B/lane) streaming reads, so reads are counted as
2 x FETCH_SIZE (an upper bound for kernels whose reads are narrower gathers).

valu_busy (cycle-weighted): 4 x (SQ_INSTS_VALU - SQ_ACTIVE_INST_VALU2) / (1024 SIMDs x GRBM_GUI_ACTIVE / 8 XCDs) -- VALU issue slots used / clock
cycles that went by, all three counters from ONE pass.  A SIMD issues one VALU instruction per 4-cycle slot, or TWO when both belong to the
2-cycle class (v_and / v_add_u32 / v_mov ...): SQ_ACTIVE_INST_VALU2 counts those second instructions (0.48 per instruction on a pure 2-cycle stream, 0
on a pure 4-cycle one, 0.02 - 0.07 on alternating patterns, whose 2-cycle instructions then cost a full slot each: profiles/r04_valu_mix_microbench.txt).
The static share of 2-cycle opcodes (tools/opcode_mix.py) is printed next to the co-issued share for information only.  GRBM_GUI_ACTIVE covers a
little more than the kernel (tens of microseconds of dispatch around it): the busy fraction of kernels shorter than ~100 us is understated, and
sclk_ghz = GRBM_GUI_ACTIVE / 8 / the dispatch's duration in that pass is only printed as a clock for longer ones."""
import collections
import csv
import json
import os
import sys

# the kernel that runs once per CALL of a workload's step loop, and how many bench steps one such launch stands for
SIMDS, XCDS, NAMEPLATE_GHZ = 1024, 8, 2.4            # 256 CUs x 4 SIMDs; GRBM_GUI_ACTIVE is summed over the 8 XCDs; hipDeviceProp.clockRate
UNIT = {"2": ("k_terms_split<true", None), "4share": ("k_terms_split<true", 1.0), "5share": ("k_terms_split<true", 1.0), "3": ("k_pip_combine", 0.5)}


def kname(r):
    return r["Kernel_Name"].split("(")[0].replace("void ", "")


def main():
    out, line, files = sys.argv[1], sys.argv[2], sys.argv[3:]
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    sha = bench.source_sha256()
    bl = json.loads([l for l in open(line).read().splitlines() if l.startswith("{")][-1])
    cfg = bl["config"]["baseline_config"]
    k_per_call = bl["config"].get("batches_per_call", 1)
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    gui_ns = collections.defaultdict(list)              # kernel -> durations (ns) of its dispatches in the pass that counted GRBM_GUI_ACTIVE
    one_stream = collections.defaultdict(list)          # kernel -> durations (ns) from a kernel trace of the same command on ONE stream
    for fn in files:
        rows = list(csv.DictReader(open(fn)))
        if rows and "Counter_Name" not in rows[0]:       # a *_kernel_trace.csv: durations of lone launches
            for r in rows:
                one_stream[kname(r)].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
            continue
        for r in rows:
            agg[kname(r)][r["Counter_Name"]].append(float(r["Counter_Value"]))
            if r["Counter_Name"] == "GRBM_GUI_ACTIVE" and r.get("End_Timestamp"):
                gui_ns[kname(r)].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    # static opcode mix of each kernel (tools/opcode_mix.py): printed for information, the busy fractions are counted in issue slots by the hardware
    mix = {}
    mix_path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), bench.OPCODE_MIX)
    if os.path.exists(mix_path):
        mj = json.load(open(mix_path))
        if mj.get("_source_sha256") == sha:
            mix = mj
    res = {}
    print("# workload: --config %s, %d proofs per batch, %d batch(es) per call, %d timed steps; kernel sources sha256 %s" % (cfg, bl["config"]["batch_per_gpu"], k_per_call, bl["steps"], sha[:16]))
    print("# valu_busy = 4 x (SQ_INSTS_VALU - SQ_ACTIVE_INST_VALU2) / (%d SIMDs x GRBM_GUI_ACTIVE / %d)  -- VALU issue slots used / clock cycles elapsed, one PMC pass, <= 1   (static opcode mix: %s, %s)"
          % (SIMDS, XCDS, bench.OPCODE_MIX, "sha matches" if mix else "missing or stale"))
    print("%-34s %s" % ("kernel", "counter averages per launch (n launches)"))
    for k in sorted(agg):
        if "k_" not in k:
            continue
        row = {c: sum(v) / len(v) for c, v in agg[k].items()}
        row["launches"] = max(len(v) for v in agg[k].values())
        if "FETCH_SIZE" in row and "WRITE_SIZE" in row:
            row["hbm_bytes_per_launch"] = (2.0 * row["FETCH_SIZE"] + row["WRITE_SIZE"]) * 1024.0
        if one_stream.get(k):
            row["avg_us_one_stream"] = sum(one_stream[k]) / len(one_stream[k]) / 1e3
        m = mix.get(k)
        if m:
            row["share_int64_static"] = m["share_int64_static"]
            row["share_mad_u64_static"] = m.get("share_mad_u64_static")
            row["share_2cycle_static"] = m["share_2cycle_class"]
        if "SQ_INSTS_VALU" in row and "SQ_ACTIVE_INST_VALU2" in row:
            # issue slots: one per instruction, minus the instructions that shared a slot with another 2-cycle-class instruction
            row["valu_issue_cycles"] = 4.0 * (row["SQ_INSTS_VALU"] - row["SQ_ACTIVE_INST_VALU2"])            # summed over all SIMDs
            row["share_co_issued"] = row["SQ_ACTIVE_INST_VALU2"] / row["SQ_INSTS_VALU"] if row["SQ_INSTS_VALU"] else 0.0
            if row.get("GRBM_GUI_ACTIVE"):
                row["valu_busy"] = row["valu_issue_cycles"] / (SIMDS * row["GRBM_GUI_ACTIVE"] / XCDS)
                if gui_ns.get(k):
                    row["us_in_pmc_pass"] = sum(gui_ns[k]) / len(gui_ns[k]) / 1e3
                    row["sclk_ghz"] = row["GRBM_GUI_ACTIVE"] / XCDS / (row["us_in_pmc_pass"] * 1e3)
        if row.get("SQ_WAIT_ANY") is not None and row.get("SQ_WAVE_CYCLES"):
            # wavefront-cycles parked at s_waitcnt / barriers (memory, LDS-DMA, barrier) as a share of all wavefront-cycles -- per WAVEFRONT: with two
            # wavefronts per SIMD one's wait is the other's issue slot, so the SIMD idles for less than this (1 - valu_busy is the bound on that)
            row["wave_parked_frac"] = row["SQ_WAIT_ANY"] / row["SQ_WAVE_CYCLES"]
        if row.get("TCC_HIT_sum") is not None and (row.get("TCC_HIT_sum", 0) + row.get("TCC_MISS_sum", 0)) > 0:
            row["l2_hit_rate"] = row["TCC_HIT_sum"] / (row["TCC_HIT_sum"] + row["TCC_MISS_sum"])
        if "SQ_INSTS_VALU_INT64" in row and row.get("SQ_INSTS_VALU"):
            row["share_int64_dynamic"] = row["SQ_INSTS_VALU_INT64"] / row["SQ_INSTS_VALU"]
            # executed v_mad_u64_u32 (wave-instructions): the hardware counts the 64-bit integer class (mads, v_lshl_add_u64, 64-bit shifts); the class's static
            # composition splits it -- mads are 84 - 100 % of the class in every kernel of the path (profiles/r06_opcode_mix.json)
            if row.get("share_mad_u64_static") is not None and row.get("share_int64_static"):
                row["mad_u64_wave_instructions"] = row["SQ_INSTS_VALU_INT64"] * row["share_mad_u64_static"] / row["share_int64_static"]
        res[k] = row
        print("%-34s %s" % (k[-34:], "  ".join("%s=%.4g" % (c, v) for c, v in sorted(row.items()))))
    # whole-step totals: every kernel's per-launch average x its launches, divided by the number of bench steps the trace holds
    # (launches of the once-per-call kernel x steps per call); set-up launches (instance making, fixed-base tables) are excluded by
    # name where they have names of their own and otherwise dilute into the step count they add to
    unit, per = UNIT[cfg]
    per = float(k_per_call) if per is None else per
    steps = max([v.get("launches", 0) for k, v in res.items() if k.startswith(unit)] or [0]) * per
    if steps:
        setup = ("k_hot_", "k_terms_split<false", "k_terms_r4", "k_reduce_encode", "k_use_count", "k_class_", "k_hot_match", "k_comb_slots", "k_comb_tables<")
        if cfg == "3":
            setup = ("k_hot_",) + tuple(p for p in ("k_terms_", "k_reduce_encode", "k_use_count", "k_class_", "k_comb_", "k_encode_", "k_stmt_", "k_blind_", "k_responses", "k_decode_affine"))
        tot = sum(v["SQ_INSTS_VALU"] * v["launches"] for k, v in res.items() if "SQ_INSTS_VALU" in v and not any(k.startswith(p) or k.startswith("zkp::" + p) for p in setup)) / steps
        step_kernels = [(k, v) for k, v in res.items() if "SQ_INSTS_VALU" in v and not any(k.startswith(p) or k.startswith("zkp::" + p) for p in setup)]
        cyc = sum(v["valu_issue_cycles"] * v["launches"] for k, v in step_kernels if "valu_issue_cycles" in v) / steps
        covered = sum(v["SQ_INSTS_VALU"] * v["launches"] for k, v in step_kernels if "valu_issue_cycles" in v) / steps
        ok = covered > 0.999 * tot
        # the clock these kernels really ran at: issue-cycle-weighted over the kernels long enough for GRBM_GUI_ACTIVE / duration to be a clock
        long_k = [(v["valu_issue_cycles"] * v["launches"], v["sclk_ghz"]) for k, v in step_kernels if v.get("us_in_pmc_pass", 0) >= 200.0 and "valu_issue_cycles" in v]
        sclk = sum(w * f for w, f in long_k) / sum(w for w, _ in long_k) if long_k else None
        mads = sum(v.get("mad_u64_wave_instructions", 0.0) * v["launches"] for k, v in step_kernels) / steps
        mads_by_kernel = {k: v["mad_u64_wave_instructions"] * v["launches"] / steps for k, v in step_kernels if v.get("mad_u64_wave_instructions")}
        res["_step_totals"] = {"steps": steps, "valu_wave_instructions_per_step": tot,
                               "mad_u64_wave_instructions_per_step": mads or None, "mad_u64_wave_instructions_per_step_by_kernel": mads_by_kernel,
                               "valu_issue_cycles_per_step": cyc if ok else None,
                               "valu_floor_ms_per_step_nameplate_clock": cyc / SIMDS / (NAMEPLATE_GHZ * 1e6) if ok else None,
                               "sclk_ghz_observed": sclk,
                               "valu_floor_ms_per_step_observed_clock": cyc / SIMDS / (sclk * 1e6) if ok and sclk else None,
                               "note": "sums over the step's kernels x launches / steps in the trace (set-up kernels excluded by name).  valu_issue_cycles_per_step = "
                                       "4 x (SQ_INSTS_VALU - SQ_ACTIVE_INST_VALU2): VALU issue slots; / 1024 SIMDs / clock = the time the step's VALU work needs "
                                       "with every SIMD issuing every cycle: at the 2.4 GHz nameplate clock (a hard floor of ms_per_step) and at the clock the kernels "
                                       "were observed to run at under this load (GRBM_GUI_ACTIVE / duration, kernels of >= 200 us, weighted by issue cycles)"}
        print("%-34s valu wave-instructions per bench step = %.4g, issue cycles = %s; VALU floor %s ms per step at %.1f GHz, %s ms at the observed %s GHz  (%g steps in the trace)"
              % ("_step_totals", tot, "%.4g" % cyc if ok else "n/a", "%.4f" % res["_step_totals"]["valu_floor_ms_per_step_nameplate_clock"] if ok else "n/a", NAMEPLATE_GHZ,
                 "%.4f" % res["_step_totals"]["valu_floor_ms_per_step_observed_clock"] if ok and sclk else "n/a", "%.3f" % sclk if sclk else "n/a", steps))
    # efficiency table on one stream (what tools/kernel_efficiency.sh printed in round 3, now with the cycle-weighted busy fraction)
    rows = [(k, v) for k, v in res.items() if isinstance(v, dict) and "avg_us_one_stream" in v]
    if rows:
        tot_t = sum(v["avg_us_one_stream"] * v["launches"] for _, v in rows)
        print()
        print("# per-kernel efficiency: duration and time share on ONE stream (a kernel's own duration); busy% = valu_busy (issue cycles / GRBM_GUI_ACTIVE cycles of the PMC pass);")
        print("# sclk = clock observed in that pass (kernels >= 100 us); MeanOccupancyPerCU: 32 = full")
        print("# 2cyc% s/co = share of 2-cycle-class opcodes in the disassembly / share of instructions that were CO-ISSUED with another one (SQ_ACTIVE_INST_VALU2 / SQ_INSTS_VALU: at most 50)")
        print("%-46s %6s %10s %7s %12s %8s %7s %8s %9s %11s" % ("kernel", "calls", "avg_us", "time%", "valu_instr", "busy%", "sclk", "occ/CU", "waves", "2cyc% s/co"))
        for k, v in sorted(rows, key=lambda kv: -kv[1]["avg_us_one_stream"] * kv[1]["launches"]):
            print("%-46s %6d %10.1f %7.2f %12.4g %8s %7s %8.2f %9.0f %11s" % (k[:46], v["launches"], v["avg_us_one_stream"], 100.0 * v["avg_us_one_stream"] * v["launches"] / tot_t,
                                                                           v.get("SQ_INSTS_VALU", float("nan")),
                                                                           "%.1f" % (100 * v["valu_busy"]) if "valu_busy" in v else "n/a",
                                                                           "%.2f" % v["sclk_ghz"] if v.get("us_in_pmc_pass", 0) >= 100.0 else "-",
                                                                           v.get("MeanOccupancyPerCU", float("nan")), v.get("SQ_WAVES", float("nan")),
                                                                           "%.0f / %.1f" % (100 * v["share_2cycle_static"], 100 * v["share_co_issued"]) if "share_2cycle_static" in v and "share_co_issued" in v else "-"))
    # key the counters to the kernel sources and the workload shape they were collected from (bench.py reports them only on a match)
    res["_source_sha256"] = sha
    res["_workload"] = {"config": cfg, "batch": bl["config"]["batch_per_gpu"], "batches_per_call": k_per_call, "steps": bl["steps"], "streams": bl["config"]["streams"]}
    json.dump(res, open(out, "w"), indent=1)


if __name__ == "__main__":
    main()
