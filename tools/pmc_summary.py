#!/usr/bin/env python3
"""Per-kernel averages of rocprofv3 --pmc counter_collection CSVs (one or more passes) -> text + json.

    python tools/pmc_summary.py OUT.json BENCH_LINE.json pass1_counter_collection.csv [pass2_counter_collection.csv ...]

BENCH_LINE.json = the JSON line the profiled `python bench.py ...` command printed (workload shape: config, batch, batches per
call, steps).  HBM traffic per launch follows MI355X_MICROARCH.md section HBM: FETCH_SIZE / WRITE_SIZE are in KiB of fabric
requests; on gfx950 FETCH_SIZE reports half of the bytes of wide (16 B/lane) streaming reads, so reads are counted as
2 x FETCH_SIZE (an upper bound for kernels whose reads are narrower gathers)."""
import collections
import csv
import json
import os
import sys

# the kernel that runs once per CALL of a workload's step loop, and how many bench steps one such launch stands for
UNIT = {"2": ("k_terms_split<true", None), "4share": ("k_terms_split<true", 1.0), "5share": ("k_terms_split<true", 1.0), "3": ("k_pip_combine", 0.5)}


def main():
    out, line, files = sys.argv[1], sys.argv[2], sys.argv[3:]
    bl = json.loads([l for l in open(line).read().splitlines() if l.startswith("{")][-1])
    cfg = bl["config"]["baseline_config"]
    k_per_call = bl["config"].get("batches_per_call", 1)
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for fn in files:
        for r in csv.DictReader(open(fn)):
            agg[r["Kernel_Name"].split("(")[0].replace("void ", "")][r["Counter_Name"]].append(float(r["Counter_Value"]))
    res = {}
    print("# workload: --config %s, %d proofs per batch, %d batch(es) per call, %d timed steps" % (cfg, bl["config"]["batch_per_gpu"], k_per_call, bl["steps"]))
    print("%-34s %s" % ("kernel", "counter averages per launch (n launches)"))
    for k in sorted(agg):
        if "k_" not in k:
            continue
        row = {c: sum(v) / len(v) for c, v in agg[k].items()}
        row["launches"] = max(len(v) for v in agg[k].values())
        if "FETCH_SIZE" in row and "WRITE_SIZE" in row:
            row["hbm_bytes_per_launch"] = (2.0 * row["FETCH_SIZE"] + row["WRITE_SIZE"]) * 1024.0
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
        res["_step_totals"] = {"steps": steps, "valu_wave_instructions_per_step": tot,
                               "note": "sum over the step's kernels of SQ_INSTS_VALU x launches / steps in the trace (set-up kernels excluded by name)"}
        print("%-34s valu wave-instructions per bench step = %.4g  (%g steps in the trace)" % ("_step_totals", tot, steps))
    # key the counters to the kernel sources and the workload shape they were collected from (bench.py reports them only on a match)
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    res["_source_sha256"] = bench.source_sha256()
    res["_workload"] = {"config": cfg, "batch": bl["config"]["batch_per_gpu"], "batches_per_call": k_per_call, "steps": bl["steps"], "streams": bl["config"]["streams"]}
    json.dump(res, open(out, "w"), indent=1)


if __name__ == "__main__":
    main()
