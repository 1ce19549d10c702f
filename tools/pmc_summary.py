#!/usr/bin/env python3
"""Per-kernel averages of rocprofv3 --pmc counter_collection CSVs (one or more passes) -> text + json.

    python tools/pmc_summary.py OUT.json pass1_counter_collection.csv [pass2_counter_collection.csv ...]

HBM traffic per launch follows MI355X_MICROARCH.md section HBM: FETCH_SIZE / WRITE_SIZE are in KiB of fabric
requests; on gfx950 FETCH_SIZE reports half of the bytes of wide (16 B/lane) streaming reads, so reads are
counted as 2 x FETCH_SIZE (an upper bound for kernels whose reads are narrower gathers)."""
import collections
import csv
import json
import sys


def main():
    out, files = sys.argv[1], sys.argv[2:]
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for fn in files:
        for r in csv.DictReader(open(fn)):
            agg[r["Kernel_Name"].split("(")[0].replace("void ", "")][r["Counter_Name"]].append(float(r["Counter_Value"]))
    res = {}
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
    # whole-step totals: one k_terms_split<true> launch per bench step (set-up launches of other kernels excluded by
    # scaling every kernel's per-launch average with launches / steps, capped at what a step can contain)
    steps = max([v.get("launches", 0) for k, v in res.items() if k.startswith("k_terms_split<true")] or [0])
    if steps:
        tot = sum(v["SQ_INSTS_VALU"] * v["launches"] for k, v in res.items() if "SQ_INSTS_VALU" in v and "k_hot_" not in k) / steps
        res["_step_totals"] = {"steps": steps, "valu_wave_instructions_per_step": tot,
                               "note": "sum over kernels of SQ_INSTS_VALU x launches / steps (fixed-base table set-up kernels excluded)"}
        print("%-34s valu wave-instructions per bench step = %.4g" % ("_step_totals", tot))
    # key the counters to the kernel sources they were collected from (bench.py only reports them when the hash matches)
    import os
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    res["_source_sha256"] = bench.source_sha256()
    json.dump(res, open(out, "w"), indent=1)


if __name__ == "__main__":
    main()
