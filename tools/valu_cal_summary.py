#!/usr/bin/env python3
"""SIMD issue cycles of every opcode of tools/microbench/valu_rates.hip in the chip's OWN clock cycles (GRBM_GUI_ACTIVE), from one
rocprofv3 --pmc pass over the microbenchmark -> profiles/r04_valu_issue_cycles_pmc.txt   (tools/valu_cal.sh runs it on the GPU box)

    python tools/valu_cal_summary.py cal_counter_collection.csv

cycles per wave instruction per SIMD = (GRBM_GUI_ACTIVE / 8 XCDs) x 1024 SIMDs / SQ_INSTS_VALU: what the 2 / 4 weights of tools/opcode_mix.py are
read from.  The clock column (GRBM_GUI_ACTIVE / 8 / dispatch duration) shows why rates in lane-instructions per SECOND are the wrong yardstick:
the chip does not hold its 2.4 GHz nameplate clock under a full-width VALU stream."""
import collections
import csv
import sys

SIMDS, XCDS = 1024, 8


def main():
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    n, ns = collections.Counter(), collections.defaultdict(float)
    for r in csv.DictReader(open(sys.argv[1])):
        k = r["Kernel_Name"].split("(")[0]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Counter_Name"] == "SQ_INSTS_VALU":
            n[k] += 1
            ns[k] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    print("# rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU2 SQ_ACTIVE_INST_VALU SQ_WAVES -- tools/microbench/valu_rates   (MI355X)")
    print("# 8 waves per SIMD, 8 independent chains x 4 per loop iteration (32 VALU instructions per s_add / s_cmp / s_cbranch), averages over both launches")
    print("# last column: SQ_ACTIVE_INST_VALU2 / SQ_INSTS_VALU -- 0.46 - 0.48 for every 2-cycle opcode, 0.000 for every 4-cycle one: the one counter that tells the two")
    print("# issue classes apart DYNAMICALLY (tools/pmc_summary.py: share_2cycle_dynamic = that ratio / 0.479, next to the static share of the disassembly)")
    print("%-18s %14s %14s %9s %16s %22s %12s" % ("kernel", "SQ_INSTS_VALU", "GUI_ACTIVE/8", "sclk GHz", "cycles/instr/SIMD", "SQ_ACTIVE_INST_VALU/INSTS", "VALU2/INSTS"))
    for k, v in acc.items():
        L = n[k]
        iv, g = v["SQ_INSTS_VALU"] / L, v["GRBM_GUI_ACTIVE"] / L / XCDS
        print("%-18s %14.0f %14.0f %9.3f %16.3f %22.3f %12.4f" % (k, iv, g, g / (ns[k] / L), g * SIMDS / iv, v["SQ_ACTIVE_INST_VALU"] / v["SQ_INSTS_VALU"],
                                                                 v.get("SQ_ACTIVE_INST_VALU2", float("nan")) / v["SQ_INSTS_VALU"]))


if __name__ == "__main__":
    main()
