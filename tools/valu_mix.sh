#!/bin/bash
# Run on the GPU box: tools/microbench/valu_mix under one PMC pass -> gpurun_out/r04_valu_mix_microbench.txt
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out
mkdir -p $O
cd /tmp
rm -rf $O/valu_mix
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU2 --output-format csv -d $O/valu_mix -o mix -- $R/tools/microbench/valu_mix > $O/valu_mix.log 2>&1
cd $R
python - <<'PY' > $O/r04_valu_mix_microbench.txt
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter(); ns = collections.defaultdict(float)
for r in csv.DictReader(open(glob.glob("gpurun_out/valu_mix/**/mix_counter_collection.csv", recursive=True)[0])):
    k = r["Kernel_Name"].split("(")[0]
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Counter_Name"] == "SQ_INSTS_VALU":
        n[k] += 1; ns[k] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
print("# tools/microbench/valu_mix.hip under rocprofv3 --pmc SQ_INSTS_VALU GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU2 (MI355X): M = v_mad_u64_u32 (4-cycle class), A = v_and_b32 (2-cycle class),")
print("# (M4A8 = runs: 4 M then 8 A ...; X = v_and / v_add_u32 / v_lshrrev_b32 alternating),")
print("# independent register chains, 8 waves per SIMD, pattern x 4 per loop iteration.  'weighted' = (2 n_A + 4 n_M) / (n_A + n_M): what the opcode weights of tools/opcode_mix.py charge.")
print("%-12s %10s %12s %10s %14s %14s" % ("pattern", "weighted", "measured", "sclk GHz", "meas./weighted", "VALU2/INSTS"))
for k, v in acc.items():
    pat = k[2:]
    import re
    if re.search(r"\d", pat):                       # run-length names: M<count>A<count> / X<count> (X = a mix of 2-cycle opcodes)
        cnt = {c: int(n) for c, n in re.findall(r"([MAX])(\d+)", pat)}
        n2, n4 = cnt.get("A", 0) + cnt.get("X", 0), cnt.get("M", 0)
    else:
        n2, n4 = pat.count("A"), pat.count("M")
    w = (2.0 * n2 + 4.0 * n4) / (n2 + n4)
    L = n[k]; g = v["GRBM_GUI_ACTIVE"] / L / 8; iv = v["SQ_INSTS_VALU"] / L
    m = g * 1024 / iv
    print("%-12s %10.2f %12.3f %10.3f %14.3f %14.4f" % (pat, w, m, g / (ns[k] / L), m / w, v["SQ_ACTIVE_INST_VALU2"] / v["SQ_INSTS_VALU"]))
PY
cat $O/r04_valu_mix_microbench.txt
