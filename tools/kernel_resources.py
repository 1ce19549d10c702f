#!/usr/bin/env python3
"""Per-kernel register / LDS / spill table of a hipcc build:  hipcc ... -Rpass-analysis=kernel-resource-usage 2> build.log; tools/kernel_resources.py build.log [filter]"""
import re, subprocess, sys

def main():
    txt = open(sys.argv[1]).read()
    flt = sys.argv[2] if len(sys.argv) > 2 else ""
    rows = []
    for b in re.split(r"remark: [^\n]*Function Name: ", txt)[1:]:
        name = b.split("\n")[0].strip()
        def g(k):
            m = re.search(k + r": (\d+)", b)
            return int(m.group(1)) if m else -1
        rows.append((name, g("VGPRs"), g("AGPRs"), g(r"VGPR Spill"), g(r"SGPR Spill"), g(r"ScratchSize \[bytes/lane\]"), g(r"Occupancy \[waves/SIMD\]"), g(r"LDS Size \[bytes/block\]")))
    names = subprocess.run(["c++filt"], input="\n".join(r[0] for r in rows), capture_output=True, text=True).stdout.split("\n")
    print("%-90s %5s %5s %6s %6s %7s %4s %7s" % ("kernel", "VGPR", "AGPR", "vspill", "sspill", "scratch", "occ", "LDS"))
    for r, n in zip(rows, names):
        n = re.sub(r"\(.*", "", n).replace("zkp::", "")
        if flt in n:
            print("%-90s %5d %5d %6d %6d %7d %4d %7d" % ((n[:90],) + r[1:]))

if __name__ == "__main__":
    main()
