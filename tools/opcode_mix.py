#!/usr/bin/env python3
"""Static VALU opcode mix of every kernel in the shipped code object -> profiles/r06_opcode_mix.json (keyed to the kernel sources).  INFORMATION
about the code (share of 2-cycle-class opcodes, of 64-bit integer opcodes, the top opcodes), not the weights of the VALU ceiling any more.

History: rounds 1 - 3 divided raw SQ_INSTS_VALU counts by ONE peak, 34.5e12 lane-instructions/s -- 4 issue cycles per instruction at
one assumed clock -- and kernels made of 2-cycle-class opcodes (k_responses: 63 % v_mov / v_and / v_add_u32 ...) came out above 1.0.  This round first
charged every opcode its class weight from the single-opcode microbenchmark (2 or 4 cycles: issue_cycles_per_wave_instruction below, still written) --
which UNDER-charges: a SIMD issues one VALU instruction per 4-cycle slot and a second one in the same slot only if both are of the 2-cycle class, so a
2-cycle instruction between v_mad_u64_u32 costs a whole slot (tools/microbench/valu_mix.hip -> profiles/r04_valu_mix_microbench.txt: the alternating
pattern runs at 4.03 cycles per instruction, not 3.0).  The hardware counts the co-issued second instructions (SQ_ACTIVE_INST_VALU2: 0.48 per instruction
on a pure 2-cycle stream, 0.00 on a pure 4-cycle one), so the ceiling is now counted, not modelled (tools/pmc_summary.py, bench.py):

    valu_busy(kernel) = 4 * (SQ_INSTS_VALU - SQ_ACTIVE_INST_VALU2) / (1024 SIMDs * GRBM_GUI_ACTIVE / 8 XCDs)        all three counters from ONE rocprofv3 --pmc pass

i.e. issue slots used / clock cycles that went by, in the chip's own clock (which is NOT the 2.4 GHz nameplate under this load: the term kernel runs at
1.97 - 2.10 GHz).  The class of an opcode here is read off profiles/r04_valu_rates_microbench.txt (a pure stream reaches > 50e12 lane-instr/s => 2-cycle
class); opcodes without a measurement (< 1 % of any kernel) count as 4-cycle class.

    python tools/opcode_mix.py [OUT.json]        (needs /opt/rocm/lib/llvm/bin/{llvm-objcopy,clang-offload-bundler,llvm-objdump} and c++filt)
"""
import collections
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
LLVM = "/opt/rocm/lib/llvm/bin"
RATES_FILE = os.path.join(ROOT, "profiles", "r04_valu_rates_microbench.txt")
DEFAULT_RATE = 36.0e12            # unmeasured opcodes: the 4-cycle class


def measured_rates():
    """opcode -> lane-instructions / s, chip-wide, from the microbenchmark table (the VCC-hazard artefact rows are skipped)"""
    rates = {}
    for line in open(RATES_FILE):
        m = re.match(r"^(v_\w+)(\([^)]*\))?\s+[\d.]+\s+([\d.]+)\s+[\d.]+\s*$", line)
        if not m:
            continue
        op, variant, g = m.group(1), m.group(2) or "", float(m.group(3)) * 1e9
        if op == "v_cndmask_b32" and "sgpr" not in variant:
            continue
        rates[op] = max(rates.get(op, 0.0), g)
    return rates


def issue_cycles(rate):
    """SIMD issue cycles of one wave64 instruction: the two classes of the microbenchmark, 2 cycles (a pure stream reaches > 50e12 lane-instr/s)
    and 4 cycles (everything else incl. unmeasured opcodes); profiles/r04_valu_issue_cycles_pmc.txt has them in hardware clock cycles"""
    return 2.0 if rate > 50e12 else 4.0


def normalise(mnemonic):
    """v_and_b32_e32 -> v_and_b32 ; any DPP form -> priced as v_mov_b32_dpp (the DPP path is 4-cycle class) ; SDWA likewise"""
    if mnemonic.endswith("_dpp"):
        return "v_mov_b32_dpp"
    m = re.sub(r"_(e32|e64|sdwa)$", "", mnemonic)
    alias = {"v_subrev_u32": "v_sub_u32", "v_subrev_co_u32": "v_add_co_u32", "v_sub_co_u32": "v_add_co_u32", "v_subb_co_u32": "v_addc_co_u32",
             "v_subbrev_co_u32": "v_addc_co_u32", "v_not_b32": "v_xor_b32", "v_lshrrev_b16": "v_lshrrev_b32", "v_min_u32": "v_max_u32",
             "v_cmp_ne_u32": "v_cmp_eq_u32", "v_cmp_gt_u32": "v_cmp_eq_u32", "v_cmp_lt_u32": "v_cmp_eq_u32", "v_cmp_le_u32": "v_cmp_eq_u32",
             "v_cmp_ge_u32": "v_cmp_eq_u32", "v_cmp_ne_u64": "v_cmp_eq_u32", "v_cmp_eq_u64": "v_cmp_eq_u32", "v_cmp_gt_u64": "v_cmp_eq_u32",
             "v_cmp_lt_u64": "v_cmp_eq_u32", "v_cmp_gt_i32": "v_cmp_eq_u32", "v_cmp_lt_i32": "v_cmp_eq_u32", "v_cmp_ne_u16": "v_cmp_eq_u32",
             "v_readfirstlane_b32": "v_readlane_b32", "v_and_b16": "v_and_b32", "v_or_b16": "v_or_b32", "v_xad_u32": "v_and_or_b32", "v_mul_lo_u16": "v_mul_u32_u24"}
    return alias.get(m, m)


def disassemble(lib_path):
    with tempfile.TemporaryDirectory() as d:
        fat, co = os.path.join(d, "a.fat"), os.path.join(d, "a.co")
        subprocess.check_call([os.path.join(LLVM, "llvm-objcopy"), "-O", "binary", "--only-section=.hip_fatbin", lib_path, fat])
        subprocess.check_call([os.path.join(LLVM, "clang-offload-bundler"), "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--input=" + fat,
                               "--output=" + co, "--unbundle"])
        return subprocess.check_output([os.path.join(LLVM, "llvm-objdump"), "-d", "--no-show-raw-insn", co], text=True)


def kernel_mixes(asm):
    cur, hist = None, collections.defaultdict(collections.Counter)
    for line in asm.splitlines():
        m = re.match(r"^[0-9a-f]+ <(.+)>:", line)
        if m:
            cur = m.group(1)
            continue
        m = re.match(r"^\s+(\w+)", line)
        if m and cur:
            hist[cur][m.group(1)] += 1
    names = list(hist)
    dem = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.splitlines()
    out = {}
    for mangled, d in zip(names, dem):
        out[d.split("(")[0].replace("void ", "")] = hist[mangled]
    return out


def main():
    from zkp_amd import engine
    import bench
    out_path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "profiles", "r06_opcode_mix.json")
    rates = measured_rates()
    res = {"_source_sha256": bench.source_sha256(), "_rates_file": os.path.relpath(RATES_FILE, ROOT), "_default_rate": DEFAULT_RATE,
           "_note": "per kernel: static VALU mix of the shipped code object priced in SIMD issue cycles (2 or 4 per opcode); "
                    "share_2cycle_class = opcodes that CAN share an issue slot with another of their class; whether they do is counted by SQ_ACTIVE_INST_VALU2 "
                    "(tools/pmc_summary.py: valu_busy = 4 x (SQ_INSTS_VALU - SQ_ACTIVE_INST_VALU2) / (SIMDs x GRBM_GUI_ACTIVE per XCD)); issue_cycles_per_wave_instruction "
                    "(2 or 4 per opcode) is the lower bound a perfectly paired schedule would reach"}
    for kernel, h in sorted(kernel_mixes(disassemble(engine.LIB_PATH)).items()):
        valu = collections.Counter()
        for op, c in h.items():
            if op.startswith("v_"):
                valu[normalise(op)] += c
        n = sum(valu.values())
        if not n:
            continue
        spw = sum(c / n * 64.0 / rates.get(op, DEFAULT_RATE) for op, c in valu.items())
        cyc = sum(c / n * issue_cycles(rates.get(op, DEFAULT_RATE)) for op, c in valu.items())
        unmeasured = sum(c for op, c in valu.items() if op not in rates) / n
        two_cycle = sum(c for op, c in valu.items() if rates.get(op, DEFAULT_RATE) > 50e12) / n
        int64 = sum(c for op, c in valu.items() if op in ("v_mad_u64_u32", "v_lshl_add_u64", "v_lshlrev_b64", "v_lshrrev_b64", "v_ashrrev_i64", "v_mov_b64")) / n
        res[kernel] = {"valu_instructions_static": n, "all_instructions_static": sum(h.values()), "seconds_per_wave_instruction": spw, "issue_cycles_per_wave_instruction": cyc,
                       "equivalent_peak_lane_instr_per_s": 64.0 / spw, "share_2cycle_class": two_cycle, "share_unmeasured": unmeasured, "share_int64_static": int64,
                       "share_mad_u64_static": valu.get("v_mad_u64_u32", 0) / n,
                       "top": [[op, round(c / n, 4)] for op, c in valu.most_common(8)]}
    json.dump(res, open(out_path, "w"), indent=1)
    print("%-62s %8s %10s %8s %8s" % ("kernel", "valu", "peak T/s", "2-cycle", "unmeas."))
    for k, v in sorted(res.items(), key=lambda kv: -(kv[1].get("valu_instructions_static", 0) if isinstance(kv[1], dict) else 0)):
        if isinstance(v, dict):
            print("%-62s %8d %10.1f %8.3f %8.3f" % (k[:62], v["valu_instructions_static"], v["equivalent_peak_lane_instr_per_s"] / 1e12, v["share_2cycle_class"], v["share_unmeasured"]))


if __name__ == "__main__":
    main()
