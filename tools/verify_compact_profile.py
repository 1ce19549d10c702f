"""Where the time of verify_compact goes: zkp_fused_verify_compact_dev alone -- verifier.rs:78-121 per proof: recompute every commitment
(variable-time MSMs: responses x points - challenge x left-hand side), feed them to the transcript, compare challenges -- on one stream, K x 4096 CMZ proofs per call.

    rocprofv3 --kernel-trace --stats --output-format csv -d OUT -o vc -- python tools/verify_compact_profile.py
    rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_VALU_INT64 SQ_ACTIVE_INST_VALU2 SQ_WAVES GRBM_GUI_ACTIVE --output-format csv -d OUT -o vcp -- python tools/verify_compact_profile.py
    python tools/verify_compact_profile.py --summarise OUT/vc_kernel_trace.csv OUT/vcp_counter_collection.csv

The summary counts, per kernel of ONE verify_compact call: its duration, its executed VALU instructions and 64-bit multiply-adds (v_mad_u64_u32: 98 per field
multiplication), and the time the VALU instructions alone need at 4 cycles per wave64 instruction and SIMD -- the roofline of this integer path."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

K, N_EACH, CALLS = 50, 4096, 3


def run():
    import ctypes
    import torch
    import bench
    from zkp_amd.engine import Engine, FusedStatement, load_library
    from zkp_amd import toolbox as T
    n = K * N_EACH
    eng = Engine(0)
    rng = np.random.default_rng(5)
    st = bench.cmz_statement()
    secrets, inst, common = bench.make_instance(eng, st, n, rng)
    fst = FusedStatement(b"CMZ cred show n=10", *st)
    eng.prepare_fixed_points(common)
    dev = torch.device("cuda:0")
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    t0 = T.Transcript(b"Benchmark").state
    pos = int(t0[200]) | int(t0[201]) << 8 | int(t0[202]) << 16
    d_ts0, d_tbl = t(np.stack([t0] * n)), t(np.concatenate([common, inst.reshape(-1, 32)]))
    m, nc = len(st[0]), len(st[2])
    z = lambda *s: torch.zeros(s, dtype=torch.uint8, device=dev)
    d_ts, d_chal, d_resp, d_coms, d_st, d_res = z(n, 208), z(n, 32), z(n, m, 32), z(n, nc, 32), z(n * nc), z(n)
    d_sec, d_ent = t(secrets), t(rng.integers(0, 256, size=(n, 32), dtype=np.uint8))
    d_ts.copy_(d_ts0)
    eng.fused_prove_dev(fst, n, pos, d_ts.data_ptr(), d_sec.data_ptr(), d_tbl.data_ptr(), d_ent.data_ptr(), d_chal.data_ptr(), d_resp.data_ptr(), d_coms.data_ptr(), d_st.data_ptr())
    eng.synchronize()
    hip = load_library()
    hip.zkp_chacha20_fill_dev.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_uint64, ctypes.c_uint64, ctypes.c_void_p, ctypes.c_size_t]
    d_mark = z(64)
    for _ in range(CALLS + 1):                  # (the first call also registers what a first call registers; the summary reads the calls behind the LAST markers)
        d_ts.copy_(d_ts0)
        torch.cuda.synchronize()
        assert hip.zkp_chacha20_fill_dev(eng._h, bytes(32), 0, 0, d_mark.data_ptr(), 64) == 0       # marker: k_chacha20_fill in front of every verify_compact call
        eng.fused_verify_compact_dev(fst, n, pos, d_ts.data_ptr(), d_tbl.data_ptr(), d_chal.data_ptr(), d_resp.data_ptr(), d_res.data_ptr())
        eng.synchronize()
        assert not d_res.cpu().numpy().any(), "a proof of the prover did not verify"
    eng.close()


def _segments(rows, key):
    """launches behind each marker kernel, in dispatch order"""
    rows = sorted(rows, key=key)
    segs, cur = [], None
    for r in rows:
        name = r["Kernel_Name"].split("(")[0].replace("void ", "")
        if "k_chacha20_fill" in name:
            cur = []
            segs.append(cur)
        elif cur is not None and not name.startswith("at::") and "rocclr" not in name:
            cur.append((name, r))
    return segs


def summarise(trace_csv, pmc_csv):
    import csv, collections
    n = K * N_EACH
    tr = list(csv.DictReader(open(trace_csv)))
    segs = _segments(tr, lambda r: int(r["Start_Timestamp"]))[-CALLS:]
    dur = collections.defaultdict(list)
    for seg in segs:
        per = collections.defaultdict(float)
        for name, r in seg:
            per[name] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        for k, v in per.items():
            dur[k].append(v)
    cnt = collections.defaultdict(lambda: collections.defaultdict(list))
    pm = list(csv.DictReader(open(pmc_csv)))
    by_disp = collections.defaultdict(dict)
    for r in pm:
        d = by_disp[int(r["Dispatch_Id"])]
        d["Kernel_Name"] = r["Kernel_Name"]
        d[r["Counter_Name"]] = d.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
        if r.get("End_Timestamp"):
            d["pass_ns"] = float(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    rows = [dict(v, Dispatch_Id=k) for k, v in by_disp.items()]
    for seg in _segments(rows, lambda r: r["Dispatch_Id"])[-CALLS:]:
        per = collections.defaultdict(lambda: collections.defaultdict(float))
        for name, r in seg:
            for c in ("SQ_INSTS_VALU", "SQ_INSTS_VALU_INT64", "SQ_ACTIVE_INST_VALU2", "SQ_WAVES", "GRBM_GUI_ACTIVE", "pass_ns"):
                per[name][c] += r.get(c, 0.0)
        for k, v in per.items():
            for c, x in v.items():
                cnt[k][c].append(x)
    mean = lambda xs: sum(xs) / len(xs) if xs else 0.0
    total_us = sum(mean(v) for v in dur.values())
    print("# zkp_fused_verify_compact_dev, ONE call of %d x %d = %d CMZ proofs on one stream (mean of the last %d calls); kernels as rocprofv3 names them" % (K, N_EACH, n, CALLS))
    print("# us = duration per call (kernel trace); VALU, mad64 = wave64 instructions per call (PMC pass); field mults / proof = mad64 x 64 / 98 / proofs;")
    print("# issue us = (VALU - co-issued 2-cycle instructions) x 4 cycles / 1024 SIMDs at the clock of the PMC pass (GRBM_GUI_ACTIVE / 8 XCDs / the dispatch's duration there);")
    print("# issue/us = how much of the kernel's time its VALU issue slots alone explain (tools/pmc_summary.py: valu_busy)")
    print("%-44s %10s %7s %12s %12s %14s %9s %10s %9s" % ("kernel", "us", "share", "VALU", "mad64", "mults/proof", "GHz", "issue us", "issue/us"))
    tot_mults = tot_issue = 0.0
    for k in sorted(dur, key=lambda k: -mean(dur[k])):
        us = mean(dur[k])
        valu, mad, busy = mean(cnt[k]["SQ_INSTS_VALU"]), mean(cnt[k]["SQ_INSTS_VALU_INT64"]), mean(cnt[k]["GRBM_GUI_ACTIVE"])
        pass_ns = mean(cnt[k]["pass_ns"])
        ghz = busy / 8 / pass_ns if pass_ns else 0.0
        issue = (valu - mean(cnt[k]["SQ_ACTIVE_INST_VALU2"])) * 4 / 1024 / (ghz * 1e3) if ghz else 0.0
        mults = mad * 64 / 98 / n
        tot_mults += mults
        tot_issue += issue
        print("%-44s %10.1f %6.1f%% %12.4g %12.4g %14.0f %9.2f %10.1f %9.2f" % (k[:44], us, 100 * us / total_us, valu, mad, mults, ghz, issue, issue / us if us else 0))
    print("# total %.1f us per call = %.0f proofs/s on one stream (kernels only); %.0f field multiplications per proof executed; VALU issue time %.1f us = %.2f of the call"
          % (total_us, n / (total_us * 1e-6), tot_mults, tot_issue, tot_issue / total_us))


if __name__ == "__main__":
    if len(sys.argv) > 3 and sys.argv[1] == "--summarise":
        summarise(sys.argv[2], sys.argv[3])
    else:
        run()
