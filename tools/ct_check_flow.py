"""Constant-time evidence for the WHOLE prover flow (round 6), not only its multiscalar multiplications: zkp_fused_prove_dev -- transcripts incl. the
transcript RNG that is re-keyed with every witness (prover.rs:78-82), blindings, the constant-time MSMs, responses s * c + b (prover.rs:107-109) -- is run
with the SAME public inputs (points, transcript states) and very different SECRETS: the witness scalars and the per-proof entropy.  rocprofv3 counts the
executed instructions of every kernel; if every counter of every kernel is identical for every secret pattern, no branch and no load / store was taken or
skipped because of a secret.  (tools/ct_check.py does the same for the term path alone with eight SCALAR patterns and adds wavefront cycle counts.)

    rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_WAVES \\
              --output-format csv -d OUT -o flow -- python tools/ct_check_flow.py
    rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_INSTS_SMEM SQ_INSTS_FLAT --output-format csv -d OUT -o flow2 -- python tools/ct_check_flow.py
    python tools/ct_check_flow.py --summarise OUT/flow_counter_collection.csv OUT/flow2_counter_collection.csv

Both schedules are run: the throughput schedule of the _dev entry points (program A inside the comb tables' launch) and the latency schedule
(ZKP_OPT_DEV_OVERLAP = 2: assemble + chain kernels of their own, quad tables); and the word-operation interpreter (ZKP_OPT_TRANSCRIPT_STEPS = 0)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

L = 2**252 + 27742317777372353535851937790883648493
PATTERNS = ["witnesses 0, entropy 0", "witnesses 1, entropy ff", "witnesses l-1, entropy 55", "witnesses random-a, entropy random-a", "witnesses random-b, entropy random-b",
            "witnesses random-b, entropy random-b again"]
SCHEDULES = (("throughput schedule", ()), ("latency schedule", ((5, 2),)), ("throughput schedule, interpreter", ((15, 0),)))


def secrets_of(kind, shape, rng, last):
    n = int(np.prod(shape[:-1]))
    if kind.endswith("again"):
        return last
    if "random" in kind:
        s = rng.integers(0, 256, size=(n, 32), dtype=np.uint8)
        s[:, 31] &= 0x0f
        return s.reshape(shape)
    v = {"0": 0, "1": 1, "l-1": L - 1}[kind.split(",")[0].split()[1]]
    return np.tile(np.frombuffer(int(v).to_bytes(32, "little"), np.uint8), (n, 1)).reshape(shape)


def run():
    import torch
    import bench
    from zkp_amd.engine import Engine, FusedStatement
    from zkp_amd import toolbox as T
    n = 4096
    eng = Engine(0)
    rng = np.random.default_rng(9)
    st = bench.cmz_statement()
    secrets, inst, common = bench.make_instance(eng, st, n, rng)          # a consistent instance: only its PUBLIC half is kept
    fst = FusedStatement(b"CMZ cred show n=10", *st)
    eng.prepare_fixed_points(common)
    dev = torch.device("cuda:0")
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    t0 = T.Transcript(b"Benchmark").state
    pos = int(t0[200]) | int(t0[201]) << 8 | int(t0[202]) << 16
    d_ts0, d_tbl = t(np.stack([t0] * n)), t(np.concatenate([common, inst.reshape(-1, 32)]))
    m, nc = len(st[0]), len(st[2])
    z = lambda *s: torch.zeros(s, dtype=torch.uint8, device=dev)
    d_ts, d_chal, d_resp, d_coms, d_st = z(n, 208), z(n, 32), z(n, m, 32), z(n, nc, 32), z(n * nc)
    import ctypes
    from zkp_amd.engine import load_library
    hip = load_library()
    hip.zkp_chacha20_fill_dev.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_uint64, ctypes.c_uint64, ctypes.c_void_p, ctypes.c_size_t]
    d_mark = z(64)
    for name, opts in SCHEDULES:
        for k, v in ((5, 0), (15, 1)) + tuple(opts):
            eng.set_option(k, v)
        last_s = last_e = None
        for kind in PATTERNS:                                             # one zkp_fused_prove_dev per pattern, in this order
            sec = secrets_of(kind, (n, m, 32), rng, last_s)
            if kind.endswith("again"):
                ent = last_e
            elif "random" in kind:
                ent = rng.integers(0, 256, size=(n, 32), dtype=np.uint8)
            else:
                ent = np.full((n, 32), {"0": 0, "ff": 0xff, "55": 0x55}[kind.split("entropy ")[1]], np.uint8)
            last_s, last_e = sec, ent
            d_sec, d_ent = t(sec), t(ent)
            d_ts.copy_(d_ts0)
            torch.cuda.synchronize()
            assert hip.zkp_chacha20_fill_dev(eng._h, bytes(32), 0, 0, d_mark.data_ptr(), 64) == 0      # the marker in front of every pattern's call (k_chacha20_fill)
            eng.fused_prove_dev(fst, n, pos, d_ts.data_ptr(), d_sec.data_ptr(), d_tbl.data_ptr(), d_ent.data_ptr(), d_chal.data_ptr(), d_resp.data_ptr(), d_coms.data_ptr(),
                                d_st.data_ptr())
            eng.synchronize()
            assert not d_st.cpu().numpy().any()
    eng.close()


def summarise(paths):
    import csv, collections
    P, S = len(PATTERNS), len(SCHEDULES)
    # segment (schedule s, pattern p) = the launches behind the (s P + p + 1)-th marker kernel of a pass (k_chacha20_fill); per pass, since every rocprofv3 run numbers
    # its dispatches anew
    seg = collections.defaultdict(lambda: collections.defaultdict(lambda: collections.defaultdict(list)))     # (s, p) -> kernel -> counter -> values in launch order
    for path in paths:
        rows = sorted(csv.DictReader(open(path)), key=lambda r: int(r["Dispatch_Id"]))
        k, last = -1, None
        for r in rows:
            name = r["Kernel_Name"].split("(")[0].replace("void ", "")
            if "k_chacha20_fill" in name:
                if r["Dispatch_Id"] != last:
                    k += 1
                    last = r["Dispatch_Id"]
                continue
            if k < 0 or k >= S * P or name.startswith("at::") or "rocclr" in name:
                continue
            seg[(k // P, k % P)][name][r["Counter_Name"]].append(float(r["Counter_Value"]))
    print("# zkp_fused_prove_dev, 4096 CMZ proofs, the SAME public inputs: executed-instruction counters of every kernel launch for %d secret patterns" % P)
    print("# patterns: " + " | ".join(PATTERNS))
    print("# per kernel and counter: one column per pattern; a kernel that is launched several times per call (assemble, chain, ...) shows the sum and the number of launches")
    ok = True
    for si, (sname, _) in enumerate(SCHEDULES):
        print("## %s" % sname)
        kernels = sorted({k for pi in range(P) for k in seg[(si, pi)]})
        for kname in kernels:
            counters = sorted({c for pi in range(P) for c in seg[(si, pi)][kname]})
            for c in counters:
                cols = [seg[(si, pi)][kname][c] for pi in range(P)]
                same = all(col == cols[0] for col in cols)               # launch by launch, not only the sums
                ok &= same
                print("%-42s %-20s %s  x%d  %s" % (kname[:42], c, "IDENTICAL" if same else "DIFFERENT", len(cols[0]), " ".join("%.0f" % sum(col) for col in cols)))
    print("# verdict:", "every counter of every kernel launch of the prover flow is identical across the secret patterns" if ok else "counters differ")


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "--summarise":
        summarise(sys.argv[2:])
    else:
        run()
