#!/usr/bin/env python3
"""bench.py -- the hot path of dalek-cryptography/zkp on MI355X.

A "step" = one pass of the hot path over ONE batch of N = 4096 CMZ'13 10-attribute credential
presentations (BASELINE.json configs[1]; benches/zkp.rs:27-46), everything on the GPU:
  (i)  PROVE all N proofs (zkp_fused_prove_dev = N x prover.rs:76-112): Merlin transcripts, blinding factors
       from the transcript RNG, the 11 constant-time commitment MSMs per proof (45,056 MSMs / 126,976 terms)
       with compression, challenges, responses;
  (ii) BATCH-VERIFY the N proofs just made (zkp_fused_batch_verify_dev = batch_verifier.rs:67-235):
       transcripts with identity rejection, challenges, the coefficient build mod l, and the single
       random-linear-combination MSM (12 + 24 N = 98,316 terms, decompression on the GPU) down to the
       identity test.
Inputs are synthetic (random witnesses, a consistent CMZ instance made by the engine itself, fixed
entropy / weights) and RESIDENT IN HBM before the timed region; every step starts from fresh
`Transcript::new(b"Benchmark")` states like the reference's bench loop.  value = proofs per second that
were both proven and batch-verified.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--batch 4096]
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py --gpus N ...

Prints ONE JSON line (rank 0).
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

METRIC = "proofs/sec + batch-verifies/sec, CMZ13 10-attr credential, 1/2/4/8 MI355X"
VALU_PEAK_MADS = 34.5e12      # v_mad_u64_u32 lane-instructions / s, measured: profiles/r01_valu_rates_microbench.txt
HBM_PEAK_GBS = 8000.0         # MI355X_MICROARCH.md
LABEL = b"Benchmark"


def cmz_shape(n):
    """CSR shape of the prover's commitment MSMs for n CMZ proofs (benches/zkp.rs:27-46) over the stand-alone point table
    [0..11) common X_1..X_10, A ; then per proof j: P_j = 11 + 2j, Q_j = 12 + 2j (used by the full-size MSM tests and
    tools/large_batch_bench.py).  Scalar order per proof follows the constraints: (m_i, z_i) x 10, then m_1..m_10, minus_z_Q."""
    import numpy as np
    per = []
    for i in range(10):
        per += [("P", None), ("A", None)]
    per += [("X", i) for i in range(10)] + [("Q", None)]
    off_one = np.array([2 * i for i in range(11)] + [31], dtype=np.uint32)      # 10 x 2 terms, 1 x 11 terms
    pidx = np.zeros((n, 31), dtype=np.uint32)
    j = np.arange(n, dtype=np.uint32)
    for t, (kind, i) in enumerate(per):
        if kind == "P":
            pidx[:, t] = 11 + 2 * j
        elif kind == "Q":
            pidx[:, t] = 12 + 2 * j
        elif kind == "A":
            pidx[:, t] = 10
        else:
            pidx[:, t] = i
    off = (off_one[None, :-1] + 31 * j[:, None]).reshape(-1)
    off = np.concatenate([off, np.array([31 * n], dtype=np.uint32)]).astype(np.uint32)
    return off, pidx.reshape(-1), 11 + 2 * n


def cmz_statement():
    """cred_show_10 (benches/zkp.rs:27-46) in the argument form of zkp_amd.engine.FusedStatement."""
    secrets = [b"m_%d" % i for i in range(1, 11)] + [b"z_%d" % i for i in range(1, 11)] + [b"minus_z_Q"]
    inst = [b"C_%d" % i for i in range(1, 11)] + [b"P", b"Q", b"V"]
    common = [b"X_%d" % i for i in range(1, 11)] + [b"A", b"B"]
    points = [(x, False) for x in inst] + [(x, True) for x in common]
    pi = {name: i for i, (name, _) in enumerate(points)}
    si = {name: i for i, name in enumerate(secrets)}
    cons = [(pi[b"C_%d" % i], [(si[b"m_%d" % i], pi[b"P"]), (si[b"z_%d" % i], pi[b"A"])]) for i in range(1, 11)]
    cons.append((pi[b"V"], [(si[b"m_%d" % i], pi[b"X_%d" % i]) for i in range(1, 11)] + [(si[b"minus_z_Q"], pi[b"Q"])]))
    return secrets, points, cons


def cmz_instance(eng, n, rng):
    """n consistent CMZ presentations: witnesses, instance points [13][n][32] (C_1..C_10, P, Q, V), common [12][32].
    Made with the engine's own MSMs (untimed set-up)."""
    import numpy as np
    from zkp_amd.engine import ZKP_CT

    def rs(k):
        s = rng.integers(0, 256, size=(k, 32), dtype=np.uint8)
        s[:, 31] &= 0x0f                       # < 2^252 < l
        return s

    base = np.frombuffer(bytes.fromhex("e2f2ae0a6abc4e71a884a961c500515f58e30b6aa582dd8db6a65945e08d2d76"), np.uint8).reshape(1, 32)
    k = 12 + 2 * n
    pts, st = eng.msm_many(np.arange(k + 1, dtype=np.uint32), rs(k), np.zeros(k, np.uint32), base, ZKP_CT)
    assert not st.any()
    common, P, Q = pts[:12], pts[12:12 + n], pts[12 + n:]
    secrets = rs(n * 21).reshape(n, 21, 32)
    table = np.concatenate([common, P, Q])
    j = np.arange(n, dtype=np.uint32)
    pidx = np.zeros((n, 31), np.uint32)
    sidx = np.zeros(31, np.int64)
    for i in range(10):                        # C_i = m_i P + z_i A
        pidx[:, 2 * i], pidx[:, 2 * i + 1] = 12 + j, 10
        sidx[2 * i], sidx[2 * i + 1] = i, 10 + i
    for i in range(10):                        # V = sum m_i X_i + minus_z_Q Q
        pidx[:, 20 + i] = i
        sidx[20 + i] = i
    pidx[:, 30], sidx[30] = 12 + n + j, 20
    off_one = np.array([2 * i for i in range(11)], np.uint32)
    off = np.concatenate([(off_one[None, :] + 31 * j[:, None]).reshape(-1), np.array([31 * n], np.uint32)]).astype(np.uint32)
    cv, st = eng.msm_many(off, np.ascontiguousarray(secrets[:, sidx]).reshape(-1, 32), pidx.reshape(-1), table, ZKP_CT)
    assert not st.any()
    cv = cv.reshape(n, 11, 32)
    inst = np.ascontiguousarray(np.concatenate([cv[:, :10].transpose(1, 0, 2), P[None], Q[None], cv[:, 10][None]]))
    return secrets, inst, np.ascontiguousarray(common)


def pick_streams(steps):
    """Batches in flight.  A batch is a chain of ~70 kernels, several of them only a few dozen wavefronts wide, so the chip
    is filled by running independent batches side by side.  With K timed steps over S streams the last round of batches
    runs with K mod S streams busy; pick S in 12..24 that leaves the fewest idle slots (ties: more streams)."""
    if steps <= 24:
        return max(1, steps)
    best = min(range(12, 25), key=lambda s: ((-steps) % s, -s))
    return best


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=4096, help="proofs per GPU per step (BASELINE configs[1]: 4096)")
    ap.add_argument("--streams", type=int, default=0, help="independent batches in flight, each on its own HIP stream / engine context "
                                                            "(0 = automatic: 12..24, the count that splits --steps most evenly)")
    ap.add_argument("--no-graphs", action="store_true", help="enqueue every kernel of every batch from the host instead of replaying one "
                                                            "HIP graph per stream (the chain is ~75 kernels per batch: launch-bound)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--engine-opt", action="append", default=[], metavar="ID=VALUE",
                    help="zkp_ctx_set_option(ID, VALUE) on every engine context (tuning experiments; results never depend on it)")
    args = ap.parse_args()

    if args.gpus > 1 and "RANK" not in os.environ:
        # convenience: self-launch one process per GPU over RCCL exactly as the driver would
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
               "--master-addr", "127.0.0.1", "--master-port", os.environ.get("MASTER_PORT", "29533"), os.path.abspath(__file__)] + sys.argv[1:]
        sys.exit(subprocess.call(cmd))

    if args.streams <= 0:
        args.streams = pick_streams(args.steps)
    os.environ.setdefault("GPU_MAX_HW_QUEUES", str(max(1, min(args.streams, 24))))      # one hardware queue per stream (default is 4)
    import numpy as np
    import torch
    from zkp_amd.engine import Engine, FusedStatement
    from zkp_amd import toolbox as T

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if os.environ.get("ZKP_BENCH_DRYRUN_ONE_GPU"):
            # dry run of the multi-rank logic on a box with a single GPU (all ranks on cuda:0, gloo instead of RCCL):
            # exercises the barriers, the MAX over ranks and the verdict reduction -- not a measurement
            local_rank = 0
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            torch.cuda.set_device(local_rank)
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)
    n_streams = max(1, args.streams)
    engines = [Engine(local_rank) for _ in range(n_streams)]
    streams = [torch.cuda.Stream(device=dev) for _ in range(n_streams)]
    for e_, s_ in zip(engines, streams):
        e_.set_stream(s_.cuda_stream)          # engine work and the torch copies of one batch share one HIP stream
        for kv in args.engine_opt:
            e_.set_option(int(kv.split("=")[0]), int(kv.split("=")[1]))
    eng = engines[0]

    n = args.batch
    rng = np.random.default_rng(1000 + rank)
    secrets, inst, common = cmz_instance(eng, n, rng)       # each rank proves / verifies its own range of proofs
    fst = FusedStatement(b"CMZ cred show n=10", *cmz_statement())      # define_proof! label, benches/zkp.rs:29
    for e_ in engines:                         # the issuer parameters are common to every proof (benches/zkp.rs:32): fixed-base tables
        e_.prepare_fixed_points(common)
    t0s = T.Transcript(LABEL).state
    pos = int(t0s[200]) | int(t0s[201]) << 8 | int(t0s[202]) << 16
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    d_ts0 = t(np.stack([t0s] * n))
    d_sec = t(secrets)
    d_tbl = t(np.concatenate([common, inst.reshape(-1, 32)]))              # common || inst rows: the prover's point table
    d_ent = t(rng.integers(0, 256, size=(n, 32), dtype=np.uint8))          # what thread_rng() contributes (prover.rs:82)
    d_w = t(rng.integers(0, 256, size=(11, n, 16), dtype=np.uint8))        # the u128 factors of batch_verifier.rs:179
    n_msm, n_terms, n_bv = 11 * n, 31 * n, 12 + 24 * n
    z8 = lambda *shape: torch.zeros(shape, dtype=torch.uint8, device=dev)
    bufs = []
    for _ in engines:
        b = dict(ts=z8(n, 208), ts2=z8(n, 208), chal=z8(n, 32), resp=z8(n, 21, 32), coms=z8(n, 11, 32), st=z8(n_msm),
                 pts=z8(n_bv, 32), out=z8(32), bst=torch.ones(2, dtype=torch.int32, device=dev))
        b["pts"][: 12 + 13 * n] = d_tbl
        bufs.append(b)
    verdict = torch.ones(1, dtype=torch.int32, device=dev)
    torch.cuda.synchronize()

    def prove(e_, b):
        e_.fused_prove_dev(fst, n, pos, b["ts"].data_ptr(), d_sec.data_ptr(), d_tbl.data_ptr(), d_ent.data_ptr(), b["chal"].data_ptr(),
                           b["resp"].data_ptr(), b["coms"].data_ptr(), b["st"].data_ptr())

    def batch_verify(e_, b):
        e_.fused_batch_verify_dev(fst, n, pos, b["ts2"].data_ptr(), b["pts"].data_ptr(), b["coms"].data_ptr(), b["resp"].data_ptr(),
                                  d_w.data_ptr(), b["out"].data_ptr(), b["bst"].data_ptr())

    graphs = [None] * n_streams

    def enqueue(k):
        e_, b = engines[k], bufs[k]
        with torch.cuda.stream(streams[k]):
            b["ts"].copy_(d_ts0, non_blocking=True)
            b["ts2"].copy_(d_ts0, non_blocking=True)
            prove(e_, b)
            batch_verify(e_, b)

    def step(i):
        # one batch: fresh transcripts, prove all N proofs, batch-verify them.  Consecutive batches go to different engine
        # contexts = different HIP streams, so the narrow phases of one batch (transcripts, Horner, reduction tree)
        # overlap with the wide kernels of the next.  The chain of a batch (2 copies + ~75 kernels) is recorded once per
        # stream as a HIP graph and replayed with one host call per step.
        k = i % n_streams
        if graphs[k] is not None:
            graphs[k].launch()
        else:
            enqueue(k)

    def barrier():
        if dist is not None:
            dist.barrier()
        for e_ in engines:
            e_.synchronize()
        torch.cuda.synchronize()

    for k in range(n_streams):                 # first pass: plans compiled, workspaces sized (nothing may allocate while capturing)
        step(k)
    barrier()
    if not args.no_graphs:
        for k in range(n_streams):
            engines[k].capture_begin()
            enqueue(k)
            graphs[k] = engines[k].capture_end()
    for i in range(max(args.warmup, 1) * n_streams):
        step(i)
    barrier()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(i)
    t_enqueued = time.perf_counter() - t0
    if dist is not None:
        # the only cross-GPU exchange of the path: AND of the per-GPU verdict bits (int32 MIN all-reduce over RCCL)
        for e_ in engines:
            e_.synchronize()
        ok_local = all(int(b["bst"].abs().sum().item()) == 0 and not bool(b["out"].any().item()) for b in bufs)
        verdict.fill_(1 if ok_local else 0)
        dist.all_reduce(verdict, op=dist.ReduceOp.MIN)
    barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    for b in bufs:
        assert not bool(b["st"].any().item()), "prover: an input point failed to decode"
        assert int(b["bst"].abs().sum().item()) == 0 and not bool(b["out"].any().item()), "the batch of fresh proofs did not verify"
    assert all(bool((b["chal"] == bufs[0]["chal"]).all().item()) and bool((b["resp"] == bufs[0]["resp"]).all().item()) for b in bufs), "streams disagree"
    # the proofs must be REAL proofs: a flipped response bit makes the batch check fail
    b = bufs[0]
    with torch.cuda.stream(streams[0]):
        b["resp"][n // 2, 3, 0] ^= 1
        b["ts2"].copy_(d_ts0)
        batch_verify(eng, b)
    eng.synchronize()
    torch.cuda.synchronize()
    assert bool(b["out"].any().item()) or int(b["bst"].abs().sum().item()) != 0, "a corrupted proof passed the batch check"

    # ---- per-kernel timing with HIP events on the engine's stream (separate, profiled passes) ---------
    eng.set_profiling(True)
    reps = 5
    k_prove = {}
    k_verify = {}
    with torch.cuda.stream(streams[0]):
        for _ in range(reps):
            b["ts"].copy_(d_ts0)
            b["ts2"].copy_(d_ts0)
            prove(eng, b)
            km, tot = eng.last_timing()
            for k, v in km.items():
                k_prove[k] = k_prove.get(k, 0.0) + v / reps
            k_prove["total"] = k_prove.get("total", 0.0) + tot / reps
            batch_verify(eng, b)
            km, tot = eng.last_timing()
            for k, v in km.items():
                k_verify[k] = k_verify.get(k, 0.0) + v / reps
            k_verify["total"] = k_verify.get("total", 0.0) + tot / reps
    eng.set_profiling(False)

    if rank != 0:
        if dist is not None:
            dist.destroy_process_group()
        return

    ms_per_step = elapsed * 1e3 / args.steps
    value = world * n * args.steps / elapsed
    # dominant kernel: k_terms_split<CT> (one launch per step).  Algorithmic bytes per launch:
    # 64 B per (scalar, point) term in + 32 B per MSM out = 2,336 B per CMZ proof.
    algo_bytes = 64.0 * n_terms + 32.0 * n_msm
    t_terms = k_prove["terms"] * 1e-3
    achieved = algo_bytes / t_terms / 1e9 if t_terms > 0 else 0.0
    # executed v_mad_u64_u32 in that launch (fe_sq = 62, fe_mul = 98):
    #   term on a per-proof point (11 of 31 per proof), comb walk: 16 windows x (4 doublings + 4 additions) + 1 addition
    #   term on a common point X_1..X_10, A (20 of 31), fixed-base walk: 65 mixed additions (7 mul each)
    mads_comb_term = 16 * (16 * 62 + (3 + 3 + 3 + 4 + 4 * 8) * 98) + 8 * 98
    mads_fixed_term = 65 * 7 * 98
    mads = n * (11 * mads_comb_term + 20 * mads_fixed_term)
    valu = mads / t_terms if t_terms > 0 else 0.0
    traffic = None
    step_valu = None
    kernel_valu = None
    pmc = os.path.join(ROOT, "profiles", "r01_pmc_counters.json")       # rocprofv3 --pmc passes of this same command
    if os.path.exists(pmc) and n == 4096:
        try:
            pj = json.load(open(pmc))
            traffic = pj["k_terms_split<true>"]["hbm_bytes_per_launch"]
            kernel_valu = pj["k_terms_split<true>"]["SQ_INSTS_VALU"]
            step_valu = pj["_step_totals"]["valu_wave_instructions_per_step"]
        except Exception:
            pass
    msm_only = lambda d: sum(d.get(k, 0.0) for k in ("decode", "terms", "reduce", "sort", "bucket", "combine"))
    out = {
        "metric": METRIC, "value": value, "unit": "proofs/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": ms_per_step, "host_enqueue_ms_per_step": t_enqueued * 1e3 / args.steps, "hip_graphs": not args.no_graphs,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u32x9 (29-bit limbs, u64 accumulate)",
        "data": "synthetic",
        "config": {"workload": "CMZ'13 10-hidden-attribute credential, batch of %d proofs per GPU: complete proving (Merlin transcripts, "
                               "blindings, 11 constant-time commitment MSMs / 31 terms per proof, challenges, responses) + complete batch "
                               "verification of those proofs (transcripts, coefficient build, one MSM of 12 + 24 N terms)" % n,
                   "batch_per_gpu": n, "streams": n_streams, "sharding": "independent proof ranges per GPU, AND of verdict bits"},
        "prove_proofs_per_s": world * n / (k_prove["total"] * 1e-3),
        "batch_verifies_per_s": world * n / (k_verify["total"] * 1e-3),
        "msm_only_proofs_per_s": {"prove": world * n / (msm_only(k_prove) * 1e-3), "batch_verify": world * n / (msm_only(k_verify) * 1e-3)},
        "kernel_ms": {"prove": k_prove, "batch_verify": k_verify},
        "roofline": {"bound": "hbm", "kernel": "k_terms_split<CT>", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                     "note": "integer-VALU bound by construction (SURVEY.md 8(d)); see valu_* for the binding roofline",
                     "valu_achieved_mads_per_s": valu, "valu_peak_mads_per_s": VALU_PEAK_MADS, "valu_frac": valu / VALU_PEAK_MADS,
                     "algorithmic_bytes_per_launch": algo_bytes, "launch_ms": k_prove["terms"]},
    }
    if kernel_valu:
        # every VALU instruction of the dominant kernel (PMC SQ_INSTS_VALU: multiplications, carries, constant-time selects ...)
        # per second of its launch, against the same issue peak
        out["roofline"]["valu_issue_lane_instr_per_s"] = kernel_valu * 64.0 / t_terms
        out["roofline"]["valu_issue_frac"] = kernel_valu * 64.0 / t_terms / VALU_PEAK_MADS
    if step_valu:
        # every VALU instruction of one step (PMC SQ_INSTS_VALU, all kernels) against the measured 4-cycle-class issue peak:
        # how close the pipelined step as a whole runs to the integer-VALU roofline
        lane_instr = step_valu * 64.0
        out["step_valu"] = {"wave_instructions_per_step": step_valu, "lane_instructions_per_s": lane_instr / (ms_per_step * 1e-3) * world / world,
                            "peak_lane_instructions_per_s": VALU_PEAK_MADS, "frac": lane_instr / (ms_per_step * 1e-3) / VALU_PEAK_MADS,
                            "note": "simple 32-bit adds / fma issue at twice this class's rate, so frac slightly understates the headroom"}
    if not args.no_cpu_baseline and world == 1:          # reported at N = 1 only (rank 0's host cores)
        out["cpu_baseline"] = cpu_baseline(n, secrets, inst, common, d_ent.cpu().numpy(), d_w.cpu().numpy())
        out["cpu_baseline_all_cores"] = cpu_baseline_all_cores()
    print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


def cpu_baseline(n, secrets, inst, common, entropy, weights):
    """The oracle's dalek-style CPU port of the SAME flows (Merlin/STROBE, radix-16 constant-time Straus for the
    commitments, scalar arithmetic mod l, Pippenger w = 8 for the batch check), one thread, on a bounded sample of the
    same workload: the first 512 proofs proven one by one, then batch-verified.  The reference's own Rust cannot be
    built on this box (no toolchain)."""
    import numpy as np
    from oracle import cbind as C
    from oracle import model as M
    C.build()
    m = min(n, 512)
    cst = C.Statement.from_model(M.cmz_statement(10))
    coms = np.zeros((m, 11, 32), np.uint8)
    resp = np.zeros((m, 21, 32), np.uint8)
    t0 = time.perf_counter()
    for j in range(m):
        pts = np.concatenate([inst[:, j], common])
        _, er, ek, _ = C.prove(cst, LABEL, secrets[j], pts, entropy[j].tobytes())
        coms[j], resp[j] = ek, er
    t1 = time.perf_counter()
    rc = C.batch_verify(cst, LABEL, m, np.ascontiguousarray(inst[:, :m]), common, coms, resp, np.ascontiguousarray(weights[:, :m]))
    t2 = time.perf_counter()
    assert rc == 0, "oracle: the sample batch did not verify"
    return {"value": m / (t2 - t0), "unit": "proofs/s", "cores": 1, "kind": "port",
            "sample": "%d proofs: proven one by one %.3f s (Merlin + radix-16 constant-time Straus + responses) + one batch "
                      "verification %.3f s (Merlin + coefficients + %d-term Pippenger w=8 incl. decompression); gcc -O3 -march=native, "
                      "5x51-bit limbs" % (m, t1 - t0, t2 - t1, 12 + 24 * m),
            "prove_proofs_per_s": m / (t1 - t0), "batch_verifies_per_s": m / (t2 - t1)}


def cpu_baseline_all_cores():
    """The same port on every CPU the box lets this job use: oracle/cpu_bench.py in a clean
    subprocess, one worker process per usable hardware thread (affinity mask and cgroup quota respected), each proving and
    batch-verifying 1024 CMZ presentations (about 1 s of work per worker)."""
    try:
        outp = subprocess.run([sys.executable, "-m", "oracle.cpu_bench", "--per", "1024"], cwd=ROOT, capture_output=True, text=True, timeout=240)
        return json.loads(outp.stdout.strip().splitlines()[-1])
    except Exception as e:      # a reported baseline, never the measurement: do not fail the bench line over it
        return {"value": None, "unit": "proofs/s", "cores": os.cpu_count(), "kind": "port", "sample": "failed: %r" % (e,)}


if __name__ == "__main__":
    main()
