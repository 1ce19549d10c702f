#!/usr/bin/env python3
"""bench.py -- the hot path of dalek-cryptography/zkp on MI355X.

A "step" = one pass of the hot path over ONE batch of N = 4096 CMZ'13 10-attribute credential
presentations (BASELINE.json configs[1]):
  (i)  prover commitments: the 11 per-proof multiscalar multiplications of prover.rs:94 for every
       proof of the batch (45,056 MSMs / 126,976 terms, ZKP_CT) fused with compression, and
  (ii) batch verification: the single random-linear-combination MSM of batch_verifier.rs:219
       (12 + 24 N = 98,316 terms, with on-GPU decompression) down to the identity test.
Inputs are synthetic (random scalars; valid random ristretto points produced by the engine itself)
and RESIDENT IN HBM before the timed region; host Merlin transcripts are outside this path.

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


def cmz_shape(n):
    """CSR shape of the prover's commitment MSMs for n CMZ proofs (benches/zkp.rs:27-46).
    Point table: [0..11) common X_1..X_10, A ; then per proof j: P_j = 11 + 2j, Q_j = 12 + 2j.
    Scalar order per proof follows the constraints: (m_i, z_i) x 10, then m_1..m_10, minus_z_Q."""
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=4096, help="proofs per GPU per step (BASELINE configs[1]: 4096)")
    ap.add_argument("--streams", type=int, default=8, help="independent batches in flight, each on its own HIP stream / engine context")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    if args.gpus > 1 and "RANK" not in os.environ:
        # convenience: self-launch one process per GPU over RCCL exactly as the driver would
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
               "--master-addr", "127.0.0.1", "--master-port", os.environ.get("MASTER_PORT", "29533"), os.path.abspath(__file__)] + sys.argv[1:]
        sys.exit(subprocess.call(cmd))

    os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")      # one hardware queue per stream (default is 4)
    import numpy as np
    import torch
    from zkp_amd.engine import Engine, ZKP_CT

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)
    eng = Engine(local_rank)
    engines = [eng] + [Engine(local_rank) for _ in range(max(1, args.streams) - 1)]

    n = args.batch
    rng = np.random.default_rng(1000 + rank)

    def rand_scalars(k):
        s = rng.integers(0, 256, size=(k, 32), dtype=np.uint8)
        s[:, 31] &= 0x0f                       # < 2^252 < l: canonical, like every scalar the toolbox produces
        return s

    # ---- synthetic, valid inputs (made by the engine, untimed) --------------------------------------
    base = np.frombuffer(bytes.fromhex("e2f2ae0a6abc4e71a884a961c500515f58e30b6aa582dd8db6a65945e08d2d76"), np.uint8).reshape(1, 32)
    off, pidx, n_pts = cmz_shape(n)
    pts, st = eng.msm_many(np.arange(n_pts + 1, dtype=np.uint32), rand_scalars(n_pts), np.zeros(n_pts, np.uint32), base, ZKP_CT)
    assert not st.any()
    for e_ in engines:                         # the issuer parameters X_1..X_10, A are common to every proof (benches/zkp.rs:32)
        e_.prepare_fixed_points(pts[:11])
    n_msm, n_terms = 11 * n, 31 * n
    blind = rand_scalars(n_terms)              # the blinding scalars b[sc_var] of prover.rs:95
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    d_off, d_pidx, d_pts, d_blind = t(off.view(np.int32)), t(pidx.view(np.int32)), t(pts), t(blind)
    d_coms = torch.zeros((n_msm, 32), dtype=torch.uint8, device=dev)
    d_cstat = torch.zeros(n_msm, dtype=torch.uint8, device=dev)
    # batch-verification MSM: 12 static + (13 instance + 11 commitment) rows x n   (batch_verifier.rs:219-228)
    n_bv = 12 + 24 * n
    inst, _ = eng.msm_many(np.arange(13 * n + 12 + 1, dtype=np.uint32), rand_scalars(13 * n + 12), np.zeros(13 * n + 12, np.uint32), base, ZKP_CT)
    d_bv_pts = torch.zeros((n_bv, 32), dtype=torch.uint8, device=dev)
    d_bv_pts[: 12 + 13 * n] = t(inst)
    # scalars as batch_verifier.rs:173-206 produces them: static coefficients and instance rows are products/sums mod l
    # (uniform), the 11 commitment rows are -r mod l for fresh 128-bit r (:179-183)
    bv_sc = rand_scalars(n_bv)
    L = 2**252 + 27742317777372353535851937790883648493
    r128 = rng.integers(0, 256, size=(11 * n, 16), dtype=np.uint8)
    neg = np.zeros((11 * n, 32), np.uint8)
    for i in range(11 * n):
        neg[i] = np.frombuffer((L - int.from_bytes(r128[i].tobytes(), "little")).to_bytes(32, "little"), np.uint8)
    bv_sc[12 + 13 * n:] = neg
    d_bv_sc = t(bv_sc)
    d_bv_out = torch.zeros(32, dtype=torch.uint8, device=dev)
    d_bv_st = torch.zeros(1, dtype=torch.int32, device=dev)
    verdict = torch.ones(1, dtype=torch.int32, device=dev)
    torch.cuda.synchronize()

    # commitments must be valid encodings for the batch MSM: run the prover half once and splice them in
    eng.msm_many_dev(n_msm, d_off.data_ptr(), d_blind.data_ptr(), d_pidx.data_ptr(), d_pts.data_ptr(), n_pts, n_terms, ZKP_CT,
                     d_coms.data_ptr(), d_cstat.data_ptr())
    eng.synchronize()
    d_bv_pts[12 + 13 * n:] = d_coms
    torch.cuda.synchronize()

    # per-stream output buffers; inputs are shared (read-only)
    outs = [(d_coms, d_cstat, d_bv_out, d_bv_st)] + [
        (torch.zeros_like(d_coms), torch.zeros_like(d_cstat), torch.zeros_like(d_bv_out), torch.zeros_like(d_bv_st)) for _ in engines[1:]]

    def step(i):
        # one batch: (i) commitments of all N proofs, (ii) the batch-verification MSM.  Consecutive batches go to
        # different engine contexts = different HIP streams, so the single-wave tails of one batch (Horner, reduction
        # tree) overlap with the wide kernels of the next.
        e_ = engines[i % len(engines)]
        coms, cstat, bv_out, bv_st = outs[i % len(engines)]
        e_.msm_many_dev(n_msm, d_off.data_ptr(), d_blind.data_ptr(), d_pidx.data_ptr(), d_pts.data_ptr(), n_pts, n_terms,
                        ZKP_CT, coms.data_ptr(), cstat.data_ptr())
        e_.msm_optional_dev(n_bv, d_bv_sc.data_ptr(), d_bv_pts.data_ptr(), bv_out.data_ptr(), bv_st.data_ptr())

    def barrier():
        if dist is not None:
            dist.barrier()
        for e_ in engines:
            e_.synchronize()
        torch.cuda.synchronize()

    for i in range(args.warmup * len(engines)):
        step(i)
    barrier()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(i)
    if dist is not None:
        # the only cross-GPU exchange of the path: AND of the per-GPU verdict bits (int32 MIN all-reduce over RCCL)
        for e_ in engines:
            e_.synchronize()
        ok_local = all(int(o[3].item()) == 0 for o in outs)
        verdict.fill_(1 if ok_local else 0)
        dist.all_reduce(verdict, op=dist.ReduceOp.MIN)
    barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    for o in outs:
        assert not bool(o[1].any().item()) and int(o[3].item()) == 0, "engine reported a decode failure on valid inputs"
    assert all(bool((o[0] == outs[0][0]).all().item()) and bool((o[2] == outs[0][2]).all().item()) for o in outs), "streams disagree"

    # ---- per-kernel timing with HIP events on the engine's stream (separate, profiled passes) ---------
    eng.set_profiling(True)
    reps = 5
    k_prove = {}
    k_verify = {}
    for _ in range(reps):
        eng.msm_many_dev(n_msm, d_off.data_ptr(), d_blind.data_ptr(), d_pidx.data_ptr(), d_pts.data_ptr(), n_pts, n_terms, ZKP_CT,
                         d_coms.data_ptr(), d_cstat.data_ptr())
        km, tot = eng.last_timing()
        for k, v in km.items():
            k_prove[k] = k_prove.get(k, 0.0) + v / reps
        k_prove["total"] = k_prove.get("total", 0.0) + tot / reps
        eng.msm_optional_dev(n_bv, d_bv_sc.data_ptr(), d_bv_pts.data_ptr(), d_bv_out.data_ptr(), d_bv_st.data_ptr())
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
    pmc = os.path.join(ROOT, "profiles", "r01_pmc_counters.json")       # rocprofv3 --pmc passes of this same command
    if os.path.exists(pmc) and n == 4096:
        try:
            traffic = json.load(open(pmc))["k_terms_split<true>"]["hbm_bytes_per_launch"]
        except Exception:
            traffic = None
    out = {
        "metric": METRIC, "value": value, "unit": "proofs/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u32x9 (29-bit limbs, u64 accumulate)",
        "data": "synthetic",
        "config": {"workload": "CMZ'13 10-hidden-attribute credential, batch of %d proofs per GPU: prover commitment MSMs "
                               "(11 MSMs / 31 terms per proof, constant-time) + one batch-verification MSM (12 + 24 N terms)" % n,
                   "batch_per_gpu": n, "streams": len(engines), "sharding": "independent proof ranges per GPU, AND of verdict bits"},
        "prove_proofs_per_s": world * n / (k_prove["total"] * 1e-3),
        "batch_verifies_per_s": world * n / (k_verify["total"] * 1e-3),
        "kernel_ms": {"prove": k_prove, "batch_verify": k_verify},
        "roofline": {"bound": "hbm", "kernel": "k_terms_split<CT>", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                     "note": "integer-VALU bound by construction (SURVEY.md 8(d)); see valu_* for the binding roofline",
                     "valu_achieved_mads_per_s": valu, "valu_peak_mads_per_s": VALU_PEAK_MADS, "valu_frac": valu / VALU_PEAK_MADS,
                     "algorithmic_bytes_per_launch": algo_bytes, "launch_ms": k_prove["terms"]},
    }
    if not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(n, off, pidx, pts, blind, d_bv_sc.cpu().numpy(), d_bv_pts.cpu().numpy())
    print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


def cpu_baseline(n, off, pidx, pts, blind, bv_sc, bv_pts):
    """The oracle's dalek-style CPU port (radix-16 constant-time Straus for the commitments, Pippenger w = 8 for the
    batch check), one thread, on a bounded sample of the SAME workload: the first 512 proofs' commitment MSMs and a
    512-proof batch-verification MSM.  The reference's own Rust backends cannot be built on this box (no toolchain)."""
    import numpy as np
    from oracle import cbind as C
    C.build()
    m = min(n, 512)
    t0 = time.perf_counter()
    C.msm_many(off[: 11 * m + 1], blind[: 31 * m], pidx[: 31 * m], pts, 1)
    t1 = time.perf_counter()
    k = 12 + 24 * m
    C.msm_optional(bv_sc[:k], bv_pts[:k])
    t2 = time.perf_counter()
    return {"value": m / (t2 - t0), "unit": "proofs/s", "cores": 1, "kind": "port",
            "sample": "%d proofs: commitment MSMs %.3f s (radix-16 Straus, constant-time) + one %d-term batch MSM %.3f s "
                      "(Pippenger w=8 incl. decompression); gcc -O3 -march=native, 5x51-bit limbs" % (m, t1 - t0, k, t2 - t1),
            "prove_proofs_per_s": m / (t1 - t0), "batch_verifies_per_s": m / (t2 - t1)}


if __name__ == "__main__":
    main()
