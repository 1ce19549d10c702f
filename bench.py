#!/usr/bin/env python3
"""bench.py -- the hot path of dalek-cryptography/zkp on MI355X.

A "step" = one pass of the hot path over ONE batch of proofs of one statement, everything on the GPU.  Default workload
(--config 2 = BASELINE.json configs[1]): N = 4096 CMZ'13 10-attribute credential presentations (benches/zkp.rs:27-46):
  (i)  PROVE all N proofs (zkp_fused_prove_dev = N x prover.rs:76-112): Merlin transcripts, blinding factors
       from the transcript RNG, the 11 constant-time commitment MSMs per proof (45,056 MSMs / 126,976 terms)
       with compression, challenges, responses;
  (ii) BATCH-VERIFY the N proofs just made (zkp_fused_batch_verify_dev = batch_verifier.rs:67-235):
       transcripts with identity rejection, challenges, the coefficient build mod l, and the single
       random-linear-combination MSM (12 + 24 N = 98,316 terms, decompression on the GPU) down to the
       identity test.
Inputs are synthetic (random witnesses, a consistent instance made by the engine itself, fixed entropy / weights) and
RESIDENT IN HBM before the timed region; every step starts from fresh `Transcript::new(b"Benchmark")` states like the
reference's bench loop.  value = proofs per second that went through the whole step.

Batches per call (--batches-per-call K, config.batches_per_call): a server that proves / verifies batch after batch does not
have to hand them to the GPU one by one.  K steps = K batches of N proofs travel in ONE call chain: the K N proofs are proven
by one zkp_fused_prove_dev call (proving has no batch semantics) and verified by one zkp_fused_batch_verify_many_dev call --
K independent batch verifications (own weights, own static-coefficient sums, own MSM, own verdict: K x
batch_verifier.rs:137-235) through one transcript launch, one coefficient grid and one segmented Pippenger.  The narrow
kernels of a single batch (transcripts, table chains, bucket tree, Horner) are then K times wider and a handful of streams
fills the chip instead of 20 - 25.  K = 1 is the round-2 loop.  The timed region still covers exactly --steps batches.

What `value` does NOT contain (the reference pays it on the host in every call): drawing the 32 bytes of thread_rng per proof
(prover.rs:82) and the u128 weights (batch_verifier.rs:179) -- entropy and weights are fixed arrays resident in HBM, so every
step re-proves the same proofs -- and host-side transcript work.  "e2e_host_buffers" in the line is the same flow through
the host toolbox (host buffers in and out over PCIe, OS entropy + ChaCha20 for blindings and weights) for comparison.

Other workloads (one JSON line each, same keys):
  --config 3       BASELINE configs[2]: BatchVerifier over 2^20 mixed DLEQ proofs -- 2^19 in define_proof! form (1 + 5 N
                   terms, benches/zkp.rs:49) + 2^19 in constraint-API form (static G, H: 2 + 4 N terms, benches/dleq.rs:188-241);
                   a step = both batch verifications of proofs made beforehand (untimed) by this engine's prover
  --config 4share  one GPU's share of configs[3]: CMZ, 524,288 proofs per step (prove + batch verify)
  --config 5share  one GPU's share of configs[4]: the 64-term wide statement, 32,768 proofs per step (prove + batch verify)

  python bench.py [--gpus N] [--steps K] [--warmup W] [--config 2|3|4share|5share] [--batch n] [--batches-per-call K] [--streams S]
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py --gpus N ...

Prints ONE JSON line (rank 0).  Everything in it is measured in this run, except the PMC-derived fields (roofline.traffic,
*_valu_busy, step_valu), which need a rocprofv3 --pmc pass: they are taken from --pmc-json only when that file was
collected from exactly these kernel sources (sha256 of zkp_amd/csrc), and say so in "pmc_source"; otherwise they are null.
"""
import argparse
import datetime
import hashlib
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

METRIC = "proofs/sec + batch-verifies/sec, CMZ13 10-attr credential, 1/2/4/8 MI355X"
# The VALU ceiling is counted in ISSUE SLOTS (round 4): a SIMD issues one VALU instruction per 4-cycle slot, or two when both are of the 2-cycle class
# (v_and / v_add_u32 / v_mov ...) -- SQ_ACTIVE_INST_VALU2 counts those second instructions; an isolated 2-cycle instruction between v_mad_u64_u32 costs a
# whole slot (profiles/r04_valu_mix_microbench.txt).  valu_busy = 4 x (SQ_INSTS_VALU - SQ_ACTIVE_INST_VALU2) / (1024 SIMDs x GRBM_GUI_ACTIVE / 8) per kernel
# (one PMC pass, the chip's own clock), and issue cycles / (1024 x 2.4 GHz nameplate x ms_per_step) for the timed step -- <= 1 by construction.
# profiles/r05_opcode_mix.json (tools/opcode_mix.py) is the static opcode mix of every kernel, keyed to the kernel sources: information, not the weights.
OPCODE_MIX = os.path.join("profiles", "r06_opcode_mix.json")
HBM_PEAK_GBS = 8000.0         # MI355X_MICROARCH.md
# how the constant-time prover MSMs (prover.rs:94) pick the table entry a secret digit names: ZKP_OPT_CT_LOOKUP (include/zkp_mi355x.h)
CT_SCHEDULES = {0: "lane crossbar: rows in registers, entries by ds_bpermute_b32 from one half of the wavefront, 8-entry private rows scanned with v_cndmask -- constant time BY CONSTRUCTION "
                   "(no address, bank or branch depends on a secret; ZKP_OPT_CT_LOOKUP = 0, the default)",
                1: "masked scans over every row (curve25519-dalek's LookupTable::select) -- constant time by construction (ZKP_OPT_CT_LOOKUP = 1)",
                2: "rows replicated in LDS and read at the digit's index from banks no other lane of the service group uses -- constant time under the LDS bank model, "
                   "NOT by construction (ZKP_OPT_CT_LOOKUP = 2, the default of rounds 2 - 4)"}
LABEL = b"Benchmark"
BASE = bytes.fromhex("e2f2ae0a6abc4e71a884a961c500515f58e30b6aa582dd8db6a65945e08d2d76")
PMC_PATTERN = os.path.join("profiles", "r06_pmc_counters_cfg%s_k%d.json")   # one counter file per workload and batches-per-call (tools/collect_profiles.sh)
# what the HIP-event timing kinds of zkp_ctx_last_timing are, per flow: (kernel names as rocprofv3 prints them, launches per call).
# roofline.kernel is chosen among the GROUPS below by kernel name, summed across flows (k_transcript_run* = 3 launches per step).
KERNELS = {
    ("prove", "transcript"): ("k_transcript_run", 2), ("prove", "tables"): ("k_comb_tables_lane<16>", 1),
    ("prove", "terms"): ("k_terms_split<true, 16, false>", 1), ("prove", "reduce"): ("k_encode_prepare + k_encode_invert + k_encode_finish", 3),
    ("prove", "sort"): ("k_stmt_classify", 1), ("prove", "decode"): ("k_decode_affine", 1),
    ("prove", "scalars"): ("k_blind_scalars + k_responses", 2),
    ("batch_verify", "transcript"): ("k_transcript_run", 1), ("batch_verify", "decode"): ("k_pip_prepare<c>", 1),
    ("batch_verify", "sort"): ("k_pip_tile_hist/total/scan/base/scatter", 5), ("batch_verify", "bucket"): ("k_pip_vmap + k_pip_bucket_part + k_pip_bucket_merge", 3),
    ("batch_verify", "combine"): ("k_pip_reduce_lvl x levels + k_pip_combine (last level + Horner)", 4),
    ("batch_verify", "scalars"): ("k_batch_after_transcript + k_coeff_build + k_coeff_static_final", 3),
}


# ---- statements ----------------------------------------------------------------------------------------------------
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


def _index_form(secrets, points, cons):
    pi = {name: i for i, (name, _) in enumerate(points)}
    si = {name: i for i, name in enumerate(secrets)}
    return secrets, points, [(pi[l], [(si[s], pi[p]) for s, p in lc]) for l, lc in cons]


def cmz_statement():
    """cred_show_10 (benches/zkp.rs:27-46) in the argument form of zkp_amd.engine.FusedStatement: define_proof! allocates
    the secrets, then the instance points, then the common points (macros.rs:215-242)."""
    secrets = [b"m_%d" % i for i in range(1, 11)] + [b"z_%d" % i for i in range(1, 11)] + [b"minus_z_Q"]
    inst = [b"C_%d" % i for i in range(1, 11)] + [b"P", b"Q", b"V"]
    common = [b"X_%d" % i for i in range(1, 11)] + [b"A", b"B"]
    cons = [(b"C_%d" % i, [(b"m_%d" % i, b"P"), (b"z_%d" % i, b"A")]) for i in range(1, 11)]
    cons.append((b"V", [(b"m_%d" % i, b"X_%d" % i) for i in range(1, 11)] + [(b"minus_z_Q", b"Q")]))
    return _index_form(secrets, [(x, False) for x in inst] + [(x, True) for x in common], cons)


def dleq_macro_statement():
    """define_proof! {dleq, "DLEQ proof", (x), (A, B, H), (G) : A = (x * G), B = (x * H)}  (benches/zkp.rs:49)"""
    return _index_form([b"x"], [(b"A", False), (b"B", False), (b"H", False), (b"G", True)], [(b"A", [(b"x", b"G")]), (b"B", [(b"x", b"H")])])


def dleq_capi_statement():
    """benches/dleq.rs:188-241: the constraint-system form; the static points G, H are allocated before the instance points"""
    return _index_form([b"x"], [(b"G", True), (b"H", True), (b"A", False), (b"B", False)], [(b"A", [(b"x", b"G")]), (b"B", [(b"x", b"H")])])


def w64_statement():
    """the wide statement of configs[4]: Q = sum_{i < 64} x_i * G_i, all generators common"""
    xs = [b"x_%d" % i for i in range(64)]
    gs = [b"G_%d" % i for i in range(64)]
    return _index_form(xs, [(b"Q", False)] + [(g, True) for g in gs], [(b"Q", [(x, g) for x, g in zip(xs, gs)])])


def w64_constraints_statement(terms_per_constraint=1):
    """the OTHER reading of configs[4] ("64-constraint Schnorr"): 64 constraints Q_i = x_i * G_i (+ y_i * G_(i+1 mod 64)), the 64 generators common
    (one fixed-base table each) -- 64 (128) secrets, 64 instance left-hand sides, 64 commitments per proof; the prover's 64 MSMs have 1 (2) terms each"""
    xs = [b"x_%d" % i for i in range(64)]
    ys = [b"y_%d" % i for i in range(64)] if terms_per_constraint == 2 else []
    points = [(b"Q_%d" % i, False) for i in range(64)] + [(b"G_%d" % i, True) for i in range(64)]
    cons = [(b"Q_%d" % i, [(b"x_%d" % i, b"G_%d" % i)] + ([(b"y_%d" % i, b"G_%d" % ((i + 1) % 64))] if ys else [])) for i in range(64)]
    return _index_form(xs + ys, points, cons)


W64_FORMS = {"terms": w64_statement, "constraints": lambda: w64_constraints_statement(1), "constraints2": lambda: w64_constraints_statement(2)}

WORKLOADS = {
    # name: (description, [(statement label, statement fn, share of the batch, flows)], default batch, default streams (0 = auto), default steps)
    # (--config 2 packs K batches into one call chain: pick_call_shape)
    "2": ("CMZ'13 10-hidden-attribute credential, batch of %d proofs per GPU: complete proving (Merlin transcripts, blindings, 11 constant-time "
          "commitment MSMs / 31 terms per proof, challenges, responses) + complete batch verification of those proofs (transcripts, coefficient "
          "build, one MSM of 12 + 24 N terms)", [(b"CMZ cred show n=10", cmz_statement, 1.0, ("prove", "batch_verify"))], 4096, 0, 1000),
    "3": ("BatchVerifier over %d mixed DLEQ proofs per GPU: half in define_proof! form (one MSM of 1 + 5 N terms), half in constraint-API form "
          "(static G, H: 2 + 4 N terms); complete batch verifications (transcripts, coefficient build, MSM incl. decompression)",
          [(b"DLEQ proof", dleq_macro_statement, 0.5, ("batch_verify",)), (b"DLEQProof", dleq_capi_statement, 0.5, ("batch_verify",))], 1 << 20, 2, 6),
    "4share": ("one GPU's share of 2^22 CMZ'13 proofs over 8 GPUs: %d proofs per step, complete proving + complete batch verification",
               [(b"CMZ cred show n=10", cmz_statement, 1.0, ("prove", "batch_verify"))], 1 << 19, 2, 6),
    "5share": ("one GPU's share of 2^18 proofs of the 64-term wide statement over 8 GPUs: %d proofs per step, complete proving + complete batch "
               "verification (64 + 2 N terms)", [(b"W64", w64_statement, 1.0, ("prove", "batch_verify"))], 1 << 15, 8, 48),
}


def make_instance(eng, st, n, rng):
    """n consistent assignments of a statement: secrets [n][m][32], inst [ni][n][32], common [ns][32].  Common points and the
    instance points no constraint defines are random multiples of the basepoint; every left-hand side is computed from its
    constraint with the engine's own MSMs (untimed set-up)."""
    import numpy as np
    from zkp_amd.engine import ZKP_VARTIME          # set-up only (and its kernels then carry other names than the timed, constant-time ones)
    secrets_l, points, cons = st
    m = len(secrets_l)

    def rs(k):
        s = rng.integers(0, 256, size=(k, 32), dtype=np.uint8)
        s[:, 31] &= 0x0f                       # < 2^252 < l
        return s

    com_rank, inst_rank = {}, {}
    for i, (_, c) in enumerate(points):
        (com_rank if c else inst_rank)[i] = len(com_rank if c else inst_rank)
    ns, ni = len(com_rank), len(inst_rank)
    lhs = [l for l, _ in cons]
    assert all(l in inst_rank for l in lhs) and len(set(lhs)) == len(lhs)
    free = [i for i in inst_rank if i not in lhs]
    base = np.frombuffer(BASE, np.uint8).reshape(1, 32)
    k = ns + len(free) * n
    pts, st8 = eng.msm_many(np.arange(k + 1, dtype=np.uint32), rs(k), np.zeros(k, np.uint32), base, ZKP_VARTIME)
    assert not st8.any()
    common = np.ascontiguousarray(pts[:ns])
    inst = np.zeros((ni, n, 32), np.uint8)
    for a, i in enumerate(free):
        inst[inst_rank[i]] = pts[ns + a * n: ns + (a + 1) * n]
    secrets = rs(n * m).reshape(n, m, 32)
    if cons:
        j = np.arange(n, dtype=np.uint32)
        terms = [(s, p) for _, lc in cons for s, p in lc]
        T = len(terms)
        pidx = np.zeros((n, T), np.uint32)
        for t, (_, p) in enumerate(terms):
            pidx[:, t] = com_rank[p] if p in com_rank else ns + inst_rank[p] * n + j
        sidx = np.array([s for s, _ in terms], np.int64)
        off_one = np.cumsum([0] + [len(lc) for _, lc in cons])[:-1].astype(np.uint32)
        off = np.concatenate([(off_one[None, :] + np.uint32(T) * j[:, None]).reshape(-1), np.array([T * n], np.uint32)]).astype(np.uint32)
        table = np.concatenate([common, inst.reshape(-1, 32)])
        out, st8 = eng.msm_many(off, np.ascontiguousarray(secrets[:, sidx]).reshape(-1, 32), pidx.reshape(-1), table, ZKP_VARTIME)
        assert not st8.any()
        out = out.reshape(n, len(cons), 32)
        for kk, l in enumerate(lhs):
            inst[inst_rank[l]] = out[:, kk]
    return secrets, np.ascontiguousarray(inst), common


def cmz_instance(eng, n, rng):
    """n consistent CMZ presentations: witnesses, instance points [13][n][32] (C_1..C_10, P, Q, V), common [12][32]."""
    return make_instance(eng, cmz_statement(), n, rng)


# ---- "1 point decode per distinct point + 1 scalar-mult-equivalent per term; report field-mults actually executed alongside" ----------------------
MADS_PER_FE_MUL, MADS_PER_FE_SQ = 98, 62          # v_mad_u64_u32 of one 9 x 29-bit multiplication (81 + 17 for the fold) / squaring (45 + 17): zkp_amd/csrc/fe25519.h


def cmz_field_mult_model(n, K=5):
    """Field multiplications per CMZ proof of one bench step (prove + batch verify), three ways.  Unit: one multiplication = 98 v_mad_u64_u32; a squaring
    counts 62 / 98 of one (what fe_sq executes).
      floor      the algorithm-independent reference count SURVEY 8(d) defines: per distinct point one decode (254 S + 11 M), per (scalar, point) term one
                 scalar-multiplication equivalent = signed radix-16 double-and-add on extended coordinates, 252 doublings (4 S + 4 M) + 64 additions (8 M).
                 It is a unit, not a bound: fixed-base tables and Pippenger do fewer operations per term than one scalar multiplication.
      algorithm  what THIS library's algorithms ask for at the call shape of `value` (point operations of DESIGN.md section 5, counted, not measured).
      (executed = v_mad_u64_u32 the hardware counted / 98 comes from the PMC file.)"""
    S = MADS_PER_FE_SQ / MADS_PER_FE_MUL
    dbl, dblf, add8, add7, dec = 4 * S + 4, 4 * S + 3, 8.0, 7.0, 254 * S + 11     # dblf: a doubling whose result is doubled again needs no T coordinate (ge_double<false>)
    smul = 252 * dbl + 64 * add8
    terms_p, terms_v = 31, 24 + 12.0 / n
    floor = {"prove": terms_p * smul + (2 + 11.0 / n) * dec, "batch_verify": terms_v * (smul + dec)}
    # prove (throughput schedule, calls of K x n proofs): P and Q decoded; a 16-teeth comb table (per tooth a run of 16 doublings to the next tooth's base, four
    # doublings and three additions for the eight multiples; 129 conversions to the cached form, 1 M each); 20 fixed-base terms of 37 mixed additions; P's 10 terms
    # on the grouped walk (4 windows x (4 doublings + 16 additions) + the carry window's addition); Q on a table like P's, or in wide calls on a signed radix-16 ladder
    # of its own (7 operations for its eight multiples, 64 x 4 doublings, 65 additions); 20 additions of partial sums; the batched encoder (~25 M per commitment).
    # Common points: 11 fixed-base tables per CONTEXT, not per call.
    run4 = 3 * dblf + dbl
    table = 16 * (15 * dblf + dbl) + 64 * dbl + 48 * add8 + 129
    walk = 4 * run4 + 65 * add8
    wide = K * n * terms_p >= 250000              # zkp_ctx::kWideCallTerms
    prove = {"decode": 2 * dec, "comb_tables": (1 if wide else 2) * table, "fixed_base_terms": 20 * 37 * add7,
             "comb_terms": 10 * walk + ((4 * dbl + 3 * add8 + 64 * run4 + 65 * add8) if wide else walk), "sum_and_encode": 20 * add8 + 11 * 25.0}
    # batch verify: 24 decodes; Pippenger with 11-bit windows: one mixed addition per term and populated window -- 23 windows for the 13 instance rows, 12 for the 11
    # commitment rows, whose coefficients -r are 128 bits wide after sign folding (batch_verifier.rs:183) --, the bucket tree (2^10 buckets x 24 windows, two additions
    # each, shared by the batch's n proofs) and the 253-doubling Horner tail (shared likewise)
    verify = {"decode": terms_v * dec, "bucket_additions": (13 * 23 + 11 * 12 + 12.0 * 23 / n) * add7, "bucket_tree_and_horner": (24 * 1024 * 2 * add8 + 253 * dbl) / n}
    algo = {"prove": sum(prove.values()), "batch_verify": sum(verify.values())}
    return {"floor": floor, "algorithm": algo, "algorithm_by_phase": {"prove": prove, "batch_verify": verify},
            "floor_per_proof": sum(floor.values()), "algorithm_per_proof": sum(algo.values())}


def pick_streams(steps):
    """Round-2 loop (one batch per call, --batches-per-call 1): batches in flight.  A batch is a chain of ~32 kernels, several of
    them only a few dozen wavefronts wide, so the chip is filled by running independent batches side by side (measured at 200
    steps: 16 streams 4.61, 20 streams 4.85, 25 streams 4.98, 28 streams 4.84, 32 streams 4.59, 40 streams 3.72 M proofs/s).
    With K timed steps over S streams the last round of batches runs with K mod S streams busy; pick S in 12..25 that leaves
    the fewest idle slots (ties: more streams)."""
    if steps <= 25:
        return max(1, steps)
    return min(range(12, 26), key=lambda s: ((-steps) % s, -s))


MAX_BATCHES_PER_CALL = 50      # 204,800 CMZ proofs per call (workspace ~ 8 GB per stream)


def pick_call_shape(steps, want_k=0, want_streams=0):
    """(K, S) for --config 2: K batches per call chain, S call chains in flight.  K must divide --steps (exactly --steps batches
    are timed).  Wide calls fill the chip on their own, so few streams are needed: four of them overlap one call's remaining
    narrow kernels (the K Horner quads, the inversion of the batched encoder, table and transcript chains of the smaller calls)
    with the others' wide ones.  Default: the largest divisor of steps that is <= 50 and leaves at least 4 calls, 4 streams --
    measured (profiles/r03_ab_experiments.txt): 20 steps: 1 x 20 calls 3.9, 2 x 10 4.6, 4 x 5 5.6, 5 x 4 6.15, 10 x 2 5.9, 20 x 1
    5.6 M proofs/s; 1000 steps: 5 x 4 streams 6.1, 10 x 4 6.5, 20 x 4 6.8, 25 x 8 6.8, 50 x 4 6.85, 50 x 2 6.7."""
    if want_k > 0:
        if steps % want_k:
            raise SystemExit("--batches-per-call must divide --steps (exactly --steps batches are timed)")
        k = want_k
    else:
        k = max([d for d in range(1, MAX_BATCHES_PER_CALL + 1) if steps % d == 0 and steps // d >= min(4, steps)] or [1])
    calls = steps // k
    s = want_streams if want_streams > 0 else (pick_streams(steps) if k == 1 and steps > 4 else 4)
    return k, max(1, min(s, calls))


_HIP = None


def _priority_stream(dev, priority):
    """a torch stream object over hipStreamCreateWithPriority(non-blocking, priority): torch's own pools offer "high" and "normal" only"""
    import ctypes
    import torch
    global _HIP
    if _HIP is None:
        _HIP = ctypes.CDLL("libamdhip64.so")
    torch.cuda.set_device(dev)
    h = ctypes.c_void_p()
    rc = _HIP.hipStreamCreateWithPriority(ctypes.byref(h), ctypes.c_uint(1), ctypes.c_int(priority))
    if rc != 0:
        raise RuntimeError("hipStreamCreateWithPriority(%d) -> %d" % (priority, rc))
    return torch.cuda.ExternalStream(h.value, device=dev)


def _sleep_cycles_per_ms(dev):
    """torch.cuda._sleep counts in the device's own clock ticks: calibrate once"""
    import torch
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda._sleep(1000)
    torch.cuda.synchronize()
    n = 2_000_000
    a.record()
    torch.cuda._sleep(n)
    b.record()
    torch.cuda.synchronize()
    return n / max(a.elapsed_time(b), 1e-3)


def source_sha256():
    """sha256 over the kernel sources: keys PMC counter files to the code they were collected from"""
    h = hashlib.sha256()
    d = os.path.join(ROOT, "zkp_amd", "csrc")
    for f in sorted(os.listdir(d)):
        if f.endswith((".hip", ".h")):
            h.update(f.encode())
            h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()


def init_distributed(world, rank, local_rank):
    """One process per GPU.  The default group is gloo (host side: barriers, the MAX of the elapsed times, and the agreement on
    which backend carries the verdict); the single data-path exchange -- the AND of the per-GPU verdict bits, an int32 MIN
    all-reduce -- goes over RCCL (backend "nccl").  If RCCL cannot be brought up on every rank the job still completes:
    the 4-byte verdict then travels over gloo and the JSON line says "collective": "gloo-fallback"."""
    import torch
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=600))
    info = {"collective": "gloo-fallback", "backend_world_size": dist.get_world_size(), "rccl_error": None}
    group = None
    dry = bool(os.environ.get("ZKP_BENCH_DRYRUN_ONE_GPU"))
    ok = 1
    try:
        if dry:
            # dry run of the multi-rank control flow on a box with a single GPU (all ranks on cuda:0, not a measurement):
            # RCCL cannot span ranks that share a device, so this run takes -- and thereby tests -- the fallback branch
            raise RuntimeError("dry run: all ranks share GPU 0")
        if os.environ.get("ZKP_BENCH_FORCE_RCCL_FAILURE"):
            raise RuntimeError("forced by ZKP_BENCH_FORCE_RCCL_FAILURE (test of the fallback)")
        group = dist.new_group(backend="nccl", timeout=datetime.timedelta(seconds=180))
        probe = torch.ones(1, dtype=torch.int32, device=torch.device("cuda", local_rank))
        dist.all_reduce(probe, op=dist.ReduceOp.SUM, group=group)
        torch.cuda.synchronize()
        ok = 1 if int(probe.item()) == world else 0
    except Exception as e:                      # noqa: BLE001 -- any failure of the RCCL bring-up selects the fallback
        info["rccl_error"] = repr(e)[:300]
        ok = 0
    agree = torch.tensor([ok], dtype=torch.int32)
    dist.all_reduce(agree, op=dist.ReduceOp.MIN)          # gloo: every rank takes the same branch
    if int(agree.item()) == 1:
        info["collective"] = "rccl"
    else:
        group = None
    return dist, group, info


class Ctx:
    """what every workload of a run shares: the rank's device, the process groups"""
    pass


def run_workload(cx, args, cfg, n, steps, warmup, K, n_streams, primary):
    """One workload of WORKLOADS on this rank: set-up (untimed), the timed loop (barrier + synchronize on both sides, MAX over
    ranks), and -- for the primary workload of the run -- the single-flow lines, per-kernel HIP-event timing and roofline.
    A call = K steps = K batches of n proofs in one chain (K = 1 except --config 2).  Returns the result dictionary."""
    import numpy as np
    import torch
    from zkp_amd.engine import Engine, FusedStatement
    from zkp_amd import toolbox as T

    desc, parts, _, _, _ = WORKLOADS[cfg]
    dist, group, dev, rank, world = cx.dist, cx.group, cx.dev, cx.rank, cx.world
    assert steps % K == 0
    calls = steps // K
    n_streams = max(1, min(n_streams, calls))
    engines = [Engine(cx.local_rank) for _ in range(n_streams)]
    prios = [int(x) for x in args.stream_priorities.split(",")] if getattr(args, "stream_priorities", "") else []
    if prios:
        # call chains on streams of different HIP priorities (-1 high, 0 normal, 1 low; cycled over the streams): identical chains that start together
        # stay in phase -- their narrow kernels coincide -- unless the dispatcher prefers one queue's wide kernel over the others'
        streams = [_priority_stream(dev, prios[k % len(prios)]) for k in range(n_streams)]
    else:
        streams = [torch.cuda.Stream(device=dev) for _ in range(n_streams)]
    stagger_cycles = [0] * n_streams
    if getattr(args, "stagger_ms", 0.0) > 0.0 and n_streams > 1:
        per_ms = _sleep_cycles_per_ms(dev)
        stagger_cycles = [int(k * args.stagger_ms * per_ms) for k in range(n_streams)]
    for e_, s_ in zip(engines, streams):
        e_.set_stream(s_.cuda_stream)          # engine work and the torch copies of one call share one HIP stream
        for kv in args.engine_opt:
            e_.set_option(int(kv.split("=")[0]), int(kv.split("=")[1]))
    eng = engines[0]
    rng = np.random.default_rng(1000 + rank)   # each rank proves / verifies its own range of proofs
    t0s = T.Transcript(LABEL).state
    pos = int(t0s[200]) | int(t0s[201]) << 8 | int(t0s[202]) << 16
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    z8 = lambda *shape: torch.zeros(shape, dtype=torch.uint8, device=dev)

    # ---- per part of the workload: statement, instance, device inputs; per stream: outputs ------------------------------
    class Part:
        pass
    ps = []
    for label, st_fn, share, flows in parts:
        p = Part()
        p.st = st_fn()
        p.n_each = max(1, int(round(n * share)))       # proofs of one batch (= one step)
        p.K = K
        p.n = p.n_each * K                             # proofs of one call
        p.flows = flows
        p.secrets, p.inst, p.common = make_instance(eng, p.st, p.n, rng)
        p.fst = FusedStatement(label, *p.st)
        p.m, p.nc = len(p.st[0]), len(p.st[2])
        p.ns, p.ni = len(p.common), len(p.inst)
        p.T = sum(len(lc) for _, lc in p.st[2])
        p.n_bv_each = p.ns + (p.ni + p.nc) * p.n_each  # terms of one batch's MSM (batch_verifier.rs:219-228)
        p.n_pts = p.ns + (p.ni + p.nc) * p.n
        p.d_ts0 = t(np.stack([t0s] * p.n))
        p.d_sec = t(p.secrets)
        p.d_tbl = t(np.concatenate([p.common, p.inst.reshape(-1, 32)]))        # common || inst rows: the prover's point table
        p.d_ent = t(rng.integers(0, 256, size=(p.n, 32), dtype=np.uint8))      # what thread_rng() contributes (prover.rs:82)
        p.d_w = t(rng.integers(0, 256, size=(p.nc, p.n, 16), dtype=np.uint8))  # the u128 factors of batch_verifier.rs:179
        p.bufs = []
        for _ in engines:
            b = dict(ts=z8(p.n, 208), ts2=z8(p.n, 208), ts3=z8(p.n, 208), chal=z8(p.n, 32), resp=z8(p.n, p.m, 32), coms=z8(p.n, p.nc, 32),
                     st=z8(p.n * p.nc), pts=z8(p.n_pts, 32), out=z8(K, 32), bst=torch.ones((K, 2), dtype=torch.int32, device=dev), res=z8(p.n))
            b["pts"][: p.ns + p.ni * p.n] = p.d_tbl
            p.bufs.append(b)
        ps.append(p)
    for e_ in engines:                         # the common points of a statement are the same for every proof (benches/zkp.rs:32): fixed-base tables
        e_.prepare_fixed_points(np.concatenate([p.common for p in ps]))
    verdict = torch.ones(1, dtype=torch.int32, device=dev)
    torch.cuda.synchronize()

    def prove(e_, p, b):
        e_.fused_prove_dev(p.fst, p.n, pos, b["ts"].data_ptr(), p.d_sec.data_ptr(), p.d_tbl.data_ptr(), p.d_ent.data_ptr(), b["chal"].data_ptr(),
                           b["resp"].data_ptr(), b["coms"].data_ptr(), b["st"].data_ptr())

    def batch_verify(e_, p, b):
        e_.fused_batch_verify_many_dev(p.fst, p.K, p.n_each, pos, b["ts2"].data_ptr(), b["pts"].data_ptr(), b["coms"].data_ptr(), b["resp"].data_ptr(),
                                       p.d_w.data_ptr(), b["out"].data_ptr(), b["bst"].data_ptr())

    def verify_compact(e_, p, b):
        e_.fused_verify_compact_dev(p.fst, p.n, pos, b["ts3"].data_ptr(), p.d_tbl.data_ptr(), b["chal"].data_ptr(), b["resp"].data_ptr(), b["res"].data_ptr())

    def verify_batchable(e_, p, b):
        # verifier.rs:123-173 per proof: operand table = the prover's point table, then the proofs' commitments [n][nc]
        if "tbl_each" not in b:
            b["tbl_each"] = z8(p.ns + p.ni * p.n + p.n * p.nc, 32)
            b["tbl_each"][: p.ns + p.ni * p.n] = p.d_tbl
        b["tbl_each"][p.ns + p.ni * p.n:].copy_(b["coms"].reshape(-1, 32), non_blocking=True)
        e_.fused_verify_batchable_dev(p.fst, p.n, pos, b["ts3"].data_ptr(), b["tbl_each"].data_ptr(), b["resp"].data_ptr(), p.d_w.data_ptr(), b["res"].data_ptr())

    def enqueue(k, which=None):
        """one call (= K steps) on stream k: per part, fresh transcripts and its flows (or only the flow `which`)"""
        e_ = engines[k]
        with torch.cuda.stream(streams[k]):
            if stagger_cycles[k]:
                torch.cuda._sleep(stagger_cycles[k])           # (--stagger-ms: chain k starts k x stagger late; one idle wavefront, inside the timed region)
            for p in ps:
                b = p.bufs[k]
                for flow in ((which,) if which else p.flows):
                    if flow == "prove":
                        b["ts"].copy_(p.d_ts0, non_blocking=True)
                        prove(e_, p, b)
                    elif flow == "batch_verify":
                        b["ts2"].copy_(p.d_ts0, non_blocking=True)
                        batch_verify(e_, p, b)
                    elif flow == "verify_batchable":
                        b["ts3"].copy_(p.d_ts0, non_blocking=True)
                        verify_batchable(e_, p, b)
                    else:
                        b["ts3"].copy_(p.d_ts0, non_blocking=True)
                        verify_compact(e_, p, b)

    def barrier():
        if dist is not None:
            dist.barrier()
        for e_ in engines:
            e_.synchronize()
        torch.cuda.synchronize()
        if getattr(dist, "in_process", False):
            dist.barrier()         # ranks that are threads of one process: nobody starts a stream capture while another is still inside a device synchronize

    def batches_ok(b):
        return int(b["bst"].abs().sum().item()) == 0 and not bool(b["out"].any().item())

    # proofs for the parts whose step only verifies: made once, untimed
    for k in range(n_streams):
        with torch.cuda.stream(streams[k]):
            for p in ps:
                if "prove" not in p.flows:
                    p.bufs[k]["ts"].copy_(p.d_ts0, non_blocking=True)
                    prove(engines[k], p, p.bufs[k])
    barrier()

    def timed_loop(which, n_calls, warm):
        """n_calls calls over the streams, each replaying the per-stream HIP graph of its chain: barrier + synchronize on both
        sides; returns (elapsed, host time to enqueue, per-rank elapsed)."""
        graphs = [None] * n_streams
        for k in range(n_streams):             # first pass: plans compiled, workspaces sized (nothing may allocate while capturing)
            enqueue(k, which)
        barrier()
        if not args.no_graphs:
            for k in range(n_streams):
                with engines[k].capture() as cap:              # a failing call aborts the capture instead of leaving the stream capturing
                    enqueue(k, which)
                graphs[k] = cap.graph

        def call(i):
            # consecutive calls go to different engine contexts = different HIP streams, so the narrow phases of one call
            # (table chains, Horner quads, the encoder's inversion) overlap with the wide kernels of the others
            k = i % n_streams
            if graphs[k] is not None:
                graphs[k].launch()
            else:
                enqueue(k, which)

        for i in range(max(warm, 1) * n_streams):
            call(i)
        barrier()
        t0 = time.perf_counter()
        for i in range(n_calls):
            call(i)
        t_enq = time.perf_counter() - t0
        if which is None and dist is not None:
            # the only cross-GPU exchange of the path: AND of the per-GPU verdict bits (int32 MIN all-reduce)
            for e_ in engines:
                e_.synchronize()
            ok_local = all(batches_ok(b) for p in ps if "batch_verify" in p.flows for b in p.bufs)
            verdict.fill_(1 if ok_local else 0)
            if group is not None:
                dist.all_reduce(verdict, op=dist.ReduceOp.MIN, group=group)           # RCCL
            else:
                v = verdict.cpu()
                dist.all_reduce(v, op=dist.ReduceOp.MIN)                              # gloo fallback
                verdict.copy_(v)
        barrier()
        elapsed = own = time.perf_counter() - t0
        if dist is not None:
            tt = torch.tensor([elapsed], dtype=torch.float64)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            elapsed = float(tt.item())
        for g in graphs:
            if g is not None:
                g.close()
        return elapsed, t_enq, own

    elapsed, t_enqueued, own_elapsed = timed_loop(None, calls, warmup)
    if dist is not None:
        assert int(verdict.item()) == 1, "a rank reported a batch that did not verify"
    for p in ps:
        for b in p.bufs:
            assert not bool(b["st"].any().item()), "prover: an input point failed to decode"
            if "batch_verify" in p.flows:
                assert batches_ok(b), "a batch of fresh proofs did not verify"
        assert all(bool((b["chal"] == p.bufs[0]["chal"]).all().item()) and bool((b["resp"] == p.bufs[0]["resp"]).all().item()) for b in p.bufs), "streams disagree"
    # the proofs must be REAL proofs: a flipped response bit makes its batch check fail -- and only that one
    for p in ps:
        if "batch_verify" not in p.flows:
            continue
        b = p.bufs[0]
        kb = p.K // 2
        j = kb * p.n_each + p.n_each // 2
        with torch.cuda.stream(streams[0]):
            b["resp"][j, p.m // 2, 0] ^= 1
            b["ts2"].copy_(p.d_ts0)
            batch_verify(eng, p, b)
        eng.synchronize()
        torch.cuda.synchronize()
        bad = [i for i in range(p.K) if bool(b["out"][i].any().item()) or int(b["bst"][i].abs().sum().item()) != 0]
        assert bad == [kb], "a corrupted proof must fail exactly its own batch: %r" % (bad,)
        with torch.cuda.stream(streams[0]):
            b["resp"][j, p.m // 2, 0] ^= 1
        torch.cuda.synchronize()

    total_n = sum(p.n_each for p in ps)                # proofs per step
    res = {"elapsed": elapsed, "own_elapsed": own_elapsed, "ms_per_step": elapsed * 1e3 / steps, "value": world * total_n * steps / elapsed,
           "host_enqueue_ms_per_step": t_enqueued * 1e3 / steps, "streams": n_streams, "ps": ps, "total_n": total_n}
    if not primary:
        for e_ in engines:
            e_.close()
        return res

    # ---- pipelined single-flow lines (same loop, one flow) -------------------------------------------------------------
    flow_lines = {}
    if not args.no_flow_lines:
        flows_present = sorted({f for p in ps for f in p.flows})
        for which in flows_present + (["verify_compact", "verify_batchable"] if cfg != "3" else []):
            if len(flows_present) == 1 and which == flows_present[0]:
                continue                       # the step itself is that flow
            el, _, _ = timed_loop(which, calls, 1)
            flow_lines[which] = world * total_n * steps / el
            if which in ("verify_compact", "verify_batchable"):
                assert all(not bool(b["res"].any().item()) for p in ps for b in p.bufs), which + " rejected a fresh proof"
    res["flow_lines"] = flow_lines

    # ---- per-kernel timing with HIP events on the engine's stream (separate, profiled passes on one stream) -------------
    eng.set_profiling(True)
    reps = 5 if total_n * K <= (1 << 16) else 2
    kms, samples, launched = {}, {}, {}
    flows_timed = sorted({f for p in ps for f in p.flows}) + ([] if args.no_flow_lines or cfg == "3" else ["verify_compact", "verify_batchable"])
    with torch.cuda.stream(streams[0]):
        for _ in range(reps):
            for p in ps:
                b = p.bufs[0]
                for flow in flows_timed:
                    if flow not in ("verify_compact", "verify_batchable") and flow not in p.flows:
                        continue
                    if flow == "prove":
                        b["ts"].copy_(p.d_ts0)
                        prove(eng, p, b)
                    elif flow == "batch_verify":
                        b["ts2"].copy_(p.d_ts0)
                        batch_verify(eng, p, b)
                    elif flow == "verify_batchable":
                        b["ts3"].copy_(p.d_ts0)
                        verify_batchable(eng, p, b)
                    else:
                        b["ts3"].copy_(p.d_ts0)
                        verify_compact(eng, p, b)
                    km, tot = eng.last_timing()
                    d = samples.setdefault((flow, id(p)), [])
                    d.append(dict(km, total=tot))
                    launched.setdefault(flow, {}).update(eng.last_kernels())     # the variants the library picked, by their rocprofv3 names
    eng.set_profiling(False)
    # per (flow, part): the repetition with the median total -- one host hiccup between two launches (a busy node: 10 ms seen) would
    # otherwise show up as kernel time of whatever phase it fell into; then summed over the parts of the workload
    for (flow, _), reps_ in samples.items():
        pick = sorted(reps_, key=lambda r: r["total"])[(len(reps_) - 1) // 2]
        d = kms.setdefault(flow, {})
        for k, v in pick.items():
            d[k] = d.get(k, 0.0) + v
    res["kms"] = kms                           # ms per CALL (K steps) on a lone stream
    res["launched"] = launched                 # {flow: {timing kind: [kernel names]}} as reported by zkp_ctx_last_kernels
    res["engines"] = engines
    return res


def e2e_host_buffers(eng, n=4096, reps=3, pinned=False):
    """The CMZ step THROUGH the host toolbox (include/zkp_toolbox.h): host buffers in and out over PCIe, the per-proof entropy
    and the u128 weights drawn inside the call (getrandom + ChaCha20) -- what `value` leaves out.  Best of `reps`.
    pinned: every buffer the caller hands over lives in pinned host memory (zkp_host_alloc, what INTEGRATION.md recommends to a service): the engine's
    copies are then DMA that returns at once -- inputs cross the link under the first kernels, the commitments leave under the last ones."""
    import numpy as np
    from zkp_amd import toolbox as T
    mod = T.cmz_module(10)
    rng = np.random.default_rng(77)
    secrets, inst, common = make_instance(eng, cmz_statement(), n, rng)
    t0s = np.stack([T.Transcript(LABEL).state] * n)
    out = None
    if pinned:
        secrets, inst, common = T.pinned_copy(secrets), T.pinned_copy(inst), T.pinned_copy(common)
        out = (T.pinned_empty((n, 32)), T.pinned_empty((n, mod.statement.m, 32)), T.pinned_empty((n, mod.statement.nc, 32)))
        ts_buf = T.pinned_empty((n, 208))
    best = None
    for _ in range(reps + 1):
        if pinned:
            ts = ts_buf
            ts[:] = t0s
        else:
            ts = t0s.copy()
        t0 = time.perf_counter()
        chal, resp, coms = T.prove_batch(eng, mod.statement, ts, secrets, inst, common, out=out)     # entropy = None: from the OS
        t1 = time.perf_counter()
        if pinned:
            ts[:] = t0s
        else:
            ts = t0s.copy()
        t1b = time.perf_counter()
        T.batch_verify(eng, mod.statement, ts, inst, common, coms, resp)                             # weights = None: from the OS
        t2 = time.perf_counter()
        cur = (t1 - t0, t2 - t1b)
        if best is None or sum(cur) < sum(best):
            best = cur
    return {"proofs": n, "prove_ms": best[0] * 1e3, "batch_verify_ms": best[1] * 1e3, "proofs_per_s": n / sum(best),
            "host_buffers": "pinned (zkp_host_alloc)" if pinned else "ordinary (pageable) memory",
            "note": "zkp_prove_batch + zkp_batch_verify (fused route) on host buffers: PCIe copies, OS entropy for the blindings (prover.rs:82) and "
                    "ChaCha20 weights (batch_verifier.rs:179) included; one synchronous call each, nothing pipelined (see `pipelined`)"}


def e2e_threads(n=4096, K=10, threads=6, jobs=24, mem="pageable", opts=((14, 1),), device=0):
    """The same step issued the plainest way a service would: `threads` host threads, each with its OWN context, calling the SYNCHRONOUS
    zkp_prove_batch + zkp_batch_verify_many of include/zkp_toolbox.h on K x n proofs (ordinary memory allocated once per thread and reused, OS entropy
    and weights inside the calls).  opts = zkp_ctx_set_option pairs for every context (14 = ZKP_OPT_SYNC_SCHEDULE: 1 = the synchronous calls run the
    jobs' throughput schedule).  mem = "fresh": the Python wrappers, which allocate their outputs per call."""
    import ctypes
    import threading
    import numpy as np
    from zkp_amd import toolbox as T
    from zkp_amd.engine import Engine
    L, _p = T.lib(), T._p
    st = T.cmz_module(10).statement
    nn = n * K
    e0 = Engine(device)
    secrets, inst, common = make_instance(e0, cmz_statement(), nn, np.random.default_rng(78))
    e0.close()
    ts0 = np.stack([T.Transcript(LABEL).state] * nn)
    engines = [Engine(device) for _ in range(threads)]
    for e in engines:
        for k, v in opts:
            e.set_option(int(k), int(v))
    bufs = {}
    if mem != "fresh":
        mk = T.pinned_copy if mem == "pinned" else (lambda x: np.array(x, copy=True, order="C"))
        for e in engines:
            bufs[id(e)] = dict(ts0=mk(ts0), ts=mk(ts0), sec=mk(secrets), inst=mk(inst), com=mk(common), chal=mk(np.ones((nn, 32), np.uint8)),
                               resp=mk(np.ones((nn, st.m, 32), np.uint8)), coms=mk(np.ones((nn, st.nc, 32), np.uint8)), v=(ctypes.c_int * K)())
    todo, lock, errors = {"left": 0}, threading.Lock(), []

    def worker(e):
        try:
            while True:
                with lock:
                    if todo["left"] <= 0:
                        return
                    todo["left"] -= 1
                if mem == "fresh":
                    ts = ts0.copy()
                    chal, resp, coms = T.prove_batch(e, st, ts, secrets, inst, common)
                    ts = ts0.copy()
                    v = T.batch_verify_many(e, st, K, ts, inst, common, coms, resp)
                else:
                    b = bufs[id(e)]
                    b["ts"][...] = b["ts0"]
                    rc = L.zkp_prove_batch(e._h, st._h, ctypes.c_uint32(nn), _p(b["ts"]), _p(b["sec"]), _p(b["inst"]), _p(b["com"]), None, 0, _p(b["chal"]), _p(b["resp"]), _p(b["coms"]))
                    assert rc == 0, "zkp_prove_batch -> %d" % rc
                    b["ts"][...] = b["ts0"]
                    rc = L.zkp_batch_verify_many(e._h, st._h, ctypes.c_uint32(K), ctypes.c_uint32(n), ctypes.c_uint32(nn), _p(b["ts"]), _p(b["inst"]), _p(b["com"]), _p(b["coms"]),
                                                 _p(b["resp"]), None, 0, b["v"])
                    assert rc == 0, "zkp_batch_verify_many -> %d" % rc
                    v = np.array(list(b["v"]))
                assert not v.any(), "a batch of fresh proofs did not verify"
        except Exception as ex:          # noqa: BLE001
            errors.append(ex)

    def run(n_jobs):
        todo["left"] = n_jobs
        th = [threading.Thread(target=worker, args=(e,)) for e in engines]
        t0 = time.perf_counter()
        for t in th:
            t.start()
        for t in th:
            t.join()
        return time.perf_counter() - t0

    try:
        run(2 * threads)                                # plans, workspaces, tables
        el = run(jobs)
    finally:
        for e in engines:
            e.close()
    if errors:
        raise errors[0]
    return {"proofs_per_s": jobs * nn / el, "threads": threads, "proofs_per_batch": n, "batches_per_call": K, "calls": 2 * jobs, "elapsed_ms": el * 1e3, "host_buffers": mem,
            "engine_options": {str(k): int(v) for k, v in opts},
            "note": "synchronous zkp_prove_batch + zkp_batch_verify_many from `threads` host threads, one context each (no zkp_pipe, no jobs); ZKP_OPT_SYNC_SCHEDULE = 1 "
                    "puts the synchronous calls on the throughput schedule"}


def e2e_pipelined(n=4096, K=5, contexts=6, jobs=24, pinned=True, devices=(0,), submit_threads=-1):
    """The CMZ step through the BOUNDARY a Rust caller would bind, pipelined (include/zkp_toolbox.h: zkp_pipe): host buffers in, host
    buffers out, `jobs` prove jobs of K batches of n proofs each (zkp_prove_batch_submit) and, for every finished prove job, one
    zkp_batch_verify_many_submit over the proofs it returned (K verdicts) -- at most `contexts` jobs in flight per device, entropy
    (prover.rs:82) and weights (batch_verifier.rs:179) drawn inside the calls (getrandom seed per job, ChaCha20 on the device), every
    proof starting from Transcript::new(b"Benchmark") as in the reference's bench loop.  pinned = the caller's buffers live in pinned
    memory (zkp_host_alloc); otherwise they are ordinary numpy arrays and the pipe stages them through its own pinned rings."""
    import collections
    import numpy as np
    from zkp_amd import toolbox as T
    from zkp_amd.engine import Engine
    mod = T.cmz_module(10)
    st = mod.statement
    nn = n * K
    eng = Engine(devices[0])
    secrets, inst, common = make_instance(eng, cmz_statement(), nn, np.random.default_rng(78))
    eng.close()
    t0s = T.Transcript(LABEL).state
    mk = T.pinned_copy if pinned else np.ascontiguousarray
    empty = T.pinned_empty if pinned else (lambda shape: np.zeros(shape, np.uint8))
    a_sec, a_inst, a_com, a_t0 = mk(secrets), mk(inst), mk(common), mk(t0s)
    in_flight = contexts * len(devices)
    free_out = collections.deque(dict(chal=empty((nn, 32)), resp=empty((nn, st.m, 32)), coms=empty((nn, st.nc, 32))) for _ in range(2 * in_flight + 2))
    with T.Pipe(tuple(devices), contexts) as pipe:
        host = {"submit_s": 0.0, "wait_s": 0.0}
        phases = {"P": [0.0, 0.0, 0.0, 0], "V": [0.0, 0.0, 0.0, 0]}
        if submit_threads >= 0:
            pipe.set_submit_threads(submit_threads)        # (default: one submitter thread per entry of the device list when there is more than one)
        if os.environ.get("ZKP_BENCH_JOB_TIMING"):
            pipe.set_profiling(True)
        if os.environ.get("ZKP_X_DEFER") is not None:          # (tools/x/ab_defer.py: A/B of ZKP_OPT_JOB_DEFER_D2H)
            pipe.set_option(13, int(os.environ["ZKP_X_DEFER"]))

        def run(n_jobs):
            pending, to_verify = collections.deque(), collections.deque()
            submitted = verified = 0
            while verified < n_jobs:
                ta = time.perf_counter()
                if to_verify and len(pending) < in_flight:
                    o = to_verify.popleft()
                    pending.append(("V", o, pipe.submit_batch_verify_many(st, K, n, a_t0, a_inst, a_com, o["coms"], o["resp"])))
                    host["submit_s"] += time.perf_counter() - ta
                elif submitted < n_jobs and len(pending) < in_flight and free_out:
                    o = free_out.popleft()          # (first in, first out: the warm-up pass touches every buffer set once)
                    pending.append(("P", o, pipe.submit_prove(st, nn, a_t0, a_sec, a_inst, a_com, out=o)))
                    submitted += 1
                    host["submit_s"] += time.perf_counter() - ta
                else:
                    # retire a job that is done if there is one (jobs on different contexts finish in any order), else wait for the oldest
                    idx = next((i for i, e in enumerate(pending) if e[2].done()), 0)
                    kind, o, job = pending[idx]
                    del pending[idx]
                    outs = job.wait()
                    host["wait_s"] += time.perf_counter() - ta
                    if os.environ.get("ZKP_BENCH_JOB_TIMING"):
                        ms = pipe.job_timing(job.context)
                        for q in range(3):
                            phases[kind][q] += ms[q]
                        phases[kind][3] += 1
                    if kind == "P":
                        to_verify.append(o)
                    else:
                        assert not outs[0].any(), "a batch of fresh proofs did not verify"
                        free_out.append(o)
                        verified += 1
        run(len(free_out) + in_flight)                  # plans compiled, workspaces and staging rings sized, fixed-base tables built, every buffer touched
        host["submit_s"] = host["wait_s"] = 0.0
        t0 = time.perf_counter()
        run(jobs)
        el = time.perf_counter() - t0
        # a corrupted proof must fail exactly its own batch
        o = free_out[0]
        job = pipe.submit_prove(st, nn, a_t0, a_sec, a_inst, a_com, out=o)
        job.wait()
        o["resp"][(K // 2) * n + 7, 3, 0] ^= 1
        (v,) = pipe.submit_batch_verify_many(st, K, n, a_t0, a_inst, a_com, o["coms"], o["resp"]).wait()
        assert [int(x) for x in v] == [1 if b == K // 2 else 0 for b in range(K)], "a corrupted proof must fail exactly its own batch"
    h2d = jobs * (2 * a_inst.nbytes + a_sec.nbytes + o["coms"].nbytes + o["resp"].nbytes + 2 * (a_com.nbytes + 208))
    d2h = jobs * (o["chal"].nbytes + o["resp"].nbytes + o["coms"].nbytes)
    return {"proofs_per_s": jobs * nn / el, "proofs": jobs * nn, "elapsed_ms": el * 1e3, "proofs_per_batch": n, "batches_per_submit": K, "jobs_in_flight": in_flight,
            "job_stream_ms": {k: {"h2d": v[0] / max(v[3], 1), "kernels": v[1] / max(v[3], 1), "d2h": v[2] / max(v[3], 1)} for k, v in phases.items()} if os.environ.get("ZKP_BENCH_JOB_TIMING") else None,
            "devices": list(devices), "submit_threads": (len(devices) > 1) if submit_threads < 0 else bool(submit_threads), "host_buffers": "pinned (zkp_host_alloc)" if pinned else "ordinary memory, staged through the pipe's pinned rings",
            "h2d_GBps": h2d / el / 1e9, "d2h_GBps": d2h / el / 1e9, "host_ms_in_submit": host["submit_s"] * 1e3, "host_ms_in_wait": host["wait_s"] * 1e3, "bytes_per_proof": {"h2d": h2d / (jobs * nn), "d2h": d2h / (jobs * nn)}}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None, help="timed steps = batches (default: 1000 for --config 2 = 0.7 s, fewer for the large workloads)")
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", default="2", choices=sorted(WORKLOADS), help="BASELINE.json workload (2 = configs[1], the metric's configuration)")
    ap.add_argument("--batch", type=int, default=None, help="proofs per GPU per step (default: the workload's)")
    ap.add_argument("--batches-per-call", type=int, default=0, help="--config 2: K steps (batches) travel in one call chain -- one wide prove call + "
                                                                     "one K-batch verification; must divide --steps (0 = automatic, 1 = one batch per call)")
    ap.add_argument("--streams", type=int, default=0, help="independent call chains in flight, each on its own HIP stream / engine context (0 = automatic)")
    ap.add_argument("--stream-priorities", default="", help="experiment: HIP stream priorities of the call chains, cycled over the streams (-1 high, 0 normal, 1 low), e.g. -1,0,0,1")
    ap.add_argument("--stagger-ms", type=float, default=0.0, help="experiment: call chain k starts k x this many ms late (a one-wavefront sleep at the head of its stream, inside the timed region)")
    ap.add_argument("--max-hw-queues", type=int, default=8, help="cap of GPU_MAX_HW_QUEUES (one hardware queue per stream up to this)")
    ap.add_argument("--no-graphs", action="store_true", help="enqueue every kernel of every call from the host instead of replaying one "
                                                            "HIP graph per stream")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-flow-lines", action="store_true", help="skip the pipelined prove-only / batch-verify-only / verify_compact measurements")
    ap.add_argument("--no-multi-configs", action="store_true", help="--gpus > 1: skip the strong-scaling sub-records of BASELINE configs[3] / configs[4]")
    ap.add_argument("--multi-total4", type=int, default=1 << 22, help="--gpus > 1: total proofs of the configs[3] sub-record (2^22 CMZ proofs over the GPUs)")
    ap.add_argument("--multi-total5", type=int, default=1 << 18, help="--gpus > 1: total proofs of the configs[4] sub-record (2^18 W64 proofs over the GPUs)")
    ap.add_argument("--pmc-json", default=None, help="rocprofv3 --pmc summary (tools/pmc_summary.py); default profiles/r05_pmc_counters_cfg<config>_k<batches per call>.json; "
                                                     "used only if its source hash and workload shape match")
    ap.add_argument("--in-process", action="store_true", help="--gpus N > 1 WITHOUT torchrun / gloo / RCCL: this one process drives the N GPUs, one host "
                                                             "thread and one set of engine contexts per GPU, the verdict AND is taken on the host "
                                                             "(the C-ABI counterpart is zkp_pipe over N devices); same JSON line")
    ap.add_argument("--pipe-batches", type=int, default=10, help="e2e_host_buffers.pipelined: batches of --batch proofs per submitted job (rounds 3-4 measured 5: "
                                                                 "profiles/r04_ab_experiments.txt block n has 5 / 10 / 20 side by side)")
    ap.add_argument("--pipe-contexts", type=int, default=6, help="e2e_host_buffers.pipelined: contexts (= jobs in flight) of the zkp_pipe")
    ap.add_argument("--w64-form", default="terms", choices=sorted(W64_FORMS), help="--config 5share: the reading of BASELINE configs[4] -- `terms` = ONE constraint of 64 terms "
                    "(SURVEY section 8), `constraints` = 64 constraints of one term, `constraints2` = 64 constraints of two terms (a per-constraint generator + one shared)")
    ap.add_argument("--no-lone-call", action="store_true", help="--config 2: skip the `lone_call` sub-record (one batch per call, one call chain in flight, on both schedules)")
    ap.add_argument("--no-sustained", action="store_true", help="--config 2: skip the `sustained` sub-record (4000 more steps at 50 batches per call after 0.6 s under load: "
                    "the steady-state rate next to the short timed region) and the `ct` sub-record (the same steps on the other look-ups of ZKP_OPT_CT_LOOKUP)")
    ap.add_argument("--sustained-steps", type=int, default=4000, help="steps of the `sustained` sub-record (a multiple of 50: >= 2 s of GPU time at the default)")
    ap.add_argument("--engine-opt", action="append", default=[], metavar="ID=VALUE",
                    help="zkp_ctx_set_option(ID, VALUE) on every engine context (tuning experiments; results never depend on it)")
    args = ap.parse_args()

    if args.gpus > 1 and "RANK" not in os.environ and not args.in_process:
        # convenience: self-launch one process per GPU exactly as the driver would
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
               "--master-addr", "127.0.0.1", "--master-port", os.environ.get("MASTER_PORT", "29533"), os.path.abspath(__file__)] + sys.argv[1:]
        sys.exit(subprocess.call(cmd))

    if args.w64_form != "terms":
        d5, p5, b5, s5, t5 = WORKLOADS["5share"]
        what = "64 constraints of %d term%s: %d + %d N terms in the batch check" % ((1, "", 64, 128) if args.w64_form == "constraints" else (2, "s", 64, 128))
        WORKLOADS["5share"] = (d5.replace("(64 + 2 N terms)", "(%s)" % what).replace("the 64-term wide statement", "the 64-CONSTRAINT wide statement"),
                               [(p5[0][0], W64_FORMS[args.w64_form]) + tuple(p5[0][2:])], b5, s5, t5)
    desc, parts, def_batch, def_streams, def_steps = WORKLOADS[args.config]
    if args.steps is None:
        args.steps = def_steps
    n = args.batch or def_batch
    if args.config == "2":
        K, n_streams = pick_call_shape(args.steps, args.batches_per_call, args.streams)
    else:
        if args.batches_per_call > 1:
            raise SystemExit("--batches-per-call applies to --config 2 (the other workloads are one wide batch per step)")
        K, n_streams = 1, max(1, min(args.streams or def_streams or pick_streams(args.steps), max(1, args.steps)))
    # one hardware queue per stream in flight (the runtime's default is 4): the call chains of the timed loop, and the 8 contexts of the zkp_pipe
    # behind e2e_host_buffers.pipelined (6 by default; a context whose stream shares a hardware queue with another's is serialised behind it)
    want_q = max(n_streams, 8 if (args.gpus > 1 or (args.config == "2" and not args.no_flow_lines)) else 1)
    os.environ.setdefault("GPU_MAX_HW_QUEUES", str(max(1, min(want_q, args.max_hw_queues))))
    import torch

    if args.in_process and args.gpus > 1 and "RANK" not in os.environ:
        # one process, one host thread per GPU ("one host thread + stream per device ... AND of G verdict bits -- on the host")
        import threading
        torch.cuda.init()
        tg = ThreadGroup(args.gpus)
        errors = []

        def worker(r):
            try:
                tg.bind(r)
                rank_main(args, desc, n, K, n_streams, r, 0 if os.environ.get("ZKP_BENCH_DRYRUN_ONE_GPU") else r, args.gpus, tg)
            except BaseException as e:      # noqa: BLE001 -- a dead rank must not leave the others waiting at a barrier forever
                errors.append(e)
                tg.abort()
        th = [threading.Thread(target=worker, args=(r,)) for r in range(1, args.gpus)]
        for t_ in th:
            t_.start()
        worker(0)
        for t_ in th:
            t_.join()
        if errors:
            raise errors[0]
        return
    rank_main(args, desc, n, K, n_streams, int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")), None)


class ThreadGroup:
    """What bench.py needs of torch.distributed, for ranks that are THREADS of one process (--in-process): barrier and all_reduce of
    small CPU tensors.  No sockets, no gloo, no RCCL -- the only cross-GPU exchange of this path is the AND of verdict bits."""

    in_process = True

    class ReduceOp:
        MIN, MAX, SUM = "min", "max", "sum"

    def __init__(self, world):
        import threading
        self.world = world
        self._bar = threading.Barrier(world)
        self._slots = [None] * world
        self._tls = threading.local()

    def bind(self, rank):
        self._tls.rank = rank

    def abort(self):
        self._bar.abort()

    def barrier(self):
        self._bar.wait()

    def all_reduce(self, t, op=None, group=None):
        import torch
        self._slots[self._tls.rank] = t.detach().cpu().clone()
        self._bar.wait()
        v = torch.stack(self._slots)
        res = v.min(0).values if op == "min" else (v.max(0).values if op == "max" else v.sum(0))
        self._bar.wait()
        t.copy_(res)

    def destroy_process_group(self):
        pass


def rank_main(args, desc, n, K, n_streams, rank_, local_rank_, world_, thread_group):
    import numpy as np
    import torch

    cx = Ctx()
    cx.rank, cx.local_rank, cx.world = rank_, local_rank_, world_
    cx.dist, cx.group, dinfo = None, None, {"collective": None, "backend_world_size": 1, "rccl_error": None}
    if thread_group is not None:
        cx.dist, cx.group = thread_group, None
        dinfo = {"collective": "host-and (in-process: one thread per GPU, no process group)", "backend_world_size": world_, "rccl_error": None}
    elif cx.world > 1:
        if os.environ.get("ZKP_BENCH_DRYRUN_ONE_GPU"):
            cx.local_rank = 0
        else:
            torch.cuda.set_device(cx.local_rank)
        cx.dist, cx.group, dinfo = init_distributed(cx.world, cx.rank, cx.local_rank)
    cx.dev = torch.device("cuda", cx.local_rank)
    torch.cuda.set_device(cx.dev)
    rank, world, dist = cx.rank, cx.world, cx.dist

    r = run_workload(cx, args, args.config, n, args.steps, args.warmup, K, n_streams, primary=True)
    ps, kms, flow_lines, total_n = r["ps"], r["kms"], r["flow_lines"], r["total_n"]
    eng = r["engines"][0]

    # ---- BASELINE configs[3] and configs[4] are 8-GPU configurations: measured whenever the job spans several GPUs -----------
    multi = None
    e2e = None
    if world == 1 and args.config == "2" and not args.no_flow_lines:
        e2e = e2e_host_buffers(eng)
        e2e["pinned"] = e2e_host_buffers(eng, pinned=True)           # the same two synchronous calls when the caller's buffers are pinned
    for e_ in r["engines"]:
        e_.close()
    # ---- the reference-shaped call: ONE batch of n proofs per call, one call chain in flight, inputs resident -- prove, then batch-verify ----
    lone = None
    if args.config == "2" and not args.no_flow_lines and not args.no_lone_call:
        import copy
        torch.cuda.empty_cache()
        lone = {"proofs_per_call": n, "calls_timed": 40, "unit": "proofs/s",
                "note": "K = 1: one prove call of %d proofs, then one batch verification of them, on ONE stream, replayed as a HIP graph, inputs resident in HBM (`value` packs 5 batches "
                        "per call on 4 streams).  throughput_schedule = the library's default for the asynchronous _dev entry points (everything on one stream, one-lane comb tables, "
                        "transcript program A inside the tables' launch); latency_schedule = ZKP_OPT_DEV_OVERLAP = 2: what the synchronous host-pointer calls run -- the point halves "
                        "(decode, comb tables; the batch MSM's decompressions) on a second stream next to the transcript chains, four lanes per comb table" % n}
        for name, opt in (("throughput_schedule", None), ("latency_schedule", "5=2")):
            a2 = copy.copy(args)
            a2.engine_opt = list(args.engine_opt) + ([opt] if opt else [])
            rl = run_workload(cx, a2, "2", n, 40, 3, 1, 1, primary=False)
            lone[name] = {"proofs_per_s": rl["value"], "ms_per_call": rl["ms_per_step"]}
        lone["proofs_per_s"] = max(lone["throughput_schedule"]["proofs_per_s"], lone["latency_schedule"]["proofs_per_s"])
        torch.cuda.empty_cache()
    # ---- what the short timed region cannot show, under the same clock: the steady-state rate, and the same steps on the other look-ups ----
    sustained, ct = None, None
    user_lookup = next((int(kv.split("=")[1]) for kv in args.engine_opt if int(kv.split("=")[0]) == 9), 0)
    if args.config == "2" and not args.no_sustained and not args.no_flow_lines:
        import copy
        torch.cuda.empty_cache()
        ss = max(50, args.sustained_steps // 50 * 50)
        # 5 warm-up calls per stream = 20 calls of 204,800 proofs ~ 0.6 s under load before the clock starts; the timed part is >= 2 s
        rs = run_workload(cx, args, "2", n, ss, 5, 50, 4, primary=False)
        sustained = {"value": rs["value"], "unit": "proofs/s", "steps": ss, "ms_per_step": rs["ms_per_step"], "elapsed_s": rs["elapsed"], "batches_per_call": 50, "streams": rs["streams"],
                     "warmup_calls_per_stream": 5, "note": "same flows, same statement, same process as `value`: %d more steps in calls of 50 batches on 4 streams, timed after 20 untimed calls "
                     "(~0.6 s under load: the chip has settled at the clock it holds under these kernels); `value` stays the short region the driver asks for" % ss}
        torch.cuda.empty_cache()
        if world == 1 and len(eng.ct_lookups) > 1:       # (-DZKP_HOT_W=6 builds only: the shipped library has the crossbar look-up alone)
            ct = {}
            for name, lk in (("secret_indexed_lds_rows", 2), ("masked_scans", 1)):
                a2 = copy.copy(args)
                a2.engine_opt = [kv for kv in args.engine_opt if int(kv.split("=")[0]) != 9] + ["9=%d" % lk]
                rr = run_workload(cx, a2, "2", n, args.steps, args.warmup, K, n_streams, primary=False)
                ct[name] = rr["value"]
                torch.cuda.empty_cache()
    if e2e is not None:
        # the same flows through the zkp_pipe of include/zkp_toolbox.h: jobs of K batches, several in flight, host buffers both ways
        torch.cuda.empty_cache()
        e2e["pipelined"] = e2e_pipelined(n=n, K=args.pipe_batches, contexts=args.pipe_contexts, jobs=24, pinned=True)
        e2e["pipelined_staged"] = e2e_pipelined(n=n, K=args.pipe_batches, contexts=args.pipe_contexts, jobs=24, pinned=False)
        e2e["threads"] = e2e_threads(n=n, K=args.pipe_batches, threads=args.pipe_contexts, jobs=24)
    if world > 1 and args.config == "2" and not args.no_multi_configs:
        multi = {}
        for key, cfg, total, cap, streams_cap in (("4", "4share", args.multi_total4, 1 << 19, 2), ("5", "5share", args.multi_total5, 1 << 15, 8)):
            share = max(1, total // world)                  # strong scaling: the total is fixed, each rank takes a contiguous range of it
            chunk = min(share, cap)                         # per call: at most the per-GPU share of the 8-GPU configuration (workspace ~ 40 GB at 2^19 CMZ proofs)
            csteps = max(1, share // chunk)
            torch.cuda.empty_cache()
            rr = run_workload(cx, args, cfg, chunk, csteps, 1, 1, min(streams_cap, csteps), primary=False)
            tmin = torch.tensor([rr["own_elapsed"]], dtype=torch.float64)
            tmax = tmin.clone()
            dist.all_reduce(tmin, op=dist.ReduceOp.MIN)
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            multi[key] = {"workload": WORKLOADS[cfg][0] % chunk, "baseline_config": "configs[%d]" % (int(key) - 1), "proofs_total": world * chunk * csteps,
                          "proofs_per_gpu": chunk * csteps, "proofs_per_call": chunk, "calls_per_gpu": csteps, "streams": rr["streams"],
                          "value": world * chunk * csteps / rr["elapsed"], "unit": "proofs/s", "scaling": "strong", "elapsed_ms": rr["elapsed"] * 1e3,
                          "ms_per_call": rr["elapsed"] * 1e3 / csteps, "per_rank_elapsed_ms": {"min": float(tmin.item()) * 1e3, "max": float(tmax.item()) * 1e3},
                          "collective": dinfo["collective"], "verdict": "every batch of every rank verified (asserted)"}

    if thread_group is not None and args.config == "2" and not args.no_flow_lines:
        # the same N GPUs through the C-ABI boundary a Rust caller binds: ONE zkp_pipe over all devices, host buffers both ways
        dist.barrier()
        if rank == 0:
            devs = [0] * world if os.environ.get("ZKP_BENCH_DRYRUN_ONE_GPU") else list(range(world))
            e2e = {"pipelined": e2e_pipelined(n=n, K=min(5, K) if n < 4096 else 5, contexts=3, jobs=8 * world, pinned=True, devices=devs)}
        dist.barrier()
    if rank != 0:
        if dist is not None:
            dist.destroy_process_group()
        return

    ms_per_step = r["ms_per_step"]
    value = r["value"]
    # ---- roofline of the dominant kernel: the kernel (by name, summed over the flows that launch it) with the largest share of the
    #      step's kernel time, HIP events on a lone stream -----------------------------------------------------------------------------
    algo = {"prove": sum((64.0 * p.T + 32.0 * p.nc) * p.n for p in ps if "prove" in p.flows),                 # 64 B per term in + 32 B per MSM out
            "batch_verify": sum(64.0 * p.n_bv_each * p.K for p in ps if "batch_verify" in p.flows)}           #              64 B per operand of every MSM of the call
    step_flows = sorted({f for p in ps for f in p.flows})
    # kernel names: the size- / option-dependent variants come from the library itself (zkp_ctx_last_kernels, exact rocprofv3 names), the
    # fixed ones from KERNELS; launches per call of the transcript kind = programs the flow runs on their own (program A may ride in the tables' launch)
    names = dict(KERNELS)
    for f, kinds in r["launched"].items():
        for kind, ks in kinds.items():
            if (f, kind) not in names:
                continue
            launches = names[(f, kind)][1]
            if kind == "transcript" and f == "prove" and any("k_tables_transcript" in x for x in kinds.get("tables", [])):
                launches = 1
            names[(f, kind)] = (" + ".join(ks), launches)
    groups = {}
    for f in step_flows:
        for k, ms in kms[f].items():
            if k == "total" or ms <= 0:
                continue
            nm, launches = names.get((f, k), (k, 1))
            launches *= sum(1 for p in ps if f in p.flows)
            g = groups.setdefault(nm, {"kernels": nm, "ms": 0.0, "launches": 0, "bytes": 0.0, "kinds": []})
            g["ms"] += ms
            g["launches"] += launches
            g["bytes"] += algo[f] * launches
            g["kinds"].append("%s/%s" % (f, k))
    t_all = sum(g["ms"] for g in groups.values())
    by_kernel = [{"kernels": g["kernels"], "timing_kinds": g["kinds"], "launches_per_call": g["launches"], "ms_per_call": g["ms"], "share": g["ms"] / t_all,
                  "avg_launch_ms": g["ms"] / g["launches"], "GB/s": g["bytes"] / (g["ms"] * 1e-3) / 1e9}
                 for g in sorted(groups.values(), key=lambda g: -g["ms"])]
    dom = by_kernel[0]
    achieved = dom["GB/s"]
    roof = {"bound": "hbm", "kernel": "%s  (%d launch%s per call of %d batch%s, %.0f %% of a call's kernel time on a lone stream)"
                                      % (dom["kernels"], dom["launches_per_call"], "" if dom["launches_per_call"] == 1 else "es", K, "" if K == 1 else "es", 100 * dom["share"]),
            "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": None,
            "algorithmic_bytes_per_launch": achieved * 1e9 * dom["avg_launch_ms"] * 1e-3, "launch_ms": dom["avg_launch_ms"],
            "launch_ms_note": "HIP events on the engine's stream around the launch, one call chain in flight (the kernel alone on the chip), the repetition with the median call time; "
                              "launch_ms_rocprof_one_stream (when pmc_source is set) = rocprofv3's average for the same kernel and shape on ONE stream "
                              "(profiles/r05_kernel_stats_cfg<config>_k<K>_one_stream.txt), a few per cent longer -- the profiler's own dispatch overhead; in the timed loop "
                              "several chains share the chip and a launch takes correspondingly longer (profiles/r05_kernel_stats_cfg<config>_k<K>.txt)",
            "note": "integer-VALU bound by construction (SURVEY.md 8(d)): algorithmic bytes = 64 B per (scalar, point) term + 32 B per output, "
                    "so the HBM fraction of ANY kernel of this path is ~1e-3; the binding roofline is step_valu / *_valu_busy (PMC, VALU issue slots)",
            "by_kernel": by_kernel}
    # ---- PMC-derived fields: only from a counter file collected from exactly these sources and this call shape ---------------------------
    sha = source_sha256()
    pmc_source, step_valu, roof_valu = None, None, None
    pmc_rel = args.pmc_json or (PMC_PATTERN % (args.config, K))
    pmc_path = pmc_rel if os.path.isabs(pmc_rel) else os.path.join(ROOT, pmc_rel)
    if os.path.exists(pmc_path):
        try:
            pj = json.load(open(pmc_path))
            shape = pj.get("_workload", {})
            if pj.get("_source_sha256") == sha and shape.get("batch") == n and shape.get("batches_per_call") == K and shape.get("config") == args.config:
                pmc_source = "%s: rocprofv3 --pmc passes of `python bench.py --config %s --steps %s` at kernel-source sha256 %s (tools/collect_profiles.sh)" % (
                    pmc_rel, args.config, shape.get("steps"), sha[:12])
                def rows_of(label):
                    """PMC rows of a kernel group by EXACT name (the names come from zkp_ctx_last_kernels / KERNELS); a label that is a description
                    rather than a name (the fixed multi-kernel kinds of KERNELS) matches by the bare kernel names it lists"""
                    rows = []
                    for part in label.split(" + "):
                        part = part.strip()
                        if part in pj:
                            rows.append(pj[part])
                        else:
                            base = part.split(" ")[0].split("<")[0]
                            rows += [v for k_, v in pj.items() if isinstance(v, dict) and k_.split("<")[0].split("::")[-1] == base.split("::")[-1] and "SQ_INSTS_VALU" in v]
                    return rows

                def busy(rows):
                    """VALU issue slots the rows' instructions took (4 cycles each; co-issued 2-cycle instructions share one) / SIMD clock cycles that went by (one PMC pass)"""
                    if not rows or any("valu_busy" not in v for v in rows):
                        return None
                    return sum(v["valu_issue_cycles"] for v in rows) / sum(v["valu_issue_cycles"] / v["valu_busy"] for v in rows)
                drows = rows_of(dom["kernels"])
                if drows:
                    roof["traffic"] = sum(v.get("hbm_bytes_per_launch", 0.0) for v in drows) or None
                    roof["dominant_kernel_valu_busy"] = busy(drows)
                    if all("avg_us_one_stream" in v for v in drows):
                        roof["launch_ms_rocprof_one_stream"] = sum(v["avg_us_one_stream"] for v in drows) / len(drows) / 1e3
                tname = next((x for x in r["launched"].get("prove", {}).get("terms", [])), None)
                if tname and tname in pj and kms.get("prove", {}).get("terms"):
                    tv = pj[tname]
                    roof["terms_kernel"] = tname
                    roof["terms_kernel_valu_busy"] = busy([tv])
                    roof["terms_kernel_sclk_ghz"] = tv.get("sclk_ghz")
                    if "avg_us_one_stream" in tv:
                        roof["terms_kernel_launch_ms"] = {"hip_events_this_run": kms["prove"]["terms"], "rocprof_one_stream": tv["avg_us_one_stream"] / 1e3,
                                                          "rocprof_pmc_pass": tv.get("us_in_pmc_pass", 0.0) / 1e3 or None}
                    roof["terms_kernel_traffic"] = tv.get("hbm_bytes_per_launch")
                    # does that traffic cost the kernel anything?  wavefront-cycles parked at s_waitcnt / barriers, the L2's hit rate, TCP stall cycles
                    roof["terms_kernel_memory_wait"] = {
                        "wave_parked_frac": tv.get("wave_parked_frac"), "l2_hit_rate": tv.get("l2_hit_rate"),
                        "tcp_pending_stall_cycles_per_launch": tv.get("TCP_PENDING_STALL_CYCLES_sum"), "sq_busy_cycles_per_launch": tv.get("SQ_BUSY_CYCLES"),
                        "issue_stall_frac": (tv["SQ_WAIT_INST_ANY"] / tv["SQ_WAVE_CYCLES"]) if tv.get("SQ_WAIT_INST_ANY") and tv.get("SQ_WAVE_CYCLES") else None,
                        "note": "wave_parked_frac = SQ_WAIT_ANY / SQ_WAVE_CYCLES (a wavefront parked at s_waitcnt or a barrier; with two wavefronts per SIMD one's wait is the other's issue "
                                "slot); issue_stall_frac = SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES (dependency / pipe stalls).  The traffic is 40 x the algorithmic bytes and is NOT free: it "
                                "is bounded by the 1 - terms_kernel_valu_busy of issue slots the kernel leaves empty, all other stalls included"}
                    roof["terms_kernel_algorithmic_bytes"] = algo["prove"]
                    roof["terms_kernel_share_int64"] = {"static": tv.get("share_int64_static"), "dynamic_pmc": tv.get("share_int64_dynamic")}
                st_tot = pj.get("_step_totals", {})
                if st_tot.get("valu_wave_instructions_per_step"):
                    floor, floor_obs = st_tot.get("valu_floor_ms_per_step_nameplate_clock"), st_tot.get("valu_floor_ms_per_step_observed_clock")
                    step_valu = {"wave_instructions_per_step": st_tot["valu_wave_instructions_per_step"], "issue_cycles_per_step": st_tot.get("valu_issue_cycles_per_step"),
                                 "valu_floor_ms_per_step": floor, "frac": floor / ms_per_step if floor else None,
                                 "sclk_ghz_observed": st_tot.get("sclk_ghz_observed"),
                                 "valu_floor_ms_per_step_at_observed_sclk": floor_obs, "frac_at_observed_sclk": floor_obs / ms_per_step if floor_obs else None,
                                 "note": "VALU busy fraction of the timed step in issue slots: every kernel's 4 x (SQ_INSTS_VALU - SQ_ACTIVE_INST_VALU2) (PMC, from pmc_source: one "
                                         "slot of 4 cycles per instruction, two co-issued 2-cycle-class instructions share one), summed over the step's kernels, / (1024 SIMDs x clock x "
                                         "this run's ms_per_step).  frac uses the 2.4 GHz nameplate clock: a hard bound, <= 1 whatever the chip does.  Under this load the chip clocks lower "
                                         "(sclk_ghz_observed = GRBM_GUI_ACTIVE / duration of the step's long kernels in the PMC pass), so frac_at_observed_sclk is the share of the "
                                         "issue cycles that were really there -- an estimate: the clock of the four-stream run is not observable from HIP.  "
                                         "Rounds 1-3 charged 4 cycles to every instruction at one assumed clock; this round first charged 2 to every 2-cycle-class opcode "
                                         "(too few: isolated ones cost 4, profiles/r04_valu_mix_microbench.txt)."}
                    mads = st_tot.get("mad_u64_wave_instructions_per_step")
                    if args.config == "2" and mads:
                        # executed field multiplications next to the reference count and to what the library's own algorithms ask for
                        md = cmz_field_mult_model(n, K)
                        ex = mads * 64.0 / MADS_PER_FE_MUL / n
                        cyc = st_tot.get("valu_issue_cycles_per_step")
                        roof_valu = {"bound": "integer VALU issue slots (v_mad_u64_u32 class, 4 cycles per wave64 instruction per SIMD)",
                                     "field_mults_executed_per_proof": ex, "field_mults_floor_per_proof": md["floor_per_proof"], "field_mults_algorithm_per_proof": md["algorithm_per_proof"],
                                     "executed_over_floor": ex / md["floor_per_proof"], "efficiency": md["algorithm_per_proof"] / ex,
                                     "mad_share_of_issue_slots": 4.0 * mads / cyc if cyc else None, "issue_slot_frac": floor / ms_per_step if floor else None,
                                     "achieved_field_mults_per_s": ex * value, "peak_field_mults_per_s": 1024 * 2.4e9 / 4.0 * 64.0 / MADS_PER_FE_MUL,
                                     "algorithm_by_phase": md["algorithm_by_phase"], "floor_by_flow": md["floor"],
                                     "note": "one field multiplication = 98 v_mad_u64_u32 (a squaring executes 62 and counts 0.63).  executed = the step's v_mad_u64_u32 wave-instructions "
                                             "(SQ_INSTS_VALU_INT64 per kernel x the static share of mads in that counter's class, pmc_source + profiles/r06_opcode_mix.json) x 64 lanes / 98 / proofs; "
                                             "floor = SURVEY 8(d)'s reference count (a decode per distinct point + one radix-16 double-and-add per term) -- a unit, not a bound: executed_over_floor "
                                             "< 1 is what fixed-base tables and Pippenger save; algorithm = the point operations this library's algorithms ask for (bench.py: cmz_field_mult_model); "
                                             "efficiency = algorithm / executed: what is NOT lost to idle lanes, masked scans, recomputation; mad_share_of_issue_slots x issue_slot_frac = the share of "
                                             "the chip's VALU issue slots in the timed region that executed a multiplication's product; peak = every SIMD issuing a mad every slot at 2.4 GHz"}
        except Exception:                       # noqa: BLE001 -- a reported extra, never the measurement
            pmc_source = None
    out = {
        "metric": METRIC, "value": value, "unit": "proofs/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u32x9 (29-bit limbs, u64 accumulate)",
        "data": "synthetic", "seed": 1000,                                      # numpy default_rng(seed + rank) drives every witness, point, entropy and weight
        "config": {"workload": desc % n, "baseline_config": args.config, "w64_form": args.w64_form if args.config == "5share" else None, "batch_per_gpu": n, "batches_per_call": K, "calls": args.steps // K,
                   "streams": r["streams"], "hip_graphs": not args.no_graphs,
                   "ct_schedule": CT_SCHEDULES[user_lookup],
                   "gpu_max_hw_queues": int(os.environ["GPU_MAX_HW_QUEUES"]), "sharding": "independent proof ranges per GPU, AND of verdict bits",
                   "collective": dinfo["collective"], "backend_world_size": dinfo["backend_world_size"],
                   "excluded_from_value": "per-proof entropy (prover.rs:82 thread_rng) and batch weights (batch_verifier.rs:179) are fixed arrays resident in HBM, "
                                          "so every step re-proves the same proofs; host-side RNG / weight generation is not timed (see e2e_host_buffers)"},
        "host_enqueue_ms_per_step": r["host_enqueue_ms_per_step"],
        "pipelined_proofs_per_s": flow_lines,                                   # same loop, one flow only
        "single_stream_proofs_per_s": {f: world * K * sum(p.n_each for p in ps if f in ("verify_compact", "verify_batchable") or f in p.flows) / (kms[f]["total"] * 1e-3) for f in kms},
        "kernel_ms_per_call": kms, "roofline": roof, "pmc_source": pmc_source, "source_sha256": sha,
    }
    if dinfo.get("rccl_error"):
        out["config"]["rccl_error"] = dinfo["rccl_error"]
    if step_valu:
        out["step_valu"] = step_valu
    if roof_valu:
        out["roofline_valu"] = roof_valu
        out["roofline"]["bound"] = "valu"                     # (what binds; achieved / peak / frac stay the HBM figures the contract asks for, roofline_valu has the VALU ones)
    if lone is not None:
        out["lone_call"] = lone
    if sustained is not None:
        try:                                       # its own VALU busy fraction, from the counter file of ITS call shape (K = 50), same rule: only at this source sha
            pj50 = json.load(open(os.path.join(ROOT, PMC_PATTERN % ("2", 50))))
            st50 = pj50.get("_step_totals", {})
            if pj50.get("_source_sha256") == sha and pj50.get("_workload", {}).get("batch") == n and st50.get("valu_floor_ms_per_step_nameplate_clock"):
                f50, f50o = st50["valu_floor_ms_per_step_nameplate_clock"], st50.get("valu_floor_ms_per_step_observed_clock")
                sustained["step_valu"] = {"issue_cycles_per_step": st50.get("valu_issue_cycles_per_step"), "valu_floor_ms_per_step": f50, "frac": f50 / sustained["ms_per_step"],
                                          "sclk_ghz_observed": st50.get("sclk_ghz_observed"), "frac_at_observed_sclk": f50o / sustained["ms_per_step"] if f50o else None,
                                          "pmc_source": PMC_PATTERN % ("2", 50)}
        except Exception:                          # noqa: BLE001 -- a reported extra
            pass
        out["sustained"] = sustained
    if user_lookup in (0, 1):
        out["value_ct_by_construction"] = value                     # the prover's constant-time promise (prover.rs:94) holds by construction on the schedule `value` ran on
    if ct is not None:
        out["ct"] = {"schedule_of_value": CT_SCHEDULES[user_lookup], "by_construction": user_lookup in (0, 1),
                     "value_by_lookup": dict({"lane_crossbar" if user_lookup == 0 else ("masked_scans" if user_lookup == 1 else "secret_indexed_lds_rows"): value}, **ct),
                     "note": "the same timed region (same steps, warm-up, call shape) under each look-up of ZKP_OPT_CT_LOOKUP; every one gives the same bytes (tests/test_gpu_device_entry.py)"}
    if multi is not None:
        out["configs"] = multi
    if e2e is not None:
        out["e2e_host_buffers"] = e2e
    if not args.no_cpu_baseline and world == 1:          # reported at N = 1 only (rank 0's host cores)
        sample = 4096 if args.config == "2" else 1024
        base, oracle_proofs = cpu_baseline(ps, LABEL, sample)
        out["cpu_baseline"] = base
        # every proof the CPU leg recomputed (same secrets, same entropy) against what the TIMED steps left in stream 0's buffers
        checked, equal = 0, True
        for p, (o_chal, o_resp, o_coms) in zip(ps, oracle_proofs):
            mm = len(o_resp)
            b = p.bufs[0]
            g_resp, g_coms = b["resp"][:mm].cpu().numpy(), b["coms"][:mm].cpu().numpy()
            equal = equal and bool((g_resp == o_resp).all()) and bool((g_coms == o_coms).all())
            if "prove" in p.flows:
                equal = equal and bool((b["chal"][:mm].cpu().numpy() == o_chal).all())
            checked += mm
        out["parity_checked"] = {"proofs": checked, "fields": "challenges, responses, commitments of the proofs the timed steps produced (stream 0) vs the oracle's "
                                 "prover on the same secrets and entropy", "equal": equal}
        assert equal, "GPU proofs differ from the oracle's"
        if args.config == "2":
            out["cpu_baseline_all_cores"] = cpu_baseline_all_cores()
    print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


def cpu_baseline(ps, label, sample):
    """The oracle's dalek-style CPU port of the SAME flows (Merlin/STROBE, radix-16 constant-time Straus for the
    commitments, scalar arithmetic mod l, Pippenger w = 6..8 for the batch check), one thread, on a bounded sample of the
    same workload: the first `sample` proofs of each part proven one by one (where the step proves) and batch-verified.
    The reference's own Rust cannot be built on this box (no toolchain)."""
    import numpy as np
    from oracle import cbind as C
    C.build()
    t_prove = t_bv = 0.0
    n_done = 0
    notes = []
    proofs = []                                # per part: (challenges, responses, commitments) the oracle made -- bench.py's parity check
    for p in ps:
        m = min(p.n, sample)
        secrets_l, points, cons = p.st
        names = [nm.decode() for nm, _ in points]
        cst = C.Statement(p.fst._label, [s.decode() for s in secrets_l], [(nm.decode(), c) for nm, c in points],
                          [(names[l], [(secrets_l[s].decode(), names[q]) for s, q in lc]) for l, lc in cons])
        com_rank, inst_rank = {}, {}
        for i, (_, c) in enumerate(points):
            (com_rank if c else inst_rank)[i] = len(com_rank if c else inst_rank)
        coms = np.zeros((m, p.nc, 32), np.uint8)
        resp = np.zeros((m, p.m, 32), np.uint8)
        chal = np.zeros((m, 32), np.uint8)
        ent = p.d_ent[:m].cpu().numpy()
        w = np.ascontiguousarray(p.d_w[:, :m].cpu().numpy())
        t0 = time.perf_counter()
        for j in range(m):
            pts = np.stack([p.common[com_rank[i]] if i in com_rank else p.inst[inst_rank[i], j] for i in range(len(points))])
            ec, er, ek, _ = C.prove(cst, label, p.secrets[j], pts, ent[j].tobytes())
            coms[j], resp[j], chal[j] = ek, er, ec
        t1 = time.perf_counter()
        rc = C.batch_verify(cst, label, m, np.ascontiguousarray(p.inst[:, :m]), p.common, coms, resp, w)
        t2 = time.perf_counter()
        assert rc == 0, "oracle: the sample batch did not verify"
        if "prove" in p.flows:
            t_prove += t1 - t0
        t_bv += t2 - t1
        n_done += m
        proofs.append((chal, resp, coms))
        notes.append("%d proofs of \"%s\": proven one by one %.3f s%s, one batch verification %.3f s (%d-term Pippenger incl. decompression)"
                     % (m, p.fst._label.decode(), t1 - t0, "" if "prove" in p.flows else " (set-up, not counted)", t2 - t1, p.ns + (p.ni + p.nc) * m))
    # the same sample with the MSM inner loops on vectors (oracle/c/simd_ifma.c + simd_x4.inc: the design of curve25519-dalek's simd_backend, which
    # north_star names and which cannot be built here): a MEASURED number instead of a recalled factor -- on the best instruction set the host CPU has
    # (AVX-512 IFMA, else AVX2); where IFMA exists, AVX2 is timed too, on a sub-sample
    ISA_TEXT = {"avx512ifma": "AVX-512 IFMA (vpmadd52, radix 2^51, 4 x 64-bit lanes = the four coordinates of a point)",
                "avx2": "AVX2 (vpmuludq, radix 2^25.5 with one limb per vector, 4 x 64-bit lanes = the four coordinates of a point)",
                "avx2p": "AVX2 (vpmuludq, radix 2^25.5 in dalek's packed FieldElement2625x4 layout: five vectors of eight 32-bit limbs hold the four coordinates of a point)"}
    simd = {"value": None, "isa": None, "note": "this host CPU (or the oracle's build) has neither AVX-512 IFMA nor AVX2: not measured"}

    def simd_run(isa, m):
        assert C.set_simd(isa)
        try:
            t0 = time.perf_counter()
            equal = True
            for j in range(m):
                ec, er, ek, _ = C.prove(cst, label, p.secrets[j], np.stack([p.common[com_rank[i]] if i in com_rank else p.inst[inst_rank[i], j] for i in range(len(points))]), ent[j].tobytes())
                equal = equal and bool((er == resp[j]).all()) and bool((ek == coms[j]).all())
            t1 = time.perf_counter()
            rc = C.batch_verify(cst, label, m, np.ascontiguousarray(p.inst[:, :m]), p.common, np.ascontiguousarray(coms[:m]), np.ascontiguousarray(resp[:m]), np.ascontiguousarray(w[:, :m]))
            t2 = time.perf_counter()
        finally:
            C.set_simd(False)
        assert rc == 0 and equal, "the vector backend (%s) of the CPU baseline disagrees with the scalar port" % isa
        return {"value": m / (t2 - t0), "unit": "proofs/s", "cores": 1, "isa": ISA_TEXT[isa], "proofs": m, "prove_proofs_per_s": m / (t1 - t0), "batch_verifies_per_s": m / (t2 - t1),
                "equal_to_scalar_port": True}

    isas = C.simd_isas()
    if isas and len(ps) == 1 and "prove" in ps[0].flows:
        p = ps[0]
        m = min(p.n, sample)
        simd = simd_run(isas[0], m)
        simd["note"] = ("MSM inner loops (point addition / doubling in the 4-way parallel formulas, constant-time and NAF Straus, Pippenger) vectorised as in "
                        "curve25519-dalek's simd_backend; decompression, compression, Merlin and scalar arithmetic stay scalar, as there")
        if "avx2p" in isas and isas[0] != "avx2p":
            simd["avx2"] = simd_run("avx2p", min(m, 512))
            simd["avx2"]["note"] = ("the same backend on AVX2 for hosts without IFMA, in curve25519-dalek's packed layout (FieldElement2625x4: four field elements in five vectors of "
                                    "eight 32-bit limbs, vpmuludq on operands unpacked per product -- backend/vector/avx2/field.rs), timed on the first %d proofs of the sample.  "
                                    "On a core with two 64-bit multipliers per cycle (Zen 5) a 32 x 32 -> 64 vector multiplier does not beat the scalar port -- reported, never the baseline"
                                    % min(m, 512))
            simd["avx2"]["one_limb_per_vector"] = simd_run("avx2", min(m, 256))["value"]      # round 4's form (ten vectors per four elements): what the packing buys
    outd = {"value": n_done / (t_prove + t_bv), "unit": "proofs/s", "cores": 1, "kind": "port", "simd": simd,
            "sample": "; ".join(notes) + "; Merlin + radix-16 constant-time Straus + responses / Merlin + coefficients + Pippenger; gcc -O3 -march=native, 5x51-bit limbs",
            "batch_verifies_per_s": n_done / t_bv,
            "note": "scalar u64-style port of the reference's flows (dalek's u64_backend algorithms); the reference's own Rust cannot be built on this box; "
                    "`simd` = the same port with the MSM inner loops on AVX-512 IFMA (else AVX2) vectors in the design of dalek's simd_backend (Cargo.toml:37-40), measured here"}
    if t_prove:
        outd["prove_proofs_per_s"] = n_done / t_prove
    return outd, proofs


def cpu_baseline_all_cores():
    """The same port on every CPU the box lets this job use: oracle/cpu_bench.py in a clean
    subprocess, one worker process per usable hardware thread (affinity mask and cgroup quota respected), each proving and
    batch-verifying 1024 CMZ presentations (about 1 s of work per worker)."""
    try:
        outp = subprocess.run([sys.executable, "-m", "oracle.cpu_bench", "--per", "1024"], cwd=ROOT, capture_output=True, text=True, timeout=240)
        j = json.loads(outp.stdout.strip().splitlines()[-1])
        j["note"] = "same scalar port on every usable host thread"
        outs = subprocess.run([sys.executable, "-m", "oracle.cpu_bench", "--per", "1024", "--simd"], cwd=ROOT, capture_output=True, text=True, timeout=240)
        js = json.loads(outs.stdout.strip().splitlines()[-1])
        j["simd"] = {"value": js["value"] if js.get("isa") != "scalar u64" else None, "isa": js.get("isa"), "cores": js["cores"],
                     "note": "the vector backend (oracle/c/simd_ifma.c, best instruction set of the host) on every usable host thread"}
        return j
    except Exception as e:      # a reported baseline, never the measurement: do not fail the bench line over it
        return {"value": None, "unit": "proofs/s", "cores": os.cpu_count(), "kind": "port", "sample": "failed: %r" % (e,)}


if __name__ == "__main__":
    main()
