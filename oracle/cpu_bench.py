"""All-core CPU baseline -- test infrastructure like everything under oracle/: the C
restatement of the reference's flows timed on every host core, one worker PROCESS per hardware thread (no GIL, no GPU
runtime in these processes), each proving and then batch-verifying its own CMZ'13 presentations.  Called by bench.py's
cpu_baseline leg as a subprocess; prints one JSON line.

    python -m oracle.cpu_bench [--workers W] [--per P]"""
import argparse
import json
import multiprocessing as mp
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import cbind as C          # noqa: E402
from oracle import model as M          # noqa: E402

LABEL = b"Benchmark"
BASEPOINT = bytes.fromhex("e2f2ae0a6abc4e71a884a961c500515f58e30b6aa582dd8db6a65945e08d2d76")


def cmz_instance(n, seed):
    """n consistent presentations of cred_show_10 (benches/zkp.rs:27-46), made with the oracle's own arithmetic."""
    rng = np.random.default_rng(seed)

    def rs(k):
        s = rng.integers(0, 256, size=(k, 32), dtype=np.uint8)
        s[:, 31] &= 0x0f
        return s

    base = np.frombuffer(BASEPOINT, np.uint8).reshape(1, 32)
    k = 12 + 2 * n
    pts, _ = C.msm_many(np.arange(k + 1, dtype=np.uint32), rs(k), np.zeros(k, np.uint32), base, 0)
    common, P, Q = pts[:12], pts[12:12 + n], pts[12 + n:]
    secrets = rs(n * 21).reshape(n, 21, 32)
    table = np.concatenate([common, P, Q])
    off, scal, pidx = [0], [], []
    for j in range(n):
        for i in range(10):
            scal += [secrets[j, i], secrets[j, 10 + i]]
            pidx += [12 + j, 10]
            off.append(len(pidx))
        for i in range(10):
            scal.append(secrets[j, i]); pidx.append(i)
        scal.append(secrets[j, 20]); pidx.append(12 + n + j)
        off.append(len(pidx))
    cv, st = C.msm_many(np.array(off, np.uint32), np.stack(scal), np.array(pidx, np.uint32), table, 0)
    assert not st.any()
    cv = cv.reshape(n, 11, 32)
    inst = np.ascontiguousarray(np.concatenate([cv[:, :10].transpose(1, 0, 2), P[None], Q[None], cv[:, 10][None]]))
    return secrets, inst, np.ascontiguousarray(common), rng.integers(0, 256, size=(n, 32), dtype=np.uint8), \
        rng.integers(0, 256, size=(11, n, 16), dtype=np.uint8)


def _work(args):
    simd, per, data = args
    secrets, inst, common, entropy, weights = data
    C.set_simd(simd or False)                       # False, True (best instruction set) or "avx2"
    cst = C.Statement.from_model(M.cmz_statement(10))
    coms = np.zeros((per, 11, 32), np.uint8)
    resp = np.zeros((per, 21, 32), np.uint8)
    t0 = time.perf_counter()
    for j in range(per):
        _, er, ek, _ = C.prove(cst, LABEL, secrets[j], np.concatenate([inst[:, j], common]), entropy[j].tobytes())
        coms[j], resp[j] = ek, er
    rc = C.batch_verify(cst, LABEL, per, inst, common, coms, resp, weights)
    return rc, time.perf_counter() - t0


def usable_cpus():
    """Hardware threads this process may actually use: affinity mask, capped by the cgroup CPU quota (cpu.max)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period) + 0.5)))
    except Exception:
        pass
    return n


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workers", type=int, default=usable_cpus())
    ap.add_argument("--per", type=int, default=96)
    ap.add_argument("--simd", action="store_true", help="MSM inner loops on the vector backend (oracle/c/simd_ifma.c): AVX-512 IFMA where the CPU has it, else AVX2")
    ap.add_argument("--simd-isa", choices=["avx512ifma", "avx2p", "avx2"], default=None, help="the vector backend on this instruction set (implies --simd)")
    a = ap.parse_args()
    C.build()
    data = cmz_instance(a.per, 7)            # every worker handles an identical range: same work, no data skew
    warm = cmz_instance(2, 8)
    ctx = mp.get_context("fork")
    with ctx.Pool(a.workers) as pool:
        simd = False
        if a.simd_isa:
            simd = a.simd_isa if a.simd_isa in C.simd_isas() else False
        elif a.simd and C.simd_available():
            simd = C.simd_isas()[0]
        pool.map(_work, [(simd, 2, warm)] * a.workers)       # start the workers, load the library
        t0 = time.perf_counter()
        res = pool.map(_work, [(simd, a.per, data)] * a.workers)
        wall = time.perf_counter() - t0
    assert not any(rc for rc, _ in res), "a sample batch did not verify"
    print(json.dumps({"value": a.workers * a.per / wall, "unit": "proofs/s", "cores": a.workers, "kind": "port", "isa": simd or "scalar u64",
                      "sample": "%d worker processes (usable CPUs: affinity %d, cgroup quota applied; %d visible) x %d proofs, each proven one by one "
                                "and batch-verified by its worker; %.2f s wall, slowest worker %.2f s; gcc -O3 -march=native, 5x51-bit limbs"
                                % (a.workers, len(os.sched_getaffinity(0)), os.cpu_count() or 0, a.per, wall, max(t for _, t in res))}))


if __name__ == "__main__":
    main()
