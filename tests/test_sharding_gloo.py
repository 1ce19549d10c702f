"""world_size-2 tests of the multi-GPU path on CPU (gloo): contiguous proof ranges per rank, local batch checks, AND of the verdict
bits.  Two local checks: the oracle (test_sharded_batch_verify_world2: the sharding logic against the checker) and -- the row (e)
evidence -- the PRODUCT: zkp_amd.toolbox.batch_verify on the host backend (ctx == NULL, csrc/host/host_backend.cpp; on the GPU box the
same call is bound to an Engine), over zkp_amd.sharding's ranges and over the ranges zkp_pipe_batch_verify itself cuts
(zkp_pipe_shard_plan = pipe.cpp's make_shards), with a tampered proof in each range in turn."""
import os
import random
import socket
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, tamper, out):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from oracle import cbind as C
    from oracle import model as M
    from zkp_amd import sharding
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    rng = random.Random(5)                       # same data on every rank
    mst = M.dleq_statement()
    cst = C.Statement.from_model(mst)
    n, label = 7, b"DLEQBatchTest"
    G = M.ristretto_encode(M.BASEPOINT)
    inst_rows = {"A": [], "B": [], "H": []}
    coms, resps = [], []
    for j in range(n):
        x = rng.randrange(1, M.L)
        H = M.ristretto_from_uniform_bytes(bytes(rng.randrange(256) for _ in range(64)))
        enc = {"A": M.ristretto_encode(M.pt_mul(x, M.BASEPOINT)), "B": M.ristretto_encode(M.pt_mul(x, H)), "H": M.ristretto_encode(H), "G": G}
        _, r, k, _ = C.prove(cst, label, np.frombuffer(x.to_bytes(32, "little"), np.uint8).reshape(1, 32),
                             np.frombuffer(b"".join(enc[p] for p in cst.points), np.uint8).reshape(-1, 32), bytes([j]) * 32)
        for p in inst_rows:
            inst_rows[p].append(enc[p])
        coms.append(k); resps.append(r)
    inst = np.frombuffer(b"".join(e for p in ("A", "B", "H") for e in inst_rows[p]), np.uint8).reshape(3, n, 32)
    coms, resps = np.stack(coms), np.stack(resps)
    if tamper is not None:
        resps = resps.copy()
        resps[tamper, 0, 3] ^= 4
    common = np.frombuffer(G, np.uint8).reshape(1, 32)

    def local_check(ts, inst_l, coms_l, resps_l):
        m = len(coms_l)
        w16 = np.frombuffer(os.urandom(16 * 2 * m), np.uint8).reshape(2, m, 16)
        return C.batch_verify(cst, label, m, inst_l, common, coms_l, resps_l, w16) == 0

    lo, hi = sharding.shard_range(n, rank, world)
    ok = sharding.batch_verify_sharded(local_check, inst, coms, resps, np.zeros((n, 208), np.uint8), rank, world)
    out.put((rank, lo, hi, ok))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("tamper,expect", [(None, True), (1, False), (6, False)])
def test_sharded_batch_verify_world2(tamper, expect):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, tamper, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert [(r[1], r[2]) for r in res] == [(0, 3), (3, 7)]       # contiguous ranges covering all 7 proofs
    assert all(r[3] == expect for r in res)                       # every rank sees the AND of both verdicts


def test_shard_ranges_cover_exactly():
    from zkp_amd.sharding import shard_range
    for n in (0, 1, 7, 4096, 2**22):
        for world in (1, 2, 3, 8):
            rs = [shard_range(n, r, world) for r in range(world)]
            assert rs[0][0] == 0 and rs[-1][1] == n
            assert all(rs[i][1] == rs[i + 1][0] for i in range(world - 1))
            assert max(h - l for l, h in rs) - min(h - l for l, h in rs) <= 1


# ---- the product as the local check ------------------------------------------------------------------------------------------
def _product_worker(rank, world, port, tamper, plan, out):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from zkp_amd import sharding
    from zkp_amd import toolbox as T
    from tests.test_gpu_toolbox import _cmz_batch
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    n, label = 9, b"Benchmark"
    mod, secrets, inst, common = _cmz_batch(n, 23)               # same data on every rank (seeded)
    st, host = mod.statement, T.HostEngine()
    entropy = np.random.default_rng(3).integers(0, 256, size=(n, 32), dtype=np.uint8)
    t0 = lambda k: np.stack([T.Transcript(label).state] * k)      # noqa: E731
    _, resp, coms = T.prove_batch(host, st, t0(n), secrets, inst, common, entropy)      # the product's prover, host backend
    if tamper is not None:
        resp = resp.copy()
        resp[tamper, 5, 1] ^= 0x10

    def local_check(ts, inst_l, coms_l, resps_l):                 # BatchVerifier::verify_batchable over the rank's range, own OS-random weights
        try:
            T.batch_verify(host, st, ts, inst_l, common, coms_l, resps_l)
            return True
        except T.VerificationFailure:
            return False

    if plan == "sharding":
        lo, hi = sharding.shard_range(n, rank, world)
        ok = sharding.batch_verify_sharded(local_check, inst, coms, resp, t0(n), rank, world)
    else:
        # the ranges zkp_pipe_batch_verify gives the contexts of a pipe over `world` devices (pipe.cpp: make_shards -> slot g = device g), this rank
        # playing device `rank`; fused_min_batch = 1 so that nine proofs do spread over two contexts as 2^22 do over eight
        ranges = T.pipe_shard_plan(n, world, 1, fused_min_batch=1)
        assert ranges == [sharding.shard_range(n, r, world) for r in range(world)]
        lo, hi = ranges[rank]
        ok = local_check(t0(hi - lo), np.ascontiguousarray(inst[:, lo:hi]), coms[lo:hi], resp[lo:hi])
        ok = sharding.and_reduce(ok)
    out.put((rank, lo, hi, ok))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("plan", ["sharding", "pipe"])
@pytest.mark.parametrize("tamper,expect", [(None, True), (2, False), (7, False)])      # a bad proof in rank 0's range, in rank 1's range
def test_product_host_backend_sharded_batch_verify_world2(plan, tamper, expect):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_product_worker, args=(r, 2, port, tamper, plan, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=180) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert [(r[1], r[2]) for r in res] == [(0, 4), (4, 9)]
    assert all(r[3] == expect for r in res)


def test_pipe_shard_plan_arithmetic():
    """pipe.cpp's partition (zkp_pipe_shard_plan): contiguous, covering, balanced, never below the fused threshold, whole batches for the K-batch call;
    range g on slot g, and zkp_pipe_create's slot order (slot = k * n_devices + d) puts G <= #contexts ranges on the GPUs round robin."""
    from zkp_amd import toolbox as T
    from zkp_amd.sharding import shard_range
    for n in (1, 31, 32, 100, 4096, 2**22):
        for ctxs in (1, 3, 8, 24):
            for fmin in (1, 32):
                rs = T.pipe_shard_plan(n, ctxs, 1, fused_min_batch=fmin)
                G = len(rs)
                assert 1 <= G <= ctxs and rs[0][0] == 0 and rs[-1][1] == n
                assert all(rs[i][1] == rs[i + 1][0] for i in range(G - 1))
                assert rs == [shard_range(n, g, G) for g in range(G)]
                assert G == max(1, min(ctxs, n, n // fmin))
                if n >= fmin:
                    assert min(h - l for l, h in rs) >= fmin
    # whole batches: 50 batches of 4096 proofs over 8 contexts
    rs = T.pipe_shard_plan(50, 8, 4096, fused_min_batch=32)
    assert len(rs) == 8 and rs[0] == (0, 6) and rs[-1][1] == 50
    # 8 GPUs x 3 contexts, 15 ranges: devices of slots 0 .. 14 = slot % 8 -> every GPU has one range before any has two
    devs = [g % 8 for g in range(15)]
    assert sorted(set(devs)) == list(range(8)) and max(devs.count(d) for d in range(8)) == 2
    assert T.pipe_shard_plan(0, 8) == []
