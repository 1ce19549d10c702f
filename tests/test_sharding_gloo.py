"""world_size-2 test of the multi-GPU path on CPU (gloo): contiguous proof ranges per rank, local batch checks,
AND of the verdict bits.  The local check is played by the oracle here (no GPU in this container); on the GPU
box the same function is driven by zkp_amd.toolbox.batch_verify."""
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
