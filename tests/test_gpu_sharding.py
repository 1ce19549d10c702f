"""The multi-GPU path with the REAL engine: two ranks (both on GPU 0 of the test box, gloo for the one exchange), each
proving and batch-verifying its contiguous proof range through the fused device flows, verdicts AND-ed
(zkp_amd/sharding.py).  On an 8-GPU node the same code runs one rank per GPU over RCCL (bench.py --gpus N)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
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
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    import bench
    from zkp_amd import sharding
    from zkp_amd import toolbox as T
    from zkp_amd.engine import Engine
    eng = Engine(0)
    n = 600
    rng = np.random.default_rng(21)                       # the same instance on every rank; each rank works on its range
    secrets, inst, common = bench.cmz_instance(eng, n, rng)
    mod = T.cmz_module(10)
    st = mod.statement
    label = b"sharded"
    entropy = rng.integers(0, 256, size=(n, 32), dtype=np.uint8)
    lo, hi = sharding.shard_range(n, rank, world)
    t0 = T.Transcript(label).state
    ts = np.repeat(t0[None], hi - lo, axis=0)
    chal, resp, coms = T.prove_batch(eng, st, ts, secrets[lo:hi], np.ascontiguousarray(inst[:, lo:hi]), common, entropy[lo:hi])
    # all-gather of the proofs is not part of the path (each verifier rank checks the range it is handed): rebuild the
    # full arrays locally by proving the other range too, so that both ranks hold identical inputs for the sharded check
    ts_all = np.repeat(t0[None], n, axis=0)
    chal_all, resp_all, coms_all = T.prove_batch(eng, st, ts_all, secrets, inst, common, entropy)
    assert (chal_all[lo:hi] == chal).all() and (resp_all[lo:hi] == resp).all() and (coms_all[lo:hi] == coms).all()
    if tamper is not None:
        resp_all = resp_all.copy()
        resp_all[tamper, 5, 2] ^= 8

    def local_check(ts_l, inst_l, coms_l, resp_l):
        try:
            T.batch_verify(eng, st, ts_l, inst_l, common, coms_l, resp_l)
            return True
        except T.VerificationFailure:
            return False

    ok = sharding.batch_verify_sharded(local_check, inst, coms_all, resp_all, np.repeat(t0[None], n, axis=0), rank, world)
    out.put((rank, lo, hi, ok))
    dist.barrier()
    dist.destroy_process_group()
    eng.close()


@pytest.mark.parametrize("tamper,expect", [(None, True), (17, False), (599, False)])
def test_sharded_prove_and_batch_verify_two_ranks_on_the_gpu(tamper, expect):
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, tamper, out)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(out.get(timeout=300) for _ in procs)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert [r[1:3] for r in res] == [(0, 300), (300, 600)]
    assert all(r[3] == expect for r in res)           # the AND reaches every rank


def test_bench_eight_rank_control_flow_on_one_gpu():
    """bench.py --gpus 8 exactly as the driver launches it (torch.distributed.run, one process per rank), with all eight
    ranks on this box's single GPU (ZKP_BENCH_DRYRUN_ONE_GPU): barriers, the per-rank stream pools and HIP graphs, the
    verdict reduction through the fallback branch, the MAX of the elapsed times and rank 0's JSON line.  Not a measurement."""
    import json
    import subprocess
    env = dict(os.environ, ZKP_BENCH_DRYRUN_ONE_GPU="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("GPU_MAX_HW_QUEUES", None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=8", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "4", "--warmup", "1",
           "--batch", "256", "--batches-per-call", "2", "--streams", "2", "--no-flow-lines", "--multi-total4", "2048", "--multi-total5", "1024"]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]                       # ONE line, from rank 0
    j = json.loads(lines[0])
    assert j["n_gpus"] == 8 and j["steps"] == 4 and j["value"] > 0 and j["scaling"] == "weak"
    assert j["config"]["backend_world_size"] == 8 and j["config"]["collective"] == "gloo-fallback" and "dry run" in j["config"]["rccl_error"]
    assert abs(j["value"] - 8 * 256 * 4 / (j["ms_per_step"] * 4e-3)) < 1e-6 * j["value"]
    assert "cpu_baseline" not in j                                  # reported at N = 1 only
    assert j["config"]["batches_per_call"] == 2 and j["config"]["calls"] == 2          # 4 steps = 2 calls of 2 batches each
    # BASELINE configs[3] / configs[4] (the 8-GPU configurations) ride along as strong-scaling sub-records of the same line
    c4, c5 = j["configs"]["4"], j["configs"]["5"]
    assert c4["proofs_total"] == 2048 and c4["proofs_per_gpu"] == 256 and c4["scaling"] == "strong" and c4["value"] > 0
    assert c5["proofs_total"] == 1024 and c5["proofs_per_gpu"] == 128 and c5["value"] > 0
    assert c4["per_rank_elapsed_ms"]["min"] <= c4["per_rank_elapsed_ms"]["max"] <= c4["elapsed_ms"] * 1.001 + 1e-6
    assert c4["collective"] == "gloo-fallback"


def test_bench_in_process_multi_gpu_control_flow_on_one_gpu():
    """bench.py --gpus 3 --in-process: ONE process, one host thread per rank, no torchrun / gloo / RCCL (all three ranks on this box's
    single GPU): thread barriers, host AND of the verdicts, MAX of the elapsed times, the configs[3] / configs[4] sub-records and the
    zkp_pipe line over the same devices.  Not a measurement."""
    import json
    import subprocess
    env = dict(os.environ, ZKP_BENCH_DRYRUN_ONE_GPU="1")
    env.pop("GPU_MAX_HW_QUEUES", None)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "3", "--in-process", "--steps", "4", "--warmup", "1", "--batch", "256", "--batches-per-call", "2",
           "--streams", "2", "--multi-total4", "1536", "--multi-total5", "768", "--no-cpu-baseline"]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    j = json.loads(lines[0])
    assert j["n_gpus"] == 3 and j["steps"] == 4 and j["value"] > 0 and j["scaling"] == "weak"
    assert j["config"]["backend_world_size"] == 3 and j["config"]["collective"].startswith("host-and")
    assert abs(j["value"] - 3 * 256 * 4 / (j["ms_per_step"] * 4e-3)) < 1e-6 * j["value"]
    c4, c5 = j["configs"]["4"], j["configs"]["5"]
    assert c4["proofs_total"] == 1536 and c4["proofs_per_gpu"] == 512 and c5["proofs_total"] == 768 and c4["value"] > 0 and c5["value"] > 0
    pl = j["e2e_host_buffers"]["pipelined"]
    assert pl["devices"] == [0, 0, 0] and pl["jobs_in_flight"] == 9 and pl["proofs_per_s"] > 0
