#!/usr/bin/env python3
"""What do the copies of a SYNCHRONOUS toolbox call cost when the caller's memory is ordinary (pageable) instead of pinned?  zkp_prove_batch + zkp_batch_verify,
same inputs in both kinds of memory, outputs preallocated and touched; best of 5.  (GPU box)"""
import ctypes, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import torch  # noqa: F401
torch.cuda.init()
import bench
from zkp_amd import toolbox as T
from zkp_amd.engine import Engine
L = T.lib()
_p = T._p
mod = T.cmz_module(10)
st = mod.statement
t0s = T.Transcript(bench.LABEL).state
for n in (4096, 20480, 40960):
    eng = Engine(0)
    secrets, inst, common = bench.make_instance(eng, bench.cmz_statement(), n, np.random.default_rng(78))
    for kind in ("pageable", "pinned", "pageable", "pinned"):
        mk = T.pinned_copy if kind == "pinned" else (lambda a: np.array(a, copy=True, order="C"))
        a_sec, a_inst, a_com, ts0 = mk(secrets), mk(inst), mk(common), mk(np.stack([t0s] * n))
        ts, chal, resp, coms = mk(ts0), mk(np.ones((n, 32), np.uint8)), mk(np.ones((n, st.m, 32), np.uint8)), mk(np.ones((n, st.nc, 32), np.uint8))
        best = [1e9, 1e9]
        for _ in range(6):
            ts[...] = ts0
            t0 = time.perf_counter()
            rc = L.zkp_prove_batch(eng._h, st._h, ctypes.c_uint32(n), _p(ts), _p(a_sec), _p(a_inst), _p(a_com), None, 0, _p(chal), _p(resp), _p(coms))
            t1 = time.perf_counter()
            assert rc == 0, rc
            ts[...] = ts0
            t2 = time.perf_counter()
            rc = L.zkp_batch_verify(eng._h, st._h, ctypes.c_uint32(n), ctypes.c_uint32(n), _p(ts), _p(a_inst), _p(a_com), _p(coms), _p(resp), None, 0)
            t3 = time.perf_counter()
            assert rc == 0, rc
            best = [min(best[0], t1 - t0), min(best[1], t3 - t2)]
        print("n = %6d  %-8s  prove %.3f ms  batch_verify %.3f ms  -> %.3f M proofs/s" % (n, kind, best[0] * 1e3, best[1] * 1e3, n / sum(best) / 1e6), flush=True)
    eng.close()
