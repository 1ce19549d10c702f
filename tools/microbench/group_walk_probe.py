"""The CMZ prover's MSM job (4096 proofs, constant time) with the grouped comb walk through LDS (ZKP_OPT_GROUPED_COMB = 1) and with
the masked scans (0), three calls each -- to be run under rocprofv3 (--kernel-trace --stats, or --pmc ...)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import bench
from zkp_amd.engine import Engine, ZKP_CT

eng = Engine(0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
rng = np.random.default_rng(5)
off, pidx, n_pts = bench.cmz_shape(n)
base = np.frombuffer(bytes.fromhex("e2f2ae0a6abc4e71a884a961c500515f58e30b6aa582dd8db6a65945e08d2d76"), np.uint8).reshape(1, 32)
ks = rng.integers(0, 256, size=(n_pts, 32), dtype=np.uint8)
ks[:, 31] &= 0x0f
pts, st = eng.msm_many(np.arange(n_pts + 1, dtype=np.uint32), ks, np.zeros(n_pts, np.uint32), base, ZKP_CT)
eng.prepare_fixed_points(pts[:11])
sc = rng.integers(0, 256, size=(31 * n, 32), dtype=np.uint8)
sc[:, 31] &= 0x0f
ref = None
if os.environ.get('PROBE_OPT3'):
    eng.set_option(3, int(os.environ['PROBE_OPT3']))
for grouped in (1, 0, 1, 0, 1, 0):
    eng.set_option(6, grouped)
    out, st = eng.msm_many(off, sc, pidx, pts, ZKP_CT)
    if not os.environ.get('PROBE_NOCHECK'):
        assert not st.any()
        ref = out if ref is None else ref
        assert (out == ref).all()
eng.close()
print("ok")
