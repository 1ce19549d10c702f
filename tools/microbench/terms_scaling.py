"""How does the generic term kernel (k_terms_r4) scale with the number of terms?  Flat time = per-wave latency bound."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from zkp_amd.engine import Engine, ZKP_CT
eng = Engine(0)
rng = np.random.default_rng(1)
base = np.frombuffer(bytes.fromhex("e2f2ae0a6abc4e71a884a961c500515f58e30b6aa582dd8db6a65945e08d2d76"), np.uint8).reshape(1, 32)
def rs(k):
    s = rng.integers(0, 256, size=(k, 32), dtype=np.uint8); s[:, 31] &= 0x0f; return s
pts, _ = eng.msm_many(np.arange(513, dtype=np.uint32), rs(512), np.zeros(512, np.uint32), base, ZKP_CT)
dev = torch.device("cuda", 0)
d_pts = torch.from_numpy(pts).to(dev)
eng.set_profiling(True)
for n in [64, 4096, 16384, 32768, 45056, 65536, 98304, 131072, 262144, 524288]:
    d_sc = torch.from_numpy(rs(n)).to(dev)
    d_pidx = torch.from_numpy(rng.integers(0, 512, size=n).astype(np.int32)).to(dev)
    d_off = torch.arange(n + 1, dtype=torch.int32, device=dev)
    d_out = torch.zeros((n, 32), dtype=torch.uint8, device=dev); d_st = torch.zeros(n, dtype=torch.uint8, device=dev)
    for _ in range(2):
        eng.msm_many_dev(n, d_off.data_ptr(), d_sc.data_ptr(), d_pidx.data_ptr(), d_pts.data_ptr(), 512, n, ZKP_CT, d_out.data_ptr(), d_st.data_ptr())
        km, tot = eng.last_timing()
    print("n_terms %7d  waves %6d  terms %8.3f ms  reduce %7.3f ms   -> %6.1f ns/term" % (n, n // 64, km["terms"], km["reduce"], km["terms"] * 1e6 / n))
