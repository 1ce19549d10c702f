import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from zkp_amd import toolbox as T
from zkp_amd.engine import Engine
from tests.test_gpu_toolbox import _cmz_batch

def t0(n=None):
    s = T.Transcript(b"pipe").state
    return s if n is None else np.stack([s] * n)

mod, secrets, inst, common = _cmz_batch(1024, 31)
st = mod.statement
entropy = np.random.default_rng(7).integers(0, 256, size=(1024, 32), dtype=np.uint8)
n, K = 128, 4
nn = n * K
eng = Engine(0)
c_s, r_s, k_s = T.prove_batch(eng, st, t0(nn), secrets[:nn], np.ascontiguousarray(inst[:, :nn]), common, entropy[:nn])
eng.close()
def diff(tag, out):
    for name, a, b in zip(("chal", "resp", "coms"), out, (c_s, r_s, k_s)):
        bad = np.argwhere((a != b).reshape(len(a), -1).any(axis=1)).ravel()
        if len(bad):
            print(tag, name, "differs in", len(bad), "proofs, first", bad[:8], "last", bad[-3:], "zero rows:", int((a.reshape(len(a), -1) == 0).all(axis=1).sum()))
            return False
    return True
for pinned in (False, True):
    mk = T.pinned_copy if pinned else np.ascontiguousarray
    a_sec, a_inst, a_com, a_ent = mk(secrets[:nn]), mk(inst[:, :nn]), mk(common), mk(entropy[:nn])
    for devs, ctxs, thr in (((0,), 2, -1), ((0, 0, 0), 2, 0), ((0, 0, 0), 2, 1), ((0, 0), 1, 1)):
        with T.Pipe(devs, ctxs) as pipe:
            if thr >= 0:
                pipe.set_submit_threads(thr)
            for rnd in range(3):
                jobs = [pipe.submit_prove(st, nn, t0(), a_sec, a_inst, a_com, a_ent) for _ in range(len(devs) * ctxs)]
                ok = [diff("pinned=%s devs=%d ctxs=%d threads=%d round %d job %d (ctx %d)" % (pinned, len(devs), ctxs, thr, rnd, i, j.context), j.wait()) for i, j in enumerate(jobs)]
                print("pinned=%s devs=%d ctxs=%d threads=%d round %d:" % (pinned, len(devs), ctxs, thr, rnd), "all equal" if all(ok) else "MISMATCH %s" % ok, flush=True)
