#!/usr/bin/env python3
"""The reference's DLEQ example (README.md / tests/zkp.rs) for a BATCH of proofs on an MI355X:

    define_proof! {dleq, "DLEQ Example Proof", (x), (A, B, H), (G) : A = (x * G), B = (x * H) }

Everything that is arithmetic -- transcripts included for batches of 32 proofs or more -- runs on the GPU behind the C ABI
of include/zkp_mi355x.h; this script only prepares inputs and checks verdicts.  Needs a gfx950 device.

    python examples/dleq_batch.py [N]"""
import hashlib
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from zkp_amd import toolbox as T
from zkp_amd.engine import Engine, ZKP_CT

BASEPOINT = bytes.fromhex("e2f2ae0a6abc4e71a884a961c500515f58e30b6aa582dd8db6a65945e08d2d76")


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
    eng = Engine(0)
    dleq = T.define_proof("dleq", b"DLEQ Example Proof", ["x"], ["A", "B", "H"], ["G"], [("A", [("x", "G")]), ("B", [("x", "H")])])
    st = dleq.statement
    rng = np.random.default_rng(int.from_bytes(hashlib.sha256(b"example").digest()[:8], "little"))
    G = np.frombuffer(BASEPOINT, np.uint8).reshape(1, 32).copy()

    def scalars(k):
        s = rng.integers(0, 256, size=(k, 32), dtype=np.uint8)
        s[:, 31] &= 0x0f
        return s

    # per-proof H, secret x, and the public A = x G, B = x H (computed with the engine's own multiscalar entry point)
    iota = np.arange(n + 1, dtype=np.uint32)
    H, _ = eng.msm_many(iota, scalars(n), np.zeros(n, np.uint32), G, ZKP_CT)
    x = scalars(n)
    A, _ = eng.msm_many(iota, x, np.zeros(n, np.uint32), G, ZKP_CT)
    B, _ = eng.msm_many(iota, x, np.arange(n, dtype=np.uint32), H, ZKP_CT)
    inst = np.ascontiguousarray(np.stack([A, B, H]))                       # instance points, row per variable
    secrets = np.ascontiguousarray(x.reshape(n, 1, 32))

    transcripts = np.stack([T.Transcript(b"DLEQTest").state] * n)           # Transcript::new(b"DLEQTest") per proof
    chal, resp, coms = T.prove_batch(eng, st, transcripts, secrets, inst, G)  # entropy from the OS (ChaCha20 stream)
    print("proved %d statements: %d-byte compact proofs, %d-byte batchable proofs" % (n, 32 + 32 * st.m, 32 * (st.nc + st.m)))

    transcripts = np.stack([T.Transcript(b"DLEQTest").state] * n)
    verdicts = T.verify_compact_batch(eng, st, transcripts, inst, G, chal, resp)
    print("verify_compact: %d of %d accepted" % (int((verdicts == 0).sum()), n))

    transcripts = np.stack([T.Transcript(b"DLEQTest").state] * n)
    T.batch_verify(eng, st, transcripts, inst, G, coms, resp)               # raises VerificationFailure otherwise
    print("batch verification of all %d proofs: ok" % n)

    resp[n // 2, 0, 0] ^= 1                                                 # corrupt one proof
    transcripts = np.stack([T.Transcript(b"DLEQTest").state] * n)
    try:
        T.batch_verify(eng, st, transcripts, inst, G, coms, resp)
        print("ERROR: a corrupted proof passed")
    except T.VerificationFailure:
        transcripts = np.stack([T.Transcript(b"DLEQTest").state] * n)
        each = T.verify_batchable_each(eng, st, transcripts, inst, G, coms, resp)
        print("batch with one corrupted proof rejected; per-proof check points at proof", int(np.nonzero(each)[0][0]))
    # many batches in one pass (zkp_batch_verify_many): the same proofs as 4 consecutive batches, each with a verdict of its own -- what a
    # service that would otherwise call BatchVerifier::verify_batchable four times hands to the GPU in one call
    if n % 4 == 0:
        transcripts = np.stack([T.Transcript(b"DLEQTest").state] * n)
        verdicts = T.batch_verify_many(eng, st, 4, transcripts, inst, G, coms, resp)
        print("4 batches of %d proofs in one pass: verdicts %s (0 = ok; the corrupted proof %d sits in batch %d)" % (n // 4, verdicts.tolist(), n // 2, (n // 2) // (n // 4)))
    eng.close()


if __name__ == "__main__":
    main()
