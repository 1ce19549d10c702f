"""Interop fixtures (tests/golden/interop/) on the MI355X path:
  * the product (zkp_amd.toolbox over the HIP library, proofs serialised by the C codec) reproduces the committed from_repo/*.bin
    byte for byte -- these are the files the Rust crate verifies in rust/interop/tests/verify_repo_proofs.rs;
  * the product verifies everything in from_repo/ and, once somebody ran rust/interop's emit_crate_proofs with cargo and
    committed the output, everything the REAL crate wrote to from_crate/ : verify_compact (the recomputed challenge must equal
    the proof's), verify_batchable, and batch_verify of the proofs that form a batch -- through both routes (host transcripts /
    fused on the device)."""
import numpy as np
import pytest

from zkp_amd import toolbox as T
from tests.test_oracle_interop import G, _load

pytestmark = pytest.mark.gpu
NEVER = 0xFFFFFFFF


@pytest.fixture(scope="module")
def eng():
    from zkp_amd.engine import Engine
    e = Engine(0)
    yield e
    e.close()
    T.set_fused_min_batch(32)


def test_gpu_path_reproduces_the_committed_fixtures(eng):
    man, files = _load("from_repo")
    assert man is not None
    got, _ = G.produce("gpu", eng)
    assert sorted(got) == sorted(files)
    for fn in files:
        assert got[fn] == files[fn], fn


def _verify(eng, e, data):
    secrets, cons = G.statement_of(e)
    pts = {p["name"]: bytes.fromhex(p["hex"]) for p in e["points"]}
    label = e["transcript_label"].encode()
    proof = (T.CompactProof if e["kind"] == "compact" else T.BatchableProof).from_bytes(data)
    if e["api"] == "constraint_api":
        v = T.Verifier(e["proof_label"].encode(), T.Transcript(label), eng)
        sv = {n: v.allocate_scalar(n.encode()) for n in secrets}
        pv = {p["name"]: v.allocate_point(p["name"].encode(), pts[p["name"]]) for p in e["points"]}
        for lhs, lc in cons:
            v.constrain(pv[lhs], [(sv[a], pv[b]) for a, b in lc])
        return v.verify_compact(proof) if e["kind"] == "compact" else v.verify_batchable(proof)
    mod = T.define_proof(e["statement"], e["proof_label"].encode(), secrets, [p["name"] for p in e["points"] if not p["common"]],
                         [p["name"] for p in e["points"] if p["common"]], cons)
    return (mod.verify_compact if e["kind"] == "compact" else mod.verify_batchable)(eng, proof, T.Transcript(label), pts)


@pytest.mark.parametrize("which", ["from_repo", "from_crate"])
@pytest.mark.parametrize("route", ["host", "fused"])
def test_product_verifies_interop_proofs(eng, which, route):
    man, files = _load(which)
    if man is None:
        pytest.skip("tests/golden/interop/%s is empty (from_crate: run rust/interop's emit_crate_proofs with cargo and commit the files)" % which)
    T.set_fused_min_batch(NEVER if route == "host" else 0)
    try:
        batches = {}
        for e in man["proofs"]:
            _verify(eng, e, files[e["file"]])                    # raises VerificationFailure if the recomputed challenge / the MSM disagrees
            bad = bytearray(files[e["file"]])
            bad[-32] ^= 1
            with pytest.raises((T.VerificationFailure, ValueError)):
                _verify(eng, e, bytes(bad))
            if e.get("batch"):
                batches.setdefault(e["batch"], []).append(e)
        for name, es in batches.items():
            e0 = es[0]
            secrets, cons = G.statement_of(e0)
            inst_n = [p["name"] for p in e0["points"] if not p["common"]]
            com_n = [p["name"] for p in e0["points"] if p["common"]]
            mod = T.define_proof(e0["statement"], e0["proof_label"].encode(), secrets, inst_n, com_n, cons)
            proofs = [T.BatchableProof.from_bytes(files[e["file"]]) for e in es]
            enc = [{p["name"]: bytes.fromhex(p["hex"]) for p in e["points"]} for e in es]
            label = e0["transcript_label"].encode()
            mod.batch_verify(eng, proofs, [T.Transcript(label) for _ in es], {n: [x[n] for x in enc] for n in inst_n}, {n: enc[0][n] for n in com_n})
            proofs[1].responses[0] = bytes([proofs[1].responses[0][0] ^ 1]) + proofs[1].responses[0][1:]
            with pytest.raises(T.VerificationFailure):
                mod.batch_verify(eng, proofs, [T.Transcript(label) for _ in es], {n: [x[n] for x in enc] for n in inst_n}, {n: enc[0][n] for n in com_n})
    finally:
        T.set_fused_min_batch(32)
