"""Interop fixtures (tests/golden/interop/), CPU side -- no GPU needed:
  * from_repo/*.bin (committed; made by the HIP path on an MI355X) are exactly what the ORACLE's prover + an independent bincode
    writer produce for the same statements, secrets and entropy, and parse with the product's C wire codec;
  * the oracle's restatement of the reference verifiers accepts them (alone and as batches) and rejects them after a bit flip;
  * whatever the Rust crate wrote to from_crate/ (rust/interop/tests/emit_crate_proofs.rs) is accepted by the oracle too
    (skipped while that directory is empty).
The Rust half (rust/interop/tests/verify_repo_proofs.rs, UNBUILT here) makes the real crate verify from_repo/."""
import importlib.util
import json
import os

import numpy as np
import pytest

from oracle import cbind as C

HERE = os.path.dirname(os.path.abspath(__file__))
_spec = importlib.util.spec_from_file_location("make_from_repo", os.path.join(HERE, "golden", "interop", "make_from_repo.py"))
G = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(G)


def _load(which):
    d = G.FROM_REPO if which == "from_repo" else G.FROM_CRATE
    mp = os.path.join(d, "manifest.json")
    if not os.path.exists(mp):
        return None, {}
    man = json.load(open(mp))
    return man, {e["file"]: open(os.path.join(d, e["file"]), "rb").read() for e in man["proofs"]}


def parse(kind, data):
    """independent reader of the bincode layout -> (challenge | commitments, responses)"""
    import struct
    if kind == "compact":
        m = struct.unpack_from("<Q", data, 32)[0]
        assert len(data) == 40 + 32 * m
        return data[:32], [data[40 + 32 * i: 72 + 32 * i] for i in range(m)]
    n = struct.unpack_from("<Q", data, 0)[0]
    coms = [data[8 + 32 * i: 40 + 32 * i] for i in range(n)]
    m = struct.unpack_from("<Q", data, 8 + 32 * n)[0]
    o = 16 + 32 * n
    assert len(data) == o + 32 * m
    return coms, [data[o + 32 * i: o + 32 * i + 32] for i in range(m)]


def oracle_verify(entry, data, weights=None):
    secrets, cons = G.statement_of(entry)
    cst = C.Statement(entry["proof_label"].encode(), secrets, [(p["name"], p["common"]) for p in entry["points"]], cons)
    pts = np.frombuffer(b"".join(bytes.fromhex(p["hex"]) for p in entry["points"]), np.uint8).reshape(-1, 32)
    a, resp = parse(entry["kind"], data)
    resp = np.frombuffer(b"".join(resp), np.uint8).reshape(-1, 32)
    label = entry["transcript_label"].encode()
    if entry["kind"] == "compact":
        return C.verify_compact(cst, label, pts, np.frombuffer(a, np.uint8), resp)
    w = weights if weights is not None else np.arange(16 * len(cons), dtype=np.uint8).reshape(len(cons), 16) + 1
    return C.verify_batchable(cst, label, pts, np.frombuffer(b"".join(a), np.uint8).reshape(-1, 32), resp, w)


def oracle_batch_verify(entries, datas):
    e0 = entries[0]
    secrets, cons = G.statement_of(e0)
    cst = C.Statement(e0["proof_label"].encode(), secrets, [(p["name"], p["common"]) for p in e0["points"]], cons)
    n = len(entries)
    inst_names = [p["name"] for p in e0["points"] if not p["common"]]
    inst = np.zeros((len(inst_names), n, 32), np.uint8)
    for j, e in enumerate(entries):
        enc = {p["name"]: bytes.fromhex(p["hex"]) for p in e["points"]}
        for r, nm in enumerate(inst_names):
            inst[r, j] = np.frombuffer(enc[nm], np.uint8)
    common = np.frombuffer(b"".join(bytes.fromhex(p["hex"]) for p in e0["points"] if p["common"]), np.uint8).reshape(-1, 32)
    coms, resp = zip(*[parse("batchable", d) for d in datas])
    coms = np.frombuffer(b"".join(b"".join(c) for c in coms), np.uint8).reshape(n, len(cons), 32)
    resp = np.frombuffer(b"".join(b"".join(r) for r in resp), np.uint8).reshape(n, len(secrets), 32)
    w = (np.arange(16 * len(cons) * n, dtype=np.uint32) * 37 % 251).astype(np.uint8).reshape(len(cons), n, 16)
    return C.batch_verify(cst, e0["transcript_label"].encode(), n, inst, common, coms, resp, w)


def test_committed_from_repo_fixtures_are_what_the_oracle_produces():
    man, files = _load("from_repo")
    assert man is not None, "tests/golden/interop/from_repo/ is missing: run tests/golden/interop/make_from_repo.py on a GPU box and commit the output"
    want, want_man = G.produce("oracle")
    assert sorted(files) == sorted(want) and len(files) >= 12
    for fn in want:
        assert files[fn] == want[fn], fn
    assert [e["points"] for e in man["proofs"]] == [e["points"] for e in want_man["proofs"]]
    assert "HIP path" in man["produced_by"]             # the committed bytes came out of the GPU library, not out of this oracle


def test_product_wire_codec_parses_the_fixtures():
    from zkp_amd import toolbox as T
    man, files = _load("from_repo")
    for e in man["proofs"]:
        a, resp = parse(e["kind"], files[e["file"]])
        p = (T.CompactProof if e["kind"] == "compact" else T.BatchableProof).from_bytes(files[e["file"]])
        assert p.responses == resp and (p.challenge == a if e["kind"] == "compact" else p.commitments == a)
        assert p.to_bytes() == files[e["file"]]


@pytest.mark.parametrize("which", ["from_repo", "from_crate"])
def test_oracle_verifiers_accept_the_fixtures_and_reject_corrupted_ones(which):
    man, files = _load(which)
    if man is None:
        pytest.skip("tests/golden/interop/%s is empty (from_crate: run rust/interop's emit_crate_proofs with cargo)" % which)
    batches = {}
    for e in man["proofs"]:
        data = files[e["file"]]
        assert oracle_verify(e, data) == 0, e["file"]
        bad = bytearray(data)
        bad[-32] ^= 1
        assert oracle_verify(e, bytes(bad)) == 1, e["file"]
        if e.get("batch"):
            batches.setdefault(e["batch"], []).append(e)
    assert which != "from_repo" or len(batches) == 2
    for name, es in batches.items():
        assert oracle_batch_verify(es, [files[e["file"]] for e in es]) == 0, name
        bad = bytearray(files[es[1]["file"]])
        bad[-1] ^= 1                                       # still canonical or not: the batch must fail either way
        assert oracle_batch_verify(es, [files[es[0]["file"]], bytes(bad)] + [files[e["file"]] for e in es[2:]]) == 1, name
