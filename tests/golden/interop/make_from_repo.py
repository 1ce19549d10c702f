#!/usr/bin/env python3
"""Interoperability fixtures between THIS repository and the Rust crate (dalek-cryptography/zkp 0.7) -- the thing that turns
SURVEY.md row 8(c) from "parity unpinned" into pinned once somebody with `cargo` runs rust/interop (one `cargo test`).

    from_repo/*.bin  + from_repo/manifest.json    bincode `CompactProof` / `BatchableProof` bytes (proofs.rs:14-32) made HERE
                                                  for the reference's own deterministic statements; the crate must accept them
                                                  (rust/interop/tests/verify_repo_proofs.rs)
    from_crate/*.bin + from_crate/manifest.json   the same statements proven BY THE CRATE (rust/interop/tests/emit_crate_proofs.rs
                                                  writes them); tests/test_gpu_interop.py verifies whatever it finds there

`verify_compact` recomputes the challenge from the transcript, so ONE accepted compact proof pins transcript labels and order,
the MSMs, the point codec, scalar reduction and the wire codec in one shot; accepted batchable proofs pin the batch path.

Statements (all inputs deterministic; the entropy that replaces thread_rng at prover.rs:82 is recorded in the manifest):
  dleq_compact / dleq_batchable            tests/zkp.rs:28-70 / :72-113     define_proof! dleq "DLEQ Example Proof", x = 1/89327492234,
                                                                            H = hash_from_bytes::<Sha512>("A VRF input, for instance")
  dleq_batch4_{0..3}                       tests/zkp.rs:115-175             x = 89327492234 (i + 1), H = hash(message i); one batch of 4
  capi_dleq_compact / capi_dleq_batchable  tests/dleq_using_constraint_api.rs:41-127   Prover / Verifier API, allocation order x, B, H, A, G
  cmz10_batch4_{0..3}                      benches/zkp.rs:25-46             cred_show_10 "CMZ cred show n=10", one batch of 4

Two producers must give the same bytes (tests/test_oracle_interop.py on the CPU, tests/test_gpu_interop.py on the GPU):
  produce("oracle")  -- oracle.cbind's prover + a bincode writer of its own (plain struct packing)
  produce("gpu")     -- the product: zkp_amd.toolbox over the HIP library, proofs serialised by the C codec (zkp_proof_*)

    python tests/golden/interop/make_from_repo.py [gpu|oracle]     # rewrites tests/golden/interop/from_repo/
"""
import hashlib
import json
import os
import struct
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(HERE)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

L = 2**252 + 27742317777372353535851937790883648493
BASEPOINT = bytes.fromhex("e2f2ae0a6abc4e71a884a961c500515f58e30b6aa582dd8db6a65945e08d2d76")
FROM_REPO = os.path.join(HERE, "from_repo")
FROM_CRATE = os.path.join(HERE, "from_crate")
TRANSCRIPT_LABEL = b"DLEQTest"
CMZ_TRANSCRIPT_LABEL = b"CMZTest"
MESSAGES = [b"One message", b"Another message", b"A third message", b"A fourth message"]


def sc(x: int) -> bytes:
    return (x % L).to_bytes(32, "little")


def entropy_for(name: str) -> bytes:
    """the 32 bytes that stand in for thread_rng (prover.rs:82): any value gives a valid proof; fixed so the bytes are reproducible"""
    return hashlib.sha256(b"zkp-mi355x interop fixture: " + name.encode()).digest()


def _oracle():
    from oracle import cbind as C
    C.build()
    return C


def mul(x: int, enc: bytes) -> bytes:
    C = _oracle()
    return C.msm_algo("straus_vartime", np.frombuffer(sc(x), np.uint8).reshape(1, 32), np.frombuffer(enc, np.uint8).reshape(1, 32))


def hash_to_point(msg: bytes) -> bytes:
    return _oracle().from_uniform_bytes(hashlib.sha512(msg).digest())       # RistrettoPoint::hash_from_bytes::<Sha512>


def cmz_names(n=10):
    ms = [f"m_{i}" for i in range(1, n + 1)]
    zs = [f"z_{i}" for i in range(1, n + 1)]
    cs = [f"C_{i}" for i in range(1, n + 1)]
    xs = [f"X_{i}" for i in range(1, n + 1)]
    cons = [(cs[i], [(ms[i], "P"), (zs[i], "A")]) for i in range(n)]
    cons.append(("V", [(ms[i], xs[i]) for i in range(n)] + [("minus_z_Q", "Q")]))
    return ms + zs + ["minus_z_Q"], cs + ["P", "Q", "V"], xs + ["A", "B"], cons


def cases():
    """-> list of dicts: name, kind, api, proof_label, transcript_label, secrets {name: int}, points [(name, is_common, enc)] in
    ALLOCATION order, constraints, batch (name of the batch the proof belongs to, or None)"""
    out = []
    dleq_cons = [("A", [("x", "G")]), ("B", [("x", "H")])]
    # tests/zkp.rs:28-113
    H = hash_to_point(b"A VRF input, for instance")
    x = pow(89327492234, L - 2, L)
    pts = [("A", False, mul(x, BASEPOINT)), ("B", False, mul(x, H)), ("H", False, H), ("G", True, BASEPOINT)]
    for kind in ("compact", "batchable"):
        out.append(dict(name="dleq_" + kind, kind=kind, api="define_proof", proof_label="DLEQ Example Proof", transcript_label=TRANSCRIPT_LABEL.decode(),
                        secrets={"x": x}, points=pts, constraints=dleq_cons, batch=None, reference="tests/zkp.rs:28-113", statement="dleq"))
    # tests/zkp.rs:115-175
    for i, msg in enumerate(MESSAGES):
        Hm = hash_to_point(msg)
        xi = 89327492234 * (i + 1)
        pts = [("A", False, mul(xi, BASEPOINT)), ("B", False, mul(xi, Hm)), ("H", False, Hm), ("G", True, BASEPOINT)]
        out.append(dict(name="dleq_batch4_%d" % i, kind="batchable", api="define_proof", proof_label="DLEQ Example Proof", transcript_label=TRANSCRIPT_LABEL.decode(),
                        secrets={"x": xi}, points=pts, constraints=dleq_cons, batch="dleq_batch4", reference="tests/zkp.rs:115-175", statement="dleq"))
    # tests/dleq_using_constraint_api.rs:41-127: A = x B, G = x H; allocation order x, B, H, A, G; every point allocated per proof
    B = BASEPOINT
    Hc = hash_to_point(B)
    xc = 89327492234
    pts = [("B", False, B), ("H", False, Hc), ("A", False, mul(xc, B)), ("G", False, mul(xc, Hc))]
    capi_cons = [("A", [("x", "B")]), ("G", [("x", "H")])]
    for kind in ("compact", "batchable"):
        out.append(dict(name="capi_dleq_" + kind, kind=kind, api="constraint_api", proof_label="DLEQProof", transcript_label=TRANSCRIPT_LABEL.decode(),
                        secrets={"x": xc}, points=pts, constraints=capi_cons, batch=None, reference="tests/dleq_using_constraint_api.rs:41-127", statement="capi_dleq"))
    # benches/zkp.rs:25-46, four presentations under common issuer parameters
    secrets_l, inst_l, common_l, cons = cmz_names()
    h = lambda tag: int.from_bytes(hashlib.sha512(b"zkp-mi355x interop cmz: " + tag.encode()).digest(), "little") % L
    common = {nm: mul(h(nm), BASEPOINT) for nm in common_l}
    for j in range(4):
        s = {nm: h("%s/%d" % (nm, j)) for nm in secrets_l}
        P, Q = mul(h("P/%d" % j), BASEPOINT), mul(h("Q/%d" % j), BASEPOINT)
        C = _oracle()
        encs = dict(common, P=P, Q=Q)

        def lc(terms):
            scal = np.frombuffer(b"".join(sc(s[a]) for a, _ in terms), np.uint8).reshape(-1, 32)
            ptsa = np.frombuffer(b"".join(encs[b] for _, b in terms), np.uint8).reshape(-1, 32)
            return C.msm_algo("straus_vartime", scal, ptsa)
        for lhs, terms in cons:
            encs[lhs] = lc(terms)
        pts = [(nm, False, encs[nm]) for nm in inst_l] + [(nm, True, encs[nm]) for nm in common_l]
        out.append(dict(name="cmz10_batch4_%d" % j, kind="batchable", api="define_proof", proof_label="CMZ cred show n=10", transcript_label=CMZ_TRANSCRIPT_LABEL.decode(),
                        secrets=s, points=pts, constraints=cons, batch="cmz10_batch4", reference="benches/zkp.rs:25-46", secret_order=secrets_l, statement="cmz10"))
    return out


# ---- bincode 1.x of the serde-derived structs (proofs.rs:14-32): Scalar / CompressedRistretto = 32 raw bytes, Vec = u64 LE length + elements
def bincode_compact(challenge: bytes, responses) -> bytes:
    return challenge + struct.pack("<Q", len(responses)) + b"".join(responses)


def bincode_batchable(commitments, responses) -> bytes:
    return struct.pack("<Q", len(commitments)) + b"".join(commitments) + struct.pack("<Q", len(responses)) + b"".join(responses)


def _secret_names(case):
    return case.get("secret_order") or list(case["secrets"].keys())


def produce_oracle(case) -> bytes:
    C = _oracle()
    names = _secret_names(case)
    cst = C.Statement(case["proof_label"].encode(), names, [(n, c) for n, c, _ in case["points"]], case["constraints"])
    secrets = np.frombuffer(b"".join(sc(case["secrets"][n]) for n in names), np.uint8).reshape(-1, 32)
    pts = np.frombuffer(b"".join(e for _, _, e in case["points"]), np.uint8).reshape(-1, 32)
    chal, resp, coms, _ = C.prove(cst, case["transcript_label"].encode(), secrets, pts, entropy_for(case["name"]))
    r = [x.tobytes() for x in resp]
    return bincode_compact(chal.tobytes(), r) if case["kind"] == "compact" else bincode_batchable([x.tobytes() for x in coms], r)


def produce_gpu(case, eng) -> bytes:
    from zkp_amd import toolbox as T
    names = _secret_names(case)
    label = case["transcript_label"].encode()
    ent = entropy_for(case["name"])
    if case["api"] == "constraint_api":
        prover = T.Prover(case["proof_label"].encode(), T.Transcript(label), eng)
        sv = {n: prover.allocate_scalar(n.encode(), case["secrets"][n]) for n in names}
        pv = {n: prover.allocate_point(n.encode(), e)[0] for n, _, e in case["points"]}
        for lhs, lc in case["constraints"]:
            prover.constrain(pv[lhs], [(sv[a], pv[b]) for a, b in lc])
        proof = prover.prove_compact(ent) if case["kind"] == "compact" else prover.prove_batchable(ent)
    else:
        mod = T.define_proof(case["name"], case["proof_label"].encode(), names, [n for n, c, _ in case["points"] if not c], [n for n, c, _ in case["points"] if c],
                             case["constraints"])
        pts = {n: e for n, _, e in case["points"]}
        proof = (mod.prove_compact if case["kind"] == "compact" else mod.prove_batchable)(eng, T.Transcript(label), case["secrets"], pts, ent)
    return proof.to_bytes()


def manifest(cs):
    return {"_about": "bincode proofs made by zkp-mi355x for the reference's own statements; see tests/golden/interop/make_from_repo.py and rust/interop/",
            "wire_format": "bincode 1.x of proofs.rs:14-32: CompactProof = challenge[32] | u64le m | m x 32; BatchableProof = u64le n | n x 32 | u64le m | m x 32",
            "proofs": [{"file": c["name"] + ".bin", "statement": c["statement"], "kind": c["kind"], "api": c["api"], "proof_label": c["proof_label"], "transcript_label": c["transcript_label"],
                        "batch": c["batch"], "reference": c["reference"], "entropy_hex": entropy_for(c["name"]).hex(),
                        "points": [{"name": n, "common": bool(cm), "hex": e.hex()} for n, cm, e in c["points"]]} for c in cs]}


def statement_of(entry):
    """(secret names, constraints) of a manifest entry's statement id -- what a consumer needs besides the manifest"""
    if entry["statement"] == "dleq":
        return ["x"], [("A", [("x", "G")]), ("B", [("x", "H")])]
    if entry["statement"] == "capi_dleq":
        return ["x"], [("A", [("x", "B")]), ("G", [("x", "H")])]
    if entry["statement"] == "cmz10":
        secrets_l, _, _, cons = cmz_names()
        return secrets_l, cons
    raise ValueError("unknown statement id %r" % entry["statement"])


def produce(backend: str, eng=None):
    """-> ({file name: bytes}, manifest dict)"""
    cs = cases()
    files = {}
    for c in cs:
        files[c["name"] + ".bin"] = produce_oracle(c) if backend == "oracle" else produce_gpu(c, eng)
    return files, manifest(cs)


def main():
    backend = sys.argv[1] if len(sys.argv) > 1 else "gpu"
    eng = None
    if backend == "gpu":
        from zkp_amd.engine import Engine
        eng = Engine(0)
    files, man = produce(backend, eng)
    out_dir = sys.argv[2] if len(sys.argv) > 2 else FROM_REPO
    os.makedirs(out_dir, exist_ok=True)
    for fn, data in files.items():
        open(os.path.join(out_dir, fn), "wb").write(data)
    man["produced_by"] = "zkp_amd (HIP path on MI355X): zkp_prove_batch + zkp_proof_*_encode" if backend == "gpu" else "oracle.cbind (CPU restatement)"
    json.dump(man, open(os.path.join(out_dir, "manifest.json"), "w"), indent=1)
    print("wrote %d proofs to %s (%s)" % (len(files), out_dir, backend))


if __name__ == "__main__":
    main()
