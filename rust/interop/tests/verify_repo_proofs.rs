// UNBUILT SOURCE (no Rust toolchain in the build image).
//
// The pin of SURVEY.md row 8(c): every proof under tests/golden/interop/from_repo/ was made by zkp-mi355x (HIP path; the same
// bytes come out of its CPU oracle) for one of the reference's own statements.  The REAL crate must
//   * deserialise it with bincode (proofs.rs:14-32: wire layout + canonical scalars),
//   * accept it: verify_compact recomputes the Fiat-Shamir challenge, so acceptance pins the transcript labels and their order
//     (macros.rs:206-311, mod.rs:165-228), the MSMs and the point codec; verify_batchable / batch_verify pin the batch path
//     (verifier.rs:123-173, batch_verifier.rs:67-235),
//   * reject it after a one-bit change (so "accept" is not vacuous).
#![allow(non_snake_case)]
#[macro_use]
extern crate zkp;
mod common;
use common::*;
use curve25519_dalek::ristretto::CompressedRistretto;
use std::collections::BTreeMap;
use zkp::toolbox::{batch_verifier::BatchVerifier, verifier::Verifier, SchnorrCS};
use zkp::{BatchableProof, CompactProof, Transcript};

define_proof! {dleq, "DLEQ Example Proof", (x), (A, B, H), (G) : A = (x * G), B = (x * H) }          // tests/zkp.rs:28

define_proof! {                                                                                          // benches/zkp.rs:25-46
    cred_show_10, "CMZ cred show n=10",
    (m_1, m_2, m_3, m_4, m_5, m_6, m_7, m_8, m_9, m_10, z_1, z_2, z_3, z_4, z_5, z_6, z_7, z_8, z_9, z_10, minus_z_Q),
    (C_1, C_2, C_3, C_4, C_5, C_6, C_7, C_8, C_9, C_10, P, Q, V),
    (X_1, X_2, X_3, X_4, X_5, X_6, X_7, X_8, X_9, X_10, A, B)
    :
    C_1 = (m_1 * P + z_1 * A), C_2 = (m_2 * P + z_2 * A), C_3 = (m_3 * P + z_3 * A), C_4 = (m_4 * P + z_4 * A),
    C_5 = (m_5 * P + z_5 * A), C_6 = (m_6 * P + z_6 * A), C_7 = (m_7 * P + z_7 * A), C_8 = (m_8 * P + z_8 * A),
    C_9 = (m_9 * P + z_9 * A), C_10 = (m_10 * P + z_10 * A),
    V = (m_1*X_1 + m_2*X_2 + m_3*X_3 + m_4*X_4 + m_5*X_5 + m_6*X_6 + m_7*X_7 + m_8*X_8 + m_9*X_9 + m_10*X_10 + minus_z_Q*Q)
}

fn capi_statement<CS: SchnorrCS>(cs: &mut CS, x: CS::ScalarVar, A: CS::PointVar, G: CS::PointVar, B: CS::PointVar, H: CS::PointVar) {
    cs.constrain(A, vec![(x, B)]);                                                                       // tests/dleq_using_constraint_api.rs:30-39
    cs.constrain(G, vec![(x, H)]);
}

fn cmz_assignments<'a>(p: &'a BTreeMap<String, CompressedRistretto>) -> cred_show_10::VerifyAssignments<'a> {
    cred_show_10::VerifyAssignments {
        C_1: &p["C_1"], C_2: &p["C_2"], C_3: &p["C_3"], C_4: &p["C_4"], C_5: &p["C_5"], C_6: &p["C_6"], C_7: &p["C_7"], C_8: &p["C_8"],
        C_9: &p["C_9"], C_10: &p["C_10"], P: &p["P"], Q: &p["Q"], V: &p["V"],
        X_1: &p["X_1"], X_2: &p["X_2"], X_3: &p["X_3"], X_4: &p["X_4"], X_5: &p["X_5"], X_6: &p["X_6"], X_7: &p["X_7"], X_8: &p["X_8"],
        X_9: &p["X_9"], X_10: &p["X_10"], A: &p["A"], B: &p["B"],
    }
}

fn verify_one(e: &ProofEntry, bytes: &[u8]) -> Result<(), zkp::ProofError> {
    let pts: BTreeMap<String, CompressedRistretto> = e.points.iter().map(|p| (p.name.clone(), point(e, &p.name))).collect();
    let mut t = Transcript::new(Box::leak(e.transcript_label.clone().into_boxed_str()).as_bytes());
    match (e.statement.as_str(), e.kind.as_str()) {
        ("dleq", "compact") => {
            let proof: CompactProof = bincode::deserialize(bytes).expect("bincode");
            dleq::verify_compact(&proof, &mut t, dleq::VerifyAssignments { A: &pts["A"], B: &pts["B"], H: &pts["H"], G: &pts["G"] })
        }
        ("dleq", "batchable") => {
            let proof: BatchableProof = bincode::deserialize(bytes).expect("bincode");
            dleq::verify_batchable(&proof, &mut t, dleq::VerifyAssignments { A: &pts["A"], B: &pts["B"], H: &pts["H"], G: &pts["G"] })
        }
        ("cmz10", "batchable") => {
            let proof: BatchableProof = bincode::deserialize(bytes).expect("bincode");
            cred_show_10::verify_batchable(&proof, &mut t, cmz_assignments(&pts))
        }
        ("capi_dleq", kind) => {
            let mut v = Verifier::new(b"DLEQProof", &mut t);
            let x = v.allocate_scalar(b"x");                                                             // allocation order: x, B, H, A, G
            let B = v.allocate_point(b"B", pts["B"])?;
            let H = v.allocate_point(b"H", pts["H"])?;
            let A = v.allocate_point(b"A", pts["A"])?;
            let G = v.allocate_point(b"G", pts["G"])?;
            capi_statement(&mut v, x, A, G, B, H);
            if kind == "compact" {
                v.verify_compact(&bincode::deserialize::<CompactProof>(bytes).expect("bincode"))
            } else {
                v.verify_batchable(&bincode::deserialize::<BatchableProof>(bytes).expect("bincode"))
            }
        }
        other => panic!("unknown statement / kind {:?}", other),
    }
}

#[test]
fn crate_accepts_every_proof_made_by_zkp_mi355x() {
    let dir = interop_dir("from_repo");
    let m: Manifest = serde_json::from_slice(&std::fs::read(dir.join("manifest.json")).expect("manifest")).unwrap();
    assert!(m.proofs.len() >= 12);
    for e in &m.proofs {
        let bytes = std::fs::read(dir.join(&e.file)).unwrap();
        assert!(verify_one(e, &bytes).is_ok(), "{} (made by {}) was rejected by the crate", e.file, m.produced_by);
        // one flipped bit in the last response: must be rejected (or fail to deserialise)
        let mut bad = bytes.clone();
        let n = bad.len();
        bad[n - 32] ^= 1;
        let r = std::panic::catch_unwind(|| verify_one(e, &bad));
        assert!(!matches!(r, Ok(Ok(()))), "{}: a corrupted proof was accepted", e.file);
    }
}

#[test]
fn crate_batch_verifies_the_batches_made_by_zkp_mi355x() {
    let dir = interop_dir("from_repo");
    let m: Manifest = serde_json::from_slice(&std::fs::read(dir.join("manifest.json")).unwrap()).unwrap();
    let mut batches: BTreeMap<String, Vec<&ProofEntry>> = BTreeMap::new();
    for e in &m.proofs {
        if let Some(b) = &e.batch { batches.entry(b.clone()).or_default().push(e); }
    }
    assert!(batches.len() >= 2);
    for (name, es) in &batches {
        let proofs: Vec<BatchableProof> = es.iter().map(|e| bincode::deserialize(&std::fs::read(dir.join(&e.file)).unwrap()).unwrap()).collect();
        let label: &'static [u8] = Box::leak(es[0].transcript_label.clone().into_boxed_str()).as_bytes();
        let mut ts = vec![Transcript::new(label); es.len()];
        let col = |n: &str| -> Vec<CompressedRistretto> { es.iter().map(|e| point(e, n)).collect() };
        let ok = match es[0].statement.as_str() {
            "dleq" => dleq::batch_verify(&proofs, ts.iter_mut().collect(), dleq::BatchVerifyAssignments { A: col("A"), B: col("B"), H: col("H"), G: point(es[0], "G") }),
            "cmz10" => cred_show_10::batch_verify(&proofs, ts.iter_mut().collect(), cred_show_10::BatchVerifyAssignments {
                C_1: col("C_1"), C_2: col("C_2"), C_3: col("C_3"), C_4: col("C_4"), C_5: col("C_5"), C_6: col("C_6"), C_7: col("C_7"), C_8: col("C_8"),
                C_9: col("C_9"), C_10: col("C_10"), P: col("P"), Q: col("Q"), V: col("V"),
                X_1: point(es[0], "X_1"), X_2: point(es[0], "X_2"), X_3: point(es[0], "X_3"), X_4: point(es[0], "X_4"), X_5: point(es[0], "X_5"),
                X_6: point(es[0], "X_6"), X_7: point(es[0], "X_7"), X_8: point(es[0], "X_8"), X_9: point(es[0], "X_9"), X_10: point(es[0], "X_10"),
                A: point(es[0], "A"), B: point(es[0], "B") }),
            s => panic!("no batch form for statement {}", s),
        };
        assert!(ok.is_ok(), "batch {} (made by {}) was rejected by the crate's BatchVerifier", name, m.produced_by);
    }
    let _ = BatchVerifier::new;      // (the constraint-API batch form is covered by the reference's own tests; nothing to exchange here)
}
