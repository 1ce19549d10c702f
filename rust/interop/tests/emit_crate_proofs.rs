// UNBUILT SOURCE (no Rust toolchain in the build image).
//
// The other direction: the REAL crate proves the same statements (with its own thread_rng blindings) and writes
// tests/golden/interop/from_crate/{*.bin, manifest.json}; commit those files and `pytest -m gpu tests/test_gpu_interop.py`
// verifies every one of them on the MI355X path (and on the CPU oracle in tests/test_oracle_interop.py), asserting for
// compact proofs that the recomputed challenge equals the proof's.
//     cargo test --manifest-path rust/interop/Cargo.toml --test emit_crate_proofs -- --nocapture
#![allow(non_snake_case)]
#[macro_use]
extern crate zkp;
mod common;
use common::*;
use curve25519_dalek::constants as dalek_constants;
use curve25519_dalek::ristretto::{CompressedRistretto, RistrettoPoint};
use curve25519_dalek::scalar::Scalar;
use sha2::Sha512;
use zkp::toolbox::{prover::Prover, SchnorrCS};
use zkp::Transcript;

define_proof! {dleq, "DLEQ Example Proof", (x), (A, B, H), (G) : A = (x * G), B = (x * H) }

define_proof! {
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

fn pe(name: &str, common: bool, c: &CompressedRistretto) -> PointEntry {
    PointEntry { name: name.into(), common, hex: hex::encode(c.as_bytes()) }
}
fn entry(file: &str, statement: &str, kind: &str, api: &str, proof_label: &str, tlabel: &str, batch: Option<&str>, reference: &str, points: Vec<PointEntry>) -> ProofEntry {
    ProofEntry { file: file.into(), statement: statement.into(), kind: kind.into(), api: api.into(), proof_label: proof_label.into(),
                 transcript_label: tlabel.into(), batch: batch.map(|s| s.into()), reference: reference.into(), entropy_hex: String::new(), points }
}

#[test]
fn emit() {
    let dir = interop_dir("from_crate");
    std::fs::create_dir_all(&dir).unwrap();
    let mut proofs = vec![];
    let G = dalek_constants::RISTRETTO_BASEPOINT_POINT;
    // tests/zkp.rs:28-113
    {
        let H = RistrettoPoint::hash_from_bytes::<Sha512>(b"A VRF input, for instance");
        let x = Scalar::from(89327492234u64).invert();
        let (A, B) = (&x * &dalek_constants::RISTRETTO_BASEPOINT_TABLE, &x * &H);
        let mut t = Transcript::new(b"DLEQTest");
        let (p, pts) = dleq::prove_compact(&mut t, dleq::ProveAssignments { x: &x, A: &A, B: &B, G: &G, H: &H });
        std::fs::write(dir.join("dleq_compact.bin"), bincode::serialize(&p).unwrap()).unwrap();
        let points = vec![pe("A", false, &pts.A), pe("B", false, &pts.B), pe("H", false, &pts.H), pe("G", true, &pts.G)];
        proofs.push(entry("dleq_compact.bin", "dleq", "compact", "define_proof", "DLEQ Example Proof", "DLEQTest", None, "tests/zkp.rs:28-70", points.clone()));
        let mut t = Transcript::new(b"DLEQTest");
        let (p, _) = dleq::prove_batchable(&mut t, dleq::ProveAssignments { x: &x, A: &A, B: &B, G: &G, H: &H });
        std::fs::write(dir.join("dleq_batchable.bin"), bincode::serialize(&p).unwrap()).unwrap();
        proofs.push(entry("dleq_batchable.bin", "dleq", "batchable", "define_proof", "DLEQ Example Proof", "DLEQTest", None, "tests/zkp.rs:72-113", points));
    }
    // tests/zkp.rs:115-175
    for (i, msg) in ["One message", "Another message", "A third message", "A fourth message"].iter().enumerate() {
        let H = RistrettoPoint::hash_from_bytes::<Sha512>(msg.as_bytes());
        let x = Scalar::from(89327492234u64) * Scalar::from((i + 1) as u64);
        let (A, B) = (&x * &dalek_constants::RISTRETTO_BASEPOINT_TABLE, &x * &H);
        let mut t = Transcript::new(b"DLEQTest");
        let (p, pts) = dleq::prove_batchable(&mut t, dleq::ProveAssignments { x: &x, A: &A, B: &B, G: &G, H: &H });
        let f = format!("dleq_batch4_{}.bin", i);
        std::fs::write(dir.join(&f), bincode::serialize(&p).unwrap()).unwrap();
        proofs.push(entry(&f, "dleq", "batchable", "define_proof", "DLEQ Example Proof", "DLEQTest", Some("dleq_batch4"), "tests/zkp.rs:115-175",
                          vec![pe("A", false, &pts.A), pe("B", false, &pts.B), pe("H", false, &pts.H), pe("G", true, &pts.G)]));
    }
    // tests/dleq_using_constraint_api.rs:41-127
    for kind in ["compact", "batchable"].iter() {
        let B = dalek_constants::RISTRETTO_BASEPOINT_POINT;
        let H = RistrettoPoint::hash_from_bytes::<Sha512>(B.compress().as_bytes());
        let x = Scalar::from(89327492234u64);
        let (A, Gp) = (B * x, H * x);
        let mut t = Transcript::new(b"DLEQTest");
        let mut prover = Prover::new(b"DLEQProof", &mut t);
        let vx = prover.allocate_scalar(b"x", x);
        let (vB, cB) = prover.allocate_point(b"B", B);
        let (vH, cH) = prover.allocate_point(b"H", H);
        let (vA, cA) = prover.allocate_point(b"A", A);
        let (vG, cG) = prover.allocate_point(b"G", Gp);
        prover.constrain(vA, vec![(vx, vB)]);
        prover.constrain(vG, vec![(vx, vH)]);
        let f = format!("capi_dleq_{}.bin", kind);
        let bytes = if *kind == "compact" { bincode::serialize(&prover.prove_compact()).unwrap() } else { bincode::serialize(&prover.prove_batchable()).unwrap() };
        std::fs::write(dir.join(&f), bytes).unwrap();
        proofs.push(entry(&f, "capi_dleq", kind, "constraint_api", "DLEQProof", "DLEQTest", None, "tests/dleq_using_constraint_api.rs:41-127",
                          vec![pe("B", false, &cB), pe("H", false, &cH), pe("A", false, &cA), pe("G", false, &cG)]));
    }
    // benches/zkp.rs:25-46: four presentations under common issuer parameters (any valid assignment will do)
    {
        let s = |tag: &str| Scalar::hash_from_bytes::<Sha512>(format!("zkp interop cmz: {}", tag).as_bytes());
        let X: Vec<RistrettoPoint> = (1..=10).map(|i| &s(&format!("X_{}", i)) * &dalek_constants::RISTRETTO_BASEPOINT_TABLE).collect();
        let (A, Bc) = (&s("A") * &dalek_constants::RISTRETTO_BASEPOINT_TABLE, &s("B") * &dalek_constants::RISTRETTO_BASEPOINT_TABLE);
        for j in 0..4 {
            let m: Vec<Scalar> = (1..=10).map(|i| s(&format!("m_{}/{}", i, j))).collect();
            let z: Vec<Scalar> = (1..=10).map(|i| s(&format!("z_{}/{}", i, j))).collect();
            let mzq = s(&format!("minus_z_Q/{}", j));
            let (P, Q) = (&s(&format!("P/{}", j)) * &dalek_constants::RISTRETTO_BASEPOINT_TABLE, &s(&format!("Q/{}", j)) * &dalek_constants::RISTRETTO_BASEPOINT_TABLE);
            let C: Vec<RistrettoPoint> = (0..10).map(|i| m[i] * P + z[i] * A).collect();
            let V = (0..10).map(|i| m[i] * X[i]).fold(mzq * Q, |a, b| a + b);
            let mut t = Transcript::new(b"CMZTest");
            let (p, pts) = cred_show_10::prove_batchable(&mut t, cred_show_10::ProveAssignments {
                m_1: &m[0], m_2: &m[1], m_3: &m[2], m_4: &m[3], m_5: &m[4], m_6: &m[5], m_7: &m[6], m_8: &m[7], m_9: &m[8], m_10: &m[9],
                z_1: &z[0], z_2: &z[1], z_3: &z[2], z_4: &z[3], z_5: &z[4], z_6: &z[5], z_7: &z[6], z_8: &z[7], z_9: &z[8], z_10: &z[9], minus_z_Q: &mzq,
                C_1: &C[0], C_2: &C[1], C_3: &C[2], C_4: &C[3], C_5: &C[4], C_6: &C[5], C_7: &C[6], C_8: &C[7], C_9: &C[8], C_10: &C[9], P: &P, Q: &Q, V: &V,
                X_1: &X[0], X_2: &X[1], X_3: &X[2], X_4: &X[3], X_5: &X[4], X_6: &X[5], X_7: &X[6], X_8: &X[7], X_9: &X[8], X_10: &X[9], A: &A, B: &Bc });
            let f = format!("cmz10_batch4_{}.bin", j);
            std::fs::write(dir.join(&f), bincode::serialize(&p).unwrap()).unwrap();
            let cp = [&pts.C_1, &pts.C_2, &pts.C_3, &pts.C_4, &pts.C_5, &pts.C_6, &pts.C_7, &pts.C_8, &pts.C_9, &pts.C_10];
            let xp = [&pts.X_1, &pts.X_2, &pts.X_3, &pts.X_4, &pts.X_5, &pts.X_6, &pts.X_7, &pts.X_8, &pts.X_9, &pts.X_10];
            let mut points: Vec<PointEntry> = (0..10).map(|i| pe(&format!("C_{}", i + 1), false, cp[i])).collect();
            points.extend(vec![pe("P", false, &pts.P), pe("Q", false, &pts.Q), pe("V", false, &pts.V)]);
            points.extend((0..10).map(|i| pe(&format!("X_{}", i + 1), true, xp[i])));
            points.extend(vec![pe("A", true, &pts.A), pe("B", true, &pts.B)]);
            proofs.push(entry(&f, "cmz10", "batchable", "define_proof", "CMZ cred show n=10", "CMZTest", Some("cmz10_batch4"), "benches/zkp.rs:25-46", points));
        }
    }
    let m = Manifest { produced_by: "dalek-cryptography/zkp (Rust crate), rust/interop/tests/emit_crate_proofs.rs".into(), proofs };
    std::fs::write(dir.join("manifest.json"), serde_json::to_vec_pretty(&m).unwrap()).unwrap();
    println!("wrote {} proofs to {}", m.proofs.len(), dir.display());
}
