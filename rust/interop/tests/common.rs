// UNBUILT SOURCE -- shared by the two interop tests: the statements (the reference's own, reproduced as macro invocations
// with the labels the fixtures were made with) and the manifest schema of tests/golden/interop/*/manifest.json.
#![allow(non_snake_case, dead_code)]
use serde::{Deserialize, Serialize};
use std::path::PathBuf;

#[derive(Serialize, Deserialize, Clone)]
pub struct PointEntry {
    pub name: String,
    pub common: bool,
    pub hex: String,
}
#[derive(Serialize, Deserialize, Clone)]
pub struct ProofEntry {
    pub file: String,
    pub statement: String,        // "dleq" | "capi_dleq" | "cmz10"
    pub kind: String,             // "compact" | "batchable"
    pub api: String,              // "define_proof" | "constraint_api"
    pub proof_label: String,
    pub transcript_label: String,
    pub batch: Option<String>,    // proofs with the same batch name are ALSO verified together with batch_verify
    pub reference: String,
    #[serde(default)]
    pub entropy_hex: String,
    pub points: Vec<PointEntry>,  // in allocation order
}
#[derive(Serialize, Deserialize)]
pub struct Manifest {
    #[serde(default)]
    pub produced_by: String,
    pub proofs: Vec<ProofEntry>,
}

pub fn interop_dir(which: &str) -> PathBuf {
    PathBuf::from(env!("CARGO_MANIFEST_DIR")).join("../../tests/golden/interop").join(which)
}
pub fn point(e: &ProofEntry, name: &str) -> curve25519_dalek::ristretto::CompressedRistretto {
    let p = e.points.iter().find(|p| p.name == name).unwrap_or_else(|| panic!("{}: no point {}", e.file, name));
    curve25519_dalek::ristretto::CompressedRistretto::from_slice(&hex::decode(&p.hex).unwrap())
}
