//! UNBUILT SOURCE.  Nothing here: the crate exists for its two integration tests (tests/verify_repo_proofs.rs,
//! tests/emit_crate_proofs.rs), which exchange proofs between the real `zkp` crate and zkp-mi355x through files.
