/* The reference's DLEQ test (tests/dleq_using_constraint_api.rs:41-56, benches/dleq.rs:51-56 = BASELINE configs[0]) in plain C99 through
 * include/zkp_toolbox.h alone -- no Python, no torch, nothing but the two shared libraries a Rust `-sys` crate would link:
 *
 *     A = x * B,  G = x * H      (B = the ristretto basepoint, H = hash(B), x = 89327492234; labels "DLEQTest" / "DLEQProof")
 *
 * prove -> bincode wire format (proofs.rs:14-32) -> parse -> verify, compact and batchable, then a tampered response must be refused, then a batch
 * of N copies through BatchVerifier's entry point.
 *
 *     gcc -std=c99 -I include examples/dleq_c_abi.c -L zkp_amd -lzkp_toolbox -lzkp_mi355x -Wl,-rpath,$PWD/zkp_amd -o /tmp/dleq_c_abi
 *     /tmp/dleq_c_abi            host backend (ctx == NULL: no GPU needed)
 *     /tmp/dleq_c_abi gpu 4096   the same calls on GPU 0 for a batch of 4096 proofs
 *
 * The instance points are constants (the C ABI has no stand-alone scalar multiplication on the host; a Rust caller has dalek's): tests/test_host_backend.py
 * derives the same bytes with the oracle.  tests/test_host_toolbox.py builds and runs this file and compares the proof it prints with the Python
 * object layer's for the same entropy. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "zkp_toolbox.h"

static void unhex(uint8_t* out, const char* hex, size_t n) {
  size_t i;
  for (i = 0; i < n; ++i) {
    unsigned v = 0;
    sscanf(hex + 2 * i, "%2x", &v);
    out[i] = (uint8_t)v;
  }
}
static void print_hex(const char* name, const uint8_t* p, size_t n) {
  size_t i;
  printf("%s ", name);
  for (i = 0; i < n; ++i) printf("%02x", p[i]);
  printf("\n");
}
#define CHECK(cond, what)                                                                        \
  do {                                                                                           \
    if (!(cond)) {                                                                               \
      fprintf(stderr, "FAILED: %s (line %d; zkp_last_error: %s)\n", what, __LINE__, zkp_last_error()); \
      return 1;                                                                                  \
    }                                                                                            \
  } while (0)

int main(int argc, char** argv) {
  const char* hex_B = "e2f2ae0a6abc4e71a884a961c500515f58e30b6aa582dd8db6a65945e08d2d76";
  const char* hex_H = "90ca11cd6c6227cb0abc39e2710c444ae6617ea81898e716353f3410d9656605";
  const char* hex_A = "241dacbe397b94c04f21eaa1f1df11878ae3b9833351d8cc957c9c7c13a36800";
  const char* hex_G = "9ea41491e551a36b01db7bcf332ec44633d747e0a9ef73b5da77be43bbe5687e";
  const char* hex_x = "8a5c55cc14000000000000000000000000000000000000000000000000000000";
  const int use_gpu = argc > 1 && strcmp(argv[1], "gpu") == 0;
  const uint32_t N = use_gpu && argc > 2 ? (uint32_t)strtoul(argv[2], NULL, 10) : 1;
  zkp_ctx* ctx = NULL;
  zkp_statement* st;
  int vx, vB, vH, vA, vG, rc;
  uint32_t sec[1], pt[1], j, i;
  uint8_t *ts, *secrets, *inst, *entropy, *chal, *resp, *coms, *results;
  uint8_t t0[ZKP_TRANSCRIPT_BYTES];
  uint8_t wire[256], chal2[32], resp2[32], coms2[64];
  uint32_t m2 = 0, nc2 = 0;
  size_t used = 0;

  if (use_gpu) CHECK(zkp_ctx_create(&ctx, 0) == 0, "zkp_ctx_create");

  /* ---- the statement: variables in ALLOCATION order (it fixes the transcript): x, B, H, A, G -- all per-proof points, as the test allocates them */
  st = zkp_statement_new("DLEQProof");
  vx = zkp_statement_add_secret(st, "x");
  vB = zkp_statement_add_point(st, "B", 0);
  vH = zkp_statement_add_point(st, "H", 0);
  vA = zkp_statement_add_point(st, "A", 0);
  vG = zkp_statement_add_point(st, "G", 0);
  CHECK(vx == 0 && vB == 0 && vH == 1 && vA == 2 && vG == 3, "variable indices");
  sec[0] = (uint32_t)vx; pt[0] = (uint32_t)vB;
  CHECK(zkp_statement_constrain(st, (uint32_t)vA, 1, sec, pt) == ZKP_TB_OK, "constrain A = x * B");
  pt[0] = (uint32_t)vH;
  CHECK(zkp_statement_constrain(st, (uint32_t)vG, 1, sec, pt) == ZKP_TB_OK, "constrain G = x * H");
  CHECK(zkp_statement_num_secrets(st) == 1 && zkp_statement_num_instance(st) == 4 && zkp_statement_num_common(st) == 0 && zkp_statement_num_constraints(st) == 2,
        "statement shape");

  /* ---- buffers: transcripts [N][208], secrets [N][1][32], instance points [4][N][32] (row = variable), outputs */
  ts = malloc((size_t)N * ZKP_TRANSCRIPT_BYTES);
  secrets = malloc((size_t)N * 32);
  inst = malloc((size_t)4 * N * 32);
  entropy = malloc((size_t)N * 32);
  chal = malloc((size_t)N * 32);
  resp = malloc((size_t)N * 32);
  coms = malloc((size_t)N * 64);
  results = malloc(N);
  CHECK(ts && secrets && inst && entropy && chal && resp && coms && results, "malloc");
  zkp_transcript_init(t0, (const uint8_t*)"DLEQTest", 8);
  for (j = 0; j < N; ++j) {
    memcpy(ts + (size_t)j * ZKP_TRANSCRIPT_BYTES, t0, ZKP_TRANSCRIPT_BYTES);
    unhex(secrets + 32 * (size_t)j, hex_x, 32);
    unhex(inst + 32 * ((size_t)0 * N + j), hex_B, 32);
    unhex(inst + 32 * ((size_t)1 * N + j), hex_H, 32);
    unhex(inst + 32 * ((size_t)2 * N + j), hex_A, 32);
    unhex(inst + 32 * ((size_t)3 * N + j), hex_G, 32);
    for (i = 0; i < 32; ++i) entropy[32 * (size_t)j + i] = (uint8_t)(i + j);      /* what thread_rng() contributes (prover.rs:82); NULL = from the OS */
  }

  /* ---- prove: both proof formats' fields in one call (proofs.rs:15-20, 27-32) */
  rc = zkp_prove_batch(ctx, st, N, ts, secrets, inst, NULL, entropy, 0, chal, resp, coms);
  CHECK(rc == ZKP_TB_OK, "zkp_prove_batch");
  print_hex("challenge", chal, 32);
  print_hex("response", resp, 32);
  print_hex("commitments", coms, 64);

  /* ---- compact proof over the wire and back, then verify (tests/zkp.rs:53-54) */
  CHECK(zkp_proof_compact_size(1) == 72, "compact size");
  CHECK(zkp_proof_compact_encode(chal, resp, 1, wire, sizeof(wire)) == ZKP_TB_OK, "compact encode");
  CHECK(zkp_proof_compact_decode(wire, 72, chal2, resp2, 1, &m2, &used) == ZKP_TB_OK && m2 == 1 && used == 72, "compact decode");
  CHECK(memcmp(chal2, chal, 32) == 0 && memcmp(resp2, resp, 32) == 0, "compact round trip");
  for (j = 0; j < N; ++j) memcpy(ts + (size_t)j * ZKP_TRANSCRIPT_BYTES, t0, ZKP_TRANSCRIPT_BYTES);
  memset(results, 7, N);
  rc = zkp_verify_compact_batch(ctx, st, N, ts, inst, NULL, chal, resp, 0, results);
  CHECK(rc == ZKP_TB_OK, "zkp_verify_compact_batch");
  for (j = 0; j < N; ++j) CHECK(results[j] == 0, "compact proof verifies");

  /* ---- batchable proof over the wire and back, verified one by one and as a batch (tests/zkp.rs:96-97, 142-152) */
  CHECK(zkp_proof_batchable_encode(coms, 2, resp, 1, wire, sizeof(wire)) == ZKP_TB_OK, "batchable encode");
  CHECK(zkp_proof_batchable_decode(wire, zkp_proof_batchable_size(2, 1), coms2, 2, &nc2, resp2, 1, &m2, &used) == ZKP_TB_OK && nc2 == 2 && m2 == 1, "batchable decode");
  CHECK(memcmp(coms2, coms, 64) == 0 && memcmp(resp2, resp, 32) == 0, "batchable round trip");
  for (j = 0; j < N; ++j) memcpy(ts + (size_t)j * ZKP_TRANSCRIPT_BYTES, t0, ZKP_TRANSCRIPT_BYTES);
  memset(results, 7, N);
  rc = zkp_verify_batchable_each(ctx, st, N, ts, inst, NULL, coms, resp, NULL, 0, results);
  CHECK(rc == ZKP_TB_OK, "zkp_verify_batchable_each");
  for (j = 0; j < N; ++j) CHECK(results[j] == 0, "batchable proof verifies");
  for (j = 0; j < N; ++j) memcpy(ts + (size_t)j * ZKP_TRANSCRIPT_BYTES, t0, ZKP_TRANSCRIPT_BYTES);
  rc = zkp_batch_verify(ctx, st, N, N, ts, inst, NULL, coms, resp, NULL, 0);
  CHECK(rc == ZKP_TB_OK, "zkp_batch_verify accepts");

  /* ---- a tampered response: refused by every verifier; the batch verifier says which proof */
  resp[32 * (size_t)(N / 2)] ^= 1;
  for (j = 0; j < N; ++j) memcpy(ts + (size_t)j * ZKP_TRANSCRIPT_BYTES, t0, ZKP_TRANSCRIPT_BYTES);
  rc = zkp_verify_compact_batch(ctx, st, N, ts, inst, NULL, chal, resp, 0, results);
  CHECK(rc == ZKP_TB_OK, "zkp_verify_compact_batch (tampered)");
  for (j = 0; j < N; ++j) CHECK(results[j] == (j == N / 2 ? 1 : 0), "only the tampered compact proof is refused");
  for (j = 0; j < N; ++j) memcpy(ts + (size_t)j * ZKP_TRANSCRIPT_BYTES, t0, ZKP_TRANSCRIPT_BYTES);
  rc = zkp_batch_verify_locate(ctx, st, N, N, ts, inst, NULL, coms, resp, NULL, 0, results);
  CHECK(rc == ZKP_TB_VERIFICATION_FAILURE, "zkp_batch_verify_locate refuses the batch");
  for (j = 0; j < N; ++j) CHECK(results[j] == (j == N / 2 ? 1 : 0), "... and names the tampered proof");
  /* a batch whose sizes disagree is BatchSizeMismatch (batch_verifier.rs:72-74), not a verdict */
  if (N > 1) CHECK(zkp_batch_verify(ctx, st, N, N - 1, ts, inst, NULL, coms, resp, NULL, 0) == ZKP_TB_BATCH_SIZE_MISMATCH, "batch size mismatch");

  printf("all checks passed (%s, N = %u)\n", use_gpu ? "GPU 0" : "host backend", (unsigned)N);
  free(ts); free(secrets); free(inst); free(entropy); free(chal); free(resp); free(coms); free(results);
  zkp_statement_free(st);
  if (ctx) zkp_ctx_destroy(ctx);
  return 0;
}
