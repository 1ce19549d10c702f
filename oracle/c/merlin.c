/* ORACLE (test infrastructure): Keccak-f[1600], STROBE-128 and Merlin transcripts as used by the
 * reference through `merlin = "2"` (Cargo.toml:21, not vendored; call sites src/toolbox/mod.rs:165-228,
 * prover.rs:78-89).  Restated from the STROBE v1.0.2 / Merlin v1.0 specifications; pinned by Merlin's
 * published test vector in tests/test_oracle_c.py. */
#include <string.h>
#include "oracle.h"

static uint64_t g_keccak_count = 0;
uint64_t orc_keccak_count(void) { return g_keccak_count; }

static const uint64_t KRC[24] = {
  0x0000000000000001ULL, 0x0000000000008082ULL, 0x800000000000808aULL, 0x8000000080008000ULL,
  0x000000000000808bULL, 0x0000000080000001ULL, 0x8000000080008081ULL, 0x8000000000008009ULL,
  0x000000000000008aULL, 0x0000000000000088ULL, 0x0000000080008009ULL, 0x000000008000000aULL,
  0x000000008000808bULL, 0x800000000000008bULL, 0x8000000000008089ULL, 0x8000000000008003ULL,
  0x8000000000008002ULL, 0x8000000000000080ULL, 0x000000000000800aULL, 0x800000008000000aULL,
  0x8000000080008081ULL, 0x8000000000008080ULL, 0x0000000080000001ULL, 0x8000000080008008ULL};
static const int KROT[24] = {1, 3, 6, 10, 15, 21, 28, 36, 45, 55, 2, 14, 27, 41, 56, 8, 25, 43, 62, 18, 39, 61, 20, 44};
static const int KPIL[24] = {10, 7, 11, 17, 18, 3, 5, 16, 8, 21, 24, 4, 15, 23, 19, 13, 12, 2, 20, 14, 22, 9, 6, 1};
#define ROL64(x, n) (((x) << (n)) | ((x) >> (64 - (n))))

static void keccak_f1600(uint8_t st8[200]) {
  uint64_t st[25], bc[5], t;
  memcpy(st, st8, 200);
  ++g_keccak_count;
  for (int round = 0; round < 24; ++round) {
    for (int i = 0; i < 5; ++i) bc[i] = st[i] ^ st[i + 5] ^ st[i + 10] ^ st[i + 15] ^ st[i + 20];
    for (int i = 0; i < 5; ++i) {
      t = bc[(i + 4) % 5] ^ ROL64(bc[(i + 1) % 5], 1);
      for (int j = 0; j < 25; j += 5) st[j + i] ^= t;
    }
    t = st[1];
    for (int i = 0; i < 24; ++i) {
      const int j = KPIL[i];
      const uint64_t b = st[j];
      st[j] = ROL64(t, KROT[i]);
      t = b;
    }
    for (int j = 0; j < 25; j += 5) {
      for (int i = 0; i < 5; ++i) bc[i] = st[j + i];
      for (int i = 0; i < 5; ++i) st[j + i] ^= (~bc[(i + 1) % 5]) & bc[(i + 2) % 5];
    }
    st[0] ^= KRC[round];
  }
  memcpy(st8, st, 200);
}

#define STROBE_R 166
enum { FLAG_I = 1, FLAG_A = 2, FLAG_C = 4, FLAG_T = 8, FLAG_M = 16, FLAG_K = 32 };

static void strobe_run_f(orc_strobe* s) {
  s->st[s->pos] ^= s->pos_begin;
  s->st[s->pos + 1] ^= 0x04;
  s->st[STROBE_R + 1] ^= 0x80;
  keccak_f1600(s->st);
  s->pos = 0;
  s->pos_begin = 0;
}
static void strobe_absorb(orc_strobe* s, const uint8_t* d, size_t n) {
  for (size_t i = 0; i < n; ++i) { s->st[s->pos++] ^= d[i]; if (s->pos == STROBE_R) strobe_run_f(s); }
}
static void strobe_overwrite(orc_strobe* s, const uint8_t* d, size_t n) {
  for (size_t i = 0; i < n; ++i) { s->st[s->pos++] = d[i]; if (s->pos == STROBE_R) strobe_run_f(s); }
}
static void strobe_squeeze(orc_strobe* s, uint8_t* d, size_t n) {
  for (size_t i = 0; i < n; ++i) { d[i] = s->st[s->pos]; s->st[s->pos++] = 0; if (s->pos == STROBE_R) strobe_run_f(s); }
}
static void strobe_begin_op(orc_strobe* s, uint8_t flags, int more) {
  if (more) return;
  const uint8_t old_begin = s->pos_begin;
  s->pos_begin = (uint8_t)(s->pos + 1);
  s->cur_flags = flags;
  const uint8_t hdr[2] = {old_begin, flags};
  strobe_absorb(s, hdr, 2);
  if ((flags & (FLAG_C | FLAG_K)) && s->pos != 0) strobe_run_f(s);
}
static void strobe_meta_ad(orc_strobe* s, const uint8_t* d, size_t n, int more) { strobe_begin_op(s, FLAG_M | FLAG_A, more); strobe_absorb(s, d, n); }
static void strobe_ad(orc_strobe* s, const uint8_t* d, size_t n, int more) { strobe_begin_op(s, FLAG_A, more); strobe_absorb(s, d, n); }
static void strobe_prf(orc_strobe* s, uint8_t* d, size_t n, int more) { strobe_begin_op(s, FLAG_I | FLAG_A | FLAG_C, more); strobe_squeeze(s, d, n); }
static void strobe_key(orc_strobe* s, const uint8_t* d, size_t n, int more) { strobe_begin_op(s, FLAG_A | FLAG_C, more); strobe_overwrite(s, d, n); }

static void strobe_init(orc_strobe* s, const char* proto) {
  memset(s, 0, sizeof(*s));
  const uint8_t hdr[6] = {1, STROBE_R + 2, 1, 0, 1, 96};
  memcpy(s->st, hdr, 6);
  memcpy(s->st + 6, "STROBEv1.0.2", 12);
  keccak_f1600(s->st);
  strobe_meta_ad(s, (const uint8_t*)proto, strlen(proto), 0);
}

static void le32(uint8_t b[4], size_t n) { b[0] = (uint8_t)n; b[1] = (uint8_t)(n >> 8); b[2] = (uint8_t)(n >> 16); b[3] = (uint8_t)(n >> 24); }

void orc_transcript_append(orc_transcript* t, const char* label, const uint8_t* msg, size_t len) {
  uint8_t l[4];
  le32(l, len);
  strobe_meta_ad(&t->s, (const uint8_t*)label, strlen(label), 0);
  strobe_meta_ad(&t->s, l, 4, 1);
  strobe_ad(&t->s, msg, len, 0);
}
void orc_transcript_init(orc_transcript* t, const uint8_t* label, size_t len) {
  strobe_init(&t->s, "Merlin v1.0");
  orc_transcript_append(t, "dom-sep", label, len);
}
void orc_transcript_challenge(orc_transcript* t, const char* label, uint8_t* out, size_t len) {
  uint8_t l[4];
  le32(l, len);
  strobe_meta_ad(&t->s, (const uint8_t*)label, strlen(label), 0);
  strobe_meta_ad(&t->s, l, 4, 1);
  strobe_prf(&t->s, out, len, 0);
}

/* TranscriptRng (merlin 2.x transcript.rs): build_rng / rekey_with_witness_bytes / finalize / fill_bytes */
static void rng_rekey(orc_strobe* s, const char* label, const uint8_t* w, size_t n) {
  uint8_t l[4];
  le32(l, n);
  strobe_meta_ad(s, (const uint8_t*)label, strlen(label), 0);
  strobe_meta_ad(s, l, 4, 1);
  strobe_key(s, w, n, 0);
}
static void rng_finalize(orc_strobe* s, const uint8_t entropy[32]) {
  strobe_meta_ad(s, (const uint8_t*)"rng", 3, 0);
  strobe_key(s, entropy, 32, 0);
}
static void rng_fill(orc_strobe* s, uint8_t* out, size_t n) {
  uint8_t l[4];
  le32(l, n);
  strobe_meta_ad(s, l, 4, 0);
  strobe_prf(s, out, n, 0);
}
