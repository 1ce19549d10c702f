// Keccak-f[1600] / STROBE-128 / Merlin / TranscriptProtocol -- see merlin.hpp.
#include "merlin.hpp"
#include "scalar.hpp"

#include <cstdio>
#include <cstdlib>

namespace zkp::host {

namespace {
constexpr uint64_t kRoundConstants[24] = {
    0x0000000000000001ULL, 0x0000000000008082ULL, 0x800000000000808aULL, 0x8000000080008000ULL, 0x000000000000808bULL,
    0x0000000080000001ULL, 0x8000000080008081ULL, 0x8000000000008009ULL, 0x000000000000008aULL, 0x0000000000000088ULL,
    0x0000000080008009ULL, 0x000000008000000aULL, 0x000000008000808bULL, 0x800000000000008bULL, 0x8000000000008089ULL,
    0x8000000000008003ULL, 0x8000000000008002ULL, 0x8000000000000080ULL, 0x000000000000800aULL, 0x800000008000000aULL,
    0x8000000080008081ULL, 0x8000000000008080ULL, 0x0000000080000001ULL, 0x8000000080008008ULL};
}  // namespace

// Keccak-f[1600] on 25 named lanes (A[x + 5y]), rho/pi written out with constant rotation amounts so the compiler
// keeps the state in registers: this is the host's hot loop once the MSMs live on the GPU (59 permutations per CMZ proof).
#define KROT(v, n) (((v) << (n)) | ((v) >> (64 - (n))))
void keccak_f1600(uint64_t A[25]) {
  uint64_t a00 = A[0], a10 = A[1], a20 = A[2], a30 = A[3], a40 = A[4];
  uint64_t a01 = A[5], a11 = A[6], a21 = A[7], a31 = A[8], a41 = A[9];
  uint64_t a02 = A[10], a12 = A[11], a22 = A[12], a32 = A[13], a42 = A[14];
  uint64_t a03 = A[15], a13 = A[16], a23 = A[17], a33 = A[18], a43 = A[19];
  uint64_t a04 = A[20], a14 = A[21], a24 = A[22], a34 = A[23], a44 = A[24];
  for (int round = 0; round < 24; ++round) {
    // theta
    const uint64_t c0 = a00 ^ a01 ^ a02 ^ a03 ^ a04, c1 = a10 ^ a11 ^ a12 ^ a13 ^ a14, c2 = a20 ^ a21 ^ a22 ^ a23 ^ a24,
                   c3 = a30 ^ a31 ^ a32 ^ a33 ^ a34, c4 = a40 ^ a41 ^ a42 ^ a43 ^ a44;
    const uint64_t d0 = c4 ^ KROT(c1, 1), d1 = c0 ^ KROT(c2, 1), d2 = c1 ^ KROT(c3, 1), d3 = c2 ^ KROT(c4, 1), d4 = c3 ^ KROT(c0, 1);
    a00 ^= d0; a01 ^= d0; a02 ^= d0; a03 ^= d0; a04 ^= d0;
    a10 ^= d1; a11 ^= d1; a12 ^= d1; a13 ^= d1; a14 ^= d1;
    a20 ^= d2; a21 ^= d2; a22 ^= d2; a23 ^= d2; a24 ^= d2;
    a30 ^= d3; a31 ^= d3; a32 ^= d3; a33 ^= d3; a34 ^= d3;
    a40 ^= d4; a41 ^= d4; a42 ^= d4; a43 ^= d4; a44 ^= d4;
    // rho + pi:  B[y][2x+3y] = rot(A[x][y], r[x][y])
    const uint64_t b00 = a00,           b13 = KROT(a01, 36), b21 = KROT(a02, 3),  b34 = KROT(a03, 41), b42 = KROT(a04, 18);
    const uint64_t b02 = KROT(a10, 1),  b10 = KROT(a11, 44), b23 = KROT(a12, 10), b31 = KROT(a13, 45), b44 = KROT(a14, 2);
    const uint64_t b04 = KROT(a20, 62), b12 = KROT(a21, 6),  b20 = KROT(a22, 43), b33 = KROT(a23, 15), b41 = KROT(a24, 61);
    const uint64_t b01 = KROT(a30, 28), b14 = KROT(a31, 55), b22 = KROT(a32, 25), b30 = KROT(a33, 21), b43 = KROT(a34, 56);
    const uint64_t b03 = KROT(a40, 27), b11 = KROT(a41, 20), b24 = KROT(a42, 39), b32 = KROT(a43, 8),  b40 = KROT(a44, 14);
    // chi
    a00 = b00 ^ (~b10 & b20); a10 = b10 ^ (~b20 & b30); a20 = b20 ^ (~b30 & b40); a30 = b30 ^ (~b40 & b00); a40 = b40 ^ (~b00 & b10);
    a01 = b01 ^ (~b11 & b21); a11 = b11 ^ (~b21 & b31); a21 = b21 ^ (~b31 & b41); a31 = b31 ^ (~b41 & b01); a41 = b41 ^ (~b01 & b11);
    a02 = b02 ^ (~b12 & b22); a12 = b12 ^ (~b22 & b32); a22 = b22 ^ (~b32 & b42); a32 = b32 ^ (~b42 & b02); a42 = b42 ^ (~b02 & b12);
    a03 = b03 ^ (~b13 & b23); a13 = b13 ^ (~b23 & b33); a23 = b23 ^ (~b33 & b43); a33 = b33 ^ (~b43 & b03); a43 = b43 ^ (~b03 & b13);
    a04 = b04 ^ (~b14 & b24); a14 = b14 ^ (~b24 & b34); a24 = b24 ^ (~b34 & b44); a34 = b34 ^ (~b44 & b04); a44 = b44 ^ (~b04 & b14);
    // iota
    a00 ^= kRoundConstants[round];
  }
  A[0] = a00; A[1] = a10; A[2] = a20; A[3] = a30; A[4] = a40;
  A[5] = a01; A[6] = a11; A[7] = a21; A[8] = a31; A[9] = a41;
  A[10] = a02; A[11] = a12; A[12] = a22; A[13] = a32; A[14] = a42;
  A[15] = a03; A[16] = a13; A[17] = a23; A[18] = a33; A[19] = a43;
  A[20] = a04; A[21] = a14; A[22] = a24; A[23] = a34; A[24] = a44;
}
#undef KROT

// ---- STROBE-128 (v1.0.2), exactly the subset Merlin uses -------------------------------------------
Strobe128::Strobe128(const char* protocol_label) {
  std::memset(&st_, 0, sizeof(st_));
  const uint8_t init[6] = {1, kRate + 2, 1, 0, 1, 96};
  std::memcpy(st_.bytes, init, 6);
  std::memcpy(st_.bytes + 6, "STROBEv1.0.2", 12);
  keccak_f1600(st_.lanes);
  meta_ad(protocol_label, std::strlen(protocol_label), false);
}
void Strobe128::run_f() {
  st_.bytes[pos_] ^= pos_begin_;
  st_.bytes[pos_ + 1] ^= 0x04;
  st_.bytes[kRate + 1] ^= 0x80;
  keccak_f1600(st_.lanes);
  pos_ = 0;
  pos_begin_ = 0;
}
void Strobe128::absorb(const uint8_t* d, size_t n) {
  for (size_t i = 0; i < n; ++i) {
    st_.bytes[pos_++] ^= d[i];
    if (pos_ == kRate) run_f();
  }
}
void Strobe128::overwrite(const uint8_t* d, size_t n) {
  for (size_t i = 0; i < n; ++i) {
    st_.bytes[pos_++] = d[i];
    if (pos_ == kRate) run_f();
  }
}
void Strobe128::squeeze(uint8_t* d, size_t n) {
  for (size_t i = 0; i < n; ++i) {
    d[i] = st_.bytes[pos_];
    st_.bytes[pos_++] = 0;
    if (pos_ == kRate) run_f();
  }
}
void Strobe128::begin_op(uint8_t flags, bool more) {
  if (more) return;                         // continuation of the previous operation (same flags)
  const uint8_t old_begin = pos_begin_;
  pos_begin_ = (uint8_t)(pos_ + 1);
  cur_flags_ = flags;
  const uint8_t hdr[2] = {old_begin, flags};
  absorb(hdr, 2);
  if ((flags & (kC | kK)) && pos_ != 0) run_f();
}
void Strobe128::meta_ad(const void* d, size_t n, bool more) { begin_op(kM | kA, more); absorb(static_cast<const uint8_t*>(d), n); }
void Strobe128::ad(const void* d, size_t n, bool more) { begin_op(kA, more); absorb(static_cast<const uint8_t*>(d), n); }
void Strobe128::prf(void* out, size_t n, bool more) { begin_op(kI | kA | kC, more); squeeze(static_cast<uint8_t*>(out), n); }
void Strobe128::key(const void* d, size_t n, bool more) { begin_op(kA | kC, more); overwrite(static_cast<const uint8_t*>(d), n); }

static inline void u32le(uint8_t out[4], size_t n) {
  out[0] = (uint8_t)n; out[1] = (uint8_t)(n >> 8); out[2] = (uint8_t)(n >> 16); out[3] = (uint8_t)(n >> 24);
}

// ---- Merlin ------------------------------------------------------------------------------------------
// ZKP_DEBUG_TRANSCRIPT=1 in the environment: every Merlin operation of the HOST transcripts is written to stderr (label,
// length, leading bytes) -- what the reference's `debug-transcript` feature (Cargo.toml:35, merlin's own) prints.  A
// challenge mismatch against the Rust crate is then found by diffing the two op logs.
static bool debug_transcript() {
  static const bool on = [] { const char* e = std::getenv("ZKP_DEBUG_TRANSCRIPT"); return e && *e && *e != '0'; }();
  return on;
}
static void debug_op(const char* what, const char* label, const void* data, size_t len) {
  std::fprintf(stderr, "[merlin] %-9s label=\"%s\" len=%zu", what, label, len);
  const uint8_t* p = static_cast<const uint8_t*>(data);
  if (p && len) {
    std::fprintf(stderr, " data=");
    for (size_t i = 0; i < len && i < 32; ++i) std::fprintf(stderr, "%02x", p[i]);
    if (len > 32) std::fprintf(stderr, "..");
  }
  std::fputc('\n', stderr);
}

Transcript::Transcript(const void* label, size_t len) : strobe_("Merlin v1.0") { append_message("dom-sep", label, len); }
void Transcript::append_message(const char* label, const void* msg, size_t len) {
  if (debug_transcript()) debug_op("append", label, msg, len);
  uint8_t l[4];
  u32le(l, len);
  strobe_.meta_ad(label, std::strlen(label), false);
  strobe_.meta_ad(l, 4, true);
  strobe_.ad(msg, len, false);
}
void Transcript::challenge_bytes(const char* label, void* out, size_t len) {
  uint8_t l[4];
  u32le(l, len);
  strobe_.meta_ad(label, std::strlen(label), false);
  strobe_.meta_ad(l, 4, true);
  strobe_.prf(out, len, false);
  if (debug_transcript()) debug_op("challenge", label, out, len);
}
void TranscriptRng::rekey_with_witness_bytes(const char* label, const void* w, size_t len) {
  if (debug_transcript()) debug_op("rng-rekey", label, nullptr, len);      // (witness bytes are never printed)
  uint8_t l[4];
  u32le(l, len);
  strobe_.meta_ad(label, std::strlen(label), false);
  strobe_.meta_ad(l, 4, true);
  strobe_.key(w, len, false);
}
void TranscriptRng::finalize(const uint8_t entropy[32]) {
  strobe_.meta_ad("rng", 3, false);
  strobe_.key(entropy, 32, false);
}
void TranscriptRng::fill_bytes(void* out, size_t len) {
  uint8_t l[4];
  u32le(l, len);
  strobe_.meta_ad(l, 4, false);
  strobe_.prf(out, len, false);
}

// ---- TranscriptProtocol ------------------------------------------------------------------------------
static const uint8_t kIdentity[32] = {0};

void Transcript::domain_sep(const char* label) {
  append_message("dom-sep", "schnorrzkp/1.0/ristretto255", 27);
  append_message("dom-sep", label, std::strlen(label));
}
void Transcript::append_scalar_var(const char* label) { append_message("scvar", label, std::strlen(label)); }
void Transcript::append_point_var(const char* label, const uint8_t enc[32]) {
  append_message("ptvar", label, std::strlen(label));
  append_message("val", enc, 32);
}
bool Transcript::validate_and_append_point_var(const char* label, const uint8_t enc[32]) {
  if (std::memcmp(enc, kIdentity, 32) == 0) return false;
  append_point_var(label, enc);
  return true;
}
void Transcript::append_blinding_commitment(const char* label, const uint8_t enc[32]) {
  append_message("blindcom", label, std::strlen(label));
  append_message("val", enc, 32);
}
bool Transcript::validate_and_append_blinding_commitment(const char* label, const uint8_t enc[32]) {
  if (std::memcmp(enc, kIdentity, 32) == 0) return false;
  append_blinding_commitment(label, enc);
  return true;
}
void Transcript::get_challenge(const char* label, uint8_t out_scalar[32]) {
  uint8_t wide[64];
  challenge_bytes(label, wide, 64);
  Scalar::from_bytes_mod_order_wide(wide).to_bytes(out_scalar);
}

}  // namespace zkp::host
