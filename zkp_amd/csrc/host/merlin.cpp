// Keccak-f[1600] / STROBE-128 / Merlin / TranscriptProtocol -- see merlin.hpp.
#include "merlin.hpp"
#include "scalar.hpp"

namespace zkp::host {

namespace {
constexpr uint64_t kRoundConstants[24] = {
    0x0000000000000001ULL, 0x0000000000008082ULL, 0x800000000000808aULL, 0x8000000080008000ULL, 0x000000000000808bULL,
    0x0000000080000001ULL, 0x8000000080008081ULL, 0x8000000000008009ULL, 0x000000000000008aULL, 0x0000000000000088ULL,
    0x0000000080008009ULL, 0x000000008000000aULL, 0x000000008000808bULL, 0x800000000000008bULL, 0x8000000000008089ULL,
    0x8000000000008003ULL, 0x8000000000008002ULL, 0x8000000000000080ULL, 0x000000000000800aULL, 0x800000008000000aULL,
    0x8000000080008081ULL, 0x8000000000008080ULL, 0x0000000080000001ULL, 0x8000000080008008ULL};
inline uint64_t rotl(uint64_t x, unsigned n) { return n ? (x << n) | (x >> (64 - n)) : x; }
}  // namespace

// Straightforward theta / rho+pi / chi / iota on a 5x5 lane matrix A[x + 5y]; rho offsets are generated from
// the (x, y) -> (y, 2x + 3y) walk of the specification rather than tabulated.
void keccak_f1600(uint64_t A[25]) {
  for (int round = 0; round < 24; ++round) {
    uint64_t Cx[5], D[5];
    for (int x = 0; x < 5; ++x) Cx[x] = A[x] ^ A[x + 5] ^ A[x + 10] ^ A[x + 15] ^ A[x + 20];
    for (int x = 0; x < 5; ++x) D[x] = Cx[(x + 4) % 5] ^ rotl(Cx[(x + 1) % 5], 1);
    for (int i = 0; i < 25; ++i) A[i] ^= D[i % 5];
    uint64_t B[25];
    B[0] = A[0];
    int x = 1, y = 0;
    for (int t = 0; t < 24; ++t) {
      const unsigned r = (unsigned)(((t + 1) * (t + 2) / 2) % 64);
      const int nx = y, ny = (2 * x + 3 * y) % 5;
      B[nx + 5 * ny] = rotl(A[x + 5 * y], r);
      x = nx;
      y = ny;
    }
    for (int yy = 0; yy < 25; yy += 5)
      for (int xx = 0; xx < 5; ++xx) A[yy + xx] = B[yy + xx] ^ (~B[yy + (xx + 1) % 5] & B[yy + (xx + 2) % 5]);
    A[0] ^= kRoundConstants[round];
  }
}

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
Transcript::Transcript(const void* label, size_t len) : strobe_("Merlin v1.0") { append_message("dom-sep", label, len); }
void Transcript::append_message(const char* label, const void* msg, size_t len) {
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
}
void TranscriptRng::rekey_with_witness_bytes(const char* label, const void* w, size_t len) {
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
