// Host-side Fiat-Shamir transcripts: Keccak-f[1600] -> STROBE-128 -> Merlin, plus the zkp
// TranscriptProtocol framing.  The reference keeps this on the host too (src/toolbox/mod.rs:165-228 over
// the `merlin = "2"` crate, Cargo.toml:21); the wire format must match byte for byte because it decides
// every challenge.  Transcripts are small value types (203 bytes) so that a batch can clone a shared
// prefix (labels common to all proofs of a statement) instead of re-hashing it per proof.
#pragma once
#include <cstddef>
#include <cstdint>
#include <cstring>
#include <string>

namespace zkp::host {

void keccak_f1600(uint64_t lanes[25]);

class Strobe128 {
 public:
  struct uninitialized_t {};
  explicit Strobe128(uninitialized_t) {}
  explicit Strobe128(const char* protocol_label);
  void meta_ad(const void* data, size_t len, bool more);
  void ad(const void* data, size_t len, bool more);
  void prf(void* out, size_t len, bool more);
  void key(const void* data, size_t len, bool more);

 private:
  static constexpr unsigned kRate = 166;
  enum : uint8_t { kI = 1, kA = 2, kC = 4, kT = 8, kM = 16, kK = 32 };
  void begin_op(uint8_t flags, bool more);
  void absorb(const uint8_t* d, size_t n);
  void overwrite(const uint8_t* d, size_t n);
  void squeeze(uint8_t* d, size_t n);
  void run_f();
  union { uint64_t lanes[25]; uint8_t bytes[200]; } st_;
  uint8_t pos_ = 0, pos_begin_ = 0, cur_flags_ = 0;
};

// merlin::TranscriptRng as used by Prover::prove_impl (prover.rs:78-89)
class TranscriptRng {
 public:
  explicit TranscriptRng(const Strobe128& s) : strobe_(s) {}
  void rekey_with_witness_bytes(const char* label, const void* witness, size_t len);
  void finalize(const uint8_t entropy[32]);      // the 32 bytes the external RNG contributes
  void fill_bytes(void* out, size_t len);

 private:
  Strobe128 strobe_;
};

class Transcript {
 public:
  Transcript(const void* label, size_t len);                       // merlin::Transcript::new
  explicit Transcript(const std::string& label) : Transcript(label.data(), label.size()) {}
  // plain-old-data (de)serialisation: Clone == memcpy (ZKP_TRANSCRIPT_BYTES in include/zkp_toolbox.h)
  // 203 live bytes (200 of STROBE state, pos, pos_begin, cur_flags) + 5 bytes of padding, which are written as zeros and
  // never read: blobs of equal transcripts are equal byte for byte, and no stack garbage reaches caller buffers
  static constexpr size_t kLiveBytes = 203;
  static Transcript from_bytes(const uint8_t* blob) {
    Transcript t{Strobe128::uninitialized_t{}};
    std::memcpy(static_cast<void*>(&t), blob, kLiveBytes);
    return t;
  }
  void to_bytes(uint8_t* blob) const {
    std::memcpy(blob, static_cast<const void*>(this), kLiveBytes);
    std::memset(blob + kLiveBytes, 0, sizeof(Transcript) - kLiveBytes);
  }
  void append_message(const char* label, const void* msg, size_t len);
  void challenge_bytes(const char* label, void* out, size_t len);
  TranscriptRng build_rng() const { return TranscriptRng(strobe_); }

  // ---- TranscriptProtocol (src/toolbox/mod.rs:165-228) ----
  void domain_sep(const char* label);                                            // :166-169
  void append_scalar_var(const char* label);                                     // :171-173
  void append_point_var(const char* label, const uint8_t enc[32]);               // :175-184 (encoding supplied)
  bool validate_and_append_point_var(const char* label, const uint8_t enc[32]);  // :186-197  false = identity -> reject
  void append_blinding_commitment(const char* label, const uint8_t enc[32]);     // :199-208
  bool validate_and_append_blinding_commitment(const char* label, const uint8_t enc[32]);   // :210-221
  void get_challenge(const char* label, uint8_t out_scalar[32]);                 // :223-227 (64 bytes mod l)

 private:
  explicit Transcript(Strobe128::uninitialized_t u) : strobe_(u) {}
  Strobe128 strobe_;
};
static_assert(sizeof(Transcript) == 208, "ZKP_TRANSCRIPT_BYTES");

}  // namespace zkp::host
