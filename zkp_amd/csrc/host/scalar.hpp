// Host-side scalars modulo l = 2^252 + 27742317777372353535851937790883648493 (the ristretto255 group
// order): the curve25519-dalek `Scalar` operations the toolbox performs between GPU calls
// (reference: mod.rs:226 from_bytes_mod_order_wide; prover.rs:108 s*c + b; verifier.rs:95,142 negation;
// verifier.rs:153-158 and batch_verifier.rs:179-204 random-weight products and accumulation).
// Header-only; Montgomery multiplication (R = 2^256) on four 64-bit limbs with unsigned __int128.
#pragma once
#include <cstdint>
#include <cstring>

namespace zkp::host {

class Scalar {
 public:
  Scalar() : w_{0, 0, 0, 0} {}
  static Scalar zero() { return Scalar(); }
  static Scalar from_u64(uint64_t x) { Scalar s; s.w_[0] = x; return s; }
  // Scalar::from(u128): the 16 little-endian bytes, zero extended (verifier.rs:153, batch_verifier.rs:179)
  static Scalar from_u128_le(const uint8_t b[16]) { Scalar s; std::memcpy(s.w_, b, 16); return s; }
  // canonical or not, reduced on entry
  static Scalar from_bytes_mod_order(const uint8_t b[32]) {
    Scalar s;
    std::memcpy(s.w_, b, 32);
    s.reduce_once_or_more();
    return s;
  }
  static Scalar from_bytes_mod_order_wide(const uint8_t b[64]) {
    Scalar lo, hi;
    std::memcpy(lo.w_, b, 32);
    std::memcpy(hi.w_, b + 32, 32);
    // lo + hi * 2^256 = mont(lo, R) + mont(hi, R^2)   (mont(a, b) = a b R^-1 mod l, R = 2^256)
    return mont(lo, R1()) + mont(hi, RR());
  }
  void to_bytes(uint8_t out[32]) const { std::memcpy(out, w_, 32); }
  bool is_zero() const { return (w_[0] | w_[1] | w_[2] | w_[3]) == 0; }
  bool operator==(const Scalar& o) const { return std::memcmp(w_, o.w_, 32) == 0; }
  bool operator!=(const Scalar& o) const { return !(*this == o); }

  Scalar operator+(const Scalar& o) const {
    Scalar r;
    unsigned __int128 c = 0;
    for (int i = 0; i < 4; ++i) { c += (unsigned __int128)w_[i] + o.w_[i]; r.w_[i] = (uint64_t)c; c >>= 64; }
    r.cond_sub_l();      // both < l < 2^253, so the sum is < 2^254: no carry out, one subtraction suffices
    return r;
  }
  Scalar operator-() const {
    if (is_zero()) return *this;
    Scalar r;
    unsigned __int128 br = 0;
    for (int i = 0; i < 4; ++i) {
      const unsigned __int128 t = (unsigned __int128)L(i) - w_[i] - (uint64_t)br;
      r.w_[i] = (uint64_t)t;
      br = (t >> 64) & 1;
    }
    return r;
  }
  Scalar operator-(const Scalar& o) const { return *this + (-o); }
  Scalar operator*(const Scalar& o) const { return mont(mont(*this, o), RR()); }
  Scalar& operator+=(const Scalar& o) { *this = *this + o; return *this; }
  Scalar& operator-=(const Scalar& o) { *this = *this - o; return *this; }

 private:
  uint64_t w_[4];

  static constexpr uint64_t L(int i) {
    return i == 0 ? 0x5812631a5cf5d3edULL : i == 1 ? 0x14def9dea2f79cd6ULL : i == 2 ? 0ULL : 0x1000000000000000ULL;
  }
  // -l^-1 mod 2^64, by Newton iteration on the low limb
  static constexpr uint64_t n0inv() {
    uint64_t x = 1;
    for (int i = 0; i < 7; ++i) x *= 2 - L(0) * x;
    return (uint64_t)0 - x;
  }
  bool geq_l() const {
    for (int i = 3; i >= 0; --i) { if (w_[i] > L(i)) return true; if (w_[i] < L(i)) return false; }
    return true;
  }
  void sub_l() {
    unsigned __int128 br = 0;
    for (int i = 0; i < 4; ++i) {
      const unsigned __int128 t = (unsigned __int128)w_[i] - L(i) - (uint64_t)br;
      w_[i] = (uint64_t)t;
      br = (t >> 64) & 1;
    }
  }
  void cond_sub_l() { if (geq_l()) sub_l(); }
  void reduce_once_or_more() { while (geq_l()) sub_l(); }     // input < 2^256 < 16 l

  // CIOS Montgomery product a * b * 2^-256 mod l.  Requires b < l (a may be any 256-bit value).
  static Scalar mont(const Scalar& a, const Scalar& b) {
    uint64_t t[6] = {0, 0, 0, 0, 0, 0};
    for (int i = 0; i < 4; ++i) {
      unsigned __int128 c = 0;
      for (int j = 0; j < 4; ++j) { c += (unsigned __int128)a.w_[i] * b.w_[j] + t[j]; t[j] = (uint64_t)c; c >>= 64; }
      c += t[4]; t[4] = (uint64_t)c; t[5] = (uint64_t)(c >> 64);
      const uint64_t m = t[0] * n0inv();
      c = (unsigned __int128)m * L(0) + t[0];
      c >>= 64;
      for (int j = 1; j < 4; ++j) { c += (unsigned __int128)m * L(j) + t[j]; t[j - 1] = (uint64_t)c; c >>= 64; }
      c += t[4]; t[3] = (uint64_t)c; t[4] = t[5] + (uint64_t)(c >> 64);
    }
    Scalar r;
    std::memcpy(r.w_, t, 32);
    // result < 2 l when b < l ... a < 2^256 gives < (2^256 l + l 2^256)/2^256 = 2 l;  t[4] == 0 here
    r.reduce_once_or_more();
    return r;
  }
  // R mod l and R^2 mod l with R = 2^256, computed once by repeated doubling
  static Scalar pow2(int k) {
    Scalar x = from_u64(1);
    for (int i = 0; i < k; ++i) x = x + x;
    return x;
  }
  static const Scalar& R1() { static const Scalar v = pow2(256); return v; }
  static const Scalar& RR() { static const Scalar v = pow2(512); return v; }
};

}  // namespace zkp::host
