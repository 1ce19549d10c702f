// Host execution of the engine's entry points for TINY calls ("a CPU backend implementing the same
// header"; BASELINE configs[0]: "DLEQ proof ... single prove + verify on CPU (plumbing, no GPU)").
//
// A single DLEQ verification through the GPU is a chain of ~10 kernel launches, ~1 ms whatever the size; the reference needs ~0.1 ms
// on one core (benches/dleq.rs:49-90).  For calls below a handful of terms the host toolbox therefore runs the group arithmetic HERE:
// the very header the kernels' point arithmetic is compiled from -- ge25519.h (extended coordinates, ristretto255 codec) -- compiled for
// the host by g++ over a field of 5 x 51-bit limbs with the same interface as the device's 9 x 29-bit one (host/fe51.h: a 64-bit core
// needs 25 products per multiplication instead of 98) -- behind the same zkp_toolbox.h calls, and also when no GPU context is given at all
// (ctx == NULL).  Same bytes as the device path (outputs are canonical encodings), checked by tests/test_host_backend.py against the
// oracle and by tests/test_gpu_toolbox.py against the GPU route.  This is product code: nothing under oracle/ is involved.
//
// Algorithms (host/host_backend.hpp has the interface):
//   msm_many      per MSM a Straus walk over signed radix-16 digits, table {1..8} P per distinct point of the MSM.  flags = ZKP_CT: the
//                 table entry is picked with masks over all eight entries and zero digits are added like any other (what
//                 curve25519-dalek's constant-time multiscalar_mul does, prover.rs:94); ZKP_VARTIME: indexed, zeros skipped.
//   msm_optional  the same walk over n points with decode-or-None (verifier.rs:162-166, batch_verifier.rs:219-228).
//   decode_check  ristretto decode.
#include "host_backend.hpp"

#include <cstring>
#include <vector>

#define ZKP_HOST_FE51 1        // this translation unit (and no other of the library) gets the 5 x 51-bit field under the device's point formulas
#include "../ge25519.h"

namespace zkp {
namespace hostbk {

namespace {

inline uint32_t decode(ge_p3& p, const uint8_t* enc) {
  uint32_t w[8];
  std::memcpy(w, enc, 32);
  return ristretto_decode(p, w);
}
inline void encode(uint8_t* out, const ge_p3& p) {
  uint32_t w[8];
  ristretto_encode(w, p);
  std::memcpy(out, w, 32);
}

// signed radix-16 digits of any 256-bit integer: s = sum d[i] 16^i, d[i] in [-8, 8) for i < 64, d[64] in {0, 1}
inline void recode16(int8_t d[65], const uint8_t s[32]) {
  int carry = 0;
  for (int i = 0; i < 64; ++i) {
    int v = ((s[i >> 1] >> (4 * (i & 1))) & 15) + carry;
    carry = v >= 8;
    d[i] = (int8_t)(v - 16 * carry);
  }
  d[64] = (int8_t)carry;
}

struct Table { ge_cached e[8]; };          // 1P .. 8P
inline void build_table(Table& t, const ge_p3& p) {
  ge_p3 acc = p;
  ge_to_cached(t.e[0], p);
  for (int k = 1; k < 8; ++k) {
    ge_add_cached(acc, acc, t.e[0]);
    ge_to_cached(t.e[k], acc);
  }
}
// acc += d * P with the table of P.  ct: every entry is touched, the sign is applied with a mask, d == 0 adds the neutral element
template <bool CT>
inline void add_digit(ge_p3& acc, const Table& t, int d) {
  if (!CT) {
    if (d == 0) return;
    if (d > 0) ge_add_cached(acc, acc, t.e[d - 1]);
    else ge_sub_cached(acc, acc, t.e[-d - 1]);
    return;
  }
  const uint32_t neg = (uint32_t)(d >> 7) & 1u;                   // d is an int8 value
  const uint32_t mag = (uint32_t)((d ^ -(int)neg) + (int)neg);      // |d| without a branch
  ge_cached sel;
  ge_cached_identity(sel);
  for (uint32_t k = 1; k <= 8; ++k) {
    const uint32_t x = mag ^ k;
    ge_cached_cmov(sel, t.e[k - 1], ((x - 1u) >> 31) & 1u);          // x == 0
  }
  ge_cached_cneg(sel, neg);
  ge_add_cached(acc, acc, sel);
}

template <bool CT>
void straus(ge_p3& acc, size_t k, const int8_t (*digits)[65], const Table* const* tables) {
  ge_identity(acc);
  for (int w = 64; w >= 0; --w) {
    if (w != 64) { ge_double<false>(acc, acc); ge_double<false>(acc, acc); ge_double<false>(acc, acc); ge_double<true>(acc, acc); }
    for (size_t i = 0; i < k; ++i) add_digit<CT>(acc, *tables[i], digits[i][w]);
  }
}

}  // namespace

int msm_many(uint32_t n_msm, const uint32_t* off, const uint8_t* scalars, const uint32_t* pidx, const uint8_t* points, uint32_t n_points, int flags,
             uint8_t* out, uint8_t* status) {
  if (n_msm == 0) return ZKP_OK;
  if (!off || !out || !status || off[0] != 0) return ZKP_ERR_ARG;
  const uint32_t T = off[n_msm];
  if (T && (!scalars || !pidx || !points)) return ZKP_ERR_ARG;
  // decode and tabulate each referenced point once
  std::vector<int32_t> slot(n_points, -1);
  std::vector<Table> tabs;
  std::vector<uint8_t> ok;
  for (uint32_t t = 0; t < T; ++t) {
    if (pidx[t] >= n_points) return ZKP_ERR_ARG;
    if (slot[pidx[t]] >= 0) continue;
    ge_p3 p;
    const uint32_t good = decode(p, points + 32 * (size_t)pidx[t]);
    slot[pidx[t]] = (int32_t)tabs.size();
    tabs.emplace_back();
    ok.push_back((uint8_t)good);
    if (good) build_table(tabs.back(), p);
  }
  std::vector<int8_t> dig;
  std::vector<const Table*> tp;
  for (uint32_t i = 0; i < n_msm; ++i) {
    if (off[i + 1] < off[i]) return ZKP_ERR_ARG;
    const size_t k = off[i + 1] - off[i];
    dig.resize(65 * (k ? k : 1));
    tp.resize(k);
    bool bad = false;
    for (size_t j = 0; j < k; ++j) {
      const uint32_t t = off[i] + (uint32_t)j;
      recode16(reinterpret_cast<int8_t(*)[65]>(dig.data())[j], scalars + 32 * (size_t)t);
      tp[j] = &tabs[slot[pidx[t]]];
      bad |= !ok[slot[pidx[t]]];
    }
    status[i] = bad ? 1 : 0;
    if (bad) { std::memset(out + 32 * (size_t)i, 0, 32); continue; }
    ge_p3 acc;
    if (flags == ZKP_CT) straus<true>(acc, k, reinterpret_cast<const int8_t(*)[65]>(dig.data()), tp.data());
    else straus<false>(acc, k, reinterpret_cast<const int8_t(*)[65]>(dig.data()), tp.data());
    encode(out + 32 * (size_t)i, acc);
  }
  return ZKP_OK;
}

int msm_optional(uint64_t n, const uint8_t* scalars, const uint8_t* points, uint8_t out_point[32], int* status) {
  if (!out_point || !status || (n && (!scalars || !points))) return ZKP_ERR_ARG;
  std::vector<Table> tabs(n);
  std::vector<int8_t> dig(65 * (n ? n : 1));
  std::vector<const Table*> tp(n);
  bool bad = false;
  for (uint64_t i = 0; i < n; ++i) {
    ge_p3 p;
    if (!decode(p, points + 32 * i)) { bad = true; break; }
    build_table(tabs[i], p);
    recode16(reinterpret_cast<int8_t(*)[65]>(dig.data())[i], scalars + 32 * i);
    tp[i] = &tabs[i];
  }
  std::memset(out_point, 0, 32);
  *status = bad ? 1 : 0;
  if (bad) return ZKP_OK;
  ge_p3 acc;
  straus<false>(acc, (size_t)n, reinterpret_cast<const int8_t(*)[65]>(dig.data()), tp.data());
  encode(out_point, acc);
  return ZKP_OK;
}

int decode_check(uint64_t n, const uint8_t* points, uint8_t* status) {
  if (n && (!points || !status)) return ZKP_ERR_ARG;
  for (uint64_t i = 0; i < n; ++i) {
    ge_p3 p;
    status[i] = decode(p, points + 32 * i) ? 0 : 1;
  }
  return ZKP_OK;
}

}  // namespace hostbk
}  // namespace zkp
