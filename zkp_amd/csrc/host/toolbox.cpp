// Host side of the zkp toolbox on top of the MI355X engine: the Prover / Verifier / BatchVerifier flows
// of the reference (src/toolbox/{prover,verifier,batch_verifier}.rs), restructured as
//      phase A (host, all proofs, threaded)  ->  ONE GPU call  ->  phase B (host, all proofs, threaded)
// because the reference's one-MSM-per-constraint call pattern (prover.rs:93-97, verifier.rs:96-106) would
// be pure launch overhead on a GPU.  Transcript traffic is byte-identical to the reference's, so proofs
// are interchangeable with proofs made by the Rust crate.  All group arithmetic goes through
// include/zkp_mi355x.h; there is no CPU fallback here.
#include <sys/random.h>

#include <cerrno>

#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <cstring>
#include <string>
#include <thread>
#include <utility>
#include <vector>

#include "merlin.hpp"
#include "scalar.hpp"

using zkp::host::Scalar;
using zkp::host::Transcript;

#include "host_backend.hpp"
#include "toolbox_internal.hpp"

using namespace zkp::host;

static std::atomic<uint32_t> g_fused_min_batch{32};
// calls whose group arithmetic is at most this many (scalar, point) terms run on the host cores (host_backend.cpp) instead of paying the
// ~1 ms launch chain of the GPU; a NULL context always does (no GPU needed at all)
static std::atomic<uint32_t> g_host_max_terms{16};

namespace {
inline bool on_host(const zkp_ctx* ctx, uint64_t terms) { return !ctx || terms <= g_host_max_terms.load(); }
// the engine entry points of the host-transcript route, on the GPU or -- tiny calls, or no context -- on the host
int be_msm_many(zkp_ctx* ctx, uint32_t n_msm, const uint32_t* off, const uint8_t* scalars, const uint32_t* pidx, const uint8_t* points, uint32_t n_points, int flags,
                uint8_t* out, uint8_t* status) {
  if (on_host(ctx, n_msm ? off[n_msm] : 0)) return zkp::hostbk::msm_many(n_msm, off, scalars, pidx, points, n_points, flags, out, status);
  return zkp_msm_many(ctx, n_msm, off, scalars, pidx, points, n_points, flags, out, status);
}
int be_decode_check(zkp_ctx* ctx, bool host, uint64_t n, const uint8_t* points, uint8_t* status) {
  if (host) return zkp::hostbk::decode_check(n, points, status);
  return zkp_decode_check(ctx, n, points, status, nullptr);
}
}  // namespace

namespace zkp {
namespace host {

// false = the operating system gave no entropy (getrandom failed with anything but EINTR): callers fail closed
bool os_entropy(uint8_t* out, size_t len) {
  size_t got = 0;
  while (got < len) {
    const ssize_t r = getrandom(out + got, len - got, 0);
    if (r > 0) got += (size_t)r;
    else if (r < 0 && errno == EINTR) continue;
    else return false;
  }
  return true;
}

// ChaCha20 block function (RFC 8439 section 2.3), used as the stream generator below
inline uint32_t rotl32(uint32_t v, int n) { return (v << n) | (v >> (32 - n)); }
void chacha20_block(const uint32_t key[8], uint64_t counter, uint64_t nonce, uint8_t out[64]) {
  uint32_t st[16] = {0x61707865u, 0x3320646eu, 0x79622d32u, 0x6b206574u, key[0], key[1], key[2], key[3], key[4], key[5], key[6], key[7],
                     (uint32_t)counter, (uint32_t)(counter >> 32), (uint32_t)nonce, (uint32_t)(nonce >> 32)};
  uint32_t x[16];
  std::memcpy(x, st, sizeof(x));
#define ZKP_QR(a, b, c, d) x[a] += x[b]; x[d] = rotl32(x[d] ^ x[a], 16); x[c] += x[d]; x[b] = rotl32(x[b] ^ x[c], 12); \
                           x[a] += x[b]; x[d] = rotl32(x[d] ^ x[a], 8);  x[c] += x[d]; x[b] = rotl32(x[b] ^ x[c], 7);
  for (int r = 0; r < 10; ++r) {
    ZKP_QR(0, 4, 8, 12) ZKP_QR(1, 5, 9, 13) ZKP_QR(2, 6, 10, 14) ZKP_QR(3, 7, 11, 15)
    ZKP_QR(0, 5, 10, 15) ZKP_QR(1, 6, 11, 12) ZKP_QR(2, 7, 8, 13) ZKP_QR(3, 4, 9, 14)
  }
#undef ZKP_QR
  for (int i = 0; i < 16; ++i) { const uint32_t v = x[i] + st[i]; std::memcpy(out + 4 * i, &v, 4); }
}

// What `thread_rng()` is to the reference (prover.rs:82, verifier.rs:153, batch_verifier.rs:179): a ChaCha stream keyed
// with 32 bytes from the operating system for every call -- getrandom() itself delivers only a few hundred MB/s, which
// for the batch verifier's 16 bytes per (constraint, proof) would cost more than the whole GPU side of the call.
bool os_random(uint8_t* out, size_t len) {
  if (len <= 256) return os_entropy(out, len);
  uint32_t key[8];
  uint64_t nonce;
  if (!os_entropy(reinterpret_cast<uint8_t*>(key), sizeof(key)) || !os_entropy(reinterpret_cast<uint8_t*>(&nonce), sizeof(nonce))) return false;
  const uint64_t blocks = (len + 63) / 64;
  parallel_for((uint32_t)std::min<uint64_t>(blocks, 0xffffffffu), 0, [&](uint32_t lo, uint32_t hi) {
    uint8_t tmp[64];
    for (uint64_t b = lo; b < hi; ++b) {
      const size_t o = (size_t)b * 64;
      if (o + 64 <= len) chacha20_block(key, b, nonce, out + o);
      else { chacha20_block(key, b, nonce, tmp); std::memcpy(out + o, tmp, len - o); }
    }
  });
  return true;
}

}  // namespace host
}  // namespace zkp

namespace {

// encoding of point variable p for proof j
inline const uint8_t* point_enc(const zkp_statement& st, uint32_t p, uint32_t j, uint32_t N, const uint8_t* inst,
                                const uint8_t* common) {
  const auto& pt = st.points[p];
  return pt.common ? common + 32 * (size_t)pt.rank : inst + 32 * ((size_t)pt.rank * N + j);
}
// index of point variable p of proof j in the device point table  common || inst (row-major [ni][N])
inline uint32_t table_index(const zkp_statement& st, uint32_t p, uint32_t j, uint32_t N) {
  const auto& pt = st.points[p];
  return pt.common ? pt.rank : st.ns + pt.rank * N + j;
}

// Scalar::from_canonical_bytes: the value must be < l = 2^252 + 27742317777372353535851937790883648493.  The reference's proofs
// reach its verifiers through serde (proofs.rs:14-32), and dalek's Deserialize refuses any other encoding of a Scalar; the
// verify entry points below take proofs as raw bytes, so they apply the same rule: a response (or challenge) >= l is a
// VerificationFailure for that proof / batch.
inline bool scalar_is_canonical(const uint8_t s[32]) {
  static const uint8_t L[32] = {0xed, 0xd3, 0xf5, 0x5c, 0x1a, 0x63, 0x12, 0x58, 0xd6, 0x9c, 0xf7, 0xa2, 0xde, 0xf9, 0xde, 0x14,
                                0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0x10};
  for (int i = 31; i >= 0; --i) {
    if (s[i] < L[i]) return true;
    if (s[i] > L[i]) return false;
  }
  return false;                                                      // == l
}
// proofs (of N) whose m responses are not all canonical -> flag[j] = 1; returns whether any is
bool flag_noncanonical(uint32_t N, uint32_t m, const uint8_t* responses, uint8_t* flag) {
  bool any = false;
  for (uint32_t j = 0; j < N; ++j)
    for (uint32_t i = 0; i < m; ++i)
      if (!scalar_is_canonical(responses + 32 * ((size_t)j * m + i))) { if (flag) flag[j] = 1; any = true; break; }
  return any;
}

bool all_transcripts_equal(const uint8_t* ts, uint32_t N) {
  for (uint32_t j = 1; j < N; ++j)
    if (std::memcmp(ts, ts + TB * (size_t)j, Transcript::kLiveBytes) != 0) return false;       // (the 5 padding bytes carry nothing)
  return true;
}

// Prover::new + the allocate_scalar calls that precede the first point (prover.rs:41-57) / the same on the verifier
// side.  When all N incoming transcripts are equal (the usual case) this prefix is hashed once and cloned.
void apply_prefix(const zkp_statement& st, uint32_t N, uint8_t* ts, int n_threads) {
  const uint32_t lead = st.scalar_prefix();
  auto prefix = [&](Transcript& t) {
    t.domain_sep(st.label.c_str());
    for (uint32_t a = 0; a < lead; ++a) t.append_scalar_var(st.secrets[st.alloc[a].idx].c_str());
  };
  if (N > 1 && all_transcripts_equal(ts, N)) {
    Transcript t = Transcript::from_bytes(ts);
    prefix(t);
    for (uint32_t j = 0; j < N; ++j) t.to_bytes(ts + TB * (size_t)j);
    return;
  }
  parallel_for(N, n_threads, [&](uint32_t lo, uint32_t hi) {
    for (uint32_t j = lo; j < hi; ++j) {
      Transcript t = Transcript::from_bytes(ts + TB * (size_t)j);
      prefix(t);
      t.to_bytes(ts + TB * (size_t)j);
    }
  });
}

// The allocations after the scalar prefix, in the caller's order (points and any scalar allocated after a point).
// validate = the verifier-side appends, which reject the identity encoding (mod.rs:186-196); false on rejection.
bool replay_allocations(const zkp_statement& st, Transcript& t, uint32_t j, uint32_t N, const uint8_t* inst, const uint8_t* common,
                        bool validate) {
  for (uint32_t a = st.scalar_prefix(); a < st.alloc.size(); ++a) {
    const auto& al = st.alloc[a];
    if (!al.is_point) { t.append_scalar_var(st.secrets[al.idx].c_str()); continue; }
    const char* name = st.points[al.idx].name.c_str();
    const uint8_t* enc = point_enc(st, al.idx, j, N, inst, common);
    if (!validate) t.append_point_var(name, enc);
    else if (!t.validate_and_append_point_var(name, enc)) return false;
  }
  return true;
}

// point variables no constraint mentions (e.g. `B` of the CMZ statement, benches/zkp.rs:32): verify_compact
// still insists that they decompress (verifier.rs:87-92)
std::vector<uint32_t> unreferenced_points(const zkp_statement& st) {
  std::vector<char> used(st.points.size(), 0);
  for (const auto& c : st.cons) {
    used[c.lhs] = 1;
    for (const auto& t : c.lc) used[t.second] = 1;
  }
  std::vector<uint32_t> out;
  for (uint32_t p = 0; p < st.points.size(); ++p)
    if (!used[p]) out.push_back(p);
  return out;
}

std::vector<uint8_t> point_table(const zkp_statement& st, uint32_t N, const uint8_t* inst, const uint8_t* common) {
  std::vector<uint8_t> tbl(32 * ((size_t)st.ns + (size_t)st.ni * N));
  if (st.ns) std::memcpy(tbl.data(), common, 32 * (size_t)st.ns);
  if (st.ni) std::memcpy(tbl.data() + 32 * (size_t)st.ns, inst, 32 * (size_t)st.ni * N);
  return tbl;
}

}  // namespace

extern "C" {

void zkp_chacha20_block(const uint8_t key[32], uint64_t counter, uint64_t nonce, uint8_t out[64]) {
  uint32_t k[8];
  std::memcpy(k, key, 32);
  chacha20_block(k, counter, nonce, out);
}
void zkp_toolbox_set_host_max_terms(uint32_t n) { g_host_max_terms = n; }
uint32_t zkp_toolbox_get_host_max_terms(void) { return g_host_max_terms.load(); }
void zkp_toolbox_set_fused_min_batch(uint32_t n) { g_fused_min_batch = n; }
uint32_t zkp_toolbox_get_fused_min_batch(void) { return g_fused_min_batch.load(); }

// ---- transcripts / scalars -----------------------------------------------------------------------------
void zkp_transcript_init(uint8_t* t, const uint8_t* label, size_t len) { Transcript(label, len).to_bytes(t); }
// merlin's encode_usize_as_u32 asserts that lengths fit 32 bits (the > 4 GiB case of tests/sig_and_vrf_example.rs:224-241):
// an error here, never a truncated length prefix
static bool fits_u32(size_t n) { return (uint64_t)n <= 0xffffffffull; }
int zkp_transcript_append_message(uint8_t* t, const char* label, const uint8_t* msg, size_t len) {
  if (!t || !label || (len && !msg)) return ZKP_TB_BAD_STATEMENT;
  if (!fits_u32(len) || !fits_u32(std::strlen(label))) return ZKP_TB_TOO_LONG;
  Transcript x = Transcript::from_bytes(t);
  x.append_message(label, msg, len);
  x.to_bytes(t);
  return ZKP_TB_OK;
}
int zkp_transcript_challenge_bytes(uint8_t* t, const char* label, uint8_t* out, size_t len) {
  if (!t || !label || (len && !out)) return ZKP_TB_BAD_STATEMENT;
  if (!fits_u32(len) || !fits_u32(std::strlen(label))) return ZKP_TB_TOO_LONG;
  Transcript x = Transcript::from_bytes(t);
  x.challenge_bytes(label, out, len);
  x.to_bytes(t);
  return ZKP_TB_OK;
}
void zkp_scalar_from_wide(uint8_t out[32], const uint8_t in[64]) { Scalar::from_bytes_mod_order_wide(in).to_bytes(out); }
void zkp_scalar_muladd(uint8_t out[32], const uint8_t a[32], const uint8_t b[32], const uint8_t c[32]) {
  (Scalar::from_bytes_mod_order(a) * Scalar::from_bytes_mod_order(b) + Scalar::from_bytes_mod_order(c)).to_bytes(out);
}
void zkp_scalar_neg(uint8_t out[32], const uint8_t a[32]) { (-Scalar::from_bytes_mod_order(a)).to_bytes(out); }

// ---- proof wire format (proofs.rs:14-32 under bincode 1.x; see zkp_toolbox.h) -----------------------------------------
namespace {
inline void put_u64le(uint8_t* out, uint64_t v) { for (int i = 0; i < 8; ++i) out[i] = (uint8_t)(v >> (8 * i)); }
inline uint64_t get_u64le(const uint8_t* in) { uint64_t v = 0; for (int i = 0; i < 8; ++i) v |= (uint64_t)in[i] << (8 * i); return v; }
// u64 count + count x 32 bytes at in[pos..]; false = truncated / count larger than the remaining bytes
inline bool read_vec32(const uint8_t* in, size_t len, size_t& pos, uint64_t& count) {
  if (len - pos < 8) return false;
  count = get_u64le(in + pos);
  pos += 8;
  if (count > (len - pos) / 32) return false;
  return true;
}
}  // namespace

size_t zkp_proof_compact_size(uint32_t m) { return 40 + 32 * (size_t)m; }
size_t zkp_proof_batchable_size(uint32_t nc, uint32_t m) { return 16 + 32 * ((size_t)nc + m); }

int zkp_proof_compact_encode(const uint8_t challenge[32], const uint8_t* responses, uint32_t m, uint8_t* out, size_t out_len) {
  if (!challenge || !out || (m && !responses) || out_len < zkp_proof_compact_size(m)) return ZKP_TB_BAD_STATEMENT;
  std::memcpy(out, challenge, 32);
  put_u64le(out + 32, m);
  if (m) std::memcpy(out + 40, responses, 32 * (size_t)m);
  return ZKP_TB_OK;
}
int zkp_proof_compact_decode(const uint8_t* in, size_t len, uint8_t challenge[32], uint8_t* responses, uint32_t max_m, uint32_t* m,
                             size_t* consumed) {
  if (!in || !challenge || !m || (max_m && !responses)) return ZKP_TB_BAD_STATEMENT;
  if (len < 32) return ZKP_TB_BAD_ENCODING;
  size_t pos = 32;
  uint64_t cnt = 0;
  if (!read_vec32(in, len, pos, cnt)) return ZKP_TB_BAD_ENCODING;
  if (cnt > max_m) return ZKP_TB_BAD_STATEMENT;
  if (!scalar_is_canonical(in)) return ZKP_TB_BAD_ENCODING;
  for (uint64_t i = 0; i < cnt; ++i)
    if (!scalar_is_canonical(in + pos + 32 * i)) return ZKP_TB_BAD_ENCODING;
  std::memcpy(challenge, in, 32);
  if (cnt) std::memcpy(responses, in + pos, 32 * (size_t)cnt);
  *m = (uint32_t)cnt;
  if (consumed) *consumed = pos + 32 * (size_t)cnt;
  return ZKP_TB_OK;
}
int zkp_proof_batchable_encode(const uint8_t* commitments, uint32_t nc, const uint8_t* responses, uint32_t m, uint8_t* out, size_t out_len) {
  if (!out || (nc && !commitments) || (m && !responses) || out_len < zkp_proof_batchable_size(nc, m)) return ZKP_TB_BAD_STATEMENT;
  put_u64le(out, nc);
  if (nc) std::memcpy(out + 8, commitments, 32 * (size_t)nc);
  uint8_t* q = out + 8 + 32 * (size_t)nc;
  put_u64le(q, m);
  if (m) std::memcpy(q + 8, responses, 32 * (size_t)m);
  return ZKP_TB_OK;
}
int zkp_proof_batchable_decode(const uint8_t* in, size_t len, uint8_t* commitments, uint32_t max_nc, uint32_t* nc, uint8_t* responses,
                               uint32_t max_m, uint32_t* m, size_t* consumed) {
  if (!in || !nc || !m || (max_nc && !commitments) || (max_m && !responses)) return ZKP_TB_BAD_STATEMENT;
  size_t pos = 0;
  uint64_t c = 0, r = 0;
  if (!read_vec32(in, len, pos, c)) return ZKP_TB_BAD_ENCODING;
  const size_t cpos = pos;
  pos += 32 * (size_t)c;
  if (!read_vec32(in, len, pos, r)) return ZKP_TB_BAD_ENCODING;
  if (c > max_nc || r > max_m) return ZKP_TB_BAD_STATEMENT;
  for (uint64_t i = 0; i < r; ++i)
    if (!scalar_is_canonical(in + pos + 32 * i)) return ZKP_TB_BAD_ENCODING;
  if (c) std::memcpy(commitments, in + cpos, 32 * (size_t)c);      // CompressedRistretto: any 32 bytes (validity = decompress())
  if (r) std::memcpy(responses, in + pos, 32 * (size_t)r);
  *nc = (uint32_t)c;
  *m = (uint32_t)r;
  if (consumed) *consumed = pos + 32 * (size_t)r;
  return ZKP_TB_OK;
}

// ---- statements -------------------------------------------------------------------------------------------
zkp_statement* zkp_statement_new(const char* proof_label) {
  auto* st = new zkp_statement();
  st->label = proof_label ? proof_label : "";
  return st;
}
void zkp_statement_free(zkp_statement* st) { delete st; }
int zkp_statement_add_secret(zkp_statement* st, const char* name) {
  if (!st || !name) return ZKP_TB_BAD_STATEMENT;
  st->secrets.emplace_back(name);
  st->alloc.push_back({false, (uint32_t)st->secrets.size() - 1});
  return (int)st->secrets.size() - 1;
}
int zkp_statement_add_point(zkp_statement* st, const char* name, int is_common) {
  if (!st || !name) return ZKP_TB_BAD_STATEMENT;
  st->points.push_back({name, is_common != 0, is_common ? st->ns++ : st->ni++});
  st->alloc.push_back({true, (uint32_t)st->points.size() - 1});
  return (int)st->points.size() - 1;
}
int zkp_statement_constrain(zkp_statement* st, uint32_t lhs, uint32_t n_terms, const uint32_t* secrets, const uint32_t* points) {
  if (!st || lhs >= st->points.size() || (n_terms && (!secrets || !points))) return ZKP_TB_BAD_STATEMENT;
  zkp_statement::Constraint c;
  c.lhs = lhs;
  for (uint32_t i = 0; i < n_terms; ++i) {
    if (secrets[i] >= st->secrets.size() || points[i] >= st->points.size()) return ZKP_TB_BAD_STATEMENT;
    c.lc.emplace_back(secrets[i], points[i]);
  }
  st->terms += n_terms;
  st->cons.push_back(std::move(c));
  return ZKP_TB_OK;
}
uint32_t zkp_statement_num_secrets(const zkp_statement* st) { return st ? (uint32_t)st->secrets.size() : 0; }
uint32_t zkp_statement_num_instance(const zkp_statement* st) { return st ? st->ni : 0; }
uint32_t zkp_statement_num_common(const zkp_statement* st) { return st ? st->ns : 0; }
uint32_t zkp_statement_num_constraints(const zkp_statement* st) { return st ? (uint32_t)st->cons.size() : 0; }
uint32_t zkp_statement_num_terms(const zkp_statement* st) { return st ? st->terms : 0; }

// ---- prover -----------------------------------------------------------------------------------------------
int zkp_prove_phase_a(const zkp_statement* stp, uint32_t N, uint8_t* ts, const uint8_t* secrets, const uint8_t* inst,
                      const uint8_t* common, const uint8_t* entropy, int n_threads, uint8_t* blindings, uint32_t* off,
                      uint8_t* scalars, uint32_t* pidx) {
  if (!stp || !ts || !off) return ZKP_TB_BAD_STATEMENT;
  // empty parts of a statement come with empty (possibly NULL) buffers
  if ((!stp->secrets.empty() && (!secrets || !blindings)) || (stp->terms && (!scalars || !pidx))) return ZKP_TB_BAD_STATEMENT;
  const zkp_statement& st = *stp;
  const uint32_t m = (uint32_t)st.secrets.size(), nc = (uint32_t)st.cons.size(), T = st.terms;
  apply_prefix(st, N, ts, n_threads);                                  // prover.rs:42, :54
  std::vector<uint8_t> own_entropy;
  if (!entropy) {                                                      // prover.rs:82 `thread_rng()`
    own_entropy.resize(32 * (size_t)N);
    if (!os_random(own_entropy.data(), own_entropy.size())) return ZKP_TB_NO_ENTROPY;
    entropy = own_entropy.data();
  }
  parallel_for(N, n_threads, [&](uint32_t lo, uint32_t hi) {
    for (uint32_t j = lo; j < hi; ++j) {
      Transcript t = Transcript::from_bytes(ts + TB * (size_t)j);
      replay_allocations(st, t, j, N, inst, common, false);            // prover.rs:52-73 (encodings are supplied)
      zkp::host::TranscriptRng rng = t.build_rng();                    // prover.rs:78-82
      for (uint32_t i = 0; i < m; ++i) rng.rekey_with_witness_bytes("", secrets + 32 * ((size_t)j * m + i), 32);
      rng.finalize(entropy + 32 * (size_t)j);
      uint8_t* b = blindings + 32 * (size_t)j * m;
      for (uint32_t i = 0; i < m; ++i) {                               // prover.rs:85-89 Scalar::random
        uint8_t wide[64];
        rng.fill_bytes(wide, 64);
        Scalar::from_bytes_mod_order_wide(wide).to_bytes(b + 32 * i);
      }
      size_t q = (size_t)j * T;                                        // prover.rs:94-97 operand lists
      for (uint32_t k = 0; k < nc; ++k) {
        off[(size_t)j * nc + k] = (uint32_t)q;
        for (const auto& term : st.cons[k].lc) {
          std::memcpy(scalars + 32 * q, b + 32 * term.first, 32);
          pidx[q] = table_index(st, term.second, j, N);
          ++q;
        }
      }
      t.to_bytes(ts + TB * (size_t)j);
    }
  });
  off[(size_t)N * nc] = (uint32_t)((size_t)N * T);
  return ZKP_TB_OK;
}

int zkp_prove_phase_b(const zkp_statement* stp, uint32_t N, uint8_t* ts, const uint8_t* secrets, const uint8_t* blindings,
                      const uint8_t* commitments, int n_threads, uint8_t* challenges, uint8_t* responses) {
  if (!stp || !ts || !challenges || (!stp->cons.empty() && !commitments) || (!stp->secrets.empty() && !responses)) return ZKP_TB_BAD_STATEMENT;
  const zkp_statement& st = *stp;
  const uint32_t m = (uint32_t)st.secrets.size(), nc = (uint32_t)st.cons.size();
  parallel_for(N, n_threads, [&](uint32_t lo, uint32_t hi) {
    for (uint32_t j = lo; j < hi; ++j) {
      Transcript t = Transcript::from_bytes(ts + TB * (size_t)j);
      for (uint32_t k = 0; k < nc; ++k)                                // prover.rs:98-100
        t.append_blinding_commitment(st.points[st.cons[k].lhs].name.c_str(), commitments + 32 * ((size_t)j * nc + k));
      uint8_t* c = challenges + 32 * (size_t)j;
      t.get_challenge("chal", c);                                      // prover.rs:106
      const Scalar cs = Scalar::from_bytes_mod_order(c);
      for (uint32_t i = 0; i < m; ++i) {                               // prover.rs:107-109  s * c + b
        const size_t o = 32 * ((size_t)j * m + i);
        (Scalar::from_bytes_mod_order(secrets + o) * cs + Scalar::from_bytes_mod_order(blindings + o)).to_bytes(responses + o);
      }
      t.to_bytes(ts + TB * (size_t)j);
    }
  });
  return ZKP_TB_OK;
}

int zkp_prove_batch(zkp_ctx* ctx, const zkp_statement* st, uint32_t N, uint8_t* ts, const uint8_t* secrets,
                    const uint8_t* inst, const uint8_t* common, const uint8_t* entropy, int n_threads, uint8_t* challenges,
                    uint8_t* responses, uint8_t* commitments) {
  if (!st) return ZKP_TB_BAD_STATEMENT;                                  // (ctx == NULL: the host backend, host_backend.cpp)
  if (N == 0) return ZKP_TB_OK;
  if (ctx && ts && use_fused(ts, N)) {
    // no entropy from the caller: 40 bytes of getrandom() key a ChaCha20 stream that is expanded on the device (what thread_rng() is to prover.rs:82)
    uint8_t seed[40];
    if (!entropy && !os_entropy(seed, sizeof(seed))) return ZKP_TB_NO_ENTROPY;
    if (st->ns && N >= 32) { const int rc = zkp_ctx_prepare_fixed_points(ctx, st->ns, common); if (rc) return rc; }
    FusedView fv(*st);
    int invalid = 0;
    const int rc = entropy ? zkp_fused_prove(ctx, &fv.fs, N, ts, secrets, inst, common, entropy, challenges, responses, commitments, &invalid)
                           : zkp_fused_prove_seeded(ctx, &fv.fs, N, ts, secrets, inst, common, seed, challenges, responses, commitments, &invalid);
    if (rc) return rc;
    return invalid ? ZKP_TB_INVALID_POINT : ZKP_TB_OK;
  }
  const uint32_t m = (uint32_t)st->secrets.size(), nc = (uint32_t)st->cons.size(), T = st->terms;
  std::vector<uint8_t> blind(32 * (size_t)N * m), scalars(32 * (size_t)N * T), status((size_t)N * nc);
  std::vector<uint32_t> off((size_t)N * nc + 1), pidx((size_t)N * T);
  int rc = zkp_prove_phase_a(st, N, ts, secrets, inst, common, entropy, n_threads, blind.data(), off.data(), scalars.data(), pidx.data());
  if (rc) return rc;
  const std::vector<uint8_t> tbl = point_table(*st, N, inst, common);
  if (ctx && st->ns && N >= 32) { rc = zkp_ctx_prepare_fixed_points(ctx, st->ns, common); if (rc) return rc; }   // common points: fixed-base tables
  // prover.rs:94 RistrettoPoint::multiscalar_mul for every constraint of every proof, + compress (mod.rs:204)
  rc = be_msm_many(ctx, N * nc, off.data(), scalars.data(), pidx.data(), tbl.data(), (uint32_t)(tbl.size() / 32), ZKP_CT,
                   commitments, status.data());
  if (rc) return rc;
  for (uint8_t s : status)
    if (s) return ZKP_TB_INVALID_POINT;     // the reference prover holds decoded points; an undecodable input is a caller bug
  return zkp_prove_phase_b(st, N, ts, secrets, blind.data(), commitments, n_threads, challenges, responses);
}

// ---- single-proof verification, batched over N independent proofs -----------------------------------------
// Verifier::new + allocate_* (verifier.rs:47-77); failed[j] = 1 when an identity point is rejected
static void build_verifiers(const zkp_statement& st, uint32_t N, uint8_t* ts, const uint8_t* inst, const uint8_t* common,
                            int n_threads, uint8_t* failed) {
  apply_prefix(st, N, ts, n_threads);
  parallel_for(N, n_threads, [&](uint32_t lo, uint32_t hi) {
    for (uint32_t j = lo; j < hi; ++j) {
      Transcript t = Transcript::from_bytes(ts + TB * (size_t)j);
      if (!replay_allocations(st, t, j, N, inst, common, true)) failed[j] = 1;
      t.to_bytes(ts + TB * (size_t)j);
    }
  });
}

int zkp_verify_compact_batch(zkp_ctx* ctx, const zkp_statement* stp, uint32_t N, uint8_t* ts, const uint8_t* inst,
                             const uint8_t* common, const uint8_t* challenges, const uint8_t* responses, int n_threads,
                             uint8_t* results) {
  if (!stp || !ts || !results || !challenges || (!stp->secrets.empty() && !responses)) return ZKP_TB_BAD_STATEMENT;
  if (N == 0) return ZKP_TB_OK;
  const zkp_statement& st = *stp;
  if (ctx && use_fused(ts, N)) {
    if (st.ns && N >= 32) { const int rc = zkp_ctx_prepare_fixed_points(ctx, st.ns, common); if (rc) return rc; }
    FusedView fv(st);
    return zkp_fused_verify_compact(ctx, &fv.fs, N, ts, inst, common, challenges, responses, results);
  }
  const uint32_t m = (uint32_t)st.secrets.size(), nc = (uint32_t)st.cons.size(), T1 = st.terms + nc;
  std::memset(results, 0, N);
  flag_noncanonical(N, m, responses, results);                         // proofs.rs:15-20 through serde: s >= l never reaches the verifier
  build_verifiers(st, N, ts, inst, common, n_threads, results);
  // verifier.rs:95-106: per constraint, responses over the rhs points and (-c) over the lhs point
  std::vector<uint8_t> scalars(32 * (size_t)N * T1), coms(32 * (size_t)N * nc), status((size_t)N * nc);
  std::vector<uint32_t> off((size_t)N * nc + 1), pidx((size_t)N * T1);
  parallel_for(N, n_threads, [&](uint32_t lo, uint32_t hi) {
    for (uint32_t j = lo; j < hi; ++j) {
      uint8_t minus_c[32];
      (-Scalar::from_bytes_mod_order(challenges + 32 * (size_t)j)).to_bytes(minus_c);
      size_t q = (size_t)j * T1;
      for (uint32_t k = 0; k < nc; ++k) {
        off[(size_t)j * nc + k] = (uint32_t)q;
        for (const auto& term : st.cons[k].lc) {
          std::memcpy(scalars.data() + 32 * q, responses + 32 * ((size_t)j * m + term.first), 32);
          pidx[q++] = table_index(st, term.second, j, N);
        }
        std::memcpy(scalars.data() + 32 * q, minus_c, 32);
        pidx[q++] = table_index(st, st.cons[k].lhs, j, N);
      }
    }
  });
  off[(size_t)N * nc] = (uint32_t)((size_t)N * T1);
  const std::vector<uint8_t> tbl = point_table(st, N, inst, common);
  int rc = (ctx && st.ns && N >= 32) ? zkp_ctx_prepare_fixed_points(ctx, st.ns, common) : 0;
  if (rc) return rc;
  const bool host = on_host(ctx, (uint64_t)N * T1);
  rc = be_msm_many(ctx, N * nc, off.data(), scalars.data(), pidx.data(), tbl.data(), (uint32_t)(tbl.size() / 32),
                   ZKP_VARTIME, coms.data(), status.data());
  if (rc) return rc;
  // verifier.rs:87-92 decompresses EVERY allocated point, also those no constraint uses
  const std::vector<uint32_t> unref = unreferenced_points(st);
  if (!unref.empty()) {
    std::vector<uint8_t> encs, st8;
    std::vector<std::pair<uint32_t, int>> owner;     // (proof or ~0 for all, unused)
    for (uint32_t p : unref) {
      if (st.points[p].common) { encs.insert(encs.end(), common + 32 * (size_t)st.points[p].rank, common + 32 * (size_t)st.points[p].rank + 32); owner.emplace_back(~0u, 0); }
      else for (uint32_t j = 0; j < N; ++j) { const uint8_t* e = point_enc(st, p, j, N, inst, common); encs.insert(encs.end(), e, e + 32); owner.emplace_back(j, 0); }
    }
    st8.resize(owner.size());
    rc = be_decode_check(ctx, host, owner.size(), encs.data(), st8.data());
    if (rc) return rc;
    for (size_t i = 0; i < owner.size(); ++i)
      if (st8[i]) {
        if (owner[i].first == ~0u) std::memset(results, 1, N);
        else results[owner[i].first] = 1;
      }
  }
  parallel_for(N, n_threads, [&](uint32_t lo, uint32_t hi) {
    for (uint32_t j = lo; j < hi; ++j) {
      if (results[j]) continue;
      bool bad = false;
      for (uint32_t k = 0; k < nc; ++k) bad |= status[(size_t)j * nc + k] != 0;
      if (bad) { results[j] = 1; continue; }
      Transcript t = Transcript::from_bytes(ts + TB * (size_t)j);
      for (uint32_t k = 0; k < nc; ++k)                                // verifier.rs:108 (non-validating append)
        t.append_blinding_commitment(st.points[st.cons[k].lhs].name.c_str(), coms.data() + 32 * ((size_t)j * nc + k));
      uint8_t c[32];
      t.get_challenge("chal", c);                                      // verifier.rs:113-119
      // the recomputed challenge is canonical and is compared with the claimed BYTES: c + l is a different `Scalar`
      // for the reference (and does not even deserialise), so it must not verify here either
      results[j] = std::memcmp(c, challenges + 32 * (size_t)j, 32) == 0 ? 0 : 1;
      t.to_bytes(ts + TB * (size_t)j);
    }
  });
  return ZKP_TB_OK;
}

int zkp_verify_batchable_each(zkp_ctx* ctx, const zkp_statement* stp, uint32_t N, uint8_t* ts, const uint8_t* inst,
                              const uint8_t* common, const uint8_t* commitments, const uint8_t* responses,
                              const uint8_t* weights16, int n_threads, uint8_t* results) {
  if (!stp || !ts || !results || (!stp->cons.empty() && !commitments) || (!stp->secrets.empty() && !responses)) return ZKP_TB_BAD_STATEMENT;
  if (N == 0) return ZKP_TB_OK;
  const zkp_statement& st = *stp;
  const uint32_t m = (uint32_t)st.secrets.size(), nc = (uint32_t)st.cons.size(), np = (uint32_t)st.points.size();
  std::vector<uint8_t> own_w;
  if (!weights16) { own_w.resize(16 * (size_t)N * nc); if (!os_random(own_w.data(), own_w.size())) return ZKP_TB_NO_ENTROPY; weights16 = own_w.data(); }
  if (ctx && use_fused(ts, N)) {
    if (st.ns && N >= 32) { const int rc = zkp_ctx_prepare_fixed_points(ctx, st.ns, common); if (rc) return rc; }
    FusedView fv(st);
    return zkp_fused_verify_batchable(ctx, &fv.fs, N, ts, inst, common, commitments, responses, weights16, results);
  }
  std::memset(results, 0, N);
  flag_noncanonical(N, m, responses, results);                         // proofs.rs:27-32 through serde
  build_verifiers(st, N, ts, inst, common, n_threads, results);
  // one (np + nc)-term MSM per proof over  points || commitments   (verifier.rs:144-166)
  const uint32_t K = np + nc;
  std::vector<uint8_t> scalars(32 * (size_t)N * K), out(32 * (size_t)N), status(N);
  std::vector<uint32_t> off((size_t)N + 1), pidx((size_t)N * K);
  // point table: common || inst || commitments [N][nc]
  std::vector<uint8_t> tbl = point_table(st, N, inst, common);
  const uint32_t com_base = (uint32_t)(tbl.size() / 32);
  tbl.insert(tbl.end(), commitments, commitments + 32 * (size_t)N * nc);
  parallel_for(N, n_threads, [&](uint32_t lo, uint32_t hi) {
    std::vector<Scalar> coeffs(K);
    for (uint32_t j = lo; j < hi; ++j) {
      off[j] = j * K;
      if (results[j]) { for (uint32_t i = 0; i < K; ++i) pidx[(size_t)j * K + i] = 0; continue; }
      Transcript t = Transcript::from_bytes(ts + TB * (size_t)j);
      for (uint32_t k = 0; k < nc && !results[j]; ++k)                 // verifier.rs:134-140
        if (!t.validate_and_append_blinding_commitment(st.points[st.cons[k].lhs].name.c_str(), commitments + 32 * ((size_t)j * nc + k)))
          results[j] = 1;
      uint8_t c[32];
      t.get_challenge("chal", c);
      t.to_bytes(ts + TB * (size_t)j);
      const Scalar minus_c = -Scalar::from_bytes_mod_order(c);         // verifier.rs:142
      std::fill(coeffs.begin(), coeffs.end(), Scalar::zero());
      for (uint32_t k = 0; k < nc; ++k) {                              // verifier.rs:151-160
        const Scalar r = Scalar::from_u128_le(weights16 + 16 * ((size_t)j * nc + k));
        coeffs[np + k] -= r;
        coeffs[st.cons[k].lhs] += r * minus_c;
        for (const auto& term : st.cons[k].lc)
          coeffs[term.second] += r * Scalar::from_bytes_mod_order(responses + 32 * ((size_t)j * m + term.first));
      }
      for (uint32_t i = 0; i < K; ++i) {
        coeffs[i].to_bytes(scalars.data() + 32 * ((size_t)j * K + i));
        pidx[(size_t)j * K + i] = i < np ? table_index(st, i, j, N) : com_base + j * nc + (i - np);
      }
    }
  });
  off[N] = N * K;
  int rc = (ctx && st.ns && N >= 32) ? zkp_ctx_prepare_fixed_points(ctx, st.ns, common) : 0;
  if (rc) return rc;
  rc = be_msm_many(ctx, N, off.data(), scalars.data(), pidx.data(), tbl.data(), (uint32_t)(tbl.size() / 32), ZKP_VARTIME,
                   out.data(), status.data());
  if (rc) return rc;
  static const uint8_t zero[32] = {0};
  for (uint32_t j = 0; j < N; ++j)                                     // verifier.rs:162-172
    if (!results[j]) results[j] = (status[j] || std::memcmp(out.data() + 32 * (size_t)j, zero, 32) != 0) ? 1 : 0;
  return ZKP_TB_OK;
}

// ---- batch verification --------------------------------------------------------------------------------------
int zkp_batch_verify_build(const zkp_statement* stp, uint32_t N, uint32_t n_transcripts, uint8_t* ts, const uint8_t* inst,
                           const uint8_t* common, const uint8_t* commitments, const uint8_t* responses,
                           const uint8_t* weights16, int n_threads, uint8_t* msm_scalars, uint8_t* msm_points) {
  if (!stp || !msm_scalars || !msm_points) return ZKP_TB_BAD_STATEMENT;
  if (n_transcripts != N) return ZKP_TB_BATCH_SIZE_MISMATCH;           // batch_verifier.rs:72-74
  const zkp_statement& st = *stp;
  const uint32_t m = (uint32_t)st.secrets.size(), nc = (uint32_t)st.cons.size(), ni = st.ni, ns = st.ns;
  const size_t rows = (size_t)ni + nc;
  if (N == 0) { std::memset(msm_scalars, 0, 32 * (size_t)ns); if (ns) std::memcpy(msm_points, common, 32 * (size_t)ns); return ZKP_TB_OK; }
  if (flag_noncanonical(N, m, responses, nullptr)) return ZKP_TB_VERIFICATION_FAILURE;   // proofs.rs:27-32 through serde
  std::vector<uint8_t> failed(N, 0);
  build_verifiers(st, N, ts, inst, common, n_threads, failed.data());  // :75-77, :92-94, :105-107, :125-128
  for (uint8_t f : failed) if (f) return ZKP_TB_VERIFICATION_FAILURE;
  std::vector<uint8_t> own_w;
  if (!weights16) { own_w.resize(16 * (size_t)N * nc); if (!os_random(own_w.data(), own_w.size())) return ZKP_TB_NO_ENTROPY; weights16 = own_w.data(); }
  uint8_t* inst_coeffs = msm_scalars + 32 * (size_t)ns;                // Matrix(rows, N), entries[cols * r + c] (util.rs:24)
  std::atomic<int> any_fail{0};
  // per-thread partial sums of the static coefficients, reduced afterwards (:187, :198 sum over the whole batch)
  unsigned n_workers = n_threads > 0 ? (unsigned)n_threads : std::thread::hardware_concurrency();
  if (n_workers == 0) n_workers = 1;
  std::vector<std::vector<Scalar>> static_parts;
  std::vector<Scalar> static_total(ns);
  std::atomic<unsigned> slot{0};
  static_parts.resize(std::max(n_workers, 128u) + 1, std::vector<Scalar>(ns));
  parallel_for(N, n_threads, [&](uint32_t lo, uint32_t hi) {
    std::vector<Scalar>& sp = static_parts[slot.fetch_add(1) % static_parts.size()];
    std::vector<Scalar> col(rows);
    for (uint32_t j = lo; j < hi; ++j) {
      Transcript t = Transcript::from_bytes(ts + TB * (size_t)j);
      for (uint32_t k = 0; k < nc; ++k)                                // :152-160
        if (!t.validate_and_append_blinding_commitment(st.points[st.cons[k].lhs].name.c_str(), commitments + 32 * ((size_t)j * nc + k))) any_fail = 1;
      uint8_t c[32];
      t.get_challenge("chal", c);                                      // :163-167
      t.to_bytes(ts + TB * (size_t)j);
      const Scalar minus_c = -Scalar::from_bytes_mod_order(c);
      std::fill(col.begin(), col.end(), Scalar::zero());
      for (uint32_t k = 0; k < nc; ++k) {                              // :176-206 (loop order swapped: proof-major)
        const Scalar r = Scalar::from_u128_le(weights16 + 16 * ((size_t)k * N + j));
        col[ni + k] -= r;                                              // :183
        const auto& lhs = st.points[st.cons[k].lhs];
        if (lhs.common) sp[lhs.rank] += r * minus_c; else col[lhs.rank] += r * minus_c;     // :185-192
        for (const auto& term : st.cons[k].lc) {                       // :194-204
          const Scalar rr = r * Scalar::from_bytes_mod_order(responses + 32 * ((size_t)j * m + term.first));
          const auto& pt = st.points[term.second];
          if (pt.common) sp[pt.rank] += rr; else col[pt.rank] += rr;
        }
      }
      for (size_t r = 0; r < rows; ++r) col[r].to_bytes(inst_coeffs + 32 * (r * N + j));
    }
  });
  if (any_fail) return ZKP_TB_VERIFICATION_FAILURE;
  for (const auto& part : static_parts)
    for (uint32_t i = 0; i < ns; ++i) static_total[i] += part[i];
  for (uint32_t i = 0; i < ns; ++i) static_total[i].to_bytes(msm_scalars + 32 * (size_t)i);
  // points: static || instance rows || commitment rows (transposed to row = constraint, column = proof) :208-217
  if (ns) std::memcpy(msm_points, common, 32 * (size_t)ns);
  if (ni) std::memcpy(msm_points + 32 * (size_t)ns, inst, 32 * (size_t)ni * N);
  uint8_t* com_rows = msm_points + 32 * ((size_t)ns + (size_t)ni * N);
  for (uint32_t j = 0; j < N; ++j)
    for (uint32_t k = 0; k < nc; ++k) std::memcpy(com_rows + 32 * ((size_t)k * N + j), commitments + 32 * ((size_t)j * nc + k), 32);
  return ZKP_TB_OK;
}

// Same checks, with the coefficient build on the GPU (zkp_batch_check): the host keeps the transcripts (validation,
// commitments, challenges :152-167), the device does :173-228 -- scalar arithmetic mod l included.
int zkp_batch_verify_coeffs(zkp_ctx* ctx, const zkp_statement* stp, uint32_t N, uint32_t n_transcripts, uint8_t* ts, const uint8_t* inst,
                            const uint8_t* common, const uint8_t* commitments, const uint8_t* responses, const uint8_t* weights16,
                            int n_threads, uint8_t* coeffs) {
  if (!stp) return ZKP_TB_BAD_STATEMENT;
  if (n_transcripts != N) return ZKP_TB_BATCH_SIZE_MISMATCH;           // batch_verifier.rs:72-74
  const zkp_statement& st = *stp;
  const uint32_t m = (uint32_t)st.secrets.size(), nc = (uint32_t)st.cons.size(), ni = st.ni, ns = st.ns;
  if (on_host(ctx, (uint64_t)ns + ((uint64_t)ni + nc) * N)) {
    // tiny batch (or no GPU context): batch_verifier.rs:67-235 entirely on the host -- the operand list of :219-228 from
    // zkp_batch_verify_build, the MSM from the host backend
    const size_t total = (size_t)ns + ((size_t)ni + nc) * N;
    std::vector<uint8_t> sc(32 * (total ? total : 1)), pts(32 * (total ? total : 1));
    const int rcb = zkp_batch_verify_build(stp, N, n_transcripts, ts, inst, common, commitments, responses, weights16, n_threads, sc.data(), pts.data());
    if (rcb) return rcb;
    if (coeffs) std::memcpy(coeffs, sc.data(), 32 * total);
    uint8_t out[32];
    int status = 1;
    const int rcm = zkp::hostbk::msm_optional(total, sc.data(), pts.data(), out, &status);
    if (rcm) return rcm;
    static const uint8_t zero32[32] = {0};
    return (!status && std::memcmp(out, zero32, 32) == 0) ? ZKP_TB_OK : ZKP_TB_VERIFICATION_FAILURE;      // :230-234
  }
  if (ts && use_fused(ts, N)) {
    FusedView fv(st);
    int verdict = 1;
    if (!weights16 && !coeffs && nc) {
      // no weights from the caller: 40 bytes of getrandom() key a ChaCha20 stream that is expanded on the device (batch_verifier.rs:179's thread_rng())
      uint8_t seed[40];
      if (!os_entropy(seed, sizeof(seed))) return ZKP_TB_NO_ENTROPY;
      const int rc = zkp_fused_batch_verify_many_seeded(ctx, &fv.fs, 1, N, ts, inst, common, commitments, responses, seed, &verdict);
      if (rc) return rc;
      return verdict ? ZKP_TB_VERIFICATION_FAILURE : ZKP_TB_OK;
    }
    std::vector<uint8_t> own_w;
    if (!weights16) { own_w.resize(16 * (size_t)N * nc); if (!os_random(own_w.data(), own_w.size())) return ZKP_TB_NO_ENTROPY; weights16 = own_w.data(); }
    const int rc = zkp_fused_batch_verify(ctx, &fv.fs, N, ts, inst, common, commitments, responses, weights16, &verdict, coeffs);
    if (rc) return rc;
    return verdict ? ZKP_TB_VERIFICATION_FAILURE : ZKP_TB_OK;
  }
  std::vector<uint8_t> minus_c(32 * (size_t)N);
  if (N) {
    if (flag_noncanonical(N, m, responses, nullptr)) return ZKP_TB_VERIFICATION_FAILURE;   // proofs.rs:27-32 through serde
    std::vector<uint8_t> failed(N, 0);
    build_verifiers(st, N, ts, inst, common, n_threads, failed.data());  // :75-77, :92-94, :105-107, :125-128
    for (uint8_t f : failed) if (f) return ZKP_TB_VERIFICATION_FAILURE;
    std::atomic<int> any_fail{0};
    parallel_for(N, n_threads, [&](uint32_t lo, uint32_t hi) {
      for (uint32_t j = lo; j < hi; ++j) {
        Transcript t = Transcript::from_bytes(ts + TB * (size_t)j);
        for (uint32_t k = 0; k < nc; ++k)                                // :152-160
          if (!t.validate_and_append_blinding_commitment(st.points[st.cons[k].lhs].name.c_str(), commitments + 32 * ((size_t)j * nc + k))) any_fail = 1;
        uint8_t c[32];
        t.get_challenge("chal", c);                                      // :163-167
        t.to_bytes(ts + TB * (size_t)j);
        (-Scalar::from_bytes_mod_order(c)).to_bytes(minus_c.data() + 32 * (size_t)j);
      }
    });
    if (any_fail) return ZKP_TB_VERIFICATION_FAILURE;
  }
  std::vector<uint8_t> own_w;
  if (!weights16) { own_w.resize(16 * (size_t)N * nc); if (!os_random(own_w.data(), own_w.size())) return ZKP_TB_NO_ENTROPY; weights16 = own_w.data(); }
  // statement incidence in point-id form (static ids first, then instance ids)
  auto pid = [&](uint32_t v) { return st.points[v].common ? st.points[v].rank : ns + st.points[v].rank; };
  std::vector<uint32_t> lhs(nc), off(nc + 1, 0), csc, cpt;
  for (uint32_t k = 0; k < nc; ++k) {
    lhs[k] = pid(st.cons[k].lhs);
    for (const auto& term : st.cons[k].lc) { csc.push_back(term.first); cpt.push_back(pid(term.second)); }
    off[k + 1] = (uint32_t)csc.size();
  }
  zkp_batch_statement bs{m, ns, ni, nc, lhs.data(), off.data(), csc.data(), cpt.data()};
  uint8_t out[32];
  int status = 1;
  int rc = zkp_batch_check(ctx, &bs, N, minus_c.data(), responses, weights16, common, inst, commitments, out, &status, coeffs);
  if (rc) return rc;
  if (status) return ZKP_TB_VERIFICATION_FAILURE;                      // some point failed to decompress -> None
  static const uint8_t zero[32] = {0};
  return std::memcmp(out, zero, 32) == 0 ? ZKP_TB_OK : ZKP_TB_VERIFICATION_FAILURE;   // :230-234
}

int zkp_batch_verify(zkp_ctx* ctx, const zkp_statement* st, uint32_t N, uint32_t n_transcripts, uint8_t* ts, const uint8_t* inst,
                     const uint8_t* common, const uint8_t* commitments, const uint8_t* responses, const uint8_t* weights16,
                     int n_threads) {
  return zkp_batch_verify_coeffs(ctx, st, N, n_transcripts, ts, inst, common, commitments, responses, weights16, n_threads, nullptr);
}

int zkp_batch_verify_many(zkp_ctx* ctx, const zkp_statement* stp, uint32_t K, uint32_t N_each, uint32_t n_transcripts, uint8_t* ts, const uint8_t* inst,
                          const uint8_t* common, const uint8_t* commitments, const uint8_t* responses, const uint8_t* weights16, int n_threads,
                          int* verdicts) {
  if (!stp || !verdicts || K == 0 || N_each == 0 || (uint64_t)K * N_each > 0x7fffffffull) return ZKP_TB_BAD_STATEMENT;
  const uint32_t N = K * N_each;
  if (n_transcripts != N) return ZKP_TB_BATCH_SIZE_MISMATCH;           // batch_verifier.rs:72-74
  if (!ts) return ZKP_TB_BAD_STATEMENT;
  const zkp_statement& st = *stp;
  const uint32_t m = (uint32_t)st.secrets.size(), nc = (uint32_t)st.cons.size(), ni = st.ni;
  std::vector<uint8_t> own_w;
  if (!weights16 && ctx && nc && use_fused(ts, N)) {                   // weights drawn on the device from 40 bytes of getrandom() (see zkp_batch_verify)
    uint8_t seed[40];
    if (!os_entropy(seed, sizeof(seed))) return ZKP_TB_NO_ENTROPY;
    FusedView fv(st);
    const int rc = zkp_fused_batch_verify_many_seeded(ctx, &fv.fs, K, N_each, ts, inst, common, commitments, responses, seed, verdicts);
    if (rc) return rc;
    for (uint32_t b = 0; b < K; ++b) verdicts[b] = verdicts[b] ? ZKP_TB_VERIFICATION_FAILURE : ZKP_TB_OK;
    return ZKP_TB_OK;
  }
  if (!weights16) { own_w.resize(16 * (size_t)N * nc); if (!os_random(own_w.data(), own_w.size())) return ZKP_TB_NO_ENTROPY; weights16 = own_w.data(); }
  if (ctx && use_fused(ts, N)) {
    FusedView fv(st);
    const int rc = zkp_fused_batch_verify_many(ctx, &fv.fs, K, N_each, ts, inst, common, commitments, responses, weights16, verdicts, nullptr);
    if (rc) return rc;
    for (uint32_t b = 0; b < K; ++b) verdicts[b] = verdicts[b] ? ZKP_TB_VERIFICATION_FAILURE : ZKP_TB_OK;
    return ZKP_TB_OK;
  }
  // one by one: batch b's columns of the instance rows and of the weights, gathered into arrays of its own
  std::vector<uint8_t> inst_b(32 * (size_t)ni * N_each), w_b(16 * (size_t)nc * N_each);
  for (uint32_t b = 0; b < K; ++b) {
    const size_t j0 = (size_t)b * N_each;
    for (uint32_t r = 0; r < ni; ++r) std::memcpy(inst_b.data() + 32 * (size_t)r * N_each, inst + 32 * ((size_t)r * N + j0), 32 * (size_t)N_each);
    for (uint32_t k = 0; k < nc; ++k) std::memcpy(w_b.data() + 16 * (size_t)k * N_each, weights16 + 16 * ((size_t)k * N + j0), 16 * (size_t)N_each);
    const int rc = zkp_batch_verify(ctx, stp, N_each, N_each, ts + TB * j0, inst_b.data(), common, commitments + 32 * j0 * nc, responses + 32 * j0 * m,
                                    w_b.data(), n_threads);
    if (rc < 0 || rc == ZKP_TB_BATCH_SIZE_MISMATCH) return rc;
    verdicts[b] = rc;
  }
  return ZKP_TB_OK;
}

int zkp_batch_verify_locate(zkp_ctx* ctx, const zkp_statement* st, uint32_t N, uint32_t n_transcripts, uint8_t* ts, const uint8_t* inst,
                            const uint8_t* common, const uint8_t* commitments, const uint8_t* responses, const uint8_t* weights16,
                            int n_threads, uint8_t* results) {
  if (!st || !results || (N && !ts)) return ZKP_TB_BAD_STATEMENT;
  if (n_transcripts != N) return ZKP_TB_BATCH_SIZE_MISMATCH;           // batch_verifier.rs:72-74
  std::memset(results, 0, N);
  const std::vector<uint8_t> saved(ts, ts + TB * (size_t)N);           // the per-proof pass starts where the batch check started
  const int rc = zkp_batch_verify(ctx, st, N, n_transcripts, ts, inst, common, commitments, responses, weights16, n_threads);
  if (rc != ZKP_TB_VERIFICATION_FAILURE || N == 0) return rc;
  std::vector<uint8_t> again(saved);
  // fresh per-proof weights (weights16 == NULL: from the OS): the batch weights are laid out [constraint][proof], these [proof][constraint]
  const int rc2 = zkp_verify_batchable_each(ctx, st, N, again.data(), inst, common, commitments, responses, nullptr, n_threads, results);
  if (rc2 < 0) return rc2;                                              // infrastructure failure: nothing may be trusted
  return ZKP_TB_VERIFICATION_FAILURE;
}

}  // extern "C"
