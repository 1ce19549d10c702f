// Host build of the transcript-program compiler and interpreter (zkp_amd/csrc/merlin_prog.h, the code the GPU kernel
// runs per lane) checked against the host Merlin implementation (zkp_amd/csrc/host/merlin.cpp) on random statements:
// the prover's whole transcript life (prefix, point appends, rng clone + rekey + finalize + fill, commitments,
// challenge) and the verifier's validating appends.  Built by tests/test_host_field.py.
#include "../../zkp_amd/csrc/merlin_prog.h"
#include "../../zkp_amd/csrc/host/merlin.hpp"
#include <cstdio>
#include <random>
#include <string>
#include <vector>
using namespace zkp;
using zkp::host::Transcript;

extern "C" int t_tr_selftest(uint32_t seed, uint32_t N, uint32_t m, uint32_t np, uint32_t nc, uint32_t zero_at) {
  std::mt19937_64 rng(seed);
  auto rand_label = [&](size_t maxlen) {
    std::string s(1 + rng() % maxlen, 'a');
    for (auto& ch : s) ch = (char)('a' + rng() % 26);
    return s;
  };
  const bool long_labels = seed >= 1000;                  // labels longer than a STROBE block (166 bytes)
  const std::string tlabel = rand_label(40), plabel = rand_label(long_labels ? 500 : 30);
  std::vector<std::string> sl(m), pl(np);
  for (auto& s : sl) s = rand_label(long_labels ? 300 : 12);
  for (auto& s : pl) s = rand_label(long_labels ? 300 : 12);
  const std::string pre = rand_label(200);                 // a message appended before the proof: varies pos
  std::vector<uint8_t> secrets(32 * (size_t)N * m), pts(32 * (size_t)np * N), ent(32 * (size_t)N), coms(32 * (size_t)N * nc), msg(32 * (size_t)N);
  for (auto* v : {&secrets, &pts, &ent, &coms, &msg}) for (auto& b : *v) b = (uint8_t)rng();
  if (zero_at < N && np) memset(&pts[32 * ((size_t)(np - 1) * N + zero_at)], 0, 32);
  const uint32_t n_const = np / 3;                         // the first n_const points are shared ("common"): proof 0's value
  // ---- expected: host Merlin, proof by proof ----
  std::vector<uint8_t> blobs(208 * (size_t)N), want_blobs(208 * (size_t)N), want_wide(64 * (size_t)N * m), want_chal(64 * (size_t)N);
  std::vector<uint8_t> want_fail(N, 0);
  for (uint32_t j = 0; j < N; ++j) {
    Transcript t(tlabel);
    t.append_message("pre", pre.data(), pre.size());
    t.append_message("per-proof", &msg[32 * (size_t)j], 32);       // different state per proof, same position
    t.to_bytes(&blobs[208 * (size_t)j]);
    t.domain_sep(plabel.c_str());
    for (auto& s : sl) t.append_scalar_var(s.c_str());
    for (uint32_t p = 0; p < np; ++p) {
      const uint8_t* enc = &pts[32 * ((size_t)p * N + (p < n_const ? 0 : j))];
      if (p + 1 == np) { if (!t.validate_and_append_point_var(pl[p].c_str(), enc)) { want_fail[j] = 1; t.append_point_var(pl[p].c_str(), enc); } }
      else t.append_point_var(pl[p].c_str(), enc);
    }
    zkp::host::TranscriptRng r = t.build_rng();
    for (uint32_t i = 0; i < m; ++i) r.rekey_with_witness_bytes("", &secrets[32 * ((size_t)j * m + i)], 32);
    r.finalize(&ent[32 * (size_t)j]);
    for (uint32_t i = 0; i < m; ++i) r.fill_bytes(&want_wide[64 * ((size_t)j * m + i)], 64);
    for (uint32_t k = 0; k < nc; ++k) t.append_blinding_commitment(pl[k % (np ? np : 1)].c_str(), &coms[32 * ((size_t)j * nc + k)]);
    t.challenge_bytes("chal", &want_chal[64 * (size_t)j], 64);
    t.to_bytes(&want_blobs[208 * (size_t)j]);
  }
  // ---- compiled program ----
  TrCompiler c(blobs[200], blobs[201], blobs[202]);
  enum { B_SECRETS = 0, B_PTS = 1, B_ENT = 2, B_COMS = 3 };
  enum { D_WIDE = 0, D_CHAL = 1 };
  c.domain_sep(plabel.c_str());
  for (auto& s : sl) c.append_scalar_var(s.c_str());
  for (uint32_t p = 0; p < np; ++p) {
    if (p < n_const) c.append_point_var(pl[p].c_str(), &pts[32 * (size_t)p * N]);
    else c.append_point_var_var(pl[p].c_str(), tr_ref{B_PTS, 32, 32 * (uint64_t)p * N}, p + 1 == np);
  }
  c.save();
  for (uint32_t i = 0; i < m; ++i) c.rng_rekey_with_witness_var("", tr_ref{B_SECRETS, 32 * m, 32ull * i}, 32);
  c.rng_finalize_var(tr_ref{B_ENT, 32, 0});
  for (uint32_t i = 0; i < m; ++i) c.rng_fill_bytes(tr_ref{D_WIDE, 64 * m, 64ull * i}, 64);
  c.restore();
  for (uint32_t k = 0; k < nc; ++k) c.append_blinding_commitment_var(pl[k % (np ? np : 1)].c_str(), tr_ref{B_COMS, 32 * nc, 32ull * k}, false);
  c.get_challenge_wide("chal", tr_ref{D_CHAL, 64, 0});
  uint8_t tail[3];
  const std::vector<tr_op> prog = c.finish(tail);
  std::vector<uint8_t> got_wide(64 * (size_t)N * m + 8), got_chal(64 * (size_t)N + 8);
  tr_bufs bufs{};
  bufs.src[B_SECRETS] = secrets.data(); bufs.src[B_PTS] = pts.data(); bufs.src[B_ENT] = ent.data(); bufs.src[B_COMS] = coms.data();
  bufs.dst[D_WIDE] = got_wide.data(); bufs.dst[D_CHAL] = got_chal.data();
  int bad = 0;
  for (uint32_t j = 0; j < N; ++j) {
    uint64_t S[25], saved[25];
    memcpy(S, &blobs[208 * (size_t)j], 200);
    uint32_t failed = 0;
    tr_run_one(prog.data(), (uint32_t)prog.size(), c.tables().data(), j, bufs, S, 1, saved, 1, &failed);
    if (memcmp(S, &want_blobs[208 * (size_t)j], 200) != 0) bad |= 1;
    if (tail[0] != want_blobs[208 * (size_t)j + 200] || tail[1] != want_blobs[208 * (size_t)j + 201] || tail[2] != want_blobs[208 * (size_t)j + 202]) bad |= 2;
    if ((failed != 0) != (want_fail[j] != 0)) bad |= 4;
  }
  if (memcmp(got_wide.data(), want_wide.data(), want_wide.size()) != 0) bad |= 8;
  if (memcmp(got_chal.data(), want_chal.data(), want_chal.size()) != 0) bad |= 16;
  // ---- the same program in step form (round 6: assemble + chain; tr_steps_build / tr_steps_run_one) ----
  const tr_step_prog sp = tr_steps_build(prog, c.tables());
  std::vector<uint8_t> s_wide(64 * (size_t)N * m + 8), s_chal(64 * (size_t)N + 8);
  bufs.dst[D_WIDE] = s_wide.data(); bufs.dst[D_CHAL] = s_chal.data();
  size_t perms = 0;
  for (const tr_step& st : sp.steps) perms += (st.flags & TS_PERMUTE) ? 1 : 0;
  if (perms != c.permutations()) bad |= 32;
  for (const tr_step& st : sp.steps) if (st.emit_n > 64) bad |= 32;                 // (the chain kernel fetches a step's emit operations with one load)
  for (uint32_t j = 0; j < N; ++j) {
    uint64_t S[25], saved[25];
    memcpy(S, &blobs[208 * (size_t)j], 200);
    uint32_t failed = 0;
    tr_steps_run_one(sp, j, bufs, S, saved, &failed);
    if (memcmp(S, &want_blobs[208 * (size_t)j], 200) != 0) bad |= 64;
    if ((failed != 0) != (want_fail[j] != 0)) bad |= 128;
  }
  if (memcmp(s_wide.data(), want_wide.data(), want_wide.size()) != 0) bad |= 256;
  if (memcmp(s_chal.data(), want_chal.data(), want_chal.size()) != 0) bad |= 512;
  return bad ? -bad : (int)prog.size();
}

// tr_steps_build folds a step without a permutation into its successor only when nothing of the first is cleared by the second's keep words.  The compiler never
// emits such a pair (bytes only advance inside a block), so it is built by hand here: step 1 absorbs 8 bytes into word 3, step 2 OVERWRITES 4 bytes of word 3
// (a KEY operation on bytes 2..5) and permutes.  The step form must give what the operation list gives (tr_run_one), merged or not, and must NOT have merged.
extern "C" int t_tr_merge_selftest() {
  std::vector<uint8_t> a(64), b(64);
  for (size_t i = 0; i < 64; ++i) { a[i] = (uint8_t)(17 * i + 3); b[i] = (uint8_t)(29 * i + 5); }
  auto word_op = [](uint8_t w, uint8_t nb, uint8_t lb, uint8_t buf, uint64_t off, bool overwrite) {
    tr_op_wide o{};
    const uint64_t mask = (nb >= 8 ? ~0ULL : ((1ULL << (8 * nb)) - 1)) << (8 * lb);
    o.keep = overwrite ? ~mask : ~0ULL;
    if (overwrite) o.flags |= TR_OVERWRITE;
    o.w = w; o.nb = nb; o.lb = lb; o.src_buf = (uint8_t)(buf + 1); o.src_stride = 32; o.src_off = off;
    return tr_pack(o);
  };
  auto apply_op = [](uint64_t table, bool permute) {
    tr_op_wide o{};
    o.keep = ~0ULL;
    o.flags = (uint8_t)(TR_APPLY | (permute ? TR_PERMUTE : 0));
    o.src_off = table;
    return tr_pack(o);
  };
  std::vector<uint64_t> tables(2 * TR_TABLE_WORDS, 0);
  for (int t = 0; t < 2; ++t) for (int w = 0; w < 21; ++w) { tables[t * TR_TABLE_WORDS + w] = ~0ULL; tables[t * TR_TABLE_WORDS + 21 + w] = 0x0101010101010101ULL * (uint64_t)(t + 1) * (w == 5); }
  int bad = 0;
  for (int variant = 0; variant < 2; ++variant) {           // 0: the second step overwrites bytes the first absorbed (no merge); 1: it touches another word (merge)
    std::vector<tr_op> ops;
    ops.push_back(word_op(3, 8, 0, 0, 0, false));
    ops.push_back(apply_op(0, false));
    ops.push_back(word_op(variant == 0 ? 3 : 4, 4, 2, 1, 8, true));
    ops.push_back(apply_op(1, true));
    const tr_step_prog sp = tr_steps_build(ops, tables);
    size_t real = sp.steps.size() ? sp.steps.size() - 1 : 0;     // (the sentinel)
    if (variant == 0 && real != 2) bad |= 1;
    if (variant == 1 && real != 1) bad |= 2;
    tr_bufs bufs{};
    bufs.src[0] = a.data(); bufs.src[1] = b.data();
    for (uint64_t j = 0; j < 2; ++j) {
      uint64_t S1[25], S2[25], sv[25];
      for (int i = 0; i < 25; ++i) S1[i] = S2[i] = 0x9e3779b97f4a7c15ULL * (uint64_t)(i + 1 + 31 * j);
      uint32_t f1 = 0, f2 = 0;
      tr_run_one(ops.data(), (uint32_t)ops.size(), tables.data(), j, bufs, S1, 1, sv, 1, &f1);
      tr_steps_run_one(sp, j, bufs, S2, sv, &f2);
      if (memcmp(S1, S2, 200) != 0) bad |= 4 << variant;
    }
  }
  return bad ? -bad : 1;
}

// The pairing rule of variable-time statement jobs (zkp_amd/csrc/stmt_pairs.h: which verifier terms share a chain of doublings), as the plan calls it.
#include "../../zkp_amd/csrc/stmt_pairs.h"
extern "C" int t_pair_terms(const uint32_t* toff, const uint32_t* tpt, uint32_t T1, uint32_t nc, uint32_t ns, uint32_t np, uint32_t* out) {
  const std::vector<uint32_t> p = zkp::pair_terms(toff, tpt, T1, nc, ns, np);
  if (p.empty()) return 0;
  for (uint32_t i = 0; i < T1; ++i) out[i] = p[i];
  int absorbed = 0;
  for (uint32_t i = 0; i < T1; ++i) absorbed += zkp::stmt_absorbed(p.data(), i) ? 1 : 0;
  return absorbed;
}
