// Batched Merlin transcripts on the GPU (src/toolbox/mod.rs:165-228 over merlin 2.x / STROBE-128).
//
// Every proof of one batch call runs the SAME sequence of transcript operations (same labels, same lengths); only
// the 32-byte values differ.  STROBE's byte position is therefore known on the host, and a transcript "program" can
// be compiled once per (statement, batch size): per Keccak block
//   * word operations that move this proof's bytes:  emit bytes of state[w] (PRF output), or
//        state[w] = (state[w] & keep) ^ (bytes from this proof's inputs << 8*lb)             (absorb / key)
//   * one APPLY operation: state[w] = (state[w] & KEEP[w]) ^ CX[w] for the 21 rate words, with everything constant
//     (labels, length prefixes, STROBE framing and padding, zeroing of squeezed bytes) folded into one 336-byte table,
//     then the permutation.
// The GPU interprets the list with the state in LDS columns (dynamic word index without scratch), no divergence.
//
// TrCompiler mirrors the host classes in host/merlin.hpp method for method; tests/test_host_field.py runs compiled
// programs through tr_run_one on the CPU and compares with the host Merlin byte for byte.
#pragma once
#include <stdint.h>
#include <string.h>
#include "fe25519.h"   // ZKP_HD

namespace zkp {

constexpr int TR_MAX_BUFS = 4;
constexpr uint8_t TR_PERMUTE = 1, TR_SAVE = 2, TR_RESTORE = 4, TR_CHECK_NONZERO = 8, TR_APPLY = 16;
constexpr int TR_TABLE_WORDS = 42;   // per block: keep[21] | cx[21] over the 168 rate + padding bytes

constexpr uint8_t TR_OVERWRITE = 32;  // (word operation) the source bytes replace the state bytes instead of XORing into them

// One operation as the interpreters read it: 16 bytes, so that a wavefront fetches 64 of them with one coalesced load.
//   ctl = flags[5:0] | w[10:6] | nb[14:11] | lb[17:15] | dnb[21:18] | dlb[24:22] | src_buf[27:25] | dst_buf[30:28]
//   stride / off: bytes per proof and byte offset of the source or destination (an operation has at most one of them);
//   for TR_APPLY, off = index of the block's constant table.
struct tr_op { uint32_t ctl, stride; uint64_t off; };
static_assert(sizeof(tr_op) == 16, "tr_op layout");
struct tr_fields {
  uint32_t flags, w, nb, lb, dnb, dlb, src_buf, dst_buf;   // src_buf / dst_buf: 0 = none, else 1 + index into tr_bufs
  uint32_t stride;
  uint64_t off, keep;
};
ZKP_HD tr_fields tr_unpack(uint32_t ctl, uint32_t stride, uint64_t off) {
  tr_fields f;
  f.flags = ctl & 63u; f.w = (ctl >> 6) & 31u; f.nb = (ctl >> 11) & 15u; f.lb = (ctl >> 15) & 7u;
  f.dnb = (ctl >> 18) & 15u; f.dlb = (ctl >> 22) & 7u; f.src_buf = (ctl >> 25) & 7u; f.dst_buf = (ctl >> 28) & 7u;
  f.stride = stride;
  f.off = off;
  const uint64_t mask = (f.nb >= 8 ? ~0ULL : ((1ULL << (8 * f.nb)) - 1)) << (8 * f.lb);
  f.keep = (f.flags & TR_OVERWRITE) ? ~mask : ~0ULL;
  return f;
}
// the compiler's working form
struct tr_op_wide {
  uint64_t keep = ~0ULL;
  uint64_t src_off = 0, dst_off = 0;
  uint32_t src_stride = 0, dst_stride = 0;
  uint8_t w = 0, nb = 0, lb = 0, src_buf = 0, dst_buf = 0, flags = 0, dnb = 0, dlb = 0;
};
inline tr_op tr_pack(const tr_op_wide& o) {
  tr_op p;
  p.ctl = (uint32_t)o.flags | (uint32_t)o.w << 6 | (uint32_t)o.nb << 11 | (uint32_t)o.lb << 15 | (uint32_t)o.dnb << 18 |
          (uint32_t)o.dlb << 22 | (uint32_t)o.src_buf << 25 | (uint32_t)o.dst_buf << 28;
  p.stride = o.dst_buf ? o.dst_stride : o.src_stride;
  p.off = o.dst_buf ? o.dst_off : o.src_off;
  return p;
}

struct tr_bufs {
  const uint8_t* src[TR_MAX_BUFS];
  uint8_t* dst[TR_MAX_BUFS];
};

// 64-bit rotation by a compile-time amount.  On the GPU: two full-rate v_alignbit_b32 on the 32-bit halves (the
// compiler's own choice, 64-bit shifts + or, runs at a quarter of that rate).
template <int N>
ZKP_HD uint64_t tr_rotl_c(uint64_t v) {
#ifdef __HIP_DEVICE_COMPILE__
  const uint32_t lo = (uint32_t)v, hi = (uint32_t)(v >> 32);
  constexpr int n = N & 31;
  uint32_t rlo, rhi;
  if (n == 0) { rlo = lo; rhi = hi; }
  else { rhi = __builtin_amdgcn_alignbit(hi, lo, 32 - n); rlo = __builtin_amdgcn_alignbit(lo, hi, 32 - n); }
  return (N & 32) ? ((uint64_t)rlo << 32 | rhi) : ((uint64_t)rhi << 32 | rlo);
#else
  return (v << N) | (v >> (64 - N));
#endif
}
#define tr_rotl(v, n) tr_rotl_c<n>(v)

// a ^ b ^ c ^ d ^ e: on the GPU two v_bitop3_b32 (truth table 0x96) per 32-bit half instead of four v_xor_b32
ZKP_HD uint32_t tr_xor5_32(uint32_t a, uint32_t b, uint32_t c, uint32_t d, uint32_t e) {
#ifdef __HIP_DEVICE_COMPILE__
  return (uint32_t)__builtin_amdgcn_bitop3_b32((uint32_t)__builtin_amdgcn_bitop3_b32(a, b, c, 0x96), d, e, 0x96);
#else
  return a ^ b ^ c ^ d ^ e;
#endif
}
ZKP_HD uint64_t tr_xor5(uint64_t a, uint64_t b, uint64_t c, uint64_t d, uint64_t e) {
  return (uint64_t)tr_xor5_32((uint32_t)(a >> 32), (uint32_t)(b >> 32), (uint32_t)(c >> 32), (uint32_t)(d >> 32), (uint32_t)(e >> 32)) << 32 |
         tr_xor5_32((uint32_t)a, (uint32_t)b, (uint32_t)c, (uint32_t)d, (uint32_t)e);
}

// a ^ (~b & c)  (chi): one v_bitop3_b32 (truth table 0xD2 = 0xF0 ^ (~0xCC & 0xAA)) per half; left to itself the compiler
// builds the 64-bit form from v_bfi_b32 + v_xor_b32
ZKP_HD uint32_t tr_chi32(uint32_t a, uint32_t b, uint32_t c) {
#ifdef __HIP_DEVICE_COMPILE__
  return (uint32_t)__builtin_amdgcn_bitop3_b32(a, b, c, 0xD2);
#else
  return a ^ (~b & c);
#endif
}
ZKP_HD uint64_t tr_chi(uint64_t a, uint64_t b, uint64_t c) {
  return (uint64_t)tr_chi32((uint32_t)(a >> 32), (uint32_t)(b >> 32), (uint32_t)(c >> 32)) << 32 | tr_chi32((uint32_t)a, (uint32_t)b, (uint32_t)c);
}

// The APPLY operation on a strided column (S[i * stride]): the block's constant table, then (permute) Keccak-f[1600]
// with the 25 lanes held in registers for the 24 rounds.
ZKP_HD void tr_apply_block(uint64_t* S, int stride, const uint64_t* tbl, bool permute) {
  const uint64_t RC[24] = {
      0x0000000000000001ULL, 0x0000000000008082ULL, 0x800000000000808aULL, 0x8000000080008000ULL, 0x000000000000808bULL,
      0x0000000080000001ULL, 0x8000000080008081ULL, 0x8000000000008009ULL, 0x000000000000008aULL, 0x0000000000000088ULL,
      0x0000000080008009ULL, 0x000000008000000aULL, 0x000000008000808bULL, 0x800000000000008bULL, 0x8000000000008089ULL,
      0x8000000000008003ULL, 0x8000000000008002ULL, 0x8000000000000080ULL, 0x000000000000800aULL, 0x800000008000000aULL,
      0x8000000080008081ULL, 0x8000000000008080ULL, 0x0000000080000001ULL, 0x8000000080008008ULL};
  uint64_t a00 = S[0 * stride], a10 = S[1 * stride], a20 = S[2 * stride], a30 = S[3 * stride], a40 = S[4 * stride];
  uint64_t a01 = S[5 * stride], a11 = S[6 * stride], a21 = S[7 * stride], a31 = S[8 * stride], a41 = S[9 * stride];
  uint64_t a02 = S[10 * stride], a12 = S[11 * stride], a22 = S[12 * stride], a32 = S[13 * stride], a42 = S[14 * stride];
  uint64_t a03 = S[15 * stride], a13 = S[16 * stride], a23 = S[17 * stride], a33 = S[18 * stride], a43 = S[19 * stride];
  uint64_t a04 = S[20 * stride], a14 = S[21 * stride], a24 = S[22 * stride], a34 = S[23 * stride], a44 = S[24 * stride];
#define TR_T(v, i) v = (v & tbl[i]) ^ tbl[21 + i]
  TR_T(a00, 0); TR_T(a10, 1); TR_T(a20, 2); TR_T(a30, 3); TR_T(a40, 4); TR_T(a01, 5); TR_T(a11, 6);
  TR_T(a21, 7); TR_T(a31, 8); TR_T(a41, 9); TR_T(a02, 10); TR_T(a12, 11); TR_T(a22, 12); TR_T(a32, 13);
  TR_T(a42, 14); TR_T(a03, 15); TR_T(a13, 16); TR_T(a23, 17); TR_T(a33, 18); TR_T(a43, 19); TR_T(a04, 20);
#undef TR_T
#pragma unroll 1
  for (int round = 0; round < (permute ? 24 : 0); ++round) {
    const uint64_t c0 = tr_xor5(a00, a01, a02, a03, a04), c1 = tr_xor5(a10, a11, a12, a13, a14), c2 = tr_xor5(a20, a21, a22, a23, a24),
                   c3 = tr_xor5(a30, a31, a32, a33, a34), c4 = tr_xor5(a40, a41, a42, a43, a44);
    const uint64_t d0 = c4 ^ tr_rotl(c1, 1), d1 = c0 ^ tr_rotl(c2, 1), d2 = c1 ^ tr_rotl(c3, 1), d3 = c2 ^ tr_rotl(c4, 1),
                   d4 = c3 ^ tr_rotl(c0, 1);
    a00 ^= d0; a01 ^= d0; a02 ^= d0; a03 ^= d0; a04 ^= d0;
    a10 ^= d1; a11 ^= d1; a12 ^= d1; a13 ^= d1; a14 ^= d1;
    a20 ^= d2; a21 ^= d2; a22 ^= d2; a23 ^= d2; a24 ^= d2;
    a30 ^= d3; a31 ^= d3; a32 ^= d3; a33 ^= d3; a34 ^= d3;
    a40 ^= d4; a41 ^= d4; a42 ^= d4; a43 ^= d4; a44 ^= d4;
    const uint64_t b00 = a00,               b13 = tr_rotl(a01, 36), b21 = tr_rotl(a02, 3),  b34 = tr_rotl(a03, 41), b42 = tr_rotl(a04, 18);
    const uint64_t b02 = tr_rotl(a10, 1),   b10 = tr_rotl(a11, 44), b23 = tr_rotl(a12, 10), b31 = tr_rotl(a13, 45), b44 = tr_rotl(a14, 2);
    const uint64_t b04 = tr_rotl(a20, 62),  b12 = tr_rotl(a21, 6),  b20 = tr_rotl(a22, 43), b33 = tr_rotl(a23, 15), b41 = tr_rotl(a24, 61);
    const uint64_t b01 = tr_rotl(a30, 28),  b14 = tr_rotl(a31, 55), b22 = tr_rotl(a32, 25), b30 = tr_rotl(a33, 21), b43 = tr_rotl(a34, 56);
    const uint64_t b03 = tr_rotl(a40, 27),  b11 = tr_rotl(a41, 20), b24 = tr_rotl(a42, 39), b32 = tr_rotl(a43, 8),  b40 = tr_rotl(a44, 14);
    a00 = tr_chi(b00, b10, b20); a10 = tr_chi(b10, b20, b30); a20 = tr_chi(b20, b30, b40); a30 = tr_chi(b30, b40, b00); a40 = tr_chi(b40, b00, b10);
    a01 = tr_chi(b01, b11, b21); a11 = tr_chi(b11, b21, b31); a21 = tr_chi(b21, b31, b41); a31 = tr_chi(b31, b41, b01); a41 = tr_chi(b41, b01, b11);
    a02 = tr_chi(b02, b12, b22); a12 = tr_chi(b12, b22, b32); a22 = tr_chi(b22, b32, b42); a32 = tr_chi(b32, b42, b02); a42 = tr_chi(b42, b02, b12);
    a03 = tr_chi(b03, b13, b23); a13 = tr_chi(b13, b23, b33); a23 = tr_chi(b23, b33, b43); a33 = tr_chi(b33, b43, b03); a43 = tr_chi(b43, b03, b13);
    a04 = tr_chi(b04, b14, b24); a14 = tr_chi(b14, b24, b34); a24 = tr_chi(b24, b34, b44); a34 = tr_chi(b34, b44, b04); a44 = tr_chi(b44, b04, b14);
    a00 ^= RC[round];
  }
  S[0 * stride] = a00; S[1 * stride] = a10; S[2 * stride] = a20; S[3 * stride] = a30; S[4 * stride] = a40;
  S[5 * stride] = a01; S[6 * stride] = a11; S[7 * stride] = a21; S[8 * stride] = a31; S[9 * stride] = a41;
  S[10 * stride] = a02; S[11 * stride] = a12; S[12 * stride] = a22; S[13 * stride] = a32; S[14 * stride] = a42;
  S[15 * stride] = a03; S[16 * stride] = a13; S[17 * stride] = a23; S[18 * stride] = a33; S[19 * stride] = a43;
  S[20 * stride] = a04; S[21 * stride] = a14; S[22 * stride] = a24; S[23 * stride] = a34; S[24 * stride] = a44;
}

// bufs.src[i] / bufs.dst[i] for a run-time (wave-uniform) i: a select chain, so that the table can be a by-value kernel
// argument without being copied to scratch for indexing
ZKP_HD const uint8_t* tr_src_ptr(const tr_bufs& b, uint32_t i) {
  const uint8_t* p = b.src[0];
#pragma unroll
  for (int k = 1; k < TR_MAX_BUFS; ++k) p = (i == (uint32_t)k) ? b.src[k] : p;
  return p;
}
ZKP_HD uint8_t* tr_dst_ptr(const tr_bufs& b, uint32_t i) {
  uint8_t* p = b.dst[0];
#pragma unroll
  for (int k = 1; k < TR_MAX_BUFS; ++k) p = (i == (uint32_t)k) ? b.dst[k] : p;
  return p;
}

ZKP_HD uint64_t tr_bytemask(uint32_t nb) { return nb >= 8 ? ~0ULL : ((1ULL << (8 * nb)) - 1); }

// One operation of a program for proof j on the state column S (stride in words).  `saved` = this proof's clone slot
// (saved[i * saved_stride]); *failed is set when a checked encoding is all zero (mod.rs:191, :215).
// This is the reference semantics of a program: the host tests run it (tr_run_one), the GPU runs it one lane per proof in
// k_transcript_run1 and, on 32-bit half-words split across lane pairs, in k_transcript_run (fused_flows.h).
ZKP_HD void tr_exec_op(const tr_fields& op, const uint64_t* tables, uint64_t j, const tr_bufs& bufs, uint64_t* S, int stride,
                       uint64_t* saved, size_t saved_stride, uint32_t* failed, bool store) {
  if (op.flags & TR_RESTORE)
    for (int i = 0; i < 25; ++i) S[i * stride] = saved[i * saved_stride];
  if (op.flags & TR_CHECK_NONZERO) {
    const uint64_t* p = reinterpret_cast<const uint64_t*>(tr_src_ptr(bufs, op.src_buf - 1u) + j * op.stride + op.off);
    if ((p[0] | p[1] | p[2] | p[3]) == 0) *failed = 1;
  }
  if (op.dst_buf && store) {
    uint8_t* d = tr_dst_ptr(bufs, op.dst_buf - 1u) + j * op.stride + op.off;
    const uint64_t e = S[op.w * stride] >> (8 * op.dlb);
    if (op.dnb == 8) {
      __builtin_memcpy(d, &e, 8);
    } else if (op.dnb == 4) {
      const uint32_t e4 = (uint32_t)e;
      __builtin_memcpy(d, &e4, 4);
    } else {
      for (uint32_t i = 0; i < op.dnb; ++i) d[i] = (uint8_t)(e >> (8 * i));
    }
  }
  if (op.src_buf && !(op.flags & TR_CHECK_NONZERO)) {
    const uint64_t addr = j * op.stride + op.off;
    const uint32_t sh = (uint32_t)(addr & 7);
    const uint64_t* p = reinterpret_cast<const uint64_t*>(tr_src_ptr(bufs, op.src_buf - 1u) + (addr - sh));
    uint64_t x = p[0] >> (8 * sh);
    if (sh + op.nb > 8) x |= p[1] << (64 - 8 * sh);
    x = (x & tr_bytemask(op.nb)) << (8 * op.lb);
    S[op.w * stride] = (S[op.w * stride] & op.keep) ^ x;
  }
  if (op.flags & TR_APPLY) tr_apply_block(S, stride, tables + (size_t)TR_TABLE_WORDS * op.off, (op.flags & TR_PERMUTE) != 0);
  if ((op.flags & TR_SAVE) && store)
    for (int i = 0; i < 25; ++i) saved[i * saved_stride] = S[i * stride];
}
ZKP_HD void tr_run_one(const tr_op* prog, uint32_t n_ops, const uint64_t* tables, uint64_t j, const tr_bufs& bufs, uint64_t* S,
                       int stride, uint64_t* saved, size_t saved_stride, uint32_t* failed) {
  for (uint32_t q = 0; q < n_ops; ++q) tr_exec_op(tr_unpack(prog[q].ctl, prog[q].stride, prog[q].off), tables, j, bufs, S, stride, saved, saved_stride, failed, true);
}

}  // namespace zkp

namespace zkp {
// ---- round 6: the same program as STEPS -- a wide "assemble" pass plus a chain of bare permutations ---------------------------------
// Measured (tools/microbench/keccak_lat.hip, profiles/r06_keccak_microbench.txt): a lone wavefront needs 5.5 - 6.2 us per Keccak-f[1600] in the
// lane-pair layout, but the interpreter above spends 9.4 - 14 us per permutation: every word operation is a DEPENDENT global load (address -> load ->
// shift -> LDS read-modify-write, ~1 us each, ~480 of them per CMZ proof) in a chain whose only true dependency is the hash state.  None of those
// loads depends on the state.  So the operation list is regrouped per flush unit ("step": everything up to and including one APPLY):
//   image   img[step][w][proof] = CX[w] ^ (the per-proof bytes of the step's word operations, shifted into place)       -- k_transcript_assemble,
//           a lane per proof and seven words, every load independent, the whole chip busy for ~15 us per 4096 CMZ proofs;
//   chain   per step: [restore] -> [emit PRF bytes of the CURRENT state] -> state[w] = (state[w] & KEEP[w]) ^ img[w] -> [permute] -> [save]
//           with the state in 25 registers per lane, the next step's image (and keep words) prefetched under the permutation -- k_transcript_chain.
// KEEP[w] = the table's keep word AND the keep masks of the step's overwriting operations (their bytes are disjoint from every other
// operation's and from the table's cleared bytes, so the nested ((s & k1) ^ x1) & k2 ... collapses).  Identity checks (mod.rs:191, :215) do not
// touch the state at all: they go to the assemble pass.  Semantics = tr_steps_run_one below, which the host tests compare with tr_run_one.
constexpr uint32_t TS_RESTORE = 1, TS_IMG = 2, TS_KEEP = 4, TS_PERMUTE = 8, TS_SAVE = 16, TS_EMIT = 32;
struct tr_step { uint32_t flags, img, emit_lo, emit_n; };      // img: index of the step's image / keep row; emit ops [emit_lo, emit_lo + emit_n) of tr_step_prog::emit
static_assert(sizeof(tr_step) == 16, "tr_step layout");

}  // namespace zkp
#include <vector>
namespace zkp {

struct tr_step_prog {
  std::vector<tr_step> steps;
  std::vector<tr_op> emit;          // PRF-output operations (dst_buf != 0), executed by the chain from the step's incoming state
  std::vector<tr_op> src;           // word operations with a per-proof source, grouped by (image, word)
  std::vector<uint32_t> src_off;    // [n_img * 21 + 1]: operations src[src_off[k] .. src_off[k + 1]) feed image word k = img * 21 + w
  std::vector<uint64_t> cx, keep;   // [n_img * 21] each
  std::vector<tr_op> chk;           // identity checks
  uint32_t n_img = 0;
};

inline tr_step_prog tr_steps_build(const std::vector<tr_op>& ops, const std::vector<uint64_t>& tables) {
  // 1. logical steps: everything up to and including one APPLY (or a lone marker)
  struct L { uint32_t flags = 0; std::vector<tr_op> emit; std::vector<tr_op> src[21]; uint64_t keep[21], cx[21]; bool any = false;
             L() { for (int w = 0; w < 21; ++w) { keep[w] = ~0ULL; cx[w] = 0; } } };
  std::vector<L> ls;
  std::vector<tr_op> chk;
  L cur;
  auto close = [&](const uint64_t* tbl, bool permute) {
    for (int w = 0; w < 21 && tbl; ++w) { cur.keep[w] &= tbl[w]; cur.cx[w] = tbl[21 + w]; }
    if (permute) cur.flags |= TS_PERMUTE;
    ls.push_back(cur);
    cur = L();
  };
  for (const tr_op& o : ops) {
    const tr_fields f = tr_unpack(o.ctl, o.stride, o.off);
    if (f.flags & TR_RESTORE) {
      if (cur.any) close(nullptr, false);
      cur.flags |= TS_RESTORE;
      cur.any = true;
    }
    if (f.flags & TR_CHECK_NONZERO) { chk.push_back(o); continue; }
    if (f.dst_buf) { cur.emit.push_back(o); cur.any = true; }
    if (f.src_buf) { cur.src[f.w].push_back(o); cur.keep[f.w] &= f.keep; cur.any = true; }
    if (f.flags & TR_APPLY) close(tables.data() + (size_t)TR_TABLE_WORDS * f.off, (f.flags & TR_PERMUTE) != 0);
    if (f.flags & TR_SAVE) {
      if (cur.any) close(nullptr, false);                  // (a restore or word operations without their APPLY: does not happen, handled all the same)
      if (ls.empty()) close(nullptr, false);
      ls.back().flags |= TS_SAVE;
    }
  }
  if (cur.any) close(nullptr, false);
  // 2. a step WITHOUT a permutation has nothing to hide its successor's image loads behind (the verifiers' validating appends flush before every identity
  // check: 36 such steps per CMZ batch-verification program, ~0.8 us of exposed load latency each).  It folds into the next step when that one reads
  // nothing in between (no restore, no PRF output) and nothing of the first is cleared by the second's keep words:
  //     ((s & k1) ^ x1) & k2) ^ x2  =  (s & k1 & k2) ^ x1 ^ x2      iff  x1 & ~k2 = 0  (bytes only ever advance inside a block, so it holds; checked all the same)
  auto mask_of = [](const tr_op& o) { const tr_fields f = tr_unpack(o.ctl, o.stride, o.off); return tr_bytemask(f.nb) << (8 * f.lb); };
  std::vector<L> ms;
  for (L& l : ls) {
    bool merged = false;
    if (!ms.empty()) {
      L& a = ms.back();
      bool ok = !(a.flags & (TS_PERMUTE | TS_SAVE)) && !(l.flags & TS_RESTORE) && l.emit.empty();
      for (int w = 0; w < 21 && ok; ++w) {
        if (a.cx[w] & ~l.keep[w]) ok = false;
        for (const tr_op& o : a.src[w]) if (mask_of(o) & ~l.keep[w]) ok = false;
      }
      if (ok) {
        for (int w = 0; w < 21; ++w) {
          a.keep[w] &= l.keep[w];
          a.cx[w] ^= l.cx[w];
          a.src[w].insert(a.src[w].end(), l.src[w].begin(), l.src[w].end());
        }
        a.flags |= l.flags;
        merged = true;
      }
    }
    if (!merged) ms.push_back(l);
  }
  // 3. the arrays.  EVERY step gets an image, so that the chain applies and prefetches unconditionally: steps that change no rate word share one identity
  // image (cx = 0, keep = all ones); one sentinel step behind the last is what the last step's look-ahead reads
  tr_step_prog sp;
  sp.chk = chk;
  sp.src_off.push_back(0);
  uint32_t ident = ~0u;
  auto add_image = [&](const L* l) {
    for (int w = 0; w < 21; ++w) {
      sp.cx.push_back(l ? l->cx[w] : 0);
      sp.keep.push_back(l ? l->keep[w] : ~0ULL);
      if (l) sp.src.insert(sp.src.end(), l->src[w].begin(), l->src[w].end());
      sp.src_off.push_back((uint32_t)sp.src.size());
    }
    return sp.n_img++;
  };
  for (const L& l : ms) {
    tr_step st{l.flags, 0, (uint32_t)sp.emit.size(), (uint32_t)l.emit.size()};
    if (!l.emit.empty()) st.flags |= TS_EMIT;
    sp.emit.insert(sp.emit.end(), l.emit.begin(), l.emit.end());
    bool img = false, keep = false;
    for (int w = 0; w < 21; ++w) { img = img || l.cx[w] != 0 || !l.src[w].empty(); keep = keep || l.keep[w] != ~0ULL; }
    if (img || keep) {
      st.flags |= TS_IMG | (keep ? TS_KEEP : 0);
      st.img = add_image(&l);
    } else {
      if (ident == ~0u) ident = add_image(nullptr);
      st.img = ident;
    }
    sp.steps.push_back(st);
  }
  if (!sp.steps.empty()) sp.steps.push_back(tr_step{0, sp.steps.back().img, 0, 0});
  return sp;
}

// the 64-bit value a source operation contributes for proof j
ZKP_HD uint64_t tr_src_word(const tr_fields& op, const tr_bufs& bufs, uint64_t j) {
  const uint64_t addr = j * op.stride + op.off;
  const uint32_t sh = (uint32_t)(addr & 7);
  const uint64_t* p = reinterpret_cast<const uint64_t*>(tr_src_ptr(bufs, op.src_buf - 1u) + (addr - sh));
  uint64_t x = p[0] >> (8 * sh);
  if (sh + op.nb > 8) x |= p[1] << (64 - 8 * sh);
  return (x & tr_bytemask(op.nb)) << (8 * op.lb);
}
// reference semantics of the step form for one proof (host tests; the kernels split it into assemble + chain)
inline void tr_steps_run_one(const tr_step_prog& sp, uint64_t j, const tr_bufs& bufs, uint64_t S[25], uint64_t saved[25], uint32_t* failed) {
  for (const tr_op& o : sp.chk) {
    const tr_fields f = tr_unpack(o.ctl, o.stride, o.off);
    const uint64_t* p = reinterpret_cast<const uint64_t*>(tr_src_ptr(bufs, f.src_buf - 1u) + j * f.stride + f.off);
    if ((p[0] | p[1] | p[2] | p[3]) == 0) *failed = 1;
  }
  const uint64_t ident[TR_TABLE_WORDS] = {~0ULL, ~0ULL, ~0ULL, ~0ULL, ~0ULL, ~0ULL, ~0ULL, ~0ULL, ~0ULL, ~0ULL, ~0ULL, ~0ULL, ~0ULL, ~0ULL, ~0ULL, ~0ULL, ~0ULL, ~0ULL, ~0ULL, ~0ULL, ~0ULL};
  for (const tr_step& st : sp.steps) {
    if (st.flags & TS_RESTORE) for (int i = 0; i < 25; ++i) S[i] = saved[i];
    for (uint32_t q = 0; q < st.emit_n; ++q) {
      const tr_op& o = sp.emit[st.emit_lo + q];
      tr_exec_op(tr_unpack(o.ctl, o.stride, o.off), nullptr, j, bufs, S, 1, saved, 1, failed, true);     // (a pure dst operation)
    }
    if (st.flags & TS_IMG) {
      for (int w = 0; w < 21; ++w) {
        const size_t k = (size_t)st.img * 21 + w;
        uint64_t x = sp.cx[k];
        for (uint32_t q = sp.src_off[k]; q < sp.src_off[k + 1]; ++q) x ^= tr_src_word(tr_unpack(sp.src[q].ctl, sp.src[q].stride, sp.src[q].off), bufs, j);
        S[w] = (S[w] & sp.keep[k]) ^ x;
      }
    }
    if (st.flags & TS_PERMUTE) tr_apply_block(S, 1, ident, true);
    if (st.flags & TS_SAVE) for (int i = 0; i < 25; ++i) saved[i] = S[i];
  }
}

}  // namespace zkp

// ---- host side: the compiler (plain C++, also parsed by the device pass of hipcc, never emitted there) ----
#include <string>
#include <vector>
namespace zkp {

struct tr_ref { uint8_t buf; uint32_t stride; uint64_t off; };     // buf = index into tr_bufs (src or dst)

// Compile-time mirror of Strobe128 / merlin::Transcript / TranscriptRng / TranscriptProtocol (host/merlin.cpp).
class TrCompiler {
 public:
  TrCompiler(uint8_t pos, uint8_t pos_begin, uint8_t cur_flags) : pos_(pos), pos_begin_(pos_begin), cur_flags_(cur_flags) { reset_block(); }

  // ---- merlin::Transcript ----
  void append_message(const char* label, const void* msg, size_t len) {
    frame(label, len);
    begin_op(kA, false);
    absorb_const(static_cast<const uint8_t*>(msg), len);
  }
  void append_message_var(const char* label, tr_ref src, size_t len) {
    frame(label, len);
    begin_op(kA, false);
    for (size_t i = 0; i < len; ++i) data_byte(0xff, 0, &src, i, nullptr, 0);
  }
  void challenge_bytes(const char* label, tr_ref dst, size_t len) {
    frame(label, len);
    begin_op(kI | kA | kC, false);
    for (size_t i = 0; i < len; ++i) data_byte(0, 0, nullptr, 0, &dst, i);
  }
  // ---- merlin::TranscriptRng (build_rng = save ... restore around the rng's operations) ----
  void save() { flush(false); marker(TR_SAVE); s_pos_ = pos_; s_pos_begin_ = pos_begin_; s_cur_flags_ = cur_flags_; }
  void restore() { flush(false); marker(TR_RESTORE); pos_ = s_pos_; pos_begin_ = s_pos_begin_; cur_flags_ = s_cur_flags_; }
  void rng_rekey_with_witness_var(const char* label, tr_ref src, size_t len) {
    frame(label, len);
    begin_op(kA | kC, false);
    for (size_t i = 0; i < len; ++i) data_byte(0, 0, &src, i, nullptr, 0);
  }
  void rng_finalize_var(tr_ref entropy) {
    begin_op(kM | kA, false);
    absorb_const(reinterpret_cast<const uint8_t*>("rng"), 3);
    begin_op(kA | kC, false);
    for (size_t i = 0; i < 32; ++i) data_byte(0, 0, &entropy, i, nullptr, 0);
  }
  void rng_fill_bytes(tr_ref dst, size_t len) {
    uint8_t l[4] = {(uint8_t)len, (uint8_t)(len >> 8), (uint8_t)(len >> 16), (uint8_t)(len >> 24)};
    begin_op(kM | kA, false);
    absorb_const(l, 4);
    begin_op(kI | kA | kC, false);
    for (size_t i = 0; i < len; ++i) data_byte(0, 0, nullptr, 0, &dst, i);
  }
  // ---- TranscriptProtocol (src/toolbox/mod.rs:165-228) ----
  void domain_sep(const char* label) {
    append_message("dom-sep", "schnorrzkp/1.0/ristretto255", 27);
    append_message("dom-sep", label, strlen(label));
  }
  void append_scalar_var(const char* label) { append_message("scvar", label, strlen(label)); }
  void append_point_var(const char* label, const uint8_t enc[32]) {
    append_message("ptvar", label, strlen(label));
    append_message("val", enc, 32);
  }
  void append_point_var_var(const char* label, tr_ref enc, bool validate) {
    if (validate) check_nonzero(enc);
    append_message("ptvar", label, strlen(label));
    append_message_var("val", enc, 32);
  }
  void append_blinding_commitment_var(const char* label, tr_ref enc, bool validate) {
    if (validate) check_nonzero(enc);
    append_message("blindcom", label, strlen(label));
    append_message_var("val", enc, 32);
  }
  void get_challenge_wide(const char* label, tr_ref dst) { challenge_bytes(label, dst, 64); }

  void check_nonzero(tr_ref src) {
    flush(false);
    tr_op_wide op{};
    op.keep = ~0ULL;
    op.flags = TR_CHECK_NONZERO;
    op.src_buf = (uint8_t)(src.buf + 1);
    op.src_stride = src.stride;
    op.src_off = src.off;
    ops_.push_back(op);
  }
  // ends the program; the transcript's trailing bytes (pos, pos_begin, cur_flags) after it
  std::vector<tr_op> finish(uint8_t tail[3]) {
    flush(false);
    tail[0] = pos_; tail[1] = pos_begin_; tail[2] = cur_flags_;
    std::vector<tr_op> packed;
    for (const auto& o : ops_) packed.push_back(tr_pack(o));
    return packed;
  }
  const std::vector<uint64_t>& tables() const { return tables_; }    // TR_TABLE_WORDS words per APPLY operation
  size_t permutations() const { return n_perm_; }

 private:
  static constexpr unsigned kRate = 166;
  enum : uint8_t { kI = 1, kA = 2, kC = 4, kT = 8, kM = 16, kK = 32 };
  struct ByteEff { uint8_t keep, cx; bool has_src, has_dst; tr_ref src, dst; };

  void frame(const char* label, size_t len) {
    uint8_t l[4] = {(uint8_t)len, (uint8_t)(len >> 8), (uint8_t)(len >> 16), (uint8_t)(len >> 24)};
    begin_op(kM | kA, false);
    absorb_const(reinterpret_cast<const uint8_t*>(label), strlen(label));
    absorb_const(l, 4);                     // meta_ad(.., more = true): continuation, no new header
  }
  void reset_block() {
    for (auto& e : eff_) e = ByteEff{0xff, 0, false, false, {}, {}};
    dirty_ = false;
  }
  void data_byte(uint8_t keep, uint8_t cx, const tr_ref* src, size_t si, const tr_ref* dst, size_t di) {
    ByteEff& e = eff_[pos_];
    e.keep &= keep;
    e.cx = (uint8_t)((e.cx & keep) ^ cx);
    if (src) { e.has_src = true; e.src = *src; e.src.off += si; }
    if (dst) { e.has_dst = true; e.dst = *dst; e.dst.off += di; }
    dirty_ = true;
    if (++pos_ == kRate) run_f();
  }
  void absorb_const(const uint8_t* d, size_t n) { for (size_t i = 0; i < n; ++i) data_byte(0xff, d[i], nullptr, 0, nullptr, 0); }
  void run_f() {
    eff_[pos_].cx ^= pos_begin_;
    eff_[pos_ + 1].cx ^= 0x04;
    eff_[kRate + 1].cx ^= 0x80;
    dirty_ = true;
    flush(true);
    pos_ = 0;
    pos_begin_ = 0;
  }
  void begin_op(uint8_t flags, bool more) {
    if (more) return;
    const uint8_t old_begin = pos_begin_;
    pos_begin_ = (uint8_t)(pos_ + 1);
    cur_flags_ = flags;
    const uint8_t hdr[2] = {old_begin, flags};
    absorb_const(hdr, 2);
    if ((flags & (kC | kK)) && pos_ != 0) run_f();
  }
  void marker(uint8_t flag) {
    tr_op_wide op{};
    op.keep = ~0ULL;
    op.flags = flag;
    ops_.push_back(op);
  }
  // emit the pending byte effects: word operations for the bytes that move, one APPLY with the block's constant
  // table for everything else; permute = the block ends here
  void flush(bool permute) {
    if (!dirty_ && !permute) return;
    uint64_t keep[21], cx[21];
    for (unsigned w = 0; w < 21; ++w) {
      const ByteEff* e = &eff_[8 * w];
      keep[w] = 0; cx[w] = 0;
      for (unsigned b = 0; b < 8; ++b) {
        // bytes overwritten from a per-proof source are cleared by their own operation, not by the table
        const uint8_t k = e[b].has_src ? 0xff : e[b].keep;
        keep[w] |= (uint64_t)k << (8 * b);
        cx[w] |= (uint64_t)e[b].cx << (8 * b);
      }
      // bytes leaving the state (squeeze), grouped into runs with contiguous destinations
      for (unsigned b = 0; b < 8;) {
        if (!e[b].has_dst) { ++b; continue; }
        unsigned n = 1;
        while (b + n < 8 && e[b + n].has_dst && e[b + n].dst.buf == e[b].dst.buf && e[b + n].dst.off == e[b].dst.off + n) ++n;
        tr_op_wide op{};
        op.keep = ~0ULL;
        op.w = (uint8_t)w;
        op.dst_buf = (uint8_t)(e[b].dst.buf + 1);
        op.dst_stride = e[b].dst.stride;
        op.dst_off = e[b].dst.off;
        op.dnb = (uint8_t)n;
        op.dlb = (uint8_t)b;
        ops_.push_back(op);
        b += n;
      }
      for (unsigned b = 0; b < 8;) {
        if (!e[b].has_src) { ++b; continue; }
        unsigned n = 1;
        // a run must stay inside one 32-byte source item so that the two aligned loads of the interpreter do too,
        // and must be all-absorb or all-overwrite
        while (b + n < 8 && e[b + n].has_src && e[b + n].src.buf == e[b].src.buf && e[b + n].src.off == e[b].src.off + n &&
               e[b + n].keep == e[b].keep && ((e[b].src.off + n) & 31) != 0) ++n;
        tr_op_wide op{};
        const uint64_t mask = (n >= 8 ? ~0ULL : ((1ULL << (8 * n)) - 1)) << (8 * b);
        op.keep = e[b].keep ? ~0ULL : ~mask;
        if (!e[b].keep) op.flags |= TR_OVERWRITE;
        op.w = (uint8_t)w;
        op.src_buf = (uint8_t)(e[b].src.buf + 1);
        op.src_stride = e[b].src.stride;
        op.src_off = e[b].src.off;
        op.nb = (uint8_t)n;
        op.lb = (uint8_t)b;
        ops_.push_back(op);
        b += n;
      }
    }
    tr_op_wide ap{};
    ap.keep = ~0ULL;
    ap.flags = (uint8_t)(TR_APPLY | (permute ? TR_PERMUTE : 0));
    ap.src_off = tables_.size() / TR_TABLE_WORDS;
    tables_.insert(tables_.end(), keep, keep + 21);
    tables_.insert(tables_.end(), cx, cx + 21);
    ops_.push_back(ap);
    if (permute) ++n_perm_;
    reset_block();
  }

  std::vector<tr_op_wide> ops_;
  std::vector<uint64_t> tables_;
  ByteEff eff_[168];
  bool dirty_ = false;
  uint8_t pos_, pos_begin_, cur_flags_;
  uint8_t s_pos_ = 0, s_pos_begin_ = 0, s_cur_flags_ = 0;
  size_t n_perm_ = 0;
};

}  // namespace zkp
