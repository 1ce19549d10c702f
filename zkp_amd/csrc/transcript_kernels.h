// The Merlin transcript interpreters on the GPU (semantics: tr_exec_op / tr_run_one in merlin_prog.h) and the launch that shares
// them with the comb-table construction.  Included by zkp_kernels.hip before the term path (which issues the shared launch)
// and by fused_flows.h.
#pragma once
#include "merlin_prog.h"
#include "comb_tables.h"

namespace zkp {

// ---- the transcript interpreter on the GPU ---------------------------------------------------------------------------
// A lone wavefront issues one VALU instruction every 4 cycles, and a batch of a few thousand proofs is only a few dozen
// wavefronts: the kernel is latency-bound, so each proof is spread over a PAIR of lanes.  Lane h of the pair holds the
// h-th 32-bit half of every 64-bit STROBE word (LDS columns of uint32: dynamic word index without scratch).  XOR / AND /
// NOT are half-local; a 64-bit rotation takes the partner's half with one DPP quad_perm move and one v_alignbit_b32:
// ~160 full-rate VALU operations per Keccak round per lane instead of 264.  Semantics = tr_run_one (merlin_prog.h).
__device__ __forceinline__ uint32_t pair_swap(uint32_t v) {
  return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1 /* quad_perm [1,0,3,2] */, 0xf, 0xf, true);
}
// half of rotl64 by R that this lane keeps (mine / partner = this lane's and the other lane's half of the word)
__device__ __forceinline__ uint32_t rotl_half(uint32_t mine, uint32_t partner, int R) {
  const int n = R & 31;
  if (R & 32) return n ? __builtin_amdgcn_alignbit(partner, mine, 32 - n) : partner;
  return n ? __builtin_amdgcn_alignbit(mine, partner, 32 - n) : mine;
}
__device__ __forceinline__ uint32_t half_of(uint64_t v, uint32_t h) { return h ? (uint32_t)(v >> 32) : (uint32_t)v; }

template <bool UNROLL = false>
__device__ __forceinline__ void keccak_f1600_split(uint32_t a[25], uint32_t h) {
  constexpr uint64_t RC[24] = {
      0x0000000000000001ULL, 0x0000000000008082ULL, 0x800000000000808aULL, 0x8000000080008000ULL, 0x000000000000808bULL,
      0x0000000080000001ULL, 0x8000000080008081ULL, 0x8000000000008009ULL, 0x000000000000008aULL, 0x0000000000000088ULL,
      0x0000000080008009ULL, 0x000000008000000aULL, 0x000000008000808bULL, 0x800000000000008bULL, 0x8000000000008089ULL,
      0x8000000000008003ULL, 0x8000000000008002ULL, 0x8000000000000080ULL, 0x000000000000800aULL, 0x800000008000000aULL,
      0x8000000080008081ULL, 0x8000000000008080ULL, 0x0000000080000001ULL, 0x8000000080008008ULL};
  constexpr int RHO[25] = {0, 1, 62, 28, 27, 36, 44, 6, 55, 20, 3, 10, 43, 25, 39, 41, 45, 15, 21, 8, 18, 2, 61, 56, 14};   // [x + 5 y]
  auto one_round = [&](const uint64_t rc) {
    uint32_t c[5], b[25];
#pragma unroll
    for (int x = 0; x < 5; ++x) c[x] = tr_xor5_32(a[x], a[x + 5], a[x + 10], a[x + 15], a[x + 20]);
#pragma unroll
    for (int x = 0; x < 5; ++x) {                       // theta: D[x] = C[x-1] ^ rotl(C[x+1], 1)
      const uint32_t cn = c[(x + 1) % 5];
      const uint32_t d = c[(x + 4) % 5] ^ rotl_half(cn, pair_swap(cn), 1);
#pragma unroll
      for (int y = 0; y < 5; ++y) a[x + 5 * y] ^= d;
    }
#pragma unroll
    for (int y = 0; y < 5; ++y)                         // rho + pi: B[y][2x+3y] = rotl(A[x][y], r[x][y])
#pragma unroll
      for (int x = 0; x < 5; ++x) {
        const uint32_t v = a[x + 5 * y];
        b[y + 5 * ((2 * x + 3 * y) % 5)] = RHO[x + 5 * y] ? rotl_half(v, pair_swap(v), RHO[x + 5 * y]) : v;
      }
#pragma unroll
    for (int y = 0; y < 5; ++y)                         // chi
#pragma unroll
      for (int x = 0; x < 5; ++x) a[x + 5 * y] = b[x + 5 * y] ^ (~b[(x + 1) % 5 + 5 * y] & b[(x + 2) % 5 + 5 * y]);
    a[0] ^= half_of(rc, h);                             // iota
  };
  if constexpr (UNROLL) {                               // the chain kernel: round constants as literals, no loop (5.5 instead of 6.2 us per permutation on a lone
#pragma unroll                                          // wavefront, profiles/r06_keccak_microbench.txt)
    for (int round = 0; round < 24; ++round) one_round(RC[round]);
  } else {
    static const uint64_t RCT[24] = {
        0x0000000000000001ULL, 0x0000000000008082ULL, 0x800000000000808aULL, 0x8000000080008000ULL, 0x000000000000808bULL,
        0x0000000080000001ULL, 0x8000000080008081ULL, 0x8000000000008009ULL, 0x000000000000008aULL, 0x0000000000000088ULL,
        0x0000000080008009ULL, 0x000000008000000aULL, 0x000000008000808bULL, 0x800000000000008bULL, 0x8000000000008089ULL,
        0x8000000000008003ULL, 0x8000000000008002ULL, 0x8000000000000080ULL, 0x000000000000800aULL, 0x800000008000000aULL,
        0x8000000080008081ULL, 0x8000000000008080ULL, 0x0000000080000001ULL, 0x8000000080008008ULL};
#pragma unroll 1
    for (int round = 0; round < 24; ++round) one_round(RCT[round]);
  }
}

constexpr int TR_BLOCK = 64;       // lanes per workgroup (one wavefront) = 32 proofs
// The operation list is fetched 64 operations at a time: lane i loads operation base + i (one coalesced 1 KiB load), and
// the wavefront then walks them with v_readlane -- no scalar-memory latency inside the loop (an s_load per operation
// would be waited for at every LDS access, since both count on lgkmcnt).
__device__ __forceinline__ void transcript_pair_block(uint32_t bid, uint32_t* S /*LDS [25 * TR_BLOCK]*/, const tr_op* __restrict__ prog, uint32_t n_ops,
                                                      const uint64_t* __restrict__ tables, uint32_t N, const tr_bufs& bufs, uint8_t* __restrict__ ts,
                                                      uint32_t* __restrict__ saved /*[25][2N]*/, uint32_t* __restrict__ failed, uint32_t tail) {
  const uint32_t lane = threadIdx.x & (TR_BLOCK - 1), h = lane & 1;     // (one wavefront per transcript block; k_tables_transcript_pc runs two in a workgroup)
  const uint32_t j_raw = bid * (TR_BLOCK / 2) + (lane >> 1);
  const bool live = j_raw < N;                          // lanes past the end shadow the last proof (they must stay in the
  const uint32_t j = live ? j_raw : N - 1;              // wavefront: they carry operations for v_readlane) and store nothing
  uint32_t* col = S + lane;
  uint32_t* blob = reinterpret_cast<uint32_t*>(ts + 208 * (size_t)j);
#pragma unroll
  for (int i = 0; i < 25; ++i) col[TR_BLOCK * i] = blob[2 * i + h];
  uint32_t* sv = saved + 2 * (size_t)j + h;
  const size_t sv_stride = 2 * (size_t)N;
  uint32_t bad = 0;
  for (uint32_t base = 0; base < n_ops; base += TR_BLOCK) {
    const uint32_t cnt = n_ops - base < TR_BLOCK ? n_ops - base : TR_BLOCK;
    uint4 mine = make_uint4(0, 0, 0, 0);
    if (lane < cnt) mine = reinterpret_cast<const uint4*>(prog)[base + lane];
    for (uint32_t i = 0; i < cnt; ++i) {
      const uint32_t o_ctl = (uint32_t)__builtin_amdgcn_readlane((int)mine.x, (int)i);
      const uint32_t o_stride = (uint32_t)__builtin_amdgcn_readlane((int)mine.y, (int)i);
      const uint64_t o_off = (uint32_t)__builtin_amdgcn_readlane((int)mine.z, (int)i) | (uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)mine.w, (int)i) << 32;
      const tr_fields op = tr_unpack(o_ctl, o_stride, o_off);
      if (op.flags & TR_RESTORE) {
#pragma unroll
        for (int k = 0; k < 25; ++k) col[TR_BLOCK * k] = sv[k * sv_stride];
      }
      if (op.flags & TR_CHECK_NONZERO) {
        const uint4* p = reinterpret_cast<const uint4*>(tr_src_ptr(bufs, op.src_buf - 1u) + (size_t)j * op.stride + op.off);
        const uint4 lo = p[0], hi = p[1];
        if ((lo.x | lo.y | lo.z | lo.w | hi.x | hi.y | hi.z | hi.w) == 0) bad = 1;
      }
      if (op.dst_buf) {                                 // PRF output: the bytes of word w that live in this half
        uint8_t* d = tr_dst_ptr(bufs, op.dst_buf - 1u) + (size_t)j * op.stride + op.off;
        // this half holds bytes [4h, 4h + 4) of the word; its share of [dlb, dlb + dnb) is one run: a whole half goes
        // out as one (possibly unaligned) 32-bit store
        const uint32_t lo = op.dlb > 4 * h ? op.dlb : 4 * h;
        const uint32_t hi = op.dlb + op.dnb < 4 * h + 4 ? op.dlb + op.dnb : 4 * h + 4;
        if (hi > lo && live) {
          const uint32_t v = col[TR_BLOCK * op.w] >> (8 * (lo - 4 * h));
          uint8_t* dp = d + (lo - op.dlb);
          if (hi - lo == 4) {
            __builtin_memcpy(dp, &v, 4);
          } else {
            for (uint32_t k = 0; k < hi - lo; ++k) dp[k] = (uint8_t)(v >> (8 * k));
          }
        }
      }
      if (op.src_buf && !(op.flags & TR_CHECK_NONZERO)) {
        const uint64_t addr = (uint64_t)j * op.stride + op.off;
        const uint32_t sh = (uint32_t)(addr & 7);
        const uint64_t* p = reinterpret_cast<const uint64_t*>(tr_src_ptr(bufs, op.src_buf - 1u) + (addr - sh));
        uint64_t x = p[0] >> (8 * sh);
        if (sh + op.nb > 8) x |= p[1] << (64 - 8 * sh);
        x = (x & tr_bytemask(op.nb)) << (8 * op.lb);
        col[TR_BLOCK * op.w] = (col[TR_BLOCK * op.w] & half_of(op.keep, h)) ^ half_of(x, h);
      }
      if (op.flags & TR_APPLY) {
        const uint64_t* tbl = tables + (size_t)TR_TABLE_WORDS * op.off;
        uint32_t a[25];
#pragma unroll
        for (int k = 0; k < 25; ++k) a[k] = col[TR_BLOCK * k];
#pragma unroll
        for (int k = 0; k < 21; ++k) a[k] = (a[k] & half_of(tbl[k], h)) ^ half_of(tbl[21 + k], h);
        if (op.flags & TR_PERMUTE) keccak_f1600_split(a, h);
#pragma unroll
        for (int k = 0; k < 25; ++k) col[TR_BLOCK * k] = a[k];
      }
      if ((op.flags & TR_SAVE) && live) {
#pragma unroll
        for (int k = 0; k < 25; ++k) sv[k * sv_stride] = col[TR_BLOCK * k];
      }
    }
  }
  if (!live) return;
#pragma unroll
  for (int i = 0; i < 25; ++i) blob[2 * i + h] = col[TR_BLOCK * i];
  if (h == 0) { blob[50] = tail & 0xffffffu; blob[51] = 0; }
  // bit 31 of `tail`: this program owns the rejection flags (it writes 0 too, so that nobody has to clear them first)
  if (h == 0 && (bad || (tail >> 31))) failed[j] = bad;
  if ((tail >> 31) && j_raw == 0 && h == 0) failed[N] = 0;      // the spare word behind the flags: the flow's shared status bits
}

__global__ void __launch_bounds__(TR_BLOCK)
k_transcript_run(const tr_op* __restrict__ prog, uint32_t n_ops, const uint64_t* __restrict__ tables, uint32_t N, const tr_bufs bufs,
                 uint8_t* __restrict__ ts, uint32_t* __restrict__ saved /*[25][2N]*/, uint32_t* __restrict__ failed, uint32_t tail) {
  __shared__ uint32_t S[25 * TR_BLOCK];
  transcript_pair_block(blockIdx.x, S, prog, n_ops, tables, N, bufs, ts, saved, failed, tail);
}

// ---- round 6: assemble + chain (the step form of a program, merlin_prog.h) ------------------------------------------------
struct tr_steps_dev {
  const tr_step* steps = nullptr;          // [n_steps]: the last one is a sentinel (never executed: the last real step's look-ahead)
  const tr_op* emit = nullptr;
  const tr_op* src = nullptr;
  const uint32_t* src_off = nullptr;       // [n_img * 21 + 1]
  const uint64_t* cx = nullptr;            // [n_img * 21]
  const uint32_t* keep32 = nullptr;        // [n_img * 21][2]: the halves of the 64-bit keep words
  const tr_op* chk = nullptr;
  uint32_t n_steps = 0, n_img = 0, n_chk = 0, tail = 0;
};

// Lane j builds proof j's 64-bit image words (every load independent of every other), seven words per lane;
// the last row checks the encodings the verifiers must reject as identity (mod.rs:191, :215) and owns the rejection flags.
// img = [n_img * 21][N] uint64 (the chain reads it as [..][2 N] uint32: proof j's halves next to each other): coalesced on both sides.  (A proof-major
// layout -- 168 contiguous bytes per proof and image, one pointer + immediate offsets in the chain -- was measured: assemble 15 -> 31 us, chain no faster.)
constexpr uint32_t TA_WORDS = 7, TA_BLOCK = 64;     // image words per lane / lanes per workgroup of k_transcript_assemble
__global__ void __launch_bounds__(TA_BLOCK)
k_transcript_assemble(const tr_steps_dev p, uint32_t N, const tr_bufs bufs, uint64_t* __restrict__ img, uint32_t* __restrict__ failed) {
  // blockIdx.y = a third of an image (7 of its 21 words), or the check row: one launch of (N / 64) x (3 n_img + 1) wavefronts.  (One word per lane and
  // 256-lane workgroups, the first version, was 188 k workgroups of ~20 instructions per wide call: bound by the dispatcher, 1.6 ms per 40,960 proofs.)
  const uint32_t j = blockIdx.x * TA_BLOCK + threadIdx.x, y = blockIdx.y;
  if (j >= N) return;
  if (y < p.n_img * 3u) {
    const uint32_t k0 = (y / 3u) * 21u + (y % 3u) * TA_WORDS;
#pragma unroll 1
    for (uint32_t k = k0; k < k0 + TA_WORDS; ++k) {
      uint64_t x = p.cx[k];
      for (uint32_t q = p.src_off[k]; q < p.src_off[k + 1]; ++q) {
        const tr_op o = p.src[q];
        x ^= tr_src_word(tr_unpack(o.ctl, o.stride, o.off), bufs, j);
      }
      img[(size_t)k * N + j] = x;
    }
  } else {
    uint32_t bad = 0;
    for (uint32_t q = 0; q < p.n_chk; ++q) {
      const tr_op o = p.chk[q];
      const tr_fields f = tr_unpack(o.ctl, o.stride, o.off);
      const uint4* s = reinterpret_cast<const uint4*>(tr_src_ptr(bufs, f.src_buf - 1u) + (size_t)j * f.stride + f.off);
      const uint4 lo = s[0], hi = s[1];
      if ((lo.x | lo.y | lo.z | lo.w | hi.x | hi.y | hi.z | hi.w) == 0) bad = 1;
    }
    // bit 31 of the tail: this program owns the rejection flags (it writes 0 too, so that nobody has to clear them first)
    if (bad || (p.tail >> 31)) failed[j] = bad;
    if ((p.tail >> 31) && j == 0) failed[N] = 0;        // the spare word behind the flags: the flow's shared status bits
  }
}

// The chain: a lane pair per proof, the state in 25 registers per lane, per step 21 + 21 loads that were issued one permutation earlier.
// PRF output (rare: 21 blinding strings, one challenge) goes through an LDS column like in the interpreter above.
__device__ __forceinline__ void transcript_chain_block(uint32_t bid, uint32_t* S /*LDS [25 * TR_BLOCK]*/, const tr_steps_dev& p, const uint32_t* __restrict__ img32, uint32_t N,
                                                       const tr_bufs& bufs, uint8_t* __restrict__ ts, uint32_t* __restrict__ saved /*[25][2N]*/) {
  const uint32_t lane = threadIdx.x & (TR_BLOCK - 1), h = lane & 1;
  const uint32_t j_raw = bid * (TR_BLOCK / 2) + (lane >> 1);
  const bool live = j_raw < N;                          // lanes past the end shadow the last proof and store nothing
  const uint32_t j = live ? j_raw : N - 1;
  uint32_t* col = S + lane;
  uint32_t* blob = reinterpret_cast<uint32_t*>(ts + 208 * (size_t)j);
  uint32_t a[25], ni[21], nk[21];
#pragma unroll
  for (int i = 0; i < 25; ++i) a[i] = blob[2 * i + h];
  const size_t wstride = 2 * (size_t)N;
  const uint32_t* mycol = img32 + 2 * (size_t)j + h;
  uint32_t* sv = saved + 2 * (size_t)j + h;
  // Every step has an image row (steps that change nothing share an identity row) and the list ends in a sentinel, so the loop below has ONE place where
  // the look-ahead image is loaded and one where it is consumed -- nothing merges in front of the permutation, and the loads' s_waitcnt lands behind it.
  // (The first version loaded "if the next step has an image" and fell back to loading on demand: the compiler resolved that merge with register copies
  // in front of the permutation, i.e. a full vmcnt(0) wait per step -- 8.2 us per step instead of 7.3.)
  const uint32_t n_real = p.n_steps - 1;                // the sentinel is not executed
  {
    const uint32_t im0 = p.steps[0].img;
#pragma unroll
    for (int k = 0; k < 21; ++k) { ni[k] = mycol[((size_t)im0 * 21 + k) * wstride]; nk[k] = p.keep32[(size_t)im0 * 42 + 2 * k + h]; }
  }
  constexpr uint32_t CH = TR_BLOCK - 1;                  // steps per descriptor fetch: lane CH holds the look-ahead of the chunk's last step
  for (uint32_t base = 0; base < n_real; base += CH) {
    const uint32_t cnt = n_real - base < CH ? n_real - base : CH;
    uint4 mine = make_uint4(0, 0, 0, 0);
    if (lane <= cnt) mine = reinterpret_cast<const uint4*>(p.steps)[base + lane];
    for (uint32_t i = 0; i < cnt; ++i) {
      const uint32_t fl = (uint32_t)__builtin_amdgcn_readlane((int)mine.x, (int)i);
      const uint32_t nim = (uint32_t)__builtin_amdgcn_readlane((int)mine.y, (int)(i + 1));
      if (fl & TS_RESTORE) {
#pragma unroll
        for (int k = 0; k < 25; ++k) a[k] = sv[k * wstride];
      }
      if (fl & TS_EMIT) {
        const uint32_t elo = (uint32_t)__builtin_amdgcn_readlane((int)mine.z, (int)i), en = (uint32_t)__builtin_amdgcn_readlane((int)mine.w, (int)i);
#pragma unroll
        for (int k = 0; k < 21; ++k) col[TR_BLOCK * k] = a[k];
        uint4 eo = make_uint4(0, 0, 0, 0);
        if (lane < en) eo = reinterpret_cast<const uint4*>(p.emit)[elo + lane];
        for (uint32_t q = 0; q < en; ++q) {
          const uint32_t o_ctl = (uint32_t)__builtin_amdgcn_readlane((int)eo.x, (int)q), o_stride = (uint32_t)__builtin_amdgcn_readlane((int)eo.y, (int)q);
          const uint64_t o_off = (uint32_t)__builtin_amdgcn_readlane((int)eo.z, (int)q) | (uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)eo.w, (int)q) << 32;
          const tr_fields op = tr_unpack(o_ctl, o_stride, o_off);
          uint8_t* d = tr_dst_ptr(bufs, op.dst_buf - 1u) + (size_t)j * op.stride + op.off;
          // this half holds bytes [4h, 4h + 4) of the word; its share of [dlb, dlb + dnb) is one run
          const uint32_t lo = op.dlb > 4 * h ? op.dlb : 4 * h;
          const uint32_t hi = op.dlb + op.dnb < 4 * h + 4 ? op.dlb + op.dnb : 4 * h + 4;
          if (hi > lo && live) {
            const uint32_t v = col[TR_BLOCK * op.w] >> (8 * (lo - 4 * h));
            uint8_t* dp = d + (lo - op.dlb);
            if (hi - lo == 4) {
              __builtin_memcpy(dp, &v, 4);
            } else {
              for (uint32_t k = 0; k < hi - lo; ++k) dp[k] = (uint8_t)(v >> (8 * k));
            }
          }
        }
      }
      // state[w] = (state[w] & KEEP[w]) ^ image[w]: one v_bitop3_b32 per word (truth table (a & b) ^ c = 0x6A)
#pragma unroll
      for (int k = 0; k < 21; ++k) a[k] = (uint32_t)__builtin_amdgcn_bitop3_b32((int)a[k], (int)nk[k], (int)ni[k], 0x6A);
      // the next step's image and keep words travel under this step's permutation
#pragma unroll
      for (int k = 0; k < 21; ++k) { ni[k] = mycol[((size_t)nim * 21 + k) * wstride]; nk[k] = p.keep32[(size_t)nim * 42 + 2 * k + h]; }
      if (fl & TS_PERMUTE) keccak_f1600_split<true>(a, h);
      if ((fl & TS_SAVE) && live) {
#pragma unroll
        for (int k = 0; k < 25; ++k) sv[k * wstride] = a[k];
      }
    }
  }
  if (!live) return;
#pragma unroll
  for (int i = 0; i < 25; ++i) blob[2 * i + h] = a[i];
  if (h == 0) { blob[50] = p.tail & 0xffffffu; blob[51] = 0; }
}

// ALONE = the latency schedule's instantiation: the wavefront claims (almost) a whole SIMD's register file -- 192 VGPRs + 240 accumulation registers nobody
// uses -- so that no wavefront of the point phase that runs on the side stream at the same time (comb tables: 86 registers, decode: 186) can be placed on ITS SIMD.
// A chain wavefront that shares its SIMD issues every other slot: next to k_comb_tables the chain of program A took 549 us instead of 345 (rocprofv3
// timeline of a synchronous call, profiles/r06_ab_experiments.txt block c), which is the whole gain of running the two side by side.  128 wavefronts on a chip
// of 1024 SIMDs: the registers cost nothing.  Pipelined callers (throughput schedule) take ALONE = false: their other streams' kernels should fill those SIMDs.
template <bool ALONE>
__global__ void __launch_bounds__(TR_BLOCK)
k_transcript_chain(const tr_steps_dev p, const uint32_t* __restrict__ img32, uint32_t N, const tr_bufs bufs, uint8_t* __restrict__ ts, uint32_t* __restrict__ saved) {
  __shared__ uint32_t S[25 * TR_BLOCK];
  if constexpr (ALONE) asm volatile("" ::: "a239");
  transcript_chain_block(blockIdx.x, S, p, img32, N, bufs, ts, saved);
}

// Transcript program + comb-table construction in ONE launch (asynchronous _dev flows): both are a few dozen wavefronts of long
// dependent chains (Keccak-f rounds; 256 doublings per table) that do not depend on each other, and the _dev flows keep
// everything on one stream (a second stream per context costs the pipelined caller more than the overlap returns:
// ZKP_OPT_DEV_OVERLAP).  Rounds 3 - 4 ran one wavefront per workgroup -- a transcript block, or 64 tables built by one lane each
// (comb_table_lane: 368 sequential point operations, the 0.9 ms pole of the launch).  Round 5:
// the table builder is split into a producer and a consumer wavefront (comb_table_pc): workgroups of two wavefronts -- two transcript blocks of 32
// proofs, or 64 tables.  The launch's pole drops from the 368-operation table chain to the transcript program: a lone call of 4096 proofs 1.75 -> 1.88 M
// proofs/s, four such calls in flight 3.67 -> 4.03 M, wide pipelined calls unchanged (profiles/r05_ab_experiments.txt, block c).
template <int TEETH>
__global__ void __launch_bounds__(2 * TR_BLOCK, 2)
k_tables_transcript_pc(uint32_t tr_blocks, const tr_op* __restrict__ prog, uint32_t n_ops, const uint64_t* __restrict__ tables, uint32_t N, const tr_bufs bufs,
                       uint8_t* __restrict__ ts, uint32_t* __restrict__ saved, uint32_t* __restrict__ failed, uint32_t tail,
                       const uint32_t* __restrict__ n_slots, uint32_t max_tables, const uint32_t* __restrict__ slot_pt,
                       const dev_affine* __restrict__ pts, dev_ext* __restrict__ comb) {
  __shared__ uint32_t S[2 * 36 * 64];                      // two transcript states (2 x 25 x 64 words), or the producer's double-buffered hand-over
  static_assert(2 * 36 * 64 >= 2 * 25 * TR_BLOCK, "LDS of k_tables_transcript_pc");
  const uint32_t wave = threadIdx.x >> 6, trb = (tr_blocks + 1) / 2;
  if (blockIdx.x < trb) {
    const uint32_t bid = 2 * blockIdx.x + wave;
    if (bid < tr_blocks) transcript_pair_block(bid, S + wave * 25 * TR_BLOCK, prog, n_ops, tables, N, bufs, ts, saved, failed, tail);
  } else {
    comb_table_pc<TEETH>((blockIdx.x - trb) * 64u, n_slots, max_tables, slot_pt, pts, comb, S);
  }
}

// the shared launch with the chain of a step program in place of the interpreter (its assemble pass has run before)
template <int TEETH>
__global__ void __launch_bounds__(2 * TR_BLOCK, 2)
k_tables_chain_pc(uint32_t tr_blocks, const tr_steps_dev p, const uint32_t* __restrict__ img32, uint32_t N, const tr_bufs bufs, uint8_t* __restrict__ ts,
                  uint32_t* __restrict__ saved, const uint32_t* __restrict__ n_slots, uint32_t max_tables, const uint32_t* __restrict__ slot_pt,
                  const dev_affine* __restrict__ pts, dev_ext* __restrict__ comb) {
  __shared__ uint32_t S[2 * 36 * 64];
  const uint32_t wave = threadIdx.x >> 6, trb = (tr_blocks + 1) / 2;
  if (blockIdx.x < trb) {
    const uint32_t bid = 2 * blockIdx.x + wave;
    if (bid < tr_blocks) transcript_chain_block(bid, S + wave * 25 * TR_BLOCK, p, img32, N, bufs, ts, saved);
  } else {
    comb_table_pc<TEETH>((blockIdx.x - trb) * 64u, n_slots, max_tables, slot_pt, pts, comb, S);
  }
}

// The same interpreter with ONE lane per proof (64-bit words, tr_exec_op): 208 instead of 2 x 135 VALU operations per Keccak
// round and proof, and twice the latency -- the choice of the asynchronous _dev entry points, whose callers keep calls in
// flight (ZKP_OPT_TRANSCRIPT_LANES).  State in LDS columns of uint64 (dynamic word index without scratch); the clone slots
// have the pair kernel's layout.
__global__ void __launch_bounds__(TR_BLOCK)
k_transcript_run1(const tr_op* __restrict__ prog, uint32_t n_ops, const uint64_t* __restrict__ tables, uint32_t N, const tr_bufs bufs,
                  uint8_t* __restrict__ ts, uint64_t* __restrict__ saved /*[25][N]*/, uint32_t* __restrict__ failed, uint32_t tail) {
  __shared__ uint64_t S[25 * TR_BLOCK];
  const uint32_t lane = threadIdx.x;
  const uint32_t j_raw = blockIdx.x * TR_BLOCK + lane;
  const bool live = j_raw < N;                          // lanes past the end shadow the last proof and store nothing
  const uint32_t j = live ? j_raw : N - 1;
  uint64_t* col = S + lane;
  uint64_t* blob = reinterpret_cast<uint64_t*>(ts + 208 * (size_t)j);
#pragma unroll
  for (int i = 0; i < 25; ++i) col[TR_BLOCK * i] = blob[i];
  uint32_t bad = 0;
  for (uint32_t base = 0; base < n_ops; base += TR_BLOCK) {
    const uint32_t cnt = n_ops - base < TR_BLOCK ? n_ops - base : TR_BLOCK;
    uint4 mine = make_uint4(0, 0, 0, 0);
    if (lane < cnt) mine = reinterpret_cast<const uint4*>(prog)[base + lane];
    for (uint32_t i = 0; i < cnt; ++i) {
      const uint32_t o_ctl = (uint32_t)__builtin_amdgcn_readlane((int)mine.x, (int)i);
      const uint32_t o_stride = (uint32_t)__builtin_amdgcn_readlane((int)mine.y, (int)i);
      const uint64_t o_off = (uint32_t)__builtin_amdgcn_readlane((int)mine.z, (int)i) | (uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)mine.w, (int)i) << 32;
      tr_exec_op(tr_unpack(o_ctl, o_stride, o_off), tables, j, bufs, col, TR_BLOCK, saved + j, N, &bad, live);
    }
  }
  if (!live) return;
#pragma unroll
  for (int i = 0; i < 25; ++i) blob[i] = col[TR_BLOCK * i];
  blob[25] = tail & 0xffffffu;
  if (bad || (tail >> 31)) failed[j] = bad;
  if ((tail >> 31) && j_raw == 0) failed[N] = 0;
}

}  // namespace zkp
