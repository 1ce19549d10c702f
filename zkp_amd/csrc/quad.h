// 4-lane cooperative point arithmetic for the LATENCY-bound phases (Pippenger bucket merge, reduction tree,
// Horner): a point is spread over a quad of lanes, lane q (= lane id & 3) holding one coordinate
//        q = 0: X     q = 1: Y     q = 2: Z     q = 3: T
// so that the four independent field multiplications of each half of an addition / doubling run in parallel and a
// dependent chain of point operations advances ~2.3x faster than with one lane per point (a lone wave needs 1.74 us
// per doubling and 1.9 us per addition, tools/microbench/ge_chain.hip).  Operands move between the lanes of a quad with
// DPP quad_perm moves (full-rate VALU, no LDS).  All lanes of a quad execute the same instruction stream; lane-specific
// results are picked with masks.
#pragma once
#include "dev_layout.h"

namespace zkp {

#define ZKP_QP(a, b, c, d) ((a) | ((b) << 2) | ((c) << 4) | ((d) << 6))

template <int CTRL>
__device__ __forceinline__ void fe_dpp(fe& r, const fe& a) {
#pragma unroll
  for (int i = 0; i < 9; ++i)
    r.v[i] = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)a.v[i], CTRL, 0xf, 0xf, true);
}
// r = (cond ? a : r), cond is per lane
__device__ __forceinline__ void fe_pick(fe& r, const fe& a, bool cond) {
#pragma unroll
  for (int i = 0; i < 9; ++i) r.v[i] = cond ? a.v[i] : r.v[i];
}

struct qpt { fe c; };          // extended point, one coordinate per lane (X, Y, Z, T)
struct qcached { fe c; };      // cached operand: lane 0: Y-X, 1: Y+X, 2: 2Z, 3: 2dT  (all tight)

__device__ __forceinline__ void q_identity(qpt& p, int q) {
  fe_0(p.c);
  p.c.v[0] = (q == 1 || q == 2) ? 1u : 0u;
}
__device__ __forceinline__ void q_load_ext(qpt& p, const dev_ext* src, int q) {
  const uint32_t* w = reinterpret_cast<const uint32_t*>(src) + 9 * q;
#pragma unroll
  for (int i = 0; i < 9; ++i) p.c.v[i] = w[i];
}
__device__ __forceinline__ void q_store_ext(dev_ext* dst, const qpt& p, int q) {
  uint32_t* w = reinterpret_cast<uint32_t*>(dst) + 9 * q;
#pragma unroll
  for (int i = 0; i < 9; ++i) w[i] = p.c.v[i];
}
// affine niels operand (y+x, y-x, 2dxy) of a decoded point as a quad cached operand (Z = 1 -> 2Z = 2)
__device__ __forceinline__ void q_load_niels(qcached& c, const dev_niels* src, int q, uint32_t negate) {
  // lane 0 wants y-x, lane 1 y+x, lane 3 2dxy; negation swaps the first two and negates the last
  const uint32_t* w = reinterpret_cast<const uint32_t*>(src);
  const int sel = q == 0 ? (negate ? 0 : 9) : (q == 1 ? (negate ? 9 : 0) : 18);
  fe t;
#pragma unroll
  for (int i = 0; i < 9; ++i) t.v[i] = w[sel + i];
  fe two;
  fe_0(two);
  two.v[0] = 2;
  fe n;
  fe_neg(n, t);
  fe_carry(n, n);
  c.c = t;
  fe_pick(c.c, n, q == 3 && negate);
  fe_pick(c.c, two, q == 2);
}

// cached form of a quad point: (Y-X, Y+X, 2Z, 2d T)   -- one multiplication level (only lane 3 needs it)
__device__ __forceinline__ void q_to_cached(qcached& r, const qpt& p, int q) {
  fe o, s, a, d2, one, m;
  fe_dpp<ZKP_QP(1, 0, 3, 2)>(o, p.c);       // lane 0 <- Y, lane 1 <- X
  fe_sub(s, o, p.c);                         // lane 0: Y - X
  fe_add(a, p.c, o);                         // lane 1: Y + X ; lane 2 (o = T): unused
  fe z2;
  fe_add(z2, p.c, p.c);                      // lane 2: 2Z
  fe t = p.c;
  fe_pick(t, s, q == 0);
  fe_pick(t, a, q == 1);
  fe_pick(t, z2, q == 2);
  fe_carry(t, t);
  fe_from_const(d2, FE_D2);
  fe_1(one);
  m = one;
  fe_pick(m, d2, q == 3);
  fe_mul(r.c, t, m);                         // lanes 0..2: x1 (stays tight), lane 3: T * 2d
}

// r = p + c   (two multiplication levels)
__device__ __forceinline__ void q_add_cached(qpt& r, const qpt& p, const qcached& c, int q) {
  fe o, s, a, t, u, v;
  fe_dpp<ZKP_QP(1, 0, 3, 2)>(o, p.c);
  fe_sub(s, o, p.c);                         // lane 0: Y1 - X1
  fe_add(a, p.c, o);                         // lane 1: Y1 + X1
  t = p.c;                                   // lane 2: Z1, lane 3: T1
  fe_pick(t, s, q == 0);
  fe_pick(t, a, q == 1);
  fe_mul(u, t, c.c);                         // lane 0: A, 1: B, 2: D = Z1 * 2Z2, 3: C = T1 * 2dT2
  fe_dpp<ZKP_QP(1, 0, 3, 2)>(o, u);
  fe sum, d1, d2;
  fe_add(sum, u, o);                         // lane 1: H = B + A ; lane 3: G = C + D
  fe_sub(d1, o, u);                          // lane 0: E = B - A
  fe_sub(d2, u, o);                          // lane 2: F = D - C
  v = sum;
  fe_pick(v, d1, q == 0);
  fe_pick(v, d2, q == 2);
  fe_carry(v, v);                            // lane 0: E, 1: H, 2: F, 3: G   (tight)
  // X3 = E F (lane 0 x lane 2), T3 = H E (lane 1 x lane 0), Z3 = F G (lane 2 x lane 3), Y3 = G H (lane 3 x lane 1)
  fe_dpp<ZKP_QP(2, 0, 3, 1)>(o, v);
  fe_mul(u, v, o);                           // lane 0: X3, 1: T3, 2: Z3, 3: Y3
  fe_dpp<ZKP_QP(0, 3, 2, 1)>(r.c, u);        // back to (X, Y, Z, T)
}

// r = p + s for two quad points (three multiplication levels)
__device__ __forceinline__ void q_add(qpt& r, const qpt& p, const qpt& s, int q) {
  qcached c;
  q_to_cached(c, s, q);
  q_add_cached(r, p, c, q);
}

// r = 2p   (one squaring level + one multiplication level)
__device__ __forceinline__ void q_double(qpt& r, const qpt& p, int q) {
  fe x, y, t, s, a, b, h, g, e, f, w;
  fe_dpp<ZKP_QP(0, 0, 0, 0)>(x, p.c);
  fe_dpp<ZKP_QP(1, 1, 1, 1)>(y, p.c);
  fe_add(w, x, y);
  t = p.c;
  fe_pick(t, w, q == 3);                     // lane 3 squares X + Y instead of T
  fe_sq(s, t);                               // lane 0: XX, 1: YY, 2: ZZ, 3: (X+Y)^2
  fe_dpp<ZKP_QP(0, 0, 0, 0)>(a, s);
  fe_dpp<ZKP_QP(1, 1, 1, 1)>(b, s);
  fe_add(h, b, a);                           // YY + XX  (completed Y)
  fe_sub(g, b, a);                           // YY - XX  (completed Z)
  fe_sub4(e, s, h);                          // lane 3: (X+Y)^2 - H  (completed X)
  fe_add(w, s, s);
  fe_sub4(f, w, g);                          // lane 2: 2 ZZ - G     (completed T)
  fe_carry(e, e);
  fe_carry(f, f);
  fe ee, ff;
  fe_dpp<ZKP_QP(3, 3, 3, 3)>(ee, e);
  fe_dpp<ZKP_QP(2, 2, 2, 2)>(ff, f);
  // lane 0: X3 = E F, lane 1: Y3 = G H, lane 2: Z3 = F G, lane 3: T3 = E H
  fe m1 = ee, m2 = h;
  fe_pick(m1, g, q == 1);
  fe_pick(m1, ff, q == 2);
  fe_pick(m2, ff, q == 0);
  fe_pick(m2, g, q == 2);
  fe_mul(r.c, m1, m2);
}

// gather the four coordinates of a quad point into every lane (for code that continues with one lane per point)
__device__ __forceinline__ void q_gather(ge_p3& out, const qpt& p) {
  fe_dpp<ZKP_QP(0, 0, 0, 0)>(out.X, p.c);
  fe_dpp<ZKP_QP(1, 1, 1, 1)>(out.Y, p.c);
  fe_dpp<ZKP_QP(2, 2, 2, 2)>(out.Z, p.c);
  fe_dpp<ZKP_QP(3, 3, 3, 3)>(out.T, p.c);
}

}  // namespace zkp
