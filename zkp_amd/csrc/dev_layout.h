// HBM record layouts shared by the kernels, and 16-byte vectorised load/store helpers.
//
// All per-point records are array-of-structures with a 16-byte aligned size: the consumers are
// GATHERS (a lane fetches the point some index names), so one lane must find its whole record in
// one or two contiguous sectors; a limb-major SoA layout would turn every gather into 27-36 scattered
// 4-byte reads.  Streams that are read in lane order (encodings, scalars, digits, sorted index lists)
// are plain contiguous arrays so that a wave reads 1-2 KiB contiguous per instruction.
#pragma once
#include <hip/hip_runtime.h>
#include "ge25519.h"

namespace zkp {

struct alignas(16) dev_affine {   // decoded point, Z = 1 implied            112 B
  uint32_t x[9], y[9], t[9];
  uint32_t valid;
};
struct alignas(16) dev_niels {    // (y+x, y-x, 2dxy) of a decoded point      112 B
  uint32_t ypx[9], ymx[9], xy2d[9];
  uint32_t valid;
};
struct alignas(16) dev_ext {      // extended (X:Y:Z:T)                       144 B
  uint32_t X[9], Y[9], Z[9], T[9];
};
static_assert(sizeof(dev_affine) == 112 && sizeof(dev_niels) == 112 && sizeof(dev_ext) == 144, "layout");

template <int NVEC>
__device__ __forceinline__ void load_vec(uint32_t* dst, const void* src) {
  const uint4* s = reinterpret_cast<const uint4*>(src);
#pragma unroll
  for (int i = 0; i < NVEC; ++i) {
    const uint4 v = s[i];
    dst[4 * i + 0] = v.x; dst[4 * i + 1] = v.y; dst[4 * i + 2] = v.z; dst[4 * i + 3] = v.w;
  }
}
template <int NVEC>
__device__ __forceinline__ void store_vec(void* dst, const uint32_t* src) {
  uint4* d = reinterpret_cast<uint4*>(dst);
#pragma unroll
  for (int i = 0; i < NVEC; ++i) d[i] = make_uint4(src[4 * i + 0], src[4 * i + 1], src[4 * i + 2], src[4 * i + 3]);
}

__device__ __forceinline__ void fe_set(fe& r, const uint32_t* w) {
#pragma unroll
  for (int i = 0; i < 9; ++i) r.v[i] = w[i];
}
__device__ __forceinline__ void fe_get(uint32_t* w, const fe& a) {
#pragma unroll
  for (int i = 0; i < 9; ++i) w[i] = a.v[i];
}

__device__ __forceinline__ void load_ext(ge_p3& p, const dev_ext* src) {
  uint32_t w[36];
  load_vec<9>(w, src);
  fe_set(p.X, w); fe_set(p.Y, w + 9); fe_set(p.Z, w + 18); fe_set(p.T, w + 27);
}
__device__ __forceinline__ void store_ext(dev_ext* dst, const ge_p3& p) {
  uint32_t w[36];
  fe_get(w, p.X); fe_get(w + 9, p.Y); fe_get(w + 18, p.Z); fe_get(w + 27, p.T);
  store_vec<9>(dst, w);
}
// returns the valid flag
__device__ __forceinline__ uint32_t load_affine(ge_p3& p, const dev_affine* src) {
  uint32_t w[28];
  load_vec<7>(w, src);
  fe_set(p.X, w); fe_set(p.Y, w + 9); fe_1(p.Z); fe_set(p.T, w + 18);
  return w[27];
}
__device__ __forceinline__ void store_affine(dev_affine* dst, const ge_p3& p, uint32_t valid) {
  uint32_t w[28];
  fe_get(w, p.X); fe_get(w + 9, p.Y); fe_get(w + 18, p.T);
  w[27] = valid;
  store_vec<7>(dst, w);
}
__device__ __forceinline__ uint32_t load_niels(ge_niels& q, const dev_niels* src) {
  uint32_t w[28];
  load_vec<7>(w, src);
  fe_set(q.ypx, w); fe_set(q.ymx, w + 9); fe_set(q.xy2d, w + 18);
  return w[27];
}
__device__ __forceinline__ void store_niels(dev_niels* dst, const ge_niels& q, uint32_t valid) {
  uint32_t w[28];
  fe_get(w, q.ypx); fe_get(w + 9, q.ymx); fe_get(w + 18, q.xy2d);
  w[27] = valid;
  store_vec<7>(dst, w);
}

// Pins every limb of p in a VGPR.  Needed where a kernel works on wave-UNIFORM data (one lane, or all lanes
// reading the same address): hipcc's uniformity analysis would otherwise move the whole field arithmetic to
// the scalar ALU (s_mul_hi_u32 / s_mul_i32 / s_addc_u32), measured 2.3x slower than v_mad_u64_u32 for a lone
// wave (tools/microbench/ge_chain.hip: 4.05 us vs 1.74 us per doubling).
__device__ __forceinline__ void fe_pin_vgpr(fe& a) {
#pragma unroll
  for (int i = 0; i < 9; ++i) asm volatile("" : "+v"(a.v[i]));
}
__device__ __forceinline__ void ge_pin_vgpr(ge_p3& p) {
  fe_pin_vgpr(p.X); fe_pin_vgpr(p.Y); fe_pin_vgpr(p.Z); fe_pin_vgpr(p.T);
}

// r = p + q for two extended points (9M)
__device__ __forceinline__ void ge_add_p3(ge_p3& r, const ge_p3& p, const ge_p3& q) {
  ge_cached c;
  ge_to_cached(c, q);
  ge_add_cached(r, p, c);
}

}  // namespace zkp
