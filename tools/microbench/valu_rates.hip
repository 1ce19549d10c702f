// gfx950 VALU issue-rate micro-benchmark for the integer / fp64 instructions a 255-bit
// field multiplier can be built from.  SURVEY.md §7/§8(d): the local guides give no
// integer-multiply throughput, so the limb representation is picked from these numbers.
//
// Each kernel runs ITER iterations of 8 independent dependency chains of ONE instruction,
// with 8 waves per SIMD (2048 threads per CU) so issue, not latency, is measured.
// Output: wave-instructions per cycle per CU relative to v_fma_f32 (known: 2 cyc / wave64 / SIMD).
//
// Build: hipcc --offload-arch=gfx950 -O3 tools/microbench/valu_rates.hip -o tools/microbench/valu_rates
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>
#include <string>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { \
  fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1);} } while (0)

constexpr int ITER = 2048;
constexpr int CHAINS = 8;
constexpr int REP = 4;       // the 8-chain group is repeated REP times per loop iteration: 32 VALU instructions per s_add / s_cmp / s_cbranch
#define R8(ASM) ASM(0) ASM(1) ASM(2) ASM(3) ASM(4) ASM(5) ASM(6) ASM(7)

// 32-bit chains: x_k = OP(x_k, a, b)
#define KERNEL32(NAME, ASM)                                                            \
__global__ void __launch_bounds__(256) NAME(uint32_t* out, uint32_t a, uint32_t b) {   \
  uint32_t x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3,                    \
           x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;                         \
  uint32_t va = a + threadIdx.x, vb = b ^ threadIdx.x;                                 \
  asm volatile("s_mov_b32 s20, 0x55555555\n s_mov_b32 s21, 0x55555555\n s_mov_b32 vcc_lo, 0x33333333\n s_mov_b32 vcc_hi, 0x33333333" ::: "s20", "s21", "vcc"); \
  for (int i = 0; i < ITER; ++i) {                                                     \
    asm volatile(R8(ASM) R8(ASM) R8(ASM) R8(ASM)                                  \
      : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) \
        : "v"(va), "v"(vb) : "vcc", "s20", "s21", "s22");                                                     \
  }                                                                                    \
  out[blockIdx.x * blockDim.x + threadIdx.x] = x0 ^ x1 ^ x2 ^ x3 ^ x4 ^ x5 ^ x6 ^ x7;  \
}

// 64-bit chains: x_k (register pair) = OP(a, b, x_k)
#define KERNEL64(NAME, ASM)                                                            \
__global__ void __launch_bounds__(256) NAME(uint32_t* out, uint32_t a, uint32_t b) {   \
  uint64_t x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3,                    \
           x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;                         \
  uint32_t va = a + threadIdx.x, vb = b ^ threadIdx.x;                                 \
  asm volatile("s_mov_b32 s20, 0x55555555\n s_mov_b32 s21, 0x55555555\n s_mov_b32 vcc_lo, 0x33333333\n s_mov_b32 vcc_hi, 0x33333333" ::: "s20", "s21", "vcc"); \
  for (int i = 0; i < ITER; ++i) {                                                     \
    asm volatile(R8(ASM) R8(ASM) R8(ASM) R8(ASM)                                  \
      : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) \
        : "v"(va), "v"(vb) : "vcc", "s20", "s21", "s22");                                                     \
  }                                                                                    \
  uint64_t r = x0 ^ x1 ^ x2 ^ x3 ^ x4 ^ x5 ^ x6 ^ x7;                                  \
  out[blockIdx.x * blockDim.x + threadIdx.x] = (uint32_t)r ^ (uint32_t)(r >> 32);      \
}

// fp64 chains
#define KERNELF64(NAME, ASM)                                                           \
__global__ void __launch_bounds__(256) NAME(uint32_t* out, uint32_t a, uint32_t b) {   \
  double x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3,                      \
         x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;                           \
  double va = 1.0 + 1e-9 * a, vb = 1e-9 * b;                                           \
  asm volatile("s_mov_b32 s20, 0x55555555\n s_mov_b32 s21, 0x55555555\n s_mov_b32 vcc_lo, 0x33333333\n s_mov_b32 vcc_hi, 0x33333333" ::: "s20", "s21", "vcc"); \
  for (int i = 0; i < ITER; ++i) {                                                     \
    asm volatile(R8(ASM) R8(ASM) R8(ASM) R8(ASM)                                  \
      : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) \
        : "v"(va), "v"(vb) : "vcc", "s20", "s21", "s22");                                                     \
  }                                                                                    \
  double r = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;                                    \
  out[blockIdx.x * blockDim.x + threadIdx.x] = (uint32_t)(int)r;                       \
}

#define A_FMA_F32(k)      "v_fma_f32 %" #k ", %8, %9, %" #k "\n"
#define A_ADD_U32(k)      "v_add_u32 %" #k ", %8, %" #k "\n"
#define A_ADD3_U32(k)     "v_add3_u32 %" #k ", %8, %9, %" #k "\n"
#define A_ADD_CO(k)       "v_add_co_u32 %" #k ", vcc, %8, %" #k "\n"
#define A_ADDC_CO(k)      "v_addc_co_u32 %" #k ", vcc, %8, %" #k ", vcc\n"
#define A_MUL_LO(k)       "v_mul_lo_u32 %" #k ", %8, %" #k "\n"
#define A_MUL_HI(k)       "v_mul_hi_u32 %" #k ", %8, %" #k "\n"
#define A_MUL_U24(k)      "v_mul_u32_u24 %" #k ", %8, %" #k "\n"
#define A_MUL_HI_U24(k)   "v_mul_hi_u32_u24 %" #k ", %8, %" #k "\n"
#define A_MAD_U24(k)      "v_mad_u32_u24 %" #k ", %8, %9, %" #k "\n"
#define A_MAD_U16(k)      "v_mad_u32_u16 %" #k ", %8, %9, %" #k "\n"
#define A_AND_OR(k)       "v_and_or_b32 %" #k ", %8, %9, %" #k "\n"
#define A_ALIGNBIT(k)     "v_alignbit_b32 %" #k ", %8, %" #k ", 13\n"
#define A_LSHL_ADD(k)     "v_lshl_add_u32 %" #k ", %8, 3, %" #k "\n"
#define A_CNDMASK(k)      "v_cndmask_b32 %" #k ", %8, %" #k ", vcc\n"
#define A_DOT2_U16(k)     "v_dot2_u32_u16 %" #k ", %8, %9, %" #k "\n"
#define A_DOT4_U8(k)      "v_dot4_u32_u8 %" #k ", %8, %9, %" #k "\n"
#define A_PK_MAD_U16(k)   "v_pk_mad_u16 %" #k ", %8, %9, %" #k "\n"
#define A_PK_MUL_LO(k)    "v_pk_mul_lo_u16 %" #k ", %8, %" #k "\n"

#define A_MAD_U64(k)      "v_mad_u64_u32 %" #k ", vcc, %8, %9, %" #k "\n"
#define A_LSHL_ADD_U64(k) "v_lshl_add_u64 %" #k ", %" #k ", 0, %" #k "\n"
#define A_LSHLREV_B64(k)  "v_lshlrev_b64 %" #k ", 1, %" #k "\n"
#define A_LSHRREV_B64(k)  "v_lshrrev_b64 %" #k ", 1, %" #k "\n"

#define A_FMA_F64(k)      "v_fma_f64 %" #k ", %8, %" #k ", %9\n"
#define A_MUL_F64(k)      "v_mul_f64 %" #k ", %8, %" #k "\n"
#define A_ADD_F64(k)      "v_add_f64 %" #k ", %9, %" #k "\n"
#define A_PK_FMA_F32(k)   "v_pk_fma_f32 %" #k ", %" #k ", %" #k ", %" #k "\n"


// ---- round 4: every opcode with >= 1 % of the shipped kernels' VALU mix (tools/opcode_mix.py), so that the VALU ceiling can be
//      cycle-weighted instead of dividing raw instruction counts by the 4-cycle peak
#define A_AND_B32(k)      "v_and_b32 %" #k ", %8, %" #k "\n"
#define A_OR_B32(k)       "v_or_b32 %" #k ", %8, %" #k "\n"
#define A_XOR_B32(k)      "v_xor_b32 %" #k ", %8, %" #k "\n"
#define A_SUB_U32(k)      "v_sub_u32 %" #k ", %" #k ", %8\n"
#define A_LSHRREV_B32(k)  "v_lshrrev_b32 %" #k ", 3, %" #k "\n"
#define A_LSHLREV_B32(k)  "v_lshlrev_b32 %" #k ", 3, %" #k "\n"
#define A_ASHRREV_I32(k)  "v_ashrrev_i32 %" #k ", 3, %" #k "\n"
#define A_MOV_B32(k)      "v_mov_b32 %" #k ", %8\n"
#define A_MOV_DPP(k)      "v_mov_b32_dpp %" #k ", %" #k " quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
#define A_BFE_U32(k)      "v_bfe_u32 %" #k ", %" #k ", 3, 17\n"
#define A_LSHL_OR(k)      "v_lshl_or_b32 %" #k ", %8, 3, %" #k "\n"
#define A_OR3(k)          "v_or3_b32 %" #k ", %8, %9, %" #k "\n"
#define A_BITOP3(k)       "v_bitop3_b32 %" #k ", %8, %9, %" #k " bitop3:0xd2\n"
#define A_BFI(k)          "v_bfi_b32 %" #k ", %8, %9, %" #k "\n"
#define A_CNDMASK_S(k)    "v_cndmask_b32 %" #k ", %8, %" #k ", s[20:21]\n"
#define A_CNDMASK_VCC(k)  "v_cndmask_b32 %" #k ", %8, %" #k ", vcc\n"
#define A_CMP_EQ(k)       "v_cmp_eq_u32 vcc, %8, %" #k "\n"
#define A_MAX_U32(k)      "v_max_u32 %" #k ", %8, %" #k "\n"
#define A_ADD_LSHL(k)     "v_add_lshl_u32 %" #k ", %8, %" #k ", 1\n"
#define A_MOV_B64(k)      "v_mov_b64 %" #k ", %" #k "\n"
#define A_READLANE(k)     "v_readlane_b32 s22, %" #k ", 5\n"

KERNEL32(k_fma_f32, A_FMA_F32)
KERNEL32(k_add_u32, A_ADD_U32)
KERNEL32(k_add3_u32, A_ADD3_U32)
KERNEL32(k_add_co, A_ADD_CO)
KERNEL32(k_addc_co, A_ADDC_CO)
KERNEL32(k_mul_lo, A_MUL_LO)
KERNEL32(k_mul_hi, A_MUL_HI)
KERNEL32(k_mul_u24, A_MUL_U24)
KERNEL32(k_mul_hi_u24, A_MUL_HI_U24)
KERNEL32(k_mad_u24, A_MAD_U24)
KERNEL32(k_mad_u16, A_MAD_U16)
KERNEL32(k_and_or, A_AND_OR)
KERNEL32(k_alignbit, A_ALIGNBIT)
KERNEL32(k_lshl_add, A_LSHL_ADD)
KERNEL32(k_cndmask, A_CNDMASK)
KERNEL32(k_dot2_u16, A_DOT2_U16)
KERNEL32(k_dot4_u8, A_DOT4_U8)
KERNEL32(k_pk_mad_u16, A_PK_MAD_U16)
KERNEL32(k_pk_mul_lo, A_PK_MUL_LO)
KERNEL64(k_mad_u64, A_MAD_U64)
KERNEL64(k_lshl_add_u64, A_LSHL_ADD_U64)
KERNEL64(k_lshlrev_b64, A_LSHLREV_B64)
KERNEL64(k_lshrrev_b64, A_LSHRREV_B64)
KERNEL64(k_pk_fma_f32, A_PK_FMA_F32)
KERNEL32(k_and_b32, A_AND_B32)
KERNEL32(k_or_b32, A_OR_B32)
KERNEL32(k_xor_b32, A_XOR_B32)
KERNEL32(k_sub_u32, A_SUB_U32)
KERNEL32(k_lshrrev_b32, A_LSHRREV_B32)
KERNEL32(k_lshlrev_b32, A_LSHLREV_B32)
KERNEL32(k_ashrrev_i32, A_ASHRREV_I32)
KERNEL32(k_mov_b32, A_MOV_B32)
KERNEL32(k_mov_dpp, A_MOV_DPP)
KERNEL32(k_bfe_u32, A_BFE_U32)
KERNEL32(k_lshl_or, A_LSHL_OR)
KERNEL32(k_or3, A_OR3)
KERNEL32(k_bitop3, A_BITOP3)
KERNEL32(k_bfi, A_BFI)
KERNEL32(k_cndmask_s, A_CNDMASK_S)
KERNEL32(k_cndmask_vcc, A_CNDMASK_VCC)
KERNEL32(k_cmp_eq, A_CMP_EQ)
KERNEL32(k_max_u32, A_MAX_U32)
KERNEL32(k_add_lshl, A_ADD_LSHL)
KERNEL32(k_readlane, A_READLANE)
KERNEL64(k_mov_b64, A_MOV_B64)
KERNELF64(k_fma_f64, A_FMA_F64)
KERNELF64(k_mul_f64, A_MUL_F64)
KERNELF64(k_add_f64, A_ADD_F64)

typedef void (*kern_t)(uint32_t*, uint32_t, uint32_t);
struct Entry { const char* name; kern_t k; };

int main() {
  hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  const double clk_ghz = prop.clockRate * 1e-6;
  printf("device %s  CUs %d  clockRate %.3f GHz\n", prop.name, cus, clk_ghz);
  const int blocks = cus * 8;   // 8 blocks of 256 threads per CU = 8 waves / SIMD
  uint32_t* out; CK(hipMalloc(&out, (size_t)blocks * 256 * 4));
  std::vector<Entry> es = {
    {"v_fma_f32", k_fma_f32}, {"v_add_u32", k_add_u32}, {"v_add3_u32", k_add3_u32},
    {"v_add_co_u32", k_add_co}, {"v_addc_co_u32", k_addc_co},
    {"v_mul_lo_u32", k_mul_lo}, {"v_mul_hi_u32", k_mul_hi},
    {"v_mul_u32_u24", k_mul_u24}, {"v_mul_hi_u32_u24", k_mul_hi_u24}, {"v_mad_u32_u24", k_mad_u24},
    {"v_mad_u32_u16", k_mad_u16}, {"v_and_or_b32", k_and_or}, {"v_alignbit_b32", k_alignbit},
    {"v_lshl_add_u32", k_lshl_add}, {"v_cndmask_b32", k_cndmask},
    {"v_dot2_u32_u16", k_dot2_u16}, {"v_dot4_u32_u8", k_dot4_u8},
    {"v_pk_mad_u16", k_pk_mad_u16}, {"v_pk_mul_lo_u16", k_pk_mul_lo},
    {"v_mad_u64_u32", k_mad_u64}, {"v_lshl_add_u64", k_lshl_add_u64},
    {"v_lshlrev_b64", k_lshlrev_b64}, {"v_lshrrev_b64", k_lshrrev_b64},
    {"v_pk_fma_f32", k_pk_fma_f32},
    {"v_and_b32", k_and_b32},
    {"v_or_b32", k_or_b32},
    {"v_xor_b32", k_xor_b32},
    {"v_sub_u32", k_sub_u32},
    {"v_lshrrev_b32", k_lshrrev_b32},
    {"v_lshlrev_b32", k_lshlrev_b32},
    {"v_ashrrev_i32", k_ashrrev_i32},
    {"v_mov_b32", k_mov_b32},
    {"v_mov_b32_dpp(quad_perm)", k_mov_dpp},
    {"v_bfe_u32", k_bfe_u32},
    {"v_lshl_or_b32", k_lshl_or},
    {"v_or3_b32", k_or3},
    {"v_bitop3_b32", k_bitop3},
    {"v_bfi_b32", k_bfi},
    {"v_cndmask_b32(sgpr mask)", k_cndmask_s},
    {"v_cndmask_b32(vcc, vcc const)", k_cndmask_vcc},
    {"v_cmp_eq_u32(->vcc)", k_cmp_eq},
    {"v_max_u32", k_max_u32},
    {"v_add_lshl_u32", k_add_lshl},
    {"v_readlane_b32", k_readlane},
    {"v_mov_b64", k_mov_b64},
    {"v_fma_f64", k_fma_f64}, {"v_mul_f64", k_mul_f64}, {"v_add_f64", k_add_f64},
  };
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  printf("%-30s %10s %14s %16s\n", "instr", "ms", "Ginstr-lane/s", "cyc/wave/SIMD@clk");
  for (auto& e : es) {
    for (int rep = 0; rep < 2; ++rep) {   // first = warm-up
      CK(hipEventRecord(e0));
      hipLaunchKernelGGL(e.k, dim3(blocks), dim3(256), 0, 0, out, 3u, 5u);
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      if (rep == 0) continue;
      const double lane_instr = (double)blocks * 256 * ITER * CHAINS * REP;
      const double wave_instr_per_simd = (double)blocks * 4 / (cus * 4.0) * ITER * CHAINS * REP;  // waves per SIMD * instr
      const double cyc = ms * 1e-3 * clk_ghz * 1e9 / wave_instr_per_simd;
      printf("%-30s %10.3f %14.1f %16.2f\n", e.name, ms, lane_instr / (ms * 1e-3) * 1e-9, cyc);
    }
  }
  CK(hipFree(out));
  return 0;
}
