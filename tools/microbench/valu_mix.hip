// gfx950: what does a 2-cycle-class VALU instruction cost when it sits BETWEEN 4-cycle-class ones?  (profiles/r04_valu_mix_microbench.txt)
//
// valu_rates.hip measures single-opcode streams: 2 issue cycles per wave64 instruction for v_and / v_add_u32 / v_mov ..., 4 for v_mad_u64_u32 and the
// rest.  The kernels of this repository are MIXES (term kernel: 23 % 2-cycle opcodes scattered between v_mad_u64_u32), and SQ_ACTIVE_INST_VALU2 -- which
// reads 0.479 per instruction on a pure 2-cycle stream and 0 on a pure 4-cycle one -- reads far less on them than their static mix predicts.  So: do
// isolated 2-cycle instructions still issue in 2 cycles?  Each kernel below repeats one pattern of M = v_mad_u64_u32 and A = v_and_b32 on independent
// register chains, 8 waves per SIMD; cycles per instruction come from GRBM_GUI_ACTIVE (tools/valu_mix.sh), VALU2 alongside.
//
// Build: hipcc --offload-arch=gfx950 -O3 tools/microbench/valu_mix.hip -o tools/microbench/valu_mix
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1);} } while (0)
constexpr int ITER = 2048;

// operands: %0..%3 64-bit chains (mad), %4..%7 32-bit chains (and), %8 / %9 inputs
#define M(k) "v_mad_u64_u32 %" #k ", vcc, %8, %9, %" #k "\n"
#define A(k) "v_and_b32 %" #k ", %8, %" #k "\n"
#define KERNEL(NAME, BODY, NINSTR)                                                                         \
__global__ void __launch_bounds__(256) NAME(uint32_t* out, uint32_t a, uint32_t b) {                       \
  uint64_t y0 = threadIdx.x, y1 = y0 + 1, y2 = y0 + 2, y3 = y0 + 3;                                        \
  uint32_t x0 = threadIdx.x + 4, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3;                                    \
  uint32_t va = a + threadIdx.x, vb = b ^ threadIdx.x;                                                     \
  for (int i = 0; i < ITER; ++i) {                                                                         \
    asm volatile(BODY BODY BODY BODY                                                                       \
      : "+v"(y0), "+v"(y1), "+v"(y2), "+v"(y3), "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(va), "v"(vb) : "vcc"); \
  }                                                                                                        \
  const uint64_t r = y0 ^ y1 ^ y2 ^ y3;                                                                    \
  out[blockIdx.x * blockDim.x + threadIdx.x] = (uint32_t)r ^ (uint32_t)(r >> 32) ^ x0 ^ x1 ^ x2 ^ x3;      \
}
KERNEL(k_MMMMMMMM, M(0) M(1) M(2) M(3) M(0) M(1) M(2) M(3), 8)
KERNEL(k_AAAAAAAA, A(4) A(5) A(6) A(7) A(4) A(5) A(6) A(7), 8)
KERNEL(k_MAMAMAMA, M(0) A(4) M(1) A(5) M(2) A(6) M(3) A(7), 8)
KERNEL(k_MMAAMMAA, M(0) M(1) A(4) A(5) M(2) M(3) A(6) A(7), 8)
KERNEL(k_MMMAMMMA, M(0) M(1) M(2) A(4) M(3) M(0) M(1) A(5), 8)
KERNEL(k_MAAAMAAA, M(0) A(4) A(5) A(6) M(1) A(7) A(4) A(5), 8)
KERNEL(k_MMMMAAAA, M(0) M(1) M(2) M(3) A(4) A(5) A(6) A(7), 8)
KERNEL(k_MMMMMMMA, M(0) M(1) M(2) M(3) M(0) M(1) M(2) A(4), 8)
// run length: how long must a run of 2-cycle instructions be before they share slots?  (names: M<count>A<count>)
#define M4 M(0) M(1) M(2) M(3)
#define A4 A(4) A(5) A(6) A(7)
// B = v_add_u32, S = v_lshrrev_b32: other opcodes of the 2-cycle class, mixed into a run
#define B(k) "v_add_u32 %" #k ", %8, %" #k "\n"
#define S(k) "v_lshrrev_b32 %" #k ", 3, %" #k "\n"
KERNEL(k_M4A8, M4 A4 A4, 12)
KERNEL(k_M8A8, M4 M4 A4 A4, 16)
KERNEL(k_M8A16, M4 M4 A4 A4 A4 A4, 24)
KERNEL(k_M16A16, M4 M4 M4 M4 A4 A4 A4 A4, 32)
KERNEL(k_M16A32, M4 M4 M4 M4 A4 A4 A4 A4 A4 A4 A4 A4, 48)
KERNEL(k_M8X8, M4 M4 A(4) B(5) S(6) A(7) B(4) S(5) A(6) B(7), 16)
KERNEL(k_X8, A(4) B(5) S(6) A(7) B(4) S(5) A(6) B(7), 8)

typedef void (*kern_t)(uint32_t*, uint32_t, uint32_t);
struct Entry { const char* name; kern_t k; int n2, n4; };

int main() {
  hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  const int blocks = cus * 8;
  uint32_t* out; CK(hipMalloc(&out, (size_t)blocks * 256 * 4));
  Entry es[] = {{"MMMMMMMM", k_MMMMMMMM, 0, 8}, {"AAAAAAAA", k_AAAAAAAA, 8, 0}, {"MAMAMAMA", k_MAMAMAMA, 4, 4}, {"MMAAMMAA", k_MMAAMMAA, 4, 4}, {"MMMAMMMA", k_MMMAMMMA, 2, 6},
                {"MAAAMAAA", k_MAAAMAAA, 6, 2}, {"MMMMAAAA", k_MMMMAAAA, 4, 4}, {"MMMMMMMA", k_MMMMMMMA, 1, 7},
                {"M4A8", k_M4A8, 8, 4}, {"M8A8", k_M8A8, 8, 8}, {"M8A16", k_M8A16, 16, 8}, {"M16A16", k_M16A16, 16, 16}, {"M16A32", k_M16A32, 32, 16},
                {"M8X8", k_M8X8, 8, 8}, {"X8", k_X8, 8, 0}};
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  printf("%-10s %8s %8s %10s %22s\n", "pattern", "2-cycle", "4-cycle", "ms", "weighted cycles/instr");
  for (auto& e : es) {
    for (int rep = 0; rep < 2; ++rep) {
      CK(hipEventRecord(e0));
      hipLaunchKernelGGL(e.k, dim3(blocks), dim3(256), 0, 0, out, 3u, 5u);
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      if (rep) printf("%-10s %8d %8d %10.3f %22.2f\n", e.name, e.n2, e.n4, ms, (2.0 * e.n2 + 4.0 * e.n4) / (e.n2 + e.n4));
    }
  }
  CK(hipFree(out));
  return 0;
}
