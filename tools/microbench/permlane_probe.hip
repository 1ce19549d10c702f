// What v_permlane32_swap_b32 / v_permlane16_swap_b32 (gfx950) do to a wavefront, printed as "source row of every output row":
//   hipcc --offload-arch=gfx950 -O2 tools/microbench/permlane_probe.hip -o /tmp/permlane_probe && /tmp/permlane_probe
// zkp_amd/csrc/rowfe.h relies on: swap32(x, x) = { rows 0 1 0 1, rows 2 3 2 3 }; swap16(y, y) with y = rows (p q p q) = { p p p p, q q q q }.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
__global__ void k(uint32_t* o) {
  const uint32_t x = threadIdx.x;                       // lane id
  const auto h = __builtin_amdgcn_permlane32_swap(x, x, false, false);
  const auto lo = __builtin_amdgcn_permlane16_swap(h[0], h[0], false, false);
  const auto hi = __builtin_amdgcn_permlane16_swap(h[1], h[1], false, false);
  const auto raw16 = __builtin_amdgcn_permlane16_swap(x, x, false, false);
  uint32_t v[8] = {h[0], h[1], lo[0], lo[1], hi[0], hi[1], raw16[0], raw16[1]};
  for (int i = 0; i < 8; ++i) o[64 * i + threadIdx.x] = v[i];
}
int main() {
  uint32_t* d;
  uint32_t h[8 * 64];
  if (hipMalloc(&d, sizeof(h)) != hipSuccess) return 1;
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
  if (hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost) != hipSuccess) return 1;
  const char* name[8] = {"swap32(x,x)[0]", "swap32(x,x)[1]", "swap16(h0,h0)[0]", "swap16(h0,h0)[1]", "swap16(h1,h1)[0]", "swap16(h1,h1)[1]", "swap16(x,x)[0]", "swap16(x,x)[1]"};
  for (int i = 0; i < 8; ++i) {
    printf("%-18s rows from:", name[i]);
    bool clean = true;
    for (int r = 0; r < 4; ++r) {
      const uint32_t src = h[64 * i + 16 * r];
      for (int k2 = 0; k2 < 16; ++k2) clean = clean && h[64 * i + 16 * r + k2] == src + k2;
      printf(" %u", src >> 4);
    }
    printf("%s\n", clean ? "   (lane order inside rows kept)" : "   (NOT whole rows)");
  }
  return 0;
}
