// v_permlane32_swap_b32 on gfx950: which halves of which operand change places?  (used by the weight gradient's consumer waves to
// hand the centre block of a window to the other half-wave instead of reading it from LDS again)
//   hipcc --offload-arch=gfx950 -O3 tools/permlane_probe.hip -o /tmp/permlane_probe && /tmp/permlane_probe
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned* out) {
  unsigned a = threadIdx.x, b = threadIdx.x + 100;
  auto r = __builtin_amdgcn_permlane32_swap(a, b, false, false);
  out[threadIdx.x] = r[0];
  out[64 + threadIdx.x] = r[1];
}
int main() {
  unsigned* d; unsigned h[128];
  (void)hipMalloc(&d, sizeof(h));
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
  (void)hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  printf("a = lane, b = lane + 100;  (a', b') = permlane32_swap(a, b)\n");
  for (int l : {0, 31, 32, 63}) printf("  lane %2d: a' = %3u  b' = %3u\n", l, h[l], h[64 + l]);
  return 0;
}
