// What does SQ_LDS_BANK_CONFLICT count for 16-byte LDS reads?  (every variant of the convolution shows
// "conflicts = 50 % of LDS-active cycles".)  Five read patterns, each its own kernel so that rocprofv3 --pmc reports them apart:
//   b128_linear   lane i reads 16-byte entry i            — conflict-free by construction (64 lanes x 16 B = 1 KB contiguous)
//   b64_linear    lane i reads 8-byte entry i              — conflict-free
//   b32_linear    lane i reads dword i                     — conflict-free
//   b128_stride2  lane i reads entry 2i                    — a genuine 2-way conflict
//   b128_same     every lane reads entry 0                 — broadcast
//   hipcc --offload-arch=gfx950 -O2 tools/lds_b128_probe.hip -o /tmp/lds_probe
//   rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT -d out --output-format csv -- /tmp/lds_probe
#include <hip/hip_runtime.h>
#include <stdio.h>

template <int BYTES, int STRIDE, bool SAME>
__global__ __launch_bounds__(256) void probe(float* out, int iters) {
  __shared__ __attribute__((aligned(16))) unsigned lds[16384];          // 64 KB
  for (int i = threadIdx.x; i < 16384; i += 256) lds[i] = i;
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  unsigned acc = 0;
  for (int it = 0; it < iters; ++it) {
    const int base = ((it * 4 + wave) * 64 * (BYTES / 4) * STRIDE) & 8191;   // dword index, stays inside the array
    const int idx = base + (SAME ? 0 : lane * (BYTES / 4) * STRIDE);
    if constexpr (BYTES == 16) { const uint4 v = *reinterpret_cast<const uint4*>(lds + (idx & ~3)); acc += v.x ^ v.y ^ v.z ^ v.w; }
    else if constexpr (BYTES == 8) { const uint2 v = *reinterpret_cast<const uint2*>(lds + (idx & ~1)); acc += v.x ^ v.y; }
    else acc += lds[idx];
  }
  if (acc == 0x12345678u) out[threadIdx.x] = (float)acc;
}
template <int BYTES, int STRIDE, bool SAME> __global__ void dummy() {}

#define RUN(name, B, S, SM) hipLaunchKernelGGL((probe<B, S, SM>), dim3(1024), dim3(256), 0, 0, out, 4096); if (hipDeviceSynchronize() != hipSuccess) { printf(name " failed\n"); return 1; } printf(name " ok\n");
int main() {
  float* out;
  if (hipMalloc(&out, 4096) != hipSuccess) return 1;
  for (int rep = 0; rep < 3; ++rep) {
    RUN("b128_linear", 16, 1, false)
    RUN("b64_linear", 8, 1, false)
    RUN("b32_linear", 4, 1, false)
    RUN("b128_stride2", 16, 2, false)
    RUN("b128_same", 16, 1, true)
  }
  return 0;
}
