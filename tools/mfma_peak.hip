// Practical ceiling of v_mfma_f32_32x32x16_bf16 on this part: every SIMD issues back-to-back MFMAs on NACC independent
// accumulators from registers only (no memory).  Prints TFLOP/s for 1 and 2 waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
template <int NACC>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
  f32x16 acc[NACC];
  for (int i = 0; i < NACC; ++i) for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
  bf16x8 a, b;
  for (int e = 0; e < 8; ++e) { a[e] = (__bf16)(float)(threadIdx.x & 3); b[e] = (__bf16)(float)(threadIdx.x & 1); }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i], 0, 0, 0);
  }
  float s = 0.f;
  for (int i = 0; i < NACC; ++i) for (int e = 0; e < 16; ++e) s += acc[i][e];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int NACC>
void run(int wgs_per_cu) {
  const int iters = 20000, grid = 256 * wgs_per_cu;
  float* out; (void)hipMalloc(&out, (size_t)grid * 256 * 4);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  hipLaunchKernelGGL(k<NACC>, dim3(grid), dim3(256), 0, 0, out, 100);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  hipLaunchKernelGGL(k<NACC>, dim3(grid), dim3(256), 0, 0, out, iters);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  const double flops = (double)grid * 4 * iters * NACC * 2.0 * 32 * 32 * 16;
  printf("NACC %d, %d WG/CU (= %d waves/SIMD): %.3f ms, %.1f TFLOP/s, %.1f cycles/MFMA/SIMD at 2.4 GHz\n", NACC, wgs_per_cu, wgs_per_cu, ms,
         flops / ms / 1e9, ms * 1e-3 * 2.4e9 / ((double)iters * NACC * wgs_per_cu));
  (void)hipFree(out);
}
int main() { run<4>(1); run<4>(2); run<8>(1); run<8>(2); run<1>(1); run<1>(2); run<2>(2); return 0; }
