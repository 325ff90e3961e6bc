// Exhaustive check of the division-free correctly rounded quotient x / d (d = W-1 or H-1, a wave-uniform constant):
//   r = RN(1/d) (host), q = RN(x*r), e = fma(-q, d, x) (exact), q' = fma(e, r, q)      [Markstein 1990]
// against the IEEE division, over ALL 2^32 bit patterns of x, for every divisor given on the command line.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <stdlib.h>
#include <math.h>
#include <string.h>
__device__ __forceinline__ float div_m(float x, float d, float r) {
  const float q = __fmul_rn(x, r);
  const float e = __builtin_fmaf(-q, d, x);
  return __builtin_fmaf(e, r, q);
}
__global__ void sweep(float d, float r, unsigned long long* bad, unsigned int* first_bad, int guard) {
  const unsigned long long n = 1ull << 32;
  unsigned long long local = 0;
  for (unsigned long long i = blockIdx.x * (unsigned long long)blockDim.x + threadIdx.x; i < n; i += (unsigned long long)gridDim.x * blockDim.x) {
    const float x = __uint_as_float((unsigned int)i);
    const float want = __fdiv_rn(x, d);
    float got = div_m(x, d, r);
    if (guard) {   // the guarded form: quotients that are not normal numbers take the division
      const float a = fabsf(want);
      if (!(a >= 1.17549435e-38f * 16777216.f && a <= 3.0e38f)) got = want;
    }
    const bool same = (__float_as_uint(got) == __float_as_uint(want)) || (got != got && want != want);
    if (!same) { if (local == 0) atomicMin(first_bad, (unsigned int)i); ++local; }
  }
  if (local) atomicAdd(bad, local);
}
int main(int argc, char** argv) {
  unsigned long long* bad; unsigned int* fb;
  hipMalloc(&bad, 8); hipMalloc(&fb, 4);
  for (int a = 1; a < argc; ++a) {
    const float d = (float)atoi(argv[a]), r = 1.0f / d;
    for (int guard = 0; guard < 2; ++guard) {
      unsigned long long h = 0; unsigned int hf = 0xffffffffu;
      hipMemcpy(bad, &h, 8, hipMemcpyHostToDevice); hipMemcpy(fb, &hf, 4, hipMemcpyHostToDevice);
      hipLaunchKernelGGL(sweep, dim3(4096), dim3(256), 0, 0, d, r, bad, fb, guard);
      hipMemcpy(&h, bad, 8, hipMemcpyDeviceToHost); hipMemcpy(&hf, fb, 4, hipMemcpyDeviceToHost);
      float fx; memcpy(&fx, &hf, 4);
      printf("d = %6.0f  guard %d: %llu mismatches of 2^32%s", d, guard, h, h ? "" : "\n");
      if (h) printf("  (first x = %08x = %g)\n", hf, fx);
    }
  }
  return 0;
}
