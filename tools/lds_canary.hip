// Does a kernel of libupflow_hip.so write into ANOTHER workgroup's LDS (or VGPR-held data) when it runs concurrently?
// Canary workgroups (576 threads, 68 KB of LDS like corr81_allc_kernel) fill their LDS with an address pattern, wait ~40 us,
// and verify it; the candidate kernel runs on a second stream meanwhile.
//   hipcc --offload-arch=gfx950 -O2 tools/lds_canary.hip -o /tmp/lds_canary -Lupflow_pytorch_amd -lupflow_hip -Wl,-rpath,$PWD/upflow_pytorch_amd && /tmp/lds_canary
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "../include/upflow_hip.h"

__global__ __launch_bounds__(576) void canary(unsigned* log, unsigned* count, int words, long long spin) {
  extern __shared__ unsigned lds[];
  for (int i = threadIdx.x; i < words; i += blockDim.x) lds[i] = 0xC0DE0000u ^ (unsigned)i ^ (blockIdx.x << 20);
  __syncthreads();
  const long long t0 = clock64();
  while (clock64() - t0 < spin) __builtin_amdgcn_s_sleep(8);
  __syncthreads();
  for (int i = threadIdx.x; i < words; i += blockDim.x) {
    const unsigned want = 0xC0DE0000u ^ (unsigned)i ^ (blockIdx.x << 20), got = lds[i];
    if (got != want) {
      const unsigned k = atomicAdd(count, 1u);
      if (k < 64) { log[3 * k] = (unsigned)i; log[3 * k + 1] = got; log[3 * k + 2] = blockIdx.x; }
    }
  }
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

int main(int argc, char** argv) {
  const int dtype = argc > 1 && !strcmp(argv[1], "fp16") ? UPF_F16 : UPF_BF16;
  const int B = 8, H = 96, W = 320;
  hipStream_t sa, sb;
  CK(hipStreamCreate(&sa)); CK(hipStreamCreate(&sb));
  struct Case { int K, Cout, yc8; } cases[] = {{184, 3, 0}, {176, 8, 1}, {160, 16, 1}, {192, 3, 0}, {184, 16, 1}, {160, 3, 0}};
  unsigned *log, *count;
  CK(hipMalloc(&log, 64 * 3 * 4)); CK(hipMalloc(&count, 4));
  const int lds_bytes = (argc > 2 ? atoi(argv[2]) : 68 * 1024);
  CK(hipFuncSetAttribute((const void*)canary, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes));
  for (auto c : cases) {
    const int noct = c.K / 8;
    void *x8, *wp, *y; float* bias;
    const size_t xb = (size_t)B * noct * H * W * 16;
    CK(hipMalloc(&x8, xb)); CK(hipMemset(x8, 0x3f, xb));
    const long long wb = upf_conv_packed_bytes_k16((c.K + 31) / 32 * 32, c.Cout);
    CK(hipMalloc(&wp, wb)); CK(hipMemset(wp, 0x3c, wb));
    CK(hipMalloc(&bias, 64)); CK(hipMemset(bias, 0, 64));
    const size_t yb = (size_t)B * 16 * H * W * 2;
    CK(hipMalloc(&y, yb));
    unsigned total = 0, runs = 0;
    unsigned hlog[64 * 3];
    for (int it = 0; it < 30; ++it) {
      CK(hipMemsetAsync(count, 0, 4, sa));
      CK(hipStreamSynchronize(sa));
      for (int r = 0; r < 12; ++r) {
        int rc = upf_conv_forward_c8_narrow(x8, (long long)noct * H * W * 8, noct, wp, bias, y, c.yc8 ? (long long)((c.Cout + 7) / 8) * H * W * 8 : (long long)c.Cout * H * W, c.yc8, B, c.Cout, H, W, 0.1f, dtype, sb);
        if (rc) { printf("narrow launch failed: %s\n", upf_last_error()); return 1; }
      }
      hipLaunchKernelGGL(canary, dim3(argc > 3 ? atoi(argv[3]) : 1024), dim3(576), lds_bytes, sa, log, count, lds_bytes / 4, (long long)3000);   // ~40 us at 100 MHz clock64
      CK(hipDeviceSynchronize());
      unsigned n;
      CK(hipMemcpy(&n, count, 4, hipMemcpyDeviceToHost));
      if (n && !total) CK(hipMemcpy(hlog, log, sizeof(hlog), hipMemcpyDeviceToHost));
      total += n; runs += n ? 1 : 0;
    }
    printf("%s narrow K=%d Cout=%d %s: corrupted LDS words %u in %u of 30 runs\n", dtype == UPF_F16 ? "fp16" : "bf16", c.K, c.Cout, c.yc8 ? "c8-out" : "nchw-out", total, runs);
    if (total) for (int k = 0; k < 12; ++k) printf("    word %u (byte %u) = %08x  wg %u\n", hlog[3 * k], hlog[3 * k] * 4, hlog[3 * k + 1], hlog[3 * k + 2]);
    hipFree(x8); hipFree(wp); hipFree(bias); hipFree(y);
  }
  return 0;
}
