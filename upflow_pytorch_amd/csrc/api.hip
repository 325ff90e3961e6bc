// Library identity and error plumbing of libupflow_hip.so.
#include "common.hpp"

namespace upf {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

int check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    set_error("%s: HIP launch failed: %s", what, hipGetErrorString(e));
    return (int)e > 0 ? (int)e : 1;
  }
  return UPF_OK;
}

__global__ void zero_fill_u64_kernel(unsigned long long* __restrict__ p, long long n) {
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) p[i] = 0ull;
}

int zero_fill_u64(void* p, long long n, hipStream_t s) {
  if (n <= 0) return UPF_OK;
  const long long blocks = (n + 255) / 256;
  hipLaunchKernelGGL(zero_fill_u64_kernel, dim3((unsigned)(blocks > 8192 ? 8192 : blocks)), dim3(256), 0, s, (unsigned long long*)p, n);
  return check_launch("zero_fill");
}

}  // namespace upf

extern "C" const char* upf_version(void) { return "upflow_hip 0.1.0 gfx950"; }
extern "C" const char* upf_last_error(void) { return upf::g_err; }
