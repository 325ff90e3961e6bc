// Does raw_buffer_load_b128 at a 2-byte-aligned byte offset return the right bytes on gfx950, and how does the
// descriptor's bounds check treat a 16-byte load that straddles num_records?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
__global__ void k(const uint16_t* p, u32x4* o, int n, int nrec_bytes) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(p), 0, (uint32_t)nrec_bytes, 0x00020000);
  o[i] = __builtin_amdgcn_raw_buffer_load_b128(r, (uint32_t)i * 2u, 0, 0);
}
int main() {
  const int n = 4096, nrec = 2 * 4000 + 6;   // records end mid-way
  std::vector<uint16_t> h(n + 16);
  for (size_t i = 0; i < h.size(); ++i) h[i] = (uint16_t)(i * 7 + 3);
  uint16_t* d; u32x4* o;
  hipMalloc(&d, h.size() * 2); hipMalloc(&o, n * 16);
  hipMemcpy(d, h.data(), h.size() * 2, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, d, o, n, nrec);
  std::vector<uint16_t> r(n * 8);
  hipMemcpy(r.data(), o, n * 16, hipMemcpyDeviceToHost);
  int bad_in = 0, first_bad = -1;
  for (int i = 0; i < n; ++i) {
    bool fully_in = (i * 2 + 16 <= nrec);
    if (!fully_in) continue;
    for (int e = 0; e < 8; ++e) if (r[i * 8 + e] != h[i + e]) { ++bad_in; if (first_bad < 0) first_bad = i; break; }
  }
  printf("b128 buffer loads at 2B alignment, fully in range: %d mismatching lanes (first %d)\n", bad_in, first_bad);
  for (int i = 3996; i < 4008; ++i) {
    printf("lane %d (bytes %d..%d, nrec %d):", i, 2 * i, 2 * i + 15, nrec);
    for (int e = 0; e < 8; ++e) printf(" %s", r[i * 8 + e] == h[i + e] ? "ok" : (r[i * 8 + e] == 0 ? "0" : "??"));
    printf("\n");
  }
  return 0;
}
