// Does a 4-byte global load at a 2-byte-aligned address return the right bytes on gfx950 (unaligned access mode)?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
__global__ void k(const uint16_t* p, uint32_t* o, uint2* o2, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  o[i] = *reinterpret_cast<const uint32_t*>(p + i);          // address = base + 2*i : odd i is misaligned
  o2[i] = *reinterpret_cast<const uint2*>(reinterpret_cast<const float*>(p) + i);   // 8-byte load at 4-byte alignment
}
int main() {
  const int n = 4096;
  std::vector<uint16_t> h(n * 2 + 8);
  for (size_t i = 0; i < h.size(); ++i) h[i] = (uint16_t)(i * 7 + 3);
  uint16_t* d; uint32_t* o; uint2* o2;
  hipMalloc(&d, h.size() * 2); hipMalloc(&o, n * 4); hipMalloc(&o2, n * 8);
  hipMemcpy(d, h.data(), h.size() * 2, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, d, o, o2, n);
  std::vector<uint32_t> r(n); std::vector<uint2> r2(n);
  hipMemcpy(r.data(), o, n * 4, hipMemcpyDeviceToHost); hipMemcpy(r2.data(), o2, n * 8, hipMemcpyDeviceToHost);
  int bad = 0, bad2 = 0;
  for (int i = 0; i < n; ++i) {
    uint32_t want = (uint32_t)h[i] | ((uint32_t)h[i + 1] << 16);
    if (r[i] != want) ++bad;
  }
  const uint32_t* hw = reinterpret_cast<const uint32_t*>(h.data());
  for (int i = 0; i < n / 2 - 2; ++i) if (r2[i].x != hw[i] || r2[i].y != hw[i + 1]) ++bad2;
  printf("unaligned 4B loads at 2B alignment: %d mismatches of %d; 8B loads at 4B alignment: %d mismatches\n", bad, n, bad2);
  return 0;
}
