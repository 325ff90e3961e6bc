// How much HBM bandwidth does the conv kernel's x-staging ACCESS PATTERN get by itself?  A [N, C, H, W] bf16 tensor is read
// once (every byte exactly once per tile incl. a halo, like the kernel) and reduced to one dword per thread:
//   mode 0: linear streaming (each thread 16 B, consecutive threads consecutive addresses)
//   mode 1: the kernel's pattern — workgroup = TH x TW pixel tile (+halo HY rows, 8 px each side) of CH channels per step,
//           thread task = (channel octet, row, 8-pixel group) -> 8 loads of 16 B from 8 channel planes, steps over all channels
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
__global__ __launch_bounds__(256) void k_linear(const uint4* __restrict__ p, unsigned* out, size_t n16) {
  size_t i = blockIdx.x * (size_t)256 + threadIdx.x;
  unsigned acc = 0;
  for (; i < n16; i += (size_t)gridDim.x * 256) { uint4 v = p[i]; acc ^= v.x ^ v.y ^ v.z ^ v.w; }
  out[blockIdx.x * 256 + threadIdx.x] = acc;
}
template <int TH, int TW, int NOCTS>
__global__ __launch_bounds__(256, 2) void k_tile(const uint16_t* __restrict__ x, unsigned* out, int C, int H, int W, int tiles_x, int tiles_y) {
  int bid = blockIdx.x;
  { const int NX = 8, nb = gridDim.x; int xcd = bid % NX, idx = bid / NX, q = nb / NX, r = nb % NX; bid = ((xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx; }   // the kernel's xcd_remap
  const int tx = bid % tiles_x, ty = (bid / tiles_x) % tiles_y, n = bid / (tiles_x * tiles_y);
  const int x0 = tx * TW, y0 = ty * TH;
  constexpr int rows = TH + 2, XW = TW + 16, ngroups = XW / 8, ntasks = NOCTS * rows * ngroups;
  const uint32_t plane = (uint32_t)H * W * 2u;
  __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(x + (size_t)n * C * H * W), 0, (uint32_t)C * plane, 0x00020000);
  unsigned acc = 0;
  const int nchunks = C / (NOCTS * 8);
  for (int cc = 0; cc < nchunks; ++cc) {
    for (int t = threadIdx.x; t < ntasks; t += 256) {
      const int oct = t / (rows * ngroups), rem = t - oct * (rows * ngroups), rr = rem / ngroups, g = rem - rr * ngroups;
      const int gy = y0 - 1 + rr, gx = x0 - 8 + 8 * g;
      const bool in = gy >= 0 && gy < H && gx >= 0 && gx < W;
      const uint32_t off = in ? ((uint32_t)((cc * NOCTS * 8 + oct * 8) * H * W + gy * W + gx) * 2u) : 0x80000000u;
#pragma unroll
      for (int k = 0; k < 8; ++k) { u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, off + k * plane, 0, 0); acc ^= v[0] ^ v[1] ^ v[2] ^ v[3]; }
    }
  }
  out[blockIdx.x * 256 + threadIdx.x] = acc;
}
template <typename F> float timeit(F f) {
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (int i = 0; i < 3; ++i) f();
  (void)hipEventRecord(e0);
  for (int i = 0; i < 20; ++i) f();
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1); return ms / 20 * 1e3f;
}
int main() {
  const int N = 8, C = 512, H = 96, W = 384;   // (384 = 3 x 128: no partial tiles for any width tried except 320)
  const size_t elems = (size_t)N * C * H * W;
  uint16_t* x; unsigned* out; (void)hipMalloc(&x, elems * 2); (void)hipMalloc(&out, 1 << 24);
  (void)hipMemset(x, 1, elems * 2);
  const double mb = elems * 2 / 1e6;
  float t = timeit([&] { hipLaunchKernelGGL(k_linear, dim3(256 * 8), dim3(256), 0, 0, (const uint4*)x, out, elems / 8); });
  printf("linear stream            : %7.1f us  %.2f TB/s\n", t, mb / t);
#define RUN(TH, TW, NO)                                                                                                   \
  {                                                                                                                       \
    const int tx = (W + TW - 1) / TW, ty = (H + TH - 1) / TH;                                                               \
    t = timeit([&] { hipLaunchKernelGGL((k_tile<TH, TW, NO>), dim3(N * tx * ty), dim3(256), 0, 0, x, out, C, H, W, tx, ty); }); \
    printf("tile %2dx%-3d chunk %2d ch  : %7.1f us  %.2f TB/s (algorithmic bytes; halo overfetch %.2fx through L2)\n", TH, TW, NO * 8, t, mb / t, \
           (double)(TH + 2) * (TW + 16) / (TH * TW));                                                                      \
  }
  RUN(8, 32, 4) RUN(16, 32, 4) RUN(8, 64, 4) RUN(16, 64, 2) RUN(8, 128, 2) RUN(4, 128, 4) RUN(16, 128, 1) RUN(8, 320, 1) RUN(2, 320, 4)
  return 0;
}
