// Where do the ~4.5 us go that the normalising cost volume (corr81_allc_kernel<..., NORM>) takes over the plain one?
// Standalone (no torch): the product kernel compiled with -DUPF_ALLC_ABL=<bits> (see corr81_allc_kernel.hpp), timed with
// HIP events around each of 100 back-to-back launches at the 1/4-resolution level of config 2, [8,32,96,320] bf16.
//   for a in 0 1 2 4 8 15; do hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -DUPF_ALLC_ABL=$a -I upflow_pytorch_amd/csrc -I include \
//       tools/corr_norm_ablate.hip upflow_pytorch_amd/csrc/api.hip -o /tmp/cna_$a; done
#include "corr81_allc_kernel.hpp"
#include <hip/hip_ext.h>
#include <vector>
#include <cstdio>
#include <cstdlib>
#include <algorithm>
using namespace upf;

template <bool NORM>
float run(const bf16_t* f1, const bf16_t* f2, bf16_t* out, const float* ws, int B, int C, int H, int W, int nseg, int nrep) {
  using G = corrx::Geo<32, 4>;
  const int tiles_x = cdiv(W, G::TW), tiles_y = cdiv(H, G::TH), nblocks = B * tiles_x * tiles_y;
  const size_t lds = corrx::lds_bytes<32, 4>((C + 3) / 4, false, NORM);
  auto kern = &corrx::corr81_allc_kernel<bf16_t, 32, 4, 4, false, NORM, 1>;
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  std::vector<hipEvent_t> ev(2 * nrep);
  for (auto& e : ev) (void)hipEventCreate(&e);
  for (int i = 0; i < nrep; ++i)
    hipExtLaunchKernelGGL(kern, dim3(nblocks), dim3(corrx::NTHREADS), lds, 0, ev[2 * i], ev[2 * i + 1], 0,
                          f1, f2, out, C, H, W, tiles_x, tiles_y, (long long)81 * H * W, 0.1f, ws, ws + (size_t)B * C * 2, nseg, nblocks, W);   // (round 4: final (mean, 1/std) pairs)
  (void)hipDeviceSynchronize();
  std::vector<float> t(nrep);
  for (int i = 0; i < nrep; ++i) (void)hipEventElapsedTime(&t[i], ev[2 * i], ev[2 * i + 1]);
  for (auto& e : ev) (void)hipEventDestroy(e);
  std::sort(t.begin(), t.end());
  return t[nrep / 2] * 1e3f;
}

int main() {
  const int B = 8, C = 32, H = 96, W = 320, nseg = 2;
  const size_t n_in = (size_t)B * C * H * W, n_out = (size_t)B * 81 * H * W;
  bf16_t *f1, *f2, *out; float* ws;
  (void)hipMalloc(&f1, n_in * 2); (void)hipMalloc(&f2, n_in * 2); (void)hipMalloc(&out, n_out * 2);
  (void)hipMalloc(&ws, (size_t)2 * B * C * 2 * 4);
  std::vector<uint16_t> h(n_in);
  for (auto& v : h) v = (uint16_t)(0x3f00 + (rand() & 0xff));
  (void)hipMemcpy(f1, h.data(), n_in * 2, hipMemcpyHostToDevice);
  for (auto& v : h) v = (uint16_t)(0x3f00 + (rand() & 0xff));
  (void)hipMemcpy(f2, h.data(), n_in * 2, hipMemcpyHostToDevice);
  std::vector<float> hw((size_t)2 * B * C * 2);
  for (size_t i = 0; i < hw.size(); i += 2) { hw[i] = 0.6f + 0.001f * (i % 7); hw[i + 1] = 1.0f / (1.0f + 0.01f * (i % 11)); }
  (void)hipMemcpy(ws, hw.data(), hw.size() * 4, hipMemcpyHostToDevice);
  run<false>(f1, f2, out, ws, B, C, H, W, nseg, 20); run<true>(f1, f2, out, ws, B, C, H, W, nseg, 20);
  for (int rep = 0; rep < 3; ++rep) {
    const float a = run<false>(f1, f2, out, ws, B, C, H, W, nseg, 100), b = run<true>(f1, f2, out, ws, B, C, H, W, nseg, 100);
    printf("ABL=%d  plain %.2f us   NORM %.2f us   (+%.2f)\n", UPF_ALLC_ABL, a, b, b - a);
  }
  return 0;
}
