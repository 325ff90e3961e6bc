// Ablation micro-benchmark of the corr81 forward kernel (standalone, no torch).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I upflow_pytorch_amd/csrc tools/corr_ablate.hip upflow_pytorch_amd/csrc/api.hip -o /tmp/corr_ablate
// Times template<ABL> variants of the product kernel with hipEvents around each launch.
#include "corr81_fwd_kernel.hpp"
#include "corr81_mfma_kernel.hpp"
#include <hip/hip_ext.h>
#include <vector>
#include <cstdlib>
#include <algorithm>

using namespace upf;
using namespace upf::corr;

template <typename T, int KC, int ABL>
float run(const T* f1, const T* f2, T* out, int B, int C, int H, int W, int nrep) {
  const int tiles_x = cdiv(W, TW), tiles_y = cdiv(H, TH);
  const int nblocks = B * tiles_x * tiles_y;
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&corr81_fwd_kernel<T, true, KC, ABL>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes(KC));
  std::vector<hipEvent_t> ev(2 * nrep);
  for (auto& e : ev) (void)hipEventCreate(&e);
  for (int i = 0; i < nrep; ++i)
    hipExtLaunchKernelGGL((corr81_fwd_kernel<T, true, KC, ABL>), dim3(nblocks), dim3(NTHREADS), lds_bytes(KC), 0, ev[2 * i], ev[2 * i + 1], 0,
                          f1, f2, out, C, H, W, tiles_x, tiles_y, (long long)81 * H * W, 0.1f);
  (void)hipDeviceSynchronize();
  std::vector<float> t(nrep);
  for (int i = 0; i < nrep; ++i) (void)hipEventElapsedTime(&t[i], ev[2 * i], ev[2 * i + 1]);
  for (auto& e : ev) (void)hipEventDestroy(e);
  std::sort(t.begin(), t.end());
  return t[nrep / 2] * 1e3f;   // median, us
}

template <typename T, bool SINGLE, int ABL>
float run_m(const T* f1, const T* f2, T* out, int B, int C, int H, int W, int nrep) {
  const int tiles_x = cdiv(W, corrm::TW), tiles_y = cdiv(H, corrm::TH);
  const int nblocks = B * tiles_x * tiles_y;
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&corrm::corr81_mfma_kernel<T, SINGLE, ABL>), hipFuncAttributeMaxDynamicSharedMemorySize, corrm::LDS_BYTES);
  std::vector<hipEvent_t> ev(2 * nrep);
  for (auto& e : ev) (void)hipEventCreate(&e);
  for (int i = 0; i < nrep; ++i)
    hipExtLaunchKernelGGL((corrm::corr81_mfma_kernel<T, SINGLE, ABL>), dim3(nblocks), dim3(corrm::NTHREADS), corrm::LDS_BYTES, 0, ev[2 * i], ev[2 * i + 1], 0,
                          f1, f2, out, C, H, W, tiles_x, tiles_y, (long long)81 * H * W, 0.1f);
  (void)hipDeviceSynchronize();
  std::vector<float> t(nrep);
  for (int i = 0; i < nrep; ++i) (void)hipEventElapsedTime(&t[i], ev[2 * i], ev[2 * i + 1]);
  for (auto& e : ev) (void)hipEventDestroy(e);
  std::sort(t.begin(), t.end());
  return t[nrep / 2] * 1e3f;
}

template <typename T, bool SINGLE>
void sweep_m(const char* name, int B, int C, int H, int W) {
  size_t n_in = (size_t)B * C * H * W, n_out = (size_t)B * 81 * H * W;
  T *f1, *f2, *out;
  (void)hipMalloc(&f1, n_in * sizeof(T)); (void)hipMalloc(&f2, n_in * sizeof(T)); (void)hipMalloc(&out, n_out * sizeof(T));
  std::vector<uint16_t> h(n_in);
  for (auto& v : h) v = (uint16_t)(0x3c00 + (rand() & 0x3ff));
  (void)hipMemcpy(f1, h.data(), n_in * sizeof(T), hipMemcpyHostToDevice);
  (void)hipMemcpy(f2, h.data(), n_in * sizeof(T), hipMemcpyHostToDevice);
  run_m<T, SINGLE, 0>(f1, f2, out, B, C, H, W, 20);
  const double bytes = (double)sizeof(T) * B * H * W * (2 * C + 81);
  float t0 = run_m<T, SINGLE, 0>(f1, f2, out, B, C, H, W, 100);
  float t1 = run_m<T, SINGLE, 1>(f1, f2, out, B, C, H, W, 100);
  float t2 = run_m<T, SINGLE, 2>(f1, f2, out, B, C, H, W, 100);
  float t4 = run_m<T, SINGLE, 4>(f1, f2, out, B, C, H, W, 100);
  float t6 = run_m<T, SINGLE, 6>(f1, f2, out, B, C, H, W, 100);
  float t5 = run_m<T, SINGLE, 5>(f1, f2, out, B, C, H, W, 100);
  printf("MFMA %-5s B%d C%3d %4dx%-4d full %6.2f us (%5.1f%% of 8TB/s) | -stage %6.2f | -mma %6.2f | -store %6.2f | only-stage %6.2f | only-mma+epi %6.2f\n",
         name, B, C, H, W, t0, bytes / t0 / 1e3 / 80.0, t1, t2, t4, t6, t5);
  (void)hipFree(f1); (void)hipFree(f2); (void)hipFree(out);
}

template <typename T, int KC>
void sweep(const char* name, int B, int C, int H, int W) {
  size_t n_in = (size_t)B * C * H * W, n_out = (size_t)B * 81 * H * W;
  T *f1, *f2, *out;
  (void)hipMalloc(&f1, n_in * sizeof(T)); (void)hipMalloc(&f2, n_in * sizeof(T)); (void)hipMalloc(&out, n_out * sizeof(T));
  std::vector<uint16_t> h(n_in * sizeof(T) / 2);
  for (auto& v : h) v = (uint16_t)(0x3c00 + (rand() & 0x3ff));   // benign finite bit patterns for f16/bf16/f32 halves
  (void)hipMemcpy(f1, h.data(), n_in * sizeof(T), hipMemcpyHostToDevice);
  (void)hipMemcpy(f2, h.data(), n_in * sizeof(T), hipMemcpyHostToDevice);
  run<T, KC, 0>(f1, f2, out, B, C, H, W, 20);
  const double bytes = (double)sizeof(T) * B * H * W * (2 * C + 81);
  float t0 = run<T, KC, 0>(f1, f2, out, B, C, H, W, 100);
  float t1 = run<T, KC, 1>(f1, f2, out, B, C, H, W, 100);
  float t2 = run<T, KC, 2>(f1, f2, out, B, C, H, W, 100);
  float t4 = run<T, KC, 4>(f1, f2, out, B, C, H, W, 100);
  float t3 = run<T, KC, 3>(f1, f2, out, B, C, H, W, 100);
  float t6 = run<T, KC, 6>(f1, f2, out, B, C, H, W, 100);
  float t5 = run<T, KC, 5>(f1, f2, out, B, C, H, W, 100);
  float t7 = run<T, KC, 7>(f1, f2, out, B, C, H, W, 100);
  printf("KC%d %-5s B%d C%3d %4dx%-4d full %6.2f us (%5.1f%% of 8TB/s) | -stage %6.2f | -mac %6.2f | -store %6.2f | only-store %6.2f | only-stage %6.2f | only-mac %6.2f | empty %6.2f\n",
         KC, name, B, C, H, W, t0, bytes / t0 / 1e3 / 80.0, t1, t2, t4, t3, t6, t5, t7);
  (void)hipFree(f1); (void)hipFree(f2); (void)hipFree(out);
}

__global__ void empty_kernel(int* p) { if (p && threadIdx.x == 9999) *p = 1; }
template <int LDS>
__global__ void empty_lds_kernel(int* p) { extern __shared__ int sm[]; if (p && threadIdx.x == 9999) *p = sm[0]; }

static void floor_probe() {
  struct Cfg { int blocks, threads, lds; } cfgs[] = {{1, 64, 0}, {256, 256, 0}, {480, 576, 0}, {480, 576, 61440}, {960, 576, 30720}, {240, 576, 61440},
                                                     {2048, 256, 0}, {480, 256, 61440}, {4320, 64, 0}, {1024, 1024, 0}};
  for (auto c : cfgs) {
    const int nrep = 50;
    std::vector<hipEvent_t> ev(2 * nrep);
    for (auto& e : ev) (void)hipEventCreate(&e);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&empty_lds_kernel<0>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    for (int i = 0; i < nrep; ++i)
      hipExtLaunchKernelGGL(empty_lds_kernel<0>, dim3(c.blocks), dim3(c.threads), c.lds, 0, ev[2 * i], ev[2 * i + 1], 0, (int*)nullptr);
    (void)hipDeviceSynchronize();
    std::vector<float> t(nrep);
    for (int i = 0; i < nrep; ++i) (void)hipEventElapsedTime(&t[i], ev[2 * i], ev[2 * i + 1]);
    std::sort(t.begin(), t.end());
    printf("empty launch %5d blocks x %4d threads, %5d B LDS: %6.2f us\n", c.blocks, c.threads, c.lds, t[nrep / 2] * 1e3f);
    for (auto& e : ev) (void)hipEventDestroy(e);
  }
}

int main() {
  sweep_m<bf16_t, true>("bf16", 4, 32, 96, 320);
  sweep_m<f16_t, true>("f16", 4, 32, 96, 320);
  sweep_m<bf16_t, true>("bf16", 1, 32, 240, 720);
  sweep_m<bf16_t, true>("bf16", 8, 32, 112, 256);
  sweep_m<bf16_t, false>("bf16", 4, 64, 48, 160);
  sweep_m<bf16_t, false>("bf16", 4, 96, 24, 80);
  sweep_m<bf16_t, false>("bf16", 4, 128, 12, 40);
  sweep<bf16_t, 4>("bf16", 4, 32, 96, 320);
  sweep<bf16_t, 8>("bf16", 4, 32, 96, 320);
  sweep<float, 4>("f32", 4, 32, 96, 320);
  sweep<float, 8>("f32", 4, 32, 96, 320);
  sweep<bf16_t, 4>("bf16", 1, 32, 240, 720);
  sweep<bf16_t, 8>("bf16", 1, 32, 240, 720);
  sweep<bf16_t, 4>("bf16", 4, 64, 48, 160);
  sweep<bf16_t, 8>("bf16", 4, 64, 48, 160);
  sweep<bf16_t, 4>("bf16", 4, 196, 6, 20);
  sweep<bf16_t, 8>("bf16", 4, 196, 6, 20);
  sweep<bf16_t, 8>("bf16", 8, 32, 112, 256);
  return 0;
}
