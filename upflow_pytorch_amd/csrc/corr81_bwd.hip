// 81-neighbour cost volume, backward — gfx950.
//
// Replaces correlation_backward_input1/2<T> (/root/reference/model/correlation_package/
// correlation_cuda_kernel.cu:116-300) and their host loop of 2*B launches over a (H,W,C) grid of
// 32-thread blocks (:488-520), plus the two NHWC staging passes (:461-483).
//
//   g1[n,c,y,x] = (1/C) sum_d gO[n,d,y,x]       * f2[n,c,y+dy,x+dx]
//   g2[n,c,y,x] = (1/C) sum_d gO[n,d,y-dy,x-dx] * f1[n,c,y-dy,x-dx]      (out-of-range terms dropped)
//
// Both are "81 per-pixel weights times a 9x9 neighbourhood of one feature channel".  A thread owns
// one pixel: it loads its 81 weights ONCE into registers (gO is the big tensor: 81 channels), then
// walks its slice of the C channels gathering the neighbourhood (L1/L2 hits: neighbouring lanes read
// neighbouring addresses).  gO is therefore read exactly once per direction from HBM; fp32
// accumulation.  One launch covers the whole batch and both gradients (blockIdx.z selects which).
#include "common.hpp"

namespace upf {
namespace corr {

constexpr int BT = 256;

template <typename T>
__global__ __launch_bounds__(BT)
void corr81_bwd_kernel(const T* __restrict__ f1, const T* __restrict__ f2, const T* __restrict__ gO,
                       T* __restrict__ g1, T* __restrict__ g2, int B, int C, int H, int W, int cpt) {
  const int HW = H * W;
  const int p = blockIdx.x * BT + threadIdx.x;
  if (p >= HW) return;
  const int which = blockIdx.z / B;          // 0: g1 (neighbourhood of f2 at +d), 1: g2 (f1 at -d)
  const int n = blockIdx.z - which * B;
  const int y = p / W, x = p - y * W;
  const int sgn = which ? -1 : 1;
  const T* feat = (which ? f1 : f2) + (size_t)n * C * HW;
  T* gout = (which ? g2 : g1) + (size_t)n * C * HW;
  const T* go = gO + (size_t)n * 81 * HW;

  float wgt[81];
  int off[81];
#pragma unroll
  for (int d = 0; d < 81; ++d) {
    const int dy = d / 9 - 4, dx = d % 9 - 4;
    const int yy = y + sgn * dy, xx = x + sgn * dx;
    const bool in = yy >= 0 && yy < H && xx >= 0 && xx < W;
    const int q = in ? yy * W + xx : p;
    off[d] = q;
    // g1: weight gO[d] at the pixel itself; g2: gO[d] at the displaced source pixel
    wgt[d] = in ? Elem<T>::load(go + (size_t)d * HW + (which ? q : p)) : 0.f;
  }
  const float invC = 1.0f / (float)C;
  const int c0 = blockIdx.y * cpt, c1 = min(C, c0 + cpt);
  for (int c = c0; c < c1; ++c) {
    const T* fc = feat + (size_t)c * HW;
    float acc = 0.f;
#pragma unroll
    for (int d = 0; d < 81; ++d) acc = __builtin_fmaf(wgt[d], Elem<T>::load(fc + off[d]), acc);
    Elem<T>::store(gout + (size_t)c * HW + p, acc * invC);
  }
}

}  // namespace corr
}  // namespace upf

extern "C" int upf_corr81_backward(const void* f1, const void* f2, const void* grad_out, void* g1, void* g2,
                                   int B, int C, int H, int W, int dtype, void* stream) {
  using namespace upf;
  UPF_REQUIRE(f1 && f2 && grad_out && g1 && g2, UPF_EINVAL, "corr81_backward: null pointer");
  UPF_REQUIRE(B > 0 && 2 * B <= 65535 && C > 0 && H > 0 && W > 0, UPF_EINVAL, "corr81_backward: bad shape B=%d C=%d H=%d W=%d", B, C, H, W);
  const int HW = H * W;
  // split channels over blockIdx.y until there are a few thousand waves in flight
  int split = 1;
  while (split < C && (long long)2 * B * cdiv(HW, corr::BT) * split < 4096 && C / (split * 2) >= 2) split *= 2;
  const int cpt = cdiv(C, split);
  dim3 grid(cdiv(HW, corr::BT), cdiv(C, cpt), 2 * B);
  UPF_DISPATCH(dtype, T,
               hipLaunchKernelGGL((corr::corr81_bwd_kernel<T>), grid, dim3(corr::BT), 0, (hipStream_t)stream,
                                  (const T*)f1, (const T*)f2, (const T*)grad_out, (T*)g1, (T*)g2, B, C, H, W, cpt));
  return check_launch("corr81_backward");
}
