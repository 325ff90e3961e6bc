// Feature normalisation and occlusion check — gfx950.
//
// normalize: network_tools.normalize_features with the inference flags of test.py:22-30
// (/root/reference/model/upflow.py:94-137): per sample, per channel mean and UNBIASED variance over
// H*W, y = (x - mean) / sqrt(var + 1e-16).  The reference issues ~8 ATen launches per tensor (mean,
// var, add, sqrt, sub, div ...); here one workgroup owns one (n,c) row: the row is read once from
// HBM (it stays in L2 for the second and third sweep) and y is written once.
//
// occ_check: tools.occ_check_model(obj) (utils/tools.py:519-588, 641-677): two unmasked warps of the
// opposite flow, |.|_1 magnitudes, threshold, outgoing-flow mask — one launch.
// Compiled with -ffp-contract=off.
#include "sampling.hpp"

namespace upf {
namespace misc {

constexpr int NT = 512;

__device__ __forceinline__ float block_sum(float v, float* sh) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  const int wid = threadIdx.x >> 6, lane = threadIdx.x & 63;
  __syncthreads();
  if (lane == 0) sh[wid] = v;
  __syncthreads();
  float r = 0.f;
#pragma unroll
  for (int k = 0; k < NT / 64; ++k) r += sh[k];
  return r;
}

template <typename T>
__global__ __launch_bounds__(NT)
void normalize_fwd_kernel(const T* __restrict__ x, T* __restrict__ y, float* __restrict__ mean_out,
                          float* __restrict__ rstd_out, int HW) {
  __shared__ float sh[NT / 64];
  const size_t row = blockIdx.x;
  const T* xr = x + row * HW;
  T* yr = y + row * HW;
  float s = 0.f;
  for (int i = threadIdx.x; i < HW; i += NT) s += Elem<T>::load(xr + i);
  const float mean = block_sum(s, sh) / (float)HW;
  float ss = 0.f;
  for (int i = threadIdx.x; i < HW; i += NT) { const float d = Elem<T>::load(xr + i) - mean; ss += d * d; }
  const float var = block_sum(ss, sh) / (float)(HW - 1);          // unbiased, torch.var default (upflow.py:114)
  const float std = sqrtf(var + 1e-16f);                           // upflow.py:126
  for (int i = threadIdx.x; i < HW; i += NT) Elem<T>::store(yr + i, (Elem<T>::load(xr + i) - mean) / std);
  if (threadIdx.x == 0) {
    if (mean_out) mean_out[row] = mean;
    if (rstd_out) rstd_out[row] = 1.0f / std;
  }
}

// gx = rstd * (g - mean(g) - y * sum(g*y)/(HW-1))
template <typename T>
__global__ __launch_bounds__(NT)
void normalize_bwd_kernel(const T* __restrict__ y, const T* __restrict__ gy, const float* __restrict__ rstd,
                          T* __restrict__ gx, int HW) {
  __shared__ float sh[NT / 64];
  const size_t row = blockIdx.x;
  const T* yr = y + row * HW;
  const T* gr = gy + row * HW;
  float sg = 0.f, sgy = 0.f;
  for (int i = threadIdx.x; i < HW; i += NT) { const float g = Elem<T>::load(gr + i); sg += g; sgy += g * Elem<T>::load(yr + i); }
  const float mg = block_sum(sg, sh) / (float)HW;
  const float k = block_sum(sgy, sh) / (float)(HW - 1);
  const float r = rstd[row];
  T* o = gx + row * HW;
  for (int i = threadIdx.x; i < HW; i += NT)
    Elem<T>::store(o + i, r * (Elem<T>::load(gr + i) - mg - Elem<T>::load(yr + i) * k));
}

__device__ __forceinline__ void warp2(const float* __restrict__ f, int HW, const Taps& t, int H, int W, float& a, float& b) {
  const int xa = min(max(t.x0, 0), W - 1), xb = min(max(t.x0 + 1, 0), W - 1);
  const int ya = min(max(t.y0, 0), H - 1), yb = min(max(t.y0 + 1, 0), H - 1);
  const int o0 = ya * W + xa, o1 = ya * W + xb, o2 = yb * W + xa, o3 = yb * W + xb;
  const float w0 = t.in[0] ? t.w[0] : 0.f, w1 = t.in[1] ? t.w[1] : 0.f;
  const float w2 = t.in[2] ? t.w[2] : 0.f, w3 = t.in[3] ? t.w[3] : 0.f;
  a = ((f[o0] * w0 + f[o1] * w1) + f[o2] * w2) + f[o3] * w3;
  b = ((f[HW + o0] * w0 + f[HW + o1] * w1) + f[HW + o2] * w2) + f[HW + o3] * w3;
}

__global__ __launch_bounds__(256)
void occ_check_kernel(const float* __restrict__ ff, const float* __restrict__ fb, float* __restrict__ occ_fw,
                      float* __restrict__ occ_bw, int H, int W, float a1, float a2) {
  const int HW = H * W;
  const int p = blockIdx.x * 256 + threadIdx.x;
  if (p >= HW) return;
  const int n = blockIdx.y;
  const int i = p / W, j = p - i * W;
  const float* F = ff + (size_t)n * 2 * HW;
  const float* Bk = fb + (size_t)n * 2 * HW;
  const float fx = F[p], fy = F[HW + p], bx = Bk[p], by = Bk[HW + p];
  const float mag = (fabsf(fx) + fabsf(fy)) + (fabsf(bx) + fabsf(by));      // tools.py:559, :573
  const float thr = a1 * mag + a2;                                          // :578
  float wx, wy;
  warp2(Bk, HW, make_taps(j, i, fx, fy, H, W), H, W, wx, wy);               // flow_bw warped by flow_fw, :574
  const bool cf = (fabsf(fx + wx) + fabsf(fy + wy)) < thr;                  // :576, :579
  warp2(F, HW, make_taps(j, i, bx, by, H, W), H, W, wx, wy);                // :575
  const bool cb = (fabsf(bx + wx) + fabsf(by + wy)) < thr;
  // outgoing mask (tools.py:657-667) and obj merge (:672-676): 1 where consistent OR flow leaves the image
  const float pxf = (float)j + fx, pyf = (float)i + fy, pxb = (float)j + bx, pyb = (float)i + by;
  const bool inf_ = !(pxf > (float)(W - 1)) && !(pxf < 0.f) && !(pyf > (float)(H - 1)) && !(pyf < 0.f);
  const bool inb_ = !(pxb > (float)(W - 1)) && !(pxb < 0.f) && !(pyb > (float)(H - 1)) && !(pyb < 0.f);
  occ_fw[(size_t)n * HW + p] = (cf || !inf_) ? 1.f : 0.f;
  occ_bw[(size_t)n * HW + p] = (cb || !inb_) ? 1.f : 0.f;
}

}  // namespace misc
}  // namespace upf

extern "C" int upf_normalize_forward(const void* x, void* y, float* mean, float* rstd, long long N, int HW,
                                     int dtype, void* stream) {
  using namespace upf;
  UPF_REQUIRE(x && y, UPF_EINVAL, "normalize_forward: null pointer");
  UPF_REQUIRE(N > 0 && N < (1ll << 31) && HW > 0, UPF_EINVAL, "normalize_forward: bad shape N=%lld HW=%d", N, HW);
  UPF_DISPATCH(dtype, T,
               hipLaunchKernelGGL((misc::normalize_fwd_kernel<T>), dim3((unsigned)N), dim3(misc::NT), 0, (hipStream_t)stream,
                                  (const T*)x, (T*)y, mean, rstd, HW));
  return check_launch("normalize_forward");
}

extern "C" int upf_normalize_backward(const void* y, const void* grad_y, const float* rstd, void* gx, long long N, int HW,
                                      int dtype, void* stream) {
  using namespace upf;
  UPF_REQUIRE(y && grad_y && rstd && gx, UPF_EINVAL, "normalize_backward: null pointer");
  UPF_REQUIRE(N > 0 && N < (1ll << 31) && HW > 0, UPF_EINVAL, "normalize_backward: bad shape");
  UPF_DISPATCH(dtype, T,
               hipLaunchKernelGGL((misc::normalize_bwd_kernel<T>), dim3((unsigned)N), dim3(misc::NT), 0, (hipStream_t)stream,
                                  (const T*)y, (const T*)grad_y, rstd, (T*)gx, HW));
  return check_launch("normalize_backward");
}

extern "C" int upf_occ_check(const float* flow_f, const float* flow_b, float* occ_fw, float* occ_bw, int B, int H, int W,
                             float alpha1, float alpha2, void* stream) {
  using namespace upf;
  UPF_REQUIRE(flow_f && flow_b && occ_fw && occ_bw, UPF_EINVAL, "occ_check: null pointer");
  UPF_REQUIRE(B > 0 && B <= 65535 && H > 0 && W > 0, UPF_EINVAL, "occ_check: bad shape");
  dim3 grid(cdiv(H * W, 256), B);
  hipLaunchKernelGGL(misc::occ_check_kernel, grid, dim3(256), 0, (hipStream_t)stream, flow_f, flow_b, occ_fw, occ_bw, H, W, alpha1, alpha2);
  return check_launch("occ_check");
}
