// Loss-side operators of the unsupervised training step — gfx950, fp32.
//
//   boundary_warp   tools.boundary_dilated_warp.warp_im (/root/reference/utils/tools.py:351-499): the photometric
//                   loss samples the UN-cropped frame at (pixel + crop offset + flow) with clamp-to-edge bilinear taps
//                   whose weights come from the CLAMPED corner coordinates (:409-412, :458-466).  The reference builds
//                   a grid on the CPU, copies it to the device, flattens / permutes the image and issues four
//                   torch.gather calls per warp; here it is ONE gather launch, and one for the gradient wrt the flow.
//   robust_loss     network_tools.photo_loss_multi_type, 'abs_robust' (model/upflow.py:265-288):
//                   sum over a [B,C,H,W] pair of (|x - y| + eps)^q, optionally weighted by a [B,1,H,W] occlusion mask —
//                   the photometric term and every pyramid-distillation term (:461-487) are this reduction.  The
//                   reference spells it as sub / abs / add / pow / mul / sum: six passes forward and as many backward.
//   smooth_edge1    network_tools.edge_aware_smoothness_order1 (model/upflow.py:197-216): first-order flow differences
//                   weighted by exp(-mean_c |image difference|), both directions, one launch forward and one backward.
// Reductions are deterministic: every workgroup writes its partial sum, the partials are summed in fixed order by the
// caller (a [nblocks, 2] tensor); backward kernels are gathers.  Compiled with -ffp-contract=off.
#include "common.hpp"

namespace upf {
namespace loss {

constexpr int NT = 256;

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

// ---- boundary-dilated warp ------------------------------------------------------------------------------------------
struct BTaps { int o00, o10, o01, o11; float x, y, x0f, x1f, y0f, y1f; };

__device__ __forceinline__ BTaps btaps(int i, int j, float sx, float sy, float fx, float fy, int Hi, int Wi) {
  BTaps t;
  t.x = ((float)j + sx) + fx;                        // grid + start, then + flow (tools.py:366, :497)
  t.y = ((float)i + sy) + fy;
  // floor().int() then clamp (tools.py:404-412); huge / NaN positions are clamped before the conversion (no UB)
  const float fx0 = floorf(t.x), fy0 = floorf(t.y);
  const float cx = fminf(fmaxf(fx0, -2.0f), (float)Wi + 1.0f), cy = fminf(fmaxf(fy0, -2.0f), (float)Hi + 1.0f);
  const int x0r = (fx0 == fx0) ? (int)cx : 0, y0r = (fy0 == fy0) ? (int)cy : 0;
  const int x0 = min(max(x0r, 0), Wi - 1), x1 = min(max(x0r + 1, 0), Wi - 1);
  const int y0 = min(max(y0r, 0), Hi - 1), y1 = min(max(y0r + 1, 0), Hi - 1);
  t.o00 = y0 * Wi + x0; t.o10 = y1 * Wi + x0; t.o01 = y0 * Wi + x1; t.o11 = y1 * Wi + x1;
  t.x0f = (float)x0; t.x1f = (float)x1; t.y0f = (float)y0; t.y1f = (float)y1;
  return t;
}

__global__ __launch_bounds__(NT)
void boundary_warp_fwd_kernel(const float* __restrict__ I, const float* __restrict__ flow, const float* __restrict__ start,
                              float* __restrict__ out, int C, int Hi, int Wi, int h, int w) {
  const int hw = h * w;
  const int p = blockIdx.x * NT + threadIdx.x;
  if (p >= hw) return;
  const int n = blockIdx.y;
  const int i = p / w, j = p - i * w;
  const BTaps t = btaps(i, j, start[2 * n], start[2 * n + 1], flow[((size_t)n * 2) * hw + p], flow[((size_t)n * 2 + 1) * hw + p], Hi, Wi);
  const float wa = (t.x1f - t.x) * (t.y1f - t.y), wb = (t.x1f - t.x) * (t.y - t.y0f);
  const float wc = (t.x - t.x0f) * (t.y1f - t.y), wd = (t.x - t.x0f) * (t.y - t.y0f);
  for (int c = 0; c < C; ++c) {
    const float* im = I + ((size_t)n * C + c) * Hi * Wi;
    out[((size_t)n * C + c) * hw + p] = ((wa * im[t.o00] + wb * im[t.o10]) + wc * im[t.o01]) + wd * im[t.o11];   // tools.py:467
  }
}

// d out / d flow: the floor / clamp have zero derivative, so only the four weights depend on the position
__global__ __launch_bounds__(NT)
void boundary_warp_bwd_kernel(const float* __restrict__ I, const float* __restrict__ flow, const float* __restrict__ start,
                              const float* __restrict__ gout, float* __restrict__ gflow, int C, int Hi, int Wi, int h, int w) {
  const int hw = h * w;
  const int p = blockIdx.x * NT + threadIdx.x;
  if (p >= hw) return;
  const int n = blockIdx.y;
  const int i = p / w, j = p - i * w;
  const BTaps t = btaps(i, j, start[2 * n], start[2 * n + 1], flow[((size_t)n * 2) * hw + p], flow[((size_t)n * 2 + 1) * hw + p], Hi, Wi);
  const float ax = t.x1f - t.x, bx = t.x - t.x0f, ay = t.y1f - t.y, by = t.y - t.y0f;
  float gx = 0.f, gy = 0.f;
  for (int c = 0; c < C; ++c) {
    const float* im = I + ((size_t)n * C + c) * Hi * Wi;
    const float g = gout[((size_t)n * C + c) * hw + p];
    const float a = im[t.o00], b = im[t.o10], cc = im[t.o01], d = im[t.o11];
    gx += g * (((-ay) * a + (-by) * b) + (ay * cc + by * d));
    gy += g * (((-ax) * a + ax * b) + ((-bx) * cc + bx * d));
  }
  gflow[((size_t)n * 2) * hw + p] = gx;
  gflow[((size_t)n * 2 + 1) * hw + p] = gy;
}

// ---- robust (abs_robust) loss sum --------------------------------------------------------------------------------------
// partials[block] = { sum_{c,p in block} (|x - y| + eps)^q * occ[p],  sum_{p in block} occ[p] }
__global__ __launch_bounds__(NT)
void robust_fwd_kernel(const float* __restrict__ x, const float* __restrict__ y, const float* __restrict__ occ,
                       float* __restrict__ partials, int C, int HW, long long npix, float eps, float q) {
  __shared__ float sh[NT / 64];
  float s = 0.f, so = 0.f;
  for (long long p = blockIdx.x * (long long)NT + threadIdx.x; p < npix; p += (long long)gridDim.x * NT) {
    const long long n = p / HW;
    const int i = (int)(p - n * HW);
    const float o = occ ? occ[p] : 1.0f;
    so += o;
    const float* xb = x + (size_t)n * C * HW + i;
    const float* yb = y + (size_t)n * C * HW + i;
    float t = 0.f;
    for (int c = 0; c < C; ++c) t += powf(fabsf(xb[(size_t)c * HW] - (y ? yb[(size_t)c * HW] : 0.f)) + eps, q);
    s += t * o;
  }
  const float a = block_sum(s, sh), b = block_sum(so, sh);
  if (threadIdx.x == 0) { partials[2 * blockIdx.x] = a; partials[2 * blockIdx.x + 1] = b; }
}

// gx = coef * occ * q * (|d| + eps)^(q-1) * sign(d), gy = -gx   (coef: device scalar = upstream gradient / denominator)
__global__ __launch_bounds__(NT)
void robust_bwd_kernel(const float* __restrict__ x, const float* __restrict__ y, const float* __restrict__ occ,
                       const float* __restrict__ coef, float* __restrict__ gx, float* __restrict__ gy,
                       int C, int HW, long long total, float eps, float q) {
  const long long e = blockIdx.x * (long long)NT + threadIdx.x;
  if (e >= total) return;
  const long long nc = e / HW;
  const int i = (int)(e - nc * HW);
  const long long n = nc / C;
  const float d = x[e] - y[e];
  const float o = occ ? occ[n * HW + i] : 1.0f;
  const float sg = (d > 0.f) ? 1.0f : ((d < 0.f) ? -1.0f : 0.f);
  const float g = coef[0] * o * q * powf(fabsf(d) + eps, q - 1.0f) * sg;
  if (gx) gx[e] = g;
  if (gy) gy[e] = -g;
}

// grey = 0.2989 r + 0.5870 g + 0.1140 b, evaluated left to right with separate roundings like the reference's expression
// (utils/loss.py:53-55; this file is compiled with -ffp-contract=off): one launch instead of five element-wise ones per image
__global__ void grey_kernel(const float* __restrict__ img, float* __restrict__ out, int HW, long long total) {
  const long long e = blockIdx.x * (long long)blockDim.x + threadIdx.x;
  if (e >= total) return;
  const long long n = e / HW;
  const float* p = img + (size_t)n * 3 * HW + (e - n * HW);
  out[e] = (0.2989f * p[0] + 0.5870f * p[HW]) + 0.1140f * p[2 * (size_t)HW];
}

// ---- first-order edge-aware smoothness ------------------------------------------------------------------------------------
// partials[block] = { sum |pred(i,j) - pred(i+1,j)| * wx(i,j),  sum |pred(i,j) - pred(i,j+1)| * wy(i,j) }
__device__ __forceinline__ float edge_w(const float* __restrict__ img, int Ci, int HW, int a, int b) {
  float s = 0.f;
  for (int c = 0; c < Ci; ++c) s += fabsf(img[(size_t)c * HW + a] - img[(size_t)c * HW + b]);
  return expf(-(s / (float)Ci));
}

__global__ __launch_bounds__(NT)
void smooth1_fwd_kernel(const float* __restrict__ img, const float* __restrict__ pred, float* __restrict__ partials,
                        int Ci, int Cp, int H, int W, long long npix) {
  __shared__ float sh[NT / 64];
  const int HW = H * W;
  float sx = 0.f, sy = 0.f;
  for (long long p = blockIdx.x * (long long)NT + threadIdx.x; p < npix; p += (long long)gridDim.x * NT) {
    const long long n = p / HW;
    const int q = (int)(p - n * HW), i = q / W, j = q - i * W;
    const float* im = img + (size_t)n * Ci * HW;
    const float* pr = pred + (size_t)n * Cp * HW;
    if (i + 1 < H) {
      const float wgt = edge_w(im, Ci, HW, q, q + W);
      float t = 0.f;
      for (int c = 0; c < Cp; ++c) t += fabsf(pr[(size_t)c * HW + q] - pr[(size_t)c * HW + q + W]);
      sx += t * wgt;
    }
    if (j + 1 < W) {
      const float wgt = edge_w(im, Ci, HW, q, q + 1);
      float t = 0.f;
      for (int c = 0; c < Cp; ++c) t += fabsf(pr[(size_t)c * HW + q] - pr[(size_t)c * HW + q + 1]);
      sy += t * wgt;
    }
  }
  const float a = block_sum(sx, sh), b = block_sum(sy, sh);
  if (threadIdx.x == 0) { partials[2 * blockIdx.x] = a; partials[2 * blockIdx.x + 1] = b; }
}

__device__ __forceinline__ float sgn(float d) { return (d > 0.f) ? 1.0f : ((d < 0.f) ? -1.0f : 0.f); }

// gather: pixel (i,j) appears as the minuend of its own two differences and as the subtrahend of its upper / left ones
__global__ __launch_bounds__(NT)
void smooth1_bwd_kernel(const float* __restrict__ img, const float* __restrict__ pred, const float* __restrict__ gup,
                        float* __restrict__ gpred, int Ci, int Cp, int H, int W, long long npix, float inv_nx, float inv_ny) {
  const long long p = blockIdx.x * (long long)NT + threadIdx.x;
  if (p >= npix) return;
  const int HW = H * W;
  const long long n = p / HW;
  const int q = (int)(p - n * HW), i = q / W, j = q - i * W;
  const float* im = img + (size_t)n * Ci * HW;
  const float* pr = pred + (size_t)n * Cp * HW;
  const float cx = gup[0] * inv_nx, cy = gup[0] * inv_ny;
  const float wd = (i + 1 < H) ? edge_w(im, Ci, HW, q, q + W) : 0.f, wu = (i > 0) ? edge_w(im, Ci, HW, q - W, q) : 0.f;
  const float wr = (j + 1 < W) ? edge_w(im, Ci, HW, q, q + 1) : 0.f, wl = (j > 0) ? edge_w(im, Ci, HW, q - 1, q) : 0.f;
  for (int c = 0; c < Cp; ++c) {
    const float* pc = pr + (size_t)c * HW;
    const float v = pc[q];
    float g = 0.f;
    if (i + 1 < H) g += cx * wd * sgn(v - pc[q + W]);
    if (i > 0) g -= cx * wu * sgn(pc[q - W] - v);
    if (j + 1 < W) g += cy * wr * sgn(v - pc[q + 1]);
    if (j > 0) g -= cy * wl * sgn(pc[q - 1] - v);
    gpred[((size_t)n * Cp + c) * HW + q] = g;
  }
}

static int red_blocks(long long n) {
  long long b = (n + NT - 1) / NT;
  return (int)(b < 1 ? 1 : (b > 1024 ? 1024 : b));
}

}  // namespace loss
}  // namespace upf

extern "C" int upf_boundary_warp_forward(const float* image, const float* flow, const float* start, float* out,
                                         int B, int C, int Hi, int Wi, int h, int w, void* stream) {
  using namespace upf;
  UPF_REQUIRE(image && flow && start && out, UPF_EINVAL, "boundary_warp_forward: null pointer");
  UPF_REQUIRE(B > 0 && B <= 65535 && C > 0 && Hi > 0 && Wi > 0 && h > 0 && w > 0, UPF_EINVAL, "boundary_warp_forward: bad shape");
  hipLaunchKernelGGL(loss::boundary_warp_fwd_kernel, dim3(cdiv(h * w, loss::NT), B), dim3(loss::NT), 0, (hipStream_t)stream,
                     image, flow, start, out, C, Hi, Wi, h, w);
  return check_launch("boundary_warp_forward");
}

extern "C" int upf_boundary_warp_backward(const float* image, const float* flow, const float* start, const float* grad_out,
                                          float* grad_flow, int B, int C, int Hi, int Wi, int h, int w, void* stream) {
  using namespace upf;
  UPF_REQUIRE(image && flow && start && grad_out && grad_flow, UPF_EINVAL, "boundary_warp_backward: null pointer");
  UPF_REQUIRE(B > 0 && B <= 65535 && C > 0 && Hi > 0 && Wi > 0 && h > 0 && w > 0, UPF_EINVAL, "boundary_warp_backward: bad shape");
  hipLaunchKernelGGL(loss::boundary_warp_bwd_kernel, dim3(cdiv(h * w, loss::NT), B), dim3(loss::NT), 0, (hipStream_t)stream,
                     image, flow, start, grad_out, grad_flow, C, Hi, Wi, h, w);
  return check_launch("boundary_warp_backward");
}

extern "C" int upf_loss_partials(long long n_pixels) { return upf::loss::red_blocks(n_pixels); }

extern "C" int upf_robust_loss_forward(const float* x, const float* y, const float* occ, float* partials,
                                       int B, int C, int HW, float eps, float q, void* stream) {
  using namespace upf;
  UPF_REQUIRE(x && y && partials, UPF_EINVAL, "robust_loss_forward: null pointer");
  UPF_REQUIRE(B > 0 && C > 0 && HW > 0, UPF_EINVAL, "robust_loss_forward: bad shape");
  const long long npix = (long long)B * HW;
  hipLaunchKernelGGL(loss::robust_fwd_kernel, dim3(loss::red_blocks(npix)), dim3(loss::NT), 0, (hipStream_t)stream,
                     x, y, occ, partials, C, HW, npix, eps, q);
  return check_launch("robust_loss_forward");
}

extern "C" int upf_robust_loss_backward(const float* x, const float* y, const float* occ, const float* coef,
                                        float* grad_x, float* grad_y, int B, int C, int HW, float eps, float q, void* stream) {
  using namespace upf;
  UPF_REQUIRE(x && y && coef && (grad_x || grad_y), UPF_EINVAL, "robust_loss_backward: null pointer");
  UPF_REQUIRE(B > 0 && C > 0 && HW > 0, UPF_EINVAL, "robust_loss_backward: bad shape");
  const long long total = (long long)B * C * HW;
  UPF_REQUIRE((total + loss::NT - 1) / loss::NT < (1ll << 31), UPF_EINVAL, "robust_loss_backward: grid too large");
  hipLaunchKernelGGL(loss::robust_bwd_kernel, dim3((unsigned)((total + loss::NT - 1) / loss::NT)), dim3(loss::NT), 0, (hipStream_t)stream,
                     x, y, occ, coef, grad_x, grad_y, C, HW, total, eps, q);
  return check_launch("robust_loss_backward");
}

extern "C" int upf_grey_forward(const float* image, float* grey, int B, int HW, void* stream) {
  using namespace upf;
  UPF_REQUIRE(image && grey && B > 0 && HW > 0, UPF_EINVAL, "grey_forward: bad arguments");
  const long long total = (long long)B * HW;
  UPF_REQUIRE((total + 255) / 256 < (1ll << 31), UPF_EINVAL, "grey_forward: tensor too large");
  hipLaunchKernelGGL(loss::grey_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, image, grey, HW, total);
  return check_launch("grey_forward");
}

extern "C" int upf_smooth_edge1_forward(const float* img, const float* pred, float* partials,
                                        int B, int Ci, int Cp, int H, int W, void* stream) {
  using namespace upf;
  UPF_REQUIRE(img && pred && partials, UPF_EINVAL, "smooth_edge1_forward: null pointer");
  UPF_REQUIRE(B > 0 && Ci > 0 && Cp > 0 && H > 0 && W > 0, UPF_EINVAL, "smooth_edge1_forward: bad shape");
  const long long npix = (long long)B * H * W;
  hipLaunchKernelGGL(loss::smooth1_fwd_kernel, dim3(loss::red_blocks(npix)), dim3(loss::NT), 0, (hipStream_t)stream,
                     img, pred, partials, Ci, Cp, H, W, npix);
  return check_launch("smooth_edge1_forward");
}

extern "C" int upf_smooth_edge1_backward(const float* img, const float* pred, const float* grad_up, float* grad_pred,
                                         int B, int Ci, int Cp, int H, int W, void* stream) {
  using namespace upf;
  UPF_REQUIRE(img && pred && grad_up && grad_pred, UPF_EINVAL, "smooth_edge1_backward: null pointer");
  UPF_REQUIRE(B > 0 && Ci > 0 && Cp > 0 && H > 1 && W > 1, UPF_EINVAL, "smooth_edge1_backward: bad shape");
  const long long npix = (long long)B * H * W;
  const float inv_nx = 1.0f / ((float)B * Cp * (H - 1) * W), inv_ny = 1.0f / ((float)B * Cp * H * (W - 1));
  hipLaunchKernelGGL(loss::smooth1_bwd_kernel, dim3((unsigned)((npix + loss::NT - 1) / loss::NT)), dim3(loss::NT), 0, (hipStream_t)stream,
                     img, pred, grad_up, grad_pred, Ci, Cp, H, W, npix, inv_nx, inv_ny);
  return check_launch("smooth_edge1_backward");
}
