// 81-neighbour cost volume, backward — gfx950.
//
// Replaces correlation_backward_input1/2<T> (/root/reference/model/correlation_package/
// correlation_cuda_kernel.cu:116-300) and their host loop of 2*B launches over a (H,W,C) grid of
// 32-thread blocks (:488-520), plus the two NHWC staging passes (:461-483).
//
//   g1[n,c,y,x] = (1/C) sum_d gO[n,d,y,x]       * f2[n,c,y+dy,x+dx]
//   g2[n,c,y,x] = (1/C) sum_d gO[n,d,y-dy,x-dx] * f1[n,c,y-dy,x-dx]      (out-of-range terms dropped)
//
// Both are "81 per-pixel weights times a 9x9 neighbourhood of one feature channel".  Two kernels: the LDS-tiled one
// further down (W >= 4) and this gather kernel (any shape; the fallback), in which a thread owns
// one pixel: it loads its 81 weights ONCE into registers (gO is the big tensor: 81 channels), then
// walks its slice of the C channels gathering the neighbourhood (L1/L2 hits: neighbouring lanes read
// neighbouring addresses).  gO is therefore read exactly once per direction from HBM; fp32
// accumulation.  One launch covers the whole batch and both gradients (blockIdx.z selects which).
#include "common.hpp"
#include <stdlib.h>

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


// ---- LDS-tiled version (W % 4 == 0) ----------------------------------------------------------------------------
// The gather kernel above issues 81 four-byte loads per (pixel, channel): TA-bound at a few percent of the HBM
// roofline.  Here a workgroup owns a 16 x 64 pixel tile and CC channels of ONE gradient (WHICH = 0: g1, 1: g2):
//   * the feature tile (+ 4-pixel halo, zero padded through the buffer descriptor) of the CC channels is staged once
//     into LDS as fp32;
//   * a thread owns 4 consecutive pixels: per displacement row it fetches its 9 x 4 weights (gO at the pixel itself for
//     g1, at the displaced source pixel for g2) and then, per channel, reads the 12-float window that holds every
//     neighbour of its 4 pixels with three 16-byte LDS reads: 36 FMAs per 3 LDS reads instead of per 36 global loads.
// gO is read once per channel chunk (L2-resident after the first), the features once, fp32 accumulation.
constexpr int BTH = 16, BTW = 64, BCC = 4;
constexpr int BROWS = BTH + 8, BPITCH = BTW + 8;                 // tile incl. halo, floats

// RW (ragged width, W % 4 != 0 — the 13- and 26-pixel levels of the trainer's crops, which took the gather kernel at
// 90 us per launch): the 4-column group that straddles the row end is loaded whole (2-byte-aligned 8-byte buffer loads are
// legal on gfx950) and the columns beyond W are zeroed / not stored.
template <typename T, bool WHICH, bool RW>
__global__ __launch_bounds__(256, 4)
void corr81_bwd_tiled_kernel(const T* __restrict__ feat, const T* __restrict__ gO, T* __restrict__ gout,
                             int C, int H, int W, int tiles_x) {
  __shared__ __attribute__((aligned(16))) float tile[BCC * BROWS * BPITCH];
  const int tx = blockIdx.x % tiles_x, ty = blockIdx.x / tiles_x;
  const int x0 = tx * BTW, y0 = ty * BTH;
  const int c0 = blockIdx.y * BCC;
  const int n = blockIdx.z;
  const int HW = H * W;
  const int tid = threadIdx.x;
  constexpr int ES = sizeof(typename Elem<T>::store_t);
  const uint32_t plane = (uint32_t)HW * ES;
  __amdgpu_buffer_rsrc_t fr = __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(feat + (size_t)n * C * HW), 0, (uint32_t)C * plane, 0x00020000);
  __amdgpu_buffer_rsrc_t gr = __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(gO + (size_t)n * 81 * HW), 0, 81u * plane, 0x00020000);
  auto load4 = [&](__amdgpu_buffer_rsrc_t r, uint32_t off, float (&v)[4]) {       // 4 consecutive elements -> fp32
    if constexpr (ES == 4) {
      const auto t = __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 0);
      v[0] = __uint_as_float(t[0]); v[1] = __uint_as_float(t[1]); v[2] = __uint_as_float(t[2]); v[3] = __uint_as_float(t[3]);
    } else {
      const auto t = __builtin_amdgcn_raw_buffer_load_b64(r, off, 0, 0);
      T e[4];
      e[0].v = (uint16_t)t[0]; e[1].v = (uint16_t)(t[0] >> 16); e[2].v = (uint16_t)t[1]; e[3].v = (uint16_t)(t[1] >> 16);
#pragma unroll
      for (int i = 0; i < 4; ++i) v[i] = Elem<T>::load(&e[i]);
    }
  };
  // elements [xx, xx + 4) of the row that starts at element `row`; RW: a vector that straddles the row end is loaded
  // SHIFTED LEFT so that it ends at the row end and shifted back in registers (zeros beyond the row): no load leaves its
  // row, so nothing depends on what follows the tensor (a load straddling the end of the descriptor loses its last
  // partial dword — the last pixel of the last plane).
  auto load4r = [&](__amdgpu_buffer_rsrc_t r, int row, int xx, bool in, float (&v)[4]) {
    int sh = 0;
    if constexpr (RW) sh = (in && xx < W && xx + 4 > W) ? xx + 4 - W : 0;
    load4(r, in ? (uint32_t)(row + xx - sh) * ES : 0x80000000u, v);
    if constexpr (RW) {
      if (sh) {
        const float a = v[0], b = v[1], c = v[2], d = v[3];
        (void)a;
        v[0] = sh == 1 ? b : (sh == 2 ? c : d);
        v[1] = sh == 1 ? c : (sh == 2 ? d : 0.f);
        v[2] = sh == 1 ? d : 0.f;
        v[3] = 0.f;
      }
    }
  };
  // ---- stage the feature tile: groups of 4 columns (!RW: all in or all out of the row: x0 - 4 and W are multiples of 4)
  constexpr int GPR = BPITCH / 4;                                // 18 groups per tile row
  for (int g = tid; g < BCC * BROWS * GPR; g += 256) {
    const int c = g / (BROWS * GPR), rem = g - c * (BROWS * GPR), r = rem / GPR, k = rem - r * GPR;
    const int gy = y0 - 4 + r, gx = x0 - 4 + 4 * k;
    const bool in = gy >= 0 && gy < H && gx >= 0 && gx < W;      // (channels >= C fall off the descriptor)
    float v[4];
    load4r(fr, (c0 + c) * HW + gy * W, gx, in, v);
    *reinterpret_cast<float4*>(&tile[(c * BROWS + r) * BPITCH + 4 * k]) = make_float4(v[0], v[1], v[2], v[3]);
  }
  __syncthreads();

  const int r = tid >> 4, cg = tid & 15;                         // tile row, 4-pixel column group
  const int y = y0 + r, x = x0 + 4 * cg;
  float acc[BCC][4];
#pragma unroll
  for (int c = 0; c < BCC; ++c)
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[c][i] = 0.f;
  const bool live = y < H && x < W;
  // weights of displacement row dyi: g1: gO[d] at the pixels themselves; g2: gO[d] at the displaced SOURCE pixels.
  // Columns of a displaced 4-vector that leave the row [0, W) are elements of the NEIGHBOURING row in memory: they meet the
  // zero halo of the feature tile, but 0 * (Inf / NaN of that other row) would still poison the border pixels (the
  // reference drops those terms, correlation_cuda_kernel.cu:228-262), so they are zeroed here — only lanes within 4
  // pixels of the left / right image border ever take that branch.
  const bool edge = WHICH && (x < 4 || x + 8 > W);
  auto load_w = [&](int dyi, float (&w)[9][4]) {
    const int dy = dyi - 4;
#pragma unroll
    for (int dxi = 0; dxi < 9; ++dxi) {
      const int yy = WHICH ? y - dy : y, xx = WHICH ? x - (dxi - 4) : x;
      const bool in = live && yy >= 0 && yy < H;
      load4r(gr, ((dyi * 9 + dxi) * H + yy) * W, xx, in, w[dxi]);
      if (edge) {
#pragma unroll
        for (int i = 0; i < 4; ++i) w[dxi][i] = (xx + i >= 0 && xx + i < W) ? w[dxi][i] : 0.f;
      }
    }
  };
  auto compute = [&](int dyi, const float (&w)[9][4]) {
    const int dy = dyi - 4;
    const int srow = r + 4 + (WHICH ? -dy : dy);
#pragma unroll
    for (int c = 0; c < BCC; ++c) {
      const float* row = &tile[(c * BROWS + srow) * BPITCH + 4 * cg];
      const float4 a = *reinterpret_cast<const float4*>(row), b = *reinterpret_cast<const float4*>(row + 4), d = *reinterpret_cast<const float4*>(row + 8);
      const float win[12] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w, d.x, d.y, d.z, d.w};
#pragma unroll
      for (int dxi = 0; dxi < 9; ++dxi)
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[c][i] = __builtin_fmaf(w[dxi][i], win[WHICH ? i + 8 - dxi : i + dxi], acc[c][i]);
    }
  };
  // (latency of the 9 weight loads per row is hidden by occupancy: 27 KB of LDS and ~100 registers per workgroup)
  float w[9][4];
#pragma unroll 1
  for (int dyi = 0; dyi < 9; ++dyi) {
    load_w(dyi, w);
    compute(dyi, w);
  }
  if (!live) return;
  const float invC = 1.0f / (float)C;
#pragma unroll
  for (int c = 0; c < BCC; ++c) {
    if (c0 + c >= C) break;
    T* dst = gout + ((size_t)n * C + c0 + c) * HW + (size_t)y * W + x;
    if constexpr (RW) {
#pragma unroll
      for (int i = 0; i < 4; ++i)
        if (x + i < W) Elem<T>::store(dst + i, acc[c][i] * invC);
    } else if constexpr (ES == 4) {
      *reinterpret_cast<float4*>(dst) = make_float4(acc[c][0] * invC, acc[c][1] * invC, acc[c][2] * invC, acc[c][3] * invC);
    } else {
      *reinterpret_cast<uint2*>(dst) = make_uint2(pack2<T>(acc[c][0] * invC, acc[c][1] * invC), pack2<T>(acc[c][2] * invC, acc[c][3] * invC));
    }
  }
}

// ---- general-parameter gradients (kernel_size 1, stride1 1: where the reference's backward kernels ARE the gradient of its
// forward, oracle/ops.py:correlation_backward_supported).  One thread per input element, both gradients:
//   g1[n,c,y,x] = (1/C) sum_tc gO[n,tc,oy,ox]               * in2[n,c,y + j2, x + i2],   (oy,ox) = (y,x) + pad - md
//   g2[n,c,y,x] = (1/C) sum_tc gO[n,tc,oy - j2, ox - i2]    * in1[n,c,y - j2, x - i2],   (i2,j2) = displacement of channel tc
// (correlation_cuda_kernel.cu:116-207, :209-300 with xmin = xmax, ymin = ymax); terms whose output pixel or input pixel lies
// outside are zero.  Not a tuned kernel: the model never instantiates these parameter sets.
template <typename T>
__global__ void corr_general_bwd_kernel(const T* __restrict__ in1, const T* __restrict__ in2, const T* __restrict__ gO,
                                        T* __restrict__ g1, T* __restrict__ g2, int B, int C, int H, int W, int pad, int md,
                                        int s2, int dr, int oH, int oW) {
  const int ds = 2 * dr + 1;
  const long long total = (long long)B * C * H * W;
  const float invC = 1.0f / (float)C;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int x = (int)(i % W), y = (int)((i / W) % H);
    const int c = (int)((i / ((long long)W * H)) % C), n = (int)(i / ((long long)W * H * C));
    const T* a = in1 + ((size_t)n * C + c) * H * W;
    const T* b = in2 + ((size_t)n * C + c) * H * W;
    const T* go = gO + (size_t)n * ds * ds * oH * oW;
    const int oy = y + pad - md, ox = x + pad - md;
    float acc1 = 0.f, acc2 = 0.f;
    for (int tc = 0; tc < ds * ds; ++tc) {
      const int i2 = (tc % ds - dr) * s2, j2 = (tc / ds - dr) * s2;
      if (oy >= 0 && oy < oH && ox >= 0 && ox < oW) {
        const int yb = y + j2, xb = x + i2;
        if (yb >= 0 && yb < H && xb >= 0 && xb < W)
          acc1 = __builtin_fmaf(Elem<T>::load(go + ((size_t)tc * oH + oy) * oW + ox), Elem<T>::load(b + (size_t)yb * W + xb), acc1);
      }
      const int oy2 = oy - j2, ox2 = ox - i2, ya = y - j2, xa = x - i2;
      if (oy2 >= 0 && oy2 < oH && ox2 >= 0 && ox2 < oW && ya >= 0 && ya < H && xa >= 0 && xa < W)
        acc2 = __builtin_fmaf(Elem<T>::load(go + ((size_t)tc * oH + oy2) * oW + ox2), Elem<T>::load(a + (size_t)ya * W + xa), acc2);
    }
    Elem<T>::store(g1 + i, acc1 * invC);
    Elem<T>::store(g2 + i, acc2 * invC);
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
  const size_t es = dtype == UPF_F32 ? 4 : 2;
  const bool fits = (size_t)81 * HW * es < (1ull << 31) && (size_t)C * HW * es < (1ull << 31) && getenv("UPF_CORR_BWD_GATHER") == nullptr;
  const bool aligned = W % 4 == 0 && aligned_to(f1, 16) && aligned_to(f2, 16) && aligned_to(grad_out, 8) && aligned_to(g1, 16) && aligned_to(g2, 16);
  if (fits && (aligned || W >= 4)) {
    const int tiles_x = cdiv(W, corr::BTW), tiles_y = cdiv(H, corr::BTH);
    dim3 grid(tiles_x * tiles_y, cdiv(C, corr::BCC), B);
    if (aligned) {
      UPF_DISPATCH(dtype, T,
                   hipLaunchKernelGGL((corr::corr81_bwd_tiled_kernel<T, false, false>), grid, dim3(256), 0, (hipStream_t)stream, (const T*)f2, (const T*)grad_out, (T*)g1, C, H, W, tiles_x);
                   hipLaunchKernelGGL((corr::corr81_bwd_tiled_kernel<T, true, false>), grid, dim3(256), 0, (hipStream_t)stream, (const T*)f1, (const T*)grad_out, (T*)g2, C, H, W, tiles_x));
    } else {
      UPF_DISPATCH(dtype, T,
                   hipLaunchKernelGGL((corr::corr81_bwd_tiled_kernel<T, false, true>), grid, dim3(256), 0, (hipStream_t)stream, (const T*)f2, (const T*)grad_out, (T*)g1, C, H, W, tiles_x);
                   hipLaunchKernelGGL((corr::corr81_bwd_tiled_kernel<T, true, true>), grid, dim3(256), 0, (hipStream_t)stream, (const T*)f1, (const T*)grad_out, (T*)g2, C, H, W, tiles_x));
    }
    return check_launch("corr81_backward");
  }
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

extern "C" int upf_correlation_backward(const void* in1, const void* in2, const void* grad_out, void* g1, void* g2,
                                        int B, int C, int H, int W, int dtype, int pad_size, int kernel_size, int max_displacement,
                                        int stride1, int stride2, int corr_type_multiply, void* stream) {
  using namespace upf;
  (void)corr_type_multiply;
  if (pad_size == 4 && kernel_size == 1 && max_displacement == 4 && stride1 == 1 && stride2 == 1)
    return upf_corr81_backward(in1, in2, grad_out, g1, g2, B, C, H, W, dtype, stream);
  int oc, oh, ow;
  int rc = upf_correlation_out_shape(H, W, pad_size, kernel_size, max_displacement, stride1, stride2, &oc, &oh, &ow);
  if (rc != UPF_OK) return rc;
  UPF_REQUIRE(in1 && in2 && grad_out && g1 && g2 && B > 0 && C > 0, UPF_EINVAL, "correlation_backward: bad arguments");
  UPF_REQUIRE(kernel_size == 1 && stride1 == 1, UPF_EUNSUPPORTED,
              "correlation_backward: kernel_size %d / stride1 %d — the reference's backward kernels are the gradient of its forward only for "
              "kernel_size 1, stride1 1 (correlation_cuda_kernel.cu:129-141); not implemented elsewhere", kernel_size, stride1);
  const long long total = (long long)B * C * H * W;
  const int threads = 256;
  const int blocks = (int)((total + threads - 1) / threads > 65535 * 16 ? 65535 * 16 : (total + threads - 1) / threads);
  UPF_DISPATCH(dtype, T,
               hipLaunchKernelGGL((corr::corr_general_bwd_kernel<T>), dim3(blocks), dim3(threads), 0, (hipStream_t)stream,
                                  (const T*)in1, (const T*)in2, (const T*)grad_out, (T*)g1, (T*)g2, B, C, H, W, pad_size,
                                  max_displacement, stride2, max_displacement / stride2, oh, ow));
  return check_launch("correlation_backward");
}
