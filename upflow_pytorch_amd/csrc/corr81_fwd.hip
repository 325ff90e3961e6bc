// 81-neighbour cost volume, forward — hand-written for gfx950 (MI355X / CDNA4).
//
// Replaces correlation_forward<T> + 2x channels_first<T> of the reference
// (/root/reference/model/correlation_package/correlation_cuda_kernel.cu:15-114, :302-393), which run
// one 32-thread block per output pixel, 81 barrier+serial-reduce rounds, on padded NHWC copies.
//
// Design:
//   * reads NCHW directly — no NHWC staging pass; the layout change happens on the way into LDS;
//   * one workgroup = 9 wavefronts = one 8x32 pixel tile; wavefront w owns displacement row
//     dy = w-4, so `dy` is wave-uniform and every lane keeps only 9(dx) x 4(px) = 36 fp32
//     accumulators; a lane owns 4 consecutive pixels of one row;
//   * LDS holds the f1 tile and the f2 tile with its 4-pixel halo for a chunk of 16 "k-slots";
//     a k-slot is one fp32 channel, or a PAIR of bf16/fp16 channels interleaved per pixel so that
//     one v_dot2c_f32_bf16 / v_dot2c_f32_f16 retires two channels (fp32 accumulate);
//   * per k-slot a lane issues 4 ds_read_b128 (4 dwords of f1, 12 of f2) for 36 MACs/dot2s;
//     the lane -> (row, x-block) map follows the hardware's 16-lane ds_read_b128 groups so that each
//     group touches 16 distinct 16-byte LDS slots (rows r and r+4 are 8 slots apart at stride 40);
//   * epilogue divides by C (like `reduce_sum / nelems`, correlation_cuda_kernel.cu:108), optionally
//     applies LeakyReLU (model/upflow.py:563-564) and writes 4 pixels per store, optionally into a
//     wider channel buffer (out_batch_stride);
//   * blockIdx is remapped so that each XCD (private L2) gets a contiguous run of tiles.
#include "common.hpp"
#include <hip/hip_ext.h>

namespace upf {
namespace corr {

constexpr int R = 4, D = 9, ND = 81;
constexpr int TH = 8, TW = 32, PX = 4, XB = TW / PX;
constexpr int S1 = TW + 8;            // f1 LDS row stride (dwords); +8 keeps rows r, r+4 on disjoint slots
constexpr int S2 = TW + 2 * R;        // f2 tile width incl. halo = 40 dwords
constexpr int R2 = TH + 2 * R;        // 16 rows incl. halo
constexpr int SLOT1 = TH * S1;        // dwords per k-slot, f1
constexpr int SLOT2 = R2 * S2;        // dwords per k-slot, f2
constexpr int KC = 16;                // k-slots per LDS chunk
constexpr int NWAVES = D;
constexpr int NTHREADS = NWAVES * 64;
constexpr int LDS_BYTES = KC * (SLOT1 + SLOT2) * 4;   // 61,440 B -> two workgroups per CU


template <typename T> struct Slot;
template <> struct Slot<float> {
  static constexpr int CH = 1;   // channels per k-slot
  static __device__ __forceinline__ float mac(uint32_t a, uint32_t b, float c) {
    return __builtin_fmaf(__uint_as_float(a), __uint_as_float(b), c);
  }
};
template <> struct Slot<bf16_t> {
  static constexpr int CH = 2;
  static __device__ __forceinline__ float mac(uint32_t a, uint32_t b, float c) {
    return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2_t, a), __builtin_bit_cast(bf16x2_t, b), c, false);
  }
};
template <> struct Slot<f16_t> {
  static constexpr int CH = 2;
  static __device__ __forceinline__ float mac(uint32_t a, uint32_t b, float c) {
    return __builtin_amdgcn_fdot2(__builtin_bit_cast(f16x2_t, a), __builtin_bit_cast(f16x2_t, b), c, false);
  }
};

// Load the 4 dwords (4 consecutive pixels) of one k-slot for image row `gy`, columns gx..gx+3.
// ALIGNED: W % 4 == 0 and 4-element-aligned base pointers, so a quad is entirely inside or outside.
template <typename T, bool ALIGNED>
__device__ __forceinline__ uint4 load_quad(const T* __restrict__ f, int C, int H, int W, int n, int kslot, int gy, int gx) {
  uint4 r = make_uint4(0u, 0u, 0u, 0u);
  if (gy < 0 || gy >= H) return r;
  if constexpr (Slot<T>::CH == 1) {
    const float* p = reinterpret_cast<const float*>(f) + (((size_t)n * C + kslot) * H + gy) * (size_t)W;
    if constexpr (ALIGNED) {
      if (gx >= 0 && gx < W) r = *reinterpret_cast<const uint4*>(p + gx);
    } else {
      uint32_t v[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) { int x = gx + i; v[i] = (x >= 0 && x < W) ? __float_as_uint(p[x]) : 0u; }
      r = make_uint4(v[0], v[1], v[2], v[3]);
    }
  } else {
    const int c0 = 2 * kslot;
    const uint16_t* p0 = reinterpret_cast<const uint16_t*>(f) + (((size_t)n * C + c0) * H + gy) * (size_t)W;
    const uint16_t* p1 = p0 + (size_t)H * W;
    const bool has1 = (c0 + 1) < C;
    if constexpr (ALIGNED) {
      if (gx >= 0 && gx < W) {
        uint2 a = *reinterpret_cast<const uint2*>(p0 + gx);
        uint2 b = has1 ? *reinterpret_cast<const uint2*>(p1 + gx) : make_uint2(0u, 0u);
        // interleave channel c0 (low half) with c0+1 (high half), per pixel
        r.x = __builtin_amdgcn_perm(b.x, a.x, 0x05040100u);
        r.y = __builtin_amdgcn_perm(b.x, a.x, 0x07060302u);
        r.z = __builtin_amdgcn_perm(b.y, a.y, 0x05040100u);
        r.w = __builtin_amdgcn_perm(b.y, a.y, 0x07060302u);
      }
    } else {
      uint32_t v[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        int x = gx + i;
        uint32_t lo = 0u, hi = 0u;
        if (x >= 0 && x < W) { lo = p0[x]; hi = has1 ? p1[x] : 0u; }
        v[i] = lo | (hi << 16);
      }
      r = make_uint4(v[0], v[1], v[2], v[3]);
    }
  }
  return r;
}

template <typename T, bool ALIGNED>
__global__ __launch_bounds__(NTHREADS)
void corr81_fwd_kernel(const T* __restrict__ f1, const T* __restrict__ f2, T* __restrict__ out,
                       int C, int H, int W, int tiles_x, int tiles_y, long long out_bs, float slope) {
  extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
  uint32_t* s1 = smem;
  uint32_t* s2 = smem + KC * SLOT1;

  const int bid = xcd_remap(blockIdx.x, gridDim.x);
  const int tx = bid % tiles_x;
  const int ty = (bid / tiles_x) % tiles_y;
  const int n = bid / (tiles_x * tiles_y);
  const int x0 = tx * TW, y0 = ty * TH;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int dyi = __builtin_amdgcn_readfirstlane(tid >> 6);     // 0..8  <->  dy = dyi - 4

  // lane -> (row, x-block) following the ds_read_b128 service groups
  // {0-3,12-15,20-27} {4-11,16-19,28-31} {32-35,44-47,52-59} {36-43,48-51,60-63}
  int row, xb;
  {
    const int l = lane & 31;
    const bool g1 = (l >= 4 && l < 12) || (l >= 16 && l < 20) || (l >= 28);
    int k;
    if (!g1) k = (l < 4) ? l : (l < 16 ? l - 8 : l - 12);
    else     k = (l < 12) ? l - 4 : (l < 20 ? l - 8 : l - 16);
    const int g = (lane >> 5) * 2 + (g1 ? 1 : 0);
    row = g + 4 * (k >> 3);
    xb = k & 7;
  }

  float acc[D][PX];
#pragma unroll
  for (int d = 0; d < D; ++d)
#pragma unroll
    for (int p = 0; p < PX; ++p) acc[d][p] = 0.f;

  const int nslots = (C + Slot<T>::CH - 1) / Slot<T>::CH;

  for (int kb = 0; kb < nslots; kb += KC) {
    const int kcnt = min(KC, nslots - kb);
    if (kb > 0) __syncthreads();
    // ---- stage f1 tile: kcnt x 8 rows x 8 quads
    for (int idx = tid; idx < kcnt * (TH * XB); idx += NTHREADS) {
      const int k = idx >> 6, rem = idx & 63, r = rem >> 3, q = rem & 7;
      uint4 v = load_quad<T, ALIGNED>(f1, C, H, W, n, kb + k, y0 + r, x0 + 4 * q);
      *reinterpret_cast<uint4*>(s1 + k * SLOT1 + r * S1 + 4 * q) = v;
    }
    // ---- stage f2 tile with halo: kcnt x 16 rows x 10 quads
    for (int idx = tid; idx < kcnt * (R2 * (S2 / 4)); idx += NTHREADS) {
      const int k = idx / (R2 * (S2 / 4));
      const int rem = idx - k * (R2 * (S2 / 4));
      const int r = rem / (S2 / 4), q = rem - r * (S2 / 4);
      uint4 v = load_quad<T, ALIGNED>(f2, C, H, W, n, kb + k, y0 - R + r, x0 - R + 4 * q);
      *reinterpret_cast<uint4*>(s2 + k * SLOT2 + r * S2 + 4 * q) = v;
    }
    __syncthreads();
    // ---- accumulate
    const uint32_t* p1 = s1 + row * S1 + 4 * xb;
    const uint32_t* p2 = s2 + (row + dyi) * S2 + 4 * xb;
#pragma unroll 2
    for (int k = 0; k < kcnt; ++k) {
      const uint4 a4 = *reinterpret_cast<const uint4*>(p1 + k * SLOT1);
      const uint4 b0 = *reinterpret_cast<const uint4*>(p2 + k * SLOT2);
      const uint4 b1 = *reinterpret_cast<const uint4*>(p2 + k * SLOT2 + 4);
      const uint4 b2 = *reinterpret_cast<const uint4*>(p2 + k * SLOT2 + 8);
      const uint32_t a[4] = {a4.x, a4.y, a4.z, a4.w};
      const uint32_t b[12] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w, b2.x, b2.y, b2.z, b2.w};
#pragma unroll
      for (int d = 0; d < D; ++d)
#pragma unroll
        for (int p = 0; p < PX; ++p) acc[d][p] = Slot<T>::mac(a[p], b[p + d], acc[d][p]);
    }
  }

  // ---- epilogue
  const int y = y0 + row, x = x0 + 4 * xb;
  if (y >= H || x >= W) return;
  const float fC = (float)C, invC = 1.0f / fC;
  using st = typename Elem<T>::store_t;
  st* obase = reinterpret_cast<st*>(out) + (size_t)n * out_bs + ((size_t)(dyi * D) * H + y) * W + x;
  const size_t cstride = (size_t)H * W;
#pragma unroll
  for (int d = 0; d < D; ++d) {
    float v[PX];
#pragma unroll
    for (int p = 0; p < PX; ++p) {
      float t = acc[d][p] * invC;
      if constexpr (sizeof(st) == 4) {
        // fp32 output is the parity mode: one Newton step makes acc*invC the correctly rounded acc/C
        // (`reduce_sum / nelems`, correlation_cuda_kernel.cu:108) without a 10-instruction IEEE divide
        const float r = __builtin_fmaf(-t, fC, acc[d][p]);
        t = __builtin_fmaf(r, invC, t);
      }
      v[p] = (slope != 0.f) ? fmaxf(t, t * slope) : t;     // LeakyReLU for 0 < slope < 1
    }
    st* o = obase + d * cstride;
    if constexpr (ALIGNED) {
      if constexpr (sizeof(st) == 4) {
        *reinterpret_cast<float4*>(o) = make_float4(v[0], v[1], v[2], v[3]);
      } else {
        *reinterpret_cast<uint2*>(o) = make_uint2(pack2<T>(v[0], v[1]), pack2<T>(v[2], v[3]));
      }
    } else {
#pragma unroll
      for (int p = 0; p < PX; ++p)
        if (x + p < W) Elem<T>::store(reinterpret_cast<T*>(o) + p, v[p]);
    }
  }
}

template <typename T>
int launch_fwd(const void* f1, const void* f2, void* out, int B, int C, int H, int W,
               long long out_bs, float slope, hipStream_t stream, hipEvent_t ev0 = nullptr, hipEvent_t ev1 = nullptr) {
  const int tiles_x = cdiv(W, TW), tiles_y = cdiv(H, TH);
  const long long nblocks = (long long)B * tiles_x * tiles_y;
  UPF_REQUIRE(nblocks < (1ll << 31), UPF_EINVAL, "corr81_forward: grid too large");
  const size_t va = 4 * sizeof(typename Elem<T>::store_t);   // bytes of a 4-pixel vector
  const bool aligned = (W % 4 == 0) && (out_bs % 4 == 0) && aligned_to(f1, va) && aligned_to(f2, va) && aligned_to(out, va);
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&corr81_fwd_kernel<T, true>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&corr81_fwd_kernel<T, false>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    attr_set = true;
  }
  // hipExtLaunchKernelGGL == hipLaunchKernelGGL plus optional start/stop events recorded right around
  // THIS kernel on its stream (used by upf_corr81_forward_timed; null events = plain launch)
  if (aligned)
    hipExtLaunchKernelGGL((corr81_fwd_kernel<T, true>), dim3((unsigned)nblocks), dim3(NTHREADS), LDS_BYTES, stream, ev0, ev1, 0,
                          (const T*)f1, (const T*)f2, (T*)out, C, H, W, tiles_x, tiles_y, out_bs, slope);
  else
    hipExtLaunchKernelGGL((corr81_fwd_kernel<T, false>), dim3((unsigned)nblocks), dim3(NTHREADS), LDS_BYTES, stream, ev0, ev1, 0,
                          (const T*)f1, (const T*)f2, (T*)out, C, H, W, tiles_x, tiles_y, out_bs, slope);
  return check_launch("corr81_forward");
}

// ---- general-parameter fallback: one thread per output element --------------------------------
// Same arithmetic as correlation_forward<T> (correlation_cuda_kernel.cu:41-114) on virtual zero
// padding; used only for parameter sets the model never instantiates.
template <typename T>
__global__ void corr_general_kernel(const T* __restrict__ in1, const T* __restrict__ in2, T* __restrict__ out,
                                    int B, int C, int H, int W, int pad, int kr, int md, int s1, int s2,
                                    int dr, int oH, int oW, int ksize) {
  const int ds = 2 * dr + 1;
  const long long total = (long long)B * ds * ds * oH * oW;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int ox = (int)(i % oW);
    const int oy = (int)((i / oW) % oH);
    const int tc = (int)((i / ((long long)oW * oH)) % (ds * ds));
    const int n = (int)(i / ((long long)oW * oH * ds * ds));
    const int ti = tc % ds - dr, tj = tc / ds - dr;
    const int y1 = oy * s1 + md - pad, x1 = ox * s1 + md - pad;        // un-padded coordinates
    const int y2 = y1 + tj * s2, x2 = x1 + ti * s2;
    float acc = 0.f;
    for (int j = -kr; j <= kr; ++j)
      for (int ii = -kr; ii <= kr; ++ii) {
        const int ya = y1 + j, xa = x1 + ii, yb = y2 + j, xb = x2 + ii;
        if (ya < 0 || ya >= H || xa < 0 || xa >= W || yb < 0 || yb >= H || xb < 0 || xb >= W) continue;
        for (int c = 0; c < C; ++c)
          acc += Elem<T>::load(in1 + (((size_t)n * C + c) * H + ya) * W + xa) * Elem<T>::load(in2 + (((size_t)n * C + c) * H + yb) * W + xb);
      }
    Elem<T>::store(out + i, acc / (float)(ksize * ksize * C));
  }
}

}  // namespace corr
}  // namespace upf

extern "C" int upf_corr81_forward(const void* f1, const void* f2, void* out, int B, int C, int H, int W, int dtype,
                                  long long out_batch_stride, float leaky_slope, void* stream) {
  using namespace upf;
  UPF_REQUIRE(f1 && f2 && out, UPF_EINVAL, "corr81_forward: null pointer");
  UPF_REQUIRE(B > 0 && C > 0 && H > 0 && W > 0, UPF_EINVAL, "corr81_forward: bad shape B=%d C=%d H=%d W=%d", B, C, H, W);
  if (out_batch_stride == 0) out_batch_stride = (long long)corr::ND * H * W;
  UPF_REQUIRE(out_batch_stride >= (long long)corr::ND * H * W, UPF_EINVAL, "corr81_forward: out_batch_stride %lld < 81*H*W", out_batch_stride);
  UPF_DISPATCH(dtype, T, return corr::launch_fwd<T>(f1, f2, out, B, C, H, W, out_batch_stride, leaky_slope, (hipStream_t)stream));
  return UPF_OK;
}

extern "C" int upf_corr81_forward_timed(const void* f1, const void* f2, void* out, int B, int C, int H, int W, int dtype,
                                        long long out_batch_stride, float leaky_slope, void* stream, int nrep,
                                        float* avg_us, float* min_us) {
  using namespace upf;
  UPF_REQUIRE(f1 && f2 && out && avg_us, UPF_EINVAL, "corr81_forward_timed: null pointer");
  UPF_REQUIRE(B > 0 && C > 0 && H > 0 && W > 0 && nrep > 0 && nrep <= 1024, UPF_EINVAL, "corr81_forward_timed: bad arguments");
  if (out_batch_stride == 0) out_batch_stride = (long long)corr::ND * H * W;
  hipStream_t s = (hipStream_t)stream;
  hipEvent_t* ev = new hipEvent_t[2 * nrep];
  for (int i = 0; i < 2 * nrep; ++i) (void)hipEventCreate(&ev[i]);
  int rc = UPF_OK;
  for (int i = 0; i < nrep && rc == UPF_OK; ++i) {
    UPF_DISPATCH(dtype, T, rc = corr::launch_fwd<T>(f1, f2, out, B, C, H, W, out_batch_stride, leaky_slope, s, ev[2 * i], ev[2 * i + 1]));
  }
  hipError_t e = hipStreamSynchronize(s);
  if (rc == UPF_OK && e != hipSuccess) { set_error("corr81_forward_timed: %s", hipGetErrorString(e)); rc = (int)e; }
  double sum = 0.0, mn = 1e30;
  if (rc == UPF_OK)
    for (int i = 0; i < nrep; ++i) {
      float ms = 0.f;
      (void)hipEventElapsedTime(&ms, ev[2 * i], ev[2 * i + 1]);
      sum += ms; if (ms < mn) mn = ms;
    }
  for (int i = 0; i < 2 * nrep; ++i) (void)hipEventDestroy(ev[i]);
  delete[] ev;
  if (rc == UPF_OK) { *avg_us = (float)(sum / nrep * 1e3); if (min_us) *min_us = (float)(mn * 1e3); }
  return rc;
}

extern "C" int upf_correlation_out_shape(int H, int W, int pad_size, int kernel_size, int max_displacement,
                                         int stride1, int stride2, int* out_channels, int* out_h, int* out_w) {
  using namespace upf;
  UPF_REQUIRE(stride1 > 0 && stride2 > 0 && kernel_size > 0 && pad_size >= 0 && max_displacement >= 0, UPF_EINVAL,
              "correlation: bad parameters pad=%d k=%d md=%d s1=%d s2=%d", pad_size, kernel_size, max_displacement, stride1, stride2);
  const int kr = (kernel_size - 1) / 2, br = kr + max_displacement;             // correlation_cuda.cc:24-25
  const int dr = max_displacement / stride2;
  const int oh = (H + 2 * pad_size - 2 * br + stride1 - 1) / stride1;           // ceil, :33-34
  const int ow = (W + 2 * pad_size - 2 * br + stride1 - 1) / stride1;
  UPF_REQUIRE(oh > 0 && ow > 0, UPF_EINVAL, "correlation: empty output %dx%d", oh, ow);
  if (out_channels) *out_channels = (2 * dr + 1) * (2 * dr + 1);
  if (out_h) *out_h = oh;
  if (out_w) *out_w = ow;
  return UPF_OK;
}

extern "C" int upf_correlation_forward(const void* in1, const void* in2, void* out, int B, int C, int H, int W, int dtype,
                                       int pad_size, int kernel_size, int max_displacement, int stride1, int stride2,
                                       int corr_type_multiply, void* stream) {
  using namespace upf;
  (void)corr_type_multiply;   // accepted and unused, like correlation_cuda_kernel.cu:302-393
  if (pad_size == 4 && kernel_size == 1 && max_displacement == 4 && stride1 == 1 && stride2 == 1)
    return upf_corr81_forward(in1, in2, out, B, C, H, W, dtype, 0, 0.f, stream);
  int oc, oh, ow;
  int rc = upf_correlation_out_shape(H, W, pad_size, kernel_size, max_displacement, stride1, stride2, &oc, &oh, &ow);
  if (rc != UPF_OK) return rc;
  UPF_REQUIRE(in1 && in2 && out && B > 0 && C > 0, UPF_EINVAL, "correlation_forward: bad arguments");
  const long long total = (long long)B * oc * oh * ow;
  const int threads = 256;
  const int blocks = (int)((total + threads - 1) / threads > 65535 * 16 ? 65535 * 16 : (total + threads - 1) / threads);
  const int kr = (kernel_size - 1) / 2, dr = max_displacement / stride2;
  UPF_DISPATCH(dtype, T,
               hipLaunchKernelGGL((corr::corr_general_kernel<T>), dim3(blocks), dim3(threads), 0, (hipStream_t)stream,
                                  (const T*)in1, (const T*)in2, (T*)out, B, C, H, W, pad_size, kr, max_displacement,
                                  stride1, stride2, dr, oh, ow, kernel_size));
  return check_launch("correlation_forward");
}
