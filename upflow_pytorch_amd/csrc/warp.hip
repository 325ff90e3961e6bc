// Backward flow warp (bilinear grid_sample + validity mask), forward and backward — gfx950.
//
// Replaces WarpingLayer_no_div.forward (/root/reference/model/pwc_modules.py:184-207: meshgrid built
// on the CPU and copied to the device every call, two grid_sample launches, a materialised ones
// tensor, a compare and a multiply) and tools.torch_warp (utils/tools.py:1274-1319) with ONE launch:
// the sampling position, the four weights and the mask bit are computed once per pixel in registers
// and reused for every channel.
//
// HBM-bound: algorithmic bytes = B*H*W*(2*s*C + 8) (x read once through the taps, y written once,
// flow read as fp32).  A thread owns one pixel and CPT channels; lanes of a wave are consecutive in
// x, so the y stores are fully coalesced and the four tap loads of neighbouring lanes fall into the
// same or adjacent cache lines for smooth flows.
//
// Finite-input assumption: a clamped pair load may multiply a column that is not a tap by an exact zero weight, so a
// non-finite FEATURE value next to an image-border tap can turn that output into NaN (ATen multiplies its in-bounds
// zero-weight taps too; only which neighbour is touched differs).  Non-finite FLOWS are handled (positions are clamped).
// Compiled with -ffp-contract=off (see sampling.hpp: the mask is bit-sensitive).
#include "sampling.hpp"

namespace upf {
namespace warp {

constexpr int THREADS = 256;

// Per-pixel sampling state, computed once and reused for every channel.  The two horizontally adjacent
// taps of a row are fetched with ONE load of two elements ("pair"): pc = clamp(x0, 0, W-2) is the pair's
// first column and (wl, wh) are the bilinear weights of columns pc and pc+1 (zero where the tap lies
// outside the image), so a clamped pair needs no per-channel fix-up.  Halving the number of gather
// instructions matters: the kernel is bound by the texture-address pipe (64 distinct addresses per
// instruction), not by bytes.  2-byte-aligned 4-byte loads are legal on gfx950 (tools/unaligned_probe.hip).
struct PixelTaps {
  int oT, oB;                 // element offsets of the top / bottom pair inside a channel plane
  float wlT, whT, wlB, whB;   // weights of (pc, pc+1) in the top and bottom row
  bool valid;
};

// xp: row pitch of x in elements (>= W; = W for a contiguous plane)
__device__ __forceinline__ PixelTaps make_pixel(const float* __restrict__ flow_n, int i, int j, int H, int W, int xp, int mask_mode, SampleGeom sg) {
  PixelTaps r;
  const int HW = H * W, p = i * W + j;
  const float fx = flow_n[p], fy = flow_n[HW + p];
  const Taps t = make_taps(j, i, fx, fy, H, W, sg);
  r.valid = taps_valid(t, mask_mode, j, i, fx, fy, H, W);
  const int pc = min(max(t.x0, 0), W - 2);
  const int ya = min(max(t.y0, 0), H - 1), yb = min(max(t.y0 + 1, 0), H - 1);
  r.oT = ya * xp + pc;
  r.oB = yb * xp + pc;
  // weight of image column c for the (x0, x0+1) taps of a row whose taps have weights (w_a, w_b)
  auto colw = [&](int c, float w_a, bool in_a, float w_b, bool in_b) {
    return (c == t.x0 && in_a) ? w_a : ((c == t.x0 + 1 && in_b) ? w_b : 0.f);
  };
  r.wlT = colw(pc, t.w[0], t.in[0], t.w[1], t.in[1]);
  r.whT = colw(pc + 1, t.w[0], t.in[0], t.w[1], t.in[1]);
  r.wlB = colw(pc, t.w[2], t.in[2], t.w[3], t.in[3]);
  r.whB = colw(pc + 1, t.w[2], t.in[2], t.w[3], t.in[3]);
  return r;
}

template <typename T> struct Pair;
template <> struct Pair<float> {
  static __device__ __forceinline__ void load(const float* p, float& lo, float& hi) {
    const float2 v = *reinterpret_cast<const float2*>(p); lo = v.x; hi = v.y;       // 8-byte load, 4-byte aligned
  }
};
template <> struct Pair<bf16_t> {
  static __device__ __forceinline__ void load(const bf16_t* p, float& lo, float& hi) {
    const uint32_t v = *reinterpret_cast<const uint32_t*>(p);                       // 4-byte load, 2-byte aligned
    lo = __uint_as_float(v << 16); hi = __uint_as_float(v & 0xffff0000u);
  }
};
template <> struct Pair<f16_t> {
  static __device__ __forceinline__ void load(const f16_t* p, float& lo, float& hi) {
    const uint32_t v = *reinterpret_cast<const uint32_t*>(p);
    lo = f16_bits_to_f32(v & 0xffffu); hi = f16_bits_to_f32(v >> 16);
  }
};

// same association as ATen: ((nw*w + ne*w) + sw*w) + se*w   (terms of out-of-image taps are exact zeros)
template <typename T>
__device__ __forceinline__ float sample(const T* __restrict__ plane, const PixelTaps& s) {
  float a, b, c, d;
  Pair<T>::load(plane + s.oT, a, b);
  Pair<T>::load(plane + s.oB, c, d);
  return ((a * s.wlT + b * s.whT) + c * s.wlB) + d * s.whB;
}

// PXT consecutive pixels of a row per thread, W >= 2: 4 when rows keep pixel quads aligned (8-byte stores for 16-bit
// features, 16-byte for fp32; 4x the gathers in flight per wave), 2 for 16-bit types with even rows, else 1.
// Round 5: rows of x / y are `xp` / `yp` elements apart (the flow is a contiguous fp32 tensor).  A thread owns the pixels
// [PXT * g, PXT * g + PXT) of ONE row (g < ceil(W / PXT)); with a ragged W and a pitch that covers the rounded-up row the
// pixels at or beyond W are computed as zeros and stored into the row's own pitch padding — so odd-width pyramid levels
// (311, 39 ...) keep the 4-byte stores.  For W % PXT == 0 and yp = W this is the flat mapping of rounds 1-4.
template <typename T, int PXT>
__global__ __launch_bounds__(THREADS)
void warp_fwd_kernel(const T* __restrict__ x, long long xbs, const float* __restrict__ flow, T* __restrict__ y, long long ybs,
                     int C, int H, int W, int cpt, int mask_mode, int shift, SampleGeom sg, int xp, int yp) {
  const int Wg = (W + PXT - 1) / PXT;
  const int g0 = blockIdx.x * THREADS + threadIdx.x;
  if (g0 >= H * Wg) return;
  const int i = g0 / Wg, j0 = (g0 - i * Wg) * PXT;
  const int n = blockIdx.z;
  const int ns = (n + shift) % (int)gridDim.z;          // source image (batch_shift: sample the OTHER frame of a stacked pair)
  const int c_begin = blockIdx.y * cpt, c_end = min(C, c_begin + cpt);
  const float* fl = flow + (size_t)n * 2 * H * W;
  PixelTaps s[PXT];
#pragma unroll
  for (int k = 0; k < PXT; ++k) {
    if (PXT == 1 || j0 + k < W) s[k] = make_pixel(fl, i, j0 + k, H, W, xp, mask_mode, sg);
    else { s[k].valid = false; s[k].oT = s[k].oB = 0; s[k].wlT = s[k].whT = s[k].wlB = s[k].whB = 0.f; }
  }
  const size_t xplane = (size_t)H * xp, yplane = (size_t)H * yp;
  const T* xb = x + (size_t)ns * xbs + (size_t)c_begin * xplane;          // x / y may be channel slices of wider buffers
  T* yb = y + (size_t)n * ybs + (size_t)c_begin * yplane + (size_t)i * yp + j0;
#pragma unroll 4
  for (int c = c_begin; c < c_end; ++c, xb += xplane, yb += yplane) {
    float r[PXT];
#pragma unroll
    for (int k = 0; k < PXT; ++k) r[k] = s[k].valid ? sample<T>(xb, s[k]) : 0.f;
    if constexpr (PXT == 4 && sizeof(typename Elem<T>::store_t) == 2) {
      *reinterpret_cast<uint2*>(yb) = make_uint2(pack2<T>(r[0], r[1]), pack2<T>(r[2], r[3]));
    } else if constexpr (PXT == 4) {
      *reinterpret_cast<float4*>(yb) = make_float4(r[0], r[1], r[2], r[3]);
    } else if constexpr (PXT == 2 && sizeof(typename Elem<T>::store_t) == 2) {
      *reinterpret_cast<uint32_t*>(yb) = pack2<T>(r[0], r[1]);
    } else {
#pragma unroll
      for (int k = 0; k < PXT; ++k) Elem<T>::store(yb + k, r[k]);
    }
  }
}

// W == 1 (no pair exists): plain four-tap version
template <typename T>
__global__ __launch_bounds__(THREADS)
void warp_fwd_narrow_kernel(const T* __restrict__ x, long long xbs, const float* __restrict__ flow, T* __restrict__ y, long long ybs,
                            int C, int H, int W, int mask_mode, int shift, SampleGeom sg) {
  const int HW = H * W;
  const int p = blockIdx.x * THREADS + threadIdx.x;
  if (p >= HW) return;
  const int n = blockIdx.z;
  const int ns = (n + shift) % (int)gridDim.z;
  const int i = p / W, j = p - i * W;
  const float fx = flow[((size_t)n * 2 + 0) * HW + p], fy = flow[((size_t)n * 2 + 1) * HW + p];
  const Taps t = make_taps(j, i, fx, fy, H, W, sg);
  const bool valid = taps_valid(t, mask_mode, j, i, fx, fy, H, W);
  const int xa = min(max(t.x0, 0), W - 1), xb1 = min(max(t.x0 + 1, 0), W - 1);
  const int ya = min(max(t.y0, 0), H - 1), yb1 = min(max(t.y0 + 1, 0), H - 1);
  const int o0 = ya * W + xa, o1 = ya * W + xb1, o2 = yb1 * W + xa, o3 = yb1 * W + xb1;
  const float w0 = t.in[0] ? t.w[0] : 0.f, w1 = t.in[1] ? t.w[1] : 0.f, w2 = t.in[2] ? t.w[2] : 0.f, w3 = t.in[3] ? t.w[3] : 0.f;
  for (int c = 0; c < C; ++c) {
    const T* xc = x + (size_t)ns * xbs + (size_t)c * HW;
    const float r = ((Elem<T>::load(xc + o0) * w0 + Elem<T>::load(xc + o1) * w1) + Elem<T>::load(xc + o2) * w2) + Elem<T>::load(xc + o3) * w3;
    Elem<T>::store(y + (size_t)n * ybs + (size_t)c * HW + p, valid ? r : 0.f);
  }
}

#ifndef UPF_WARP_MERGE
#define UPF_WARP_MERGE 1                                 // (0: one atomic per tap — tools/warp_bwd_ab.py compares the two builds bit for bit)
#endif
// Backward: one thread per OUTPUT pixel, loops over its channel slice and scatters into the source image:
//   gx[tap] += w_tap * gy            gflow += gy * d(sample)/d(pos)
// A scatter because the inverse map (which output pixels sample a given source pixel) is unbounded for arbitrary flows.
// DETERMINISTIC all the same: contributions are accumulated as 64-bit FIXED-POINT integers (value * 2^44, native 64-bit
// integer atomics) — integer addition is associative, so the sum does not depend on the order in which the workgroups
// arrive, unlike the fp32 atomicAdd of the first version.  Quantum 2^-44 = 5.7e-14 (below fp32 resolution for every
// gradient >= 1e-6, absolute error 5.7e-14 below that), range +-2^19; a second launch converts to the output type.
//
// Round 4: HALF the atomics.  With a smooth flow the lanes of a wave (consecutive pixels of a row) sample consecutive source
// pixels: the right-hand taps of pixel j are the left-hand taps of pixel j + 1.  Each lane passes the fixed-point values of its
// two right-hand contributions up to the next lane (two 64-bit lane shifts per channel), which adds them to its own left-hand
// ones where the ADDRESSES agree (checked per lane pair, once per pixel; anything else — row ends, a flow discontinuity, taps
// outside the image — keeps its own atomic).  The merge is an integer addition of the values the two atomics would have added,
// so the accumulators, and with them every output bit, are unchanged (tests/test_hip_ops.py::test_warp_*); the kernel was bound
// by the L2's atomic rate (13.6 M 64-bit atomics = 50 us at the 1/4 level of config 3).
template <typename T>
__global__ __launch_bounds__(THREADS)
void warp_bwd_kernel(const T* __restrict__ x, const float* __restrict__ flow, const T* __restrict__ gy,
                     unsigned long long* __restrict__ gx64, unsigned long long* __restrict__ gf64, float* __restrict__ gflow,
                     int C, int H, int W, int cpt, int mask_mode, int shift, SampleGeom sg) {
  const int HW = H * W;
  const int p_raw = blockIdx.x * THREADS + threadIdx.x;
  const bool inside = p_raw < HW;
  const int p = inside ? p_raw : HW - 1;               // (no early exits: every lane takes part in the lane shifts)
  const int lane = threadIdx.x & 63;
  const int n = blockIdx.z;
  const int ns = (n + shift) % (int)gridDim.z;
  const int c_begin = blockIdx.y * cpt, c_end = min(C, c_begin + cpt);
  const int i = p / W, j = p - i * W;
  const float fx = flow[((size_t)n * 2 + 0) * HW + p];
  const float fy = flow[((size_t)n * 2 + 1) * HW + p];
  const Taps t = make_taps(j, i, fx, fy, H, W, sg);
  float* gf = gflow + (size_t)n * 2 * HW + p;
  const bool valid = inside && taps_valid(t, mask_mode, j, i, fx, fy, H, W);   // the mask is a constant factor: zero gradients
  if (inside && !valid && gridDim.y == 1) { gf[0] = 0.f; gf[HW] = 0.f; }
  const int xa = min(max(t.x0, 0), W - 1), xb1 = min(max(t.x0 + 1, 0), W - 1);
  const int ya = min(max(t.y0, 0), H - 1), yb1 = min(max(t.y0 + 1, 0), H - 1);
  const int o0 = ya * W + xa, o1 = ya * W + xb1, o2 = yb1 * W + xa, o3 = yb1 * W + xb1;
  const bool in0 = valid && t.in[0], in1 = valid && t.in[1], in2 = valid && t.in[2], in3 = valid && t.in[3];
  // my right-hand taps join the next lane's left-hand ones (m1, m3); the previous lane's join mine (p1, p3)
  const int o0n = __shfl_down(o0, 1), o2n = __shfl_down(o2, 1);
  const int in0n = __shfl_down((int)in0, 1), in2n = __shfl_down((int)in2, 1);
  const bool m1 = UPF_WARP_MERGE && lane < 63 && in1 && in0n && o1 == o0n;
  const bool m3 = UPF_WARP_MERGE && lane < 63 && in3 && in2n && o3 == o2n;
  const bool p1 = __shfl_up((int)m1, 1) && lane > 0, p3 = __shfl_up((int)m3, 1) && lane > 0;
  // d w / d ix, d w / d iy  for nw, ne, sw, se
  const float ax = (float)(t.x0 + 1) - t.ix, bx = t.ix - (float)t.x0;
  const float ay = (float)(t.y0 + 1) - t.iy, by = t.iy - (float)t.y0;
  const T* xb = x + ((size_t)ns * C + c_begin) * HW;
  const T* gb = gy + ((size_t)n * C + c_begin) * HW + p;
  unsigned long long* gxb = gx64 + ((size_t)ns * C + c_begin) * HW;
  float gix = 0.f, giy = 0.f;
  for (int c = c_begin; c < c_end; ++c, xb += HW, gb += HW, gxb += HW) {      // (uniform bounds)
    const float g = valid ? Elem<T>::load(gb) : 0.f;
    long long q0 = 0, q1 = 0, q2 = 0, q3 = 0;
    if (in0) { const float v = Elem<T>::load(xb + o0); q0 = fix_q(t.w[0] * g); gix -= v * ay * g; giy -= v * ax * g; }
    if (in1) { const float v = Elem<T>::load(xb + o1); q1 = fix_q(t.w[1] * g); gix += v * ay * g; giy -= v * bx * g; }
    if (in2) { const float v = Elem<T>::load(xb + o2); q2 = fix_q(t.w[2] * g); gix -= v * by * g; giy += v * ax * g; }
    if (in3) { const float v = Elem<T>::load(xb + o3); q3 = fix_q(t.w[3] * g); gix += v * by * g; giy += v * bx * g; }
    const long long r1 = __shfl_up(q1, 1), r3 = __shfl_up(q3, 1);
    if (in0) fix_emit(gxb + o0, q0, p1 ? r1 : 0ll);
    if (in1 && !m1) fix_emit(gxb + o1, q1, 0ll);
    if (in2) fix_emit(gxb + o2, q2, p3 ? r3 : 0ll);
    if (in3 && !m3) fix_emit(gxb + o3, q3, 0ll);
  }
  if (!valid) return;
  // chain rule through un-normalise ((W-1)/2) and the python-side normalise (2/max(W-1,1))
  const float mx = ((float)(W - 1) * 0.5f) * (2.0f / (float)max(W - 1, 1));
  const float my = ((float)(H - 1) * 0.5f) * (2.0f / (float)max(H - 1, 1));
  if (gridDim.y == 1) { gf[0] = gix * mx; gf[HW] = giy * my; }
  else {                                                 // channel range split over blockIdx.y: fixed-point partial sums
    unsigned long long* q = gf64 + (size_t)n * 2 * HW + p;
    fix_add(q, gix * mx); fix_add(q + HW, giy * my);
  }
}

// fixed point -> output type (gx in the feature dtype; gflow fp32 when the channel range was split)
template <typename T>
__global__ void warp_bwd_finish_kernel(const unsigned long long* __restrict__ gx64, T* __restrict__ gx, long long n_gx,
                                       const unsigned long long* __restrict__ gf64, float* __restrict__ gflow, long long n_gf) {
  const long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x;
  if (i < n_gx) Elem<T>::store(gx + i, fix_get(gx64[i]));
  if (gf64 && i < n_gf) gflow[i] = fix_get(gf64[i]);
}

static int pick_cpt(int B, int C, int HW) {
  // split channels over blockIdx.y until there are ~4 workgroups per CU, but keep >= 8 channels per
  // thread: the per-pixel set-up (two IEEE divisions, weights, mask) costs about as much as 6 channels
  const long long pix_blocks = (long long)B * cdiv(HW, THREADS);
  int split = 1;
  while (split < C && pix_blocks * split < 1024 && C / (split * 2) >= 8) split *= 2;
  return cdiv(C, split);
}


// ---- the same warp on CHANNEL-OCTET tensors ([n][c/8][H][W][8], include/upflow_hip.h "C8"): the SGU stack's input is a C8
// buffer whose first half (feature_1) a convolution wrote as octets; the warped other frame goes straight into its second half.
// One thread per pixel: position, weights and mask once, then per octet four 16-byte gathers (8 channels of a tap each — the
// NCHW form needs 2 x 4-byte gathers per CHANNEL) and one 16-byte store.  Same arithmetic per channel as sample():
// ((nw*w0 + ne*w1) + sw*w2) + se*w3 with exact-zero weights for taps outside the image, so the values equal the NCHW kernel's.
template <typename T> __device__ __forceinline__ void unpack2(uint32_t v, float& lo, float& hi);
template <> __device__ __forceinline__ void unpack2<bf16_t>(uint32_t v, float& lo, float& hi) { lo = __uint_as_float(v << 16); hi = __uint_as_float(v & 0xffff0000u); }
template <> __device__ __forceinline__ void unpack2<f16_t>(uint32_t v, float& lo, float& hi) { lo = f16_bits_to_f32(v & 0xffffu); hi = f16_bits_to_f32(v >> 16); }

template <typename T>
__global__ __launch_bounds__(THREADS)
void warp_c8_kernel(const T* __restrict__ x, long long xbs, const float* __restrict__ flow, T* __restrict__ y, long long ybs,
                    int noct, int H, int W, int mask_mode, int shift, SampleGeom sg) {
  const int HW = H * W;
  const int p = blockIdx.x * THREADS + threadIdx.x;
  if (p >= HW) return;
  const int n = blockIdx.z;
  const int ns = (n + shift) % (int)gridDim.z;
  const int i = p / W, j = p - i * W;
  const float fx = flow[((size_t)n * 2 + 0) * HW + p], fy = flow[((size_t)n * 2 + 1) * HW + p];
  const Taps t = make_taps(j, i, fx, fy, H, W, sg);
  const bool valid = taps_valid(t, mask_mode, j, i, fx, fy, H, W);
  const int xa = min(max(t.x0, 0), W - 1), xb1 = min(max(t.x0 + 1, 0), W - 1);
  const int ya = min(max(t.y0, 0), H - 1), yb1 = min(max(t.y0 + 1, 0), H - 1);
  const int o0 = ya * W + xa, o1 = ya * W + xb1, o2 = yb1 * W + xa, o3 = yb1 * W + xb1;
  const float w0 = t.in[0] ? t.w[0] : 0.f, w1 = t.in[1] ? t.w[1] : 0.f, w2 = t.in[2] ? t.w[2] : 0.f, w3 = t.in[3] ? t.w[3] : 0.f;
  const uint4* xo = reinterpret_cast<const uint4*>(x + (size_t)ns * xbs);
  uint4* yo = reinterpret_cast<uint4*>(y + (size_t)n * ybs) + (size_t)blockIdx.y * HW + p;
  for (int g = blockIdx.y; g < noct; g += gridDim.y, yo += (size_t)gridDim.y * HW) {
    uint4 out = make_uint4(0u, 0u, 0u, 0u);
    if (valid) {
      const uint4* xp = xo + (size_t)g * HW;
      const uint4 a = xp[o0], b = xp[o1], c = xp[o2], d = xp[o3];
      const uint32_t av[4] = {a.x, a.y, a.z, a.w}, bv[4] = {b.x, b.y, b.z, b.w}, cv[4] = {c.x, c.y, c.z, c.w}, dv[4] = {d.x, d.y, d.z, d.w};
      uint32_t r[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        float lo[4], hi[4];
        unpack2<T>(av[q], lo[0], hi[0]); unpack2<T>(bv[q], lo[1], hi[1]); unpack2<T>(cv[q], lo[2], hi[2]); unpack2<T>(dv[q], lo[3], hi[3]);
        r[q] = pack2<T>(((lo[0] * w0 + lo[1] * w1) + lo[2] * w2) + lo[3] * w3, ((hi[0] * w0 + hi[1] * w1) + hi[2] * w2) + hi[3] * w3);
      }
      out = make_uint4(r[0], r[1], r[2], r[3]);
    }
    *yo = out;
  }
}

}  // namespace warp
}  // namespace upf

extern "C" int upf_warp_forward_pitched(const void* x, long long x_batch_stride, int x_row_pitch, const float* flow, void* y, long long y_batch_stride,
                                        int y_row_pitch, int B, int C, int H, int W, int dtype, int mask_mode, int batch_shift, void* stream) {
  using namespace upf;
  UPF_REQUIRE(x && flow && y, UPF_EINVAL, "warp_forward: null pointer");
  UPF_REQUIRE(B > 0 && C > 0 && H > 0 && W > 0 && B <= 65535, UPF_EINVAL, "warp_forward: bad shape B=%d C=%d H=%d W=%d", B, C, H, W);
  UPF_REQUIRE(mask_mode >= UPF_MASK_NONE && mask_mode <= UPF_MASK_ROBUST, UPF_EINVAL, "warp_forward: bad mask_mode %d", mask_mode);
  UPF_REQUIRE(batch_shift >= 0 && batch_shift < B, UPF_EINVAL, "warp_forward: batch_shift %d not in [0,%d)", batch_shift, B);
  const int xp = x_row_pitch ? x_row_pitch : W, yp = y_row_pitch ? y_row_pitch : W;
  UPF_REQUIRE(xp >= W && yp >= W, UPF_EINVAL, "warp_forward: row pitch smaller than W (%d, %d < %d)", xp, yp, W);
  const int HW = H * W;
  const long long xbs = x_batch_stride ? x_batch_stride : (long long)C * H * xp, ybs = y_batch_stride ? y_batch_stride : (long long)C * H * yp;
  UPF_REQUIRE(xbs >= (long long)C * H * xp - (xp - W) && ybs >= (long long)C * H * yp - (yp - W), UPF_EINVAL, "warp_forward: batch stride smaller than C*H*pitch");
  hipStream_t st = (hipStream_t)stream;
  const SampleGeom sg = make_sample_geom(H, W);
  if (W < 2) {
    UPF_REQUIRE(xp == W && yp == W, UPF_EUNSUPPORTED, "warp_forward: W < 2 with pitched rows");
    UPF_DISPATCH(dtype, T, hipLaunchKernelGGL((warp::warp_fwd_narrow_kernel<T>), dim3(cdiv(HW, warp::THREADS), 1, B), dim3(warp::THREADS), 0, st,
                                              (const T*)x, xbs, flow, (T*)y, ybs, C, H, W, mask_mode, batch_shift, sg));
    return check_launch("warp_forward");
  }
  // (4 consecutive pixels per thread with 8 / 16-byte stores were measured SLOWER for 16-bit features, 12.6 -> 16.2 us at
  //  [4,32,96,320]: every gather instruction then spans 4x the cache lines; fp32 gained 12 %: kept for fp32 only)
  // pixel pairs / quads: the OUTPUT rows must keep them aligned and hold the rounded-up row (W % n == 0, or pitch padding behind it)
  const bool four = (dtype == UPF_F32) && (yp % 4 == 0) && (W % 4 == 0) && aligned_to(y, 16) && ybs % 4 == 0 && (long long)B * HW >= 4 * 64 * 256;
  const bool two = !four && (dtype != UPF_F32) && (yp % 2 == 0) && aligned_to(y, 4) && ybs % 2 == 0;     // (yp even => yp >= W rounded up to 2)
  const int pxt = four ? 4 : (two ? 2 : 1);
  const int ngroups = H * cdiv(W, pxt);
  const int cpt = warp::pick_cpt(B, C, ngroups);
  dim3 grid(cdiv(ngroups, warp::THREADS), cdiv(C, cpt), B);
  UPF_DISPATCH(dtype, T,
               if (four) hipLaunchKernelGGL((warp::warp_fwd_kernel<T, 4>), grid, dim3(warp::THREADS), 0, st, (const T*)x, xbs, flow, (T*)y, ybs, C, H, W, cpt, mask_mode, batch_shift, sg, xp, yp);
               else if (two) hipLaunchKernelGGL((warp::warp_fwd_kernel<T, 2>), grid, dim3(warp::THREADS), 0, st, (const T*)x, xbs, flow, (T*)y, ybs, C, H, W, cpt, mask_mode, batch_shift, sg, xp, yp);
               else hipLaunchKernelGGL((warp::warp_fwd_kernel<T, 1>), grid, dim3(warp::THREADS), 0, st, (const T*)x, xbs, flow, (T*)y, ybs, C, H, W, cpt, mask_mode, batch_shift, sg, xp, yp));
  return check_launch("warp_forward");
}

extern "C" int upf_warp_forward_strided(const void* x, long long x_batch_stride, const float* flow, void* y, long long y_batch_stride,
                                        int B, int C, int H, int W, int dtype, int mask_mode, int batch_shift, void* stream) {
  return upf_warp_forward_pitched(x, x_batch_stride, 0, flow, y, y_batch_stride, 0, B, C, H, W, dtype, mask_mode, batch_shift, stream);
}

extern "C" int upf_warp_forward(const void* x, const float* flow, void* y, int B, int C, int H, int W,
                                int dtype, int mask_mode, int batch_shift, void* stream) {
  return upf_warp_forward_strided(x, 0, flow, y, 0, B, C, H, W, dtype, mask_mode, batch_shift, stream);
}

extern "C" long long upf_warp_backward_workspace_bytes(int B, int C, int H, int W) {
  return ((long long)B * C * H * W + (long long)B * 2 * H * W) * (long long)sizeof(unsigned long long);
}

extern "C" int upf_warp_backward(const void* x, const float* flow, const void* grad_y, void* gx, float* gflow, void* workspace,
                                 int B, int C, int H, int W, int dtype, int mask_mode, int batch_shift, void* stream) {
  using namespace upf;
  UPF_REQUIRE(x && flow && grad_y && gx && gflow && workspace, UPF_EINVAL, "warp_backward: null pointer");
  UPF_REQUIRE(B > 0 && C > 0 && H > 0 && W > 0 && B <= 65535, UPF_EINVAL, "warp_backward: bad shape");
  UPF_REQUIRE(mask_mode >= UPF_MASK_NONE && mask_mode <= UPF_MASK_ROBUST, UPF_EINVAL, "warp_backward: bad mask_mode %d", mask_mode);
  UPF_REQUIRE(batch_shift >= 0 && batch_shift < B, UPF_EINVAL, "warp_backward: batch_shift %d not in [0,%d)", batch_shift, B);
  const int HW = H * W;
  hipStream_t s = (hipStream_t)stream;
  const long long n_gx = (long long)B * C * HW, n_gf = (long long)B * 2 * HW;
  unsigned long long* gx64 = (unsigned long long*)workspace;
  unsigned long long* gf64 = gx64 + n_gx;
  const int cpt = warp::pick_cpt(B, C, HW);
  const bool split = cdiv(C, cpt) > 1;
  { const int zrc = zero_fill_u64(gx64, n_gx + (split ? n_gf : 0), s); if (zrc) return zrc; }      // (a kernel, not a memset node: common.hpp)
  dim3 grid(cdiv(HW, warp::THREADS), cdiv(C, cpt), B);
  const long long n_fin = n_gx > n_gf ? n_gx : n_gf;
  UPF_REQUIRE((n_fin + 255) / 256 < (1ll << 31), UPF_EINVAL, "warp_backward: tensor too large");
  UPF_DISPATCH(dtype, T,
               hipLaunchKernelGGL((warp::warp_bwd_kernel<T>), grid, dim3(warp::THREADS), 0, s,
                                  (const T*)x, flow, (const T*)grad_y, gx64, gf64, gflow, C, H, W, cpt, mask_mode, batch_shift, make_sample_geom(H, W));
               hipLaunchKernelGGL((warp::warp_bwd_finish_kernel<T>), dim3((unsigned)((n_fin + 255) / 256)), dim3(256), 0, s,
                                  gx64, (T*)gx, n_gx, split ? gf64 : nullptr, gflow, n_gf));
  return check_launch("warp_backward");
}

extern "C" int upf_warp_forward_c8(const void* x8, long long x_batch_stride, const float* flow, void* y8, long long y_batch_stride,
                                   int B, int n_oct, int H, int W, int dtype, int mask_mode, int batch_shift, void* stream) {
  using namespace upf;
  UPF_REQUIRE(x8 && flow && y8, UPF_EINVAL, "warp_forward_c8: null pointer");
  UPF_REQUIRE(B > 0 && B <= 65535 && n_oct > 0 && H > 0 && W > 0, UPF_EINVAL, "warp_forward_c8: bad shape B=%d octets=%d H=%d W=%d", B, n_oct, H, W);
  UPF_REQUIRE(dtype == UPF_BF16 || dtype == UPF_F16, UPF_EDTYPE, "warp_forward_c8: bf16 / fp16 only (C8 tensors are 16-bit)");
  UPF_REQUIRE(mask_mode >= UPF_MASK_NONE && mask_mode <= UPF_MASK_ROBUST, UPF_EINVAL, "warp_forward_c8: bad mask_mode %d", mask_mode);
  UPF_REQUIRE(batch_shift >= 0 && batch_shift < B, UPF_EINVAL, "warp_forward_c8: batch_shift %d not in [0,%d)", batch_shift, B);
  UPF_REQUIRE(aligned_to(x8, 16) && aligned_to(y8, 16) && x_batch_stride % 8 == 0 && y_batch_stride % 8 == 0, UPF_EINVAL, "warp_forward_c8: 16-byte aligned octet tensors expected");
  const int HW = H * W;
  const int gy = n_oct < 4 ? n_oct : 4;                 // octets over blockIdx.y (each thread loops over its share)
  dim3 grid(cdiv(HW, warp::THREADS), gy, B);
  const SampleGeom sg = make_sample_geom(H, W);
  if (dtype == UPF_BF16)
    hipLaunchKernelGGL((warp::warp_c8_kernel<bf16_t>), grid, dim3(warp::THREADS), 0, (hipStream_t)stream, (const bf16_t*)x8, x_batch_stride, flow, (bf16_t*)y8, y_batch_stride, n_oct, H, W, mask_mode, batch_shift, sg);
  else
    hipLaunchKernelGGL((warp::warp_c8_kernel<f16_t>), grid, dim3(warp::THREADS), 0, (hipStream_t)stream, (const f16_t*)x8, x_batch_stride, flow, (f16_t*)y8, y_batch_stride, n_oct, H, W, mask_mode, batch_shift, sg);
  return check_launch("warp_forward_c8");
}
