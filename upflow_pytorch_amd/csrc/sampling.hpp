// Bilinear sampling arithmetic shared by the warp, SGU-blend and occlusion kernels.
//
// The reference samples through ATen grid_sample (bilinear, zeros padding) with the torch-1.1
// align_corners=True convention, after normalising meshgrid+flow by (W-1),(H-1):
//   /root/reference/model/pwc_modules.py:187-200, utils/tools.py:1286-1304.
// The validity mask `grid_sample(ones) >= 1.0` (pwc_modules.py:201-206) is rounding-sensitive
// (SURVEY.md §7-H2): it is reproduced bit-exactly only if every step below is a separately rounded
// IEEE fp32 operation in this exact order.  Files including this header are compiled with
// -ffp-contract=off and the mask sum additionally uses __fadd_rn/__fmul_rn.
#pragma once
#include "common.hpp"

namespace upf {

struct Taps {
  float w[4];      // nw, ne, sw, se bilinear weights
  int   x0, y0;    // north-west tap
  bool  in[4];     // tap inside the image
  float ix, iy;    // sampling position (after the normalise/un-normalise round trip)
};

// The two IEEE divisions of a sampling position divide by a launch constant (W-1, H-1): x / d is computed WITHOUT the
// division sequence (~40 instructions each, a third of these kernels' instruction stream in round 2) and still correctly
// rounded.  With r = RN(1/d) from the host (SampleGeom):  q = RN(x*r);  e = fma(-q, d, x) (exact);  q' = fma(e, r, q)
// is RN(x/d) for every x whose quotient is a normal number [Markstein, IBM J. R&D 1990] — checked EXHAUSTIVELY on the
// GPU, all 2^32 bit patterns of x, for every divisor a pyramid level of the BASELINE configurations has
// (tools/div_exact_probe.hip, tests/test_hip_ops.py::test_division_free_quotient_is_exact).  Zeros keep their sign bit's
// effect (x == 0 returns x); magnitudes outside [2^-100, 2^100], infinities and NaNs take the division (a wave-uniformly
// rare branch: sampling positions are pixel coordinates).
struct SampleGeom { float rW, rH; };       // RN(1 / max(W-1, 1)), RN(1 / max(H-1, 1))
static inline SampleGeom make_sample_geom(int H, int W) {
  SampleGeom g;
  g.rW = 1.0f / (float)(W - 1 > 1 ? W - 1 : 1);
  g.rH = 1.0f / (float)(H - 1 > 1 ? H - 1 : 1);
  return g;
}
__device__ __forceinline__ float div_by_const(float x, float d, float r) {
  const float a = fabsf(x);
  if (!(a >= 7.888609e-31f && a <= 1.2676506e30f)) return (x == 0.f) ? x : __fdiv_rn(x, d);     // 2^-100 .. 2^100
  const float q = __fmul_rn(x, r);
  const float e = __builtin_fmaf(-q, d, x);
  return __builtin_fmaf(e, r, q);
}

// position for output pixel (j, i) displaced by (fx, fy)
__device__ __forceinline__ Taps make_taps(int j, int i, float fx, float fy, int H, int W, SampleGeom sg) {
  Taps t;
  const float dW = (float)max(W - 1, 1), dH = (float)max(H - 1, 1);
  // vgrid = 2*(grid+flow)/max(W-1,1) - 1                       pwc_modules.py:195-198
  const float gx = __fsub_rn(div_by_const(__fmul_rn(2.0f, __fadd_rn((float)j, fx)), dW, sg.rW), 1.0f);
  const float gy = __fsub_rn(div_by_const(__fmul_rn(2.0f, __fadd_rn((float)i, fy)), dH, sg.rH), 1.0f);
  // grid_sampler_unnormalize(align_corners=True): ((g+1)/2)*(size-1)
  t.ix = __fmul_rn(__fdiv_rn(__fadd_rn(gx, 1.0f), 2.0f), (float)(W - 1));
  t.iy = __fmul_rn(__fdiv_rn(__fadd_rn(gy, 1.0f), 2.0f), (float)(H - 1));
  const float fx0 = floorf(t.ix), fy0 = floorf(t.iy);
  const float fx1 = __fadd_rn(fx0, 1.0f), fy1 = __fadd_rn(fy0, 1.0f);
  const float ax = __fsub_rn(fx1, t.ix), bx = __fsub_rn(t.ix, fx0);
  const float ay = __fsub_rn(fy1, t.iy), by = __fsub_rn(t.iy, fy0);
  t.w[0] = __fmul_rn(ax, ay);
  t.w[1] = __fmul_rn(bx, ay);
  t.w[2] = __fmul_rn(ax, by);
  t.w[3] = __fmul_rn(bx, by);
  // clamp before the int conversion so that huge/NaN positions stay out of bounds without UB
  const float cx = fminf(fmaxf(fx0, -2.0f), (float)W + 1.0f), cy = fminf(fmaxf(fy0, -2.0f), (float)H + 1.0f);
  t.x0 = (fx0 == fx0) ? (int)cx : -2;
  t.y0 = (fy0 == fy0) ? (int)cy : -2;
  const bool xin0 = t.x0 >= 0 && t.x0 <= W - 1, xin1 = t.x0 + 1 >= 0 && t.x0 + 1 <= W - 1;
  const bool yin0 = t.y0 >= 0 && t.y0 <= H - 1, yin1 = t.y0 + 1 >= 0 && t.y0 + 1 <= H - 1;
  t.in[0] = xin0 && yin0;
  t.in[1] = xin1 && yin0;
  t.in[2] = xin0 && yin1;
  t.in[3] = xin1 && yin1;
  return t;
}

// validity mask of WarpingLayer_no_div
__device__ __forceinline__ bool taps_valid(const Taps& t, int mask_mode, int j, int i, float fx, float fy, int H, int W) {
  if (mask_mode == UPF_MASK_NONE) return true;
  if (mask_mode == UPF_MASK_ROBUST) {
    const float px = __fadd_rn((float)j, fx), py = __fadd_rn((float)i, fy);
    return px >= 0.f && px <= (float)(W - 1) && py >= 0.f && py <= (float)(H - 1);
  }
  // literal: ((nw + ne) + sw) + se of the in-bounds weights, compared with 1.0
  float s = t.in[0] ? t.w[0] : 0.f;
  s = __fadd_rn(s, t.in[1] ? t.w[1] : 0.f);
  s = __fadd_rn(s, t.in[2] ? t.w[2] : 0.f);
  s = __fadd_rn(s, t.in[3] ? t.w[3] : 0.f);
  return s >= 1.0f;
}

// align_corners=True bilinear resize source position (ATen area_pixel_compute_source_index)
struct Lerp { int i0, i1; float l0, l1; };
__device__ __forceinline__ Lerp make_lerp(int dst, int in_size, int out_size) {
  Lerp r;
  const float scale = (out_size > 1) ? (float)(in_size - 1) / (float)(out_size - 1) : 0.f;
  const float src = __fmul_rn(scale, (float)dst);
  r.i0 = min((int)src, in_size - 1);
  r.i1 = min(r.i0 + 1, in_size - 1);
  r.l1 = __fsub_rn(src, (float)r.i0);
  r.l0 = __fsub_rn(1.0f, r.l1);
  return r;
}

// the same with the scale (in_size-1)/(out_size-1) precomputed (host: lerp_scale — the identical IEEE fp32 division)
static inline float lerp_scale(int in_size, int out_size) {
  return (out_size > 1) ? (float)(in_size - 1) / (float)(out_size - 1) : 0.f;
}
__device__ __forceinline__ Lerp make_lerp_scaled(int dst, int in_size, float scale) {
  Lerp r;
  const float src = __fmul_rn(scale, (float)dst);
  r.i0 = min((int)src, in_size - 1);
  r.i1 = min(r.i0 + 1, in_size - 1);
  r.l1 = __fsub_rn(src, (float)r.i0);
  r.l0 = __fsub_rn(1.0f, r.l1);
  return r;
}

}  // namespace upf
