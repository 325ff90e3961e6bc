// Self-guided-upsample interpolation-blend and flow up-sampling — gfx950.
//
// sgu_blend replaces the tail of sgu_model.forward (/root/reference/model/upflow.py:79-88): slice,
// sigmoid, (at the final level) two bilinear up-samplings with flow rescale, a torch_warp of the
// 2-channel flow and the blend — ~8 ATen launches and as many intermediate tensors — with ONE launch
// that reads flow_init and x_out once and writes flow_up once:
//   bytes (decoder level) = B*H*W*(8 + 3*s + 8)      bytes (final level) = B*Hf*Wf*16 + B*h*w*3*s
// flow_upsample replaces upsample2d_flow_as / upsample_flow (model/pwc_modules.py:77-104).
// Compiled with -ffp-contract=off.
#include "sampling.hpp"

namespace upf {
namespace sgu {

constexpr int THREADS = 256;

__device__ __forceinline__ float sigmoidf(float v) { return 1.0f / (1.0f + expf(-v)); }

// inter_flow (2) and inter_mask (1) at output pixel (i, j): direct read at a decoder level,
// align_corners bilinear up-sampling of x_out[:, :2] * ratio and of sigmoid(x_out[:, 2]) at the final
// level (upflow.py:82 applies the sigmoid BEFORE the up-sampling of :86).
template <typename T>
__device__ __forceinline__ void inter_at(const T* __restrict__ xo, int h, int w, int Hf, int Wf, int i, int j,
                                         float& ifx, float& ify, float& m, Lerp& ly, Lerp& lx) {
  const int hw = h * w;
  if (Hf == h && Wf == w) {
    const int q = i * w + j;
    ifx = Elem<T>::load(xo + q);
    ify = Elem<T>::load(xo + hw + q);
    m = sigmoidf(Elem<T>::load(xo + 2 * hw + q));
    return;
  }
  ly = make_lerp(i, h, Hf);
  lx = make_lerp(j, w, Wf);
  const int q00 = ly.i0 * w + lx.i0, q01 = ly.i0 * w + lx.i1, q10 = ly.i1 * w + lx.i0, q11 = ly.i1 * w + lx.i1;
  auto lerp4 = [&](float a, float b, float c, float d) {
    return ly.l0 * (lx.l0 * a + lx.l1 * b) + ly.l1 * (lx.l0 * c + lx.l1 * d);
  };
  const float su = (float)((double)Wf / (double)w), sv = (float)((double)Hf / (double)h);
  ifx = lerp4(Elem<T>::load(xo + q00), Elem<T>::load(xo + q01), Elem<T>::load(xo + q10), Elem<T>::load(xo + q11)) * su;
  ify = lerp4(Elem<T>::load(xo + hw + q00), Elem<T>::load(xo + hw + q01), Elem<T>::load(xo + hw + q10), Elem<T>::load(xo + hw + q11)) * sv;
  m = lerp4(sigmoidf(Elem<T>::load(xo + 2 * hw + q00)), sigmoidf(Elem<T>::load(xo + 2 * hw + q01)),
            sigmoidf(Elem<T>::load(xo + 2 * hw + q10)), sigmoidf(Elem<T>::load(xo + 2 * hw + q11)));
}

// sigmoid of the mask logits at LOW resolution (final level only): every full-resolution pixel interpolates four of them,
// so evaluating the sigmoid once per low-resolution pixel instead of four times per output pixel removes 4 expf + 4
// divisions per output pixel; the interpolated value is bit-identical (same sigmoid, same lerp).
template <typename T>
__global__ void sigmoid_map_kernel(const T* __restrict__ x_out, float* __restrict__ sig, int hw, long long total) {
  const long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x;
  if (i >= total) return;
  const long long n = i / hw;
  sig[i] = sigmoidf(Elem<T>::load(x_out + (size_t)n * 3 * hw + 2 * hw + (i - n * hw)));
}

// One pixel per thread on a 2-D grid: a wave = 64 consecutive pixels of ONE row (coalesced gathers and stores), the four
// waves of a workgroup take four consecutive rows.  Everything that depends only on the sizes (the two bilinear scales,
// the flow rescale factors) is a kernel argument computed once on the host with the same IEEE operations, and the row
// interpolation is wave-uniform — the first version spent most of its ~500 lane-cycles per pixel on per-thread integer
// and double-precision divisions, and on evaluating the sigmoid at 4 taps per output pixel.
// (Measured and NOT kept: 4 pixels per thread, consecutive (each gather instruction spans 4x the cache lines, 1.6x slower)
//  or strided by HW/4 (1.1x slower: the kernel is instruction-bound, not latency-bound).)
template <typename T>
__global__ __launch_bounds__(THREADS)
void blend_fwd_kernel(const float* __restrict__ flow_init, const T* __restrict__ x_out, const float* __restrict__ sig,
                      float* __restrict__ flow_up, float* __restrict__ inter_flow, float* __restrict__ inter_mask,
                      int h, int w, int Hf, int Wf, float scale_y, float scale_x, float su, float sv, SampleGeom geo,
                      T* __restrict__ flow16 = nullptr, long long f16bs = 0, int f16_c8 = 0) {
  const int HW = Hf * Wf, hw = h * w;
  const int j = blockIdx.x * 64 + (threadIdx.x & 63);
  const int i = __builtin_amdgcn_readfirstlane(blockIdx.y * 4 + (threadIdx.x >> 6));
  if (i >= Hf || j >= Wf) return;
  const int n = blockIdx.z;
  const int p = i * Wf + j;
  const T* xo = x_out + (size_t)n * 3 * hw;
  const float* f0 = flow_init + (size_t)n * 2 * HW;
  float ifx, ify, m;
  if (Hf == h && Wf == w) {
    ifx = Elem<T>::load(xo + p);
    ify = Elem<T>::load(xo + hw + p);
    m = sigmoidf(Elem<T>::load(xo + 2 * hw + p));
  } else {
    const Lerp ly = make_lerp_scaled(i, h, scale_y), lx = make_lerp_scaled(j, w, scale_x);
    const int q00 = ly.i0 * w + lx.i0, q01 = ly.i0 * w + lx.i1, q10 = ly.i1 * w + lx.i0, q11 = ly.i1 * w + lx.i1;
    auto lerp4 = [&](float a, float b, float c, float d) {
      return ly.l0 * (lx.l0 * a + lx.l1 * b) + ly.l1 * (lx.l0 * c + lx.l1 * d);
    };
    ifx = lerp4(Elem<T>::load(xo + q00), Elem<T>::load(xo + q01), Elem<T>::load(xo + q10), Elem<T>::load(xo + q11)) * su;
    ify = lerp4(Elem<T>::load(xo + hw + q00), Elem<T>::load(xo + hw + q01), Elem<T>::load(xo + hw + q10), Elem<T>::load(xo + hw + q11)) * sv;
    const float* sg = sig + (size_t)n * hw;                 // sigmoid evaluated once per low-resolution pixel
    m = lerp4(sg[q00], sg[q01], sg[q10], sg[q11]);
  }
  const Taps t = make_taps(j, i, ifx, ify, Hf, Wf, geo);
  const int xa = min(max(t.x0, 0), Wf - 1), xb = min(max(t.x0 + 1, 0), Wf - 1);
  const int ya = min(max(t.y0, 0), Hf - 1), yb = min(max(t.y0 + 1, 0), Hf - 1);
  const int o0 = ya * Wf + xa, o1 = ya * Wf + xb, o2 = yb * Wf + xa, o3 = yb * Wf + xb;
  const float w0 = t.in[0] ? t.w[0] : 0.f, w1 = t.in[1] ? t.w[1] : 0.f;
  const float w2 = t.in[2] ? t.w[2] : 0.f, w3 = t.in[3] ? t.w[3] : 0.f;
  const float om = 1.0f - m;
  float fu[2];
#pragma unroll
  for (int c = 0; c < 2; ++c) {
    const float* f = f0 + c * HW;
    const float warped = ((f[o0] * w0 + f[o1] * w1) + f[o2] * w2) + f[o3] * w3;
    fu[c] = warped * om + f[p] * m;                                     // upflow.py:88
    flow_up[((size_t)n * 2 + c) * HW + p] = fu[c];
  }
  if (flow16) {
    // (round 6) the blended flow rounded to the decoder's 16-bit type straight into the flow estimator's input buffer — an octet entry
    // [u, v, 0 x 6] or two NCHW planes — what upf_flow_update(_c8)(flow_up) wrote in a launch of its own (the same rounding of the same value)
    if (f16_c8) {
      *reinterpret_cast<uint4*>(flow16 + (size_t)n * f16bs + (size_t)p * 8) = make_uint4(pack2<T>(fu[0], fu[1]), 0u, 0u, 0u);
    } else {
      Elem<T>::store(flow16 + (size_t)n * f16bs + p, fu[0]);
      Elem<T>::store(flow16 + (size_t)n * f16bs + HW + p, fu[1]);
    }
  }
  if (inter_flow) {
    inter_flow[((size_t)n * 2 + 0) * HW + p] = ifx;
    inter_flow[((size_t)n * 2 + 1) * HW + p] = ify;
  }
  if (inter_mask) inter_mask[(size_t)n * HW + p] = m;
}

// Backward of the blend.  g = grad of flow_up (2 channels).
//   d/d flow_init : m*g at p, plus (1-m)*g*w_tap scattered to the 4 taps
//   d/d inter_flow: (1-m) * sum_c g_c * d(warp_c)/d(pos)
//   d/d m         : sum_c g_c * (flow_init_c(p) - warped_c);  d/d logit through sigmoid' (and through
//                   the bilinear up-sampling weights at the final level: a scatter into the low-resolution map)
// Both scatters accumulate in 64-bit fixed point (common.hpp: fix_add) => bit-reproducible; a second launch converts.
template <typename T>
__global__ __launch_bounds__(THREADS)
void blend_bwd_kernel(const float* __restrict__ flow_init, const T* __restrict__ x_out, const float* __restrict__ g_up,
                      unsigned long long* __restrict__ g_init, float* __restrict__ g_full, float* __restrict__ g_xo,
                      int h, int w, int Hf, int Wf, SampleGeom geo) {
  const int HW = Hf * Wf, hw = h * w;
  const int p_raw = blockIdx.x * THREADS + threadIdx.x;
  const bool inside = p_raw < HW;
  const int p = inside ? p_raw : HW - 1;              // (no early exits: every lane takes part in the lane shifts below)
  const int lane = threadIdx.x & 63;
  const int n = blockIdx.y;
  const int i = p / Wf, j = p - i * Wf;
  float ifx, ify, m;
  Lerp ly, lx;
  const T* xo = x_out + (size_t)n * 3 * hw;
  inter_at<T>(xo, h, w, Hf, Wf, i, j, ifx, ify, m, ly, lx);
  const float* f0 = flow_init + (size_t)n * 2 * HW;
  const Taps t = make_taps(j, i, ifx, ify, Hf, Wf, geo);
  const int xa = min(max(t.x0, 0), Wf - 1), xb = min(max(t.x0 + 1, 0), Wf - 1);
  const int ya = min(max(t.y0, 0), Hf - 1), yb = min(max(t.y0 + 1, 0), Hf - 1);
  const int o[4] = {ya * Wf + xa, ya * Wf + xb, yb * Wf + xa, yb * Wf + xb};
  const float ax = (float)(t.x0 + 1) - t.ix, bx = t.ix - (float)t.x0;
  const float ay = (float)(t.y0 + 1) - t.iy, by = t.iy - (float)t.y0;
  const float dwx[4] = {-ay, ay, -by, by}, dwy[4] = {-ax, -bx, ax, bx};
  const float om = 1.0f - m;
  float gix = 0.f, giy = 0.f, gm = 0.f;
  unsigned long long* gi = g_init + (size_t)n * 2 * HW;
  // HALF the tap atomics (round 5, as in warp_bwd_kernel): the interpolation map is an up-sampled low-resolution field, so the lanes
  // of a wave (consecutive pixels of a row) sample consecutive pixels — the right-hand taps of pixel j are the left-hand taps of
  // pixel j + 1.  Each lane hands the fixed-point values of its right-hand contributions to the next lane, which adds them to its
  // own left-hand ones where the ADDRESSES agree (anything else keeps its own atomic).  Integer additions of the values the two
  // atomics would have added: every accumulator, and with it every output bit, is unchanged.
  const bool tin[4] = {inside && t.in[0], inside && t.in[1], inside && t.in[2], inside && t.in[3]};
  const int o0n = __shfl_down(o[0], 1), o2n = __shfl_down(o[2], 1);
  const int in0n = __shfl_down((int)tin[0], 1), in2n = __shfl_down((int)tin[2], 1);
  const bool m1 = lane < 63 && tin[1] && in0n && o[1] == o0n;
  const bool m3 = lane < 63 && tin[3] && in2n && o[3] == o2n;
  const bool p1 = __shfl_up((int)m1, 1) && lane > 0, p3 = __shfl_up((int)m3, 1) && lane > 0;
#pragma unroll
  for (int c = 0; c < 2; ++c) {
    const float* f = f0 + c * HW;
    const float g = inside ? g_up[((size_t)n * 2 + c) * HW + p] : 0.f;
    float warped = 0.f;
    long long q[4] = {0, 0, 0, 0};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      if (!tin[k]) continue;
      const float v = f[o[k]];
      warped += v * t.w[k];
      q[k] = fix_q(om * g * t.w[k]);
      gix += v * dwx[k] * g;
      giy += v * dwy[k] * g;
    }
    const long long r1 = __shfl_up(q[1], 1), r3 = __shfl_up(q[3], 1);
    if (tin[0]) fix_emit(gi + c * HW + o[0], q[0], p1 ? r1 : 0ll);
    if (tin[1] && !m1) fix_emit(gi + c * HW + o[1], q[1], 0ll);
    if (tin[2]) fix_emit(gi + c * HW + o[2], q[2], p3 ? r3 : 0ll);
    if (tin[3] && !m3) fix_emit(gi + c * HW + o[3], q[3], 0ll);
    if (inside) fix_add(gi + c * HW + p, m * g);
    gm += g * (f[p] - warped);
  }
  if (!inside) return;
  const float mx = ((float)(Wf - 1) * 0.5f) * (2.0f / (float)max(Wf - 1, 1));
  const float my = ((float)(Hf - 1) * 0.5f) * (2.0f / (float)max(Hf - 1, 1));
  gix *= om * mx;
  giy *= om * my;
  if (Hf == h && Wf == w) {
    float* gx = g_xo + (size_t)n * 3 * hw;
    gx[p] = gix;
    gx[hw + p] = giy;
    gx[2 * hw + p] = gm * m * (1.0f - m);
    return;
  }
  // final level: gradients of the UP-SAMPLED (inter_flow * ratio, mask) at full resolution; the resize gradient is a
  // gather (upsample_bwd_kernel) and the sigmoid' factor is applied per low-resolution pixel afterwards.  (The first
  // version scattered 12 fixed-point atomics per output pixel into the low-resolution map, ~64 colliding adds per
  // address: 750 us of an 830 us launch.)
  const float su = (float)((double)Wf / (double)w), sv = (float)((double)Hf / (double)h);
  float* gf = g_full + (size_t)n * 3 * HW;
  gf[p] = gix * su;
  gf[HW + p] = giy * sv;
  gf[2 * HW + p] = gm;
}

__global__ void blend_bwd_finish_kernel(const unsigned long long* __restrict__ gi64, float* __restrict__ g_init, long long n_init) {
  const long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x;
  if (i < n_init) g_init[i] = fix_get(gi64[i]);
}

// ---- flow up-sampling -------------------------------------------------------------------------
// 4 consecutive pixels of a row per thread with a 16-byte store when W % 4 == 0 (the source map is tiny and cache
// resident, so the wider per-lane footprint costs nothing here: 14.5 -> 8.3 us at [4,2,384,1280]); scales from the host.
template <int PX>
__global__ __launch_bounds__(THREADS)
void upsample_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, int C, int h, int w, int H, int W, int if_rate,
                         float scale_y, float scale_x, float rate_x, float rate_y) {
  const int HW = H * W;
  const int p0 = (blockIdx.x * THREADS + threadIdx.x) * PX;
  if (p0 >= HW) return;
  const int nc = blockIdx.y, c = nc % C;
  const int i = p0 / W, j0 = p0 - i * W;
  const Lerp ly = make_lerp_scaled(i, h, scale_y);
  const float* s0 = x + (size_t)nc * h * w + ly.i0 * w;
  const float* s1 = x + (size_t)nc * h * w + ly.i1 * w;
  const float rate = (c == 0) ? rate_x : rate_y;                      // pwc_modules.py:84-88
  float v[PX];
#pragma unroll
  for (int k = 0; k < PX; ++k) {
    const Lerp lx = make_lerp_scaled(j0 + k, w, scale_x);
    v[k] = ly.l0 * (lx.l0 * s0[lx.i0] + lx.l1 * s0[lx.i1]) + ly.l1 * (lx.l0 * s1[lx.i0] + lx.l1 * s1[lx.i1]);
    if (if_rate) v[k] *= rate;
  }
  if constexpr (PX == 4) *reinterpret_cast<float4*>(y + (size_t)nc * HW + p0) = make_float4(v[0], v[1], v[2], v[3]);
  else y[(size_t)nc * HW + p0] = v[0];
}

// Backward of the resize as a GATHER: every input pixel (a, b) sums the output pixels whose two-tap interpolation
// reads it — no atomics, deterministic.  (The first version scattered with fp32 atomics from one thread per OUTPUT
// pixel: the pyramid-distillation loss up-samples 4x13 flows to 256x832, i.e. 4096 threads per target address, and one
// such launch took 1.9 ms — a third of a training step.)  The candidate output range of an input row/column is
// bracketed generously and each candidate is re-tested with the forward's own make_lerp, so the weights are exactly
// the forward's.  WG = false: one thread per input pixel (small footprints); WG = true: one workgroup per input pixel,
// threads stride over the footprint, fixed-order LDS tree reduction.
__device__ __forceinline__ void upsample_bwd_range(int a, int in_size, int out_size, int& lo, int& hi) {
  if (in_size <= 1 || out_size <= 1) { lo = 0; hi = out_size - 1; return; }
  const float inv = (float)(out_size - 1) / (float)(in_size - 1);
  lo = max(0, (int)floorf((float)(a - 1) * inv) - 1);
  hi = min(out_size - 1, (int)ceilf((float)(a + 1) * inv) + 1);
}
__device__ __forceinline__ float upsample_bwd_weight(const Lerp& l, int a) {
  return (l.i0 == a ? l.l0 : 0.f) + (l.i1 == a ? l.l1 : 0.f);
}
// Group variant (footprints up to a few hundred outputs): G lanes per input pixel, lane r takes the footprint rows
// ilo + r, ilo + r + G, ...  The column weights depend only on (j, b) and the row weight only on (i, a): the <= MAXNJ column
// weights are computed once per group into an LDS column and the row weight once per row, and the G partial sums are
// combined by an xor-shuffle tree (fixed order).  (The first version: one thread per input pixel evaluating both
// interpolations — two IEEE divisions each — for every footprint pixel, ~100 per input pixel at the 4x resize of the last
// level and 361 at the 8x resize of the distillation loss: 117-138 us per launch.)
constexpr int MAXNJ = 24;
template <int G>
__global__ __launch_bounds__(THREADS)
void upsample_bwd_group_kernel(const float* __restrict__ gy, float* __restrict__ gx, int C, int h, int w, int H, int W, int if_rate, long long total) {
  constexpr int NG = THREADS / G;
  __shared__ float wxs[MAXNJ * NG];
  const int grp = threadIdx.x / G, r = threadIdx.x % G;
  const long long idx0 = blockIdx.x * (long long)NG + grp;
  const bool live = idx0 < total;
  const long long idx = live ? idx0 : total - 1;
  const int b = (int)(idx % w), a = (int)((idx / w) % h);
  const long long nc = idx / ((long long)w * h);
  const int c = (int)(nc % C);
  int ilo, ihi, jlo, jhi;
  upsample_bwd_range(a, h, H, ilo, ihi);
  upsample_bwd_range(b, w, W, jlo, jhi);
  const float* g = gy + (size_t)nc * H * W;
  const int nj = jhi - jlo + 1;
  const bool table = nj <= MAXNJ;
  float* wx = wxs + grp;
  if (table)
    for (int k = r; k < nj; k += G) wx[k * NG] = upsample_bwd_weight(make_lerp(jlo + k, w, W), b);
  __syncthreads();
  float acc = 0.f;
  for (int i = ilo + r; i <= ihi; i += G) {
    const float wy = upsample_bwd_weight(make_lerp(i, h, H), a);
    if (wy == 0.f) continue;
    const float* row = g + (size_t)i * W + jlo;
    if (table) {
      for (int k = 0; k < nj; ++k) acc += row[k] * wy * wx[k * NG];
    } else {
      for (int k = 0; k < nj; ++k) acc += row[k] * wy * upsample_bwd_weight(make_lerp(jlo + k, w, W), b);
    }
  }
#pragma unroll
  for (int o = G / 2; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
  if (r != 0 || !live) return;
  if (if_rate) acc *= (c == 0) ? (float)((double)W / (double)w) : (float)((double)H / (double)h);
  gx[idx] = acc;
}

// Workgroup variant (large footprints: the pyramid-distillation loss resizes 4x13 flows to 256x832): the 256 threads stride
// over the footprint, fixed-order LDS tree reduction.  Row and column weights come from two LDS tables filled once per
// workgroup (footprints up to WGT x WGT; the first version evaluated both interpolations for each of the ~17 K footprint
// pixels: 30 us per launch) — the same products in the same order, so the same bits.
constexpr int WGT = 160;
__global__ __launch_bounds__(THREADS)
void upsample_bwd_wg_kernel(const float* __restrict__ gy, float* __restrict__ gx, int C, int h, int w, int H, int W, int if_rate, long long total) {
  __shared__ float red[THREADS];
  __shared__ float wys[WGT], wxs[WGT];
  const long long idx = (long long)blockIdx.x;
  const int b = (int)(idx % w), a = (int)((idx / w) % h);
  const long long nc = idx / ((long long)w * h);
  const int c = (int)(nc % C);
  int ilo, ihi, jlo, jhi;
  upsample_bwd_range(a, h, H, ilo, ihi);
  upsample_bwd_range(b, w, W, jlo, jhi);
  const float* g = gy + (size_t)nc * H * W;
  const int ni = ihi - ilo + 1, nj = jhi - jlo + 1, cnt = ni * nj;
  const bool table = ni <= WGT && nj <= WGT;
  if (table) {
    for (int t = (int)threadIdx.x; t < ni; t += THREADS) wys[t] = upsample_bwd_weight(make_lerp(ilo + t, h, H), a);
    for (int t = (int)threadIdx.x; t < nj; t += THREADS) wxs[t] = upsample_bwd_weight(make_lerp(jlo + t, w, W), b);
  }
  __syncthreads();
  float acc = 0.f;
  for (int t = (int)threadIdx.x; t < cnt; t += THREADS) {
    const int di = t / nj, dj = t - di * nj;
    const float wy = table ? wys[di] : upsample_bwd_weight(make_lerp(ilo + di, h, H), a);
    const float wx = table ? wxs[dj] : upsample_bwd_weight(make_lerp(jlo + dj, w, W), b);
    if (wy != 0.f && wx != 0.f) acc += g[(size_t)(ilo + di) * W + jlo + dj] * wy * wx;
  }
  red[threadIdx.x] = acc;
  __syncthreads();
  for (int sft = THREADS / 2; sft > 0; sft >>= 1) {
    if ((int)threadIdx.x < sft) red[threadIdx.x] += red[threadIdx.x + sft];
    __syncthreads();
  }
  if (threadIdx.x != 0) return;
  acc = red[0];
  if (if_rate) acc *= (c == 0) ? (float)((double)W / (double)w) : (float)((double)H / (double)h);
  gx[idx] = acc;
}

// Band variant (round 5; every footprint larger than a few pixels): the two interpolations are SEPARABLE —
//     gx[a][b] = sum_j wx(j, b) * ( sum_i wy(i, a) * gy[i][j] )
// — so a workgroup of 1024 threads takes one input row a and NB consecutive input columns: threads = 4 row groups x 256
// output columns (two columns each: coalesced 1 KB row segments, four row groups in flight), the column sums of the row groups
// are added in LDS in fixed order, and 16 lanes per input pixel contract them with the column weights (xor-shuffle tree).
// Every output pixel is read once per input row that uses it (twice in all) instead of once per INPUT PIXEL that uses it (four
// times, by one thread per footprint pixel with an integer division each), and the pyramid-distillation resizes — 4x13 ... 64x208
// flows to 256x832, 52 / 17 / 21 / 12 / 12 us per launch in a config-3 step — stop being latency chains of one workgroup per
// input pixel.  Weights: the forward's own make_lerp (scales precomputed on the host with the identical IEEE division).
constexpr int BAND_T = 1024, BAND_COLS = 512, BAND_ROWS = 1024;
__global__ __launch_bounds__(BAND_T)
void upsample_bwd_band_kernel(const float* __restrict__ gy, float* __restrict__ gx, int C, int h, int w, int H, int W, int if_rate,
                              int NB, int nbb, float sy, float sx) {
  __shared__ float cs[4][BAND_COLS];
  __shared__ float wys[BAND_ROWS];
  const int tid = threadIdx.x, rg = tid >> 8, cj = tid & 255;
  const int bb = blockIdx.x % nbb, a = (blockIdx.x / nbb) % h;
  const long long nc = blockIdx.x / ((long long)nbb * h);
  const int b0 = bb * NB, b1 = min(w, b0 + NB);
  int ilo, ihi, jlo, jhi, t0, t1;
  upsample_bwd_range(a, h, H, ilo, ihi);
  upsample_bwd_range(b0, w, W, jlo, t0);
  upsample_bwd_range(b1 - 1, w, W, t1, jhi);
  const int nrows = ihi - ilo + 1;
  const bool table = nrows <= BAND_ROWS;
  if (table)
    for (int t = tid; t < nrows; t += BAND_T) wys[t] = upsample_bwd_weight(make_lerp_scaled(ilo + t, h, sy), a);
  __syncthreads();
  const float* g = gy + (size_t)nc * H * W;
  const int j0 = jlo + cj, j1 = j0 + 256;
  const bool in0 = j0 <= jhi, in1 = j1 <= jhi;
  float acc0 = 0.f, acc1 = 0.f;
  for (int i = ilo + rg; i <= ihi; i += 4) {
    const float wy = table ? wys[i - ilo] : upsample_bwd_weight(make_lerp_scaled(i, h, sy), a);
    if (wy == 0.f) continue;
    const float* row = g + (size_t)i * W;
    if (in0) acc0 += row[j0] * wy;
    if (in1) acc1 += row[j1] * wy;
  }
  cs[rg][cj] = acc0; cs[rg][cj + 256] = acc1;
  __syncthreads();
  if (tid < BAND_COLS) cs[0][tid] = ((cs[0][tid] + cs[1][tid]) + cs[2][tid]) + cs[3][tid];
  __syncthreads();
  const int b = b0 + (tid >> 4), r = tid & 15;
  float acc = 0.f;
  if (b < b1) {
    int jl, jh;
    upsample_bwd_range(b, w, W, jl, jh);
    for (int j = jl + r; j <= jh; j += 16) acc += cs[0][j - jlo] * upsample_bwd_weight(make_lerp_scaled(j, w, sx), b);
  }
#pragma unroll
  for (int o = 8; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
  if (r != 0 || b >= b1) return;
  if (if_rate) acc *= ((int)(nc % C) == 0) ? (float)((double)W / (double)w) : (float)((double)H / (double)h);
  gx[(nc * h + a) * w + b] = acc;
}

// ---- pyramid-distillation term, style 'upup' (model/upflow.py:461-487): every level's flow, up-sampled to the label's size, against
// the detached final flow under the occlusion mask:   weight * sum_l  [ sum (|up(x_l) - y| + eps)^q * occ ] / (sum occ + 1e-6).
// The reference composition — up-sample, robust sum, three scalar kernels per level and direction, and their backward — is 20
// launches and a full-resolution intermediate written and read twice per level and direction (0.57 ms of a 10.2 ms config-3
// step).  Here, per direction: ONE forward pass over the label reads y and occ once and interpolates all levels on the fly (the
// level maps are a few KB: cache resident); one finishing launch; ONE backward launch whose workgroups are the band kernel's
// (a low-resolution row and column strip each, all levels side by side in the grid) with the robust term's derivative evaluated
// in place of a stored gradient.
constexpr int MSD_MAXL = 6, MSD_T = 256;
struct MsdLevels {
  const float* x[MSD_MAXL]; float* gx[MSD_MAXL];
  int h[MSD_MAXL], w[MSD_MAXL], NB[MSD_MAXL], nbb[MSD_MAXL];
  float sy[MSD_MAXL], sx[MSD_MAXL], rx[MSD_MAXL], ry[MSD_MAXL];
  unsigned blk0[MSD_MAXL + 1];
  int n;
};
// x^e for x > 0 through the hardware's log2 / exp2 (1 ulp each): the libm powf of the composition is ~60 instructions, and the term
// is evaluated ~20 M times per direction in the backward (the launch was 122 us, bound by it)
__device__ __forceinline__ float msd_pow(float x, float e) { return __builtin_amdgcn_exp2f(e * __builtin_amdgcn_logf(x)); }
__device__ __forceinline__ float msd_up(const float* __restrict__ xl, int w, const Lerp& ly, const Lerp& lx) {
  const float* s0 = xl + ly.i0 * w; const float* s1 = xl + ly.i1 * w;
  return ly.l0 * (lx.l0 * s0[lx.i0] + lx.l1 * s0[lx.i1]) + ly.l1 * (lx.l0 * s1[lx.i0] + lx.l1 * s1[lx.i1]);       // (upsample_fwd_kernel's expression)
}
// partials[block][MSD_MAXL + 1] = { sum_l over the block's pixels ..., sum occ }
__global__ __launch_bounds__(MSD_T)
void msd_fwd_kernel(const MsdLevels L, const float* __restrict__ y, const float* __restrict__ occ, float* __restrict__ partials,
                    int H, int W, long long npix, float eps, float q) {
  __shared__ float sh[MSD_T / 64];
  const int HW = H * W;
  float s[MSD_MAXL], so = 0.f;
#pragma unroll
  for (int l = 0; l < MSD_MAXL; ++l) s[l] = 0.f;
  for (long long p = blockIdx.x * (long long)MSD_T + threadIdx.x; p < npix; p += (long long)gridDim.x * MSD_T) {
    const long long n = p / HW;
    const int r = (int)(p - n * HW), i = r / W, j = r - i * W;
    const float o = occ ? occ[p] : 1.0f;
    so += o;
    const float y0 = y[(size_t)(n * 2) * HW + r], y1 = y[(size_t)(n * 2 + 1) * HW + r];
#pragma unroll
    for (int l = 0; l < MSD_MAXL; ++l)
      if (l < L.n) {
        const Lerp ly = make_lerp_scaled(i, L.h[l], L.sy[l]), lx = make_lerp_scaled(j, L.w[l], L.sx[l]);
        const size_t hw = (size_t)L.h[l] * L.w[l];
        const float v0 = msd_up(L.x[l] + (size_t)(n * 2) * hw, L.w[l], ly, lx) * L.rx[l];
        const float v1 = msd_up(L.x[l] + (size_t)(n * 2 + 1) * hw, L.w[l], ly, lx) * L.ry[l];
        const float t = msd_pow(fabsf(v0 - y0) + eps, q) + msd_pow(fabsf(v1 - y1) + eps, q);
        s[l] += t * o;
      }
  }
  auto bsum = [&](float v) {
#pragma unroll
    for (int o2 = 32; o2 > 0; o2 >>= 1) v += __shfl_xor(v, o2, 64);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    float r = 0.f;
#pragma unroll
    for (int k = 0; k < MSD_T / 64; ++k) r += sh[k];
    return r;
  };
  float* out = partials + (size_t)blockIdx.x * (MSD_MAXL + 1);
#pragma unroll
  for (int l = 0; l < MSD_MAXL; ++l) { const float v = bsum(s[l]); if (threadIdx.x == 0) out[l] = v; }
  const float v = bsum(so);
  if (threadIdx.x == 0) out[MSD_MAXL] = v;
}
// out[0] = weight * sum_l S_l / den,  out[1] = den   (den = sum occ + 1e-6, or den_const > 0: the element count without a mask)
__global__ __launch_bounds__(MSD_T)
void msd_finish_kernel(const float* __restrict__ partials, int nb, int nl, float weight, float den_const, float* __restrict__ out) {
  __shared__ float sh[MSD_T / 64];
  __shared__ float tot[MSD_MAXL + 1];
  for (int k = 0; k <= MSD_MAXL; ++k) {
    float v = 0.f;
    for (int b = threadIdx.x; b < nb; b += MSD_T) v += partials[(size_t)b * (MSD_MAXL + 1) + k];
#pragma unroll
    for (int o2 = 32; o2 > 0; o2 >>= 1) v += __shfl_xor(v, o2, 64);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) tot[k] = (sh[0] + sh[1]) + (sh[2] + sh[3]);
  }
  __syncthreads();
  if (threadIdx.x != 0) return;
  const float den = den_const > 0.f ? den_const : tot[MSD_MAXL] + 1e-6f;
  float acc = 0.f;
  for (int l = 0; l < nl; ++l) acc += tot[l] / den;
  out[0] = weight * acc; out[1] = den;
}
// backward: workgroup -> (level, plane nc, input row a, column strip); see upsample_bwd_band_kernel
__global__ __launch_bounds__(BAND_T)
void msd_bwd_kernel(const MsdLevels L, const float* __restrict__ y, const float* __restrict__ occ, const float* __restrict__ gout,
                    const float* __restrict__ fwd_out /* [1] = den */, float weight, int H, int W, float eps, float q) {
  // (both flow components of an image in ONE workgroup: they share the interpolation geometry, the mask and the row / column weights)
  __shared__ float cs[2][4][BAND_COLS];
  __shared__ float wys[BAND_ROWS];
  int l = 0;
#pragma unroll
  for (int k = 1; k < MSD_MAXL; ++k) if (k < L.n && blockIdx.x >= L.blk0[k]) l = k;
  const float* xl = L.x[0]; float* gxl = L.gx[0];
  int h = L.h[0], w = L.w[0], NB = L.NB[0], nbb = L.nbb[0]; float sy = L.sy[0], sx = L.sx[0], rx = L.rx[0], ry = L.ry[0]; unsigned b0blk = L.blk0[0];
#pragma unroll
  for (int k = 1; k < MSD_MAXL; ++k)
    if (k == l) { xl = L.x[k]; gxl = L.gx[k]; h = L.h[k]; w = L.w[k]; NB = L.NB[k]; nbb = L.nbb[k]; sy = L.sy[k]; sx = L.sx[k]; rx = L.rx[k]; ry = L.ry[k]; b0blk = L.blk0[k]; }
  const unsigned bid = blockIdx.x - b0blk;
  const int tid = threadIdx.x, rg = tid >> 8, cj = tid & 255;
  const int bb = bid % nbb, a = (bid / nbb) % h;
  const int n = bid / ((unsigned)nbb * h);
  const int b0 = bb * NB, b1 = min(w, b0 + NB);
  int ilo, ihi, jlo, jhi, t0, t1;
  upsample_bwd_range(a, h, H, ilo, ihi);
  upsample_bwd_range(b0, w, W, jlo, t0);
  upsample_bwd_range(b1 - 1, w, W, t1, jhi);
  const int nrows = ihi - ilo + 1;
  const bool table = nrows <= BAND_ROWS;
  if (table)
    for (int t = tid; t < nrows; t += BAND_T) wys[t] = upsample_bwd_weight(make_lerp_scaled(ilo + t, h, sy), a);
  __syncthreads();
  const size_t HW = (size_t)H * W, hw = (size_t)h * w;
  const float* y0p = y + (size_t)n * 2 * HW;
  const float* oc = occ ? occ + (size_t)n * HW : nullptr;
  const float* x0p = xl + (size_t)n * 2 * hw;
  const int j0 = jlo + cj, j1 = j0 + 256;
  const bool in0 = j0 <= jhi, in1 = j1 <= jhi;
  const Lerp lx0 = make_lerp_scaled(in0 ? j0 : jlo, w, sx), lx1 = make_lerp_scaled(in1 ? j1 : jlo, w, sx);
  // d/dx of (|x - y| + eps)^q * occ at one label pixel, both components (robust_bwd_kernel's expression without the scalar coefficient)
  auto dterm = [&](const Lerp& ly, const Lerp& lx, int i, int j, float wy, float& a0, float& a1) {
    const size_t pix = (size_t)i * W + j;
    const float o = (oc ? oc[pix] : 1.0f) * q * wy;
    const float d0 = msd_up(x0p, w, ly, lx) * rx - y0p[pix], d1 = msd_up(x0p + hw, w, ly, lx) * ry - y0p[HW + pix];
    const float s0 = (d0 > 0.f) ? 1.0f : ((d0 < 0.f) ? -1.0f : 0.f), s1 = (d1 > 0.f) ? 1.0f : ((d1 < 0.f) ? -1.0f : 0.f);
    a0 += o * msd_pow(fabsf(d0) + eps, q - 1.0f) * s0;
    a1 += o * msd_pow(fabsf(d1) + eps, q - 1.0f) * s1;
  };
  float acc[2][2] = {{0.f, 0.f}, {0.f, 0.f}};              // [component][column]
  for (int i = ilo + rg; i <= ihi; i += 4) {
    const Lerp ly = make_lerp_scaled(i, h, sy);
    const float wy = table ? wys[i - ilo] : upsample_bwd_weight(ly, a);
    if (wy == 0.f) continue;
    if (in0) dterm(ly, lx0, i, j0, wy, acc[0][0], acc[1][0]);
    if (in1) dterm(ly, lx1, i, j1, wy, acc[0][1], acc[1][1]);
  }
#pragma unroll
  for (int c = 0; c < 2; ++c) { cs[c][rg][cj] = acc[c][0]; cs[c][rg][cj + 256] = acc[c][1]; }
  __syncthreads();
  {
    const int c = tid >> 9, t = tid & (BAND_COLS - 1);       // 1024 threads = 2 components x 512 columns
    cs[c][0][t] = ((cs[c][0][t] + cs[c][1][t]) + cs[c][2][t]) + cs[c][3][t];
  }
  __syncthreads();
  const int b = b0 + (tid >> 4), r = tid & 15;
  float s0 = 0.f, s1 = 0.f;
  if (b < b1) {
    int jl, jh;
    upsample_bwd_range(b, w, W, jl, jh);
    for (int j = jl + r; j <= jh; j += 16) {
      const float wx = upsample_bwd_weight(make_lerp_scaled(j, w, sx), b);
      s0 += cs[0][0][j - jlo] * wx; s1 += cs[1][0][j - jlo] * wx;
    }
  }
#pragma unroll
  for (int o2 = 8; o2 > 0; o2 >>= 1) { s0 += __shfl_xor(s0, o2, 64); s1 += __shfl_xor(s1, o2, 64); }
  if (r != 0 || b >= b1) return;
  const float coef = gout[0] * weight / fwd_out[1];
  gxl[((size_t)(n * 2) * h + a) * w + b] = s0 * rx * coef;
  gxl[((size_t)(n * 2 + 1) * h + a) * w + b] = s1 * ry * coef;
}

// g_logit *= s (1 - s), s = sigmoid(logit): the mask channel of the final-level blend after its resize gradient
template <typename T>
__global__ void sigmoid_grad_kernel(const T* __restrict__ x_out, float* __restrict__ g_xo, int hw, long long total) {
  const long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x;
  if (i >= total) return;
  const long long n = i / hw;
  const size_t o = (size_t)n * 3 * hw + 2 * hw + (i - n * hw);
  const float sg = sigmoidf(Elem<T>::load(x_out + o));
  g_xo[o] *= sg * (1.0f - sg);
}

// out[n, :] = cast(a + (b + c))  — the flow bookkeeping of one pyramid level (model/upflow.py:566-572:
// `flow_up + res` fed to the context network, `flow_up + (res + fine)` handed to the next level) and the fp32 -> 16-bit
// copies of the flow into estimator-input slots, each ONE launch instead of a convert / add / convert chain of ATen
// kernels.  a: fp32 [N, per]; b, c: optional 16-bit [N, per] (conv outputs); the additions happen in fp32 in exactly
// this association.  out: fp32 or 16-bit, rows `obs` elements apart (a channel slice of a wider buffer).
template <typename T, typename TO>
__global__ void flow_update_kernel(const float* __restrict__ a, const T* __restrict__ b, const T* __restrict__ c,
                                   TO* __restrict__ out, long long obs, int per, long long total) {
  const long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x;
  if (i >= total) return;
  const long long n = i / per;
  const int r = (int)(i - n * per);
  float v = a[i];
  if (b) {
    float t = Elem<T>::load(b + i);
    if (c) t = t + Elem<T>::load(c + i);
    v = v + t;
  }
  Elem<TO>::store(out + n * obs + r, v);
}

// The same sum written as ONE channel-octet entry per pixel of a C8 buffer (conv_c8.hip): the 2 flow components in positions
// 0, 1 and zeros in 2..7 (a whole 16-byte store: the padding positions of the octet are defined, not left-over memory).
template <typename T>
__global__ void flow_update_c8_kernel(const float* __restrict__ a, const T* __restrict__ b, const T* __restrict__ c,
                                      uint4* __restrict__ out, long long obs16, int HW, long long total) {
  const long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x;        // (item, pixel)
  if (i >= total) return;
  const long long n = i / HW;
  const int r = (int)(i - n * HW);
  float v[2];
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const long long j = (n * 2 + k) * HW + r;
    v[k] = a[j];
    if (b) {
      float t = Elem<T>::load(b + j);
      if (c) t = t + Elem<T>::load(c + j);
      v[k] = v[k] + t;
    }
  }
  out[n * obs16 + r] = make_uint4(pack2<T>(v[0], v[1]), 0u, 0u, 0u);
}

}  // namespace sgu
}  // namespace upf

extern "C" long long upf_sgu_blend_forward_workspace_bytes(int B, int h, int w, int Hf, int Wf) {
  return (Hf == h && Wf == w) ? 0 : (long long)B * h * w * (long long)sizeof(float);
}

extern "C" int upf_sgu_blend_forward(const float* flow_init, const void* x_out, float* flow_up, float* inter_flow,
                                     float* inter_mask, void* workspace, int B, int h, int w, int Hf, int Wf, int dtype, void* stream) {
  using namespace upf;
  UPF_REQUIRE(flow_init && x_out && flow_up, UPF_EINVAL, "sgu_blend_forward: null pointer");
  UPF_REQUIRE(B > 0 && B <= 65535 && h > 0 && w > 0 && Hf >= h && Wf >= w, UPF_EINVAL,
              "sgu_blend_forward: bad shape B=%d x_out %dx%d flow %dx%d", B, h, w, Hf, Wf);
  const bool level = (Hf == h && Wf == w);
  UPF_REQUIRE(level || workspace, UPF_EINVAL, "sgu_blend_forward: the final level needs a workspace of upf_sgu_blend_forward_workspace_bytes");
  hipStream_t st = (hipStream_t)stream;
  float* sig = (float*)workspace;
  UPF_REQUIRE(Hf <= 4 * 65535, UPF_EINVAL, "sgu_blend_forward: image too tall");
  dim3 grid(cdiv(Wf, 64), cdiv(Hf, 4), B);
  // the reference's scales, computed once with the same IEEE operations the kernels used per thread
  const float scale_y = lerp_scale(h, Hf), scale_x = lerp_scale(w, Wf);
  const float su = (float)((double)Wf / (double)w), sv = (float)((double)Hf / (double)h);
  const long long nsig = (long long)B * h * w;
  UPF_DISPATCH(dtype, T,
               if (!level) hipLaunchKernelGGL((sgu::sigmoid_map_kernel<T>), dim3((unsigned)((nsig + 255) / 256)), dim3(256), 0, st, (const T*)x_out, sig, h * w, nsig);
               hipLaunchKernelGGL((sgu::blend_fwd_kernel<T>), grid, dim3(sgu::THREADS), 0, st, flow_init, (const T*)x_out, sig, flow_up, inter_flow, inter_mask,
                                  h, w, Hf, Wf, scale_y, scale_x, su, sv, make_sample_geom(Hf, Wf)));
  return check_launch("sgu_blend_forward");
}

extern "C" int upf_sgu_blend_forward_flow16(const float* flow_init, const void* x_out, float* flow_up, void* flow16, long long flow16_batch_stride,
                                            int flow16_is_c8, int B, int H, int W, int dtype, void* stream) {
  using namespace upf;
  UPF_REQUIRE(flow_init && x_out && flow_up && flow16, UPF_EINVAL, "sgu_blend_forward_flow16: null pointer");
  UPF_REQUIRE(B > 0 && B <= 65535 && H > 0 && W > 0 && H <= 4 * 65535, UPF_EINVAL, "sgu_blend_forward_flow16: bad shape B=%d %dx%d", B, H, W);
  UPF_REQUIRE(dtype == UPF_BF16 || dtype == UPF_F16, UPF_EDTYPE, "sgu_blend_forward_flow16: bf16 / fp16 x_out (the 16-bit flow takes its type)");
  UPF_REQUIRE(!flow16_is_c8 || (aligned_to(flow16, 16) && flow16_batch_stride % 8 == 0), UPF_EINVAL, "sgu_blend_forward_flow16: the octet output must be 16-byte aligned");
  dim3 grid(cdiv(W, 64), cdiv(H, 4), B);
  hipStream_t st = (hipStream_t)stream;
  if (dtype == UPF_BF16)
    hipLaunchKernelGGL((sgu::blend_fwd_kernel<bf16_t>), grid, dim3(sgu::THREADS), 0, st, flow_init, (const bf16_t*)x_out, (const float*)nullptr, flow_up, (float*)nullptr, (float*)nullptr,
                       H, W, H, W, 1.f, 1.f, 1.f, 1.f, make_sample_geom(H, W), (bf16_t*)flow16, flow16_batch_stride, flow16_is_c8);
  else
    hipLaunchKernelGGL((sgu::blend_fwd_kernel<f16_t>), grid, dim3(sgu::THREADS), 0, st, flow_init, (const f16_t*)x_out, (const float*)nullptr, flow_up, (float*)nullptr, (float*)nullptr,
                       H, W, H, W, 1.f, 1.f, 1.f, 1.f, make_sample_geom(H, W), (f16_t*)flow16, flow16_batch_stride, flow16_is_c8);
  return check_launch("sgu_blend_forward_flow16");
}

extern "C" long long upf_sgu_blend_backward_workspace_bytes(int B, int h, int w, int Hf, int Wf) {
  const bool final_level = !(Hf == h && Wf == w);
  return (long long)B * 2 * Hf * Wf * (long long)sizeof(unsigned long long) + (final_level ? (long long)B * 3 * Hf * Wf * (long long)sizeof(float) : 0);
}

static int launch_upsample_backward(const float* grad_y, float* gx, long long BC, int C, int h, int w, int H, int W, int if_rate, hipStream_t stream) {
  using namespace upf;
  const long long total = BC * h * w;
  // footprint of one input pixel ~ (2H/h) x (2W/w) outputs: 1 / 4 / 16 lanes per input pixel, a workgroup beyond a few hundred
  const long long fp = (2LL * cdiv(H, h) + 2) * (2LL * cdiv(W, w) + 2);
  auto groups = [&](int G) { return dim3((unsigned)((total + sgu::THREADS / G - 1) / (sgu::THREADS / G))); };
  // band kernel: NB input columns per workgroup such that their output columns fit its 512-column strip
  // (from the 8x resize up: at 4x — ten output rows per input row — the per-pixel groups below are faster: 9.9 against 16.6 us)
  if (fp > 200 && w > 1 && W > 1) {
    const double inv = (double)(W - 1) / (double)(w - 1);
    int NB = (int)((sgu::BAND_COLS - 8) / inv) - 1;
    if (NB > 64) NB = 64;
    if (NB > w) NB = w;
    const long long nbb = NB >= 1 ? cdiv(w, NB) : 0, nwg = BC * h * nbb;
    if (NB >= 1 && nwg < (1LL << 31)) {
      hipLaunchKernelGGL(sgu::upsample_bwd_band_kernel, dim3((unsigned)nwg), dim3(sgu::BAND_T), 0, stream, grad_y, gx, C, h, w, H, W, if_rate, NB, (int)nbb,
                         lerp_scale(h, H), lerp_scale(w, W));
      return 0;
    }
  }
  if (fp > 512 && total < (1LL << 31))
    hipLaunchKernelGGL(sgu::upsample_bwd_wg_kernel, dim3((unsigned)total), dim3(sgu::THREADS), 0, stream, grad_y, gx, C, h, w, H, W, if_rate, total);
  else if (fp > 200)
    hipLaunchKernelGGL(sgu::upsample_bwd_group_kernel<16>, groups(16), dim3(sgu::THREADS), 0, stream, grad_y, gx, C, h, w, H, W, if_rate, total);
  else if (fp > 16)
    hipLaunchKernelGGL(sgu::upsample_bwd_group_kernel<4>, groups(4), dim3(sgu::THREADS), 0, stream, grad_y, gx, C, h, w, H, W, if_rate, total);
  else
    hipLaunchKernelGGL(sgu::upsample_bwd_group_kernel<1>, groups(1), dim3(sgu::THREADS), 0, stream, grad_y, gx, C, h, w, H, W, if_rate, total);
  return 0;
}

extern "C" int upf_sgu_blend_backward(const float* flow_init, const void* x_out, const float* grad_flow_up,
                                      float* g_flow_init32, float* g_x_out32, void* workspace, int B, int h, int w, int Hf, int Wf,
                                      int dtype, void* stream) {
  using namespace upf;
  UPF_REQUIRE(flow_init && x_out && grad_flow_up && g_flow_init32 && g_x_out32 && workspace, UPF_EINVAL, "sgu_blend_backward: null pointer");
  UPF_REQUIRE(B > 0 && B <= 65535 && h > 0 && w > 0 && Hf >= h && Wf >= w, UPF_EINVAL, "sgu_blend_backward: bad shape");
  hipStream_t s = (hipStream_t)stream;
  const long long n_init = (long long)B * 2 * Hf * Wf;
  const bool final_level = !(Hf == h && Wf == w);
  unsigned long long* gi64 = (unsigned long long*)workspace;
  float* g_full = (float*)(gi64 + n_init);                  // final level: [B, 3, Hf, Wf] gradients before the resize gradient
  { const int zrc = zero_fill_u64(gi64, n_init, s); if (zrc) return zrc; }      // (a kernel, not a memset node: common.hpp)
  dim3 grid(cdiv(Hf * Wf, sgu::THREADS), B);
  UPF_DISPATCH(dtype, T,
               hipLaunchKernelGGL((sgu::blend_bwd_kernel<T>), grid, dim3(sgu::THREADS), 0, s,
                                  flow_init, (const T*)x_out, grad_flow_up, gi64, g_full, g_x_out32, h, w, Hf, Wf, make_sample_geom(Hf, Wf)));
  hipLaunchKernelGGL(sgu::blend_bwd_finish_kernel, dim3((unsigned)((n_init + 255) / 256)), dim3(256), 0, s, gi64, g_flow_init32, n_init);
  if (final_level) {
    launch_upsample_backward(g_full, g_x_out32, (long long)B * 3, 3, h, w, Hf, Wf, 0, s);
    const long long nm = (long long)B * h * w;
    UPF_DISPATCH(dtype, T, hipLaunchKernelGGL((sgu::sigmoid_grad_kernel<T>), dim3((unsigned)((nm + 255) / 256)), dim3(256), 0, s, (const T*)x_out, g_x_out32, h * w, nm));
  }
  return check_launch("sgu_blend_backward");
}

static int msd_levels(upf::sgu::MsdLevels& L, const float* const* x_low, float* const* gx_low, const int* hs, const int* ws, int nl, int B, int H, int W) {
  using namespace upf;
  UPF_REQUIRE(x_low && hs && ws && nl >= 1 && nl <= sgu::MSD_MAXL && B > 0 && H > 1 && W > 1, UPF_EINVAL, "msd_upup: 1..%d levels, a label of at least 2x2", sgu::MSD_MAXL);
  L = upf::sgu::MsdLevels{};
  unsigned blk = 0;
  for (int l = 0; l < nl; ++l) {
    UPF_REQUIRE(x_low[l] && hs[l] >= 1 && ws[l] >= 2 && hs[l] <= H && ws[l] <= W, UPF_EUNSUPPORTED, "msd_upup: level %d (%dx%d) must be no larger than the label and at least 2 wide", l, hs[l], ws[l]);
    L.x[l] = x_low[l]; L.gx[l] = gx_low ? gx_low[l] : nullptr;
    L.h[l] = hs[l]; L.w[l] = ws[l];
    L.sy[l] = lerp_scale(hs[l], H); L.sx[l] = lerp_scale(ws[l], W);
    L.rx[l] = (float)((double)W / (double)ws[l]); L.ry[l] = (float)((double)H / (double)hs[l]);
    const double inv = (double)(W - 1) / (double)(ws[l] - 1);
    int NB = (int)((sgu::BAND_COLS - 8) / inv) - 1;
    if (NB > 64) NB = 64;
    if (NB > ws[l]) NB = ws[l];
    UPF_REQUIRE(NB >= 1, UPF_EUNSUPPORTED, "msd_upup: resize ratio too large for the band kernel (level %d)", l);
    L.NB[l] = NB; L.nbb[l] = cdiv(ws[l], NB);
    L.blk0[l] = blk;
    const long long nwg = (long long)B * hs[l] * L.nbb[l];          // one workgroup per (image, input row, column strip): both components
    UPF_REQUIRE((long long)blk + nwg < (1ll << 31), UPF_EINVAL, "msd_upup: grid too large");
    blk += (unsigned)nwg;
  }
  L.blk0[nl] = blk;
  for (int l = nl + 1; l <= sgu::MSD_MAXL; ++l) L.blk0[l] = blk;
  L.n = nl;
  return UPF_OK;
}

extern "C" int upf_msd_upup_partials(int B, int H, int W) {
  const long long b = ((long long)B * H * W + upf::sgu::MSD_T - 1) / upf::sgu::MSD_T;
  return (int)(b < 1 ? 1 : (b > 1024 ? 1024 : b));
}

extern "C" int upf_msd_upup_forward(const float* const* x_low, const int* hs, const int* ws, int nlevels, const float* y, const float* occ,
                                    float* partials, float* out2, int B, int H, int W, float weight, float eps, float q, void* stream) {
  using namespace upf;
  UPF_REQUIRE(y && partials && out2, UPF_EINVAL, "msd_upup_forward: null pointer");
  sgu::MsdLevels L;
  if (int rc = msd_levels(L, x_low, nullptr, hs, ws, nlevels, B, H, W)) return rc;
  const int nb = upf_msd_upup_partials(B, H, W);
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(sgu::msd_fwd_kernel, dim3(nb), dim3(sgu::MSD_T), 0, s, L, y, occ, partials, H, W, (long long)B * H * W, eps, q);
  hipLaunchKernelGGL(sgu::msd_finish_kernel, dim3(1), dim3(sgu::MSD_T), 0, s, (const float*)partials, nb, nlevels, weight,
                     occ ? 0.f : (float)((double)B * 2 * H * W), out2);
  return check_launch("msd_upup_forward");
}

extern "C" int upf_msd_upup_backward(const float* const* x_low, float* const* grad_x_low, const int* hs, const int* ws, int nlevels, const float* y,
                                     const float* occ, const float* grad_out, const float* fwd_out2, int B, int H, int W, float weight, float eps, float q,
                                     void* stream) {
  using namespace upf;
  UPF_REQUIRE(y && grad_out && fwd_out2 && grad_x_low, UPF_EINVAL, "msd_upup_backward: null pointer");
  sgu::MsdLevels L;
  if (int rc = msd_levels(L, x_low, grad_x_low, hs, ws, nlevels, B, H, W)) return rc;
  for (int l = 0; l < nlevels; ++l) UPF_REQUIRE(grad_x_low[l], UPF_EINVAL, "msd_upup_backward: null gradient buffer (level %d)", l);
  hipLaunchKernelGGL(sgu::msd_bwd_kernel, dim3(L.blk0[nlevels]), dim3(sgu::BAND_T), 0, (hipStream_t)stream, L, y, occ, grad_out, fwd_out2, weight, H, W, eps, q);
  return check_launch("msd_upup_backward");
}

extern "C" int upf_flow_upsample_forward(const float* x, float* y, int B, int C, int h, int w, int H, int W,
                                         int if_rate, void* stream) {
  using namespace upf;
  UPF_REQUIRE(x && y, UPF_EINVAL, "flow_upsample_forward: null pointer");
  UPF_REQUIRE(B > 0 && C > 0 && (long long)B * C <= 65535 && h > 0 && w > 0 && H > 0 && W > 0, UPF_EINVAL, "flow_upsample_forward: bad shape");
  UPF_REQUIRE(!if_rate || C == 2, UPF_EINVAL, "flow_upsample_forward: if_rate needs a 2-channel flow, got C=%d", C);
  const bool four = (W % 4 == 0) && aligned_to(y, 16);
  dim3 grid(cdiv(cdiv(H * W, four ? 4 : 1), sgu::THREADS), B * C);
  const float sy = lerp_scale(h, H), sx = lerp_scale(w, W);
  const float rx = (float)((double)W / (double)w), ry = (float)((double)H / (double)h);
  if (four) hipLaunchKernelGGL(sgu::upsample_fwd_kernel<4>, grid, dim3(sgu::THREADS), 0, (hipStream_t)stream, x, y, C, h, w, H, W, if_rate, sy, sx, rx, ry);
  else hipLaunchKernelGGL(sgu::upsample_fwd_kernel<1>, grid, dim3(sgu::THREADS), 0, (hipStream_t)stream, x, y, C, h, w, H, W, if_rate, sy, sx, rx, ry);
  return check_launch("flow_upsample_forward");
}

extern "C" int upf_flow_upsample_backward(const float* grad_y, float* gx, int B, int C, int h, int w, int H, int W,
                                          int if_rate, void* stream) {
  using namespace upf;
  UPF_REQUIRE(grad_y && gx, UPF_EINVAL, "flow_upsample_backward: null pointer");
  UPF_REQUIRE(B > 0 && C > 0 && (long long)B * C <= 65535 && h > 0 && w > 0 && H > 0 && W > 0, UPF_EINVAL, "flow_upsample_backward: bad shape");
  UPF_REQUIRE(!if_rate || C == 2, UPF_EINVAL, "flow_upsample_backward: if_rate needs a 2-channel flow, got C=%d", C);
  launch_upsample_backward(grad_y, gx, (long long)B * C, C, h, w, H, W, if_rate, (hipStream_t)stream);
  return check_launch("flow_upsample_backward");
}

extern "C" int upf_flow_update(const float* a, const void* b, const void* c, void* out, long long out_batch_stride,
                               int out_is_f32, int N, int per_item, int dtype, void* stream) {
  using namespace upf;
  UPF_REQUIRE(a && out && N > 0 && per_item > 0, UPF_EINVAL, "flow_update: bad arguments");
  UPF_REQUIRE(b || !c, UPF_EINVAL, "flow_update: c without b");
  UPF_REQUIRE(dtype == UPF_BF16 || dtype == UPF_F16, UPF_EDTYPE, "flow_update: b, c and a 16-bit out are bf16 / fp16");
  const long long obs = out_batch_stride ? out_batch_stride : per_item;
  UPF_REQUIRE(obs >= per_item, UPF_EINVAL, "flow_update: out_batch_stride smaller than a row");
  const long long total = (long long)N * per_item;
  const dim3 grid((unsigned)((total + 255) / 256)), block(256);
  hipStream_t st = (hipStream_t)stream;
#define UPF_FU(T)                                                                                                                     \
  if (out_is_f32) hipLaunchKernelGGL((sgu::flow_update_kernel<T, float>), grid, block, 0, st, a, (const T*)b, (const T*)c, (float*)out, obs, per_item, total); \
  else hipLaunchKernelGGL((sgu::flow_update_kernel<T, T>), grid, block, 0, st, a, (const T*)b, (const T*)c, (T*)out, obs, per_item, total);
  if (dtype == UPF_BF16) { UPF_FU(bf16_t) } else { UPF_FU(f16_t) }
#undef UPF_FU
  return check_launch("flow_update");
}

extern "C" int upf_flow_update_c8(const float* a, const void* b, const void* c, void* out8, long long out8_batch_stride, int N, int HW,
                                  int dtype, void* stream) {
  using namespace upf;
  UPF_REQUIRE(a && out8 && N > 0 && HW > 0, UPF_EINVAL, "flow_update_c8: bad arguments");
  UPF_REQUIRE(b || !c, UPF_EINVAL, "flow_update_c8: c without b");
  UPF_REQUIRE(dtype == UPF_BF16 || dtype == UPF_F16, UPF_EDTYPE, "flow_update_c8: bf16 / fp16 buffers");
  UPF_REQUIRE(out8_batch_stride % 8 == 0 && out8_batch_stride >= (long long)HW * 8 && aligned_to(out8, 16), UPF_EINVAL,
              "flow_update_c8: out8 must be a 16-byte aligned octet of a C8 buffer");
  const long long total = (long long)N * HW;
  const dim3 grid((unsigned)((total + 255) / 256)), block(256);
  hipStream_t st = (hipStream_t)stream;
  if (dtype == UPF_BF16) hipLaunchKernelGGL((sgu::flow_update_c8_kernel<bf16_t>), grid, block, 0, st, a, (const bf16_t*)b, (const bf16_t*)c, (uint4*)out8, out8_batch_stride / 8, HW, total);
  else hipLaunchKernelGGL((sgu::flow_update_c8_kernel<f16_t>), grid, block, 0, st, a, (const f16_t*)b, (const f16_t*)c, (uint4*)out8, out8_batch_stride / 8, HW, total);
  return check_launch("flow_update_c8");
}
