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
#include "norm_merge.hpp"
#include "internal.hpp"

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

// PITCHED planes (round 5): the H rows of a plane are `pitch` elements apart (pitch > W; upf_conv_forward_pitched).  The statistics
// visit the plane's H*W LOGICAL elements in exactly the order — and with exactly the thread assignment — of the contiguous form
// (flat index i <-> row i / W, column i % W), so a pitched tensor and its contiguous copy get the same bits; only the addresses
// differ, and a run of 8 flat elements, which may cross a row end, is gathered element by element.
struct FlatWalk {
  int r, c;
  __device__ __forceinline__ FlatWalk(int i, int W) { r = i / W; c = i - r * W; }
  __device__ __forceinline__ void step(int qr, int qc, int W) { r += qr; c += qc; if (c >= W) { c -= W; ++r; } }
  __device__ __forceinline__ size_t at(int pitch) const { return (size_t)r * pitch + c; }
};
template <typename T, int V>
__device__ __forceinline__ void gather_run(const T* __restrict__ plane, FlatWalk w, int W, int pitch, float (&v)[V]) {
#pragma unroll
  for (int k = 0; k < V; ++k) { v[k] = Elem<T>::load(plane + w.at(pitch)); w.step(0, 1, W); }
}

// Rows are split into `nseg` segments so that the grid fills the chip even when B*C is small
// (config 5: 32 rows of 172,800 elements).  Deterministic two-launch scheme, no atomics:
//   stats kernel : every (row, segment) workgroup writes (count, mean, M2) of its segment to `ws`;
//   apply kernel : every (row, segment) workgroup merges the row's nseg partials in fixed order with
//                  Chan's parallel-variance formula (no E[x^2]-E[x]^2 cancellation), then normalises
//                  its own segment.  The second read of x comes from L2 / infinity cache.
// VEC: 16-byte accesses (4 fp32 / 8 bf16 / 8 fp16 per lane) when HW, the segment length and the base
// pointers allow it; otherwise element accesses.
template <typename T, bool VEC>
__global__ __launch_bounds__(NT)
void normalize_stats_kernel(const T* __restrict__ x, float* __restrict__ ws, int HW, int nseg, int seglen,
                            const T* __restrict__ x2 = nullptr, size_t rows1 = ~(size_t)0, float2* __restrict__ fin = nullptr,
                            int W = 0, int pitch = 0) {
  __shared__ float sh[NT / 64];
  constexpr int V = VEC ? VecIO<T>::N : 1;
  const size_t row = blockIdx.x / nseg;
  const int seg = blockIdx.x - (int)row * nseg;
  const int i0 = seg * seglen, i1 = min(HW, i0 + seglen);
  const bool pitched = W > 0 && pitch != W;                                   // (uniform)
  const size_t pstride = pitched ? (size_t)(HW / W) * pitch : (size_t)HW;     // elements per (n, c) plane
  const T* xr = (row < rows1) ? x + row * pstride : x2 + (row - rows1) * pstride;      // (two tensors in one launch: rows >= rows1 -> x2)
  const float cnt = (float)(i1 - i0);
  float mean, m2;
  if constexpr (sizeof(typename Elem<T>::store_t) == 2) {
    // 16-bit features: ONE sweep — sums of (x - K) and (x - K)^2 about a pivot K taken from the segment
    // itself, so that M2 = s2 - s1^2/n loses at most a bit or two (far below the bf16/fp16 output rounding)
    float s1 = 0.f, s2 = 0.f;
    float K;
    if (pitched) {
      K = Elem<T>::load(xr + FlatWalk(i0, W).at(pitch));
      FlatWalk w(i0 + threadIdx.x * V, W);
      const int qr = (NT * V) / W, qc = (NT * V) - qr * W;
      for (int i = i0 + threadIdx.x * V; i < i1; i += NT * V, w.step(qr, qc, W)) {
        float v[V];
        gather_run<T, V>(xr, w, W, pitch, v);
#pragma unroll
        for (int k = 0; k < V; ++k) { const float d = v[k] - K; s1 += d; s2 += d * d; }
      }
    } else {
    K = Elem<T>::load(xr + i0);
    for (int i = i0 + threadIdx.x * V; i < i1; i += NT * V) {
      if constexpr (VEC) { float v[VecIO<T>::N]; VecIO<T>::load(xr + i, v);
#pragma unroll
        for (int k = 0; k < V; ++k) { const float d = v[k] - K; s1 += d; s2 += d * d; }
      } else { const float d = Elem<T>::load(xr + i) - K; s1 += d; s2 += d * d; }
    }
    }
    const float t1 = block_sum(s1, sh), t2 = block_sum(s2, sh);
    mean = K + t1 / cnt;
    m2 = fmaxf(t2 - t1 * (t1 / cnt), 0.f);
  } else {
    // fp32 is the parity mode: exact two-sweep mean / M2 (the second sweep hits L2)
    float s = 0.f;
    for (int i = i0 + threadIdx.x * V; i < i1; i += NT * V) {
      if constexpr (VEC) { float v[VecIO<T>::N]; VecIO<T>::load(xr + i, v);
#pragma unroll
        for (int k = 0; k < V; ++k) s += v[k];
      } else s += Elem<T>::load(xr + i);
    }
    mean = block_sum(s, sh) / cnt;
    float ss = 0.f;
    for (int i = i0 + threadIdx.x * V; i < i1; i += NT * V) {
      if constexpr (VEC) { float v[VecIO<T>::N]; VecIO<T>::load(xr + i, v);
#pragma unroll
        for (int k = 0; k < V; ++k) { const float d = v[k] - mean; ss += d * d; }
      } else { const float d = Elem<T>::load(xr + i) - mean; ss += d * d; }
    }
    m2 = block_sum(ss, sh);
  }
  if (threadIdx.x == 0) {
    float* w = ws + ((size_t)row * nseg + seg) * 3;
    w[0] = cnt; w[1] = mean; w[2] = m2;
    if (fin && nseg == 1) {                          // a one-segment row: its final (mean, 1/std), the same arithmetic as the merge
      const MergeState st = {cnt, mean, m2};
      const RowStats r = norm_merge_finish(st, HW);
      fin[row] = make_float2(r.mean, r.rstd);
    }
  }
}

// rows with several segments: merge the partials in fixed order (norm_merge_full, what normalize_apply_kernel does per thread)
__global__ void stats_finalize_kernel(const float* __restrict__ ws, float2* __restrict__ fin, long long rows, int nseg, int HW) {
  const long long row = blockIdx.x * (long long)blockDim.x + threadIdx.x;
  if (row >= rows) return;
  const RowStats r = norm_merge_full(ws + (size_t)row * nseg * 3, nseg, HW);
  fin[row] = make_float2(r.mean, r.rstd);
}

// Small planes (the coarse pyramid levels: 24x80 pixels and fewer), 16-bit features: ONE WAVE per row, eight rows per
// workgroup, no workgroup barrier — the kernel above spends its time there on four __syncthreads of 512 mostly idle threads
// and runs its 3000 workgroups in two rounds (8.4 us whatever the size; this form: one round).  Same one-sweep pivot
// arithmetic; the sums are taken in another order, so both users of the statistics (normalize_forward and the fused
// loader of the cost volume) go through the same dispatch (launch_stats) and keep agreeing bit for bit.
constexpr int WAVE_ROWS = 8;
template <typename T, bool VEC>
__global__ __launch_bounds__(WAVE_ROWS * 64)
void normalize_stats_wave_kernel(const T* __restrict__ x, float* __restrict__ ws, int HW, size_t rows,
                                 const T* __restrict__ x2, size_t rows1, float2* __restrict__ fin = nullptr, int W = 0, int pitch = 0) {
  constexpr int V = VEC ? VecIO<T>::N : 1;
  const size_t row = (size_t)blockIdx.x * WAVE_ROWS + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int lane = threadIdx.x & 63;
  const bool pitched = W > 0 && pitch != W;
  const size_t pstride = pitched ? (size_t)(HW / W) * pitch : (size_t)HW;
  const T* xr = (row < rows1) ? x + row * pstride : x2 + (row - rows1) * pstride;
  const float K = Elem<T>::load(xr);
  float s1 = 0.f, s2 = 0.f;
  if (pitched) {
    FlatWalk w(lane * V, W);
    const int qr = (64 * V) / W, qc = (64 * V) - qr * W;
    for (int i = lane * V; i < HW; i += 64 * V, w.step(qr, qc, W)) {
      float v[V];
      gather_run<T, V>(xr, w, W, pitch, v);
#pragma unroll
      for (int k = 0; k < V; ++k) { const float d = v[k] - K; s1 += d; s2 += d * d; }
    }
  } else {
  for (int i = lane * V; i < HW; i += 64 * V) {
    if constexpr (VEC) { float v[VecIO<T>::N]; VecIO<T>::load(xr + i, v);
#pragma unroll
      for (int k = 0; k < V; ++k) { const float d = v[k] - K; s1 += d; s2 += d * d; }
    } else { const float d = Elem<T>::load(xr + i) - K; s1 += d; s2 += d * d; }
  }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { s1 += __shfl_xor(s1, o, 64); s2 += __shfl_xor(s2, o, 64); }
  if (lane == 0) {
    const float cnt = (float)HW;
    float* w = ws + row * 3;
    const float mean = K + s1 / cnt, m2 = fmaxf(s2 - s1 * (s1 / cnt), 0.f);
    w[0] = cnt; w[1] = mean; w[2] = m2;
    if (fin) {
      const MergeState st = {cnt, mean, m2};
      const RowStats r = norm_merge_finish(st, HW);
      fin[row] = make_float2(r.mean, r.rstd);
    }
  }
}

// fp32 (the parity mode): (x - mean) / std with an IEEE division, the reference's arithmetic (model/upflow.py:130-134).
// 16-bit: (x - mean) * (1/std) — the fused loader of the cost volume (corr81_allc_kernel.hpp, NORM) computes exactly
// this, so the fused and the two-kernel paths agree bit for bit; the difference to the division (<= 1 ulp of fp32)
// is far below the 16-bit output rounding.
template <typename T>
__device__ __forceinline__ float apply1(float v, float mean, float std, float rstd) {
  if constexpr (sizeof(typename Elem<T>::store_t) == 4) return (v - mean) / std;
  else return (v - mean) * rstd;
}

template <typename T, bool VEC>
__global__ __launch_bounds__(NT)
void normalize_apply_kernel(const T* __restrict__ x, T* __restrict__ y, const float* __restrict__ ws,
                            float* __restrict__ mean_out, float* __restrict__ rstd_out, int HW, int nseg, int seglen) {
  constexpr int V = VEC ? VecIO<T>::N : 1;
  const size_t row = blockIdx.x / nseg;
  const int seg = blockIdx.x - (int)row * nseg;
  // merge the partials (every thread redundantly: nseg is small and the values are L2-resident)
  const RowStats rs = norm_merge_full(ws + (size_t)row * nseg * 3, nseg, HW);
  const float mean = rs.mean, std = rs.std, rstd = rs.rstd;
  const int i0 = seg * seglen, i1 = min(HW, i0 + seglen);
  const T* xr = x + row * HW;
  T* yr = y + row * HW;
  for (int i = i0 + threadIdx.x * V; i < i1; i += NT * V) {
    if constexpr (VEC) { float v[VecIO<T>::N]; VecIO<T>::load(xr + i, v);
#pragma unroll
      for (int k = 0; k < V; ++k) v[k] = apply1<T>(v[k], mean, std, rstd);
      VecIO<T>::store(yr + i, v);
    } else Elem<T>::store(yr + i, apply1<T>(Elem<T>::load(xr + i), mean, std, rstd));
  }
  if (threadIdx.x == 0 && seg == 0) {
    if (mean_out) mean_out[row] = mean;
    if (rstd_out) rstd_out[row] = rstd;
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

// One pixel per thread on a 2-D grid (a wave = 64 consecutive pixels of one row, four rows per workgroup): no integer
// division per thread.  (4 pixels per thread, consecutive or strided, were measured slower: the kernel is bound by its
// ~300 instructions per pixel — four IEEE divisions of the exact-rounding sampling positions — not by latency.)
__global__ __launch_bounds__(256)
void occ_check_kernel(const float* __restrict__ ff, const float* __restrict__ fb, float* __restrict__ occ_fw,
                      float* __restrict__ occ_bw, int H, int W, float a1, float a2, SampleGeom geo) {
  const int HW = H * W;
  const int j = blockIdx.x * 64 + (threadIdx.x & 63);
  const int i = __builtin_amdgcn_readfirstlane(blockIdx.y * 4 + (threadIdx.x >> 6));
  if (i >= H || j >= W) return;
  const int n = blockIdx.z;
  const int p = i * W + j;
  const float* F = ff + (size_t)n * 2 * HW;
  const float* Bk = fb + (size_t)n * 2 * HW;
  const float fx = F[p], fy = F[HW + p], bx = Bk[p], by = Bk[HW + p];
  const float mag = (fabsf(fx) + fabsf(fy)) + (fabsf(bx) + fabsf(by));      // tools.py:559, :573
  const float thr = a1 * mag + a2;                                          // :578
  float wx, wy;
  warp2(Bk, HW, make_taps(j, i, fx, fy, H, W, geo), H, W, wx, wy);               // flow_bw warped by flow_fw, :574
  const bool cf = (fabsf(fx + wx) + fabsf(fy + wy)) < thr;                  // :576, :579
  warp2(F, HW, make_taps(j, i, bx, by, H, W, geo), H, W, wx, wy);                // :575
  const bool cb = (fabsf(bx + wx) + fabsf(by + wy)) < thr;
  // outgoing mask (tools.py:657-667) and obj merge (:672-676): 1 where consistent OR flow leaves the image
  const float pxf = (float)j + fx, pyf = (float)i + fy, pxb = (float)j + bx, pyb = (float)i + by;
  const bool inf_ = !(pxf > (float)(W - 1)) && !(pxf < 0.f) && !(pyf > (float)(H - 1)) && !(pyf < 0.f);
  const bool inb_ = !(pxb > (float)(W - 1)) && !(pxb < 0.f) && !(pyb > (float)(H - 1)) && !(pyb < 0.f);
  occ_fw[(size_t)n * HW + p] = (cf || !inf_) ? 1.f : 0.f;
  occ_bw[(size_t)n * HW + p] = (cb || !inb_) ? 1.f : 0.f;
}


// ---- soft census (ternary) distance  (utils/loss.py:50-91) -------------------------------------
// The reference builds the 7x7 neighbourhoods with a 49-channel identity conv2d, i.e. two [B,49,H,W] tensors per call
// and ~10 element-wise passes over them (and their autograd twins).  Here the whole per-pixel distance is one launch:
//     t_k(I, p) = u / sqrt(0.81 + u^2),  u = I(p+k) - I(p)   (I = grey image, zero padded, k over the (2R+1)^2 offsets)
//     dist(p)   = sum_k d_k / (0.1 + d_k),  d_k = (t_k(I1,p) - t_k(I2,p))^2
// and the backward is a GATHER (deterministic, no atomics): I(q) appears in dist(q) as the centre of all its terms and
// in dist(q-k) as the neighbour of offset k, so  dI(q) = sum_k [ G(q-k) c_k(q-k) ] - G(q) sum_k c_k(q)  with
// c_k(p) = d dist(p) / d u_k, every factor recomputed from the two grey images.
// t(u) = u / sqrt(0.81 + u^2) and dt/du = 0.81 / (0.81 + u^2)^(3/2) from ONE hardware reciprocal square root (1 ulp):
// the first version spent most of its time in IEEE sqrt / division sequences (per neighbour 4 square roots and 5 divisions
// in the backward pass); the loss tolerances of the reference's goldens (2e-6) leave three orders of magnitude of room.
__device__ __forceinline__ float census_r(float u) { return __builtin_amdgcn_rsqf(0.81f + u * u); }

__global__ __launch_bounds__(256)
void census_fwd_kernel(const float* __restrict__ g1, const float* __restrict__ g2, float* __restrict__ dist, int H, int W, int R) {
  const int HW = H * W;
  const int p = blockIdx.x * 256 + threadIdx.x;
  if (p >= HW) return;
  const int n = blockIdx.y;
  const int i = p / W, j = p - i * W;
  const float* a = g1 + (size_t)n * HW;
  const float* b = g2 + (size_t)n * HW;
  const float ca = a[p], cb = b[p];
  float acc = 0.f;
  for (int dy = -R; dy <= R; ++dy) {
    const int y = i + dy;
    const bool iny = y >= 0 && y < H;
    for (int dx = -R; dx <= R; ++dx) {
      const int x = j + dx;
      const bool in = iny && x >= 0 && x < W;
      const float va = in ? a[y * W + x] : 0.f, vb = in ? b[y * W + x] : 0.f;
      const float u1 = va - ca, u2 = vb - cb;
      const float t = u1 * census_r(u1) - u2 * census_r(u2);
      const float d = t * t;
      acc += d * __builtin_amdgcn_rcpf(0.1f + d);
    }
  }
  dist[(size_t)n * HW + p] = acc;
}

// d dist / d u1 and d dist / d u2 of one term
__device__ __forceinline__ void census_c(float u1, float u2, float& c1, float& c2) {
  const float r1 = census_r(u1), r2 = census_r(u2);
  const float df = u1 * r1 - u2 * r2, d = df * df;
  const float inv = __builtin_amdgcn_rcpf(0.1f + d);
  const float k = (0.1f * inv * inv) * 2.f * df * 0.81f;         // d/dd of d/(0.1+d), times 2 df, times the 0.81 of dt/du
  c1 = k * (r1 * r1 * r1);
  c2 = -k * (r2 * r2 * r2);
}

// grad wrt g1 (W1) and / or g2 (W2); G = grad of dist [B,1,H,W]
template <bool W1, bool W2>
__global__ __launch_bounds__(256)
void census_bwd_kernel(const float* __restrict__ g1, const float* __restrict__ g2, const float* __restrict__ G,
                       float* __restrict__ gg1, float* __restrict__ gg2, int H, int W, int R) {
  const int HW = H * W;
  const int q = blockIdx.x * 256 + threadIdx.x;
  if (q >= HW) return;
  const int n = blockIdx.y;
  const int i = q / W, j = q - i * W;
  const float* a = g1 + (size_t)n * HW;
  const float* b = g2 + (size_t)n * HW;
  const float* Gn = G + (size_t)n * HW;
  const float aq = a[q], bq = b[q], Gq = Gn[q];
  // q is the CENTRE of its own 49 terms (neighbour value at q+k, zero outside the image: du/dI(q) = -1) and the NEIGHBOUR of offset k
  // of every centre p = q - k inside the image (du/dI(q) = +1, u = I(q) - I(p)).  The second family needs no evaluation of its own:
  // with k' = -k it is the pair (q, q+k') seen from the other end, u -> -u, and c is ODD in (u1, u2) — every operation of census_c
  // commutes with the sign exactly — so
  //     dI(q) = - sum_k c_k(q) * ( G(q) + [q+k inside] G(q+k) )
  // (round 5: half the evaluations — 2 rsq + 1 rcp each — and a third of the loads of the first version: 88 -> 45 us).
  float s1 = 0.f, s2 = 0.f;
  for (int dy = -R; dy <= R; ++dy)
    for (int dx = -R; dx <= R; ++dx) {
      const int y = i + dy, x = j + dx;
      const bool in = y >= 0 && y < H && x >= 0 && x < W;
      const int pp = in ? y * W + x : q;
      const float va = in ? a[pp] : 0.f, vb = in ? b[pp] : 0.f;
      float c1, c2;
      census_c(va - aq, vb - bq, c1, c2);
      const float gsum = Gq + (in ? Gn[pp] : 0.f);
      if (W1) s1 -= gsum * c1;
      if (W2) s2 -= gsum * c2;
    }
  if (W1) gg1[(size_t)n * HW + q] = s1;
  if (W2) gg2[(size_t)n * HW + q] = s2;
}

// xs[n, c*4 + p*2 + q, i, j] = x[n, c, 2i+p, 2j+q] (F.pixel_unshuffle(x, 2)) and its inverse, 16-bit elements: the layout glue
// of the stride-2 layers' gradients (ops.py: _s2d_ok).  A thread moves V input pixels of one row (V = 8: one 16-byte access
// split into / merged from the two column phases; V = 2 for widths that are not multiples of 8).
template <int V, bool INVERSE>
__global__ void space_to_depth2_kernel(const uint16_t* __restrict__ src, uint16_t* __restrict__ dst, int C, int H, int W, long long total) {
  const long long t = blockIdx.x * (long long)blockDim.x + threadIdx.x;
  if (t >= total) return;
  const int wv = W / V;
  const int jv = (int)(t % wv), r = (int)((t / wv) % H);
  const long long nc = t / ((long long)wv * H);               // n * C + c
  const int i = r >> 1, p = r & 1, h2 = H / 2, w2 = W / 2;
  const uint16_t* full = (INVERSE ? dst : src) + ((size_t)nc * H + r) * W + (size_t)jv * V;            // x row segment
  const uint16_t* q0c = (INVERSE ? src : dst) + (((size_t)nc * 4 + p * 2) * h2 + i) * w2 + (size_t)jv * (V / 2);   // phase q = 0
  const uint16_t* q1c = q0c + (size_t)h2 * w2;                                                                       // phase q = 1
  uint16_t* fullw = const_cast<uint16_t*>(full); uint16_t* q0 = const_cast<uint16_t*>(q0c); uint16_t* q1 = const_cast<uint16_t*>(q1c);
  if constexpr (V == 8) {
    if constexpr (!INVERSE) {
      const uint4 v = *reinterpret_cast<const uint4*>(full);
      uint2 e, o;
      e.x = __builtin_amdgcn_perm(v.y, v.x, 0x05040100u); o.x = __builtin_amdgcn_perm(v.y, v.x, 0x07060302u);
      e.y = __builtin_amdgcn_perm(v.w, v.z, 0x05040100u); o.y = __builtin_amdgcn_perm(v.w, v.z, 0x07060302u);
      *reinterpret_cast<uint2*>(q0) = e;
      *reinterpret_cast<uint2*>(q1) = o;
    } else {
      const uint2 e = *reinterpret_cast<const uint2*>(q0c), o = *reinterpret_cast<const uint2*>(q1c);
      uint4 v;
      v.x = (e.x & 0xffffu) | (o.x << 16); v.y = (e.x >> 16) | (o.x & 0xffff0000u);
      v.z = (e.y & 0xffffu) | (o.y << 16); v.w = (e.y >> 16) | (o.y & 0xffff0000u);
      *reinterpret_cast<uint4*>(fullw) = v;
    }
  } else {
    if constexpr (!INVERSE) { q0[0] = full[0]; q1[0] = full[1]; }
    else { fullw[0] = q0c[0]; fullw[1] = q1c[0]; }
  }
}

}  // namespace misc
}  // namespace upf

extern "C" int upf_space_to_depth2(const void* src, void* dst, int B, int C, int H, int W, int inverse, int dtype, void* stream) {
  using namespace upf;
  UPF_REQUIRE(src && dst && B > 0 && C > 0 && H > 0 && W > 0 && H % 2 == 0 && W % 2 == 0, UPF_EINVAL, "space_to_depth2: even H, W expected (H %d, W %d)", H, W);
  UPF_REQUIRE(dtype == UPF_BF16 || dtype == UPF_F16, UPF_EDTYPE, "space_to_depth2: bf16 / fp16 only");
  const bool v8 = W % 16 == 0 && aligned_to(src, 16) && aligned_to(dst, 16);
  const long long total = (long long)B * C * H * (W / (v8 ? 8 : 2));
  UPF_REQUIRE((total + 255) / 256 < (1ll << 31), UPF_EINVAL, "space_to_depth2: tensor too large");
  const dim3 grid((unsigned)((total + 255) / 256));
  hipStream_t s = (hipStream_t)stream;
  const uint16_t* a = (const uint16_t*)src; uint16_t* b = (uint16_t*)dst;
  if (v8) {
    if (inverse) hipLaunchKernelGGL((misc::space_to_depth2_kernel<8, true>), grid, dim3(256), 0, s, a, b, C, H, W, total);
    else hipLaunchKernelGGL((misc::space_to_depth2_kernel<8, false>), grid, dim3(256), 0, s, a, b, C, H, W, total);
  } else {
    if (inverse) hipLaunchKernelGGL((misc::space_to_depth2_kernel<2, true>), grid, dim3(256), 0, s, a, b, C, H, W, total);
    else hipLaunchKernelGGL((misc::space_to_depth2_kernel<2, false>), grid, dim3(256), 0, s, a, b, C, H, W, total);
  }
  return check_launch("space_to_depth2");
}

// segments per row: enough workgroups for ~2 per CU (round 4: 512, was 1024 — a one-segment row gets its final statistics from
// the statistics kernel itself, no merge launch), at least 2048 elements per segment
static int normalize_nseg(long long N, int HW) {
  int nseg = 1;
  while (N * nseg < 512 && HW / (nseg * 2) >= 2048) nseg *= 2;
  return nseg;
}

// small planes of 16-bit features: the one-wave-per-row kernel (see normalize_stats_wave_kernel)
static bool stats_wave_form(int nseg, int HW, int dtype) { return nseg == 1 && HW <= 4096 && dtype != UPF_F32; }
template <typename T>
static void launch_stats_wave(const T* x1, const T* x2, float* ws, size_t rows, size_t rows1, int HW, bool vec, hipStream_t stream, float2* fin = nullptr,
                              int W = 0, int pitch = 0) {
  const unsigned grid = (unsigned)((rows + upf::misc::WAVE_ROWS - 1) / upf::misc::WAVE_ROWS);
  if (vec) hipLaunchKernelGGL((upf::misc::normalize_stats_wave_kernel<T, true>), dim3(grid), dim3(upf::misc::WAVE_ROWS * 64), 0, stream, x1, ws, HW, rows, x2, rows1, fin, W, pitch);
  else hipLaunchKernelGGL((upf::misc::normalize_stats_wave_kernel<T, false>), dim3(grid), dim3(upf::misc::WAVE_ROWS * 64), 0, stream, x1, ws, HW, rows, x2, rows1, fin, W, pitch);
}

// statistics of TWO [N,HW] tensors in one launch -> ws[(2N rows)][nseg][3] partials and fin[2N] final (mean, 1/std) pairs (the
// statistics kernel writes them itself where a row is one segment; a small second launch merges the partials otherwise);
// returns nseg (internal.hpp)
int upf::misc::launch_stats2(const void* x1, const void* x2, float* ws, float2* fin, long long N, int HW, int dtype, hipStream_t stream, int W, int pitch) {
  const int nseg = normalize_nseg(2 * N, HW);
  const int seglen = cdiv(HW, nseg);
  const unsigned grid = (unsigned)(2 * N * nseg);
  const int vn = (dtype == UPF_F32) ? 4 : 8;
  const bool pitched = W > 0 && pitch != W;
  // (pitched planes: the summation order of the CONTIGUOUS form of the same shape — 8-element runs when its rows would allow vector
  //  loads — with gathered addresses)
  const bool vec = (HW % vn == 0) && (seglen % vn == 0) && (pitched || (aligned_to(x1, 16) && aligned_to(x2, 16)));
  if (!pitched) { W = 0; pitch = 0; }
  if (stats_wave_form(nseg, HW, dtype)) {
    if (dtype == UPF_BF16) launch_stats_wave<bf16_t>((const bf16_t*)x1, (const bf16_t*)x2, ws, (size_t)(2 * N), (size_t)N, HW, vec, stream, fin, W, pitch);
    else launch_stats_wave<f16_t>((const f16_t*)x1, (const f16_t*)x2, ws, (size_t)(2 * N), (size_t)N, HW, vec, stream, fin, W, pitch);
    return nseg;
  }
  UPF_DISPATCH(dtype, T,
               if (vec) hipLaunchKernelGGL((misc::normalize_stats_kernel<T, true>), dim3(grid), dim3(misc::NT), 0, stream, (const T*)x1, ws, HW, nseg, seglen, (const T*)x2, (size_t)N, fin, W, pitch);
               else hipLaunchKernelGGL((misc::normalize_stats_kernel<T, false>), dim3(grid), dim3(misc::NT), 0, stream, (const T*)x1, ws, HW, nseg, seglen, (const T*)x2, (size_t)N, fin, W, pitch));
  if (fin && nseg > 1)
    hipLaunchKernelGGL(misc::stats_finalize_kernel, dim3((unsigned)((2 * N + 255) / 256)), dim3(256), 0, stream, (const float*)ws, fin, 2 * N, nseg, HW);
  return nseg;
}
int upf::misc::stats2_nseg(long long N, int HW) { return normalize_nseg(2 * N, HW); }

extern "C" long long upf_normalize_workspace_bytes(long long N, int HW) {
  return (long long)N * normalize_nseg(N, HW) * 3 * sizeof(float);
}

extern "C" int upf_normalize_forward(const void* x, void* y, float* mean, float* rstd, void* workspace, long long N, int HW,
                                     int dtype, void* stream) {
  using namespace upf;
  UPF_REQUIRE(x && y && workspace, UPF_EINVAL, "normalize_forward: null pointer");
  UPF_REQUIRE(N > 0 && HW > 0, UPF_EINVAL, "normalize_forward: bad shape N=%lld HW=%d", N, HW);
  const int nseg = normalize_nseg(N, HW);
  const int seglen = cdiv(HW, nseg);
  UPF_REQUIRE(N * nseg < (1ll << 31), UPF_EINVAL, "normalize_forward: grid too large");
  const unsigned grid = (unsigned)(N * nseg);
  const int vn = (dtype == UPF_F32) ? 4 : 8;
  const bool vec = (HW % vn == 0) && (seglen % vn == 0) && aligned_to(x, 16) && aligned_to(y, 16);
#define UPF_NORM_LAUNCH(VEC)                                                                                                   \
  if (stats_wave_form(nseg, HW, dtype)) launch_stats_wave<T>((const T*)x, (const T*)nullptr, (float*)workspace, (size_t)N, (size_t)N, HW, VEC, (hipStream_t)stream); \
  else hipLaunchKernelGGL((misc::normalize_stats_kernel<T, VEC>), dim3(grid), dim3(misc::NT), 0, (hipStream_t)stream, (const T*)x,  \
                     (float*)workspace, HW, nseg, seglen);                                                                     \
  hipLaunchKernelGGL((misc::normalize_apply_kernel<T, VEC>), dim3(grid), dim3(misc::NT), 0, (hipStream_t)stream, (const T*)x,  \
                     (T*)y, (const float*)workspace, mean, rstd, HW, nseg, seglen)
  UPF_DISPATCH(dtype, T, if (vec) { UPF_NORM_LAUNCH(true); } else { UPF_NORM_LAUNCH(false); });
#undef UPF_NORM_LAUNCH
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
  UPF_REQUIRE(H <= 4 * 65535, UPF_EINVAL, "occ_check: image too tall");
  dim3 grid(cdiv(W, 64), cdiv(H, 4), B);
  hipLaunchKernelGGL(misc::occ_check_kernel, grid, dim3(256), 0, (hipStream_t)stream, flow_f, flow_b, occ_fw, occ_bw, H, W, alpha1, alpha2, make_sample_geom(H, W));
  return check_launch("occ_check");
}

extern "C" int upf_census_forward(const float* gray1, const float* gray2, float* dist, int B, int H, int W, int max_distance, void* stream) {
  using namespace upf;
  UPF_REQUIRE(gray1 && gray2 && dist, UPF_EINVAL, "census_forward: null pointer");
  UPF_REQUIRE(B > 0 && B <= 65535 && H > 0 && W > 0 && max_distance >= 1 && max_distance <= 8, UPF_EINVAL, "census_forward: bad shape / max_distance");
  dim3 grid(cdiv(H * W, 256), B);
  hipLaunchKernelGGL(misc::census_fwd_kernel, grid, dim3(256), 0, (hipStream_t)stream, gray1, gray2, dist, H, W, max_distance);
  return check_launch("census_forward");
}

extern "C" int upf_census_backward(const float* gray1, const float* gray2, const float* grad_dist, float* g_gray1, float* g_gray2,
                                   int B, int H, int W, int max_distance, void* stream) {
  using namespace upf;
  UPF_REQUIRE(gray1 && gray2 && grad_dist && (g_gray1 || g_gray2), UPF_EINVAL, "census_backward: null pointer");
  UPF_REQUIRE(B > 0 && B <= 65535 && H > 0 && W > 0 && max_distance >= 1 && max_distance <= 8, UPF_EINVAL, "census_backward: bad shape / max_distance");
  dim3 grid(cdiv(H * W, 256), B);
  hipStream_t s = (hipStream_t)stream;
  if (g_gray1 && g_gray2) hipLaunchKernelGGL((misc::census_bwd_kernel<true, true>), grid, dim3(256), 0, s, gray1, gray2, grad_dist, g_gray1, g_gray2, H, W, max_distance);
  else if (g_gray1) hipLaunchKernelGGL((misc::census_bwd_kernel<true, false>), grid, dim3(256), 0, s, gray1, gray2, grad_dist, g_gray1, g_gray2, H, W, max_distance);
  else hipLaunchKernelGGL((misc::census_bwd_kernel<false, true>), grid, dim3(256), 0, s, gray1, gray2, grad_dist, g_gray1, g_gray2, H, W, max_distance);
  return check_launch("census_backward");
}

// ---- self-test of the division-free quotient of sampling.hpp (div_by_const) ---------------------------------------------
namespace upf { namespace misc {
__global__ void div_selftest_kernel(float d, float r, unsigned long long* __restrict__ bad) {
  unsigned long long local = 0;
  for (unsigned long long i = blockIdx.x * (unsigned long long)blockDim.x + threadIdx.x; i < (1ull << 32); i += (unsigned long long)gridDim.x * blockDim.x) {
    const float x = __uint_as_float((unsigned int)i);
    const float want = __fdiv_rn(x, d), got = div_by_const(x, d, r);
    if (__float_as_uint(got) != __float_as_uint(want) && !(got != got && want != want)) ++local;
  }
  if (local) atomicAdd(bad, local);
}
}}  // namespace upf::misc

extern "C" int upf_div_selftest(int size, unsigned long long* mismatches /* device, zeroed by the caller */, void* stream) {
  using namespace upf;
  UPF_REQUIRE(size >= 1 && mismatches, UPF_EINVAL, "div_selftest: bad arguments");
  const SampleGeom g = make_sample_geom(size, size);                       // the reciprocal exactly as the launchers compute it
  const float d = (float)(size - 1 > 1 ? size - 1 : 1);
  hipLaunchKernelGGL(misc::div_selftest_kernel, dim3(4096), dim3(256), 0, (hipStream_t)stream, d, g.rW, mismatches);
  return check_launch("div_selftest");
}
