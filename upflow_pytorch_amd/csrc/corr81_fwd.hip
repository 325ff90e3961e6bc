// 81-neighbour cost volume, forward — hand-written for gfx950 (MI355X / CDNA4).
//
// Replaces correlation_forward<T> + 2x channels_first<T> of the reference
// (/root/reference/model/correlation_package/correlation_cuda_kernel.cu:15-114, :302-393), which run
// one 32-thread block per output pixel, 81 barrier+serial-reduce rounds, on padded NHWC copies.
//
// Two kernels, both reading NCHW directly (no NHWC staging pass):
//   corr81_fwd_kernel.hpp   wave-level VALU kernel — fp32 (the parity mode) and ragged/unaligned 16-bit
//     * workgroup = 9 waves = one 8x32 pixel tile; wave w owns displacement row dy = w-4, a lane owns 4
//       consecutive pixels of one row and keeps 9(dx) x 4(px) fp32 accumulators;
//     * LDS holds the f1 tile and the f2 tile + 4-pixel halo for a chunk of KC "k-slots" (one fp32
//       channel, or a PAIR of 16-bit channels interleaved per pixel so one v_dot2c retires two channels),
//       double-buffered: the next chunk's global loads are in flight during the MAC loop of this one;
//     * aligned inputs are staged with raw buffer loads whose bounds check supplies every zero (halo
//       outside the image, channels beyond C); per k-slot a lane issues 4 ds_read_b128 for 36 MACs;
//       the lane -> (row, x-block) map follows the hardware's 16-lane ds_read_b128 service groups;
//   corr81_mfma_kernel.hpp  matrix-core kernel — bf16/fp16 with W % 8 == 0 (every large level)
//     * the VALU kernel is bound by v_dot2c's half-rate issue; the 16-block 4x4x4 MFMA does the same
//       channel contraction 4.4x faster with 75 % of its products wanted (dense 12-candidate band);
//     * finished units are de-skewed, transposed through a per-wave LDS patch and stored while the next
//       unit computes, so HBM writes overlap the matrix work.
// Common: epilogue divides by C (like `reduce_sum / nelems`, correlation_cuda_kernel.cu:108), optionally
// applies LeakyReLU (model/upflow.py:563-564), writes 4-8 pixels per store, optionally into a wider
// channel buffer (out_batch_stride); blockIdx is remapped so each XCD (private L2) gets a contiguous
// run of tiles.
#include "corr81_fwd_kernel.hpp"
#include "corr81_mfma_kernel.hpp"
#include "corr81_allc_kernel.hpp"
#include "internal.hpp"
#include <hip/hip_ext.h>
#include <stdlib.h>
#include <string.h>

namespace upf {
namespace corr {

#ifndef UPF_CORR_KC
#define UPF_CORR_KC 4
#endif

template <typename T, int KC>
void launch_kc(bool aligned, unsigned nblocks, hipStream_t stream, hipEvent_t ev0, hipEvent_t ev1, const T* f1, const T* f2,
               T* out, int C, int H, int W, int tiles_x, int tiles_y, long long out_bs, float slope) {
  static LdsOptIn opt_a, opt_u;   // > 48 KiB of dynamic LDS: opted into per kernel and per device
  opt_a.ensure(reinterpret_cast<const void*>(&corr81_fwd_kernel<T, true, KC>), lds_bytes(KC));
  opt_u.ensure(reinterpret_cast<const void*>(&corr81_fwd_kernel<T, false, KC>), lds_bytes(KC));
  // hipExtLaunchKernelGGL == hipLaunchKernelGGL plus optional start/stop events recorded right around
  // THIS kernel on its stream (used by upf_corr81_forward_timed; null events = plain launch)
  if (aligned)
    hipExtLaunchKernelGGL((corr81_fwd_kernel<T, true, KC>), dim3(nblocks), dim3(NTHREADS), lds_bytes(KC), stream, ev0, ev1, 0,
                          f1, f2, out, C, H, W, tiles_x, tiles_y, out_bs, slope);
  else
    hipExtLaunchKernelGGL((corr81_fwd_kernel<T, false, KC>), dim3(nblocks), dim3(NTHREADS), lds_bytes(KC), stream, ev0, ev1, 0,
                          f1, f2, out, C, H, W, tiles_x, tiles_y, out_bs, slope);
}

// bf16 / fp16 with W % 8 == 0: matrix-core kernel (corr81_mfma_kernel.hpp)
template <typename T, bool SINGLE>
void launch_mfma(unsigned nblocks, hipStream_t stream, hipEvent_t ev0, hipEvent_t ev1, const T* f1, const T* f2, T* out,
                 int C, int H, int W, int tiles_x, int tiles_y, long long out_bs, float slope) {
  static LdsOptIn opt;
  opt.ensure(reinterpret_cast<const void*>(&corrm::corr81_mfma_kernel<T, SINGLE>), corrm::LDS_BYTES);
  hipExtLaunchKernelGGL((corrm::corr81_mfma_kernel<T, SINGLE>), dim3(nblocks), dim3(corrm::NTHREADS), corrm::LDS_BYTES, stream, ev0, ev1, 0,
                        f1, f2, out, C, H, W, tiles_x, tiles_y, out_bs, slope);
}

// ---- bf16 / fp16, C <= 208: the whole channel depth resident in LDS (corr81_allc_kernel.hpp) ----------------------
static int g_variant = -1;     // upf_corr_set_option("variant"): -1 = choose, 0..3 = force where it fits
static int g_old_path = 0;     // upf_corr_set_option("old_path"): 1 = the chunked round-1 kernels (A/B runs)
struct AllcVariant { int uw, nu, nt; };
constexpr AllcVariant ALLC[] = {{32, 4, 4}, {32, 2, 8}, {32, 1, 8}, {16, 1, 8}};
constexpr int NALLC = 4;
constexpr size_t LDS_MAX = 160 * 1024;

template <int UW, int NU, int NT>
static bool allc_fits(int KQ, bool ragged, bool norm) {
  return corrx::ntasks<UW, NU>(KQ) <= NT * corrx::NTHREADS && corrx::lds_bytes<UW, NU>(KQ, ragged, norm) <= LDS_MAX;
}
static bool allc_fits(int v, int KQ, bool ragged, bool norm) {
  switch (v) {
    case 0: return allc_fits<32, 4, 4>(KQ, ragged, norm);
    case 1: return allc_fits<32, 2, 8>(KQ, ragged, norm);
    case 2: return allc_fits<32, 1, 8>(KQ, ragged, norm);
    case 3: return allc_fits<16, 1, 8>(KQ, ragged, norm);
  }
  return false;
}
static long long allc_nwg(int v, int B, int H, int W) {
  const int th = ALLC[v].nu * (64 / ALLC[v].uw);
  return (long long)B * cdiv(H, th) * cdiv(W, ALLC[v].uw);
}
static size_t allc_lds(int v, int KQ, bool ragged, bool norm) {
  switch (v) {
    case 0: return corrx::lds_bytes<32, 4>(KQ, ragged, norm);
    case 1: return corrx::lds_bytes<32, 2>(KQ, ragged, norm);
    case 2: return corrx::lds_bytes<32, 1>(KQ, ragged, norm);
    default: return corrx::lds_bytes<16, 1>(KQ, ragged, norm);
  }
}
// Tile geometry by shape (measured on MI355X, tools/corr_levels.py -> profiles/r02_corr81_levels.txt):
//   * never a geometry whose grid needs a second round of workgroups when a smaller tile fits in one round
//     (workgroups per CU = min(2, 160 KB / LDS per workgroup));
//   * with >= 200 workgroups in one round: the LARGEST such tile (least halo staging per output pixel);
//   * fewer than that (the coarse levels): the geometry with the MOST workgroups — the kernel is a latency chain there
//     and a smaller tile shortens it;
//   * everything needs several rounds (the large levels): the largest tile.
// -1: nothing fits (C > 208).  -2: C > 40 on a grid of >= 160 8x32 tiles — the channel-chunked kernel with its larger
// tile stages 36 % fewer bytes per pixel and wins there (1/8-resolution level of the large configurations).
static int allc_pick(int B, int C, int H, int W, bool ragged, bool norm, bool chunked_ok) {
  const int forced = g_variant;
  const int KQ = (C + 3) / 4;
  if (forced >= 0 && forced < NALLC && allc_fits(forced, KQ, ragged, norm)) return forced;
  if (chunked_ok && !norm && C > 40 && (long long)B * cdiv(H, 8) * cdiv(W, 32) >= 160) return -2;
  int first = -1, big = -1, most = -1;
  long long most_n = -1;
  for (int v = 0; v < NALLC; ++v) {
    if (!allc_fits(v, KQ, ragged, norm)) continue;
    if (first < 0) first = v;
    const long long nwg = allc_nwg(v, B, H, W);
    const long long per_cu = LDS_MAX / allc_lds(v, KQ, ragged, norm) >= 2 ? 2 : 1;
    if (nwg > 256 * per_cu) continue;                              // more than one round
    if (nwg >= 200 && big < 0) big = v;
    if (nwg > most_n) { most = v; most_n = nwg; }
  }
  if (big >= 0) return big;
  if (most >= 0) return most;
  return first;
}

template <typename T, int UW, int NU, int NT, bool RAGGED, bool NORM, bool OC8 = false, bool PADW = false, typename TO = T>
int launch_allc_one(const T* f1, const T* f2, TO* out, int B, int C, int H, int W, long long out_bs, float slope,
                    const float* ws1, const float* ws2, int nseg, hipStream_t stream, hipEvent_t ev0, hipEvent_t ev1, int fpitch = 0) {
  if (fpitch == 0) fpitch = W;
  using G = corrx::Geo<UW, NU>;
  const int tiles_x = cdiv(W, G::TW), tiles_y = cdiv(H, G::TH);
  const long long nblocks = (long long)B * tiles_x * tiles_y;
  UPF_REQUIRE(nblocks < (1ll << 31), UPF_EINVAL, "corr81_forward: grid too large");
  const size_t lds = corrx::lds_bytes<UW, NU>((C + 3) / 4, RAGGED, NORM);
  static LdsOptIn opt;
  auto kern = &corrx::corr81_allc_kernel<T, UW, NU, NT, RAGGED, NORM, 1, OC8, PADW, TO>;
  opt.ensure(reinterpret_cast<const void*>(kern), lds);
  // (the timed entry points pass start / stop events -> hipExtLaunchKernel; everything else takes the ordinary launch path)
  if (ev0 || ev1)
    hipExtLaunchKernelGGL(kern, dim3((unsigned)nblocks), dim3(corrx::NTHREADS), lds, stream, ev0, ev1, 0,
                          f1, f2, out, C, H, W, tiles_x, tiles_y, out_bs, slope, ws1, ws2, nseg, (int)nblocks, fpitch);
  else
    hipLaunchKernelGGL(kern, dim3((unsigned)nblocks), dim3(corrx::NTHREADS), lds, stream,
                       f1, f2, out, C, H, W, tiles_x, tiles_y, out_bs, slope, ws1, ws2, nseg, (int)nblocks, fpitch);
  return check_launch("corr81_forward");
}

template <typename T, bool RAGGED, bool NORM, typename TO = T>
int launch_allc(int v, const T* f1, const T* f2, TO* out, int B, int C, int H, int W, long long out_bs, float slope,
                const float* ws1, const float* ws2, int nseg, hipStream_t stream, hipEvent_t ev0, hipEvent_t ev1, int fpitch = 0) {
#define UPF_ALLC(UW, NU, NT) return launch_allc_one<T, UW, NU, NT, RAGGED, NORM, false, false, TO>(f1, f2, out, B, C, H, W, out_bs, slope, ws1, ws2, nseg, stream, ev0, ev1, fpitch)
  switch (v) {
    case 0: UPF_ALLC(32, 4, 4);
    case 1: UPF_ALLC(32, 2, 8);
    case 2: UPF_ALLC(32, 1, 8);
    case 3: UPF_ALLC(16, 1, 8);
  }
#undef UPF_ALLC
  set_error("corr81_forward: internal routing error (variant %d)", v);
  return UPF_EUNSUPPORTED;
}

// normalising cost volume into channel octets (upf_corr81_norm_forward_c8): !RAGGED, NORM, OC8
// (PADW: ragged logical W on pitched rows, corr81_allc_kernel.hpp)
template <typename T, bool PADW = false, typename TO = T>
int launch_allc_c8(int v, const T* f1, const T* f2, TO* out, int B, int C, int H, int W, long long out_bs, float slope,
                   const float* ws1, const float* ws2, int nseg, hipStream_t stream, hipEvent_t ev0 = nullptr, hipEvent_t ev1 = nullptr, int fpitch = 0) {
#define UPF_ALLC(UW, NU, NT) return launch_allc_one<T, UW, NU, NT, false, true, true, PADW, TO>(f1, f2, out, B, C, H, W, out_bs, slope, ws1, ws2, nseg, stream, ev0, ev1, fpitch)
  switch (v) {
    case 0: UPF_ALLC(32, 4, 4);
    case 1: UPF_ALLC(32, 2, 8);
    case 2: UPF_ALLC(32, 1, 8);
    case 3: UPF_ALLC(16, 1, 8);
  }
#undef UPF_ALLC
  set_error("corr81_norm_forward_c8: internal routing error (variant %d)", v);
  return UPF_EUNSUPPORTED;
}

// -> UPF_OK / error, or 1 = "not applicable" (C too deep, item too large): the caller takes the chunked kernels
template <typename T, bool NORM, typename TO = T>
int try_allc(const void* f1, const void* f2, void* out, int B, int C, int H, int W, long long out_bs, float slope,
             const float* ws1, const float* ws2, int nseg, hipStream_t stream, hipEvent_t ev0, hipEvent_t ev1, int fpitch = 0) {
  if (fpitch == 0) fpitch = W;
  if ((size_t)C * H * fpitch * 2 >= (1ull << 31)) return 1;                  // buffer-descriptor range
  // (the NCHW output is written in 8-pixel segments by the aligned form: W % 8 == 0; a pitched input with a ragged W takes the ragged form)
  const bool ragged = !((W % 8 == 0) && (fpitch % 8 == 0) && (out_bs % 8 == 0) && aligned_to(f1, 16) && aligned_to(f2, 16) && aligned_to(out, 16));
  if (ragged && W < 4) return 1;                                             // (rows shorter than a staging quad)
  const int v = allc_pick(B, C, H, W, ragged, NORM, !ragged && fpitch == W);
  if (v == -2 && fpitch != W) return 1;
  if (v < 0) return 1;
  if (ragged) return launch_allc<T, true, NORM, TO>(v, (const T*)f1, (const T*)f2, (TO*)out, B, C, H, W, out_bs, slope, ws1, ws2, nseg, stream, ev0, ev1, fpitch);
  return launch_allc<T, false, NORM, TO>(v, (const T*)f1, (const T*)f2, (TO*)out, B, C, H, W, out_bs, slope, ws1, ws2, nseg, stream, ev0, ev1, fpitch);
}

template <typename T>
int launch_fwd(const void* f1, const void* f2, void* out, int B, int C, int H, int W,
               long long out_bs, float slope, hipStream_t stream, hipEvent_t ev0 = nullptr, hipEvent_t ev1 = nullptr) {
  const int tiles_x = cdiv(W, TW), tiles_y = cdiv(H, TH);
  const long long nblocks = (long long)B * tiles_x * tiles_y;
  UPF_REQUIRE(nblocks < (1ll << 31), UPF_EINVAL, "corr81_forward: grid too large");
  const size_t va = 4 * sizeof(typename Elem<T>::store_t);   // bytes of a 4-pixel vector
  const bool aligned = (W % 4 == 0) && (out_bs % 4 == 0) && aligned_to(f1, va) && aligned_to(f2, va) && aligned_to(out, va) &&
                       (size_t)C * H * W * sizeof(typename Elem<T>::store_t) < (1ull << 31);   // buffer-descriptor range
  if constexpr (sizeof(typename Elem<T>::store_t) == 2) {
    if (!g_old_path && getenv("UPF_CORR_NO_MFMA") == nullptr) {
      const int rc = try_allc<T, false>(f1, f2, out, B, C, H, W, out_bs, slope, nullptr, nullptr, 0, stream, ev0, ev1);
      if (rc != 1) return rc;
    }
    const bool mfma_ok = aligned && (W % 8 == 0) && (out_bs % 8 == 0) && aligned_to(f1, 16) && aligned_to(f2, 16) && aligned_to(out, 16) &&
                         getenv("UPF_CORR_NO_MFMA") == nullptr;
    if (mfma_ok) {
      if (C <= 4 * corrm::KQ)
        launch_mfma<T, true>((unsigned)nblocks, stream, ev0, ev1, (const T*)f1, (const T*)f2, (T*)out, C, H, W, tiles_x, tiles_y, out_bs, slope);
      else
        launch_mfma<T, false>((unsigned)nblocks, stream, ev0, ev1, (const T*)f1, (const T*)f2, (T*)out, C, H, W, tiles_x, tiles_y, out_bs, slope);
      return check_launch("corr81_forward");
    }
  }
  // chunk depth: deeper chunks mean fewer workgroup barriers, shallower ones a shorter exposed prologue
  // and three workgroups per CU; measured on MI355X (tools/corr_ablate.hip) KC=%d wins at every level
  launch_kc<T, UPF_CORR_KC>(aligned, (unsigned)nblocks, stream, ev0, ev1, (const T*)f1, (const T*)f2, (T*)out, C, H, W, tiles_x,
                            tiles_y, out_bs, slope);
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
  // correlation_forward<T> reads padded rows y1 + tj*stride2 + j >= max_displacement - (md/s2)*s2 - kernel_rad (correlation_cuda_kernel.cu:62,
  // :87-91): negative = the reference reads what precedes its buffer (undefined) — nothing to be a drop-in for
  UPF_REQUIRE(max_displacement - (max_displacement / stride2) * stride2 - (kernel_size - 1) / 2 >= 0, UPF_EUNSUPPORTED,
              "correlation_forward: (k=%d, md=%d, s2=%d) — the reference reads outside its padded buffer for these parameters (undefined)",
              kernel_size, max_displacement, stride2);
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

// ---- normalisation fused into the cost volume's loader ----------------------------------------------------------
extern "C" int upf_corr81_norm_supported(int C, int dtype) {
  return (dtype == UPF_F16 || dtype == UPF_BF16) && C > 0 && upf::corr::allc_fits(upf::corr::NALLC - 1, (C + 3) / 4, true, true);
}

extern "C" long long upf_corr81_norm_workspace_bytes(int B, int C, int H, int W) {
  const long long part = ((long long)2 * B * C * upf::misc::stats2_nseg((long long)B * C, H * W) * 3 + 3) / 4 * 4;   // floats, rounded to 16 bytes
  return part * (long long)sizeof(float) + (long long)2 * B * C * (long long)sizeof(float2);
}

extern "C" int upf_corr81_norm_forward_mixed(const void* f1, const void* f2, int f_row_pitch, void* out, int B, int C, int H, int W, int dtype, int out_dtype,
                                             long long out_batch_stride, float leaky_slope, void* workspace, void* stream) {
  using namespace upf;
  UPF_REQUIRE(out_dtype == dtype || ((dtype == UPF_F16 || dtype == UPF_BF16) && (out_dtype == UPF_F16 || out_dtype == UPF_BF16)), UPF_EDTYPE,
              "corr81_norm_forward: out_dtype %d (bf16 / fp16)", out_dtype);
  UPF_REQUIRE(f1 && f2 && out && workspace, UPF_EINVAL, "corr81_norm_forward: null pointer");
  UPF_REQUIRE(B > 0 && C > 0 && H > 0 && W > 0, UPF_EINVAL, "corr81_norm_forward: bad shape B=%d C=%d H=%d W=%d", B, C, H, W);
  UPF_REQUIRE(upf_corr81_norm_supported(C, dtype), UPF_EUNSUPPORTED,
              "corr81_norm_forward: bf16 / fp16 with C <= 208 only (dtype %d, C %d): use upf_normalize_forward + upf_corr81_forward", dtype, C);
  const int fp = f_row_pitch ? f_row_pitch : W;
  UPF_REQUIRE(fp >= W, UPF_EINVAL, "corr81_norm_forward: row pitch %d < W = %d", fp, W);
  UPF_REQUIRE((size_t)C * H * fp * 2 < (1ull << 31), UPF_EUNSUPPORTED, "corr81_norm_forward: batch item >= 2 GiB");
  if (out_batch_stride == 0) out_batch_stride = (long long)corr::ND * H * W;
  UPF_REQUIRE(out_batch_stride >= (long long)corr::ND * H * W, UPF_EINVAL, "corr81_norm_forward: out_batch_stride %lld < 81*H*W", out_batch_stride);
  hipStream_t s = (hipStream_t)stream;
  float* ws = (float*)workspace;
  const long long N = (long long)B * C;
  float2* fin = reinterpret_cast<float2*>(ws + ((size_t)2 * N * misc::stats2_nseg(N, H * W) * 3 + 3) / 4 * 4);    // final (mean, 1/std) pairs behind the partials (16-byte aligned)
  const int nseg = misc::launch_stats2(f1, f2, ws, fin, N, H * W, dtype, s, W, fp);
  int rc = check_launch("corr81_norm_forward (statistics)");
  if (rc != UPF_OK) return rc;
  const float* ws1 = reinterpret_cast<const float*>(fin);            // what the cost volume reads: final pairs of f1's rows ...
  const float* ws2 = reinterpret_cast<const float*>(fin + N);        // ... and of f2's
  if (dtype == UPF_BF16 && out_dtype == UPF_F16) rc = corr::try_allc<bf16_t, true, f16_t>(f1, f2, out, B, C, H, W, out_batch_stride, leaky_slope, ws1, ws2, nseg, s, nullptr, nullptr, fp);
  else if (dtype == UPF_F16 && out_dtype == UPF_BF16) rc = corr::try_allc<f16_t, true, bf16_t>(f1, f2, out, B, C, H, W, out_batch_stride, leaky_slope, ws1, ws2, nseg, s, nullptr, nullptr, fp);
  else if (dtype == UPF_BF16) rc = corr::try_allc<bf16_t, true>(f1, f2, out, B, C, H, W, out_batch_stride, leaky_slope, ws1, ws2, nseg, s, nullptr, nullptr, fp);
  else rc = corr::try_allc<f16_t, true>(f1, f2, out, B, C, H, W, out_batch_stride, leaky_slope, ws1, ws2, nseg, s, nullptr, nullptr, fp);
  UPF_REQUIRE(rc != 1, UPF_EUNSUPPORTED, "corr81_norm_forward: no kernel variant fits C=%d W=%d (W >= 4 required)", C, W);
  return rc;
}

extern "C" int upf_corr81_norm_forward_pitched(const void* f1, const void* f2, int f_row_pitch, void* out, int B, int C, int H, int W, int dtype,
                                               long long out_batch_stride, float leaky_slope, void* workspace, void* stream) {
  return upf_corr81_norm_forward_mixed(f1, f2, f_row_pitch, out, B, C, H, W, dtype, dtype, out_batch_stride, leaky_slope, workspace, stream);
}

extern "C" int upf_corr81_norm_forward(const void* f1, const void* f2, void* out, int B, int C, int H, int W, int dtype,
                                       long long out_batch_stride, float leaky_slope, void* workspace, void* stream) {
  return upf_corr81_norm_forward_pitched(f1, f2, 0, out, B, C, H, W, dtype, out_batch_stride, leaky_slope, workspace, stream);
}

// octet output: the feature rows must be 16-byte aligned — W % 8 == 0, or (round 5) any W on rows PITCHED to a multiple of 8 elements
static int norm_c8_check(const void* f1, const void* f2, const void* out8, long long out8_batch_stride, int H, int W, int fp, const char* who) {
  using namespace upf;
  UPF_REQUIRE(fp >= W && fp % 8 == 0 && aligned_to(f1, 16) && aligned_to(f2, 16) && aligned_to(out8, 16) && out8_batch_stride % 8 == 0, UPF_EUNSUPPORTED,
              "%s: the feature rows must be 16-byte aligned (W %% 8 == 0, or a row pitch that is a multiple of 8) and the operands 16-byte aligned (W = %d, pitch = %d)", who, W, fp);
  UPF_REQUIRE(W >= 4, UPF_EUNSUPPORTED, "%s: W = %d < 4", who, W);
  UPF_REQUIRE(out8_batch_stride >= (long long)11 * H * W * 8, UPF_EINVAL, "%s: out8_batch_stride %lld < 11 octets", who, out8_batch_stride);
  return UPF_OK;
}

extern "C" int upf_corr81_norm_forward_c8_mixed(const void* f1, const void* f2, int f_row_pitch, void* out8, long long out8_batch_stride, int B, int C, int H, int W,
                                                int dtype, int out_dtype, float leaky_slope, void* workspace, void* stream) {
  using namespace upf;
  UPF_REQUIRE(out_dtype == dtype || ((dtype == UPF_F16 || dtype == UPF_BF16) && (out_dtype == UPF_F16 || out_dtype == UPF_BF16)), UPF_EDTYPE,
              "corr81_norm_forward_c8: out_dtype %d (bf16 / fp16)", out_dtype);
  UPF_REQUIRE(f1 && f2 && out8 && workspace, UPF_EINVAL, "corr81_norm_forward_c8: null pointer");
  UPF_REQUIRE(B > 0 && C > 0 && H > 0 && W > 0, UPF_EINVAL, "corr81_norm_forward_c8: bad shape B=%d C=%d H=%d W=%d", B, C, H, W);
  UPF_REQUIRE(upf_corr81_norm_supported(C, dtype), UPF_EUNSUPPORTED,
              "corr81_norm_forward_c8: bf16 / fp16 with C <= 208 only (dtype %d, C %d)", dtype, C);
  const int fp = f_row_pitch ? f_row_pitch : W;
  UPF_REQUIRE((size_t)C * H * fp * 2 < (1ull << 31), UPF_EUNSUPPORTED, "corr81_norm_forward_c8: batch item >= 2 GiB");
  int rc = norm_c8_check(f1, f2, out8, out8_batch_stride, H, W, fp, "corr81_norm_forward_c8");
  if (rc != UPF_OK) return rc;
  hipStream_t s = (hipStream_t)stream;
  float* ws = (float*)workspace;
  const long long N = (long long)B * C;
  float2* fin = reinterpret_cast<float2*>(ws + ((size_t)2 * N * misc::stats2_nseg(N, H * W) * 3 + 3) / 4 * 4);    // final (mean, 1/std) pairs behind the partials (16-byte aligned)
  const int nseg = misc::launch_stats2(f1, f2, ws, fin, N, H * W, dtype, s, W, fp);
  rc = check_launch("corr81_norm_forward_c8 (statistics)");
  if (rc != UPF_OK) return rc;
  const float* ws1 = reinterpret_cast<const float*>(fin);            // what the cost volume reads: final pairs of f1's rows ...
  const float* ws2 = reinterpret_cast<const float*>(fin + N);        // ... and of f2's
  const int v = corr::allc_pick(B, C, H, W, false, true, false);
  UPF_REQUIRE(v >= 0, UPF_EUNSUPPORTED, "corr81_norm_forward_c8: no kernel variant fits C=%d", C);
  const bool padw = (W % 8 != 0);
#define UPF_C8X(TI, TOUT) return padw ? corr::launch_allc_c8<TI, true, TOUT>(v, (const TI*)f1, (const TI*)f2, (TOUT*)out8, B, C, H, W, out8_batch_stride, leaky_slope, ws1, ws2, nseg, s, nullptr, nullptr, fp) \
                                     : corr::launch_allc_c8<TI, false, TOUT>(v, (const TI*)f1, (const TI*)f2, (TOUT*)out8, B, C, H, W, out8_batch_stride, leaky_slope, ws1, ws2, nseg, s, nullptr, nullptr, fp)
  if (dtype == UPF_F16 && out_dtype == UPF_BF16) { UPF_C8X(f16_t, bf16_t); }
  if (dtype == UPF_BF16 && out_dtype == UPF_F16) { UPF_C8X(bf16_t, f16_t); }
#undef UPF_C8X
  if (dtype == UPF_BF16)
    return padw ? corr::launch_allc_c8<bf16_t, true>(v, (const bf16_t*)f1, (const bf16_t*)f2, (bf16_t*)out8, B, C, H, W, out8_batch_stride, leaky_slope, ws1, ws2, nseg, s, nullptr, nullptr, fp)
                : corr::launch_allc_c8<bf16_t, false>(v, (const bf16_t*)f1, (const bf16_t*)f2, (bf16_t*)out8, B, C, H, W, out8_batch_stride, leaky_slope, ws1, ws2, nseg, s, nullptr, nullptr, fp);
  return padw ? corr::launch_allc_c8<f16_t, true>(v, (const f16_t*)f1, (const f16_t*)f2, (f16_t*)out8, B, C, H, W, out8_batch_stride, leaky_slope, ws1, ws2, nseg, s, nullptr, nullptr, fp)
              : corr::launch_allc_c8<f16_t, false>(v, (const f16_t*)f1, (const f16_t*)f2, (f16_t*)out8, B, C, H, W, out8_batch_stride, leaky_slope, ws1, ws2, nseg, s, nullptr, nullptr, fp);
}

extern "C" int upf_corr81_norm_forward_c8_pitched(const void* f1, const void* f2, int f_row_pitch, void* out8, long long out8_batch_stride, int B, int C, int H, int W,
                                                  int dtype, float leaky_slope, void* workspace, void* stream) {
  return upf_corr81_norm_forward_c8_mixed(f1, f2, f_row_pitch, out8, out8_batch_stride, B, C, H, W, dtype, dtype, leaky_slope, workspace, stream);
}

extern "C" int upf_corr81_norm_forward_c8(const void* f1, const void* f2, void* out8, long long out8_batch_stride, int B, int C, int H, int W,
                                          int dtype, float leaky_slope, void* workspace, void* stream) {
  return upf_corr81_norm_forward_c8_pitched(f1, f2, 0, out8, out8_batch_stride, B, C, H, W, dtype, leaky_slope, workspace, stream);
}

// The NORM cost volume — the variant inside the inference step — timed like upf_corr81_forward_timed: one (untimed) statistics
// launch, then nrep launches of the cost-volume kernel, each between its own pair of HIP events on the launch stream.
static int norm_forward_timed_impl(bool c8, const void* f1, const void* f2, void* out, int B, int C, int H, int W, int dtype,
                                   long long out_batch_stride, float leaky_slope, void* workspace, void* stream, int nrep,
                                   float* avg_us, float* min_us, int fp = 0, int out_dtype = -1) {
  using namespace upf;
  if (out_dtype < 0) out_dtype = dtype;
  UPF_REQUIRE(out_dtype == dtype || (c8 && dtype == UPF_F16 && out_dtype == UPF_BF16), UPF_EUNSUPPORTED,
              "corr81_norm_forward_timed: a second output type is timed for fp16 features -> bf16 octets only (the `pyramid_dtype` step)");
  UPF_REQUIRE(f1 && f2 && out && workspace && avg_us, UPF_EINVAL, "corr81_norm_forward_timed: null pointer");
  UPF_REQUIRE(B > 0 && C > 0 && H > 0 && W > 0 && nrep > 0 && nrep <= 1024, UPF_EINVAL, "corr81_norm_forward_timed: bad arguments");
  if (fp == 0) fp = W;
  UPF_REQUIRE(upf_corr81_norm_supported(C, dtype) && (size_t)C * H * fp * 2 < (1ull << 31), UPF_EUNSUPPORTED, "corr81_norm_forward_timed: unsupported shape / dtype");
  if (c8) {
    const int rc8 = norm_c8_check(f1, f2, out, out_batch_stride, H, W, fp, "corr81_norm_forward_c8_timed");
    if (rc8 != UPF_OK) return rc8;
  } else if (out_batch_stride == 0) out_batch_stride = (long long)corr::ND * H * W;
  const bool padw = c8 && (W % 8 != 0);
  const int vc8 = c8 ? corr::allc_pick(B, C, H, W, false, true, false) : 0;
  UPF_REQUIRE(vc8 >= 0, UPF_EUNSUPPORTED, "corr81_norm_forward_c8_timed: no kernel variant fits C=%d", C);
  hipStream_t s = (hipStream_t)stream;
  float* ws = (float*)workspace;
  const long long N = (long long)B * C;
  float2* fin = reinterpret_cast<float2*>(ws + ((size_t)2 * N * misc::stats2_nseg(N, H * W) * 3 + 3) / 4 * 4);    // final (mean, 1/std) pairs behind the partials (16-byte aligned)
  const int nseg = misc::launch_stats2(f1, f2, ws, fin, N, H * W, dtype, s, W, fp);
  int rc = check_launch("corr81_norm_forward_timed (statistics)");
  if (rc != UPF_OK) return rc;
  const float* ws1 = reinterpret_cast<const float*>(fin);            // what the cost volume reads: final pairs of f1's rows ...
  const float* ws2 = reinterpret_cast<const float*>(fin + N);        // ... and of f2's
  hipEvent_t* ev = new hipEvent_t[2 * nrep];
  for (int i = 0; i < 2 * nrep; ++i) (void)hipEventCreate(&ev[i]);
  for (int i = 0; i < nrep && rc == UPF_OK; ++i) {
    if (out_dtype != dtype) rc = padw ? corr::launch_allc_c8<f16_t, true, bf16_t>(vc8, (const f16_t*)f1, (const f16_t*)f2, (bf16_t*)out, B, C, H, W, out_batch_stride, leaky_slope, ws1, ws2, nseg, s, ev[2 * i], ev[2 * i + 1], fp)
                                      : corr::launch_allc_c8<f16_t, false, bf16_t>(vc8, (const f16_t*)f1, (const f16_t*)f2, (bf16_t*)out, B, C, H, W, out_batch_stride, leaky_slope, ws1, ws2, nseg, s, ev[2 * i], ev[2 * i + 1], fp);
    else if (padw && dtype == UPF_BF16) rc = corr::launch_allc_c8<bf16_t, true>(vc8, (const bf16_t*)f1, (const bf16_t*)f2, (bf16_t*)out, B, C, H, W, out_batch_stride, leaky_slope, ws1, ws2, nseg, s, ev[2 * i], ev[2 * i + 1], fp);
    else if (padw) rc = corr::launch_allc_c8<f16_t, true>(vc8, (const f16_t*)f1, (const f16_t*)f2, (f16_t*)out, B, C, H, W, out_batch_stride, leaky_slope, ws1, ws2, nseg, s, ev[2 * i], ev[2 * i + 1], fp);
    else if (c8 && dtype == UPF_BF16) rc = corr::launch_allc_c8<bf16_t>(vc8, (const bf16_t*)f1, (const bf16_t*)f2, (bf16_t*)out, B, C, H, W, out_batch_stride, leaky_slope, ws1, ws2, nseg, s, ev[2 * i], ev[2 * i + 1], fp);
    else if (c8) rc = corr::launch_allc_c8<f16_t>(vc8, (const f16_t*)f1, (const f16_t*)f2, (f16_t*)out, B, C, H, W, out_batch_stride, leaky_slope, ws1, ws2, nseg, s, ev[2 * i], ev[2 * i + 1], fp);
    else if (dtype == UPF_BF16) rc = corr::try_allc<bf16_t, true>(f1, f2, out, B, C, H, W, out_batch_stride, leaky_slope, ws1, ws2, nseg, s, ev[2 * i], ev[2 * i + 1], fp);
    else rc = corr::try_allc<f16_t, true>(f1, f2, out, B, C, H, W, out_batch_stride, leaky_slope, ws1, ws2, nseg, s, ev[2 * i], ev[2 * i + 1], fp);
    if (rc == 1) { set_error("corr81_norm_forward_timed: no kernel variant fits C=%d W=%d", C, W); rc = UPF_EUNSUPPORTED; }
  }
  hipError_t e = hipStreamSynchronize(s);
  if (rc == UPF_OK && e != hipSuccess) { set_error("corr81_norm_forward_timed: %s", hipGetErrorString(e)); rc = (int)e; }
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

extern "C" int upf_corr81_norm_forward_timed(const void* f1, const void* f2, void* out, int B, int C, int H, int W, int dtype,
                                             long long out_batch_stride, float leaky_slope, void* workspace, void* stream, int nrep,
                                             float* avg_us, float* min_us) {
  return norm_forward_timed_impl(false, f1, f2, out, B, C, H, W, dtype, out_batch_stride, leaky_slope, workspace, stream, nrep, avg_us, min_us);
}
extern "C" int upf_corr81_norm_forward_c8_timed(const void* f1, const void* f2, void* out8, long long out8_batch_stride, int B, int C, int H, int W,
                                                int dtype, float leaky_slope, void* workspace, void* stream, int nrep, float* avg_us, float* min_us) {
  return norm_forward_timed_impl(true, f1, f2, out8, B, C, H, W, dtype, out8_batch_stride, leaky_slope, workspace, stream, nrep, avg_us, min_us);
}

extern "C" int upf_corr81_norm_forward_c8_timed_pitched(const void* f1, const void* f2, int f_row_pitch, void* out8, long long out8_batch_stride, int B, int C, int H, int W,
                                                        int dtype, float leaky_slope, void* workspace, void* stream, int nrep, float* avg_us, float* min_us) {
  return norm_forward_timed_impl(true, f1, f2, out8, B, C, H, W, dtype, out8_batch_stride, leaky_slope, workspace, stream, nrep, avg_us, min_us, f_row_pitch);
}

extern "C" int upf_corr81_norm_forward_c8_timed_mixed(const void* f1, const void* f2, int f_row_pitch, void* out8, long long out8_batch_stride, int B, int C, int H, int W,
                                                      int dtype, int out_dtype, float leaky_slope, void* workspace, void* stream, int nrep, float* avg_us, float* min_us) {
  return norm_forward_timed_impl(true, f1, f2, out8, B, C, H, W, dtype, out8_batch_stride, leaky_slope, workspace, stream, nrep, avg_us, min_us, f_row_pitch, out_dtype);
}

extern "C" int upf_corr_set_option(const char* name, int value) {
  using namespace upf::corr;
  int* slot = nullptr;
  if (name && !strcmp(name, "variant")) slot = &g_variant;
  else if (name && !strcmp(name, "old_path")) slot = &g_old_path;
  if (!slot) return -1000;
  const int prev = *slot;
  *slot = value;
  return prev;
}
