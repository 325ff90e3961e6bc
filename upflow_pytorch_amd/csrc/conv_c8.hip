// Convolution on the matrix cores with operands in the CHANNEL-OCTET layout ("C8": [n][c/8][y][x][8], the 8 channels of a
// pixel are one 16-byte entry) — gfx950.  The kernel is conv_kernel of conv_kernel.hpp (XL / YC8 template arguments: LDS-DMA
// staging into two LDS buffers, 8-byte stores straight from the accumulators); this file holds its instantiations, the
// weight packing for a mixed K order and the C-ABI entry.
//
// Why a second layout: the dense estimator / context / SGU stacks of UPFlow
// (/root/reference/model/pwc_modules.py:250-286, :396-412, model/upflow.py:24-60) are chains of convolutions that read what
// convolutions wrote.  In NCHW every consumer transposes 8 channel rows x 8 pixels in registers on the way into LDS (8 loads,
// 32 v_perm, 8 LDS stores per 128 bytes, 32 staging registers per thread), and every producer transposes back through an LDS
// patch.  With the octet layout between them both passes vanish and the staging becomes asynchronous DMA.  The tensors that
// other operators produce or consume plane-wise (the cost volume, flows, pyramid features, the 2- / 3-channel outputs) stay
// NCHW: a layer's input is a C8 slice followed by an NCHW tail, its output either layout.
#include "conv_kernel.hpp"

namespace upf {
namespace conv {

// w [Cout, Cin, k, k] -> the packed MFMA-lane-order operand of conv3x3.hip, with the K axis (input channels) gathered through
// kmap: packed channel kk reads w[:, kmap[kk]] (kmap[kk] < 0 or kk >= K: zero) — the K order [C8 slice | NCHW tail] of a
// mixed-layout layer, each part padded to 32 channels.
template <typename T>
__global__ void pack_weights_kmap_kernel(const T* __restrict__ w, T* __restrict__ wp, int Cin, int Cout, int ntaps, const int* __restrict__ kmap, int K) {
  const int cop = pad32(Cout), nk = K / 16;
  const long long total = (long long)ntaps * cop * K;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int j = (int)(i & 7), px = (int)((i >> 3) & 31), kg = (int)((i >> 8) & 1);
    const long long b = i >> 9;                      // (slab * nk + kstep) * ntaps + tap
    const int tap = (int)(b % ntaps), kstep = (int)((b / ntaps) % nk), slab = (int)(b / ((long long)ntaps * nk));
    const int co = slab * 32 + px, kk = kstep * 16 + kg * 8 + j;
    const int ci = kmap[kk];
    T v; v.v = 0;
    if (ci >= 0 && ci < Cin && co < Cout) v = w[((size_t)co * Cin + ci) * ntaps + tap];
    wp[i] = v;
  }
}

// N16 operand order (conv_kernel<..., N16>): [16-channel output block][32-channel chunk][tap][lane = co % 16 + 16 * k-octet][8 k]
template <typename T>
__global__ void pack_weights_kmap16_kernel(const T* __restrict__ w, T* __restrict__ wp, int Cin, int Cout, int ntaps, const int* __restrict__ kmap, int K) {
  const int nblk = (Cout + 15) / 16, nch = K / 32;
  const long long total = (long long)nblk * nch * ntaps * 512;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int j = (int)(i & 7), lane = (int)((i >> 3) & 63);
    const long long b = i >> 9;                      // (block * nch + chunk) * ntaps + tap
    const int tap = (int)(b % ntaps), chunk = (int)((b / ntaps) % nch), blk = (int)(b / ((long long)ntaps * nch));
    const int co = blk * 16 + (lane & 15), kk = chunk * 32 + (lane >> 4) * 8 + j;
    const int ci = kmap[kk];
    T v; v.v = 0;
    if (ci >= 0 && ci < Cin && co < Cout) v = w[((size_t)co * Cin + ci) * ntaps + tap];
    wp[i] = v;
  }
}

struct ArgsC8 {
  const void* x8; long long x8bs; int n8oct;       // C8 slice of the input (XL >= 1)
  const void* x2; long long x2bs; int C2;          // NCHW part of the input (XL == 0 or 2)
  const void* wp; const float* bias; void* y; long long ybs;
  int B, Cout, H, W, d, stride, ntaps; float slope; hipStream_t stream;
  int x2pitch, ypitch;                               // row pitch of the NCHW input part / of an NCHW output (elements; conv_kernel.hpp)
};

template <typename T, int MTW, int RPW, int S, int NOCTS, int D, int XL, bool YC8, bool N16 = false>
int launch_one_c8(const ArgsC8& a, int slabs) {
  constexpr int TH = (4 / MTW) * RPW;
  constexpr bool PH = (D >= 2 && S == 1);
  constexpr int DV = PH ? 1 : D;
  constexpr int rowsC = S * (TH - 1) + 2 * DV + 1;
  constexpr int XWP = xwp_eff(XL, S, D);
  constexpr int EB = NOCTS * rowsC * XWP, EBP = (EB + 63) & ~63;
  const int Ho = (a.H - 1) / S + 1, Wo = (a.W - 1) / S + 1;
  const int tiles_x = cdiv(Wo, TW), tiles_y = PH ? cdiv(Ho, D * TH) * D : cdiv(Ho, TH);
  size_t lds = (XL >= 1) ? (size_t)2 * EBP * 16 : (size_t)EB * 16;
  if (!YC8 && !N16 && lds < 4 * EPI_WAVE_BYTES) lds = 4 * EPI_WAVE_BYTES;
  UPF_REQUIRE(lds <= 160 * 1024, UPF_EUNSUPPORTED, "conv_forward_c8: tile does not fit LDS (dilation %d)", a.d);
  static LdsOptIn opt;
  auto kern = &conv_kernel<T, MTW, RPW, S, NOCTS, D, false, false, XL, YC8, N16>;
  opt.ensure(reinterpret_cast<const void*>(kern), lds);
  hipLaunchKernelGGL(kern, dim3((unsigned)(a.B * tiles_x * tiles_y), slabs), dim3(NTHREADS), lds, a.stream, (const T*)a.x2, a.x2bs,
                     (const T*)a.wp, a.bias, (T*)a.y, a.ybs, a.C2, a.Cout, a.H, a.W, Ho, Wo, g_ablate, tiles_x, tiles_y, a.slope,
                     (const T*)a.x8, a.x8bs, a.n8oct, a.x2pitch, a.ypitch);
  return check_launch("conv_forward_c8");
}

inline int g_c8_rpw4 = 1;      // Cout <= 32 layers on large grids: 16-row tiles (1) or 8-row tiles (0)
inline int g_c8_narrow_tall = 0;

// (MTW, tile rows) by Cout and grid like launch() of conv3x3.hip; stride 1; the instantiated subset:
//   dilation 1 (and 1x1 with an NCHW input): any input layout, both output layouts (NCHW in -> NCHW out is conv3x3.hip's);
//   dilation 2 / 4 / 8 / 16: C8 in, C8 out
template <typename T, int XL, bool YC8>
int launch_c8(const ArgsC8& a) {
  const int mt = cdiv(a.Cout, 32);
  const long long tiles = (long long)a.B * cdiv(a.W, TW) * cdiv(a.H, 8);
  int mtw = mt >= 3 ? 4 : mt;
  if (tiles <= g_small_grid && mt > 1) mtw = 2;
  if (g_force_mtw == 1 || g_force_mtw == 2 || g_force_mtw == 4) mtw = g_force_mtw < (mt >= 3 ? 4 : mt) ? g_force_mtw : (mt >= 3 ? 4 : mt);
  const int slabs = cdiv(mt, mtw);
  if (a.ntaps == 1) {
    if constexpr (XL == 0 && YC8) {
      if (mtw == 1) return launch_one_c8<T, 1, 2, 1, 4, 0, 0, true>(a, slabs);
    }
    set_error("conv_forward_c8: 1x1 kernels take an NCHW input, a C8 output and Cout <= 32");
    return UPF_EUNSUPPORTED;
  }
  if (a.stride == 2) {                               // NCHW in, C8 out, stride 2: the last layer of the SGU guidance stem
    if constexpr (XL == 0 && YC8) {
      if (a.d == 1 && a.Cout <= 32 && a.C2 > 16) return launch_one_c8<T, 1, 2, 2, 4, 1, 0, true>(a, 1);
    }
    set_error("conv_forward_c8: stride 2 takes an NCHW input of more than 16 channels, a C8 output of at most 32 channels, dilation 1");
    return UPF_EUNSUPPORTED;
  }
  if (a.d == 1) {
    if constexpr (XL == 0 && YC8) {                  // NCHW in, C8 out: the first layer of a C8 chain (context network conv0)
      if (mtw == 4) return launch_one_c8<T, 4, 8, 1, 2, 1, 0, true>(a, slabs);
      if (mtw == 2) return launch_one_c8<T, 2, 4, 1, 4, 1, 0, true>(a, slabs);
      return launch_one_c8<T, 1, 2, 1, 4, 1, 0, true>(a, slabs);
    }
    if constexpr (XL >= 1) {
      if (mtw == 4) return launch_one_c8<T, 4, 8, 1, 2, 1, XL, YC8>(a, slabs);
      if (mtw == 2) return launch_one_c8<T, 2, 4, 1, 4, 1, XL, YC8>(a, slabs);
      if (g_c8_rpw4 && (long long)a.B * cdiv(a.W, TW) * cdiv(a.H, 16) >= g_rpw4_min) return launch_one_c8<T, 1, 4, 1, 2, 1, XL, YC8>(a, slabs);
      return launch_one_c8<T, 1, 2, 1, 4, 1, XL, YC8>(a, slabs);
    }
  }
  if constexpr (XL == 1 && YC8) {
    if (a.d == 2 || a.d == 4 || a.d == 8 || a.d == 16) {
      const int th = g_ph_fit ? ph_tile_rows(a.H, a.d, mtw) : 8;
#define UPF_C8_PH(MT, RP)                                                              \
      switch (a.d) {                                                                   \
        case 2: return launch_one_c8<T, MT, RP, 1, wide_nocts<MT, 2>(), 2, 1, true>(a, slabs);   \
        case 4: return launch_one_c8<T, MT, RP, 1, wide_nocts<MT, 4>(), 4, 1, true>(a, slabs);   \
        case 8: return launch_one_c8<T, MT, RP, 1, wide_nocts<MT, 8>(), 8, 1, true>(a, slabs);   \
        default: return launch_one_c8<T, MT, RP, 1, wide_nocts<MT, 16>(), 16, 1, true>(a, slabs); \
      }
      if (mtw == 4) { if (th == 6) { UPF_C8_PH(4, 6) } if (th == 4) { UPF_C8_PH(4, 4) } UPF_C8_PH(4, 8) }
      if (mtw == 2) { if (th == 6) { UPF_C8_PH(2, 3) } if (th == 4) { UPF_C8_PH(2, 2) } UPF_C8_PH(2, 4) }
      if (th == 4) { UPF_C8_PH(1, 1) }
      UPF_C8_PH(1, 2)
#undef UPF_C8_PH
    }
  }
  set_error("conv_forward_c8: no kernel for this combination (kernel %d taps, dilation %d, input layout %d, output C8 %d)", a.ntaps, a.d, XL, (int)YC8);
  return UPF_EUNSUPPORTED;
}

template <typename T>
int dispatch_c8(const ArgsC8& a, bool y_c8) {
  const int xl = a.x8 ? (a.x2 ? 2 : 1) : 0;
  if (xl == 2) return y_c8 ? launch_c8<T, 2, true>(a) : launch_c8<T, 2, false>(a);
  if (xl == 1) return y_c8 ? launch_c8<T, 1, true>(a) : launch_c8<T, 1, false>(a);
  if (y_c8) return launch_c8<T, 0, true>(a);
  set_error("conv_forward_c8: NCHW in, NCHW out is upf_conv_forward");
  return UPF_EINVAL;
}

}  // namespace conv
}  // namespace upf

extern "C" long long upf_conv_packed_bytes_k(int K, int Cout, int kernel_size) {
  return (long long)kernel_size * kernel_size * upf::conv::pad32(Cout) * (long long)K * 2;
}

extern "C" int upf_conv_c8_k(int n8_oct, int C2) {
  return (n8_oct > 0 ? upf::conv::pad32(n8_oct * 8) : 0) + (C2 > 0 ? upf::conv::pad32(C2) : 0);
}

extern "C" int upf_conv_pack_weights_kmap(const void* w, void* w_packed, int Cin, int Cout, int kernel_size, const int* kmap, int K,
                                          int dtype, void* stream) {
  using namespace upf;
  UPF_REQUIRE(w && w_packed && kmap && Cin > 0 && Cout > 0, UPF_EINVAL, "conv_pack_weights_kmap: bad arguments");
  UPF_REQUIRE(kernel_size == 3 || kernel_size == 1, UPF_EUNSUPPORTED, "conv_pack_weights_kmap: kernel_size %d (1 or 3)", kernel_size);
  UPF_REQUIRE(K > 0 && K % 32 == 0, UPF_EINVAL, "conv_pack_weights_kmap: K = %d must be a positive multiple of 32", K);
  UPF_REQUIRE(dtype == UPF_BF16 || dtype == UPF_F16, UPF_EDTYPE, "conv: bf16 / fp16 only");
  const int ntaps = kernel_size * kernel_size;
  const long long total = (long long)ntaps * conv::pad32(Cout) * K;
  const int blocks = (int)((total + 255) / 256 > 4096 ? 4096 : (total + 255) / 256);
  if (dtype == UPF_BF16)
    hipLaunchKernelGGL((conv::pack_weights_kmap_kernel<bf16_t>), dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)w, (bf16_t*)w_packed, Cin, Cout, ntaps, kmap, K);
  else
    hipLaunchKernelGGL((conv::pack_weights_kmap_kernel<f16_t>), dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const f16_t*)w, (f16_t*)w_packed, Cin, Cout, ntaps, kmap, K);
  return check_launch("conv_pack_weights_kmap");
}

extern "C" int upf_conv_c8_set_option(const char* name, int value) {
  using namespace upf::conv;
  int* slot = nullptr;
  if (name && !strcmp(name, "rpw4")) slot = &g_c8_rpw4;
  else if (name && !strcmp(name, "narrow_tall")) slot = &g_c8_narrow_tall;
  if (!slot) return INT32_MIN;
  const int prev = *slot;
  *slot = value;
  return prev;
}

extern "C" int upf_conv_forward_c8_pitched(const void* x8, long long x8_batch_stride, int n8_oct, const void* x2, long long x2_batch_stride, int x2_row_pitch, int C2,
                                           const void* w_packed, const float* bias, void* y, long long y_batch_stride, int y_row_pitch, int y_is_c8,
                                           int B, int Cout, int H, int W, int kernel_size, int dilation, int stride, float leaky_slope,
                                           int dtype, void* stream) {
  using namespace upf;
  UPF_REQUIRE((x8 || x2) && w_packed && bias && y, UPF_EINVAL, "conv_forward_c8: null pointer");
  UPF_REQUIRE((x8 != nullptr) == (n8_oct > 0) && (x2 != nullptr) == (C2 > 0), UPF_EINVAL, "conv_forward_c8: a pointer and its channel count disagree");
  UPF_REQUIRE(B > 0 && Cout > 0 && H > 0 && W > 0, UPF_EINVAL, "conv_forward_c8: bad shape B=%d Cout=%d H=%d W=%d", B, Cout, H, W);
  UPF_REQUIRE(dtype == UPF_BF16 || dtype == UPF_F16, UPF_EDTYPE, "conv_forward_c8: bf16 / fp16 only");
  UPF_REQUIRE(kernel_size == 3 || kernel_size == 1, UPF_EUNSUPPORTED, "conv_forward_c8: kernel_size %d (1 or 3)", kernel_size);
  UPF_REQUIRE(stride == 1 || (stride == 2 && kernel_size == 3 && dilation == 1 && !x8), UPF_EUNSUPPORTED, "conv_forward_c8: stride %d (1, or 2 for a 3x3 with an NCHW input)", stride);
  UPF_REQUIRE(dilation >= 1 && dilation <= conv::MAXD, UPF_EUNSUPPORTED, "conv_forward_c8: dilation %d not in [1,%d]", dilation, conv::MAXD);
  const int Ho = (H - 1) / stride + 1, Wo = (W - 1) / stride + 1;
  if (x2_row_pitch == 0) x2_row_pitch = W;
  if (y_row_pitch == 0) y_row_pitch = Wo;
  // C8 operands: a pixel is a 16-byte entry, rows are aligned for every W.  The NCHW input part is staged with aligned 16-byte
  // loads: its row PITCH must be a multiple of 8 elements (the logical W may be ragged; round 5).
  UPF_REQUIRE(!x2 || (x2_row_pitch >= W && x2_row_pitch % 8 == 0), UPF_EUNSUPPORTED,
              "conv_forward_c8: the NCHW input's row pitch %d must be a multiple of 8 and >= W = %d (pitched rows, or upf_conv_forward)", x2_row_pitch, W);
  UPF_REQUIRE(y_is_c8 || y_row_pitch >= Wo, UPF_EINVAL, "conv_forward_c8: output row pitch %d < %d", y_row_pitch, Wo);
  UPF_REQUIRE(!x8 || (aligned_to(x8, 16) && x8_batch_stride % 8 == 0), UPF_EINVAL, "conv_forward_c8: the C8 input must be 16-byte aligned");
  UPF_REQUIRE(!x2 || (aligned_to(x2, 16) && x2_batch_stride % 8 == 0), UPF_EINVAL, "conv_forward_c8: the NCHW input must be 16-byte aligned");
  UPF_REQUIRE(!y_is_c8 || (aligned_to(y, 16) && y_batch_stride % 8 == 0), UPF_EINVAL, "conv_forward_c8: the C8 output must be 16-byte aligned");
  UPF_REQUIRE((long long)n8_oct * H * W * 16 < (1ll << 31) && (long long)C2 * H * x2_row_pitch * 2 < (1ll << 31), UPF_EINVAL, "conv_forward_c8: image too large for one buffer descriptor");
  UPF_REQUIRE((long long)((Cout + 7) / 8) * (H / stride + 1) * (W / stride + 1) * 16 < (1ll << 31) && (y_is_c8 || (long long)Cout * Ho * y_row_pitch * 2 < (1ll << 31)), UPF_EINVAL,
              "conv_forward_c8: output too large for one buffer descriptor");
  UPF_REQUIRE(leaky_slope >= 0.f && leaky_slope <= 1.f, UPF_EINVAL, "conv_forward_c8: leaky_slope %g not in [0,1]", (double)leaky_slope);
  conv::ArgsC8 a{x8, x8_batch_stride, n8_oct, x2, x2_batch_stride, C2, w_packed, bias, y, y_batch_stride,
                 B, Cout, H, W, kernel_size == 1 ? 0 : dilation, stride, kernel_size * kernel_size, leaky_slope == 0.f ? 1.f : leaky_slope, (hipStream_t)stream,
                 x2_row_pitch, y_row_pitch};
  return dtype == UPF_BF16 ? conv::dispatch_c8<bf16_t>(a, y_is_c8 != 0) : conv::dispatch_c8<f16_t>(a, y_is_c8 != 0);
}

extern "C" int upf_conv_forward_c8(const void* x8, long long x8_batch_stride, int n8_oct, const void* x2, long long x2_batch_stride, int C2,
                                   const void* w_packed, const float* bias, void* y, long long y_batch_stride, int y_is_c8,
                                   int B, int Cout, int H, int W, int kernel_size, int dilation, int stride, float leaky_slope,
                                   int dtype, void* stream) {
  return upf_conv_forward_c8_pitched(x8, x8_batch_stride, n8_oct, x2, x2_batch_stride, 0, C2, w_packed, bias, y, y_batch_stride, 0, y_is_c8,
                                     B, Cout, H, W, kernel_size, dilation, stride, leaky_slope, dtype, stream);
}

// ---- layers with at most 16 output channels on the 16-channel matrix instruction (conv_kernel<..., N16>) -------------------------
extern "C" long long upf_conv_packed_bytes_k16(int K, int Cout) {
  return (long long)((Cout + 15) / 16) * (upf::conv::pad32(K) / 32) * 9 * 1024;
}

extern "C" int upf_conv_pack_weights_kmap16(const void* w, void* w_packed, int Cin, int Cout, const int* kmap, int K, int dtype, void* stream) {
  using namespace upf;
  UPF_REQUIRE(w && w_packed && kmap && Cin > 0 && Cout > 0 && K > 0 && K % 32 == 0, UPF_EINVAL, "conv_pack_weights_kmap16: bad arguments (K %% 32 == 0)");
  UPF_REQUIRE(dtype == UPF_BF16 || dtype == UPF_F16, UPF_EDTYPE, "conv_pack_weights_kmap16: bf16 / fp16 only");
  const long long total = (long long)((Cout + 15) / 16) * (K / 32) * 9 * 512;
  const int blocks = (int)((total + 255) / 256 > 4096 ? 4096 : (total + 255) / 256);
  if (dtype == UPF_BF16)
    hipLaunchKernelGGL((conv::pack_weights_kmap16_kernel<bf16_t>), dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)w, (bf16_t*)w_packed, Cin, Cout, 9, kmap, K);
  else
    hipLaunchKernelGGL((conv::pack_weights_kmap16_kernel<f16_t>), dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const f16_t*)w, (f16_t*)w_packed, Cin, Cout, 9, kmap, K);
  return check_launch("conv_pack_weights_kmap16");
}

extern "C" int upf_conv_forward_c8_narrow(const void* x8, long long x8_batch_stride, int n8_oct, const void* w_packed16, const float* bias,
                                          void* y, long long y_batch_stride, int y_is_c8, int B, int Cout, int H, int W, float leaky_slope,
                                          int dtype, void* stream) {
  using namespace upf;
  UPF_REQUIRE(x8 && n8_oct > 0 && w_packed16 && bias && y, UPF_EINVAL, "conv_forward_c8_narrow: null pointer");
  UPF_REQUIRE(B > 0 && Cout > 0 && H > 0 && W > 0, UPF_EINVAL, "conv_forward_c8_narrow: bad shape B=%d Cout=%d H=%d W=%d", B, Cout, H, W);
  UPF_REQUIRE(dtype == UPF_BF16 || dtype == UPF_F16, UPF_EDTYPE, "conv_forward_c8_narrow: bf16 / fp16 only");
  UPF_REQUIRE(aligned_to(x8, 16) && x8_batch_stride % 8 == 0, UPF_EINVAL, "conv_forward_c8_narrow: the C8 input must be 16-byte aligned");
  UPF_REQUIRE(!y_is_c8 || (aligned_to(y, 16) && y_batch_stride % 8 == 0), UPF_EINVAL, "conv_forward_c8_narrow: the C8 output must be 16-byte aligned");
  UPF_REQUIRE((long long)n8_oct * H * W * 16 < (1ll << 31), UPF_EINVAL, "conv_forward_c8_narrow: input too large for one buffer descriptor");
  UPF_REQUIRE((long long)(y_is_c8 ? (Cout + 7) / 8 * 8 : Cout) * H * W * 2 < (1ll << 31), UPF_EINVAL, "conv_forward_c8_narrow: output too large for one buffer descriptor");
  UPF_REQUIRE(leaky_slope >= 0.f && leaky_slope <= 1.f, UPF_EINVAL, "conv_forward_c8_narrow: leaky_slope %g not in [0,1]", (double)leaky_slope);
  conv::ArgsC8 a{x8, x8_batch_stride, n8_oct, nullptr, 0, 0, w_packed16, bias, y, y_batch_stride, B, Cout, H, W, 1, 1, 9,
                 leaky_slope == 0.f ? 1.f : leaky_slope, (hipStream_t)stream, W, W};
  const int blocks16 = (Cout + 15) / 16;
  // 8-row tiles: with 32-channel chunks a 16-row tile's two LDS buffers (87 KB) leave ONE workgroup per CU (measured: no faster
  // than the 32-channel kernel); `narrow_tall` = 1 selects them for experiments
  const bool tall = conv::g_c8_narrow_tall && (long long)B * cdiv(W, conv::TW) * cdiv(H, 16) >= conv::g_rpw4_min;
#define UPF_N16(T)                                                                                                   \
  if (y_is_c8) return tall ? conv::launch_one_c8<T, 1, 4, 1, 4, 1, 1, true, true>(a, blocks16) : conv::launch_one_c8<T, 1, 2, 1, 4, 1, 1, true, true>(a, blocks16);   \
  return tall ? conv::launch_one_c8<T, 1, 4, 1, 4, 1, 1, false, true>(a, blocks16) : conv::launch_one_c8<T, 1, 2, 1, 4, 1, 1, false, true>(a, blocks16);
  if (dtype == UPF_BF16) { UPF_N16(bf16_t) }
  UPF_N16(f16_t)
#undef UPF_N16
}

// ---- merged narrow tail of a dense stack (conv_kernel.hpp: SplitOut / AccInit; round 6) -----------------------------------------
namespace upf {
namespace conv {

template <typename T, int MTW, int RPW, int NOCTS>
int launch_split(const ArgsC8& a, const SplitOut& so) {
  constexpr int TH = (4 / MTW) * RPW;
  constexpr int rowsC = (TH - 1) + 2 + 1;
  constexpr int XWP = xwp_eff(1, 1, 1);
  constexpr int EB = NOCTS * rowsC * XWP, EBP = (EB + 63) & ~63;
  const int tiles_x = cdiv(a.W, TW), tiles_y = cdiv(a.H, TH);
  const size_t lds = (size_t)2 * EBP * 16;
  static LdsOptIn opt;
  auto kern = &conv_split_kernel<T, MTW, RPW, NOCTS>;
  opt.ensure(reinterpret_cast<const void*>(kern), lds);
  hipLaunchKernelGGL(kern, dim3((unsigned)(a.B * tiles_x * tiles_y), (unsigned)cdiv(cdiv(a.Cout, 32), MTW)), dim3(NTHREADS), lds, a.stream, (const T*)a.wp, a.bias, (T*)a.y, a.ybs,
                     a.Cout, a.H, a.W, tiles_x, tiles_y, a.slope, (const T*)a.x8, a.x8bs, a.n8oct, so);
  return check_launch("conv_forward_c8_split");
}

// the launch shapes launch_c8 gives the same layer without the partial rows
template <typename T>
int dispatch_split(const ArgsC8& a, const SplitOut& so) {
  const int mt = cdiv(a.Cout, 32);
  const long long tiles = (long long)a.B * cdiv(a.W, TW) * cdiv(a.H, 8);
  int mtw = mt >= 3 ? 4 : mt;
  if (tiles <= g_small_grid && mt > 1) mtw = 2;
  if (mtw == 4) return launch_split<T, 4, 8, 2>(a, so);
  if (mtw == 2) return launch_split<T, 2, 4, 4>(a, so);
  if (g_c8_rpw4 && (long long)a.B * cdiv(a.W, TW) * cdiv(a.H, 16) >= g_rpw4_min) return launch_split<T, 1, 4, 2>(a, so);
  return launch_split<T, 1, 2, 4>(a, so);
}

template <typename T, int RPW, bool YC8>
int launch_accinit(const ArgsC8& a, const AccInit& ai) {
  constexpr int TH = 4 * RPW;
  constexpr int rowsC = (TH - 1) + 2 + 1;
  constexpr int XWP = xwp_eff(1, 1, 1);
  constexpr int EB = 4 * rowsC * XWP, EBP = (EB + 63) & ~63;
  const int tiles_x = cdiv(a.W, TW), tiles_y = cdiv(a.H, TH);
  const size_t lds = (size_t)2 * EBP * 16;
  static LdsOptIn opt;
  auto kern = &conv_accinit_kernel<T, RPW, YC8>;
  opt.ensure(reinterpret_cast<const void*>(kern), lds);
  hipLaunchKernelGGL(kern, dim3((unsigned)(a.B * tiles_x * tiles_y), (unsigned)cdiv(a.Cout, 16)), dim3(NTHREADS), lds, a.stream, (const T*)a.wp, (T*)a.y, a.ybs,
                     a.Cout, a.H, a.W, tiles_x, tiles_y, a.slope, (const T*)a.x8, a.x8bs, a.n8oct, ai);
  return check_launch("conv_forward_c8_narrow_init");
}

}  // namespace conv
}  // namespace upf

extern "C" int upf_conv_forward_c8_split(const void* x8, long long x8_batch_stride, int n8_oct, const void* w_packed, const float* bias,
                                         void* y8, long long y_batch_stride, int C_main, float* partial, long long partial_batch_stride, int partial_pitch,
                                         int B, int Cout, int H, int W, float leaky_slope, int dtype, void* stream) {
  using namespace upf;
  UPF_REQUIRE(x8 && n8_oct > 0 && w_packed && bias && y8 && partial, UPF_EINVAL, "conv_forward_c8_split: null pointer");
  UPF_REQUIRE(B > 0 && H > 0 && W > 0, UPF_EINVAL, "conv_forward_c8_split: bad shape B=%d H=%d W=%d", B, H, W);
  UPF_REQUIRE(C_main > 0 && C_main % 8 == 0 && Cout > C_main && Cout <= 128, UPF_EINVAL,
              "conv_forward_c8_split: C_main = %d must be a positive multiple of 8 below Cout = %d <= 128", C_main, Cout);
  UPF_REQUIRE(partial_pitch % 4 == 0 && partial_pitch >= (Cout - C_main + 3) / 4 * 4, UPF_EINVAL,
              "conv_forward_c8_split: partial pitch %d must be a multiple of 4 floats holding the %d partial channels", partial_pitch, Cout - C_main);
  UPF_REQUIRE(dtype == UPF_BF16 || dtype == UPF_F16, UPF_EDTYPE, "conv_forward_c8_split: bf16 / fp16 only");
  UPF_REQUIRE(aligned_to(x8, 16) && x8_batch_stride % 8 == 0 && aligned_to(y8, 16) && y_batch_stride % 8 == 0 && aligned_to(partial, 16) && partial_batch_stride % 4 == 0,
              UPF_EINVAL, "conv_forward_c8_split: operands must be 16-byte aligned");
  UPF_REQUIRE((long long)n8_oct * H * W * 16 < (1ll << 31) && (long long)H * W * partial_pitch * 4 < (1ll << 31), UPF_EINVAL,
              "conv_forward_c8_split: image too large for one buffer descriptor");
  UPF_REQUIRE(leaky_slope >= 0.f && leaky_slope <= 1.f, UPF_EINVAL, "conv_forward_c8_split: leaky_slope %g not in [0,1]", (double)leaky_slope);
  conv::ArgsC8 a{x8, x8_batch_stride, n8_oct, nullptr, 0, 0, w_packed, bias, y8, y_batch_stride, B, Cout, H, W, 1, 1, 9,
                 leaky_slope == 0.f ? 1.f : leaky_slope, (hipStream_t)stream, W, W};
  const conv::SplitOut so{partial, partial_batch_stride, partial_pitch, C_main};
  return dtype == UPF_BF16 ? conv::dispatch_split<bf16_t>(a, so) : conv::dispatch_split<f16_t>(a, so);
}

extern "C" int upf_conv_forward_c8_narrow_init(const void* x8, long long x8_batch_stride, int n8_oct, const void* w_packed16,
                                               const float* partial, long long partial_batch_stride, int partial_pitch, int partial_offset,
                                               void* y, long long y_batch_stride, int y_is_c8, int B, int Cout, int H, int W, float leaky_slope,
                                               int dtype, void* stream) {
  using namespace upf;
  UPF_REQUIRE(x8 && n8_oct > 0 && w_packed16 && partial && y, UPF_EINVAL, "conv_forward_c8_narrow_init: null pointer");
  UPF_REQUIRE(B > 0 && Cout > 0 && Cout <= 16 && H > 0 && W > 0, UPF_EINVAL, "conv_forward_c8_narrow_init: bad shape B=%d Cout=%d (<= 16) H=%d W=%d", B, Cout, H, W);
  UPF_REQUIRE(partial_pitch % 4 == 0 && partial_offset % 4 == 0 && partial_offset >= 0 && partial_offset + (Cout + 3) / 4 * 4 <= partial_pitch, UPF_EINVAL,
              "conv_forward_c8_narrow_init: partial pitch %d / offset %d must be multiples of 4 floats holding the layer's %d channels", partial_pitch, partial_offset, Cout);
  UPF_REQUIRE(dtype == UPF_BF16 || dtype == UPF_F16, UPF_EDTYPE, "conv_forward_c8_narrow_init: bf16 / fp16 only");
  UPF_REQUIRE(aligned_to(x8, 16) && x8_batch_stride % 8 == 0 && aligned_to(partial, 16) && partial_batch_stride % 4 == 0, UPF_EINVAL,
              "conv_forward_c8_narrow_init: the C8 input and the partial must be 16-byte aligned");
  UPF_REQUIRE(!y_is_c8 || (aligned_to(y, 16) && y_batch_stride % 8 == 0), UPF_EINVAL, "conv_forward_c8_narrow_init: the C8 output must be 16-byte aligned");
  UPF_REQUIRE((long long)n8_oct * H * W * 16 < (1ll << 31) && (long long)H * W * partial_pitch * 4 < (1ll << 31), UPF_EINVAL,
              "conv_forward_c8_narrow_init: image too large for one buffer descriptor");
  UPF_REQUIRE((long long)(y_is_c8 ? (Cout + 7) / 8 * 8 : Cout) * H * W * 2 < (1ll << 31), UPF_EINVAL, "conv_forward_c8_narrow_init: output too large for one buffer descriptor");
  UPF_REQUIRE(leaky_slope >= 0.f && leaky_slope <= 1.f, UPF_EINVAL, "conv_forward_c8_narrow_init: leaky_slope %g not in [0,1]", (double)leaky_slope);
  conv::ArgsC8 a{x8, x8_batch_stride, n8_oct, nullptr, 0, 0, w_packed16, nullptr, y, y_batch_stride, B, Cout, H, W, 1, 1, 9,
                 leaky_slope == 0.f ? 1.f : leaky_slope, (hipStream_t)stream, W, W};
  const conv::AccInit ai{partial, partial_batch_stride, partial_pitch, partial_offset};
#define UPF_N16I(T) return y_is_c8 ? conv::launch_accinit<T, 2, true>(a, ai) : conv::launch_accinit<T, 2, false>(a, ai);
  if (dtype == UPF_BF16) { UPF_N16I(bf16_t) }
  UPF_N16I(f16_t)
#undef UPF_N16I
}
