// Interleaved-staging instantiations of conv_kernel (conv_kernel.hpp, template argument IL): two LDS buffers, the next chunk
// transposed and written between the MFMAs of the current one, one barrier per chunk.  Stride 1, 3x3, rows of W % 8 == 0
// pixels, dilation 1 / 2 / 4 / 8 / 16, any Cin > 16 / Cout; everything else keeps the two-phase loop of conv3x3.hip.
#include "conv_kernel.hpp"

namespace upf {
namespace conv {

template <typename T, int MTW, int RPW, int NOCTS, int D, int NPRE = 1>
int launch_one_il(const Args& a, int slabs) {
  constexpr int TH = (4 / MTW) * RPW;
  constexpr bool PH = (D >= 2);
  constexpr int DV = PH ? 1 : D;
  constexpr int rows = (TH - 1) + 2 * DV + 1;
  constexpr int XWP = xw(1, margin_of(D)) + xw(1, margin_of(D)) / 16;
  constexpr int GE = (NOCTS * rows * XWP * 16 < 4 * EPI_WAVE_BYTES) ? 4 * EPI_WAVE_BYTES / 16 : NOCTS * rows * XWP;    // = the kernel's IL_EB
  const int tiles_x = cdiv(a.W, TW), tiles_y = PH ? cdiv(a.H, D * TH) * D : cdiv(a.H, TH);
  const int tiles = a.B * tiles_x * tiles_y;
  const size_t lds = (size_t)2 * GE * 16;
  UPF_REQUIRE(lds <= 160 * 1024, UPF_EUNSUPPORTED, "conv_forward: two tile images do not fit LDS (dilation %d)", a.d);
  static LdsOptIn opt;
  auto kern = &conv_kernel<T, MTW, RPW, 1, NOCTS, D, false, false, 0, false, true, NPRE>;
  opt.ensure(reinterpret_cast<const void*>(kern), lds);
  hipLaunchKernelGGL(kern, dim3((unsigned)tiles, slabs), dim3(NTHREADS), lds, a.stream, (const T*)a.x, a.xbs,
                     (const T*)a.wp, a.bias, (T*)a.y, a.ybs, a.Cin, a.Cout, a.H, a.W, a.H, a.W, g_ablate, tiles_x, tiles_y, a.slope,
                     (const T*)g_dbg_buffer, 0ll, 0);
  return check_launch("conv_forward");
}

template <typename T>
int launch_il(const Args& a, int mtw, int slabs) {
  if (a.d == 1) {
    if (mtw == 4) return launch_one_il<T, 4, 8, 2, 1>(a, slabs);
    if (mtw == 2) return launch_one_il<T, 2, 4, 4, 1>(a, slabs);
    if ((long long)a.B * cdiv(a.W, TW) * cdiv(a.H, 16) >= g_rpw4_min) return g_il_npre == 2 ? launch_one_il<T, 1, 4, 2, 1, 2>(a, slabs) : launch_one_il<T, 1, 4, 2, 1>(a, slabs);     // (16-channel chunks: one staging task per thread, all prefetched)
    return g_il_npre == 2 ? launch_one_il<T, 1, 2, 4, 1, 2>(a, slabs) : launch_one_il<T, 1, 2, 4, 1>(a, slabs);
  }
  const int th = g_ph_fit ? ph_tile_rows(a.H, a.d, mtw) : 8;
#define UPF_IL_PH(MT, RP)                                                               \
  switch (a.d) {                                                                        \
    case 2: return launch_one_il<T, MT, RP, wide_nocts<MT, 2>(), 2>(a, slabs);          \
    case 4: return launch_one_il<T, MT, RP, wide_nocts<MT, 4>(), 4>(a, slabs);          \
    case 8: return launch_one_il<T, MT, RP, wide_nocts<MT, 8>(), 8>(a, slabs);          \
    default: return launch_one_il<T, MT, RP, wide_nocts<MT, 16>(), 16>(a, slabs);       \
  }
  if (mtw == 4) { if (th == 6) { UPF_IL_PH(4, 6) } if (th == 4) { UPF_IL_PH(4, 4) } UPF_IL_PH(4, 8) }
  if (mtw == 2) { if (th == 6) { UPF_IL_PH(2, 3) } if (th == 4) { UPF_IL_PH(2, 2) } UPF_IL_PH(2, 4) }
  if (th == 4) { UPF_IL_PH(1, 1) }
  UPF_IL_PH(1, 2)
#undef UPF_IL_PH
}

template int launch_il<bf16_t>(const Args&, int, int);
template int launch_il<f16_t>(const Args&, int, int);

}  // namespace conv
}  // namespace upf
