// Two convolutions as ONE launch: a 3x3 layer (Cin <= 16 -> C1 = 16 | 32 channels, LeakyReLU) followed by a 3x3 layer (C1 -> C2 <= 32 channels,
// LeakyReLU), strides (1, 2) — the two halves of the SGU guidance stem (/root/reference/model/upflow.py:30-33: conv(3, 16), conv(16, 16,
// stride=2), conv(16, 32), conv(32, 32, stride=2), run on both frames at full resolution) — or (2, 1) — the first two stages of the feature
// pyramid (model/pwc_modules.py:122-142: conv(3, 16, stride=2), conv(16, 16); conv(16, 32, stride=2), conv(32, 32)) — gfx950.
//
// Why (round 6): these layers are pure bandwidth — at 384x1280, 2B = 8 frames, the first layer WRITES 126 MB of 16-channel activations
// that the second reads straight back (52 + 36 us), the third writes 63 MB for the fourth (21 + 28 us): 7 % of a 2.8 ms step moving
// an intermediate that nothing else consumes.  Here a workgroup produces a TH x 32 tile of the stride-2 layer's output: it stages the
// input tile + halo once, computes the first layer on the (2 TH + 1) x 65 pixels the second layer reads — rounded to the 16-bit type
// and zero outside the image, exactly what the two-launch form would have read back — into LDS as channel octets, and multiplies the
// second layer from there.  HBM sees the input and the output only.  The first layer is recomputed on the one-pixel overlap of
// neighbouring tiles (6 % more of a layer that is 0.1 % of a step's flop).
//
// Both layers are v_mfma_f32_32x32x16 implicit GEMMs, D[co][pixel]:
//   layer A: B operand = one LDS entry (8 input channels of a pixel) per lane; with Cin <= 8 a k-step of 16 holds TWO TAPS (lane half kg
//            reads tap 2s + kg): 5 instructions per 32 pixels; with Cin <= 4 (the frames) entries are 4 channels and a k-step holds FOUR
//            taps (3 instructions); with Cin <= 16 a k-step is one tap (kg = channel octet): 9;
//   layer B: as conv_kernel's stride-2 form, entries (octet 2ks + kg, row 2r + ky, column 2px + kx) of the mid tile.
// Packed weights (host side, ops.conv_pair_pack): A: [step][lane = co + 32 kg][8 k], B: [tap][k-step][lane][8 k].
#include "conv_kernel.hpp"

namespace upf {
namespace convp {

using conv::f32x16;
using conv::Mma32;
using conv::u32x2;

constexpr int NT = 256;
// geometry of a TH x 32 output tile: layer B (stride SB) reads MR x MC mid pixels, layer A (stride SA) computes them from IR x IC
// input pixels; XO: the input columns start one pixel early where that makes the first column even (rows are loaded as pixel pairs)
template <int SA, int SB, int TH> struct Geo {
  static constexpr int MR = SB * (TH - 1) + 3, MC = SB * 31 + 3;
  static constexpr int XO = (SA == 2) ? 1 : 0;
  static constexpr int IR = SA * (MR - 1) + 3, IC = (SA * (MC - 1) + 3 + XO + 1) / 2 * 2;
};

template <typename T, int K0, int C1O, int TH, bool YC8, int SA, int SB>
__global__ __launch_bounds__(NT, 2)
void conv_pair_kernel(const T* __restrict__ x, long long xbs, int xpitch, int Cin, const T* __restrict__ wa, const float* __restrict__ ba, float slope_a,
                      const T* __restrict__ wb, const float* __restrict__ bb, float slope_b, T* __restrict__ y, long long ybs, int ypitch, int C2,
                      int H, int W, int Hm, int Wm, int Ho, int Wo, int tiles_x, int tiles_y) {
  using G = Geo<SA, SB, TH>;
  constexpr int MR = G::MR, MC = G::MC, IR = G::IR, IC = G::IC, XO = G::XO;
  constexpr bool QUAD = (K0 == 0);                   // Cin <= 4 (RGB): 8-byte entries of 4 channels, FOUR taps per k-step of 16
  constexpr int KO = QUAD ? 1 : K0;                  // staged channel planes (octets, or the one quad)
  constexpr int NSTEP = QUAD ? 3 : ((K0 == 1) ? 5 : 9);
  constexpr int KSB = C1O / 2;                       // k-steps of 16 channels of layer B
  constexpr int RPWB = TH / 4;                       // output rows per wave
  extern __shared__ __attribute__((aligned(16))) uint4 lds[];
  uint4* in_s = lds;                                 // [KO][IR][IC] entries: 8 (QUAD: 4) input channels of one pixel
  uint4* mid_s = lds + (QUAD ? (IR * IC + 1) / 2 : K0 * IR * IC);   // [C1O][MR][MC] entries: 8 channels of layer A's output

  const int bid = xcd_remap(blockIdx.x, gridDim.x);
  const int tx = bid % tiles_x, ty = (bid / tiles_x) % tiles_y, n = bid / (tiles_x * tiles_y);
  const int x0 = tx * 32, y0 = ty * TH;              // output coordinates of the tile
  const int my0 = SB * y0 - 1, mx0 = SB * x0 - 1;    // mid (layer A's output) coordinates of the mid tile's first pixel
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int px = lane & 31, kg = lane >> 5;

  // ---- phase 0: the input tile + halo, rows [SA my0 - 1, +IR), columns [SA mx0 - 1 - XO, +IC) as pixel PAIRS (4-byte loads: even pitch),
  // transposed to octet entries.  Planes >= Cin fall off the descriptor, pixels outside the image get the offset marker: zeros.
  {
    const uint32_t plane = (uint32_t)(H * xpitch) * 2u;
    __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(x + (size_t)n * xbs), 0, (uint32_t)Cin * plane, 0x00020000);
    const int iy0 = SA * my0 - 1, ix0 = SA * mx0 - 1 - XO;
    for (int item = tid; item < IR * (IC / 2); item += NT) {
      const int ir = item / (IC / 2), pp = item - ir * (IC / 2);
      const int gy = iy0 + ir, gx = ix0 + 2 * pp;
      const bool in = gy >= 0 && gy < H && gx >= 0 && gx < W;
      const uint32_t off = in ? (uint32_t)(gy * xpitch + gx) * 2u : 0x80000000u;
      const uint32_t hm = (gx + 1 < W) ? 0xffffffffu : 0x0000ffffu;       // odd W: the second pixel of the last pair is padding
      if constexpr (QUAD) {
        uint32_t ch[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) ch[c] = __builtin_amdgcn_raw_buffer_load_b32(xr, off + (uint32_t)c * plane, 0, 0) & hm;
        uint2 e0, e1;
        e0.x = __builtin_amdgcn_perm(ch[1], ch[0], 0x05040100u); e1.x = __builtin_amdgcn_perm(ch[1], ch[0], 0x07060302u);
        e0.y = __builtin_amdgcn_perm(ch[3], ch[2], 0x05040100u); e1.y = __builtin_amdgcn_perm(ch[3], ch[2], 0x07060302u);
        uint2* q = reinterpret_cast<uint2*>(in_s);
        q[ir * IC + 2 * pp] = e0;
        q[ir * IC + 2 * pp + 1] = e1;
      }
#pragma unroll
      for (int o = 0; o < K0; ++o) {
        uint32_t ch[8];
#pragma unroll
        for (int c = 0; c < 8; ++c)          // (planes >= Cin: out of the descriptor's range, zeros without a memory access.  A uniform branch
                                             // around those loads serialises the others — measured: the kernel twice as slow)
          ch[c] = __builtin_amdgcn_raw_buffer_load_b32(xr, off + (uint32_t)(8 * o + c) * plane, 0, 0) & hm;
        uint4 e0, e1;
        e0.x = __builtin_amdgcn_perm(ch[1], ch[0], 0x05040100u); e1.x = __builtin_amdgcn_perm(ch[1], ch[0], 0x07060302u);
        e0.y = __builtin_amdgcn_perm(ch[3], ch[2], 0x05040100u); e1.y = __builtin_amdgcn_perm(ch[3], ch[2], 0x07060302u);
        e0.z = __builtin_amdgcn_perm(ch[5], ch[4], 0x05040100u); e1.z = __builtin_amdgcn_perm(ch[5], ch[4], 0x07060302u);
        e0.w = __builtin_amdgcn_perm(ch[7], ch[6], 0x05040100u); e1.w = __builtin_amdgcn_perm(ch[7], ch[6], 0x07060302u);
        in_s[(o * IR + ir) * IC + 2 * pp] = e0;
        in_s[(o * IR + ir) * IC + 2 * pp + 1] = e1;
      }
    }
  }
  // layer A's operands while the loads fly: weights [step][lane][8], bias of channel (e & 3) + 8 (e >> 2) + 4 kg
  uint4 wA[NSTEP];
#pragma unroll
  for (int s = 0; s < NSTEP; ++s) wA[s] = *reinterpret_cast<const uint4*>(wa + ((size_t)s * 64 + lane) * 8);
  constexpr int C1 = 8 * C1O;
  f32x16 biasA;
#pragma unroll
  for (int e = 0; e < 16; ++e) {
    const int c = (e & 3) + 8 * (e >> 2) + 4 * kg;
    biasA[e] = (c < C1) ? ba[c] : 0.f;
  }
  // this lane's B-operand offset of every k-step inside the input tile: (channel octet, kernel row, kernel column)
  int offA[NSTEP], offA2[NSTEP];                     // (offA2: QUAD only — the second tap of the lane's k-octet)
#pragma unroll
  for (int s = 0; s < NSTEP; ++s) {
    int tap, oct;
    if constexpr (QUAD) { tap = 4 * s + 2 * kg; oct = 0; }
    else if constexpr (K0 == 1) { tap = 2 * s + kg; oct = 0; }
    else { tap = s; oct = kg; }
    const int t1 = tap > 8 ? 8 : tap, t2 = tap + 1 > 8 ? 8 : tap + 1;      // (taps 9 .. 11 do not exist: their weights are zero)
    offA[s] = (oct * IR + t1 / 3) * IC + t1 % 3 + XO;
    offA2[s] = (t2 / 3) * IC + t2 % 3 + XO;
  }
  __syncthreads();

  // ---- phase A: layer A on the MR x MC pixels layer B reads, 32 at a time -> mid tile (16-bit, zero outside layer A's output)
  {
    constexpr int NPX = MR * MC, NTILE = (NPX + 31) / 32;
    for (int t = wave; t < NTILE; t += 4) {
      const int p = t * 32 + px;
      const bool pv = p < NPX;
      const int pc = pv ? p : NPX - 1;
      const int mr = pc / MC, mc = pc - mr * MC;
      f32x16 acc = biasA;
      if constexpr (QUAD) {
        const uint2* ib = reinterpret_cast<const uint2*>(in_s) + SA * (mr * IC + mc);
#pragma unroll
        for (int s = 0; s < NSTEP; ++s) {
          const uint2 lo = ib[offA[s]], hi = ib[offA2[s]];
          acc = Mma32<T>::mma(wA[s], make_uint4(lo.x, lo.y, hi.x, hi.y), acc);
        }
      } else {
        const uint4* ib = in_s + SA * (mr * IC + mc);
#pragma unroll
        for (int s = 0; s < NSTEP; ++s) acc = Mma32<T>::mma(wA[s], ib[offA[s]], acc);
      }
      const int my = my0 + mr, mx = mx0 + mc;
      const bool inside = my >= 0 && my < Hm && mx >= 0 && mx < Wm;
      if (pv) {
#pragma unroll
        for (int o = 0; o < C1O; ++o) {
          float v[4];
#pragma unroll
          for (int q = 0; q < 4; ++q) { v[q] = acc[4 * o + q]; v[q] = fmaxf(v[q], v[q] * slope_a); }
          u32x2 w2;
          w2.x = inside ? pack2<T>(v[0], v[1]) : 0u;
          w2.y = inside ? pack2<T>(v[2], v[3]) : 0u;
          *reinterpret_cast<u32x2*>(reinterpret_cast<unsigned char*>(&mid_s[(o * MR + mr) * MC + mc]) + kg * 8) = w2;
        }
      }
    }
  }
  // layer B's operands
  uint4 wB[9][KSB];
#pragma unroll
  for (int tap = 0; tap < 9; ++tap)
#pragma unroll
    for (int ks = 0; ks < KSB; ++ks) wB[tap][ks] = *reinterpret_cast<const uint4*>(wb + ((size_t)(tap * KSB + ks) * 64 + lane) * 8);
  f32x16 acc[RPWB];
#pragma unroll
  for (int e = 0; e < 16; ++e) {
    const int c = (e & 3) + 8 * (e >> 2) + 4 * kg;
    const float bv = (c < C2) ? bb[c] : 0.f;
#pragma unroll
    for (int rr = 0; rr < RPWB; ++rr) acc[rr][e] = bv;
  }
  __syncthreads();

  // ---- phase B: the second layer from the mid tile
#pragma unroll
  for (int tap = 0; tap < 9; ++tap) {
    const int ky = tap / 3, kx = tap - 3 * ky;
#pragma unroll
    for (int ks = 0; ks < KSB; ++ks)
#pragma unroll
      for (int rr = 0; rr < RPWB; ++rr) {
        const int r = wave * RPWB + rr;
        const uint4 b = mid_s[((2 * ks + kg) * MR + SB * r + ky) * MC + SB * px + kx];
        acc[rr] = Mma32<T>::mma(wB[tap][ks], b, acc[rr]);
      }
  }
  if constexpr (!YC8) {
    // 16-byte stores through a per-wave LDS patch (conv_kernel.hpp epilogue_wide: 2-byte stores issue at ~2 B/clk/CU — 31 MB of output
    // would take 29 us by themselves) when the output rows are 16-byte aligned; the patches reuse the tiles' LDS
    if (((ypitch & 7) | (int)(ybs & 7) | (int)(reinterpret_cast<uintptr_t>(y) & 15)) == 0) {
      __syncthreads();                               // every wave is done with the mid tile
      conv::epilogue_wide<T, RPWB>(acc, reinterpret_cast<unsigned char*>(lds) + wave * conv::EPI_WAVE_BYTES, y + (size_t)n * ybs, C2, Ho, Wo,
                                   0, lane, x0, __builtin_amdgcn_readfirstlane(y0 + wave * RPWB), 1, slope_b, ypitch);
      return;
    }
  }
  const int gx = x0 + px;
#pragma unroll
  for (int rr = 0; rr < RPWB; ++rr) {
    const int gy = y0 + wave * RPWB + rr;
    if (gy >= Ho || gx >= Wo) continue;
    if constexpr (YC8) {
      // octets [c/8][Ho][Wo][8]: registers 4g .. 4g+3 are bytes [8 kg, +8) of entry (octet g, pixel)
      T* yo = y + (size_t)n * ybs;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        if (8 * g >= C2) continue;
        float v[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) { v[q] = acc[rr][4 * g + q]; v[q] = fmaxf(v[q], v[q] * slope_b); }
        u32x2 o2;
        o2.x = pack2<T>(v[0], v[1]); o2.y = pack2<T>(v[2], v[3]);
        *reinterpret_cast<u32x2*>(yo + (((size_t)g * Ho + gy) * Wo + gx) * 8 + kg * 4) = o2;
      }
    } else {
      typename Elem<T>::store_t* yo = reinterpret_cast<typename Elem<T>::store_t*>(y + (size_t)n * ybs);
#pragma unroll
      for (int e = 0; e < 16; e += 2) {
        const int c = (e & 3) + 8 * (e >> 2) + 4 * kg;
        float v0 = acc[rr][e], v1 = acc[rr][e + 1];
        v0 = fmaxf(v0, v0 * slope_b); v1 = fmaxf(v1, v1 * slope_b);
        const uint32_t p2 = pack2<T>(v0, v1);
        if (c < C2) yo[((size_t)c * Ho + gy) * ypitch + gx] = (typename Elem<T>::store_t)(p2 & 0xffffu);
        if (c + 1 < C2) yo[((size_t)(c + 1) * Ho + gy) * ypitch + gx] = (typename Elem<T>::store_t)(p2 >> 16);
      }
    }
  }
}

struct PairArgs {
  const void* x; long long xbs; int xpitch, Cin; const void* wa; const float* ba; float slope_a; int C1;
  const void* wb; const float* bb; float slope_b; int C2; void* y; long long ybs; int ypitch; int B, H, W; hipStream_t stream;
  int sa, sb;
};

template <typename T, int K0, int C1O, int TH, bool YC8, int SA, int SB>
int launch_pair(const PairArgs& a) {
  using G = Geo<SA, SB, TH>;
  const int Hm = (a.H - 1) / SA + 1, Wm = (a.W - 1) / SA + 1;
  const int Ho = (Hm - 1) / SB + 1, Wo = (Wm - 1) / SB + 1;
  const int tiles_x = cdiv(Wo, 32), tiles_y = cdiv(Ho, TH);
  const size_t ldsb = (size_t)((K0 == 0 ? (G::IR * G::IC + 1) / 2 : K0 * G::IR * G::IC) + C1O * G::MR * G::MC) * 16;
  static LdsOptIn opt;
  auto kern = &conv_pair_kernel<T, K0, C1O, TH, YC8, SA, SB>;
  opt.ensure(reinterpret_cast<const void*>(kern), ldsb);
  hipLaunchKernelGGL(kern, dim3((unsigned)(a.B * tiles_x * tiles_y)), dim3(NT), ldsb, a.stream, (const T*)a.x, a.xbs, a.xpitch, a.Cin,
                     (const T*)a.wa, a.ba, a.slope_a, (const T*)a.wb, a.bb, a.slope_b, (T*)a.y, a.ybs, a.ypitch, a.C2, a.H, a.W, Hm, Wm, Ho, Wo, tiles_x, tiles_y);
  return check_launch("conv_pair_forward");
}

// tile heights: two workgroups per CU (LDS: input tile + mid tile <= 80 KB)
template <typename T, bool YC8>
int dispatch_pair(const PairArgs& a) {
  const int k0 = a.Cin <= 4 ? 0 : (a.Cin <= 8 ? 1 : 2), c1o = a.C1 / 8;
  if (a.sa == 1) {                                   // [stride 1, stride 2]: the SGU guidance stem
    if (k0 == 0 && c1o == 2) return launch_pair<T, 0, 2, 8, YC8, 1, 2>(a);
    if (k0 == 0 && c1o == 4) return launch_pair<T, 0, 4, 4, YC8, 1, 2>(a);
    if (k0 == 1 && c1o == 2 && conv::g_pair_th == 4) return launch_pair<T, 1, 2, 4, YC8, 1, 2>(a);
    if (k0 == 1 && c1o == 2) return launch_pair<T, 1, 2, 8, YC8, 1, 2>(a);
    if (k0 == 1 && c1o == 4) return launch_pair<T, 1, 4, 4, YC8, 1, 2>(a);
    if (k0 == 2 && c1o == 2) return launch_pair<T, 2, 2, 8, YC8, 1, 2>(a);
    return launch_pair<T, 2, 4, 4, YC8, 1, 2>(a);
  }
  if constexpr (!YC8) {                              // [stride 2, stride 1]: a stage of the feature pyramid (NCHW out)
    if (k0 == 0 && c1o == 2) return launch_pair<T, 0, 2, 8, false, 2, 1>(a);
    if (k0 == 0 && c1o == 4) return launch_pair<T, 0, 4, 8, false, 2, 1>(a);
    if (k0 == 1 && c1o == 2 && conv::g_pair_th == 4) return launch_pair<T, 1, 2, 4, false, 2, 1>(a);
    if (k0 == 1 && c1o == 2) return launch_pair<T, 1, 2, 8, false, 2, 1>(a);
    if (k0 == 1 && c1o == 4) return launch_pair<T, 1, 4, 8, false, 2, 1>(a);
    if (k0 == 2 && c1o == 2) return launch_pair<T, 2, 2, 8, false, 2, 1>(a);
    return launch_pair<T, 2, 4, 8, false, 2, 1>(a);
  }
  set_error("conv_pair_forward: [stride 2, stride 1] writes NCHW");
  return UPF_EUNSUPPORTED;
}

}  // namespace convp
}  // namespace upf

extern "C" int upf_conv_pair_forward(const void* x, long long x_batch_stride, int x_row_pitch, int Cin,
                                     const void* wa_packed, const float* bias_a, float slope_a, int C1, int stride_a,
                                     const void* wb_packed, const float* bias_b, float slope_b, int C2, int stride_b,
                                     void* y, long long y_batch_stride, int y_row_pitch, int y_is_c8,
                                     int B, int H, int W, int dtype, void* stream) {
  using namespace upf;
  UPF_REQUIRE(x && wa_packed && bias_a && wb_packed && bias_b && y, UPF_EINVAL, "conv_pair_forward: null pointer");
  UPF_REQUIRE(B > 0 && H > 0 && W > 0, UPF_EINVAL, "conv_pair_forward: bad shape B=%d H=%d W=%d", B, H, W);
  UPF_REQUIRE(dtype == UPF_BF16 || dtype == UPF_F16, UPF_EDTYPE, "conv_pair_forward: bf16 / fp16 only");
  UPF_REQUIRE(Cin >= 1 && Cin <= 16 && (C1 == 16 || C1 == 32) && C2 >= 1 && C2 <= 32, UPF_EUNSUPPORTED,
              "conv_pair_forward: Cin <= 16, C1 = 16 | 32, C2 <= 32 (got %d, %d, %d)", Cin, C1, C2);
  UPF_REQUIRE((stride_a == 1 && stride_b == 2) || (stride_a == 2 && stride_b == 1), UPF_EUNSUPPORTED,
              "conv_pair_forward: strides (1, 2) or (2, 1), got (%d, %d)", stride_a, stride_b);
  if (x_row_pitch == 0) x_row_pitch = W;
  const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;      // (one of the two layers halves the size)
  if (y_row_pitch == 0) y_row_pitch = Wo;
  UPF_REQUIRE(x_row_pitch >= W && x_row_pitch % 2 == 0 && aligned_to(x, 4) && x_batch_stride % 2 == 0, UPF_EUNSUPPORTED,
              "conv_pair_forward: the input rows are read as pixel pairs: an even row pitch (%d) and a 4-byte aligned base", x_row_pitch);
  UPF_REQUIRE(slope_a >= 0.f && slope_a <= 1.f && slope_b >= 0.f && slope_b <= 1.f, UPF_EINVAL, "conv_pair_forward: leaky slopes in [0,1]");
  UPF_REQUIRE(aligned_to(wa_packed, 16) && aligned_to(wb_packed, 16), UPF_EINVAL, "conv_pair_forward: packed weights must be 16-byte aligned");
  UPF_REQUIRE(y_is_c8 ? (aligned_to(y, 16) && y_batch_stride % 8 == 0) : (y_row_pitch >= Wo), UPF_EINVAL, "conv_pair_forward: bad output layout");
  UPF_REQUIRE((long long)Cin * H * x_row_pitch * 2 < (1ll << 31), UPF_EINVAL, "conv_pair_forward: image too large for one buffer descriptor");
  convp::PairArgs a{x, x_batch_stride, x_row_pitch, Cin, wa_packed, bias_a, slope_a == 0.f ? 1.f : slope_a, C1,
                    wb_packed, bias_b, slope_b == 0.f ? 1.f : slope_b, C2, y, y_batch_stride, y_row_pitch, B, H, W, (hipStream_t)stream, stride_a, stride_b};
  if (dtype == UPF_BF16) return y_is_c8 ? convp::dispatch_pair<bf16_t, true>(a) : convp::dispatch_pair<bf16_t, false>(a);
  return y_is_c8 ? convp::dispatch_pair<f16_t, true>(a) : convp::dispatch_pair<f16_t, false>(a);
}
