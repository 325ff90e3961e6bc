// 3x3 (dilated) convolution as an implicit GEMM on the bf16/fp16 matrix cores — gfx950.
//
// the dense flow-estimator / context / SGU-estimator convolutions
// (/root/reference/model/pwc_modules.py:250-286, :396-412, model/upflow.py:24-60) are where an
// inference step actually spends its time (≈1.5 TFLOP per 384x1280 batch-4 step; MIOpen reaches ≈55
// TFLOP/s on them through im2col + GEMM + NCHW<->NHWC transposes + separate bias / LeakyReLU / concat
// kernels).  BASELINE.json's north star allows MFMA exactly here ("a real contraction").
//
//   y[n, co, i, j] = act( bias[co] + sum_{ci,ky,kx} w[co,ci,ky,kx] * x[n, ci, i+(ky-1)d, j+(kx-1)d] )
//   stride 1, padding = dilation d (same-size output), zero padding, act = LeakyReLU(slope) or identity.
//
// x and y are CHANNEL SLICES of larger contiguous NCHW buffers (batch strides given): the dense
// estimator's growing concatenation (`x1 = cat([conv1(x), x])`, pwc_modules.py:280-285) becomes one
// 565-channel buffer that every conv reads a suffix of and writes its own slice of — no concat copy,
// no separate bias or activation pass, no layout transposes, no im2col buffer.
//
// GEMM view per image: D[co][pixel] = sum_tap sum_ci W[tap][co][ci] * X[ci][pixel + shift(tap)],
// v_mfma_f32_32x32x16_{bf16,f16}: A = 32 output channels x 16 k, B = 16 k x 32 pixels (one tile row).
//   * workgroup = 4 waves = an 8x32 pixel tile of one image, ALL output channels (<= 128, MT tiles of 32);
//     wave w owns tile rows 2w, 2w+1 -> 2*MT accumulator tiles of 16 fp32 registers;
//   * per chunk of 32 input channels the x tile + halo is staged ONCE into LDS, transposed in registers
//     (8 channel rows x 8 pixels -> 8 pixels x 8 channels, 32 v_perm) into 16-byte entries
//     [channel-octet][row][col]; the 9 taps then read SHIFTED windows of it (per-lane LDS addresses), so
//     the im2col expansion exists only as LDS read addresses;  halo zeros and channels >= Cin come from
//     the buffer descriptor's bounds check;
//   * weights are pre-packed once ([tap][co][ci], ci padded to 32, co to 32) so that a wave's A operand
//     is a 16-byte LDS read; the 8 KB weight slice of the next tap is staged (double-buffered) while the
//     current tap's MFMAs run;
//   * epilogue: + bias, LeakyReLU, convert, store.
#include "common.hpp"
#include <cstdlib>

namespace upf {
namespace conv {

constexpr int TW = 32, NTHREADS = 256;   // tile = (4*RPW) rows x 32 pixels; RPW = rows per wave (2, or 4 for Cout <= 64)
// staged columns [S*x0 - marg, S*x0 + S*32 + marg), marg = 8 (d <= 8) or 16: 16-byte aligned global loads
__host__ __device__ constexpr int xw(int S, int marg) { return S * TW + 2 * marg; }
constexpr int MAXD = 16;                  // dilation limit (the context network's largest)

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

template <typename T> struct Mma32;
template <> struct Mma32<bf16_t> {
  static __device__ __forceinline__ f32x16 mma(uint4 a, uint4 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
  }
};
template <> struct Mma32<f16_t> {
  static __device__ __forceinline__ f32x16 mma(uint4 a, uint4 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
  }
};

__host__ __device__ constexpr int pad32(int v) { return (v + 31) / 32 * 32; }

// w [Cout, Cin, k, k] (k*k = ntaps) -> packed [slab = co/32][k-step = ci/16][tap][kg = (ci/8)%2][px = co%32][ci%8],
// zero padded to pad32(Cout) x pad32(Cin): the 1 KB block of one (slab, k-step, tap) is exactly the A operand of one
// v_mfma_f32_32x32x16 in lane order (lane = kg*32 + px holds 8 consecutive input channels of output channel px),
// so a wave fetches it with ONE fully coalesced 16-byte-per-lane load, and an LDS weight slice is 512-byte runs.
template <typename T>
__global__ void pack_weights_kernel(const T* __restrict__ w, T* __restrict__ wp, int Cin, int Cout, int ntaps) {
  const int cip = pad32(Cin), cop = pad32(Cout), nk = cip / 16;
  const long long total = (long long)ntaps * cop * cip;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int j = (int)(i & 7), px = (int)((i >> 3) & 31), kg = (int)((i >> 8) & 1);
    const long long b = i >> 9;                      // (slab * nk + kstep) * ntaps + tap
    const int tap = (int)(b % ntaps), kstep = (int)((b / ntaps) % nk), slab = (int)(b / ((long long)ntaps * nk));
    const int co = slab * 32 + px, ci = kstep * 16 + kg * 8 + j;
    T v; v.v = 0;
    if (ci < Cin && co < Cout) v = w[((size_t)co * Cin + ci) * ntaps + tap];
    wp[i] = v;
  }
}

// MT = number of 32-wide output-channel tiles (1..4).
// ALLTAPS (MT <= 2): the weight slices of all 9 taps of a channel chunk are staged together (9*MT*2 KB) and
// the tap loop runs without barriers — with few output channels the MFMA phase of one tap is far too short
// to hide the L2 latency of the next tap's weight prefetch, which then dominates (measured: 563->2 channels
// spent 5.7 us per chunk in nine exposed prefetch+barrier rounds).  MT >= 3 keeps the per-tap double buffer.
// RPW (rows per wave): Cout <= 32 uses 4 rows per wave = a 16x32 tile per workgroup — twice the pixels per
// staged weight slice, halo row and barrier, which is what bounds the narrow layers.
// S = stride (1 or 2; the feature pyramid's down-sampling convs): output pixel (i,j) reads input
// (S*i + (ky-1)d, S*j + (kx-1)d); only the staged window and the LDS read addresses change.
// VAR: 0 = 3x3, 8-column margins (dilation <= 8); 1 = 3x3, 16-column margins (dilation 16); 2 = 1x1;
// 3 = 3x3 with dilation exactly 1.  Compile-time so that the tap loop unrolls and the window addressing folds
// into immediates.
// MT == 1 (Cout <= 32 per workgroup — the narrow layers, which are most of the launches): with one 32-channel tile
// the LDS pipe, not the matrix core, is the bound (1 A + RPW B reads of 1 KB per RPW MFMAs, four waves on one
// pipe), and the per-chunk weight staging (L2 latency + a barrier) heads every chunk.  So the weights skip LDS:
// each lane keeps its A operand of all 9 taps x 2 k-steps of the current chunk in 72 registers, loaded straight
// from the packed weights (L2-resident, identical for every workgroup) and RE-loaded for the next chunk tap by tap
// right after the tap's last use — the L2 latency hides under the remaining taps and the next chunk's staging.
// With dilation 1 / stride 1 (VAR 3) a staged row s feeds output rows r = s-ky, so each B window is read once
// per (s, kx) instead of once per (r, ky, kx): (RPW+2)*3 reads for 9*RPW MFMAs per k-step.
// GEN: the staged rows need not be 16-byte aligned (W % 8 != 0, or x is an odd channel slice): the 8-pixel group
// that would cross the end of its image row is loaded SHIFTED LEFT so that it ends at the row end (gfx950 executes
// the 2-byte-aligned 16-byte load; tools/unaligned_b128_probe.hip), and the shift is undone by the LDS entry index
// each transposed pixel is written to — no load ever leaves its row, so nothing depends on what follows the buffer.
template <typename T, int MT, bool ALLTAPS = (MT <= 2), int RPW = 2, int S = 1, int NOCTS = 4, int VAR = 0, bool GEN = false>
__global__ __launch_bounds__(NTHREADS, 2)
void conv3x3_kernel(const T* __restrict__ x, long long xbs, const T* __restrict__ wp, const float* __restrict__ bias,
                    T* __restrict__ y, long long ybs, int Cin, int Cout, int H, int W, int Ho, int Wo, int d,
                    int tiles_x, int tiles_y, float slope) {
  constexpr int nocts = NOCTS;
  constexpr int marg = (VAR == 1) ? 16 : 8;
  constexpr int ntaps = (VAR == 2) ? 1 : 9;
  constexpr bool WREG = (MT == 1);                   // weights live in registers, not LDS (see the tap loops)
  constexpr int KS = nocts / 2;                      // k-steps of 16 channels per chunk
  if constexpr (VAR == 3) d = 1;                     // compile-time dilation 1: the window addressing folds
  // nocts: channel octets per chunk (4 = 32 channels; 2 when Cin <= 16 or when the dilation-16 halo would not
  // fit LDS otherwise).  ntaps: 9, or 1 for a 1x1 convolution (then d = 0 and only the centre tap exists).
  extern __shared__ __attribute__((aligned(16))) uint4 smem[];
  constexpr int TH = 4 * RPW;
  constexpr int XW = xw(S, marg);
  const int rows = S * (TH - 1) + 2 * d + 1;         // staged input rows
  const int XS_E = nocts * rows * XW;                // entries (16 B = 8 channels of one pixel) of the x tile
  constexpr int AS_MAX = nocts * MT * 32;
  constexpr int AS_E = nocts * MT * 32;              // entries of one weight slice: [octet][co]
  constexpr int KCH = nocts * 8;                     // input channels per chunk
  uint4* xs = smem;                                  // [octet 4][rows][XW]
  uint4* as = smem + XS_E;                           // [2 (or 9 with ALLTAPS)][octet 4][MT*32]

  const int bid = xcd_remap(blockIdx.x, gridDim.x);
  const int tx = bid % tiles_x, ty = (bid / tiles_x) % tiles_y, n = bid / (tiles_x * tiles_y);
  const int x0 = tx * TW, y0 = ty * TH;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int cip = pad32(Cin), cop = MT * 32;         // cop: output channels of THIS workgroup (one blockIdx.y slab)
  const int copt = pad32(Cout);                      // packed weight rows
  const int co0 = blockIdx.y * cop;                  // first output channel of the slab (Cout split over blockIdx.y)
  const int HW = H * W;

  // buffer descriptor over this image's Cin input planes: rows/cols outside the image get offset
  // 0x80000000, channel planes >= Cin fall off the end -> the hardware returns the zero padding
  const uint32_t plane = (uint32_t)HW * 2u;
  __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(x + (size_t)n * xbs), 0, (uint32_t)Cin * plane, 0x00020000);
  __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(wp), 0, (uint32_t)ntaps * (uint32_t)copt * (uint32_t)cip * 2u, 0x00020000);

  // x staging tasks: (octet, row, 8-pixel group) -> 8 loads (8 channel rows), 32 perms, 8 LDS entries
  constexpr int ngroups = XW / 8;
  const int ntasks = nocts * rows * ngroups;

  f32x16 acc[RPW][MT];
#pragma unroll
  for (int r = 0; r < RPW; ++r)
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[r][m][e] = 0.f;

  const int px = lane & 31, kg = lane >> 5;          // MFMA operand lane: column / row index, k-octet within the k-step
  const int nchunks = cip / KCH;
  // byte offset of the 1 KB packed block (slab m of this workgroup, global k-step, tap)
  const int nksteps = cip / 16;
  auto woff = [&](int m, int kstep, int tap) { return (uint32_t)((((int)blockIdx.y * MT + m) * nksteps + kstep) * ntaps + tap) * 1024u; };

  // x staging task t -> (channel octet, staged row, 8-pixel group): buffer-load offset of channel 0 of the
  // octet in chunk 0 (0x80000000 = outside the image) and the LDS entry it fills
  // sh (GEN only): pixels by which the load window is shifted left so that it ends at the row end
  auto task_geom = [&](int t, uint32_t& off, int& dst, int& sh) {
    const int oct = t / (rows * ngroups), rem = t - oct * (rows * ngroups), r = rem / ngroups, g = rem - r * ngroups;
    const int gy = S * y0 - d + r, gx = S * x0 - marg + 8 * g;
    const bool in = (t < ntasks) && gy >= 0 && gy < H && gx >= 0 && gx < W;      // !GEN: W % 8 == 0, a group is all in or all out
    sh = (GEN && in && gx + 8 > W) ? gx + 8 - W : 0;
    off = in ? ((uint32_t)((oct * 8) * HW + gy * W + gx - sh) * 2u) : 0x80000000u;
    dst = (oct * rows + r) * XW + 8 * g;
  };
  auto task_load = [&](uint32_t off, int cc, u32x4 (&ch)[8]) {
    const uint32_t o = off + (uint32_t)cc * KCH * plane;                           // stays >= 2^31 for outside tasks
#pragma unroll
    for (int k = 0; k < 8; ++k) ch[k] = __builtin_amdgcn_raw_buffer_load_b128(xr, o + k * plane, 0, 0);
  };
  auto task_store = [&](int dsti, int sh, const u32x4 (&ch)[8]) {                 // 8 channel rows x 8 px -> 8 px x 8 channels
    uint4* dst = xs + dsti;
#pragma unroll
    for (int pp = 0; pp < 4; ++pp) {               // dword pp of every channel row = pixels 2pp, 2pp+1
      uint4 e0, e1;
      e0.x = __builtin_amdgcn_perm(ch[1][pp], ch[0][pp], 0x05040100u); e1.x = __builtin_amdgcn_perm(ch[1][pp], ch[0][pp], 0x07060302u);
      e0.y = __builtin_amdgcn_perm(ch[3][pp], ch[2][pp], 0x05040100u); e1.y = __builtin_amdgcn_perm(ch[3][pp], ch[2][pp], 0x07060302u);
      e0.z = __builtin_amdgcn_perm(ch[5][pp], ch[4][pp], 0x05040100u); e1.z = __builtin_amdgcn_perm(ch[5][pp], ch[4][pp], 0x07060302u);
      e0.w = __builtin_amdgcn_perm(ch[7][pp], ch[6][pp], 0x05040100u); e1.w = __builtin_amdgcn_perm(ch[7][pp], ch[6][pp], 0x07060302u);
      if constexpr (GEN) {                           // loaded pixel q is column q - sh of the group; columns >= 8 - sh are past the row end
        const uint32_t m0 = (2 * pp >= sh) ? 0xffffffffu : 0u, m1 = (2 * pp + 1 >= sh) ? 0xffffffffu : 0u;
        e0.x &= m0; e0.y &= m0; e0.z &= m0; e0.w &= m0;
        e1.x &= m1; e1.y &= m1; e1.z &= m1; e1.w &= m1;
        dst[(2 * pp - sh) & 7] = e0;
        dst[(2 * pp + 1 - sh) & 7] = e1;
      } else {
        dst[2 * pp] = e0;
        dst[2 * pp + 1] = e1;
      }
    }
  };
  // PREFETCH (MT <= 3, where the register budget allows 32 more VGPRs): this thread's first x task of chunk
  // cc+1 is loaded into registers BEFORE the tap loop of chunk cc and lands in LDS after it, so the HBM/L2
  // latency of the staging hides under the matrix work instead of heading every chunk.
  constexpr bool PREFETCH = (MT <= 3) && !(MT == 1 && RPW == 4 && NOCTS == 4);   // (that one holds 72 weight registers)
  uint32_t off0 = 0x80000000u; int dst0 = 0, sh0 = 0;
  u32x4 pre[8];
  if constexpr (PREFETCH) {
    task_geom(tid, off0, dst0, sh0);
    task_load(off0, 0, pre);
  }

  // WREG: this lane's A operands (row px of the weight tile, k-octet kg of each k-step) for every tap of a chunk
  uint4 wa[WREG ? ntaps : 1][KS];
  auto wload = [&](int cc, int tap, int ks) {
    const uint32_t off = (cc < nchunks) ? woff(0, cc * KS + ks, tap) + (uint32_t)lane * 16u : 0x80000000u;
    return __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(wr, off, 0, 0));
  };
  if constexpr (WREG) {
#pragma unroll
    for (int tap = 0; tap < ntaps; ++tap)
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) wa[tap][ks] = wload(0, tap, ks);
  }

  for (int cc = 0; cc < nchunks; ++cc) {
    __syncthreads();                                 // previous chunk fully consumed
    // ---- stage the x tile (+halo) of channels [32cc, 32cc+32)
    if constexpr (PREFETCH) { if (tid < ntasks) task_store(dst0, sh0, pre); }
    for (int t = tid + (PREFETCH ? NTHREADS : 0); t < ntasks; t += NTHREADS) {
      uint32_t off; int dsti, sh;
      task_geom(t, off, dsti, sh);
      u32x4 ch[8];
      task_load(off, cc, ch);
      task_store(dsti, sh, ch);
    }
    // ---- weight slices: tap 0 -> as[0]  (ALLTAPS: all 9 taps -> as[0..8]);  WREG: nothing to stage
    for (int e = tid; !WREG && e < (ALLTAPS ? ntaps : 1) * AS_E; e += NTHREADS) {
      const int tap0 = e / AS_E, r0 = e - tap0 * AS_E;
      const int oct = r0 / cop, co = r0 - oct * cop;
      const uint32_t off = woff(co >> 5, cc * KS + (oct >> 1), tap0) + (uint32_t)(((oct & 1) * 32 + (co & 31)) * 16);
      as[e] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(wr, off, 0, 0));
    }
    __syncthreads();
    if constexpr (PREFETCH) { if (cc + 1 < nchunks) task_load(off0, cc + 1, pre); }

    if constexpr (WREG) {
      if constexpr (VAR == 3 && S == 1) {
#pragma unroll
        for (int sr = 0; sr < RPW + 2; ++sr) {       // staged row sr of this wave's strip feeds output rows r = sr - ky
#pragma unroll
          for (int kx = 0; kx < 3; ++kx)
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
              const uint4 b = xs[((2 * ks + kg) * rows + RPW * wave + sr) * XW + marg + px + kx - 1];
#pragma unroll
              for (int r = 0; r < RPW; ++r)
                if (sr - r >= 0 && sr - r <= 2) acc[r][0] = Mma32<T>::mma(wa[(sr - r) * 3 + kx][ks], b, acc[r][0]);
            }
          if (sr - (RPW - 1) >= 0 && sr - (RPW - 1) <= 2) {   // kernel row ky = sr-(RPW-1) is finished: fetch the next chunk's
#pragma unroll
            for (int kx = 0; kx < 3; ++kx)
#pragma unroll
              for (int ks = 0; ks < KS; ++ks) wa[(sr - (RPW - 1)) * 3 + kx][ks] = wload(cc + 1, (sr - (RPW - 1)) * 3 + kx, ks);
          }
        }
      } else {
#pragma unroll
        for (int tap = 0; tap < ntaps; ++tap) {
          const int ky = (ntaps == 1) ? 1 : tap / 3, kx = (ntaps == 1) ? 1 : tap - 3 * (tap / 3);
          const int col = marg + S * px + (kx - 1) * d;
#pragma unroll
          for (int ks = 0; ks < KS; ++ks)
#pragma unroll
            for (int r = 0; r < RPW; ++r) {
              const uint4 b = xs[((2 * ks + kg) * rows + S * (RPW * wave + r) + ky * d) * XW + col];
              acc[r][0] = Mma32<T>::mma(wa[tap][ks], b, acc[r][0]);
            }
#pragma unroll
          for (int ks = 0; ks < KS; ++ks) wa[tap][ks] = wload(cc + 1, tap, ks);
        }
      }
    } else {
      for (int tap = 0; tap < ntaps; ++tap) {
        const uint4* acur = as + (ALLTAPS ? tap : (tap & 1)) * AS_E;
        // prefetch the next tap's weight slice into the other buffer (consumed after the barrier below)
        u32x4 wpre[(AS_MAX + NTHREADS - 1) / NTHREADS];
        if (!ALLTAPS && tap + 1 < ntaps) {
#pragma unroll
          for (int j = 0; j < (AS_MAX + NTHREADS - 1) / NTHREADS; ++j) {
            const int e = tid + j * NTHREADS;
            const int oct = e / cop, co = e - oct * cop;
            const uint32_t off = (e < AS_E) ? woff(co >> 5, cc * KS + (oct >> 1), tap + 1) + (uint32_t)(((oct & 1) * 32 + (co & 31)) * 16) : 0x80000000u;
            wpre[j] = __builtin_amdgcn_raw_buffer_load_b128(wr, off, 0, 0);
          }
        }
        const int ky = (ntaps == 1) ? 1 : tap / 3, kx = (ntaps == 1) ? 1 : tap - 3 * (tap / 3);
        // shifted window: output pixel (row, px) reads staged entry (row + ky*d, 8 + px + (kx-1)*d)
        const int col = marg + S * px + (kx - 1) * d;
#pragma unroll
        for (int ks = 0; ks < nocts / 2; ++ks) {       // k-steps of 16 channels
          const int oct = 2 * ks + kg;
          uint4 a[MT];
#pragma unroll
          for (int m = 0; m < MT; ++m) a[m] = acur[oct * cop + m * 32 + px];
#pragma unroll
          for (int r = 0; r < RPW; ++r) {
            const uint4 b = xs[(oct * rows + S * (RPW * wave + r) + ky * d) * XW + col];
#pragma unroll
            for (int m = 0; m < MT; ++m) acc[r][m] = Mma32<T>::mma(a[m], b, acc[r][m]);
          }
        }
        if (!ALLTAPS && tap + 1 < ntaps) {
          uint4* anext = as + ((tap + 1) & 1) * AS_E;
#pragma unroll
          for (int j = 0; j < (AS_MAX + NTHREADS - 1) / NTHREADS; ++j) {
            const int e = tid + j * NTHREADS;
            if (e < AS_E) anext[e] = __builtin_bit_cast(uint4, wpre[j]);
          }
          __syncthreads();
        }
      }
  
    }
  }

  // ---- epilogue: D[co][pixel]; lane -> pixel column px, regs -> co = (e&3) + 8*(e>>2) + 4*kg
  using st = uint16_t;
  st* yb = reinterpret_cast<st*>(y) + (size_t)n * ybs;
#pragma unroll
  for (int r = 0; r < RPW; ++r) {
    const int gy = y0 + RPW * wave + r, gx = x0 + px;
    if (gy >= Ho || gx >= Wo) continue;
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int co = co0 + m * 32 + (e & 3) + 8 * (e >> 2) + 4 * kg;
        if (co < Cout) {
          float v = acc[r][m][e] + bias[co];
          v = (slope != 0.f) ? fmaxf(v, v * slope) : v;
          T tmp;
          Elem<T>::store(&tmp, v);
          yb[(size_t)co * (Ho * Wo) + gy * Wo + gx] = tmp.v;
        }
      }
  }
}

struct Args {
  const void* x; long long xbs; const void* wp; const float* bias; void* y; long long ybs;
  int B, Cin, Cout, H, W, d, stride, ntaps; float slope; hipStream_t stream;
};

template <typename T, int MT, int RPW, int S, bool GEN = false>
int launch_rpw(const Args& a, int slabs = 1) {
  constexpr int TH = 4 * RPW;
  const int Ho = (a.H - 1) / S + 1, Wo = (a.W - 1) / S + 1;
  const int tiles_x = cdiv(Wo, TW), tiles_y = cdiv(Ho, TH);
  const int rows = S * (TH - 1) + 2 * a.d + 1;
  const int marg = (a.d <= 8) ? 8 : 16;
  const int wslices = (MT == 1) ? 0 : (MT <= 2) ? a.ntaps : 2;      // MT == 1 keeps its weights in registers
  auto lds_for = [&](int nocts) { return (size_t)(nocts * rows * xw(S, marg) + wslices * nocts * MT * 32) * 16; };
  int nocts = (a.Cin <= 16 || marg == 16) ? 2 : 4;
  const size_t lds = lds_for(nocts);
  UPF_REQUIRE(lds <= 160 * 1024, UPF_EUNSUPPORTED, "conv_forward: tile does not fit LDS (dilation %d, stride %d)", a.d, S);
  const dim3 grid((unsigned)(a.B * tiles_x * tiles_y), slabs);
#define UPF_CONV_LAUNCH(NO, VAR)                                                                                                   \
  {                                                                                                                                \
    static size_t attr_lds = 0;                                                                                                    \
    if (lds > attr_lds) {                                                                                                          \
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_kernel<T, MT, (MT <= 2), RPW, S, NO, VAR, GEN>),            \
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                                             \
      attr_lds = lds;                                                                                                              \
    }                                                                                                                              \
    hipLaunchKernelGGL((conv3x3_kernel<T, MT, (MT <= 2), RPW, S, NO, VAR, GEN>), grid, dim3(NTHREADS), lds, a.stream, (const T*)a.x, \
                       a.xbs, (const T*)a.wp, a.bias, (T*)a.y, a.ybs, a.Cin, a.Cout, a.H, a.W, Ho, Wo, a.d, tiles_x, tiles_y,      \
                       a.slope);                                                                                                   \
  }
  if constexpr (S == 1 && RPW == 2) {
    if (a.ntaps == 1) { if (nocts == 4) UPF_CONV_LAUNCH(4, 2) else UPF_CONV_LAUNCH(2, 2) return check_launch("conv_forward"); }
    if (marg == 16) { UPF_CONV_LAUNCH(2, 1) return check_launch("conv_forward"); }
  }
  if constexpr (MT == 1) {
    if (a.d == 1 && a.ntaps == 9) { if (nocts == 4) UPF_CONV_LAUNCH(4, 3) else UPF_CONV_LAUNCH(2, 3) return check_launch("conv_forward"); }
  }
  if (nocts == 4) UPF_CONV_LAUNCH(4, 0) else UPF_CONV_LAUNCH(2, 0)
#undef UPF_CONV_LAUNCH
  return check_launch("conv_forward");
}

template <typename T, int MT>
int launch(const Args& a) {
  if (a.stride == 2) return launch_rpw<T, MT, 2, 2>(a);
  if constexpr (MT == 1) {
    // narrow layers: 16x32 tiles (4 rows per wave) halve the weight / halo / barrier cost per pixel, but only when
    // the grid still fills the chip twice over (256 CUs x 2 resident workgroups)
    static const int rpw4_min = getenv("UPF_CONV_RPW4_MIN") ? atoi(getenv("UPF_CONV_RPW4_MIN")) : 256;
    if ((long long)a.B * cdiv(a.W, TW) * cdiv(a.H, 16) >= rpw4_min && a.d <= 8 && a.ntaps == 9) return launch_rpw<T, MT, 4, 1>(a);
  }
  if constexpr (MT > 1) {
    // coarse pyramid levels: too few pixel tiles to fill 256 CUs -> split the OUTPUT CHANNELS over blockIdx.y
    // (each slab re-stages the x tile, which is irrelevant when the grid is latency-bound)
    const long long tiles = (long long)a.B * cdiv(a.W, TW) * cdiv(a.H, 8);
    if (tiles * 2 <= 512) {
      if (MT % 2 == 0 && tiles * (MT / 2) >= 384) return launch_rpw<T, 2, 2, 1>(a, MT / 2);
      return launch_rpw<T, 1, 2, 1>(a, MT);
    }
  }
  return launch_rpw<T, MT, 2, 1>(a);
}

// Output-channel slabs of 32 or 64 over blockIdx.y: any Cout (the pyramid's 196-channel level), and the GEN
// (unaligned-row) variant, which only the small pyramid levels of inputs that are not multiples of 512 reach.
template <typename T, bool GEN>
int launch_slabs(const Args& a) {
  const int mt = cdiv(a.Cout, 32);
  const int Ho = (a.H - 1) / a.stride + 1, Wo = (a.W - 1) / a.stride + 1;
  const long long tiles = (long long)a.B * cdiv(Wo, TW) * cdiv(Ho, 8);
  const bool two = mt >= 2 && tiles * ((mt + 1) / 2) >= 384;     // 64-wide slabs once they still fill the chip
  if (a.stride == 2) return two ? launch_rpw<T, 2, 2, 2, GEN>(a, (mt + 1) / 2) : launch_rpw<T, 1, 2, 2, GEN>(a, mt);
  return two ? launch_rpw<T, 2, 2, 1, GEN>(a, (mt + 1) / 2) : launch_rpw<T, 1, 2, 1, GEN>(a, mt);
}

}  // namespace conv
}  // namespace upf

extern "C" long long upf_conv_packed_bytes(int Cin, int Cout, int kernel_size) {
  return (long long)kernel_size * kernel_size * upf::conv::pad32(Cout) * upf::conv::pad32(Cin) * 2;
}

extern "C" int upf_conv_pack_weights(const void* w, void* w_packed, int Cin, int Cout, int kernel_size, int dtype, void* stream) {
  using namespace upf;
  UPF_REQUIRE(w && w_packed && Cin > 0 && Cout > 0, UPF_EINVAL, "conv_pack_weights: bad arguments");
  UPF_REQUIRE(kernel_size == 3 || kernel_size == 1, UPF_EUNSUPPORTED, "conv_pack_weights: kernel_size %d (1 or 3)", kernel_size);
  UPF_REQUIRE(dtype == UPF_BF16 || dtype == UPF_F16, UPF_EDTYPE, "conv: bf16 / fp16 only (fp32 convolutions stay with MIOpen)");
  const int ntaps = kernel_size * kernel_size;
  const long long total = (long long)ntaps * conv::pad32(Cout) * conv::pad32(Cin);
  const int blocks = (int)((total + 255) / 256 > 4096 ? 4096 : (total + 255) / 256);
  if (dtype == UPF_BF16)
    hipLaunchKernelGGL((conv::pack_weights_kernel<bf16_t>), dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)w, (bf16_t*)w_packed, Cin, Cout, ntaps);
  else
    hipLaunchKernelGGL((conv::pack_weights_kernel<f16_t>), dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const f16_t*)w, (f16_t*)w_packed, Cin, Cout, ntaps);
  return check_launch("conv_pack_weights");
}

extern "C" int upf_conv_forward(const void* x, long long x_batch_stride, const void* w_packed, const float* bias,
                                void* y, long long y_batch_stride, int B, int Cin, int Cout, int H, int W,
                                int kernel_size, int dilation, int stride, float leaky_slope, int dtype, void* stream) {
  using namespace upf;
  UPF_REQUIRE(x && w_packed && bias && y, UPF_EINVAL, "conv_forward: null pointer");
  UPF_REQUIRE(B > 0 && Cin > 0 && Cout > 0 && H > 0 && W > 0, UPF_EINVAL,
              "conv_forward: bad shape B=%d Cin=%d Cout=%d H=%d W=%d", B, Cin, Cout, H, W);
  UPF_REQUIRE(dtype == UPF_BF16 || dtype == UPF_F16, UPF_EDTYPE, "conv_forward: bf16 / fp16 only");
  UPF_REQUIRE(kernel_size == 3 || kernel_size == 1, UPF_EUNSUPPORTED, "conv_forward: kernel_size %d (1 or 3)", kernel_size);
  UPF_REQUIRE(dilation >= 1 && dilation <= conv::MAXD, UPF_EUNSUPPORTED, "conv_forward: dilation %d not in [1,%d]", dilation, conv::MAXD);
  UPF_REQUIRE(stride == 1 || (stride == 2 && dilation == 1 && kernel_size == 3), UPF_EUNSUPPORTED, "conv_forward: stride %d (1, or 2 for a 3x3 with dilation 1)", stride);
  const bool gen = !(W % 8 == 0 && aligned_to(x, 16) && x_batch_stride % 8 == 0);   // rows not 16-byte aligned
  UPF_REQUIRE(!gen || W >= 8, UPF_EUNSUPPORTED, "conv_forward: W = %d < 8 with unaligned rows", W);
  UPF_REQUIRE((long long)Cin * H * W * 2 < (1ll << 31), UPF_EINVAL, "conv_forward: image too large for one buffer descriptor");
  conv::Args a{x, x_batch_stride, w_packed, bias, y, y_batch_stride, B, Cin, Cout, H, W,
               kernel_size == 1 ? 0 : dilation, stride, kernel_size * kernel_size, leaky_slope, (hipStream_t)stream};
  const int mt = (Cout + 31) / 32;
  if (gen) return dtype == UPF_BF16 ? conv::launch_slabs<bf16_t, true>(a) : conv::launch_slabs<f16_t, true>(a);
  if (mt > 4) return dtype == UPF_BF16 ? conv::launch_slabs<bf16_t, false>(a) : conv::launch_slabs<f16_t, false>(a);
#define UPF_CONV_CASE(T)                          \
  switch (mt) {                                   \
    case 1: return conv::launch<T, 1>(a);         \
    case 2: return conv::launch<T, 2>(a);         \
    case 3: return conv::launch<T, 3>(a);         \
    default: return conv::launch<T, 4>(a);        \
  }
  if (dtype == UPF_BF16) { UPF_CONV_CASE(bf16_t) } else { UPF_CONV_CASE(f16_t) }
#undef UPF_CONV_CASE
}
