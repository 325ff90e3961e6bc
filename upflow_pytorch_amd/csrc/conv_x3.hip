// Split-precision convolution for the fp32 PARITY mode — gfx950 (round 4).
//
// The reference's estimator / context / pyramid convolutions are fp32 (/root/reference/model/pwc_modules.py:122-142, :250-286,
// :396-412; model/upflow.py:24-60), and the <= 1e-4 px end-point-error bar of BASELINE.json is only reachable with fp32-class
// products.  gfx950 has no reduced-precision fast path for fp32 MFMA inputs (no xf32; v_mfma_f32_32x32x2_f32 runs at the
// VALU rate, 1/16 of the 16-bit matrix rate), so until now the parity mode ran every convolution through PyTorch-ROCm (MIOpen).
// This kernel keeps fp32 tensors in HBM and multiplies on the fp16 matrix cores with BOTH operands split in two halves,
//     a = a_hi + a_lo,  a_hi = fp16(a),  a_lo = fp16(a - a_hi)          (22-23 significant bits together)
//     a * b ~= a_hi*b_hi + a_lo*b_hi + a_hi*b_lo   (+ a_lo*b_lo with NPROD = 4)      fp32 accumulation inside the MFMA,
// i.e. three (four) v_mfma_f32_32x32x16_f16 per 16-bit-path MFMA: relative error of a product <= ~2^-21 (2^-22 for the
// dropped a_lo*b_lo and each truncation), against 2^-24 for an fp32 product — the sums of a few thousand such products that a
// layer forms land within ~1e-6 relative of the fp32 convolution, measured per layer in tests/test_hip_conv_x3.py and as whole-net
// EPE vs the reference in tests/test_hip_net.py.  Range: fp16's — |x|, |w| < 65504 (inf beyond; frames are O(1), activations of
// this network O(10)); fp16 subnormal halves are multiplied un-flushed (default denormal mode, checked by upf_mfma_f16_denorm_probe).
//
// Structure (the NCHW kernel of conv_kernel.hpp, reduced to what the parity mode needs): a workgroup (4 waves) owns an
// 8 x 32 pixel tile of one image and MTW * 32 output channels; per chunk of 16 input channels the fp32 tile + halo is loaded
// (two 16-byte loads per 8 pixels of a channel row, or element-wise with bounds at ragged / unaligned edges: ANY H, W >= 1),
// split on the way in, transposed in registers (stage_store) into TWO LDS tile images (hi, lo) of 16-byte [8 channels x 1 pixel]
// entries; the taps read shifted windows of them.  Weights: packed once per layer as fp16 hi / lo operand blocks in MFMA lane
// order, kept in registers per chunk.  Dilation 1..16 by ROW PHASE (a tile's rows are d image rows apart: 2 halo rows whatever d
// is), stride 2, 1x1.  x / y are channel slices of contiguous NCHW fp32 buffers (the concat-free dense-stack buffers work in fp32
// too); the epilogue adds nothing (the accumulators start at the bias), applies LeakyReLU and stores fp32.
#include <cstring>
#include <cstdint>
#include "conv_kernel.hpp"

// Ablation switches for tools/x3_ablate.hip (always 0 in the library): 1 = no matrix phase, 2 = no split / transposition / LDS
// stores (the loaded values are only kept alive), 4 = no global loads of the tile, 8 = the matrix phase multiplies registers
// (no LDS reads).
#ifndef UPF_X3_ABL
#define UPF_X3_ABL 0
#endif

namespace upf {
namespace convx3 {
using namespace upf::conv;

constexpr int TH = 8;                    // tile rows
static int g_sk_max_tiles = 96;          // split-K kernel where the image has at most this many 8 x 32 tiles (upf_conv_x3_set_option)
constexpr int NOCT = 2;                  // channel octets per chunk (16 input channels)

__host__ __device__ constexpr int pad16(int v) { return (v + 15) / 16 * 16; }

// 8 consecutive fp32 pixels of one channel row -> 4 dwords of fp16 hi halves, 4 dwords of fp16 lo halves
__device__ __forceinline__ void split8(const float (&v)[8], u32x4& hi, u32x4& lo) {
#pragma unroll
  for (int pp = 0; pp < 4; ++pp) {
    const uint32_t h = pack2<f16_t>(v[2 * pp], v[2 * pp + 1]);
    const float r0 = v[2 * pp] - f16_bits_to_f32(h & 0xffffu), r1 = v[2 * pp + 1] - f16_bits_to_f32(h >> 16);
    hi[pp] = h;
    lo[pp] = pack2<f16_t>(r0, r1);
  }
}

// 4 consecutive fp32 pixels of one channel row -> 2 dwords of fp16 hi halves, 2 dwords of fp16 lo halves
__device__ __forceinline__ void split4(const f32x4& v, u32x2& hi, u32x2& lo) {
#pragma unroll
  for (int pp = 0; pp < 2; ++pp) {
    const uint32_t h = pack2<f16_t>(v[2 * pp], v[2 * pp + 1]);
    hi[pp] = h;
    lo[pp] = pack2<f16_t>(v[2 * pp] - f16_bits_to_f32(h & 0xffffu), v[2 * pp + 1] - f16_bits_to_f32(h >> 16));
  }
}
// 8 channel rows x 4 pixels (half of an 8-pixel group: pixels 4*half .. 4*half+3) -> 4 LDS entries of 8 channels x 1 pixel,
// at the rotated slots of stage_store (conv_kernel.hpp): entry of pixel q of the group = dst[(q + rot) & 7]
__device__ __forceinline__ void stage_store_half(uint4* tile, int enc, int half, const u32x2 (&ch)[8]) {
  uint4* dst = tile + (enc >> 3);
  const int rot = enc & 7;
#pragma unroll
  for (int pp = 0; pp < 2; ++pp) {
    uint4 e0, e1;
    e0.x = __builtin_amdgcn_perm(ch[1][pp], ch[0][pp], 0x05040100u); e1.x = __builtin_amdgcn_perm(ch[1][pp], ch[0][pp], 0x07060302u);
    e0.y = __builtin_amdgcn_perm(ch[3][pp], ch[2][pp], 0x05040100u); e1.y = __builtin_amdgcn_perm(ch[3][pp], ch[2][pp], 0x07060302u);
    e0.z = __builtin_amdgcn_perm(ch[5][pp], ch[4][pp], 0x05040100u); e1.z = __builtin_amdgcn_perm(ch[5][pp], ch[4][pp], 0x07060302u);
    e0.w = __builtin_amdgcn_perm(ch[7][pp], ch[6][pp], 0x05040100u); e1.w = __builtin_amdgcn_perm(ch[7][pp], ch[6][pp], 0x07060302u);
    dst[(4 * half + 2 * pp + rot) & 7] = e0;
    dst[(4 * half + 2 * pp + 1 + rot) & 7] = e1;
  }
}

// Packed operand = a 1 KB header {|w|max bits, scale 2^s, 2^-s, 0...} + the blocks
//   [slab = co/32][k-step = ci/16][tap][half: hi, lo][kg = (ci/8)%2][px = co%32][ci%8]   of fp16(w * 2^s) halves:
// the 1 KB block of one (slab, k-step, tap, half) is the A operand of one v_mfma_f32_32x32x16_f16 in lane order.
// The per-layer POWER-OF-TWO scale 2^s puts max|w| into [2^13, 2^14): MSRA-sized weights (~0.02) would otherwise have low halves
// of ~2^-17, deep inside fp16's subnormal range (spacing 2^-24): hi + lo then represents w to only ~2^-20 relative, and that —
// not the dropped lo*lo product, not the accumulation chain — was the whole error of the first form of this kernel (per-layer rms
// 8.7e-7 vs fp64, reproduced on the CPU with exact accumulation).  Scaled, the low halves are normal numbers with their own 11
// bits.  The kernel starts its accumulators at bias * 2^s and multiplies the result by 2^-s in the epilogue: both exact.
constexpr int HDR_F16 = 512;                          // header size in fp16 elements (1 KB)
__global__ void x3_hdr_zero_kernel(uint32_t* hdr) { if (threadIdx.x < 256) hdr[threadIdx.x] = 0u; }
__global__ void x3_absmax_kernel(const float* __restrict__ w, long long n, uint32_t* hdr) {
  float m = 0.f;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    const float a = fabsf(w[i]);
    m = (a == a && a > m) ? a : m;                    // (NaN weights do not define the scale)
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
  if ((threadIdx.x & 63) == 0) atomicMax(hdr, __float_as_uint(m));           // non-negative floats order like their bit patterns
}
__device__ __forceinline__ float x3_scale_of(uint32_t absmax_bits) {
  const float m = __uint_as_float(absmax_bits);
  if (!(m > 0.f) || m > 3.0e38f) return 1.f;
  int e = (int)((absmax_bits >> 23) & 0xffu) - 127;   // floor(log2(m)) for normal m (a subnormal maximum: e = -127, clamped below)
  int sft = 13 - e;
  sft = sft > 100 ? 100 : (sft < -100 ? -100 : sft);
  return __uint_as_float((uint32_t)(127 + sft) << 23);
}
__global__ void pack_x3_kernel(const float* __restrict__ w, f16_t* __restrict__ wp, int Cin, int Cout, int ntaps) {
  const int cip = pad16(Cin), cop = pad32(Cout), nk = cip / 16;
  const float sc = x3_scale_of(reinterpret_cast<const uint32_t*>(wp)[0]);
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    reinterpret_cast<float*>(wp)[1] = sc;
    reinterpret_cast<float*>(wp)[2] = 1.0f / sc;
  }
  f16_t* blocks = wp + HDR_F16;
  const long long total = (long long)ntaps * cop * cip * 2;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int j = (int)(i & 7), px = (int)((i >> 3) & 31), kg = (int)((i >> 8) & 1), hl = (int)((i >> 9) & 1);
    const long long b = i >> 10;                     // (slab * nk + kstep) * ntaps + tap
    const int tap = (int)(b % ntaps), kstep = (int)((b / ntaps) % nk), slab = (int)(b / ((long long)ntaps * nk));
    const int co = slab * 32 + px, ci = kstep * 16 + kg * 8 + j;
    float v = 0.f;
    if (ci < Cin && co < Cout) v = w[((size_t)co * Cin + ci) * ntaps + tap] * sc;
    const uint16_t h = f32_to_f16_bits(v);
    blocks[i].v = hl ? f32_to_f16_bits(v - f16_bits_to_f32(h)) : h;
  }
}

// MTW: 32-channel output blocks per workgroup (1 or 2);  S: stride;  MARG: staged halo columns (>= the dilation, whole 4-pixel
// groups);  K1: 1x1 kernel;  NPROD: products per (a, b) pair, 3 or 4.
// SPLITACC: the low-order products (a_lo*b_hi, a_hi*b_lo, a_lo*b_lo: 2^-11 of the result) accumulate in their OWN registers and
// join the main sum once, in the epilogue.  Every MFMA rounds its fp32 accumulator once; a 565-channel 3x3 layer chains
// 36 chunks x 9 taps x 3 products = 972 of them, and the random walk of those roundings (~sqrt(972) * 2^-25 ~ 1e-6 relative),
// not the operand split (~2^-23), is the kernel's error.  With the low-order terms out of the chain it is a third as long, and
// the roundings of the low-order chain are 2^-11 smaller.  Costs RPW * 16 more accumulator registers: the 64-channel
// workgroups then run one per CU (accumulators in AGPRs).
// (Measured and not kept, round 4: TWO chunks of staging data in flight per thread in the 32-channel workgroups — the layers with
//  Cout <= 32 did not move (531 -> 32 at 96x320: 286.5 -> 286.8 us).  tools/x3_ablate.hip says why: with the matrix phase removed the
//  same launch still takes 204 us — 1.04 GB of tile + halo at the ~5 TB/s the memory system delivers to this access pattern — and
//  with the loads removed 212 us; the two overlap to 287-350 us.  These layers are bandwidth bound like their bf16 twins (101 us
//  for half the bytes), not latency bound.)
template <int MTW, int S, int MARG, bool K1, int NPROD, bool SPLITACC>
__global__ __launch_bounds__(NTHREADS, (SPLITACC && MTW == 2) ? 1 : 2)
void conv_x3_kernel(const float* __restrict__ x, long long xbs, const f16_t* __restrict__ wp, const float* __restrict__ bias,
                    float* __restrict__ y, long long ybs, int Cin, int Cout, int H, int W, int Ho, int Wo, int d,
                    int tiles_x, int tiles_y, int nslabs, float slope, int aligned) {
  constexpr int ntaps = K1 ? 1 : 9;
  constexpr int RG = 4 / MTW, RPW = TH / RG;         // MTW = 2: two row groups of 4 rows;  MTW = 1: four of 2
  constexpr int XW = S * TW + 2 * MARG, XWP = XW + XW / 16;
  constexpr int ROWS = K1 ? TH : S * (TH - 1) + 3;   // staged rows (row phase: 2 halo rows whatever the dilation)
  constexpr int IMG = NOCT * ROWS * XWP;             // entries of one tile image
  extern __shared__ __attribute__((aligned(16))) uint4 xs[];
  uint4* xhi = xs;
  uint4* xlo = xs + IMG;

  const int RS = (S == 1 && !K1) ? d : 1;            // image rows between consecutive tile rows
  // the workgroups of ONE pixel tile (its Cout / (32 MTW) channel slabs) are consecutive block indices: they run at the same time on
  // the same XCD, so the tile is fetched into that XCD's L2 once (as gridDim.y they ran a whole grid apart and each fetched it
  // from beyond L2: 565 -> 128 moved 2 x 555 MB that way)
  const int bid0 = xcd_remap(blockIdx.x, gridDim.x);
  const int bid = bid0 / nslabs, slab_w = bid0 - bid * nslabs;
  const int tx = bid % tiles_x, ty = (bid / tiles_x) % tiles_y, n = bid / (tiles_x * tiles_y);
  const int x0 = tx * TW, y0 = (ty / RS) * (RS * TH) + ty % RS;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int px = lane & 31, kg = lane >> 5;
  const int cb = wave % MTW, rg = wave / MTW;
  const int slab = slab_w * MTW + cb;
  const int cip = pad16(Cin), nchunks = cip / 16;
  const int HW = H * W;
  const float* xn = x + (size_t)n * xbs;
  __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(xn), 0, (uint32_t)Cin * (uint32_t)HW * 4u, 0x00020000);
  __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc(const_cast<f16_t*>(wp + HDR_F16), 0, (uint32_t)ntaps * (uint32_t)pad32(Cout) * (uint32_t)cip * 4u, 0x00020000);
  const float wsc = reinterpret_cast<const float*>(wp)[1], winv = reinterpret_cast<const float*>(wp)[2];   // the layer's weight scale 2^s, 2^-s
  __amdgpu_buffer_rsrc_t br = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(bias), 0, (uint32_t)Cout * 4u, 0x00020000);

  f32x16 acc[RPW], accs[SPLITACC ? RPW : 1];
#pragma unroll
  for (int e = 0; e < 16; ++e) {
    const float bv = wsc * __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(br, (uint32_t)(slab * 32 + (e & 3) + 8 * (e >> 2) + 4 * kg) * 4u, 0, 0));
#pragma unroll
    for (int r = 0; r < RPW; ++r) acc[r][e] = bv;
#pragma unroll
    for (int r = 0; r < (SPLITACC ? RPW : 1); ++r) accs[r][e] = 0.f;
  }

  uint4 wh[ntaps], wl[ntaps];
  const int colx[3] = {swz(MARG + S * px - (K1 ? 0 : d)), swz(MARG + S * px), swz(MARG + S * px + (K1 ? 0 : d))};

  // ---- staging tasks: (octet, staged row, 4-pixel half of an 8-pixel group) -> 8 channel rows x 4 fp32 pixels (eight 16-byte
  // loads, or element-wise with bounds at ragged / unaligned places), split into fp16 hi / lo halves, transposed into 4 entries
  // of each tile image.  240 tasks per chunk at stride 1: one per thread, whose loads for chunk c + 1 are issued BEFORE the matrix
  // phase of chunk c (HBM latency under the MFMAs) and land in LDS after it.
  constexpr int nhalves = XW / 4, ntasks = NOCT * ROWS * nhalves;
  auto task_geom = [&](int t, long long& e0, int& enc, int& half, int& mode) {
    const int oct = t / (ROWS * nhalves), rem = t - oct * (ROWS * nhalves), r = rem / nhalves, hq = rem - r * nhalves;
    const int g = hq >> 1;
    half = hq & 1;
    const int gy = K1 ? y0 + r : (S == 1 ? y0 + (r - 1) * RS : 2 * y0 - 1 + r), gx = S * x0 - MARG + 4 * hq;
    const bool row_ok = gy >= 0 && gy < H;
    e0 = (long long)(oct * 8) * HW + (long long)gy * W + gx;              // element index of (channel 8*oct of chunk 0, gy, gx)
    enc = ((oct * ROWS + r) * XWP + 8 * g) * 8 + ((g >> 1) & 7);
    // mode 0: nothing inside the image (zeros);  1: four aligned pixels inside -> 16-byte loads;  2: element-wise with bounds
    mode = (t >= ntasks || !row_ok || gx + 4 <= 0 || gx >= W) ? 0 : ((aligned && gx >= 0 && gx + 4 <= W) ? 1 : 2);
    if (mode == 2) e0 = (long long)(oct * 8) * HW + (long long)gy * W;    // (row start: the columns are checked one by one)
  };
  auto task_load = [&](long long e0, int mode, int cc, int gx, f32x4 (&v)[8]) {
    const long long base = e0 + (long long)cc * 16 * HW;
    if constexpr ((UPF_X3_ABL & 4) != 0) {
#pragma unroll
      for (int k = 0; k < 8; ++k) v[k] = f32x4{slope, slope * (float)k, slope + (float)cc, slope};
      return;
    }
    if (mode == 1) {
#pragma unroll
      for (int k = 0; k < 8; ++k)                                          // (channels >= Cin fall off the descriptor: zeros)
        v[k] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xr, (uint32_t)((base + (long long)k * HW) * 4), 0, 0));
    } else {
#pragma unroll
      for (int k = 0; k < 8; ++k)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const bool ok = mode == 2 && gx + q >= 0 && gx + q < W;
          const uint32_t off = ok ? (uint32_t)((base + (long long)k * HW + gx + q) * 4) : 0x80000000u;
          v[k][q] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xr, off, 0, 0));
        }
    }
  };
  auto task_store = [&](int enc, int half, const f32x4 (&v)[8]) {
    if constexpr ((UPF_X3_ABL & 2) != 0) {
#pragma unroll
      for (int k = 0; k < 8; ++k) asm volatile("" ::"v"(v[k]));
      return;
    }
    u32x2 hi[8], lo[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) split4(v[k], hi[k], lo[k]);
    stage_store_half(xhi, enc, half, hi);
    stage_store_half(xlo, enc, half, lo);
  };
  // this thread's first task: geometry once, data prefetched a chunk ahead
  long long pe0; int penc, phalf, pmode;
  task_geom(tid, pe0, penc, phalf, pmode);
  const int pgx = S * x0 - MARG + 4 * ((tid % (ROWS * nhalves)) % nhalves);
  f32x4 pre[8];
  task_load(pe0, pmode, 0, pgx, pre);

  for (int cc = 0; cc < nchunks; ++cc) {
    // this chunk's weight operands (L2-resident; their latency hides under the staging below)
#pragma unroll
    for (int tap = 0; tap < ntaps; ++tap) {
      const uint32_t off = (uint32_t)(((slab * nchunks + cc) * ntaps + tap) * 2) * 1024u + (uint32_t)lane * 16u;
      wh[tap] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(wr, off, 0, 0));
      wl[tap] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(wr, off + 1024u, 0, 0));
    }
    __syncthreads();                                 // the previous chunk is fully consumed
    // ---- stage channels [16 cc, 16 cc + 16)
    if (tid < ntasks) task_store(penc, phalf, pre);
    for (int t = tid + NTHREADS; t < ntasks; t += NTHREADS) {
      long long e0; int enc, half, mode;
      task_geom(t, e0, enc, half, mode);
      f32x4 v[8];
      task_load(e0, mode, cc, S * x0 - MARG + 4 * ((t % (ROWS * nhalves)) % nhalves), v);
      task_store(enc, half, v);
    }
    __syncthreads();
    if (cc + 1 < nchunks) task_load(pe0, pmode, cc + 1, pgx, pre);

    // ---- matrix phase: lane (px, kg) reads entry (octet kg, row, column) of both images
    if constexpr ((UPF_X3_ABL & 1) != 0) continue;
    auto mm = [&](int tap, const uint4& bh_, const uint4& bl_, int r) {
      const uint4 bh = (UPF_X3_ABL & 8) ? wh[(tap + 1) % ntaps] : bh_, bl = (UPF_X3_ABL & 8) ? wl[(tap + 1) % ntaps] : bl_;
      f32x16& lo = SPLITACC ? accs[SPLITACC ? r : 0] : acc[r];
      lo = Mma32<f16_t>::mma(wl[tap], bh, lo);
      lo = Mma32<f16_t>::mma(wh[tap], bl, lo);
      if constexpr (NPROD == 4) lo = Mma32<f16_t>::mma(wl[tap], bl, lo);
      acc[r] = Mma32<f16_t>::mma(wh[tap], bh, acc[r]);
    };
    if constexpr (K1) {
#pragma unroll
      for (int r = 0; r < RPW; ++r) {
        const int e = (kg * ROWS + RPW * rg + r) * XWP + colx[1];
        mm(0, xhi[e], xlo[e], r);
      }
    } else if constexpr (S == 1) {
      // staged row sr of this wave's strip feeds output rows r = sr - ky: each window is read once and used by up to three taps
#pragma unroll
      for (int sr = 0; sr < RPW + 2; ++sr)
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
          const int e = (kg * ROWS + RPW * rg + sr) * XWP + colx[kx];
          const uint4 bh = xhi[e], bl = xlo[e];
#pragma unroll
          for (int ky = 0; ky < 3; ++ky)
            if (sr - ky >= 0 && sr - ky < RPW) mm(ky * 3 + kx, bh, bl, sr - ky);
        }
    } else {
#pragma unroll
      for (int r = 0; r < RPW; ++r)
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
          const int ky = tap / 3, kx = tap - 3 * ky;
          const int e = (kg * ROWS + 2 * (RPW * rg + r) + ky) * XWP + colx[kx];
          mm(tap, xhi[e], xlo[e], r);
        }
    }
  }

  // ---- epilogue: LeakyReLU, fp32 stores (lane = pixel column; register e = channel (e&3) + 8*(e>>2) + 4*kg of the block)
  const uint32_t plane = (uint32_t)(Ho * Wo) * 4u;
  __amdgpu_buffer_rsrc_t yr = __builtin_amdgcn_make_buffer_rsrc(y + (size_t)n * ybs, 0, (uint32_t)Cout * plane, 0x00020000);
  const int gx = x0 + px;
#pragma unroll
  for (int r = 0; r < RPW; ++r) {
    const int gy = y0 + (RPW * rg + r) * RS;         // uniform
    if (gy < Ho) {
      const uint32_t base = (gx < Wo) ? (uint32_t)(gy * Wo + gx) * 4u : 0x80000000u;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        float v = acc[r][e];
        if constexpr (SPLITACC) v += accs[r][e];
        v *= winv;
        v = fmaxf(v, v * slope);
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(uint32_t, v), yr, base + (uint32_t)(slab * 32 + (e & 3) + 8 * (e >> 2) + 4 * kg) * plane, 0, 0);
      }
    }
  }
}


// Split-K variant for the COARSE pyramid levels and other small grids (6x20 .. 24x80 pixels: the kernel above is then a serial
// chain of Cin/16 chunks of ~2.5 us on a few dozen workgroups — 531 -> 32 at 6x20 took 84 us for 3 us of matrix work).  As in
// conv3x3.hip's conv_sk_kernel: a workgroup owns a 2 x 32 pixel tile (rows d apart: row phase) and 32 output channels, its four
// waves take the 16-channel chunks w, w + 4, ..., each staging into its OWN pair of LDS images (no workgroup barrier inside the
// loop; the LDS operations of one wave execute in order), both staging tasks of a lane prefetched a chunk ahead, the weight
// operands of a kernel row reloaded for the wave's next chunk right after their last use; the four partial tiles are summed
// through LDS in wave order (deterministic) and leave through the same scale / LeakyReLU epilogue.
template <int MARG, bool K1>
__global__ __launch_bounds__(NTHREADS, 2)
void conv_x3_sk_kernel(const float* __restrict__ x, long long xbs, const f16_t* __restrict__ wp, const float* __restrict__ bias,
                       float* __restrict__ y, long long ybs, int Cin, int Cout, int H, int W, int d,
                       int tiles_x, int tiles_y, int nslabs, float slope, int aligned) {
  constexpr int ntaps = K1 ? 1 : 9;
  constexpr int RPW = 2;
  constexpr int XW = TW + 2 * MARG, XWP = XW + XW / 16;
  constexpr int ROWS = K1 ? RPW : RPW + 2;
  constexpr int IMG = NOCT * ROWS * XWP;             // entries of one image of one wave
  constexpr int nhalves = XW / 4, ntasks = NOCT * ROWS * nhalves, NT = (ntasks + 63) / 64;
  extern __shared__ __attribute__((aligned(16))) uint4 xs[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  uint4* xhi = xs + wave * 2 * IMG;
  uint4* xlo = xhi + IMG;

  const int RS = K1 ? 1 : d;
  const int bid0 = xcd_remap(blockIdx.x, gridDim.x);                  // (slabs of one tile = consecutive blocks, as above)
  const int bid = bid0 / nslabs, slab = bid0 - bid * nslabs;
  const int tx = bid % tiles_x, ty = (bid / tiles_x) % tiles_y, n = bid / (tiles_x * tiles_y);
  const int x0 = tx * TW, y0 = (ty / RS) * (RS * RPW) + ty % RS;
  const int px = lane & 31, kg = lane >> 5;
  const int cip = pad16(Cin), nchunks = cip / 16;
  const int HW = H * W;
  __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(x + (size_t)n * xbs), 0, (uint32_t)Cin * (uint32_t)HW * 4u, 0x00020000);
  __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc(const_cast<f16_t*>(wp + HDR_F16), 0, (uint32_t)ntaps * (uint32_t)pad32(Cout) * (uint32_t)cip * 4u, 0x00020000);
  const float wsc = reinterpret_cast<const float*>(wp)[1], winv = reinterpret_cast<const float*>(wp)[2];
  // wave 0's partial tile starts at bias * 2^s (the other waves, and channels >= Cout, read 0 through the descriptor)
  __amdgpu_buffer_rsrc_t br = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(bias), 0, wave == 0 ? (uint32_t)Cout * 4u : 0u, 0x00020000);
  f32x16 acc[RPW];
#pragma unroll
  for (int e = 0; e < 16; ++e) {
    const float bv = wsc * __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(br, (uint32_t)(slab * 32 + (e & 3) + 8 * (e >> 2) + 4 * kg) * 4u, 0, 0));
#pragma unroll
    for (int r = 0; r < RPW; ++r) acc[r][e] = bv;
  }

  // this lane's staging tasks (octet, staged row, 4-pixel half-group), the same for every chunk
  long long te0[NT]; int tenc[NT], thalf[NT], tmode[NT], tgx[NT];
#pragma unroll
  for (int i = 0; i < NT; ++i) {
    const int t = lane + 64 * i;
    const int oct = t / (ROWS * nhalves), rem = t - oct * (ROWS * nhalves), r = rem / nhalves, hq = rem - r * nhalves;
    const int g = hq >> 1;
    const int gy = K1 ? y0 + r * RS : y0 + (r - 1) * RS, gx = x0 - MARG + 4 * hq;
    thalf[i] = hq & 1;
    tgx[i] = gx;
    tenc[i] = ((oct * ROWS + r) * XWP + 8 * g) * 8 + ((g >> 1) & 7);
    tmode[i] = (t >= ntasks || gy < 0 || gy >= H || gx + 4 <= 0 || gx >= W) ? 0 : ((aligned && gx >= 0 && gx + 4 <= W) ? 1 : 2);
    te0[i] = (long long)(oct * 8) * HW + (long long)gy * W + (tmode[i] == 1 ? gx : 0);
  }
  auto task_load = [&](int i, int cc, f32x4 (&v)[8]) {
    const long long base = te0[i] + (long long)cc * 16 * HW;
    if (tmode[i] == 1) {
#pragma unroll
      for (int k = 0; k < 8; ++k)
        v[k] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xr, (uint32_t)((base + (long long)k * HW) * 4), 0, 0));
    } else {
#pragma unroll
      for (int k = 0; k < 8; ++k)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const bool ok = tmode[i] == 2 && tgx[i] + q >= 0 && tgx[i] + q < W;
          const uint32_t off = ok ? (uint32_t)((base + (long long)k * HW + tgx[i] + q) * 4) : 0x80000000u;
          v[k][q] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xr, off, 0, 0));
        }
    }
  };
  auto task_store = [&](int i, const f32x4 (&v)[8]) {
    u32x2 hi[8], lo[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) split4(v[k], hi[k], lo[k]);
    stage_store_half(xhi, tenc[i], thalf[i], hi);
    stage_store_half(xlo, tenc[i], thalf[i], lo);
  };
  f32x4 pre[NT][8];
#pragma unroll
  for (int i = 0; i < NT; ++i)
    if (wave < nchunks) task_load(i, wave, pre[i]);

  uint4 wh[ntaps], wl[ntaps];
  auto wload = [&](int cc, int tap) {
    const uint32_t off = (cc < nchunks) ? (uint32_t)(((slab * nchunks + cc) * ntaps + tap) * 2) * 1024u + (uint32_t)lane * 16u : 0x80000000u;
    wh[tap] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(wr, off, 0, 0));
    wl[tap] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(wr, off + 1024u, 0, 0));
  };
#pragma unroll
  for (int tap = 0; tap < ntaps; ++tap) wload(wave, tap);
  const int colx[3] = {swz(MARG + px - (K1 ? 0 : d)), swz(MARG + px), swz(MARG + px + (K1 ? 0 : d))};

  for (int cc = wave; cc < nchunks; cc += 4) {
#pragma unroll
    for (int i = 0; i < NT; ++i)
      if (lane + 64 * i < ntasks) task_store(i, pre[i]);
#pragma unroll
    for (int i = 0; i < NT; ++i)
      if (cc + 4 < nchunks) task_load(i, cc + 4, pre[i]);
    auto mm = [&](int tap, const uint4& bh, const uint4& bl, int r) {
      acc[r] = Mma32<f16_t>::mma(wl[tap], bh, acc[r]);
      acc[r] = Mma32<f16_t>::mma(wh[tap], bl, acc[r]);
      acc[r] = Mma32<f16_t>::mma(wh[tap], bh, acc[r]);
    };
    if constexpr (K1) {
#pragma unroll
      for (int r = 0; r < RPW; ++r) {
        const int e = (kg * ROWS + r) * XWP + colx[1];
        mm(0, xhi[e], xlo[e], r);
      }
      wload(cc + 4, 0);
    } else {
#pragma unroll
      for (int sr = 0; sr < RPW + 2; ++sr) {
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
          const int e = (kg * ROWS + sr) * XWP + colx[kx];
          const uint4 bh = xhi[e], bl = xlo[e];
#pragma unroll
          for (int ky = 0; ky < 3; ++ky)
            if (sr - ky >= 0 && sr - ky < RPW) mm(ky * 3 + kx, bh, bl, sr - ky);
        }
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
          if (sr == ky + RPW - 1) {
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) wload(cc + 4, ky * 3 + kx);
          }
      }
    }
  }

  // ---- sum the four partial tiles in wave order: part[((w * 2 + r) * 16 + e) * 64 + lane]
  __syncthreads();
  float* part = reinterpret_cast<float*>(xs);
#pragma unroll
  for (int r = 0; r < RPW; ++r)
#pragma unroll
    for (int e = 0; e < 16; ++e) part[((wave * 2 + r) * 16 + e) * 64 + lane] = acc[r][e];
  __syncthreads();
  // wave w finishes output row r = w / 2, accumulator registers [8 * (w % 2), 8 * (w % 2) + 8)
  const int r = wave >> 1, ebase = 8 * (wave & 1);
  const int Ho = H, Wo = W;
  const uint32_t plane = (uint32_t)(Ho * Wo) * 4u;
  __amdgpu_buffer_rsrc_t yr = __builtin_amdgcn_make_buffer_rsrc(y + (size_t)n * ybs, 0, (uint32_t)Cout * plane, 0x00020000);
  const int gx = x0 + px, gy = y0 + r * RS;
  if (gy < Ho) {
    const uint32_t base = (gx < Wo) ? (uint32_t)(gy * Wo + gx) * 4u : 0x80000000u;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int e = ebase + j;
      float v = 0.f;
#pragma unroll
      for (int w = 0; w < 4; ++w) v += part[((w * 2 + r) * 16 + e) * 64 + lane];
      v *= winv;
      v = fmaxf(v, v * slope);
      __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(uint32_t, v), yr, base + (uint32_t)(slab * 32 + (e & 3) + 8 * (e >> 2) + 4 * kg) * plane, 0, 0);
    }
  }
}

struct Args {
  const float* x; long long xbs; const void* wp; const float* bias; float* y; long long ybs;
  int B, Cin, Cout, H, W, d, stride, k; float slope; int nprod; hipStream_t stream;
};

template <int MTW, int S, int MARG, bool K1, int NPROD, bool SPLITACC>
int launch_one(const Args& a) {
  const int Ho = (a.H - 1) / S + 1, Wo = (a.W - 1) / S + 1;
  const int rs = (S == 1 && !K1) ? a.d : 1;
  const int tiles_x = cdiv(Wo, TW), tiles_y = cdiv(Ho, rs * TH) * rs;
  constexpr int XW = S * TW + 2 * MARG, XWP = XW + XW / 16;
  constexpr int ROWS = K1 ? TH : S * (TH - 1) + 3;
  const size_t lds = (size_t)2 * NOCT * ROWS * XWP * 16;
  const bool aligned = a.W % 4 == 0 && aligned_to(a.x, 16) && a.xbs % 4 == 0;
  static LdsOptIn opt;
  auto kern = &conv_x3_kernel<MTW, S, MARG, K1, NPROD, SPLITACC>;
  opt.ensure(reinterpret_cast<const void*>(kern), lds);
  const int slabs = cdiv(cdiv(a.Cout, 32), MTW);
  hipLaunchKernelGGL(kern, dim3((unsigned)(a.B * tiles_x * tiles_y * slabs)), dim3(NTHREADS), lds, a.stream, a.x, a.xbs, (const f16_t*)a.wp, a.bias,
                     a.y, a.ybs, a.Cin, a.Cout, a.H, a.W, Ho, Wo, K1 ? 1 : a.d, tiles_x, tiles_y, slabs, a.slope, (int)aligned);
  return check_launch("conv_x3_forward");
}

template <int MARG, bool K1>
int launch_sk_one(const Args& a) {
  const int rs = K1 ? 1 : a.d;
  const int tiles_x = cdiv(a.W, TW), tiles_y = cdiv(a.H, rs * 2) * rs;
  constexpr int XW = TW + 2 * MARG, XWP = XW + XW / 16;
  constexpr int ROWS = K1 ? 2 : 4;
  size_t lds = (size_t)4 * 2 * NOCT * ROWS * XWP * 16;
  if (lds < 4 * 2 * 16 * 64 * 4) lds = 4 * 2 * 16 * 64 * 4;      // (the partial-sum exchange reuses the region)
  const bool aligned = a.W % 4 == 0 && aligned_to(a.x, 16) && a.xbs % 4 == 0;
  static LdsOptIn opt;
  auto kern = &conv_x3_sk_kernel<MARG, K1>;
  opt.ensure(reinterpret_cast<const void*>(kern), lds);
  const int slabs = cdiv(a.Cout, 32);
  hipLaunchKernelGGL(kern, dim3((unsigned)(a.B * tiles_x * tiles_y * slabs)), dim3(NTHREADS), lds, a.stream, a.x, a.xbs, (const f16_t*)a.wp,
                     a.bias, a.y, a.ybs, a.Cin, a.Cout, a.H, a.W, K1 ? 1 : a.d, tiles_x, tiles_y, slabs, a.slope, (int)aligned);
  return check_launch("conv_x3_forward");
}
int launch_sk(const Args& a) {
  if (a.k == 1) return launch_sk_one<0, true>(a);
  if (a.d <= 4) return launch_sk_one<4, false>(a);
  if (a.d <= 8) return launch_sk_one<8, false>(a);
  return launch_sk_one<16, false>(a);
}

template <int MTW, int NPROD, bool SPLITACC>
int launch_shape(const Args& a) {
  if (a.k == 1) return launch_one<MTW, 1, 0, true, NPROD, SPLITACC>(a);
  if (a.stride == 2) return launch_one<MTW, 2, 8, false, NPROD, SPLITACC>(a);
  if (a.d <= 4) return launch_one<MTW, 1, 4, false, NPROD, SPLITACC>(a);      // (staging works in 4-pixel units: a 4-column halo is enough)
  if (a.d <= 8) return launch_one<MTW, 1, 8, false, NPROD, SPLITACC>(a);
  return launch_one<MTW, 1, 16, false, NPROD, SPLITACC>(a);
}
template <int MTW>
int launch_mode(const Args& a) {
  // (NPROD = 4, the a_lo*b_lo product, was measured and changes nothing: per layer 2.41e-6 -> 2.41e-6 max, whole net 7.1e-5 ->
  // 6.9e-5 px for 8 % more time — profiles/r04_fp32_conv_modes.txt; not instantiated)
  return a.nprod == 3 ? launch_shape<MTW, 3, false>(a) : launch_shape<MTW, 3, true>(a);
}

// probe: does v_mfma_f32_32x32x16_f16 multiply fp16 SUBNORMAL inputs un-flushed?  A = subnormal 2^-20 in every (row, k),
// B = 2^10 in every (k, col): every output = 16 * 2^-20 * 2^10 = 2^-6 if the inputs are kept, 0 if they are flushed.
__global__ void mfma_f16_denorm_probe_kernel(float* out) {
  const uint32_t sub = 0x00100010u;                  // two fp16 subnormals 2^-20 (bit 4 of the mantissa: 2^-24 * 16)
  const uint32_t big = 0x64006400u;                  // two fp16 1024.0
  uint4 a = {sub, sub, sub, sub}, b = {big, big, big, big};
  f32x16 c;
#pragma unroll
  for (int e = 0; e < 16; ++e) c[e] = 0.f;
  c = Mma32<f16_t>::mma(a, b, c);
  if (threadIdx.x == 0) out[0] = c[0];
}

}  // namespace convx3
}  // namespace upf

extern "C" long long upf_conv_x3_packed_bytes(int Cin, int Cout, int kernel_size) {
  return (long long)upf::convx3::HDR_F16 * 2 + (long long)kernel_size * kernel_size * upf::conv::pad32(Cout) * upf::convx3::pad16(Cin) * 2 * 2;
}

extern "C" int upf_conv_x3_pack_weights(const float* w, void* w_packed, int Cin, int Cout, int kernel_size, void* stream) {
  using namespace upf;
  UPF_REQUIRE(w && w_packed && Cin > 0 && Cout > 0, UPF_EINVAL, "conv_x3_pack_weights: bad arguments");
  UPF_REQUIRE(kernel_size == 3 || kernel_size == 1, UPF_EUNSUPPORTED, "conv_x3_pack_weights: kernel_size %d (1 or 3)", kernel_size);
  const int ntaps = kernel_size * kernel_size;
  const long long total = (long long)ntaps * conv::pad32(Cout) * convx3::pad16(Cin) * 2;
  const int blocks = (int)((total + 255) / 256 > 4096 ? 4096 : (total + 255) / 256);
  const long long nw = (long long)Cout * Cin * ntaps;
  hipLaunchKernelGGL(convx3::x3_hdr_zero_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, (uint32_t*)w_packed);
  hipLaunchKernelGGL(convx3::x3_absmax_kernel, dim3((unsigned)((nw + 255) / 256 > 1024 ? 1024 : (nw + 255) / 256)), dim3(256), 0, (hipStream_t)stream, w, nw, (uint32_t*)w_packed);
  hipLaunchKernelGGL(convx3::pack_x3_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, w, (f16_t*)w_packed, Cin, Cout, ntaps);
  return check_launch("conv_x3_pack_weights");
}

extern "C" int upf_conv_x3_forward(const float* x, long long x_batch_stride, const void* w_packed, const float* bias, float* y,
                                   long long y_batch_stride, int B, int Cin, int Cout, int H, int W, int kernel_size, int dilation,
                                   int stride, float leaky_slope, int nprod, void* stream) {
  using namespace upf;
  UPF_REQUIRE(x && w_packed && bias && y, UPF_EINVAL, "conv_x3_forward: null pointer");
  UPF_REQUIRE(B > 0 && Cin > 0 && Cout > 0 && H > 0 && W > 0, UPF_EINVAL, "conv_x3_forward: bad shape B=%d Cin=%d Cout=%d H=%d W=%d", B, Cin, Cout, H, W);
  UPF_REQUIRE(kernel_size == 3 || kernel_size == 1, UPF_EUNSUPPORTED, "conv_x3_forward: kernel_size %d (1 or 3)", kernel_size);
  UPF_REQUIRE(dilation >= 1 && dilation <= conv::MAXD, UPF_EUNSUPPORTED, "conv_x3_forward: dilation %d not in [1,%d]", dilation, conv::MAXD);
  UPF_REQUIRE(stride == 1 || (stride == 2 && dilation == 1 && kernel_size == 3), UPF_EUNSUPPORTED, "conv_x3_forward: stride %d (1, or 2 for a 3x3 with dilation 1)", stride);
  UPF_REQUIRE(nprod == 3 || nprod == 11, UPF_EINVAL,
              "conv_x3_forward: nprod %d (3 products per operand pair; 11 = 3 + 8: the low-order products in their own accumulators)", nprod);
  UPF_REQUIRE(leaky_slope >= 0.f && leaky_slope <= 1.f, UPF_EINVAL, "conv_x3_forward: leaky_slope %g not in [0,1]", (double)leaky_slope);
  const int Ho = (H - 1) / stride + 1, Wo = (W - 1) / stride + 1;
  UPF_REQUIRE((long long)Cin * H * W * 4 < (1ll << 31) && (long long)Cout * Ho * Wo * 4 < (1ll << 31) &&
                  (long long)upf::convx3::pad16(Cin) * H * W * 4 < (1ll << 32), UPF_EINVAL,      // (the padding channels' offsets must not wrap)
              "conv_x3_forward: image too large for one buffer descriptor");
  convx3::Args a{x, x_batch_stride, w_packed, bias, y, y_batch_stride, B, Cin, Cout, H, W, dilation, stride, kernel_size,
                 leaky_slope == 0.f ? 1.f : leaky_slope, nprod, (hipStream_t)stream};
  // 64-channel workgroups where there are enough tiles to fill the chip with them; 32-channel ones for narrow layers and small grids
  const long long tiles = (long long)B * cdiv(Wo, conv::TW) * cdiv(Ho, convx3::TH);
  // small grids (the coarse pyramid levels): the input channels split across the four waves of 2-row tiles
  // (measured per layer, tools/conv_layers.py --fp32 --sweep: faster at 6x20 .. 24x80 x 8 items whatever Cout, slower from 48x160 on —
  // except the dilation-16 layer there, whose 8-row row-phase tiles waste 5/8 of a 48-row image)
  if (nprod == 3 && stride == 1 && convx3::pad16(Cin) / 16 >= 4 &&
      (tiles <= convx3::g_sk_max_tiles || (dilation == 16 && kernel_size == 3 && H < 64 && tiles <= 5ll * convx3::g_sk_max_tiles)))
    return convx3::launch_sk(a);
  const bool two = Cout > 32 && tiles * cdiv(cdiv(Cout, 32), 2) >= 256;
  return two ? convx3::launch_mode<2>(a) : convx3::launch_mode<1>(a);
}

extern "C" int upf_conv_x3_set_option(const char* name, int value) {
  int* slot = nullptr;
  if (name && !strcmp(name, "sk_max_tiles")) slot = &upf::convx3::g_sk_max_tiles;
  if (!slot) return INT32_MIN;
  const int prev = *slot;
  *slot = value;
  return prev;
}

extern "C" int upf_mfma_f16_denorm_probe(float* out_device, void* stream) {
  using namespace upf;
  UPF_REQUIRE(out_device, UPF_EINVAL, "mfma_f16_denorm_probe: null pointer");
  hipLaunchKernelGGL(convx3::mfma_f16_denorm_probe_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, out_device);
  return check_launch("mfma_f16_denorm_probe");
}
