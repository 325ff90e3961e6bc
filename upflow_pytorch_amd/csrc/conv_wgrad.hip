// Weight gradient of the 3x3 (dilated) / 1x1 stride-1 convolutions on the bf16 / fp16 matrix cores — gfx950.
//
// Training (BASELINE config 3) spends 60 % of its step in MIOpen's fp32 convolution kernels, a third of that in the
// weight-gradient kernels and their NCHW<->NHWC transposes (profiles/r02_train_fp32_step_kernels.txt).  The data gradient
// of a stride-1 convolution IS a convolution (flipped, transposed weights), so it runs on the forward kernel of
// conv3x3.hip; the weight gradient is the one new contraction:
//
//     dW[co][ci][ky][kx] = sum_{n,y,x} g[n,co,y,x] * X[n,ci, y+(ky-1)d, x+(kx-1)d]          (X zero padded)
//
// a GEMM per tap with M = co, N = ci and K = PIXELS.  v_mfma_f32_32x32x16: A = 32 co x 16 pixels, B = 16 pixels x 32 ci;
// both operands want 8 consecutive K-elements per lane — 8 consecutive pixels of one channel row, which is exactly how
// NCHW stores them: tiles are staged into LDS with plain 16-byte copies (no register transposition, unlike the forward
// kernel whose K is the channel dimension).
//
//   * workgroup = 4 waves = a 64 co x 64 ci block of dW (wave w: co block w&1, ci block w>>1), all taps: 9 x 16
//     accumulator registers per lane; grid = (K-split, block pairs).  A workgroup walks its share of the pixel tiles and
//     writes ONE partial block at the end; partials are summed in fixed order by a second kernel (deterministic).
//   * pixel tile = TR rows x 32 pixels.  Dilation by ROW PHASE: the TR rows of a tile are d image rows apart, so the
//     three kernel rows read TR + 2 staged rows of X whatever d is; horizontally the taps are windows shifted by -d, 0,
//     +d pixels inside the staged rows: whole 8-pixel blocks for d = 8, 16, dword selects for d = 2, 4, and a 2-byte
//     funnel shift (v_alignbyte) for d = 1 — computed once per staged row and k-step, shared by the three kernel rows.
//   * per staged row and 16-pixel k-step: 3-5 ds_read_b128 of X + 1 of g feed up to 9 MFMAs (each g row operand is kept
//     in registers for the three kernel rows that use it).
// Requirements: stride 1, W >= 8 (rows that are not 16-byte aligned — the coarse levels of the 256x832 training crops are
// 52, 26 and 13 pixels wide — take the RAGGED staging variant).
#include "common.hpp"
#include <cstring>
#include <cstdio>
#include <stdlib.h>

namespace upf {
namespace wgrad {

constexpr int NTHREADS = 256, TR = 4, TWP = 32;       // tile: 4 rows x 32 pixels
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

// halo (pixels) on each side of the staged X rows: whole 8-pixel blocks
__host__ __device__ constexpr int halo_of(int D) { return D == 16 ? 16 : (D == 0 ? 0 : 8); }
// LDS geometry (16-byte blocks of 8 pixels).  Channel pitches are ODD numbers of blocks so that the 16 lanes of a
// ds_read_b128 phase (consecutive channels, same row / column) fall on 16 different 16-byte bank groups.
template <int D> struct Geo {
  static constexpr int NT = (D == 0) ? 1 : 9;
  static constexpr int XR = (D == 0) ? TR : TR + 2;                     // staged X rows
  static constexpr int XB = (TWP + 2 * halo_of(D)) / 8;                  // blocks per staged X row
  static constexpr int XCH = (XR * XB) | 1;                              // blocks per X channel (odd)
  static constexpr int GB = TWP / 8;                                     // blocks per g row (4)
  static constexpr int GCH = (TR * GB) | 1;                              // blocks per g channel (17)
  static constexpr int X_BLOCKS = 64 * XCH, G_BLOCKS = 64 * GCH;
  static constexpr int LDS_BYTES = (X_BLOCKS + G_BLOCKS) * 16;
};

// window of 8 pixels starting SH pixels before (SH > 0) / after (SH < 0) the start of block `c`, from the neighbouring
// aligned blocks: p2, p1 = the two blocks before c, n1, n2 = the two after
template <int SH>
__device__ __forceinline__ uint4 window(const uint4& p2, const uint4& p1, const uint4& c, const uint4& n1, const uint4& n2) {
  if constexpr (SH == 0) return c;
  else if constexpr (SH == 16) return p2;
  else if constexpr (SH == -16) return n2;
  else if constexpr (SH == 8) return p1;
  else if constexpr (SH == -8) return n1;
  else if constexpr (SH == 4) return make_uint4(p1.z, p1.w, c.x, c.y);
  else if constexpr (SH == -4) return make_uint4(c.z, c.w, n1.x, n1.y);
  else if constexpr (SH == 2) return make_uint4(p1.w, c.x, c.y, c.z);
  else if constexpr (SH == -2) return make_uint4(c.y, c.z, c.w, n1.x);
  else if constexpr (SH == 1)
    return make_uint4(__builtin_amdgcn_alignbyte(c.x, p1.w, 2), __builtin_amdgcn_alignbyte(c.y, c.x, 2),
                      __builtin_amdgcn_alignbyte(c.z, c.y, 2), __builtin_amdgcn_alignbyte(c.w, c.z, 2));
  else  // SH == -1
    return make_uint4(__builtin_amdgcn_alignbyte(c.y, c.x, 2), __builtin_amdgcn_alignbyte(c.z, c.y, 2),
                      __builtin_amdgcn_alignbyte(c.w, c.z, 2), __builtin_amdgcn_alignbyte(n1.x, c.w, 2));
}

// partial: [ksplit][tap][pad64(Cout)][pad64(Cin)] fp32
// One workgroup per CU (512 registers per wave): 144 accumulators + the NEXT tile's staging data in registers — the global
// loads of tile t+1 are issued before the 72 MFMAs of tile t and written to LDS after them, so the matrix pipe only waits
// for two barriers and 13 ds_write_b128 per tile (the first version loaded, waited, computed: ~25 % MFMA utilisation).
// RAGGED (any W >= 8, any alignment): the 8-pixel block that straddles the row end is loaded shifted left so that it
// ENDS at the row end (2-byte-aligned 16-byte buffer loads are legal on gfx950) and shifted back in registers, zeros
// filling the pixels beyond the row.
// a wave-uniform pointer, in scalar registers whatever the compiler's divergence analysis concluded
template <typename P> __device__ __forceinline__ P* uniform_ptr(P* p) {
  const unsigned long long v = (unsigned long long)p;
  const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v), hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
  return (P*)(((unsigned long long)hi << 32) | lo);
}
__device__ __forceinline__ u32x4 shr_pixels(u32x4 v, int sh) {            // 128-bit logical shift right by sh pixels (16 bits each)
  const unsigned long long lo = ((unsigned long long)v[1] << 32) | v[0], hi = ((unsigned long long)v[3] << 32) | v[2];
  unsigned long long rl, rh;
  if (sh >= 4) { rl = (sh >= 8) ? 0ull : (hi >> (16 * (sh - 4))); rh = 0ull; }
  else if (sh == 0) { rl = lo; rh = hi; }
  else { rl = (lo >> (16 * sh)) | (hi << (64 - 16 * sh)); rh = hi >> (16 * sh); }
  u32x4 r;
  r[0] = (uint32_t)rl; r[1] = (uint32_t)(rl >> 32); r[2] = (uint32_t)rh; r[3] = (uint32_t)(rh >> 32);
  return r;
}

// The K dimension of one launch may span several PYRAMID LEVELS: the decoder's weights are shared by all levels
// (model/upflow.py:535-573 calls the same flow_estimators / context_networks five times), so their weight gradient is one
// contraction over the pixels of every level.  A launch takes up to MAXL (x, g) pairs of different sizes; tile indices run
// through the levels back to back (tile0 = first tile of a level) and a workgroup's tiles ks_id, ks_id + ksplit, ... are
// spread over all of them.
constexpr int MAXL = 6, LT_STRIDE = 16;                  // (LT_STRIDE: ints per row of wgrad_pc_kernel's level table in LDS)
struct KLevel {
  const void* x; const void* g;
  long long xbs, gbs;
  int H, W, tiles_x, tiles_y, tile0, ragged;     // ragged: rows / bases not 16-byte aligned (wgrad_pc_kernel<.., RAGGED = true> only)
};
struct KLevels {
  KLevel lv[MAXL];
  int n, ntiles;
};

// MB = 1: 4 waves, 64 co x 64 ci per workgroup;  MB = 2: 8 waves (two per SIMD, 256 registers each), 128 co x 64 ci.  The
// kernel is bound by its staging traffic (~53 KB per tile and workgroup at MB = 1, the X tile re-fetched by every co
// block): with 128 co per workgroup the X tile — the larger operand, with its halo — is fetched ONCE for Cout <= 128.
// (288 accumulators per lane in 4 waves were tried first: the allocator spills 100-200 registers per lane.)
template <typename T, int D, bool RAGGED, int MB>
__global__ __launch_bounds__(NTHREADS * MB, MB)
void wgrad_kernel(const KLevels L, float* __restrict__ partial, int Cin, int Cout, int nci2, int J) {
  using G = Geo<D>;
  constexpr int COB = 64 * MB;                                             // output channels per workgroup
  constexpr int DD = (D == 0) ? 1 : D;
  constexpr int HALO = halo_of(D);
  constexpr int NTH = NTHREADS * MB;
  constexpr int NXT = (64 * G::XR * G::XB + NTH - 1) / NTH;                // X staging tasks per thread (9; 12 for D = 16; half with 8 waves)
  constexpr int NGT = (COB * TR * G::GB + NTH - 1) / NTH;                  // g staging tasks per thread (4)
  extern __shared__ __attribute__((aligned(16))) uint4 smem[];
  uint4* xs = smem;
  uint4* gs = smem + G::X_BLOCKS;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // XCD-aware decomposition.  Workgroup L runs on XCD L % 8 (round-robin dispatch).  Each XCD owns a contiguous EIGHTH of
  // the tile range and runs, side by side, every (co block, ci block) pair of J consecutive tiles: the co blocks of a
  // tile share its X slab, the ci blocks its g slab, horizontally adjacent tiles their halo sectors — all through that
  // XCD's L2, so the fabric carries every byte about once.  (The first mapping — K-split index = blockIdx.x — spread the
  // sharers over all XCDs: every layer ran at the ~3.7 TB/s this access pattern gets from HBM / MALL, 3.8 us per tile
  // whatever its shape, a quarter of the MFMA rate.)  K-split slot (partial block) = xcd * J + j.
  const int nblocks = gridDim.x / (8 * J);
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int bpair = slot % nblocks, jj = slot / nblocks;
  const int ks_id = xcd * J + jj;
  const int co2 = bpair / nci2, ci2 = bpair - co2 * nci2;                // (COB-channel co block, 64-channel ci block) of this workgroup
  const int cob = wave % (2 * MB), cib = wave / (2 * MB);
  const int ch = lane & 31, kg = lane >> 5;
  const int t_lo = (int)((long long)L.ntiles * xcd / 8), t_hi = (int)((long long)L.ntiles * (xcd + 1) / 8);   // this XCD's tiles
  const int cop = (Cout + COB - 1) / COB * COB, cip = (Cin + 63) / 64 * 64;
  const uint32_t xnch = (uint32_t)max(min(Cin - ci2 * 64, 64), 0), gnch = (uint32_t)max(min(Cout - co2 * COB, COB), 0);

  // per-thread staging geometry (the same for every tile), one packed word per task:
  // LDS slot (12 bits, 0xfff = no task) | channel << 12 (7 bits) | staged row << 19 (4 bits) | 8-pixel block << 23
  int xg[NXT], gg[NGT];
#pragma unroll
  for (int i = 0; i < NXT; ++i) {
    const int t = tid + i * NTH;
    const int c = t / (G::XR * G::XB), rem = t - c * (G::XR * G::XB), sr = rem / G::XB, bb = rem - sr * G::XB;
    xg[i] = ((t < 64 * G::XR * G::XB) ? c * G::XCH + sr * G::XB + bb : 0xfff) | (c << 12) | (sr << 19) | (bb << 23);
  }
#pragma unroll
  for (int i = 0; i < NGT; ++i) {
    const int t = tid + i * NTH;
    const int c = t / (TR * G::GB), rem = t - c * (TR * G::GB), k = rem / G::GB, bb = rem - k * G::GB;
    gg[i] = ((t < COB * TR * G::GB) ? c * G::GCH + k * G::GB + bb : 0xfff) | (c << 12) | (k << 19) | (bb << 23);
  }
  static_assert(64 * G::XCH < 0xfff && 128 * G::GCH < 0xfff, "LDS slot index must fit 12 bits");
  u32x4 px[NXT], pg[NGT];
  int sx[RAGGED ? NXT : 1], sg[RAGGED ? NGT : 1];                         // RAGGED: left shift of each staged block
  auto issue = [&](int tile) {                                           // global loads of one tile -> registers
    // the level this (uniform) tile index belongs to: scalar selects over the kernel arguments
    const T* x = (const T*)L.lv[0].x; const T* g = (const T*)L.lv[0].g;
    long long xbs = L.lv[0].xbs, gbs = L.lv[0].gbs;
    int H = L.lv[0].H, W = L.lv[0].W, tiles_x = L.lv[0].tiles_x, tiles_y = L.lv[0].tiles_y, t0 = 0;
#pragma unroll
    for (int i = 1; i < MAXL; ++i)
      if (i < L.n && tile >= L.lv[i].tile0) {
        x = (const T*)L.lv[i].x; g = (const T*)L.lv[i].g; xbs = L.lv[i].xbs; gbs = L.lv[i].gbs;
        H = L.lv[i].H; W = L.lv[i].W; tiles_x = L.lv[i].tiles_x; tiles_y = L.lv[i].tiles_y; t0 = L.lv[i].tile0;
      }
    const int lt = tile - t0;
    const int tx = lt % tiles_x, ty = (lt / tiles_x) % tiles_y, n = lt / (tiles_x * tiles_y);
    const int phase = ty % DD, q = ty / DD;
    const int y0 = phase + DD * q * TR, x0 = tx * TWP;                  // output rows y0 + k*DD, k < TR
    const bool live = tile < t_hi;
    const uint32_t plane = (uint32_t)H * (uint32_t)W * 2u;
    const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(x + (size_t)n * xbs + (size_t)ci2 * 64 * H * W), 0, live ? xnch * plane : 0u, 0x00020000);
    const __amdgpu_buffer_rsrc_t gr = __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(g + (size_t)n * gbs + (size_t)co2 * COB * H * W), 0, live ? gnch * plane : 0u, 0x00020000);
#pragma unroll
    for (int i = 0; i < NXT; ++i) {
      const int xc = (xg[i] >> 12) & 127, xsr = (xg[i] >> 19) & 15, xb8 = xg[i] >> 23;
      const int gy = (D == 0) ? y0 + xsr : y0 + (xsr - 1) * DD, gx = x0 - HALO + 8 * xb8;
      const bool in = gy >= 0 && gy < H && gx >= 0 && gx < W;
      int sh = 0;
      if constexpr (RAGGED) { sh = (in && gx + 8 > W) ? gx + 8 - W : 0; sx[i] = sh; }
      const uint32_t off = in ? (uint32_t)xc * plane + (uint32_t)(gy * W + gx - sh) * 2u : 0x80000000u;
      px[i] = __builtin_amdgcn_raw_buffer_load_b128(xr, off, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < NGT; ++i) {
      const int gc = (gg[i] >> 12) & 127, gk = (gg[i] >> 19) & 15, gb8 = gg[i] >> 23;
      const int gy = y0 + gk * DD, gx = x0 + 8 * gb8;
      const bool in = gy < H && gx < W;
      int sh = 0;
      if constexpr (RAGGED) { sh = (in && gx + 8 > W) ? gx + 8 - W : 0; sg[i] = sh; }
      const uint32_t off = in ? (uint32_t)gc * plane + (uint32_t)(gy * W + gx - sh) * 2u : 0x80000000u;
      pg[i] = __builtin_amdgcn_raw_buffer_load_b128(gr, off, 0, 0);
    }
  };
  auto land = [&]() {                                                    // registers -> LDS
#pragma unroll
    for (int i = 0; i < NXT; ++i)
      if ((xg[i] & 0xfff) != 0xfff) xs[xg[i] & 0xfff] = __builtin_bit_cast(uint4, RAGGED ? shr_pixels(px[i], sx[RAGGED ? i : 0]) : px[i]);
#pragma unroll
    for (int i = 0; i < NGT; ++i)
      if ((gg[i] & 0xfff) != 0xfff) gs[gg[i] & 0xfff] = __builtin_bit_cast(uint4, RAGGED ? shr_pixels(pg[i], sg[RAGGED ? i : 0]) : pg[i]);
  };

  f32x16 acc[G::NT];
#pragma unroll
  for (int t = 0; t < G::NT; ++t)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[t][e] = 0.f;

  issue(t_lo + jj);
  land();
  __syncthreads();
  const uint4* xw = xs + (cib * 32 + ch) * G::XCH;
  const uint4* gw = gs + (cob * 32 + ch) * G::GCH;
  for (int tile = t_lo + jj; tile < t_hi; tile += J) {
    issue(tile + J);                                                     // (beyond the last tile: a null descriptor, zeros)
    // ---- matrix work on the tile in LDS.  The LDS reads of staged row s+1 (and of the next k-step's g operands) are issued
    // BEFORE the MFMAs of row s: left to itself hipcc places each ds_read right before its first use, and with one wave
    // per SIMD nothing else hides the ~130-cycle LDS latency — the matrix pipe idled three quarters of the time.
    if constexpr (D == 0) {
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        const int blk = 2 * ks + kg;                                     // this lane's 8-pixel block inside the 32-pixel row
#pragma unroll
        for (int k = 0; k < TR; ++k) acc[0] = Mma32<T>::mma(gw[k * G::GB + blk], xw[k * G::XB + blk], acc[0]);
      }
    } else if constexpr (MB == 2) {
      // two waves per SIMD on 256 registers each: they cover for each other's LDS latency (reading ahead would spill)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        const int blk = 2 * ks + kg;
        uint4 a[TR];
#pragma unroll
        for (int k = 0; k < TR; ++k) a[k] = gw[k * G::GB + blk];
#pragma unroll
        for (int s2 = 0; s2 < G::XR; ++s2) {
          const uint4* row = xw + s2 * G::XB + HALO / 8 + blk;
          uint4 p2 = make_uint4(0, 0, 0, 0), n2 = p2;
          const uint4 p1 = row[-1], c = row[0], n1 = row[1];
          if constexpr (D == 16) { p2 = row[-2]; n2 = row[2]; }
          const uint4 w0 = window<DD>(p2, p1, c, n1, n2), w2 = window<-DD>(p2, p1, c, n1, n2);
#pragma unroll
          for (int ky = 0; ky < 3; ++ky) {
            const int k = s2 - ky;
            if (k >= 0 && k < TR) {
              acc[ky * 3 + 0] = Mma32<T>::mma(a[k], w0, acc[ky * 3 + 0]);
              acc[ky * 3 + 1] = Mma32<T>::mma(a[k], c, acc[ky * 3 + 1]);
              acc[ky * 3 + 2] = Mma32<T>::mma(a[k], w2, acc[ky * 3 + 2]);
            }
          }
        }
      }
    } else {
      struct Row { uint4 p2, p1, c, n1, n2; };
      auto rload = [&](int ks, int s2) {
        const uint4* row = xw + s2 * G::XB + HALO / 8 + 2 * ks + kg;
        Row r;
        r.p2 = r.n2 = make_uint4(0, 0, 0, 0);
        r.p1 = row[-1]; r.c = row[0]; r.n1 = row[1];
        if constexpr (D == 16) { r.p2 = row[-2]; r.n2 = row[2]; }
        return r;
      };
      uint4 a[2][TR];
#pragma unroll
      for (int k = 0; k < TR; ++k) a[0][k] = gw[k * G::GB + kg];
      Row cur = rload(0, 0);
#pragma unroll
      for (int it = 0; it < 2 * G::XR; ++it) {
        const int ks = it / G::XR, s2 = it % G::XR;
        Row nxt = cur;
        if (it + 1 < 2 * G::XR) nxt = rload((it + 1) / G::XR, (it + 1) % G::XR);
        if (ks == 0 && s2 == 0) {
#pragma unroll
          for (int k = 0; k < TR; ++k) a[1][k] = gw[k * G::GB + 2 + kg];
        }
        __builtin_amdgcn_sched_barrier(0);
        const uint4 w0 = window<DD>(cur.p2, cur.p1, cur.c, cur.n1, cur.n2), w2 = window<-DD>(cur.p2, cur.p1, cur.c, cur.n1, cur.n2);
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
          const int k = s2 - ky;                                         // staged row s2 = output row k + kernel row ky
          if (k >= 0 && k < TR) {
            acc[ky * 3 + 0] = Mma32<T>::mma(a[ks][k], w0, acc[ky * 3 + 0]);
            acc[ky * 3 + 1] = Mma32<T>::mma(a[ks][k], cur.c, acc[ky * 3 + 1]);
            acc[ky * 3 + 2] = Mma32<T>::mma(a[ks][k], w2, acc[ky * 3 + 2]);
          }
        }
        __builtin_amdgcn_sched_barrier(0);
        cur = nxt;
      }
    }
    __syncthreads();                                                     // every wave is done reading this tile
    land();
    __syncthreads();
  }
  // ---- partial block: D layout col = lane & 31 (ci), row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5) (co)
  float* pb = partial + (size_t)ks_id * G::NT * cop * cip;
  const int ci = ci2 * 64 + cib * 32 + ch;
#pragma unroll
  for (int t = 0; t < G::NT; ++t)
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int co = co2 * COB + cob * 32 + (e & 3) + 8 * (e >> 2) + 4 * kg;
      pb[((size_t)t * cop + co) * cip + ci] = acc[t][e];
    }
}

// ---- producer / consumer form -------------------------------------------------------------------------------------------
// PMC picture of wgrad_kernel (profiles/r02_wgrad_567_128_pmc.txt): 11 K cycles per tile and SIMD against 4.6 K of MFMA issue —
// every wave stages (address arithmetic, loads, LDS writes), waits at a workgroup barrier, multiplies, waits again: the
// phases of all waves coincide, so the matrix pipe idles while the tile is staged.  Here the roles are split: waves 0-3
// (consumers) only read LDS and issue MFMAs, waves 4-7 (producers) only stage — loads of tile t+3 in flight in one of two
// register sets while tile t+1 is written to the OTHER of two LDS buffers — so that a SIMD always holds one wave of each
// kind and issues MFMAs and staging instructions side by side; one barrier per tile swaps the buffers.  64 co x 64 ci per
// workgroup, same partial layout, same XCD-aware tile order as wgrad_kernel<.., MB = 1>.
// RAGGED here = "some level of this launch is ragged" (KLevel::ragged says which): the producers pick the shifted staging
// PER TILE by a wave-uniform branch, so aligned and ragged pyramid levels share one launch, one accumulation and one set
// of partial blocks.  (Round 4 gave the ragged levels — 52, 26, 13 pixels wide, 6 % of a config-3 step's pixels — their own
// launch and their own K-splits: 25 launches of ~20 us and half of the partial blocks; pushing ALL tiles through the shifted
// staging instead cost the aligned 94 % more than those launches did.)
template <typename T, int D, bool RAGGED>
__global__ __launch_bounds__(2 * NTHREADS, 2)
void wgrad_pc_kernel(const KLevels L, float* __restrict__ partial, int Cin, int Cout, int nci2, int J_ns, int co2_base) {
  using G = Geo<D>;
  constexpr int DD = (D == 0) ? 1 : D;
  constexpr int HALO = halo_of(D);
  constexpr int NXT = (64 * G::XR * G::XB + NTHREADS - 1) / NTHREADS;      // X staging tasks per producer thread
  constexpr int NGT = (64 * TR * G::GB + NTHREADS - 1) / NTHREADS;         // g staging tasks per producer thread
  constexpr int BUF = G::X_BLOCKS + G::G_BLOCKS;                            // 16-byte blocks per LDS buffer
  extern __shared__ __attribute__((aligned(16))) uint4 smem[];            // two buffers: [X | g] [X | g]
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // J_ns = J | NS << 16: the tile range is cut into NS slices (8 = one per XCD, the mapping described at wgrad_kernel; fewer for
  // the layers whose K is a handful of tiles: see pick_split) and J workgroups per block pair walk a slice side by side
  const int J = J_ns & 0xffff, NS = J_ns >> 16;
  const int nblocks = gridDim.x / (NS * J);
  const int xcd = blockIdx.x % NS, slot = blockIdx.x / NS;
  const int bpair = slot % nblocks, jj = slot / nblocks;
  const int ks_id = xcd * J + jj;
  // experiments (UPF_WGRAD_ABLATE, read by the host): 1 = no matrix phase, 2 = null descriptors (loads return zeros, no traffic)
  const int abl = co2_base >> 16;
  const int co2 = bpair / nci2 + (co2_base & 0xffff), ci2 = bpair % nci2;
  const int t_lo = (int)((long long)L.ntiles * xcd / NS), t_hi = (int)((long long)L.ntiles * (xcd + 1) / NS);
  const int first = t_lo + jj;
  const int niter = first < t_hi ? (t_hi - first + J - 1) / J : 0;       // tiles of this workgroup (uniform)
  const int cop = (Cout + 63) / 64 * 64, cip = (Cin + 63) / 64 * 64;

  // Level table in LDS (behind the two staging buffers): the producers fetch a level's parameters when their tile index crosses
  // into it.  (Rounds 2-5 selected them out of the kernel arguments with a compare / select tree in EVERY tile, which kept
  // ~100 scalars live across the loop: the register allocator spilled them into vector lanes — 118 v_readlane per tile.)
  int* ltab = reinterpret_cast<int*>(smem + 2 * BUF);
  if (tid == 0) {
#pragma unroll
    for (int i = 0; i < MAXL; ++i) {
      int* r = ltab + i * LT_STRIDE;
      const unsigned long long px = (unsigned long long)L.lv[i].x, pg = (unsigned long long)L.lv[i].g;
      r[0] = (int)(uint32_t)px; r[1] = (int)(uint32_t)(px >> 32); r[2] = (int)(uint32_t)pg; r[3] = (int)(uint32_t)(pg >> 32);
      r[4] = (int)(uint32_t)L.lv[i].xbs; r[5] = (int)(uint32_t)((unsigned long long)L.lv[i].xbs >> 32);
      r[6] = (int)(uint32_t)L.lv[i].gbs; r[7] = (int)(uint32_t)((unsigned long long)L.lv[i].gbs >> 32);
      r[8] = L.lv[i].H; r[9] = L.lv[i].W; r[10] = L.lv[i].tiles_x; r[11] = L.lv[i].tiles_y;
      r[12] = L.lv[i].tile0; r[13] = L.lv[i].ragged; r[14] = (i + 1 < L.n) ? L.lv[i + 1 < MAXL ? i + 1 : i].tile0 : 0x7fffffff; r[15] = 0;
    }
  }
  __syncthreads();

  if (wave >= 4) {
    // ================= producers
    // A producer wave shares its SIMD with a consumer wave and one wave issues at most one vector instruction every four
    // cycles, so the producers are bound by their INSTRUCTION COUNT long before memory (tools/wgrad_ablate.py: with null
    // descriptors — no traffic at all — and no matrix phase the kernel took 92 % of its full time).  Per tile and thread: the
    // tile decode (two reciprocals), then for an INTERIOR tile (no staged block leaves the image: most of them) 13 loads whose
    // vector offset is a per-level constant and whose tile origin rides in the scalar offset operand — no vector arithmetic —
    // and 13 LDS writes.  Only border tiles compute per-block validity (and, on ragged levels, the shift of a straddling block).
    static_assert((64 * G::XR * G::XB) % NTHREADS == 0 && (64 * TR * G::GB) % NTHREADS == 0, "every producer thread has the same number of live tasks");
    const int ptid = tid - NTHREADS;
    const int xnch = max(min(Cin - ci2 * 64, 64), 0), gnch = max(min(Cout - co2 * 64, 64), 0);
    constexpr uint32_t OOB = 0x80000000u;
    // tasks of this thread (the same for every tile): LDS slot; row / column relative to the tile origin, packed
    int xslot[NXT], xrc[NXT], gslot[NGT], grc[NGT];
#pragma unroll
    for (int i = 0; i < NXT; ++i) {
      const int t = ptid + i * NTHREADS;
      const int c = t / (G::XR * G::XB), rem = t - c * (G::XR * G::XB), sr = rem / G::XB, bb = rem - sr * G::XB;
      xslot[i] = c * G::XCH + sr * G::XB + bb;
      xrc[i] = ((((D == 0) ? sr : (sr - 1) * DD) + 64) << 16) | (8 * bb - HALO + 64);       // (+64: both halves non-negative)
    }
#pragma unroll
    for (int i = 0; i < NGT; ++i) {
      const int t = ptid + i * NTHREADS;
      const int c = t / (TR * G::GB), rem = t - c * (TR * G::GB), k = rem / G::GB, bb = rem - k * G::GB;
      gslot[i] = c * G::GCH + k * G::GB + bb;
      grc[i] = ((k * DD + 64) << 16) | (8 * bb + 64);
    }
    // per level: vector offsets of the tasks relative to the tile origin, shifted up by `bias` bytes so that they are never
    // negative (the descriptor's base is moved down by the same amount); OOB for the channels beyond Cin / Cout
    uint32_t xrel[NXT], grel[NGT];
    int lev = -1, next_t0 = 0;
    bool lrag = false;
    const T* lx = nullptr; const T* lg = nullptr;
    long long lxbs = 0, lgbs = 0;
    int H = 1, W = 1, tiles_x = 1, txy = 1, t0 = 0, bias = 0;
    float rtx = 1.f, rtxy = 1.f;
    uint32_t plane = 0;
    auto rfl = [](int v) { return __builtin_amdgcn_readfirstlane(v); };
    auto enter_level = [&](int tile) {
      while (tile >= next_t0) {                                          // (uniform; next_t0 of the last level: INT_MAX)
        ++lev;
        const int4* r = reinterpret_cast<const int4*>(ltab + lev * LT_STRIDE);
        const int4 a = r[0], b = r[1], c = r[2], d = r[3];
        lx = (const T*)(((unsigned long long)(uint32_t)rfl(a.y) << 32) | (uint32_t)rfl(a.x));
        lg = (const T*)(((unsigned long long)(uint32_t)rfl(a.w) << 32) | (uint32_t)rfl(a.z));
        lxbs = (long long)(((unsigned long long)(uint32_t)rfl(b.y) << 32) | (uint32_t)rfl(b.x));
        lgbs = (long long)(((unsigned long long)(uint32_t)rfl(b.w) << 32) | (uint32_t)rfl(b.z));
        H = rfl(c.x); W = rfl(c.y); tiles_x = rfl(c.z); txy = tiles_x * rfl(c.w);
        t0 = rfl(d.x); next_t0 = rfl(d.z);
        if constexpr (RAGGED) lrag = rfl(d.y) != 0;
        plane = (uint32_t)H * (uint32_t)W * 2u;
        bias = 2 * (DD * W + HALO + 8);
        rtx = 1.0f / (float)tiles_x; rtxy = 1.0f / (float)txy;
#pragma unroll
        for (int i = 0; i < NXT; ++i) {
          const int ch = (ptid + i * NTHREADS) / (G::XR * G::XB), row = (xrc[i] >> 16) - 64, col = (xrc[i] & 0xffff) - 64;
          xrel[i] = ch < xnch ? (uint32_t)ch * plane + (uint32_t)((row * W + col) * 2 + bias) : OOB;
        }
#pragma unroll
        for (int i = 0; i < NGT; ++i) {
          const int ch = (ptid + i * NTHREADS) / (TR * G::GB), row = (grc[i] >> 16) - 64, col = (grc[i] & 0xffff) - 64;
          grel[i] = ch < gnch ? (uint32_t)ch * plane + (uint32_t)((row * W + col) * 2 + bias) : OOB;
        }
      }
    };
    // a / b for 0 <= a < 2^24 through the reciprocal (exact after one correction step either way)
    auto fdiv = [](int a, int b, float rb) { int q = (int)((float)a * rb); int r = a - q * b; q += (r >= b) - (r < 0); return q; };
    // one staged tile in flight: the loaded blocks and, for a border tile of a ragged level, the left shift of every block (4 bits each)
    struct Set { u32x4 px[NXT], pg[NGT]; unsigned long long shx; uint32_t shg; bool rag; };
    Set S0, S1;
    S0.rag = S1.rag = false; S0.shx = S1.shx = 0; S0.shg = S1.shg = 0;
    auto issue = [&](int tile, Set& S) {
      enter_level(tile);
      const int lt = tile - t0;
      const int n = rfl(fdiv(lt, txy, rtxy)), r2 = lt - n * txy;
      const int ty = rfl(fdiv(r2, tiles_x, rtx)), tx = r2 - ty * tiles_x;
      const int phase = ty % DD, q = ty / DD;                            // (compile-time divisor)
      const int y0 = phase + DD * q * TR, x0 = tx * TWP;
      const bool live = tile < t_hi && !(abl & 2);                      // (dead: null descriptors, every load returns zeros)
      const char* xb = reinterpret_cast<const char*>(lx + (size_t)n * lxbs + (size_t)ci2 * 64 * H * W) - bias;
      const char* gb = reinterpret_cast<const char*>(lg + (size_t)n * lgbs + (size_t)co2 * 64 * H * W) - bias;
      const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(uniform_ptr(xb)), 0, rfl(live ? (int)((uint32_t)xnch * plane) + bias : 0), 0x00020000);
      const __amdgpu_buffer_rsrc_t gr = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(uniform_ptr(gb)), 0, rfl(live ? (int)((uint32_t)gnch * plane) + bias : 0), 0x00020000);
      const int torg = rfl((y0 * W + x0) * 2);                           // the tile origin: scalar offset operand of every load
      const bool interior = (D == 0 ? true : y0 >= DD) && y0 + (D == 0 ? TR - 1 : TR * DD) < H && x0 >= HALO && x0 + TWP + HALO <= W;
      uint32_t vx[NXT], vg[NGT];
      if (interior) {
#pragma unroll
        for (int i = 0; i < NXT; ++i) vx[i] = xrel[i];
#pragma unroll
        for (int i = 0; i < NGT; ++i) vg[i] = grel[i];
        if constexpr (RAGGED) S.rag = false;
      } else {
        unsigned long long shx = 0; uint32_t shg = 0;
#pragma unroll
        for (int i = 0; i < NXT; ++i) {
          const int gy = y0 + (xrc[i] >> 16) - 64, gx = x0 + (xrc[i] & 0xffff) - 64;
          const bool in = (unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W;
          int sh = 0;
          if constexpr (RAGGED) { sh = (in && gx + 8 > W) ? gx + 8 - W : 0; shx |= (unsigned long long)sh << (4 * i); }
          vx[i] = in ? xrel[i] - (uint32_t)(2 * sh) : OOB;
        }
#pragma unroll
        for (int i = 0; i < NGT; ++i) {
          const int gy = y0 + (grc[i] >> 16) - 64, gx = x0 + (grc[i] & 0xffff) - 64;
          const bool in = gy < H && gx < W;
          int sh = 0;
          if constexpr (RAGGED) { sh = (in && gx + 8 > W) ? gx + 8 - W : 0; shg |= (uint32_t)sh << (4 * i); }
          vg[i] = in ? grel[i] - (uint32_t)(2 * sh) : OOB;
        }
        if constexpr (RAGGED) { S.rag = lrag; S.shx = shx; S.shg = shg; }
      }
      // ONE load sequence behind the branch (loads inside its arms would be scheduled differently in each, and the wait-count
      // insertion would have to assume the worst position of every register at the join)
#pragma unroll
      for (int i = 0; i < NXT; ++i) S.px[i] = __builtin_amdgcn_raw_buffer_load_b128(xr, vx[i], torg, 0);
#pragma unroll
      for (int i = 0; i < NGT; ++i) S.pg[i] = __builtin_amdgcn_raw_buffer_load_b128(gr, vg[i], torg, 0);
    };
    auto land = [&](const Set& S, uint4* buf) {
#ifdef UPF_WGRAD_NO_LAND                                                 // experiment build: no LDS staging writes
      return;
#endif
      uint4* xs = buf; uint4* gs = buf + G::X_BLOCKS;
      if (RAGGED && S.rag) {
#pragma unroll
        for (int i = 0; i < NXT; ++i) xs[xslot[i]] = __builtin_bit_cast(uint4, shr_pixels(S.px[i], (int)((S.shx >> (4 * i)) & 15)));
#pragma unroll
        for (int i = 0; i < NGT; ++i) gs[gslot[i]] = __builtin_bit_cast(uint4, shr_pixels(S.pg[i], (int)((S.shg >> (4 * i)) & 15)));
      } else {
#pragma unroll
        for (int i = 0; i < NXT; ++i) xs[xslot[i]] = __builtin_bit_cast(uint4, S.px[i]);
#pragma unroll
        for (int i = 0; i < NGT; ++i) gs[gslot[i]] = __builtin_bit_cast(uint4, S.pg[i]);
      }
    };
    // tile k of this workgroup = first + k * J (dead beyond niter: a null descriptor, zeros).  Set k & 1 carries tile k.
    issue(first, S0);
    issue(first + J, S1);
    land(S0, smem);                                                      // tile 0 -> buffer 0
    issue(first + 2 * J, S0);
    __syncthreads();
    // Both halves of an iteration run UNCONDITIONALLY (a tile beyond the last one is dead: zeros into a buffer nobody reads any
    // more; the consumers take the matching barrier).  Rounds 2-4 skipped the second half after an odd last tile — and hipcc's
    // wait-count insertion, which must assume at the loop header that the skipped path was taken (then S1's loads are the most
    // RECENT ones), drained the whole queue (s_waitcnt vmcnt(12) ... vmcnt(0)) before landing S1 in EVERY iteration: the
    // loads of the other set, issued half an iteration earlier, never stayed in flight across a landing.  Now vmcnt(25) ... (13).
    for (int it = 0; it < niter; it += 2) {
      // consumers multiply tile `it` out of buffer 0
      land(S1, smem + BUF);                                              // tile it+1 -> buffer 1
      issue(first + (it + 3) * J, S1);
      __syncthreads();
      // consumers multiply tile it+1 out of buffer 1
      land(S0, smem);                                                    // tile it+2 -> buffer 0
      issue(first + (it + 4) * J, S0);
      __syncthreads();
    }
    return;
  }

  // ================= consumers
  const int cob = wave & 1, cib = wave >> 1;
  const int ch = lane & 31, kg = lane >> 5;
  f32x16 acc[G::NT];
#pragma unroll
  for (int t = 0; t < G::NT; ++t)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[t][e] = 0.f;
  auto multiply = [&](const uint4* buf) {
    const uint4* xw = buf + (cib * 32 + ch) * G::XCH;
    const uint4* gw = buf + G::X_BLOCKS + (cob * 32 + ch) * G::GCH;
    if constexpr (D == 0) {
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        const int blk = 2 * ks + kg;
#pragma unroll
        for (int k = 0; k < TR; ++k) acc[0] = Mma32<T>::mma(gw[k * G::GB + blk], xw[k * G::XB + blk], acc[0]);
      }
    } else {
      struct Row { uint4 p2, p1, c, n1, n2; };
      auto rload = [&](int ks, int s2) {
        const uint4* row = xw + s2 * G::XB + HALO / 8 + 2 * ks + kg;
        Row r;
        r.p2 = r.n2 = make_uint4(0, 0, 0, 0);
        r.c = row[0];
        if constexpr (D == 16) {
          r.p1 = r.n1 = make_uint4(0, 0, 0, 0);                        // (window<+-16> = the blocks two before / after)
          r.p2 = row[-2]; r.n2 = row[2];
        } else {
          // The two half-waves (kg = 0 / 1) work on neighbouring blocks b and b + 1, so the centre block of one IS the side block
          // of the other: each half reads its centre and its OUTER side block only (two ds_read_b128 instead of three: the
          // consumers are LDS-issue bound) and the halves exchange the dwords of their centres that the shifted windows use
          // through v_permlane32_swap (a'.hi = b.lo, b'.lo = a.hi: tools/permlane_probe.hip).
          const uint4 side = row[kg ? 1 : -1];
          r.p1 = r.n1 = side;
          auto swp = [](uint32_t a, uint32_t b, uint32_t& a_out, uint32_t& b_out) {
            const auto q = __builtin_amdgcn_permlane32_swap(a, b, false, false);
            a_out = q[0]; b_out = q[1];
          };
          uint32_t fa, fb;                                             // fa: for kg = 1 the other half's centre dword; fb: for kg = 0
          if constexpr (DD == 1 || DD == 2) {
            swp(r.c.x, r.c.w, fa, fb);
            if (kg) r.p1.w = fa; else r.n1.x = fb;
          } else if constexpr (DD == 4) {
            swp(r.c.x, r.c.z, fa, fb);
            if (kg) r.p1.z = fa; else r.n1.x = fb;
            swp(r.c.y, r.c.w, fa, fb);
            if (kg) r.p1.w = fa; else r.n1.y = fb;
          } else {                                                     // DD == 8: whole blocks
            swp(r.c.x, r.c.x, fa, fb); if (kg) r.p1.x = fa; else r.n1.x = fb;
            swp(r.c.y, r.c.y, fa, fb); if (kg) r.p1.y = fa; else r.n1.y = fb;
            swp(r.c.z, r.c.z, fa, fb); if (kg) r.p1.z = fa; else r.n1.z = fb;
            swp(r.c.w, r.c.w, fa, fb); if (kg) r.p1.w = fa; else r.n1.w = fb;
          }
        }
        return r;
      };
      uint4 a[2][TR];
#pragma unroll
      for (int k = 0; k < TR; ++k) a[0][k] = gw[k * G::GB + kg];
      // rows are read TWO steps ahead of their MFMAs (a step is 3-9 MFMAs = 100-300 cycles, an LDS read under load more)
      constexpr bool DEEP = (D != 16);
      Row cur = rload(0, 0);
      Row nx1 = rload(0, 1);
#pragma unroll
      for (int it = 0; it < 2 * G::XR; ++it) {
        const int ks = it / G::XR, s2 = it % G::XR;
        Row nxt = nx1;
        if constexpr (DEEP) {
          if (it + 2 < 2 * G::XR) nx1 = rload((it + 2) / G::XR, (it + 2) % G::XR);
        } else {
          if (it + 1 < 2 * G::XR) nxt = rload((it + 1) / G::XR, (it + 1) % G::XR);
        }
        if (ks == 0 && s2 == 0) {
#pragma unroll
          for (int k = 0; k < TR; ++k) a[1][k] = gw[k * G::GB + 2 + kg];
        }
        __builtin_amdgcn_sched_barrier(0);
        const uint4 w0 = window<DD>(cur.p2, cur.p1, cur.c, cur.n1, cur.n2), w2 = window<-DD>(cur.p2, cur.p1, cur.c, cur.n1, cur.n2);
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
          const int k = s2 - ky;
          if (k >= 0 && k < TR) {
            acc[ky * 3 + 0] = Mma32<T>::mma(a[ks][k], w0, acc[ky * 3 + 0]);
            acc[ky * 3 + 1] = Mma32<T>::mma(a[ks][k], cur.c, acc[ky * 3 + 1]);
            acc[ky * 3 + 2] = Mma32<T>::mma(a[ks][k], w2, acc[ky * 3 + 2]);
          }
        }
        __builtin_amdgcn_sched_barrier(0);
        cur = nxt;
      }
    }
  };
  __syncthreads();                                                       // tile 0 is in buffer 0
  // a consumer wave whose 32 output channels (or 32 input channels) are all padding — Cout <= 32: half of the waves of every
  // narrow layer — has nothing to multiply: it only takes the barriers (its LDS reads and MFMA issue slots go to the others)
  const bool work = !(abl & 1) && co2 * 64 + cob * 32 < Cout && ci2 * 64 + cib * 32 < Cin;
  for (int it = 0; it < niter; it += 2) {
    if (work) multiply(smem);
    __syncthreads();
    if (work && it + 1 < niter) multiply(smem + BUF);
    __syncthreads();
  }
  // partial block; rows / columns beyond Cout / Cin are never read by the reduction and are not written (a 563 -> 2 layer
  // used to write 32 MB of zeros per launch)
  float* pb = partial + (size_t)ks_id * G::NT * cop * cip;
  const int ci = ci2 * 64 + cib * 32 + ch;
  if (ci < ((Cin + 3) & ~3)) {
#pragma unroll
    for (int t = 0; t < G::NT; ++t)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int co = co2 * 64 + cob * 32 + (e & 3) + 8 * (e >> 2) + 4 * kg;
        if (co < Cout) pb[((size_t)t * cop + co) * cip + ci] = acc[t][e];
      }
  }
}

// dw[co][ci][tap] = sum over the K-splits, fixed order.  A thread owns 4 consecutive ci of one co for ALL taps: its reads
// are 16-byte loads that a wave lays side by side (the partial layout is [split][tap][co][ci], ci fastest), NT independent
// loads per split in flight, and its NT x 4 results are 36 consecutive floats of dw.  The four waves of a workgroup take
// every fourth split and are summed through LDS in wave order.  (The first version — one thread per element, a dependent
// chain of 4-byte loads over the splits, 36-byte-strided stores — ran at 1 TB/s: 36 us per layer, 1.15 ms per step.)
// second stage of the bias gradient over the first-stage sums (C x BIAS_NCH, act_grad_kernel / bias_grad_partial_kernel) of several
// uses of one bias (the pyramid levels), in the order given
constexpr int MAXP = 8, BIAS_NCH = 32;
struct BiasParts { const float* p[MAXP]; int n; float* db; };
__device__ __forceinline__ void bias_finish(const BiasParts& P, int co, int Cout) {
  if (co >= Cout) return;
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < MAXP; ++i)
    if (i < P.n) {
      float t = 0.f;
      for (int k = 0; k < BIAS_NCH; ++k) t += P.p[i][co * BIAS_NCH + k];
      s += t;
    }
  P.db[co] = s;
}

template <int NT, int SL>
__global__ __launch_bounds__(256)
void wgrad_reduce_kernel(const float* __restrict__ partial, float* __restrict__ dw, int ksplit, int Cout, int Cin, int cop, int s2d, int nwblocks, const BiasParts BP) {
  // (the blocks behind the weight items finish the layer's BIAS gradient — one launch fewer per layer and step)
  if ((int)blockIdx.x >= nwblocks) { bias_finish(BP, ((int)blockIdx.x - nwblocks) * 256 + (int)threadIdx.x, Cout); return; }
  // SL slices of the split range per item, 256 / SL items per workgroup.  SL = 4 streams the wide layers (many items: the grid
  // fills the chip, a thread walks ksplit / 4 splits); the narrow layers (Cout <= 32: 3 ... 70 workgroups of SL = 4, each thread
  // a chain of 6-16 dependent rounds of loads: 19-35 us for a few MB, round 5's timeline) take SL = 16: four times the
  // workgroups, a quarter of the rounds.  Summation order: slice s holds splits s, s + SL, ... in order; slices are added in order.
  constexpr int IPW = 256 / SL;
  __shared__ float4 sh[SL - 1][NT][IPW];
  const int cip = (Cin + 63) / 64 * 64, c4n = (Cin + 3) / 4;               // (columns beyond Cin are padding: not written, not read)
  const int q = threadIdx.x % IPW, slice = threadIdx.x / IPW;
  const long long item = blockIdx.x * (long long)IPW + q;
  const bool live = item < (long long)Cout * c4n;
  const int co = live ? (int)(item / c4n) : 0, c4 = live ? (int)(item - (long long)co * c4n) : 0;
  float4 acc[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) acc[t] = make_float4(0.f, 0.f, 0.f, 0.f);
  const size_t tstride = (size_t)cop * cip, kstride = (size_t)NT * tstride;
  if (live) {
    const float* p = partial + (size_t)co * cip + (size_t)c4 * 4 + (size_t)slice * kstride;
    for (int k = slice; k < ksplit; k += SL, p += SL * kstride) {
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const float4 v = *reinterpret_cast<const float4*>(p + t * tstride);
        acc[t].x += v.x; acc[t].y += v.y; acc[t].z += v.z; acc[t].w += v.w;
      }
    }
  }
  if (slice) {
#pragma unroll
    for (int t = 0; t < NT; ++t) sh[slice - 1][t][q] = acc[t];
  }
  __syncthreads();
  if (slice || !live) return;
  float out[4][NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    float4 r = acc[t];
#pragma unroll
    for (int s = 0; s < SL - 1; ++s) {
      const float4 a = sh[s][t][q];
      r.x += a.x; r.y += a.y; r.z += a.z; r.w += a.w;
    }
    out[0][t] = r.x; out[1][t] = r.y; out[2][t] = r.z; out[3][t] = r.w;
  }
  if constexpr (NT == 9) {
    if (s2d) {
      // space-to-depth form of a stride-2 layer (ops.py: _s2d_ok): this thread's 4 channels are the 4 phases (p, q) of input
      // channel c4; dw[co][c4][ky][kx] = (phase (P[ky], P[kx]), tap (A[ky], A[kx])) with P = {1,0,1}, A = {0,1,1}
      if (c4 * 4 < Cin) {
        float* d = dw + ((size_t)co * (Cin / 4) + c4) * 9;
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
          for (int kx = 0; kx < 3; ++kx) {
            const int P[3] = {1, 0, 1}, A[3] = {0, 1, 1};
            d[ky * 3 + kx] = out[P[ky] * 2 + P[kx]][A[ky] * 3 + A[kx]];
          }
      }
      return;
    }
  }
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int ci = c4 * 4 + e;
    if (ci < Cin) {
      float* d = dw + ((size_t)co * Cin + ci) * NT;
#pragma unroll
      for (int t = 0; t < NT; ++t) d[t] = out[e][t];
    }
  }
}

// g = gy * (y > 0 ? 1 : slope): gradient through the fused LeakyReLU of the forward kernel (y = its OUTPUT; for
// slope > 0 the sign of the output is the sign of the pre-activation); 8 elements per thread.
template <typename T>
__global__ void leaky_bwd_kernel(const T* __restrict__ gy, const T* __restrict__ y, T* __restrict__ g, long long n8, float slope) {
  const long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x;
  if (i >= n8) return;
  float a[8], b[8];
  VecIO<T>::load(gy + 8 * i, a);
  VecIO<T>::load(y + 8 * i, b);
#pragma unroll
  for (int k = 0; k < 8; ++k) a[k] = (b[k] > 0.f) ? a[k] : a[k] * slope;
  VecIO<T>::store(g + 8 * i, a);
}

// db[co] = sum over n, pixels of g[n, co, :].  Two launches, fixed summation order: (channel, chunk) workgroups reduce
// 1/NCH of a channel's pixels each (one workgroup per channel left 2..128 workgroups on 256 CUs: 69 us per layer),
// then one thread per channel adds the NCH partial sums in order.
template <typename T>
__global__ __launch_bounds__(256)
void bias_grad_partial_kernel(const T* __restrict__ g, long long gbs, float* __restrict__ part, int B, int HW) {
  __shared__ float sh[4];
  const int co = blockIdx.x, chunk = blockIdx.y;
  const long long total = (long long)B * HW, per = (total + BIAS_NCH - 1) / BIAS_NCH;
  const long long e0 = chunk * per, e1 = min(total, e0 + per);
  float s = 0.f;
  for (long long e = e0 + threadIdx.x; e < e1; e += 256) {
    const long long n = e / HW;
    s += Elem<T>::load(g + (size_t)n * gbs + (size_t)co * HW + (e - n * HW));
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) part[co * BIAS_NCH + chunk] = (sh[0] + sh[1]) + (sh[2] + sh[3]);
}
__global__ void bias_grad_final_kernel(const float* __restrict__ part, float* __restrict__ db, int Cout) {
  const int co = blockIdx.x * blockDim.x + threadIdx.x;
  if (co >= Cout) return;
  float s = 0.f;
  for (int k = 0; k < BIAS_NCH; ++k) s += part[co * BIAS_NCH + k];
  db[co] = s;
}

// The gradient entering a layer's pre-activation, in ONE pass over a channel-sliced tensor (the dense stacks keep their
// gradients in one buffer, see ops.DenseStackTrainFunction):
//     dst = (src + add) * (y > 0 ? 1 : slope)        add, y optional
// and, while the values are in registers, the first stage of the bias gradient: part[co][chunk] = sum of the ROUNDED dst
// values over 1/BIAS_NCH of the channel's (image, pixel) range — the same (channel, chunk) grid and fixed summation
// order as bias_grad_partial_kernel.  V = elements per access (4: rows of 4k pixels at 8-byte aligned slices, else 1).
template <typename T> __device__ __forceinline__ float bits_f(unsigned short b) { T t; t.v = b; return Elem<T>::load(&t); }
template <typename T> __device__ __forceinline__ float round16(float v) { T t; Elem<T>::store(&t, v); return Elem<T>::load(&t); }
// NTH threads per (channel, chunk) workgroup: 256, or 1024 where the grid is small and the chunks are long (16 ... 32 channels at 1/2
// ... 1/4 resolution: 512 workgroups of 256 threads streamed 41 MB in 63 us)
template <typename T, int V, int NTH = 256>
__global__ __launch_bounds__(NTH)
void act_grad_kernel(const T* __restrict__ src, long long sbs, const T* __restrict__ add, long long abs_, const T* __restrict__ y, long long ybs,
                     T* __restrict__ dst, long long dbs, float* __restrict__ part, int B, int HW, float slope) {
  __shared__ float sh[NTH / 64];
  const int co = blockIdx.x, chunk = blockIdx.y;
  const long long total = (long long)B * (HW / V), per = (total + BIAS_NCH - 1) / BIAS_NCH;
  const long long e0 = chunk * per, e1 = min(total, e0 + per);
  const int hwv = HW / V;
  float s = 0.f;
  for (long long e = e0 + threadIdx.x; e < e1; e += NTH) {
    const long long n = e / hwv;
    const size_t o = (size_t)co * HW + (size_t)(e - n * hwv) * V;
    float v[V];
    if constexpr (V == 4) {
      const uint2 r = *reinterpret_cast<const uint2*>(src + (size_t)n * sbs + o);
      v[0] = bits_f<T>((unsigned short)r.x); v[1] = bits_f<T>((unsigned short)(r.x >> 16));
      v[2] = bits_f<T>((unsigned short)r.y); v[3] = bits_f<T>((unsigned short)(r.y >> 16));
      if (add) {
        const uint2 a = *reinterpret_cast<const uint2*>(add + (size_t)n * abs_ + o);
        // 16-bit sum, rounded like the tensor add it replaces
        v[0] = round16<T>(v[0] + bits_f<T>((unsigned short)a.x)); v[1] = round16<T>(v[1] + bits_f<T>((unsigned short)(a.x >> 16)));
        v[2] = round16<T>(v[2] + bits_f<T>((unsigned short)a.y)); v[3] = round16<T>(v[3] + bits_f<T>((unsigned short)(a.y >> 16)));
      }
      if (y) {
        const uint2 q = *reinterpret_cast<const uint2*>(y + (size_t)n * ybs + o);
        if (!(bits_f<T>((unsigned short)q.x) > 0.f)) v[0] *= slope;
        if (!(bits_f<T>((unsigned short)(q.x >> 16)) > 0.f)) v[1] *= slope;
        if (!(bits_f<T>((unsigned short)q.y) > 0.f)) v[2] *= slope;
        if (!(bits_f<T>((unsigned short)(q.y >> 16)) > 0.f)) v[3] *= slope;
      }
      uint2 w;
      w.x = pack2<T>(v[0], v[1]); w.y = pack2<T>(v[2], v[3]);
      if (dst) *reinterpret_cast<uint2*>(dst + (size_t)n * dbs + o) = w;
      s += (bits_f<T>((unsigned short)w.x) + bits_f<T>((unsigned short)(w.x >> 16))) +
           (bits_f<T>((unsigned short)w.y) + bits_f<T>((unsigned short)(w.y >> 16)));
    } else {
      v[0] = Elem<T>::load(src + (size_t)n * sbs + o);
      if (add) v[0] = round16<T>(v[0] + Elem<T>::load(add + (size_t)n * abs_ + o));
      if (y && !(Elem<T>::load(y + (size_t)n * ybs + o) > 0.f)) v[0] *= slope;
      if (dst) Elem<T>::store(dst + (size_t)n * dbs + o, v[0]);
      s += round16<T>(v[0]);
    }
  }
  if (!part) return;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    float t = (sh[0] + sh[1]) + (sh[2] + sh[3]);
#pragma unroll
    for (int k = 4; k < NTH / 64; k += 4) t += (sh[k] + sh[k + 1]) + (sh[k + 2] + sh[k + 3]);
    part[co * BIAS_NCH + chunk] = t;
  }
}

__global__ void bias_grad_multi_final_kernel(const BiasParts P, int Cout) { bias_finish(P, blockIdx.x * blockDim.x + threadIdx.x, Cout); }

// K-splits = 8 XCDs x J: J concurrent tiles per XCD such that J x (block pairs) workgroups fill its 32 CUs (the kernel
// keeps a whole tile in registers: 1 workgroup per CU is resident), bounded by the tiles an XCD has and by 40 MB of fp32
// partial blocks per 64 output channels of a workgroup.
// kernel choice: 0 = producer / consumer kernel (64 co x 64 ci), 1 = wgrad_kernel (64 co, or 128 co for Cout > 64);
// UPF_WGRAD_MODE overrides (A/B runs)
static int wgrad_mode() {
  static const int m = [] { const char* e = getenv("UPF_WGRAD_MODE"); return e ? atoi(e) : 0; }();
  return m;
}
static int co_block(int Cout) { return (wgrad_mode() == 1 && Cout > 64) ? 128 : 64; }
static int pick_ksplit(int ntiles, int nblocks, int ntaps, int cob) {
  int J = 32 / nblocks;
  const long long per_split = (long long)nblocks * ntaps * cob * 64 * 4;
  const int cap = (int)(((40ll << 20) * (cob / 64)) / (8 * per_split));
  if (J > cap) J = cap;
  if (J > (ntiles + 7) / 8) J = (ntiles + 7) / 8;
  return 8 * (J < 1 ? 1 : J);
}

// One group of levels (all aligned, or all through the RAGGED staging) -> `ksplit` partial blocks starting at `ws`.
template <typename T, int D, bool RAGGED, int MB>
void launch_group(const upf_wgrad_level* lv, const int* idx, int n, float* ws, int ksplit, int Cin, int Cout, hipStream_t stream) {
  using G = Geo<D>;
  constexpr int DD = (D == 0) ? 1 : D;
  KLevels L;
  memset(&L, 0, sizeof(L));
  int t0 = 0;
  for (int i = 0; i < n; ++i) {
    const upf_wgrad_level& a = lv[idx[i]];
    KLevel& k = L.lv[i];
    k.x = a.x; k.g = a.grad_pre;
    k.xbs = a.x_batch_stride ? a.x_batch_stride : (long long)Cin * a.H * a.W;
    k.gbs = a.g_batch_stride ? a.g_batch_stride : (long long)Cout * a.H * a.W;
    k.H = a.H; k.W = a.W;
    k.tiles_x = cdiv(a.W, TWP); k.tiles_y = DD * cdiv(cdiv(a.H, DD), TR);
    k.tile0 = t0;
    t0 += a.B * k.tiles_x * k.tiles_y;
  }
  L.n = n; L.ntiles = t0;
  const int nco2 = cdiv(Cout, 64 * MB), nci2 = cdiv(Cin, 64);
  constexpr int lds_bytes = (G::X_BLOCKS + MB * G::G_BLOCKS) * 16;
  static LdsOptIn opt;
  auto kern = &wgrad_kernel<T, D, RAGGED, MB>;
  opt.ensure(reinterpret_cast<const void*>(kern), lds_bytes);
  hipLaunchKernelGGL(kern, dim3(ksplit * nco2 * nci2), dim3(NTHREADS * MB), lds_bytes, stream, L, ws, Cin, Cout, nci2, ksplit / 8);
}

static bool level_ragged(const upf_wgrad_level& a, int Cin, int Cout) {
  const long long xbs = a.x_batch_stride ? a.x_batch_stride : (long long)Cin * a.H * a.W;
  const long long gbs = a.g_batch_stride ? a.g_batch_stride : (long long)Cout * a.H * a.W;
  return !(a.W % 8 == 0 && xbs % 8 == 0 && gbs % 8 == 0 && aligned_to(a.x, 16) && aligned_to(a.grad_pre, 16));
}
static int level_tiles(const upf_wgrad_level& a, int dd) { return a.B * cdiv(a.W, TWP) * dd * cdiv(cdiv(a.H, dd), TR); }

// More than 16 block pairs cannot run two tile lanes per XCD (32 CUs): a 567 -> 128 layer (18 pairs) would leave 14 of 32
// CUs idle.  Its co blocks then go in separate launches (9 pairs, 3 lanes, 27 CUs per XCD); X is read once per launch.
static bool pc_split_co(int nco2, int nci2) { return nco2 > 1 && nco2 * nci2 > 16 && nci2 <= 16; }

template <typename T, int D, bool RAGGED>
void launch_group_pc(const upf_wgrad_level* lv, const int* idx, int n, float* ws, int ns, int J, bool split_co, int Cin, int Cout, hipStream_t stream) {
  using G = Geo<D>;
  constexpr int DD = (D == 0) ? 1 : D;
  KLevels L;
  memset(&L, 0, sizeof(L));
  int t0 = 0;
  for (int i = 0; i < n; ++i) {
    const upf_wgrad_level& a = lv[idx[i]];
    KLevel& k = L.lv[i];
    k.x = a.x; k.g = a.grad_pre;
    k.xbs = a.x_batch_stride ? a.x_batch_stride : (long long)Cin * a.H * a.W;
    k.gbs = a.g_batch_stride ? a.g_batch_stride : (long long)Cout * a.H * a.W;
    k.H = a.H; k.W = a.W;
    k.tiles_x = cdiv(a.W, TWP); k.tiles_y = DD * cdiv(cdiv(a.H, DD), TR);
    k.tile0 = t0;
    k.ragged = level_ragged(a, Cin, Cout) ? 1 : 0;
    t0 += a.B * k.tiles_x * k.tiles_y;
  }
  L.n = n; L.ntiles = t0;
  const int nco2 = cdiv(Cout, 64), nci2 = cdiv(Cin, 64);
  constexpr int lds_bytes = 2 * (G::X_BLOCKS + G::G_BLOCKS) * 16 + MAXL * LT_STRIDE * 4;      // two staging buffers + the level table
  static LdsOptIn opt;
  auto kern = &wgrad_pc_kernel<T, D, RAGGED>;
  opt.ensure(reinterpret_cast<const void*>(kern), lds_bytes);
  static const int abl = [] { const char* e = getenv("UPF_WGRAD_ABLATE"); return e ? atoi(e) << 16 : 0; }();
  const int ksplit = ns * J, J_ns = J | (ns << 16);
  if (split_co) {                                // one launch per co block (see pc_split_co)
    for (int c = 0; c < nco2; ++c)
      hipLaunchKernelGGL(kern, dim3(ksplit * nci2), dim3(2 * NTHREADS), lds_bytes, stream, L, ws, Cin, Cout, nci2, J_ns, c | abl);
  } else {
    hipLaunchKernelGGL(kern, dim3(ksplit * nco2 * nci2), dim3(2 * NTHREADS), lds_bytes, stream, L, ws, Cin, Cout, nci2, J_ns, abl);
  }
}


// The plan of one multi-level weight gradient.  Producer / consumer kernel (the default): ALL levels in one launch — one K
// dimension, one set of partial blocks — with the shifted staging chosen per tile (`mixed`: some level is ragged).
// wgrad_kernel (UPF_WGRAD_MODE=1, kept for A/B runs): aligned levels in one launch, ragged ones in a second, their K-splits
// back to back in the workspace.  One reduction over all partial blocks either way.
struct Plan {
  int ia[MAXL], na = 0, ir[MAXL], nr = 0, ks_a = 0, ks_r = 0;
  bool mixed = false;
  int ns = 8, J = 1;                    // producer / consumer kernel: ks_a = ns * J
  bool split_co = false;
};

// Producer / consumer kernel: how many slices of the tile range (ns) and workgroups per block pair and slice (J).  Every
// workgroup ends by writing its 64 x 64 x taps fp32 block, which the reduction reads back: with the chip filled whatever the
// layer (round 4: ns = 8, J = 32 / block pairs) a 96 -> 96 layer at 16x52 — 64 tiles — wrote and re-read 38 MB for 1 GFLOP, and
// the twelve single-level layers of the feature pyramid took 25-80 us each.  Cost model (us): tiles per workgroup x the
// measured tile time + the partial blocks both ways at ~3.5 TB/s; UPF_WGRAD_SPLIT="ns,J" overrides (A/B runs).
static void pick_split(int ntiles, int nb_launch, int nb_total, int ntaps, int& ns_out, int& J_out) {
  static const char* ov = getenv("UPF_WGRAD_SPLIT");
  if (ov) { int a = 8, b = 1; if (sscanf(ov, "%d,%d", &a, &b) == 2 && a >= 1 && b >= 1) { ns_out = a; J_out = b; return; } }
  const double tile_us = ntaps == 9 ? 1.8 : 0.5, block_us = (double)ntaps * 64 * 64 * 4 * 2 / 3.5e6;
  double best = 1e30;
  for (int ns = 8; ns >= 1; ns >>= 1) {
    if (ns < 8 && ntiles > 64) break;                                    // (fewer slices than XCDs only where K is a handful of tiles)
    for (int J = 1; J <= 32; ++J) {
      if (ns * J * nb_launch > 256 && J > 1) break;                      // one workgroup per CU
      const int tiles_wg = cdiv(cdiv(ntiles, ns), J);
      const double waves = (double)cdiv(ns * J * nb_launch, 256);
      const double t = waves * tiles_wg * tile_us + (double)ns * J * nb_total * block_us + (tiles_wg * J * ns > ntiles + ns * J ? 0.5 : 0.0);
      if (t < best - 1e-9) { best = t; ns_out = ns; J_out = J; }
    }
  }
}
static Plan make_plan(const upf_wgrad_level* lv, int n, int Cin, int Cout, int kernel_size, int dilation) {
  Plan p;
  const int dd = kernel_size == 1 ? 1 : dilation, nt = kernel_size == 1 ? 1 : 9;
  int ta = 0, tr = 0;
  for (int i = 0; i < n; ++i) {
    const bool rag = level_ragged(lv[i], Cin, Cout);
    if (rag && wgrad_mode() != 0) { p.ir[p.nr++] = i; tr += level_tiles(lv[i], dd); }
    else { p.ia[p.na++] = i; ta += level_tiles(lv[i], dd); p.mixed = p.mixed || rag; }
  }
  const int cob = co_block(Cout);
  int nblocks = cdiv(Cout, cob) * cdiv(Cin, 64);
  if (wgrad_mode() == 0) {
    p.split_co = pc_split_co(cdiv(Cout, 64), cdiv(Cin, 64)) && ta >= 256;
    pick_split(ta, p.split_co ? cdiv(Cin, 64) : nblocks, nblocks, nt, p.ns, p.J);
    p.ks_a = p.ns * p.J;
    return p;
  }
  if (p.na) p.ks_a = pick_ksplit(ta, nblocks, nt, cob);
  if (p.nr) p.ks_r = pick_ksplit(tr, nblocks, nt, cob);
  return p;
}

template <typename T, int D>
int run(const upf_wgrad_level* lv, int n, float* dw, float* ws, int Cin, int Cout, int kernel_size, int dilation, hipStream_t stream, int s2d = 0,
        const BiasParts* bias = nullptr) {
  using G = Geo<D>;
  const Plan p = make_plan(lv, n, Cin, Cout, kernel_size, dilation);
  const int cob = co_block(Cout), cop = cdiv(Cout, cob) * cob;
  const size_t per_split = (size_t)G::NT * cop * (cdiv(Cin, 64) * 64);
  if (wgrad_mode() == 0) {
    if (p.mixed) launch_group_pc<T, D, true>(lv, p.ia, p.na, ws, p.ns, p.J, p.split_co, Cin, Cout, stream);
    else launch_group_pc<T, D, false>(lv, p.ia, p.na, ws, p.ns, p.J, p.split_co, Cin, Cout, stream);
  } else if (cob == 128) {
    if (p.na) launch_group<T, D, false, 2>(lv, p.ia, p.na, ws, p.ks_a, Cin, Cout, stream);
    if (p.nr) launch_group<T, D, true, 2>(lv, p.ir, p.nr, ws + (size_t)p.ks_a * per_split, p.ks_r, Cin, Cout, stream);
  } else {
    if (p.na) launch_group<T, D, false, 1>(lv, p.ia, p.na, ws, p.ks_a, Cin, Cout, stream);
    if (p.nr) launch_group<T, D, true, 1>(lv, p.ir, p.nr, ws + (size_t)p.ks_a * per_split, p.ks_r, Cin, Cout, stream);
  }
  const long long items = (long long)Cout * cdiv(Cin, 4);
  const int ks = p.ks_a + p.ks_r;
  BiasParts BP;
  memset(&BP, 0, sizeof(BP));
  if (bias) BP = *bias;
  const int nbias = bias ? cdiv(Cout, 256) : 0;
  if (items <= 64 * 128 && ks >= 16) {      // narrow layers: 16 slices per item (see the kernel)
    const int nw = (int)((items + 15) / 16);
    hipLaunchKernelGGL((wgrad_reduce_kernel<G::NT, 16>), dim3((unsigned)(nw + nbias)), dim3(256), 0, stream, ws, dw, ks, Cout, Cin, cop, s2d, nw, BP);
  } else {
    const int nw = (int)((items + 63) / 64);
    hipLaunchKernelGGL((wgrad_reduce_kernel<G::NT, 4>), dim3((unsigned)(nw + nbias)), dim3(256), 0, stream, ws, dw, ks, Cout, Cin, cop, s2d, nw, BP);
  }
  return check_launch("conv_wgrad");
}

}  // namespace wgrad
}  // namespace upf

extern "C" int upf_conv_wgrad_supported(int Cin, int Cout, int H, int W, int kernel_size, int dilation, int stride, int dtype) {
  return (dtype == UPF_F16 || dtype == UPF_BF16) && stride == 1 && W >= 8 && Cin > 0 && Cout > 0 && H > 0 &&
         ((kernel_size == 1 && dilation == 1) || (kernel_size == 3 && (dilation == 1 || dilation == 2 || dilation == 4 || dilation == 8 || dilation == 16)));
}

static int wgrad_check_levels(const upf_wgrad_level* lv, int n, int Cin, int Cout, int kernel_size, int dilation, int dtype) {
  using namespace upf;
  UPF_REQUIRE(lv && n >= 1 && n <= wgrad::MAXL, UPF_EINVAL, "conv_wgrad: 1..%d levels", wgrad::MAXL);
  for (int i = 0; i < n; ++i) {
    const upf_wgrad_level& a = lv[i];
    UPF_REQUIRE(a.x && a.grad_pre && a.B > 0, UPF_EINVAL, "conv_wgrad: null pointer / empty batch (level %d)", i);
    UPF_REQUIRE(upf_conv_wgrad_supported(Cin, Cout, a.H, a.W, kernel_size, dilation, 1, dtype), UPF_EUNSUPPORTED,
                "conv_wgrad: bf16 / fp16, stride 1, W >= 8, 1x1 or 3x3 with dilation 1/2/4/8/16 only (k %d, d %d, W %d)", kernel_size, dilation, a.W);
    const long long xbs = a.x_batch_stride ? a.x_batch_stride : (long long)Cin * a.H * a.W, gbs = a.g_batch_stride ? a.g_batch_stride : (long long)Cout * a.H * a.W;
    UPF_REQUIRE(xbs >= (long long)Cin * a.H * a.W && gbs >= (long long)Cout * a.H * a.W, UPF_EINVAL, "conv_wgrad: bad batch stride");
    UPF_REQUIRE((size_t)64 * a.H * a.W * 2 < (1ull << 31), UPF_EUNSUPPORTED, "conv_wgrad: image too large");
  }
  return UPF_OK;
}

extern "C" long long upf_conv_wgrad_multi_workspace_bytes(const upf_wgrad_level* levels, int nlevels, int Cin, int Cout, int kernel_size, int dilation) {
  using namespace upf;
  if (!levels || nlevels < 1 || nlevels > wgrad::MAXL) return -1;
  const int nt = kernel_size == 1 ? 1 : 9;
  const wgrad::Plan p = wgrad::make_plan(levels, nlevels, Cin, Cout, kernel_size, kernel_size == 1 ? 1 : dilation);
  const int cob = wgrad::co_block(Cout);
  return (long long)(p.ks_a + p.ks_r) * nt * (cdiv(Cout, cob) * cob) * (cdiv(Cin, 64) * 64) * (long long)sizeof(float);
}

extern "C" int upf_conv_wgrad_multi_bias(const upf_wgrad_level* levels, int nlevels, float* grad_w, void* workspace, int Cin, int Cout,
                                         int kernel_size, int dilation, const float* const* bias_partials, int npartials, float* grad_bias,
                                         int dtype, void* stream) {
  using namespace upf;
  UPF_REQUIRE(grad_w && workspace, UPF_EINVAL, "conv_wgrad: null pointer");
  if (int rc = wgrad_check_levels(levels, nlevels, Cin, Cout, kernel_size, dilation, dtype)) return rc;
  wgrad::BiasParts BP;
  memset(&BP, 0, sizeof(BP));
  if (npartials) {
    UPF_REQUIRE(bias_partials && grad_bias && npartials >= 1 && npartials <= wgrad::MAXP, UPF_EINVAL, "conv_wgrad_multi_bias: 1..%d bias partial buffers and grad_bias", wgrad::MAXP);
    for (int i = 0; i < npartials; ++i) {
      UPF_REQUIRE(bias_partials[i], UPF_EINVAL, "conv_wgrad_multi_bias: null partial buffer");
      BP.p[i] = bias_partials[i];
    }
    BP.n = npartials; BP.db = grad_bias;
  }
  const wgrad::BiasParts* bp = npartials ? &BP : nullptr;
  hipStream_t s = (hipStream_t)stream;
  const int D = kernel_size == 1 ? 0 : dilation;
#define UPF_WG(DV) case DV: return dtype == UPF_BF16 ? wgrad::run<bf16_t, DV>(levels, nlevels, grad_w, (float*)workspace, Cin, Cout, kernel_size, dilation, s, 0, bp) \
                                                     : wgrad::run<f16_t, DV>(levels, nlevels, grad_w, (float*)workspace, Cin, Cout, kernel_size, dilation, s, 0, bp);
  switch (D) {
    UPF_WG(0) UPF_WG(1) UPF_WG(2) UPF_WG(4) UPF_WG(8) UPF_WG(16)
  }
#undef UPF_WG
  set_error("conv_wgrad: internal routing error");
  return UPF_EUNSUPPORTED;
}

extern "C" int upf_conv_wgrad_multi(const upf_wgrad_level* levels, int nlevels, float* grad_w, void* workspace, int Cin, int Cout,
                                    int kernel_size, int dilation, int dtype, void* stream) {
  return upf_conv_wgrad_multi_bias(levels, nlevels, grad_w, workspace, Cin, Cout, kernel_size, dilation, nullptr, 0, nullptr, dtype, stream);
}

extern "C" int upf_conv_wgrad_s2d(const upf_wgrad_level* levels, int nlevels, float* grad_w, void* workspace, int Cin, int Cout, int dtype, void* stream) {
  using namespace upf;
  UPF_REQUIRE(grad_w && workspace, UPF_EINVAL, "conv_wgrad_s2d: null pointer");
  if (int rc = wgrad_check_levels(levels, nlevels, 4 * Cin, Cout, 3, 1, dtype)) return rc;
  hipStream_t s = (hipStream_t)stream;
  return dtype == UPF_BF16 ? wgrad::run<bf16_t, 1>(levels, nlevels, grad_w, (float*)workspace, 4 * Cin, Cout, 3, 1, s, 1)
                           : wgrad::run<f16_t, 1>(levels, nlevels, grad_w, (float*)workspace, 4 * Cin, Cout, 3, 1, s, 1);
}

extern "C" long long upf_conv_wgrad_workspace_bytes(int B, int Cin, int Cout, int H, int W, int kernel_size, int dilation) {
  // (alignment unknown here: aligned and ragged plans of ONE level use the same split count)
  upf_wgrad_level a = {nullptr, 0, nullptr, 0, B, H, W};
  return upf_conv_wgrad_multi_workspace_bytes(&a, 1, Cin, Cout, kernel_size, dilation);
}

extern "C" int upf_conv_wgrad(const void* x, long long x_batch_stride, const void* grad_y, long long g_batch_stride, float* grad_w,
                              void* workspace, int B, int Cin, int Cout, int H, int W, int kernel_size, int dilation, int dtype, void* stream) {
  const upf_wgrad_level a = {x, x_batch_stride, grad_y, g_batch_stride, B, H, W};
  return upf_conv_wgrad_multi(&a, 1, grad_w, workspace, Cin, Cout, kernel_size, dilation, dtype, stream);
}

extern "C" int upf_leaky_backward(const void* grad_y, const void* y, void* grad_pre, long long n, float slope, int dtype, void* stream) {
  using namespace upf;
  UPF_REQUIRE(grad_y && y && grad_pre && n > 0 && n % 8 == 0, UPF_EINVAL, "leaky_backward: null pointer or element count not a multiple of 8");
  UPF_REQUIRE(dtype == UPF_F16 || dtype == UPF_BF16, UPF_EDTYPE, "leaky_backward: bf16 / fp16 only");
  UPF_REQUIRE(aligned_to(grad_y, 16) && aligned_to(y, 16) && aligned_to(grad_pre, 16), UPF_EALIGN, "leaky_backward: 16-byte alignment");
  const long long n8 = n / 8;
  UPF_REQUIRE((n8 + 255) / 256 < (1ll << 31), UPF_EINVAL, "leaky_backward: tensor too large");
  const dim3 grid((unsigned)((n8 + 255) / 256));
  if (dtype == UPF_BF16) hipLaunchKernelGGL((wgrad::leaky_bwd_kernel<bf16_t>), grid, dim3(256), 0, (hipStream_t)stream, (const bf16_t*)grad_y, (const bf16_t*)y, (bf16_t*)grad_pre, n8, slope);
  else hipLaunchKernelGGL((wgrad::leaky_bwd_kernel<f16_t>), grid, dim3(256), 0, (hipStream_t)stream, (const f16_t*)grad_y, (const f16_t*)y, (f16_t*)grad_pre, n8, slope);
  return check_launch("leaky_backward");
}

extern "C" int upf_act_grad(const void* src, long long src_batch_stride, const void* add, long long add_batch_stride, const void* y,
                            long long y_batch_stride, void* dst, long long dst_batch_stride, float* bias_partial, int B, int C, int HW,
                            float slope, int dtype, void* stream) {
  using namespace upf;
  UPF_REQUIRE(src && (dst || bias_partial) && B > 0 && C > 0 && HW > 0, UPF_EINVAL, "act_grad: bad arguments");
  UPF_REQUIRE(dtype == UPF_F16 || dtype == UPF_BF16, UPF_EDTYPE, "act_grad: bf16 / fp16 only");
  const long long chw = (long long)C * HW;
  const long long sbs = src_batch_stride ? src_batch_stride : chw, abs_ = add_batch_stride ? add_batch_stride : chw;
  const long long ybs = y_batch_stride ? y_batch_stride : chw, dbs = dst_batch_stride ? dst_batch_stride : chw;
  const bool v4 = HW % 4 == 0 && sbs % 4 == 0 && abs_ % 4 == 0 && ybs % 4 == 0 && dbs % 4 == 0 && aligned_to(src, 8) && (!dst || aligned_to(dst, 8)) &&
                  (!add || aligned_to(add, 8)) && (!y || aligned_to(y, 8));
  hipStream_t s = (hipStream_t)stream;
  const dim3 grid(C, wgrad::BIAS_NCH);
  // few channels, long chunks: 1024 threads per workgroup (see the kernel)
  const bool wide = v4 && C * wgrad::BIAS_NCH <= 1024 && (long long)B * (HW / 4) / wgrad::BIAS_NCH >= 2048;
#define UPF_AG(TT, VV) hipLaunchKernelGGL((wgrad::act_grad_kernel<TT, VV>), grid, dim3(256), 0, s, (const TT*)src, sbs, (const TT*)add, abs_, (const TT*)y, ybs, (TT*)dst, dbs, bias_partial, B, HW, slope)
#define UPF_AGW(TT) hipLaunchKernelGGL((wgrad::act_grad_kernel<TT, 4, 1024>), grid, dim3(1024), 0, s, (const TT*)src, sbs, (const TT*)add, abs_, (const TT*)y, ybs, (TT*)dst, dbs, bias_partial, B, HW, slope)
  if (dtype == UPF_BF16) { if (wide) UPF_AGW(bf16_t); else if (v4) UPF_AG(bf16_t, 4); else UPF_AG(bf16_t, 1); }
  else { if (wide) UPF_AGW(f16_t); else if (v4) UPF_AG(f16_t, 4); else UPF_AG(f16_t, 1); }
#undef UPF_AG
#undef UPF_AGW
  return check_launch("act_grad");
}

extern "C" int upf_conv_bias_grad_finish(const float* const* partials, int npartials, float* grad_bias, int Cout, void* stream) {
  using namespace upf;
  UPF_REQUIRE(partials && grad_bias && Cout > 0 && npartials >= 1 && npartials <= wgrad::MAXP, UPF_EINVAL, "conv_bias_grad_finish: 1..%d partial buffers", wgrad::MAXP);
  wgrad::BiasParts P;
  memset(&P, 0, sizeof(P));
  for (int i = 0; i < npartials; ++i) {
    UPF_REQUIRE(partials[i], UPF_EINVAL, "conv_bias_grad_finish: null partial buffer");
    P.p[i] = partials[i];
  }
  P.n = npartials; P.db = grad_bias;
  hipLaunchKernelGGL(wgrad::bias_grad_multi_final_kernel, dim3(cdiv(Cout, 128)), dim3(128), 0, (hipStream_t)stream, P, Cout);
  return check_launch("conv_bias_grad_finish");
}

extern "C" long long upf_conv_bias_grad_workspace_bytes(int Cout) { return (long long)Cout * upf::wgrad::BIAS_NCH * (long long)sizeof(float); }

extern "C" int upf_conv_bias_grad(const void* grad_y, long long g_batch_stride, float* grad_bias, void* workspace, int B, int Cout, int HW,
                                  int dtype, void* stream) {
  using namespace upf;
  UPF_REQUIRE(grad_y && grad_bias && workspace && B > 0 && Cout > 0 && HW > 0, UPF_EINVAL, "conv_bias_grad: bad arguments");
  const long long gbs = g_batch_stride ? g_batch_stride : (long long)Cout * HW;
  hipStream_t s = (hipStream_t)stream;
  UPF_DISPATCH(dtype, T, hipLaunchKernelGGL((wgrad::bias_grad_partial_kernel<T>), dim3(Cout, wgrad::BIAS_NCH), dim3(256), 0, s, (const T*)grad_y, gbs, (float*)workspace, B, HW));
  hipLaunchKernelGGL(wgrad::bias_grad_final_kernel, dim3(cdiv(Cout, 128)), dim3(128), 0, s, (const float*)workspace, grad_bias, Cout);
  return check_launch("conv_bias_grad");
}
