// 81-neighbour cost volume, forward, bf16 / fp16 — matrix-core variant for gfx950.
//
// Why a second kernel: the wave-level dot2 kernel (corr81_fwd_kernel.hpp) is VALU-bound —
// v_dot2c_f32_bf16 issues at half rate (measured 4 cycles per wave64) and every output needs C/2 of
// them, which costs more than moving the data.  The per-pixel contraction over channels IS a small
// matrix product, and gfx950 still carries the 16-block 4x4x4 MFMA:
//     v_mfma_f32_4x4x4_16b_{bf16,f16}: 16 independent blocks  D_b[4x4] += A_b[4x4(k)] * B_b[4(k)x4]
// (measured 8 cycles per instruction = 4.4x the MAC rate of v_dot2c).  One instruction handles 16
// pixel-quads: block b multiplies 4 candidate pixels of f2 (rows of A) with 4 pixels of f1 (columns of
// B) over 4 channels.  For a pixel quad at columns x..x+3 and displacement row dy the 9 horizontal
// displacements live in the 12 candidates x-4..x+7, i.e. three candidate quads q=0,1,2: 36 of the 48
// products are wanted (75 %) — this is NOT a reshaping of byte work into a GEMM, the band is dense.
//
// Mapping (verified on hardware by tools/mfma_probe.hip):
//     A: lane l -> block l/4, row  i=l%4, holds k=0..3      B: lane l -> block l/4, col j=l%4, k=0..3
//     D: lane l, reg r -> block l/4, D[i=r][j=l%4]
// so with A = f2 and B = f1 every lane owns ONE pixel (j) and its 4 registers are 4 candidates.
//
// Structure
//   * workgroup = 9 waves = one 8x32 pixel tile; wave w owns displacement row dy = w-4 (like the VALU
//     kernel) and walks 4 "units": unit = 2 image rows x 32 pixels = the 16 blocks of one MFMA;
//   * LDS holds, for a chunk of 8 channel QUADS (32 channels), the f1 tile and the f2 tile + halo with
//     the 4 channels of a pixel interleaved into one 8-byte entry (what an A/B operand lane needs):
//     the NCHW -> channel-quad layout change happens once, in registers, on the way into LDS
//     (4 buffer loads + 8 v_perm + 2 ds_write_b128 per 4 pixels x 4 channels; halo/tail zeros come from
//     the buffer descriptor's bounds check);
//   * per unit and channel quad: 4 ds_read_b64 + 3 MFMA; a unit's 12 accumulators are complete after
//     C/4 steps, so with C <= 32 (the large 1/4-resolution level) each unit is finished, de-skewed and
//     stored while the next one computes: HBM writes overlap the matrix work instead of forming a tail;
//   * epilogue: lane p=l%4 needs candidates p..p+8 of its 12 -> a two-stage barrel shift by the lane's
//     2-bit position (20 v_cndmask), scale by 1/C, LeakyReLU, convert; the 9x64 results are transposed
//     through a 1152-byte per-wave LDS patch so that they leave as 16-byte stores (8 pixels of one
//     output channel per lane).
#pragma once
#include "common.hpp"

namespace upf {
namespace corrm {

constexpr int R = 4, D = 9, TH = 8, TW = 32;
constexpr int KQ = 8;                        // channel quads per LDS chunk (32 channels)
constexpr int F2W = TW + 2 * R;              // 40
constexpr int F2H = TH + 2 * R;              // 16
constexpr int F1_E = TH * TW;                // 8-byte entries per channel quad, f1 tile   (256)
constexpr int F2_E = F2H * F2W;              // f2 tile incl. halo                          (640)
constexpr int CHUNK_E = KQ * (F1_E + F2_E);  // 7168 entries = 57,344 B
constexpr int NWAVES = D, NTHREADS = NWAVES * 64;
constexpr int PATCH_BYTES = D * 64 * 2;      // per-wave output transposition patch (1152 B)
constexpr int LDS_BYTES = CHUNK_E * 8 + NWAVES * PATCH_BYTES;          // 67,712 B -> two workgroups per CU
constexpr int N1_TASKS = KQ * TH * (TW / 4);                           // 512 (8 full waves)
constexpr int N2_TASKS = KQ * F2H * (F2W / 4);                         // 1280
constexpr int TASKS_PER_THREAD = (N1_TASKS + N2_TASKS + NTHREADS - 1) / NTHREADS;   // 4

typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(4))) _Float16 f16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;

template <typename T> struct Mma;
template <> struct Mma<bf16_t> {
  static __device__ __forceinline__ f32x4 mma(uint2 a, uint2 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(__builtin_bit_cast(s16x4, a), __builtin_bit_cast(s16x4, b), c, 0, 0, 0);
  }
};
template <> struct Mma<f16_t> {
  static __device__ __forceinline__ f32x4 mma(uint2 a, uint2 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_4x4x4f16(__builtin_bit_cast(f16x4, a), __builtin_bit_cast(f16x4, b), c, 0, 0, 0);
  }
};

struct Task {
  int lds;         // 8-byte entry index of the first of 4 pixels; -1 = no task
  uint32_t voff;   // byte offset of (channel quad kq, row gy, column gx) inside the batch item; 0x80000000 = outside
};

__device__ __forceinline__ Task make_task(int t, int y0, int x0, int H, int W) {
  Task s;
  s.lds = -1; s.voff = 0x80000000u;
  if (t >= N1_TASKS + N2_TASKS) return s;
  int kq, gy, gx;
  if (t < N1_TASKS) {
    kq = t / (TH * 8);
    const int rem = t - kq * (TH * 8), r = rem >> 3, g = rem & 7;
    gy = y0 + r; gx = x0 + 4 * g;
    s.lds = kq * F1_E + r * TW + 4 * g;
  } else {
    const int u = t - N1_TASKS;
    kq = u / (F2H * 10);
    const int rem = u - kq * (F2H * 10), r = rem / 10, g = rem - r * 10;
    gy = y0 - R + r; gx = x0 - R + 4 * g;
    s.lds = KQ * F1_E + kq * F2_E + r * F2W + 4 * g;
  }
  if (gy >= 0 && gy < H && gx >= 0 && gx < W) s.voff = (uint32_t)((kq * 4 * H + gy) * W + gx) * 2u;
  return s;
}

// out: [B,81,H,W] (batch stride out_bs); requires W % 8 == 0, 16-byte aligned pointers, item < 2 GiB.
// SINGLE: C <= 32, all channels in one LDS chunk -> a unit is stored as soon as it is complete.
template <typename T, bool SINGLE, int ABL = 0>
__global__ __launch_bounds__(NTHREADS, 5)
void corr81_mfma_kernel(const T* __restrict__ f1, const T* __restrict__ f2, T* __restrict__ out,
                        int C, int H, int W, int tiles_x, int tiles_y, long long out_bs, float slope) {
  extern __shared__ __attribute__((aligned(16))) uint2 lds[];          // CHUNK_E entries, then the patches
  const int bid = xcd_remap(blockIdx.x, gridDim.x);
  const int tx = bid % tiles_x;
  const int ty = (bid / tiles_x) % tiles_y;
  const int n = bid / (tiles_x * tiles_y);
  const int x0 = tx * TW, y0 = ty * TH;
  const int tid = threadIdx.x, lane = tid & 63;
  const int dyi = __builtin_amdgcn_readfirstlane(tid >> 6);            // 0..8 <-> dy = dyi-4

  const size_t item = (size_t)n * C * H * W;
  const uint32_t plane = (uint32_t)H * (uint32_t)W * 2u;               // bytes per channel plane
  const uint32_t item_bytes = (uint32_t)C * plane;
  Task task[TASKS_PER_THREAD];
  __amdgpu_buffer_rsrc_t rsrc[TASKS_PER_THREAD];
#pragma unroll
  for (int j = 0; j < TASKS_PER_THREAD; ++j) {
    task[j] = make_task(tid + j * NTHREADS, y0, x0, H, W);
    const bool from_f2 = __builtin_amdgcn_readfirstlane((tid & ~63) + j * NTHREADS) >= N1_TASKS;   // wave-uniform
    rsrc[j] = __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>((from_f2 ? f2 : f1) + item), 0, item_bytes, 0x00020000);
  }

  // operand addresses of this lane: block b = lane/4 -> (row within the pair, pixel quad), j = lane%4
  const int rsel = lane >> 5, pix = lane & 31;                          // pix = 4*quad + j
  const int p = lane & 3;
  const uint32_t m1 = (p & 1) ? 0xffffffffu : 0u, m2 = (p & 2) ? 0xffffffffu : 0u;   // per-lane select masks
  const int nquads = (C + 3) >> 2;
  const int nchunks = (nquads + KQ - 1) / KQ;
  const float invC = 1.0f / (float)C;
  uint16_t* patch = reinterpret_cast<uint16_t*>(lds + CHUNK_E) + (tid >> 6) * (PATCH_BYTES / 2);
  using st = uint16_t;
  st* obase = reinterpret_cast<st*>(out) + (size_t)n * out_bs + (size_t)(dyi * D) * H * W;

  constexpr int NACC = SINGLE ? 1 : 4;          // SINGLE: the 12 accumulators are recycled unit by unit
  f32x4 acc[NACC][3];
#pragma unroll
  for (int u = 0; u < NACC; ++u)
#pragma unroll
    for (int q = 0; q < 3; ++q) acc[u][q] = f32x4{0.f, 0.f, 0.f, 0.f};

  auto epilogue = [&](int u, f32x4 a0, f32x4 a1, f32x4 a2) {
    // candidates c = 0..11 of this lane's pixel; displacement t = dx+4 is candidate t + p
    // Two-stage barrel shift by the lane's 2-bit pixel position.  Written as bit-selects (v_bfi_b32) on
    // scalars: given float arrays and `m ? C[c+1] : C[c]`, hipcc turns the select into a lane-indexed
    // load from a SCRATCH copy of the array (148 scratch ops in this loop).
    auto sel = [](uint32_t mask, float a, float b) {
      return __uint_as_float((__float_as_uint(a) & mask) | (__float_as_uint(b) & ~mask));
    };
    const float c0 = a0[0], c1 = a0[1], c2 = a0[2], c3 = a0[3], c4 = a1[0], c5 = a1[1], c6 = a1[2], c7 = a1[3];
    const float c8 = a2[0], c9 = a2[1], c10 = a2[2], c11 = a2[3];
    const float t0 = sel(m1, c1, c0), t1 = sel(m1, c2, c1), t2 = sel(m1, c3, c2), t3 = sel(m1, c4, c3);
    const float t4 = sel(m1, c5, c4), t5 = sel(m1, c6, c5), t6 = sel(m1, c7, c6), t7 = sel(m1, c8, c7);
    const float t8 = sel(m1, c9, c8), t9 = sel(m1, c10, c9), t10 = sel(m1, c11, c10);
    const float f[9] = {sel(m2, t2, t0), sel(m2, t3, t1), sel(m2, t4, t2), sel(m2, t5, t3), sel(m2, t6, t4),
                        sel(m2, t7, t5), sel(m2, t8, t6), sel(m2, t9, t7), sel(m2, t10, t8)};
#pragma unroll
    for (int t = 0; t < D; ++t) {
      float v = f[t] * invC;
      v = (slope != 0.f) ? fmaxf(v, v * slope) : v;
      T tmp;
      Elem<T>::store(&tmp, v);
      patch[t * 64 + lane] = tmp.v;                                    // [t][row-in-pair][32 px]
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    // 72 chunks of 8 pixels (16 B): chunk L = (t, row-in-pair, 8-px segment)
    const int yb = y0 + 2 * u;
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
      const int L = lane + 64 * pass;
      if (L < D * 8) {
        const int t = L >> 3, rr = (L >> 2) & 1, seg = L & 3;
        const uint4 v = *reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(patch) + L * 16);
        const int y = yb + rr, x = x0 + 8 * seg;
        if constexpr (!(ABL & 4)) {
          if (y < H && x < W) *reinterpret_cast<uint4*>(obase + ((size_t)t * H + y) * W + x) = v;
        } else {
          asm volatile("" ::"v"(v.x), "v"(v.y), "v"(v.z), "v"(v.w));
        }
      }
    }
    __builtin_amdgcn_wave_barrier();
  };

  for (int c = 0; c < nchunks; ++c) {
    if (c > 0) __syncthreads();                                        // everyone is done reading the chunk
    // ---- stage chunk c: 4 channel rows x 4 pixels -> 4 pixel entries of 4 channels
    if constexpr (!(ABL & 1)) {
      const uint32_t coff = (uint32_t)c * (KQ * 4) * plane;
#pragma unroll
      for (int j0 = 0; j0 < TASKS_PER_THREAD; j0 += 2) {              // two tasks (8 loads) in flight per round
        u32x2 raw[2][4];
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) {
          const uint32_t off = task[j0 + jj].voff + coff;
#pragma unroll
          for (int k = 0; k < 4; ++k) raw[jj][k] = __builtin_amdgcn_raw_buffer_load_b64(rsrc[j0 + jj], off + k * plane, 0, 0);
        }
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) {
          if (task[j0 + jj].lds < 0) continue;
          const u32x2 a = raw[jj][0], b = raw[jj][1], cc = raw[jj][2], d = raw[jj][3];
          uint4 lo, hi;                                                // pixels 0,1 | pixels 2,3
          lo.x = __builtin_amdgcn_perm(b.x, a.x, 0x05040100u);  lo.y = __builtin_amdgcn_perm(d.x, cc.x, 0x05040100u);
          lo.z = __builtin_amdgcn_perm(b.x, a.x, 0x07060302u);  lo.w = __builtin_amdgcn_perm(d.x, cc.x, 0x07060302u);
          hi.x = __builtin_amdgcn_perm(b.y, a.y, 0x05040100u);  hi.y = __builtin_amdgcn_perm(d.y, cc.y, 0x05040100u);
          hi.z = __builtin_amdgcn_perm(b.y, a.y, 0x07060302u);  hi.w = __builtin_amdgcn_perm(d.y, cc.y, 0x07060302u);
          *reinterpret_cast<uint4*>(lds + task[j0 + jj].lds) = lo;
          *reinterpret_cast<uint4*>(lds + task[j0 + jj].lds + 2) = hi;
        }
      }
    }
    __syncthreads();
    // ---- matrix work: 4 units (row pairs) x KQ channel quads x 3 candidate quads
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      constexpr int Z = 0;
      const int au = SINGLE ? Z : u;
      if constexpr (SINGLE) acc[0][0] = acc[0][1] = acc[0][2] = f32x4{0.f, 0.f, 0.f, 0.f};
      if constexpr (!(ABL & 2)) {
        const uint2* pb = lds + (2 * u + rsel) * TW + pix;                                   // f1 (B operand)
        const uint2* pa = lds + KQ * F1_E + (2 * u + rsel + dyi) * F2W + pix;                // f2 (A operand), q=0
#pragma unroll 4
        for (int kq = 0; kq < KQ; ++kq) {
          const uint2 bv = pb[kq * F1_E];
          const uint2 a0 = pa[kq * F2_E], a1 = pa[kq * F2_E + 4], a2 = pa[kq * F2_E + 8];
          acc[au][0] = Mma<T>::mma(a0, bv, acc[au][0]);
          acc[au][1] = Mma<T>::mma(a1, bv, acc[au][1]);
          acc[au][2] = Mma<T>::mma(a2, bv, acc[au][2]);
        }
      }
      if constexpr (SINGLE) epilogue(u, acc[0][0], acc[0][1], acc[0][2]);   // finished: store while the next unit computes
    }
  }
  if constexpr (!SINGLE) {
#pragma unroll
    for (int u = 0; u < 4; ++u) epilogue(u, acc[u][0], acc[u][1], acc[u][2]);
  }
}

}  // namespace corrm
}  // namespace upf
