// 81-neighbour cost volume, forward, bf16 / fp16 — matrix-core kernel with the WHOLE channel depth resident in LDS.
//
// Same matrix mapping as corr81_mfma_kernel.hpp (v_mfma_f32_4x4x4_16b: block b multiplies 4 candidate pixels of f2 by
// 4 pixels of f1 over a channel quad; each lane ends up owning one pixel and 12 candidates; a two-stage barrel shift
// aligns them to the 9 dx), but the structure around it is different:
//
//   * NO channel loop.  A workgroup stages ALL ceil(C/4) channel quads of its f1 tile and f2 tile + halo at once: every
//     global load of the workgroup is in flight together, ONE barrier, then units (64 pixels x one dy per wave) are
//     computed and stored one after the other, stores of unit u overlapping the matrix work of unit u+1.  The chunked
//     kernel walked C/32 chunks serially (load latency + 2 barriers each): at the coarse pyramid levels (C = 196 / 128 /
//     96 on 6x20 ... 24x80 pixels, 8-16 workgroups on 256 CUs) that serial chain WAS the kernel time (33 / 18 / 15 us).
//   * Tile geometry is a template parameter so that the tile shrinks as C grows (LDS is 160 KB) and as the image
//     shrinks (more workgroups): unit = UR x UW pixels (2x32 or 4x16), tile = NU units stacked vertically.
//         <UW=32,NU=4>  8x32 tile   C <=  40   (1/4-resolution level)      <UW=32,NU=1>  2x32 tile   C <= 156
//         <UW=32,NU=2>  4x32 tile   C <= 120                               <UW=16,NU=1>  4x16 tile   C <= 208
//   * RAGGED: any W (rows not 8-byte aligned: 2-byte-aligned b64 buffer loads are legal on gfx950, the quad that
//     straddles the row end is masked; results leave as per-pixel 2-byte stores) — so every level of every
//     configuration (W = 20, 45, 90, 13, 26, 311 ...) takes the matrix-core path, none falls back to the VALU kernel.
//   * NORM: network_tools.normalize_features (model/upflow.py:94-137, inference flags) fused into the loader — the
//     statistics launch leaves the FINAL (mean, 1/std) pair of every (item, channel) row (round 4: rounds 2-3 left per-row
//     (count, mean, M2) partials that every workgroup merged itself: partial loads, two IEEE divisions + a square root + a
//     reciprocal per row, an LDS hand-off and a workgroup barrier in front of the loader), and every element is normalised
//     and re-rounded to 16 bits on its way into LDS: bit-identical to normalize_apply + corr81, without the write + read of
//     both normalised feature maps and without that launch.  With <= 4 staging tasks per thread (the 8x32 and 4x32 tiles) a
//     task's four (mean, 1/std) pairs travel with its feature loads into registers (two 16-byte loads): no LDS, no barrier.
#pragma once
#include "common.hpp"
#include "norm_merge.hpp"

#ifndef UPF_ALLC_STREG
#define UPF_ALLC_STREG 0   // 1: a task's (mean, 1/std) pairs travel with its feature loads into registers (measured SLOWER: see the header)
#endif
#ifndef UPF_ALLC_ABL
#define UPF_ALLC_ABL 0     // tools/corr_norm_ablate.hip: 1 no normalisation arithmetic, 4 no statistics loads, 8 no barrier behind the LDS copy of the statistics (8-task variants)
#endif

namespace upf {
namespace corrx {

constexpr int R = 4, D = 9;
constexpr int NWAVES = D, NTHREADS = NWAVES * 64;
constexpr int PATCH_BYTES = D * 64 * 2;      // per-wave output transposition patch (1152 B)

template <int UW, int NU> struct Geo {
  static constexpr int UR = 64 / UW, TH = NU * UR, TW = UW;
  static constexpr int F2W = UW + 2 * R, F2H = TH + 2 * R;
  static constexpr int F1_E = TH * TW, F2_E = F2H * F2W, E = F1_E + F2_E;     // 8-byte entries per channel quad
  static constexpr int Q1 = F1_E / 4, Q2 = F2_E / 4;                           // 4-pixel staging tasks per channel quad
};

template <int UW, int NU>
__host__ __device__ constexpr size_t lds_bytes(int KQ, bool ragged, bool norm) {
  return (size_t)KQ * Geo<UW, NU>::E * 8 + (ragged ? 0 : NWAVES * PATCH_BYTES) + (norm ? (size_t)2 * KQ * 4 * 8 : 0);   // (norm: the LDS copy of the statistics, used by the 8-task variants)
}
template <int UW, int NU>
__host__ __device__ constexpr int ntasks(int KQ) {       // f1 tasks padded to a whole number of waves, then the f2 tasks
  return ((KQ * Geo<UW, NU>::Q1 + 63) & ~63) + KQ * Geo<UW, NU>::Q2;
}

typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(4))) _Float16 f16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;

template <typename T> struct Mma;
template <> struct Mma<bf16_t> {
  static __device__ __forceinline__ f32x4 mma(uint2 a, uint2 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(__builtin_bit_cast(s16x4, a), __builtin_bit_cast(s16x4, b), c, 0, 0, 0);
  }
  static __device__ __forceinline__ float lo(uint32_t v) { return __uint_as_float(v << 16); }
  static __device__ __forceinline__ float hi(uint32_t v) { return __uint_as_float(v & 0xffff0000u); }
};
template <> struct Mma<f16_t> {
  static __device__ __forceinline__ f32x4 mma(uint2 a, uint2 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_4x4x4f16(__builtin_bit_cast(f16x4, a), __builtin_bit_cast(f16x4, b), c, 0, 0, 0);
  }
  static __device__ __forceinline__ float lo(uint32_t v) { return f16_bits_to_f32(v & 0xffffu); }
  static __device__ __forceinline__ float hi(uint32_t v) { return f16_bits_to_f32(v >> 16); }
};

struct Task {      // two registers per task: NT = 8 tasks + their 64 raw-data registers must stay under 96 VGPRs
  int meta;        // 8-byte LDS entry index of the first of 4 pixels (bits 0-15) | channel quad << 16 | pixels of the quad
                   // inside the row (0 = outside the image, 1..4) << 24;  < 0 = no task
  uint32_t voff;   // byte offset of (channel quad, row, column) inside the batch item; 0x80000000 = outside the image
  __device__ __forceinline__ int lds() const { return meta & 0xffff; }
  __device__ __forceinline__ int kq() const { return (meta >> 16) & 0xff; }
  __device__ __forceinline__ int nv() const { return (meta >> 24) & 0x7; }
};

// out: [B,81,H,W] (batch stride out_bs).  !RAGGED requires W % 8 == 0 and 16-byte aligned pointers / strides.
// ws1 / ws2 (NORM): the final (mean, 1/std) pairs of the rows of f1 / f2, float2 [B*C] each (misc::launch_stats2); nseg unused.
// TPW (tiles per workgroup; the product instantiates 1): with 2 the workgroup issues the global loads of a SECOND tile
// before the matrix work of the first and lands them in LDS afterwards.  Measured on MI355X and NOT used: 31.5 us instead
// of 23.8 us at [8,32,96,320] (57.6 vs 39.4 at [16,32,112,256]) — the two tiles of a workgroup serialise (load, compute,
// land, compute) while two rounds of independent workgroups, two resident per CU, overlap each other's phases for free.
// OC8 (!RAGGED): the 81 channels leave as 11 channel octets of a C8 buffer [n][octet][H][W][8] (conv_c8.hip), out_bs = its batch
// stride in elements, `out` = the first of the 11 octets.  Octet j < 9 holds displacements (dy = j - 4, dx = -4 .. +3), octet 9
// position p holds (dy = p - 4, dx = +4), octet 10 position 0 holds (+4, +4) and zeros: a wave (= one dy) stores ONE whole
// 16-byte entry per pixel straight from its registers (no transposition patch) + one 2-byte element.  The convolution that
// reads the buffer gets this order through its k-map (upf_conv_pack_weights_kmap; ops.corr81_c8_channel_map).
// fpitch (round 5): elements between consecutive rows of f1 / f2 (>= W; plane stride H * fpitch).  PADW (!RAGGED, OC8): the LOGICAL
// width W is ragged but the rows are pitched to whole 16-byte groups — aligned quad loads like the W % 8 == 0 form, the quad that
// straddles W has its trailing pixels (pitch padding, whatever it holds) replaced by the zero padding, and the octet output, one
// 16-byte entry per pixel, is aligned for every W: KITTI's native frames (W = 311, 156 at the two fine levels) stay on this path.
// TO (round 5): storage type of `out` when it differs from the features' (`pyramid_dtype`: fp16 features, bf16 estimator buffers).
template <typename T, int UW, int NU, int NT, bool RAGGED, bool NORM, int TPW = 1, bool OC8 = false, bool PADW = false, typename TO = T>
__global__ __launch_bounds__(NTHREADS, 5)       // <= 102 VGPRs: two 9-wave workgroups per CU
void corr81_allc_kernel(const T* __restrict__ f1, const T* __restrict__ f2, TO* __restrict__ out,
                        int C, int H, int W, int tiles_x, int tiles_y, long long out_bs, float slope,
                        const float* __restrict__ ws1, const float* __restrict__ ws2, int nseg, int total_tiles, int fpitch) {
  static_assert(!PADW || (!RAGGED && OC8), "PADW: aligned pitched rows, octet output");
  using G = Geo<UW, NU>;
  extern __shared__ __attribute__((aligned(16))) uint2 lds[];
  const int ntiles = tiles_x * tiles_y;                                 // per batch item
  const int tid = threadIdx.x, lane = tid & 63;
  const int dyi = __builtin_amdgcn_readfirstlane(tid >> 6);            // 0..8 <-> dy = dyi-4
  const int KQ = (C + 3) >> 2;
  uint2* const lds_f1 = lds;
  uint2* const lds_f2 = lds + KQ * G::F1_E;
  const uint32_t plane = (uint32_t)H * (uint32_t)fpitch * 2u;          // bytes per channel plane
  const uint32_t item_bytes = (uint32_t)C * plane;
  const int n1 = KQ * G::Q1, N1 = (n1 + 63) & ~63, n2 = KQ * G::Q2;
  float2* const st = reinterpret_cast<float2*>(lds + KQ * G::E + (RAGGED ? 0 : NWAVES * PATCH_BYTES / 8));

  Task task[NT];
  u32x2 raw[NT][4];
  constexpr bool STREG = NORM && NT <= 4 && (UPF_ALLC_STREG != 0);   // statistics in registers, per task (else: staged through LDS once)
  f32x4 sreg[STREG ? NT : 1][2];                        // (mean, 1/std) of the task's 4 channels
  // ---- staging: every load of a tile is issued before anything is consumed
  auto issue = [&](int tile, bool live) {
    const int tx = tile % tiles_x, ty = (tile / tiles_x) % tiles_y, n = tile / ntiles;
    const int x0 = tx * G::TW, y0 = ty * G::TH;
    const size_t item = (size_t)n * C * H * fpitch;
    const __amdgpu_buffer_rsrc_t r1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(f1 + item), 0, live ? item_bytes : 0u, 0x00020000);
    const __amdgpu_buffer_rsrc_t r2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(f2 + item), 0, live ? item_bytes : 0u, 0x00020000);
    // (NORM) the item's (mean, 1/std) pairs: channels >= C fall off the descriptor -> (0, 0): (x - 0) * 0 keeps the zero padding
    const __amdgpu_buffer_rsrc_t q1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(ws1 + (NORM ? (size_t)n * C * 2 : 0)), 0, (live && NORM) ? (uint32_t)C * 8u : 0u, 0x00020000);
    const __amdgpu_buffer_rsrc_t q2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(ws2 + (NORM ? (size_t)n * C * 2 : 0)), 0, (live && NORM) ? (uint32_t)C * 8u : 0u, 0x00020000);
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      const int t = tid + j * NTHREADS;
      const bool from_f2 = __builtin_amdgcn_readfirstlane((tid & ~63) + j * NTHREADS) >= N1;     // wave-uniform
      Task s;
      s.meta = -1; s.voff = 0x80000000u;
      int kq = 0, gy = -1, gx = -1, lds_at = -1;
      if (!from_f2) {
        if (t < n1) {
          kq = t / G::Q1;
          const int rem = t - kq * G::Q1, r = rem / (G::TW / 4), g = rem - r * (G::TW / 4);
          gy = y0 + r; gx = x0 + 4 * g;
          lds_at = kq * G::F1_E + r * G::TW + 4 * g;
        }
      } else {
        const int u = t - N1;
        if (u < n2) {
          kq = u / G::Q2;
          const int rem = u - kq * G::Q2, r = rem / (G::F2W / 4), g = rem - r * (G::F2W / 4);
          gy = y0 - R + r; gx = x0 - R + 4 * g;
          lds_at = KQ * G::F1_E + kq * G::F2_E + r * G::F2W + 4 * g;
        }
      }
      if (lds_at >= 0) s.meta = lds_at | (kq << 16);
      if (lds_at >= 0 && gy >= 0 && gy < H && gx >= 0 && gx < W) {
        const int nv = min(W - gx, 4);                   // pixels of the quad inside the row
        // RAGGED (W >= 4): the quad that straddles the row end is loaded shifted left so that it ENDS at the row end
        // (every byte of every load then lies inside the row: nothing is read past the tensor, and the bounds check of
        // the descriptor, which works in whole dwords, never cuts off an odd-sized tensor's last element); the shift
        // is undone in registers when the task lands
        s.voff = (uint32_t)((kq * 4 * H + gy) * fpitch + gx - (RAGGED ? 4 - nv : 0)) * 2u;
        s.meta |= nv << 24;
      }
      task[j] = s;
      const __amdgpu_buffer_rsrc_t rs = from_f2 ? r2 : r1;
#pragma unroll
      for (int k = 0; k < 4; ++k) raw[j][k] = __builtin_amdgcn_raw_buffer_load_b64(rs, s.voff + k * plane, 0, 0);
      if constexpr (STREG) {
        const uint32_t so = (lds_at >= 0 && !(UPF_ALLC_ABL & 4)) ? (uint32_t)kq * 32u : 0x80000000u;
        const __amdgpu_buffer_rsrc_t qs = from_f2 ? q2 : q1;
        sreg[j][0] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(qs, so, 0, 0));
        sreg[j][1] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(qs, so + 16u, 0, 0));
      }
    }
  };
  // ---- NORM, 8-task variants: the 2 x C final (mean, 1/std) pairs of item n -> LDS (one 8-byte load per thread, issued BEFORE the
  // feature loads: vmcnt retires in order, so the wait for it does not wait for the features), one barrier
  float2 pst = make_float2(0.f, 0.f);
  auto load_stats = [&](int n) {
    if constexpr (NORM && !STREG) {
      const int c4 = KQ * 4;
      pst = make_float2(0.f, 0.f);                   // channels >= C: (x - 0) * 0 keeps the zero padding of the quad
      if (tid < 2 * c4) {
        const int sel = tid >= c4, c = tid - sel * c4;
        if (c < C && !(UPF_ALLC_ABL & 4)) pst = *reinterpret_cast<const float2*>((sel ? ws2 : ws1) + ((size_t)n * C + c) * 2);
      }
    }
  };
  auto merge_stats = [&]() {
    if constexpr (NORM && !STREG) {
      if (tid < 2 * KQ * 4) st[tid] = pst;
      if (!(UPF_ALLC_ABL & 8)) __syncthreads();
    }
  };
  // ---- 4 channel rows x 4 pixels -> 4 pixel entries of 4 channels (v_perm), 2 ds_write_b128 per task
  auto land = [&]() {
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      if (task[j].meta < 0) continue;
      if constexpr (RAGGED) {                             // the quad that straddles the row end
        const int nv = task[j].nv();                      // 0 (outside: loads returned zeros) or 1..4
        if (nv > 0 && nv < 4) {                           // undo the left shift of the load: pixel i = loaded pixel i + (4 - nv)
          const int sh = 16 * (4 - nv);
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            const unsigned long long q = (((unsigned long long)raw[j][k].y << 32) | raw[j][k].x) >> sh;
            raw[j][k].x = (uint32_t)q; raw[j][k].y = (uint32_t)(q >> 32);
          }
        }
      }
      u32x2 v[4] = {raw[j][0], raw[j][1], raw[j][2], raw[j][3]};
      if constexpr (NORM && !(UPF_ALLC_ABL & 1)) {
        if (task[j].nv() != 0) {
          const bool from_f2 = __builtin_amdgcn_readfirstlane((tid & ~63) + j * NTHREADS) >= N1;
          const float2* sp = st + (from_f2 ? KQ * 4 : 0) + task[j].kq() * 4;
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            float2 ms;
            if constexpr (STREG) ms = make_float2(sreg[j][k >> 1][2 * (k & 1)], sreg[j][k >> 1][2 * (k & 1) + 1]);
            else ms = sp[k];
            // (x - mean) * rstd in separately rounded fp32 steps, then ONE rounding to the storage type (pack2), exactly
            // like normalize_apply_kernel
            // NOTE (round 4): hipcc's SLP vectoriser used to turn these four sub / mul pairs into v_pk_add_f32 / v_pk_mul_f32 (bf16
            // only: the fp16 conversions keep the lanes apart).  On MI355X those packed-fp32 VALU instructions return WRONG results
            // now and then while a wave of a v_mfma_f32_16x16x32 kernel (the narrow convolution) shares the SIMD: alone on the GPU
            // this kernel is bit-reproducible, beside the narrow convolution on a second stream 29 of 30 launches differed (whole
            // tiles, |diff| up to 0.3), and with the same arithmetic in scalar fp32 instructions 0 of 30 (tools/corr_race3.py).
            // The whole library is therefore built without the packed-fp32 target feature (upflow_pytorch_amd/_build.py).
            v[k].x = pack2<T>(__fmul_rn(__fsub_rn(Mma<T>::lo(v[k].x), ms.x), ms.y), __fmul_rn(__fsub_rn(Mma<T>::hi(v[k].x), ms.x), ms.y));
            v[k].y = pack2<T>(__fmul_rn(__fsub_rn(Mma<T>::lo(v[k].y), ms.x), ms.y), __fmul_rn(__fsub_rn(Mma<T>::hi(v[k].y), ms.x), ms.y));
          }
        }
      }
      if constexpr ((RAGGED && NORM) || PADW) {           // pixels >= W were normalised zeros (PADW: or pitch padding): drop them
        const int nv = task[j].nv();
        const uint32_t mx = nv >= 2 ? 0xffffffffu : (nv == 1 ? 0x0000ffffu : 0u);
        const uint32_t my = nv >= 4 ? 0xffffffffu : (nv == 3 ? 0x0000ffffu : 0u);
#pragma unroll
        for (int k = 0; k < 4; ++k) { v[k].x &= mx; v[k].y &= my; }
      }
      uint4 lo, hi;                                       // pixels 0,1 | pixels 2,3
      lo.x = __builtin_amdgcn_perm(v[1].x, v[0].x, 0x05040100u);  lo.y = __builtin_amdgcn_perm(v[3].x, v[2].x, 0x05040100u);
      lo.z = __builtin_amdgcn_perm(v[1].x, v[0].x, 0x07060302u);  lo.w = __builtin_amdgcn_perm(v[3].x, v[2].x, 0x07060302u);
      hi.x = __builtin_amdgcn_perm(v[1].y, v[0].y, 0x05040100u);  hi.y = __builtin_amdgcn_perm(v[3].y, v[2].y, 0x05040100u);
      hi.z = __builtin_amdgcn_perm(v[1].y, v[0].y, 0x07060302u);  hi.w = __builtin_amdgcn_perm(v[3].y, v[2].y, 0x07060302u);
      *reinterpret_cast<uint4*>(lds + task[j].lds()) = lo;
      *reinterpret_cast<uint4*>(lds + task[j].lds() + 2) = hi;
    }
  };

  // ---- matrix work on the tile in LDS: NU units x KQ channel quads x 3 candidate quads; a finished unit is stored while
  // the next computes
  const int rsel = lane / UW, pix = lane % UW;                          // pix = 4*quad + j
  const int p = lane & 3;
  const uint32_t m1 = (p & 1) ? 0xffffffffu : 0u, m2 = (p & 2) ? 0xffffffffu : 0u;   // per-lane select masks
  const float invC = 1.0f / (float)C;
  using st16 = uint16_t;
  uint16_t* patch = reinterpret_cast<uint16_t*>(lds + KQ * G::E) + (tid >> 6) * (PATCH_BYTES / 2);
  auto sel = [](uint32_t mask, float a, float b) {
    return __uint_as_float((__float_as_uint(a) & mask) | (__float_as_uint(b) & ~mask));
  };
  auto compute = [&](int tile) {
    const int tx = tile % tiles_x, ty = (tile / tiles_x) % tiles_y, n = tile / ntiles;
    const int x0 = tx * G::TW, y0 = ty * G::TH;
    st16* obase = reinterpret_cast<st16*>(out) + (size_t)n * out_bs + (OC8 ? (size_t)0 : (size_t)(dyi * D) * H * W);
#pragma unroll
    for (int u = 0; u < NU; ++u) {
      f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = a0, a2 = a0;
      const uint2* pb = lds_f1 + (u * G::UR + rsel) * G::TW + pix;                           // f1 (B operand)
      const uint2* pa = lds_f2 + (u * G::UR + rsel + dyi) * G::F2W + pix;                    // f2 (A operand), q = 0
      // channel quads four at a time (16 ds_read_b64 in flight for 12 MFMAs), written out because hipcc does not
      // partially unroll the run-time-bounded loop around the MFMA builtins.  (A hand-pipelined version that loads group
      // g+1 during the MFMAs of group g was measured SLOWER at every level — 13.9 -> 17.6 us at the 1/4-resolution level:
      // twice the operand registers and a block of moves per group; the 2-3 waves per SIMD already hide the LDS latency.)
      int kq = 0;
      for (; kq + 4 <= KQ; kq += 4) {
        uint2 bv[4], c0[4], c1[4], c2[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          bv[i] = pb[(kq + i) * G::F1_E];
          c0[i] = pa[(kq + i) * G::F2_E]; c1[i] = pa[(kq + i) * G::F2_E + 4]; c2[i] = pa[(kq + i) * G::F2_E + 8];
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          a0 = Mma<T>::mma(c0[i], bv[i], a0);
          a1 = Mma<T>::mma(c1[i], bv[i], a1);
          a2 = Mma<T>::mma(c2[i], bv[i], a2);
        }
      }
      for (; kq < KQ; ++kq) {
        const uint2 bv = pb[kq * G::F1_E];
        const uint2 c0 = pa[kq * G::F2_E], c1 = pa[kq * G::F2_E + 4], c2 = pa[kq * G::F2_E + 8];
        a0 = Mma<T>::mma(c0, bv, a0);
        a1 = Mma<T>::mma(c1, bv, a1);
        a2 = Mma<T>::mma(c2, bv, a2);
      }
      // candidates 0..11 of this lane's pixel; displacement t = dx+4 is candidate t + p: two-stage barrel shift by the
      // lane's 2-bit position, as bit-selects on scalars (arrays indexed by a lane-dependent value go to scratch)
      const float t0 = sel(m1, a0[1], a0[0]), t1 = sel(m1, a0[2], a0[1]), t2 = sel(m1, a0[3], a0[2]), t3 = sel(m1, a1[0], a0[3]);
      const float t4 = sel(m1, a1[1], a1[0]), t5 = sel(m1, a1[2], a1[1]), t6 = sel(m1, a1[3], a1[2]), t7 = sel(m1, a2[0], a1[3]);
      const float t8 = sel(m1, a2[1], a2[0]), t9 = sel(m1, a2[2], a2[1]), t10 = sel(m1, a2[3], a2[2]);
      const float f[9] = {sel(m2, t2, t0), sel(m2, t3, t1), sel(m2, t4, t2), sel(m2, t5, t3), sel(m2, t6, t4),
                          sel(m2, t7, t5), sel(m2, t8, t6), sel(m2, t9, t7), sel(m2, t10, t8)};
      if constexpr (OC8) {
        static_assert(!OC8 || !RAGGED, "octet output needs W % 8 == 0");
        const int y = y0 + u * G::UR + rsel, x = x0 + pix;
        if (y < H && x < W) {
          float v[D];
#pragma unroll
          for (int t = 0; t < D; ++t) { v[t] = f[t] * invC; v[t] = (slope != 0.f) ? fmaxf(v[t], v[t] * slope) : v[t]; }
          const size_t px = (size_t)y * W + x, oct = (size_t)H * W * 8;
          *reinterpret_cast<uint4*>(obase + dyi * oct + px * 8) = make_uint4(pack2<TO>(v[0], v[1]), pack2<TO>(v[2], v[3]), pack2<TO>(v[4], v[5]), pack2<TO>(v[6], v[7]));
          const uint32_t last = pack2<TO>(v[8], 0.f);
          if (dyi < 8) obase[9 * oct + px * 8 + dyi] = (st16)(last & 0xffffu);
          else *reinterpret_cast<uint4*>(obase + 10 * oct + px * 8) = make_uint4(last & 0xffffu, 0u, 0u, 0u);
        }
      } else if constexpr (RAGGED) {
        const int y = y0 + u * G::UR + rsel, x = x0 + pix;
        if (y < H && x < W) {
          st16* o = obase + (size_t)y * W + x;
#pragma unroll
          for (int t = 0; t < D; ++t) {
            float v = f[t] * invC;
            v = (slope != 0.f) ? fmaxf(v, v * slope) : v;
            TO tmp;
            Elem<TO>::store(&tmp, v);
            o[(size_t)t * H * W] = tmp.v;
          }
        }
      } else {
#pragma unroll
        for (int t = 0; t < D; ++t) {
          float v = f[t] * invC;
          v = (slope != 0.f) ? fmaxf(v, v * slope) : v;
          TO tmp;
          Elem<TO>::store(&tmp, v);
          patch[t * 64 + lane] = tmp.v;                                  // [t][row-in-unit][UW px]
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        // 72 chunks of 8 pixels (16 B): chunk L = (t, row-in-unit, 8-px segment)
        const int yb = y0 + u * G::UR;
#pragma unroll
        for (int pass = 0; pass < 2; ++pass) {
          const int L = lane + 64 * pass;
          if (L < D * 8) {
            constexpr int SPR = UW / 8;
            const int t = L >> 3, rr = (L / SPR) % G::UR, seg = L % SPR;
            const uint4 v = *reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(patch) + L * 16);
            const int y = yb + rr, x = x0 + 8 * seg;
            if (y < H && x < W) *reinterpret_cast<uint4*>(obase + ((size_t)t * H + y) * W + x) = v;
          }
        }
        __builtin_amdgcn_wave_barrier();
      }
    }
  };

  // tile order: consecutive workgroups (after the XCD remap) own consecutive tiles; with TPW = 2 a workgroup's two tiles
  // are gridDim.x apart, so that both halves of the grid sweep the images in the same order
  const int bid = xcd_remap(blockIdx.x, gridDim.x);
  load_stats(bid / ntiles);
  issue(bid, true);
  merge_stats();
  land();
  __syncthreads();
  if constexpr (TPW == 2) {
    const int tile2 = bid + (int)gridDim.x;
    const bool live2 = tile2 < total_tiles;                             // (workgroup-uniform)
    issue(live2 ? tile2 : bid, live2);                                  // in flight during the matrix work of the first tile
    compute(bid);
    __syncthreads();                                                    // every wave is done reading the first tile
    if (live2) {
      load_stats(tile2 / ntiles);
      merge_stats();
      land();
      __syncthreads();
      compute(tile2);
    }
  } else {
    compute(bid);
  }
}

}  // namespace corrx
}  // namespace upf
