// Shared device code of the matrix-core convolution (csrc/conv3x3.hip: NCHW operands; csrc/conv_c8.hip: operands in the
// channel-octet layout): MFMA wrappers, staging helpers, epilogues, conv_kernel, launch_one and the launch heuristics' state.
#pragma once
#include "common.hpp"
#include <cstring>
#include <cstdint>

namespace upf {
namespace conv {

constexpr int TW = 32, NTHREADS = 256;
constexpr int MAXD = 16;                  // dilation limit (the context network's largest)
// D template values: 0 = 1x1; 1,2,4,8,16 = 3x3 with that dilation; -1 = 3x3, run-time dilation (1..16)
__host__ __device__ constexpr int margin_of(int D) { return (D == 16 || D < 0) ? 16 : 8; }
// staged columns [S*x0 - marg, S*x0 + S*32 + marg): whole 8-pixel groups -> 16-byte global loads
__host__ __device__ constexpr int xw(int S, int marg) { return S * TW + 2 * marg; }
// Round 5: a PURE octet input (XL == 1) is staged by LDS-DMA entry by entry — no 8-pixel groups to keep aligned — so its LDS rows hold
// only the columns the taps read, [x0 - D, x0 + 32 + D): 34 entries (+1: odd pitch) instead of 51 for dilation 1.  The narrow layers
// are LDS-capacity limited (two 32-KB buffers per workgroup = two workgroups per CU): at 22 KB per buffer a third workgroup fits.
#ifndef UPF_C8_TIGHT
#define UPF_C8_TIGHT 1
#endif
__host__ __device__ constexpr bool tight_rows(int XL) { return XL == 1 && UPF_C8_TIGHT != 0; }
__host__ __device__ constexpr int marg_eff(int XL, int D) { return tight_rows(XL) ? (D > 0 ? D : 0) : margin_of(D); }
__host__ __device__ constexpr int xw_eff(int XL, int S, int D) { return xw(S, marg_eff(XL, D)); }
__host__ __device__ constexpr int xwp_eff(int XL, int S, int D) { return tight_rows(XL) ? (xw_eff(XL, S, D) | 1) : xw_eff(XL, S, D) + xw_eff(XL, S, D) / 16; }

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;

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

// 16 output channels x 16 pixels x 32 input channels per instruction (the narrow layers, N16): A = weights [16 co][32 k]
// (lane = co + 16 * k-octet), B = x [32 k][16 px] (lane = px + 16 * k-octet: one LDS entry of 8 channels), D: lane holds
// channels 4 * (lane / 16) + i, i = 0..3, of pixel lane % 16
template <typename T> struct Mma16;
template <> struct Mma16<bf16_t> {
  static __device__ __forceinline__ f32x4 mma(uint4 a, uint4 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
  }
};
template <> struct Mma16<f16_t> {
  static __device__ __forceinline__ f32x4 mma(uint4 a, uint4 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
  }
};

__host__ __device__ constexpr int pad32(int v) { return (v + 31) / 32 * 32; }

// LDS x tile: row pitch XW + XW/16 entries, and inside every 8-pixel group the pixel slots are ROTATED by group/2:
// a staging write phase (16 lanes = consecutive 8-pixel groups of a few rows, each lane writing pixel q of its group)
// then hits 16 different bank quads (tools: brute-force search over rotations and pitches) instead of piling onto two
// of them, while the global loads stay coalesced (lanes adjacent along the row).  A 16-column read window that starts
// inside a group sees one 2-way conflict at most.  swz(col) = position of staged column `col` inside its row.
__device__ __forceinline__ int swz(int col) { return (col & ~7) | ((col + (col >> 4)) & 7); }

// ---- x staging helpers shared by both kernels -------------------------------------------------------------------------
// 8 channel rows x 8 pixels (eight 16-byte loads of one 8-pixel group) -> 8 LDS entries of 8 channels x 1 pixel.
// `enc` = (entry index of the group's first pixel) * 8 + slot rotation (see swz); GEN: `sh` = pixels the load window was
// shifted left so that it ends at the row end — pixel q sits at column q - sh, columns >= 8 - sh are zero padding.
// !GEN (rows with a 16-byte aligned PITCH; the logical width W may be ragged, round 5): `sh` = trailing pixels of the group
// that lie at or beyond W — what the load fetched there (pitch padding, whatever it holds) is replaced by the zero padding.
template <bool GEN>
__device__ __forceinline__ void stage_store(uint4* tile, int enc, int sh, const u32x4 (&ch_in)[8]) {
  uint4* dst = tile + (enc >> 3);
  const int rot = enc & 7;
  u32x4 ch[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) ch[k] = ch_in[k];
  if constexpr (!GEN) {
    if (sh) {                                        // (rare: the last group of a ragged row, right-edge tiles only)
      const int nv = 8 - sh;                         // valid pixels
#pragma unroll
      for (int pp = 0; pp < 4; ++pp) {
        const uint32_t m = (2 * pp + 1 < nv) ? 0xffffffffu : ((2 * pp < nv) ? 0x0000ffffu : 0u);
#pragma unroll
        for (int k = 0; k < 8; ++k) ch[k][pp] &= m;
      }
    }
  }
#pragma unroll
  for (int pp = 0; pp < 4; ++pp) {               // dword pp of every channel row = pixels 2pp, 2pp+1
    uint4 e0, e1;
    e0.x = __builtin_amdgcn_perm(ch[1][pp], ch[0][pp], 0x05040100u); e1.x = __builtin_amdgcn_perm(ch[1][pp], ch[0][pp], 0x07060302u);
    e0.y = __builtin_amdgcn_perm(ch[3][pp], ch[2][pp], 0x05040100u); e1.y = __builtin_amdgcn_perm(ch[3][pp], ch[2][pp], 0x07060302u);
    e0.z = __builtin_amdgcn_perm(ch[5][pp], ch[4][pp], 0x05040100u); e1.z = __builtin_amdgcn_perm(ch[5][pp], ch[4][pp], 0x07060302u);
    e0.w = __builtin_amdgcn_perm(ch[7][pp], ch[6][pp], 0x05040100u); e1.w = __builtin_amdgcn_perm(ch[7][pp], ch[6][pp], 0x07060302u);
    if constexpr (GEN) {
      const uint32_t m0 = (2 * pp >= sh) ? 0xffffffffu : 0u, m1 = (2 * pp + 1 >= sh) ? 0xffffffffu : 0u;
      e0.x &= m0; e0.y &= m0; e0.z &= m0; e0.w &= m0;
      e1.x &= m1; e1.y &= m1; e1.z &= m1; e1.w &= m1;
      dst[(2 * pp - sh + rot) & 7] = e0;
      dst[(2 * pp + 1 - sh + rot) & 7] = e1;
    } else {
      dst[(2 * pp + rot) & 7] = e0;
      dst[(2 * pp + 1 + rot) & 7] = e1;
    }
  }
}

// ---- epilogue: accumulators -> LeakyReLU -> 16-bit -> y.  D[co][pixel]: lane -> pixel column px, register e ->
// output channel (e&3) + 8*(e>>2) + 4*kg of the 32-channel block.  Lane pairs (px, px^1) swap halves (one DPP move +
// one v_perm): the even lane stores pixels (px, px+1) of the even registers' channels, the odd lane pixels (px-1, px) of
// the odd registers' — 4-byte stores, half as many.  Stores go through a buffer descriptor over the image's Cout output
// planes, so channels >= Cout and columns >= Wo are dropped by the bounds check instead of by branches.
// ---- gated epilogue (round 5; training only).  The data-gradient convolutions of a dense stack were each followed by one pass
// (upf_act_grad) over their output: + the gradient arriving from outside the stack, x the LeakyReLU mask of the forward
// activation.  Removing those passes from the captured config-3 step took 1.13 ms off 9.51 (tools/act_grad_cost_probe.py), so the
// two operands are applied HERE, to the 16-bit values the epilogue is about to store — the same arithmetic on the same rounded
// values, in the same order, as the pass it replaces (conv_wgrad.hip act_grad_kernel): bit-identical outputs.
struct NoGate { static constexpr bool on = false, split = false, init = false, dual = false; };
// ---- merged narrow tail of a dense stack (round 6; inference, octet operands).  The layers of a dense stack read NESTED channel
// suffixes of one buffer (pwc_modules.py:279-286, model/upflow.py:53-60): conv_last reads [conv5 | what conv5 read], so
//   conv_last = W_last[:, conv5 part] * conv5_out + W_last[:, rest] * rest
// and the second term shares its input with conv5 exactly.  A launch with 2 ... 32 output channels costs what its staging costs
// (the 563 -> 2 head of the flow estimator took as long as the 531 -> 32 layer before it), so ONE pass over the shared input
// computes the first layer of the tail completely and, in the unused output channels of the same 32-channel block(s), the
// shared-input part of every later layer:
//   SplitOut: output channels [0, cmain) are a layer's own — bias, LeakyReLU, 16-bit octets, as always; channels [cmain, Cout)
//             are fp32 PARTIAL pre-activations of later layers (their bias included, no activation), stored as planes of channel
//             QUADS [n][ppitch / 4][y][x][4] — the 16 bytes of a lane's register quad next to its neighbour pixels' (a half-wave
//             writes, and the finishing launch's 16 lanes read, whole contiguous segments);
//   AccInit:  the finishing launch of a later layer (16-channel matrix instruction, its input = the few channels the tail's earlier
//             layers produced) starts its accumulators from that partial instead of from the bias.
// The sum is the same fp32 sum in another order (the layer's own K order put the tail's channels first, here they come last);
// no 16-bit rounding happens in between.
struct SplitOut {
  static constexpr bool on = false, split = true, init = false, dual = false;
  float* part; long long pbs; int ppitch; int cmain;      // cmain % 8 == 0
};
struct AccInit {
  static constexpr bool on = false, split = false, init = true, dual = false;
  const float* part; long long pbs; int ppitch; int coff;  // coff % 4 == 0: first float of this layer's partial within a pixel
};
// DualOut (round 6): the octets are stored TWICE — the 1x1 projection of a level's features is the input of two dense stacks (the flow
// estimator's buffer and the SGU estimator's, model/upflow.py:546-553 + :71-75) and used to be computed by two launches.
template <typename TO> struct DualOut {
  static constexpr bool on = false, split = false, init = false, dual = true;
  TO* y2; long long y2bs;
};
template <typename T> struct ActGate {
  static constexpr bool on = true, split = false, init = false, dual = false;
  const T* add; long long abs_;        // optional [B,Cout,Ho,Wo] channel slice added first (16-bit sum, rounded)
  const T* y; long long ybs;           // optional forward activation output: elements with !(y > 0) are scaled by slope
  float slope;
};
struct GateRsrc { __amdgpu_buffer_rsrc_t ar, mr; float slope; };
template <typename T, typename G>
__device__ __forceinline__ GateRsrc gate_init(const G& g, int n, uint32_t bytes) {
  GateRsrc r;
  // (an absent operand: a descriptor of zero bytes — every load returns 0; without y the host passes slope 1)
  r.ar = __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(g.add ? g.add + (size_t)n * g.abs_ : g.y), 0, g.add ? bytes : 0u, 0x00020000);
  r.mr = __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(g.y ? g.y + (size_t)n * g.ybs : g.add), 0, g.y ? bytes : 0u, 0x00020000);
  r.slope = g.slope;
  return r;
}
template <typename T> __device__ __forceinline__ float gate_f(uint32_t bits16) { T t; t.v = (unsigned short)bits16; return Elem<T>::load(&t); }
// v, a, q: two 16-bit elements each (value, addend, forward activation)
template <typename T>
__device__ __forceinline__ uint32_t gate2(uint32_t v, uint32_t a, uint32_t q, float slope) {
  const uint32_t s = pack2<T>(gate_f<T>(v & 0xffffu) + gate_f<T>(a & 0xffffu), gate_f<T>(v >> 16) + gate_f<T>(a >> 16));
  float f0 = gate_f<T>(s & 0xffffu), f1 = gate_f<T>(s >> 16);
  if (!(gate_f<T>(q & 0xffffu) > 0.f)) f0 *= slope;
  if (!(gate_f<T>(q >> 16) > 0.f)) f1 *= slope;
  return pack2<T>(f0, f1);
}

struct Epilogue {
  __amdgpu_buffer_rsrc_t yr;
  uint32_t off32, off16;   // this lane's byte offset at (first channel of its pair set, row 0, its pixel pair) or 0x80000000
  uint32_t sel;            // v_perm selector merging own and partner halves
  uint32_t plane2;         // bytes per output plane
};
template <typename T, bool GEN>
__device__ __forceinline__ void epilogue_init(Epilogue& ep, T* y_img, int Cout, int Ho, int Wo, int slab, int lane, int x0, int ypitch) {
  const int px = lane & 31, kg = lane >> 5;
  const bool odd = px & 1;
  const int gx = x0 + (px & ~1);
  ep.plane2 = (uint32_t)(Ho * ypitch) * 2u;
  ep.yr = __builtin_amdgcn_make_buffer_rsrc(y_img, 0, (uint32_t)Cout * ep.plane2, 0x00020000);
  const uint32_t off = (uint32_t)(slab * 32 + 4 * kg + (odd ? 1 : 0)) * ep.plane2 + (uint32_t)gx * 2u;
  ep.off32 = (gx + 1 < Wo) ? off : 0x80000000u;
  ep.off16 = (GEN && gx + 1 == Wo) ? off : 0x80000000u;     // odd Wo: the last column is a 2-byte store
  ep.sel = odd ? 0x03020706u : 0x05040100u;
}
// v0, v1: channels c and c+1 of this lane's pixel (registers e = 2j, 2j+1); soff: uniform byte offset of
// (channel offset of register 2j, output row) = ({0,2,8,10,16,18,24,26}[j] * Ho*Wo + gy*Wo) * 2
template <typename T, bool GEN, bool GATED = false>
__device__ __forceinline__ void epilogue_store(const Epilogue& ep, float v0, float v1, uint32_t soff, float slope, const GateRsrc* g = nullptr) {
  v0 = fmaxf(v0, v0 * slope); v1 = fmaxf(v1, v1 * slope);                  // slope = 1 -> identity
  const uint32_t p = pack2<T>(v0, v1);                                     // lo = channel c, hi = channel c+1 of pixel px
  const uint32_t recv = (uint32_t)__builtin_amdgcn_mov_dpp((int)p, 0xB1, 0xF, 0xF, true);   // quad_perm [1,0,3,2]
  uint32_t out = __builtin_amdgcn_perm(recv, p, ep.sel);
  if constexpr (GATED) {
    // (one of the two offsets is the out-of-range marker and reads 0)
    uint32_t a = __builtin_amdgcn_raw_buffer_load_b32(g->ar, ep.off32 + soff, 0, 0), q = __builtin_amdgcn_raw_buffer_load_b32(g->mr, ep.off32 + soff, 0, 0);
    if constexpr (GEN) {
      a |= (uint32_t)__builtin_amdgcn_raw_buffer_load_b16(g->ar, ep.off16 + soff, 0, 0);
      q |= (uint32_t)__builtin_amdgcn_raw_buffer_load_b16(g->mr, ep.off16 + soff, 0, 0);
    }
    out = gate2<T>(out, a, q, g->slope);
  }
  __builtin_amdgcn_raw_buffer_store_b32(out, ep.yr, ep.off32 + soff, 0, 0);
  if constexpr (GEN) __builtin_amdgcn_raw_buffer_store_b16((unsigned short)out, ep.yr, ep.off16 + soff, 0, 0);
}
__device__ __forceinline__ uint32_t epilogue_choff(int j) { return (uint32_t)((2 * j & 3) + 8 * (2 * j >> 2)); }   // channel offset of register 2j

// Wide epilogue (Wo % 8 == 0, 16-byte aligned rows): 4-byte stores issue at ~4 B/clk/CU, which bounds every layer with a
// large output (the 3->16 full-resolution layer wrote 126 MB in 94 us).  So each wave transposes its tile through a
// private LDS patch, two tile rows at a time — [row][channel][32 px] with an 80-byte channel pitch — and writes 16 bytes
// (8 pixels) per lane: 4x fewer store instructions, each covering whole 64-byte row segments.
constexpr int EPI_PITCH = 80;                                  // bytes per (row, channel) in the patch
constexpr int EPI_WAVE_BYTES = 2 * 32 * EPI_PITCH;             // two tile rows of one wave
template <typename T, int RPW, bool GATED = false>
__device__ __forceinline__ void epilogue_wide(const f32x16 (&acc)[RPW], unsigned char* patch, T* y_img, int Cout, int Ho, int Wo,
                                              int slab, int lane, int x0, int gy0, int row_step, float slope, int ypitch, const GateRsrc* g = nullptr) {
  // (pitched rows: an 8-pixel segment that starts inside the logical row is stored whole — its tail lands in the row's own
  // pitch padding, which no consumer depends on)
  const int px = lane & 31, kg = lane >> 5;
  const bool odd = px & 1;
  const uint32_t sel = odd ? 0x03020706u : 0x05040100u;
  const uint32_t plane2 = (uint32_t)(Ho * ypitch) * 2u;
  __amdgpu_buffer_rsrc_t yr = __builtin_amdgcn_make_buffer_rsrc(y_img, 0, (uint32_t)Cout * plane2, 0x00020000);
  // write side: this lane's pixel pair of channel (its register pair's channel) -> patch[row][channel][pixel pair]
  unsigned char* wbase = patch + (4 * kg + (odd ? 1 : 0)) * EPI_PITCH + (px & ~1) * 2;
  // read side: lane -> (channel l/4 of a 16-channel half, 8-pixel segment l%4)
  const int rc = lane >> 2, seg = lane & 3;
  const unsigned char* rbase = patch + rc * EPI_PITCH + seg * 16;
  const int gx = x0 + seg * 8;
  const uint32_t goff = (gx < Wo) ? ((uint32_t)(slab * 32 + rc) * plane2 + (uint32_t)gx * 2u) : 0x80000000u;
#pragma unroll
  for (int rb = 0; rb < RPW; rb += 2) {
    const int nrr = (rb + 2 <= RPW) ? 2 : 1;         // (compile-time after unrolling: an odd RPW ends on a single row)
    // gate operands of this row pair: issued before the transposition so that they arrive under it (rows >= Ho read the next
    // plane or zeros and are not stored)
    u32x4 ga[2][2], gq[2][2];
    if constexpr (GATED) {
#pragma unroll
      for (int rr = 0; rr < 2; ++rr)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          if (rr >= nrr) continue;
          const uint32_t off = goff + (uint32_t)(h * 16) * plane2 + (uint32_t)((gy0 + (rb + rr) * row_step) * ypitch) * 2u;
          ga[rr][h] = __builtin_amdgcn_raw_buffer_load_b128(g->ar, off, 0, 0);
          gq[rr][h] = __builtin_amdgcn_raw_buffer_load_b128(g->mr, off, 0, 0);
        }
    }
#pragma unroll
    for (int rr = 0; rr < 2; ++rr)
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        if (rr >= nrr) continue;
        constexpr int RMAX = RPW - 1;
        const int ri = (rb + rr < RPW) ? rb + rr : RMAX;
        float v0 = acc[ri][2 * j], v1 = acc[ri][2 * j + 1];
        v0 = fmaxf(v0, v0 * slope); v1 = fmaxf(v1, v1 * slope);
        const uint32_t p = pack2<T>(v0, v1);
        const uint32_t recv = (uint32_t)__builtin_amdgcn_mov_dpp((int)p, 0xB1, 0xF, 0xF, true);
        *reinterpret_cast<uint32_t*>(wbase + (rr * 32 + (int)epilogue_choff(j)) * EPI_PITCH) = __builtin_amdgcn_perm(recv, p, sel);
      }
#pragma unroll
    for (int rr = 0; rr < 2; ++rr) {
      if (rr >= nrr) continue;
      const int gy = gy0 + (rb + rr) * row_step;               // uniform
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        u32x4 v = *reinterpret_cast<const u32x4*>(rbase + (rr * 32 + h * 16) * EPI_PITCH);
        const uint32_t off = goff + (uint32_t)(h * 16) * plane2 + (uint32_t)(gy * ypitch) * 2u;
        if constexpr (GATED) {
          const u32x4 a = ga[rr][h], q = gq[rr][h];
          v.x = gate2<T>(v.x, a.x, q.x, g->slope); v.y = gate2<T>(v.y, a.y, q.y, g->slope);
          v.z = gate2<T>(v.z, a.z, q.z, g->slope); v.w = gate2<T>(v.w, a.w, q.w, g->slope);
        }
        if (gy < Ho) __builtin_amdgcn_raw_buffer_store_b128(v, yr, off, 0, 0);
      }
    }
  }
}

// MTW: 32-channel output blocks per workgroup (1, 2, 4) = waves along Cout;  RPW: tile rows per wave
// (tile height TH = (4/MTW)*RPW);  S: stride;  NOCTS: channel octets per chunk (4 = 32 channels, 2 = 16);
// D: compile-time dilation (see margin_of);  GEN: the staged rows need not be 16-byte aligned (W % 8 != 0, or x is an
// odd channel slice): the 8-pixel group that would cross the end of its image row is loaded SHIFTED LEFT so that it
// ends at the row end (gfx950 executes the 2-byte-aligned 16-byte load; tools/unaligned_b128_probe.hip), and the
// shift is undone by the LDS entry index each transposed pixel is written to — no load ever leaves its row, so
// nothing depends on what follows the buffer.
// ONE: Cin <= 16 = a single 16-channel chunk (the RGB and 16-channel layers at full / half resolution: pure
// bandwidth, thousands of short-lived workgroups): no chunk loop, nothing loop-carried, so the register budget allows
// four workgroups per CU to overlap their load / store latencies.
//
// XL / YC8 (round 3) — operands in the CHANNEL-OCTET layout [n][c/8][y][x][8] ("C8": the 8 channels of a pixel are one
// 16-byte entry — exactly one LDS entry of the tile image and one k-octet of an MFMA operand):
//   XL = 1: x is a C8 tensor slice (x8: first octet plane, n8oct octets);  XL = 2: the C8 slice is followed, in the K order
//   of the packed weights, by an NCHW tail (x, Cin planes: the cost volume and the flow, whose producers write planes);
//   XL = 0: NCHW only (the round-1/2 path, unchanged).  YC8: y is written as C8 octets.
// Staging a C8 chunk needs no registers, no transposition and no LDS store instructions: the tile image is filled by
// LDS-DMA (`buffer_load_dwordx4 ... lds`, 64 entries = 1 KB per wave instruction; lanes outside the image or past the last
// octet get zeros from the descriptor's bounds check), into the OTHER of two LDS buffers while the matrix phase of the
// current chunk runs — one barrier per chunk, every load of chunk c+1 in flight under the MFMAs of chunk c.  The round-2
// ablations (profiles/r03_conv_ablate.txt) had shown the NCHW form's phases adding up instead of overlapping: 565->128
// at 96x320 took 266 us against 181 us for its matrix phase alone and 115 us for its staging alone.
// The YC8 epilogue is 8-byte stores straight from the accumulators (a lane holds 4 consecutive channels of its pixel
// per octet): no LDS patch, no lane exchange.
//
// Measured and NOT kept (round 3; evidence: profiles/r03_conv_ablate.txt, r03_conv_timeline.txt, r03_conv_pmc_il_power.txt).
// The ablations showed the two phases of the NCHW loop adding up instead of overlapping (565->128 at 96x320: 266 us; matrix
// phase alone 181 us, staging alone 115 us), so two restructurings were built, tested bit-identical, and measured:
//   * "ping-pong" workgroups of 8 waves whose two 4-wave groups alternate matrix and staging turns by construction: 271 us.
//     The in-kernel time stamps say why: ONE wave per SIMD issues a 32x32x16 MFMA every ~40 cycles, not every 32 — a single
//     wave's MFMA stream caps at ~80 % of the pipe; only two waves per SIMD that are both multiplying fill it.
//   * interleaved staging (two LDS buffers, chunk c + 1 transposed and written between the MFMAs of chunk c, loads of chunk
//     c + 2 — or c + 3 from a second register set — right behind them, one barrier per chunk): 255-270 us, and 3.347 vs
//     3.349 ms for the whole step (tools/ab_bench.py, same box).  PMC: the MFMA utilisation rises (60.5 -> 66.1 %) and the clock
//     falls (1.88 -> 1.70 GHz) — utilisation x clock is constant.  With an all-zero input (no operand toggling) the same
//     kernels hold 2.1 GHz and the interleaved form IS 10 % faster.  The wide layers are POWER-bound at ~1.1 PFLOP/s on
//     random bf16 data; a better schedule buys nothing there, and the narrow layers (Cout <= 32) did not move either
//     (their ~2.8 TB/s is set by the 96-byte row segments of a 32-pixel tile: two cache lines each).
// N16 (round 3; C8 input, 3x3, dilation 1, stride 1, MTW = 1, 32-channel chunks): layers with at most 16 output channels (the
// 563->2 / 184->3 heads, 176->8, 160->16) on `v_mfma_f32_16x16x32` instead of 32x32x16 — a 16-channel output block, so half the
// matrix work of the 32-channel block that is mostly padding for them (ablation, 568->2 at 96x320: matrix phase alone 58 us,
// staging alone 45 us, together 73 us).  blockIdx.y = the 16-channel block; weights packed by pack_weights_kmap16_kernel.
// TO (round 5): storage type of y when it differs from the operands' (the `pyramid_dtype` option: fp16 features and weights, bf16
// decoder buffers — the 1x1 projection of the pyramid features multiplies in fp16 and rounds its fp32 sums ONCE, to bf16).
template <typename T, int MTW, int RPW, int S, int NOCTS, int D, bool GEN, bool ONE, int XL, bool YC8, bool N16, typename TO, typename G>
__device__ __forceinline__
void conv_body(const T* __restrict__ x, long long xbs, const T* __restrict__ wp, const float* __restrict__ bias,
               TO* __restrict__ y, long long ybs, int Cin, int Cout, int H, int W, int Ho, int Wo, int d_rt,
               int tiles_x, int tiles_y, float slope, const T* __restrict__ x8, long long x8bs, int n8oct, int xpitch, int ypitch, const G gate) {
  static_assert(!G::on || (XL == 0 && !YC8 && !N16 && sizeof(TO) == sizeof(T)), "gated epilogue: NCHW operands of one type");
  static_assert(!G::split || (XL == 1 && YC8 && !N16 && D == 1 && S == 1), "split epilogue: octet operands, 3x3, dilation 1, stride 1");
  static_assert(!G::init || (N16 && XL == 1), "accumulator initialisation from a partial: the 16-channel kernel");
  // xpitch / ypitch (round 5): elements between consecutive rows of the NCHW operands x / y (plane stride = rows * pitch).  The
  // LOGICAL width stays W / Wo: with a pitch that is a multiple of 8 every row is 16-byte aligned whatever W is, so ragged
  // pyramid levels (KITTI's native 375x1242: W = 621, 311, 156, 78, 39, 20) take the aligned staging (!GEN) and the 16-byte
  // epilogue, and nothing depends on what the pitch padding holds (staging masks it, the outputs' padding is never read as data).
  // C8 operands have no pitch: a pixel is one 16-byte entry, their rows are aligned for every W.
  static_assert(XL == 0 || (D >= 0 && !GEN && !ONE), "C8 input: compile-time dilation, aligned rows");
  static_assert(!N16 || (XL == 1 && MTW == 1 && NOCTS == 4 && D == 1 && S == 1 && RPW >= 2), "N16: C8 input, 3x3, dilation 1, stride 1");
  constexpr int ntaps = (D == 0) ? 1 : 9;
  constexpr int marg = marg_eff(XL, D);
  constexpr int KS = NOCTS / 2, KCH = NOCTS * 8;     // k-steps of 16 channels / channels per chunk
  constexpr int RG = 4 / MTW, TH = RG * RPW;
  constexpr int XW = xw_eff(XL, S, D);
  constexpr int XWP = xwp_eff(XL, S, D);             // LDS row pitch in entries (with the rotated pixel slots of swz: the
                                                     // staging writes of a 16-lane phase hit 16 different bank quads)
  // PH (compile-time dilation >= 2): ROW-PHASE decomposition.  A workgroup's TH output rows are D image rows apart
  // (rows y0 + D*r of one phase y0 % D), so the three kernel rows read ADJACENT staged rows — the vertical halo is 2
  // rows instead of 2*D (dilation 8: 10 staged rows per 8 output rows instead of 24), and only the horizontal taps
  // keep the dilation, as a window shift inside the LDS row.
  constexpr bool PH = (D >= 2 && S == 1);
  constexpr int RS = PH ? D : 1;                     // image rows between consecutive tile rows
  constexpr int DV = PH ? 1 : D;                     // staged rows between consecutive kernel rows (D >= 0)
  const int d = (D >= 0) ? D : d_rt;
  // experiments (upf_conv_set_option "ablate", compile-time dilations only): 1 = no matrix phase, 2 = no x loads, 4 = no LDS staging writes, 8 = no weight loads
  const int abl = (D >= 0) ? d_rt : 0;
  const int rows = (D >= 0) ? S * (TH - 1) + 2 * DV + 1 : S * (TH - 1) + 2 * d + 1;         // staged input rows
  extern __shared__ __attribute__((aligned(16))) uint4 xs[];   // [octet][rows][XWP] entries of 8 channels x 1 pixel

  const int bid = xcd_remap(blockIdx.x, gridDim.x);
  const int tx = bid % tiles_x, ty = (bid / tiles_x) % tiles_y, n = bid / (tiles_x * tiles_y);
  const int x0 = tx * TW, y0 = PH ? (ty / RS) * (RS * TH) + ty % RS : ty * TH;     // PH: tiles_y counts (row block, phase) pairs
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int px = lane & 31, kg = lane >> 5;          // MFMA operand lane: column / row index, k-octet within the k-step
  const int cb = wave % MTW, rg = wave / MTW;
  const int slab = blockIdx.y * MTW + cb;            // this wave's 32-channel output block
  // K order of the packed weights: [C8 part, padded to 32 channels | NCHW part, padded to 32 channels]
  const int c8p = (XL >= 1) ? pad32(n8oct * 8) : 0;
  const int cip = c8p + ((XL == 1) ? 0 : pad32(Cin));
  const int n8c = c8p / KCH;                                  // chunks staged from the C8 slice
  const int nchunks = ONE ? 1 : cip / KCH, nksteps = cip / 16;
  const int HW = H * W;                               // entries per octet plane of a C8 operand
  const int HWx = H * xpitch;                         // elements per channel plane of the NCHW operand

  // buffer descriptor over this image's Cin input planes: rows/cols outside the image get offset
  // 0x80000000, channel planes >= Cin fall off the end -> the hardware returns the zero padding
  const uint32_t plane = (uint32_t)HWx * 2u;
  __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(x + (size_t)n * xbs), 0, (uint32_t)Cin * plane, 0x00020000);
  __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(wp), 0, (uint32_t)ntaps * (uint32_t)(N16 ? (Cout + 15) / 16 * 16 : pad32(Cout)) * (uint32_t)cip * 2u, 0x00020000);

  // accumulators start at the bias (channels >= Cout read 0 through the descriptor)
  __amdgpu_buffer_rsrc_t br = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(bias), 0, (uint32_t)Cout * 4u, 0x00020000);
  f32x16 acc[N16 ? 1 : RPW];
  f32x4 acc16[N16 ? RPW : 1][2];                     // N16: [tile row][pixel half], channels 4 * (lane / 16) + i of pixel lane % 16 (+ 16)
  const int p16 = lane & 15, ko = lane >> 4;         // N16 operand lane: pixel of its half / k-octet (= output channel quad)
  if constexpr (N16 && G::init) {
    // the accumulators start at the fp32 partial the merged pass left for this layer (its bias is in there): one 16-byte load per
    // (tile row, pixel half); pixels outside the image and channel quads past Cout start at 0 (they are never stored)
    __amdgpu_buffer_rsrc_t pr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(gate.part + (size_t)n * gate.pbs), 0,
                                                                  (uint32_t)(H * W) * (uint32_t)gate.ppitch * 4u, 0x00020000);
    const int gyi = y0 + RPW * rg;
#pragma unroll
    for (int r = 0; r < RPW; ++r)
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int gx = x0 + p16 + 16 * h, gy = gyi + r;
        const uint32_t off = (gx < W && gy < H && slab * 16 + 4 * ko < Cout) ? (uint32_t)(((gate.coff >> 2) + slab * 4 + ko) * (H * W) + gy * W + gx) * 16u : 0x80000000u;
        acc16[r][h] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(pr, off, 0, 0));
      }
  } else if constexpr (N16) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float bv = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(br, (uint32_t)(slab * 16 + 4 * ko + i) * 4u, 0, 0));
#pragma unroll
      for (int r = 0; r < RPW; ++r) { acc16[r][0][i] = bv; acc16[r][1][i] = bv; }
    }
  } else {
#pragma unroll
  for (int e = 0; e < 16; ++e) {
    const float bv = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(br, (uint32_t)(slab * 32 + (e & 3) + 8 * (e >> 2) + 4 * kg) * 4u, 0, 0));
#pragma unroll
    for (int r = 0; r < RPW; ++r) acc[r][e] = bv;
  }
  }

  // x staging tasks: (octet, row, 8-pixel group) -> 8 loads (8 channel rows), 32 perms, 8 LDS entries
  constexpr int ngroups = XW / 8;
  const int ntasks = NOCTS * rows * ngroups;
  // task t -> buffer-load offset of channel 0 of its octet in chunk 0 (0x80000000 = outside the image), the LDS
  // entry it fills, and (GEN) the pixels by which the load window is shifted left to end at the row end
  auto task_geom = [&](int t, uint32_t& off, int& dst, int& sh) {
    const int oct = t / (rows * ngroups), rem = t - oct * (rows * ngroups), r = rem / ngroups, g = rem - r * ngroups;   // group fastest: coalesced loads
    const int gy = PH ? y0 + (r - 1) * RS : S * y0 - d + r, gx = S * x0 - marg + 8 * g;
    const bool in = (t < ntasks) && gy >= 0 && gy < H && gx >= 0 && gx < W && !(abl & 2);
    // the group that straddles the row end: GEN loads it shifted left by sh; !GEN (aligned pitch) loads it in place and
    // stage_store zeroes its sh trailing pixels
    sh = (in && gx + 8 > W) ? gx + 8 - W : 0;
    off = in ? ((uint32_t)((oct * 8) * HWx + gy * xpitch + gx - (GEN ? sh : 0)) * 2u) : 0x80000000u;
    dst = ((oct * rows + r) * XWP + 8 * g) * 8 + ((g >> 1) & 7);   // entry index * 8 + slot rotation of the group
  };
  auto task_load = [&](uint32_t off, int cc, u32x4 (&ch)[8]) {
    const uint32_t o = off + (uint32_t)cc * KCH * plane;                           // stays >= 2^31 for outside tasks
#pragma unroll
    for (int k = 0; k < 8; ++k) ch[k] = __builtin_amdgcn_raw_buffer_load_b128(xr, o + k * plane, 0, 0);
  };
  auto task_store = [&](int enc, int sh, const u32x4 (&ch)[8]) { if (!(abl & 4)) stage_store<GEN>(xs, enc, sh, ch); };
  // this thread's first x task of chunk cc+1 is loaded into registers BEFORE the matrix phase of chunk cc and
  // lands in LDS after it
  constexpr bool PRE = (XL == 2) || (XL == 0 && !(MTW == 1 && RPW == 4 && NOCTS == 4));     // (that one would spill)
  uint32_t off0 = 0x80000000u; int dst0 = 0, sh0 = 0;
  u32x4 pre[8];
  if constexpr (PRE) {
    task_geom(tid, off0, dst0, sh0);
    if constexpr (XL == 0) task_load(off0, 0, pre);
  }

  // this lane's A operands (row px of the weight tile, k-octet kg of each k-step) for every tap of a chunk
  uint4 wa[ntaps][KS];
  auto wload = [&](int cc, int tap, int ks) {
    // (N16: one 1 KB operand per 32-channel chunk and tap, [16-channel block][chunk][tap][lane]; ks is 0)
    const uint32_t idx = N16 ? (uint32_t)((slab * nchunks + cc) * ntaps + tap) : (uint32_t)((slab * nksteps + cc * KS + ks) * ntaps + tap);
    const uint32_t off = (cc < nchunks && !(abl & 8)) ? idx * 1024u + (uint32_t)lane * 16u : 0x80000000u;
    return __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(wr, off, 0, 0));
  };
  constexpr int KSW = N16 ? 1 : KS;                  // weight operands per tap and chunk
#pragma unroll
  for (int tap = 0; tap < ntaps; ++tap)
#pragma unroll
    for (int ks = 0; ks < KSW; ++ks) wa[tap][ks] = wload(0, tap, ks);

  constexpr bool REUSE = (S == 1 && D >= 1 && (RPW + 2 * DV) * 3 < 9 * RPW);
  // Round 4: with a pure C8 input (XL == 1) the tile image is filled by LDS-DMA — contiguous 1 KB pieces, conflict-free whatever
  // the layout — so its rows are stored LINEARLY (position = column): a read window of 16 consecutive columns is then 256
  // contiguous bytes = 16 different bank quads from ANY start.  The slot rotation of swz() exists for the register-staged
  // (NCHW) writes; it costs the shifted read windows a 2-way conflict per 16-lane pass (tools/lds_b128_probe.hip: the 50 % of
  // LDS-active cycles that SQ_LDS_BANK_CONFLICT reports for every variant are real).  The odd row pitch (XWP = 51 / 85 entries)
  // keeps rows apart.  UPF_C8_LINEAR=0 restores the rotated image for A/B runs.
#ifndef UPF_C8_LINEAR
#define UPF_C8_LINEAR 1
#endif
  constexpr bool LIN = (XL == 1) && (UPF_C8_LINEAR != 0);
  auto pos_of = [](int col) { return LIN ? col : swz(col); };
  const int colx[3] = {pos_of(marg + px - (D > 0 ? D : 0)), pos_of(marg + px), pos_of(marg + px + (D > 0 ? D : 0))};   // REUSE windows

  // ---- the matrix phase of chunk cc on the tile image at xb (also fetches the weights of chunk cc + 1)
  // (N16, measured and not kept: the four waves of a workgroup multiply by the SAME weights, and their 4 x 9 KB of operand loads
  // per chunk are more L2 traffic than the 22 KB of x — 1.2 GB against 0.28 GB of input for 568->2.  Passing the operands through
  // LDS once per workgroup needs 18 KB more LDS: one workgroup per CU instead of two, 66 -> 109 us.)
  auto matrix_phase = [&](const uint4* __restrict__ xb, int cc) {
    // matrix phase at raised wave priority: the CU's other workgroup is usually in its staging phase, and the arbiter then
    // serves the MFMA stream first (A/B on one box, three runs each: 1165 -> 1173 frame-pairs/s)
    __builtin_amdgcn_s_setprio(2);

    if (abl & 1) {
    } else if constexpr (N16) {
      // staged row sr feeds output rows sr - ky; per row three windows (kx) for each 16-pixel half, each used by up to three MFMAs
      constexpr int NR = RPW + 2;
      const int c16[2][3] = {{pos_of(marg + p16 - 1), pos_of(marg + p16), pos_of(marg + p16 + 1)}, {pos_of(marg + p16 + 15), pos_of(marg + p16 + 16), pos_of(marg + p16 + 17)}};
      uint4 bq[2][6];
      auto bload = [&](int sr, uint4 (&b)[6]) {
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
          for (int kx = 0; kx < 3; ++kx) b[h * 3 + kx] = xb[(ko * rows + RPW * rg + sr) * XWP + c16[h][kx]];
      };
      bload(0, bq[0]);
#pragma unroll
      for (int sr = 0; sr < NR; ++sr) {
        if (sr + 1 < NR) bload(sr + 1, bq[(sr + 1) & 1]);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int kx = 0; kx < 3; ++kx)
#pragma unroll
          for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int ky = 0; ky < 3; ++ky)
              if (sr - ky >= 0 && sr - ky < RPW)
                acc16[sr - ky][h] = Mma16<T>::mma(wa[ky * 3 + kx][0], bq[sr & 1][h * 3 + kx], acc16[sr - ky][h]);
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
          if (sr == ky + RPW - 1) {
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) wa[ky * 3 + kx][0] = wload(cc + 1, ky * 3 + kx, 0);
          }
        __builtin_amdgcn_sched_barrier(0);
      }
    } else if constexpr (REUSE) {
      // staged row sr of this wave's strip feeds output rows r = sr - ky*DV.  The three windows (kx) of row sr+1 are
      // read from LDS while the (up to 9*KS) MFMAs of row sr run: left to itself hipcc issues each ds_read right
      // before its first use and the ~130-cycle LDS latency stalls the matrix pipe twice per row.
      constexpr int NR = RPW + 2 * DV;
      uint4 bq[2][3 * KS];
      auto bload = [&](int sr, uint4 (&b)[3 * KS]) {
#pragma unroll
        for (int kx = 0; kx < 3; ++kx)
#pragma unroll
          for (int ks = 0; ks < KS; ++ks) b[kx * KS + ks] = xb[((2 * ks + kg) * rows + RPW * rg + sr) * XWP + colx[kx]];
      };
      constexpr bool PIPEB = !ONE;                   // (the single-chunk variants run 4 workgroups per CU on 128 registers)
      if constexpr (PIPEB) bload(0, bq[0]);
#pragma unroll
      for (int sr = 0; sr < NR; ++sr) {
        if constexpr (PIPEB) { if (sr + 1 < NR) bload(sr + 1, bq[(sr + 1) & 1]); }
        else bload(sr, bq[sr & 1]);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int kx = 0; kx < 3; ++kx)
#pragma unroll
          for (int ks = 0; ks < KS; ++ks)
#pragma unroll
            for (int ky = 0; ky < 3; ++ky)
              if (sr - ky * DV >= 0 && sr - ky * DV < RPW)
                acc[sr - ky * DV] = Mma32<T>::mma(wa[ky * 3 + kx][ks], bq[sr & 1][kx * KS + ks], acc[sr - ky * DV]);
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
          if (sr == ky * DV + RPW - 1) {              // kernel row ky is finished: fetch the next chunk's
#pragma unroll
            for (int kx = 0; kx < 3; ++kx)
#pragma unroll
              for (int ks = 0; ks < KS; ++ks) wa[ky * 3 + kx][ks] = wload(cc + 1, ky * 3 + kx, ks);
          }
        __builtin_amdgcn_sched_barrier(0);
      }
    } else {
#pragma unroll
      for (int tap = 0; tap < ntaps; ++tap) {
        const int ky = (ntaps == 1) ? 1 : tap / 3, kx = (ntaps == 1) ? 1 : tap - 3 * (tap / 3);
        // shifted window: output pixel (row, px) reads staged entry (S*row + ky*d, marg + S*px + (kx-1)*d)
        const int col = pos_of(marg + S * px + (kx - 1) * d);
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
#pragma unroll
          for (int r = 0; r < RPW; ++r) {
            const uint4 b = xb[((2 * ks + kg) * rows + S * (RPW * rg + r) + ky * (PH ? 1 : d)) * XWP + col];   // (PH: kernel rows are adjacent staged rows)
            acc[r] = Mma32<T>::mma(wa[tap][ks], b, acc[r]);
          }
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) wa[tap][ks] = wload(cc + 1, tap, ks);
      }
    }
    __builtin_amdgcn_s_setprio(0);
  };

  if constexpr (XL >= 1) {
    // ---- C8 input: two LDS buffers, chunk cc + 1 staged under the matrix phase of chunk cc
    constexpr int rowsC = S * (TH - 1) + 2 * DV + 1;
    constexpr int EB = NOCTS * rowsC * XWP, EBP = (EB + 63) & ~63;      // entries per buffer (whole 64-entry DMA pieces)
    constexpr int NSL = (EBP / 64 + 3) / 4;                              // DMA pieces per wave and chunk
    constexpr int DH = (D > 0) ? D : 0;
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    __amdgpu_buffer_rsrc_t x8r = __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(x8 + (size_t)n * x8bs), 0, (uint32_t)n8oct * (uint32_t)HW * 16u, 0x00020000);
    // this lane's entry of each of its wave's pieces: LDS position -> (octet, row, column) -> byte offset in chunk 0
    uint32_t voff8[NSL];
#pragma unroll
    for (int j = 0; j < NSL; ++j) {
      const int p = (j * 4 + wave_u) * 64 + lane;
      const int oct = p / (rowsC * XWP), rem = p - oct * (rowsC * XWP), r = rem / XWP, pos = rem - r * XWP;
      const int grp = pos & ~7, col = LIN ? pos : (grp | ((pos - (grp >> 4)) & 7));   // inverse of the position map
      const int gy = PH ? y0 + (r - 1) * RS : S * y0 - DV + r, gx = S * x0 - marg + col;
      const bool in = p < EB && pos < XW && col >= marg - DH && col < marg + S * TW + DH && gy >= 0 && gy < H && gx >= 0 && gx < W && !(abl & 2);
      voff8[j] = in ? (uint32_t)((oct * HW + gy * W + gx) * 16) : 0x80000000u;
    }
    auto land = [&](int cc) {                           // NCHW chunks: registers -> transposed LDS entries
      if constexpr (XL == 2) {
        if (cc >= n8c && cc < nchunks) {
          uint4* xb = xs + (cc & 1) * EBP;
          if (tid < ntasks) stage_store<false>(xb, dst0, sh0, pre);
          for (int t = tid + NTHREADS; t < ntasks; t += NTHREADS) {
            uint32_t off; int dsti, sh;
            task_geom(t, off, dsti, sh);
            u32x4 ch[8];
            task_load(off, cc - n8c, ch);
            stage_store<false>(xb, dsti, sh, ch);
          }
        }
      }
    };
    // (iteration -1 is the prologue: chunk 0 is staged, nothing multiplied)
    for (int cc = -1; cc < nchunks; ++cc) {
      const int nx = cc + 1;                              // chunk nx -> buffer nx & 1
      if (nx < n8c) {
        const uint32_t cadd = (uint32_t)nx * (uint32_t)(NOCTS * 16) * (uint32_t)HW;     // (offsets >= 2^31 stay there: outside)
#pragma unroll
        for (int j = 0; j < NSL; ++j) {
          const int piece = j * 4 + wave_u;
          if (piece * 64 < EBP)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(x8r, (__attribute__((address_space(3))) void*)(xs + ((nx & 1) * EBP + piece * 64)), 16, (int)(voff8[j] + cadd), 0, 0, 0);   // (the explicit int cast matters: without it hipcc's HOST pass silently drops this kernel template and the launch fails to link)
        }
      } else if constexpr (XL == 2) {
        if (nx < nchunks) task_load(off0, nx - n8c, pre);
      }
      if (cc >= 0) matrix_phase(xs + (cc & 1) * EBP, cc);
      land(nx);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the DMA pieces of chunk cc + 1 have landed
      __syncthreads();                                    // ... in every wave, and chunk cc is fully consumed
    }
  } else {
  for (int cc = 0; cc < nchunks; ++cc) {
      __syncthreads();                                 // previous chunk fully consumed
      // ---- stage the x tile (+halo) of channels [KCH*cc, KCH*cc + KCH)
      if constexpr (PRE) { if (tid < ntasks) task_store(dst0, sh0, pre); }
      for (int t = tid + (PRE ? NTHREADS : 0); t < ntasks; t += NTHREADS) {
        uint32_t off; int dsti, sh;
        task_geom(t, off, dsti, sh);
        u32x4 ch[8];
        task_load(off, cc, ch);
        task_store(dsti, sh, ch);
      }
      __syncthreads();
      if constexpr (PRE) { if (cc + 1 < nchunks) task_load(off0, cc + 1, pre); }
      matrix_phase(xs, cc);
    }
  }

  // ---- epilogue (bias is already in the accumulators)
  const int gy0 = __builtin_amdgcn_readfirstlane(y0 + RPW * rg * RS);
  if constexpr (N16) {
    // lane: channels slab*16 + 4*ko + i of pixels x0 + p16 (+ 16).  YC8: the quad is half an octet entry -> one 8-byte store;
    // NCHW (the 2- / 3-channel heads): one 2-byte store per channel, channels >= Cout fall off the descriptor.
    const uint32_t plane16 = (uint32_t)(Ho * Wo) * 16u, plane2 = (uint32_t)(Ho * ypitch) * 2u;
    __amdgpu_buffer_rsrc_t yr = __builtin_amdgcn_make_buffer_rsrc(y + (size_t)n * ybs, 0, YC8 ? (uint32_t)((Cout + 7) / 8) * plane16 : (uint32_t)Cout * plane2, 0x00020000);
#pragma unroll
    for (int r = 0; r < RPW; ++r) {
      const int gy = gy0 + r;                        // uniform
      if (gy < Ho) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const int gx = x0 + p16 + 16 * h;
          float v[4];
#pragma unroll
          for (int i = 0; i < 4; ++i) { v[i] = acc16[r][h][i]; v[i] = fmaxf(v[i], v[i] * slope); }
          if constexpr (YC8) {
            const uint32_t off = (gx < Wo) ? (uint32_t)(slab * 2 + (ko >> 1)) * plane16 + (uint32_t)(gy * Wo + gx) * 16u + (uint32_t)(ko & 1) * 8u : 0x80000000u;
            u32x2 o;
            o.x = pack2<TO>(v[0], v[1]); o.y = pack2<TO>(v[2], v[3]);
            __builtin_amdgcn_raw_buffer_store_b64(o, yr, off, 0, 0);
          } else {
            const uint32_t off = (gx < Wo) ? (uint32_t)(slab * 16 + 4 * ko) * plane2 + (uint32_t)(gy * ypitch + gx) * 2u : 0x80000000u;
            const uint32_t p01 = pack2<TO>(v[0], v[1]), p23 = pack2<TO>(v[2], v[3]);
            __builtin_amdgcn_raw_buffer_store_b16((unsigned short)p01, yr, off, 0, 0);
            __builtin_amdgcn_raw_buffer_store_b16((unsigned short)(p01 >> 16), yr, off + plane2, 0, 0);
            __builtin_amdgcn_raw_buffer_store_b16((unsigned short)p23, yr, off + 2u * plane2, 0, 0);
            __builtin_amdgcn_raw_buffer_store_b16((unsigned short)(p23 >> 16), yr, off + 3u * plane2, 0, 0);
          }
        }
      }
    }
    return;
  }
  if constexpr (!N16) {
  if constexpr (YC8) {
    // y as octets [c/8][Ho][Wo][8]: register e of this lane is channel (e&3) + 8*(e>>2) + 4*kg of the wave's 32-block, so
    // registers 4g..4g+3 are bytes [8*kg, 8*kg + 8) of the entry (octet g, pixel): one 8-byte store; the two half-waves
    // together write 32 whole entries = 512 contiguous bytes per (octet, row).  Octets past ceil(Cout / 8) fall off the
    // descriptor; the channels that pad the last octet are exact zeros (zero weights, zero bias).
    const uint32_t plane16 = (uint32_t)(Ho * Wo) * 16u;
    int cout_y = Cout;                                // channels that leave as octets
    if constexpr (G::split) cout_y = gate.cmain;
    __amdgpu_buffer_rsrc_t yr = __builtin_amdgcn_make_buffer_rsrc(y + (size_t)n * ybs, 0, (uint32_t)((cout_y + 7) / 8) * plane16, 0x00020000);
    const int gx = x0 + px;
    const uint32_t lane_off = (gx < Wo) ? (uint32_t)(slab * 4) * plane16 + (uint32_t)gx * 16u + (uint32_t)kg * 8u : 0x80000000u;
    __amdgpu_buffer_rsrc_t pr;
    if constexpr (G::split)
      pr = __builtin_amdgcn_make_buffer_rsrc(gate.part + (size_t)n * gate.pbs, 0, (uint32_t)(Ho * Wo) * (uint32_t)gate.ppitch * 4u, 0x00020000);
#pragma unroll
    for (int r = 0; r < RPW; ++r) {
      const int gy = gy0 + r * RS;                   // uniform
      if (gy < Ho) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          if constexpr (G::split) {
            // registers 4g .. 4g+3 = channels c0 .. c0+3, c0 = 32*slab + 8*g + 4*kg: past cmain (uniform: cmain % 8 == 0) they are
            // partial pre-activations of later layers -> raw fp32, quad plane (c0 - cmain) / 4 of the partial
            if (slab * 32 + 8 * g >= gate.cmain) {
              const int c0 = slab * 32 + 8 * g + 4 * kg;
              const uint32_t off = (gx < Wo && c0 < Cout) ? (uint32_t)(((c0 - gate.cmain) >> 2) * (Ho * Wo) + gy * Wo + gx) * 16u : 0x80000000u;
              f32x4 pv;
#pragma unroll
              for (int q = 0; q < 4; ++q) pv[q] = acc[r][4 * g + q];
              __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, pv), pr, off, 0, 0);
              continue;
            }
          }
          float v[4];
#pragma unroll
          for (int q = 0; q < 4; ++q) { v[q] = acc[r][4 * g + q]; v[q] = fmaxf(v[q], v[q] * slope); }
          u32x2 o;
          o.x = pack2<TO>(v[0], v[1]); o.y = pack2<TO>(v[2], v[3]);
          __builtin_amdgcn_raw_buffer_store_b64(o, yr, lane_off + (uint32_t)g * plane16 + (uint32_t)(gy * Wo) * 16u, 0, 0);
          if constexpr (G::dual) {
            __amdgpu_buffer_rsrc_t yr2 = __builtin_amdgcn_make_buffer_rsrc(gate.y2 + (size_t)n * gate.y2bs, 0, (uint32_t)((cout_y + 7) / 8) * plane16, 0x00020000);
            __builtin_amdgcn_raw_buffer_store_b64(o, yr2, lane_off + (uint32_t)g * plane16 + (uint32_t)(gy * Wo) * 16u, 0, 0);
          }
        }
      }
    }
    return;
  }
  if constexpr (!GEN) {
    // uniform: 16-byte stores through a per-wave LDS patch when the OUTPUT rows are 16-byte aligned (pitch, base, batch stride)
    if (((ypitch & 7) | (int)(ybs & 7) | (int)(reinterpret_cast<uintptr_t>(y) & 15)) == 0) {
      __syncthreads();                               // every wave is done with the x tile
      if constexpr (G::on) {
        const GateRsrc gr = gate_init<T>(gate, n, (uint32_t)Cout * (uint32_t)(Ho * ypitch) * 2u);
        epilogue_wide<TO, RPW, true>(acc, reinterpret_cast<unsigned char*>(xs) + wave * EPI_WAVE_BYTES, y + (size_t)n * ybs, Cout, Ho, Wo,
                                    slab, lane, x0, gy0, RS, slope, ypitch, &gr);
      } else {
        epilogue_wide<TO, RPW>(acc, reinterpret_cast<unsigned char*>(xs) + wave * EPI_WAVE_BYTES, y + (size_t)n * ybs, Cout, Ho, Wo,
                              slab, lane, x0, gy0, RS, slope, ypitch);
      }
      return;
    }
  }
  // (always the general form: an aligned input says nothing about the output's width — a pitched x with an un-pitched odd-width y)
  Epilogue ep;
  epilogue_init<TO, true>(ep, y + (size_t)n * ybs, Cout, Ho, Wo, slab, lane, x0, ypitch);
  GateRsrc gr;
  if constexpr (G::on) gr = gate_init<T>(gate, n, (uint32_t)Cout * ep.plane2);
#pragma unroll
  for (int r = 0; r < RPW; ++r) {
    if (gy0 + r * RS < Ho) {
#pragma unroll
      for (int j = 0; j < 8; ++j)
        epilogue_store<TO, true, G::on>(ep, acc[r][2 * j], acc[r][2 * j + 1], epilogue_choff(j) * ep.plane2 + (uint32_t)((gy0 + r * RS) * ypitch) * 2u, slope, &gr);
    }
  }
  }  // !N16
}

template <typename T, int MTW, int RPW, int S, int NOCTS, int D, bool GEN, bool ONE = false, int XL = 0, bool YC8 = false, bool N16 = false, typename TO = T>
__global__ __launch_bounds__(NTHREADS, (ONE && (MTW == 1 || (MTW == 2 && S == 1))) ? 4 : 2)
void conv_kernel(const T* __restrict__ x, long long xbs, const T* __restrict__ wp, const float* __restrict__ bias,
                 TO* __restrict__ y, long long ybs, int Cin, int Cout, int H, int W, int Ho, int Wo, int d_rt,
                 int tiles_x, int tiles_y, float slope, const T* __restrict__ x8, long long x8bs, int n8oct, int xpitch, int ypitch) {
  conv_body<T, MTW, RPW, S, NOCTS, D, GEN, ONE, XL, YC8, N16, TO, NoGate>(x, xbs, wp, bias, y, ybs, Cin, Cout, H, W, Ho, Wo, d_rt, tiles_x, tiles_y, slope,
                                                                          x8, x8bs, n8oct, xpitch, ypitch, NoGate{});
}

// the same kernel with the gated epilogue (NCHW operands: the data-gradient convolutions of the dense stacks)
template <typename T, int MTW, int RPW, int S, int NOCTS, int D, bool GEN, bool ONE = false>
__global__ __launch_bounds__(NTHREADS, (ONE && (MTW == 1 || (MTW == 2 && S == 1))) ? 4 : 2)
void conv_gated_kernel(const T* __restrict__ x, long long xbs, const T* __restrict__ wp, const float* __restrict__ bias,
                       T* __restrict__ y, long long ybs, int Cin, int Cout, int H, int W, int Ho, int Wo, int d_rt,
                       int tiles_x, int tiles_y, float slope, int xpitch, int ypitch, const ActGate<T> gate) {
  conv_body<T, MTW, RPW, S, NOCTS, D, GEN, ONE, 0, false, false, T, ActGate<T>>(x, xbs, wp, bias, y, ybs, Cin, Cout, H, W, Ho, Wo, d_rt, tiles_x, tiles_y, slope,
                                                                                nullptr, 0ll, 0, xpitch, ypitch, gate);
}

// the merged narrow tail of a dense stack (SplitOut / AccInit above): octet input, octet output + fp32 partials; and the finishing
// launch of a later layer on the 16-channel instruction, its accumulators starting from the partial
template <typename T, int MTW, int RPW, int NOCTS>
__global__ __launch_bounds__(NTHREADS, 2)
void conv_split_kernel(const T* __restrict__ wp, const float* __restrict__ bias, T* __restrict__ y, long long ybs, int Cout, int H, int W,
                       int tiles_x, int tiles_y, float slope, const T* __restrict__ x8, long long x8bs, int n8oct, const SplitOut so) {
  conv_body<T, MTW, RPW, 1, NOCTS, 1, false, false, 1, true, false, T, SplitOut>(nullptr, 0ll, wp, bias, y, ybs, 0, Cout, H, W, H, W, 0, tiles_x, tiles_y, slope,
                                                                                  x8, x8bs, n8oct, W, W, so);
}
template <typename T, int RPW, bool YC8>
__global__ __launch_bounds__(NTHREADS, 2)
void conv_accinit_kernel(const T* __restrict__ wp, T* __restrict__ y, long long ybs, int Cout, int H, int W,
                         int tiles_x, int tiles_y, float slope, const T* __restrict__ x8, long long x8bs, int n8oct, const AccInit ai) {
  conv_body<T, 1, RPW, 1, 4, 1, false, false, 1, YC8, true, T, AccInit>(nullptr, 0ll, wp, nullptr, y, ybs, 0, Cout, H, W, H, W, 0, tiles_x, tiles_y, slope,
                                                                        x8, x8bs, n8oct, W, W, ai);
}

// the 1x1 projection NCHW -> octets, stored into two buffers (DualOut)
template <typename T, typename TO>
__global__ __launch_bounds__(NTHREADS, 2)
void conv1x1_dual_kernel(const T* __restrict__ x, long long xbs, const T* __restrict__ wp, const float* __restrict__ bias, TO* __restrict__ y, long long ybs,
                         int Cin, int Cout, int H, int W, int tiles_x, int tiles_y, float slope, int xpitch, const DualOut<TO> d) {
  conv_body<T, 1, 2, 1, 4, 0, false, false, 0, true, false, TO, DualOut<TO>>(x, xbs, wp, bias, y, ybs, Cin, Cout, H, W, H, W, 0, tiles_x, tiles_y, slope,
                                                                            nullptr, 0ll, 0, xpitch, W, d);
}

// launch heuristics and experiment switches (upf_conv_set_option)
inline int g_pair_th = 0;              // conv_pair.hip: tile height override (experiments): 0 = the default per class, 4 | 8
inline int g_sk_grid = 48, g_sk_grid_narrow = 96, g_sk_grid_d4 = 16, g_small_grid = 256, g_rpw4_min = 256, g_ph_fit = 1, g_force_mtw = 0, g_force_sk = -1, g_ablate = 0;

struct Args {
  const void* x; long long xbs; const void* wp; const float* bias; void* y; long long ybs;
  int B, Cin, Cout, H, W, d, stride, ntaps; float slope; hipStream_t stream;
  int xpitch, ypitch;                                // elements between rows of x / y (>= W / Wo)
  // gated epilogue (gate_add or gate_y != nullptr; 3x3, stride 1, dilation 1 only): see ActGate
  const void* gate_add = nullptr; long long gate_abs = 0; const void* gate_y = nullptr; long long gate_ybs = 0; float gate_slope = 0.f;
};

template <typename T, int MTW, int RPW, int S, int NOCTS, int D, bool GEN, bool ONE = false>
int launch_one(const Args& a, int slabs) {
  constexpr int TH = (4 / MTW) * RPW;
  constexpr bool PH = (D >= 2 && S == 1);            // row-phase decomposition (see the kernel)
  const int Ho = (a.H - 1) / S + 1, Wo = (a.W - 1) / S + 1;
  const int tiles_x = cdiv(Wo, TW), tiles_y = PH ? cdiv(Ho, D * TH) * D : cdiv(Ho, TH);
  const int rows = PH ? TH + 2 : S * (TH - 1) + 2 * a.d + 1;
  size_t lds = (size_t)NOCTS * rows * (xw(S, margin_of(D)) + xw(S, margin_of(D)) / 16) * 16;
  if (lds < 4 * EPI_WAVE_BYTES) lds = 4 * EPI_WAVE_BYTES;        // the wide epilogue's patches reuse the region
  UPF_REQUIRE(lds <= 160 * 1024, UPF_EUNSUPPORTED, "conv_forward: tile does not fit LDS (dilation %d, stride %d)", a.d, S);
  if (a.gate_y || a.gate_add) {
    if constexpr (S == 1 && D == 1) {
      static LdsOptIn gopt;
      auto gkern = &conv_gated_kernel<T, MTW, RPW, S, NOCTS, D, GEN, ONE>;
      gopt.ensure(reinterpret_cast<const void*>(gkern), lds);
      const ActGate<T> gate{(const T*)a.gate_add, a.gate_abs, (const T*)a.gate_y, a.gate_ybs, a.gate_slope};
      hipLaunchKernelGGL(gkern, dim3((unsigned)(a.B * tiles_x * tiles_y), slabs), dim3(NTHREADS), lds, a.stream, (const T*)a.x, a.xbs,
                         (const T*)a.wp, a.bias, (T*)a.y, a.ybs, a.Cin, a.Cout, a.H, a.W, Ho, Wo, g_ablate, tiles_x, tiles_y, a.slope,
                         a.xpitch, a.ypitch, gate);
      return check_launch("conv_forward_gated");
    } else {
      set_error("conv_forward_gated: 3x3, stride 1, dilation 1 only");
      return UPF_EUNSUPPORTED;
    }
  }
  static LdsOptIn opt;
  auto kern = &conv_kernel<T, MTW, RPW, S, NOCTS, D, GEN, ONE>;
  opt.ensure(reinterpret_cast<const void*>(kern), lds);
  hipLaunchKernelGGL(kern, dim3((unsigned)(a.B * tiles_x * tiles_y), slabs), dim3(NTHREADS), lds, a.stream, (const T*)a.x, a.xbs,
                     (const T*)a.wp, a.bias, (T*)a.y, a.ybs, a.Cin, a.Cout, a.H, a.W, Ho, Wo, D >= 0 ? g_ablate : a.d, tiles_x, tiles_y, a.slope,
                     (const T*)nullptr, 0ll, 0, a.xpitch, a.ypitch);
  return check_launch("conv_forward");
}

// run-time (kernel size, dilation, Cin) -> compile-time (D, NOCTS).  16-channel chunks (NOCTS = 2) where 72 weight
// registers + the accumulators would spill (four channel blocks per workgroup: 128 accumulator registers per wave),
// where the halo of the dilation would not fit LDS, and for the RGB / 16-channel layers (half the staging).
template <int MTW, int D> constexpr int wide_nocts() { return (MTW == 4 || D < 0 || (MTW == 2 && D == 16)) ? 2 : 4; }

// Row-phase layers (compile-time dilation >= 2): a tile holds TH rows of ONE phase, and a phase has only ceil(Ho / d)
// rows — 6 at the 1/4-resolution level of config 2 for dilation 16, 3 one level down — so 8-row tiles compute up to
// 2.7x the rows that exist (round 2: the dilation-8 / -16 layers ran at 50-60 % of their neighbours' rate).  Pick the
// tile height among those the wave split offers, by staged rows (TH + 2 per tile, + the fixed cost of a tile).
inline int ph_tile_rows(int Ho, int d, int mtw) {
  const int rpp = cdiv(Ho, d);
  int best = 8; float best_cost = 1e30f;
  for (int th : {8, 6, 4}) {
    if (mtw == 1 && th == 6) continue;               // (four row groups: TH = 4 * RPW)
    const float cost = (float)cdiv(rpp, th) * ((float)th + 2.5f);
    if (cost < best_cost - 1e-3f) { best_cost = cost; best = th; }
  }
  return best;
}

}  // namespace conv
}  // namespace upf
