// 3x3 (dilated / strided) and 1x1 convolution as an implicit GEMM on the bf16/fp16 matrix cores — gfx950.
//
// the dense flow-estimator / context / SGU-estimator / feature-pyramid convolutions
// (/root/reference/model/pwc_modules.py:250-286, :396-412, :172-199, model/upflow.py:24-60) are where an
// inference step actually spends its time (≈1.5 TFLOP per 384x1280 batch-4 step; MIOpen reaches ≈55 TFLOP/s on
// them through im2col + GEMM + NCHW<->NHWC transposes + separate bias / LeakyReLU / concat kernels).
// BASELINE.json's north star allows MFMA exactly here ("a real contraction").
//
//   y[n, co, i, j] = act( bias[co] + sum_{ci,ky,kx} w[co,ci,ky,kx] * x[n, ci, S*i+(ky-1)d, S*j+(kx-1)d] )
//   padding = dilation d (zero padding), stride S in {1,2}, act = LeakyReLU(slope) or identity.
//
// x and y are CHANNEL SLICES of larger contiguous NCHW buffers (batch strides given): the dense estimator's
// growing concatenation (`x1 = cat([conv1(x), x])`, pwc_modules.py:280-285) becomes one 565-channel buffer that
// every conv reads a suffix of and writes its own slice of — no concat copy, no separate bias or activation
// pass, no layout transposes, no im2col buffer.
//
// GEMM view per image: D[co][pixel] = sum_tap sum_ci W[tap][co][ci] * X[ci][pixel + shift(tap)],
// v_mfma_f32_32x32x16_{bf16,f16}: A = 32 output channels x 16 k, B = 16 k x 32 pixels (one tile row).
//   * workgroup = 4 waves = a TH x 32 pixel tile of one image and MTW*32 output channels.  The waves split the
//     OUTPUT CHANNELS first (wave -> 32-channel block cb = wave % MTW) and the tile rows second
//     (row group rg = wave / MTW, RPW rows each): MTW=4 -> every wave owns all 8 rows of its 32 channels.
//   * the x tile + halo of a chunk of 32 (16) input channels is staged ONCE into LDS, transposed in registers
//     (8 channel rows x 8 pixels -> 8 pixels x 8 channels, 32 v_perm) into 16-byte entries
//     [channel-octet][row][col]; taps read SHIFTED windows of it, so the im2col expansion exists only as LDS read
//     addresses.  Halo zeros and channels >= Cin come from the buffer descriptor's bounds check.  The first
//     staging task of the NEXT chunk is loaded into registers before the matrix phase (HBM latency hides under it).
//   * the WEIGHTS never touch LDS: they are pre-packed in MFMA lane order (pack_weights_kernel), each lane keeps
//     its A operand of all 9 taps x 2 k-steps of the current chunk in 72 registers, loaded straight from the
//     packed buffer (L2-resident, identical for every workgroup), and re-loads a kernel row for the next chunk
//     right after that row's last use.  (The first version staged weights per tap through LDS: one barrier and one
//     exposed L2 round trip per tap, and 1 A + 0.5 B LDS reads per MFMA — LDS-pipe bound at ~30 % of the MFMA peak.)
//   * compile-time dilation D (1,2,4,8): staged row s feeds output rows r = s - ky*D, so each B window is read
//     once per (s, kx) and used by up to three MFMAs: (RPW+2D)*3 LDS reads instead of 9*RPW per k-step.
//   * epilogue: + bias, LeakyReLU, convert, lane pairs exchange so that every lane stores two adjacent pixels.
#include "conv_kernel.hpp"

namespace upf {
namespace conv {

// w [Cout, Cin, k, k] (k*k = ntaps) -> packed [slab = co/32][k-step = ci/16][tap][kg = (ci/8)%2][px = co%32][ci%8],
// zero padded to pad32(Cout) x pad32(Cin): the 1 KB block of one (slab, k-step, tap) is exactly the A operand of one
// v_mfma_f32_32x32x16 in lane order (lane = kg*32 + px holds 8 consecutive input channels of output channel px),
// so a wave fetches it with ONE fully coalesced 16-byte-per-lane load.
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

// The same packing straight from the fp32 MASTER weights of a training step (one launch instead of cast + pack), for the
// forward convolution (dgrad = 0) or for its DATA GRADIENT (dgrad = 1): gx = conv(g, w') with w'[ci][co][tap] =
// w[co][ci][ntaps-1-tap] — the spatially flipped, channel-transposed kernel — so the packed operand has Cout' = Cin rows
// and Cin' = Cout k-columns.
// dgrad = 2: the data gradient of a STRIDE-2 3x3 layer in its space-to-depth form (ops.py: _s2d_ok).  The layer equals a
// stride-1 convolution of xs[(ci,p,q), i, j] = x[ci, 2i+p, 2j+q] with the kernel w4[co][(ci,p,q)][a][b] = w[co][ci][ky][kx]
// at (p, a) = s2d(ky), (q, b) = s2d(kx) (ky = 0: phase 1 of the row above; 1: phase 0; 2: phase 1 of the same row) and zero
// elsewhere; packed here is w4's data-gradient operand (Cin' = Cout k-columns, Cout' = 4*Cin rows) straight from w.
__device__ __forceinline__ int s2d_tap(int phase, int a) { return a == 0 ? (phase == 1 ? 0 : -1) : (a == 1 ? (phase == 0 ? 1 : 2) : -1); }
template <typename T>
__device__ __forceinline__ void pack_weights_f32_body(const float* __restrict__ w, T* __restrict__ wp, int Cin, int Cout, int ntaps, int dgrad,
                                                      unsigned bid, unsigned nblocks) {
  const int cinp = dgrad ? Cout : Cin, coutp = dgrad == 2 ? 4 * Cin : (dgrad ? Cin : Cout);   // channel counts of the convolution being packed
  const int cip = pad32(cinp), cop = pad32(coutp), nk = cip / 16;
  const long long total = (long long)ntaps * cop * cip;
  for (long long i = bid * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)nblocks * blockDim.x) {
    const int j = (int)(i & 7), px = (int)((i >> 3) & 31), kg = (int)((i >> 8) & 1);
    const long long b = i >> 9;
    const int tap = (int)(b % ntaps), kstep = (int)((b / ntaps) % nk), slab = (int)(b / ((long long)ntaps * nk));
    const int co = slab * 32 + px, ci = kstep * 16 + kg * 8 + j;
    float v = 0.f;
    if (ci < cinp && co < coutp) {
      if (dgrad == 2) {
        const int ft = ntaps - 1 - tap, ky = s2d_tap((co >> 1) & 1, ft / 3), kx = s2d_tap(co & 1, ft % 3);
        if (ky >= 0 && kx >= 0) v = w[((size_t)ci * Cin + (co >> 2)) * 9 + ky * 3 + kx];
      } else {
        v = dgrad ? w[((size_t)ci * Cin + co) * ntaps + (ntaps - 1 - tap)] : w[((size_t)co * Cin + ci) * ntaps + tap];
      }
    }
    Elem<T>::store(wp + i, v);
  }
}
template <typename T>
__global__ void pack_weights_f32_kernel(const float* __restrict__ w, T* __restrict__ wp, int Cin, int Cout, int ntaps, int dgrad) {
  pack_weights_f32_body<T>(w, wp, Cin, Cout, ntaps, dgrad, blockIdx.x, gridDim.x);
}
// Every layer of a training step in ONE launch (upf_conv_pack_weights_f32_multi): the job table travels in the kernel
// arguments, a workgroup finds its job by its block index (a step packs ~60 layers: 60 launches of 3-5 us otherwise).
constexpr int PACK_JOBS = 56;
struct PackJobs {
  const float* w[PACK_JOBS];
  void* wp[PACK_JOBS];
  int Cin[PACK_JOBS], Cout[PACK_JOBS];
  unsigned char ntaps[PACK_JOBS], dgrad[PACK_JOBS];
  unsigned blk0[PACK_JOBS + 1];
  int n;
};
template <typename T>
__global__ void pack_weights_f32_multi_kernel(PackJobs J) {
  int j = 0;
  while (j + 1 < J.n && blockIdx.x >= J.blk0[j + 1]) ++j;           // (workgroup-uniform)
  pack_weights_f32_body<T>(J.w[j], (T*)J.wp[j], J.Cin[j], J.Cout[j], J.ntaps[j], J.dgrad[j], blockIdx.x - J.blk0[j], J.blk0[j + 1] - J.blk0[j]);
}

// The data-gradient operands of a DENSE STACK in one launch (upf_conv_pack_stacked_dgrad; ops.DenseStackTrainFunction).  The
// gradient of buffer slice k is one convolution of the pre-activation gradients of the layers after k with the rows of their
// kernels that read slice k, stacked along K: operand k = pack(dgrad of cat_j(w_j[:, col_k - hi_j : + f_k])) for the first
// npos_k layers of one ordered list (last layer first).  Was a torch.cat and a pack launch per slice (24 launches per step for
// the two stacks); here the concatenation is an index computation.
constexpr int STACK_LAYERS = 8, STACK_TARGETS = 8;
struct StackPack {
  const float* w[STACK_LAYERS];                      // [Cout_j, Cin_j, 3, 3] fp32 masters, in gradient-buffer order
  int Cin[STACK_LAYERS], Cout[STACK_LAYERS], hi[STACK_LAYERS];   // hi: first buffer channel layer j reads
  void* wp[STACK_TARGETS];
  int col[STACK_TARGETS], f[STACK_TARGETS], npos[STACK_TARGETS];  // buffer channels [col, col + f), consumers = layers [0, npos)
  unsigned blk0[STACK_TARGETS + 1];
  int nt;
};
template <typename T>
__global__ void pack_stacked_dgrad_kernel(StackPack J) {
  int t = 0;
  while (t + 1 < J.nt && blockIdx.x >= J.blk0[t + 1]) ++t;          // (workgroup-uniform)
  int K = 0;
  for (int j = 0; j < J.npos[t]; ++j) K += J.Cout[j];
  const int f = J.f[t], col = J.col[t];
  const int cip = pad32(K), cop = pad32(f), nk = cip / 16;
  const long long total = 9ll * cop * cip;
  const unsigned nblocks = J.blk0[t + 1] - J.blk0[t];
  T* wp = (T*)J.wp[t];
  for (long long i = (blockIdx.x - J.blk0[t]) * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)nblocks * blockDim.x) {
    const int j8 = (int)(i & 7), px = (int)((i >> 3) & 31), kg = (int)((i >> 8) & 1);
    const long long b = i >> 9;
    const int tap = (int)(b % 9), kstep = (int)((b / 9) % nk), slab = (int)(b / (9ll * nk));
    const int co = slab * 32 + px;
    int ci = kstep * 16 + kg * 8 + j8;
    float v = 0.f;
    if (ci < K && co < f) {
      int j = 0;
      while (ci >= J.Cout[j]) { ci -= J.Cout[j]; ++j; }
      v = J.w[j][((size_t)ci * J.Cin[j] + (col - J.hi[j] + co)) * 9 + (8 - tap)];
    }
    Elem<T>::store(wp + i, v);
  }
}

// Split-K variant for the COARSE pyramid levels (a handful of pixel tiles on 256 CUs: the kernel above is then a
// serial chain of Cin/32 chunks, ~1.5 us each, on a few dozen workgroups).  Here a workgroup owns a 2 x 32 pixel
// tile and 32 output channels, and its four waves split the INPUT-CHANNEL chunks (wave w takes chunks w, w+4, ...):
// every wave stages its own chunk into a private LDS region — no workgroup barrier in the loop — and the four
// partial accumulators are summed through LDS at the end in a fixed order (deterministic).  4x the workgroups,
// a quarter of the chain each.
template <typename T, int NOCTS, int D, bool GEN, typename TO = T, typename G = NoGate>
__global__ __launch_bounds__(NTHREADS, 2)
void conv_sk_kernel(const T* __restrict__ x, long long xbs, const T* __restrict__ wp, const float* __restrict__ bias,
                    TO* __restrict__ y, long long ybs, int Cin, int Cout, int H, int W, int tiles_x, int tiles_y, float slope,
                    int xpitch, int ypitch, const G gate) {
  static_assert(!G::on || sizeof(TO) == sizeof(T), "gated epilogue: operands of one type");
  constexpr int ntaps = (D == 0) ? 1 : 9;
  constexpr int marg = margin_of(D);
  constexpr int KS = NOCTS / 2, KCH = NOCTS * 8;
  constexpr int RPW = 2, TH = 2;
  constexpr int XW = xw(1, marg), XWP = XW + XW / 16;
  constexpr int rows = TH + 2 * D;                   // staged input rows
  constexpr int ngroups = XW / 8;
  constexpr int ntasks = NOCTS * rows * ngroups;     // per wave and chunk
  constexpr int NT = (ntasks + 63) / 64;             // staging tasks per lane
  constexpr int NPF = NT < 2 ? NT : 2;               // of which prefetched a chunk ahead
  constexpr int WAVE_ENTRIES = NOCTS * rows * XWP;
  extern __shared__ __attribute__((aligned(16))) uint4 smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  uint4* xs = smem + wave * WAVE_ENTRIES;            // this wave's private x tile

  const int bid = xcd_remap(blockIdx.x, gridDim.x);
  const int tx = bid % tiles_x, ty = (bid / tiles_x) % tiles_y, n = bid / (tiles_x * tiles_y);
  const int x0 = tx * TW, y0 = ty * TH;
  const int px = lane & 31, kg = lane >> 5;
  const int slab = blockIdx.y;
  const int cip = pad32(Cin);
  const int nchunks = cip / KCH, nksteps = cip / 16;
  const int HW = H * xpitch;                          // elements per channel plane of x (rows `xpitch` apart, conv_kernel.hpp)
  const uint32_t plane = (uint32_t)HW * 2u;
  __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(x + (size_t)n * xbs), 0, (uint32_t)Cin * plane, 0x00020000);
  __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(wp), 0, (uint32_t)ntaps * (uint32_t)pad32(Cout) * (uint32_t)cip * 2u, 0x00020000);

  // wave 0's partial tile starts at the bias (channels >= Cout read 0 through the descriptor)
  __amdgpu_buffer_rsrc_t br = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(bias), 0, wave == 0 ? (uint32_t)Cout * 4u : 0u, 0x00020000);
  f32x16 acc[RPW];
#pragma unroll
  for (int e = 0; e < 16; ++e) {
    const float bv = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(br, (uint32_t)(slab * 32 + (e & 3) + 8 * (e >> 2) + 4 * kg) * 4u, 0, 0));
#pragma unroll
    for (int r = 0; r < RPW; ++r) acc[r][e] = bv;
  }

  auto task_geom = [&](int t, uint32_t& off, int& dst, int& sh) {
    const int oct = t / (rows * ngroups), rem = t - oct * (rows * ngroups), r = rem / ngroups, g = rem - r * ngroups;   // group fastest: coalesced loads
    const int gy = y0 - D + r, gx = x0 - marg + 8 * g;
    const bool in = (t < ntasks) && gy >= 0 && gy < H && gx >= 0 && gx < W;
    sh = (in && gx + 8 > W) ? gx + 8 - W : 0;          // (GEN: load shifted left; !GEN: trailing pixels masked — conv_kernel.hpp)
    off = in ? ((uint32_t)((oct * 8) * HW + gy * xpitch + gx - (GEN ? sh : 0)) * 2u) : 0x80000000u;
    dst = ((oct * rows + r) * XWP + 8 * g) * 8 + ((g >> 1) & 7);   // entry index * 8 + slot rotation of the group
  };
  auto task_load = [&](uint32_t off, int cc, u32x4 (&ch)[8]) {
    const uint32_t o = (cc < nchunks) ? off + (uint32_t)cc * KCH * plane : 0x80000000u;
#pragma unroll
    for (int k = 0; k < 8; ++k) ch[k] = __builtin_amdgcn_raw_buffer_load_b128(xr, o + k * plane, 0, 0);
  };
  auto task_store = [&](int enc, int sh, const u32x4 (&ch)[8]) { stage_store<GEN>(xs, enc, sh, ch); };

  // geometry of this lane's staging tasks (the same for every chunk) and the prefetch registers of the first NPF
  uint32_t toff[NT]; int tdst[NT], tsh[NT];
#pragma unroll
  for (int i = 0; i < NT; ++i) task_geom(lane + 64 * i, toff[i], tdst[i], tsh[i]);
  u32x4 pre[NPF][8];
#pragma unroll
  for (int i = 0; i < NPF; ++i) task_load(toff[i], wave, pre[i]);

  uint4 wa[ntaps][KS];
  auto wload = [&](int cc, int tap, int ks) {
    const uint32_t off = (cc < nchunks) ? (uint32_t)(((slab * nksteps + cc * KS + ks) * ntaps + tap) * 1024 + lane * 16) : 0x80000000u;
    return __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(wr, off, 0, 0));
  };
#pragma unroll
  for (int tap = 0; tap < ntaps; ++tap)
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) wa[tap][ks] = wload(wave, tap, ks);

  constexpr bool REUSE = (D >= 1 && (RPW + 2 * D) * 3 < 9 * RPW);
  const int colx[3] = {swz(marg + px - D), swz(marg + px), swz(marg + px + D)};

  for (int cc = wave; cc < nchunks; cc += 4) {
    // ---- stage this wave's chunk (LDS operations of one wave execute in order: no barrier against its own reads)
#pragma unroll
    for (int i = 0; i < NT; ++i) {
      if (i < NPF) {
        if (lane + 64 * i < ntasks) task_store(tdst[i], tsh[i], pre[i < NPF ? i : 0]);
      } else if (lane + 64 * i < ntasks) {
        u32x4 ch[8];
        task_load(toff[i], cc, ch);
        task_store(tdst[i], tsh[i], ch);
      }
    }
#pragma unroll
    for (int i = 0; i < NPF; ++i) task_load(toff[i], cc + 4, pre[i]);

    if constexpr (REUSE) {
      constexpr int NR = RPW + 2 * D;
#pragma unroll
      for (int sr = 0; sr < NR; ++sr) {
#pragma unroll
        for (int kx = 0; kx < 3; ++kx)
#pragma unroll
          for (int ks = 0; ks < KS; ++ks) {
            const uint4 b = xs[((2 * ks + kg) * rows + sr) * XWP + colx[kx]];
#pragma unroll
            for (int ky = 0; ky < 3; ++ky)
              if (sr - ky * D >= 0 && sr - ky * D < RPW) acc[sr - ky * D] = Mma32<T>::mma(wa[ky * 3 + kx][ks], b, acc[sr - ky * D]);
          }
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
          if (sr == ky * D + RPW - 1) {
#pragma unroll
            for (int kx = 0; kx < 3; ++kx)
#pragma unroll
              for (int ks = 0; ks < KS; ++ks) wa[ky * 3 + kx][ks] = wload(cc + 4, ky * 3 + kx, ks);
          }
      }
    } else {
#pragma unroll
      for (int tap = 0; tap < ntaps; ++tap) {
        const int ky = (ntaps == 1) ? 1 : tap / 3, kx = (ntaps == 1) ? 1 : tap - 3 * (tap / 3);
        const int col = swz(marg + px + (kx - 1) * D);
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
#pragma unroll
          for (int r = 0; r < RPW; ++r) {
            const uint4 b = xs[((2 * ks + kg) * rows + r + ky * D) * XWP + col];
            acc[r] = Mma32<T>::mma(wa[tap][ks], b, acc[r]);
          }
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) wa[tap][ks] = wload(cc + 4, tap, ks);
      }
    }
  }

  // ---- sum the four partial tiles in wave order: part[(w*2 + r)*16 + e][lane]
  __syncthreads();
  float* part = reinterpret_cast<float*>(smem);
#pragma unroll
  for (int r = 0; r < RPW; ++r)
#pragma unroll
    for (int e = 0; e < 16; ++e) part[((wave * 2 + r) * 16 + e) * 64 + lane] = acc[r][e];
  __syncthreads();
  // wave w finishes output row r = w/2, accumulator registers [8*(w%2), 8*(w%2)+8)
  const int r = wave >> 1, jbase = 4 * (wave & 1);
  Epilogue ep;
  epilogue_init<TO, true>(ep, y + (size_t)n * ybs, Cout, H, W, slab, lane, x0, ypitch);
  GateRsrc gr;
  if constexpr (G::on) gr = gate_init<T>(gate, n, (uint32_t)Cout * ep.plane2);
  if (y0 + r < H) {
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) {
      const int e0 = 2 * (jbase + jj), e1 = e0 + 1;
      float v0 = 0.f, v1 = 0.f;
#pragma unroll
      for (int w = 0; w < 4; ++w) {
        v0 += part[((w * 2 + r) * 16 + e0) * 64 + lane];
        v1 += part[((w * 2 + r) * 16 + e1) * 64 + lane];
      }
      epilogue_store<TO, true, G::on>(ep, v0, v1, epilogue_choff(jbase + jj) * ep.plane2 + (uint32_t)((y0 + r) * ypitch) * 2u, slope, &gr);
    }
  }
}

template <typename T, int MTW, int RPW, int S, bool GEN>
int launch_shape(const Args& a, int slabs) {
  const bool narrow = a.Cin <= 16;
#define UPF_CONV_D(DD) return narrow ? launch_one<T, MTW, RPW, S, 2, DD, GEN, true>(a, slabs) : launch_one<T, MTW, RPW, S, wide_nocts<MTW, DD>(), DD, GEN>(a, slabs);
  if constexpr (S == 2) {
    UPF_CONV_D(1)
  } else {
    if (a.ntaps == 1) { UPF_CONV_D(0) }
    switch (a.d) {
      case 1: UPF_CONV_D(1)
      case 2: UPF_CONV_D(2)
      case 4: UPF_CONV_D(4)
      case 8: UPF_CONV_D(8)
      case 16: UPF_CONV_D(16)
    }
    if constexpr (MTW < 4) {                         // (launch() never sends a run-time dilation to four-block workgroups)
      UPF_CONV_D(-1)
    }
    set_error("conv_forward: internal routing error (dilation %d, MTW %d)", a.d, MTW);
    return UPF_EUNSUPPORTED;
  }
#undef UPF_CONV_D
}

template <typename T, int NOCTS, int D, bool GEN>
int launch_sk_one(const Args& a) {
  const int tiles_x = cdiv(a.W, TW), tiles_y = cdiv(a.H, 2);
  constexpr int rows = 2 + 2 * D;
  size_t lds = (size_t)4 * NOCTS * rows * (xw(1, margin_of(D)) + xw(1, margin_of(D)) / 16) * 16;
  if (lds < 4 * 2 * 16 * 64 * 4) lds = 4 * 2 * 16 * 64 * 4;       // the partial-sum exchange reuses the region
  if (a.gate_y || a.gate_add) {
    if constexpr (D == 1) {
      static LdsOptIn gopt;
      auto gkern = &conv_sk_kernel<T, NOCTS, D, GEN, T, ActGate<T>>;
      gopt.ensure(reinterpret_cast<const void*>(gkern), lds);
      const ActGate<T> gate{(const T*)a.gate_add, a.gate_abs, (const T*)a.gate_y, a.gate_ybs, a.gate_slope};
      hipLaunchKernelGGL(gkern, dim3((unsigned)(a.B * tiles_x * tiles_y), cdiv(a.Cout, 32)), dim3(NTHREADS), lds, a.stream, (const T*)a.x, a.xbs,
                         (const T*)a.wp, a.bias, (T*)a.y, a.ybs, a.Cin, a.Cout, a.H, a.W, tiles_x, tiles_y, a.slope, a.xpitch, a.ypitch, gate);
      return check_launch("conv_forward_gated");
    } else {
      set_error("conv_forward_gated: 3x3, stride 1, dilation 1 only");
      return UPF_EUNSUPPORTED;
    }
  }
  static LdsOptIn opt;
  auto kern = &conv_sk_kernel<T, NOCTS, D, GEN>;
  opt.ensure(reinterpret_cast<const void*>(kern), lds);
  hipLaunchKernelGGL(kern, dim3((unsigned)(a.B * tiles_x * tiles_y), cdiv(a.Cout, 32)), dim3(NTHREADS), lds, a.stream, (const T*)a.x, a.xbs,
                     (const T*)a.wp, a.bias, (T*)a.y, a.ybs, a.Cin, a.Cout, a.H, a.W, tiles_x, tiles_y, a.slope, a.xpitch, a.ypitch, NoGate{});
  return check_launch("conv_forward");
}

// split-K applies to stride 1, kernel 1x1 or dilation 1/2/4, at least 3 chunks of 32 input channels
template <typename T, bool GEN>
int launch_sk(const Args& a) {
  if (a.ntaps == 1) return launch_sk_one<T, 4, 0, GEN>(a);
  if (a.d == 1) return launch_sk_one<T, 4, 1, GEN>(a);
  if (a.d == 2) return launch_sk_one<T, 4, 2, GEN>(a);
  return launch_sk_one<T, 4, 4, GEN>(a);
}


// the extra tile heights of the row-phase layers: D in {2, 4, 8, 16} only
template <typename T, int MTW, int RPW, bool GEN>
int launch_shape_ph(const Args& a, int slabs) {
  switch (a.d) {
    case 2: return launch_one<T, MTW, RPW, 1, wide_nocts<MTW, 2>(), 2, GEN>(a, slabs);
    case 4: return launch_one<T, MTW, RPW, 1, wide_nocts<MTW, 4>(), 4, GEN>(a, slabs);
    case 8: return launch_one<T, MTW, RPW, 1, wide_nocts<MTW, 8>(), 8, GEN>(a, slabs);
    case 16: return launch_one<T, MTW, RPW, 1, wide_nocts<MTW, 16>(), 16, GEN>(a, slabs);
  }
  set_error("conv_forward: internal routing error (row-phase dilation %d)", a.d);
  return UPF_EUNSUPPORTED;
}

// Cout and the grid size -> (MTW, RPW, slabs over blockIdx.y)
template <typename T, bool GEN>
int launch(const Args& a) {
  const int mt = cdiv(a.Cout, 32);
  const int Ho = (a.H - 1) / a.stride + 1, Wo = (a.W - 1) / a.stride + 1;
  const long long tiles = (long long)a.B * cdiv(Wo, TW) * cdiv(Ho, 8);
  const int small_grid = g_small_grid, rpw4_min = g_rpw4_min;
  // Split-K (round 3, per-layer sweep of config 2: profiles/r03_conv_layers_sweep.txt): the coarsest grids always; grids of
  // 49..96 tiles only for Cout <= 64 (wider layers run faster as 64-channel workgroups of the tiled kernel); dilation 4
  // stages 10 rows per wave for 2 output rows, so only the very coarsest grids gain.
  const bool sk_ok = a.stride == 1 && a.Cin > 64 && (a.ntaps == 1 || a.d == 1 || a.d == 2 || a.d == 4);
  const bool sk_auto = a.d == 4 ? tiles <= g_sk_grid_d4 : (tiles <= g_sk_grid || (tiles <= g_sk_grid_narrow && a.Cout <= 64));
  if (sk_ok && (g_force_sk < 0 ? sk_auto : g_force_sk == 1)) return launch_sk<T, GEN>(a);
  int mtw = mt >= 3 ? 4 : mt;
  if (tiles <= small_grid && mt > 1) {
    // coarse pyramid levels: too few pixel tiles to fill 256 CUs -> more, narrower workgroups over blockIdx.y (each
    // re-stages the x tile, which is irrelevant when the grid is latency-bound): 64-channel workgroups from 16 tiles up,
    // 32-channel ones below; 96 output channels of a deep layer on a small grid would waste a third of a second 64-block
    mtw = (tiles >= 16) ? 2 : 1;
    if (mt == 3 && a.Cin > 192 && tiles <= 96) mtw = 1;
  }
  if (g_force_mtw == 1 || g_force_mtw == 2 || g_force_mtw == 4) mtw = g_force_mtw < (mt >= 3 ? 4 : mt) ? g_force_mtw : (mt >= 3 ? 4 : mt);
  // four-block workgroups keep 128 accumulator registers per wave: only the window-reuse loops fit beside them
  const bool reuse_d = a.ntaps == 1 || a.d == 1 || a.d == 2 || a.d == 4 || a.d == 8 || a.d == 16;
  if (mtw == 4 && (a.stride == 2 || !reuse_d)) mtw = 2;
  const int slabs = cdiv(mt, mtw);
  if (a.stride == 2) {
    if (mtw == 2) return launch_shape<T, 2, 4, 2, GEN>(a, slabs);
    return launch_shape<T, 1, 2, 2, GEN>(a, slabs);
  }
  if (g_ph_fit && a.ntaps == 9 && (a.d == 2 || a.d == 4 || a.d == 8 || a.d == 16)) {
    const int th = ph_tile_rows(Ho, a.d, mtw);
    if (mtw == 4 && th == 6) return launch_shape_ph<T, 4, 6, GEN>(a, slabs);
    if (mtw == 4 && th == 4) return launch_shape_ph<T, 4, 4, GEN>(a, slabs);
    if (mtw == 2 && th == 6) return launch_shape_ph<T, 2, 3, GEN>(a, slabs);
    if (mtw == 2 && th == 4) return launch_shape_ph<T, 2, 2, GEN>(a, slabs);
    if (mtw == 1 && th == 4) return launch_shape_ph<T, 1, 1, GEN>(a, slabs);
  }
  if (mtw == 4) return launch_shape<T, 4, 8, 1, GEN>(a, slabs);
  if (mtw == 2) return launch_shape<T, 2, 4, 1, GEN>(a, slabs);
  if constexpr (!GEN) {
    // narrow layers on large grids: 16x32 tiles (4 rows per wave) halve the halo rows and barriers per pixel
    if ((long long)a.B * cdiv(Wo, TW) * cdiv(Ho, 16) >= rpw4_min && a.d <= 2 && a.ntaps == 9) return launch_shape<T, 1, 4, 1, false>(a, slabs);
  }
  return launch_shape<T, 1, 2, 1, GEN>(a, slabs);
}

}  // namespace conv
}  // namespace upf

extern "C" int upf_conv_set_option(const char* name, int value) {
  using namespace upf::conv;
  int* slot = nullptr;
  if (name && !strcmp(name, "sk_grid")) slot = &g_sk_grid;
  else if (name && !strcmp(name, "sk_grid_narrow")) slot = &g_sk_grid_narrow;
  else if (name && !strcmp(name, "sk_grid_d4")) slot = &g_sk_grid_d4;
  else if (name && !strcmp(name, "ablate")) slot = &g_ablate;            // experiments: see conv_kernel
  else if (name && !strcmp(name, "small_grid")) slot = &g_small_grid;
  else if (name && !strcmp(name, "rpw4_min")) slot = &g_rpw4_min;
  else if (name && !strcmp(name, "ph_fit")) slot = &g_ph_fit;          // 0: row-phase layers always on 8-row tiles (round-2 behaviour)
  else if (name && !strcmp(name, "force_mtw")) slot = &g_force_mtw;    // experiments: 0 = heuristic, 1 / 2 / 4
  else if (name && !strcmp(name, "force_sk")) slot = &g_force_sk;      // experiments: -1 = heuristic, 0 = never, 1 = wherever eligible
  else if (name && !strcmp(name, "pair_th")) slot = &g_pair_th;        // experiments (conv_pair.hip): 0 = default tile height, 4 | 8
  if (!slot) return INT32_MIN;
  const int prev = *slot;
  *slot = value;
  return prev;
}

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

extern "C" int upf_conv_pack_weights_f32(const float* w, void* w_packed, int Cin, int Cout, int kernel_size, int dtype, int dgrad, void* stream) {
  using namespace upf;
  UPF_REQUIRE(w && w_packed && Cin > 0 && Cout > 0, UPF_EINVAL, "conv_pack_weights_f32: bad arguments");
  UPF_REQUIRE(kernel_size == 3 || kernel_size == 1, UPF_EUNSUPPORTED, "conv_pack_weights_f32: kernel_size %d (1 or 3)", kernel_size);
  UPF_REQUIRE(dtype == UPF_BF16 || dtype == UPF_F16, UPF_EDTYPE, "conv_pack_weights_f32: packs to bf16 / fp16");
  UPF_REQUIRE(dgrad >= 0 && dgrad <= 2 && (dgrad != 2 || kernel_size == 3), UPF_EINVAL, "conv_pack_weights_f32: dgrad 0 / 1 / 2 (2: 3x3 only)");
  const int ntaps = kernel_size * kernel_size;
  const long long total = (long long)ntaps * conv::pad32(dgrad == 2 ? 4 * Cin : Cout) * conv::pad32(dgrad == 2 ? Cout : Cin);
  const int blocks = (int)((total + 255) / 256 > 4096 ? 4096 : (total + 255) / 256);
  if (dtype == UPF_BF16)
    hipLaunchKernelGGL((conv::pack_weights_f32_kernel<bf16_t>), dim3(blocks), dim3(256), 0, (hipStream_t)stream, w, (bf16_t*)w_packed, Cin, Cout, ntaps, dgrad);
  else
    hipLaunchKernelGGL((conv::pack_weights_f32_kernel<f16_t>), dim3(blocks), dim3(256), 0, (hipStream_t)stream, w, (f16_t*)w_packed, Cin, Cout, ntaps, dgrad);
  return check_launch("conv_pack_weights_f32");
}

extern "C" int upf_conv_pack_weights_f32_multi(const float* const* w, void* const* w_packed, const int* Cin, const int* Cout, const int* kernel_size,
                                               const int* dgrad, int njobs, int dtype, void* stream) {
  using namespace upf;
  UPF_REQUIRE(w && w_packed && Cin && Cout && kernel_size && dgrad && njobs > 0, UPF_EINVAL, "conv_pack_weights_f32_multi: bad arguments");
  UPF_REQUIRE(dtype == UPF_BF16 || dtype == UPF_F16, UPF_EDTYPE, "conv_pack_weights_f32_multi: packs to bf16 / fp16");
  for (int j = 0; j < njobs; ++j) {
    UPF_REQUIRE(w[j] && w_packed[j] && Cin[j] > 0 && Cout[j] > 0, UPF_EINVAL, "conv_pack_weights_f32_multi: job %d: bad arguments", j);
    UPF_REQUIRE(kernel_size[j] == 3 || kernel_size[j] == 1, UPF_EUNSUPPORTED, "conv_pack_weights_f32_multi: job %d: kernel_size %d (1 or 3)", j, kernel_size[j]);
    UPF_REQUIRE(dgrad[j] >= 0 && dgrad[j] <= 2 && (dgrad[j] != 2 || kernel_size[j] == 3), UPF_EINVAL, "conv_pack_weights_f32_multi: job %d: dgrad 0 / 1 / 2 (2: 3x3 only)", j);
  }
  for (int j0 = 0; j0 < njobs; j0 += conv::PACK_JOBS) {
    conv::PackJobs J;
    J.n = njobs - j0 < conv::PACK_JOBS ? njobs - j0 : conv::PACK_JOBS;
    unsigned nb = 0;
    for (int j = 0; j < J.n; ++j) {
      const int q = j0 + j, ntaps = kernel_size[q] * kernel_size[q];
      J.w[j] = w[q]; J.wp[j] = w_packed[q]; J.Cin[j] = Cin[q]; J.Cout[j] = Cout[q]; J.ntaps[j] = (unsigned char)ntaps; J.dgrad[j] = (unsigned char)dgrad[q];
      const long long total = (long long)ntaps * conv::pad32(dgrad[q] == 2 ? 4 * Cin[q] : (dgrad[q] ? Cin[q] : Cout[q])) * conv::pad32(dgrad[q] ? Cout[q] : Cin[q]);
      J.blk0[j] = nb;
      nb += (unsigned)((total + 255) / 256 > 256 ? 256 : (total + 255) / 256);
    }
    J.blk0[J.n] = nb;
    if (dtype == UPF_BF16) hipLaunchKernelGGL((conv::pack_weights_f32_multi_kernel<bf16_t>), dim3(nb), dim3(256), 0, (hipStream_t)stream, J);
    else hipLaunchKernelGGL((conv::pack_weights_f32_multi_kernel<f16_t>), dim3(nb), dim3(256), 0, (hipStream_t)stream, J);
  }
  return check_launch("conv_pack_weights_f32_multi");
}

extern "C" int upf_conv_pack_stacked_dgrad(const float* const* w, const int* Cin, const int* Cout, const int* first_channel, int nlayers,
                                           void* const* w_packed, const int* slice_channel, const int* slice_width, const int* nconsumers,
                                           int nslices, int dtype, void* stream) {
  using namespace upf;
  UPF_REQUIRE(w && Cin && Cout && first_channel && w_packed && slice_channel && slice_width && nconsumers, UPF_EINVAL, "conv_pack_stacked_dgrad: null pointer");
  UPF_REQUIRE(nlayers > 0 && nlayers <= conv::STACK_LAYERS && nslices > 0 && nslices <= conv::STACK_TARGETS, UPF_EINVAL,
              "conv_pack_stacked_dgrad: %d layers (<= %d), %d slices (<= %d)", nlayers, conv::STACK_LAYERS, nslices, conv::STACK_TARGETS);
  UPF_REQUIRE(dtype == UPF_BF16 || dtype == UPF_F16, UPF_EDTYPE, "conv_pack_stacked_dgrad: packs to bf16 / fp16");
  conv::StackPack J;
  for (int j = 0; j < nlayers; ++j) {
    UPF_REQUIRE(w[j] && Cin[j] > 0 && Cout[j] > 0 && first_channel[j] >= 0, UPF_EINVAL, "conv_pack_stacked_dgrad: layer %d: bad arguments", j);
    J.w[j] = w[j]; J.Cin[j] = Cin[j]; J.Cout[j] = Cout[j]; J.hi[j] = first_channel[j];
  }
  unsigned nb = 0;
  for (int t = 0; t < nslices; ++t) {
    UPF_REQUIRE(w_packed[t] && slice_width[t] > 0 && nconsumers[t] > 0 && nconsumers[t] <= nlayers, UPF_EINVAL, "conv_pack_stacked_dgrad: slice %d: bad arguments", t);
    int K = 0;
    for (int j = 0; j < nconsumers[t]; ++j) {
      // every consumer reads the whole slice: [slice_channel, + width) inside its input range [first_channel, first_channel + Cin)
      UPF_REQUIRE(slice_channel[t] >= first_channel[j] && slice_channel[t] + slice_width[t] <= first_channel[j] + Cin[j], UPF_EINVAL,
                  "conv_pack_stacked_dgrad: slice %d is not inside the input of layer %d", t, j);
      K += Cout[j];
    }
    J.wp[t] = w_packed[t]; J.col[t] = slice_channel[t]; J.f[t] = slice_width[t]; J.npos[t] = nconsumers[t];
    const long long total = 9ll * conv::pad32(slice_width[t]) * conv::pad32(K);
    J.blk0[t] = nb;
    nb += (unsigned)((total + 255) / 256 > 256 ? 256 : (total + 255) / 256);
  }
  J.blk0[nslices] = nb; J.nt = nslices;
  if (dtype == UPF_BF16) hipLaunchKernelGGL((conv::pack_stacked_dgrad_kernel<bf16_t>), dim3(nb), dim3(256), 0, (hipStream_t)stream, J);
  else hipLaunchKernelGGL((conv::pack_stacked_dgrad_kernel<f16_t>), dim3(nb), dim3(256), 0, (hipStream_t)stream, J);
  return check_launch("conv_pack_stacked_dgrad");
}

// ---- 1x1 convolution whose OUTPUT type differs from its operands' (conv_kernel.hpp: TO) -----------------------------------------
namespace upf {
namespace conv {
template <typename T, typename TO, bool GEN>
int launch_1x1_mixed(const Args& a) {                  // the ntaps == 1, Cout <= 32 subset of launch()
  const long long tiles = (long long)a.B * cdiv(a.W, TW) * cdiv(a.H, 8);
  const bool sk_auto = tiles <= g_sk_grid || tiles <= g_sk_grid_narrow;
  if (a.Cin > 64 && (g_force_sk < 0 ? sk_auto : g_force_sk == 1)) {
    const int tiles_x = cdiv(a.W, TW), tiles_y = cdiv(a.H, 2);
    size_t lds = (size_t)4 * 4 * 2 * (xw(1, margin_of(0)) + xw(1, margin_of(0)) / 16) * 16;
    if (lds < 4 * 2 * 16 * 64 * 4) lds = 4 * 2 * 16 * 64 * 4;
    static LdsOptIn opt;
    auto kern = &conv_sk_kernel<T, 4, 0, GEN, TO>;
    opt.ensure(reinterpret_cast<const void*>(kern), lds);
    hipLaunchKernelGGL(kern, dim3((unsigned)(a.B * tiles_x * tiles_y), 1), dim3(NTHREADS), lds, a.stream, (const T*)a.x, a.xbs,
                       (const T*)a.wp, a.bias, (TO*)a.y, a.ybs, a.Cin, a.Cout, a.H, a.W, tiles_x, tiles_y, a.slope, a.xpitch, a.ypitch, NoGate{});
    return check_launch("conv1x1_forward_mixed");
  }
  constexpr int TH = 4 * 2;
  const int tiles_x = cdiv(a.W, TW), tiles_y = cdiv(a.H, TH);
  size_t lds = (size_t)4 * TH * (xw(1, margin_of(0)) + xw(1, margin_of(0)) / 16) * 16;
  if (lds < 4 * EPI_WAVE_BYTES) lds = 4 * EPI_WAVE_BYTES;
  static LdsOptIn opt;
  auto kern = &conv_kernel<T, 1, 2, 1, 4, 0, GEN, false, 0, false, false, TO>;
  opt.ensure(reinterpret_cast<const void*>(kern), lds);
  hipLaunchKernelGGL(kern, dim3((unsigned)(a.B * tiles_x * tiles_y), 1), dim3(NTHREADS), lds, a.stream, (const T*)a.x, a.xbs,
                     (const T*)a.wp, a.bias, (TO*)a.y, a.ybs, a.Cin, a.Cout, a.H, a.W, a.H, a.W, g_ablate, tiles_x, tiles_y, a.slope,
                     (const T*)nullptr, 0ll, 0, a.xpitch, a.ypitch);
  return check_launch("conv1x1_forward_mixed");
}
template <typename T, typename TO>
int launch_1x1_mixed_c8(const Args& a) {               // NCHW (pitch-aligned rows) -> channel octets of type TO
  constexpr int TH = 4 * 2;
  const int tiles_x = cdiv(a.W, TW), tiles_y = cdiv(a.H, TH);
  const size_t lds = (size_t)4 * TH * (xw(1, margin_of(0)) + xw(1, margin_of(0)) / 16) * 16;
  static LdsOptIn opt;
  auto kern = &conv_kernel<T, 1, 2, 1, 4, 0, false, false, 0, true, false, TO>;
  opt.ensure(reinterpret_cast<const void*>(kern), lds);
  hipLaunchKernelGGL(kern, dim3((unsigned)(a.B * tiles_x * tiles_y), 1), dim3(NTHREADS), lds, a.stream, (const T*)a.x, a.xbs,
                     (const T*)a.wp, a.bias, (TO*)a.y, a.ybs, a.Cin, a.Cout, a.H, a.W, a.H, a.W, g_ablate, tiles_x, tiles_y, a.slope,
                     (const T*)nullptr, 0ll, 0, a.xpitch, a.ypitch);
  return check_launch("conv1x1_forward_mixed");
}
}  // namespace conv
}  // namespace upf

extern "C" int upf_conv1x1_forward_mixed(const void* x, long long x_batch_stride, int x_row_pitch, const void* w_packed, const float* bias,
                                         void* y, long long y_batch_stride, int y_row_pitch, int y_is_c8, int B, int Cin, int Cout, int H, int W,
                                         float leaky_slope, int dtype, int out_dtype, void* stream) {
  using namespace upf;
  UPF_REQUIRE(x && w_packed && bias && y, UPF_EINVAL, "conv1x1_forward_mixed: null pointer");
  UPF_REQUIRE(B > 0 && Cin > 0 && Cout > 0 && Cout <= 32 && H > 0 && W > 0, UPF_EINVAL, "conv1x1_forward_mixed: bad shape B=%d Cin=%d Cout=%d (<= 32) H=%d W=%d", B, Cin, Cout, H, W);
  UPF_REQUIRE((dtype == UPF_BF16 || dtype == UPF_F16) && (out_dtype == UPF_BF16 || out_dtype == UPF_F16) && dtype != out_dtype, UPF_EDTYPE,
              "conv1x1_forward_mixed: bf16 / fp16 operands and the OTHER 16-bit type for y (same types: upf_conv_forward)");
  if (x_row_pitch == 0) x_row_pitch = W;
  if (y_row_pitch == 0) y_row_pitch = W;
  UPF_REQUIRE(x_row_pitch >= W && (y_is_c8 || y_row_pitch >= W), UPF_EINVAL, "conv1x1_forward_mixed: row pitch smaller than the row");
  const bool gen = !(x_row_pitch % 8 == 0 && aligned_to(x, 16) && x_batch_stride % 8 == 0);
  UPF_REQUIRE(!gen || W >= 8, UPF_EUNSUPPORTED, "conv1x1_forward_mixed: W = %d < 8 with unaligned rows", W);
  UPF_REQUIRE(!y_is_c8 || (!gen && aligned_to(y, 16) && y_batch_stride % 8 == 0), UPF_EUNSUPPORTED, "conv1x1_forward_mixed: octet output needs 16-byte aligned input rows and output");
  UPF_REQUIRE((long long)Cin * H * x_row_pitch * 2 < (1ll << 31) && (long long)32 * H * (y_is_c8 ? W : y_row_pitch) * 2 < (1ll << 31), UPF_EINVAL, "conv1x1_forward_mixed: image too large for one buffer descriptor");
  UPF_REQUIRE(leaky_slope >= 0.f && leaky_slope <= 1.f, UPF_EINVAL, "conv1x1_forward_mixed: leaky_slope %g not in [0,1]", (double)leaky_slope);
  conv::Args a{x, x_batch_stride, w_packed, bias, y, y_batch_stride, B, Cin, Cout, H, W, 0, 1, 1, leaky_slope == 0.f ? 1.f : leaky_slope, (hipStream_t)stream,
               x_row_pitch, y_row_pitch};
  if (dtype == UPF_F16) {
    if (y_is_c8) return conv::launch_1x1_mixed_c8<f16_t, bf16_t>(a);
    return gen ? conv::launch_1x1_mixed<f16_t, bf16_t, true>(a) : conv::launch_1x1_mixed<f16_t, bf16_t, false>(a);
  }
  if (y_is_c8) return conv::launch_1x1_mixed_c8<bf16_t, f16_t>(a);
  return gen ? conv::launch_1x1_mixed<bf16_t, f16_t, true>(a) : conv::launch_1x1_mixed<bf16_t, f16_t, false>(a);
}

namespace upf {
namespace conv {
template <typename T, typename TO>
int launch_1x1_dual(const Args& a, void* y2, long long y2bs) {
  constexpr int TH = 4 * 2;
  const int tiles_x = cdiv(a.W, TW), tiles_y = cdiv(a.H, TH);
  const size_t lds = (size_t)4 * TH * (xw(1, margin_of(0)) + xw(1, margin_of(0)) / 16) * 16;
  static LdsOptIn opt;
  auto kern = &conv1x1_dual_kernel<T, TO>;
  opt.ensure(reinterpret_cast<const void*>(kern), lds);
  const DualOut<TO> d{(TO*)y2, y2bs};
  hipLaunchKernelGGL(kern, dim3((unsigned)(a.B * tiles_x * tiles_y), 1), dim3(NTHREADS), lds, a.stream, (const T*)a.x, a.xbs,
                     (const T*)a.wp, a.bias, (TO*)a.y, a.ybs, a.Cin, a.Cout, a.H, a.W, tiles_x, tiles_y, a.slope, a.xpitch, d);
  return check_launch("conv1x1_forward_c8_dual");
}
}  // namespace conv
}  // namespace upf

extern "C" int upf_conv1x1_forward_c8_dual(const void* x, long long x_batch_stride, int x_row_pitch, const void* w_packed, const float* bias,
                                           void* y8_a, long long ya_batch_stride, void* y8_b, long long yb_batch_stride,
                                           int B, int Cin, int Cout, int H, int W, float leaky_slope, int dtype, int out_dtype, void* stream) {
  using namespace upf;
  UPF_REQUIRE(x && w_packed && bias && y8_a && y8_b, UPF_EINVAL, "conv1x1_forward_c8_dual: null pointer");
  UPF_REQUIRE(B > 0 && Cin > 0 && Cout > 0 && Cout <= 32 && H > 0 && W > 0, UPF_EINVAL, "conv1x1_forward_c8_dual: bad shape B=%d Cin=%d Cout=%d (<= 32) H=%d W=%d", B, Cin, Cout, H, W);
  UPF_REQUIRE((dtype == UPF_BF16 || dtype == UPF_F16) && (out_dtype == UPF_BF16 || out_dtype == UPF_F16), UPF_EDTYPE, "conv1x1_forward_c8_dual: bf16 / fp16");
  if (x_row_pitch == 0) x_row_pitch = W;
  UPF_REQUIRE(x_row_pitch >= W && x_row_pitch % 8 == 0 && aligned_to(x, 16) && x_batch_stride % 8 == 0, UPF_EUNSUPPORTED,
              "conv1x1_forward_c8_dual: the input rows must be 16-byte aligned (W %% 8 == 0 or a row pitch that is a multiple of 8)");
  UPF_REQUIRE(aligned_to(y8_a, 16) && ya_batch_stride % 8 == 0 && aligned_to(y8_b, 16) && yb_batch_stride % 8 == 0, UPF_EINVAL, "conv1x1_forward_c8_dual: the octet outputs must be 16-byte aligned");
  UPF_REQUIRE((long long)Cin * H * x_row_pitch * 2 < (1ll << 31) && (long long)32 * H * W * 2 < (1ll << 31), UPF_EINVAL, "conv1x1_forward_c8_dual: image too large for one buffer descriptor");
  UPF_REQUIRE(leaky_slope >= 0.f && leaky_slope <= 1.f, UPF_EINVAL, "conv1x1_forward_c8_dual: leaky_slope %g not in [0,1]", (double)leaky_slope);
  conv::Args a{x, x_batch_stride, w_packed, bias, y8_a, ya_batch_stride, B, Cin, Cout, H, W, 0, 1, 1, leaky_slope == 0.f ? 1.f : leaky_slope, (hipStream_t)stream,
               x_row_pitch, W};
  if (dtype == UPF_F16) return out_dtype == UPF_F16 ? conv::launch_1x1_dual<f16_t, f16_t>(a, y8_b, yb_batch_stride) : conv::launch_1x1_dual<f16_t, bf16_t>(a, y8_b, yb_batch_stride);
  return out_dtype == UPF_BF16 ? conv::launch_1x1_dual<bf16_t, bf16_t>(a, y8_b, yb_batch_stride) : conv::launch_1x1_dual<bf16_t, f16_t>(a, y8_b, yb_batch_stride);
}

extern "C" int upf_conv_forward_pitched(const void* x, long long x_batch_stride, int x_row_pitch, const void* w_packed, const float* bias,
                                        void* y, long long y_batch_stride, int y_row_pitch, int B, int Cin, int Cout, int H, int W,
                                        int kernel_size, int dilation, int stride, float leaky_slope, int dtype, void* stream) {
  using namespace upf;
  UPF_REQUIRE(x && w_packed && bias && y, UPF_EINVAL, "conv_forward: null pointer");
  UPF_REQUIRE(B > 0 && Cin > 0 && Cout > 0 && H > 0 && W > 0, UPF_EINVAL,
              "conv_forward: bad shape B=%d Cin=%d Cout=%d H=%d W=%d", B, Cin, Cout, H, W);
  UPF_REQUIRE(dtype == UPF_BF16 || dtype == UPF_F16, UPF_EDTYPE, "conv_forward: bf16 / fp16 only");
  UPF_REQUIRE(kernel_size == 3 || kernel_size == 1, UPF_EUNSUPPORTED, "conv_forward: kernel_size %d (1 or 3)", kernel_size);
  UPF_REQUIRE(dilation >= 1 && dilation <= conv::MAXD, UPF_EUNSUPPORTED, "conv_forward: dilation %d not in [1,%d]", dilation, conv::MAXD);
  UPF_REQUIRE(stride == 1 || (stride == 2 && dilation == 1 && kernel_size == 3), UPF_EUNSUPPORTED, "conv_forward: stride %d (1, or 2 for a 3x3 with dilation 1)", stride);
  const int Ho = (H - 1) / stride + 1, Wo = (W - 1) / stride + 1;
  if (x_row_pitch == 0) x_row_pitch = W;
  if (y_row_pitch == 0) y_row_pitch = Wo;
  UPF_REQUIRE(x_row_pitch >= W && y_row_pitch >= Wo, UPF_EINVAL, "conv_forward: row pitch smaller than the row (x %d < %d or y %d < %d)", x_row_pitch, W, y_row_pitch, Wo);
  const bool gen = !(x_row_pitch % 8 == 0 && aligned_to(x, 16) && x_batch_stride % 8 == 0);   // rows not 16-byte aligned
  UPF_REQUIRE(!gen || W >= 8, UPF_EUNSUPPORTED, "conv_forward: W = %d < 8 with unaligned rows", W);
  UPF_REQUIRE((long long)Cin * H * x_row_pitch * 2 < (1ll << 31), UPF_EINVAL, "conv_forward: image too large for one buffer descriptor");
  UPF_REQUIRE(leaky_slope >= 0.f && leaky_slope <= 1.f, UPF_EINVAL, "conv_forward: leaky_slope %g not in [0,1]", (double)leaky_slope);
  UPF_REQUIRE((long long)Cout * Ho * y_row_pitch * 2 < (1ll << 31), UPF_EINVAL, "conv_forward: output too large for one buffer descriptor");
  // the kernels compute max(v, v*slope): "no activation" (0) becomes slope 1
  conv::Args a{x, x_batch_stride, w_packed, bias, y, y_batch_stride, B, Cin, Cout, H, W,
               kernel_size == 1 ? 0 : dilation, stride, kernel_size * kernel_size, leaky_slope == 0.f ? 1.f : leaky_slope, (hipStream_t)stream,
               x_row_pitch, y_row_pitch};
  if (dtype == UPF_BF16) return gen ? conv::launch<bf16_t, true>(a) : conv::launch<bf16_t, false>(a);
  return gen ? conv::launch<f16_t, true>(a) : conv::launch<f16_t, false>(a);
}

// The 3x3 stride-1 convolution with the gated epilogue (conv_kernel.hpp ActGate): y = ((conv(x) rounded to 16 bits) + add, rounded)
// x (act > 0 ? 1 : mask_slope).  add (optional) and act are [B,Cout,H,W] channel slices with y's row pitch.  Replaces, for the
// data-gradient convolutions of the dense stacks, the separate pass of upf_act_grad (whose bias partial sums are then taken by
// ONE pass over the whole stack's gradients).
extern "C" int upf_conv_forward_gated(const void* x, long long x_batch_stride, int x_row_pitch, const void* w_packed, const float* bias,
                                      void* y, long long y_batch_stride, int y_row_pitch, const void* add, long long add_batch_stride,
                                      const void* act, long long act_batch_stride, float mask_slope,
                                      int B, int Cin, int Cout, int H, int W, int dtype, void* stream) {
  using namespace upf;
  UPF_REQUIRE(x && w_packed && bias && y && (act || add), UPF_EINVAL, "conv_forward_gated: null pointer");
  if (!act) mask_slope = 1.f;                        // no mask: every element takes the "slope" 1
  UPF_REQUIRE(B > 0 && Cin > 0 && Cout > 0 && H > 0 && W > 0, UPF_EINVAL,
              "conv_forward_gated: bad shape B=%d Cin=%d Cout=%d H=%d W=%d", B, Cin, Cout, H, W);
  UPF_REQUIRE(dtype == UPF_BF16 || dtype == UPF_F16, UPF_EDTYPE, "conv_forward_gated: bf16 / fp16 only");
  if (x_row_pitch == 0) x_row_pitch = W;
  if (y_row_pitch == 0) y_row_pitch = W;
  UPF_REQUIRE(x_row_pitch >= W && y_row_pitch >= W, UPF_EINVAL, "conv_forward_gated: row pitch smaller than the row");
  const bool gen = !(x_row_pitch % 8 == 0 && aligned_to(x, 16) && x_batch_stride % 8 == 0);
  UPF_REQUIRE(!gen || W >= 8, UPF_EUNSUPPORTED, "conv_forward_gated: W = %d < 8 with unaligned rows", W);
  UPF_REQUIRE((long long)Cin * H * x_row_pitch * 2 < (1ll << 31) && (long long)Cout * H * y_row_pitch * 2 < (1ll << 31), UPF_EINVAL,
              "conv_forward_gated: image too large for one buffer descriptor");
  UPF_REQUIRE(aligned_to(add, 2) && aligned_to(act, 2) && mask_slope >= 0.f && mask_slope <= 1.f, UPF_EINVAL, "conv_forward_gated: bad gate operands");
  conv::Args a{x, x_batch_stride, w_packed, bias, y, y_batch_stride, B, Cin, Cout, H, W, 1, 1, 9, 1.f, (hipStream_t)stream, x_row_pitch, y_row_pitch};
  a.gate_add = add; a.gate_abs = add_batch_stride; a.gate_y = act; a.gate_ybs = act_batch_stride; a.gate_slope = mask_slope;
  if (dtype == UPF_BF16) return gen ? conv::launch<bf16_t, true>(a) : conv::launch<bf16_t, false>(a);
  return gen ? conv::launch<f16_t, true>(a) : conv::launch<f16_t, false>(a);
}

extern "C" int upf_conv_forward(const void* x, long long x_batch_stride, const void* w_packed, const float* bias,
                                void* y, long long y_batch_stride, int B, int Cin, int Cout, int H, int W,
                                int kernel_size, int dilation, int stride, float leaky_slope, int dtype, void* stream) {
  return upf_conv_forward_pitched(x, x_batch_stride, 0, w_packed, bias, y, y_batch_stride, 0, B, Cin, Cout, H, W, kernel_size, dilation, stride,
                                  leaky_slope, dtype, stream);
}
