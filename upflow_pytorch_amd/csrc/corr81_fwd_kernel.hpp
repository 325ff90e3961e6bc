// Device code of the 81-neighbour cost-volume forward kernel (see corr81_fwd.hip for the design notes).
// Kept in a header so that tools/corr_ablate.hip can instantiate ablated variants of the SAME code.
#pragma once
#include "common.hpp"

namespace upf {
namespace corr {

constexpr int R = 4, D = 9, ND = 81;
constexpr int TH = 8, TW = 32, PX = 4, XB = TW / PX;
constexpr int S1 = TW + 8;            // f1 LDS row stride (dwords); +8 keeps rows r, r+4 on disjoint slots
constexpr int S2 = TW + 2 * R;        // f2 tile width incl. halo = 40 dwords
constexpr int R2 = TH + 2 * R;        // 16 rows incl. halo
constexpr int SLOT1 = TH * S1;        // dwords per k-slot, f1
constexpr int SLOT2 = R2 * S2;        // dwords per k-slot, f2
constexpr int NWAVES = D;
constexpr int NTHREADS = NWAVES * 64;
// KC = k-slots per LDS chunk (template parameter of the kernel: 4 or 8; a k-slot is 1 fp32 channel or
// 2 bf16/fp16 channels).  Two staging buffers of KC*(SLOT1+SLOT2) dwords: 30,720 B (KC=4) / 61,440 B (KC=8).
constexpr int buf_dwords(int kc) { return kc * (SLOT1 + SLOT2); }
constexpr int lds_bytes(int kc) { return 2 * buf_dwords(kc) * 4; }
constexpr int tasks_per_thread(int kc) { return (kc * (TH * XB + R2 * (S2 / 4)) + NTHREADS - 1) / NTHREADS; }


template <typename T> struct Slot;
template <> struct Slot<float> {
  static constexpr int CH = 1;   // channels per k-slot
  static __device__ __forceinline__ float mac(uint32_t a, uint32_t b, float c) {
    return __builtin_fmaf(__uint_as_float(a), __uint_as_float(b), c);
  }
};
template <> struct Slot<bf16_t> {
  static constexpr int CH = 2;
  static __device__ __forceinline__ float mac(uint32_t a, uint32_t b, float c) {
    return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2_t, a), __builtin_bit_cast(bf16x2_t, b), c, false);
  }
};
template <> struct Slot<f16_t> {
  static constexpr int CH = 2;
  static __device__ __forceinline__ float mac(uint32_t a, uint32_t b, float c) {
    return __builtin_amdgcn_fdot2(__builtin_bit_cast(f16x2_t, a), __builtin_bit_cast(f16x2_t, b), c, false);
  }
};

// ---- staging ------------------------------------------------------------------------------------
// One staging task = the 4 dwords (4 consecutive pixels) of one k-slot at one tile position.  The
// geometry (tile row / quad, LDS address, inside the image or not) is the same for every channel
// chunk, so it is decoded ONCE per thread; per chunk only a wave-uniform byte offset is added.
//
// ALIGNED path (W % 4 == 0, 4-element aligned pointers, tensor < 2 GiB): raw BUFFER loads through a
// descriptor that spans exactly this batch item's [C,H,W] tensor.  The hardware bounds check then
// supplies every zero the algorithm needs for free: halo quads outside the image get the offset
// 0x80000000 (out of range), channel planes >= C (tail chunk, odd C) fall off the end of the
// descriptor.  Per task and chunk that leaves: 1-2 v_add, 1-2 buffer_load, (16-bit) 4 v_perm,
// 1 ds_write_b128 — the staging must stay this lean because the kernel is VALU-bound.
// The loads are unconditional and nothing consumes them before the MAC loop of the current chunk
// (hipcc waits vmcnt(0) at the end of any branch containing a load: cdna_hip_programming.md §5 (c)).
struct StageTask {
  int lds;          // LDS dword offset inside one buffer; -1 = no task
  uint32_t voff;    // ALIGNED: byte offset inside the batch item's tensor (0x80000000 = outside the image)
  int gx, goff;     // !ALIGNED: first column of the quad / gy*W (INT_MIN = row outside)
};
constexpr int NO_ROW = -2147483647 - 1;
constexpr int Q2_TASKS = R2 * (S2 / 4);           // f2 quads per k-slot

template <typename T, int KC>
__device__ __forceinline__ StageTask make_task(int t, int y0, int x0, int H, int W) {
  constexpr int N1_TASKS = KC * TH * XB;          // f1 tasks per chunk (a multiple of the wave size)
  StageTask s;
  s.lds = -1; s.voff = 0x80000000u; s.gx = 0; s.goff = NO_ROW;
  if (t >= N1_TASKS + KC * Q2_TASKS) return s;
  int gy, gx, k;
  if (t < N1_TASKS) {
    k = t / (TH * XB);
    const int rem = t - k * (TH * XB), r = rem / XB, q = rem - r * XB;
    gy = y0 + r; gx = x0 + 4 * q;
    s.lds = k * SLOT1 + r * S1 + 4 * q;
  } else {
    const int u = t - N1_TASKS;
    k = u / Q2_TASKS;
    const int rem = u - k * Q2_TASKS, r = rem / (S2 / 4), q = rem - r * (S2 / 4);
    gy = y0 - R + r; gx = x0 - R + 4 * q;
    s.lds = KC * SLOT1 + k * SLOT2 + r * S2 + 4 * q;
  }
  s.gx = gx;
  const bool row_ok = gy >= 0 && gy < H;
  s.goff = row_ok ? gy * W : NO_ROW;
  if (row_ok && gx >= 0 && gx < W)
    s.voff = (uint32_t)((k * Slot<T>::CH * H + gy) * W + gx) * (uint32_t)sizeof(typename Elem<T>::store_t);
  // the unaligned path also needs the k-slot: keep it in the low bits of goff's companion (gx is < 2^24)
  s.gx = (gx & 0x0fffffff) | (k << 28);
  return s;
}

// Raw data of one task for the chunk whose first k-slot is kb.
//   fp32  : 4 pixels of channel ks
//   16-bit: {row of channel c0 (x,y), row of channel c0+1 (z,w)}
template <typename T, bool ALIGNED>
__device__ __forceinline__ uint4 task_load(const StageTask& s, __amdgpu_buffer_rsrc_t rsrc, const T* __restrict__ fb,
                                           int C, int H, int W, int kb) {
  constexpr int CH = Slot<T>::CH;
  constexpr uint32_t ES = sizeof(typename Elem<T>::store_t);
  const uint32_t plane = (uint32_t)H * (uint32_t)W * ES;            // bytes per channel plane (uniform)
  if constexpr (ALIGNED) {
    const uint32_t off = s.voff + (uint32_t)kb * CH * plane;
    if constexpr (CH == 1) {
      return __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, off, 0, 0));
    } else {
      typedef __attribute__((ext_vector_type(2))) unsigned int u2;
      const u2 a = __builtin_amdgcn_raw_buffer_load_b64(rsrc, off, 0, 0);
      const u2 b = __builtin_amdgcn_raw_buffer_load_b64(rsrc, off + plane, 0, 0);
      return make_uint4(a.x, a.y, b.x, b.y);
    }
  } else {
    // ragged widths / unaligned pointers: element loads with clamped addresses, masked in task_finish
    const int k = (int)((uint32_t)s.gx >> 28), gx = (s.gx << 4) >> 4;
    const int ks = kb + k;
    const size_t HW = (size_t)H * W;
    const int rowbase = (s.lds >= 0 && s.goff != NO_ROW) ? s.goff : 0;
    uint32_t v[4];
    if constexpr (CH == 1) {
      const float* p = reinterpret_cast<const float*>(fb) + (size_t)min(ks, C - 1) * HW + rowbase;
#pragma unroll
      for (int i = 0; i < 4; ++i) v[i] = __float_as_uint(p[min(max(gx + i, 0), W - 1)]);
    } else {
      const uint16_t* p0 = reinterpret_cast<const uint16_t*>(fb) + (size_t)min(2 * ks, C - 1) * HW + rowbase;
      const uint16_t* p1 = reinterpret_cast<const uint16_t*>(fb) + (size_t)min(2 * ks + 1, C - 1) * HW + rowbase;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int x = min(max(gx + i, 0), W - 1);
        v[i] = (uint32_t)p0[x] | ((uint32_t)p1[x] << 16);
      }
    }
    return make_uint4(v[0], v[1], v[2], v[3]);
  }
}

// Raw loads -> LDS image.  ALIGNED: just the per-pixel interleave of channel c0 (low half) with
// c0+1 (high half) for 16-bit types.  !ALIGNED: also zero what is outside the image / beyond C.
template <typename T, bool ALIGNED>
__device__ __forceinline__ uint4 task_finish(const StageTask& s, uint4 v, int C, int W, int kb) {
  constexpr int CH = Slot<T>::CH;
  if constexpr (ALIGNED) {
    if constexpr (CH == 2)
      v = make_uint4(__builtin_amdgcn_perm(v.z, v.x, 0x05040100u), __builtin_amdgcn_perm(v.z, v.x, 0x07060302u),
                     __builtin_amdgcn_perm(v.w, v.y, 0x05040100u), __builtin_amdgcn_perm(v.w, v.y, 0x07060302u));
    return v;
  } else {
    const int k = (int)((uint32_t)s.gx >> 28), gx = (s.gx << 4) >> 4;
    const int ks = kb + k;
    const bool row_ok = s.goff != NO_ROW;
    const bool ch0 = CH * ks < C, ch1 = (CH == 2) && (2 * ks + 1 < C);
    uint32_t e[4] = {v.x, v.y, v.z, v.w};
    const uint32_t keep = (ch0 ? (CH == 2 ? 0x0000ffffu : 0xffffffffu) : 0u) | (ch1 ? 0xffff0000u : 0u);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int x = gx + i;
      e[i] = (row_ok && x >= 0 && x < W) ? (e[i] & keep) : 0u;
    }
    return make_uint4(e[0], e[1], e[2], e[3]);
  }
}

// ABL (ablation bits, 0 in the product): 1 = skip staging, 2 = skip the MAC loop, 4 = skip the stores
template <typename T, bool ALIGNED, int KC = 4, int ABL = 0>
__global__ __launch_bounds__(NTHREADS, 5)   // <= 96 VGPRs: two 9-wave workgroups per CU
void corr81_fwd_kernel(const T* __restrict__ f1, const T* __restrict__ f2, T* __restrict__ out,
                       int C, int H, int W, int tiles_x, int tiles_y, long long out_bs, float slope) {
  extern __shared__ __attribute__((aligned(16))) uint32_t smem[];   // two buffers of BUF_DWORDS each
  constexpr int BUF_DWORDS = buf_dwords(KC);
  constexpr int TASKS_PER_THREAD = tasks_per_thread(KC);
  constexpr int N1_TASKS = KC * TH * XB;

  const int bid = xcd_remap(blockIdx.x, gridDim.x);
  const int tx = bid % tiles_x;
  const int ty = (bid / tiles_x) % tiles_y;
  const int n = bid / (tiles_x * tiles_y);
  const int x0 = tx * TW, y0 = ty * TH;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int dyi = __builtin_amdgcn_readfirstlane(tid >> 6);     // 0..8  <->  dy = dyi - 4

  // lane -> (row, x-block) following the ds_read_b128 service groups
  // {0-3,12-15,20-27} {4-11,16-19,28-31} {32-35,44-47,52-59} {36-43,48-51,60-63}
  int row, xb;
  {
    const int l = lane & 31;
    const bool g1 = (l >= 4 && l < 12) || (l >= 16 && l < 20) || (l >= 28);
    int k;
    if (!g1) k = (l < 4) ? l : (l < 16 ? l - 8 : l - 12);
    else     k = (l < 12) ? l - 4 : (l < 20 ? l - 8 : l - 16);
    const int g = (lane >> 5) * 2 + (g1 ? 1 : 0);
    row = g + 4 * (k >> 3);
    xb = k & 7;
  }

  float acc[D][PX];
#pragma unroll
  for (int d = 0; d < D; ++d)
#pragma unroll
    for (int p = 0; p < PX; ++p) acc[d][p] = 0.f;

  const int nslots = (C + Slot<T>::CH - 1) / Slot<T>::CH;
  const int nchunks = (nslots + KC - 1) / KC;

  // staging tasks of this thread (TASKS_PER_THREAD x NTHREADS >= tasks per chunk).  Tasks [0, 256) read
  // f1, the rest f2: the split falls on a wave boundary, so the tensor (and its buffer descriptor)
  // is wave-uniform for every task slot.
  const size_t item = (size_t)n * C * H * W;
  const T* f1b = f1 + item;
  const T* f2b = f2 + item;
  const uint32_t item_bytes = (uint32_t)C * (uint32_t)H * (uint32_t)W * (uint32_t)sizeof(typename Elem<T>::store_t);
  StageTask task[TASKS_PER_THREAD];
  __amdgpu_buffer_rsrc_t rsrc[TASKS_PER_THREAD];
  const T* fsel[TASKS_PER_THREAD];
#pragma unroll
  for (int j = 0; j < TASKS_PER_THREAD; ++j) {
    task[j] = make_task<T, KC>(tid + j * NTHREADS, y0, x0, H, W);
    const bool from_f2 = __builtin_amdgcn_readfirstlane((tid & ~63) + j * NTHREADS) >= N1_TASKS;
    fsel[j] = from_f2 ? f2b : f1b;
    rsrc[j] = __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(fsel[j]), 0, ALIGNED ? item_bytes : 0u, 0x00020000);
  }

  uint4 pre[TASKS_PER_THREAD];
  // ---- prologue: chunk 0 -> LDS buffer 0
  if constexpr (!(ABL & 1)) {
#pragma unroll
    for (int j = 0; j < TASKS_PER_THREAD; ++j) pre[j] = task_load<T, ALIGNED>(task[j], rsrc[j], fsel[j], C, H, W, 0);
#pragma unroll
    for (int j = 0; j < TASKS_PER_THREAD; ++j)
      if (task[j].lds >= 0) *reinterpret_cast<uint4*>(smem + task[j].lds) = task_finish<T, ALIGNED>(task[j], pre[j], C, W, 0);
  }
  __syncthreads();

  const int rd1 = row * S1 + 4 * xb;
  const int rd2 = KC * SLOT1 + (row + dyi) * S2 + 4 * xb;
  for (int c = 0; c < nchunks; ++c) {
    const uint32_t* buf = smem + (c & 1) * BUF_DWORDS;
    const bool more = (c + 1) < nchunks;
    // ---- issue the global loads of the NEXT chunk; they stay in flight across the MAC loop
    if constexpr (!(ABL & 1)) {
      if (more) {
#pragma unroll
        for (int j = 0; j < TASKS_PER_THREAD; ++j) pre[j] = task_load<T, ALIGNED>(task[j], rsrc[j], fsel[j], C, H, W, (c + 1) * KC);
      }
    }
    // ---- accumulate this chunk (k-slots beyond C were staged as zeros, so the trip count is fixed)
    if constexpr (!(ABL & 2)) {
#pragma unroll 2      // two k-slots of LDS operands in flight; a full unroll hoists all 16 reads and spills
      for (int k = 0; k < KC; ++k) {
        const uint4 a4 = *reinterpret_cast<const uint4*>(buf + rd1 + k * SLOT1);
        const uint4 b0 = *reinterpret_cast<const uint4*>(buf + rd2 + k * SLOT2);
        const uint4 b1 = *reinterpret_cast<const uint4*>(buf + rd2 + k * SLOT2 + 4);
        const uint4 b2 = *reinterpret_cast<const uint4*>(buf + rd2 + k * SLOT2 + 8);
        const uint32_t a[4] = {a4.x, a4.y, a4.z, a4.w};
        const uint32_t b[12] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w, b2.x, b2.y, b2.z, b2.w};
#pragma unroll
        for (int d = 0; d < D; ++d)
#pragma unroll
          for (int p = 0; p < PX; ++p) acc[d][p] = Slot<T>::mac(a[p], b[p + d], acc[d][p]);
      }
    }
    // ---- land the prefetched chunk in the other buffer (last read one barrier ago)
    if constexpr (!(ABL & 1)) {
      if (more) {
        uint32_t* nb = smem + ((c + 1) & 1) * BUF_DWORDS;
#pragma unroll
        for (int j = 0; j < TASKS_PER_THREAD; ++j)
          if (task[j].lds >= 0) *reinterpret_cast<uint4*>(nb + task[j].lds) = task_finish<T, ALIGNED>(task[j], pre[j], C, W, (c + 1) * KC);
      }
    }
    if (more) __syncthreads();
  }

  // ---- epilogue
  const int y = y0 + row, x = x0 + 4 * xb;
  if (y >= H || x >= W) return;
  const float fC = (float)C, invC = 1.0f / fC;
  using st = typename Elem<T>::store_t;
  st* obase = reinterpret_cast<st*>(out) + (size_t)n * out_bs + ((size_t)(dyi * D) * H + y) * W + x;
  const size_t cstride = (size_t)H * W;
#pragma unroll
  for (int d = 0; d < D; ++d) {
    float v[PX];
#pragma unroll
    for (int p = 0; p < PX; ++p) {
      float t = acc[d][p] * invC;
      if constexpr (sizeof(st) == 4) {
        // fp32 output is the parity mode: one Newton step makes acc*invC the correctly rounded acc/C
        // (`reduce_sum / nelems`, correlation_cuda_kernel.cu:108) without a 10-instruction IEEE divide
        const float r = __builtin_fmaf(-t, fC, acc[d][p]);
        t = __builtin_fmaf(r, invC, t);
      }
      v[p] = (slope != 0.f) ? fmaxf(t, t * slope) : t;     // LeakyReLU for 0 < slope < 1
    }
    st* o = obase + d * cstride;
    if constexpr (ABL & 4) {
#pragma unroll
      for (int p = 0; p < PX; ++p) asm volatile("" ::"v"(v[p]));
    } else if constexpr (ALIGNED) {
      if constexpr (sizeof(st) == 4) {
        *reinterpret_cast<float4*>(o) = make_float4(v[0], v[1], v[2], v[3]);
      } else {
        *reinterpret_cast<uint2*>(o) = make_uint2(pack2<T>(v[0], v[1]), pack2<T>(v[2], v[3]));
      }
    } else {
#pragma unroll
      for (int p = 0; p < PX; ++p)
        if (x + p < W) Elem<T>::store(reinterpret_cast<T*>(o) + p, v[p]);
    }
  }
}

}  // namespace corr
}  // namespace upf
