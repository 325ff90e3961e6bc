// Shared device/host helpers for libupflow_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>

#include "../../include/upflow_hip.h"

namespace upf {

// ---- error plumbing ---------------------------------------------------------------------------
void set_error(const char* fmt, ...);
int  check_launch(const char* what);          // hipGetLastError() -> 0 / positive hipError_t

#define UPF_REQUIRE(cond, code, ...)                 \
  do {                                               \
    if (!(cond)) {                                   \
      ::upf::set_error(__VA_ARGS__);                 \
      return (code);                                 \
    }                                                \
  } while (0)

static inline bool aligned_to(const void* p, size_t a) { return (reinterpret_cast<uintptr_t>(p) % a) == 0; }
static inline int cdiv(int a, int b) { return (a + b - 1) / b; }

// ---- 16-bit element helpers ---------------------------------------------------------------------
struct bf16_t { uint16_t v; };   // storage-only tags; arithmetic is always fp32
struct f16_t  { uint16_t v; };

__device__ __forceinline__ float bf16_bits_to_f32(uint32_t b) { return __uint_as_float(b << 16); }
__device__ __forceinline__ uint16_t f32_to_bf16_bits(float f) {            // round-to-nearest-even
  return __builtin_bit_cast(uint16_t, (__bf16)f);
}
__device__ __forceinline__ float f16_bits_to_f32(uint32_t b) {
  return (float)__builtin_bit_cast(_Float16, (uint16_t)b);
}
__device__ __forceinline__ uint16_t f32_to_f16_bits(float f) {
  return __builtin_bit_cast(uint16_t, (_Float16)f);
}

template <typename T> struct Elem;
template <> struct Elem<float> {
  using store_t = float;
  static __device__ __forceinline__ float load(const float* p) { return *p; }
  static __device__ __forceinline__ void store(float* p, float v) { *p = v; }
};
template <> struct Elem<bf16_t> {
  using store_t = uint16_t;
  static __device__ __forceinline__ float load(const bf16_t* p) { return bf16_bits_to_f32(p->v); }
  static __device__ __forceinline__ void store(bf16_t* p, float v) { p->v = f32_to_bf16_bits(v); }
};
template <> struct Elem<f16_t> {
  using store_t = uint16_t;
  static __device__ __forceinline__ float load(const f16_t* p) { return f16_bits_to_f32(p->v); }
  static __device__ __forceinline__ void store(f16_t* p, float v) { p->v = f32_to_f16_bits(v); }
};

// two fp32 -> one dword of two 16-bit elements (lo = first).  clang lowers the __bf16 conversion to
// v_cvt_pk_bf16_f32 on gfx950 (round-to-nearest-even, the same rounding as torch's .to(bfloat16)).
typedef __attribute__((ext_vector_type(2))) float f32x2_t;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
typedef __attribute__((ext_vector_type(2))) _Float16 f16x2_t;
template <typename T> __device__ __forceinline__ uint32_t pack2(float a, float b);
template <> __device__ __forceinline__ uint32_t pack2<bf16_t>(float a, float b) {
  f32x2_t v = {a, b};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2_t));
}
template <> __device__ __forceinline__ uint32_t pack2<f16_t>(float a, float b) {
  f32x2_t v = {a, b};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, f16x2_t));
}
template <> __device__ __forceinline__ uint32_t pack2<float>(float a, float) { return __float_as_uint(a); }

// XCD-aware block remap: the dispatcher places block b on XCD b % 8 (MI355X_MICROARCH.md), each
// XCD has a private 4 MiB L2.  Give every XCD a contiguous run of tiles so that neighbouring tiles
// (which share their 4-pixel halos) hit the same L2.  Bijective for any grid size.
__device__ __forceinline__ int xcd_remap(int bid, int nblocks) {
  const int NX = 8;
  int xcd = bid % NX, idx = bid / NX;
  int q = nblocks / NX, r = nblocks % NX;
  int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + idx;
}

}  // namespace upf

// dtype dispatch for host launchers
#define UPF_DISPATCH(dtype, T, ...)                                              \
  switch (dtype) {                                                               \
    case UPF_F32:  { using T = float;        __VA_ARGS__; break; }               \
    case UPF_F16:  { using T = ::upf::f16_t;  __VA_ARGS__; break; }              \
    case UPF_BF16: { using T = ::upf::bf16_t; __VA_ARGS__; break; }              \
    default: ::upf::set_error("unknown dtype %d", (int)(dtype)); return UPF_EDTYPE; \
  }
