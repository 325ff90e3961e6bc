// Which fp32 -> fp16 conversions round ties to even on gfx950?  scalar (_Float16) cast vs __builtin_convertvector pair.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstring>
typedef __attribute__((ext_vector_type(2))) float f32x2_t;
typedef __attribute__((ext_vector_type(2))) _Float16 f16x2_t;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
__global__ void k(const float* in, uint32_t* out, int n) {
  int i = threadIdx.x;
  if (i >= n) return;
  float f = in[i];
  uint16_t a = __builtin_bit_cast(uint16_t, (_Float16)f);
  f32x2_t v = {f, f};
  uint32_t b = __builtin_bit_cast(uint32_t, __builtin_convertvector(v, f16x2_t));
  uint16_t c = __builtin_bit_cast(uint16_t, (__bf16)f);
  uint32_t d = __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2_t));
  out[4 * i] = a; out[4 * i + 1] = b; out[4 * i + 2] = c; out[4 * i + 3] = d;
}
int main() {
  // fp16 has 10 mantissa bits: fp32 mantissa bit 12 set and bits 0..11 clear = exact tie; even/odd target mantissa
  uint32_t bits[] = {0x3f801000u /*1 + 2^-11: tie, even below*/, 0x3f803000u /*tie, odd below*/, 0x3f801001u, 0x3f800fffu,
                     0x3e801000u, 0x3e803000u, 0xbf801000u, 0xbf803000u,
                     0x3f808000u /*bf16 tie even*/, 0x3f818000u /*bf16 tie odd*/, 0x33801000u /*fp16 denormal range*/, 0x38001000u};
  const int n = sizeof(bits) / 4;
  float h[n]; memcpy(h, bits, sizeof(bits));
  float* d; uint32_t* o;
  hipMalloc(&d, sizeof(h)); hipMalloc(&o, n * 16);
  hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, o, n);
  uint32_t r[n * 4];
  hipMemcpy(r, o, n * 16, hipMemcpyDeviceToHost);
  for (int i = 0; i < n; ++i)
    printf("%08x (%.9g): f16 scalar %04x  f16 pair %08x   bf16 scalar %04x  bf16 pair %08x\n", bits[i], h[i], r[4 * i], r[4 * i + 1], r[4 * i + 2], r[4 * i + 3]);
  return 0;
}
