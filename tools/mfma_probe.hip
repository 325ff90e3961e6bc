// Probe: (1) throughput of candidate MAC instructions on gfx950, (2) operand layout of v_mfma_f32_4x4x4_16b_bf16.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cstdint>
#include <cstring>
typedef __attribute__((ext_vector_type(4))) short s4;
typedef __attribute__((ext_vector_type(8))) short s8;
typedef __attribute__((ext_vector_type(4))) float f4;
typedef __attribute__((ext_vector_type(2))) __bf16 bf2;

template <int MODE>
__global__ void thr(float* out, int iters) {
  float a0 = threadIdx.x * 1e-3f, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
  uint32_t x = 0x3f803f80u + threadIdx.x, y = 0x3f003f00u;
  f4 m0 = {0, 0, 0, 0}, m1 = m0, m2 = m0, m3 = m0, m4 = m0, m5 = m0, m6 = m0, m7 = m0;
  s4 sa = {(short)0x3f80, (short)0x3f80, (short)0x3f80, (short)0x3f80}, sb = sa;
  s8 ta = {(short)0x3f80, (short)0x3f80, (short)0x3f80, (short)0x3f80, (short)0x3f80, (short)0x3f80, (short)0x3f80, (short)0x3f80}, tb = ta;
  for (int i = 0; i < iters; ++i) {
    if constexpr (MODE == 0) {
#define D2(acc) acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf2, x), __builtin_bit_cast(bf2, y), acc, false);
      D2(a0) D2(a1) D2(a2) D2(a3) D2(a4) D2(a5) D2(a6) D2(a7) D2(a0) D2(a1) D2(a2) D2(a3) D2(a4) D2(a5) D2(a6) D2(a7)
    } else if constexpr (MODE == 1) {
      float fx = __uint_as_float(x), fy = __uint_as_float(y);
#define FM(acc) acc = __builtin_fmaf(fx, fy, acc);
      FM(a0) FM(a1) FM(a2) FM(a3) FM(a4) FM(a5) FM(a6) FM(a7) FM(a0) FM(a1) FM(a2) FM(a3) FM(a4) FM(a5) FM(a6) FM(a7)
    } else if constexpr (MODE == 2) {
#define M4(acc) acc = __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(sa, sb, acc, 0, 0, 0);
      M4(m0) M4(m1) M4(m2) M4(m3) M4(m4) M4(m5) M4(m6) M4(m7) M4(m0) M4(m1) M4(m2) M4(m3) M4(m4) M4(m5) M4(m6) M4(m7)
    } else if constexpr (MODE == 3) {
#define M16(acc) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ta, tb, acc, 0, 0, 0);
      M16(m0) M16(m1) M16(m2) M16(m3) M16(m4) M16(m5) M16(m6) M16(m7) M16(m0) M16(m1) M16(m2) M16(m3) M16(m4) M16(m5) M16(m6) M16(m7)
    }
    asm volatile("" : "+v"(x));
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + m0[0] + m1[1] + m2[2] + m3[3] + m4[0] + m5[0] + m6[0] + m7[0];
}

template <int MODE>
void bench(const char* name, double macs_per_lane_instr) {
  float* out; (void)hipMalloc(&out, 1024 * 1024 * 4);
  const int blocks = 1024, threads = 256, iters = 2000;
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  hipLaunchKernelGGL(thr<MODE>, dim3(blocks), dim3(threads), 0, 0, out, 10);
  (void)hipEventRecord(e0);
  hipLaunchKernelGGL(thr<MODE>, dim3(blocks), dim3(threads), 0, 0, out, iters);
  (void)hipEventRecord(e1); (void)hipDeviceSynchronize();
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  double waves = blocks * threads / 64.0, instr = waves * iters * 16.0;
  double per_simd_cycles = ms * 1e-3 * 2.4e9 / (instr / 1024.0);   // 1024 SIMDs, assuming 2.4 GHz
  printf("%-28s %8.3f ms  %.2f cycles/wave-instr/SIMD @2.4GHz  -> %.1f TMAC/s\n", name, ms, per_simd_cycles, instr * macs_per_lane_instr / (ms * 1e-3) / 1e12);
  (void)hipFree(out);
}

__global__ void layout(const uint16_t* A, const uint16_t* B, float* D) {
  // lane l supplies 4 bf16 of A and of B; result 4 floats
  int l = threadIdx.x;
  s4 a, b;
  for (int k = 0; k < 4; ++k) { a[k] = (short)A[l * 4 + k]; b[k] = (short)B[l * 4 + k]; }
  f4 acc = {0, 0, 0, 0};
  acc = __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(a, b, acc, 0, 0, 0);
  for (int r = 0; r < 4; ++r) D[l * 4 + r] = acc[r];
}

static uint16_t bf(float f) { uint32_t u; memcpy(&u, &f, 4); return (uint16_t)(u >> 16); }

int main() {
  bench<0>("v_dot2c_f32_bf16", 64 * 2);
  bench<1>("v_fma_f32", 64 * 1);
  bench<2>("v_mfma_f32_4x4x4_16b_bf16", 1024);
  bench<3>("v_mfma_f32_16x16x32_bf16", 16 * 16 * 32);
  // layout probe: hypothesis  A: lane l -> block l/4, row i=l%4, k=0..3 ; B: lane l -> block l/4, col j=l%4, k=0..3 ;
  //               D: lane l, reg r -> block l/4, D[i=r][j=l%4]
  std::vector<uint16_t> A(256), B(256);
  std::vector<float> Af(256), Bf(256);
  for (int l = 0; l < 64; ++l) for (int k = 0; k < 4; ++k) {
    Af[l * 4 + k] = (float)((l % 4) + 1) * (k == 0 ? 1 : (k == 1 ? 8 : (k == 2 ? 64 : 0)));   // encodes row i in base-8 digits of k
    Bf[l * 4 + k] = (k == (l % 4) % 3) ? (float)(1 + (l / 4 % 2)) : 0.f;                        // picks a k depending on column j
    A[l * 4 + k] = bf(Af[l * 4 + k]); B[l * 4 + k] = bf(Bf[l * 4 + k]);
  }
  uint16_t *dA, *dB; float* dD;
  (void)hipMalloc(&dA, 512); (void)hipMalloc(&dB, 512); (void)hipMalloc(&dD, 1024);
  (void)hipMemcpy(dA, A.data(), 512, hipMemcpyHostToDevice); (void)hipMemcpy(dB, B.data(), 512, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(layout, dim3(1), dim3(64), 0, 0, dA, dB, dD);
  std::vector<float> D(256);
  (void)hipMemcpy(D.data(), dD, 1024, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int l = 0; l < 64; ++l) for (int r = 0; r < 4; ++r) {
    int blk = l / 4, j = l % 4, i = r;
    float want = 0;
    for (int k = 0; k < 4; ++k) want += Af[(blk * 4 + i) * 4 + k] * Bf[(blk * 4 + j) * 4 + k];
    if (want != D[l * 4 + r]) { if (bad < 8) printf("mismatch lane %d reg %d: got %g want %g\n", l, r, D[l * 4 + r], want); ++bad; }
  }
  printf("4x4x4_16b layout hypothesis: %s (%d mismatches)\n", bad ? "WRONG" : "CONFIRMED", bad);
  return 0;
}
