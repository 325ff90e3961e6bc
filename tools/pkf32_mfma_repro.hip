// Standalone reproducer attempt for DESIGN §4c: do packed-fp32 VALU instructions (v_pk_add_f32 / v_pk_mul_f32)
// return wrong results while a v_mfma_f32_16x16x32 kernel shares the SIMDs?  NO library code: two ~50-line kernels.
//   pk_kernel    every thread walks a table of fp32 pairs held in registers / L1 and computes  r = (x - m) * s  twice per element pair:
//                once with v_pk_add_f32 + v_pk_mul_f32 in the operand-select / negation forms the compiler had generated (inline asm), once with v_sub_f32 + v_mul_f32 (inline asm), compares the
//                bits and counts mismatches in a global counter.  Alone on the GPU the two agree by construction.
//   mfma_kernel  back-to-back v_mfma_f32_16x16x32_bf16 (MODE 0) or v_mfma_f32_32x32x16_bf16 (MODE 1) on register operands.
// Runs: pk alone; pk beside mfma16x16x32 on a second stream; pk beside mfma32x32x16 (control); each REP times.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/pkf32_mfma_repro.hip -o /tmp/pkf32_repro && /tmp/pkf32_repro
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;

typedef __attribute__((ext_vector_type(4))) short s16x4;
// OWN_MFMA: the wave also issues v_mfma_f32_4x4x4_16b_bf16 between its packed operations (like corr81_allc_kernel, whose loader's packed
// fp32 arithmetic is followed by its own 4x4x4 matrix work)
// FORM: 0 = plain v_pk_add / v_pk_mul on a negated copy; 1 = operand-select broadcasts only; 2 = negation modifiers only; 3 = both (the compiler's forms)
template <bool OWN_MFMA, int FORM = 3>
__global__ __launch_bounds__(576) void pk_kernel(const float* __restrict__ tab, int n, int iters, unsigned long long* __restrict__ bad, float* __restrict__ sink) {
  extern __shared__ float lds[];                       // (68 KB like corr81_allc_kernel: two workgroups per CU)
  const int tid = threadIdx.x;
  for (int i = tid; i < 17408; i += blockDim.x) lds[i] = tab[i % n];
  __syncthreads();
  unsigned long long local_bad = 0;
  float acc = 0.f;
  f32x4 macc = {0.f, 0.f, 0.f, 0.f};
  for (int it = 0; it < iters; ++it) {
#pragma unroll 4
    for (int k = 0; k < 64; ++k) {
      const int j = (tid * 2 + k * 1152 + it * 7) % 17400;
      f32x2 x = {lds[j], lds[j + 1]};
      const f32x2 m = {lds[(j + 5) % 17400], lds[(j + 5) % 17400]}, s = {lds[(j + 9) % 17400], lds[(j + 9) % 17400]};
      // the EXACT forms hipcc generated in corr81_allc_kernel's loader (disassembly of a build with packed fp32 enabled): the (mean, 1/std)
      // pair sits in one 64-bit register pair, the subtraction broadcasts its LOW half with negation modifiers, the multiplication its HIGH half
      const f32x2 ms = {m.x, s.x};
      f32x2 d, r;
      if constexpr (FORM == 3) {
        asm volatile("v_pk_add_f32 %0, %1, %2 op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]" : "=v"(d) : "v"(x), "v"(ms));
        asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1]" : "=v"(r) : "v"(d), "v"(ms));
      } else if constexpr (FORM == 1) {
        const f32x2 nms = {-m.x, s.x};
        asm volatile("v_pk_add_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(d) : "v"(x), "v"(nms));
        asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1]" : "=v"(r) : "v"(d), "v"(nms));
      } else if constexpr (FORM == 2) {
        const f32x2 mm = {m.x, m.x}, ss = {s.x, s.x};
        asm volatile("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(d) : "v"(x), "v"(mm));
        asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(r) : "v"(d), "v"(ss));
      } else {
        const f32x2 nm = {-m.x, -m.x}, ss = {s.x, s.x};
        asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(d) : "v"(x), "v"(nm));
        asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(r) : "v"(d), "v"(ss));
      }
      float d0, d1, r0, r1;
      asm volatile("v_sub_f32 %0, %1, %2" : "=v"(d0) : "v"(x.x), "v"(m.x));
      asm volatile("v_sub_f32 %0, %1, %2" : "=v"(d1) : "v"(x.y), "v"(m.y));
      asm volatile("v_mul_f32 %0, %1, %2" : "=v"(r0) : "v"(d0), "v"(s.x));
      asm volatile("v_mul_f32 %0, %1, %2" : "=v"(r1) : "v"(d1), "v"(s.y));
      local_bad += (__float_as_uint(r.x) != __float_as_uint(r0)) + (__float_as_uint(r.y) != __float_as_uint(r1));
      acc += r.x + r.y;
      if constexpr (OWN_MFMA) {
        const s16x4 ma = {(short)(j & 0x3f7f), (short)(k | 0x3f00), (short)0x3f80, (short)0x3e80};
        macc = __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(ma, ma, macc, 0, 0, 0);
      }
    }
  }
  acc += macc[0] + macc[3];
  if (local_bad) atomicAdd(bad, local_bad);
  if (acc == 123.456f) sink[0] = acc;
}

template <int MODE>
__global__ __launch_bounds__(256) void mfma_kernel(int iters, float* __restrict__ sink, int prio) {
  __shared__ bf16x8 ops[256];
  if (prio) __builtin_amdgcn_s_setprio(2);             // (conv_kernel raises its wave priority for the matrix phase)
  bf16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(0.001f * (threadIdx.x + i)); b[i] = (__bf16)(0.002f * (threadIdx.x * 3 + i)); }
  if constexpr (MODE == 0) {
    f32x4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
    ops[threadIdx.x] = a;
    __syncthreads();
    for (int it = 0; it < iters; ++it) {
      b = ops[(threadIdx.x + it) & 255];               // (an LDS operand read between the matrix instructions, like the real kernel)
      c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c0, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c1, 0, 0, 0);
      c2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c2, 0, 0, 0);
      c3 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c3, 0, 0, 0);
    }
    if (c0[0] + c1[1] + c2[2] + c3[3] == 123.456f) sink[1] = c0[0];
  } else {
    f32x16 c0, c1;
    for (int i = 0; i < 16; ++i) { c0[i] = 0.f; c1[i] = 0.f; }
    for (int it = 0; it < iters; ++it) {
      c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c0, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c1, 0, 0, 0);
    }
    if (c0[0] + c1[1] == 123.456f) sink[1] = c0[0];
  }
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

int main() {
  const int n = 4096, REP = 12;
  float* htab = new float[n];
  uint32_t st = 12345u;
  for (int i = 0; i < n; ++i) { st = st * 1664525u + 1013904223u; htab[i] = ((st >> 8) * (1.0f / 8388608.0f) - 1.0f) * 3.0f; }
  float *tab, *sink;
  unsigned long long* bad;
  CK(hipMalloc(&tab, n * 4)); CK(hipMalloc(&sink, 64)); CK(hipMalloc(&bad, 8));
  CK(hipMemcpy(tab, htab, n * 4, hipMemcpyHostToDevice));
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&pk_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, 69632));
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&pk_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, 69632));
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&pk_kernel<false, 0>), hipFuncAttributeMaxDynamicSharedMemorySize, 69632));
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&pk_kernel<false, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, 69632));
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&pk_kernel<false, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, 69632));
  hipStream_t sa, sb;
  CK(hipStreamCreate(&sa)); CK(hipStreamCreate(&sb));
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  printf("# %s, %d CUs; pk_kernel: 480 workgroups x 576 threads, 68 KB LDS; mfma kernels: 2048 x 256 threads; %d repetitions each\n", prop.name, prop.multiProcessorCount, REP);
  const char* names[3] = {"pk alone", "pk beside v_mfma_f32_16x16x32_bf16", "pk beside v_mfma_f32_32x32x16_bf16 (control)"};
  for (int own = 0; own < 2; ++own)
    for (int prio = 0; prio < 2; ++prio)
      for (int grid = 512; grid <= 2048; grid *= 4) {
        printf("## pk kernel %s its own 4x4x4 MFMAs; neighbour: %d workgroups, wave priority %d\n", own ? "WITH" : "without", grid, prio ? 2 : 0);
        for (int mode = (prio || grid > 512) ? 1 : 0; mode < 3; ++mode) {
          unsigned long long total = 0; int bad_runs = 0;
          for (int r = 0; r < REP; ++r) {
            CK(hipMemset(bad, 0, 8));
            CK(hipDeviceSynchronize());
            if (mode == 1) hipLaunchKernelGGL(mfma_kernel<0>, dim3(grid), dim3(256), 0, sb, 100000 * (2048 / grid), sink, prio);
            if (mode == 2) hipLaunchKernelGGL(mfma_kernel<1>, dim3(grid), dim3(256), 0, sb, 50000 * (2048 / grid), sink, prio);
            if (own) hipLaunchKernelGGL(pk_kernel<true>, dim3(480), dim3(576), 69632, sa, tab, n, 300, bad, sink);
            else hipLaunchKernelGGL(pk_kernel<false>, dim3(480), dim3(576), 69632, sa, tab, n, 300, bad, sink);
            CK(hipDeviceSynchronize());
            unsigned long long h = 0;
            CK(hipMemcpy(&h, bad, 8, hipMemcpyDeviceToHost));
            total += h; bad_runs += h != 0;
          }
          printf("%-48s mismatching launches %d / %d, mismatching results %llu of %llu\n", names[mode], bad_runs, REP, total, (unsigned long long)REP * 480ull * 576ull * 300ull * 64ull * 2ull);
        }
      }
  printf("## which instruction form: pk kernel (no own MFMAs) alone / beside 512 workgroups of v_mfma_f32_16x16x32_bf16, priority 0\n");
  const char* forms[4] = {"plain (no modifiers)", "op_sel / op_sel_hi broadcasts only", "neg_lo / neg_hi only", "broadcasts + negation (the compiler's forms)"};
  for (int form = 0; form < 4; ++form)
    for (int mode = 0; mode < 2; ++mode) {
      unsigned long long total = 0; int bad_runs = 0;
      for (int r = 0; r < REP; ++r) {
        CK(hipMemset(bad, 0, 8));
        CK(hipDeviceSynchronize());
        if (mode == 1) hipLaunchKernelGGL(mfma_kernel<0>, dim3(512), dim3(256), 0, sb, 400000, sink, 0);
        if (form == 0) hipLaunchKernelGGL((pk_kernel<false, 0>), dim3(480), dim3(576), 69632, sa, tab, n, 300, bad, sink);
        if (form == 1) hipLaunchKernelGGL((pk_kernel<false, 1>), dim3(480), dim3(576), 69632, sa, tab, n, 300, bad, sink);
        if (form == 2) hipLaunchKernelGGL((pk_kernel<false, 2>), dim3(480), dim3(576), 69632, sa, tab, n, 300, bad, sink);
        if (form == 3) hipLaunchKernelGGL((pk_kernel<false, 3>), dim3(480), dim3(576), 69632, sa, tab, n, 300, bad, sink);
        CK(hipDeviceSynchronize());
        unsigned long long h = 0;
        CK(hipMemcpy(&h, bad, 8, hipMemcpyDeviceToHost));
        total += h; bad_runs += h != 0;
      }
      printf("%-48s %-20s mismatching launches %d / %d, mismatching results %llu\n", forms[form], mode ? "beside 16x16x32" : "alone", bad_runs, REP, total);
    }
  return 0;
}
