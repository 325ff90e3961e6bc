// Merge of the per-segment (count, mean, M2) partials of normalize_stats_kernel into a row's mean and 1/std — shared by
// normalize_apply_kernel (csrc/misc.hip) and the fused loader of corr81_allc_kernel.hpp, which must agree BIT FOR BIT
// (the fused path is tested for bit-equality against normalize + corr81).  Contraction is switched off inside the
// function: HIP's __fmul_rn / __fadd_rn are plain `*` / `+` (clang's __clang_hip_math.h:271) and WOULD be fused into
// FMAs under the default -ffp-contract=fast — the statistics of one row in a few thousand then differ by one ulp between
// translation units.  (Both users are also compiled with -ffp-contract=off, upflow_pytorch_amd/_build.py.)
#pragma once
#include "common.hpp"

namespace upf {

struct RowStats { float mean, std, rstd; };

// w: [nseg][3] partials of one row, merged in fixed order with Chan's parallel-variance formula;
// unbiased variance over HW elements (torch.var default, model/upflow.py:114), std = sqrt(var + 1e-16) (:126).
// (the three steps separately, so that a caller can hold the partials in registers — corr81_allc_kernel.hpp loads them
// BEFORE its feature loads and merges while those are in flight; one definition of the arithmetic = one set of bits)
struct MergeState { float n, mean, m2; };
__device__ __forceinline__ void norm_merge_add(MergeState& s, float nb, float mb, float m2b) {
#pragma clang fp contract(off)
  const float tot = __fadd_rn(s.n, nb), delta = __fsub_rn(mb, s.mean);
  s.mean = __fadd_rn(s.mean, __fmul_rn(delta, __fdiv_rn(nb, tot)));
  s.m2 = __fadd_rn(__fadd_rn(s.m2, m2b), __fmul_rn(__fmul_rn(delta, delta), __fdiv_rn(__fmul_rn(s.n, nb), tot)));
  s.n = tot;
}
__device__ __forceinline__ RowStats norm_merge_finish(const MergeState& s, int HW) {
#pragma clang fp contract(off)
  RowStats r;
  r.mean = s.mean;
  r.std = __fsqrt_rn(__fadd_rn(__fdiv_rn(s.m2, (float)(HW - 1)), 1e-16f));
  r.rstd = __fdiv_rn(1.0f, r.std);
  return r;
}
__device__ __forceinline__ RowStats norm_merge_full(const float* __restrict__ w, int nseg, int HW) {
  MergeState s = {w[0], w[1], w[2]};
  for (int k = 1; k < nseg; ++k) norm_merge_add(s, w[3 * k], w[3 * k + 1], w[3 * k + 2]);
  return norm_merge_finish(s, HW);
}

__device__ __forceinline__ float2 norm_merge(const float* __restrict__ w, int nseg, int HW) {
  const RowStats r = norm_merge_full(w, nseg, HW);
  return make_float2(r.mean, r.rstd);
}

}  // namespace upf
