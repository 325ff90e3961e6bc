// Cross-translation-unit helpers of libupflow_hip.so (C++ linkage, not part of the C-ABI).
#pragma once
#include "common.hpp"

namespace upf {
namespace misc {
// (count, mean, M2) partials of the rows of two [N,HW] tensors in ONE launch: ws[2N][nseg][3], and their final (mean, 1/std)
// pairs fin[2N] (nullable); returns nseg.
// W / pitch (optional): the planes' rows are `pitch` elements apart (pitch > W); same statistics, bit for bit, as the contiguous form.
int launch_stats2(const void* x1, const void* x2, float* ws, float2* fin, long long N, int HW, int dtype, hipStream_t stream, int W = 0, int pitch = 0);
int stats2_nseg(long long N, int HW);
}  // namespace misc
}  // namespace upf
