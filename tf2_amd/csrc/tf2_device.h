// Device-side helpers shared by the HIP kernels (include after <hip/hip_runtime.h>).
#pragma once
#include <hip/hip_runtime.h>
#include "tf2_internal.h"

namespace tf2 {

// n / d with the (m, s) pair of set_fast_div (tf2_internal.h): one v_mul_hi_u32 and a shift instead of the
// ~25-instruction integer division sequence -- the pixel decode runs once per block and lane, and the short-K
// layers are bound by VALU issue.
__device__ __forceinline__ int fast_div(int n, uint32_t m, int s) {
  return s < 0 ? n : (int)(__umulhi((unsigned)n, m) >> s);
}

}  // namespace tf2
