// qmps_roto_math.h - the update rules of the rotosolve drivers (gfx950 only), shared by roto_update_kernel, the whole-run
// D = 2 kernel (qmps_kernels.hip) and the whole-run D = 8 kernel (qmps_roto_d8.hip).
#pragma once
#include <hip/hip_runtime.h>

#include "qmps_device.h"
#include "qmps_roto_rule.h"

namespace qmps {

// The reference wraps an angle by arctan2(sin x, cos x) (qmps/rotosolve.py:110, 176-177): the representative of x in
// [-pi, pi].  Same value to rounding without the three transcendental calls (they were ~1.5 us of every parameter update
// of the whole-run kernels, executed by one thread): x - 2 pi rint(x / 2 pi), 2 pi in two pieces.
__device__ __forceinline__ double wrap_pi(double x) {
  const double n = __builtin_rint(x * 0.15915494309189535);
  return dfma(-n, 2.4492935982947064e-16, dfma(-n, 6.283185307179586, x));
}

}  // namespace qmps
