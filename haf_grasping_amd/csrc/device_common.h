// device_common.h -- small device helpers shared by the kernel translation units (prestages.hip, features.hip, contraction.hip,
// recheck.hip, vote.hip).
#pragma once
#include "kernels.h"
#include "decq.h"
#include <string.h>
#include <type_traits>

namespace haf {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef double f64x4 __attribute__((ext_vector_type(4)));
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ int f2key(float f)
{
    int b = __float_as_int(f);
    return b >= 0 ? b : (b ^ 0x7FFFFFFF);
}
__device__ __forceinline__ float key2f(int k)
{
    return __int_as_float(k >= 0 ? k : (k ^ 0x7FFFFFFF));
}

// entries of a list of `total` that fall into the window [off, off + cap)
__device__ __forceinline__ int window_count(int total, int off, int cap) { return max(0, min(total - off, cap)); }

// Upper bound of sqrt(x) to 1e-9 relative without a transcendental instruction (the guard band is never checked bit for
// bit by a test, so nothing in it may hang on the v_exp/v_rsq result hazard described in screen.hip): 1/sqrt(x) by the
// exponent-halving bit trick and four Newton steps r <- r (1.5 - 0.5 x r^2), which only multiply and add.
__device__ __forceinline__ double sqrt_upper(double x)
{
    if (!(x > 0.0)) return x == 0.0 ? 0.0 : x + x;      // -0/+0 -> 0; negative or NaN -> NaN (the evaluation is then never trusted)
    double r = __longlong_as_double(0x5FE6EB50C7B537A9LL - (__double_as_longlong(x) >> 1));
#pragma unroll
    for (int it = 0; it < 4; it++) r = r * fma(-0.5 * x * r, r, 1.5);
    return x * r * (1.0 + 1e-9);
}
// 2^z - 1 <= ln2 z + 0.26 z^2 for 0 <= z < 0.05 (y = z ln2: e^y - 1 <= y + y^2/2 e^y)
__device__ __forceinline__ double exp2m1_upper(double z) { return 0.69314718056 * z + 0.26 * z * z; }

}  // namespace haf
