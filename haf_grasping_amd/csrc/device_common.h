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

// Ordered hand-over lists (round 5).  A tier's combine kernel leaves one bit per entry of its window -- "undecided" -- in ballot words
// (word w = entries 64 w .. 64 w + 63 of the window); list_compact_body, ONE workgroup of kListCompactThreads, appends the flagged
// entries' evaluations to the next tier's list in the order of the window and publishes the new total: the list is the same from run to
// run (until then every undecided entry took its slot with an atomicAdd, whoever came first).  Windows follow each other on the
// stream, so "append" is counters[out_slot] as the previous window left it.
constexpr int kListCompactThreads = 1024;
__device__ __forceinline__ void list_compact_body(const unsigned long long *__restrict__ words, int n_entries, const int *__restrict__ src,
                                                  int *__restrict__ dst, int dst_cap, int *__restrict__ counters, int out_slot, int *s_scan)
{
    const int t = threadIdx.x;
    const int n_words = (n_entries + 63) >> 6;
    const int chunk = (n_words + kListCompactThreads - 1) / kListCompactThreads;
    const int w0 = min(t * chunk, n_words), w1 = min(w0 + chunk, n_words);
    int mine = 0;
    for (int w = w0; w < w1; w++) mine += __popcll(words[w]);
    s_scan[t] = mine;
    __syncthreads();
    for (int o = 1; o < kListCompactThreads; o <<= 1) {
        const int v = (t >= o) ? s_scan[t - o] : 0;
        __syncthreads();
        s_scan[t] += v;
        __syncthreads();
    }
    const int base0 = counters[out_slot];
    int slot = base0 + s_scan[t] - mine;
    const int total = s_scan[kListCompactThreads - 1];
    for (int w = w0; w < w1; w++) {
        unsigned long long m = words[w];
        while (m) {
            const int b = __ffsll((long long)m) - 1;
            m &= m - 1;
            if (slot < dst_cap) dst[slot] = src[w * 64 + b];
            slot++;
        }
    }
    __syncthreads();                                     // every thread has read the old total
    if (t == 0) counters[out_slot] = base0 + total;
}

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
