// testkernels.hip -- device code that exists in libhafgrasp_testing.so only (build.py links it into the testing build).
//
// haf_test_mfma_accum: evidence for the ONE assumption about undocumented hardware behaviour in the screening band
// (DESIGN.md 2): "a chain of ten v_mfma_f32_16x16x32_f16 that starts from C deviates from the exact value of
// C + sum_k a_k b_k by at most 2^-18 (|C| + sum_k |a_k b_k|)".  The products of fp16 operands are exact in fp32; what is not
// documented is how the 32 products of an instruction and the running value are added inside the matrix core.  The kernel
// runs the chain exactly as k_svm_screen does (same builtin, same operand layout, C operand of the first instruction = start
// value) on data the host chose, and returns the raw accumulators; the host compares with an fp64 evaluation.
#include "kernels.h"

namespace haf {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 half8 __attribute__((ext_vector_type(8)));

// a: [trial][16 rows][320] fp16, b: [trial][320][16 cols] fp16, c0: [trial][16 cols] start value of every row of that column,
// out: [trial][16 rows][16 cols]
__global__ __launch_bounds__(64) void k_mfma_accum(const _Float16 *__restrict__ a, const _Float16 *__restrict__ b,
                                                   const float *__restrict__ c0, float *__restrict__ out)
{
    const int t = blockIdx.x, lane = threadIdx.x;
    const _Float16 *at = a + (size_t)t * 16 * 320, *bt = b + (size_t)t * 320 * 16;
    const float start = c0[t * 16 + (lane & 15)];
    f32x4 acc = {start, start, start, start};
    for (int s = 0; s < 10; s++) {
        half8 av, bv;
        for (int j = 0; j < 8; j++) {
            const int k = 32 * s + 8 * (lane >> 4) + j;
            av[j] = at[(lane & 15) * 320 + k];             // A[row = lane & 15][k]
            bv[j] = bt[k * 16 + (lane & 15)];               // B[k][col = lane & 15]
        }
        acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(av, bv, acc, 0, 0, 0);
    }
    for (int r = 0; r < 4; r++) out[(size_t)t * 256 + (4 * (lane >> 4) + r) * 16 + (lane & 15)] = acc[r];   // D[row = 4 (lane >> 4) + r][col = lane & 15]
}

// haf_test_mfma_rate: what THIS GPU sustains on the instruction the screening kernel is made of, for bench.py's context line
// (the fp16 MFMA pipe runs at the clock the chip holds under the load, which differs from box to box).  Random
// fp16 operands in registers, eight independent accumulators, nothing but v_mfma_f32_16x16x32_f16 in the loop; two 4-wave
// workgroups per CU like k_svm_screen.  flop = blocks * 4 waves * iters * 32 * 16384.
__global__ __launch_bounds__(256, 2) void k_mfma_rate(const half8 *__restrict__ in, float *__restrict__ out, int iters)
{
    const int tid = blockIdx.x * 256 + threadIdx.x;
    half8 a[8], b[8];
    for (int i = 0; i < 8; i++) { a[i] = in[(tid * 16 + i) & 65535]; b[i] = in[(tid * 16 + 8 + i) & 65535]; }
    f32x4 acc[8];
    for (int i = 0; i < 8; i++) acc[i] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < 8; i++)
#pragma unroll
            for (int j = 0; j < 4; j++)
                acc[(i * 4 + j) & 7] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[i], b[(i + j) & 7], acc[(i * 4 + j) & 7], 0, 0, 0);
    }
    float s = 0.0f;
    for (int i = 0; i < 8; i++) for (int j = 0; j < 4; j++) s += acc[i][j];
    out[tid] = s;
}

void launch_mfma_rate_test(const void *in, float *out, int blocks, int iters, hipStream_t s)
{
    hipLaunchKernelGGL(k_mfma_rate, dim3(blocks), dim3(256), 0, s, (const half8 *)in, out, iters);
}

void launch_mfma_accum_test(const void *a, const void *b, const float *c0, float *out, int trials, hipStream_t s)
{
    hipLaunchKernelGGL(k_mfma_accum, dim3(trials), dim3(64), 0, s, (const _Float16 *)a, (const _Float16 *)b, c0, out);
}

}  // namespace haf
