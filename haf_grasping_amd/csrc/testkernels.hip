// testkernels.hip -- device code that exists in libhafgrasp_testing.so only (build.py links it into the testing build).
//
// haf_test_mfma_accum: evidence for the ONE assumption about undocumented hardware behaviour in the screening band
// (DESIGN.md 2): "a chain of ten v_mfma_f32_16x16x32_f16 that starts from C deviates from the exact value of
// C + sum_k a_k b_k by at most 2^-18 (|C| + sum_k |a_k b_k|)".  The products of fp16 operands are exact in fp32; what is not
// documented is how the 32 products of an instruction and the running value are added inside the matrix core.  The kernel
// runs the chain exactly as k_svm_screen does (same builtin, same operand layout, C operand of the first instruction = start
// value) on data the host chose, and returns the raw accumulators; the host compares with an fp64 evaluation.
#include "kernels.h"
#include "decq.h"

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

// the same loop with v_mfma_i32_16x16x64_i8 (the exact-integer tier's instruction): op = blocks * 4 waves * iters * 32 * 32768
typedef int ri32x4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256, 2) void k_mfma_rate_i8(const half8 *__restrict__ in, float *__restrict__ out, int iters)
{
    const int tid = blockIdx.x * 256 + threadIdx.x;
    ri32x4 a[8], b[8];
    for (int i = 0; i < 8; i++) {
        a[i] = __builtin_bit_cast(ri32x4, in[(tid * 16 + i) & 65535]);
        b[i] = __builtin_bit_cast(ri32x4, in[(tid * 16 + 8 + i) & 65535]);
    }
    ri32x4 acc[8];
    for (int i = 0; i < 8; i++) acc[i] = ri32x4{0, 0, 0, 0};
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < 8; i++)
#pragma unroll
            for (int j = 0; j < 4; j++)
                acc[(i * 4 + j) & 7] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a[i], b[(i + j) & 7], acc[(i * 4 + j) & 7], 0, 0, 0);
    }
    int s = 0;
    for (int i = 0; i < 8; i++) for (int j = 0; j < 4; j++) s += acc[i][j];
    out[tid] = (float)s;
}

void launch_mfma_rate_test(const void *in, float *out, int blocks, int iters, hipStream_t s)
{
    if (iters < 0) hipLaunchKernelGGL(k_mfma_rate_i8, dim3(blocks), dim3(256), 0, s, (const half8 *)in, out, -iters);      // (negative: the int8 form)
    else hipLaunchKernelGGL(k_mfma_rate, dim3(blocks), dim3(256), 0, s, (const half8 *)in, out, iters);
}

// Feasibility model of the screening kernel's inner loop for MB row blocks of 16 evaluations per wave (4 = the shipped tiling at
// two workgroups per CU, 8 = one 512-register wave per SIMD): static SV tile images in LDS (no DMA), A fragments in registers,
// per k-step one B fragment read and MB MFMAs, the epilogue (one v_exp_f32 + one fma per accumulator value of the previous column
// block) left to hipcc's scheduling, a barrier per tile.  Timing only.
template <int MB, int THREADS = 256>
__global__ __launch_bounds__(THREADS, (MB == 8 || THREADS == 512) ? 1 : 2) void k_mfma_model(const half8 *__restrict__ in, float *__restrict__ out, int tiles)
{
    __shared__ __attribute__((aligned(16))) char lds[3 * 21504];
    const int tid = threadIdx.x, lane = tid & 63;
    for (int i = tid; i < 3 * 21504 / 16; i += THREADS) reinterpret_cast<half8 *>(lds)[i] = in[(blockIdx.x * 977 + i) & 65535];
    half8 a[10][MB];
#pragma unroll
    for (int s = 0; s < 10; s++)
#pragma unroll
        for (int m = 0; m < MB; m++) a[s][m] = in[(blockIdx.x * 256 + tid + 131 * (s * MB + m)) & 65535];
    __syncthreads();
    f32x4 acc0[MB], acc1[MB];
    float sum[MB][4];
#pragma unroll
    for (int m = 0; m < MB; m++) {
        acc0[m] = f32x4{0, 0, 0, 0}; acc1[m] = f32x4{0, 0, 0, 0};
#pragma unroll
        for (int r = 0; r < 4; r++) sum[m][r] = 0.0f;
    }
    const float cf = 0.37f;
    for (int t = 0; t < tiles; t++) {
        const char *cur = lds + (t % 3) * 21504 + lane * 16;
#pragma unroll
        for (int n = 0; n < 2; n++) {
            f32x4 *acc = n ? acc1 : acc0;
            f32x4 *old = n ? acc0 : acc1;
            half8 b = *reinterpret_cast<const half8 *>(cur + n * 1024);
#pragma unroll
            for (int s = 0; s < 10; s++) {
                half8 b1 = b;
                if (s + 1 < 10) b1 = *reinterpret_cast<const half8 *>(cur + n * 1024 + (s + 1) * 2048);
#pragma unroll
                for (int m = 0; m < MB; m++) {
                    acc[m] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[s][m], b, s == 0 ? f32x4{-1.0f, -1.0f, -1.0f, -1.0f} : acc[m], 0, 0, 0);
                    // epilogue of the previous column block: MB * 4 values over the ten k-steps
                    const int v0 = (s * MB * 4) / 10 + (m * (((s + 1) * MB * 4) / 10 - (s * MB * 4) / 10)) / MB;
                    const int v1 = (s * MB * 4) / 10 + ((m + 1) * (((s + 1) * MB * 4) / 10 - (s * MB * 4) / 10)) / MB;
#pragma unroll
                    for (int v = v0; v < v1; v++) sum[v >> 2][v & 3] = fmaf(cf, __builtin_amdgcn_exp2f(old[v >> 2][v & 3] * 1e-3f), sum[v >> 2][v & 3]);
                }
                b = b1;
            }
        }
        __syncthreads();
    }
    float r = 0.0f;
#pragma unroll
    for (int m = 0; m < MB; m++)
#pragma unroll
        for (int q = 0; q < 4; q++) r += sum[m][q] + acc0[m][q] + acc1[m][q];
    out[blockIdx.x * THREADS + tid] = r;
}

// The same model for 8 row blocks per wave with the register classes chosen by hand: the A fragments of k-steps 3..9 (56 of
// 80) are loaded straight into AccVGPRs and named as such to the MFMA (inline asm, "a" constraint); accumulators, sums and B
// fragments stay in VGPRs, so nothing is copied between the two files inside the loop.
__global__ __launch_bounds__(256, 1) void k_mfma_model8a(const half8 *__restrict__ in, float *__restrict__ out, int tiles)
{
    constexpr int MB = 8, KV = 3;                        // k-steps whose A fragments live in VGPRs
    __shared__ __attribute__((aligned(16))) char lds[3 * 21504];
    const int tid = threadIdx.x, lane = tid & 63;
    for (int i = tid; i < 3 * 21504 / 16; i += 256) reinterpret_cast<half8 *>(lds)[i] = in[(blockIdx.x * 977 + i) & 65535];
    half8 av[KV][MB], aa[10 - KV][MB];
#pragma unroll
    for (int s = 0; s < KV; s++)
#pragma unroll
        for (int m = 0; m < MB; m++) av[s][m] = in[(blockIdx.x * 256 + tid + 131 * (s * MB + m)) & 65535];
#pragma unroll
    for (int s = KV; s < 10; s++)
#pragma unroll
        for (int m = 0; m < MB; m++) {
            const half8 *p = in + ((blockIdx.x * 256 + tid + 131 * (s * MB + m)) & 65535);
            asm volatile("global_load_dwordx4 %0, %1, off" : "=a"(aa[s - KV][m]) : "v"(p) : "memory");
        }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    f32x4 acc0[MB], acc1[MB];
    float sum[MB][4];
#pragma unroll
    for (int m = 0; m < MB; m++) {
        acc0[m] = f32x4{0, 0, 0, 0}; acc1[m] = f32x4{0, 0, 0, 0};
#pragma unroll
        for (int r = 0; r < 4; r++) sum[m][r] = 0.0f;
    }
    const float cf = 0.37f;
    for (int t = 0; t < tiles; t++) {
        const char *cur = lds + (t % 3) * 21504 + lane * 16;
#pragma unroll
        for (int n = 0; n < 2; n++) {
            f32x4 *acc = n ? acc1 : acc0;
            f32x4 *old = n ? acc0 : acc1;
            half8 b = *reinterpret_cast<const half8 *>(cur + n * 1024);
#pragma unroll
            for (int s = 0; s < 10; s++) {
                half8 b1 = b;
                if (s + 1 < 10) b1 = *reinterpret_cast<const half8 *>(cur + n * 1024 + (s + 1) * 2048);
#pragma unroll
                for (int m = 0; m < MB; m++) {
                    if (s < KV) acc[m] = __builtin_amdgcn_mfma_f32_16x16x32_f16(av[s][m], b, s == 0 ? f32x4{-1.0f, -1.0f, -1.0f, -1.0f} : acc[m], 0, 0, 0);
                    else asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc[m]) : "a"(aa[s - KV][m]), "v"(b));
                    const int v0 = (s * MB * 4) / 10 + (m * (((s + 1) * MB * 4) / 10 - (s * MB * 4) / 10)) / MB;
                    const int v1 = (s * MB * 4) / 10 + ((m + 1) * (((s + 1) * MB * 4) / 10 - (s * MB * 4) / 10)) / MB;
#pragma unroll
                    for (int v = v0; v < v1; v++) sum[v >> 2][v & 3] = fmaf(cf, __builtin_amdgcn_exp2f(old[v >> 2][v & 3] * 1e-3f), sum[v >> 2][v & 3]);
                }
                b = b1;
            }
        }
        __syncthreads();
    }
    float r = 0.0f;
#pragma unroll
    for (int m = 0; m < MB; m++)
#pragma unroll
        for (int q = 0; q < 4; q++) r += sum[m][q] + acc0[m][q] + acc1[m][q];
    out[blockIdx.x * 256 + tid] = r;
}

// returns nothing; flop = blocks * 4 waves * tiles * 20 * MB * 16384
void launch_mfma_model_test(const void *in, float *out, int mb, int blocks, int tiles, hipStream_t s)
{
    if (mb == 5) hipLaunchKernelGGL((k_mfma_model<4, 512>), dim3(blocks), dim3(512), 0, s, (const half8 *)in, out, tiles);   // ONE 8-wave workgroup per CU
    else if (mb == 9) hipLaunchKernelGGL(k_mfma_model8a, dim3(blocks), dim3(256), 0, s, (const half8 *)in, out, tiles);   // 8 row blocks, hand-placed AccVGPRs
    else if (mb == 8) hipLaunchKernelGGL(k_mfma_model<8>, dim3(blocks), dim3(256), 0, s, (const half8 *)in, out, tiles);
    else hipLaunchKernelGGL(k_mfma_model<4>, dim3(blocks), dim3(256), 0, s, (const half8 *)in, out, tiles);
}

void launch_mfma_accum_test(const void *a, const void *b, const float *c0, float *out, int trials, hipStream_t s)
{
    hipLaunchKernelGGL(k_mfma_accum, dim3(trials), dim3(64), 0, s, (const _Float16 *)a, (const _Float16 *)b, c0, out);
}

// Operand layout of v_mfma_i32_16x16x64_i8 checked with exact integer data (the CDNA guide documents the bf16 maps only): lane l
// holds A[row l&15][k = 16 (l>>4) + j] and B[k = 16 (l>>4) + j][col l&15] in byte j of its 16-byte fragment, C/D in the
// dtype-independent map col = l&15, row = 4 (l>>4) + reg.  a: [16][64] int8 row-major, b: [64][16] row-major, c: [16][16] int32.
typedef int i32x4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(64) void k_i8_layout_probe(const signed char *__restrict__ a, const signed char *__restrict__ b, int *__restrict__ c)
{
    const int l = threadIdx.x, rc = l & 15, kb = l >> 4;
    union { signed char q[16]; i32x4 v; } fa, fb;
    for (int j = 0; j < 16; j++) {
        fa.q[j] = a[rc * 64 + 16 * kb + j];
        fb.q[j] = b[(16 * kb + j) * 16 + rc];
    }
    i32x4 acc = {0, 0, 0, 0};
    acc = __builtin_amdgcn_mfma_i32_16x16x64_i8(fa.v, fb.v, acc, 0, 0, 0);
    for (int r = 0; r < 4; r++) c[(4 * kb + r) * 16 + rc] = acc[r];
}
// v_mfma_f32_16x16x32_f16 on host-chosen data, one wave per trial: a [16][32] fp16, b [32][16] fp16, c [16][16] fp32 (row-major)
// -> d [16][16] fp32; chain > 1 feeds the result back as C that many times with the same A and B.  (tools/mfma_rounding_probe.py:
// how does the matrix core round its 33-term sums?)
__global__ __launch_bounds__(64) void k_f16_mfma_probe(const _Float16 *__restrict__ a, const _Float16 *__restrict__ b, const float *__restrict__ c,
                                                        float *__restrict__ d, int chain)
{
    typedef _Float16 h8 __attribute__((ext_vector_type(8)));
    typedef float f4 __attribute__((ext_vector_type(4)));
    const int t = blockIdx.x, l = threadIdx.x, rc = l & 15, kb = l >> 4;
    a += (size_t)t * 512; b += (size_t)t * 512; c += (size_t)t * 256; d += (size_t)t * 256;
    h8 fa, fb;
    for (int j = 0; j < 8; j++) {
        fa[j] = a[rc * 32 + 8 * kb + j];
        fb[j] = b[(8 * kb + j) * 16 + rc];
    }
    f4 acc;
    for (int r = 0; r < 4; r++) acc[r] = c[(4 * kb + r) * 16 + rc];
    for (int i = 0; i < chain; i++) {
        acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa, fb, acc, 0, 0, 0);
        asm volatile("" : "+v"(acc));
    }
    for (int r = 0; r < 4; r++) d[(4 * kb + r) * 16 + rc] = acc[r];
}
void launch_f16_mfma_probe(const void *a, const void *b, const float *c, float *d, int trials, int chain, hipStream_t s)
{
    hipLaunchKernelGGL(k_f16_mfma_probe, dim3((unsigned)trials), dim3(64), 0, s, (const _Float16 *)a, (const _Float16 *)b, c, d, chain);
}

void launch_i8_layout_probe(const void *a, const void *b, int *c, hipStream_t s)
{
    hipLaunchKernelGGL(k_i8_layout_probe, dim3(1), dim3(64), 0, s, (const signed char *)a, (const signed char *)b, c);
}

// ---------------------------------------------------------------------------------------------------
// device-side checks of the decimal round-trip arithmetic (tests/test_engine_gpu.py)
// ---------------------------------------------------------------------------------------------------
__global__ void k_decq_test(const double *__restrict__ in, double *__restrict__ out, int n, int P)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = (P == 40) ? hafq::decq4_float((float)in[i]) : hafq::decq(in[i], P);   // 40: the fp32 "%.4g" entry
}
void launch_decq_test(const double *in, double *out, int n, int P, hipStream_t s)
{
    hipLaunchKernelGGL(k_decq_test, dim3((n + 255) / 256), dim3(256), 0, s, in, out, n, P);
}

__global__ void k_scale_test(const double *__restrict__ q4, const double *__restrict__ fmin, const double *__restrict__ fmax,
                             double lower, double upper, double *__restrict__ out, int n)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        const double range = fmax[i] - fmin[i];
        out[i] = hafq::scale_q6(q4[i], fmin[i], fmax[i], range, 1.0 / range, lower, upper);
    }
}
void launch_scale_test(const double *q4, const double *fmin, const double *fmax, double lower, double upper, double *out,
                       int n, hipStream_t s)
{
    hipLaunchKernelGGL(k_scale_test, dim3((n + 255) / 256), dim3(256), 0, s, q4, fmin, fmax, lower, upper, out, n);
}

}  // namespace haf
