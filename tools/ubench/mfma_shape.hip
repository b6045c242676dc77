// Micro-benchmark: sustained rate of the two fp16 MFMA shapes on random operands held in registers
// (MI355X_MICROARCH.md "DVFS give-back" item 7 reports the 16x16x32 bf16 loop ~1.15x faster than 32x32x16).
// Build: hipcc --offload-arch=gfx950 -O3 mfma_shape.hip -o mfma_shape ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int SHAPE>
__global__ __launch_bounds__(512, 2) void k(const half8 *__restrict__ in, float *out, int iters)
{
    const int tid = blockIdx.x * blockDim.x + threadIdx.x;
    half8 a[8], b[8];
    for (int i = 0; i < 8; i++) { a[i] = in[(tid * 16 + i) % 65536]; b[i] = in[(tid * 16 + 8 + i) % 65536]; }
    if (SHAPE == 32) {
        f32x16 acc0 = {0}, acc1 = {0};
        for (int it = 0; it < iters; it++) {
#pragma unroll
            for (int i = 0; i < 8; i++) {
                acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[i], b[i], acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[i], b[(i + 1) & 7], acc1, 0, 0, 0);
            }
        }
        float s = 0;
        for (int i = 0; i < 16; i++) s += acc0[i] + acc1[i];
        out[tid] = s;
    } else {
        f32x4 acc[8];
        for (int i = 0; i < 8; i++) acc[i] = f32x4{0, 0, 0, 0};
        for (int it = 0; it < iters; it++) {
#pragma unroll
            for (int i = 0; i < 8; i++) {
#pragma unroll
                for (int j = 0; j < 4; j++)   // 4 x (16x16x32) = the flops of one 32x32x16
                    acc[(i * 4 + j) & 7] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[i], b[(i + j) & 7], acc[(i * 4 + j) & 7], 0, 0, 0);
            }
        }
        float s = 0;
        for (int i = 0; i < 8; i++) for (int j = 0; j < 4; j++) s += acc[i][j];
        out[tid] = s;
    }
}

int main()
{
    std::vector<_Float16> h(65536 * 8);
    srand(1);
    for (auto &v : h) v = (_Float16)((rand() / (float)RAND_MAX) * 2 - 1);
    half8 *d; float *o;
    hipMalloc(&d, h.size() * 2); hipMalloc(&o, 256 * 2 * 512 * 4);
    hipMemcpy(d, h.data(), h.size() * 2, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 20000;
    for (int rep = 0; rep < 3; rep++)
        for (int shape : {32, 16}) {
            hipEventRecord(e0);
            if (shape == 32) hipLaunchKernelGGL(k<32>, dim3(256), dim3(512), 0, 0, d, o, iters);
            else hipLaunchKernelGGL(k<16>, dim3(256), dim3(512), 0, 0, d, o, iters);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            double flop = 256.0 * 8 /*waves*/ * iters * 16 /*mfma 32x32x16 equivalents*/ * 32768.0;
            printf("shape %dx%d: %.2f ms  %.0f TFLOP/s\n", shape, shape, ms, flop / ms / 1e9);
        }
    return 0;
}
