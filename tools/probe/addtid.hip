// probe: address semantics of ds_read_addtid_b32 on gfx950 (M0 base + offset + lane*4 ?)
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(float *out, int off) {
    __shared__ float buf[2048];
    for (int i = threadIdx.x; i < 2048; i += blockDim.x) buf[i] = (float)i;
    __syncthreads();
    float v;
    unsigned base = (unsigned)(size_t)buf + (unsigned)off * 4u;   // wave-uniform
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tds_read_addtid_b32 %0 offset:8\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "s"(base) : "m0", "memory");
    out[threadIdx.x] = v;
}
int main() {
    float *d; hipMalloc(&d, 256 * 4);
    hipLaunchKernelGGL(k, dim3(1), dim3(256), 0, 0, d, 100);
    float h[256]; hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    printf("off=100 offset:8 -> t0 %g t1 %g t63 %g t64 %g t65 %g t255 %g\n", h[0], h[1], h[63], h[64], h[65], h[255]);
    return 0;
}
