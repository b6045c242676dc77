// contraction.hip -- the RBF decision as MFMA contractions (libsvm svm_predict_values / Kernel::k_function RBF,
// svm.cpp:325-365, 2478-2532); the single-pass screening kernel k_svm_screen is in screen.hip:
//   k_svm_rbf_h      three fp16 MFMA passes on the hi/lo halves of the fp32 operands, exp (or centred-remainder polynomial) epilogue,
//                    guard band (tier 1: on the list in the default mode, on everything with HAF_FLAG_SPLIT_F16)
//   k_svm_rbf        one fp32 MFMA pass (HAF_FLAG_FP32_MFMA)
//   k_svm_h_combine / k_svm_h_combine_cr   partial sums of the SV ranges of a list launch -> decision + band
//
// Built with -ffp-contract=off: every fp32/fp64 expression that must match the CPU restatement bit for bit is
// written with explicit *_rn intrinsics as well; fma() is used only where a fused operation is intended.
#include "device_common.h"
#ifndef HAF_ABL
#define HAF_ABL 0     // timing experiments on k_svm_rbf_h<true> (tools/ablate_h.sh): never defined in a build that is kept
#endif

namespace haf {

// ---------------------------------------------------------------------------------------------------
// a8: RBF decision as an fp32 MFMA contraction.
//   dec(e) = sum_n coef_n * exp(-gamma * |x_e - s_n|^2) - rho,   |x-s|^2 = |x|^2 + |s|^2 - 2 x.s
// Workgroup = 8 waves = 256 evals.  Each wave keeps its 32 evals x 324 attributes in 162 VGPRs (the A operand of
// v_mfma_f32_32x32x2_f32, loaded once) and sweeps every 32-SV tile: the tile image [328][32] (324 attribute rows,
// one row of -g2*|s|^2, one row of coefficients) is streamed global -> LDS by LDS-DMA (global_load_lds_dwordx4),
// double buffered, and read back as the B operand with conflict-free 256-byte ds_read_b32.  The 32x32 fp32
// accumulator goes straight through exp2 and the coefficient FMA in registers; only 4 bytes per eval leave the CU.
// Two waves per SIMD: one wave's exp/FMA epilogue hides under the other's MFMAs.
// ---------------------------------------------------------------------------------------------------
// The DMA is issued from inline asm on purpose: hipcc tracks a builtin LDS-DMA like an ordinary load and parks an
// s_waitcnt vmcnt(0) in front of the first ds_read of the tile being computed, which serialises load and compute.
// Hidden from its scoreboard, the pieces of tile t+1 stay in flight under the 162 MFMAs of tile t; the explicit
// s_waitcnt vmcnt(0) + barrier at the end of the iteration is the only wait (cdna_hip_programming.md §5.7).
__device__ __forceinline__ void stage_sv_tile(const float *__restrict__ gtile, unsigned lds_byte_off, int wave, int lane)
{
    // 41 KiB = 41 wave-instructions of 1 KiB; LDS destination = M0 (wave-uniform) + lane*16
    for (int p = wave; p < kTileFloats / 256; p += 8) {
        const char *g = reinterpret_cast<const char *>(gtile) + p * 1024 + lane * 16;
        unsigned l = __builtin_amdgcn_readfirstlane(lds_byte_off + p * 1024);
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" ::"s"(l), "v"(g) : "memory", "m0");
    }
}

__global__ __launch_bounds__(kSvmThreads, 2) void k_svm_rbf(const float *__restrict__ X, const float *__restrict__ ax,
                                                            const float *__restrict__ svt,
                                                            const int *__restrict__ evalcell,
                                                            const int *__restrict__ counters, SvmParams p,
                                                            float *__restrict__ dec, int8_t *__restrict__ labels,
                                                            int *__restrict__ flag_list, int flag_cap,
                                                            int *__restrict__ counters_rw, Dims d, unsigned char *__restrict__ t1flags)
{
    __shared__ __attribute__((aligned(16))) float lds[2 * kTileFloats];   // the ONLY LDS object: two SV tile images
    const int n_evals = counters[CNT_EVALS];
    const long base = (long)blockIdx.x * kSvmBlockEvals;
    if (base >= n_evals) return;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const long tile32 = (base >> 5) + wave;

    const unsigned lds0 = (unsigned)(uintptr_t)lds;                  // LDS byte address of buffer 0
    stage_sv_tile(svt, lds0, wave, lane);                           // tile 0 in flight while A loads

    float a[kKSteps];
    {
        const float *xt = X + (size_t)tile32 * kTileFloats + lane;
#pragma unroll
        for (int s = 0; s < kKSteps; s++) a[s] = xt[s * 64];        // A[i = lane&31][k = 2s + (lane>>5)]
    }
    float axr[16], part[16], pabs[16];
#pragma unroll
    for (int r = 0; r < 16; r++) {
        int row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);          // C/D row of register r (32x32 layout)
        axr[r] = ax[tile32 * kTile + row];
        part[r] = 0.0f;
        pabs[r] = 0.0f;
    }
    // Pin every compiler-issued load BEFORE the main loop: the loop's LDS-DMA is invisible to hipcc's vmcnt
    // bookkeeping, so one of its counted waits for a still-pending A/ax load would come up short once younger DMA
    // pieces sit behind it in the queue.  An empty asm that consumes each register makes the compiler finish them here.
#pragma unroll
    for (int s = 0; s < kKSteps; s++) asm volatile("" : "+v"(a[s]));
#pragma unroll
    for (int r = 0; r < 16; r++) asm volatile("" : "+v"(axr[r]));
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");               // this wave's DMA pieces of tile 0 have landed
    __syncthreads();

    const int nt = d.n_sv_tiles;
    for (int t = 0; t < nt; t++) {
        float *cur = lds + (t & 1) * kTileFloats;
        if (t + 1 < nt)
            stage_sv_tile(svt + (size_t)(t + 1) * kTileFloats, lds0 + ((t + 1) & 1) * kTileFloats * 4, wave, lane);

        f32x16 acc = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        const float *bl = cur + lane;
#pragma unroll
        for (int s = 0; s < kKSteps; s++) {
            float b = bl[s * 64];                                   // B[k = 2s + (lane>>5)][j = lane&31]
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s], b, acc, 0, 0, 0);
        }
        const float as_ = cur[kKP * kTile + (lane & 31)];           // -g2*|s_j|^2
        const float cf = cur[(kKP + 1) * kTile + (lane & 31)];      // coef_j (0 for padding SVs)
        // HAZARD (measured on gfx950, two waves per SIMD; screen.hip has the details): a VALU instruction that reads a
        // v_exp_f32 result within a few instructions of the v_exp_f32 can read the register before it is written.  All
        // sixteen exps are issued first and pinned there; their consumers follow.
#pragma unroll
        for (int r = 0; r < 16; r++)
            acc[r] = __builtin_amdgcn_exp2f(fmaf(p.two_gamma2, acc[r], axr[r] + as_));   // -g2*(|x|^2 + |s|^2 - 2 x.s)
        // the wait states hang on the data: the asm reads and "writes" all sixteen results, so every exp is in front of it
        // and every consumer behind it whatever the scheduler does (a free-standing s_nop asm was moved to the end of the
        // tile by hipcc in one build; haf_grasping_amd/build.py now checks the distance in the ISA)
        asm volatile("s_nop 7\n\ts_nop 7" : "+v"(acc));             // the last exp gets 16 wait states before any consumer
#pragma unroll
        for (int r = 0; r < 16; r++) {
            part[r] = fmaf(cf, acc[r], part[r]);
            pabs[r] = fmaf(fabsf(cf), acc[r], pabs[r]);              // sum |coef| K: scale of the rounding error
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // this wave's DMA pieces of tile t+1 have landed
        __syncthreads();                                            // ... and everybody is done reading tile t
    }

    // sum the 32 SV columns held by the 32 lanes of each half
#pragma unroll
    for (int r = 0; r < 16; r++) {
        float v = part[r];
        v += __shfl_xor(v, 16, 64);
        v += __shfl_xor(v, 8, 64);
        v += __shfl_xor(v, 4, 64);
        v += __shfl_xor(v, 2, 64);
        v += __shfl_xor(v, 1, 64);
        part[r] = v;
        float w = pabs[r];
        w += __shfl_xor(w, 16, 64);
        w += __shfl_xor(w, 8, 64);
        w += __shfl_xor(w, 4, 64);
        w += __shfl_xor(w, 2, 64);
        w += __shfl_xor(w, 1, 64);
        pabs[r] = w;
    }
    if ((lane & 31) == 0) {
#pragma unroll
        for (int r = 0; r < 16; r++) {
            int row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            long e = tile32 * kTile + row;
            if (e < n_evals) {
                float dv = part[r] - p.rho;
                dec[e] = dv;
                labels[evalcell[e]] = (int8_t)(dv > 0.0f ? p.gv0 : p.gv1);      // svm.cpp:2516-2531
                // guard band: the fp32 error of the sum is at most (guard_acc + guard_dot*(|a_x| + max|a_s|)) * sum|coef|K
                // (DESIGN.md §2); inside it the evaluation goes to the fp64 tiers.  Also catches NaN.
                const bool undecided = !(fabsf(dv) > (p.guard_acc + p.guard_dot * (p.as_max + fabsf(axr[r]))) * pabs[r] + p.guard_abs);
                if (t1flags) t1flags[e] = undecided ? 1 : 0;          // ordered hand-over (k_t1_handover, below)
                else if (undecided) {
                    int slot = atomicAdd(&counters_rw[CNT_FLAGGED], 1);
                    if (slot < flag_cap) flag_list[slot] = (int)e;
                }
            }
        }
    }
}

// Ordered hand-over of tier 1 (round 5): the contraction kernels leave one byte per entry of their iteration space -- "undecided" -- and
// two launches append those entries' evaluations to the exact tiers' list in that order: k_t1_count (a workgroup per 4096 entries:
// 16 bytes per thread) and k_t1_compact (the workgroups in front of mine summed, an ordered scan inside, the total published by the
// last one).  The list's counter is zero when tier 1 starts (a request's counters are zeroed; a screening pass that writes the tier
// list itself means tier 1 does not run), so the total IS the counter.  The grid is sized for the capacity; workgroups beyond the live
// count leave at once.  (A first version did this with ONE workgroup: 5.6 ms for the 7.9 M entries of the all-evaluations modes.)
constexpr int kT1Chunk = 4096;
__device__ __forceinline__ int t1_flags16(const unsigned char *flags, long b, long n)
{
    if (b + 16 <= n) {
        const uint4 v = *reinterpret_cast<const uint4 *>(flags + b);       // (flags are 0 / 1; b is a multiple of 16)
        return __popc(v.x) + __popc(v.y) + __popc(v.z) + __popc(v.w);
    }
    int c = 0;
    for (long q = b; q < n; q++) c += flags[q] ? 1 : 0;
    return c;
}
__global__ __launch_bounds__(256) void k_t1_count(const unsigned char *__restrict__ flags, const int *__restrict__ counters, int count_slot,
                                                  int in_cap, int *__restrict__ blkcount)
{
    __shared__ int red[4];
    const long n = min(counters[count_slot], in_cap);
    const long b = (long)blockIdx.x * kT1Chunk + threadIdx.x * 16;
    int c = (b < n) ? t1_flags16(flags, b, n) : 0;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) blkcount[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}
__global__ __launch_bounds__(256) void k_t1_compact(const unsigned char *__restrict__ flags, const int *__restrict__ counters, int count_slot,
                                                    int in_cap, const int *__restrict__ blkcount, const int *__restrict__ idx_list,
                                                    int *__restrict__ flag_list, int flag_cap, int *__restrict__ counters_rw)
{
    __shared__ int part[256];
    __shared__ int s_base;
    const long n = min(counters[count_slot], in_cap);
    const int n_blk = (int)((n + kT1Chunk - 1) / kT1Chunk);
    if ((int)blockIdx.x >= n_blk && blockIdx.x != 0) return;
    const int t = threadIdx.x;
    int before = 0, total = 0;
    const bool last = (int)blockIdx.x == max(n_blk - 1, 0);
    const int upto = last ? n_blk : (int)blockIdx.x;
    for (int j = t; j < upto; j += 256) {
        const int c = blkcount[j];
        total += c;
        if (j < (int)blockIdx.x) before += c;
    }
    part[t] = before;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (t < o) part[t] += part[t + o];
        __syncthreads();
    }
    if (t == 0) s_base = part[0];
    __syncthreads();
    if (last) {
        part[t] = total;
        __syncthreads();
        for (int o = 128; o > 0; o >>= 1) {
            if (t < o) part[t] += part[t + o];
            __syncthreads();
        }
        if (t == 0) counters_rw[CNT_FLAGGED] = part[0] + counters_rw[CNT_T1_ADD];   // (+ what the short-list gate put in front: kernels.h)
        __syncthreads();
    }
    const long b = (long)blockIdx.x * kT1Chunk + t * 16;
    const int mine = (b < n) ? t1_flags16(flags, b, n) : 0;
    part[t] = mine;
    __syncthreads();
    for (int o = 1; o < 256; o <<= 1) {
        const int v = (t >= o) ? part[t - o] : 0;
        __syncthreads();
        part[t] += v;
        __syncthreads();
    }
    int slot = s_base + part[t] - mine;
    if (mine) {
        const long e1 = min(b + 16, n);
        for (long q = b; q < e1; q++)
            if (flags[q]) {
                if (slot < flag_cap) flag_list[slot] = idx_list ? idx_list[q] : (int)q;
                slot++;
            }
    }
}
// Lists of at most 65 536 entries (a small request's: the reference's own grids) in ONE workgroup, 64 flags per thread: the two launches
// above are 5 + 8 us of a C3 request against a big model.  Same list.
__global__ __launch_bounds__(kListCompactThreads) void k_t1_handover_small(const unsigned char *__restrict__ flags, const int *__restrict__ counters,
                                                                           int count_slot, int in_cap, const int *__restrict__ idx_list,
                                                                           int *__restrict__ flag_list, int flag_cap, int *__restrict__ counters_rw)
{
    __shared__ int s_wave[kListCompactThreads / 64];
    const int n = min(min(counters[count_slot], in_cap), kListCompactThreads * 64);
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int b = t * 64;
    // the thread's 64 flags in sixteen registers (the buffer is a whole number of 64-byte blocks; bytes from n on are stale: masked)
    unsigned w[16];
#pragma unroll
    for (int q = 0; q < 4; q++) {
        uint4 v = uint4{0u, 0u, 0u, 0u};
        if (b + 16 * q < n) v = *reinterpret_cast<const uint4 *>(flags + b + 16 * q);
        w[4 * q] = v.x; w[4 * q + 1] = v.y; w[4 * q + 2] = v.z; w[4 * q + 3] = v.w;
    }
    int mine = 0;
#pragma unroll
    for (int k = 0; k < 16; k++) {
        const int keep = n - (b + 4 * k);                                    // entries of this word below n
        if (keep < 4) w[k] = keep <= 0 ? 0u : (w[k] & ((1u << (8 * keep)) - 1u));
        mine += __popc(w[k]);                                                // (flags are 0 / 1)
    }
    int incl = mine;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int v = __shfl_up(incl, o, 64);
        if (lane >= o) incl += v;
    }
    if (lane == 63) s_wave[wave] = incl;
    __syncthreads();
    int before = 0, total = 0;
#pragma unroll
    for (int k = 0; k < kListCompactThreads / 64; k++) {
        const int c = s_wave[k];
        if (k < wave) before += c;
        total += c;
    }
    int slot = before + incl - mine;
    if (mine) {
#pragma unroll
        for (int k = 0; k < 16; k++) {
            unsigned m = w[k];
            while (m) {
                const int j = (__ffs((int)m) - 1) >> 3;
                m &= ~(0xffu << (8 * j));
                const int q = b + 4 * k + j;
                if (slot < flag_cap) flag_list[slot] = idx_list ? idx_list[q] : q;
                slot++;
            }
        }
    }
    if (t == 0) counters_rw[CNT_FLAGGED] = total + counters_rw[CNT_T1_ADD];
}
// the flag buffer's layout (engine_tables.cpp sizes it with t1_flag_bytes): [max_entries bytes, rounded up to 64][one int per 4096 entries]
static int *t1_blkcount(unsigned char *flags, long max_entries) { return reinterpret_cast<int *>(flags + (((size_t)max_entries + 63) / 64) * 64); }
static void launch_t1_handover(const unsigned char *flags, const int *counters, int count_slot, int in_cap, const int *idx_list, int *flag_list,
                               int flag_cap, int *counters_rw, hipStream_t s, long max_entries, int *blkcount)
{
    const int n_blk = (int)((max_entries + kT1Chunk - 1) / kT1Chunk);
    if (n_blk <= 0) return;
    if (max_entries <= (long)kListCompactThreads * 64) {
        hipLaunchKernelGGL(k_t1_handover_small, dim3(1), dim3(kListCompactThreads), 0, s, flags, counters, count_slot, in_cap, idx_list, flag_list,
                           flag_cap, counters_rw);
        return;
    }
    hipLaunchKernelGGL(k_t1_count, dim3(n_blk), dim3(256), 0, s, flags, counters, count_slot, in_cap, blkcount);
    hipLaunchKernelGGL(k_t1_compact, dim3(n_blk), dim3(256), 0, s, flags, counters, count_slot, in_cap, blkcount, idx_list, flag_list, flag_cap, counters_rw);
}

__global__ __launch_bounds__(256) void k_short_list_gate(int *__restrict__ counters, int src_slot, const int *__restrict__ src_list, int src_cap,
                                                         int *__restrict__ dst_list, int dst_slot, int dst_direct, int max_entries)
{
    const int n = counters[src_slot];
    const bool take = n > 0 && n <= max_entries && n <= src_cap;
    if (take)
        for (int i = threadIdx.x; i < n; i += 256) dst_list[i] = src_list[i];
    __syncthreads();                                       // (every thread has read the source counter)
    if (threadIdx.x == 0) {
        counters[CNT_T1_N] = take ? 0 : n;
        if (take) {
            counters[CNT_BYPASS] = n;
            counters[dst_direct ? dst_slot : CNT_T1_ADD] = n;
        }
    }
}

void launch_short_list_gate(int *counters, int src_slot, const int *src_list, int src_cap, int *dst_list, int dst_slot, bool dst_direct,
                            int max_entries, hipStream_t s)
{
    hipLaunchKernelGGL(k_short_list_gate, dim3(1), dim3(256), 0, s, counters, src_slot, src_list, src_cap, dst_list, dst_slot, dst_direct ? 1 : 0,
                       max_entries);
}

void launch_svm(const float *X, const float *ax, const float *svt, const int *evalcell, const int *counters, SvmParams p,
                float *dec, int8_t *labels, int *flag_list, int flag_cap, int *counters_rw, Dims d, long max_evals,
                hipStream_t s, unsigned char *t1flags)
{
    long blocks = (max_evals + kSvmBlockEvals - 1) / kSvmBlockEvals;
    if (blocks <= 0) return;
    hipLaunchKernelGGL(k_svm_rbf, dim3((unsigned)blocks), dim3(kSvmThreads), 0, s, X, ax, svt, evalcell, counters, p,
                       dec, labels, flag_list, flag_cap, counters_rw, d, t1flags);
    if (t1flags) launch_t1_handover(t1flags, counters, CNT_EVALS, 0x7fffffff, nullptr, flag_list, flag_cap, counters_rw, s, max_evals, t1_blkcount(t1flags, max_evals));
}

// ---------------------------------------------------------------------------------------------------
// a8, split-fp16 form of the same contraction: x = xh + xl, s = sh + sl with fp16 halves (22 significant bits, i.e. the
// fp32 operand to within one ulp), x.s = xh.sh + xl.sh + xh.sl as three fp16 MFMA passes into ONE fp32 accumulator (the
// dropped xl.sl term is 2^-22 relative).  Every fp16 x fp16 product is exact in fp32, so the error budget is the fp32
// kernel's (accumulation) plus 2^-22 per term, covered by the same guard band; the MFMA work per 32x32 output tile
// drops from 162 x 64 to 2016 cycles.  MFMA shape: v_mfma_f32_16x16x32_f16 (10 k-steps) + v_mfma_f32_16x16x16_f16
// (K tail), 2x2 sub-tiles per wave: same cycles per FLOP as 32x32x16 but the chip holds a ~15 % higher clock on it.
// Same structure as k_svm_rbf: 8 waves x 32 evals, A fragments (hi and lo: 168 VGPRs) loaded once, SV tile images
// streamed by LDS-DMA -- here through a 3-deep LDS ring with a counted vmcnt, because a tile is consumed in ~4k cycles,
// about the latency of one DMA round trip.
// ---------------------------------------------------------------------------------------------------
typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ int stage_sv_tile_h(const char *__restrict__ gtile, unsigned lds_byte_off, int wave, int lane)
{
    int issued = 0;
    for (int p = wave; p < kHSvPieces; p += 8) {
        const char *g = gtile + p * 1024 + lane * 16;
        unsigned l = __builtin_amdgcn_readfirstlane(lds_byte_off + p * 1024);
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" ::"s"(l), "v"(g) : "memory", "m0");
        issued++;
    }
    return issued;
}

// four fp32 additions as two packed instructions (same IEEE results; the element-wise loop compiles to four v_add_f32)
__device__ __forceinline__ f32x4 h_add4(f32x4 a, f32x4 b)
{
    typedef float f2 __attribute__((ext_vector_type(2)));
    const f2 lo = f2{a[0], a[1]} + f2{b[0], b[1]}, hi = f2{a[2], a[3]} + f2{b[2], b[3]};
    return f32x4{lo[0], lo[1], hi[0], hi[1]};
}
__device__ __forceinline__ f32x4 h_add4s(f32x4 a, float s) { return h_add4(a, f32x4{s, s, s, s}); }
__device__ __forceinline__ f32x4 h_fma4s(float s, f32x4 v, f32x4 c)        // fma(s, v[i], c[i]): two v_pk_fma_f32
{
    typedef float f2 __attribute__((ext_vector_type(2)));
    const f2 ss = {s, s};
    const f2 lo = __builtin_elementwise_fma(ss, f2{v[0], v[1]}, f2{c[0], c[1]}), hi = __builtin_elementwise_fma(ss, f2{v[2], v[3]}, f2{c[2], c[3]});
    return f32x4{lo[0], lo[1], hi[0], hi[1]};
}

// PRECISE (the list mode behind the screening pass, where speed does not matter): the dominant pass xh.sh goes ONE k-step
// at a time into a fresh accumulator that is added to the running sum by the VALU, and the two small passes form their
// own MFMA chain first.  Whatever order the matrix core adds the 32 products of an instruction in, the error is then at
// most 31 u T_s per instruction (T_s = the step's sum of |products|) + one rounding per VALU add: 43 u sum|x_i s_i| in all
// instead of one rounding per product of a 3 x 336-term chain, and the guard band shrinks with it (guard_dot_p).
// CRP (PRECISE list mode only; round 4): the centred-remainder form of tier 1 behind SCREEN_CR_POLY.  X and the SV tiles hold x - m
// and s - m, the tile tail holds b_n = c_n 2^(-gamma'|s_n - m|^2) where the plain form has the coefficient (and 0 where it has
// -gamma'|s|^2), the epilogue accumulates b psi(z), z = 2 gamma' (x - m).(s - m), psi(z) = z^2 (a2 + a3 z + a4 z^2 + a5 z^3) -- no
// transcendental, relative accuracy -- and k_svm_h_combine_cr forms dec = 2^(a_x) (B0 + L + P + N) - rho with its band.
constexpr float kPsiA2h = 0.240226506959101f, kPsiA3h = 0.0555041086648216f, kPsiA4h = 0.00961812910762848f, kPsiA5h = 0.00133335581464284f;
// List mode: the number of SV tile ranges a list of n evaluations is cut into.  gridDim.y = kHListParts for the long lists of a
// bench-sized request; a request that cannot fill the chip anyway (the host launches gridDim.y = kHListPartsShort then) takes as many
// ranges as put ~512 workgroups on the chip -- from the LIVE length, on the device.  The part buffer holds 2 x kHListParts x part_stride
// sums: P > kHListParts ranges use the stride part_stride kHListParts / P, for lists that stride holds.
// (Round 5 tried 32 ranges and one workgroup per CU -- 256 / blocks -- for C3 against the 8 964-SV model, 1 964 entries: 78 us against
// 69 with 16 ranges.  A workgroup's prologue -- 336 KB of hi / lo operand fragments -- is paid per range and costs more than nine tiles.)
constexpr int kHListPartsShort = 16;
__device__ __forceinline__ int h_list_parts(int n_evals, int max_parts, long part_stride)
{
    if (max_parts <= kHListParts) return max_parts;
    const int blocks = (n_evals + kSvmBlockEvals - 1) / kSvmBlockEvals;
    int want = min(max_parts, max(kHListParts, blocks > 0 ? 512 / blocks : max_parts));
    while (want > kHListParts && (long)n_evals * want > part_stride * kHListParts) want--;
    return want;
}
__device__ __forceinline__ long h_list_stride(long part_stride, int parts) { return parts > kHListParts ? part_stride * kHListParts / parts : part_stride; }

template <bool PRECISE, bool CRP = false>
__global__ __launch_bounds__(kSvmThreads, 2) void k_svm_rbf_h(const char *__restrict__ X, const float *__restrict__ ax,
                                                              const char *__restrict__ svt,
                                                              const int *__restrict__ evalcell,
                                                              const int *__restrict__ counters, SvmParams p,
                                                              float *__restrict__ dec, int8_t *__restrict__ labels,
                                                              int *__restrict__ flag_list, int flag_cap,
                                                              int *__restrict__ counters_rw, Dims d,
                                                              const int *__restrict__ idx_list, int list_counter, int list_cap,
                                                              double *__restrict__ part_out, long part_stride, unsigned char *__restrict__ t1flags)
{
    // the ONLY LDS object: 3 SV tile images + per wave one row of a_x (fp32) and one row of positive-group sums (fp64)
    __shared__ __attribute__((aligned(16))) char lds[kHBuffers * kHSvTileBytes + 3 * 8 * kTile * 4];
    // list mode (behind the screening pass): slot j of X / ax holds evaluation idx_list[j]
    const int n_evals = idx_list ? min(counters[list_counter], list_cap) : counters[CNT_EVALS];
    const long base = (long)blockIdx.x * kSvmBlockEvals;
    if (base >= n_evals) return;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const long tile32 = (base >> 5) + wave;
    const unsigned lds0 = (unsigned)(uintptr_t)lds;
    // part_out (list mode): the SV tiles are cut into gridDim.y ranges and workgroup (x, y) sums range y only; its two class
    // sums go to part_out and k_svm_h_combine finishes the evaluation.  A list of 150 k evaluations is 589 workgroups for
    // 512 slots: whole sweeps would run as two rounds with the second one 15 % full, quarter sweeps pack the slots.
    const int n_parts = part_out ? h_list_parts(n_evals, (int)gridDim.y, part_stride) : 1;
    if ((int)blockIdx.y >= n_parts) return;
    part_stride = h_list_stride(part_stride, n_parts);
    const int t0 = part_out ? (int)((long)d.n_sv_tiles * blockIdx.y / n_parts) : 0;
    const int nt = part_out ? (int)((long)d.n_sv_tiles * (blockIdx.y + 1) / n_parts) : d.n_sv_tiles;
    float *axs = reinterpret_cast<float *>(lds + kHBuffers * kHSvTileBytes) + wave * kTile;
    double *pos = reinterpret_cast<double *>(lds + kHBuffers * kHSvTileBytes + 8 * kTile * 4) + wave * kTile;

    if (t0 < nt) stage_sv_tile_h(svt + (size_t)t0 * kHSvTileBytes, lds0, wave, lane);                          // first tile
    if (t0 + 1 < nt) stage_sv_tile_h(svt + (size_t)(t0 + 1) * kHSvTileBytes, lds0 + kHSvTileBytes, wave, lane);   // second

    // A fragments: [k-step][row block m][hi|lo]; lane holds A[row 16m + (lane&15)][k = 32s + 8(lane>>4) + j]
    half8 ah[kHFull][2], al[kHFull][2];
    half4 aht[2], alt[2];                                            // K tail: A[row][k = 320 + 4(lane>>4) + j]
    {
        const char *xt = X + (size_t)tile32 * kHXTileBytes;
#pragma unroll
        for (int s = 0; s < kHFull; s++)
#pragma unroll
            for (int m = 0; m < 2; m++) {
                ah[s][m] = *reinterpret_cast<const half8 *>(xt + (s * 2 + m) * 1024 + lane * 16);
                al[s][m] = *reinterpret_cast<const half8 *>(xt + kHMatBytes + (s * 2 + m) * 1024 + lane * 16);
            }
#pragma unroll
        for (int m = 0; m < 2; m++) {
            aht[m] = *reinterpret_cast<const half4 *>(xt + kHTailOff + m * 512 + lane * 8);
            alt[m] = *reinterpret_cast<const half4 *>(xt + kHMatBytes + kHTailOff + m * 512 + lane * 8);
        }
    }
    if (lane < kTile) { axs[lane] = ax[tile32 * kTile + lane]; pos[lane] = 0.0; }
    // rows 16m + 4(lane>>4) + r, summed over this lane's columns.  Two levels: `lo` (fp32) takes the products of kFold tiles -- a
    // chain of 2 kFold fmas -- and is then added to `part`.  PRECISE: kFold = 1 (16 conversions and adds next to ~200 vector
    // instructions of the tile) and `part` is fp64 like everything behind it (lane reduction, class sums, the ranges of the list
    // mode): a term of the coefficient sum passes through two fp32 roundings and no more.  Bulk form: kFold = 8, fp32 throughout
    // (its registers are spoken for), so the error grows with tiles/8 + 8 instead of tiles.
    constexpr int kFold = PRECISE ? 1 : 8;
    typedef typename std::conditional<PRECISE, double, float>::type part_t;
    part_t part[2][4];
    float lo[2][4];
#pragma unroll
    for (int m = 0; m < 2; m++)
#pragma unroll
        for (int r = 0; r < 4; r++) { part[m][r] = (part_t)0; lo[m][r] = 0.0f; }
    // pin the compiler-issued loads before any further (asm, uncounted) DMA is queued behind them (see k_svm_rbf)
#pragma unroll
    for (int s = 0; s < kHFull; s++)
#pragma unroll
        for (int m = 0; m < 2; m++) {
            asm volatile("" : "+v"(ah[s][m]));
            asm volatile("" : "+v"(al[s][m]));
        }
#pragma unroll
    for (int m = 0; m < 2; m++) {
        asm volatile("" : "+v"(aht[m]));
        asm volatile("" : "+v"(alt[m]));
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                // tiles 0 and 1 (this wave's pieces) have landed
    __syncthreads();

    const int my_pieces = (kHSvPieces - wave + 7) / 8;              // DMA instructions this wave issues per tile (6 or 5)
    float axr[2][4];                                                // a_x of this lane's 8 rows (LDS reads cannot be hoisted
    if (!PRECISE) {                                                 //  over the asm DMA by the compiler, so do it by hand)
#pragma unroll
        for (int m = 0; m < 2; m++)
#pragma unroll
            for (int r = 0; r < 4; r++) axr[m][r] = axs[16 * m + 4 * (lane >> 4) + r];
    }
    for (int t = t0; t < nt; t++) {
        const char *cur = lds + ((t - t0) % kHBuffers) * kHSvTileBytes;
        const bool more = t + 2 < nt;
        if (more && !(PRECISE && HAF_ABL == 4))
            stage_sv_tile_h(svt + (size_t)(t + 2) * kHSvTileBytes, lds0 + ((t - t0 + 2) % kHBuffers) * kHSvTileBytes, wave, lane);
        if (t == d.sv_tile_neg) {
            // The tile images hold the non-negative coefficients first: what has been summed so far is
            // P = sum_{coef>0} coef*K, what follows is N = sum_{coef<0} coef*K.  dec = P + N - rho and the guard scale
            // sum|coef|K = P - N come from the same accumulator; P is parked in LDS (once per workgroup).
#pragma unroll
            for (int m = 0; m < 2; m++)
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    part_t v = part[m][r] + (part_t)lo[m][r];
                    v += __shfl_xor(v, 8, 64);
                    v += __shfl_xor(v, 4, 64);
                    v += __shfl_xor(v, 2, 64);
                    v += __shfl_xor(v, 1, 64);
                    if ((lane & 15) == 0) pos[16 * m + 4 * (lane >> 4) + r] = v;
                    part[m][r] = (part_t)0;
                    lo[m][r] = 0.0f;
                }
        }

        // 2x2 sub-tiles of 16x16; B fragments run one k-step (12 MFMAs) ahead of the MFMAs that consume them
        f32x4 acc[2][2];
#pragma unroll
        for (int m = 0; m < 2; m++)
#pragma unroll
            for (int n = 0; n < 2; n++) acc[m][n] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
        const char *bl = cur + lane * 16;
        if (PRECISE) {
            // The B fragments of the whole tile as ONE sequence of 60 reads -- hi image (sweep 1a: xl.sh), lo image (sweep 1b: xh.sl),
            // hi image again (sweep 2: xh.sh) -- through a ring of three registers, each read TWO steps (four MFMAs of this wave,
            // and as many of the SIMD's other wave) ahead of its use.  One step ahead (round 2) left every step waiting for its
            // fragment: at two waves per SIMD the LDS round trip is longer than the other wave's two MFMAs, the waves spent 43 % of
            // their cycles in s_waitcnt and the matrix pipe was busy half the time (profiles/README.md, round 3).
            half8 fb[3];
#define HAF_H_FRAG(g) (((g) >= 20 && (g) < 40) ? bl + kHMatBytes + ((g) - 20) * 1024 : bl + ((g) % 20) * 1024)
            fb[0] = *reinterpret_cast<const half8 *>(HAF_H_FRAG(0));
            fb[1] = *reinterpret_cast<const half8 *>(HAF_H_FRAG(1));
            // sweep 1: xl.sh, then xh.sl, magnitudes 2^-11 of the main pass: a plain MFMA chain (its roundings are negligible);
            // consecutive steps alternate the column block, so an accumulator is needed again only four MFMAs later.
            half4 bht[2], bqt[2];
#pragma unroll
            for (int g = (HAF_ABL == 3 ? 38 : 0); g < 40; g++) {
                const int sstep = (g % 20) >> 1, n = g & 1;
                fb[(g + 2) % 3] = *reinterpret_cast<const half8 *>(HAF_H_FRAG(g + 2));     // (g = 38, 39: the first two of sweep 2)
                if (g == 36) {
#pragma unroll
                    for (int nn = 0; nn < 2; nn++) bht[nn] = *reinterpret_cast<const half4 *>(cur + kHTailOff + nn * 512 + lane * 8);
                }
                if (g == 37) {
#pragma unroll
                    for (int nn = 0; nn < 2; nn++) bqt[nn] = *reinterpret_cast<const half4 *>(cur + kHMatBytes + kHTailOff + nn * 512 + lane * 8);
                }
                const half8 b = fb[g % 3];
#pragma unroll
                for (int m = 0; m < 2; m++)
                    acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(g < 20 ? al[sstep][m] : ah[sstep][m], b, acc[m][n], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);                   // one step's fragments live at a time: no spills
            }
#pragma unroll
            for (int n = 0; n < 2; n++) {
#pragma unroll
                for (int m = 0; m < 2; m++) acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x16f16(alt[m], bht[n], acc[m][n], 0, 0, 0);
#pragma unroll
                for (int m = 0; m < 2; m++) acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x16f16(aht[m], bqt[n], acc[m][n], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
            // sweep 2: xh.sh, each k-step into a fresh accumulator, summed by the VALU -- one step behind: the adds of a step
            // are issued after the MFMAs of the next one, so the matrix pipe does not idle under the result latency
            const f32x4 zero = {0.0f, 0.0f, 0.0f, 0.0f};
            f32x4 tp[2] = {zero, zero};
#pragma unroll
            for (int g = 40; g < 60; g++) {
                const int sstep = (g - 40) >> 1, n = g & 1;
                if (g + 2 < 60) fb[(g + 2) % 3] = *reinterpret_cast<const half8 *>(HAF_H_FRAG(g + 2));
                if (g == 57) {
#pragma unroll
                    for (int nn = 0; nn < 2; nn++) bht[nn] = *reinterpret_cast<const half4 *>(cur + kHTailOff + nn * 512 + lane * 8);
                }
                const half8 bhv = fb[g % 3];
                f32x4 t4[2];
                const int pn = n ^ 1;                                  // the previous step had the other column block
                // MFMA, the adds of the previous step's FIRST result in its shadow, MFMA, the adds of the second: pinned, because
                // left alone hipcc puts a step's adds right behind the MFMAs that produce their operands and fills the gap with
                // s_nop 5 (measured: the adds of this sweep cost 29 % of the kernel)
#pragma unroll
                for (int m = 0; m < 2; m++) {
                    t4[m] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[sstep][m], bhv, zero, 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                    if (g > 40 && HAF_ABL != 1) acc[m][pn] = h_add4(acc[m][pn], tp[m]);    // (packed: two v_pk_add_f32)
                    __builtin_amdgcn_sched_barrier(0);
                }
                tp[0] = t4[0];
                tp[1] = t4[1];
            }
#undef HAF_H_FRAG
#pragma unroll
            for (int n = 0; n < 2; n++) {
                f32x4 t4[2];
#pragma unroll
                for (int m = 0; m < 2; m++) t4[m] = __builtin_amdgcn_mfma_f32_16x16x16f16(aht[m], bht[n], zero, 0, 0, 0);
                const int pn = n ^ 1;                                  // (kHFull - 1, 1) before tail 0, tail 0 before tail 1
#pragma unroll
                for (int m = 0; m < 2; m++) acc[m][pn] = h_add4(acc[m][pn], tp[m]);
                tp[0] = t4[0];
                tp[1] = t4[1];
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int m = 0; m < 2; m++) acc[m][1] = h_add4(acc[m][1], tp[m]);            // tail 1
        } else {
        half8 bh[2][2], bq[2][2];                                    // [ring][column block n]
#pragma unroll
        for (int n = 0; n < 2; n++) {
            bh[0][n] = *reinterpret_cast<const half8 *>(bl + n * 1024);                  // B[k = 32s + 8(lane>>4) + j][col 16n + (lane&15)]
            bq[0][n] = *reinterpret_cast<const half8 *>(bl + kHMatBytes + n * 1024);
        }
#pragma unroll
        for (int s = 0; s < kHFull; s++) {
            const int c = s & 1, nx = c ^ 1;
            if (s + 1 < kHFull) {
#pragma unroll
                for (int n = 0; n < 2; n++) {
                    bh[nx][n] = *reinterpret_cast<const half8 *>(bl + ((s + 1) * 2 + n) * 1024);
                    bq[nx][n] = *reinterpret_cast<const half8 *>(bl + kHMatBytes + ((s + 1) * 2 + n) * 1024);
                }
            }
#pragma unroll
            for (int m = 0; m < 2; m++)
#pragma unroll
                for (int n = 0; n < 2; n++) {
                    acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[s][m], bh[c][n], acc[m][n], 0, 0, 0);
                    acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[s][m], bh[c][n], acc[m][n], 0, 0, 0);
                    acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[s][m], bq[c][n], acc[m][n], 0, 0, 0);
                }
        }
        {   // K tail: attributes 320..335, 16x16x16 form (4 halfs per lane)
            half4 bht[2], bqt[2];
#pragma unroll
            for (int n = 0; n < 2; n++) {
                bht[n] = *reinterpret_cast<const half4 *>(cur + kHTailOff + n * 512 + lane * 8);
                bqt[n] = *reinterpret_cast<const half4 *>(cur + kHMatBytes + kHTailOff + n * 512 + lane * 8);
            }
#pragma unroll
            for (int m = 0; m < 2; m++)
#pragma unroll
                for (int n = 0; n < 2; n++) {
                    acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x16f16(aht[m], bht[n], acc[m][n], 0, 0, 0);
                    acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x16f16(alt[m], bht[n], acc[m][n], 0, 0, 0);
                    acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x16f16(aht[m], bqt[n], acc[m][n], 0, 0, 0);
                }
        }
        }
        const float *tail = reinterpret_cast<const float *>(cur + 2 * kHMatBytes);
        // HAZARD (measured on gfx950, two waves per SIMD; screen.hip has the details): a VALU instruction that reads a
        // v_exp_f32 result within a few instructions of the v_exp_f32 can read the register before it is written.  All
        // sixteen exps are issued first and pinned there; the coefficient fmas follow.
        if (PRECISE) {
            // the arithmetic around the sixteen exps in packed fp32 (same results): 24 vector instructions instead of 48; a_x comes
            // back from LDS (its eight registers go to the fragment ring during the sweeps); `lo` starts from zero in every tile
            f32x4 ax4[2], l4[2];
#pragma unroll
            for (int m = 0; m < 2; m++) ax4[m] = *reinterpret_cast<const f32x4 *>(axs + 16 * m + 4 * (lane >> 4));
            float cfn[2];
#pragma unroll
            for (int n = 0; n < 2; n++) {
                const float as_ = tail[16 * n + (lane & 15)];        // -g2*|s_j|^2
                cfn[n] = tail[kTile + 16 * n + (lane & 15)];         // coef_j (0 for padding SVs)
#pragma unroll
                for (int m = 0; m < 2; m++) {                        // 16x16 C/D layout: col = lane&15, row = 16m + 4(lane>>4) + reg
                    if (CRP) {
#pragma unroll
                        for (int r = 0; r < 4; r++) {
                            const float z = p.two_gamma2 * acc[m][n][r];
                            acc[m][n][r] = (z * z) * fmaf(fmaf(fmaf(z, kPsiA5h, kPsiA4h), z, kPsiA3h), z, kPsiA2h);
                        }
                        continue;
                    }
                    const f32x4 arg = h_fma4s(p.two_gamma2, acc[m][n], h_add4s(ax4[m], as_));
#pragma unroll
                    for (int r = 0; r < 4; r++) acc[m][n][r] = (HAF_ABL == 2) ? arg[r] : __builtin_amdgcn_exp2f(arg[r]);
                }
            }
            if (!CRP) asm volatile("s_nop 7\n\ts_nop 7" : "+v"(acc[0][0]), "+v"(acc[0][1]), "+v"(acc[1][0]), "+v"(acc[1][1]));
#pragma unroll
            for (int m = 0; m < 2; m++) l4[m] = h_fma4s(cfn[1], acc[m][1], h_fma4s(cfn[0], acc[m][0], f32x4{0.0f, 0.0f, 0.0f, 0.0f}));
#pragma unroll
            for (int m = 0; m < 2; m++)
#pragma unroll
                for (int r = 0; r < 4; r++) part[m][r] += (part_t)l4[m][r];
        } else {
        float cfn[2];
#pragma unroll
        for (int n = 0; n < 2; n++) {
            const float as_ = tail[16 * n + (lane & 15)];            // -g2*|s_j|^2
            cfn[n] = tail[kTile + 16 * n + (lane & 15)];             // coef_j (0 for padding SVs)
#pragma unroll
            for (int m = 0; m < 2; m++)
#pragma unroll
                for (int r = 0; r < 4; r++)                          // 16x16 C/D layout: col = lane&15, row = 16m + 4(lane>>4) + reg
                    acc[m][n][r] = __builtin_amdgcn_exp2f(fmaf(p.two_gamma2, acc[m][n][r], axr[m][r] + as_));
        }
        // the wait states hang on the data (see k_svm_rbf): all sixteen results go through the asm
        asm volatile("s_nop 7\n\ts_nop 7" : "+v"(acc[0][0]), "+v"(acc[0][1]), "+v"(acc[1][0]), "+v"(acc[1][1]));
#pragma unroll
        for (int n = 0; n < 2; n++)
#pragma unroll
            for (int m = 0; m < 2; m++)
#pragma unroll
                for (int r = 0; r < 4; r++) lo[m][r] = fmaf(cfn[n], acc[m][n][r], lo[m][r]);
        if ((t & (kFold - 1)) == kFold - 1) {
#pragma unroll
            for (int m = 0; m < 2; m++)
#pragma unroll
                for (int r = 0; r < 4; r++) { part[m][r] += (part_t)lo[m][r]; lo[m][r] = 0.0f; }
        }
        }
        // tile t+1 must have landed before anyone reads it; the pieces of tile t+2 (just issued) may stay in flight
        // ALL of this wave's pieces, those of tile t+2 included (they have had this tile's whole time to land), not a counted
        // vmcnt(pieces of t+2): with the counted wait the bulk form turned nondeterministic inside its band -- a few evaluations per
        // workgroup off by one lo-image fragment's worth, i.e. a late piece of tile t+1 read before it had landed -- in a build that
        // differed only in the code of other kernels, and was deterministic again with this wait ("Counted waits on
        // LDS-DMA do not hold": the likely reading is that LDS-DMA loads do not complete in issue order, so counting them proves nothing).
        (void)my_pieces;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");                              // no LDS read of the next tile may move above the barrier
    }

#pragma unroll
    for (int m = 0; m < 2; m++)
#pragma unroll
        for (int r = 0; r < 4; r++) {
            part_t v = part[m][r] + (part_t)lo[m][r];
            v += __shfl_xor(v, 8, 64);
            v += __shfl_xor(v, 4, 64);
            v += __shfl_xor(v, 2, 64);
            v += __shfl_xor(v, 1, 64);
            part[m][r] = v;
        }
    if ((lane & 15) == 0) {
        const bool has_neg = d.sv_tile_neg < nt;      // (a range that starts behind the class boundary found pos[] = 0)
#pragma unroll
        for (int m = 0; m < 2; m++)
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int row = 16 * m + 4 * (lane >> 4) + r;
                const long es = tile32 * kTile + row;
                if (es < n_evals) {
                    const int e = idx_list ? idx_list[es] : (int)es;
                    const part_t P = has_neg ? (part_t)pos[row] : part[m][r];         // (bulk form: pos[] holds an fp32 value)
                    const part_t N = has_neg ? part[m][r] : (part_t)0;
                    if (part_out) {
                        part_out[(2 * blockIdx.y) * part_stride + es] = P;
                        part_out[(2 * blockIdx.y + 1) * part_stride + es] = N;
                        continue;
                    }
                    const float dv = (float)((P + N) - (part_t)p.rho);   // (PRECISE: one rounding)
                    const float sabs = (float)(P - N);                   // sum |coef| K
                    dec[e] = dv;
                    labels[evalcell[e]] = (int8_t)(dv > 0.0f ? p.gv0 : p.gv1);
                    const float gdot = PRECISE ? p.guard_dot_p : p.guard_dot;
                    const float gacc = PRECISE ? p.guard_acc_l : p.guard_acc;
                    const bool undecided = !(fabsf(dv) > (gacc + gdot * (p.as_max + fabsf(axs[row]))) * sabs + p.guard_abs);
                    if (t1flags) t1flags[es] = undecided ? 1 : 0;
                    else if (undecided) {
                        int slot = atomicAdd(&counters_rw[CNT_FLAGGED], 1);
                        if (slot < flag_cap) flag_list[slot] = e;
                    }
                }
            }
    }
}

// list mode: sums the class sums of the kHListParts tile ranges in a fixed order and finishes the evaluation exactly as the
// kernel's own epilogue does
__global__ __launch_bounds__(256) void k_svm_h_combine(const double *__restrict__ part_out, long part_stride, int parts,
                                                       const float *__restrict__ ax, const int *__restrict__ evalcell,
                                                       const int *__restrict__ counters, SvmParams p, float *__restrict__ dec,
                                                       int8_t *__restrict__ labels, int *__restrict__ flag_list, int flag_cap,
                                                       int *__restrict__ counters_rw, const int *__restrict__ idx_list,
                                                       int list_counter, int list_cap, unsigned char *__restrict__ t1flags)
{
    const int n_evals = min(counters[list_counter], list_cap);
    parts = h_list_parts(n_evals, parts, part_stride);
    part_stride = h_list_stride(part_stride, parts);
    for (long es = (long)blockIdx.x * 256 + threadIdx.x; es < n_evals; es += (long)gridDim.x * 256) {
        double P = 0.0, N = 0.0;                        // the ranges are added in fp64 like the sums inside them
        for (int y = 0; y < parts; y++) {
            P += part_out[(2 * y) * part_stride + es];
            N += part_out[(2 * y + 1) * part_stride + es];
        }
        const int e = idx_list[es];
        const float dv = (float)((P + N) - (double)p.rho);
        const float sabs = (float)(P - N);              // sum |coef| K
        dec[e] = dv;
        labels[evalcell[e]] = (int8_t)(dv > 0.0f ? p.gv0 : p.gv1);
        const bool undecided = !(fabsf(dv) > (p.guard_acc_l + p.guard_dot_p * (p.as_max + fabsf(ax[es]))) * sabs + p.guard_abs);
        if (t1flags) t1flags[es] = undecided ? 1 : 0;
        else if (undecided) {
            int slot = atomicAdd(&counters_rw[CNT_FLAGGED], 1);
            if (slot < flag_cap) flag_list[slot] = e;
        }
    }
}

// centred-remainder form of the list mode (k_svm_rbf_h<true, true>): P, N are the class sums of b psi(z); dec = A (B0 + L + P + N) - rho in
// fp64, A = 2^(a_x) with a_x = -gamma'|x~ - m|^2 of the operand the passes multiplied.  Band: screen_finish_cr's bound with the
// operands' errors those of the hi + lo split and the accumulation that of the PRECISE form (CrT1Params).
__global__ __launch_bounds__(256) void k_svm_h_combine_cr(const double *__restrict__ part_out, long part_stride, int parts,
                                                          const float *__restrict__ ax, const double *__restrict__ Lbuf,
                                                          const int *__restrict__ evalcell, const int *__restrict__ counters, CrT1Params c,
                                                          float *__restrict__ dec, int8_t *__restrict__ labels,
                                                          int *__restrict__ flag_list, int flag_cap, int *__restrict__ counters_rw,
                                                          const int *__restrict__ idx_list, int list_counter, int list_cap,
                                                          unsigned char *__restrict__ t1flags)
{
    const int n_evals = min(counters[list_counter], list_cap);
    parts = h_list_parts(n_evals, parts, part_stride);
    part_stride = h_list_stride(part_stride, parts);
    const double ln2 = 0.69314718056, u24 = 5.9604644775390625e-08;
    for (long es = (long)blockIdx.x * 256 + threadIdx.x; es < n_evals; es += (long)gridDim.x * 256) {
        double P = 0.0, N = 0.0;
        for (int y = 0; y < parts; y++) {
            P += part_out[(2 * y) * part_stride + es];
            N += part_out[(2 * y + 1) * part_stride + es];
        }
        const int e = idx_list[es];
        const double a_x = -(double)ax[es];                             // gamma'|x~ - m|^2 (fp32: u a_x in the exponent)
        const double A = exp2(-a_x);
        const double L = Lbuf[es];
        const double dvd = A * ((c.B0 + L) + (P + N)) - c.rho;
        const float dv = (float)dvd;
        const double spsi = P - N;
        // |p~| = c |x~ - m| = sqrt(2 a_x); p~ against the true p: fp32 rounding of x - m, the fp16 hi + lo split, flushed lo subnormals
        const double ph = sqrt_upper(2.0 * a_x * (1.0 + 2.0 * u24)) + 1e-30;
        const double dn = c.dp_rel * ph + c.dp_abs, un = ph + dn;
        const double eps = dn * c.qmax + un * c.dqmax + c.acc_rel * ph * c.qmax + 2.0 * u24 * ph * c.qmax;   // (+ the fp32 product with 2 gamma')
        const double zmax = ph * c.qmax + eps;
        const double zf = floor(zmax);
        const double p2 = (zmax < 60.0) ? ldexp(1.0 + (zmax - zf), (int)zf) : (double)__builtin_inff();
        const double acc_sum = fmin(ph * un * c.Ca, sqrt_upper(c.nHabs) * un * sqrt_upper(c.nHaa) * ph);    // (screen_finish_cr: the same two bounds)
        const double quad1 = ln2 * ln2 * (c.nN * un * dn + c.nM * un * un + (c.acc_rel + 2.0 * u24) * acc_sum);
        const double quad2 = 1.5 * ln2 * ln2 * (c.nHabs * dn * dn + c.nDabs * un * un + c.acc_rel * c.acc_rel * ph * ph * c.Cqq);
        const double cub2 = ln2 * ln2 * (p2 - 1.0) * eps * eps * c.Babs * 1.01;
        const double t = ln2 * zmax;
        double k_psi = ln2 * eps * 1.01 + c.sum_rel + 4.1 * t * t * t * t / 360.0;
        if (!(t <= 1.0)) k_psi = (double)__builtin_inff();
        // the common factor: a_x in fp32, x~ against x in |x - m|^2 (2 |p||dp| + |dp|^2 in the exponent, log2 units: x 1/2), exp2 in fp64
        const double D = u24 * a_x + un * dn + 0.5 * dn * dn + 1e-12;
        const double cm = exp2m1_upper(D);
        // L: an fp64 sum of exact terms (the attributes are the exact ones): 2^-50 of the terms' magnitudes at most
        const double cL = 1e-15 * (fabs(L) + 1.0);
        const double err = ((A * (quad1 + quad2 + cub2 + cL + k_psi * spsi) * (1.0 + cm) + cm * (fabs(dvd) + fabs(c.rho))) * c.scale + 2.4e-7 * fabs(dvd)) * 1.002 +
                           (double)c.guard_abs;
        dec[e] = dv;
        labels[evalcell[e]] = (int8_t)(dv > 0.0f ? c.gv0 : c.gv1);
        const bool undecided = !(fabs(dvd) > err) || !(D < 0.05);
        if (t1flags) t1flags[es] = undecided ? 1 : 0;
        else if (undecided) {
            int slot = atomicAdd(&counters_rw[CNT_FLAGGED], 1);
            if (slot < flag_cap) flag_list[slot] = e;
        }
    }
}

void launch_svm_h(const void *Xh, const float *ax, const void *svt_h, const int *evalcell, const int *counters, SvmParams p,
                  float *dec, int8_t *labels, int *flag_list, int flag_cap, int *counters_rw, Dims d, long max_evals,
                  const int *idx_list, int list_counter, int list_cap, double *part_out, long part_stride, hipStream_t s,
                  const CrT1Params *cr, const double *Lbuf, unsigned char *t1flags)
{
    long blocks = (max_evals + kSvmBlockEvals - 1) / kSvmBlockEvals;
    if (blocks <= 0) return;
    // SV tile ranges of the list mode: kHListParts for long lists; for a request that cannot fill the chip (at most 256 workgroups even
    // if everything were listed) up to kHListPartsShort, settled on the device from the live length (h_list_parts)
    const bool short_req = blocks <= 256 && d.n_sv_tiles >= 2 * kHListPartsShort;
    const int list_parts = d.n_sv_tiles >= 4 * kHListParts ? (short_req ? kHListPartsShort : kHListParts) : 1;
    if (idx_list && cr && part_out) {
        const int parts = list_parts;
        hipLaunchKernelGGL((k_svm_rbf_h<true, true>), dim3((unsigned)blocks, (unsigned)parts), dim3(kSvmThreads), 0, s, (const char *)Xh, ax,
                           (const char *)svt_h, evalcell, counters, p, dec, labels, flag_list, flag_cap, counters_rw, d, idx_list,
                           list_counter, list_cap, part_out, part_stride, t1flags);
        hipLaunchKernelGGL(k_svm_h_combine_cr, dim3(1024), dim3(256), 0, s, part_out, part_stride, parts, ax, Lbuf, evalcell, counters, *cr, dec,
                           labels, flag_list, flag_cap, counters_rw, idx_list, list_counter, list_cap, t1flags);
        if (t1flags) launch_t1_handover(t1flags, counters, list_counter, list_cap, idx_list, flag_list, flag_cap, counters_rw, s, max_evals, t1_blkcount(t1flags, max_evals));
        return;
    }
    if (idx_list) {
        const int parts = part_out ? list_parts : 1;     // engine.cpp: guard_acc_l follows this rule
        double *po = parts > 1 ? part_out : nullptr;
        hipLaunchKernelGGL(k_svm_rbf_h<true>, dim3((unsigned)blocks, (unsigned)parts), dim3(kSvmThreads), 0, s, (const char *)Xh, ax,
                           (const char *)svt_h, evalcell, counters, p, dec, labels, flag_list, flag_cap, counters_rw, d, idx_list,
                           list_counter, list_cap, po, part_stride, t1flags);
        if (po)
            hipLaunchKernelGGL(k_svm_h_combine, dim3(1024), dim3(256), 0, s, po, part_stride, parts, ax, evalcell, counters, p, dec,
                               labels, flag_list, flag_cap, counters_rw, idx_list, list_counter, list_cap, t1flags);
        if (t1flags) launch_t1_handover(t1flags, counters, list_counter, list_cap, idx_list, flag_list, flag_cap, counters_rw, s, max_evals, t1_blkcount(t1flags, max_evals));
    } else {
        hipLaunchKernelGGL(k_svm_rbf_h<false>, dim3((unsigned)blocks), dim3(kSvmThreads), 0, s, (const char *)Xh, ax, (const char *)svt_h,
                           evalcell, counters, p, dec, labels, flag_list, flag_cap, counters_rw, d, idx_list, list_counter, list_cap,
                           (double *)nullptr, 0L, t1flags);
        if (t1flags) launch_t1_handover(t1flags, counters, CNT_EVALS, 0x7fffffff, nullptr, flag_list, flag_cap, counters_rw, s, max_evals, t1_blkcount(t1flags, max_evals));
    }
}

size_t t1_flag_bytes(long max_entries) { return (((size_t)max_entries + 63) / 64) * 64 + ((size_t)max_entries / kT1Chunk + 2) * sizeof(int); }

}  // namespace haf
