// exact8.hip -- tier 2a of the RBF decision: the exact-integer tier (kernels.h: "tier 2a").
//
// The evaluations the three-pass kernel could not decide (|dec| inside ~5e-6 sum|coef|K: the fp32 accumulation of the matrix
// core) used to go straight to the fp64 MFMA tier, which runs at 1/32 of the fp16 matrix rate and was the largest cost on
// models whose decision values crowd around zero.  Here they first meet a contraction that has NO accumulation error at all:
//   * attributes (k_features_small<XMODE_I8>) and support vectors (engine.cpp) as fixed-point integers -- kI8Q = 23 fractional
//     bits for the attributes (|value| < 15.87), q_s for the support vectors: as many as the model's largest component leaves room
//     for (26 for SVs inside +-1.98) -- each split into four balanced base-128 digits (int8, -64..63);
//   * the digit-by-digit products of the 384-long dot product through v_mfma_i32_16x16x64_i8, one int32 accumulator per digit
//     weight 128^(6-w), w = j + k: at most 4 x 384 x 64 x 64 < 2^23 per accumulator, so every partial sum is EXACT.  The three
//     products of weights 128 and 1 (w = 5, 6) are left out: together at most (2 x 128 + 1) x 64 x 64 x 2^-(23 + q_s) <= 1.5e-8 per
//     attribute, |error of xq.sq| <= 4.9e-6 -- a tenth of what the quantisation costs -- and part of the band (I8Params::drop);
//   * |xq - sq|^2 = |xq|^2 + |sq|^2 - 2 xq.sq from the exact integers (the accumulators combined in int32 pairs, then three fp64
//     operations), 2^(-gamma' d^2) by range reduction and a degree-11 polynomial (1e-14 relative), fp64 sums.  (What bounds this
//     kernel is the SIMD's vector issue port: with libm's exp and seven conversions per element it ran at 60 % of its MFMA time.)
// The only error of a kernel value is the quantisation of the operands, |(x - xq) - (s - sq)|_2 <= delta = sqrt(324) (2^-24 + 2^-(q_s+1)):
//   | |x-s|^2 - |xq-sq|^2 | <= delta (2 |xq - sq| + delta) <= delta (2 (|xq| + max|sq|) + delta),
// i.e. |dec_q - dec| <= (exp(gamma * that) - 1) * S -- about 2e-7 S on the bench models (q_s = 26), 20x inside the three-pass band; what
// is still closer to zero than that goes on to the fp64 MFMA tier (k_recheck_mfma) exactly as before.  Same task structure as
// that tier: a workgroup = 4 waves x 16 evaluations, one of kMSplit ranges of SV tiles, partial sums added in a fixed order.
#include "device_common.h"

namespace haf {

typedef int i32x4 __attribute__((ext_vector_type(4)));

constexpr int kI8Waves = 4;
constexpr int kI8Evals = 16 * kI8Waves;
constexpr int kI8Split = 8;                        // SV ranges per group of evaluations (= kMSplit of the fp64 tier: same part64 layout)
static_assert(kRecheckPartRows == 2 * kI8Split + 1, "part64 layout shared with the fp64 tier");
constexpr int kI8TileLoads = kI8GroupBytes / (256 * 16);      // 16-byte loads per thread per tile image (6)
static_assert(kI8TileLoads * 256 * 16 == kI8GroupBytes, "tile image = whole 16-byte loads of 256 threads");

__device__ __forceinline__ int window_count8(int total, int off, int cap) { return max(0, min(total - off, cap)); }

// 2^y for y <= 0.5: n = rint(y), f = y - n in [-0.5, 0.5] exactly, Taylor polynomial of 2^f of degree 11 (truncation 8.8e-15
// relative, Horner roundings ~1.5e-15), v_ldexp_f64.  No table, no branch, 15 instructions instead of libm's ~40.
__device__ __forceinline__ double exp2_poly(double y)
{
    const double n = rint(y);
    const double f = y - n;
    double p = 4.4455382718708101e-10;
    p = fma(p, f, 7.0549116208011209e-09);
    p = fma(p, f, 1.0178086009239696e-07);
    p = fma(p, f, 1.3215486790144305e-06);
    p = fma(p, f, 1.5252733804059838e-05);
    p = fma(p, f, 0.00015403530393381606);
    p = fma(p, f, 0.0013333558146428441);
    p = fma(p, f, 0.0096181291076284769);
    p = fma(p, f, 0.055504108664821576);
    p = fma(p, f, 0.24022650695910069);
    p = fma(p, f, 0.69314718055994529);
    p = fma(p, f, 1.0);
    return ldexp(p, (int)fmax(n, -1200.0));
}

// the 13 x 6 MFMAs of one SV tile (16 SVs) against this wave's 16 evaluations
__device__ __forceinline__ void i8_tile_mfma(const char *bt, int lane, const i32x4 (&a)[kI8Slices][kI8Steps], i32x4 (&acc)[5])
{
#pragma unroll
    for (int w = 0; w < 5; w++) acc[w] = i32x4{0, 0, 0, 0};
    // B fragments one k-step ahead of their MFMAs (the LDS round trip of a k-step's four reads hides behind the previous step's 13 MFMAs)
    i32x4 b[kI8Slices], bn[kI8Slices];
#pragma unroll
    for (int k = 0; k < kI8Slices; k++) b[k] = *reinterpret_cast<const i32x4 *>(bt + (k * kI8Steps) * 1024 + lane * 16);
#pragma unroll
    for (int ks = 0; ks < kI8Steps; ks++) {
        if (ks + 1 < kI8Steps) {
#pragma unroll
            for (int k = 0; k < kI8Slices; k++) bn[k] = *reinterpret_cast<const i32x4 *>(bt + (k * kI8Steps + ks + 1) * 1024 + lane * 16);
        }
        // the 13 digit pairs (j, k), j + k <= 4, in an order that brings the same accumulator back every third instruction at the
        // earliest (also across the k-steps): a dependent MFMA does not issue back to back
#define HAF_I8_MFMA(j, k) acc[(j) + (k)] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a[j][ks], b[k], acc[(j) + (k)], 0, 0, 0)
        HAF_I8_MFMA(0, 3); HAF_I8_MFMA(1, 3); HAF_I8_MFMA(0, 2); HAF_I8_MFMA(1, 2); HAF_I8_MFMA(0, 1); HAF_I8_MFMA(2, 2); HAF_I8_MFMA(2, 1);
        HAF_I8_MFMA(1, 1); HAF_I8_MFMA(0, 0); HAF_I8_MFMA(3, 0); HAF_I8_MFMA(3, 1); HAF_I8_MFMA(2, 0); HAF_I8_MFMA(1, 0);
#undef HAF_I8_MFMA
#pragma unroll
        for (int k = 0; k < kI8Slices; k++) b[k] = bn[k];
    }
}

// kernel values and class sums of one finished tile: this lane's column (SV) against its four rows (evaluations)
__device__ __forceinline__ void i8_tile_epilogue(const i32x4 (&acc)[5], const double (&xx)[4], double ss, double cf, double gamma2, double dq_scale,
                                                 double (&part)[4], double (&pabs)[4])
{
#pragma unroll
    for (int r = 0; r < 4; r++) {
        // xq.sq 2^(kI8Q + q_s) = acc0 128^6 + (acc1 128 + acc2) 128^4 + (acc3 128 + acc4) 128^2: the pairs in int32 (< 2^30), then
        // two fp64 fmas on exact integers (< 2^63: one rounding of 2^-53 relative at most)
        const int u1 = (acc[1][r] << 7) + acc[2][r], u2 = (acc[3][r] << 7) + acc[4][r];
        const double dq = fma(fma((double)acc[0][r], 16384.0, (double)u1), 16384.0, (double)u2);
        const double d2 = fma(dq_scale, dq, xx[r] + ss);
        const double kv = exp2_poly(-gamma2 * d2);
        part[r] = fma(cf, kv, part[r]);
        pabs[r] = fma(fabs(cf), kv, pabs[r]);
    }
}

// SV ranges of a group of evaluations and the row pitch of part64, as recheck_split of the fp64 tier (recheck.hip): kI8Split ranges on
// one row per slot of the window -- or, for a list of at most 1 024 entries that fills at most a quarter of the window, up to four times
// as many on a quarter of the pitch, as many as give the chip's 512 workgroup slots one task each (round 5: C3 against a 4 096-SV model
// leaves this tier 31 entries -- one group, eight tasks of 32 tiles, 47 us).  Both kernels derive the same numbers from the same counter.
// (|xq|^2, which the feature kernel writes, stays at row 2 kI8Split of the FULL pitch: 64 rows of a quarter pitch end exactly there.)
__device__ __forceinline__ void i8_split(int n_flag, int flag_cap, int n_tiles, int &splits, size_t &pitch)
{
    const bool fine = (long)n_flag * 4 <= (long)flag_cap && n_tiles >= 16 * kI8Split && flag_cap >= 4 && n_flag <= 1024;
    const int n_groups = (n_flag + kI8Evals - 1) / kI8Evals;
    splits = fine ? max(kI8Split, min(4 * kI8Split, 512 / max(n_groups, 1))) : kI8Split;
    pitch = fine ? (size_t)(flag_cap / 4) : (size_t)flag_cap;
}

__global__ __launch_bounds__(256, 2) void k_recheck_i8(const char *__restrict__ ximg, const char *__restrict__ svimg, I8Params p,
                                                       int flag_cap, int list_off, const int *__restrict__ counters, int cslot,
                                                       double *__restrict__ part64)
{
    __shared__ __attribute__((aligned(16))) char bt[kI8SvTileBytes];
    const int n_flag = window_count8(counters[cslot], list_off, flag_cap);
    const int n_groups = (n_flag + kI8Evals - 1) / kI8Evals;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int n_tiles = p.n_sv_pad / 16;
    int splits;
    size_t pitch;
    i8_split(n_flag, flag_cap, n_tiles, splits, pitch);
    const int tiles_per_part = (n_tiles + splits - 1) / splits;
    const double *xnorm = part64 + (size_t)(2 * kI8Split) * flag_cap;      // |xq|^2 per slot, written by the feature kernel
    for (int task = blockIdx.x; task < n_groups * splits; task += gridDim.x) {
        const int g = task / splits, h = task - g * splits;
        const int t_begin = h * tiles_per_part, t_end = min(n_tiles, t_begin + tiles_per_part);
        const int grp = g * kI8Waves + wave;                         // this wave's 16 slots
        // ---- A operand: the digit image of 16 evaluations, [digit][k-step][lane][16 int8]: 1 KiB per wave load ----
        i32x4 a[kI8Slices][kI8Steps];
        {
            const char *xg = ximg + (size_t)grp * kI8GroupBytes + lane * 16;
#pragma unroll
            for (int j = 0; j < kI8Slices; j++)
#pragma unroll
                for (int ks = 0; ks < kI8Steps; ks++) a[j][ks] = *reinterpret_cast<const i32x4 *>(xg + (j * kI8Steps + ks) * 1024);
        }
        double xx[4];
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const int sl = grp * 16 + 4 * (lane >> 4) + r;            // C/D map: row = 4 (lane >> 4) + reg
            xx[r] = (sl < n_flag) ? xnorm[sl] : 0.0;
        }
        // the next tile waits in registers while this one is consumed (explicit loops: through a lambda hipcc parks them in scratch)
        uint4 pre0, pre1, pre2, pre3, pre4, pre5, pre_c = uint4{0, 0, 0, 0};
#define HAF_I8_TILE_LOAD(t)                                                                                   \
        {                                                                                                         \
            const char *src__ = svimg + (size_t)(t) * kI8SvTileBytes + (size_t)tid * 16;                          \
            pre0 = *reinterpret_cast<const uint4 *>(src__);                                                       \
            pre1 = *reinterpret_cast<const uint4 *>(src__ + 4096);                                                \
            pre2 = *reinterpret_cast<const uint4 *>(src__ + 8192);                                                \
            pre3 = *reinterpret_cast<const uint4 *>(src__ + 12288);                                               \
            pre4 = *reinterpret_cast<const uint4 *>(src__ + 16384);                                               \
            pre5 = *reinterpret_cast<const uint4 *>(src__ + 20480);                                               \
            if (tid < 16) pre_c = *reinterpret_cast<const uint4 *>(src__ + kI8GroupBytes);                        \
        }
#define HAF_I8_TILE_STORE()                                                                                   \
        {                                                                                                         \
            char *dst__ = bt + (size_t)tid * 16;                                                                  \
            *reinterpret_cast<uint4 *>(dst__) = pre0;                                                             \
            *reinterpret_cast<uint4 *>(dst__ + 4096) = pre1;                                                      \
            *reinterpret_cast<uint4 *>(dst__ + 8192) = pre2;                                                      \
            *reinterpret_cast<uint4 *>(dst__ + 12288) = pre3;                                                     \
            *reinterpret_cast<uint4 *>(dst__ + 16384) = pre4;                                                     \
            *reinterpret_cast<uint4 *>(dst__ + 20480) = pre5;                                                     \
            if (tid < 16) *reinterpret_cast<uint4 *>(dst__ + kI8GroupBytes) = pre_c;                              \
        }
        static_assert(kI8TileLoads == 6, "six 16-byte loads per thread");
        __syncthreads();                                             // (the previous task's last tile is no longer read)
        if (t_begin < t_end) { HAF_I8_TILE_LOAD(t_begin); HAF_I8_TILE_STORE(); }
        __syncthreads();
        double part[4] = {0, 0, 0, 0}, pabs[4] = {0, 0, 0, 0};
        // Software pipeline over the tiles: the fp64 epilogue of tile t-1 (its accumulators and its column constants wait in
        // registers) is issued BETWEEN the MFMAs of tile t -- one MFMA, two vector instructions, 78 times -- instead of behind
        // them: the wave keeps the matrix pipe fed while its own vector work runs (MFMA busy 58 % -> see DESIGN.md 5).
        const double *cst = reinterpret_cast<const double *>(bt + kI8GroupBytes);
        i32x4 acc_prev[5], acc_cur[5];
        double ss_prev = 0.0, cf_prev = 0.0;
        if (t_begin < t_end) {                                         // first tile: nothing to overlap with
            if (t_begin + 1 < t_end) HAF_I8_TILE_LOAD(t_begin + 1);
            i8_tile_mfma(bt, lane, a, acc_prev);
            ss_prev = cst[lane & 15]; cf_prev = cst[16 + (lane & 15)];  // this lane's column: |sq|^2 and coef (0: padding)
            __syncthreads();                                          // everyone is done reading the tile
            if (t_begin + 1 < t_end) HAF_I8_TILE_STORE();
            __syncthreads();
        }
        for (int t = t_begin + 1; t < t_end; t++) {
            if (t + 1 < t_end) HAF_I8_TILE_LOAD(t + 1);
            i8_tile_mfma(bt, lane, a, acc_cur);
            i8_tile_epilogue(acc_prev, xx, ss_prev, cf_prev, p.gamma2, p.dq_scale, part, pabs);
            __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);         // the first k-step's B fragments
#pragma unroll
            for (int ks = 0; ks < kI8Steps; ks++) {
                if (ks + 1 < kI8Steps) __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);   // the next k-step's B fragments (LDS reads)
#pragma unroll
                for (int q = 0; q < 13; q++) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);     // one MFMA
                    __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);     // two VALU instructions of the previous tile's epilogue
                }
            }
            ss_prev = cst[lane & 15]; cf_prev = cst[16 + (lane & 15)];
#pragma unroll
            for (int w = 0; w < 5; w++) acc_prev[w] = acc_cur[w];
            __syncthreads();                          // everyone is done reading the tile
            if (t + 1 < t_end) HAF_I8_TILE_STORE();
            __syncthreads();
        }
        if (t_begin < t_end) i8_tile_epilogue(acc_prev, xx, ss_prev, cf_prev, p.gamma2, p.dq_scale, part, pabs);
#pragma unroll
        for (int r = 0; r < 4; r++) {
            double v = part[r], w = pabs[r];
            v += __shfl_xor(v, 8, 64); w += __shfl_xor(w, 8, 64);
            v += __shfl_xor(v, 4, 64); w += __shfl_xor(w, 4, 64);
            v += __shfl_xor(v, 2, 64); w += __shfl_xor(w, 2, 64);
            v += __shfl_xor(v, 1, 64); w += __shfl_xor(w, 1, 64);
            part[r] = v; pabs[r] = w;
        }
        if ((lane & 15) == 0) {
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int sl = grp * 16 + 4 * (lane >> 4) + r;
                if (sl < n_flag) {
                    part64[(size_t)(2 * h) * pitch + sl] = part[r];
                    part64[(size_t)(2 * h + 1) * pitch + sl] = pabs[r];
                }
            }
        }
    }
}

// sum of the kI8Split partial decision values (fixed order), label, and what is still inside the quantisation band goes on to
// the fp64 MFMA tier's list
__global__ __launch_bounds__(256) void k_recheck_i8_combine(const double *__restrict__ part64, const int *__restrict__ evalcell, I8Params p,
                                                            const int *__restrict__ flag_list, int flag_cap, int list_off,
                                                            int *__restrict__ counters, int cslot, double *__restrict__ dec_exact,
                                                            int8_t *__restrict__ labels, int *__restrict__ flagi_list, int flagi_cap,
                                                            unsigned long long *__restrict__ words)
{
    const int n_flag = window_count8(counters[cslot], list_off, flag_cap);
    const int lane = threadIdx.x & 63;
    int splits;
    size_t pitch;
    i8_split(n_flag, flag_cap, p.n_sv_pad / 16, splits, pitch);
    // (wave-uniform trip count: every lane takes part in the ballot of its 64 entries; words != nullptr: ordered hand-over, device_common.h)
    for (int base = blockIdx.x * 256 + (threadIdx.x & ~63); base < n_flag; base += gridDim.x * 256) {
        const int sl = base + lane;
        bool undecided = false;
        if (sl < n_flag) {
        double P = 0.0, S = 0.0;
        for (int h = 0; h < splits; h++) {
            P += part64[(size_t)(2 * h) * pitch + sl];
            S += part64[(size_t)(2 * h + 1) * pitch + sl];
        }
        const double dv = P - p.rho;
        const double xq2 = part64[(size_t)(2 * kI8Split) * flag_cap + sl];
        const int e = flag_list[sl];
        bool decided = false;
        if (xq2 >= 0.0) {                                            // (negative: an attribute beyond the fixed-point range)
            // relative error of every kernel value: exp(gamma | |x-s|^2 - |xq-sq|^2 |) - 1, with |xq - sq| <= |xq| + max|sq|;
            // e^y - 1 <= y (1 + y) for y < 1; the fp64 roundings of exp and of the sums are inside 2^-40 like the fp64 tier's
            // (+ drop: the digit products of weights 128 and 1 the contraction leaves out, as an error of |xq - sq|^2)
            const double y = p.gamma * (p.delta * (2.0 * (sqrt(xq2) * (1.0 + 1e-15) + p.s_max) + p.delta) + p.drop);
            const double band = (y * (1.0 + y) * 1.01 + 9.1e-13) * p.guard_scale * S;
            decided = (y < 0.5) && (fabs(dv) > band);
        }
        dec_exact[sl] = dv;
        labels[evalcell[e]] = (int8_t)(dv > 0.0 ? p.gv0 : p.gv1);
        undecided = !decided;
        if (!decided && !words) {
            const int s2 = atomicAdd(&counters[CNT_FLAGGEDI], 1);
            if (s2 < flagi_cap) flagi_list[s2] = e;
        }
        }
        if (words) {
            const unsigned long long bal = __ballot(undecided);
            if (lane == 0) words[base >> 6] = bal;
        }
    }
}

__global__ __launch_bounds__(kListCompactThreads) void k_i8_handover(const unsigned long long *__restrict__ words, const int *__restrict__ flag_list,
                                                                     int flag_cap, int list_off, int *__restrict__ counters, int cslot,
                                                                     int *__restrict__ flagi_list, int flagi_cap)
{
    __shared__ int s_scan[kListCompactThreads];
    list_compact_body(words, window_count8(counters[cslot], list_off, flag_cap), flag_list, flagi_list, flagi_cap, counters, CNT_FLAGGEDI, s_scan);
}

// One window [list_off, list_off + window_cap) of the list counted by counters[CNT_FLAGGED]: digit image (the feature kernel), the
// int8 contraction, combine.  ximg holds kI8GroupBytes per 16 slots of a window, part64 the fp64 tier's [2 kMSplit + 1][window_cap].
void launch_recheck_i8(const float *ii, const int *evalcell, const FeatDesc *fd, const void *sv_i8, I8Params p, double lower, double upper,
                       const int *flag_list, int window_cap, int list_off, int *counters, void *ximg, double *part64, double *dec_exact,
                       int8_t *labels, int *flagi_list, int flagi_cap, Dims d, hipStream_t s, unsigned long long *words)
{
    const int groups = (window_cap + kI8Evals - 1) / kI8Evals;
    if (groups <= 0) return;
    flag_list += list_off;
    dec_exact += list_off;
    launch_features(ii, evalcell, counters, fd, reinterpret_cast<float *>(ximg), reinterpret_cast<float *>(part64 + (size_t)(2 * kI8Split) * window_cap),
                    d, lower, upper, 0.0f, window_cap, XMODE_I8, ScreenParams{}, flag_list, CNT_FLAGGED, window_cap, false, window_cap, nullptr,
                    nullptr, s, list_off);
    const long tasks = (long)groups * 4 * kI8Split;                     // (the kernel strides over the tasks the list really has)
    hipLaunchKernelGGL(k_recheck_i8, dim3((unsigned)(tasks < 8192 ? tasks : 8192)), dim3(256), 0, s, (const char *)ximg, (const char *)sv_i8, p,
                       window_cap, list_off, counters, CNT_FLAGGED, part64);
    const int blocks = groups < 2048 ? groups : 2048;
    hipLaunchKernelGGL(k_recheck_i8_combine, dim3(blocks), dim3(256), 0, s, part64, evalcell, p, flag_list, window_cap, list_off, counters,
                       CNT_FLAGGED, dec_exact, labels, flagi_list, flagi_cap, words);
    // (flag_list already points at the window's first entry)
    if (words) hipLaunchKernelGGL(k_i8_handover, dim3(1), dim3(kListCompactThreads), 0, s, words, flag_list, window_cap, list_off, counters, CNT_FLAGGED,
                                  flagi_list, flagi_cap);
}

}  // namespace haf
