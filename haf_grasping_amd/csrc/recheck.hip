// recheck.hip -- exact re-evaluation of what the contraction tiers could not decide:
//   k_recheck_mfma / k_recheck_combine   guard-band evaluations again as an fp64 MFMA contraction (tier 2)
//   k_recheck        what is still within 2^-40 of zero, in libsvm's exact fp64 summation order (tier 3; svm.cpp:327-364, 2509-2513)
//   k_recheck_terms / k_recheck_sum   the same order for a handful of evaluations spread over the chip
// (the exact-integer tier 2a is exact8.hip)
//
// Built with -ffp-contract=off: every fp32/fp64 expression that must match the CPU restatement bit for bit is
// written with explicit *_rn intrinsics as well; fma() is used only where a fused operation is intended.
#include "feature_device.h"

namespace haf {

// ---------------------------------------------------------------------------------------------------
// a8 exact: guard-band evaluations re-done in libsvm's own order: d2 summed over attributes in index order in
// fp64 without fusion (svm.cpp:327-364), K = exp(-gamma*d2), decision summed over SVs in model order (2509-2513).
// A workgroup takes kRB flagged evaluations at once: their attribute vectors sit in LDS (broadcast reads), each
// thread owns one support vector of the current 256-SV chunk and streams its fp64 column ONCE for all kRB
// evaluations (kRB-fold less L2 traffic than one evaluation per workgroup); the 256 products coef*K of a chunk go
// to LDS and one thread per evaluation adds them in model order, carrying the running sum across chunks.
// ---------------------------------------------------------------------------------------------------
constexpr int kRB = 16;
constexpr int kRChunk = 256;

__global__ __launch_bounds__(256) void k_recheck(const float *__restrict__ ii, const int *__restrict__ evalcell,
                                                 const FeatDesc *__restrict__ fd, const double *__restrict__ sv64,
                                                 const double *__restrict__ coef64, ExactParams p,
                                                 const int *__restrict__ flag_list, int flag_cap,
                                                 const int *__restrict__ counters, int counter_slot,
                                                 double *__restrict__ dec_exact, int8_t *__restrict__ labels, Dims d)
{
    __shared__ double xs[kRB][kKP];
    __shared__ double terms[kRB][kRChunk + 1];
    __shared__ double run_sum[kRB];
    int n_flag = counters[counter_slot];
    if (n_flag > flag_cap) n_flag = flag_cap;
    const int n_groups = (n_flag + kRB - 1) / kRB;
    const int H = d.H, W = d.W;
    const rsrc_t iir = make_ii_rsrc(ii, d);
    const int tid = threadIdx.x;
    for (int g = blockIdx.x; g < n_groups; g += gridDim.x) {
        for (int it = tid; it < kRB * p.kx; it += 256) {
            const int ev = it / p.kx, f = it - ev * p.kx;
            const int slot = g * kRB + ev;
            double x = 0.0;
            if (slot < n_flag && f < d.nf && !fd[f].skip)
                x = attribute_value(SrcBuf<false>{iir, window_origin(evalcell[flag_list[slot]], H, W)}, fd[f], p.lower, p.upper, hafq::GlobalTabs());
            xs[ev][f] = x;
        }
        if (tid < kRB) run_sum[tid] = 0.0;
        __syncthreads();
        for (int n0 = 0; n0 < p.n_sv; n0 += kRChunk) {
            const int n = n0 + tid;
            if (n < p.n_sv) {
                double sum[kRB];
#pragma unroll
                for (int ev = 0; ev < kRB; ev++) sum[ev] = 0.0;
                const double *col = sv64 + n;
                const double c = coef64[n];
                if (p.kernel_type == 2) {                                          // RBF (wave-uniform)
                for (int k = 0; k < p.kx; k++) {
                    const double s = col[(size_t)k * p.n_sv_pad];
#pragma unroll
                    for (int ev = 0; ev < kRB; ev++) {
                        double dd = __dsub_rn(xs[ev][k], s);
                        sum[ev] = __dadd_rn(sum[ev], __dmul_rn(dd, dd));          // svm.cpp:333-334, 342, 347
                    }
                }
#pragma unroll
                for (int ev = 0; ev < kRB; ev++) terms[ev][tid] = __dmul_rn(c, exp(__dmul_rn(-p.gamma, sum[ev])));
                } else {
                    // round 5: libsvm's other vector kernels (Kernel::k_function, svm.cpp:318-371), for models the fast tiers are not built
                    // for.  Kernel::dot (299-316): the products of the attributes both vectors carry, in index order; a zero on either side
                    // adds +-0, which leaves the sum as it is.  powi: svm.cpp:26-36.
                    for (int k = 0; k < p.kx; k++) {
                        const double s = col[(size_t)k * p.n_sv_pad];
#pragma unroll
                        for (int ev = 0; ev < kRB; ev++) sum[ev] = __dadd_rn(sum[ev], __dmul_rn(xs[ev][k], s));
                    }
#pragma unroll
                    for (int ev = 0; ev < kRB; ev++) {
                        double kv = sum[ev];                                       // LINEAR
                        if (p.kernel_type != 0) {
                            const double arg = __dadd_rn(__dmul_rn(p.gamma, sum[ev]), p.coef0);
                            if (p.kernel_type == 1) {                              // POLY: powi(gamma * dot + coef0, degree)
                                double tmp = arg, ret = 1.0;
                                for (int t = p.degree; t > 0; t /= 2) {
                                    if (t % 2 == 1) ret = __dmul_rn(ret, tmp);
                                    tmp = __dmul_rn(tmp, tmp);
                                }
                                kv = ret;
                            } else {
                                kv = tanh(arg);                                    // SIGMOID (last bit: the host's tanh decides near zero, engine_request.cpp)
                            }
                        }
                        terms[ev][tid] = __dmul_rn(c, kv);
                    }
                }
            }
            __syncthreads();
            if (tid < kRB) {
                double s = run_sum[tid];
                const int cnt = min(kRChunk, p.n_sv - n0);
                for (int q = 0; q < cnt; q++) s = __dadd_rn(s, terms[tid][q]);    // model order (2509-2512)
                run_sum[tid] = s;
            }
            __syncthreads();
        }
        if (tid < kRB) {
            const int slot = g * kRB + tid;
            if (slot < n_flag) {
                const double dv = __dsub_rn(run_sum[tid], p.rho);                 // 2513
                dec_exact[slot] = dv;
                labels[evalcell[flag_list[slot]]] = (int8_t)(dv > 0.0 ? p.gv0 : p.gv1);
            }
        }
        __syncthreads();
    }
}

// The same tier when the HOST knows the list's length (it does whenever the tier runs behind the others: the counters came back with
// the roll records) -- spread out: a workgroup per (group of kRB evaluations, chunk of kRChunk support vectors) writes the products
// coef_n K_n to a scratch row per evaluation, then a workgroup per evaluation adds them IN MODEL ORDER (one thread, chunk by chunk
// through LDS).  Every operation and the order of the sum are k_recheck's; only who computes which K differs.  A model of 8964 SVs
// kept one workgroup busy for 4.3 ms (35 chunks one after the other) for the four evaluations a C5 request leaves within 2^-40 S
// of zero; spread over 35 workgroups + the sum it is ~0.15 ms.
__global__ __launch_bounds__(256) void k_recheck_terms(const float *__restrict__ ii, const int *__restrict__ evalcell,
                                                       const FeatDesc *__restrict__ fd, const double *__restrict__ sv64,
                                                       const double *__restrict__ coef64, ExactParams p,
                                                       const int *__restrict__ flag_list, int list_off, int n_win,
                                                       double *__restrict__ terms_out, Dims d)
{
    __shared__ double xs[kRB][kKP];
    const int H = d.H, W = d.W;
    const rsrc_t iir = make_ii_rsrc(ii, d);
    const int tid = threadIdx.x, g = blockIdx.y, n0 = blockIdx.x * kRChunk;
    for (int it = tid; it < kRB * p.kx; it += 256) {
        const int ev = it / p.kx, f = it - ev * p.kx;
        const int slot = g * kRB + ev;
        double x = 0.0;
        if (slot < n_win && f < d.nf && !fd[f].skip)
            x = attribute_value(SrcBuf<false>{iir, window_origin(evalcell[flag_list[list_off + slot]], H, W)}, fd[f], p.lower, p.upper, hafq::GlobalTabs());
        xs[ev][f] = x;
    }
    __syncthreads();
    const int n = n0 + tid;
    if (n >= p.n_sv) return;
    double sum[kRB];
#pragma unroll
    for (int ev = 0; ev < kRB; ev++) sum[ev] = 0.0;
    const double *col = sv64 + n;
    for (int k = 0; k < p.kx; k++) {
        const double sv = col[(size_t)k * p.n_sv_pad];
#pragma unroll
        for (int ev = 0; ev < kRB; ev++) {
            double dd = __dsub_rn(xs[ev][k], sv);
            sum[ev] = __dadd_rn(sum[ev], __dmul_rn(dd, dd));          // svm.cpp:333-334, 342, 347
        }
    }
    const double c = coef64[n];
#pragma unroll
    for (int ev = 0; ev < kRB; ev++) {
        const int slot = g * kRB + ev;
        if (slot < n_win) terms_out[(size_t)slot * p.n_sv_pad + n] = __dmul_rn(c, exp(__dmul_rn(-p.gamma, sum[ev])));
    }
}

__global__ __launch_bounds__(256) void k_recheck_sum(const double *__restrict__ terms_in, ExactParams p, const int *__restrict__ evalcell,
                                                     const int *__restrict__ flag_list, int list_off, int n_win,
                                                     double *__restrict__ dec_exact, int8_t *__restrict__ labels)
{
    __shared__ double t[kRChunk];
    const int slot = blockIdx.x, tid = threadIdx.x;
    if (slot >= n_win) return;
    const double *row = terms_in + (size_t)slot * p.n_sv_pad;
    double s = 0.0;
    for (int n0 = 0; n0 < p.n_sv; n0 += kRChunk) {
        if (n0 + tid < p.n_sv) t[tid] = row[n0 + tid];
        __syncthreads();
        if (tid == 0) {
            const int cnt = min(kRChunk, p.n_sv - n0);
            for (int q = 0; q < cnt; q++) s = __dadd_rn(s, t[q]);      // model order (2509-2512)
        }
        __syncthreads();
    }
    if (tid == 0) {
        const double dv = __dsub_rn(s, p.rho);                         // 2513
        dec_exact[list_off + slot] = dv;
        labels[evalcell[flag_list[list_off + slot]]] = (int8_t)(dv > 0.0 ? p.gv0 : p.gv1);
    }
}

void launch_recheck_known(const float *ii, const int *evalcell, const FeatDesc *fd, const double *sv64, const double *coef64,
                          ExactParams p, const int *flag_list, int n_flag, double *terms, int terms_slots,
                          double *dec_exact, int8_t *labels, Dims d, hipStream_t s)
{
    const int chunks = (p.n_sv + kRChunk - 1) / kRChunk;
    for (int off = 0; off < n_flag; off += terms_slots) {
        const int n_win = std::min(terms_slots, n_flag - off);
        hipLaunchKernelGGL(k_recheck_terms, dim3((unsigned)chunks, (unsigned)((n_win + kRB - 1) / kRB)), dim3(256), 0, s, ii, evalcell, fd, sv64,
                           coef64, p, flag_list, off, n_win, terms, d);
        hipLaunchKernelGGL(k_recheck_sum, dim3((unsigned)n_win), dim3(256), 0, s, terms, p, evalcell, flag_list, off, n_win, dec_exact, labels);
    }
}

void launch_recheck(const float *ii, const int *evalcell, const FeatDesc *fd, const double *sv64, const double *coef64,
                    ExactParams p, const int *flag_list, int flag_cap, const int *counters, int counter_slot,
                    double *dec_exact, int8_t *labels, Dims d, hipStream_t s)
{
    int groups = (flag_cap + kRB - 1) / kRB;
    int blocks = groups < 2048 ? groups : 2048;
    if (blocks <= 0) return;
    hipLaunchKernelGGL(k_recheck, dim3(blocks), dim3(256), 0, s, ii, evalcell, fd, sv64, coef64, p, flag_list, flag_cap,
                       counters, counter_slot, dec_exact, labels, d);
}

// ---------------------------------------------------------------------------------------------------
// a8, middle tier: the guard-band evaluations of the fp32/fp16 contraction re-done as an fp64 MFMA contraction
// (v_mfma_f64_16x16x4_f64, GEMM form, fp64 exp).  Its error is ~2^-44 of sum|coef|K, so only evaluations with
// |dec| <= 2^-40 * T * sum|coef|K (practically none) still need libsvm's strict summation order (k_recheck).
// Workgroup = 4 waves x 16 flagged evaluations; each wave keeps its 16 x 324 fp64 attributes as the A operand in 162
// VGPRs (loaded from the image the feature kernels write in their XMODE_F64 form), the fp64 SV tile
// (326 x 16: attributes, |s|^2, coef) is shared through LDS, double buffered with a register-staged prefetch.
// ---------------------------------------------------------------------------------------------------
constexpr int kMWaves = 4;
constexpr int kMSplit = 8;                        // SV ranges a group of evaluations is split over (k_recheck_mfma tasks)
static_assert(kRecheckPartRows == 2 * kMSplit + 1, "part64 layout");
constexpr int kMEvals = 16 * kMWaves;
constexpr int kMSteps = kKP / 4;                 // 81 k-steps of 4
constexpr int kMTileDoubles = kM64Rows * 16;     // 5216 doubles = 41728 B
constexpr int kMLoads = (kMTileDoubles / 2 + 255) / 256;   // 16-byte loads per thread per tile (11)

// How many ranges of SV tiles a group of 64 evaluations is split over, and the row pitch of part64: kMSplit ranges and one row per
// list slot of the window -- or, when the list fills at most a quarter of the window (the rule at C5: a few thousand entries of a
// window of two million), FOUR times as many ranges on a quarter of the pitch (65 rows x flag_cap / 4 fit the 17 x flag_cap doubles
// of the buffer): a task's chain of dependent MFMAs and tile hand-overs is a quarter as long and four times as many CUs have one,
// which is what a short list needs (C5, 383 entries: 108 -> 35 us).  Both kernels of the tier derive the same numbers from the
// same counter, so the partial sums are added in a fixed order for a given list length.
__device__ __forceinline__ void recheck_split(int n_flag, int flag_cap, int n_tiles, int &splits, size_t &pitch)
{
    // (only while the coarse split leaves half of the CUs without a task: from ~1 500 entries on the tier is bound by the fp64 matrix
    // rate -- 1 574 entries x 4096 SVs are 54 us at its peak -- and finer tasks only reload the A operands: 118 -> 131 us measured)
    const bool fine = (long)n_flag * 4 <= (long)flag_cap && n_tiles >= 16 * kMSplit && flag_cap >= 4 && n_flag <= 1024;
    // (round 5: as many ranges as give every CU ONE task -- the kernel's 290 registers leave a CU one workgroup, and 9 groups x 32
    // ranges were 288 tasks for 256 CUs: two rounds, 125 us for C3's 557 entries against the 8 964-SV model)
    const int n_groups = (n_flag + kMEvals - 1) / kMEvals;
    splits = fine ? max(kMSplit, min(4 * kMSplit, 256 / max(n_groups, 1))) : kMSplit;
    pitch = fine ? (size_t)(flag_cap / 4) : (size_t)flag_cap;
}

// The fp64 attribute image of the flagged evaluations -- [group of 16][324][16] doubles, the register image of the fp64 MFMA A
// operand (64 consecutive doubles per k-step) -- is written by the feature kernels in their XMODE_F64 form (list mode, windows
// staged in LDS).  Round 2 had a kernel of its own for it (one thread per (evaluation, attribute), per-lane descriptors and
// corner loads): 19 ns per evaluation against 3 ns here.
__global__ __launch_bounds__(256) void k_recheck_mfma(const double *__restrict__ x64, const int *__restrict__ evalcell,
                                                      const double *__restrict__ sv64,
                                                      ExactParams p, const int *__restrict__ flag_list, int flag_cap,
                                                      int list_off, int *__restrict__ counters, double *__restrict__ part64, Dims d,
                                                      int cslot)
{
    // ONE SV tile in LDS (41 KiB): the next tile waits in registers while this one is consumed, and both barriers of the
    // hand-over are needed with one buffer or two -- with one, three workgroups fit a CU instead of one
    __shared__ __attribute__((aligned(16))) double bt[1][kMTileDoubles];
    __shared__ double xxs[kMWaves][16];
    const int n_flag = window_count(counters[cslot], list_off, flag_cap);
    const int n_groups = (n_flag + kMEvals - 1) / kMEvals;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int n_tiles = p.n_sv_pad / 16;
    typedef double double2_t __attribute__((ext_vector_type(2)));

    // a task = (group of 64 evaluations, one of kMSplit ranges of SV tiles): a few thousand flagged evaluations would
    // otherwise occupy a fraction of the CUs for the full length of the model; k_recheck_combine adds the partial sums in
    // a fixed order
    int splits;
    size_t pitch;
    recheck_split(n_flag, flag_cap, n_tiles, splits, pitch);
    const int tiles_per_part = (n_tiles + splits - 1) / splits;
    for (int task = blockIdx.x; task < n_groups * splits; task += gridDim.x) {
        const int g = task / splits, h = task - g * splits;
        const int t_begin = h * tiles_per_part, t_end = min(n_tiles, t_begin + tiles_per_part);
        // ---- A operand: lane holds x[eval lane&15][k = 4s + (lane>>4)], s = 0..80, from the XMODE_F64 image ----
        const int grp = g * kMWaves + wave;                          // 16 flagged evaluations
        double a[kMSteps];
        double xxp = 0.0;
        {
            const double *xg = x64 + (size_t)grp * kKP * 16 + lane;  // [(grp*324 + 4s + (lane>>4))*16 + (lane&15)]
#pragma unroll
            for (int s = 0; s < kMSteps; s++) {
                a[s] = xg[s * 64];
                xxp = fma(a[s], a[s], xxp);
            }
        }
        xxp += __shfl_xor(xxp, 16, 64);
        xxp += __shfl_xor(xxp, 32, 64);
        if ((lane >> 4) == 0) xxs[wave][lane & 15] = xxp;

        // ---- SV tiles ----
        double2_t pre[kMLoads];
        auto tile_load = [&](int t) {
#pragma unroll
            for (int q = 0; q < kMLoads; q++) {
                const int idx = tid + q * 256;               // pair index: row = idx / 8, column pair = idx % 8
                if (idx < kMTileDoubles / 2)
                    pre[q] = *reinterpret_cast<const double2_t *>(sv64 + (size_t)(idx >> 3) * p.n_sv_pad + t * 16 + (idx & 7) * 2);
            }
        };
        auto tile_store = [&](int buf) {
#pragma unroll
            for (int q = 0; q < kMLoads; q++) {
                const int idx = tid + q * 256;
                if (idx < kMTileDoubles / 2) *reinterpret_cast<double2_t *>(&bt[buf][idx * 2]) = pre[q];
            }
        };
        if (t_begin < t_end) { tile_load(t_begin); tile_store(0); }
        __syncthreads();

        double part[4] = {0, 0, 0, 0}, pabs[4] = {0, 0, 0, 0};
        for (int t = t_begin; t < t_end; t++) {
            const double *B = bt[0];
            if (t + 1 < t_end) tile_load(t + 1);
            f64x4 acc = {0, 0, 0, 0};
#pragma unroll
            for (int s = 0; s < kMSteps; s++) {
                const double b = B[(4 * s + (lane >> 4)) * 16 + (lane & 15)];      // B[k = 4s + (lane>>4)][j = lane&15]
                acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[s], b, acc, 0, 0, 0);
            }
            const double ss = B[kKP * 16 + (lane & 15)];
            const double cf = B[(kKP + 1) * 16 + (lane & 15)];
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int row = (lane >> 4) + 4 * r;                               // f64 C/D layout: row = (lane>>4) + 4*reg
                const double d2 = fma(-2.0, acc[r], xxs[wave][row] + ss);
                const double kv = exp(-p.gamma * d2);
                part[r] = fma(cf, kv, part[r]);
                pabs[r] = fma(fabs(cf), kv, pabs[r]);
            }
            __syncthreads();                          // everyone is done reading the tile
            if (t + 1 < t_end) tile_store(0);
            __syncthreads();
        }
#pragma unroll
        for (int r = 0; r < 4; r++) {
            double v = part[r], w = pabs[r];
            v += __shfl_xor(v, 8, 64); w += __shfl_xor(w, 8, 64);
            v += __shfl_xor(v, 4, 64); w += __shfl_xor(w, 4, 64);
            v += __shfl_xor(v, 2, 64); w += __shfl_xor(w, 2, 64);
            v += __shfl_xor(v, 1, 64); w += __shfl_xor(w, 1, 64);
            part[r] = v; pabs[r] = w;
        }
        // lane with (lane&15)==0 of 16-lane group q holds rows q + 4r: partial sums of this SV range (+ |x|^2 once)
        if ((lane & 15) == 0) {
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int row = (lane >> 4) + 4 * r;
                const int sl = g * kMEvals + wave * 16 + row;
                if (sl < n_flag) {
                    part64[(size_t)(2 * h) * pitch + sl] = part[r];
                    part64[(size_t)(2 * h + 1) * pitch + sl] = pabs[r];
                    if (h == 0) part64[(size_t)(2 * splits) * pitch + sl] = xxs[wave][row];
                }
            }
        }
        __syncthreads();
    }
}

// sum of the kMSplit partial decision values of every flagged evaluation (fixed order), label, and what is still too
// close to zero for the GEMM form goes on to the strict-order kernel
__global__ __launch_bounds__(256) void k_recheck_combine(const double *__restrict__ part64, const int *__restrict__ evalcell,
                                                         ExactParams p, const int *__restrict__ flag_list, int flag_cap,
                                                         int list_off, int *__restrict__ counters, double *__restrict__ dec_exact,
                                                         int8_t *__restrict__ labels, int *__restrict__ flag2_list, int flag2_cap,
                                                         int cslot, unsigned long long *__restrict__ words)
{
    const int n_flag = window_count(counters[cslot], list_off, flag_cap);
    int splits;
    size_t pitch;
    recheck_split(n_flag, flag_cap, p.n_sv_pad / 16, splits, pitch);
    const int lane = threadIdx.x & 63;
    // (wave-uniform trip count; words != nullptr: ordered hand-over to the strict tier's list, device_common.h)
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
        dec_exact[sl] = dv;
        const int e = flag_list[sl];
        labels[evalcell[e]] = (int8_t)(dv > 0.0 ? p.gv0 : p.gv1);
        const double T = p.as_max1 + p.gamma2 * part64[(size_t)(2 * splits) * pitch + sl];
        undecided = !(fabs(dv) > p.guard2 * T * S);
        if (undecided && !words) {
            int s2 = atomicAdd(&counters[CNT_FLAGGED2], 1);
            if (s2 < flag2_cap) flag2_list[s2] = e;
        }
        }
        if (words) {
            const unsigned long long bal = __ballot(undecided);
            if (lane == 0) words[base >> 6] = bal;
        }
    }
}

__global__ __launch_bounds__(kListCompactThreads) void k_strict_handover(const unsigned long long *__restrict__ words, const int *__restrict__ flag_list,
                                                                         int flag_cap, int list_off, int *__restrict__ counters, int cslot,
                                                                         int *__restrict__ flag2_list, int flag2_cap)
{
    __shared__ int s_scan[kListCompactThreads];
    list_compact_body(words, window_count(counters[cslot], list_off, flag_cap), flag_list, flag2_list, flag2_cap, counters, CNT_FLAGGED2, s_scan);
}

// flag_list and dec_exact are the WHOLE lists (one entry per flagged evaluation, sized for every evaluation of a request);
// the launch works on the window [list_off, list_off + window_cap) of them, which is what x64 / part64 are sized for.  The
// host runs window 0 with every request and further windows only when more evaluations were flagged than one window holds.
void launch_recheck_mfma(const float *ii, const int *evalcell, const FeatDesc *fd, const double *sv64, ExactParams p,
                         const int *flag_list, int window_cap, int list_off, int *counters, double *x64, double *part64,
                         double *dec_exact, int8_t *labels, int *flag2_list, int flag2_cap, Dims d, hipStream_t s, AttrRecord *dbg,
                         bool have_x64, int counter_slot, unsigned long long *words)
{
    int groups = (window_cap + kMEvals - 1) / kMEvals;
    int blocks = groups < 2048 ? groups : 2048;
    if (blocks <= 0) return;
    flag_list += list_off;
    dec_exact += list_off;
    // (have_x64: a request that went straight to this tier -- its feature kernel wrote the image for the identity list)
    if (!have_x64)
        launch_features(ii, evalcell, counters, fd, reinterpret_cast<float *>(x64), nullptr, d, p.lower, p.upper, 0.0f, window_cap, XMODE_F64,
                        ScreenParams{}, flag_list, counter_slot, window_cap, false, window_cap, dbg, nullptr, s, list_off);
    const long tasks = (long)groups * 4 * kMSplit;                      // (the kernel strides over the tasks the list really has)
    hipLaunchKernelGGL(k_recheck_mfma, dim3((unsigned)(tasks < 4096 ? tasks : 4096)), dim3(256), 0, s, x64, evalcell, sv64, p,
                       flag_list, window_cap, list_off, counters, part64, d, counter_slot);
    hipLaunchKernelGGL(k_recheck_combine, dim3(blocks), dim3(256), 0, s, part64, evalcell, p, flag_list, window_cap, list_off, counters,
                       dec_exact, labels, flag2_list, flag2_cap, counter_slot, words);
    if (words) hipLaunchKernelGGL(k_strict_handover, dim3(1), dim3(kListCompactThreads), 0, s, words, flag_list, window_cap, list_off, counters,
                                  counter_slot, flag2_list, flag2_cap);
}

}  // namespace haf
