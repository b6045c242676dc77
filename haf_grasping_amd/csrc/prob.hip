// prob.hip -- probability-output mode (HAF_FLAG_PROBABILITY).
//
// What the reference does with svm_with_probability (server.cpp:383 passes false, so this branch is dead there; it is restated
// because §8 lists it): predict_bestgp_withsvm runs "svm-predict -b 1" (791), whose output is a "labels a b" header and one
// line "label p(a) p(b)" per feature row ("%g" each, svm-predict.c:60-64, 111-118); show_predicted_gps (815-816, 824-848) reads
// ONE line before its loops and one more after every masked cell, so the k-th masked cell (row-major) is filled from line k of
// the file: the header for the first cell, the prediction of the masked cell BEFORE it for every other cell, the last
// prediction for nobody.  From a line it takes res = (int)atof(first two characters) and the first number behind the label, or
// the second when res > 0, as a float; the cell holds res*prob.  The vote (865-880) is then an fp32 sum of 29 products in source
// order, topval_gp stays an int (every assignment truncates), and the run centring (904-932) looks for cells EQUAL to that int.
//
// Decision values come from the strict tier (k_recheck: libsvm's own summation order) for every evaluation: this mode is about
// completeness, not speed.  The probability estimates are svm_predict_probability (svm.cpp:2550-2587) operation for operation in
// fp64 (sigmoid_predict 1818-1826, the [1e-7, 1 - 1e-7] clamp, multiclass_probability 1829-1888 for two classes); what crosses
// into the grid is their "%g" text form (decq, six significant digits), so the one operation that is not bit-pinned to glibc,
// exp(), matters only when an estimate sits within an ulp of a decimal rounding boundary.
#include <algorithm>

#include "decq.h"
#include "kernels.h"

namespace haf {

// every evaluation onto a tier's list (the strict tier's in the probability branch; the fp64 MFMA tier's for a tiny request)
__global__ __launch_bounds__(256) void k_prob_list(int *__restrict__ counters, int slot, int *__restrict__ list, int cap)
{
    const int n = min(counters[CNT_EVALS], cap);
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) list[i] = i;
    if (blockIdx.x == 0 && threadIdx.x == 0) counters[slot] = n;
}

// `bend` multiplies the exp() result: 1 in the estimate itself; 1 +- 2^-50 where k_prob_eval asks whether a last-bit difference
// between this exp and the C library's could move a printed digit
__device__ __forceinline__ double sigmoid_predict(double dec, double A, double B, double bend = 1.0)          // svm.cpp:1818-1826
{
    const double fApB = __dadd_rn(__dmul_rn(dec, A), B);
    if (fApB >= 0.0) {
        const double t = exp(-fApB) * bend;
        return __ddiv_rn(t, __dadd_rn(1.0, t));
    }
    return __ddiv_rn(1.0, __dadd_rn(1.0, exp(fApB) * (2.0 - bend)));      // (1 / bend to first order: a larger bend raises the estimate in both branches)
}

// multiclass_probability (svm.cpp:1829-1888) for k = 2: the same loops, the same operations in the same order
__device__ __forceinline__ void multiclass_probability2(const double r[2][2], double p[2])
{
    const int k = 2;
    double Q[2][2], Qp[2], pQp;
    const double eps = 0.005 / k;
    for (int t = 0; t < k; t++) {
        p[t] = 1.0 / k;
        Q[t][t] = 0.0;
        for (int j = 0; j < t; j++) { Q[t][t] = __dadd_rn(Q[t][t], __dmul_rn(r[j][t], r[j][t])); Q[t][j] = Q[j][t]; }
        for (int j = t + 1; j < k; j++) { Q[t][t] = __dadd_rn(Q[t][t], __dmul_rn(r[j][t], r[j][t])); Q[t][j] = __dmul_rn(-r[j][t], r[t][j]); }
    }
    for (int iter = 0; iter < 100; iter++) {
        pQp = 0.0;
        for (int t = 0; t < k; t++) {
            Qp[t] = 0.0;
            for (int j = 0; j < k; j++) Qp[t] = __dadd_rn(Qp[t], __dmul_rn(Q[t][j], p[j]));
            pQp = __dadd_rn(pQp, __dmul_rn(p[t], Qp[t]));
        }
        double max_error = 0.0;
        for (int t = 0; t < k; t++) {
            const double error = fabs(__dsub_rn(Qp[t], pQp));
            if (error > max_error) max_error = error;
        }
        if (max_error < eps) break;
        for (int t = 0; t < k; t++) {
            const double diff = __ddiv_rn(__dadd_rn(-Qp[t], pQp), Q[t][t]);
            p[t] = __dadd_rn(p[t], diff);
            const double one_d = __dadd_rn(1.0, diff);
            const double inner = __dadd_rn(__dmul_rn(diff, Q[t][t]), __dmul_rn(2.0, Qp[t]));
            pQp = __ddiv_rn(__ddiv_rn(__dadd_rn(pQp, __dmul_rn(diff, inner)), one_d), one_d);
            for (int j = 0; j < k; j++) {
                Qp[j] = __ddiv_rn(__dadd_rn(Qp[j], __dmul_rn(diff, Q[t][j])), one_d);
                p[j] = __ddiv_rn(p[j], one_d);
            }
        }
    }
}

// svm_predict_probability for two classes from a decision value: p[0], p[1] and the index of the first maximum
__device__ __forceinline__ int prob_estimates(double dec, const ProbParams &P, double bend, double p[2])
{
    const double min_prob = 1e-7;
    double s = sigmoid_predict(dec, P.A, P.B, bend);
    s = s > min_prob ? s : min_prob;                                          // max(.., min_prob)       svm.cpp:2569
    const double hi = __dsub_rn(1.0, min_prob);
    s = s < hi ? s : hi;                                                      // min(.., 1 - min_prob)
    double r[2][2] = {{0.0, s}, {__dsub_rn(1.0, s), 0.0}};
    multiclass_probability2(r, p);
    return p[1] > p[0] ? 1 : 0;                                               // first maximum (2575-2578)
}

// per evaluation: probability estimates, the label svm_predict_probability returns, the "%g" forms, and the value a grid cell
// would take from this evaluation's output line.  Round 4: the device's exp is not glibc's -- both are within an
// ulp, and so are the two libsvm-order decision values (2^-52 sum|coef|, `dec_slack` carries 4x that).  The estimate is therefore
// formed three times -- as is, and with the decision value and the exp result pushed to either side by more than the two
// libraries can differ -- and where the label or a printed digit is not the same in all three, the evaluation goes on the list of
// those the HOST finishes with the C library's exp (engine.cpp: host_resolve_probability).
__global__ __launch_bounds__(256) void k_prob_eval(const double *__restrict__ dec_exact, const int *__restrict__ evalcell,
                                                   const int *__restrict__ counters, ProbParams P, int8_t *__restrict__ labels,
                                                   float *__restrict__ own, double *__restrict__ ptext,
                                                   int *__restrict__ near_list, int near_cap, int *__restrict__ counters_rw)
{
    const int n = counters[CNT_EVALS];
    for (int e = blockIdx.x * 256 + threadIdx.x; e < n; e += gridDim.x * 256) {
        double p[2];
        const double dec = dec_exact[e];
        const int idx = prob_estimates(dec, P, 1.0, p);
        const double q0 = hafq::decq(p[0], 6), q1 = hafq::decq(p[1], 6);      // " %g" (svm-predict.c:116), read back by atof
        if (near_list) {
            bool near = P.host_all != 0;
            const double bend = 8.8817841970012523e-16;                       // 2^-50
            for (int side = 0; side < 2 && !near; side++) {
                double pb[2];
                const double sgn = side ? -1.0 : 1.0;
                // (the estimate is monotone in the decision value and in the exp result: the two extreme combinations bracket it)
                const int ib = prob_estimates(dec + sgn * P.dec_slack * (P.A < 0.0 ? 1.0 : -1.0), P, 1.0 + sgn * bend, pb);
                near = ib != idx || hafq::decq(pb[0], 6) != q0 || hafq::decq(pb[1], 6) != q1;
            }
            if (near) {
                const int slot = atomicAdd(&counters_rw[CNT_FLAGGED], 1);
                if (slot < near_cap) near_list[slot] = e;
            }
        }
        const int res = idx ? P.gv1 : P.gv0;                                  // (int)atof(line.substr(0,2))  server.cpp:833
        const float prob = (float)(res > 0 ? q1 : q0);                        // 834-840
        const int cell = evalcell[e];
        own[cell] = (float)res * prob;
        labels[cell] = (int8_t)res;
        ptext[2 * (size_t)e] = q0;
        ptext[2 * (size_t)e + 1] = q1;
    }
}

// the grid show_predicted_gps builds: -1 where unmasked (828-829); a masked cell takes the value of the masked cell before it
// in row-major order, the first one what the "labels" header parses to (`hdr`).  A thread walks one grid row; what enters the
// row from the left is the last masked cell of the nearest non-empty row above.
__global__ __launch_bounds__(256) void k_prob_grid(const uint8_t *__restrict__ mask, const int *__restrict__ rowcount,
                                                   const float *__restrict__ own, float *__restrict__ gridf, float hdr, Dims d)
{
    const int br = blockIdx.x, H = d.H, W = d.W;
    const size_t base = (size_t)br * H * W;
    for (int r = threadIdx.x; r < H; r += 256) {
        int pr = r - 1;
        while (pr >= 0 && rowcount[br * H + pr] == 0) pr--;
        float prev = hdr;
        if (pr >= 0) {
            int c = W - 1;
            while (c > 0 && !mask[base + (size_t)pr * W + c]) c--;
            prev = own[base + (size_t)pr * W + c];
        }
        for (int c = 0; c < W; c++) {
            const size_t cell = base + (size_t)r * W + c;
            if (mask[cell]) { gridf[cell] = prev; prev = own[cell]; }
            else gridf[cell] = -1.0f;
        }
    }
}

// server.cpp:865-880 on float cells: 29 fp32 products summed left to right as the source spells them
__global__ __launch_bounds__(256) void k_prob_vote_cells(const float *__restrict__ gridf, float *__restrict__ evf, Dims d)
{
    const int H = d.H, W = d.W;
    const size_t n = (size_t)d.B * d.R * H * W;
    for (size_t k = (size_t)blockIdx.x * 256 + threadIdx.x; k < n; k += (size_t)gridDim.x * 256) {
        const int col = (int)(k % W), row = (int)((k / W) % H);
        const float *g = gridf + k;
        float v = 0.0f;
        if (!(g[0] < 0.0f || row < 2 || row >= H - 2 || col < 4 || col >= W - 4)) {
#define HAF_G(dr, dc) g[(dr) * W + (dc)]
#define HAF_T(w, dr, dc) __fmul_rn((float)(w), HAF_G(dr, dc))
            v = HAF_T(1, -2, -2);
            v = __fadd_rn(v, HAF_T(2, -2, -1)); v = __fadd_rn(v, HAF_T(3, -2, 0)); v = __fadd_rn(v, HAF_T(2, -2, 1)); v = __fadd_rn(v, HAF_T(1, -2, 2));
            v = __fadd_rn(v, HAF_T(2, -1, -2)); v = __fadd_rn(v, HAF_T(3, -1, -1)); v = __fadd_rn(v, HAF_T(4, -1, 0)); v = __fadd_rn(v, HAF_T(3, -1, 1)); v = __fadd_rn(v, HAF_T(2, -1, 2));
            v = __fadd_rn(v, HAF_T(2, 0, -4)); v = __fadd_rn(v, HAF_T(2, 0, -3)); v = __fadd_rn(v, HAF_T(3, 0, -2)); v = __fadd_rn(v, HAF_T(4, 0, -1)); v = __fadd_rn(v, HAF_T(55, 0, 0));
            v = __fadd_rn(v, HAF_T(4, 0, 1)); v = __fadd_rn(v, HAF_T(3, 0, 2)); v = __fadd_rn(v, HAF_T(2, 0, 3)); v = __fadd_rn(v, HAF_T(2, 0, 4));
            v = __fadd_rn(v, HAF_T(2, 1, -2)); v = __fadd_rn(v, HAF_T(3, 1, -1)); v = __fadd_rn(v, HAF_T(4, 1, 0)); v = __fadd_rn(v, HAF_T(3, 1, 1)); v = __fadd_rn(v, HAF_T(2, 1, 2));
            v = __fadd_rn(v, HAF_T(1, 2, -2)); v = __fadd_rn(v, HAF_T(2, 2, -1)); v = __fadd_rn(v, HAF_T(3, 2, 0)); v = __fadd_rn(v, HAF_T(2, 2, 1)); v = __fadd_rn(v, HAF_T(1, 2, 2));
#undef HAF_T
#undef HAF_G
        }
        evf[k] = v;
    }
}

// server.cpp:866-894 and 904-932 for one (cloud, roll), without the sequential loop: with T the running topval_gp (an int,
// starting at -1000), a cell replaces it iff v > T, and then T = (int)v >= the old T: T is the running maximum of the truncated
// votes.  So topval_gp ends as M = max (int)v, and the cell the first loop ends on is the LAST cell that exceeded the T of its
// moment: the first cell whose truncation reaches M, or a later cell with v > M (a fraction above it).  The second loop moves
// the result to the centre of the first longest run of cells EQUAL to M, if there is any such cell.
__global__ __launch_bounds__(1024) void k_prob_pick(const float *__restrict__ evf, const float *__restrict__ heights,
                                                    const int *__restrict__ brcount, RollRecordDev *__restrict__ rec, Dims d)
{
    __shared__ long long red[1024];
    __shared__ long long s_val;
    const int br = blockIdx.x, t = threadIdx.x, H = d.H, W = d.W, HW = H * W;
    const float *v = evf + (size_t)br * HW;
    auto reduce_max = [&](long long x) -> long long {
        red[t] = x;
        __syncthreads();
        for (int o = 512; o > 0; o >>= 1) {
            if (t < o && red[t + o] > red[t]) red[t] = red[t + o];
            __syncthreads();
        }
        if (t == 0) s_val = red[0];
        __syncthreads();
        const long long out = s_val;
        __syncthreads();
        return out;
    };
    long long m = -1000;
    for (int k = t; k < HW; k += 1024) m = max(m, (long long)(int)v[k]);
    const int M = (int)reduce_max(m);
    const float Mf = (float)M;
    long long k1 = -1, k2 = -1;                   // k1: first cell with (int)v == M, kept as a max of (HW - 1 - k)
    for (int k = t; k < HW; k += 1024) {
        const float x = v[k];
        if ((int)x == M) k1 = max(k1, (long long)(HW - 1 - k));
        if (x > Mf) k2 = max(k2, (long long)k);
    }
    k1 = reduce_max(k1);
    k2 = reduce_max(k2);
    long long run = 0;                             // (length << 40) | (0xFFFFF - row) << 20 | last column of the run
    for (int r = t; r < H; r += 1024) {
        int cur = 0, best = 0, best_end = 0;
        for (int c = 0; c < W; c++) {
            if (v[r * W + c] == Mf) { cur++; if (cur > best) { best = cur; best_end = c; } }
            else cur = 0;
        }
        if (best > 0) run = max(run, ((long long)best << 40) | ((long long)(0xFFFFF - r) << 20) | best_end);
    }
    run = reduce_max(run);
    if (t == 0) {
        int row = -1, col = -1;
        if (run > 0) {
            const int len = (int)(run >> 40);
            row = 0xFFFFF - (int)((run >> 20) & 0xFFFFF);
            col = (int)(run & 0xFFFFF) - len / 2;
        } else if (k1 >= 0) {
            const int first = HW - 1 - (int)k1, idx = k2 > first ? (int)k2 : first;
            row = idx / W;
            col = idx % W;
        }
        float h_locmax = -10.0f;                                            // 1342-1351
        if (row >= 0)
            for (int rz = -4; rz < 5; rz++)
                for (int cz = -4; cz < 4; cz++) {
                    const int rr = row + rz, cc = col + cz;
                    if (rr >= 0 && cc >= 0 && rr < H && cc < W) {
                        const float h = heights[(size_t)br * HW + rr * W + cc];
                        if (h_locmax < h) h_locmax = h;
                    }
                }
        RollRecordDev r;
        r.vote = M; r.row = (short)row; r.col = (short)col; r.h_locmax = h_locmax; r.n_evals = brcount[br];
        rec[br] = r;
    }
}

void launch_probability_eval(const double *dec_exact, const int *evalcell, const int *counters, ProbParams P, int8_t *labels, float *own,
                             double *ptext, int *near_list, int near_cap, int *counters_rw, long evals_cap, hipStream_t s)
{
    const int eb = (int)std::min<long>(4096, (evals_cap + 255) / 256);
    if (eb > 0) hipLaunchKernelGGL(k_prob_eval, dim3(eb), dim3(256), 0, s, dec_exact, evalcell, counters, P, labels, own, ptext, near_list,
                                   near_cap, counters_rw);
}

void launch_probability(const double *dec_exact, const int *evalcell, const int *counters, ProbParams P, int8_t *labels,
                        const uint8_t *mask, const int *rowcount, const int *brcount, const float *heights, float *own,
                        double *ptext, float *gridf, float *evf, RollRecordDev *rec, long evals_cap, Dims d, hipStream_t s)
{
    (void)dec_exact; (void)evalcell; (void)counters; (void)labels; (void)ptext; (void)evals_cap;   // (the estimates: launch_probability_eval)
    hipLaunchKernelGGL(k_prob_grid, dim3(d.B * d.R), dim3(256), 0, s, mask, rowcount, own, gridf, P.hdr, d);
    const size_t cells = (size_t)d.B * d.R * d.H * d.W;
    hipLaunchKernelGGL(k_prob_vote_cells, dim3((unsigned)std::min<size_t>(8192, (cells + 255) / 256)), dim3(256), 0, s, gridf, evf, d);
    hipLaunchKernelGGL(k_prob_pick, dim3(d.B * d.R), dim3(1024), 0, s, evf, heights, brcount, rec, d);
}

void launch_prob_list(int *counters, int slot, int *list, int cap, hipStream_t s)
{
    hipLaunchKernelGGL(k_prob_list, dim3(cap < 65536 ? 64 : 1024), dim3(256), 0, s, counters, slot, list, cap);
}

}  // namespace haf
