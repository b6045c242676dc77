// kernels.hip -- hand-written gfx950 (CDNA4) kernels of the grasp-scoring hot path.
//
// Stage -> reference function it replaces (src/calc_grasppoints_action_server.cpp unless noted):
//   k_bin            generate_grid 406-529 (transform + max-z binning)
//   k_integral_totals / k_integral_band (k_integral_seq as fallback)   generate_grid 522-528 (empty cells -> 0) + calc_intimage 577-613
//   k_mask_count / k_scan / k_compact   pnt_in_box 666-749 + the row-major cell order of calc_featurevectors 637-643
//   k_features_serial / k_features   CIntImage_to_Featurevec::calc_featurevalue (fv.cpp:141-199), the "%.4g" text
//                    round trip (fv.cpp:133 -> svm-scale.c:270), svm-scale restore+output (svm-scale.c:333-353) and
//                    the "%g" round trip (svm-scale.c:350 -> svm-predict.c:108); large / small requests
//                    (the screening form, XMODE_SCREEN, skips the second round trip and writes the guard band instead)
//   k_svm_screen     (screen.hip) svm_predict_values / Kernel::k_function RBF (libsvm svm.cpp:325-365, 2478-2532) as ONE
//                    fp16 MFMA pass, trusted outside a rigorous band (tier 0, default mode); k_screen_count /
//                    k_screen_compact build the ordered list of what it could not decide
//   k_svm_rbf_h      the same as three fp16 MFMA passes on the hi/lo halves of the fp32 operands, exp epilogue, guard
//                    band (tier 1: on the list in the default mode, on everything with HAF_FLAG_SPLIT_F16)
//   k_svm_rbf        the same as one fp32 MFMA pass (HAF_FLAG_FP32_MFMA)
//   k_features<XMODE_F64> / k_recheck_mfma / k_recheck_combine   guard-band evaluations again as an fp64 MFMA contraction (tier 2)
//   k_recheck        what is still within 2^-40 of zero, in libsvm's exact fp64 summation order (tier 3)
//   k_vote_cells / k_vote_pick   show_predicted_gps 865-932 (29-tap vote, first-wins argmax, longest-run centring) and the
//                    z window of transform_gp_in_wcs_and_publish 1342-1351
//
// Built with -ffp-contract=off: every fp32/fp64 expression that must match the CPU restatement bit for bit is
// written with explicit *_rn intrinsics as well; fma() is used only where a fused operation is intended.
#include "kernels.h"
#include "decq.h"
#include <string.h>
#include <type_traits>
#ifndef HAF_ABL
#define HAF_ABL 0     // timing experiments on k_svm_rbf_h<true> (tools/ablate_h.sh): never defined in a build that is kept
#endif

namespace haf {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef double f64x4 __attribute__((ext_vector_type(4)));

// ---------------------------------------------------------------------------------------------------
// helpers
// ---------------------------------------------------------------------------------------------------
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

__global__ void k_fill_i32(int *p, int v, size_t n)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    size_t st = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += st) p[i] = v;
}

void launch_fill_i32(int *p, int v, size_t n, hipStream_t s)
{
    if (!n) return;
    int blocks = (int)((n + 255) / 256);
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(k_fill_i32, dim3(blocks), dim3(256), 0, s, p, v, n);
}

// ---------------------------------------------------------------------------------------------------
// a1: transform + binning.  One thread per (point, roll); max-z via atomicMax on an
// order-preserving integer key (max is order independent, so the grid is deterministic).
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_bin(const CloudDev *__restrict__ clouds, const RollGeo *__restrict__ geo,
                                             int *__restrict__ hkeys, Dims d, float r_row, float r_col)
{
    // grid = (points / 256, cloud * roll): workgroups are dispatched roll by roll, so the atomics in flight at any moment
    // go to one or two height grids (1 MiB each at 512 x 512), which stay in L2; with all rolls of a point in one thread
    // they spread over every grid of the request (37 MB for C5) and miss
    const int br = blockIdx.y;
    const int b = br / d.R;
    const CloudDev c = clouds[b];
    const int i = blockIdx.x * 256 + threadIdx.x;
    if ((int)blockIdx.x * 256 >= c.n) return;                        // whole workgroup past the cloud
    const bool valid = i < c.n;                                      // every lane stays for the shuffle below
    const float *p = c.xyz + (size_t)(valid ? i : 0) * c.stride;
    const float x = p[0], y = p[1], z = p[2];
    const int HW = d.H * d.W;
    const RollGeo &g = geo[br];
    // pcl::transformPointCloud (488): fp32, left to right, unfused
    float px = __fadd_rn(__fadd_rn(__fadd_rn(__fmul_rn(g.m[0], x), __fmul_rn(g.m[1], y)), __fmul_rn(g.m[2], z)), g.m[3]);
    float py = __fadd_rn(__fadd_rn(__fadd_rn(__fmul_rn(g.m[4], x), __fmul_rn(g.m[5], y)), __fmul_rn(g.m[6], z)), g.m[7]);
    float pz = __fadd_rn(__fadd_rn(__fadd_rn(__fmul_rn(g.m[8], x), __fmul_rn(g.m[9], y)), __fmul_rn(g.m[10], z)), g.m[11]);
    int ci = -1, key = 0;
    if (valid && (px > -r_row) && (px < r_row) && (py > -r_col) && (py < r_col) && (pz == pz)) {   // 510-511; NaN z never wins 515
        int ix = (int)floorf(__fmul_rn(100.0f, __fadd_rn(px, r_row)));                   // 513
        int iy = (int)floorf(__fmul_rn(100.0f, __fadd_rn(py, r_col)));                   // 514
        if (ix >= 0 && ix < d.H && iy >= 0 && iy < d.W) { ci = ix * d.W + iy; key = f2key(pz); }
    }
    // neighbouring points of a cloud mostly share a cell: of two adjacent lanes on the same cell only the higher one (the
    // odd lane on a tie) goes to memory
    const int ci_o = __shfl_xor(ci, 1, 64), key_o = __shfl_xor(key, 1, 64);
    const bool beaten = (ci_o == ci) && (key_o > key || (key_o == key && (threadIdx.x & 1) == 0));
    if (ci >= 0 && !beaten) {
        int *cell = hkeys + (size_t)br * HW + ci;
        if (key > *cell) atomicMax(cell, key);   // stale read is safe: the cell only grows
    }
}

// Small grids (the reference's 56 x 56: 12.5 KB of keys): a dense cloud puts dozens of points into every cell, and one
// global atomicMax per point and roll is all contention.  Here a workgroup bins a chunk of kBinChunk points of one
// (cloud, roll) into a private copy of the grid in LDS (ds_max_i32) and then publishes only the cells it touched, one
// global atomicMax each.  max is order independent: the grid is the same as k_bin's.
constexpr int kBinChunk = 2048;
constexpr int kBinLdsCells = 16384;              // 64 KiB of LDS: grids up to 128 x 128

__global__ __launch_bounds__(256) void k_bin_lds(const CloudDev *__restrict__ clouds, const RollGeo *__restrict__ geo,
                                                 int *__restrict__ hkeys, Dims d, float r_row, float r_col, int key_empty)
{
    extern __shared__ int cells[];
    const int br = blockIdx.y;
    const int b = br / d.R;
    const CloudDev c = clouds[b];
    const int first = blockIdx.x * kBinChunk;
    if (first >= c.n) return;
    const int HW = d.H * d.W;
    for (int k = threadIdx.x; k < HW; k += 256) cells[k] = key_empty;
    __syncthreads();
    const RollGeo &g = geo[br];
    const int last = min(c.n, first + kBinChunk);
    for (int i = first + threadIdx.x; i < last; i += 256) {
        const float *p = c.xyz + (size_t)i * c.stride;
        const float x = p[0], y = p[1], z = p[2];
        // pcl::transformPointCloud (488): fp32, left to right, unfused
        float px = __fadd_rn(__fadd_rn(__fadd_rn(__fmul_rn(g.m[0], x), __fmul_rn(g.m[1], y)), __fmul_rn(g.m[2], z)), g.m[3]);
        float py = __fadd_rn(__fadd_rn(__fadd_rn(__fmul_rn(g.m[4], x), __fmul_rn(g.m[5], y)), __fmul_rn(g.m[6], z)), g.m[7]);
        float pz = __fadd_rn(__fadd_rn(__fadd_rn(__fmul_rn(g.m[8], x), __fmul_rn(g.m[9], y)), __fmul_rn(g.m[10], z)), g.m[11]);
        if ((px > -r_row) && (px < r_row) && (py > -r_col) && (py < r_col) && (pz == pz)) {   // 510-511; NaN z never wins 515
            int ix = (int)floorf(__fmul_rn(100.0f, __fadd_rn(px, r_row)));                   // 513
            int iy = (int)floorf(__fmul_rn(100.0f, __fadd_rn(py, r_col)));                   // 514
            if (ix >= 0 && ix < d.H && iy >= 0 && iy < d.W) atomicMax(&cells[ix * d.W + iy], f2key(pz));
        }
    }
    __syncthreads();
    int *out = hkeys + (size_t)br * HW;
    for (int k = threadIdx.x; k < HW; k += 256) {
        const int v = cells[k];
        if (v > key_empty && v > out[k]) atomicMax(&out[k], v);    // stale read is safe: the cell only grows
    }
}

// Large grids (beyond what fits LDS) with a sizeable cloud: one global atomicMax per (point, roll) is what k_bin costs --
// 19 M of them at C5, 0.40 ms, an order of magnitude above what the 75 MB they move would take.  The bucket-sorted path
// removes every global atomic:
//   once per request   k_bkt_count / k_bkt_scan / k_bkt_scatter: the cloud is grouped by WHERE ITS POINTS LAND BEFORE THE ROLL
//                      (m0 = the transform without roll and x-scale) into square buckets of kBktCells x kBktCells grid cells
//                      -- a counting sort, LDS-private histograms, the points keep their original coordinates;
//   once per roll      k_bin_tiles: a workgroup owns a 64 x 64 tile of the output grid in LDS, walks the buckets whose image
//                      under this roll can touch the tile (a conservative test with 2 mm of slack: the fp32 product
//                      S R(roll) m0 and the full matrix differ by ~1e-6 m), transforms their points with the FULL matrix
//                      exactly as k_bin does, keeps the cell maximum with ds_max, and stores the finished tile with plain
//                      coalesced stores (empty cells included: no fill launch).
// max is order independent and the cell of a point is computed by the same fp32 expression: the grid is k_bin's bit for bit.
constexpr int kBinTile = 64;
constexpr int kBktChunk = 2048;                  // points per workgroup in the counting-sort passes
constexpr int kBktMaxBuckets = 9216;             // LDS histogram (36 KiB); bin_bucket_grid keeps nb*nb below it
constexpr int kBktListCap = 1024;                // candidate buckets of one tile (a 64-cell tile reaches ~120 buckets of 8 cells)
constexpr float kBktSlack = 0.002f;              // metres

int bin_bucket_grid(int H, int *bucket_cells)
{
    int bc = H / 64 > 8 ? H / 64 : 8;              // bucket edge in cells: nb stays ~ 1.414 * 64 + 3 for any grid size
    const double r = 0.005 * H, bs = 0.01 * bc, Rb = r * 1.41422 + bs;
    int nb = (int)(2.0 * Rb / bs) + 1;
    if (bucket_cells) *bucket_cells = bc;
    return nb;
}

struct BktGrid { float Rb, inv_bs, bs; int nb; };
__host__ __device__ inline BktGrid bkt_grid(int H)
{
    int bc = H / 64 > 8 ? H / 64 : 8;
    BktGrid g;
    g.bs = 0.01f * (float)bc;
    g.Rb = 0.005f * (float)H * 1.41422f + g.bs;
    g.inv_bs = 1.0f / g.bs;
    g.nb = (int)(2.0 * (0.005 * H * 1.41422 + 0.01 * bc) / (0.01 * bc)) + 1;
    return g;
}

// bucket of a point (or -1): where m0 puts it; any consistent rule works, the test in k_bin_tiles is made for THIS one
__device__ __forceinline__ int point_bucket(const CloudDev &c, const BktGrid &g, int i)
{
    const float *p = c.xyz + (size_t)i * c.stride;
    const float x = p[0], y = p[1], z = p[2];
    const float x0 = c.m0[0] * x + c.m0[1] * y + c.m0[2] * z + c.m0[3];
    const float y0 = c.m0[4] * x + c.m0[5] * y + c.m0[6] * z + c.m0[7];
    const float fx = (x0 + g.Rb) * g.inv_bs, fy = (y0 + g.Rb) * g.inv_bs;
    if (!(fx >= 0.0f && fx < (float)g.nb && fy >= 0.0f && fy < (float)g.nb)) return -1;     // also NaN
    return (int)fy * g.nb + (int)fx;
}

__global__ __launch_bounds__(256) void k_bkt_count(const CloudDev *__restrict__ clouds, int *__restrict__ bkt_count, Dims d)
{
    __shared__ int hist[kBktMaxBuckets];
    const int b = blockIdx.y;
    const CloudDev c = clouds[b];
    const int first = blockIdx.x * kBktChunk;
    if (first >= c.n) return;
    const BktGrid g = bkt_grid(d.H);
    const int nbk = g.nb * g.nb;
    for (int k = threadIdx.x; k < nbk; k += 256) hist[k] = 0;
    __syncthreads();
    const int last = min(c.n, first + kBktChunk);
    for (int i = first + threadIdx.x; i < last; i += 256) {
        const int q = point_bucket(c, g, i);
        if (q >= 0) atomicAdd(&hist[q], 1);
    }
    __syncthreads();
    int *out = bkt_count + c.bucket_off;
    for (int k = threadIdx.x; k < nbk; k += 256)
        if (hist[k]) atomicAdd(&out[k], hist[k]);
}

// exclusive scan of one cloud's bucket counts (<= 4096 of them): offsets (+ total at [nbk]) and the scatter cursors
__global__ __launch_bounds__(1024) void k_bkt_scan(const CloudDev *__restrict__ clouds, const int *__restrict__ bkt_count,
                                                   int *__restrict__ bkt_off, int *__restrict__ bkt_cursor, Dims d)
{
    __shared__ int part[1024];
    const CloudDev c = clouds[blockIdx.x];
    const BktGrid g = bkt_grid(d.H);
    const int nbk = g.nb * g.nb, t = threadIdx.x;
    const int per = (nbk + 1023) / 1024, lo = t * per, hi = min(nbk, lo + per);
    int sum = 0;
    for (int k = lo; k < hi; k++) sum += bkt_count[c.bucket_off + k];
    part[t] = sum;
    __syncthreads();
    for (int o = 1; o < 1024; o <<= 1) {
        const int v = (t >= o) ? part[t - o] : 0;
        __syncthreads();
        part[t] += v;
        __syncthreads();
    }
    int run = part[t] - sum;
    for (int k = lo; k < hi; k++) {
        bkt_off[c.bucket_off + k] = run;
        bkt_cursor[c.bucket_off + k] = run;
        run += bkt_count[c.bucket_off + k];
    }
    if (t == 1023) bkt_off[c.bucket_off + nbk] = part[1023];
}

__global__ __launch_bounds__(256) void k_bkt_scatter(const CloudDev *__restrict__ clouds, int *__restrict__ bkt_cursor,
                                                     float *__restrict__ sorted, Dims d)
{
    __shared__ int hist[kBktMaxBuckets];             // first the chunk's histogram, then each bucket's base in the sorted copy
    const int b = blockIdx.y;
    const CloudDev c = clouds[b];
    const int first = blockIdx.x * kBktChunk;
    if (first >= c.n) return;
    const BktGrid g = bkt_grid(d.H);
    const int nbk = g.nb * g.nb;
    for (int k = threadIdx.x; k < nbk; k += 256) hist[k] = 0;
    __syncthreads();
    const int last = min(c.n, first + kBktChunk);
    for (int i = first + threadIdx.x; i < last; i += 256) {
        const int q = point_bucket(c, g, i);
        if (q >= 0) atomicAdd(&hist[q], 1);
    }
    __syncthreads();
    // reserve this chunk's share of every bucket it touches: one global atomic per (chunk, bucket)
    for (int k = threadIdx.x; k < nbk; k += 256) {
        const int n = hist[k];
        hist[k] = n ? atomicAdd(&bkt_cursor[c.bucket_off + k], n) : 0;
    }
    __syncthreads();
    float *out = sorted + (size_t)c.sorted_off * 3;
    for (int i = first + threadIdx.x; i < last; i += 256) {
        const int q = point_bucket(c, g, i);
        if (q < 0) continue;
        const int pos = atomicAdd(&hist[q], 1);      // order inside a bucket is arbitrary: the cell maximum does not care
        const float *p = c.xyz + (size_t)i * c.stride;
        out[(size_t)pos * 3] = p[0];
        out[(size_t)pos * 3 + 1] = p[1];
        out[(size_t)pos * 3 + 2] = p[2];
    }
}

constexpr int kBinTileThreads = 512;
__global__ __launch_bounds__(kBinTileThreads) void k_bin_tiles(const CloudDev *__restrict__ clouds, const RollGeo *__restrict__ geo,
                                                   const float *__restrict__ sorted, const int *__restrict__ bkt_off,
                                                   int *__restrict__ hkeys, Dims d, float r_row, float r_col, int key_empty,
                                                   int *__restrict__ counters)
{
    __shared__ int cells[kBinTile * kBinTile];
    __shared__ int lstart[kBktListCap], lend[kBktListCap];         // point ranges of the candidate buckets
    __shared__ int nlist;
    const int br = blockIdx.y, b = br / d.R;
    const CloudDev c = clouds[b];
    const RollGeo &g = geo[br];
    const int tiles_w = (d.W + kBinTile - 1) / kBinTile;
    const int tx0 = (blockIdx.x / tiles_w) * kBinTile, ty0 = (blockIdx.x % tiles_w) * kBinTile;    // first row (x-bin) / column (y-bin)
    const BktGrid bg = bkt_grid(d.H);
    for (int k = threadIdx.x; k < kBinTile * kBinTile; k += kBinTileThreads) cells[k] = key_empty;
    if (threadIdx.x == 0) nlist = 0;
    __syncthreads();
    // ---- which buckets can reach this tile?  p = S(rw) R(roll) p0; tile = [xa, xb] x [ya, yb] in p (cell = floor(100 (p + r))) ----
    const float xa = -r_row + 0.01f * (float)tx0, xb = -r_row + 0.01f * (float)min(tx0 + kBinTile, d.H);
    const float ya = -r_col + 0.01f * (float)ty0, yb = -r_col + 0.01f * (float)min(ty0 + kBinTile, d.W);
    const float cxm = 0.5f * (xa + xb), cym = 0.5f * (ya + yb);
    const float rw = g.rw, hb = 0.70711f * bg.bs;                    // half diagonal of a bucket
    const float hx = 0.5f * (xb - xa) + fabsf(rw) * hb + kBktSlack, hy = 0.5f * (yb - ya) + hb + kBktSlack;
    // candidates: the buckets inside the bounding box (in p0) of the tile grown by the slack and a bucket's half diagonal ...
    float bx_lo = 1e30f, bx_hi = -1e30f, by_lo = 1e30f, by_hi = -1e30f;
    const float inv_rw = 1.0f / rw;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const float px = ((k & 1) ? cxm + hx : cxm - hx) * inv_rw, py = (k & 2) ? cym + hy : cym - hy;
        const float x0 = g.rc * px + g.rs * py, y0 = -g.rs * px + g.rc * py;          // p0 = R(-roll) S(1/rw) p
        bx_lo = fminf(bx_lo, x0); bx_hi = fmaxf(bx_hi, x0); by_lo = fminf(by_lo, y0); by_hi = fmaxf(by_hi, y0);
    }
    const int ix0 = max(0, (int)floorf((bx_lo + bg.Rb) * bg.inv_bs) - 1), ix1 = min(bg.nb - 1, (int)floorf((bx_hi + bg.Rb) * bg.inv_bs) + 1);
    const int iy0 = max(0, (int)floorf((by_lo + bg.Rb) * bg.inv_bs) - 1), iy1 = min(bg.nb - 1, (int)floorf((by_hi + bg.Rb) * bg.inv_bs) + 1);
    const int nx = max(0, ix1 - ix0 + 1), ncand = nx * max(0, iy1 - iy0 + 1);
    // ... whose centre's image lies within the tile's half extent + a bucket's half diagonal + slack (conservative both ways)
    for (int k = threadIdx.x; k < ncand; k += kBinTileThreads) {
        const int q = (iy0 + k / nx) * bg.nb + ix0 + k % nx;
        const int i0 = bkt_off[c.bucket_off + q], i1 = bkt_off[c.bucket_off + q + 1];
        if (i1 == i0) continue;                                                        // empty bucket
        const float qx = -bg.Rb + ((float)(q % bg.nb) + 0.5f) * bg.bs, qy = -bg.Rb + ((float)(q / bg.nb) + 0.5f) * bg.bs;   // centre in p0
        const float px = rw * (g.rc * qx - g.rs * qy), py = g.rs * qx + g.rc * qy;
        if (fabsf(px - cxm) <= hx && fabsf(py - cym) <= hy) {
            const int slot = atomicAdd(&nlist, 1);
            if (slot < kBktListCap) { lstart[slot] = i0; lend[slot] = i1; }
        }
    }
    __syncthreads();
    // The cap is ~5x what a tile can reach (about 205 buckets for |x-scale| >= 1).  Should a geometry ever exceed it, the grid
    // would silently miss points: say so instead -- the host redoes the request with k_bin (engine.cpp: CNT_ERROR).
    if (threadIdx.x == 0 && nlist > kBktListCap) atomicOr(&counters[CNT_ERROR], 1);
    const int nl = min(nlist, kBktListCap);
    const float *pts = sorted + (size_t)c.sorted_off * 3;
    // a wave per candidate bucket (a bucket of 8 x 8 cells holds ~128 points: two per lane), both loads of a trip in flight
    // before either point is processed; the ranges come from LDS, so nothing in this loop waits on a dependent global load
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    auto bin_point = [&](float x, float y, float z) {
        // pcl::transformPointCloud (488): fp32, left to right, unfused -- the same expression as k_bin
        float px = __fadd_rn(__fadd_rn(__fadd_rn(__fmul_rn(g.m[0], x), __fmul_rn(g.m[1], y)), __fmul_rn(g.m[2], z)), g.m[3]);
        float py = __fadd_rn(__fadd_rn(__fadd_rn(__fmul_rn(g.m[4], x), __fmul_rn(g.m[5], y)), __fmul_rn(g.m[6], z)), g.m[7]);
        if ((px > -r_row) && (px < r_row) && (py > -r_col) && (py < r_col)) {                 // 510-511
            const int ix = (int)floorf(__fmul_rn(100.0f, __fadd_rn(px, r_row))) - tx0;       // 513
            const int iy = (int)floorf(__fmul_rn(100.0f, __fadd_rn(py, r_col))) - ty0;       // 514
            if (ix >= 0 && ix < kBinTile && iy >= 0 && iy < kBinTile && ix + tx0 < d.H && iy + ty0 < d.W) {
                float pz = __fadd_rn(__fadd_rn(__fadd_rn(__fmul_rn(g.m[8], x), __fmul_rn(g.m[9], y)), __fmul_rn(g.m[10], z)), g.m[11]);
                if (pz == pz) atomicMax(&cells[ix * kBinTile + iy], f2key(pz));              // NaN z never wins (515)
            }
        }
    };
    for (int l = wave; l < nl; l += kBinTileThreads / 64) {
        const int i0 = lstart[l], i1 = lend[l];
        for (int i = i0 + lane; i < i1; i += 128) {
            const int j = (i + 64 < i1) ? i + 64 : i;
            const float x1 = pts[(size_t)i * 3], y1 = pts[(size_t)i * 3 + 1], z1 = pts[(size_t)i * 3 + 2];
            const float x2 = pts[(size_t)j * 3], y2 = pts[(size_t)j * 3 + 1], z2 = pts[(size_t)j * 3 + 2];
            bin_point(x1, y1, z1);
            if (j != i) bin_point(x2, y2, z2);
        }
    }
    __syncthreads();
    int *out = hkeys + (size_t)br * d.H * d.W;
    for (int k = threadIdx.x; k < kBinTile * kBinTile; k += kBinTileThreads) {
        const int i = k / kBinTile, j = k % kBinTile;
        if (tx0 + i < d.H && ty0 + j < d.W) out[(size_t)(tx0 + i) * d.W + ty0 + j] = cells[k];
    }
}

bool launch_bin(const CloudDev *clouds, const CloudDev *clouds_host, int max_n, long total_n, const RollGeo *geo, int *hkeys, Dims d,
                float r_row, float r_col, bool bucket_ok, BinScratch bs, int *counters, hipStream_t s)
{
    const int HW = d.H * d.W;
    float minus_one = -1.0f;
    int key_empty;
    memcpy(&key_empty, &minus_one, 4);
    key_empty ^= 0x7FFFFFFF;                                         // ordered key of -1.0f (499-501): an empty cell
    int bc;
    const int nb = bin_bucket_grid(d.H, &bc);
    const long nbk1 = (long)nb * nb + 1;
    // the bucket-sorted path: grids too large for k_bin_lds, enough points for the three sorting passes to pay, square grid
    if (bucket_ok && HW > kBinLdsCells && total_n >= 32768 && d.H == d.W && nbk1 <= kBktMaxBuckets && total_n <= bs.sorted_cap &&
        nbk1 * d.B <= bs.bkt_cap) {
        (void)hipMemsetAsync(bs.bkt_count, 0, (size_t)nbk1 * d.B * sizeof(int), s);
        dim3 grid((max_n + kBktChunk - 1) / kBktChunk, d.B);
        hipLaunchKernelGGL(k_bkt_count, grid, dim3(256), 0, s, clouds, bs.bkt_count, d);
        hipLaunchKernelGGL(k_bkt_scan, dim3(d.B), dim3(1024), 0, s, clouds, bs.bkt_count, bs.bkt_off, bs.bkt_cursor, d);
        hipLaunchKernelGGL(k_bkt_scatter, grid, dim3(256), 0, s, clouds, bs.bkt_cursor, bs.sorted, d);
        const int tiles = ((d.H + kBinTile - 1) / kBinTile) * ((d.W + kBinTile - 1) / kBinTile);
        hipLaunchKernelGGL(k_bin_tiles, dim3(tiles, d.B * d.R), dim3(kBinTileThreads), 0, s, clouds, geo, bs.sorted, bs.bkt_off, hkeys, d, r_row, r_col,
                           key_empty, counters);
        return true;
    }
    (void)clouds_host;
    launch_fill_i32(hkeys, key_empty, (size_t)d.B * d.R * HW, s);
    if (max_n <= 0) return false;
    if (HW <= kBinLdsCells && max_n >= 4 * kBinChunk) {
        dim3 grid((max_n + kBinChunk - 1) / kBinChunk, d.B * d.R);
        hipLaunchKernelGGL(k_bin_lds, grid, dim3(256), (size_t)HW * sizeof(int), s, clouds, geo, hkeys, d, r_row, r_col, key_empty);
        return false;
    }
    dim3 grid((max_n + 255) / 256, d.B * d.R);
    hipLaunchKernelGGL(k_bin, grid, dim3(256), 0, s, clouds, geo, hkeys, d, r_row, r_col);
    return false;
}

// ---------------------------------------------------------------------------------------------------
// a1 tail + a2: finalise heights (cells < -0.99 -> 0, 522-528) and build the integral image in the
// reference's summation ORDER: running fp64 row sum, then add the row above (cv::integral CV_64F), so the
// fp64 partial sums and the fp32 narrowing (599-601) are bit-identical for any input, not only when the
// sums happen to be exact.  Thread-per-row pass, then thread-per-column pass.
// ---------------------------------------------------------------------------------------------------
// The SEQUENTIAL form -- k_integral_seq: a thread per grid row (running sum along the row), then a thread per column of the
// integral image (running sum down the column), one workgroup per grid -- is the definition of the result and, since round 2,
// the fallback: it only runs for a grid whose parallel sums (k_integral_band, below) were not all exact.
__global__ __launch_bounds__(256) void k_integral_seq(int *hk, double *__restrict__ rowsum, float *__restrict__ ii,
                                                      const int *__restrict__ inexact_flags, int *__restrict__ counters, Dims d)
{
    const int br = blockIdx.x;
    if (!inexact_flags[br]) return;                       // the parallel form was exact for this grid (the normal case)
    if (threadIdx.x == 0) atomicAdd(&counters[CNT_INEXACT], 1);
    const int H = d.H, W = d.W, W1 = W + 1;
    int *keys = hk + (size_t)br * H * W;
    float *hts = reinterpret_cast<float *>(keys);
    double *rs = rowsum + (size_t)br * H * W;
    float *I = ii + (size_t)br * (H + 1) * W1;
    // rows: one thread per grid row, running sum along the row
    for (int row = threadIdx.x; row < H; row += 256) {
        double s = 0.0;
        // the running sum is sequential by definition; the loads are not: fetch 8 values ahead of the dependent chain
        for (int c0 = 0; c0 < W; c0 += 8) {
            int kreg[8];
#pragma unroll
            for (int q = 0; q < 8; q++) kreg[q] = (c0 + q < W) ? keys[row * W + c0 + q] : 0;
#pragma unroll
            for (int q = 0; q < 8; q++) {
                if (c0 + q < W) {
                    float h = __int_as_float(kreg[q]);    // already a finalised height (k_integral_band), not a key
                    if ((double)h < -0.99) h = 0.0f;      // 524-526 (double compare; idempotent)
                    hts[row * W + c0 + q] = h;
                    s = __dadd_rn(s, (double)h);          // 589: widened before the integral
                    rs[row * W + c0 + q] = s;
                }
            }
        }
    }
    __threadfence_block();
    __syncthreads();                                      // every row sum of this grid is in place (one workgroup per grid)
    // columns: one thread per column of the integral image, running sum down the column
    for (int c = threadIdx.x; c < W1; c += 256) {
        I[c] = 0.0f;
        if (c == 0) {
            for (int r = 0; r < H; r++) I[(r + 1) * W1] = 0.0f;
        } else {
            double acc = 0.0;
#pragma unroll 8
            for (int r = 0; r < H; r++) {
                acc = __dadd_rn(acc, rs[r * W + (c - 1)]);
                I[(r + 1) * W1 + c] = (float)acc;         // 601
            }
        }
    }
}

// ---- the parallel form -----------------------------------------------------------------------------------------------
// The reference's summed-area table is a SEQUENTIAL fp64 computation, and in general a different association rounds
// differently.  But: if every addition of a parallel evaluation is EXACT (its TwoSum residual is zero), the parallel result is
// the true 2-D prefix sum; all true prefix sums are then representable in fp64, so every addition of the sequential order is
// exact as well and both give the same bits.  That is the normal case (heights are fp32 numbers of similar magnitude: a few
// hundred thousand of them add up without rounding in 53 bits).  So the integral image is built with wave
// scans and LDS tiles, every fp64 addition carries its residual into a per-(cloud, roll) flag, and only a grid whose flag
// is set is redone by the sequential kernel above (k_integral_seq: its workgroups exit at once otherwise).
//   k_integral_totals: per band of 16 grid rows the column totals of the row sums (one fp64 per column)
//   k_integral_band  : the band's row sums, carry = totals of the bands above, column scan inside the band, fp32 store;
//                      also writes the finalised heights (cells < -0.99 -> 0, 522-528) over the keys
// Traffic per roll: keys read twice (L2), II and heights written once; the 2 MB fp64 row-sum scratch is gone.
constexpr int kIBandRows = 16;
constexpr int kIChunk = 512;                      // columns per pass: 64 lanes x 8, one thread per column in the column phase
constexpr int kIThreads = 512;                    // 8 waves, two rows each

// s = a + b with the flag raised when the sum is not exact (Knuth TwoSum residual)
__device__ __forceinline__ double add_checked(double a, double b, bool &inexact)
{
    const double s = __dadd_rn(a, b);
    const double bb = __dsub_rn(s, a);
    const double err = __dadd_rn(__dsub_rn(a, __dsub_rn(s, bb)), __dsub_rn(b, bb));
    inexact |= (err != 0.0);
    return s;
}

constexpr int kIPitch = kIChunk + kIChunk / 32;   // row pitch of the LDS tile in doubles: one double of padding per 32 columns,
                                                  // so that the 8-columns-per-lane stores of the row phase spread over all banks
__device__ __forceinline__ int ipad(int c) { return c + (c >> 5); }

__device__ __forceinline__ float final_height(int key)
{
    float h = key2f(key);
    if ((double)h < -0.99) h = 0.0f;                      // 524-526 (double compare)
    return h;
}

// Column totals of the row sums of one band: T[band][c] = sum_{rows of the band} sum_{b <= c} h[row][b]
//                                                       = prefix over the columns of the band's COLUMN sums,
// so one column sum per thread (coalesced) and one scan of the band's 512-wide vector instead of sixteen row scans.  Every
// addition is checked: the totals are then the true values whatever the association.
__global__ __launch_bounds__(kIThreads) void k_integral_totals(const int *__restrict__ hk, double *__restrict__ band_tot,
                                                               int *__restrict__ inexact_flags, Dims d)
{
    __shared__ double wsum[kIThreads / 64];
    __shared__ double carry;
    const int band = blockIdx.x, br = blockIdx.y;
    const int H = d.H, W = d.W;
    const int n_bands = (H + kIBandRows - 1) / kIBandRows;
    const int *keys = hk + (size_t)br * H * W;
    double *tot = band_tot + ((size_t)br * n_bands + band) * W;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int row0 = band * kIBandRows;
    bool inexact = false;
    if (tid == 0) carry = 0.0;
    __syncthreads();
    for (int c0 = 0; c0 < W; c0 += kIThreads) {
        const int c = c0 + tid;
        double v = 0.0;
        if (c < W) {
            int kr[kIBandRows];
#pragma unroll
            for (int r = 0; r < kIBandRows; r++) kr[r] = (row0 + r < H) ? keys[(size_t)(row0 + r) * W + c] : 0;   // all loads first, then the dependent adds
#pragma unroll
            for (int r = 0; r < kIBandRows; r++)
                if (row0 + r < H) v = add_checked(v, (double)final_height(kr[r]), inexact);
        }
        double incl = v;                                  // inclusive scan over the 512 columns of this pass
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const double up = __shfl_up(incl, o, 64);
            if (lane >= o) incl = add_checked(up, incl, inexact);
        }
        if (lane == 63) wsum[wave] = incl;
        __syncthreads();
        double base = carry;
        for (int w = 0; w < wave; w++) base = add_checked(base, wsum[w], inexact);
        const double out = add_checked(base, incl, inexact);
        if (c < W) tot[c] = out;
        __syncthreads();
        if (tid == kIThreads - 1) carry = out;            // (columns past W added zeros)
        __syncthreads();
    }
    if (__syncthreads_or(inexact) && tid == 0) atomicOr(&inexact_flags[br], 1);
}

// Row sums of the band (wave scans), carry = totals of the bands above, column scan inside the band, fp32 store (601); also
// writes the finalised heights over the keys.
__global__ __launch_bounds__(kIThreads) void k_integral_band(int *hk, const double *__restrict__ band_tot, float *__restrict__ ii,
                                                             int *__restrict__ inexact_flags, Dims d)
{
    __shared__ double rs[kIBandRows][kIPitch];            // row sums of the band, one pass of 512 columns (66 KiB)
    __shared__ double row_carry[kIBandRows];              // running row sum at the end of the previous pass
    const int band = blockIdx.x, br = blockIdx.y;
    const int H = d.H, W = d.W, W1 = W + 1;
    const int n_bands = (H + kIBandRows - 1) / kIBandRows;
    int *keys = hk + (size_t)br * H * W;
    float *hts = reinterpret_cast<float *>(keys);
    float *I = ii + (size_t)br * (H + 1) * W1;
    const double *tot = band_tot + ((size_t)br * n_bands) * W;  // [band][column]
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int row0 = band * kIBandRows;
    bool inexact = false;
    if (tid < kIBandRows) row_carry[tid] = 0.0;
    if (band == 0)                                        // first row and first column of the integral image are zero (cv::integral)
        for (int c = tid; c < W1; c += kIThreads) I[c] = 0.0f;
    if (tid < kIBandRows && row0 + tid < H) I[(size_t)(row0 + tid + 1) * W1] = 0.0f;
    __syncthreads();
    for (int c0 = 0; c0 < W; c0 += kIChunk) {
        // ---- column carry of this pass: totals of the bands above, loaded eight at a time (independent loads, then the adds) ----
        const int c = c0 + tid;
        double acc = 0.0;
        if (c < W)
            for (int b0 = 0; b0 < band; b0 += 8) {
                double t8[8];
#pragma unroll
                for (int k = 0; k < 8; k++) t8[k] = (b0 + k < band) ? tot[(size_t)(b0 + k) * W + c] : 0.0;
#pragma unroll
                for (int k = 0; k < 8; k++) acc = add_checked(acc, t8[k], inexact);
            }
        // ---- row phase: wave w scans rows 2w, 2w + 1 of the band over columns c0 .. c0 + 511 (8 per lane) ----
#pragma unroll
        for (int q = 0; q < kIBandRows / 8; q++) {
            const int rl = wave * (kIBandRows / 8) + q, row = row0 + rl;
            double v[8];
            const int cb = c0 + lane * 8;
            int kr[8];
            if (row < H && cb + 7 < W && (W & 3) == 0) {  // two 16-byte loads (rows start 16-byte aligned when W % 4 == 0)
                const int4 k0 = *reinterpret_cast<const int4 *>(keys + (size_t)row * W + cb);
                const int4 k1 = *reinterpret_cast<const int4 *>(keys + (size_t)row * W + cb + 4);
                kr[0] = k0.x; kr[1] = k0.y; kr[2] = k0.z; kr[3] = k0.w; kr[4] = k1.x; kr[5] = k1.y; kr[6] = k1.z; kr[7] = k1.w;
                float hf[8];
#pragma unroll
                for (int k = 0; k < 8; k++) { hf[k] = final_height(kr[k]); v[k] = (double)hf[k]; }   // 589: widened before the integral
                // the finalised heights replace the keys right here: nobody else reads these eight cells (k_integral_totals has run)
                *reinterpret_cast<float4 *>(hts + (size_t)row * W + cb) = float4{hf[0], hf[1], hf[2], hf[3]};
                *reinterpret_cast<float4 *>(hts + (size_t)row * W + cb + 4) = float4{hf[4], hf[5], hf[6], hf[7]};
            } else {
#pragma unroll
                for (int k = 0; k < 8; k++) {
                    v[k] = 0.0;
                    if (row < H && cb + k < W) {
                        const float hfk = final_height(keys[(size_t)row * W + cb + k]);
                        hts[(size_t)row * W + cb + k] = hfk;
                        v[k] = (double)hfk;
                    }
                }
            }
            // inclusive prefix inside the lane, exclusive scan of the lane totals over the wave, carry of the earlier passes
#pragma unroll
            for (int k = 1; k < 8; k++) v[k] = add_checked(v[k - 1], v[k], inexact);
            double incl = v[7];
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const double up = __shfl_up(incl, o, 64);
                if (lane >= o) incl = add_checked(up, incl, inexact);
            }
            double excl = __shfl_up(incl, 1, 64);
            if (lane == 0) excl = 0.0;
            const double base = add_checked(row_carry[rl], excl, inexact);
#pragma unroll
            for (int k = 0; k < 8; k++) rs[rl][ipad(lane * 8 + k)] = add_checked(base, v[k], inexact);
        }
        __syncthreads();
        if (tid < kIBandRows) row_carry[tid] = rs[tid][ipad(kIChunk - 1)];    // (columns past W hold the row's total: they added zeros)
        // ---- column phase: thread t owns column c0 + t ----
        if (c < W) {
#pragma unroll
            for (int r = 0; r < kIBandRows; r++) {
                acc = add_checked(acc, rs[r][ipad(tid)], inexact);
                if (row0 + r < H) I[(size_t)(row0 + r + 1) * W1 + c + 1] = (float)acc;     // 601
            }
        }
        __syncthreads();
    }
    if (__syncthreads_or(inexact) && tid == 0) atomicOr(&inexact_flags[br], 1);
}

// Small grids (the reference's 56 x 56, up to ~70 x 70): the whole grid of a roll fits LDS, and three launches cost more than the
// work.  One workgroup per (cloud, roll) does the SEQUENTIAL summation itself -- a thread per row, then a thread per column,
// fp64, in LDS -- which is the reference's order by construction (no exactness check needed), in one launch.
constexpr int kISmallCells = 8192;               // H * W up to this is worth checking; the LDS need decides (launch_integral)
__global__ __launch_bounds__(256) void k_integral_small(int *hk, float *__restrict__ ii, Dims d)
{
    extern __shared__ double s_rs[];                      // [H][pitch] row sums, then the heights as floats
    const int br = blockIdx.x, tid = threadIdx.x;
    const int H = d.H, W = d.W, W1 = W + 1;
    const int pitch = ((W + 15) / 16) * 16 + 1;           // = 1 (mod 16) doubles: the 64 rows of a column read hit distinct banks
    float *s_h = reinterpret_cast<float *>(s_rs + (size_t)H * pitch);
    int *keys = hk + (size_t)br * H * W;
    float *hts = reinterpret_cast<float *>(keys);
    float *I = ii + (size_t)br * (H + 1) * W1;
    for (int k = tid; k < H * W; k += 256) {              // coalesced: finalise the heights (522-528), keep a copy in LDS
        const float h = final_height(keys[k]);
        s_h[k] = h;
        hts[k] = h;
    }
    __syncthreads();
    for (int row = tid; row < H; row += 256) {            // running sum along the row (589-595)
        double s = 0.0;
        for (int c = 0; c < W; c++) {
            s = __dadd_rn(s, (double)s_h[row * W + c]);
            s_rs[(size_t)row * pitch + c] = s;
        }
    }
    __syncthreads();
    for (int c = tid; c < W1; c += 256) {                 // running sum down the column, fp32 store (601)
        I[c] = 0.0f;
        if (c == 0) {
            for (int r = 0; r < H; r++) I[(size_t)(r + 1) * W1] = 0.0f;
        } else {
            double acc = 0.0;
            for (int r = 0; r < H; r++) {
                acc = __dadd_rn(acc, s_rs[(size_t)r * pitch + (c - 1)]);
                I[(size_t)(r + 1) * W1 + c] = (float)acc;
            }
        }
    }
}

void launch_integral(int *hk, double *rowsum, float *ii, int *inexact_flags, int *counters, Dims d, hipStream_t s)
{
    if (d.H * d.W <= kISmallCells) {
        const int pitch = ((d.W + 15) / 16) * 16 + 1;
        const size_t lds = (size_t)d.H * pitch * sizeof(double) + (size_t)d.H * d.W * sizeof(float);
        if (lds <= 64 * 1024) {                               // (the default dynamic-LDS limit: grids up to ~70 x 70)
            hipLaunchKernelGGL(k_integral_small, dim3(d.B * d.R), dim3(256), lds, s, hk, ii, d);
            return;
        }
    }
    // rowsum doubles as the band-total scratch of the parallel form ([B*R][bands][W] doubles, far smaller) and as the row-sum
    // scratch of the sequential fallback
    const int n_bands = (d.H + kIBandRows - 1) / kIBandRows;
    (void)hipMemsetAsync(inexact_flags, 0, (size_t)d.B * d.R * sizeof(int), s);
    hipLaunchKernelGGL(k_integral_totals, dim3(n_bands, d.B * d.R), dim3(kIThreads), 0, s, hk, rowsum, inexact_flags, d);
    hipLaunchKernelGGL(k_integral_band, dim3(n_bands, d.B * d.R), dim3(kIThreads), 0, s, hk, rowsum, ii, inexact_flags, d);
    // sequential order for the grids whose parallel sums were not exact (practically never; the kernels exit at once otherwise)
    hipLaunchKernelGGL(k_integral_seq, dim3(d.B * d.R), dim3(256), 0, s, hk, rowsum, ii, inexact_flags, counters, d);
}

// ---------------------------------------------------------------------------------------------------
// a3: mask.  One wave per grid row; the rotated-rectangle scalars come from the host (glibc sinf/cosf with the
// reference's float/double mix), the per-cell tests are plain IEEE fp32 operations.
// ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ bool cell_in_box(const float *__restrict__ I, int W1, int H, int i, int j, const RollGeo &g)
{
    if (!(i > 6 && i < H - 7 && j > 6 && j < H - 7)) return false;                                 // 713
    const int th = 4;
    float box = __fsub_rn(I[(i + th) * W1 + (j + th)], I[(i - th - 1) * W1 + (j + th)]);
    box = __fsub_rn(box, I[(i + th) * W1 + (j - th - 1)]);
    box = __fadd_rn(box, I[(i - th - 1) * W1 + (j - th - 1)]);                                       // 714-717
    if (!(box > 0.03f)) return false;
    const float fj = (float)j, fi = (float)i;
    float t1 = __fadd_rn(__fmul_rn(-g.sa, __fadd_rn(-g.cx1, fj)), __fmul_rn(g.ca, __fadd_rn(-g.cy1, fi)));   // 718
    float t2 = __fadd_rn(__fmul_rn(-g.sa, __fadd_rn(-g.cx2, fj)), __fmul_rn(g.ca, __fadd_rn(-g.cy2, fi)));   // 719
    float t3 = __fadd_rn(__fmul_rn(g.ca, __fadd_rn(-g.cx3, fj)), __fmul_rn(g.sa, __fadd_rn(-g.cy3, fi)));    // 720
    float t4 = __fadd_rn(__fmul_rn(g.ca, __fadd_rn(-g.cx4, fj)), __fmul_rn(g.sa, __fadd_rn(-g.cy4, fi)));    // 721
    return ((double)t1 < 0.00001) && ((double)t2 > -0.00001) && ((double)t3 > -0.00001) && ((double)t4 < 0.00001);
}

// (a wave per grid row, kRowsPerWg rows per workgroup: 18 432 one-wave workgroups at C5 spent more on dispatch than on their rows)
constexpr int kRowsPerWg = 4;
__global__ __launch_bounds__(64 * kRowsPerWg) void k_mask_count(const float *__restrict__ ii, const RollGeo *__restrict__ geo,
                                                                uint8_t *__restrict__ mask, int *__restrict__ rowcount, Dims d)
{
    const int i = blockIdx.x * kRowsPerWg + (threadIdx.x >> 6), br = blockIdx.y, lane = threadIdx.x & 63;
    const int H = d.H, W = d.W, W1 = W + 1;
    if (i >= H) return;
    const float *I = ii + (size_t)br * (H + 1) * W1;
    const RollGeo &g = geo[br];
    uint8_t *mrow = mask + ((size_t)br * H + i) * W;
    int cnt = 0;
    for (int j0 = 0; j0 < W; j0 += 64) {
        int j = j0 + lane;
        bool m = (j < W) && cell_in_box(I, W1, H, i, j, g);
        if (j < W) mrow[j] = m ? 1 : 0;
        cnt += __popcll(__ballot(m));
    }
    if (lane == 0) rowcount[br * H + i] = cnt;
}

void launch_mask_count(const float *ii, const RollGeo *geo, uint8_t *mask, int *rowcount, Dims d, hipStream_t s)
{
    hipLaunchKernelGGL(k_mask_count, dim3((d.H + kRowsPerWg - 1) / kRowsPerWg, d.B * d.R), dim3(64 * kRowsPerWg), 0, s, ii, geo, mask, rowcount, d);
}

// Evaluation order.  The evaluations of a grid row are its masked cells from left to right, cut into chunks of 64: the
// whole chunks of all rows come first (region A, row by row), the left-over chunks (fewer than 64 cells) of all rows follow
// (region B).  A wave of the feature kernel takes 64 consecutive evaluations, so in region A it nearly always holds 64
// neighbouring cells of one row -- the case its LDS window band is made for (k_features_serial) -- and only region B and rows
// with holes fall back to per-lane addressing.  Nothing downstream depends on the order: labels, decision values and votes
// are written per cell through evalcell.
// Exclusive scans of the per-row counts (n = B*R*H entries), single workgroup: rowoff[k] = start of row k's whole chunks,
// rowoff[n + 1 + k] = start of its left-over chunk.
// One workgroup per (cloud, roll): it sums the counts of all the rows in front of its grid (coalesced, a few loads per thread:
// no atomics, so the offsets are deterministic), scans its own H rows in LDS and writes their offsets.  (One workgroup for
// the whole request took 32 us at C5: eighteen dependent loads per thread and a 1024-wide scan on a single CU.)
__device__ __forceinline__ unsigned long long row_pack(unsigned c) { return ((unsigned long long)(c & ~63u) << 32) | (c & 63u); }

__global__ __launch_bounds__(1024) void k_scan(const int *__restrict__ rowcount, int *__restrict__ rowoff,
                                               int *__restrict__ brcount, int *__restrict__ counters, Dims d)
{
    __shared__ unsigned long long part[1024];            // (whole-chunk cells << 32) | left-over cells
    __shared__ unsigned long long red[2][16];
    const int n = d.B * d.R * d.H, br = blockIdx.x, H = d.H;
    const int t = threadIdx.x, wave = t >> 6, lane = t & 63;
    // (1) everything in front of this grid, and everything at all (the left-over region starts behind ALL whole chunks)
    unsigned long long before = 0, all = 0;
    const int mine0 = br * H;
    for (int k = t; k < n; k += 1024) {
        const unsigned long long v = row_pack((unsigned)rowcount[k]);
        all += v;
        if (k < mine0) before += v;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { before += __shfl_xor(before, o, 64); all += __shfl_xor(all, o, 64); }
    if (lane == 0) { red[0][wave] = before; red[1][wave] = all; }
    __syncthreads();
    before = 0; all = 0;
#pragma unroll
    for (int w = 0; w < 16; w++) { before += red[0][w]; all += red[1][w]; }
    const int total_a = (int)(all >> 32), total = total_a + (int)(all & 0xffffffffu);
    // (2) this grid's rows: chunk of consecutive rows per thread, LDS scan of the chunk sums
    const int chunk = (H + 1023) / 1024;
    const int lo = t * chunk, hi = min(H, lo + chunk);
    unsigned long long s = 0;
    for (int k = lo; k < hi; k++) s += row_pack((unsigned)rowcount[mine0 + k]);
    part[t] = s;
    __syncthreads();
    for (int o = 1; o < 1024; o <<= 1) {
        unsigned long long v = (t >= o) ? part[t - o] : 0;
        __syncthreads();
        part[t] += v;
        __syncthreads();
    }
    unsigned long long run = before + part[t] - s;
    for (int k = lo; k < hi; k++) {
        rowoff[mine0 + k] = (int)(run >> 32);
        rowoff[n + 1 + mine0 + k] = total_a + (int)(run & 0xffffffffu);
        run += row_pack((unsigned)rowcount[mine0 + k]);
    }
    if (t == 1023) {
        const unsigned long long own = part[1023];
        brcount[br] = (int)(own >> 32) + (int)(own & 0xffffffffu);
        if (br == d.B * d.R - 1) { rowoff[n] = total_a; rowoff[2 * n + 1] = total; counters[CNT_EVALS] = total; }
    }
}

void launch_scan(const int *rowcount, int *rowoff, int *brcount, int *counters, Dims d, hipStream_t s)
{
    hipLaunchKernelGGL(k_scan, dim3(d.B * d.R), dim3(1024), 0, s, rowcount, rowoff, brcount, counters, d);
}

__global__ __launch_bounds__(64 * kRowsPerWg) void k_compact(const uint8_t *__restrict__ mask, const int *__restrict__ rowcount,
                                                             const int *__restrict__ rowoff, int *__restrict__ evalcell, Dims d)
{
    const int i = blockIdx.x * kRowsPerWg + (threadIdx.x >> 6), br = blockIdx.y, lane = threadIdx.x & 63;
    const int H = d.H, W = d.W, n = d.B * d.R * H;
    if (i >= H) return;
    const uint8_t *mrow = mask + ((size_t)br * H + i) * W;
    const int whole = rowcount[br * H + i] & ~63;
    const int base_a = rowoff[br * H + i], base_b = rowoff[n + 1 + br * H + i] - whole;
    int done = 0;
    for (int j0 = 0; j0 < W; j0 += 64) {
        int j = j0 + lane;
        bool m = (j < W) && mrow[j];
        unsigned long long bal = __ballot(m);
        if (m) {
            const int rank = done + __popcll(bal & ((1ull << lane) - 1ull));
            evalcell[(rank < whole ? base_a : base_b) + rank] = (br * H + i) * W + j;
        }
        done += __popcll(bal);
    }
}

void launch_compact(const uint8_t *mask, const int *rowcount, const int *rowoff, int *evalcell, Dims d, hipStream_t s)
{
    hipLaunchKernelGGL(k_compact, dim3((d.H + kRowsPerWg - 1) / kRowsPerWg, d.B * d.R), dim3(64 * kRowsPerWg), 0, s, mask, rowcount, rowoff, evalcell, d);
}

// ---------------------------------------------------------------------------------------------------
// Small grids, ONE launch for a1 (tail) + a2 + a3 + a4: the reference's own 56 x 56 grid fits LDS with everything derived from it,
// and a small request is bound by the number of dependent launches (fill, bin, integral, mask, scan, compact: six launches of
// 4-7 us each for ~10 us of work).  One workgroup per (cloud, roll):
//   cells <- -1 keys; [BIN: transform + ds_max of the cloud's points, exactly k_bin_lds's arithmetic]  or  cells <- the keys a
//   binning kernel left in global memory (clouds too large for one workgroup per roll); heights finalised (522-528);
//   SEQUENTIAL fp64 row sums, then column sums (the reference's order by construction, as k_integral_small); mask and row counts
//   (k_mask_count's cell_in_box on the LDS copy); the roll's evaluations appended to the global list in row-major order.
// The list segment of a roll is reserved with ONE atomicAdd on counters[CNT_EVALS]: the order of the rolls inside the list
// depends on which workgroup gets there first, nothing else does (labels, decision values and votes are written per cell through
// evalcell).  The labels of the roll's grid are initialised here too (-1: no feature vector, server.cpp:828-829).
// DIRECT: also enter every evaluation into the fp64 tier's list (requests so small that the exact tier costs less than the
// fast ones' launches: engine.cpp).
// ---------------------------------------------------------------------------------------------------
constexpr int kSmallPreThreads = 1024;
constexpr int kSmallPreMaxPoints = 16384;        // BIN inside the kernel up to this many points per cloud (one workgroup per roll reads them all)

__host__ __device__ inline int small_pre_pitch(int W) { return ((W + 15) / 16) * 16 + 1; }
size_t small_pre_lds(int H, int W)
{
    return (size_t)H * small_pre_pitch(W) * sizeof(double) + (size_t)H * W * 4 + (size_t)(H + 1) * (W + 1) * 4 + (size_t)H * 4;
}

template <bool BIN>
__global__ __launch_bounds__(kSmallPreThreads) void k_small_pre(const CloudDev *__restrict__ clouds, const RollGeo *__restrict__ geo,
                                                                int *hk, float *__restrict__ ii, uint8_t *__restrict__ mask,
                                                                int *__restrict__ rowcount, int *__restrict__ brcount,
                                                                int8_t *__restrict__ labels, int *__restrict__ evalcell,
                                                                int *__restrict__ counters, int *__restrict__ flag_list, int direct,
                                                                Dims d, float r_row, float r_col, int key_empty)
{
    extern __shared__ double s_rs[];                      // [H][pitch] fp64 row sums | [H*W] keys -> heights | [(H+1)*(W+1)] II | [H] counts
    __shared__ int s_base;
    const int br = blockIdx.x, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int H = d.H, W = d.W, W1 = W + 1, HW = H * W;
    const int pitch = small_pre_pitch(W);
    int *cells = reinterpret_cast<int *>(s_rs + (size_t)H * pitch);
    float *s_h = reinterpret_cast<float *>(cells);
    float *s_I = reinterpret_cast<float *>(cells + HW);
    int *s_cnt = reinterpret_cast<int *>(s_I + (size_t)(H + 1) * W1);
    int *keys = hk + (size_t)br * HW;
    float *hts = reinterpret_cast<float *>(keys);
    float *I = ii + (size_t)br * (H + 1) * W1;
    const RollGeo &g = geo[br];
    int8_t *lab = labels + (size_t)br * HW;
    for (int k = tid; k < HW; k += kSmallPreThreads) {
        cells[k] = BIN ? key_empty : keys[k];
        lab[k] = (int8_t)-1;
    }
    __syncthreads();
    if (BIN) {
        const CloudDev c = clouds[br / d.R];
        for (int i = tid; i < c.n; i += kSmallPreThreads) {
            const float *p = c.xyz + (size_t)i * c.stride;
            const float x = p[0], y = p[1], z = p[2];
            // pcl::transformPointCloud (488): fp32, left to right, unfused
            float px = __fadd_rn(__fadd_rn(__fadd_rn(__fmul_rn(g.m[0], x), __fmul_rn(g.m[1], y)), __fmul_rn(g.m[2], z)), g.m[3]);
            float py = __fadd_rn(__fadd_rn(__fadd_rn(__fmul_rn(g.m[4], x), __fmul_rn(g.m[5], y)), __fmul_rn(g.m[6], z)), g.m[7]);
            float pz = __fadd_rn(__fadd_rn(__fadd_rn(__fmul_rn(g.m[8], x), __fmul_rn(g.m[9], y)), __fmul_rn(g.m[10], z)), g.m[11]);
            if ((px > -r_row) && (px < r_row) && (py > -r_col) && (py < r_col) && (pz == pz)) {   // 510-511; NaN z never wins 515
                int ix = (int)floorf(__fmul_rn(100.0f, __fadd_rn(px, r_row)));                   // 513
                int iy = (int)floorf(__fmul_rn(100.0f, __fadd_rn(py, r_col)));                   // 514
                if (ix >= 0 && ix < H && iy >= 0 && iy < W) atomicMax(&cells[ix * W + iy], f2key(pz));
            }
        }
        __syncthreads();
    }
    for (int k = tid; k < HW; k += kSmallPreThreads) {     // finalise the heights (522-528): LDS copy + the grid a11 and the debug fetch read
        const float h = final_height(cells[k]);
        s_h[k] = h;
        hts[k] = h;
    }
    __syncthreads();
    for (int row = tid; row < H; row += kSmallPreThreads) {   // running sum along the row (589-595)
        double s = 0.0;
        for (int c = 0; c < W; c++) {
            s = __dadd_rn(s, (double)s_h[row * W + c]);
            s_rs[(size_t)row * pitch + c] = s;
        }
    }
    __syncthreads();
    for (int c = tid; c < W1; c += kSmallPreThreads) {        // running sum down the column, fp32 store (601)
        s_I[c] = 0.0f;
        I[c] = 0.0f;
        if (c == 0) {
            for (int r = 0; r < H; r++) { s_I[(size_t)(r + 1) * W1] = 0.0f; I[(size_t)(r + 1) * W1] = 0.0f; }
        } else {
            double acc = 0.0;
            for (int r = 0; r < H; r++) {
                acc = __dadd_rn(acc, s_rs[(size_t)r * pitch + (c - 1)]);
                const float v = (float)acc;
                s_I[(size_t)(r + 1) * W1 + c] = v;
                I[(size_t)(r + 1) * W1 + c] = v;
            }
        }
    }
    __syncthreads();
    // mask (666-749) and row counts: a wave per grid row
    uint8_t *mgrid = mask + (size_t)br * HW;
    for (int i = wave; i < H; i += kSmallPreThreads / 64) {
        int cnt = 0;
        for (int j0 = 0; j0 < W; j0 += 64) {
            const int j = j0 + lane;
            const bool m = (j < W) && cell_in_box(s_I, W1, H, i, j, g);
            if (j < W) mgrid[i * W + j] = m ? 1 : 0;
            cnt += __popcll(__ballot(m));
        }
        if (lane == 0) { s_cnt[i] = cnt; rowcount[br * H + i] = cnt; }
    }
    __syncthreads();
    if (tid == 0) {                                        // exclusive prefix over the rows (H <= a few dozen), segment reservation
        int run = 0;
        for (int i = 0; i < H; i++) { const int c = s_cnt[i]; s_cnt[i] = run; run += c; }
        brcount[br] = run;
        s_base = run ? atomicAdd(&counters[CNT_EVALS], run) : 0;
        if (direct && run) atomicAdd(&counters[CNT_FLAGGED], run);
    }
    __syncthreads();
    const int base = s_base;
    for (int i = wave; i < H; i += kSmallPreThreads / 64) {   // (each lane re-reads the mask bytes it wrote itself)
        int done = s_cnt[i];
        for (int j0 = 0; j0 < W; j0 += 64) {
            const int j = j0 + lane;
            const bool m = (j < W) && mgrid[i * W + j];
            const unsigned long long bal = __ballot(m);
            if (m) {
                const int e = base + done + __popcll(bal & ((1ull << lane) - 1ull));
                evalcell[e] = (br * H + i) * W + j;
                if (direct) flag_list[e] = e;
            }
            done += __popcll(bal);
        }
    }
}

// true when the fused form ran (then nothing else of a1 tail / a2 / a3 / a4 has to be launched, and the labels are initialised)
bool launch_small_pre(const CloudDev *clouds, const RollGeo *geo, int max_n, int *hkeys, float *ii, uint8_t *mask, int *rowcount,
                      int *brcount, int8_t *labels, int *evalcell, int *counters, int *flag_list, bool direct, Dims d, float r_row,
                      float r_col, hipStream_t s)
{
    const size_t lds = small_pre_lds(d.H, d.W);
    if (lds > 64 * 1024) return false;                    // (the default dynamic-LDS limit: grids up to ~58 x 58)
    float minus_one = -1.0f;
    int key_empty;
    memcpy(&key_empty, &minus_one, 4);
    key_empty ^= 0x7FFFFFFF;                              // ordered key of -1.0f (499-501): an empty cell
    if (max_n <= kSmallPreMaxPoints) {
        hipLaunchKernelGGL(k_small_pre<true>, dim3(d.B * d.R), dim3(kSmallPreThreads), lds, s, clouds, geo, hkeys, ii, mask, rowcount, brcount,
                           labels, evalcell, counters, flag_list, direct ? 1 : 0, d, r_row, r_col, key_empty);
    } else {
        // a large cloud: many workgroups bin it (k_bin_lds: LDS-private grids, one global atomicMax per touched cell), then the rest
        launch_fill_i32(hkeys, key_empty, (size_t)d.B * d.R * d.H * d.W, s);
        dim3 grid((max_n + kBinChunk - 1) / kBinChunk, d.B * d.R);
        hipLaunchKernelGGL(k_bin_lds, grid, dim3(256), (size_t)d.H * d.W * sizeof(int), s, clouds, geo, hkeys, d, r_row, r_col, key_empty);
        hipLaunchKernelGGL(k_small_pre<false>, dim3(d.B * d.R), dim3(kSmallPreThreads), lds, s, clouds, geo, hkeys, ii, mask, rowcount, brcount,
                           labels, evalcell, counters, flag_list, direct ? 1 : 0, d, r_row, r_col, key_empty);
    }
    return true;
}

// ---------------------------------------------------------------------------------------------------
// a5/a6: one feature value from the 15x15 integral window (fv.cpp:141-199).  fp32, strict order, unfused.
// ---------------------------------------------------------------------------------------------------
// Integral-image reads go through a buffer descriptor: address = descriptor base + 32-bit VGPR byte offset (the window
// origin of the lane's cell) + SGPR byte offset (the region corner from the wave-uniform feature descriptor), i.e.
// `buffer_load_dword v, v_off, s[rsrc], s_off offen` with NO vector address arithmetic per load.  UNI = false (feature
// index differs per lane: recheck kernels) folds the corner offset into the VGPR instead.
typedef __amdgpu_buffer_rsrc_t rsrc_t;
__device__ __forceinline__ rsrc_t make_ii_rsrc(const float *ii, Dims d)
{
    const unsigned bytes = (unsigned)d.B * (unsigned)d.R * (unsigned)((d.H + 1) * (d.W + 1)) * 4u;   // < 2^32, checked in haf_create
    return __builtin_amdgcn_make_buffer_rsrc((void *)ii, 0, (int)bytes, 0x00020000);
}

template <bool UNI>
__device__ __forceinline__ float ii_load(rsrc_t r, unsigned w0b, int off)
{
    if (UNI) return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, (int)w0b, off * 4, 0));
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, (int)(w0b + (unsigned)off * 4u), 0, 0));
}

// Where the corners of a feature's regions come from: the integral image through the buffer descriptor (corner offset in
// an SGPR when the feature is wave-uniform, UNI, else folded into the VGPR), or a copy of the evaluation's 15x15 window in
// LDS (k_features).  Same values, same arithmetic.
template <bool UNI>
struct SrcBuf {
    rsrc_t r;
    unsigned w0b;
    __device__ __forceinline__ float corner(const FeatDesc &f, int k, int j) const { return ii_load<UNI>(r, w0b, f.off[k][j]); }
};
struct SrcWin {
    const float *win;             // this lane's window, row pitch 15
    __device__ __forceinline__ float corner(const FeatDesc &f, int k, int j) const { return win[f.offw[k][j]]; }
};

template <class Src>
__device__ __forceinline__ float region_sum(const Src &src, const FeatDesc &f, int k)
{
    float s = __fsub_rn(src.corner(f, k, 0), src.corner(f, k, 1));
    s = __fsub_rn(s, src.corner(f, k, 2));
    return __fadd_rn(s, src.corner(f, k, 3));                            // fv.cpp:161-162 / 183-184
}

template <class Src>
__device__ __forceinline__ float feature_value(const Src &src, const FeatDesc &f)
{
    if (!f.shaf) {
        float rv = 0.0f;
#pragma unroll
        for (int k = 0; k < 3; k++)
            if (f.active & (1 << k)) rv = __fadd_rn(rv, __fmul_rn(f.w[k], region_sum(src, f, k)));
        return rv;
    }
    float r[3] = {0.0f, 0.0f, 0.0f};
#pragma unroll
    for (int k = 0; k < 3; k++)
        if (f.active & (1 << k)) r[k] = __fmul_rn(f.w[k], region_sum(src, f, k));
    if (r[1] > r[0] && r[1] > r[2]) {                                    // fv.cpp:187-191
        float a = __fsub_rn(r[1], r[0]), b = __fsub_rn(r[1], r[2]);
        return (b < a) ? b : a;
    }
    return -1.0f;
}

// fp32 feature -> attribute value svm-predict would parse (both decimal text round trips emulated exactly)
template <class Src, class Tabs>
__device__ __forceinline__ double attribute_value(const Src &src, const FeatDesc &f, double lower, double upper, const Tabs &tb)
{
    float v = feature_value(src, f);
    double q4 = hafq::decq4_float(v, tb);
    return hafq::scale_q6(q4, f.fmin, f.fmax, f.range, f.inv_range, lower, upper, tb);
}

// The same, leaving the three stages of the attribute behind for haf_debug_fetch_attr (HAF_FLAG_KEEP_DEBUG): rec == nullptr in
// every production call.  An attribute svm-scale drops (f.skip) is 0 for the contraction; its feature and "%.4g" value are
// still what fv.cpp writes into the text file, so the record keeps them.
template <class Src, class Tabs>
__device__ __forceinline__ double attribute_value_rec(const Src &src, const FeatDesc &f, double lower, double upper, const Tabs &tb,
                                                      AttrRecord *rec)
{
    if (f.skip && !rec) return 0.0;
    const float v = feature_value(src, f);
    const double q4 = hafq::decq4_float(v, tb);
    const double x = f.skip ? 0.0 : hafq::scale_q6(q4, f.fmin, f.fmax, f.range, f.inv_range, lower, upper, tb);
    if (rec) { rec->feature = v; rec->pad = 0.0f; rec->q4 = q4; rec->scaled = x; }
    return x;
}

// Attribute for the SCREENING pass only, already multiplied by c (kernels.h: ScreenParams): the "%.4g" round trip through the
// table-driven decq4_float_scr, svm-scale's formula in plain fp64 with the constants folded on the host (u' = fma(q4,
// scr_mul, scr_add): one instruction with two scalar operands), and NO "%g" round trip.  With x the value svm-predict would parse and u = c x, the result u' satisfies
// |u' - u| <= 5e-6 |u'| (six significant decimal digits: half a unit of the sixth digit is <= 5e-6 relative) plus, in norm over
// the attributes, ScreenParams::eta_abs (engine.cpp: the fp64 roundings of both evaluations of the formula, the exact-zero
// omission and the min/max shortcuts).  screen_finish() carries that difference through the guard band; evaluations the
// screening pass cannot decide get the exact attributes in the three-pass tier.  An fp32 feature outside the decimal
// path's range comes back NaN and poisons the norms: that evaluation is never trusted.
constexpr double kScreenEtaRel = 5.0e-6 * (1.0 + 1e-6);
template <class Src>
__device__ __forceinline__ double screen_attribute(const Src &src, const FeatDesc &f, const hafq::ScrTabs &st)
{
    const float v = feature_value(src, f);
    return fma(hafq::decq4_float_scr(v, st), f.scr_mul, f.scr_add);
}

// ---- the fast form of the screening feature pass ------------------------------------------------------------------
// What bounds the per-lane form (buffer loads at window origin + corner offset) is the vector L1: the texture addresser
// coalesces 16 lanes at a time, 64 consecutive floats at an arbitrary alignment cost ~7.5 tag accesses per load, and with
// ~2400 loads per evaluation the TA is 97 % busy (profiles/README.md).  So a wave whose 64 evaluations are 64 neighbouring
// cells of one row (k_scan's order makes that the rule) first copies the band of the integral image its windows cover --
// 15 rows x 78 columns -- into LDS, and then reads every corner with ds_read_addtid_b32: LDS address = M0 + lane * 4, M0 =
// band + corner offset from the wave-uniform descriptor, so a corner costs two scalar instructions and one conflict-free
// LDS read, no vector address arithmetic, no L1 traffic.
constexpr int kBandRows = 15;
constexpr int kBandFloats4 = kBandRows * kBandPitch;  // per wave

// Four attribute slots of a "fast" group (ScreenParams::fast_groups: plain HAF features of at most two regions): the 32
// corner reads go out back to back before anything waits on them, and nothing branches.  An inactive region has weight 0
// and its corners at the window origin: it adds 0.0f * 0.0f, which leaves the sum of fv.cpp:164 as it is.
// hipcc does not know that the asm reads are asynchronous: the registers are handed on only through the s_waitcnt statement.
// The descriptors are read through the constant address space: the memory clobbers around the band would otherwise make
// hipcc fetch every wave-uniform descriptor word with a vector load.
typedef const ScrDesc __attribute__((address_space(4))) *ScrDescK;
__device__ __forceinline__ ScrDescK constant_ptr(const ScrDesc *p) { return (ScrDescK)(unsigned long long)p; }

__device__ __forceinline__ void screen_quad(unsigned band, ScrDescK sd, const hafq::ScrTabs &st, double *ud)
{
    float c[4][8];
    unsigned adr[4][8];                               // all descriptor words first: a volatile asm pins what follows it
#pragma unroll
    for (int q = 0; q < 4; q++)
#pragma unroll
        for (int j = 0; j < 8; j++) adr[q][j] = band + (unsigned)sd[q].off[j];
#pragma unroll
    for (int q = 0; q < 4; q++)
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const unsigned a = adr[q][j];
            asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tds_read_addtid_b32 %0" : "=v"(c[q][j]) : "s"(a) : "m0");
        }
    asm volatile("s_waitcnt lgkmcnt(0)"
                 : "+v"(c[0][0]), "+v"(c[0][1]), "+v"(c[0][2]), "+v"(c[0][3]), "+v"(c[0][4]), "+v"(c[0][5]), "+v"(c[0][6]), "+v"(c[0][7]));
#pragma unroll
    for (int q = 1; q < 4; q++)
        asm volatile("" : "+v"(c[q][0]), "+v"(c[q][1]), "+v"(c[q][2]), "+v"(c[q][3]), "+v"(c[q][4]), "+v"(c[q][5]), "+v"(c[q][6]),
                          "+v"(c[q][7]));
#pragma unroll
    for (int q = 0; q < 4; q++) {
        const float r0 = __fmul_rn(sd[q].w[0], __fadd_rn(__fsub_rn(__fsub_rn(c[q][0], c[q][1]), c[q][2]), c[q][3]));
        const float r1 = __fmul_rn(sd[q].w[1], __fadd_rn(__fsub_rn(__fsub_rn(c[q][4], c[q][5]), c[q][6]), c[q][7]));
        const float v = __fadd_rn(r0, r1);      // 0.0f + r0 first (fv.cpp:164) only turns a -0 into +0: same decimal, same u'
        ud[q] = fma(hafq::decq4_float_scr(v, st), sd[q].scr_mul, sd[q].scr_add);
    }
}

// Two attribute slots of any other group, from the band: three regions each, the HAF sum or the SHAF rule (feature_value).
// A slot of a dropped or absent attribute has scr_mul = scr_add = 0: its u' is 0 (NaN if its feature value left the decimal
// path's range, which only costs that evaluation the screening pass).
typedef const ScrDesc3 __attribute__((address_space(4))) *ScrDesc3K;
__device__ __forceinline__ ScrDesc3K constant_ptr(const ScrDesc3 *p) { return (ScrDesc3K)(unsigned long long)p; }

__device__ __forceinline__ void screen_pair3(unsigned band, ScrDesc3K sd, const hafq::ScrTabs &st, double *ud)
{
    float c[2][12];
    unsigned adr[2][12];
#pragma unroll
    for (int q = 0; q < 2; q++)
#pragma unroll
        for (int j = 0; j < 12; j++) adr[q][j] = band + (unsigned)sd[q].off[j];
#pragma unroll
    for (int q = 0; q < 2; q++)
#pragma unroll
        for (int j = 0; j < 12; j++) {
            const unsigned a = adr[q][j];
            asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tds_read_addtid_b32 %0" : "=v"(c[q][j]) : "s"(a) : "m0");
        }
    asm volatile("s_waitcnt lgkmcnt(0)"
                 : "+v"(c[0][0]), "+v"(c[0][1]), "+v"(c[0][2]), "+v"(c[0][3]), "+v"(c[0][4]), "+v"(c[0][5]), "+v"(c[0][6]), "+v"(c[0][7]),
                   "+v"(c[0][8]), "+v"(c[0][9]), "+v"(c[0][10]), "+v"(c[0][11]));
    asm volatile("" : "+v"(c[1][0]), "+v"(c[1][1]), "+v"(c[1][2]), "+v"(c[1][3]), "+v"(c[1][4]), "+v"(c[1][5]), "+v"(c[1][6]), "+v"(c[1][7]),
                      "+v"(c[1][8]), "+v"(c[1][9]), "+v"(c[1][10]), "+v"(c[1][11]));
#pragma unroll
    for (int q = 0; q < 2; q++) {
        float r[3];
#pragma unroll
        for (int k = 0; k < 3; k++)
            r[k] = __fmul_rn(sd[q].w[k], __fadd_rn(__fsub_rn(__fsub_rn(c[q][4 * k], c[q][4 * k + 1]), c[q][4 * k + 2]), c[q][4 * k + 3]));
        float v;
        if (sd[q].shaf) {                                              // wave-uniform
            v = -1.0f;
            if (r[1] > r[0] && r[1] > r[2]) {                          // fv.cpp:187-191
                const float a = __fsub_rn(r[1], r[0]), b = __fsub_rn(r[1], r[2]);
                v = (b < a) ? b : a;
            }
        } else {
            v = __fadd_rn(__fadd_rn(r[0], r[1]), r[2]);
        }
        ud[q] = fma(hafq::decq4_float_scr(v, st), sd[q].scr_mul, sd[q].scr_add);
    }
}

// the decimal tables (95 doubles) in LDS: call from every thread of the workgroup before any divergent return
__device__ __forceinline__ hafq::PtrTabs load_decimal_tables(double *lds_tab)
{
    if (threadIdx.x < hafq::kTabDoubles) lds_tab[threadIdx.x] = hafq::tab_entry((int)threadIdx.x);
    __syncthreads();
    hafq::PtrTabs tb;
    tb.t = lds_tab;
    return tb;
}

// the screening decimal tables (decq.h: 256 exponent entries + 15 pairs, 2288 bytes) in LDS; workgroups of >= 256 threads
__device__ __forceinline__ hafq::ScrTabs load_screen_tables(unsigned long long *lds_tab)
{
    if (threadIdx.x < hafq::kScrExpEntries) lds_tab[threadIdx.x] = hafq::scr_tab_word((int)threadIdx.x);
    if (threadIdx.x < 2 * hafq::kScrPairs) lds_tab[hafq::kScrExpEntries + threadIdx.x] = hafq::scr_tab_word(hafq::kScrExpEntries + (int)threadIdx.x);
    __syncthreads();
    hafq::ScrTabs st;
    st.w = lds_tab;
    return st;
}

// BYTE offset of the 15x15 window origin II[i-7][j-7] of a cell id (br*H + i)*W + j inside the integral-image buffer
__device__ __forceinline__ unsigned window_origin(int cell, int H, int W)
{
    const int br = cell / (H * W);
    const int rem = cell - br * H * W;
    const int i = rem / W, j = rem - i * W;
    return ((unsigned)br * (unsigned)((H + 1) * (W + 1)) + (unsigned)((i - 7) * (W + 1) + (j - 7))) * 4u;
}

// X image, fp32 form: tiles of 32 evals, k-major inside a tile ([tile][kDP][32] fp32) -- the exact register image of
// the fp32 MFMA A operand, so the contraction kernel fills its A fragments with fully coalesced 256-byte loads.
// X image, split-fp16 form: per tile of 32 evals two operand images (hi, lo) of [21 k-steps][2 k-halves][32 evals][8 fp16];
// a thread finishes 8 attributes, then stores them as one 16-byte vector per image (512 contiguous bytes per 32 lanes).
// X image, screening form: per tile of 32 evals ONE operand image of the same layout holding fp16(c*x) plus the norm
// slots (kernels.h); the per-evaluation guard band goes where the other forms keep a_x.
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));

// stores attributes 8g..8g+7 of tile row r into one operand image (h_image_offset): one 16-byte vector for the 16x16x32
// steps, two 8-byte vectors for the 16-wide K tail
__device__ __forceinline__ void store_group_img(char *img, int r, int g, half8 v)
{
    // streaming stores: the operand images (5.3 GB at C5) are read once, by the contraction kernel, long after they have left
    // every cache; written with the nt hint they do not push the integral image and the descriptors out on their way (-3 %)
#define HAF_X_STORE(p, v) __builtin_nontemporal_store(v, p)
    if (g < kHFull * 4) {
        HAF_X_STORE(reinterpret_cast<half8 *>(img + h_image_offset(r, g * 8)), v);
    } else {
        const half4 v0 = {v[0], v[1], v[2], v[3]}, v1 = {v[4], v[5], v[6], v[7]};
        HAF_X_STORE(reinterpret_cast<half4 *>(img + h_image_offset(r, g * 8)), v0);
        HAF_X_STORE(reinterpret_cast<half4 *>(img + h_image_offset(r, g * 8 + 4)), v1);
    }
}
__device__ __forceinline__ void store_group_h(char *xtile, int r, int g, half8 hi, half8 lo)
{
    store_group_img(xtile, r, g, hi);
    store_group_img(xtile + kHMatBytes, r, g, lo);
}

// screening operand of one attribute: u' in fp64 (screen_attribute / screen_quad), u^ = fp16(fl32(u')); accumulates |fl32(u')|^2 and |u^ - fl32(u')|^2 in fp32, the two norms the guard band of the screening pass is made of
// Centred form of the band (kernels.h: ScreenParams): three more fp32 sums over the slots -- cr = (u^ - u').G + u'.Hd, the first-order
// error of the evaluation-independent part of the coefficient-weighted kernel vector, which the contraction kernel SUBTRACTS, and
// ub = u'.ubar for |u' - ubar|.  (u^ - u' is exact in fp32, so the first dot product does not cancel.)
struct ScreenSums { float su2, sd2, cr, ub; };
template <class Corr>
__device__ __forceinline__ _Float16 screen_operand(double ud, ScreenSums &a, const Corr &k)
{
    const float f = (float)ud;                       // fl32(u'): |f - u'| <= 2^-24 |u'|
    _Float16 h = (_Float16)f;                        // subnormal results stay: the matrix core multiplies them as they are
#ifdef HAF_FLUSH_F16_SUBNORMALS                      // (checked at haf_create: probe_f16_subnormal_mfma, screen.hip)
    if (fabsf((float)h) < kF16MinNormal) h = (_Float16)0.0f;
#endif
    const float du = (float)h - f;                   // exact in fp32 (h is f rounded to fewer bits, or 0)
    a.su2 = fmaf(f, f, a.su2);                       // all sums in fp32: screen_finish() carries the 326 roundings
    a.sd2 = fmaf(du, du, a.sd2);
    a.cr = fmaf(du, k.g, a.cr);
    a.cr = fmaf(f, k.hd, a.cr);
    a.ub = fmaf(f, k.ub, a.ub);
    return h;
}
typedef const ScrCorr __attribute__((address_space(4))) *ScrCorrK;
__device__ __forceinline__ ScrCorrK constant_ptr(const ScrCorr *p) { return (ScrCorrK)(unsigned long long)p; }

// The same for two slots at once, every sum in packed fp32 (v_pk_fma_f32: two lanes of a sum per instruction, added up in
// screen_sums()): 9 vector instructions per pair -- one packed RN conversion to fp16, two conversions back, a packed subtraction,
// five packed fmas -- where the scalar form costs 17.  The sums only feed the band, whose fp32-accumulation term (kF32Acc) counts
// roundings per summand, not their order.
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 half2v __attribute__((ext_vector_type(2)));
struct ScreenSums2 { f32x2 su2, sd2, cr, ub; };
typedef const ScrCorr2 __attribute__((address_space(4))) *ScrCorr2K;
__device__ __forceinline__ ScrCorr2K constant_ptr(const ScrCorr2 *p) { return (ScrCorr2K)(unsigned long long)p; }
__device__ __forceinline__ half2v screen_operand2(double u0, double u1, ScreenSums2 &a, ScrCorr2K k)
{
    const f32x2 f = {(float)u0, (float)u1};
    half2v h = __builtin_convertvector(f, half2v);   // RN (v_cvt_pk_f16_f32)
#ifdef HAF_FLUSH_F16_SUBNORMALS
    if (fabsf((float)h[0]) < kF16MinNormal) h[0] = (_Float16)0.0f;
    if (fabsf((float)h[1]) < kF16MinNormal) h[1] = (_Float16)0.0f;
#endif
    const f32x2 du = __builtin_convertvector(h, f32x2) - f;
    const f32x2 g = {k->g[0], k->g[1]}, hd = {k->hd[0], k->hd[1]}, ub = {k->ub[0], k->ub[1]};
    a.su2 = __builtin_elementwise_fma(f, f, a.su2);
    a.sd2 = __builtin_elementwise_fma(du, du, a.sd2);
    a.cr = __builtin_elementwise_fma(du, g, a.cr);
    a.cr = __builtin_elementwise_fma(f, hd, a.cr);
    a.ub = __builtin_elementwise_fma(f, ub, a.ub);
    return h;
}
__device__ __forceinline__ ScreenSums screen_sums(const ScreenSums2 &a)
{
    return ScreenSums{a.su2[0] + a.su2[1], a.sd2[0] + a.sd2[1], a.cr[0] + a.cr[1], a.ub[0] + a.ub[1]};
}

// attributes that share a slot beyond the first count once more in |u|^2 (the common factor), not in the operand: sx gets
// extra * u'^2 for the slots of group g that have any (wave-uniform; three slots of the reference's feature file)
__device__ __forceinline__ void screen_extra_norm(const ScreenParams &sp, int g, const double *ud, float &sx)
{
    if (!((sp.extra_groups >> g) & 1)) return;
#pragma unroll
    for (int q = 0; q < 8; q++) {
        const float ex = constant_ptr(sp.sd)[g * 8 + q].extra;
        if (ex != 0.0f) {
            const float f = (float)ud[q];
            sx = fmaf(ex * f, f, sx);
        }
    }
}

// guard band of an evaluation (and -|u|^2/2 for the common factor).  u' = c x' is what the
// feature kernel has (screen_attribute), u = c x the true operand: |u' - u| <= eta := 5e-6 |u'| + tiny, component-wise
// and therefore in norm.  With e_n the error of the exp2 argument of SV n,
//   dec^ + rho = 2^D * sum_n c_n K_n 2^e_n,
//   e_n = (u^-u).w^_n + u.(w^_n - w_n) + [fl32(t_n) - t_n + fp32 accumulation in the matrix core]   (slot space, kernels.h),
//   D   = the error of the common factor 2^(-|u|^2/2) (computed from u' in fp32): the SAME factor for every SV.
// Common factor: dec^ - dec = (2^D - 1)(dec + rho) + 2^D E with E = sum_n c_n K_n (2^e_n - 1); it costs
// (2^D - 1)(|dec^| + |rho|), next to nothing where it matters (dec near 0), instead of D * S.
// E = ln2 * sum_n c_n K_n e_n + second order.  The bilinear part of e_n sums to (u^-u).(V^' w) + u.(dV' w), w_n = c_n K_n,
// and is bounded TWICE:
//   (a) per SV by Cauchy-Schwarz:   <= (|u^-u| max|v^_n| + |u| max|v^_n - v_n|) * S                          =: d_max * S
//   (b) through the spectral norms: <= (|u^-u| sigma(V^) + |u| sigma(dV)) * |w|_2,  |w|_2^2 <= max|c_n| * S   (K_n <= 1)
// (b) grows with sqrt(S) only and is ~7x tighter on a 4096-SV model; the kernel takes the smaller of the two.
// |u^-u| <= |u^-u'| + eta and |u| <= |u'| + eta (norms; |u^-u'| and |u'| are accumulated exactly in fp64).
// Second order: |2^e - 1 - ln2 e| <= 0.6 (ln2 e)^2 for |e| < 0.05.  The bracket is bounded per unit of S: norm split exactly
// (das_max), matrix-core accumulation generously (ten accumulating instructions, kappa u of |c| + sum|products| each:
// ScreenParams::acc_rel).  The kernel measures S^ = 2^D sum|c_n| K_n 2^e_n: the true S is at most S^ * 2^(|D| + max|e_n|),
// folded into the outputs.  Output {gA, gB, gC, cm} (kernels.h), scaled by sp.scale:
//   |dec^ - dec| <= [min(gA |w|_2^, gC S^) + (guard_acc0' + gB) S^ + cm (|dec^| + |rho|)] * 1.002,
//   |w|_2^ = sqrt(max|c_n| S^) or, in the kernel's SUMSQ variant, the measured sqrt(sum_n (c_n K_n)^2)
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

// The band of the CENTRED-REMAINDER form (kernels.h: ScreenParams::cr; derivation in DESIGN.md 2).  The feature kernels are the same
// code: p' = u' - mu through the descriptors' scr_add, su2 / sd2 / sx2 are the norms of p', p^ - p' and p' over all attributes,
// lsum = sum fl32(p'_s) fl32(ln2 g_s) = L in fp32.  With p the TRUE centred operand, dp = p^ - p, dq_n = q^_n - q_n, a_n the rounding of the
// matrix core's fp32 accumulation (|a_n| <= acc_rel |p^||q^_n|), eps_n = dp.q^_n + p.dq_n + a_n the error of z_n:
//   R^ - R = sum b_n (psi(z_n + eps_n) - psi(z_n)),  psi(z) = (ln2 z)^2/2 + psi3(z),  psi3' = ln2 psi >= 0
//   quadratic part:  ln2^2 [p'N dp + p'M p + sum b_n z_n a_n] + ln2^2/2 sum b_n eps_n^2
//                    |.| <= ln2^2 (|N||p||dp| + |M_s||p|^2 + acc_rel |p^||p| C_a) + 1.5 ln2^2 (|H_abs||dp|^2 + |D_abs||p|^2 + acc_rel^2|p^|^2 C_qq)
//   the rest:        |sum b_n (psi3(z^_n) - psi3(z_n))| <= ln2 eps_max sum|b_n| psi(xi_n),  psi(xi) <= psi(z^) + ln2 (2^zmax - 1) eps_max
// Output {L, c_abs, k_psi, cm}: the contraction kernel forms dec^ = A^ (B0 + L + R^) - rho and trusts it when
//   |dec^| > [A^ (c_abs + (guard_acc0' + k_psi) S_psi^) + cm (|dec^| + |rho|)] * 1.002 + guard_abs,   S_psi^ = sum|b_n| psi^(z^_n) as measured.
__device__ __forceinline__ void screen_finish_cr(double su2, double sd2, double sx2, double lsum, const ScreenParams &sp, float *band, float &nax)
{
    constexpr double kF32Acc = 326.0 * 5.9604644775390625e-08 * 1.01;
    const double ln2 = 0.69314718056;
    const double a_x = 0.5 * sx2;
    nax = (float)(-a_x);
    const double un_t = sqrt_upper(sx2 * (1.0 + kF32Acc));                                     // |p'| over all attributes
    const double eta_t = kScreenEtaRel * (un_t + sp.cr_mu_norm_t) + sp.eta_abs;                // |p' - p| = |u' - u| <= 5e-6 |u'| + ..., |u'| <= |p'| + |mu|
    const double un1 = sqrt_upper(su2 * (1.0 + kF32Acc));                                      // |p'| in slot space
    const double dn1 = sqrt_upper(sd2 * (1.0 + kF32Acc)) + 5.97e-8 * un1 + 1e-17;              // |p^ - p'|
    const double eta = kScreenEtaRel * (un1 + sp.cr_mu_norm) + sp.eta_abs;
    const double un = un1 + eta, dn = dn1 + eta, ph = un1 + dn1;                               // |p|, |p^ - p|, |p^|
    const double D = (kF32Acc + 6.0e-8) * a_x + un_t * eta_t + 0.5 * eta_t * eta_t + 6.0e-7;   // error of the common factor's exponent (screen_finish)
    const double eps = dn * sp.cr_qmax + un * sp.cr_dqmax + sp.acc_rel * ph * sp.cr_qmax;      // sup_n |eps_n|
    const double zmax = ph * sp.cr_qmax + eps;                                                 // sup_n of |z^_n| and |z_n|
    const double zf = floor(zmax);
    const double p2 = (zmax < 60.0) ? ldexp(1.0 + (zmax - zf), (int)zf) : (double)__builtin_inff();   // >= 2^zmax (chord of the convex 2^x)
    // accumulation inside the matrix core: |a_n| <= acc_rel sum_k|p^_k q^_nk|; per SV through Cauchy-Schwarz (C_a) or over the SVs
    // through the spectral norms of sqrt|b| Q and sqrt|b| |Q^| (kernels.h: cr_nHaa), whichever is smaller
    const double sHq = sqrt_upper(sp.cr_nHabs) + sqrt_upper(sp.cr_nDabs);
    const double acc_sum = fmin(ph * un * sp.cr_Ca, sHq * un * sqrt_upper(sp.cr_nHaa) * ph);
    const double quad1 = ln2 * ln2 * (sp.cr_nN * un * dn + sp.cr_nM * un * un + sp.acc_rel * acc_sum);
    const double quad2 = 1.5 * ln2 * ln2 * (sp.cr_nHabs * dn * dn + sp.cr_nDabs * un * un + sp.acc_rel * sp.acc_rel * ph * ph * sp.cr_Cqq);
    const double cub2 = ln2 * ln2 * (p2 - 1.0) * eps * eps * sp.cr_Babs * 1.01;
    // L: fl32 of p' and of ln2 g, 320 fp32 fmas (|sum of the terms' magnitudes| <= |p'||g|), p' against p
    const double cL = ln2 * sp.cr_gnorm * (eta + 330.0 * 5.97e-8 * un1) * 1.01 + 1.2e-7 * fabs(lsum);
    double c_abs = quad1 + quad2 + cub2 + cL;
    double k_psi = ln2 * eps * 1.01;
    if (sp.cr_poly) {
        // psi(t) = t^2 (1/2 + t/6 + t^2/24 + t^3/120), t = z ln2: the dropped tail is at most 4.1 t^4/360 of psi for |t| <= 1; the fp32
        // roundings per element -- z^2, the four constants b a_k (folded per block), three Horner steps whose partial sums are at most
        // 1.95 P(z) for z < 0 -- come to less than 8 u of |b| psi; 10 u charged
        const double t = ln2 * zmax;
        k_psi += 4.1 * t * t * t * t / 360.0 + 10.0 * 5.97e-8;
        if (!(t <= 1.0)) k_psi = (double)__builtin_inff();
    } else {
        // 2^z by v_exp_f32 (an ulp of 2^z: taken as 2^-22), the constant ln2 in fp32: relative to sum|b_n| 2^z_n <= S_psi + B_abs + ln2 sum|b_n||z_n|
        const double uexp = 2.4e-7;
        k_psi += uexp;
        c_abs += uexp * (sp.cr_Babs + ln2 * ph * sp.cr_Cq1) * 1.01;
    }
    const double infl = 1.0 + exp2m1_upper(D);
    band[0] = (float)lsum;
    band[1] = (float)(c_abs * infl * sp.scale * (1.0 + 1e-6));
    band[2] = (float)(k_psi * infl * sp.scale * (1.0 + 1e-6));
    band[3] = (float)(exp2m1_upper(D) * sp.scale);
    band[4] = 0.0f; band[5] = 0.0f; band[6] = 0.0f; band[7] = 0.0f;
    if (!(D < 0.05) || !(a_x < 30.0) || !(zmax < 60.0)) band[1] = __builtin_inff();
}

__device__ __forceinline__ void screen_finish(double su2, double sd2, double sx2, double cr, double ubd, const ScreenParams &sp, float *band,
                                              float &nax)
{
    if (sp.cr) {                                         // wave-uniform: the centred-remainder form has its own band
        screen_finish_cr(su2, sd2, sx2, cr, sp, band, nax);
        return;
    }
    // su2 = sum over the SLOTS of fl32(u')^2 and sd2 = sum over the slots of (u^ - fl32(u'))^2: the two norms of the operand the
    // contraction sees (kernels.h: attributes that share a slot are one operand).  sx2 = su2 + the squares of the attributes
    // beyond the first of every slot = |u'|^2 over ALL attributes, which is what the common factor 2^(-|u|^2/2) needs.
    // All three are fp32 sums (screen_operand) of at most 324 squares of fp32-rounded terms: off by at most 326 * 2^-24
    // relative.  For the norms that is an inflation; for a_x it is one more part of D, the error of the common factor, and costs
    // (2^D - 1)(|dec^| + |rho|) like the rest of D.
    constexpr double kF32Acc = 326.0 * 5.9604644775390625e-08 * 1.01;
    const double a_x = 0.5 * sx2;
    nax = (float)(-a_x);                                 // k_svm_screen multiplies both class sums by exp2(nax)
    const double un_t = sqrt_upper(sx2 * (1.0 + kF32Acc));                        // |u'| over all attributes
    const double eta_t = kScreenEtaRel * un_t + sp.eta_abs;                      // |u' - u| over all attributes
    const double un1 = sqrt_upper(su2 * (1.0 + kF32Acc));                         // |u'| in slot space
    const double dn1 = sqrt_upper(sd2 * (1.0 + kF32Acc)) + 5.97e-8 * un1 + 1e-17;  // |u^ - u'| <= |u^ - fl32(u')| + 2^-24 |u'|
    const double eta = kScreenEtaRel * un1 + sp.eta_abs;                        // |u' - u| in slot space
    const double un = un1 + eta, dn = dn1 + eta;                                // |u|, |u^ - u|
    const double ln2 = 0.69314718056;
    const double d_max = dn * sp.v_max + (un + dn) * sp.dv_max;
    // fp32 accumulation inside the matrix core: ten accumulating instructions per element, each off by at most kappa u of its
    // |c| + sum|products| <= |t_n| + |u||w^_n| (the chain starts at t_n; kappa: the measured property of screen.hip's
    // probe_mfma_rounding() with its margin, 8.2 on the devices seen so far: sp.acc_rel = 82 u)
    const double acc = sp.acc_rel * (un * sp.v_max + sp.as_max);
    // D = | log2 of (the factor the kernel applies / 2^(-|u|^2/2)) |: a_x from the fp32 sums and u' instead of u, its cast to
    // fp32, v_exp_f32 and the two products (3 * 2^-23 relative = 5.2e-7 in the exponent)
    const double D = (kF32Acc + 6.0e-8) * a_x + un_t * eta_t + 0.5 * eta_t * eta_t + 6.0e-7;
    const double e_max = d_max + sp.das_max + acc;
    const double infl = 1.0 + exp2m1_upper(e_max + D);   // meaningful below 0.05 only: beyond it the band is infinite anyway
    const double gA = ln2 * (dn * sp.sigma_v + (un + dn) * sp.sigma_dv);   // per unit of |w|_2, which the contraction kernel supplies
    const double gB = ln2 * (sp.das_max + acc) + 0.6 * (ln2 * e_max) * (ln2 * e_max);
    band[0] = (float)(gA * infl * sp.scale);             // (|w|_2 measured: inflated like S; bounded through sqrt(S): sqrt(infl) <= infl)
    band[1] = (float)(gB * infl * sp.scale);             // scale = 1.001: the roundings of these expressions and of the casts are far inside 0.1 %
    band[2] = (float)(ln2 * d_max * infl * sp.scale);
    band[3] = (float)(exp2m1_upper(D) * sp.scale);
    // outside the range the bounds were derived for, or a common factor 2^(a_x) that fp32 sums could overflow on: never trusted
    // (2^(2 a_x) must stay finite in fp32 for the SUMSQ variant's sum of squares)
    if (!(e_max + D < 0.05) || !(a_x < 30.0)) band[1] = __builtin_inff();
    // ---- the centred estimate dec^ - corr * sc (kernels.h: ScreenParams; derivation in DESIGN.md 2) ----
    // w_n = c_n K_n = sc c_n kappa_n + sc c_n (k_n - kappa_n), kappa_n = 2^(t_n + ubar.w^_n), k_n = 2^(t_n + u.w_n) (raw space, TRUE
    // operands).  First-order error of the first part: ln2 sc [(u^-u).G + u.Hd], known up to u' - u and fp32 roundings: corrected.
    // Second part: k_n - kappa_n = kappa_n (2^zeta_n - 1), zeta_n = u.w_n - ubar.w^_n = (u^ - ubar).w^_n - e_n with e_n the bilinear
    // part of the exp2 argument's error, |e|_2 <= |u^-u| sigma(W^) + |u| sigma(dW) =: e2 and |e_n| <= d_max.  With
    // |2^zeta - 1| <= ln2 |zeta| 2^|zeta|:   |c (k - kappa)|_2 <= ln2 2^zmax (sigma(diag(c kappa) W^) |u^ - ubar| + max|c kappa| e2).
    band[4] = 0.0f; band[5] = __builtin_inff(); band[6] = 0.0f; band[7] = 0.0f;
    if (sp.sigma_dk < 1e300) {
        // |fl32(u') - ubar|^2 from the fp32 sums: each is off by at most kF32Acc of the sum of its terms' magnitudes
        const double ubn = sqrt_upper(sp.ubar2);
        double du2 = su2 - 2.0 * ubd + sp.ubar2 + kF32Acc * (su2 + 2.0 * un1 * ubn) + 1e-30;
        if (!(du2 > 0.0)) du2 = (du2 == du2) ? 0.0 : du2;                        // (NaN stays NaN: never trusted)
        const double dun = sqrt_upper(du2) + dn1;                                // |u^ - ubar| <= |fl32(u') - ubar| + |u^ - fl32(u')|
        const double e2 = dn * sp.sigma_v + (un + dn) * sp.sigma_dv;
        const double zmax = dun * sp.v_max + e_max;                              // sup_n |zeta_n|
        const double zf = floor(zmax);
        // 2^zmax <= (1 + frac) 2^floor: the chord of the convex 2^x over [0, 1] (no transcendental instruction: see sqrt_upper)
        const double p2 = (zmax < 60.0) ? ldexp(1.0 + (zmax - zf), (int)zf) : (double)__builtin_inff();
        const double dev = ln2 * p2 * (sp.sigma_dk * dun + sp.ck_max * e2);
        // what the computed correction misses: u' against u in both dot products ((u'-u).(G - Hd)), the 2 x 320 fp32 roundings of
        // its accumulation and the fp32 rounding of the constants
        const double cerr = ln2 * (eta * (sp.g_norm + sp.hd_norm) + 4.2e-5 * (dn1 * sp.g_norm + un1 * sp.hd_norm));
        band[4] = (float)(ln2 * cr);
        band[5] = (float)((ln2 * e2 * dev * infl + cerr) * sp.scale);
        if (!(e_max + D < 0.05) || !(a_x < 30.0) || !(zmax < 60.0)) band[5] = __builtin_inff();
    }
}

__device__ __forceinline__ void store_band(float *dst, const float *band)
{
    static_assert(kBandFloats == 8, "two 16-byte stores");
    reinterpret_cast<float4 *>(dst)[0] = float4{band[0], band[1], band[2], band[3]};
    reinterpret_cast<float4 *>(dst)[1] = float4{band[4], band[5], band[6], band[7]};
}

// ---- XMODE_I8: the int8 digit image of the exact-integer tier (exact8.hip; kernels.h "tier 2a") ----
constexpr int kI8SmallList = 16384;                    // lists up to this length: k_features_small, beyond: k_features<.., 16>
// fixed point with kI8Q fractional bits, round to nearest (the scaling by 2^kI8Q is exact): |x - X 2^-kI8Q| <= 2^-(kI8Q+1); balanced
// base-128 digits, d in [-64, 63], X = ((d0 128 + d1) 128 + d2) 128 + d3; attribute q of this thread's group goes to byte q of the
// four digit planes.  xx: sum of X^2 (exact in int64: < 324 * 2^54); ovf: an attribute beyond the fixed-point range (or NaN).
__device__ __forceinline__ void i8_digits(double xd, int q, unsigned long long (&dig)[4], long long &xx, int &ovf)
{
    double sc = rint(xd * (double)(1 << kI8Q));
    if (!(fabs(sc) <= (double)kI8Max)) { ovf = 1; sc = 0.0; }
    const int X = (int)sc;
    xx += (long long)X * (long long)X;
    int t = X;
    const int d3 = ((t + 64) & 127) - 64; t = (t - d3) >> 7;
    const int d2 = ((t + 64) & 127) - 64; t = (t - d2) >> 7;
    const int d1 = ((t + 64) & 127) - 64; t = (t - d1) >> 7;
    const int d0 = t;
    dig[0] |= (unsigned long long)(unsigned char)d0 << (8 * q);
    dig[1] |= (unsigned long long)(unsigned char)d1 << (8 * q);
    dig[2] |= (unsigned long long)(unsigned char)d2 << (8 * q);
    dig[3] |= (unsigned long long)(unsigned char)d3 << (8 * q);
}
// A-operand image of v_mfma_i32_16x16x64_i8 (checked on hardware: testkernels.hip): lane = 16 (k % 64 / 16) + row holds bytes
// k % 16 = 0..15; the attributes 8g..8g+7 of slot e are half of one lane's fragment: one 8-byte store per digit plane.  The groups
// 44..47 (attributes 352..383: padding of the sixth k-step) have no thread of their own: the threads of groups 40..43 zero them.
__device__ __forceinline__ void i8_store(float *X, long e, int g, const unsigned long long (&dig)[4])
{
    char *img = reinterpret_cast<char *>(X) + (size_t)(e >> 4) * kI8GroupBytes;
    const int row = (int)(e & 15);
#pragma unroll
    for (int j = 0; j < kI8Slices; j++) {
        *reinterpret_cast<unsigned long long *>(img + (j * kI8Steps + (g >> 3)) * 1024 + (((g & 7) >> 1) * 16 + row) * 16 + (g & 1) * 8) = dig[j];
        if (g >= 40) {
            const int g2 = g + 4;
            *reinterpret_cast<unsigned long long *>(img + (j * kI8Steps + (g2 >> 3)) * 1024 + (((g2 & 7) >> 1) * 16 + row) * 16 + (g2 & 1) * 8) = 0ull;
        }
    }
}

// Large requests: one thread per evaluation walks all attributes (best throughput: no per-workgroup tail, 35 k
// workgroups for C5).  Small requests use k_features below.
template <int MODE>
__global__ __launch_bounds__(256) void k_features_serial(const float *__restrict__ ii, const int *__restrict__ evalcell,
                                                  const int *__restrict__ counters, const FeatDesc *__restrict__ fd,
                                                  float *__restrict__ X, float *__restrict__ ax, Dims d, double lower,
                                                  double upper, float neg_gamma2, ScreenParams sp,
                                                  const int *__restrict__ idx_list, int list_counter, int list_cap,
                                                  AttrRecord *__restrict__ dbg, float *__restrict__ ax2)
{
    constexpr int kBlock = (MODE == XMODE_SCREEN) ? kS0BlockEvals : kSvmBlockEvals;
    const int n_evals = idx_list ? min(counters[list_counter], list_cap) : counters[CNT_EVALS];
    const long n_pad = ((long)n_evals + kBlock - 1) / kBlock * kBlock;
    const long e = (long)blockIdx.x * 256 + threadIdx.x;
    if ((long)blockIdx.x * 256 >= n_pad) return;
    __shared__ double s_tab[MODE == XMODE_SCREEN ? 1 : hafq::kTabDoubles];
    __shared__ unsigned long long s_scr[MODE == XMODE_SCREEN ? hafq::kScrTabWords : 1];
    hafq::PtrTabs tb{};
    hafq::ScrTabs st{};
    if (MODE == XMODE_SCREEN) st = load_screen_tables(s_scr);
    else tb = load_decimal_tables(s_tab);
    float *xcol = X + (size_t)(e >> 5) * kTileFloats + (e & 31);
    char *xtile = reinterpret_cast<char *>(X) + (size_t)(e >> 5) * (MODE == XMODE_SCREEN ? kS0MatBytes : kHXTileBytes);
    const int r = (int)(e & 31);
    // screening form: is this wave 64 neighbouring cells of one row?  (cell ids are row-major and a masked cell is never in
    // the first or last 7 columns, so consecutive ids are neighbours in one row)
    __shared__ float s_band[MODE == XMODE_SCREEN ? (256 / 64) * kBandFloats4 : 1];
    bool fastwave = false;
    unsigned band = 0;
    if (MODE == XMODE_SCREEN && !idx_list) {
        const int lane = threadIdx.x & 63;
        const int cell = (e < n_evals) ? evalcell[e] : -1;
        const int cell0 = __builtin_amdgcn_readfirstlane(cell);
        fastwave = __ballot(cell >= 0 && cell == cell0 + lane) == ~0ull;
        if (fastwave) {
            const rsrc_t iir0 = make_ii_rsrc(ii, d);
            const unsigned w0l = window_origin(cell, d.H, d.W);            // this lane's window origin
            float *bw = s_band + (threadIdx.x >> 6) * kBandFloats4;
            const int ldb = (d.W + 1) * 4;
#pragma unroll
            for (int x = 0; x < kBandRows; x++) {
                bw[x * kBandPitch + lane] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(iir0, (int)w0l, x * ldb, 0));
                if (lane < 14)
                    bw[x * kBandPitch + 64 + lane] =
                        __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(iir0, (int)w0l, x * ldb + 256, 0));
            }
            band = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(size_t)bw);
            asm volatile("" :: "v"(bw) : "memory");       // the band is read by asm only: keep its stores, and keep them here
        }
    }
    if (e >= n_evals) {                       // padding rows of the last block: zeros
        const half8 z = {0, 0, 0, 0, 0, 0, 0, 0};
        if (MODE == XMODE_SPLIT) {
            for (int g = 0; g < 2 * kHSteps; g++) store_group_h(xtile, r, g, z, z);
        } else if (MODE == XMODE_SCREEN) {
            for (int g = 0; g < kS0Groups; g++) store_group_img(xtile, r, g, z);
        } else {
            for (int k = 0; k < kKP; k++) xcol[k * kTile] = 0.0f;
        }
        if (MODE == XMODE_SCREEN) { const float zb[kBandFloats] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f}; store_band(ax + kBandFloats * e, zb); ax2[e] = 0.0f; }
        else ax[e] = 0.0f;
        return;
    }
    const rsrc_t iir = make_ii_rsrc(ii, d);
    const int e_src = idx_list ? idx_list[e] : (int)e;                 // the evaluation this slot holds
    const unsigned w0 = window_origin(evalcell[e_src], d.H, d.W);
    AttrRecord *rec = (MODE != XMODE_SCREEN && dbg) ? dbg + (size_t)e_src * kKP : nullptr;   // KEEP_DEBUG only
    double xx = 0.0;
    if (MODE == XMODE_SCREEN) {
        ScreenSums2 acc2{};
        float sx = 0.0f;
        for (int g = 0; g < kS0Groups; g++) {             // 40 groups of 8 SLOTS (kernels.h)
            double ud[8];
            if (fastwave && ((sp.fast_groups >> g) & 1)) {   // wave-uniform
                screen_quad(band, constant_ptr(sp.sd) + g * 8, st, ud);
                screen_quad(band, constant_ptr(sp.sd) + g * 8 + 4, st, ud + 4);
            } else if (fastwave) {
#pragma unroll
                for (int q = 0; q < 8; q += 2) screen_pair3(band, constant_ptr(sp.sd3) + g * 8 + q, st, ud + q);
            } else {
#pragma unroll
                for (int q = 0; q < 8; q++) {
                    const FeatDesc &F = fd[g * 8 + q];               // screening form: fd = one descriptor per SLOT (an unused slot has skip = 1)
                    ud[q] = F.skip ? 0.0 : screen_attribute(SrcBuf<true>{iir, w0}, F, st);
                }
            }
            half8 hi;
#pragma unroll
            for (int q = 0; q < 8; q += 2) {
                const half2v h = screen_operand2(ud[q], ud[q + 1], acc2, constant_ptr(sp.corr2) + g * 4 + (q >> 1));
                hi[q] = h[0]; hi[q + 1] = h[1];
            }
            screen_extra_norm(sp, g, ud, sx);
            store_group_img(xtile, r, g, hi);
        }
        const ScreenSums acc = screen_sums(acc2);
        float band[kBandFloats], nax;
        screen_finish((double)acc.su2, (double)acc.sd2, (double)acc.su2 + (double)sx, (double)acc.cr, (double)acc.ub, sp, band, nax);
        store_band(ax + kBandFloats * e, band);
        ax2[e] = nax;
        return;
    }
    if (MODE == XMODE_SPLIT) {
        for (int g = 0; g < 2 * kHSteps; g++) {           // 42 groups of 8 attributes
            half8 hi, lo;
#pragma unroll
            for (int q = 0; q < 8; q++) {
                const int f = g * 8 + q;
                float xf = 0.0f;
                if (f < d.nf) xf = (float)attribute_value_rec(SrcBuf<true>{iir, w0}, fd[f], lower, upper, tb, rec ? rec + f : nullptr);
                const _Float16 h = (_Float16)xf;                       // RN
                const _Float16 l = (_Float16)(xf - (float)h);          // exact difference, then RN
                hi[q] = h;
                lo[q] = l;
                const float xe = (float)h + (float)l;                  // the value the three passes actually multiply
                xx = fma((double)xe, (double)xe, xx);
            }
            store_group_h(xtile, r, g, hi, lo);
        }
    } else {
        for (int f = 0; f < d.nf; f++) {
            const float xf = (float)attribute_value_rec(SrcBuf<true>{iir, w0}, fd[f], lower, upper, tb, rec ? rec + f : nullptr);
            xcol[f * kTile] = xf;
            xx = fma((double)xf, (double)xf, xx);
        }
        for (int k = d.nf; k < kKP; k++) xcol[k * kTile] = 0.0f;
    }
    ax[e] = neg_gamma2 * (float)xx;           // -gamma*log2(e)*|x|^2, folded into the exp2 argument
}

// Workgroup = 64 evals x 8 waves: wave w evaluates the groups of 8 attributes w, w+8, ... (42 groups of 8 = 336
// attribute slots) for the same 64 evals, so the attribute index stays wave-uniform (feature descriptors by scalar
// loads, no divergence), a small request (a few thousand evals) still fills the chip, and a single evaluation is
// never one long serial chain of 324 attributes.
constexpr int kFeatEvals = 64;
constexpr int kWinPitch = 225;        // floats per staged window (15 x 15); odd, so the 64 lanes of a read hit 32 banks twice over
// kFeatWaves = 8 or 16 waves per workgroup, each taking the attribute groups w, w + kFeatWaves, ...: 16 halves the serial
// chain of a thread (a few thousand evaluations, the refinement list), 8 keeps more evaluations resident when there are
// enough of them to fill the chip.
template <int MODE, int kFeatWaves>
__global__ __launch_bounds__(kFeatWaves * 64) void k_features(const float *__restrict__ ii, const int *__restrict__ evalcell,
                                                  const int *__restrict__ counters, const FeatDesc *__restrict__ fd,
                                                  float *__restrict__ X, float *__restrict__ ax, Dims d, double lower,
                                                  double upper, float neg_gamma2, ScreenParams sp,
                                                  const int *__restrict__ idx_list, int list_counter, int list_cap,
                                                  AttrRecord *__restrict__ dbg, float *__restrict__ ax2, int list_off)
{
    // XMODE_F64: the fp64 attribute image of the fp64 MFMA tier (k_recheck_mfma), [group of 16 slots][324][16] doubles, for a
    // window [list_off, list_off + list_cap) of the tier's list (idx_list already points at entry list_off)
    constexpr int kBlock = (MODE == XMODE_SCREEN) ? kS0BlockEvals : (MODE == XMODE_F64 || MODE == XMODE_I8) ? 64 : kSvmBlockEvals;
    __shared__ long long red_ll[(MODE == XMODE_I8) ? kFeatWaves : 1][kFeatEvals];
    __shared__ int red_ovf[(MODE == XMODE_I8) ? kFeatWaves : 1][kFeatEvals];
    constexpr int kFeatFinisher = 0;                          // the wave that sums up the partial norms
    __shared__ double red[kFeatWaves][kFeatEvals];
    __shared__ float s_win[kFeatEvals * kWinPitch];
    __shared__ unsigned s_w0[kFeatEvals];
    __shared__ double red2[(MODE == XMODE_SCREEN) ? kFeatWaves : 1][kFeatEvals];
    __shared__ double red3[(MODE == XMODE_SCREEN) ? kFeatWaves : 1][kFeatEvals];
    __shared__ float red4[(MODE == XMODE_SCREEN) ? kFeatWaves : 1][kFeatEvals], red5[(MODE == XMODE_SCREEN) ? kFeatWaves : 1][kFeatEvals];
    // XMODE_SPLIT in the centred-remainder form (tier 1 behind SCREEN_CR_POLY, ScreenParams::cr_t1_tab): the centre is subtracted from
    // the exact attribute (fp64) before the hi/lo split and L = sum (x_f - m_f) gl_f is summed in fp64
    __shared__ double red_l[(MODE == XMODE_SPLIT) ? kFeatWaves : 1][kFeatEvals];
    const double *t1_tab = (MODE == XMODE_SPLIT) ? sp.cr_t1_tab : nullptr;
    const int n_evals = idx_list ? window_count(counters[list_counter], list_off, list_cap) : counters[CNT_EVALS];
    if (MODE == XMODE_I8 && n_evals <= kI8SmallList) return;           // short lists are k_features_small's (see there)
    const long n_pad = ((long)n_evals + kBlock - 1) / kBlock * kBlock;
    if ((long)blockIdx.x * kFeatEvals >= n_pad) return;
    __shared__ double s_tab[MODE == XMODE_SCREEN ? 1 : hafq::kTabDoubles];
    __shared__ unsigned long long s_scr[MODE == XMODE_SCREEN ? hafq::kScrTabWords : 1];
    hafq::PtrTabs tb{};
    hafq::ScrTabs st{};
    if (MODE == XMODE_SCREEN) st = load_screen_tables(s_scr);
    else tb = load_decimal_tables(s_tab);
    const int ev = threadIdx.x & 63, gl = threadIdx.x >> 6;
    // grid-stride over blocks of 64 evaluations: a list launch is sized for a few thousand workgroups, not for the list's
    // capacity (tens of thousands of workgroups that would only find out that there is nothing for them)
    for (long blk = blockIdx.x; blk * kFeatEvals < n_pad; blk += gridDim.x) {
    const long e = blk * kFeatEvals + ev;
    const long tile = e >> 5;
    const int r = (int)(e & 31);
    float *xcol = X + (size_t)tile * kTileFloats + (e & 31);
    char *xtile = reinterpret_cast<char *>(X) + (size_t)tile * (MODE == XMODE_SCREEN ? kS0MatBytes : kHXTileBytes);
    const int n_groups = (MODE == XMODE_I8) ? 44 : (MODE == XMODE_F32 || MODE == XMODE_F64) ? (kKP + 7) / 8 : (MODE == XMODE_SCREEN) ? kS0Groups : 2 * kHSteps;   // 44 / 41 / 40 / 42
    double *x64 = reinterpret_cast<double *>(X) + (size_t)(e >> 4) * kKP * 16 + (e & 15);
    const bool live = e < n_evals;
    const rsrc_t iir = make_ii_rsrc(ii, d);
    const int e_src = live ? (idx_list ? idx_list[e] : (int)e) : 0;  // the evaluation this slot holds
    const unsigned w0 = live ? window_origin(evalcell[e_src], d.H, d.W) : 0u;
    AttrRecord *rec = (MODE != XMODE_SCREEN && dbg && live) ? dbg + (size_t)e_src * kKP : nullptr;   // KEEP_DEBUG only
    // The 15x15 windows of the block's 64 evaluations go to LDS first (every wave works on the same 64): the evaluations of
    // a list are scattered cells, so a corner load of 64 lanes is 64 separate L1 accesses, ~2700 times per evaluation and
    // wave group -- the vector L1 was what bounded this kernel.  Staged, a window row is one or two accesses, once.
    if (gl == 0) s_w0[ev] = live ? w0 : 0xffffffffu;
    __syncthreads();
    for (int idx = threadIdx.x; idx < kFeatEvals * 15 * 16; idx += kFeatWaves * 64) {
        const int col = idx & 15, seg = idx >> 4, wev = seg & (kFeatEvals - 1), x = seg >> 6;      // 16 lanes = one window row
        const unsigned o = s_w0[wev];
        if (col < 15)
            s_win[wev * kWinPitch + x * 15 + col] =
                (o != 0xffffffffu) ? __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(iir, (int)(o + (unsigned)(x * (d.W + 1) + col) * 4u), 0, 0)) : 0.0f;
    }
    __syncthreads();
    const SrcWin src{s_win + ev * kWinPitch};
    double xx = 0.0;
    ScreenSums acc{0.0f, 0.0f, 0.0f, 0.0f};                            // screening form: fp32 partial sums of this wave's groups
    float sx = 0.0f;
    long long xx_ll = 0;                                               // XMODE_I8 (see i8_digits)
    int ovf = 0;
    double lsum = 0.0;
    for (int g = gl; g < n_groups; g += kFeatWaves) {
        half8 hi = {0, 0, 0, 0, 0, 0, 0, 0}, lo = {0, 0, 0, 0, 0, 0, 0, 0};
        double udv[8];
        unsigned long long dig[4] = {0, 0, 0, 0};
#pragma unroll
        for (int q = 0; q < 8; q++) {
            const int f = g * 8 + q;
            double xd = 0.0;
            if (MODE == XMODE_SCREEN) {
                const FeatDesc &F = fd[f];                               // screening form: fd = one descriptor per SLOT (an unused slot has skip = 1)
                if (live && !F.skip) xd = screen_attribute(src, F, st);  // u' = c x', not x'
            } else if (live && f < d.nf) {
                xd = attribute_value_rec(src, fd[f], lower, upper, tb, rec ? rec + f : nullptr);
                if (MODE == XMODE_SPLIT && t1_tab && f < kKP) {          // (wave-uniform f: the two constants come by scalar loads)
                    xd -= t1_tab[f];
                    lsum = fma(xd, t1_tab[kKP + f], lsum);
                }
            }
            udv[q] = xd;
            const float xf = (float)xd;
            if (MODE == XMODE_SCREEN) {
                hi[q] = screen_operand(xd, acc, constant_ptr(sp.corr)[f]);
            } else if (MODE == XMODE_SPLIT) {
                const _Float16 h = (_Float16)xf;                       // RN
                const _Float16 l = (_Float16)(xf - (float)h);          // exact difference, then RN
                hi[q] = h;
                lo[q] = l;
                const float xe = (float)h + (float)l;                  // the value the three passes actually multiply
                xx = fma((double)xe, (double)xe, xx);
            } else if (MODE == XMODE_F64) {
                if (f < kKP) x64[(size_t)f * 16] = xd;                  // unused slots and attributes beyond the feature file: zeros
            } else if (MODE == XMODE_I8) {
                i8_digits(xd, q, dig, xx_ll, ovf);
            } else {
                if (f < kDP) xcol[f * kTile] = xf;                     // rows >= nf (padding up to the tile image) are zero
                xx = fma((double)xf, (double)xf, xx);
            }
        }
        if (MODE == XMODE_SPLIT) store_group_h(xtile, r, g, hi, lo);
        if (MODE == XMODE_SCREEN) {
            screen_extra_norm(sp, g, udv, sx);
            store_group_img(xtile, r, g, hi);
        }
        if (MODE == XMODE_I8) i8_store(X, e, g, dig);
    }
    if (MODE == XMODE_I8) { red_ll[gl][ev] = xx_ll; red_ovf[gl][ev] = ovf; }
    if (MODE == XMODE_SPLIT) red_l[gl][ev] = lsum;
    red[gl][ev] = (MODE == XMODE_SCREEN) ? (double)acc.su2 : xx;
    if (MODE == XMODE_SCREEN) { red2[gl][ev] = (double)acc.sd2; red3[gl][ev] = (double)sx; red4[gl][ev] = acc.cr; red5[gl][ev] = acc.ub; }
    __syncthreads();
    if (gl == kFeatFinisher) {
        double t = 0.0, t2 = 0.0, t3 = 0.0, t4 = 0.0, t5 = 0.0;
#pragma unroll
        for (int k = 0; k < kFeatWaves; k++) t += red[k][ev];         // fixed order: deterministic
        if (MODE == XMODE_SCREEN) {
#pragma unroll
            for (int k = 0; k < kFeatWaves; k++) { t2 += red2[k][ev]; t3 += red3[k][ev]; t4 += (double)red4[k][ev]; t5 += (double)red5[k][ev]; }
            float band[kBandFloats] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f}, nax = 0.0f;
            if (live) screen_finish(t, t2, t + t3, t4, t5, sp, band, nax);
            store_band(ax + kBandFloats * e, band);
            ax2[e] = nax;
        } else if (MODE == XMODE_I8) {
            long long s2 = 0;
            int any = 0;
#pragma unroll
            for (int k2 = 0; k2 < kFeatWaves; k2++) { s2 += red_ll[k2][ev]; any |= red_ovf[k2][ev]; }      // exact: < 324 * 2^54
            reinterpret_cast<double *>(ax)[e] = any ? -1.0 : ldexp((double)s2, -2 * kI8Q);
        } else if (MODE != XMODE_F64) {
            ax[e] = neg_gamma2 * (float)t;                             // -gamma*log2(e)*|x|^2, folded into the exp2 argument
            if (MODE == XMODE_SPLIT && t1_tab) {
                double l = 0.0;
#pragma unroll
                for (int k = 0; k < kFeatWaves; k++) l += red_l[k][ev];   // fixed order
                sp.cr_t1_L[e] = l;
            }
        }
    }
    __syncthreads();                                                   // red / red2 are reused by the next block
    }
}

// Small requests (a few thousand evaluations: the reference's own 56 x 56 grid): k_features would occupy one CU per 64
// evaluations and leave most of the chip idle while each thread walks three groups of attributes.  Here a workgroup takes 16
// evaluations and a quarter wave one group of 8 attributes of them: 44 quarter waves cover the 41 / 42 groups at once, four
// times as many workgroups, a third of the chain per thread.  The attribute index differs between the quarters of a wave, so
// the descriptors come by vector loads (four addresses per wave) and the HAF / SHAF branch may diverge in the one group
// where both occur.  Same arithmetic, same operand images.
constexpr int kSmEvals = 16;
constexpr int kSmWaves = 12;
constexpr int kSmSlots = kSmWaves * 4;                 // quarter waves: >= 42 attribute groups
constexpr long kSmallEvals = 12288;                    // requests of up to this many evaluations (host estimate) take k_features_small

template <int MODE>
__global__ __launch_bounds__(kSmWaves * 64) void k_features_small(const float *__restrict__ ii, const int *__restrict__ evalcell,
                                                  const int *__restrict__ counters, const FeatDesc *__restrict__ fd,
                                                  float *__restrict__ X, float *__restrict__ ax, Dims d, double lower,
                                                  double upper, float neg_gamma2, ScreenParams sp,
                                                  AttrRecord *__restrict__ dbg, float *__restrict__ ax2,
                                                  const int *__restrict__ idx_list, int list_counter, int list_cap, int list_off)
{
    // XMODE_I8 (list mode): the int8 digit image of the exact-integer tier (exact8.hip), kI8GroupBytes per 16 slots, and |xq|^2
    constexpr int kBlock = (MODE == XMODE_SCREEN) ? kS0BlockEvals : (MODE == XMODE_F64 || MODE == XMODE_I8) ? 64 : kSvmBlockEvals;
    __shared__ long long red_ll[(MODE == XMODE_I8) ? kSmSlots : 1][kSmEvals];
    __shared__ int red_ovf[(MODE == XMODE_I8) ? kSmSlots : 1][kSmEvals];
    constexpr int kFinisher = 40;                     // the quarter wave that sums up the partial norms (one without a group of its own in the screening form)
    static_assert(kSmSlots >= 44 && kSmSlots >= 2 * kHSteps && kFinisher < kSmSlots, "slots cover the groups");
    __shared__ double red[kSmSlots][kSmEvals];
    __shared__ double red2[(MODE == XMODE_SCREEN) ? kSmSlots : 1][kSmEvals];
    __shared__ double red3[(MODE == XMODE_SCREEN) ? kSmSlots : 1][kSmEvals];
    __shared__ float red4[(MODE == XMODE_SCREEN) ? kSmSlots : 1][kSmEvals], red5[(MODE == XMODE_SCREEN) ? kSmSlots : 1][kSmEvals];
    __shared__ float s_win[kSmEvals * kWinPitch];
    __shared__ unsigned s_w0[kSmEvals];
    // (list mode: slot j holds evaluation idx_list[j] of the window [list_off, list_off + list_cap) of the list counted by list_counter)
    const int n_evals = idx_list ? window_count(counters[list_counter], list_off, list_cap) : counters[CNT_EVALS];
    // XMODE_I8: lists are of unknown length at launch; both feature kernels are launched and the list's length decides on the
    // device which of them works -- this one (a third of the serial chain per thread: latency) up to kI8SmallList entries, the
    // 64-evaluation workgroups of k_features (half the time per evaluation at 100 k entries: 2.9 against 5.5 ns) beyond
    if (MODE == XMODE_I8 && n_evals > kI8SmallList) return;
    const long n_pad = ((long)n_evals + kBlock - 1) / kBlock * kBlock;
    if ((long)blockIdx.x * kSmEvals >= n_pad) return;
    // (list mode is launched for the list's CAPACITY -- at C5 123 k workgroups for a list of a few hundred entries, 90 us of empty
    // workgroups -- so its grid is capped and the workgroups stride over the blocks of 16 evaluations)
    __shared__ double s_tab[MODE == XMODE_SCREEN ? 1 : hafq::kTabDoubles];
    __shared__ unsigned long long s_scr[MODE == XMODE_SCREEN ? hafq::kScrTabWords : 1];
    hafq::PtrTabs tb{};
    hafq::ScrTabs st{};
    if (MODE == XMODE_SCREEN) st = load_screen_tables(s_scr);
    else tb = load_decimal_tables(s_tab);
    const int ev = threadIdx.x & 15, slot = threadIdx.x >> 4;
    for (long blk = blockIdx.x; blk * kSmEvals < n_pad; blk += gridDim.x) {
    const long e = blk * kSmEvals + ev;
    const long tile = e >> 5;
    const int r = (int)(e & 31);
    float *xcol = X + (size_t)tile * kTileFloats + (e & 31);
    char *xtile = reinterpret_cast<char *>(X) + (size_t)tile * (MODE == XMODE_SCREEN ? kS0MatBytes : kHXTileBytes);
    const int n_groups = (MODE == XMODE_I8) ? 44 : (MODE == XMODE_F32 || MODE == XMODE_F64) ? (kKP + 7) / 8 : (MODE == XMODE_SCREEN) ? kS0Groups : 2 * kHSteps;   // 44 (i8_store zeroes 44..47 itself) / 41 / 40 / 42
    double *x64 = reinterpret_cast<double *>(X) + (size_t)(e >> 4) * kKP * 16 + (e & 15);      // XMODE_F64: see k_features
    const bool live = e < n_evals;
    const rsrc_t iir = make_ii_rsrc(ii, d);
    const int e_src = live ? (idx_list ? idx_list[e] : (int)e) : 0;   // the evaluation this slot holds
    if (slot == 0) s_w0[ev] = live ? window_origin(evalcell[e_src], d.H, d.W) : 0xffffffffu;
    __syncthreads();
    for (int idx = threadIdx.x; idx < kSmEvals * 15 * 16; idx += kSmWaves * 64) {
        const int col = idx & 15, seg = idx >> 4, wev = seg & (kSmEvals - 1), x = seg >> 4;       // 16 lanes = one window row
        const unsigned o = s_w0[wev];
        if (col < 15)
            s_win[wev * kWinPitch + x * 15 + col] =
                (o != 0xffffffffu) ? __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(iir, (int)(o + (unsigned)(x * (d.W + 1) + col) * 4u), 0, 0)) : 0.0f;
    }
    __syncthreads();
    const SrcWin src{s_win + ev * kWinPitch};
    double xx = 0.0;
    ScreenSums acc{0.0f, 0.0f, 0.0f, 0.0f};
    float sx = 0.0f;
    half8 hi = {0, 0, 0, 0, 0, 0, 0, 0}, lo = {0, 0, 0, 0, 0, 0, 0, 0};
    const int g = slot;
    const bool has_group = g < n_groups;
    long long xx_ll = 0;                                   // XMODE_I8: sum of the squared fixed-point attributes of this group (exact)
    int ovf = 0;
    unsigned long long dig[4] = {0, 0, 0, 0};             // XMODE_I8: the four digit planes of this thread's 8 attributes
    if (has_group) {
#pragma unroll
        for (int q = 0; q < 8; q++) {
            const int f = g * 8 + q;
            double xd = 0.0;
            if (MODE == XMODE_SCREEN) {
                const FeatDesc &F = fd[f];                               // screening form: fd = one descriptor per SLOT (an unused slot has skip = 1)
                if (live && !F.skip) xd = screen_attribute(src, F, st);
                const float ex = F.scr_extra;                            // (per quarter wave here: the group differs between them)
                if (ex != 0.0f) { const float ff = (float)xd; sx = fmaf(ex * ff, ff, sx); }
            } else if (live && f < d.nf) {
                xd = attribute_value_rec(src, fd[f], lower, upper, tb, (dbg) ? dbg + (size_t)e_src * kKP + f : nullptr);
            }
            const float xf = (float)xd;
            if (MODE == XMODE_SCREEN) {
                hi[q] = screen_operand(xd, acc, sp.corr[f]);           // (per quarter wave: a 16-byte vector load)
            } else if (MODE == XMODE_SPLIT) {
                const _Float16 h = (_Float16)xf;                       // RN
                const _Float16 l = (_Float16)(xf - (float)h);          // exact difference, then RN
                hi[q] = h;
                lo[q] = l;
                const float xe = (float)h + (float)l;                  // the value the three passes actually multiply
                xx = fma((double)xe, (double)xe, xx);
            } else if (MODE == XMODE_F64) {
                if (f < kKP) x64[(size_t)f * 16] = xd;
            } else if (MODE == XMODE_I8) {
                i8_digits(xd, q, dig, xx_ll, ovf);
            } else {
                if (f < kDP) xcol[f * kTile] = xf;                     // rows >= nf (padding up to the tile image) are zero
                xx = fma((double)xf, (double)xf, xx);
            }
        }
        if (MODE == XMODE_SPLIT) store_group_h(xtile, r, g, hi, lo);
        if (MODE == XMODE_SCREEN) store_group_img(xtile, r, g, hi);
        if (MODE == XMODE_I8) i8_store(X, e, g, dig);
    }
    if (MODE == XMODE_I8) { red_ll[slot][ev] = xx_ll; red_ovf[slot][ev] = ovf; }
    red[slot][ev] = (MODE == XMODE_SCREEN) ? (double)acc.su2 : xx;
    if (MODE == XMODE_SCREEN) { red2[slot][ev] = (double)acc.sd2; red3[slot][ev] = (double)sx; red4[slot][ev] = acc.cr; red5[slot][ev] = acc.ub; }
    __syncthreads();
    if (slot == kFinisher) {
        double t = 0.0, t2 = 0.0, t3 = 0.0, t4 = 0.0, t5 = 0.0;
#pragma unroll
        for (int k = 0; k < kSmSlots; k++) t += red[k][ev];            // fixed order: deterministic
        if (MODE == XMODE_SCREEN) {
#pragma unroll
            for (int k = 0; k < kSmSlots; k++) { t2 += red2[k][ev]; t3 += red3[k][ev]; t4 += (double)red4[k][ev]; t5 += (double)red5[k][ev]; }
            float band[kBandFloats] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f}, nax = 0.0f;
            if (live) screen_finish(t, t2, t + t3, t4, t5, sp, band, nax);
            store_band(ax + kBandFloats * e, band);
            ax2[e] = nax;
        } else if (MODE == XMODE_I8) {
            long long s2 = 0;
            int any = 0;
#pragma unroll
            for (int k2 = 0; k2 < kSmSlots; k2++) { s2 += red_ll[k2][ev]; any |= red_ovf[k2][ev]; }      // exact: < 324 * 2^54
            // |xq|^2 in real units (one rounding: 2^-53 relative); negative = an attribute beyond the fixed-point range
            reinterpret_cast<double *>(ax)[e] = any ? -1.0 : ldexp((double)s2, -2 * kI8Q);
        } else if (MODE != XMODE_F64) {
            ax[e] = neg_gamma2 * (float)t;                             // -gamma*log2(e)*|x|^2, folded into the exp2 argument
        }
    }
    __syncthreads();                                                   // the next block reuses s_w0, s_win and the reduction arrays
    }
}

// Tiny requests (the whole SVM work of the request is a few hundred thousand evaluation x support-vector pairs: C2 with a model of
// a few hundred support vectors), ONE launch for a5-a8: a workgroup takes 16 evaluations exactly as k_features_small does -- a
// quarter wave per group of 8 attributes, both text round trips in exact arithmetic -- but leaves the fp64 attributes in LDS, and
// its eleven waves then share the SV tiles of the fp64 MFMA contraction between them (v_mfma_f64_16x16x4_f64, the B operand straight
// from the model image in L2: a tile is used once per workgroup).  Same arithmetic as tier 2 (k_recheck_mfma / k_recheck_combine)
// except for the order in which the partial sums of the SV tiles are added -- eleven waves instead of eight ranges, fixed -- and
// the same hand-over to the strict tier for |dec| <= guard2 * T * S.  Replaces three launches and the 10 MB round trip of the
// attribute image for such a request.
constexpr int kSdMSteps = kKP / 4;                // 81 k-steps of 4
__global__ __launch_bounds__(kSmWaves * 64) void k_small_direct(const float *__restrict__ ii, const int *__restrict__ evalcell,
                                                                const FeatDesc *__restrict__ fd, const double *__restrict__ sv64,
                                                                ExactParams p, Dims d, double *__restrict__ dec_exact,
                                                                int8_t *__restrict__ labels, int *__restrict__ flag2_list, int flag2_cap,
                                                                int *counters, AttrRecord *__restrict__ dbg,
                                                                const int *__restrict__ idx_list, int list_counter, int list_cap, int list_off)
{
    // (list mode, idx_list != nullptr: the window [list_off, list_off + list_cap) of a tier's list instead of every evaluation --
    // the exact stage of a SMALL request behind the three-pass kernel in one launch; slot j holds evaluation idx_list[j], its
    // decision value goes to dec_exact[j]; idx_list and dec_exact already point at entry list_off)
    __shared__ double s_x[kKP * 16];                  // [attribute][evaluation]: the A operand of the fp64 MFMA, k-major
    __shared__ double s_part[kSmWaves][16][2];        // per wave: sum coef*K and sum |coef|*K of its SV tiles, per evaluation
    __shared__ double s_xx[16];
    __shared__ float s_win[kSmEvals * kWinPitch];
    __shared__ unsigned s_w0[kSmEvals];
    __shared__ double s_tab[hafq::kTabDoubles];
    const int n_evals = idx_list ? window_count(counters[list_counter], list_off, list_cap) : counters[CNT_EVALS];
    if ((long)blockIdx.x * kSmEvals >= n_evals) return;
    const hafq::PtrTabs tb = load_decimal_tables(s_tab);
    const int ev = threadIdx.x & 15, slot = threadIdx.x >> 4, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const long e = (long)blockIdx.x * kSmEvals + ev;
    const bool live = e < n_evals;
    const int e_src = live ? (idx_list ? idx_list[e] : (int)e) : 0;    // the evaluation this slot holds
    const rsrc_t iir = make_ii_rsrc(ii, d);
    if (slot == 0) s_w0[ev] = live ? window_origin(evalcell[e_src], d.H, d.W) : 0xffffffffu;
    __syncthreads();
    for (int idx = threadIdx.x; idx < kSmEvals * 15 * 16; idx += kSmWaves * 64) {
        const int col = idx & 15, seg = idx >> 4, wev = seg & (kSmEvals - 1), x = seg >> 4;       // 16 lanes = one window row
        const unsigned o = s_w0[wev];
        if (col < 15)
            s_win[wev * kWinPitch + x * 15 + col] =
                (o != 0xffffffffu) ? __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(iir, (int)(o + (unsigned)(x * (d.W + 1) + col) * 4u), 0, 0)) : 0.0f;
    }
    __syncthreads();
    const SrcWin src{s_win + ev * kWinPitch};
    const int g = slot;
    if (g < (kKP + 7) / 8) {
#pragma unroll
        for (int q = 0; q < 8; q++) {
            const int f = g * 8 + q;
            double xd = 0.0;
            if (live && f < d.nf) xd = attribute_value_rec(src, fd[f], p.lower, p.upper, tb, dbg ? dbg + (size_t)e_src * kKP + f : nullptr);
            if (f < kKP) s_x[f * 16 + ev] = xd;
        }
    }
    __syncthreads();
    if (threadIdx.x < 16) {                           // |x|^2, attributes in index order (as k_recheck_mfma's per-lane sums would not be: fixed here)
        double xx = 0.0;
        for (int k = 0; k < kKP; k++) xx = fma(s_x[k * 16 + threadIdx.x], s_x[k * 16 + threadIdx.x], xx);
        s_xx[threadIdx.x] = xx;
    }
    __syncthreads();
    // ---- fp64 MFMA over this wave's SV tiles: A[row = lane&15][k = 4s + (lane>>4)] from LDS, B[k][col = lane&15] from the model ----
    const int n_tiles = p.n_sv_pad / 16;
    double part[4] = {0, 0, 0, 0}, pabs[4] = {0, 0, 0, 0};
    for (int t = wave; t < n_tiles; t += kSmWaves) {
        const double *Bg = sv64 + (size_t)t * 16 + (lane & 15);
        f64x4 acc = {0, 0, 0, 0};
#pragma unroll 9
        for (int s = 0; s < kSdMSteps; s++) {
            const int k = 4 * s + (lane >> 4);
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(s_x[k * 16 + (lane & 15)], Bg[(size_t)k * p.n_sv_pad], acc, 0, 0, 0);
        }
        const double ss = Bg[(size_t)kKP * p.n_sv_pad];
        const double cf = Bg[(size_t)(kKP + 1) * p.n_sv_pad];
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const int row = (lane >> 4) + 4 * r;                               // f64 C/D layout: row = (lane>>4) + 4*reg
            const double d2 = fma(-2.0, acc[r], s_xx[row] + ss);
            const double kv = exp(-p.gamma * d2);
            part[r] = fma(cf, kv, part[r]);
            pabs[r] = fma(fabs(cf), kv, pabs[r]);
        }
    }
#pragma unroll
    for (int r = 0; r < 4; r++) {
        double v = part[r], w = pabs[r];
        v += __shfl_xor(v, 8, 64); w += __shfl_xor(w, 8, 64);
        v += __shfl_xor(v, 4, 64); w += __shfl_xor(w, 4, 64);
        v += __shfl_xor(v, 2, 64); w += __shfl_xor(w, 2, 64);
        v += __shfl_xor(v, 1, 64); w += __shfl_xor(w, 1, 64);
        if ((lane & 15) == 0) { s_part[wave][(lane >> 4) + 4 * r][0] = v; s_part[wave][(lane >> 4) + 4 * r][1] = w; }
    }
    __syncthreads();
    if (threadIdx.x < 16 && live) {
        double P = 0.0, S = 0.0;
#pragma unroll
        for (int w = 0; w < kSmWaves; w++) { P += s_part[w][threadIdx.x][0]; S += s_part[w][threadIdx.x][1]; }   // fixed order
        const double dv = P - p.rho;
        dec_exact[e] = dv;
        labels[evalcell[e_src]] = (int8_t)(dv > 0.0 ? p.gv0 : p.gv1);
        const double T = p.as_max1 + p.gamma2 * s_xx[threadIdx.x];
        if (!(fabs(dv) > p.guard2 * T * S)) {
            const int s2 = atomicAdd(&counters[CNT_FLAGGED2], 1);
            if (s2 < flag2_cap) flag2_list[s2] = e_src;
        }
    }
}

void launch_small_direct(const float *ii, const int *evalcell, int *counters, const FeatDesc *fd, const double *sv64, ExactParams p, Dims d,
                         long max_evals, double *dec_exact, int8_t *labels, int *flag2_list, int flag2_cap, AttrRecord *dbg, hipStream_t s,
                         const int *idx_list, int list_counter, int list_off)
{
    const long nb = (max_evals + kSmEvals - 1) / kSmEvals;
    if (nb <= 0) return;
    if (idx_list) { idx_list += list_off; dec_exact += list_off; }
    hipLaunchKernelGGL(k_small_direct, dim3((unsigned)nb), dim3(kSmWaves * 64), 0, s, ii, evalcell, fd, sv64, p, d, dec_exact, labels,
                       flag2_list, flag2_cap, counters, dbg, idx_list, list_counter, (int)max_evals, list_off);
}

template <int MODE>
static void launch_features_mode(const float *ii, const int *evalcell, const int *counters, const FeatDesc *fd, float *X, float *ax,
                                 Dims d, double lower, double upper, float neg_gamma2, long max_evals, ScreenParams sp,
                                 const int *idx_list, int list_counter, int list_cap, bool large, long sel_evals,
                                 AttrRecord *dbg, float *ax2, hipStream_t s, int list_off = 0)
{
    constexpr int kBlock = (MODE == XMODE_SCREEN) ? kS0BlockEvals : (MODE == XMODE_F64 || MODE == XMODE_I8) ? 64 : kSvmBlockEvals;
    if (MODE == XMODE_I8) {                                   // list mode only; the list's length decides on the device which kernel works
        const long cap_small = std::min<long>(max_evals, kI8SmallList);
        const long nb = ((cap_small + kBlock - 1) / kBlock) * (kBlock / kSmEvals);
        hipLaunchKernelGGL(k_features_small<MODE>, dim3((unsigned)nb), dim3(kSmWaves * 64), 0, s, ii, evalcell, counters, fd, X, ax, d,
                           lower, upper, neg_gamma2, sp, dbg, ax2, idx_list, list_counter, list_cap, list_off);
        if (max_evals > kI8SmallList) {
            long blocks = ((max_evals + kBlock - 1) / kBlock) * (kBlock / kFeatEvals);
            if (blocks > 4096) blocks = 4096;                 // grid-stride inside the kernel
            hipLaunchKernelGGL((k_features<MODE, 16>), dim3((unsigned)blocks), dim3(16 * 64), 0, s, ii, evalcell, counters, fd, X, ax, d,
                               lower, upper, neg_gamma2, sp, idx_list, list_counter, list_cap, dbg, ax2, list_off);
        }
        return;
    }
    if (large && MODE != XMODE_F64) {
        // enough evaluations to fill the chip with one thread each
        long blocks = (max_evals + kBlock - 1) / kBlock * (kBlock / 256);
        hipLaunchKernelGGL(k_features_serial<MODE>, dim3((unsigned)blocks), dim3(256), 0, s, ii, evalcell, counters, fd, X, ax, d,
                           lower, upper, neg_gamma2, sp, idx_list, list_counter, list_cap, dbg, ax2);
        return;
    }
    // small requests -- and every list of the fp64 tier, which is short unless the model is ill-conditioned: a third of the
    // serial chain per thread (C3's ~8 000 flagged evaluations: 27 us against 56 us with the 64-evaluation workgroups)
    if ((!idx_list && sel_evals <= kSmallEvals) || (idx_list && MODE == XMODE_F64)) {
        long nb = ((max_evals + kBlock - 1) / kBlock) * (kBlock / kSmEvals);
        if (idx_list && nb > 2048) nb = 2048;                          // grid-stride inside the kernel
        hipLaunchKernelGGL(k_features_small<MODE>, dim3((unsigned)nb), dim3(kSmWaves * 64), 0, s, ii, evalcell, counters, fd, X, ax, d,
                           lower, upper, neg_gamma2, sp, dbg, ax2, idx_list, list_counter, list_cap, list_off);
        return;
    }
    long blocks = ((max_evals + kBlock - 1) / kBlock) * (kBlock / kFeatEvals);
    if (idx_list && blocks > 4096) blocks = 4096;                      // grid-stride inside the kernel
    if (idx_list || sel_evals <= 24576)
        hipLaunchKernelGGL((k_features<MODE, 16>), dim3((unsigned)blocks), dim3(16 * 64), 0, s, ii, evalcell, counters, fd, X, ax, d,
                           lower, upper, neg_gamma2, sp, idx_list, list_counter, list_cap, dbg, ax2, list_off);
    else
        hipLaunchKernelGGL((k_features<MODE, 8>), dim3((unsigned)blocks), dim3(8 * 64), 0, s, ii, evalcell, counters, fd, X, ax, d,
                           lower, upper, neg_gamma2, sp, idx_list, list_counter, list_cap, dbg, ax2, list_off);
}

void launch_features(const float *ii, const int *evalcell, const int *counters, const FeatDesc *fd, float *X, float *ax,
                     Dims d, double lower, double upper, float neg_gamma2, long max_evals, int xmode, ScreenParams sp,
                     const int *idx_list, int list_counter, int list_cap, bool large, long sel_evals, AttrRecord *dbg,
                     float *ax2, hipStream_t s, int list_off)
{
    if (max_evals <= 0) return;
    if (xmode == XMODE_I8) {
        launch_features_mode<XMODE_I8>(ii, evalcell, counters, fd, X, ax, d, lower, upper, neg_gamma2, max_evals, sp, idx_list,
                                       list_counter, list_cap, false, sel_evals, dbg, ax2, s, list_off);
        return;
    }
    if (xmode == XMODE_F64) {
        launch_features_mode<XMODE_F64>(ii, evalcell, counters, fd, X, ax, d, lower, upper, neg_gamma2, max_evals, sp, idx_list,
                                        list_counter, list_cap, false, sel_evals, dbg, ax2, s, list_off);
        return;
    }
    // (screening form: the kernels index their descriptor argument by SLOT; through the __restrict__ kernel argument the
    // wave-uniform descriptor words arrive by scalar loads -- through the pointer inside ScreenParams they would not)
    if (xmode == XMODE_SCREEN)
        launch_features_mode<XMODE_SCREEN>(ii, evalcell, counters, sp.fd_slot, X, ax, d, lower, upper, neg_gamma2, max_evals, sp, idx_list,
                                           list_counter, list_cap, large, sel_evals, dbg, ax2, s);
    else if (xmode == XMODE_SPLIT)
        launch_features_mode<XMODE_SPLIT>(ii, evalcell, counters, fd, X, ax, d, lower, upper, neg_gamma2, max_evals, sp, idx_list,
                                          list_counter, list_cap, large, sel_evals, dbg, ax2, s);
    else
        launch_features_mode<XMODE_F32>(ii, evalcell, counters, fd, X, ax, d, lower, upper, neg_gamma2, max_evals, sp, idx_list,
                                        list_counter, list_cap, large, sel_evals, dbg, ax2, s);
}

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
                                                            int *__restrict__ counters_rw, Dims d)
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
                if (!(fabsf(dv) > (p.guard_acc + p.guard_dot * (p.as_max + fabsf(axr[r]))) * pabs[r] + p.guard_abs)) {
                    int slot = atomicAdd(&counters_rw[CNT_FLAGGED], 1);
                    if (slot < flag_cap) flag_list[slot] = (int)e;
                }
            }
        }
    }
}

void launch_svm(const float *X, const float *ax, const float *svt, const int *evalcell, const int *counters, SvmParams p,
                float *dec, int8_t *labels, int *flag_list, int flag_cap, int *counters_rw, Dims d, long max_evals,
                hipStream_t s)
{
    long blocks = (max_evals + kSvmBlockEvals - 1) / kSvmBlockEvals;
    if (blocks <= 0) return;
    hipLaunchKernelGGL(k_svm_rbf, dim3((unsigned)blocks), dim3(kSvmThreads), 0, s, X, ax, svt, evalcell, counters, p,
                       dec, labels, flag_list, flag_cap, counters_rw, d);
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
template <bool PRECISE, bool CRP = false>
__global__ __launch_bounds__(kSvmThreads, 2) void k_svm_rbf_h(const char *__restrict__ X, const float *__restrict__ ax,
                                                              const char *__restrict__ svt,
                                                              const int *__restrict__ evalcell,
                                                              const int *__restrict__ counters, SvmParams p,
                                                              float *__restrict__ dec, int8_t *__restrict__ labels,
                                                              int *__restrict__ flag_list, int flag_cap,
                                                              int *__restrict__ counters_rw, Dims d,
                                                              const int *__restrict__ idx_list, int list_counter, int list_cap,
                                                              double *__restrict__ part_out, long part_stride)
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
    const int t0 = part_out ? (int)((long)d.n_sv_tiles * blockIdx.y / gridDim.y) : 0;
    const int nt = part_out ? (int)((long)d.n_sv_tiles * (blockIdx.y + 1) / gridDim.y) : d.n_sv_tiles;
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
                    if (!(fabsf(dv) > (gacc + gdot * (p.as_max + fabsf(axs[row]))) * sabs + p.guard_abs)) {
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
                                                       int list_counter, int list_cap)
{
    const int n_evals = min(counters[list_counter], list_cap);
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
        if (!(fabsf(dv) > (p.guard_acc_l + p.guard_dot_p * (p.as_max + fabsf(ax[es]))) * sabs + p.guard_abs)) {
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
                                                          const int *__restrict__ idx_list, int list_counter, int list_cap)
{
    const int n_evals = min(counters[list_counter], list_cap);
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
        if (!(fabs(dvd) > err) || !(D < 0.05)) {
            int slot = atomicAdd(&counters_rw[CNT_FLAGGED], 1);
            if (slot < flag_cap) flag_list[slot] = e;
        }
    }
}

void launch_svm_h(const void *Xh, const float *ax, const void *svt_h, const int *evalcell, const int *counters, SvmParams p,
                  float *dec, int8_t *labels, int *flag_list, int flag_cap, int *counters_rw, Dims d, long max_evals,
                  const int *idx_list, int list_counter, int list_cap, double *part_out, long part_stride, hipStream_t s,
                  const CrT1Params *cr, const double *Lbuf)
{
    long blocks = (max_evals + kSvmBlockEvals - 1) / kSvmBlockEvals;
    if (blocks <= 0) return;
    if (idx_list && cr && part_out) {
        const int parts = (d.n_sv_tiles >= 4 * kHListParts) ? kHListParts : 1;
        hipLaunchKernelGGL((k_svm_rbf_h<true, true>), dim3((unsigned)blocks, (unsigned)parts), dim3(kSvmThreads), 0, s, (const char *)Xh, ax,
                           (const char *)svt_h, evalcell, counters, p, dec, labels, flag_list, flag_cap, counters_rw, d, idx_list,
                           list_counter, list_cap, part_out, part_stride);
        hipLaunchKernelGGL(k_svm_h_combine_cr, dim3(1024), dim3(256), 0, s, part_out, part_stride, parts, ax, Lbuf, evalcell, counters, *cr, dec,
                           labels, flag_list, flag_cap, counters_rw, idx_list, list_counter, list_cap);
        return;
    }
    if (idx_list) {
        const int parts = (part_out && d.n_sv_tiles >= 4 * kHListParts) ? kHListParts : 1;     // engine.cpp: guard_acc_l follows this rule
        double *po = parts > 1 ? part_out : nullptr;
        hipLaunchKernelGGL(k_svm_rbf_h<true>, dim3((unsigned)blocks, (unsigned)parts), dim3(kSvmThreads), 0, s, (const char *)Xh, ax,
                           (const char *)svt_h, evalcell, counters, p, dec, labels, flag_list, flag_cap, counters_rw, d, idx_list,
                           list_counter, list_cap, po, part_stride);
        if (po)
            hipLaunchKernelGGL(k_svm_h_combine, dim3(1024), dim3(256), 0, s, po, part_stride, parts, ax, evalcell, counters, p, dec,
                               labels, flag_list, flag_cap, counters_rw, idx_list, list_counter, list_cap);
    } else {
        hipLaunchKernelGGL(k_svm_rbf_h<false>, dim3((unsigned)blocks), dim3(kSvmThreads), 0, s, (const char *)Xh, ax, (const char *)svt_h,
                           evalcell, counters, p, dec, labels, flag_list, flag_cap, counters_rw, d, idx_list, list_counter, list_cap,
                           (double *)nullptr, 0L);
    }
}

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
                for (int k = 0; k < p.kx; k++) {
                    const double s = col[(size_t)k * p.n_sv_pad];
#pragma unroll
                    for (int ev = 0; ev < kRB; ev++) {
                        double dd = __dsub_rn(xs[ev][k], s);
                        sum[ev] = __dadd_rn(sum[ev], __dmul_rn(dd, dd));          // svm.cpp:333-334, 342, 347
                    }
                }
                const double c = coef64[n];
#pragma unroll
                for (int ev = 0; ev < kRB; ev++) terms[ev][tid] = __dmul_rn(c, exp(__dmul_rn(-p.gamma, sum[ev])));
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
    splits = fine ? 4 * kMSplit : kMSplit;
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
                                                         int cslot)
{
    const int n_flag = window_count(counters[cslot], list_off, flag_cap);
    int splits;
    size_t pitch;
    recheck_split(n_flag, flag_cap, p.n_sv_pad / 16, splits, pitch);
    for (int sl = blockIdx.x * 256 + threadIdx.x; sl < n_flag; sl += gridDim.x * 256) {
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
        if (!(fabs(dv) > p.guard2 * T * S)) {
            int s2 = atomicAdd(&counters[CNT_FLAGGED2], 1);
            if (s2 < flag2_cap) flag2_list[s2] = e;
        }
    }
}

// flag_list and dec_exact are the WHOLE lists (one entry per flagged evaluation, sized for every evaluation of a request);
// the launch works on the window [list_off, list_off + window_cap) of them, which is what x64 / part64 are sized for.  The
// host runs window 0 with every request and further windows only when more evaluations were flagged than one window holds.
void launch_recheck_mfma(const float *ii, const int *evalcell, const FeatDesc *fd, const double *sv64, ExactParams p,
                         const int *flag_list, int window_cap, int list_off, int *counters, double *x64, double *part64,
                         double *dec_exact, int8_t *labels, int *flag2_list, int flag2_cap, Dims d, hipStream_t s, AttrRecord *dbg,
                         bool have_x64, int counter_slot)
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
                       dec_exact, labels, flag2_list, flag2_cap, counter_slot);
}

// ---------------------------------------------------------------------------------------------------
// a10: 29-tap weighted vote, first-wins argmax, longest-run centring; plus the 9x8 z window of a11.
// One workgroup per (cloud, roll).
// ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ int vote_at(const int8_t *__restrict__ g, int W, int row, int col)
{
#define G(dr, dc) ((int)g[(row + (dr)) * W + (col + (dc))])
    return 1 * G(-2, -2) + 2 * G(-2, -1) + 3 * G(-2, 0) + 2 * G(-2, 1) + 1 * G(-2, 2) +
           2 * G(-1, -2) + 3 * G(-1, -1) + 4 * G(-1, 0) + 3 * G(-1, 1) + 2 * G(-1, 2) +
           2 * G(0, -4) + 2 * G(0, -3) + 3 * G(0, -2) + 4 * G(0, -1) + 55 * G(0, 0) + 4 * G(0, 1) + 3 * G(0, 2) +
           2 * G(0, 3) + 2 * G(0, 4) +
           2 * G(1, -2) + 3 * G(1, -1) + 4 * G(1, 0) + 3 * G(1, 1) + 2 * G(1, 2) +
           1 * G(2, -2) + 2 * G(2, -1) + 3 * G(2, 0) + 2 * G(2, 1) + 1 * G(2, 2);       // 873-878
#undef G
}

// pass 1: vote grid + per-roll maximum.  Many workgroups per (cloud, roll); the roll's top is an atomicMax on a 64-bit
// key (vote, then smallest linear index): max is order independent, so the result is deterministic.
constexpr int kVoteCellsPerBlock = 2048;

// W % 4 == 0 (every grid the engine is normally used with): a thread computes FOUR horizontally adjacent cells from fifteen
// aligned 4-byte loads (5 rows x 12 labels) instead of 4 x 29 single-byte loads -- the kernel was bound by the texture
// addresser, not by arithmetic.  Same integer sum, same first-wins key.
__device__ __forceinline__ void vote_quad(const int8_t *__restrict__ g, int W, int row, int c, int (&v)[4])
{
    int b[5][12];
#pragma unroll
    for (int dr = 0; dr < 5; dr++) {
        const int *p = reinterpret_cast<const int *>(g + (size_t)(row + dr - 2) * W + c - 4);
#pragma unroll
        for (int w = 0; w < 3; w++) {
            const int x = p[w];
#pragma unroll
            for (int k = 0; k < 4; k++) b[dr][4 * w + k] = (int)(int8_t)(x >> (8 * k));
        }
    }
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const int j = 4 + k;
        v[k] = 1 * b[0][j - 2] + 2 * b[0][j - 1] + 3 * b[0][j] + 2 * b[0][j + 1] + 1 * b[0][j + 2] +
               2 * b[1][j - 2] + 3 * b[1][j - 1] + 4 * b[1][j] + 3 * b[1][j + 1] + 2 * b[1][j + 2] +
               2 * b[2][j - 4] + 2 * b[2][j - 3] + 3 * b[2][j - 2] + 4 * b[2][j - 1] + 55 * b[2][j] + 4 * b[2][j + 1] + 3 * b[2][j + 2] +
               2 * b[2][j + 3] + 2 * b[2][j + 4] +
               2 * b[3][j - 2] + 3 * b[3][j - 1] + 4 * b[3][j] + 3 * b[3][j + 1] + 2 * b[3][j + 2] +
               1 * b[4][j - 2] + 2 * b[4][j - 1] + 3 * b[4][j] + 2 * b[4][j + 1] + 1 * b[4][j + 2];       // 873-878
        if (b[2][j] < 0) v[k] = 0;                                   // 870-871: a cell without a positive label scores 0
    }
}

__global__ __launch_bounds__(256) void k_vote_cells(const int8_t *__restrict__ labels, short *__restrict__ ev16,
                                                    unsigned long long *__restrict__ topkey, int *__restrict__ rowmax, Dims d)
{
    __shared__ unsigned long long red[256];
    __shared__ int rmax[kVoteCellsPerBlock / 8 + 2];                  // best vote of every grid row this block touches (W >= 15)
    const int br = blockIdx.y, t = threadIdx.x;
    const int H = d.H, W = d.W, HW = H * W;
    const int8_t *g = labels + (size_t)br * HW;
    short *ev = ev16 + (size_t)br * HW;
    unsigned long long best = 0;
    const int lo = blockIdx.x * kVoteCellsPerBlock, hi = min(HW, lo + kVoteCellsPerBlock);
    const int row_lo = lo / W, n_rows = (hi - 1) / W - row_lo + 1;
    for (int k = t; k < n_rows; k += 256) rmax[k] = 0;
    __syncthreads();
    if ((W & 3) == 0) {
        for (int idx = lo + 4 * t; idx < hi; idx += 4 * 256) {       // (kVoteCellsPerBlock and W are multiples of 4: a quad never straddles)
            const int row = idx / W, col = idx - row * W;
            int v[4] = {0, 0, 0, 0};
            if (row >= 2 && row < H - 2 && col >= 4 && col + 4 <= W - 4) vote_quad(g, W, row, col, v);   // 870-879 (the border scores 0)
            typedef short short4v __attribute__((ext_vector_type(4)));
            *reinterpret_cast<short4v *>(ev + idx) = short4v{(short)v[0], (short)v[1], (short)v[2], (short)v[3]};
            const int vm = max(max(v[0], v[1]), max(v[2], v[3]));
            if (vm > 0) atomicMax(&rmax[row - row_lo], vm);
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const unsigned long long key = ((unsigned long long)(unsigned)(v[k] + 32768) << 32) | (unsigned)(0x7FFFFFFF - (idx + k));
                if (key > best) best = key;                           // larger vote, then smaller index (first wins, 882)
            }
        }
    } else {
        for (int idx = lo + t; idx < hi; idx += 256) {
            int row = idx / W, col = idx - row * W;
            int v = 0;
            if (g[idx] >= 0 && row >= 2 && row < H - 2 && col >= 4 && col < W - 4) v = vote_at(g, W, row, col);   // 870-879
            ev[idx] = (short)v;
            if (v > 0) atomicMax(&rmax[row - row_lo], v);
            unsigned long long key = ((unsigned long long)(unsigned)(v + 32768) << 32) | (unsigned)(0x7FFFFFFF - idx);
            if (key > best) best = key;                               // larger vote, then smaller index (first wins, 882)
        }
    }
    red[t] = best;
    __syncthreads();
    // per-row maxima for k_vote_pick (a row may be shared with the neighbouring blocks: atomicMax, votes are >= 0)
    for (int k = t; k < n_rows; k += 256)
        if (rmax[k] > 0) atomicMax(&rowmax[(size_t)br * H + row_lo + k], rmax[k]);
    for (int o = 128; o > 0; o >>= 1) {
        if (t < o && red[t + o] > red[t]) red[t] = red[t + o];
        __syncthreads();
    }
    if (t == 0 && red[0]) atomicMax(&topkey[br], red[0]);
}

// pass 2: longest-run centring on the roll's top value (904-932).  One wave per 64 grid rows, a thread per row: the row goes by
// in 16-byte pieces (8 votes), and only a piece that holds the top value is looked at vote by vote -- hardly any does.  The
// best (longest run, then smallest row) of the roll is a 64-bit atomicMax; k_vote_record turns it into the roll record.
__global__ __launch_bounds__(64) void k_vote_pick(const short *__restrict__ ev16, unsigned long long *__restrict__ keys3,
                                                  const int *__restrict__ rowmax, Dims d)
{
    const int br = blockIdx.y, lane = threadIdx.x;
    const int H = d.H, W = d.W, HW = H * W, BR = d.B * d.R;
    const unsigned long long *topkey = keys3;
    unsigned long long *runkey = keys3 + BR;
    const short *ev = ev16 + (size_t)br * HW;
    const int top = (int)(topkey[br] >> 32) - 32768;
    const int row = blockIdx.x * 64 + lane;
    unsigned long long rbest = 0;
    // only a row whose best vote IS the roll's top can hold a run of it (k_vote_cells left the row maxima; top = 0 means every
    // row qualifies: the run of zeros of row 0 wins then, and the scan below finds it)
    if (row < H && rowmax[(size_t)br * H + row] == top) {
        int cur = 0, longest = 0, endc = 0;
        const short *er = ev + (size_t)row * W;
        auto step = [&](int v, int col) {
            if (v == top) {
                cur++;
                if (cur > longest) { longest = cur; endc = col; }
            } else cur = 0;
        };
        if ((W & 7) == 0) {
            typedef short short8 __attribute__((ext_vector_type(8)));
            for (int c0 = 0; c0 < W; c0 += 8) {
                const short8 v = *reinterpret_cast<const short8 *>(er + c0);
                bool any = false;
#pragma unroll
                for (int k = 0; k < 8; k++) any |= (v[k] == top);
                if (!any) { cur = 0; continue; }
#pragma unroll
                for (int k = 0; k < 8; k++) step(v[k], c0 + k);
            }
        } else {
            for (int col = 0; col < W; col++) step(er[col], col);
        }
        if (longest > 0) {
            const int bc = endc - longest / 2;            // first longest run wins, column = run end - len/2 (926-932)
            rbest = ((unsigned long long)(unsigned)longest << 40) | ((unsigned long long)(unsigned)(0xFFFF - row) << 20) | (unsigned)bc;
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const unsigned long long other = __shfl_xor(rbest, o, 64);
        if (other > rbest) rbest = other;                 // longer run, then smaller row
    }
    if (lane == 0 && rbest) atomicMax(&runkey[br], rbest);
}

// pass 3: the roll record from the longest-run key: run centre, z window of a11 (1342-1351), evaluation count.  Its own tiny
// launch: finishing the record inside k_vote_pick behind a completion counter needs a device-scope fence per workgroup, and
// on this chip each of those is an L2 write-back -- the 288 fences of a C5 request cost more than this launch.
__global__ __launch_bounds__(64) void k_vote_record(const float *__restrict__ heights, const int *__restrict__ brcount,
                                                    const unsigned long long *__restrict__ keys3, RollRecordDev *__restrict__ rec, Dims d)
{
    const int br = blockIdx.x, lane = threadIdx.x;
    const int H = d.H, W = d.W, HW = H * W, BR = d.B * d.R;
    const int top = (int)(keys3[br] >> 32) - 32768;
    const unsigned long long best = keys3[BR + br];
    const int brow = 0xFFFF - (int)((best >> 20) & 0xFFFFF), bcol = (int)(best & 0xFFFFF);
    // z estimate window rows brow-4..brow+4, cols bcol-4..bcol+3 (1342-1351), as an ordered-key max
    int zk = f2key(-10.0f);
    for (int t = lane; t < 72; t += 64) {
        const int rr = brow + (t / 8) - 4, cc = bcol + (t % 8) - 4;
        if (rr >= 0 && cc >= 0 && rr < H && cc < W) {
            const float h = heights[(size_t)br * HW + rr * W + cc];
            if (-10.0f < h) zk = max(zk, f2key(h));
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) zk = max(zk, __shfl_xor(zk, o, 64));
    if (lane == 0) {
        RollRecordDev r;
        r.vote = top; r.row = (short)brow; r.col = (short)bcol;
        r.h_locmax = key2f(zk);
        r.n_evals = brcount[br];
        rec[br] = r;
    }
}

// Small grids: labels, votes, argmax, run centring, z window and the record of one (cloud, roll) in ONE workgroup and one launch
// (five launches and two memsets otherwise: more than the work at 56 x 56).
constexpr int kVoteSmallCells = 16384;
__global__ __launch_bounds__(256) void k_vote_small(const int8_t *__restrict__ labels, const float *__restrict__ heights,
                                                    const int *__restrict__ brcount, short *__restrict__ ev16,
                                                    RollRecordDev *__restrict__ rec, Dims d)
{
    extern __shared__ short s_ev[];                       // [H*W] votes, then [H*W] labels as bytes
    __shared__ unsigned long long red[256];
    __shared__ int s_top, s_row, s_col;
    const int br = blockIdx.x, t = threadIdx.x;
    const int H = d.H, W = d.W, HW = H * W;
    int8_t *s_g = reinterpret_cast<int8_t *>(s_ev + HW);
    const int8_t *g = labels + (size_t)br * HW;
    short *ev = ev16 + (size_t)br * HW;
    for (int k = t; k < HW; k += 256) s_g[k] = g[k];
    __syncthreads();
    unsigned long long best = 0;
    for (int idx = t; idx < HW; idx += 256) {
        const int row = idx / W, col = idx - row * W;
        int v = 0;
        if (s_g[idx] >= 0 && row >= 2 && row < H - 2 && col >= 4 && col < W - 4) v = vote_at(s_g, W, row, col);   // 870-879
        s_ev[idx] = (short)v;
        ev[idx] = (short)v;
        const unsigned long long key = ((unsigned long long)(unsigned)(v + 32768) << 32) | (unsigned)(0x7FFFFFFF - idx);
        if (key > best) best = key;                       // larger vote, then smaller index (first wins, 882)
    }
    red[t] = best;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (t < o && red[t + o] > red[t]) red[t] = red[t + o];
        __syncthreads();
    }
    if (t == 0) s_top = (int)(red[0] >> 32) - 32768;
    __syncthreads();
    const int top = s_top;
    // longest horizontal run of `top` per row (904-932): first longest run wins, column = run end - len/2
    unsigned long long rbest = 0;
    for (int row = t; row < H; row += 256) {
        int cur = 0, longest = 0, endc = 0;
        for (int col = 0; col < W; col++) {
            if (s_ev[row * W + col] == top) {
                cur++;
                if (cur > longest) { longest = cur; endc = col; }
            } else cur = 0;
        }
        if (longest > 0) {
            const int bc = endc - longest / 2;
            const unsigned long long key = ((unsigned long long)(unsigned)longest << 40) | ((unsigned long long)(unsigned)(0xFFFF - row) << 20) | (unsigned)bc;
            if (key > rbest) rbest = key;                 // longer run, then smaller row
        }
    }
    __syncthreads();
    red[t] = rbest;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (t < o && red[t + o] > red[t]) red[t] = red[t + o];
        __syncthreads();
    }
    if (t == 0) { s_row = 0xFFFF - (int)((red[0] >> 20) & 0xFFFFF); s_col = (int)(red[0] & 0xFFFFF); }
    __syncthreads();
    const int brow = s_row, bcol = s_col;
    // z estimate window rows brow-4..brow+4, cols bcol-4..bcol+3 (1342-1351), as an ordered-key max
    if (t < 64) {
        int zk = f2key(-10.0f);
        for (int q = t; q < 72; q += 64) {
            const int rr = brow + (q / 8) - 4, cc = bcol + (q % 8) - 4;
            if (rr >= 0 && cc >= 0 && rr < H && cc < W) {
                const float h = heights[(size_t)br * HW + rr * W + cc];
                if (-10.0f < h) zk = max(zk, f2key(h));
            }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) zk = max(zk, __shfl_xor(zk, o, 64));
        if (t == 0) {
            RollRecordDev r;
            r.vote = top; r.row = (short)brow; r.col = (short)bcol;
            r.h_locmax = key2f(zk);
            r.n_evals = brcount[br];
            rec[br] = r;
        }
    }
}

void launch_vote(const int8_t *labels, const float *heights, const int *brcount, short *ev16, unsigned long long *topkey,
                 int *rowmax, RollRecordDev *rec, Dims d, hipStream_t s)
{
    if (d.H * d.W <= kVoteSmallCells) {                   // 3 bytes of LDS per cell: 48 KiB at most
        hipLaunchKernelGGL(k_vote_small, dim3(d.B * d.R), dim3(256), (size_t)d.H * d.W * 3 + 16, s, labels, heights, brcount, ev16, rec, d);
        return;
    }
    (void)hipMemsetAsync(rowmax, 0, (size_t)d.B * d.R * d.H * sizeof(int), s);
    // topkey: two arrays of B*R 64-bit words (top vote key, longest-run key)
    (void)hipMemsetAsync(topkey, 0, (size_t)2 * d.B * d.R * sizeof(unsigned long long), s);
    const int HW = d.H * d.W;
    hipLaunchKernelGGL(k_vote_cells, dim3((HW + kVoteCellsPerBlock - 1) / kVoteCellsPerBlock, d.B * d.R), dim3(256), 0, s, labels, ev16,
                       topkey, rowmax, d);
    hipLaunchKernelGGL(k_vote_pick, dim3((d.H + 63) / 64, d.B * d.R), dim3(64), 0, s, ev16, topkey, rowmax, d);
    hipLaunchKernelGGL(k_vote_record, dim3(d.B * d.R), dim3(64), 0, s, heights, brcount, topkey, rec, d);
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
