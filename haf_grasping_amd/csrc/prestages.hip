// prestages.hip -- gfx950 kernels of everything in front of the feature stage (src/calc_grasppoints_action_server.cpp):
//   k_bin / k_bin_lds / k_bkt_* / k_bin_tiles   generate_grid 406-529 (transform + max-z binning)
//   k_integral_totals / k_integral_band (k_integral_seq as fallback, k_integral_small)   generate_grid 522-528 (empty cells -> 0) +
//                    calc_intimage 577-613
//   k_mask_count / k_scan / k_compact   pnt_in_box 666-749 + the row-major cell order of calc_featurevectors 637-643
//   k_small_pre      all of the above for a small request in one launch
//
// Built with -ffp-contract=off: every fp32/fp64 expression that must match the CPU restatement bit for bit is
// written with explicit *_rn intrinsics as well; fma() is used only where a fused operation is intended.
#include "device_common.h"

namespace haf {

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
    const int last = min(c.n, first + kBinChunk);
    // the thread's kBinChunk / 256 points are loaded up front, behind them the private grid is cleared (as a loop: a chain of eight global-load latencies)
    static_assert(kBinChunk % 256 == 0, "points per thread");
    float xs[kBinChunk / 256], ys[kBinChunk / 256], zs[kBinChunk / 256];
#pragma unroll
    for (int j = 0; j < kBinChunk / 256; j++) {
        const int i = first + threadIdx.x + 256 * j;
        const float *p = c.xyz + (size_t)(i < last ? i : first) * c.stride;
        xs[j] = p[0]; ys[j] = p[1]; zs[j] = p[2];
    }
    for (int k = threadIdx.x; k < HW; k += 256) cells[k] = key_empty;
    __syncthreads();
    const RollGeo &g = geo[br];
#pragma unroll
    for (int j = 0; j < kBinChunk / 256; j++) {
        if (first + (int)threadIdx.x + 256 * j >= last) continue;
        const float x = xs[j], y = ys[j], z = zs[j];
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
    // (eight cells per thread and step, their global reads in flight together: one read -> compare -> atomic per step was a chain of
    // HW / 256 L2 latencies, 13 of them on the reference's grid -- most of this kernel's 17.6 us at C3)
    for (int k0 = threadIdx.x; k0 < HW; k0 += 256 * 8) {
        int v[8], o[8];
#pragma unroll
        for (int j = 0; j < 8; j++) v[j] = (k0 + 256 * j < HW) ? cells[k0 + 256 * j] : key_empty;
#pragma unroll
        for (int j = 0; j < 8; j++) o[j] = out[min(k0 + 256 * j, HW - 1)];
#pragma unroll
        for (int j = 0; j < 8; j++)
            if (v[j] > key_empty && v[j] > o[j]) atomicMax(&out[k0 + 256 * j], v[j]);    // stale read is safe: the cell only grows
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
    if (!(inexact_flags[br] & 1)) return;                 // the parallel form was exact for this grid (the normal case); bit 1: a negative height (low-rank form)
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
                                                               int *__restrict__ inexact_flags, unsigned long long *__restrict__ abs_total, Dims d)
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
    bool inexact = false, negative = false;
    // sum of |height| over this thread's cells, in units of 2^-20 m rounded UP (integers: the total is the same in whatever order the
    // workgroups add it -- identical calls take identical paths): no corner of the integral image exceeds the grid's total
    unsigned long long habs = 0ull;
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
                if (row0 + r < H) {
                    const float hh = final_height(kr[r]);
                    negative |= !(hh >= 0.0f);            // (also a NaN height)
                    habs += (unsigned long long)ceil(fmin((double)fabsf(hh), 1048576.0) * 1048576.0) + ((fabsf(hh) <= 1048576.0f) ? 0ull : (1ull << 62));
                    v = add_checked(v, (double)hh, inexact);
                }
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
    // bit 1: the grid holds a negative height (heights in (-0.99, 0) survive generate_grid, server.cpp:522-528): the integral image is
    // then not monotone and the exactness argument of the low-rank screening form (features.hip: k_features_serial, LR) does not hold
    if (__syncthreads_or(negative) && tid == 0) atomicOr(&inexact_flags[br], 2);
    if (abs_total) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) habs += __shfl_xor(habs, o, 64);
        __shared__ unsigned long long wabs[kIThreads / 64];
        if (lane == 0) wabs[wave] = habs;
        __syncthreads();
        if (tid == 0) {
            unsigned long long t = 0ull;
            for (int w = 0; w < kIThreads / 64; w++) t += wabs[w];
            atomicAdd(&abs_total[br], t);                 // one per workgroup (integer sum: order-independent; a height beyond 2^20 m sets bit 62: the bound is then useless, as it must be)
        }
    }
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

void launch_integral(int *hk, double *rowsum, float *ii, int *inexact_flags, int *counters, Dims d, hipStream_t s, unsigned long long *abs_total)
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
    if (abs_total) (void)hipMemsetAsync(abs_total, 0, (size_t)d.B * d.R * sizeof(unsigned long long), s);
    hipLaunchKernelGGL(k_integral_totals, dim3(n_bands, d.B * d.R), dim3(kIThreads), 0, s, hk, rowsum, inexact_flags, abs_total, d);
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
    // (round 5: the row's REMAINDER -- its count modulo 64 -- is taken from the row's START, the whole waves from behind it: a row of a
    // full-grid request starts in column 7, and a wave whose first window starts in column 0 of the integral image never passes the
    // low-rank feature kernel's wave-wide exactness test (features.hip: T < dmin) -- at C5 that was one of the seven whole waves of
    // every row on the slower per-region bounds.  The remainders form waves that are no run of neighbours either way.)
#ifdef HAF_OLD_CHUNKING      // (variant builds only: the A/B of the round-5 change -- the remainder at the row's END, as until then)
    const int cnt = rowcount[br * H + i], whole = cnt & ~63, rem = 0;
    const int base_a = rowoff[br * H + i], base_b = rowoff[n + 1 + br * H + i] - whole;
#else
    const int cnt = rowcount[br * H + i], whole = cnt & ~63, rem = cnt - whole;
    const int base_a = rowoff[br * H + i] - rem, base_b = rowoff[n + 1 + br * H + i];
#endif
    int done = 0;
    for (int j0 = 0; j0 < W; j0 += 64) {
        int j = j0 + lane;
        bool m = (j < W) && mrow[j];
        unsigned long long bal = __ballot(m);
        if (m) {
            const int rank = done + __popcll(bal & ((1ull << lane) - 1ull));
#ifdef HAF_OLD_CHUNKING
            evalcell[(rank < whole ? base_a : base_b) + rank] = (br * H + i) * W + j;
#else
            evalcell[(rank >= rem ? base_a : base_b) + rank] = (br * H + i) * W + j;
#endif
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
                                                                Dims d, float r_row, float r_col, int key_empty,
                                                                unsigned long long *__restrict__ brslot, unsigned epoch)
{
    extern __shared__ double s_rs[];                      // [H][pitch] fp64 row sums | [H*W] keys -> heights | [(H+1)*(W+1)] II | [H] counts
    __shared__ int s_base;
    __shared__ int s_red[kSmallPreThreads / 64];
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
        // Where this (cloud, roll)'s evaluations go in the list.  Round 5: the evaluations of (cloud, roll) 0, 1, 2 ... follow each other
        // in THAT order, run after run (until then a segment was reserved with the atomicAdd's return value: whichever workgroup came
        // first; labels never depended on it, the contents of list windows and the debug lists did).  Every workgroup publishes its
        // count together with the request's epoch in ONE 64-bit word -- no fence, no zeroing between requests -- and sums the words of the
        // workgroups in front of it (below).  Grids of up to 256 workgroups only: those are all resident at once, so nobody waits for a
        // workgroup that cannot start; larger grids keep the reservation by arrival.
        const bool ordered = brslot != nullptr && gridDim.x <= 256;
        if (ordered) __hip_atomic_store(&brslot[br], ((unsigned long long)epoch << 32) | (unsigned)run, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int arrived = run ? atomicAdd(&counters[CNT_EVALS], run) : 0;
        s_base = ordered ? -1 : arrived;
        if (direct && run) atomicAdd(&counters[CNT_FLAGGED], run);
    }
    __syncthreads();
    if (s_base < 0) {                                      // (uniform: ordered)
        int part = 0;
        for (int t = tid; t < br; t += kSmallPreThreads) {
            unsigned long long v;
            do {
                v = __hip_atomic_load(&brslot[t], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if ((unsigned)(v >> 32) != epoch) __builtin_amdgcn_s_sleep(1);
            } while ((unsigned)(v >> 32) != epoch);
            part += (int)(unsigned)(v & 0xffffffffull);
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) part += __shfl_xor(part, o, 64);
        if (lane == 0) s_red[wave] = part;
        __syncthreads();
        if (tid == 0) {
            int b = 0;
            for (int w = 0; w < kSmallPreThreads / 64; w++) b += s_red[w];
            s_base = b;
        }
        __syncthreads();
    }
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
                      float r_col, hipStream_t s, unsigned long long *brslot, unsigned epoch)
{
    const size_t lds = small_pre_lds(d.H, d.W);
    if (lds > 64 * 1024) return false;                    // (the default dynamic-LDS limit: grids up to ~58 x 58)
    float minus_one = -1.0f;
    int key_empty;
    memcpy(&key_empty, &minus_one, 4);
    key_empty ^= 0x7FFFFFFF;                              // ordered key of -1.0f (499-501): an empty cell
    if (max_n <= kSmallPreMaxPoints) {
        hipLaunchKernelGGL(k_small_pre<true>, dim3(d.B * d.R), dim3(kSmallPreThreads), lds, s, clouds, geo, hkeys, ii, mask, rowcount, brcount,
                           labels, evalcell, counters, flag_list, direct ? 1 : 0, d, r_row, r_col, key_empty, brslot, epoch);
    } else {
        // a large cloud: many workgroups bin it (k_bin_lds: LDS-private grids, one global atomicMax per touched cell), then the rest
        launch_fill_i32(hkeys, key_empty, (size_t)d.B * d.R * d.H * d.W, s);
        dim3 grid((max_n + kBinChunk - 1) / kBinChunk, d.B * d.R);
        hipLaunchKernelGGL(k_bin_lds, grid, dim3(256), (size_t)d.H * d.W * sizeof(int), s, clouds, geo, hkeys, d, r_row, r_col, key_empty);
        hipLaunchKernelGGL(k_small_pre<false>, dim3(d.B * d.R), dim3(kSmallPreThreads), lds, s, clouds, geo, hkeys, ii, mask, rowcount, brcount,
                           labels, evalcell, counters, flag_list, direct ? 1 : 0, d, r_row, r_col, key_empty, brslot, epoch);
    }
    return true;
}

}  // namespace haf
