// vote.hip -- k_vote_cells / k_vote_pick / k_vote_record / k_vote_small: show_predicted_gps 865-932 (29-tap vote, first-wins
// argmax, longest-run centring) and the z window of transform_gp_in_wcs_and_publish 1342-1351
// (src/calc_grasppoints_action_server.cpp)
//
// Built with -ffp-contract=off: every fp32/fp64 expression that must match the CPU restatement bit for bit is
// written with explicit *_rn intrinsics as well; fma() is used only where a fused operation is intended.
#include "device_common.h"

namespace haf {

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
    extern __shared__ __attribute__((aligned(16))) short s_ev[];   // [H*W] votes, then [H*W] labels as bytes
    __shared__ unsigned long long red[256];
    __shared__ int s_top, s_row, s_col;
    const int br = blockIdx.x, t = threadIdx.x;
    const int H = d.H, W = d.W, HW = H * W;
    int8_t *s_g = reinterpret_cast<int8_t *>(s_ev + HW);
    const int8_t *g = labels + (size_t)br * HW;
    short *ev = ev16 + (size_t)br * HW;
    if ((HW & 15) == 0) {                                 // sixteen labels per load (byte by byte this loop was HW / 256 dependent round trips)
        for (int k = t; k < HW / 16; k += 256) reinterpret_cast<uint4 *>(s_g)[k] = reinterpret_cast<const uint4 *>(g)[k];
    } else {
        for (int k = t; k < HW; k += 256) s_g[k] = g[k];
    }
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

}  // namespace haf
