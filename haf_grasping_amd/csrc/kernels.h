// kernels.h -- device data layout and launch wrappers shared by engine.cpp and kernels.hip.
#pragma once

#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

namespace haf {

// ---- fixed geometry of the RBF contraction kernel -------------------------------------------------
constexpr int kKP = 324;            // attribute dimension padded to an even count (2 k-values per MFMA step)
constexpr int kKSteps = kKP / 2;    // 162 v_mfma_f32_32x32x2_f32 per 32x32 output tile
constexpr int kDP = 328;            // rows per tile image: kKP attribute rows + row kKP (a_s) + row kKP+1 (coef), padded to 8
constexpr int kTile = 32;           // evals per A tile / SVs per B tile
constexpr int kTileFloats = kDP * kTile;          // 10496 floats = 41 KiB: 41 LDS-DMA wave instructions of 1 KiB
constexpr int kSvmBlockEvals = 256; // 8 waves x 32 evals
constexpr int kSvmThreads = 512;

// ---- split-fp16 variant of the contraction (three fp16 MFMA passes: hi*hi + lo*hi + hi*lo) ----
// MFMA shape 16x16x32 (+ one 16x16x16 step for the K tail): on MI355X the chip sustains ~15 % more FLOP/s on this
// shape than on 32x32x16 under DVFS (tools/ubench/mfma_shape.hip: 1.92 vs 1.67 PFLOP/s), at equal cycles per FLOP.
constexpr int kHK = 336;                              // attributes padded to 10 k-steps of 32 + one of 16
constexpr int kHFull = 10;                            // 16x16x32 steps
constexpr int kHSteps = kHK / 16;                     // 21 groups of 16 attributes (layout bookkeeping)
constexpr int kHMatBytes = kHSteps * 1024;            // one 32 x 336 fp16 operand image = 21 KiB, see h_image_offset()
constexpr int kHTailOff = kHFull * 2048;              // byte offset of the 16-wide K tail inside an operand image
constexpr int kHXTileBytes = 2 * kHMatBytes;          // X tile: hi image + lo image = 42 KiB per 32 evals
constexpr int kHSvTileBytes = 44032;                  // SV tile: hi + lo + 32 a_s + 32 coef (43264 B) padded to 43 KiB
constexpr int kHSvPieces = kHSvTileBytes / 1024;      // 43 LDS-DMA wave instructions
constexpr int kHBuffers = 3;                          // LDS ring: two tiles in flight behind the one being computed

// Byte offset of element (row r of 32, attribute k of 336) inside one fp16 operand image.  The image is the register
// image of the MFMA operands: for k-step s (32 attributes) and row block m (16 rows) the 64 lanes of a wave hold
// 8 consecutive attributes each, lane = ((k%32)/8)*16 + r%16  ->  [s][m][lane][8 halfs]; the K tail (k >= 320) is the
// 16x16x16 form with 4 attributes per lane -> [m][lane][4 halfs].
__host__ __device__ inline int h_image_offset(int r, int k)
{
    const int m = r >> 4, row = r & 15;
    if (k < kHFull * 32) return (((k >> 5) * 2 + m) * 64 + ((k >> 3) & 3) * 16 + row) * 16 + (k & 7) * 2;
    const int kk = k - kHFull * 32;
    return kHTailOff + (m * 64 + (kk >> 2) * 16 + row) * 8 + (kk & 3) * 2;
}

struct CloudDev {
    const float *xyz;
    int n;
    int stride;      // floats per point
};

// per (cloud, roll): rows 0..2 of the fp32 transform (server.cpp:483) and the rotated-rectangle scalars of
// pnt_in_box (server.cpp:679-696), all computed on the host with the reference's float/double mix
struct RollGeo {
    float m[12];
    float sa, ca;                 // sinf(alpha), cosf(alpha)
    float cx1, cy1, cx2, cy2, cx3, cy3, cx4, cy4;
    float pad[2];
};

// one HAF/SHAF feature (fv.cpp:141-199) with its svm-scale range (svm-scale.c:333-353)
struct FeatDesc {
    int   off[3][4];              // II offsets of the 4 corners of region k relative to the window origin: A-B-C+D
    float w[3];                   // region weights; the 4th region's weight is always 0 in the reference (CHaarFeature.cpp:56-60)
    int   active;                 // bit k: region k survives the skip rule (fv.cpp:155-159)
    int   shaf;                   // feature index >= nr_features_without_shaf
    int   skip;                   // attribute dropped by svm-scale (feature_max == feature_min, svm-scale.c:336)
    int   pad;
    double fmin, fmax;
    double range, inv_range;      // fmax - fmin and RN(1 / (fmax - fmin))
};

struct Dims {
    int H, W, R, B;               // grid, rolls in this launch, clouds
    int nf;                       // feature rows (324)
    int n_sv, n_sv_tiles;         // SV tile images: coefficients >= 0 first, padded to whole tiles, then the negative ones
    int sv_tile_neg;              // first tile of the negative-coefficient group (== n_sv_tiles when there is none)
};

struct SvmParams {
    float two_gamma2;             // 2*gamma*log2(e)
    float neg_gamma2;             // -gamma*log2(e)
    float rho;
    // |dec| <= (guard_acc + guard_dot * (|a_x| + as_max)) * sum|coef|K + guard_abs  ->  fp64 recheck tier
    float guard_acc;              // fp32 accumulation of the coefficient sum over the SV tiles, exp2, argument roundings
    float guard_dot;              // 324-term fp32 dot-product chain, per unit of (a_x + a_s)
    float guard_abs;
    float as_max;                 // max_n gamma*log2(e)*|s_n|^2
    int   gv0, gv1;               // grid values of label[0] / label[1] (atoi of the "%g" label text, server.cpp:843)
};

struct ExactParams {
    double gamma, rho, lower, upper;
    double guard2, as_max1;       // fp64 GEMM-form guard: |dec| <= guard2 * (as_max1 + gamma'|x|^2) * sum|coef|K -> strict order
    double gamma2;                // gamma * log2(e)
    int n_sv, n_sv_pad, kx;       // kx = rows of the fp64 k-major SV image
    int gv0, gv1;
};

// counters[] slots in device memory
enum { CNT_EVALS = 0, CNT_FLAGGED = 1, CNT_ERROR = 2, CNT_FLAGGED2 = 3, CNT_COUNT = 8 };

// fp64 model image for the rechecks: attribute-major [kM64Rows][n_sv_pad]; rows 0..323 attributes (model order of SVs),
// row 324 |s|^2, row 325 coef
constexpr int kM64Rows = 326;

struct RollRecordDev { int vote; short row, col; float h_locmax; int n_evals; };

void launch_fill_i32(int *p, int v, size_t n, hipStream_t s);
void launch_bin(const CloudDev *clouds, int max_n, const RollGeo *geo, int *hkeys, Dims d, float r_row, float r_col,
                hipStream_t s);
void launch_integral(int *hkeys_heights, double *rowsum, float *ii, Dims d, hipStream_t s);
void launch_mask_count(const float *ii, const RollGeo *geo, uint8_t *mask, int *rowcount, Dims d, hipStream_t s);
void launch_scan(const int *rowcount, int *rowoff, int *brcount, int *counters, Dims d, hipStream_t s);
void launch_compact(const uint8_t *mask, const int *rowoff, int *evalcell, Dims d, hipStream_t s);
void launch_features(const float *ii, const int *evalcell, const int *counters, const FeatDesc *fd, float *X, float *ax,
                     Dims d, double lower, double upper, float neg_gamma2, long max_evals, bool split_f16, hipStream_t s);
void launch_svm(const float *X, const float *ax, const float *svt, const int *evalcell, const int *counters,
                SvmParams p, float *dec, int8_t *labels, int *flag_list, int flag_cap, int *counters_rw, Dims d,
                long max_evals, hipStream_t s);
void launch_svm_h(const void *Xh, const float *ax, const void *svt_h, const int *evalcell, const int *counters,
                  SvmParams p, float *dec, int8_t *labels, int *flag_list, int flag_cap, int *counters_rw, Dims d,
                  long max_evals, hipStream_t s);
void launch_recheck(const float *ii, const int *evalcell, const FeatDesc *fd, const double *sv64, const double *coef64,
                    ExactParams p, const int *flag_list, int flag_cap, const int *counters, int counter_slot,
                    double *dec_exact, int8_t *labels, Dims d, hipStream_t s);
void launch_recheck_mfma(const float *ii, const int *evalcell, const FeatDesc *fd, const double *sv64, ExactParams p,
                         const int *flag_list, int flag_cap, int *counters, double *x64, double *dec_exact, int8_t *labels,
                         int *flag2_list, int flag2_cap, Dims d, hipStream_t s);
void launch_vote(const int8_t *labels, const float *heights, const int *brcount, short *ev16, unsigned long long *topkey,
                 RollRecordDev *rec, Dims d, hipStream_t s);
void launch_decq_test(const double *in, double *out, int n, int P, hipStream_t s);
void launch_scale_test(const double *q4, const double *fmin, const double *fmax, double lower, double upper, double *out,
                       int n, hipStream_t s);

}  // namespace haf
