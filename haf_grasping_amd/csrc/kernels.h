// kernels.h -- device data layout and launch wrappers shared by engine.cpp and the kernel translation units (prestages.hip, features.hip, contraction.hip, screen.hip, recheck.hip, exact8.hip, vote.hip, prob.hip).
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
constexpr int kHListParts = 4;      // list mode of the three-pass kernel: SV tile ranges per evaluation block (k_svm_h_combine)

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

// ---- screening pass (tier 0): ONE fp16 MFMA pass on operands pre-scaled by c = sqrt(2*gamma*log2 e) ----
// u = c*x, v = c*s  =>  exp2 argument = u.v - |u|^2/2 - |v|^2/2.  Only u.v goes through the matrix core:
//   * -|v_n|^2/2 is a constant of SV n: it is the INITIAL VALUE of the accumulator column (t_n, one fp32 per SV in the
//     tile image), so the accumulator comes out as log2(K_n) + |u|^2/2 and the epilogue is v_exp_f32 + one fma with the
//     coefficient per (evaluation, SV);
//   * 2^(-|u|^2/2) is common to all SVs of an evaluation: applied once to the two class sums after the sweep.
// K holds SLOTS, not attributes: attributes that are the same function of the window with the same svm-scale range (the
// reference's Features.txt has three such pairs: rows 4/5, 19/20 and -- once the 4th weight is dropped -- 297/302) share a
// slot whose SV-side operand is the SUM of their SV components: u_a v_a + u_b v_b = u_a (v_a + v_b) exactly.  323
// attributes -> 320 slots = 10 k-steps of 32: no K padding and no half-filled tail step.  A feature file with more than
// 320 distinct attributes is served without the screening pass (engine.cpp).
// A wave keeps 64 evals x 320 slots in 160 VGPRs; a workgroup is 4 waves = 256 evals.
constexpr int kS0WaveEvals = 64;
constexpr int kS0Waves = 4;                           // waves per workgroup: TWO workgroups share a CU (one wave of each per SIMD)
constexpr int kS0BlockEvals = kS0Waves * kS0WaveEvals;
constexpr int kS0K = 320;                             // slots
constexpr int kS0Groups = kS0K / 8;                   // 40 groups of 8 slots (feature kernels)
constexpr int kS0MatBytes = kHFull * 2048;            // one 32 x 320 fp16 operand image = 20 KiB (h_image_offset, k < 320)
constexpr int kS0WavePieces = 5;                      // LDS-DMA pieces of the operand image a wave stages per tile (4 x 5 = 20 KiB);
                                                      // the 21st piece (t_n, coefficients) is wave 0's
constexpr int kS0SvTileBytes = 21504;                 // fp16 image (20480 B) + 32 floats t_n = -|v_n|^2/2 + 32 coefficients, padded to 21 KiB
constexpr int kS0Pieces = kS0SvTileBytes / 1024;      // 21 LDS-DMA wave instructions
constexpr int kS0Buffers = 3;
constexpr float kF16MinNormal = 6.103515625e-05f;     // 2^-14: smaller fp16 magnitudes are flushed in software
// ---- low-rank form of the screening pass (round 4, second half; DESIGN.md 2 "projected operand") ----
// The 299 HAF slots are LINEAR functionals of the 15x15 integral-image window (fv.cpp:141-199) up to the "%.4g" round trip and the
// fp32 roundings of the reference's own arithmetic; with the reference's Features.txt those functionals span 158 dimensions.  The
// centred operand p therefore lies within |nu| of a 158 + 21 (SHAF, passed through) = 179-dimensional subspace range(B), B with
// orthonormal columns: p.q_n = (B^'p).q~_n + p_perp.r_n with q~_n = (B'B^)^-1 B'q_n, r_n = (I - BB')q_n - (B^ - B)q~_n.  k_project
// forms y^ = fp16(B^'p^) on the matrix core (6 k-steps instead of 10 in the sweep that follows), the sweep runs on y^ and the
// images of q~_n, and the band carries |p_perp| <= |nu| (feature kernel) and the rounding of y (measured by k_project).
constexpr int kLrK = 192;                             // slots of the projected operand: 6 k-steps of 32
constexpr int kLrSteps = kLrK / 32;
constexpr int kLrMatBytes = kLrSteps * 2048;          // one 32 x 192 fp16 operand image = 12 KiB (h_image_offset, k < 192)
constexpr int kLrSvTileBytes = kLrMatBytes + 1024;    // + 32 floats t_n (0) + 32 coefficients b_n, padded to 13 KiB
constexpr int kLrWavePieces = 3;                      // LDS-DMA pieces of the image per wave and tile (4 x 3 = 12 KiB)
constexpr int kLrProjTileBytes = kHFull * 2048;       // one projection tile: 32 output slots x 320 input slots of B^' = 20 KiB (an "SV tile" of the 10-step form)

// a = h + (m + l) * 2^-12 with fp16 h, m, l (subnormal halves flushed to 0, so the value is the same whether or not the
// matrix core honours fp16 denormals); returns the value the three products actually add up to
__host__ __device__ inline double split3_f16(double a, _Float16 out[3])
{
    auto rn = [](double v) {
        _Float16 h = (_Float16)(float)v;
        float f = (float)h;
        if ((f < 0 ? -f : f) < kF16MinNormal) h = (_Float16)0.0f;
        return h;
    };
    out[0] = rn(a);
    const double r1 = a - (double)(float)out[0];
    out[1] = rn(r1 * 4096.0);
    const double r2 = r1 - (double)(float)out[1] / 4096.0;
    out[2] = rn(r2 * 4096.0);
    return (double)(float)out[0] + ((double)(float)out[1] + (double)(float)out[2]) / 4096.0;
}

// constants of the screening pass that the feature kernel needs to write the per-evaluation guard band
struct ScreenParams {
    double c;                     // sqrt(2*gamma*log2 e)
    // the SV side in SLOT space: w_n[s] = sum of c*s_n[k] over the attributes k of slot s, w^_n = fp16(w_n)
    double v_max;                 // max_n |w^_n|  (stored fp16 operand)
    double dv_max;                // max_n |w^_n - w_n|
    double das_max;               // max_n |fl32(t_n) - t_n|: the accumulator's initial value -|v_n|^2/2 held in fp32
    double as_max;                // max_n |t_n| (what the fp32 accumulation inside the matrix core starts from)
    double sigma_v;               // upper bound of the largest singular value of the N x 320 matrix of the w^_n
    double sigma_dv;              // the same for the matrix of the w^_n - w_n
    double sqrt_cmax;             // sqrt(max_n |coef_n|)
    double scale;                 // 1.001 (roundings of the band expression itself) x HAF_GUARD0_REL
    double eta_abs;               // |u' - u| beyond the relative part: fp64 roundings of the screening attribute formula (norm)
    // ---- centred form of the bilinear band term ----
    // w_n = c_n K_n is split into c_n kappa_n sc + c_n (k_n - kappa_n) sc with kappa_n = 2^(t_n + ubar.w^_n) -- the raw kernel value at
    // a reference operand ubar (the |c| kappa-weighted centroid of the support vectors; sc = the common factor 2^(-|u|^2/2)).  The first
    // part does not depend on the evaluation: its first-order error is the dot product of the operand residuals with the model constants
    // G = W^'(c kappa), Hd = (W^ - W)'(c kappa) and is CORRECTED; only the second part is bounded through the spectral norms, with
    // |c (k - kappa)|_2 <= ln2 2^zmax (sigma_dk |u^ - ubar| + ck_max |e|_2), zmax = |u^ - ubar| max|w^_n| + max|e_n|.
    double sigma_dk;              // upper bound of the largest singular value of diag(c kappa) W^   (inf: centred form switched off)
    double ck_max;                // max_n |c_n kappa_n|
    double ubar2;                 // |ubar|^2
    double acc_rel;               // fp32 accumulation inside the matrix core over the ten instructions of a chain, per unit of |t_n| + |u||w^_n|:
                                  // 10 kappa 2^-24 with kappa from probe_mfma_rounding() (engine.cpp)
    double g_norm, hd_norm;       // |G|_2, |Hd|_2 (the fp32 roundings of the correction and of u' against u are bounded through them)
    const struct ScrCorr *corr;   // kS0K per-slot constants {G, Hd, ubar} (device)
    const struct ScrCorr2 *corr2; // the same constants per PAIR of slots (packed-fp32 form of the sums: k_features_serial)
    const struct ScrDesc *sd;     // kS0K compact descriptors (device), one per SLOT, for the two-region groups
    const struct ScrDesc3 *sd3;   // kS0K general descriptors (device), one per slot
    const struct FeatDesc *fd_slot;   // kS0K feature descriptors (device): the representative attribute of every slot
    unsigned long long fast_groups;   // bit g: every slot of group g is a plain HAF feature of at most two regions
    unsigned long long extra_groups;  // bit g: some slot of group g is shared by more than one attribute (ScrDesc::extra != 0)
    // ---- centred-remainder form (round 4; cr != 0: this instance serves k_svm_screen<SCREEN_CR_EXP / SCREEN_CR_POLY>) ----
    // The kernel is translation invariant: with a centre mu (slot space, the |c| kappa-weighted centroid of the support vectors),
    // p = u - mu, q_n = w_n - mu, b_n = c_n 2^(-|q_n|^2/2), A = 2^(-|p|^2/2), z_n = p.q_n, psi(z) = 2^z - 1 - z ln2 >= 0:
    //     dec + rho = A [ B0 + L + sum_n b_n psi(z_n) ],   B0 = sum b_n,  L = ln2 p.g,  g = sum b_n q_n   (model constants, fp64 at load).
    // B0 and L are exact per evaluation; only the remainder goes through the fp16 matrix core, and everything its roundings are
    // relative to is S_psi = sum|b_n| psi(z_n) -- for a trained model with a large C 1e4 times smaller than S = sum|c_n| K_n.  The
    // translation is folded into the descriptors' scr_add (the feature kernels are the same code), L rides in the `cr` sum
    // (corr.hd = ln2 g, corr.g = corr.ub = 0).  Band (screen_finish_cr): psi(z) = (ln2 z)^2/2 + psi3(z); the quadratic part's
    // first-order error is p'N dp + p'M p with the SIGNED matrices N = Q'BQ^, M = Q'B(Q^ - Q) -- the two classes cancel in them --
    // bounded through their spectral norms; psi3' = ln2 psi >= 0 costs ln2 eps_max S_psi.
    int    cr;                    // 0: the plain / SUMSQ forms above
    int    cr_poly;               // the band is written for the polynomial epilogue (SCREEN_CR_POLY): truncation instead of the v_exp_f32 term
    double cr_nN, cr_nM;          // |N|_2, |sym M|_2
    double cr_nHabs, cr_nDabs;    // |Q^' |B| Q^|_2, |dQ' |B| dQ|_2   (second order)
    double cr_nHaa;               // | |Q^|' |B| |Q^| |_2 (entries' magnitudes): sum|b_n||z_n||a_n| <= acc_rel sqrt(p'H_abs p) sqrt(|p^|'H_aa|p^|),
                                  // since |a_n| <= acc_rel sum_k |p^_k q^_nk| -- for support vectors spread in all directions far below acc_rel |p^||p| C_a
    double cr_Ca, cr_Cq1, cr_Cqq; // sum|b||q^||q|, sum|b||q^|, sum|b||q^|^2   (accumulation inside the matrix core; sum|b|2^z through sum|b||z|)
    double cr_Babs;               // sum|b_n|
    double cr_qmax, cr_dqmax;     // max_n |q^_n|, max_n |q^_n - q_n|
    double cr_gnorm;              // |g|_2
    double cr_mu_norm, cr_mu_norm_t;   // |mu| in slot space / over all attributes: |u'| <= |p'| + |mu| (the missing "%g" round trip is relative to u')
    // tier 1 in the centred-remainder form (the three-pass list kernel behind SCREEN_CR_POLY, contraction.hip): the exact-form feature
    // kernel subtracts the centre (raw attribute units) before the hi/lo split and sums L = sum (x_f - m_f) gl_f in fp64
    const double *cr_t1_tab;      // device: [kKP] centre m_f, then [kKP] gl_f = ln2 * 2 gamma' * sum_n b_n (s_nf - m_f);  nullptr: plain form
    double *cr_t1_L;              // device: L per list slot
    // ---- low-rank form (cr != 0 only; lr != 0: k_features_serial leaves the RAW sums {su2, sd2, sx2, L, nu2} where the band goes and
    // k_svm_screen<., ., kLrSteps> finishes the band in its tail: screen_finish_cr_lr, features.hip / screen_band.h) ----
    int    lr;
    double lr_rho;                // what the columns of the slots' linear map stick out of range(B) (fp64 roundings of the basis), per unit of the largest window corner
    const int *lr_negflags;       // device, per (cloud, roll): bit 1 = the grid holds a negative height (the wave-level exactness test of the region sums needs a monotone integral image)
    const unsigned long long *lr_iiabs;   // device, per (cloud, roll): sum of |height| over the grid in units of 2^-20 m (rounded up) >= every |corner| of its integral image
};
// constants of the low-rank band (LrBand: by value to k_svm_screen's low-rank instantiation; everything rounded up at load)
struct LrBand {
    double sigB, sigAbsB;         // sigma(B^), sigma(|B^|) (entries' magnitudes: accumulation error of the projection)
    double acc10, acc6;           // 10 kappa u / 6 kappa u: the projection's and the sweep's chains
    double qmax, dqmax, rmax;     // max_n |q~^_n|, max_n |q~^_n - q~_n|, max_n |r_n|
    double nN1, nM1, nN2;         // |Q'b Q~^|, |sym(Q'b dQ~ B^')|, |Q'b R|   (signed: the classes cancel)
    double nHabs, nDabs, nRabs;   // sigma(sqrt|b| Q~^)^2, sigma(sqrt|b| dQ~)^2, sigma(sqrt|b| R)^2
    double sQb, sQtaa;            // sigma(sqrt|b| Q), sigma(sqrt|b| |Q~^|): spectral form of the accumulation term
    double Ca, Cq1, Cqq, Babs;    // sum|b||q~^||q|, sum|b||q~^|, sum|b||q~^|^2, sum|b|
    double gnorm;                 // |g|
    double mu_norm, mu_norm_t;    // as ScreenParams::cr_mu_norm*
    double eta_abs, scale;
    int poly, pad;
    // ---- plain epilogue on the same operands (SCREEN_PLAIN in the low-rank form: sum b_n 2^(z_n), one v_exp_f32 and one fma per element;
    // lr_finish_band_plain): the plain form's band (features.hip: screen_finish) in the projected coordinates, centre = reference operand ----
    double sig_q, sig_dq, sig_r;  // sigma(Q~^), sigma(dQ~), sigma(R)
    double sbq, sbr;              // sigma(diag(b) Q~), sigma(diag(b) R): |b (2^z - 1)|_2 <= ln2 2^zmax (sbq |y_e| + sbr |p_perp|)
    double rho_norm;              // |R'b|: the first-order effect of p_perp on the evaluation-independent part (bounded, not corrected)
    double gt_norm;               // |Q~^'b| (the rounding of y against it: bounded)
    double bg_norm, bh_norm;      // |B^ Q~^'b|, |B^ dQ~'b|: the two correction vectors of the feature kernel (ScrCorr::g, ::hd)
    double gabsB;                 // | |B^| |Q~^'b| |_2: accumulation error of the projection against the first correction vector
};
// constants of tier 1's centred-remainder band (k_svm_h_combine_cr): the same bound as screen_finish_cr with the operands'
// errors those of the hi+lo split (2^-22 relative) and the accumulation that of the PRECISE three-pass form ((kappa + 14) u)
struct CrT1Params {
    double B0, rho;
    double c;                     // sqrt(2 gamma log2 e): |p| = c |x - m|
    double nN, nM, nHabs, nDabs, nHaa, Ca, Cqq, Babs, qmax, dqmax;   // as ScreenParams::cr_*, with Q~ = hi + lo of fl32(s - m) in place of Q^
    double acc_rel;               // (kappa + 14) 2^-24
    double dp_rel, dp_abs;        // |p~ - p| <= dp_rel |p| + dp_abs (fp32 rounding of x - m, fp16 hi + lo, flushed lo subnormals)
    double sum_rel;               // fp32 part of the coefficient sum + the polynomial's roundings, relative to S_psi
    double scale;                 // 1.001 x HAF_GUARD_REL
    float guard_abs;
    int gv0, gv1, pad;
};
// constants of the centred-remainder form that the contraction kernel's tail needs (fp64: B0 + L cancels against rho)
struct CrParams { double B0, rho; };
// variants of k_svm_screen (screen.hip)
enum { SCREEN_PLAIN = 0, SCREEN_SUMSQ = 1, SCREEN_CR_EXP = 2, SCREEN_CR_POLY = 3, SCREEN_VARIANTS = 4 };
// Compact descriptor of one attribute slot for the screening feature pass: 64 bytes, one s_load_dwordx16.  A slot without a
// feature (beyond the feature file, norm slots) is all zero and evaluates to exactly 0.
struct ScrDesc {
    int    off[8];                // BYTE offsets of the corners of regions 0 and 1 (A-B-C+D each) from the window origin, in
                                  // the LDS band of a wave (kBandPitch floats per row, features.hip: screen_quad)
    float  w[2];                  // region weights (0: region inactive, its corners point at the window origin)
    // round 5: the scaling runs in fp32 -- u' = fmaf(fl32(q4), scr_mul, scr_add) -- with the fp64 constants of FeatDesc rounded once
    // (what that costs, 3 u (|u'| + |scr_add|) per slot, is part of eta: kScreenEtaRel, ScreenParams::eta_abs)
    float  scr_mul;               // fl32(c * (upper - lower) * RN(1/(fmax - fmin)))   (0 for an attribute svm-scale drops)
    float  scr_add;               // fl32(c * lower - fmin * scr_mul [- centre])        (0 likewise)
    float  extra;                 // attributes sharing this slot beyond the first (0 almost everywhere): |u|^2 counts u'^2 that often more
    float  pad;                   // low-rank form: |scr_mul| rounded up (0 for a slot that is not a linear functional of the window: passed through)
    float  rsv[2];
};
static_assert(sizeof(ScrDesc) == 64, "ScrDesc is one 64-byte scalar load");
// per-slot constants of the centred band (ScreenParams): one 16-byte scalar load
struct ScrCorr { float g, hd, ub, pad; };
struct ScrCorr2 { float g[2], hd[2], ub[2], pad[2]; };   // slots 2p, 2p+1
// The same for any feature (three regions, HAF or SHAF rule): the groups that are not in ScreenParams::fast_groups
struct ScrDesc3 {
    int    off[12];               // band BYTE offsets of the corners of regions 0..2
    float  w[3];
    int    shaf;                  // fv.cpp:187-191 instead of the weighted sum
    float  scr_mul, scr_add;      // as in ScrDesc
    float  extra;                 // as in ScrDesc
    float  pad[5];                // pad[0]: as ScrDesc::pad
};
static_assert(sizeof(ScrDesc3) == 96, "ScrDesc3 layout");
constexpr int kBandPitch = 80;     // floats per row of a wave's integral-image band in LDS: 64 + 14 columns, padded
// per-evaluation guard band of the screening pass, written by the feature kernel (4 floats per evaluation):
//   |dec^ - dec| <= min(gA * |w|_2, gC * S) + (guard_acc0 + gB) * S + cm * (|dec^| + |rho|) + guard_abs,  S = sum|coef|K,
//   |w|_2 = sqrt(sum (coef K)^2), measured by the SUMSQ variant of k_svm_screen or bounded by sqrt(max|coef| * S)
// with {gA, gB, gC, cm} per evaluation, and for the CENTRED estimate dec^ - corr * sc (ScreenParams):
//   |dec^ - corr sc - dec| <= abs_c * sc + (guard_acc0 + gB) * S + cm (|dec^ - corr sc| + |corr sc| + |rho|) + guard_abs
// {corr, abs_c} in raw space (the contraction kernel applies the common factor sc); the kernel decides from whichever of the
// two estimates has the narrower band.  8 floats per evaluation: {gA, gB, gC, cm, corr, abs_c, 0, 0}
constexpr int kBandFloats = 8;

struct CloudDev {
    const float *xyz;
    int n;
    int stride;      // floats per point
    float m0[8];     // rows 0 and 1 of the transform WITHOUT roll and x-scale (T(0,0,z_shift) Rx Rz T(-centre), server.cpp:462-483):
                     // where a point lands before the gripper roll -- only used to pre-sort the cloud into spatial buckets
    int sorted_off;  // first point of this cloud in the bucket-sorted copy (points), bucket_off: its first bucket counter
    int bucket_off;
};

// per (cloud, roll): rows 0..2 of the fp32 transform (server.cpp:483) and the rotated-rectangle scalars of
// pnt_in_box (server.cpp:679-696), all computed on the host with the reference's float/double mix
struct RollGeo {
    float m[12];
    float sa, ca;                 // sinf(alpha), cosf(alpha)
    float cx1, cy1, cx2, cy2, cx3, cy3, cx4, cy4;
    float rc, rs, rw;             // cos / sin of the roll angle and the x-scale of this roll's transform: m = S(rw) R(roll) m0
    float pad;
};

// one HAF/SHAF feature (fv.cpp:141-199) with its svm-scale range (svm-scale.c:333-353)
struct FeatDesc {
    int   off[3][4];              // II offsets of the 4 corners of region k relative to the window origin: A-B-C+D
    int   offw[3][4];             // the same inside a 15x15 copy of the window (row pitch 15)
    float w[3];                   // region weights; the 4th region's weight is always 0 in the reference (CHaarFeature.cpp:56-60)
    int   active;                 // bit k: region k survives the skip rule (fv.cpp:155-159)
    int   shaf;                   // feature index >= nr_features_without_shaf
    int   skip;                   // attribute dropped by svm-scale (feature_max == feature_min, svm-scale.c:336)
    int   pad;
    double fmin, fmax;
    double range, inv_range;      // fmax - fmin and RN(1 / (fmax - fmin))
    double scr_mul, scr_add;      // screening pass (ScrDesc below): u' = fma(q4, scr_mul, scr_add)
    float  scr_extra;             // fd_slot entries only: attributes sharing the slot beyond the first
    float  pad2;                  // fd_slot entries only: |scr_mul| rounded up (low-rank form; as ScrDesc::pad)
};

// haf_attr_record of include/hafgrasp.h: the three stages of one attribute of one evaluation (HAF_FLAG_KEEP_DEBUG)
struct AttrRecord { float feature, pad; double q4, scaled; };

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
    float guard_dot_p;            // the same for the three-pass kernel's PRECISE form (list mode behind the screening pass)
    float guard_acc0;             // screening pass, plain variant: two-level fp32 coefficient sum + exp2 + product, per unit of sum|coef|K
    float guard_acc0_s;           // screening pass, SUMSQ variant: single-level sum
    float guard_acc_l;            // guard_acc of the three-pass kernel's list mode when the SV tiles are cut into ranges (k_svm_h_combine)
    int   gv0, gv1;               // grid values of label[0] / label[1] (atoi of the "%g" label text, server.cpp:843)
    float sqrt_cmax;              // screening pass, plain variant: |w|_2 <= sqrt(max|coef| * S)
};

struct ExactParams {
    double gamma, rho, lower, upper;
    double guard2, as_max1;       // fp64 GEMM-form guard: |dec| <= guard2 * (as_max1 + gamma'|x|^2) * sum|coef|K -> strict order
    double gamma2;                // gamma * log2(e)
    int n_sv, n_sv_pad, kx;       // kx = rows of the fp64 k-major SV image
    int gv0, gv1;
    int kernel_type, degree;      // libsvm's kernel_type (parsers.h: HAF_KERNEL_*; 2 = RBF) -- the strict tier serves all four vector kernels
    double coef0;
};

// counters[] slots in device memory
enum { CNT_EVALS = 0, CNT_FLAGGED = 1, CNT_ERROR = 2, CNT_FLAGGED2 = 3, CNT_FLAGGED0 = 4,
       CNT_INEXACT = 5,    // (cloud, roll) grids whose integral image needed the sequential summation order
       CNT_FLAGGEDI = 6,   // evaluations the exact-integer tier (exact8.hip) handed on to the fp64 MFMA tier
       CNT_FLAGGED0B = 7,  // evaluations the second screening pass (centred-remainder form on the first one's list, "tier 0b") could not decide
       // the short-list gate in front of tier 1 (launch_short_list_gate, contraction.hip):
       CNT_BYPASS = 8,     // evaluations it sent straight to the fp64 MFMA tier
       CNT_T1_N = 9,       // the length of tier 1's input list as tier 1 sees it (0 when the gate took the list)
       CNT_T1_ADD = 10,    // what tier 1's hand-over adds to CNT_FLAGGED (the gate's entries when no exact-integer tier stands in between)
       CNT_COUNT = 12 };

// ---- tier 2a: the decision function on EXACT integer dot products (exact8.hip) ------------------------------------------
// What limits the fp16 contractions is the fp32 accumulation inside the matrix core (~2.6e-6 relative per kernel value); what
// makes the fp64 MFMA tier slow is the fp64 matrix rate (1/32 of fp16).  In between: attributes and support vectors as
// fixed-point numbers (kI8Q fractional bits for the attributes, |value| < 15.9; q_s <= 30 for the SVs, as many as the model's largest
// component allows), split into four balanced base-128 digits (-64..63, int8); the 16
// digit-by-digit products of a 384-long dot product go through v_mfma_i32_16x16x64_i8 and are EXACT in int32 (<= 4 x 384 x 4096 per
// accumulator, one accumulator per digit weight), so |xq - sq|^2 of the quantised vectors is exact and the only error of a kernel
// value is the quantisation: |d - dq| <= sqrt(324) (2^-(kI8Q+1) + 2^-(q_s+1)), i.e. ~2e-7 relative instead of 5.6e-6.  An evaluation still
// inside that (9x narrower) band goes on to the fp64 MFMA tier as before.
// (The range: svm-scale does not clamp, and a SHAF attribute of -1 against a range [0, 0.30] scales to -7.64: the attributes of
// real windows reach +-7.7, so 23 fractional bits -- +-15.87 -- it is; an evaluation or a model beyond that skips the tier.)
constexpr int kI8Q = 23;                          // fractional bits of the attributes: |value| <= 63 * (2^21 + 2^14 + 2^7 + 1) * 2^-23 = 15.87
constexpr long kI8Max = 63L * (2097152L + 16384L + 128L + 1L);
constexpr int kI8Slices = 4;                      // digits per value
constexpr int kI8Steps = 6;                       // k-steps of 64 attributes: 384 >= 324
constexpr int kI8GroupBytes = kI8Slices * kI8Steps * 1024;   // operand image of 16 evaluations (or 16 SVs): [digit][k-step][lane][16 int8]
constexpr int kI8SvTileBytes = kI8GroupBytes + 256;          // + 16 doubles |sq_n|^2 + 16 doubles coef
struct I8Params {
    double gamma, rho;
    double gamma2;                // gamma * log2(e)
    double drop;                  // | |xq - sq|^2 computed - exact | from the digit products the contraction leaves out (weights 128, 1)
    double delta;                 // |(x - xq) - (s - sq)|_2 <= delta = sqrt(324) (2^-(kI8Q+1) + 2^-(q_s+1))
    double dq_scale;              // -2 * 2^(14 - kI8Q - q_s): the integer dot product (in units of 128^2) as a term of |xq - sq|^2
    double s_max;                 // max_n |sq_n|_2
    double guard_scale;           // 1 (HAF_GUARD_I8_REL in the testing build)
    int n_sv_pad, gv0, gv1, pad;
};
void launch_recheck_i8(const float *ii, const int *evalcell, const FeatDesc *fd, const void *sv_i8, I8Params p, double lower, double upper,
                       const int *flag_list, int window_cap, int list_off, int *counters, void *ximg, double *part64, double *dec_exact,
                       int8_t *labels, int *flagi_list, int flagi_cap, Dims d, hipStream_t s,
                       unsigned long long *words = nullptr);   // != nullptr: one ballot word per 64 entries of a window -> ORDERED hand-over (device_common.h)

// fp64 model image for the rechecks: attribute-major [kM64Rows][n_sv_pad]; rows 0..323 attributes (model order of SVs),
// row 324 |s|^2, row 325 coef
constexpr int kM64Rows = 326;

struct RollRecordDev { int vote; short row, col; float h_locmax; int n_evals; };

// probability-output mode (prob.hip)
struct ProbParams {
    double A, B;          // probA, probB of the model
    int gv0, gv1;         // (int)atof(first two characters of "%g" of label[0] / label[1])  (server.cpp:833)
    float hdr;            // what the "labels a b" header line parses to: the value of the first masked cell of a roll
    int host_all;         // testing build (HAF_PROB_HOST_ALL): every estimate is finished on the host
    double dec_slack;     // how far the device's libsvm-order decision value may be from glibc's: 4 x 2^-52 sum|coef| (k_prob_eval)
};
void launch_probability_eval(const double *dec_exact, const int *evalcell, const int *counters, ProbParams P, int8_t *labels, float *own,
                             double *ptext, int *near_list, int near_cap, int *counters_rw, long evals_cap, hipStream_t s);
void launch_prob_list(int *counters, int slot, int *list, int cap, hipStream_t s);
void launch_probability(const double *dec_exact, const int *evalcell, const int *counters, ProbParams P, int8_t *labels,
                        const uint8_t *mask, const int *rowcount, const int *brcount, const float *heights, float *own,
                        double *ptext, float *gridf, float *evf, RollRecordDev *rec, long evals_cap, Dims d, hipStream_t s);

void launch_fill_i32(int *p, int v, size_t n, hipStream_t s);
// scratch of the bucket-sorted binning path (large grids): see prestages.hip
struct BinScratch {
    float *sorted;       // [total points][3] the clouds' points grouped by bucket (original coordinates)
    int *bkt_count;      // [B][nb*nb + 1]
    int *bkt_off;        // [B][nb*nb + 1] exclusive scan of the counts (+ total)
    int *bkt_cursor;     // [B][nb*nb + 1]
    long sorted_cap;     // points the sorted copy holds
    int bkt_cap;         // bucket counters per array
};
int bin_bucket_grid(int H, int *bucket_cells);     // buckets per side for a grid of H cells; bucket edge in cells
// returns true when the bucket-sorted path ran (it also fills the empty cells: no separate fill launch needed)
bool launch_bin(const CloudDev *clouds, const CloudDev *clouds_host, int max_n, long total_n, const RollGeo *geo, int *hkeys, Dims d,
                float r_row, float r_col, bool bucket_ok, BinScratch bs, int *counters, hipStream_t s);   // counters[CNT_ERROR]: a tile's bucket list overflowed
// small grids: a1 (tail) + a2 + a3 + a4 in one launch, one workgroup per (cloud, roll); false when the grid does not fit LDS
bool launch_small_pre(const CloudDev *clouds, const RollGeo *geo, int max_n, int *hkeys, float *ii, uint8_t *mask, int *rowcount,
                      int *brcount, int8_t *labels, int *evalcell, int *counters, int *flag_list, bool direct, Dims d, float r_row,
                      float r_col, hipStream_t s, unsigned long long *brslot = nullptr, unsigned epoch = 0);
size_t small_pre_lds(int H, int W);
void launch_integral(int *hkeys_heights, double *rowsum, float *ii, int *inexact_flags, int *counters, Dims d, hipStream_t s,
                     unsigned long long *abs_total = nullptr);   // per (cloud, roll): sum of |height| in units of 2^-20 m, rounded up (parallel form only; low-rank screening form)
void launch_mask_count(const float *ii, const RollGeo *geo, uint8_t *mask, int *rowcount, Dims d, hipStream_t s);
void launch_scan(const int *rowcount, int *rowoff, int *brcount, int *counters, Dims d, hipStream_t s);
void launch_compact(const uint8_t *mask, const int *rowcount, const int *rowoff, int *evalcell, Dims d, hipStream_t s);
// operand image the feature kernels write
enum { XMODE_F32 = 0, XMODE_SPLIT = 1, XMODE_SCREEN = 2,
       XMODE_F64 = 3,     // the fp64 attribute image of the fp64 MFMA tier (X = double *, [group of 16][324][16]); no ax
       XMODE_I8 = 4 };    // the int8 digit image of the exact-integer tier (kI8GroupBytes per 16 slots); ax = double * |xq|^2 per slot
// idx_list == nullptr: evaluations 0..counters[CNT_EVALS]; otherwise slot j takes evaluation idx_list[j],
// j < min(counters[list_counter], list_cap) (the screened evaluations that go on to the three-pass kernel)
void launch_features(const float *ii, const int *evalcell, const int *counters, const FeatDesc *fd, float *X, float *ax,
                     Dims d, double lower, double upper, float neg_gamma2, long max_evals, int xmode, ScreenParams sp,
                     const int *idx_list, int list_counter, int list_cap, bool large, long sel_evals, AttrRecord *dbg,
                     float *ax2, hipStream_t s,    // ax2: screening form only, -|u|^2/2 per evaluation
                     int list_off = 0);            // list mode: idx_list points at entry list_off of the list counted by list_counter
double probe_mfma_rounding(hipStream_t s, double *worst16);  // largest error of one v_mfma_f32_16x16x32_f16 (*worst16: 16x16x16f16) in units of 2^-24 (|c| + sum|a b|) over adversarial inputs; < 0: HIP error
int probe_f16_subnormal_mfma(hipStream_t s);   // 1: the MFMA takes fp16 subnormal operands at their value, 0: it flushes, -1: HIP error
// low-rank form: Y = fp16(B^' X) per evaluation (X: 10-step operand images, Y: 6-step images), |y^ - y|^2 into raw[kBandFloats e + 5]
void launch_project(const void *X0, const void *btiles, void *Y, float *raw, const int *counters, long max_evals, hipStream_t s);
void launch_svm_screen_lr(const void *Y, float *raw, const float *nax, const void *svt_lr, const int *evalcell, const int *counters,
                          SvmParams p, float *dec, int8_t *labels, unsigned long long *flag0_words, int *wgcount, int *flag0_list,
                          int flag0_cap, int *counters_rw, Dims d, long max_evals, float *margin, int variant, CrParams cr, LrBand lb,
                          hipStream_t s, int also_counter = -1,
                          const void *ptiles = nullptr,
                          // round 5, GATHER form ("tier 0b" behind a low-rank first pass): the centred-remainder/exp epilogue on the LIST idx_list
                          // (counters[count_slot] entries, at most flag0_cap) straight from the first pass's images and raw sums; the
                          // compacted output list's length goes to counters[out_slot]
                          const int *idx_list = nullptr, int count_slot = CNT_EVALS, int out_slot = CNT_FLAGGED0);   // != nullptr: the FUSED form -- Y holds the 10-step images, ptiles the projection tiles by input k-step (no launch_project)
void launch_svm_screen(const void *X0, const float *gband, const float *nax, const void *svt0, const int *evalcell, const int *counters,
                       SvmParams p, float *dec, int8_t *labels, unsigned long long *flag0_words, int *wgcount, int *flag0_list,
                       int flag0_cap, int *counters_rw, Dims d, long max_evals, float *margin, int variant, CrParams cr, hipStream_t s,
                       int also_counter = -1,    // >= 0: the compacted list's length is published in this counter too (capped at flag0_cap)
                       const int *idx_list = nullptr, int count_slot = CNT_EVALS,   // list mode: slot j of X0 / gband / nax holds evaluation idx_list[j], counters[count_slot] of them
                       int out_slot = CNT_FLAGGED0,
                       void *part_buf = nullptr, int parts = 0);   // SV-range split of requests that do not fill the chip (screen.hip: PART form): a buffer of screen_part_bytes(); parts 0 = by the live count, 1 = never, n = forced (tests)
size_t screen_part_bytes();                               // the counter that receives the length of the compacted list
void launch_svm(const float *X, const float *ax, const float *svt, const int *evalcell, const int *counters,
                SvmParams p, float *dec, int8_t *labels, int *flag_list, int flag_cap, int *counters_rw, Dims d,
                long max_evals, hipStream_t s,
                unsigned char *t1flags = nullptr);   // != nullptr: one byte per entry -> ORDERED hand-over to the exact tiers' list (contraction.hip: k_t1_handover)
// A SHORT list in front of tier 1 (a small request against a big model: a handful of undecided evaluations, three tiers of launches and
// minimum chains behind them) goes straight to the fp64 MFMA tier's list: at most max_entries entries of src_list (counters[src_slot] of
// them) are copied to dst_list and counters[CNT_T1_N] -- what tier 1 reads as its list's length from then on -- is 0, else the length.
// dst_direct: the count goes to counters[dst_slot] (the exact-integer tier's OUTPUT list: that tier then sees an empty input);
// otherwise to CNT_T1_ADD, which tier 1's hand-over adds to the CNT_FLAGGED it publishes (dst_list is the list it appends to).
void launch_short_list_gate(int *counters, int src_slot, const int *src_list, int src_cap, int *dst_list, int dst_slot, bool dst_direct,
                            int max_entries, hipStream_t s);
size_t t1_flag_bytes(long max_entries);                   // size of the t1flags buffer for launches of up to max_entries entries (flags + per-block counts)
void launch_svm_h(const void *Xh, const float *ax, const void *svt_h, const int *evalcell, const int *counters,
                  SvmParams p, float *dec, int8_t *labels, int *flag_list, int flag_cap, int *counters_rw, Dims d,
                  long max_evals, const int *idx_list, int list_counter, int list_cap, double *part_out, long part_stride,
                  hipStream_t s,
                  const CrT1Params *cr = nullptr, const double *Lbuf = nullptr,    // list mode only: the centred-remainder form (svt_h = its images)
                  unsigned char *t1flags = nullptr);   // as launch_svm
// tiny requests: exact attributes + fp64 MFMA decision + label in one launch (tier 2's arithmetic for every evaluation)
void launch_small_direct(const float *ii, const int *evalcell, int *counters, const FeatDesc *fd, const double *sv64, ExactParams p, Dims d,
                         long max_evals, double *dec_exact, int8_t *labels, int *flag2_list, int flag2_cap, AttrRecord *dbg, hipStream_t s,
                         const int *idx_list = nullptr, int list_counter = 0, int list_off = 0);   // list mode: window [list_off, list_off + max_evals) of idx_list
void launch_recheck(const float *ii, const int *evalcell, const FeatDesc *fd, const double *sv64, const double *coef64,
                    ExactParams p, const int *flag_list, int flag_cap, const int *counters, int counter_slot,
                    double *dec_exact, int8_t *labels, Dims d, hipStream_t s);
// the same tier for a list whose length the host knows: spread over (evaluation groups x SV chunks), terms = scratch of
// terms_slots x n_sv_pad doubles (recheck.hip)
void launch_recheck_known(const float *ii, const int *evalcell, const FeatDesc *fd, const double *sv64, const double *coef64,
                          ExactParams p, const int *flag_list, int n_flag, double *terms, int terms_slots,
                          double *dec_exact, int8_t *labels, Dims d, hipStream_t s);
void launch_recheck_mfma(const float *ii, const int *evalcell, const FeatDesc *fd, const double *sv64, ExactParams p,
                         const int *flag_list, int window_cap, int list_off, int *counters, double *x64, double *part64,
                         double *dec_exact, int8_t *labels, int *flag2_list, int flag2_cap, Dims d, hipStream_t s,
                         AttrRecord *dbg = nullptr,    // dbg: attribute records of a request that went straight to this tier
                         bool have_x64 = false,        // the attribute image is in place already (launch_features XMODE_F64)
                         int counter_slot = CNT_FLAGGED,    // the counter of flag_list (CNT_FLAGGEDI behind the exact-integer tier)
                         unsigned long long *words = nullptr);   // as launch_recheck_i8: ordered hand-over to the strict tier's list
constexpr int kRecheckPartRows = 2 * 8 + 1;       // part64: [2 * kMSplit + 1][flag_cap] doubles (partial sums + |x|^2)
void launch_vote(const int8_t *labels, const float *heights, const int *brcount, short *ev16, unsigned long long *topkey,
                 int *rowmax, RollRecordDev *rec, Dims d, hipStream_t s);
void launch_mfma_accum_test(const void *a, const void *b, const float *c0, float *out, int trials, hipStream_t s);   // testkernels.hip (testing build)
void launch_mfma_rate_test(const void *in, float *out, int blocks, int iters, hipStream_t s);                       // testkernels.hip (testing build)
void launch_mfma_model_test(const void *in, float *out, int mb, int blocks, int tiles, hipStream_t s);             // testkernels.hip (testing build)
void launch_i8_layout_probe(const void *a, const void *b, int *c, hipStream_t s);                                 // testkernels.hip (testing build)
void launch_f16_mfma_probe(const void *a, const void *b, const float *c, float *d, int trials, int chain, hipStream_t s);   // testkernels.hip (testing build)
void launch_decq_test(const double *in, double *out, int n, int P, hipStream_t s);
void launch_scale_test(const double *q4, const double *fmin, const double *fmax, double lower, double upper, double *out,
                       int n, hipStream_t s);

}  // namespace haf
