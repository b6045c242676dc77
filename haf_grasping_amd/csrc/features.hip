// features.hip -- gfx950 feature kernels:
//   k_features_serial / k_features / k_features_small   CIntImage_to_Featurevec::calc_featurevalue (fv.cpp:141-199), the "%.4g" text
//                    round trip (fv.cpp:133 -> svm-scale.c:270), svm-scale restore+output (svm-scale.c:333-353) and
//                    the "%g" round trip (svm-scale.c:350 -> svm-predict.c:108); large / small requests
//                    (the screening form, XMODE_SCREEN, skips the second round trip and writes the guard band instead:
//                    screen_finish / screen_finish_cr)
//   k_features<XMODE_F64>   the attribute vectors of the guard-band evaluations for the fp64 MFMA tier (recheck.hip)
//   k_small_direct   a small request: features and the exact decision in one launch
// (the device functions that evaluate one feature value are in feature_device.h)
//
// Built with -ffp-contract=off: every fp32/fp64 expression that must match the CPU restatement bit for bit is
// written with explicit *_rn intrinsics as well; fma() is used only where a fused operation is intended.
#include "feature_device.h"

namespace haf {

// X image, split-fp16 form: per tile of 32 evals two operand images (hi, lo) of [21 k-steps][2 k-halves][32 evals][8 fp16];
// a thread finishes 8 attributes, then stores them as one 16-byte vector per image (512 contiguous bytes per 32 lanes).
// X image, screening form: per tile of 32 evals ONE operand image of the same layout holding fp16(c*x) plus the norm
// slots (kernels.h); the per-evaluation guard band goes where the other forms keep a_x.

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
__device__ __forceinline__ _Float16 screen_operand(float ud, ScreenSums &a, const Corr &k)
{
    const float f = ud;                              // u' IS an fp32 number since round 5 (feature_device.h: the scaling runs in fp32)
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
__device__ __forceinline__ half2v screen_operand2(float u0, float u1, ScreenSums2 &a, ScrCorr2K k)
{
    const f32x2 f = {u0, u1};
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
__device__ __forceinline__ void screen_extra_norm(const ScreenParams &sp, int g, const float *ud, float &sx)
{
    if (!((sp.extra_groups >> g) & 1)) return;
#pragma unroll
    for (int q = 0; q < 8; q++) {
        const float ex = constant_ptr(sp.sd)[g * 8 + q].extra;
        if (ex != 0.0f) {
            const float f = ud[q];
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
constexpr int kSplitSmallList = 8192;                  // the same rule for tier 1's lists (XMODE_SPLIT in list mode): 512 workgroups of 16 fill the chip once
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
// LR (screening form only; kernels.h: kLrK, ScreenParams::lr): the wave also sums nu2 >= |p' - p_lin|^2 over the HAF slots and the
// kernel leaves the RAW sums {su2, sd2, sx2, L, nu2} where the finished band would go (k_project adds |y^ - y32|^2, the sweep's tail
// finishes the band: screen_band.h).  Per slot the bound is the "%.4g" rounding as it happened plus the fp32 roundings of the products
// and their sum, bounded (round 5; the fast paths measured them until then) (feature_device.h: screen_quad / screen_pair3 / screen_attribute_lr), plus, for a region whose sum is not
// provably EXACT in the reference's own order ((a - b) - c) + d (fv.cpp:161-162), the three roundings of that order.  Three paths,
// wave-uniform: (A) a run of 64 neighbours whose windows pass the exactness test AS A WHOLE -- no negative height in the grid (integral
// image monotone), bottom row of the band <= 2 x its top row in each of the lane's 15 columns (a - b exact by Sterbenz), the window's
// total below its top-left corner (then |s2| = d - R <= d and s3 = R are multiples of ulp(d) below 2^24 ulp(d)) -- pays three
// instructions per slot; (B) any other run of neighbours tests every region on its own four corners (region_round_bound); (C) a wave
// that is not a run of neighbours does the same through the per-lane loads.
// (round 5: SEVEN waves per SIMD -- 72 registers; the low-rank screening instance sat at 73, one register above it, and runs 5 % faster
// with three of them spilled: 3.34 -> 3.18 ms at C5, A/B on one box, profiles/r05_serial_waves_ab.txt.  HAF_SERIAL_WAVES: variant builds)
#ifndef HAF_SERIAL_WAVES
#define HAF_SERIAL_WAVES 7
#endif
template <int MODE, bool LR>
__global__ __launch_bounds__(256, HAF_SERIAL_WAVES) void k_features_serial(
const float *__restrict__ ii, const int *__restrict__ evalcell,
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
    bool lr_exact = false, lr_neg = false;
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
            if (LR) {
                const volatile float *vb = bw;
                const float tl = vb[lane], bl = vb[14 * kBandPitch + lane], tr = vb[lane + 14], br = vb[14 * kBandPitch + lane + 14];
                // per column of the band: bottom row <= 2 x top row (78 bits); a lane's window needs its own 15 columns
                const unsigned long long colok0 = __ballot(bl <= 2.0f * tl);
                const unsigned long long colok1 = __ballot(lane >= 14 || vb[14 * kBandPitch + 64 + lane] <= 2.0f * vb[64 + lane]) & 0x3fffull;
                const unsigned long long mine = (lane == 0) ? colok0 : ((colok0 >> lane) | (colok1 << (64 - lane)));
                bool ok = (mine & 0x7fffull) == 0x7fffull;
                // the window's total, rounded up past its own three roundings
                const float tA = __fsub_rn(br, tr), tB = __fsub_rn(bl, tl);
                const float T = (__fsub_rn(tA, tB) + 2.0e-7f * (fabsf(tA) + fabsf(tB))) * 1.0001f;
                // every region's d = II[x1][y1] is at least the window's first corner -- or, for the window that starts in column 0 of the
                // integral image (which is all zeros: a region with y1 = 0 has c = d = 0 and nothing to round), its second
                const int broll = cell0 / (d.H * d.W);
                const int col0 = cell0 - (cell0 / d.W) * d.W - 7;                      // first column of the band
                const float dmin = (col0 == 0 && lane == 0) ? vb[1] : tl;
                ok = ok && T < dmin;
                lr_neg = (sp.lr_negflags[broll] & 2) != 0;
                lr_exact = __ballot(ok) == ~0ull && !lr_neg;
                asm volatile("" ::: "memory");
            }
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
        if (MODE == XMODE_SCREEN) { const float zb[kBandFloats] = {0.0f, 0.0f, 0.0f, 0.0f, LR ? __builtin_inff() : 0.0f, 0.0f, 0.0f, 0.0f}; store_band(ax + kBandFloats * e, zb); ax2[e] = 0.0f; }
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
        const bool lr_nb = LR && fastwave && lr_exact;    // wave-uniform: every region sum of the wave is exact
        float nu2 = 0.0f;
        float rmin = 0.0f;                                // path A: the smallest computed region sum of this evaluation (feature_device.h: screen_quad)
        for (int g = 0; g < kS0Groups; g++) {             // 40 groups of 8 SLOTS (kernels.h)
            float ud[8];
            if (lr_nb) {
                if ((sp.fast_groups >> g) & 1) {
                    screen_quad<1>(band, constant_ptr(sp.sd) + g * 8, st, ud, nu2, rmin);
                    screen_quad<1>(band, constant_ptr(sp.sd) + g * 8 + 4, st, ud + 4, nu2, rmin);
                } else {
#pragma unroll
                    for (int q = 0; q < 8; q++) screen_pair3<1, 1>(band, constant_ptr(sp.sd3) + g * 8 + q, st, ud + q, nu2, rmin);
                }
            } else if (LR && fastwave) {                  // a wave with regions that may round: bounded region by region
                if ((sp.fast_groups >> g) & 1) {
                    screen_quad<2>(band, constant_ptr(sp.sd) + g * 8, st, ud, nu2, rmin);
                    screen_quad<2>(band, constant_ptr(sp.sd) + g * 8 + 4, st, ud + 4, nu2, rmin);
                } else {
#pragma unroll
                    for (int q = 0; q < 8; q++) screen_pair3<2, 1>(band, constant_ptr(sp.sd3) + g * 8 + q, st, ud + q, nu2, rmin);
                }
            } else if (fastwave && ((sp.fast_groups >> g) & 1)) {   // wave-uniform
                screen_quad(band, constant_ptr(sp.sd) + g * 8, st, ud);
                screen_quad(band, constant_ptr(sp.sd) + g * 8 + 4, st, ud + 4);
            } else if (fastwave) {
#pragma unroll
                for (int q = 0; q < 8; q += 2) screen_pair3(band, constant_ptr(sp.sd3) + g * 8 + q, st, ud + q);
            } else if (LR && !fastwave) {
#pragma unroll
                for (int q = 0; q < 8; q++) {
                    const FeatDesc &F = fd[g * 8 + q];
                    float nbq = 0.0f;
                    ud[q] = F.skip ? 0.0f : screen_attribute_lr(SrcBuf<true>{iir, w0}, F, st, nbq);
                    nu2 = fmaf(nbq, nbq, nu2);
                }
            } else {
#pragma unroll
                for (int q = 0; q < 8; q++) {
                    const FeatDesc &F = fd[g * 8 + q];               // screening form: fd = one descriptor per SLOT (an unused slot has skip = 1)
                    ud[q] = F.skip ? 0.0f : screen_attribute(SrcBuf<true>{iir, w0}, F, st);
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
        if (LR) {
            // raw sums; the part of the slots' linear map that the fp64 roundings of the basis leave outside range(B): lr_rho per unit of the largest corner
            // (no corner of the integral image exceeds the sum of |height| over its grid: prestages.hip, k_integral_totals)
            const float lr_M = (float)((double)sp.lr_iiabs[evalcell[e_src] / (d.H * d.W)] * 9.5367431640625e-07) * 1.0001f;
            const float rho = (float)sp.lr_rho * lr_M * 1.000001f;
            nu2 = fmaf(rho, rho, nu2);
            if (rmin < 0.0f) nu2 = __builtin_inff();                              // (path A met a negative computed region sum: never trusted, next tier)
            const float sx2 = acc.su2 + sx;                                   // (one more fp32 rounding of |p'|^2: inside kF32Acc's 326)
            band[0] = acc.su2; band[1] = acc.sd2; band[2] = sx2; band[3] = acc.cr; band[4] = nu2; band[5] = 0.0f; band[6] = acc.ub; band[7] = 0.0f;   // ([6]: L of the centred-remainder form when this pass runs the plain epilogue, else 0: engine_tables.cpp)
            nax = -0.5f * sx2;
        } else {
            screen_finish((double)acc.su2, (double)acc.sd2, (double)acc.su2 + (double)sx, (double)acc.cr, (double)acc.ub, sp, band, nax);
        }
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
// The 15 x 15 integral-image windows of a workgroup's NEV evaluations go to LDS (s_win[evaluation * kWinPitch + row * 15 + col], origin
// of evaluation v in s_w0[v], 0xffffffff = none: zeros).  Sixteen lanes take one window row.  Every load of a thread is in flight before
// the first is stored: as a loop this compiled to load -> s_waitcnt vmcnt(0) -> ds_write per step, NEV * 240 / THREADS serial L2
// latencies in front of every block of evaluations (round 5: 30 of them in k_features<., 8>, ~20 us of C3's 49 us feature launch).
template <int NEV, int THREADS>
__device__ __forceinline__ void stage_windows(const rsrc_t iir, const unsigned *s_w0, float *s_win, int row_floats)
{
    // per step only a compile-time LDS offset and a wave-uniform row offset (the load's SGPR operand) change: a thread keeps its
    // evaluation(s) -- with per-step vector addresses the thirty of them were all computed up front, 38 VGPRs
    constexpr int kRows16 = THREADS / 16;                            // window rows one step of the workgroup covers
    static_assert((NEV & (NEV - 1)) == 0 && THREADS % 16 == 0 && (kRows16 % NEV == 0 || NEV % kRows16 == 0), "a row per 16 lanes");
    constexpr int kEvs = kRows16 >= NEV ? 1 : NEV / kRows16;         // evaluations a thread alternates between
    constexpr int kRowStep = kRows16 >= NEV ? kRows16 / NEV : 1;     // rows it advances per visit of the same evaluation
    constexpr int kSteps = kEvs * ((15 + kRowStep - 1) / kRowStep);
    const int s0 = threadIdx.x >> 4, col = threadIdx.x & 15;
    const int wev0 = kEvs == 1 ? (s0 & (NEV - 1)) : s0, x0 = kEvs == 1 ? s0 / NEV : 0;
    int voff[kEvs];
    bool ok[kEvs];
#pragma unroll
    for (int m = 0; m < kEvs; m++) {
        const unsigned o = s_w0[wev0 + kRows16 * m];
        ok[m] = o != 0xffffffffu && col < 15;
        voff[m] = ok[m] ? (int)(o + (unsigned)(x0 * row_floats + col) * 4u) : 0;
    }
    float *dst = s_win + wev0 * kWinPitch + x0 * 15 + col;
    float v[kSteps];
#pragma unroll
    for (int k = 0; k < kSteps; k++) {
        const int m = k % kEvs, x = (k / kEvs) * kRowStep;
        const float t = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(iir, voff[m], x * row_floats * 4, 0));
        v[k] = ok[m] ? t : 0.0f;
    }
#pragma unroll
    for (int k = 0; k < kSteps; k++) {
        const int m = k % kEvs, x = (k / kEvs) * kRowStep;
        if (col < 15 && x0 + x < 15) dst[m * kRows16 * kWinPitch + x * 15] = v[k];
    }
}

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
    if ((MODE == XMODE_SPLIT || MODE == XMODE_SCREEN) && idx_list && n_evals <= kSplitSmallList) return;   // (tier 1's and tier 0b's lists)
    const long n_pad = ((long)n_evals + kBlock - 1) / kBlock * kBlock;
    if ((long)blockIdx.x * kFeatEvals >= n_pad) return;
    __shared__ double s_tab[MODE == XMODE_SCREEN ? 1 : hafq::kTabDoubles];
    __shared__ unsigned long long s_scr[MODE == XMODE_SCREEN ? hafq::kScrTabWords : 1];
    hafq::PtrTabs tb{};
    hafq::ScrTabs st{};
    if (MODE == XMODE_SCREEN) st = load_screen_tables(s_scr);
    else tb = load_decimal_tables(s_tab);
    const int ev = threadIdx.x & 63, gl = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // (the wave index in an SGPR: descriptor words by scalar loads -- as a VGPR expression they were ~17 vector loads and as many waits per slot)
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
    stage_windows<kFeatEvals, kFeatWaves * 64>(iir, s_w0, s_win, d.W + 1);
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
        float udf[8];
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
            const float xf = (float)xd;                                // (screening form: xd IS an fp32 number)
            udf[q] = xf;
            if (MODE == XMODE_SCREEN) {
                hi[q] = screen_operand(xf, acc, constant_ptr(sp.corr)[f]);
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
            screen_extra_norm(sp, g, udf, sx);
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
    __shared__ double red_l[(MODE == XMODE_SPLIT) ? kSmSlots : 1][kSmEvals];   // centred-remainder form of tier 1: see k_features
    const double *t1_tab = (MODE == XMODE_SPLIT) ? sp.cr_t1_tab : nullptr;
    // (list mode: slot j holds evaluation idx_list[j] of the window [list_off, list_off + list_cap) of the list counted by list_counter)
    const int n_evals = idx_list ? window_count(counters[list_counter], list_off, list_cap) : counters[CNT_EVALS];
    // XMODE_I8: lists are of unknown length at launch; both feature kernels are launched and the list's length decides on the
    // device which of them works -- this one (a third of the serial chain per thread: latency) up to kI8SmallList entries, the
    // 64-evaluation workgroups of k_features (half the time per evaluation at 100 k entries: 2.9 against 5.5 ns) beyond
    if (MODE == XMODE_I8 && n_evals > kI8SmallList) return;
    // XMODE_SPLIT in list mode (tier 1; round 5): the same rule with kSplitSmallList -- C3 against the 8 964-SV model leaves tier 1 a list of
    // ~4 000 entries, 72 workgroups of k_features<., 16> on 256 CUs: 54 us
    if ((MODE == XMODE_SPLIT || MODE == XMODE_SCREEN) && idx_list && n_evals > kSplitSmallList) return;    // (the same for tier 0b's list: C3 against a 4 096-SV model, k_features<2, 16> 34 us)
    const long n_pad = ((long)n_evals + kBlock - 1) / kBlock * kBlock;
    if ((long)blockIdx.x * kSmEvals >= n_pad) return;
    // (list mode is launched for the list's CAPACITY -- at C5 123 k workgroups for a list of a few hundred entries, 90 us of empty
    // workgroups -- so its grid is capped and the workgroups stride over the blocks of 16 evaluations)
    __shared__ double s_tab[MODE == XMODE_SCREEN ? 1 : hafq::kTabDoubles];
    __shared__ unsigned long long s_scr[MODE == XMODE_SCREEN ? hafq::kScrTabWords : 1];
    // the descriptors this mode reads, once per workgroup (feature_device.h: FeatDescX / FeatDescS)
    __shared__ FeatDescX s_fdx[MODE == XMODE_SCREEN ? 1 : kKP];
    __shared__ FeatDescS s_fds[MODE == XMODE_SCREEN ? kS0K : 1];
    if (MODE == XMODE_SCREEN) stage_descriptors<kSmWaves * 64>(fd, kS0K, s_fds);
    else stage_descriptors<kSmWaves * 64>(fd, min(d.nf, kKP), s_fdx);
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
    stage_windows<kSmEvals, kSmWaves * 64>(iir, s_w0, s_win, d.W + 1);
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
    double lsum = 0.0;
    if (has_group) {
#pragma unroll
        for (int q = 0; q < 8; q++) {
            const int f = g * 8 + q;
            double xd = 0.0;
            if (MODE == XMODE_SCREEN) {
                const FeatDescS &F = s_fds[f];                           // screening form: fd = one descriptor per SLOT (an unused slot has skip = 1)
                if (live && !F.skip) xd = screen_attribute(src, F, st);
                const float ex = F.scr_extra;                            // (per quarter wave here: the group differs between them)
                if (ex != 0.0f) { const float ff = (float)xd; sx = fmaf(ex * ff, ff, sx); }
            } else if (live && f < d.nf && f < kKP) {
                xd = attribute_value_rec(src, s_fdx[f], lower, upper, tb, (dbg) ? dbg + (size_t)e_src * kKP + f : nullptr);
                if (MODE == XMODE_SPLIT && t1_tab && f < kKP) {
                    xd -= t1_tab[f];
                    lsum = fma(xd, t1_tab[kKP + f], lsum);
                }
            }
            const float xf = (float)xd;
            if (MODE == XMODE_SCREEN) {
                hi[q] = screen_operand(xf, acc, sp.corr[f]);           // (per quarter wave: a 16-byte vector load; xd IS an fp32 number here)
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
    if (MODE == XMODE_SPLIT) red_l[slot][ev] = lsum;
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
            if (MODE == XMODE_SPLIT && t1_tab) {
                double l = 0.0;
#pragma unroll
                for (int k = 0; k < kSmSlots; k++) l += red_l[k][ev];    // fixed order
                sp.cr_t1_L[e] = l;
            }
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
// -DHAF_PHASE_CLOCK (variant builds only: python -m haf_grasping_amd.build --variant phase -DHAF_PHASE_CLOCK; tools/phase_clock.py):
// workgroup 0 of k_small_direct leaves the 100 MHz wall clock at its phase boundaries
#ifdef HAF_PHASE_CLOCK
__device__ unsigned long long g_phase_clock[16];
#define HAF_PHASE(i) do { if (blockIdx.x == 0 && threadIdx.x == 0) g_phase_clock[i] = wall_clock64(); } while (0)
#else
#define HAF_PHASE(i) do { } while (0)
#endif
constexpr int kSdMSteps = kKP / 4;                // 81 k-steps of 4
// (round 5, measured and dropped: EIGHT evaluations per workgroup and four attributes per thread -- kSdEvals = 8 below still compiles --
// halves the attribute chain and doubles the workgroups, but every workgroup streams the whole fp64 model through its MFMA phase, whose
// sixteen rows are then half empty: 34.5 us either way at C3, 48 -> 70 us for C4's 6 476 evaluations.)
constexpr int kSdEvals = 16;
constexpr int kSdSlots = kSmWaves * 64 / kSdEvals;     // 48
constexpr int kSdAttrs = 128 / kSdEvals;               // attributes per thread: 8
static_assert(kSdSlots * kSdAttrs >= kKP && (kSdEvals == 8 || kSdEvals == 16), "slots cover the attributes");
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
    __shared__ double s_x[kKP * kSdEvals];            // [attribute][evaluation]: the A operand of the fp64 MFMA, k-major
    __shared__ double s_part[kSmWaves][kSdEvals][2];  // per wave: sum coef*K and sum |coef|*K of its SV tiles, per evaluation
    __shared__ double s_xx[kSdEvals];
    __shared__ float s_win[kSdEvals * kWinPitch];
    __shared__ unsigned s_w0[kSdEvals];
    __shared__ double s_tab[hafq::kTabDoubles];
    __shared__ FeatDescX s_fd[kKP];                   // the descriptors, once per workgroup (feature_device.h)
    const int n_evals = idx_list ? window_count(counters[list_counter], list_off, list_cap) : counters[CNT_EVALS];
    if ((long)blockIdx.x * kSdEvals >= n_evals) return;
    HAF_PHASE(0);
    stage_descriptors<kSmWaves * 64>(fd, min(d.nf, kKP), s_fd);
    const hafq::PtrTabs tb = load_decimal_tables(s_tab);
    HAF_PHASE(1);
    const int ev = threadIdx.x & (kSdEvals - 1), slot = threadIdx.x / kSdEvals, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const long e = (long)blockIdx.x * kSdEvals + ev;
    const bool live = e < n_evals;
    const int e_src = live ? (idx_list ? idx_list[e] : (int)e) : 0;    // the evaluation this slot holds
    const rsrc_t iir = make_ii_rsrc(ii, d);
    if (slot == 0) s_w0[ev] = live ? window_origin(evalcell[e_src], d.H, d.W) : 0xffffffffu;
    __syncthreads();
    HAF_PHASE(2);
    stage_windows<kSdEvals, kSmWaves * 64>(iir, s_w0, s_win, d.W + 1);
    __syncthreads();
    HAF_PHASE(3);
    const SrcWin src{s_win + ev * kWinPitch};
    const int g = slot;
    if (g < (kKP + kSdAttrs - 1) / kSdAttrs) {
#pragma unroll
        for (int q = 0; q < kSdAttrs; q++) {
            const int f = g * kSdAttrs + q;
            double xd = 0.0;
            if (live && f < d.nf && f < kKP) xd = attribute_value_rec(src, s_fd[f], p.lower, p.upper, tb, dbg ? dbg + (size_t)e_src * kKP + f : nullptr);
            if (f < kKP) s_x[f * kSdEvals + ev] = xd;
        }
    }
    __syncthreads();
    HAF_PHASE(4);
    if (threadIdx.x < 4 * kSdEvals) {                 // |x|^2: four index ranges per evaluation, each in index order, then (s0 + s1) + (s2 + s3): fixed
        static_assert(kKP % 4 == 0 && (kKP / 4) % 9 == 0 && 4 * kSdEvals <= 64, "one wave, whole batches");
        const int xe = threadIdx.x & (kSdEvals - 1), k_lo = (threadIdx.x / kSdEvals) * (kKP / 4);
        double xx = 0.0;
#pragma unroll 1
        for (int k0 = k_lo; k0 < k_lo + kKP / 4; k0 += 9) {   // (nine LDS reads in flight, then the chain: read -> wait -> fma per step was 324 LDS latencies)
            double v[9];
#pragma unroll
            for (int j = 0; j < 9; j++) v[j] = s_x[(k0 + j) * kSdEvals + xe];
#pragma unroll
            for (int j = 0; j < 9; j++) xx = fma(v[j], v[j], xx);
        }
        xx += __shfl_xor(xx, kSdEvals, 64);
        xx += __shfl_xor(xx, 2 * kSdEvals, 64);
        if (threadIdx.x < kSdEvals) s_xx[threadIdx.x] = xx;
    }
    __syncthreads();
    HAF_PHASE(5);
    // ---- fp64 MFMA over this wave's SV tiles: A[row = lane&15][k = 4s + (lane>>4)] from LDS (row & 7: the evaluation), B[k][col = lane&15] from the model ----
    const int n_tiles = p.n_sv_pad / 16;
    double part[4] = {0, 0, 0, 0}, pabs[4] = {0, 0, 0, 0};
    for (int t = wave; t < n_tiles; t += kSmWaves) {
        const double *Bg = sv64 + (size_t)t * 16 + (lane & 15);
        f64x4 acc = {0, 0, 0, 0};
        // (the B operand comes from L2 as the loop asks for it, six loads in flight.  Round 5 tried three batches of 27 loads, each in
        // flight before its first MFMA: the phase went from 8.3 to 6.7 us for wave 0 and the barrier behind it from 4.1 to 7.2 -- all
        // 255 workgroups stream the whole fp64 model here, 117 MB in ~10 us: L2 bandwidth, not latency -- at 108 instead of 76 VGPRs)
#pragma unroll 27
        for (int s = 0; s < kSdMSteps; s++) {
            const int k = 4 * s + (lane >> 4);
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(s_x[k * kSdEvals + (lane & (kSdEvals - 1))], Bg[(size_t)k * p.n_sv_pad], acc, 0, 0, 0);
        }
        const double ss = Bg[(size_t)kKP * p.n_sv_pad];
        const double cf = Bg[(size_t)(kKP + 1) * p.n_sv_pad];
#pragma unroll
        for (int r = 0; r < kSdEvals / 4; r++) {                               // (kSdEvals = 8: rows 8-15, registers 2 and 3, repeat rows 0-7)
            const int row = (lane >> 4) + 4 * r;                               // f64 C/D layout: row = (lane>>4) + 4*reg
            const double d2 = fma(-2.0, acc[r], s_xx[row] + ss);
            const double kv = exp(-p.gamma * d2);
            part[r] = fma(cf, kv, part[r]);
            pabs[r] = fma(fabs(cf), kv, pabs[r]);
        }
    }
    HAF_PHASE(6);
#pragma unroll
    for (int r = 0; r < kSdEvals / 4; r++) {
        double v = part[r], w = pabs[r];
        v += __shfl_xor(v, 8, 64); w += __shfl_xor(w, 8, 64);
        v += __shfl_xor(v, 4, 64); w += __shfl_xor(w, 4, 64);
        v += __shfl_xor(v, 2, 64); w += __shfl_xor(w, 2, 64);
        v += __shfl_xor(v, 1, 64); w += __shfl_xor(w, 1, 64);
        if ((lane & 15) == 0) { s_part[wave][(lane >> 4) + 4 * r][0] = v; s_part[wave][(lane >> 4) + 4 * r][1] = w; }
    }
    __syncthreads();
    HAF_PHASE(7);
    if (threadIdx.x < kSdEvals && live) {
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
    HAF_PHASE(8);
}

void launch_small_direct(const float *ii, const int *evalcell, int *counters, const FeatDesc *fd, const double *sv64, ExactParams p, Dims d,
                         long max_evals, double *dec_exact, int8_t *labels, int *flag2_list, int flag2_cap, AttrRecord *dbg, hipStream_t s,
                         const int *idx_list, int list_counter, int list_off)
{
    const long nb = (max_evals + kSdEvals - 1) / kSdEvals;
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
        if (MODE == XMODE_SCREEN && sp.lr && !idx_list)
            hipLaunchKernelGGL((k_features_serial<MODE, MODE == XMODE_SCREEN>), dim3((unsigned)blocks), dim3(256), 0, s, ii, evalcell, counters, fd, X, ax, d,
                               lower, upper, neg_gamma2, sp, idx_list, list_counter, list_cap, dbg, ax2);
        else
            hipLaunchKernelGGL((k_features_serial<MODE, false>), dim3((unsigned)blocks), dim3(256), 0, s, ii, evalcell, counters, fd, X, ax, d,
                               lower, upper, neg_gamma2, sp, idx_list, list_counter, list_cap, dbg, ax2);
        return;
    }
    // small requests -- and every list of the fp64 tier, which is short unless the model is ill-conditioned: a third of the
    // serial chain per thread (C3's ~8 000 flagged evaluations: 27 us against 56 us with the 64-evaluation workgroups)
    if ((MODE == XMODE_SPLIT || MODE == XMODE_SCREEN) && idx_list) {   // tier 1's / tier 0b's list: its length decides on the device (kSplitSmallList)
        const long cap_small = std::min<long>(max_evals, kSplitSmallList);
        long nb = ((cap_small + kBlock - 1) / kBlock) * (kBlock / kSmEvals);
        hipLaunchKernelGGL(k_features_small<MODE>, dim3((unsigned)nb), dim3(kSmWaves * 64), 0, s, ii, evalcell, counters, fd, X, ax, d,
                           lower, upper, neg_gamma2, sp, dbg, ax2, idx_list, list_counter, list_cap, list_off);
        if (max_evals <= kSplitSmallList) return;
    }
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

}  // namespace haf

#ifdef HAF_PHASE_CLOCK
extern "C" int haf_phase_clock(unsigned long long *out, int n)
{
    unsigned long long h[16];
    if (hipMemcpyFromSymbol(h, HIP_SYMBOL(haf::g_phase_clock), sizeof(h)) != hipSuccess) return -1;
    for (int i = 0; i < n && i < 16; i++) out[i] = h[i];
    return 0;
}
#endif
