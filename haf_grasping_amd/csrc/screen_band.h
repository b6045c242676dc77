// screen_band.h -- the guard band of the LOW-RANK centred-remainder form of the screening pass (kernels.h: kLrK, LrBand).
// Evaluated in the tail of k_svm_screen's low-rank instantiation, one evaluation per lane, from the raw sums the feature kernel
// (k_features_serial, ScreenParams::lr) and the projection (k_project) left behind:
//   raw = {su2, sd2, sx2, L, nu2, sdy2, 0, 0}: |fl32(p')|^2 and |p^ - fl32(p')|^2 over the slots, |p'|^2 over all attributes, the
//   linear term, an upper bound of |p' - p_lin|^2 (p_lin: the exactly linear part of the HAF slots, which lies in range(B)), and
//   |y^ - y32|^2 (the fp16 rounding of the projected operand, exact differences summed in fp32).
// With p the TRUE centred operand: p = B y* + p_perp, |p_perp| <= |p - p_lin| =: nun; y_e = B^'p; the sweep computes
// z^_n = y^.q~^_n + a_n where z_n = p.q_n = y_e.q~_n + p_perp.r_n (q~_n = (B'B^)^-1 B'q_n, r_n = (I - BB')q_n - (B^ - B)q~_n), so
//   eps_n = z^_n - z_n = dy.q~^_n + y_e.dq~_n + a_n - p_perp.r_n,
//   dy = y^ - y_e = (y^ - y32) + (y32 - B^'p^) + B^'(p^ - p):  |dy| <= sqrt(sdy2) + acc10 sigma(|B^|) |p^| + sigma(B^) |p^ - p|.
// The rest is screen_finish_cr (features.hip) term by term, with four error sources instead of three:
//   quadratic part, first order:  ln2^2 [p'N1 dy + p'(M1 B^')p + sum b_n z_n a_n + p'N2 p_perp],  N1 = Q'bQ~^, M1 = Q'b dQ~, N2 = Q'bR (signed)
//   second order:                 ln2^2/2 sum|b_n| eps_n^2 <= 2 ln2^2 (|H~abs||dy|^2 + |D~abs||y_e|^2 + acc6^2 |y^|^2 C~qq + |Rabs| nun^2)
//   psi3 and the exp / polynomial terms: through eps = sup|eps_n| and zmax as before.
#pragma once
#include "device_common.h"
#include "feature_device.h"

namespace haf {

// returns false when the evaluation must never be trusted (outside the range the bounds were derived for)
__device__ __forceinline__ void lr_finish_band(const float *raw, const LrBand &lb, float &L, float &c_abs_out, float &k_psi_out, float &cm_out)
{
    constexpr double kF32Acc = 326.0 * 5.9604644775390625e-08 * 1.01;
    const double ln2 = 0.69314718056;
    const double su2 = (double)raw[0], sd2 = (double)raw[1], sx2 = (double)raw[2], lsum = (double)raw[3], nu2 = (double)raw[4], sdy2 = (double)raw[5];
    const double a_x = 0.5 * sx2;
    const double un_t = sqrt_upper(sx2 * (1.0 + kF32Acc));
    const double eta_t = kScreenEtaRel * (un_t + lb.mu_norm_t) + lb.eta_abs;
    const double un1 = sqrt_upper(su2 * (1.0 + kF32Acc));
    const double dn1 = sqrt_upper(sd2 * (1.0 + kF32Acc)) + 5.97e-8 * un1 + 1e-17;
    const double eta = kScreenEtaRel * (un1 + lb.mu_norm) + lb.eta_abs;
    const double un = un1 + eta, dn = dn1 + eta, ph = un1 + dn1;                               // |p|, |p^ - p|, |p^|
    const double D = (kF32Acc + 6.0e-8) * a_x + un_t * eta_t + 0.5 * eta_t * eta_t + 6.0e-7;
    const double nun = sqrt_upper(nu2 * (1.0 + kF32Acc)) + eta;                                // |p - p_lin| >= |p_perp|
    const double dyn = sqrt_upper(sdy2 * (1.0 + kF32Acc)) + lb.acc10 * lb.sigAbsB * ph + lb.sigB * dn;   // |y^ - y_e|
    const double yen = lb.sigB * un, yhn = yen + dyn;                                           // |y_e|, |y^|
    const double eps = dyn * lb.qmax + yen * lb.dqmax + lb.acc6 * yhn * lb.qmax + nun * lb.rmax;
    const double zmax = yhn * lb.qmax + eps;
    const double zf = floor(zmax);
    const double p2 = (zmax < 60.0) ? ldexp(1.0 + (zmax - zf), (int)zf) : (double)__builtin_inff();
    const double acc_sum = fmin(yhn * un * lb.Ca, lb.sQb * un * lb.sQtaa * yhn);
    const double quad1 = ln2 * ln2 * (lb.nN1 * un * dyn + lb.nM1 * un * un + lb.acc6 * acc_sum + lb.nN2 * un * nun);
    const double quad2 = 2.0 * ln2 * ln2 * (lb.nHabs * dyn * dyn + lb.nDabs * yen * yen + lb.acc6 * lb.acc6 * yhn * yhn * lb.Cqq + lb.nRabs * nun * nun);
    const double cub2 = ln2 * ln2 * (p2 - 1.0) * eps * eps * lb.Babs * 1.01;
    const double cL = ln2 * lb.gnorm * (eta + 330.0 * 5.97e-8 * un1) * 1.01 + 1.2e-7 * fabs(lsum);
    double c_abs = quad1 + quad2 + cub2 + cL;
    double k_psi = ln2 * eps * 1.01;
    if (lb.poly) {
        // the low-rank sweep's polynomial has DEGREE 4: psi(t) = e^t - 1 - t, P4 = t^2/2 + t^3/6 + t^4/24; the dropped tail is at most
        // |t|^5/120 / (1 - |t|/6) and psi(t) >= t^2/2 (1 - |t|/3) for |t| <= 1: the tail is at most (|t|^3/60) / ((1 - |t|/6)(1 - |t|/3)) of psi
        // (0.03 at |t| = 1, 3e-5 at 0.12: the trained model); the fp32 roundings per element -- z^2, the three constants b a_k (folded per
        // block), two Horner steps -- stay below the 10 u charged
        const double t = ln2 * zmax;
        k_psi += 1.01 * (t * t * t / 60.0) / ((1.0 - t / 6.0) * (1.0 - t / 3.0)) + 10.0 * 5.97e-8;
        if (!(t <= 1.0)) k_psi = (double)__builtin_inff();
    } else {
        const double uexp = 2.4e-7;
        k_psi += uexp;
        c_abs += uexp * (lb.Babs + ln2 * yhn * lb.Cq1) * 1.01;
    }
    const double infl = 1.0 + exp2m1_upper(D);
    L = (float)lsum;
    c_abs_out = (float)(c_abs * infl * lb.scale * (1.0 + 1e-6));
    k_psi_out = (float)(k_psi * infl * lb.scale * (1.0 + 1e-6));
    cm_out = (float)(exp2m1_upper(D) * lb.scale);
    if (!(D < 0.05) || !(a_x < 30.0) || !(zmax < 60.0) || !(c_abs == c_abs)) c_abs_out = __builtin_inff();
}

// The PLAIN epilogue on the projected operands: the sweep forms D^ = sum_n b_n 2^(z^_n) (dec + rho = A D); with w_n = b_n 2^(z_n),
// e_n = z^_n - z_n as above, D^ - D = ln2 sum w_n e_n + second order.  Term by term the plain form's band (features.hip: screen_finish):
//   bilinear part of e_n:  per SV  d_max S  |  through the spectral norms  e2 |w|_2,  e2 = |dy| sigma(Q~^) + |y_e| sigma(dQ~) + |p_perp| sigma(R),
//                          |w|_2^2 <= max|c_n| S / A (the tail's sqrt_cmax sqrt(S) after the common factor);
//   accumulation a_n and second order: per unit of S (gB);
//   centred estimate: w_n = b_n + b_n (2^(z_n) - 1) (reference operand = the centre, p = 0).  First-order error of the first part:
//     ln2 [dy.g~ + y_e.h~ + sum b_n a_n - p_perp.rho~],  g~ = Q~^'b, h~ = dQ~'b, rho~ = R'b.  What is KNOWN of it -- (p^ - fl32 p').(B^ g~) +
//     fl32(p').(B^ h~), the feature kernel's `cr` sum -- is subtracted; the rest is bounded: the rounding of y against g~, the projection's
//     accumulation, p' against p, the fp32 accumulation of the sum itself, |p_perp||rho~| (abs_c); sum b_n a_n stays with gB.
//     Second part: |sum b_n (2^(z_n) - 1) e_n| <= |b (2^z - 1)|_2 e2 <= ln2 2^zmax (sbq |y_e| + sbr |p_perp|) e2.
// Output as screen_finish: g = {gA, gB, gC, cm}, g2 = {corr, abs_c}.
__device__ __forceinline__ void lr_finish_band_plain(const float *raw, const LrBand &lb, float4 &g, float4 &g2)
{
    constexpr double kF32Acc = 326.0 * 5.9604644775390625e-08 * 1.01;
    const double ln2 = 0.69314718056;
    const double su2 = (double)raw[0], sd2 = (double)raw[1], sx2 = (double)raw[2], cr = (double)raw[3], nu2 = (double)raw[4], sdy2 = (double)raw[5];
    const double a_x = 0.5 * sx2;
    const double un_t = sqrt_upper(sx2 * (1.0 + kF32Acc));
    const double eta_t = kScreenEtaRel * (un_t + lb.mu_norm_t) + lb.eta_abs;
    const double un1 = sqrt_upper(su2 * (1.0 + kF32Acc));
    const double dn1 = sqrt_upper(sd2 * (1.0 + kF32Acc)) + 5.97e-8 * un1 + 1e-17;
    const double eta = kScreenEtaRel * (un1 + lb.mu_norm) + lb.eta_abs;
    const double un = un1 + eta, dn = dn1 + eta, ph = un1 + dn1;
    const double D = (kF32Acc + 6.0e-8) * a_x + un_t * eta_t + 0.5 * eta_t * eta_t + 6.0e-7;
    const double nun = sqrt_upper(nu2 * (1.0 + kF32Acc)) + eta;
    const double sdy = sqrt_upper(sdy2 * (1.0 + kF32Acc));
    const double dyn = sdy + lb.acc10 * lb.sigAbsB * ph + lb.sigB * dn;
    const double yen = lb.sigB * un, yhn = yen + dyn;
    const double d_max = dyn * lb.qmax + yen * lb.dqmax + nun * lb.rmax;
    const double acc = lb.acc6 * yhn * lb.qmax;
    const double e_max = d_max + acc;
    const double infl = 1.0 + exp2m1_upper(e_max + D);
    const double e2 = dyn * lb.sig_q + yen * lb.sig_dq + nun * lb.sig_r;
    const double gB = ln2 * acc + 0.6 * (ln2 * e_max) * (ln2 * e_max);
    g.x = (float)(ln2 * e2 * infl * lb.scale);
    g.y = (float)(gB * infl * lb.scale);
    g.z = (float)(ln2 * d_max * infl * lb.scale);
    g.w = (float)(exp2m1_upper(D) * lb.scale);
    const bool bad = !(e_max + D < 0.05) || !(a_x < 30.0);
    if (bad) g.y = __builtin_inff();
    const double zmax = yhn * lb.qmax + e_max;
    const double zf = floor(zmax);
    const double p2 = (zmax < 60.0) ? ldexp(1.0 + (zmax - zf), (int)zf) : (double)__builtin_inff();
    const double dev = ln2 * p2 * (lb.sbq * yen + lb.sbr * nun);
    const double cerr = ln2 * (sdy * lb.gt_norm + lb.acc10 * ph * lb.gabsB + (eta + 6.0e-8 * un1) * (lb.bg_norm + lb.bh_norm) +
                               4.2e-5 * (dn1 * lb.bg_norm + un1 * lb.bh_norm) + nun * lb.rho_norm);
    g2.x = (float)(ln2 * cr);
    g2.y = (float)((e2 * dev * infl + cerr) * lb.scale);
    if (bad || !(zmax < 60.0) || !(g2.y == g2.y)) g2.y = __builtin_inff();
    g2.z = 0.0f; g2.w = 0.0f;
}

}  // namespace haf
