#!/usr/bin/env python3
"""Offline (CPU: numpy + the oracle) study of the CENTRED-REMAINDER form of the RBF decision.

    dec + rho = sum_n c_n exp(-gamma |x - s_n|^2)
              = A(x) [ B0 + ln2 * p.g + sum_n b_n psi(z_n) ],      p = c (x - m), q_n = c (s_n - m), c = sqrt(2 gamma log2 e)
      A = 2^(-|p|^2/2),  b_n = c_n 2^(-|q_n|^2/2),  z_n = p.q_n,  psi(z) = 2^z - 1 - z ln2  (>= 0),
      B0 = sum b_n, g = sum b_n q_n   (model constants, fp64 at load)

for ANY centre m (translation invariance of the kernel).  B0 and the linear term are exact per evaluation (one fp64 dot
product in the feature kernel); only the remainder goes through the fp16 matrix core, and its cancellation scale
S_psi = A sum |b_n| psi(z_n) is what the fp32 roundings and the operand roundings are relative to -- against
S = sum |c_n| K_n for the plain form.  For a trained model with a large C and a small gamma S is 1e6..1e7 times |dec|
(nothing an fp32 sum can decide) while S_psi is a few times |dec|.

Prints, per model and input set: |dec| / S and |dec| / S_psi quantiles, the range of z, the error of an emulated single
fp16 pass in both forms (operands rounded to fp16, sums exact) and the share of evaluations a first-order worst-case band
would leave undecided.

  python tools/centre_study.py --model /tmp/haf_trained/trained.model [--pcd pcd2,table1...] [--grid 96] [--samples 1500]
Not a test; no GPU.
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np  # noqa: E402

from band_study import F, R, f16, sample_rows, sigma_upper  # noqa: E402
from oracle import oracle as O  # noqa: E402

U24 = 2.0 ** -24
LN2 = np.log(2.0)


def psi(z):
    return np.where(np.abs(z) < 1e-3, (LN2 * z) ** 2 / 2 * (1 + LN2 * z / 3), np.exp2(z) - 1 - z * LN2)


def quant(a, qs=(0.01, 0.1, 0.5, 0.9)):
    return " ".join("%.3g" % v for v in np.quantile(a, qs))


def study(name, model_path, X, centre="wmean"):
    o = O.Oracle(F, R, model_path)
    m = o.model_arrays()
    gamma, coef, sv, rho = m["gamma"], m["coef"], m["sv"], m["rho"]
    c = np.sqrt(2 * gamma * np.log2(np.e))
    # ---- plain form (round 3) ----
    U, V = X * c, sv * c
    t = -0.5 * (V * V).sum(1)
    ax = 0.5 * (U * U).sum(1)
    z0 = U @ V.T
    K = np.exp2(z0 + t[None, :] - ax[:, None])
    dec = K @ coef - rho
    S = K @ np.abs(coef)
    Uh, Vh = f16(U), f16(V)
    dech = np.exp2(Uh @ Vh.T + t[None, :] - ax[:, None]) @ coef - rho
    # ---- centred-remainder form ----
    if centre == "wmean":
        wgt = np.abs(coef) * np.exp2(t)
        mu = (wgt[:, None] * V).sum(0) / wgt.sum()
    elif centre == "mean":
        mu = V.mean(0)
    else:
        mu = np.zeros(V.shape[1])
    P, Q = U - mu, V - mu
    tq = -0.5 * (Q * Q).sum(1)
    A = np.exp2(-0.5 * (P * P).sum(1))
    b = coef * np.exp2(tq)
    B0 = b.sum()
    g = Q.T @ b
    z = P @ Q.T
    Rm = psi(z) @ b
    dec_c = A * (B0 + LN2 * (P @ g) + Rm) - rho
    Spsi = A * (psi(z) @ np.abs(b))
    Ph, Qh = f16(P), f16(Q)
    zh = Ph @ Qh.T
    dec_ch = A * (B0 + LN2 * (P @ g) + psi(zh) @ b) - rho
    # first-order worst-case band of the operand roundings: sum |b_n| |psi'(z_n)| |dz_n|, |dz_n| <= |p^-p||q^_n| + |p||q^_n-q_n|
    dP, dQ = np.linalg.norm(Ph - P, axis=1), np.linalg.norm(Qh - Q, axis=1)
    dz = dP[:, None] * np.linalg.norm(Qh, axis=1)[None, :] + np.linalg.norm(P, axis=1)[:, None] * dQ[None, :]
    band_op = A * ((LN2 * np.abs(np.exp2(z) - 1) * dz * 1.05) @ np.abs(b))
    # spectral form of the same: |sum b_n psi'(z_n) (dp.q^_n + p.dq_n)| <= (|dp| sigma(Q^) + |p| sigma(dQ)) |b psi'(z)|_2
    sQ, sdQ = sigma_upper(Qh), sigma_upper(Qh - Q)
    w2 = A * np.sqrt((((LN2 * (np.exp2(z) - 1)) * b[None, :]) ** 2).sum(1))
    band_sp = (dP * sQ + np.linalg.norm(P, axis=1) * sdQ) * w2 * 1.05
    band_acc = 60 * U24 * Spsi                      # fp32 sums + psi evaluation, ~60 roundings of the scale
    band = np.minimum(band_op, band_sp) + band_acc
    # the per-SV form with ONE bound for all dz_n and sum|b||2^z - 1| measured by the kernel (a second accumulator) ...
    dzmax = dP * np.linalg.norm(Qh, axis=1).max() + np.linalg.norm(P, axis=1) * dQ.max()
    lin_meas = A * (np.abs(np.exp2(z) - 1) @ np.abs(b))
    band_lin = LN2 * dzmax * lin_meas * 1.05
    # ... or bounded through S_psi alone: sum|b||e-1| <= sqrt(sum|b|) sqrt(sum|b|(e-1)^2), (e-1)^2 <= 2 max(1, 2^zmax) psi(z)
    zmx = np.maximum(z.max(1), 0.0)
    band_cs = LN2 * dzmax * np.sqrt(A * np.abs(b).sum() * 2.0 * np.exp2(zmx) * Spsi) * 1.05
    # translation alone (operands p, q; plain coefficient sum): the operand term relative to S, the fp32 sum relative to S
    band_tr = LN2 * dzmax * S * 1.05 + 60 * U24 * S
    # exp-based psi: v_exp_f32 is off by an ulp of 2^z, i.e. 2 u S on top of the S_psi-relative roundings
    band_acc_exp = band_acc + 2 * U24 * S
    # IMPLEMENTABLE spectral band with no extra accumulator: |b psi'(z)|_2 <= ln2 2^zmax |diag(b) Q^ p^|_2 <= ln2 2^zmax sigma(diag(b) Q^) |p^|,
    # |eps|_2 <= sigma(Q^) dP + sigma(dQ) |P| + acc_rel |p^| |Q^|_F  (matrix-core accumulation, 82 u per unit of |p^||q^_n|)
    sbQ = sigma_upper(Qh * b[:, None])
    pn, phn = np.linalg.norm(P, axis=1), np.linalg.norm(Ph, axis=1)
    zcs = phn * np.linalg.norm(Qh, axis=1).max()
    eps2 = sQ * dP + sdQ * pn + 82 * U24 * phn * np.linalg.norm(Qh)
    band_an = A * LN2 * LN2 * np.exp2(zcs) * sbQ * phn * eps2 * 1.05
    tau = (LN2 * zcs) ** 4 / 360.0                  # truncation of psi(t) = t^2/2 (1 + t/3 + t^2/12 + t^3/60) relative to psi
    band_an_tot = band_an + (60 * U24 + tau) * Spsi
    # ---- the band that is implemented: psi(z) = (ln2 z)^2/2 + psi3(z).  The QUADRATIC part's first-order error is
    # p'N dp + p'M p with the SIGNED model matrices N = Q'BQ^, M = Q'B dQ (the two classes cancel in them: |N| << sigma(Q)^2 max|b|),
    # bounded through their spectral norms; the rest (psi3' = ln2 psi >= 0) costs ln2 eps_max S_psi; accumulation in the matrix core
    # through C_a = sum|b||q^||q|; second order through the unsigned matrices ----
    Bm = b[:, None]
    nN = sigma_upper(((Q * Bm).T @ Qh))
    Ms = (Q * Bm).T @ (Qh - Q)
    nM = sigma_upper(0.5 * (Ms + Ms.T))
    qn, qhn, dqn = np.linalg.norm(Q, axis=1), np.linalg.norm(Qh, axis=1), np.linalg.norm(Qh - Q, axis=1)
    Ca, Cq1, Babs = (np.abs(b) * qhn * qn).sum(), (np.abs(b) * qhn).sum(), np.abs(b).sum()
    nHabs, nDabs = sigma_upper(Qh * np.sqrt(np.abs(Bm))) ** 2, sigma_upper((Qh - Q) * np.sqrt(np.abs(Bm))) ** 2
    accr = 82 * U24
    eps_max = dP * qhn.max() + pn * dqn.max() + accr * phn * qhn.max()
    Spsi_raw = psi(zh) @ np.abs(b)
    quad1 = LN2 ** 2 * (nN * pn * dP + nM * pn ** 2 + accr * phn * pn * Ca)
    quad2 = 0.5 * LN2 ** 2 * 3 * (nHabs * dP ** 2 + nDabs * pn ** 2 + accr ** 2 * phn ** 2 * (np.abs(b) * qhn ** 2).sum())
    cubic = LN2 * eps_max * (Spsi_raw + LN2 * (np.exp2(zcs) - 1) * eps_max * Babs) * 1.01
    sums = 60 * U24 * Spsi_raw
    uexp = 1.2e-7 * (Spsi_raw + Babs + LN2 * phn * Cq1)            # exp-based psi: v_exp_f32 and the constant ln2, an ulp of sum|b|2^z
    band_new_poly = A * (quad1 + quad2 + cubic + sums + tau * Spsi_raw)
    band_new_exp = A * (quad1 + quad2 + cubic + sums + uexp)
    # plain-form equivalents
    dU, dV = np.linalg.norm(Uh - U, axis=1), np.linalg.norm(Vh - V, axis=1)
    band0 = LN2 * (dU * np.linalg.norm(Vh, axis=1).max() + np.linalg.norm(U, axis=1) * dV.max()) * S + 60 * U24 * S
    print("== %s (%s): nSV %d gamma %.4g rho %.4g max|coef| %.4g  rows %d  positive %.1f %%" %
          (name, centre, len(coef), gamma, rho, np.abs(coef).max(), len(X), 100 * np.mean(dec > 0)))
    print("   check: max |dec_centred - dec| %.3g (|dec| median %.3g)" % (np.abs(dec_c - dec).max(), np.median(np.abs(dec))))
    print("   |dec|/S     q01 q10 q50 q90: %s    S median %.4g" % (quant(np.abs(dec) / S), np.median(S)))
    print("   |dec|/S_psi q01 q10 q50 q90: %s    S_psi median %.4g   (S/S_psi median %.3g)" %
          (quant(np.abs(dec) / Spsi), np.median(Spsi), np.median(S / Spsi)))
    print("   z = p.q: min %.3g max %.3g  rms %.3g   |p| median %.3g  |q| median %.3g  (plain: |u| %.3g |v| %.3g, u.v rms %.3g)" %
          (z.min(), z.max(), np.sqrt((z * z).mean()), np.median(np.linalg.norm(P, axis=1)), np.median(np.linalg.norm(Q, axis=1)),
           np.median(np.linalg.norm(U, axis=1)), np.median(np.linalg.norm(V, axis=1)), np.sqrt((z0 * z0).mean())))
    e0, e1 = np.abs(dech - dec), np.abs(dec_ch - dec)
    print("   one fp16 pass, sums exact: plain form error median %.3g max %.3g | centred-remainder median %.3g max %.3g" %
          (np.median(e0), e0.max(), np.median(e1), e1.max()))
    print("   |error|/|dec|: plain q50 %.3g q90 %.3g | centred q50 %.3g q90 %.3g ;  wrong sign: plain %d, centred %d of %d" %
          (np.median(e0 / np.abs(dec)), np.quantile(e0 / np.abs(dec), 0.9), np.median(e1 / np.abs(dec)), np.quantile(e1 / np.abs(dec), 0.9),
           int(np.sum((dech > 0) != (dec > 0))), int(np.sum((dec_ch > 0) != (dec > 0))), len(dec)))
    print("   undecided by a worst-case band: plain form %.4f | centred: per-SV %.4f spectral %.4f min+acc %.4f   (max err/band centred %.3f)" %
          (np.mean(np.abs(dec) <= band0), np.mean(np.abs(dec) <= band_op + band_acc), np.mean(np.abs(dec) <= band_sp + band_acc),
           np.mean(np.abs(dec) <= band), (e1 / band).max()))
    print("   band terms (median): operand per-SV %.3g | spectral %.3g | acc %.3g" % (np.median(band_op), np.median(band_sp), np.median(band_acc)))
    print("   analytic spectral band (no accumulator): first-order median %.3g, poly truncation+acc median %.3g, zmax(CS) median %.3g max %.3g;  undecided %.4f  (max err/band %.3f); sigma(Q^) %.3g sigma(dQ) %.3g sigma(bQ^) %.4g |Q^|_F %.3g" %
          (np.median(band_an), np.median((60 * U24 + tau) * Spsi), np.median(zcs), zcs.max(), np.mean(np.abs(dec) <= band_an_tot), (e1 / band_an_tot).max(), sQ, sdQ, sbQ, np.linalg.norm(Qh)))
    print("   NEW band: |N| %.4g |M_s| %.4g C_a %.4g B_abs %.4g |H_abs| %.4g;  terms (median): quad1 %.3g (N %.3g M %.3g acc %.3g) quad2 %.3g cubic %.3g sums %.3g uexp %.3g trunc %.3g" %
          (nN, nM, Ca, Babs, nHabs, np.median(A * quad1), np.median(A * LN2 ** 2 * nN * pn * dP), np.median(A * LN2 ** 2 * nM * pn ** 2),
           np.median(A * LN2 ** 2 * accr * phn * pn * Ca), np.median(A * quad2), np.median(A * cubic), np.median(A * sums), np.median(A * uexp), np.median(A * tau * Spsi_raw)))
    print("   NEW band undecided: poly variant %.4f (max err/band %.3f) | exp variant %.4f (max err/band %.3f)" %
          (np.mean(np.abs(dec) <= band_new_poly), (e1 / band_new_poly).max(), np.mean(np.abs(dec) <= band_new_exp), (e1 / band_new_exp).max()))
    print("   undecided: translation only %.4f | one dz bound x measured sum|b||e-1| %.4f | the same through sqrt(S_psi) %.4f | per-SV with exp-based psi (+2uS) %.4f" %
          (np.mean(np.abs(dec) <= band_tr), np.mean(np.abs(dec) <= band_lin + band_acc), np.mean(np.abs(dec) <= band_cs + band_acc),
           np.mean(np.abs(dec) <= band_lin + band_acc_exp)))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", required=True)
    ap.add_argument("--pcd", default="pcd2,table1_mult_obj_rcs_1428580506606673")
    ap.add_argument("--grid", type=int, default=96)
    ap.add_argument("--samples", type=int, default=1200)
    ap.add_argument("--centres", default="wmean,mean,none")
    a = ap.parse_args()
    o = O.Oracle(F, R, a.model)
    D = o.model_arrays()["D"]
    sets = [("synthetic %d^2" % a.grid, sample_rows(o, D, a.grid, a.samples))]
    for p in [s for s in a.pcd.split(",") if s]:
        sets.append((p, sample_rows(o, D, a.grid, a.samples, p)))
    for name, X in sets:
        for ctr in a.centres.split(","):
            study(name, a.model, X, ctr)


if __name__ == "__main__":
    main()
