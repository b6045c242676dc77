#!/usr/bin/env python3
"""Offline (CPU: numpy + the oracle) study of the screening band's terms, per model seed.

For a sample of evaluations of the C5 cloud it emulates the screening pass (fp16 operands, exact sums) and prints, per model:
the share of each band term (bilinear term through |w|_2, S-proportional terms, common factor), the share of evaluations
inside the band, and the same for the CENTRED form of the bilinear term: w_n = c_n K_n is split into the part
c_n kappa_n that does not depend on the evaluation (kappa_n = 2^(t_n), the kernel value at u.v_n = 0) -- whose first-order
error is a dot product with two model constants G = V^'(c kappa), Hd = dV'(c kappa) and is CORRECTED instead of bounded --
and the remainder c_n (k_n - kappa_n), which is bounded through the spectral norm as before.

  python tools/band_study.py [--nsv 4096] [--seeds 1234,7,11,23,42] [--hard] [--samples 1500]
Not a test; no GPU.
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402

import models  # noqa: E402
from oracle import oracle as O  # noqa: E402

DATA = os.path.join(ROOT, "tests", "golden", "data")
F, R = os.path.join(DATA, "Features.txt"), os.path.join(DATA, "range21062012_allfeatures")
U24 = 2.0 ** -24


def f16(a):
    return a.astype(np.float32).astype(np.float16).astype(np.float64)


def sigma_upper(M, squarings=7):
    G = M.T @ M
    scale, p = 0.0, 1.0
    for _ in range(squarings):
        t = np.trace(G)
        G = G / t
        scale += np.log(t) / p
        G = G @ G
        p *= 2.0
    return np.sqrt(np.exp(scale + np.log(np.trace(G)) / p))


def sample_rows(o, D, grid, n_samples, pcd=None):
    if pcd:
        import pcdio
        xyz = pcdio.load_pcd(os.path.join(DATA, pcd + ".pcd"))
        cfg = O.make_cfg(n_rolls=1)
        res = o.run(xyz, cfg, O.make_input(length_x=56, length_y=56), debug=True)
    else:
        xyz = models.synthetic_cloud(grid=grid, k=2, seed=0)
        cfg = O.make_cfg(H=grid, W=grid, n_rolls=1, roll_step_deg=5)
        res = o.run(xyz, cfg, O.make_input(length_x=grid, length_y=grid), debug=True)
    ii = res["integral"][0]
    cells = np.argwhere(res["mask"][0] == 1)
    rng = np.random.RandomState(0)
    cells = cells[rng.choice(len(cells), min(n_samples, len(cells)), replace=False)]
    skip = np.zeros(325, np.uint8)
    skip[324] = 1
    return np.stack([o.scale_row(np.array([O.q4(v) for v in o.feature_values(ii[i - 7:i + 8, j - 7:j + 8])]), D, skip)
                     for i, j in cells])


def study(name, path, X=None, grid=96, n_samples=1500, pcd=None):
    o = O.Oracle(F, R, path)
    m = o.model_arrays()
    if X is None:
        X = sample_rows(o, m["D"], grid, n_samples, pcd)
    gamma, coef, sv, rho = m["gamma"], m["coef"], m["sv"], m["rho"]
    n_sv = len(coef)
    c = np.sqrt(2 * gamma * np.log2(np.e))
    U, V = X * c, sv * c
    Uh, Vh = f16(U), f16(V)
    dU, dV = Uh - U, Vh - V
    t = -0.5 * (V * V).sum(1)
    ax = 0.5 * (U * U).sum(1)
    z, zh = U @ V.T, Uh @ Vh.T
    K = np.exp2(z + t[None, :] - ax[:, None])
    Kh = np.exp2(zh + t[None, :] - ax[:, None])
    dec, dech = K @ coef - rho, Kh @ coef - rho
    S = K @ np.abs(coef)
    err = np.abs(dech - dec)
    un, dn = np.linalg.norm(U, axis=1), np.linalg.norm(dU, axis=1)
    vmax, dvmax = np.linalg.norm(Vh, axis=1).max(), np.linalg.norm(dV, axis=1).max()
    sV, sdV = sigma_upper(Vh), sigma_upper(dV)
    cmax = np.abs(coef).max()
    ln2 = np.log(2)
    tiles = (n_sv + 31) // 32 + 1
    guard_acc0 = (16 + 1 + (tiles / 8 + 1) + 2 + 4 + 2 + 6 + 2) * U24 * 1.04
    acc = 2.0 ** -18 * (un * vmax + np.abs(t).max())
    d_max = dn * vmax + (un + dn) * dvmax
    gA = ln2 * (dn * sV + (un + dn) * sdV)
    gB = ln2 * acc + 0.6 * (ln2 * d_max) ** 2
    gC = ln2 * d_max
    w2_meas = np.sqrt(((K * coef[None, :]) ** 2).sum(1))
    w2_bound = np.sqrt(cmax * S)
    sterm = (guard_acc0 + gB) * S
    band_plain = np.minimum(gA * w2_bound, gC * S) + sterm
    band_sumsq = np.minimum(gA * w2_meas, gC * S) + sterm

    # ---- centred form ----
    sc = np.exp2(-ax)
    kap = np.exp2(t)                                     # raw-space kernel value at u.v = 0
    ck = coef * kap
    G, Hd = Vh.T @ ck, dV.T @ ck                        # model constants (fp64 at model load)
    corr = ln2 * sc * (dU @ G + U @ Hd)                  # first-order error of the evaluation-independent part: corrected
    dech_c = dech - corr
    err_c = np.abs(dech_c - dec)
    kraw = np.exp2(z + t[None, :])
    dev_meas = np.sqrt((((kraw - kap[None, :]) * coef[None, :]) ** 2).sum(1)) * sc     # |c (k - kappa)|_2, measured
    # analytic: |c kappa (2^z - 1)|_2 <= ln2 2^zmax |D V^ u^|_2 <= ln2 2^zmax sigma(D V^) |u^|  (+ the tiny e_n part)
    sDV = sigma_upper(Vh * ck[:, None])
    uh_n = np.linalg.norm(Uh, axis=1)
    zmax = uh_n * vmax + d_max
    dev_bound = ln2 * np.exp2(zmax) * (sDV * uh_n + np.abs(ck).max() * (dn * sV + un * sdV)) * sc
    dev_exactq = ln2 * np.exp2(zmax) * np.linalg.norm((Uh @ Vh.T) * ck[None, :], axis=1) * sc    # |D V^ u^| itself (a quadratic form)
    band_c_meas = np.minimum(gA * np.minimum(dev_meas, w2_meas), gC * S) + sterm
    band_c_bound = np.minimum(gA * np.minimum(dev_bound, w2_bound), gC * S) + sterm
    band_c_quad = np.minimum(gA * np.minimum(dev_exactq, w2_bound), gC * S) + sterm

    def share(b):
        return float(np.mean(np.abs(dec) <= b))

    print("== %s: nSV=%d gamma=%.4g rho=%.3g  samples=%d  positive %.1f %%  median|dec| %.3g  S %.4g  |u| %.3f  zmax %.2f" %
          (name, n_sv, gamma, rho, len(X), 100 * np.mean(dec > 0), np.median(np.abs(dec)), np.median(S), np.median(un), np.median(zmax)))
    print("   sigma(V^) %.3f  sigma(dV) %.2e  sigma(DV^) %.3f  max c*kappa %.3f   |w|2: measured %.2f, bound %.2f;  |c(k-kappa)|2: measured %.2f, "
          "bound %.2f, quad %.2f" % (sV, sdV, sDV, np.abs(ck).max(), np.median(w2_meas), np.median(w2_bound), np.median(dev_meas),
                                    np.median(dev_bound), np.median(dev_exactq)))
    print("   band terms (median): bilinear plain %.3g | sumsq %.3g | centred-bound %.3g | centred-quad %.3g | centred-measured %.3g ;  "
          "S-terms %.3g (acc0 %.3g, gB %.3g)" %
          (np.median(np.minimum(gA * w2_bound, gC * S)), np.median(np.minimum(gA * w2_meas, gC * S)),
           np.median(gA * np.minimum(dev_bound, w2_bound)), np.median(gA * np.minimum(dev_exactq, w2_bound)),
           np.median(gA * np.minimum(dev_meas, w2_meas)), np.median(sterm), np.median(guard_acc0 * S), np.median(gB * S)))
    print("   inside band: plain %.4f | sumsq %.4f | centred-bound %.4f | centred-quad %.4f | centred-measured %.4f" %
          (share(band_plain), share(band_sumsq), share(band_c_bound), share(band_c_quad), share(band_c_meas)))
    print("   actual error: uncorrected median %.3g max %.3g (max err/band plain %.3f);  corrected median %.3g max %.3g (max err/band: bound %.3f, "
          "measured %.3f)" % (np.median(err), err.max(), (err / band_plain).max(), np.median(err_c), err_c.max(),
                              (err_c / band_c_bound).max(), (err_c / band_c_meas).max()))
    # ---- centring about a reference operand ubar: kappa_n = 2^(t_n + ubar.v^_n) ----
    for label, ub in (("SV centroid", Vh.mean(0)), ("|c|kappa-weighted SV centroid", (np.abs(ck)[:, None] * Vh).sum(0) / np.abs(ck).sum()),
                      ("data mean (not available at model load)", Uh.mean(0))):
        mn = Vh @ ub
        kap2 = np.exp2(t + mn)
        ck2 = coef * kap2
        G2, H2 = Vh.T @ ck2, dV.T @ ck2
        corr2 = ln2 * sc * (dU @ G2 + U @ H2)
        err2 = np.abs(dech - corr2 - dec)
        du_n = np.linalg.norm(Uh - ub[None, :], axis=1)
        zm2 = du_n * vmax + d_max
        sDV2 = sigma_upper(Vh * ck2[:, None])
        dev2 = ln2 * np.exp2(zm2) * (sDV2 * du_n + np.abs(ck2).max() * (dn * sV + un * sdV)) * sc
        dev2m = np.sqrt((((kraw - kap2[None, :]) * coef[None, :]) ** 2).sum(1)) * sc
        b2 = np.minimum(gA * np.minimum(dev2, w2_bound), gC * S) + sterm
        b2m = np.minimum(gA * np.minimum(dev2m, w2_meas), gC * S) + sterm
        print("   ubar = %-42s |ubar| %.3f  |u^-ubar| median %.3f  zmax' %.2f  sigma(D'V^) %.3g  |G| %.3g |Hd| %.3g  |c(k-kappa)|2: measured %.4g bound %.4g"
              "  inside band: bound %.4f measured %.4f   corrected err median %.3g max %.3g" %
              (label, np.linalg.norm(ub), np.median(du_n), np.median(zm2), sDV2, np.linalg.norm(G2), np.linalg.norm(H2),
               np.median(dev2m), np.median(dev2), share(b2), share(b2m), np.median(err2), err2.max()))
    return X


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--nsv", type=int, default=4096)
    ap.add_argument("--seeds", default="1234,7,11,23,42")
    ap.add_argument("--hard", action="store_true")
    ap.add_argument("--surrogate", action="store_true")
    ap.add_argument("--samples", type=int, default=1500)
    ap.add_argument("--grid", type=int, default=96)
    ap.add_argument("--pcd", default=None, help="rows from tests/golden/data/<name>.pcd (56x56, roll 0) instead of the synthetic cloud")
    a = ap.parse_args()
    X = None
    for seed in [int(s) for s in a.seeds.split(",") if s]:
        path = "/tmp/band_study_rand%d_%d.model" % (a.nsv, seed)
        models.write_random_model(path, a.nsv, seed=seed, balanced=True)
        X = study("seed %d" % seed, path, X, a.grid, a.samples, a.pcd)
    if a.hard:
        path = "/tmp/band_study_hard.model"
        models.write_replicated_model(path, os.path.join(ROOT, "tests", "golden", "surrogate.model"), copies=24, jitter=0.01, seed=5)
        X = study("hard (surrogate x 24)", path, X, a.grid, a.samples, a.pcd)
    if a.surrogate:
        X = study("surrogate", os.path.join(ROOT, "tests", "golden", "surrogate.model"), X, a.grid, a.samples, a.pcd)


if __name__ == "__main__":
    main()
