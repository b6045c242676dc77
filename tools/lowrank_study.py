#!/usr/bin/env python3
"""Offline (CPU: numpy + the oracle) study of a LOW-RANK operand for the screening pass.

The 302 HAF attributes are linear functionals of the 14x14 window heights (fv.cpp:141-199: sums of weighted region sums);
their span has rank 158 (Features.txt), so u.w_n over the HAF slots is a 158-term dot product in any basis of that span --
if the attribute were the exact linear functional.  It is not: the reference evaluates ((a-b)-c)+d in fp32 on integral-image
corners (rounding of the intermediate, large on a 512^2 grid), prints "%.4g" (4 digits) and scales.  nu = x' - x_lin is
that noise.  This tool measures |nu| against the fp16 rounding |u^ - u| the band already carries, the decision error that
dropping nu causes (actual, not bounded), and the first-order bounds in the plain and the centred-remainder form.

  python tools/lowrank_study.py [--grid 512] [--samples 800] [--trained] [--seeds 42,11]
Not a test; no GPU.
"""
import argparse
import os
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np  # noqa: E402

import models  # noqa: E402
from band_study import F, R, f16, sigma_upper, DATA  # noqa: E402
from oracle import oracle as O  # noqa: E402

LN2 = np.log(2.0)
U24 = 2.0 ** -24


def linear_map(o):
    """(324 x 225) matrix of the HAF functionals on the 15x15 integral-image window (row-major), zero rows for SHAF/phantom."""
    reg, w = o.feature_table()
    n = reg.shape[0]
    A = np.zeros((n, 225))
    for a in range(min(n, 302)):
        for k in range(3):                       # the 4th weight is never assigned (CHaarFeature.cpp:56-60)
            x1, x2, y1, y2 = reg[a, 4 * k:4 * k + 4]
            wk = float(np.float32(w[a, k]))
            if wk == 0 or x2 < x1 or y2 < y1 or (x2 == 0 and y2 == 0):
                continue
            for (r, c, s) in ((x2 + 1, y2 + 1, 1), (x1, y2 + 1, -1), (x2 + 1, y1, -1), (x1, y1, 1)):
                A[a, r * 15 + c] += s * wk
    return A


def windows(o, grid, n_samples, pcd=None, model=None):
    if pcd:
        import pcdio
        xyz = pcdio.load_pcd(os.path.join(DATA, pcd + ".pcd"))
        cfg = O.make_cfg(n_rolls=1)
        res = o.run(xyz, cfg, O.make_input(length_x=56, length_y=56, center=(0.13, 0.25, 0)), debug=True)
    else:
        xyz = models.synthetic_cloud(grid=grid, k=2, seed=0)
        cfg = O.make_cfg(H=grid, W=grid, n_rolls=1, roll_step_deg=5)
        res = o.run(xyz, cfg, O.make_input(length_x=grid, length_y=grid), debug=True)
    ii = res["integral"][0]
    cells = np.argwhere(res["mask"][0] == 1)
    rng = np.random.RandomState(0)
    cells = cells[rng.choice(len(cells), min(n_samples, len(cells)), replace=False)]
    return [ii[i - 7:i + 8, j - 7:j + 8].copy() for i, j in cells], cells


def rows(o, wins, A, D):
    lower, upper, fmin, fmax, present = o.range_table()
    skip = np.zeros(325, np.uint8)
    skip[324] = 1
    X, XL = [], []
    for wdw in wins:
        f = o.feature_values(wdw)
        q = np.array([O.q4(v) for v in f])
        X.append(o.scale_row(q, D, skip))
        lin = A @ wdw.astype(np.float64).ravel()              # exact linear functionals (fp64 sums of fp32 corners: exact to 1e-16)
        xl = np.array(X[-1])
        for a in range(302):
            xl[a] = lower + (upper - lower) * (lin[a] - fmin[a + 1]) / (fmax[a + 1] - fmin[a + 1])
        XL.append(xl)
    return np.array(X), np.array(XL)


def study(name, model_path, X, XL):
    o = O.Oracle(F, R, model_path)
    m = o.model_arrays()
    gamma, coef, sv, rho = m["gamma"], m["coef"], m["sv"], m["rho"]
    c = np.sqrt(2 * gamma * np.log2(np.e))
    U, UL, V = X * c, XL * c, sv * c
    nu = U - UL
    Uh = f16(U)
    t = -0.5 * (V * V).sum(1)

    def dec_of(Ux):
        ax = 0.5 * (Ux * Ux).sum(1)
        K = np.exp2(Ux @ V.T + t[None, :] - ax[:, None])
        return K @ coef - rho, K @ np.abs(coef), K
    dec, S, K = dec_of(U)
    decL, _, _ = dec_of(UL)
    dech = np.exp2(Uh @ f16(V).T + t[None, :] - 0.5 * (U * U).sum(1)[:, None]) @ coef - rho
    nn, dn, un = np.linalg.norm(nu, axis=1), np.linalg.norm(Uh - U, axis=1), np.linalg.norm(U, axis=1)
    print("== %s: nSV %d gamma %.4g; rows %d" % (name, len(coef), gamma, len(X)))
    print("   |nu|_2 median %.3g max %.3g | fp16 rounding |u^-u|_2 median %.3g max %.3g | |u| median %.3g ;  |nu|/|u^-u| median %.2f q90 %.2f max %.2f" %
          (np.median(nn), nn.max(), np.median(dn), dn.max(), np.median(un), np.median(nn / dn), np.quantile(nn / dn, 0.9), (nn / dn).max()))
    e_nu, e_h = np.abs(decL - dec), np.abs(dech - dec)
    print("   |dec| median %.3g S median %.3g;  actual error: dropping nu median %.3g max %.3g | fp16 pass median %.3g max %.3g;  ratio of medians %.2f; wrong sign by nu %d, by fp16 %d" %
          (np.median(np.abs(dec)), np.median(S), np.median(e_nu), e_nu.max(), np.median(e_h), e_h.max(), np.median(e_nu) / np.median(e_h),
           int(np.sum((decL > 0) != (dec > 0))), int(np.sum((dech > 0) != (dec > 0)))))
    # plain form, spectral bound of the first-order term: |x|.sigma(V^)|w|_2, |w|_2 measured
    w2 = np.sqrt(((K * coef[None, :]) ** 2).sum(1))
    sV, sVh = sigma_upper(f16(V)), sigma_upper(f16(V)[:, :302])
    b_fp16, b_nu = LN2 * dn * sV * w2, LN2 * nn * sVh * w2
    print("   plain first-order spectral bound: fp16 term median %.3g | nu term median %.3g  (x %.2f)" % (np.median(b_fp16), np.median(b_nu), np.median(b_nu / b_fp16)))
    # centred-remainder form
    wgt = np.abs(coef) * np.exp2(t)
    mu = (wgt[:, None] * V).sum(0) / wgt.sum()
    P, Q = U - mu, V - mu
    b = coef * np.exp2(-0.5 * (Q * Q).sum(1))
    Qh = f16(Q)
    nN = sigma_upper((Q * b[:, None]).T @ Qh)
    nNh = sigma_upper(((Q * b[:, None]).T @ Q)[:, :302])
    pn, dP = np.linalg.norm(P, axis=1), np.linalg.norm(f16(P) - P, axis=1)
    A_ = np.exp2(-0.5 * (P * P).sum(1))
    q_fp16, q_nu = A_ * LN2 ** 2 * nN * pn * dP, A_ * LN2 ** 2 * nNh * pn * nn
    z = P @ Q.T
    Spsi = A_ * ((np.exp2(z) - 1 - z * LN2) @ np.abs(b))
    print("   centred-remainder: |N| %.4g |N_haf| %.4g; first-order: fp16 term median %.3g | nu term median %.3g (x %.2f); S_psi median %.3g; |dec| q10 %.3g q50 %.3g" %
          (nN, nNh, np.median(q_fp16), np.median(q_nu), np.median(q_nu / q_fp16), np.median(Spsi), np.quantile(np.abs(dec), 0.1), np.median(np.abs(dec))))
    qmax = np.linalg.norm(Q[:, :302], axis=1).max()
    print("   centred-remainder cubic-type term ln2 |nu| max|q_haf| S_psi: median %.3g" % np.median(LN2 * nn * qmax * Spsi))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--grid", type=int, default=512)
    ap.add_argument("--samples", type=int, default=600)
    ap.add_argument("--seeds", default="42")
    ap.add_argument("--trained", action="store_true")
    ap.add_argument("--pcd", default="table1_mult_obj_rcs_1428580506606673")
    a = ap.parse_args()
    tmp = tempfile.mkdtemp()
    tiny = os.path.join(tmp, "tiny.model")
    models.write_random_model(tiny, 8, seed=1, balanced=True)
    o = O.Oracle(F, R, tiny)
    A = linear_map(o)
    s = np.linalg.svd(A[:302], compute_uv=False)
    print("rank of the HAF functionals on the window: %d (sigma %.3g .. %.3g)" % (np.sum(s > 1e-9 * s[0]), s[0], s[np.sum(s > 1e-9 * s[0]) - 1]))
    sets = []
    wins, _ = windows(o, a.grid, a.samples)
    sets.append(("synthetic %d^2" % a.grid, rows(o, wins, A, 323)))
    for p in [x for x in a.pcd.split(",") if x]:
        wins, _ = windows(o, 56, a.samples, p)
        sets.append((p, rows(o, wins, A, 323)))
    mods = []
    for sd in [int(x) for x in a.seeds.split(",") if x]:
        pth = os.path.join(tmp, "r%d.model" % sd)
        models.write_random_model(pth, 4096, seed=sd, balanced=True, rho=0.01, gamma=1.0 / 323)
        mods.append(("random seed %d" % sd, pth))
    if a.trained:
        pth = os.path.join(tmp, "trained.model")
        models.unpack_trained_model(os.path.join(ROOT, "tests", "golden", "trained.model.npz"), pth)
        mods.append(("trained", pth))
    for sname, (X, XL) in sets:
        for mname, pth in mods:
            study("%s / %s" % (sname, mname), pth, X, XL)


if __name__ == "__main__":
    main()
