#!/usr/bin/env python3
"""Offline (CPU, numpy + oracle) study of the screening pass's guard band on the bench model: how many evaluations fall
inside (a) the per-SV Cauchy-Schwarz band and (b) the spectral-norm band, and how far the actual single-pass fp16 error
stays below either.  Not a test; run by hand."""
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


def f16(a):
    h = a.astype(np.float32).astype(np.float16).astype(np.float64)
    h[np.abs(h) < 2.0 ** -14] = 0.0
    return h


def lam_max_upper(G, squarings=8):
    """rigorous upper bound of the largest eigenvalue of a PSD matrix: (trace G^(2^j))^(1/2^j)"""
    scale = 0.0
    A = G.copy()
    p = 1.0
    for _ in range(squarings):
        t = np.trace(A)
        A = A / t
        scale += np.log(t) / p
        A = A @ A
        p *= 2.0
    return np.exp(scale + np.log(np.trace(A)) / p)


def main(nsv=4096, model=None, n_samples=1500, grid=96):
    path = model or "/tmp/analyze_rand%d.model" % nsv
    if not model:
        models.write_random_model(path, nsv, seed=1234, balanced=True)
    o = O.Oracle(F, R, path)
    m = o.model_arrays()
    xyz = models.synthetic_cloud(grid=grid, k=2, seed=0)
    cfg = O.make_cfg(H=grid, W=grid, n_rolls=1, roll_step_deg=5)
    res = o.run(xyz, cfg, O.make_input(length_x=grid, length_y=grid))
    ii = res["integral"][0]
    cells = np.argwhere(res["mask"][0] == 1)
    rng = np.random.RandomState(0)
    cells = cells[rng.choice(len(cells), min(n_samples, len(cells)), replace=False)]
    skip = np.zeros(325, np.uint8)
    skip[324] = 1
    X = np.stack([o.scale_row(np.array([O.q4(v) for v in o.feature_values(ii[i - 7:i + 8, j - 7:j + 8])]), m["D"], skip)
                  for i, j in cells])
    gamma, coef, sv = m["gamma"], m["coef"], m["sv"]
    g2 = gamma * np.log2(np.e)
    c = np.sqrt(2 * g2)
    U, V = X * c, sv * c
    Uh, Vh = f16(U), f16(V)
    dU, dV = Uh - U, Vh - V
    arg = U @ V.T - 0.5 * (U * U).sum(1)[:, None] - 0.5 * (V * V).sum(1)[None, :]
    arg_h = Uh @ Vh.T - 0.5 * (U * U).sum(1)[:, None] - 0.5 * (V * V).sum(1)[None, :]
    K, Kh = np.exp2(arg), np.exp2(arg_h)
    dec = K @ coef - m["rho"]
    S = K @ np.abs(coef)
    err = np.abs((Kh - K) @ coef)
    un, dun = np.linalg.norm(U, axis=1), np.linalg.norm(dU, axis=1)
    vmax, dvmax = np.linalg.norm(Vh, axis=1).max(), np.linalg.norm(dV, axis=1).max()
    band_cs = np.log(2) * (dun * vmax + (un + dun) * dvmax) * S
    sV = np.sqrt(lam_max_upper(Vh.T @ Vh))
    sdV = np.sqrt(lam_max_upper(dV.T @ dV))
    cmax = np.abs(coef).max()
    w2 = np.sqrt(((K * coef[None, :]) ** 2).sum(1))
    lin = np.log(2) * (dun * sV + (un + dun) * sdV)
    second = 0.6 * (np.log(2) * (dun * vmax + (un + dun) * dvmax)) ** 2
    band_w2 = lin * w2 + second * S
    band_sqrt = lin * np.sqrt(cmax * S) + second * S
    print("model nSV=%d gamma=%g  samples=%d   |u| median %.3f  |du|/|u| median %.2e  sigma(V^)=%.3f (Vmax %.3f)  sigma(dV)=%.2e (dVmax %.2e)"
          % (len(coef), gamma, len(X), np.median(un), np.median(dun / un), sV, vmax, sdV, dvmax))
    print("|dec|/S median %.3e   S median %.1f  |w|2 median %.2f  sqrt(cmax S) median %.2f" % (np.median(np.abs(dec) / S), np.median(S), np.median(w2), np.median(np.sqrt(cmax * S))))
    for name, b in (("Cauchy-Schwarz per SV (current)", band_cs), ("spectral, exact |w|2", band_w2), ("spectral, sqrt(cmax*S)", band_sqrt)):
        acc = 2.0 ** -24 * (2 * len(coef) / 32 + 12) * S
        print("%-34s band/S median %.2e  inside-band share %.4f  (with acc term %.4f)   max err/band %.3f" %
              (name, np.median(b / S), np.mean(np.abs(dec) <= b), np.mean(np.abs(dec) <= b + acc), (err / b).max()))
    print("actual single-pass error / S: median %.2e max %.2e" % (np.median(err / S), (err / S).max()))


if __name__ == "__main__":
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 4096, sys.argv[2] if len(sys.argv) > 2 else None)
