#!/usr/bin/env python3
"""Measures the decision error of a fast tier relative to S = sum|coef|K against the fp64 oracle, with its guard band
disabled, to calibrate the band.  HAF_DIAG_MODE = screen (default: the single-pass fp16 screening kernel,
HAF_GUARD0_REL=0), split (three-pass kernel) or f32 (fp32 MFMA kernel), both with HAF_GUARD_REL=0.  GPU box only."""
import os
import sys

os.environ["HAF_GUARD_REL"] = "0"
os.environ["HAF_GUARD0_REL"] = "0"
MODE = os.environ.get("HAF_DIAG_MODE", "screen")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))  # repo root (this file lives in tests/)
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402

import models  # noqa: E402
from haf_grasping_amd import capi  # noqa: E402
from oracle import oracle as O  # noqa: E402

DATA = os.path.join(ROOT, "tests", "golden", "data")
F, R = os.path.join(DATA, "Features.txt"), os.path.join(DATA, "range21062012_allfeatures")


def run(model, names, tag):
    o = O.Oracle(F, R, model)
    eng = capi.Engine(F, R, model, flags=capi.FLAG_KEEP_DEBUG | {"screen": 0, "split": capi.FLAG_SPLIT_F16, "f32": capi.FLAG_FP32_MFMA}[MODE],
                      max_points=1 << 18)
    worst, errs = 0.0, []
    for name in names:
        xyz = capi.load_pcd(os.path.join(DATA, name + ".pcd"))
        want = o.run(xyz, O.make_cfg(), O.make_input(length_y=44))
        eng.score(xyz, capi.default_input())
        for roll in range(12):
            d = eng.debug(capi.DBG_DECISION, 0, roll)
            m = want["mask"][roll] == 1
            if m.any():
                rel = np.abs(d[m] - want["dec"][roll][m]) / want["sabs"][roll][m]
                errs.append(rel)
    errs = np.concatenate(errs)
    print("%-28s n=%6d  max %.3e  p99.9 %.3e  p99 %.3e  median %.3e   (2^-15=%.2e 2^-17=%.2e 2^-18=%.2e)" %
          (tag, len(errs), errs.max(), np.quantile(errs, 0.999), np.quantile(errs, 0.99), np.median(errs),
           2.0 ** -15, 2.0 ** -17, 2.0 ** -18))
    eng.close()


if __name__ == "__main__":
    run(os.path.join(ROOT, "tests", "golden", "surrogate.model"), ["pcd1", "pcd2", "pcd3", "pcd12", "plastic_mug2", "pcd10"], "surrogate nSV=172")
    for nsv, seed in ((64, 1), (512, 2), (2048, 3)):
        p = "/tmp/diag_%d.model" % nsv
        models.write_random_model(p, nsv, seed=seed, balanced=True)
        run(p, ["pcd2", "pcd3"] if nsv > 1000 else ["pcd1", "pcd2", "pcd3", "pcd12"], "random nSV=%d" % nsv)
    p = "/tmp/diag_g.model"
    models.write_random_model(p, 256, seed=9, balanced=True, gamma=0.05)
    run(p, ["pcd2", "pcd3"], "random nSV=256 gamma=0.05")
