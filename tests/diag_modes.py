#!/usr/bin/env python3
"""Diagnostic (GPU box only): per contraction mode, label mismatches and decision errors against the oracle on C1."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402

from haf_grasping_amd import capi  # noqa: E402
from oracle import oracle as O  # noqa: E402

DATA = os.path.join(ROOT, "tests", "golden", "data")
F, R = os.path.join(DATA, "Features.txt"), os.path.join(DATA, "range21062012_allfeatures")
MODEL = os.path.join(ROOT, "tests", "golden", "surrogate.model")

if __name__ == "__main__":
    xyz = capi.load_pcd(os.path.join(DATA, "pcd2.pcd"))
    o = O.Oracle(F, R, MODEL)
    NR = int(os.environ.get("HAF_DIAG_ROLLS", "1"))
    want = o.run(xyz, O.make_cfg(n_rolls=NR), O.make_input(length_x=32, length_y=32))
    for name, fl, g0 in (("f32", capi.FLAG_FP32_MFMA, None), ("split", capi.FLAG_SPLIT_F16, None), ("screen", 0, None),
                         ("screen, band 0: tier 0 only", 0, "0"), ("screen, band inf: all through the list-mode three-pass kernel", 0, "1e30")):
        os.environ.pop("HAF_GUARD0_REL", None)
        if g0 is not None:
            os.environ["HAF_GUARD0_REL"] = g0
        eng = capi.Engine(F, R, MODEL, n_rolls=NR, flags=capi.FLAG_KEEP_DEBUG | capi.FLAG_PROFILE | fl)
        eng.score(xyz, capi.default_input(grasp_area_length_x=32, grasp_area_length_y=32))
        for roll in range(1, NR):
            dr = eng.debug(capi.DBG_DECISION, 0, roll)
            mr = want["mask"][roll] == 1
            if mr.any():
                relr = np.abs(dr[mr] - want["dec"][roll][mr]) / want["sabs"][roll][mr]
                if relr.max() > 1e-5:
                    print("   roll", roll, "rel err max %.3e at masked index %d of %d" % (relr.max(), int(relr.argmax()), int(mr.sum())),
                          "labels wrong", int(((eng.debug(capi.DBG_LABELS, 0, roll) != want["labels"][roll]) & mr).sum()))
        lab = eng.debug(capi.DBG_LABELS, 0, 0)
        d = eng.debug(capi.DBG_DECISION, 0, 0)
        m = want["mask"][0] == 1
        bad = (lab != want["labels"][0]) & m
        rel = np.abs(d[m] - want["dec"][0][m]) / want["sabs"][0][m]
        print(name, eng.last_counts(), "label mismatches", int(bad.sum()), "rel err max %.3e median %.3e" % (rel.max(), np.median(rel)))
        order = {tuple(c): k for k, c in enumerate(np.argwhere(m))}          # row-major = evaluation index
        relmap = np.abs(d - want["dec"][0]) / np.where(m, want["sabs"][0], 1.0)
        big = np.argwhere(m & (relmap > 1e-5))
        idx = sorted(order[tuple(c)] for c in big)
        print("   evaluations with rel err > 1e-5: n=%d  waves(e//64)=%s  rows(e%%64)=%s" %
              (len(idx), sorted(set(i // 64 for i in idx)), sorted(set(i % 64 for i in idx))))
        for i, j in np.argwhere(bad)[:6]:
            print("   cell", i, j, "dec", d[i, j], "oracle", want["dec"][0][i, j], "sabs", want["sabs"][0][i, j])
        eng.close()
