#!/usr/bin/env python3
"""Which path of the low-rank feature kernel do the evaluations of a C5 request take?  From the engine's own integral image and mask
(HAF_FLAG_KEEP_DEBUG) the device's evaluation order (k_compact) and the kernel's wave-wide exactness rule (features.hip) are rebuilt in
numpy, per roll: share of the evaluations in whole waves that pass the rule (path A: region sums exact, six instructions of noise bound
per slot), in whole waves that do not (per-region bounds), and in the row remainders (no run of neighbours: per-lane corner loads).
On a GPU box: python tools/path_a_share.py [--rolls 0,7,18,27]"""
import argparse, os, sys
import numpy as np
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import tempfile
import models
from haf_grasping_amd import capi
ap = argparse.ArgumentParser()
ap.add_argument("--rolls", default="0,7,18,27")
ap.add_argument("--grid", type=int, default=512)
a = ap.parse_args()
G = a.grid
D = os.path.join(ROOT, "tests", "golden", "data")
mp = models.write_random_model(os.path.join(tempfile.mkdtemp(), "m.model"), 256, D=323, seed=42, balanced=True)
xyz = models.synthetic_cloud(grid=G, k=2, seed=0)
eng = capi.Engine(os.path.join(D, "Features.txt"), os.path.join(D, "range21062012_allfeatures"), mp, grid_h=G, grid_w=G, n_rolls=36, roll_step_deg=5,
                  max_clouds=1, max_points=G * G * 2, flags=capi.FLAG_KEEP_DEBUG)
eng.score(xyz, capi.default_input(grasp_area_length_x=G, grasp_area_length_y=G))
tot = np.zeros(4, np.int64)
for roll in [int(t) for t in a.rolls.split(",")]:
    II = eng.debug(capi.DBG_INTEGRAL, 0, roll)
    msk = eng.debug(capi.DBG_MASK, 0, roll) == 1
    neg = bool((eng.debug(capi.DBG_HEIGHTS, 0, roll) < 0).any())
    n_rem = n_fast = n_a = n_notrun = 0
    for i in range(G):
        cols = np.nonzero(msk[i])[0]
        rem = len(cols) % 64
        n_rem += rem
        for c0 in range(rem, len(cols), 64):
            cj = cols[c0:c0 + 64]
            if not (np.diff(cj) == 1).all():
                n_notrun += 64
                continue
            n_fast += 64
            ci = np.full(64, i)
            tl, bl, tr, br = II[ci - 7, cj - 7], II[ci + 7, cj - 7], II[ci - 7, cj + 7], II[ci + 7, cj + 7]
            ok = np.ones(64, bool)
            for c in range(15):
                ok &= II[ci + 7, cj - 7 + c] <= np.float32(2.0) * II[ci - 7, cj - 7 + c]
            tA, tB = (br - tr).astype(np.float32), (bl - tl).astype(np.float32)
            T = ((tA - tB).astype(np.float32) + np.float32(2.0e-7) * (np.abs(tA) + np.abs(tB))) * np.float32(1.0001)
            dmin = tl.copy()
            if cj[0] - 7 == 0:
                dmin[0] = II[i - 7, 1]
            if (ok & (T < dmin)).all() and not neg:
                n_a += 64
    n = n_rem + n_fast + n_notrun
    print("roll %2d: %7d evaluations; whole waves on the exact path %.3f, whole waves on per-region bounds %.3f, waves of a row with gaps %.3f, row remainders %.3f%s" %
          (roll, n, n_a / n, (n_fast - n_a) / n, n_notrun / n, n_rem / n, "  (negative heights: no exact path)" if neg else ""), flush=True)
    tot += np.array([n_a, n_fast - n_a, n_notrun, n_rem])
print("all: exact %.3f, per-region %.3f, gaps %.3f, remainders %.3f" % tuple(tot / tot.sum()))
eng.close()
