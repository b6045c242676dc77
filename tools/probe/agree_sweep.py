#!/usr/bin/env python3
"""Ad-hoc confidence run (not a test): default (screened) mode and fp32 mode vs three-pass mode, every label of C5-size requests, over several
seeded models and clouds, including unbalanced coefficients and other gammas."""
import os, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import models
from haf_grasping_amd import capi
data = os.path.join(ROOT, "tests", "golden", "data")
feat, rng_file = os.path.join(data, "Features.txt"), os.path.join(data, "range21062012_allfeatures")
tmp = tempfile.mkdtemp()
G, R = 512, 36
inp = capi.default_input(grasp_area_length_x=G, grasp_area_length_y=G)
bad = 0
for nsv, mseed, cseed, bal in ((1024, 11, 1, True), (2048, 12, 2, False), (777, 13, 3, True), (4096, 14, 4, False), (300, 15, 5, True)):
    path = os.path.join(tmp, "m%d.model" % mseed)
    models.write_random_model(path, nsv, seed=mseed, balanced=bal)
    xyz = models.synthetic_cloud(grid=G, k=2, seed=cseed)
    ref = None
    for flags in (capi.FLAG_SPLIT_F16, 0, capi.FLAG_FP32_MFMA):
        eng = capi.Engine(feat, rng_file, path, grid_h=G, grid_w=G, n_rolls=R, roll_step_deg=5, max_clouds=1, max_points=1 << 20, flags=flags | capi.FLAG_KEEP_DEBUG)
        rec = eng.score_rolls([xyz], [inp], 0, R)[0]
        c = eng.last_counts()
        labels = np.stack([eng.debug(capi.DBG_LABELS, 0, roll) for roll in range(R)])
        eng.close()
        if ref is None:
            ref, ref_rec = labels, rec
        else:
            diff = int((labels != ref).sum())
            bad += diff
            print("nSV %4d model seed %d cloud seed %d balanced %s: evals %d refined %d (%.2f%%) fp64 %d  differing labels %d  records equal %s  +1 share %.3f"
                  % (nsv, mseed, cseed, bal, c["n_evals"], c["n_refined"], 100.0 * c["n_refined"] / max(1, c["n_evals"]), c["n_rechecked"], diff,
                     bool((rec == ref_rec).all()), float((labels == 1).sum()) / max(1, (labels != 0).sum() if False else (np.abs(labels) == 1).sum())))
print("TOTAL differing labels:", bad)
