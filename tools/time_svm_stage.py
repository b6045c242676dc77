#!/usr/bin/env python3
"""Times the SVM stage (HIP events) of the default bench workload for the library named by HAF_LIB; prints one line.
Used for A/B timing of kernel variants on one GPU box (results of ablated variants are not checked)."""
import os, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import models
from haf_grasping_amd import capi
data = os.path.join(ROOT, "tests", "golden", "data")
feat, rng = os.path.join(data, "Features.txt"), os.path.join(data, "range21062012_allfeatures")
mp = os.path.join(tempfile.mkdtemp(), "m.model")
models.write_random_model(mp, 4096, D=323, seed=1234, balanced=True)
xyz = models.synthetic_cloud(grid=512, k=2, seed=0)
d = torch.from_numpy(xyz).cuda()
eng = capi.Engine(feat, rng, mp, grid_h=512, grid_w=512, n_rolls=36, roll_step_deg=5, max_points=1 << 20, flags=capi.FLAG_PROFILE)
inp = capi.default_input(grasp_area_length_x=512, grasp_area_length_y=512)
ts = []
for i in range(int(os.environ.get('HAF_ITERS', '4'))):
    try:
        eng.score_rolls([(d.data_ptr(), xyz.shape[0], 3)], [inp], 0, 36)
    except capi.HafError:
        pass
    ts.append(eng.stage_ms()["svm"])
print(os.path.basename(os.environ.get("HAF_LIB", "default")), "svm stage ms:", ["%.2f" % t for t in ts])
