#!/usr/bin/env python3
"""Stage times (HIP events) of the default bench workload for the library named by HAF_LIB; one line.  A/B timing of builds."""
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
acc = {}
for i in range(6):
    eng.score_rolls([(d.data_ptr(), xyz.shape[0], 3)], [inp], 0, 36)
    if i >= 2:
        for k, v in eng.stage_ms().items(): acc[k] = acc.get(k, 0) + v / 4
print("%-22s" % os.path.basename(os.environ.get("HAF_LIB", "default")), " ".join("%s=%.3f" % (k, v) for k, v in acc.items() if v > 0.3), "sum=%.3f" % sum(acc.values()))
