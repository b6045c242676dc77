#!/usr/bin/env python3
"""What do the row remainders cost the thread-per-evaluation feature kernel?  Of every grid row the first floor(count / 64) * 64 masked
cells are waves of 64 neighbours (the band path: windows staged per wave, corners by ds_read_addtid); the remainder of every row ends
up in waves that are NOT a run of neighbours (per-lane corner loads, per-region rounding bounds).  All 36 rolls (the rotated area cuts the rows differently per roll: the remainders are ~32 cells on average then), grasp areas whose
rows hold 448 (= 7 x 64: no remainder), 498 (the bench's: 50 left over) and 480 (32 left over) masked cells: feature-stage time per
evaluation.  On a GPU box: python tools/seg_b_cost.py"""
import os, sys, tempfile
import numpy as np
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import models
from haf_grasping_amd import capi
D = os.path.join(ROOT, "tests", "golden", "data")
feat, rng = os.path.join(D, "Features.txt"), os.path.join(D, "range21062012_allfeatures")
mp = models.write_random_model(os.path.join(tempfile.mkdtemp(), "m.model"), 4096, D=323, seed=42, balanced=True)
G = 512
xyz = models.synthetic_cloud(grid=G, k=2, seed=0)
d_xyz = torch.from_numpy(xyz).cuda()
cloud = (d_xyz.data_ptr(), xyz.shape[0], 3)
eng = capi.Engine(feat, rng, mp, device=0, grid_h=G, grid_w=G, n_rolls=36, roll_step_deg=5, max_clouds=1, max_points=G * G * 2, flags=capi.FLAG_PROFILE)
for ly in (462, 512, 494, 462, 512):
    inp = capi.default_input(grasp_area_length_x=G, grasp_area_length_y=ly)
    ts, ev = [], 0
    for _ in range(6):
        rec = eng.score_rolls([cloud], [inp], 0, 36)[0]
        ev = int(rec["n_evals"].sum())
        ts.append(eng.stage_ms()["features"])
    t = float(np.median(ts[2:]))
    print("grasp area %d cells wide: %d evaluations (%d per row, %d in whole waves), features %.3f ms = %.3f ns per evaluation" %
          (ly, ev, ly - 14, (ly - 14) // 64 * 64, t, 1e6 * t / ev), flush=True)
eng.close()
