#!/usr/bin/env python3
"""Grids far beyond the benchmark's (GPU box): 700 x 700 (bucket-sorted binning near its bucket limit) and 1000 x 1000 (past it: the
atomic binning path), a few rolls, small random model, every stage against the oracle.  python tools/big_grid_check.py"""
import os, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import models
import test_engine_gpu as T
from haf_grasping_amd import capi
from oracle import oracle as O
DATA = os.path.join(ROOT, "tests", "golden", "data")
tmp = tempfile.mkdtemp()
mp = models.write_random_model(os.path.join(tmp, "m.model"), 33, seed=5, balanced=True)
orc = O.Oracle(os.path.join(DATA, "Features.txt"), os.path.join(DATA, "range21062012_allfeatures"), mp)
for G, rolls, step, mode in [(700, 3, 50, 0), (1000, 2, 75, 0), (1000, 2, 75, capi.FLAG_SPLIT_F16)]:
    t0 = time.time()
    xyz = models.synthetic_cloud(grid=G, k=1, seed=G)
    eng = T.make_engine(DATA, mp, mode, grid_h=G, grid_w=G, n_rolls=rolls, roll_step_deg=step, max_points=G * G + 16)
    got, want = T.compare_full(eng, orc, xyz, dict(n_rolls=rolls, roll_step_deg=step, grid_h=G, grid_w=G),
                               dict(grasp_area_length_x=G, grasp_area_length_y=G - 100))
    eng.close()
    print("grid %d, %d rolls, mode %d: %d evaluations, every stage identical to the oracle (%.0f s)" % (G, rolls, mode, want["n_evals"], time.time() - t0), flush=True)
