#!/usr/bin/env python3
"""What haf_create costs (tables, probes, model classification) on the GPU box: python tools/create_time.py  (from the repo root)"""
import os, sys, time
sys.path.insert(0, os.getcwd()); sys.path.insert(0, "tests")
from haf_grasping_amd import capi
import models, tempfile
D = "tests/golden/data"
F, R = os.path.join(D, "Features.txt"), os.path.join(D, "range21062012_allfeatures")
for name, model, cfg in (("56x56 surrogate", "tests/golden/surrogate.model", dict()), ("56x56 surrogate again", "tests/golden/surrogate.model", dict()),):
    t0 = time.perf_counter(); e = capi.Engine(F, R, model, **cfg); t1 = time.perf_counter(); e.close()
    print("haf_create %-24s %.1f ms" % (name, 1e3 * (t1 - t0)))
mp = os.path.join(tempfile.mkdtemp(), "m.model"); models.write_random_model(mp, 4096, D=323, seed=42, balanced=True)
for i in range(2):
    t0 = time.perf_counter(); e = capi.Engine(F, R, mp, grid_h=512, grid_w=512, n_rolls=36, roll_step_deg=5, max_points=1 << 20); t1 = time.perf_counter(); e.close()
    print("haf_create 512x512 nSV 4096 (%d)   %.1f ms" % (i, 1e3 * (t1 - t0)))
