#!/usr/bin/env python3
"""One C5 request through haf_score_sharded with k shards on ONE GPU (k engines, k streams, k host threads): does splitting the
rolls of a request over concurrent streams of the same device help?  python tools/time_shards.py [k ...]"""
import os, sys, tempfile, time
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
inp = capi.default_input(grasp_area_length_x=512, grasp_area_length_y=512)
for k in [int(a) for a in sys.argv[1:]] or [1, 2, 3, 4]:
    me = capi.MultiEngine(feat, rng, mp, [0] * k, capi.SHARD_ROLLS, grid_h=512, grid_w=512, n_rolls=36, roll_step_deg=5, max_clouds=1,
                          max_points=1 << 20)
    cloud = (d.data_ptr(), xyz.shape[0], 3)
    for _ in range(3):
        out = me.score_sharded(cloud, inp)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 10
    for _ in range(n):
        out = me.score_sharded(cloud, inp)
    dt = (time.perf_counter() - t0) / n
    print("shards on one GPU: %d  ms/request %.2f  evals/s %.3e  best %s" % (k, 1e3 * dt, out["n_evals"] / dt,
          (out["eval"], out["best_row"], out["best_col"], out["best_roll"])), flush=True)
    me.close()
