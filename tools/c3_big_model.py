#!/usr/bin/env python3
"""The reference's C3 request (table1_mult_obj, 56x56 cm, 20 rolls of 9 degrees, host cloud) against BIG models -- the trained 8964-SV
model and a seeded random 4096-SV one: wall time per request and the stage times (HAF_FLAG_PROFILE).  On a GPU box:
    python tools/c3_big_model.py [trained|rand4096] [--requests N]"""
import os, sys, time, tempfile
import numpy as np
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import models
from haf_grasping_amd import capi
G = os.path.join(ROOT, "tests", "golden"); D = os.path.join(G, "data")
tmp = tempfile.mkdtemp()
tr = os.path.join(tmp, "trained.model"); models.unpack_trained_model(os.path.join(G, "trained.model.npz"), tr)
rnd = os.path.join(tmp, "rand4096.model"); models.write_random_model(rnd, 4096, seed=42, balanced=True)
xyz = capi.load_pcd(os.path.join(D, "table1_mult_obj_rcs_1428580506606673.pcd"))
inp = capi.default_input(grasp_area_length_x=56, grasp_area_length_y=56, grasp_area_center=(0.13, 0.25, 0.0))
which = [a for a in sys.argv[1:] if not a.startswith("-")]
nreq = int(sys.argv[sys.argv.index("--requests") + 1]) if "--requests" in sys.argv else 20
for name, m in (("trained", tr), ("rand4096", rnd)):
    if which and name not in which:
        continue
    for flags in ((capi.FLAG_PROFILE, 0) if "--requests" not in sys.argv else (0,)):
        eng = capi.Engine(os.path.join(D, "Features.txt"), os.path.join(D, "range21062012_allfeatures"), m, flags=flags, max_points=1 << 18, n_rolls=20, roll_step_deg=9)
        for _ in range(3): out = eng.score(xyz, inp)
        ts = []
        for _ in range(nreq):
            t0 = time.perf_counter(); out = eng.score(xyz, inp); ts.append(time.perf_counter() - t0)
        print(name, "profile" if flags else "plain", "median %.3f ms" % (1e3 * np.median(ts)), eng.screen_form(), eng.last_counts(), eng.last_exact_tiers(), flush=True)
        if flags: print("   ", {k: round(v, 3) for k, v in eng.stage_ms().items()})
        eng.close()
