#!/usr/bin/env python3
"""Stage-by-stage latency of small requests (BASELINE configs C2/C3/C4) on the GPU box."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

from haf_grasping_amd import capi  # noqa: E402

DATA = os.path.join(ROOT, "tests", "golden", "data")
F, R = os.path.join(DATA, "Features.txt"), os.path.join(DATA, "range21062012_allfeatures")
MODEL = os.path.join(ROOT, "tests", "golden", "surrogate.model")


def bench(name, clouds, inputs, **cfg):
    prof = "--no-profile" not in sys.argv      # the stage events cost ~5 us each: wall times WITHOUT them are what a client sees
    eng = capi.Engine(F, R, MODEL, flags=capi.FLAG_PROFILE if prof else 0, max_clouds=len(clouds), max_points=1 << 20, **cfg)
    if "--registered" in sys.argv:             # the caller's buffers page-locked once (haf_register_host_cloud): no staging copy on the host
        clouds = [np.ascontiguousarray(c, dtype=np.float32) for c in clouds]
        for c in clouds:
            eng.register_host(c)
    for _ in range(3):
        out = eng.score_batch(clouds, inputs)
    ts, acc = [], {}
    for _ in range(30):
        t0 = time.perf_counter()
        out = eng.score_batch(clouds, inputs)
        ts.append(time.perf_counter() - t0)
        if prof:
            for k, v in eng.stage_ms().items():
                acc[k] = acc.get(k, 0) + v / 30
    ev = sum(o["n_evals"] for o in out)
    print("   tiers:", eng.last_counts(), eng.last_exact_tiers(), "screened" if eng.last_counts().get("n_refined", 0) or False else "")
    print("%-28s wall median %.3f ms  min %.3f ms  evals %d  (%.2e evals/s)  gpu stages: %s  sum %.3f" %
          (name, 1e3 * np.median(ts), 1e3 * min(ts), ev, ev / np.median(ts),
           " ".join("%s=%.3f" % (k, v) for k, v in acc.items()), sum(acc.values())))
    eng.close()


if __name__ == "__main__":
    only = sys.argv[sys.argv.index("--only") + 1] if "--only" in sys.argv else None
    if only in (None, "C2"):
        pcd2 = capi.load_pcd(os.path.join(DATA, "pcd2.pcd"))
        bench("C2 pcd2 32x32 12 rolls", [pcd2], [capi.default_input(grasp_area_length_x=32, grasp_area_length_y=32)])
    if only in (None, "C3"):
        t1 = capi.load_pcd(os.path.join(DATA, "table1_mult_obj_rcs_1428580506606673.pcd"))
        bench("C3 table1 56x56 20 rolls", [t1], [capi.default_input(grasp_area_length_x=56, grasp_area_length_y=56,
                                                                     grasp_area_center=(0.13, 0.25, 0.0))], n_rolls=20, roll_step_deg=9)
    if only in (None, "C4"):
        clouds = [capi.load_pcd(os.path.join(DATA, "pcd%d.pcd" % i)) for i in range(1, 9)]
        bench("C4 pcd1-8 batch 20 rolls", clouds, [capi.default_input() for _ in clouds], n_rolls=20, roll_step_deg=9)
