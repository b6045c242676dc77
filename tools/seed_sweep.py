#!/usr/bin/env python3
"""C5 step per model seed, with and without the centred form of the screening band (testing build: HAF_SCREEN_NO_CENTRE):
stage times (HIP events), tier counts, same-labels check between the two.  One line per (seed, variant).

  python tools/seed_sweep.py [--seeds 1234,7,11,23,42] [--nsv 4096] [--steps 4] [--hard] [--no-ab]
"""
import argparse
import json
import os
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import models  # noqa: E402
from haf_grasping_amd import capi  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seeds", default="1234,7,11,23,42")
    ap.add_argument("--nsv", type=int, default=4096)
    ap.add_argument("--steps", type=int, default=4)
    ap.add_argument("--grid", type=int, default=512)
    ap.add_argument("--rolls", type=int, default=36)
    ap.add_argument("--hard", action="store_true")
    ap.add_argument("--trained", action="store_true", help="the 8964-SV trained model (tests/golden/trained.model.npz)")
    ap.add_argument("--no-ab", action="store_true", help="only the default build behaviour")
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    data = os.path.join(ROOT, "tests", "golden", "data")
    feat, rng = os.path.join(data, "Features.txt"), os.path.join(data, "range21062012_allfeatures")
    tmp = tempfile.mkdtemp(prefix="seedsweep_")
    G = a.grid
    xyz = models.synthetic_cloud(grid=G, k=2, seed=0)
    d = torch.from_numpy(xyz).cuda()
    cloud = (d.data_ptr(), xyz.shape[0], 3)
    inp = capi.default_input(grasp_area_length_x=G, grasp_area_length_y=G)
    cases = [("seed%d" % int(s), int(s)) for s in a.seeds.split(",") if s.strip()]
    if a.hard:
        cases.append(("hard", None))
    if a.trained:
        cases.append(("trained", "trained"))
    rows = []
    for name, seed in cases:
        mp = os.path.join(tmp, name + ".model")
        if seed == "trained":
            models.unpack_trained_model(os.path.join(ROOT, "tests", "golden", "trained.model.npz"), mp)
        elif seed is None:
            models.write_replicated_model(mp, os.path.join(ROOT, "tests", "golden", "surrogate.model"), copies=24, jitter=0.01, seed=5)
        else:
            models.write_random_model(mp, a.nsv, D=323, seed=seed, balanced=True)
        ref = None
        for variant in (("centred", "plain") if not a.no_ab else ("centred",)):
            os.environ.pop("HAF_SCREEN_NO_CENTRE", None)
            if variant == "plain":
                os.environ["HAF_SCREEN_NO_CENTRE"] = "1"
            eng = capi.Engine(feat, rng, mp, testing=True, grid_h=G, grid_w=G, n_rolls=a.rolls, roll_step_deg=5, max_points=G * G * 2,
                              flags=capi.FLAG_PROFILE)
            acc, cnt, rec = {}, None, None
            for i in range(a.steps + 3):                      # (the first calls may still switch the kernel variant)
                rec = eng.score_rolls([cloud], [inp], 0, a.rolls)[0]
                if i >= 3:
                    for k, v in eng.stage_ms().items():
                        acc[k] = acc.get(k, 0.0) + v / a.steps
                    cnt = eng.last_counts()
            st = eng.screen_state()
            ex = eng.last_exact_tiers()
            eng.close()
            key = [(int(r["vote"]), int(r["row"]), int(r["col"])) for r in rec]
            if ref is None:
                ref = key
            row = dict(model=name, variant=variant, ms=sum(acc.values()), features=acc.get("features"), svm=acc.get("svm"),
                       refine=acc.get("refine"), recheck=acc.get("recheck"), n_evals=cnt["n_evals"], refined=cnt["n_refined"],
                       refined_share=cnt["n_refined"] / max(1, cnt["n_evals"]), fp64=cnt["n_rechecked"], strict=cnt["n_strict"],
                       same_records=(key == ref), state=st, exact=ex)
            rows.append(row)
            print("%-9s %-8s step %.2f ms  features %.2f svm %.2f refine %.2f recheck %.2f  refined %.3f %% (%d)  exact tiers %d (i8 %d fp64 %d) strict %d  same %s  form %d%s%s"
                  % (name, variant, row["ms"], row["features"], row["svm"], row["refine"], row["recheck"], 100 * row["refined_share"],
                     row["refined"], row["fp64"], ex["n_integer"], ex["n_fp64"], row["strict"], row["same_records"], st["variant"],
                     "+0b" if st["tier0b"] else "", " t1-skip" if st["tier1_skipped"] else ""), flush=True)
    if a.out:
        with open(a.out, "w") as f:
            json.dump(rows, f, indent=1)


if __name__ == "__main__":
    main()
