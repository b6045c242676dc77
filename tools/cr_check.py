#!/usr/bin/env python3
"""GPU check of the screening pass's forms (kernels.h SCREEN_*): for each model and each pinned form (testing build,
HAF_SCREEN_VARIANT) one C2 and one C3 request against the oracle -- labels, winner -- with the tier counts, then the form
calibrate() picks by itself.   python tools/cr_check.py [--models surrogate,trained,rand7,hard] [--c5]"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402

import models  # noqa: E402

DATA = os.path.join(ROOT, "tests", "golden", "data")
F, R = os.path.join(DATA, "Features.txt"), os.path.join(DATA, "range21062012_allfeatures")


def model_path(name):
    if name == "surrogate":
        return os.path.join(ROOT, "tests", "golden", "surrogate.model")
    p = "/tmp/cr_check_%s.model" % name
    if os.path.exists(p):
        return p
    if name == "trained":
        models.unpack_trained_model(os.path.join(ROOT, "tests", "golden", "trained.model.npz"), p)
    elif name == "hard":
        models.write_replicated_model(p, os.path.join(ROOT, "tests", "golden", "surrogate.model"), copies=24, jitter=0.01, seed=5)
    elif name.startswith("rand"):
        models.write_random_model(p, 4096, seed=int(name[4:]), balanced=True)
    return p


def child(model, variant, c5):
    from haf_grasping_amd import capi
    from oracle import oracle as O
    import pcdio
    res = {}
    mp = model_path(model)
    cases = [("pcd2", dict(n_rolls=12), dict(length_x=32, length_y=32)),
             ("table1_mult_obj_rcs_1428580506606673", dict(n_rolls=20, roll_step_deg=9), dict(length_x=56, length_y=56))]
    orc = O.Oracle(F, R, mp)
    for cloud, cfgk, inpk in cases:
        xyz = pcdio.load_pcd(os.path.join(DATA, cloud + ".pcd"))
        t0 = time.time()
        eng = capi.Engine(F, R, mp, testing=True, flags=capi.FLAG_KEEP_DEBUG, **cfgk)
        t_create = time.time() - t0
        got = eng.score(xyz, capi.default_input(grasp_area_length_x=inpk["length_x"], grasp_area_length_y=inpk["length_y"]))
        st = eng.screen_state()
        cnt = eng.last_counts()
        ex = eng.last_exact_tiers()
        want = orc.run(xyz, O.make_cfg(**cfgk), O.make_input(**inpk))
        bad = 0
        worst = 0.0
        for r in range(cfgk["n_rolls"]):
            lab = eng.debug(capi.DBG_LABELS, 0, r)
            bad += int((lab != want["labels"][r]).sum())
            dec = eng.debug(capi.DBG_DECISION, 0, r)
            m = want["mask"][r] == 1
            if m.any():
                worst = max(worst, float(np.nanmax(np.abs(dec[m] - want["dec"][r][m]))))
        same = (got["eval"], got["best_row"], got["best_col"], got["best_roll"]) == (want["eval"], want["row"], want["col"], want["roll_idx"])
        res[cloud[:6]] = dict(state=st, counts=cnt, exact=ex, label_mismatches=bad, same_best=bool(same), worst_dec_err=worst, create_s=round(t_create, 2))
        eng.close()
    if c5:
        cloud = models.synthetic_cloud(512, 2, 0)
        eng = capi.Engine(F, R, mp, testing=True, flags=capi.FLAG_PROFILE, grid_h=512, grid_w=512, n_rolls=36, roll_step_deg=5, max_points=1 << 20)
        inp = capi.default_input(grasp_area_length_x=512, grasp_area_length_y=512)
        best = None
        for it in range(3):
            t0 = time.time()
            got = eng.score(cloud, inp)
            dt = time.time() - t0
        st = eng.stage_ms()
        res["c5"] = dict(state=eng.screen_state(), counts=eng.last_counts(), exact=eng.last_exact_tiers(), ms=round(dt * 1e3, 2),
                         stages={k: round(v, 3) for k, v in st.items()}, best=(got["eval"], got["best_row"], got["best_col"], got["best_roll"]))
        eng.close()
    print("RESULT " + json.dumps(res))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--models", default="surrogate,trained,rand7,hard")
    ap.add_argument("--variants", default="0,1,2,3,auto")
    ap.add_argument("--c5", action="store_true")
    ap.add_argument("--child", nargs=2)
    a = ap.parse_args()
    if a.child:
        child(a.child[0], a.child[1], a.c5)
        return
    for m in a.models.split(","):
        model_path(m)
        for v in a.variants.split(","):
            env = dict(os.environ)
            if v != "auto":
                env["HAF_SCREEN_VARIANT"] = v
            out = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", m, v] + (["--c5"] if a.c5 else []), env=env,
                                 stdout=subprocess.PIPE, stderr=subprocess.STDOUT).stdout.decode()
            line = [ln for ln in out.splitlines() if ln.startswith("RESULT ")]
            if not line:
                print(m, v, "FAILED:\n" + out[-2000:], flush=True)
                continue
            r = json.loads(line[0][7:])
            for case, d in r.items():
                print("%-9s variant %-4s %-6s state %s  counts %s exact %s  %s" %
                      (m, v, case, d["state"]["variant"] if d["state"]["active"] else "off", d["counts"], d["exact"],
                       {k: d[k] for k in d if k not in ("state", "counts", "exact")}), flush=True)
                if v == "auto":
                    print("          calibration shares", ["%.4f" % s for s in d["state"]["shares"]], flush=True)


if __name__ == "__main__":
    main()
