#!/usr/bin/env python3
"""How much does the third-party arithmetic that the reference leaves unpinned matter?  (CPU only: the oracle.)

server.cpp evaluates its six-matrix product with Eigen (483), transforms the cloud with pcl::transformPointCloud (488) and builds
the summed-area table with cv::integral (595); none of the three libraries is vendored or version-pinned and the reference has no
golden outputs, so the oracle DEFINES one evaluation order (variant 0) and the product reproduces it bit for bit.  This script
runs every golden (cloud x configuration) of tests/golden/g6_end_to_end.json again under the plausible alternative orders
(oracle/haf_oracle.h: HAFO_V_*) and counts what changes: height cells (bitwise / by more than 1e-6 m, i.e. a point that landed in
another cell or another point that became a cell's maximum), mask cells, labels, per-roll winners, the final grasp, and the largest
shift of the returned grasp points.

  python tools/unpinned_arithmetic.py [--out profiles/r03_unpinned_arithmetic.json] [--only pcd2/C2,pcd3/C4]
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import numpy as np  # noqa: E402

import pcdio  # noqa: E402
from oracle import oracle as O  # noqa: E402

DATA = os.path.join(ROOT, "tests", "golden", "data")
FEATURES, RANGE = os.path.join(DATA, "Features.txt"), os.path.join(DATA, "range21062012_allfeatures")
MODEL = os.path.join(ROOT, "tests", "golden", "surrogate.model")

VARIANTS = [
    ("eigen_tree", O.V_EIGEN_TREE, "Eigen 4-term reduction (a0b0+a1b1)+(a2b2+a3b3) in every 4x4 product (server.cpp:483, 1334)"),
    ("chain_rtl", O.V_CHAIN_RTL, "six-matrix product associated from the right (server.cpp:483)"),
    ("pcl_sse", O.V_PCL_SSE, "pcl::transformPointCloud as (m0x+m1y)+(m2z+m3) (PCL >= 1.8 SSE path; server.cpp:488)"),
    ("fma", O.V_FMA, "point transform with a*b+c contracted to fma (server.cpp:488 built with contraction)"),
    ("integral_colfirst", O.V_INTEGRAL_COLFIRST, "summed-area table by running column sums (server.cpp:595)"),
    ("all_alternatives", O.V_EIGEN_TREE | O.V_CHAIN_RTL | O.V_PCL_SSE | O.V_INTEGRAL_COLFIRST, "tree + right-to-left + SSE + column-first together"),
]


def configs():
    import make_fixtures as MF          # CONFIGS / CLOUD_CONFIGS of the committed goldens (data only)
    return MF.CONFIGS, MF.CLOUD_CONFIGS


def run_case(orc, xyz, spec, variant):
    O.set_variant(variant)
    try:
        return orc.run(xyz, O.make_cfg(**spec["cfg"]), O.make_input(**spec["inp"]))
    finally:
        O.set_variant(0)


def compare(base, alt):
    hb, ha = base["heights"], alt["heights"]
    bits = int((hb.view(np.uint32) != ha.view(np.uint32)).sum())
    moved = int((np.abs(hb - ha) > 1e-6).sum())
    mask_changed = int((base["mask"] != alt["mask"]).sum())
    both = (base["mask"] == 1) & (alt["mask"] == 1)
    labels_changed = int((base["labels"][both] != alt["labels"][both]).sum())
    rolls = min(base["rolls_done"], alt["rolls_done"])
    winners_changed = int((base["roll_best"][:rolls] != alt["roll_best"][:rolls]).any(axis=1).sum())
    grasp_changed = int((base["row"], base["col"], base["roll_idx"], base["eval"]) != (alt["row"], alt["col"], alt["roll_idx"], alt["eval"]))
    shift = float(max(np.abs(np.array(base["gp1"]) - np.array(alt["gp1"])).max(), np.abs(np.array(base["gp2"]) - np.array(alt["gp2"])).max()))
    return dict(height_cells_bits=bits, height_cells_moved=moved, mask_cells=mask_changed, labels=labels_changed,
                roll_winners=winners_changed, grasp=grasp_changed, grasp_point_shift_m=shift if not grasp_changed else None,
                cells=int(hb.size), masked=int((base["mask"] == 1).sum()), rolls=int(rolls))


def campaign(only=None, threads=None):
    if threads:
        os.environ["HAFO_THREADS"] = str(threads)
    CONFIGS, CLOUD_CONFIGS = configs()
    orc = O.Oracle(FEATURES, RANGE, MODEL)
    cases = {}
    for name, cfgs in CLOUD_CONFIGS:
        xyz = None
        for cname in cfgs:
            key = "%s/%s" % (name, cname)
            if only and key not in only:
                continue
            if xyz is None:
                xyz = pcdio.load_pcd(os.path.join(DATA, name + ".pcd"))
            base = run_case(orc, xyz, CONFIGS[cname], 0)
            cases[key] = {vn: compare(base, run_case(orc, xyz, CONFIGS[cname], vf)) for vn, vf, _ in VARIANTS}
    return cases


def summarise(cases):
    out = {}
    for vn, _, what in VARIANTS:
        rows = [c[vn] for c in cases.values()]
        out[vn] = dict(what=what, cases=len(rows),
                       **{k: int(sum(r[k] for r in rows)) for k in ("height_cells_bits", "height_cells_moved", "mask_cells", "labels",
                                                                    "roll_winners", "grasp", "cells", "masked", "rolls")},
                       cases_with_a_changed_grasp=[k for k, c in cases.items() if c[vn]["grasp"]],
                       max_grasp_point_shift_m=max([r["grasp_point_shift_m"] or 0.0 for r in rows] + [0.0]))
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(ROOT, "profiles", "r03_unpinned_arithmetic.json"))
    ap.add_argument("--only", default=None)
    ap.add_argument("--threads", type=int, default=len(os.sched_getaffinity(0)))
    a = ap.parse_args()
    cases = campaign(set(a.only.split(",")) if a.only else None, a.threads)
    summ = summarise(cases)
    with open(a.out, "w") as f:
        json.dump(dict(model="tests/golden/surrogate.model", variants={vn: what for vn, _, what in VARIANTS}, summary=summ, cases=cases),
                  f, indent=1, sort_keys=True)
    print("%-20s %6s %10s %10s %8s %8s %8s %6s  %s" % ("variant", "cases", "h.bits", "h.moved", "mask", "labels", "winners", "grasp", "max shift [m]"))
    for vn, r in summ.items():
        print("%-20s %6d %10d %10d %8d %8d %8d %6d  %.2e" % (vn, r["cases"], r["height_cells_bits"], r["height_cells_moved"], r["mask_cells"],
                                                             r["labels"], r["roll_winners"], r["grasp"], r["max_grasp_point_shift_m"]))
    print("of %d height cells, %d masked cells, %d rolls in all" % (next(iter(summ.values()))["cells"], next(iter(summ.values()))["masked"],
                                                                   next(iter(summ.values()))["rolls"]))


if __name__ == "__main__":
    main()
