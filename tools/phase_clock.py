#!/usr/bin/env python3
"""Where k_small_direct's time goes: workgroup 0's 100 MHz wall clock at its phase boundaries, from a -DHAF_PHASE_CLOCK variant build
(never the product).  On a GPU box:
    python -m haf_grasping_amd.build --variant phase -DHAF_PHASE_CLOCK && HAF_LIB=haf_grasping_amd/variants/libhafgrasp_phase.so python tools/phase_clock.py"""
import ctypes, os, sys
import numpy as np
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from haf_grasping_amd import capi
D = os.path.join(ROOT, "tests", "golden", "data")
lib = ctypes.CDLL(capi.LIB_PATH)
if not hasattr(lib, "haf_phase_clock"):
    sys.exit("%s was not built with -DHAF_PHASE_CLOCK" % capi.LIB_PATH)
names = ["descriptors + tables", "list / cell / origin", "window staging", "attributes", "|x|^2", "fp64 MFMA (wave 0)", "reduce", "tail"]
for label, pcd, inp, rolls in (("C3", "table1_mult_obj_rcs_1428580506606673.pcd", capi.default_input(grasp_area_length_x=56, grasp_area_length_y=56, grasp_area_center=(0.13, 0.25, 0.0)), 20),
                               ("C2", "pcd2.pcd", capi.default_input(grasp_area_length_x=32, grasp_area_length_y=32), 12)):
    xyz = capi.load_pcd(os.path.join(D, pcd))
    eng = capi.Engine(os.path.join(D, "Features.txt"), os.path.join(D, "range21062012_allfeatures"), os.path.join(ROOT, "tests", "golden", "surrogate.model"),
                      max_points=1 << 18, **(dict(n_rolls=20, roll_step_deg=9) if rolls == 20 else {}))
    rows = []
    for _ in range(8):
        eng.score(xyz, inp)
        out = (ctypes.c_ulonglong * 16)()
        assert lib.haf_phase_clock(out, 16) == 0
        rows.append([out[i] for i in range(9)])
    r = np.array(rows[2:], dtype=np.int64)
    d = np.median(np.diff(r, axis=1), axis=0) / 100.0                      # us
    print(label, "k_small_direct workgroup 0, us per phase (median of 6 requests): total %.1f" % d.sum())
    for n, v in zip(names, d):
        print("   %-24s %6.2f" % (n, v))
    eng.close()
